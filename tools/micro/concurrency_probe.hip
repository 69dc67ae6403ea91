// do two kernels on two HIP streams run concurrently on this machine?  Each kernel keeps only `wgs` workgroups busy for ~T us.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/concurrency_probe.hip -o gpurun_out/concurrency_probe && gpurun_out/concurrency_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(unsigned long long ticks, int* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (ticks == 1) *sink = 1;
}
int main() {
    int* sink; hipMalloc(&sink, 4);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int wgs : {40, 216, 256, 1024}) {
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9;
            for (int r = 0; r < 5; ++r) {
                hipDeviceSynchronize();
                hipEventRecord(a, s1);
                hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, s1, 100000ull, sink);
                if (mode == 0) hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, s1, 100000ull, sink);
                else {
                    hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, s2, 100000ull, sink);
                    hipEvent_t e; hipEventCreate(&e); hipEventRecord(e, s2); hipStreamWaitEvent(s1, e, 0); hipEventDestroy(e);
                }
                hipEventRecord(b, s1); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("%4d workgroups x 2 kernels, %s: %7.1f us\n", wgs, mode ? "two streams" : "one stream ", best * 1e3);
        }
    }
    return 0;
}
