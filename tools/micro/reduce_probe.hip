// micro-benchmark: how long does a column reduction over R partial rows take as a function of the number of independent jobs,
// the workgroup shape and the data's residency (hot = just written, cold = 1 GB streamed in between)?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/reduce_probe.hip -o gpurun_out/reduce_probe && gpurun_out/reduce_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Job { const float* part; float* out; int R, N; };
struct Jobs { Job job[6]; int tile_start[7]; int count; };

template <int LANES>   // 16 column quads x LANES row lanes
__global__ __launch_bounds__(16 * LANES) void red_kernel(Jobs jobs) {
    __shared__ f32x4 l[LANES][17];
    int j = 0;
    while (j + 1 < jobs.count && (int)blockIdx.x >= jobs.tile_start[j + 1]) ++j;
    const Job jb = jobs.job[j];
    const int cq = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int i = (blockIdx.x - jobs.tile_start[j]) * 64 + cq * 4, P = jb.R;
    const float* base = jb.part + (size_t)(i >> 6) * P * 64 + (i & 63);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int p = lane; p < P; p += LANES * 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (p + LANES * u < P) ? *(const f32x4*)(base + (size_t)(p + LANES * u) * 64) : (f32x4){0.f, 0.f, 0.f, 0.f};
        s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    l[lane][cq] = s;
    __syncthreads();
    if (lane != 0) return;
    for (int k = 1; k < LANES; ++k) s += l[k][cq];
    *(f32x4*)(jb.out + i) = s;
}
__global__ void stream_kernel(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.f;
}
int main() {
    const int R = 512, N = 2304, NJ = 6;
    float *part, *out, *big;
    hipMalloc(&part, (size_t)NJ * R * N * 4); hipMalloc(&out, NJ * N * 4); hipMalloc(&big, (size_t)1 << 30);
    hipMemset(part, 0, (size_t)NJ * R * N * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int cold = 0; cold < 2; ++cold)
        for (int lanes : {16, 64})
            for (int nj = 1; nj <= NJ; nj += (nj < 2 ? 1 : 2)) {
                Jobs jobs; jobs.count = nj; jobs.tile_start[0] = 0;
                for (int k = 0; k < nj; ++k) { jobs.job[k] = {part + (size_t)k * R * N, out + k * N, R, N}; jobs.tile_start[k + 1] = jobs.tile_start[k] + N / 64; }
                float tot = 0;
                const int reps = 20;
                for (int r = 0; r < reps + 2; ++r) {
                    if (cold) hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, 0, big, (size_t)1 << 28);
                    hipEventRecord(a);
                    if (lanes == 16) hipLaunchKernelGGL(red_kernel<16>, dim3(jobs.tile_start[nj]), dim3(256), 0, 0, jobs);
                    else hipLaunchKernelGGL(red_kernel<64>, dim3(jobs.tile_start[nj]), dim3(1024), 0, 0, jobs);
                    hipEventRecord(b); hipEventSynchronize(b);
                    float ms; hipEventElapsedTime(&ms, a, b);
                    if (r >= 2) tot += ms;
                }
                printf("%s lanes %2d jobs %d (%5.1f MB, %4d workgroups): %7.2f us\n", cold ? "cold" : "hot ", lanes, nj, nj * R * N * 4 / 1e6, jobs.tile_start[nj], tot / reps * 1e3);
            }
    return 0;
}
