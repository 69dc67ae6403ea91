// what the chip sustains on a pure v_mfma_f32_16x16x4_f32 loop (no loads, no LDS): the practical ceiling of the exact-fp32 convolutions
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}
template <int WAVES>
void run(int wgs_per_cu) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL((k<WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 4 * 16.0 * iters * WAVES * grid;
    printf("waves/WG %d, WGs/CU %d: %.1f TF\n", WAVES, wgs_per_cu, flops / (ms * 1e-3) / 1e12);
}
int main() { run<4>(1); run<4>(2); run<4>(4); run<8>(2); return 0; }
