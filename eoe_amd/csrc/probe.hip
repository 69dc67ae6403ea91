// Box calibration probes (round 5): two fixed kernels bench.py times in front of the timed region, so that the driver's number can be read
// against the box it ran on (MI355X boxes of this pool differ by +-4 % on the same code: DESIGN.md section 5).  Neither is on the product path.
#include "common.h"

namespace {

// bare MFMA loop: every wave keeps 8 independent 16x16 accumulators and issues `iters` rounds of 8 v_mfma_f32_16x16x32_f16 on register
// operands (non-trivial values: the chip's clock under an MFMA-dense loop depends on the data, MI355X_MICROARCH.md "DVFS give-back")
__global__ __launch_bounds__(256) void probe_mfma_f16_kernel(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (f16_t)(0.001f * (float)((lane * 7 + i * 13 + (int)blockIdx.x) % 97) - 0.05f);
        b[i] = (f16_t)(0.002f * (float)((lane * 5 + i * 3) % 89) - 0.09f);
    }
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) s += acc[j];
    if (out) out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// streaming copy, 16 bytes per lane and step, grid-stride
__global__ __launch_bounds__(256) void probe_copy_kernel(f32x4* __restrict__ dst, const f32x4* __restrict__ src, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}

}  // namespace

extern "C" int eoe_probe_mfma_f16(float* out, int iters, int blocks, void* stream) {
    EOE_CHECK_ARG(iters > 0 && blocks > 0, "eoe_probe_mfma_f16: iters and blocks must be positive");
    hipLaunchKernelGGL(probe_mfma_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
    EOE_CHECK_LAUNCH("probe_mfma_f16");
    return 0;
}

extern "C" int eoe_probe_copy(void* dst, const void* src, int64_t bytes, void* stream) {
    EOE_CHECK_ARG(dst && src && bytes > 0 && (bytes & 15) == 0 && ((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0, "eoe_probe_copy: 16-byte aligned pointers and size");
    hipLaunchKernelGGL(probe_copy_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (f32x4*)dst, (const f32x4*)src, (size_t)bytes / 16);
    EOE_CHECK_LAUNCH("probe_copy");
    return 0;
}
