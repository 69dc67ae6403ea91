// WideResNet-specific kernels (reference `src/eoe/models/resnet.py:85-109,130-149`, `src/eoe/models/cbam.py:31-107`):
// MaxPool 3x3/2, the CBAM channel gate (global avg+max pool -> shared MLP -> sigmoid -> scale) and spatial gate
// (channel max/mean -> 7x7 conv 2->1 -> BatchNorm -> sigmoid -> scale), residual add + ReLU, global average pool.
// Everything here is HBM-bound elementwise / reduction work on fp32 NHWC activations: 16-B accesses along the channel
// axis, reductions in LDS, no MFMA.  The 3x3 / 1x1 / 7x7-stem convolutions run as im2col + MFMA GEMM (conv.hip, gemm.hip).
#include "common.h"

namespace {

int grid_for(size_t total, int cap = 8192) {
    size_t g = (total + 255) / 256;
    if (g > (size_t)cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

__device__ __forceinline__ float sigmoidf(float v) { return 1.0f / (1.0f + __expf(-v)); }

// ---------------------------------------------------------------------------------------------- MaxPool2d(k, stride, pad)
// out[n,Ho,Wo,C] = max over the window (padding never wins); idx = winning tap ky*k+kx (first maximum, as max_pool2d)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, T* __restrict__ out16,
                                                          uint8_t* __restrict__ idx, int n, int H, int W, int C, int k, int stride,
                                                          int pad, int Ho, int Wo) {
    const int cc = C / 4;
    const size_t total = (size_t)n * Ho * Wo * cc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cc) * 4;
        const size_t op = i / cc;
        const int wo = (int)(op % Wo), ho = (int)((op / Wo) % Ho), img = (int)(op / ((size_t)Wo * Ho));
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int arg[4] = {0, 0, 0, 0};
        for (int tap = 0; tap < k * k; ++tap) {
            const int h = ho * stride + tap / k - pad, w = wo * stride + tap % k - pad;
            if (h < 0 || h >= H || w < 0 || w >= W) continue;
            const f32x4 v = *(const f32x4*)(x + (((size_t)img * H + h) * W + w) * C + c);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (v[r] > best[r]) { best[r] = v[r]; arg[r] = tap; }
        }
        *(f32x4*)(out + op * C + c) = (f32x4){best[0], best[1], best[2], best[3]};
        if (out16) *(u32x2*)(out16 + op * C + c) = pack4<T>(best[0], best[1], best[2], best[3]);
        *(uint32_t*)(idx + op * C + c) = (uint32_t)arg[0] | ((uint32_t)arg[1] << 8) | ((uint32_t)arg[2] << 16) | ((uint32_t)arg[3] << 24);
    }
}
// dx[n,H,W,C] (gather form: every input pixel sums the windows it won)
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ idx,
                                                          float* __restrict__ dx, int n, int H, int W, int C, int k, int stride,
                                                          int pad, int Ho, int Wo) {
    const int cc = C / 4;
    const size_t total = (size_t)n * H * W * cc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cc) * 4;
        const size_t ip = i / cc;
        const int w = (int)(ip % W), h = (int)((ip / W) % H), img = (int)(ip / ((size_t)W * H));
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int tap = 0; tap < k * k; ++tap) {
            const int th = h + pad - tap / k, tw = w + pad - tap % k;
            if (th < 0 || tw < 0 || th % stride || tw % stride) continue;
            const int ho = th / stride, wo = tw / stride;
            if (ho >= Ho || wo >= Wo) continue;
            const size_t o = (((size_t)img * Ho + ho) * Wo + wo) * C + c;
            const uint32_t a = *(const uint32_t*)(idx + o);
            const f32x4 d = *(const f32x4*)(dout + o);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if ((int)((a >> (8 * r)) & 255u) == tap) acc[r] += d[r];
        }
        *(f32x4*)(dx + ip * C + c) = (f32x4){acc[0], acc[1], acc[2], acc[3]};
    }
}

// ---------------------------------------------------------------------------------------------- per-(image, channel) reductions
// MODE 0: pooled[n,0,C] = mean_hw x, pooled[n,1,C] = max_hw x, argmax[n,C] = position of the max (cbam.py:47-55)
// MODE 1: red[n,C] = sum_hw a*b  (gradient of the channel scale)
// grid (n, C / (4*cpb)); 256 threads = cpb thread-columns (4 channels each) x rpb row lanes
template <int MODE>
__global__ __launch_bounds__(256) void chan_reduce_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ o0, int* __restrict__ oarg, int HW, int C, int cpb) {
    __shared__ f32x4 ls[256];
    __shared__ f32x4 lm[256];
    __shared__ int la[256][4];
    const int img = blockIdx.x, rpb = 256 / cpb;
    const int tc = threadIdx.x % cpb, rl = threadIdx.x / cpb;
    const int c = (blockIdx.y * cpb + tc) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int arg[4] = {0, 0, 0, 0};
    const float* pa = a + (size_t)img * HW * C + c;
    const float* pb = MODE == 1 ? b + (size_t)img * HW * C + c : nullptr;
    // one workgroup per CU at C = 64 (grid = n x 1): keep U independent 16-B loads in flight per thread -- with a single
    // outstanding load per thread this pass ran at 1.7 TB/s
    constexpr int U = MODE == 0 ? 8 : 4;
    int hw = rl;
    for (; hw + (U - 1) * rpb < HW; hw += U * rpb) {
        f32x4 v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = *(const f32x4*)(pa + (size_t)(hw + u * rpb) * C);
            if (MODE == 1) w[u] = *(const f32x4*)(pb + (size_t)(hw + u * rpb) * C);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 0) {
                s += v[u];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (v[u][r] > m[r]) { m[r] = v[u][r]; arg[r] = hw + u * rpb; }      // increasing hw: first maximum wins
            } else {
                s += v[u] * w[u];
            }
        }
    }
    for (; hw < HW; hw += rpb) {
        const f32x4 v = *(const f32x4*)(pa + (size_t)hw * C);
        if (MODE == 0) {
            s += v;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (v[r] > m[r]) { m[r] = v[r]; arg[r] = hw; }
        } else {
            s += v * *(const f32x4*)(pb + (size_t)hw * C);
        }
    }
    ls[threadIdx.x] = s;
    if (MODE == 0) {
        lm[threadIdx.x] = m;
#pragma unroll
        for (int r = 0; r < 4; ++r) la[threadIdx.x][r] = arg[r];
    }
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < rpb; ++k) {
            const int t = threadIdx.x + k * cpb;
            s += ls[t];
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = lm[t][r];
                    const int av = la[t][r];
                    if (v > m[r] || (v == m[r] && av < arg[r])) { m[r] = v; arg[r] = av; }
                }
            }
        }
        if (MODE == 0) {
            const float inv = 1.0f / (float)HW;
            *(f32x4*)(o0 + ((size_t)img * 2 + 0) * C + c) = s * inv;
            *(f32x4*)(o0 + ((size_t)img * 2 + 1) * C + c) = m;
#pragma unroll
            for (int r = 0; r < 4; ++r) oarg[(size_t)img * C + c + r] = arg[r];
        } else {
            *(f32x4*)(o0 + (size_t)img * C + c) = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------- channel gate MLP
// one workgroup per image: hidden[n,k,:] = relu(W1 pooled[n,k,:] + b1) for k in {avg, max};
// scale[n,:] = sigmoid(W2 (h_avg + h_max) + 2 b2)   (cbam.py:57-66: the two MLP outputs are summed before the sigmoid)
__global__ __launch_bounds__(256) void cgate_mlp_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                            const float* __restrict__ b1, const float* __restrict__ w2,
                                                            const float* __restrict__ b2, float* __restrict__ hidden,
                                                            float* __restrict__ scale, int C, int Ch) {
    extern __shared__ float lds[];                // [2*C] pooled, [2*Ch] hidden
    float* lp = lds;
    float* lh = lds + 2 * C;
    const int img = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * C; i += 256) lp[i] = pooled[(size_t)img * 2 * C + i];
    __syncthreads();
    for (int o = wave; o < 2 * Ch; o += 4) {
        const int k = o / Ch, j = o % Ch;
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc += w1[(size_t)j * C + c] * lp[k * C + c];
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) acc += __shfl_xor(acc, sft, 64);
        if (lane == 0) {
            const float h = fmaxf(acc + b1[j], 0.f);
            lh[o] = h;
            hidden[(size_t)img * 2 * Ch + o] = h;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float att = 2.f * b2[c];
        for (int j = 0; j < Ch; ++j) att += w2[(size_t)c * Ch + j] * (lh[j] + lh[Ch + j]);
        scale[(size_t)img * C + c] = sigmoidf(att);
    }
}
// backward, per image: datt = dscale * s(1-s) (overwrites dscale); dhidden[n,k,:] = (W2^T datt) * (hidden > 0);
// dpooled[n,k,:] = W1^T dhidden[n,k,:]
__global__ __launch_bounds__(256) void cgate_mlp_bwd_kernel(const float* __restrict__ scale, const float* __restrict__ hidden,
                                                            const float* __restrict__ w1, const float* __restrict__ w2,
                                                            float* __restrict__ dscale, float* __restrict__ dhidden,
                                                            float* __restrict__ dpooled, int C, int Ch, int nz = 1) {
    // nz > 1 (the fused CBAM unit): dscale arrives as nz partial slices [z][n][C] (the pixel range of an image cut over nz workgroups),
    // added here in slice order; slice 0 then holds the result
    extern __shared__ float lds[];                // [C] datt, [2*Ch] dhidden
    float* ld = lds;
    float* lh = lds + C;
    const int img = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float s = scale[(size_t)img * C + c];
        float ds = dscale[(size_t)img * C + c];
        for (int z = 1; z < nz; ++z) ds += dscale[((size_t)z * gridDim.x + img) * C + c];
        const float d = ds * s * (1.f - s);
        ld[c] = d;
        dscale[(size_t)img * C + c] = d;
    }
    __syncthreads();
    for (int j = wave; j < Ch; j += 4) {
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc += w2[(size_t)c * Ch + j] * ld[c];
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) acc += __shfl_xor(acc, sft, 64);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float g = hidden[(size_t)img * 2 * Ch + k * Ch + j] > 0.f ? acc : 0.f;
                lh[k * Ch + j] = g;
                dhidden[(size_t)img * 2 * Ch + k * Ch + j] = g;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int k = i / C, c = i % C;
        float acc = 0.f;
        for (int j = 0; j < Ch; ++j) acc += w1[(size_t)j * C + c] * lh[k * Ch + j];
        dpooled[(size_t)img * 2 * C + i] = acc;
    }
}
// parameter gradients of the MLP: sums over the images.  A workgroup owns 16 (c, j) pairs x 16 image slices (one thread per
// (c, j) walking all n images serially was a 100-us chain of dependent L2 loads); the slices are added in a fixed order.
__global__ __launch_bounds__(256) void cgate_mlp_wgrad_kernel(const float* __restrict__ pooled, const float* __restrict__ hidden,
                                                              const float* __restrict__ datt, const float* __restrict__ dhidden,
                                                              float* __restrict__ dw1, float* __restrict__ db1,
                                                              float* __restrict__ dw2, float* __restrict__ db2, int n, int C, int Ch) {
    __shared__ float red[4][16][17];
    const int pl = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + pl;
    const bool ok = i < C * Ch;
    const int j = ok ? i % Ch : 0, c = ok ? i / Ch : 0;
    float a1 = 0.f, a2 = 0.f, s1 = 0.f, s2 = 0.f;
    if (ok)
        for (int img = slice; img < n; img += 16) {
            const float da = datt[(size_t)img * C + c];
            const float* h = hidden + (size_t)img * 2 * Ch;
            const float* dh = dhidden + (size_t)img * 2 * Ch;
            const float* p = pooled + (size_t)img * 2 * C;
            a2 += da * (h[j] + h[Ch + j]);
            a1 += dh[j] * p[c] + dh[Ch + j] * p[C + c];
            s2 += da;
            s1 += dh[j] + dh[Ch + j];
        }
    red[0][slice][pl] = a1;
    red[1][slice][pl] = a2;
    red[2][slice][pl] = s1;
    red[3][slice][pl] = s2;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int which = threadIdx.x >> 4;               // pl = threadIdx.x & 15 as above
    if (!ok) return;
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += red[which][k][pl];
    if (which == 0) dw1[(size_t)j * C + c] = s;
    else if (which == 1) dw2[(size_t)c * Ch + j] = s;
    else if (which == 2) { if (c == 0) db1[j] = s; }
    else if (j == 0) db2[c] = 2.f * s;
}

// ---------------------------------------------------------------------------------------------- gate application
// out = x * sc[img, c] (channel gate) or x * sp[img, hw] (spatial gate)
__global__ __launch_bounds__(256) void gate_scale_kernel(const float* __restrict__ x, const float* __restrict__ sc,
                                                         const float* __restrict__ sp, float* __restrict__ out, int n, int HW, int C) {
    const int cc = C / 4;
    const size_t total = (size_t)n * HW * cc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cc) * 4;
        const size_t p = i / cc;
        f32x4 v = *(const f32x4*)(x + p * C + c);
        if (sc) v *= *(const f32x4*)(sc + (p / HW) * C + c);
        else v *= sp[p];
        *(f32x4*)(out + p * C + c) = v;
    }
}
// spatial gate + residual junction in one pass (resnet.py:143-147): out = relu(x * sp[img, hw] + res), optional 16-bit copy
template <typename T>
__global__ __launch_bounds__(256) void gate_scale_add_relu_kernel(const float* __restrict__ x, const float* __restrict__ sp,
                                                                  const float* __restrict__ res, float* __restrict__ out,
                                                                  T* __restrict__ out16, size_t P, int C) {
    const int cc = C / 4;
    const size_t total = P * cc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cc) * 4;
        const size_t p = i / cc;
        f32x4 v = *(const f32x4*)(x + p * C + c) * sp[p] + *(const f32x4*)(res + p * C + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        *(f32x4*)(out + p * C + c) = v;
        if (out16) *(u32x2*)(out16 + p * C + c) = pack4<T>(v[0], v[1], v[2], v[3]);
    }
}
// channel gate: dx = dout * s[img,c] + dpooled_avg[img,c] / HW + [hw == argmax[img,c]] * dpooled_max[img,c]
__global__ __launch_bounds__(256) void cgate_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ scale,
                                                              const float* __restrict__ dpooled, const int* __restrict__ argmax,
                                                              float* __restrict__ dx, int n, int HW, int C) {
    const int cc = C / 4;
    const size_t total = (size_t)n * HW * cc;
    const float inv = 1.0f / (float)HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cc) * 4;
        const size_t p = i / cc;
        const size_t img = p / HW;
        const int hw = (int)(p % HW);
        const f32x4 d = *(const f32x4*)(dout + p * C + c);
        const f32x4 s = *(const f32x4*)(scale + img * C + c);
        const f32x4 da = *(const f32x4*)(dpooled + (img * 2 + 0) * C + c);
        const f32x4 dm = *(const f32x4*)(dpooled + (img * 2 + 1) * C + c);
        f32x4 o = d * s + da * inv;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (argmax[img * C + c + r] == hw) o[r] += dm[r];
        *(f32x4*)(dx + p * C + c) = o;
    }
}

// ---------------------------------------------------------------------------------------------- spatial gate
// 16 lanes per pixel.  MODE 0: comp[p,0] = max_c x, comp[p,1] = mean_c x, argmax[p] = channel of the max (cbam.py:76-79);
// MODE 1: o0[p] = sum_c a*b  (gradient of the spatial scale)
template <int MODE>
__global__ __launch_bounds__(256) void pix_reduce_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ o0, int* __restrict__ oarg, size_t P, int C) {
    const int sub = threadIdx.x & 15;
    for (size_t p = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4); p < (P + 15) / 16 * 16; p += (size_t)gridDim.x * 16) {
        const bool live = p < P;
        float s = 0.f, m = -INFINITY;
        int arg = 0;
        if (live) {
            for (int c = sub * 4; c < C; c += 64) {
                const f32x4 v = *(const f32x4*)(a + p * C + c);
                if (MODE == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s += v[r];
                        if (v[r] > m) { m = v[r]; arg = c + r; }
                    }
                } else {
                    const f32x4 w = *(const f32x4*)(b + p * C + c);
                    s += v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
                }
            }
        }
#pragma unroll
        for (int sft = 8; sft >= 1; sft >>= 1) {
            s += __shfl_xor(s, sft, 16);
            if (MODE == 0) {
                const float om = __shfl_xor(m, sft, 16);
                const int oa = __shfl_xor(arg, sft, 16);
                if (om > m || (om == m && oa < arg)) { m = om; arg = oa; }
            }
        }
        if (live && sub == 0) {
            if (MODE == 0) {
                o0[p * 2 + 0] = m;
                o0[p * 2 + 1] = s / (float)C;
                oarg[p] = arg;
            } else {
                o0[p] = s;
            }
        }
    }
}
// z[n,H,W] = conv7x7(comp; w[1,2,7,7], pad 3), direct (98 MACs per pixel, operands in L1/L2)
__global__ __launch_bounds__(256) void sgate_conv_fwd_kernel(const float* __restrict__ comp, const float* __restrict__ w,
                                                             float* __restrict__ z, int n, int H, int W) {
    __shared__ float lw[98];
    if (threadIdx.x < 98) lw[threadIdx.x] = w[threadIdx.x];
    __syncthreads();
    const size_t total = (size_t)n * H * W;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(p % W), y = (int)((p / W) % H);
        const size_t base = p - (size_t)y * W - x;
        // (round 4: every tap's load unconditional -- a tap outside the map re-reads the pixel itself and is dropped; behind `continue` each of
        //  the 49 loads was waited for where it was issued.  Same taps added in the same order.)
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            const int yy = y + ky - 3;
            const bool oky = yy >= 0 && yy < H;
            f32x2 v[7];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const int xx = x + kx - 3;
                const bool ok = oky && xx >= 0 && xx < W;
                v[kx] = *(const f32x2*)(comp + (ok ? base + (size_t)yy * W + xx : p) * 2);
            }
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const int xx = x + kx - 3;
                if (oky && xx >= 0 && xx < W) acc += v[kx][0] * lw[ky * 7 + kx] + v[kx][1] * lw[49 + ky * 7 + kx];
            }
        }
        z[p] = acc;
    }
}
// scale[p] = sigmoid(gamma * (z - mean) * rstd + beta)
__global__ __launch_bounds__(256) void sgate_sigmoid_kernel(const float* __restrict__ z, const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ scale, size_t P) {
    const float mu = stats[0], rs = stats[1], g = gamma ? gamma[0] : 1.f, b = beta ? beta[0] : 0.f;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (size_t)gridDim.x * blockDim.x)
        scale[p] = sigmoidf((z[p] - mu) * rs * g + b);
}
// g[p] = dscale[p] * s(1-s) (overwrites dscale); workgroup b writes (sum g, sum g*xhat) over its pixels to red[2 + 2b ..]
// (no atomics, no zero-init: sgate_bn_bwd_apply_kernel adds the pairs up in a fixed order)
__global__ __launch_bounds__(256) void sgate_bn_bwd_reduce_kernel(const float* __restrict__ z, const float* __restrict__ stats,
                                                                  const float* __restrict__ scale, float* __restrict__ dscale,
                                                                  float* __restrict__ red, size_t P) {
    __shared__ float l0[4], l1[4];
    const float mu = stats[0], rs = stats[1];
    float s0 = 0.f, s1 = 0.f;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (size_t)gridDim.x * blockDim.x) {
        const float s = scale[p];
        const float g = dscale[p] * s * (1.f - s);
        dscale[p] = g;
        s0 += g;
        s1 += g * (z[p] - mu) * rs;
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) { s0 += __shfl_xor(s0, sft, 64); s1 += __shfl_xor(s1, sft, 64); }
    if ((threadIdx.x & 63) == 0) { l0[threadIdx.x >> 6] = s0; l1[threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        red[2 + 2 * blockIdx.x] = l0[0] + l0[1] + l0[2] + l0[3];
        red[3 + 2 * blockIdx.x] = l1[0] + l1[1] + l1[2] + l1[3];
    }
}
// dz[p] = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)) (training) or gamma*rstd*g (running statistics); in place over g
__global__ __launch_bounds__(256) void sgate_bn_bwd_apply_kernel(const float* __restrict__ z, const float* __restrict__ stats,
                                                                 const float* __restrict__ gamma, const float* __restrict__ red,
                                                                 int nparts, float* __restrict__ g, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, size_t P, int training) {
    __shared__ float l0[4], l1[4];
    const float mu = stats[0], rs = stats[1], ga = gamma ? gamma[0] : 1.f;
    // every workgroup adds the nparts partial pairs in the same order (a few KB from L2)
    float t0 = 0.f, t1 = 0.f;
    for (int k = threadIdx.x; k < nparts; k += blockDim.x) { t0 += red[2 + 2 * k]; t1 += red[3 + 2 * k]; }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) { t0 += __shfl_xor(t0, sft, 64); t1 += __shfl_xor(t1, sft, 64); }
    if ((threadIdx.x & 63) == 0) { l0[threadIdx.x >> 6] = t0; l1[threadIdx.x >> 6] = t1; }
    __syncthreads();
    float r0 = (l0[0] + l0[1]) + (l0[2] + l0[3]), r1 = (l1[0] + l1[1]) + (l1[2] + l1[3]);
    float cnt = (float)P;
    // nparts < 0: synchronised BatchNorm -- red[0..2] = (sum g, sum g*xhat, pixels) already totalled over the ranks, this rank's own
    // sums (the parameter gradients) in red[3..4] (sgate_bn_bwd_total_kernel)
    if (nparts < 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0 && dgamma) { dgamma[0] = red[4]; dbeta[0] = red[3]; }
        r0 = red[0]; r1 = red[1]; cnt = red[2];
    } else if (blockIdx.x == 0 && threadIdx.x == 0 && dgamma) { dgamma[0] = r1; dbeta[0] = r0; }
    const float m0 = r0 / cnt, m1 = r1 / cnt;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (size_t)gridDim.x * blockDim.x) {
        const float xh = (z[p] - mu) * rs;
        g[p] = training ? ga * rs * (g[p] - m0 - xh * m1) : ga * rs * g[p];
    }
}
// synchronised BatchNorm: the partial pairs -> red[0..2] = (sum g, sum g*xhat, P) for the all-reduce, red[3..4] = a copy that stays local
__global__ __launch_bounds__(256) void sgate_bn_bwd_total_kernel(float* __restrict__ red, int nparts, float P) {
    __shared__ float l0[4], l1[4];
    float t0 = 0.f, t1 = 0.f;
    for (int k = threadIdx.x; k < nparts; k += blockDim.x) { t0 += red[2 + 2 * k]; t1 += red[3 + 2 * k]; }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) { t0 += __shfl_xor(t0, sft, 64); t1 += __shfl_xor(t1, sft, 64); }
    if ((threadIdx.x & 63) == 0) { l0[threadIdx.x >> 6] = t0; l1[threadIdx.x >> 6] = t1; }
    __syncthreads();                 // every partial pair has been read: the head of red may be overwritten
    if (threadIdx.x == 0) {
        const float r0 = (l0[0] + l0[1]) + (l0[2] + l0[3]), r1 = (l1[0] + l1[1]) + (l1[2] + l1[3]);
        red[0] = r0; red[1] = r1; red[2] = P; red[3] = r0; red[4] = r1;
    }
}
// dcomp[p, ch] = sum_taps dz[y-ky+3, x-kx+3] * w[ch, ky, kx]
// (pack != 0, the fused CBAM unit: dcomp holds FOUR floats per pixel -- {scale[p], dcomp0, dcomp1, argmax[p] as bits} -- so that the two passes
//  that rebuild the spatial gate's input gradient fetch a pixel's scalars with one 16-byte load instead of three narrow ones)
__global__ __launch_bounds__(256) void sgate_conv_bwd_data_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                                  float* __restrict__ dcomp, int n, int H, int W, int pack = 0,
                                                                  const float* __restrict__ sp = nullptr, const int* __restrict__ am = nullptr) {
    __shared__ float lw[98];
    if (threadIdx.x < 98) lw[threadIdx.x] = w[threadIdx.x];
    __syncthreads();
    const size_t total = (size_t)n * H * W;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(p % W), y = (int)((p / W) % H);
        const size_t base = p - (size_t)y * W - x;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {                   // (loads unconditional, as in sgate_conv_fwd_kernel)
            const int yy = y - ky + 3;
            const bool oky = yy >= 0 && yy < H;
            float d[7];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const int xx = x - kx + 3;
                d[kx] = dz[(oky && xx >= 0 && xx < W) ? base + (size_t)yy * W + xx : p];
            }
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const int xx = x - kx + 3;
                if (oky && xx >= 0 && xx < W) {
                    a0 += d[kx] * lw[ky * 7 + kx];
                    a1 += d[kx] * lw[49 + ky * 7 + kx];
                }
            }
        }
        if (pack) *(f32x4*)(dcomp + p * 4) = (f32x4){sp[p], a0, a1, __int_as_float(am[p])};
        else *(f32x2*)(dcomp + p * 2) = (f32x2){a0, a1};
    }
}
// dw[ch, ky, kx] = sum_p dz[p] * comp[(y+ky-3, x+kx-3), ch].  One workgroup per image: the image's comp plane sits in LDS with
// a 3-pixel zero halo (no bounds tests in the tap loop), a thread walks its pixels with all 98 taps in registers (49 8-byte
// LDS reads + 98 FMAs per pixel; the first version ran one workgroup per (tap, pixel chunk) and re-read dz / comp 98 times
// through L2: 52 us average).  The 256 x 98 per-thread sums are added through LDS in two halves of 49 taps, in a fixed
// order, into part[img][98]; sgate_wpart_sum_kernel adds the images up -- no atomics.
__global__ __launch_bounds__(256) void sgate_conv_bwd_weight_kernel(const float* __restrict__ dz, const float* __restrict__ comp,
                                                                    float* __restrict__ part, int H, int W) {
    extern __shared__ float lc[];                  // [(H+6)*(W+6)][2] padded plane, then reused as [256][49] + [4][49]
    const int img = blockIdx.x, HW = H * W, Wp = W + 6, plane = (H + 6) * Wp * 2;
    for (int i = threadIdx.x; i < plane; i += blockDim.x) lc[i] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        const int y = i / W, x = i - y * W;
        *(f32x2*)(lc + ((y + 3) * Wp + x + 3) * 2) = *(const f32x2*)(comp + ((size_t)img * HW + i) * 2);
    }
    __syncthreads();
    float acc[98];
#pragma unroll
    for (int t = 0; t < 98; ++t) acc[t] = 0.f;
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        const int y = p / W, x = p - y * W;
        const float d = dz[(size_t)img * HW + p];
        const float* base = lc + (y * Wp + x) * 2;         // window origin (y-3, x-3) in padded coordinates
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const f32x2 c = *(const f32x2*)(base + (ky * Wp + kx) * 2);
                acc[ky * 7 + kx] += d * c[0];
                acc[49 + ky * 7 + kx] += d * c[1];
            }
    }
    float* stage = lc;                             // [256][49]
    float* quarter = lc + 256 * 49;                // [4][49]
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                           // plane (half 0) / previous stage (half 1) fully consumed
#pragma unroll
        for (int t = 0; t < 49; ++t) stage[threadIdx.x * 49 + t] = acc[half * 49 + t];
        __syncthreads();
        if (threadIdx.x < 196) {                   // 4 row quarters x 49 taps
            const int t = threadIdx.x % 49, q = threadIdx.x / 49;
            float v = 0.f;
            for (int r = q * 64; r < q * 64 + 64; ++r) v += stage[r * 49 + t];
            quarter[q * 49 + t] = v;
        }
        __syncthreads();
        if (threadIdx.x < 49)
            part[(size_t)img * 98 + half * 49 + threadIdx.x] =
                (quarter[threadIdx.x] + quarter[49 + threadIdx.x]) + (quarter[98 + threadIdx.x] + quarter[147 + threadIdx.x]);
    }
}
// dw[t] = sum_img part[img][t]: one wavefront per tap
__global__ __launch_bounds__(64) void sgate_wpart_sum_kernel(const float* __restrict__ part, float* __restrict__ dw, int n) {
    float v = 0.f;
    for (int img = threadIdx.x; img < n; img += 64) v += part[(size_t)img * 98 + blockIdx.x];
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft, 64);
    if (threadIdx.x == 0) dw[blockIdx.x] = v;
}
// dx = dout * scale[p] + [c == argmax[p]] * dcomp[p,0] + dcomp[p,1] / C
__global__ __launch_bounds__(256) void sgate_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ scale,
                                                              const float* __restrict__ dcomp, const int* __restrict__ argmax,
                                                              float* __restrict__ dx, size_t P, int C) {
    const int cc = C / 4;
    const size_t total = P * cc;
    const float inv = 1.0f / (float)C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cc) * 4;
        const size_t p = i / cc;
        const f32x4 d = *(const f32x4*)(dout + p * C + c);
        const f32x2 dc = *(const f32x2*)(dcomp + p * 2);
        const int am = argmax[p];
        f32x4 o = d * scale[p] + dc[1] * inv;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (c + r == am) o[r] += dc[0];
        *(f32x4*)(dx + p * C + c) = o;
    }
}

// ---------------------------------------------------------------------------------------------- residual add + ReLU, avg pool
template <typename T>
__global__ __launch_bounds__(256) void add_relu_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ out, T* __restrict__ out16, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 v = ((const f32x4*)a)[i] + ((const f32x4*)b)[i];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        ((f32x4*)out)[i] = v;
        if (out16) ((u32x2*)out16)[i] = pack4<T>(v[0], v[1], v[2], v[3]);
    }
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                       float* __restrict__ g, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 d = ((const f32x4*)dout)[i];
        const f32x4 o = ((const f32x4*)out)[i];
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = o[r] > 0.f ? d[r] : 0.f;
        ((f32x4*)g)[i] = d;
    }
}
// dx[n,hw,c] = dout[n,c] / HW
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int n, int HW, int C) {
    const int cc = C / 4;
    const size_t total = (size_t)n * HW * cc;
    const float inv = 1.0f / (float)HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cc) * 4;
        const size_t p = i / cc;
        *(f32x4*)(dx + p * C + c) = *(const f32x4*)(dout + (p / HW) * C + c) * inv;
    }
}


// ---------------------------------------------------------------------------------------------- CBAM + residual junction, fused (round 3)
// A BasicBlock's tail is out = relu(sgate(cgate(x)) + res) (resnet.py:143-147, cbam.py:100-106).  Run as separate units it makes 7.5 passes
// over the activation forward (30 B per element) and 11 backward (44 B): the channel-gated tensor xc = x * sc is written, read again by the
// spatial pooling, by the junction, and twice in backward, and the junction gradient is a pass of its own.  Here xc is never written:
// every consumer multiplies x by sc[img, c] on the fly (the same single fp32 product: the results keep their bits), the junction's
// ReLU mask is applied where the gradient is first read, and the spatial gate's backward feeds the channel gate's reduction and its
// final pass without materialising its output -- 22 B forward, 32 B backward per element.

// forward: comp[p] = (max_c, mean_c) of x * sc, argmax[p]; 16 lanes per pixel
__global__ __launch_bounds__(256) void cbam_pix_pool_kernel(const float* __restrict__ x, const float* __restrict__ sc, float* __restrict__ comp,
                                                            int* __restrict__ oarg, size_t P, int HW, int C) {
    const int sub = threadIdx.x & 15;
    for (size_t p = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4); p < (P + 15) / 16 * 16; p += (size_t)gridDim.x * 16) {
        const bool live = p < P;
        float s = 0.f, m = -INFINITY;
        int arg = 0;
        if (live) {
            const float* scp = sc + (p / HW) * C;
            for (int c = sub * 4; c < C; c += 64) {
                const f32x4 v = *(const f32x4*)(x + p * C + c) * *(const f32x4*)(scp + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s += v[r];
                    if (v[r] > m) { m = v[r]; arg = c + r; }
                }
            }
        }
#pragma unroll
        for (int sft = 8; sft >= 1; sft >>= 1) {
            s += __shfl_xor(s, sft, 16);
            const float om = __shfl_xor(m, sft, 16);
            const int oa = __shfl_xor(arg, sft, 16);
            if (om > m || (om == m && oa < arg)) { m = om; arg = oa; }
        }
        if (live && sub == 0) {
            comp[p * 2 + 0] = m;
            comp[p * 2 + 1] = s / (float)C;
            oarg[p] = arg;
        }
    }
}
// forward: out = relu((x * sc[img, c]) * sp[p] + res) (+ 16-bit copy)
template <typename T>
__global__ __launch_bounds__(256) void cbam_apply_add_relu_kernel(const float* __restrict__ x, const float* __restrict__ sc,
                                                                  const float* __restrict__ sp, const float* __restrict__ res,
                                                                  float* __restrict__ out, T* __restrict__ out16, size_t P, int HW, int C) {
    const int cc = C / 4;
    const size_t total = P * cc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cc) * 4;
        const size_t p = i / cc;
        const f32x4 xc = *(const f32x4*)(x + p * C + c) * *(const f32x4*)(sc + (p / HW) * C + c);
        f32x4 v = xc * sp[p] + *(const f32x4*)(res + p * C + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        *(f32x4*)(out + p * C + c) = v;
        if (out16) *(u32x2*)(out16 + p * C + c) = pack4<T>(v[0], v[1], v[2], v[3]);
    }
}
// backward 1: g = dout * [out > 0] (written: it is also the shortcut's gradient), dsp[p] = sum_c g * (x * sc); 16 lanes per pixel
__global__ __launch_bounds__(256) void cbam_relu_pix_reduce_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                                   const float* __restrict__ x, const float* __restrict__ sc,
                                                                   float* __restrict__ g, float* __restrict__ dsp, size_t P, int HW, int C) {
    const int sub = threadIdx.x & 15;
    for (size_t p = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4); p < (P + 15) / 16 * 16; p += (size_t)gridDim.x * 16) {
        const bool live = p < P;
        float s = 0.f;
        if (live) {
            const float* scp = sc + (p / HW) * C;
            for (int c = sub * 4; c < C; c += 64) {
                f32x4 d = *(const f32x4*)(dout + p * C + c);
                const f32x4 o = *(const f32x4*)(out + p * C + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r] = o[r] > 0.f ? d[r] : 0.f;
                *(f32x4*)(g + p * C + c) = d;
                const f32x4 w = *(const f32x4*)(x + p * C + c) * *(const f32x4*)(scp + c);
                s += d[0] * w[0] + d[1] * w[1] + d[2] * w[2] + d[3] * w[3];
            }
        }
#pragma unroll
        for (int sft = 8; sft >= 1; sft >>= 1) s += __shfl_xor(s, sft, 16);
        if (live && sub == 0) dsp[p] = s;
    }
}
// the spatial gate's input gradient at (p, c..c+3), never stored: dxc = g * sp[p] + [c == argmax_p] * dcomp[p, 0] + dcomp[p, 1] / C;
// pk[p] = {sp, dcomp0, dcomp1, argmax bits} (sgate_conv_bwd_data_kernel, pack)
__device__ __forceinline__ f32x4 cbam_dxc(const float* __restrict__ g, const float* __restrict__ pk, size_t p, int c, int C, float invC) {
    const f32x4 d = *(const f32x4*)(g + p * C + c);
    const f32x4 q = *(const f32x4*)(pk + p * 4);
    const int am = __float_as_int(q[3]);
    f32x4 o = d * q[0] + q[2] * invC;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (c + r == am) o[r] += q[1];
    return o;
}
// backward 2: dsc[img, c] = sum_hw dxc * x   (the layout of chan_reduce_kernel<1>)
// (grid.z workgroups share an image's pixels -- one workgroup per image is one per CU at 256 images: too few loads in flight -- and write
//  partial slices dsc[z][img][C], added in slice order by cgate_mlp_bwd_kernel)
__global__ __launch_bounds__(256) void cbam_chan_reduce_kernel(const float* __restrict__ g, const float* __restrict__ pk,
                                                               const float* __restrict__ x, float* __restrict__ dsc, int HW_all, int C, int cpb) {
    __shared__ f32x4 ls[256];
    const int img = blockIdx.x, rpb = 256 / cpb;
    const int tc = threadIdx.x % cpb, rl = threadIdx.x / cpb;
    const int c = (blockIdx.y * cpb + tc) * 4;
    const float invC = 1.0f / (float)C;
    const int per = (HW_all + (int)gridDim.z - 1) / (int)gridDim.z, hw_lo = (int)blockIdx.z * per;
    const int HW = min(HW_all, hw_lo + per);                   // this workgroup's pixels: [hw_lo, HW)
    const size_t p0 = (size_t)img * HW_all;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;
    int hw = hw_lo + rl;
    for (; hw + (U - 1) * rpb < HW; hw += U * rpb) {
        f32x4 v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = cbam_dxc(g, pk, p0 + hw + u * rpb, c, C, invC);
            w[u] = *(const f32x4*)(x + (p0 + hw + u * rpb) * C + c);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u] * w[u];
    }
    for (; hw < HW; hw += rpb) s += cbam_dxc(g, pk, p0 + hw, c, C, invC) * *(const f32x4*)(x + (p0 + hw) * C + c);
    ls[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < rpb; ++k) s += ls[threadIdx.x + k * cpb];
        *(f32x4*)(dsc + ((size_t)blockIdx.z * gridDim.x + img) * C + c) = s;
    }
}
// backward 3: dx = dxc * sc[img, c] + dpooled_avg[img, c] / HW + [hw == argmax_c[img, c]] * dpooled_max[img, c]
__global__ __launch_bounds__(256) void cbam_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ pk,
                                                             const float* __restrict__ sc, const float* __restrict__ dpooled,
                                                             const int* __restrict__ am_c, float* __restrict__ dx, int n, int HW, int C) {
    const int cc = C / 4;
    const size_t total = (size_t)n * HW * cc;
    const float inv = 1.0f / (float)HW, invC = 1.0f / (float)C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cc) * 4;
        const size_t p = i / cc;
        const size_t img = p / HW;
        const int hw = (int)(p % HW);
        const f32x4 d = cbam_dxc(g, pk, p, c, C, invC);
        const f32x4 s = *(const f32x4*)(sc + img * C + c);
        const f32x4 da = *(const f32x4*)(dpooled + (img * 2 + 0) * C + c);
        const f32x4 dm = *(const f32x4*)(dpooled + (img * 2 + 1) * C + c);
        f32x4 o = d * s + da * inv;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (am_c[img * C + c + r] == hw) o[r] += dm[r];
        *(f32x4*)(dx + p * C + c) = o;
    }
}

int chan_cpb(int C) { return C / 4 < 64 ? C / 4 : 64; }

}  // namespace

#define DISPATCH_T(dtype, ...)                                  \
    do {                                                        \
        if ((dtype) == EOE_F16) { typedef f16_t T; __VA_ARGS__; } \
        else if ((dtype) == EOE_BF16) { typedef bf16_t T; __VA_ARGS__; } \
        else return eoe_set_error(EOE_ERR_ARG, "bad dtype %d", (int)(dtype)); \
    } while (0)

extern "C" int eoe_maxpool_fwd(const float* x, float* out, void* out16, uint8_t* idx, int n, int H, int W, int C, int k, int stride,
                               int pad, int dtype, void* stream) {
    EOE_CHECK_ARG(x && out && idx && n > 0 && C % 4 == 0 && k >= 1 && k * k <= 255 && stride >= 1 && pad >= 0 && 2 * pad <= k,
                  "maxpool_fwd: bad args");
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    EOE_CHECK_ARG(Ho >= 1 && Wo >= 1, "maxpool_fwd: empty output");
    ProfScope ps("maxpool_fwd", 0, 4.0 * n * H * W * C + 5.0 * n * Ho * Wo * C, stream);
    DISPATCH_T(dtype, hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3(grid_for((size_t)n * Ho * Wo * C / 4)), dim3(256), 0,
                                         (hipStream_t)stream, x, out, (T*)out16, idx, n, H, W, C, k, stride, pad, Ho, Wo));
    EOE_CHECK_LAUNCH("maxpool_fwd");
    return 0;
}

extern "C" int eoe_maxpool_bwd(const float* dout, const uint8_t* idx, float* dx, int n, int H, int W, int C, int k, int stride,
                               int pad, void* stream) {
    EOE_CHECK_ARG(dout && dx && idx && n > 0 && C % 4 == 0 && k >= 1 && stride >= 1 && pad >= 0, "maxpool_bwd: bad args");
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    ProfScope ps("maxpool_bwd", 0, 4.0 * n * H * W * C + 5.0 * n * Ho * Wo * C, stream);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((size_t)n * H * W * C / 4)), dim3(256), 0, (hipStream_t)stream, dout, idx,
                       dx, n, H, W, C, k, stride, pad, Ho, Wo);
    EOE_CHECK_LAUNCH("maxpool_bwd");
    return 0;
}

static int check_cgate(const eoe_cgate_args* a) {
    EOE_CHECK_ARG(a && a->x && a->w1 && a->b1 && a->w2 && a->b2 && a->pooled && a->argmax && a->hidden && a->scale, "cgate: null args");
    EOE_CHECK_ARG(a->n > 0 && a->HW > 0 && a->C >= 16 && a->C % 16 == 0 && a->C <= 4096 && a->Ch >= 1 && a->Ch <= 256,
                  "cgate: bad shape n=%d HW=%d C=%d Ch=%d", a->n, a->HW, a->C, a->Ch);
    const int cpb = chan_cpb(a->C);
    EOE_CHECK_ARG(256 % cpb == 0 && (a->C / 4) % cpb == 0, "cgate: C = %d not supported", a->C);
    return 0;
}

extern "C" int eoe_cgate_fwd(const eoe_cgate_args* a, void* stream) {
    EOE_TRY(check_cgate(a));
    EOE_CHECK_ARG(a->out != nullptr, "cgate_fwd: null out");
    hipStream_t s = (hipStream_t)stream;
    const int cpb = chan_cpb(a->C);
    ProfScope ps("cgate_fwd", 0, 3 * 4.0 * a->n * a->HW * a->C, stream);
    hipLaunchKernelGGL(chan_reduce_kernel<0>, dim3(a->n, a->C / 4 / cpb), dim3(256), 0, s, a->x, (const float*)nullptr, a->pooled,
                       a->argmax, a->HW, a->C, cpb);
    EOE_CHECK_LAUNCH("cgate_pool");
    hipLaunchKernelGGL(cgate_mlp_fwd_kernel, dim3(a->n), dim3(256), (2 * a->C + 2 * a->Ch) * sizeof(float), s, a->pooled, a->w1, a->b1,
                       a->w2, a->b2, a->hidden, a->scale, a->C, a->Ch);
    EOE_CHECK_LAUNCH("cgate_mlp_fwd");
    hipLaunchKernelGGL(gate_scale_kernel, dim3(grid_for((size_t)a->n * a->HW * a->C / 4)), dim3(256), 0, s, a->x, a->scale,
                       (const float*)nullptr, a->out, a->n, a->HW, a->C);
    EOE_CHECK_LAUNCH("cgate_scale");
    return 0;
}

extern "C" int eoe_cgate_bwd(const eoe_cgate_bwd_args* b, void* stream) {
    EOE_CHECK_ARG(b != nullptr, "cgate_bwd: null args");
    const eoe_cgate_args* a = &b->f;
    EOE_TRY(check_cgate(a));
    EOE_CHECK_ARG(b->dout && b->dx && b->dscale && b->dpooled && b->dhidden && b->dw1 && b->db1 && b->dw2 && b->db2,
                  "cgate_bwd: null args");
    hipStream_t s = (hipStream_t)stream;
    const int cpb = chan_cpb(a->C);
    ProfScope ps("cgate_bwd", 0, 5 * 4.0 * a->n * a->HW * a->C, stream);
    hipLaunchKernelGGL(chan_reduce_kernel<1>, dim3(a->n, a->C / 4 / cpb), dim3(256), 0, s, b->dout, a->x, b->dscale, (int*)nullptr,
                       a->HW, a->C, cpb);
    EOE_CHECK_LAUNCH("cgate_bwd_reduce");
    hipLaunchKernelGGL(cgate_mlp_bwd_kernel, dim3(a->n), dim3(256), (a->C + 2 * a->Ch) * sizeof(float), s, a->scale, a->hidden, a->w1,
                       a->w2, b->dscale, b->dhidden, b->dpooled, a->C, a->Ch);
    EOE_CHECK_LAUNCH("cgate_mlp_bwd");
    hipLaunchKernelGGL(cgate_mlp_wgrad_kernel, dim3(cdiv(a->C * a->Ch, 16)), dim3(256), 0, s, a->pooled, a->hidden, b->dscale,
                       b->dhidden, b->dw1, b->db1, b->dw2, b->db2, a->n, a->C, a->Ch);
    EOE_CHECK_LAUNCH("cgate_mlp_wgrad");
    hipLaunchKernelGGL(cgate_bwd_apply_kernel, dim3(grid_for((size_t)a->n * a->HW * a->C / 4)), dim3(256), 0, s, b->dout, a->scale,
                       b->dpooled, a->argmax, b->dx, a->n, a->HW, a->C);
    EOE_CHECK_LAUNCH("cgate_bwd_apply");
    return 0;
}

static int check_sgate(const eoe_sgate_args* a) {
    EOE_CHECK_ARG(a && a->x && a->w && a->comp && a->argmax && a->z && a->stats && a->scale && a->sums, "sgate: null args");
    EOE_CHECK_ARG((a->gamma == nullptr) == (a->beta == nullptr), "sgate: gamma/beta must both be given or both NULL");
    EOE_CHECK_ARG(a->n > 0 && a->H > 0 && a->W > 0 && a->C >= 4 && a->C % 4 == 0, "sgate: bad shape");
    return 0;
}

extern "C" int eoe_sgate_fwd(const eoe_sgate_args* a, void* stream) {
    EOE_TRY(check_sgate(a));
    EOE_CHECK_ARG(a->out != nullptr, "sgate_fwd: null out");
    hipStream_t s = (hipStream_t)stream;
    const size_t P = (size_t)a->n * a->H * a->W;
    EOE_CHECK_ARG(P < 0x7fffffffull, "sgate: too many pixels");
    {
        ProfScope ps("sgate_fwd", 0, 3 * 4.0 * P * a->C, stream);
        hipLaunchKernelGGL(pix_reduce_kernel<0>, dim3(grid_for(P * 16)), dim3(256), 0, s, a->x, (const float*)nullptr, a->comp,
                           a->argmax, P, a->C);
        EOE_CHECK_LAUNCH("sgate_pool");
        hipLaunchKernelGGL(sgate_conv_fwd_kernel, dim3(grid_for(P)), dim3(256), 0, s, a->comp, a->w, a->z, a->n, a->H, a->W);
        EOE_CHECK_LAUNCH("sgate_conv_fwd");
    }
    EOE_TRY(eoe_bn_stats(a->z, a->sums, a->stats, a->running_mean, a->running_var, a->num_batches_tracked, (int)P, 1, a->eps,
                         a->momentum, a->training, stream));
    ProfScope ps("sgate_fwd", 0, 2 * 4.0 * P * a->C, stream);
    hipLaunchKernelGGL(sgate_sigmoid_kernel, dim3(grid_for(P)), dim3(256), 0, s, a->z, a->stats, a->gamma, a->beta, a->scale, P);
    EOE_CHECK_LAUNCH("sgate_sigmoid");
    if (a->res) {      // fused residual junction: out = relu(x * scale + res) (+ 16-bit copy)
        EOE_CHECK_ARG(!a->out16 || a->dtype == EOE_F16 || a->dtype == EOE_BF16, "sgate_fwd: bad dtype %d", a->dtype);
        if (a->out16 && a->dtype == EOE_BF16)
            hipLaunchKernelGGL((gate_scale_add_relu_kernel<bf16_t>), dim3(grid_for(P * a->C / 4)), dim3(256), 0, s, a->x,
                               (const float*)a->scale, a->res, a->out, (bf16_t*)a->out16, P, a->C);
        else
            hipLaunchKernelGGL((gate_scale_add_relu_kernel<f16_t>), dim3(grid_for(P * a->C / 4)), dim3(256), 0, s, a->x,
                               (const float*)a->scale, a->res, a->out, (f16_t*)a->out16, P, a->C);
        EOE_CHECK_LAUNCH("sgate_scale_add_relu");
        return 0;
    }
    hipLaunchKernelGGL(gate_scale_kernel, dim3(grid_for(P * a->C / 4)), dim3(256), 0, s, a->x, (const float*)nullptr, a->scale, a->out,
                       a->n, a->H * a->W, a->C);
    EOE_CHECK_LAUNCH("sgate_scale");
    return 0;
}

extern "C" int eoe_sgate_bwd(const eoe_sgate_bwd_args* b, void* stream) {
    EOE_CHECK_ARG(b != nullptr, "sgate_bwd: null args");
    const eoe_sgate_args* a = &b->f;
    EOE_TRY(check_sgate(a));
    EOE_CHECK_ARG(b->dout && b->dx && b->dscale && b->dcomp && b->red && b->dw && b->wpart, "sgate_bwd: null args");
    EOE_CHECK_ARG((a->H + 6) * (a->W + 6) <= 8192, "sgate_bwd: feature map too large (%d x %d)", a->H, a->W);
    EOE_CHECK_ARG((b->dgamma == nullptr) == (b->dbeta == nullptr), "sgate_bwd: dgamma/dbeta must both be given or both NULL");
    hipStream_t s = (hipStream_t)stream;
    const size_t P = (size_t)a->n * a->H * a->W;
    ProfScope ps("sgate_bwd", 0, 5 * 4.0 * P * a->C, stream);
    hipLaunchKernelGGL(pix_reduce_kernel<1>, dim3(grid_for(P * 16)), dim3(256), 0, s, b->dout, a->x, b->dscale, (int*)nullptr, P, a->C);
    EOE_CHECK_LAUNCH("sgate_bwd_reduce");
    int g = grid_for(P, EOE_SGATE_PARTIALS);
    hipLaunchKernelGGL(sgate_bn_bwd_reduce_kernel, dim3(g), dim3(256), 0, s, a->z, a->stats, a->scale, b->dscale, b->red, P);
    EOE_CHECK_LAUNCH("sgate_bn_bwd_reduce");
    if (a->training && eoe_bn_sync_active()) {
        hipLaunchKernelGGL(sgate_bn_bwd_total_kernel, dim3(1), dim3(256), 0, s, b->red, g, (float)P);
        EOE_CHECK_LAUNCH("sgate_bn_bwd_total");
        EOE_TRY(eoe_bn_sync_allreduce(b->red, 3, 0, stream));
        g = -1;
    }
    hipLaunchKernelGGL(sgate_bn_bwd_apply_kernel, dim3(grid_for(P)), dim3(256), 0, s, a->z, a->stats, a->gamma, (const float*)b->red, g,
                       b->dscale, b->dgamma, b->dbeta, P, a->training);
    EOE_CHECK_LAUNCH("sgate_bn_bwd_apply");
    hipLaunchKernelGGL(sgate_conv_bwd_data_kernel, dim3(grid_for(P)), dim3(256), 0, s, (const float*)b->dscale, a->w, b->dcomp, a->n,
                       a->H, a->W);
    EOE_CHECK_LAUNCH("sgate_conv_bwd_data");
    {
        size_t lds = (size_t)(a->H + 6) * (a->W + 6) * 2 * sizeof(float);
        if (lds < (256 + 4) * 49 * sizeof(float)) lds = (256 + 4) * 49 * sizeof(float);
        hipLaunchKernelGGL(sgate_conv_bwd_weight_kernel, dim3(a->n), dim3(256), lds, s, (const float*)b->dscale, a->comp, b->wpart, a->H,
                           a->W);
        hipLaunchKernelGGL(sgate_wpart_sum_kernel, dim3(98), dim3(64), 0, s, (const float*)b->wpart, b->dw, a->n);
    }
    EOE_CHECK_LAUNCH("sgate_conv_bwd_weight");
    hipLaunchKernelGGL(sgate_bwd_apply_kernel, dim3(grid_for(P * a->C / 4)), dim3(256), 0, s, b->dout, a->scale, (const float*)b->dcomp,
                       a->argmax, b->dx, P, a->C);
    EOE_CHECK_LAUNCH("sgate_bwd_apply");
    return 0;
}

// CBAM + residual junction in one unit (see the kernels above): cg = the channel gate's arguments (cg->out unused), sg = the spatial
// gate's (sg->x unused: its input x * scale is never written; sg->res / out / out16 as in eoe_sgate_fwd)
extern "C" int eoe_cbam_junction_fwd(const eoe_cgate_args* cg, const eoe_sgate_args* sg, void* stream) {
    EOE_TRY(check_cgate(cg));
    EOE_CHECK_ARG(sg && sg->w && sg->comp && sg->argmax && sg->z && sg->stats && sg->scale && sg->sums && sg->res && sg->out, "cbam_junction_fwd: null args");
    EOE_CHECK_ARG((sg->gamma == nullptr) == (sg->beta == nullptr), "cbam_junction: gamma/beta must both be given or both NULL");
    EOE_CHECK_ARG(sg->n == cg->n && sg->H * sg->W == cg->HW && sg->C == cg->C, "cbam_junction: the two gates disagree on the shape");
    EOE_CHECK_ARG(!sg->out16 || sg->dtype == EOE_F16 || sg->dtype == EOE_BF16, "cbam_junction_fwd: bad dtype %d", sg->dtype);
    hipStream_t s = (hipStream_t)stream;
    const int cpb = chan_cpb(cg->C), C = cg->C, HW = cg->HW;
    const size_t P = (size_t)cg->n * HW;
    EOE_CHECK_ARG(P < 0x7fffffffull, "cbam_junction: too many pixels");
    {
        ProfScope ps("cgate_fwd", 0, 4.0 * P * C, stream);
        hipLaunchKernelGGL(chan_reduce_kernel<0>, dim3(cg->n, C / 4 / cpb), dim3(256), 0, s, cg->x, (const float*)nullptr, cg->pooled,
                           cg->argmax, HW, C, cpb);
        EOE_CHECK_LAUNCH("cbam_chan_pool");
        hipLaunchKernelGGL(cgate_mlp_fwd_kernel, dim3(cg->n), dim3(256), (2 * C + 2 * cg->Ch) * sizeof(float), s, cg->pooled, cg->w1, cg->b1,
                           cg->w2, cg->b2, cg->hidden, cg->scale, C, cg->Ch);
        EOE_CHECK_LAUNCH("cbam_mlp_fwd");
    }
    {
        ProfScope ps("sgate_fwd", 0, 4.0 * P * C, stream);
        hipLaunchKernelGGL(cbam_pix_pool_kernel, dim3(grid_for(P * 16)), dim3(256), 0, s, cg->x, (const float*)cg->scale, sg->comp, sg->argmax,
                           P, HW, C);
        EOE_CHECK_LAUNCH("cbam_pix_pool");
        hipLaunchKernelGGL(sgate_conv_fwd_kernel, dim3(grid_for(P)), dim3(256), 0, s, sg->comp, sg->w, sg->z, sg->n, sg->H, sg->W);
        EOE_CHECK_LAUNCH("cbam_conv_fwd");
    }
    EOE_TRY(eoe_bn_stats(sg->z, sg->sums, sg->stats, sg->running_mean, sg->running_var, sg->num_batches_tracked, (int)P, 1, sg->eps,
                         sg->momentum, sg->training, stream));
    ProfScope ps("sgate_fwd", 0, 14.0 * P * C, stream);
    hipLaunchKernelGGL(sgate_sigmoid_kernel, dim3(grid_for(P)), dim3(256), 0, s, sg->z, sg->stats, sg->gamma, sg->beta, sg->scale, P);
    EOE_CHECK_LAUNCH("cbam_sigmoid");
    if (sg->out16 && sg->dtype == EOE_BF16)
        hipLaunchKernelGGL((cbam_apply_add_relu_kernel<bf16_t>), dim3(grid_for(P * C / 4)), dim3(256), 0, s, cg->x, (const float*)cg->scale,
                           (const float*)sg->scale, sg->res, sg->out, (bf16_t*)sg->out16, P, HW, C);
    else
        hipLaunchKernelGGL((cbam_apply_add_relu_kernel<f16_t>), dim3(grid_for(P * C / 4)), dim3(256), 0, s, cg->x, (const float*)cg->scale,
                           (const float*)sg->scale, sg->res, sg->out, (f16_t*)sg->out16, P, HW, C);
    EOE_CHECK_LAUNCH("cbam_apply_add_relu");
    return 0;
}

// backward of the unit: dout = gradient at the block's output, out = the block's output (the ReLU mask); g (written) = the junction
// gradient = the residual branch's gradient; cb->dx = the gradient of the channel gate's input; sb->dout / sb->dx unused
extern "C" int eoe_cbam_junction_bwd(const eoe_cgate_bwd_args* cb, const eoe_sgate_bwd_args* sb, const float* dout, const float* out, float* g,
                                     void* stream) {
    EOE_CHECK_ARG(cb && sb && dout && out && g, "cbam_junction_bwd: null args");
    const eoe_cgate_args* cg = &cb->f;
    const eoe_sgate_args* sg = &sb->f;
    EOE_TRY(check_cgate(cg));
    EOE_CHECK_ARG(cb->dx && cb->dscale && cb->dpooled && cb->dhidden && cb->dw1 && cb->db1 && cb->dw2 && cb->db2, "cbam_junction_bwd: null cgate args");
    EOE_CHECK_ARG(sg->w && sg->comp && sg->argmax && sg->z && sg->stats && sg->scale && sb->dscale && sb->dcomp && sb->red && sb->dw && sb->wpart,
                  "cbam_junction_bwd: null sgate args");
    EOE_CHECK_ARG((sb->dgamma == nullptr) == (sb->dbeta == nullptr), "cbam_junction_bwd: dgamma/dbeta must both be given or both NULL");
    EOE_CHECK_ARG(sg->n == cg->n && sg->H * sg->W == cg->HW && sg->C == cg->C, "cbam_junction: the two gates disagree on the shape");
    EOE_CHECK_ARG((sg->H + 6) * (sg->W + 6) <= 8192, "cbam_junction_bwd: feature map too large (%d x %d)", sg->H, sg->W);
    hipStream_t s = (hipStream_t)stream;
    const int cpb = chan_cpb(cg->C), C = cg->C, HW = cg->HW;
    const size_t P = (size_t)cg->n * HW;
    {
        ProfScope ps("sgate_bwd", 0, 16.0 * P * C, stream);
        hipLaunchKernelGGL(cbam_relu_pix_reduce_kernel, dim3(grid_for(P * 16)), dim3(256), 0, s, dout, out, cg->x, (const float*)cg->scale, g,
                           sb->dscale, P, HW, C);
        EOE_CHECK_LAUNCH("cbam_relu_pix_reduce");
        int gr = grid_for(P, EOE_SGATE_PARTIALS);
        hipLaunchKernelGGL(sgate_bn_bwd_reduce_kernel, dim3(gr), dim3(256), 0, s, sg->z, sg->stats, sg->scale, sb->dscale, sb->red, P);
        EOE_CHECK_LAUNCH("cbam_bn_bwd_reduce");
        if (sg->training && eoe_bn_sync_active()) {
            hipLaunchKernelGGL(sgate_bn_bwd_total_kernel, dim3(1), dim3(256), 0, s, sb->red, gr, (float)P);
            EOE_CHECK_LAUNCH("cbam_bn_bwd_total");
            EOE_TRY(eoe_bn_sync_allreduce(sb->red, 3, 0, stream));
            gr = -1;
        }
        hipLaunchKernelGGL(sgate_bn_bwd_apply_kernel, dim3(grid_for(P)), dim3(256), 0, s, sg->z, sg->stats, sg->gamma, (const float*)sb->red, gr,
                           sb->dscale, sb->dgamma, sb->dbeta, P, sg->training);
        EOE_CHECK_LAUNCH("cbam_bn_bwd_apply");
        hipLaunchKernelGGL(sgate_conv_bwd_data_kernel, dim3(grid_for(P)), dim3(256), 0, s, (const float*)sb->dscale, sg->w, sb->dcomp, sg->n,
                           sg->H, sg->W, 1, (const float*)sg->scale, (const int*)sg->argmax);      // dcomp: 4 floats per pixel here
        EOE_CHECK_LAUNCH("cbam_conv_bwd_data");
        size_t lds = (size_t)(sg->H + 6) * (sg->W + 6) * 2 * sizeof(float);
        if (lds < (256 + 4) * 49 * sizeof(float)) lds = (256 + 4) * 49 * sizeof(float);
        hipLaunchKernelGGL(sgate_conv_bwd_weight_kernel, dim3(sg->n), dim3(256), lds, s, (const float*)sb->dscale, sg->comp, sb->wpart, sg->H,
                           sg->W);
        hipLaunchKernelGGL(sgate_wpart_sum_kernel, dim3(98), dim3(64), 0, s, (const float*)sb->wpart, sb->dw, sg->n);
        EOE_CHECK_LAUNCH("cbam_conv_bwd_weight");
    }
    ProfScope ps("cgate_bwd", 0, 16.0 * P * C, stream);
    // cb->dscale: EOE_CBAM_DSCALE_SLICES slices of [n, C] here
    int nz = 1;
    while (nz < EOE_CBAM_DSCALE_SLICES && (size_t)cg->n * (C / 4 / cpb) * nz < 1024 && HW / (nz * 2) >= 64) nz *= 2;
    hipLaunchKernelGGL(cbam_chan_reduce_kernel, dim3(cg->n, C / 4 / cpb, nz), dim3(256), 0, s, (const float*)g, (const float*)sb->dcomp, cg->x,
                       cb->dscale, HW, C, cpb);
    EOE_CHECK_LAUNCH("cbam_chan_reduce");
    hipLaunchKernelGGL(cgate_mlp_bwd_kernel, dim3(cg->n), dim3(256), (C + 2 * cg->Ch) * sizeof(float), s, cg->scale, cg->hidden, cg->w1,
                       cg->w2, cb->dscale, cb->dhidden, cb->dpooled, C, cg->Ch, nz);
    EOE_CHECK_LAUNCH("cbam_mlp_bwd");
    hipLaunchKernelGGL(cgate_mlp_wgrad_kernel, dim3(cdiv(C * cg->Ch, 16)), dim3(256), 0, s, cg->pooled, cg->hidden, cb->dscale,
                       cb->dhidden, cb->dw1, cb->db1, cb->dw2, cb->db2, cg->n, C, cg->Ch);
    EOE_CHECK_LAUNCH("cbam_mlp_wgrad");
    hipLaunchKernelGGL(cbam_bwd_apply_kernel, dim3(grid_for(P * C / 4)), dim3(256), 0, s, (const float*)g, (const float*)sb->dcomp,
                       (const float*)cg->scale, (const float*)cb->dpooled, (const int*)cg->argmax, cb->dx, cg->n, HW, C);
    EOE_CHECK_LAUNCH("cbam_bwd_apply");
    return 0;
}

extern "C" int eoe_add_relu_fwd(const float* a, const float* b, float* out, void* out16, int dtype, int64_t count, void* stream) {
    EOE_CHECK_ARG(a && b && out && count > 0 && count % 4 == 0, "add_relu_fwd: bad args");
    ProfScope ps("add_relu_fwd", 0, 12.0 * count, stream);
    DISPATCH_T(dtype, hipLaunchKernelGGL((add_relu_fwd_kernel<T>), dim3(grid_for((size_t)count / 4)), dim3(256), 0, (hipStream_t)stream,
                                         a, b, out, (T*)out16, (size_t)count / 4));
    EOE_CHECK_LAUNCH("add_relu_fwd");
    return 0;
}

extern "C" int eoe_relu_bwd(const float* dout, const float* out, float* g, int64_t count, void* stream) {
    EOE_CHECK_ARG(dout && out && g && count > 0 && count % 4 == 0, "relu_bwd: bad args");
    ProfScope ps("relu_bwd", 0, 12.0 * count, stream);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for((size_t)count / 4)), dim3(256), 0, (hipStream_t)stream, dout, out, g,
                       (size_t)count / 4);
    EOE_CHECK_LAUNCH("relu_bwd");
    return 0;
}

extern "C" int eoe_avgpool_fwd(const float* x, float* pooled_scratch, int* argmax_scratch, int n, int HW, int C, void* stream) {
    EOE_CHECK_ARG(x && pooled_scratch && argmax_scratch && n > 0 && HW > 0 && C >= 16 && C % 16 == 0, "avgpool_fwd: bad args");
    const int cpb = chan_cpb(C);
    EOE_CHECK_ARG(256 % cpb == 0 && (C / 4) % cpb == 0, "avgpool_fwd: C = %d not supported", C);
    ProfScope ps("avgpool_fwd", 0, 4.0 * n * HW * C, stream);
    hipLaunchKernelGGL(chan_reduce_kernel<0>, dim3(n, C / 4 / cpb), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr,
                       pooled_scratch, argmax_scratch, HW, C, cpb);
    EOE_CHECK_LAUNCH("avgpool_fwd");
    return 0;
}

extern "C" int eoe_avgpool_bwd(const float* dout, float* dx, int n, int HW, int C, void* stream) {
    EOE_CHECK_ARG(dout && dx && n > 0 && HW > 0 && C % 4 == 0, "avgpool_bwd: bad args");
    ProfScope ps("avgpool_bwd", 0, 4.0 * n * HW * C, stream);
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((size_t)n * HW * C / 4)), dim3(256), 0, (hipStream_t)stream, dout, dx, n, HW, C);
    EOE_CHECK_LAUNCH("avgpool_bwd");
    return 0;
}
