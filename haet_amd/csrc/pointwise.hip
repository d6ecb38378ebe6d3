// Point-wise callers either side of the path (the Transolver block around the attention,
// models/HAETransolver_Irregular_Mesh.py:114-134 and the preprocess MLP :186, 338-376): the LayerNorms and
// the narrow first Linear of the preprocess MLP.  Both are pure HBM streams over ~1e6 rows; the stock ops
// run them at ~1/4 of the copy rate (LayerNorm) or in three passes (Linear, bias, GELU).
#include "common.h"

namespace haet {

namespace {

typedef float f4v __attribute__((ext_vector_type(4)));

// nn.GELU(approximate='none') = 0.5 v (1 + erf(v / sqrt 2)).  The narrow Linear has no matrix-core work to hide the
// library erff (~35 instructions with branches) behind, so erf is Abramowitz & Stegun 7.1.26 here: 1 - P5(t) e^{-x^2},
// t = 1 / (1 + p |x|), |error| <= 1.5e-7 -- below fp32 rounding of the GELU value for |v| < 1 and a few ulp above.
__device__ __forceinline__ float gelu_erf(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __expf(-x * x);  // erf(|v| / sqrt 2)
    return 0.5f * v * (1.0f + copysignf(e, v));
}

}  // namespace

// One wave per row: D / 4 float4 chunks over the 64 lanes (D <= 1024), two-pass statistics in registers
// (mean, then the centred sum of squares: torch's numerics), biased variance, eps inside the root.
// RES: the row is x + res (the block's residual add, HAETransolver_Irregular_Mesh.py:125-128), written to `sum`
// on the way -- one pass instead of an add kernel and a LayerNorm kernel.
template <int CH, bool RES = false>  // chunks per lane
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, int rows, int D,
                                                        float* __restrict__ out, const float* __restrict__ res = nullptr,
                                                        float* __restrict__ sum = nullptr) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int nch = D >> 2;
    f4v g[CH], b[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int ch = lane + 64 * c;
        const bool ok = ch < nch;
        g[c] = ok ? *reinterpret_cast<const f4v*>(gamma + 4 * ch) : f4v{0.f, 0.f, 0.f, 0.f};
        b[c] = ok ? *reinterpret_cast<const f4v*>(beta + 4 * ch) : f4v{0.f, 0.f, 0.f, 0.f};
    }
    const float inv_d = 1.0f / (float)D;
    for (int r = wave; r < rows; r += nwaves) {
        const float* xr = x + (size_t)r * D;
        f4v v[CH];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int ch = lane + 64 * c;
            v[c] = ch < nch ? *reinterpret_cast<const f4v*>(xr + 4 * ch) : f4v{0.f, 0.f, 0.f, 0.f};
            if (RES && ch < nch) {
                v[c] += *reinterpret_cast<const f4v*>(res + (size_t)r * D + 4 * ch);
                *reinterpret_cast<f4v*>(sum + (size_t)r * D + 4 * ch) = v[c];
            }
            s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
        }
        const float mean = wave_sum(s) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[c][j] - mean;
                    q = fmaf(d, d, q);
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
        float* orow = out + (size_t)r * D;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                f4v o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = fmaf((v[c][j] - mean) * rstd, g[c][j], b[c][j]);
                *reinterpret_cast<f4v*>(orow + 4 * ch) = o;
            }
        }
    }
}

// Backward of the same LayerNorm (training through the block, exp_airfoil.py:249-255): per row, with
// xhat = (x - mean) rstd and gg = gy gamma,   dx = rstd (gg - mean(gg) - xhat mean(gg xhat));
// dgamma = sum_rows gy xhat, dbeta = sum_rows gy: a lane owns the same 4 * CH columns in every row it visits,
// so the column sums stay in registers and leave as one partial record per workgroup ([dgamma D | dbeta D]).
template <int CH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ gy, float eps, int rows, int D,
                                                            float* __restrict__ dx, float* __restrict__ parts) {
    __shared__ float red[4][2 * 1024];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
    const int nch = D >> 2;
    const f4v zero4 = {0.f, 0.f, 0.f, 0.f};
    f4v g[CH], dg[CH], db[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int ch = lane + 64 * c;
        g[c] = ch < nch ? *reinterpret_cast<const f4v*>(gamma + 4 * ch) : zero4;
        dg[c] = zero4;
        db[c] = zero4;
    }
    const float inv_d = 1.0f / (float)D;
    for (int r = wave; r < rows; r += nwaves) {
        const float* xr = x + (size_t)r * D;
        const float* gr = gy + (size_t)r * D;
        f4v v[CH], go[CH];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int ch = lane + 64 * c;
            v[c] = ch < nch ? *reinterpret_cast<const f4v*>(xr + 4 * ch) : zero4;
            go[c] = ch < nch ? *reinterpret_cast<const f4v*>(gr + 4 * ch) : zero4;
            s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
        }
        const float mean = wave_sum(s) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (lane + 64 * c < nch) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = v[c][j] - mean;
                    q = fmaf(d, d, q);
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (lane + 64 * c < nch) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (v[c][j] - mean) * rstd, gg = go[c][j] * g[c][j];
                    v[c][j] = xh;  // keep xhat
                    s1 += gg;
                    s2 = fmaf(gg, xh, s2);
                    dg[c][j] = fmaf(go[c][j], xh, dg[c][j]);
                    db[c][j] += go[c][j];
                }
            }
        }
        s1 = wave_sum(s1) * inv_d;
        s2 = wave_sum(s2) * inv_d;
        float* orow = dx + (size_t)r * D;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nch) {
                f4v o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = rstd * (go[c][j] * g[c][j] - s1 - v[c][j] * s2);
                *reinterpret_cast<f4v*>(orow + 4 * ch) = o;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nch) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                red[wv][4 * ch + j] = dg[c][j];
                red[wv][D + 4 * ch + j] = db[c][j];
            }
        }
    }
    __syncthreads();
    float* rec = parts + (size_t)blockIdx.x * 2 * D;
    for (int i = threadIdx.x; i < 2 * D; i += 256) rec[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// out[n][o] = act(sum_k x[n][k] W[o][k] + b[o]) for a narrow input (K <= 16: coordinates + a few fields):
// a thread owns 4 consecutive outputs of one row; W^T sits in LDS as [k][O]; the row's K inputs are
// wave-uniform scalar loads when O/4 >= 64.  One pass, the output written once.
template <int ACT>
__global__ __launch_bounds__(256) void linear_narrow_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                            const float* __restrict__ bias, int rows, int K, int O,
                                                            float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [K][O] W^T | [O] bias
    for (int i = threadIdx.x; i < K * O; i += 256) {
        const int k = i / O, o = i % O;
        lds[i] = W[o * K + k];
    }
    for (int o = threadIdx.x; o < O; o += 256) lds[K * O + o] = bias ? bias[o] : 0.f;
    __syncthreads();
    const int o4 = O >> 2;
    const size_t total = (size_t)rows * o4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / o4;
        const int c = (int)(i % o4) * 4;
        f4v acc = *reinterpret_cast<const f4v*>(lds + K * O + c);
        const float* xr = x + r * K;
        for (int k = 0; k < K; ++k) {
            const float xv = xr[k];
            const f4v w = *reinterpret_cast<const f4v*>(lds + k * O + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(xv, w[j], acc[j]);
        }
        if (ACT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = gelu_erf(acc[j]);
        }
        *reinterpret_cast<f4v*>(out + r * O + c) = acc;
    }
}

// Last layer of the model (HAETransolver_Irregular_Mesh.py:129-132): out = mlp2(ln_3(x + res)) with a narrow output
// (out_dim <= 16).  One wave per row like the LayerNorm kernel; neither the sum nor the normalised row is written --
// nothing after the last block reads them -- only the OD outputs: dots of the lane's 4 normalised channels with its
// 4 columns of W, reduced over the wave.  D = 256 (64 lanes x 4 channels).
template <int OD, bool RES>
__global__ __launch_bounds__(256) void ln_linear_small_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float eps, const float* __restrict__ W,
                                                              const float* __restrict__ bias, int rows, int od,
                                                              float* __restrict__ out) {
    constexpr int D = 256;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const f4v g = *reinterpret_cast<const f4v*>(gamma + 4 * lane), b = *reinterpret_cast<const f4v*>(beta + 4 * lane);
    f4v w[OD];
#pragma unroll
    for (int j = 0; j < OD; ++j) w[j] = j < od ? *reinterpret_cast<const f4v*>(W + (size_t)j * D + 4 * lane) : f4v{0.f, 0.f, 0.f, 0.f};
    const float bl = (bias && lane < od) ? bias[lane] : 0.f;
    for (int r = wave; r < rows; r += nwaves) {
        f4v v = *reinterpret_cast<const f4v*>(x + (size_t)r * D + 4 * lane);
        if (RES) v += *reinterpret_cast<const f4v*>(res + (size_t)r * D + 4 * lane);
        const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] -= mean;
            q = fmaf(v[j], v[j], q);
        }
        const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + eps);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j] * rstd, g[j], b[j]);
        float mine = 0.f;  // lane j < od ends up with output j
#pragma unroll
        for (int j = 0; j < OD; ++j) {
            float d = v[0] * w[j][0];
            d = fmaf(v[1], w[j][1], d);
            d = fmaf(v[2], w[j][2], d);
            d = fmaf(v[3], w[j][3], d);
            d = wave_sum(d);
            mine = lane == j ? d : mine;
        }
        if (lane < od) out[(size_t)r * od + lane] = mine + bl;
    }
}

// The same Linear when O / 4 is a multiple of 64 (every wave's 64 lanes = 256 consecutive outputs of ONE row): the
// lane keeps its 4 weight columns in registers for the whole kernel, the row's K inputs are wave-uniform (scalar
// loads), and the kernel is a pure output stream -- the generic kernel above spends its time on 7 wave-wide loads of
// one address and 8 LDS reads per 4 outputs (0.99 ms for the 7 -> 512 layer at 1M rows, 2 GB written).
template <int ACT>
__global__ __launch_bounds__(256) void linear_narrow_rows_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                 const float* __restrict__ bias, int rows, int K, int O,
                                                                 float* __restrict__ out) {
    const int wpr = O >> 8;                                   // waves per row
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int part = gw % wpr, lane = threadIdx.x & 63;
    const int c = (part * 64 + lane) * 4;                     // this lane's 4 output columns, fixed for the kernel
    float w[16][4];
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) w[k][j] = k < K ? W[(size_t)(c + j) * K + k] : 0.f;
    f4v b = {0.f, 0.f, 0.f, 0.f};
    if (bias) b = *reinterpret_cast<const f4v*>(bias + c);
    // waves gw, gw + nw, ... with nw a multiple of wpr keep their `part`; row = wave index / wpr
    for (int wi = gw; wi < rows * wpr; wi += nw) {
        const int r = __builtin_amdgcn_readfirstlane(wi / wpr);
        const float* xr = x + (size_t)r * K;
        f4v acc = b;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k < K) {
                const float xv = xr[k];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(xv, w[k][j], acc[j]);
            }
        }
        if (ACT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = gelu_erf(acc[j]);
        }
        *reinterpret_cast<f4v*>(out + (size_t)r * O + c) = acc;
    }
}

}  // namespace haet

using namespace haet;

extern "C" int haet_layernorm_supported(int D) { return (D >= 4 && D % 4 == 0 && D <= 1024) ? 1 : 0; }

extern "C" int haet_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, int rows, int D,
                                  float* out, haet_stream_t stream) {
    if (!x || !gamma || !beta || !out || rows < 1) return fail(HAET_E_ARG, "layernorm_fwd: bad arguments");
    if (!haet_layernorm_supported(D)) return fail(HAET_E_ARG, "layernorm_fwd: D=%d unsupported (multiple of 4, <= 1024)", D);
    const int blocks = rows < 4 * 2048 ? (rows + 3) / 4 : 2048;  // 8 workgroups of 4 waves per CU
    const int ch = (D / 4 + 63) / 64;
    hipStream_t st = as_stream(stream);
    switch (ch) {
        case 1: hipLaunchKernelGGL((layernorm_kernel<1, false>), dim3(blocks), dim3(256), 0, st, x, gamma, beta, eps, rows, D, out, (const float*)nullptr, (float*)nullptr); break;
        case 2: hipLaunchKernelGGL((layernorm_kernel<2, false>), dim3(blocks), dim3(256), 0, st, x, gamma, beta, eps, rows, D, out, (const float*)nullptr, (float*)nullptr); break;
        case 3: hipLaunchKernelGGL((layernorm_kernel<3, false>), dim3(blocks), dim3(256), 0, st, x, gamma, beta, eps, rows, D, out, (const float*)nullptr, (float*)nullptr); break;
        default: hipLaunchKernelGGL((layernorm_kernel<4, false>), dim3(blocks), dim3(256), 0, st, x, gamma, beta, eps, rows, D, out, (const float*)nullptr, (float*)nullptr); break;
    }
    return check_launch("layernorm_kernel");
}

extern "C" int haet_add_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta,
                                      float eps, int rows, int D, float* sum, float* out, haet_stream_t stream) {
    if (!x || !res || !gamma || !beta || !sum || !out || rows < 1) return fail(HAET_E_ARG, "add_layernorm_fwd: bad arguments");
    if (!haet_layernorm_supported(D)) return fail(HAET_E_ARG, "add_layernorm_fwd: D=%d unsupported (multiple of 4, <= 1024)", D);
    const int blocks = rows < 4 * 2048 ? (rows + 3) / 4 : 2048;
    const int ch = (D / 4 + 63) / 64;
    hipStream_t st = as_stream(stream);
    switch (ch) {
        case 1: hipLaunchKernelGGL((layernorm_kernel<1, true>), dim3(blocks), dim3(256), 0, st, x, gamma, beta, eps, rows, D, out, res, sum); break;
        case 2: hipLaunchKernelGGL((layernorm_kernel<2, true>), dim3(blocks), dim3(256), 0, st, x, gamma, beta, eps, rows, D, out, res, sum); break;
        case 3: hipLaunchKernelGGL((layernorm_kernel<3, true>), dim3(blocks), dim3(256), 0, st, x, gamma, beta, eps, rows, D, out, res, sum); break;
        default: hipLaunchKernelGGL((layernorm_kernel<4, true>), dim3(blocks), dim3(256), 0, st, x, gamma, beta, eps, rows, D, out, res, sum); break;
    }
    return check_launch("layernorm_kernel<res>");
}

extern "C" int haet_ln_linear_small_supported(int D, int OD) { return (D == 256 && OD >= 1 && OD <= 16) ? 1 : 0; }

extern "C" int haet_ln_linear_small_fwd(const float* x, const float* res /*or NULL*/, const float* gamma,
                                        const float* beta, float eps, const float* W /*[OD, D]*/, const float* bias,
                                        int rows, int D, int OD, float* out, haet_stream_t stream) {
    if (!x || !gamma || !beta || !W || !out || rows < 1) return fail(HAET_E_ARG, "ln_linear_small_fwd: bad arguments");
    if (!haet_ln_linear_small_supported(D, OD)) return fail(HAET_E_ARG, "ln_linear_small_fwd: D=%d (256) / OD=%d (<= 16) unsupported", D, OD);
    const int blocks = rows < 4 * 2048 ? (rows + 3) / 4 : 2048;
    hipStream_t st = as_stream(stream);
#define HAET_LNL(ODT)                                                                                                   \
    do {                                                                                                                \
        if (res) hipLaunchKernelGGL((ln_linear_small_kernel<ODT, true>), dim3(blocks), dim3(256), 0, st, x, res, gamma, beta, eps, W, bias, rows, OD, out); \
        else hipLaunchKernelGGL((ln_linear_small_kernel<ODT, false>), dim3(blocks), dim3(256), 0, st, x, res, gamma, beta, eps, W, bias, rows, OD, out);    \
    } while (0)
    if (OD <= 4) HAET_LNL(4);
    else if (OD <= 8) HAET_LNL(8);
    else HAET_LNL(16);
#undef HAET_LNL
    return check_launch("ln_linear_small_kernel");
}

extern "C" int haet_layernorm_bwd_num_parts(int rows) { return rows < 4 * 1024 ? (rows + 3) / 4 : 1024; }

extern "C" int haet_layernorm_bwd(const float* x, const float* gamma, const float* gy, float eps, int rows, int D,
                                  float* dx, float* parts, haet_stream_t stream) {
    if (!x || !gamma || !gy || !dx || !parts || rows < 1) return fail(HAET_E_ARG, "layernorm_bwd: bad arguments");
    if (!haet_layernorm_supported(D)) return fail(HAET_E_ARG, "layernorm_bwd: D=%d unsupported (multiple of 4, <= 1024)", D);
    const int blocks = haet_layernorm_bwd_num_parts(rows);
    const int ch = (D / 4 + 63) / 64;
    hipStream_t st = as_stream(stream);
    switch (ch) {
        case 1: hipLaunchKernelGGL(layernorm_bwd_kernel<1>, dim3(blocks), dim3(256), 0, st, x, gamma, gy, eps, rows, D, dx, parts); break;
        case 2: hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(blocks), dim3(256), 0, st, x, gamma, gy, eps, rows, D, dx, parts); break;
        case 3: hipLaunchKernelGGL(layernorm_bwd_kernel<3>, dim3(blocks), dim3(256), 0, st, x, gamma, gy, eps, rows, D, dx, parts); break;
        default: hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(blocks), dim3(256), 0, st, x, gamma, gy, eps, rows, D, dx, parts); break;
    }
    return check_launch("layernorm_bwd_kernel");
}

extern "C" int haet_linear_narrow_supported(int K, int O) { return (K >= 1 && K <= 16 && O >= 4 && O % 4 == 0 && O <= 2048) ? 1 : 0; }

extern "C" int haet_linear_narrow_fwd(const float* x, const float* W, const float* bias, int rows, int K, int O,
                                      int act, float* out, haet_stream_t stream) {
    if (!x || !W || !out || rows < 1) return fail(HAET_E_ARG, "linear_narrow_fwd: bad arguments");
    if (!haet_linear_narrow_supported(K, O) || (act != 0 && act != 1))
        return fail(HAET_E_ARG, "linear_narrow_fwd: K=%d (<= 16) / O=%d (multiple of 4, <= 2048) / act=%d unsupported", K, O, act);
    const size_t lds = ((size_t)K * O + O) * sizeof(float);
    const size_t total = (size_t)rows * (O / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(HAET_E_LAUNCH, "linear_narrow: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, as_stream(stream), x, W, bias, rows, K, O, out);
        return HAET_OK;
    };
    if (O % 256 == 0) {  // one row per wave group, weights in registers
        const int wpr = O / 256;
        int nblk = 2048 * wpr;                                  // waves per launch: a multiple of wpr
        const long long need = ((long long)rows * wpr + 3) / 4;
        if (need < nblk) nblk = (int)(((need + wpr - 1) / wpr) * wpr);
        if (act)
            hipLaunchKernelGGL(linear_narrow_rows_kernel<1>, dim3(nblk), dim3(256), 0, as_stream(stream), x, W, bias, rows, K, O, out);
        else
            hipLaunchKernelGGL(linear_narrow_rows_kernel<0>, dim3(nblk), dim3(256), 0, as_stream(stream), x, W, bias, rows, K, O, out);
        return check_launch("linear_narrow_rows_kernel");
    }
    int rc = act ? go(linear_narrow_kernel<1>) : go(linear_narrow_kernel<0>);
    if (rc != HAET_OK) return rc;
    return check_launch("linear_narrow_kernel");
}
