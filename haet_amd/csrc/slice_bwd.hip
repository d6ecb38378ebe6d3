// Backward of the slice / de-slice passes: the N-scaled part of
// loss.backward() through Physics_Attention_* (exp_airfoil.py:249-255).  First, correctness-first
// version on the fp32 vector ALUs, same work split as slice_valu.hip (lane = slice column g, x rows
// broadcast with v_readlane); the slice weights are recomputed from x_lin and the column statistics,
// never stored.
//
// Forward (models/PhysicsAttention/IrregularMesh.py:168-188, 228), per cloud b and head h:
//   t = x_n.wt + bt,  tau = base + clamp(t, -0.4, 0.4),  z[n,g] = (x_n.Ws[g] + bs[g] - c0) / tau
//   w[n,g] = e^{z - m_g} / l_g   (softmax over the POINT axis n)
//   S[g] = k sum_n w[n,g] x_n,  k = 1/(1 + 1e-5)     ->  S2 = Erwin(S)     ->  out_n = sum_g w[n,g] S2[g]
// Backward, given go = dL/d out [B,N,H*C]:
//   pass 1 (deslice_bwd_tokens):  dS2[g] = sum_n w[n,g] go_n                 -> Erwin backward gives dS
//   pass 2 (slice_bwd_points):    a[n,g] = go_n.S2[g] + k x_n.dS[g]
//        A[g]   = sum_n w[n,g] a[n,g] = dS2[g].S2[g] + S[g].dS[g]            (host, B*H*M floats)
//        dz     = w (a - A);   dlogit = dz / tau;   dtau = -sum_g dz z / tau;   dt = dtau [|t| <= 0.4]
//        dx_n   = sum_g dlogit Ws[g] + k sum_g w dS[g] + dt wt
//        dWs[g] += dlogit x_n;  dbs[g] += dlogit;  dwt += dt x_n;  dbt += dt  (per-workgroup partials)
#include <cstdlib>

#include "common.h"

namespace haet {

namespace {

constexpr float kNormK = 1.0f / (1.0f + 1e-5f);  // sum_n softmax == 1 (IrregularMesh.py:177,186-188)

// Broadcast one point's row out of the wave-wide load and evaluate t, 1/tau and this lane's logits.
template <int C, int NG>
__device__ __forceinline__ void point_eval(float xv, int s, const float (&w)[NG][C], const float (&bias)[NG],
                                           const float* __restrict__ wt, float bt, float base_temp,
                                           float (&xs)[C], float (&z)[NG], float& t, float& rtau) {
    t = bt;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        xs[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), s * C + c));
        t = fmaf(xs[c], wt[c], t);
    }
    rtau = 1.0f / ada_temp(t, base_temp);
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        float d = bias[j];
#pragma unroll
        for (int c = 0; c < C; ++c) d = fmaf(xs[c], w[j][c], d);
        z[j] = d * rtau;
    }
}

}  // namespace

// ----------------------------------------------------------------------------- pass 1
template <int C, int NG>
__global__ __launch_bounds__(256) void deslice_bwd_tokens_kernel(
    const float* __restrict__ x, const float* __restrict__ Ws, const float* __restrict__ bs,
    const float* __restrict__ wt, float bt, float base_temp, float c0,
    const float* __restrict__ col_max, const float* __restrict__ col_sum,
    const float* __restrict__ go, int N, int H, int M, int chunk, float* __restrict__ parts) {
    constexpr int PPL = 64 / C, W = 64 * NG;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.y, b = bh / H, h = bh % H, BH = gridDim.y;
    const int n0 = blockIdx.x * chunk, n1 = min(N, n0 + chunk);

    float w[NG][C], bias[NG], cm[NG], rcs[NG], acc[NG][C];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int g = lane + 64 * j;
        const bool valid = g < M;
        bias[j] = valid ? bs[g] - c0 : 0.f;
        cm[j] = valid ? col_max[bh * M + g] : 0.f;
        rcs[j] = valid ? 1.0f / col_sum[bh * M + g] : 0.f;  // 0 => weight 0 in the padding columns
#pragma unroll
        for (int c = 0; c < C; ++c) {
            w[j][c] = valid ? Ws[g * C + c] : 0.f;
            acc[j][c] = 0.f;
        }
    }
    const int sub = lane / C, cl = lane % C;
    const size_t row = (size_t)H * C;
    const size_t off = ((size_t)b * N * H + h) * C + cl;
    const float* xb = x + off;
    const float* gb = go + off;

    int base = n0 + wave * PPL;
    float xv = (base + sub < n1) ? xb[(size_t)(base + sub) * row] : 0.f;
    float gv = (base + sub < n1) ? gb[(size_t)(base + sub) * row] : 0.f;
    while (base < n1) {
        const int nb = base + 4 * PPL;
        const float xn = (nb + sub < n1) ? xb[(size_t)(nb + sub) * row] : 0.f;  // next iteration's rows
        const float gn = (nb + sub < n1) ? gb[(size_t)(nb + sub) * row] : 0.f;
#pragma unroll
        for (int s = 0; s < PPL; ++s) {
            if (base + s < n1) {  // wave-uniform
                float xs[C], z[NG], t, rtau;
                point_eval<C, NG>(xv, s, w, bias, wt, bt, base_temp, xs, z, t, rtau);
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    const float p = __expf(z[j] - cm[j]) * rcs[j];
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float g_c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gv), s * C + c));
                        acc[j][c] = fmaf(p, g_c, acc[j][c]);
                    }
                }
            }
        }
        xv = xn;
        gv = gn;
        base = nb;
    }

    // sum the 4 waves through LDS ([c][W] floats), one at a time
    for (int src = 1; src < 4; ++src) {
        if (wave == src) {
#pragma unroll
            for (int j = 0; j < NG; ++j)
#pragma unroll
                for (int c = 0; c < C; ++c) lds[c * W + lane + 64 * j] = acc[j][c];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < NG; ++j)
#pragma unroll
                for (int c = 0; c < C; ++c) acc[j][c] += lds[c * W + lane + 64 * j];
        }
        __syncthreads();
    }
    if (wave == 0) {
        float* rec = parts + ((size_t)blockIdx.x * BH + bh) * (size_t)M * C;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int g = lane + 64 * j;
            if (g < M) {
#pragma unroll
                for (int c = 0; c < C; ++c) rec[g * C + c] = acc[j][c];
            }
        }
    }
}

// ----------------------------------------------------------------------------- pass 2
// LDS (floats): S2^T [C][W] | k dS^T [C][W] | Ws [M][C] | k dS [M][C] | per wave [PPL][2][W] coefficient rows
template <int C, int NG>
__global__ __launch_bounds__(256) void slice_bwd_points_kernel(
    const float* __restrict__ x, const float* __restrict__ Ws, const float* __restrict__ bs,
    const float* __restrict__ wt, float bt, float base_temp, float c0,
    const float* __restrict__ col_max, const float* __restrict__ col_sum,
    const float* __restrict__ go, const float* __restrict__ S2, const float* __restrict__ dS,
    const float* __restrict__ A, int N, int H, int M, int chunk,
    float* __restrict__ dx, float* __restrict__ pparts) {
    constexpr int PPL = 64 / C, W = 64 * NG;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s2T = lds;
    float* dsT = s2T + C * W;
    float* WsL = dsT + C * W;
    float* dsL = WsL + M * C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* wb = dsL + M * C + wave * (PPL * 2 * W);
    const int bh = blockIdx.y, b = bh / H, h = bh % H, BH = gridDim.y;
    const int n0 = blockIdx.x * chunk, n1 = min(N, n0 + chunk);

    for (int i = threadIdx.x; i < C * W; i += 256) {
        const int c = i / W, g = i % W;
        s2T[i] = g < M ? S2[((size_t)bh * M + g) * C + c] : 0.f;
        dsT[i] = g < M ? kNormK * dS[((size_t)bh * M + g) * C + c] : 0.f;
    }
    for (int i = threadIdx.x; i < M * C; i += 256) {
        WsL[i] = Ws[i];
        dsL[i] = kNormK * dS[(size_t)bh * M * C + i];
    }
    float w[NG][C], bias[NG], cm[NG], rcs[NG], Ac[NG], dW[NG][C], db[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int g = lane + 64 * j;
        const bool valid = g < M;
        bias[j] = valid ? bs[g] - c0 : 0.f;
        cm[j] = valid ? col_max[bh * M + g] : 0.f;
        rcs[j] = valid ? 1.0f / col_sum[bh * M + g] : 0.f;
        Ac[j] = valid ? A[bh * M + g] : 0.f;
        db[j] = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            w[j][c] = valid ? Ws[g * C + c] : 0.f;
            dW[j][c] = 0.f;
        }
    }
    __syncthreads();
    const int sub = lane / C, cl = lane % C;
    const size_t row = (size_t)H * C;
    const size_t off = ((size_t)b * N * H + h) * C + cl;
    const float* xb = x + off;
    const float* gb = go + off;
    float* ob = dx + off;
    const float wt_cl = wt[cl];
    float dwt = 0.f, dbt = 0.f;

    int base = n0 + wave * PPL;
    float xv = (base + sub < n1) ? xb[(size_t)(base + sub) * row] : 0.f;
    float gv = (base + sub < n1) ? gb[(size_t)(base + sub) * row] : 0.f;
    while (base < n1) {
        const int nb = base + 4 * PPL;
        const float xn = (nb + sub < n1) ? xb[(size_t)(nb + sub) * row] : 0.f;
        const float gn = (nb + sub < n1) ? gb[(size_t)(nb + sub) * row] : 0.f;
        float dt_mine = 0.f;
#pragma unroll
        for (int s = 0; s < PPL; ++s) {
            if (base + s < n1) {  // wave-uniform
                float xs[C], z[NG], t, rtau, a[NG];
                point_eval<C, NG>(xv, s, w, bias, wt, bt, base_temp, xs, z, t, rtau);
#pragma unroll
                for (int j = 0; j < NG; ++j) a[j] = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float g_c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gv), s * C + c));
#pragma unroll
                    for (int j = 0; j < NG; ++j) {
                        a[j] = fmaf(g_c, s2T[c * W + lane + 64 * j], a[j]);
                        a[j] = fmaf(xs[c], dsT[c * W + lane + 64 * j], a[j]);
                    }
                }
                float part = 0.f;
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    const float p = __expf(z[j] - cm[j]) * rcs[j];
                    const float dz = p * (a[j] - Ac[j]);
                    const float dl = dz * rtau;
                    part = fmaf(dz, z[j], part);
                    wb[(s * 2 + 0) * W + lane + 64 * j] = dl;
                    wb[(s * 2 + 1) * W + lane + 64 * j] = p;
                    db[j] += dl;
#pragma unroll
                    for (int c = 0; c < C; ++c) dW[j][c] = fmaf(dl, xs[c], dW[j][c]);
                }
                const float dtau = -wave_sum(part) * rtau;
                const float dt = (t >= -0.4f && t <= 0.4f) ? dtau : 0.f;  // clamp passes the gradient inside its range
                dbt += dt;
                if (sub == s) {
                    dt_mine = dt;
                    dwt = fmaf(dt, xv, dwt);
                }
            }
        }
        // wave-local LDS hand-off (this wave wrote wb): ordering only, no barrier
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        if (base + sub < n1) {
            float o = dt_mine * wt_cl;
            const float* r0 = wb + (sub * 2 + 0) * W;
            const float* r1 = wb + (sub * 2 + 1) * W;
            for (int g = 0; g < M; ++g) {
                o = fmaf(r0[g], WsL[g * C + cl], o);
                o = fmaf(r1[g], dsL[g * C + cl], o);
            }
            ob[(size_t)(base + sub) * row] = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        xv = xn;
        gv = gn;
        base = nb;
    }

    // parameter-gradient partials of this workgroup: sum the 4 waves through LDS (the S2^T / dS^T region)
    for (int o = C; o < 64; o <<= 1) dwt += __shfl_xor(dwt, o);  // lanes of the same channel
    float* red = lds;  // [C][W] dW | [W] db | [C] dwt | [1] dbt   (<= 2 C W floats)
    __syncthreads();
    for (int src = 1; src < 4; ++src) {
        if (wave == src) {
#pragma unroll
            for (int j = 0; j < NG; ++j) {
#pragma unroll
                for (int c = 0; c < C; ++c) red[c * W + lane + 64 * j] = dW[j][c];
                red[C * W + lane + 64 * j] = db[j];
            }
            if (lane < C) red[(C + 1) * W + lane] = dwt;
            if (lane == 0) red[(C + 1) * W + C] = dbt;
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < NG; ++j) {
#pragma unroll
                for (int c = 0; c < C; ++c) dW[j][c] += red[c * W + lane + 64 * j];
                db[j] += red[C * W + lane + 64 * j];
            }
            if (lane < C) dwt += red[(C + 1) * W + lane];
            dbt += red[(C + 1) * W + C];
        }
        __syncthreads();
    }
    if (wave == 0) {
        float* rec = pparts + ((size_t)blockIdx.x * BH + bh) * (size_t)(M * C + M + C + 1);
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int g = lane + 64 * j;
            if (g < M) {
#pragma unroll
                for (int c = 0; c < C; ++c) rec[g * C + c] = dW[j][c];
                rec[M * C + g] = db[j];
            }
        }
        if (lane < C) rec[M * C + M + lane] = dwt;
        if (lane == 0) rec[M * C + M + C] = dbt;
    }
}

// ----------------------------------------------------------------------------- host side
int slice_shape_ok(int B, int N, int H, int C, int M);  // slice_valu.hip
// slice_bwd_mfma.hip: the same passes on v_mfma_f32_32x32x2_f32 (C = 32, M in {32, 64, 128}; C = 64, M in {32, 64}), same records
bool slice_bwd_mfma_ok(int C, int M);
int slice_bwd_mfma(int mode, const float*, const float*, const float*, const float*, float, float, float,
                   const float*, const float*, const float*, const float*, const float*, const float*, int, int, int,
                   int, int, float*, float*, hipStream_t);

// slice_bwd_x3.hip: the same passes on split-bf16 products (C = 32, M in {32, 64}), same records, same chunking
bool slice_bwd_x3_ok(int C, int M);
int slice_bwd_x3(int mode, const float*, const float*, const float*, const float*, float, float, float,
                 const float*, const float*, const float*, const float*, const float*, const float*, int, int, int,
                 int, int, float*, float*, hipStream_t);

static bool use_bwd_mfma(int C, int M) {  // HAET_FORCE_VALU=1 routes every shape through the generic kernels (tests)
    return !env_flag("HAET_FORCE_VALU") && slice_bwd_mfma_ok(C, M);
}
// the split-bf16 passes (C = 32, M in {32, 64}) unless the caller asked for exact-f32 products: HAET_FORCE_F32_MFMA=1 is the
// library-wide "no split-bf16 envelope" switch of include/haet_hip.h (forward AND backward), HAET_BWD_F32=1 the
// backward-only one
static bool use_bwd_x3(int C, int M) {
    return use_bwd_mfma(C, M) && !env_flag("HAET_FORCE_F32_MFMA") && !env_flag("HAET_BWD_F32") && slice_bwd_x3_ok(C, M);
}
// the family the two entry points dispatch (C, M) to -- what haet_slice_bwd_num_parts sizes the caller's buffers for
static int bwd_family(int C, int M) { return use_bwd_x3(C, M) ? kBwdX3 : use_bwd_mfma(C, M) ? kBwdF32 : kBwdValu; }

template <int C, int NG>
static int launch_tokens(dim3 grid, hipStream_t st, const float* x, const float* Ws, const float* bs,
                         const float* wt, float bt, float base_temp, float c0, const float* cm, const float* cs,
                         const float* go, int N, int H, int M, int chunk, float* parts) {
    const size_t lds = (size_t)C * 64 * NG * sizeof(float);
    hipLaunchKernelGGL((deslice_bwd_tokens_kernel<C, NG>), grid, dim3(256), lds, st, x, Ws, bs, wt, bt, base_temp,
                       c0, cm, cs, go, N, H, M, chunk, parts);
    return check_launch("deslice_bwd_tokens_kernel");
}

template <int C, int NG>
static int launch_points(dim3 grid, hipStream_t st, const float* x, const float* Ws, const float* bs,
                         const float* wt, float bt, float base_temp, float c0, const float* cm, const float* cs,
                         const float* go, const float* S2, const float* dS, const float* A, int N, int H, int M,
                         int chunk, float* dx, float* pparts) {
    constexpr int PPL = 64 / C, W = 64 * NG;
    const size_t lds = ((size_t)2 * C * W + 2 * (size_t)M * C + 4 * PPL * 2 * W) * sizeof(float);
    auto kern = slice_bwd_points_kernel<C, NG>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(HAET_E_LAUNCH, "slice_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, Ws, bs, wt, bt, base_temp, c0, cm, cs, go, S2, dS, A, N,
                       H, M, chunk, dx, pparts);
    return check_launch("slice_bwd_points_kernel");
}

#define HAET_BWD_DISPATCH(FN, ...)                                  \
    do {                                                            \
        const int NG_ = (M + 63) / 64;                              \
        switch (C * 4 + NG_) {                                      \
            case 8 * 4 + 1: return FN<8, 1>(__VA_ARGS__);           \
            case 8 * 4 + 2: return FN<8, 2>(__VA_ARGS__);           \
            case 16 * 4 + 1: return FN<16, 1>(__VA_ARGS__);         \
            case 16 * 4 + 2: return FN<16, 2>(__VA_ARGS__);         \
            case 32 * 4 + 1: return FN<32, 1>(__VA_ARGS__);         \
            case 32 * 4 + 2: return FN<32, 2>(__VA_ARGS__);         \
            case 64 * 4 + 1: return FN<64, 1>(__VA_ARGS__);         \
            default: return FN<64, 2>(__VA_ARGS__);                 \
        }                                                           \
    } while (0)

}  // namespace haet

using namespace haet;

extern "C" int haet_slice_bwd_num_parts(int B, int N, int H, int C, int M) {
    if (slice_shape_ok(B, N, H, C, M) != HAET_OK) return HAET_E_ARG;
    const int chunk = slice_bwd_pick_chunk(N, bwd_family(C, M));  // (the same predicates as the two entry points below)
    return (N + chunk - 1) / chunk;
}

extern "C" size_t haet_slice_bwd_param_floats(int C, int M) { return (size_t)M * C + M + C + 1; }

extern "C" int haet_deslice_bwd_tokens(const float* x_lin, const float* Ws, const float* bs, const float* wt,
                                       float bt, float base_temp, float c0, const float* col_max,
                                       const float* col_sum, const float* g_out, int B, int N, int H, int C,
                                       int M, float* parts, haet_stream_t stream) {
    if (!x_lin || !Ws || !bs || !wt || !col_max || !col_sum || !g_out || !parts)
        return fail(HAET_E_ARG, "deslice_bwd_tokens: null pointer");
    int rc = slice_shape_ok(B, N, H, C, M);
    if (rc != HAET_OK) return rc;
    if (use_bwd_x3(C, M))
        return slice_bwd_x3(0, x_lin, Ws, bs, wt, bt, base_temp, c0, col_max, col_sum, g_out, nullptr, nullptr, nullptr,
                            B, N, H, C, M, nullptr, parts, as_stream(stream));
    if (use_bwd_mfma(C, M))
        return slice_bwd_mfma(0, x_lin, Ws, bs, wt, bt, base_temp, c0, col_max, col_sum, g_out, nullptr, nullptr,
                              nullptr, B, N, H, C, M, nullptr, parts, as_stream(stream));
    const int chunk = slice_bwd_pick_chunk(N, kBwdValu);
    dim3 grid((N + chunk - 1) / chunk, B * H);
    HAET_BWD_DISPATCH(launch_tokens, grid, as_stream(stream), x_lin, Ws, bs, wt, bt, base_temp, c0, col_max,
                      col_sum, g_out, N, H, M, chunk, parts);
}

extern "C" int haet_slice_bwd_points(const float* x_lin, const float* Ws, const float* bs, const float* wt,
                                     float bt, float base_temp, float c0, const float* col_max,
                                     const float* col_sum, const float* g_out, const float* S2,
                                     const float* dS, const float* A, int B, int N, int H, int C, int M,
                                     float* dx_lin, float* param_parts, haet_stream_t stream) {
    if (!x_lin || !Ws || !bs || !wt || !col_max || !col_sum || !g_out || !S2 || !dS || !A || !dx_lin || !param_parts)
        return fail(HAET_E_ARG, "slice_bwd_points: null pointer");
    int rc = slice_shape_ok(B, N, H, C, M);
    if (rc != HAET_OK) return rc;
    if (use_bwd_x3(C, M))
        return slice_bwd_x3(1, x_lin, Ws, bs, wt, bt, base_temp, c0, col_max, col_sum, g_out, S2, dS, A, B, N, H, C, M,
                            dx_lin, param_parts, as_stream(stream));
    if (use_bwd_mfma(C, M))
        return slice_bwd_mfma(1, x_lin, Ws, bs, wt, bt, base_temp, c0, col_max, col_sum, g_out, S2, dS, A, B, N, H, C, M,
                              dx_lin, param_parts, as_stream(stream));
    const int chunk = slice_bwd_pick_chunk(N, kBwdValu);
    dim3 grid((N + chunk - 1) / chunk, B * H);
    HAET_BWD_DISPATCH(launch_points, grid, as_stream(stream), x_lin, Ws, bs, wt, bt, base_temp, c0, col_max,
                      col_sum, g_out, S2, dS, A, N, H, M, chunk, dx_lin, param_parts);
}
