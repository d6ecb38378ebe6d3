// Generic (any C in {16,32,64}, any M <= 128) fp32 VALU kernels for the slice and
// de-slice passes.  Correctness-first shape-generic path; the MFMA kernels in
// slice_mfma.hip take over for the shapes they support.
//
// Data flow (reference models/PhysicsAttention/IrregularMesh.py):
//   z[n,g]  = (x[n,h,:] . Ws[g,:] + bs[g] - c0) / tau_n          (:172-174)
//   tau_n   = base_temp + clamp(x[n,h,:] . wt + bt, -0.4, 0.4)   (:168)
//   slice   : online softmax over the POINT axis n, St[g,:] = sum_n e^{z-m_g} x[n,h,:]
//   deslice : w[n,g] = e^{z - m_g} / l_g ; out[n,h,:] = sum_g w[n,g] S2[h,g,:]  (:228)
// One wave owns up to 64*NG slice columns (lane = g); x rows are loaded once per
// wave with coalesced dword loads and broadcast with v_readlane.
#include "common.h"

namespace haet {

template <int C>
struct RowBcast {
    static constexpr int PPL = 64 / C;  // points per wave-wide load
};

// z for this lane's NG columns of one point whose row sits in lanes [s*C, s*C+C) of xv
template <int C, int NG>
__device__ __forceinline__ void point_logits(float xv, int s, const float (&w)[NG][C],
                                             const float (&bias)[NG], const float* __restrict__ wt,
                                             float bt, float base_temp, float (&xs)[C],
                                             float (&z)[NG]) {
    float t = bt;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        xs[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), s * C + c));
        t = fmaf(xs[c], wt[c], t);
    }
    const float tau = ada_temp(t, base_temp);
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        float d = bias[j];
#pragma unroll
        for (int c = 0; c < C; ++c) d = fmaf(xs[c], w[j][c], d);
        z[j] = d / tau;
    }
}

template <int C, int NG>
__global__ __launch_bounds__(256) void slice_valu_kernel(
    const float* __restrict__ x, const float* __restrict__ Ws, const float* __restrict__ bs,
    const float* __restrict__ wt, float bt, float base_temp, float c0, int N, int H, int M,
    int chunk, float* __restrict__ parts) {
    constexpr int PPL = RowBcast<C>::PPL;
    constexpr int U = 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.y, b = bh / H, h = bh % H, BH = gridDim.y;
    const int n0 = blockIdx.x * chunk, n1 = min(N, n0 + chunk);

    float w[NG][C], bias[NG], m[NG], l[NG], acc[NG][C];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int g = lane + 64 * j;
        const bool valid = g < M;
        bias[j] = valid ? bs[g] - c0 : 0.f;
        m[j] = -INFINITY;
        l[j] = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            w[j][c] = valid ? Ws[g * C + c] : 0.f;
            acc[j][c] = 0.f;
        }
    }
    const int sub = lane / C, cl = lane % C;
    const float* xb = x + ((size_t)b * N * H + h) * C + cl;

    for (int base = n0 + wave; base < n1; base += 4 * PPL * U) {
        float xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = base + (u * PPL + sub) * 4;
            xv[u] = (n < n1) ? xb[(size_t)n * H * C] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int s = 0; s < PPL; ++s) {
                const int n = base + (u * PPL + s) * 4;  // wave-uniform
                if (n < n1) {
                    float xs[C], z[NG];
                    point_logits<C, NG>(xv[u], s, w, bias, wt, bt, base_temp, xs, z);
                    bool grow = false;
#pragma unroll
                    for (int j = 0; j < NG; ++j) grow |= z[j] > m[j];
                    if (__any(grow)) {  // rare after the first few points
#pragma unroll
                        for (int j = 0; j < NG; ++j) {
                            const float mn = fmaxf(m[j], z[j]);
                            const float a = __expf(m[j] - mn);  // m=-inf -> 0
                            l[j] *= a;
#pragma unroll
                            for (int c = 0; c < C; ++c) acc[j][c] *= a;
                            m[j] = mn;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < NG; ++j) {
                        const float p = __expf(z[j] - m[j]);
                        l[j] += p;
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[j][c] = fmaf(p, xs[c], acc[j][c]);
                    }
                }
            }
        }
    }

    // merge the 4 waves through LDS, one at a time (LDS: (C+2) x 64*NG floats)
    constexpr int W = 64 * NG;
    for (int src = 1; src < 4; ++src) {
        if (wave == src) {
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                const int col = lane + 64 * j;
                lds[col] = m[j];
                lds[W + col] = l[j];
#pragma unroll
                for (int c = 0; c < C; ++c) lds[(2 + c) * W + col] = acc[j][c];
            }
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                const int col = lane + 64 * j;
                const float m2 = lds[col], l2 = lds[W + col];
                const float mn = fmaxf(m[j], m2);
                // both -inf only if neither wave saw a point
                const float a1 = (m[j] == -INFINITY) ? 0.f : __expf(m[j] - mn);
                const float a2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
                l[j] = l[j] * a1 + l2 * a2;
#pragma unroll
                for (int c = 0; c < C; ++c) acc[j][c] = acc[j][c] * a1 + lds[(2 + c) * W + col] * a2;
                m[j] = mn;
            }
        }
        __syncthreads();
    }
    if (wave == 0) {
        float* rec = parts + ((size_t)blockIdx.x * BH + bh) * (size_t)M * (C + 2);
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int g = lane + 64 * j;
            if (g < M) {
                rec[g] = m[j];
                rec[M + g] = l[j];
#pragma unroll
                for (int c = 0; c < C; ++c) rec[2 * M + g * C + c] = acc[j][c];
            }
        }
    }
}

// ----------------------------------------------------------------------------- de-slice
template <int C, int NG>
__global__ __launch_bounds__(256) void deslice_valu_kernel(
    const float* __restrict__ x, const float* __restrict__ Ws, const float* __restrict__ bs,
    const float* __restrict__ wt, float bt, float base_temp, float c0,
    const float* __restrict__ col_max, const float* __restrict__ col_sum,
    const float* __restrict__ S2, int N, int H, int M, int chunk, float* __restrict__ out) {
    constexpr int PPL = RowBcast<C>::PPL;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // LDS: S2 of this head [M][C] | per-wave weight rows [4][PPL][M]
    float* s2 = lds;
    float* wb = lds + M * C + (threadIdx.x >> 6) * PPL * M;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int n0 = blockIdx.x * chunk, n1 = min(N, n0 + chunk);

    for (int i = threadIdx.x; i < M * C; i += 256) s2[i] = S2[(size_t)bh * M * C + i];
    float w[NG][C], bias[NG], cm[NG], cs[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int g = lane + 64 * j;
        const bool valid = g < M;
        bias[j] = valid ? bs[g] - c0 : 0.f;
        cm[j] = valid ? col_max[bh * M + g] : 0.f;
        cs[j] = valid ? col_sum[bh * M + g] : 1.f;
#pragma unroll
        for (int c = 0; c < C; ++c) w[j][c] = valid ? Ws[g * C + c] : 0.f;
    }
    __syncthreads();
    const int sub = lane / C, cl = lane % C;
    const size_t row = (size_t)H * C;
    const float* xb = x + ((size_t)b * N * H + h) * C + cl;
    float* ob = out + ((size_t)b * N * H + h) * C + cl;

    for (int base = n0 + wave * PPL; base < n1; base += 4 * PPL) {
        const int nl = base + sub;
        const float xv = (nl < n1) ? xb[(size_t)nl * row] : 0.f;
#pragma unroll
        for (int s = 0; s < PPL; ++s) {
            if (base + s < n1) {  // wave-uniform
                float xs[C], z[NG];
                point_logits<C, NG>(xv, s, w, bias, wt, bt, base_temp, xs, z);
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    const int g = lane + 64 * j;
                    if (g < M) wb[s * M + g] = __expf(z[j] - cm[j]) / cs[j];
                }
            }
        }
        // wave-local LDS hand-off (same wave wrote wb): no barrier needed, just ordering
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        if (nl < n1) {
            float o = 0.f;
            const float* wr = wb + sub * M;
            for (int g = 0; g < M; ++g) o = fmaf(wr[g], s2[g * C + cl], o);
            ob[(size_t)nl * row] = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
}

// ----------------------------------------------------------------------------- host side
static int pick_chunk(int N) {
    // ~256 chunks at large N (partials stay ~2% of the streamed bytes), >= 256 points each
    int c = (N + 255) / 256;
    c = ((c + 255) / 256) * 256;
    return c < 256 ? 256 : c;
}

template <int C>
static int launch_slice(int NG, dim3 grid, size_t lds, hipStream_t st, const float* x,
                        const float* Ws, const float* bs, const float* wt, float bt,
                        float base_temp, float c0, int N, int H, int M, int chunk, float* parts) {
    if (NG == 1)
        hipLaunchKernelGGL((slice_valu_kernel<C, 1>), grid, dim3(256), lds, st, x, Ws, bs, wt, bt,
                           base_temp, c0, N, H, M, chunk, parts);
    else
        hipLaunchKernelGGL((slice_valu_kernel<C, 2>), grid, dim3(256), lds, st, x, Ws, bs, wt, bt,
                           base_temp, c0, N, H, M, chunk, parts);
    return check_launch("slice_valu_kernel");
}

template <int C>
static int launch_deslice(int NG, dim3 grid, size_t lds, hipStream_t st, const float* x,
                          const float* Ws, const float* bs, const float* wt, float bt,
                          float base_temp, float c0, const float* cm, const float* cs,
                          const float* S2, int N, int H, int M, int chunk, float* out) {
    if (NG == 1)
        hipLaunchKernelGGL((deslice_valu_kernel<C, 1>), grid, dim3(256), lds, st, x, Ws, bs, wt, bt,
                           base_temp, c0, cm, cs, S2, N, H, M, chunk, out);
    else
        hipLaunchKernelGGL((deslice_valu_kernel<C, 2>), grid, dim3(256), lds, st, x, Ws, bs, wt, bt,
                           base_temp, c0, cm, cs, S2, N, H, M, chunk, out);
    return check_launch("deslice_valu_kernel");
}

int slice_shape_ok(int B, int N, int H, int C, int M) {
    if (B < 1 || N < 1 || H < 1 || M < 1) return fail(HAET_E_ARG, "slice: B,N,H,M must be >= 1");
    if (C != 8 && C != 16 && C != 32 && C != 64)
        return fail(HAET_E_ARG, "slice: dim_head C=%d unsupported (8, 16, 32 or 64)", C);
    if (M > 128) return fail(HAET_E_ARG, "slice: slice_num M=%d unsupported (<= 128)", M);
    if ((long long)B * H > 65535) return fail(HAET_E_ARG, "slice: B*H too large");
    return HAET_OK;
}

int slice_valu_num_parts(int N) { return (N + pick_chunk(N) - 1) / pick_chunk(N); }

int slice_valu_fwd(const float* x, const float* Ws, const float* bs, const float* wt, float bt,
                   float base_temp, float c0, int B, int N, int H, int C, int M, float* parts,
                   hipStream_t st) {
    const int chunk = pick_chunk(N), P = (N + chunk - 1) / chunk, NG = (M + 63) / 64;
    dim3 grid(P, B * H);
    const size_t lds = (size_t)(C + 2) * 64 * NG * sizeof(float);
    switch (C) {
        case 8: return launch_slice<8>(NG, grid, lds, st, x, Ws, bs, wt, bt, base_temp, c0, N, H, M, chunk, parts);
        case 16: return launch_slice<16>(NG, grid, lds, st, x, Ws, bs, wt, bt, base_temp, c0, N, H, M, chunk, parts);
        case 32: return launch_slice<32>(NG, grid, lds, st, x, Ws, bs, wt, bt, base_temp, c0, N, H, M, chunk, parts);
        default: return launch_slice<64>(NG, grid, lds, st, x, Ws, bs, wt, bt, base_temp, c0, N, H, M, chunk, parts);
    }
}

int deslice_valu_fwd(const float* x, const float* Ws, const float* bs, const float* wt, float bt,
                     float base_temp, float c0, const float* cm, const float* cs, const float* S2,
                     int B, int N, int H, int C, int M, float* out, hipStream_t st) {
    // 1024 points per workgroup at large N; at small N enough workgroups to fill the chip (~512 in all)
    int chunk = (N + (512 / (B * H > 512 ? 512 : B * H)) - 1) / (512 / (B * H > 512 ? 512 : B * H));
    chunk = ((chunk + 63) / 64) * 64;
    chunk = chunk < 64 ? 64 : (chunk > 1024 ? 1024 : chunk);
    const int NG = (M + 63) / 64;
    dim3 grid((N + chunk - 1) / chunk, B * H);
    const size_t lds = ((size_t)M * C + 4 * (64 / C) * M) * sizeof(float);
    switch (C) {
        case 8: return launch_deslice<8>(NG, grid, lds, st, x, Ws, bs, wt, bt, base_temp, c0, cm, cs, S2, N, H, M, chunk, out);
        case 16: return launch_deslice<16>(NG, grid, lds, st, x, Ws, bs, wt, bt, base_temp, c0, cm, cs, S2, N, H, M, chunk, out);
        case 32: return launch_deslice<32>(NG, grid, lds, st, x, Ws, bs, wt, bt, base_temp, c0, cm, cs, S2, N, H, M, chunk, out);
        default: return launch_deslice<64>(NG, grid, lds, st, x, Ws, bs, wt, bt, base_temp, c0, cm, cs, S2, N, H, M, chunk, out);
    }
}

}  // namespace haet
