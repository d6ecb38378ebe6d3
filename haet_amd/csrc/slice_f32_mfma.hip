// Slice / de-slice forward passes on the f32 matrix cores (v_mfma_f32_32x32x2_f32) for the shapes the
// split-bf16 kernels of slice_mfma.hip do not cover: dim_head C = 64 (BASELINE config C2: H = 4, C = 64,
// M = 32), C = 32 with M in {32, 64, 128} as well.  Same data flow as the backward kernels of
// slice_bwd_mfma.hip: one wave per 32-point tile of one (cloud, head), the POINT in the lane, the slice column
// in the accumulator's register index (row = 8 (v / 4) + 4 (lane / 32) + v % 4 per 32-row block), x rows
// straight from global memory (C / 2 channels per lane), exact f32 products.
//
//   slice   (IrregularMesh.py:168-188): Z^T = Ws X^T (+ b), tau per lane, p = e^{z - m_ref[g]} with a deferred
//           reference maximum (rescale only when a tile exceeds it), St[g][c] += P[g][n] X[n][c] through a
//           wave-private LDS tile (the one contraction over the points), l[g] += p per lane; the workgroup's
//           waves merge by the log-sum-exp rule into ONE record of haet_slice_fwd's format.
//   deslice (IrregularMesh.py:228): same logits, w = e^{z - m_g} / l_g, Out^T [c][n] = S2^T [c][g] W^T [g][n]
//           with the accumulators as B operands; nothing goes through LDS but the constant A images.
#include "common.h"

namespace haet {

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr float kRescaleThr = 16.0f;  // natural-log units: p <= e^16 before a rescale, far inside f32

__device__ __forceinline__ f16v mfma(float a, float b, f16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float half_max(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

int pick_chunk(int N, int BH) {  // ~512 workgroups at small N (each builds its operand images and merges its waves once)
    int P = 512 / (BH > 0 ? BH : 1);
    if (P < 1) P = 1;
    int c = (N + P - 1) / P;
    c = ((c + 255) / 256) * 256;  // whole 32-point tiles for all 8 waves
    if (c > 4096) c = 4096;       // and at most 16 tiles per wave: short f32 accumulation chains, the merge adds the rest
    return c < 256 ? 256 : c;
}

// A image of the logits product: [blk][q][lane][4] = Ws[32 blk + l%32][4q + j + (C/2) (l/32)]
template <int CB, int MB>
__device__ __forceinline__ void build_ws_image(float* WsA, const float* __restrict__ Ws, int nthr) {
    constexpr int C = 32 * CB, QN = 4 * CB;
    for (int i = threadIdx.x; i < MB * QN * 256; i += nthr) {
        const int j = i & 3, l = (i >> 2) & 63, q = (i >> 8) % QN, blk = (i >> 8) / QN;
        WsA[i] = Ws[(32 * blk + (l & 31)) * C + 4 * q + j + (C / 2) * (l >> 5)];
    }
}

// this lane's C/2 channels of point n (zero past the end), and the logits Z^T of the tile + 1/tau
template <int CB, int MB>
__device__ __forceinline__ void tile_logits(const f4v (&xq)[4 * CB], const float* WsA, const float* biasL,
                                            const float* wtL, float bt, float base_temp, int lane, int half,
                                            f16v (&z)[MB], float& rtau) {
    constexpr int QN = 4 * CB;
#pragma unroll
    for (int blk = 0; blk < MB; ++blk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f4v b4 = *reinterpret_cast<const f4v*>(biasL + 32 * blk + 8 * q + 4 * half);
#pragma unroll
            for (int j = 0; j < 4; ++j) z[blk][4 * q + j] = b4[j];
        }
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            const f4v w4 = *reinterpret_cast<const f4v*>(WsA + ((blk * QN + q) * 64 + lane) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) z[blk] = mfma(w4[j], xq[q][j], z[blk]);
        }
    }
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < QN; ++q) {
        const f4v w4 = *reinterpret_cast<const f4v*>(wtL + 16 * CB * half + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) t = fmaf(xq[q][j], w4[j], t);
    }
    t += __shfl_xor(t, 32);
    rtau = 1.0f / ada_temp(t + bt, base_temp);
}

template <int CB>
__device__ __forceinline__ void load_rows(f4v (&xq)[4 * CB], const float* __restrict__ x, size_t off, bool ok) {
    const f4v zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4 * CB; ++q) xq[q] = ok ? *reinterpret_cast<const f4v*>(x + off + 4 * q) : zero4;
}

}  // namespace

// ----------------------------------------------------------------------------- slice
// LDS (floats): WsA [MB*4CB*256] | bias [M] | wt [C] | per wave: T [32][36] + X [32][C + 4];
// after the loop the head of it is the workgroup's merged record {m [M], l [M], St [M][C]}.
template <int CB, int MB, int NW>
__global__ __launch_bounds__(NW * 64) void slice_f32_kernel(
    const float* __restrict__ x, const float* __restrict__ Ws, const float* __restrict__ bs,
    const float* __restrict__ wt, float bt, float base_temp, float c0, int N, int H, int chunk,
    float* __restrict__ parts) {
    constexpr int C = 32 * CB, M = 32 * MB, QN = 4 * CB, XS = C + 4, TS = 36;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* WsA = lds;
    float* biasL = WsA + MB * QN * 256;
    float* wtL = biasL + M;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    float* T = wtL + C + wave * (32 * TS + 32 * XS);
    float* X = T + 32 * TS;
    const int bh = blockIdx.y, b = bh / H, h = bh % H, BH = gridDim.y;
    const int n0 = blockIdx.x * chunk, n1 = min(N, n0 + chunk);

    build_ws_image<CB, MB>(WsA, Ws, NW * 64);
    for (int g = threadIdx.x; g < M; g += NW * 64) biasL[g] = bs[g] - c0;
    for (int c = threadIdx.x; c < C; c += NW * 64) wtL[c] = wt[c];
    __syncthreads();

    f16v St[MB][CB], mref[MB], lsum[MB];
#pragma unroll
    for (int blk = 0; blk < MB; ++blk)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            mref[blk][v] = -INFINITY;
            lsum[blk][v] = 0.f;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) St[blk][cb][v] = 0.f;
        }

    const size_t row = (size_t)H * C, col0 = (size_t)h * C + (C / 2) * half;
    const int tiles = (n1 - n0 + 31) / 32;
    f4v xq[QN];
    {
        const int n = n0 + 32 * wave + l31;
        const bool ok = wave < tiles && n < n1;
        load_rows<CB>(xq, x, ((size_t)b * N + (ok ? n : 0)) * row + col0, ok);
    }
    for (int tl = wave; tl < tiles; tl += NW) {
        const int n = n0 + 32 * tl + l31;
        const bool valid = n < n1;
#pragma unroll
        for (int q = 0; q < QN; ++q) *reinterpret_cast<f4v*>(X + l31 * XS + (C / 2) * half + 4 * q) = xq[q];
        f16v z[MB];
        float rtau;
        tile_logits<CB, MB>(xq, WsA, biasL, wtL, bt, base_temp, lane, half, z, rtau);
        {  // next tile's rows: in flight during the rest of this one
            const int nn = n + 32 * NW;
            const bool ok = tl + NW < tiles && nn < n1;
            load_rows<CB>(xq, x, ((size_t)b * N + (ok ? nn : 0)) * row + col0, ok);
        }
        bool over = false;
#pragma unroll
        for (int blk = 0; blk < MB; ++blk)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                z[blk][v] = valid ? z[blk][v] * rtau : -INFINITY;
                over |= z[blk][v] > mref[blk][v] + kRescaleThr;
            }
        if (__any(over)) {  // rare after the first tile: raise the reference maxima, rescale the running sums
#pragma unroll
            for (int blk = 0; blk < MB; ++blk)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const float mn = fmaxf(mref[blk][v], half_max(z[blk][v]));  // row g over this half-wave's 32 points
                    const float f = __expf(mref[blk][v] - mn);                  // -inf -> 0
                    lsum[blk][v] *= f;
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) St[blk][cb][v] *= f;
                    mref[blk][v] = mn;
                }
        }
#pragma unroll
        for (int blk = 0; blk < MB; ++blk)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float p = __expf(z[blk][v] - mref[blk][v]);  // invalid points: e^{-inf} = 0
                lsum[blk][v] += p;
                z[blk][v] = p;
            }
        // St[g][c] += sum_n P[g][n] X[n][c]: both operands through the wave-private tiles
        wave_lds_fence();
        float xb[CB][16];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int k = 0; k < 16; ++k) xb[cb][k] = X[(k + 16 * half) * XS + 32 * cb + l31];
#pragma unroll
        for (int blk = 0; blk < MB; ++blk) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) T[(8 * q + 4 * half + j) * TS + l31] = z[blk][4 * q + j];
            wave_lds_fence();
            f4v d4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) d4[q] = *reinterpret_cast<const f4v*>(T + l31 * TS + 16 * half + 4 * q);
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) St[blk][cb] = mfma(d4[q][j], xb[cb][4 * q + j], St[blk][cb]);
            wave_lds_fence();
        }
    }

    // ---- one record per workgroup: log-sum-exp merge of the waves, in wave order, through LDS
    // St is in accumulator layout (lane = channel); m and l are per row, identical in the 32 lanes of a half
    // after the sum over the half's points.
#pragma unroll
    for (int blk = 0; blk < MB; ++blk)
#pragma unroll
        for (int v = 0; v < 16; ++v) lsum[blk][v] = half_sum(lsum[blk][v]);
    __syncthreads();
    float* rm = lds;            // [M]
    float* rl = rm + M;         // [M]
    float* rS = rl + M;         // [M][C]
    for (int w = 0; w < NW; ++w) {
        if (wave == w) {
#pragma unroll
            for (int blk = 0; blk < MB; ++blk)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int g = 32 * blk + 8 * (v >> 2) + 4 * half + (v & 3);
                    const float m1 = w ? rm[g] : -INFINITY, m2 = mref[blk][v];
                    const float mn = fmaxf(m1, m2);
                    const float a1 = (m1 == -INFINITY) ? 0.f : __expf(m1 - mn);
                    const float a2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) {
                        float* dst = rS + g * C + 32 * cb + l31;
                        *dst = (w ? *dst * a1 : 0.f) + St[blk][cb][v] * a2;
                    }
                    const float ln = (w ? rl[g] * a1 : 0.f) + lsum[blk][v] * a2;
                    wave_lds_fence();  // every lane has read rm / rl of this row before lane 0 replaces them
                    if (l31 == 0) {
                        rm[g] = mn;
                        rl[g] = ln;
                    }
                }
        }
        __syncthreads();
    }
    float* rec = parts + ((size_t)blockIdx.x * BH + bh) * (size_t)M * (C + 2);
    for (int i = threadIdx.x; i < M * (C + 2); i += NW * 64) rec[i] = lds[i];
}

// ----------------------------------------------------------------------------- de-slice
// LDS (floats): WsA | S2T [CB][MB][4][64][4] | bias [M] | cml [M] | wt [C]
template <int CB, int MB, int NW>
__global__ __launch_bounds__(NW * 64) void deslice_f32_kernel(
    const float* __restrict__ x, const float* __restrict__ Ws, const float* __restrict__ bs,
    const float* __restrict__ wt, float bt, float base_temp, float c0,
    const float* __restrict__ col_max, const float* __restrict__ col_sum, const float* __restrict__ S2,
    int N, int H, int chunk, float* __restrict__ out) {
    constexpr int C = 32 * CB, M = 32 * MB, QN = 4 * CB;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* WsA = lds;
    float* S2T = WsA + MB * QN * 256;     // A of the output product: S2[32 blk + 8q + 4 (l/32) + j][32 cb + l%32]
    float* biasL = S2T + CB * MB * 1024;
    float* cmlL = biasL + M;
    float* wtL = cmlL + M;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int n0 = blockIdx.x * chunk, n1 = min(N, n0 + chunk);

    build_ws_image<CB, MB>(WsA, Ws, NW * 64);
    for (int i = threadIdx.x; i < CB * MB * 1024; i += NW * 64) {
        const int j = i & 3, l = (i >> 2) & 63, q = (i >> 8) & 3, blk = (i >> 10) % MB, cb = (i >> 10) / MB;
        S2T[i] = S2[((size_t)bh * M + 32 * blk + 8 * q + 4 * (l >> 5) + j) * C + 32 * cb + (l & 31)];
    }
    for (int g = threadIdx.x; g < M; g += NW * 64) {
        biasL[g] = bs[g] - c0;
        cmlL[g] = col_max[bh * M + g] + __logf(col_sum[bh * M + g]);
    }
    for (int c = threadIdx.x; c < C; c += NW * 64) wtL[c] = wt[c];
    __syncthreads();

    const size_t row = (size_t)H * C, col0 = (size_t)h * C + (C / 2) * half;
    const int tiles = (n1 - n0 + 31) / 32;
    f4v xq[QN];
    {
        const int n = n0 + 32 * wave + l31;
        const bool ok = wave < tiles && n < n1;
        load_rows<CB>(xq, x, ((size_t)b * N + (ok ? n : 0)) * row + col0, ok);
    }
    for (int tl = wave; tl < tiles; tl += NW) {
        const int n = n0 + 32 * tl + l31;
        const bool valid = n < n1;
        f16v z[MB];
        float rtau;
        tile_logits<CB, MB>(xq, WsA, biasL, wtL, bt, base_temp, lane, half, z, rtau);
        {
            const int nn = n + 32 * NW;
            const bool ok = tl + NW < tiles && nn < n1;
            load_rows<CB>(xq, x, ((size_t)b * N + (ok ? nn : 0)) * row + col0, ok);
        }
#pragma unroll
        for (int blk = 0; blk < MB; ++blk)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4v cm4 = *reinterpret_cast<const f4v*>(cmlL + 32 * blk + 8 * q + 4 * half);
#pragma unroll
                for (int j = 0; j < 4; ++j) z[blk][4 * q + j] = __expf(z[blk][4 * q + j] * rtau - cm4[j]);
            }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            f16v o;
#pragma unroll
            for (int v = 0; v < 16; ++v) o[v] = 0.f;
#pragma unroll
            for (int blk = 0; blk < MB; ++blk)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f4v s4 = *reinterpret_cast<const f4v*>(S2T + (((cb * MB + blk) * 4 + q) * 64 + lane) * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o = mfma(s4[j], z[blk][4 * q + j], o);
                }
            if (valid) {
                float* orow = out + ((size_t)b * N + n) * row + (size_t)h * C + 32 * cb + 4 * half;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f4v s = {o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
                    *reinterpret_cast<f4v*>(orow + 8 * q) = s;
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------- host side
bool slice_f32_ok(int C, int M) { return (C == 32 || C == 64) && (M == 32 || M == 64 || M == 128); }
int slice_f32_num_parts(int N, int BH) { return (N + pick_chunk(N, BH) - 1) / pick_chunk(N, BH); }

template <int CB, int MB, int NW>
static int launch_slice(dim3 grid, hipStream_t st, const float* x, const float* Ws, const float* bs, const float* wt,
                        float bt, float base_temp, float c0, int N, int H, int chunk, float* parts) {
    constexpr int C = 32 * CB, M = 32 * MB;
    size_t fl = (size_t)MB * 4 * CB * 256 + M + C + (size_t)NW * (32 * 36 + 32 * (C + 4));
    const size_t rec = (size_t)M * (C + 2);
    if (fl < rec) fl = rec;
    const size_t lds = fl * sizeof(float);
    auto kern = slice_f32_kernel<CB, MB, NW>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(HAET_E_LAUNCH, "slice_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, st, x, Ws, bs, wt, bt, base_temp, c0, N, H, chunk, parts);
    return check_launch("slice_f32_kernel");
}

template <int CB, int MB, int NW>
static int launch_deslice(dim3 grid, hipStream_t st, const float* x, const float* Ws, const float* bs, const float* wt,
                          float bt, float base_temp, float c0, const float* cm, const float* cs, const float* S2,
                          int N, int H, int chunk, float* out) {
    constexpr int C = 32 * CB, M = 32 * MB;
    const size_t lds = ((size_t)MB * 4 * CB * 256 + (size_t)CB * MB * 1024 + 2 * M + C) * sizeof(float);
    auto kern = deslice_f32_kernel<CB, MB, NW>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(HAET_E_LAUNCH, "deslice_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, st, x, Ws, bs, wt, bt, base_temp, c0, cm, cs, S2, N, H, chunk,
                       out);
    return check_launch("deslice_f32_kernel");
}

#define HAET_F32_DISPATCH(FN, ...)                                        \
    switch (C / 32 * 8 + M / 32) {                                        \
        case 8 + 1: return FN<1, 1, 8>(__VA_ARGS__);                      \
        case 8 + 2: return FN<1, 2, 8>(__VA_ARGS__);                      \
        case 8 + 4: return FN<1, 4, 4>(__VA_ARGS__);                      \
        case 16 + 1: return FN<2, 1, 8>(__VA_ARGS__);                     \
        case 16 + 2: return FN<2, 2, 4>(__VA_ARGS__);                     \
        default: return FN<2, 4, 4>(__VA_ARGS__);                         \
    }

int slice_f32_fwd(const float* x, const float* Ws, const float* bs, const float* wt, float bt, float base_temp,
                  float c0, int B, int N, int H, int C, int M, float* parts, hipStream_t st) {
    const int chunk = pick_chunk(N, B * H);
    dim3 grid((N + chunk - 1) / chunk, B * H);
    HAET_F32_DISPATCH(launch_slice, grid, st, x, Ws, bs, wt, bt, base_temp, c0, N, H, chunk, parts)
}

int deslice_f32_fwd(const float* x, const float* Ws, const float* bs, const float* wt, float bt, float base_temp,
                    float c0, const float* cm, const float* cs, const float* S2, int B, int N, int H, int C, int M,
                    float* out, hipStream_t st) {
    const int chunk = pick_chunk(N, B * H);
    dim3 grid((N + chunk - 1) / chunk, B * H);
    HAET_F32_DISPATCH(launch_deslice, grid, st, x, Ws, bs, wt, bt, base_temp, c0, cm, cs, S2, N, H, chunk, out)
}

}  // namespace haet
