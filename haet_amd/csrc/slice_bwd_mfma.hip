// Backward of the slice / de-slice passes on the matrix cores (dim_head C = 32 with slice_num M in {32, 64, 128},
// C = 64 with M in {32, 64}):
// the same two passes as slice_bwd.hip (maths and record layouts in its header comment), with every
// contraction on v_mfma_f32_32x32x2_f32 -- f32 operands, f32 accumulate, exact f32 products, so the
// gradients need no split-precision envelope.
//
// One wave owns tiles of 32 points of one (cloud, head).  Everything point-wise lives in the MFMA
// accumulator layout with the POINT in the lane (column n = lane % 32) and the slice column g in the
// register index (row = 8 (v / 4) + 4 (lane / 32) + v % 4 per 32-row block):
//   Z^T [g][n] = Ws [g][c] X^T [c][n]                         (step 1; X^T straight from global, 16 channels per lane)
//   a^T [g][n] = S2 [g][c] GO^T [c][n] + k dS [g][c] X^T      (step 2)
//   tau, the clamp mask and sum_g dz z are per-lane scalars (one shuffle across the two half-waves)
//   dX^T [c][n] = Ws^T [c][g] dlogit^T [g][n] + k dS^T P^T    (step 3: the accumulators ARE the B operands: the
//                 two half-waves hold rows g and g + 4 of register v, which is one k-pair of the instruction)
//   dWs [g][c] += dlogit [g][n] X [n][c]                      (step 4: the one contraction over the points; both
//                 operands go through a wave-private LDS tile to get n into the register index)
// The constant A operands are per-workgroup LDS images read with one ds_read_b128 per four MFMAs.
#include "common.h"

namespace haet {

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr float kNormK = 1.0f / (1.0f + 1e-5f);
constexpr int kTS = 36;  // row stride (floats) of the wave-private P / dlogit tile: 16-B aligned rows, conflict-free b128 access

__device__ __forceinline__ f16v mfma(float a, float b, f16v c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float half_sum(float v) {  // sum over the 32 lanes of a half-wave
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ void wave_lds_fence() {  // wave-private LDS hand-off: ordering only
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace

// MODE 0: pass 1 (dS2 partials).  MODE 1: pass 2 (dx_lin + parameter-gradient partials).
// CB = C / 32 channel blocks (dim_head 32 or 64), MB = M / 32 slice-column blocks.
template <int CB, int MB, int NW, int MODE>
__global__ __launch_bounds__(NW * 64) void slice_bwd_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ Ws, const float* __restrict__ bs,
    const float* __restrict__ wt, float bt, float base_temp, float c0,
    const float* __restrict__ col_max, const float* __restrict__ col_sum,
    const float* __restrict__ go, const float* __restrict__ S2, const float* __restrict__ dS,
    const float* __restrict__ A, int N, int H, int chunk,
    float* __restrict__ dx, float* __restrict__ parts) {
    constexpr int M = 32 * MB, C = 32 * CB, IMG = M * C, QN = 4 * CB, XS = C + 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // A images of the products over the channels, [blk][q < QN][lane][4]: row 32 blk + l%32, channel 4q+j + (C/2)(l/32)
    float* WsA = lds;
    float* S2A = WsA + IMG;                             // MODE 1 from here on
    float* dSA = S2A + (MODE ? IMG : 0);
    // A images of the dx product, [cb][blk][q < 4][lane][4]: row g = 32 blk + 8q + 4 (l/32) + j, channel 32 cb + l%32
    float* WsT = dSA + (MODE ? IMG : 0);
    float* dST = WsT + (MODE ? IMG : 0);
    float* biasL = dST + (MODE ? IMG : 0);              // [M] bs - c0
    float* cmlL = biasL + M;                            // [M] col_max + ln col_sum
    float* ArL = cmlL + M;                              // [M] A
    float* wtL = ArL + M;                               // [C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    float* T = wtL + C + wave * (32 * kTS + 32 * XS);   // [32 g][kTS]: one 32-row block of dlogit / P, n in the columns
    float* X = T + 32 * kTS;                            // [32 n][XS]: the tile's x (MODE 1) or g_out (MODE 0) rows
                                                        //            (+ dt of the point in column C)
    const int bh = blockIdx.y, b = bh / H, h = bh % H, BH = gridDim.y;
    const int n0 = blockIdx.x * chunk, n1 = min(N, n0 + chunk);

    for (int i = threadIdx.x; i < IMG; i += NW * 64) {
        const int j = i & 3, l = (i >> 2) & 63;
        {
            const int q = (i >> 8) % QN, blk = (i >> 8) / QN;
            const int ga = 32 * blk + (l & 31), ca = 4 * q + j + (C / 2) * (l >> 5);
            WsA[i] = Ws[ga * C + ca];
            if (MODE) {
                S2A[i] = S2[((size_t)bh * M + ga) * C + ca];
                dSA[i] = kNormK * dS[((size_t)bh * M + ga) * C + ca];
            }
        }
        if (MODE) {
            const int q = (i >> 8) & 3, blk = (i >> 10) % MB, cb = (i >> 10) / MB;
            const int gt = 32 * blk + 8 * q + 4 * (l >> 5) + j, ct = 32 * cb + (l & 31);
            WsT[i] = Ws[gt * C + ct];
            dST[i] = kNormK * dS[((size_t)bh * M + gt) * C + ct];
        }
    }
    for (int g = threadIdx.x; g < M; g += NW * 64) {
        biasL[g] = bs[g] - c0;
        cmlL[g] = col_max[bh * M + g] + __logf(col_sum[bh * M + g]);
        ArL[g] = MODE ? A[bh * M + g] : 0.f;
    }
    for (int c = threadIdx.x; c < C; c += NW * 64) wtL[c] = wt[c];
    __syncthreads();

    // persistent: dWs (MODE 1) / dS2 (MODE 0) in accumulator layout; dbs of row g = 32 blk + lane % 32 (this
    // half-wave's 16 points of every tile); dwt of channel 32 cb + lane % 32 (ditto); dbt of point column lane % 32
    f16v dW[MB][CB];
    float dbr[MB], dwt[CB], dbt = 0.f;
#pragma unroll
    for (int blk = 0; blk < MB; ++blk) {
        dbr[blk] = 0.f;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int v = 0; v < 16; ++v) dW[blk][cb][v] = 0.f;
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) dwt[cb] = 0.f;

    const size_t row = (size_t)H * C;
    const size_t col0 = (size_t)h * C + (C / 2) * half;
    const int tiles = (n1 - n0 + 31) / 32;
    const f4v zero4 = {0.f, 0.f, 0.f, 0.f};

    f4v xq[QN], gq[QN];  // this lane's C/2 channels of its point: x_lin and g_out
    {
        const int n = n0 + 32 * wave + l31;
        const bool ok = wave < tiles && n < n1;
        const size_t off = ((size_t)b * N + (ok ? n : 0)) * row + col0;
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            xq[q] = ok ? *reinterpret_cast<const f4v*>(x + off + 4 * q) : zero4;
            gq[q] = ok ? *reinterpret_cast<const f4v*>(go + off + 4 * q) : zero4;
        }
    }
    for (int tl = wave; tl < tiles; tl += NW) {
        const int n = n0 + 32 * tl + l31;
        const bool valid = n < n1;
        // the tile's rows for step 4, channel in the lane after the round trip
#pragma unroll
        for (int q = 0; q < QN; ++q)
            *reinterpret_cast<f4v*>(X + l31 * XS + (C / 2) * half + 4 * q) = MODE ? xq[q] : gq[q];

        // ---- step 1: logits, Ada-Temp
        f16v z[MB];
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
            const f4v w4 = *reinterpret_cast<const f4v*>(wtL + (C / 2) * half + 4 * q);
#pragma unroll
            for (int j = 0; j < 4; ++j) t = fmaf(xq[q][j], w4[j], t);
        }
        t += __shfl_xor(t, 32);
        t += bt;
        const float rtau = 1.0f / ada_temp(t, base_temp);

        // ---- step 2: a = g_out . S2 + k x . dS
        f16v a[MB];
        if (MODE) {
#pragma unroll
            for (int blk = 0; blk < MB; ++blk) {
#pragma unroll
                for (int v = 0; v < 16; ++v) a[blk][v] = 0.f;
#pragma unroll
                for (int q = 0; q < QN; ++q) {
                    const f4v s4 = *reinterpret_cast<const f4v*>(S2A + ((blk * QN + q) * 64 + lane) * 4);
                    const f4v d4 = *reinterpret_cast<const f4v*>(dSA + ((blk * QN + q) * 64 + lane) * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        a[blk] = mfma(s4[j], gq[q][j], a[blk]);
                        a[blk] = mfma(d4[j], xq[q][j], a[blk]);
                    }
                }
            }
        }

        // ---- point-wise: softmax weight, dz, dlogit, dtau
        float part = 0.f;
#pragma unroll
        for (int blk = 0; blk < MB; ++blk)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4v cm4 = *reinterpret_cast<const f4v*>(cmlL + 32 * blk + 8 * q + 4 * half);
                const f4v A4 = *reinterpret_cast<const f4v*>(ArL + 32 * blk + 8 * q + 4 * half);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int v = 4 * q + j;
                    const float zz = z[blk][v] * rtau;
                    const float p = valid ? __expf(zz - cm4[j]) : 0.f;
                    if (MODE) {
                        const float dz = p * (a[blk][v] - A4[j]);
                        part = fmaf(dz, zz, part);
                        const float dl = dz * rtau;
                        z[blk][v] = dl;
                        a[blk][v] = p;
                    } else {
                        z[blk][v] = p;
                    }
                }
            }
        float dt = 0.f;
        if (MODE) {
            part += __shfl_xor(part, 32);
            const float dtau = -part * rtau;
            dt = (t >= -0.4f && t <= 0.4f) ? dtau : 0.f;  // clamp passes the gradient inside its range
            dbt += half ? 0.f : dt;
            if (half == 0) X[l31 * XS + C] = dt;  // pad column of the x tile: dt by point, for dwt in step 4
        }

        // next tile's rows: in flight during steps 3 and 4
        {
            const int nn = n + 32 * NW;
            const bool ok = tl + NW < tiles && nn < n1;
            const size_t off = ((size_t)b * N + (ok ? nn : 0)) * row + col0;
#pragma unroll
            for (int q = 0; q < QN; ++q) {
                xq[q] = ok ? *reinterpret_cast<const f4v*>(x + off + 4 * q) : zero4;
                gq[q] = ok ? *reinterpret_cast<const f4v*>(go + off + 4 * q) : zero4;
            }
        }

        // ---- step 3: dx^T [c][n], one 32-channel block at a time
        if (MODE) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                f16v o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f4v w4 = *reinterpret_cast<const f4v*>(wtL + 32 * cb + 8 * q + 4 * half);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[4 * q + j] = dt * w4[j];
                }
#pragma unroll
                for (int blk = 0; blk < MB; ++blk)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f4v w4 = *reinterpret_cast<const f4v*>(WsT + (((cb * MB + blk) * 4 + q) * 64 + lane) * 4);
                        const f4v d4 = *reinterpret_cast<const f4v*>(dST + (((cb * MB + blk) * 4 + q) * 64 + lane) * 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            o = mfma(w4[j], z[blk][4 * q + j], o);
                            o = mfma(d4[j], a[blk][4 * q + j], o);
                        }
                    }
                if (valid) {
                    float* orow = dx + ((size_t)b * N + n) * row + (size_t)h * C + 32 * cb + 4 * half;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f4v s = {o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
                        *reinterpret_cast<f4v*>(orow + 8 * q) = s;
                    }
                }
            }
        }

        // ---- step 4: the contraction over the tile's points, one 32-row block at a time
        float xb[CB][16];
        wave_lds_fence();  // X written at the top of the tile
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int k = 0; k < 16; ++k) xb[cb][k] = X[(k + 16 * half) * XS + 32 * cb + l31];
        if (MODE) {  // dwt[c] += sum_n dt_n x[n][c] over this half-wave's 16 points
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float dtk = X[(k + 16 * half) * XS + C];
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) dwt[cb] = fmaf(dtk, xb[cb][k], dwt[cb]);
            }
        }
#pragma unroll
        for (int blk = 0; blk < MB; ++blk) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) T[(8 * q + 4 * half + j) * kTS + l31] = z[blk][4 * q + j];
            wave_lds_fence();
            f4v d4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) d4[q] = *reinterpret_cast<const f4v*>(T + l31 * kTS + 16 * half + 4 * q);
            if (MODE) {  // dbs[g] += sum_n dlogit[n][g]: the row sums of the tile
#pragma unroll
                for (int q = 0; q < 4; ++q) dbr[blk] += (d4[q][0] + d4[q][1]) + (d4[q][2] + d4[q][3]);
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) dW[blk][cb] = mfma(d4[q][j], xb[cb][4 * q + j], dW[blk][cb]);
            wave_lds_fence();  // reads done before the next block / tile overwrites T and X
        }
    }

    // ---- workgroup record: sum the waves in a fixed order through the (now idle) image region
    __syncthreads();
    float* red = lds;                  // [M][C] dWs | [M] dbs | [C] dwt | dbt
    if (MODE) {
#pragma unroll
        for (int blk = 0; blk < MB; ++blk) dbr[blk] += __shfl_xor(dbr[blk], 32);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) dwt[cb] += __shfl_xor(dwt[cb], 32);
        dbt = half_sum(dbt);
    }
    for (int w = 0; w < NW; ++w) {
        if (wave == w) {
#pragma unroll
            for (int blk = 0; blk < MB; ++blk) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int g = 32 * blk + 8 * (v >> 2) + 4 * half + (v & 3);
                        float* dst = red + g * C + 32 * cb + l31;
                        *dst = (w ? *dst : 0.f) + dW[blk][cb][v];
                    }
                if (MODE && half == 0) red[IMG + 32 * blk + l31] = (w ? red[IMG + 32 * blk + l31] : 0.f) + dbr[blk];
            }
            if (MODE && half == 0) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    red[IMG + M + 32 * cb + l31] = (w ? red[IMG + M + 32 * cb + l31] : 0.f) + dwt[cb];
                if (l31 == 0) red[IMG + M + C] = (w ? red[IMG + M + C] : 0.f) + dbt;
            }
        }
        __syncthreads();
    }
    const int R = MODE ? IMG + M + C + 1 : IMG;
    float* rec = parts + ((size_t)blockIdx.x * BH + bh) * (size_t)R;
    for (int i = threadIdx.x; i < R; i += NW * 64) rec[i] = red[i];
}

// ----------------------------------------------------------------------------- host side
template <int CB, int MB, int NW, int MODE>
static int launch(dim3 grid, hipStream_t st, const float* x, const float* Ws, const float* bs, const float* wt,
                  float bt, float base_temp, float c0, const float* cm, const float* cs, const float* go,
                  const float* S2, const float* dS, const float* A, int N, int H, int chunk, float* dx,
                  float* parts) {
    constexpr int M = 32 * MB, C = 32 * CB;
    size_t fl = (size_t)(MODE ? 5 : 1) * M * C + 3 * M + C + (size_t)NW * (32 * kTS + 32 * (C + 4));
    const size_t rec = (size_t)M * C + M + C + 1;  // the workgroup record is assembled over the images
    if (fl < rec) fl = rec;
    const size_t lds = fl * sizeof(float);
    auto kern = slice_bwd_mfma_kernel<CB, MB, NW, MODE>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(HAET_E_LAUNCH, "slice_bwd_mfma: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, st, x, Ws, bs, wt, bt, base_temp, c0, cm, cs, go, S2, dS, A,
                       N, H, chunk, dx, parts);
    return check_launch(MODE ? "slice_bwd_mfma_kernel<points>" : "slice_bwd_mfma_kernel<tokens>");
}

// C = 32 with M in {32, 64, 128}; C = 64 with M in {32, 64} (M = 128 would need 164 KB of operand images)
bool slice_bwd_mfma_ok(int C, int M) {
    return (C == 32 && (M == 32 || M == 64 || M == 128)) || (C == 64 && (M == 32 || M == 64));
}

// go / S2 / dS / A / dx as in haet_slice_bwd_points; mode 0 ignores S2, dS, A, dx
int slice_bwd_mfma(int mode, const float* x, const float* Ws, const float* bs, const float* wt, float bt,
                   float base_temp, float c0, const float* cm, const float* cs, const float* go, const float* S2,
                   const float* dS, const float* A, int B, int N, int H, int C, int M, float* dx, float* parts,
                   hipStream_t st) {
    const int chunk = slice_bwd_pick_chunk(N, kBwdF32);
    dim3 grid((N + chunk - 1) / chunk, B * H);
#define HAET_BWD_MFMA(CB, MB, NW)                                                                                     \
    return mode ? launch<CB, MB, NW, 1>(grid, st, x, Ws, bs, wt, bt, base_temp, c0, cm, cs, go, S2, dS, A, N, H, chunk, \
                                        dx, parts)                                                                    \
                : launch<CB, MB, NW, 0>(grid, st, x, Ws, bs, wt, bt, base_temp, c0, cm, cs, go, S2, dS, A, N, H, chunk, \
                                        dx, parts)
    if (C == 32) {
        switch (M) {
            case 32: HAET_BWD_MFMA(1, 1, 8);
            case 64: HAET_BWD_MFMA(1, 2, 8);
            default: HAET_BWD_MFMA(1, 4, 4);
        }
    }
    if (M == 32) { HAET_BWD_MFMA(2, 1, 8); }
    HAET_BWD_MFMA(2, 2, 4);
#undef HAET_BWD_MFMA
}

}  // namespace haet
