// Backward of the slice / de-slice passes on the BF16 matrix cores (dim_head C = 32, slice_num M in {32, 64}): the two
// passes of slice_bwd_mfma.hip (maths and record layouts: slice_bwd.hip's header) with every contraction as a split-bf16
// v_mfma_f32_32x32x16_bf16 triple (hi*hi + hi*lo + lo*hi, fp32 accumulate: ~2^-16 per product, the forward's grade)
// instead of the exact-f32 v_mfma_f32_32x32x2_f32, which runs at 1/16 of the rate: pass 2 was 200 GFLOP at 62 % of the
// f32 matrix peak (2.05 ms per 1M points), three times its HBM time.
//
// One wave owns 32-point tiles of one (cloud, head); the POINT sits on the lane and the slice column g in the register
// index of the accumulator (row = 8 (v / 4) + 4 (lane / 32) + v % 4), as in the f32 kernel:
//   Z^T [g][n]  = Ws [g][c] X^T [c][n]                   A = constant fragments (LDS), B = the lane's channels, split once
//   a^T [g][n]  = S2 [g][c] GO^T [c][n] + k dS [g][c] X^T
//   dX^T [c][n] = Ws^T [c][g] dl^T [g][n] + k dS^T P^T    B = the ACCUMULATORS dl, P as they stand (k in accumulator order:
//                                                         the constant A fragments are laid out to match)
//   dWs [g][c] += dl [n][g]^T X [n][c]                    the one contraction over the points: dl and X go through wave-private
//                                                         {hi, lo} bf16 images [32 points][32] and come back as fragments over
//                                                         the ROW index with ds_read_b64_tr_b16 (the recipe of gemm_tn.hip)
// The row sums db_s, dw_t, db_t are per-lane running sums reduced once at the end of the kernel.
#include <cstdlib>

#include "common.h"

namespace haet {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 xb_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 xb_bf16x2;
typedef __attribute__((ext_vector_type(16))) float xb_f16v;
typedef __attribute__((ext_vector_type(4))) float xb_f4v;
typedef __attribute__((ext_vector_type(2))) float xb_f2v;
typedef __attribute__((ext_vector_type(4))) unsigned xb_u4;
typedef __attribute__((ext_vector_type(2))) unsigned xb_u2;
typedef __attribute__((ext_vector_type(4))) short xb_s16x4;

constexpr float kNormK = 1.0f / (1.0f + 1e-5f);
constexpr int kImgB = 2048 + 32;  // bytes of one [32][32] bf16 image plane (+ 8 banks of skew between planes)

extern __shared__ __attribute__((aligned(16))) char xsm[];

__device__ __forceinline__ unsigned xb_pack(float a, float b) {
    xb_f2v v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, xb_bf16x2));
}
struct XFrag {
    xb_u4 hi, lo;
};
__device__ __forceinline__ XFrag xb_split8(const float (&v)[8]) {
    XFrag f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h, l;
        split_bf16_pair(v[2 * i], v[2 * i + 1], h, l);
        f.hi[i] = h, f.lo[i] = l;
    }
    return f;
}
template <int S>
__device__ __forceinline__ XFrag xb_acc_frag(const xb_f16v& a) {  // registers 8S .. 8S+7 of an accumulator = k-step S
    const float v[8] = {a[8 * S], a[8 * S + 1], a[8 * S + 2], a[8 * S + 3], a[8 * S + 4], a[8 * S + 5], a[8 * S + 6], a[8 * S + 7]};
    return xb_split8(v);
}
__device__ __forceinline__ xb_f16v xb_mfma3(const XFrag& a, const XFrag& b, xb_f16v c) {
    const xb_bf16x8 ah = __builtin_bit_cast(xb_bf16x8, a.hi), al = __builtin_bit_cast(xb_bf16x8, a.lo);
    const xb_bf16x8 bh = __builtin_bit_cast(xb_bf16x8, b.hi), bl = __builtin_bit_cast(xb_bf16x8, b.lo);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ int xb_img_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// constant fragment piece `p` (1 KB: 64 lanes x 16 B; hi plane at p, lo plane 1 KB later)
__device__ __forceinline__ XFrag xb_const(int base, int piece, int lane) {
    XFrag f;
    const int off = base + piece * 2048 + lane * 16;
    f.hi = *reinterpret_cast<const xb_u4*>(xsm + off);
    f.lo = *reinterpret_cast<const xb_u4*>(xsm + off + 1024);
    return f;
}
// fragment of k-step s (rows 16 s .. 16 s + 15) for column lane % 32 of a wave-private image: element j of lane half hf
// is row 16 s + 8 (j >> 2) + 4 hf + (j & 3) -- the same order for both operands of the product
__device__ __forceinline__ XFrag xb_read_cols(int img, int s, int lane) {
    const int i = lane & 15, cb = (lane >> 4) & 1, hf = lane >> 5;
    const int chunk = 2 * cb + ((i & 3) >> 1), sub = (i & 1) << 3;
    XFrag f;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const int base = img + part * kImgB;
        const int r0 = 16 * s + 4 * hf + (i >> 2), r1 = r0 + 8;
        const xb_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) xb_s16x4*)(xsm + base + xb_img_off(r0, chunk) + sub));
        const xb_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) xb_s16x4*)(xsm + base + xb_img_off(r1, chunk) + sub));
        const xb_u2 a2 = __builtin_bit_cast(xb_u2, a), b2 = __builtin_bit_cast(xb_u2, b);
        const xb_u4 v = {a2[0], a2[1], b2[0], b2[1]};
        if (part == 0) f.hi = v;
        else f.lo = v;
    }
    return f;
}
__device__ __forceinline__ void xb_wave_fence() {  // wave-private LDS hand-off: ordering only
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float xb_half_sum(float v) {  // sum over the 32 lanes of a half-wave
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

}  // namespace

// MODE 0: pass 1 (dS2 partials).  MODE 1: pass 2 (dx_lin + parameter-gradient partials).  C = 32, M = 32 MB.
template <int MB, int NW, int MODE>
__global__ __launch_bounds__(NW * 64) void slice_bwd_x3_kernel(
    const float* __restrict__ x, const float* __restrict__ Ws, const float* __restrict__ bs,
    const float* __restrict__ wt, float bt, float base_temp, float c0,
    const float* __restrict__ col_max, const float* __restrict__ col_sum,
    const float* __restrict__ go, const float* __restrict__ S2, const float* __restrict__ dS,
    const float* __restrict__ A, int N, int H, int chunk,
    float* __restrict__ dx, float* __restrict__ parts) {
    constexpr int M = 32 * MB, C = 32, IMG = M * C;
    // LDS (bytes): constant fragment pieces (2 KB each: hi | lo) | vectors | wave-private images
    constexpr int nWsA = MB * 2, nT = 2 * MB;  // pieces per operand: [blk][k-step over c] / [k-step over g]
    constexpr int oWsA = 0, oS2A = oWsA + nWsA * 2048, odSA = oS2A + (MODE ? nWsA * 2048 : 0);
    constexpr int oWsT = odSA + (MODE ? nWsA * 2048 : 0), odST = oWsT + (MODE ? nT * 2048 : 0);
    constexpr int oVec = odST + (MODE ? nT * 2048 : 0);          // biasL [M] | cmlL [M] | ArL [M] | wtL [C]
    constexpr int oWave = oVec + (3 * M + C) * 4;
    constexpr int kWaveB = (1 + MB) * 2 * kImgB;                 // XI (or GO) image + MB dl (or P) images
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hf = lane >> 5;
    float* biasL = reinterpret_cast<float*>(xsm + oVec);
    float* cmlL = biasL + M;
    float* ArL = cmlL + M;
    float* wtL = ArL + M;
    const int ximg = oWave + wave * kWaveB, dimg = ximg + 2 * kImgB;  // dimg + blk * 2 * kImgB
    const int bh = blockIdx.y, b = bh / H, h = bh % H, BH = gridDim.y;
    const int n0 = blockIdx.x * chunk, n1 = min(N, n0 + chunk);

    // ---- constant fragments, built once per workgroup.  Piece p of an operand, lane (r, hf), element j:
    //   A of the products over the channels (WsA, S2A, dSA): p = blk * 2 + s: row g = 32 blk + r, k = channel 16 s + 8 hf + j
    //   A of the dx product (WsT, dST): p = s4: row c = r, k = slice g = 16 s4 + 8 (j >> 2) + 4 hf + (j & 3) (accumulator order)
    {
        constexpr int nPieces = nWsA * (MODE ? 3 : 1) + (MODE ? 2 * nT : 0);
        for (int i = threadIdx.x; i < nPieces * 64; i += NW * 64) {
            const int ln = i & 63, r = ln & 31, f = ln >> 5;
            int p = i >> 6, base;
            const float* src;
            float scale = 1.f;
            bool transposed = false;
            if (p < nWsA) {
                base = oWsA, src = Ws;
            } else if (MODE && (p -= nWsA) < nWsA) {
                base = oS2A, src = S2 + (size_t)bh * IMG;
            } else if (MODE && (p -= nWsA) < nWsA) {
                base = odSA, src = dS + (size_t)bh * IMG, scale = kNormK;
            } else if (MODE && (p -= nWsA) < nT) {
                base = oWsT, src = Ws, transposed = true;
            } else {
                p -= nT;
                base = odST, src = dS + (size_t)bh * IMG, scale = kNormK, transposed = true;
            }
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int g, c;
                if (!transposed) {
                    g = 32 * (p >> 1) + r;
                    c = 16 * (p & 1) + 8 * f + j;
                } else {
                    g = 16 * p + 8 * (j >> 2) + 4 * f + (j & 3);
                    c = r;
                }
                v[j] = scale * src[g * C + c];
            }
            const XFrag fr = xb_split8(v);
            *reinterpret_cast<xb_u4*>(xsm + base + p * 2048 + ln * 16) = fr.hi;
            *reinterpret_cast<xb_u4*>(xsm + base + p * 2048 + 1024 + ln * 16) = fr.lo;
        }
        for (int g = threadIdx.x; g < M; g += NW * 64) {
            biasL[g] = bs[g] - c0;
            cmlL[g] = col_max[bh * M + g] + __logf(col_sum[bh * M + g]);
            ArL[g] = MODE ? A[bh * M + g] : 0.f;
        }
        for (int c = threadIdx.x; c < C; c += NW * 64) wtL[c] = wt[c];
    }
    __syncthreads();

    // persistent sums: dWs (MODE 1) / dS2 (MODE 0) in accumulator layout (row g, column c on the lane); per-lane running
    // sums of dl (-> db_s), dt x (-> dw_t) and dt (-> db_t) over this lane's points
    xb_f16v dW[MB];
    float dbacc[MB][16], dwacc[16], dbt = 0.f;
#pragma unroll
    for (int blk = 0; blk < MB; ++blk)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            dW[blk][v] = 0.f;
            dbacc[blk][v] = 0.f;
        }
#pragma unroll
    for (int k = 0; k < 16; ++k) dwacc[k] = 0.f;

    const size_t row = (size_t)H * C;
    const size_t col0 = (size_t)h * C + 8 * hf;  // this lane's channels: 8 hf .. 8 hf + 7 and 16 + 8 hf .. 16 + 8 hf + 7
    const int tiles = (n1 - n0 + 31) / 32;
    const xb_f4v zero4 = {0.f, 0.f, 0.f, 0.f};

    xb_f4v xq[4], gq[4];  // x_lin and g_out: [2 s], [2 s + 1] = the 8 channels of k-step s
    {
        const int n = n0 + 32 * wave + l31;
        const bool ok = wave < tiles && n < n1;
        const size_t off = ((size_t)b * N + (ok ? n : 0)) * row + col0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int co = 16 * (q >> 1) + 4 * (q & 1);
            xq[q] = ok ? *reinterpret_cast<const xb_f4v*>(x + off + co) : zero4;
            gq[q] = ok ? *reinterpret_cast<const xb_f4v*>(go + off + co) : zero4;
        }
    }
    for (int tl = wave; tl < tiles; tl += NW) {
        const int n = n0 + 32 * tl + l31;
        const bool valid = n < n1;
        // this lane's channels as B fragments (natural k order), and the image of the tile's rows for the contraction
        // over the points: row = point, 16-B chunk 2 s + hf = this lane's 8 channels of k-step s
        XFrag xf[2], gf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float xv[8] = {xq[2 * s][0], xq[2 * s][1], xq[2 * s][2], xq[2 * s][3], xq[2 * s + 1][0], xq[2 * s + 1][1], xq[2 * s + 1][2], xq[2 * s + 1][3]};
            const float gv[8] = {gq[2 * s][0], gq[2 * s][1], gq[2 * s][2], gq[2 * s][3], gq[2 * s + 1][0], gq[2 * s + 1][1], gq[2 * s + 1][2], gq[2 * s + 1][3]};
            xf[s] = xb_split8(xv);
            gf[s] = xb_split8(gv);
            const XFrag& im = MODE ? xf[s] : gf[s];
            const int off = ximg + xb_img_off(l31, 2 * s + hf);
            *reinterpret_cast<xb_u4*>(xsm + off) = im.hi;
            *reinterpret_cast<xb_u4*>(xsm + off + kImgB) = im.lo;
        }

        // Ada-Temp of the point
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const xb_f4v w4 = *reinterpret_cast<const xb_f4v*>(wtL + 8 * hf + 16 * (q >> 1) + 4 * (q & 1));
#pragma unroll
            for (int j = 0; j < 4; ++j) t = fmaf(xq[q][j], w4[j], t);
        }
        t += __shfl_xor(t, 32);
        t += bt;
        const float rtau = 1.0f / ada_temp(t, base_temp);

        // One 32-column block of slices at a time through steps 1-3 (its logits and `a` are dead before the next block
        // starts: with both blocks alive the M = 64 variant spilled 32 registers).  dx^T = dt w_t + sum over the blocks; the
        // dt term needs every block's part of sum_g dz z and is added last.
        xb_f16v o;
#pragma unroll
        for (int v = 0; v < 16; ++v) o[v] = 0.f;
        float part = 0.f;
#pragma unroll
        for (int blk = 0; blk < MB; ++blk) {
            // ---- step 1: logits
            xb_f16v z;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const xb_f4v b4 = *reinterpret_cast<const xb_f4v*>(biasL + 32 * blk + 8 * q + 4 * hf);
#pragma unroll
                for (int j = 0; j < 4; ++j) z[4 * q + j] = b4[j];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) z = xb_mfma3(xb_const(oWsA, 2 * blk + s, lane), xf[s], z);
            // ---- step 2: a = g_out . S2 + k x . dS
            xb_f16v a;
            if (MODE) {
#pragma unroll
                for (int v = 0; v < 16; ++v) a[v] = 0.f;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    a = xb_mfma3(xb_const(oS2A, 2 * blk + s, lane), gf[s], a);
                    a = xb_mfma3(xb_const(odSA, 2 * blk + s, lane), xf[s], a);
                }
            }
            // ---- point-wise: softmax weight, dz, dlogit
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const xb_f4v cm4 = *reinterpret_cast<const xb_f4v*>(cmlL + 32 * blk + 8 * q + 4 * hf);
                const xb_f4v A4 = *reinterpret_cast<const xb_f4v*>(ArL + 32 * blk + 8 * q + 4 * hf);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int v = 4 * q + j;
                    const float zz = z[v] * rtau;
                    const float p = valid ? __expf(zz - cm4[j]) : 0.f;
                    if (MODE) {
                        const float dz = p * (a[v] - A4[j]);
                        part = fmaf(dz, zz, part);
                        const float dl = dz * rtau;
                        z[v] = dl;
                        a[v] = p;
                        dbacc[blk][v] += dl;
                    } else {
                        z[v] = p;
                    }
                }
            }
            // ---- the split of dl (MODE 0: of P) serves step 3 as B fragments and step 4 as image rows: row = this lane's
            // point; elements 0-3 of k-step s are slice columns 16 s + 4 hf .. + 3 (chunk 2 s, 8-B half hf), elements 4-7
            // columns 16 s + 8 + 4 hf .. (chunk 2 s + 1)
            const XFrag d0 = xb_acc_frag<0>(z), d1 = xb_acc_frag<1>(z);
            const int ib = dimg + blk * (2 * kImgB);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const XFrag& d = s ? d1 : d0;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int off = ib + xb_img_off(l31, 2 * s + e) + 8 * hf;
                    const xb_u2 hi = {d.hi[2 * e], d.hi[2 * e + 1]}, lo = {d.lo[2 * e], d.lo[2 * e + 1]};
                    *reinterpret_cast<xb_u2*>(xsm + off) = hi;
                    *reinterpret_cast<xb_u2*>(xsm + off + kImgB) = lo;
                }
            }
            // ---- step 3: this block's share of dx^T
            if (MODE) {
                o = xb_mfma3(xb_const(oWsT, 2 * blk, lane), d0, o);
                o = xb_mfma3(xb_const(oWsT, 2 * blk + 1, lane), d1, o);
                o = xb_mfma3(xb_const(odST, 2 * blk, lane), xb_acc_frag<0>(a), o);
                o = xb_mfma3(xb_const(odST, 2 * blk + 1, lane), xb_acc_frag<1>(a), o);
            }
        }
        if (MODE) {
            part += __shfl_xor(part, 32);
            const float dtau = -part * rtau;
            const float dt = (t >= -0.4f && t <= 0.4f) ? dtau : 0.f;  // clamp passes the gradient inside its range
            dbt += dt;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) dwacc[4 * q + j] = fmaf(dt, xq[q][j], dwacc[4 * q + j]);
            if (valid) {
                float* orow = dx + ((size_t)b * N + n) * row + (size_t)h * C + 4 * hf;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const xb_f4v w4 = *reinterpret_cast<const xb_f4v*>(wtL + 8 * q + 4 * hf);
                    const xb_f4v s4 = {fmaf(dt, w4[0], o[4 * q]), fmaf(dt, w4[1], o[4 * q + 1]), fmaf(dt, w4[2], o[4 * q + 2]),
                                       fmaf(dt, w4[3], o[4 * q + 3])};
                    *reinterpret_cast<xb_f4v*>(orow + 8 * q) = s4;
                }
            }
        }

        // next tile's rows: in flight during step 4 and the next tile's split
        {
            const int nn = n + 32 * NW;
            const bool ok = tl + NW < tiles && nn < n1;
            const size_t off = ((size_t)b * N + (ok ? nn : 0)) * row + col0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = 16 * (q >> 1) + 4 * (q & 1);
                xq[q] = ok ? *reinterpret_cast<const xb_f4v*>(x + off + co) : zero4;
                gq[q] = ok ? *reinterpret_cast<const xb_f4v*>(go + off + co) : zero4;
            }
        }

        // ---- step 4: the contraction over the tile's points
        xb_wave_fence();  // the images of this tile are complete (every lane of the wave takes part in the transposed reads)
        {
            const XFrag b0 = xb_read_cols(ximg, 0, lane), b1 = xb_read_cols(ximg, 1, lane);
#pragma unroll
            for (int blk = 0; blk < MB; ++blk) {
                const int ib = dimg + blk * (2 * kImgB);
                dW[blk] = xb_mfma3(xb_read_cols(ib, 0, lane), b0, dW[blk]);
                dW[blk] = xb_mfma3(xb_read_cols(ib, 1, lane), b1, dW[blk]);
            }
        }
        xb_wave_fence();  // reads done before the next tile overwrites the images
    }

    // ---- workgroup record: sum the waves in a fixed order through the (now idle) constant region
    // db_s [g] = sum over the points (lanes of a half-wave, both halves hold disjoint g) of dbacc; dw_t [c], db_t likewise
    float dbr[MB][16];
    if (MODE) {
#pragma unroll
        for (int blk = 0; blk < MB; ++blk)
#pragma unroll
            for (int v = 0; v < 16; ++v) dbr[blk][v] = xb_half_sum(dbacc[blk][v]);
#pragma unroll
        for (int k = 0; k < 16; ++k) dwacc[k] = xb_half_sum(dwacc[k]);
        dbt = xb_half_sum(dbt);
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(xsm);  // [M][C] dWs | [M] dbs | [C] dwt | dbt
    for (int w = 0; w < NW; ++w) {
        if (wave == w) {
#pragma unroll
            for (int blk = 0; blk < MB; ++blk) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int g = 32 * blk + 8 * (v >> 2) + 4 * hf + (v & 3);
                    float* dst = red + g * C + l31;
                    *dst = (w ? *dst : 0.f) + dW[blk][v];
                    if (MODE && l31 == 0) red[IMG + g] = (w ? red[IMG + g] : 0.f) + dbr[blk][v];
                }
            }
            if (MODE && l31 == 0) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int c = 8 * hf + 16 * (k >> 3) + (k & 7);  // xq[q][j]: channel 8 hf + 16 (q >> 1) + 4 (q & 1) + j, k = 4 q + j
                    red[IMG + M + c] = (w ? red[IMG + M + c] : 0.f) + dwacc[k];
                }
                if (hf == 0) red[IMG + M + C] = (w ? red[IMG + M + C] : 0.f) + dbt;
            }
        }
        __syncthreads();
    }
    const int R = MODE ? IMG + M + C + 1 : IMG;
    float* rec = parts + ((size_t)blockIdx.x * BH + bh) * (size_t)R;
    for (int i = threadIdx.x; i < R; i += NW * 64) rec[i] = red[i];
}

template <int MB, int NW, int MODE>
static int x3_launch(dim3 grid, hipStream_t st, const float* x, const float* Ws, const float* bs, const float* wt,
                     float bt, float base_temp, float c0, const float* cm, const float* cs, const float* go,
                     const float* S2, const float* dS, const float* A, int N, int H, int chunk, float* dx,
                     float* parts) {
    constexpr int M = 32 * MB, C = 32;
    const size_t consts = (size_t)(MB * 2 * (MODE ? 3 : 1) + (MODE ? 2 * 2 * MB : 0)) * 2048;
    size_t lds = consts + (3 * M + C) * 4 + (size_t)NW * (1 + MB) * 2 * kImgB;
    const size_t rec = ((size_t)M * C + M + C + 1) * 4;  // the workgroup record is assembled over the constants
    if (lds < rec) lds = rec;
    auto kern = slice_bwd_x3_kernel<MB, NW, MODE>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(HAET_E_LAUNCH, "slice_bwd_x3: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, st, x, Ws, bs, wt, bt, base_temp, c0, cm, cs, go, S2, dS, A,
                       N, H, chunk, dx, parts);
    return check_launch(MODE ? "slice_bwd_x3_kernel<points>" : "slice_bwd_x3_kernel<tokens>");
}

// the shapes this family covers (the precision switches are the dispatcher's business: slice_bwd.hip use_bwd_x3)
bool slice_bwd_x3_ok(int C, int M) { return C == 32 && (M == 32 || M == 64); }

// same contract (and the same record layout and chunking) as slice_bwd_mfma
int slice_bwd_x3(int mode, const float* x, const float* Ws, const float* bs, const float* wt, float bt,
                 float base_temp, float c0, const float* cm, const float* cs, const float* go, const float* S2,
                 const float* dS, const float* A, int B, int N, int H, int C, int M, float* dx, float* parts,
                 hipStream_t st) {
    const int chunk = slice_bwd_pick_chunk(N, kBwdX3);
    dim3 grid((N + chunk - 1) / chunk, B * H);
#define HAET_BWD_X3(MB)                                                                                            \
    return mode ? x3_launch<MB, 8, 1>(grid, st, x, Ws, bs, wt, bt, base_temp, c0, cm, cs, go, S2, dS, A, N, H, chunk, \
                                      dx, parts)                                                                   \
                : x3_launch<MB, 8, 0>(grid, st, x, Ws, bs, wt, bt, base_temp, c0, cm, cs, go, S2, dS, A, N, H, chunk, \
                                      dx, parts)
    if (M == 32) { HAET_BWD_X3(1); }
    HAET_BWD_X3(2);
#undef HAET_BWD_X3
}

}  // namespace haet
