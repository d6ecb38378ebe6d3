// MFMA kernels for the two streaming passes (slice, de-slice) -- gfx950 only.
//
// Roofline: both passes are HBM-bound (read H*C*4 B per point; de-slice also writes as
// much): 1024 / 2048 B per point at H*C = 256.  The arithmetic (2*H*C*M MACs per point
// and pass) runs on the matrix cores as split-bf16: every fp32 operand is carried as
// hi + lo bf16 (hi = RNE(x), lo = RNE(x - hi)) and each product is three MFMAs
// (hi*hi + hi*lo + lo*hi) accumulated in fp32 -- ~2^-16 relative operand error, which
// keeps the path inside the 1e-3 fp32 parity bar (plain bf16 does not: SURVEY.md s.7).
//
// Slice pass == single-query-block flash attention with Q = W_s (M slice queries),
// K = V = x_proj tile:
//   Y[n,g]  = X[n,:] . Ws[g,:] + b_g          v_mfma_f32_32x32x16_bf16, A = X rows, B = Ws
//   p[n,g]  = exp2(Y * s_n - m2_g)            s_n = log2(e)/tau_n ; lane = g, regs = n
//   Z[g,c] += sum_n p[n,g] X[n,c]             A = P straight from the accumulator
//                                             (acc tile as next operand), B = X via
//                                             ds_read_b64_tr_b16 (transposed LDS read)
// The column max is a *reference* m2_g (deferred: rescale only when the tile max exceeds it
// by 2^20), so (m, l, St) stay a valid log-sum-exp record without tracking the exact max.
//
// De-slice keeps the POINT on the lane through the whole chain, so nothing is transposed:
//   Yt[g,n] = Ws[g,:] . X[n,:] + b_g          A = Ws rows, B = X rows
//   pt[g,n] = exp2(Yt * s_n - m2_g)
//   Ot[c,n] = sum_g (S'[g,c]/l_g) pt[g,n]     A = S' (pre-scaled), B = P from the accumulator
//
// Tile = 32 points x all heads.  All 8 waves stage it: coalesced 1 KB row loads -> fp32 ->
// {hi, lo} bf16 -> LDS images [head][hi|lo][32 rows][C] (64-B rows, 16-B chunks XOR-swizzled
// by (row>>2)&3: conflict-free for both the b128 row reads and the tr_b16 reads), and
// Ada-Temp tau per (point, head) is reduced in the loader (8 lanes per head, DPP).  Loads
// for tile t+1 are issued before computing tile t and written to the other LDS buffer
// after it (one barrier per tile).  Tiles are dealt round-robin to one workgroup per CU.
//
// SPLIT = true (what the module runs at H = 8 / 16): the input is not fp32 but the projection
// GEMM's fused operand format (haet_project_split_fwd, gemm_x3.hip) -- 16-B cells {4 x bf16 hi,
// 4 x bf16 lo} at the fp32 row stride plus s[n][h] = log2(e)/tau -- and the loader only copies.
// Both passes are bound by vector issue (exponent, exp2, the hi/lo split of P), the GEMM is not.
#include "slice_mfma_dev.h"

namespace haet {

// ----------------------------------------------------------------------------- slice
// IPW items (head, 32-column block) per wave; MB = M/32 column blocks per head; NWV waves.
// Measured at C3 (ablation builds, 1M points): loader + staging + barriers alone 0.24 ms, the
// logits MFMAs +0.10, the Z MFMAs with the hi/lo split of P +0.10, exp2 +0.03 -- the parts add up
// instead of overlapping (two waves per SIMD at ~230 VGPRs).  Tried without gain: 16 waves with
// one item each (128 VGPRs only with spills; 25 % slower), running the two waves of a SIMD in
// opposite stage/compute order.
// CB = dim_head / 32 (round 4; BASELINE config C2 is H = 4, dim_head 64, M = 32): the tile is staged as H * CB virtual
// heads of 32 channels; an item's logits contract over both channel blocks, its P feeds CB accumulators Z.
template <int LI, int IPW, int MB, int NWV = NW, bool SPLIT = false, bool PRIV = false, int CB = 1>
__global__ __launch_bounds__(NWV * 64) void slice_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ Ws, const float* __restrict__ bs,
    const float* __restrict__ wt, float bt, float base_temp, float c0, int N, int H,
    int tiles_per_chunk, float* __restrict__ parts, const float* __restrict__ s_in, int M_rec, int g_off) {
    // M_rec / g_off: the records are laid out for M_rec slice columns and this launch fills columns g_off .. g_off + M
    // (M = 128 runs as two launches of the M = 64 kernel: its four-items-per-wave variant spills)
    constexpr int M = 32 * MB;
    constexpr int IPH = (IPW < MB) ? IPW : MB;  // items of one head handled by this wave
    constexpr int HPW = IPW / IPH;              // heads per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hf = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y, BH = gridDim.y * H;
    // Tiles are dealt round-robin: workgroup i takes tiles i, i + G, i + 2G, ... (G = gridDim.x), so
    // all workgroups sweep the cloud together (neighbouring tiles = neighbouring DRAM pages), every
    // workgroup gets the same number of tiles +-1, and there are only G partial records to merge.
    const int n_tiles = (N + TN - 1) / TN, G = gridDim.x;
    const int t_begin = blockIdx.x, t_end = n_tiles;
    // LDS (byte offsets): tile buffers [2][H][hi|lo] | s_n [2][H][32] | rescale scratch [NW][32] | Ws images
    constexpr int C = kC * CB;
    const int Hv = H * CB;  // virtual heads of 32 channels
    const int buf_bytes = Hv * 2 * IMG;
    const int sarr0 = 2 * buf_bytes, sarr_bytes = Hv * TN * 4;
    const int scratch = sarr0 + 2 * sarr_bytes + wave * 32 * 4;
    const int wimg = sarr0 + 2 * sarr_bytes + NWV * 32 * 4;  // [MB][CB][hi|lo] images of Ws
    const float* xb = x + (size_t)b * N * H * C;

    static_assert(!PRIV || (SPLIT && LI == 4), "wave-private loading: split input, H = 8 (one head per wave)");
    static_assert(CB == 1 || (!SPLIT && !PRIV), "dim_head 64: fp32 input (Ada-Temp in the loader)");
    Loader<LI, NWV * 64, SPLIT, PRIV, CB> ld;
    ld.init(wt, xb, Hv, t_begin * TN, SPLIT ? s_in + (size_t)b * N * H : nullptr);
    ld.tile_stride *= G;
    if (SPLIT) ld.s_stride *= G;
    stage_ws<CB>(wimg, Ws, M);

    // per-item state (the Ws fragments -- B operand: col g = lane&31, k = c -- come from LDS)
    float bias[IPW], m2[IPW], lsum[IPW];
    f32x16 Z[IPW][CB];
    const int item0 = wave * IPW;
#pragma unroll
    for (int k = 0; k < IPW; ++k) {
        const int id = item0 + k, gb = id % MB;
        const int g = 32 * gb + l31;
        bias[k] = bs[g] - c0;
        m2[k] = -INFINITY;
        lsum[k] = 0.f;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) Z[k][cb][r] = 0.f;
    }
    const bool active = item0 < H * MB;

    ld.fetch(N - t_begin * TN);
    ld.stage(0, sarr0, bt, base_temp);
    __syncthreads();
    int cur = 0;
    for (int t = t_begin; t < t_end; t += G) {
        const bool more = t + G < t_end;
        if (more) ld.fetch(N - (t + G) * TN);
        if (active) {
            const int nvalid = N - t * TN;  // >= 32 except on the last tile
#pragma unroll
            for (int hh = 0; hh < HPW; ++hh) {
                const int h = (item0 + hh * IPH) / MB;
                const int img = cur * buf_bytes + h * CB * (2 * IMG);  // (+ cb * 2 IMG: the head's channel blocks)
                const int sp = sarr0 + cur * sarr_bytes + 4 * (h * CB * TN);
                // rows of this lane's accumulator registers: n = (r&3) + 8(r>>2) + 4hf
                const int sq = sp + 16 * hf;  // s_n of those rows: four 16-B pieces, 32 B apart
#pragma unroll
                for (int kk = 0; kk < IPH; ++kk) {
                    const int k = hh * IPH + kk;
                    const int wi = wimg + ((item0 + k) % MB) * CB * (2 * IMG);
                    // Live ranges are kept short on purpose (fragments read where they are used, the
                    // logits not kept for the rare rescale): the one-item-per-wave variant has to fit
                    // 128 VGPRs to run four waves per SIMD.
                    auto logits = [&]() {
                        f32x16 Y;
#pragma unroll
                        for (int r = 0; r < 16; ++r) Y[r] = bias[k];
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb) {
                            Y = mfma3(read_rows(img + cb * (2 * IMG), l31, hf), read_rows(wi + cb * (2 * IMG), l31, hf), Y);
                            Y = mfma3(read_rows(img + cb * (2 * IMG), l31, 2 + hf), read_rows(wi + cb * (2 * IMG), l31, 2 + hf), Y);
                        }
                        return Y;
                    };
                    // p = exp2(Y s_n - m2): one fma + one v_exp per element.  The running maximum is a
                    // deferred REFERENCE: only when some exponent exceeds 2^20 (or on the first tile,
                    // m2 = -inf) is it raised and Z, l rescaled (rare; recomputes the logits).
                    float p[16];
                    float tmax = -INFINITY;
                    {
                        const f32x16 Y = logits();
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x4 s4 = *reinterpret_cast<const f32x4*>(smem + sq + 32 * i);
#pragma unroll
                            for (int q = 0; q < 4; ++q) p[4 * i + q] = fmaf(Y[4 * i + q], s4[q], -m2[k]);
                        }
                    }
                    if (nvalid < TN) {  // last tile: rows past the end take no part in the maximum either
                        asm volatile("" ::: "memory");  // (a cloud of one point must give weight 1, not 0/0)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if ((r & 3) + 8 * (r >> 2) + 4 * hf >= nvalid) p[r] = -INFINITY;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, p[r]);
                    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
                    if (__any(tmax > kThr2)) {
                        const f32x16 Y = logits();
                        float ys[16];
                        float zmax = -INFINITY;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x4 s4 = *reinterpret_cast<const f32x4*>(smem + sq + 32 * i);
#pragma unroll
                            for (int q = 0; q < 4; ++q) ys[4 * i + q] = Y[4 * i + q] * s4[q];
                        }
                        if (nvalid < TN) {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                if ((r & 3) + 8 * (r >> 2) + 4 * hf >= nvalid) ys[r] = -INFINITY;
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) zmax = fmaxf(zmax, ys[r]);
                        zmax = fmaxf(zmax, __shfl_xor(zmax, 32));
                        const float mn = fmaxf(m2[k], zmax);
                        const float a = __builtin_amdgcn_exp2f(m2[k] - mn);  // m2 = -inf -> 0
                        lsum[k] *= a;
                        if (hf == 0) *reinterpret_cast<float*>(smem + scratch + 4 * l31) = a;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x4 a4 = *reinterpret_cast<const f32x4*>(smem + scratch + 4 * (8 * i + 4 * hf));
#pragma unroll
                            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                                for (int q = 0; q < 4; ++q) Z[k][cb][4 * i + q] *= a4[q];
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        m2[k] = mn;
#pragma unroll
                        for (int r = 0; r < 16; ++r) p[r] = ys[r] - mn;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(p[r]);
                    if (nvalid < TN) {  // last tile only; the asm keeps it a branch (no per-element selects)
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if ((r & 3) + 8 * (r >> 2) + 4 * hf >= nvalid) p[r] = 0.f;
                    }
                    float ps = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) ps += p[r];
                    lsum[k] += ps;
                    // Z[g,c] += P^T X : accumulator tile reused as the A operand, one half at a time
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        float ph[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) ph[j] = p[8 * s + j];
                        const Frag pf = split8(ph);
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb) Z[k][cb] = mfma3(pf, read_cols(img + cb * (2 * IMG), s, lane), Z[k][cb]);
                    }
                }
            }
        }
        if (more) ld.stage((cur ^ 1) * buf_bytes, sarr0 + (cur ^ 1) * sarr_bytes, bt, base_temp);
        if (!PRIV) __syncthreads();  // (PRIV: every wave reads only what it wrote itself)
        cur ^= 1;
    }

    // record of this chunk: [B*H][ m[M] | l[M] | St[M][C] ]
    if (active) {
#pragma unroll
        for (int k = 0; k < IPW; ++k) {
            const int id = item0 + k, h = id / MB, gb = id % MB;
            float* rec = parts + ((size_t)blockIdx.x * BH + b * H + h) * (size_t)M_rec * (C + 2);
            const float l = lsum[k] + __shfl_xor(lsum[k], 32);
            if (hf == 0) {
                rec[g_off + 32 * gb + l31] = m2[k] * kLn2;
                rec[M_rec + g_off + 32 * gb + l31] = l;
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int g = g_off + 32 * gb + (r & 3) + 8 * (r >> 2) + 4 * hf;
                    rec[2 * M_rec + g * C + 32 * cb + l31] = Z[k][cb][r];
                }
        }
    }
}

// ----------------------------------------------------------------------------- de-slice
// HPW heads per wave; all MB column blocks of a head are summed into one output tile.
template <int LI, int HPW, int MB, bool SPLIT = false, int CB = 1>
__global__ __launch_bounds__(NTHR) void deslice_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ Ws, const float* __restrict__ bs,
    const float* __restrict__ wt, float bt, float base_temp, float c0,
    const float* __restrict__ col_max, const float* __restrict__ col_sum,
    const float* __restrict__ S2, int N, int H, int tiles_per_chunk, float* __restrict__ out,
    const float* __restrict__ s_in) {
    constexpr int M = 32 * MB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hf = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y;
    const int n_tiles = (N + TN - 1) / TN, G = gridDim.x;  // tiles dealt round-robin, see the slice kernel
    const int t_begin = blockIdx.x, t_end = n_tiles;
    // LDS (byte offsets): tile buffers | s_n [2][H CB][32] | bias [H][M] | m2 [H][M] | Ws images
    constexpr int C = kC * CB;
    const int Hv = H * CB;
    const int buf_bytes = Hv * 2 * IMG;
    const int sarr0 = 2 * buf_bytes, sarr_bytes = Hv * TN * 4;
    const int biasL = sarr0 + 2 * sarr_bytes;  // [H][M]  bs - c0
    const int m2L = biasL + H * M * 4;         // [H][M]  col_max * log2e
    const int wimg = m2L + H * M * 4;          // [MB][CB][hi|lo] images of Ws
    const float* xb = x + (size_t)b * N * H * C;
    float* ob = out + (size_t)b * N * H * C;
    static_assert(CB == 1 || !SPLIT, "dim_head 64: fp32 input (Ada-Temp in the loader)");

    for (int i = threadIdx.x; i < H * M; i += NTHR) {
        *reinterpret_cast<float*>(smem + biasL + 4 * i) = bs[i % M] - c0;
        *reinterpret_cast<float*>(smem + m2L + 4 * i) = col_max[(size_t)b * H * M + i] * kLog2e;
    }
    Loader<LI, NTHR, SPLIT, false, CB> ld;
    ld.init(wt, xb, Hv, t_begin * TN, SPLIT ? s_in + (size_t)b * N * H : nullptr);
    ld.tile_stride *= G;
    if (SPLIT) ld.s_stride *= G;
    stage_ws<CB>(wimg, Ws, M);

    // per-head constants: the S'/l fragments (A operand: row c = lane&31 of channel block cb, k = g in accumulator
    // order); the Ws fragments (A operand: row g = lane&31, k = c) come from LDS
    Frag sf[HPW][MB][2][CB];
    const int head0 = wave * HPW;
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh) {
        const int h = min(head0 + hh, H - 1);
        const float* s2h = S2 + ((size_t)b * H + h) * M * C;
        const float* csh = col_sum + ((size_t)b * H + h) * M;
#pragma unroll
        for (int gb = 0; gb < MB; ++gb) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int g = 32 * gb + 16 * s + 8 * (j >> 2) + 4 * hf + (j & 3);
                        v[j] = s2h[g * C + 32 * cb + l31] / csh[g];
                    }
                    sf[hh][gb][s][cb] = split8(v);
                }
            }
        }
    }

    ld.fetch(N - t_begin * TN);
    ld.stage(0, sarr0, bt, base_temp);
    __syncthreads();
    int cur = 0;
    for (int t = t_begin; t < t_end; t += G) {
        const bool more = t + G < t_end;
        if (more) ld.fetch(N - (t + G) * TN);
        const int nvalid = N - t * TN;
#pragma unroll
        for (int hh = 0; hh < HPW; ++hh) {
            const int h = head0 + hh;
            if (h < H) {
                const int img = cur * buf_bytes + h * CB * (2 * IMG);
                const float sn = *reinterpret_cast<const float*>(smem + sarr0 + cur * sarr_bytes + 4 * (h * CB * TN + l31));
                Frag x0[CB], x1[CB];
                f32x16 O[CB];
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    x0[cb] = read_rows(img + cb * (2 * IMG), l31, hf);
                    x1[cb] = read_rows(img + cb * (2 * IMG), l31, 2 + hf);
#pragma unroll
                    for (int r = 0; r < 16; ++r) O[cb][r] = 0.f;
                }
#pragma unroll
                for (int gb = 0; gb < MB; ++gb) {
                    const int bp = biasL + 4 * (h * M + 32 * gb + 4 * hf);
                    const int mp = m2L + 4 * (h * M + 32 * gb + 4 * hf);
                    f32x16 Y;
                    f32x4 m4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(smem + bp + 32 * i);
                        m4[i] = *reinterpret_cast<const f32x4*>(smem + mp + 32 * i);
#pragma unroll
                        for (int q = 0; q < 4; ++q) Y[4 * i + q] = b4[q];
                    }
                    const int wi = wimg + gb * CB * (2 * IMG);
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) {
                        Y = mfma3(read_rows(wi + cb * (2 * IMG), l31, hf), x0[cb], Y);  // Yt[g,n]: rows g in regs, point on the lane
                        Y = mfma3(read_rows(wi + cb * (2 * IMG), l31, 2 + hf), x1[cb], Y);
                    }
                    float p0[8], p1[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        p0[j] = __builtin_amdgcn_exp2f(fmaf(Y[j], sn, -m4[j >> 2][j & 3]));
                        p1[j] = __builtin_amdgcn_exp2f(fmaf(Y[8 + j], sn, -m4[2 + (j >> 2)][j & 3]));
                    }
                    const Frag pf0 = split8(p0), pf1 = split8(p1);
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) {
                        O[cb] = mfma3(sf[hh][gb][0][cb], pf0, O[cb]);  // Ot[c,n] += (S'/l)^T P
                        O[cb] = mfma3(sf[hh][gb][1][cb], pf1, O[cb]);
                    }
                }
                if (l31 < nvalid) {
                    float* orow = ob + ((size_t)(t * TN + l31) * H + h) * C + 4 * hf;
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x4 v = {O[cb][4 * i], O[cb][4 * i + 1], O[cb][4 * i + 2], O[cb][4 * i + 3]};
                            *reinterpret_cast<f32x4*>(orow + 32 * cb + 8 * i) = v;
                        }
                }
            }
        }
        if (more) ld.stage((cur ^ 1) * buf_bytes, sarr0 + (cur ^ 1) * sarr_bytes, bt, base_temp);
        __syncthreads();
        cur ^= 1;
    }
}

// ----------------------------------------------------------------------------- host side
struct MfmaPlan {
    int li, ipw, mb, hpw, tiles_per_chunk, parts, cb;
    int s_nw, s_li;  // slice pass: waves per workgroup and loader chunks per thread
    size_t lds_slice, lds_deslice;
};

// shapes the MFMA path covers: C = 32, M in {32, 64, 128}, even H <= 16; round 4: C = 64 (two 32-channel blocks per head)
// where every (head, slice block) item has its own wave and the de-slice's S'/l fragments fit the registers:
// H * M / 32 <= 8, M <= 64, H in {2, 4, 8} -- BASELINE config C2 (H = 4, M = 32) is inside
bool mfma_plan(int B, int N, int H, int C, int M, MfmaPlan* p) {
    if ((C != kC && C != 2 * kC) || (M != 32 && M != 64 && M != 128)) return false;
    const int cbk = C / kC;
    if (cbk == 2 && !((H == 2 || H == 4 || H == 8) && M <= 64 && H * (M / 32) <= 8)) return false;
    if (cbk == 1 && (H < 2 || H > 16 || (H & 1))) return false;
    const int mb = M / 32, items = H * mb;
    const int ipw = (items + NW - 1) / NW;
    if (ipw != 1 && ipw != 2 && ipw != 4) return false;
    const int iph = ipw < mb ? ipw : mb;
    if (mb % iph || ipw % iph) return false;
    const int hpw = (H + NW - 1) / NW;
    if (hpw != 1 && hpw != 2) return false;
    if (p) {
        p->li = (H * cbk + 1) / 2;
        p->cb = cbk;
        p->ipw = ipw;
        p->mb = mb;
        p->hpw = hpw;
        const int n_tiles = (N + TN - 1) / TN;
        int wgs = 256;  // workgroups per launch (one per CU: ~230 VGPRs); HAET_SLICE_CHUNKS overrides (tuning)
        if (const char* e = getenv("HAET_SLICE_CHUNKS")) wgs = atoi(e) > 0 ? atoi(e) : wgs;
        int target = wgs / (B < 1 ? 1 : B);
        if (target < 1) target = 1;
        p->parts = n_tiles < target ? n_tiles : target;
        p->tiles_per_chunk = (n_tiles + p->parts - 1) / p->parts;  // (informational: tiles per workgroup)
        const size_t bufs = (size_t)2 * H * cbk * 2 * IMG + (size_t)2 * H * cbk * TN * 4;
        const size_t wimg = (size_t)mb * cbk * 2 * IMG;
        p->s_nw = NW;  // (16 waves with one item each fit 128 VGPRs only with spills and ran 25 % slower)
        p->s_li = p->li;
        p->lds_slice = bufs + p->s_nw * 32 * 4 + wimg;
        p->lds_deslice = bufs + (size_t)2 * H * M * 4 + wimg;
    }
    return true;
}

template <typename K>
static int set_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return fail(HAET_E_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    return HAET_OK;
}

#define SLICE_LAUNCH(LI_, IPW_, MB_, NW_, SP_)                                                    \
    {                                                                                                 \
        auto kern = slice_mfma_kernel<LI_, IPW_, MB_, NW_, SP_>;                                      \
        if (int rc = set_lds(kern, p.lds_slice)) return rc;                                          \
        hipLaunchKernelGGL(kern, grid, dim3(NW_ * 64), p.lds_slice, st, x, Ws, bs, wt, bt, base_temp, \
                           c0, N, H, p.tiles_per_chunk, parts, s_in, M_rec, g_off);                  \
        return check_launch("slice_mfma_kernel");                                                    \
    }
#define SLICE_LAUNCH_P(LI_, IPW_, MB_, NW_)                                                        \
    {                                                                                                 \
        auto kern = slice_mfma_kernel<LI_, IPW_, MB_, NW_, true, true>;                               \
        if (int rc = set_lds(kern, p.lds_slice)) return rc;                                          \
        hipLaunchKernelGGL(kern, grid, dim3(NW_ * 64), p.lds_slice, st, x, Ws, bs, wt, bt, base_temp, \
                           c0, N, H, p.tiles_per_chunk, parts, s_in, M_rec, g_off);                  \
        return check_launch("slice_mfma_kernel");                                                    \
    }
#define SLICE_CASE(LI_, IPW_, MB_, NW_)                                                            \
    if (p.s_li == LI_ && p.ipw == IPW_ && p.mb == MB_ && p.s_nw == NW_) {                            \
        if (s_in) {                                                                                   \
            if constexpr (LI_ == 4 && NW_ == 8) {                                                     \
                if (!shared_load) SLICE_LAUNCH_P(LI_, IPW_, MB_, NW_)                                 \
            }                                                                                         \
            if constexpr (LI_ >= 4) SLICE_LAUNCH(LI_, IPW_, MB_, NW_, true)                           \
            return fail(HAET_E_ARG, "slice_mfma: split input needs H = 8 or 16");                     \
        }                                                                                             \
        SLICE_LAUNCH(LI_, IPW_, MB_, NW_, false)                                                      \
    }

static int slice_mfma_launch(const float* x, const float* Ws, const float* bs, const float* wt, float bt,
                             float base_temp, float c0, int B, int N, int H, int C, int M, float* parts,
                             hipStream_t st, const float* s_in, int M_rec, int g_off);

int slice_mfma_fwd(const float* x, const float* Ws, const float* bs, const float* wt, float bt,
                   float base_temp, float c0, int B, int N, int H, int C, int M, float* parts,
                   hipStream_t st, const float* s_in) {
    static const bool one_launch = getenv("HAET_SLICE_M128_ONE") != nullptr;  // A/B: the four-items-per-wave kernel
    if (M == 128 && !one_launch && mfma_plan(B, N, H, C, 64, nullptr)) {
        // two launches of the M = 64 kernel, each reading x once: 2 x 0.30 ms per 1M points against 1.4 ms for
        // the M = 128 variant (64 accumulator + 64 logit registers per wave: spills)
        for (int half = 0; half < 2; ++half) {
            int rc = slice_mfma_launch(x, Ws + (size_t)64 * half * kC, bs + 64 * half, wt, bt, base_temp, c0, B, N, H, C,
                                       64, parts, st, s_in, 128, 64 * half);
            if (rc != HAET_OK) return rc;
        }
        return HAET_OK;
    }
    if (M == 64 && H == 16 && C == kC && !one_launch) {
        // H = 16 at M = 64 is four items per wave too (69-71 spilled registers): two launches of its M = 32 variant
        for (int half = 0; half < 2; ++half) {
            int rc = slice_mfma_launch(x, Ws + (size_t)32 * half * kC, bs + 32 * half, wt, bt, base_temp, c0, B, N, H, C, 32, parts, st,
                                       s_in, 64, 32 * half);
            if (rc != HAET_OK) return rc;
        }
        return HAET_OK;
    }
    return slice_mfma_launch(x, Ws, bs, wt, bt, base_temp, c0, B, N, H, C, M, parts, st, s_in, M, 0);
}

static int slice_mfma_launch(const float* x, const float* Ws, const float* bs, const float* wt, float bt,
                             float base_temp, float c0, int B, int N, int H, int C, int M, float* parts,
                             hipStream_t st, const float* s_in, int M_rec, int g_off) {
    MfmaPlan p;
    if (!mfma_plan(B, N, H, C, M, &p)) return fail(HAET_E_ARG, "slice_mfma: unsupported shape");
    dim3 grid(p.parts, B);
    static const bool shared_load = getenv("HAET_SLICE_SHARED_LOAD") != nullptr;  // A/B: cooperative loader
    if (p.cb == 2) {  // dim_head 64: fp32 input only, one item per wave
        if (s_in) return fail(HAET_E_ARG, "slice_mfma: the split operand format needs dim_head 32");
#define SLICE_CASE64(LI_, MB_)                                                                                  \
    if (p.s_li == LI_ && p.ipw == 1 && p.mb == MB_) {                                                           \
        auto kern = slice_mfma_kernel<LI_, 1, MB_, NW, false, false, 2>;                                        \
        if (int rc = set_lds(kern, p.lds_slice)) return rc;                                                     \
        hipLaunchKernelGGL(kern, grid, dim3(NW * 64), p.lds_slice, st, x, Ws, bs, wt, bt, base_temp, c0, N, H,  \
                           p.tiles_per_chunk, parts, s_in, M_rec, g_off);                                      \
        return check_launch("slice_mfma_kernel<dim_head 64>");                                                  \
    }
        SLICE_CASE64(4, 1)  // H = 4, M = 32 (C2)
        SLICE_CASE64(4, 2)  // H = 4, M = 64
        SLICE_CASE64(8, 1)  // H = 8, M = 32
        SLICE_CASE64(2, 1)  // H = 2
        SLICE_CASE64(2, 2)
#undef SLICE_CASE64
        return fail(HAET_E_ARG, "slice_mfma: no dim_head-64 instantiation for H=%d M=%d", H, M);
    }
    SLICE_CASE(4, 2, 2, 8)   // H=8,  M=64  (C3/C4)
    SLICE_CASE(4, 1, 1, 8)   // H=8,  M=32
    SLICE_CASE(4, 4, 4, 8)   // H=8,  M=128 (C5)
    SLICE_CASE(2, 1, 2, 8)   // H=4,  M=64
    SLICE_CASE(2, 1, 1, 8)   // H=4,  M=32
    SLICE_CASE(2, 2, 4, 8)   // H=4,  M=128
    SLICE_CASE(1, 1, 1, 8)   // H=2
    SLICE_CASE(1, 1, 2, 8)
    SLICE_CASE(1, 1, 4, 8)
    SLICE_CASE(8, 2, 1, 8)   // H=16, M=32
    SLICE_CASE(8, 4, 2, 8)   // H=16, M=64
    return fail(HAET_E_ARG, "slice_mfma: no instantiation for H=%d M=%d", H, M);
}

#define DESLICE_LAUNCH(LI_, HPW_, MB_, SP_)                                                       \
    {                                                                                               \
        auto kern = deslice_mfma_kernel<LI_, HPW_, MB_, SP_>;                                       \
        if (int rc = set_lds(kern, p.lds_deslice)) return rc;                                      \
        hipLaunchKernelGGL(kern, grid, dim3(NTHR), p.lds_deslice, st, x, Ws, bs, wt, bt, base_temp, \
                           c0, cm, cs, S2, N, H, p.tiles_per_chunk, out, s_in);                    \
        return check_launch("deslice_mfma_kernel");                                                \
    }
#define DESLICE_CASE(LI_, HPW_, MB_)                                                              \
    if (p.li == LI_ && p.hpw == HPW_ && p.mb == MB_) {                                             \
        if (s_in) {                                                                                 \
            if constexpr (LI_ >= 4) DESLICE_LAUNCH(LI_, HPW_, MB_, true)                            \
            return fail(HAET_E_ARG, "deslice_mfma: split input needs H = 8 or 16");                 \
        }                                                                                           \
        DESLICE_LAUNCH(LI_, HPW_, MB_, false)                                                       \
    }

int deslice_mfma_fwd(const float* x, const float* Ws, const float* bs, const float* wt, float bt,
                     float base_temp, float c0, const float* cm, const float* cs, const float* S2,
                     int B, int N, int H, int C, int M, float* out, hipStream_t st, const float* s_in) {
    MfmaPlan p;
    if (!mfma_plan(B, N, H, C, M, &p)) return fail(HAET_E_ARG, "deslice_mfma: unsupported shape");
    dim3 grid(p.parts, B);
    if (p.cb == 2) {
        if (s_in) return fail(HAET_E_ARG, "deslice_mfma: the split operand format needs dim_head 32");
#define DESLICE_CASE64(LI_, MB_)                                                                                     \
    if (p.li == LI_ && p.mb == MB_) {                                                                                \
        auto kern = deslice_mfma_kernel<LI_, 1, MB_, false, 2>;                                                      \
        if (int rc = set_lds(kern, p.lds_deslice)) return rc;                                                        \
        hipLaunchKernelGGL(kern, grid, dim3(NTHR), p.lds_deslice, st, x, Ws, bs, wt, bt, base_temp, c0, cm, cs, S2,  \
                           N, H, p.tiles_per_chunk, out, s_in);                                                     \
        return check_launch("deslice_mfma_kernel<dim_head 64>");                                                     \
    }
        DESLICE_CASE64(4, 1)
        DESLICE_CASE64(4, 2)
        DESLICE_CASE64(8, 1)
        DESLICE_CASE64(2, 1)
        DESLICE_CASE64(2, 2)
#undef DESLICE_CASE64
        return fail(HAET_E_ARG, "deslice_mfma: no dim_head-64 instantiation for H=%d M=%d", H, M);
    }
    DESLICE_CASE(4, 1, 2)
    DESLICE_CASE(4, 1, 1)
    DESLICE_CASE(4, 1, 4)
    DESLICE_CASE(2, 1, 2)
    DESLICE_CASE(2, 1, 1)
    DESLICE_CASE(2, 1, 4)
    DESLICE_CASE(1, 1, 1)
    DESLICE_CASE(1, 1, 2)
    DESLICE_CASE(1, 1, 4)
    DESLICE_CASE(8, 2, 1)
    DESLICE_CASE(8, 2, 2)
    return fail(HAET_E_ARG, "deslice_mfma: no instantiation for H=%d M=%d", H, M);
}

}  // namespace haet
