// De-slice fused with the to_out Linear (models/PhysicsAttention/IrregularMesh.py:221-235):
//
//     out[n, :] = W_out . concat_h( sum_g w[n,h,g] S'[h,g,:] ) + b_out
//
// in ONE pass over the points.  Unfused, the de-slice pass writes out_x [N, H*C] (1 KB per point at H*C = 256) and
// the to_out GEMM reads it back: 2 KB per point of the path's 7 KB for a tensor that is consumed once, row by row,
// by the very next kernel.  Here out_x never leaves the CU: the de-slice accumulators become -- through a 32 KB
// LDS image -- the B operand of the to_out product.
//
// Shape: H = 8 heads x C = 32 channels -> D = 256 outputs (the C3 / C5 widths), M in {32, 64}, input in the fused
// operand format of haet_project_split_fwd (16-B cells {4 x bf16 hi, 4 x bf16 lo} + s = log2(e)/tau).
// One persistent workgroup per CU, 8 waves; wave w is BOTH head w of the de-slice and output columns 32w..32w+31 of
// the GEMM:
//   [A] per 32-point tile, wave h (points on the lanes, as deslice_mfma_kernel):
//         Yt[g,n] = Ws[g,:] . X_h[n,:] + b_g ;  pt = exp2(Yt s_n - m2_g) ;  Ot[c,n] = sum_g (S'[g,c]/l_g) pt[g,n]
//       X_h is loaded wave-privately (head h's 32 x 128 B block of the tile: no barrier for it) and de-sliced IN PLACE:
//       Ot overwrites it as the head's O image [hi|lo][32 points][32 channels] (split bf16).
//   -- ONE workgroup barrier per tile; two buffers of 8 x {hi, lo} images --
//   [B] wave w: out^T[o,n] = b_o + sum_{h,c} W_out[o, 32h+c] O_h[n,c]  (A = W_out rows o = 32w.., kept in 128 VGPRs
//       for the lifetime of the kernel like gemm_ws_kernel's; B = rows of the O image) as four 16 x 16 tiles on
//       v_mfma_f32_16x16x32_bf16 (96 MFMAs of 16 cycles; same matrix-pipe cycles as 48 of the 32x32x16 shape, but the
//       chip holds a higher clock under it: 0.594 -> 0.578 ms, same-box A/B), then four 16-B stores per lane.
// Round 3: [A] of tile t + 1 and [B] of tile t are independent, so the two waves of a SIMD (w, w + 4) run them half a
// tile apart -- waves 4-7 take the barrier BEFORE their [B] -- and inside [A] the instruction order is pinned by hand
// (vector work behind the MFMA it shadows, LDS reads a stage early); see the loop.  Measured with the stamped build
// (tools/deslice_out_stamps.py), per wave and tile: [A] ~2.5k cycles alone / ~3.7k beside the partner, [B] ~2.1k / ~2.5k,
// against 72 x 32 = 2.3k cycles of matrix pipe: the kernel is bound by the three issue resources of a SIMD together
// (matrix pipe ~58 % busy, vector port and LDS ~50 %), not by HBM (the same loop without arithmetic runs in 0.38 ms).
// Algorithmic HBM traffic: 1 KB read + 1 KB written per point (+ 32 B of s), against 4 KB for the two kernels.
#include <cstdlib>

#include "slice_mfma_dev.h"

namespace haet {

// The x / O images of THIS kernel use the chunk swizzle (-(row >> 2)) & 3 instead of slice_mfma_dev.h's (row >> 2) & 3 (round 4).
// Both are conflict-free for the 32 x 32 x 16 fragment reads of phase [A] (lane = row, the ds_read_b128 lane groups {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31} see rows with row >> 2 = {0, 3, 5, 6} / {1, 2, 4, 7}: four different xor values either way), but
// phase [B] has read the O image as 16 x 16 x 32 fragments since round 3 (lane = (row % 16, chunk = lane / 16): a lane group
// sees rows 0-3 and 12-15 with chunk c and rows 4-11 with chunk c + 1), where (row >> 2) & 3 puts rows r and r + 4 on the same
// banks: SQ_LDS_BANK_CONFLICT 40 M cycles per launch (15 % of the kernel's cycles, profiles/r04_pmc.json).  Rows 8 apart still
// differ in bit 1 of the xor, so the loader's two-offset scheme holds.  (The Ws images keep the shared layout: stage_ws.)
__device__ __forceinline__ int do_img_off(int row, int chunk) { return row * 64 + ((chunk ^ ((0 - (row >> 2)) & 3)) << 4); }
__device__ __forceinline__ Frag do_read_rows(int img, int row, int chunk) {
    Frag f;
    const int off = img + do_img_off(row, chunk);
    f.hi = *reinterpret_cast<const u32x4*>(smem + off);
    f.lo = *reinterpret_cast<const u32x4*>(smem + off + IMG);
    return f;
}

constexpr int kDO = 256;  // outputs = H * C = 8 * 32

#ifdef HAET_STAMPS
// Diagnostic build only (make EXTRA=-DHAET_STAMPS; tools/deslice_out_stamps.py): per wave of workgroup 0 the s_memtime
// cycles spent in each part of the loop, summed over its tiles in scalar registers (no memory operation inside the
// loop) and written once at the end.  Never quote this build's run time.
__device__ unsigned long long haet_do_stamps[8 * 8];
#define DO_T0() unsigned long long st_prev = __builtin_amdgcn_s_memtime(), st_acc[6] = {0, 0, 0, 0, 0, 0}
#define DO_T(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_prev; st_prev = now_; } while (0)
#else
#define DO_T0() do { } while (0)
#define DO_T(i) do { } while (0)
#endif

template <int MB, bool RES>
__global__ __launch_bounds__(512) void deslice_out_kernel(
    const float* __restrict__ x, const float* __restrict__ s_in, const float* __restrict__ Ws,
    const float* __restrict__ bs, float c0, const float* __restrict__ col_max, const float* __restrict__ col_sum,
    const float* __restrict__ S2, const unsigned short* __restrict__ Whi, const unsigned short* __restrict__ Wlo,
    const float* __restrict__ bias_out, const float* __restrict__ residual, int N, float* __restrict__ out, int dbg_arg) {
#ifdef HAET_STAMPS
    const int dbg = dbg_arg;
#else
    constexpr int dbg = 0;  // (the ablation switches exist in the diagnostic build only: a branch per head otherwise)
#endif
    // dbg (HAET_DO_DBG, stamped build only): 1 = no de-slice MFMAs, 2 = no GEMM MFMAs, 4 = no stagger, 8 = no s_setprio,
    // 16 = waves 4-7 idle (standalone phase times), 32 = priority on [B], 64 = waves 4-7 first -- wave-uniform
    // branches around matrix work only (none around a memory operation), results are wrong by construction
    constexpr int M = 32 * MB, H = 8;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), hf = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y;
    const int n_tiles = (N + TN - 1) / TN, G = gridDim.x;
    // LDS (byte offsets): x images [H][hi|lo] | s_n [H][32] | bias [H][M] | m2 [H][M] | out bias [256] | Ws images | O images [2][H][hi|lo]
    constexpr int XB = H * 2 * IMG;
    constexpr int sarr = XB, biasL = sarr + H * TN * 4, m2L = biasL + H * M * 4, boL = m2L + H * M * 4;
    constexpr int wimg = boL + kDO * 4, oimg0 = wimg + MB * 2 * IMG;
    constexpr int sfL = oimg0 + XB;            // S'/l fragments [wave][gb][half][hi|lo][64 lanes] x 16 B
    constexpr int SFW = MB * 2 * 2 * 1024;     // bytes per wave
    const float* xb = x + (size_t)b * N * H * kC;
    const float* sb = s_in + (size_t)b * N * H;
    float* ob = out + (size_t)b * N * kDO;
    const float* rb = residual ? residual + (size_t)b * N * kDO : nullptr;  // out += residual (the block's `+ fx`)

    for (int i = threadIdx.x; i < H * M; i += 512) {
        *reinterpret_cast<float*>(smem + biasL + 4 * i) = bs[i % M] - c0;
        *reinterpret_cast<float*>(smem + m2L + 4 * i) = col_max[(size_t)b * H * M + i] * kLog2e;
    }
    for (int i = threadIdx.x; i < kDO; i += 512) *reinterpret_cast<float*>(smem + boL + 4 * i) = bias_out ? bias_out[i] : 0.f;
    stage_ws(wimg, Ws, M);

    // stationary A fragments of the GEMM ([B] runs v_mfma_f32_16x16x32_bf16: the chip holds a higher clock under this
    // shape than under 32x32x16 at equal cycles per flop, MI355X_MICROARCH.md 'DVFS give-back' (7)): rows
    // o = 32 wave + 16 io + lane % 16 of W_out, k-step ks = head (32 wide): k = 32 ks + 8 (lane / 16) ..
    const int l15 = lane & 15, kg = lane >> 4;
    u32x4 wh[16], wl[16];
    {
        // (from the tile-ordered copies behind the planes -- gemm_x3.hip split_bf16_tiled_kernel: piece (row, k-chunk
        //  4 ks + kg) at [row / 16][chunk][row % 16], so a wave instruction reads ONE run of 1 KB, not 16 of 64 B)
#pragma unroll
        for (int io = 0; io < 2; ++io) {
            const size_t piece0 = (((size_t)(2 * wave + io) * (kDO / 8) + kg) << 4) + l15;
            const unsigned short* th = Whi + (size_t)kDO * kDO + piece0 * 8;
            const unsigned short* tl = Wlo + (size_t)kDO * kDO + piece0 * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                wh[2 * ks + io] = *reinterpret_cast<const u32x4*>(th + (size_t)ks * 4 * 16 * 8);
                wl[2 * ks + io] = *reinterpret_cast<const u32x4*>(tl + (size_t)ks * 4 * 16 * 8);
            }
        }
    }
    // S'/l fragments of this wave's head (A operand of the de-slice product: row c = l31, k = g in accumulator order)
    // (kept in LDS, not in registers: with 128 registers of W_out the wave has none to spare, and a spilled
    // register costs far more than the two ds_read_b128 per use)
    {
        const float* s2h = S2 + ((size_t)b * H + wave) * M * kC;
        const float* csh = col_sum + ((size_t)b * H + wave) * M;
#pragma unroll
        for (int gb = 0; gb < MB; ++gb) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int g = 32 * gb + 16 * s + 8 * (j >> 2) + 4 * hf + (j & 3);
                    v[j] = s2h[g * kC + l31] / csh[g];
                }
                const Frag f = split8(v);
                const int off = sfL + wave * SFW + (gb * 2 + s) * 2048 + lane * 16;
                *reinterpret_cast<u32x4*>(smem + off) = f.hi;
                *reinterpret_cast<u32x4*>(smem + off + 1024) = f.lo;
            }
        }
    }

    // wave-private loader: head `wave`'s 32 x 128 B block of a tile = 256 cells of 16 B, four per lane
    //   cell e = lane + 64 i: row r = e >> 3 (= (lane >> 3) + 8 i), 16-B cell q = lane & 7 of the head's row
    const int lr = lane >> 3, lq = lane & 7;
    const int ximg = wave * (2 * IMG);
    // LDS byte offset of the hi half of cell i: rows 8 apart flip bit 1 of the chunk swizzle (-(row >> 2) & 3)
    const int xoff0 = ximg + do_img_off(lr, lq >> 1) + ((lq & 1) << 3), xoff1 = ximg + do_img_off(lr + 8, lq >> 1) + ((lq & 1) << 3) - 512;
    auto xoff = [&](int i) { return ((i & 1) ? xoff1 : xoff0) + 512 * i; };
    f32x4 xr[4];
    float sv = 1.f;
    // Every global load of the loop is UNCONDITIONAL and goes through a buffer descriptor over the tile's VALID rows
    // (tiles clamped to this workgroup's last one: a ghost fetch re-reads it): a load under a branch -- even a
    // wave-uniform one -- makes hipcc's waitcnt pass give up counting and wait vmcnt(0) where the cells are staged, i.e.
    // for the previous tile's output STORES as well (loads and stores share one in-order counter).  Rows past the end of
    // the cloud are outside the descriptor's range and read zeros; their results are never stored.  The descriptor is
    // scalar work per tile; the lane's offset never changes (no 64-bit vector address arithmetic in the loop).
    const int fvoff = lr * (H * kC * 4) + (wave * kC + 4 * lq) * 4;
    const int svoff = ((lane & 31) * H + wave) * 4;
    auto fetch = [&](int t) {
        const int rows = min(N - t * TN, TN);
        __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(xb)) + (size_t)t * TN * (H * kC * 4), 0, rows * (H * kC * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            xr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, fvoff, i * 8 * (H * kC * 4), 0));
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(sb)) + (size_t)t * TN * (H * 4), 0, rows * (H * 4), 0x00020000);
        sv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, svoff, 0, 0));
    };
    // L2 touch two tiles ahead: one dword per 128-B line of this wave's block of a tile (32 rows x 128 B: one line per
    // lane 0-31, the second half-wave the same lines), result folded into a word nothing reads.  The real 16-B loads of
    // that tile, issued one iteration later, then find their lines in L2.
    unsigned touch_acc = 0, touch_pend = 0;
    const int tvoff = (lane & 31) * (H * kC * 4) + wave * (kC * 4);
    auto touch = [&](int t) {
        touch_acc ^= touch_pend;
        const int rows = min(N - t * TN, TN);
        __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(xb)) + (size_t)t * TN * (H * kC * 4), 0, rows * (H * kC * 4), 0x00020000);
        touch_pend = __builtin_amdgcn_raw_buffer_load_b32(rx, tvoff, 0, 0);
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 w = __builtin_bit_cast(u32x4, xr[i]);
            const u32x2 hi = {w[0], w[1]}, lo = {w[2], w[3]};
            *reinterpret_cast<u32x2*>(smem + buf + xoff(i)) = hi;
            *reinterpret_cast<u32x2*>(smem + buf + xoff(i) + IMG) = lo;
        }
        *reinterpret_cast<float*>(smem + sarr + 4 * (wave * TN + (lane & 31))) = sv;  // (both half-waves: same value)
    };

    const int t_begin = blockIdx.x;
    if (t_begin >= n_tiles) return;
    const int t_last = t_begin + ((n_tiles - 1 - t_begin) / G) * G;  // this workgroup's last tile

    // ---- [A] de-slice of head `wave`, IN PLACE: reads its x block at `hb` (this wave's 2 images of the buffer),
    //      leaves Ot[c, n] there as the O image of the head (lane = point n, registers = channels).
    // The instruction ORDER is pinned (sched_barrier): a wave issues in order, so a vector instruction only runs in the
    // shadow of an MFMA if it stands behind it in the stream, and an LDS read only hides its latency if it is issued a
    // stage early.  Stages s = (column block, half) = 0 .. 2 MB - 1:
    //   V(s): pt = exp2(Y s_n - m2) and its hi / lo split (40 vector instructions)      M(s): O += (S'/l)_s . pt  (3 MFMAs)
    //   Y(0) (6 MFMAs), V(0), then M(s) with V(s + 1) woven between its MFMAs; Y(1) (M = 64) goes between M(0) and M(1),
    //   when the first column block's logits are dead (16 registers for Y, not 32).  The three products of a stage run
    //   lo.hi, hi.lo, hi.hi: the lo plane of S'/l is free after the first and reloaded for the next stage a stage early,
    //   the hi plane after the last, ~150 cycles before its next use; m2 is reloaded right after the exponentials.
    auto phaseA = [&](int hb) {
        if (!(dbg & 8)) __builtin_amdgcn_s_setprio(3);  // the dependency chain goes first; the partner's [B] MFMAs fill its gaps
        const float sn = *reinterpret_cast<const float*>(smem + sarr + 4 * (wave * TN + l31));
        constexpr int NS = 2 * MB;
        f32x16 Y;
        auto bias_init = [&](int gb) {
            const int bp = biasL + 4 * (wave * M + 32 * gb + 4 * hf);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(smem + bp + 32 * i);
#pragma unroll
                for (int q = 0; q < 4; ++q) Y[4 * i + q] = b4[q];
            }
        };
        // m2 of stage s: registers 8 half + j <-> slices 32 gb + 16 half + 8 (j >> 2) + 4 hf + (j & 3)
        f32x4 ma, mb;
        auto load_m = [&](int s_) {
            const int mp = m2L + 4 * (wave * M + 32 * (s_ >> 1) + 4 * hf) + 64 * (s_ & 1);
            ma = *reinterpret_cast<const f32x4*>(smem + mp);
            mb = *reinterpret_cast<const f32x4*>(smem + mp + 32);
        };
        u32x4 sfh, sfl;  // S'/l fragment of the running stage
        const int sfo = sfL + wave * SFW + lane * 16;
        load_m(0);
        sfh = *reinterpret_cast<const u32x4*>(smem + sfo);
        sfl = *reinterpret_cast<const u32x4*>(smem + sfo + 1024);
        {
            const Frag x0 = do_read_rows(hb, l31, hf), x1 = do_read_rows(hb, l31, 2 + hf);
            bias_init(0);
            if (!(dbg & 1)) {
                Y = mfma3(read_rows(wimg, l31, hf), x0, Y);
                Y = mfma3(read_rows(wimg, l31, 2 + hf), x1, Y);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 O;
#pragma unroll
        for (int r = 0; r < 16; ++r) O[r] = 0.f;
        float pv[8];
        Frag pf, pn;  // operand of the running M(s) / being built by V(s + 1)
        // V(0) alone
#pragma unroll
        for (int j = 0; j < 4; j += 2) {  // (plain v_fma_f32: the packed v_pk_fma_f32 measured 1.4 % slower)
            const f32x2 ya = {Y[j], Y[j + 1]}, yc = {Y[4 + j], Y[5 + j]};
            const f32x2 m0 = {ma[j], ma[j + 1]}, m1 = {mb[j], mb[j + 1]};
            const f32x2 ea = {fmaf(ya[0], sn, -m0[0]), fmaf(ya[1], sn, -m0[1])}, ec = {fmaf(yc[0], sn, -m1[0]), fmaf(yc[1], sn, -m1[1])};
            pv[j] = __builtin_amdgcn_exp2f(ea[0]), pv[j + 1] = __builtin_amdgcn_exp2f(ea[1]);
            pv[4 + j] = __builtin_amdgcn_exp2f(ec[0]), pv[5 + j] = __builtin_amdgcn_exp2f(ec[1]);
        }
        load_m(1);
        pf = split8(pv);
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const bool more = s_ + 1 < NS;
            const int yb = 8 * ((s_ + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (!(dbg & 1))
                O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, sfl), __builtin_bit_cast(bf16x8, pf.hi), O, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                sfl = *reinterpret_cast<const u32x4*>(smem + sfo + (s_ + 1) * 2048 + 1024);  // next stage's lo plane
#pragma unroll
                for (int j = 0; j < 4; j += 2) {  // (plain v_fma_f32: the packed v_pk_fma_f32 measured 1.4 % slower)
                    const f32x2 ya = {Y[yb + j], Y[yb + j + 1]}, yc = {Y[yb + 4 + j], Y[yb + 5 + j]};
                    const f32x2 m0 = {ma[j], ma[j + 1]}, m1 = {mb[j], mb[j + 1]};
                    const f32x2 ea = {fmaf(ya[0], sn, -m0[0]), fmaf(ya[1], sn, -m0[1])}, ec = {fmaf(yc[0], sn, -m1[0]), fmaf(yc[1], sn, -m1[1])};
                    pv[j] = __builtin_amdgcn_exp2f(ea[0]), pv[j + 1] = __builtin_amdgcn_exp2f(ea[1]);
                    pv[4 + j] = __builtin_amdgcn_exp2f(ec[0]), pv[5 + j] = __builtin_amdgcn_exp2f(ec[1]);
                }
                if (s_ + 2 < NS) load_m(s_ + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(dbg & 1))
                O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, sfh), __builtin_bit_cast(bf16x8, pf.lo), O, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    unsigned h, l;
                    split2(pv[2 * i], pv[2 * i + 1], h, l);
                    pn.hi[i] = h, pn.lo[i] = l;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(dbg & 1))
                O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, sfh), __builtin_bit_cast(bf16x8, pf.hi), O, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
                sfh = *reinterpret_cast<const u32x4*>(smem + sfo + (s_ + 1) * 2048);  // (into the registers the MFMA above has just read)
#pragma unroll
                for (int i = 2; i < 4; ++i) {
                    unsigned h, l;
                    split2(pv[2 * i], pv[2 * i + 1], h, l);
                    pn.hi[i] = h, pn.lo[i] = l;
                }
                pf = pn;
            }
            if (MB == 2 && s_ == 0) {  // Y(1): Y(0) is dead (one exposed LDS latency per tile; the partner wave fills it)
                __builtin_amdgcn_sched_barrier(0);
                const Frag x0b = do_read_rows(hb, l31, hf), w0b = read_rows(wimg + 2 * IMG, l31, hf);
                const Frag x1b = do_read_rows(hb, l31, 2 + hf), w1b = read_rows(wimg + 2 * IMG, l31, 2 + hf);
                bias_init(1);
                if (!(dbg & 1)) {
                    Y = mfma3(w0b, x0b, Y);
                    Y = mfma3(w1b, x1b, Y);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // O image of this head: row = point l31, channels c = 8 i + 4 hf .. + 3 from registers 4i .. 4i + 3
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned h01, l01, h23, l23;
            split2(O[4 * i], O[4 * i + 1], h01, l01);
            split2(O[4 * i + 2], O[4 * i + 3], h23, l23);
            const u32x2 hi = {h01, h23};
            const u32x2 lo = {l01, l23};
            // (= hb + do_img_off(l31, i) + 8 hf: hb is a multiple of 64, so the swizzle's xor on bits 4-5 commutes with the add)
            const int off = (hb + do_img_off(l31, 0) + (hf << 3)) ^ (i << 4);
            *reinterpret_cast<u32x2*>(smem + off) = hi;
            *reinterpret_cast<u32x2*>(smem + off + IMG) = lo;
        }
        if (!(dbg & 8)) __builtin_amdgcn_s_setprio(0);
    };
    // ---- [B] to_out of tile t from the O images at `ob_` (all heads): out^T[o, n] for o = 32 wave + (r&3) + 8 (r>>2) + 4 hf, n = l31
    //      (the next head's four fragment reads are issued before this head's six MFMAs: ~190 cycles of cover)
    auto phaseB = [&](int ob_, int t, bool real) {
        const int nvalid = real ? N - t * TN : 0;  // (a ghost turn: a descriptor of zero rows drops the stores)
        // four 16 x 16 tiles: acc[io][jn][r] = out^T[o = 32 wave + 16 io + 4 kg + r][n = 16 jn + lane % 16]
        f32x4 acc[2][2];
#pragma unroll
        for (int io = 0; io < 2; ++io) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(smem + boL + 4 * (32 * wave + 16 * io + 4 * kg));
            acc[io][0] = b4, acc[io][1] = b4;
        }
        // residual rows of this lane's points (the block's `Attn(ln_1(fx)) + fx`, HAETransolver_Irregular_Mesh.py:125)
        f32x4 res[2][2];
        // B operand of k-step h (= head h): column n = 16 jn + lane % 16, k = 8 kg .. 8 kg + 7 of the head's 32 channels
        Frag oc0 = do_read_rows(ob_, l15, kg), oc1 = do_read_rows(ob_, 16 + l15, kg);
#pragma unroll
        for (int h = 0; h < H; ++h) {
            Frag on0, on1;
            if (h + 1 < H) {
                on0 = do_read_rows(ob_ + (h + 1) * (2 * IMG), l15, kg);
                on1 = do_read_rows(ob_ + (h + 1) * (2 * IMG), 16 + l15, kg);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(dbg & 2)) {
#pragma unroll
                for (int io = 0; io < 2; ++io) {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, wh[2 * h + io]), al = __builtin_bit_cast(bf16x8, wl[2 * h + io]);
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) {
                        const Frag& of = jn ? oc1 : oc0;
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, of.hi), bl = __builtin_bit_cast(bf16x8, of.lo);
                        acc[io][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[io][jn], 0, 0, 0);
                        acc[io][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[io][jn], 0, 0, 0);
                        acc[io][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[io][jn], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (h + 1 < H) oc0 = on0, oc1 = on1;
            if (RES && h == 1) {  // requested behind the first MFMAs (registers), added after the last (rows past the end: clamped, never stored)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    const float* rrow = rb + (size_t)min((real ? t : t_begin) * TN + 16 * jn + l15, N - 1) * kDO + 32 * wave + 4 * kg;
#pragma unroll
                    for (int io = 0; io < 2; ++io) res[io][jn] = *reinterpret_cast<const f32x4*>(rrow + 16 * io);
                }
            }
        }
        // Stores through a buffer descriptor over THIS tile's valid rows: rows past the end of the cloud fall outside
        // its range and the hardware drops them -- no branch around the stores (a conditional store, like a
        // conditional load, costs the loop its counted waits), no per-lane address clamping.
        const int rows = nvalid < TN ? nvalid : TN;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char*>(ob) + (size_t)(real ? t : t_begin) * TN * (kDO * 4), 0, rows * (kDO * 4), 0x00020000);
        const int voff = l15 * (kDO * 4) + (32 * wave + 4 * kg) * 4;
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int io = 0; io < 2; ++io) {
                f32x4 v = acc[io][jn];
                if (RES) v += res[io][jn];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, voff, 16 * jn * (kDO * 4) + 64 * io, 0);
            }
    };

    // ---- the loop.  Two buffers of 8 x {hi, lo} images and ONE workgroup barrier per tile.  A turn of the loop:
    //   stage   the cells of tile t' (the one after t) leave their registers for this wave's own 4 KB of buffer `nxt`
    //   [A]     de-slices them in place                      fetch   the loads of the tile after t' (the cell registers are
    //           dead during [A], the phase with the highest register pressure; the loads -- L2 hits thanks to the touch two
    //           tiles ahead -- are in flight across [B] and the barrier)
    //   [B]     waves 0-3: of tile t  from buffer `cur`, then the barrier     (O(t') complete, every wave has read O(t))
    //           waves 4-7: the barrier first, then [B] of tile t' from buffer `nxt`
    // i.e. the two waves of a SIMD (w, w + 4) run the same code half a tile apart: one is in its matrix-dense phase
    // (48 MFMAs) while the other is in its vector-dense one (MI355X_MICROARCH.md, co-residence item 9).  Only the barrier
    // stands under a (wave-uniform) branch; every load and store is unconditional, ghosts included: the first turn's [B]
    // of waves 0-3 and the last turn's [A] (all waves) and [B] (waves 4-7) work on a tile that does not exist -- their
    // loads are clamped to the last tile, their stores go through a descriptor of zero rows.
    fetch(t_begin);
    touch(min(t_begin + G, t_last));
    {
        // The loop's steady state has four output stores in flight behind these loads.  hipcc's waitcnt pass merges the
        // loop-entry state with the back edge's conservatively: without the same four stores pending at the entry it
        // waits, every iteration, for the previous tile's stores to drain before staging the cells.  So the prologue
        // issues them too: zeros to this workgroup's first tile, overwritten by its [B].
        const int rows0 = N - t_begin * TN < TN ? N - t_begin * TN : TN;
        __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char*>(ob) + (size_t)t_begin * TN * (kDO * 4), 0, rows0 * (kDO * 4), 0x00020000);
        const int voff = l15 * (kDO * 4) + (32 * wave + 4 * kg) * 4;
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(z, r0, voff, 16 * (i >> 1) * (kDO * 4) + 64 * (i & 1), 0);
    }
    __syncthreads();  // constants (bias, m2, Ws images) staged by all threads
    int cur = 0, nxt = oimg0;
    const bool late = wave >= 4 && !(dbg & 4);  // [B] behind the barrier
    if ((dbg & 64) && wave >= 4) __builtin_amdgcn_s_setprio(2);  // (diagnostic: the younger wave of each SIMD first)
    DO_T0();
    for (int t = t_begin - G; t <= t_last; t += G) {  // t: the tile of waves 0-3's [B] (first turn: none)
        const int ta = t + G;                         // tile of [A] and of waves 4-7's [B] (last turn: none)
        const bool idle = (dbg & 16) && wave >= 4;  // (diagnostic: the SIMD partner does nothing -- standalone phase times)
        if (!(dbg & 8)) __builtin_amdgcn_s_setprio(3);
        if (!idle) stage(nxt);
        DO_T(0);
        if (!idle) {
            fetch(min(ta + G, t_last));
            touch(min(ta + 2 * G, t_last));
        }
        DO_T(3);
        if (!idle) phaseA(nxt + ximg);
        DO_T(1);
        if (late) __syncthreads();
        const int tb = late ? ta : t;
        if (dbg & 32) __builtin_amdgcn_s_setprio(3);
        if (!idle) phaseB(late ? nxt : cur, tb, tb >= t_begin && tb <= t_last);
        if (dbg & 32) __builtin_amdgcn_s_setprio(0);
        DO_T(2);
        if (!late) __syncthreads();
        DO_T(4);
        const int sw = cur;
        cur = nxt, nxt = sw;
    }
#ifdef HAET_STAMPS
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
        for (int i = 0; i < 5; ++i) haet_do_stamps[wave * 8 + i] = st_acc[i];
        haet_do_stamps[wave * 8 + 5] = (n_tiles - 1 - t_begin) / G + 1;
    }
#endif
    touch_acc ^= touch_pend;
    asm volatile("" ::"v"(touch_acc));
}

static int do_dbg() {  // HAET_DO_DBG: ablation switches of the stamped build (read per launch there, ignored otherwise)
#ifdef HAET_STAMPS
    const char* e = getenv("HAET_DO_DBG");
    return e ? atoi(e) : 0;
#else
    return 0;
#endif
}

template <int MB>
static int deslice_out_launch(const float* x, const float* s_in, const float* Ws, const float* bs, float c0,
                              const float* cm, const float* cs, const float* S2, const unsigned short* Whi,
                              const unsigned short* Wlo, const float* bias, const float* residual, int B, int N,
                              float* out, hipStream_t st) {
    constexpr int M = 32 * MB, H = 8;
    const size_t lds = (size_t)2 * H * 2 * IMG + H * TN * 4 + 2 * H * M * 4 + kDO * 4 + MB * 2 * IMG + (size_t)H * MB * 4096;
    auto kern = residual ? deslice_out_kernel<MB, true> : deslice_out_kernel<MB, false>;
    static PerDeviceOnce attr;
    if (!attr.done()) {
        for (const void* k : {reinterpret_cast<const void*>(deslice_out_kernel<MB, true>),
                              reinterpret_cast<const void*>(deslice_out_kernel<MB, false>)}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(HAET_E_LAUNCH, "deslice_out: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        attr.set();
    }
    const int n_tiles = (N + TN - 1) / TN;
    int wgs = 256 / (B < 1 ? 1 : B);
    if (wgs < 1) wgs = 1;
    dim3 grid(n_tiles < wgs ? n_tiles : wgs, B);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, x, s_in, Ws, bs, c0, cm, cs, S2, Whi, Wlo, bias, residual, N, out,
                       do_dbg());
    return check_launch("deslice_out_kernel");
}

}  // namespace haet

using namespace haet;

#ifdef HAET_STAMPS
extern "C" int haet_debug_read_do_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(haet_do_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int haet_deslice_out_supported(int H, int C, int M, int O) {
    static const bool off = getenv("HAET_NO_DESLICE_OUT") != nullptr;  // A/B switch, read once
    return H == 8 && C == 32 && O == 256 && (M == 32 || M == 64) && !off;
}

extern "C" int haet_deslice_out_fwd_split(const float* cells, const float* s, const float* Ws, const float* bs, float c0,
                                          const float* col_max, const float* col_sum, const float* S2,
                                          const uint16_t* Whi, const uint16_t* Wlo, const float* bias,
                                          const float* residual, int B, int N, int H, int C, int M, int O, float* out,
                                          haet_stream_t stream) {
    if (!cells || !s || !Ws || !bs || !col_max || !col_sum || !S2 || !Whi || !Wlo || !out)
        return fail(HAET_E_ARG, "deslice_out: null pointer");
    if (B < 1 || N < 1) return fail(HAET_E_ARG, "deslice_out: B=%d N=%d", B, N);
    if (!(H == 8 && C == 32 && O == 256 && (M == 32 || M == 64)))
        return fail(HAET_E_ARG, "deslice_out: H=%d C=%d M=%d O=%d outside the fused kernel's envelope (8 x 32 -> 256, M 32 / 64)", H, C, M, O);
    if (M == 32)
        return deslice_out_launch<1>(cells, s, Ws, bs, c0, col_max, col_sum, S2, Whi, Wlo, bias, residual, B, N, out, as_stream(stream));
    return deslice_out_launch<2>(cells, s, Ws, bs, c0, col_max, col_sum, S2, Whi, Wlo, bias, residual, B, N, out, as_stream(stream));
}
