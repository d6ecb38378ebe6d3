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
//       X_h is loaded wave-privately (head h's 32 x 128 B block of the tile: no barrier for it), Ot goes to the
//       tile's O image [head][hi|lo][32 points][32 channels] as split bf16.
//   -- one workgroup barrier per tile (the O image is double-buffered) --
//   [B] wave w: out^T[o,n] = b_o + sum_{h,c} W_out[o, 32h+c] O_h[n,c]  (A = W_out rows o = 32w.., kept in 128 VGPRs
//       for the lifetime of the kernel like gemm_ws_kernel's; B = rows of the O image), 48 MFMAs, then four 16-B
//       stores per lane (its point's 16 outputs of this wave's 32 columns).
// Algorithmic HBM traffic: 1 KB read + 1 KB written per point (+ 32 B of s), against 4 KB for the two kernels.
// MFMA work per tile and wave: 24 (de-slice, M = 64) + 48 (GEMM) x 32 cycles; at two waves per SIMD that is
// ~0.27-0.33 ms per 1M points, the same as the 2 GB of traffic at the achievable HBM rate.
#include <cstdlib>

#include "slice_mfma_dev.h"

namespace haet {

constexpr int kDO = 256;  // outputs = H * C = 8 * 32

template <int MB, bool RES>
__global__ __launch_bounds__(512) void deslice_out_kernel(
    const float* __restrict__ x, const float* __restrict__ s_in, const float* __restrict__ Ws,
    const float* __restrict__ bs, float c0, const float* __restrict__ col_max, const float* __restrict__ col_sum,
    const float* __restrict__ S2, const unsigned short* __restrict__ Whi, const unsigned short* __restrict__ Wlo,
    const float* __restrict__ bias_out, const float* __restrict__ residual, int N, float* __restrict__ out, int dbg) {
    // dbg (HAET_DO_DBG, diagnostic runs only; 0 in production): 1 = no de-slice math, 2 = no GEMM MFMAs -- wave-uniform
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

    // stationary A fragments of the GEMM: rows o = 32 wave + l31 of W_out, k-step ks (16 wide): k = 16 ks + 8 hf ..
    u32x4 wh[16], wl[16];
    {
        const size_t o = (size_t)(32 * wave + l31) * kDO + 8 * hf;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            wh[ks] = *reinterpret_cast<const u32x4*>(Whi + o + 16 * ks);
            wl[ks] = *reinterpret_cast<const u32x4*>(Wlo + o + 16 * ks);
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
    // LDS byte offset of the hi half of cell i: rows 8 apart flip bit 1 of the chunk swizzle ((row >> 2) & 3)
    const int xoff0 = ximg + img_off(lr, lq >> 1) + ((lq & 1) << 3), xoff1 = ximg + img_off(lr + 8, lq >> 1) + ((lq & 1) << 3) - 512;
    auto xoff = [&](int i) { return ((i & 1) ? xoff1 : xoff0) + 512 * i; };
    f32x4 xr[4];
    float sv = 1.f;
    // Every global load of the loop is UNCONDITIONAL (rows clamped to the cloud, tiles clamped to this workgroup's last
    // one: a ghost fetch re-reads it): a load under a branch -- even a wave-uniform one -- makes hipcc's waitcnt pass give
    // up counting and wait vmcnt(0) where the cells are staged, i.e. for the previous tile's output STORES as well
    // (loads and stores share one in-order counter).  Rows past the end of the cloud read the last row again; their
    // results are never stored.
    auto fetch = [&](int t) {
        const char* tb = reinterpret_cast<const char*>(xb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = min(t * TN + lr + 8 * i, N - 1);
            xr[i] = *reinterpret_cast<const f32x4*>(tb + (size_t)row * (H * kC * 4) + (wave * kC + 4 * lq) * 4);
        }
        sv = sb[(size_t)min(t * TN + (lane & 31), N - 1) * H + wave];
    };
    // L2 touch two tiles ahead: one dword per 128-B line of this wave's block of a tile (32 rows x 128 B: two rows... one
    // line per lane 0-31, the second half-wave the same lines), result folded into a word nothing reads.  The real 16-B
    // loads of that tile, issued one iteration later, then find their lines in L2.
    unsigned touch_acc = 0, touch_pend = 0;
    auto touch = [&](int t) {
        touch_acc ^= touch_pend;
        const int row = min(t * TN + (lane & 31), N - 1);
        touch_pend = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(xb) + (size_t)row * (H * kC * 4) + wave * (kC * 4));
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 w = __builtin_bit_cast(u32x4, xr[i]);
            const u32x2 hi = {w[0], w[1]}, lo = {w[2], w[3]};
            *reinterpret_cast<u32x2*>(smem + xoff(i)) = hi;
            *reinterpret_cast<u32x2*>(smem + xoff(i) + IMG) = lo;
        }
        *reinterpret_cast<float*>(smem + sarr + 4 * (wave * TN + (lane & 31))) = sv;  // (both half-waves: same value)
    };

    const int t_begin = blockIdx.x;
    if (t_begin >= n_tiles) return;
    const int t_last = t_begin + ((n_tiles - 1 - t_begin) / G) * G;  // this workgroup's last tile
    fetch(t_begin);
    stage();
    fetch(min(t_begin + G, t_last));
    touch(min(t_begin + 2 * G, t_last));
    {
        // The loop's steady state has four output stores in flight behind these loads.  hipcc's waitcnt pass merges the
        // loop-entry state with the back edge's conservatively: without the same four stores pending at the entry it
        // waits, every iteration, for the previous tile's stores to drain before staging the cells (vmcnt(1) instead of
        // vmcnt(5)).  So the prologue issues them too: zeros to this workgroup's first tile, overwritten by its [B].
        const int rows0 = N - t_begin * TN < TN ? N - t_begin * TN : TN;
        __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char*>(ob) + (size_t)t_begin * TN * (kDO * 4), 0, rows0 * (kDO * 4), 0x00020000);
        const int voff = l31 * (kDO * 4) + (32 * wave + 4 * hf) * 4;
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(z, r0, voff + 32 * i, 0, 0);
    }
    __syncthreads();  // constants (bias, m2, Ws images) staged by all threads
    // Tile t + G sits in registers while tile t is de-sliced; its cells go to LDS after [A], and the loads of tile
    // t + 2G are issued right then, so that they are in flight across the GEMM phase and the next de-slice phase
    // (a whole iteration) instead of across [A] alone.
    for (int t = t_begin; t < n_tiles; t += G) {
        const int nvalid = N - t * TN;
        const int oimg = oimg0;  // (single image: the second barrier below frees it for the next tile)
        // ---- [A] de-slice of head `wave`: Ot[c, n] (lane = point n, registers = channels)
        {
            const float sn = *reinterpret_cast<const float*>(smem + sarr + 4 * (wave * TN + l31));
            f32x16 O;
#pragma unroll
            for (int r = 0; r < 16; ++r) O[r] = 0.f;
#pragma unroll
            for (int gb = 0; gb < MB; ++gb) {
                if (dbg & 1) break;
                // Short live ranges on purpose (the wave also holds 128 registers of W_out and the next tile's
                // cells): the x fragments are re-read per column block, the column maxima read where they are used,
                // and the column blocks are not interleaved.
                __builtin_amdgcn_sched_barrier(0);
                const int bp = biasL + 4 * (wave * M + 32 * gb + 4 * hf);
                const int mp = m2L + 4 * (wave * M + 32 * gb + 4 * hf);
                f32x16 Y;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(smem + bp + 32 * i);
#pragma unroll
                    for (int q = 0; q < 4; ++q) Y[4 * i + q] = b4[q];
                }
                const int wi = wimg + gb * (2 * IMG);
                Y = mfma3(read_rows(wi, l31, hf), read_rows(ximg, l31, hf), Y);
                Y = mfma3(read_rows(wi, l31, 2 + hf), read_rows(ximg, l31, 2 + hf), Y);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const f32x4 ma = *reinterpret_cast<const f32x4*>(smem + mp + 32 * (2 * half));
                    const f32x4 mb = *reinterpret_cast<const f32x4*>(smem + mp + 32 * (2 * half + 1));
                    float pv[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        pv[j] = __builtin_amdgcn_exp2f(fmaf(Y[8 * half + j], sn, -ma[j]));
                        pv[4 + j] = __builtin_amdgcn_exp2f(fmaf(Y[8 * half + 4 + j], sn, -mb[j]));
                    }
                    Frag sfr;
                    {
                        const int off = sfL + wave * SFW + (gb * 2 + half) * 2048 + lane * 16;
                        sfr.hi = *reinterpret_cast<const u32x4*>(smem + off);
                        sfr.lo = *reinterpret_cast<const u32x4*>(smem + off + 1024);
                    }
                    O = mfma3(sfr, split8(pv), O);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // O image of this head: row = point l31, channels c = 8 i + 4 hf .. + 3 from registers 4i .. 4i + 3
            const int oh = oimg + wave * (2 * IMG);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned h01 = pack_bf16(O[4 * i], O[4 * i + 1]), h23 = pack_bf16(O[4 * i + 2], O[4 * i + 3]);
                const u32x2 hi = {h01, h23};
                const u32x2 lo = {pack_bf16(O[4 * i] - __uint_as_float(h01 << 16), O[4 * i + 1] - __uint_as_float(h01 & 0xffff0000u)),
                                  pack_bf16(O[4 * i + 2] - __uint_as_float(h23 << 16), O[4 * i + 3] - __uint_as_float(h23 & 0xffff0000u))};
                const int off = oh + img_off(l31, i) + (hf << 3);
                *reinterpret_cast<u32x2*>(smem + off) = hi;
                *reinterpret_cast<u32x2*>(smem + off + IMG) = lo;
            }
        }
        stage();  // this wave's own x block: nobody else reads it (after the last tile: a ghost)
        fetch(min(t + 2 * G, t_last));
        touch(min(t + 3 * G, t_last));
        __syncthreads();    // the tile's O image is complete
        // ---- [B] to_out: out^T[o, n] for o = 32 wave + (r&3) + 8 (r>>2) + 4 hf, n = l31
        {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(smem + boL + 4 * (32 * wave + 8 * i + 4 * hf));
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[4 * i + q] = b4[q];
            }
            // residual rows of this lane's point (the block's `Attn(ln_1(fx)) + fx`,
            // HAETransolver_Irregular_Mesh.py:125): requested before the 48 MFMAs, added after them
            f32x4 res[4];
            if (RES) {  // (rows past the end: clamped, never stored)
                const float* rrow = rb + (size_t)min(t * TN + l31, N - 1) * kDO + 32 * wave + 4 * hf;
#pragma unroll
                for (int i = 0; i < 4; ++i) res[i] = *reinterpret_cast<const f32x4*>(rrow + 8 * i);
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                if (dbg & 2) break;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const Frag of = read_rows(oimg + h * (2 * IMG), l31, 2 * s + hf);
                    Frag wf;
                    wf.hi = wh[2 * h + s];
                    wf.lo = wl[2 * h + s];
                    acc = mfma3(wf, of, acc);
                }
            }
            // Stores through a buffer descriptor over THIS tile's valid rows: rows past the end of the cloud fall outside
            // its range and the hardware drops them -- no branch around the stores (a conditional store, like a
            // conditional load, costs the loop its counted waits), no per-lane address clamping.
            {
                const int rows = nvalid < TN ? nvalid : TN;
                __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    reinterpret_cast<char*>(ob) + (size_t)t * TN * (kDO * 4), 0, rows * (kDO * 4), 0x00020000);
                const int voff = l31 * (kDO * 4) + (32 * wave + 4 * hf) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f32x4 v = {acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]};
                    if (RES) v += res[i];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, voff + 32 * i, 0, 0);
                }
            }
        }
        __syncthreads();  // every wave has read the O image
    }
    touch_acc ^= touch_pend;
    asm volatile("" ::"v"(touch_acc));
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
                       getenv("HAET_DO_DBG") ? atoi(getenv("HAET_DO_DBG")) : 0);
    return check_launch("deslice_out_kernel");
}

}  // namespace haet

using namespace haet;

extern "C" int haet_deslice_out_supported(int H, int C, int M, int O) {
    return H == 8 && C == 32 && O == 256 && (M == 32 || M == 64) && !getenv("HAET_NO_DESLICE_OUT");
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
