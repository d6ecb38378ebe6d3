// Conv2d / Conv3d in_project_x of the structured-mesh modules (models/PhysicsAttention/StructuredMesh2D.py:173-189,
// StructuredMesh3D.py:177-193) as an implicit GEMM built around LDS-DMA -- round 4's replacement of gemm_x3_kernel<CONV>
// for the wide shapes (O = 256: C5's 256 -> 8 x 32 projection).
//
// What the old kernel paid per 32-wide k-tile and 128-row tile: the fp32 -> {hi, lo} split of every x element AGAIN
// for each of the 27 taps, 12 LDS stores per thread, the neighbour / bounds arithmetic per 16-B chunk, one barrier --
// its vector issue was as loaded as its matrix pipe (50 % busy).  Here:
//   * x is split ONCE per forward into k-tile cells (haet_conv_presplit: per row and 32-channel block 128 B =
//     {32 x bf16 hi | 32 x bf16 lo}; same bytes as the fp32 tensor), the row after the last is all zeros: a tap that
//     leaves the grid reads THAT row -- zero padding with no select on the data path;
//   * both operands travel HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: one wave instruction = 8 rows x 128 B,
//     whole lines, no registers, no LDS store instructions, no conversion): 8 per wave and k-tile;
//   * workgroup tile 256 rows x 256 columns (all output channels), 8 waves as 2 x 4, wave tile 128 x 64 = 32 accumulator
//     tiles of v_mfma_f32_16x16x32_bf16; a k-tile = 32 channels of one tap: 96 MFMAs (split-bf16 x 3) per wave against
//     24 ds_read_b128 -- the three products of an operand pair share their four fragments;
//   * two 64-KB LDS stages, ONE barrier per k-tile: the DMA of tile t + 1 is issued right behind the barrier that ends
//     tile t - 1's reads and has the 3k matrix-pipe cycles of tile t to land;
//   * LDS image: row-major 128-B rows, 16-B slot (plane * 4 + chunk) XOR ((row >> 1) & 7): conflict-free for the
//     ds_read_b128 fragment reads of the 16 x 16 x 32 shape; LDS-DMA writes linearly, so the permutation is applied to
//     the SOURCE address of each lane (cdna_hip_programming.md rule 21);
//   * K order [channel block][tap] (taps of one 128-B channel block follow each other: shifted rows re-read cached
//     lines) and an XCD-aware tile order (each XCD walks a contiguous run of row tiles, so the halo lines of
//     neighbouring tiles are shared in ITS L2).
#include "common.h"

namespace haet {
namespace c8 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

extern __shared__ __attribute__((aligned(1024))) char csm[];

constexpr int kThreads = 512;
constexpr int BM = 256, BN = 256;
constexpr int kAStage = BM * 128, kWStage = BN * 128, kStage = kAStage + kWStage;  // 64 KB
constexpr int kLds = 2 * kStage;
constexpr int kDefaultVar = 5;

struct Geom {
    int Cin, Npts, gH, gW, gD, kH, kW, kD;
};
struct Split {          // ACT == 1: cells + s = log2(e) / tau (Ada-Temp over each 32-channel head)
    const float* wt;    // ada_temp_linear.weight [32]
    float bt, base_temp;
    float* s_out;       // [Nrows][O / 32]
};

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

__device__ __forceinline__ void dma16(const void* src, int lds_off) {
    __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(csm + lds_off), 16, 0, 0);
}
__device__ __forceinline__ u32x4 lds16(int off) { return *reinterpret_cast<const u32x4*>(csm + off); }

__device__ __forceinline__ f32x4 mfma3(const u32x4 ah, const u32x4 al, const u32x4 bh, const u32x4 bl, f32x4 c) {
    const bf16x8 a_h = __builtin_bit_cast(bf16x8, ah), a_l = __builtin_bit_cast(bf16x8, al);
    const bf16x8 b_h = __builtin_bit_cast(bf16x8, bh), b_l = __builtin_bit_cast(bf16x8, bl);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_l, b_h, c, 0, 0, 0);
    return c;
}

// PREC = 1 ("fp16x2", opt-in: haet_amd.set_precision / the trailing argument of the entry points): activations as fp16 hi +
// fp16 lo, the weight as ONE fp16 plane -- two products per term instead of three; ~2e-4 against fp64 (bf16x3: 4e-6),
// inside north_star's 1e-3 and the reference's own fp16-autocast grade; BASELINE config C5 names "fp16 MFMA".
__device__ __forceinline__ f32x4 mfma2h(const u32x4 ah, const u32x4 al, const u32x4 b, f32x4 c) {
    const f16x8 a_h = __builtin_bit_cast(f16x8, ah), a_l = __builtin_bit_cast(f16x8, al), b_h = __builtin_bit_cast(f16x8, b);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, b_h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l, b_h, c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ void split_f16_pair(float a, float b, unsigned& hi, unsigned& lo) {
    const f32x2 v = {a, b};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    const f32x2 r = {a - hf[0], b - hf[1]};
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

// x fp32 [rows][Cin] -> cells [rows + 1][Cin / 32][{32 hi | 32 lo}] bf16, row `rows` zeroed.  One thread = 8 channels
// (two 16-B loads, one 16-B store into each plane half); LNORM = 1: nn.LayerNorm(Cin) over the row first (ln_1 in
// front of the projection, HAETransolver_Structured_Mesh_3D.py block: the normalised activations never exist in HBM;
// Cin = 256: a row = 32 threads).
template <int LNORM, int PREC>
__global__ __launch_bounds__(256) void presplit_kernel(const float* __restrict__ x, long long rows, int Cin,
                                                       char* __restrict__ cells, const float* __restrict__ ln_g,
                                                       const float* __restrict__ ln_b, float ln_eps) {
    const int tpr = Cin >> 3;  // threads per row
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long row = gid / tpr;
    const int t = (int)(gid - row * tpr);
    if (row > rows) return;
    const int cb = t >> 2, q = t & 3;  // channel block, 8-channel chunk inside it
    char* dst = cells + (size_t)row * Cin * 4 + cb * 128 + q * 16;
    if (row == rows) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(dst) = z;
        *reinterpret_cast<u32x4*>(dst + 64) = z;
        return;
    }
    const float* src = x + (size_t)row * Cin + 8 * t;
    f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
    if (LNORM) {  // Cin == 256 only (32 threads = half a wave per row): two-pass statistics like torch
        float s = (a[0] + a[1]) + (a[2] + a[3]) + (b[0] + b[1]) + (b[2] + b[3]);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * (1.0f / 256.0f);
        float qv = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] -= mean, b[e] -= mean;
            qv = fmaf(a[e], a[e], qv);
            qv = fmaf(b[e], b[e], qv);
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) qv += __shfl_xor(qv, o);
        const float rstd = rsqrtf(qv * (1.0f / 256.0f) + ln_eps);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(ln_g + 8 * t), g1 = *reinterpret_cast<const f32x4*>(ln_g + 8 * t + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(ln_b + 8 * t), b1 = *reinterpret_cast<const f32x4*>(ln_b + 8 * t + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = fmaf(a[e] * rstd, g0[e], b0[e]);
            b[e] = fmaf(b[e] * rstd, g1[e], b1[e]);
        }
    }
    unsigned h[4], l[4];
    if (PREC == 0) {
        split_bf16_pair(a[0], a[1], h[0], l[0]);
        split_bf16_pair(a[2], a[3], h[1], l[1]);
        split_bf16_pair(b[0], b[1], h[2], l[2]);
        split_bf16_pair(b[2], b[3], h[3], l[3]);
    } else {
        split_f16_pair(a[0], a[1], h[0], l[0]);
        split_f16_pair(a[2], a[3], h[1], l[1]);
        split_f16_pair(b[0], b[1], h[2], l[2]);
        split_f16_pair(b[2], b[3], h[3], l[3]);
    }
    *reinterpret_cast<u32x4*>(dst) = u32x4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<u32x4*>(dst + 64) = u32x4{l[0], l[1], l[2], l[3]};
}

// ACT 0: out fp32 [rows][256] (+ bias); ACT 1: cells + s (the streaming passes' operand format, gemm_ws_kernel<SPLIT>'s)
// Schedule parameters (A/B switch HAET_C8_VAR, tools/conv8_probe.py; the shipped choice is kDefaultVar):
//   P0 / P1: the A-fragment step of the k-tile behind which waves 0-3 / 4-7 issue their DMAs (-1: right behind the barrier).
//     With both halves right behind the barrier the two waves of every SIMD stand in the DMA issue queue together while the
//     matrix pipe idles;
//   PRIO: 1 = static s_setprio 1 for waves 4-7 (the second-dispatched half loses the arbitration otherwise:
//     MI355X_MICROARCH.md, two waves per SIMD, item 4), 2 = for waves 0-3 instead;
// Measured on C5's slab (same process, interleaved; tools/conv8_probe.py): both halves behind the barrier, no priority 16.3 ms;
// P1 = 3 alone 16.0; priority alone 16.1; P1 = 3 or 4 with priority 14.9-15.1 (shipped: P1 = 4); P1 = 2: 15.4; priority for
// waves 0-3 instead: 16.1; the W DMAs two steps behind the A DMAs: 15.6; waves 4-7 holding the last A-step's MFMAs back
// across the barrier (a deferred-epilogue stagger): 15.9 with mid-tile DMAs, 16.7 without -- not kept.
template <int ACT, int P0, int P1, int PRIO, int PREC>
__global__ __launch_bounds__(kThreads) void conv8_kernel(const char* __restrict__ xc, const unsigned short* __restrict__ Whi,
                                                         const unsigned short* __restrict__ Wlo, const float* __restrict__ bias,
                                                         int Nrows, int n_row_tiles, float* __restrict__ out, Geom cg, Split cs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kg = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;
    const int taps = cg.kH * cg.kW * cg.kD, ncb = cg.Cin >> 5, nkt = ncb * taps, K = nkt * 32;
    const size_t rowB = (size_t)cg.Cin * 4;  // bytes of one cell row

    // XCD-aware persistent schedule: workgroups b, b + 8, ... share an XCD (round-robin dispatch: speed only); XCD x walks
    // the contiguous run of row tiles [x * chunk, (x + 1) * chunk)
    const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3, wper = gridDim.x >> 3;
    const int chunk = (n_row_tiles + 7) >> 3;
    const int t_lo = xcd * chunk, t_hi = min(n_row_tiles, t_lo + chunk);
    const int my_tiles = t_lo + widx < t_hi ? (t_hi - t_lo - widx + wper - 1) / wper : 0;
    if (my_tiles == 0) return;
    const int total = my_tiles * nkt;

    // ---- DMA source geometry.  Instruction j (0..3) of this wave covers rows 32 wave + 8 j + lane / 8 of the A stage
    // (and the same output columns of the W stage); this lane's 16 B are physical slot lane % 8 of that row, i.e. logical
    // slot q = pslot ^ ((row >> 1) & 7) = q0 for even j and q0 ^ 4 (the other plane, same chunk) for odd j.
    // Kept small on purpose (the fetch is instantiated at two places of the loop, VAR & 1): per row ONE register -- a bit
    // per tap: "the neighbour at this tap is inside the grid" -- made once per row tile.
    const int lrow = lane >> 3, pslot = lane & 7;
    const int rloc = 32 * wave + lrow;                 // + 8 j
    const int q0 = pslot ^ ((rloc >> 1) & 7);
    const int aq0 = q0 * 16;
    const char* const w_even = reinterpret_cast<const char*>((q0 >> 2) ? Wlo : Whi) + ((size_t)rloc * K + (q0 & 3) * 8) * 2;
    const char* const w_odd = reinterpret_cast<const char*>((q0 >> 2) ? Whi : Wlo) + ((size_t)rloc * K + (q0 & 3) * 8) * 2;
    // PREC = 1: the W stage is ONE plane, 64-B rows: a DMA instruction = 16 columns x 64 B, two per wave and k-tile;
    // 16-B chunk c of column n sits in slot c ^ (-(n >> 2) & 3) (conflict-free ds_read_b128 fragment reads of 64-B rows)
    const int wcol1 = 32 * wave + (lane >> 2);  // + 16 j
    const char* const w_one = reinterpret_cast<const char*>(Whi) + ((size_t)wcol1 * K + (((lane & 3) ^ (-(wcol1 >> 2) & 3)) * 8)) * 2;
    const int a_lds0 = (32 * wave) * 128, w_lds0 = kAStage + (32 * wave) * 128;  // + j * 1024; the hardware adds lane * 16
    // fetch cursor (one item ahead of the compute cursor): scalars
    int f_tile = t_lo + widx, f_cb = 0, f_a = 0, f_b = 0, f_c = 0, f_kt = 0, f_tap = 0;
    unsigned tapmask[4];
    auto tile_masks = [&](int tile) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = tile * BM + rloc + 8 * j;
            unsigned m = 0;
            if (r < Nrows) {  // (a row past the end keeps an empty mask: every tap reads the zero row)
                const int p = r % cg.Npts, ij = p / cg.gD, k0 = p - ij * cg.gD, i0 = ij / cg.gW, j0 = ij - i0 * cg.gW;
                int bit = 0;
                for (int a = 0; a < cg.kH; ++a)
                    for (int b = 0; b < cg.kW; ++b)
                        for (int c = 0; c < cg.kD; ++c, ++bit) {
                            const bool ok = (unsigned)(i0 + a - (cg.kH >> 1)) < (unsigned)cg.gH &&
                                            (unsigned)(j0 + b - (cg.kW >> 1)) < (unsigned)cg.gW &&
                                            (unsigned)(k0 + c - (cg.kD >> 1)) < (unsigned)cg.gD;
                            m |= ok ? 1u << bit : 0u;
                        }
            }
            tapmask[j] = m;
        }
    };
    tile_masks(f_tile);
    auto fetch = [&](int stage) {
        const int shift = ((f_a - (cg.kH >> 1)) * cg.gW + (f_b - (cg.kW >> 1))) * cg.gD + (f_c - (cg.kD >> 1));
        const int cb_off = f_cb * 128, r0 = f_tile * BM + rloc + shift;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sr = (tapmask[j] >> f_tap) & 1u ? r0 + 8 * j : Nrows;
            dma16(xc + (size_t)sr * rowB + cb_off + (aq0 ^ ((j & 1) << 6)), stage + a_lds0 + j * 1024);
        }
        if (PREC == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                dma16(((j & 1) ? w_odd : w_even) + (size_t)j * 8 * K * 2 + (size_t)f_kt * 64, stage + w_lds0 + j * 1024);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)  // ((-(n >> 2)) & 3 is the same for n and n + 16)
                dma16(w_one + (size_t)j * 16 * K * 2 + (size_t)f_kt * 64, stage + kAStage + (32 * wave + 16 * j) * 64);
        }
    };
    auto advance = [&]() {  // tap fastest, then channel block, then the row tile
        ++f_kt, ++f_tap;
        if (++f_c == cg.kD) {
            f_c = 0;
            if (++f_b == cg.kW) {
                f_b = 0;
                if (++f_a == cg.kH) {
                    f_a = 0, f_tap = 0;
                    if (++f_cb == ncb) {
                        f_cb = 0, f_kt = 0;
                        f_tile += wper;
                        if (f_tile < t_hi) tile_masks(f_tile);  // (past the last tile the ghost fetch re-reads the last tile's rows)
                        else f_tile -= wper;
                    }
                }
            }
        }
    };

    // ---- fragment read offsets (see the header: slot XOR (row >> 1) & 7; the lo plane is the hi slot XOR 4)
    const int sw = (l15 >> 1) & 7;
    const int a_rd = (128 * wr + l15) * 128 + ((kg ^ sw) << 4);
    const int b_rd = PREC == 0 ? kAStage + (64 * wc + l15) * 128 + ((kg ^ sw) << 4)
                               : kAStage + (64 * wc + l15) * 64 + ((kg ^ (-(l15 >> 2) & 3)) << 4);  // + j * 1024 (16 columns x 64 B)

    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = bias ? bias[64 * wc + 16 * j + l15] : 0.f;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{bv[j], bv[j], bv[j], bv[j]};
    float wt4[4][4];
    if (ACT == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) wt4[j][q] = cs.wt[16 * (j & 1) + (l15 & ~3) + q];
    }

    fetch(0);
    advance();
    if (PRIO && wr == (PRIO == 1 ? 1 : 0)) __builtin_amdgcn_s_setprio(1);
    auto epilogue = [&](int tile) __attribute__((always_inline)) {
        const long long row0 = (long long)tile * BM + 128 * wr;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long long rbase = row0 + 16 * i + 4 * kg;  // first of this lane's 4 rows of tile row block i
            if (ACT == 1) {
                // a 4 x 4 transpose across the lane quad turns four rows of ONE column into ONE row (rbase + lane % 4) and
                // four consecutive columns = one 16-B cell {4 x hi, 4 x lo}; the Ada-Temp dot over a head's 32 channels =
                // column blocks (2 h, 2 h + 1) of this lane + the other three quads of its 16-lane row
                const long long rowi = rbase + (lane & 3);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float t = 0.f;
                    u32x4 o4[2];
#pragma unroll
                    for (int jo = 0; jo < 2; ++jo) {
                        const int j = 2 * h + jo;
                        int v[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = __float_as_int(acc[i][j][q]);
                        const bool odd = lane & 1, up = lane & 2;
#pragma unroll
                        for (int m = 0; m < 4; m += 2) {
                            const int send = odd ? v[m] : v[m + 1];
                            const int recv = __builtin_amdgcn_update_dpp(0, send, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
                            v[m] = odd ? recv : v[m];
                            v[m + 1] = odd ? v[m + 1] : recv;
                        }
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const int send = up ? v[m] : v[m + 2];
                            const int recv = __builtin_amdgcn_update_dpp(0, send, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
                            v[m] = up ? recv : v[m];
                            v[m + 2] = up ? v[m + 2] : recv;
                        }
                        const float f0 = __int_as_float(v[0]), f1 = __int_as_float(v[1]), f2 = __int_as_float(v[2]), f3 = __int_as_float(v[3]);
                        unsigned h01, l01, h23, l23;
                        split_bf16_pair(f0, f1, h01, l01);
                        split_bf16_pair(f2, f3, h23, l23);
                        o4[jo] = u32x4{h01, h23, l01, l23};
                        t = fmaf(f0, wt4[j][0], t);
                        t = fmaf(f1, wt4[j][1], t);
                        t = fmaf(f2, wt4[j][2], t);
                        t = fmaf(f3, wt4[j][3], t);
                    }
                    t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x124, 0xF, 0xF, false));  // row_ror:4
                    t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x128, 0xF, 0xF, false));  // row_ror:8
                    const float tau = cs.base_temp + fminf(fmaxf(t + cs.bt, -0.4f), 0.4f);
                    if (rowi < Nrows) {
                        const int head = 2 * wc + h;
#pragma unroll
                        for (int jo = 0; jo < 2; ++jo)
                            *reinterpret_cast<u32x4*>(out + rowi * BN + 32 * head + 16 * jo + (l15 & ~3)) = o4[jo];
                        cs.s_out[rowi * (BN / 32) + head] = 1.4426950408889634f * __builtin_amdgcn_rcpf(tau);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float* oc = out + rbase * BN + 64 * wc + 16 * j + l15;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (rbase + r < Nrows) oc[(size_t)r * BN] = acc[i][j][r];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{bv[j], bv[j], bv[j], bv[j]};
        }
    };
    int c_tile = t_lo + widx, c_kt = 0;
    for (int it = 0; it < total; ++it) {
        const int st = (it & 1) * kStage;
        // tile `it` has landed (this wave's DMAs), then: everyone's have, and everyone is done reading the other stage
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // (the fetches are unconditional: the item past the end is a ghost into the stage nobody reads again)
        if (P0 == P1 && P0 < 0) {
            fetch(st ^ kStage);
        } else {
            if (P0 < 0 && wr == 0) fetch(st ^ kStage);
            if (P1 < 0 && wr == 1) fetch(st ^ kStage);
        }

        u32x4 bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bh[j] = lds16(st + b_rd + j * (PREC == 0 ? 2048 : 1024));
            if (PREC == 0) bl[j] = lds16(st + (b_rd ^ 64) + j * 2048);
        }
        u32x4 ah = lds16(st + a_rd), al = lds16(st + (a_rd ^ 64));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            u32x4 nh = ah, nl = al;
            if (i < 7) {
                nh = lds16(st + a_rd + (i + 1) * 2048);
                nl = lds16(st + (a_rd ^ 64) + (i + 1) * 2048);
            }
            if (P0 == P1) {
                if (i == P0) fetch(st ^ kStage);
            } else {
                if (i == P0 && wr == 0) fetch(st ^ kStage);
                if (i == P1 && wr == 1) fetch(st ^ kStage);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = PREC == 0 ? mfma3(ah, al, bh[j], bl[j], acc[i][j]) : mfma2h(ah, al, bh[j], acc[i][j]);
            ah = nh, al = nl;
        }
        advance();
        if (++c_kt == nkt) {  // row tile finished: its stores go out behind the next barrier (waves 4-7: behind their held-back MFMAs)
            c_kt = 0;
            epilogue(c_tile);
            c_tile += wper;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ghost fetch must not outlive the workgroup's LDS
}

}  // namespace c8
}  // namespace haet

using namespace haet;

extern "C" int haet_conv_cells_supported(int Cin, int O) { return (Cin >= 32 && Cin % 32 == 0 && O == 256) ? 1 : 0; }

extern "C" size_t haet_conv_cells_bytes(long long rows, int Cin) { return (size_t)(rows + 1) * (size_t)Cin * 4; }

extern "C" int haet_conv_presplit(const float* x, long long rows, int Cin, void* cells, const float* ln_g, const float* ln_b,
                                  float ln_eps, int fp16x2, haet_stream_t stream) {
    if (!x || !cells || rows < 1) return fail(HAET_E_ARG, "conv_presplit: null pointer / no rows");
    if (Cin < 32 || Cin % 32) return fail(HAET_E_ARG, "conv_presplit: C_in=%d is not a multiple of 32", Cin);
    if ((ln_g != nullptr) != (ln_b != nullptr)) return fail(HAET_E_ARG, "conv_presplit: ln_g / ln_b go together");
    if (ln_g && Cin != 256) return fail(HAET_E_ARG, "conv_presplit: the fused LayerNorm needs C_in = 256");
    const long long threads = (rows + 1) * (Cin / 8);
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, as_stream(stream), x, rows, Cin, reinterpret_cast<char*>(cells), ln_g,
                           ln_b, ln_eps);
    };
    if (ln_g) fp16x2 ? go(c8::presplit_kernel<1, 1>) : go(c8::presplit_kernel<1, 0>);
    else fp16x2 ? go(c8::presplit_kernel<0, 1>) : go(c8::presplit_kernel<0, 0>);
    return check_launch("presplit_kernel");
}

static int conv_cells_impl(const void* xc, const uint16_t* W_hi, const uint16_t* W_lo, const float* bias, int B, int gH, int gW,
                           int gD, int Cin, int O, int kH, int kW, int kD, float* out, haet_stream_t stream,
                           const c8::Split* split, int fp16x2) {
    if (!xc || !W_hi || (!W_lo && !fp16x2) || !out) return fail(HAET_E_ARG, "conv_project_cells_fwd: null pointer");
    if (!haet_conv_cells_supported(Cin, O))
        return fail(HAET_E_ARG, "conv_project_cells_fwd: C_in=%d (multiple of 32) / O=%d (256) unsupported", Cin, O);
    if (B < 1 || gH < 1 || gW < 1 || gD < 1 || kH < 1 || kW < 1 || kD < 1 || !(kH & kW & kD & 1))
        return fail(HAET_E_ARG, "conv_project_cells_fwd: bad grid / kernel (odd kernel sizes only)");
    const long long rows = (long long)B * gH * gW * gD;
    if (rows > 0x7fffffffLL - 256) return fail(HAET_E_ARG, "conv_project_cells_fwd: too many points");
    const int Nrows = (int)rows, tiles = (Nrows + c8::BM - 1) / c8::BM;
    int cus = resident_workgroup_limit();
    cus -= cus % 8;
    if (cus < 8) cus = 8;
    int grid = ((tiles + 7) / 8) * 8;
    if (grid > cus) grid = cus;
    const c8::Geom cg{Cin, gH * gW * gD, gH, gW, gD, kH, kW, kD};
    static PerDeviceOnce once[10];
    auto launch = [&](auto kern, int which) {
        if (!once[which].done()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, c8::kLds);
            if (e != hipSuccess) return fail(HAET_E_LAUNCH, "conv_project_cells_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
            once[which].set();
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(c8::kThreads), c8::kLds, as_stream(stream), reinterpret_cast<const char*>(xc), W_hi,
                           W_lo, bias, Nrows, tiles, out, cg, split ? *split : c8::Split{});
        return check_launch("conv8_kernel");
    };
    const int taps = kH * kW * kD;
    if (taps > 32) return fail(HAET_E_ARG, "conv_project_cells_fwd: %d taps (at most 32: one validity bit per tap)", taps);
    const char* ve = getenv("HAET_C8_VAR");  // A/B switch, read per call (tools/conv8_probe.py)
    const int var = ve ? atoi(ve) : c8::kDefaultVar;
    if (fp16x2) return split ? launch(c8::conv8_kernel<1, -1, 4, 1, 1>, 9) : launch(c8::conv8_kernel<0, -1, 4, 1, 1>, 8);
    if (split) return launch(c8::conv8_kernel<1, -1, 4, 1, 0>, 7);
    switch (var) {
        case 0: return launch(c8::conv8_kernel<0, -1, -1, 0, 0>, 0);
        case 1: return launch(c8::conv8_kernel<0, -1, 3, 0, 0>, 1);
        case 2: return launch(c8::conv8_kernel<0, -1, -1, 1, 0>, 2);
        case 3: return launch(c8::conv8_kernel<0, -1, 3, 1, 0>, 3);
        default: return launch(c8::conv8_kernel<0, -1, 4, 1, 0>, 5);
    }
}

extern "C" int haet_conv_project_cells_fwd(const void* x_cells, const uint16_t* W_hi, const uint16_t* W_lo, const float* bias,
                                           int B, int gH, int gW, int gD, int Cin, int O, int kH, int kW, int kD, float* out,
                                           int fp16x2, haet_stream_t stream) {
    return conv_cells_impl(x_cells, W_hi, W_lo, bias, B, gH, gW, gD, Cin, O, kH, kW, kD, out, stream, nullptr, fp16x2);
}

extern "C" int haet_conv_project_cells_split_fwd(const void* x_cells, const uint16_t* W_hi, const uint16_t* W_lo,
                                                 const float* bias, const float* wt, float bt, float base_temp, int B, int gH,
                                                 int gW, int gD, int Cin, int O, int kH, int kW, int kD, void* cells_out,
                                                 float* s_out, int fp16x2, haet_stream_t stream) {
    if (!wt || !s_out) return fail(HAET_E_ARG, "conv_project_cells_split_fwd: null pointer");
    const c8::Split cs{wt, bt, base_temp, s_out};
    return conv_cells_impl(x_cells, W_hi, W_lo, bias, B, gH, gW, gD, Cin, O, kH, kW, kD, reinterpret_cast<float*>(cells_out), stream, &cs, fp16x2);
}
