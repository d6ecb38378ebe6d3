// Per-point projections (in_project_x, to_out) as a split-bf16 MFMA GEMM:
//     out[n, o] = sum_k x[n, k] W[o, k] + b[o]        x, out fp32;  K, O = 256 at C3
// The reference runs these as fp32 cuBLAS GEMMs (models/PhysicsAttention/IrregularMesh.py:158,
// 235).  fp32 on MI355X is VALU/“f32 MFMA” rate (157 TF): 131 kflop/point makes the GEMM
// compute-bound at ~1.15 ms per 1M points (rocBLAS, measured), 3x the HBM time.  Split-bf16
// (hi*hi + hi*lo + lo*hi, fp32 accumulate, ~2^-16 operand error) runs on the bf16 matrix cores
// at 3 MFMAs per fp32 product and makes the GEMM HBM-bound again (2 * K * 4 B per point).
//
// Workgroup tile = 128 rows x BN columns, 8 waves as 2 (rows) x 4 (columns); K is walked in
// 32-wide tiles: x tile fp32 -> {hi, lo} bf16 in the loader, W comes pre-split (haet_split_bf16,
// done once per weight version), both as swizzled [32][32] bf16 LDS images read with
// ds_read_b128.  Persistent workgroups with a 3-deep (x) / 2-deep (W) register prefetch keep the
// HBM stream full across row tiles (see the kernel).
#include "common.h"

#include <type_traits>

namespace haet {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int GBK = 32;    // k-tile
constexpr int GIMG = 2048; // bytes of one [32][32] bf16 image

__device__ __forceinline__ unsigned g_pack(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// (round 4: the chunk swizzle is -(row >> 2) & 3, not (row >> 2) & 3 -- both are conflict-free for fragment reads of the
// 32 x 32 x 16 shape, only this one also for the 16 x 16 x 32 shape the streaming kernel's products moved to in round 3,
// whose ds_read_b128 lane groups see rows r and r + 4 together: deslice_out.hip, do_img_off)
__device__ __forceinline__ int g_off(int row, int chunk) {
    return row * 64 + ((chunk ^ ((0 - (row >> 2)) & 3)) << 4);
}
struct GFrag {
    u32x4 hi, lo;
};
__device__ __forceinline__ f32x16 g_mfma3(const GFrag& a, const GFrag& b, f32x16 c) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a.hi), al = __builtin_bit_cast(bf16x8, a.lo);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, b.hi), bl = __builtin_bit_cast(bf16x8, b.lo);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
    return c;
}

__device__ __forceinline__ f32x4 g_mfma3s(const GFrag& a, const GFrag& b, f32x4 c) {  // the same three products on 16x16x32 tiles
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a.hi), al = __builtin_bit_cast(bf16x8, a.lo);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, b.hi), bl = __builtin_bit_cast(bf16x8, b.lo);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    return c;
}

// PREC = 1 ("fp16x2", opt-in: haet_conv_project_f16_fwd): the activations as fp16 hi + fp16 lo, the weight as ONE fp16
// plane -- two v_mfma_f32_32x32x16_f16 per product instead of three bf16 ones.  What it costs: the weight's 11-bit
// mantissa, ~2e-4 relative on a K = 256 .. 6912 contraction (measured; bf16x3: 4e-6) against north_star's 1e-3 bar and
// the reference's own fp16-autocast GPU path; BASELINE config C5 names "fp16 MFMA".
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ unsigned g_pack_h(float a, float b) {  // v_cvt_pk_f16_f32 (RNE)
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ f32x2 g_unpack_h(unsigned h) {
    return __builtin_convertvector(__builtin_bit_cast(f16x2, h), f32x2);
}
__device__ __forceinline__ f32x16 g_mfma2h(const GFrag& a, const u32x4 b, f32x16 c) {
    const f16x8 ah = __builtin_bit_cast(f16x8, a.hi), al = __builtin_bit_cast(f16x8, a.lo), bh = __builtin_bit_cast(f16x8, b);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
    return c;
}

__device__ __forceinline__ f32x4 g_mfma2hs(const GFrag& a, const u32x4 b, f32x4 c) {  // fp16x2 on 16x16x32 tiles
    const f16x8 ah = __builtin_bit_cast(f16x8, a.hi), al = __builtin_bit_cast(f16x8, a.lo), bh = __builtin_bit_cast(f16x8, b);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
    return c;
}

// optional epilogue activation (ACT = 1): nn.GELU(approximate='none'), the MLPs of the Transolver block
__device__ __forceinline__ float g_gelu(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// fp32 -> bf16 hi / lo planes (weights; once per parameter version)
__global__ void split_bf16_kernel(const float* __restrict__ w, size_t n, unsigned short* __restrict__ hi,
                                  unsigned short* __restrict__ lo) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < n) {
        const float a = w[2 * i], b = w[2 * i + 1];
        const unsigned h = g_pack(a, b);
        const unsigned l = g_pack(a - __uint_as_float(h << 16), b - __uint_as_float(h & 0xffff0000u));
        reinterpret_cast<unsigned*>(hi)[i] = h;
        reinterpret_cast<unsigned*>(lo)[i] = l;
    } else if (2 * i < n) {
        const float a = w[2 * i];
        const unsigned h = g_pack(a, 0.f);
        hi[2 * i] = (unsigned short)h;
        lo[2 * i] = (unsigned short)g_pack(a - __uint_as_float(h << 16), 0.f);
    }
}

// The same planes [O][K], each FOLLOWED by a tile-ordered copy of itself: 16-B pieces (8 consecutive k of one row) stored
// [row / 16][k / 8][row % 16].  The kernels that keep W STATIONARY in registers (gemm_ws_kernel, deslice_out_kernel) fetch
// one such piece per lane and k-step for 16 or 32 different rows per wave instruction: from the row-major plane that is
// 16-32 separate 16-64-B runs per instruction (~11 B/cycle per CU: 256 KB per workgroup, the 16-23 us of fixed cost
// tools/gemm_fixed_cost_probe.py measures), from this copy runs of 512 B - 1 KB.
__global__ void split_bf16_tiled_kernel(const float* __restrict__ w, int O, int K, unsigned short* __restrict__ hi,
                                        unsigned short* __restrict__ lo) {
    const int kch = K >> 3;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // chunk (row o, k-chunk kc)
    if (i >= (size_t)O * kch) return;
    const int o = (int)(i / kch), kc = (int)(i - (size_t)o * kch);
    const f32x4 a = *reinterpret_cast<const f32x4*>(w + (size_t)o * K + 8 * kc);
    const f32x4 b = *reinterpret_cast<const f32x4*>(w + (size_t)o * K + 8 * kc + 4);
    u32x4 h, l;
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = g_pack(v[2 * j], v[2 * j + 1]);
        l[j] = g_pack(v[2 * j] - __uint_as_float(h[j] << 16), v[2 * j + 1] - __uint_as_float(h[j] & 0xffff0000u));
    }
    const size_t plane = (size_t)O * K;
    const size_t rowmajor = (size_t)o * K + 8 * kc;
    const size_t tiled = plane + ((((size_t)(o >> 4) * kch + kc) << 4) + (o & 15)) * 8;
    *reinterpret_cast<u32x4*>(hi + rowmajor) = h;
    *reinterpret_cast<u32x4*>(lo + rowmajor) = l;
    *reinterpret_cast<u32x4*>(hi + tiled) = h;
    *reinterpret_cast<u32x4*>(lo + tiled) = l;
}

// LDS by byte offset only (generic pointers to LDS turn into flat_* accesses, which wait for
// the global prefetch too).
extern __shared__ __attribute__((aligned(16))) char gsm[];

__device__ __forceinline__ GFrag g_read(int img, int row, int chunk) {
    GFrag f;
    const int off = img + g_off(row, chunk);
    f.hi = *reinterpret_cast<const u32x4*>(gsm + off);
    f.lo = *reinterpret_cast<const u32x4*>(gsm + off + GIMG);
    return f;
}

// NCB = BN / 32 column blocks per workgroup (8 -> BN = 256, 4 -> BN = 128).
// Persistent workgroups (one per CU: 8 waves, ~170 VGPRs) walk a flattened stream of
// (row tile, k-tile) items.  The x loads of item i+3 and the W loads of item i+2 are issued into
// named register sets while item i is multiplied, so ~48 KB of x per CU are always in flight
// (HBM latency x per-CU bandwidth ~ 40-50 KB) and the stream never drains at a row-tile boundary.
// CONV: the same GEMM as an implicit-GEMM convolution over a channels-last grid (the Conv2d / Conv3d
// in_project_x of the structured-mesh modules, StructuredMesh2D.py:173-189, StructuredMesh3D.py:177-193):
// row n of A for k-tile (tap, channel block) is the row of the neighbour n + tap offset, or zeros
// outside the grid; W is [O][C_in / 32][taps][32], K = taps * C_in.  x stays [B, N, C_in]: no NCHW round trip.
struct ConvGeom {
    int Cin, Npts, gH, gW, gD, kH, kW, kD;
};
// ACT == 4 (CONV): the output leaves in the streaming passes' operand format -- 16-B cells {4 x bf16 hi, 4 x bf16 lo} per
// 4 channels + s[n][h] = log2(e) / tau(n, h) (Ada-Temp over each 32-channel head, StructuredMesh2D.py:177-181,
// StructuredMesh3D.py:181-185) -- exactly what gemm_ws_kernel<SPLIT> writes for the Linear projection
struct ConvSplit {
    const float* wt;  // ada_temp_linear.weight [32]
    float bt, base_temp;
    float* s_out;     // [Nrows][O / 32]
};
template <int NCB, int RB, bool CONV = false, int ACT = 0, int PREC = 0>  // RB = 32-row blocks per wave: workgroup tile = 64*RB rows x 32*NCB columns
__global__ __launch_bounds__(512) void gemm_x3_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ Whi,
    const unsigned short* __restrict__ Wlo, const float* __restrict__ bias, int Nrows, int K, int O,
    int n_row_tiles, float* __restrict__ out, ConvGeom cg, ConvSplit cs) {
    constexpr int BN = 32 * NCB;
    constexpr int WCB = NCB / 4;  // column blocks per wave (4 waves across the columns)
    constexpr int WPT = NCB / 4;  // 16-B W chunks per thread per plane and k-tile
    constexpr int BM = 64 * RB;   // rows per workgroup tile
    constexpr int XPT = BM / 64;  // 16-B x chunks per thread and k-tile
    constexpr int XIMG = 0, WIMG = (BM / 32) * 2 * GIMG;
    constexpr int BUF = (BM / 32 + NCB) * 2 * GIMG;  // one (x tile, W tile) LDS buffer; two of them
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 2, wn = wave & 3;
    const int col0 = blockIdx.y * BN;
    // ACT == 3 (CONV only): the k-tiles are split over gridDim.z workgroups per output tile, which ADD their partial sums
    // into a zeroed `out` with float atomics (the bias goes in with z = 0).  For grids of a few thousand points (C1: 64 x 64)
    // the launch is otherwise ~30 workgroups walking 72 items one after the other: pure latency.
    constexpr bool kSplitK = ACT == 3;
    const int nkt = kSplitK ? (K / GBK) / (int)gridDim.z : K / GBK;
    const int kt0 = kSplitK ? (int)blockIdx.z * nkt : 0;
    const int my_tiles = n_row_tiles > (int)blockIdx.x ? (n_row_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int total = my_tiles * nkt;
    if (total == 0) return;

    // loader geometry: x tile = BM rows x 8 chunks(16 B) -> XPT chunks per thread (rows +64 apart)
    const int xrow0 = tid >> 3, xq = tid & 7;
    // W chunk i: row + 128 i.  Split-bf16 (PREC 0): from the TILE-ORDERED copy behind each plane (split_bf16_tiled_kernel:
    // pieces at [row / 16][k / 8][row % 16]) with lane -> (row % 16, chunk), so that a wave's 64 pieces of a k-tile are ONE run
    // of 1 KB instead of 16 runs of 64 B; the fp16 plane of PREC 1 has no such copy: rows of 64 B as before
    const int wrow0 = PREC == 0 ? 16 * wave + (lane & 15) : tid >> 2, wch = PREC == 0 ? lane >> 4 : tid & 3;
    const int x_st0 = XIMG + (xrow0 >> 5) * (2 * GIMG) + g_off(xrow0 & 31, xq >> 1) + ((xq & 1) << 3);
    int w_st[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i)
        w_st[i] = WIMG + ((wrow0 + 128 * i) >> 5) * (2 * GIMG) + g_off(wrow0 & 31, wch);

    // Accumulators: every 32 x 32 block (i, j) of the wave's output as four 16 x 16 tiles of v_mfma_f32_16x16x32_bf16 --
    // acc[i][j][in][jo][r] = out[row 32 i + 16 in + 4 kg + r][column 32 j + 16 jo + lane % 16].  Same matrix-pipe cycles per
    // flop as 32x32x16, same LDS reads (a 32-wide k-tile is ONE k-step), but the chip holds a higher clock under this shape in
    // an MFMA-bound loop fed from LDS (MI355X_MICROARCH.md, 'DVFS give-back' item 7).
    const int l15 = lane & 15, kg = lane >> 4;
    f32x4 acc[RB][WCB][2][2];
    float wt4[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // ACT == 4: Ada-Temp weights of this lane's four channels (per jo) after the quad transpose
    if (ACT == 4) {
#pragma unroll
        for (int jo = 0; jo < 2; ++jo)
#pragma unroll
            for (int q = 0; q < 4; ++q) wt4[jo][q] = cs.wt[16 * jo + (l15 & ~3) + q];
    }
    f32x16 accL[RB == 4 ? RB : 1][WCB];  // (RB == 4 only, see compute32)
    float bvL[WCB];
#pragma unroll
    for (int j = 0; j < WCB; ++j) bvL[j] = (bias && (!kSplitK || blockIdx.z == 0)) ? bias[col0 + 32 * (WCB * wn + j) + l31] : 0.f;
    if (RB == 4) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < WCB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accL[RB == 4 ? i : 0][j][r] = bvL[j];
    }
    float bv[WCB][2];  // bias of this lane's output columns: loaded (and waited for) once, up front
#pragma unroll
    for (int j = 0; j < WCB; ++j)
#pragma unroll
        for (int jo = 0; jo < 2; ++jo)
            bv[j][jo] = (bias && (!kSplitK || blockIdx.z == 0)) ? bias[col0 + 32 * (WCB * wn + j) + 16 * jo + l15] : 0.f;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < WCB; ++j)
#pragma unroll
            for (int in = 0; in < 2; ++in)
#pragma unroll
                for (int jo = 0; jo < 2; ++jo) acc[i][j][in][jo] = f32x4{bv[j][jo], bv[j][jo], bv[j][jo], bv[j][jo]};  // accumulate on top of the bias

    // fetch streams (x three items ahead, W two items ahead), each with its own (tile, kt) cursor
    f32x4 xs0[XPT], xs1[XPT], xs2[XPT];
    u32x4 ws0[2 * WPT], ws1[2 * WPT];
    // Every workgroup walks K in a rotated order (start at k-tile blockIdx % nkt): with all CUs on the
    // same k-tile at the same time, every access hits bytes [128 kt, 128 kt + 128) of a 1 KB row,
    // i.e. the same 1/8 of the address-interleaved HBM channels.
    // (not for CONV: there W -- taps * C_in * O, 14 MB at 3x3x3 x 256 x 256 -- is far larger than an XCD's
    // L2, and only workgroups that walk the k-tiles TOGETHER find each other's W tile still cached)
    const int krot = CONV ? 0 : blockIdx.x % nkt;
    int fx_tile = blockIdx.x, fx_kt = 0;
    int fw_kt = 0;
    int cgi[XPT], cgj[XPT], cgk[XPT], cg_tile = -1;  // CONV: grid coordinates of this thread's rows
    auto fetch_x = [&](f32x4 (&d)[XPT]) {
        {   // unconditional (rows are clamped, ghost items past the end load valid memory): a conditional
            // load inside the prefetch ring costs the waitcnt pass its counted waits
            int kk = fx_kt + krot;
            kk = ((kk >= nkt ? kk - nkt : kk) + kt0) * GBK;
            if (!CONV) {
#pragma unroll
                for (int i = 0; i < XPT; ++i) {
                    const int r = min(fx_tile * BM + xrow0 + 64 * i, Nrows - 1);
                    d[i] = *reinterpret_cast<const f32x4*>(x + (size_t)r * K + kk + 4 * xq);
                }
            } else {
                if (cg_tile != fx_tile) {  // once per row tile
                    cg_tile = fx_tile;
#pragma unroll
                    for (int i = 0; i < XPT; ++i) {
                        const int r = min(fx_tile * BM + xrow0 + 64 * i, Nrows - 1);
                        const int p = r % cg.Npts, ij = p / cg.gD;
                        cgk[i] = p - ij * cg.gD;
                        cgi[i] = ij / cg.gW;
                        cgj[i] = ij - cgi[i] * cg.gW;
                    }
                }
                // K is ordered [channel block][tap][32 channels]: the taps of one 128-B channel block follow each
                // other, so rows shifted by one point re-read lines that are still in L1 / L2 (tap-major order
                // re-fetched every tap's rows from the memory side: 27x read amplification in 3-D)
                const int taps = cg.kH * cg.kW * cg.kD, kt32 = kk / GBK;
                const int cb = kt32 / taps, tap = kt32 - cb * taps, c0 = cb * GBK;  // wave-uniform
                const int a = tap / (cg.kW * cg.kD), bc = tap - a * (cg.kW * cg.kD), b = bc / cg.kD, c = bc - b * cg.kD;
                const int da = a - cg.kH / 2, db = b - cg.kW / 2, dc = c - cg.kD / 2;
                const int shift = (da * cg.gW + db) * cg.gD + dc;
#pragma unroll
                for (int i = 0; i < XPT; ++i) {
                    const int r = min(fx_tile * BM + xrow0 + 64 * i, Nrows - 1);
                    const bool ok = (unsigned)(cgi[i] + da) < (unsigned)cg.gH && (unsigned)(cgj[i] + db) < (unsigned)cg.gW &&
                                    (unsigned)(cgk[i] + dc) < (unsigned)cg.gD;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)(ok ? r + shift : r) * cg.Cin + c0 + 4 * xq);
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    d[i] = ok ? v : z;  // zero padding; the load itself is unconditional
                }
            }
        }
        if (++fx_kt == nkt) {
            fx_kt = 0;
            fx_tile += gridDim.x;
        }
    };
    auto fetch_w = [&](u32x4 (&d)[2 * WPT]) {
        {
            int kk = fw_kt + krot;
            kk = ((kk >= nkt ? kk - nkt : kk) + kt0) * GBK;
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                const size_t row = (size_t)(col0 + wrow0 + 128 * i);
                const size_t o = PREC == 0 ? (size_t)O * K + ((((row >> 4) * (size_t)(K >> 3) + (kk >> 3) + wch) << 4) + (row & 15)) * 8
                                           : row * K + kk + 8 * wch;
                d[2 * i] = *reinterpret_cast<const u32x4*>(Whi + o);
                if (PREC == 0) d[2 * i + 1] = *reinterpret_cast<const u32x4*>(Wlo + o);  // (fp16x2: one weight plane)
            }
        }
        if (++fw_kt == nkt) fw_kt = 0;
    };
    auto stage = [&](int buf, const f32x4 (&xv)[XPT], const u32x4 (&wv)[2 * WPT]) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const f32x4 v = xv[i];
            u32x2 hi, lo;
            if (PREC == 0) {
                unsigned h01, l01, h23, l23;
                split_bf16_pair(v[0], v[1], h01, l01);
                split_bf16_pair(v[2], v[3], h23, l23);
                hi = u32x2{h01, h23};
                lo = u32x2{l01, l23};
            } else {
                const unsigned h01 = g_pack_h(v[0], v[1]), h23 = g_pack_h(v[2], v[3]);
                const f32x2 f01 = g_unpack_h(h01), f23 = g_unpack_h(h23);
                hi = u32x2{h01, h23};
                lo = u32x2{g_pack_h(v[0] - f01[0], v[1] - f01[1]), g_pack_h(v[2] - f23[0], v[3] - f23[1])};
            }
            const int off = buf + x_st0 + i * 2 * (2 * GIMG);  // row + 64 i: two row blocks further
            *reinterpret_cast<u32x2*>(gsm + off) = hi;
            *reinterpret_cast<u32x2*>(gsm + off + GIMG) = lo;
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            *reinterpret_cast<u32x4*>(gsm + buf + w_st[i]) = wv[2 * i];
            if (PREC == 0) *reinterpret_cast<u32x4*>(gsm + buf + w_st[i] + GIMG) = wv[2 * i + 1];
        }
    };
    int c_tile = blockIdx.x, c_kt = 0;
    auto compute32 = [&](int buf) __attribute__((always_inline)) {  // RB == 4 (256-row tiles): the 32x32x16 form -- eight B fragments of the 16 x 16 form do not fit beside 128 accumulators
        if constexpr (RB == 4) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            GFrag b[WCB];
#pragma unroll
            for (int j = 0; j < WCB; ++j) {
                if (PREC == 0) b[j] = g_read(buf + WIMG + (WCB * wn + j) * (2 * GIMG), l31, hf + 2 * ks);
                else b[j].hi = *reinterpret_cast<const u32x4*>(gsm + buf + WIMG + (WCB * wn + j) * (2 * GIMG) + g_off(l31, hf + 2 * ks));
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {  // one A fragment at a time: RB = 4 leaves no room for four
                const GFrag a = g_read(buf + XIMG + (RB * wm + i) * (2 * GIMG), l31, hf + 2 * ks);
#pragma unroll
                for (int j = 0; j < WCB; ++j) accL[i][j] = PREC == 0 ? g_mfma3(a, b[j], accL[i][j]) : g_mfma2h(a, b[j].hi, accL[i][j]);
            }
        }
        if (++c_kt == nkt) {  // row tile finished: lane = output column, registers = rows
            const int row0 = c_tile * BM;
            const bool full = row0 + BM <= Nrows;  // wave-uniform: no per-store branch (a branch per
            // store made hipcc wait vmcnt(0) -- prefetch stream included -- before every store)
#pragma unroll
            for (int j = 0; j < WCB; ++j) {
                float* oc = out + (size_t)(row0 + 32 * RB * wm + 4 * hf) * O + col0 + 32 * (WCB * wn + j) + l31;
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    if (ACT == 4) {
                        // (once per row tile, i.e. per K / 32 k-tiles: per-row branches around the stores cost nothing here)
                        const int cb = WCB * wn + j;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            // registers 4g .. 4g+3 = rows 8g + 4hf + (0..3) of this lane's column; a 4 x 4 transpose across
                            // the lane quad turns them into ONE row (8g + 4hf + lane % 4) and four consecutive columns
                            int v[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] = __float_as_int(accL[i][j][4 * g + q]);
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
                            const u32x4 o4 = {h01, h23, l01, l23};
                            float t = f0 * wt4[0][0];
                            t = fmaf(f1, wt4[0][1], t);
                            t = fmaf(f2, wt4[0][2], t);
                            t = fmaf(f3, wt4[0][3], t);
                            t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x124, 0xF, 0xF, false));  // row_ror:4
                            t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x128, 0xF, 0xF, false));  // row_ror:8
                            t += __shfl_xor(t, 16);
                            const float tau = cs.base_temp + fminf(fmaxf(t + cs.bt, -0.4f), 0.4f);
                            const long long rowi = (long long)row0 + 32 * (RB * wm + i) + 8 * g + 4 * hf + (lane & 3);
                            if (rowi < Nrows) {
                                *reinterpret_cast<u32x4*>(out + rowi * O + col0 + 32 * cb + (l31 & ~3)) = o4;
                                cs.s_out[rowi * (O / 32) + col0 / 32 + cb] = 1.4426950408889634f * __builtin_amdgcn_rcpf(tau);
                            }
                        }
                    } else if (kSplitK) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rr = 32 * i + (r & 3) + 8 * (r >> 2);
                            if (row0 + 32 * RB * wm + 4 * hf + rr < Nrows) atomicAdd(oc + (size_t)rr * O, accL[i][j][r]);
                        }
                    } else if (full) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            oc[(size_t)(32 * i + (r & 3) + 8 * (r >> 2)) * O] = ACT == 1 ? g_gelu(accL[i][j][r]) : accL[i][j][r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rr = 32 * i + (r & 3) + 8 * (r >> 2);
                            if (row0 + 32 * RB * wm + 4 * hf + rr < Nrows)
                                oc[(size_t)rr * O] = ACT == 1 ? g_gelu(accL[i][j][r]) : accL[i][j][r];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) accL[i][j][r] = bvL[j];
                }
            }
            c_kt = 0;
            c_tile += gridDim.x;
        }
        }
    };


    auto compute16 = [&](int buf) __attribute__((always_inline)) {
        {
            // B fragments (W): column 16 jo + lane % 16 of block j, k = 8 kg .. 8 kg + 7 of the 32-wide k-tile
            GFrag b[WCB][2];
#pragma unroll
            for (int j = 0; j < WCB; ++j)
#pragma unroll
                for (int jo = 0; jo < 2; ++jo) {
                    const int wi = buf + WIMG + (WCB * wn + j) * (2 * GIMG);
                    if (PREC == 0) b[j][jo] = g_read(wi, 16 * jo + l15, kg);
                    else b[j][jo].hi = *reinterpret_cast<const u32x4*>(gsm + wi + g_off(16 * jo + l15, kg));
                }
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int in = 0; in < 2; ++in) {  // one A fragment at a time
                    const GFrag a = g_read(buf + XIMG + (RB * wm + i) * (2 * GIMG), 16 * in + l15, kg);
#pragma unroll
                    for (int j = 0; j < WCB; ++j)
#pragma unroll
                        for (int jo = 0; jo < 2; ++jo)
                            acc[i][j][in][jo] = PREC == 0 ? g_mfma3s(a, b[j][jo], acc[i][j][in][jo]) : g_mfma2hs(a, b[j][jo].hi, acc[i][j][in][jo]);
                }
        }
        if (++c_kt == nkt) {  // row tile finished: lane = output column (two per block), registers = four consecutive rows
            const int row0 = c_tile * BM;
            const bool full = row0 + BM <= Nrows;  // wave-uniform: no per-store branch (a branch per
            // store made hipcc wait vmcnt(0) -- prefetch stream included -- before every store)
#pragma unroll
            for (int j = 0; j < WCB; ++j) {
                const int cb = WCB * wn + j;
#pragma unroll
                for (int i = 0; i < RB; ++i) {
#pragma unroll
                    for (int in = 0; in < 2; ++in) {
                        const long long rbase = (long long)row0 + 32 * (RB * wm + i) + 16 * in + 4 * kg;  // first of this lane's 4 rows
                        if (ACT == 4) {
                            // (once per row tile, i.e. per K / 32 k-tiles: per-row branches around the stores cost nothing here)
                            // A 4 x 4 transpose across the lane quad turns a tile's four rows of ONE column into ONE row
                            // (rbase + lane % 4) and four consecutive columns = one 16-B cell {4 x hi, 4 x lo}; the Ada-Temp dot over
                            // the head's 32 channels = both column halves of this lane + the other three quads of its 16-lane row.
                            const long long rowi = rbase + (lane & 3);
                            float t = 0.f;
                            u32x4 o4[2];
#pragma unroll
                            for (int jo = 0; jo < 2; ++jo) {
                                int v[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q) v[q] = __float_as_int(acc[i][j][in][jo][q]);
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
                                t = fmaf(f0, wt4[jo][0], t);
                                t = fmaf(f1, wt4[jo][1], t);
                                t = fmaf(f2, wt4[jo][2], t);
                                t = fmaf(f3, wt4[jo][3], t);
                            }
                            t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x124, 0xF, 0xF, false));  // row_ror:4
                            t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x128, 0xF, 0xF, false));  // row_ror:8
                            const float tau = cs.base_temp + fminf(fmaxf(t + cs.bt, -0.4f), 0.4f);
                            if (rowi < Nrows) {
#pragma unroll
                                for (int jo = 0; jo < 2; ++jo)
                                    *reinterpret_cast<u32x4*>(out + rowi * O + col0 + 32 * cb + 16 * jo + (l15 & ~3)) = o4[jo];
                                cs.s_out[rowi * (O / 32) + col0 / 32 + cb] = 1.4426950408889634f * __builtin_amdgcn_rcpf(tau);
                            }
                        } else {
#pragma unroll
                            for (int jo = 0; jo < 2; ++jo) {
                                float* oc = out + rbase * O + col0 + 32 * cb + 16 * jo + l15;
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const float v = acc[i][j][in][jo][r];
                                    if (kSplitK) {
                                        if (rbase + r < Nrows) atomicAdd(oc + (size_t)r * O, v);
                                    } else if (full) {
                                        oc[(size_t)r * O] = ACT == 1 ? g_gelu(v) : v;
                                    } else if (rbase + r < Nrows) {
                                        oc[(size_t)r * O] = ACT == 1 ? g_gelu(v) : v;
                                    }
                                }
                            }
                        }
#pragma unroll
                        for (int jo = 0; jo < 2; ++jo) acc[i][j][in][jo] = f32x4{bv[j][jo], bv[j][jo], bv[j][jo], bv[j][jo]};
                    }
                }
            }
            c_kt = 0;
            c_tile += gridDim.x;
        }
    };

    auto compute = [&](int buf) __attribute__((always_inline)) {
        if constexpr (RB == 4) compute32(buf);
        else compute16(buf);
    };

    // prologue: x items 0,1,2 and W items 0,1 in flight; item 0 staged; then x item 3 / W item 2
    // (the 256-row variant has 128 accumulator registers: it keeps ONE W set, i.e. W runs one item
    // ahead instead of two -- W comes from L2, and an item is 2x longer there)
    constexpr bool kTwoW = RB < 4, kThreeX = RB < 4;  // 256-row variant: x two items ahead, W one
    u32x4 (&wsB)[2 * WPT] = kTwoW ? ws1 : ws0;
    f32x4 (&x_s0)[XPT] = xs1;
    f32x4 (&x_s1)[XPT] = kThreeX ? xs2 : xs0;
    f32x4 (&x_s2)[XPT] = kThreeX ? xs0 : xs1;
    f32x4 (&x_s3)[XPT] = kThreeX ? xs1 : xs0;
    f32x4 (&x_s4)[XPT] = kThreeX ? xs2 : xs1;
    f32x4 (&x_s5)[XPT] = xs0;
    fetch_x(xs0);
    fetch_x(xs1);
    if (kThreeX) fetch_x(xs2);
    fetch_w(ws0);
    if (kTwoW) fetch_w(ws1);
    stage(0, xs0, ws0);
    fetch_x(xs0);
    fetch_w(ws0);
    __syncthreads();
    // steady state, unrolled by 6 so that the register sets are statically named.  Two LDS buffers,
    // ONE barrier per item: item it+1 is staged (from x set (it+1)%3 / W set (it+1)%2, which are then
    // refilled with items it+4 / it+3) into the other buffer BEFORE item it is multiplied, so the
    // two waves of a SIMD overlap one's staging VALU with the other's MFMAs.
#define GEMM_STEP(XS, WS)                                \
    if (it < total) {                                    \
        if (it + 1 < total) {                            \
            stage(((it + 1) & 1) * BUF, XS, WS);         \
            fetch_x(XS);                                 \
            fetch_w(WS);                                 \
        }                                                \
        compute((it & 1) * BUF);                         \
        __syncthreads();                                 \
        ++it;                                            \
    }
    for (int it = 0; it < total;) {
        GEMM_STEP(x_s0, wsB)
        GEMM_STEP(x_s1, ws0)
        GEMM_STEP(x_s2, wsB)
        GEMM_STEP(x_s3, ws0)
        GEMM_STEP(x_s4, wsB)
        GEMM_STEP(x_s5, ws0)
    }
#undef GEMM_STEP
}


// ---------------------------------------------------------------------------------------------
// W-stationary variant (K = 128 / 256, O % 256 == 0): the whole split weight lives in REGISTERS.
// Each of the 8 waves of a (persistent, one per CU) workgroup owns 32 output columns and keeps
// their W rows -- K x 32 x {hi, lo} bf16 = K/2 VGPRs per lane, 128 at K = 256 -- as ready-made
// MFMA B fragments for the lifetime of the kernel (gfx950: 512 registers per SIMD lane, 2 waves
// per SIMD -> 256 each).  What streams is x only, in 64-row tiles:
//   * x is read in whole rows -- one wave instruction = 1 KB contiguous -- converted fp32 ->
//     {hi, lo} bf16 by the loader and parked in LDS as the [32 rows][32 k] images of ALL k-items of
//     the tile (64 KB; two tiles resident), one barrier per TILE,
//   * no W traffic at all (the streaming kernel above re-stages 32 KB of W per 32 KB of x and is
//     ~15 % slower; it stays for the other shapes),
//   * stores and loads share ONE in-order counter on gfx9 (vmcnt): a wait for a prefetched load
//     that is older than a burst of stores asks for the stores to drain.  So the epilogue is spread
//     out (see run_tile), the loop is fully unrolled with all fetches unconditional, and the
//     prologue imitates the steady-state operation pattern: every wait in the loop is vmcnt(8).
// Budget at 1M x 256 x 256 (measured on this part): HBM at copy rate 0.40 ms; the 3 x 12M MFMAs
// 0.23 ms (random operands clock the matrix cores at ~1.65 GHz; 0.157 ms is the 2.4 GHz figure);
// this kernel 0.52-0.55 ms stand-alone.  Prefetch depth (S = 2/4/8), LDS fragment look-ahead and
// store width were each varied without effect; what is left is the overlap of a single
// barrier-coupled workgroup per CU, the price of 128 registers of W per wave.
#ifdef HAET_STAMPS
// Diagnostic build only (make EXTRA=-DHAET_STAMPS; tools/gemm_stamps.py): s_memtime along one tile
// of one workgroup, waves 0 and 4 (same SIMD).  Never quote this build's run time.
__device__ unsigned long long haet_gemm_stamps[64];
#define GSTAMP(i) gst[i] = __builtin_amdgcn_s_memtime()
#else
#define GSTAMP(i) do { } while (0)
#endif
constexpr int WS_IMGP = 2 * GIMG + 32;  // {hi, lo} image pair + 8 banks of skew between k-items
// SPLIT (the in_project_x of the C = 32 fast path): the output goes out in the streaming passes' own
// operand format instead of fp32 -- per 4 channels 16 B = {4 x bf16 hi, 4 x bf16 lo}, same row stride as
// fp32 -- together with s[n][h] = log2(e) / tau(n, h) (Ada-Temp, IrregularMesh.py:168), so the slice
// and de-slice loaders only copy.  They are bound by vector issue; the GEMM has the slack.
// Sum over the LPR (64 or 32) consecutive lanes that hold one x row, in every lane, without LDS-crossbar
// operations (the loader runs inside the counted-wait loop): rotations inside the 16-lane DPP rows, then the four
// row sums through scalar registers.
template <int LPR>
__device__ __forceinline__ float row_lanes_sum(float v, int hf) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));  // row_ror:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false));  // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xF, 0xF, false));  // row_ror:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xF, 0xF, false));  // row_ror:1
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    if (LPR == 64) return (r0 + r1) + (r2 + r3);
    return hf ? r2 + r3 : r0 + r1;
}

struct SplitArgs {
    const float* wt;  // ada_temp_linear.weight [32]
    float bt, base_temp;
    float* s_out;     // [Nrows][O / 32]
    // LNORM: nn.LayerNorm(K) applied to every x row in the loader (ln_1 in front of in_project_x, ln_2 in
    // front of the block MLP: HAETransolver_Irregular_Mesh.py:125-128) -- the row is in one wave instruction
    // anyway, the normalised activations never exist in HBM
    const float* ln_g;
    const float* ln_b;
    float ln_eps;
};
template <int NKT, int S, bool SPLIT = false, int ACT = 0, int LNORM = 0>  // LNORM: 0 none, 1 LayerNorm, 2 RMSNorm (weight only)
__global__ __launch_bounds__(512) void gemm_ws_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ Whi,
    const unsigned short* __restrict__ Wlo, const float* __restrict__ bias, int Nrows, int O,
    int n_row_tiles, float* __restrict__ out, SplitArgs sa) {
    constexpr int K = NKT * GBK, BM = 64;
    constexpr int RBSZ = NKT * WS_IMGP, TBUF = 2 * RBSZ;  // one 32-row block, one tile
    constexpr int LPR = K / 4, RPI = 64 / LPR;            // lanes per x row, rows per wave load
    static_assert(NKT % S == 0 && S >= 2 && (NKT == 4 || NKT == 8), "prefetch ring must divide the tile");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, l31 = lane & 31;
    const int col = blockIdx.y * 256 + 32 * wave + l31;  // this lane's output column
    const int my_tiles = n_row_tiles > (int)blockIdx.x ? (n_row_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    if (my_tiles == 0) return;

    // stationary B fragments: k-step s (16 wide) -> k = 16 s + 8 hf .. + 8 of row `col` of W
    // (from the TILE-ORDERED copy behind each plane -- split_bf16_tiled_kernel: piece (row, k-chunk 2 s + hf) at
    //  [row / 16][chunk][row % 16]: a wave instruction reads two runs of 512 B instead of 32 pieces of 32 B)
    u32x4 wh[2 * NKT], wl[2 * NKT];
    {
        const size_t piece0 = (((size_t)(col >> 4) * (K / 8) + hf) << 4) + (col & 15);
        const unsigned short* th = Whi + (size_t)O * K + piece0 * 8;
        const unsigned short* tl = Wlo + (size_t)O * K + piece0 * 8;
#pragma unroll
        for (int s = 0; s < 2 * NKT; ++s) {
            wh[s] = *reinterpret_cast<const u32x4*>(th + (size_t)s * 2 * 16 * 8);
            wl[s] = *reinterpret_cast<const u32x4*>(tl + (size_t)s * 2 * 16 * 8);
        }
    }
    const float bv = bias ? bias[col] : 0.f;
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = bv;

    // loader: fetch j of a tile = rows 8 wave + RPI j (+ lane / LPR), 16 B per lane, whole rows.
    // The last (and every ghost) tile is SHIFTED to end at row Nrows instead of being clamped row by
    // row: it recomputes a few rows of its neighbour (same operands, same values) and nothing on
    // the load or store path needs a per-row address.
    const int lrow = lane / LPR, lcol = lane % LPR;  // row within the wave load, 16-B chunk of the row
    const int lkt = lcol >> 3, lq = lcol & 7;        // k-item and 16-B chunk inside the 32-wide item
    auto tile_row0 = [&](int t) {                    // first row of this workgroup's t-th tile
        const long long r = ((long long)blockIdx.x + (long long)t * gridDim.x) * BM;
        return r + BM <= Nrows ? r : (long long)Nrows - BM;
    };
    const unsigned x_voff = (unsigned)(((8 * wave + lrow) * K + 4 * lcol) * 4);  // bytes from the tile base
    auto fetch = [&](const char* tile_base, int j) {  // RPI rows further = exactly 1 KB
        return *reinterpret_cast<const f32x4*>(tile_base + x_voff + 1024 * j);
    };
    // LDS address of fetch j: row = 8 wave + RPI j + lrow; the swizzle term changes once, at j = NKT/2
    int st_off[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 8 * wave + 4 * h + lrow;
        st_off[h] = (row >> 5) * RBSZ + lkt * WS_IMGP + g_off(row & 31, lq >> 1) + ((lq & 1) << 3) - 4 * h * 64;
    }
    f32x4 lng = {1.f, 1.f, 1.f, 1.f}, lnb = {0.f, 0.f, 0.f, 0.f};
    if (LNORM) {
        lng = *reinterpret_cast<const f32x4*>(sa.ln_g + 4 * lcol);
        if (LNORM == 1) lnb = *reinterpret_cast<const f32x4*>(sa.ln_b + 4 * lcol);
    }
    auto stage = [&](int tbuf, int j, const f32x4 vin) {
        f32x4 v = vin;
        if (LNORM == 2) {  // nn.RMSNorm(K) (eps = sa.ln_eps): x * rsqrt(mean(x^2) + eps) * g
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) q = fmaf(v[e], v[e], q);
            const float rstd = rsqrtf(row_lanes_sum<LPR>(q, hf) * (1.0f / K) + sa.ln_eps);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * rstd * lng[e];
        }
        if (LNORM == 1) {  // two-pass statistics over the LPR lanes that hold this row (torch's numerics)
            static_assert(!LNORM || LPR == 64 || LPR == 32, "one or two rows per wave load");
            const float mean = row_lanes_sum<LPR>((v[0] + v[1]) + (v[2] + v[3]), hf) * (1.0f / K);
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] -= mean;
                q = fmaf(v[e], v[e], q);
            }
            const float rstd = rsqrtf(row_lanes_sum<LPR>(q, hf) * (1.0f / K) + sa.ln_eps);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e] * rstd, lng[e], lnb[e]);
        }
        const int off = tbuf + st_off[j >= NKT / 2] + RPI * j * 64;
        unsigned h01, l01, h23, l23;
        split_bf16_pair(v[0], v[1], h01, l01);
        split_bf16_pair(v[2], v[3], h23, l23);
        const u32x2 hi = {h01, h23};
        const u32x2 lo = {l01, l23};
        *reinterpret_cast<u32x2*>(gsm + off) = hi;
        *reinterpret_cast<u32x2*>(gsm + off + GIMG) = lo;
    };
    auto mfma3 = [&](f32x16& c, const GFrag& f, const u32x4 bhv, const u32x4 blv) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bhv), bl = __builtin_bit_cast(bf16x8, blv);
        const bf16x8 fh = __builtin_bit_cast(bf16x8, f.hi), fl = __builtin_bit_cast(bf16x8, f.lo);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, bh, c, 0, 0, 0);
    };

    // Epilogue stores.  An accumulator holds, per lane, ONE column and 4 x 4 consecutive rows; a
    // 4 x 4 transpose across each quad of lanes (two DPP butterflies, 16 VALU per register group)
    // turns that into one row and 4 consecutive columns, so a group goes out as one
    // global_store_dwordx4 covering 8 rows x 128 B instead of four dword stores of 2 rows x 128 B
    // each: the memory pipeline takes ~170 cycles per store INSTRUCTION when every CU is storing.
    const int col4 = blockIdx.y * 256 + 32 * wave + (l31 & ~3);  // first of this lane's 4 output columns
    const unsigned o_voff = (unsigned)(((4 * hf + (lane & 3)) * O + col4) * 4);  // bytes from the tile's first row
    constexpr int GPS = NKT == 8 ? 2 : 1;  // sub-steps per stored register group (4 groups per phase)
    float wt4[4] = {0.f, 0.f, 0.f, 0.f};
    char* const sb = reinterpret_cast<char*>(sa.s_out);
    const unsigned s_voff = (unsigned)(((4 * hf + (lane & 3)) * (O / 32) + blockIdx.y * 8 + wave) * 4);
    if (SPLIT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) wt4[q] = sa.wt[(l31 & ~3) + q];
    }
    auto store_group = [&](long long row0, const f32x16& c, int i, int g) {  // row0: first row of the tile
        char* const ob = reinterpret_cast<char*>(out) + row0 * ((long long)O * 4);
        int v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = __float_as_int(c[4 * g + q]);
        const bool odd = lane & 1, up = lane & 2;
#pragma unroll
        for (int m = 0; m < 4; m += 2) {  // lanes ^ 1: (reg m+1, even lane) <-> (reg m, odd lane)
            const int send = odd ? v[m] : v[m + 1];
            const int recv = __builtin_amdgcn_update_dpp(0, send, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
            v[m] = odd ? recv : v[m];
            v[m + 1] = odd ? v[m + 1] : recv;
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {  // lanes ^ 2: (reg m+2, lanes 0,1) <-> (reg m, lanes 2,3)
            const int send = up ? v[m] : v[m + 2];
            const int recv = __builtin_amdgcn_update_dpp(0, send, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
            v[m] = up ? recv : v[m];
            v[m + 2] = up ? v[m + 2] : recv;
        }
        if (!SPLIT && ACT == 2) {
            // SwiGLU gate in the epilogue (components/mlp.py:17-18): the weight rows arrive INTERLEAVED (row 2j = w1[j],
            // row 2j + 1 = w2[j]), so this lane's four consecutive columns are (u1, u2) of two hidden units and the
            // output H [rows][O / 2] = u2 * silu(u1) is written instead of U: the [rows][2 hid] tensor never exists
            const float a0 = __int_as_float(v[0]), b0 = __int_as_float(v[1]), a1 = __int_as_float(v[2]), b1 = __int_as_float(v[3]);
            const f32x2 h2 = {b0 * a0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * a0)),
                              b1 * a1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * a1))};
            char* const hb = reinterpret_cast<char*>(out) + row0 * ((long long)O * 2);
            *reinterpret_cast<f32x2*>(hb + (o_voff >> 1) + (unsigned)((32 * i + 8 * g) * O * 2)) = h2;
        } else if (!SPLIT) {
            f32x4 o4 = {__int_as_float(v[0]), __int_as_float(v[1]), __int_as_float(v[2]), __int_as_float(v[3])};
            if (ACT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) o4[q] = g_gelu(o4[q]);
            }
            *reinterpret_cast<f32x4*>(ob + o_voff + (unsigned)((32 * i + 8 * g) * O * 4)) = o4;
        } else {
            // this lane: ONE row, four consecutive channels of head `wave` (+ 8 blockIdx.y)
            const float f0 = __int_as_float(v[0]), f1 = __int_as_float(v[1]), f2 = __int_as_float(v[2]), f3 = __int_as_float(v[3]);
            unsigned h01, l01, h23, l23;
            split_bf16_pair(f0, f1, h01, l01);
            split_bf16_pair(f2, f3, h23, l23);
            const u32x4 o4 = {h01, h23, l01, l23};
            *reinterpret_cast<u32x4*>(ob + o_voff + (unsigned)((32 * i + 8 * g) * O * 4)) = o4;
            // Ada-Temp: dot over the head's 32 channels = my 4 + the other 7 quads of my row (lanes with
            // the same lane & 3: rotations by 4 and 8 inside the 16-lane row, then the other row)
            float t = f0 * wt4[0];
            t = fmaf(f1, wt4[1], t);
            t = fmaf(f2, wt4[2], t);
            t = fmaf(f3, wt4[3], t);
            t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x124, 0xF, 0xF, false));  // row_ror:4
            t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x128, 0xF, 0xF, false));  // row_ror:8
            t += __shfl_xor(t, 16);
            const float tau = sa.base_temp + fminf(fmaxf(t + sa.bt, -0.4f), 0.4f);
            // every lane stores (the 8 lanes of a row write the same value to the same word: a
            // conditional store here would cost the prefetch ring its counted waits)
            *reinterpret_cast<float*>(sb + row0 * ((long long)(O / 32) * 4) + s_voff + (unsigned)((32 * i + 8 * g) * (O / 32) * 4)) =
                1.4426950408889634f * __builtin_amdgcn_rcpf(tau);
        }
    };

    // prologue: tile 0 staged, fetches 0 .. S-1 of tile 1 in flight.  Each fetch is followed by the
    // stores that follow it in the steady state (here: the bias into rows of tile 0, which this very
    // lane overwrites later): the waitcnt pass merges the pending-operation state of the loop entry
    // with the back edge's conservatively, and without them every tile started by draining its
    // predecessor's stores (vmcnt(3) where vmcnt(19) would do).
    f32x4 xs[S];
    const char* const xb = reinterpret_cast<const char*>(x);
    {
        const char* b0 = xb + tile_row0(0) * (K * 4);
#pragma unroll
        for (int j = 0; j < NKT; ++j) stage(0, j, fetch(b0, j));
        const char* b1 = xb + tile_row0(1) * (K * 4);
        const long long o0 = tile_row0(0);
#pragma unroll
        for (int i = 0; i < S; ++i) {
            __builtin_amdgcn_sched_barrier(0);  // x is __restrict__: the loads would drift past the stores
            xs[i] = fetch(b1, i);
            __builtin_amdgcn_sched_barrier(0);
            if (GPS == 1) store_group(o0, acc[0], 0, (2 * i) & 3);
            store_group(o0, acc[0], 0, (GPS == 1 ? 2 * i + 1 : i) & 3);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();

    // A tile is walked as two 32-row phases (accumulator 0, then 1), NKT sub-steps of 6 MFMAs each.
    // While one accumulator is on the matrix cores the OTHER, finished one is written out, one
    // register group every GPS sub-steps: a wave issues in order, and a burst of stores at the tile
    // end (the memory system takes a store instruction every ~170 cycles when every CU stores at
    // once) keeps the wave, and the loads queued behind the burst, away from the MFMAs.  No extra
    // registers: the idle accumulator is the store buffer.
    auto run_tile = [&](int t) {
#ifdef HAET_STAMPS
        unsigned long long gst[NKT + 2];
#endif
        GSTAMP(0);
        const int cur = (t & 1) * TBUF, nxt = TBUF - cur;
        const char* b1 = xb + tile_row0(t + 1) * (K * 4);
        const char* b2 = xb + tile_row0(t + 2) * (K * 4);
        // (t = 0: nothing to store yet -- the bias goes to this tile's own rows and is overwritten later)
        const long long o_prev = tile_row0(t ? t - 1 : 0);
        const long long o_cur = tile_row0(t);
        GFrag a = g_read(cur, l31, hf);
#pragma unroll
        for (int u = 0; u < 2 * NKT; ++u) {
            const int rb = u / NKT, kt = u % NKT;
            if (u % 2 == 0) {  // stage fetch u/2 of tile t + 1 from its set, refill the set S fetches on
                const int j = u / 2;
                stage(nxt, j, xs[j % S]);
                xs[j % S] = j + S < NKT ? fetch(b1, j + S) : fetch(b2, j + S - NKT);
            }
            if (kt % GPS == GPS - 1) {
                if (rb == 0) store_group(o_prev, acc[1], 1, kt / GPS);
                else store_group(o_cur, acc[0], 0, kt / GPS);
            }
            const int img = cur + rb * RBSZ + kt * WS_IMGP;
            const GFrag f = g_read(img, l31, hf + 2);
            // sched_barrier: left alone, the scheduler sinks every fragment read to just before its
            // first use (fewest live registers), which is exactly the LDS round trip to hide
            __builtin_amdgcn_sched_barrier(0);
            mfma3(acc[rb], a, wh[2 * kt], wl[2 * kt]);
            __builtin_amdgcn_sched_barrier(0);
            if (u + 1 < 2 * NKT) a = g_read(cur + ((u + 1) / NKT) * RBSZ + ((u + 1) % NKT) * WS_IMGP, l31, hf);
            __builtin_amdgcn_sched_barrier(0);
            mfma3(acc[rb], f, wh[2 * kt + 1], wl[2 * kt + 1]);
            if (kt == NKT - 1) {  // the other accumulator is stored by now: back to the bias
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[1 - rb][r] = bv;
            }
            if (u % 2 == 1) GSTAMP(1 + u / 2);
        }
        __syncthreads();
        GSTAMP(NKT + 1);
#ifdef HAET_STAMPS
        if (blockIdx.x == 5 && t == 10 && (tid & 255) == 0) {
#pragma unroll
            for (int i = 0; i < NKT + 2; ++i) haet_gemm_stamps[(wave >> 2) * 32 + i] = gst[i];
        }
        if (blockIdx.x == 5 && t == 11 && (tid & 255) == 0) haet_gemm_stamps[(wave >> 2) * 32 + NKT + 2] = gst[0];
#endif
    };
    {
        int t = 0;  // do-while: a guarded loop gets a preheader, and the prologue's loads sink into it
        do run_tile(t);
        while (++t < my_tiles);
    }
    {
        const long long ob = tile_row0(my_tiles - 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) store_group(ob, acc[1], 1, g);
    }
}

}  // namespace haet

using namespace haet;

extern "C" int haet_split_bf16(const float* w, size_t n, uint16_t* hi, uint16_t* lo,
                               haet_stream_t stream) {
    if (!w || !hi || !lo || n == 0) return fail(HAET_E_ARG, "split_bf16: bad arguments");
    const size_t pairs = (n + 1) / 2;
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0,
                       as_stream(stream), w, n, hi, lo);
    return check_launch("split_bf16_kernel");
}

extern "C" int haet_split_bf16_tiled(const float* w, int O, int K, uint16_t* hi, uint16_t* lo, haet_stream_t stream) {
    if (!w || !hi || !lo || O < 16 || K < 8 || O % 16 || K % 8)
        return fail(HAET_E_ARG, "split_bf16_tiled: O=%d (multiple of 16) / K=%d (multiple of 8) unsupported or null pointer", O, K);
    const size_t chunks = (size_t)O * (K / 8);
    hipLaunchKernelGGL(split_bf16_tiled_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, as_stream(stream), w,
                       O, K, hi, lo);
    return check_launch("split_bf16_tiled_kernel");
}

#ifdef HAET_STAMPS
extern "C" int haet_debug_read_gemm_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(haet_gemm_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int haet_linear_supported(int K, int O) {
    return (K >= 32 && K % 32 == 0 && O >= 128 && O % 128 == 0) ? 1 : 0;
}

static int linear_fwd_impl(const float* x, const uint16_t* W_hi, const uint16_t* W_lo, const float* bias, int Nrows,
                           int K, int O, int act, float* out, haet_stream_t stream, const float* ln_g = nullptr,
                           const float* ln_b = nullptr, float ln_eps = 0.f) {
    if (!x || !W_hi || !W_lo || !out || Nrows < 1 || (act != 0 && act != 1)) return fail(HAET_E_ARG, "linear_fwd: bad arguments");
    if (!haet_linear_supported(K, O))
        return fail(HAET_E_ARG, "linear_fwd: K=%d (multiple of 32) / O=%d (multiple of 128) unsupported", K, O);
    // W-stationary kernel wherever the weight fits the register file
    static const bool force_stream = getenv("HAET_GEMM_STREAM") != nullptr;
    if (!force_stream && O % 256 == 0 && (K == 128 || K == 256) && Nrows >= 64) {
        const int tiles = (Nrows + 63) / 64;
        const int ny = O / 256;
        const int gx = tiles < 256 / ny ? tiles : (256 / ny > 0 ? 256 / ny : 1);
        const size_t lds = (size_t)2 * 2 * (K / GBK) * WS_IMGP;  // two resident tiles: 129 KB at K = 256
        auto go = [&](auto kern) {
            if (lds > 64 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(gx, ny), dim3(512), lds, as_stream(stream), x, W_hi, W_lo, bias, Nrows,
                               O, tiles, out, SplitArgs{nullptr, 0.f, 0.f, nullptr, ln_g, ln_b, ln_eps});
        };
        if (ln_g) {
            if (K == 256) { if (act) go(gemm_ws_kernel<8, 4, false, 1, true>); else go(gemm_ws_kernel<8, 4, false, 0, true>); }
            else { if (act) go(gemm_ws_kernel<4, 4, false, 1, true>); else go(gemm_ws_kernel<4, 4, false, 0, true>); }
            return check_launch("gemm_ws_kernel<ln>");
        }
        if (K == 256) { if (act) go(gemm_ws_kernel<8, 4, false, 1>); else go(gemm_ws_kernel<8, 4>); }
        else { if (act) go(gemm_ws_kernel<4, 4, false, 1>); else go(gemm_ws_kernel<4, 4>); }
        return check_launch("gemm_ws_kernel");
    }
    if (ln_g) return fail(HAET_E_ARG, "linear_ln_fwd: K=%d O=%d rows=%d outside the W-stationary kernel", K, O, Nrows);
    // 256-row workgroup tiles (W staged once per 256 rows) when there are enough rows to fill the
    // chip, 128-row tiles otherwise; 128-column variant for O % 256 != 0
    const bool big = Nrows >= 256 * 256 && O % 256 == 0;
    const int BM = big ? 256 : 128;
    const int tiles = (Nrows + BM - 1) / BM;
    const int NCBv = (O % 256 == 0) ? 8 : 4;
    const int ny = O / (32 * NCBv);
    const int gx = tiles < 256 / ny ? tiles : (256 / ny > 0 ? 256 / ny : 1);  // one workgroup per CU
    const size_t lds = (size_t)2 * (BM / 32 + NCBv) * 2 * GIMG;
    auto launch = [&](auto kern) -> int {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(HAET_E_LAUNCH, "linear_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(kern, dim3(gx, ny), dim3(512), lds, as_stream(stream), x, W_hi, W_lo, bias, Nrows,
                           K, O, tiles, out, ConvGeom{}, ConvSplit{});
        return HAET_OK;
    };
    int rc;
    if (act) {
        if (big) rc = launch(gemm_x3_kernel<8, 4, false, 1>);
        else if (NCBv == 8) rc = launch(gemm_x3_kernel<8, 2, false, 1>);
        else rc = launch(gemm_x3_kernel<4, 2, false, 1>);
    } else {
        if (big) rc = launch(gemm_x3_kernel<8, 4>);
        else if (NCBv == 8) rc = launch(gemm_x3_kernel<8, 2>);
        else rc = launch(gemm_x3_kernel<4, 2>);
    }
    if (rc != HAET_OK) return rc;
    return check_launch("gemm_x3_kernel");
}

extern "C" int haet_linear_fwd(const float* x, const uint16_t* W_hi, const uint16_t* W_lo, const float* bias,
                               int Nrows, int K, int O, float* out, haet_stream_t stream) {
    return linear_fwd_impl(x, W_hi, W_lo, bias, Nrows, K, O, 0, out, stream);
}

extern "C" int haet_linear_act_fwd(const float* x, const uint16_t* W_hi, const uint16_t* W_lo, const float* bias,
                                   int Nrows, int K, int O, int act, float* out, haet_stream_t stream) {
    return linear_fwd_impl(x, W_hi, W_lo, bias, Nrows, K, O, act, out, stream);
}

// SwiGLU pair (w1 | w2 interleaved row by row, see the kernel's ACT == 2 epilogue): H [Nrows][O / 2] = (x w2^T + b2) silu(x w1^T + b1)
extern "C" int haet_linear_swiglu_supported(int K, int O, int Nrows) { return haet_project_split_supported(K, O, Nrows); }

extern "C" int haet_linear_swiglu_fwd(const float* x, const float* rms_w, float rms_eps, const uint16_t* W_hi, const uint16_t* W_lo,
                                      const float* bias, int Nrows, int K, int O, float* H, haet_stream_t stream) {
    if (!x || !W_hi || !W_lo || !H || Nrows < 1) return fail(HAET_E_ARG, "linear_swiglu_fwd: bad arguments");
    if (!haet_linear_swiglu_supported(K, O, Nrows))
        return fail(HAET_E_ARG, "linear_swiglu_fwd: K=%d O=%d rows=%d outside the W-stationary kernel", K, O, Nrows);
    const int tiles = (Nrows + 63) / 64, ny = O / 256;
    const int gx = tiles < 256 / ny ? tiles : (256 / ny > 0 ? 256 / ny : 1);
    const size_t lds = (size_t)2 * 2 * (K / GBK) * WS_IMGP;
    auto go = [&](auto kern) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(gx, ny), dim3(512), lds, as_stream(stream), x, W_hi, W_lo, bias, Nrows, O, tiles, H,
                           SplitArgs{nullptr, 0.f, 0.f, nullptr, rms_w, nullptr, rms_eps});
    };
    if (rms_w) {  // nn.RMSNorm(K) of the rows in the loader: the normalised rows never exist in HBM
        if (K == 256) go(gemm_ws_kernel<8, 4, false, 2, 2>);
        else go(gemm_ws_kernel<4, 4, false, 2, 2>);
    } else {
        if (K == 256) go(gemm_ws_kernel<8, 4, false, 2>);
        else go(gemm_ws_kernel<4, 4, false, 2>);
    }
    return check_launch("gemm_ws_kernel<swiglu>");
}

// LayerNorm(K) fused into the loader of the W-stationary kernel (same envelope as haet_project_split_fwd)
extern "C" int haet_linear_ln_fwd(const float* x, const float* ln_g, const float* ln_b, float ln_eps,
                                  const uint16_t* W_hi, const uint16_t* W_lo, const float* bias, int Nrows, int K,
                                  int O, int act, float* out, haet_stream_t stream) {
    if (!ln_g || !ln_b) return fail(HAET_E_ARG, "linear_ln_fwd: null LayerNorm parameters");
    return linear_fwd_impl(x, W_hi, W_lo, bias, Nrows, K, O, act, out, stream, ln_g, ln_b, ln_eps);
}

extern "C" int haet_project_split_supported(int K, int O, int Nrows) {
    return (O % 256 == 0 && (K == 128 || K == 256) && Nrows >= 64) ? 1 : 0;
}

static int project_split_impl(const float* x, const uint16_t* W_hi, const uint16_t* W_lo,
                              const float* bias, const float* wt, float bt, float base_temp,
                              int Nrows, int K, int O, void* x_split, float* s_out,
                              haet_stream_t stream, const float* ln_g, const float* ln_b, float ln_eps) {
    if (!x || !W_hi || !W_lo || !wt || !x_split || !s_out)
        return fail(HAET_E_ARG, "project_split_fwd: null pointer");
    if (!haet_project_split_supported(K, O, Nrows))
        return fail(HAET_E_ARG, "project_split_fwd: K=%d O=%d rows=%d outside the W-stationary kernel", K, O, Nrows);
    const int tiles = (Nrows + 63) / 64, ny = O / 256;
    const int gx = tiles < 256 / ny ? tiles : (256 / ny > 0 ? 256 / ny : 1);
    const size_t lds = (size_t)2 * 2 * (K / GBK) * WS_IMGP;
    auto go = [&](auto kern) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(gx, ny), dim3(512), lds, as_stream(stream), x, W_hi, W_lo, bias, Nrows, O,
                           tiles, reinterpret_cast<float*>(x_split), SplitArgs{wt, bt, base_temp, s_out, ln_g, ln_b, ln_eps});
    };
    if (ln_g) {
        if (K == 256) go(gemm_ws_kernel<8, 4, true, 0, true>);
        else go(gemm_ws_kernel<4, 4, true, 0, true>);
        return check_launch("gemm_ws_kernel<split, ln>");
    }
    if (K == 256) go(gemm_ws_kernel<8, 4, true>);
    else go(gemm_ws_kernel<4, 4, true>);
    return check_launch("gemm_ws_kernel<split>");
}

extern "C" int haet_project_split_fwd(const float* x, const uint16_t* W_hi, const uint16_t* W_lo,
                                      const float* bias, const float* wt, float bt, float base_temp,
                                      int Nrows, int K, int O, void* x_split, float* s_out,
                                      haet_stream_t stream) {
    return project_split_impl(x, W_hi, W_lo, bias, wt, bt, base_temp, Nrows, K, O, x_split, s_out, stream, nullptr,
                              nullptr, 0.f);
}

extern "C" int haet_project_split_ln_fwd(const float* x, const float* ln_g, const float* ln_b, float ln_eps,
                                         const uint16_t* W_hi, const uint16_t* W_lo, const float* bias,
                                         const float* wt, float bt, float base_temp, int Nrows, int K, int O,
                                         void* x_split, float* s_out, haet_stream_t stream) {
    if (!ln_g || !ln_b) return fail(HAET_E_ARG, "project_split_ln_fwd: null LayerNorm parameters");
    return project_split_impl(x, W_hi, W_lo, bias, wt, bt, base_temp, Nrows, K, O, x_split, s_out, stream, ln_g, ln_b,
                              ln_eps);
}

// Conv2d / Conv3d projection (kernel k, stride 1, padding k/2) over a channels-last grid, as an
// implicit GEMM on the matrix cores.  W_hi / W_lo: split planes of the weight re-laid as
// [O][C_in / 32][taps][32] (tap order = the reference's kernel index order kH, kW, kD; 2-D: gD = kD = 1).
extern "C" int haet_conv_project_supported(int Cin, int O) {
    return (Cin >= 32 && Cin % 32 == 0 && O >= 128 && O % 128 == 0) ? 1 : 0;
}

static int conv_project_impl(const float* x, const uint16_t* W_hi, const uint16_t* W_lo,
                             const float* bias, int B, int gH, int gW, int gD, int Cin, int O,
                             int kH, int kW, int kD, float* out, haet_stream_t stream, const ConvSplit* split, bool f16 = false,
                             bool fixed_order = false) {
    if (!x || !W_hi || (!W_lo && !f16) || !out) return fail(HAET_E_ARG, "conv_project_fwd: null pointer");
    if (!haet_conv_project_supported(Cin, O))
        return fail(HAET_E_ARG, "conv_project_fwd: C_in=%d (multiple of 32) / O=%d (multiple of 128) unsupported", Cin, O);
    if (B < 1 || gH < 1 || gW < 1 || gD < 1 || kH < 1 || kW < 1 || kD < 1 || !(kH & kW & kD & 1))
        return fail(HAET_E_ARG, "conv_project_fwd: bad grid / kernel (odd kernel sizes only)");
    const long long rows = (long long)B * gH * gW * gD;
    if (rows > 0x7fffffffLL) return fail(HAET_E_ARG, "conv_project_fwd: too many points");
    const int Nrows = (int)rows, K = kH * kW * kD * Cin;
    // 128-row tiles always: with K = taps * C_in in the thousands the kernel is compute-bound and the
    // 256-row variant (one W register set, two x sets: 256 VGPRs) pipelines worse -- 19.5 vs 11.9 ms at 96^3
    const bool big = false;
    const int BM = big ? 256 : 128;
    const int tiles = (Nrows + BM - 1) / BM;
    const int NCBv = (O % 256 == 0) ? 8 : 4;
    const int ny = O / (32 * NCBv);
    const int gx = tiles < 256 / ny ? tiles : (256 / ny > 0 ? 256 / ny : 1);
    const size_t lds = (size_t)2 * (BM / 32 + NCBv) * 2 * GIMG;
    const ConvGeom cg{Cin, gH * gW * gD, gH, gW, gD, kH, kW, kD};
    // few row tiles (small grids): split the k-tiles over Z workgroups per tile (one channel block each) that add into a
    // zeroed output -- 32 workgroups x 72 items become 256 x 9
    const int ncb = Cin / GBK;
    static const bool no_split = getenv("HAET_CONV_NO_SPLITK") != nullptr;
    // (float atomics: the summation order varies from run to run -- not taken in deterministic mode)
    if (!split && !f16 && !no_split && !fixed_order && !deterministic_mode() && !big && tiles * ny * 4 <= 256 && ncb >= 2) {
        int Z = ncb;
        while (Z > 1 && tiles * ny * Z > 512) Z /= 2;
        if (Z > 1 && ncb % Z == 0) {
            if (zero_async(out, (size_t)Nrows * O * sizeof(float), as_stream(stream)) != hipSuccess)  // (a kernel: common.h)
                return fail(HAET_E_LAUNCH, "conv_project_fwd: zeroing the output failed");
            auto launch_z = [&](auto kern) {
                if (lds > 64 * 1024)
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(kern, dim3(tiles, ny, Z), dim3(512), lds, as_stream(stream), x, W_hi, W_lo, bias, Nrows, K, O,
                                   tiles, out, cg, ConvSplit{});
            };
            if (NCBv == 8) launch_z(gemm_x3_kernel<8, 2, true, 3>);
            else launch_z(gemm_x3_kernel<4, 2, true, 3>);
            return check_launch("gemm_x3_kernel<conv, split-k>");
        }
    }
    auto launch = [&](auto kern) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(gx, ny), dim3(512), lds, as_stream(stream), x, W_hi, W_lo, bias, Nrows, K, O,
                           tiles, out, cg, split ? *split : ConvSplit{});
    };
    if (split) {  // cells + s instead of fp32 rows
        if (f16) {
            if (NCBv == 8) launch(gemm_x3_kernel<8, 2, true, 4, 1>);
            else launch(gemm_x3_kernel<4, 2, true, 4, 1>);
        } else {
            if (NCBv == 8) launch(gemm_x3_kernel<8, 2, true, 4>);
            else launch(gemm_x3_kernel<4, 2, true, 4>);
        }
        return check_launch("gemm_x3_kernel<conv, split format>");
    }
    if (f16) {
        if (NCBv == 8) launch(gemm_x3_kernel<8, 2, true, 0, 1>);
        else launch(gemm_x3_kernel<4, 2, true, 0, 1>);
        return check_launch("gemm_x3_kernel<conv, fp16x2>");
    }
    if (NCBv == 8) launch(gemm_x3_kernel<8, 2, true>);  // (the 256-row variant spilled: DESIGN.md 5b; it is not instantiated)
    else launch(gemm_x3_kernel<4, 2, true>);
    return check_launch("gemm_x3_kernel<conv>");
}

extern "C" int haet_conv_project_fwd(const float* x, const uint16_t* W_hi, const uint16_t* W_lo,
                                     const float* bias, int B, int gH, int gW, int gD, int Cin, int O,
                                     int kH, int kW, int kD, float* out, haet_stream_t stream) {
    return conv_project_impl(x, W_hi, W_lo, bias, B, gH, gW, gD, Cin, O, kH, kW, kD, out, stream, nullptr);
}

// The fixed-summation-order variant as a PER-CALL choice (never the split-k / float-atomics path), whatever the
// process-wide haet_set_deterministic says: what a host passes torch.use_deterministic_algorithms(True) through.
extern "C" int haet_conv_project_det_fwd(const float* x, const uint16_t* W_hi, const uint16_t* W_lo,
                                         const float* bias, int B, int gH, int gW, int gD, int Cin, int O,
                                         int kH, int kW, int kD, float* out, haet_stream_t stream) {
    return conv_project_impl(x, W_hi, W_lo, bias, B, gH, gW, gD, Cin, O, kH, kW, kD, out, stream, nullptr, false, true);
}

// fp16x2 arithmetic (opt-in; see PREC above): W_f16 = the weight of haet_conv_project_fwd's layout as ONE fp16 plane
extern "C" int haet_conv_project_f16_fwd(const float* x, const uint16_t* W_f16, const float* bias, int B, int gH, int gW,
                                         int gD, int Cin, int O, int kH, int kW, int kD, float* out, haet_stream_t stream) {
    return conv_project_impl(x, W_f16, nullptr, bias, B, gH, gW, gD, Cin, O, kH, kW, kD, out, stream, nullptr, true);
}
extern "C" int haet_conv_project_split_f16_fwd(const float* x, const uint16_t* W_f16, const float* bias, const float* wt,
                                               float bt, float base_temp, int B, int gH, int gW, int gD, int Cin, int O,
                                               int kH, int kW, int kD, void* cells, float* s_out, haet_stream_t stream) {
    if (!wt || !s_out) return fail(HAET_E_ARG, "conv_project_split_f16_fwd: null pointer");
    const ConvSplit cs{wt, bt, base_temp, s_out};
    return conv_project_impl(x, W_f16, nullptr, bias, B, gH, gW, gD, Cin, O, kH, kW, kD, reinterpret_cast<float*>(cells), stream,
                             &cs, true);
}

// The same convolution with the output in the fused operand format of the streaming passes (haet_project_split_fwd's):
// cells [B*N][O] (4-byte slots: per 4 channels {4 x bf16 hi, 4 x bf16 lo}) + s [B*N][O / 32] = log2(e) / tau, tau =
// base_temp + clamp(x_lin[head] . wt + bt, -0.4, 0.4) per 32-channel head (dim_head = 32).
extern "C" int haet_conv_project_split_fwd(const float* x, const uint16_t* W_hi, const uint16_t* W_lo,
                                           const float* bias, const float* wt, float bt, float base_temp, int B, int gH,
                                           int gW, int gD, int Cin, int O, int kH, int kW, int kD, void* cells,
                                           float* s_out, haet_stream_t stream) {
    if (!wt || !s_out) return fail(HAET_E_ARG, "conv_project_split_fwd: null pointer");
    const ConvSplit cs{wt, bt, base_temp, s_out};
    return conv_project_impl(x, W_hi, W_lo, bias, B, gH, gW, gD, Cin, O, kH, kW, kD, reinterpret_cast<float*>(cells), stream, &cs);
}
