// Device helpers shared by the streaming-pass kernels (slice_mfma.hip: slice / de-slice; deslice_out.hip: de-slice
// fused with the to_out GEMM): split-bf16 fragments, the swizzled [32][32] LDS images, the tile loader.
#pragma once
#include <cstdlib>

#include "common.h"

namespace haet {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int TN = 32;          // points per tile
constexpr int NW = 8;           // waves per workgroup
constexpr int NTHR = NW * 64;
constexpr int kC = 32;          // dim_head of this fast path
constexpr int IMG = 2048 + 32;  // bytes per [32][32] bf16 image (+32: de-phase heads in the banks)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kThr2 = 20.0f;  // deferred-max threshold, log2 units

struct Frag {
    u32x4 hi, lo;
};

// ALL LDS addressing is by byte offset into this one array: a generic pointer to LDS (kept in an
// array like bufs[cur], or passed through a helper) becomes flat_load / flat_store, and a flat
// access waits vmcnt(0) AND lgkmcnt(0) -- i.e. for the NEXT tile's global prefetch as well, which
// serialises the load/compute overlap this kernel is built around.
extern __shared__ __attribute__((aligned(16))) char smem[];

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) { split_bf16_pair(a, b, hi, lo); }  // (common.h)

// 8 fp32 -> split-bf16 operand fragment (element j in bf16 slot j)
__device__ __forceinline__ Frag split8(const float (&v)[8]) {
    Frag f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h, l;
        split_bf16_pair(v[2 * i], v[2 * i + 1], h, l);
        f.hi[i] = h;
        f.lo[i] = l;
    }
    return f;
}

__device__ __forceinline__ f32x16 mfma3(const Frag& a, const Frag& b, f32x16 c) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a.hi), al = __builtin_bit_cast(bf16x8, a.lo);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, b.hi), bl = __builtin_bit_cast(bf16x8, b.lo);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
    return c;
}

// byte offset of (row, 16-B chunk) in a swizzled [32][32] bf16 image
__device__ __forceinline__ int img_off(int row, int chunk) {
    return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// ----------------------------------------------------------------------------- tile loader
// LI = 16-B chunks per thread per tile (32 rows x H*8 chunks / NT threads): ceil(H/2) at 512 threads.
// Everything that does not change from tile to tile is computed once in init(): the row / head of
// each chunk, its LDS offsets, its global pointer (advanced by one tile per fetch).
// CB = dim_head / 32 (round 4: dim_head 64 = two 32-channel "virtual heads" per head; H below is then the number of
// VIRTUAL heads, H_real * CB, and Ada-Temp -- a dot over all 64 channels -- sums the two 8-lane groups of a 16-lane row).
template <int LI, int NT = NTHR, bool SPLIT = false, bool PRIV = false, int CB = 1>
struct Loader {
    f32x4 xr[LI];
    float wt4[4];
    const float* src[LI];  // next tile's source of chunk i
    int row[LI];           // row inside the tile (>= TN: chunk unused for this H)
    int loff[LI];          // byte offset of the hi 8-B piece inside a tile buffer
    int soff[LI];          // byte offset of s_n inside an s array (q8 == 0 lanes)
    int q8;                // chunk index inside the head: c0 = 4*q8
    size_t tile_stride;    // floats per tile (32 rows)
    // SPLIT input (haet_project_split_fwd): rows arrive as {4 x bf16 hi, 4 x bf16 lo} per 16 B and
    // s_n = log2(e) / tau as an [N][H] array -- the loader only copies
    const float* sptr;     // next tile's s value of this thread (threads < 32 * H)
    size_t s_stride;
    float sv;
    int srow, soff2;

    __device__ __forceinline__ void init(const float* __restrict__ wt, const float* xb, int H, int row0,
                                         const float* sb = nullptr) {
        q8 = threadIdx.x & 7;
        if (PRIV) {
            // Wave-private loading (one head per wave): wave w copies ONLY head w's 32 x 32 block of the
            // tile (4 chunks per lane) and its 32 s values into head w's LDS images, which no other
            // wave reads -- the tile loop then needs no workgroup barrier and the waves drift apart.
            const int lane = threadIdx.x & 63, h = threadIdx.x >> 6, cprp = H * 8;
            tile_stride = (size_t)TN * cprp * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) wt4[i] = 0.f;
#pragma unroll
            for (int i = 0; i < LI; ++i) {
                const int e = lane + 64 * i, r = e >> 3, q = e & 7;
                row[i] = r;
                src[i] = xb + ((size_t)(row0 + r) * cprp + h * 8 + q) * 4;
                loff[i] = h * (2 * IMG) + img_off(r, q >> 1) + ((q & 1) << 3);
                soff[i] = 0;
            }
            srow = lane;  // lanes >= 32: no s value
            sptr = sb + (size_t)(row0 + (lane & (TN - 1))) * H + h;
            s_stride = (size_t)TN * H;
            soff2 = 4 * (h * TN + (lane & (TN - 1)));
            sv = 1.f;
            return;
        }
        if (SPLIT) {
            srow = (int)threadIdx.x / H;  // >= TN: no s value on this thread
            sptr = sb + (size_t)row0 * H + threadIdx.x;
            s_stride = (size_t)TN * H;
            soff2 = 4 * (((int)threadIdx.x % H) * TN + (srow & (TN - 1)));
            sv = 1.f;
        }
        const int cpr = H * 8;  // chunks per row
        tile_stride = (size_t)TN * cpr * 4;
        // (cpr divides NT: a thread's chunks all sit in the same column q of their rows, i.e. in the same virtual head)
        const int vh0 = ((int)threadIdx.x % cpr) >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) wt4[i] = SPLIT ? 0.f : wt[32 * (vh0 % CB) + 4 * q8 + i];
#pragma unroll
        for (int i = 0; i < LI; ++i) {
            const int e = threadIdx.x + NT * i;
            const int r = e / cpr, q = e - r * cpr, h = q >> 3;
            row[i] = r;
            src[i] = xb + ((size_t)(row0 + r) * cpr + q) * 4;
            loff[i] = h * (2 * IMG) + img_off(r & (TN - 1), q8 >> 1) + ((q8 & 1) << 3);
            soff[i] = 4 * (h * TN + (r & (TN - 1)));
        }
    }
    // issue the global loads of the next tile (rows >= nvalid read as zero)
    __device__ __forceinline__ void fetch(int nvalid) {
#pragma unroll
        for (int i = 0; i < LI; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row[i] < TN && row[i] < nvalid) v = *reinterpret_cast<const f32x4*>(src[i]);
            xr[i] = v;
            src[i] += tile_stride;
        }
        if (SPLIT) {
            float v = 1.f;
            if (srow < TN && srow < nvalid) v = *sptr;
            sv = v;
            sptr += s_stride;
        }
    }
    // split to bf16 hi/lo, write the LDS images, reduce Ada-Temp and write s_n = log2e / tau
    __device__ __forceinline__ void stage(int buf, int sarr, float bt, float base_temp) {
        if (SPLIT) {
#pragma unroll
            for (int i = 0; i < LI; ++i) {
                if (row[i] < TN) {
                    const u32x4 w = __builtin_bit_cast(u32x4, xr[i]);
                    const u32x2 hi = {w[0], w[1]}, lo = {w[2], w[3]};
                    *reinterpret_cast<u32x2*>(smem + buf + loff[i]) = hi;
                    *reinterpret_cast<u32x2*>(smem + buf + loff[i] + IMG) = lo;
                }
            }
            if (srow < TN) *reinterpret_cast<float*>(smem + sarr + soff2) = sv;
            return;
        }
#pragma unroll
        for (int i = 0; i < LI; ++i) {
            if (row[i] < TN) {
                const f32x4 v = xr[i];
                float t = v[0] * wt4[0];
                t = fmaf(v[1], wt4[1], t);
                t = fmaf(v[2], wt4[2], t);
                t = fmaf(v[3], wt4[3], t);
                t += dpp<0xB1>(t);   // quad_perm [1,0,3,2]
                t += dpp<0x4E>(t);   // quad_perm [2,3,0,1]
                t += dpp<0x141>(t);  // row_half_mirror: the other quad of the 8-lane group
                if (CB == 2) t += dpp<0x128>(t);  // row_ror:8: the head's other 32 channels (the neighbouring virtual head)
                unsigned h01, l01, h23, l23;
                split_bf16_pair(v[0], v[1], h01, l01);
                split_bf16_pair(v[2], v[3], h23, l23);
                const u32x2 hi = {h01, h23};
                const u32x2 lo = {l01, l23};
                *reinterpret_cast<u32x2*>(smem + buf + loff[i]) = hi;
                *reinterpret_cast<u32x2*>(smem + buf + loff[i] + IMG) = lo;
                if (q8 == 0)  // v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division
                    *reinterpret_cast<float*>(smem + sarr + soff[i]) =
                        kLog2e * __builtin_amdgcn_rcpf(ada_temp(t + bt, base_temp));
            }
        }
    }
};

// split-bf16 fragment of 8 consecutive channels of `row` (A operand of Y = X.Ws, B operand of Yt)
__device__ __forceinline__ Frag read_rows(int img, int row, int chunk) {
    Frag f;
    const int off = img + img_off(row, chunk);
    f.hi = *reinterpret_cast<const u32x4*>(smem + off);
    f.lo = *reinterpret_cast<const u32x4*>(smem + off + IMG);
    return f;
}

// transposed fragment: lane (c = lane&31, hf) gets X[n = 16s + 8(j>>2) + 4hf + (j&3)][c], j = 0..7
// -- the k order an accumulator tile has when it is reused as the A operand (guide s.3).
__device__ __forceinline__ Frag read_cols(int img, int s, int lane) {
    const int i = lane & 15, cb = (lane >> 4) & 1, hf = lane >> 5;
    const int chunk = 2 * cb + ((i & 3) >> 1), sub = (i & 1) << 3;
    Frag f;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const int base = img + part * IMG;
        const int r0 = 16 * s + 4 * hf + (i >> 2), r1 = r0 + 8;
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(smem + base + img_off(r0, chunk) + sub));
        const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(smem + base + img_off(r1, chunk) + sub));
        const u32x2 a2 = __builtin_bit_cast(u32x2, a), b2 = __builtin_bit_cast(u32x2, b);
        const u32x4 v = {a2[0], a2[1], b2[0], b2[1]};
        if (part == 0) f.hi = v;
        else f.lo = v;
    }
    return f;
}

// Ws as MB x CB swizzled [32][32] split-bf16 image pairs in LDS (same for every head; image (gb, cb) = slice columns
// 32 gb .., channels 32 cb ..): both passes re-read their Ws fragments per tile instead of pinning 16 VGPRs per column block.
template <int CB = 1>
__device__ __forceinline__ void stage_ws(int wimg, const float* __restrict__ Ws, int M) {
    for (int e = threadIdx.x; e < M * 4 * CB; e += blockDim.x) {
        const int g = e / (4 * CB), rest = e - g * (4 * CB), cb = rest >> 2, chunk = rest & 3;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = Ws[g * (kC * CB) + 32 * cb + 8 * chunk + j];
        const Frag f = split8(v);
        const int off = wimg + ((g >> 5) * CB + cb) * (2 * IMG) + img_off(g & 31, chunk);
        *reinterpret_cast<u32x4*>(smem + off) = f.hi;
        *reinterpret_cast<u32x4*>(smem + off + IMG) = f.lo;
    }
}


}  // namespace haet
