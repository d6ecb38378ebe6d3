// Per-point projections (in_project_x, to_out) as a split-bf16 MFMA GEMM:
//     out[n, o] = sum_k x[n, k] W[o, k] + b[o]        x, out fp32;  K, O = 256 at C3
// The reference runs these as fp32 cuBLAS GEMMs (models/PhysicsAttention/IrregularMesh.py:158,
// 235).  fp32 on MI355X is VALU/“f32 MFMA” rate (157 TF): 131 kflop/point makes the GEMM
// compute-bound at ~1.15 ms per 1M points (rocBLAS, measured), 3x the HBM time.  Split-bf16
// (hi*hi + hi*lo + lo*hi, fp32 accumulate, ~2^-16 operand error) runs on the bf16 matrix cores
// at 3 MFMAs per fp32 product and makes the GEMM HBM-bound again (2 * K * 4 B per point).
//
// Workgroup = 128 rows x BN columns, 8 waves as 2 (rows) x 4 (columns); K is walked in 32-wide
// tiles: x tile fp32 -> {hi, lo} bf16 in the loader, W comes pre-split (haet_split_bf16, done
// once per weight version), both as swizzled [32][32] bf16 LDS images read with ds_read_b128.
// Loads of k-tile t+1 are issued into registers before the MFMAs of tile t (2 WGs per CU
// interleave the rest).
#include "common.h"

namespace haet {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int GBM = 128;   // rows per workgroup
constexpr int GBK = 32;    // k-tile
constexpr int GIMG = 2048; // bytes of one [32][32] bf16 image

__device__ __forceinline__ unsigned g_pack(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ int g_off(int row, int chunk) {
    return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
}
struct GFrag {
    u32x4 hi, lo;
};
__device__ __forceinline__ GFrag g_read(const char* img, int row, int chunk) {
    GFrag f;
    const int off = g_off(row, chunk);
    f.hi = *reinterpret_cast<const u32x4*>(img + off);
    f.lo = *reinterpret_cast<const u32x4*>(img + GIMG + off);
    return f;
}
__device__ __forceinline__ f32x16 g_mfma3(const GFrag& a, const GFrag& b, f32x16 c) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a.hi), al = __builtin_bit_cast(bf16x8, a.lo);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, b.hi), bl = __builtin_bit_cast(bf16x8, b.lo);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
    return c;
}

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

// NCB = BN / 32 column blocks per workgroup (8 -> BN = 256, 4 -> BN = 128)
template <int NCB>
__global__ __launch_bounds__(512, 4) void gemm_x3_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ Whi,
    const unsigned short* __restrict__ Wlo, const float* __restrict__ bias, int Nrows, int K, int O,
    float* __restrict__ out) {
    constexpr int BN = 32 * NCB;
    constexpr int WCB = NCB / 4;  // column blocks per wave (4 waves across the columns)
    constexpr int XIMGS = GBM / 32, WLD = (NCB * 4) / 2;  // 16-B W chunks per thread per plane... see below
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ximg = smem;                       // [4 row blocks][hi|lo][2048]
    char* wimg = smem + XIMGS * 2 * GIMG;    // [NCB col blocks][hi|lo][2048]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hf = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 2, wn = wave & 3;
    const int row0 = blockIdx.x * GBM, col0 = blockIdx.y * BN;
    (void)WLD;

    // loader geometry: x tile = 128 rows x 8 chunks(16 B) -> 2 chunks per thread
    int xrow[2], xq[2];
    const float* xp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + 512 * i;
        xrow[i] = e >> 3;
        xq[i] = e & 7;
        const int gr = min(row0 + xrow[i], Nrows - 1);
        xp[i] = x + (size_t)gr * K + 4 * xq[i];
    }
    // W tile = BN rows x 4 chunks(16 B of bf16) per plane -> NCB/4 chunks per thread per plane
    constexpr int WPT = NCB / 4;
    int wrow[WPT], wch[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int e = tid + 512 * i;
        wrow[i] = e >> 2;
        wch[i] = e & 3;
    }

    f32x16 acc[2][WCB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WCB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 xr[2];
    u32x4 whr[WPT], wlr[WPT];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) xr[i] = *reinterpret_cast<const f32x4*>(xp[i] + k0);
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const size_t o = ((size_t)(col0 + wrow[i]) * K + k0 + 8 * wch[i]);
            whr[i] = *reinterpret_cast<const u32x4*>(Whi + o);
            wlr[i] = *reinterpret_cast<const u32x4*>(Wlo + o);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 v = xr[i];
            const unsigned h01 = g_pack(v[0], v[1]), h23 = g_pack(v[2], v[3]);
            const u32x2 hi = {h01, h23};
            const u32x2 lo = {g_pack(v[0] - __uint_as_float(h01 << 16), v[1] - __uint_as_float(h01 & 0xffff0000u)),
                              g_pack(v[2] - __uint_as_float(h23 << 16), v[3] - __uint_as_float(h23 & 0xffff0000u))};
            char* img = ximg + (xrow[i] >> 5) * (2 * GIMG);
            const int off = g_off(xrow[i] & 31, xq[i] >> 1) + ((xq[i] & 1) << 3);
            *reinterpret_cast<u32x2*>(img + off) = hi;
            *reinterpret_cast<u32x2*>(img + GIMG + off) = lo;
        }
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            char* img = wimg + (wrow[i] >> 5) * (2 * GIMG);
            const int off = g_off(wrow[i] & 31, wch[i]);
            *reinterpret_cast<u32x4*>(img + off) = whr[i];
            *reinterpret_cast<u32x4*>(img + GIMG + off) = wlr[i];
        }
    };

    const int nkt = K / GBK;
    fetch(0);
    stage();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) fetch((kt + 1) * GBK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            GFrag a[2], b[WCB];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = g_read(ximg + (2 * wm + i) * (2 * GIMG), l31, hf + 2 * ks);
#pragma unroll
            for (int j = 0; j < WCB; ++j) b[j] = g_read(wimg + (WCB * wn + j) * (2 * GIMG), l31, hf + 2 * ks);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < WCB; ++j) acc[i][j] = g_mfma3(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
        if (kt + 1 < nkt) {
            stage();
            __syncthreads();
        }
    }
    // epilogue: lane = output column, registers = rows
#pragma unroll
    for (int j = 0; j < WCB; ++j) {
        const int col = col0 + 32 * (WCB * wn + j) + l31;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hf;
                if (row < Nrows) out[(size_t)row * O + col] = acc[i][j][r] + bv;
            }
        }
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

extern "C" int haet_linear_supported(int K, int O) {
    return (K >= 32 && K % 32 == 0 && O >= 128 && O % 128 == 0) ? 1 : 0;
}

extern "C" int haet_linear_fwd(const float* x, const uint16_t* W_hi, const uint16_t* W_lo,
                               const float* bias, int Nrows, int K, int O, float* out,
                               haet_stream_t stream) {
    if (!x || !W_hi || !W_lo || !out || Nrows < 1) return fail(HAET_E_ARG, "linear_fwd: bad arguments");
    if (!haet_linear_supported(K, O))
        return fail(HAET_E_ARG, "linear_fwd: K=%d (multiple of 32) / O=%d (multiple of 128) unsupported", K, O);
    const int gx = (Nrows + GBM - 1) / GBM;
    if (O % 256 == 0) {
        const size_t lds = (GBM / 32 + 8) * 2 * GIMG;
        auto kern = gemm_x3_kernel<8>;
        hipLaunchKernelGGL(kern, dim3(gx, O / 256), dim3(512), lds, as_stream(stream), x, W_hi, W_lo,
                           bias, Nrows, K, O, out);
    } else {
        const size_t lds = (GBM / 32 + 4) * 2 * GIMG;
        auto kern = gemm_x3_kernel<4>;
        hipLaunchKernelGGL(kern, dim3(gx, O / 128), dim3(512), lds, as_stream(stream), x, W_hi, W_lo,
                           bias, Nrows, K, O, out);
    }
    return check_launch("gemm_x3_kernel");
}
