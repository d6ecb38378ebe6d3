// Weight gradient of a Linear over the points: dW[O][K] = sum_n G[n][o] X[n][k], db[o] = sum_n G[n][o]
// (to_out and in_project_x of the Physics_Attention modules, the block's MLPs, the convolution's taps; the reference gets them from torch autograd,
// exp_airfoil.py:249-255).  N is ~1e6, O = K = 256: an output-stationary GEMM whose contraction runs
// over the ROW index of both operands, so the f32 MFMA (v_mfma_f32_32x32x2_f32, exact f32 products)
// takes both operands as plain row-major tiles, the output column in the lane: no transposes.  A workgroup
// (8 waves, 2 x 4) keeps a 256 x 256 block of dW in its accumulators over its chunk of rows, stages 32-row
// tiles of both operands through a double-buffered LDS copy (each element leaves HBM / L2 once per workgroup)
// and writes one partial record; the caller sums the records.
#include <cstdlib>

#include "common.h"

namespace haet {

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned tn_pack(float a, float b) {  // v_cvt_pk_bf16_f32 (RNE)
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void tn_split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h, l;
        split_bf16_pair(v[2 * i], v[2 * i + 1], h, l);
        hi[i] = h, lo[i] = l;
    }
}

constexpr int kRowsTile = 32;

int wgrad_chunk(int N) {  // rows per workgroup: ~256 workgroups, whole 32-row tiles
    static const int forced = [] {
        const char* e = getenv("HAET_WGRAD_CHUNK");  // tuning: fixed rows per workgroup
        const int v = e ? atoi(e) : 0;
        return v >= kRowsTile ? (v / kRowsTile) * kRowsTile : 0;
    }();
    if (forced) return forced;
    int c = (N + 255) / 256;
    c = ((c + kRowsTile - 1) / kRowsTile) * kRowsTile;
    return c < kRowsTile ? kRowsTile : c;
}

}  // namespace

// LDS: two buffers of one 32-row tile of both operands, [G 32 x 256 | X 32 x 256] floats = 64 KB each.
constexpr int kTileFloats = 2 * kRowsTile * 256;

// TAP: the X row of output row n is row n + (dh gW + dw) gD + dd of the channels-last grid [B][gH][gW][gD], zero
// where that neighbour lies outside the grid -- one tap of a convolution's weight gradient
// (dW[o][c][tap] = sum_n g[n][o] x[n + off(tap)][c], StructuredMesh2D.py:109 / StructuredMesh3D.py:113).
struct TapGeom {
    int gH, gW, gD, dh, dw, dd;
};

// X3: split-bf16 products (hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_bf16, each wave splitting the fragments it
// reads) instead of exact f32 MFMA products: 1/5 of the matrix-core cycles, which leaves the kernel HBM-bound.
template <bool TAP, bool X3>
__global__ __launch_bounds__(512) void gemm_tn_f32_kernel(const float* __restrict__ G, const float* __restrict__ X,
                                                          int N, int O, int K, int chunk,
                                                          float* __restrict__ parts, TapGeom tg) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int kblocks = (K + 255) / 256;
    const int ob0 = (blockIdx.y / kblocks) * 256, kb0 = (blockIdx.y % kblocks) * 256;  // this workgroup's block of dW
    const int ol = 128 * (wave >> 2), kl = 64 * (wave & 3);                             // this wave's 128 x 64 part
    const int n0 = blockIdx.x * chunk, n1 = min(N, n0 + chunk);

    f16v acc[4][2];
    float dbl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[ob][kb][v] = 0.f;

    // staging: thread t moves float4 column t % 64 of rows t / 64 + 8 i (i < 4) of both operands -- whole 1-KB
    // rows per 64 lanes from global, consecutive 16-B cells into LDS
    const int sr = tid >> 6, sc = (tid & 63) * 4;
    const f4v zero4 = {0.f, 0.f, 0.f, 0.f};
    f4v sg[4], sx[4];
    auto fetch = [&](int nb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = nb + sr + 8 * i;
            const bool ok = n < n1;
            const size_t r = ok ? (size_t)n : (size_t)n0;  // unconditional loads: the waits stay counted
            size_t rx = r;
            bool okx = ok;
            if (TAP) {
                const int d = (int)(r % tg.gD), w = (int)((r / tg.gD) % tg.gW), hh = (int)((r / ((size_t)tg.gD * tg.gW)) % tg.gH);
                okx = ok && (unsigned)(hh + tg.dh) < (unsigned)tg.gH && (unsigned)(w + tg.dw) < (unsigned)tg.gW &&
                      (unsigned)(d + tg.dd) < (unsigned)tg.gD;
                if (okx) rx = (size_t)((long long)r + ((long long)tg.dh * tg.gW + tg.dw) * tg.gD + tg.dd);
            }
            // (O, K multiples of 128: the columns of a 256-wide block past the matrix read a clamped address as zero)
            const bool cg = ob0 + sc < O, cx = kb0 + sc < K;
            const f4v g4 = *reinterpret_cast<const f4v*>(G + r * O + (cg ? ob0 + sc : 0));
            const f4v x4 = *reinterpret_cast<const f4v*>(X + rx * K + (cx ? kb0 + sc : 0));
            sg[i] = ok && cg ? g4 : zero4;
            sx[i] = okx && cx ? x4 : zero4;
        }
    };
    auto stash = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f4v*>(buf + (sr + 8 * i) * 256 + sc) = sg[i];
            *reinterpret_cast<f4v*>(buf + kRowsTile * 256 + (sr + 8 * i) * 256 + sc) = sx[i];
        }
    };
    fetch(n0);
    stash(lds);
    __syncthreads();
    int cur = 0;
    for (int nb = n0; nb < n1; nb += kRowsTile) {
        fetch(nb + kRowsTile);  // in flight under this tile's 128 MFMAs
        const float* gs = lds + cur * kTileFloats + ol + l31;
        const float* xs = lds + cur * kTileFloats + kRowsTile * 256 + kl + l31;
        if (X3) {
            // k-step s of the 32x32x16 instruction: this lane's 8 k-elements = rows 16 s + 8 half + j of the tile
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 ah[4], al[4], bh[2], bl[2];
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = gs[(16 * ks + 8 * half + j) * 256 + 32 * ob];
                    dbl[ob] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                    tn_split8(v, ah[ob], al[ob]);
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = xs[(16 * ks + 8 * half + j) * 256 + 32 * kb];
                    tn_split8(v, bh[kb], bl[kb]);
                }
#pragma unroll
                for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        const bf16x8 a0 = __builtin_bit_cast(bf16x8, ah[ob]), a1 = __builtin_bit_cast(bf16x8, al[ob]);
                        const bf16x8 b0 = __builtin_bit_cast(bf16x8, bh[kb]), b1 = __builtin_bit_cast(bf16x8, bl[kb]);
                        acc[ob][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[ob][kb], 0, 0, 0);
                        acc[ob][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[ob][kb], 0, 0, 0);
                        acc[ob][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[ob][kb], 0, 0, 0);
                    }
            }
        }
        // k-pair kk of the instruction = rows kk (lanes 0-31) and kk + 16 (lanes 32-63) of the tile, both operands
#pragma unroll
        for (int kk = 0; kk < (X3 ? 0 : 16); ++kk) {
            float a[4], b[2];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) a[ob] = gs[(kk + 16 * half) * 256 + 32 * ob];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) b[kb] = xs[(kk + 16 * half) * 256 + 32 * kb];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                dbl[ob] += a[ob];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
                    acc[ob][kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ob], b[kb], acc[ob][kb], 0, 0, 0);
            }
        }
        cur ^= 1;
        stash(lds + cur * kTileFloats);  // last read as tile nb - 32, before the barrier that ended that iteration
        __syncthreads();
    }

    float* rec = parts + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (size_t)(256 * 256 + 256);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int o = ol + 32 * ob + 8 * (v >> 2) + 4 * half + (v & 3);
                rec[o * 256 + kl + 32 * kb + l31] = acc[ob][kb][v];
            }
        const float s = dbl[ob] + __shfl_xor(dbl[ob], 32);
        if ((wave & 3) == 0 && half == 0) rec[256 * 256 + ol + 32 * ob + l31] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same contraction with the split done ONCE per element: the staging threads convert each 32-row tile of G and X
// to {hi, lo} bf16 images [32 rows][32 columns] (64-B rows, 16-B chunks XOR-swizzled: the layout of slice_mfma.hip),
// and every wave takes its A / B fragments -- 8 consecutive ROWS of one column, i.e. the contraction index in the
// k slot -- with the transposed LDS read ds_read_b64_tr_b16.  No per-wave split (each element was split by up to
// four waves above), a quarter of the LDS read instructions.
constexpr int kImg = 2048 + 32;                 // bytes per image plane
constexpr int kTileImgBytes = 2 * 8 * 2 * kImg;  // G and X: 8 column blocks each, {hi, lo}

__device__ __forceinline__ int tn_img_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

typedef __attribute__((ext_vector_type(4))) short tn_s16x4;
typedef __attribute__((ext_vector_type(2))) unsigned tn_u32x2;

// fragment of k-step s (rows 16 s .. 16 s + 15 of the tile) for column lane % 32 of the image at byte offset img
__device__ __forceinline__ void tn_read_cols(const char* smem, int img, int s, int lane, u32x4& hi, u32x4& lo) {
    const int i = lane & 15, cb = (lane >> 4) & 1, hf = lane >> 5;
    const int chunk = 2 * cb + ((i & 3) >> 1), sub = (i & 1) << 3;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const int base = img + part * kImg;
        const int r0 = 16 * s + 4 * hf + (i >> 2), r1 = r0 + 8;
        const tn_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) tn_s16x4*)(smem + base + tn_img_off(r0, chunk) + sub));
        const tn_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) tn_s16x4*)(smem + base + tn_img_off(r1, chunk) + sub));
        const tn_u32x2 a2 = __builtin_bit_cast(tn_u32x2, a), b2 = __builtin_bit_cast(tn_u32x2, b);
        const u32x4 v = {a2[0], a2[1], b2[0], b2[1]};
        if (part == 0) hi = v;
        else lo = v;
    }
}

template <bool TAP>
__global__ __launch_bounds__(512) void gemm_tn_x3_kernel(const float* __restrict__ G, const float* __restrict__ X,
                                                         int N, int O, int K, int chunk,
                                                         float* __restrict__ parts, TapGeom tg) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char tsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int kblocks = (K + 255) / 256;
    const int ob0 = (blockIdx.y / kblocks) * 256, kb0 = (blockIdx.y % kblocks) * 256;
    const int wo = wave >> 2, wk = wave & 3;  // this wave's 128 x 64 part of the 256 x 256 block
    const int n0 = blockIdx.x * chunk, n1 = min(N, n0 + chunk);

    f16v acc[4][2];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[ob][kb][v] = 0.f;

    // staging: thread t = float4 column t % 64 of rows t / 64 + 8 i (i < 4) of both operands
    const int sr = tid >> 6, sc = (tid & 63) * 4, q8 = tid & 7, blk = (tid & 63) >> 3;
    const f4v zero4 = {0.f, 0.f, 0.f, 0.f};
    f4v sg[4], sx[4], dbs = zero4;  // dbs: this thread's 4 columns of sum_n G over its rows (the bias gradient)
    auto fetch = [&](int nb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = nb + sr + 8 * i;
            const bool ok = n < n1;
            const size_t r = ok ? (size_t)n : (size_t)n0;  // unconditional loads: the waits stay counted
            size_t rx = r;
            bool okx = ok;
            if (TAP) {
                const int d = (int)(r % tg.gD), w = (int)((r / tg.gD) % tg.gW), hh = (int)((r / ((size_t)tg.gD * tg.gW)) % tg.gH);
                okx = ok && (unsigned)(hh + tg.dh) < (unsigned)tg.gH && (unsigned)(w + tg.dw) < (unsigned)tg.gW &&
                      (unsigned)(d + tg.dd) < (unsigned)tg.gD;
                if (okx) rx = (size_t)((long long)r + ((long long)tg.dh * tg.gW + tg.dw) * tg.gD + tg.dd);
            }
            // (O, K multiples of 128: the columns of a 256-wide block past the matrix read a clamped address as zero)
            const bool cg = ob0 + sc < O, cx = kb0 + sc < K;
            const f4v g4 = *reinterpret_cast<const f4v*>(G + r * O + (cg ? ob0 + sc : 0));
            const f4v x4 = *reinterpret_cast<const f4v*>(X + rx * K + (cx ? kb0 + sc : 0));
            sg[i] = ok && cg ? g4 : zero4;
            sx[i] = okx && cx ? x4 : zero4;
        }
    };
    auto put = [&](int base, int row, const f4v v) {  // 4 consecutive columns of one row -> the 8-B {hi} and {lo} pieces
        unsigned h01, l01, h23, l23;
        split_bf16_pair(v[0], v[1], h01, l01);
        split_bf16_pair(v[2], v[3], h23, l23);
        const tn_u32x2 hi = {h01, h23};
        const tn_u32x2 lo = {l01, l23};
        const int off = base + tn_img_off(row, q8 >> 1) + ((q8 & 1) << 3);
        *reinterpret_cast<tn_u32x2*>(tsm + off) = hi;
        *reinterpret_cast<tn_u32x2*>(tsm + off + kImg) = lo;
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dbs += sg[i];
            put(buf + blk * (2 * kImg), sr + 8 * i, sg[i]);
            put(buf + (8 + blk) * (2 * kImg), sr + 8 * i, sx[i]);
        }
    };
    // O, K are multiples of 128 and a wave owns 128 x 64 of the block: at O = 128 or K = 128 (the reference's PDE widths)
    // whole waves lie outside the matrix -- they stage like the others but neither multiply nor store (the host reads the
    // O x K corner of a record only): 1/4 of the partial-record traffic at 128 x 128
    const bool wave_on = ob0 + 128 * wo < O && kb0 + 64 * wk < K;
    fetch(n0);
    stash(0);
    __syncthreads();
    int cur = 0;
    for (int nb = n0; nb < n1; nb += kRowsTile) {
        fetch(nb + kRowsTile);  // in flight under this tile's MFMAs
        const int gimg = cur * kTileImgBytes + (4 * wo) * (2 * kImg);
        const int ximg = cur * kTileImgBytes + (8 + 2 * wk) * (2 * kImg);
#pragma unroll
        for (int ks = 0; wave_on && ks < 2; ++ks) {
            u32x4 ah[4], al[4], bh[2], bl[2];
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) tn_read_cols(tsm, gimg + ob * (2 * kImg), ks, lane, ah[ob], al[ob]);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) tn_read_cols(tsm, ximg + kb * (2 * kImg), ks, lane, bh[kb], bl[kb]);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const bf16x8 a0 = __builtin_bit_cast(bf16x8, ah[ob]), a1 = __builtin_bit_cast(bf16x8, al[ob]);
                    const bf16x8 b0 = __builtin_bit_cast(bf16x8, bh[kb]), b1 = __builtin_bit_cast(bf16x8, bl[kb]);
                    acc[ob][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[ob][kb], 0, 0, 0);
                    acc[ob][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[ob][kb], 0, 0, 0);
                    acc[ob][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[ob][kb], 0, 0, 0);
                }
        }
        cur ^= 1;
        stash(cur * kTileImgBytes);  // last read as tile nb - 32, before the barrier that ended that iteration
        __syncthreads();
    }
    // (the last stash staged an all-zero tile past the chunk and added zeros to dbs)

    float* rec = parts + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (size_t)(256 * 256 + 256);
    const int ol = 128 * wo, kl = 64 * wk;
    if (wave_on) {
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int o = ol + 32 * ob + 8 * (v >> 2) + 4 * half + (v & 3);
                    rec[o * 256 + kl + 32 * kb + l31] = acc[ob][kb][v];
                }
    }
    // bias gradient: wave w holds the sums over rows = w (mod 8) of all 256 columns (lane l: columns 4l .. 4l+3)
    float* red = reinterpret_cast<float*>(tsm);
    *reinterpret_cast<f4v*>(red + wave * 256 + 4 * lane) = dbs;
    __syncthreads();
    if (tid < 256) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) sum += red[w * 256 + tid];
        rec[256 * 256 + tid] = sum;
    }
}

static bool wgrad_exact_f32() {  // HAET_WGRAD_F32=1: exact f32 MFMA products (A/B runs, reference for the tests)
    static const bool on = getenv("HAET_WGRAD_F32") != nullptr;
    return on;
}
static bool wgrad_wave_split() {  // HAET_WGRAD_WAVE_SPLIT=1: the split-bf16 variant that splits per wave (A/B runs)
    static const bool on = getenv("HAET_WGRAD_WAVE_SPLIT") != nullptr;
    return on;
}
static int x3_attr() {
    static PerDeviceOnce done;
    if (!done.done()) {
        for (const void* k : {reinterpret_cast<const void*>(gemm_tn_x3_kernel<false>),
                              reinterpret_cast<const void*>(gemm_tn_x3_kernel<true>)}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kTileImgBytes);
            if (e != hipSuccess) return fail(HAET_E_LAUNCH, "linear_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        done.set();
    }
    return HAET_OK;
}

}  // namespace haet

using namespace haet;

extern "C" int haet_linear_wgrad_supported(int O, int K) { return (O > 0 && K > 0 && O % 128 == 0 && K % 128 == 0) ? 1 : 0; }

extern "C" int haet_linear_wgrad_num_parts(int N) {
    if (N < 1) return HAET_E_ARG;
    const int chunk = wgrad_chunk(N);
    return (N + chunk - 1) / chunk;
}

extern "C" int haet_linear_wgrad(const float* G, const float* X, int N, int O, int K, float* parts,
                                 haet_stream_t stream) {
    if (!G || !X || !parts) return fail(HAET_E_ARG, "linear_wgrad: null pointer");
    if (N < 1 || !haet_linear_wgrad_supported(O, K))
        return fail(HAET_E_ARG, "linear_wgrad: N=%d O=%d K=%d unsupported (O, K multiples of 128)", N, O, K);
    const int chunk = wgrad_chunk(N);
    dim3 grid((N + chunk - 1) / chunk, ((O + 255) / 256) * ((K + 255) / 256));
    const size_t lds = 2 * (size_t)kTileFloats * sizeof(float);  // 128 KB
    static PerDeviceOnce attr_set;
    if (!attr_set.done()) {
        for (const void* k : {reinterpret_cast<const void*>(gemm_tn_f32_kernel<false, false>),
                              reinterpret_cast<const void*>(gemm_tn_f32_kernel<false, true>)}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(HAET_E_LAUNCH, "linear_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        attr_set.set();
    }
    if (wgrad_exact_f32())
        hipLaunchKernelGGL((gemm_tn_f32_kernel<false, false>), grid, dim3(512), lds, as_stream(stream), G, X, N, O, K,
                           chunk, parts, TapGeom{});
    else if (wgrad_wave_split())
        hipLaunchKernelGGL((gemm_tn_f32_kernel<false, true>), grid, dim3(512), lds, as_stream(stream), G, X, N, O, K,
                           chunk, parts, TapGeom{});
    else {
        if (int rc = x3_attr()) return rc;
        hipLaunchKernelGGL((gemm_tn_x3_kernel<false>), grid, dim3(512), 2 * kTileImgBytes, as_stream(stream), G, X, N, O,
                           K, chunk, parts, TapGeom{});
    }
    return check_launch("gemm_tn kernel");
}

// One tap (dh, dw, dd) of the weight gradient of a same-padded stride-1 convolution over the channels-last grid
// x [B, gH*gW*gD, Cin], g = dL/d output [B, gH*gW*gD, O]: records as haet_linear_wgrad (K = Cin).
extern "C" int haet_conv_wgrad_tap(const float* G, const float* X, int B, int gH, int gW, int gD, int O, int Cin,
                                   int dh, int dw, int dd, float* parts, haet_stream_t stream) {
    if (!G || !X || !parts) return fail(HAET_E_ARG, "conv_wgrad_tap: null pointer");
    const long long rows = (long long)B * gH * gW * gD;
    if (B < 1 || gH < 1 || gW < 1 || gD < 1 || rows > 0x7fffffffLL || !haet_linear_wgrad_supported(O, Cin))
        return fail(HAET_E_ARG, "conv_wgrad_tap: bad grid or O=%d / C_in=%d not multiples of 128", O, Cin);
    const int N = (int)rows, chunk = wgrad_chunk(N);
    dim3 grid((N + chunk - 1) / chunk, ((O + 255) / 256) * ((Cin + 255) / 256));
    const size_t lds = 2 * (size_t)kTileFloats * sizeof(float);
    static PerDeviceOnce attr_set;
    if (!attr_set.done()) {
        for (const void* k : {reinterpret_cast<const void*>(gemm_tn_f32_kernel<true, false>),
                              reinterpret_cast<const void*>(gemm_tn_f32_kernel<true, true>)}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(HAET_E_LAUNCH, "conv_wgrad_tap: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        attr_set.set();
    }
    if (wgrad_exact_f32())
        hipLaunchKernelGGL((gemm_tn_f32_kernel<true, false>), grid, dim3(512), lds, as_stream(stream), G, X, N, O, Cin,
                           chunk, parts, TapGeom{gH, gW, gD, dh, dw, dd});
    else if (wgrad_wave_split())
        hipLaunchKernelGGL((gemm_tn_f32_kernel<true, true>), grid, dim3(512), lds, as_stream(stream), G, X, N, O, Cin,
                           chunk, parts, TapGeom{gH, gW, gD, dh, dw, dd});
    else {
        if (int rc = x3_attr()) return rc;
        hipLaunchKernelGGL((gemm_tn_x3_kernel<true>), grid, dim3(512), 2 * kTileImgBytes, as_stream(stream), G, X, N, O,
                           Cin, chunk, parts, TapGeom{gH, gW, gD, dh, dw, dd});
    }
    return check_launch("gemm_tn_f32_kernel<tap>");
}
