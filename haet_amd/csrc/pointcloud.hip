// Stand-alone ErwinFlash on RAW point clouds (models/components/erwinflash/erwin_flash.py:138-195 with
// embed=True / mp_steps>0, benchmarks/02.1-Car-Design-ShapeNetCar/main.py:50-99): tens of thousands to millions of
// points instead of <= 512 slice tokens, so the state lives in HBM ([tree slots][channels] fp32, tree order) and every
// stage is a streaming pass over it -- the Linears are the split-bf16 GEMMs of gemm_x3.hip, everything in between is
// here: radius graph, MPNN message / aggregate / update, RMSNorm + positional term, per-ball attention, SwiGLU gate,
// BallPooling / BallUnpooling gathers, BatchNorm1d (batch statistics or running statistics), gathers / scatters of
// the tree order.  One wave per row, 16-B accesses, fp32 arithmetic throughout.
#include "common.h"

#include <cfloat>
#include <cstdlib>

namespace haet {

typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kRowsPerBlock = 4;  // 256 threads = 4 waves = 4 rows

__device__ __forceinline__ float pc_gelu(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float pc_silu(float v) { return v / (1.0f + expf(-v)); }

// a row of c floats (c % 4 == 0, c <= 256 NV) spread over the wave: lane holds chunks lane, lane + 64, ...
template <int NV>
struct Row {
    f32x4 v[NV];
    __device__ __forceinline__ void load(const float* p, int c, int lane) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int ch = 4 * (lane + 64 * i);
            v[i] = ch < c ? *reinterpret_cast<const f32x4*>(p + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __device__ __forceinline__ void store(float* p, int c, int lane) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int ch = 4 * (lane + 64 * i);
            if (ch < c) *reinterpret_cast<f32x4*>(p + ch) = v[i];
        }
    }
    __device__ __forceinline__ float sum() const {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        return wave_sum(s);
    }
    __device__ __forceinline__ float sumsq() const {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) s = fmaf(v[i][q], v[i][q], s);
        return wave_sum(s);
    }
};

// sum over the 64 lanes of the wave, or (HALF) over the 32 lanes of this lane's half
template <bool HALF>
__device__ __forceinline__ float red_sum(float v) {
#pragma unroll
    for (int o = HALF ? 16 : 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int NV>
__device__ __forceinline__ float row_partial(const f32x4 (&v)[NV]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    return s;
}

#define PC_ROW_PROLOGUE(ROWS)                                          \
    const int lane = threadIdx.x & 63;                                 \
    const long long row = (long long)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6); \
    if (row >= (ROWS)) return;

// dispatch on the row width (c <= 1024, c % 4 == 0)
#define PC_DISPATCH(c, KERNEL, ...)                                                            \
    do {                                                                                       \
        if ((c) <= 256) hipLaunchKernelGGL((KERNEL<1>), __VA_ARGS__);                          \
        else if ((c) <= 512) hipLaunchKernelGGL((KERNEL<2>), __VA_ARGS__);                     \
        else hipLaunchKernelGGL((KERNEL<4>), __VA_ARGS__);                                     \
    } while (0)

static inline dim3 row_grid(long long rows) { return dim3((unsigned)((rows + kRowsPerBlock - 1) / kRowsPerBlock)); }
static inline bool width_ok(int c) { return c >= 4 && c % 4 == 0 && c <= 1024; }

// ------------------------------------------------------------------------------------------------ gathers
template <int NV>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, int ldx, const int32_t* __restrict__ idx,
                                                         long long rows, int c, float* __restrict__ out, int ldo) {
    PC_ROW_PROLOGUE(rows)
    Row<NV> r;
    r.load(x + (size_t)idx[row] * ldx, c, lane);
    r.store(out + (size_t)row * ldo, c, lane);
}

// out[tree_idx[s]] = X[s] for the slots that hold an original point (erwin_flash.py:193)
template <int NV>
__global__ __launch_bounds__(256) void scatter_masked_kernel(const float* __restrict__ X, int ldx, const int32_t* __restrict__ tree_idx,
                                                            const uint8_t* __restrict__ mask, long long rows, int c,
                                                            float* __restrict__ out, int ldo) {
    PC_ROW_PROLOGUE(rows)
    if (!mask[row]) return;
    Row<NV> r;
    r.load(X + (size_t)row * ldx, c, lane);
    r.store(out + (size_t)tree_idx[row] * ldo, c, lane);
}

// X[perm[p]] (or X[p]) += R[p]
template <int NV>
__global__ __launch_bounds__(256) void rows_add_kernel(float* __restrict__ X, int ldx, const int32_t* __restrict__ perm,
                                                      const float* __restrict__ R, int ldr, long long rows, int c) {
    PC_ROW_PROLOGUE(rows)
    const long long slot = perm ? perm[row] : row;
    Row<NV> a, b;
    a.load(X + (size_t)slot * ldx, c, lane);
    b.load(R + (size_t)row * ldr, c, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i) a.v[i] += b.v[i];
    a.store(X + (size_t)slot * ldx, c, lane);
}

// ------------------------------------------------------------------------------------------------ Erwin block pieces
// rel[p] = P[slot(p)] - mean over p's ball (attention.py:35-40); one wave per ball, sequence order
__global__ __launch_bounds__(256) void ball_rel_kernel(const float* __restrict__ P, const int32_t* __restrict__ perm, long long n_balls,
                                                      int ball, int d, float* __restrict__ rel) {
    PC_ROW_PROLOGUE(n_balls)
    float s[3] = {0.f, 0.f, 0.f};
    for (int q = lane; q < ball; q += 64) {
        const long long p = row * ball + q, slot = perm ? perm[p] : p;
        for (int k = 0; k < d; ++k) s[k] += P[(size_t)slot * d + k];
    }
    for (int k = 0; k < d; ++k) s[k] = wave_sum(s[k]) / (float)ball;
    for (int q = lane; q < ball; q += 64) {
        const long long p = row * ball + q, slot = perm ? perm[p] : p;
        for (int k = 0; k < d; ++k) rel[(size_t)p * d + k] = P[(size_t)slot * d + k] - s[k];
    }
}

// Y[p] = RMSNorm(X[slot(p)]) * w (+ pe_w rel[p] + pe_b)   (layers.py:25-32, attention.py:59-61; nn.RMSNorm eps=None
// -> machine epsilon)
template <int NV>
__global__ __launch_bounds__(256) void rms_pe_kernel(const float* __restrict__ X, int ldx, const int32_t* __restrict__ perm,
                                                    const float* __restrict__ rel, long long rows, int c, int d,
                                                    const float* __restrict__ w, const float* __restrict__ pe_w,
                                                    const float* __restrict__ pe_b, float* __restrict__ Y, int ldy) {
    PC_ROW_PROLOGUE(rows)
    const long long slot = perm ? perm[row] : row;
    Row<NV> r;
    r.load(X + (size_t)slot * ldx, c, lane);
    const float rstd = rsqrtf(r.sumsq() / (float)c + FLT_EPSILON);
    float rp[3] = {0.f, 0.f, 0.f};
    if (pe_w)
        for (int k = 0; k < d; ++k) rp[k] = rel[(size_t)row * d + k];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int ch = 4 * (lane + 64 * i);
        if (ch < c) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + ch);
            f32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                o[q] = r.v[i][q] * rstd * wv[q];
                if (pe_w) {
                    float pe = pe_b[ch + q];
                    for (int k = 0; k < d; ++k) pe = fmaf(pe_w[(size_t)(ch + q) * d + k], rp[k], pe);
                    o[q] += pe;
                }
            }
            *reinterpret_cast<f32x4*>(Y + (size_t)row * ldy + ch) = o;
        }
    }
}

// softmax(Q K^T / sqrt(hd)) V inside every ball, per head (attention.py:63-91; flash_attn_qkvpacked_func's contract).
// One workgroup per ball: K and V of all heads in LDS, one (query, head) pair per thread at a time, online softmax.
extern __shared__ __attribute__((aligned(16))) float pc_lds[];
constexpr int kAttThreads = 512;  // one (query, head) pair per thread at ball 64 x 8 heads; the ball's K / V in LDS bound the workgroups per CU
template <int HD>
__global__ __launch_bounds__(kAttThreads) void ball_attention_kernel(const float* __restrict__ qkv, int ldq, int c, int heads, int ball,
                                                            float scale, float* __restrict__ out, int ldo) {
    float* const Ks = pc_lds;
    float* const Vs = pc_lds + (size_t)ball * c;
    const size_t r0 = (size_t)blockIdx.x * ball;
    for (int i = threadIdx.x; i < ball * (c / 4); i += kAttThreads) {
        const int j = i / (c / 4), ch = 4 * (i % (c / 4));
        *reinterpret_cast<f32x4*>(Ks + (size_t)j * c + ch) = *reinterpret_cast<const f32x4*>(qkv + (r0 + j) * ldq + c + ch);
        *reinterpret_cast<f32x4*>(Vs + (size_t)j * c + ch) = *reinterpret_cast<const f32x4*>(qkv + (r0 + j) * ldq + 2 * c + ch);
    }
    __syncthreads();
    constexpr float kL2e = 1.4426950408889634f;
    for (int t = threadIdx.x; t < ball * heads; t += kAttThreads) {
        const int h = t / ball, i = t % ball;
        float q[HD], acc[HD];
#pragma unroll
        for (int e = 0; e < HD; e += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(qkv + (r0 + i) * ldq + h * HD + e);
#pragma unroll
            for (int u = 0; u < 4; ++u) q[e + u] = v[u] * (scale * kL2e), acc[e + u] = 0.f;  // scores in log2 units
        }
        // online softmax over the keys in chunks of 16: one rescale of the accumulator per chunk, one v_exp per key
        float mx = -INFINITY, l = 0.f;
        for (int j0 = 0; j0 < ball; j0 += 16) {
            float sc[16];
            float cm = mx;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const f32x4* kj = reinterpret_cast<const f32x4*>(Ks + (size_t)(j0 + u < ball ? j0 + u : ball - 1) * c + h * HD);
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < HD; e += 4) {
                    const f32x4 k4 = kj[e / 4];  // (one 16-B LDS read per four channels; all lanes of a head read the same row)
                    s = fmaf(q[e], k4[0], fmaf(q[e + 1], k4[1], fmaf(q[e + 2], k4[2], fmaf(q[e + 3], k4[3], s))));
                }
                sc[u] = j0 + u < ball ? s : -INFINITY;
                cm = fmaxf(cm, sc[u]);
            }
            const float corr = __builtin_amdgcn_exp2f(mx - cm);  // (first chunk: exp2(-inf) = 0)
            l *= corr;
#pragma unroll
            for (int e = 0; e < HD; ++e) acc[e] *= corr;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float pj = __builtin_amdgcn_exp2f(sc[u] - cm);
                const f32x4* vj = reinterpret_cast<const f32x4*>(Vs + (size_t)(j0 + u < ball ? j0 + u : ball - 1) * c + h * HD);
                l += pj;
#pragma unroll
                for (int e = 0; e < HD; e += 4) {
                    const f32x4 v4 = vj[e / 4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[e + r] = fmaf(pj, v4[r], acc[e + r]);
                }
            }
            mx = cm;
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int e = 0; e < HD; e += 4) {
            const f32x4 o = {acc[e] * inv, acc[e + 1] * inv, acc[e + 2] * inv, acc[e + 3] * inv};
            *reinterpret_cast<f32x4*>(out + (r0 + i) * ldo + h * HD + e) = o;
        }
    }
}

// The same attention on the matrix cores (round 4; power-of-two balls of 32 .. 128 tokens, head dims 8 / 16 / 32 -- the
// ShapeNetCar configuration: c = 128, 8 heads x 16, balls 64 / 32).  One workgroup per ball, K and V of all heads in LDS as
// above; an item = (32-query tile, head), dealt to the 8 waves:
//   St[k][q] = sum_d K[k][d] Q[q][d]   per 32-key tile: A = K rows from LDS, B = the lane's own query row from global memory
//                                      (scaled to log2 units), split bf16 x 3 on v_mfma_f32_32x32x16_bf16 (head dim 8: the
//                                      upper half of the 16-wide k-step is zero)
//   the softmax of a query is over ONE lane's registers (16 per key tile) + its partner lane (xor 32): no LDS, no atomics
//   Ot[d][q] = sum_k Vt[d][k] P[k][q]  A = V gathered from LDS in the accumulator's key order, B = P as it stands
// against ~2 ball^2 head_dim fused multiply-adds per (query, head) THREAD in the kernel above.
typedef __attribute__((ext_vector_type(16))) float pc_f16v;
typedef __attribute__((ext_vector_type(8))) __bf16 pc_bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned pc_u4;
__device__ __forceinline__ void pc_split8(const float (&v)[8], pc_u4& hi, pc_u4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h, l;
        split_bf16_pair(v[2 * i], v[2 * i + 1], h, l);
        hi[i] = h, lo[i] = l;
    }
}
__device__ __forceinline__ pc_f16v pc_mfma3(pc_u4 ah, pc_u4 al, pc_u4 bh, pc_u4 bl, pc_f16v c) {
    const pc_bf16x8 a0 = __builtin_bit_cast(pc_bf16x8, ah), a1 = __builtin_bit_cast(pc_bf16x8, al);
    const pc_bf16x8 b0 = __builtin_bit_cast(pc_bf16x8, bh), b1 = __builtin_bit_cast(pc_bf16x8, bl);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);
    return c;
}
template <int HD, int KT>  // KT = ball / 32 key tiles
__global__ __launch_bounds__(kAttThreads) void ball_attention_mfma_kernel(const float* __restrict__ qkv, int ldq, int c, int heads,
                                                                         float scale, float* __restrict__ out, int ldo) {
    constexpr int ball = 32 * KT, NS = HD > 16 ? HD / 16 : 1;  // 16-wide k-steps of the score product
    float* const Ks = pc_lds;
    float* const Vs = pc_lds + (size_t)ball * c;
    const size_t r0 = (size_t)blockIdx.x * ball;
    for (int i = threadIdx.x; i < ball * (c / 4); i += kAttThreads) {
        const int j = i / (c / 4), ch = 4 * (i % (c / 4));
        *reinterpret_cast<f32x4*>(Ks + (size_t)j * c + ch) = *reinterpret_cast<const f32x4*>(qkv + (r0 + j) * ldq + c + ch);
        *reinterpret_cast<f32x4*>(Vs + (size_t)j * c + ch) = *reinterpret_cast<const f32x4*>(qkv + (r0 + j) * ldq + 2 * c + ch);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hf = lane >> 5, l31 = lane & 31;
    const float qs = scale * 1.4426950408889634f;  // scores in log2 units
    for (int item = wave; item < KT * heads; item += kAttThreads / 64) {
        const int qt = item / heads, h = item - qt * heads;
        const float* qrow = qkv + (r0 + 32 * qt + l31) * ldq + h * HD;
        // B operand: this lane's query, elements d = 16 s + 8 hf .. + 7 of k-step s (head dim 8: the hf = 1 half is zero)
        pc_u4 qh[NS], ql[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int d0 = HD == 8 ? 0 : 16 * s + 8 * hf;
            const f32x4 a = *reinterpret_cast<const f32x4*>(qrow + d0), b = *reinterpret_cast<const f32x4*>(qrow + d0 + 4);
            const bool pad = HD == 8 && hf;
            const float v[8] = {pad ? 0.f : a[0] * qs, pad ? 0.f : a[1] * qs, pad ? 0.f : a[2] * qs, pad ? 0.f : a[3] * qs,
                                pad ? 0.f : b[0] * qs, pad ? 0.f : b[1] * qs, pad ? 0.f : b[2] * qs, pad ? 0.f : b[3] * qs};
            pc_split8(v, qh[s], ql[s]);
        }
        pc_f16v S[KT];
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) S[kt][r] = 0.f;
            const float* krow = Ks + (size_t)(32 * kt + l31) * c + h * HD;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int d0 = HD == 8 ? 0 : 16 * s + 8 * hf;
                const f32x4 a = *reinterpret_cast<const f32x4*>(krow + d0), b = *reinterpret_cast<const f32x4*>(krow + d0 + 4);
                const bool pad = HD == 8 && hf;
                const float v[8] = {pad ? 0.f : a[0], pad ? 0.f : a[1], pad ? 0.f : a[2], pad ? 0.f : a[3],
                                    pad ? 0.f : b[0], pad ? 0.f : b[1], pad ? 0.f : b[2], pad ? 0.f : b[3]};
                pc_u4 kh, kl;
                pc_split8(v, kh, kl);
                S[kt] = pc_mfma3(kh, kl, qh[s], ql[s], S[kt]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, S[kt][r]);
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                S[kt][r] = __builtin_amdgcn_exp2f(S[kt][r] - m);
                l += S[kt][r];
            }
        l += __shfl_xor(l, 32);
        pc_f16v O;
#pragma unroll
        for (int r = 0; r < 16; ++r) O[r] = 0.f;
        const int dcol = h * HD + (l31 & (HD - 1));
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float vv[8], pv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int key = 32 * kt + 16 * s + 8 * (j >> 2) + 4 * hf + (j & 3);  // the accumulator's key order
                    const float v1 = Vs[(size_t)key * c + dcol];  // (unconditional read; rows d >= HD of Vt are zero)
                    vv[j] = l31 < HD ? v1 : 0.f;
                    pv[j] = S[kt][8 * s + j];
                }
                pc_u4 vh, vl, ph, pl;
                pc_split8(vv, vh, vl);
                pc_split8(pv, ph, pl);
                O = pc_mfma3(vh, vl, ph, pl, O);
            }
        // lane (q, hf) holds O[q][d] for d = (r & 3) + 8 (r >> 2) + 4 hf, r < HD / 2: four consecutive d per register quad
        const float inv = 1.0f / l;
        float* orow = out + (r0 + 32 * qt + l31) * ldo + h * HD + 4 * hf;
#pragma unroll
        for (int g = 0; g < HD / 8; ++g) {
            const f32x4 o = {O[4 * g] * inv, O[4 * g + 1] * inv, O[4 * g + 2] * inv, O[4 * g + 3] * inv};
            *reinterpret_cast<f32x4*>(orow + 8 * g) = o;
        }
    }
}

// H[p][j] = U[p][hid + j] * silu(U[p][j])   (mlp.py:17-18 with w1 | w2 as one GEMM)
template <int NV>
__global__ __launch_bounds__(256) void swiglu_kernel(const float* __restrict__ U, int ldu, long long rows, int hid,
                                                    float* __restrict__ H, int ldh) {
    PC_ROW_PROLOGUE(rows)
    Row<NV> a, b;
    a.load(U + (size_t)row * ldu, hid, lane);
    b.load(U + (size_t)row * ldu + hid, hid, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) a.v[i][q] = b.v[i][q] * pc_silu(a.v[i][q]);
    a.store(H + (size_t)row * ldh, hid, lane);
}

// BallPooling input (pooling.py:28-37): Y[i] = [x of the stride children | their positions relative to the centre | 0 pad],
// centre[i] = mean of the children's positions
template <int NV>
__global__ __launch_bounds__(256) void pool_gather_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ P,
                                                         long long rows, int c, int d, int st, float* __restrict__ Y, int ldy,
                                                         float* __restrict__ Pc) {
    PC_ROW_PROLOGUE(rows)
    for (int q = 0; q < st; ++q) {
        Row<NV> r;
        r.load(X + (size_t)(row * st + q) * ldx, c, lane);
        r.store(Y + (size_t)row * ldy + (size_t)q * c, c, lane);
    }
    float ctr[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < d; ++k) {
        float s = P[(size_t)(row * st) * d + k];
        for (int q = 1; q < st; ++q) s += P[(size_t)(row * st + q) * d + k];
        ctr[k] = s / (float)st;
    }
    const int tail0 = st * c, tail = ldy - tail0;
    for (int t = lane; t < tail; t += 64) {
        float v = 0.f;
        if (t < st * d) v = P[(size_t)(row * st + t / d) * d + t % d] - ctr[t % d];
        Y[(size_t)row * ldy + tail0 + t] = v;
    }
    if (lane < d) Pc[(size_t)row * d + lane] = ctr[lane];
}

// BallUnpooling input (pooling.py:58-66): Y[i] = [x_parent | child positions relative to the parent | 0 pad]
template <int NV>
__global__ __launch_bounds__(256) void unpool_input_kernel(const float* __restrict__ Xp, int ldx, const float* __restrict__ Pp,
                                                          const float* __restrict__ Pchild, long long rows, int c, int d, int st,
                                                          float* __restrict__ Y, int ldy) {
    PC_ROW_PROLOGUE(rows)
    Row<NV> r;
    r.load(Xp + (size_t)row * ldx, c, lane);
    r.store(Y + (size_t)row * ldy, c, lane);
    const int tail = ldy - c;
    for (int t = lane; t < tail; t += 64) {
        float v = 0.f;
        if (t < st * d) v = Pchild[(size_t)(row * st + t / d) * d + t % d] - Pp[(size_t)row * d + t % d];
        Y[(size_t)row * ldy + c + t] = v;
    }
}

// child[t] = child_skip[t] + Yp[t / st][(t % st) c ...]   (pooling.py:67)
template <int NV>
__global__ __launch_bounds__(256) void unpool_merge_kernel(const float* __restrict__ skip, int lds_, const float* __restrict__ Yp, int ldy,
                                                          long long rows, int c, int st, float* __restrict__ out, int ldo) {
    PC_ROW_PROLOGUE(rows)
    Row<NV> a, b;
    a.load(skip + (size_t)row * lds_, c, lane);
    b.load(Yp + (size_t)(row / st) * ldy + (size_t)(row % st) * c, c, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i) a.v[i] += b.v[i];
    a.store(out + (size_t)row * ldo, c, lane);
}

// ------------------------------------------------------------------------------------------------ BatchNorm1d
// per-workgroup (count, mean, M2) of every channel over a chunk of rows; combined by bn_finalize_kernel (Chan et al.)
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ X, int ldx, long long rows, int c,
                                                        long long rows_per_block, float* __restrict__ part /*[blocks][2][c]*/) {
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    const float n = (float)(r1 - r0);
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        float s = 0.f;
        for (long long r = r0; r < r1; ++r) s += X[(size_t)r * ldx + ch];
        const float mean = s / n;
        float m2 = 0.f;
        for (long long r = r0; r < r1; ++r) {
            const float dlt = X[(size_t)r * ldx + ch] - mean;
            m2 = fmaf(dlt, dlt, m2);
        }
        part[((size_t)blockIdx.x * 2) * c + ch] = mean;
        part[((size_t)blockIdx.x * 2 + 1) * c + ch] = m2;
    }
}
// stats[0][ch] = mean, stats[1][ch] = biased variance; running statistics updated like nn.BatchNorm1d in train mode
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int blocks, long long rows,
                                                         long long rows_per_block, int c, float momentum,
                                                         float* __restrict__ stats, float* __restrict__ run_mean,
                                                         float* __restrict__ run_var, long long* __restrict__ nbt) {
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch == 0 && nbt) *nbt += 1;
    if (ch >= c) return;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int b = 0; b < blocks; ++b) {
        const long long r0 = (long long)b * rows_per_block;
        const double nb = (double)((r0 + rows_per_block < rows ? r0 + rows_per_block : rows) - r0);
        const double mb = part[((size_t)b * 2) * c + ch], m2b = part[((size_t)b * 2 + 1) * c + ch];
        const double dlt = mb - mean, nn = n + nb;
        mean += dlt * nb / nn;
        m2 += m2b + dlt * dlt * n * nb / nn;
        n = nn;
    }
    const float var = (float)(m2 / n);
    stats[ch] = (float)mean;
    stats[c + ch] = var;
    if (run_mean) {
        run_mean[ch] = (1.f - momentum) * run_mean[ch] + momentum * (float)mean;
        run_var[ch] = (1.f - momentum) * run_var[ch] + momentum * (float)(m2 / (n > 1.0 ? n - 1.0 : 1.0));
    }
}
template <int NV>
__global__ __launch_bounds__(256) void bn_apply_kernel(float* __restrict__ X, int ldx, long long rows, int c,
                                                      const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta) {
    PC_ROW_PROLOGUE(rows)
    Row<NV> r;
    r.load(X + (size_t)row * ldx, c, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int ch = 4 * (lane + 64 * i);
        if (ch < c) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                r.v[i][q] = (r.v[i][q] - mean[ch + q]) * rsqrtf(var[ch + q] + eps) * gamma[ch + q] + beta[ch + q];
        }
    }
    r.store(X + (size_t)row * ldx, c, lane);
}

// ------------------------------------------------------------------------------------------------ MPNN (mpnn.py:48-75)
// Linear([h_j, h_i, pos_j - pos_i]) = A[j] + B[i] + Wc (pos_j - pos_i) + b with A = h Wa^T, B = h Wb^T computed ONCE PER
// NODE (AB = [A | B], one GEMM over N rows instead of one over E edges); this kernel finishes the message per edge --
// GELU, LayerNorm without affine -- and averages the messages that END in node i (scatter_mean over edge_index[1]):
// one wave per receiving node over its CSR segment, no atomics.
// HALF (c <= 128: a row is 32 lanes x 16 B): the two halves of a wave take one receiving node each
template <int NV, bool HALF = false>
__global__ __launch_bounds__(256) void mpnn_message_kernel(const float* __restrict__ AB, int ldab, const float* __restrict__ pos, int d,
                                                          const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
                                                          const float* __restrict__ Wc /*[c][d]*/, const float* __restrict__ bias,
                                                          long long rows, int c, float eps, float* __restrict__ out, int ldo) {
    const int lane = HALF ? (threadIdx.x & 31) : (threadIdx.x & 63);
    const long long wrow = (long long)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    const long long row = HALF ? 2 * wrow + ((threadIdx.x >> 5) & 1) : wrow;
    if (row >= rows) return;
    Row<NV> b, acc;
    b.load(AB + (size_t)row * ldab + c, c, lane);
    f32x4 wc[NV][3];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int ch = 4 * (lane + 64 * i);
        acc.v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (ch < c) b.v[i][q] += bias[ch + q];
            for (int k = 0; k < 3; ++k) wc[i][k][q] = (ch < c && k < d) ? Wc[(size_t)(ch + q) * d + k] : 0.f;
        }
    }
    float pi[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < d; ++k) pi[k] = pos[(size_t)row * d + k];
    const int e0 = rowptr[row], e1 = rowptr[row + 1];
    for (int e = e0; e < e1; ++e) {
        const int j = src[e];
        float dp[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < d; ++k) dp[k] = pos[(size_t)j * d + k] - pi[k];
        Row<NV> m;
        m.load(AB + (size_t)j * ldab, c, lane);
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = m.v[i][q] + b.v[i][q];
                v = fmaf(wc[i][0][q], dp[0], v);
                v = fmaf(wc[i][1][q], dp[1], v);
                v = fmaf(wc[i][2][q], dp[2], v);
                m.v[i][q] = (4 * (lane + 64 * i) < c) ? pc_gelu(v) : 0.f;
            }
        const float mean = red_sum<HALF>(row_partial<NV>(m.v)) / (float)c;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float dv = (4 * (lane + 64 * i) < c) ? m.v[i][q] - mean : 0.f;
                m.v[i][q] = dv;
                ss = fmaf(dv, dv, ss);
            }
        const float rstd = rsqrtf(red_sum<HALF>(ss) / (float)c + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) acc.v[i] += m.v[i] * rstd;
    }
    const float inv = 1.0f / (float)(e1 > e0 ? e1 - e0 : 1);
#pragma unroll
    for (int i = 0; i < NV; ++i) acc.v[i] *= inv;
    acc.store(out + (size_t)row * ldo, c, lane);
}

// h += LayerNorm_noaffine(u); h is also mirrored into a second buffer (the left half of the update GEMM's input)
template <int NV>
__global__ __launch_bounds__(256) void ln_residual_kernel(const float* __restrict__ u, int ldu, long long rows, int c, float eps,
                                                         float* __restrict__ h, int ldh, float* __restrict__ h2, int ldh2) {
    PC_ROW_PROLOGUE(rows)
    Row<NV> r, hv;
    r.load(u + (size_t)row * ldu, c, lane);
    hv.load(h + (size_t)row * ldh, c, lane);
    const float mean = r.sum() / (float)c;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float dv = (4 * (lane + 64 * i) < c) ? r.v[i][q] - mean : 0.f;
            r.v[i][q] = dv;
            ss = fmaf(dv, dv, ss);
        }
    const float rstd = rsqrtf(wave_sum(ss) / (float)c + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) hv.v[i] += r.v[i] * rstd;
    hv.store(h + (size_t)row * ldh, c, lane);
    if (h2) hv.store(h2 + (size_t)row * ldh2, c, lane);
}

// ------------------------------------------------------------------------------------------------ radius graph
// torch_cluster.radius_graph(pos, r, batch, loop=True, max_num_neighbors=32) as erwin_flash.py:160-166 calls it: for
// every centre i the first `max_nbrs` points j of the same cloud, in index order, with |x_j - x_i|^2 < r^2 (fp32,
// squared differences summed dimension by dimension, no fused multiply-add).  One thread per centre; the candidates of
// the workgroup's cloud range are staged through LDS in tiles.
constexpr int kRadTile = 1024;
__global__ __launch_bounds__(256) void radius_kernel(const float* __restrict__ pos, const int32_t* __restrict__ cloud_lo,
                                                    const int32_t* __restrict__ cloud_hi, int N, int d, float r2, int max_nbrs,
                                                    int32_t* __restrict__ nbr, int32_t* __restrict__ cnt) {
    __shared__ float tile[3 * kRadTile];
    __shared__ int s_lo, s_hi, s_live;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool have = i < N;
    const int lo = have ? cloud_lo[i] : 0x7fffffff, hi = have ? cloud_hi[i] : 0;
    float xi[3] = {0.f, 0.f, 0.f};
    if (have)
        for (int k = 0; k < d; ++k) xi[k] = pos[(size_t)i * d + k];
    if (threadIdx.x == 0) s_lo = 0x7fffffff, s_hi = 0;
    __syncthreads();
    atomicMin(&s_lo, lo);
    atomicMax(&s_hi, hi);
    __syncthreads();
    const int blo = s_lo, bhi = s_hi;
    int count = 0;
    for (int t0 = blo; t0 < bhi; t0 += kRadTile) {
        const int tn = min(kRadTile, bhi - t0);
        __syncthreads();
        if (threadIdx.x == 0) s_live = 0;
        for (int q = threadIdx.x; q < tn * d; q += 256) tile[(q % d) * kRadTile + q / d] = pos[(size_t)t0 * d + q];
        __syncthreads();
        if (have && count < max_nbrs) {
            const int j0 = max(lo, t0) - t0, j1 = min(hi, t0 + tn) - t0;
            for (int j = j0; j < j1; ++j) {
                float d2 = 0.f;
                for (int k = 0; k < d; ++k) {
                    const float df = __fsub_rn(tile[k * kRadTile + j], xi[k]);
                    d2 = __fadd_rn(d2, __fmul_rn(df, df));
                }
                if (d2 < r2) {
                    nbr[(size_t)i * max_nbrs + count] = t0 + j;
                    if (++count >= max_nbrs) break;
                }
            }
            if (count < max_nbrs && hi > t0 + tn) s_live = 1;
        }
        __syncthreads();
        if (!s_live) break;  // every centre of this workgroup is full or past its cloud
    }
    if (have) cnt[i] = count;
}
// edge_index [2][E] (int64): row 0 = neighbour j, row 1 = centre i, grouped by centre (flow source_to_target)
__global__ __launch_bounds__(256) void radius_fill_kernel(const int32_t* __restrict__ nbr, const int32_t* __restrict__ cnt,
                                                         const int64_t* __restrict__ rowptr, int N, int max_nbrs, long long E,
                                                         int64_t* __restrict__ edge_index) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const long long e0 = rowptr[i];
    for (int q = 0; q < cnt[i]; ++q) {
        edge_index[e0 + q] = nbr[(size_t)i * max_nbrs + q];
        edge_index[E + e0 + q] = i;
    }
}

// ------------------------------------------------------------------------------------------------ block backward
// y = x * rsqrt(mean(x^2) + eps) * w (+ positional term without x): dx = r (g - xhat mean(xhat g)), g = dy w, xhat = x r.
// dX[slot(p)] (+)= dx of sequence row p.
template <int NV>
__global__ __launch_bounds__(256) void rms_bwd_kernel(const float* __restrict__ X, int ldx, const int32_t* __restrict__ perm,
                                                     const float* __restrict__ dY, int ldy, const float* __restrict__ w,
                                                     long long rows, int c, float* __restrict__ dX, int lddx, int accumulate) {
    PC_ROW_PROLOGUE(rows)
    const long long slot = perm ? perm[row] : row;
    Row<NV> x, g;
    x.load(X + (size_t)slot * ldx, c, lane);
    g.load(dY + (size_t)row * ldy, c, lane);
    const float r = rsqrtf(x.sumsq() / (float)c + FLT_EPSILON);
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int ch = 4 * (lane + 64 * i);
        if (ch < c) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + ch);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                g.v[i][q] *= wv[q];
                x.v[i][q] *= r;  // xhat
                dot = fmaf(x.v[i][q], g.v[i][q], dot);
            }
        }
    }
    dot = wave_sum(dot) / (float)c;
    Row<NV> o;
    if (accumulate) o.load(dX + (size_t)slot * lddx, c, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float d = r * (g.v[i][q] - x.v[i][q] * dot);
            o.v[i][q] = accumulate ? o.v[i][q] + d : d;
        }
    o.store(dX + (size_t)slot * lddx, c, lane);
}

// H = U2 silu(U1): dU1 = dH U2 sig (1 + U1 (1 - sig)), dU2 = dH silu(U1)
template <int NV>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const float* __restrict__ U, int ldu, const float* __restrict__ dH, int ldh,
                                                        long long rows, int hid, float* __restrict__ dU, int lddu) {
    PC_ROW_PROLOGUE(rows)
    Row<NV> a, b, g;
    a.load(U + (size_t)row * ldu, hid, lane);
    b.load(U + (size_t)row * ldu + hid, hid, lane);
    g.load(dH + (size_t)row * ldh, hid, lane);
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float u1 = a.v[i][q], sig = 1.0f / (1.0f + expf(-u1));
            a.v[i][q] = g.v[i][q] * b.v[i][q] * sig * (1.0f + u1 * (1.0f - sig));
            b.v[i][q] = g.v[i][q] * u1 * sig;
        }
    a.store(dU + (size_t)row * lddu, hid, lane);
    b.store(dU + (size_t)row * lddu + hid, hid, lane);
}

// Backward of the per-ball attention (flash-attention style: nothing but qkv is kept from the forward).  One workgroup
// per ball; Q, K, V, dO of all heads in LDS.  Pass A, one (query, head) per thread: row statistics (max, sum), O_i,
// D_i = dO_i . O_i and dQ_i.  Pass B, one (key, head) per thread: dK_j, dV_j over the queries with the stored statistics.
// HD consecutive floats of a 16-B aligned LDS row as 16-B reads
template <int HD>
__device__ __forceinline__ void row_load(const float* p, float (&v)[HD]) {
#pragma unroll
    for (int e = 0; e < HD; e += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p + e);
        v[e] = t[0], v[e + 1] = t[1], v[e + 2] = t[2], v[e + 3] = t[3];
    }
}

template <int HD>
__global__ __launch_bounds__(kAttThreads) void ball_attention_bwd_kernel(const float* __restrict__ qkv, int ldq, const float* __restrict__ dO,
                                                                int ldo, int c, int heads, int ball, float scale,
                                                                float* __restrict__ dqkv, int ldd) {
    const int cs = c + 4;  // LDS row stride: lanes of a wave read DIFFERENT rows at the same channel
    constexpr float kL2e = 1.4426950408889634f;  // scores are kept in log2 units: exp = one v_exp_f32
    float* const Qs = pc_lds;
    float* const Ks = Qs + (size_t)ball * cs;
    float* const Vs = Ks + (size_t)ball * cs;
    float* const Gs = Vs + (size_t)ball * cs;           // dO
    float* const st = Gs + (size_t)ball * cs;           // [heads][ball][3]: max, 1 / sum, D
    const size_t r0 = (size_t)blockIdx.x * ball;
    for (int i = threadIdx.x; i < ball * (c / 4); i += kAttThreads) {
        const int j = i / (c / 4), ch = 4 * (i % (c / 4));
        *reinterpret_cast<f32x4*>(Qs + (size_t)j * cs + ch) = *reinterpret_cast<const f32x4*>(qkv + (r0 + j) * ldq + ch);
        *reinterpret_cast<f32x4*>(Ks + (size_t)j * cs + ch) = *reinterpret_cast<const f32x4*>(qkv + (r0 + j) * ldq + c + ch);
        *reinterpret_cast<f32x4*>(Vs + (size_t)j * cs + ch) = *reinterpret_cast<const f32x4*>(qkv + (r0 + j) * ldq + 2 * c + ch);
        *reinterpret_cast<f32x4*>(Gs + (size_t)j * cs + ch) = *reinterpret_cast<const f32x4*>(dO + (r0 + j) * ldo + ch);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ball * heads; t += kAttThreads) {
        const int h = t / ball, i = t % ball;
        float q[HD], go[HD], acc[HD];
#pragma unroll
        for (int e = 0; e < HD; ++e) q[e] = Qs[(size_t)i * cs + h * HD + e] * (scale * kL2e), go[e] = Gs[(size_t)i * cs + h * HD + e], acc[e] = 0.f;
        float mx = -INFINITY, l = 0.f;
        for (int j = 0; j < ball; ++j) {  // statistics and O_i
            float kj[HD], vj[HD];
            row_load<HD>(Ks + (size_t)j * cs + h * HD, kj);
            row_load<HD>(Vs + (size_t)j * cs + h * HD, vj);
            float sc = 0.f;
#pragma unroll
            for (int e = 0; e < HD; ++e) sc = fmaf(q[e], kj[e], sc);
            const float nm = fmaxf(mx, sc), corr = __builtin_amdgcn_exp2f(mx - nm), pj = __builtin_amdgcn_exp2f(sc - nm);
            l = fmaf(l, corr, pj);
#pragma unroll
            for (int e = 0; e < HD; ++e) acc[e] = fmaf(acc[e], corr, pj * vj[e]);
            mx = nm;
        }
        const float inv = 1.0f / l;
        float D = 0.f;
#pragma unroll
        for (int e = 0; e < HD; ++e) D = fmaf(go[e], acc[e] * inv, D);
        st[(h * ball + i) * 3 + 0] = mx;
        st[(h * ball + i) * 3 + 1] = inv;
        st[(h * ball + i) * 3 + 2] = D;
        float dq[HD];
#pragma unroll
        for (int e = 0; e < HD; ++e) dq[e] = 0.f;
        for (int j = 0; j < ball; ++j) {  // dQ_i = scale sum_j dS_ij K_j
            float kj[HD], vj[HD];
            row_load<HD>(Ks + (size_t)j * cs + h * HD, kj);
            row_load<HD>(Vs + (size_t)j * cs + h * HD, vj);
            float sc = 0.f, dp = 0.f;
#pragma unroll
            for (int e = 0; e < HD; ++e) sc = fmaf(q[e], kj[e], sc), dp = fmaf(go[e], vj[e], dp);
            const float ds = __builtin_amdgcn_exp2f(sc - mx) * inv * (dp - D);
#pragma unroll
            for (int e = 0; e < HD; ++e) dq[e] = fmaf(ds, kj[e], dq[e]);
        }
#pragma unroll
        for (int e = 0; e < HD; e += 4) {
            const f32x4 o = {dq[e] * scale, dq[e + 1] * scale, dq[e + 2] * scale, dq[e + 3] * scale};
            *reinterpret_cast<f32x4*>(dqkv + (r0 + i) * ldd + h * HD + e) = o;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ball * heads; t += kAttThreads) {
        const int h = t / ball, j = t % ball;
        float k[HD], v[HD], dk[HD], dv[HD];
#pragma unroll
        for (int e = 0; e < HD; ++e) k[e] = Ks[(size_t)j * cs + h * HD + e], v[e] = Vs[(size_t)j * cs + h * HD + e], dk[e] = 0.f, dv[e] = 0.f;
        for (int i = 0; i < ball; ++i) {
            float qi[HD], gi[HD];
            row_load<HD>(Qs + (size_t)i * cs + h * HD, qi);
            row_load<HD>(Gs + (size_t)i * cs + h * HD, gi);
            float sc = 0.f, dp = 0.f;
#pragma unroll
            for (int e = 0; e < HD; ++e) sc = fmaf(qi[e], k[e], sc), dp = fmaf(gi[e], v[e], dp);
            const float* s3 = st + (h * ball + i) * 3;
            const float p = __builtin_amdgcn_exp2f(sc * (scale * kL2e) - s3[0]) * s3[1];
            const float ds = p * (dp - s3[2]) * scale;
#pragma unroll
            for (int e = 0; e < HD; ++e) dv[e] = fmaf(p, gi[e], dv[e]), dk[e] = fmaf(ds, qi[e], dk[e]);
        }
#pragma unroll
        for (int e = 0; e < HD; e += 4) {
            const f32x4 a = {dk[e], dk[e + 1], dk[e + 2], dk[e + 3]}, b = {dv[e], dv[e + 1], dv[e + 2], dv[e + 3]};
            *reinterpret_cast<f32x4*>(dqkv + (r0 + j) * ldd + c + h * HD + e) = a;
            *reinterpret_cast<f32x4*>(dqkv + (r0 + j) * ldd + 2 * c + h * HD + e) = b;
        }
    }
}

// u = LayerNorm_noaffine(v): dv = rstd (g - mean(g) - xhat mean(g xhat))
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ V, int ldv, const float* __restrict__ dU, int ldu,
                                                    long long rows, int c, float eps, float* __restrict__ dV, int lddv) {
    PC_ROW_PROLOGUE(rows)
    Row<NV> x, g;
    x.load(V + (size_t)row * ldv, c, lane);
    g.load(dU + (size_t)row * ldu, c, lane);
    const float mean = x.sum() / (float)c;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float dv = (4 * (lane + 64 * i) < c) ? x.v[i][q] - mean : 0.f;
            x.v[i][q] = dv;
            ss = fmaf(dv, dv, ss);
        }
    const float rstd = rsqrtf(wave_sum(ss) / (float)c + eps);
    float gm = 0.f, gx = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            x.v[i][q] *= rstd;  // xhat (0 past c)
            gm += g.v[i][q];
            gx = fmaf(g.v[i][q], x.v[i][q], gx);
        }
    gm = wave_sum(gm) / (float)c;
    gx = wave_sum(gx) / (float)c;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) g.v[i][q] = rstd * (g.v[i][q] - gm - x.v[i][q] * gx);
    g.store(dV + (size_t)row * lddv, c, lane);
}

// Backward of mpnn_message_kernel.  One wave per receiving node i walks its edges again (z, GELU, LayerNorm statistics
// recomputed), with dm_e = dagg_i / deg_i:  dz_e = gelu'(z_e) LN_bwd(dm_e);  dA[src_e] += dz_e (float atomics: a sender has
// edges in many segments),  dB[i] = sum_e dz_e,  and this wave's share of dWc = sum_e dz_e (pos_src - pos_i)^T goes to
// its own row of `wc_part` (persistent waves: the caller adds the rows).  d bias = sum_i dB[i].
template <int NV, bool HALF = false>
__global__ __launch_bounds__(256) void mpnn_message_bwd_kernel(const float* __restrict__ AB, int ldab, const float* __restrict__ pos, int d,
                                                              const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
                                                              const float* __restrict__ Wc, const float* __restrict__ bias,
                                                              const float* __restrict__ dagg, int ldg, long long rows, int c, float eps,
                                                              float* __restrict__ dAB, int ldd, float* __restrict__ wc_part) {
    constexpr int E = 4 * NV;  // channels per lane
    const int lane = HALF ? (threadIdx.x & 31) : (threadIdx.x & 63);
    const long long w0 = (long long)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6), nw0 = (long long)gridDim.x * kRowsPerBlock;
    // (HALF: every half-wave is a worker of its own: one receiving node at a time each)
    const long long wid = HALF ? 2 * w0 + ((threadIdx.x >> 5) & 1) : w0, nw = HALF ? 2 * nw0 : nw0;
    // channel of element x of this lane.  HALF: lane + 32 x -- the 32 lanes of one atomic instruction then add to 128
    // CONSECUTIVE bytes of the sender's row (with four consecutive channels per lane an instruction touches every fourth
    // dword of 512 bytes, and the memory-side adders take a 64-B request for every 16 B of it)
    auto chan = [&](int x) { return HALF ? lane + 32 * x : 4 * (lane + 64 * (x >> 2)) + (x & 3); };
    auto load_row = [&](const float* p, float (&v)[E]) {
        if (HALF) {
#pragma unroll
            for (int x = 0; x < E; ++x) v[x] = chan(x) < c ? p[chan(x)] : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int ch = 4 * (lane + 64 * i);
                const f32x4 t = ch < c ? *reinterpret_cast<const f32x4*>(p + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) v[4 * i + q] = t[q];
            }
        }
    };
    float wc[E][3], dwc[E][3], bs[E];
#pragma unroll
    for (int x = 0; x < E; ++x) {
        const int ch = chan(x);
        bs[x] = ch < c ? bias[ch] : 0.f;
        for (int k = 0; k < 3; ++k) wc[x][k] = (ch < c && k < d) ? Wc[(size_t)ch * d + k] : 0.f, dwc[x][k] = 0.f;
    }
    for (long long row = wid; row < rows; row += nw) {
        float b[E], gi[E], dB[E];
        load_row(AB + (size_t)row * ldab + c, b);
        load_row(dagg + (size_t)row * ldg, gi);
        const int e0 = rowptr[row], e1 = rowptr[row + 1];
        const float inv = 1.0f / (float)(e1 > e0 ? e1 - e0 : 1);
        float pi[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < d; ++k) pi[k] = pos[(size_t)row * d + k];
#pragma unroll
        for (int x = 0; x < E; ++x) b[x] += bs[x], gi[x] *= inv, dB[x] = 0.f;
        for (int e = e0; e < e1; ++e) {
            const int j = src[e];
            float dp[3] = {0.f, 0.f, 0.f};
            for (int k = 0; k < d; ++k) dp[k] = pos[(size_t)j * d + k] - pi[k];
            float z[E], m[E];
            load_row(AB + (size_t)j * ldab, z);
            float sum = 0.f;
#pragma unroll
            for (int x = 0; x < E; ++x) {
                float v = z[x] + b[x];
                v = fmaf(wc[x][0], dp[0], v);
                v = fmaf(wc[x][1], dp[1], v);
                v = fmaf(wc[x][2], dp[2], v);
                const bool live = chan(x) < c;
                z[x] = live ? v : 0.f;
                m[x] = live ? pc_gelu(v) : 0.f;
                sum += m[x];
            }
            const float mean = red_sum<HALF>(sum) / (float)c;
            float ss = 0.f;
#pragma unroll
            for (int x = 0; x < E; ++x) {
                m[x] = chan(x) < c ? m[x] - mean : 0.f;
                ss = fmaf(m[x], m[x], ss);
            }
            const float rstd = rsqrtf(red_sum<HALF>(ss) / (float)c + eps);
            float gm = 0.f, gx = 0.f;
#pragma unroll
            for (int x = 0; x < E; ++x) {
                m[x] *= rstd;  // xhat
                gm += gi[x];
                gx = fmaf(gi[x], m[x], gx);
            }
            gm = red_sum<HALF>(gm) / (float)c;
            gx = red_sum<HALF>(gx) / (float)c;
#pragma unroll
            for (int x = 0; x < E; ++x) {
                const int ch = chan(x);
                const float zz = z[x];
                const float dg = rstd * (gi[x] - gm - m[x] * gx);
                const float gp = 0.5f * (1.0f + erff(zz * 0.70710678118654752440f)) + zz * 0.3989422804014327f * expf(-0.5f * zz * zz);
                const float dz = ch < c ? dg * gp : 0.f;
                dB[x] += dz;
                dwc[x][0] = fmaf(dz, dp[0], dwc[x][0]);
                dwc[x][1] = fmaf(dz, dp[1], dwc[x][1]);
                dwc[x][2] = fmaf(dz, dp[2], dwc[x][2]);
                if (ch < c) atomicAdd(dAB + (size_t)j * ldd + ch, dz);
            }
        }
#pragma unroll
        for (int x = 0; x < E; ++x)
            if (chan(x) < c) dAB[(size_t)row * ldd + c + chan(x)] = dB[x];
    }
    if (wid < nw) {
#pragma unroll
        for (int x = 0; x < E; ++x) {
            const int ch = chan(x);
            if (ch < c)
                for (int k = 0; k < 3; ++k) wc_part[((size_t)wid * c + ch) * 3 + k] = dwc[x][k];
        }
    }
}

// The same edge list through a uniform grid (cells of edge >= r, keyed by (cloud, cell); points sorted by key, so a cell
// lists its points in index order): a centre only looks at the 3^d cells around its own and keeps the max_nbrs SMALLEST
// neighbour indices -- which is what "the first max_nbrs in index order" selects.  O(N * neighbours) instead of O(N^2).
struct RadGrid {
    float mn[3];
    float inv_cell;
    int G[3];
};
__device__ __forceinline__ int rad_cell(float x, float mn, float inv_cell, int G) {
    const int c = (int)floorf((x - mn) * inv_cell);
    return c < 0 ? 0 : (c >= G ? G - 1 : c);
}
__global__ __launch_bounds__(256) void radius_cellkey_kernel(const float* __restrict__ pos, const int32_t* __restrict__ cloud_of, int N,
                                                            int d, RadGrid g, int64_t* __restrict__ key) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    long long k = cloud_of[i];
    for (int a = d - 1; a >= 0; --a) k = k * g.G[a] + rad_cell(pos[(size_t)i * d + a], g.mn[a], g.inv_cell, g.G[a]);
    key[i] = k;
}
constexpr int kRadList = 33;  // ints per thread in LDS (max_nbrs <= 32, + 1 against bank conflicts)
__global__ __launch_bounds__(256) void radius_grid_kernel(const float* __restrict__ pos, const int32_t* __restrict__ cloud_of,
                                                         const int64_t* __restrict__ sorted_idx, const int32_t* __restrict__ cell_ptr,
                                                         int N, int d, RadGrid g, float r2, int max_nbrs,
                                                         int32_t* __restrict__ nbr, int32_t* __restrict__ cnt) {
    __shared__ int lst[256 * kRadList];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= N) return;
    int* const mine = lst + threadIdx.x * kRadList;
    const int i = (int)sorted_idx[t];
    float xi[3] = {0.f, 0.f, 0.f};
    int ci[3] = {0, 0, 0};
    for (int k = 0; k < d; ++k) {
        xi[k] = pos[(size_t)i * d + k];
        ci[k] = rad_cell(xi[k], g.mn[k], g.inv_cell, g.G[k]);
    }
    const long long cloud_base = (long long)cloud_of[i] * g.G[0] * g.G[1] * (d == 3 ? g.G[2] : 1);
    int count = 0;
    const int z0 = d == 3 ? max(ci[2] - 1, 0) : 0, z1 = d == 3 ? min(ci[2] + 1, g.G[2] - 1) : 0;
    for (int cz = z0; cz <= z1; ++cz)
        for (int cy = max(ci[1] - 1, 0); cy <= min(ci[1] + 1, g.G[1] - 1); ++cy)
            for (int cx = max(ci[0] - 1, 0); cx <= min(ci[0] + 1, g.G[0] - 1); ++cx) {
                const long long cell = cloud_base + ((long long)cz * g.G[1] + cy) * g.G[0] + cx;
                const int e0 = cell_ptr[cell], e1 = cell_ptr[cell + 1];
                for (int e = e0; e < e1; ++e) {
                    const int j = (int)sorted_idx[e];
                    if (count == max_nbrs && j > mine[count - 1]) break;  // the cell lists ascending indices
                    float d2 = 0.f;
                    for (int k = 0; k < d; ++k) {
                        const float df = __fsub_rn(pos[(size_t)j * d + k], xi[k]);
                        d2 = __fadd_rn(d2, __fmul_rn(df, df));
                    }
                    if (!(d2 < r2)) continue;
                    int q = count < max_nbrs ? count : max_nbrs - 1;  // insert j into the ascending list
                    while (q > 0 && mine[q - 1] > j) {
                        mine[q] = mine[q - 1];
                        --q;
                    }
                    mine[q] = j;
                    if (count < max_nbrs) ++count;
                }
            }
    for (int q = 0; q < count; ++q) nbr[(size_t)i * max_nbrs + q] = mine[q];
    cnt[i] = count;
}

}  // namespace haet

using namespace haet;

#define PC_CHECK_WIDTH(name, c) \
    if (!width_ok(c)) return fail(HAET_E_ARG, name ": width %d (multiple of 4, <= 1024)", c)

extern "C" int haet_pc_gather_rows(const float* x, int ldx, const int32_t* idx, long long rows, int c, float* out, int ldo,
                                   haet_stream_t stream) {
    if (!x || !idx || !out || rows < 1) return fail(HAET_E_ARG, "pc_gather_rows: bad arguments");
    PC_CHECK_WIDTH("pc_gather_rows", c);
    PC_DISPATCH(c, gather_rows_kernel, row_grid(rows), dim3(256), 0, as_stream(stream), x, ldx, idx, rows, c, out, ldo);
    return check_launch("gather_rows_kernel");
}

extern "C" int haet_pc_scatter_masked(const float* X, int ldx, const int32_t* tree_idx, const uint8_t* mask, long long rows, int c,
                                      float* out, int ldo, haet_stream_t stream) {
    if (!X || !tree_idx || !mask || !out || rows < 1) return fail(HAET_E_ARG, "pc_scatter_masked: bad arguments");
    PC_CHECK_WIDTH("pc_scatter_masked", c);
    PC_DISPATCH(c, scatter_masked_kernel, row_grid(rows), dim3(256), 0, as_stream(stream), X, ldx, tree_idx, mask, rows, c, out, ldo);
    return check_launch("scatter_masked_kernel");
}

extern "C" int haet_pc_rows_add(float* X, int ldx, const int32_t* perm, const float* R, int ldr, long long rows, int c,
                                haet_stream_t stream) {
    if (!X || !R || rows < 1) return fail(HAET_E_ARG, "pc_rows_add: bad arguments");
    PC_CHECK_WIDTH("pc_rows_add", c);
    PC_DISPATCH(c, rows_add_kernel, row_grid(rows), dim3(256), 0, as_stream(stream), X, ldx, perm, R, ldr, rows, c);
    return check_launch("rows_add_kernel");
}

extern "C" int haet_pc_ball_rel(const float* P, const int32_t* perm, long long rows, int ball, int d, float* rel, haet_stream_t stream) {
    if (!P || !rel || rows < 1 || ball < 1 || rows % ball || (d != 2 && d != 3)) return fail(HAET_E_ARG, "pc_ball_rel: bad arguments");
    hipLaunchKernelGGL(ball_rel_kernel, row_grid(rows / ball), dim3(256), 0, as_stream(stream), P, perm, rows / ball, ball, d, rel);
    return check_launch("ball_rel_kernel");
}

extern "C" int haet_pc_rms_pe(const float* X, int ldx, const int32_t* perm, const float* rel, long long rows, int c, int d,
                              const float* w, const float* pe_w, const float* pe_b, float* Y, int ldy, haet_stream_t stream) {
    if (!X || !w || !Y || rows < 1 || (pe_w && (!pe_b || !rel))) return fail(HAET_E_ARG, "pc_rms_pe: bad arguments");
    PC_CHECK_WIDTH("pc_rms_pe", c);
    PC_DISPATCH(c, rms_pe_kernel, row_grid(rows), dim3(256), 0, as_stream(stream), X, ldx, perm, rel, rows, c, d, w, pe_w, pe_b, Y, ldy);
    return check_launch("rms_pe_kernel");
}

extern "C" int haet_pc_ball_attention_supported(int c, int heads, int ball) {
    if (c < 8 || heads < 1 || c % heads || c % 4) return 0;
    const int hd = c / heads;
    if (hd != 8 && hd != 16 && hd != 32 && hd != 64) return 0;
    return (ball >= 1 && (size_t)ball * c * 8 <= 144 * 1024) ? 1 : 0;
}

extern "C" int haet_pc_ball_attention(const float* qkv, int ldq, long long rows, int c, int heads, int ball, float* out, int ldo,
                                      haet_stream_t stream) {
    if (!qkv || !out || rows < 1 || ball < 1 || rows % ball) return fail(HAET_E_ARG, "pc_ball_attention: bad arguments");
    if (!haet_pc_ball_attention_supported(c, heads, ball))
        return fail(HAET_E_ARG, "pc_ball_attention: c=%d heads=%d ball=%d unsupported", c, heads, ball);
    const int hd = c / heads;
    const size_t lds = (size_t)ball * c * 8;
    const float scale = 1.0f / sqrtf((float)hd);
    auto go = [&](auto kern) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)(rows / ball)), dim3(kAttThreads), lds, as_stream(stream), qkv, ldq, c, heads, ball, scale, out, ldo);
    };
    // matrix-core kernel: balls of 32 / 64 / 128 tokens, head dims 8 / 16 / 32 (HAET_PC_ATT_VALU=1: the vector-ALU kernel, A/B)
    static const bool valu = getenv("HAET_PC_ATT_VALU") != nullptr;
    if (!valu && (ball == 32 || ball == 64 || (ball == 128 && hd != 16)) && (hd == 8 || hd == 16 || hd == 32) && ldq % 4 == 0 &&
        ldo % 4 == 0) {  // (ball 128 at head dim 16 spills 28 registers: it keeps the vector-ALU kernel)
        auto gom = [&](auto kern) {
            if (lds > 64 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3((unsigned)(rows / ball)), dim3(kAttThreads), lds, as_stream(stream), qkv, ldq, c, heads, scale, out, ldo);
        };
#define HAET_PC_ATT(HD_)                                            \
    if (hd == HD_) {                                                \
        if (ball == 32) gom(ball_attention_mfma_kernel<HD_, 1>);    \
        else if (ball == 64) gom(ball_attention_mfma_kernel<HD_, 2>); \
        else gom(ball_attention_mfma_kernel<HD_, 4>);               \
    }
        HAET_PC_ATT(8) HAET_PC_ATT(16) HAET_PC_ATT(32)
#undef HAET_PC_ATT
        return check_launch("ball_attention_mfma_kernel");
    }
    switch (hd) {
        case 8: go(ball_attention_kernel<8>); break;
        case 16: go(ball_attention_kernel<16>); break;
        case 32: go(ball_attention_kernel<32>); break;
        default: go(ball_attention_kernel<64>); break;
    }
    return check_launch("ball_attention_kernel");
}

extern "C" int haet_pc_swiglu(const float* U, int ldu, long long rows, int hid, float* H, int ldh, haet_stream_t stream) {
    if (!U || !H || rows < 1) return fail(HAET_E_ARG, "pc_swiglu: bad arguments");
    PC_CHECK_WIDTH("pc_swiglu", hid);
    PC_DISPATCH(hid, swiglu_kernel, row_grid(rows), dim3(256), 0, as_stream(stream), U, ldu, rows, hid, H, ldh);
    return check_launch("swiglu_kernel");
}

extern "C" int haet_pc_pool_gather(const float* X, int ldx, const float* P, long long rows, int c, int d, int st, float* Y, int ldy,
                                   float* Pc, haet_stream_t stream) {
    if (!X || !P || !Y || !Pc || rows < 1 || st < 1 || ldy < st * (c + d)) return fail(HAET_E_ARG, "pc_pool_gather: bad arguments");
    PC_CHECK_WIDTH("pc_pool_gather", c);
    PC_DISPATCH(c, pool_gather_kernel, row_grid(rows), dim3(256), 0, as_stream(stream), X, ldx, P, rows, c, d, st, Y, ldy, Pc);
    return check_launch("pool_gather_kernel");
}

extern "C" int haet_pc_unpool_input(const float* Xp, int ldx, const float* Pp, const float* Pchild, long long rows, int c, int d, int st,
                                    float* Y, int ldy, haet_stream_t stream) {
    if (!Xp || !Pp || !Pchild || !Y || rows < 1 || st < 1 || ldy < c + st * d) return fail(HAET_E_ARG, "pc_unpool_input: bad arguments");
    PC_CHECK_WIDTH("pc_unpool_input", c);
    PC_DISPATCH(c, unpool_input_kernel, row_grid(rows), dim3(256), 0, as_stream(stream), Xp, ldx, Pp, Pchild, rows, c, d, st, Y, ldy);
    return check_launch("unpool_input_kernel");
}

extern "C" int haet_pc_unpool_merge(const float* skip, int lds_, const float* Yp, int ldy, long long rows, int c, int st, float* out,
                                    int ldo, haet_stream_t stream) {
    if (!skip || !Yp || !out || rows < 1 || st < 1 || rows % st) return fail(HAET_E_ARG, "pc_unpool_merge: bad arguments");
    PC_CHECK_WIDTH("pc_unpool_merge", c);
    PC_DISPATCH(c, unpool_merge_kernel, row_grid(rows), dim3(256), 0, as_stream(stream), skip, lds_, Yp, ldy, rows, c, st, out, ldo);
    return check_launch("unpool_merge_kernel");
}

extern "C" int haet_pc_bn_blocks(long long rows) {
    const long long b = (rows + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

extern "C" int haet_pc_bn_stats(const float* X, int ldx, long long rows, int c, float momentum, float* part, float* stats,
                                float* run_mean, float* run_var, long long* nbt, haet_stream_t stream) {
    if (!X || !part || !stats || rows < 1 || c < 1) return fail(HAET_E_ARG, "pc_bn_stats: bad arguments");
    const int blocks = haet_pc_bn_blocks(rows);
    const long long rpb = (rows + blocks - 1) / blocks;
    const int used = (int)((rows + rpb - 1) / rpb);
    hipLaunchKernelGGL(bn_partial_kernel, dim3(used), dim3(256), 0, as_stream(stream), X, ldx, rows, c, rpb, part);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 255) / 256), dim3(256), 0, as_stream(stream), part, used, rows, rpb, c, momentum,
                       stats, run_mean, run_var, nbt);
    return check_launch("bn_stats");
}

extern "C" int haet_pc_bn_apply(float* X, int ldx, long long rows, int c, const float* mean, const float* var, float eps,
                                const float* gamma, const float* beta, haet_stream_t stream) {
    if (!X || !mean || !var || !gamma || !beta || rows < 1) return fail(HAET_E_ARG, "pc_bn_apply: bad arguments");
    PC_CHECK_WIDTH("pc_bn_apply", c);
    PC_DISPATCH(c, bn_apply_kernel, row_grid(rows), dim3(256), 0, as_stream(stream), X, ldx, rows, c, mean, var, eps, gamma, beta);
    return check_launch("bn_apply_kernel");
}

extern "C" int haet_pc_mpnn_message(const float* AB, int ldab, const float* pos, int d, const int32_t* rowptr, const int32_t* src,
                                    const float* Wc, const float* bias, long long rows, int c, float eps, float* out, int ldo,
                                    haet_stream_t stream) {
    if (!AB || !pos || !rowptr || !src || !Wc || !bias || !out || rows < 1 || (d != 2 && d != 3))
        return fail(HAET_E_ARG, "pc_mpnn_message: bad arguments");
    PC_CHECK_WIDTH("pc_mpnn_message", c);
    if (c <= 128) {
        hipLaunchKernelGGL((mpnn_message_kernel<1, true>), row_grid((rows + 1) / 2), dim3(256), 0, as_stream(stream), AB, ldab, pos, d,
                           rowptr, src, Wc, bias, rows, c, eps, out, ldo);
        return check_launch("mpnn_message_kernel<half>");
    }
    PC_DISPATCH(c, mpnn_message_kernel, row_grid(rows), dim3(256), 0, as_stream(stream), AB, ldab, pos, d, rowptr, src, Wc, bias, rows,
                c, eps, out, ldo);
    return check_launch("mpnn_message_kernel");
}

extern "C" int haet_pc_ln_residual(const float* u, int ldu, long long rows, int c, float eps, float* h, int ldh, float* h2, int ldh2,
                                   haet_stream_t stream) {
    if (!u || !h || rows < 1) return fail(HAET_E_ARG, "pc_ln_residual: bad arguments");
    PC_CHECK_WIDTH("pc_ln_residual", c);
    PC_DISPATCH(c, ln_residual_kernel, row_grid(rows), dim3(256), 0, as_stream(stream), u, ldu, rows, c, eps, h, ldh, h2, ldh2);
    return check_launch("ln_residual_kernel");
}

extern "C" int haet_pc_radius_neighbours(const float* pos, const int32_t* cloud_lo, const int32_t* cloud_hi, int N, int d, float r,
                                         int max_nbrs, int32_t* nbr, int32_t* cnt, haet_stream_t stream) {
    if (!pos || !cloud_lo || !cloud_hi || !nbr || !cnt || N < 1 || (d != 2 && d != 3) || max_nbrs < 1)
        return fail(HAET_E_ARG, "pc_radius_neighbours: bad arguments");
    hipLaunchKernelGGL(radius_kernel, dim3((N + 255) / 256), dim3(256), 0, as_stream(stream), pos, cloud_lo, cloud_hi, N, d, r * r,
                       max_nbrs, nbr, cnt);
    return check_launch("radius_kernel");
}

extern "C" int haet_pc_radius_fill(const int32_t* nbr, const int32_t* cnt, const int64_t* rowptr, int N, int max_nbrs, long long E,
                                   int64_t* edge_index, haet_stream_t stream) {
    if (!nbr || !cnt || !rowptr || !edge_index || N < 1 || E < 0) return fail(HAET_E_ARG, "pc_radius_fill: bad arguments");
    if (E == 0) return HAET_OK;
    hipLaunchKernelGGL(radius_fill_kernel, dim3((N + 255) / 256), dim3(256), 0, as_stream(stream), nbr, cnt, rowptr, N, max_nbrs, E,
                       edge_index);
    return check_launch("radius_fill_kernel");
}

/* grid variant: key [N] int64 = (cloud, cell) of every point for the caller's stable sort; then the neighbour search on
 * sorted_idx (point ids by key) and cell_ptr (first sorted position of every key, n_clouds * G0 G1 G2 + 1 entries).   */
static RadGrid rad_grid(const float* mn, float cell, const int32_t* G) {
    RadGrid g;
    for (int k = 0; k < 3; ++k) g.mn[k] = mn[k], g.G[k] = G[k];
    g.inv_cell = 1.0f / cell;
    return g;
}
extern "C" int haet_pc_radius_cell_keys(const float* pos, const int32_t* cloud_of, int N, int d, const float* mn, float cell,
                                        const int32_t* G, int64_t* key, haet_stream_t stream) {
    if (!pos || !cloud_of || !mn || !G || !key || N < 1 || (d != 2 && d != 3) || !(cell > 0.f))
        return fail(HAET_E_ARG, "pc_radius_cell_keys: bad arguments");
    hipLaunchKernelGGL(radius_cellkey_kernel, dim3((N + 255) / 256), dim3(256), 0, as_stream(stream), pos, cloud_of, N, d,
                       rad_grid(mn, cell, G), key);
    return check_launch("radius_cellkey_kernel");
}
extern "C" int haet_pc_radius_neighbours_grid(const float* pos, const int32_t* cloud_of, const int64_t* sorted_idx,
                                              const int32_t* cell_ptr, int N, int d, const float* mn, float cell, const int32_t* G,
                                              float r, int max_nbrs, int32_t* nbr, int32_t* cnt, haet_stream_t stream) {
    if (!pos || !cloud_of || !sorted_idx || !cell_ptr || !mn || !G || !nbr || !cnt || N < 1 || (d != 2 && d != 3))
        return fail(HAET_E_ARG, "pc_radius_neighbours_grid: bad arguments");
    if (max_nbrs < 1 || max_nbrs > kRadList - 1) return fail(HAET_E_ARG, "pc_radius_neighbours_grid: max_nbrs=%d (<= 32)", max_nbrs);
    if (!(cell >= r)) return fail(HAET_E_ARG, "pc_radius_neighbours_grid: cell %g < r %g", cell, r);
    hipLaunchKernelGGL(radius_grid_kernel, dim3((N + 255) / 256), dim3(256), 0, as_stream(stream), pos, cloud_of, sorted_idx, cell_ptr,
                       N, d, rad_grid(mn, cell, G), r * r, max_nbrs, nbr, cnt);
    return check_launch("radius_grid_kernel");
}

/* backward stages of an Erwin block on the point-cloud path (training): see haet_amd/components/erwinflash/pointcloud.py */
extern "C" int haet_pc_rms_bwd(const float* X, int ldx, const int32_t* perm, const float* dY, int ldy, const float* w, long long rows,
                               int c, float* dX, int lddx, int accumulate, haet_stream_t stream) {
    if (!X || !dY || !w || !dX || rows < 1) return fail(HAET_E_ARG, "pc_rms_bwd: bad arguments");
    PC_CHECK_WIDTH("pc_rms_bwd", c);
    PC_DISPATCH(c, rms_bwd_kernel, row_grid(rows), dim3(256), 0, as_stream(stream), X, ldx, perm, dY, ldy, w, rows, c, dX, lddx, accumulate);
    return check_launch("rms_bwd_kernel");
}

extern "C" int haet_pc_swiglu_bwd(const float* U, int ldu, const float* dH, int ldh, long long rows, int hid, float* dU, int lddu,
                                  haet_stream_t stream) {
    if (!U || !dH || !dU || rows < 1) return fail(HAET_E_ARG, "pc_swiglu_bwd: bad arguments");
    PC_CHECK_WIDTH("pc_swiglu_bwd", hid);
    PC_DISPATCH(hid, swiglu_bwd_kernel, row_grid(rows), dim3(256), 0, as_stream(stream), U, ldu, dH, ldh, rows, hid, dU, lddu);
    return check_launch("swiglu_bwd_kernel");
}

extern "C" int haet_pc_ball_attention_bwd_supported(int c, int heads, int ball) {
    if (!haet_pc_ball_attention_supported(c, heads, ball)) return 0;
    return ((size_t)ball * (c + 4) * 16 + (size_t)heads * ball * 12 <= 156 * 1024) ? 1 : 0;
}

extern "C" int haet_pc_ball_attention_bwd(const float* qkv, int ldq, const float* dO, int ldo, long long rows, int c, int heads, int ball,
                                          float* dqkv, int ldd, haet_stream_t stream) {
    if (!qkv || !dO || !dqkv || rows < 1 || ball < 1 || rows % ball) return fail(HAET_E_ARG, "pc_ball_attention_bwd: bad arguments");
    if (!haet_pc_ball_attention_bwd_supported(c, heads, ball))
        return fail(HAET_E_ARG, "pc_ball_attention_bwd: c=%d heads=%d ball=%d unsupported", c, heads, ball);
    const int hd = c / heads;
    const size_t lds = (size_t)ball * (c + 4) * 16 + (size_t)heads * ball * 12;
    const float scale = 1.0f / sqrtf((float)hd);
    auto go = [&](auto kern) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)(rows / ball)), dim3(kAttThreads), lds, as_stream(stream), qkv, ldq, dO, ldo, c, heads, ball,
                           scale, dqkv, ldd);
    };
    switch (hd) {
        case 8: go(ball_attention_bwd_kernel<8>); break;
        case 16: go(ball_attention_bwd_kernel<16>); break;
        case 32: go(ball_attention_bwd_kernel<32>); break;
        default: go(ball_attention_bwd_kernel<64>); break;
    }
    return check_launch("ball_attention_bwd_kernel");
}

extern "C" int haet_pc_ln_bwd(const float* V, int ldv, const float* dU, int ldu, long long rows, int c, float eps, float* dV, int lddv,
                              haet_stream_t stream) {
    if (!V || !dU || !dV || rows < 1) return fail(HAET_E_ARG, "pc_ln_bwd: bad arguments");
    PC_CHECK_WIDTH("pc_ln_bwd", c);
    PC_DISPATCH(c, ln_bwd_kernel, row_grid(rows), dim3(256), 0, as_stream(stream), V, ldv, dU, ldu, rows, c, eps, dV, lddv);
    return check_launch("ln_bwd_kernel");
}

/* wc_part: [haet_pc_mpnn_bwd_waves(rows)][c][3] floats (one row per persistent wave; the caller adds them);
 * dAB [rows][ldd]: the A half (columns [0, c)) must be zero on entry (float atomics), the B half is written. */
static int mpnn_bwd_blocks(long long rows) {
    const long long blocks = (rows + kRowsPerBlock - 1) / kRowsPerBlock;
    return (int)(blocks < 2048 ? blocks : 2048);
}
/* rows of wc_part (zero-initialised by the caller: the half-wave variant fills all of them, the others the first half) */
extern "C" int haet_pc_mpnn_bwd_waves(long long rows) { return 2 * mpnn_bwd_blocks(rows) * kRowsPerBlock; }

extern "C" int haet_pc_mpnn_message_bwd(const float* AB, int ldab, const float* pos, int d, const int32_t* rowptr, const int32_t* src,
                                        const float* Wc, const float* bias, const float* dagg, int ldg, long long rows, int c, float eps,
                                        float* dAB, int ldd, float* wc_part, haet_stream_t stream) {
    if (!AB || !pos || !rowptr || !src || !Wc || !bias || !dagg || !dAB || !wc_part || rows < 1 || (d != 2 && d != 3))
        return fail(HAET_E_ARG, "pc_mpnn_message_bwd: bad arguments");
    if (!width_ok(c) || c > 512) return fail(HAET_E_ARG, "pc_mpnn_message_bwd: width %d (multiple of 4, <= 512)", c);
    const dim3 grid((unsigned)mpnn_bwd_blocks(rows));
    if (c <= 128)
        hipLaunchKernelGGL((mpnn_message_bwd_kernel<1, true>), grid, dim3(256), 0, as_stream(stream), AB, ldab, pos, d, rowptr, src, Wc,
                           bias, dagg, ldg, rows, c, eps, dAB, ldd, wc_part);
    else if (c <= 256)
        hipLaunchKernelGGL((mpnn_message_bwd_kernel<1>), grid, dim3(256), 0, as_stream(stream), AB, ldab, pos, d, rowptr, src, Wc, bias,
                           dagg, ldg, rows, c, eps, dAB, ldd, wc_part);
    else
        hipLaunchKernelGGL((mpnn_message_bwd_kernel<2>), grid, dim3(256), 0, as_stream(stream), AB, ldab, pos, d, rowptr, src, Wc, bias,
                           dagg, ldg, rows, c, eps, dAB, ldd, wc_part);
    return check_launch("mpnn_message_bwd_kernel");
}
