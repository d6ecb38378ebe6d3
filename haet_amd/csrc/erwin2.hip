// Token stage, second generation ("fragment kernel"): positions + ball tree + the whole Erwin transformer of one
// (batch, head) cloud in ONE workgroup, restructured around WAVE-OWNED work items so that a transformer block is two
// phases instead of eight (models/components/erwinflash/erwin_flash.py:138-195; layers.py:30-32,72-92;
// attention.py:58-88; mlp.py:17-18; pooling.py:25-72):
//
//   attention phase   item = (32-token tile, head):  x = X + pending -> RMSNorm + rel-pos PE -> q, k, v of ITS head ->
//                     softmax(q k^T) v -> its head's slice of the proj product (split-K over the heads) -> partial slab
//   MLP phase         item = (32-token tile, 32 hidden units):  x = X + b_proj + sum of the heads' partials -> RMSNorm ->
//                     its 32 rows of w1 / w2 -> SwiGLU gate -> its K-slice of the w3 product -> partial slab
//
// Every product is a v_mfma_f32_32x32x16_bf16 triple (split bf16: hi*hi + hi*lo + lo*hi, fp32 accumulate) whose operands
// never leave the wave's registers: the token tile stays in the ACCUMULATOR layout (lane = token, registers = channels)
// from the load to the partial sums, and an accumulator is the next product's operand as it stands (the k index of a
// step is then permuted: element j of lane half h is row 16s + 8(j>>2) + 4h + (j&3) -- the weights are re-laid on the
// host side of this file into exactly that order, one 16-B load per lane and plane: haet_erwin_frag_pack).  V is
// computed with the operand roles swapped (tokens as A), which lands it transposed -- feature on the lane -- as the
// PV product wants it.  Nothing but the split-K partial sums and the token state X goes through LDS, and the only
// workgroup barriers are one per phase plus a cheap one between a phase's loads and its stores (all waves arrive
// together), 4 per block against ~10 in erwin.hip, with none of its ~50 LDS round trips per block.
//
// Envelope (haet_erwin_frag_bytes != 0): eval-mode BatchNorm, head dim 8, channels 32 or 64, 32 | tokens per level,
// (tokens / 32) * heads <= 8 and (tokens / 32) * hidden / 32 <= 8 (every item of a phase on its own wave), power-of-two
// balls <= 32, stride 2, power-of-two clouds <= 64 tokens (the wave tree builder).  BASELINE config C3 (M = 64: 64 x 32
// and 32 x 64 tokens x channels) is inside; everything else, training (tape, batch statistics) included, stays on
// erwin.hip.
#include "common.h"
#include "balltree_wave_dev.h"

namespace haet {
namespace e2 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef const __attribute__((address_space(1))) u32x4* gv4;
typedef const __attribute__((address_space(1))) f32x4* gf4;

constexpr int kThreads = 512;
constexpr int kMaxOps = 16;
constexpr int kPad = 4;  // fp32 row padding of the token state (floats)
enum { OP_BLOCK = 0, OP_FLUSH = 1, OP_POOL = 2, OP_UNPOOL = 3 };

__host__ __device__ inline int round32(int k) { return (k + 31) & ~31; }

struct Op {
    int kind, lvl;             // level the op works on (pool / unpool: the finer level)
    int n, c, nc;              // tokens, channels, channels / 32 of that level
    int heads, ball, chunks;   // block: heads, ball size, hidden / 32
    int hid, rot, pbuf;        // block: hidden width, 1 = rotated partition, parameter buffer (0 / 1)
    int small_off, small_len;  // the block's small-parameter group in the v1 blob (floats)
    int norm1, norm2, pe_w, pe_b, qkv_b, proj_b, w1_b, w2_b, w3_b;  // float offsets in the v1 blob
    int qkv_w, proj_w, w1_w, w2_w, w3_w;                            // (for the packer)
    unsigned qk, v, pj, w1, w2, w3;  // byte offsets of the fragment sections in the fragment blob
    int next_off, next_len;          // small-parameter group of the NEXT block to stage (len 0: none)
    int w, b, bn_scale, bn_shift, stride, c2;  // resample: v1 offsets, stride, width of the coarse level
};

struct Dir {  // LDS byte offsets
    int X[HAET_ERWIN_MAX_LEVELS], ld[HAET_ERWIN_MAX_LEVELS], n[HAET_ERWIN_MAX_LEVELS], P[HAET_ERWIN_MAX_LEVELS];
    int rel[HAET_ERWIN_MAX_LEVELS][2], rotm[HAET_ERWIN_MAX_LEVELS], invr[HAET_ERWIN_MAX_LEVELS];
    int ident, tmap, par[2], slab, total;
};

struct Plan {
    int n_ops, n_levels, d, M, c0, rotate;
    int ball[HAET_ERWIN_MAX_LEVELS], stride[HAET_ERWIN_MAX_LEVELS];
    unsigned frag_begin, total_bytes;
    Dir dir;
    Op ops[kMaxOps];
};
constexpr unsigned kPlanBytes = (sizeof(Plan) + 1023) & ~1023u;

__device__ __forceinline__ float& F(int off) { return *reinterpret_cast<float*>(smem + off); }
__device__ __forceinline__ int& I(int off) { return *reinterpret_cast<int*>(smem + off); }
__device__ __forceinline__ f32x4& F4(int off) { return *reinterpret_cast<f32x4*>(smem + off); }

// Workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains the vector-memory counter,
// i.e. the weight fragments requested at the top of a phase -- the loads this structure wants in flight ACROSS the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (RNE)
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned h = pack2_bf16(v[2 * i], v[2 * i + 1]);
        hi[i] = h;
        lo[i] = pack2_bf16(v[2 * i] - __uint_as_float(h << 16), v[2 * i + 1] - __uint_as_float(h & 0xffff0000u));
    }
}
// registers 8s .. 8s+7 of an accumulator as the {hi, lo} fragments of k-step s
template <int S>
__device__ __forceinline__ void acc_frag(const f32x16& a, u32x4& hi, u32x4& lo) {
    const float v[8] = {a[8 * S], a[8 * S + 1], a[8 * S + 2], a[8 * S + 3], a[8 * S + 4], a[8 * S + 5], a[8 * S + 6], a[8 * S + 7]};
    split8(v, hi, lo);
}
__device__ __forceinline__ void quad_frag(float a, float b, float c, float d, u32x4& hi, u32x4& lo) {  // k-slots 4..7 zero
    const float v[8] = {a, b, c, d, 0.f, 0.f, 0.f, 0.f};
    split8(v, hi, lo);
}
__device__ __forceinline__ f32x16 mfma3(u32x4 ah, u32x4 al, u32x4 bh, u32x4 bl, f32x16 c) {
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, ah), a1 = __builtin_bit_cast(bf16x8, al);
    const bf16x8 b0 = __builtin_bit_cast(bf16x8, bh), b1 = __builtin_bit_cast(bf16x8, bl);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}
__device__ __forceinline__ u32x4 ldg16(const char* p) { return *(gv4)p; }

// What the previous phase left to be added to X before anyone may use it: a bias row and `np` partial slabs per
// 32-token tile, written in the PREVIOUS phase's place order (`inv` maps a token to its place there).
struct Pending {
    int np, bias, inv;  // np = 0: nothing pending; bias, inv: LDS byte offsets
};

// x (accumulator layout: lane = (token place p, half hf), register 4q+i of tile nc = channel 32nc + 8q + 4hf + i) of
// token `tau` = X[tau] + pending
template <int NC>
__device__ __forceinline__ void load_x(f32x16 (&x)[NC], int X, int ld, int tau, int hf, const Pending& pd, int slab) {
    constexpr int slabB = NC * 4096;
    const int xrow = X + 4 * (tau * ld + 4 * hf);
#pragma unroll
    for (int nc = 0; nc < NC; ++nc)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = F4(xrow + 4 * (32 * nc + 8 * q));
#pragma unroll
            for (int i = 0; i < 4; ++i) x[nc][4 * q + i] = v[i];
        }
    if (pd.np) {  // (wave-uniform)
        const int pp = I(pd.inv + 4 * tau);
        const int sl = slab + (pp >> 5) * pd.np * slabB + ((pp & 31) + 32 * hf) * 16;
#pragma unroll
        for (int nc = 0; nc < NC; ++nc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = F4(pd.bias + 4 * (32 * nc + 8 * q + 4 * hf));
#pragma unroll
                for (int i = 0; i < 4; ++i) x[nc][4 * q + i] += b[i];
            }
        for (int k = 0; k < pd.np; ++k) {
#pragma unroll
            for (int nc = 0; nc < NC; ++nc)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = F4(sl + k * slabB + (nc * 4 + q) * 1024);
#pragma unroll
                    for (int i = 0; i < 4; ++i) x[nc][4 * q + i] += v[i];
                }
        }
    }
}
template <int NC>
__device__ __forceinline__ void store_x(const f32x16 (&x)[NC], int X, int ld, int tau, int hf) {
    const int xrow = X + 4 * (tau * ld + 4 * hf);
#pragma unroll
    for (int nc = 0; nc < NC; ++nc)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {x[nc][4 * q], x[nc][4 * q + 1], x[nc][4 * q + 2], x[nc][4 * q + 3]};
            F4(xrow + 4 * (32 * nc + 8 * q)) = v;
        }
}
template <int NC>
__device__ __forceinline__ void store_slab(const f32x16 (&y)[NC], int slab, int item, int lane) {
    const int base = slab + item * (NC * 4096) + lane * 16;
#pragma unroll
    for (int nc = 0; nc < NC; ++nc)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {y[nc][4 * q], y[nc][4 * q + 1], y[nc][4 * q + 2], y[nc][4 * q + 3]};
            F4(base + (nc * 4 + q) * 1024) = v;
        }
}
// nn.RMSNorm(eps=None): eps = finfo(float32).eps (layers.py:25-26); returns 1 / rms over the c channels of the lane's token
template <int NC>
__device__ __forceinline__ float rms_inv(const f32x16 (&x)[NC], int c) {
    float ss = 0.f;
#pragma unroll
    for (int nc = 0; nc < NC; ++nc)
#pragma unroll
        for (int r = 0; r < 16; ++r) ss = fmaf(x[nc][r], x[nc][r], ss);
    ss += __shfl_xor(ss, 32);
    return rsqrtf(ss / (float)c + 1.1920928955078125e-07f);
}
template <int NC>
__device__ __forceinline__ void split_tiles(const f32x16 (&x)[NC], u32x4 (&xh)[NC][2], u32x4 (&xl)[NC][2]) {
#pragma unroll
    for (int nc = 0; nc < NC; ++nc) {
        acc_frag<0>(x[nc], xh[nc][0], xl[nc][0]);
        acc_frag<1>(x[nc], xh[nc][1], xl[nc][1]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// attention phase of a block: BallMSA up to the proj product's split-K partials (attention.py:58-88)
template <int NC>
__device__ __forceinline__ void attn_phase(const Op& op, const Dir& D, const char* __restrict__ FR, int d, const Pending& pd) {
    constexpr int KS = 2 * NC;
    const int lane = threadIdx.x & 63, p = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int heads = op.heads, l = op.lvl, c = op.c;
    const int nitems = (op.n >> 5) * heads;
    const bool active = wave < nitems;
    const int item = active ? wave : nitems - 1;  // idle waves shadow the last item (no stores): one code path
    const int T = item / heads, h = item - T * heads;
    // weight fragments of this item: requested first, consumed after the token state is ready
    u32x4 wqk[KS][2], wv[KS][2], wp[NC][2];
    {
        const char* bq = FR + op.qk + (size_t)(h >> 1) * (KS * 2048) + (((p + 16 * (h & 1)) & 31) + 32 * hf) * 16;
        const char* bv = FR + op.v + (size_t)(h >> 2) * (KS * 2048) + (((p + 8 * (h & 3)) & 31) + 32 * hf) * 16;
        const char* bp = FR + op.pj + (size_t)h * (NC * 2048) + lane * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                wqk[s][pl] = ldg16(bq + (s * 2 + pl) * 1024);
                wv[s][pl] = ldg16(bv + (s * 2 + pl) * 1024);
            }
#pragma unroll
        for (int nc = 0; nc < NC; ++nc)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wp[nc][pl] = ldg16(bp + (nc * 2 + pl) * 1024);
    }
    const int perm = op.rot ? D.rotm[l] : D.ident;
    const int tau = I(perm + 4 * (32 * T + p));
    f32x16 x[NC];
    load_x<NC>(x, D.X[l], D.ld[l], tau, hf, pd, D.slab);
    float rl[3] = {0.f, 0.f, 0.f};
    {
        const int rel = D.rel[l][op.rot ? 1 : 0] + 4 * tau * d;
        rl[0] = F(rel);
        rl[1] = F(rel + 4);
        if (d > 2) rl[2] = F(rel + 8);
    }
    lds_barrier();  // every wave has read X and the pending slabs: both may be overwritten from here on
    if (active && h == 0 && pd.np) store_x<NC>(x, D.X[l], D.ld[l], tau, hf);
    // x~ = RMSNorm(x) * w + pe_b + rel . pe_w                  (attention.py:59-61)
    const int par = D.par[op.pbuf] - 4 * op.small_off;  // par + 4 * <v1 blob offset> addresses the LDS copy
    {
        const float rinv = rms_inv<NC>(x, c);
#pragma unroll
        for (int nc = 0; nc < NC; ++nc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = 32 * nc + 8 * q + 4 * hf;
                const f32x4 w = F4(par + 4 * (op.norm1 + ch));
                f32x4 pe = F4(par + 4 * (op.pe_b + ch));
                pe += rl[0] * F4(par + 4 * (op.pe_w + ch));
                pe += rl[1] * F4(par + 4 * (op.pe_w + c + ch));
                if (d > 2) pe += rl[2] * F4(par + 4 * (op.pe_w + 2 * c + ch));
#pragma unroll
                for (int i = 0; i < 4; ++i) x[nc][4 * q + i] = x[nc][4 * q + i] * rinv * w[i] + pe[i];
            }
    }
    u32x4 xh[NC][2], xl[NC][2];
    split_tiles<NC>(x, xh, xl);
    // rows 0-7 q of my head, rows 8-15 k (the lane's tile rows are rotated so that my head comes first)
    f32x16 qk = zero16(), vt = zero16();
#pragma unroll
    for (int nc = 0; nc < NC; ++nc)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qk = mfma3(wqk[2 * nc + s][0], wqk[2 * nc + s][1], xh[nc][s], xl[nc][s], qk);
            vt = mfma3(xh[nc][s], xl[nc][s], wv[2 * nc + s][0], wv[2 * nc + s][1], vt);  // V[token][feature]: feature on the lane
        }
    f32x16 S;
    {
        const float qscale = 1.4426950408889634f * 0.35355339059327373f;  // log2(e) / sqrt(8): softmax in base 2
        const f32x4 bq = F4(par + 4 * (op.qkv_b + 8 * h + 4 * hf)), bk = F4(par + 4 * (op.qkv_b + c + 8 * h + 4 * hf));
        u32x4 qh, ql, kh, kl;
        quad_frag((qk[0] + bq[0]) * qscale, (qk[1] + bq[1]) * qscale, (qk[2] + bq[2]) * qscale, (qk[3] + bq[3]) * qscale, qh, ql);
        quad_frag(qk[4] + bk[0], qk[5] + bk[1], qk[6] + bk[2], qk[7] + bk[3], kh, kl);
        S = mfma3(kh, kl, qh, ql, zero16());  // S[key][query]: lane = query, registers = keys
    }
    {
        const float bv = F(par + 4 * (op.qkv_b + 2 * c + 8 * h + (p & 7)));
#pragma unroll
        for (int r = 0; r < 16; ++r) vt[r] += bv;  // (lanes >= 8 hold other heads' features: never used)
    }
    const int bmask = 31 & ~(op.ball - 1);
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * hf;
        if ((key ^ p) & bmask) S[r] = -INFINITY;  // a key of another ball of this tile
        m = fmaxf(m, S[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        S[r] = __builtin_amdgcn_exp2f(S[r] - m);
        lsum += S[r];
    }
    lsum += __shfl_xor(lsum, 32);
    f32x16 O = zero16();
    {
        u32x4 vh, vl, ph, pl;
        acc_frag<0>(vt, vh, vl);
        acc_frag<0>(S, ph, pl);
        O = mfma3(vh, vl, ph, pl, O);
        acc_frag<1>(vt, vh, vl);
        acc_frag<1>(S, ph, pl);
        O = mfma3(vh, vl, ph, pl, O);  // O^T[feature][query]: rows 0-7 = my head, registers 0-3 = features 4hf ..
    }
    f32x16 y[NC];
    {
        const float inv = 1.0f / lsum;
        u32x4 oh, ol;
        quad_frag(O[0] * inv, O[1] * inv, O[2] * inv, O[3] * inv, oh, ol);
#pragma unroll
        for (int nc = 0; nc < NC; ++nc) y[nc] = mfma3(wp[nc][0], wp[nc][1], oh, ol, zero16());
    }
    if (active) store_slab<NC>(y, D.slab, item, lane);
    lds_barrier();
}

// MLP phase of a block: x += proj(...) (completing the attention), then SwiGLU up to the w3 product's split-K partials
template <int NC>
__device__ __forceinline__ void mlp_phase(const Op& op, const Dir& D, const char* __restrict__ FR, const float* __restrict__ W) {
    constexpr int KS = 2 * NC;
    const int lane = threadIdx.x & 63, p = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunks = op.chunks, l = op.lvl, c = op.c;
    const int nitems = (op.n >> 5) * chunks;
    const bool active = wave < nitems;
    const int item = active ? wave : nitems - 1;
    const int T = item / chunks, j = item - T * chunks;
    u32x4 w1[KS][2], w2[KS][2], w3[NC][2][2];
    {
        const char* b1 = FR + op.w1 + (size_t)j * (KS * 2048) + lane * 16;
        const char* b2 = FR + op.w2 + (size_t)j * (KS * 2048) + lane * 16;
        const char* b3 = FR + op.w3 + (size_t)j * (NC * 4096) + lane * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                w1[s][pl] = ldg16(b1 + (s * 2 + pl) * 1024);
                w2[s][pl] = ldg16(b2 + (s * 2 + pl) * 1024);
            }
#pragma unroll
        for (int nc = 0; nc < NC; ++nc)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) w3[nc][s][pl] = ldg16(b3 + ((nc * 2 + s) * 2 + pl) * 1024);
    }
    const int perm = op.rot ? D.rotm[l] : D.ident;
    const int tau = I(perm + 4 * (32 * T + p));
    const int par = D.par[op.pbuf] - 4 * op.small_off;
    f32x16 x[NC];
    {
        const Pending pd = {op.heads, par + 4 * op.proj_b, D.ident};  // the heads' partials, in this block's own place order
        // (place -> place: the identity map; the token's place is 32T + p)
        const int xrow = D.X[l] + 4 * (tau * D.ld[l] + 4 * hf);
        const int sl = D.slab + T * pd.np * (NC * 4096) + lane * 16;
#pragma unroll
        for (int nc = 0; nc < NC; ++nc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = F4(xrow + 4 * (32 * nc + 8 * q)) + F4(pd.bias + 4 * (32 * nc + 8 * q + 4 * hf));
#pragma unroll
                for (int i = 0; i < 4; ++i) x[nc][4 * q + i] = v[i];
            }
        for (int k = 0; k < pd.np; ++k) {
#pragma unroll
            for (int nc = 0; nc < NC; ++nc)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = F4(sl + k * (NC * 4096) + (nc * 4 + q) * 1024);
#pragma unroll
                    for (int i = 0; i < 4; ++i) x[nc][4 * q + i] += v[i];
                }
        }
    }
    lds_barrier();
    if (active && j == 0) store_x<NC>(x, D.X[l], D.ld[l], tau, hf);
    if (op.next_len) {  // the next block's small parameters into the other buffer (its attention phase reads them)
        const int dst = D.par[op.pbuf ^ 1];
        for (int i = threadIdx.x * 4; i < op.next_len; i += kThreads * 4) F4(dst + 4 * i) = *(gf4)(W + op.next_off + i);
    }
    {
        const float rinv = rms_inv<NC>(x, c);
#pragma unroll
        for (int nc = 0; nc < NC; ++nc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w = F4(par + 4 * (op.norm2 + 32 * nc + 8 * q + 4 * hf));
#pragma unroll
                for (int i = 0; i < 4; ++i) x[nc][4 * q + i] = x[nc][4 * q + i] * rinv * w[i];
            }
    }
    u32x4 xh[NC][2], xl[NC][2];
    split_tiles<NC>(x, xh, xl);
    f32x16 h1 = zero16(), h2 = zero16();
#pragma unroll
    for (int nc = 0; nc < NC; ++nc)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            h1 = mfma3(w1[2 * nc + s][0], w1[2 * nc + s][1], xh[nc][s], xl[nc][s], h1);
            h2 = mfma3(w2[2 * nc + s][0], w2[2 * nc + s][1], xh[nc][s], xl[nc][s], h2);
        }
    // g = w2(x^) * silu(w1(x^))                                (mlp.py:17-18)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b1 = F4(par + 4 * (op.w1_b + 32 * j + 8 * q + 4 * hf)), b2 = F4(par + 4 * (op.w2_b + 32 * j + 8 * q + 4 * hf));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = h1[4 * q + i] + b1[i];
            const float sig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
            h1[4 * q + i] = (h2[4 * q + i] + b2[i]) * (v * sig);
        }
    }
    f32x16 y[NC];
    {
        u32x4 gh0, gl0, gh1, gl1;
        acc_frag<0>(h1, gh0, gl0);
        acc_frag<1>(h1, gh1, gl1);
#pragma unroll
        for (int nc = 0; nc < NC; ++nc) {
            y[nc] = mfma3(w3[nc][0][0], w3[nc][0][1], gh0, gl0, zero16());
            y[nc] = mfma3(w3[nc][1][0], w3[nc][1][1], gh1, gl1, y[nc]);
        }
    }
    if (active) store_slab<NC>(y, D.slab, item, lane);
    lds_barrier();
}

// X += pending, for every token of the level (before a pool / unpool / the output reads X as plain rows)
__device__ __forceinline__ void flush_level(const Dir& D, int l, int c, const Pending& pd) {
    if (pd.np) {
        const int n = D.n[l], ld = D.ld[l], X = D.X[l], c4 = c >> 2, nc_l = c >> 5;
        const int slabB = nc_l * 4096;
        for (int idx = threadIdx.x; idx < n * c4; idx += kThreads) {
            const int tau = idx / c4, g = idx - tau * c4;
            const int pp = I(pd.inv + 4 * tau);
            const int sl = D.slab + (pp >> 5) * pd.np * slabB + (((g >> 3) * 4 + ((g & 7) >> 1)) * 64 + (pp & 31) + 32 * (g & 1)) * 16;
            f32x4 v = F4(X + 4 * (tau * ld + 4 * g)) + F4(pd.bias + 16 * g);
            for (int k = 0; k < pd.np; ++k) v += F4(sl + k * slabB);
            F4(X + 4 * (tau * ld + 4 * g)) = v;
        }
    }
    lds_barrier();
}

// BallPooling (pooling.py:25-40, eval BatchNorm folded) level l -> l+1 and BallUnpooling (pooling.py:58-72) l+1 -> l:
// item = (32 coarse tokens, 32 outputs); rows are built in registers in natural k order, the weights are the v1 blob's
// [out][K~] bf16 planes.
template <bool UNPOOL>
__device__ __forceinline__ void resample(const Op& op, const Dir& D, const float* __restrict__ W, int d) {
    const int lane = threadIdx.x & 63, p = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = op.lvl, st = op.stride, c = op.c, c2 = op.c2;
    const int n2 = D.n[l + 1];
    const int K = UNPOOL ? c2 + st * d : st * c + st * d, Kp = round32(K), Oc = UNPOOL ? st * c : c2;
    const int X0 = D.X[l], ld0 = D.ld[l], X1 = D.X[l + 1], ld1 = D.ld[l + 1], P0 = D.P[l], P1 = D.P[l + 1];
    const int ksteps = (K + 15) >> 4, otiles = Oc >> 5, nitems = (n2 >> 5) * otiles;
    const unsigned short* Wh = reinterpret_cast<const unsigned short*>(W + op.w);
    const size_t plane = (size_t)Oc * Kp;
    for (int item = wave; item < nitems; item += kThreads / 64) {
        const int Tt = item / otiles, ot = item - Tt * otiles;
        const int pt = 32 * Tt + p;  // coarse token of this lane
        const unsigned short* wrow = Wh + (size_t)(32 * ot + p) * Kp + 8 * hf;
        f32x16 acc = zero16();
        for (int ks = 0; ks < ksteps; ++ks) {
            const u32x4 ah = *(gv4)(wrow + 16 * ks), al = *(gv4)(wrow + plane + 16 * ks);
            const int k0 = 16 * ks + 8 * hf;
            float v[8];
            const int feat = UNPOOL ? c2 : st * c;  // the first `feat` k are token features, the rest relative positions
            if (k0 + 8 <= feat) {
                int src;
                if (UNPOOL) {
                    src = X1 + 4 * (pt * ld1 + k0);
                } else {
                    const int child = k0 / c, ch = k0 - child * c;
                    src = X0 + 4 * ((pt * st + child) * ld0 + ch);
                }
                const f32x4 a = F4(src), b = F4(src + 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = a[i];
                    v[4 + i] = b[i];
                }
            } else {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int r = k0 + jj - feat;
                    float t = 0.f;
                    if (r >= 0 && r < st * d) {
                        const int child = r / d, dim = r - child * d;
                        t = F(P0 + 4 * ((pt * st + child) * d + dim)) - F(P1 + 4 * (pt * d + dim));
                    }
                    v[jj] = t;
                }
            }
            u32x4 bh, bl;
            split8(v, bh, bl);
            acc = mfma3(ah, al, bh, bl, acc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o4 = 32 * ot + 8 * q + 4 * hf;
            const f32x4 b = *(gf4)(W + op.b + o4);
            if (UNPOOL) {
                const int child = o4 / c, ch = o4 - child * c;
                const f32x4 sc = *(gf4)(W + op.bn_scale + ch), sh = *(gf4)(W + op.bn_shift + ch);
                const int dst = X0 + 4 * ((pt * st + child) * ld0 + ch);
                const f32x4 old = F4(dst);
                f32x4 yv;
#pragma unroll
                for (int i = 0; i < 4; ++i) yv[i] = (old[i] + acc[4 * q + i] + b[i]) * sc[i] + sh[i];
                F4(dst) = yv;
            } else {
                const f32x4 sc = *(gf4)(W + op.bn_scale + o4), sh = *(gf4)(W + op.bn_shift + o4);
                f32x4 yv;
#pragma unroll
                for (int i = 0; i < 4; ++i) yv[i] = (acc[4 * q + i] + b[i]) * sc[i] + sh[i];
                F4(X1 + 4 * (pt * ld1 + o4)) = yv;
            }
        }
    }
    __syncthreads();
}

// rel[tok] = P[tok] - mean over its ball (attention.py:35-40); ball b = places [b*ball, ..) of `perm`; power-of-two balls
__device__ __forceinline__ void ball_rel(int P, int n, int d, int ball, int perm, int rel) {
    const int total = (n * d + 63) & ~63;  // whole waves: every lane takes part in the shuffles
    for (int i = threadIdx.x; i < total; i += kThreads) {
        const int j = i / n, pl = i - j * n;
        const bool ok = j < d;
        const int tok = ok ? I(perm + 4 * pl) : 0;
        const float v = ok ? F(P + 4 * (tok * d + j)) : 0.f;
        float sum = v;
        for (int off = 1; off < ball; off <<= 1) sum += __shfl_xor(sum, off);
        if (ok) F(rel + 4 * (tok * d + j)) = v - sum / (float)ball;
    }
}

constexpr int kWarm = 8;  // 16-B loads per thread per builder step of the L2 warm-up

__global__ __launch_bounds__(kThreads) void erwin2_kernel(const float* __restrict__ tokens, int32_t* __restrict__ tree_idx,
                                                         uint8_t* __restrict__ tree_mask, int32_t* __restrict__ rot,
                                                         const float* __restrict__ W, const char* __restrict__ FR,
                                                         float* __restrict__ out, WTreeCfg tcfg, float* __restrict__ pos_out) {
    const Plan& PL = *reinterpret_cast<const Plan*>(FR);
    const Dir& D = PL.dir;
    const int cloud = blockIdx.x, n_clouds = gridDim.x, d = PL.d, nl = PL.n_levels, M = PL.M, c0 = PL.c0;
    {
        const WTreeLds sink = {D.tmap, D.P[0], {D.rotm[0], D.rotm[1], D.rotm[2], D.rotm[3]}};
        // L2 warm-up under the builder's latency chain (the weight fragments have been evicted by the streaming kernels
        // since the last forward): step i folds what step i-1 requested into a checksum nothing reads, then requests
        // its own pieces -- unconditional, clamped addresses
        u32x4 pend[kWarm];
        unsigned sink_x = 0;
        const unsigned pieces = PL.total_bytes >> 4;
#pragma unroll
        for (int k = 0; k < kWarm; ++k) pend[k] = u32x4{0u, 0u, 0u, 0u};
        auto step = [&](int i) {
#pragma unroll
            for (int k = 0; k < kWarm; ++k) sink_x ^= pend[k][0];
            const unsigned base = (unsigned)i * (kThreads * kWarm) + threadIdx.x;
#pragma unroll
            for (int k = 0; k < kWarm; ++k) pend[k] = ((gv4)FR)[min(base + (unsigned)k * kThreads, pieces - 1)];
        };
        w_tree_build(tcfg, nullptr, tokens, c0, cloud, n_clouds, tree_idx, tree_mask, rot, pos_out, sink, step);
#pragma unroll
        for (int k = 0; k < kWarm; ++k) sink_x ^= pend[k][0];
        asm volatile("" ::"v"(sink_x));
    }
    __syncthreads();
    // the cloud in tree order (erwin_flash.py:176-181), index maps, pooled positions, first block's parameters
    {
        const int X0 = D.X[0], ld0 = D.ld[0];
        for (int i = threadIdx.x; i < M * (c0 >> 2); i += kThreads) {
            const int slot = i / (c0 >> 2), g = i - slot * (c0 >> 2);
            F4(X0 + 4 * (slot * ld0 + 4 * g)) = *(gf4)(tokens + ((size_t)cloud * M + I(D.tmap + 4 * slot)) * c0 + 4 * g);
        }
        for (int i = threadIdx.x; i < M; i += kThreads) I(D.ident + 4 * i) = i;
        if (PL.rotate)
            for (int l = 0; l < nl; ++l)
                for (int i = threadIdx.x; i < D.n[l]; i += kThreads) I(D.invr[l] + 4 * I(D.rotm[l] + 4 * i)) = i;
        const Op& first = PL.ops[0];
        if (first.kind == OP_BLOCK)
            for (int i = threadIdx.x * 4; i < first.small_len; i += kThreads * 4)
                F4(D.par[first.pbuf] + 4 * i) = *(gf4)(W + first.small_off + i);
    }
    for (int l = 0; l + 1 < nl; ++l) {  // pooled positions (pooling.py:29-33): mean of the children, sequential sum
        const int st = PL.stride[l], n2 = D.n[l + 1];
        for (int i = threadIdx.x; i < n2 * d; i += kThreads) {
            const int pt = i / d, j = i - pt * d;
            float sum = 0.f;
            for (int q = 0; q < st; ++q) sum += F(D.P[l] + 4 * ((pt * st + q) * d + j));
            F(D.P[l + 1] + 4 * i) = sum / (float)st;
        }
        __syncthreads();
    }
    __syncthreads();
    for (int l = 0; l < nl; ++l) {  // relative positions of both partitions of every level
        ball_rel(D.P[l], D.n[l], d, PL.ball[l], D.ident, D.rel[l][0]);
        if (PL.rotate) ball_rel(D.P[l], D.n[l], d, PL.ball[l], D.rotm[l], D.rel[l][1]);
    }
    __syncthreads();

    Pending pd = {0, 0, 0};
    for (int oi = 0; oi < PL.n_ops; ++oi) {
        const Op& op = PL.ops[oi];
        if (op.kind == OP_BLOCK) {
            if (op.nc == 1) {
                attn_phase<1>(op, D, FR, d, pd);
                mlp_phase<1>(op, D, FR, W);
            } else {
                attn_phase<2>(op, D, FR, d, pd);
                mlp_phase<2>(op, D, FR, W);
            }
            pd.np = op.chunks;
            pd.bias = D.par[op.pbuf] + 4 * (op.w3_b - op.small_off);
            pd.inv = op.rot ? D.invr[op.lvl] : D.ident;
        } else if (op.kind == OP_FLUSH) {
            flush_level(D, op.lvl, op.c, pd);
            pd.np = 0;
        } else if (op.kind == OP_POOL) {
            resample<false>(op, D, W, d);
        } else {
            resample<true>(op, D, W, d);
        }
    }
    // node.x[tree_mask][argsort(tree_idx[tree_mask])]  (erwin_flash.py:193); every slot of a power-of-two cloud holds a token
    {
        const int X0 = D.X[0], ld0 = D.ld[0];
        for (int i = threadIdx.x; i < M * (c0 >> 2); i += kThreads) {
            const int slot = i / (c0 >> 2), g = i - slot * (c0 >> 2);
            *reinterpret_cast<f32x4*>(out + ((size_t)cloud * M + I(D.tmap + 4 * slot)) * c0 + 4 * g) = F4(X0 + 4 * (slot * ld0 + 4 * g));
        }
    }
}

// --------------------------------------------------------------------------------------------------------------------- packer
// One 64-thread workgroup per 1-KB fragment piece (32 rows x one 16-wide k-step x one bf16 plane): lane (r, hf) gathers its
// eight k values of row r from the v1 blob's [out][K~] planes in the order the kernel's operands have them.
__global__ __launch_bounds__(64) void e2_pack_kernel(const float* __restrict__ W, char* __restrict__ FR, Op op) {
    const int lane = threadIdx.x, r = lane & 31, hf = lane >> 5;
    const int NC = op.nc, KS = 2 * NC, C = op.c, heads = op.heads, chunks = op.chunks;
    const int nQK = ((heads + 1) >> 1) * KS * 2, nV = ((heads + 3) >> 2) * KS * 2, nPJ = heads * NC * 2;
    const int nW12 = chunks * KS * 2;
    int piece = blockIdx.x;
    const unsigned dst = op.qk + (unsigned)piece * 1024u + lane * 16;
    int src, Orows, Kp, row, kbase, kmode = 0, pl;
    bool valid = true;
    const int grp = r >> 3, w8 = r & 7;
    if (piece < nQK) {
        const int t = piece / (KS * 2), rem = piece - t * (KS * 2);
        pl = rem & 1;
        kbase = 16 * (rem >> 1);
        const int head = 2 * t + (grp >> 1);
        src = op.qkv_w, Orows = 3 * C, Kp = round32(C);
        row = ((grp & 1) ? C : 0) + 8 * head + w8;
        valid = head < heads;
    } else if ((piece -= nQK) < nV) {
        const int t = piece / (KS * 2), rem = piece - t * (KS * 2);
        pl = rem & 1;
        kbase = 16 * (rem >> 1);
        const int head = 4 * t + grp;
        src = op.qkv_w, Orows = 3 * C, Kp = round32(C);
        row = 2 * C + 8 * head + w8;
        valid = head < heads;
    } else if ((piece -= nV) < nPJ) {
        const int hh = piece / (NC * 2), rem = piece - hh * (NC * 2);
        pl = rem & 1;
        src = op.proj_w, Orows = C, Kp = round32(C);
        row = 32 * (rem >> 1) + r;
        kbase = 8 * hh;
        kmode = 1;
    } else if ((piece -= nPJ) < 2 * nW12) {
        const bool second = piece >= nW12;
        if (second) piece -= nW12;
        const int jj = piece / (KS * 2), rem = piece - jj * (KS * 2);
        pl = rem & 1;
        kbase = 16 * (rem >> 1);
        src = second ? op.w2_w : op.w1_w, Orows = op.hid, Kp = round32(C);
        row = 32 * jj + r;
    } else {
        piece -= 2 * nW12;
        const int jj = piece / (NC * 4), rem = piece - jj * (NC * 4);
        pl = rem & 1;
        src = op.w3_w, Orows = C, Kp = round32(op.hid);
        row = 32 * (rem >> 2) + r;
        kbase = 32 * jj + 16 * ((rem >> 1) & 1);
    }
    const unsigned short* plane = reinterpret_cast<const unsigned short*>(W + src) + (size_t)pl * Orows * Kp;
    unsigned short e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int k;
        if (kmode == 0) k = kbase + 8 * (j >> 2) + 4 * hf + (j & 3);
        else k = j < 4 ? kbase + 4 * hf + j : -1;
        e[j] = (valid && k >= 0) ? plane[(size_t)row * Kp + k] : (unsigned short)0;
    }
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (unsigned)e[2 * i] | ((unsigned)e[2 * i + 1] << 16);
    *reinterpret_cast<u32x4*>(FR + dst) = o;
}

__global__ void e2_plan_kernel(char* __restrict__ FR, Plan plan) {
    const int* src = reinterpret_cast<const int*>(&plan);
    int* dst = reinterpret_cast<int*>(FR);
    for (unsigned i = threadIdx.x; i < kPlanBytes / 4; i += blockDim.x) dst[i] = i < sizeof(Plan) / 4 ? src[i] : 0;
}

// --------------------------------------------------------------------------------------------------------------------- host
static void fill_block(Op& o, int lvl, int n, const haet_erwin_layer& L, const haet_erwin_block_w& w, bool rot) {
    o = Op{};
    o.kind = OP_BLOCK, o.lvl = lvl, o.n = n, o.c = L.c, o.nc = L.c / 32;
    o.heads = L.heads, o.ball = L.ball, o.hid = L.hidden, o.chunks = L.hidden / 32, o.rot = rot ? 1 : 0;
    o.small_off = w.small_off, o.small_len = w.small_len;
    o.norm1 = w.norm1, o.norm2 = w.norm2, o.pe_w = w.pe_w, o.pe_b = w.pe_b, o.qkv_b = w.qkv_b, o.proj_b = w.proj_b;
    o.w1_b = w.w1_b, o.w2_b = w.w2_b, o.w3_b = w.w3_b;
    o.qkv_w = w.qkv_w, o.proj_w = w.proj_w, o.w1_w = w.w1_w, o.w2_w = w.w2_w, o.w3_w = w.w3_w;
}

static unsigned block_pieces(const Op& o) {
    const int KS = 2 * o.nc;
    return ((o.heads + 1) / 2) * KS * 2 + ((o.heads + 3) / 4) * KS * 2 + o.heads * o.nc * 2 + 2 * o.chunks * KS * 2 +
           o.chunks * o.nc * 4;
}

// false: the shape is outside the fragment kernel's envelope
static bool make_plan(const haet_erwin_desc& D, int M, Plan* out) {
    Plan P = {};
    const int nl = D.n_levels;
    if (nl < 1 || nl > HAET_ERWIN_MAX_LEVELS || !D.decode || D.bn_train || (D.d != 2 && D.d != 3)) return false;
    if (M < 32 || M > 64 || (M & (M - 1))) return false;
    P.n_levels = nl, P.d = D.d, P.M = M, P.c0 = D.enc[0].c, P.rotate = D.rotate ? 1 : 0;
    int n = M, nlv[HAET_ERWIN_MAX_LEVELS];
    for (int l = 0; l < nl; ++l) {
        nlv[l] = n;
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && l == nl - 1) break;
            const haet_erwin_layer& L = pass ? D.dec[l] : D.enc[l];
            if (L.depth < 0 || L.depth > HAET_ERWIN_MAX_DEPTH) return false;
            if (L.c != 32 && L.c != 64) return false;
            if (pass && L.c != D.enc[l].c) return false;
            if (L.heads < 1 || L.c != 8 * L.heads) return false;  // head dim 8
            if (L.ball < 2 || L.ball > 32 || (L.ball & (L.ball - 1)) || n % L.ball) return false;
            if (pass && L.ball != D.enc[l].ball) return false;
            if (L.hidden < 32 || L.hidden % 32) return false;
            if (n % 32 || (n / 32) * L.heads > 8 || (n / 32) * (L.hidden / 32) > 8) return false;
        }
        P.ball[l] = D.enc[l].ball;
        if (l < nl - 1) {
            if (D.pool[l].stride != 2 || D.unpool[l].stride != 2) return false;
            P.stride[l] = 2;
            n /= 2;
        }
    }
    // op list (erwin_flash.py:183-192)
    int no = 0;
    auto push = [&](const Op& o) {
        if (no < kMaxOps) P.ops[no] = o;
        ++no;
    };
    auto blocks = [&](int l, const haet_erwin_layer& L) {
        for (int j = 0; j < L.depth; ++j) {
            Op o;
            fill_block(o, l, nlv[l], L, L.blocks[j], D.rotate && L.depth > 1 && (j & 1));  // layers.py:66,75-89
            push(o);
        }
    };
    auto simple = [&](int kind, int l, const haet_erwin_resample* R) {
        Op o = {};
        o.kind = kind, o.lvl = l, o.n = nlv[l], o.c = D.enc[l].c, o.nc = o.c / 32;
        if (R) {
            o.w = R->w, o.b = R->b, o.bn_scale = R->bn_scale, o.bn_shift = R->bn_shift, o.stride = R->stride;
            o.c2 = D.enc[l + 1].c;
        }
        push(o);
    };
    for (int l = 0; l < nl; ++l) {
        blocks(l, D.enc[l]);
        if (l < nl - 1) {
            simple(OP_FLUSH, l, nullptr);
            simple(OP_POOL, l, &D.pool[l]);
        }
    }
    for (int l = nl - 2; l >= 0; --l) {
        simple(OP_FLUSH, l + 1, nullptr);
        simple(OP_UNPOOL, l, &D.unpool[l]);
        blocks(l, D.dec[l]);
    }
    simple(OP_FLUSH, 0, nullptr);
    if (no > kMaxOps) return false;
    P.n_ops = no;
    // parameter buffers alternate over the BLOCK ops; every block stages its successor's group
    int max_small = 4, nb = 0, prev = -1;
    for (int i = 0; i < no; ++i) {
        Op& o = P.ops[i];
        if (o.kind != OP_BLOCK) continue;
        o.pbuf = nb & 1;
        ++nb;
        max_small = max_small > o.small_len ? max_small : o.small_len;
        if (prev >= 0) P.ops[prev].next_off = o.small_off, P.ops[prev].next_len = o.small_len;
        prev = i;
    }
    if (nb == 0) return false;
    // fragment sections
    unsigned off = kPlanBytes;
    P.frag_begin = off;
    for (int i = 0; i < no; ++i) {
        Op& o = P.ops[i];
        if (o.kind != OP_BLOCK) continue;
        const int KS = 2 * o.nc;
        o.qk = off;
        o.v = o.qk + ((o.heads + 1) / 2) * KS * 2 * 1024u;
        o.pj = o.v + ((o.heads + 3) / 4) * KS * 2 * 1024u;
        o.w1 = o.pj + o.heads * o.nc * 2 * 1024u;
        o.w2 = o.w1 + o.chunks * KS * 2 * 1024u;
        o.w3 = o.w2 + o.chunks * KS * 2 * 1024u;
        off = o.qk + block_pieces(o) * 1024u;
    }
    P.total_bytes = off;
    // LDS directory
    Dir& T = P.dir;
    int p = 0, maxslab = 0;
    auto take = [&](int bytes) {
        const int at = p;
        p += (bytes + 15) & ~15;
        return at;
    };
    for (int l = 0; l < nl; ++l) {
        T.n[l] = nlv[l];
        T.ld[l] = D.enc[l].c + kPad;
        T.X[l] = take(4 * nlv[l] * T.ld[l]);
        const int s = 8 * (D.enc[l].c / 32) * 4096;
        maxslab = maxslab > s ? maxslab : s;
    }
    for (int l = 0; l < nl; ++l) T.P[l] = take(4 * nlv[l] * D.d);
    for (int l = 0; l < nl; ++l) {
        T.rel[l][0] = take(4 * nlv[l] * D.d);
        T.rel[l][1] = take(4 * nlv[l] * D.d);
        T.rotm[l] = take(4 * nlv[l]);
        T.invr[l] = take(4 * nlv[l]);
    }
    for (int l = nl; l < HAET_ERWIN_MAX_LEVELS; ++l) T.rotm[l] = -1;
    T.ident = take(4 * M);
    T.tmap = take(4 * M);
    T.par[0] = take(4 * max_small);
    T.par[1] = take(4 * max_small);
    T.slab = take(maxslab);
    T.total = p;
    if (p + 8 * 1024 > 160 * 1024) return false;  // (+ the tree builder's static tables)
    *out = P;
    return true;
}

}  // namespace e2

bool balltree_wave_ok(int M, int n_levels, const int* st);  // balltree_wave.hip
void rotation_matrix_f32(float angle_deg, int d, float* r9);  // balltree.hip

}  // namespace haet

using namespace haet;

extern "C" size_t haet_erwin_frag_bytes(const haet_erwin_desc* desc, int M) {
    static const bool off = getenv("HAET_ERWIN_V1") != nullptr;  // A/B switch: every shape on erwin.hip
    e2::Plan P;
    if (off || !desc || !e2::make_plan(*desc, M, &P)) return 0;
    int st[3] = {1, 1, 1};
    for (int l = 0; l < desc->n_levels - 1 && l < 3; ++l) st[l] = desc->pool[l].stride;
    if (!balltree_wave_ok(M, desc->n_levels, st)) return 0;
    return P.total_bytes;
}

extern "C" int haet_erwin_frag_pack(const float* weights, size_t weight_floats, const haet_erwin_desc* desc, int M,
                                    void* frags, size_t frag_bytes, haet_stream_t stream) {
    if (!weights || !desc || !frags) return fail(HAET_E_ARG, "erwin_frag_pack: null pointer");
    if (weight_floats * 4 < sizeof(haet_erwin_desc)) return fail(HAET_E_ARG, "erwin_frag_pack: weight blob smaller than its header");
    e2::Plan P;
    if (!e2::make_plan(*desc, M, &P)) return fail(HAET_E_ARG, "erwin_frag_pack: shape outside the fragment kernel's envelope");
    if (frag_bytes < P.total_bytes) return fail(HAET_E_ARG, "erwin_frag_pack: %zu bytes given, %u needed", frag_bytes, P.total_bytes);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(e2::e2_plan_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<char*>(frags), P);
    for (int i = 0; i < P.n_ops; ++i) {
        const e2::Op& o = P.ops[i];
        if (o.kind != e2::OP_BLOCK) continue;
        hipLaunchKernelGGL(e2::e2_pack_kernel, dim3(e2::block_pieces(o)), dim3(64), 0, st, weights, reinterpret_cast<char*>(frags), o);
    }
    return check_launch("e2_pack_kernel");
}

extern "C" int haet_token_stage_fwd_frag(const float* S, int n_clouds, int M, float angle_deg, const float* weights,
                                         size_t weight_floats, const void* frags, size_t frag_bytes,
                                         const haet_erwin_desc* desc, float* pos_out, int32_t* tree_idx,
                                         uint8_t* tree_mask, int32_t* rot, float* out, haet_stream_t stream) {
    if (!S || !weights || !frags || !out || !desc || n_clouds < 1) return fail(HAET_E_ARG, "token_stage_frag: null pointer");
    if (weight_floats * 4 < sizeof(haet_erwin_desc)) return fail(HAET_E_ARG, "token_stage_frag: weight blob smaller than its header");
    e2::Plan P;
    if (!e2::make_plan(*desc, M, &P) || !haet_erwin_frag_bytes(desc, M))
        return fail(HAET_E_ARG, "token_stage_frag: shape outside the fragment kernel's envelope (haet_erwin_frag_bytes)");
    if (frag_bytes < P.total_bytes) return fail(HAET_E_ARG, "token_stage_frag: fragment blob too small");
    if ((tree_idx == nullptr) != (tree_mask == nullptr)) return fail(HAET_E_ARG, "token_stage_frag: tree_idx / tree_mask go together");
    if ((desc->rotate != 0) != (angle_deg > 0.f)) return fail(HAET_E_ARG, "token_stage_frag: desc->rotate and angle disagree");
    WTreeCfg cfg = {};
    cfg.M = M, cfg.nd = desc->d, cfg.n_levels = desc->n_levels;
    cfg.rotate = desc->rotate && angle_deg > 0.f;
    cfg.s[0] = cfg.s[1] = cfg.s[2] = 1;
    for (int l = 0; l < desc->n_levels; ++l) {
        // balltree.pyx:643: int(log2(TOTAL leaf count / ball size)) -- the total over all clouds (reference quirk)
        const double q = (double)n_clouds * M / (double)desc->enc[l].ball;
        int tg = 0;
        while ((double)(2ll << tg) <= q) ++tg;
        cfg.t[l] = q >= 1.0 ? tg : 0;
        if (l < desc->n_levels - 1) cfg.s[l] = desc->pool[l].stride;
    }
    rotation_matrix_f32(angle_deg, desc->d, cfg.R.r);
    const size_t bytes = (size_t)P.dir.total;
    static std::atomic<size_t> attr_set[64];
    std::atomic<size_t>& attr_dev = attr_set[PerDeviceOnce::bit()];
    if (bytes > 48 * 1024 && bytes > attr_dev.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(e2::erwin2_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return fail(HAET_E_LAUNCH, "token_stage_frag: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_dev.store(bytes, std::memory_order_release);
    }
    hipLaunchKernelGGL(e2::erwin2_kernel, dim3(n_clouds), dim3(e2::kThreads), bytes, as_stream(stream), S, tree_idx, tree_mask,
                       rot, weights, reinterpret_cast<const char*>(frags), out, cfg, pos_out);
    return check_launch("erwin2_kernel");
}
