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
// PV product wants it.  Nothing but the split-K partial sums and the token state X goes through LDS: a phase ends with
// its partial slabs stored, one barrier, a reduction step in which every thread adds bias + partials into ONE 16-B
// piece of X (each partial element is read once -- letting every consumer wave sum all slabs itself cost 4-8x the LDS
// traffic and was the largest part of a phase), and a second barrier: 4 per block against ~10 in erwin.hip, with none
// of its ~50 LDS round trips per block.
//
// Envelope (haet_erwin_frag_bytes != 0): eval-mode BatchNorm, head dim 8, channels 32 or 64, 32 | tokens per level (round 4: a
// 16-token level below level 0 runs as one tile padded with phantom tokens: M = 32, the velocity-mem-checker shape),
// (tokens / 32) * heads <= 8 and (tokens / 32) * hidden / 32 <= 8 (every item of a phase on its own wave), power-of-two
// balls <= 32, stride 2, power-of-two clouds <= 64 tokens (the wave tree builder).  BASELINE config C3 (M = 64: 64 x 32
// and 32 x 64 tokens x channels) is inside; everything else, training (tape, batch statistics) included, stays on
// erwin.hip.
#include <cstdlib>

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

#ifdef HAET_STAMPS
// Diagnostic build only (make EXTRA=-DHAET_STAMPS; tools/erwin2_stamps.py): s_memtime at phase boundaries of cloud 0,
// written to a buffer nothing else reads.  Never quote this build's run time.
__device__ unsigned long long haet_stamps2[128];
__device__ int haet_stamp2_op;  // block op whose inner phases are stamped (slots 32..)
#define STAMP2(i)                                                                       \
    do {                                                                                \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                      \
            haet_stamps2[i] = __builtin_amdgcn_s_memtime();                             \
            haet_stamps2[64 + (i)] = __builtin_amdgcn_s_memrealtime();                  \
        }                                                                               \
    } while (0)
#define PSTAMP(i) do { if (stamp_on) STAMP2(i); } while (0)
#else
#define STAMP2(i) do { } while (0)
#define PSTAMP(i) do { } while (0)
#endif

constexpr int kThreads = 512;
constexpr int kMaxOps = 16;
constexpr int kPad = 4;  // fp32 row padding of the token state (floats)
enum { OP_BLOCK = 0, OP_POOL = 2, OP_UNPOOL = 3 };

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

constexpr int kOpInts = sizeof(Op) / 4;
static_assert(kOpInts <= 64, "an Op is fetched from LDS with one dword per lane");

struct Dir {  // LDS byte offsets; n = tokens of the level PADDED to whole 32-token tiles, nr = the real count (round 4: the
              // 16-token coarse level of M = 32 clouds -- the reference's velocity-mem-checker shape -- runs as one tile
              // whose places 16..31 are phantom tokens: finite, a ball of their own, never read back)
    int X[HAET_ERWIN_MAX_LEVELS], ld[HAET_ERWIN_MAX_LEVELS], n[HAET_ERWIN_MAX_LEVELS], nr[HAET_ERWIN_MAX_LEVELS], P[HAET_ERWIN_MAX_LEVELS];
    int rel[HAET_ERWIN_MAX_LEVELS][2], rotm[HAET_ERWIN_MAX_LEVELS];
    int ident, tmap, par[2], slab, ops, total;
};

// The head travels to the kernel as an argument (scalar registers from the first instruction on); the op table sits
// behind it in the fragment blob and is copied to LDS once: read from global memory where they are used, the fields of
// an op were 4-5 DEPENDENT scalar-cache misses at the top of every phase (~1.3k cycles each time).
struct Head {
    int n_ops, n_levels, d, M, c0, rotate;
    int ball[HAET_ERWIN_MAX_LEVELS], stride[HAET_ERWIN_MAX_LEVELS];
    unsigned frag_begin, total_bytes;
    unsigned v1_lo, v1_bytes;  // byte range of the v1 blob that holds the pooling / unpooling Linears (L2 warm-up)
    unsigned stamp;            // FNV-1a over the whole Plan (this field zero): the kernel checks the blob's against its argument's
    Dir dir;
};
struct Plan {
    Head h;
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
        unsigned h, l;
        split_bf16_pair(v[2 * i], v[2 * i + 1], h, l);
        hi[i] = h, lo[i] = l;
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

// op `oi` of the table in LDS: one dword per lane; a field is one v_readlane where it is used (OF), so that no phase
// keeps 41 scalars alive
__device__ __forceinline__ int fetch_op(int opsL, int oi) {
    const int lane = threadIdx.x & 63;
    return I(opsL + 4 * (oi * kOpInts + min(lane, kOpInts - 1)));
}
#define OF(field) __builtin_amdgcn_readlane(opf, offsetof(Op, field) / 4)
// the LDS directory the same way: one dword per lane for the kernel's lifetime (39 scalars otherwise, most of them
// spilled), DF(field) / DFI(array, i) = one v_readlane
#define DF(field) __builtin_amdgcn_readlane(dirf, offsetof(Dir, field) / 4)
#define DFI(field, i) __builtin_amdgcn_readlane(dirf, offsetof(Dir, field) / 4 + (i))
static_assert(sizeof(Dir) / 4 <= 64, "the directory is held with one dword per lane");

// x (accumulator layout: lane = (token place p, half hf), register 4q+i of tile nc = channel 32nc + 8q + 4hf + i) of
// token `tau`
template <int NC>
__device__ __forceinline__ void load_x(f32x16 (&x)[NC], int X, int ld, int tau, int hf) {
    const int xrow = X + 4 * (tau * ld + 4 * hf);
#pragma unroll
    for (int nc = 0; nc < NC; ++nc)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = F4(xrow + 4 * (32 * nc + 8 * q));
#pragma unroll
            for (int i = 0; i < 4; ++i) x[nc][4 * q + i] = v[i];
        }
}
// End of a phase: X[token at place] += bias + the `np` partial slabs of its tile (fixed order), one 16-B piece of X per
// thread and step; a wave reads 64 consecutive lanes of one slab row (1 KB, conflict-free).  Barriers on both sides.
__device__ __forceinline__ void reduce_phase(const int tid, const int dirf, int l, int c, int np, int bias, int perm) {
    lds_barrier();
    const int n = DFI(n, l), ld = DFI(ld, l), X = DFI(X, l), ncl = c >> 5, slabB = ncl * 4096;
    for (int idx = tid; idx < n * (c >> 2); idx += kThreads) {
        const int ln = idx & 63, combo = idx >> 6, q = combo & 3, rest = combo >> 2;
        const int tile = rest / ncl, nc = rest - tile * ncl;
        const int tau = I(perm + 4 * (32 * tile + (ln & 31)));
        const int ch = 32 * nc + 8 * q + 4 * (ln >> 5);
        const int sl = DF(slab) + tile * np * slabB + (nc * 4 + q) * 1024 + ln * 16;
        f32x4 v = F4(X + 4 * (tau * ld + ch)) + F4(bias + 4 * ch);
        for (int k = 0; k < np; ++k) v += F4(sl + k * slabB);
        F4(X + 4 * (tau * ld + ch)) = v;
    }
    lds_barrier();
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
__device__ __forceinline__ void attn_phase(const int tid, const int opf, const int dirf, const char* __restrict__ FR, unsigned frag_bytes,
                                           int d, bool stamp_on) {
    constexpr int KS = 2 * NC;
    PSTAMP(32);
    const int lane = tid & 63, p = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int heads = OF(heads), l = OF(lvl), c = OF(c);
    const int nitems = (OF(n) >> 5) * heads;
    const bool active = wave < nitems;
    const int item = active ? wave : nitems - 1;  // idle waves shadow the last item (no stores): one code path
    const int T = item / heads, h = item - T * heads;
    // weight fragments of this item: requested first, consumed after the token state is ready
    u32x4 wqk[KS][2], wv[KS][2], wp[NC][2];
    const char* bp = FR + OF(pj) + (size_t)h * (NC * 2048) + lane * 16;
    {
        // The QK tile holds a head PAIR and the V tile a head QUAD (dense bytes in HBM); this item uses 16 / 8 of the 32
        // rows.  The other lanes' offsets are pushed out of the buffer's range: the hardware returns zeros for them and
        // fetches nothing -- half / three quarters of the phase's L2 -> CU bytes were rows nobody used.
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(FR), 0, (int)frag_bytes, 0x00020000);
        const unsigned oq = p < 16 ? OF(qk) + (unsigned)(h >> 1) * (KS * 2048) + (((p + 16 * (h & 1)) & 31) + 32 * hf) * 16 : 0xfffff000u;
        const unsigned ov = p < 8 ? OF(v) + (unsigned)(h >> 2) * (KS * 2048) + (((p + 8 * (h & 3)) & 31) + 32 * hf) * 16 : 0xfffff000u;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                wqk[s][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, oq, (s * 2 + pl) * 1024, 0);
                wv[s][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, ov, (s * 2 + pl) * 1024, 0);
            }
    }
    const int perm = OF(rot) ? DFI(rotm, l) : DF(ident);
    const int tau = I(perm + 4 * (32 * T + p));
    f32x16 x[NC];
    load_x<NC>(x, DFI(X, l), DFI(ld, l), tau, hf);
    float rl[3] = {0.f, 0.f, 0.f};
    {
        const int rel = DFI(rel, 2 * (l) + (OF(rot) ? 1 : 0)) + 4 * tau * d;
        rl[0] = F(rel);
        rl[1] = F(rel + 4);
        if (d > 2) rl[2] = F(rel + 8);
    }
    PSTAMP(33);
    PSTAMP(34);
    // x~ = RMSNorm(x) * w + pe_b + rel . pe_w                  (attention.py:59-61)
    const int par = DFI(par, OF(pbuf)) - 4 * OF(small_off);  // par + 4 * <v1 blob offset> addresses the LDS copy
    {
        const float rinv = rms_inv<NC>(x, c);
#pragma unroll
        for (int nc = 0; nc < NC; ++nc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = 32 * nc + 8 * q + 4 * hf;
                const f32x4 w = F4(par + 4 * (OF(norm1) + ch));
                f32x4 pe = F4(par + 4 * (OF(pe_b) + ch));
                pe += rl[0] * F4(par + 4 * (OF(pe_w) + ch));
                pe += rl[1] * F4(par + 4 * (OF(pe_w) + c + ch));
                if (d > 2) pe += rl[2] * F4(par + 4 * (OF(pe_w) + 2 * c + ch));
#pragma unroll
                for (int i = 0; i < 4; ++i) x[nc][4 * q + i] = x[nc][4 * q + i] * rinv * w[i] + pe[i];
            }
    }
    u32x4 xh[NC][2], xl[NC][2];
    split_tiles<NC>(x, xh, xl);
    PSTAMP(35);
    // (the proj fragments are requested here: the softmax and the PV product cover their latency, and they do not sit in
    // registers through the norm)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nc = 0; nc < NC; ++nc)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wp[nc][pl] = ldg16(bp + (nc * 2 + pl) * 1024);
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
        const f32x4 bq = F4(par + 4 * (OF(qkv_b) + 8 * h + 4 * hf)), bk = F4(par + 4 * (OF(qkv_b) + c + 8 * h + 4 * hf));
        u32x4 qh, ql, kh, kl;
        quad_frag((qk[0] + bq[0]) * qscale, (qk[1] + bq[1]) * qscale, (qk[2] + bq[2]) * qscale, (qk[3] + bq[3]) * qscale, qh, ql);
        quad_frag(qk[4] + bk[0], qk[5] + bk[1], qk[6] + bk[2], qk[7] + bk[3], kh, kl);
        S = mfma3(kh, kl, qh, ql, zero16());  // S[key][query]: lane = query, registers = keys
    }
    {
        const float bv = F(par + 4 * (OF(qkv_b) + 2 * c + 8 * h + (p & 7)));
#pragma unroll
        for (int r = 0; r < 16; ++r) vt[r] += bv;  // (lanes >= 8 hold other heads' features: never used)
    }
    PSTAMP(36);
    const int bmask = 31 & ~(OF(ball) - 1);
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
    PSTAMP(37);
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
    PSTAMP(38);
    if (active) store_slab<NC>(y, DF(slab), item, lane);
    reduce_phase(tid, dirf, l, c, heads, par + 4 * OF(proj_b), perm);  // x += proj(attention) + b
    PSTAMP(39);
}

// MLP phase of a block: x += proj(...) (completing the attention), then SwiGLU up to the w3 product's split-K partials
template <int NC>
__device__ __forceinline__ void mlp_phase(const int tid, const int opf, const int dirf, const char* __restrict__ FR, const float* __restrict__ W,
                                          bool stamp_on) {
    constexpr int KS = 2 * NC;
    const int lane = tid & 63, p = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunks = OF(chunks), l = OF(lvl), c = OF(c);
    const int nitems = (OF(n) >> 5) * chunks;
    const bool active = wave < nitems;
    const int item = active ? wave : nitems - 1;
    const int T = item / chunks, j = item - T * chunks;
    u32x4 w1[KS][2], w2[KS][2], w3[NC][2][2];
    {
        const char* b1 = FR + OF(w1) + (size_t)j * (KS * 2048) + lane * 16;
        const char* b2 = FR + OF(w2) + (size_t)j * (KS * 2048) + lane * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                w1[s][pl] = ldg16(b1 + (s * 2 + pl) * 1024);
                w2[s][pl] = ldg16(b2 + (s * 2 + pl) * 1024);
            }
    }
    const int perm = OF(rot) ? DFI(rotm, l) : DF(ident);
    const int tau = I(perm + 4 * (32 * T + p));
    const int par = DFI(par, OF(pbuf)) - 4 * OF(small_off);
    f32x16 x[NC];
    load_x<NC>(x, DFI(X, l), DFI(ld, l), tau, hf);
    PSTAMP(40);
    PSTAMP(41);
    if (OF(next_len)) {  // the next block's small parameters into the other buffer (its attention phase reads them)
        const int dst = DFI(par, OF(pbuf) ^ 1);
        for (int i = tid * 4; i < OF(next_len); i += kThreads * 4) F4(dst + 4 * i) = *(gf4)(W + OF(next_off) + i);
    }
    {
        const float rinv = rms_inv<NC>(x, c);
#pragma unroll
        for (int nc = 0; nc < NC; ++nc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w = F4(par + 4 * (OF(norm2) + 32 * nc + 8 * q + 4 * hf));
#pragma unroll
                for (int i = 0; i < 4; ++i) x[nc][4 * q + i] = x[nc][4 * q + i] * rinv * w[i];
            }
    }
    u32x4 xh[NC][2], xl[NC][2];
    split_tiles<NC>(x, xh, xl);
    PSTAMP(42);
    __builtin_amdgcn_sched_barrier(0);
    {   // (requested here: the w1 / w2 products and the gate cover their latency)
        const char* b3 = FR + OF(w3) + (size_t)j * (NC * 4096) + lane * 16;
#pragma unroll
        for (int nc = 0; nc < NC; ++nc)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) w3[nc][s][pl] = ldg16(b3 + ((nc * 2 + s) * 2 + pl) * 1024);
    }
    f32x16 h1 = zero16(), h2 = zero16();
#pragma unroll
    for (int nc = 0; nc < NC; ++nc)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            h1 = mfma3(w1[2 * nc + s][0], w1[2 * nc + s][1], xh[nc][s], xl[nc][s], h1);
            h2 = mfma3(w2[2 * nc + s][0], w2[2 * nc + s][1], xh[nc][s], xl[nc][s], h2);
        }
    PSTAMP(43);
    // g = w2(x^) * silu(w1(x^))                                (mlp.py:17-18)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b1 = F4(par + 4 * (OF(w1_b) + 32 * j + 8 * q + 4 * hf)), b2 = F4(par + 4 * (OF(w2_b) + 32 * j + 8 * q + 4 * hf));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = h1[4 * q + i] + b1[i];
            const float sig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
            h1[4 * q + i] = (h2[4 * q + i] + b2[i]) * (v * sig);
        }
    }
    PSTAMP(44);
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
    PSTAMP(45);
    if (active) store_slab<NC>(y, DF(slab), item, lane);
    reduce_phase(tid, dirf, l, c, chunks, par + 4 * OF(w3_b), perm);  // x += w3(...) + b
    PSTAMP(46);
}

// BallPooling (pooling.py:25-40, eval BatchNorm folded) level l -> l+1 and BallUnpooling (pooling.py:58-72) l+1 -> l:
// item = (32 coarse tokens, 32 outputs); rows are built in registers in natural k order, the weights are the v1 blob's
// [out][K~] bf16 planes.
template <bool UNPOOL>
__device__ __forceinline__ void resample(const int tid, const int opf, const int dirf, const float* __restrict__ W, int d) {
    const int lane = tid & 63, p = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = OF(lvl), st = OF(stride), c = OF(c), c2 = OF(c2);
    const int n2 = DFI(n, l + 1), nr2 = DFI(nr, l + 1);  // (padded / real coarse tokens)
    const int K = UNPOOL ? c2 + st * d : st * c + st * d, Kp = round32(K), Oc = UNPOOL ? st * c : c2;
    const int X0 = DFI(X, l), ld0 = DFI(ld, l), X1 = DFI(X, l + 1), ld1 = DFI(ld, l + 1), P0 = DFI(P, l), P1 = DFI(P, l + 1);
    const int otiles = Oc >> 5, nitems = (n2 >> 5) * otiles;
    const int lc = c == 64 ? 6 : 5;
    const unsigned short* Wh = reinterpret_cast<const unsigned short*>(W + OF(w));
    const size_t plane = (size_t)Oc * Kp;
    // k-steps 0 .. kf-1 are token features (two 16-B LDS reads per lane), k-step kf holds the st * d relative positions
    // (built element by element: ONE instance of that code, not one per unrolled step -- it was 1100 vector instructions)
    constexpr int KF = 4;  // feature k-steps held in registers (<= 64 feature columns: make_plan)
    const int feat = UNPOOL ? c2 : st * c, kf = feat >> 4;
    for (int item = wave; item < nitems; item += kThreads / 64) {
        const int Tt = item / otiles, ot = item - Tt * otiles;
        const int pt_place = 32 * Tt + p;              // coarse token of this lane; a phantom one (>= nr2) mirrors the last real
        const int pt = min(pt_place, nr2 - 1);         // token: pooling gives it that token's (finite) value, unpooling stores nothing
        const unsigned short* wrow = Wh + (size_t)(32 * ot + p) * Kp + 8 * hf;
        u32x4 ah[KF], al[KF];
#pragma unroll
        for (int ks = 0; ks < KF; ++ks) {  // all requested up front (clamped: unconditional loads)
            const int kk = 16 * min(ks, kf - 1);
            ah[ks] = *(gv4)(wrow + kk);
            al[ks] = *(gv4)(wrow + plane + kk);
        }
        const u32x4 rh = *(gv4)(wrow + 16 * kf), rl = *(gv4)(wrow + plane + 16 * kf);
        // bias and folded BatchNorm of this lane's outputs: requested now, used after the product
        f32x4 eb[4], esc[4], esh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o4 = 32 * ot + 8 * q + 4 * hf, ch = UNPOOL ? (o4 & (c - 1)) : o4;
            eb[q] = *(gf4)(W + OF(b) + o4);
            esc[q] = *(gf4)(W + OF(bn_scale) + ch);
            esh[q] = *(gf4)(W + OF(bn_shift) + ch);
        }
        f32x16 acc = zero16();
        {   // relative positions of the children (pooling.py:30-35,60-63): k = feat + child * d + dim
            float v[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int r = 8 * hf + jj;
                float t = 0.f;
                if (r < st * d) {
                    const int child = r >= d ? 1 : 0, dim = r - child * d;  // (stride 2)
                    t = F(P0 + 4 * ((pt * st + child) * d + dim)) - F(P1 + 4 * (pt * d + dim));
                }
                v[jj] = t;
            }
            u32x4 bh, bl;
            split8(v, bh, bl);
            acc = mfma3(rh, rl, bh, bl, acc);
        }
#pragma unroll
        for (int ks = 0; ks < KF; ++ks) {
            if (ks < kf) {  // (wave-uniform)
                const int k0 = 16 * ks + 8 * hf;
                int src;
                if (UNPOOL) {
                    src = X1 + 4 * (pt * ld1 + k0);
                } else {
                    const int child = k0 >> lc, ch = k0 & (c - 1);  // (c is 32 or 64)
                    src = X0 + 4 * ((pt * st + child) * ld0 + ch);
                }
                const f32x4 a = F4(src), b = F4(src + 16);
                const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                u32x4 bh, bl;
                split8(v, bh, bl);
                acc = mfma3(ah[ks], al[ks], bh, bl, acc);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o4 = 32 * ot + 8 * q + 4 * hf;
            const f32x4 b = eb[q], sc = esc[q], sh = esh[q];
            if (UNPOOL) {
                const int child = o4 >> lc, ch = o4 & (c - 1);
                const int dst = X0 + 4 * ((pt * st + child) * ld0 + ch);
                const f32x4 old = F4(dst);
                f32x4 yv;
#pragma unroll
                for (int i = 0; i < 4; ++i) yv[i] = (old[i] + acc[4 * q + i] + b[i]) * sc[i] + sh[i];
                if (pt_place < nr2) F4(dst) = yv;
            } else {
                f32x4 yv;
#pragma unroll
                for (int i = 0; i < 4; ++i) yv[i] = (acc[4 * q + i] + b[i]) * sc[i] + sh[i];
                F4(X1 + 4 * (pt_place * ld1 + o4)) = yv;
            }
        }
    }
    lds_barrier();
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

constexpr int kWarm = 8;  // 16-B loads per thread per builder step of the L2 warm-up (64 KB per step and workgroup)

__global__ __launch_bounds__(kThreads) void erwin2_kernel(const float* __restrict__ tokens, int32_t* __restrict__ tree_idx,
                                                         uint8_t* __restrict__ tree_mask, int32_t* __restrict__ rot,
                                                         const float* __restrict__ W, const char* __restrict__ FR,
                                                         float* __restrict__ out, WTreeCfg tcfg, float* __restrict__ pos_out,
                                                         Head PL, int warm, int xcd_pack) {
    // xcd_pack = the number of clouds (else 0): an 8x grid of which every eighth workgroup works -- all clouds on ONE XCD,
    // sharing its L2 for the fragment blob they all stream (erwin.hip, erwin_kernel)
    if (xcd_pack && (blockIdx.x & 7)) return;
    const int cloud_ = xcd_pack ? blockIdx.x >> 3 : blockIdx.x, n_clouds_ = xcd_pack ? xcd_pack : gridDim.x;
    STAMP2(0);
    // The directory and the op table are read from the blob's own copy of the Plan: refuse a blob that was packed for
    // another descriptor or M (uniform exit before any barrier; the output goes out as NaN so that the caller SEES it --
    // with foreign offsets the kernel would address LDS and weights out of bounds instead)
    if (*reinterpret_cast<const unsigned*>(FR + offsetof(Plan, h) + offsetof(Head, stamp)) != PL.stamp) {
        for (int i = threadIdx.x; i < PL.M * PL.c0; i += kThreads) out[(size_t)cloud_ * PL.M * PL.c0 + i] = __int_as_float(0x7fc00000);
        return;
    }
    const int dirf_all = reinterpret_cast<const int*>(FR + offsetof(Plan, h) + offsetof(Head, dir))[min((int)(threadIdx.x & 63), (int)(sizeof(Dir) / 4) - 1)];
    const int dirf = dirf_all;
    const int cloud = cloud_, n_clouds = n_clouds_, d = PL.d, nl = PL.n_levels, M = PL.M, c0 = PL.c0;
    for (int i = threadIdx.x; i < PL.n_ops * kOpInts; i += kThreads)  // the op table (visible after the builder's barriers)
        I(DF(ops) + 4 * i) = reinterpret_cast<const int*>(FR + offsetof(Plan, ops))[i];
    {
        const WTreeLds sink = {DF(tmap), DFI(P, 0), {DFI(rotm, 0), DFI(rotm, 1), DFI(rotm, 2), DFI(rotm, 3)}};
        // Optional L2 warm-up under the builder's latency chain (HAET_ERWIN_WARM=1; OFF by default: this kernel requests a
        // phase's fragments a whole phase ahead, and with the warm-up the builder ran 8-20k cycles longer for blocks that
        // were not a cycle faster): step i folds what step i-1 requested into a checksum nothing reads, then requests
        // its own pieces -- unconditional, clamped addresses
        u32x4 pend[kWarm];
        unsigned sink_x = 0;
        const unsigned pieces = PL.total_bytes >> 4, pieces1 = PL.v1_bytes >> 4;
        const char* W1 = reinterpret_cast<const char*>(W) + PL.v1_lo;
#pragma unroll
        for (int k = 0; k < kWarm; ++k) pend[k] = u32x4{0u, 0u, 0u, 0u};
        auto step = [&](int i) {  // pieces 0 .. pieces-1: the fragment blob; then the resample Linears of the v1 blob
            if (!warm) return;  // (uniform)
#pragma unroll
            for (int k = 0; k < kWarm; ++k) sink_x ^= pend[k][0];
            const unsigned base = (unsigned)i * (kThreads * kWarm) + threadIdx.x;
#pragma unroll
            for (int k = 0; k < kWarm; ++k) {
                const unsigned pc = base + (unsigned)k * kThreads;
                const bool second = pc >= pieces;  // (a select between two addresses, not a branch around the load)
                const char* src = second ? W1 + (size_t)min(pc - pieces, pieces1 - 1) * 16 : FR + (size_t)pc * 16;
                pend[k] = *(gv4)src;
            }
        };
        w_tree_build(tcfg, nullptr, tokens, c0, cloud, n_clouds, tree_idx, tree_mask, rot, pos_out, sink, step);
#pragma unroll
        for (int k = 0; k < kWarm; ++k) sink_x ^= pend[k][0];
        asm volatile("" ::"v"(sink_x));
    }
    lds_barrier();
    STAMP2(1);
    // the cloud in tree order (erwin_flash.py:176-181), the first block's parameters: requested from HBM now, parked in
    // LDS after the position work below has covered their latency (<= 2 x 16 B per thread: M * c0 <= 4096)
    const int ng = M * (c0 >> 2);
    int first_off, first_len, first_pbuf;
    {
        const int opf = fetch_op(DF(ops), 0);
        first_off = OF(small_off), first_len = OF(small_len), first_pbuf = OF(pbuf);
    }
    f32x4 tokv[2], parv;
    int tokdst[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = min((int)threadIdx.x + u * kThreads, ng - 1);
        const int slot = i / (c0 >> 2), g = i - slot * (c0 >> 2);
        tokv[u] = *(gf4)(tokens + ((size_t)cloud * M + I(DF(tmap) + 4 * slot)) * c0 + 4 * g);
        tokdst[u] = DFI(X, 0) + 4 * (slot * DFI(ld, 0) + 4 * g);
    }
    parv = *(gf4)(W + first_off + min((int)threadIdx.x * 4, first_len - 4));
    for (int i = threadIdx.x; i < M; i += kThreads) I(DF(ident) + 4 * i) = i;
    for (int l = 0; l + 1 < nl; ++l) {  // pooled positions (pooling.py:29-33): mean of the children, sequential sum
        const int st = PL.stride[l], n2 = DFI(n, l + 1), nr2 = DFI(nr, l + 1);
        for (int i = threadIdx.x; i < n2 * d; i += kThreads) {
            const int pt = i / d, j = i - pt * d;
            float sum = 0.f;
            if (pt < nr2)
                for (int q = 0; q < st; ++q) sum += F(DFI(P, l) + 4 * ((pt * st + q) * d + j));
            F(DFI(P, l + 1) + 4 * i) = sum / (float)st;  // (phantom places of a padded level: 0)
        }
        if (PL.rotate)  // the builder wrote the rotated partition of the REAL tokens; phantom places keep themselves
            for (int i = nr2 + (int)threadIdx.x; i < n2; i += kThreads) I(DFI(rotm, l + 1) + 4 * i) = i;
        lds_barrier();
    }
    if (nl == 1) lds_barrier();  // (the identity map)
    for (int l = 0; l < nl; ++l) {  // relative positions of both partitions of every level
        ball_rel(DFI(P, l), DFI(n, l), d, PL.ball[l], DF(ident), DFI(rel, 2 * (l) + (0)));
        if (PL.rotate) ball_rel(DFI(P, l), DFI(n, l), d, PL.ball[l], DFI(rotm, l), DFI(rel, 2 * (l) + (1)));
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if ((int)threadIdx.x + u * kThreads < ng) F4(tokdst[u]) = tokv[u];
    if ((int)threadIdx.x * 4 < first_len) F4(DFI(par, first_pbuf) + 16 * threadIdx.x) = parv;
    lds_barrier();

    STAMP2(2);
    for (int oi = 0; oi < PL.n_ops; ++oi) {
        const int opf = fetch_op(DF(ops), oi);
        const int kind = OF(kind);
        int dirv = dirf_all;  // (opaque copy per op: keeps the compiler from hoisting ~40 loop-invariant v_readlane results
        asm volatile("" : "+v"(dirv));  //  into scalar registers for the whole loop -- they were spilled to scratch)
        const int dirf = dirv;
        int tid = threadIdx.x;  // (the same for the thread index: lane-derived addresses are re-made per op instead of being
        asm volatile("" : "+v"(tid));  //  hoisted out of the loop and spilled)
#ifdef HAET_STAMPS
        const bool stamp_on = oi == haet_stamp2_op;
#else
        const bool stamp_on = false;
#endif
        if (kind == OP_BLOCK) {
            if (OF(nc) == 1) {
                attn_phase<1>(tid, opf, dirf, FR, PL.total_bytes, d, stamp_on);
                mlp_phase<1>(tid, opf, dirf, FR, W, stamp_on);
            } else {
                attn_phase<2>(tid, opf, dirf, FR, PL.total_bytes, d, stamp_on);
                mlp_phase<2>(tid, opf, dirf, FR, W, stamp_on);
            }
        } else if (kind == OP_POOL) {
            resample<false>(tid, opf, dirf, W, d);
        } else {
            resample<true>(tid, opf, dirf, W, d);
        }
        STAMP2(3 + oi);
    }
    // node.x[tree_mask][argsort(tree_idx[tree_mask])]  (erwin_flash.py:193); every slot of a power-of-two cloud holds a token
    {
        const int X0 = DFI(X, 0), ld0 = DFI(ld, 0);
        for (int i = threadIdx.x; i < M * (c0 >> 2); i += kThreads) {
            const int slot = i / (c0 >> 2), g = i - slot * (c0 >> 2);
            *reinterpret_cast<f32x4*>(out + ((size_t)cloud * M + I(DF(tmap) + 4 * slot)) * c0 + 4 * g) = F4(X0 + 4 * (slot * ld0 + 4 * g));
        }
    }
    STAMP2(20);
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
    Plan PP = {};
    Head& P = PP.h;
    const int nl = D.n_levels;
    if (nl < 1 || nl > HAET_ERWIN_MAX_LEVELS || !D.decode || D.bn_train || (D.d != 2 && D.d != 3)) return false;
    if (M < 32 || M > 64 || (M & (M - 1))) return false;
    P.n_levels = nl, P.d = D.d, P.M = M, P.c0 = D.enc[0].c, P.rotate = D.rotate ? 1 : 0;
    int n = M, nlv[HAET_ERWIN_MAX_LEVELS], nreal[HAET_ERWIN_MAX_LEVELS];
    for (int l = 0; l < nl; ++l) {
        nreal[l] = n;
        nlv[l] = (n + 31) & ~31;  // whole tiles: a 16-token level (below level 0) is padded with phantom tokens
        if (n % 32 && (l == 0 || n != 16)) return false;
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
            if ((nlv[l] / 32) * L.heads > 8 || (nlv[l] / 32) * (L.hidden / 32) > 8) return false;
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
        if (no < kMaxOps) PP.ops[no] = o;
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
        if (l < nl - 1) simple(OP_POOL, l, &D.pool[l]);
    }
    for (int l = nl - 2; l >= 0; --l) {
        simple(OP_UNPOOL, l, &D.unpool[l]);
        blocks(l, D.dec[l]);
    }
    if (no > kMaxOps) return false;
    P.n_ops = no;
    // parameter buffers alternate over the BLOCK ops; every block stages its successor's group
    int max_small = 4, nb = 0, prev = -1;
    for (int i = 0; i < no; ++i) {
        Op& o = PP.ops[i];
        if (o.kind != OP_BLOCK) continue;
        o.pbuf = nb & 1;
        ++nb;
        max_small = max_small > o.small_len ? max_small : o.small_len;
        if (prev >= 0) PP.ops[prev].next_off = o.small_off, PP.ops[prev].next_len = o.small_len;
        prev = i;
    }
    if (nb == 0 || PP.ops[0].kind != OP_BLOCK || max_small > 2048) return false;  // (the prologue parks the first group with one 16-B load per thread)
    for (int l = 0; l + 1 < nl; ++l)
        if (2 * D.enc[l].c > 64 || D.enc[l + 1].c > 64 || 2 * D.d > 16) return false;  // resample(): <= 4 feature k-steps + 1
    // fragment sections
    unsigned off = kPlanBytes;
    P.frag_begin = off;
    for (int i = 0; i < no; ++i) {
        Op& o = PP.ops[i];
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
    {   // the resample Linears' planes in the v1 blob ([out][K~] bf16 hi + lo = out * K~ floats), for the warm-up
        long lo = -1, hi = 0;
        for (int i = 0; i < no; ++i) {
            const Op& o = PP.ops[i];
            if (o.kind == OP_BLOCK) continue;
            const int K = o.kind == OP_UNPOOL ? o.c2 + 2 * D.d : 2 * o.c + 2 * D.d, Oc = o.kind == OP_UNPOOL ? 2 * o.c : o.c2;
            const long a = (long)o.w * 4, b = a + (long)Oc * round32(K) * 4;
            lo = lo < 0 || a < lo ? a : lo;
            hi = b > hi ? b : hi;
        }
        P.v1_lo = lo < 0 ? 0u : (unsigned)lo;
        P.v1_bytes = lo < 0 ? 16u : (unsigned)(hi - lo);
    }
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
        T.nr[l] = nreal[l];
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
    }
    for (int l = nl; l < HAET_ERWIN_MAX_LEVELS; ++l) T.rotm[l] = -1;
    T.ident = take(4 * M);
    T.tmap = take(4 * M);
    T.par[0] = take(4 * max_small);
    T.par[1] = take(4 * max_small);
    T.slab = take(maxslab);
    T.ops = take(kMaxOps * (int)sizeof(Op));
    T.total = p;
    if (p + 8 * 1024 > 160 * 1024) return false;  // (+ the tree builder's static tables)
    {   // the blob carries a copy of this Plan (directory + op table, read by the kernel from the blob): stamp it, so that a
        // blob packed for another descriptor / M -- large enough to pass the size check -- is recognised by the kernel
        P.stamp = 0;
        unsigned hsh = 2166136261u;
        const unsigned char* b = reinterpret_cast<const unsigned char*>(&PP);
        for (size_t i = 0; i < sizeof(Plan); ++i) hsh = (hsh ^ b[i]) * 16777619u;
        P.stamp = hsh ? hsh : 1u;
    }
    *out = PP;
    return true;
}

}  // namespace e2

bool balltree_wave_ok(int M, int n_levels, const int* st);  // balltree_wave.hip
void rotation_matrix_f32(float angle_deg, int d, float* r9);  // balltree.hip

}  // namespace haet

using namespace haet;

#ifdef HAET_STAMPS
extern "C" int haet_debug_read_tree_stamps2(unsigned long long* host_out) {  // the builder's stamps inside THIS kernel
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(haet_tree_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -1;
}
extern "C" int haet_debug_read_stamps2(unsigned long long* host_out, int stamp_op) {
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(e2::haet_stamps2), sizeof(unsigned long long) * 128) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(e2::haet_stamp2_op), &stamp_op, sizeof(int)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" size_t haet_erwin_frag_bytes(const haet_erwin_desc* desc, int M) {
    static const bool off = getenv("HAET_ERWIN_V1") != nullptr;  // A/B switch: every shape on erwin.hip
    e2::Plan P;
    if (off || !desc || !e2::make_plan(*desc, M, &P)) return 0;
    int st[3] = {1, 1, 1};
    for (int l = 0; l < desc->n_levels - 1 && l < 3; ++l) st[l] = desc->pool[l].stride;
    if (!balltree_wave_ok(M, desc->n_levels, st)) return 0;
    return P.h.total_bytes;
}

extern "C" int haet_erwin_frag_pack(const float* weights, size_t weight_floats, const haet_erwin_desc* desc, int M,
                                    void* frags, size_t frag_bytes, haet_stream_t stream) {
    if (!weights || !desc || !frags) return fail(HAET_E_ARG, "erwin_frag_pack: null pointer");
    if (weight_floats * 4 < sizeof(haet_erwin_desc)) return fail(HAET_E_ARG, "erwin_frag_pack: weight blob smaller than its header");
    e2::Plan P;
    if (!e2::make_plan(*desc, M, &P)) return fail(HAET_E_ARG, "erwin_frag_pack: shape outside the fragment kernel's envelope");
    if (frag_bytes < P.h.total_bytes) return fail(HAET_E_ARG, "erwin_frag_pack: %zu bytes given, %u needed", frag_bytes, P.h.total_bytes);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(e2::e2_plan_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<char*>(frags), P);
    for (int i = 0; i < P.h.n_ops; ++i) {
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
    if (frag_bytes < P.h.total_bytes) return fail(HAET_E_ARG, "token_stage_frag: fragment blob too small");
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
    static const int warm = getenv("HAET_ERWIN_WARM") ? atoi(getenv("HAET_ERWIN_WARM")) : 0;
    const size_t bytes = (size_t)P.h.dir.total;
    static std::atomic<size_t> attr_set[64];
    std::atomic<size_t>& attr_dev = attr_set[PerDeviceOnce::bit()];
    if (bytes > 48 * 1024 && bytes > attr_dev.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(e2::erwin2_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return fail(HAET_E_LAUNCH, "token_stage_frag: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_dev.store(bytes, std::memory_order_release);
    }
    const char* pe = getenv("HAET_ERWIN_XCD_PACK");  // A/B switch, off by default ("1": all clouds on one XCD; measured without effect)
    const bool pack = n_clouds <= 32 && pe && pe[0] == '1';
    hipLaunchKernelGGL(e2::erwin2_kernel, dim3(pack ? 8 * n_clouds : n_clouds), dim3(e2::kThreads), bytes, as_stream(stream), S,
                       tree_idx, tree_mask, rot, weights, reinterpret_cast<const char*>(frags), out, cfg, pos_out, P.h, warm,
                       pack ? n_clouds : 0);
    return check_launch("erwin2_kernel");
}
