// Backward of the Erwin stage (haet_erwin_fwd / haet_token_stage_fwd) in ONE launch, one workgroup per cloud.
//
// The reference differentiates ~100 small autograd nodes per forward here (models/components/erwinflash/*); the
// interim path of this package did the same through a torch restatement (~600-700 launch-bound kernels, 2.2 ms of a
// 10 ms training step on 512 tokens).  This kernel walks the op list of erwin_kernel in reverse; every op is recomputed
// from its TAPED input (the forward kernel records the input tokens of each block / pool / unpool: 8 KB per op and
// cloud at the C3 shape) in exact fp32 on the vector ALUs and differentiated in place, all activations in LDS:
//
//   block (layers.py:30-32)     x_mid = x + proj(attn(qkv(RMSNorm(x) g1 + pe(rel_pos))));  y = x_mid + W3(W2 n2 * silu(W1 n2))
//     backward                  SwiGLU (hidden width in chunks) -> RMSNorm -> proj -> per-ball softmax attention
//                               (dq per query thread; dk, dv per key thread, recomputing the probabilities from the
//                               stored row statistics: no atomics, deterministic) -> qkv -> rel-pos PE -> RMSNorm
//   pool / unpool (pooling.py)  Linear over [children | rel_pos] + BatchNorm1d; eval: the running statistics; train:
//                               the batch statistics of the forward (haet_erwin_train.batch_stats) -- the two
//                               cross-cloud means of the BatchNorm backward are summed from every cloud's partial
//                               d_gamma / d_beta after a grid barrier
//
// Positions carry no gradient (attention.py:35, pooling.py:29,59: no_grad in the reference).  Linear layers:
//   y = x W^T + b   forward   gemm(T.WSTG, y, x, W^T)      thread = (output, 8 tokens), weights read coalesced from the
//   dx = dy W       backward  gemm(T.WSTG, dx, dy, W)      transposed / original fp32 copy in the blob
//   dW = dy^T x                wgrad               thread = one weight, loop over the cloud's tokens in LDS
// Every parameter gradient of a cloud is produced by exactly one thread in a fixed order and written to that cloud's
// slice of `grads`; the caller sums the slices.  Cost at the C3 shape: ~0.3 ms for all clouds (8 workgroups), against
// ~3.6 ms (train) for the restatement's forward + backward.
#include "common.h"

namespace haet {

extern __shared__ __attribute__((aligned(16))) char smem[];

constexpr int kBT = 512;   // threads
constexpr float kRmsEps = 1.1920928955078125e-07f;  // nn.RMSNorm(eps=None): finfo(float32).eps

__device__ __forceinline__ float* SM() { return reinterpret_cast<float*>(smem); }
#ifdef HAET_STAMPS
// Diagnostic build only (make EXTRA=-DHAET_STAMPS; tools/erwin_bwd_stamps.py): s_memtime at phase boundaries of the
// first block processed by cloud 0, written to a buffer nothing else reads.  Never quote this build's run time.
__device__ unsigned long long haet_bwd_stamps[64];
__device__ int haet_bwd_stamp_on;
#define BSTAMP(i)                                                                                         \
    do {                                                                                                  \
        if (blockIdx.x == 0 && threadIdx.x == 0 && haet_bwd_stamp_on) haet_bwd_stamps[i] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define BSTAMP(i) do { } while (0)
#endif

// counter barrier over the workgroups of the launch (same protocol as erwin.hip's; cdna_hip_programming.md G16)
__device__ __attribute__((noinline)) void bwd_grid_barrier(unsigned* counter, unsigned target, unsigned* tmo) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1u << 22)) {
                __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// Y[t*ldy + o] (=, +=) bias[o] + sum_k X[t*ldx + k] * Wg[k*ldw + o];  X, Y: float offsets into LDS; Wg, bias: global.
// The weights go through LDS in tiles of [K][ow] (<= kWStage floats, staged by all threads with one coalesced round
// trip): read straight from global inside the k-loop every eight k were a dependent L2 round trip for the whole wave
// (stamps: 5k cycles per round of a 32-deep product).  thread = (output o, TCH tokens): weights conflict-free across
// o, activations LDS broadcasts.  Called by ALL threads (workgroup barriers inside).
constexpr int kWStage = 4096;
template <int TCH>
__device__ __forceinline__ void gemm_tile(int Y, int ldy, int X, int ldx, int WST, int ow, int o0,
                                          const float* __restrict__ bias, int n, int K, bool accumulate) {
    float* sm = SM();
    const int ntc = (n + TCH - 1) / TCH;
    for (int u = threadIdx.x; u < ow * ntc; u += kBT) {
        const int o = u % ow, t0 = (u / ow) * TCH;
        float acc[TCH];
        int xo[TCH];
        const float b0 = bias ? bias[o0 + o] : 0.f;
#pragma unroll
        for (int i = 0; i < TCH; ++i) {
            xo[i] = X + min(t0 + i, n - 1) * ldx;
            acc[i] = accumulate ? sm[Y + min(t0 + i, n - 1) * ldy + o0 + o] : b0;
        }
        int k = 0;
        for (; k + 4 <= K; k += 4) {
            float w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = sm[WST + (k + j) * ow + o];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < TCH; ++i) acc[i] = fmaf(sm[xo[i] + k + j], w[j], acc[i]);
            }
        }
        for (; k < K; ++k) {
            const float w = sm[WST + k * ow + o];
#pragma unroll
            for (int i = 0; i < TCH; ++i) acc[i] = fmaf(sm[xo[i] + k], w, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < TCH; ++i)
            if (t0 + i < n) sm[Y + (t0 + i) * ldy + o0 + o] = acc[i];
    }
}
// Register-tiled variant: thread = FOUR consecutive outputs x TT tokens; per four k it reads 4 (weights) + TT
// (activations) 16-B vectors for 16 TT multiply-adds -- the one-read-per-FMA loop above is bound by LDS issue (stamps:
// 10k cycles for a 64 x 32 x 48 product).  Needs K, ow, ldx, X multiples of 4.
template <int TT>
__device__ __forceinline__ void gemm_tile4(int Y, int ldy, int X, int ldx, int WST, int ow, int o0,
                                           const float* __restrict__ bias, int n, int K, bool accumulate) {
    typedef __attribute__((ext_vector_type(4))) float f4;
    float* sm = SM();
    const int ntc = (n + TT - 1) / TT, og = ow >> 2;
    for (int u = threadIdx.x; u < og * ntc; u += kBT) {
        const int o = (u % og) << 2, t0 = (u / og) * TT;
        f4 acc[TT];
        int xo[TT];
        f4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (bias) b4 = f4{bias[o0 + o], bias[o0 + o + 1], bias[o0 + o + 2], bias[o0 + o + 3]};
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            const int t = min(t0 + i, n - 1);
            xo[i] = X + t * ldx;
            if (accumulate) {
                const int yo = Y + t * ldy + o0 + o;
                acc[i] = f4{sm[yo], sm[yo + 1], sm[yo + 2], sm[yo + 3]};
            } else {
                acc[i] = b4;
            }
        }
        for (int k = 0; k < K; k += 4) {
            f4 w[4], xv[TT];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const f4*>(sm + WST + (k + j) * ow + o);
#pragma unroll
            for (int i = 0; i < TT; ++i) xv[i] = *reinterpret_cast<const f4*>(sm + xo[i] + k);
#pragma unroll
            for (int i = 0; i < TT; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i] += xv[i][j] * w[j];
            }
        }
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            if (t0 + i < n) {
                const int yo = Y + (t0 + i) * ldy + o0 + o;
                sm[yo] = acc[i][0], sm[yo + 1] = acc[i][1], sm[yo + 2] = acc[i][2], sm[yo + 3] = acc[i][3];
            }
        }
    }
}
__device__ __forceinline__ void gemm(int WST, int Y, int ldy, int X, int ldx, const float* __restrict__ Wg, int ldw,
                                     const float* __restrict__ bias, int n, int K, int O, bool accumulate) {
    float* sm = SM();
    int ot = (WST >> 16) / K;  // WST = offset | stage floats << 16 (BwdDir::WSTG)
    WST &= 0xffff;
    ot = ot >= 32 ? (ot & ~31) : (ot >= 16 ? 16 : (ot >= 8 ? 8 : (ot > 0 ? ot : 1)));
    for (int o0 = 0; o0 < O; o0 += ot) {
        const int ow = min(ot, O - o0);
        __syncthreads();  // the previous tile's readers (and the producers of X) are done
        for (int e = threadIdx.x; e < K * ow; e += kBT) {
            const int k = e / ow, o = e - k * ow;
            sm[WST + e] = Wg[(size_t)k * ldw + o0 + o];
        }
        __syncthreads();
        if (((K | ow | ldx | X) & 3) == 0) {  // register tiles of 4 outputs x TT tokens, as many units as threads
            const int og = ow >> 2;
            if (og * ((n + 3) / 4) >= kBT) gemm_tile4<4>(Y, ldy, X, ldx, WST, ow, o0, bias, n, K, accumulate);
            else if (og * ((n + 1) / 2) >= kBT) gemm_tile4<2>(Y, ldy, X, ldx, WST, ow, o0, bias, n, K, accumulate);
            else gemm_tile4<1>(Y, ldy, X, ldx, WST, ow, o0, bias, n, K, accumulate);
            continue;
        }
        // enough (output, token-chunk) units for all 512 threads: narrow tiles take fewer tokens per thread
        if (ow * ((n + 7) / 8) >= kBT) gemm_tile<8>(Y, ldy, X, ldx, WST, ow, o0, bias, n, K, accumulate);
        else if (ow * ((n + 3) / 4) >= kBT) gemm_tile<4>(Y, ldy, X, ldx, WST, ow, o0, bias, n, K, accumulate);
        else gemm_tile<2>(Y, ldy, X, ldx, WST, ow, o0, bias, n, K, accumulate);
    }
}

// dW[o*ldw + k] = sum_t A[t*lda + o] * B[t*ldb + k]  (k < K), db[o] = sum_t A[t*lda + o]  (global outputs, may be null)
// thread = one output o and FOUR consecutive k (one 16-B LDS read of B per token and four FMAs per A read) where K and
// the operands' alignment allow, one weight otherwise; the token loop is unrolled so that its LDS reads overlap.
__device__ __forceinline__ void wgrad(float* __restrict__ dW, int ldw, float* __restrict__ db, int A, int lda, int B,
                                      int ldb, int n, int O, int K) {
    typedef __attribute__((ext_vector_type(4))) float f4;
    const float* sm = SM();
    if (dW && (K & 3) == 0 && (ldb & 3) == 0 && (B & 3) == 0) {
        const int K4 = K >> 2;
        for (int e = threadIdx.x; e < O * K4; e += kBT) {
            const int o = e / K4, k = (e - o * K4) << 2;
            f4 s = {0.f, 0.f, 0.f, 0.f};
            int t = 0;
            for (; t + 4 <= n; t += 4) {
                float av[4];
                f4 bv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    av[j] = sm[A + (t + j) * lda + o];
                    bv[j] = *reinterpret_cast<const f4*>(sm + B + (t + j) * ldb + k);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) s += av[j] * bv[j];
            }
            for (; t < n; ++t) s += sm[A + t * lda + o] * *reinterpret_cast<const f4*>(sm + B + t * ldb + k);
            float* d = dW + (size_t)o * ldw + k;
            d[0] = s[0], d[1] = s[1], d[2] = s[2], d[3] = s[3];
        }
    } else if (dW) {
        for (int e = threadIdx.x; e < O * K; e += kBT) {
            const int o = e / K, k = e - o * K;
            float s0 = 0.f;
#pragma unroll 8
            for (int t = 0; t < n; ++t) s0 = fmaf(sm[A + t * lda + o], sm[B + t * ldb + k], s0);
            dW[(size_t)o * ldw + k] = s0;
        }
    }
    if (db) {
        for (int o = threadIdx.x; o < O; o += kBT) {
            float s = 0.f;
#pragma unroll 8
            for (int t = 0; t < n; ++t) s += sm[A + t * lda + o];
            db[o] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the block's Linear products on the bf16 matrix cores (split bf16 x 3, fp32 accumulate: ~2^-16 per product, the
// forward's grade).  The vector-ALU routines above cost 5-15k cycles per product whatever its size -- a weight-staging
// round trip through LDS with two barriers, then one LDS read per multiply-add -- and a block has ~40 of them.  Here a wave
// owns a 32 x 32 output tile: activations come from LDS as 16-B row pieces, weights from the blob copy whose k index is
// contiguous ([out][in] for y = x W^T, the transposed copy for dx = dy W) as 16-B global loads requested up front,
// both split to {hi, lo} in registers; no staging, no barrier inside.  Shapes outside 32 | n, 16 | K, 32 | O (the pooling
// Linears' K = 70) keep the routines above.
typedef __attribute__((ext_vector_type(8))) __bf16 eb_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 eb_bf16x2;
typedef __attribute__((ext_vector_type(16))) float eb_f16v;
typedef __attribute__((ext_vector_type(4))) float eb_f4;
typedef __attribute__((ext_vector_type(2))) float eb_f2;
typedef __attribute__((ext_vector_type(4))) unsigned eb_u4;
struct EbFrag {
    eb_u4 hi, lo;
};
__device__ __forceinline__ unsigned eb_pack(float a, float b) {
    eb_f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, eb_bf16x2));
}
__device__ __forceinline__ EbFrag eb_split8(const float (&v)[8]) {
    EbFrag f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h, l;
        split_bf16_pair(v[2 * i], v[2 * i + 1], h, l);
        f.hi[i] = h, f.lo[i] = l;
    }
    return f;
}
__device__ __forceinline__ EbFrag eb_split(const eb_f4 a, const eb_f4 b) {
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return eb_split8(v);
}
__device__ __forceinline__ eb_f16v eb_mfma3(const EbFrag& a, const EbFrag& b, eb_f16v c) {
    const eb_bf16x8 ah = __builtin_bit_cast(eb_bf16x8, a.hi), al = __builtin_bit_cast(eb_bf16x8, a.lo);
    const eb_bf16x8 bh = __builtin_bit_cast(eb_bf16x8, b.hi), bl = __builtin_bit_cast(eb_bf16x8, b.lo);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
    return c;
}
// One barrier-separated PHASE of independent 32 x 32 tile jobs, dealt round-robin to the eight waves:
//   gemm part   Y[m][t][o] (=, +=) bias[m][o] + sum_k X[t][k] W[m][o][k]  (+ sum_k X2[t][k] W2[o][k])     m < nmat
//               (W*: the blob copy with k contiguous; X, Y in LDS; the accumulator is Y^T: rows = outputs, lane = token)
//   wgrad part  dW[m][o][k] = sum_t A[m][t][o] B[t][k]   -- the contraction runs over the tokens, i.e. over the ROWS of both
//               LDS buffers: a lane gathers its column's eight tokens of a k-step with eight 4-B reads (consecutive lanes =
//               consecutive columns: conflict-free); gmode: B = B[t][k] * silu(B2[t][k]) computed in the gather (the
//               SwiGLU gate is never stored)
//   db part     db[m][o] = sum_t A[m][t][o]   (eight partial sums per column, three shuffle steps)
// The weight pieces of a tile are requested in groups of six k-steps, ahead of the activations' LDS reads.  NOT inlined, and
// the job descriptor travels through LDS (two alternating slots in the idle weight-staging region, written by one thread,
// read by everybody behind the phase's opening barrier): inlined at its seven call sites the routine was 38k lines of ISA
// with 161 spill operations, and as a by-value argument the descriptor went through scratch memory.
struct EbJob {
    int g_on, Y, Ystep, ldy, X, ldx, X2, n, K, K2, O, nmat, acc;
    int tnks, tot0, ts0, tnks2, ts02;  // W / Wb / W2 are TILED copies: k-steps per full row, row-tile and k-step offsets of the sub-block
    const float *W, *Wb, *W2, *bias, *biasb;
    int w_on, ldw, A, Astep, lda, B, B2, ldb, wO, wK, wmat, gmode;
    float *dW, *dWb, *db, *dbb;
    int stamp;  // (diagnostic build: stamp this phase's inner steps into slots 40..)
};
__device__ __forceinline__ int EU(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename P>
__device__ __forceinline__ P* EUP(P* p) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return reinterpret_cast<P*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void eb_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __attribute__((noinline)) void eb_phase(int JOB) {
    float* sm = SM();
    typedef const __attribute__((address_space(1))) eb_f4* gf4;
#ifdef HAET_STAMPS
    const unsigned long long st_in = __builtin_amdgcn_s_memtime();
#endif
    eb_lds_barrier();  // the job descriptor is in LDS; the producers of the phase's inputs (and the last readers of its outputs) are done
    const EbJob j = *reinterpret_cast<const EbJob*>(sm + EU(JOB));
#ifdef HAET_STAMPS
    // (the flag is read ONCE: as a global read inside every stamp it cost ~2k cycles per stamp and distorted the picture)
    const bool st_on = EU(j.stamp) != 0 && blockIdx.x == 0 && threadIdx.x == 0 && haet_bwd_stamp_on;
#define ESTAMP(i) do { if (st_on) haet_bwd_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
    if (st_on) haet_bwd_stamps[52] = st_in;
    ESTAMP(53);
#else
#define ESTAMP(i) do { } while (0)
#endif
    // uniform copies
    const int g_on = EU(j.g_on), Y0 = EU(j.Y), Ystep = EU(j.Ystep), ldy = EU(j.ldy), X = EU(j.X), ldx = EU(j.ldx), X2 = EU(j.X2);
    const int n = EU(j.n), K = EU(j.K), K2 = EU(j.K2), O = EU(j.O), nmat = EU(j.nmat), accm = EU(j.acc);
    const int tnks = EU(j.tnks), tot0 = EU(j.tot0), ts0 = EU(j.ts0), tnks2 = EU(j.tnks2), ts02 = EU(j.ts02);
    const float *Wa = EUP(j.W), *Wb = EUP(j.Wb), *W2 = EUP(j.W2), *ba = EUP(j.bias), *bb = EUP(j.biasb);
    const int w_on = EU(j.w_on), ldw = EU(j.ldw), A0 = EU(j.A), Astep = EU(j.Astep), lda = EU(j.lda), B = EU(j.B), B2 = EU(j.B2), ldb = EU(j.ldb);
    const int wO = EU(j.wO), wK = EU(j.wK), wmat = EU(j.wmat), gmode = EU(j.gmode);
    float *dWa = EUP(j.dW), *dWb = EUP(j.dWb), *dba = EUP(j.db), *dbb = EUP(j.dbb);
    const int lane = threadIdx.x & 63, wave = EU(threadIdx.x >> 6), l31 = lane & 31, hf = lane >> 5;
    constexpr int NWV = kBT / 64, KG = 6;
    const int wkt = wK >> 5, wtiles = w_on ? wmat * (wO >> 5) * wkt : 0;
    const int ot_n = O >> 5, gper = (n >> 5) * ot_n, gtiles = g_on ? nmat * gper : 0;
    const int total = wtiles + gtiles;
    const int ks1 = K >> 4, ks2 = K2 >> 4, ks_n = ks1 + ks2;  // the k-steps of the second pair follow the first's
    // decode a gemm tile and fetch one group of its weight pieces
    struct GT {
        int m, tt, ot;
    };
    auto gdecode = [&](int tile) {
        GT g;
        const int r = tile - wtiles;
        g.m = r / gper;
        const int q = r - g.m * gper;
        g.tt = q / ot_n;
        g.ot = q - g.tt * ot_n;
        return g;
    };
    auto wfetch = [&](const GT& g, int s0, eb_f4 (&w0)[KG], eb_f4 (&w1)[KG]) {
        // Weights come from the TILE-ORDERED copies behind the row-major tensors of the fp32 blob (pack.py, pack_erwin_f32;
        // round 4): per (32-row tile ot, 16-wide k-step s) two 1-KB pieces -- lane (r, hf) holds W[32 ot + r][16 s + 8 hf
        // + 4 which ..] -- so that a wave load is 1 KB of consecutive bytes.  From the row-major tensor the same load was 32
        // rows x 2 x 16 B with neighbouring lanes in different rows (the shape that held erwin.hip's Linears at 11 B/cycle
        // per CU): bwd_erwin 0.91 -> 0.81 ms in a same-box A/B of the access shape alone.
        const float* r1 = (g.m ? Wb : Wa) + ((size_t)(tot0 + g.ot) * tnks + ts0) * 512 + lane * 4;
        const float* r2 = W2 ? W2 + ((size_t)(tot0 + g.ot) * tnks2 + ts02) * 512 + lane * 4 : r1;
#pragma unroll
        for (int u = 0; u < KG; ++u) {  // (clamped: unconditional loads, an address select instead of a branch)
            const int s = min(s0 + u, ks_n - 1);
            const float* wr = s < ks1 ? r1 + 512 * s : r2 + 512 * (s - ks1);
            w0[u] = *(gf4)wr;
            w1[u] = *(gf4)(wr + 256);
        }
    };
    eb_f4 w0[KG], w1[KG];
    ESTAMP(54);
    for (int tile = wave; tile < total; tile += NWV) {
        eb_f16v acc;
        if (tile < wtiles) {
            // ---- weight-gradient tile
            const int m = tile / ((wO >> 5) * wkt), q = tile - m * ((wO >> 5) * wkt);
            const int ot = q / wkt, kt = q - ot * wkt;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const int ao = A0 + m * Astep + 32 * ot + l31, bo = B + 32 * kt + l31, bo2 = B2 + 32 * kt + l31;
            ESTAMP(61);
            for (int s = 0; s < (n >> 4); ++s) {
                float av[8], bv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int t = 16 * s + 8 * hf + e;
                    av[e] = sm[ao + t * lda];
                    bv[e] = sm[bo + t * ldb];
                    if (gmode) {  // g = h2 silu(h1): B holds h2, B2 holds h1
                        const float h1 = sm[bo2 + t * ldb];  // silu via v_exp + v_rcp (the IEEE division was ~10 instructions per element)
                        bv[e] *= h1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * h1));
                    }
                }
                acc = eb_mfma3(eb_split8(av), eb_split8(bv), acc);
            }
            // (explicitly GLOBAL stores: the pointers came through the job descriptor in LDS, i.e. generic, and a flat store
            // counts in lgkmcnt too -- the next phase's opening barrier then waited for these stores' round trip)
            ESTAMP(62);
            typedef __attribute__((address_space(1))) float gfl;
            gfl* dW = (gfl*)(m ? dWb : dWa);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dW[(size_t)(32 * ot + (r & 3) + 8 * (r >> 2) + 4 * hf) * ldw + 32 * kt + l31] = acc[r];
        } else {
            // ---- gemm tile
            const GT g = gdecode(tile);
            const int Yb = Y0 + g.m * Ystep, t = 32 * g.tt + l31;
            const float* bias = g.m ? bb : ba;
            wfetch(g, 0, w0, w1);  // (ahead of the accumulator's initial values: both are memory round trips)
            ESTAMP(59);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o4 = 32 * g.ot + 8 * q + 4 * hf;
                eb_f4 v = {0.f, 0.f, 0.f, 0.f};
                if (accm) v = *reinterpret_cast<const eb_f4*>(sm + Yb + t * ldy + o4);
                else if (bias) v = *(gf4)(bias + o4);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[4 * q + i] = v[i];
            }
            ESTAMP(55);
            for (int s0 = 0; s0 < ks_n; s0 += KG) {
                if (s0) wfetch(g, s0, w0, w1);
                ESTAMP(56);
#pragma unroll
                for (int u = 0; u < KG; ++u) {
                    const int s = s0 + u;
                    if (s < ks_n) {
                        const int xr = s < ks1 ? X + t * ldx + 16 * s + 8 * hf : X2 + t * ldx + 16 * (s - ks1) + 8 * hf;
                        const eb_f4 x0 = *reinterpret_cast<const eb_f4*>(sm + xr);
                        const eb_f4 x1 = *reinterpret_cast<const eb_f4*>(sm + xr + 4);
                        acc = eb_mfma3(eb_split(w0[u], w1[u]), eb_split(x0, x1), acc);
                    }
                }
            }
            ESTAMP(60);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const eb_f4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                *reinterpret_cast<eb_f4*>(sm + Yb + t * ldy + 32 * g.ot + 8 * q + 4 * hf) = v;
            }
        }
    }
    ESTAMP(57);
    // ---- bias gradients: column sums of A[m] over the tokens, eight partial sums per column
    if (dba) {
        for (int idx = threadIdx.x; idx < wmat * wO * 8; idx += kBT) {
            const int part = idx & 7, col = idx >> 3, m = col / wO, o = col - m * wO;
            float sacc = 0.f;
            for (int t = part; t < n; t += 8) sacc += sm[A0 + m * Astep + t * lda + o];
            sacc += __shfl_xor(sacc, 1);
            sacc += __shfl_xor(sacc, 2);
            sacc += __shfl_xor(sacc, 4);
            if (part == 0) ((__attribute__((address_space(1))) float*)(m ? dbb : dba))[o] = sacc;
        }
    }
    ESTAMP(58);
}
// run one phase: descriptor -> LDS slot (alternating: a slot is rewritten only after a full phase has passed), call
#define EB_RUN(j)                                                                          \
    do {                                                                                   \
        const int slot_ = T.WST + (eb_seq & 1) * 64;                                       \
        ++eb_seq;                                                                          \
        if (threadIdx.x == 0) *reinterpret_cast<EbJob*>(SM() + slot_) = (j);               \
        eb_phase(slot_);                                                                   \
    } while (0)
static_assert(sizeof(EbJob) <= 64 * 4, "job slot");

struct BwdArgs {
    const float* g_out;
    const float* tape;
    const float* pos;
    const int32_t* tree_idx;
    const uint8_t* tree_mask;
    const int32_t* rot;
    const float* Wf;        // fp32 blob (begins with haet_erwin_f32_layout)
    const float* Wfwd;      // forward blob (begins with haet_erwin_desc)
    const float* bn_stats;  // train: [2*MAX_LEVELS][2][cmax] mean, invstd of the forward
    unsigned* bar;          // train: grid-barrier counter (+1: timeout word), zeroed before the launch
    float* g_in;
    float* grads;           // [n_clouds][total]
    int M, Mp, n_ops, tape_op_floats, cmax, HB;
};

// LDS directory (float offsets), computed by the host
struct BwdDir {
    int dX[HAET_ERWIN_MAX_LEVELS], SK[HAET_ERWIN_MAX_LEVELS], P[HAET_ERWIN_MAX_LEVELS], PERM[HAET_ERWIN_MAX_LEVELS];
    int N[HAET_ERWIN_MAX_LEVELS];
    int IDENT, XIN, N1, AO, XMID, N2, TMP, QKV, HREG, R1, R2, ST, REL, WST, WSTG, total;
};

// whether the per-token reductions can run with c / 4 lanes per token (a power of two of lanes inside one wave, 16-B aligned rows)
__device__ __forceinline__ bool lanes_per_token(int c) {
    const int G = c >> 2;
    return (c & 3) == 0 && G >= 1 && G <= 64 && (G & (G - 1)) == 0;
}

// ---- RMSNorm forward: out[t][k] = x[t][k] r[t] g[k] (+ pe_b[k] + sum_j rel[t][j] pe_w[k][j]),  r -> R[t]
__device__ __forceinline__ void rms_fwd(int OUT, int X, int R, const float* __restrict__ g, int n, int c, int REL, int d,
                                        const float* __restrict__ pe_w, const float* __restrict__ pe_b) {
    float* sm = SM();
    if (lanes_per_token(c)) {
        // c / 4 lanes per token, one 16-B read each, the token's sum by xor-shuffles (round 4: one thread per token walking
        // its c channels read the SAME LDS bank as every other thread of its wave -- 64-way conflicts, 2-4k cycles per norm)
        const int G = c >> 2;
        for (int base = 0; base < n * G; base += kBT) {
            const int i = base + (int)threadIdx.x;
            const bool ok = i < n * G;
            const int t = ok ? i / G : 0, k4 = ok ? i - t * G : 0;
            const eb_f4 v = *reinterpret_cast<const eb_f4*>(sm + X + t * c + 4 * k4);
            float ss = ok ? (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]) : 0.f;
            for (int off = 1; off < G; off <<= 1) ss += __shfl_xor(ss, off);
            if (ok && k4 == 0) sm[R + t] = rsqrtf(ss / (float)c + kRmsEps);
        }
    } else {
        for (int t = threadIdx.x; t < n; t += kBT) {
            float ss = 0.f;
            for (int k = 0; k < c; ++k) ss = fmaf(sm[X + t * c + k], sm[X + t * c + k], ss);
            sm[R + t] = rsqrtf(ss / (float)c + kRmsEps);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n * c; i += kBT) {
        const int t = i / c, k = i - t * c;
        float v = sm[X + i] * sm[R + t] * g[k];
        if (pe_w) {
            v += pe_b[k];
            for (int j = 0; j < d; ++j) v = fmaf(sm[REL + t * d + j], pe_w[k * d + j], v);
        }
        sm[OUT + i] = v;
    }
    __syncthreads();
}

// ---- RMSNorm backward: DX[t][k] += r (dn g) - x r^3 mean_k(dn g x);  dg[k] = sum_t dn[t][k] x[t][k] r[t]
__device__ __forceinline__ void rms_bwd(int DX, int DN, int X, int R, const float* __restrict__ g, float* __restrict__ dg,
                                        int n, int c, int SCR, int SCR2 /* kBT floats of scratch, or -1 */) {
    float* sm = SM();
    const int P = kBT / c;  // token groups of the dg sums
    if (lanes_per_token(c) && ((X | DN) & 3) == 0 && P >= 1 && SCR2 >= 0) {
        // (round 4) per-token sums with c / 4 lanes per token; dg with kBT / c token groups per channel: partial sums to LDS,
        // then one thread per channel adds them in a fixed order.  The serial forms below took 12-14k cycles per call.
        const int G = c >> 2;
        for (int base = 0; base < n * G; base += kBT) {
            const int i = base + (int)threadIdx.x;
            const bool ok = i < n * G;
            const int t = ok ? i / G : 0, k4 = ok ? i - t * G : 0;
            const eb_f4 dn = *reinterpret_cast<const eb_f4*>(sm + DN + t * c + 4 * k4);
            const eb_f4 xv = *reinterpret_cast<const eb_f4*>(sm + X + t * c + 4 * k4);
            const eb_f4 gv = *reinterpret_cast<const eb_f4*>(g + 4 * k4);
            float sv = ok ? (dn[0] * gv[0] * xv[0] + dn[1] * gv[1] * xv[1]) + (dn[2] * gv[2] * xv[2] + dn[3] * gv[3] * xv[3]) : 0.f;
            for (int off = 1; off < G; off <<= 1) sv += __shfl_xor(sv, off);
            if (ok && k4 == 0) sm[SCR + t] = sv / (float)c;
        }
        const int per = (n + P - 1) / P;
        for (int i = threadIdx.x; i < P * c; i += kBT) {
            const int part = i / c, k = i - part * c;
            float sv = 0.f;
            for (int t = part * per; t < min(n, (part + 1) * per); ++t) sv = fmaf(sm[DN + t * c + k] * sm[X + t * c + k], sm[R + t], sv);
            sm[SCR2 + i] = sv;
        }
        __syncthreads();
        for (int k = threadIdx.x; k < c; k += kBT) {
            float sv = 0.f;
            for (int part = 0; part < P; ++part) sv += sm[SCR2 + part * c + k];
            dg[k] = sv;
        }
    } else {
        for (int t = threadIdx.x; t < n; t += kBT) {
            float sv = 0.f;
            for (int k = 0; k < c; ++k) sv = fmaf(sm[DN + t * c + k] * g[k], sm[X + t * c + k], sv);
            sm[SCR + t] = sv / (float)c;
        }
        for (int k = threadIdx.x; k < c; k += kBT) {
            float sv = 0.f;
            for (int t = 0; t < n; ++t) sv = fmaf(sm[DN + t * c + k] * sm[X + t * c + k], sm[R + t], sv);
            dg[k] = sv;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n * c; i += kBT) {
        const int t = i / c, k = i - t * c;
        const float r = sm[R + t];
        sm[DX + i] += r * (sm[DN + i] * g[k]) - sm[X + i] * (r * r * r) * sm[SCR + t];
    }
    __syncthreads();
}

// rel[tok][j] = P[tok][j] - mean over its ball (attention.py:35-40); ball b = places [b*ball, ..) of `perm`
__device__ __forceinline__ void rel_pos(int REL, int P, int PERM, int n, int d, int ball) {
    float* sm = SM();
    const int* pm = reinterpret_cast<const int*>(smem) + PERM;
    const int nb = n / ball;
    if (ball <= 64 && (ball & (ball - 1)) == 0 && n % ball == 0 && (n & 63) % ball == 0) {
        // (round 4) one thread per (dimension, place): the places of a ball are neighbouring lanes, its mean a xor-shuffle sum
        // (the serial form below had n / ball * d threads -- six at C3 -- walking a ball twice: 6k cycles per block)
        const int total = (n * d + 63) & ~63;  // whole waves: every lane takes part in the shuffles
        for (int i = threadIdx.x; i < total; i += kBT) {
            const int j = i / n, pl = i - j * n;
            const bool ok = j < d;
            const int tok = ok ? pm[pl] : 0;
            const float v = ok ? sm[P + tok * d + j] : 0.f;
            float sum = v;
            for (int off = 1; off < ball; off <<= 1) sum += __shfl_xor(sum, off);
            if (ok) sm[REL + tok * d + j] = v - sum / (float)ball;
        }
        __syncthreads();
        return;
    }
    for (int i = threadIdx.x; i < nb * d; i += kBT) {
        const int b = i / d, j = i - b * d;
        float s = 0.f;
        for (int p = 0; p < ball; ++p) s += sm[P + pm[b * ball + p] * d + j];
        const float mean = s / (float)ball;
        for (int p = 0; p < ball; ++p) {
            const int tok = pm[b * ball + p];
            sm[REL + tok * d + j] = sm[P + tok * d + j] - mean;
        }
    }
    __syncthreads();
}

// ---- per-ball softmax attention (attention.py:68-84), thread = (place, head), the head dim a template parameter so
// that q / k / v rows live in registers and their LDS reads are issued back to back (with a run-time head dim every
// multiply waited for its own two LDS reads: 350k of a block's 700k cycles).  Row statistics (max, 1/sum) -> ST.
template <int HD>
__device__ __forceinline__ void attn_fwd_t(const BwdDir& T, int QKV, int AO, int n, int c, int heads, int ball, int PERM,
                                           float scale) {
    float* sm = SM();
    const int* pm = reinterpret_cast<const int*>(smem) + PERM;
    for (int i = threadIdx.x; i < n * heads; i += kBT) {
        const int place = i / heads, h = i - place * heads;
        const int tq = pm[place], b0 = (place / ball) * ball;
        float q[HD], acc[HD];
#pragma unroll
        for (int e = 0; e < HD; ++e) {
            q[e] = sm[QKV + tq * 3 * c + h * HD + e] * scale;
            acc[e] = 0.f;
        }
        float m = -INFINITY;
        for (int p = 0; p < ball; ++p) {
            const int ko = QKV + pm[b0 + p] * 3 * c + c + h * HD;
            float sv = 0.f;
#pragma unroll
            for (int e = 0; e < HD; ++e) sv = fmaf(q[e], sm[ko + e], sv);
            m = fmaxf(m, sv);
        }
        float l = 0.f;
        for (int p = 0; p < ball; ++p) {
            const int tk = pm[b0 + p];
            const int ko = QKV + tk * 3 * c + c + h * HD, vo = ko + c;
            float sv = 0.f;
#pragma unroll
            for (int e = 0; e < HD; ++e) sv = fmaf(q[e], sm[ko + e], sv);
            const float pr = __expf(sv - m);
            l += pr;
#pragma unroll
            for (int e = 0; e < HD; ++e) acc[e] = fmaf(pr, sm[vo + e], acc[e]);
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int e = 0; e < HD; ++e) sm[AO + tq * c + h * HD + e] = acc[e] * inv;
        sm[T.ST + 3 * (tq * heads + h)] = m;
        sm[T.ST + 3 * (tq * heads + h) + 1] = inv;
    }
    __syncthreads();
}

// backward: GO = dL/d(attention output) [n][c]; DQ [n][3c] receives dq | dk | dv.
//   query thread: D_i = sum_j P_ij dP_ij, dq_i = scale (sum_j P_ij dP_ij k_j - D_i sum_j P_ij k_j)   (one pass)
//   key thread:   dk_j = scale sum_i P_ij (dP_ij - D_i) q_i,  dv_j = sum_i P_ij dO_i  (probabilities recomputed from the
//                 stored row statistics: no atomics, a fixed summation order)
template <int HD>
__device__ __forceinline__ void attn_bwd_t(const BwdDir& T, int QKV, int GO, int DQ, int n, int c, int heads, int ball,
                                           int PERM, float scale) {
    float* sm = SM();
    const int* pm = reinterpret_cast<const int*>(smem) + PERM;
    for (int i = threadIdx.x; i < n * heads; i += kBT) {
        const int place = i / heads, h = i - place * heads;
        const int tq = pm[place], b0 = (place / ball) * ball;
        const float m = sm[T.ST + 3 * (tq * heads + h)], inv = sm[T.ST + 3 * (tq * heads + h) + 1];
        float q[HD], go[HD], a1[HD], a2[HD];
#pragma unroll
        for (int e = 0; e < HD; ++e) {
            q[e] = sm[QKV + tq * 3 * c + h * HD + e] * scale;
            go[e] = sm[GO + tq * c + h * HD + e];
            a1[e] = a2[e] = 0.f;
        }
        float Dsum = 0.f;
        for (int p = 0; p < ball; ++p) {
            const int tk = pm[b0 + p];
            const int ko = QKV + tk * 3 * c + c + h * HD, vo = ko + c;
            float kv[HD];
            float sv = 0.f, dp = 0.f;
#pragma unroll
            for (int e = 0; e < HD; ++e) {
                kv[e] = sm[ko + e];
                sv = fmaf(q[e], kv[e], sv);
                dp = fmaf(go[e], sm[vo + e], dp);
            }
            const float pr = __expf(sv - m) * inv;
            Dsum = fmaf(pr, dp, Dsum);
#pragma unroll
            for (int e = 0; e < HD; ++e) {
                a1[e] = fmaf(pr * dp, kv[e], a1[e]);
                a2[e] = fmaf(pr, kv[e], a2[e]);
            }
        }
        sm[T.ST + 3 * (tq * heads + h) + 2] = Dsum;
#pragma unroll
        for (int e = 0; e < HD; ++e) sm[DQ + tq * 3 * c + h * HD + e] = scale * (a1[e] - Dsum * a2[e]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n * heads; i += kBT) {
        const int place = i / heads, h = i - place * heads;
        const int tk = pm[place], b0 = (place / ball) * ball;
        float kk[HD], vv[HD], dk[HD], dv[HD];
#pragma unroll
        for (int e = 0; e < HD; ++e) {
            kk[e] = sm[QKV + tk * 3 * c + c + h * HD + e];
            vv[e] = sm[QKV + tk * 3 * c + 2 * c + h * HD + e];
            dk[e] = dv[e] = 0.f;
        }
        for (int p = 0; p < ball; ++p) {
            const int tq = pm[b0 + p];
            const int qo = QKV + tq * 3 * c + h * HD, go = GO + tq * c + h * HD;
            const float m = sm[T.ST + 3 * (tq * heads + h)], inv = sm[T.ST + 3 * (tq * heads + h) + 1];
            const float Dsum = sm[T.ST + 3 * (tq * heads + h) + 2];
            float qv[HD], gv[HD];
            float sv = 0.f, dp = 0.f;
#pragma unroll
            for (int e = 0; e < HD; ++e) {
                qv[e] = sm[qo + e];
                gv[e] = sm[go + e];
                sv = fmaf(qv[e], kk[e], sv);
                dp = fmaf(gv[e], vv[e], dp);
            }
            const float pr = __expf(sv * scale - m) * inv;
            const float ds = pr * (dp - Dsum) * scale;
#pragma unroll
            for (int e = 0; e < HD; ++e) {
                dk[e] = fmaf(ds, qv[e], dk[e]);
                dv[e] = fmaf(pr, gv[e], dv[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < HD; ++e) {
            sm[DQ + tk * 3 * c + c + h * HD + e] = dk[e];
            sm[DQ + tk * 3 * c + 2 * c + h * HD + e] = dv[e];
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void attn_fwd(const BwdDir& T, int QKV, int AO, int n, int c, int heads, int hd, int ball,
                                         int PERM, float scale) {
    switch (hd) {
        case 4: attn_fwd_t<4>(T, QKV, AO, n, c, heads, ball, PERM, scale); break;
        case 8: attn_fwd_t<8>(T, QKV, AO, n, c, heads, ball, PERM, scale); break;
        case 16: attn_fwd_t<16>(T, QKV, AO, n, c, heads, ball, PERM, scale); break;
        default: attn_fwd_t<32>(T, QKV, AO, n, c, heads, ball, PERM, scale); break;
    }
}
__device__ __forceinline__ void attn_bwd(const BwdDir& T, int QKV, int GO, int DQ, int n, int c, int heads, int hd,
                                         int ball, int PERM, float scale) {
    switch (hd) {
        case 4: attn_bwd_t<4>(T, QKV, GO, DQ, n, c, heads, ball, PERM, scale); break;
        case 8: attn_bwd_t<8>(T, QKV, GO, DQ, n, c, heads, ball, PERM, scale); break;
        case 16: attn_bwd_t<16>(T, QKV, GO, DQ, n, c, heads, ball, PERM, scale); break;
        default: attn_bwd_t<32>(T, QKV, GO, DQ, n, c, heads, ball, PERM, scale); break;
    }
}

// one ErwinTransformerBlock, forward recompute + backward; DX (LDS) holds dL/dy on entry, dL/dx on exit
// (round 4: INLINED into its one call site.  As a non-inlined function its four by-reference arguments -- the kernel's own
// argument structs and two descriptors in the weight blobs -- were generic pointers: 172 flat_load instructions re-reading
// their fields where they are used, each waited for with vmcnt(0) AND lgkmcnt(0); inlined, the fields are scalar loads
// from the kernel-argument segment / the blob: 312 flat operations in the kernel -> 6.)
__device__ __forceinline__ void block_bwd(const BwdArgs& a, const BwdDir& T, const haet_erwin_layer& Ly,
                                          const haet_erwin_block_f32& w, float* __restrict__ G, int lvl, int d, int PERM,
                                          const float* __restrict__ tape_in) {
    float* sm = SM();
    const float* Wf = a.Wf;
    const int n = T.N[lvl], c = Ly.c, hid = Ly.hidden, heads = Ly.heads, hd = c / heads, ball = Ly.ball;
    const int DX = T.dX[lvl], XIN = T.XIN, N1 = T.N1, AO = T.AO, XMID = T.XMID, N2 = T.N2, TMP = T.TMP, QKV = T.QKV;
    const int* pm = reinterpret_cast<const int*>(smem) + PERM;
    const float scale = 1.0f / sqrtf((float)hd);
    // the Linear products as 32 x 32 matrix-core tiles (eb_phase) where the level's shape allows: 32 | tokens, 32 | channels
    // and hidden width, every buffer 16-B aligned -- the C3 shape at both levels
    int eb_seq = 0;
    const bool fast = ((n & 31) | (c & 31) | (hid & 31) | ((DX | XIN | N1 | AO | XMID | N2 | TMP | QKV | T.HREG | a.HB) & 3)) == 0;
    // ---------------- forward recompute
    BSTAMP(0);
    for (int i = threadIdx.x; i < n * c; i += kBT) sm[XIN + i] = tape_in[i];
    rel_pos(T.REL, T.P[lvl], PERM, n, d, ball);  // (ends with a barrier)
    rms_fwd(N1, XIN, T.R1, Wf + w.norm1, n, c, T.REL, d, Wf + w.pe_w, Wf + w.pe_b);
    BSTAMP(1);
    if (fast) {
        EbJob j = {};
        j.g_on = 1, j.Y = QKV, j.ldy = 3 * c, j.X = N1, j.ldx = c, j.n = n, j.K = c, j.O = 3 * c, j.nmat = 1;
        j.W = Wf + w.qkv_w + 3 * c * c, j.tnks = c >> 4, j.bias = Wf + w.qkv_b;  // (tiled copy: right behind the [3c][c] tensor)
        EB_RUN(j);
    } else {
        gemm(T.WSTG, QKV, 3 * c, N1, c, Wf + w.qkv_wT, 3 * c, Wf + w.qkv_b, n, c, 3 * c, false);
    }
    __syncthreads();
    BSTAMP(2);
    attn_fwd(T, QKV, AO, n, c, heads, hd, ball, PERM, scale);
    BSTAMP(3);
    // x_mid = x + proj(ao)
    for (int i = threadIdx.x; i < n * c; i += kBT) sm[XMID + i] = sm[XIN + i];
    __syncthreads();
    if (fast) {
        EbJob j = {};
        j.g_on = 1, j.Y = XMID, j.ldy = c, j.X = AO, j.ldx = c, j.n = n, j.K = c, j.O = c, j.nmat = 1, j.acc = 1;
        j.W = Wf + w.proj_w + c * c, j.tnks = c >> 4;
        EB_RUN(j);
    } else {
        gemm(T.WSTG, XMID, c, AO, c, Wf + w.proj_wT, c, nullptr, n, c, c, true);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n * c; i += kBT) sm[XMID + i] += Wf[w.proj_b + i % c];
    __syncthreads();
    rms_fwd(N2, XMID, T.R2, Wf + w.norm2, n, c, 0, 0, nullptr, nullptr);
    BSTAMP(4);
    // ---------------- SwiGLU backward, hidden width in chunks of hc (three n x hc buffers in the H region)
    for (int i = threadIdx.x; i < n * c; i += kBT) sm[TMP + i] = 0.f;  // dn2 accumulates over the chunks
    const int hq = a.HB / n;  // chunk width: a multiple of 32 where possible (matrix-core tiles), else of 4
    int hc = min(hid, hq >= 32 ? (hq & ~31) : max(4, hq & ~3));
    if ((hc & 31) == 0 && (hid & 31) == 0) hc = ((hid / ((hid + hc - 1) / hc) + 31) & ~31);  // equal chunks
    const int H1 = T.HREG, H2 = T.HREG + a.HB, GG = T.HREG + 2 * a.HB;
    wgrad(nullptr, 0, G + w.w3_b, DX, c, 0, 0, n, c, 0);  // db3 = sum_t dy
    for (int j0 = 0; j0 < hid; j0 += hc) {
        const int hw = min(hc, hid - j0);
        if (fast && (hw & 31) == 0) {
            // Four phases per chunk, every 32 x 32 tile of a phase on its own wave (eb_phase):
            //   A  h1 | h2 = n2 W1c^T + b1 | n2 W2c^T + b2
            //   C  dW3c = dy^T g  (g = h2 silu(h1) formed in the operand gather, never stored)   ||   dg = dy W3c
            //   D  dh1, dh2 (element-wise, in place of h1, h2)
            //   E  dW1c = dh1^T n2, dW2c = dh2^T n2, db1c, db2c   ||   dn2 += dh1 W1c + dh2 W2c
            if (j0 == 0) BSTAMP(20);
            {
                EbJob j = {};
                j.g_on = 1, j.Y = H1, j.Ystep = H2 - H1, j.ldy = hw, j.X = N2, j.ldx = c, j.n = n, j.K = c, j.O = hw, j.nmat = 2;
                j.W = Wf + w.w1_w + hid * c, j.Wb = Wf + w.w2_w + hid * c, j.tnks = c >> 4, j.tot0 = j0 >> 5;  // rows j0 .. of [hid][c]
                j.bias = Wf + w.w1_b + j0, j.biasb = Wf + w.w2_b + j0;
#ifndef HAET_STAMP_PHASE_C
                j.stamp = j0 == 0;
#endif
                EB_RUN(j);
            }
            if (j0 == 0) BSTAMP(21);
            if (j0 == 0) BSTAMP(22);
            {
                EbJob j = {};
                j.w_on = 1, j.dW = G + w.w3_w + j0, j.ldw = hid, j.A = DX, j.lda = c, j.B = H2, j.B2 = H1, j.ldb = hw;
                j.wO = c, j.wK = hw, j.wmat = 1, j.gmode = 1;
                j.g_on = 1, j.Y = GG, j.ldy = hw, j.X = DX, j.ldx = c, j.n = n, j.K = c, j.O = hw, j.nmat = 1;
                j.W = Wf + w.w3_wT + hid * c, j.tnks = c >> 4, j.tot0 = j0 >> 5;  // rows j0 .. of w3^T [hid][c]
#ifdef HAET_STAMP_PHASE_C
                j.stamp = j0 == 0;
#endif
                EB_RUN(j);
            }
            __syncthreads();
            if (j0 == 0) BSTAMP(23);
            if (j0 == 0) BSTAMP(24);
            for (int i = threadIdx.x; i < n * hw; i += kBT) {
                const float h1 = sm[H1 + i], h2 = sm[H2 + i], dg = sm[GG + i];
                const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * h1));
                sm[H2 + i] = dg * (h1 * sg);                            // dh2
                sm[H1 + i] = dg * h2 * (sg * (1.0f + h1 * (1.0f - sg)));  // dh1
            }
            if (j0 == 0) BSTAMP(25);
            {
                EbJob j = {};
                j.w_on = 1, j.dW = G + w.w1_w + (size_t)j0 * c, j.dWb = G + w.w2_w + (size_t)j0 * c, j.ldw = c;
                j.A = H1, j.Astep = H2 - H1, j.lda = hw, j.B = N2, j.ldb = c, j.wO = hw, j.wK = c, j.wmat = 2;
                j.db = G + w.w1_b + j0, j.dbb = G + w.w2_b + j0;
                j.g_on = 1, j.Y = TMP, j.ldy = c, j.X = H1, j.X2 = H2, j.ldx = hw, j.n = n, j.K = hw, j.K2 = hw, j.O = c, j.nmat = 1;
                j.acc = 1, j.W = Wf + w.w1_wT + c * hid, j.tnks = hid >> 4, j.ts0 = j0 >> 4;  // columns j0 .. of w1^T [c][hid]
                j.W2 = Wf + w.w2_wT + c * hid, j.tnks2 = hid >> 4, j.ts02 = j0 >> 4;
                EB_RUN(j);
            }
            if (j0 == 0) BSTAMP(26);
            if (j0 == 0) BSTAMP(27);
            continue;
        }
        __syncthreads();
        if (j0 == 0) BSTAMP(20);
        gemm(T.WSTG, H1, hw, N2, c, Wf + w.w1_wT + j0, hid, Wf + w.w1_b + j0, n, c, hw, false);
        gemm(T.WSTG, H2, hw, N2, c, Wf + w.w2_wT + j0, hid, Wf + w.w2_b + j0, n, c, hw, false);
        __syncthreads();
        if (j0 == 0) BSTAMP(21);
        for (int i = threadIdx.x; i < n * hw; i += kBT) {  // g = h2 silu(h1)
            const float h1 = sm[H1 + i];
            sm[GG + i] = sm[H2 + i] * (h1 / (1.0f + __expf(-h1)));
        }
        __syncthreads();
        if (j0 == 0) BSTAMP(22);
        wgrad(G + w.w3_w + j0, hid, nullptr, DX, c, GG, hw, n, c, hw);  // dW3[o][j0 + j] = sum_t dy[t][o] g[t][j]
        __syncthreads();
        if (j0 == 0) BSTAMP(23);
        gemm(T.WSTG, GG, hw, DX, c, Wf + w.w3_w + j0, hid, nullptr, n, c, hw, false);  // dg[t][j] = sum_o dy[t][o] W3[o][j0 + j]
        __syncthreads();
        if (j0 == 0) BSTAMP(24);
        for (int i = threadIdx.x; i < n * hw; i += kBT) {
            const float h1 = sm[H1 + i], h2 = sm[H2 + i], dg = sm[GG + i];
            const float sg = 1.0f / (1.0f + __expf(-h1));
            sm[H2 + i] = dg * (h1 * sg);                            // dh2
            sm[H1 + i] = dg * h2 * (sg * (1.0f + h1 * (1.0f - sg)));  // dh1
        }
        __syncthreads();
        if (j0 == 0) BSTAMP(25);
        wgrad(G + w.w1_w + (size_t)j0 * c, c, G + w.w1_b + j0, H1, hw, N2, c, n, hw, c);
        wgrad(G + w.w2_w + (size_t)j0 * c, c, G + w.w2_b + j0, H2, hw, N2, c, n, hw, c);
        if (j0 == 0) BSTAMP(26);
        gemm(T.WSTG, TMP, c, H1, hw, Wf + w.w1_w + (size_t)j0 * c, c, nullptr, n, hw, c, true);  // dn2 += dh1 W1[j0.., :]
        __syncthreads();
        gemm(T.WSTG, TMP, c, H2, hw, Wf + w.w2_w + (size_t)j0 * c, c, nullptr, n, hw, c, true);
        if (j0 == 0) BSTAMP(27);
    }
    __syncthreads();
    BSTAMP(5);
    rms_bwd(DX, TMP, XMID, T.R2, Wf + w.norm2, G + w.norm2, n, c, T.R2 + n, T.HREG);  // DX = dL/dx_mid now
    BSTAMP(6);
    // ---------------- attention branch: x_mid = x + proj(ao) + b
    if (fast) {  // dWp = dx_mid^T ao, dbp   ||   d ao = dx_mid Wp
        EbJob j = {};
        j.w_on = 1, j.dW = G + w.proj_w, j.ldw = c, j.A = DX, j.lda = c, j.B = AO, j.ldb = c, j.wO = c, j.wK = c, j.wmat = 1;
        j.db = G + w.proj_b;
        j.g_on = 1, j.Y = TMP, j.ldy = c, j.X = DX, j.ldx = c, j.n = n, j.K = c, j.O = c, j.nmat = 1, j.W = Wf + w.proj_wT + c * c, j.tnks = c >> 4;
        EB_RUN(j);
    } else {
        wgrad(G + w.proj_w, c, G + w.proj_b, DX, c, AO, c, n, c, c);
        gemm(T.WSTG, TMP, c, DX, c, Wf + w.proj_w, c, nullptr, n, c, c, false);  // d ao
    }
    __syncthreads();
    const int DQ = T.HREG;  // dqkv [n][3c] lives in the (now free) H region
    BSTAMP(7);
    attn_bwd(T, QKV, TMP, DQ, n, c, heads, hd, ball, PERM, scale);
    BSTAMP(8);
    // qkv Linear, PE, RMSNorm 1
    if (fast) {  // dWqkv = dqkv^T n1, dbqkv   ||   dn1 = dqkv Wqkv
        EbJob j = {};
        j.w_on = 1, j.dW = G + w.qkv_w, j.ldw = c, j.A = DQ, j.lda = 3 * c, j.B = N1, j.ldb = c, j.wO = 3 * c, j.wK = c, j.wmat = 1;
        j.db = G + w.qkv_b;
        j.g_on = 1, j.Y = TMP, j.ldy = c, j.X = DQ, j.ldx = 3 * c, j.n = n, j.K = 3 * c, j.O = c, j.nmat = 1;
        j.W = Wf + w.qkv_wT + 3 * c * c, j.tnks = (3 * c) >> 4;
        EB_RUN(j);
    } else {
        wgrad(G + w.qkv_w, c, G + w.qkv_b, DQ, 3 * c, N1, c, n, 3 * c, c);
        gemm(T.WSTG, TMP, c, DQ, 3 * c, Wf + w.qkv_w, c, nullptr, n, 3 * c, c, false);  // dn1
    }
    __syncthreads();
    wgrad(G + w.pe_w, d, G + w.pe_b, TMP, c, T.REL, d, n, c, d);  // d pe_w[k][j] = sum_t dn1[t][k] rel[t][j]
    BSTAMP(9);
    rms_bwd(DX, TMP, XIN, T.R1, Wf + w.norm1, G + w.norm1, n, c, T.R1 + n, T.HREG);  // DX = dL/dx
    BSTAMP(10);
}

// BatchNorm backward on DZ [n][c] (in place -> dL/dy) with y (pre-normalisation) in Y; d_gamma / d_beta to G.
// train: mu / invstd are the forward's batch statistics and the two means over ALL clouds' tokens come from every
// cloud's partial d_gamma / d_beta after a grid barrier; eval: running statistics, no coupling.
__device__ __forceinline__ void bn_bwd(const BwdArgs& a, const BwdDir& T, int DZ, int ldz, int Y, int n, int c,
                                       const haet_erwin_resample_f32& rw, float eps, bool train, int slot, int cloud,
                                       int n_clouds, int& epoch, int total) {
    float* sm = SM();
    const float* Wf = a.Wf;
    float* G = a.grads + (size_t)cloud * total;
    const int MU = T.ST, IS = T.ST + c;  // per-channel mean / invstd scratch (ST is free between blocks)
    for (int k = threadIdx.x; k < c; k += kBT) {
        const float mu = train ? a.bn_stats[(size_t)slot * 2 * a.cmax + k] : Wf[rw.rmean + k];
        const float is = train ? a.bn_stats[(size_t)slot * 2 * a.cmax + a.cmax + k] : 1.0f / sqrtf(Wf[rw.rvar + k] + eps);
        sm[MU + k] = mu;
        sm[IS + k] = is;
        float dg = 0.f, db = 0.f;
        for (int t = 0; t < n; ++t) {
            const float dz = sm[DZ + t * ldz + k];
            dg = fmaf(dz, (sm[Y + t * c + k] - mu) * is, dg);
            db += dz;
        }
        G[rw.gamma + k] = dg;
        G[rw.beta + k] = db;
    }
    if (train) {
        bwd_grid_barrier(a.bar, (unsigned)n_clouds * (unsigned)(epoch + 1), a.bar + 1);
        ++epoch;
        const float cnt = (float)n * (float)n_clouds;
        for (int k = threadIdx.x; k < c; k += kBT) {
            float sg = 0.f, sb = 0.f;
            for (int q = 0; q < n_clouds; ++q) {
                sg += a.grads[(size_t)q * total + rw.gamma + k];
                sb += a.grads[(size_t)q * total + rw.beta + k];
            }
            sm[T.ST + 2 * c + k] = sg / cnt;  // mean(dz xhat)
            sm[T.ST + 3 * c + k] = sb / cnt;  // mean(dz)
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n * c; i += kBT) {
        const int t = i / c, k = i - t * c;
        const float gi = Wf[rw.gamma + k] * sm[IS + k];
        float dz = sm[DZ + t * ldz + k];
        if (train) dz = dz - sm[T.ST + 3 * c + k] - (sm[Y + i] - sm[MU + k]) * sm[IS + k] * sm[T.ST + 2 * c + k];
        sm[DZ + t * ldz + k] = dz * gi;
    }
    __syncthreads();
}

__global__ __launch_bounds__(kBT) void erwin_bwd_kernel(BwdArgs a, BwdDir T) {
    float* sm = SM();
    int* smi = reinterpret_cast<int*>(smem);
    const haet_erwin_desc& D = *reinterpret_cast<const haet_erwin_desc*>(a.Wfwd);
    const haet_erwin_f32_layout& Lw = *reinterpret_cast<const haet_erwin_f32_layout*>(a.Wf);
    const int cloud = blockIdx.x, n_clouds = gridDim.x, d = D.d, nl = D.n_levels, Mp = a.Mp, c0 = D.enc[0].c;
    const int total = Lw.total;
    float* G = a.grads + (size_t)cloud * total;
    const bool train = D.bn_train != 0;
    // The fp32 blob (1.6 MB at C3) has been evicted from this XCD's L2 by the streaming kernels since the last step, and
    // every Linear below walks its weights as a chain of dependent loads: from HBM each link is a ~2 us miss (1.1 of the
    // kernel's 1.6 ms).  One streaming read of the whole blob first -- 16 independent 16-B loads per thread and round --
    // takes ~40 us and leaves the chains on L2 hits.
    {
        typedef __attribute__((ext_vector_type(4))) unsigned u4;
        const u4* wb = reinterpret_cast<const u4*>(a.Wf);
        const unsigned pieces = (unsigned)total / 4u;
        unsigned keep = 0;
        for (unsigned base = 0; base < pieces; base += kBT * 16) {
            u4 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = wb[min(base + (unsigned)j * kBT + threadIdx.x, pieces - 1)];
#pragma unroll
            for (int j = 0; j < 16; ++j) keep ^= v[j][0];
        }
        asm volatile("" ::"v"(keep));
    }
    // positions per level (tree order at level 0, centres above), index maps, dL/d(final level-0 tokens)
    for (int i = threadIdx.x; i < Mp * d; i += kBT)
        sm[T.P[0] + i] = a.pos[(size_t)a.tree_idx[(size_t)cloud * Mp + i / d] * d + i % d];
    for (int i = threadIdx.x; i < Mp; i += kBT) smi[T.IDENT + i] = i;
    if (D.rotate) {
        size_t off = 0;
        for (int l = 0; l < nl; ++l) {
            for (int i = threadIdx.x; i < T.N[l]; i += kBT) smi[T.PERM[l] + i] = a.rot[off + (size_t)cloud * T.N[l] + i] - cloud * T.N[l];
            off += (size_t)n_clouds * T.N[l];
        }
    }
    for (int i = threadIdx.x; i < Mp * c0; i += kBT) {
        const size_t g = (size_t)cloud * Mp + i / c0;
        sm[T.dX[0] + i] = a.tree_mask[g] ? a.g_out[(size_t)a.tree_idx[g] * c0 + i % c0] : 0.f;  // erwin_flash.py:193
    }
    __syncthreads();
    for (int l = 0; l + 1 < nl; ++l) {  // ball centres (pooling.py:29-31)
        const int st = D.pool[l].stride;
        for (int i = threadIdx.x; i < T.N[l + 1] * d; i += kBT) {
            const int p = i / d, j = i - p * d;
            float s = 0.f;
            for (int q = 0; q < st; ++q) s += sm[T.P[l] + (p * st + q) * d + j];
            sm[T.P[l + 1] + i] = st == 1 ? sm[T.P[l] + i] : s / (float)st;
        }
        __syncthreads();
    }
    // the op list of erwin_kernel, in reverse: op -> (kind, level, block | decoder flag)
    auto op_at = [&](int want, int& kind, int& la, int& b) {
        int i = 0;
        for (int l = 0; l < nl; ++l) {
            for (int j = 0; j < D.enc[l].depth; ++j, ++i)
                if (i == want) { kind = 0; la = l; b = j; }
            if (l < nl - 1) { if (i == want) { kind = 1; la = l; b = 0; } ++i; }
        }
        for (int l = nl - 2; l >= 0; --l) {
            if (i == want) { kind = 2; la = l; b = 0; }
            ++i;
            for (int j = 0; j < D.dec[l].depth; ++j, ++i)
                if (i == want) { kind = 0; la = l; b = j | 256; }
        }
        return i;
    };
    int epoch = 0;
#ifdef HAET_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) haet_bwd_stamps[40] = __builtin_amdgcn_s_memtime();
#endif
    for (int op = a.n_ops - 1; op >= 0; --op) {
#ifdef HAET_STAMPS
        if (threadIdx.x == 0) haet_bwd_stamp_on = (op == a.n_ops - 1);
        if (blockIdx.x == 0 && threadIdx.x == 0) haet_bwd_stamps[48 - (a.n_ops - 1 - op)] = __builtin_amdgcn_s_memtime();
#endif
        int kind = 0, l = 0, b = 0;
        op_at(op, kind, l, b);
        const float* tp = a.tape + ((size_t)cloud * a.n_ops + op) * a.tape_op_floats;
        if (kind == 0) {
            const bool dec = b & 256;
            const haet_erwin_layer& Ly = dec ? D.dec[l] : D.enc[l];
            const int j = b & 255;
            const bool rt = D.rotate && Ly.depth > 1 && (j & 1);
            block_bwd(a, T, Ly, dec ? Lw.dec[l][j] : Lw.enc[l][j], G, l, d, rt ? T.PERM[l] : T.IDENT, tp);
        } else if (kind == 2) {
            // ---- BallUnpooling (pooling.py:58-72), level l+1 -> l: z = BN(children.x + split(proj([x, rel_children])))
            const haet_erwin_resample_f32& rw = Lw.unpool[l];
            const int st = D.unpool[l].stride, n2 = T.N[l + 1], c2 = D.enc[l + 1].c, c = D.enc[l].c, n = T.N[l];
            const int K = c2 + st * d, ROWS = T.HREG, U = T.QKV;  // rows [n2][K], u [n2][st*c]
            // children.x = the encoder's output at level l = the taped input of the pool op of level l
            int pk = 0, pl = 0, pb = 0, pool_op = 0;
            for (int o2 = 0; o2 < a.n_ops; ++o2) {
                op_at(o2, pk, pl, pb);
                if (pk == 1 && pl == l) pool_op = o2;
            }
            const float* cx = a.tape + ((size_t)cloud * a.n_ops + pool_op) * a.tape_op_floats;
            for (int i = threadIdx.x; i < n2 * K; i += kBT) {
                const int p = i / K, k = i - p * K;
                float v;
                if (k < c2) v = tp[p * c2 + k];
                else {
                    const int r = k - c2, q = r / d, jj = r - q * d;
                    v = sm[T.P[l] + (p * st + q) * d + jj] - sm[T.P[l + 1] + p * d + jj];
                }
                sm[ROWS + i] = v;
            }
            __syncthreads();
            gemm(T.WSTG, U, st * c, ROWS, K, a.Wf + rw.wT, st * c, a.Wf + rw.b, n2, K, st * c, false);
            __syncthreads();
            // s (pre-BatchNorm) [n][c] = children.x + 'n (m d) -> (n m) d' of u
            for (int i = threadIdx.x; i < n * c; i += kBT) {
                const int t = i / c, k = i - t * c;
                sm[T.N1 + i] = cx[i] + sm[U + (t / st) * (st * c) + (t % st) * c + k];
            }
            __syncthreads();
            bn_bwd(a, T, T.dX[l], c, T.N1, n, c, rw, D.unpool[l].bn_eps, train, 2 * l + 1, cloud, n_clouds, epoch, total);
            // ds -> skip (children.x) and du = ds re-arranged [n2][st*c]
            for (int i = threadIdx.x; i < n * c; i += kBT) {
                const int t = i / c, k = i - t * c;
                const float v = sm[T.dX[l] + i];
                sm[T.SK[l] + i] = v;
                sm[U + (t / st) * (st * c) + (t % st) * c + k] = v;
            }
            __syncthreads();
            wgrad(G + rw.w, K, G + rw.b, U, st * c, ROWS, K, n2, st * c, K);
            // dx (level l+1) = du W[:, :c2]
            gemm(T.WSTG, T.dX[l + 1], c2, U, st * c, a.Wf + rw.w, K, nullptr, n2, st * c, c2, false);
            __syncthreads();
        } else {
            // ---- BallPooling (pooling.py:25-40), level l -> l+1: z = BN(proj([x_(n s) c -> n (s c), rel]))
            const haet_erwin_resample_f32& rw = Lw.pool[l];
            const int st = D.pool[l].stride, n2 = T.N[l + 1], c2 = D.enc[l + 1].c, c = D.enc[l].c, n = T.N[l];
            if (st == 1) {  // "no pooling": the node passes through (pooling.py:26-27)
                for (int i = threadIdx.x; i < n * c; i += kBT) sm[T.dX[l] + i] = sm[T.dX[l + 1] + i] + sm[T.SK[l] + i];
                __syncthreads();
                continue;
            }
            const int K = st * c + st * d, ROWS = T.HREG, Yb = T.N1;
            for (int i = threadIdx.x; i < n2 * K; i += kBT) {
                const int p = i / K, k = i - p * K;
                float v;
                if (k < st * c) v = tp[(p * st + k / c) * c + k % c];
                else {
                    const int r = k - st * c, q = r / d, jj = r - q * d;
                    v = sm[T.P[l] + (p * st + q) * d + jj] - sm[T.P[l + 1] + p * d + jj];
                }
                sm[ROWS + i] = v;
            }
            __syncthreads();
            gemm(T.WSTG, Yb, c2, ROWS, K, a.Wf + rw.wT, c2, a.Wf + rw.b, n2, K, c2, false);
            __syncthreads();
            bn_bwd(a, T, T.dX[l + 1], c2, Yb, n2, c2, rw, D.pool[l].bn_eps, train, 2 * l, cloud, n_clouds, epoch, total);
            wgrad(G + rw.w, K, G + rw.b, T.dX[l + 1], c2, ROWS, K, n2, c2, K);
            // d rows[:, :st*c] -> dL/dx (level l), plus the skip gradient of the unpool
            gemm(T.WSTG, T.TMP, st * c, T.dX[l + 1], c2, a.Wf + rw.w, K, nullptr, n2, c2, st * c, false);
            __syncthreads();
            for (int i = threadIdx.x; i < n * c; i += kBT) {
                const int t = i / c, k = i - t * c;
                sm[T.dX[l] + i] = sm[T.TMP + (t / st) * (st * c) + (t % st) * c + k] + sm[T.SK[l] + i];
            }
            __syncthreads();
        }
    }
    // x = feat[tree_idx] (erwin_flash.py:176-181): scatter-ADD over duplicated (padding) slots; one thread per token channel
    for (int i = threadIdx.x; i < a.M * c0; i += kBT) {
        const int tok = i / c0, k = i - tok * c0;
        float s = 0.f;
        for (int slot = 0; slot < Mp; ++slot)
            if (a.tree_idx[(size_t)cloud * Mp + slot] == cloud * a.M + tok) s += sm[T.dX[0] + slot * c0 + k];
        a.g_in[((size_t)cloud * a.M + tok) * c0 + k] = s;
    }
}

// LDS carve-up; returns false when one cloud does not fit
static bool bwd_directory(const haet_erwin_desc& D, int Mp, BwdDir* T, int* HB_out, int* cmax_out) {
    int n = Mp, p = 0, ncmax = 0, nmax = Mp, cmax = 4, hmax = 1;
    const int d = D.d;
    for (int l = 0; l < D.n_levels; ++l) {
        T->N[l] = n;
        const int c = D.enc[l].c;
        ncmax = ncmax > n * c ? ncmax : n * c;
        cmax = cmax > c ? cmax : c;
        int heads = D.enc[l].heads;
        if (l < D.n_levels - 1 && D.dec[l].heads > heads) heads = D.dec[l].heads;
        hmax = hmax > n * heads ? hmax : n * heads;
        if (l < D.n_levels - 1) n /= D.pool[l].stride;
    }
    for (int l = 0; l < D.n_levels; ++l) {
        T->dX[l] = p;
        p += T->N[l] * D.enc[l].c;
        T->SK[l] = p;
        p += l < D.n_levels - 1 ? T->N[l] * D.enc[l].c : 0;
        T->P[l] = p;
        p += (T->N[l] * d + 3) & ~3;
        T->PERM[l] = p;
        p += T->N[l];
    }
    T->IDENT = p; p += Mp;
    T->XIN = p; p += ncmax;
    T->N1 = p; p += ncmax;
    T->AO = p; p += ncmax;
    T->XMID = p; p += ncmax;
    T->N2 = p; p += ncmax;
    T->TMP = p; p += ncmax;
    T->QKV = p; p += 3 * ncmax;
    int HB = 3072;
    // the H region also holds dqkv [n][3c] and the pooling rows [n2][K]
    int need = 3 * ncmax;
    for (int l = 0; l + 1 < D.n_levels; ++l) {
        const int st = D.pool[l].stride, n2 = T->N[l + 1];
        const int k1 = n2 * (st * D.enc[l].c + st * d), k2 = n2 * (D.enc[l + 1].c + st * d);
        need = need > k1 ? need : k1;
        need = need > k2 ? need : k2;
    }
    while (3 * HB < need) HB += 256;
    int st_floats = 3 * hmax;
    st_floats = st_floats > 4 * cmax ? st_floats : 4 * cmax;
    // Where every level has the matrix-core shape (block_bwd's `fast`), only the resampling Linears still stage their
    // weights through LDS: half the staging tile goes to the H region, and the H region then grows as far as 160 KB
    // allows -- wider SwiGLU chunks are fewer, fatter eb_phase phases (C3: 64 / 128 hidden columns per chunk, not 32 / 96)
    int wstage = kWStage;
    bool mc = true;
    for (int l = 0; l < D.n_levels; ++l) {
        mc = mc && ((T->N[l] | D.enc[l].c | D.enc[l].hidden) & 31) == 0;
        if (l < D.n_levels - 1 && D.dec[l].depth > 0) mc = mc && ((D.dec[l].c | D.dec[l].hidden) & 31) == 0;
    }
    if (mc) wstage = kWStage / 2;
    const int fixed = p + 4 * nmax + st_floats + ((nmax * d + 3) & ~3) + wstage;
    int hb_cap = 0;  // no use beyond the widest n x hidden
    for (int l = 0; l < D.n_levels; ++l) {
        int hid = D.enc[l].hidden;
        if (l < D.n_levels - 1 && D.dec[l].depth > 0 && D.dec[l].hidden > hid) hid = D.dec[l].hidden;
        hb_cap = hb_cap > T->N[l] * hid ? hb_cap : T->N[l] * hid;
    }
    while (mc && HB + 256 <= hb_cap && (size_t)(fixed + 3 * (HB + 256)) * 4 <= 160 * 1024) HB += 256;
    T->HREG = p; p += 3 * HB;
    T->R1 = p; p += 2 * nmax;
    T->R2 = p; p += 2 * nmax;
    T->ST = p; p += st_floats;
    T->REL = p; p += (nmax * d + 3) & ~3;
    // a cloud that misses 160 KB by less than the staging tile keeps a smaller tile (gemm() takes any size >= one weight
    // column: fewer output columns per staged tile) -- 128 tokens x 16 channels, the reference's elasticity
    // configuration (n_hidden 128 / 8 heads, slice_num 128), misses by 1 KB with the full tile
    const int avail = 160 * 1024 / 4 - p;
    if (avail < wstage && avail >= 1024) wstage = avail & ~255;
    T->WST = p; p += wstage;
    T->WSTG = T->WST | wstage << 16;
    T->total = p;
    *HB_out = HB;
    *cmax_out = cmax;
    return (size_t)p * 4 <= 160 * 1024;
}

static int erwin_ops(const haet_erwin_desc& D) {
    int i = 0;
    for (int l = 0; l < D.n_levels; ++l) {
        i += D.enc[l].depth;
        if (l < D.n_levels - 1) i += 2 + D.dec[l].depth;
    }
    return i;
}

}  // namespace haet

using namespace haet;

#ifdef HAET_STAMPS
extern "C" int haet_debug_read_bwd_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(haet_bwd_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

extern "C" size_t haet_erwin_tape_floats(const haet_erwin_desc* desc, int M, int* n_ops, int* op_floats) {
    if (!desc || desc->n_levels < 1 || desc->n_levels > HAET_ERWIN_MAX_LEVELS) return 0;
    int depth = 0;
    while ((2 << depth) < M) ++depth;
    int n = 2 << depth, mx = 0;
    for (int l = 0; l < desc->n_levels; ++l) {
        mx = mx > n * desc->enc[l].c ? mx : n * desc->enc[l].c;
        if (l < desc->n_levels - 1 && desc->pool[l].stride > 0) n /= desc->pool[l].stride;
    }
    const int ops = erwin_ops(*desc);
    if (n_ops) *n_ops = ops;
    if (op_floats) *op_floats = mx;
    return (size_t)ops * mx;
}

extern "C" int haet_erwin_bwd_supported(const haet_erwin_desc* desc, int n_clouds, int M) {
    if (!desc || n_clouds < 1 || M < 2 || desc->n_levels < 1 || desc->n_levels > HAET_ERWIN_MAX_LEVELS) return 0;
    if (desc->bn_train && n_clouds > resident_workgroup_limit()) return 0;
    int depth = 0;
    while ((2 << depth) < M) ++depth;
    for (int l = 0; l < desc->n_levels; ++l) {  // head dims with a register-resident attention instantiation
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && l == desc->n_levels - 1) break;
            const haet_erwin_layer& L = pass ? desc->dec[l] : desc->enc[l];
            if (L.heads < 1 || L.c % L.heads) return 0;
            const int hd = L.c / L.heads;
            if (hd != 4 && hd != 8 && hd != 16 && hd != 32) return 0;
        }
    }
    BwdDir T;
    int HB, cmax;
    return bwd_directory(*desc, 2 << depth, &T, &HB, &cmax) ? 1 : 0;
}

extern "C" int haet_erwin_bwd(const float* g_out, const float* tape, const float* pos, const int32_t* tree_idx,
                               const uint8_t* tree_mask, const int32_t* rot, const float* weights_f32,
                               const float* weights_fwd, const haet_erwin_desc* desc, const float* bn_batch_stats,
                               void* workspace, int n_clouds, int M, float* g_in, float* grads, haet_stream_t stream) {
    if (!g_out || !tape || !pos || !tree_idx || !tree_mask || !weights_f32 || !weights_fwd || !desc || !g_in || !grads)
        return fail(HAET_E_ARG, "erwin_bwd: null pointer");
    if (!haet_erwin_bwd_supported(desc, n_clouds, M))
        return fail(HAET_E_RESOURCE, "erwin_bwd: one cloud of %d tokens does not fit the backward kernel's LDS plan "
                                     "(or more clouds than compute units in train mode)", M);
    if (desc->rotate && !rot) return fail(HAET_E_ARG, "erwin_bwd: rotate set but rot is NULL");
    if (desc->bn_train && (!bn_batch_stats || !workspace))
        return fail(HAET_E_ARG, "erwin_bwd: train-mode BatchNorm needs the forward's batch statistics and a workspace");
    int depth = 0;
    while ((2 << depth) < M) ++depth;
    BwdArgs a = {};
    BwdDir T;
    bwd_directory(*desc, 2 << depth, &T, &a.HB, &a.cmax);
    a.g_out = g_out, a.tape = tape, a.pos = pos, a.tree_idx = tree_idx, a.tree_mask = tree_mask, a.rot = rot;
    a.Wf = weights_f32, a.Wfwd = weights_fwd;
    a.bn_stats = bn_batch_stats;
    a.bar = reinterpret_cast<unsigned*>(workspace);
    a.g_in = g_in, a.grads = grads, a.M = M, a.Mp = 2 << depth;
    haet_erwin_tape_floats(desc, M, &a.n_ops, &a.tape_op_floats);
    const size_t bytes = (size_t)T.total * sizeof(float);
    static std::atomic<size_t> attr_set[64];
    std::atomic<size_t>& attr_dev = attr_set[PerDeviceOnce::bit()];
    if (bytes > 48 * 1024 && bytes > attr_dev.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(erwin_bwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return fail(HAET_E_LAUNCH, "erwin_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_dev.store(bytes, std::memory_order_release);
    }
    if (desc->bn_train) {
        hipError_t e = zero_async(workspace, 4, as_stream(stream));  // the counter; the time-out word stays (sticky)
        if (e != hipSuccess) return fail(HAET_E_LAUNCH, "erwin_bwd: zeroing the barrier counter: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(erwin_bwd_kernel, dim3(n_clouds), dim3(kBT), bytes, as_stream(stream), a, T);
    return check_launch("erwin_bwd_kernel");
}
