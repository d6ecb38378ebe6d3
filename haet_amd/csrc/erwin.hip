// Fused Erwin stage: the whole hierarchical ball transformer over the slice tokens of
// one (batch, head) cloud runs in ONE workgroup with the token state resident in LDS.
// The reference runs ~100 small kernels (+ a host round trip for the tree) here
// (models/components/erwinflash/erwin_flash.py:138-195); at T <= ~1k tokens that is pure
// launch latency, and it is the fixed cost that bounds multi-GPU scaling.
//
// LDS: per level l the tokens X_l [n_l][c_l+4] fp32 and positions P_l [n_l][d] in TREE
// order; `ia` = split-bf16 (hi|lo) image of the current Linear input of width c; `wide` =
// qkv (fp32) / SwiGLU hidden / pooling rows (split-bf16 images); index maps.  Rotated blocks
// (layers.py:81-89) never move data: a position map `perm` says which token sits at which
// place of the ball sequence; everything except the per-ball attention and the ball centre
// is per-token and order-free.
//
// Every Linear runs on the matrix cores as split-bf16 (v_mfma_f32_16x16x32_bf16 x3: hi*hi +
// hi*lo + lo*hi, fp32 accumulate): producers write their output already split into bf16
// hi/lo LDS images, weights are pre-split on the host ([out][in~32] bf16 planes), so a
// 16-token x 16-output tile step is 2 ds_read_b128 + 2 global 16-B loads + 3 MFMAs with no
// conversion work; the weight loads of the next two (tile, k-step) items are in flight while
// the current one is multiplied.
#include "common.h"
#include "balltree_wave_dev.h"

namespace haet {

constexpr int kPad = 4;      // fp32 row padding (floats): rows stay 16-B aligned, bank strides break
constexpr int kImgPad = 16;  // bytes added to every bf16 image row

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#ifdef HAET_STAMPS
// Diagnostic build only (make EXTRA=-DHAET_STAMPS; tools/erwin_stamps.py): s_memtime at phase
// boundaries of cloud 0, written to a buffer nothing else reads.  Never quote this build's run time.
__device__ unsigned long long haet_stamps[128];
#define STAMP(i)                                                                        \
    do {                                                                                \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                      \
            haet_stamps[i] = __builtin_amdgcn_s_memtime();                              \
            haet_stamps[64 + (i)] = __builtin_amdgcn_s_memrealtime();                   \
        }                                                                               \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ unsigned short bf16_rne(float a) {
    f32x2 v = {a, 0.f};
    return (unsigned short)__builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// ALL LDS addressing is by BYTE OFFSET into this one array.  Generic pointers to LDS (pointers kept
// in structs / arrays / passed to non-inlined routines) make hipcc emit flat_load/flat_store, and a
// flat access must wait vmcnt(0) AND lgkmcnt(0): the first version of this kernel had 1038 flat
// and 4 ds instructions and every phase ran at one memory latency per access.
// (declared in balltree_wave_dev.h as well: one dynamic LDS array for the whole kernel)
extern __shared__ __attribute__((aligned(16))) char smem[];
// Arguments of a non-inlined routine arrive in VGPRs, i.e. "divergent" to the compiler: every
// `if (mode == ...)` becomes an exec-mask dance and every address a vector computation.  The
// wave-uniform ones are re-declared uniform with v_readfirstlane at routine entry.
__device__ __forceinline__ int U(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename P>
__device__ __forceinline__ P* UP(P* p) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return reinterpret_cast<P*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float& F(int off) { return *reinterpret_cast<float*>(smem + off); }
__device__ __forceinline__ int& I(int off) { return *reinterpret_cast<int*>(smem + off); }

// split-bf16 image: hi plane, then lo plane `plane` bytes later; rows `S` bytes apart
// The launch's block size as a constant: `blockDim.x` is a load from the dispatch packet -- a
// dependent memory round trip at the entry of every routine that uses it.
constexpr int kErwinThreads = 512;
// The phase routines of a block are inlined into erwin_block (one copy: erwin_block itself is not
// inlined): as calls each cost ~1.5k cycles -- the callee's entry waits for every outstanding memory
// operation, the arguments are re-made uniform, and the caller reloads what it had to park in scratch
// -- on phases of 3-8k cycles (kernel 392k -> 357k cycles).  -DHAET_ERWIN_NOINLINE restores the calls.
#ifdef HAET_ERWIN_NOINLINE
#define ERWIN_PHASE __attribute__((noinline))
#else
#define ERWIN_PHASE __forceinline__
#endif

struct Img {
    int base;  // byte offset in smem
    int S, plane;
    __device__ __forceinline__ void put(int t, int k, float v) const {
        const unsigned short h = bf16_rne(v);
        const unsigned short l = bf16_rne(v - __uint_as_float((unsigned)h << 16));
        *reinterpret_cast<unsigned short*>(smem + base + t * S + 2 * k) = h;
        *reinterpret_cast<unsigned short*>(smem + base + plane + t * S + 2 * k) = l;
    }
    // four consecutive columns k0..k0+3 of row t (k0 % 4 == 0, 8-B aligned rows): 8 B per plane
    __device__ __forceinline__ void put4(int t, int k0, const float (&v)[4]) const {
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
        typedef __attribute__((ext_vector_type(2))) float f2;
        typedef __attribute__((ext_vector_type(2))) __bf16 b2;
        unsigned h[2], l[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f2 a = {v[2 * i], v[2 * i + 1]};
            h[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(a, b2));
            const f2 r = {v[2 * i] - __uint_as_float(h[i] << 16), v[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u)};
            l[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, b2));
        }
        const u32x2_t hv = {h[0], h[1]}, lv = {l[0], l[1]};
        *reinterpret_cast<u32x2_t*>(smem + base + t * S + 2 * k0) = hv;
        *reinterpret_cast<u32x2_t*>(smem + base + plane + t * S + 2 * k0) = lv;
    }
};

__device__ __forceinline__ Img UI(const Img& i) { return Img{U(i.base), U(i.S), U(i.plane)}; }
__host__ __device__ inline int round32(int k) { return (k + 31) & ~31; }
__host__ __device__ inline int img_stride(int K) { return round32(K) * 2 + kImgPad; }

// LDS directory, itself in LDS (ints at smem[0..]): per level l the BYTE offsets of X_l, P_l,
// rot_l and n_l, ld_l (floats); then the shared regions.
constexpr int T_X = 0, T_P = 4, T_ROT = 8, T_N = 12, T_LD = 16;  // + level
constexpr int T_IA = 20, T_WIDE = 21, T_REL = 22, T_PAR = 23, T_IDENT = 24, T_TREE = 25, T_BN = 26, T_WORDS = 32;
__device__ __forceinline__ int T(int i) { return __builtin_amdgcn_readfirstlane(I(4 * i)); }

// copy `len` floats (16-B aligned, multiple of 4) of small parameters into LDS
__device__ __forceinline__ void stage_params(int dst, const float* __restrict__ src, int len) {
    for (int i = threadIdx.x * 4; i < len; i += kErwinThreads * 4)
        *reinterpret_cast<f32x4*>(smem + dst + 4 * i) = *reinterpret_cast<const f32x4*>(src + i);
}

__device__ __forceinline__ f32x4 mfma3(u32x4 ah, u32x4 al, u32x4 bh, u32x4 bl, f32x4 c) {
    const bf16x8 a_h = __builtin_bit_cast(bf16x8, ah), a_l = __builtin_bit_cast(bf16x8, al);
    const bf16x8 b_h = __builtin_bit_cast(bf16x8, bh), b_l = __builtin_bit_cast(bf16x8, bl);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_h, b_l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_l, b_h, c, 0, 0, 0);
    return c;
}

enum EpiMode {
    EP_QKV = 0,     // dst[t][o]  = acc + b0[o]                      (fp32 rows, ld = ldd)
    EP_RESID = 1,   // dst[t][o] += acc + b0[o]
    EP_SWIGLU = 2,  // img[t][o]  = (acc1 + b1[o]) * silu(acc0 + b0[o])   (two matrices)
    EP_BN = 3,      // dst[t][o]  = (acc + b0[o]) * sc[o] + sh[o]
    EP_UNPOOL = 4,  // dst[t*st + o/aux][o%aux] = (old + acc + b0[o]) * sc[o%aux] + sh[o%aux]
};

// Y[t][o] = sum_k X[t][k] W[o][k] on the matrix cores.  X: split image in LDS (n rows, K~32
// columns, pad columns zero).  W: global bf16 planes [O][Kp] (hi at Wa, lo at Wa + O*Kp);
// EP_SWIGLU adds a second matrix Wb on the same activations.  Two (tile, k-step) items of weights
// are in flight behind the one being multiplied.  dst / b0 / b1 / sc / sh are LDS byte offsets.
// One shared routine (runtime epilogue mode) keeps the kernel's instruction footprint small; only
// DUAL (the SwiGLU pair) is a template parameter: as a runtime `if (dual)` around the second
// matrix's loads it was a conditional VMEM operation inside the prefetch ring, and the waitcnt pass
// then waits vmcnt(0) before every use -- one item in flight instead of two, for EVERY linear.
// First two (tile, k-step) weight items of a Linear, fetched AHEAD of the phase that precedes it
// (norm, attention, the previous Linear): with everything inlined in one routine the registers simply
// stay live, and the Linear starts on operands that are already there instead of on an L2 round trip.
struct WPre {
    u32x4 a[4], b[4];
};
template <bool DUAL>
__device__ __forceinline__ WPre linear_prefetch(int K, const unsigned short* __restrict__ Wa,
                                               const unsigned short* __restrict__ Wb, int O) {
    typedef const __attribute__((address_space(1))) u32x4* gv4;
    const int lane = threadIdx.x & 63, wave = U(threadIdx.x >> 6), nwaves = kErwinThreads >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int Kp = round32(K), nks = Kp >> 5, nto = (O + 15) >> 4;
    const size_t plane = (size_t)O * Kp;
    int ot = wave % nto, ks = 0;  // item 0 of this wave (the token tile does not enter the address)
    WPre p;
    // the tile-ordered copy behind the two row-major planes (pack.py, _Blob.planes): piece (ot, ks, plane) = 1 KB, lane * 16 B
    const unsigned short* Ta = Wa + 2 * plane + lane * 8;
    const unsigned short* Tb = DUAL ? Wb + 2 * plane + lane * 8 : Ta;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const size_t off = (size_t)(ot * nks + ks) * 1024;
        u32x4(&wv)[4] = i ? p.b : p.a;
        wv[0] = *(gv4)(Ta + off);
        wv[1] = *(gv4)(Ta + off + 512);
        if (DUAL) {
            wv[2] = *(gv4)(Tb + off);
            wv[3] = *(gv4)(Tb + off + 512);
        }
        if (++ks == nks) {  // same walk as Cur::advance in linear_core_t
            ks = 0;
            ot += nwaves;
            while (ot >= nto) ot -= nto;
        }
    }
    return p;
}

// The epilogue mode is a template parameter too (five instantiations, ~3 KB each): with one
// shared routine the 20+ arguments no longer fit the argument registers, the rest came through
// scratch memory (a load and a wait at every entry), and the epilogue was a chain of mode branches.
// ex0..ex3 are mode-specific: SWIGLU {b1, image base, image row stride, image plane}; BN {sc, sh};
// UNPOOL {sc, sh, aux, st}.
template <int MODE>
__device__ ERWIN_PHASE void linear_core_t(
    const Img x_, int n, int K, const unsigned short* __restrict__ Wa,
    const unsigned short* __restrict__ Wb, int O, int dst, int ldd, int b0, int ex0, int ex1, int ex2,
    int ex3, const WPre* pre = nullptr) {
    constexpr int mode = MODE;
    constexpr bool DUAL = MODE == EP_SWIGLU;
    const Img x = UI(x_);
    n = U(n), K = U(K), O = U(O), dst = U(dst), ldd = U(ldd), b0 = U(b0);
    ex0 = U(ex0), ex1 = U(ex1), ex2 = U(ex2), ex3 = U(ex3), Wa = UP(Wa), Wb = UP(Wb);
    const int b1 = ex0, sc = ex0, sh = ex1, aux = ex2, st = ex3;
    const Img di = {ex1, ex2, ex3};
    const int lane = threadIdx.x & 63, wave = U(threadIdx.x >> 6), nwaves = kErwinThreads >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int Kp = round32(K), nks = Kp >> 5;
    const int nto = (O + 15) >> 4, ntiles = ((n + 15) >> 4) * nto;
    const bool al16 = ((b0 | dst | (4 * ldd) | (mode == EP_SWIGLU ? (di.base | di.S | di.plane) : 0)) & 15) == 0;
    const size_t plane = (size_t)O * Kp;
    const int my_tiles = ntiles > wave ? (ntiles - wave + nwaves - 1) / nwaves : 0;
    const int items = my_tiles * nks;
    constexpr bool dual = DUAL;
    if (items == 0) return;
#ifdef HAET_STAMPS
#define LSTAMP(i) do { if (mode == EP_RESID && K <= 64) STAMP(i); } while (0)
#else
#define LSTAMP(i) do { } while (0)
#endif
    LSTAMP(40);

    // Item cursors advance incrementally (no integer divisions: each costs ~40 VALU instructions and
    // there were six per item).  An item = (tile, k-step); tile = (tt, ot), tiles are dealt to the
    // waves round-robin.
    struct Cur {
        int tt, ot, ks;
    };
    const auto first = [&]() {
        Cur c;
        c.tt = wave / nto;
        c.ot = wave - c.tt * nto;
        c.ks = 0;
        return c;
    };
    const auto advance = [&](Cur& c) {
        if (++c.ks == nks) {
            c.ks = 0;
            c.ot += nwaves;
            while (c.ot >= nto) {
                c.ot -= nto;
                ++c.tt;
            }
        }
    };
    u32x4 w0[4], w1[4];
    typedef const __attribute__((address_space(1))) u32x4* gv4;  // explicit global loads (not flat)
    // Weights come from the TILE-ORDERED copy behind the two row-major planes (pack.py, _Blob.planes; round 4): piece
    // (ot, ks, plane) = 1 KB, this lane's 16 B at lane * 16 -- one wave load = 1 KB of consecutive bytes.  From the row-major
    // planes the same fragment was 16 rows x 64 B with NEIGHBOURING lanes in different rows, which the vector cache serves
    // lane by lane: ~2.8k cycles per item, 11 B/cycle per CU, whatever the prefetch depth (stamps at config C2).
    const unsigned short* Ta = Wa + 2 * plane + lane * 8;
    const unsigned short* Tb = dual ? Wb + 2 * plane + lane * 8 : Ta;
    auto fetch = [&](const Cur& c, u32x4 (&wv)[4]) {
        const size_t off = (size_t)(c.ot * nks + c.ks) * 1024;  // bf16 elements: two 1-KB pieces per (ot, ks)
        wv[0] = *(gv4)(Ta + off);
        wv[1] = *(gv4)(Ta + off + 512);
        if (dual) {
            wv[2] = *(gv4)(Tb + off);
            wv[3] = *(gv4)(Tb + off + 512);
        }
    };
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // the activation fragments are read one item ahead as well (an LDS round trip per item otherwise)
    u32x4 xh, xl;
    auto read_x = [&](const Cur& c) {
        const int xo = x.base + min(c.tt * 16 + l15, n - 1) * x.S + (32 * c.ks + 8 * kq) * 2;
        xh = *reinterpret_cast<const u32x4*>(smem + xo);
        xl = *reinterpret_cast<const u32x4*>(smem + xo + x.plane);
    };
    auto step = [&](const Cur& c, const u32x4 (&wv)[4], const u32x4 xh, const u32x4 xl) {
        LSTAMP(42);
        // W is the A operand, X the B operand: the accumulator is Y^T -- lane = token column, registers =
        // four CONSECUTIVE outputs -- so every epilogue moves 16 B (fp32 rows) or 8 B (bf16 image rows)
        // per lane instead of four strided 4-B / 2-B pieces.
        acc0 = mfma3(wv[0], wv[1], xh, xl, acc0);
        if (dual) acc1 = mfma3(wv[2], wv[3], xh, xl, acc1);
#ifdef HAET_STAMPS
        if (mode == EP_RESID && K <= 64) { asm volatile("" :: "v"(acc0)); }
#endif
        LSTAMP(43);
        if (c.ks == nks - 1) {
            const int o0 = c.ot * 16 + 4 * kq, t = c.tt * 16 + l15;
            if (t < n && o0 < O) {
                const bool vec = al16 && o0 + 3 < O;  // (true in every shape the models produce)
                float v[4];
                if (vec) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(smem + b0 + 4 * o0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc0[r] + b4[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc0[r] + F(b0 + 4 * min(o0 + r, O - 1));
                }
                if (mode == EP_QKV || mode == EP_RESID || mode == EP_BN) {
                    const int row = dst + 4 * (t * ldd + o0);
                    if (mode == EP_BN && sc >= 0) {  // (sc < 0: train-mode BatchNorm, applied by bn_train_phase)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            v[r] = v[r] * F(sc + 4 * min(o0 + r, O - 1)) + F(sh + 4 * min(o0 + r, O - 1));
                    }
                    if (vec) {
                        f32x4 y = {v[0], v[1], v[2], v[3]};
                        if (mode == EP_RESID) y += *reinterpret_cast<const f32x4*>(smem + row);
                        *reinterpret_cast<f32x4*>(smem + row) = y;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (o0 + r < O) F(row + 4 * r) = (mode == EP_RESID ? F(row + 4 * r) : 0.f) + v[r];
                    }
                } else if (mode == EP_SWIGLU) {
                    float g[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v[r]));
                        g[r] = (acc1[r] + F(b1 + 4 * min(o0 + r, O - 1))) * (v[r] * sig);
                    }
                    if (vec) {
                        di.put4(t, o0, g);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (o0 + r < O) di.put(t, o0 + r, g[r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = o0 + r;
                        if (o < O) {
                            const int q = o / aux, ch = o - q * aux;  // 'n (m d) -> (n m) d'
                            float& y = F(dst + 4 * ((t * st + q) * ldd + ch));
                            y = sc >= 0 ? (y + v[r]) * F(sc + 4 * ch) + F(sh + 4 * ch) : y + v[r];
                        }
                    }
                }
            }
            acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
            acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // The fetch cursor runs two items ahead of the compute cursor.  (Round 4, measured at BASELINE config C2 with stamps: a ring
    // of four items made every Linear SLOWER -- level-0 block 60k -> 80k cycles, the first item's wait 1.35k -> 4.3k: the
    // weight stream was throughput-bound at ~11 B/cycle per CU, not
    // latency-bound.  The cause was the ACCESS SHAPE, not the depth: see `fetch` above.  With the tile-ordered copy a ring of
    // three or four items was measured again: C2 token stage 0.177 / 0.178 / 0.188 ms for 2 / 3 / 4 -- two it stays.)
    // Fetches are UNCONDITIONAL
    // (addresses clamped) and the loop runs whole 2-step rounds (a ghost step past the end multiplies
    // garbage into an accumulator that is never written): with a conditional load between a fetch and
    // its use, hipcc's waitcnt pass cannot count the younger loads in flight and waits vmcnt(0),
    // which silently serialises the prefetch.
    Cur cf = first(), cc = first();
    if (pre) {  // (a compile-time fact after inlining)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            w0[i] = pre->a[i];
            w1[i] = pre->b[i];
        }
    } else {
        fetch(cf, w0);
    }
    advance(cf);
    if (!pre) fetch(cf, w1);
    advance(cf);
    LSTAMP(41);
    read_x(cc);
    for (int it = 0; it < items; it += 2) {  // two named register sets, statically indexed
        {
            u32x4 c0[4] = {w0[0], w0[1], w0[2], w0[3]};
            const u32x4 ah = xh, al = xl;
            fetch(cf, w0);
            advance(cf);
            Cur cn = cc;
            advance(cn);
            read_x(cn);  // (a ghost item past the end reads a clamped, valid row)
            step(cc, c0, ah, al);
            cc = cn;
        }
        {
            u32x4 c1[4] = {w1[0], w1[1], w1[2], w1[3]};
            const u32x4 ah = xh, al = xl;
            fetch(cf, w1);
            advance(cf);
            Cur cn = cc;
            advance(cn);
            read_x(cn);
            step(cc, c1, ah, al);
            cc = cn;
        }
    }
    LSTAMP(44);
}

__device__ __forceinline__ void linear_core(
    const Img x, int n, int K, const unsigned short* __restrict__ Wa,
    const unsigned short* __restrict__ Wb, int O, int mode, int dst, int ldd, int b0, int b1, int sc,
    int sh, const Img di, int aux, int st, const WPre* pre = nullptr) {
    switch (mode) {  // `mode` is a literal at every call site: this folds to one call
        case EP_QKV: linear_core_t<EP_QKV>(x, n, K, Wa, Wb, O, dst, ldd, b0, 0, 0, 0, 0, pre); break;
        case EP_RESID: linear_core_t<EP_RESID>(x, n, K, Wa, Wb, O, dst, ldd, b0, 0, 0, 0, 0, pre); break;
        case EP_SWIGLU: linear_core_t<EP_SWIGLU>(x, n, K, Wa, Wb, O, dst, ldd, b0, b1, di.base, di.S, di.plane, pre); break;
        case EP_BN: linear_core_t<EP_BN>(x, n, K, Wa, Wb, O, dst, ldd, b0, sc, sh, 0, 0); break;
        default: linear_core_t<EP_UNPOOL>(x, n, K, Wa, Wb, O, dst, ldd, b0, sc, sh, aux, st); break;
    }
}

// ia[t][:] = split( RMSNorm(X[t][:]) * w  [+ pe_b + rel[t] . pe_w] ).  A thread owns 4 consecutive
// channels of one token (one ds_read_b128), the c/4 threads of a row reduce the sum of squares
// with xor-shuffles: the dependent chain is ~5 LDS round trips whatever c is (it was one per
// channel).  nn.RMSNorm(eps=None): eps = finfo(float32).eps (layers.py:25-26); PE:
// attention.py:59-61.  X, w, rel, pe_w, pe_b: LDS byte offsets (pe_w < 0: no positional term).
__device__ ERWIN_PHASE void norm_rows(int X, int ldx, int n, int c, int w, int rel,
                                                    int d, int pe_w, int pe_b, const Img out_) {
    const Img out = UI(out_);
    X = U(X), ldx = U(ldx), n = U(n), c = U(c), w = U(w), rel = U(rel), d = U(d), pe_w = U(pe_w), pe_b = U(pe_b);
    const int cp = round32(c), tpr = cp >> 2;  // threads per row: 8, 16 or 32 (power of two)
    const int sub = threadIdx.x & (tpr - 1), ch = 4 * sub;
    const int rows_per_pass = kErwinThreads / tpr;
    const int passes = (n + rows_per_pass - 1) / rows_per_pass;
    for (int ps = 0; ps < passes; ++ps) {
        const int t = ps * rows_per_pass + threadIdx.x / tpr;
        const bool live = t < n && ch < c;
        f32x4 xv = {0.f, 0.f, 0.f, 0.f}, wv = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            xv = *reinterpret_cast<const f32x4*>(smem + X + 4 * (t * ldx + ch));
            wv = *reinterpret_cast<const f32x4*>(smem + w + 4 * ch);
        }
        float ss = fmaf(xv[0], xv[0], fmaf(xv[1], xv[1], fmaf(xv[2], xv[2], xv[3] * xv[3])));
        for (int off = 1; off < tpr; off <<= 1) ss += __shfl_xor(ss, off);
        const float r = rsqrtf(ss / (float)c + 1.1920928955078125e-07f);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            v = xv * r * wv;
            if (pe_w >= 0) {
                f32x4 pe = *reinterpret_cast<const f32x4*>(smem + pe_b + 4 * ch);
                for (int j = 0; j < d; ++j)
                    pe += F(rel + 4 * (t * d + j)) * *reinterpret_cast<const f32x4*>(smem + pe_w + 4 * (j * c + ch));
                v += pe;
            }
        }
        if (t < n) {  // pad channels (c..cp) are written as zeros
#pragma unroll
            for (int e = 0; e < 4; ++e) out.put(t, ch + e, v[e]);
        }
    }
}

// rel[tok] = P[tok] - mean over its ball (attention.py:35-40); ball b = places [b*ball, ..).
// One thread per (dimension, place): a ball's places are `ball` consecutive lanes, so for the
// power-of-two ball sizes (<= 64) the ball sum is a segmented xor-butterfly.
__device__ ERWIN_PHASE void ball_rel_pos(int P, int n, int d, int ball, int perm, int rel) {
    P = U(P), n = U(n), d = U(d), ball = U(ball), perm = U(perm), rel = U(rel);
    if (ball <= 64 && (ball & (ball - 1)) == 0) {
        const int total = (n * d + 63) & ~63;  // whole waves: every lane takes part in the shuffles
        for (int i = threadIdx.x; i < total; i += kErwinThreads) {
            const int j = i / n, p = i - j * n;
            const bool ok = j < d;
            const int tok = ok ? I(perm + 4 * p) : 0;
            const float v = ok ? F(P + 4 * (tok * d + j)) : 0.f;
            float sum = v;
            for (int off = 1; off < ball; off <<= 1) sum += __shfl_xor(sum, off);
            if (ok) F(rel + 4 * (tok * d + j)) = v - sum / (float)ball;
        }
        return;
    }
    const int nb = n / ball;
    for (int i = threadIdx.x; i < nb * d; i += kErwinThreads) {
        const int b = i / d, j = i % d;
        float sm = 0.f;
        for (int p = 0; p < ball; ++p) sm += F(P + 4 * (I(perm + 4 * (b * ball + p)) * d + j));
        const float mean = sm / (float)ball;
        for (int p = 0; p < ball; ++p) {
            const int tok = I(perm + 4 * (b * ball + p));
            F(rel + 4 * (tok * d + j)) = F(P + 4 * (tok * d + j)) - mean;
        }
    }
}

// exact softmax(q k^T / sqrt(hd)) v inside each ball (attention.py:68-84; the reference calls
// flash_attn_qkvpacked_func here).  One thread per (place, head); qkv rows (fp32, LDS) are
// [q(h,hd) | k(h,hd) | v(h,hd)] ("(three h d)").  The output goes straight into the split image
// that feeds the proj Linear.
template <int HD>
__device__ __attribute__((noinline)) void ball_attention(int qkv, int ldq, int n, int c, int heads,
                                                         int ball, int perm, const Img out_) {
    constexpr int KC = 8;  // keys per step: 8 independent score chains, one rescale per step
    const Img out = UI(out_);
    qkv = U(qkv), ldq = U(ldq), n = U(n), c = U(c), heads = U(heads), ball = U(ball), perm = U(perm);
    const float scale = 1.0f / sqrtf((float)HD);
    for (int i = threadIdx.x; i < n * heads; i += kErwinThreads) {
        const int place = i / heads, h = i % heads;
        const int tq = I(perm + 4 * place), b0 = (place / ball) * ball;
        float q[HD], acc[HD];
#pragma unroll
        for (int e = 0; e < HD; ++e) {
            q[e] = F(qkv + 4 * (tq * ldq + h * HD + e)) * scale;
            acc[e] = 0.f;
        }
        float m = -INFINITY, l = 0.f;
        for (int p0 = 0; p0 < ball; p0 += KC) {
            int kr[KC];
            float sc[KC];
            float mx = m;
#pragma unroll
            for (int u = 0; u < KC; ++u) {
                const int p = min(p0 + u, ball - 1);
                kr[u] = qkv + 4 * (I(perm + 4 * (b0 + p)) * ldq + c + h * HD);
            }
#pragma unroll
            for (int u = 0; u < KC; ++u) {
                float sv = 0.f;
#pragma unroll
                for (int e = 0; e < HD; ++e) sv = fmaf(q[e], F(kr[u] + 4 * e), sv);
                sc[u] = (p0 + u < ball) ? sv : -INFINITY;
                mx = fmaxf(mx, sc[u]);
            }
            const float a = __expf(m - mx);  // m = -inf on the first step -> 0
            l *= a;
#pragma unroll
            for (int e = 0; e < HD; ++e) acc[e] *= a;
#pragma unroll
            for (int u = 0; u < KC; ++u) {
                const float pr = __expf(sc[u] - mx);
                l += pr;
#pragma unroll
                for (int e = 0; e < HD; ++e) acc[e] = fmaf(pr, F(kr[u] + 4 * (c + e)), acc[e]);
            }
            m = mx;
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int e = 0; e < HD; ++e) out.put(tq, h * HD + e, acc[e] * inv);
    }
}

// Ball attention on the matrix cores for head dims 8 (both hierarchy levels of the C3 shape), 16 and 32
// (config C2) and balls of <= 32 tokens: one wave per (32-token tile, head).
//   St[k][q] = sum_d K[k][d] Q[q][d]        v_mfma_f32_32x32x16_bf16 x 3 (split bf16; d padded 8 -> 16
//                                           with zeros: the hf = 1 halves of both operands are zero)
//   lane = query column, registers = the 16 keys (r&3) + 8(r>>2) + 4hf: the softmax over the keys of a
//   query is over one lane's registers plus its partner lane (xor 32); keys of another ball in the
//   same tile (ball < 32) are masked to -inf.
//   Ot[d][q] = sum_k Vt[d][k] Pt[k][q]      A = V gathered in the accumulator's key order, B = P straight
//                                           from the accumulator (no lane movement); rows d >= 8 are zero
//   lane (q, hf) ends up with O[q][4hf .. 4hf+3] in registers 0..3 and the softmax sum of ITS query.
// The scalar version above reads every K and V row once per (query, head) thread -- 0.5 MB of LDS
// traffic per level-0 block, 10k cycles; this one reads them once per (tile, head).
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (RNE)
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split8v(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned h, l;
        split_bf16_pair(v[2 * i], v[2 * i + 1], h, l);
        hi[i] = h, lo[i] = l;
    }
}
__device__ __forceinline__ f32x16 mfma3_32(u32x4 ah, u32x4 al, u32x4 bh, u32x4 bl, f32x16 c) {
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, ah), a1 = __builtin_bit_cast(bf16x8, al);
    const bf16x8 b0 = __builtin_bit_cast(bf16x8, bh), b1 = __builtin_bit_cast(bf16x8, bl);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);
    return c;
}

template <int HD>  // 8, 16 or 32
__device__ ERWIN_PHASE void ball_attention_mfma(int qkv, int ldq, int n, int c, int heads,
                                                int ball, int perm, const Img out_) {
    constexpr int NS = HD > 16 ? HD / 16 : 1;  // 16-wide k-steps of the score product
    const Img out = UI(out_);
    qkv = U(qkv), ldq = U(ldq), n = U(n), c = U(c), heads = U(heads), ball = U(ball), perm = U(perm);
    const int lane = threadIdx.x & 63, wave = U(threadIdx.x >> 6), hf = lane >> 5, l31 = lane & 31;
    const float qscale = 1.4426950408889634f / sqrtf((float)HD);  // softmax in base 2
    // (round 4: n need not be a multiple of 32 -- the 16-token coarse level of M = 32 clouds, BASELINE config C2 and the
    // reference's velocity-mem-checker shape, ran on the generic vector-ALU attention: 46.7k cycles per block against 9.2k
    // here: the lanes past the end mirror the last token, take no part as keys and store nothing)
    const int items = ((n + 31) >> 5) * heads, bmask = 31 & ~(ball - 1);  // ball: a power of two <= 32
    for (int item = wave; item < items; item += kErwinThreads / 64) {
        const int tile = item / heads, h = item - tile * heads;
        const int nv = n - tile * 32;  // valid tokens of this tile (>= 32 except on a partial last tile)
        const int t = I(perm + 4 * min(tile * 32 + l31, n - 1));  // my token: key row of St, query column of St
        const int row = qkv + 4 * (t * ldq + h * HD);
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            // this lane's 8 head-dim elements of the step: d = 16 s + 8 hf .. + 7 (HD = 8: the hf = 1 half is
            // the zero padding of d = 8..15; every lane still reads -- rows are 16-B aligned: kPad)
            const int d0 = HD == 8 ? 0 : 16 * s + 8 * hf;
            const bool pad = HD == 8 && hf;
            float kv[8], qv[8];
#pragma unroll
            for (int e4 = 0; e4 < 2; ++e4) {
                const f32x4 k4 = *reinterpret_cast<const f32x4*>(smem + row + 4 * (c + d0 + 4 * e4));
                const f32x4 q4 = *reinterpret_cast<const f32x4*>(smem + row + 4 * (d0 + 4 * e4));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    kv[4 * e4 + e] = pad ? 0.f : k4[e];
                    qv[4 * e4 + e] = pad ? 0.f : q4[e] * qscale;
                }
            }
            u32x4 kh, kl, qh, ql;
            split8v(kv, kh, kl);
            split8v(qv, qh, ql);
            S = mfma3_32(kh, kl, qh, ql, S);
        }
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * hf;
            if (((key ^ l31) & bmask) || key >= nv) S[r] = -INFINITY;  // another ball of the same tile / past the end
            m = fmaxf(m, S[r]);
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            S[r] = __builtin_amdgcn_exp2f(S[r] - m);
            l += S[r];
        }
        l += __shfl_xor(l, 32);
        f32x16 O;
#pragma unroll
        for (int r = 0; r < 16; ++r) O[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float vv[8], pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int key = 16 * s + 8 * (j >> 2) + 4 * hf + (j & 3);  // the accumulator's key order
                const int tk = __shfl(t, key);
                const float v1 = F(qkv + 4 * (tk * ldq + 2 * c + h * HD + (l31 & (HD - 1))));  // unconditional read
                vv[j] = l31 < HD ? v1 : 0.f;
                pv[j] = S[8 * s + j];
            }
            u32x4 vh, vl, ph, pl;
            split8v(vv, vh, vl);
            split8v(pv, ph, pl);
            O = mfma3_32(vh, vl, ph, pl, O);
        }
        // lane (q, hf) holds O[q][d] for d = (r & 3) + 8 (r >> 2) + 4 hf, r < HD / 2 (rows d >= HD are zero)
        const float inv = 1.0f / l;
        if (l31 < nv) {
#pragma unroll
            for (int r = 0; r < HD / 2; ++r) out.put(t, h * HD + (r & 3) + 8 * (r >> 2) + 4 * hf, O[r] * inv);
        }
    }
}

__device__ ERWIN_PHASE void attention_dispatch(int hd, int qkv, int ldq, int n, int c,
                                                             int heads, int ball, int perm, const Img out) {
    if (ball <= 32 && (ball & (ball - 1)) == 0 && n % ball == 0) {
        if (hd == 8) return ball_attention_mfma<8>(qkv, ldq, n, c, heads, ball, perm, out);
        if (hd == 16) return ball_attention_mfma<16>(qkv, ldq, n, c, heads, ball, perm, out);
        if (hd == 32) return ball_attention_mfma<32>(qkv, ldq, n, c, heads, ball, perm, out);
    }
    switch (hd) {
        case 1: ball_attention<1>(qkv, ldq, n, c, heads, ball, perm, out); break;
        case 2: ball_attention<2>(qkv, ldq, n, c, heads, ball, perm, out); break;
        case 4: ball_attention<4>(qkv, ldq, n, c, heads, ball, perm, out); break;
        case 8: ball_attention<8>(qkv, ldq, n, c, heads, ball, perm, out); break;
        case 16: ball_attention<16>(qkv, ldq, n, c, heads, ball, perm, out); break;
        case 32: ball_attention<32>(qkv, ldq, n, c, heads, ball, perm, out); break;
        default: ball_attention<64>(qkv, ldq, n, c, heads, ball, perm, out); break;
    }
}

__device__ __forceinline__ const unsigned short* wptr(const float* W, int off) {
    return reinterpret_cast<const unsigned short*>(W + off);
}

// ErwinTransformerBlock (layers.py:30-32) on the tokens of one level, in place
#if defined(HAET_STAMPS) && defined(HAET_STAMP_LVL)
#define BSTAMP(i) do { if (lvl == HAET_STAMP_LVL) STAMP(i); } while (0)  // phases of one hierarchy level only
#else
#define BSTAMP(i) STAMP(i)
#endif
__device__ ERWIN_PHASE void erwin_block(int lvl, const haet_erwin_layer& L,
                                                      const haet_erwin_block_w& w,
                                                      const float* __restrict__ W, int d, int perm) {
    lvl = U(lvl), d = U(d), perm = U(perm), W = UP(W);
    const int X = T(T_X + lvl), n = T(T_N + lvl), ld = T(T_LD + lvl), c = L.c, hid = L.hidden;
    const int ldq = 3 * c + kPad, wide = T(T_WIDE);
    const Img ia = {T(T_IA), img_stride(c), n * img_stride(c)};
    const Img ih = {wide, img_stride(hid), n * img_stride(hid)};
    const Img none = {0, 0, 0};
    // norm weights, positional projection and every bias of this block: one coalesced copy to LDS
    // (they are re-read per token / per tile; from global each use is a dependent L2/HBM trip)
    const int par = T(T_PAR) - 4 * w.small_off;  // par + 4 * <blob offset> addresses the LDS copy
    BSTAMP(16);
    const WPre p_qkv = linear_prefetch<false>(c, wptr(W, w.qkv_w), nullptr, 3 * c);
    stage_params(T(T_PAR), W + w.small_off, w.small_len);
    // x~ = RMSNorm(x) + pe_proj(pos - ball centre)            (attention.py:59-61)
    ball_rel_pos(T(T_P + lvl), n, d, L.ball, perm, T(T_REL));
    __syncthreads();
    BSTAMP(17);
    norm_rows(X, ld, n, c, par + 4 * w.norm1, T(T_REL), d, par + 4 * w.pe_w, par + 4 * w.pe_b, ia);
    __syncthreads();
    BSTAMP(18);
    linear_core(ia, n, c, wptr(W, w.qkv_w), nullptr, 3 * c, EP_QKV, wide, ldq, par + 4 * w.qkv_b, 0, 0, 0,
                none, 0, 0, &p_qkv);
    __syncthreads();
    BSTAMP(19);
    const WPre p_proj = linear_prefetch<false>(c, wptr(W, w.proj_w), nullptr, c);
    attention_dispatch(c / L.heads, wide, ldq, n, c, L.heads, L.ball, perm, ia);
    __syncthreads();
    BSTAMP(20);
    linear_core(ia, n, c, wptr(W, w.proj_w), nullptr, c, EP_RESID, X, ld, par + 4 * w.proj_b, 0, 0, 0, none,
                0, 0, &p_proj);
    __syncthreads();
    BSTAMP(21);
    const WPre p_12 = linear_prefetch<true>(c, wptr(W, w.w1_w), wptr(W, w.w2_w), hid);
    // x += W3( W2 x^ * silu(W1 x^) ),  x^ = RMSNorm(x)          (mlp.py:17-18)
    norm_rows(X, ld, n, c, par + 4 * w.norm2, 0, 0, -1, 0, ia);
    {
        const int padk = round32(hid) - hid;  // zero the k padding of the hidden image
        for (int i = threadIdx.x; i < n * padk; i += kErwinThreads) ih.put(i / padk, hid + i % padk, 0.f);
    }
    __syncthreads();
    BSTAMP(22);
    const WPre p_3 = linear_prefetch<false>(hid, wptr(W, w.w3_w), nullptr, c);
    linear_core(ia, n, c, wptr(W, w.w1_w), wptr(W, w.w2_w), hid, EP_SWIGLU, 0, 0, par + 4 * w.w1_b,
                par + 4 * w.w2_b, 0, 0, ih, 0, 0, &p_12);
    __syncthreads();
    BSTAMP(23);
    linear_core(ih, n, hid, wptr(W, w.w3_w), nullptr, c, EP_RESID, X, ld, par + 4 * w.w3_b, 0, 0, 0, none, 0,
                0, &p_3);
    __syncthreads();
    BSTAMP(24);
}

// ---------------------------------------------------------------------------------------------------------
// Train-mode BatchNorm1d of BallPooling / BallUnpooling (pooling.py:23,38,56,68): the statistics run over the
// tokens of ALL clouds, i.e. over all workgroups of the launch.  Every workgroup reduces its own tokens to a
// per-channel (mean, M2) pair, publishes it, the workgroups meet at a grid barrier, and each combines all pairs
// itself (Chan's formula with equal counts: no cancellation, same result on every workgroup).  The launch has
// one workgroup per cloud and at most one per CU (checked on the host), so all of them are resident.
struct ErwinTrain {
    float* rmean[2 * HAET_ERWIN_MAX_LEVELS];  // [2l] pool l, [2l+1] unpool l: running statistics, updated in place
    float* rvar[2 * HAET_ERWIN_MAX_LEVELS];
    long long* nbt[2 * HAET_ERWIN_MAX_LEVELS];
    float* stats;     // out [2*MAX_LEVELS][2][cmax]: batch mean, 1/sqrt(var + eps) (nullptr: not wanted)
    unsigned* bar;    // grid-barrier counter (zeroed by the launch function before every launch)
    unsigned* tmo;    // set to 1 if a barrier gave up (bounded spin)
    float* partial;   // [2*MAX_LEVELS][n_clouds][2][cmax]
    int cmax, on;
    // tape for the backward kernel (erwin_bwd.hip), independent of `on`: the tokens of the op's input level at the
    // start of every op, [n_clouds][n_ops][tape_op_floats] (rows of c floats, no padding); nullptr: not recorded
    float* tape;
    int tape_op_floats;
};

// Counter barrier over the workgroups of the launch (cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md
// 'barrier-counter'): every storing wave drains its stores, the workgroup meets, ONE lane releases at agent scope,
// arrives, polls relaxed (bounded), acquires at agent scope; the workgroup barrier after it holds the other waves
// until the invalidate has completed.
__device__ __attribute__((noinline)) void erwin_grid_barrier(unsigned* counter, unsigned target, unsigned* tmo) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1u << 22)) {  // ~1 s: a workgroup is not resident -- give up rather than hang the GPU
                __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// X [n][ld] (LDS, fp32) holds the pre-BatchNorm values of this cloud; normalised in place with the batch
// statistics of all clouds.  gamma / beta: LDS byte offsets of the affine parameters.
__device__ __attribute__((noinline)) void bn_train_phase(int X, int ld, int n, int c, int gamma, int beta, float eps,
                                                         float momentum, int slot, const ErwinTrain& tr, int cloud,
                                                         int n_clouds, int epoch) {
    X = U(X), ld = U(ld), n = U(n), c = U(c), gamma = U(gamma), beta = U(beta), slot = U(slot);
    const int bn = T(T_BN), cmax = tr.cmax, cmp = (cmax + 3) & ~3;
    float* part = tr.partial + ((size_t)slot * n_clouds + cloud) * 2 * cmax;
    const int ch = threadIdx.x;
    if (ch < c) {
        float sum = 0.f;
        for (int t = 0; t < n; ++t) sum += F(X + 4 * (t * ld + ch));
        const float mean = sum / (float)n;
        float m2 = 0.f;
        for (int t = 0; t < n; ++t) {
            const float dlt = F(X + 4 * (t * ld + ch)) - mean;
            m2 = fmaf(dlt, dlt, m2);
        }
        part[ch] = mean;
        part[cmax + ch] = m2;
    }
    erwin_grid_barrier(tr.bar, (unsigned)n_clouds * (unsigned)(epoch + 1), tr.tmo);
    if (ch < c) {
        const float* all = tr.partial + (size_t)slot * n_clouds * 2 * cmax;
        float msum = 0.f;
        for (int k = 0; k < n_clouds; ++k) msum += all[(size_t)k * 2 * cmax + ch];
        const float mean = msum / (float)n_clouds;
        float m2 = 0.f;
        for (int k = 0; k < n_clouds; ++k) {
            const float dk = all[(size_t)k * 2 * cmax + ch] - mean;
            m2 += all[(size_t)k * 2 * cmax + cmax + ch] + (float)n * dk * dk;
        }
        const float cnt = (float)n * (float)n_clouds;
        const float var = m2 / cnt, invstd = 1.0f / sqrtf(var + eps);
        const float sc = F(gamma + 4 * ch) * invstd;
        F(bn + 4 * ch) = sc;
        F(bn + 4 * (cmp + ch)) = F(beta + 4 * ch) - mean * sc;
        if (cloud == 0) {
            if (tr.stats) {
                tr.stats[(size_t)slot * 2 * cmax + ch] = mean;
                tr.stats[(size_t)slot * 2 * cmax + cmax + ch] = invstd;
            }
            if (tr.rmean[slot]) {  // nn.BatchNorm1d: running = (1 - momentum) running + momentum batch (unbiased variance)
                tr.rmean[slot][ch] = (1.0f - momentum) * tr.rmean[slot][ch] + momentum * mean;
                tr.rvar[slot][ch] = (1.0f - momentum) * tr.rvar[slot][ch] + momentum * (m2 / fmaxf(cnt - 1.0f, 1.0f));
            }
            if (ch == 0 && tr.nbt[slot]) *tr.nbt[slot] += 1;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n * c; i += kErwinThreads) {
        const int t = i / c, k = i - t * c;
        float& y = F(X + 4 * (t * ld + k));
        y = fmaf(y, F(bn + 4 * k), F(bn + 4 * (cmp + k)));
    }
    __syncthreads();
}

// BallPooling (pooling.py:25-40): level l -> l+1
__device__ ERWIN_PHASE void erwin_pool(int l, const haet_erwin_desc& D, const float* __restrict__ W,
                                       const ErwinTrain& tr, int cloud, int n_clouds, int& epoch) {
    const haet_erwin_resample& R = D.pool[l];
    const int d = D.d, st = R.stride, n2 = T(T_N + l + 1), c = D.enc[l].c, K = st * c + st * d;
    const int X0 = T(T_X + l), ld0 = T(T_LD + l), P0 = T(T_P + l), P1 = T(T_P + l + 1);
    const Img rows = {T(T_WIDE), img_stride(K), n2 * img_stride(K)};
    const Img none = {0, 0, 0};
    if (st == 1) {  // "no pooling" (pooling.py:26-27): the node passes through, no projection, no BatchNorm
        const int X1 = T(T_X + l + 1), ld1 = T(T_LD + l + 1);
        for (int i = threadIdx.x; i < n2 * c; i += kErwinThreads) {
            const int p = i / c, k = i % c;
            F(X1 + 4 * (p * ld1 + k)) = F(X0 + 4 * (p * ld0 + k));
        }
        for (int i = threadIdx.x; i < n2 * d; i += kErwinThreads) F(P1 + 4 * i) = F(P0 + 4 * i);
        __syncthreads();
        return;
    }
    stage_params(T(T_PAR), W + R.small_off, R.small_len);
    for (int i = threadIdx.x; i < n2 * d; i += kErwinThreads) {
        const int p = i / d, j = i % d;
        float sum = 0.f;
        for (int q = 0; q < st; ++q) sum += F(P0 + 4 * ((p * st + q) * d + j));
        F(P1 + 4 * i) = sum / (float)st;
    }
    __syncthreads();
    const int Kp = round32(K);
    for (int i = threadIdx.x; i < n2 * Kp; i += kErwinThreads) {
        const int p = i / Kp, k = i % Kp;
        float v = 0.f;
        if (k < st * c) {
            v = F(X0 + 4 * ((p * st + k / c) * ld0 + k % c));
        } else if (k < K) {
            const int r = k - st * c, q = r / d, j = r % d;
            v = F(P0 + 4 * ((p * st + q) * d + j)) - F(P1 + 4 * (p * d + j));
        }
        rows.put(p, k, v);
    }
    __syncthreads();
    const int par = T(T_PAR) - 4 * R.small_off;
    const bool train = tr.on != 0;  // batch statistics: the epilogue leaves y, bn_train_phase normalises it
    linear_core(rows, n2, K, wptr(W, R.w), nullptr, D.enc[l + 1].c, EP_BN, T(T_X + l + 1), T(T_LD + l + 1),
                par + 4 * R.b, 0, train ? -1 : par + 4 * R.bn_scale, par + 4 * R.bn_shift, none, 0, 0);
    __syncthreads();
    if (train) {
        bn_train_phase(T(T_X + l + 1), T(T_LD + l + 1), n2, D.enc[l + 1].c, par + 4 * R.bn_scale, par + 4 * R.bn_shift,
                       R.bn_eps, R.bn_momentum, 2 * l, tr, cloud, n_clouds, epoch);
        ++epoch;
    }
}

// BallUnpooling (pooling.py:58-72): level l+1 -> l
__device__ ERWIN_PHASE void erwin_unpool(int l, const haet_erwin_desc& D, const float* __restrict__ W,
                                         const ErwinTrain& tr, int cloud, int n_clouds, int& epoch) {
    const haet_erwin_resample& R = D.unpool[l];
    const int d = D.d, st = R.stride, n2 = T(T_N + l + 1), c2 = D.enc[l + 1].c, c = D.enc[l].c;
    const int K = c2 + st * d, Kp = round32(K);
    const int X1 = T(T_X + l + 1), ld1 = T(T_LD + l + 1), P0 = T(T_P + l), P1 = T(T_P + l + 1);
    const Img rows = {T(T_WIDE), img_stride(K), n2 * img_stride(K)};
    const Img none = {0, 0, 0};
    stage_params(T(T_PAR), W + R.small_off, R.small_len);
    for (int i = threadIdx.x; i < n2 * Kp; i += kErwinThreads) {
        const int p = i / Kp, k = i % Kp;
        float v = 0.f;
        if (k < c2) {
            v = F(X1 + 4 * (p * ld1 + k));
        } else if (k < K) {
            const int r = k - c2, q = r / d, j = r % d;
            v = F(P0 + 4 * ((p * st + q) * d + j)) - F(P1 + 4 * (p * d + j));
        }
        rows.put(p, k, v);
    }
    __syncthreads();
    const int par = T(T_PAR) - 4 * R.small_off;
    const bool train = tr.on != 0;
    linear_core(rows, n2, K, wptr(W, R.w), nullptr, st * c, EP_UNPOOL, T(T_X + l), T(T_LD + l), par + 4 * R.b, 0,
                train ? -1 : par + 4 * R.bn_scale, par + 4 * R.bn_shift, none, c, st);
    __syncthreads();
    if (train) {
        bn_train_phase(T(T_X + l), T(T_LD + l), n2 * st, c, par + 4 * R.bn_scale, par + 4 * R.bn_shift, R.bn_eps,
                       R.bn_momentum, 2 * l + 1, tr, cloud, n_clouds, epoch);
        ++epoch;
    }
}

// largest small-parameter group (floats, multiple of 4)
__host__ __device__ inline int erwin_max_small(const haet_erwin_desc& D) {
    int m = 0;
    for (int l = 0; l < D.n_levels; ++l) {
        for (int j = 0; j < D.enc[l].depth; ++j) m = m > D.enc[l].blocks[j].small_len ? m : D.enc[l].blocks[j].small_len;
        if (l < D.n_levels - 1) {
            for (int j = 0; j < D.dec[l].depth; ++j) m = m > D.dec[l].blocks[j].small_len ? m : D.dec[l].blocks[j].small_len;
            m = m > D.pool[l].small_len ? m : D.pool[l].small_len;
            m = m > D.unpool[l].small_len ? m : D.unpool[l].small_len;
        }
    }
    return (m + 3) & ~3;
}

// LDS carve-up shared by host (sizing) and device; returns the total in floats
__host__ __device__ inline int erwin_layout(const haet_erwin_desc& D, int Mp, int* ia_bytes,
                                            int* wide_bytes) {
    int n = Mp, ia = 0, wide = 0, tot = T_WORDS, idx = 0;
    const int d = D.d;
    for (int l = 0; l < D.n_levels; ++l) {
        const haet_erwin_layer& L = D.enc[l];
        tot += n * (L.c + kPad) + ((n * d + 3) & ~3);
        idx += n;
        const int a = 2 * n * img_stride(L.c);
        ia = ia > a ? ia : a;
        int hid = L.hidden;
        if (l < D.n_levels - 1 && D.dec[l].hidden > hid) hid = D.dec[l].hidden;
        int w = n * (3 * L.c + kPad) * 4;
        const int wh = 2 * n * img_stride(hid);
        w = w > wh ? w : wh;
        if (l < D.n_levels - 1) {
            const int st = D.pool[l].stride, n2 = n / st;
            const int wp = 2 * n2 * img_stride(st * L.c + st * d);
            const int wu = 2 * n2 * img_stride(D.enc[l + 1].c + st * d);
            w = w > wp ? w : wp;
            w = w > wu ? w : wu;
            n = n2;
        }
        wide = wide > w ? wide : w;
    }
    *ia_bytes = (ia + 15) & ~15;
    *wide_bytes = (wide + 15) & ~15;
    int cmax = 4;
    for (int l = 0; l < D.n_levels; ++l) cmax = cmax > D.enc[l].c ? cmax : D.enc[l].c;
    // + identity map, tree map (fused token stage), train-mode BatchNorm scale/shift scratch [2][cmax]
    tot += (*ia_bytes + *wide_bytes) / 4 + ((Mp * d + 3) & ~3) + erwin_max_small(D) + idx + 2 * Mp + 2 * ((cmax + 3) & ~3);
    return tot;
}

// The weight blob starts with a copy of the descriptor (read with scalar loads).
struct ErwinDir {
    int t[T_WORDS];
};

// byte offsets of every LDS region (the T table the device routines read through T())
inline void erwin_directory(const haet_erwin_desc& D, int Mp, ErwinDir* out) {
    int ia_bytes, wide_bytes;
    erwin_layout(D, Mp, &ia_bytes, &wide_bytes);
    int* t = out->t;
    for (int i = 0; i < T_WORDS; ++i) t[i] = 0;
    const int nl = D.n_levels, d = D.d;
    int p = 4 * T_WORDS, n = Mp;
    for (int l = 0; l < nl; ++l) {
        t[T_N + l] = n;
        t[T_LD + l] = D.enc[l].c + kPad;
        t[T_X + l] = p;
        p += 4 * n * (D.enc[l].c + kPad);
        if (l < nl - 1) n /= D.pool[l].stride;
    }
    for (int l = 0; l < nl; ++l) {
        t[T_P + l] = p;
        p += 4 * ((t[T_N + l] * d + 3) & ~3);
    }
    t[T_IA] = p;
    p += ia_bytes;
    t[T_WIDE] = p;
    p += wide_bytes;
    t[T_REL] = p;
    p += 4 * ((Mp * d + 3) & ~3);
    t[T_PAR] = p;
    p += 4 * erwin_max_small(D);
    for (int l = 0; l < nl; ++l) {
        t[T_ROT + l] = p;
        p += 4 * t[T_N + l];
    }
    t[T_IDENT] = p;
    p += 4 * Mp;
    t[T_TREE] = p;
    p += 4 * Mp;
    t[T_BN] = p;
}

// FUSED: the token stage in ONE launch -- token positions (IrregularMesh.py:204-209), the ball tree with its
// rotated re-partitions (balltree.pyx:598-662) and the Erwin transformer; the tree stays in LDS (it is also written
// to tree_idx / tree_mask / rot / pos_out for the callers that want it, off the critical path).  While the builder
// runs (a latency chain of ~30 us on registers and LDS) the workgroup reads the Erwin weight blob once, a slice per
// tree level: the blob (~0.8 MB at C3) has been evicted from this XCD's L2 by the streaming kernels since the last
// forward, and every Linear of the Erwin stage is otherwise a chain of HBM misses (w1/w2 of a level-1 block: 8
// dependent ~2k-cycle rounds).  Clouds must be powers of two <= 64 (balltree_wave_ok).
constexpr int kWarmPieces = 16;  // 16-B loads per thread per builder step (64 VGPRs in flight)
template <bool FUSED>
__global__ __launch_bounds__(512) void erwin_kernel(
    const float* __restrict__ tokens, const float* __restrict__ pos,
    int32_t* __restrict__ tree_idx, uint8_t* __restrict__ tree_mask,
    int32_t* __restrict__ rot, const float* __restrict__ W, int M, int Mp,
    float* __restrict__ out, ErwinDir dir, ErwinTrain tr, WTreeCfg tcfg, float* __restrict__ pos_out,
    unsigned blob_pieces, int xcd_pack) {
    // xcd_pack = the number of clouds (else 0): the grid is 8x as large and only every eighth workgroup works, so that all
    // clouds run on ONE XCD (workgroups are dealt round-robin over the 8 XCDs: a speed assumption only) and share its L2 --
    // every cloud streams the same weight blob, and a CU reads beyond its L2 at a third of the rate it reads from it
    if (xcd_pack && (blockIdx.x & 7)) return;
    STAMP(0);
    const haet_erwin_desc& D = *reinterpret_cast<const haet_erwin_desc*>(W);
    const int cloud = xcd_pack ? blockIdx.x >> 3 : blockIdx.x, n_clouds = xcd_pack ? xcd_pack : gridDim.x, d = D.d, nl = D.n_levels;
    // LDS directory: computed on the host (one thread walking the layout cost ~9k cycles per launch)
    if (threadIdx.x < T_WORDS) I(4 * threadIdx.x) = dir.t[threadIdx.x];
    __syncthreads();
    STAMP(1);
    const int c0 = D.enc[0].c, X0 = T(T_X), ld0 = T(T_LD), P0 = T(T_P), ident = T(T_IDENT), tmap = T(T_TREE);
    if (FUSED) {
        const WTreeLds sink = {tmap, P0, {T(T_ROT), T(T_ROT + 1), T(T_ROT + 2), T(T_ROT + 3)}};
        // L2 warm-up, software-pipelined over the builder's steps: step i first folds the pieces requested at step
        // i - 1 (long since arrived) into a checksum nothing reads, then requests its own -- no load is waited for
        // inside the step that issued it.  Unconditional, clamped addresses (counted waits).
        typedef const __attribute__((address_space(1))) u32x4* gv4;
        u32x4 pend[kWarmPieces];
        unsigned sink_x = 0;
#pragma unroll
        for (int k = 0; k < kWarmPieces; ++k) pend[k] = u32x4{0u, 0u, 0u, 0u};
        auto step = [&](int i) {
#pragma unroll
            for (int k = 0; k < kWarmPieces; ++k) sink_x ^= pend[k][0];
            const unsigned base = (unsigned)i * (kErwinThreads * kWarmPieces) + threadIdx.x;
#pragma unroll
            for (int k = 0; k < kWarmPieces; ++k)
                pend[k] = ((gv4)W)[min(base + (unsigned)k * kErwinThreads, blob_pieces - 1)];
        };
        w_tree_build(tcfg, nullptr, tokens, c0, cloud, n_clouds, tree_idx, tree_mask, rot, pos_out, sink, step);
#pragma unroll
        for (int k = 0; k < kWarmPieces; ++k) sink_x ^= pend[k][0];
        asm volatile("" ::"v"(sink_x));
        __syncthreads();
        // gather the cloud into tree order (erwin_flash.py:176-181) through the tree map in LDS
        for (int i = threadIdx.x; i < Mp * c0; i += kErwinThreads) {
            const int slot = i / c0, ch = i % c0;
            F(X0 + 4 * (slot * ld0 + ch)) = tokens[((size_t)cloud * Mp + I(tmap + 4 * slot)) * c0 + ch];
        }
        for (int i = threadIdx.x; i < Mp; i += kErwinThreads) I(ident + 4 * i) = i;
    } else {
        // gather the cloud into tree order (erwin_flash.py:176-181)
        for (int i = threadIdx.x; i < Mp * c0; i += kErwinThreads) {
            const int slot = i / c0, ch = i % c0;
            F(X0 + 4 * (slot * ld0 + ch)) = tokens[(size_t)tree_idx[(size_t)cloud * Mp + slot] * c0 + ch];
        }
        for (int i = threadIdx.x; i < Mp * d; i += kErwinThreads) {
            const int slot = i / d, j = i % d;
            F(P0 + 4 * i) = pos[(size_t)tree_idx[(size_t)cloud * Mp + slot] * d + j];
        }
        for (int i = threadIdx.x; i < Mp; i += kErwinThreads) I(ident + 4 * i) = i;
        if (D.rotate) {
            size_t off = 0;
            for (int l = 0; l < nl; ++l) {
                const int nlv = T(T_N + l), ro = T(T_ROT + l);
                for (int i = threadIdx.x; i < nlv; i += kErwinThreads)
                    I(ro + 4 * i) = rot[off + (size_t)cloud * nlv + i] - cloud * nlv;
                off += (size_t)n_clouds * nlv;
            }
        }
    }
    __syncthreads();
    STAMP(2);
    int epoch = 0;  // grid barriers passed so far (train-mode BatchNorm)

    // Op list (erwin_flash.py:183-192): [blocks_l, pool_l]* , bottleneck blocks, [unpool_l, blocks_l]*.
    // kind 0 = block (a = level, b = block index | decoder flag << 8), 1 = pool, 2 = unpool.
    auto op_at = [&](int want, int& kind, int& a, int& b) {
        int i = 0;
        for (int l = 0; l < nl; ++l) {  // encoder + bottleneck
            for (int j = 0; j < D.enc[l].depth; ++j, ++i)
                if (i == want) { kind = 0; a = l; b = j; }
            if (l < nl - 1) { if (i == want) { kind = 1; a = l; b = 0; } ++i; }
        }
        for (int l = nl - 2; l >= 0; --l) {
            if (i == want) { kind = 2; a = l; b = 0; }
            ++i;
            for (int j = 0; j < D.dec[l].depth; ++j, ++i)
                if (i == want) { kind = 0; a = l; b = j | 256; }
        }
        return i;
    };
    int nops;
    {
        int k, a, b;
        nops = op_at(-1, k, a, b);
    }
    for (int op = 0; op < nops; ++op) {
        int kind = 0, a = 0, b = 0;
        op_at(op, kind, a, b);
        if (tr.tape) {  // (uniform) checkpoint for the backward: this op's input tokens
            const int lv = kind == 2 ? a + 1 : a;
            const int Xl = T(T_X + lv), nlv = T(T_N + lv), ldl = T(T_LD + lv), cl = D.enc[lv].c;
            float* tp = tr.tape + ((size_t)cloud * nops + op) * tr.tape_op_floats;
            for (int i = threadIdx.x; i < nlv * cl; i += kErwinThreads) tp[i] = F(Xl + 4 * ((i / cl) * ldl + i % cl));
        }
        if (kind == 0) {
            const haet_erwin_layer& L = (b & 256) ? D.dec[a] : D.enc[a];
            const int j = b & 255;
            const bool rt = D.rotate && L.depth > 1 && (j & 1);  // layers.py:66,75-89
            erwin_block(a, L, L.blocks[j], W, d, rt ? T(T_ROT + a) : ident);
        } else if (kind == 1) {
            erwin_pool(a, D, W, tr, cloud, n_clouds, epoch);
        } else {
            erwin_unpool(a, D, W, tr, cloud, n_clouds, epoch);
        }
        STAMP(3 + (op < 8 ? op : 8));
    }
    // node.x[tree_mask][argsort(tree_idx[tree_mask])]  (erwin_flash.py:193)
    for (int i = threadIdx.x; i < Mp * c0; i += kErwinThreads) {
        const int slot = i / c0, ch = i % c0;
        if (FUSED) {  // (power-of-two clouds: every slot holds a token)
            out[((size_t)cloud * Mp + I(tmap + 4 * slot)) * c0 + ch] = F(X0 + 4 * (slot * ld0 + ch));
        } else {
            const size_t g = (size_t)cloud * Mp + slot;
            if (tree_mask[g]) out[(size_t)tree_idx[g] * c0 + ch] = F(X0 + 4 * (slot * ld0 + ch));
        }
    }
    STAMP(12);
}

bool balltree_wave_ok(int M, int n_levels, const int* st);             // balltree_wave.hip
void rotation_matrix_f32(float angle_deg, int d, float* r9);             // balltree.hip

static int erwin_validate(const haet_erwin_desc* D, int Mp) {
    if (!D) return fail(HAET_E_ARG, "erwin: desc is NULL");
    if (D->n_levels < 1 || D->n_levels > HAET_ERWIN_MAX_LEVELS)
        return fail(HAET_E_ARG, "erwin: n_levels=%d unsupported", D->n_levels);
    if (!D->decode) return fail(HAET_E_ARG, "erwin: decode=False is not on the PhysicsAttention path");
    if (D->d != 2 && D->d != 3) return fail(HAET_E_ARG, "erwin: d=%d", D->d);
    int n = Mp;
    for (int l = 0; l < D->n_levels; ++l) {
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && l == D->n_levels - 1) break;
            const haet_erwin_layer& L = pass ? D->dec[l] : D->enc[l];
            if (L.depth < 0 || L.depth > HAET_ERWIN_MAX_DEPTH)
                return fail(HAET_E_ARG, "erwin: depth=%d unsupported", L.depth);
            if (L.c < 4 || L.c % 4) return fail(HAET_E_ARG, "erwin: channels=%d (multiple of 4)", L.c);
            if (L.heads < 1 || L.c % L.heads) return fail(HAET_E_ARG, "erwin: heads=%d", L.heads);
            const int hd = L.c / L.heads;
            if (hd != 1 && hd != 2 && hd != 4 && hd != 8 && hd != 16 && hd != 32 && hd != 64)
                return fail(HAET_E_ARG, "erwin: head_dim=%d unsupported", hd);
            if (L.ball < 1 || n % L.ball)
                return fail(HAET_E_ARG, "erwin: ball size %d does not divide %d tokens", L.ball, n);
            if (L.hidden < 1) return fail(HAET_E_ARG, "erwin: hidden=%d", L.hidden);
            for (int j = 0; j < L.depth; ++j)
                if (L.blocks[j].small_len < 1 || (L.blocks[j].small_len & 3) || (L.blocks[j].small_off & 3))
                    return fail(HAET_E_ARG, "erwin: block small-parameter group not packed (pack.py)");
        }
        if (l < D->n_levels - 1) {
            if (D->dec[l].c != D->enc[l].c) return fail(HAET_E_ARG, "erwin: decoder width mismatch");
            const int st = D->pool[l].stride;
            if (st < 1 || n % st || D->unpool[l].stride != st)
                return fail(HAET_E_ARG, "erwin: stride %d", st);
            if (st == 1 && D->enc[l + 1].c != D->enc[l].c)  // the reference would fail in the next layer's Linear
                return fail(HAET_E_ARG, "erwin: stride 1 (no pooling) needs equal widths %d / %d", D->enc[l].c, D->enc[l + 1].c);
            n /= st;
        }
    }
    return HAET_OK;
}

}  // namespace haet

using namespace haet;

#ifdef HAET_STAMPS
extern "C" int haet_debug_read_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(haet_stamps), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -1;
}
#endif

extern "C" size_t haet_erwin_lds_bytes(const haet_erwin_desc* desc, int Mp) {
    if (erwin_validate(desc, Mp) != HAET_OK) return 0;
    int a, w;
    return (size_t)erwin_layout(*desc, Mp, &a, &w) * sizeof(float);
}

static int erwin_train_args(const haet_erwin_desc* desc, const haet_erwin_train* train, int n_clouds,
                            hipStream_t stream, ErwinTrain* out) {
    ErwinTrain t = {};
    if (train && train->tape) {
        t.tape = train->tape;
        t.tape_op_floats = train->tape_op_floats;
        if (t.tape_op_floats < 1) return fail(HAET_E_ARG, "erwin: tape given without tape_op_floats");
    }
    if (!desc->bn_train) {
        *out = t;
        return HAET_OK;
    }
    if (!train || !train->workspace)
        return fail(HAET_E_ARG, "erwin: desc->bn_train is set but no train-mode plumbing (haet_erwin_train) was passed");
    if (n_clouds > resident_workgroup_limit())
        return fail(HAET_E_RESOURCE, "erwin: train-mode BatchNorm meets all %d clouds at a grid barrier: at most %d "
                                     "(one workgroup per compute unit of this device)", n_clouds, resident_workgroup_limit());
    int cmax = 4;
    for (int l = 0; l < desc->n_levels; ++l) cmax = cmax > desc->enc[l].c ? cmax : desc->enc[l].c;
    for (int i = 0; i < 2 * HAET_ERWIN_MAX_LEVELS; ++i) {
        t.rmean[i] = train->running_mean[i];
        t.rvar[i] = train->running_var[i];
        t.nbt[i] = train->num_batches_tracked[i];
        if ((t.rmean[i] == nullptr) != (t.rvar[i] == nullptr))
            return fail(HAET_E_ARG, "erwin: running_mean / running_var must be given in pairs");
    }
    t.stats = train->batch_stats;
    t.bar = reinterpret_cast<unsigned*>(train->workspace);
    t.tmo = t.bar + 1;
    t.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(train->workspace) + 64);
    t.cmax = cmax;
    t.on = 1;
    // the barrier COUNTER is zeroed before EVERY launch (by a kernel: a kernel node under graph capture); the time-out word next
    // to it is sticky: set by a barrier that gave up, cleared only by the host that reads it (haet_erwin_barrier_status)
    hipError_t e = zero_async(train->workspace, 4, stream);  // (a kernel, not hipMemsetAsync: common.h)
    if (e != hipSuccess) return fail(HAET_E_LAUNCH, "erwin: zeroing the barrier counter: %s", hipGetErrorString(e));
    *out = t;
    return HAET_OK;
}

extern "C" size_t haet_erwin_train_workspace_bytes(const haet_erwin_desc* desc, int n_clouds) {
    if (!desc || n_clouds < 1) return 0;
    int cmax = 4;
    for (int l = 0; l < desc->n_levels && l < HAET_ERWIN_MAX_LEVELS; ++l) cmax = cmax > desc->enc[l].c ? cmax : desc->enc[l].c;
    return 64 + (size_t)2 * HAET_ERWIN_MAX_LEVELS * n_clouds * 2 * cmax * sizeof(float);
}

template <bool FUSED>
static int erwin_launch(const float* tokens, const float* pos, int32_t* tree_idx, uint8_t* tree_mask, int32_t* rot,
                        const float* weights, size_t weight_floats, const haet_erwin_desc* desc,
                        const haet_erwin_train* train, int n_clouds, int M, int Mp, float* out, hipStream_t stream,
                        const WTreeCfg& tcfg, float* pos_out) {
    int a, w;
    const size_t bytes = (size_t)erwin_layout(*desc, Mp, &a, &w) * sizeof(float);
    const size_t statics = FUSED ? 8 * 1024 : 0;  // the tree builder's static tables (balltree_wave_dev.h)
    if (bytes + statics > 160 * 1024)
        return fail(HAET_E_RESOURCE, "erwin: one cloud needs %zu B of LDS (> 160 KiB)", bytes + statics);
    static std::atomic<size_t> attr_set[2][64];  // per kernel, per device ordinal (the attribute is per device)
    std::atomic<size_t>& attr_dev = attr_set[FUSED ? 1 : 0][PerDeviceOnce::bit()];
    if (bytes > 48 * 1024 && bytes > attr_dev.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(erwin_kernel<FUSED>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return fail(HAET_E_LAUNCH, "erwin: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_dev.store(bytes, std::memory_order_release);
    }
    ErwinTrain tr;
    if (int rc = erwin_train_args(desc, train, n_clouds, stream, &tr)) return rc;
    ErwinDir dir;
    erwin_directory(*desc, Mp, &dir);
    // HAET_ERWIN_XCD_PACK=1 (A/B switch, off by default: measured without effect): eval mode, few clouds -- pack them onto one XCD
    const char* pe = getenv("HAET_ERWIN_XCD_PACK");
    const bool pack = !tr.on && !tr.tape && n_clouds <= 32 && pe && pe[0] == '1';
    hipLaunchKernelGGL(erwin_kernel<FUSED>, dim3(pack ? 8 * n_clouds : n_clouds), dim3(kErwinThreads), bytes, stream, tokens, pos,
                       tree_idx, tree_mask, rot, weights, M, Mp, out, dir, tr, tcfg, pos_out, (unsigned)(weight_floats / 4),
                       pack ? n_clouds : 0);
    return check_launch(FUSED ? "erwin_kernel<fused token stage>" : "erwin_kernel");
}

extern "C" int haet_erwin_fwd(const float* tokens, const float* pos, const int32_t* tree_idx,
                              const uint8_t* tree_mask, const int32_t* rot, const float* weights,
                              size_t weight_floats, const haet_erwin_desc* desc, const haet_erwin_train* train,
                              int n_clouds, int M, float* out, haet_stream_t stream) {
    if (weight_floats * 4 < sizeof(haet_erwin_desc))
        return fail(HAET_E_ARG, "erwin: weight blob smaller than its descriptor header");
    if (!tokens || !pos || !tree_idx || !tree_mask || !weights || !out || n_clouds < 1 || M < 2)
        return fail(HAET_E_ARG, "erwin: null pointer or bad sizes");
    int depth = 0;
    while ((2 << depth) < M) ++depth;
    const int Mp = 2 << depth;
    int rc = erwin_validate(desc, Mp);
    if (rc != HAET_OK) return rc;
    if (desc->rotate && !rot) return fail(HAET_E_ARG, "erwin: rotate set but rot is NULL");
    WTreeCfg none = {};
    return erwin_launch<false>(tokens, pos, const_cast<int32_t*>(tree_idx), const_cast<uint8_t*>(tree_mask),
                               const_cast<int32_t*>(rot), weights, weight_floats, desc, train, n_clouds, M, Mp, out,
                               as_stream(stream), none, nullptr);
}

extern "C" int haet_token_stage_supported(const haet_erwin_desc* desc, int n_clouds, int M) {
    if (!desc || n_clouds < 1 || erwin_validate(desc, M) != HAET_OK) return 0;
    int st[3] = {1, 1, 1};
    for (int l = 0; l < desc->n_levels - 1 && l < 3; ++l) st[l] = desc->pool[l].stride;
    if (desc->n_levels > 4 || !balltree_wave_ok(M, desc->n_levels, st)) return 0;
    int a, w;
    return (size_t)erwin_layout(*desc, M, &a, &w) * sizeof(float) + 8 * 1024 <= 160 * 1024;
}

extern "C" int haet_token_stage_fwd(const float* S, int n_clouds, int M, float angle_deg, const float* weights,
                                    size_t weight_floats, const haet_erwin_desc* desc, const haet_erwin_train* train,
                                    float* pos_out, int32_t* tree_idx, uint8_t* tree_mask, int32_t* rot, float* out,
                                    haet_stream_t stream) {
    if (weight_floats * 4 < sizeof(haet_erwin_desc))
        return fail(HAET_E_ARG, "token_stage: weight blob smaller than its descriptor header");
    if (!S || !weights || !out || !desc) return fail(HAET_E_ARG, "token_stage: null pointer");
    if (!haet_token_stage_supported(desc, n_clouds, M))
        return fail(HAET_E_ARG, "token_stage: needs power-of-two clouds of <= 64 tokens (M=%d) and <= 4 levels; use "
                                "haet_tokens_tree_build + haet_erwin_fwd", M);
    if ((tree_idx == nullptr) != (tree_mask == nullptr)) return fail(HAET_E_ARG, "token_stage: tree_idx / tree_mask go together");
    WTreeCfg cfg = {};
    cfg.M = M;
    cfg.nd = desc->d;
    cfg.n_levels = desc->n_levels;
    cfg.rotate = desc->rotate && angle_deg > 0.f;
    if ((desc->rotate != 0) != (angle_deg > 0.f)) return fail(HAET_E_ARG, "token_stage: desc->rotate and angle disagree");
    cfg.s[0] = cfg.s[1] = cfg.s[2] = 1;
    int n = M;
    for (int l = 0; l < desc->n_levels; ++l) {
        // balltree.pyx:643: int(log2(TOTAL leaf count / ball size)) -- the total over all clouds (reference quirk)
        const double q = (double)n_clouds * M / (double)desc->enc[l].ball;
        int tg = 0;
        while ((double)(2ll << tg) <= q) ++tg;  // floor(log2 q) for q >= 1, 0 below
        cfg.t[l] = q >= 1.0 ? tg : 0;
        if (l < desc->n_levels - 1) {
            cfg.s[l] = desc->pool[l].stride;
            n /= cfg.s[l];
        }
    }
    (void)n;
    rotation_matrix_f32(angle_deg, desc->d, cfg.R.r);
    return erwin_launch<true>(S, nullptr, tree_idx, tree_mask, rot, weights, weight_floats, desc, train, n_clouds, M, M,
                              out, as_stream(stream), cfg, pos_out);
}
