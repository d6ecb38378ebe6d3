// Device code of the register-resident ball-tree builder (see balltree_wave.hip for the design): shared by the
// stand-alone tree kernel and the fused token stage (erwin.hip, which builds the tree in the prologue of the Erwin
// launch and never writes it to HBM on its own critical path).
#pragma once
#include "common.h"

namespace haet {

extern __shared__ __attribute__((aligned(16))) char smem[];  // the including kernel's dynamic LDS (sinks below)

__device__ __forceinline__ unsigned okey(float f) {
    // order-preserving float -> uint map (ties stay ties: -0 is folded into +0 first)
    const unsigned u = __float_as_uint(f + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct LaneArr {
    unsigned key;  // lane p: key of the element at position p
    int idx;       // lane p: element id at position p

    __device__ __forceinline__ unsigned k(int p) const {
        return (unsigned)__builtin_amdgcn_readlane((int)key, p);
    }
    __device__ __forceinline__ int i(int p) const { return __builtin_amdgcn_readlane(idx, p); }
    // single-lane write as compare + select (this clang has no v_writelane builtin)
    __device__ __forceinline__ void set(int p, unsigned kk, int ii) {
        const bool me = (int)(threadIdx.x & 63) == p;
        key = me ? kk : key;
        idx = me ? ii : idx;
    }
    __device__ __forceinline__ void swap(int p, int q) {
        const unsigned kp = k(p), kq = k(q);
        const int ip = i(p), iq = i(q);
        set(p, kq, iq);
        set(q, kp, ip);
    }
};

// ---- heap fallback of introselect (depth budget spent) -- generic single-lane accesses
__device__ __forceinline__ void w_adjust_heap(LaneArr& a, int first, int hole, int len, unsigned vk, int vi) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (a.k(first + child) < a.k(first + child - 1)) child--;
        a.set(first + hole, a.k(first + child), a.i(first + child));
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a.set(first + hole, a.k(first + child - 1), a.i(first + child - 1));
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && a.k(first + parent) < vk) {
        a.set(first + hole, a.k(first + parent), a.i(first + parent));
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a.set(first + hole, vk, vi);
}

__device__ __forceinline__ void w_heap_select(LaneArr& a, int first, int middle, int last) {
    const int len = middle - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) {
            w_adjust_heap(a, first, parent, len, a.k(first + parent), a.i(first + parent));
            if (parent == 0) break;
            parent--;
        }
    }
    for (int p = middle; p < last; ++p) {
        if (a.k(p) < a.k(first)) {
            const unsigned vk = a.k(p);
            const int vi = a.i(p);
            a.set(p, a.k(first), a.i(first));
            w_adjust_heap(a, first, 0, len, vk, vi);
        }
    }
}

__device__ __forceinline__ void w_insertion_sort(LaneArr& a, int first, int last) {
    if (first == last) return;
    for (int p = first + 1; p != last; ++p) {
        const unsigned vk = a.k(p);
        const int vi = a.i(p);
        if (vk < a.k(first)) {
            for (int q = p; q != first; --q) a.set(q, a.k(q - 1), a.i(q - 1));
            a.set(first, vk, vi);
        } else {
            int hole = p, prev = p - 1;
            while (vk < a.k(prev)) {
                a.set(hole, a.k(prev), a.i(prev));
                hole = prev;
                --prev;
            }
            a.set(hole, vk, vi);
        }
    }
}

// std::__unguarded_partition(first, last, pivot) of libstdc++ in ONE step instead of a swap loop.
// The sequential scans stop at the "left stoppers" L1 < L2 < ... (positions with !(key < pivot)) and
// the "right stoppers" R1 > R2 > ... (positions with !(pivot < key)); the swapped elements land behind
// the cursors and are never looked at again, so the k-th swap exchanges Lk and Rk of the ORIGINAL
// array for every k with Lk < Rk (k = 1..K), and the loop returns min(L(K+1), RK) -- the next left
// stopper if one lies below RK, else RK itself (which now holds a stopper).  Every lane ranks itself
// among the stoppers with two ballots and popcounts, the K pairs find each other through a small
// per-wave LDS table, and all swaps are two cross-lane reads.
__shared__ int haet_wtab[8 * 2 * 64];
__shared__ int haet_wex[4 * 2 * 64];  // level exchange buffers [group][parity][lane] (by index: no generic pointers)
__device__ __forceinline__ int w_partition(LaneArr& a, int lo, int hi, unsigned pivot) {
    typedef unsigned long long u64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u64 range = (hi >= 64 ? ~0ull : ((1ull << hi) - 1)) & (~0ull << lo);
    const u64 Lm = ~__ballot(a.key < pivot) & range, Rm = ~__ballot(pivot < a.key) & range;
    const u64 below = (1ull << lane) - 1, above = lane == 63 ? 0ull : (~0ull << (lane + 1));
    const int n_above = __popcll(Rm & above);
    const int ra = __popcll(Lm & below) + 1;  // my rank among the left stoppers (if I am one)
    const int rb = n_above + 1;               // my rank among the right stoppers, from the top
    const bool swap_l = ((Lm >> lane) & 1) && n_above >= ra;  // L(ra) < R(ra)
    const u64 SL = __ballot(swap_l);
    const int K = __popcll(SL);
    const bool swap_r = ((Rm >> lane) & 1) && rb <= K;
    const u64 SR = __ballot(swap_r);
    if (K) {
        const int tl = wave * 128;  // [tl .. tl+63]: lane of L(k), [tl+64 .. tl+127]: lane of R(k)
        if (swap_l) haet_wtab[tl + ra - 1] = lane;
        if (swap_r) haet_wtab[tl + 64 + rb - 1] = lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        int partner = lane;
        if (swap_l) partner = haet_wtab[tl + 64 + ra - 1];
        if (swap_r) partner = haet_wtab[tl + rb - 1];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        a.key = (unsigned)__shfl((int)a.key, partner);
        a.idx = __shfl(a.idx, partner);
    }
    const u64 rest = Lm & ~SL;  // left stoppers that did not swap: the lowest one is L(K+1)
    const int next_l = rest ? __builtin_ctzll(rest) : 64;
    const int r_k = K ? __builtin_ctzll(SR) : 64;
    return next_l < r_k ? next_l : r_k;
}

// introselect on positions [first, last), all scalars wave-uniform
__device__ __forceinline__ void w_select_nth(LaneArr& a, int first, int nth, int last) {
    if (first == last || nth == last) return;
    int depth = 2 * (31 - __clz(last - first));
    while (last - first > 3) {
        if (depth == 0) {
            w_heap_select(a, first, nth + 1, last);
            a.swap(first, nth);
            return;
        }
        --depth;
        const int pa = first + 1, pb = first + (last - first) / 2, pc = last - 1;
        const unsigned ka = a.k(pa), kb = a.k(pb), kc = a.k(pc);
        int med;  // median of three goes to `first`
        if (ka < kb) med = (kb < kc) ? pb : ((ka < kc) ? pc : pa);
        else med = (ka < kc) ? pa : ((kb < kc) ? pc : pb);
        a.swap(first, med);
        const unsigned pivot = a.k(first);
        const int lo = w_partition(a, first + 1, last, pivot);
        if (lo <= nth) first = lo;
        else last = lo;
    }
    w_insertion_sort(a, first, last);
}

template <int CTRL>
__device__ __forceinline__ float w_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

// Split dimension and sort keys of one tree level: all nodes have `sz` points (power of two,
// lane-aligned segments).  cx,cy,cz: coordinates BY ELEMENT (lane e = element e).  Pure lane-parallel.
__device__ __forceinline__ void w_level_keys(LaneArr& a, float cx, float cy, float cz, int nd, int sz) {
    // coordinates of the element sitting at my position
    float v[3] = {__shfl(cx, a.idx), __shfl(cy, a.idx), nd > 2 ? __shfl(cz, a.idx) : 0.f};
    int best = 0;
    double best_spread = 0.0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (d < nd) {
            // min / max over the aligned segment of sz lanes: xor-butterfly; the steps inside a row of
            // 16 lanes are DPP moves (xor 1, xor 2, xor 7 = row_half_mirror, xor 15 = row_mirror: these
            // masks generate the same groups as 1, 2, 4, 8), only 16 and 32 go through the LDS crossbar
            float lo = v[d], hi = v[d];
            if (sz >= 2) {
                lo = fminf(lo, w_dpp<0xB1>(lo));
                hi = fmaxf(hi, w_dpp<0xB1>(hi));
            }
            if (sz >= 4) {
                lo = fminf(lo, w_dpp<0x4E>(lo));
                hi = fmaxf(hi, w_dpp<0x4E>(hi));
            }
            if (sz >= 8) {
                lo = fminf(lo, w_dpp<0x141>(lo));
                hi = fmaxf(hi, w_dpp<0x141>(hi));
            }
            if (sz >= 16) {
                lo = fminf(lo, w_dpp<0x140>(lo));
                hi = fmaxf(hi, w_dpp<0x140>(hi));
            }
            for (int off = 16; off < sz; off <<= 1) {
                lo = fminf(lo, __shfl_xor(lo, off));
                hi = fmaxf(hi, __shfl_xor(hi, off));
            }
            const double spread = (double)hi - (double)lo;
            if (spread > best_spread) {  // strict: ties keep the lowest dimension (balltree.h:58)
                best_spread = spread;
                best = d;
            }
        }
    }
    a.key = okey(best == 0 ? v[0] : (best == 1 ? v[1] : v[2]));
}

struct Rot3w {
    float r[9];
};

constexpr int TW = 8;  // waves per cloud

#ifdef HAET_STAMPS
// Diagnostic build only (tools/tree_stamps.py): s_memtime of wave 0 of cloud 0 along the build.
__device__ unsigned long long haet_tree_stamps[32];
#define TSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) haet_tree_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif

// Workgroup barrier of the builder: waits for this wave's LDS traffic only.  __syncthreads() also drains the vector-memory
// counter, and the fused callers keep global loads in flight under the builder (the L2 warm-up of the Erwin weights,
// one slice per level): every level then ended with a wait for HBM.  Everything the builder's waves exchange goes
// through LDS; its global stores (tree_idx, rot, pos_out) are outputs nobody in the launch reads back.
__device__ __forceinline__ void w_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One tree level shared by the `nw` waves w0 .. w0+nw-1, which all hold the same lane arrays: the
// nodes of a level are independent, so wave i runs the (sequential) introselect of nodes i, i+nw, ...
// and the waves swap their results through `ex` (double-buffered by `par`: one barrier per level).
// EVERY wave of the workgroup calls this the same number of times (the barrier is the workgroup's);
// `work` = false just keeps the barrier count.
__device__ __forceinline__ void w_shared_level(LaneArr& a, float cx, float cy, float cz, int nd, int n, int sz,
                                               bool work, int me, int nw, int ex, int sb = -1) {  // ex: index into haet_wex
    const int lane = threadIdx.x & 63;
    if (work) {
        w_level_keys(a, cx, cy, cz, nd, sz);
        if (sb >= 0) TSTAMP(sb);
        const int nodes = n / sz;
        for (int j = me; j < nodes; j += nw) w_select_nth(a, j * sz, j * sz + sz / 2, j * sz + sz);
        if (sb >= 0) TSTAMP(sb + 1);
        if (lane < n && (lane / sz) % nw == me) haet_wex[ex + lane] = a.idx;
    }
    w_barrier();
    if (work && lane < n) a.idx = haet_wex[ex + lane];
}


// LDS sinks of the fused caller (byte offsets into the dynamic LDS array `smem`; < 0: none): the tree never has to
// leave the CU between the builder and the Erwin blocks.
struct WTreeLds {
    int tree;    // int[M]      element (token of this cloud) at tree slot p
    int pos;     // float[M][d] positions in TREE order
    int rot[4];  // int[n_l]    rotated permutation of hierarchy level l, LOCAL indices (positions at that level)
};

struct WTreeCfg {
    int M, nd, n_levels, s[3], t[4], rotate;
    Rot3w R;
};

// The whole build for one cloud by the TW waves of the workgroup (every wave must call it: workgroup barriers inside).
// Inputs: `S` [n_clouds*M, C] tokens (positions = min-max normalised first nd channels over ALL tokens,
// IrregularMesh.py:204-209, written to `pos_out`) or, with S == nullptr, `pos` [n_clouds*M, nd].
// `step(i)` is called once per build step (position pass, every tree level) before its work starts: the fused kernel
// uses it to warm the L2 with the Erwin weights a slice at a time under the builder's latency.
template <typename StepFn>
__device__ __forceinline__ void w_tree_build(const WTreeCfg& cfg, const float* __restrict__ pos,
                                             const float* __restrict__ S, int C, int cloud, int n_clouds,
                                             int32_t* __restrict__ tree_idx, uint8_t* __restrict__ tree_mask,
                                             int32_t* __restrict__ rot, float* __restrict__ pos_out,
                                             const WTreeLds& sink, StepFn step) {
    __shared__ float mm[TW][2][3];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int M = cfg.M, nd = cfg.nd, n_levels = cfg.n_levels;
    int istep = 0;
    TSTAMP(0);
    step(istep++);
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (S) {
        // Fused haet_token_pos (IrregularMesh.py:204-209): every cloud's workgroup reduces the global
        // per-channel min / max over ALL tokens itself (T * d floats: a few KB), normalises its own
        // tokens with the formula of token_pos_kernel (bit-identical) and writes them for the Erwin stage.
        const int T = n_clouds * M;
        // one 16-B load per token covers all position channels (rows are 16-B aligned: C % 4 == 0); the wave's own
        // token rides along, so the whole pass is ONE global round trip (it was one per dimension plus one)
        typedef __attribute__((ext_vector_type(4))) float f4;
        typedef const __attribute__((address_space(1))) f4* gf4p;
        f4 mine = {0.f, 0.f, 0.f, 0.f};
        float lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
        const size_t own = ((size_t)cloud * M + (lane < M ? lane : 0)) * C;
        if ((C & 3) == 0) {  // (workgroup-uniform)
            mine = *(gf4p)(S + own);
            for (int t = threadIdx.x; t < T; t += 64 * TW) {
                const f4 v = *(gf4p)(S + (size_t)t * C);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    lo3[j] = fminf(lo3[j], v[j]);
                    hi3[j] = fmaxf(hi3[j], v[j]);
                }
            }
        } else {  // odd widths: scalar loads of the nd position channels
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j < nd) {
                    mine[j] = S[own + j];
                    for (int t = threadIdx.x; t < T; t += 64 * TW) {
                        const float v = S[(size_t)t * C + j];
                        lo3[j] = fminf(lo3[j], v);
                        hi3[j] = fmaxf(hi3[j], v);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float lo = wave_min(lo3[j]), hi = wave_max(hi3[j]);
            if (lane == 0) {
                mm[wave][0][j] = lo;
                mm[wave][1][j] = hi;
            }
        }
        w_barrier();
        float c3[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j < nd) {
                float lo = mm[0][0][j], hi = mm[0][1][j];
                for (int w = 1; w < TW; ++w) {
                    lo = fminf(lo, mm[w][0][j]);
                    hi = fmaxf(hi, mm[w][1][j]);
                }
                const float rng = hi - lo + 1e-8f;
                if (lane < M) {
                    c3[j] = (mine[j] - lo) / rng;
                    if (wave == 0 && pos_out) pos_out[((size_t)cloud * M + lane) * nd + j] = c3[j];
                }
            }
        }
        cx = c3[0];
        cy = c3[1];
        cz = c3[2];
    } else if (lane < M) {
        const float* p = pos + ((size_t)cloud * M + lane) * nd;
        cx = p[0];
        cy = p[1];
        if (nd > 2) cz = p[2];
    }
    TSTAMP(1);
    LaneArr a;
    a.idx = lane;
    a.key = 0;
    // main tree (balltree.h:121-186): halve until the nodes hold 2 points; all TW waves share a level
    int par = 0;
    {
        int lv = 0;
        for (int sz = M; sz > 2; sz >>= 1, par ^= 1) {
            step(istep++);
            w_shared_level(a, cx, cy, cz, nd, M, sz, true, wave, TW, par * 64, 16 + 2 * lv);
            TSTAMP(2 + lv);
            ++lv;
            (void)lv;
        }
    }
    // leaves in tree order
    const float lx = __shfl(cx, a.idx), ly = __shfl(cy, a.idx), lz = __shfl(cz, a.idx);
    if (wave == 0 && lane < M) {
        if (tree_idx) {
            tree_idx[(size_t)cloud * M + lane] = cloud * M + a.idx;
            tree_mask[(size_t)cloud * M + lane] = 1;
        }
        if (sink.tree >= 0) *reinterpret_cast<int*>(smem + sink.tree + 4 * lane) = a.idx;
        if (sink.pos >= 0) {
            *reinterpret_cast<float*>(smem + sink.pos + 4 * (lane * nd + 0)) = lx;
            *reinterpret_cast<float*>(smem + sink.pos + 4 * (lane * nd + 1)) = ly;
            if (nd > 2) *reinterpret_cast<float*>(smem + sink.pos + 4 * (lane * nd + 2)) = lz;
        }
    }
    if (!cfg.rotate) return;  // (workgroup-uniform)
    // rotated leaves (balltree.pyx:639-646)
    const Rot3w& R = cfg.R;
    float rx, ry, rz = 0.f;
    if (nd == 2) {
        rx = fmaf(ly, R.r[2], lx * R.r[0]);
        ry = fmaf(ly, R.r[3], lx * R.r[1]);
    } else {
        rx = fmaf(lz, R.r[6], fmaf(ly, R.r[3], lx * R.r[0]));
        ry = fmaf(lz, R.r[7], fmaf(ly, R.r[4], lx * R.r[1]));
        rz = fmaf(lz, R.r[8], fmaf(ly, R.r[5], lx * R.r[2]));
    }
    // The re-partitions of the hierarchy levels depend only on the (pooled) rotated leaves, not on
    // each other: wave group g builds level g, all levels at once.
    const int G = n_levels <= 1 ? 1 : (n_levels <= 2 ? 2 : 4), gsize = TW / G;
    const int grp = wave / gsize, me = wave % gsize;
    const bool mine = grp < n_levels;
    int n = M;
    size_t rot_off = 0;
    for (int lev = 0; lev < grp && lev < n_levels - 1; ++lev) {  // pool down to my level
        const int st = cfg.s[lev];
        float sx = 0.f, sy = 0.f, sz2 = 0.f;
        for (int q = 0; q < st; ++q) {  // sequential sum like .mean(axis=1)
            const int src = (lane * st + q) & 63;
            const float gx = __shfl(rx, src), gy = __shfl(ry, src), gz = __shfl(rz, src);
            sx = q ? sx + gx : gx;
            sy = q ? sy + gy : gy;
            sz2 = q ? sz2 + gz : gz;
        }
        rx = sx / (float)st;
        ry = sy / (float)st;
        rz = sz2 / (float)st;
        rot_off += (size_t)n_clouds * n;
        n /= st;
    }
    int max_splits = 0;  // same for every wave: the deepest level sets the number of barriers
    {
        int nn = M;
        for (int lev = 0; lev < n_levels; ++lev) {
            int c = 0;
            for (int sz = nn; c < cfg.t[lev] && sz > 1; sz >>= 1) ++c;
            max_splits = max(max_splits, c);
            if (lev < n_levels - 1) nn /= cfg.s[lev];
        }
    }
    TSTAMP(10);
    a.idx = lane;
    int sz = n;
    for (int it = 0; it < max_splits; ++it, par ^= 1) {  // `par` carries on: never the buffer just read
        step(istep++);
        const bool work = mine && it < cfg.t[mine ? grp : 0] && sz > 1;
        w_shared_level(a, rx, ry, rz, nd, n, sz, work, me, gsize, ((grp & 3) * 2 + par) * 64);
        if (work) sz >>= 1;
        TSTAMP(11 + it);
    }
    if (mine && me == 0 && lane < n) {
        if (rot) rot[rot_off + (size_t)cloud * n + lane] = cloud * n + a.idx;
        if (sink.rot[grp & 3] >= 0) *reinterpret_cast<int*>(smem + sink.rot[grp & 3] + 4 * lane) = a.idx;
    }
}

}  // namespace haet
