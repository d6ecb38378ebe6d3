// Ball tree of RAW point clouds (32k .. 1M+ points per cloud, clouds of unequal size) -- the tree the stand-alone
// ErwinFlash model builds on the host every forward (models/components/erwinflash/erwin_flash.py:149-157 ->
// models/components/balltree/balltree.pyx:542-662; benchmarks/02.1-Car-Design-ShapeNetCar/main.py:50-99).
//
// The permutation has to be the one libstdc++'s std::nth_element leaves behind (balltree.h:91-100): BallPooling
// concatenates the two points of a leaf in that order (pooling.py:35-38), and the order a node is left in is the input
// of its children's selects.  So the introselect is followed step for step, but its one O(n)
// part -- the unguarded Hoare partition around the median-of-three pivot -- runs data-parallel:
//   the k-th swap of the sequential loop exchanges the k-th "left stopper" (k-th position from the left whose key is
//   not < pivot) with the k-th "right stopper" (k-th position from the right whose key is not > pivot; the pivot's own
//   slot in front of the range is the last of them), for as long as the former lies left of the latter.  Both
//   sequences are prefix counts of the ORIGINAL array, the number of swaps K is where the monotone predicate
//   L_k < R_k flips, and the cut the loop returns is min(L_{K+1}, R_K).
// Levels whose nodes are larger than kSubT items run one workgroup per node on global memory (a launch per level);
// every node of at most kSubT items is finished -- all remaining levels -- by one workgroup in LDS: nodes above 96 items
// by a wave each (the same rounds with ballots), smaller ones by a lane each with the serial select of
// balltree_select_dev.h (the nodes of a level are independent).
// Keys are the fp32 coordinates (the reference's float64 conversion is exact and order preserving); spreads are
// differenced in double like the reference.
#include "common.h"
#include "balltree_select_dev.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace haet {

void rotation_matrix_f32(float angle_deg, int d, float* r);  // balltree.hip

constexpr int kSubT = 1024;       // nodes of at most this many items are finished inside one wave's LDS
constexpr int kTopThreads = 1024;
constexpr int kTopWaves = kTopThreads / 64;

struct BigCloud {
    int off;     // first item of the cloud
    int n;       // items
    int levels;  // split levels
    int top;     // levels [0, top) run as grid launches; at level `top` every node has <= kSubT items
};

// ----------------------------------------------------------------------------------------------- top levels
// INLDS: the node's frame (working array, keys, stopper lists) lives in LDS -- nodes of at most kTopLds items; a round is
// then a handful of LDS round trips instead of global-memory ones (a level launch: ~90 us -> ~15 us).
constexpr int kTopLds = 8192;
extern __shared__ __attribute__((aligned(16))) char top_lds[];
template <bool INLDS>
__global__ __launch_bounds__(kTopThreads) void big_top_kernel(
    const float* __restrict__ keys, int d, const BigCloud* __restrict__ tab, const int2* __restrict__ work, int level,
    int* __restrict__ idx, int* __restrict__ gid, float* __restrict__ knode, int* __restrict__ w,
    int* __restrict__ LS, int* __restrict__ RS, float* __restrict__ kpos) {
    __shared__ float sh_lo[3][kTopWaves], sh_hi[3][kTopWaves];
    __shared__ int sh_cl[kTopWaves], sh_cr[kTopWaves], sh_k[kTopWaves];
    __shared__ int sh_dim;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int2 wk = work[blockIdx.x];
    const BigCloud C = tab[wk.x];
    int s, e;
    node_range(C.n, level, wk.y, s, e);
    const int m = e - s;
    if (m <= 1) return;
    const size_t base = (size_t)C.off + s;
    int* const W = INLDS ? reinterpret_cast<int*>(top_lds) : w + base;
    float* const KD = INLDS ? reinterpret_cast<float*>(top_lds + 4 * kTopLds) : knode + base;
    int* const ls = INLDS ? reinterpret_cast<int*>(top_lds + 8 * kTopLds) : LS + base;
    int* const rs = INLDS ? reinterpret_cast<int*>(top_lds + 12 * kTopLds + 16) : RS + base;
    // Global tier: the keys are ALSO kept in position order (KP[p] = key of the entry at p, swapped along with W): the
    // sweeps of a round then stream KP instead of gathering KD[W[p]] -- one dependent round trip less per sweep, whole
    // lines used.  (KD by entry stays for the serial fallbacks; in LDS the gather costs nothing.)
    float* const KP = INLDS ? nullptr : kpos + base;

    // split dimension: largest max - min (first one on ties), balltree.h:32-66.  The gathers (entry -> point -> key) are
    // issued kU positions at a time: one position per loop trip is two dependent memory round trips per trip, and the
    // three sweeps of the node (this one, the frame below, the write-back) were most of a level's run time.
    constexpr int kU = 8;
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) lo[k] = INFINITY, hi[k] = -INFINITY;
    int g8[kU];        // INLDS (<= kU positions per thread): entries and keys stay in registers for the frame
    float k8[kU][3];
    for (int p0 = tid; p0 < m; p0 += kU * kTopThreads) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int p = p0 + u * kTopThreads;
            g8[u] = idx[base + (p < m ? p : 0)];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) k8[u][k] = k < d ? keys[((size_t)C.off + g8[u]) * d + k] : 0.f;
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (p0 + u * kTopThreads < m)
#pragma unroll
                for (int k = 0; k < 3; ++k) lo[k] = fminf(lo[k], k8[u][k]), hi[k] = fmaxf(hi[k], k8[u][k]);
    }
    for (int k = 0; k < d; ++k) {
        lo[k] = wave_min(lo[k]);
        hi[k] = wave_max(hi[k]);
        if (lane == 0) sh_lo[k][wave] = lo[k], sh_hi[k][wave] = hi[k];
    }
    __syncthreads();
    if (tid == 0) {
        int best = 0;
        double best_spread = 0.0;
        for (int k = 0; k < d; ++k) {
            float a = sh_lo[k][0], b = sh_hi[k][0];
            for (int q = 1; q < kTopWaves; ++q) a = fminf(a, sh_lo[k][q]), b = fmaxf(b, sh_hi[k][q]);
            const double spread = (double)b - (double)a;
            if (spread > best_spread) best_spread = spread, best = k;
        }
        sh_dim = best;
    }
    __syncthreads();
    const int dim = sh_dim;
    // the node's own frame: keys by ENTRY position, working array of entry positions
    if (INLDS) {
        static_assert(kTopLds <= kU * kTopThreads, "one sweep of kU positions per thread covers an LDS-tier node");
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int p = tid + u * kTopThreads;
            if (p < m) {
                gid[base + p] = g8[u];
                KD[p] = dim == 0 ? k8[u][0] : (dim == 1 ? k8[u][1] : k8[u][2]);
                W[p] = p;
            }
        }
    } else {
        for (int p0 = tid; p0 < m; p0 += kU * kTopThreads) {
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int p = p0 + u * kTopThreads;
                g8[u] = idx[base + (p < m ? p : 0)];
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) k8[u][0] = keys[((size_t)C.off + g8[u]) * d + dim];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int p = p0 + u * kTopThreads;
                if (p < m) {
                    gid[base + p] = g8[u];
                    KD[p] = k8[u][0];
                    KP[p] = k8[u][0];
                    W[p] = p;
                }
            }
        }
    }
    __syncthreads();

    int first = 0, last = m;
    const int nth = m / 2;  // (s + e) / 2 - s
    int depth = 2 * (31 - __clz(m));
    while (last - first > 3) {
        if (depth == 0) {  // introselect's budget is spent: heap select, one lane (adversarial inputs only)
            if (tid == 0) {
                heap_select_(W + first, W + nth + 1, W + last, KD);
                swp(W + first, W + nth);
            }
            first = last;  // done
            break;
        }
        --depth;
        // median of (first+1, mid, last-1) to the front: every thread reads the three candidates and knows the pivot
        // without waiting for lane 0's swap to come back through memory
        float v;
        int med_pos, med_val, front_old;
        float front_key = 0.f;
        {
            int *a = W + first + 1, *b = W + first + (last - first) / 2, *c = W + last - 1, *f = W + first;
            const int ia = *a, ib = *b, ic = *c;
            const float ka = INLDS ? KD[ia] : KP[first + 1], kb = INLDS ? KD[ib] : KP[first + (last - first) / 2],
                        kc = INLDS ? KD[ic] : KP[last - 1];
            int* med;  // the position whose entry goes to the front (__move_median_to_first)
            if (ka < kb) med = kb < kc ? b : (ka < kc ? c : a);
            else med = ka < kc ? a : (kb < kc ? c : b);
            v = med == a ? ka : (med == b ? kb : kc);
            med_pos = (int)(med - W);
            med_val = med == a ? ia : (med == b ? ib : ic);
            front_old = *f;  // the entry that moves to `med_pos`: readers of that slot substitute it until the swap lands
            if (!INLDS) front_key = KP[first];
        }
        const int f1 = first + 1, len = last - f1;
        // every wave owns a contiguous chunk (multiple of 64 positions) of [f1, last)
        const int chunk = ((len + kTopWaves - 1) / kTopWaves + 63) & ~63;
        const int c0 = f1 + wave * chunk, c1 = min(c0 + chunk, last);
        // pass 1: count the stoppers; their flags stay in registers (one bit per 64-position step, up to 64 steps =
        // 4096 positions per wave = nodes of 64k items; larger nodes gather twice)
        const bool cached = chunk <= 64 * 64;
        unsigned long long fl = 0ull, fr = 0ull;
        int nl = 0, nr = 0;
        if (INLDS) {
            int step = 0;
            for (int p0 = c0; p0 < c1; p0 += 64, ++step) {
                const int p = p0 + lane;
                const float x = p < c1 ? KD[p == med_pos ? front_old : W[p]] : 0.f;
                const bool isl = p < c1 && !(x < v), isr = p < c1 && !(v < x);
                nl += __popcll(__ballot(isl));
                nr += __popcll(__ballot(isr));
                if (cached) {
                    fl |= (unsigned long long)isl << step;
                    fr |= (unsigned long long)isr << step;
                }
            }
        } else {
            // global tier: eight 64-position steps per trip, their loads issued together (a wave of a 2M-item node walks
            // 2048 steps per sweep: one memory round trip per step is what a level launch cost there)
            int step = 0;
            for (int p0 = c0; p0 < c1; p0 += 8 * 64, step += 8) {
                float xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int p = p0 + 64 * u + lane;
                    xv[u] = KP[p < c1 ? p : c0];
                    if (p == med_pos) xv[u] = front_key;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int p = p0 + 64 * u + lane;
                    const bool isl = p < c1 && !(xv[u] < v), isr = p < c1 && !(v < xv[u]);
                    nl += __popcll(__ballot(isl));
                    nr += __popcll(__ballot(isr));
                    if (cached) {
                        fl |= (unsigned long long)isl << (step + u);
                        fr |= (unsigned long long)isr << (step + u);
                    }
                }
            }
        }
        if (lane == 0) sh_cl[wave] = nl, sh_cr[wave] = nr;
        __syncthreads();
        // (every thread has read the candidates and its positions: the swap of the median may land now; it is ordered
        // before the swaps of this round by the two barriers below)
        if (tid == 0) {
            W[med_pos] = front_old;
            W[first] = med_val;
            if (!INLDS) KP[med_pos] = front_key, KP[first] = v;
        }
        int bl = 0, br = 0, nL = 0, nR = 0;
        for (int q = 0; q < kTopWaves; ++q) {
            if (q < wave) bl += sh_cl[q], br += sh_cr[q];
            nL += sh_cl[q], nR += sh_cr[q];
        }
        // stopper lists: ls[k] = k-th left stopper from the left, rs[k] = k-th right stopper from the right
        // (the pivot's slot `first` is the last right stopper)
        {
            int step = 0;
            for (int p0 = c0; p0 < c1; p0 += 64, ++step) {
                const int p = p0 + lane;
                bool isl, isr;
                if (cached) {
                    isl = (fl >> step) & 1ull;
                    isr = (fr >> step) & 1ull;
                } else {
                    const float x = p < c1 ? (INLDS ? KD[p == med_pos ? front_old : W[p]] : (p == med_pos ? front_key : KP[p])) : 0.f;
                    isl = p < c1 && !(x < v);
                    isr = p < c1 && !(v < x);
                }  // (nodes above 64k items only: the flags of pass 1 did not fit the two register words)
                const unsigned long long ml = __ballot(isl), mr = __ballot(isr);
                const unsigned long long below = (1ull << lane) - 1ull;
                if (isl) ls[bl + __popcll(ml & below)] = p;
                if (isr) rs[nR - 1 - (br + __popcll(mr & below))] = p;
                bl += __popcll(ml);
                br += __popcll(mr);
            }
        }
        if (tid == 0) rs[nR] = first;
        __syncthreads();
        const int kmax = min(nL, nR + 1);
        int kc = 0;
        if (INLDS) {
            for (int k = tid; k < kmax; k += kTopThreads) kc += ls[k] < rs[k] ? 1 : 0;
        } else {
            for (int k0 = tid; k0 < kmax; k0 += 4 * kTopThreads) {
                int a4[4], b4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + u * kTopThreads;
                    a4[u] = ls[k < kmax ? k : 0];
                    b4[u] = rs[k < kmax ? k : 0];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) kc += (k0 + u * kTopThreads < kmax && a4[u] < b4[u]) ? 1 : 0;
            }
        }
        kc = (int)wave_sum((float)kc);  // (< 2^24 per wave: exact)
        if (lane == 0) sh_k[wave] = kc;
        __syncthreads();
        int K = 0;
        for (int q = 0; q < kTopWaves; ++q) K += sh_k[q];
        int cut;
        {
            const int bound = K > 0 ? rs[K - 1] : last;
            cut = bound;
            if (K < nL) {
                const int Lk = ls[K];
                cut = Lk < bound ? Lk : bound;
            }
        }
        if (INLDS) {
            for (int k = tid; k < K; k += kTopThreads) {
                const int a = ls[k], b = rs[k];
                const int t = W[a];
                W[a] = W[b];
                W[b] = t;
            }
        } else {
            for (int k0 = tid; k0 < K; k0 += 4 * kTopThreads) {
                int a4[4], b4[4], wa[4], wb[4];
                float ka4[4], kb4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + u * kTopThreads;
                    a4[u] = ls[k < K ? k : 0];
                    b4[u] = rs[k < K ? k : 0];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) wa[u] = W[a4[u]], wb[u] = W[b4[u]], ka4[u] = KP[a4[u]], kb4[u] = KP[b4[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k0 + u * kTopThreads < K) W[a4[u]] = wb[u], W[b4[u]] = wa[u], KP[a4[u]] = kb4[u], KP[b4[u]] = ka4[u];
            }
        }
        __syncthreads();
        if (cut <= nth) first = cut;
        else last = cut;
    }
    if (tid == 0 && last - first > 0) insertion_sort_(W + first, W + last, KD);
    __syncthreads();
    for (int p0 = tid; p0 < m; p0 += kU * kTopThreads) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int p = p0 + u * kTopThreads;
            g8[u] = W[p < m ? p : 0];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) g8[u] = gid[base + g8[u]];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int p = p0 + u * kTopThreads;
            if (p < m) idx[base + p] = g8[u];
        }
    }
}

// ----------------------------------------------------------------------------------------------- sub-trees
// wave_sync / wave_select / wave_split_dim: balltree_select_dev.h (shared with balltree.hip)
constexpr int kCoop = 96;   // nodes larger than this are selected by a whole wave, smaller ones by one lane each
constexpr int kSubWaves = 8;  // waves of a sub-tree workgroup: the nodes of a level are dealt to them
__global__ __launch_bounds__(64 * kSubWaves) void big_sub_kernel(const float* __restrict__ keys, int d,
                                                                 const BigCloud* __restrict__ tab,
                                                                 const int2* __restrict__ work, int* __restrict__ idx) {
    __shared__ float lkey[3 * kSubT];
    __shared__ int lidx[kSubT];
    __shared__ int lgid[kSubT];
    __shared__ int ls[kSubT];   // scratch of wave_select: a node uses the slice of its own range
    __shared__ int rsl[kSubT];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int2 wk = work[blockIdx.x];
    const BigCloud C = tab[wk.x];
    int s, e;
    node_range(C.n, C.top, wk.y, s, e);
    const int m = e - s, r = C.levels - C.top;
    if (m <= 1 || r <= 0) return;
    const size_t base = (size_t)C.off + s;
    for (int i = tid; i < m; i += 64 * kSubWaves) {
        const int g = idx[base + i];
        lgid[i] = g;
        lidx[i] = i;
        for (int k = 0; k < d; ++k) lkey[k * kSubT + i] = keys[((size_t)C.off + g) * d + k];
    }
    __syncthreads();
    for (int L = 0; L < r; ++L) {
        if (((m + (1 << L) - 1) >> L) > kCoop) {  // (an upper bound of every node size of this level)
            for (int node = wave; node < (1 << L); node += kSubWaves) {
                int ns, ne;
                node_range(m, L, node, ns, ne);
                if (ne - ns > kCoop) {
                    const int dim = wave_split_dim(lkey, kSubT, d, lidx, ns, ne, lane);
                    wave_select(lidx + ns, (ns + ne) / 2 - ns, ne - ns, lkey + dim * kSubT, ls + ns, rsl + ns, lane);
                }
            }
            __syncthreads();
        }
        for (int node = tid; node < (1 << L); node += 64 * kSubWaves) {
            int ns, ne;
            node_range(m, L, node, ns, ne);
            if (ne - ns > 1 && ne - ns <= kCoop) {
                const int dim = split_dim(lkey, kSubT, d, lidx, ns, ne);
                select_nth(lidx + ns, lidx + (ns + ne) / 2, lidx + ne, lkey + dim * kSubT);
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < m; i += 64 * kSubWaves) idx[base + i] = lgid[lidx[i]];
}

__global__ void big_iota_kernel(const BigCloud* __restrict__ tab, int n_clouds, int total, int* __restrict__ idx) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    int lo = 0, hi = n_clouds - 1;  // cloud of position p
    while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if (tab[mid].off <= p) lo = mid;
        else hi = mid - 1;
    }
    idx[p] = p - tab[lo].off;
}

// leaves of the main tree -> two slots each (balltree.h:137-155); slot_off[c] = first slot of cloud c
__global__ void big_emit_kernel(const BigCloud* __restrict__ tab, const int* __restrict__ slot_off, int n_clouds,
                                int total_leaves, const int* __restrict__ idx, int32_t* __restrict__ tree_idx,
                                uint8_t* __restrict__ tree_mask) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total_leaves) return;
    int lo = 0, hi = n_clouds - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if (slot_off[mid] <= 2 * t) lo = mid;
        else hi = mid - 1;
    }
    const BigCloud C = tab[lo];
    const int j = t - slot_off[lo] / 2;
    int s, e;
    node_range(C.n, C.levels, j, s, e);
    const int a = idx[(size_t)C.off + s], b = (e - s == 1) ? a : idx[(size_t)C.off + s + 1];
    tree_idx[2 * (size_t)t] = C.off + a;
    tree_idx[2 * (size_t)t + 1] = C.off + b;
    tree_mask[2 * (size_t)t] = 1;
    tree_mask[2 * (size_t)t + 1] = (e - s == 1) ? 0 : 1;
}

struct Rot9 {
    float r[9];
};
// rotated leaves (balltree.pyx:639-646): fp32 leaves @ R, leaves in tree-slot order
__global__ void big_rotate_kernel(const float* __restrict__ pos, const int32_t* __restrict__ tree_idx, int total, int d,
                                  Rot9 R, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float* p = pos + (size_t)tree_idx[i] * d;
    for (int j = 0; j < d; ++j) {
        float acc = p[0] * R.r[j];
        for (int k = 1; k < d; ++k) acc = fmaf(p[k], R.r[k * d + j], acc);
        out[(size_t)i * d + j] = acc;
    }
}
// mean over `st` consecutive leaves (balltree.pyx:658-660)
__global__ void big_pool_kernel(const float* __restrict__ cur, int n2, int d, int st, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    for (int j = 0; j < d; ++j) {
        float sum = cur[((size_t)i * st) * d + j];
        for (int q = 1; q < st; ++q) sum += cur[((size_t)i * st + q) * d + j];
        out[(size_t)i * d + j] = sum / (float)st;
    }
}
__global__ void big_rot_out_kernel(const BigCloud* __restrict__ tab, int n_clouds, int total, const int* __restrict__ idx,
                                   int32_t* __restrict__ rot) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    int lo = 0, hi = n_clouds - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if (tab[mid].off <= p) lo = mid;
        else hi = mid - 1;
    }
    rot[p] = tab[lo].off + idx[p];
}

// ----------------------------------------------------------------------------------------------- host side
struct BigWs {  // carve-up of the caller's workspace
    int *idx, *gid, *w, *LS, *RS;
    float *knode, *kpos;
    float *rk_a, *rk_b;
    BigCloud* tab;
    int* slot_off;
    int2* work;
    size_t work_cap;
};
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static int ceil_log2(long long n) {  // ceil(log2(n)), n >= 1
    int b = 0;
    while ((1ll << b) < n) ++b;
    return b;
}
static size_t big_slots(const int32_t* sizes, int n_clouds) {
    size_t t = 0;
    for (int c = 0; c < n_clouds; ++c) t += (size_t)1 << ceil_log2(sizes[c] < 2 ? 2 : sizes[c]);
    return t;
}
static size_t big_work_cap(size_t nmax, int n_clouds) { return 4 * (nmax / kSubT + 1) + 4 * (size_t)n_clouds + 64; }

static size_t big_carve(char* p, size_t nmax, size_t nt, int d, int n_clouds, BigWs* ws) {
    size_t o = 0;
    auto take = [&](size_t bytes) {
        char* r = p ? p + o : nullptr;
        o += align256(bytes);
        return r;
    };
    int* a = (int*)take(nmax * 4);
    int* b = (int*)take(nmax * 4);
    int* c = (int*)take(nmax * 4);
    int* e = (int*)take((nmax + 1) * 4);
    int* f = (int*)take((nmax + 1) * 4);
    float* k = (float*)take(nmax * 4);
    float* kp = (float*)take(nmax * 4);
    float* ra = (float*)take(nt * d * 4);
    float* rb = (float*)take(nt * d * 4);
    BigCloud* tab = (BigCloud*)take((size_t)n_clouds * sizeof(BigCloud));
    int* so = (int*)take((size_t)n_clouds * 4);
    const size_t cap = big_work_cap(nmax, n_clouds);
    int2* wk = (int2*)take(cap * sizeof(int2));
    if (ws) *ws = BigWs{a, b, c, e, f, k, kp, ra, rb, tab, so, wk, cap};
    return o;
}

// One partition run: clouds (off, n, levels) over item keys [total][d]; leaves the permutation in ws.idx.
static int big_partition(const float* keys, int d, std::vector<BigCloud>& clouds, int total, const BigWs& ws,
                         hipStream_t st) {
    const int nc = (int)clouds.size();
    int max_top = 0;
    for (auto& C : clouds) {
        int top = 0;
        while (top < C.levels && ((C.n + (1 << top) - 1) >> top) > kSubT) ++top;  // ceil(n / 2^top) bounds every node
        C.top = top;
        max_top = top > max_top ? top : max_top;
    }
    if (hipMemcpyAsync(ws.tab, clouds.data(), nc * sizeof(BigCloud), hipMemcpyHostToDevice, st) != hipSuccess)
        return fail(HAET_E_LAUNCH, "balltree_big: table upload failed");
    hipLaunchKernelGGL(big_iota_kernel, dim3((total + 255) / 256), dim3(256), 0, st, ws.tab, nc, total, ws.idx);
    std::vector<int2> work;
    std::vector<size_t> lvl_off(max_top + 2, 0);
    for (int L = 0; L <= max_top; ++L) {  // levels [0, max_top): top launches; entry max_top: the sub-tree roots
        lvl_off[L] = work.size();
        for (int c = 0; c < nc; ++c) {
            const BigCloud& C = clouds[c];
            if (L < max_top) {
                if (C.top > L)
                    for (int j = 0; j < (1 << L); ++j) work.push_back(int2{c, j});
            } else if (C.levels > C.top) {
                for (int j = 0; j < (1 << C.top); ++j) work.push_back(int2{c, j});
            }
        }
    }
    lvl_off[max_top + 1] = work.size();
    if (work.size() > ws.work_cap) return fail(HAET_E_ARG, "balltree_big: work list %zu > %zu", work.size(), ws.work_cap);
    if (!work.empty() &&
        hipMemcpyAsync(ws.work, work.data(), work.size() * sizeof(int2), hipMemcpyHostToDevice, st) != hipSuccess)
        return fail(HAET_E_LAUNCH, "balltree_big: work list upload failed");
    for (int L = 0; L < max_top; ++L) {
        const int nb = (int)(lvl_off[L + 1] - lvl_off[L]);
        if (nb <= 0) continue;
        int biggest = 0;  // ceil(n / 2^L) bounds every node of a cloud's level L
        for (const BigCloud& C : clouds)
            if (C.top > L) biggest = std::max(biggest, (C.n + (1 << L) - 1) >> L);
        if (biggest <= kTopLds) {
            constexpr int lds = 16 * kTopLds + 32;
            static PerDeviceOnce attr;
            if (!attr.done()) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(big_top_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        lds) != hipSuccess)
                    return fail(HAET_E_LAUNCH, "balltree_big: hipFuncSetAttribute failed");
                attr.set();
            }
            hipLaunchKernelGGL(big_top_kernel<true>, dim3(nb), dim3(kTopThreads), lds, st, keys, d, ws.tab, ws.work + lvl_off[L], L,
                               ws.idx, ws.gid, ws.knode, ws.w, ws.LS, ws.RS, ws.kpos);
        } else {
            hipLaunchKernelGGL(big_top_kernel<false>, dim3(nb), dim3(kTopThreads), 0, st, keys, d, ws.tab, ws.work + lvl_off[L], L,
                               ws.idx, ws.gid, ws.knode, ws.w, ws.LS, ws.RS, ws.kpos);
        }
    }
    const int nsub = (int)(lvl_off[max_top + 1] - lvl_off[max_top]);
    if (nsub > 0)
        hipLaunchKernelGGL(big_sub_kernel, dim3(nsub), dim3(64 * kSubWaves), 0, st, keys, d, ws.tab, ws.work + lvl_off[max_top], ws.idx);
    // the uploads above read host vectors that die with this frame
    if (hipStreamSynchronize(st) != hipSuccess) return fail(HAET_E_LAUNCH, "balltree_big: %s", hipGetErrorString(hipGetLastError()));
    return check_launch("balltree_big partition");
}

}  // namespace haet

using namespace haet;

extern "C" size_t haet_balltree_big_slots(const int32_t* cloud_sizes, int n_clouds) {
    if (!cloud_sizes || n_clouds < 1) return 0;
    return big_slots(cloud_sizes, n_clouds);
}

extern "C" size_t haet_balltree_big_workspace_bytes(const int32_t* cloud_sizes, int n_clouds, int d) {
    if (!cloud_sizes || n_clouds < 1) return 0;
    const size_t nt = big_slots(cloud_sizes, n_clouds);
    return big_carve(nullptr, nt, nt, d, n_clouds, nullptr);
}

extern "C" int haet_balltree_big_build(const float* pos, const int32_t* cloud_sizes, int n_clouds, int d, int n_levels,
                                       const int32_t* strides, const int32_t* ball_sizes, float angle_deg,
                                       int32_t* tree_idx, uint8_t* tree_mask, int32_t* rot, void* workspace,
                                       size_t workspace_bytes, haet_stream_t stream) {
    if (!pos || !cloud_sizes || !tree_idx || !tree_mask || !workspace || n_clouds < 1)
        return fail(HAET_E_ARG, "balltree_big: null pointer");
    if (d != 2 && d != 3) return fail(HAET_E_ARG, "balltree_big: d=%d", d);
    if (n_levels < 1 || n_levels > HAET_ERWIN_MAX_LEVELS) return fail(HAET_E_ARG, "balltree_big: n_levels=%d", n_levels);
    long long N = 0;
    for (int c = 0; c < n_clouds; ++c) {
        if (cloud_sizes[c] < 2) return fail(HAET_E_ARG, "balltree_big: cloud %d has %d points (at least 2)", c, cloud_sizes[c]);
        N += cloud_sizes[c];
    }
    const size_t nt = big_slots(cloud_sizes, n_clouds);
    if (nt > (size_t)1 << 30) return fail(HAET_E_ARG, "balltree_big: %zu tree slots", nt);
    BigWs ws;
    const size_t need = big_carve((char*)workspace, nt, nt, d, n_clouds, &ws);
    if (workspace_bytes < need) return fail(HAET_E_ARG, "balltree_big: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    if (stream_is_capturing(st))
        return fail(HAET_E_ARG, "balltree_big: the build synchronises the stream between its level launches and cannot "
                                "run under stream capture (build the tree before the capture begins)");

    // main tree: ceil(log2 n) - 1 levels of median splits, then leaves of one or two points (balltree.pyx:55-60)
    std::vector<BigCloud> clouds(n_clouds);
    std::vector<int> slot_off(n_clouds);
    {
        int off = 0;
        size_t so = 0;
        for (int c = 0; c < n_clouds; ++c) {
            const int lv = ceil_log2(cloud_sizes[c]) - 1;
            clouds[c] = BigCloud{off, cloud_sizes[c], lv, 0};
            slot_off[c] = (int)so;
            off += cloud_sizes[c];
            so += (size_t)2 << lv;
        }
    }
    int rc = big_partition(pos, d, clouds, (int)N, ws, st);
    if (rc != HAET_OK) return rc;
    if (hipMemcpyAsync(ws.slot_off, slot_off.data(), n_clouds * 4, hipMemcpyHostToDevice, st) != hipSuccess)
        return fail(HAET_E_LAUNCH, "balltree_big: upload failed");
    hipLaunchKernelGGL(big_emit_kernel, dim3((unsigned)((nt / 2 + 255) / 256)), dim3(256), 0, st, ws.tab, ws.slot_off, n_clouds,
                       (int)(nt / 2), ws.idx, tree_idx, tree_mask);
    if (hipStreamSynchronize(st) != hipSuccess) return fail(HAET_E_LAUNCH, "balltree_big: emit failed");
    if (!(angle_deg > 0.f)) return check_launch("balltree_big");

    // rotated re-partitions (balltree.pyx:636-660)
    if (!rot || !ball_sizes || (n_levels > 1 && !strides)) return fail(HAET_E_ARG, "balltree_big: rot/ball_sizes/strides missing");
    Rot9 R;
    rotation_matrix_f32(angle_deg, d, R.r);
    hipLaunchKernelGGL(big_rotate_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, pos, tree_idx, (int)nt, d, R, ws.rk_a);
    float *cur = ws.rk_a, *nxt = ws.rk_b;
    size_t rot_off = 0;
    long long div = 1;
    for (int lev = 0; lev < n_levels; ++lev) {
        if (ball_sizes[lev] < 1) return fail(HAET_E_ARG, "balltree_big: ball size %d", ball_sizes[lev]);
        // NB: the TOTAL level-0 leaf count over all clouds enters (reference quirk, balltree.pyx:643)
        const double ratio = (double)nt / (double)ball_sizes[lev];
        int target = (int)std::log2(ratio);
        if (target < 0) target = 0;
        int off = 0;
        for (int c = 0; c < n_clouds; ++c) {
            const long long slots = ((long long)2 << (ceil_log2(cloud_sizes[c]) - 1));
            if (slots % div) return fail(HAET_E_ARG, "balltree_big: strides do not divide cloud %d's %lld slots", c, slots);
            const int n = (int)(slots / div);
            int lg = 0;
            while ((1 << (lg + 1)) <= n) ++lg;
            // partition_ball_tree stops at the target level or at single points (balltree.h:221-229); n is a power of two
            clouds[c] = BigCloud{off, n, target < lg ? target : lg, 0};
            off += n;
        }
        const int total = off;
        rc = big_partition(cur, d, clouds, total, ws, st);
        if (rc != HAET_OK) return rc;
        hipLaunchKernelGGL(big_rot_out_kernel, dim3((total + 255) / 256), dim3(256), 0, st, ws.tab, n_clouds, total, ws.idx,
                           rot + rot_off);
        rot_off += total;
        if (lev < n_levels - 1) {
            const int stv = strides[lev];
            if (stv < 1 || total % stv) return fail(HAET_E_ARG, "balltree_big: stride %d", stv);
            hipLaunchKernelGGL(big_pool_kernel, dim3((total / stv + 255) / 256), dim3(256), 0, st, cur, total / stv, d, stv, nxt);
            float* t = cur;
            cur = nxt;
            nxt = t;
            div *= stv;
        }
        if (hipStreamSynchronize(st) != hipSuccess) return fail(HAET_E_LAUNCH, "balltree_big: rotated level %d failed", lev);
    }
    return check_launch("balltree_big");
}
