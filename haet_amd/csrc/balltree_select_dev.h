// Step-for-step device version of libstdc++'s std::nth_element (introselect) as the reference's ball tree uses it
// (models/components/balltree/balltree.h:82-100), one lane per call, plus the split-dimension rule (balltree.h:32-66)
// and the node ranges of the mid = (s + e) / 2 recursion (balltree.h:158-186).  Written from the algorithm description
// in SURVEY.md appendix A.  `idx` holds indices into the key array `kd`; any address space.
#pragma once

namespace haet {

__device__ __forceinline__ bool lt(const float* kd, int a, int b) { return kd[a] < kd[b]; }
__device__ __forceinline__ void swp(int* a, int* b) {
    const int t = *a;
    *a = *b;
    *b = t;
}

__device__ void push_heap_(int* first, int hole, int top, int value, const float* kd) {
    int parent = (hole - 1) / 2;
    while (hole > top && lt(kd, first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

__device__ void adjust_heap_(int* first, int hole, int len, int value, const float* kd) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (lt(kd, first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    push_heap_(first, hole, top, value, kd);
}

__device__ void heap_select_(int* first, int* middle, int* last, const float* kd) {
    const int len = (int)(middle - first);
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) {
            adjust_heap_(first, parent, len, first[parent], kd);
            if (parent == 0) break;
            parent--;
        }
    }
    for (int* i = middle; i < last; ++i) {
        if (lt(kd, *i, *first)) {
            const int value = *i;
            *i = *first;
            adjust_heap_(first, 0, len, value, kd);
        }
    }
}

__device__ void insertion_sort_(int* first, int* last, const float* kd) {
    if (first == last) return;
    for (int* i = first + 1; i != last; ++i) {
        const int val = *i;
        if (lt(kd, val, *first)) {
            for (int* p = i; p != first; --p) *p = *(p - 1);
            *first = val;
        } else {
            int* hole = i;
            int* prev = i - 1;
            while (lt(kd, val, *prev)) {
                *hole = *prev;
                hole = prev;
                --prev;
            }
            *hole = val;
        }
    }
}

__device__ void select_nth(int* first, int* nth, int* last, const float* kd) {
    if (first == last || nth == last) return;
    int depth = 2 * (31 - __clz((int)(last - first)));
    while (last - first > 3) {
        if (depth == 0) {
            heap_select_(first, nth + 1, last, kd);
            swp(first, nth);
            return;
        }
        --depth;
        int *a = first + 1, *b = first + (last - first) / 2, *c = last - 1;
        if (lt(kd, *a, *b)) {
            if (lt(kd, *b, *c)) swp(first, b);
            else if (lt(kd, *a, *c)) swp(first, c);
            else swp(first, a);
        } else if (lt(kd, *a, *c)) swp(first, a);
        else if (lt(kd, *b, *c)) swp(first, c);
        else swp(first, b);
        int *lo = first + 1, *hi = last;
        for (;;) {
            while (lt(kd, *lo, *first)) ++lo;
            --hi;
            while (lt(kd, *first, *hi)) --hi;
            if (!(lo < hi)) break;
            swp(lo, hi);
            ++lo;
        }
        if (lo <= nth) first = lo;
        else last = lo;
    }
    insertion_sort_(first, last, kd);
}

// balltree.h:32-66
__device__ int split_dim(const float* keys, int kstride, int nd, const int* idx, int s, int e) {
    if (e - s <= 1) return 0;
    int best = 0;
    double best_spread = 0.0;
    for (int d = 0; d < nd; ++d) {
        const float* kd = keys + d * kstride;
        float lo = kd[idx[s]], hi = lo;
        for (int i = s + 1; i < e; ++i) {
            const float v = kd[idx[i]];
            lo = v < lo ? v : lo;
            hi = hi < v ? v : hi;
        }
        const double spread = (double)hi - (double)lo;
        if (spread > best_spread) {
            best_spread = spread;
            best = d;
        }
    }
    return best;
}

// node `j` of level `L` covers idx[s, e) obtained by halving [0, n) along j's bits (MSB first)
__device__ __forceinline__ void node_range(int n, int L, int j, int& s, int& e) {
    s = 0;
    e = n;
    for (int b = L - 1; b >= 0; --b) {
        const int mid = (s + e) / 2;
        if ((j >> b) & 1) s = mid;
        else e = mid;
    }
}

// ------------------------------------------------------------------------- one wave on one node (LDS arrays)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// std::nth_element(W, W + nth, W + n) by ONE WAVE on LDS arrays: the rounds of big_top_kernel with ballots instead of
// workgroup scans.  ls / rsl: scratch of n ints each.
__device__ void wave_select(int* W, int nth, int n, const float* kd, int* ls, int* rsl, int lane) {
    int first = 0, last = n;
    int depth = 2 * (31 - __clz(n));
    while (last - first > 3) {
        if (depth == 0) {
            if (lane == 0) {
                heap_select_(W + first, W + nth + 1, W + last, kd);
                swp(W + first, W + nth);
            }
            wave_sync();
            return;
        }
        --depth;
        if (lane == 0) {
            int *a = W + first + 1, *b = W + first + (last - first) / 2, *c = W + last - 1, *f = W + first;
            if (lt(kd, *a, *b)) {
                if (lt(kd, *b, *c)) swp(f, b);
                else if (lt(kd, *a, *c)) swp(f, c);
                else swp(f, a);
            } else if (lt(kd, *a, *c)) swp(f, a);
            else if (lt(kd, *b, *c)) swp(f, c);
            else swp(f, b);
        }
        wave_sync();
        const float v = kd[W[first]];
        int nL = 0, nR = 0;  // stoppers inside [first + 1, last), ranked from the left
        for (int p0 = first + 1; p0 < last; p0 += 64) {
            const int p = p0 + lane;
            const float x = p < last ? kd[W[p]] : 0.f;
            const bool isl = p < last && !(x < v), isr = p < last && !(v < x);
            const unsigned long long ml = __ballot(isl), mr = __ballot(isr), below = (1ull << lane) - 1ull;
            if (isl) ls[nL + __popcll(ml & below)] = p;
            if (isr) rsl[nR + __popcll(mr & below)] = p;
            nL += __popcll(ml);
            nR += __popcll(mr);
        }
        wave_sync();
        // k-th right stopper from the right; the pivot's slot is the last one
        auto Rk = [&](int k) { return k < nR ? rsl[nR - 1 - k] : first; };
        const int kmax = min(nL, nR + 1);
        int K = 0;
        for (int k0 = 0; k0 < kmax; k0 += 64) {
            const int k = k0 + lane;
            K += __popcll(__ballot(k < kmax && ls[k] < Rk(k)));
        }
        const int bound = K > 0 ? Rk(K - 1) : last;
        int cut = bound;
        if (K < nL) {
            const int Lk = ls[K];
            cut = Lk < bound ? Lk : bound;
        }
        for (int k = lane; k < K; k += 64) {
            const int a = ls[k], b = Rk(k);
            const int t = W[a];
            W[a] = W[b];
            W[b] = t;
        }
        wave_sync();
        if (cut <= nth) first = cut;
        else last = cut;
    }
    if (lane == 0) insertion_sort_(W + first, W + last, kd);
    wave_sync();
}

// balltree.h:32-66 by one wave
__device__ int wave_split_dim(const float* keys, int kstride, int nd, const int* idx, int s, int e, int lane) {
    int best = 0;
    double best_spread = 0.0;
    for (int k = 0; k < nd; ++k) {
        const float* kd = keys + k * kstride;
        float lo = INFINITY, hi = -INFINITY;
        for (int i = s + lane; i < e; i += 64) {
            const float v = kd[idx[i]];
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
        lo = wave_min(lo);
        hi = wave_max(hi);
        const double spread = (double)hi - (double)lo;
        if (spread > best_spread) best_spread = spread, best = k;
    }
    return best;
}

}  // namespace haet
