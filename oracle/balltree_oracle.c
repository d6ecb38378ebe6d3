/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the ball-tree stage of the
 * PhysicsAttention hot path.  Never imported/linked by the product
 * (haet_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * Plain-C restatement of the reference's ball-tree builder:
 *   - median split along the max-spread dimension
 *       (reference: models/components/balltree/balltree.h:32-66, 82-101)
 *   - complete binary tree with 2 slots per leaf, padding duplicates masked
 *       (balltree.h:121-186)
 *   - fixed-depth re-partition used for the rotated balls
 *       (balltree.h:205-259)
 *   - per-cloud driver: depth = ceil(log2 n) - 1, idx initialised 0..n-1,
 *     cloud offset added afterwards
 *       (models/components/balltree/balltree.pyx:58-60, 62-122, 124-183)
 *
 * The reference delegates the split to std::nth_element, whose *resulting
 * permutation* (not just the partition property) is observable downstream
 * (BallPooling concatenates leaf pairs in that order,
 * models/components/erwinflash/components/pooling.py:35-38).  The third-party
 * algorithm is GNU libstdc++'s introselect (GCC 11.4 in the reference
 * container): median-of-three pivot moved to the front, unguarded Hoare
 * partition while the range is > 3, heap-select fallback when the depth budget
 * 2*floor(log2 n) is exhausted, insertion sort of the last <= 3 elements.  It
 * is restated below from that published algorithm and
 * pinned against the compiled reference (oracle/_ref/libballtree_ref.so) and
 * the committed fixtures tests/golden/tree_*.npz.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct {
    const double *data; /* [n, nf] row-major, cloud-local */
    int64_t nf;
    int64_t dim;
} keyctx;

static inline int less_idx(const keyctx *k, int64_t a, int64_t b) {
    return k->data[a * k->nf + k->dim] < k->data[b * k->nf + k->dim];
}

static inline void swap_i64(int64_t *a, int64_t *b) {
    int64_t t = *a;
    *a = *b;
    *b = t;
}

/* ---- heap helpers (only reached when the introselect depth budget is spent) */
static void push_heap_(int64_t *first, int64_t hole, int64_t top, int64_t value,
                       const keyctx *k) {
    int64_t parent = (hole - 1) / 2;
    while (hole > top && less_idx(k, first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

static void adjust_heap_(int64_t *first, int64_t hole, int64_t len, int64_t value,
                         const keyctx *k) {
    const int64_t top = hole;
    int64_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (less_idx(k, first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    push_heap_(first, hole, top, value, k);
}

static void heap_select_(int64_t *first, int64_t *middle, int64_t *last, const keyctx *k) {
    const int64_t len = middle - first;
    if (len >= 2) {
        int64_t parent = (len - 2) / 2;
        for (;;) {
            adjust_heap_(first, parent, len, first[parent], k);
            if (parent == 0) break;
            parent--;
        }
    }
    for (int64_t *i = middle; i < last; ++i) {
        if (less_idx(k, *i, *first)) {
            int64_t value = *i;
            *i = *first;
            adjust_heap_(first, 0, len, value, k);
        }
    }
}

static void insertion_sort_(int64_t *first, int64_t *last, const keyctx *k) {
    if (first == last) return;
    for (int64_t *i = first + 1; i != last; ++i) {
        int64_t val = *i;
        if (less_idx(k, val, *first)) {
            for (int64_t *p = i; p != first; --p) *p = *(p - 1);
            *first = val;
        } else {
            int64_t *hole = i;
            int64_t *prev = i - 1;
            while (less_idx(k, val, *prev)) {
                *hole = *prev;
                hole = prev;
                --prev;
            }
            *hole = val;
        }
    }
}

static int floor_log2_i64(int64_t n) {
    int l = -1;
    while (n > 0) {
        n >>= 1;
        ++l;
    }
    return l;
}

/* introselect: leaves idx[nth] in sorted position w.r.t. coordinate k->dim */
static void select_nth(int64_t *first, int64_t *nth, int64_t *last, const keyctx *k) {
    if (first == last || nth == last) return;
    int depth = 2 * floor_log2_i64(last - first);
    while (last - first > 3) {
        if (depth == 0) {
            heap_select_(first, nth + 1, last, k);
            swap_i64(first, nth);
            return;
        }
        --depth;
        int64_t *a = first + 1, *b = first + (last - first) / 2, *c = last - 1;
        /* median of (a, b, c) goes to *first */
        if (less_idx(k, *a, *b)) {
            if (less_idx(k, *b, *c)) swap_i64(first, b);
            else if (less_idx(k, *a, *c)) swap_i64(first, c);
            else swap_i64(first, a);
        } else if (less_idx(k, *a, *c)) swap_i64(first, a);
        else if (less_idx(k, *b, *c)) swap_i64(first, c);
        else swap_i64(first, b);
        /* unguarded Hoare partition of [first+1, last) around pivot *first */
        int64_t *lo = first + 1, *hi = last;
        for (;;) {
            while (less_idx(k, *lo, *first)) ++lo;
            --hi;
            while (less_idx(k, *first, *hi)) --hi;
            if (!(lo < hi)) break;
            swap_i64(lo, hi);
            ++lo;
        }
        if (lo <= nth) first = lo;
        else last = lo;
    }
    insertion_sort_(first, last, k);
}

/* balltree.h:32-66: first dimension with strictly largest spread (ties -> lowest) */
static int64_t split_dim(const double *data, const int64_t *idx, int64_t start, int64_t end,
                         int64_t nf) {
    if (end - start <= 1) return 0;
    int64_t best = 0;
    double best_spread = 0.0;
    for (int64_t d = 0; d < nf; ++d) {
        double lo = data[idx[start] * nf + d], hi = lo;
        for (int64_t i = start + 1; i < end; ++i) {
            double v = data[idx[i] * nf + d];
            lo = v < lo ? v : lo; /* std::min(lo, v) */
            hi = hi < v ? v : hi; /* std::max(hi, v) */
        }
        if (hi - lo > best_spread) {
            best_spread = hi - lo;
            best = d;
        }
    }
    return best;
}

static void split_node(const double *data, int64_t *idx, int64_t start, int64_t end, int64_t nf) {
    const int64_t n = end - start, mid = (start + end) / 2;
    keyctx k = {data, nf, split_dim(data, idx, start, end, nf)};
    if (n > 1) select_nth(idx + start, idx + mid, idx + end, &k);
}

static void build_rec(const double *data, int64_t *idx, int64_t *out, uint8_t *mask,
                      int64_t start, int64_t end, int64_t ostart, int64_t nf, int64_t level,
                      int64_t max_level) {
    const int64_t n = end - start;
    if (level == max_level) {
        if (n == 1) {
            out[ostart] = out[ostart + 1] = idx[start];
            mask[ostart] = 1;
            mask[ostart + 1] = 0;
        } else {
            out[ostart] = idx[start];
            out[ostart + 1] = idx[start + 1];
            mask[ostart] = mask[ostart + 1] = 1;
        }
        return;
    }
    split_node(data, idx, start, end, nf);
    const int64_t mid = (start + end) / 2;
    const int64_t left_slots = ((int64_t)1 << (max_level - level - 1)) * 2;
    build_rec(data, idx, out, mask, start, mid, ostart, nf, level + 1, max_level);
    build_rec(data, idx, out, mask, mid, end, ostart + left_slots, nf, level + 1, max_level);
}

static void part_rec(const double *data, int64_t *idx, int64_t *out, int64_t start, int64_t end,
                     int64_t ostart, int64_t nf, int64_t level, int64_t target) {
    const int64_t n = end - start;
    if (level == target || n <= 1) {
        for (int64_t i = 0; i < n; ++i) out[ostart + i] = idx[start + i];
        return;
    }
    split_node(data, idx, start, end, nf);
    const int64_t mid = (start + end) / 2;
    part_rec(data, idx, out, start, mid, ostart, nf, level + 1, target);
    part_rec(data, idx, out, mid, end, ostart + (mid - start), nf, level + 1, target);
}

/* balltree.pyx:58-60 */
int64_t oracle_tree_depth(int64_t n) { return (int64_t)ceil(log2((double)n)) - 1; }

/*
 * Build the trees of `nb` clouds.  offsets[nb+1] delimit the clouds in data[.,nf];
 * out/mask hold sum_b 2*2^depth(n_b) slots (caller sizes them with
 * oracle_tree_depth).  Returns 0, or -1 on allocation failure / bad offsets.
 * (balltree.pyx:185-300 + 62-122.)
 */
int oracle_build_balltree(const double *data, const int64_t *offsets, int64_t nb, int64_t nf,
                          int64_t *out, uint8_t *mask) {
    int64_t tree_off = 0;
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t s = offsets[b], e = offsets[b + 1], n = e - s;
        if (n <= 0) return -1;
        const int64_t depth = oracle_tree_depth(n);
        if (depth < 0) return -1;
        const int64_t slots = ((int64_t)1 << depth) * 2;
        int64_t *idx = (int64_t *)malloc((size_t)n * sizeof(int64_t));
        if (!idx) return -1;
        for (int64_t i = 0; i < n; ++i) idx[i] = i;
        build_rec(data + s * nf, idx, out + tree_off, mask + tree_off, 0, n, 0, nf, 0, depth);
        for (int64_t i = 0; i < slots; ++i) out[tree_off + i] += s;
        free(idx);
        tree_off += slots;
    }
    return 0;
}

/* balltree.pyx:302-422 + 124-183: `target` median splits per cloud, output same size as input */
int oracle_partition_balltree(const double *data, const int64_t *offsets, int64_t nb, int64_t nf,
                              int64_t target, int64_t *out) {
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t s = offsets[b], e = offsets[b + 1], n = e - s;
        if (n <= 0) return -1;
        int64_t *idx = (int64_t *)malloc((size_t)n * sizeof(int64_t));
        if (!idx) return -1;
        for (int64_t i = 0; i < n; ++i) idx[i] = i;
        part_rec(data + s * nf, idx, out + s, 0, n, 0, nf, 0, target);
        for (int64_t i = 0; i < n; ++i) out[s + i] += s;
        free(idx);
    }
    return 0;
}
