// Device ball-tree builder: the reference builds this tree on the HOST every forward
// (models/components/balltree/balltree.pyx:542-550 -> a GPU->CPU->GPU sync); here one
// wave per cloud builds it in LDS.
//
// The permutation must equal what libstdc++'s std::nth_element leaves behind
// (balltree.h:91-100), because BallPooling concatenates leaf pairs in that order
// (pooling.py:35-38).  So this is a step-for-step device version of introselect
// (median-of-3 to front, unguarded Hoare partition while range > 3, heap-select when
// the 2*floor(log2 n) budget is spent, insertion sort of the last <= 3), written from
// the algorithm description in SURVEY.md appendix A.  All nodes of one tree level are
// independent: the large ones take a wave each (wave_select: the same introselect in ballot rounds),
// the small ones a lane each.
//
// Keys are the fp32 positions; the reference converts them to float64 first
// (balltree.pyx:542), an exact, order-preserving conversion, so fp32 compares give
// identical decisions.  Spreads are differenced in double like the reference.
#include <cstdlib>

#include "common.h"
#include "balltree_select_dev.h"

namespace haet {

constexpr int kMaxPts = 128;  // slots per cloud
constexpr int kMaxDim = 3;

#ifndef HAET_TREE_WAVES
#define HAET_TREE_WAVES 8
#endif
constexpr int kTreeWaves = HAET_TREE_WAVES;  // waves of a cloud's workgroup: the wave-selected nodes of a level are dealt to them
constexpr int kLaneNode = 4;   // (default; HAET_TREE_LANE_NODE overrides it for sweeps) nodes of more items than this are selected by a whole wave (wave_select), smaller ones by a lane

// split every node of level L.  A node of > kLaneNode items takes a wave (ballot rounds of the same introselect: a
// one-lane select of the 128-item root is ~10^5 cycles of dependent LDS reads, and it was 2/3 of this kernel), the
// small ones a lane each; both leave std::nth_element's permutation.  ls / rsl: scratch of n ints each.
__device__ void split_level(const float* keys, int kstride, int nd, int* idx, int* ls, int* rsl, int n, int L, int tid, int lane_node) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (((n + (1 << L) - 1) >> L) > lane_node) {  // (an upper bound of every node size of this level)
        for (int node = wave; node < (1 << L); node += kTreeWaves) {
            int s, e;
            node_range(n, L, node, s, e);
            if (e - s > lane_node) {
                const int dim = wave_split_dim(keys, kstride, nd, idx, s, e, lane);
                wave_select(idx + s, (s + e) / 2 - s, e - s, keys + dim * kstride, ls + s, rsl + s, lane);
            }
        }
    }
    if (tid < (1 << L)) {
        int s, e;
        node_range(n, L, tid, s, e);
        if (e - s > 1 && e - s <= lane_node) {
            const int dim = split_dim(keys, kstride, nd, idx, s, e);
            select_nth(idx + s, idx + (s + e) / 2, idx + e, keys + dim * kstride);
        }
    }
    __syncthreads();
}

struct Rot3 {
    float r[9];
};

__global__ __launch_bounds__(64 * kTreeWaves) void balltree_kernel(
    const float* __restrict__ pos, int M, int nd, int depth, int n_levels, int s0, int s1, int s2,
    int t0, int t1, int t2, int t3, int rotate, Rot3 R, int32_t* __restrict__ tree_idx,
    uint8_t* __restrict__ tree_mask, int32_t* __restrict__ rot, int lane_node) {
    __shared__ float kmain[kMaxDim * kMaxPts];
    __shared__ float ka[kMaxDim * kMaxPts];
    __shared__ float kb[kMaxDim * kMaxPts];
    __shared__ int idx[kMaxPts];
    __shared__ int tl[kMaxPts];
    __shared__ int ls[kMaxPts];
    __shared__ int rsl[kMaxPts];
    constexpr int kT = 64 * kTreeWaves;
    const int lane = threadIdx.x, cloud = blockIdx.x, n_clouds = gridDim.x;  // (`lane`: the thread of the workgroup)
    const int Mp = 2 << depth;
    const int strides[3] = {s0, s1, s2};
    const int targets[4] = {t0, t1, t2, t3};

    for (int i = lane; i < M; i += kT) {
        idx[i] = i;
        for (int d = 0; d < nd; ++d) kmain[d * kMaxPts + i] = pos[((size_t)cloud * M + i) * nd + d];
    }
    __syncthreads();
    // main tree: balltree.h:121-186
    for (int L = 0; L < depth; ++L) split_level(kmain, kMaxPts, nd, idx, ls, rsl, M, L, lane, lane_node);
    if (lane < (1 << depth)) {
        int s, e;
        node_range(M, depth, lane, s, e);
        const int a = idx[s], b = (e - s == 1) ? a : idx[s + 1];
        tl[2 * lane] = a;
        tl[2 * lane + 1] = b;
        const size_t o = (size_t)cloud * Mp + 2 * lane;
        tree_idx[o] = cloud * M + a;
        tree_idx[o + 1] = cloud * M + b;
        tree_mask[o] = 1;
        tree_mask[o + 1] = (e - s == 1) ? 0 : 1;
    }
    __syncthreads();
    if (!rotate) return;

    // rotated leaves (balltree.pyx:639-646): fp32 leaves @ R
    for (int i = lane; i < Mp; i += kT) {
        const int p = tl[i];
        for (int j = 0; j < nd; ++j) {
            float acc = kmain[p] * R.r[j];
            for (int k = 1; k < nd; ++k) acc = fmaf(kmain[k * kMaxPts + p], R.r[k * nd + j], acc);
            ka[j * kMaxPts + i] = acc;
        }
    }
    float* cur = ka;
    float* nxt = kb;
    int n = Mp;
    size_t rot_off = 0;
    for (int lev = 0; lev < n_levels; ++lev) {
        for (int i = lane; i < n; i += kT) idx[i] = i;
        __syncthreads();
        // balltree.h:205-259: `target` halvings, nodes of <= 1 point stop early
        for (int L = 0; L < targets[lev] && (n >> L) > 1; ++L)
            split_level(cur, kMaxPts, nd, idx, ls, rsl, n, L, lane, lane_node);
        for (int i = lane; i < n; i += kT)
            rot[rot_off + (size_t)cloud * n + i] = cloud * n + idx[i];
        rot_off += (size_t)n_clouds * n;
        if (lev < n_levels - 1) {
            // mean-pool consecutive `stride` leaves (balltree.pyx:658-660)
            const int st = strides[lev], n2 = n / st;
            for (int i = lane; i < n2; i += kT)
                for (int j = 0; j < nd; ++j) {
                    float sum = cur[j * kMaxPts + i * st];
                    for (int q = 1; q < st; ++q) sum += cur[j * kMaxPts + i * st + q];
                    nxt[j * kMaxPts + i] = sum / (float)st;
                }
            float* t = cur;
            cur = nxt;
            nxt = t;
            n = n2;
            __syncthreads();
        }
    }
}

// balltree_wave.hip: register-resident variant for power-of-two clouds of <= 64 points
bool balltree_wave_ok(int M, int n_levels, const int* st);
int token_pos_launch(const float* S, int T, int C, int d, float* pos, hipStream_t stream);
int balltree_wave_launch(const float* pos, int n_clouds, int M, int d, int n_levels, const int* st,
                         const int* tg, int rotate, const float* R9, int32_t* tree_idx,
                         uint8_t* tree_mask, int32_t* rot, hipStream_t stream, const float* S, int C,
                         float* pos_out);

// generate_rotation_matrix (balltree.pyx:576-596): python doubles -> fp32 tensor
void rotation_matrix_f32(float angle_deg, int d, float* r) {
    const double a = (double)angle_deg * (3.14159265358979323846 / 180.0);  // math.radians
    const double c = cos(a), s = sin(a);
    if (d == 2) {
        const double m[4] = {c, -s, s, c};
        for (int i = 0; i < 4; ++i) r[i] = (float)m[i];
    } else {
        const double m[9] = {c * c, s * c * (s - 1), s * (s + c * c), s * c, s * s * s + c * c,
                             s * c * (s - 1), -s, s * c, c * c};
        for (int i = 0; i < 9; ++i) r[i] = (float)m[i];
    }
}

}  // namespace haet

using namespace haet;

static int balltree_build_impl(const float* pos, int n_clouds, int M, int d, int n_levels,
                               const int* strides, const int* targets, float angle_deg,
                               int32_t* tree_idx, uint8_t* tree_mask, int32_t* rot,
                               haet_stream_t stream, const float* S, int C) {
    // S != nullptr: `pos` is an OUTPUT (token positions derived from the tokens' first d channels)
    if (!pos || !tree_idx || !tree_mask || n_clouds < 1)
        return fail(HAET_E_ARG, "balltree: null pointer or n_clouds < 1");
    if (M < 2 || M > kMaxPts) return fail(HAET_E_ARG, "balltree: M=%d unsupported (2..128)", M);
    if (d != 2 && d != 3) return fail(HAET_E_ARG, "balltree: d=%d unsupported (2 or 3)", d);
    if (n_levels < 1 || n_levels > 4) return fail(HAET_E_ARG, "balltree: n_levels=%d", n_levels);
    // depth = ceil(log2 M) - 1   (balltree.pyx:58-60)
    int depth = 0;
    while ((2 << depth) < M) ++depth;
    const int Mp = 2 << depth;
    int st[3] = {1, 1, 1}, tg[4] = {0, 0, 0, 0};
    const int rotate = angle_deg > 0.f;
    if (rotate) {
        if (!rot || !targets || (n_levels > 1 && !strides))
            return fail(HAET_E_ARG, "balltree: rot/targets/strides missing");
        int n = Mp;
        for (int i = 0; i < n_levels; ++i) {
            tg[i] = targets[i];
            if (tg[i] < 0) return fail(HAET_E_ARG, "balltree: negative target");
            if (i < n_levels - 1) {
                st[i] = strides[i];
                if (st[i] < 1 || n % st[i]) return fail(HAET_E_ARG, "balltree: stride %d", st[i]);
                n /= st[i];
            }
        }
    }
    Rot3 R = {};
    rotation_matrix_f32(angle_deg, d, R.r);
    const char* force_lds = getenv("HAET_TREE_LDS");
    if (!(force_lds && force_lds[0] == '1') && balltree_wave_ok(M, n_levels, st))
        return balltree_wave_launch(pos, n_clouds, M, d, n_levels, st, tg, rotate, R.r, tree_idx,
                                    tree_mask, rot, as_stream(stream), S, C, const_cast<float*>(pos));
    if (S) {  // generic tree kernel: positions by the stand-alone kernel first
        const int rc = token_pos_launch(S, n_clouds * M, C, d, const_cast<float*>(pos), as_stream(stream));
        if (rc != HAET_OK) return rc;
    }
    static const int lane_node = [] {
        const char* e = getenv("HAET_TREE_LANE_NODE");
        const int v = e ? atoi(e) : kLaneNode;
        return v < 1 ? 1 : v;
    }();
    hipLaunchKernelGGL(balltree_kernel, dim3(n_clouds), dim3(64 * kTreeWaves), 0, as_stream(stream), pos, M, d,
                       depth, n_levels, st[0], st[1], st[2], tg[0], tg[1], tg[2], tg[3], rotate, R,
                       tree_idx, tree_mask, rot, lane_node);
    return check_launch("balltree_kernel");
}

extern "C" int haet_balltree_build(const float* pos, int n_clouds, int M, int d, int n_levels,
                                   const int* strides, const int* targets, float angle_deg,
                                   int32_t* tree_idx, uint8_t* tree_mask, int32_t* rot,
                                   haet_stream_t stream) {
    return balltree_build_impl(pos, n_clouds, M, d, n_levels, strides, targets, angle_deg, tree_idx,
                               tree_mask, rot, stream, nullptr, 0);
}

extern "C" int haet_tokens_tree_build(const float* S, int C, float* pos, int n_clouds, int M, int d,
                                      int n_levels, const int* strides, const int* targets,
                                      float angle_deg, int32_t* tree_idx, uint8_t* tree_mask,
                                      int32_t* rot, haet_stream_t stream) {
    if (!S || C < d) return fail(HAET_E_ARG, "tokens_tree_build: S is NULL or C < d");
    return balltree_build_impl(pos, n_clouds, M, d, n_levels, strides, targets, angle_deg, tree_idx,
                               tree_mask, rot, stream, S, C);
}
