// Register-resident ball-tree builder (power-of-two clouds of <= 64 points): eight waves per
// cloud that each hold the whole cloud and share the independent nodes of a tree level between
// them; the working arrays of std::nth_element live ACROSS THE LANES of two VGPRs
// (lane p = array position p: an order-preserving integer key and the point index), and the
// sequential introselect runs as scalar control flow that reads/writes single lanes with
// v_readlane and single-lane selects.  The two Hoare scans ("while key[lo] < pivot", "while pivot <
// key[hi]") are one ballot + find-first-bit each instead of a dependent LDS round trip per
// element, and the split dimension / key gather of ALL nodes of a tree level is one
// segmented butterfly.  Same algorithm, same permutation as balltree.hip (and as the
// reference's libstdc++ nth_element, models/components/balltree/balltree.h:91-100), ~15x
// less latency: the tree is on the critical path of every forward.
#include "balltree_wave_dev.h"

namespace haet {

__global__ __launch_bounds__(64 * TW) void balltree_wave_kernel(
    const float* __restrict__ pos, int M, int nd, int n_levels, int s0, int s1, int s2, int t0,
    int t1, int t2, int t3, int rotate, Rot3w R, int32_t* __restrict__ tree_idx,
    uint8_t* __restrict__ tree_mask, int32_t* __restrict__ rot, const float* __restrict__ S, int C,
    float* __restrict__ pos_out) {
    const WTreeCfg cfg = {M, nd, n_levels, {s0, s1, s2}, {t0, t1, t2, t3}, rotate, R};
    const WTreeLds none = {-1, -1, {-1, -1, -1, -1}};
    w_tree_build(cfg, pos, S, C, blockIdx.x, gridDim.x, tree_idx, tree_mask, rot, pos_out, none, [](int) {});
}

#ifdef HAET_STAMPS
extern "C" int haet_debug_read_tree_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(haet_tree_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -1;
}
#endif

bool balltree_wave_ok(int M, int n_levels, const int* st) {
    if (M < 2 || M > 64 || (M & (M - 1))) return false;
    for (int i = 0; i < n_levels - 1; ++i)
        if (st[i] < 1 || (st[i] & (st[i] - 1))) return false;
    return true;
}

int balltree_wave_launch(const float* pos, int n_clouds, int M, int d, int n_levels, const int* st,
                         const int* tg, int rotate, const float* R9, int32_t* tree_idx,
                         uint8_t* tree_mask, int32_t* rot, hipStream_t stream, const float* S, int C,
                         float* pos_out) {
    Rot3w R;
    for (int i = 0; i < 9; ++i) R.r[i] = R9[i];
    hipLaunchKernelGGL(balltree_wave_kernel, dim3(n_clouds), dim3(64 * TW), 0, stream, pos, M, d, n_levels,
                       st[0], st[1], st[2], tg[0], tg[1], tg[2], tg[3], rotate, R, tree_idx, tree_mask,
                       rot, S, C, pos_out);
    return check_launch("balltree_wave_kernel");
}

}  // namespace haet
