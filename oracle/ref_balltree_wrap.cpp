// TEST INFRASTRUCTURE ONLY (oracle/_ref build input).
//
// Thin extern "C" shim that #includes the reference's own header-only C++
// ball-tree (models/components/balltree/balltree.h, compiled where it lies
// under /root/reference -- no reference source is copied into this repo) so
// that tests can drive the *real* reference build_tree()/partition_ball_tree()
// (and therefore the real libstdc++ std::nth_element ordering) through ctypes.
// Built by oracle/Makefile into oracle/_ref/libballtree_ref.so.
#include "balltree.h"  // -I/root/reference/models/components/balltree

extern "C" {

// reference: balltree.h:121-186 (build_tree)
void ref_build_tree(const double* data, long* idx_array, long* out_idx,
                    bool* out_mask, long n, long num_features, long max_level) {
    build_tree(data, idx_array, out_idx, out_mask, 0, n, 0, num_features, 0, max_level);
}

// reference: balltree.h:205-259 (partition_ball_tree)
void ref_partition_ball_tree(const double* data, long* idx_array, long* out_idx,
                             long n, long num_features, long target_level) {
    partition_ball_tree(data, idx_array, out_idx, 0, n, 0, num_features, 0, target_level);
}

}  // extern "C"
