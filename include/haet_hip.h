/*
 * haet_hip.h -- C ABI of libhaet_hip.so: the MI355X (gfx950) implementation of
 * HAET's PhysicsAttention hot path.
 *
 * The reference (pedrocurvo/HAET) has no FFI layer of its own; its seams on this
 * path are Python calls into third-party / native code.  Each entry point below
 * replaces one of them (reference file:line given per function).  A maintainer
 * binds these with ctypes; this repo's own binding is
 * haet_amd/_lib.py.
 *
 * Conventions
 *   - plain C, no exceptions: return 0 on success, a negative HAET_E_* code on
 *     error; haet_last_error_string() describes the last error of the calling
 *     thread.
 *   - every pointer is a DEVICE pointer (hipMalloc / torch CUDA tensor storage),
 *     row-major, contiguous, fp32 unless stated; the caller owns all buffers.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*,
 *     NULL = the null stream); no allocation, no synchronisation inside, so the
 *     sequence can be captured into a hipGraph.
 *   - workspaces are sized by the *_workspace_bytes / *_parts queries first.
 *   - single-threaded caller per stream; functions are re-entrant (no globals
 *     except the thread-local error string).
 *
 * Notation: B batch, N points, H heads, C = dim_head, M = slice_num,
 *           T = B*H*M slice tokens, d = token-position dimensionality (2 or 3).
 */
#ifndef HAET_HIP_H
#define HAET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HAET_ABI_VERSION 9  /* 2: + haet_tokens_tree_build, haet_project_split_fwd, haet_*_fwd_split; 4: train-mode
                             BatchNorm (haet_erwin_train), haet_token_stage_fwd; 6: haet_token_stage_fwd_frag (the
                             fragment kernel), haet_erwin_barrier_status, haet_set_deterministic; 7: tile-ordered weight copies in the
                             Erwin blob, haet_conv_presplit / haet_conv_project_cells_*, haet_conv_project_det_fwd; 8: haet_slice_mfma_supported; 9: the W planes
                             of the Linear entry points are followed by their tile-ordered copies (haet_split_bf16_tiled) */

#define HAET_OK 0
#define HAET_E_ARG (-1)      /* bad argument / unsupported shape */
#define HAET_E_LAUNCH (-2)   /* hip launch error */
#define HAET_E_RESOURCE (-3) /* needs more LDS / workspace than available */

typedef void* haet_stream_t;

int haet_abi_version(void);
const char* haet_last_error_string(void);
/* Deterministic mode (process-wide; default: environment HAET_DETERMINISTIC=1, else off): entry points that have a
 * faster variant whose floating-point summation ORDER varies from run to run take the fixed-order variant instead.
 * Today that is the split-k convolution projection of small grids (haet_conv_project_fwd: k-tiles over several
 * workgroups added with float atomics; deterministic mode runs one workgroup per output tile).  The Python host does
 * NOT write this process-wide flag: under torch.use_deterministic_algorithms(True) it calls the per-call entry point
 * haet_conv_project_det_fwd, so an explicit setting made by a C caller or another thread is never overridden.  Every other FORWARD
 * entry point sums in a fixed order always; the backward of the stand-alone point-cloud MPNN
 * (haet_pc_mpnn_message_bwd) adds sender gradients with float atomics and has no fixed-order variant. */
void haet_set_deterministic(int on);
int haet_get_deterministic(void);

/* ------------------------------------------------------------------ per-point projections
 * in_project_x (models/PhysicsAttention/IrregularMesh.py:157-158) and to_out (:235) are
 * plain Linear layers, fp32 cuBLAS GEMMs in the reference.  haet_linear_fwd computes
 *     out[n, o] = sum_k x[n, k] W[o, k] + bias[o]
 * on the bf16 matrix cores as split-bf16 (3 MFMAs per fp32 product, fp32 accumulate,
 * ~2^-16 operand error), which turns the GEMM from compute-bound (fp32 rate) into HBM-bound.
 * W is passed pre-split: haet_split_bf16 writes bf16 hi = RNE(w), lo = RNE(w - hi) planes
 * ([O][K] each) once per weight version.  Supported when haet_linear_supported(K, O) != 0
 * (K % 32 == 0, O % 128 == 0); callers use a library GEMM otherwise.                    */
int haet_split_bf16(const float* w, size_t n, uint16_t* hi, uint16_t* lo, haet_stream_t stream);
/* ABI 9 -- what every Linear entry point of this header expects behind `W_hi` / `W_lo` (haet_linear_*_fwd,
 * haet_project_split_*fwd, haet_linear_swiglu_fwd, haet_deslice_out_fwd_split; the convolution entry points read the
 * row-major planes only): buffers of 2 * O * K bf16 each -- the [O][K] plane as haet_split_bf16 writes it, then a
 * TILE-ORDERED copy of the same values: 16-B pieces (8 consecutive k of one row) at
 *     piece(row, k / 8) = ((row / 16) * (K / 8) + k / 8) * 16 + row % 16.
 * The kernels that keep W stationary in registers fetch one piece per lane and k-step for 16 or 32 rows per wave
 * instruction: 16-32 separate runs of the row-major plane, one or two runs of this copy (the per-launch fixed cost of those
 * kernels, 16-23 us, was this fetch).  O % 16 == 0, K % 8 == 0.                                                   */
int haet_split_bf16_tiled(const float* w /*[O][K]*/, int O, int K, uint16_t* hi /*[2*O*K]*/, uint16_t* lo /*[2*O*K]*/,
                          haet_stream_t stream);
int haet_linear_supported(int K, int O);
int haet_linear_fwd(const float* x /*[Nrows,K]*/, const uint16_t* W_hi, const uint16_t* W_lo,
                    const float* bias /*[O] or NULL*/, int Nrows, int K, int O,
                    float* out /*[Nrows,O]*/, haet_stream_t stream);

/* ------------------------------------------------------------------ slice pass
 * Replaces the materialised Rep-Slice softmax + eidetic einsum of
 * models/PhysicsAttention/IrregularMesh.py:168-188 (Ada-Temp, logits, softmax over
 * the POINT axis, slice_norm, 'bhnc,bhng->bhgc', normalise) by one streaming pass
 * with an online softmax over N.
 *
 * x_lin [B, N, H*C] is the output of in_project_x in its natural layout (the
 * reference's permute to [B,H,N,C], IrregularMesh.py:157-164, is never done).
 *
 * ACCURACY DOMAIN (declared): at the C = 32 shapes the logits come from split-bf16 products (operand error ~2^-16) and
 * enter an exponential, so the relative error of a slice weight grows with the logit: measured end to end against
 * the fp32 reference 6e-6 at the trained-weight scale (|z| up to ~15), <= 1e-3 -- the bar of this path -- for
 * |z| <~ 60, 4e-3 at |z| ~ 300 (tests/test_gpu_parity.py::test_c3_module_small_and_scaled_clouds).
 * Inputs beyond that are served by the exact-f32 families, which every shape can be routed to: HAET_FORCE_F32_MFMA=1
 * (v_mfma_f32_32x32x2_f32, exact f32 products) or HAET_FORCE_VALU=1 (generic fp32 kernels); both are read PER CALL and
 * apply to the forward passes AND to the backward passes below (haet_deslice_bwd_tokens / haet_slice_bwd_points).
 *
 * The pass produces `P = haet_slice_num_parts(...)` partial records, one per
 * point chunk.  A record holds, for every (b,h,g):  m = max_n z, l = sum_n
 * exp(z - m), St[c] = sum_n exp(z - m) x[n,c]  with z = (x.Ws[g] + bs[g] - c0)/tau_n,
 * laid out as  [B*H][ m[M] | l[M] | St[M][C] ]  (B*H*M*(C+2) floats).
 */
int haet_slice_num_parts(int B, int N, int H, int C, int M);
size_t haet_slice_part_floats(int B, int H, int C, int M); /* floats per record */

int haet_slice_fwd(const float* x_lin, const float* Ws /*[M,C]*/, const float* bs /*[M]*/,
                   const float* wt /*[C]*/, float bt, float base_temp, float c0 /*log(-log eps)*/,
                   int B, int N, int H, int C, int M,
                   float* parts /*[P][record]*/, haet_stream_t stream);

/* Log-sum-exp merge of R records (workgroup partials of one GPU, or the
 * all-gathered records of R ranks -- SURVEY.md 8e):
 *   m = max_r m_r;  l = sum_r l_r e^{m_r-m};  St = sum_r St_r e^{m_r-m}.
 * finalize == 0: write one merged record to `out_record`.
 * finalize != 0: also write col_max[B*H*M], col_sum[B*H*M] and the normalised
 *   eidetic states S[B*H*M, C] = (St / l) / (1 + 1e-5)   (IrregularMesh.py:177,186-188;
 *   sum_n softmax == 1). out_record may be NULL then.
 */
int haet_slice_merge(const float* records, int R, int B, int H, int C, int M, int finalize,
                     float* out_record, float* col_max, float* col_sum, float* S,
                     haet_stream_t stream);

/* ------------------------------------------------------------------ token positions
 * models/PhysicsAttention/IrregularMesh.py:204-209: per-channel min/max over ALL T
 * tokens, pos[t, j] = (S[t,j] - min_j) / (max_j - min_j + 1e-8), j < d.            */
int haet_token_pos(const float* S /*[T,C]*/, int T, int C, int d, float* pos /*[T,d]*/,
                   haet_stream_t stream);

/* ------------------------------------------------------------------ ball tree
 * Device replacement of balltree.build_balltree_with_rotations
 * (models/components/balltree/balltree.pyx:598-662 -> balltree.h:32-259), which the
 * reference runs on the HOST with a GPU->CPU->GPU sync every forward.
 * n_clouds = B*H equally sized clouds of M points (batch_idx = arange(B*H)
 * .repeat_interleave(M), IrregularMesh.py:213).  Mp = 2*2^(ceil(log2 M)-1) slots per
 * cloud.  Outputs (device):
 *   tree_idx  int32[n_clouds*Mp]  global token index per slot (libstdc++ nth_element order)
 *   tree_mask uint8[n_clouds*Mp]  0 = padding duplicate
 *   rot       int32[...]          for level i (0..n_levels-1) the rotated re-partition,
 *                                 n_clouds*Mp/prod(strides[:i]) entries each, concatenated;
 *                                 values index positions in tree order at that level.
 *                                 Not written when angle <= 0.
 * targets[i] = max(0, int(log2(total_leaves / ball_sizes[i])))  (balltree.pyx:643) is
 * computed by the caller.  Indices are int32 (T < 2^31); the reference returns int64.
 */
int haet_balltree_build(const float* pos /*[n_clouds*M, d]*/, int n_clouds, int M, int d,
                        int n_levels, const int* strides /*[n_levels-1], host*/,
                        const int* targets /*[n_levels], host*/, float angle_deg,
                        int32_t* tree_idx, uint8_t* tree_mask, int32_t* rot,
                        haet_stream_t stream);

/* haet_token_pos + haet_balltree_build in one launch (what the module calls: the two are
 * back to back on the critical path of every forward, IrregularMesh.py:204-213 followed by
 * erwin_flash.py:161-174).  S [n_clouds*M, C] are the slice tokens, pos [n_clouds*M, d] is
 * WRITTEN (same values as haet_token_pos), the tree outputs are those of haet_balltree_build. */
int haet_tokens_tree_build(const float* S, int C, float* pos, int n_clouds, int M, int d,
                           int n_levels, const int* strides, const int* targets, float angle_deg,
                           int32_t* tree_idx, uint8_t* tree_mask, int32_t* rot,
                           haet_stream_t stream);

/* ------------------------------------------------------------------ Erwin stage
 * Replaces ErwinTransformer.forward with embed=False, mp_steps=0
 * (models/components/erwinflash/erwin_flash.py:138-195; BasicLayer layers.py:72-92;
 * ErwinTransformerBlock layers.py:30-32; BallMSA attention.py:42-91 incl. the
 * flash_attn_qkvpacked_func call :82-84; SwiGLU mlp.py:8-18; BallPooling/
 * BallUnpooling pooling.py:25-72) for eval-mode BatchNorm, one workgroup per cloud.
 *
 * `desc` (host) describes the network and is used for validation and LDS sizing; `weights`
 * (device) is ONE packed blob that BEGINS WITH A COPY OF *desc (sizeof(haet_erwin_desc) bytes,
 * padded to 16) followed by the tensors, laid out by haet_amd/components/erwinflash/pack.py.
 * Returns HAET_E_RESOURCE when the token state of one cloud does not fit the 160 KiB LDS of a CU.
 */
#define HAET_ERWIN_MAX_LEVELS 4
#define HAET_ERWIN_MAX_DEPTH 4

typedef struct {
    /* offsets (in floats) from the start of the blob.  Linear weights keep PyTorch's [out][in]
     * layout, rows zero-padded to in~ = roundup(in, 32), pre-split into two bf16 planes: hi =
     * RNE(w) at the offset, lo = RNE(w - hi) out*in~ bf16 later (one lane of the MFMA B operand
     * reads 8 consecutive k of one output row) -- what haet_erwin_frag_pack and the fragment kernel's
     * resampling read -- FOLLOWED (2*out*in~ bf16 from the offset) by a tile-ordered copy of both planes,
     * which is what haet_erwin_fwd / haet_token_stage_fwd stream (ABI 7): for every (16-row tile ot,
     * 32-wide k-step ks, plane) one 1-KB piece, order [ot][ks][plane], in which the 16 bytes at 16*(16*kq + r)
     * are W[16 ot + r][32 ks + 8 kq .. + 8]; rows past `out` are zero: roundup(out,16)*in~*2 bf16.
     * pe_w is fp32, transposed [d][c].                                                          */
    int32_t norm1, norm2;            /* [c] */
    int32_t pe_w, pe_b;              /* [d][c], [c] */
    int32_t qkv_w, qkv_b;            /* [3c][c~], [3c] */
    int32_t proj_w, proj_b;          /* [c][c~], [c] */
    int32_t w1_w, w1_b, w2_w, w2_b;  /* [hidden][c~], [hidden] */
    int32_t w3_w, w3_b;              /* [c][hidden~], [c] */
    /* norm1, norm2, pe_w, pe_b and the five biases are packed contiguously (each 16-B aligned) in
     * blob[small_off, small_off + small_len): the kernel copies that group to LDS per block */
    int32_t small_off, small_len;
} haet_erwin_block_w;

typedef struct {
    int32_t c, heads, ball, depth, hidden; /* channels, heads, ball size, #blocks, SwiGLU width */
    haet_erwin_block_w blocks[HAET_ERWIN_MAX_DEPTH];
} haet_erwin_layer;

typedef struct {
    int32_t stride;             /* 1 in a pool: "no pooling", the node passes through (pooling.py:26-27) */
    int32_t w, b;               /* Linear, [out][K~] / [out] */
    int32_t bn_scale, bn_shift; /* eval BatchNorm folded to y*scale+shift, [c_out]; with desc.bn_train: gamma, beta */
    int32_t small_off, small_len; /* b | bn_scale | bn_shift, contiguous like above */
    float bn_eps, bn_momentum;  /* nn.BatchNorm1d eps / momentum (train mode) */
} haet_erwin_resample;

typedef struct {
    int32_t n_levels; /* encoder layers + bottleneck (>= 1) */
    int32_t d;        /* position dimensionality (2 or 3) */
    int32_t rotate;   /* 1: odd blocks run on the rotated partition */
    int32_t decode;   /* must be 1 */
    int32_t bn_train; /* 1: BatchNorm uses batch statistics over the tokens of ALL clouds (module.train()) and the
                         launch needs a haet_erwin_train; 0: running statistics, folded by pack.py */
    haet_erwin_layer enc[HAET_ERWIN_MAX_LEVELS];       /* enc[n_levels-1] is the bottleneck */
    haet_erwin_layer dec[HAET_ERWIN_MAX_LEVELS];       /* dec[i] runs at level i (i < n_levels-1) */
    haet_erwin_resample pool[HAET_ERWIN_MAX_LEVELS];   /* level i -> i+1 */
    haet_erwin_resample unpool[HAET_ERWIN_MAX_LEVELS]; /* level i+1 -> i */
} haet_erwin_desc;

size_t haet_erwin_lds_bytes(const haet_erwin_desc* desc, int Mp);

/* Train-mode BatchNorm plumbing (BallPooling / BallUnpooling `norm` under module.train(), pooling.py:23,38,56,68:
 * batch statistics over the tokens of ALL clouds, running statistics updated with `momentum`, num_batches_tracked
 * incremented).  Slot 2l = pool of level l, 2l+1 = unpool of level l.  The clouds' workgroups meet at a grid
 * barrier per BatchNorm, so a train-mode launch takes at most haet_resident_workgroup_limit() clouds (256 on
 * MI355X; HAET_E_RESOURCE beyond) -- see the residency contract below. */
typedef struct {
    float* running_mean[2 * HAET_ERWIN_MAX_LEVELS];          /* device, updated in place; NULL: slot unused / not tracked */
    float* running_var[2 * HAET_ERWIN_MAX_LEVELS];
    long long* num_batches_tracked[2 * HAET_ERWIN_MAX_LEVELS];
    float* batch_stats; /* out (may be NULL): [2*HAET_ERWIN_MAX_LEVELS][2][cmax] batch mean, 1/sqrt(var+eps); cmax = max channels */
    void* workspace;    /* device, haet_erwin_train_workspace_bytes() bytes, caller-owned scratch */
    /* Tape for haet_erwin_bwd (also honoured with desc->bn_train == 0): the input tokens of every op of the stage
     * (blocks, pools, unpools in execution order), [n_clouds][n_ops][tape_op_floats] fp32; NULL: not recorded.
     * tape_op_floats >= max over the levels of tokens * channels (haet_erwin_tape_floats gives n_ops * that). */
    float* tape;
    int32_t tape_op_floats;
} haet_erwin_train;

size_t haet_erwin_train_workspace_bytes(const haet_erwin_desc* desc, int n_clouds);
/* Residency contract of the train-mode launches (haet_erwin_fwd / haet_token_stage_fwd with desc->bn_train,
 * haet_erwin_bwd with desc->bn_train): one workgroup per cloud, and ALL of them must be resident at once for the
 * grid barriers to complete -- n_clouds <= haet_resident_workgroup_limit() (the device's compute-unit count; checked,
 * HAET_E_RESOURCE beyond) AND a device that is otherwise idle while the launch runs (no other stream or process
 * holding compute units: not checked, cannot be).  A barrier whose workgroups do not all arrive gives up after ~1 s
 * instead of hanging the GPU and sets the workspace's time-out word (bytes 4..7; sticky: launches clear only the
 * counter in bytes 0..3); the results of that launch are invalid.  The caller copies the first 8 bytes of the
 * workspace to the host at its next synchronisation point and passes them to haet_erwin_barrier_status:
 * HAET_OK, or HAET_E_LAUNCH with the explanation in haet_last_error_string(); to re-arm, zero bytes 4..7. */
int haet_resident_workgroup_limit(void);
int haet_erwin_barrier_status(const void* workspace_first_8_bytes_on_host);

int haet_erwin_fwd(const float* tokens /*[n_clouds*M, c0]*/, const float* pos /*[n_clouds*M, d]*/,
                   const int32_t* tree_idx, const uint8_t* tree_mask, const int32_t* rot,
                   const float* weights, size_t weight_floats /* length of the blob */,
                   const haet_erwin_desc* desc, const haet_erwin_train* train /* NULL unless desc->bn_train */,
                   int n_clouds, int M, float* out /*[n_clouds*M, c0]*/, haet_stream_t stream);

/* The whole token stage in ONE launch: haet_token_pos + haet_balltree_build + haet_erwin_fwd
 * (IrregularMesh.py:196-218 = positions from the tokens' first d channels, build_balltree_with_rotations,
 * ErwinTransformer.forward) for power-of-two clouds of <= 64 tokens (haet_token_stage_supported).  S [n_clouds*M, c0]
 * are the slice tokens; the tree stays on the CU.  pos_out [n_clouds*M, d], tree_idx / tree_mask [n_clouds*M] and
 * rot (levels concatenated, as haet_balltree_build) are optional outputs (NULL: not written).  `angle_deg` is the
 * module's `rotate` (<= 0 with desc->rotate == 0). */
int haet_token_stage_supported(const haet_erwin_desc* desc, int n_clouds, int M);
int haet_token_stage_fwd(const float* S, int n_clouds, int M, float angle_deg, const float* weights,
                         size_t weight_floats, const haet_erwin_desc* desc, const haet_erwin_train* train,
                         float* pos_out, int32_t* tree_idx, uint8_t* tree_mask, int32_t* rot, float* out,
                         haet_stream_t stream);

/* The same launch on the FRAGMENT kernel (csrc/erwin2.hip; eval-mode BatchNorm, no tape): a transformer block is two
 * barrier-separated phases of wave-owned work items -- (32-token tile, head) through q/k/v, ball attention and its
 * slice of the proj product; (tile, 32 hidden units) through w1/w2, the SwiGLU gate and its slice of the w3 product --
 * whose operands stay in the wave's registers in the MFMA accumulator layout; only split-K partial sums go through
 * LDS.  It reads the Linear weights from a second blob laid out in MFMA fragment order for exactly that data flow:
 *   haet_erwin_frag_bytes   size of that blob for (desc, M), 0 = shape outside the kernel's envelope (head dim 8,
 *                           32 / 64 channels, 32 | tokens per level, one wave per work item, stride 2, power-of-two
 *                           clouds <= 64 tokens: BASELINE config C3) -- or HAET_ERWIN_V1 set in the environment;
 *   haet_erwin_frag_pack    builds it on the device from the packed blob `weights` of haet_erwin_fwd (call once per
 *                           weight version; stream-ordered, no synchronisation);
 *   haet_token_stage_fwd_frag  same arguments and results as haet_token_stage_fwd (train = NULL) plus the blob.
 * `weights` is still read for the small fp32 parameters and the pooling / unpooling Linears. */
size_t haet_erwin_frag_bytes(const haet_erwin_desc* desc, int M);
int haet_erwin_frag_pack(const float* weights, size_t weight_floats, const haet_erwin_desc* desc, int M, void* frags,
                         size_t frag_bytes, haet_stream_t stream);
int haet_token_stage_fwd_frag(const float* S, int n_clouds, int M, float angle_deg, const float* weights,
                              size_t weight_floats, const void* frags, size_t frag_bytes, const haet_erwin_desc* desc,
                              float* pos_out, int32_t* tree_idx, uint8_t* tree_mask, int32_t* rot, float* out,
                              haet_stream_t stream);

/* ------------------------------------------------------------------ Erwin stage, backward
 * Backward of haet_erwin_fwd / haet_token_stage_fwd (the training scripts call loss.backward(),
 * exp_airfoil.py:249-255) in ONE launch, one workgroup per cloud: every block / pool / unpool is recomputed from its
 * taped input (haet_erwin_train.tape) in fp32 and differentiated -- RMSNorm, rel-pos PE, QKV, per-ball softmax
 * attention, proj, SwiGLU, BallPooling / BallUnpooling with eval- or train-mode BatchNorm (train: the cross-cloud
 * terms of the BatchNorm backward meet at a grid barrier, <= haet_resident_workgroup_limit() clouds, idle device).  Positions carry no gradient (the
 * reference computes rel_pos / centres under no_grad: attention.py:35, pooling.py:29,59).
 *
 * Weights come as ONE fp32 blob that BEGINS WITH haet_erwin_f32_layout (offsets in floats of every tensor, original
 * [out][in] layout `*_w` and its transpose `*_wT`; ABI 7: each of a block's ten Linear tensors [R][K] is followed by a
 * tile-ordered copy of itself of the same size -- per (32-row tile ot, 16-wide k-step s) two 1-KB pieces, order
 * [ot][s][which], whose 16 bytes at 16 * (r + 32 hf) are W[32 ot + r][16 s + 8 hf + 4 which .. + 4] -- which is what the
 * matrix-core phases stream), laid out by haet_amd/components/erwinflash/pack.py; `grads`
 * [n_clouds][layout.total] receives every parameter gradient of each cloud at the SAME offsets (the caller sums over
 * the clouds; the `*_wT` / statistics slots stay untouched). */
typedef struct {
    int32_t norm1, norm2, pe_w /* [c][d] */, pe_b, qkv_w, qkv_wT, qkv_b, proj_w, proj_wT, proj_b,
            w1_w, w1_wT, w1_b, w2_w, w2_wT, w2_b, w3_w, w3_wT, w3_b;
} haet_erwin_block_f32;
typedef struct {
    int32_t w, wT, b, gamma, beta, rmean, rvar;
} haet_erwin_resample_f32;
typedef struct {
    int32_t total;  /* floats in the blob (header included) */
    int32_t pad[3];
    haet_erwin_block_f32 enc[HAET_ERWIN_MAX_LEVELS][HAET_ERWIN_MAX_DEPTH];
    haet_erwin_block_f32 dec[HAET_ERWIN_MAX_LEVELS][HAET_ERWIN_MAX_DEPTH];
    haet_erwin_resample_f32 pool[HAET_ERWIN_MAX_LEVELS];
    haet_erwin_resample_f32 unpool[HAET_ERWIN_MAX_LEVELS];
} haet_erwin_f32_layout;

size_t haet_erwin_tape_floats(const haet_erwin_desc* desc, int M, int* n_ops, int* op_floats); /* per cloud */
int haet_erwin_bwd_supported(const haet_erwin_desc* desc, int n_clouds, int M);
int haet_erwin_bwd(const float* g_out /*[n_clouds*M, c0] dL/d tokens'*/, const float* tape, const float* pos,
                   const int32_t* tree_idx, const uint8_t* tree_mask, const int32_t* rot,
                   const float* weights_f32 /* begins with haet_erwin_f32_layout */,
                   const float* weights_fwd /* the forward's blob (begins with the descriptor) */,
                   const haet_erwin_desc* desc, const float* bn_batch_stats /* train: haet_erwin_train.batch_stats */,
                   void* workspace /* train: haet_erwin_train_workspace_bytes() */, int n_clouds, int M,
                   float* g_in /*[n_clouds*M, c0] dL/d tokens*/, float* grads /*[n_clouds][layout.total]*/,
                   haet_stream_t stream);

/* ------------------------------------------------------------------ de-slice + to_out in one pass
 * Replaces einsum('bhgc,bhng->bhnc') + rearrange + self.to_out[0] (models/PhysicsAttention/IrregularMesh.py:221-235)
 * for the fused operand format: out [B, N, O] = to_out( de-slice(cells, s; col_max, col_sum, S2) ); the [B, N, H*C]
 * de-sliced tensor is never written.  Whi / Wlo / bias: the to_out Linear as haet_split_bf16 planes [O][H*C].
 * Envelope: H = 8, C = 32, O = 256, M in {32, 64} (haet_deslice_out_supported). */
int haet_deslice_out_supported(int H, int C, int M, int O);
int haet_deslice_out_fwd_split(const float* cells, const float* s, const float* Ws, const float* bs, float c0,
                               const float* col_max, const float* col_sum, const float* S2,
                               const uint16_t* Whi, const uint16_t* Wlo, const float* bias,
                               const float* residual /* [B, N, O] added to the output (the block's `+ fx`,
                               HAETransolver_Irregular_Mesh.py:125), or NULL */,
                               int B, int N, int H, int C, int M, int O, float* out, haet_stream_t stream);

/* ------------------------------------------------------------------ de-slice pass
 * Replaces einsum('bhgc,bhng->bhnc') + rearrange 'b h n d -> b n (h d)'
 * (models/PhysicsAttention/IrregularMesh.py:228-232): recomputes the slice weights
 * w[n,g] = exp(z - col_max[g]) / col_sum[g] from x_lin and writes
 * out_x[b,n,h*C+c] = sum_g w[n,g] * S2[b,h,g,c]   directly in [B,N,H*C].            */
int haet_deslice_fwd(const float* x_lin, const float* Ws, const float* bs, const float* wt,
                     float bt, float base_temp, float c0,
                     const float* col_max /*[B*H*M]*/, const float* col_sum /*[B*H*M]*/,
                     const float* S2 /*[B*H*M, C]*/, int B, int N, int H, int C, int M,
                     float* out_x /*[B,N,H*C]*/, haet_stream_t stream);

/* Whether haet_slice_fwd / haet_deslice_fwd run (B, N, H, C, M) on the split-bf16 matrix-core kernels (C = 32 with even
 * H <= 16, C = 64 with H * M / 32 <= 8; M in {32, 64, 128}) rather than on the generic vector-ALU ones.  The host side
 * uses it to run dim_head 16 -- the reference's PDE-benchmark width, n_hidden 128 / 8 heads -- as dim_head 32 with 16
 * zero channels per head (x_lin padded, W_s / w_t zero-padded: exact), haet_amd/PhysicsAttention/_base.py.        */
int haet_slice_mfma_supported(int B, int N, int H, int C, int M);

/* ------------------------------------------------------------------ fused operand format
 * The C = 32 fast path of the module keeps x_lin in the streaming passes' own operand format
 * instead of fp32: per 4 channels 16 B = {4 x bf16 hi, 4 x bf16 lo} (hi = RNE(x), lo = RNE(x - hi)),
 * row stride H*C*4 B like the fp32 tensor, plus s[n][h] = log2(e) / tau(n,h) (Ada-Temp,
 * IrregularMesh.py:168).  haet_project_split_fwd is in_project_x (IrregularMesh.py:157-164) writing
 * that format; the two *_split passes are haet_slice_fwd / haet_deslice_fwd reading it (same
 * records, same outputs).  The projection GEMM has vector-issue slack, the passes do not.        */
int haet_project_split_supported(int K, int O, int Nrows);
int haet_project_split_fwd(const float* x /*[Nrows,K]*/, const uint16_t* W_hi, const uint16_t* W_lo,
                           const float* bias /*[O] or NULL*/, const float* wt /*[32] ada_temp weight*/,
                           float bt, float base_temp, int Nrows, int K, int O,
                           void* x_split /*[Nrows, O] 4-B cells*/, float* s_out /*[Nrows, O/32]*/,
                           haet_stream_t stream);
int haet_slice_split_supported(int B, int N, int H, int C, int M);
int haet_slice_fwd_split(const void* x_split, const float* s, const float* Ws, const float* bs, float c0,
                         int B, int N, int H, int C, int M, float* parts, haet_stream_t stream);
int haet_deslice_fwd_split(const void* x_split, const float* s, const float* Ws, const float* bs,
                           float c0, const float* col_max, const float* col_sum, const float* S2,
                           int B, int N, int H, int C, int M, float* out_x, haet_stream_t stream);

/* ------------------------------------------------------------------ backward of the two passes
 * The N-scaled part of loss.backward() through Physics_Attention_* (the reference relies on torch
 * autograd over models/PhysicsAttention/IrregularMesh.py:168-188,228; scripts: exp_airfoil.py:249-255).
 * The slice weights w[n,g] are recomputed from x_lin and the forward's column statistics.
 *   g_out = dL/d out_x [B,N,H*C]  (gradient arriving at haet_deslice_fwd's output)
 * pass 1, haet_deslice_bwd_tokens: partial sums of dS2[b,h,g,:] = sum_n w[n,g] g_out[b,n,h,:]
 *   -> parts [P][B*H*M*C], P = haet_slice_bwd_num_parts(...); the caller sums over P and runs the
 *   Erwin backward, which yields dS = dL/dS (S = the normalised eidetic states of haet_slice_merge).
 * pass 2, haet_slice_bwd_points: with A[b,h,g] = dS2[g].S2[g] + S[g].dS[g] from the caller,
 *   dx_lin [B,N,H*C] (complete gradient w.r.t. x_lin: de-slice + slice + Ada-Temp terms) and
 *   param_parts [P][B*H][M*C + M + C + 1] = per-workgroup partials of
 *   (d in_project_slice.weight [M,C], d in_project_slice.bias [M], d ada_temp_linear.weight [C],
 *   d ada_temp_linear.bias), to be summed over P and B*H by the caller.
 * ARITHMETIC / ACCURACY (declared).  At C = 32 with M in {32, 64} (the C3 shape) both passes run on split-bf16 products
 * (slice_bwd_x3.hip: 3 bf16 MFMAs per fp32 product, fp32 accumulate), so the gradients carry the forward's logit envelope
 * (|z| * 2^-16 inside the exponential).  Measured against fp64 autograd on the kernel-level shapes of
 * tests/test_gpu_backward.py (unit-scale inputs, |z| <~ 10; the tests print and bound these): dS2 <= 3e-4, dx_lin / dW_s /
 * db_s <= 1.5e-3, dw_t / db_t <= 7.5e-3 relative L2 (the last two are sums of clamp-masked terms with heavy cancellation).
 * The exact-f32 matrix-core kernels (slice_bwd_mfma.hip: v_mfma_f32_32x32x2_f32; what M = 128 and C = 64 always run)
 * hold 5e-5 / 5e-4 / 2.5e-3 on the same shapes and are selected for every shape by HAET_BWD_F32=1 (backward only) or
 * HAET_FORCE_F32_MFMA=1 (forward and backward), the generic fp32 vector kernels by HAET_FORCE_VALU=1; all read per call. */
int haet_slice_bwd_num_parts(int B, int N, int H, int C, int M);
size_t haet_slice_bwd_param_floats(int C, int M);
int haet_deslice_bwd_tokens(const float* x_lin, const float* Ws, const float* bs, const float* wt,
                            float bt, float base_temp, float c0, const float* col_max,
                            const float* col_sum, const float* g_out, int B, int N, int H, int C, int M,
                            float* parts, haet_stream_t stream);
int haet_slice_bwd_points(const float* x_lin, const float* Ws, const float* bs, const float* wt,
                          float bt, float base_temp, float c0, const float* col_max,
                          const float* col_sum, const float* g_out, const float* S2 /*[B*H*M, C]*/,
                          const float* dS /*[B*H*M, C]*/, const float* A /*[B*H*M]*/,
                          int B, int N, int H, int C, int M, float* dx_lin, float* param_parts,
                          haet_stream_t stream);

/* Weight gradient of a Linear over the points (to_out, in_project_x): per-workgroup partial records of
 *   dW[O][K] = sum_n G[n][o] X[n][k],  db[o] = sum_n G[n][o]        (G = dL/d output [N,O], X = the input [N,K])
 * parts [P][(O/256)*(K/256)][256*256 + 256]: per 256 x 256 block of dW (row-major blocks, o-block major)
 * the block and its 256 entries of db; the caller sums over P.  O and K must be multiples of 128 (a 256-wide
 * block that sticks out of the matrix holds zeros there; ceil(O/256) * ceil(K/256) blocks).                  */
int haet_linear_wgrad_supported(int O, int K);
int haet_linear_wgrad_num_parts(int N);
int haet_linear_wgrad(const float* G, const float* X, int N, int O, int K, float* parts, haet_stream_t stream);
/* One tap (dh, dw, dd) of the weight gradient of the structured variants' convolution projection
 * (nn.Conv2d / nn.Conv3d, same padding, stride 1; 2-D: gD = 1, dd = 0):
 *   dW[o][c][tap] = sum_n g[n][o] x[n + (dh gW + dw) gD + dd][c]   (zero where the neighbour is outside the grid)
 * over the channels-last grid; parts as haet_linear_wgrad with K = C_in (its db entries are sum_n g, tap-independent). */
int haet_conv_wgrad_tap(const float* G, const float* X, int B, int gH, int gW, int gD, int O, int Cin,
                        int dh, int dw, int dd, float* parts, haet_stream_t stream);

/* ------------------------------------------------------------------ the block around the path
 * Point-wise callers either side of Physics_Attention inside a Transolver block
 * (models/HAETransolver_Irregular_Mesh.py:114-134: LayerNorm -> Attn -> +res -> LayerNorm -> MLP -> +res;
 * preprocess MLP :186, 338-376); forward kernels, plus the LayerNorm backward for training.
 *   haet_linear_act_fwd:    haet_linear_fwd with an epilogue activation (act 0 = none, 1 = GELU erf): the
 *                           Linear + nn.GELU pairs of models/components' MLP
 *   haet_layernorm_fwd:     nn.LayerNorm(D) over rows (biased variance, eps inside the root)
 *   haet_linear_narrow_fwd: Linear (+ GELU) with K <= 16 inputs (coordinates + fields), fp32 vector ALUs  */
int haet_linear_act_fwd(const float* x, const uint16_t* W_hi, const uint16_t* W_lo, const float* bias,
                        int Nrows, int K, int O, int act, float* out, haet_stream_t stream);
/* nn.LayerNorm(K) fused into the loader of the GEMM (ln_1 in front of in_project_x, ln_2 in front of the block
 * MLP): the envelope of haet_project_split_supported (K in {128, 256}, O % 256 = 0, >= 64 rows).        */
int haet_linear_ln_fwd(const float* x, const float* ln_gamma, const float* ln_beta, float ln_eps,
                       const uint16_t* W_hi, const uint16_t* W_lo, const float* bias, int Nrows, int K, int O,
                       int act, float* out, haet_stream_t stream);
int haet_project_split_ln_fwd(const float* x, const float* ln_gamma, const float* ln_beta, float ln_eps,
                              const uint16_t* W_hi, const uint16_t* W_lo, const float* bias, const float* wt,
                              float bt, float base_temp, int Nrows, int K, int O, void* x_split, float* s_out,
                              haet_stream_t stream);
/* SwiGLU pair in one GEMM (components/mlp.py:17-18): W rows INTERLEAVED (row 2j = w1[j], row 2j + 1 = w2[j]; bias
 * likewise), H [Nrows][O / 2] = (x w2^T + b2) * silu(x w1^T + b1); the envelope of haet_project_split_supported.     */
int haet_linear_swiglu_supported(int K, int O, int Nrows);
/* rms_w != NULL: nn.RMSNorm(K, eps = rms_eps) with that weight is applied to the rows of x in the loader (layers.py:31) */
int haet_linear_swiglu_fwd(const float* x, const float* rms_w, float rms_eps, const uint16_t* W_hi, const uint16_t* W_lo,
                           const float* bias, int Nrows, int K, int O, float* H, haet_stream_t stream);
int haet_layernorm_supported(int D);
int haet_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, int rows, int D,
                       float* out, haet_stream_t stream);
/* sum = x + res and out = LayerNorm(sum) in one pass (the block's residual add feeding its next LayerNorm) */
int haet_add_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float eps,
                           int rows, int D, float* sum, float* out, haet_stream_t stream);
/* out [rows, OD] = Linear(LayerNorm(x (+ res))) with a narrow output (OD <= 16, D = 256): the model's last layer,
 * mlp2(ln_3(.)) (HAETransolver_Irregular_Mesh.py:129-132); neither the sum nor the normalised rows are written. */
int haet_ln_linear_small_supported(int D, int OD);
int haet_ln_linear_small_fwd(const float* x, const float* res /*or NULL*/, const float* gamma, const float* beta,
                             float eps, const float* W /*[OD, D]*/, const float* bias, int rows, int D, int OD,
                             float* out, haet_stream_t stream);
/* backward of haet_layernorm_fwd: dx [rows, D] and per-workgroup partials parts [P][2 D] = {dgamma, dbeta}
 * (P = haet_layernorm_bwd_num_parts(rows)); the statistics are recomputed from x.                        */
int haet_layernorm_bwd_num_parts(int rows);
int haet_layernorm_bwd(const float* x, const float* gamma, const float* gy, float eps, int rows, int D,
                       float* dx, float* parts, haet_stream_t stream);
int haet_linear_narrow_supported(int K, int O);
int haet_linear_narrow_fwd(const float* x, const float* W /*[O,K]*/, const float* bias, int rows, int K, int O,
                           int act, float* out, haet_stream_t stream);

/* ------------------------------------------------------------------ structured-mesh projection
 * in_project_x of the structured modules (nn.Conv2d / nn.Conv3d, kernel k, padding k/2:
 * StructuredMesh2D.py:109,173-189, StructuredMesh3D.py:113,177-193) as an implicit GEMM on the
 * matrix cores over the channels-last grid x [B, gH*gW*gD, C_in] -> out [B, gH*gW*gD, O]
 * (2-D: gD = kD = 1).  W_hi / W_lo: haet_split_bf16 of the weight re-laid as
 * [O][C_in/32][kH*kW*kD][32] (channel block, then tap, then the block's 32 channels).              */
/* (grids of a few thousand points: the k-tiles of an output tile are split over several workgroups whose partial sums
 * are added with float atomics -- the fp32 summation order, hence the last bit, can vary from run to run there;
 * HAET_CONV_NO_SPLITK=1 keeps one workgroup per tile)                                                         */
int haet_conv_project_supported(int Cin, int O);
int haet_conv_project_fwd(const float* x, const uint16_t* W_hi, const uint16_t* W_lo, const float* bias,
                          int B, int gH, int gW, int gD, int Cin, int O, int kH, int kW, int kD,
                          float* out, haet_stream_t stream);
/* Same contract, fixed summation order as a PER-CALL choice: never the split-k / float-atomics variant, independent of
 * (and without touching) the process-wide haet_set_deterministic.  Two calls on the same operands are bit-identical. */
int haet_conv_project_det_fwd(const float* x, const uint16_t* W_hi, const uint16_t* W_lo, const float* bias,
                              int B, int gH, int gW, int gD, int Cin, int O, int kH, int kW, int kD,
                              float* out, haet_stream_t stream);
/* ---- Round 4: the wide convolution (O = 256) around LDS-DMA (csrc/conv8.hip).  Same arithmetic (split-bf16 x 3, fp32
 * accumulate), same weight planes; the ACTIVATIONS arrive pre-split, once per forward instead of once per tap:
 *   x_cells [(rows + 1)][C_in] 4-byte slots: per row and 32-channel block 128 B = {32 x bf16 hi | 32 x bf16 lo}
 *   (hi = RNE(x), lo = RNE(x - hi)); row `rows` is all zeros -- the padding every out-of-grid tap reads.
 * haet_conv_presplit writes that buffer (haet_conv_cells_bytes(rows, C_in) bytes, caller-owned) from fp32 rows and can
 * apply nn.LayerNorm(C_in) to each row on the way (ln_g / ln_b non-NULL; C_in = 256): ln_1 in front of the projection
 * (models/HAETransolver_Structured_Mesh_3D.py block; StructuredMesh3D.py:177-193) never exists in HBM.
 * haet_conv_project_cells_fwd / _split_fwd: the contracts of haet_conv_project_fwd / haet_conv_project_split_fwd on that
 * operand.  Workgroup tile 256 rows x 256 columns, both operands HBM/L2 -> LDS by global_load_lds_dwordx4, two 64-KB
 * stages, one barrier per 32-channel k-tile, fixed summation order.  Supported: haet_conv_cells_supported (C_in % 32
 * = 0, O = 256), odd kernels, rows < 2^31 - 256.                                                                  */
int haet_conv_cells_supported(int Cin, int O);
size_t haet_conv_cells_bytes(long long rows, int Cin);
/* fp16x2 != 0 (opt-in, see the "fp16x2" note below): the cells hold fp16 hi / fp16 lo instead of bf16, W_hi is the weight as
 * ONE fp16 plane (haet_conv_project_f16_fwd's W_f16) and W_lo is ignored (may be NULL): 2 products per term, ~2e-4.
 * The pre-split pass and the convolution must be called with the same setting.                                   */
int haet_conv_presplit(const float* x /*[rows, C_in]*/, long long rows, int Cin, void* x_cells,
                       const float* ln_g /*[C_in] or NULL*/, const float* ln_b, float ln_eps, int fp16x2,
                       haet_stream_t stream);
int haet_conv_project_cells_fwd(const void* x_cells, const uint16_t* W_hi, const uint16_t* W_lo, const float* bias,
                                int B, int gH, int gW, int gD, int Cin, int O, int kH, int kW, int kD,
                                float* out, int fp16x2, haet_stream_t stream);
int haet_conv_project_cells_split_fwd(const void* x_cells, const uint16_t* W_hi, const uint16_t* W_lo, const float* bias,
                                      const float* wt, float bt, float base_temp, int B, int gH, int gW, int gD, int Cin,
                                      int O, int kH, int kW, int kD, void* cells, float* s_out, int fp16x2,
                                      haet_stream_t stream);
/* The same convolution with Ada-Temp in its epilogue and the output in the fused operand format of the streaming
 * passes (what haet_project_split_fwd writes for the Linear projection; StructuredMesh2D.py:173-181,
 * StructuredMesh3D.py:177-185): cells [B*N][O] 4-byte slots -- per 4 channels {4 x bf16 hi, 4 x bf16 lo} -- and
 * s [B*N][O / 32] = log2(e) / tau, tau = base_temp + clamp(x_lin[head] . wt + bt, -0.4, 0.4), dim_head = 32.
 * Always the one-workgroup-per-tile path (no float atomics). */
int haet_conv_project_split_fwd(const float* x, const uint16_t* W_hi, const uint16_t* W_lo, const float* bias,
                                const float* wt, float bt, float base_temp, int B, int gH, int gW, int gD, int Cin,
                                int O, int kH, int kW, int kD, void* cells, float* s_out, haet_stream_t stream);
/* Opt-in reduced-term arithmetic for the convolution ("fp16x2": BASELINE config C5 names fp16 MFMA; the reference's own
 * GPU path is fp16 autocast, velocity-mem-checker.py:49): activations as fp16 hi + fp16 lo (split in the loader), the
 * weight as ONE fp16 plane W_f16 (same [O][C_in/32][taps][32] layout) -- 2 matrix-core products per term instead of 3.
 * Measured ~2e-4 relative against fp64 (split-bf16: 4e-6); north_star's bar is 1e-3.  The default everywhere stays
 * split-bf16; the Python host takes these entry points under haet_amd.set_precision("fp16x2") / HAET_PRECISION=fp16x2.
 * RANGE CONTRACT: every weight and activation must lie inside the fp16 range, |v| <= 65504 (beyond it hi = inf and
 * lo = v - inf: the output is NaN; below ~6e-5 fp16 is subnormal and the operand loses bits).  The Python host checks
 * the weight once per version and keeps a layer whose weight is out of range on bf16x3; activations are the caller's. */
int haet_conv_project_f16_fwd(const float* x, const uint16_t* W_f16, const float* bias, int B, int gH, int gW, int gD,
                              int Cin, int O, int kH, int kW, int kD, float* out, haet_stream_t stream);
int haet_conv_project_split_f16_fwd(const float* x, const uint16_t* W_f16, const float* bias, const float* wt, float bt,
                                    float base_temp, int B, int gH, int gW, int gD, int Cin, int O, int kH, int kW, int kD,
                                    void* cells, float* s_out, haet_stream_t stream);

/* ------------------------------------------------------------------ stand-alone ErwinFlash on raw point clouds
 * (ErwinTransformer.forward with embed=True / mp_steps>0 on 32k..1M-point clouds,
 * models/components/erwinflash/erwin_flash.py:138-195, benchmarks/02.1-Car-Design-ShapeNetCar/main.py:50-99).
 *
 * haet_balltree_big_build replaces build_balltree_with_rotations (balltree.pyx:598-662) for clouds of any size
 * (>= 2 points each, unequal sizes, padded trees): cloud_sizes is a HOST array; tree_idx / tree_mask have
 * haet_balltree_big_slots() entries (cloud after cloud, 2^ceil(log2 n) slots each), rot holds the rotated
 * re-partitions of the n_levels layers one after the other (slots, slots / stride0, ...), written only when
 * angle_deg > 0.  Same permutation as the reference (libstdc++ introselect followed step for step, its Hoare
 * partition data-parallel).  The call synchronises the stream (the reference synchronises through the host too). */
size_t haet_balltree_big_slots(const int32_t* cloud_sizes, int n_clouds);
size_t haet_balltree_big_workspace_bytes(const int32_t* cloud_sizes, int n_clouds, int d);
int haet_balltree_big_build(const float* pos, const int32_t* cloud_sizes, int n_clouds, int d, int n_levels,
                            const int32_t* strides, const int32_t* ball_sizes, float angle_deg, int32_t* tree_idx,
                            uint8_t* tree_mask, int32_t* rot, void* workspace, size_t workspace_bytes,
                            haet_stream_t stream);
/* torch_cluster.radius_graph(pos, r, batch, loop=True, max_num_neighbors) (erwin_flash.py:160-166): cloud_lo / cloud_hi
 * [N] = first / one-past-last point of each centre's cloud; nbr [N][max_nbrs], cnt [N]; then, with rowptr = exclusive
 * prefix sum of cnt, edge_index [2][E] int64 = [[neighbour j ...], [centre i ...]] grouped by centre.          */
int haet_pc_radius_neighbours(const float* pos, const int32_t* cloud_lo, const int32_t* cloud_hi, int N, int d, float r,
                              int max_nbrs, int32_t* nbr, int32_t* cnt, haet_stream_t stream);
/* the same neighbour lists through a uniform grid (cells of edge `cell` >= r, G[k] cells along axis k from mn[k]):
 * haet_pc_radius_cell_keys writes key[i] = (cloud, cell) of point i; the caller sorts the points by key (stable) into
 * sorted_idx [N] and builds cell_ptr [n_clouds * G0 G1 G2 + 1] (first sorted position of every key); max_nbrs <= 32. */
int haet_pc_radius_cell_keys(const float* pos, const int32_t* cloud_of, int N, int d, const float* mn /*[3], host*/, float cell,
                             const int32_t* G /*[3], host*/, int64_t* key, haet_stream_t stream);
int haet_pc_radius_neighbours_grid(const float* pos, const int32_t* cloud_of, const int64_t* sorted_idx, const int32_t* cell_ptr,
                                   int N, int d, const float* mn, float cell, const int32_t* G, float r, int max_nbrs,
                                   int32_t* nbr, int32_t* cnt, haet_stream_t stream);
int haet_pc_radius_fill(const int32_t* nbr, const int32_t* cnt, const int64_t* rowptr, int N, int max_nbrs, long long E,
                        int64_t* edge_index, haet_stream_t stream);
/* MPNN step (components/mpnn.py:48-68): AB [N][ldab] = [h Wa^T | h Wb^T] (Wa, Wb = the h_j and h_i column blocks of the
 * message Linear), Wc [c][d] its position columns; out[i] = mean over the edges (src[e] -> i), e in
 * [rowptr[i], rowptr[i+1]), of LayerNorm(GELU(A[src] + B[i] + Wc (pos[src] - pos[i]) + bias)); zero without edges. */
int haet_pc_mpnn_message(const float* AB, int ldab, const float* pos, int d, const int32_t* rowptr, const int32_t* src,
                         const float* Wc, const float* bias, long long rows, int c, float eps, float* out, int ldo,
                         haet_stream_t stream);
/* h += LayerNorm_noaffine(u) (mpnn.py:64-66); the new h is also written to h2 when given                       */
int haet_pc_ln_residual(const float* u, int ldu, long long rows, int c, float eps, float* h, int ldh, float* h2, int ldh2,
                        haet_stream_t stream);
/* streaming stages of the Erwin body over the tree-ordered state X [slots][ld] (layers.py, attention.py, mlp.py,
 * pooling.py); perm = rotated order of a block (NULL: tree order); widths are multiples of 4, <= 1024          */
int haet_pc_gather_rows(const float* x, int ldx, const int32_t* idx, long long rows, int c, float* out, int ldo,
                        haet_stream_t stream);
int haet_pc_scatter_masked(const float* X, int ldx, const int32_t* tree_idx, const uint8_t* mask, long long rows, int c,
                           float* out, int ldo, haet_stream_t stream);
int haet_pc_rows_add(float* X, int ldx, const int32_t* perm, const float* R, int ldr, long long rows, int c,
                     haet_stream_t stream);
int haet_pc_ball_rel(const float* P, const int32_t* perm, long long rows, int ball, int d, float* rel, haet_stream_t stream);
int haet_pc_rms_pe(const float* X, int ldx, const int32_t* perm, const float* rel, long long rows, int c, int d,
                   const float* w, const float* pe_w /*[c][d] or NULL*/, const float* pe_b, float* Y, int ldy,
                   haet_stream_t stream);
int haet_pc_ball_attention_supported(int c, int heads, int ball);
int haet_pc_ball_attention(const float* qkv, int ldq, long long rows, int c, int heads, int ball, float* out, int ldo,
                           haet_stream_t stream);
int haet_pc_swiglu(const float* U, int ldu, long long rows, int hid, float* H, int ldh, haet_stream_t stream);
int haet_pc_pool_gather(const float* X, int ldx, const float* P, long long rows, int c, int d, int st, float* Y, int ldy,
                        float* Pc, haet_stream_t stream);
int haet_pc_unpool_input(const float* Xp, int ldx, const float* Pp, const float* Pchild, long long rows, int c, int d, int st,
                         float* Y, int ldy, haet_stream_t stream);
int haet_pc_unpool_merge(const float* skip, int lds_, const float* Yp, int ldy, long long rows, int c, int st, float* out,
                         int ldo, haet_stream_t stream);
/* backward of the block stages (training of the stand-alone model: the blocks' forward and backward run on these
 * kernels and the GEMMs; pooling / BatchNorm / MPNN differentiate through torch device ops):
 *   haet_pc_rms_bwd:             dX[slot(p)] (+)= d RMSNorm(X[slot(p)]) * w  w.r.t. X, from dY[p]
 *   haet_pc_swiglu_bwd:          dU [rows][2 hid] from dH and U = [U1 | U2]
 *   haet_pc_ball_attention_bwd:  dqkv [rows][3c] from qkv and dO (statistics recomputed per ball)               */
int haet_pc_rms_bwd(const float* X, int ldx, const int32_t* perm, const float* dY, int ldy, const float* w, long long rows,
                    int c, float* dX, int lddx, int accumulate, haet_stream_t stream);
int haet_pc_swiglu_bwd(const float* U, int ldu, const float* dH, int ldh, long long rows, int hid, float* dU, int lddu,
                       haet_stream_t stream);
int haet_pc_ball_attention_bwd_supported(int c, int heads, int ball);
int haet_pc_ball_attention_bwd(const float* qkv, int ldq, const float* dO, int ldo, long long rows, int c, int heads, int ball,
                               float* dqkv, int ldd, haet_stream_t stream);
/* backward of an MPNN step: LayerNorm without affine (haet_pc_ln_bwd), and the per-edge message stage
 * (haet_pc_mpnn_message_bwd: dAB [rows][2c] = [dA | dB] -- the A half zero on entry, senders are added with float atomics --
 * and wc_part [haet_pc_mpnn_bwd_waves(rows)][c][3], the persistent waves' shares of d Wc, to be summed by the caller) */
int haet_pc_ln_bwd(const float* V, int ldv, const float* dU, int ldu, long long rows, int c, float eps, float* dV, int lddv,
                   haet_stream_t stream);
int haet_pc_mpnn_bwd_waves(long long rows);
int haet_pc_mpnn_message_bwd(const float* AB, int ldab, const float* pos, int d, const int32_t* rowptr, const int32_t* src,
                             const float* Wc, const float* bias, const float* dagg, int ldg, long long rows, int c, float eps,
                             float* dAB, int ldd, float* wc_part, haet_stream_t stream);
/* nn.BatchNorm1d over rows (pooling.py:23,38,56,68): batch statistics (stats [2][c] = mean, biased variance; part
 * [haet_pc_bn_blocks(rows)][2][c] scratch; running statistics / num_batches_tracked updated when given), then apply */
int haet_pc_bn_blocks(long long rows);
int haet_pc_bn_stats(const float* X, int ldx, long long rows, int c, float momentum, float* part, float* stats,
                     float* run_mean, float* run_var, long long* nbt, haet_stream_t stream);
int haet_pc_bn_apply(float* X, int ldx, long long rows, int c, const float* mean, const float* var, float eps,
                     const float* gamma, const float* beta, haet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HAET_HIP_H */
