// Weight gradient of a Linear over the points: dW[O][K] = sum_n G[n][o] X[n][k], db[o] = sum_n G[n][o]
// (to_out and in_project_x of the Physics_Attention modules; the reference gets them from torch autograd,
// exp_airfoil.py:249-255).  N is ~1e6, O = K = 256: an output-stationary GEMM whose contraction runs
// over the ROW index of both operands, so the f32 MFMA (v_mfma_f32_32x32x2_f32, exact f32 products)
// takes both operands straight from global memory, the output column in the lane: no LDS, no transposes.
// A workgroup (8 waves, 2 x 4) keeps a 256 x 256 block of dW in its accumulators over its chunk of rows
// and writes one partial record; the caller sums the records.
#include "common.h"

namespace haet {

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kRowsTile = 32;

int wgrad_chunk(int N) {  // rows per workgroup: ~256 workgroups, whole 32-row tiles
    int c = (N + 255) / 256;
    c = ((c + kRowsTile - 1) / kRowsTile) * kRowsTile;
    return c < kRowsTile ? kRowsTile : c;
}

}  // namespace

__global__ __launch_bounds__(512) void gemm_tn_f32_kernel(const float* __restrict__ G, const float* __restrict__ X,
                                                          int N, int O, int K, int chunk,
                                                          float* __restrict__ parts) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const int kblocks = K / 256;
    const int o0 = (blockIdx.y / kblocks) * 256 + 128 * (wave >> 2);  // this wave's 128 x 64 block of dW
    const int k0 = (blockIdx.y % kblocks) * 256 + 64 * (wave & 3);
    const int n0 = blockIdx.x * chunk, n1 = min(N, n0 + chunk);

    f16v acc[4][2];
    float dbl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[ob][kb][v] = 0.f;

    const float* gp = G + o0 + l31;
    const float* xp = X + k0 + l31;
    // A 32-row tile is two groups of 8 k-pairs; k-pair kk of the instruction = rows nb + kk (lanes 0-31) and
    // nb + kk + 16 (lanes 32-63) of both operands.  The loads of one group are in flight while the 64 MFMAs of
    // the other run (4096 cycles per SIMD: more than the memory latency); they are unconditional (rows past the
    // chunk read row n0 and select 0) so that the waits stay counted.
    auto load_group = [&](int nb, int grp, float (&a)[4][8], float (&b)[2][8]) {
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
            const int n = nb + 8 * grp + k8 + 16 * half;
            const bool ok = n < n1;
            const size_t r = ok ? (size_t)n : (size_t)n0;
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                const float v = gp[r * O + 32 * ob];
                a[ob][k8] = ok ? v : 0.f;
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const float v = xp[r * K + 32 * kb];
                b[kb][k8] = ok ? v : 0.f;
            }
        }
    };
    auto mma_group = [&](const float (&a)[4][8], const float (&b)[2][8]) {
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                dbl[ob] += a[ob][k8];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
                    acc[ob][kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ob][k8], b[kb][k8], acc[ob][kb], 0, 0, 0);
            }
    };
    float a0[4][8], b0[2][8], a1[4][8], b1[2][8];
    load_group(n0, 0, a0, b0);
    for (int nb = n0; nb < n1; nb += kRowsTile) {
        load_group(nb, 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mma_group(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        load_group(nb + kRowsTile, 0, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma_group(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }

    float* rec = parts + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (size_t)(256 * 256 + 256);
    const int ol = 128 * (wave >> 2), kl = 64 * (wave & 3);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int o = ol + 32 * ob + 8 * (v >> 2) + 4 * half + (v & 3);
                rec[o * 256 + kl + 32 * kb + l31] = acc[ob][kb][v];
            }
        const float s = dbl[ob] + __shfl_xor(dbl[ob], 32);
        if ((wave & 3) == 0 && half == 0) rec[256 * 256 + ol + 32 * ob + l31] = s;
    }
}

}  // namespace haet

using namespace haet;

extern "C" int haet_linear_wgrad_supported(int O, int K) { return (O > 0 && K > 0 && O % 256 == 0 && K % 256 == 0) ? 1 : 0; }

extern "C" int haet_linear_wgrad_num_parts(int N) {
    if (N < 1) return HAET_E_ARG;
    const int chunk = wgrad_chunk(N);
    return (N + chunk - 1) / chunk;
}

extern "C" int haet_linear_wgrad(const float* G, const float* X, int N, int O, int K, float* parts,
                                 haet_stream_t stream) {
    if (!G || !X || !parts) return fail(HAET_E_ARG, "linear_wgrad: null pointer");
    if (N < 1 || !haet_linear_wgrad_supported(O, K))
        return fail(HAET_E_ARG, "linear_wgrad: N=%d O=%d K=%d unsupported (O, K multiples of 256)", N, O, K);
    const int chunk = wgrad_chunk(N);
    dim3 grid((N + chunk - 1) / chunk, (O / 256) * (K / 256));
    hipLaunchKernelGGL(gemm_tn_f32_kernel, grid, dim3(512), 0, as_stream(stream), G, X, N, O, K, chunk, parts);
    return check_launch("gemm_tn_f32_kernel");
}
