// Log-sum-exp merge of slice partial records and the token-position kernel.
#include "common.h"

namespace haet {

// One 256-thread workgroup per (b,h,g).  Phase 1: reference max and the weights
// a_r = e^{m_r - max} of the R records (kept in LDS), l = sum_r l_r a_r.  Phase 2: the C
// channels x (256/C) record groups accumulate sum_r a_r St_r[c] in a fixed order
// (deterministic), reduced through LDS.  R = a few hundred workgroup partials, or <= 8 ranks.
constexpr int kMergeMaxR = 2048;

__global__ __launch_bounds__(256) void slice_merge_kernel(
    const float* __restrict__ rec, int R, int BH, int C, int M, int finalize,
    float* __restrict__ out_rec, float* __restrict__ col_max, float* __restrict__ col_sum,
    float* __restrict__ S) {
    __shared__ float aw[kMergeMaxR];
    __shared__ float red[256];
    __shared__ float wred[8];
    const int bhg = blockIdx.x, bh = bhg / M, g = bhg % M, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const size_t rstride = (size_t)BH * M * (C + 2);
    const size_t base = (size_t)bh * M * (C + 2);
    // All three kinds of loads (m, l, St) are issued before anything is computed on them: the kernel is a chain of
    // dependent round trips to L2 / HBM otherwise (~17 us for 256 records; it runs twice per forward on the
    // point-sharded path).  St: this thread's records r = grp, grp + ngrp, ... in blocks of kPre.
    constexpr int kPre = 32;
    const int ngrp = 256 / C, c = tid % C, grp = tid / C;
    const float* sp = rec + base + 2 * M + (size_t)g * C + c;
    float pre[kPre];
    if (grp < ngrp) {
#pragma unroll
        for (int i = 0; i < kPre; ++i) {
            const int r = grp + i * ngrp;
            pre[i] = r < R ? sp[r * rstride] : 0.f;
        }
    }
    float mx = -INFINITY;
    for (int r = tid; r < R; r += 256) {
        const float mr = rec[r * rstride + base + g];
        red[tid] = rec[r * rstride + base + M + g];  // l_r, parked until the weights exist (R <= 256: one per thread)
        aw[r] = mr;
        mx = fmaxf(mx, mr);
        if (R > 256) break;  // (larger R: the second loop below re-reads l)
    }
    if (R > 256)
        for (int r = tid + 256; r < R; r += 256) {
            const float mr = rec[r * rstride + base + g];
            aw[r] = mr;
            mx = fmaxf(mx, mr);
        }
    mx = wave_max(mx);
    if (lane == 0) wred[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
    float lsum = 0.f;
    for (int r = tid; r < R; r += 256) {
        const float mr = aw[r];
        const float a = (mr == -INFINITY) ? 0.f : __expf(mr - mx);
        aw[r] = a;
        const float lr = (R <= 256) ? red[tid] : rec[r * rstride + base + M + g];
        lsum = fmaf(lr, a, lsum);
    }
    lsum = wave_sum(lsum);
    if (lane == 0) wred[4 + wave] = lsum;
    __syncthreads();
    lsum = (wred[4] + wred[5]) + (wred[6] + wred[7]);
    // phase 2
    float s = 0.f;
    if (grp < ngrp) {
#pragma unroll
        for (int i = 0; i < kPre; ++i) {
            const int r = grp + i * ngrp;
            if (r < R) s = fmaf(pre[i], aw[r], s);
        }
        for (int r = grp + kPre * ngrp; r < R; r += ngrp) s = fmaf(sp[r * rstride], aw[r], s);
    }
    __syncthreads();  // red[] held the l values until here
    red[tid] = s;
    __syncthreads();
    if (tid < C) {
        float tot = 0.f;
        for (int k = 0; k < ngrp; ++k) tot += red[k * C + tid];
        if (out_rec) out_rec[base + 2 * M + (size_t)g * C + tid] = tot;
        // IrregularMesh.py:186-188 with slice_norm == sum_n softmax == 1
        if (finalize) S[(size_t)bhg * C + tid] = (tot / lsum) / (1.0f + 1e-5f);
    }
    if (tid == 0) {
        if (out_rec) {
            out_rec[base + g] = mx;
            out_rec[base + M + g] = lsum;
        }
        if (finalize) {
            col_max[bhg] = mx;
            col_sum[bhg] = lsum;
        }
    }
}

// models/PhysicsAttention/IrregularMesh.py:204-209.  Single workgroup: T <= a few thousand.
__global__ __launch_bounds__(256) void token_pos_kernel(const float* __restrict__ S, int T, int C,
                                                        int d, float* __restrict__ pos) {
    __shared__ float smin[4][4], smax[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = 0; j < d; ++j) {
        float lo = INFINITY, hi = -INFINITY;
        for (int t = threadIdx.x; t < T; t += 256) {
            const float v = S[(size_t)t * C + j];
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
        lo = wave_min(lo);
        hi = wave_max(hi);
        if (lane == 0) {
            smin[wave][j] = lo;
            smax[wave][j] = hi;
        }
    }
    __syncthreads();
    for (int j = 0; j < d; ++j) {
        const float lo = fminf(fminf(smin[0][j], smin[1][j]), fminf(smin[2][j], smin[3][j]));
        const float hi = fmaxf(fmaxf(smax[0][j], smax[1][j]), fmaxf(smax[2][j], smax[3][j]));
        const float rng = hi - lo + 1e-8f;
        for (int t = threadIdx.x; t < T; t += 256)
            pos[(size_t)t * d + j] = (S[(size_t)t * C + j] - lo) / rng;
    }
}

int token_pos_launch(const float* S, int T, int C, int d, float* pos, hipStream_t stream) {
    hipLaunchKernelGGL(token_pos_kernel, dim3(1), dim3(256), 0, stream, S, T, C, d, pos);
    return check_launch("token_pos_kernel");
}

}  // namespace haet

using namespace haet;

extern "C" int haet_slice_merge(const float* records, int R, int B, int H, int C, int M,
                                int finalize, float* out_record, float* col_max, float* col_sum,
                                float* S, haet_stream_t stream) {
    if (!records || R < 1 || B < 1 || H < 1 || C < 1 || M < 1)
        return fail(HAET_E_ARG, "slice_merge: bad arguments");
    if (finalize && (!col_max || !col_sum || !S))
        return fail(HAET_E_ARG, "slice_merge: finalize needs col_max, col_sum and S");
    if (!finalize && !out_record) return fail(HAET_E_ARG, "slice_merge: out_record is NULL");
    if (R > kMergeMaxR) return fail(HAET_E_ARG, "slice_merge: R=%d records > %d", R, kMergeMaxR);
    if (C > 256 || 256 % C) return fail(HAET_E_ARG, "slice_merge: C=%d must divide 256", C);
    hipLaunchKernelGGL(slice_merge_kernel, dim3(B * H * M), dim3(256), 0, as_stream(stream), records,
                       R, B * H, C, M, finalize, out_record, col_max, col_sum, S);
    return check_launch("slice_merge_kernel");
}

extern "C" int haet_token_pos(const float* S, int T, int C, int d, float* pos,
                              haet_stream_t stream) {
    if (!S || !pos || T < 1 || d < 1 || d > 4 || d > C)
        return fail(HAET_E_ARG, "token_pos: bad arguments (T=%d C=%d d=%d)", T, C, d);
    return token_pos_launch(S, T, C, d, pos, as_stream(stream));
}
