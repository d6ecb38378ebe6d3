// Fused Erwin stage: the whole hierarchical ball transformer over the slice tokens of
// one (batch, head) cloud runs in ONE workgroup with the token state resident in LDS.
// The reference runs ~100 small kernels (+ a host round trip for the tree) here
// (models/components/erwinflash/erwin_flash.py:138-195); at T <= ~1k tokens that is pure
// launch latency, and it is the fixed cost that bounds multi-GPU scaling.
//
// Layout in LDS (floats): per level l the tokens X_l [n_l][c_l+4] and positions
// P_l [n_l][d] in TREE order, two scratch buffers, index maps.  Rotated blocks
// (layers.py:81-89) never move data: a position map `perm` says which token sits at
// which place of the ball sequence; everything except the per-ball attention and
// the ball-centre is per-token and order-free.
//
// Every Linear runs on the matrix cores as split-bf16 (v_mfma_f32_16x16x32_bf16 x3: hi*hi +
// hi*lo + lo*hi, fp32 accumulate): a wave owns a 16-token x 16-output tile, reads the
// activations from LDS and its weight fragments straight from global/L2 ([out][in] rows,
// 8 consecutive k per lane), all of a tile's weight loads in flight before the first MFMA.
#include "common.h"

namespace haet {

constexpr int kPad = 4;  // row padding (floats): keeps rows 16-B aligned, breaks bank strides

struct Lds {
    float* X[HAET_ERWIN_MAX_LEVELS];
    float* P[HAET_ERWIN_MAX_LEVELS];
    int* rot[HAET_ERWIN_MAX_LEVELS];
    int n[HAET_ERWIN_MAX_LEVELS];
    int ld[HAET_ERWIN_MAX_LEVELS];
    float* A;    // [n][c+4]  normed input / attention output
    float* Bf;   // [n][wide] qkv / SwiGLU hidden / pool & unpool rows
    float* rel;  // [n][d]
    int* ident;
};

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct Frag {
    u32x4 hi, lo;
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// 8 fp32 -> split-bf16 MFMA operand (hi = RNE(x), lo = RNE(x - hi)): fp32-class products
__device__ __forceinline__ Frag split8(const f32x4 a, const f32x4 b) {
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    Frag f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned h = pack_bf16(v[2 * i], v[2 * i + 1]);
        f.hi[i] = h;
        f.lo[i] = pack_bf16(v[2 * i] - __uint_as_float(h << 16),
                            v[2 * i + 1] - __uint_as_float(h & 0xffff0000u));
    }
    return f;
}

__device__ __forceinline__ f32x4 mfma3(const Frag& a, const Frag& b, f32x4 c) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, a.hi), al = __builtin_bit_cast(bf16x8, a.lo);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, b.hi), bl = __builtin_bit_cast(bf16x8, b.lo);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    return c;
}

constexpr int kKC = 4;  // k-steps (of 32) whose weight loads are put in flight together

// Y[t][o] = sum_k X[t*ldx + k] * W[o*Kp + k]  on the matrix cores (v_mfma_f32_16x16x32_bf16,
// split-bf16 x3).  X: fp32 rows in LDS; W: global, PyTorch [out][in] layout with rows zero-padded
// to Kp = roundup(K, 32).  One wave owns a 16-token x 16-output tile; its weight fragments
// (lane = output row, 8 consecutive k: two 16-B loads per k-step) are all issued before the
// first MFMA, so a tile costs ~one L2 latency instead of one per k.  NMAT = 2 evaluates two
// weight matrices on the same activations (SwiGLU's w1/w2).  epi(t, o, acc...) finishes.
template <int NMAT, typename Epi>
__device__ __forceinline__ void linear_mfma(const float* X, int ldx, int n, int K,
                                            const float* __restrict__ W0,
                                            const float* __restrict__ W1, int O, Epi epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int Kp = (K + 31) & ~31, nks = Kp >> 5;
    const int nto = (O + 15) >> 4, ntiles = ((n + 15) >> 4) * nto;
    for (int tile = wave; tile < ntiles; tile += nwaves) {
        const int tt = tile / nto, ot = tile - tt * nto;
        const int trow = min(tt * 16 + l15, n - 1);
        const int orow = min(ot * 16 + l15, O - 1);
        const float* xr = X + trow * ldx + 8 * kq;
        const float* w0 = W0 + (size_t)orow * Kp + 8 * kq;
        const float* w1 = NMAT == 2 ? W1 + (size_t)orow * Kp + 8 * kq : nullptr;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < nks; k0 += kKC) {
            f32x4 wa[kKC][2], wb[kKC][2];
#pragma unroll
            for (int i = 0; i < kKC; ++i) {
                if (k0 + i < nks) {
                    wa[i][0] = *reinterpret_cast<const f32x4*>(w0 + 32 * (k0 + i));
                    wa[i][1] = *reinterpret_cast<const f32x4*>(w0 + 32 * (k0 + i) + 4);
                    if (NMAT == 2) {
                        wb[i][0] = *reinterpret_cast<const f32x4*>(w1 + 32 * (k0 + i));
                        wb[i][1] = *reinterpret_cast<const f32x4*>(w1 + 32 * (k0 + i) + 4);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < kKC; ++i) {
                if (k0 + i < nks) {
                    const int kb = 32 * (k0 + i) + 8 * kq;  // this lane's first k
                    f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = {0.f, 0.f, 0.f, 0.f};
                    if (kb + 8 <= K) {
                        x0 = *reinterpret_cast<const f32x4*>(xr + 32 * (k0 + i));
                        x1 = *reinterpret_cast<const f32x4*>(xr + 32 * (k0 + i) + 4);
                    } else {  // ragged tail (K = 70, 3, ...): never read past the row
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (kb + j < K) x0[j] = xr[32 * (k0 + i) + j];
                            if (kb + 4 + j < K) x1[j] = xr[32 * (k0 + i) + 4 + j];
                        }
                    }
                    const Frag xa = split8(x0, x1);
                    acc0 = mfma3(xa, split8(wa[i][0], wa[i][1]), acc0);
                    if (NMAT == 2) acc1 = mfma3(xa, split8(wb[i][0], wb[i][1]), acc1);
                }
            }
        }
        const int o = ot * 16 + l15;
        if (o < O) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = tt * 16 + 4 * kq + r;
                if (t < n) epi(t, o, acc0[r], acc1[r]);
            }
        }
    }
}

// A[t][:] = RMSNorm(X[t][:]) * w  (+ extra[t][:] if given); 4 threads per token row.
// nn.RMSNorm(eps=None): eps = finfo(float32).eps  (layers.py:25-26)
__device__ void rms_norm_rows(const float* X, int ldx, int n, int c, const float* __restrict__ w,
                              float* A, int lda) {
    const int sub = threadIdx.x & 3;
    for (int t = threadIdx.x >> 2; t < n; t += (blockDim.x >> 2)) {
        const float* xr = X + t * ldx;
        float ss = 0.f;
        for (int ch = sub; ch < c; ch += 4) ss = fmaf(xr[ch], xr[ch], ss);
        ss += __shfl_xor(ss, 1);
        ss += __shfl_xor(ss, 2);
        const float r = rsqrtf(ss / (float)c + 1.1920928955078125e-07f);
        for (int ch = sub; ch < c; ch += 4) A[t * lda + ch] = xr[ch] * r * w[ch];
    }
}

// rel[tok] = P[tok] - mean over its ball (attention.py:35-40); ball b = places [b*ball, ..)
__device__ void ball_rel_pos(const float* P, int n, int d, int ball, const int* perm, float* rel) {
    const int nb = n / ball;
    for (int i = threadIdx.x; i < nb * d; i += blockDim.x) {
        const int b = i / d, j = i % d;
        float s = 0.f;
        for (int p = 0; p < ball; ++p) s += P[perm[b * ball + p] * d + j];
        const float mean = s / (float)ball;
        for (int p = 0; p < ball; ++p) {
            const int tok = perm[b * ball + p];
            rel[tok * d + j] = P[tok * d + j] - mean;
        }
    }
}

// exact softmax(q k^T / sqrt(hd)) v inside each ball (attention.py:68-84; the reference calls
// flash_attn_qkvpacked_func here).  One thread per (place, head); qkv rows are
// [q(h,hd) | k(h,hd) | v(h,hd)] ("(three h d)").  Output overwrites A[tok][h*hd..].
template <int HD>
__device__ void ball_attention(const float* qkv, int ldq, int n, int c, int heads, int ball,
                               const int* perm, float* A, int lda) {
    const float scale = 1.0f / sqrtf((float)HD);
    for (int i = threadIdx.x; i < n * heads; i += blockDim.x) {
        const int place = i / heads, h = i % heads;
        const int tq = perm[place], b0 = (place / ball) * ball;
        float q[HD], acc[HD];
#pragma unroll
        for (int e = 0; e < HD; ++e) {
            q[e] = qkv[tq * ldq + h * HD + e] * scale;
            acc[e] = 0.f;
        }
        float m = -INFINITY, l = 0.f;
        for (int p = 0; p < ball; ++p) {
            const float* kr = qkv + perm[b0 + p] * ldq + c + h * HD;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < HD; ++e) s = fmaf(q[e], kr[e], s);
            const float mn = fmaxf(m, s);
            const float a = __expf(m - mn), pr = __expf(s - mn);
            l = l * a + pr;
            const float* vr = kr + c;
#pragma unroll
            for (int e = 0; e < HD; ++e) acc[e] = fmaf(pr, vr[e], acc[e] * a);
            m = mn;
        }
        const float inv = 1.0f / l;
#pragma unroll
        for (int e = 0; e < HD; ++e) A[tq * lda + h * HD + e] = acc[e] * inv;
    }
}

__device__ void attention_dispatch(int hd, const float* qkv, int ldq, int n, int c, int heads,
                                   int ball, const int* perm, float* A, int lda) {
    switch (hd) {
        case 1: ball_attention<1>(qkv, ldq, n, c, heads, ball, perm, A, lda); break;
        case 2: ball_attention<2>(qkv, ldq, n, c, heads, ball, perm, A, lda); break;
        case 4: ball_attention<4>(qkv, ldq, n, c, heads, ball, perm, A, lda); break;
        case 8: ball_attention<8>(qkv, ldq, n, c, heads, ball, perm, A, lda); break;
        case 16: ball_attention<16>(qkv, ldq, n, c, heads, ball, perm, A, lda); break;
        case 32: ball_attention<32>(qkv, ldq, n, c, heads, ball, perm, A, lda); break;
        default: ball_attention<64>(qkv, ldq, n, c, heads, ball, perm, A, lda); break;
    }
}

// ErwinTransformerBlock (layers.py:30-32) on the tokens of one level, in place
__device__ void erwin_block(const Lds& s, int lvl, const haet_erwin_layer& L,
                            const haet_erwin_block_w& w, const float* __restrict__ W, int d,
                            const int* perm) {
    float* X = s.X[lvl];
    const float* P = s.P[lvl];
    const int n = s.n[lvl], c = L.c, ld = s.ld[lvl], hid = L.hidden;
    const int ldq = 3 * c + kPad, ldh = hid + kPad;
    // x~ = RMSNorm(x) + pe_proj(pos - ball centre)            (attention.py:59-61)
    ball_rel_pos(P, n, d, L.ball, perm, s.rel);
    rms_norm_rows(X, ld, n, c, W + w.norm1, s.A, ld);
    __syncthreads();
    {
        const float* pw = W + w.pe_w;
        const float* pb = W + w.pe_b;
        for (int i = threadIdx.x; i < n * c; i += blockDim.x) {
            const int t = i / c, ch = i % c;
            float v = pb[ch];
            for (int j = 0; j < d; ++j) v = fmaf(s.rel[t * d + j], pw[j * c + ch], v);
            s.A[t * ld + ch] += v;
        }
    }
    __syncthreads();
    {
        const float* qb = W + w.qkv_b;
        float* Q = s.Bf;
        linear_mfma<1>(s.A, ld, n, c, W + w.qkv_w, nullptr, 3 * c,
                       [=](int t, int o, float v, float) { Q[t * ldq + o] = v + qb[o]; });
    }
    __syncthreads();
    attention_dispatch(c / L.heads, s.Bf, ldq, n, c, L.heads, L.ball, perm, s.A, ld);
    __syncthreads();
    {
        const float* pb = W + w.proj_b;
        linear_mfma<1>(s.A, ld, n, c, W + w.proj_w, nullptr, c,
                       [=](int t, int o, float v, float) { X[t * ld + o] += v + pb[o]; });
    }
    __syncthreads();
    // x += W3( W2 x^ * silu(W1 x^) ),  x^ = RMSNorm(x)          (mlp.py:17-18)
    rms_norm_rows(X, ld, n, c, W + w.norm2, s.A, ld);
    __syncthreads();
    {
        const float* b1 = W + w.w1_b;
        const float* b2 = W + w.w2_b;
        float* Hd = s.Bf;
        linear_mfma<2>(s.A, ld, n, c, W + w.w1_w, W + w.w2_w, hid, [=](int t, int o, float u, float v) {
            const float g = u + b1[o];
            Hd[t * ldh + o] = (v + b2[o]) * (g / (1.0f + __expf(-g)));
        });
    }
    __syncthreads();
    {
        const float* b3 = W + w.w3_b;
        linear_mfma<1>(s.Bf, ldh, n, hid, W + w.w3_w, nullptr, c,
                       [=](int t, int o, float v, float) { X[t * ld + o] += v + b3[o]; });
    }
    __syncthreads();
}

__device__ void run_layer(const Lds& s, int lvl, const haet_erwin_layer& L,
                          const float* __restrict__ W, int d, int rotate) {
    for (int j = 0; j < L.depth; ++j) {
        const bool rot = rotate && L.depth > 1 && (j & 1);  // layers.py:66,75-89
        erwin_block(s, lvl, L, L.blocks[j], W, d, rot ? s.rot[lvl] : s.ident);
    }
}

__global__ __launch_bounds__(512) void erwin_kernel(
    const float* __restrict__ tokens, const float* __restrict__ pos,
    const int32_t* __restrict__ tree_idx, const uint8_t* __restrict__ tree_mask,
    const int32_t* __restrict__ rot, const float* __restrict__ W, const haet_erwin_desc D, int M,
    int Mp, int wideB, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int cloud = blockIdx.x, n_clouds = gridDim.x, d = D.d, nl = D.n_levels;
    Lds s;
    {
        float* p = lds;
        int n = Mp, maxA = 0, maxn = Mp;
        for (int l = 0; l < nl; ++l) {
            s.n[l] = n;
            s.ld[l] = D.enc[l].c + kPad;
            s.X[l] = p;
            p += n * s.ld[l];
            maxA = max(maxA, n * s.ld[l]);
            if (l < nl - 1) n /= D.pool[l].stride;
        }
        for (int l = 0; l < nl; ++l) {
            s.P[l] = p;
            p += ((s.n[l] * d + 3) & ~3);
        }
        s.A = p;
        p += maxA;
        s.Bf = p;
        p += wideB;
        s.rel = p;
        p += ((maxn * d + 3) & ~3);
        int* q = reinterpret_cast<int*>(p);
        for (int l = 0; l < nl; ++l) {
            s.rot[l] = q;
            q += s.n[l];
        }
        s.ident = q;
    }
    // gather the cloud into tree order (erwin_flash.py:176-181)
    const int c0 = D.enc[0].c;
    for (int i = threadIdx.x; i < Mp * c0; i += blockDim.x) {
        const int slot = i / c0, ch = i % c0;
        s.X[0][slot * s.ld[0] + ch] = tokens[(size_t)tree_idx[(size_t)cloud * Mp + slot] * c0 + ch];
    }
    for (int i = threadIdx.x; i < Mp * d; i += blockDim.x) {
        const int slot = i / d, j = i % d;
        s.P[0][i] = pos[(size_t)tree_idx[(size_t)cloud * Mp + slot] * d + j];
    }
    for (int i = threadIdx.x; i < Mp; i += blockDim.x) s.ident[i] = i;
    if (D.rotate) {
        size_t off = 0;
        for (int l = 0; l < nl; ++l) {
            for (int i = threadIdx.x; i < s.n[l]; i += blockDim.x)
                s.rot[l][i] = rot[off + (size_t)cloud * s.n[l] + i] - cloud * s.n[l];
            off += (size_t)n_clouds * s.n[l];
        }
    }
    __syncthreads();

    // encoder: blocks then BallPooling (pooling.py:25-40)
    for (int l = 0; l < nl - 1; ++l) {
        run_layer(s, l, D.enc[l], W, d, D.rotate);
        const haet_erwin_resample& R = D.pool[l];
        const int st = R.stride, n2 = s.n[l + 1], c = D.enc[l].c, K = st * c + st * d;
        const int ldk = ((K + 3) & ~3) + kPad;
        for (int i = threadIdx.x; i < n2 * d; i += blockDim.x) {
            const int p = i / d, j = i % d;
            float sum = 0.f;
            for (int q = 0; q < st; ++q) sum += s.P[l][(p * st + q) * d + j];
            s.P[l + 1][i] = sum / (float)st;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n2 * K; i += blockDim.x) {
            const int p = i / K, k = i % K;
            float v;
            if (k < st * c) {
                v = s.X[l][(p * st + k / c) * s.ld[l] + k % c];
            } else {
                const int r = k - st * c, q = r / d, j = r % d;
                v = s.P[l][(p * st + q) * d + j] - s.P[l + 1][p * d + j];
            }
            s.Bf[p * ldk + k] = v;
        }
        __syncthreads();
        {
            float* Y = s.X[l + 1];
            const int ldy = s.ld[l + 1];
            const float* b = W + R.b;
            const float* sc = W + R.bn_scale;
            const float* sh = W + R.bn_shift;
            linear_mfma<1>(s.Bf, ldk, n2, K, W + R.w, nullptr, D.enc[l + 1].c,
                           [=](int t, int o, float v, float) { Y[t * ldy + o] = (v + b[o]) * sc[o] + sh[o]; });
        }
        __syncthreads();
    }
    run_layer(s, nl - 1, D.enc[nl - 1], W, d, D.rotate);
    // decoder: BallUnpooling (pooling.py:58-72) then blocks
    for (int l = nl - 2; l >= 0; --l) {
        const haet_erwin_resample& R = D.unpool[l];
        const int st = R.stride, n2 = s.n[l + 1], c2 = D.enc[l + 1].c, c = D.enc[l].c;
        const int K = c2 + st * d, ldk = ((K + 3) & ~3) + kPad;
        for (int i = threadIdx.x; i < n2 * K; i += blockDim.x) {
            const int p = i / K, k = i % K;
            float v;
            if (k < c2) {
                v = s.X[l + 1][p * s.ld[l + 1] + k];
            } else {
                const int r = k - c2, q = r / d, j = r % d;
                v = s.P[l][(p * st + q) * d + j] - s.P[l + 1][p * d + j];
            }
            s.Bf[p * ldk + k] = v;
        }
        __syncthreads();
        {
            float* Y = s.X[l];
            const int ldy = s.ld[l];
            const float* b = W + R.b;
            const float* sc = W + R.bn_scale;
            const float* sh = W + R.bn_shift;
            linear_mfma<1>(s.Bf, ldk, n2, K, W + R.w, nullptr, st * c, [=](int t, int o, float v, float) {
                const int q = o / c, ch = o % c;  // 'n (m d) -> (n m) d'
                float* y = Y + (t * st + q) * ldy + ch;
                *y = (*y + v + b[o]) * sc[ch] + sh[ch];
            });
        }
        __syncthreads();
        run_layer(s, l, D.dec[l], W, d, D.rotate);
    }
    // node.x[tree_mask][argsort(tree_idx[tree_mask])]  (erwin_flash.py:193)
    for (int i = threadIdx.x; i < Mp * c0; i += blockDim.x) {
        const int slot = i / c0, ch = i % c0;
        const size_t g = (size_t)cloud * Mp + slot;
        if (tree_mask[g]) out[(size_t)tree_idx[g] * c0 + ch] = s.X[0][slot * s.ld[0] + ch];
    }
}

static int erwin_wide(const haet_erwin_desc* D, int Mp, int* total_floats) {
    int n = Mp, wide = 0, maxA = 0, tot = 0, idx = 0, d = D->d;
    int nn[HAET_ERWIN_MAX_LEVELS];
    for (int l = 0; l < D->n_levels; ++l) {
        nn[l] = n;
        const haet_erwin_layer& L = D->enc[l];
        const int ld = L.c + kPad;
        tot += n * ld + ((n * d + 3) & ~3);
        maxA = maxA > n * ld ? maxA : n * ld;
        idx += n;
        int w = n * (3 * L.c + kPad);
        int h = n * (L.hidden + kPad);
        if (l < D->n_levels - 1) {
            const haet_erwin_layer& Ld = D->dec[l];
            h = h > n * (Ld.hidden + kPad) ? h : n * (Ld.hidden + kPad);
            const int st = D->pool[l].stride;
            const int n2 = n / st;
            const int Kp = st * L.c + st * d, Ku = D->enc[l + 1].c + st * d;
            const int a = n2 * (((Kp + 3) & ~3) + kPad), b = n2 * (((Ku + 3) & ~3) + kPad);
            w = w > a ? w : a;
            w = w > b ? w : b;
            n = n2;
        }
        wide = wide > w ? wide : w;
        wide = wide > h ? wide : h;
    }
    (void)nn;
    tot += maxA + wide + ((Mp * d + 3) & ~3) + idx + Mp;
    *total_floats = tot;
    return wide;
}

static int erwin_validate(const haet_erwin_desc* D, int Mp) {
    if (!D) return fail(HAET_E_ARG, "erwin: desc is NULL");
    if (D->n_levels < 1 || D->n_levels > HAET_ERWIN_MAX_LEVELS)
        return fail(HAET_E_ARG, "erwin: n_levels=%d unsupported", D->n_levels);
    if (!D->decode) return fail(HAET_E_ARG, "erwin: decode=False is not on the PhysicsAttention path");
    if (D->d != 2 && D->d != 3) return fail(HAET_E_ARG, "erwin: d=%d", D->d);
    int n = Mp;
    for (int l = 0; l < D->n_levels; ++l) {
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && l == D->n_levels - 1) break;
            const haet_erwin_layer& L = pass ? D->dec[l] : D->enc[l];
            if (L.depth < 0 || L.depth > HAET_ERWIN_MAX_DEPTH)
                return fail(HAET_E_ARG, "erwin: depth=%d unsupported", L.depth);
            if (L.c < 4 || L.c % 4) return fail(HAET_E_ARG, "erwin: channels=%d (multiple of 4)", L.c);
            if (L.heads < 1 || L.c % L.heads) return fail(HAET_E_ARG, "erwin: heads=%d", L.heads);
            const int hd = L.c / L.heads;
            if (hd != 1 && hd != 2 && hd != 4 && hd != 8 && hd != 16 && hd != 32 && hd != 64)
                return fail(HAET_E_ARG, "erwin: head_dim=%d unsupported", hd);
            if (L.ball < 1 || n % L.ball)
                return fail(HAET_E_ARG, "erwin: ball size %d does not divide %d tokens", L.ball, n);
            if (L.hidden < 1) return fail(HAET_E_ARG, "erwin: hidden=%d", L.hidden);
        }
        if (l < D->n_levels - 1) {
            if (D->dec[l].c != D->enc[l].c) return fail(HAET_E_ARG, "erwin: decoder width mismatch");
            const int st = D->pool[l].stride;
            if (st < 1 || n % st || D->unpool[l].stride != st)
                return fail(HAET_E_ARG, "erwin: stride %d", st);
            n /= st;
        }
    }
    return HAET_OK;
}

}  // namespace haet

using namespace haet;

extern "C" size_t haet_erwin_lds_bytes(const haet_erwin_desc* desc, int Mp) {
    if (erwin_validate(desc, Mp) != HAET_OK) return 0;
    int tot = 0;
    erwin_wide(desc, Mp, &tot);
    return (size_t)tot * sizeof(float);
}

extern "C" int haet_erwin_fwd(const float* tokens, const float* pos, const int32_t* tree_idx,
                              const uint8_t* tree_mask, const int32_t* rot, const float* weights,
                              const haet_erwin_desc* desc, int n_clouds, int M, float* out,
                              haet_stream_t stream) {
    if (!tokens || !pos || !tree_idx || !tree_mask || !weights || !out || n_clouds < 1 || M < 2)
        return fail(HAET_E_ARG, "erwin: null pointer or bad sizes");
    int depth = 0;
    while ((2 << depth) < M) ++depth;
    const int Mp = 2 << depth;
    int rc = erwin_validate(desc, Mp);
    if (rc != HAET_OK) return rc;
    if (desc->rotate && !rot) return fail(HAET_E_ARG, "erwin: rotate set but rot is NULL");
    int tot = 0;
    const int wide = erwin_wide(desc, Mp, &tot);
    const size_t bytes = (size_t)tot * sizeof(float);
    if (bytes > 160 * 1024)
        return fail(HAET_E_RESOURCE, "erwin: one cloud needs %zu B of LDS (> 160 KiB)", bytes);
    static thread_local size_t attr_set = 0;
    if (bytes > 64 * 1024 && bytes > attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(erwin_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return fail(HAET_E_LAUNCH, "erwin: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = bytes;
    }
    hipLaunchKernelGGL(erwin_kernel, dim3(n_clouds), dim3(512), bytes, as_stream(stream), tokens,
                       pos, tree_idx, tree_mask, rot, weights, *desc, M, Mp, wide, out);
    return check_launch("erwin_kernel");
}
