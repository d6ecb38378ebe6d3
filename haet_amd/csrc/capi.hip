// C-ABI glue: error string, version, and the shape dispatch of the slice / de-slice passes.
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace haet {

char* err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

static std::atomic<int> g_deterministic{-1};
bool deterministic_mode() {
    int v = g_deterministic.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("HAET_DETERMINISTIC");
        v = (e && e[0] == '1') ? 1 : 0;
        g_deterministic.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}

// slice_valu.hip
int slice_shape_ok(int B, int N, int H, int C, int M);
int slice_valu_num_parts(int N);
int slice_valu_fwd(const float*, const float*, const float*, const float*, float, float, float, int,
                   int, int, int, int, float*, hipStream_t);
int deslice_valu_fwd(const float*, const float*, const float*, const float*, float, float, float,
                     const float*, const float*, const float*, int, int, int, int, int, float*,
                     hipStream_t);

// slice_mfma.hip
struct MfmaPlan {
    int li, ipw, mb, hpw, tiles_per_chunk, parts, cb;
    int s_nw, s_li;
    size_t lds_slice, lds_deslice;
};
bool mfma_plan(int B, int N, int H, int C, int M, MfmaPlan* p);
int slice_mfma_fwd(const float*, const float*, const float*, const float*, float, float, float, int,
                   int, int, int, int, float*, hipStream_t, const float* s_in);
int deslice_mfma_fwd(const float*, const float*, const float*, const float*, float, float, float,
                     const float*, const float*, const float*, int, int, int, int, int, float*,
                     hipStream_t, const float* s_in);

// slice_f32_mfma.hip: the two passes on the f32 matrix cores (C in {32, 64}, M in {32, 64, 128})
bool slice_f32_ok(int C, int M);
int slice_f32_num_parts(int N, int BH);
int slice_f32_fwd(const float*, const float*, const float*, const float*, float, float, float, int, int, int, int,
                  int, float*, hipStream_t);
int deslice_f32_fwd(const float*, const float*, const float*, const float*, float, float, float, const float*,
                    const float*, const float*, int, int, int, int, int, float*, hipStream_t);

// HAET_FORCE_VALU=1 routes every shape through the generic fp32 VALU kernels, HAET_FORCE_F32_MFMA=1 every
// shape the f32-MFMA kernels cover through those (tests)
static bool env_on(const char* name) {
    const char* e = getenv(name);
    return e && e[0] == '1';
}
static bool use_mfma(int B, int N, int H, int C, int M, MfmaPlan* p) {
    if (env_on("HAET_FORCE_VALU") || (env_on("HAET_FORCE_F32_MFMA") && slice_f32_ok(C, M))) return false;
    return mfma_plan(B, N, H, C, M, p);
}
// the split-bf16 kernels where they apply (C = 32), else the f32-MFMA kernels (C = 64), else vector ALUs
static bool use_f32_mfma(int C, int M) { return !env_on("HAET_FORCE_VALU") && slice_f32_ok(C, M); }

}  // namespace haet

using namespace haet;

extern "C" int haet_abi_version(void) { return HAET_ABI_VERSION; }
extern "C" const char* haet_last_error_string(void) { return err_buf(); }

extern "C" void haet_set_deterministic(int on) { g_deterministic.store(on ? 1 : 0, std::memory_order_relaxed); }
extern "C" int haet_get_deterministic(void) { return deterministic_mode() ? 1 : 0; }

extern "C" int haet_resident_workgroup_limit(void) { return resident_workgroup_limit(); }

extern "C" int haet_erwin_barrier_status(const void* workspace_host_copy8) {
    // (host-side decode of the first 8 bytes of a train workspace: {arrival counter, time-out flag})
    if (!workspace_host_copy8) return fail(HAET_E_ARG, "erwin_barrier_status: null pointer");
    unsigned w[2];
    memcpy(w, workspace_host_copy8, 8);
    if (w[1] != 0)
        return fail(HAET_E_LAUNCH, "erwin: a grid barrier of the train-mode Erwin stage timed out (a workgroup of the "
                                   "launch was not resident: another kernel held compute units, or two processes share "
                                   "the device) -- the BatchNorm batch statistics, running statistics and gradients of "
                                   "that step are invalid");
    return HAET_OK;
}

extern "C" int haet_slice_num_parts(int B, int N, int H, int C, int M) {
    if (slice_shape_ok(B, N, H, C, M) != HAET_OK) return HAET_E_ARG;
    MfmaPlan p;
    if (use_mfma(B, N, H, C, M, &p)) return p.parts;
    if (use_f32_mfma(C, M)) return slice_f32_num_parts(N, B * H);
    return slice_valu_num_parts(N);
}

extern "C" size_t haet_slice_part_floats(int B, int H, int C, int M) {
    return (size_t)B * H * M * (C + 2);
}

extern "C" int haet_slice_fwd(const float* x_lin, const float* Ws, const float* bs, const float* wt,
                              float bt, float base_temp, float c0, int B, int N, int H, int C, int M,
                              float* parts, haet_stream_t stream) {
    if (!x_lin || !Ws || !bs || !wt || !parts) return fail(HAET_E_ARG, "slice_fwd: null pointer");
    int rc = slice_shape_ok(B, N, H, C, M);
    if (rc != HAET_OK) return rc;
    if (use_mfma(B, N, H, C, M, nullptr))
        return slice_mfma_fwd(x_lin, Ws, bs, wt, bt, base_temp, c0, B, N, H, C, M, parts, as_stream(stream), nullptr);
    if (use_f32_mfma(C, M))
        return slice_f32_fwd(x_lin, Ws, bs, wt, bt, base_temp, c0, B, N, H, C, M, parts, as_stream(stream));
    return slice_valu_fwd(x_lin, Ws, bs, wt, bt, base_temp, c0, B, N, H, C, M, parts, as_stream(stream));
}

extern "C" int haet_deslice_fwd(const float* x_lin, const float* Ws, const float* bs, const float* wt,
                                float bt, float base_temp, float c0, const float* col_max,
                                const float* col_sum, const float* S2, int B, int N, int H, int C,
                                int M, float* out_x, haet_stream_t stream) {
    if (!x_lin || !Ws || !bs || !wt || !col_max || !col_sum || !S2 || !out_x)
        return fail(HAET_E_ARG, "deslice_fwd: null pointer");
    int rc = slice_shape_ok(B, N, H, C, M);
    if (rc != HAET_OK) return rc;
    if (use_mfma(B, N, H, C, M, nullptr))
        return deslice_mfma_fwd(x_lin, Ws, bs, wt, bt, base_temp, c0, col_max, col_sum, S2, B, N, H, C,
                                M, out_x, as_stream(stream), nullptr);
    if (use_f32_mfma(C, M))
        return deslice_f32_fwd(x_lin, Ws, bs, wt, bt, base_temp, c0, col_max, col_sum, S2, B, N, H, C, M, out_x,
                               as_stream(stream));
    return deslice_valu_fwd(x_lin, Ws, bs, wt, bt, base_temp, c0, col_max, col_sum, S2, B, N, H, C, M,
                            out_x, as_stream(stream));
}

extern "C" int haet_slice_mfma_supported(int B, int N, int H, int C, int M) {
    MfmaPlan p;
    return (slice_shape_ok(B, N, H, C, M) == HAET_OK && use_mfma(B, N, H, C, M, &p)) ? 1 : 0;
}

// ---- the same two passes on the split-format projection (haet_project_split_fwd) ----------------
extern "C" int haet_slice_split_supported(int B, int N, int H, int C, int M) {
    MfmaPlan p;
    return (slice_shape_ok(B, N, H, C, M) == HAET_OK && C == 32 && (H == 8 || H == 16) && use_mfma(B, N, H, C, M, &p)) ? 1 : 0;
}

extern "C" int haet_slice_fwd_split(const void* x_split, const float* s, const float* Ws, const float* bs,
                                    float c0, int B, int N, int H, int C, int M, float* parts,
                                    haet_stream_t stream) {
    if (!x_split || !s || !Ws || !bs || !parts) return fail(HAET_E_ARG, "slice_fwd_split: null pointer");
    if (!haet_slice_split_supported(B, N, H, C, M))
        return fail(HAET_E_ARG, "slice_fwd_split: shape B=%d N=%d H=%d C=%d M=%d is outside the MFMA path", B, N, H, C, M);
    return slice_mfma_fwd(reinterpret_cast<const float*>(x_split), Ws, bs, nullptr, 0.f, 0.f, c0, B, N, H, C, M,
                          parts, as_stream(stream), s);
}

extern "C" int haet_deslice_fwd_split(const void* x_split, const float* s, const float* Ws, const float* bs,
                                      float c0, const float* col_max, const float* col_sum,
                                      const float* S2, int B, int N, int H, int C, int M, float* out_x,
                                      haet_stream_t stream) {
    if (!x_split || !s || !Ws || !bs || !col_max || !col_sum || !S2 || !out_x)
        return fail(HAET_E_ARG, "deslice_fwd_split: null pointer");
    if (!haet_slice_split_supported(B, N, H, C, M))
        return fail(HAET_E_ARG, "deslice_fwd_split: shape B=%d N=%d H=%d C=%d M=%d is outside the MFMA path", B, N, H, C, M);
    return deslice_mfma_fwd(reinterpret_cast<const float*>(x_split), Ws, bs, nullptr, 0.f, 0.f, c0, col_max,
                            col_sum, S2, B, N, H, C, M, out_x, as_stream(stream), s);
}
