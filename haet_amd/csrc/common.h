// Shared helpers for the libhaet_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "../../include/haet_hip.h"

namespace haet {

constexpr int kWave = 64;

// thread-local "last error" text behind haet_last_error_string()
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(HAET_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return HAET_OK;
}

// haet_set_deterministic / HAET_DETERMINISTIC=1: entry points with a faster variant whose floating-point summation
// order varies from run to run (float atomics) take the fixed-order variant instead (capi.hip)
bool deterministic_mode();

inline hipStream_t as_stream(haet_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Points per workgroup of the slice / de-slice BACKWARD passes (slice_bwd.hip, slice_bwd_mfma.hip, slice_bwd_x3.hip):
// ~256 chunks at large N, so that the per-workgroup partial records stay small next to the stream.  ONE definition:
// haet_slice_bwd_num_parts sizes the caller's `parts` buffer from it and all three kernel families index that buffer
// with it.
// `family`: the kernel family that will run (0 = vector ALUs, 1 = f32 matrix cores, 2 = split-bf16 matrix cores:
// slice_bwd.hip bwd_family) -- the split-bf16 kernels pay a large fixed cost per workgroup (the constant fragments of
// its head, its partial record: pass 2 at 45k points x 8 heads took 228 us with 256-point chunks, ONE tile per wave,
// 84 us with 2048-point chunks; tools/head_pad_probe.py), the vector-ALU kernels want many short chunks (271 -> 506 us).
enum { kBwdValu = 0, kBwdF32 = 1, kBwdX3 = 2 };
inline int slice_bwd_pick_chunk(int N, int family) {
    static const int forced = [] {
        const char* e = getenv("HAET_BWD_CHUNK");  // tuning: a fixed chunk (multiple of 256) for every family
        const int v = e ? atoi(e) : 0;
        return v >= 256 ? (v / 256) * 256 : 0;
    }();
    if (forced) return forced;
    const int per = family == kBwdX3 ? 128 : 256;  // chunks per (cloud, head) at large N (1M points, split-bf16 pass 2:
    int c = (N + per - 1) / per;                   //  4096-point chunks 1.15 ms, 8192 1.01, 16384 0.95 with pass 1 slower)
    c = ((c + 255) / 256) * 256;
    const int least = family == kBwdX3 ? 2048 : 256;
    return c < least ? least : c;
}

// Zero `bytes` (a multiple of 4) at `p` (4-byte aligned) with a KERNEL: hipMemsetAsync captured into a HIP graph
// (torch.compile mode="reduce-overhead") did not reliably re-run ahead of its consumer on replay -- the split-k convolution,
// which adds into a zeroed output, returned garbage from the second replay on (tools/graph_replay_check.py).
template <int kDummy = 0>
__global__ void zero_words_kernel(unsigned* __restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0u;
}
inline hipError_t zero_async(void* p, size_t bytes, hipStream_t st) {
    const size_t n = bytes / 4;
    if (n == 0) return hipSuccess;
    const size_t want = (n + 255) / 256;
    const unsigned blocks = (unsigned)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(zero_words_kernel<0>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<unsigned*>(p), n);
    return hipGetLastError();
}

// "1"-valued environment switch, read per call (the tests flip them between calls)
inline bool env_flag(const char* name) {
    const char* e = getenv(name);
    return e && e[0] == '1';
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is PER DEVICE: a process-wide "already set" flag skips the
// call on a second device and the > 64 KiB launch fails there.  One bit per device ordinal, set atomically
// (several host threads may launch at once).
struct PerDeviceOnce {
    std::atomic<unsigned long long> mask{0};
    static int bit() {
        int d = 0;
        (void)hipGetDevice(&d);
        return d & 63;
    }
    bool done() const { return (mask.load(std::memory_order_acquire) >> bit()) & 1ull; }
    void set() { mask.fetch_or(1ull << bit(), std::memory_order_release); }
};

// Workgroups that can be co-resident when a launch places ONE per compute unit (the grid barriers of the train-mode
// Erwin kernels need every workgroup of the launch resident): the device's CU count, queried once per device
// (CPX partitions, CU masks and other gfx9 parts report fewer than MI355X's 256).  256 when there is no device to ask
// (the host-only ABI checks).
inline int resident_workgroup_limit() {
    static std::atomic<int> cached[64];
    const int b = PerDeviceOnce::bit();
    int v = cached[b].load(std::memory_order_acquire);
    if (v > 0) return v;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
        (void)hipGetLastError();
        return 256;
    }
    cached[b].store(cus, std::memory_order_release);
    return cus;
}

// Entry points that memset / memcpy / synchronise cannot run under stream capture on every path (a synchronising call
// inside a capture invalidates it and, on ROCm 7.2, has taken the runtime down): they test this first and return
// HAET_E_ARG instead of proceeding.
inline bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Ada-Temp (models/PhysicsAttention/IrregularMesh.py:168): tau = base + clamp(t, -0.4, 0.4)
__device__ __forceinline__ float ada_temp(float t, float base_temp) {
    return base_temp + fminf(fmaxf(t, -0.4f), 0.4f);
}

#if defined(__HIPCC__)
// Two fp32 -> their split-bf16 pair: hi = {RNE(a), RNE(b)} packed, lo = {RNE(a - hi_a), RNE(b - hi_b)} packed
// (five vector instructions: v_cvt_pk, shift, mask, v_pk_add, v_cvt_pk).
__device__ __forceinline__ void split_bf16_pair(float a, float b, unsigned& hi, unsigned& lo) {
    typedef __attribute__((ext_vector_type(2))) __bf16 sp_bf16x2;
    typedef __attribute__((ext_vector_type(2))) float sp_f32x2;
    const sp_f32x2 v = {a, b};
    const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(v, sp_bf16x2));
    const sp_f32x2 l = {a - __uint_as_float(hb << 16), b - __uint_as_float(hb & 0xffff0000u)};
    hi = hb;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(l, sp_bf16x2));
}
// (Measured and rejected, round 3: the residuals from v_dot2c_f32_bf16 -- a + hi . {-1, 0}, b + hi . {0, -1}, exact, four
// instructions per pair -- are no faster anywhere and slower in the GEMM loader and the slice pass; packed v_pk_fma_f32
// exponent arguments likewise.  The dot-product selector also must not be left to hipcc: 0x0000bf80 becomes the INLINE
// constant -1.0, which the hardware expands to another bit pattern for a packed-bf16 operand.  DESIGN.md 5b.)
#endif

}  // namespace haet
