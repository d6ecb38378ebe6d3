// Shared helpers for the libhaet_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>

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

}  // namespace haet
