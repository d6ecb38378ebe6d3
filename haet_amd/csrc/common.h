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
