// Micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate, 8 waves per CU (2 per SIMD), one or two
// accumulator chains per wave, operands random or zero.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int CHAINS>
__global__ __launch_bounds__(512) void mfma_loop(const u32x4* __restrict__ src, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x;
    u32x4 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = src[(lane * 8 + i) & 4095];
        b[i] = src[(lane * 8 + i + 2048) & 4095];
    }
    f32x16 acc[2] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bf16x8 av = __builtin_bit_cast(bf16x8, a[i]), bv = __builtin_bit_cast(bf16x8, b[i]);
            acc[CHAINS == 2 ? (i & 1) : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[CHAINS == 2 ? (i & 1) : 0], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int iters = 4000, wgs = 256;
    std::vector<unsigned> h(4096 * 4);
    u32x4* src;
    float* out;
    hipMalloc(&src, h.size() * 4);
    hipMalloc(&out, wgs * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (auto& v : h) v = mode ? 0u : ((unsigned)(rand() & 0x7fff) | 0x3c000000u | ((unsigned)(rand() & 0x7fff) << 16) | 0x00003c00u);
        hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int chains = 1; chains <= 2; ++chains) {
            for (int threads = 256; threads <= 512; threads += 256) {
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (chains == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(wgs), dim3(threads), 0, 0, src, out, iters);
                    else hipLaunchKernelGGL(mfma_loop<2>, dim3(wgs), dim3(threads), 0, 0, src, out, iters);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    const double n = (double)wgs * (threads / 64) * iters * 8;
                    if (rep) printf("%s operands, %d chain(s), %d waves/CU: %.3f ms, %.0f TFLOP/s, %.1f ns per MFMA per SIMD\n", mode ? "zero" : "random", chains, threads / 64, ms, n * 32768 / ms / 1e9, ms * 1e6 / (n / 1024));
                }
            }
        }
    }
    return 0;
}
