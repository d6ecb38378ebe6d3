// Micro-benchmark: what the ACCESS SHAPE of a wave's 16-B loads costs.  One 32-point x 128-B tile (a head's 32 channels of
// 32 points of a [N][1 KB] fp32 tensor) per wave and step, read as four wave-wide 16-B loads in two lane -> address maps:
//   ROWS   lane (p = lane % 32, hf = lane / 32) reads ITS point's bytes 64 hf + 16 q .. (q = 0..3): neighbouring lanes sit
//          in different rows, 1 KB apart -- what slice_bwd_x3.hip / slice_bwd_mfma.hip / slice_f32_mfma.hip do
//   LINES  lane (r = lane / 8, c = lane % 8) reads row r + 8 q, 16-B chunk c: eight neighbouring lanes = one 128-B line
// Same bytes, same lines, same number of instructions.  Optionally the same two maps for 16-B STORES.
// hipcc --offload-arch=gfx950 -O3 load_shape.hip -o load_shape && ./load_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE, int STORE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ x, float* __restrict__ y, int N, int H, int chunk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.y, n0 = blockIdx.x * chunk, n1 = min(N, n0 + chunk);
    const size_t row = (size_t)H * 32;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = n0 + 32 * wave; t < n1; t += 32 * 8) {
        f32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            size_t off;
            if (SHAPE == 0) off = (size_t)(t + (lane & 31)) * row + h * 32 + 16 * (lane >> 5) + 4 * q;
            else off = (size_t)(t + (lane >> 3) + 8 * q) * row + h * 32 + 4 * (lane & 7);
            v[q] = *reinterpret_cast<const f32x4*>(x + off);
            if (STORE) *reinterpret_cast<f32x4*>(y + off) = v[q] * 2.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += v[q];
    }
    if (!STORE && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) y[0] = 1.f;
}

template <int SHAPE, int STORE>
static float run(const float* x, float* y, int N, int H) {
    const int chunk = ((N + 255) / 256 + 255) / 256 * 256;
    dim3 grid((N + chunk - 1) / chunk, H);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<SHAPE, STORE>), grid, dim3(512), 0, 0, x, y, N, H, chunk);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<SHAPE, STORE>), grid, dim3(512), 0, 0, x, y, N, H, chunk);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / 10;
}

int main() {
    const int N = 1 << 20, H = 8;
    float *x, *y;
    hipMalloc(&x, (size_t)N * H * 32 * 4);
    hipMalloc(&y, (size_t)N * H * 32 * 4);
    hipMemset(x, 0, (size_t)N * H * 32 * 4);
    const double gb = (double)N * H * 32 * 4 / 1e9;
    const float a = run<0, 0>(x, y, N, H), b = run<1, 0>(x, y, N, H), c = run<0, 1>(x, y, N, H), d = run<1, 1>(x, y, N, H);
    printf("loads only   ROWS %.3f ms (%.2f TB/s)   LINES %.3f ms (%.2f TB/s)\n", a, gb / a, b, gb / b);
    printf("load + store ROWS %.3f ms (%.2f TB/s)   LINES %.3f ms (%.2f TB/s)\n", c, 2 * gb / c, d, 2 * gb / d);
    return 0;
}
