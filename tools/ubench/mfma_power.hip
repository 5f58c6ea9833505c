// micro-benchmark: sustained v_mfma_f32_32x32x16_f16 rate and core clock when the operands CHANGE with every instruction (4 A x 4 B
// register fragments per wave, the register reuse pattern of the convolution kernels, no memory traffic in the loop), for constant /
// random / split-like (hi, lo) operand data, 1 .. 3 waves per SIMD, long enough (tens of ms) for the power management to settle.
// usage: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(768) void k(float* out, int iters, const f16x8* in, long long* cyc) {
    long long t0 = clock64();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[((wave * 8 + i) * 64 + lane) & 4095]; b[i] = in[((wave * 8 + 4 + i) * 64 + lane) & 4095]; }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[j], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = clock64() - t0;
}

static void run(const char* what, int threads, float* out, const f16x8* in, long long* cyc, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, iters / 10, in, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, iters, in, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = 256.0 * (threads / 64) * (double)iters * 16 * 32768.0 / (ms * 1e-3) / 1e12;   // 32x32x16 MFMA = 16384 MAC = 32768 flop
    long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s waves/SIMD=%d : %8.2f ms  clock64 %.2f GHz  %.0f TF dense f16 (%.1f %% of 2500)\n", what, threads / 256, ms, hc / (ms * 1e6), tf, tf / 25.0);
}

int main() {
    float* out; f16x8* in; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&in, 4096 * 16); hipMalloc(&cyc, 8);
    _Float16* h = (_Float16*)malloc(4096 * 16);
    const int iters = 400000;
    for (int mode = 0; mode < 4; ++mode) {
        unsigned r = 12345;
        for (int i = 0; i < 4096 * 8; ++i) {
            r = r * 1664525u + 1013904223u;
            const float u = ((r >> 8) & 0xffff) / 65536.0f, v = ((r >> 4) & 1) ? 1.f : -1.f;
            float x = 0.f;
            if (mode == 1) x = 1.0f;                                   // constant
            else if (mode == 2) x = v * (0.05f + 2.0f * u);            // random sign / mantissa, activations-like magnitude
            else if (mode == 3) x = ((i / (8 * 64)) & 1) ? v * 2.0f * u : v * 2.0f * u * 4.8e-4f;   // every other fragment a lo part (2^-11 smaller)
            h[i] = (_Float16)x;
        }
        hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
        const char* names[4] = {"zero operands", "constant 1.0", "random operands", "random hi / lo fragments"};
        run(names[mode], 256, out, in, cyc, iters * 2);
        run(names[mode], 512, out, in, cyc, iters);
        run(names[mode], 768, out, in, cyc, iters * 2 / 3);
    }
    return 0;
}
