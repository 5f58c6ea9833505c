// micro-benchmark: v_mfma_f32_32x32x16_bf16 issue rate vs. number of independent accumulator chains and waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters, const bf16x8* in, long long* cyc) {
    long long t0 = clock64();
    bf16x8 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 12 / NACC; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = clock64() - t0;
}

template <int NACC>
void run(int threads, float* out, const bf16x8* in, long long* cyc) {
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, 10, in, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, iters, in, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma_per_simd = (double)iters * 12 * (threads / 64) / 4.0;
    double tf = 256.0 * (threads / 64) * iters * 12 * 32768.0 / (ms * 1e-3) / 1e12;
    long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("clk %.2f GHz  NACC=%d waves/SIMD=%d : %.1f us, %.1f ns per MFMA per SIMD (= %.1f cyc @2.4GHz), %.0f TF\n", hc / (ms * 1e6), NACC, threads / 256, ms * 1e3,
           ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, tf);
    (void)0;
}

int main() {
    float* out; bf16x8* in; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&in, 128 * 16); hipMalloc(&cyc, 8); hipMemset(in, 0x3c, 128 * 16);
    printf("constant operands\n");
    run<4>(256, out, in, cyc); run<4>(512, out, in, cyc); run<4>(512, out, in, cyc);
    unsigned short h[128 * 8]; unsigned r = 12345;
    for (int i = 0; i < 128 * 8; ++i) { r = r * 1664525u + 1013904223u; h[i] = (unsigned short)(((r >> 16) & 0x807f) | 0x3f00 | ((r >> 9) & 0x80)); }   // random sign / mantissa, |x| in [0.5, 2)
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    printf("random operands\n");
    run<4>(256, out, in, cyc); run<4>(512, out, in, cyc); run<4>(512, out, in, cyc); run<2>(1024, out, in, cyc);
    return 0;
}
