// micro-benchmark: sustained f16 MFMA rate with random operands, 32x32x16 vs 16x16x32 (same flops per iteration)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, const int* in) {
    const int lane = threadIdx.x & 63;
    f16x8 a[3], b[4];
    for (int i = 0; i < 3; ++i) a[i] = *reinterpret_cast<const f16x8*>(in + (i * 64 + lane) * 4);
    for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const f16x8*>(in + ((3 + i) * 64 + lane) * 4);
    float s = 0;
    if (MODE == 0) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 3; ++rep)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rep], b[i], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 3; ++rep)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rep], b[i & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* what, float* out, const int* in) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, 10, in);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, in);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double tf = 256.0 * 8 * iters * 12 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-36s %8.1f us per launch  %.0f TFLOP/s\n", what, ms * 1e3, tf);
}

int main() {
    float* out; int* in;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&in, 2048 * 4);
    static int h[2048];
    for (int rnd = 0; rnd < 2; ++rnd) {
        unsigned r = 777;
        for (int i = 0; i < 2048; ++i) { r = r * 1664525u + 1013904223u; h[i] = rnd ? (int)((r & 0x83ff83ffu) | 0x38003800u) : 0x3c003c00; }
        (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        printf(rnd ? "random operands\n" : "constant operands\n");
        run<0>("12 x mfma_f32_32x32x16_f16", out, in);
        run<1>("24 x mfma_f32_16x16x32_f16", out, in);
        run<0>("12 x mfma_f32_32x32x16_f16", out, in);
        run<1>("24 x mfma_f32_16x16x32_f16", out, in);
    }
    return 0;
}
