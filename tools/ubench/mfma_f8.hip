// micro-benchmark: sustained rate of v_mfma_f32_32x32x16_f16 vs v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3, MX scales) with
// random operands (the matrix cores of gfx950 are power-limited with toggling inputs), and of a 2:1 mix of the two
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0: 12 f16 MFMAs per iteration; 1: 6 fp8 K=64 MFMAs; 2: 8 f16 + 2 fp8 (same algorithmic work as 12 f16 in the split scheme)
__global__ __launch_bounds__(512) void k(float* out, int iters, const int* in) {
    const int lane = threadIdx.x & 63;
    f16x8 a[2], b[4];
    i32x8 a8, b8[4];
    for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f16x8*>(in + (i * 64 + lane) * 4);
    for (int i = 0; i < 4; ++i) b[i] = *reinterpret_cast<const f16x8*>(in + ((2 + i) * 64 + lane) * 4);
    a8 = *reinterpret_cast<const i32x8*>(in + 2048 + lane * 8);
    for (int i = 0; i < 4; ++i) b8[i] = *reinterpret_cast<const i32x8*>(in + 4096 + (i * 64 + lane) * 8);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int rep = 0; rep < 3; ++rep)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rep & 1], b[i], acc[i], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int rep = 0; rep < 3; ++rep)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[i + 2 * (rep & 1)] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8[i + (rep & 1)], acc[i + 2 * (rep & 1)], 0, 0, 0, 127, 0, 127);
        } else {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rep], b[i], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8[i], acc[i], 0, 0, 0, 127, 0, 127);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
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
    printf("%-44s %8.1f us per launch, %.2f ns per iteration per wave pair\n", what, ms * 1e3, ms * 1e6 / iters);
}

int main() {
    float* out; int* in;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&in, 8192 * 4);
    static int h[8192];
    for (int rnd = 0; rnd < 2; ++rnd) {
        unsigned r = 12345;
        for (int i = 0; i < 8192; ++i) {
            r = r * 1664525u + 1013904223u;
            unsigned v = rnd ? r : 0x3c003c00u;
            if (i < 2048) v = rnd ? ((r & 0x83ff83ffu) | 0x38003800u) : 0x3c003c00u;      // f16 pairs, |x| in [0.5, 1)
            else v = rnd ? ((r & 0x87878787u) | 0x30303030u) : 0x38383838u;               // e4m3 bytes, moderate exponents
            h[i] = (int)v;
        }
        (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        printf(rnd ? "random operands\n" : "constant operands\n");
        run<0>("12 x mfma_f32_32x32x16_f16", out, in);
        run<1>("6 x mfma_scale_f32_32x32x64_f8f6f4 (fp8)", out, in);
        run<2>("8 x f16 + 2 x fp8 K=64", out, in);
        run<0>("12 x mfma_f32_32x32x16_f16", out, in);
    }
    return 0;
}
