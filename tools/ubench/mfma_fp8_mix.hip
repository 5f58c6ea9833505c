// micro-benchmark for DESIGN.md 9 item 8: the 3-term split with its two CROSS terms on the fp8 matrix path.
//   per (A fragment, B fragment) pair and K = 64:   f16x3 = 12 x v_mfma_f32_32x32x16_f16 (4 k-steps x 3 products)
//                                                   mixed = 4 x v_mfma_f32_32x32x16_f16 (hi x hi) + 2 x v_mfma_scale_f32_32x32x64_f8f6f4 (the cross terms, e4m3)
// (1) correctness of the fp8 instruction with operands packed position-wise (lane = row / column + 32 x K-half, 32 consecutive K values per lane) against a
//     host reference, incl. the e8m0 scale operand; (2) sustained time per pair for the two mixes and for fp8 alone, random operands, 1 .. 3 waves per SIMD,
//     long enough for the power management to settle -- the part is power-bound on the fp16 MFMA (profiles/r02_mfma_power_ubench*.txt).
// usage: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_fp8_mix.hip -o /tmp/mfma_fp8_mix && /tmp/mfma_fp8_mix
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));

static float e4m3_decode(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}

__global__ void one_mfma(float* out, const v8i* a, const v8i* b, int scale_a, int scale_b) {
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, scale_a, 0, scale_b);
    for (int r = 0; r < 16; ++r) out[threadIdx.x * 16 + r] = acc[r];
}

template <int MIX>
__global__ __launch_bounds__(768) void k(float* out, int iters, const f16x8* in, long long* cyc) {
    long long t0 = clock64();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a[4], b[4];
    v8i a8[2], b8[2];
    for (int i = 0; i < 4; ++i) { a[i] = in[((wave * 8 + i) * 64 + lane) & 4095]; b[i] = in[((wave * 8 + 4 + i) * 64 + lane) & 4095]; }
    for (int i = 0; i < 2; ++i) {
        const f16x8 p = in[((wave * 8 + i) * 64 + lane + 17) & 4095], q = in[((wave * 8 + 4 + i) * 64 + lane + 29) & 4095];
        const f16x8 p2 = in[((wave * 8 + i) * 64 + lane + 170) & 4095], q2 = in[((wave * 8 + 4 + i) * 64 + lane + 290) & 4095];
        int4 pi = __builtin_bit_cast(int4, p), qi = __builtin_bit_cast(int4, q), pj = __builtin_bit_cast(int4, p2), qj = __builtin_bit_cast(int4, q2);
        a8[i] = v8i{pi.x & 0x77777777, pi.y & 0x77777777, pi.z & 0x77777777, pi.w & 0x77777777, pj.x & 0x77777777, pj.y & 0x77777777, pj.z & 0x77777777, pj.w & 0x77777777};   // random finite e4m3 bytes
        b8[i] = v8i{qi.x & 0x77777777, qi.y & 0x77777777, qi.z & 0x77777777, qi.w & 0x77777777, qj.x & 0x77777777, qj.y & 0x77777777, qj.z & 0x77777777, qj.w & 0x77777777};
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // one iteration = 4 (A, B) pairs x K = 64 (the A operand changes with every pair, as in the convolution kernels: 4 pixel fragments per A fragment)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MIX == 0) {           // f16x3: 4 k-steps x (hi hi, hi lo, lo hi)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], b[j], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], b[(j + 1) & 3], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(ks + 1) & 3], b[j], acc[j], 0, 0, 0);
                }
            } else if (MIX == 1) {    // hi hi in fp16, the two cross terms in fp8 (scaled by 2^-11: e8m0 116 on one operand)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], b[j], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[0], b8[j & 1], acc[j], 0, 0, 0, 116, 0, 127);
                acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[1], b8[(j + 1) & 1], acc[j], 0, 0, 0, 127, 0, 116);
            } else {                  // fp8 alone: 3 instructions of K = 64
                acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[0], b8[j & 1], acc[j], 0, 0, 0, 127, 0, 127);
                acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[1], b8[(j + 1) & 1], acc[j], 0, 0, 0, 127, 0, 127);
                acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[0], b8[(j + 1) & 1], acc[j], 0, 0, 0, 127, 0, 127);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = clock64() - t0;
}

template <int MIX>
static double run(const char* what, int threads, float* out, const f16x8* in, long long* cyc, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MIX>, dim3(256), dim3(threads), 0, 0, out, iters / 10, in, cyc);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MIX>, dim3(256), dim3(threads), 0, 0, out, iters, in, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long hc; (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    const double pairs = 256.0 * (threads / 64) * (double)iters * 4;                  // (A, B) pairs of K = 64
    const double ns_per_pair_simd = ms * 1e6 / ((double)iters * 4 * (threads / 256));  // per SIMD: waves / 4 SIMDs share the pipe
    const double tf_alg = pairs * 2.0 * 32 * 32 * 64 / (ms * 1e-3) / 1e12;            // algorithmic (fp32-equivalent) flops of the products
    printf("%-34s waves/SIMD=%d : %8.2f ms  clock64 %.2f GHz  %6.1f ns per pair and SIMD  %5.0f TF fp32-equivalent\n", what, threads / 256, ms, hc / (ms * 1e6), ns_per_pair_simd, tf_alg);
    return ms;
}

int main() {
    // ---- (1) correctness: D = A B^T-style product of a 32 x 64 and a 64 x 32 fp8 matrix, operands packed position-wise
    {
        std::vector<unsigned char> ha(64 * 32), hb(64 * 32);
        std::vector<float> A(32 * 64), B(64 * 32);
        unsigned r = 777;
        for (int lane = 0; lane < 64; ++lane)
            for (int byte = 0; byte < 32; ++byte) {
                r = r * 1664525u + 1013904223u; unsigned char va = (r >> 9) & 0xf7; if ((va & 0x7f) == 0x7f) va ^= 1;   // no NaN
                r = r * 1664525u + 1013904223u; unsigned char vb = (r >> 9) & 0xf7; if ((vb & 0x7f) == 0x7f) vb ^= 1;
                ha[lane * 32 + byte] = va; hb[lane * 32 + byte] = vb;
                const int rc = lane & 31, kk = (lane >> 5) * 32 + byte;      // assumed: row / column = lane % 32, K = 32 (lane / 32) + byte
                A[rc * 64 + kk] = e4m3_decode(va); B[kk * 32 + rc] = e4m3_decode(vb);
            }
        v8i *da, *db; float* dout;
        (void)hipMalloc(&da, 2048); (void)hipMalloc(&db, 2048); (void)hipMalloc(&dout, 64 * 16 * 4);
        (void)hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice);
        for (int sc = 0; sc < 2; ++sc) {
            const int sa = sc ? 116 : 127, sb = 127;
            hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, dout, da, db, sa, sb);
            std::vector<float> hd(64 * 16); (void)hipMemcpy(hd.data(), dout, 64 * 16 * 4, hipMemcpyDeviceToHost);
            double worst = 0, big = 0;
            for (int lane = 0; lane < 64; ++lane)
                for (int rr = 0; rr < 16; ++rr) {
                    const int row = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5), col = lane & 31;
                    double ref = 0; for (int kk = 0; kk < 64; ++kk) ref += (double)A[row * 64 + kk] * B[kk * 32 + col];
                    ref *= ldexp(1.0, sa - 127);
                    worst = fmax(worst, fabs(hd[lane * 16 + rr] - ref)); big = fmax(big, fabs(ref));
                }
            // (a wrong pairing of the K positions would be off by O(max |ref|); the instruction itself sums its 64 products with ~14-15 bits: 5e-5 relative measured)
            printf("fp8 32x32x64, operands packed position-wise, scale_a 2^%d: max |D - ref| = %.3g (max |ref| %.3g, relative %.1e) -> %s\n", sa - 127, worst, big, worst / big,
                   worst <= 1e-3 * big ? "layout and scale as assumed" : "MISMATCH");
        }
    }
    // ---- (2) sustained rates
    float* out; f16x8* in; long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&in, 4096 * 16); (void)hipMalloc(&cyc, 8);
    _Float16* h = (_Float16*)malloc(4096 * 16);
    unsigned r = 12345;
    for (int i = 0; i < 4096 * 8; ++i) {
        r = r * 1664525u + 1013904223u;
        const float u = ((r >> 8) & 0xffff) / 65536.0f, v = ((r >> 4) & 1) ? 1.f : -1.f;
        h[i] = (_Float16)(v * (0.05f + 2.0f * u));
    }
    (void)hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
    const int iters = 40000;
    for (int threads = 256; threads <= 768; threads += 256) {
        const int it = iters * 256 / threads * 2;
        const double t0 = run<0>("f16x3 (12 fp16 MFMAs per pair)", threads, out, in, cyc, it);
        const double t1 = run<1>("mixed (4 fp16 + 2 fp8 MFMAs)", threads, out, in, cyc, it);
        run<2>("fp8 alone (3 fp8 MFMAs of K = 64)", threads, out, in, cyc, it);
        printf("   -> mixed / f16x3 time: %.3f\n", t1 / t0);
    }
    return 0;
}
