// Self-test of the emulator's fp8 matrix instruction (tests/hipemu/hip/hip_runtime.h: v_mfma_scale_f32_32x32x64_f8f6f4, e4m3): one wave, random finite
// operands, three scale settings, against a scalar reference with the operand layout that was verified on an MI355X (tools/ubench/mfma_fp8_mix.hip).
// Built and run by tests/test_hipemu_selftest.py; test infrastructure only.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void one_mfma(float* out, const v8i* a, const v8i* b, int scale_a, int scale_b, int opsel_a) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.5f;   // a non-zero C operand
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, opsel_a, scale_a, 0, scale_b);
    for (int r = 0; r < 16; ++r) out[threadIdx.x * 16 + r] = acc[r];
}

int main() {
    std::vector<unsigned char> ha(64 * 32), hb(64 * 32);
    std::vector<double> A(32 * 64), B(64 * 32);
    unsigned r = 4242;
    for (int lane = 0; lane < 64; ++lane)
        for (int byte = 0; byte < 32; ++byte) {
            r = r * 1664525u + 1013904223u; const unsigned char va = (r >> 9) & 0xf7;     // bit 3 clear: never the NaN patterns 0x7f / 0xff
            r = r * 1664525u + 1013904223u; const unsigned char vb = (r >> 9) & 0xf7;
            ha[lane * 32 + byte] = va; hb[lane * 32 + byte] = vb;
            const int rc = lane & 31, kk = (lane >> 5) * 32 + byte;
            A[rc * 64 + kk] = hipemu_e4m3_to_f32(va); B[kk * 32 + rc] = hipemu_e4m3_to_f32(vb);
        }
    std::vector<float> out(64 * 16);
    int bad = 0;
    const int cases[3][3] = {{127, 127, 0}, {116, 127, 0}, {(120 << 8) | 127, 130, 1}};   // (scale_a register, scale_b register, opsel_a)
    for (auto& cs : cases) {
        hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, out.data(), reinterpret_cast<const v8i*>(ha.data()), reinterpret_cast<const v8i*>(hb.data()), cs[0], cs[1], cs[2]);
        const int sa = (cs[0] >> (8 * cs[2])) & 0xff, sb = cs[1] & 0xff;
        double worst = 0, big = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int rr = 0; rr < 16; ++rr) {
                const int row = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5), col = lane & 31;
                double ref = 0;
                for (int kk = 0; kk < 64; ++kk) ref += A[row * 64 + kk] * B[kk * 32 + col];
                ref = 0.5 + std::ldexp(ref, sa - 127 + sb - 127);
                worst = std::fmax(worst, std::fabs(out[lane * 16 + rr] - ref));
                big = std::fmax(big, std::fabs(ref));
            }
        std::printf("scale_a 2^%d scale_b 2^%d: max |D - ref| %.3g of %.3g\n", sa - 127, sb - 127, worst, big);
        if (!(worst <= 1e-6 * big)) bad = 1;
    }
    // a few decoder anchors: 0x38 = 1.0, 0x7e = 448 (largest), 0x01 = 2^-9 (smallest denormal), 0xc0 = -2.0
    if (hipemu_e4m3_to_f32(0x38) != 1.0f || hipemu_e4m3_to_f32(0x7e) != 448.0f || hipemu_e4m3_to_f32(0x01) != std::ldexp(1.0f, -9) || hipemu_e4m3_to_f32(0xc0) != -2.0f) bad = 1;
    std::printf(bad ? "FAIL\n" : "ok\n");
    return bad;
}
