// probe of the gfx950 fp32 -> fp8 (e4m3) conversions the fp16 + 2 x fp8 arithmetic would stage its cross-term operands with:
// v_cvt_pk_fp8_f32 and v_cvt_scalef32_pk_fp8_f32 -- rounding of ties, saturation, denormals, and what the scale operand does.
// usage: hipcc --offload-arch=gfx950 -O3 tools/ubench/fp8_cvt_probe.hip -o /tmp/fp8_cvt_probe && /tmp/fp8_cvt_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef short v2s __attribute__((ext_vector_type(2)));
__global__ void k(int* out, const float* in, float scale) {
    const float a = in[threadIdx.x];
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, 0.0f, v, false);
    v2s old = {0, 0};
    const v2s r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a, 0.0f, scale, false);
    out[threadIdx.x * 2] = v & 0xff;
    out[threadIdx.x * 2 + 1] = r[0] & 0xff;
}
static float dec(int v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    if (e == 15 && m == 7) return NAN;
    const float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}
int main() {
    const float h[] = {1.0f, 1.0625f, 1.1875f, 1.03f, 0.3f, -0.3f, 448.0f, 464.0f, 500.0f, 1e6f, 0.001953125f, 0.0009765625f, 0.0029296875f, 1e-4f, 3.0f, 15.5f, 240.0f, 0.0f};
    const int n = sizeof(h) / sizeof(h[0]);
    float* din; int* dout; int ho[2 * 32];
    (void)hipMalloc(&din, n * 4); (void)hipMalloc(&dout, 2 * n * 4);
    (void)hipMemcpy(din, h, n * 4, hipMemcpyHostToDevice);
    const float scales[3] = {1.0f, 4.0f, 0.25f};
    for (float sc : scales) {
        hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, dout, din, sc);
        (void)hipMemcpy(ho, dout, 2 * n * 4, hipMemcpyDeviceToHost);
        printf("scale operand %g:\n", sc);
        for (int i = 0; i < n; ++i) printf("  x = %-14.9g  cvt_pk_fp8 0x%02x = %-10g   cvt_scalef32_pk_fp8 0x%02x = %-10g\n", h[i], ho[2 * i], dec(ho[2 * i]), ho[2 * i + 1], dec(ho[2 * i + 1]));
    }
    return 0;
}
