// Profiling build only (liblama_hip_prof.so): synthetic co-resident loads and one-instruction probes of tools/race_probe8.py /
// race_probe9.py -- the experiments behind DESIGN.md 4.3 (packed-fp32 VALU instructions with an op_sel swizzle are corrupted by
// another kernel's MFMA on the same SIMD).  This is the ONE translation unit that is compiled WITH packed-fp32 instructions: it
// has to issue them on purpose.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// profiling build only (tools/race_probe8.py): synthetic co-resident load, one 512-thread workgroup per CU for ~iters rounds.
//   mode 0: VALU spin   1: LDS traffic inside its own 48 KB (no barrier)   2: s_barrier loop   3: MFMA loop   4: LDS + barrier
//   mode 5: like 1 but the LDS accesses run PAST the 48 KB it asked for (does the hardware clamp them?)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void debug_hog_kernel(int mode, int iters, float* out) {
    typedef float f32x16_t __attribute__((ext_vector_type(16)));
    typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
    float* lds = reinterpret_cast<float*>(lama_smem);
    const int tid = threadIdx.x;
    float acc = (float)tid;
    f32x16_t c;
    for (int r = 0; r < 16; ++r) c[r] = 0.0f;
    f16x8_t a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (_Float16)(0.001f * tid); b[r] = (_Float16)0.5f; }
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) {
            for (int k = 0; k < 64; ++k) acc = acc * 1.0001f + 0.5f;
        } else if (mode == 1 || mode == 4 || mode == 5) {
            const int span = mode == 5 ? 40960 : 12288;     // floats: 160 KB vs the 48 KB requested
            for (int k = 0; k < 24; ++k) {
                const int idx = (tid + k * 512 + it * 7) % span;
                lds[idx] = acc + (float)k;
                acc += lds[(idx + 64) % span];
            }
            if (mode == 4) __syncthreads();
        } else if (mode == 2) {
            acc += 1.0f;
            __syncthreads();
        } else if (mode == 3) {
            for (int k = 0; k < 8; ++k) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        }
    }
    if (out) out[blockIdx.x * 512 + tid] = acc + c[0] + c[5];
}

// mini-probes: what kind of work goes wrong next to the MFMA load?  256-thread workgroups like the FFT kernels, n = floats per workgroup slice (4096)
//   mode 0: VALU only (fma chain on a global value)   1: LDS round trip with barriers (write, barrier, read transposed)
//   mode 2: the twiddle table of the FFT kernels (sincospif) written out   3: one radix-8 butterfly per thread on global data
//   mode 4: v_sin / v_cos style fast transcendentals (__sinf, __cosf, __expf)   5: LDS round trip WITHOUT transposition
__global__ __launch_bounds__(256) void debug_probe_kernel(int mode, const float* in, float* out) {
    const int tid = threadIdx.x;
    const float* src = in + (long long)blockIdx.x * 4096;
    float* dst = out + (long long)blockIdx.x * 4096;
    float* lds = reinterpret_cast<float*>(lama_smem);
    if (mode == 0) {
        for (int k = 0; k < 16; ++k) {
            float v = src[tid + k * 256];
            for (int r = 0; r < 64; ++r) v = fmaf(v, 1.0001f, 0.25f);
            dst[tid + k * 256] = v;
        }
    } else if (mode == 1 || mode == 5) {
        for (int rep = 0; rep < 8; ++rep) {
            for (int k = 0; k < 16; ++k) lds[tid + k * 256] = src[tid + k * 256] + (float)rep;
            __syncthreads();
            float v[16];
            for (int k = 0; k < 16; ++k) v[k] = mode == 1 ? lds[(tid * 16 + k + rep) & 4095] : lds[((tid + 64) & 255) + k * 256];
            __syncthreads();
            for (int k = 0; k < 16; ++k) dst[tid + k * 256] = v[k];
        }
    } else if (mode == 2) {
        for (int m = tid; m < 2048; m += 256) {
            float sn, cs;
            sincospif(2.0f * (float)(m & 63) / 64.0f, &sn, &cs);
            dst[2 * m] = cs;
            dst[2 * m + 1] = -sn;
        }
    } else if (mode == 3) {
        float2 v[8];
        for (int r = 0; r < 8; ++r) v[r] = make_float2(src[tid * 16 + 2 * r], src[tid * 16 + 2 * r + 1]);
        // radix-8 butterfly written out with plain adds / muls
        float2 a0 = make_float2(v[0].x + v[4].x, v[0].y + v[4].y), a1 = make_float2(v[1].x + v[5].x, v[1].y + v[5].y);
        float2 a2 = make_float2(v[2].x + v[6].x, v[2].y + v[6].y), a3 = make_float2(v[3].x + v[7].x, v[3].y + v[7].y);
        float2 b0 = make_float2(v[0].x - v[4].x, v[0].y - v[4].y), b1 = make_float2(v[1].x - v[5].x, v[1].y - v[5].y);
        float2 b2 = make_float2(v[2].x - v[6].x, v[2].y - v[6].y), b3 = make_float2(v[3].x - v[7].x, v[3].y - v[7].y);
        const float s = 0.70710678118654752440f;
        b1 = make_float2(s * (b1.x + b1.y), s * (b1.y - b1.x));
        b2 = make_float2(b2.y, -b2.x);
        b3 = make_float2(s * (b3.y - b3.x), -s * (b3.x + b3.y));
        float2 o[8] = {make_float2(a0.x + a2.x + a1.x + a3.x, a0.y + a2.y + a1.y + a3.y), make_float2(b0.x + b2.x + b1.x + b3.x, b0.y + b2.y + b1.y + b3.y),
                       make_float2(a0.x - a2.x + a1.y - a3.y, a0.y - a2.y - a1.x + a3.x), make_float2(b0.x - b2.x + b1.y - b3.y, b0.y - b2.y - b1.x + b3.x),
                       make_float2(a0.x + a2.x - a1.x - a3.x, a0.y + a2.y - a1.y - a3.y), make_float2(b0.x + b2.x - b1.x - b3.x, b0.y + b2.y - b1.y - b3.y),
                       make_float2(a0.x - a2.x - a1.y + a3.y, a0.y - a2.y + a1.x - a3.x), make_float2(b0.x - b2.x - b1.y + b3.y, b0.y - b2.y + b1.x - b3.x)};
        for (int r = 0; r < 8; ++r) { dst[tid * 16 + 2 * r] = o[r].x; dst[tid * 16 + 2 * r + 1] = o[r].y; }
    } else if (mode == 4) {
        for (int k = 0; k < 16; ++k) {
            const float v = src[tid + k * 256];
            dst[tid + k * 256] = __sinf(v) + __cosf(v * 0.5f) + __expf(-v * v);
        }
    } else if (mode == 9 || mode == 10) {   // 9: v_pk_add_f32 with an op_sel half-swizzle; 10: v_pk_mul_f32 with an SGPR-pair source
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        for (int k = 0; k < 8; ++k) {
            f32x2_t v = {src[tid * 2 + k * 512], src[tid * 2 + 1 + k * 512]};
            const f32x2_t u = {0.001f, -0.002f};
            f32x2_t sc = {1.0001f, 0.9999f};
            for (int r = 0; r < 48; ++r) {
                if (mode == 9) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(v) : "v"(v), "v"(u)); }
                else { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(v) : "v"(v), "s"(sc)); }
            }
            dst[tid * 2 + k * 512] = v.x;
            dst[tid * 2 + 1 + k * 512] = v.y;
        }
    } else if (mode >= 6 && mode <= 8) {   // one packed-fp32 opcode at a time: 6 v_pk_add_f32, 7 v_pk_mul_f32, 8 v_pk_fma_f32
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        for (int k = 0; k < 8; ++k) {
            f32x2_t v = {src[tid * 2 + k * 512], src[tid * 2 + 1 + k * 512]};
            const f32x2_t w = {0.75f, 1.25f}, u = {0.001f, -0.002f};
            for (int r = 0; r < 48; ++r) {
                if (mode == 6) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(u)); }
                else if (mode == 7) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(r & 1 ? w : (f32x2_t){1.0f / 0.75f, 0.8f})); }
                else { asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(v), "v"((f32x2_t){1.0001f, 0.9999f}), "v"(u)); }
            }
            dst[tid * 2 + k * 512] = v.x;
            dst[tid * 2 + 1 + k * 512] = v.y;
        }
    }
}

extern "C" int lama_debug_probe(void* stream, int32_t grid, int32_t mode, const float* in, float* out) {
    hipLaunchKernelGGL(debug_probe_kernel, dim3(grid), dim3(256), 16384, (hipStream_t)stream, mode, in, out);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

extern "C" int lama_debug_hog(void* stream, int32_t grid, int32_t mode, int32_t iters, float* out) {
    hipLaunchKernelGGL(debug_hog_kernel, dim3(grid), dim3(512), 49152, (hipStream_t)stream, mode, iters, out);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

// ------------------------------------------------------------------------------------------------
// Sustained MFMA ceiling of THIS box (bench.py's roofline.peak_sustained; stand-alone version: tools/ubench/mfma_power.hip): one
// 256-thread workgroup per CU, every wave keeps 4 A and 4 B fragments of the caller's data in registers and issues
// v_mfma_f32_32x32x16_f16 over all 16 pairs, so the operands change with every instruction and nothing else runs.  On MI355X the
// rate depends on the DATA (power): all-zero operands reach 98 % of the 2.5 PF spec, random operands 67-71 %.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void debug_mfma_peak_kernel(int iters, const uint4* in, float* out) {
    typedef float f32x16_t __attribute__((ext_vector_type(16)));
    typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8_t a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(f16x8_t, in[((wave * 8 + i) * 64 + lane) & 4095]);
        b[i] = __builtin_bit_cast(f16x8_t, in[((wave * 8 + 4 + i) * 64 + lane) & 4095]);
    }
    f32x16_t acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[j], 0, 0, 0);
    }
    float s = 0.0f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (out) out[blockIdx.x * 256 + threadIdx.x] = s;
}

// in: 4096 x 16 bytes of fp16 operand data (device), out: 256 * 256 floats (device) or NULL; 256 workgroups x iters x 16 MFMAs per wave,
// 4 waves per workgroup -> flops = 256 * 4 * iters * 16 * 32768
extern "C" int lama_debug_mfma_peak(void* stream, int32_t iters, const void* in, float* out) {
    hipLaunchKernelGGL(debug_mfma_peak_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, iters, (const uint4*)in, out);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}
