// Shared declarations for the gfx950 kernels of liblama_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "lama_hip.h"

#define LAMA_NTHREADS 256

// All LDS lives in the dynamic region (16-byte aligned base, no static __shared__ in front of it).
extern __shared__ __attribute__((aligned(16))) char lama_smem[];

#define LAMA_CHECK_LAUNCH()                      \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

// Kernel-selection overrides, timing ablations (which produce WRONG results by design) and timeline tracers (which write to a
// device address taken from the environment) are hooks of tools/ and of the forced-path tests: they exist only in builds with
// -DLAMA_PROFILING (lama_amd/lib/liblama_hip_prof.so, tests/hipemu).  The shipped liblama_hip.so reads no environment variable.
#ifdef LAMA_PROFILING
#include <stdlib.h>
static inline int lama_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline unsigned long long lama_env_u64(const char* name) { const char* e = getenv(name); return e ? strtoull(e, nullptr, 0) : 0ull; }
#else
static inline constexpr int lama_env_int(const char*, int dflt) { return dflt; }
static inline constexpr unsigned long long lama_env_u64(const char*) { return 0ull; }
#endif

static inline int lama_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t lama_ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int lama_round_up(int a, int b) { return lama_ceil_div(a, b) * b; }
static inline bool lama_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int lama_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// Activation element types in HBM (lama_tensor.dtype): LAMA_DT_F32 or LAMA_DT_F16.  Kernels are templated on a bool (half I/O).
template <bool H> struct LamaAct { typedef float T; static constexpr int ES = 4; };
template <> struct LamaAct<true> { typedef _Float16 T; static constexpr int ES = 2; };

// XCD-aware remap of the linear workgroup id: hardware places block i on XCD i % 8; give every XCD a
// contiguous range of logical ids so neighbouring tiles (shared input patches / weights) hit one L2.
// Bijective for any grid size (cdna_hip_programming.md section 5, "XCD swizzle must be bijective").
__device__ __forceinline__ int lama_xcd_remap(int orig, int nwg) {
    int xcd = orig & 7, idx = orig >> 3;
    int q = nwg >> 3, r = nwg & 7;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// LDS (address space 3) pointer for __builtin_amdgcn_global_load_lds; the destination is the
// wave-uniform base, the hardware adds lane * size.  (tests/hipemu overrides this for the host build.)
#ifndef LAMA_LDS_PTR
#define LAMA_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#endif

// address space of LDS in pointer TYPES (device functions that are not inlined lose the inference: flat instead of ds instructions).
// (tests/hipemu defines it away.)
#ifndef LAMA_LDS_AS
#define LAMA_LDS_AS __attribute__((address_space(3)))
#endif

// keep a value live without cost (timing ablations must not let the compiler delete the work that produced it)
#ifndef LAMA_KEEP_LIVE
#define LAMA_KEEP_LIVE(x) asm volatile("" ::"v"(x))
#endif

// Raw buffer loads: 128-bit resource (base, byte size) in SGPRs + 32-bit lane offset + 32-bit scalar offset.  One address
// VGPR serves any number of loads that differ only in the (uniform) scalar offset, and out-of-range offsets read 0.
// (tests/hipemu overrides these for the host build.)
#ifndef LAMA_BUF_RSRC
typedef __amdgpu_buffer_rsrc_t lama_buf_t;
// the descriptor inputs go through readfirstlane so that hipcc can PROVE them wave-uniform (otherwise every buffer op is
// wrapped in a waterfall loop: cdna_hip_programming.md T20)
__device__ __forceinline__ lama_buf_t lama_make_buf(const void* ptr, long long bytes) {
    const unsigned long long a = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
#define LAMA_BUF_RSRC(ptr, bytes) lama_make_buf(ptr, bytes)
#define LAMA_BUF_LOAD_B32(rsrc, voff, soff) __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0)
#define LAMA_BUF_LOAD_B64(rsrc, voff, soff) __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0)
#define LAMA_BUF_LOAD_B128(rsrc, voff, soff) __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0)
#define LAMA_BUF_LOAD_B16(rsrc, voff, soff) __builtin_amdgcn_raw_buffer_load_b16(rsrc, voff, soff, 0)     // 16-bit element, zero-extended
#define LAMA_BUF_STORE_B16(rsrc, val, voff, soff) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(val), rsrc, voff, soff, 0)
// out-of-range buffer stores are dropped: a predicated store without a branch
#define LAMA_BUF_STORE_B32(rsrc, val, voff, soff) __builtin_amdgcn_raw_buffer_store_b32(val, rsrc, voff, soff, 0)
#define LAMA_BUF_STORE_B64(rsrc, val, voff, soff) __builtin_amdgcn_raw_buffer_store_b64(val, rsrc, voff, soff, 0)     // val: 2 x u32 vector
#define LAMA_BUF_STORE_B128(rsrc, val, voff, soff) __builtin_amdgcn_raw_buffer_store_b128(val, rsrc, voff, soff, 0)   // val: 4 x u32 vector
#define LAMA_WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#define LAMA_CLOCK() ((long long)wall_clock64())   // 100 MHz constant counter (timeline traces of the profiling tools)
#define LAMA_CYCLES() ((long long)__builtin_readcyclecounter())   // s_memtime (k-step stamps of the profiling tools)
#endif

// x - float(h) for the two fp16 halves of a packed word in ONE VALU instruction each (v_fma_mix_f32 reads an fp16 half directly;
// hipcc emits v_cvt_f32_f16 + v_sub_f32 for the same expression).  (tests/hipemu overrides these for the host build.)
#ifndef LAMA_F16_RESIDUAL_LO
#define LAMA_F16_RESIDUAL_LO(d, packed, x) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "v"(x))
#define LAMA_F16_RESIDUAL_HI(d, packed, x) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "v"(x))
#endif
// The lo word of an fp16 (hi, lo) split in TWO VALU instructions (round 6, fourth session): lo = { fp16(a - float(hi.lo)), fp16(b - float(hi.hi)) }.
// v_fma_mixlo_f16 / v_fma_mixhi_f16 form the fp32 residual (exact: |x - fp16(x)| <= half an fp16 ulp of x = at most 13 significant bits) and round it to
// fp16 INTO the low / high half of the destination, keeping its other half -- the residual + v_cvt_pk_f16_f32 of the three-instruction form in one.  Same
// bits: the intermediate fp32 residual is exact, so one rounding (RNE) either way.  (tests/hipemu overrides it for the host build.)
// Halo differences of the Winograd staging transform inside a 16-lane DPP row (wino_dev.inc, GEO 1): the subtraction itself carries the DPP operand, and the
// row's edge lane (no source lane) keeps the destination's previous value -- the caller's 0 = the difference with the reflected column.
//   LAMA_ROW_SHR1_SUB(d, s, b):  d = s[lane - 1] - b   (lanes 1 .. 15 of a row; lane 0: d = 0)
//   LAMA_ROW_SHL1_RSUB(d, s, b): d = b - s[lane + 1]   (lanes 0 .. 14 of a row; lane 15: d = 0)
// Two VALU instructions (v_mov d, 0 + v_sub / v_subrev_f32_dpp) instead of three (v_mov old + v_mov_b32_dpp + v_sub).  (tests/hipemu overrides them.)
#ifndef LAMA_ROW_SHR1_SUB
#define LAMA_ROW_SHR1_SUB(d, s, b) do { (d) = 0.0f; asm("v_sub_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(s), "v"(b)); } while (0)
#define LAMA_ROW_SHL1_RSUB(d, s, b) do { (d) = 0.0f; asm("v_subrev_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(s), "v"(b)); } while (0)
#endif
#ifndef LAMA_F16_SPLIT_LO
#define LAMA_F16_SPLIT_LO(lo, packed, a, b)                                                                                         \
    do {                                                                                                                            \
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(packed), "v"(a));                    \
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(packed), "v"(b));                    \
    } while (0)
#endif

// LDS hand-off between the lanes of ONE wave (wave-private LDS regions, no workgroup barrier): the LDS pipe executes a wave's
// instructions in order, so only the compiler has to be kept from moving accesses across this point.  (tests/hipemu maps it
// to its per-wave barrier.)
#ifndef LAMA_WAVE_SYNC
#define LAMA_WAVE_SYNC()                                              \
    do {                                                              \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        \
        __builtin_amdgcn_wave_barrier();                              \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");        \
    } while (0)
#endif

// keep a value in ACCUMULATION registers from here on (gfx90a+: one unified 512-entry file per lane at one wave per SIMD, but VALU / memory
// instructions address the 256 architectural VGPRs only, MFMA operands either half): an empty asm with an "a" constraint.
// (tests/hipemu: no-op.)
#ifndef LAMA_PIN_AGPR
#define LAMA_PIN_AGPR(x) asm volatile("" : "+a"(x))
#endif

// make a per-lane integer opaque to the optimiser at this point of the program (pins the loads that depend on it behind the
// code above: epilogue loads must not be hoisted over the main loop, where their registers are needed)
#ifndef LAMA_OPAQUE
#define LAMA_OPAQUE(x) asm volatile("" : "+v"(x))
#endif
// the same for a wave-uniform integer (scalar register): inside a loop it keeps the address arithmetic that depends on it INSIDE the loop
// (hoisted out of a fully unrolled body, a hundred loop-invariant scalar offsets spill)
#ifndef LAMA_OPAQUE_S
#define LAMA_OPAQUE_S(x) asm volatile("" : "+s"(x))
#endif

// 3-term split convolution back ends (conv_split3.inc compiled as conv_bf16x3.hip / conv_f16x3.hip), reached through
// lama_conv2d_* with LAMA_PREC_BF16X3 / LAMA_PREC_F16X3
#define LAMA_CB_DECL(sfx)                                                                                                        \
    int64_t lama_cb_packed_weight_bytes##sfx(int cout, int cin, int kh, int kw, int stride, int transposed);                       \
    int lama_cb_pack_weight##sfx(hipStream_t stream, const float* w, const float* scale, int cout, int cin, int kh, int kw,       \
                                 int stride, int transposed, void* dst);                                                         \
    int lama_cb_conv2d_fwd##sfx(hipStream_t stream, const lama_conv2d_args* a, int Ho, int Wo);
LAMA_CB_DECL(_bf16x3)
LAMA_CB_DECL(_f16x3)
LAMA_CB_DECL(_f16)
#undef LAMA_CB_DECL
// Winograd F(2x2, 3x3) form of the stride-1 3x3 reflect convolution (wino_dev.inc), reached through lama_winograd_* with LAMA_PREC_BF16X3 / F16X3
#define LAMA_WG_DECL(sfx)                                                                                                          \
    int64_t lama_cb_wino_packed_weight_bytes##sfx(int cout, int cin);                                                               \
    int lama_cb_wino_pack_weight##sfx(hipStream_t stream, const float* w, const float* scale, int cout, int cin, void* dst);         \
    int64_t lama_cb_wino_workspace_bytes##sfx(int batch, int cout, int H, int W);                                                    \
    bool lama_cb_wino_shape_ok##sfx(int cout, int cin, int H, int W);                                                                \
    bool lama_cb_wino_preferred##sfx(int batch, int cout, int cin, int H, int W);                                                    \
    int lama_cb_wino_fwd##sfx(hipStream_t stream, const lama_conv2d_args* a, void* workspace, size_t workspace_bytes);                \
    int lama_cb_wino_out_params##sfx(const lama_conv2d_args* a, void* workspace, size_t workspace_bytes, void* out);                  \
    int lama_cb_wino_out_fwd##sfx(hipStream_t stream, const lama_conv2d_args* a, void* workspace, size_t workspace_bytes);
LAMA_WG_DECL(_bf16x3)
LAMA_WG_DECL(_f16x3)
#undef LAMA_WG_DECL
int lama_wino_out_params(const lama_conv2d_args* a, void* workspace, size_t workspace_bytes, void* out);   // conv_mfma.hip -> fft.hip
