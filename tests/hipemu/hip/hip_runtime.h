// hipemu -- a tiny host-side SIMT emulator used ONLY by the CPU test-suite (tests/).
//
// The GPU-less build container cannot run gfx950 code, so tests compile the *unmodified* kernel
// sources in lama_amd/csrc/ a second time with the host compiler and `-I tests/hipemu`, which makes
// `#include <hip/hip_runtime.h>` resolve to this file.  Every thread of a workgroup runs as a
// ucontext fiber; __syncthreads(), wave shuffles and the MFMA builtins are implemented with
// counting barriers across fibers, so index arithmetic, LDS layouts, MFMA fragment maps and
// barrier placement of the real kernels are exercised bit-for-bit in program order.
// It is not a performance tool and never ships: lama_amd/ does not reference it.
//
// MFMA fragment maps follow /opt/skills/guides/cdna_hip_programming.md section 3:
//   32x32x2 f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
//   32x32x16 bf16: A row i=l&31, k=8*(l>>5)+e ; B col j=l&31, k=8*(l>>5)+e ; same D map.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }

#define __global__
#define __device__
#define __host__
#define __shared__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict

namespace hipemu {

struct Block;
struct Fiber {
    ucontext_t ctx;
    std::vector<char> stack;
    dim3 tid;
    int linear = 0;
    bool done = false;
    unsigned xcount = 0;  // wave-exchange counter (parity selects the deposit buffer)
};
struct Wave {
    int nlanes = 0, alive = 0, count = 0, gen = 0;
    // deposit buffers for cross-lane ops: [parity][lane][up to 16 dwords]
    uint32_t buf[2][64][16];
};
struct Block {
    dim3 bid, bdim, gdim;
    int nthreads = 0, alive = 0, count = 0, gen = 0;
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    ucontext_t sched;
    Fiber* cur = nullptr;
    std::function<void()> body;
};
extern Block* g_blk;
void yield();
void block_barrier();
void wave_barrier();
void run_grid(dim3 grid, dim3 block, std::function<void()> body);
inline Fiber* cur() { return g_blk->cur; }
inline Wave& cur_wave() { return g_blk->waves[g_blk->cur->linear / 64]; }
inline int lane() { return g_blk->cur->linear % 64; }

// all lanes deposit `n` dwords, synchronise, and get a pointer to the wave's deposits
inline uint32_t (*exchange(const uint32_t* mine, int n))[16] {
    Fiber* f = cur();
    Wave& w = cur_wave();
    int par = f->xcount & 1;
    f->xcount++;
    for (int i = 0; i < n; ++i) w.buf[par][lane()][i] = mine[i];
    wave_barrier();
    return w.buf[par];
}

}  // namespace hipemu

#define threadIdx (hipemu::g_blk->cur->tid)
#define blockIdx (hipemu::g_blk->bid)
#define blockDim (hipemu::g_blk->bdim)
#define gridDim (hipemu::g_blk->gdim)
#define warpSize 64

// dynamic LDS: kernels declare `extern __shared__ ... char lama_smem[];`
extern __attribute__((aligned(16))) char lama_smem[];

static inline void __syncthreads() { hipemu::block_barrier(); }
// fibers of a block run one at a time on one host thread: a plain read-modify-write is atomic here
static inline unsigned atomicOr(unsigned* p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
static inline void __builtin_amdgcn_s_barrier() { hipemu::block_barrier(); }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
static inline void __builtin_amdgcn_s_setprio(int) {}
// v_cmp_class_f32: bit 0 sNaN, 1 qNaN, 2 -inf, 9 +inf (the only classes the kernels test)
static inline bool __builtin_amdgcn_classf(float v, int mask) {
    if (v != v) return (mask & 0x3) != 0;
    if (v == -__builtin_inff()) return (mask & 0x4) != 0;
    if (v == __builtin_inff()) return (mask & 0x200) != 0;
    return false;
}
static inline void __threadfence() {}
static inline void __threadfence_block() {}

template <class T>
static inline T __shfl(T v, int src, int width = 64) {
    uint32_t mine[4] = {0, 0, 0, 0};
    static_assert(sizeof(T) <= 16, "shfl size");
    memcpy(mine, &v, sizeof(T));
    auto all = hipemu::exchange(mine, (sizeof(T) + 3) / 4);
    int l = hipemu::lane();
    int base = (l / width) * width;
    T r;
    memcpy(&r, all[base + (src % width + width) % width], sizeof(T));
    return r;
}
// v_mov_b32_dpp with a whole-wave shift by one lane (gfx9: wave_shl:1 = 0x130, wave_shr:1 = 0x138), bound_ctrl = 0 for the lane without
// a source.  wave_shr:1: lane i reads lane i - 1;  wave_shl:1: lane i reads lane i + 1.
template <class T>
static inline T hipemu_update_dpp(T oldv, T src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    uint32_t mine[1];
    static_assert(sizeof(T) == 4, "dpp moves dwords");
    memcpy(mine, &src, 4);
    auto all = hipemu::exchange(mine, 1);
    const int l = hipemu::lane();
    // wave_shr:1 / wave_shl:1 (whole wave) and row_shr:1 / row_shl:1 (inside the 16-lane row: the row's first / last lane has no source)
    int from = ctrl == 0x138 ? l - 1 : (ctrl == 0x130 ? l + 1 : -2);
    if (ctrl == 0x111) from = (l & 15) == 0 ? -1 : l - 1;
    if (ctrl == 0x101) from = (l & 15) == 15 ? -1 : l + 1;
    if (from == -2) { std::fprintf(stderr, "hipemu: dpp ctrl 0x%x not emulated\n", ctrl); std::abort(); }
    (void)row_mask; (void)bank_mask;
    if (from < 0 || from > 63) return bound_ctrl ? T(0) : oldv;
    T r;
    memcpy(&r, all[from], 4);
    return r;
}
#define __builtin_amdgcn_update_dpp(o, s, ctrl, rm, bm, bc) hipemu_update_dpp(o, s, ctrl, rm, bm, bc)

template <class T>
static inline T __shfl_xor(T v, int mask, int width = 64) { return __shfl(v, (hipemu::lane() % width) ^ mask, width); }
template <class T>
static inline T __shfl_down(T v, int d, int width = 64) {
    int l = hipemu::lane() % width;
    return __shfl(v, (l + d < width) ? l + d : l, width);
}

// ---- vector types used for MFMA fragments -------------------------------------------------
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef short hipemu_s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));

static inline float hipemu_bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static inline hipemu_f32x16 hipemu_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c) {
    uint32_t mine[2];
    memcpy(&mine[0], &a, 4);
    memcpy(&mine[1], &b, 4);
    auto all = hipemu::exchange(mine, 2);
    int l = hipemu::lane();
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            memcpy(&av, &all[row + 32 * k][0], 4);
            memcpy(&bv, &all[col + 32 * k][1], 4);
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    return c;
}
static inline hipemu_f32x4 hipemu_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c) {
    uint32_t mine[2];
    memcpy(&mine[0], &a, 4);
    memcpy(&mine[1], &b, 4);
    auto all = hipemu::exchange(mine, 2);
    int l = hipemu::lane();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            memcpy(&av, &all[row + 16 * k][0], 4);
            memcpy(&bv, &all[col + 16 * k][1], 4);
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    return c;
}
// 32x32x16 bf16: operands are 8 bf16 per lane (k = 8*(l>>5)+e), fp32 accumulate.
template <class V8>
static inline hipemu_f32x16 hipemu_mfma_f32_32x32x16_bf16(V8 a, V8 b, hipemu_f32x16 c) {
    uint32_t mine[8];
    memcpy(&mine[0], &a, 16);
    memcpy(&mine[4], &b, 16);
    auto all = hipemu::exchange(mine, 8);
    int l = hipemu::lane();
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        double acc = 0.0;  // products of bf16 are exact in fp32; hardware accumulates wider than fp32 chain
        for (int kg = 0; kg < 2; ++kg) {
            const uint16_t* ap = reinterpret_cast<const uint16_t*>(&all[row + 32 * kg][0]);
            const uint16_t* bp = reinterpret_cast<const uint16_t*>(&all[col + 32 * kg][4]);
            for (int e = 0; e < 8; ++e) acc += (double)hipemu_bf16_to_f32(ap[e]) * (double)hipemu_bf16_to_f32(bp[e]);
        }
        c[r] = (float)((double)c[r] + acc);
    }
    return c;
}
// 32x32x16 f16: same fragment map as bf16
template <class V8>
static inline hipemu_f32x16 hipemu_mfma_f32_32x32x16_f16(V8 a, V8 b, hipemu_f32x16 c) {
    uint32_t mine[8];
    memcpy(&mine[0], &a, 16);
    memcpy(&mine[4], &b, 16);
    auto all = hipemu::exchange(mine, 8);
    int l = hipemu::lane();
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        double acc = 0.0;
        for (int kg = 0; kg < 2; ++kg) {
            const _Float16* ap = reinterpret_cast<const _Float16*>(&all[row + 32 * kg][0]);
            const _Float16* bp = reinterpret_cast<const _Float16*>(&all[col + 32 * kg][4]);
            for (int e = 0; e < 8; ++e) acc += (double)(float)ap[e] * (double)(float)bp[e];
        }
        c[r] = (float)((double)c[r] + acc);
    }
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) hipemu_mfma_f32_32x32x16_f16(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipemu_mfma_f32_32x32x2f32(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu_mfma_f32_16x16x4f32(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipemu_mfma_f32_32x32x16_bf16(a, b, c)

// LDS-DMA: every lane copies `size` bytes from its own global address to (wave-uniform LDS base + lane*size)
#define LAMA_LDS_PTR(p) ((void*)(p))
#define LAMA_KEEP_LIVE(x) ((void)(x))
#define LAMA_OPAQUE(x) ((void)(x))
#define LAMA_OPAQUE_S(x) ((void)(x))
static inline void hipemu_global_load_lds(const void* g, void* l, int size) { memcpy((char*)l + hipemu::lane() * size, g, size); }
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) hipemu_global_load_lds((const char*)(g) + (off), l, size)

// raw buffer loads (common.h): base + byte size, lane offset + scalar offset, out-of-range reads return 0
struct lama_buf_t { const char* p; unsigned long long n; };
typedef unsigned hipemu_u32x4 __attribute__((ext_vector_type(4)));
static inline unsigned hipemu_buf_load_b32(lama_buf_t r, unsigned voff, unsigned soff) {
    unsigned long long o = (unsigned long long)voff + soff;
    unsigned v = 0;
    if (o + 4 <= r.n) memcpy(&v, r.p + o, 4);
    return v;
}
static inline hipemu_u32x4 hipemu_buf_load_b128(lama_buf_t r, unsigned voff, unsigned soff) {
    unsigned long long o = (unsigned long long)voff + soff;
    hipemu_u32x4 v = {0, 0, 0, 0};
    // raw buffers are range-checked per DWORD (a 16-byte load that straddles the end returns its in-range dwords, zeros behind them): the
    // any-size Winograd staging reads the plane's last quad of the last channel this way (tests/test_kernels_gpu.py checks the hardware against it)
    for (int d = 0; d < 4; ++d)
        if (o + 4 * d + 4 <= r.n) { unsigned t; memcpy(&t, r.p + o + 4 * d, 4); v[d] = t; }
    return v;
}
static inline void hipemu_buf_store_b32(lama_buf_t r, unsigned v, unsigned voff, unsigned soff) {
    unsigned long long o = (unsigned long long)voff + soff;
    if (o + 4 <= r.n) memcpy(const_cast<char*>(r.p) + o, &v, 4);
}
static inline unsigned short hipemu_buf_load_b16(lama_buf_t r, unsigned voff, unsigned soff) {
    unsigned long long o = (unsigned long long)voff + soff;
    unsigned short v = 0;
    if (o + 2 <= r.n) memcpy(&v, r.p + o, 2);
    return v;
}
static inline void hipemu_buf_store_b16(lama_buf_t r, unsigned short v, unsigned voff, unsigned soff) {
    unsigned long long o = (unsigned long long)voff + soff;
    if (o + 2 <= r.n) memcpy(const_cast<char*>(r.p) + o, &v, 2);
}
#define LAMA_BUF_LOAD_B16(rsrc, voff, soff) hipemu_buf_load_b16(rsrc, voff, soff)
#define LAMA_BUF_STORE_B16(rsrc, val, voff, soff) hipemu_buf_store_b16(rsrc, (unsigned short)(val), voff, soff)
static inline void hipemu_buf_store_b128(lama_buf_t r, hipemu_u32x4 v, unsigned voff, unsigned soff) {
    unsigned long long o = (unsigned long long)voff + soff;
    if (o + 16 <= r.n) memcpy(const_cast<char*>(r.p) + o, &v, 16);
}
#define LAMA_BUF_STORE_B128(rsrc, val, voff, soff) hipemu_buf_store_b128(rsrc, val, voff, soff)
typedef unsigned hipemu_u32x2 __attribute__((ext_vector_type(2)));
static inline hipemu_u32x2 hipemu_buf_load_b64(lama_buf_t r, unsigned voff, unsigned soff) {
    unsigned long long o = (unsigned long long)voff + soff;
    hipemu_u32x2 v = {0, 0};
    if (o + 8 <= r.n) memcpy(&v, r.p + o, 8);
    return v;
}
static inline void hipemu_buf_store_b64(lama_buf_t r, hipemu_u32x2 v, unsigned voff, unsigned soff) {
    unsigned long long o = (unsigned long long)voff + soff;
    if (o + 8 <= r.n) memcpy(const_cast<char*>(r.p) + o, &v, 8);
}
#define LAMA_BUF_LOAD_B64(rsrc, voff, soff) hipemu_buf_load_b64(rsrc, voff, soff)
#define LAMA_PIN_AGPR(x) ((void)0)
#define LAMA_LDS_AS
#define LAMA_BUF_STORE_B64(rsrc, val, voff, soff) hipemu_buf_store_b64(rsrc, val, voff, soff)
#define LAMA_BUF_STORE_B32(rsrc, val, voff, soff) hipemu_buf_store_b32(rsrc, val, voff, soff)
#define LAMA_BUF_RSRC(ptr, bytes) lama_buf_t{(const char*)(ptr), (unsigned long long)(unsigned)(bytes)}
#define LAMA_BUF_LOAD_B32(rsrc, voff, soff) hipemu_buf_load_b32(rsrc, voff, soff)
#define LAMA_BUF_LOAD_B128(rsrc, voff, soff) hipemu_buf_load_b128(rsrc, voff, soff)
// v_perm_b32: byte k of the result = byte sel[k] of the 8 bytes {s0 (4..7), s1 (0..3)}  (selector values 0..7 only: what the kernels use)
static inline unsigned hipemu_perm(unsigned s0, unsigned s1, unsigned sel) {
    const unsigned long long v = ((unsigned long long)s0 << 32) | s1;
    unsigned r = 0;
    for (int k = 0; k < 4; ++k) r |= (unsigned)((v >> (8 * ((sel >> (8 * k)) & 7))) & 0xff) << (8 * k);
    return r;
}
#define __builtin_amdgcn_perm(s0, s1, sel) hipemu_perm(s0, s1, sel)
static inline float hipemu_f16_residual(unsigned packed, float x, int hi) {
    unsigned short b = (unsigned short)(hi ? packed >> 16 : packed & 0xffffu);
    _Float16 h;
    memcpy(&h, &b, 2);
    return fmaf((float)h, -1.0f, x);
}
#define LAMA_F16_RESIDUAL_LO(d, packed, x) ((d) = hipemu_f16_residual(packed, x, 0))
#define LAMA_F16_RESIDUAL_HI(d, packed, x) ((d) = hipemu_f16_residual(packed, x, 1))
static inline unsigned hipemu_f16_split_lo(unsigned packed, float a, float b) {      // v_fma_mixlo_f16 + v_fma_mixhi_f16: the residuals rounded to fp16 (RNE) and packed
    _Float16 l0 = (_Float16)hipemu_f16_residual(packed, a, 0), l1 = (_Float16)hipemu_f16_residual(packed, b, 1);
    unsigned short u0, u1;
    memcpy(&u0, &l0, 2);
    memcpy(&u1, &l1, 2);
    return (unsigned)u0 | ((unsigned)u1 << 16);
}
#define LAMA_F16_SPLIT_LO(lo, packed, a, b) ((lo) = hipemu_f16_split_lo(packed, a, b))
// v_sub_f32_dpp row_shr:1 / v_subrev_f32_dpp row_shl:1 with a zeroed destination (the row's edge lane keeps 0)
#define LAMA_ROW_SHR1_SUB(d, s, b) do { const bool e_ = (hipemu::lane() & 15) == 0; const float t_ = hipemu_update_dpp(0.0f, (float)(s), 0x111, 0xf, 0xf, false); (d) = e_ ? 0.0f : t_ - (b); } while (0)
#define LAMA_ROW_SHL1_RSUB(d, s, b) do { const bool e_ = (hipemu::lane() & 15) == 15; const float t_ = hipemu_update_dpp(0.0f, (float)(s), 0x101, 0xf, 0xf, false); (d) = e_ ? 0.0f : (b) - t_; } while (0)
#define LAMA_WAVE_UNIFORM(x) (x)
#define LAMA_WAVE_SYNC() hipemu::wave_barrier()
#define LAMA_CLOCK() 0ll
#define LAMA_CYCLES() 0ll

// math helpers that exist in HIP device code
static inline void sincospi(double x, double* s, double* c) { *s = sin(M_PI * x); *c = cos(M_PI * x); }
static inline int __mul24(int a, int b) { return a * b; }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline void sincospif(float x, float* s, float* c) { *s = (float)sin(M_PI * (double)x); *c = (float)cos(M_PI * (double)x); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

// ---- launch --------------------------------------------------------------------------------
template <class K, class... Args>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t /*shmem*/, hipStream_t, Args... args) {
    static const bool trace = std::getenv("HIPEMU_TRACE") != nullptr;   // which launch geometry did the host code pick?
    if (trace) std::fprintf(stderr, "[hipemu] launch grid=%u block=%u args=%zu bytes\n", grid.x, block.x, (sizeof(Args) + ... + 0));
    hipemu::run_grid(grid, block, [=]() { kernel(args...); });
}
