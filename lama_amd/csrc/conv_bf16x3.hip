// Fused implicit-GEMM convolution on the gfx950 bf16 matrix cores with a 3-term split that keeps
// fp32-class accuracy (LAMA_PREC_BF16X3):
//
//   x = xh + xl,  w = wh + wl   (xh = bf16(x) RNE, xl = bf16(x - xh); same for w)
//   w*x ~= wh*xh + wh*xl + wl*xh          (dropped wl*xl term: 2^-16 relative)
//
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: three MFMAs per product at 16x the rate of the
// exact v_mfma_f32_32x32x2_f32 path (conv_mfma.hip).  Same fused math as there:
//
//   y[b,o,p] = act( sum_{c,t} W1[o,c,t] x[b,c,tap_t(p)] [+ sum_c W2[o,c] x2[b,c,p]] + bias[o] ) [+ resid[b,o,p]]
//
// GEMM view per image: M = output channels, N = output pixels (the contiguous NCHW axis -> MFMA
// columns, coalesced loads / stores with no layout transform in HBM), K = (channel chunk, tap).
//
// Workgroup = 512 threads = 8 waves (2 per SIMD), output tile BM channels x BN pixels (BN = 128, a
// TH x TW rectangle; BM = 32 uses 8 waves along N, BN = 256).  K is walked in *stages* of TG taps x
// 16*KS channels (TG*KS MFMA k-steps of 16):
//   * weights: pre-split into (hi, lo) bf16 and pre-packed in MFMA A-fragment order by
//     lama_conv2d_pack_weight, so a stage is ONE contiguous image that is copied L2 -> registers -> LDS with
//     fully coalesced 16-byte accesses (no VALU) and read back with linear, conflict-free ds_read_b128;
//     three LDS buffers, the copy runs two to three stages ahead of the MFMAs;
//   * activations: the input *patch* of a channel chunk (tile + halo, reflection / zero padding
//     applied, stride-2 columns parity-split) is loaded fp32 from HBM/L2 one chunk ahead into
//     registers, split into hi/lo bf16 ONCE per element (v_cvt_pk_bf16_f32) and written to LDS as
//     [channel octet][pixel][8 x bf16] planes: a B fragment (8 consecutive channels of one pixel) is
//     one ds_read_b128, consecutive lanes = consecutive pixels = conflict free for every tap; the
//     9x / 49x im2col reuse never leaves the CU; two patch buffers, ONE barrier per stage.
// A second K segment (1x1 over x2) chains into the same accumulators:
//   out_xg = convl2g(x_l) + convg2g.conv2(x1 + fu(x1))  (ffc.py:161,223) + BN shift + ReLU + residual.
#include "common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CB_THREADS 512
#define CB_MAX_TAPS 49

struct CbSeg {
    const float* x;
    long long bstride;
    int C, H, W;
    const char* w;       // packed weights of this sub-convolution: [stage][k-step][32-row fragment F][hi|lo][lane][8 x bf16]
    int mft;             // fragments per k-step = ceil(M / 32); a k-step image is mft * 2048 bytes
    int nchunk, NG;      // channel chunks (16*KS channels each), tap groups (TG taps each) per chunk
    int stride, pad_mode;
    int dy0, dx0;        // patch origin relative to (gy*stride, gx*stride)
    int PH, PW;          // patch rows / cols (real pixels)
    int PWs, PWh;        // LDS row pitch in pixels; parity-split half width (stride 2) or 0
    int npix;            // pixels per LDS octet plane
    int tapoff[CB_MAX_TAPS];  // LDS pixel offset of tap t inside the patch
};

struct CbParams {
    CbSeg s1, s2;
    const float* bias;
    const float* resid;
    long long resid_bstride;
    float* y;
    long long y_bstride;
    int M, MT;
    int Ho, Wo;
    int GH, GW, oy0, ox0, ostep;
    int TWlog, tiles_x, tiles_y, B;
    int act;
    int wbytes, pbytes;  // bytes of ONE weight-stage buffer / ONE patch buffer (hi + lo planes)
};

template <int BM>
struct CbGeom {
    static constexpr int WAVES_M = (BM >= 128) ? 2 : 1;   // BM <= 64: all 8 waves along N (256 pixels), 64 rows -> TM = 2
    static constexpr int WAVES_N = 8 / WAVES_M;
    static constexpr int BN = WAVES_N * 32;
    static constexpr int TM = BM / WAVES_M / 32;  // 32-row fragments per wave (BM = 192 -> 3)
    static constexpr int MF = BM / 32;            // fragments per M tile
};

__device__ __forceinline__ int cb_src_coord(int i, int n, int pad_mode) {
    if (pad_mode == LAMA_PAD_REFLECT) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
        if (i < 0) i = 0;  // only reachable for pixels of a ragged tile that are never stored
        if (i >= n) i = n - 1;
        return i;
    }
    return (i < 0 || i >= n) ? -1 : i;
}

// (hi, lo) bf16 split of two floats, packed as two dwords (element 0 in the low half)
__device__ __forceinline__ void cb_split2(float a, float b, unsigned& hi, unsigned& lo) {
    f32x2 v = {a, b};
    bf16x2 h = __builtin_convertvector(v, bf16x2);
    f32x2 r = v - __builtin_convertvector(h, f32x2);
    bf16x2 l = __builtin_convertvector(r, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

// three ints selected by a (possibly runtime) index without dynamic register indexing (an indexed array would go to scratch)
struct CbInt3 {
    int v0, v1, v2;
    __device__ __forceinline__ void set(int i, int x) {
        if (i == 0) v0 = x;
        else if (i == 1) v1 = x;
        else v2 = x;
    }
    __device__ __forceinline__ int get(int i) const { return i == 1 ? v1 : (i == 2 ? v2 : v0); }
};

template <int V>
struct CbTag {
    static constexpr int value = V;
};

// A / B fragments of one MFMA k-step (16 channels of one tap) for this wave
template <int TM>
struct CbFrag {
    bf16x8 ah[TM], al[TM], bh, bl;
};

// One K segment: accumulate into acc.
//   LDS: three weight-stage buffers (the DMA runs two stages ahead, so the first k-step of stage s+1 can
//   be fetched into registers BEFORE the barrier that ends stage s) and two patch buffers.
//   Registers: two fragment sets -- the ds_read_b128 of k-step k+1 are in flight under the MFMAs of k-step k.
// ABL: timing-only ablations selected by the environment variable LAMA_CB_ABLATE for the profiling tools (results are
// WRONG for ABL != 0): bit 0 = no MFMA, bit 1 = no fragment ds_reads in the loop, bit 2 = no staging, bit 3 = no barriers.
template <int T, int TG, int KS, int BM, int MAXU, int ABL = 0>
__device__ __forceinline__ void cb_segment(const CbSeg& s, int mt, int b, int gy0, int gx0, int TWlog, char* wbuf0, int wbytes,
                                           char* pbuf0, int pbytes, f32x16 (&acc)[CbGeom<BM>::TM]) {
    using G = CbGeom<BM>;
    constexpr int NG = T / TG;                      // stages per channel chunk
    constexpr int NKK = TG * KS;                    // MFMA k-steps per stage
    constexpr bool XPF = (NG >= 2) && (MAXU <= NG - 1);  // next chunk's patch complete (and behind a barrier) before the
                                                     // chunk's last stage -> its first B fragment can be prefetched across
    constexpr int NOCT = 2 * KS;                    // channel octets per chunk
    constexpr int BKC = 16 * KS;                    // channels per chunk
    constexpr int NPIECE = NKK * G::MF * 2;         // 1-KiB fragment images per weight stage
    constexpr int WROUNDS = (NPIECE + 7) / 8;
    constexpr int WST = NPIECE * 1024;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / G::WAVES_N, wn = wave % G::WAVES_N;
    const int khalf = lane >> 5, l31 = lane & 31;
    const int TW = 1 << TWlog;
    const int HW = s.H * s.W;
    const int NPR = s.PH * s.PW;
    const int nunits = NPR * NOCT;
    const int plo = NOCT * s.npix * 16;             // byte offset of the lo planes inside a patch buffer

    // per-thread staging units: (patch pixel, channel octet) -> element offset in the chunk, LDS byte offset
    static_assert(MAXU <= 3, "staging units per thread");
    CbInt3 ubase = {0, 0, 0}, lds_off = {-1, -1, -1}, uq8 = {0, 0, 0}, uvalid = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
        int u = i * CB_THREADS + tid;
        int so = 0, lo = -1, q = 0;
        int ok = 0;
        if (u < nunits) {
            q = u / NPR;
            int pp = u - q * NPR;
            int py = pp / s.PW, px = pp - py * s.PW;
            int iy = cb_src_coord(gy0 * s.stride + s.dy0 + py, s.H, s.pad_mode);
            int ix = cb_src_coord(gx0 * s.stride + s.dx0 + px, s.W, s.pad_mode);
            if (iy >= 0 && ix >= 0) { so = iy * s.W + ix; ok = 1; }
            int lidx = py * s.PWs + (s.PWh ? (px & 1) * s.PWh + (px >> 1) : px);
            lo = (q * s.npix + lidx) * 16;
        }
        ubase.set(i, q * 8 * HW + so);   // always an in-bounds element of the chunk (zero padding is applied at write time)
        lds_off.set(i, lo);
        uq8.set(i, q * 8);
        uvalid.set(i, ok);
    }
    // B-fragment base: this lane's pixel inside the patch, octet plane khalf
    int boff;
    {
        int n = wn * 32 + l31;
        int ty = n >> TWlog, tx = n & (TW - 1);
        int pixoff = ty * s.stride * s.PWs + (s.PWh ? tx : tx * s.stride);
        boff = (khalf * s.npix + pixoff) * 16;
    }
    const int aoff = (wm * G::TM * 2) * 1024 + lane * 16;  // A fragments of this wave inside a k-step image

    const float* xb = s.x + (long long)b * s.bstride;
    const int S = s.nchunk * NG;

    // Patch staging keeps ONE unit (8 channels of one pixel) per thread in flight: unit u of chunk ch+1 is written to LDS in
    // stage u of chunk ch and the following unit is requested right after, so a kernel with a large halo (MAXU = 3: stride 2,
    // 7x7) needs no more staging registers than the bottleneck 3x3 (MAXU = 1).
    constexpr int UF = (NG >= MAXU) ? 1 : MAXU;   // units in flight: all of them when a chunk has fewer stages than units
    float preg[UF][8];
    auto usel = [&](const CbInt3& arr, int u) { return MAXU == 1 ? arr.v0 : arr.get(u); };
    auto load_unit = [&](int ch, int u, int slot = 0) {
        const float* xc = xb + (long long)ch * BKC * HW;
        const int crem = s.C - ch * BKC;
        const int ub = usel(ubase, u);
        if (crem >= BKC) {  // (uniform) every channel of the chunk exists: unconditional loads, no per-element branches
#pragma unroll
            for (int e = 0; e < 8; ++e) preg[slot][e] = xc[(unsigned)(ub + e * HW)];   // SGPR base + 32-bit lane offset
        } else {            // channel tail: clamp the address, zero the value (the packed weights are zero there too)
            const int q8 = usel(uq8, u);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int c = q8 + e;
                int back = c < crem ? 0 : (c - (crem - 1)) * HW;
                float v = xc[(unsigned)(ub + e * HW - back)];
                preg[slot][e] = c < crem ? v : 0.0f;
            }
        }
    };
    auto write_unit = [&](char* pb, int u, int slot = 0) {
        const int lo = usel(lds_off, u);
        if (lo >= 0) {
            unsigned hh[4], ll[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) cb_split2(preg[slot][2 * e], preg[slot][2 * e + 1], hh[e], ll[e]);
            u32x4 h = {hh[0], hh[1], hh[2], hh[3]}, l = {ll[0], ll[1], ll[2], ll[3]};
            if (!usel(uvalid, u)) { h = u32x4{0, 0, 0, 0}; l = h; }   // zero padding
            *reinterpret_cast<u32x4*>(pb + lo) = h;
            *reinterpret_cast<u32x4*>(pb + plo + lo) = l;
        }
    };
    // stage g of chunk ch: write the unit(s) of chunk ch+1 that belong to this stage, keep the next one in flight
    // With a single unit and >= 3 stages per chunk the two waves of a SIMD stage their unit in DIFFERENT stages (waves 0-3 in
    // stage 0, waves 4-7 in stage 1), so the split / address VALU work of one runs under the MFMAs of the other.
    const int gskew = (MAXU == 1 && NG >= 3) ? (wave >> 2) : 0;
    auto stage_patch = [&](int ch, int g, char* pbn) {
        if (ch + 1 >= s.nchunk) return;
        if constexpr (NG >= MAXU) {
            if (MAXU == 1) {
                if (g == gskew) {
                    write_unit(pbn, 0);
                    if (ch + 2 < s.nchunk) load_unit(ch + 2, 0);
                }
            } else if (g < MAXU) {
                write_unit(pbn, g);
                if (g + 1 < MAXU) load_unit(ch + 1, g + 1);
                else if (ch + 2 < s.nchunk) load_unit(ch + 2, 0);
            }
        } else {   // fewer stages than units: every unit has its own register slot and all are written / reloaded in stage 0
            if (g == 0) {
#pragma unroll
                for (int u = 0; u < MAXU; ++u) write_unit(pbn, u, u);
                if (ch + 2 < s.nchunk) {
#pragma unroll
                    for (int u = 0; u < MAXU; ++u) load_unit(ch + 2, u, u);
                }
            }
        }
    };
    // weight stage st: global (L2-resident packed image) -> registers; written to LDS a stage later.  Plain loads + ds_write
    // instead of global_load_lds DMA: with a DMA in flight hipcc degrades every s_waitcnt to lgkmcnt(0)/vmcnt(0), which
    // serialises the ds_read prefetch below; 16 B per lane per round keeps the loads fully coalesced.
    u32x4 wreg[WROUNDS];
    constexpr int WITEMS = NPIECE * 64;   // 16-byte items per stage image
    // The packed layout does not depend on the M-tile height: this workgroup copies fragments [mt*MF, mt*MF + MF) of each
    // k-step (one contiguous MF*2 KiB run per k-step; fragments past the last one are clamped, their rows are never stored).
    auto load_w = [&](int st) {
        const u32x4* g = reinterpret_cast<const u32x4*>(s.w) + (long long)st * NKK * s.mft * 128;
#pragma unroll
        for (int r = 0; r < WROUNDS; ++r) {
            int idx = r * CB_THREADS + tid;
            if ((r + 1) * CB_THREADS > WITEMS) idx = idx < WITEMS ? idx : WITEMS - 1;   // partial last round: clamp, never branch
            const int kk = idx / (G::MF * 128), rem = idx - kk * (G::MF * 128);
            int F = mt * G::MF + (rem >> 7);
            F = F < s.mft ? F : s.mft - 1;
            wreg[r] = g[(kk * s.mft + F) * 128 + (rem & 127)];
        }
    };
    auto write_w = [&](char* wb) {
        u32x4* d = reinterpret_cast<u32x4*>(wb);
#pragma unroll
        for (int r = 0; r < WROUNDS; ++r) {
            int idx = r * CB_THREADS + tid;
            if ((r + 1) * CB_THREADS <= WITEMS || idx < WITEMS) d[idx] = wreg[r];
        }
    };
    using Frag = CbFrag<G::TM>;
    auto read_a = [&](Frag& f, const char* wb, int kk) {
#pragma unroll
        for (int i = 0; i < G::TM; ++i) {
            const char* ap = wb + (kk * G::MF * 2 + i * 2) * 1024 + aoff;
            f.ah[i] = *reinterpret_cast<const bf16x8*>(ap);
            f.al[i] = *reinterpret_cast<const bf16x8*>(ap + 1024);
        }
    };
    // LDS offset of tap (g*TG + tgi) = tgoff[tgi] + g * grow: one kernel row (or the whole transposed-class tap list) per stage,
    // so the stage loop needs no scalar loads (an SMEM load in flight would force lgkmcnt(0) on every ds_read wait)
    int tgoff[TG];
#pragma unroll
    for (int i = 0; i < TG; ++i) tgoff[i] = s.tapoff[i] * 16;
    const int grow = NG > 1 ? (s.tapoff[TG] - s.tapoff[0]) * 16 : 0;
    auto read_b = [&](Frag& f, const char* pb, int g, int tgi, int ks) {
        const char* bp = pb + (ks * 2 * s.npix) * 16 + tgoff[tgi] + g * grow + boff;
        f.bh = *reinterpret_cast<const bf16x8*>(bp);
        f.bl = *reinterpret_cast<const bf16x8*>(bp + plo);
    };

    // prologue: weight stages 0 and 1 and the chunk-0 patch in LDS; stage 2 / chunk 1 in flight to registers
    load_w(0);
    load_unit(0, 0);
    write_w(wbuf0);
    if (S > 1) load_w(1);               // in flight while the first patch unit is converted
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
        write_unit(pbuf0, u);
        if (u + 1 < MAXU) load_unit(0, u + 1);
    }
    if (s.nchunk > 1) {
        if constexpr (UF == 1) load_unit(1, 0);
        else {
#pragma unroll
            for (int u = 0; u < MAXU; ++u) load_unit(1, u, u);
        }
    }
    if (S > 1) write_w(wbuf0 + wbytes);
    if (S > 2) load_w(2);
    __syncthreads();
    Frag fr[2];   // fragment sets, indexed with compile-time parity only (two k-steps in flight, no register copies)
    read_a(fr[0], wbuf0, 0);
    read_b(fr[0], pbuf0, 0, 0, 0);

    int st = 0, wi = 0;  // wi = st % 3
    // one channel chunk = NG stages = NG*NKK k-steps; PAR = parity of the fragment set holding its first k-step
    // FLAT: stages of a chunk fully unrolled and the fragment-set parity carried at compile time (the bottleneck kernels);
    // otherwise (7x7 kernels, big staging footprints) a rolled stage loop that realigns the parity with one register copy
    // per stage, which keeps code size and register pressure down.
    constexpr bool FLAT = (NG <= 3) && (MAXU == 1);
    auto stage = [&](int ch, int g, auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        const char* pb = pbuf0 + (ch & 1) * pbytes;
        char* pbn = pbuf0 + ((ch + 1) & 1) * pbytes;
        const char* wb = wbuf0 + wi * wbytes;
        const int wi1 = wi == 2 ? 0 : wi + 1, wi2 = wi1 == 2 ? 0 : wi1 + 1;
        const char* wbn = wbuf0 + wi1 * wbytes;
        if constexpr (!(ABL & 4)) stage_patch(ch, g, pbn);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            Frag& cur = fr[(PAR + kk) & 1];
            Frag& nxt = fr[(PAR + kk + 1) & 1];
            if constexpr (ABL & 2) {
                if (kk + 1 == NKK && !(ABL & 4) && st + 2 < S) write_w(wbuf0 + wi2 * wbytes);
                nxt = cur;
            } else if (kk + 1 < NKK) {
                const int tgi = (kk + 1) / KS, ks = (kk + 1) % KS;
                read_a(nxt, wb, kk + 1);
                read_b(nxt, pb, g, tgi, ks);
            } else {
                if (st + 1 < S) {
                    // first k-step of the next stage: its weights were written before the previous barrier
                    read_a(nxt, wbn, 0);
                    if (g + 1 < NG) read_b(nxt, pb, g + 1, 0, 0);
                    else if (XPF) read_b(nxt, pbn, 0, 0, 0);   // next chunk's patch is complete and visible
                }
                // stage st+2 (in registers since the end of stage st-1) -> the buffer stage st-1 was read from
                if (!(ABL & 4) && st + 2 < S) write_w(wbuf0 + wi2 * wbytes);
            }
            // pin the order: the ds_reads of the NEXT k-step are issued before this k-step's MFMAs and are only
            // waited for after them (hipcc otherwise sinks the reads next to their use and exposes the LDS latency)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ABL & 1) {
#pragma unroll
                for (int i = 0; i < G::TM; ++i) { LAMA_KEEP_LIVE(cur.ah[i]); LAMA_KEEP_LIVE(cur.al[i]); LAMA_KEEP_LIVE(cur.bh); LAMA_KEEP_LIVE(cur.bl); }
            } else {
#pragma unroll
                for (int i = 0; i < G::TM; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.ah[i], cur.bh, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.ah[i], cur.bl, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.al[i], cur.bh, acc[i], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(ABL & 4) && st + 3 < S) load_w(st + 3);   // registers are free again: fetch three stages ahead
        if constexpr (!(ABL & 8)) __syncthreads();
        if (!XPF && g == NG - 1 && st + 1 < S) read_b(fr[(PAR + NKK) & 1], pbn, 0, 0, 0);   // chunk boundary: new patch visible only now
        wi = wi1;
        ++st;
    };
    if constexpr (FLAT) {
        // one chunk = NG*NKK k-steps; an odd count flips the parity for the next chunk -> unroll two chunks
        auto chunk = [&](int ch, auto par_tag) {
            constexpr int PAR = decltype(par_tag)::value;
            stage(ch, 0, CbTag<PAR & 1>{});
            if constexpr (NG > 1) stage(ch, 1, CbTag<(PAR + NKK) & 1>{});
            if constexpr (NG > 2) stage(ch, 2, CbTag<(PAR + 2 * NKK) & 1>{});
        };
        if constexpr ((NG * NKK) % 2 == 0) {
            for (int ch = 0; ch < s.nchunk; ++ch) chunk(ch, CbTag<0>{});
        } else {
            for (int ch = 0; ch < s.nchunk; ch += 2) {
                chunk(ch, CbTag<0>{});
                if (ch + 1 < s.nchunk) chunk(ch + 1, CbTag<1>{});
            }
        }
    } else {
        for (int ch = 0; ch < s.nchunk; ++ch) {
#pragma unroll 1
            for (int g = 0; g < NG; ++g) {
                stage(ch, g, CbTag<0>{});
                if constexpr (NKK % 2 == 1) fr[0] = fr[1];   // realign: the next stage's first k-step was fetched into set 1
            }
        }
    }
}

template <int T1, int TG1, int KS1, int T2, int TG2, int KS2, int BM, int MAXU, int ABL = 0>
__global__ __launch_bounds__(CB_THREADS) void conv_bf16x3_kernel(CbParams p) {
    using G = CbGeom<BM>;
    char* wbuf0 = lama_smem;
    char* pbuf0 = lama_smem + 3 * p.wbytes;

    const int L = lama_xcd_remap(blockIdx.x, gridDim.x);
    const int mt = L % p.MT;
    int tile = L / p.MT;
    const int tix = tile % p.tiles_x;
    tile /= p.tiles_x;
    const int tiy = tile % p.tiles_y;
    const int b = tile / p.tiles_y;
    const int TW = 1 << p.TWlog, TH = G::BN >> p.TWlog;
    const int gy0 = tiy * TH, gx0 = tix * TW;

    f32x16 acc[G::TM];
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    cb_segment<T1, TG1, KS1, BM, MAXU, ABL>(p.s1, mt, b, gy0, gx0, p.TWlog, wbuf0, p.wbytes, pbuf0, p.pbytes, acc);
    if constexpr (TG2 > 0) cb_segment<T2, TG2, KS2, BM, MAXU>(p.s2, mt, b, gy0, gx0, p.TWlog, wbuf0, p.wbytes, pbuf0, p.pbytes, acc);

    // epilogue: D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / G::WAVES_N, wn = wave % G::WAVES_N;
    const long long plane = (long long)p.Ho * p.Wo;
    const int n = wn * 32 + (lane & 31);
    const int gy = gy0 + (n >> p.TWlog), gx = gx0 + (n & (TW - 1));
    const bool pv = gy < p.GH && gx < p.GW;
    const long long pix = (long long)(gy * p.ostep + p.oy0) * p.Wo + (gx * p.ostep + p.ox0);
#pragma unroll
    for (int i = 0; i < G::TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int m = mt * BM + (wm * G::TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (pv && m < p.M) {
                float v = acc[i][r];
                if (p.bias) v += p.bias[m];
                if (p.act == LAMA_ACT_RELU) v = fmaxf(v, 0.0f);
                else if (p.act == LAMA_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
                else if (p.act == LAMA_ACT_TANH) v = tanhf(v);
                long long o = m * plane + pix;
                if (p.resid) v += p.resid[(long long)b * p.resid_bstride + o];
                p.y[(long long)b * p.y_bstride + o] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight packing: reference layout fp32 -> [stage][k-step kk = tgi*KS + ks][32-row fragment F][hi|lo][lane][8 x bf16]
//   (hi, lo) bf16 A fragments; independent of the M-tile height the launch later picks:
//   lane l of fragment F holds output channel m = F*32 + (l&31),
//   input channels c = ch*16*KS + ks*16 + 8*(l>>5) + e  (e = 0..7), tap t = g*TG + tgi
// ------------------------------------------------------------------------------------------------
struct CbPackParams {
    const float* w;
    const float* scale;
    char* dst;
    int M, C, MFT, TG, KS, NG, nchunk;
    int kh, kw, transposed;
    int tap_ky[CB_MAX_TAPS], tap_kx[CB_MAX_TAPS];
};

__global__ void conv_bf16x3_pack_kernel(CbPackParams p) {
    const int NKK = p.TG * p.KS;
    const long long total = (long long)p.nchunk * p.NG * NKK * p.MFT * 64;  // (stage, kk, F, lane) items; each writes hi and lo 16 B
    for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
        int lane = (int)(it & 63);
        long long r = it >> 6;
        int F = (int)(r % p.MFT);
        r /= p.MFT;
        int kk = (int)(r % NKK);
        r /= NKK;
        int g = (int)(r % p.NG);
        int ch = (int)(r / p.NG);
        int tgi = kk / p.KS, ks = kk - tgi * p.KS;
        int t = g * p.TG + tgi;
        int m = F * 32 + (lane & 31);
        int c0 = ch * 16 * p.KS + ks * 16 + 8 * (lane >> 5);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int c = c0 + e;
            float x = 0.0f;
            if (m < p.M && c < p.C) {
                int ky = p.tap_ky[t], kx = p.tap_kx[t];
                long long src = p.transposed ? (((long long)c * p.M + m) * p.kh + ky) * p.kw + kx
                                             : (((long long)m * p.C + c) * p.kh + ky) * p.kw + kx;
                x = p.w[src];
                if (p.scale) x *= p.scale[m];
            }
            v[e] = x;
        }
        unsigned hh[4], ll[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) cb_split2(v[2 * e], v[2 * e + 1], hh[e], ll[e]);
        long long stage = (long long)ch * p.NG + g;
        char* base = p.dst + (((stage * NKK + kk) * p.MFT + F) * 2) * 1024 + lane * 16;
        *reinterpret_cast<u32x4*>(base) = u32x4{hh[0], hh[1], hh[2], hh[3]};
        *reinterpret_cast<u32x4*>(base + 1024) = u32x4{ll[0], ll[1], ll[2], ll[3]};
    }
}

// ------------------------------------------------------------------------------------------------
// host side: plan, pack, launch
// ------------------------------------------------------------------------------------------------
namespace {

struct CbPlan {
    int mft;   // 32-row fragments = ceil(M / 32)
    int nseg;  // sub-convolutions (1, or 4 output-parity classes for ConvTranspose2d)
    int T[4], TG[4], KS[4];
    int ky[4][CB_MAX_TAPS], kx[4][CB_MAX_TAPS];  // weight tap
    int dy[4][CB_MAX_TAPS], dx[4][CB_MAX_TAPS];  // input offset of the tap
    int oy0[4], ox0[4];
    int nchunk[4];
    long long cls_bytes[4], woff[4];
    long long total_bytes;
};

// M-tile height, chosen per launch (the packed weights do not depend on it).  One workgroup per CU is resident, so the cost
// of a launch is ~ rounds * (BM + fixed per-workgroup overhead): 192-row tiles win when they save a round or the 25 % padding
// of a second 128-row tile (M = 192: 1 x 192 instead of 2 x 128; M = 384 at 256 pixel tiles: 2 rounds instead of 3).
int cb_pick_bm(int M, int T, int stride, int batch, int GH, int GW) {
    if (M <= 32) return 32;
    if (M <= 64 || T == 49) return 64;       // 7x7 kernels keep their 7-tap weight stages small
    const long long tiles = (long long)batch * lama_ceil_div64((long long)GH * GW, 128);
    long long best = 0;
    int bm = 128;
    for (int cand : {128, 192}) {
        long long rounds = lama_ceil_div64(tiles * lama_ceil_div(M, cand), 256);
        long long cost = rounds * (cand + 24);
        if (best == 0 || cost < best) { best = cost; bm = cand; }
    }
    return bm;
}

bool cb_stage_shape(int T, int* TG, int* KS) {
    switch (T) {
        case 1: *TG = 1; *KS = 2; return true;
        case 2: *TG = 2; *KS = 1; return true;
        case 4: *TG = 4; *KS = 1; return true;
        case 9: *TG = 3; *KS = 1; return true;
        case 49: *TG = 7; *KS = 1; return true;
    }
    return false;
}

bool cb_make_plan(int cout, int cin, int kh, int kw, int stride, int pad, int transposed, CbPlan* pl) {
    pl->mft = lama_ceil_div(cout, 32);
    if (transposed) {
        if (kh != 3 || kw != 3 || stride != 2 || pad != 1) return false;
        pl->nseg = 4;
        for (int cls = 0; cls < 4; ++cls) {
            int py = cls >> 1, px = cls & 1;
            int kys[2], dys[2], nky, kxs[2], dxs[2], nkx;
            if (py == 0) { nky = 1; kys[0] = 1; dys[0] = 0; } else { nky = 2; kys[0] = 0; dys[0] = 1; kys[1] = 2; dys[1] = 0; }
            if (px == 0) { nkx = 1; kxs[0] = 1; dxs[0] = 0; } else { nkx = 2; kxs[0] = 0; dxs[0] = 1; kxs[1] = 2; dxs[1] = 0; }
            int t = 0;
            for (int a = 0; a < nky; ++a)
                for (int c = 0; c < nkx; ++c) {
                    pl->ky[cls][t] = kys[a]; pl->kx[cls][t] = kxs[c];
                    pl->dy[cls][t] = dys[a]; pl->dx[cls][t] = dxs[c];
                    ++t;
                }
            pl->T[cls] = t;
            pl->oy0[cls] = py; pl->ox0[cls] = px;
        }
    } else {
        if (!((kh == 1 && kw == 1) || (kh == 3 && kw == 3) || (kh == 7 && kw == 7))) return false;
        if (stride != 1 && stride != 2) return false;
        pl->nseg = 1;
        int t = 0;
        for (int a = 0; a < kh; ++a)
            for (int c = 0; c < kw; ++c) {
                pl->ky[0][t] = a; pl->kx[0][t] = c;
                pl->dy[0][t] = a - pad; pl->dx[0][t] = c - pad;
                ++t;
            }
        pl->T[0] = t;
        pl->oy0[0] = pl->ox0[0] = 0;
    }
    long long off = 0;
    for (int cls = 0; cls < pl->nseg; ++cls) {
        if (!cb_stage_shape(pl->T[cls], &pl->TG[cls], &pl->KS[cls])) return false;
        pl->nchunk[cls] = lama_ceil_div(cin, 16 * pl->KS[cls]);
        pl->cls_bytes[cls] = (long long)pl->nchunk[cls] * pl->T[cls] * pl->KS[cls] * pl->mft * 2048;
        pl->woff[cls] = off;
        off += pl->cls_bytes[cls];
    }
    pl->total_bytes = off;
    return true;
}

// fill one segment for a tile of TH x TW output pixels; returns the staging units it needs
int cb_fill_seg(CbSeg* s, const lama_tensor& x, const char* w, const CbPlan& pl, int cls, int stride, int pad_mode, int TH, int TW,
                bool flat) {
    const int T = pl.T[cls];
    s->x = (const float*)x.ptr;
    s->bstride = x.batch_stride;
    s->C = x.C;
    s->H = flat ? 1 : x.H;
    s->W = flat ? x.H * x.W : x.W;
    s->w = w;
    s->mft = pl.mft;
    s->nchunk = pl.nchunk[cls];
    s->NG = T / pl.TG[cls];
    s->stride = stride;
    s->pad_mode = pad_mode;
    int dymin = 1 << 30, dymax = -(1 << 30), dxmin = 1 << 30, dxmax = -(1 << 30);
    for (int t = 0; t < T; ++t) {
        dymin = pl.dy[cls][t] < dymin ? pl.dy[cls][t] : dymin;
        dymax = pl.dy[cls][t] > dymax ? pl.dy[cls][t] : dymax;
        dxmin = pl.dx[cls][t] < dxmin ? pl.dx[cls][t] : dxmin;
        dxmax = pl.dx[cls][t] > dxmax ? pl.dx[cls][t] : dxmax;
    }
    s->dy0 = dymin;
    s->dx0 = dxmin;
    s->PH = (TH - 1) * stride + (dymax - dymin) + 1;
    s->PW = (TW - 1) * stride + (dxmax - dxmin) + 1;
    if (stride == 2) {
        s->PWh = (s->PW + 1) / 2;
        s->PWs = 2 * s->PWh;
    } else {
        s->PWh = 0;
        s->PWs = s->PW;
    }
    s->npix = s->PH * s->PWs;
    for (int t = 0; t < T; ++t) {
        int ry = pl.dy[cls][t] - dymin, rx = pl.dx[cls][t] - dxmin;
        s->tapoff[t] = ry * s->PWs + (s->PWh ? (rx & 1) * s->PWh + (rx >> 1) : rx);
    }
    return s->PH * s->PW * 2 * pl.KS[cls];
}

template <int T1, int TG1, int KS1, int T2, int TG2, int KS2, int BM>
int cb_launch_u(hipStream_t st, const CbParams& p, int maxu, int grid, size_t shmem) {
    if (maxu <= 1) hipLaunchKernelGGL((conv_bf16x3_kernel<T1, TG1, KS1, T2, TG2, KS2, BM, 1>), dim3(grid), dim3(CB_THREADS), shmem, st, p);
    else if (maxu <= 3) hipLaunchKernelGGL((conv_bf16x3_kernel<T1, TG1, KS1, T2, TG2, KS2, BM, 3>), dim3(grid), dim3(CB_THREADS), shmem, st, p);
    else return LAMA_ERR_UNSUPPORTED;
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

template <int T1, int TG1, int KS1, int T2, int TG2, int KS2>
int cb_launch_bm(hipStream_t st, const CbParams& p, int BM, int maxu, int grid, size_t shmem) {
    switch (BM) {
        case 192: return cb_launch_u<T1, TG1, KS1, T2, TG2, KS2, 192>(st, p, maxu, grid, shmem);
        case 128: return cb_launch_u<T1, TG1, KS1, T2, TG2, KS2, 128>(st, p, maxu, grid, shmem);
        case 64: return cb_launch_u<T1, TG1, KS1, T2, TG2, KS2, 64>(st, p, maxu, grid, shmem);
        case 32: return cb_launch_u<T1, TG1, KS1, T2, TG2, KS2, 32>(st, p, maxu, grid, shmem);
    }
    return LAMA_ERR_UNSUPPORTED;
}

int cb_launch(hipStream_t st, const CbParams& p, int T1, int T2, int BM, int maxu) {
    const size_t shmem = 3 * (size_t)p.wbytes + 2 * (size_t)p.pbytes;
    if (shmem > 160 * 1024) return LAMA_ERR_UNSUPPORTED;
    const int grid = p.B * p.tiles_x * p.tiles_y * p.MT;
    if (grid <= 0) return LAMA_OK;
    if (T1 == 9 && T2 == 0 && BM == 128 && maxu <= 1) {   // profiling tools only: timing ablations of the bottleneck 3x3 kernel
        const char* e = getenv("LAMA_CB_ABLATE");
        const int abl = e ? atoi(e) : 0;
#define CB_ABL(v) \
    if (abl == v) { hipLaunchKernelGGL((conv_bf16x3_kernel<9, 3, 1, 0, 0, 0, 128, 1, v>), dim3(grid), dim3(CB_THREADS), shmem, st, p); LAMA_CHECK_LAUNCH(); return LAMA_OK; }
        CB_ABL(1) CB_ABL(2) CB_ABL(4) CB_ABL(6) CB_ABL(7) CB_ABL(8) CB_ABL(14) CB_ABL(15)
#undef CB_ABL
    }
#define CB_CASE(t1, g1, k1, t2, g2, k2) \
    if (T1 == t1 && T2 == t2) return cb_launch_bm<t1, g1, k1, t2, g2, k2>(st, p, BM, maxu, grid, shmem);
    CB_CASE(9, 3, 1, 0, 0, 0)
    CB_CASE(9, 3, 1, 1, 1, 2)
    CB_CASE(1, 1, 2, 0, 0, 0)
    CB_CASE(49, 7, 1, 0, 0, 0)
    CB_CASE(2, 2, 1, 0, 0, 0)
    CB_CASE(4, 4, 1, 0, 0, 0)
#undef CB_CASE
    return LAMA_ERR_UNSUPPORTED;
}

}  // namespace

int64_t lama_cb_packed_weight_bytes(int cout, int cin, int kh, int kw, int stride, int transposed) {
    CbPlan pl;
    if (!cb_make_plan(cout, cin, kh, kw, stride, transposed ? 1 : kh / 2, transposed, &pl)) return LAMA_ERR_UNSUPPORTED;
    return pl.total_bytes;
}

int lama_cb_pack_weight(hipStream_t stream, const float* w, const float* scale, int cout, int cin, int kh, int kw, int stride,
                        int transposed, void* dst) {
    CbPlan pl;
    if (!cb_make_plan(cout, cin, kh, kw, stride, transposed ? 1 : kh / 2, transposed, &pl)) return LAMA_ERR_UNSUPPORTED;
    for (int cls = 0; cls < pl.nseg; ++cls) {
        CbPackParams pp;
        memset(&pp, 0, sizeof(pp));
        pp.w = w;
        pp.scale = scale;
        pp.dst = (char*)dst + pl.woff[cls];
        pp.M = cout;
        pp.C = cin;
        pp.MFT = pl.mft;
        pp.TG = pl.TG[cls];
        pp.KS = pl.KS[cls];
        pp.NG = pl.T[cls] / pl.TG[cls];
        pp.nchunk = pl.nchunk[cls];
        pp.kh = kh;
        pp.kw = kw;
        pp.transposed = transposed;
        for (int t = 0; t < pl.T[cls]; ++t) { pp.tap_ky[t] = pl.ky[cls][t]; pp.tap_kx[t] = pl.kx[cls][t]; }
        long long total = pl.cls_bytes[cls] / 32;  // items of 2 x 16 B
        int grid = (int)((total + 255) / 256);
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(conv_bf16x3_pack_kernel, dim3(grid), dim3(256), 0, stream, pp);
        LAMA_CHECK_LAUNCH();
    }
    return LAMA_OK;
}

// arguments were validated by lama_conv2d_fwd (conv_mfma.hip)
int lama_cb_conv2d_fwd(hipStream_t stream, const lama_conv2d_args* a, int Ho, int Wo) {
    const int cout = a->y.C, cin = a->x.C;
    CbPlan pl;
    if (!cb_make_plan(cout, cin, a->kh, a->kw, a->stride, a->pad, a->transposed, &pl)) return LAMA_ERR_UNSUPPORTED;
    const bool has2 = a->x2.ptr != nullptr;
    CbPlan pl2;
    if (has2 && !cb_make_plan(cout, a->x2.C, 1, 1, 1, 0, 0, &pl2)) return LAMA_ERR_UNSUPPORTED;
    int BM = cb_pick_bm(cout, pl.T[0], a->transposed ? 1 : a->stride, a->batch, a->transposed ? a->x.H : Ho, a->transposed ? a->x.W : Wo);
    if (BM < 128 && !a->transposed && a->stride == 2) BM = 128;   // a 256-pixel stride-2 tile does not leave room for two patch buffers
    const int BN = BM >= 128 ? 128 : 256;
    if ((long long)a->x.C * a->x.H * a->x.W >= (1ll << 31) || (has2 && (long long)a->x2.C * a->x2.H * a->x2.W >= (1ll << 31)))
        return LAMA_ERR_UNSUPPORTED;   // 32-bit element offsets inside one image

    for (int cls = 0; cls < pl.nseg; ++cls) {
        CbParams p;
        memset(&p, 0, sizeof(p));
        const bool flat = (pl.T[cls] == 1 && !a->transposed && a->stride == 1 && a->pad == 0 && !has2);
        p.bias = a->bias;
        p.resid = (const float*)a->resid.ptr;
        p.resid_bstride = a->resid.batch_stride;
        p.y = (float*)a->y.ptr;
        p.y_bstride = a->y.batch_stride;
        p.M = cout;
        p.MT = lama_ceil_div(cout, BM);
        p.B = a->batch;
        p.act = a->act;
        if (flat) {
            p.Ho = 1; p.Wo = Ho * Wo; p.GH = 1; p.GW = Ho * Wo; p.ostep = 1;
        } else if (a->transposed) {
            p.Ho = Ho; p.Wo = Wo; p.GH = a->x.H; p.GW = a->x.W; p.ostep = 2;
            p.oy0 = pl.oy0[cls]; p.ox0 = pl.ox0[cls];
        } else {
            p.Ho = Ho; p.Wo = Wo; p.GH = Ho; p.GW = Wo; p.ostep = 1;
        }
        const int stride = a->transposed ? 1 : a->stride;
        const int pad_mode = a->transposed ? LAMA_PAD_ZERO : a->pad_mode;
        // tile: TW = 32 output pixels of one row per wave (conflict-free B fragments), narrower only for narrow images
        int twlog = flat ? lama_ilog2(BN) : 5;
        while (twlog > 3 && (1 << (twlog - 1)) >= p.GW) --twlog;
        const int TW = 1 << twlog, TH = BN >> twlog;
        p.TWlog = twlog;
        p.tiles_x = lama_ceil_div(p.GW, TW);
        p.tiles_y = lama_ceil_div(p.GH, TH);
        int units = cb_fill_seg(&p.s1, a->x, (const char*)a->w_packed + pl.woff[cls], pl, cls, stride, pad_mode, TH, TW, flat);
        long long wbytes = (long long)pl.TG[cls] * pl.KS[cls] * (BM / 32) * 2048;
        long long pbytes = (long long)2 * 2 * pl.KS[cls] * p.s1.npix * 16;
        int TG2 = 0, KS2 = 0;
        if (has2) {
            TG2 = pl2.TG[0]; KS2 = pl2.KS[0];
            int u2 = cb_fill_seg(&p.s2, a->x2, (const char*)a->w2_packed, pl2, 0, 1, LAMA_PAD_ZERO, TH, TW, false);
            units = u2 > units ? u2 : units;
            long long w2 = (long long)TG2 * KS2 * (BM / 32) * 2048, p2 = (long long)2 * 2 * KS2 * p.s2.npix * 16;
            wbytes = w2 > wbytes ? w2 : wbytes;
            pbytes = p2 > pbytes ? p2 : pbytes;
        }
        p.wbytes = (int)wbytes;
        p.pbytes = (int)pbytes;
        int rc = cb_launch(stream, p, pl.T[cls], has2 ? 1 : 0, BM, lama_ceil_div(units, CB_THREADS));
        if (rc != LAMA_OK) return rc;
    }
    return LAMA_OK;
}
