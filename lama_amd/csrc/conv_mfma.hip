// Fused implicit-GEMM convolution on the gfx950 matrix cores.
//
//   y[b,o,p] = act( sum_{c,t} W1[o,c,t] * x[b,c,tap_t(p)]  [+ sum_c W2[o,c] * x2[b,c,p]]  + bias[o] ) [+ resid[b,o,p]]
//
// GEMM view per image: M = output channels, N = output pixels, K = (channel, tap).  Activations stay
// fp32 NCHW in HBM, so the pixel axis is the contiguous one: it maps to the MFMA "N" lanes and every
// global/LDS access along it is coalesced / bank-conflict free without any layout transform.
//
// Workgroup = 256 threads (4 waves), output tile BM channels x 128 pixels (a TH x TW rectangle of the
// output grid).  K is walked in chunks of BKC input channels x all T taps:
//   * the weight slice [T*BKC][BM] comes pre-packed K-major (lama_conv2d_pack_weight), copied with
//     16-byte loads straight into LDS rows;
//   * the input *patch* [BKC][PH][PW] (tile + halo, reflection / zero padding already applied) is
//     staged once per chunk and re-read T times from LDS with a per-tap scalar offset -- the 9x (49x)
//     im2col reuse never touches L2/HBM;
//   * global loads for chunk i+1 are issued into registers before the MFMAs of chunk i (register
//     staged pipeline, one LDS buffer, two barriers per chunk).
// Two K segments can be chained into one accumulator (3x3 over x, then 1x1 over x2) so that
//   out_xg = convl2g(x_l) + convg2g.conv2(x1 + fu(x1))   (ffc.py:161,223)
// plus BatchNorm shift, ReLU and the resnet residual (ffc.py:253-254,288) is a single pass over HBM.
//
// Precision LAMA_PREC_F32 uses v_mfma_f32_32x32x2_f32 (exact fp32, bitwise an fmaf chain).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CONV_BN 128
#define CONV_MAX_TAPS 49

struct ConvSeg {
    const float* x;
    long long bstride;
    int C, H, W;
    const float* w;  // packed [nchunk][T][BKC][Mpad]
    int stride, pad_mode;
    int dy0, dx0;  // patch origin relative to (gy*stride, gx*stride)
    int PH, PW;    // patch rows / cols
    int tapoff[CONV_MAX_TAPS];
};

struct ConvParams {
    ConvSeg s1, s2;
    const float* bias;
    const float* resid;
    long long resid_bstride;
    float* y;
    long long y_bstride;
    int M, Mpad, MT;
    int Ho, Wo;                     // full output plane
    int GH, GW, oy0, ox0, ostep;    // output grid of this launch: out = g*ostep + o0
    int TWlog;                      // tile = TH x TW pixels, TW = 1<<TWlog, TH = 128>>TWlog
    int tiles_x, tiles_y, B;
    int act;
};

template <int BM>
struct ConvGeom {
    static constexpr int WAVES_M = (BM >= 64) ? 2 : 1;
    static constexpr int WAVES_N = 4 / WAVES_M;
    static constexpr int WTM = BM / WAVES_M;       // rows per wave
    static constexpr int WTN = CONV_BN / WAVES_N;  // pixels per wave
    static constexpr int TM = WTM / 32;
    static constexpr int TN = WTN / 32;
};

__device__ __forceinline__ int conv_src_coord(int i, int n, int pad_mode) {
    // returns the source index, or -1 for a zero-padded tap
    if (pad_mode == LAMA_PAD_REFLECT) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
        if (i < 0) i = 0;          // only reachable for pixels of a ragged tile that are never stored
        if (i >= n) i = n - 1;
        return i;
    }
    return (i < 0 || i >= n) ? -1 : i;
}

// One K segment: accumulate into acc.  As / Ps are the LDS weight-slice and patch buffers.
template <int T, int BKC, int BM, int MAXP>
__device__ __forceinline__ void conv_segment(const ConvSeg& s, int Mpad, int mt, int b, int gy0, int gx0, int TWlog,
                                             float* As, float* Ps,
                                             f32x16 (&acc)[ConvGeom<BM>::TM][ConvGeom<BM>::TN]) {
    using G = ConvGeom<BM>;
    constexpr int NA4_TOTAL = T * BKC * BM / 4;
    constexpr int NA4 = (NA4_TOTAL + LAMA_NTHREADS - 1) / LAMA_NTHREADS;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / G::WAVES_N, wn = wave % G::WAVES_N;
    const int khalf = lane >> 5, l31 = lane & 31;
    const int TW = 1 << TWlog;
    const int CHS = s.PH * s.PW;
    const int NP = BKC * CHS;
    const int HW = s.H * s.W;

    // per-thread patch slots: source offset inside the [BKC][H][W] channel group (or -1 = zero)
    int src_off[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        int e = i * LAMA_NTHREADS + tid;
        int off = -1;
        if (e < NP) {
            int cl = e / CHS;
            int r = e - cl * CHS;
            int py = r / s.PW;
            int px = r - py * s.PW;
            int iy = conv_src_coord(gy0 * s.stride + s.dy0 + py, s.H, s.pad_mode);
            int ix = conv_src_coord(gx0 * s.stride + s.dx0 + px, s.W, s.pad_mode);
            if (iy >= 0 && ix >= 0) off = cl * HW + iy * s.W + ix;
        }
        src_off[i] = off;
    }
    // per-lane LDS offsets of the B (pixel) operand, one per 32-pixel sub-tile
    int pixoff[G::TN];
#pragma unroll
    for (int j = 0; j < G::TN; ++j) {
        int n = wn * G::WTN + j * 32 + l31;
        int ty = n >> TWlog, tx = n & (TW - 1);
        pixoff[j] = khalf * CHS + ty * s.stride * s.PW + tx * s.stride;
    }
    const int arow = khalf * BM + wm * G::WTM + l31;

    const int nchunk = (s.C + BKC - 1) / BKC;
    const float* xb = s.x + (long long)b * s.bstride;
    const float* wb = s.w + (long long)mt * BM;

    float4 areg[NA4];
    float preg[MAXP];
    // prefetch chunk 0
    {
        const float* wc = wb;
#pragma unroll
        for (int i = 0; i < NA4; ++i) {
            int idx = i * LAMA_NTHREADS + tid;
            if (idx < NA4_TOTAL) {
                int row = idx / (BM / 4), c4 = idx % (BM / 4);
                areg[i] = *reinterpret_cast<const float4*>(wc + (long long)row * Mpad + c4 * 4);
            }
        }
        int lim = (s.C < BKC ? s.C : BKC) * HW;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            int o = src_off[i];
            preg[i] = (o >= 0 && o < lim) ? xb[o] : 0.0f;
        }
    }
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();  // previous chunk's LDS reads are done
#pragma unroll
        for (int i = 0; i < NA4; ++i) {
            int idx = i * LAMA_NTHREADS + tid;
            if (idx < NA4_TOTAL) *reinterpret_cast<float4*>(As + idx * 4) = areg[i];
        }
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            int e = i * LAMA_NTHREADS + tid;
            if (e < NP) Ps[e] = preg[i];
        }
        __syncthreads();
        if (ch + 1 < nchunk) {  // issue next chunk's global loads; they land while the MFMAs run
            const float* wc = wb + (long long)(ch + 1) * (T * BKC) * Mpad;
#pragma unroll
            for (int i = 0; i < NA4; ++i) {
                int idx = i * LAMA_NTHREADS + tid;
                if (idx < NA4_TOTAL) {
                    int row = idx / (BM / 4), c4 = idx % (BM / 4);
                    areg[i] = *reinterpret_cast<const float4*>(wc + (long long)row * Mpad + c4 * 4);
                }
            }
            int crem = s.C - (ch + 1) * BKC;
            int lim = (crem < BKC ? crem : BKC) * HW;
            const float* xc = xb + (long long)(ch + 1) * BKC * HW;
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                int o = src_off[i];
                preg[i] = (o >= 0 && o < lim) ? xc[o] : 0.0f;
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int toff = s.tapoff[t];
#pragma unroll
            for (int q = 0; q < BKC / 2; ++q) {
                float a[G::TM], bb[G::TN];
#pragma unroll
                for (int i = 0; i < G::TM; ++i) a[i] = As[(t * BKC + 2 * q) * BM + arow + i * 32];
#pragma unroll
                for (int j = 0; j < G::TN; ++j) bb[j] = Ps[2 * q * CHS + pixoff[j] + toff];
#pragma unroll
                for (int i = 0; i < G::TM; ++i)
#pragma unroll
                    for (int j = 0; j < G::TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
            }
        }
    }
}

template <int T1, int BKC1, int T2, int BKC2, int BM>
__global__ __launch_bounds__(LAMA_NTHREADS) void conv_mfma_f32_kernel(ConvParams p) {
    using G = ConvGeom<BM>;
    constexpr int MAXP1 = (T1 == 1) ? (BKC1 * CONV_BN / LAMA_NTHREADS) : 10;
    constexpr int MAXP2 = (T2 == 1) ? (BKC2 * CONV_BN / LAMA_NTHREADS) : 10;
    constexpr int A1 = T1 * BKC1 * BM, A2 = T2 * BKC2 * BM;
    constexpr int AMAX = A1 > A2 ? A1 : A2;
    float* As = reinterpret_cast<float*>(lama_smem);
    float* Ps = As + AMAX;

    const int L = lama_xcd_remap(blockIdx.x, gridDim.x);
    const int mt = L % p.MT;
    int tile = L / p.MT;
    const int tix = tile % p.tiles_x;
    tile /= p.tiles_x;
    const int tiy = tile % p.tiles_y;
    const int b = tile / p.tiles_y;
    const int TW = 1 << p.TWlog, TH = CONV_BN >> p.TWlog;
    const int gy0 = tiy * TH, gx0 = tix * TW;

    f32x16 acc[G::TM][G::TN];
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    conv_segment<T1, BKC1, BM, MAXP1>(p.s1, p.Mpad, mt, b, gy0, gx0, p.TWlog, As, Ps, acc);
    if constexpr (T2 > 0) conv_segment<T2, BKC2, BM, MAXP2>(p.s2, p.Mpad, mt, b, gy0, gx0, p.TWlog, As, Ps, acc);

    // epilogue: D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / G::WAVES_N, wn = wave % G::WAVES_N;
    const long long plane = (long long)p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < G::TN; ++j) {
        int n = wn * G::WTN + j * 32 + (lane & 31);
        int gy = gy0 + (n >> p.TWlog), gx = gx0 + (n & (TW - 1));
        bool pv = gy < p.GH && gx < p.GW;
        long long pix = (long long)(gy * p.ostep + p.oy0) * p.Wo + (gx * p.ostep + p.ox0);
#pragma unroll
        for (int i = 0; i < G::TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = mt * BM + wm * G::WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (pv && m < p.M) {
                    float v = acc[i][j][r];
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
}

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
struct PackParams {
    const float* w;
    const float* scale;
    float* dst;
    int M, Mpad, C, T, BKC, nchunk;
    int kh, kw, transposed;
    int tap_ky[CONV_MAX_TAPS], tap_kx[CONV_MAX_TAPS];
};

__global__ void conv_pack_weight_kernel(PackParams p) {
    long long total = (long long)p.nchunk * p.T * p.BKC * p.Mpad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int m = (int)(i % p.Mpad);
        long long r = i / p.Mpad;
        int cl = (int)(r % p.BKC);
        r /= p.BKC;
        int t = (int)(r % p.T);
        int ch = (int)(r / p.T);
        int c = ch * p.BKC + cl;
        float v = 0.0f;
        if (m < p.M && c < p.C) {
            int ky = p.tap_ky[t], kx = p.tap_kx[t];
            long long src = p.transposed ? (((long long)c * p.M + m) * p.kh + ky) * p.kw + kx
                                         : (((long long)m * p.C + c) * p.kh + ky) * p.kw + kx;
            v = p.w[src];
            if (p.scale) v *= p.scale[m];
        }
        p.dst[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// host side: plan, pack, launch
// ------------------------------------------------------------------------------------------------
namespace {

struct ConvPlan {
    int BM, Mpad, MT;
    int nseg;        // sub-convolutions (1, or 4 output-parity classes for ConvTranspose2d)
    int T[4], BKC[4];
    int ky[4][CONV_MAX_TAPS], kx[4][CONV_MAX_TAPS];  // weight tap
    int dy[4][CONV_MAX_TAPS], dx[4][CONV_MAX_TAPS];  // input offset of the tap
    int oy0[4], ox0[4];
    long long woff[4];  // float offset of each class in the packed buffer
    long long total_floats;
};

int pick_bm(int M) {
    if (M > 64 && (M % 128 == 0 || M % 64 != 0)) return 128;
    if (M > 32) return 64;
    return 32;
}

bool make_plan(int cout, int cin, int kh, int kw, int stride, int pad, int transposed, ConvPlan* pl) {
    pl->BM = pick_bm(cout);
    pl->Mpad = lama_round_up(cout, pl->BM);
    pl->MT = pl->Mpad / pl->BM;
    if (transposed) {
        if (kh != 3 || kw != 3 || stride != 2 || pad != 1) return false;
        pl->nseg = 4;
        long long off = 0;
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
            pl->BKC[cls] = (t == 1) ? 16 : 8;
            pl->oy0[cls] = py; pl->ox0[cls] = px;
            pl->woff[cls] = off;
            off += (long long)lama_ceil_div(cin, pl->BKC[cls]) * t * pl->BKC[cls] * pl->Mpad;
        }
        pl->total_floats = off;
        return true;
    }
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
    pl->BKC[0] = (t == 1) ? 32 : (t == 49 ? 4 : (stride == 2 ? 4 : 8));
    pl->oy0[0] = pl->ox0[0] = 0;
    pl->woff[0] = 0;
    pl->total_floats = (long long)lama_ceil_div(cin, pl->BKC[0]) * t * pl->BKC[0] * pl->Mpad;
    return true;
}

// fill one segment; returns false when no tile shape fits the register-staged patch
bool fill_seg(ConvSeg* s, const lama_tensor& x, const float* w, const ConvPlan& pl, int cls, int stride, int pad_mode,
              int TWlog, int maxp, bool flat) {
    const int T = pl.T[cls], BKC = pl.BKC[cls];
    s->x = (const float*)x.ptr;
    s->bstride = x.batch_stride;
    s->C = x.C;
    s->H = flat ? 1 : x.H;
    s->W = flat ? x.H * x.W : x.W;
    s->w = w;
    s->stride = stride;
    s->pad_mode = pad_mode;
    int dymin = 1 << 30, dymax = -(1 << 30), dxmin = 1 << 30, dxmax = -(1 << 30);
    for (int t = 0; t < T; ++t) {
        dymin = pl.dy[cls][t] < dymin ? pl.dy[cls][t] : dymin;
        dymax = pl.dy[cls][t] > dymax ? pl.dy[cls][t] : dymax;
        dxmin = pl.dx[cls][t] < dxmin ? pl.dx[cls][t] : dxmin;
        dxmax = pl.dx[cls][t] > dxmax ? pl.dx[cls][t] : dxmax;
    }
    const int TW = 1 << TWlog, TH = CONV_BN >> TWlog;
    s->dy0 = dymin;
    s->dx0 = dxmin;
    s->PH = (TH - 1) * stride + (dymax - dymin) + 1;
    s->PW = (TW - 1) * stride + (dxmax - dxmin) + 1;
    for (int t = 0; t < T; ++t) s->tapoff[t] = (pl.dy[cls][t] - dymin) * s->PW + (pl.dx[cls][t] - dxmin);
    return BKC * s->PH * s->PW <= maxp * LAMA_NTHREADS;
}

template <int T1, int BKC1, int T2, int BKC2>
int launch_bm(hipStream_t st, const ConvParams& p, int BM, int grid, size_t shmem) {
    switch (BM) {
        case 128: hipLaunchKernelGGL((conv_mfma_f32_kernel<T1, BKC1, T2, BKC2, 128>), dim3(grid), dim3(LAMA_NTHREADS), shmem, st, p); break;
        case 64: hipLaunchKernelGGL((conv_mfma_f32_kernel<T1, BKC1, T2, BKC2, 64>), dim3(grid), dim3(LAMA_NTHREADS), shmem, st, p); break;
        case 32: hipLaunchKernelGGL((conv_mfma_f32_kernel<T1, BKC1, T2, BKC2, 32>), dim3(grid), dim3(LAMA_NTHREADS), shmem, st, p); break;
        default: return LAMA_ERR_UNSUPPORTED;
    }
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

int launch_conv(hipStream_t st, const ConvParams& p, int T1, int BKC1, int T2, int BKC2, int BM) {
    const int a1 = T1 * BKC1 * BM, a2 = T2 * BKC2 * BM;
    const int amax = a1 > a2 ? a1 : a2;
    const int p1 = BKC1 * p.s1.PH * p.s1.PW, p2 = T2 ? BKC2 * p.s2.PH * p.s2.PW : 0;
    const size_t shmem = (size_t)(amax + (p1 > p2 ? p1 : p2)) * sizeof(float);
    if (shmem > 160 * 1024) return LAMA_ERR_UNSUPPORTED;
    const int grid = p.B * p.tiles_x * p.tiles_y * p.MT;
    if (grid <= 0) return LAMA_OK;
#define CONV_CASE(t1, b1, t2, b2) \
    if (T1 == t1 && BKC1 == b1 && T2 == t2 && BKC2 == b2) return launch_bm<t1, b1, t2, b2>(st, p, BM, grid, shmem);
    CONV_CASE(9, 8, 0, 0)
    CONV_CASE(9, 8, 1, 32)
    CONV_CASE(9, 4, 0, 0)
    CONV_CASE(1, 32, 0, 0)
    CONV_CASE(49, 4, 0, 0)
    CONV_CASE(1, 16, 0, 0)
    CONV_CASE(2, 8, 0, 0)
    CONV_CASE(4, 8, 0, 0)
#undef CONV_CASE
    return LAMA_ERR_UNSUPPORTED;
}

bool tensor_ok(const lama_tensor& t) { return t.ptr && t.C > 0 && t.H > 0 && t.W > 0 && t.batch_stride >= (int64_t)t.C * t.H * t.W; }

}  // namespace

extern "C" int64_t lama_conv2d_packed_weight_bytes(int32_t cout, int32_t cin, int32_t kh, int32_t kw, int32_t stride,
                                                   int32_t transposed, int32_t precision) {
    ConvPlan pl;
    if (cout <= 0 || cin <= 0) return LAMA_ERR_UNSUPPORTED;
    if (precision == LAMA_PREC_BF16X3) return lama_cb_packed_weight_bytes_bf16x3(cout, cin, kh, kw, stride, transposed);
    if (precision == LAMA_PREC_F16X3 || precision == LAMA_PREC_F16) return lama_cb_packed_weight_bytes_f16x3(cout, cin, kh, kw, stride, transposed);
    if (precision != LAMA_PREC_F32) return LAMA_ERR_UNSUPPORTED;
    if (!make_plan(cout, cin, kh, kw, stride, transposed ? 1 : kh / 2, transposed, &pl)) return LAMA_ERR_UNSUPPORTED;
    return pl.total_floats * (int64_t)sizeof(float);
}

extern "C" int lama_conv2d_pack_weight(void* stream, const float* w, const float* scale, int32_t cout, int32_t cin,
                                       int32_t kh, int32_t kw, int32_t stride, int32_t transposed, int32_t precision,
                                       void* dst) {
    if (!w || !dst || cout <= 0 || cin <= 0) return LAMA_ERR_BAD_ARG;
    if (precision == LAMA_PREC_BF16X3) return lama_cb_pack_weight_bf16x3((hipStream_t)stream, w, scale, cout, cin, kh, kw, stride, transposed, dst);
    if (precision == LAMA_PREC_F16X3 || precision == LAMA_PREC_F16) return lama_cb_pack_weight_f16x3((hipStream_t)stream, w, scale, cout, cin, kh, kw, stride, transposed, dst);
    if (precision != LAMA_PREC_F32) return LAMA_ERR_UNSUPPORTED;
    ConvPlan pl;
    if (!make_plan(cout, cin, kh, kw, stride, transposed ? 1 : kh / 2, transposed, &pl)) return LAMA_ERR_UNSUPPORTED;
    for (int cls = 0; cls < pl.nseg; ++cls) {
        PackParams pp;
        pp.w = w;
        pp.scale = scale;
        pp.dst = (float*)dst + pl.woff[cls];
        pp.M = cout;
        pp.Mpad = pl.Mpad;
        pp.C = cin;
        pp.T = pl.T[cls];
        pp.BKC = pl.BKC[cls];
        pp.nchunk = lama_ceil_div(cin, pl.BKC[cls]);
        pp.kh = kh;
        pp.kw = kw;
        pp.transposed = transposed;
        for (int t = 0; t < pl.T[cls]; ++t) { pp.tap_ky[t] = pl.ky[cls][t]; pp.tap_kx[t] = pl.kx[cls][t]; }
        long long total = (long long)pp.nchunk * pp.T * pp.BKC * pp.Mpad;
        int grid = (int)((total + 255) / 256);
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(conv_pack_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pp);
        LAMA_CHECK_LAUNCH();
    }
    return LAMA_OK;
}

// packed K position k = 16 kq + 8 khalf + i of lama_conv2d_args.fuse1_w holds input channel 32 F + 16 ks + 8 (i / 4) + 4 khalf + i % 4 with
// kq = 2 F + ks: the row (r & 3) + 8 (r >> 2) + 4 khalf that accumulator register r = 8 ks + i of lane half khalf holds in the
// 32-row fragment F of the producing kernel (conv_wreg_dev.inc)
// ---- Winograd F(2x2, 3x3) form of the stride-1 3x3 reflect convolution (wino_dev.inc): split precisions only -------------------------
extern "C" int64_t lama_winograd_packed_weight_bytes(int32_t cout, int32_t cin, int32_t precision) {
    if (precision == LAMA_PREC_BF16X3) return lama_cb_wino_packed_weight_bytes_bf16x3(cout, cin);
    if (precision == LAMA_PREC_F16X3) return lama_cb_wino_packed_weight_bytes_f16x3(cout, cin);
    return LAMA_ERR_UNSUPPORTED;
}

extern "C" int lama_winograd_pack_weight(void* stream, const float* w, const float* scale, int32_t cout, int32_t cin, int32_t precision, void* dst) {
    if (!w || !dst || cout <= 0 || cin <= 0) return LAMA_ERR_BAD_ARG;
    if (precision == LAMA_PREC_BF16X3) return lama_cb_wino_pack_weight_bf16x3((hipStream_t)stream, w, scale, cout, cin, dst);
    if (precision == LAMA_PREC_F16X3) return lama_cb_wino_pack_weight_f16x3((hipStream_t)stream, w, scale, cout, cin, dst);
    return LAMA_ERR_UNSUPPORTED;
}

extern "C" size_t lama_winograd_workspace_bytes(int32_t batch, int32_t cout, int32_t H, int32_t W) {
    const int64_t n = lama_cb_wino_workspace_bytes_f16x3(batch, cout, H, W);     // the same for both split precisions
    return n > 0 ? (size_t)n : 0;
}

extern "C" int32_t lama_winograd_supported(int32_t cout, int32_t cin, int32_t H, int32_t W, int32_t precision) {
    if (precision != LAMA_PREC_F16X3 && precision != LAMA_PREC_BF16X3) return 0;
    return lama_cb_wino_shape_ok_f16x3(cout, cin, H, W) ? 1 : 0;                  // the same for both split precisions
}

// (v110) 1 when the Winograd form of this launch is expected to beat lama_conv2d_fwd (always for the exact geometries; a cost model of whole
// rounds of workgroups for the any-size geometry of round 6: wino_dev.inc), 0 otherwise or when unsupported
extern "C" int32_t lama_winograd_preferred(int32_t batch, int32_t cout, int32_t cin, int32_t H, int32_t W, int32_t precision) {
    if ((precision != LAMA_PREC_F16X3 && precision != LAMA_PREC_BF16X3) || batch <= 0) return 0;
    return lama_cb_wino_preferred_f16x3(batch, cout, cin, H, W) ? 1 : 0;
}

extern "C" int lama_winograd_conv3x3_fwd(void* stream, const lama_conv2d_args* a, void* workspace, size_t workspace_bytes) {
    if (!a || !tensor_ok(a->x) || !tensor_ok(a->y) || !a->w_packed || a->batch <= 0) return LAMA_ERR_BAD_ARG;
    if (a->kh != 3 || a->kw != 3 || a->stride != 1 || a->pad != 1 || (a->pad_mode != LAMA_PAD_REFLECT && a->pad_mode != LAMA_PAD_ZERO) || a->transposed)
        return LAMA_ERR_UNSUPPORTED;
    if (a->x2.ptr || a->fuse1_w) return LAMA_ERR_UNSUPPORTED;
    if (a->x.dtype != LAMA_DT_F32 || a->y.dtype != LAMA_DT_F32 || (a->resid.ptr && a->resid.dtype != LAMA_DT_F32)) return LAMA_ERR_UNSUPPORTED;
    if (a->x.H != a->y.H || a->x.W != a->y.W || a->x.H < 2 || a->x.W < 2) return LAMA_ERR_BAD_ARG;
    if (a->resid.ptr && (a->resid.C != a->y.C || a->resid.H != a->y.H || a->resid.W != a->y.W)) return LAMA_ERR_BAD_ARG;
    // (the exact geometries -- W in {32, 64, 128, 256} -- use 16-byte loads / stores on x, y and resid and want 16-byte aligned views: checked with
    // the geometry in lama_cb_wino_fwd; the any-size geometry of round 6 needs element alignment only)
    if (a->precision == LAMA_PREC_BF16X3) return lama_cb_wino_fwd_bf16x3((hipStream_t)stream, a, workspace, workspace_bytes);
    if (a->precision == LAMA_PREC_F16X3) return lama_cb_wino_fwd_f16x3((hipStream_t)stream, a, workspace, workspace_bytes);
    return LAMA_ERR_UNSUPPORTED;
}

// the checks of lama_winograd_conv3x3_fwd that the second launch needs again
static int wino_args_ok(const lama_conv2d_args* a) {
    if (!a || !tensor_ok(a->x) || !tensor_ok(a->y) || a->batch <= 0) return LAMA_ERR_BAD_ARG;
    if (a->kh != 3 || a->kw != 3 || a->stride != 1 || a->pad != 1 || a->transposed) return LAMA_ERR_UNSUPPORTED;
    if (a->y.dtype != LAMA_DT_F32 || (a->resid.ptr && a->resid.dtype != LAMA_DT_F32)) return LAMA_ERR_UNSUPPORTED;
    if (a->x.H != a->y.H || a->x.W != a->y.W) return LAMA_ERR_BAD_ARG;
    if (a->resid.ptr && (a->resid.C != a->y.C || a->resid.H != a->y.H || a->resid.W != a->y.W)) return LAMA_ERR_BAD_ARG;
    if (a->precision != LAMA_PREC_BF16X3 && a->precision != LAMA_PREC_F16X3) return LAMA_ERR_UNSUPPORTED;
    return LAMA_OK;
}

// the output transform's launch parameters (a WoParams of wino_out_dev.inc) from the arguments of the first launch; used by
// lama_rfft2_winograd_out_fwd (fft.hip)
int lama_wino_out_params(const lama_conv2d_args* a, void* workspace, size_t workspace_bytes, void* out) {
    const int rc = wino_args_ok(a);
    if (rc) return rc;
    return lama_cb_wino_out_params_f16x3(a, workspace, workspace_bytes, out);      // the same for both split precisions
}

extern "C" int lama_winograd_out_fwd(void* stream, const lama_conv2d_args* a, void* workspace, size_t workspace_bytes) {
    const int rc = wino_args_ok(a);
    if (rc) return rc;
    return lama_cb_wino_out_fwd_f16x3((hipStream_t)stream, a, workspace, workspace_bytes);
}

extern "C" void lama_fuse1_channel_order(int32_t* order) {
    for (int kq = 0; kq < 24; ++kq)
        for (int kh = 0; kh < 2; ++kh)
            for (int i = 0; i < 8; ++i) order[16 * kq + 8 * kh + i] = 32 * (kq / 2) + 16 * (kq & 1) + 8 * (i / 4) + 4 * kh + (i % 4);
}

extern "C" int lama_conv2d_fwd(void* stream, const lama_conv2d_args* a) {
    if (!a || !tensor_ok(a->x) || !tensor_ok(a->y) || !a->w_packed || a->batch <= 0) return LAMA_ERR_BAD_ARG;
    if (a->precision != LAMA_PREC_F32 && a->precision != LAMA_PREC_BF16X3 && a->precision != LAMA_PREC_F16X3 && a->precision != LAMA_PREC_F16)
        return LAMA_ERR_UNSUPPORTED;
    {   // element types: fp16 tensors only with LAMA_PREC_F16 (and LAMA_PREC_F16 only with them: at least x or y is fp16)
        const bool xh = a->x.dtype == LAMA_DT_F16, yh = a->y.dtype == LAMA_DT_F16;
        if ((a->x.dtype != LAMA_DT_F32 && !xh) || (a->y.dtype != LAMA_DT_F32 && !yh)) return LAMA_ERR_BAD_ARG;
        if (a->x2.ptr && a->x2.dtype != LAMA_DT_F32 && a->x2.dtype != LAMA_DT_F16) return LAMA_ERR_BAD_ARG;
        if (a->x2.ptr && a->x2.dtype == LAMA_DT_F16 && a->precision != LAMA_PREC_F16) return LAMA_ERR_UNSUPPORTED;
        if (a->resid.ptr && a->resid.dtype != a->y.dtype) return LAMA_ERR_BAD_ARG;
        if ((xh || yh) != (a->precision == LAMA_PREC_F16)) return LAMA_ERR_UNSUPPORTED;
    }
    const int cout = a->y.C, cin = a->x.C;
    ConvPlan pl;
    if (!make_plan(cout, cin, a->kh, a->kw, a->stride, a->pad, a->transposed, &pl)) return LAMA_ERR_UNSUPPORTED;
    // geometry checks
    int Ho, Wo;
    if (a->transposed) {
        Ho = a->x.H * 2; Wo = a->x.W * 2;
    } else {
        Ho = (a->x.H + 2 * a->pad - a->kh) / a->stride + 1;
        Wo = (a->x.W + 2 * a->pad - a->kw) / a->stride + 1;
    }
    if (Ho != a->y.H || Wo != a->y.W) return LAMA_ERR_BAD_ARG;
    if (a->pad_mode == LAMA_PAD_REFLECT && (a->pad >= a->x.H || a->pad >= a->x.W)) return LAMA_ERR_BAD_ARG;
    const bool has2 = a->x2.ptr != nullptr;
    if (has2 && (a->transposed || !a->w2_packed || a->x2.H != Ho || a->x2.W != Wo || !tensor_ok(a->x2))) return LAMA_ERR_BAD_ARG;
    if (a->resid.ptr && (a->resid.C != cout || a->resid.H != Ho || a->resid.W != Wo)) return LAMA_ERR_BAD_ARG;
    if (a->fuse1_w) {   // fused conv1 of the next layer: only the launch that produces the 384-channel state, on the split precisions
        if (!tensor_ok(a->fuse1_y) || a->fuse1_y.dtype != LAMA_DT_F32 || a->fuse1_y.C != 192 || a->fuse1_y.H != Ho || a->fuse1_y.W != Wo) return LAMA_ERR_BAD_ARG;
        if (cout != 384 || !has2 || a->kh != 3 || a->stride != 1 || a->y.dtype != LAMA_DT_F32 ||
            (a->precision != LAMA_PREC_BF16X3 && a->precision != LAMA_PREC_F16X3))
            return LAMA_ERR_UNSUPPORTED;
    }
    if (a->precision == LAMA_PREC_BF16X3) return lama_cb_conv2d_fwd_bf16x3((hipStream_t)stream, a, Ho, Wo);
    if (a->precision == LAMA_PREC_F16X3) return lama_cb_conv2d_fwd_f16x3((hipStream_t)stream, a, Ho, Wo);
    if (a->precision == LAMA_PREC_F16) return lama_cb_conv2d_fwd_f16((hipStream_t)stream, a, Ho, Wo);

    ConvPlan pl2;
    if (has2 && !make_plan(cout, a->x2.C, 1, 1, 1, 0, 0, &pl2)) return LAMA_ERR_UNSUPPORTED;
    if (has2 && pl2.BM != pl.BM) return LAMA_ERR_UNSUPPORTED;

    for (int cls = 0; cls < pl.nseg; ++cls) {
        ConvParams p;
        memset(&p, 0, sizeof(p));
        const bool flat = (pl.T[cls] == 1 && !a->transposed && a->stride == 1 && a->pad == 0 && !has2);
        p.bias = a->bias;
        p.resid = (const float*)a->resid.ptr;
        p.resid_bstride = a->resid.batch_stride;
        p.y = (float*)a->y.ptr;
        p.y_bstride = a->y.batch_stride;
        p.M = cout;
        p.Mpad = pl.Mpad;
        p.MT = pl.MT;
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
        const int maxp1 = pl.T[cls] == 1 ? pl.BKC[cls] * CONV_BN / LAMA_NTHREADS : 10;
        // tile shape: widest power-of-two row segment (<= 32 unless flat) whose patch fits the staging registers
        int twlog = flat ? 7 : 5;
        while (twlog > 3 && (1 << (twlog - 1)) >= p.GW) --twlog;
        bool ok = false;
        for (; twlog >= 3; --twlog) {
            ok = fill_seg(&p.s1, a->x, (const float*)a->w_packed + pl.woff[cls], pl, cls, stride, pad_mode, twlog, maxp1, flat);
            if (ok) break;
        }
        if (!ok) return LAMA_ERR_UNSUPPORTED;
        p.TWlog = twlog;
        p.tiles_x = lama_ceil_div(p.GW, 1 << twlog);
        p.tiles_y = lama_ceil_div(p.GH, CONV_BN >> twlog);
        int T2 = 0, BKC2 = 0;
        if (has2) {
            T2 = 1; BKC2 = pl2.BKC[0];
            if (!fill_seg(&p.s2, a->x2, (const float*)a->w2_packed, pl2, 0, 1, LAMA_PAD_ZERO, twlog, BKC2 * CONV_BN / LAMA_NTHREADS, false))
                return LAMA_ERR_UNSUPPORTED;
        }
        int rc = launch_conv((hipStream_t)stream, p, pl.T[cls], pl.BKC[cls], T2, BKC2, pl.BM);
        if (rc != LAMA_OK) return rc;
    }
    return LAMA_OK;
}
