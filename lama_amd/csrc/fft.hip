// Batched 2-D real FFTs of the FourierUnit (ffc.py:86-89,103-108), norm='ortho'.
//
//   rfft2 : x [B,C,h,w] fp32           -> spec [B,2C,h,wf]  (channel 2c = Re, 2c+1 = Im, wf = w/2+1)
//   irfft2: spec [B,2C,h,wf] (NOT Hermitian: it went through conv+BN+ReLU) -> y = [resid +] irfftn(spec)
//
// The Re/Im interleave + permute + contiguous copies of the reference (ffc.py:87-89,103-105) do not
// exist here: the forward kernel writes the two planes the spectral 1x1 GEMM reads, the inverse
// kernel reads the two planes it wrote.
//
// Fast path (h, w powers of two, 16..128): one workgroup owns PPW whole planes in LDS.
//   rows  : two real rows are transformed as ONE complex FFT (z = row_2f + i*row_2f+1) and untangled,
//   cols  : the purely real DC and Nyquist columns are packed into ONE complex column, so a plane
//           costs h/2 row FFTs of length w plus w/2 column FFTs of length h (both directions);
//   passes: Stockham autosort, radix 8/4/2, ping-pong between two LDS buffers, one barrier per pass;
//           lanes run across independent FFTs (row stride w+1 / wf float2 = odd) so every ds_read_b64 /
//           ds_write_b64 of a pass is bank-conflict free whatever the butterfly stride;
//   inverse: irfftn on a non-Hermitian spectrum = complex inverse along h, then c2r along w that
//           ignores Im of bins 0 and w/2.  Re(ifft_h(col)) of those two columns equals the inverse of
//           their Hermitian-symmetrised parts, which are packed into one complex column again.
// Generic path (any h, w): separable direct DFT, rows and columns in two kernels through a float2
// workspace [B*C][h][wf] (O(n) more flops; used for non power-of-two planes and planes > 128).
#include "common.h"

#define FFT_SQRT1_2 0.70710678118654752440f
__device__ __forceinline__ int lama_ceil_div_dev(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// multiply by -i (forward) or +i (inverse)
template <bool INV>
__device__ __forceinline__ float2 cmul_mi(float2 a) { return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x); }

template <bool INV>
__device__ __forceinline__ void dft2(float2& a, float2& b) {
    float2 t = a;
    a = cadd(t, b);
    b = csub(t, b);
}
// natural-order 4-point DFT, in place
template <bool INV>
__device__ __forceinline__ void dft4(float2& u0, float2& u1, float2& u2, float2& u3) {
    float2 c0 = cadd(u0, u2), c1 = cadd(u1, u3), d0 = csub(u0, u2), d1 = cmul_mi<INV>(csub(u1, u3));
    u0 = cadd(c0, c1);
    u2 = csub(c0, c1);
    u1 = cadd(d0, d1);
    u3 = csub(d0, d1);
}
// natural-order 8-point DFT, in place: X[2m] = DFT4(v[r]+v[r+4]), X[2m+1] = DFT4((v[r]-v[r+4]) W8^r)
template <bool INV>
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
    float2 a0 = cadd(v[0], v[4]), a1 = cadd(v[1], v[5]), a2 = cadd(v[2], v[6]), a3 = cadd(v[3], v[7]);
    float2 b0 = csub(v[0], v[4]), b1 = csub(v[1], v[5]), b2 = csub(v[2], v[6]), b3 = csub(v[3], v[7]);
    const float s = FFT_SQRT1_2;
    // W8^1 = (s, -s) fwd / (s, +s) inv ; W8^2 = -i / +i ; W8^3 = (-s, -s) fwd / (-s, +s) inv
    b1 = INV ? make_float2(s * (b1.x - b1.y), s * (b1.x + b1.y)) : make_float2(s * (b1.x + b1.y), s * (b1.y - b1.x));
    b2 = cmul_mi<INV>(b2);
    b3 = INV ? make_float2(-s * (b3.x + b3.y), s * (b3.x - b3.y)) : make_float2(s * (b3.y - b3.x), -s * (b3.x + b3.y));
    dft4<INV>(a0, a1, a2, a3);
    dft4<INV>(b0, b1, b2, b3);
    v[0] = a0; v[2] = a1; v[4] = a2; v[6] = a3;
    v[1] = b0; v[3] = b1; v[5] = b2; v[7] = b3;
}

// One Stockham pass of radix R over `nfft` length-N FFTs living in LDS.
//   element j of FFT f: src[f*fstride + j*estride]; tw[m] = exp(-/+ 2 pi i m / N).
//   out[(j/Ns)*Ns*R + (j%Ns) + r*Ns] = sum_r' in[j + r'*N/R] * w^(r'*(j%Ns)) * W_R^(r r')
template <int R, bool INV>
__device__ __forceinline__ void fft_pass(const float2* src, float2* dst, const float2* tw, int N, int Ns, int nfft,
                                         int estride, int fstride, int ninner, int ostride) {
    const int nb = N / R;
    const int items = nfft * nb;
    const int twstep = N / (Ns * R);
    for (int item = threadIdx.x; item < items; item += LAMA_NTHREADS) {
        int f = item % nfft, j = item / nfft;
        int k = j & (Ns - 1);
        int fo = f / ninner;
        const int fbase = fo * ostride + (f - fo * ninner) * fstride;  // FFT f = (outer fo, inner fi)
        const float2* s = src + fbase;
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = s[(j + r * nb) * estride];
        if (Ns > 1) {
#pragma unroll
            for (int r = 1; r < R; ++r) v[r] = cmul(v[r], tw[r * k * twstep]);
        }
        if constexpr (R == 8) {
            dft8<INV>(v);
        } else if constexpr (R == 4) {
            dft4<INV>(v[0], v[1], v[2], v[3]);
        } else {
            dft2<INV>(v[0], v[1]);
        }
        float2* d = dst + fbase;
        int j0 = (j - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) d[(j0 + r * Ns) * estride] = v[r];
    }
}

// Full FFT of length N (power of two >= 2) over nfft sequences; ping-pongs between bufA (input) and
// bufB, returns the buffer holding the result.  Ends with a barrier.
template <bool INV>
__device__ __forceinline__ float2* fft_lds(float2* bufA, float2* bufB, const float2* tw, int N, int nfft, int estride,
                                           int fstride, int ninner, int ostride) {
    float2* src = bufA;
    float2* dst = bufB;
    int Ns = 1;
    while (Ns < N) {
        int rem = N / Ns;
        if (rem >= 8) {
            fft_pass<8, INV>(src, dst, tw, N, Ns, nfft, estride, fstride, ninner, ostride);
            Ns *= 8;
        } else if (rem == 4) {
            fft_pass<4, INV>(src, dst, tw, N, Ns, nfft, estride, fstride, ninner, ostride);
            Ns *= 4;
        } else {
            fft_pass<2, INV>(src, dst, tw, N, Ns, nfft, estride, fstride, ninner, ostride);
            Ns *= 2;
        }
        __syncthreads();
        float2* t = src;
        src = dst;
        dst = t;
    }
    return src;
}

// Activation elements in HBM are fp32 or fp16 (lama_tensor.dtype; kernel template parameter HF): a typed pointer whose element
// access converts to / from float, plus 4-element vector accesses (16 B of fp32 or 8 B of fp16, naturally aligned).
template <bool HF>
struct ActP {
    using T = typename LamaAct<HF>::T;
    T* p;
    __device__ __forceinline__ ActP operator+(long long o) const { return ActP{p + o}; }
    struct Ref {
        T* q;
        __device__ __forceinline__ operator float() const { return (float)*q; }
        __device__ __forceinline__ void operator=(float v) const { *q = (T)v; }
    };
    __device__ __forceinline__ Ref operator[](long long i) const { return Ref{p + i}; }
    __device__ __forceinline__ explicit operator bool() const { return p != nullptr; }
};
__device__ __forceinline__ float4 fft_ld4(ActP<false> a) { return *reinterpret_cast<const float4*>(a.p); }
__device__ __forceinline__ void fft_st4(ActP<false> a, float4 v) { *reinterpret_cast<float4*>(a.p) = v; }
typedef _Float16 fft_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 fft_ld4(ActP<true> a) {
    const fft_h4 h = *reinterpret_cast<const fft_h4*>(a.p);
    return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
__device__ __forceinline__ void fft_st4(ActP<true> a, float4 v) {
    *reinterpret_cast<fft_h4*>(a.p) = fft_h4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
}
#define FFT_IO(p)                                                                                                    \
    const ActP<HF> px{(typename LamaAct<HF>::T*)const_cast<void*>((p).x)}, ps{(typename LamaAct<HF>::T*)(p).spec},   \
        py{(typename LamaAct<HF>::T*)(p).y};                                                                         \
    (void)px; (void)ps; (void)py

static inline int lama_side_prio() { static const int v = lama_env_int("LAMA_SIDE_PRIO", 1); return v; }   // profiling tools: 0 = default wave priority

struct FftParams {
    const void* x;       // forward: input planes; inverse: residual (may be null).  fp32 or fp16 elements (template parameter HF of
    long long x_bstride; // the kernels = lama_tensor.dtype of all three tensors); strides are in ELEMENTS
    void* spec;          // forward: output; inverse: input
    long long spec_bstride;
    void* y;             // inverse output
    long long y_bstride;
    int C, h, w, wf;
    int w1, w2, h1, h2;  // generic path: w = w1 * w2, h = h1 * h2 (one Cooley-Tukey split; w1 = 1 for a prime)
    int nplanes;         // B*C
    int ppw;             // planes per workgroup
    float scale;         // 1/sqrt(h*w)
    int prio;            // raise the wave priority (s_setprio): these launches are the short links of the spectral branch, which runs beside
                         // the local 3x3 conv of the other stream on the same SIMDs
    long long* trace;    // profiling tools only (LAMA_FFT_TRACE): 16 int64 per workgroup, 100 MHz ticks at the phase boundaries
    const float* mask;   // round 4 (fq kernels only): the output is multiplied by [mask > 0] -- the ReLU derivative that follows the transform in
    long long mask_bstride;   // the reverse pass (forward: a tensor laid out like spec; inverse: like y)
};

template <bool INV>
__device__ __forceinline__ void fft_init_twiddles(float2* tw, int N) {
    // sincospif of the exactly representable fraction 2m/N (N <= 2^24): full float accuracy at a fraction of the cost of the
    // double-precision call, which dominated the set-up of these short workgroups
    for (int m = threadIdx.x; m < N; m += LAMA_NTHREADS) {
        float s, c;
        sincospif(2.0f * (float)m / (float)N, &s, &c);
        tw[m] = make_float2(c, INV ? s : -s);
    }
}

// lane -> (f, q) map for the row-pair load/store: 16 consecutive lanes = 4 float4 columns x 4 row pairs,
// which makes the ds_write_b64/ds_read_b64 of the interleaved (row 2f, row 2f+1) float2 conflict free.
__device__ __forceinline__ void rowpair_item(int item, int wq, int& f, int& q) {
    int q_lo = item & 3, f_lo = (item >> 2) & 3, rest = item >> 4;
    int wq4 = wq >> 2;  // w/16
    int q_hi = rest % wq4, f_hi = rest / wq4;
    q = q_hi * 4 + q_lo;
    f = f_hi * 4 + f_lo;
}

// HT / WT: compile-time plane size (0 = take it from the parameters).  With constant sizes every div / mod of the item
// decomposition folds to shifts and the pass loops unroll -- the generic instantiation is VALU-bound on that index math.
// SEQ > 1 (sized instantiations with one plane in LDS): the workgroup walks SEQ consecutive planes and requests plane s + 1
// from HBM (into registers) before it transforms plane s, so only the first plane's load latency is exposed, the twiddles are
// set up once, and a launch of B*C planes is B*C / SEQ workgroups that are all resident at once (no tail round).
#define FFT_STAMP(i) do { if constexpr (TR) { if (threadIdx.x == 0) p.trace[(long long)blockIdx.x * 16 + (i)] = LAMA_CLOCK(); } } while (0)
template <int HT, int WT, int PPW, int SEQ = 1, bool TR = false, bool HF = false>
__global__ __launch_bounds__(LAMA_NTHREADS) void rfft2_lds_kernel(FftParams p) {
    FFT_IO(p);
    FFT_STAMP(0);
    static_assert(SEQ == 1 || (HT > 0 && WT > 0 && PPW == 1), "sequential planes: sized one-plane instantiations only");
    const int h = HT ? HT : p.h, w = WT ? WT : p.w, wf = w / 2 + 1, hh = h >> 1, wh = w >> 1;
    const int RSW = w + 1;
    const int bufsz = (PPW ? PPW : p.ppw) * (hh * RSW > h * wf ? hh * RSW : h * wf);
    float2* tww = reinterpret_cast<float2*>(lama_smem);
    float2* twh = tww + w;
    float2* P = twh + h;
    float2* Q = P + bufsz;
    const int tid = threadIdx.x;
    const int ppw = PPW ? PPW : p.ppw;   // the sized instantiations are only launched when ppw divides the plane count
    const int np = PPW ? PPW : ((p.nplanes - (int)blockIdx.x * ppw) < ppw ? (p.nplanes - (int)blockIdx.x * ppw) : ppw);

    fft_init_twiddles<false>(tww, w);
    fft_init_twiddles<false>(twh, h);
    // row-pair prefetch registers of the sequential variant
    constexpr int NPF = SEQ > 1 ? ((HT / 2) * (WT / 4) + LAMA_NTHREADS - 1) / LAMA_NTHREADS : 1;
    float4 pra[NPF], prb[NPF];
    auto prefetch = [&](int plane) {
        const int b = plane / p.C, c = plane - b * p.C;
        const auto base = px + (long long)b * p.x_bstride + (long long)c * h * w;
#pragma unroll
        for (int it = 0; it < NPF; ++it) {
            const int item = tid + it * LAMA_NTHREADS;
            if (item < hh * (w >> 2)) {
                int f, q;
                rowpair_item(item, w >> 2, f, q);
                const auto src = base + (2 * f) * w + q * 4;
                pra[it] = fft_ld4(src);
                prb[it] = fft_ld4(src + w);
            }
        }
    };
    if constexpr (SEQ > 1) prefetch(blockIdx.x * SEQ);
#pragma unroll 1
    for (int sq = 0; sq < SEQ; ++sq) {
    const int plane0 = (blockIdx.x * SEQ + sq) * ppw;
    // 1. load row pairs: P[pl][f][n] = (x[2f][n], x[2f+1][n])
    if constexpr (SEQ > 1) {
#pragma unroll
        for (int it = 0; it < NPF; ++it) {
            const int item = tid + it * LAMA_NTHREADS;
            if (item < hh * (w >> 2)) {
                int f, q;
                rowpair_item(item, w >> 2, f, q);
                float2* d = P + f * RSW + q * 4;
                d[0] = make_float2(pra[it].x, prb[it].x);
                d[1] = make_float2(pra[it].y, prb[it].y);
                d[2] = make_float2(pra[it].z, prb[it].z);
                d[3] = make_float2(pra[it].w, prb[it].w);
            }
        }
        if (sq + 1 < SEQ) prefetch(plane0 + 1);
    } else {
        const int wq = w >> 2;
        const int per_plane = hh * wq;
        for (int item = tid; item < np * per_plane; item += LAMA_NTHREADS) {
            int pl = item / per_plane, f, q;
            rowpair_item(item - pl * per_plane, wq, f, q);
            int plane = plane0 + pl;
            int b = plane / p.C, c = plane - b * p.C;
            const auto src = px + (long long)b * p.x_bstride + (long long)c * h * w + (2 * f) * w + q * 4;
            float4 ra = fft_ld4(src);
            float4 rb = fft_ld4(src + w);
            float2* d = P + (pl * hh + f) * RSW + q * 4;
            d[0] = make_float2(ra.x, rb.x);
            d[1] = make_float2(ra.y, rb.y);
            d[2] = make_float2(ra.z, rb.z);
            d[3] = make_float2(ra.w, rb.w);
        }
    }
    __syncthreads();
    FFT_STAMP(1);
    // 2. row FFTs (length w) of the packed row pairs
    float2* E1 = fft_lds<false>(P, Q, tww, w, np * hh, 1, RSW, np * hh, 0);
    FFT_STAMP(2);
    float2* S = (E1 == P) ? Q : P;  // spectrum buffer, [pl][h][wf]
    // 3. untangle the pairs into the half spectra of the two rows; column 0 packs (DC, Nyquist)
    for (int item = tid; item < np * hh * wh; item += LAMA_NTHREADS) {
        int fg = item % (np * hh), k = item / (np * hh);
        int pl = fg / hh, f = fg - pl * hh;
        const float2* z = E1 + fg * RSW;
        float2* sa = S + (pl * h + 2 * f) * wf;
        float2* sb = sa + wf;
        if (k == 0) {
            float2 z0 = z[0], zn = z[wh];
            sa[0] = make_float2(z0.x, zn.x);
            sb[0] = make_float2(z0.y, zn.y);
        } else {
            float2 zk = z[k], zm = z[w - k];
            sa[k] = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
            sb[k] = make_float2(0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x));
        }
    }
    __syncthreads();
    FFT_STAMP(3);
    // 4. column FFTs (length h) over columns 0..w/2-1 of every plane
    //    FFT index = (plane, column): inner stride 1 (column), outer stride h*wf (plane)
    float2* E2 = fft_lds<false>(S, E1, twh, h, np * wh, wf, 1, wh, h * wf);
    FFT_STAMP(4);
    if constexpr (SEQ > 1) {
        // 5 + 6 in one phase: float4 stores of the Re / Im planes; the DC (col 0) and Nyquist (col w/2) values are untangled
        // from the packed column 0 on the fly (two extra LDS reads for 2 of the wf columns), which saves two barriers
        const int per_plane = h * wf;
        const int b = plane0 / p.C, c = plane0 - b * p.C;
        const auto dre = ps + (long long)b * p.spec_bstride + (long long)(2 * c) * per_plane;
        for (int i4 = tid; i4 < per_plane / 4; i4 += LAMA_NTHREADS) {
            float re[4], im[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i4 * 4 + e;
                const int k = i / wf, col = i - k * wf;
                float2 v = E2[i];
                if (col == 0 || col == wh) {
                    const float2 cc = E2[k * wf], cm = E2[((h - k) & (h - 1)) * wf];
                    v = col == 0 ? make_float2(0.5f * (cc.x + cm.x), 0.5f * (cc.y - cm.y)) : make_float2(0.5f * (cc.y + cm.y), 0.5f * (cm.x - cc.x));
                }
                re[e] = v.x * p.scale;
                im[e] = v.y * p.scale;
            }
            fft_st4(dre + i4 * 4, make_float4(re[0], re[1], re[2], re[3]));
            fft_st4(dre + per_plane + i4 * 4, make_float4(im[0], im[1], im[2], im[3]));
        }
        if (sq + 1 < SEQ) __syncthreads();   // the next plane's row pairs overwrite the buffers
    } else {
    // 5. untangle column 0 into the DC (col 0) and Nyquist (col w/2) columns
    {
        float2 x0 = make_float2(0.f, 0.f), xn = x0;
        const bool act = tid < np * h;
        int pl = tid / h, k = tid - pl * h;
        if (act) {
            float2 c = E2[(pl * h + k) * wf], cm = E2[(pl * h + ((h - k) & (h - 1))) * wf];
            x0 = make_float2(0.5f * (c.x + cm.x), 0.5f * (c.y - cm.y));
            xn = make_float2(0.5f * (c.y + cm.y), 0.5f * (cm.x - c.x));
        }
        __syncthreads();
        if (act) {
            E2[(pl * h + k) * wf] = x0;
            E2[(pl * h + k) * wf + wh] = xn;
        }
    }
    __syncthreads();
    FFT_STAMP(5);
    // 6. store the Re / Im planes (channels 2c, 2c+1), ortho scale
    {
        const int per_plane = h * wf;
        for (int item = tid; item < np * per_plane; item += LAMA_NTHREADS) {
            int pl = item / per_plane, i = item - pl * per_plane;
            int plane = plane0 + pl;
            int b = plane / p.C, c = plane - b * p.C;
            float2 v = E2[pl * per_plane + i];
            const auto dst = ps + (long long)b * p.spec_bstride + (long long)(2 * c) * per_plane + i;
            dst[0] = v.x * p.scale;
            dst[per_plane] = v.y * p.scale;
        }
    }
    }
    }   // planes of this workgroup
    FFT_STAMP(6);
}

// SEQ > 1: as in rfft2_lds_kernel -- SEQ consecutive planes per workgroup, the spectrum of plane s + 1 (float4 loads of the Re / Im
// planes) is requested before plane s is transformed, and the residual of plane s right behind its own spectrum (not at store time).
template <int HT, int WT, int PPW, int SEQ = 1, bool TR = false, bool HF = false>
__global__ __launch_bounds__(LAMA_NTHREADS) void irfft2_lds_kernel(FftParams p) {
    FFT_IO(p);
    FFT_STAMP(0);
    static_assert(SEQ == 1 || (HT > 0 && WT > 0 && PPW == 1), "sequential planes: sized one-plane instantiations only");
    const int h = HT ? HT : p.h, w = WT ? WT : p.w, wf = w / 2 + 1, hh = h >> 1, wh = w >> 1;
    const int RSW = w + 1;
    const int bufsz = (PPW ? PPW : p.ppw) * (hh * RSW > h * wf ? hh * RSW : h * wf);
    float2* tww = reinterpret_cast<float2*>(lama_smem);
    float2* twh = tww + w;
    float2* P = twh + h;
    float2* Q = P + bufsz;
    const int tid = threadIdx.x;
    const int ppw = PPW ? PPW : p.ppw;   // the sized instantiations are only launched when ppw divides the plane count
    const int np = PPW ? PPW : ((p.nplanes - (int)blockIdx.x * ppw) < ppw ? (p.nplanes - (int)blockIdx.x * ppw) : ppw);
    const int per_plane = h * wf;

    fft_init_twiddles<true>(tww, w);
    fft_init_twiddles<true>(twh, h);
    constexpr int NSP = SEQ > 1 ? (HT * (WT / 2 + 1) / 4 + LAMA_NTHREADS - 1) / LAMA_NTHREADS : 1;    // float4 items of one spectrum plane
    constexpr int NRP = SEQ > 1 ? ((HT / 2) * (WT / 4) + LAMA_NTHREADS - 1) / LAMA_NTHREADS : 1;       // row-pair items of one output plane
    float4 sre[NSP], sim[NSP], rxa[NRP], rxb[NRP];
    auto prefetch_spec = [&](int plane) {
        const int b = plane / p.C, c = plane - b * p.C;
        const auto base = ps + (long long)b * p.spec_bstride + (long long)(2 * c) * per_plane;
#pragma unroll
        for (int it = 0; it < NSP; ++it) {
            const int i4 = tid + it * LAMA_NTHREADS;
            if (i4 < per_plane / 4) {
                sre[it] = fft_ld4(base + i4 * 4);
                sim[it] = fft_ld4(base + per_plane + i4 * 4);
            }
        }
    };
    auto prefetch_resid = [&](int plane) {
        const int b = plane / p.C, c = plane - b * p.C;
        const auto base = px + (long long)b * p.x_bstride + (long long)c * h * w;
#pragma unroll
        for (int it = 0; it < NRP; ++it) {
            const int item = tid + it * LAMA_NTHREADS;
            if (item < hh * (w >> 2)) {
                int f, q;
                rowpair_item(item, w >> 2, f, q);
                rxa[it] = fft_ld4(base + (2 * f) * w + q * 4);
                rxb[it] = fft_ld4(base + (2 * f + 1) * w + q * 4);
            }
        }
    };
    if constexpr (SEQ > 1) prefetch_spec(blockIdx.x * SEQ);
#pragma unroll 1
    for (int sq = 0; sq < SEQ; ++sq) {
    const int plane0 = (blockIdx.x * SEQ + sq) * ppw;
    // 1. load the Re / Im planes: P[pl][u][k]
    if constexpr (SEQ > 1) {
#pragma unroll
        for (int it = 0; it < NSP; ++it) {
            const int i4 = tid + it * LAMA_NTHREADS;
            if (i4 < per_plane / 4) {
                float2* d = P + i4 * 4;
                d[0] = make_float2(sre[it].x, sim[it].x);
                d[1] = make_float2(sre[it].y, sim[it].y);
                d[2] = make_float2(sre[it].z, sim[it].z);
                d[3] = make_float2(sre[it].w, sim[it].w);
            }
        }
        if (px) prefetch_resid(plane0);
        if (sq + 1 < SEQ) prefetch_spec(plane0 + 1);
    } else {
    for (int item = tid; item < np * per_plane; item += LAMA_NTHREADS) {
        int pl = item / per_plane, i = item - pl * per_plane;
        int plane = plane0 + pl;
        int b = plane / p.C, c = plane - b * p.C;
        const auto src = ps + (long long)b * p.spec_bstride + (long long)(2 * c) * per_plane + i;
        P[pl * per_plane + i] = make_float2(src[0], src[per_plane]);
    }
    }
    __syncthreads();
    FFT_STAMP(1);
    // 2. Hermitian-symmetrise columns 0 and w/2 along h and pack them into column 0:
    //    G = D_h + i E_h,  D_h[u] = (D[u] + conj(D[-u]))/2  (so that ifft_h(G) = Re z0 + i Re z_{w/2})
    {
        float2 g = make_float2(0.f, 0.f);
        const bool act = tid < np * h;
        int pl = tid / h, u = tid - pl * h;
        if (act) {
            const float2* r0 = P + (pl * h + u) * wf;
            const float2* r1 = P + (pl * h + ((h - u) & (h - 1))) * wf;
            float2 d = r0[0], dm = r1[0], e = r0[wh], em = r1[wh];
            float2 dh = make_float2(0.5f * (d.x + dm.x), 0.5f * (d.y - dm.y));
            float2 eh = make_float2(0.5f * (e.x + em.x), 0.5f * (e.y - em.y));
            g = make_float2(dh.x - eh.y, dh.y + eh.x);
        }
        __syncthreads();
        if (act) P[(pl * h + u) * wf] = g;
    }
    __syncthreads();
    FFT_STAMP(2);
    // 3. inverse column FFTs (length h) over columns 0..w/2-1
    float2* E1 = fft_lds<true>(P, Q, twh, h, np * wh, wf, 1, wh, per_plane);
    FFT_STAMP(3);
    float2* Z = (E1 == P) ? Q : P;  // row-pair buffer [pl][hh][RSW]
    // 4. build the Hermitian-extended row pairs: W[k] = Za[k] + i Zb[k], W[w-k] = conj(Za[k]) + i conj(Zb[k])
    for (int item = tid; item < np * hh * wh; item += LAMA_NTHREADS) {
        int fg = item % (np * hh), k = item / (np * hh);
        int pl = fg / hh, f = fg - pl * hh;
        const float2* sa = E1 + (pl * h + 2 * f) * wf;
        const float2* sb = sa + wf;
        float2* z = Z + fg * RSW;
        float2 za = sa[k], zb = sb[k];
        if (k == 0) {
            z[0] = make_float2(za.x, zb.x);
            z[wh] = make_float2(za.y, zb.y);
        } else {
            z[k] = make_float2(za.x - zb.y, za.y + zb.x);
            z[w - k] = make_float2(za.x + zb.y, zb.x - za.y);
        }
    }
    __syncthreads();
    FFT_STAMP(4);
    // 5. inverse row FFTs (length w)
    float2* O = E1;
    float2* E2 = fft_lds<true>(Z, O, tww, w, np * hh, 1, RSW, np * hh, 0);
    FFT_STAMP(5);
    // 6. store rows 2f (real part) and 2f+1 (imaginary part), fused residual add
    if constexpr (SEQ > 1) {
        const int b = plane0 / p.C, c = plane0 - b * p.C;
        const auto ybase = py + (long long)b * p.y_bstride + (long long)c * h * w;
#pragma unroll
        for (int it = 0; it < NRP; ++it) {
            const int item = tid + it * LAMA_NTHREADS;
            if (item < hh * (w >> 2)) {
                int f, q;
                rowpair_item(item, w >> 2, f, q);
                const float2* s = E2 + f * RSW + q * 4;
                float2 v0 = s[0], v1 = s[1], v2 = s[2], v3 = s[3];
                float4 ra = make_float4(v0.x * p.scale, v1.x * p.scale, v2.x * p.scale, v3.x * p.scale);
                float4 rb = make_float4(v0.y * p.scale, v1.y * p.scale, v2.y * p.scale, v3.y * p.scale);
                if (px) {
                    ra.x += rxa[it].x; ra.y += rxa[it].y; ra.z += rxa[it].z; ra.w += rxa[it].w;
                    rb.x += rxb[it].x; rb.y += rxb[it].y; rb.z += rxb[it].z; rb.w += rxb[it].w;
                }
                const auto d = ybase + (2 * f) * w + q * 4;
                fft_st4(d, ra);
                fft_st4(d + w, rb);
            }
        }
        if (sq + 1 < SEQ) __syncthreads();   // the next plane's spectrum overwrites the buffers
    } else {
        const int wq = w >> 2;
        const int pp = hh * wq;
        for (int item = tid; item < np * pp; item += LAMA_NTHREADS) {
            int pl = item / pp, f, q;
            rowpair_item(item - pl * pp, wq, f, q);
            int plane = plane0 + pl;
            int b = plane / p.C, c = plane - b * p.C;
            const float2* s = E2 + (pl * hh + f) * RSW + q * 4;
            float2 v0 = s[0], v1 = s[1], v2 = s[2], v3 = s[3];
            long long off = (long long)c * h * w + (2 * f) * w + q * 4;
            float4 ra = make_float4(v0.x * p.scale, v1.x * p.scale, v2.x * p.scale, v3.x * p.scale);
            float4 rb = make_float4(v0.y * p.scale, v1.y * p.scale, v2.y * p.scale, v3.y * p.scale);
            if (px) {
                const auto r = px + (long long)b * p.x_bstride + off;
                float4 xa = fft_ld4(r);
                float4 xb = fft_ld4(r + w);
                ra.x += xa.x; ra.y += xa.y; ra.z += xa.z; ra.w += xa.w;
                rb.x += xb.x; rb.y += xb.y; rb.z += xb.z; rb.w += xb.w;
            }
            const auto d = py + (long long)b * p.y_bstride + off;
            fft_st4(d, ra);
            fft_st4(d + w, rb);
        }
    }
    }   // planes of this workgroup
    FFT_STAMP(6);
}
#undef FFT_STAMP

// ------------------------------------------------------------------------------------------------
// 64 x 64 planes, ONE LDS buffer per workgroup (17.5 KB instead of 34 KB): every phase reads its operands into registers,
// meets at a barrier and writes its results back into the same buffer (Stockham passes, the pair untangle and the layout
// change between row pairs [32][65] and spectrum [64][33] alike).  The two-buffer kernels above fit four workgroups per CU,
// so the 1536 planes of the bottleneck run as 1.5 rounds of latency-bound phases; with one buffer seven fit (registers) and
// every plane of the launch is resident at once.
// ------------------------------------------------------------------------------------------------
#define IP_N 64
#define IP_WF 33
#define IP_RSW 65
#define IP_BUF (IP_N * IP_WF)          // float2 elements: 64 x 33 >= 32 x 65

// one radix-8 Stockham pass over 32 FFTs of length 64, in place: thread = (FFT f = tid % 32, butterfly j = tid / 32)
template <bool INV>
__device__ __forceinline__ void ip_pass(float2* buf, const float2* tw, int Ns, int estride, int fstride) {
    const int tid = threadIdx.x;
    const int f = tid & 31, j = tid >> 5;        // j in 0..7 (N / R = 8 butterflies per FFT)
    const int k = j & (Ns - 1);
    float2* s = buf + f * fstride;
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = s[(j + r * 8) * estride];
    if (Ns > 1) {
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], tw[r * k * (IP_N / (Ns * 8))]);
    }
    dft8<INV>(v);
    __syncthreads();
    const int j0 = (j - k) * 8 + k;
#pragma unroll
    for (int r = 0; r < 8; ++r) s[(j0 + r * Ns) * estride] = v[r];
    __syncthreads();
}

template <bool TR = false, bool HF = false>
__device__ __forceinline__ void rfft2_ip64_body(const FftParams& p, const int plane) {
    FFT_IO(p);
    if (p.prio) __builtin_amdgcn_s_setprio(3);
    constexpr int h = IP_N, w = IP_N, wf = IP_WF, hh = 32, wh = 32, RSW = IP_RSW;
    float2* tww = reinterpret_cast<float2*>(lama_smem);
    float2* P = tww + w;                       // h == w: one twiddle table
    const int tid = threadIdx.x;
    const int b = plane / p.C, c = plane - b * p.C;
    // 1. row pairs straight from HBM (requested before the twiddles are computed): P[f][n] = (x[2f][n], x[2f+1][n])
    float4 ra[2], rb[2];
    const auto xin = px + (long long)b * p.x_bstride + (long long)c * h * w;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        int f, q;
        rowpair_item(tid + it * LAMA_NTHREADS, w >> 2, f, q);
        ra[it] = fft_ld4(xin + (2 * f) * w + q * 4);
        rb[it] = fft_ld4(xin + (2 * f + 1) * w + q * 4);
    }
    fft_init_twiddles<false>(tww, w);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        int f, q;
        rowpair_item(tid + it * LAMA_NTHREADS, w >> 2, f, q);
        float2* d = P + f * RSW + q * 4;
        d[0] = make_float2(ra[it].x, rb[it].x);
        d[1] = make_float2(ra[it].y, rb[it].y);
        d[2] = make_float2(ra[it].z, rb[it].z);
        d[3] = make_float2(ra[it].w, rb[it].w);
    }
    __syncthreads();
    // 2. row FFTs of the 32 packed row pairs (FFT f = row pair, element stride 1, FFT stride RSW)
    ip_pass<false>(P, tww, 1, 1, RSW);
    ip_pass<false>(P, tww, 8, 1, RSW);
    // 3. untangle the pairs into the half spectra of the two rows, row-pair layout [32][65] -> spectrum layout [64][33]
    {
        float2 oa[4], ob[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = tid + it * LAMA_NTHREADS;
            const int f = item & 31, k = item >> 5;          // k in 0..31
            const float2* z = P + f * RSW;
            if (k == 0) {
                const float2 z0 = z[0], zn = z[wh];
                oa[it] = make_float2(z0.x, zn.x);
                ob[it] = make_float2(z0.y, zn.y);
            } else {
                const float2 zk = z[k], zm = z[w - k];
                oa[it] = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                ob[it] = make_float2(0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x));
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = tid + it * LAMA_NTHREADS;
            const int f = item & 31, k = item >> 5;
            P[(2 * f) * wf + k] = oa[it];
            P[(2 * f + 1) * wf + k] = ob[it];
        }
        __syncthreads();
    }
    // 4. column FFTs over columns 0..31 (column 0 packs DC + Nyquist): FFT f = column, element stride wf, FFT stride 1
    ip_pass<false>(P, tww, 1, wf, 1);
    ip_pass<false>(P, tww, 8, wf, 1);
    // 5 + 6. float4 stores of the Re / Im planes; DC (col 0) and Nyquist (col 32) untangled from the packed column 0 on the fly
    {
        const int per_plane = h * wf;
        const auto dre = ps + (long long)b * p.spec_bstride + (long long)(2 * c) * per_plane;
        for (int i4 = tid; i4 < per_plane / 4; i4 += LAMA_NTHREADS) {
            float re[4], im[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i4 * 4 + e;
                const int k = i / wf, col = i - k * wf;
                float2 v = P[col == wh ? k * wf : i];
                if (col == 0 || col == wh) {
                    const float2 cc = P[k * wf], cm = P[((h - k) & (h - 1)) * wf];
                    v = col == 0 ? make_float2(0.5f * (cc.x + cm.x), 0.5f * (cc.y - cm.y)) : make_float2(0.5f * (cc.y + cm.y), 0.5f * (cm.x - cc.x));
                }
                re[e] = v.x * p.scale;
                im[e] = v.y * p.scale;
            }
            fft_st4(dre + i4 * 4, make_float4(re[0], re[1], re[2], re[3]));
            fft_st4(dre + per_plane + i4 * 4, make_float4(im[0], im[1], im[2], im[3]));
        }
    }
}

template <bool TR = false, bool HF = false>
__global__ __launch_bounds__(LAMA_NTHREADS) void rfft2_ip64_kernel(FftParams p) {
    rfft2_ip64_body<TR, HF>(p, blockIdx.x);
}

// Round 4: rfft2 of layer l + 1 and the Winograd output transform of layer l in ONE launch.  Both are memory-bound and independent (the out
// transform's result is first read by the Winograd GEMM / global launch of layer l + 1, both behind this launch), and every rfft2 workgroup has a
// 5 us phase in LDS during which the chip's HBM is idle (all 1536 planes are resident and run load / transform / store in lock-step): the
// out transform's streaming workgroups (no LDS use, two per CU next to the six FFT workgroups) fill it.  On two streams the pair takes 28.6
// instead of 34.9 us (tools/overlap_probe.py); as one launch there is no cross-queue signal to pay.  Workgroups 0 .. nout - 1: out transform
// (grid-stride), nout .. nout + planes - 1: one plane each.
#include "wino_out_dev.inc"
__global__ __launch_bounds__(LAMA_NTHREADS) void rfft2_ip64_wino_out_kernel(FftParams p, WoParams q, int nout) {
    if ((int)blockIdx.x < nout) lama_wino_out_body(q, (long long)blockIdx.x * LAMA_NTHREADS + threadIdx.x, (long long)nout * LAMA_NTHREADS);
    else rfft2_ip64_body<false, false>(p, (int)blockIdx.x - nout);
}

template <bool TR = false, bool HF = false>
__global__ __launch_bounds__(LAMA_NTHREADS) void irfft2_ip64_kernel(FftParams p) {
    FFT_IO(p);
    if (p.prio) __builtin_amdgcn_s_setprio(3);
    constexpr int h = IP_N, w = IP_N, wf = IP_WF, hh = 32, wh = 32, RSW = IP_RSW;
    constexpr int per_plane = h * wf;
    float2* tww = reinterpret_cast<float2*>(lama_smem);
    float2* P = tww + w;
    const int tid = threadIdx.x;
    const int plane = blockIdx.x;
    const int b = plane / p.C, c = plane - b * p.C;
    // 1. the Re / Im planes (float4 loads, requested before the twiddles are computed) and the residual rows
    const auto sbase = ps + (long long)b * p.spec_bstride + (long long)(2 * c) * per_plane;
    float4 sre[3], sim[3];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int i4 = tid + it * LAMA_NTHREADS;
        if (i4 < per_plane / 4) {
            sre[it] = fft_ld4(sbase + i4 * 4);
            sim[it] = fft_ld4(sbase + per_plane + i4 * 4);
        }
    }
    float4 rxa[2], rxb[2];
    if (px) {
        const auto rbase = px + (long long)b * p.x_bstride + (long long)c * h * w;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            int f, q;
            rowpair_item(tid + it * LAMA_NTHREADS, w >> 2, f, q);
            rxa[it] = fft_ld4(rbase + (2 * f) * w + q * 4);
            rxb[it] = fft_ld4(rbase + (2 * f + 1) * w + q * 4);
        }
    }
    fft_init_twiddles<true>(tww, w);
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int i4 = tid + it * LAMA_NTHREADS;
        if (i4 < per_plane / 4) {
            float2* d = P + i4 * 4;
            d[0] = make_float2(sre[it].x, sim[it].x);
            d[1] = make_float2(sre[it].y, sim[it].y);
            d[2] = make_float2(sre[it].z, sim[it].z);
            d[3] = make_float2(sre[it].w, sim[it].w);
        }
    }
    __syncthreads();
    // 2. Hermitian-symmetrise columns 0 and w/2 along h and pack them into column 0 (see irfft2_lds_kernel)
    {
        float2 g = make_float2(0.f, 0.f);
        const bool act = tid < h;
        const int u = tid;
        if (act) {
            const float2* r0 = P + u * wf;
            const float2* r1 = P + ((h - u) & (h - 1)) * wf;
            const float2 d = r0[0], dm = r1[0], e = r0[wh], em = r1[wh];
            const float2 dh = make_float2(0.5f * (d.x + dm.x), 0.5f * (d.y - dm.y));
            const float2 eh = make_float2(0.5f * (e.x + em.x), 0.5f * (e.y - em.y));
            g = make_float2(dh.x - eh.y, dh.y + eh.x);
        }
        __syncthreads();
        if (act) P[u * wf] = g;
        __syncthreads();
    }
    // 3. inverse column FFTs over columns 0..31
    ip_pass<true>(P, tww, 1, wf, 1);
    ip_pass<true>(P, tww, 8, wf, 1);
    // 4. Hermitian-extended row pairs, spectrum layout [64][33] -> row-pair layout [32][65]
    {
        float2 o0[4], o1[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = tid + it * LAMA_NTHREADS;
            const int f = item & 31, k = item >> 5;
            const float2 za = P[(2 * f) * wf + k], zb = P[(2 * f + 1) * wf + k];
            if (k == 0) {
                o0[it] = make_float2(za.x, zb.x);
                o1[it] = make_float2(za.y, zb.y);
            } else {
                o0[it] = make_float2(za.x - zb.y, za.y + zb.x);
                o1[it] = make_float2(za.x + zb.y, zb.x - za.y);
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = tid + it * LAMA_NTHREADS;
            const int f = item & 31, k = item >> 5;
            float2* z = P + f * RSW;
            z[k] = o0[it];
            z[k == 0 ? wh : w - k] = o1[it];
        }
        __syncthreads();
    }
    // 5. inverse row FFTs
    ip_pass<true>(P, tww, 1, 1, RSW);
    ip_pass<true>(P, tww, 8, 1, RSW);
    // 6. store rows 2f (real part) and 2f+1 (imaginary part), fused residual add
    {
        const auto ybase = py + (long long)b * p.y_bstride + (long long)c * h * w;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            int f, q;
            rowpair_item(tid + it * LAMA_NTHREADS, w >> 2, f, q);
            const float2* s = P + f * RSW + q * 4;
            const float2 v0 = s[0], v1 = s[1], v2 = s[2], v3 = s[3];
            float4 oa = make_float4(v0.x * p.scale, v1.x * p.scale, v2.x * p.scale, v3.x * p.scale);
            float4 ob = make_float4(v0.y * p.scale, v1.y * p.scale, v2.y * p.scale, v3.y * p.scale);
            if (px) {
                oa.x += rxa[it].x; oa.y += rxa[it].y; oa.z += rxa[it].z; oa.w += rxa[it].w;
                ob.x += rxb[it].x; ob.y += rxb[it].y; ob.z += rxb[it].z; ob.w += rxb[it].w;
            }
            const auto d = ybase + (2 * f) * w + q * 4;
            fft_st4(d, oa);
            fft_st4(d + w, ob);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same one-buffer construction for N x N planes with more than one item per thread and phase (N = 128: the bottleneck planes
// of 1024 x 1024 inputs; passes radix 8, 8, 2).  LDS per workgroup: 67.6 KB instead of 133 KB -> two workgroups per CU.
// ------------------------------------------------------------------------------------------------
template <int R, int N, int NF, bool INV>
__device__ __forceinline__ void ipn_pass(float2* buf, const float2* tw, int Ns, int estride, int fstride) {
    constexpr int NB = N / R, ITEMS = NF * NB, IT = ITEMS / LAMA_NTHREADS;
    static_assert(ITEMS % LAMA_NTHREADS == 0, "items per pass must fill the workgroup evenly");
    const int tid = threadIdx.x;
    float2 v[IT][R];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int item = tid + it * LAMA_NTHREADS;
        const int f = item % NF, j = item / NF;
        const int k = j & (Ns - 1);
        const float2* s = buf + f * fstride;
#pragma unroll
        for (int r = 0; r < R; ++r) v[it][r] = s[(j + r * NB) * estride];
        if (Ns > 1) {
#pragma unroll
            for (int r = 1; r < R; ++r) v[it][r] = cmul(v[it][r], tw[r * k * (N / (Ns * R))]);
        }
        if constexpr (R == 8) dft8<INV>(v[it]);
        else if constexpr (R == 4) dft4<INV>(v[it][0], v[it][1], v[it][2], v[it][3]);
        else dft2<INV>(v[it][0], v[it][1]);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int item = tid + it * LAMA_NTHREADS;
        const int f = item % NF, j = item / NF;
        const int k = j & (Ns - 1);
        float2* s = buf + f * fstride;
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) s[(j0 + r * Ns) * estride] = v[it][r];
    }
    __syncthreads();
}

template <int N, bool INV>
__device__ __forceinline__ void ipn_fft(float2* buf, const float2* tw, int estride, int fstride) {
    static_assert(N == 128, "pass list");
    ipn_pass<8, N, N / 2, INV>(buf, tw, 1, estride, fstride);
    ipn_pass<8, N, N / 2, INV>(buf, tw, 8, estride, fstride);
    ipn_pass<2, N, N / 2, INV>(buf, tw, 64, estride, fstride);
}

template <int N, bool HF = false>
__global__ __launch_bounds__(LAMA_NTHREADS) void rfft2_ipn_kernel(FftParams p) {
    FFT_IO(p);
    if (p.prio) __builtin_amdgcn_s_setprio(3);
    constexpr int h = N, w = N, wf = N / 2 + 1, hh = N / 2, wh = N / 2, RSW = N + 1;
    constexpr int NLD = hh * (w / 4) / LAMA_NTHREADS;          // row-pair float4 items per thread
    constexpr int NUT = hh * wh / LAMA_NTHREADS;               // untangle items per thread
    float2* tww = reinterpret_cast<float2*>(lama_smem);
    float2* P = tww + w;                                       // h == w: one twiddle table
    const int tid = threadIdx.x;
    const int plane = blockIdx.x;
    const int b = plane / p.C, c = plane - b * p.C;
    const auto xin = px + (long long)b * p.x_bstride + (long long)c * h * w;
    {   // 1. row pairs: P[f][n] = (x[2f][n], x[2f+1][n]), in two halves (registers)
        float4 ra[NLD / 2], rb[NLD / 2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int it = 0; it < NLD / 2; ++it) {
                int f, q;
                rowpair_item(tid + (half * (NLD / 2) + it) * LAMA_NTHREADS, w >> 2, f, q);
                ra[it] = fft_ld4(xin + (2 * f) * w + q * 4);
                rb[it] = fft_ld4(xin + (2 * f + 1) * w + q * 4);
            }
            if (half == 0) fft_init_twiddles<false>(tww, w);
#pragma unroll
            for (int it = 0; it < NLD / 2; ++it) {
                int f, q;
                rowpair_item(tid + (half * (NLD / 2) + it) * LAMA_NTHREADS, w >> 2, f, q);
                float2* d = P + f * RSW + q * 4;
                d[0] = make_float2(ra[it].x, rb[it].x);
                d[1] = make_float2(ra[it].y, rb[it].y);
                d[2] = make_float2(ra[it].z, rb[it].z);
                d[3] = make_float2(ra[it].w, rb[it].w);
            }
        }
    }
    __syncthreads();
    // 2. row FFTs of the packed row pairs
    ipn_fft<N, false>(P, tww, 1, RSW);
    // 3. untangle: row-pair layout [N/2][N+1] -> spectrum layout [N][N/2+1], through registers
    {
        float2 oa[NUT], ob[NUT];
#pragma unroll
        for (int it = 0; it < NUT; ++it) {
            const int item = tid + it * LAMA_NTHREADS;
            const int f = item % hh, k = item / hh;
            const float2* z = P + f * RSW;
            if (k == 0) {
                const float2 z0 = z[0], zn = z[wh];
                oa[it] = make_float2(z0.x, zn.x);
                ob[it] = make_float2(z0.y, zn.y);
            } else {
                const float2 zk = z[k], zm = z[w - k];
                oa[it] = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
                ob[it] = make_float2(0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x));
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NUT; ++it) {
            const int item = tid + it * LAMA_NTHREADS;
            const int f = item % hh, k = item / hh;
            P[(2 * f) * wf + k] = oa[it];
            P[(2 * f + 1) * wf + k] = ob[it];
        }
        __syncthreads();
    }
    // 4. column FFTs over columns 0..N/2-1 (column 0 packs DC + Nyquist)
    ipn_fft<N, false>(P, tww, wf, 1);
    // 5 + 6. float4 stores of the Re / Im planes; DC and Nyquist untangled from the packed column 0 on the fly
    {
        constexpr int per_plane = h * wf;
        const auto dre = ps + (long long)b * p.spec_bstride + (long long)(2 * c) * per_plane;
        for (int i4 = tid; i4 < per_plane / 4; i4 += LAMA_NTHREADS) {
            float re[4], im[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i4 * 4 + e;
                const int k = i / wf, col = i - k * wf;
                float2 v = P[col == wh ? k * wf : i];
                if (col == 0 || col == wh) {
                    const float2 cc = P[k * wf], cm = P[((h - k) & (h - 1)) * wf];
                    v = col == 0 ? make_float2(0.5f * (cc.x + cm.x), 0.5f * (cc.y - cm.y)) : make_float2(0.5f * (cc.y + cm.y), 0.5f * (cm.x - cc.x));
                }
                re[e] = v.x * p.scale;
                im[e] = v.y * p.scale;
            }
            fft_st4(dre + i4 * 4, make_float4(re[0], re[1], re[2], re[3]));
            fft_st4(dre + per_plane + i4 * 4, make_float4(im[0], im[1], im[2], im[3]));
        }
    }
}

template <int N, bool HF = false>
__global__ __launch_bounds__(LAMA_NTHREADS) void irfft2_ipn_kernel(FftParams p) {
    FFT_IO(p);
    if (p.prio) __builtin_amdgcn_s_setprio(3);
    constexpr int h = N, w = N, wf = N / 2 + 1, hh = N / 2, wh = N / 2, RSW = N + 1;
    constexpr int per_plane = h * wf;
    constexpr int NUT = hh * wh / LAMA_NTHREADS;
    constexpr int NST = hh * (w / 4) / LAMA_NTHREADS;
    float2* tww = reinterpret_cast<float2*>(lama_smem);
    float2* P = tww + w;
    const int tid = threadIdx.x;
    const int plane = blockIdx.x;
    const int b = plane / p.C, c = plane - b * p.C;
    // 1. the Re / Im planes
    const auto sbase = ps + (long long)b * p.spec_bstride + (long long)(2 * c) * per_plane;
    fft_init_twiddles<true>(tww, w);
    for (int i4 = tid; i4 < per_plane / 4; i4 += LAMA_NTHREADS) {
        const float4 sre = fft_ld4(sbase + i4 * 4);
        const float4 sim = fft_ld4(sbase + per_plane + i4 * 4);
        float2* d = P + i4 * 4;
        d[0] = make_float2(sre.x, sim.x);
        d[1] = make_float2(sre.y, sim.y);
        d[2] = make_float2(sre.z, sim.z);
        d[3] = make_float2(sre.w, sim.w);
    }
    __syncthreads();
    // 2. Hermitian-symmetrise columns 0 and w/2 along h and pack them into column 0 (see irfft2_lds_kernel)
    {
        float2 g = make_float2(0.f, 0.f);
        const bool act = tid < h;
        const int u = tid;
        if (act) {
            const float2* r0 = P + u * wf;
            const float2* r1 = P + ((h - u) & (h - 1)) * wf;
            const float2 d = r0[0], dm = r1[0], e = r0[wh], em = r1[wh];
            const float2 dh = make_float2(0.5f * (d.x + dm.x), 0.5f * (d.y - dm.y));
            const float2 eh = make_float2(0.5f * (e.x + em.x), 0.5f * (e.y - em.y));
            g = make_float2(dh.x - eh.y, dh.y + eh.x);
        }
        __syncthreads();
        if (act) P[u * wf] = g;
        __syncthreads();
    }
    // 3. inverse column FFTs over columns 0..N/2-1
    ipn_fft<N, true>(P, tww, wf, 1);
    // 4. Hermitian-extended row pairs, spectrum layout -> row-pair layout, through registers
    {
        float2 o0[NUT], o1[NUT];
#pragma unroll
        for (int it = 0; it < NUT; ++it) {
            const int item = tid + it * LAMA_NTHREADS;
            const int f = item % hh, k = item / hh;
            const float2 za = P[(2 * f) * wf + k], zb = P[(2 * f + 1) * wf + k];
            if (k == 0) {
                o0[it] = make_float2(za.x, zb.x);
                o1[it] = make_float2(za.y, zb.y);
            } else {
                o0[it] = make_float2(za.x - zb.y, za.y + zb.x);
                o1[it] = make_float2(za.x + zb.y, zb.x - za.y);
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NUT; ++it) {
            const int item = tid + it * LAMA_NTHREADS;
            const int f = item % hh, k = item / hh;
            float2* z = P + f * RSW;
            z[k] = o0[it];
            z[k == 0 ? wh : w - k] = o1[it];
        }
        __syncthreads();
    }
    // 5. inverse row FFTs
    ipn_fft<N, true>(P, tww, 1, RSW);
    // 6. store rows 2f (real part) and 2f+1 (imaginary part), fused residual add
    {
        const auto ybase = py + (long long)b * p.y_bstride + (long long)c * h * w;
        const auto rbase = px + (long long)b * p.x_bstride + (long long)c * h * w;
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            int f, q;
            rowpair_item(tid + it * LAMA_NTHREADS, w >> 2, f, q);
            const float2* s = P + f * RSW + q * 4;
            const float2 v0 = s[0], v1 = s[1], v2 = s[2], v3 = s[3];
            float4 oa = make_float4(v0.x * p.scale, v1.x * p.scale, v2.x * p.scale, v3.x * p.scale);
            float4 ob = make_float4(v0.y * p.scale, v1.y * p.scale, v2.y * p.scale, v3.y * p.scale);
            if (px) {
                const float4 xa = fft_ld4(rbase + (2 * f) * w + q * 4);
                const float4 xb = fft_ld4(rbase + (2 * f + 1) * w + q * 4);
                oa.x += xa.x; oa.y += xa.y; oa.z += xa.z; oa.w += xa.w;
                ob.x += xb.x; ob.y += xb.y; ob.z += xb.z; ob.w += xb.w;
            }
            const auto d = ybase + (2 * f) * w + q * 4;
            fft_st4(d, oa);
            fft_st4(d + w, ob);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// generic path (any h, w that the kernels above do not take): separable DFT through a float2 workspace ws[plane][h][wf].
// A length n = n1 * n2 (the host picks the divisor pair closest to sqrt(n); a prime has n1 = 1) is transformed in TWO direct stages
// (one Cooley-Tukey split, input j = j1 n2 + j2, output k = k1 + n1 k2):
//     T[k1][j2] = W_n^{j2 k1} * sum_{j1 < n1} x[j1 n2 + j2] W_n1^{j1 k1}        (n1 terms)
//     X[k1 + n1 k2] = sum_{j2 < n2} T[k1][j2] W_n2^{j2 k2}                       (n2 terms)
// i.e. n (n1 + n2) complex MACs per transform instead of n^2: 240 = 15 x 16 -> 7.7x fewer, 135 = 9 x 15 -> 5.6x, 188 = 4 x 47 ->
// 3.7x (a 1080 x 1920 photo has 135 x 240 planes at the bottleneck).  All W are read from ONE table tw[m] = e^{-+2 pi i m / n}.
// ------------------------------------------------------------------------------------------------
#define DFT_ROWS_PER_WG 8
#define DFT_COLS_PER_WG 8

__device__ __forceinline__ float2 dft_cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// a / d for 0 <= a < 2^22 by a float reciprocal + one correction step each way (the runtime integer division of the index
// arithmetic cost as much as a whole 15-term stage)
__device__ __forceinline__ int dft_div(int a, int d, float invd) {
    int q = (int)((float)a * invd);
    if (q * d > a) --q;
    if ((q + 1) * d <= a) ++q;
    return q;
}
// LDS pitch of one T row (n2 elements of 8 bytes): odd, so that the stage-2 reads of neighbouring outputs (k1 = k mod n1 moves
// fastest) do not all start in the same bank
__device__ __forceinline__ int dft_pitch(int n2) { return n2 | 1; }

// forward rows: ws[plane][y][k] = sum_n x[y][n] e^{-2 pi i n k / w}, k <= w / 2
template <bool HF = false>
__global__ __launch_bounds__(LAMA_NTHREADS) void dft_rows_fwd_kernel(FftParams p, float2* ws) {
    FFT_IO(p);
    const int w = p.w, wf = p.wf, h = p.h, n1 = p.w1, n2 = p.w2, PT = dft_pitch(n2), TR = n1 * PT;
    const float iw = 1.0f / (float)w, in1 = 1.0f / (float)n1, in2 = 1.0f / (float)n2, iwf = 1.0f / (float)wf;
    float2* tw = reinterpret_cast<float2*>(lama_smem);
    float2* T = tw + w;                                               // [ROWS][n1][PT]
    float* rows = reinterpret_cast<float*>(T + DFT_ROWS_PER_WG * TR);  // [ROWS][w]
    const int tid = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * DFT_ROWS_PER_WG;
    const long long nrows = (long long)p.nplanes * h;
    fft_init_twiddles<false>(tw, w);
    for (int i = tid; i < DFT_ROWS_PER_WG * w; i += LAMA_NTHREADS) {
        int r = dft_div(i, w, iw), n = i - r * w;
        long long row = row0 + r;
        float v = 0.f;
        if (row < nrows) {
            int plane = (int)(row / h), y = (int)(row - (long long)plane * h);
            int b = plane / p.C, c = plane - b * p.C;
            v = px[(long long)b * p.x_bstride + ((long long)c * h + y) * w + n];
        }
        rows[i] = v;
    }
    __syncthreads();
    for (int i = tid; i < DFT_ROWS_PER_WG * w; i += LAMA_NTHREADS) {   // stage 1 (real input)
        const int r = dft_div(i, w, iw), rem = i - r * w;
        const int k1 = dft_div(rem, n2, in2), j2 = rem - k1 * n2;
        const float* xr = rows + r * w + j2;
        const int step = k1 * n2;                                      // < w
        float ar = 0.f, ai = 0.f;
        int idx = 0;
        for (int j1 = 0; j1 < n1; ++j1) {
            const float2 t = tw[idx];
            const float v = xr[j1 * n2];
            ar += v * t.x;
            ai += v * t.y;
            idx += step;
            if (idx >= w) idx -= w;
        }
        T[r * TR + k1 * PT + j2] = dft_cmul(make_float2(ar, ai), tw[j2 * k1]);   // j2 k1 < n2 n1 = w
    }
    __syncthreads();
    for (int i = tid; i < DFT_ROWS_PER_WG * wf; i += LAMA_NTHREADS) {  // stage 2, the non-redundant half of the spectrum
        int r = dft_div(i, wf, iwf), k = i - r * wf;
        long long row = row0 + r;
        if (row >= nrows) continue;
        const int k2 = dft_div(k, n1, in1), k1 = k - k2 * n1;
        const float2* tr = T + r * TR + k1 * PT;
        const int step = k2 * n1;                                      // < w
        float ar = 0.f, ai = 0.f;
        int idx = 0;
        for (int j2 = 0; j2 < n2; ++j2) {
            const float2 t = tw[idx], v = tr[j2];
            ar += v.x * t.x - v.y * t.y;
            ai += v.x * t.y + v.y * t.x;
            idx += step;
            if (idx >= w) idx -= w;
        }
        ws[row * wf + k] = make_float2(ar, ai);
    }
}

// forward columns: spec[u][k] = scale * sum_y ws[y][k] e^{-2 pi i y u / h}; inverse columns (INV):
// ws_out[y][k] = sum_u spec[u][k] e^{+2 pi i u y / h}
template <bool INV, bool HF = false>
__global__ __launch_bounds__(LAMA_NTHREADS) void dft_cols_kernel(FftParams p, float2* ws) {
    FFT_IO(p);
    const int wf = p.wf, h = p.h, n1 = p.h1, n2 = p.h2;
    float2* tw = reinterpret_cast<float2*>(lama_smem);
    float2* cols = tw + h;                           // [h][DFT_COLS_PER_WG]
    float2* T = cols + h * DFT_COLS_PER_WG;          // [n1][n2][DFT_COLS_PER_WG]
    const int tid = threadIdx.x;
    const int ncb = (wf + DFT_COLS_PER_WG - 1) / DFT_COLS_PER_WG;
    const int plane = blockIdx.x / ncb, k0 = (blockIdx.x - plane * ncb) * DFT_COLS_PER_WG;
    const int b = plane / p.C, c = plane - b * p.C;
    const long long per_plane = (long long)h * wf;
    const auto sre = ps + (long long)b * p.spec_bstride + (long long)(2 * c) * per_plane;
    const auto sim = sre + per_plane;
    fft_init_twiddles<INV>(tw, h);
    for (int i = tid; i < h * DFT_COLS_PER_WG; i += LAMA_NTHREADS) {
        int y = i / DFT_COLS_PER_WG, kk = i - y * DFT_COLS_PER_WG;
        int k = k0 + kk;
        float2 v = make_float2(0.f, 0.f);
        if (k < wf) v = INV ? make_float2(sre[(long long)y * wf + k], sim[(long long)y * wf + k]) : ws[(long long)plane * per_plane + (long long)y * wf + k];
        cols[i] = v;
    }
    __syncthreads();
    const float in2 = 1.0f / (float)n2;
    for (int i = tid; i < h * DFT_COLS_PER_WG; i += LAMA_NTHREADS) {   // stage 1: T[u1][j2] (element index u1 n2 + j2)
        const int e = i / DFT_COLS_PER_WG, kk = i - e * DFT_COLS_PER_WG;
        const int u1 = dft_div(e, n2, in2), j2 = e - u1 * n2;
        const int step = u1 * n2;                                      // < h
        float ar = 0.f, ai = 0.f;
        int idx = 0;
        for (int j1 = 0; j1 < n1; ++j1) {
            const float2 t = tw[idx], v = cols[(j1 * n2 + j2) * DFT_COLS_PER_WG + kk];
            ar += v.x * t.x - v.y * t.y;
            ai += v.x * t.y + v.y * t.x;
            idx += step;
            if (idx >= h) idx -= h;
        }
        T[i] = dft_cmul(make_float2(ar, ai), tw[j2 * u1]);
    }
    __syncthreads();
    // stage 2: output u = u1 + n1 u2, enumerated with u2 moving fastest: the eight-column groups of neighbouring threads then read
    // the SAME T row (an LDS broadcast) instead of rows n2 * 64 bytes apart (same banks)
    for (int i = tid; i < h * DFT_COLS_PER_WG; i += LAMA_NTHREADS) {
        const int e = i / DFT_COLS_PER_WG, kk = i - e * DFT_COLS_PER_WG;
        int k = k0 + kk;
        if (k >= wf) continue;
        const int u1 = dft_div(e, n2, in2), u2 = e - u1 * n2;
        const int u = u1 + n1 * u2;
        const int step = u2 * n1;                                      // < h
        float ar = 0.f, ai = 0.f;
        int idx = 0;
        for (int j2 = 0; j2 < n2; ++j2) {
            const float2 t = tw[idx], v = T[(u1 * n2 + j2) * DFT_COLS_PER_WG + kk];
            ar += v.x * t.x - v.y * t.y;
            ai += v.x * t.y + v.y * t.x;
            idx += step;
            if (idx >= h) idx -= h;
        }
        if (INV) {
            ws[(long long)plane * per_plane + (long long)u * wf + k] = make_float2(ar, ai);
        } else {
            sre[(long long)u * wf + k] = ar * p.scale;
            sim[(long long)u * wf + k] = ai * p.scale;
        }
    }
}

// inverse rows (c2r ignoring Im of bins 0 and w/2, like torch.fft.irfftn): the half spectrum is extended to its Hermitian whole
// Z[k], k < w, in LDS and transformed by the same two stages; only the real part of the second stage is formed.  + resid.
template <bool HF = false>
__global__ __launch_bounds__(LAMA_NTHREADS) void dft_rows_inv_kernel(FftParams p, const float2* ws) {
    FFT_IO(p);
    const int w = p.w, wf = p.wf, h = p.h, n1 = p.w1, n2 = p.w2, PT = dft_pitch(n2), TR = n1 * PT;
    const float iw = 1.0f / (float)w, in1 = 1.0f / (float)n1, in2 = 1.0f / (float)n2;
    float2* tw = reinterpret_cast<float2*>(lama_smem);
    float2* Z = tw + w;                         // [ROWS][w]
    float2* T = Z + DFT_ROWS_PER_WG * w;        // [ROWS][n1][PT]
    const int tid = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * DFT_ROWS_PER_WG;
    const long long nrows = (long long)p.nplanes * h;
    fft_init_twiddles<true>(tw, w);
    for (int i = tid; i < DFT_ROWS_PER_WG * w; i += LAMA_NTHREADS) {
        const int r = dft_div(i, w, iw), k = i - r * w;
        const long long row = row0 + r;
        float2 v = make_float2(0.f, 0.f);
        if (row < nrows) {
            if (k < wf) {
                v = ws[row * wf + k];
                if (k == 0 || 2 * k == w) v.y = 0.f;
            } else {
                v = ws[row * wf + (w - k)];
                v.y = -v.y;
            }
        }
        Z[i] = v;
    }
    __syncthreads();
    for (int i = tid; i < DFT_ROWS_PER_WG * w; i += LAMA_NTHREADS) {   // stage 1
        const int r = dft_div(i, w, iw), rem = i - r * w;
        const int k1 = dft_div(rem, n2, in2), j2 = rem - k1 * n2;
        const float2* zr = Z + r * w + j2;
        const int step = k1 * n2;
        float ar = 0.f, ai = 0.f;
        int idx = 0;
        for (int j1 = 0; j1 < n1; ++j1) {
            const float2 t = tw[idx], v = zr[j1 * n2];
            ar += v.x * t.x - v.y * t.y;
            ai += v.x * t.y + v.y * t.x;
            idx += step;
            if (idx >= w) idx -= w;
        }
        T[r * TR + k1 * PT + j2] = dft_cmul(make_float2(ar, ai), tw[j2 * k1]);
    }
    __syncthreads();
    for (int i = tid; i < DFT_ROWS_PER_WG * w; i += LAMA_NTHREADS) {   // stage 2: real part of output xx = k1 + n1 k2
        int r = dft_div(i, w, iw), xx = i - r * w;
        long long row = row0 + r;
        if (row >= nrows) continue;
        const int k2 = dft_div(xx, n1, in1), k1 = xx - k2 * n1;
        const float2* tr = T + r * TR + k1 * PT;
        const int step = k2 * n1;
        float acc = 0.f;
        int idx = 0;
        for (int j2 = 0; j2 < n2; ++j2) {
            const float2 t = tw[idx], v = tr[j2];
            acc += v.x * t.x - v.y * t.y;
            idx += step;
            if (idx >= w) idx -= w;
        }
        acc *= p.scale;
        int plane = (int)(row / h), y = (int)(row - (long long)plane * h);
        int b = plane / p.C, c = plane - b * p.C;
        long long off = ((long long)c * h + y) * w + xx;
        if (px) acc += px[(long long)b * p.x_bstride + off];
        py[(long long)b * p.y_bstride + off] = acc;
    }
}


// ------------------------------------------------------------------------------------------------
// two-pass path: power-of-two planes larger than one workgroup's LDS (256 x 256 ... 1024 x 1024, or one long side):
// row FFTs and column FFTs as two launches through the complex workspace ws[plane][h][wf].  Same butterflies (fft_lds) and
// the same row-pair packing as the one-pass kernels; the DC / Nyquist columns are simply transformed like every other
// column (wf = w/2 + 1 column FFTs instead of w/2).
// ------------------------------------------------------------------------------------------------
#define FFT2P_PAIRS 8  // row pairs per workgroup (pass R)
#define FFT2P_COLS 8    // columns per workgroup (pass C)

// forward rows: ws[plane][y][k] = half spectrum of row y (unscaled)
template <bool HF = false>
__global__ __launch_bounds__(LAMA_NTHREADS) void fft2p_rows_fwd_kernel(FftParams p, float2* ws) {
    FFT_IO(p);
    const int h = p.h, w = p.w, wf = p.wf, hh = h >> 1, wh = w >> 1;
    const int RSW = w + 1;
    float2* tw = reinterpret_cast<float2*>(lama_smem);
    float2* P = tw + w;
    float2* Q = P + FFT2P_PAIRS * RSW;
    const int tid = threadIdx.x;
    const int gpp = lama_ceil_div_dev(hh, FFT2P_PAIRS);          // pair groups per plane
    const int plane = blockIdx.x / gpp, f0 = (blockIdx.x - plane * gpp) * FFT2P_PAIRS;
    const int np = (hh - f0) < FFT2P_PAIRS ? (hh - f0) : FFT2P_PAIRS;
    const int b = plane / p.C, c = plane - b * p.C;
    fft_init_twiddles<false>(tw, w);
    const auto src = px + (long long)b * p.x_bstride + (long long)c * h * w;
    for (int i = tid; i < np * w; i += LAMA_NTHREADS) {
        int f = i / w, n = i - f * w;
        P[f * RSW + n] = make_float2(src[(long long)(2 * (f0 + f)) * w + n], src[(long long)(2 * (f0 + f) + 1) * w + n]);
    }
    __syncthreads();
    float2* E = fft_lds<false>(P, Q, tw, w, np, 1, RSW, np, 0);
    float2* out = ws + (long long)plane * h * wf;
    for (int i = tid; i < np * wf; i += LAMA_NTHREADS) {
        int f = i / wf, k = i - f * wf;
        const float2* z = E + f * RSW;
        float2 za, zb;
        if (k == 0) { za = make_float2(z[0].x, 0.f); zb = make_float2(z[0].y, 0.f); }
        else if (k == wh) { za = make_float2(z[wh].x, 0.f); zb = make_float2(z[wh].y, 0.f); }
        else {
            float2 zk = z[k], zm = z[w - k];
            za = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
            zb = make_float2(0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x));
        }
        out[(long long)(2 * (f0 + f)) * wf + k] = za;
        out[(long long)(2 * (f0 + f) + 1) * wf + k] = zb;
    }
}

// columns: forward (INV = false): spec[u][k] = scale * FFT_h(ws[.][k]); inverse: ws[y][k] = IFFT_h(spec[.][k]) (unscaled)
template <bool INV, bool HF = false>
__global__ __launch_bounds__(LAMA_NTHREADS) void fft2p_cols_kernel(FftParams p, float2* ws) {
    FFT_IO(p);
    const int h = p.h, wf = p.wf;
    const int CS = h + 1;                                        // LDS pitch of one column
    float2* tw = reinterpret_cast<float2*>(lama_smem);
    float2* P = tw + h;
    float2* Q = P + FFT2P_COLS * CS;
    const int tid = threadIdx.x;
    const int ncb = lama_ceil_div_dev(wf, FFT2P_COLS);
    const int plane = blockIdx.x / ncb, k0 = (blockIdx.x - plane * ncb) * FFT2P_COLS;
    const int nc = (wf - k0) < FFT2P_COLS ? (wf - k0) : FFT2P_COLS;
    const int b = plane / p.C, c = plane - b * p.C;
    const long long per_plane = (long long)h * wf;
    const auto sre = ps + (long long)b * p.spec_bstride + (long long)(2 * c) * per_plane;
    const auto sim = sre + per_plane;
    float2* wp = ws + (long long)plane * per_plane;
    fft_init_twiddles<INV>(tw, h);
    for (int i = tid; i < h * FFT2P_COLS; i += LAMA_NTHREADS) {
        int y = i / FFT2P_COLS, kk = i - y * FFT2P_COLS;
        if (kk < nc) {
            long long o = (long long)y * wf + k0 + kk;
            P[kk * CS + y] = INV ? make_float2(sre[o], sim[o]) : wp[o];
        }
    }
    __syncthreads();
    float2* E = fft_lds<INV>(P, Q, tw, h, nc, 1, CS, nc, 0);
    for (int i = tid; i < h * FFT2P_COLS; i += LAMA_NTHREADS) {
        int u = i / FFT2P_COLS, kk = i - u * FFT2P_COLS;
        if (kk < nc) {
            long long o = (long long)u * wf + k0 + kk;
            float2 v = E[kk * CS + u];
            if (INV) wp[o] = v;
            else { sre[o] = v.x * p.scale; sim[o] = v.y * p.scale; }
        }
    }
}

// inverse rows: c2r of ws rows (Im of bins 0 and w/2 ignored), two rows per complex FFT, fused residual add
template <bool HF = false>
__global__ __launch_bounds__(LAMA_NTHREADS) void fft2p_rows_inv_kernel(FftParams p, const float2* ws) {
    FFT_IO(p);
    const int h = p.h, w = p.w, wf = p.wf, hh = h >> 1, wh = w >> 1;
    const int RSW = w + 1;
    float2* tw = reinterpret_cast<float2*>(lama_smem);
    float2* P = tw + w;
    float2* Q = P + FFT2P_PAIRS * RSW;
    const int tid = threadIdx.x;
    const int gpp = lama_ceil_div_dev(hh, FFT2P_PAIRS);
    const int plane = blockIdx.x / gpp, f0 = (blockIdx.x - plane * gpp) * FFT2P_PAIRS;
    const int np = (hh - f0) < FFT2P_PAIRS ? (hh - f0) : FFT2P_PAIRS;
    const int b = plane / p.C, c = plane - b * p.C;
    fft_init_twiddles<true>(tw, w);
    const float2* in = ws + (long long)plane * h * wf;
    for (int i = tid; i < np * wf; i += LAMA_NTHREADS) {
        int f = i / wf, k = i - f * wf;
        float2 za = in[(long long)(2 * (f0 + f)) * wf + k], zb = in[(long long)(2 * (f0 + f) + 1) * wf + k];
        float2* z = P + f * RSW;
        if (k == 0 || k == wh) z[k] = make_float2(za.x, zb.x);
        else {
            z[k] = make_float2(za.x - zb.y, za.y + zb.x);
            z[w - k] = make_float2(za.x + zb.y, zb.x - za.y);
        }
    }
    __syncthreads();
    float2* E = fft_lds<true>(P, Q, tw, w, np, 1, RSW, np, 0);
    const long long base = (long long)c * h * w;
    for (int i = tid; i < np * w; i += LAMA_NTHREADS) {
        int f = i / w, n = i - f * w;
        float2 v = E[f * RSW + n];
        long long oa = base + (long long)(2 * (f0 + f)) * w + n, ob = oa + w;
        float ra = v.x * p.scale, rb = v.y * p.scale;
        if (px) {
            const auto r = px + (long long)b * p.x_bstride;
            ra += r[oa];
            rb += r[ob];
        }
        const auto d = py + (long long)b * p.y_bstride;
        d[oa] = ra;
        d[ob] = rb;
    }
}

// ------------------------------------------------------------------------------------------------
// Round 4: the two passes for 256 x 256 planes (the bottleneck planes of 2048 x 2048 inputs: BASELINE configs[4]) with compile-time sizes.
// The generic kernels above spend their time on run-time index arithmetic and 4-byte accesses (61 / 60 / 33 / 43 us per pass on 192 planes,
// 1.6 - 3 TB/s; 16 instead of 8 rows / columns per workgroup changes nothing: profiles/r04_fft256.txt).  Here: the one-buffer in-place passes of
// the 128 x 128 kernel (radix 8, 8, 4), 16-byte global accesses on the activation and the workspace side, and the workspace holds 128
// columns per row -- column 0 packs the DC and the Nyquist column (both real after the row pass; Hermitian-symmetrised before the inverse
// column pass, as in irfft2_ipn_kernel), so a row is 1024 bytes and a group of 16 columns is one aligned 128-byte segment per row.
//   rows:    workgroup = 8 row pairs of a plane (8 complex FFTs, 16.4 KB of LDS),   grid = planes x 16
//   columns: workgroup = 16 columns x 256 rows (16 FFTs, 34.8 KB),                   grid = planes x 8
// ------------------------------------------------------------------------------------------------
#define FQ_N 256
#define FQ_PAIRS 8
#ifndef FQ_CW
#define FQ_CW 32      // 16: 34.8 KB of LDS, four workgroups per CU, 64-byte segments of the spectrum planes; 32: 67.6 KB, two per CU, 128-byte segments
#endif
#define FQ_CP (FQ_CW + 1)

template <bool INV>
__device__ __forceinline__ void fq_fft_rows(float2* buf, const float2* tw) {      // FQ_PAIRS FFTs, element stride 1, FFT stride FQ_N + 1
    ipn_pass<8, FQ_N, FQ_PAIRS, INV>(buf, tw, 1, 1, FQ_N + 1);
    ipn_pass<8, FQ_N, FQ_PAIRS, INV>(buf, tw, 8, 1, FQ_N + 1);
    ipn_pass<4, FQ_N, FQ_PAIRS, INV>(buf, tw, 64, 1, FQ_N + 1);
}
template <bool INV>
__device__ __forceinline__ void fq_fft_cols(float2* buf, const float2* tw) {      // FQ_CW FFTs, element stride FQ_CP, FFT stride 1
    ipn_pass<8, FQ_N, FQ_CW, INV>(buf, tw, 1, FQ_CP, 1);
    ipn_pass<8, FQ_N, FQ_CW, INV>(buf, tw, 8, FQ_CP, 1);
    ipn_pass<4, FQ_N, FQ_CW, INV>(buf, tw, 64, FQ_CP, 1);
}

// forward rows: ws[plane][y][k], k = 0 .. 127 (k = 0: (DC, Nyquist) of row y), unscaled
__global__ __launch_bounds__(LAMA_NTHREADS) void fq_rows_fwd_kernel(FftParams p, float2* ws) {
    constexpr int N = FQ_N, wh = N / 2, RSW = N + 1, NP = FQ_PAIRS;
    float2* tw = reinterpret_cast<float2*>(lama_smem);
    float2* P = tw + N;
    const int tid = threadIdx.x;
    const int plane = blockIdx.x / (wh / NP), f0 = (blockIdx.x % (wh / NP)) * NP;
    const int b = plane / p.C, c = plane - b * p.C;
    const float* src = (const float*)p.x + (long long)b * p.x_bstride + (long long)c * N * N + (long long)(2 * f0) * N;
    float4 ra[2], rb[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {          // NP pairs x 64 float4 columns = 512 items
        const int item = tid + it * LAMA_NTHREADS, f = item >> 6, q = item & 63;
        ra[it] = *reinterpret_cast<const float4*>(src + (2 * f) * N + q * 4);
        rb[it] = *reinterpret_cast<const float4*>(src + (2 * f + 1) * N + q * 4);
    }
    fft_init_twiddles<false>(tw, N);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int item = tid + it * LAMA_NTHREADS, f = item >> 6, q = item & 63;
        float2* d = P + f * RSW + q * 4;
        d[0] = make_float2(ra[it].x, rb[it].x);
        d[1] = make_float2(ra[it].y, rb[it].y);
        d[2] = make_float2(ra[it].z, rb[it].z);
        d[3] = make_float2(ra[it].w, rb[it].w);
    }
    __syncthreads();
    fq_fft_rows<false>(P, tw);
    float2* out = ws + ((long long)plane * N + 2 * f0) * wh;
#pragma unroll
    for (int it = 0; it < NP * wh / LAMA_NTHREADS; ++it) {       // (pair, k) items, k fastest: 8-byte stores, 512 contiguous bytes per wave
        const int item = tid + it * LAMA_NTHREADS, f = item >> 7, k = item & (wh - 1);
        const float2* z = P + f * RSW;
        float2 za, zb;
        if (k == 0) {
            const float2 z0 = z[0], zn = z[wh];
            za = make_float2(z0.x, zn.x);
            zb = make_float2(z0.y, zn.y);
        } else {
            const float2 zk = z[k], zm = z[N - k];
            za = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
            zb = make_float2(0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x));
        }
        out[(2 * f) * wh + k] = za;
        out[(2 * f + 1) * wh + k] = zb;
    }
}

// columns.  forward: spec[u][k] = scale * FFT_h(ws[.][k]) (column 0 of ws untangled into the DC and the Nyquist column); inverse:
// ws[y][k] = IFFT_h(spec[.][k]), unscaled, columns 0 and N/2 of spec Hermitian-symmetrised along h and packed into column 0
template <bool INV>
__global__ __launch_bounds__(LAMA_NTHREADS) void fq_cols_kernel(FftParams p, float2* ws) {
    constexpr int N = FQ_N, wh = N / 2, wf = N / 2 + 1, CW = FQ_CW, CP = FQ_CP;
    constexpr int QW = CW / 2, QL = CW == 32 ? 4 : 3, CL = QL + 1;      // float4 (two columns) per row of the group, log2 of it, log2(CW)
    static_assert(CW == 16 || CW == 32, "column group");
    float2* tw = reinterpret_cast<float2*>(lama_smem);
    float2* P = tw + N;
    const int tid = threadIdx.x;
    const int plane = blockIdx.x / (wh / CW), k0 = (blockIdx.x % (wh / CW)) * CW;
    const int b = plane / p.C, c = plane - b * p.C;
    const long long per_plane = (long long)N * wf;
    float* sre = (float*)p.spec + (long long)b * p.spec_bstride + (long long)(2 * c) * per_plane;
    float* sim = sre + per_plane;
    float2* wp = ws + (long long)plane * N * wh;
    const float* mre = (!INV && p.mask) ? p.mask + (long long)b * p.mask_bstride + (long long)(2 * c) * per_plane : nullptr;
    const float* mim = mre ? mre + per_plane : nullptr;
    (void)mim;
    if constexpr (!INV) {
        // 256 rows x QW float4 (two columns each): one aligned 128 / 256-byte segment per row
        float4 v[QW];
#pragma unroll
        for (int it = 0; it < QW; ++it) {
            const int item = tid + it * LAMA_NTHREADS, y = item >> QL, q = item & (QW - 1);
            v[it] = *reinterpret_cast<const float4*>(wp + (long long)y * wh + k0 + 2 * q);
        }
        fft_init_twiddles<false>(tw, N);
#pragma unroll
        for (int it = 0; it < QW; ++it) {
            const int item = tid + it * LAMA_NTHREADS, y = item >> QL, q = item & (QW - 1);
            P[y * CP + 2 * q] = make_float2(v[it].x, v[it].y);
            P[y * CP + 2 * q + 1] = make_float2(v[it].z, v[it].w);
        }
        // the mask values of this thread's output items, requested before the transform (at the stores every load would wait for the store
        // before it -- the planes may alias for all the compiler knows: 56 instead of 32 us)
        float mkr[N * CW / LAMA_NTHREADS], mki[N * CW / LAMA_NTHREADS], mnr[N * CW / LAMA_NTHREADS], mni[N * CW / LAMA_NTHREADS];   // (kk is the same for all items of a thread)
        if (mre) {
#pragma unroll
            for (int it = 0; it < N * CW / LAMA_NTHREADS; ++it) {
                const int item = tid + it * LAMA_NTHREADS, u = item >> CL, kk = item & (CW - 1);
                mkr[it] = mre[(long long)u * wf + k0 + kk];
                mki[it] = mim[(long long)u * wf + k0 + kk];
                if (k0 + kk == 0) { mnr[it] = mre[(long long)u * wf + wh]; mni[it] = mim[(long long)u * wf + wh]; }
            }
        }
        __syncthreads();
        fq_fft_cols<false>(P, tw);
        // rows u, 16 columns each: 64-byte segments of the Re and of the Im plane (the layout lama_tensor fixes)
#pragma unroll
        for (int it = 0; it < N * CW / LAMA_NTHREADS; ++it) {
            const int item = tid + it * LAMA_NTHREADS, u = item >> CL, kk = item & (CW - 1);
            float2 v2 = P[u * CP + kk];
            if (k0 + kk == 0) {       // the packed column: DC = Hermitian part, Nyquist = anti-Hermitian part / i
                const float2 cc = v2, cm = P[((N - u) & (N - 1)) * CP];
                v2 = make_float2(0.5f * (cc.x + cm.x), 0.5f * (cc.y - cm.y));
                float2 ny = make_float2(0.5f * (cc.y + cm.y) * p.scale, 0.5f * (cm.x - cc.x) * p.scale);
                if (mre) {
                    ny.x = mnr[it] > 0.0f ? ny.x : 0.0f;
                    ny.y = mni[it] > 0.0f ? ny.y : 0.0f;
                }
                sre[(long long)u * wf + wh] = ny.x;
                sim[(long long)u * wf + wh] = ny.y;
            }
            float2 ov = make_float2(v2.x * p.scale, v2.y * p.scale);
            if (mre) {
                ov.x = mkr[it] > 0.0f ? ov.x : 0.0f;
                ov.y = mki[it] > 0.0f ? ov.y : 0.0f;
            }
            sre[(long long)u * wf + k0 + kk] = ov.x;
            sim[(long long)u * wf + k0 + kk] = ov.y;
        }
    } else {
        fft_init_twiddles<true>(tw, N);
#pragma unroll
        for (int it = 0; it < N * CW / LAMA_NTHREADS; ++it) {
            const int item = tid + it * LAMA_NTHREADS, u = item >> CL, kk = item & (CW - 1);
            const long long o = (long long)u * wf + k0 + kk;
            P[u * CP + kk] = make_float2(sre[o], sim[o]);
        }
        float2 g = make_float2(0.f, 0.f);
        if (k0 == 0) {                // workgroup-uniform
            __syncthreads();
            const int u = tid, um = (N - u) & (N - 1);                                  // 256 threads = 256 rows
            const float2 d = P[u * CP], dm = P[um * CP];
            const float2 e = make_float2(sre[(long long)u * wf + wh], sim[(long long)u * wf + wh]);
            const float2 em = make_float2(sre[(long long)um * wf + wh], sim[(long long)um * wf + wh]);
            const float2 dh = make_float2(0.5f * (d.x + dm.x), 0.5f * (d.y - dm.y));
            const float2 eh = make_float2(0.5f * (e.x + em.x), 0.5f * (e.y - em.y));
            g = make_float2(dh.x - eh.y, dh.y + eh.x);
            __syncthreads();
            P[u * CP] = g;
        }
        __syncthreads();
        fq_fft_cols<true>(P, tw);
#pragma unroll
        for (int it = 0; it < QW; ++it) {
            const int item = tid + it * LAMA_NTHREADS, y = item >> QL, q = item & (QW - 1);
            const float2 a = P[y * CP + 2 * q], bb = P[y * CP + 2 * q + 1];
            *reinterpret_cast<float4*>(wp + (long long)y * wh + k0 + 2 * q) = make_float4(a.x, a.y, bb.x, bb.y);
        }
    }
}

// inverse rows: c2r of the ws rows, two rows per complex FFT, scale, fused residual add
__global__ __launch_bounds__(LAMA_NTHREADS) void fq_rows_inv_kernel(FftParams p, const float2* ws) {
    constexpr int N = FQ_N, wh = N / 2, RSW = N + 1, NP = FQ_PAIRS;
    float2* tw = reinterpret_cast<float2*>(lama_smem);
    float2* P = tw + N;
    const int tid = threadIdx.x;
    const int plane = blockIdx.x / (wh / NP), f0 = (blockIdx.x % (wh / NP)) * NP;
    const int b = plane / p.C, c = plane - b * p.C;
    const float2* in = ws + ((long long)plane * N + 2 * f0) * wh;
    float2 za[NP * wh / LAMA_NTHREADS], zb[NP * wh / LAMA_NTHREADS];
#pragma unroll
    for (int it = 0; it < NP * wh / LAMA_NTHREADS; ++it) {
        const int item = tid + it * LAMA_NTHREADS, f = item >> 7, k = item & (wh - 1);
        za[it] = in[(2 * f) * wh + k];
        zb[it] = in[(2 * f + 1) * wh + k];
    }
    // the residual rows, requested behind the spectrum (used at the store)
    const long long obase = (long long)c * N * N + (long long)(2 * f0) * N;
    const float* rs = p.x ? (const float*)p.x + (long long)b * p.x_bstride + obase : nullptr;
    float4 xa[2], xb[2];
    if (rs) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = tid + it * LAMA_NTHREADS, f = item >> 6, q = item & 63;
            xa[it] = *reinterpret_cast<const float4*>(rs + (2 * f) * N + q * 4);
            xb[it] = *reinterpret_cast<const float4*>(rs + (2 * f + 1) * N + q * 4);
        }
    }
    float4 ma[2], mb[2];
    if (p.mask) {
        const float* mk = p.mask + (long long)b * p.mask_bstride + obase;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = tid + it * LAMA_NTHREADS, f = item >> 6, q = item & 63;
            ma[it] = *reinterpret_cast<const float4*>(mk + (2 * f) * N + q * 4);
            mb[it] = *reinterpret_cast<const float4*>(mk + (2 * f + 1) * N + q * 4);
        }
    }
    fft_init_twiddles<true>(tw, N);
#pragma unroll
    for (int it = 0; it < NP * wh / LAMA_NTHREADS; ++it) {
        const int item = tid + it * LAMA_NTHREADS, f = item >> 7, k = item & (wh - 1);
        float2* z = P + f * RSW;
        const float2 a = za[it], bb = zb[it];
        if (k == 0) {
            z[0] = make_float2(a.x, bb.x);
            z[wh] = make_float2(a.y, bb.y);
        } else {
            z[k] = make_float2(a.x - bb.y, a.y + bb.x);
            z[N - k] = make_float2(a.x + bb.y, bb.x - a.y);
        }
    }
    __syncthreads();
    fq_fft_rows<true>(P, tw);
    float* dst = (float*)p.y + (long long)b * p.y_bstride + obase;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int item = tid + it * LAMA_NTHREADS, f = item >> 6, q = item & 63;
        const float2* s = P + f * RSW + q * 4;
        const float2 v0 = s[0], v1 = s[1], v2 = s[2], v3 = s[3];
        float4 oa = make_float4(v0.x * p.scale, v1.x * p.scale, v2.x * p.scale, v3.x * p.scale);
        float4 ob = make_float4(v0.y * p.scale, v1.y * p.scale, v2.y * p.scale, v3.y * p.scale);
        if (rs) {
            oa.x += xa[it].x; oa.y += xa[it].y; oa.z += xa[it].z; oa.w += xa[it].w;
            ob.x += xb[it].x; ob.y += xb[it].y; ob.z += xb[it].z; ob.w += xb[it].w;
        }
        if (p.mask) {
            oa.x = ma[it].x > 0.0f ? oa.x : 0.0f; oa.y = ma[it].y > 0.0f ? oa.y : 0.0f; oa.z = ma[it].z > 0.0f ? oa.z : 0.0f; oa.w = ma[it].w > 0.0f ? oa.w : 0.0f;
            ob.x = mb[it].x > 0.0f ? ob.x : 0.0f; ob.y = mb[it].y > 0.0f ? ob.y : 0.0f; ob.z = mb[it].z > 0.0f ? ob.z : 0.0f; ob.w = mb[it].w > 0.0f ? ob.w : 0.0f;
        }
        *reinterpret_cast<float4*>(dst + (2 * f) * N + q * 4) = oa;
        *reinterpret_cast<float4*>(dst + (2 * f + 1) * N + q * 4) = ob;
    }
}

#include "fft_mr_dev.inc"

// ------------------------------------------------------------------------------------------------
namespace {

// two-pass LDS FFT: power-of-two planes the one-pass kernels cannot hold
bool fft_two_pass_ok(int h, int w) {
    return lama_is_pow2(h) && lama_is_pow2(w) && h >= 16 && w >= 16 && h <= 1024 && w <= 1024 && (h > 128 || w > 128);
}

// one Cooley-Tukey split of the generic path: n = n1 * n2 with n1 the largest divisor <= sqrt(n) (1 for a prime)
static void fft_split1(int n, int& n1, int& n2) {
    n1 = 1;
    for (int d = 2; (long long)d * d <= n; ++d)
        if (n % d == 0) n1 = d;
    n2 = n / n1;
}
static void fft_dft_split(FftParams& p) {
    fft_split1(p.w, p.w1, p.w2);
    fft_split1(p.h, p.h1, p.h2);
}

bool fft_fast_ok(int h, int w) { return lama_is_pow2(h) && lama_is_pow2(w) && h >= 16 && w >= 16 && h <= 128 && w <= 128; }
// the compile-time two-pass kernels (fp32 planes of 256 x 256); LAMA_FFT_Q=0 (profiling build) keeps the generic two-pass kernels
bool fq_ok(int h, int w, bool hf) {
#ifdef LAMA_PROFILING
    static const int on = lama_env_int("LAMA_FFT_Q", 1);
    if (!on) return false;
#endif
    return h == FQ_N && w == FQ_N && !hf;
}

int fft_ppw(int h, int w) {
    // planes per workgroup: keep ~256 butterflies per pass busy, LDS <= 64 KiB
    int ppw = 1;
    while (ppw < 16 && (ppw * 2) * (h / 2) * (w / 8) <= 256 && (ppw * 2) * h <= 256) ppw *= 2;
    return ppw;
}

size_t fft_lds_bytes(int h, int w, int ppw) {
    int wf = w / 2 + 1, hh = h / 2, RSW = w + 1;
    size_t buf = (size_t)ppw * (hh * RSW > h * wf ? hh * RSW : h * wf);
    return ((size_t)w + h + 2 * buf) * sizeof(float2);
}

// planes one workgroup of the sized one-plane kernels walks sequentially (request of plane s + 1 under the transform of
// plane s); LAMA_FFT_SEQ overrides it (1 = one plane per workgroup; A/B runs of the profiling tools and the tests)
int fft_seq(int nplanes) {
    const int want = lama_env_int("LAMA_FFT_SEQ", 1);   // a constant in the product build; the profiling build reads it per launch (tests switch it)
    if (want >= 3 && nplanes % 3 == 0) return 3;
    if (want >= 2 && nplanes % 2 == 0) return 2;
    return 1;
}

// profiling tools only: LAMA_FFT_TRACE = device address of (workgroups * 16) int64 -> per-workgroup phase timeline of the 64 x 64 kernels
long long* fft_trace_buf() {
    static const unsigned long long tr = lama_env_u64("LAMA_FFT_TRACE");
    return reinterpret_cast<long long*>(tr);
}

// one-buffer 64 x 64 kernels (rfft2_ip64_kernel / irfft2_ip64_kernel); LAMA_FFT_INPLACE=0 keeps the two-buffer kernels (A/B
// runs of the profiling tools and the tests)
bool fft_inplace() {
    return lama_env_int("LAMA_FFT_INPLACE", 1) != 0;   // a constant in the product build
}

// mixed-radix plane-in-LDS kernels (fft_mr_dev.inc): the pass list of one length.  Butterflies in registers exist for the radices of
// mr_radices (composites up to 16 as one Cooley-Tukey step with constant twiddles, primes up to 13); the fewest passes win, ties go to the
// list with the larger smallest radix.  Whatever is left of n (prime factors > 13) becomes thread-per-output passes.  -1: does not fit.
const int mr_radices[] = {16, 15, 14, 12, 10, 9, 8, 7, 6, 5, 4, 3, 2, 11, 13};
int mr_plan_smooth(int n, unsigned short* rad, int depth) {      // n = product of listed radices, or -1
    if (n == 1) return 0;
    if (depth >= MR_MAXP) return -1;
    int best = -1, best_min = 0;
    unsigned short tmp[MR_MAXP];
    for (int r : mr_radices) {
        if (n % r) continue;
        const int np = mr_plan_smooth(n / r, tmp + 1, depth + 1);
        if (np < 0) continue;
        tmp[0] = (unsigned short)r;
        int mn = r;
        for (int i = 1; i <= np; ++i) mn = tmp[i] < mn ? tmp[i] : mn;
        if (best < 0 || np + 1 < best || (np + 1 == best && mn > best_min)) {
            best = np + 1;
            best_min = mn;
            for (int i = 0; i <= np; ++i) rad[i] = tmp[i];
        }
    }
    return best;
}
int mr_plan(int n, unsigned short* rad) {
    int np = 0, rest = 1, m = n;
    for (int p = 2; p <= 13; ++p) while (m % p == 0) { m /= p; rest *= p; }      // rest: the part the register butterflies take
    for (int p = 17; m > 1 && p <= 65535 && np < MR_MAXP; p += 2) while (m % p == 0 && np < MR_MAXP) { rad[np++] = (unsigned short)p; m /= p; }
    if (m != 1) return -1;
    if (rest > 1) {
        const int ns = mr_plan_smooth(rest, rad + np, np);
        if (ns < 0) return -1;
        np += ns;
    }
    return np;
}
// threads per workgroup (256, or 1024 for planes of more than 5120 elements -- four waves per SIMD at 128 registers measured a little faster than 512 threads at 256, spills of the radix-14 .. 16 passes included: profiles/r06_fft_mr.txt; 0 = the plane does not fit) and the LDS bytes of the one-pass mixed-radix kernels
int mr_threads(int h, int w, size_t* lds) {
#ifdef LAMA_PROFILING
    { const int on = lama_env_int("LAMA_FFT_MR", 1); if (on != 1) return 0; }      // 0: the round-5 paths (A/B runs); 2: the two-launch form for every plane (tests); read per call
#endif
    if (h < 1 || w < 2 || h > 65535 || w > 65535) return 0;
    const long long hh = (h + 1) / 2, wf = w / 2 + 1, rsw = w | 1;
    const long long elems = hh * rsw > (long long)h * wf ? hh * rsw : (long long)h * wf;
    const long long bytes = (elems + w + h) * (long long)sizeof(float2);
    if (bytes > 160 * 1024) return 0;
    *lds = (size_t)bytes;
    for (int nt = 256; nt <= 1024; nt *= 4)
        if (elems <= (long long)nt * MR_MAXE_OF(nt) && hh * wf <= (long long)nt * (MR_MAXE_OF(nt) / 2)) return nt;
    return 0;
}
bool mr_fill(MrParams& q, const FftParams& p, int nt) {
    memset(&q, 0, sizeof(q));
    q.f = p;
    q.rsw = p.w | 1;
    q.nrp = mr_plan(p.w, q.rrad);
    q.ncp = mr_plan(p.h, q.crad);
    if (q.nrp < 0 || q.ncp < 0) return false;
    // a thread-per-output pass (prime factor > 7) walks the transforms in groups of nt * MR_ANYE / N: at least one must fit
    // (a register pass walks chunks of MAXIT * nt / (N / R) transforms -- MAXIT >= 1: the same bound covers it)
    for (int i = 0; i < q.nrp; ++i) if ((q.rrad[i] > 16 ? p.w : p.w / q.rrad[i]) > nt * (q.rrad[i] > 16 ? MR_ANYE : 1)) return false;
    for (int i = 0; i < q.ncp; ++i) if ((q.crad[i] > 16 ? p.h : p.h / q.crad[i]) > nt * (q.crad[i] > 16 ? MR_ANYE : 1)) return false;
    return true;
}
// the two-launch form (mr2_* kernels, 256 threads): pass lists + LDS bytes of the row / column workgroups; false when a length does not fit
bool mr2_fill(MrParams& q, const FftParams& p, size_t* lds_rows, size_t* lds_cols) {
#ifdef LAMA_PROFILING
    if (lama_env_int("LAMA_FFT_MR", 1) == 0) return false;
#endif
    if (p.h < 1 || p.w < 2 || p.h > 65535 || p.w > 65535) return false;
    if (!mr_fill(q, p, MR2_NT)) return false;
    *lds_rows = ((size_t)p.w + (size_t)MR2_PAIRS * q.rsw) * sizeof(float2);
    *lds_cols = ((size_t)p.h + (size_t)p.h * MR2_COLS) * sizeof(float2);
    return *lds_rows <= 160 * 1024 && *lds_cols <= 160 * 1024;
}

#ifdef MR_BENCH_ONLY      // tools/ubench/mk_mr_bench.sh: one instantiation per kernel (compile time of the ablation builds)
#define MR_GO(name, nt, lds, q) hipLaunchKernelGGL((name<MR_BENCH_ONLY, false>), dim3((q).f.nplanes), dim3(MR_BENCH_ONLY), lds, st, q)
#else
#define MR_GO(name, nt, lds, q)                                                                     \
    do {                                                                                            \
        if (nt == 256) FFT_GO(name, (256), dim3((q).f.nplanes), dim3(256), lds, q);                 \
        else FFT_GO(name, (1024), dim3((q).f.nplanes), dim3(1024), lds, q);                         \
    } while (0)
#endif

bool fft_args_ok(const lama_tensor* real, const lama_tensor* spec, int batch) {
    if (!real || !spec || !real->ptr || !spec->ptr || batch <= 0) return false;
    if (real->C <= 0 || real->H <= 0 || real->W <= 0) return false;
    if (spec->C != 2 * real->C || spec->H != real->H || spec->W != real->W / 2 + 1) return false;
    if (real->batch_stride < (int64_t)real->C * real->H * real->W) return false;
    if (spec->batch_stride < (int64_t)spec->C * spec->H * spec->W) return false;
    return true;
}

}  // namespace

extern "C" size_t lama_fft_workspace_bytes(int32_t batch, int32_t C, int32_t h, int32_t w) {
    if (batch <= 0 || C <= 0 || h <= 0 || w <= 0) return 0;
    if (fft_fast_ok(h, w)) return 0;
    return (size_t)batch * C * h * (w / 2 + 1) * sizeof(float2);
}

// launch kernel<targs..., HF> with HF = (dtype == LAMA_DT_F16); targs is a parenthesised template-argument list
#define FFT_UNPAREN(...) __VA_ARGS__
#define FFT_GO(name, targs, grid, blk, lds, ...)                                                        \
    do {                                                                                                \
        if (hf) hipLaunchKernelGGL((name<FFT_UNPAREN targs, true>), grid, blk, lds, st, __VA_ARGS__);   \
        else hipLaunchKernelGGL((name<FFT_UNPAREN targs, false>), grid, blk, lds, st, __VA_ARGS__);     \
    } while (0)

static int rfft2_impl(void* stream, const lama_tensor* x, const lama_tensor* spec, const lama_tensor* mask, int32_t batch,
                      void* workspace, size_t workspace_bytes, const WoParams* wo = nullptr) {
    if (!fft_args_ok(x, spec, batch)) return LAMA_ERR_BAD_ARG;
    if (x->dtype != spec->dtype || (x->dtype != LAMA_DT_F32 && x->dtype != LAMA_DT_F16)) return LAMA_ERR_UNSUPPORTED;
    const bool hf = x->dtype == LAMA_DT_F16;
    const int es = hf ? 2 : 4;                    // vector accesses are 4 elements: 16-byte (fp32) / 8-byte (fp16) alignment
    const uintptr_t amask = hf ? 7 : 15;
    FftParams p;
    memset(&p, 0, sizeof(p));
    p.x = x->ptr;
    p.x_bstride = x->batch_stride;
    p.spec = spec->ptr;
    p.spec_bstride = spec->batch_stride;
    p.C = x->C; p.h = x->H; p.w = x->W; p.wf = x->W / 2 + 1;
    p.nplanes = batch * x->C;
    p.scale = (float)(1.0 / sqrt((double)p.h * (double)p.w));
    hipStream_t st = (hipStream_t)stream;
    const bool spec_al = (((uintptr_t)spec->ptr | (uintptr_t)(spec->batch_stride * es)) & amask) == 0;
    if (mask) {   // only the compile-time two-pass kernels apply a mask: anything else is the caller's separate lama_act_bwd launch
        if (!mask->ptr || mask->dtype != LAMA_DT_F32 || mask->C != spec->C || mask->H != spec->H || mask->W != spec->W) return LAMA_ERR_BAD_ARG;
        if (!(fq_ok(p.h, p.w, hf) && (((uintptr_t)x->ptr | (uintptr_t)(x->batch_stride * 4) | (uintptr_t)workspace) & 15) == 0)) return LAMA_ERR_UNSUPPORTED;
        p.mask = (const float*)mask->ptr;
        p.mask_bstride = mask->batch_stride;
    }
    if (fft_fast_ok(p.h, p.w) && (((uintptr_t)x->ptr | (uintptr_t)(x->batch_stride * es)) & amask) == 0) {
        p.ppw = fft_ppw(p.h, p.w);
        size_t lds = fft_lds_bytes(p.h, p.w, p.ppw);
        const dim3 grid(lama_ceil_div(p.nplanes, p.ppw)), blk(LAMA_NTHREADS);
        const bool even = p.nplanes % p.ppw == 0;
        p.trace = hf ? nullptr : fft_trace_buf();
        p.prio = lama_side_prio();
        if (wo) {   // with the Winograd output transform of the layer before in the same launch: the one-buffer 64 x 64 kernel only
            if (!(p.h == 64 && p.w == 64 && !hf && fft_inplace() && spec_al)) return LAMA_ERR_UNSUPPORTED;
            p.trace = nullptr;
            int nout = 512;                            // two streaming workgroups per CU next to the six FFT workgroups
#ifdef LAMA_PROFILING
            { static const int v = lama_env_int("LAMA_FFT_NOUT", 0); if (v > 0) nout = v; }
#endif
            hipLaunchKernelGGL(rfft2_ip64_wino_out_kernel, dim3(nout + p.nplanes), blk, (size_t)(IP_N + IP_BUF) * sizeof(float2), st, p, *wo, nout);
        } else if (!p.trace && p.h == 64 && p.w == 64 && fft_inplace() && spec_al)
            FFT_GO(rfft2_ip64_kernel, (false), dim3(p.nplanes), blk, (size_t)(IP_N + IP_BUF) * sizeof(float2), p);
        else if (p.h == 128 && p.w == 128 && fft_inplace() && spec_al)
            FFT_GO(rfft2_ipn_kernel, (128), dim3(p.nplanes), blk, (size_t)(128 + 128 * 65) * sizeof(float2), p);
#ifdef LAMA_PROFILING
        else if (!hf && p.trace && even && p.h == 64 && p.w == 64) hipLaunchKernelGGL((rfft2_lds_kernel<64, 64, 1, 1, true>), grid, blk, lds, st, p);
        else if (!hf && spec_al && p.ppw == 1 && fft_seq(p.nplanes) == 2 && p.h == 64 && p.w == 64) hipLaunchKernelGGL((rfft2_lds_kernel<64, 64, 1, 2>), dim3(p.nplanes / 2), blk, lds, st, p);
        else if (!hf && spec_al && p.ppw == 1 && fft_seq(p.nplanes) == 3 && p.h == 64 && p.w == 64) hipLaunchKernelGGL((rfft2_lds_kernel<64, 64, 1, 3>), dim3(p.nplanes / 3), blk, lds, st, p);
        else if (!hf && spec_al && p.ppw == 1 && fft_seq(p.nplanes) == 2 && p.h == 128 && p.w == 128) hipLaunchKernelGGL((rfft2_lds_kernel<128, 128, 1, 2>), dim3(p.nplanes / 2), blk, lds, st, p);
        else if (!hf && spec_al && p.ppw == 1 && fft_seq(p.nplanes) == 3 && p.h == 128 && p.w == 128) hipLaunchKernelGGL((rfft2_lds_kernel<128, 128, 1, 3>), dim3(p.nplanes / 3), blk, lds, st, p);
#endif
        else if (!hf && even && p.h == 64 && p.w == 64 && p.ppw == 1) hipLaunchKernelGGL((rfft2_lds_kernel<64, 64, 1>), grid, blk, lds, st, p);
        else if (even && p.h == 32 && p.w == 32 && p.ppw == 4) FFT_GO(rfft2_lds_kernel, (32, 32, 4, 1, false), grid, blk, lds, p);
        else if (!hf && even && p.h == 128 && p.w == 128 && p.ppw == 1) hipLaunchKernelGGL((rfft2_lds_kernel<128, 128, 1>), grid, blk, lds, st, p);
        else FFT_GO(rfft2_lds_kernel, (0, 0, 0, 1, false), grid, blk, lds, p);
        LAMA_CHECK_LAUNCH();
        return LAMA_OK;
    }
    if (wo) return LAMA_ERR_UNSUPPORTED;
    const bool fq = fq_ok(p.h, p.w, hf) && (((uintptr_t)x->ptr | (uintptr_t)(x->batch_stride * 4) | (uintptr_t)workspace) & 15) == 0;
    if (!mask && !(fq && workspace)) {   // any plane that fits one workgroup's LDS: the mixed-radix one-pass kernel (no workspace)
        size_t lds = 0;
        MrParams q;
        const int nt = mr_threads(p.h, p.w, &lds);
        if (nt && mr_fill(q, p, nt)) {
#ifdef MR_TRACE
            q.f.trace = fft_trace_buf();
#endif
            MR_GO(rfft2_mr_kernel, nt, lds, q);
            LAMA_CHECK_LAUNCH();
            return LAMA_OK;
        }
    }
    size_t need = (size_t)p.nplanes * p.h * p.wf * sizeof(float2);
    if (!workspace || workspace_bytes < need) return LAMA_ERR_WORKSPACE;
    float2* ws = (float2*)workspace;
    if (fq) {
        hipLaunchKernelGGL(fq_rows_fwd_kernel, dim3(p.nplanes * (FQ_N / 2 / FQ_PAIRS)), dim3(LAMA_NTHREADS), (size_t)(FQ_N + FQ_PAIRS * (FQ_N + 1)) * sizeof(float2), st, p, ws);
        LAMA_CHECK_LAUNCH();
        hipLaunchKernelGGL(fq_cols_kernel<false>, dim3(p.nplanes * (FQ_N / 2 / FQ_CW)), dim3(LAMA_NTHREADS), (size_t)(FQ_N + FQ_N * FQ_CP) * sizeof(float2), st, p, ws);
        LAMA_CHECK_LAUNCH();
        return LAMA_OK;
    }
    if (fft_two_pass_ok(p.h, p.w)) {
        size_t ldsr = ((size_t)p.w + 2 * (size_t)FFT2P_PAIRS * (p.w + 1)) * sizeof(float2);
        size_t ldsc = ((size_t)p.h + 2 * (size_t)FFT2P_COLS * (p.h + 1)) * sizeof(float2);
        if (hf) hipLaunchKernelGGL(fft2p_rows_fwd_kernel<true>, dim3(p.nplanes * lama_ceil_div(p.h / 2, FFT2P_PAIRS)), dim3(LAMA_NTHREADS), ldsr, st, p, ws);
        else hipLaunchKernelGGL(fft2p_rows_fwd_kernel<false>, dim3(p.nplanes * lama_ceil_div(p.h / 2, FFT2P_PAIRS)), dim3(LAMA_NTHREADS), ldsr, st, p, ws);
        LAMA_CHECK_LAUNCH();
        FFT_GO(fft2p_cols_kernel, (false), dim3(p.nplanes * lama_ceil_div(p.wf, FFT2P_COLS)), dim3(LAMA_NTHREADS), ldsc, p, ws);
        LAMA_CHECK_LAUNCH();
        return LAMA_OK;
    }
    if (!mask) {   // any other plane: the two-launch form of the mixed-radix passes (fft_mr_dev.inc); lengths with a prime factor too large for them keep the DFT
        MrParams q;
        size_t ldr = 0, ldc = 0;
        if (mr2_fill(q, p, &ldr, &ldc)) {
            if (hf) hipLaunchKernelGGL(mr2_rows_fwd_kernel<true>, dim3(p.nplanes * lama_ceil_div((p.h + 1) / 2, MR2_PAIRS)), dim3(MR2_NT), ldr, st, q, ws);
            else hipLaunchKernelGGL(mr2_rows_fwd_kernel<false>, dim3(p.nplanes * lama_ceil_div((p.h + 1) / 2, MR2_PAIRS)), dim3(MR2_NT), ldr, st, q, ws);
            LAMA_CHECK_LAUNCH();
            FFT_GO(mr2_cols_kernel, (false), dim3(p.nplanes * lama_ceil_div(p.wf, MR2_COLS)), dim3(MR2_NT), ldc, q, ws);
            LAMA_CHECK_LAUNCH();
            return LAMA_OK;
        }
    }
    long long nrows = (long long)p.nplanes * p.h;
    fft_dft_split(p);
    size_t lds1 = (size_t)p.w * sizeof(float2) + (size_t)DFT_ROWS_PER_WG * ((size_t)(p.w + p.w1) * sizeof(float2) + p.w * sizeof(float));
    size_t lds2 = (size_t)p.h * sizeof(float2) * (1 + 2 * DFT_COLS_PER_WG);
    if (lds1 > 160 * 1024 || lds2 > 160 * 1024) return LAMA_ERR_UNSUPPORTED;
    if (hf) hipLaunchKernelGGL(dft_rows_fwd_kernel<true>, dim3((unsigned)lama_ceil_div64(nrows, DFT_ROWS_PER_WG)), dim3(LAMA_NTHREADS), lds1, st, p, ws);
    else hipLaunchKernelGGL(dft_rows_fwd_kernel<false>, dim3((unsigned)lama_ceil_div64(nrows, DFT_ROWS_PER_WG)), dim3(LAMA_NTHREADS), lds1, st, p, ws);
    LAMA_CHECK_LAUNCH();
    int ncb = lama_ceil_div(p.wf, DFT_COLS_PER_WG);
    FFT_GO(dft_cols_kernel, (false), dim3(p.nplanes * ncb), dim3(LAMA_NTHREADS), lds2, p, ws);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

static int irfft2_impl(void* stream, const lama_tensor* spec, const lama_tensor* resid, const lama_tensor* mask, const lama_tensor* y,
                       int32_t batch, void* workspace, size_t workspace_bytes) {
    if (!fft_args_ok(y, spec, batch)) return LAMA_ERR_BAD_ARG;
    const bool has_r = resid && resid->ptr;
    if (has_r && (resid->C != y->C || resid->H != y->H || resid->W != y->W)) return LAMA_ERR_BAD_ARG;
    if (y->dtype != spec->dtype || (has_r && resid->dtype != y->dtype) || (y->dtype != LAMA_DT_F32 && y->dtype != LAMA_DT_F16))
        return LAMA_ERR_UNSUPPORTED;
    const bool hf = y->dtype == LAMA_DT_F16;
    const int es = hf ? 2 : 4;
    const uintptr_t amask = hf ? 7 : 15;
    FftParams p;
    memset(&p, 0, sizeof(p));
    p.x = has_r ? resid->ptr : nullptr;
    p.x_bstride = has_r ? resid->batch_stride : 0;
    p.spec = spec->ptr;
    p.spec_bstride = spec->batch_stride;
    p.y = y->ptr;
    p.y_bstride = y->batch_stride;
    p.C = y->C; p.h = y->H; p.w = y->W; p.wf = y->W / 2 + 1;
    p.nplanes = batch * y->C;
    p.scale = (float)(1.0 / sqrt((double)p.h * (double)p.w));
    hipStream_t st = (hipStream_t)stream;
    uintptr_t al = (uintptr_t)y->ptr | (uintptr_t)(y->batch_stride * es);
    if (has_r) al |= (uintptr_t)resid->ptr | (uintptr_t)(resid->batch_stride * es);
    const bool spec_al = (((uintptr_t)spec->ptr | (uintptr_t)(spec->batch_stride * es)) & amask) == 0;
    if (mask) {
        if (!mask->ptr || mask->dtype != LAMA_DT_F32 || mask->C != y->C || mask->H != y->H || mask->W != y->W) return LAMA_ERR_BAD_ARG;
        al |= (uintptr_t)mask->ptr | (uintptr_t)(mask->batch_stride * 4);
        if (!(fq_ok(p.h, p.w, hf) && ((al | (uintptr_t)workspace) & 15) == 0)) return LAMA_ERR_UNSUPPORTED;
        p.mask = (const float*)mask->ptr;
        p.mask_bstride = mask->batch_stride;
    }
    if (fft_fast_ok(p.h, p.w) && (al & amask) == 0) {
        p.ppw = fft_ppw(p.h, p.w);
        size_t lds = fft_lds_bytes(p.h, p.w, p.ppw);
        const dim3 grid(lama_ceil_div(p.nplanes, p.ppw)), blk(LAMA_NTHREADS);
        const bool even = p.nplanes % p.ppw == 0;
        p.trace = hf ? nullptr : fft_trace_buf();
        p.prio = lama_side_prio();
        if (!p.trace && p.h == 64 && p.w == 64 && fft_inplace() && spec_al)
            FFT_GO(irfft2_ip64_kernel, (false), dim3(p.nplanes), blk, (size_t)(IP_N + IP_BUF) * sizeof(float2), p);
        else if (p.h == 128 && p.w == 128 && fft_inplace() && spec_al)
            FFT_GO(irfft2_ipn_kernel, (128), dim3(p.nplanes), blk, (size_t)(128 + 128 * 65) * sizeof(float2), p);
#ifdef LAMA_PROFILING
        else if (!hf && p.trace && even && p.h == 64 && p.w == 64) hipLaunchKernelGGL((irfft2_lds_kernel<64, 64, 1, 1, true>), grid, blk, lds, st, p);
        else if (!hf && spec_al && p.ppw == 1 && fft_seq(p.nplanes) == 2 && p.h == 64 && p.w == 64) hipLaunchKernelGGL((irfft2_lds_kernel<64, 64, 1, 2>), dim3(p.nplanes / 2), blk, lds, st, p);
        else if (!hf && spec_al && p.ppw == 1 && fft_seq(p.nplanes) == 3 && p.h == 64 && p.w == 64) hipLaunchKernelGGL((irfft2_lds_kernel<64, 64, 1, 3>), dim3(p.nplanes / 3), blk, lds, st, p);
        else if (!hf && spec_al && p.ppw == 1 && fft_seq(p.nplanes) == 2 && p.h == 128 && p.w == 128) hipLaunchKernelGGL((irfft2_lds_kernel<128, 128, 1, 2>), dim3(p.nplanes / 2), blk, lds, st, p);
        else if (!hf && spec_al && p.ppw == 1 && fft_seq(p.nplanes) == 3 && p.h == 128 && p.w == 128) hipLaunchKernelGGL((irfft2_lds_kernel<128, 128, 1, 3>), dim3(p.nplanes / 3), blk, lds, st, p);
#endif
        else if (!hf && even && p.h == 64 && p.w == 64 && p.ppw == 1) hipLaunchKernelGGL((irfft2_lds_kernel<64, 64, 1>), grid, blk, lds, st, p);
        else if (even && p.h == 32 && p.w == 32 && p.ppw == 4) FFT_GO(irfft2_lds_kernel, (32, 32, 4, 1, false), grid, blk, lds, p);
        else if (!hf && even && p.h == 128 && p.w == 128 && p.ppw == 1) hipLaunchKernelGGL((irfft2_lds_kernel<128, 128, 1>), grid, blk, lds, st, p);
        else FFT_GO(irfft2_lds_kernel, (0, 0, 0, 1, false), grid, blk, lds, p);
        LAMA_CHECK_LAUNCH();
        return LAMA_OK;
    }
    const bool fq = fq_ok(p.h, p.w, hf) && ((al | (uintptr_t)workspace) & 15) == 0;
    if (!mask && !(fq && workspace)) {
        size_t lds = 0;
        MrParams q;
        const int nt = mr_threads(p.h, p.w, &lds);
        if (nt && mr_fill(q, p, nt)) {
            MR_GO(irfft2_mr_kernel, nt, lds, q);
            LAMA_CHECK_LAUNCH();
            return LAMA_OK;
        }
    }
    size_t need = (size_t)p.nplanes * p.h * p.wf * sizeof(float2);
    if (!workspace || workspace_bytes < need) return LAMA_ERR_WORKSPACE;
    float2* ws = (float2*)workspace;
    if (fq) {
        hipLaunchKernelGGL(fq_cols_kernel<true>, dim3(p.nplanes * (FQ_N / 2 / FQ_CW)), dim3(LAMA_NTHREADS), (size_t)(FQ_N + FQ_N * FQ_CP) * sizeof(float2), st, p, ws);
        LAMA_CHECK_LAUNCH();
        hipLaunchKernelGGL(fq_rows_inv_kernel, dim3(p.nplanes * (FQ_N / 2 / FQ_PAIRS)), dim3(LAMA_NTHREADS), (size_t)(FQ_N + FQ_PAIRS * (FQ_N + 1)) * sizeof(float2), st, p, (const float2*)ws);
        LAMA_CHECK_LAUNCH();
        return LAMA_OK;
    }
    if (fft_two_pass_ok(p.h, p.w)) {
        size_t ldsr = ((size_t)p.w + 2 * (size_t)FFT2P_PAIRS * (p.w + 1)) * sizeof(float2);
        size_t ldsc = ((size_t)p.h + 2 * (size_t)FFT2P_COLS * (p.h + 1)) * sizeof(float2);
        FFT_GO(fft2p_cols_kernel, (true), dim3(p.nplanes * lama_ceil_div(p.wf, FFT2P_COLS)), dim3(LAMA_NTHREADS), ldsc, p, ws);
        LAMA_CHECK_LAUNCH();
        if (hf) hipLaunchKernelGGL(fft2p_rows_inv_kernel<true>, dim3(p.nplanes * lama_ceil_div(p.h / 2, FFT2P_PAIRS)), dim3(LAMA_NTHREADS), ldsr, st, p, (const float2*)ws);
        else hipLaunchKernelGGL(fft2p_rows_inv_kernel<false>, dim3(p.nplanes * lama_ceil_div(p.h / 2, FFT2P_PAIRS)), dim3(LAMA_NTHREADS), ldsr, st, p, (const float2*)ws);
        LAMA_CHECK_LAUNCH();
        return LAMA_OK;
    }
    if (!mask) {
        MrParams q;
        size_t ldr = 0, ldc = 0;
        if (mr2_fill(q, p, &ldr, &ldc)) {
            FFT_GO(mr2_cols_kernel, (true), dim3(p.nplanes * lama_ceil_div(p.wf, MR2_COLS)), dim3(MR2_NT), ldc, q, ws);
            LAMA_CHECK_LAUNCH();
            if (hf) hipLaunchKernelGGL(mr2_rows_inv_kernel<true>, dim3(p.nplanes * lama_ceil_div((p.h + 1) / 2, MR2_PAIRS)), dim3(MR2_NT), ldr, st, q, (const float2*)ws);
            else hipLaunchKernelGGL(mr2_rows_inv_kernel<false>, dim3(p.nplanes * lama_ceil_div((p.h + 1) / 2, MR2_PAIRS)), dim3(MR2_NT), ldr, st, q, (const float2*)ws);
            LAMA_CHECK_LAUNCH();
            return LAMA_OK;
        }
    }
    long long nrows = (long long)p.nplanes * p.h;
    fft_dft_split(p);
    size_t lds1 = ((size_t)p.w * (1 + 2 * DFT_ROWS_PER_WG) + (size_t)DFT_ROWS_PER_WG * p.w1) * sizeof(float2);
    size_t lds2 = (size_t)p.h * sizeof(float2) * (1 + 2 * DFT_COLS_PER_WG);
    if (lds1 > 160 * 1024 || lds2 > 160 * 1024) return LAMA_ERR_UNSUPPORTED;
    int ncb = lama_ceil_div(p.wf, DFT_COLS_PER_WG);
    FFT_GO(dft_cols_kernel, (true), dim3(p.nplanes * ncb), dim3(LAMA_NTHREADS), lds2, p, ws);
    LAMA_CHECK_LAUNCH();
    if (hf) hipLaunchKernelGGL(dft_rows_inv_kernel<true>, dim3((unsigned)lama_ceil_div64(nrows, DFT_ROWS_PER_WG)), dim3(LAMA_NTHREADS), lds1, st, p, (const float2*)ws);
    else hipLaunchKernelGGL(dft_rows_inv_kernel<false>, dim3((unsigned)lama_ceil_div64(nrows, DFT_ROWS_PER_WG)), dim3(LAMA_NTHREADS), lds1, st, p, (const float2*)ws);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

extern "C" int lama_rfft2_fwd(void* stream, const lama_tensor* x, const lama_tensor* spec, int32_t batch, void* workspace, size_t workspace_bytes) {
    return rfft2_impl(stream, x, spec, nullptr, batch, workspace, workspace_bytes);
}
extern "C" int lama_irfft2_fwd(void* stream, const lama_tensor* spec, const lama_tensor* resid, const lama_tensor* y, int32_t batch, void* workspace,
                               size_t workspace_bytes) {
    return irfft2_impl(stream, spec, resid, nullptr, y, batch, workspace, workspace_bytes);
}
// (v108) the transforms of the reverse pass with the ReLU derivative that follows them: out = transform(...) * [mask_y > 0].  Only where the
// compile-time two-pass kernels run (fp32 planes of 256 x 256, 16-byte aligned views); LAMA_ERR_UNSUPPORTED otherwise, nothing launched.
extern "C" int lama_rfft2_masked_fwd(void* stream, const lama_tensor* x, const lama_tensor* spec, const lama_tensor* mask_y, int32_t batch,
                                     void* workspace, size_t workspace_bytes) {
    if (!mask_y) return LAMA_ERR_BAD_ARG;
    return rfft2_impl(stream, x, spec, mask_y, batch, workspace, workspace_bytes);
}
extern "C" int lama_irfft2_masked_fwd(void* stream, const lama_tensor* spec, const lama_tensor* resid, const lama_tensor* mask_y, const lama_tensor* y,
                                      int32_t batch, void* workspace, size_t workspace_bytes) {
    if (!mask_y) return LAMA_ERR_BAD_ARG;
    return irfft2_impl(stream, spec, resid, mask_y, y, batch, workspace, workspace_bytes);
}
// (v108) rfft2 of one layer and the output transform of the Winograd local conv of the layer before (lama_winograd_conv3x3_fwd called with
// LAMA_CONV_DEFER_OUT and the same `wino_args` / workspace) in ONE launch: both are memory-bound and independent, and the FFT workgroups leave
// the HBM idle while they transform.  64 x 64 fp32 planes (the one-buffer kernel) only: LAMA_ERR_UNSUPPORTED otherwise, nothing launched --
// the caller then runs lama_winograd_out_fwd + lama_rfft2_fwd.
extern "C" int lama_rfft2_winograd_out_fwd(void* stream, const lama_tensor* x, const lama_tensor* spec, int32_t batch, void* workspace,
                                           size_t workspace_bytes, const lama_conv2d_args* wino_args, void* wino_workspace, size_t wino_workspace_bytes) {
    WoParams wo;
    const int rc = lama_wino_out_params(wino_args, wino_workspace, wino_workspace_bytes, &wo);
    if (rc) return rc;
    return rfft2_impl(stream, x, spec, nullptr, batch, workspace, workspace_bytes, &wo);
}
