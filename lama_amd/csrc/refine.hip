// Kernels of the feature-refinement path (saicinpainting/evaluation/refinement.py, BASELINE configs[4]) that are not
// convolutions / FFTs: the activation derivatives and the reflect-pad adjoint of the generator's backward (dgrad) pass, the
// image pyramid (gaussian blur 5x5 + bilinear resize, forward and adjoint), mask thresholding / erosion, the masked L1 loss
// and its gradient, and the Adam update of the optimised features (z1, z2).  The conv / FFT dgrads themselves reuse
// lama_conv2d_fwd / lama_rfft2_fwd / lama_irfft2_fwd with transposed, flipped weights (lama_amd/backward.py).
// All HBM-bound elementwise / small-stencil work: coalesced along W, one launch each, no LDS needed.
#include "common.h"

namespace {

int rf_grid(long long total) {
    long long g = (total + LAMA_NTHREADS - 1) / LAMA_NTHREADS;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}
bool rf_ok(const lama_tensor* t) { return t && t->ptr && t->dtype == LAMA_DT_F32 && t->C > 0 && t->H > 0 && t->W > 0 && t->batch_stride >= (int64_t)t->C * t->H * t->W; }
bool rf_same(const lama_tensor* a, const lama_tensor* b) { return a->C == b->C && a->H == b->H && a->W == b->W; }

__device__ __forceinline__ int rf_reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// gout = g * act'(y): ReLU (y > 0), sigmoid (y (1 - y)), tanh (1 - y^2) expressed through the layer OUTPUT y
// ------------------------------------------------------------------------------------------------
struct ActBwdParams {
    const float* g; long long g_bs;
    const float* y; long long y_bs;
    float* out; long long out_bs;
    long long per;   // C*H*W
    int B, act;
};

// V = 4: 16-byte accesses (per % 4 == 0, 16-byte aligned views -- the host decides)
template <int V>
__global__ __launch_bounds__(LAMA_NTHREADS) void act_bwd_kernel(ActBwdParams p) {
    typedef float fv_t __attribute__((ext_vector_type(V)));
    const long long perv = p.per / V, total = perv * p.B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / perv);
        const long long r = (i - (long long)b * perv) * V;
        const fv_t g = *reinterpret_cast<const fv_t*>(p.g + b * p.g_bs + r), y = *reinterpret_cast<const fv_t*>(p.y + b * p.y_bs + r);
        fv_t o;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float d = 1.0f;
            if (p.act == LAMA_ACT_RELU) d = y[e] > 0.0f ? 1.0f : 0.0f;
            else if (p.act == LAMA_ACT_SIGMOID) d = y[e] * (1.0f - y[e]);
            else if (p.act == LAMA_ACT_TANH) d = 1.0f - y[e] * y[e];
            o[e] = g[e] * d;
        }
        *reinterpret_cast<fv_t*>(p.out + b * p.out_bs + r) = o;
    }
}

extern "C" int lama_act_bwd(void* stream, const lama_tensor* g, const lama_tensor* y, int32_t act, const lama_tensor* gout, int32_t batch) {
    if (!rf_ok(g) || !rf_ok(y) || !rf_ok(gout) || batch <= 0 || !rf_same(g, y) || !rf_same(g, gout)) return LAMA_ERR_BAD_ARG;
    ActBwdParams p = {(const float*)g->ptr, g->batch_stride, (const float*)y->ptr, y->batch_stride, (float*)gout->ptr, gout->batch_stride,
                      (long long)g->C * g->H * g->W, batch, act};
    const bool v4 = p.per % 4 == 0 && ((p.g_bs | p.y_bs | p.out_bs) & 3) == 0 && (((uintptr_t)g->ptr | (uintptr_t)y->ptr | (uintptr_t)gout->ptr) & 15) == 0;
    if (v4) hipLaunchKernelGGL(act_bwd_kernel<4>, dim3(rf_grid(p.per / 4 * batch)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(act_bwd_kernel<1>, dim3(rf_grid(p.per * batch)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

// ------------------------------------------------------------------------------------------------
// out = a + b (residual joins of the backward pass)
// ------------------------------------------------------------------------------------------------
struct AddParams {
    const float* a; long long a_bs;
    const float* b; long long b_bs;
    float* out; long long out_bs;
    long long per;
    int B;
};

template <int V>
__global__ __launch_bounds__(LAMA_NTHREADS) void add_kernel(AddParams p) {
    typedef float fv_t __attribute__((ext_vector_type(V)));
    const long long perv = p.per / V, total = perv * p.B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / perv);
        const long long r = (i - (long long)b * perv) * V;
        *reinterpret_cast<fv_t*>(p.out + b * p.out_bs + r) =
            *reinterpret_cast<const fv_t*>(p.a + b * p.a_bs + r) + *reinterpret_cast<const fv_t*>(p.b + b * p.b_bs + r);
    }
}

extern "C" int lama_add_fwd(void* stream, const lama_tensor* a, const lama_tensor* b, const lama_tensor* out, int32_t batch) {
    if (!rf_ok(a) || !rf_ok(b) || !rf_ok(out) || batch <= 0 || !rf_same(a, b) || !rf_same(a, out)) return LAMA_ERR_BAD_ARG;
    AddParams p = {(const float*)a->ptr, a->batch_stride, (const float*)b->ptr, b->batch_stride, (float*)out->ptr, out->batch_stride,
                   (long long)a->C * a->H * a->W, batch};
    const bool v4 = p.per % 4 == 0 && ((p.a_bs | p.b_bs | p.out_bs) & 3) == 0 && (((uintptr_t)a->ptr | (uintptr_t)b->ptr | (uintptr_t)out->ptr) & 15) == 0;
    if (v4) hipLaunchKernelGGL(add_kernel<4>, dim3(rf_grid(p.per / 4 * batch)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(add_kernel<1>, dim3(rf_grid(p.per * batch)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

// ------------------------------------------------------------------------------------------------
// adjoint of nn.ReflectionPad2d(pad) (= of the padding_mode='reflect' of the 3x3 / 7x7 convs): g[y][x] = sum of gp over every
// padded position that reads (y, x) [+ addend].  gp [B,C,H+2p,W+2p] -> g [B,C,H,W]
// ------------------------------------------------------------------------------------------------
struct PadBwdParams {
    const float* gp; long long gp_bs;
    const float* add; long long add_bs;
    float* g; long long g_bs;
    int C, B, H, W, pad;
};

__global__ __launch_bounds__(LAMA_NTHREADS) void reflect_pad_bwd_kernel(PadBwdParams p) {
    const int Hp = p.H + 2 * p.pad, Wp = p.W + 2 * p.pad;
    const long long per = (long long)p.C * p.H * p.W, total = per * p.B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const long long r = i - (long long)b * per;
        const int x = (int)(r % p.W);
        const long long r2 = r / p.W;
        const int y = (int)(r2 % p.H);
        const int c = (int)(r2 / p.H);
        // padded rows that map to y: y + pad itself, pad - y (top mirror, 1 <= y <= pad), 2(H-1) - y + pad (bottom mirror)
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = y + p.pad;
        if (y >= 1 && y <= p.pad) ys[ny++] = p.pad - y;
        if (y <= p.H - 2 && y >= p.H - 1 - p.pad) ys[ny++] = 2 * (p.H - 1) - y + p.pad;
        xs[nx++] = x + p.pad;
        if (x >= 1 && x <= p.pad) xs[nx++] = p.pad - x;
        if (x <= p.W - 2 && x >= p.W - 1 - p.pad) xs[nx++] = 2 * (p.W - 1) - x + p.pad;
        const float* src = p.gp + b * p.gp_bs + (long long)c * Hp * Wp;
        float acc = p.add ? p.add[b * p.add_bs + r] : 0.0f;
        for (int a = 0; a < ny; ++a)
            for (int q = 0; q < nx; ++q) acc += src[(long long)ys[a] * Wp + xs[q]];
        p.g[b * p.g_bs + r] = acc;
    }
}

extern "C" int lama_reflect_pad_bwd(void* stream, const lama_tensor* gp, const lama_tensor* addend, int32_t pad, const lama_tensor* g,
                                    int32_t batch) {
    if (!rf_ok(gp) || !rf_ok(g) || batch <= 0 || pad < 0) return LAMA_ERR_BAD_ARG;
    if (gp->C != g->C || gp->H != g->H + 2 * pad || gp->W != g->W + 2 * pad || pad >= g->H || pad >= g->W) return LAMA_ERR_BAD_ARG;
    const bool has_add = addend && addend->ptr;
    if (has_add && (!rf_ok(addend) || !rf_same(addend, g))) return LAMA_ERR_BAD_ARG;
    PadBwdParams p = {(const float*)gp->ptr, gp->batch_stride, has_add ? (const float*)addend->ptr : nullptr, has_add ? addend->batch_stride : 0,
                      (float*)g->ptr, g->batch_stride, g->C, batch, g->H, g->W, pad};
    hipLaunchKernelGGL(reflect_pad_bwd_kernel, dim3(rf_grid((long long)g->C * g->H * g->W * batch)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

// ------------------------------------------------------------------------------------------------
// Round 4: the same adjoint with everything the reverse pass does to its result in the same pass over memory --
//     s = fold(gp) [+ add1] [+ add2];      g = s (optional);      gm = s * act'(mask_y) (optional)
// add1 = the 1x1 path into the same input (conv1 of the SpectralTransform), add2 = the identity path of the resnet block (ffc.py:288),
// mask_y = the taped output of the layer UPSTREAM whose activation derivative the next reverse step starts with.  One launch instead of
// fold + add_kernel + act_bwd_kernel (9.5 instead of 14.5 tensor passes per resnet block), V = 4 consecutive pixels of a row per thread
// (16-byte loads / stores on everything but the padded plane, whose rows are W + 2 pad long).  g may be add2 itself (in place).
// ------------------------------------------------------------------------------------------------
struct FoldParams {
    const float* gp; long long gp_bs;
    const float* add1; long long add1_bs;
    const float* add2; long long add2_bs;
    const float* mask; long long mask_bs;
    float* g; long long g_bs;
    float* gm; long long gm_bs;
    const float* ring;   // null: gp is the padded plane [B,C,H+2p,W+2p]; else gp is the INTERIOR [B,C,H,W] of it and ring [B][C][2(W+2)+2H] holds the one-pixel
                         // frame around it (pad = 1): rows y = -1 and y = H for x = -1..W, then columns x = -1 and x = W for y = 0..H-1 (dgrad_ring_kernel)
    int C, B, H, W, pad, act;
};

__device__ __forceinline__ float rf_act_deriv(int act, float y) {
    if (act == LAMA_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
    if (act == LAMA_ACT_SIGMOID) return y * (1.0f - y);
    if (act == LAMA_ACT_TANH) return 1.0f - y * y;
    return 1.0f;
}

template <int V>
__global__ __launch_bounds__(LAMA_NTHREADS) void reflect_pad_bwd_fused_kernel(FoldParams p) {
    typedef float fv_t __attribute__((ext_vector_type(V)));
    const int Hp = p.H + 2 * p.pad, Wp = p.W + 2 * p.pad, Wv = p.W / V;
    const long long per = (long long)p.C * p.H * Wv, total = per * p.B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const unsigned r = (unsigned)(i - (long long)b * per);          // C * H * W / V < 2^32 (checked by the host)
        const int x0 = (int)(r % (unsigned)Wv) * V;
        const unsigned r2 = r / (unsigned)Wv;
        const int y = (int)(r2 % (unsigned)p.H);
        const int c = (int)(r2 / (unsigned)p.H);
        const long long off = ((long long)c * p.H + y) * p.W + x0;
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.0f;
        if (p.add1) {
            const fv_t a = *reinterpret_cast<const fv_t*>(p.add1 + b * p.add1_bs + off);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += a[e];
        }
        if (p.add2) {
            const fv_t a = *reinterpret_cast<const fv_t*>(p.add2 + b * p.add2_bs + off);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += a[e];
        }
        if (p.ring) {
            const fv_t a = *reinterpret_cast<const fv_t*>(p.gp + b * p.gp_bs + off);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += a[e];
            const bool ytop = y == 1, ybot = y == p.H - 2;
            if (ytop || ybot || x0 <= 1 || x0 + V - 1 >= p.W - 2) {
                const float* top = p.ring + ((long long)b * p.C + c) * (2 * (p.W + 2) + 2 * p.H);
                const float *bot = top + p.W + 2, *left = bot + p.W + 2, *right = left + p.H;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const int x = x0 + e;
                    float r = 0.0f;
                    if (ytop) r += top[x + 1];
                    if (ybot) r += bot[x + 1];
                    if (x == 1) r += left[y] + (ytop ? top[0] : 0.0f) + (ybot ? bot[0] : 0.0f);
                    if (x == p.W - 2) r += right[y] + (ytop ? top[p.W + 1] : 0.0f) + (ybot ? bot[p.W + 1] : 0.0f);
                    acc[e] += r;
                }
            }
        }
        // padded rows that map to y: y + pad itself, pad - y (top mirror, 1 <= y <= pad), 2(H-1) - y + pad (bottom mirror)
        int ys[3], ny = 0;
        ys[ny++] = y + p.pad;
        if (y >= 1 && y <= p.pad) ys[ny++] = p.pad - y;
        if (y <= p.H - 2 && y >= p.H - 1 - p.pad) ys[ny++] = 2 * (p.H - 1) - y + p.pad;
        if (p.ring) ny = 0;
        const float* src = p.gp + b * p.gp_bs + (long long)c * Hp * Wp;
        const bool xedge = x0 <= p.pad || x0 + V - 1 >= p.W - 1 - p.pad;      // some pixel of the group has a mirrored column
        for (int a = 0; a < ny; ++a) {
            const float* row = src + (long long)ys[a] * Wp;
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += row[x0 + e + p.pad];
            if (xedge) {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const int x = x0 + e;
                    if (x >= 1 && x <= p.pad) acc[e] += row[p.pad - x];
                    if (x <= p.W - 2 && x >= p.W - 1 - p.pad) acc[e] += row[2 * (p.W - 1) - x + p.pad];
                }
            }
        }
        fv_t s;
#pragma unroll
        for (int e = 0; e < V; ++e) s[e] = acc[e];
        if (p.g) *reinterpret_cast<fv_t*>(p.g + b * p.g_bs + off) = s;
        if (p.gm) {
            if (p.mask) {
                const fv_t m = *reinterpret_cast<const fv_t*>(p.mask + b * p.mask_bs + off);
#pragma unroll
                for (int e = 0; e < V; ++e) s[e] *= rf_act_deriv(p.act, m[e]);
            }
            *reinterpret_cast<fv_t*>(p.gm + b * p.gm_bs + off) = s;
        }
    }
}

extern "C" int lama_reflect_pad_bwd_fused(void* stream, const lama_tensor* gp, const lama_tensor* add1, const lama_tensor* add2, int32_t pad,
                                          const lama_tensor* mask_y, int32_t act, const lama_tensor* g, const lama_tensor* gm, int32_t batch,
                                          const float* ring) {
    const lama_tensor* out = (g && g->ptr) ? g : gm;
    if (!rf_ok(gp) || !rf_ok(out) || batch <= 0 || pad < 0) return LAMA_ERR_BAD_ARG;
    const int gpad = ring ? 0 : pad;      // with a ring, gp is the interior of the padded plane
    if (ring && pad != 1) return LAMA_ERR_BAD_ARG;
    if (gp->C != out->C || gp->H != out->H + 2 * gpad || gp->W != out->W + 2 * gpad || pad >= out->H || pad >= out->W) return LAMA_ERR_BAD_ARG;
    if ((long long)out->C * out->H * out->W >= (1ll << 32)) return LAMA_ERR_UNSUPPORTED;
    FoldParams p;
    memset(&p, 0, sizeof(p));
    uintptr_t al = 0;
    long long sal = 0;
    auto opt = [&](const lama_tensor* t, const float*& ptr, long long& bs) -> bool {
        if (!t || !t->ptr) return true;
        if (!rf_ok(t) || !rf_same(t, out)) return false;
        ptr = (const float*)t->ptr; bs = t->batch_stride;
        al |= (uintptr_t)t->ptr; sal |= t->batch_stride;
        return true;
    };
    const float *pg = nullptr, *pgm = nullptr;
    if (!opt(add1, p.add1, p.add1_bs) || !opt(add2, p.add2, p.add2_bs) || !opt(mask_y, p.mask, p.mask_bs) || !opt(g, pg, p.g_bs) ||
        !opt(gm, pgm, p.gm_bs))
        return LAMA_ERR_BAD_ARG;
    p.g = const_cast<float*>(pg); p.gm = const_cast<float*>(pgm);
    if (p.gm && p.gm == p.g) return LAMA_ERR_BAD_ARG;
    p.gp = (const float*)gp->ptr; p.gp_bs = gp->batch_stride;
    p.ring = ring;
    if (ring) { al |= (uintptr_t)gp->ptr; sal |= gp->batch_stride; }
    p.C = out->C; p.B = batch; p.H = out->H; p.W = out->W; p.pad = pad; p.act = p.mask ? act : LAMA_ACT_NONE;
    const long long n = (long long)out->C * out->H * out->W * batch;
    if (out->W % 4 == 0 && (al & 15) == 0 && sal % 4 == 0) {
        hipLaunchKernelGGL(reflect_pad_bwd_fused_kernel<4>, dim3(rf_grid(n / 4)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    } else {
        hipLaunchKernelGGL(reflect_pad_bwd_fused_kernel<1>, dim3(rf_grid(n)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    }
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

// ------------------------------------------------------------------------------------------------
// Round 4: the one-pixel FRAME of a 3x3 dgrad.  The data gradient of a reflect-padded 3x3 conv is fold(gp), gp = the zero-padded correlation of
// the output gradient with the flipped weights evaluated on the PADDED plane (H+2) x (W+2).  Its interior H x W is an ordinary zero-pad-1 conv
// (which the Winograd kernel computes with 16/36 of the products and on exactly H W / 128 tiles -- the padded plane of a 256 x 256 state is
// 585 tiles of 128 pixels on 512 workgroup slots); what the fold needs beyond it is the frame: rows y = -1, H and columns x = -1, W, where
// only ONE row (column) of the 3x3 taps reaches the plane:
//     ring[e][t][o] = sum_c sum_k wr[e][c][k][o] * line_e[c][t + k - 1]
// e = top / bottom / left / right, line_e = row 0 / row H-1 / column 0 / column W-1 of the gradient (zero outside), t = -1..W (rows) or 0..H-1
// (columns), wr[top] = w'[o][c][2][k], wr[bottom] = w'[o][c][0][k], wr[left] = w'[o][c][k][2], wr[right] = w'[o][c][k][0]  (w' = the flipped,
// transposed weights of the dgrad conv).  0.4 % of the conv's products at 256 x 256: exact fp32 on the vector ALUs, the weights are read
// coalesced along o.
// ------------------------------------------------------------------------------------------------
struct RingParams {
    const float* g; long long g_bs;
    const float* wr;
    float* ring;
    int C, H, W, M, B;
};

// Workgroup = PP consecutive frame positions x 128 output channels x 8 channel groups (1024 threads): the line values of ALL input channels
// for these positions go to LDS in one round of loads (the launch is a chain of memory latencies, not of FMAs: the first version, one thread
// per (o, 8 positions) walking all 512 channels, took 228 us), every thread then walks C / 8 channels with its weights eight channels ahead,
// and the eight partial sums meet in LDS in a fixed order.
template <int PP>
__global__ __launch_bounds__(1024) void dgrad_ring_kernel(RingParams p) {
    float* const lin = reinterpret_cast<float*>(lama_smem);            // [C][PP + 2]
    const int lin_n = (p.C * (PP + 2) + 5119) / 5120 * 5120;           // whole staging rounds of 5 values x 1024 threads
    float* const red = lin + lin_n;                                    // [8][PP][128]
    const int nTB = (p.W + 2 + PP - 1) / PP, nLR = (p.H + PP - 1) / PP;
    int seg = blockIdx.x, e;
    if (seg < 2 * nTB) { e = seg / nTB; seg -= e * nTB; }
    else { seg -= 2 * nTB; e = 2 + seg / nLR; seg -= (e - 2) * nLR; }
    const int tid = threadIdx.x, cg = LAMA_WAVE_UNIFORM(tid >> 7), ol = tid & 127;
    const int o = blockIdx.y * 128 + ol, b = blockIdx.z;
    const bool rows = e < 2;
    const int elen = rows ? p.W + 2 : p.H, llen = rows ? p.W : p.H, stride = rows ? 1 : p.W;
    const int base = e == 0 ? 0 : (e == 1 ? (p.H - 1) * p.W : (e == 2 ? 0 : p.W - 1));
    const int j0 = seg * PP;                               // first frame position of this workgroup (index along the edge)
    const int i0 = j0 + (rows ? -1 : 0) - 1;               // line index of tap 0 of that position
    const long long HW = (long long)p.H * p.W;
    const lama_buf_t gres = LAMA_BUF_RSRC(p.g + (long long)b * p.g_bs, (long long)p.C * HW * 4);
    // all loads of a round in flight before the first LDS write (the compiler's own order was load, wait, write, five times over)
    for (int base_i = 0; base_i < p.C * (PP + 2); base_i += 5 * 1024) {
        float v[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int idx = base_i + u * 1024 + tid;
            int c = idx / (PP + 2);
            const int jj = idx - c * (PP + 2), i = i0 + jj;
            // raw buffer loads: a position outside the line (or past the last channel) gets an offset beyond the buffer's range and reads 0 --
            // no branch around the load (hipcc sinks a plain load into the branch of its select, and each then ends in a wait of its own)
            const unsigned off = (c < p.C && i >= 0 && i < llen) ? (unsigned)((c * (int)HW + base + i * stride) * 4) : 0xfffffff0u;
            v[u] = __builtin_bit_cast(float, LAMA_BUF_LOAD_B32(gres, off, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 5; ++u) lin[base_i + u * 1024 + tid] = v[u];       // (lin is padded to whole rounds: no condition, no branch)
    }
    __syncthreads();
    float acc[PP];
#pragma unroll
    for (int q = 0; q < PP; ++q) acc[q] = 0.0f;
    const int cper = p.C >> 3;                             // a multiple of 8 (host)
    const float* wb = p.wr + ((long long)e * p.C + cg * cper) * 3 * p.M + o;
    const float* lb = lin + cg * cper * (PP + 2);
    // weights eight channels ahead in registers: one L2 latency per eight channels, under the FMAs of the eight before
    float wc[24], wn[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) wc[k] = wb[(long long)k * p.M];
#pragma unroll 1
    for (int c0 = 0; c0 < cper; c0 += 8) {
        const int cn = c0 + 8 < cper ? c0 + 8 : c0;                      // (the last round re-reads its own weights: no branch around the loads)
#pragma unroll
        for (int k = 0; k < 24; ++k) wn[k] = wb[(long long)(cn * 3 + k) * p.M];
        __builtin_amdgcn_sched_barrier(0);                                // the requests first, then this round's FMAs
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* in = lb + (c0 + u) * (PP + 2);
#pragma unroll
            for (int q = 0; q < PP; ++q) acc[q] = fmaf(wc[3 * u], in[q], fmaf(wc[3 * u + 1], in[q + 1], fmaf(wc[3 * u + 2], in[q + 2], acc[q])));
        }
#pragma unroll
        for (int k = 0; k < 24; ++k) wc[k] = wn[k];
    }
#pragma unroll
    for (int q = 0; q < PP; ++q) red[(cg * PP + q) * 128 + ol] = acc[q];
    __syncthreads();
    {
        const int q = tid >> 7;                            // 1024 threads = PP (= 8) positions x 128 channels
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += red[(k * PP + q) * 128 + ol];
        const int RL = 2 * (p.W + 2) + 2 * p.H;
        const int eoff = e == 0 ? 0 : (e == 1 ? p.W + 2 : (e == 2 ? 2 * (p.W + 2) : 2 * (p.W + 2) + p.H));
        if (j0 + q < elen) p.ring[((long long)b * p.M + o) * RL + eoff + j0 + q] = sum;
    }
}

extern "C" size_t lama_dgrad_ring_bytes(int32_t batch, int32_t cout, int32_t H, int32_t W) {
    if (batch <= 0 || cout <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)batch * cout * (2 * (W + 2) + 2 * H) * sizeof(float);
}

extern "C" int lama_dgrad_ring_fwd(void* stream, const lama_tensor* g, const float* w_ring, int32_t cout, float* ring, int32_t batch) {
    if (!rf_ok(g) || !w_ring || !ring || batch <= 0 || cout <= 0) return LAMA_ERR_BAD_ARG;
    constexpr int PP = 8;                                   // = 1024 threads / 128 output channels (the reduction's thread map)
    const size_t shmem = (((size_t)g->C * (PP + 2) + 5119) / 5120 * 5120 + 8 * PP * 128) * sizeof(float);
    if (cout % 128 != 0 || g->C % 64 != 0 || g->H < 2 || g->W < 2 || batch > 65535 || shmem > 96 * 1024 || (long long)g->C * g->H * g->W * 4 >= (1ll << 31))
        return LAMA_ERR_UNSUPPORTED;
    RingParams p = {(const float*)g->ptr, g->batch_stride, w_ring, ring, g->C, g->H, g->W, cout, batch};
    const int nTB = (g->W + 2 + PP - 1) / PP, nLR = (g->H + PP - 1) / PP;
    hipLaunchKernelGGL(dgrad_ring_kernel<PP>, dim3(2 * nTB + 2 * nLR, cout / 128, batch), dim3(1024), shmem, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

// ------------------------------------------------------------------------------------------------
// kornia.filters.gaussian_blur2d(x, (5,5), (1,1)), border_type='reflect' (refinement.py:24,52), on the top-left crop
// [0:y.H, 0:y.W] of x (refinement.py:149 blurs pred[:, :, :orig_h, :orig_w]); and its adjoint (zeros outside the crop).
// ------------------------------------------------------------------------------------------------
struct GaussParams {
    const float* x; long long x_bs;
    float* y; long long y_bs;
    int C, B;
    int XH, XW;   // dims of the wide tensor (x in fwd, gx in bwd)
    int H, W;     // dims of the cropped / blurred tensor
    float k[5];
};

__global__ __launch_bounds__(LAMA_NTHREADS) void gauss5_fwd_kernel(GaussParams p) {
    const long long per = (long long)p.C * p.H * p.W, total = per * p.B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const long long r = i - (long long)b * per;
        const int x = (int)(r % p.W);
        const long long r2 = r / p.W;
        const int y = (int)(r2 % p.H);
        const int c = (int)(r2 / p.H);
        const float* src = p.x + b * p.x_bs + (long long)c * p.XH * p.XW;
        float acc = 0.0f;
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy) {
            const int yy = rf_reflect(y + dy, p.H);
            float row = 0.0f;
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx) row += p.k[dx + 2] * src[(long long)yy * p.XW + rf_reflect(x + dx, p.W)];
            acc += p.k[dy + 2] * row;
        }
        p.y[b * p.y_bs + r] = acc;
    }
}

// gx[y][x] = sum over padded positions (py, px) that reflect onto (y, x) of sum_{dy,dx} k[dy] k[dx] gy[py - dy][px - dx]
__global__ __launch_bounds__(LAMA_NTHREADS) void gauss5_bwd_kernel(GaussParams p) {
    // here x = gy (cropped, H x W), y = gx (wide, XH x XW)
    const long long per = (long long)p.C * p.XH * p.XW, total = per * p.B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const long long r = i - (long long)b * per;
        const int x = (int)(r % p.XW);
        const long long r2 = r / p.XW;
        const int y = (int)(r2 % p.XH);
        const int c = (int)(r2 / p.XH);
        float acc = 0.0f;
        if (y < p.H && x < p.W) {
            const float* gy = p.x + b * p.x_bs + (long long)c * p.H * p.W;
            // padded positions that reflect onto row y: y itself, its mirror about row 0 (-y, for 1 <= y <= 2) and its mirror about
            // row H - 1 (2 (H - 1) - y, for H - 3 <= y <= H - 2) -- at most three, enumerated directly; the same for columns
            int pys[3], pxs[3], ny = 0, nx = 0;
            pys[ny++] = y;
            if (y >= 1 && y <= 2) pys[ny++] = -y;
            if (y >= p.H - 3 && y <= p.H - 2) pys[ny++] = 2 * (p.H - 1) - y;
            pxs[nx++] = x;
            if (x >= 1 && x <= 2) pxs[nx++] = -x;
            if (x >= p.W - 3 && x <= p.W - 2) pxs[nx++] = 2 * (p.W - 1) - x;
            for (int iy = 0; iy < ny; ++iy) {
                const int py = pys[iy];
                for (int ix = 0; ix < nx; ++ix) {
                    const int px = pxs[ix];
                    // output pixels q = (py - dy, px - dx) that read padded position (py, px) with weight k[dy] k[dx]
                    for (int dy = -2; dy <= 2; ++dy) {
                        const int qy = py - dy;
                        if (qy < 0 || qy >= p.H) continue;
                        for (int dx = -2; dx <= 2; ++dx) {
                            const int qx = px - dx;
                            if (qx < 0 || qx >= p.W) continue;
                            acc += p.k[dy + 2] * p.k[dx + 2] * gy[(long long)qy * p.W + qx];
                        }
                    }
                }
            }
        }
        p.y[b * p.y_bs + r] = acc;
    }
}

namespace {
void rf_gauss_weights(float* k) {
    double g[5], s = 0.0;
    for (int i = 0; i < 5; ++i) { const double x = (double)(i - 2); g[i] = (double)expf((float)(-x * x / 2.0)); s += g[i]; }
    // kornia computes the window in fp32: exp(-x^2 / (2 sigma^2)) / sum
    float gf[5], sf = 0.0f;
    for (int i = 0; i < 5; ++i) { const float x = (float)(i - 2); gf[i] = expf(-x * x / 2.0f); sf += gf[i]; }
    for (int i = 0; i < 5; ++i) k[i] = gf[i] / sf;
    (void)g; (void)s;
}
}  // namespace

extern "C" int lama_gauss5_fwd(void* stream, const lama_tensor* x, const lama_tensor* y, int32_t batch) {
    if (!rf_ok(x) || !rf_ok(y) || batch <= 0 || x->C != y->C || y->H > x->H || y->W > x->W || y->H < 3 || y->W < 3) return LAMA_ERR_BAD_ARG;
    GaussParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const float*)x->ptr; p.x_bs = x->batch_stride;
    p.y = (float*)y->ptr; p.y_bs = y->batch_stride;
    p.C = x->C; p.B = batch; p.XH = x->H; p.XW = x->W; p.H = y->H; p.W = y->W;
    rf_gauss_weights(p.k);
    hipLaunchKernelGGL(gauss5_fwd_kernel, dim3(rf_grid((long long)y->C * y->H * y->W * batch)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

extern "C" int lama_gauss5_bwd(void* stream, const lama_tensor* gy, const lama_tensor* gx, int32_t batch) {
    if (!rf_ok(gy) || !rf_ok(gx) || batch <= 0 || gx->C != gy->C || gy->H > gx->H || gy->W > gx->W || gy->H < 3 || gy->W < 3) return LAMA_ERR_BAD_ARG;
    GaussParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const float*)gy->ptr; p.x_bs = gy->batch_stride;
    p.y = (float*)gx->ptr; p.y_bs = gx->batch_stride;
    p.C = gx->C; p.B = batch; p.XH = gx->H; p.XW = gx->W; p.H = gy->H; p.W = gy->W;
    rf_gauss_weights(p.k);
    hipLaunchKernelGGL(gauss5_bwd_kernel, dim3(rf_grid((long long)gx->C * gx->H * gx->W * batch)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

// ------------------------------------------------------------------------------------------------
// F.interpolate(x, size=(Ho, Wo), mode='bilinear', align_corners=False) (refinement.py:25,53,55; kornia resize :200-201) and adjoint.
// PyTorch: scale = in / out; src = max(0, (dst + 0.5) * scale - 0.5); i0 = floor(src); i1 = min(i0 + 1, in - 1); l = src - i0.
// ------------------------------------------------------------------------------------------------
struct BilinParams {
    const float* x; long long x_bs;
    float* y; long long y_bs;
    int C, B, H, W, Ho, Wo;
    float sy, sx;
};

__device__ __forceinline__ void rf_bilin_src(int o, float scale, int n, int& i0, int& i1, float& l) {
    float s = ((float)o + 0.5f) * scale - 0.5f;
    s = s < 0.0f ? 0.0f : s;
    i0 = (int)s;
    if (i0 > n - 1) i0 = n - 1;
    i1 = i0 + 1 < n ? i0 + 1 : n - 1;
    l = s - (float)i0;
}

__global__ __launch_bounds__(LAMA_NTHREADS) void bilinear_fwd_kernel(BilinParams p) {
    const long long per = (long long)p.C * p.Ho * p.Wo, total = per * p.B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const long long r = i - (long long)b * per;
        const int xo = (int)(r % p.Wo);
        const long long r2 = r / p.Wo;
        const int yo = (int)(r2 % p.Ho);
        const int c = (int)(r2 / p.Ho);
        int y0, y1, x0, x1;
        float ly, lx;
        rf_bilin_src(yo, p.sy, p.H, y0, y1, ly);
        rf_bilin_src(xo, p.sx, p.W, x0, x1, lx);
        const float* s = p.x + b * p.x_bs + (long long)c * p.H * p.W;
        const float v00 = s[(long long)y0 * p.W + x0], v01 = s[(long long)y0 * p.W + x1];
        const float v10 = s[(long long)y1 * p.W + x0], v11 = s[(long long)y1 * p.W + x1];
        // same association as ATen's upsample_bilinear2d: h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11)
        p.y[b * p.y_bs + r] = (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
    }
}

// gather form of the adjoint (deterministic, no atomics): gx[yi][xi] = sum over outputs whose (y0 | y1, x0 | x1) hit (yi, xi)
__global__ __launch_bounds__(LAMA_NTHREADS) void bilinear_bwd_kernel(BilinParams p) {
    // x = gy [Ho x Wo], y = gx [H x W]
    const long long per = (long long)p.C * p.H * p.W, total = per * p.B;
    const float isy = 1.0f / p.sy, isx = 1.0f / p.sx;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const long long r = i - (long long)b * per;
        const int xi = (int)(r % p.W);
        const long long r2 = r / p.W;
        const int yi = (int)(r2 % p.H);
        const int c = (int)(r2 / p.H);
        // outputs o with floor(src(o)) in {i - 1, i}: src(o) in [i - 1, i + 1) -> o in ((i - 0.5) / s - 0.5, (i + 1.5) / s - 0.5); clamping at the
        // borders (src < 0 -> 0, i1 = n - 1) only maps onto the first / last input, which the widened range below covers
        int yo_lo = (int)floorf(((float)yi - 0.5f) * isy - 0.5f) - 1, yo_hi = (int)ceilf(((float)yi + 1.5f) * isy - 0.5f) + 1;
        int xo_lo = (int)floorf(((float)xi - 0.5f) * isx - 0.5f) - 1, xo_hi = (int)ceilf(((float)xi + 1.5f) * isx - 0.5f) + 1;
        if (yi == 0) yo_lo = 0;
        if (xi == 0) xo_lo = 0;
        if (yi == p.H - 1) yo_hi = p.Ho - 1;
        if (xi == p.W - 1) xo_hi = p.Wo - 1;
        yo_lo = yo_lo < 0 ? 0 : yo_lo; xo_lo = xo_lo < 0 ? 0 : xo_lo;
        yo_hi = yo_hi > p.Ho - 1 ? p.Ho - 1 : yo_hi; xo_hi = xo_hi > p.Wo - 1 ? p.Wo - 1 : xo_hi;
        const float* gy = p.x + b * p.x_bs + (long long)c * p.Ho * p.Wo;
        float acc = 0.0f;
        for (int yo = yo_lo; yo <= yo_hi; ++yo) {
            int y0, y1; float ly;
            rf_bilin_src(yo, p.sy, p.H, y0, y1, ly);
            const float wy = (y0 == yi ? 1.0f - ly : 0.0f) + (y1 == yi ? ly : 0.0f);
            if (wy == 0.0f) continue;
            for (int xo = xo_lo; xo <= xo_hi; ++xo) {
                int x0, x1; float lx;
                rf_bilin_src(xo, p.sx, p.W, x0, x1, lx);
                const float wx = (x0 == xi ? 1.0f - lx : 0.0f) + (x1 == xi ? lx : 0.0f);
                if (wx != 0.0f) acc += wy * wx * gy[(long long)yo * p.Wo + xo];
            }
        }
        p.y[b * p.y_bs + r] = acc;
    }
}

extern "C" int lama_bilinear_fwd(void* stream, const lama_tensor* x, const lama_tensor* y, int32_t batch) {
    if (!rf_ok(x) || !rf_ok(y) || batch <= 0 || x->C != y->C) return LAMA_ERR_BAD_ARG;
    BilinParams p = {(const float*)x->ptr, x->batch_stride, (float*)y->ptr, y->batch_stride, x->C, batch, x->H, x->W, y->H, y->W,
                     (float)x->H / (float)y->H, (float)x->W / (float)y->W};
    hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(rf_grid((long long)y->C * y->H * y->W * batch)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

extern "C" int lama_bilinear_bwd(void* stream, const lama_tensor* gy, const lama_tensor* gx, int32_t batch) {
    if (!rf_ok(gy) || !rf_ok(gx) || batch <= 0 || gx->C != gy->C) return LAMA_ERR_BAD_ARG;
    BilinParams p = {(const float*)gy->ptr, gy->batch_stride, (float*)gx->ptr, gx->batch_stride, gx->C, batch, gx->H, gx->W, gy->H, gy->W,
                     (float)gx->H / (float)gy->H, (float)gx->W / (float)gy->W};
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(rf_grid((long long)gx->C * gx->H * gx->W * batch)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

// ------------------------------------------------------------------------------------------------
// y = x >= thr ? 1 : 0  (mask binarisation, refinement.py:57-62,70-71,206,300-301)
// ------------------------------------------------------------------------------------------------
struct ThrParams {
    const float* x; long long x_bs;
    float* y; long long y_bs;
    long long per;
    int B;
    float thr;
};

__global__ __launch_bounds__(LAMA_NTHREADS) void threshold_kernel(ThrParams p) {
    const long long total = p.per * p.B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / p.per);
        const long long r = i - (long long)b * p.per;
        p.y[b * p.y_bs + r] = p.x[b * p.x_bs + r] >= p.thr ? 1.0f : 0.0f;
    }
}

extern "C" int lama_threshold_fwd(void* stream, const lama_tensor* x, float thr, const lama_tensor* y, int32_t batch) {
    if (!rf_ok(x) || !rf_ok(y) || batch <= 0 || !rf_same(x, y)) return LAMA_ERR_BAD_ARG;
    ThrParams p = {(const float*)x->ptr, x->batch_stride, (float*)y->ptr, y->batch_stride, (long long)x->C * x->H * x->W, batch, thr};
    hipLaunchKernelGGL(threshold_kernel, dim3(rf_grid(p.per * batch)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

// ------------------------------------------------------------------------------------------------
// kornia.morphology.erosion(x, kernel) with a flat structuring element se [kh, kw] (non-zero = member), origin = centre, geodesic
// border (positions outside the image do not lower the minimum): y = min over members of x  (refinement.py:68, 15 x 15 ellipse)
// ------------------------------------------------------------------------------------------------
struct ErodeParams {
    const float* x; long long x_bs;
    float* y; long long y_bs;
    const float* se;
    int C, B, H, W, kh, kw;
    float max_val;
};

__global__ __launch_bounds__(LAMA_NTHREADS) void erode_kernel(ErodeParams p) {
    const long long per = (long long)p.C * p.H * p.W, total = per * p.B;
    const int oy = p.kh / 2, ox = p.kw / 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const long long r = i - (long long)b * per;
        const int x = (int)(r % p.W);
        const long long r2 = r / p.W;
        const int y = (int)(r2 % p.H);
        const int c = (int)(r2 / p.H);
        const float* s = p.x + b * p.x_bs + (long long)c * p.H * p.W;
        float m = p.max_val;
        for (int ky = 0; ky < p.kh; ++ky) {
            const int yy = y + ky - oy;
            if (yy < 0 || yy >= p.H) continue;
            for (int kx = 0; kx < p.kw; ++kx) {
                const int xx = x + kx - ox;
                // kornia subtracts the FLIPPED neighbourhood: member test on se[kh-1-ky][kw-1-kx]
                if (xx < 0 || xx >= p.W || p.se[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)] == 0.0f) continue;
                m = fminf(m, s[(long long)yy * p.W + xx]);
            }
        }
        p.y[b * p.y_bs + r] = m;
    }
}

extern "C" int lama_erode_fwd(void* stream, const lama_tensor* x, const float* se, int32_t kh, int32_t kw, float max_val, const lama_tensor* y,
                              int32_t batch) {
    if (!rf_ok(x) || !rf_ok(y) || !se || batch <= 0 || !rf_same(x, y) || kh <= 0 || kw <= 0 || !(kh & 1) || !(kw & 1)) return LAMA_ERR_BAD_ARG;
    ErodeParams p = {(const float*)x->ptr, x->batch_stride, (float*)y->ptr, y->batch_stride, se, x->C, batch, x->H, x->W, kh, kw, max_val};
    hipLaunchKernelGGL(erode_kernel, dim3(rf_grid((long long)x->C * x->H * x->W * batch)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

// ------------------------------------------------------------------------------------------------
// masked L1 terms of refinement.py:75-84: over the elements selected by the mask (mask < thr, or mask >= thr; a 1-channel mask
// is broadcast over the channels of pred like mask.repeat(1,3,1,1)):
//   fwd: accum[0] += sum |pred - target|, accum[1] += count          (accum: two device floats... doubles, zeroed by the caller)
//   bwd: g (+)= scale * sign(pred - target) on the selected elements, 0 elsewhere   (d mean|.| / d pred, scale = 1 / count)
// ------------------------------------------------------------------------------------------------
struct L1Params {
    const float* pred; long long pred_bs;
    const float* tgt; long long tgt_bs;
    const float* mask; long long mask_bs;
    float* g; long long g_bs;
    double* accum;
    long long hw;
    int C, B, mask_c, select_ge, accumulate;
    float thr, scale;
};

__global__ __launch_bounds__(LAMA_NTHREADS) void l1_masked_fwd_kernel(L1Params p) {
    const long long per = (long long)p.C * p.hw, total = per * p.B;
    double s = 0.0, n = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const long long r = i - (long long)b * per;
        const int c = (int)(r / p.hw);
        const long long px = r - (long long)c * p.hw;
        const float m = p.mask[b * p.mask_bs + (p.mask_c == 1 ? 0 : c) * p.hw + px];
        const bool sel = p.select_ge ? (m >= p.thr) : (m < p.thr);
        if (sel) { s += (double)fabsf(p.pred[b * p.pred_bs + r] - p.tgt[b * p.tgt_bs + r]); n += 1.0; }
    }
    // workgroup reduction through LDS, one atomic pair per workgroup
    double* red = reinterpret_cast<double*>(lama_smem);
    red[threadIdx.x] = s;
    red[LAMA_NTHREADS + threadIdx.x] = n;
    __syncthreads();
    for (int st = LAMA_NTHREADS / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { red[threadIdx.x] += red[threadIdx.x + st]; red[LAMA_NTHREADS + threadIdx.x] += red[LAMA_NTHREADS + threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicAdd(&p.accum[0], red[0]); atomicAdd(&p.accum[1], red[LAMA_NTHREADS]); }
}

__global__ __launch_bounds__(LAMA_NTHREADS) void l1_masked_bwd_kernel(L1Params p) {
    const long long per = (long long)p.C * p.hw, total = per * p.B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const long long r = i - (long long)b * per;
        const int c = (int)(r / p.hw);
        const long long px = r - (long long)c * p.hw;
        const float m = p.mask[b * p.mask_bs + (p.mask_c == 1 ? 0 : c) * p.hw + px];
        const bool sel = p.select_ge ? (m >= p.thr) : (m < p.thr);
        float g = 0.0f;
        if (sel) {
            const float d = p.pred[b * p.pred_bs + r] - p.tgt[b * p.tgt_bs + r];
            g = d > 0.0f ? p.scale : (d < 0.0f ? -p.scale : 0.0f);      // torch: abs'(0) = sign(0) = 0
        }
        float* o = p.g + b * p.g_bs + r;
        *o = p.accumulate ? *o + g : g;
    }
}

namespace {
int rf_l1_fill(L1Params& p, const lama_tensor* pred, const lama_tensor* target, const lama_tensor* mask, float thr, int select_ge, int batch) {
    if (!rf_ok(pred) || !rf_ok(target) || !rf_ok(mask) || batch <= 0 || !rf_same(pred, target)) return LAMA_ERR_BAD_ARG;
    if (mask->H != pred->H || mask->W != pred->W || (mask->C != 1 && mask->C != pred->C)) return LAMA_ERR_BAD_ARG;
    memset(&p, 0, sizeof(p));
    p.pred = (const float*)pred->ptr; p.pred_bs = pred->batch_stride;
    p.tgt = (const float*)target->ptr; p.tgt_bs = target->batch_stride;
    p.mask = (const float*)mask->ptr; p.mask_bs = mask->batch_stride;
    p.hw = (long long)pred->H * pred->W;
    p.C = pred->C; p.B = batch; p.mask_c = mask->C; p.select_ge = select_ge; p.thr = thr;
    return LAMA_OK;
}
}  // namespace

extern "C" int lama_l1_masked_fwd(void* stream, const lama_tensor* pred, const lama_tensor* target, const lama_tensor* mask, float thr,
                                  int32_t select_ge, double* accum, int32_t batch) {
    L1Params p;
    int rc = rf_l1_fill(p, pred, target, mask, thr, select_ge, batch);
    if (rc) return rc;
    if (!accum) return LAMA_ERR_BAD_ARG;
    p.accum = accum;
    int grid = rf_grid((long long)p.C * p.hw * batch);
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(l1_masked_fwd_kernel, dim3(grid), dim3(LAMA_NTHREADS), 2 * LAMA_NTHREADS * sizeof(double), (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

extern "C" int lama_l1_masked_bwd(void* stream, const lama_tensor* pred, const lama_tensor* target, const lama_tensor* mask, float thr,
                                  int32_t select_ge, float scale, int32_t accumulate, const lama_tensor* g, int32_t batch) {
    L1Params p;
    int rc = rf_l1_fill(p, pred, target, mask, thr, select_ge, batch);
    if (rc) return rc;
    if (!rf_ok(g) || !rf_same(g, pred)) return LAMA_ERR_BAD_ARG;
    p.g = (float*)g->ptr; p.g_bs = g->batch_stride;
    p.scale = scale; p.accumulate = accumulate;
    hipLaunchKernelGGL(l1_masked_bwd_kernel, dim3(rf_grid((long long)p.C * p.hw * batch)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

// ------------------------------------------------------------------------------------------------
// torch.optim.Adam([z1, z2], lr) step t (betas 0.9 / 0.999, eps 1e-8, no weight decay, no amsgrad; refinement.py:134,165) on a flat buffer
// ------------------------------------------------------------------------------------------------
struct AdamParams {
    float* p; const float* g; float* m; float* v;
    long long n;
    float lr, b1, b2, eps, bc1, bc2_sqrt;
};

__global__ __launch_bounds__(LAMA_NTHREADS) void adam_kernel(AdamParams a) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long long)gridDim.x * blockDim.x) {
        const float g = a.g[i];
        const float m = a.b1 * a.m[i] + (1.0f - a.b1) * g;
        const float v = a.b2 * a.v[i] + (1.0f - a.b2) * g * g;
        a.m[i] = m;
        a.v[i] = v;
        const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
        a.p[i] -= (a.lr / a.bc1) * (m / denom);
    }
}

extern "C" int lama_adam_step(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                              float beta2, float eps, int32_t step) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step < 1) return LAMA_ERR_BAD_ARG;
    AdamParams a = {param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, (float)(1.0 - pow((double)beta1, step)),
                    (float)sqrt(1.0 - pow((double)beta2, step))};
    hipLaunchKernelGGL(adam_kernel, dim3(rf_grid(n)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, a);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}
