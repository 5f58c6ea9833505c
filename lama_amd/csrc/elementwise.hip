// Elementwise pre/post kernels of the inference path, the composite FourierUnit entry point and the
// misc C-ABI functions.
#include "common.h"
#include <initializer_list>

struct EwParams {
    const float* img;
    long long img_bs;
    const float* mask;
    long long mask_bs;
    const float* pred;
    long long pred_bs;
    float* out;
    long long out_bs;
    long long hw;
    int B;
};

// masked_img = cat(img*(1-mask), mask)  (trainers/default.py:59,67-68)
__global__ __launch_bounds__(LAMA_NTHREADS) void mask_compose_kernel(EwParams p) {
    long long total = (long long)p.B * p.hw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int b = (int)(i / p.hw);
        long long px = i - (long long)b * p.hw;
        float m = p.mask[b * p.mask_bs + px];
        const float* im = p.img + b * p.img_bs + px;
        float* o = p.out + b * p.out_bs + px;
        float k = 1.0f - m;
        o[0] = im[0] * k;
        o[p.hw] = im[p.hw] * k;
        o[2 * p.hw] = im[2 * p.hw] * k;
        o[3 * p.hw] = m;
    }
}

// inpainted = mask*pred + (1-mask)*img  (trainers/default.py:71)
__global__ __launch_bounds__(LAMA_NTHREADS) void blend_kernel(EwParams p) {
    long long total = (long long)p.B * p.hw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int b = (int)(i / p.hw);
        long long px = i - (long long)b * p.hw;
        float m = p.mask[b * p.mask_bs + px];
        float k = 1.0f - m;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            p.out[b * p.out_bs + c * p.hw + px] = m * p.pred[b * p.pred_bs + c * p.hw + px] + k * p.img[b * p.img_bs + c * p.hw + px];
    }
}

struct QuantParams {
    const float* src;
    long long src_bs;
    uint8_t* dst;
    int B, H, W, ch, cw;
};

// np.clip(x*255, 0, 255).astype('uint8') (truncation) of x.permute(1,2,0)[:ch,:cw]  (bin/predict.py:86-92)
__global__ __launch_bounds__(LAMA_NTHREADS) void quantize_u8_hwc_kernel(QuantParams p) {
    long long total = (long long)p.B * p.ch * p.cw * 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % 3);
        long long r = i / 3;
        int x = (int)(r % p.cw);
        r /= p.cw;
        int y = (int)(r % p.ch);
        int b = (int)(r / p.ch);
        float v = p.src[b * p.src_bs + ((long long)c * p.H + y) * p.W + x] * 255.0f;
        v = fminf(fmaxf(v, 0.0f), 255.0f);
        p.dst[i] = (uint8_t)v;
    }
}

// ---- round 6 (v110): the predict step fed with what is on disk -- u8 HWC image + u8 mask, unpadded -- instead of fp32 NCHW tensors the host
// converted and padded (33.6 MB instead of 8.4 MB over PCIe per 8 x 512^2 step, and a float conversion + np.pad per image on the host).
//   image / 255 in fp32 (evaluation/data.py:12-20: astype('float32') / 255 -- the same IEEE division), symmetric padding at the bottom / right
//   up to (Hp, Wp) (data.py:29-33: padded row h + i = row h - 1 - i), mask > 0 (bin/predict.py:84) or mask / 255.
struct U8Params {
    const uint8_t* img;      // [B][Hp][Wp][3], the valid part in the top-left h x w corner of each image's slot
    const uint8_t* mask;     // [B][Hp][Wp]
    const int32_t* sizes;    // [B][2] = (h, w) of each image, or null: every image is Hp x Wp
    const float* pred;       // blend: [B,3,Hp,Wp]
    long long pred_bs;
    float* out;              // compose: [B,4,Hp,Wp] fp32
    long long out_bs;
    uint8_t* out_u8;         // blend + quantise: [B][Hp][Wp][3]
    int B, Hp, Wp, binarize;
};
__device__ __forceinline__ bool u8_src(const U8Params& p, int b, int y, int x, long long& pix) {
    int h = p.Hp, w = p.Wp;
    if (p.sizes) { h = p.sizes[2 * b]; w = p.sizes[2 * b + 1]; }
    if (h <= 0 || w <= 0) return false;                  // an empty slot of a partial batch: zeros
    const int ys = y < h ? y : 2 * h - 1 - y, xs = x < w ? x : 2 * w - 1 - x;
    pix = ((long long)b * p.Hp + (ys < 0 ? 0 : ys)) * p.Wp + (xs < 0 ? 0 : xs);
    return true;
}
// masked_img = cat(img * (1 - mask), mask)  (trainers/default.py:59,67-68) from the u8 operands
__global__ __launch_bounds__(LAMA_NTHREADS) void mask_compose_u8_kernel(U8Params p) {
    const long long hw = (long long)p.Hp * p.Wp, total = (long long)p.B * hw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / hw);
        const long long px = i - (long long)b * hw;
        const int y = (int)(px / p.Wp), x = (int)(px - (long long)y * p.Wp);
        long long sp;
        float r = 0.f, g = 0.f, bl = 0.f, m = 0.f;
        if (u8_src(p, b, y, x, sp)) {
            const uint8_t* im = p.img + sp * 3;
            r = (float)im[0] / 255.0f; g = (float)im[1] / 255.0f; bl = (float)im[2] / 255.0f;
            const uint8_t mu = p.mask[sp];
            m = p.binarize ? (mu > 0 ? 1.0f : 0.0f) : (float)mu / 255.0f;
        }
        float* o = p.out + b * p.out_bs + px;
        const float k = 1.0f - m;
        o[0] = r * k;
        o[hw] = g * k;
        o[2 * hw] = bl * k;
        o[3 * hw] = m;
    }
}
// inpainted = mask * pred + (1 - mask) * img (default.py:71), then np.clip(x * 255, 0, 255).astype('uint8') in HWC order (bin/predict.py:86-92)
__global__ __launch_bounds__(LAMA_NTHREADS) void blend_quantize_u8_kernel(U8Params p) {
    const long long hw = (long long)p.Hp * p.Wp, total = (long long)p.B * hw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / hw);
        const long long px = i - (long long)b * hw;
        const int y = (int)(px / p.Wp), x = (int)(px - (long long)y * p.Wp);
        long long sp;
        float im[3] = {0.f, 0.f, 0.f}, m = 0.f;
        if (u8_src(p, b, y, x, sp)) {
            const uint8_t* s = p.img + sp * 3;
            im[0] = (float)s[0] / 255.0f; im[1] = (float)s[1] / 255.0f; im[2] = (float)s[2] / 255.0f;
            const uint8_t mu = p.mask[sp];
            m = p.binarize ? (mu > 0 ? 1.0f : 0.0f) : (float)mu / 255.0f;
        }
        const float k = 1.0f - m;
        uint8_t* d = p.out_u8 + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = (m * p.pred[b * p.pred_bs + c * hw + px] + k * im[c]) * 255.0f;
            v = fminf(fmaxf(v, 0.0f), 255.0f);
            d[c] = (uint8_t)v;
        }
    }
}

struct AffineParams {
    const float* x;
    long long x_bs;
    const float* scale;
    const float* shift;
    float* y;
    long long y_bs;
    long long hw;
    int C, B, act;
};

// y = act(x*scale[c] + shift[c]): stand-alone BatchNorm2d(eval) / ReLU / Sigmoid / Tanh layers
// (generator.model[25], [26], [35] ... when a caller runs the Sequential layer by layer)
__global__ __launch_bounds__(LAMA_NTHREADS) void affine_act_kernel(AffineParams p) {
    long long per = (long long)p.C * p.hw, total = per * p.B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int b = (int)(i / per);
        long long r = i - (long long)b * per;
        int c = (int)(r / p.hw);
        float v = p.x[b * p.x_bs + r];
        if (p.scale) v = v * p.scale[c] + p.shift[c];
        if (p.act == LAMA_ACT_RELU) v = fmaxf(v, 0.0f);
        else if (p.act == LAMA_ACT_SIGMOID) v = 1.0f / (1.0f + expf(-v));
        else if (p.act == LAMA_ACT_TANH) v = tanhf(v);
        p.y[b * p.y_bs + r] = v;
    }
}

struct PadParams {
    const float* x;
    long long x_bs;
    float* y;
    long long y_bs;
    int C, B, H, W, pad;
};

// nn.ReflectionPad2d(pad) as a stand-alone layer (generator.model[0], [33])
__global__ __launch_bounds__(LAMA_NTHREADS) void reflect_pad_kernel(PadParams p) {
    const int Ho = p.H + 2 * p.pad, Wo = p.W + 2 * p.pad;
    long long per = (long long)p.C * Ho * Wo, total = per * p.B;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int b = (int)(i / per);
        long long r = i - (long long)b * per;
        int xo = (int)(r % Wo);
        long long r2 = r / Wo;
        int yo = (int)(r2 % Ho);
        int c = (int)(r2 / Ho);
        int yi = yo - p.pad, xi = xo - p.pad;
        if (yi < 0) yi = -yi;
        if (yi >= p.H) yi = 2 * (p.H - 1) - yi;
        if (xi < 0) xi = -xi;
        if (xi >= p.W) xi = 2 * (p.W - 1) - xi;
        p.y[b * p.y_bs + r] = p.x[b * p.x_bs + ((long long)c * p.H + yi) * p.W + xi];
    }
}

namespace {
int ew_grid(long long total) {
    long long g = (total + LAMA_NTHREADS - 1) / LAMA_NTHREADS;
    return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}
bool same_hw(const lama_tensor* a, const lama_tensor* b) { return a->H == b->H && a->W == b->W; }
// the glue kernels around the generator (image / mask side) are fp32 only; the fp16 activation path (LAMA_DT_F16) starts at the
// stem's output and ends at the head's input
bool all_f32(std::initializer_list<const lama_tensor*> ts) {
    for (const lama_tensor* t : ts)
        if (t && t->ptr && t->dtype != LAMA_DT_F32) return false;
    return true;
}
}  // namespace

extern "C" int lama_version(void) { return LAMA_HIP_VERSION; }

extern "C" const char* lama_error_string(int code) {
    switch (code) {
        case LAMA_OK: return "ok";
        case LAMA_ERR_BAD_ARG: return "bad argument";
        case LAMA_ERR_UNSUPPORTED: return "unsupported shape or option";
        case LAMA_ERR_WORKSPACE: return "workspace missing or too small";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

extern "C" int lama_mask_compose_fwd(void* stream, const lama_tensor* image, const lama_tensor* mask,
                                     const lama_tensor* out, int32_t batch) {
    if (!image || !mask || !out || !image->ptr || !mask->ptr || !out->ptr || batch <= 0) return LAMA_ERR_BAD_ARG;
    if (image->C != 3 || mask->C != 1 || out->C != 4 || !same_hw(image, mask) || !same_hw(image, out)) return LAMA_ERR_BAD_ARG;
    if (!all_f32({image, mask, out})) return LAMA_ERR_UNSUPPORTED;
    EwParams p;
    memset(&p, 0, sizeof(p));
    p.img = (const float*)image->ptr; p.img_bs = image->batch_stride;
    p.mask = (const float*)mask->ptr; p.mask_bs = mask->batch_stride;
    p.out = (float*)out->ptr; p.out_bs = out->batch_stride;
    p.hw = (long long)image->H * image->W;
    p.B = batch;
    hipLaunchKernelGGL(mask_compose_kernel, dim3(ew_grid(p.B * p.hw)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

extern "C" int lama_blend_fwd(void* stream, const lama_tensor* image, const lama_tensor* mask, const lama_tensor* pred,
                              const lama_tensor* out, int32_t batch) {
    if (!image || !mask || !pred || !out || !image->ptr || !mask->ptr || !pred->ptr || !out->ptr || batch <= 0) return LAMA_ERR_BAD_ARG;
    if (image->C != 3 || mask->C != 1 || pred->C != 3 || out->C != 3) return LAMA_ERR_BAD_ARG;
    if (!same_hw(image, mask) || !same_hw(image, pred) || !same_hw(image, out)) return LAMA_ERR_BAD_ARG;
    if (!all_f32({image, mask, pred, out})) return LAMA_ERR_UNSUPPORTED;
    EwParams p;
    memset(&p, 0, sizeof(p));
    p.img = (const float*)image->ptr; p.img_bs = image->batch_stride;
    p.mask = (const float*)mask->ptr; p.mask_bs = mask->batch_stride;
    p.pred = (const float*)pred->ptr; p.pred_bs = pred->batch_stride;
    p.out = (float*)out->ptr; p.out_bs = out->batch_stride;
    p.hw = (long long)image->H * image->W;
    p.B = batch;
    hipLaunchKernelGGL(blend_kernel, dim3(ew_grid(p.B * p.hw)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

extern "C" int lama_quantize_u8_hwc_fwd(void* stream, const lama_tensor* src, uint8_t* dst, int32_t batch,
                                        int32_t crop_h, int32_t crop_w) {
    if (!src || !src->ptr || !dst || batch <= 0 || src->C != 3) return LAMA_ERR_BAD_ARG;
    if (crop_h <= 0 || crop_w <= 0 || crop_h > src->H || crop_w > src->W) return LAMA_ERR_BAD_ARG;
    if (!all_f32({src})) return LAMA_ERR_UNSUPPORTED;
    QuantParams p;
    p.src = (const float*)src->ptr; p.src_bs = src->batch_stride;
    p.dst = dst;
    p.B = batch; p.H = src->H; p.W = src->W; p.ch = crop_h; p.cw = crop_w;
    hipLaunchKernelGGL(quantize_u8_hwc_kernel, dim3(ew_grid((long long)batch * crop_h * crop_w * 3)), dim3(LAMA_NTHREADS), 0,
                       (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

// (v110) the two ends of the predict step on u8 operands: see mask_compose_u8_kernel / blend_quantize_u8_kernel
extern "C" int lama_mask_compose_u8_fwd(void* stream, const uint8_t* image_hwc, const uint8_t* mask, const int32_t* sizes, const lama_tensor* out,
                                        int32_t batch, int32_t binarize) {
    if (!image_hwc || !mask || !out || !out->ptr || batch <= 0 || out->C != 4 || out->H <= 0 || out->W <= 0) return LAMA_ERR_BAD_ARG;
    if (!all_f32({out})) return LAMA_ERR_UNSUPPORTED;
    U8Params p;
    memset(&p, 0, sizeof(p));
    p.img = image_hwc; p.mask = mask; p.sizes = sizes;
    p.out = (float*)out->ptr; p.out_bs = out->batch_stride;
    p.B = batch; p.Hp = out->H; p.Wp = out->W; p.binarize = binarize;
    hipLaunchKernelGGL(mask_compose_u8_kernel, dim3(ew_grid((long long)batch * out->H * out->W)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

extern "C" int lama_blend_quantize_u8_fwd(void* stream, const uint8_t* image_hwc, const uint8_t* mask, const int32_t* sizes, const lama_tensor* pred,
                                          uint8_t* dst_hwc, int32_t batch, int32_t binarize) {
    if (!image_hwc || !mask || !pred || !pred->ptr || !dst_hwc || batch <= 0 || pred->C != 3 || pred->H <= 0 || pred->W <= 0) return LAMA_ERR_BAD_ARG;
    if (!all_f32({pred})) return LAMA_ERR_UNSUPPORTED;
    U8Params p;
    memset(&p, 0, sizeof(p));
    p.img = image_hwc; p.mask = mask; p.sizes = sizes;
    p.pred = (const float*)pred->ptr; p.pred_bs = pred->batch_stride;
    p.out_u8 = dst_hwc;
    p.B = batch; p.Hp = pred->H; p.Wp = pred->W; p.binarize = binarize;
    hipLaunchKernelGGL(blend_quantize_u8_kernel, dim3(ew_grid((long long)batch * pred->H * pred->W)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

extern "C" int lama_affine_act_fwd(void* stream, const lama_tensor* x, const float* scale, const float* shift, int32_t act,
                                   const lama_tensor* y, int32_t batch) {
    if (!x || !y || !x->ptr || !y->ptr || batch <= 0) return LAMA_ERR_BAD_ARG;
    if (x->C != y->C || !same_hw(x, y) || ((scale == nullptr) != (shift == nullptr))) return LAMA_ERR_BAD_ARG;
    if (!all_f32({x, y})) return LAMA_ERR_UNSUPPORTED;
    AffineParams p;
    p.x = (const float*)x->ptr; p.x_bs = x->batch_stride;
    p.scale = scale; p.shift = shift;
    p.y = (float*)y->ptr; p.y_bs = y->batch_stride;
    p.hw = (long long)x->H * x->W;
    p.C = x->C; p.B = batch; p.act = act;
    hipLaunchKernelGGL(affine_act_kernel, dim3(ew_grid(p.hw * p.C * p.B)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

extern "C" int lama_reflect_pad_fwd(void* stream, const lama_tensor* x, int32_t pad, const lama_tensor* y, int32_t batch) {
    if (!x || !y || !x->ptr || !y->ptr || batch <= 0 || pad < 0) return LAMA_ERR_BAD_ARG;
    if (x->C != y->C || y->H != x->H + 2 * pad || y->W != x->W + 2 * pad || pad >= x->H || pad >= x->W) return LAMA_ERR_BAD_ARG;
    if (!all_f32({x, y})) return LAMA_ERR_UNSUPPORTED;
    PadParams p;
    p.x = (const float*)x->ptr; p.x_bs = x->batch_stride;
    p.y = (float*)y->ptr; p.y_bs = y->batch_stride;
    p.C = x->C; p.B = batch; p.H = x->H; p.W = x->W; p.pad = pad;
    hipLaunchKernelGGL(reflect_pad_kernel, dim3(ew_grid((long long)batch * y->C * y->H * y->W)), dim3(LAMA_NTHREADS), 0, (hipStream_t)stream, p);
    LAMA_CHECK_LAUNCH();
    return LAMA_OK;
}

// ---- FourierUnit.forward (ffc.py:76-113) as three launches over caller workspace ------------------
extern "C" size_t lama_fourier_unit_workspace_bytes(int32_t batch, int32_t C, int32_t h, int32_t w) {
    if (batch <= 0 || C <= 0 || h <= 0 || w <= 0) return 0;
    size_t spec = (size_t)batch * 2 * C * h * (w / 2 + 1) * sizeof(float);
    spec = (spec + 255) & ~(size_t)255;
    return 2 * spec + lama_fft_workspace_bytes(batch, C, h, w);
}

static int fourier_unit_impl(void* stream, const lama_tensor* x, const void* w_packed, const float* bias,
                             const lama_tensor* y, int32_t batch, int32_t add_input, int32_t precision,
                             void* workspace, size_t workspace_bytes, uint32_t* range_flag,
                             const lama_conv2d_args* wino_args, void* wino_ws, size_t wino_ws_bytes, int32_t flags = 0) {
    if (!x || !y || !x->ptr || !y->ptr || !w_packed || batch <= 0) return LAMA_ERR_BAD_ARG;
    if (x->C != y->C || x->H != y->H || x->W != y->W) return LAMA_ERR_BAD_ARG;
    const int C = x->C, h = x->H, w = x->W, wf = w / 2 + 1;
    if (!workspace || workspace_bytes < lama_fourier_unit_workspace_bytes(batch, C, h, w)) return LAMA_ERR_WORKSPACE;
    size_t spec_bytes = ((size_t)batch * 2 * C * h * wf * sizeof(float) + 255) & ~(size_t)255;
    char* ws = (char*)workspace;
    if (x->dtype != y->dtype) return LAMA_ERR_UNSUPPORTED;
    lama_tensor s1 = {ws, (int64_t)2 * C * h * wf, 2 * C, h, wf, x->dtype};
    lama_tensor s2 = {ws + spec_bytes, (int64_t)2 * C * h * wf, 2 * C, h, wf, x->dtype};
    void* fws = ws + 2 * spec_bytes;
    size_t fws_bytes = workspace_bytes - 2 * spec_bytes;
    int rc;
    if (wino_args) {      // the deferred output transform of the Winograd conv of the layer before: inside the rfft2 launch where a kernel does that
        rc = lama_rfft2_winograd_out_fwd(stream, x, &s1, batch, fws, fws_bytes, wino_args, wino_ws, wino_ws_bytes);
        if (rc == LAMA_ERR_UNSUPPORTED) {
            rc = lama_winograd_out_fwd(stream, wino_args, wino_ws, wino_ws_bytes);
            if (rc) return rc;
            rc = lama_rfft2_fwd(stream, x, &s1, batch, fws, fws_bytes);
        }
    } else {
        rc = lama_rfft2_fwd(stream, x, &s1, batch, fws, fws_bytes);
    }
    if (rc) return rc;
    lama_conv2d_args a;
    memset(&a, 0, sizeof(a));
    a.x = s1;
    a.w_packed = w_packed;
    a.kh = a.kw = 1;
    a.stride = 1;
    a.bias = bias;
    a.act = LAMA_ACT_RELU;
    a.y = s2;
    a.batch = batch;
    a.precision = precision;
    a.range_flag = range_flag;
    a.flags = flags & LAMA_CONV_SIBLINGS_MASK;      // a part of the batch beside its siblings: the whole batch's kernel (v109)
    rc = lama_conv2d_fwd(stream, &a);
    if (rc) return rc;
    return lama_irfft2_fwd(stream, &s2, add_input ? x : nullptr, y, batch, fws, fws_bytes);
}

extern "C" int lama_fourier_unit_fwd(void* stream, const lama_tensor* x, const void* w_packed, const float* bias,
                                     const lama_tensor* y, int32_t batch, int32_t add_input, int32_t precision,
                                     void* workspace, size_t workspace_bytes, uint32_t* range_flag) {
    return fourier_unit_impl(stream, x, w_packed, bias, y, batch, add_input, precision, workspace, workspace_bytes, range_flag, nullptr, nullptr, 0);
}

// (v108) FourierUnit.forward of layer l + 1 that also finishes the Winograd local conv of layer l (lama_winograd_conv3x3_fwd with
// LAMA_CONV_DEFER_OUT): its output transform rides in the rfft2 launch (lama_rfft2_winograd_out_fwd) or, where no kernel does that, runs first
extern "C" int lama_fourier_unit_winograd_out_fwd(void* stream, const lama_tensor* x, const void* w_packed, const float* bias,
                                                  const lama_tensor* y, int32_t batch, int32_t add_input, int32_t precision,
                                                  void* workspace, size_t workspace_bytes, uint32_t* range_flag,
                                                  const lama_conv2d_args* wino_args, void* wino_workspace, size_t wino_workspace_bytes) {
    if (!wino_args || !wino_workspace) return LAMA_ERR_BAD_ARG;
    return fourier_unit_impl(stream, x, w_packed, bias, y, batch, add_input, precision, workspace, workspace_bytes, range_flag, wino_args,
                             wino_workspace, wino_workspace_bytes);
}

// (v109) both of the above + the LAMA_CONV_* flags of the spectral 1x1 launch
extern "C" int lama_fourier_unit_ex_fwd(void* stream, const lama_tensor* x, const void* w_packed, const float* bias, const lama_tensor* y,
                                        int32_t batch, int32_t add_input, int32_t precision, void* workspace, size_t workspace_bytes,
                                        uint32_t* range_flag, int32_t flags, const lama_conv2d_args* wino_args, void* wino_workspace,
                                        size_t wino_workspace_bytes) {
    if (wino_args && !wino_workspace) return LAMA_ERR_BAD_ARG;
    return fourier_unit_impl(stream, x, w_packed, bias, y, batch, add_input, precision, workspace, workspace_bytes, range_flag, wino_args,
                             wino_workspace, wino_workspace_bytes, flags);
}

