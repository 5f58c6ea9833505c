/* lama_hip.h -- C ABI of liblama_hip.so: the MI355X (gfx950) kernels behind the LaMa FFC generator.
 *
 * The reference (advimman/lama) has no FFI: its hot path is Python calling PyTorch ops.  This header
 * is the boundary a maintainer binds instead (ctypes stub in INTEGRATION.md).  Each entry point
 * names the reference code whose arithmetic it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = LAMA_ERR_* (bad argument / unsupported shape),
 *     >0 = hipError_t of the failing runtime call.  Nothing throws, exits or synchronises the host.
 *   - all pointers are DEVICE pointers owned by the caller (inputs, outputs, packed weights and
 *     workspace).  The library allocates nothing and keeps no state; every launch is asynchronous on
 *     the given hipStream_t (passed as void*) and is hipGraph-capturable.  The library reads no environment variable
 *     (kernel-selection overrides, ablations and tracers exist only in the -DLAMA_PROFILING build used by tools/).
 *   - activations are fp32 NCHW "views": element (b,c,y,x) of a lama_tensor lives at
 *     ptr[b*batch_stride + (c*H + y)*W + x]; batch_stride lets a view address a channel slice of a
 *     wider buffer (the bottleneck state keeps x_l | x_g in one 512-channel buffer, which makes
 *     ConcatTupleLayer, ffc.py:295-302, free).
 */
#ifndef LAMA_HIP_H
#define LAMA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LAMA_HIP_VERSION 110

#define LAMA_OK 0
#define LAMA_ERR_BAD_ARG (-1)
#define LAMA_ERR_UNSUPPORTED (-2)
#define LAMA_ERR_WORKSPACE (-3)

/* activation applied in a conv epilogue */
#define LAMA_ACT_NONE 0
#define LAMA_ACT_RELU 1
#define LAMA_ACT_SIGMOID 2
#define LAMA_ACT_TANH 3

/* padding mode of a conv */
#define LAMA_PAD_ZERO 0
#define LAMA_PAD_REFLECT 1

/* arithmetic of the MFMA contraction */
#define LAMA_PREC_F32 0    /* v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain)          */
#define LAMA_PREC_BF16X3 1 /* 3-term bf16 split (hi*hi + hi*lo + lo*hi) on v_mfma_f32_32x32x16_bf16: fp32 range  */
#define LAMA_PREC_F16X3 2  /* 3-term fp16 split on v_mfma_f32_32x32x16_f16: 22 mantissa bits, |x| <= 65504     */
#define LAMA_PREC_F16 3    /* BASELINE configs[2] "fp16": fp16 activations in HBM (lama_tensor.dtype = LAMA_DT_F16), fp16 weights (the hi
                            * part of the LAMA_PREC_F16X3 packing), ONE v_mfma_f32_32x32x16_f16 product per MAC, fp32 accumulation,
                            * fp32 epilogue (bias, activation, residual), fp16 store.  x, x2 share one element type, resid and y share
                            * one; fp32 -> fp16 (the stem) and fp16 -> fp32 (the head) are allowed.                                    */

/* element type of an activation tensor in HBM */
#define LAMA_DT_F32 0
#define LAMA_DT_F16 1 /* IEEE half: BASELINE configs[2] "fp16"; every tensor of a call is fp16 except where an entry says otherwise */

typedef struct lama_tensor {
    void* ptr;            /* device pointer to element (0,0,0,0) of the view; NULL = absent          */
    int64_t batch_stride; /* ELEMENTS between consecutive images                                     */
    int32_t C, H, W;
    int32_t dtype;        /* LAMA_DT_F32 (0) or LAMA_DT_F16                                          */
} lama_tensor;

/* One fused convolution launch:
 *     y = act( conv(x, w) [+ conv1x1(x2, w2)] + bias ) [+ resid]
 * Replaces nn.Conv2d(padding_mode='reflect') + BatchNorm2d(eval) + ReLU (+ residual add) chains:
 *   - FFC.convl2l/convl2g/convg2l 3x3 reflect convs, ffc.py:188-196, 220-223
 *   - FFC_BN_ACT bn_l/bn_g + act, ffc.py:251-255 (BN scale folded into w at pack time, shift = bias)
 *   - SpectralTransform.conv1 / conv2 and FourierUnit.conv_layer 1x1 convs, ffc.py:57-59,128-140
 *   - FFCResnetBlock residual add, ffc.py:288 (resid)
 *   - stem 7x7 / stride-2 downsamples / ConvTranspose2d upsamples / 7x7 head + sigmoid,
 *     ffc.py:315-363 (transposed = 1 means ConvTranspose2d(k3,s2,p1,op1))
 * x2/w2: optional second operand contracted with a 1x1 kernel into the same accumulator
 *        (out_xg = convl2g(x_l) + convg2g.conv2(t), ffc.py:161,223 in one pass).
 */
typedef struct lama_conv2d_args {
    lama_tensor x;
    const void* w_packed; /* from lama_conv2d_pack_weight                                  */
    int32_t kh, kw, stride, pad, pad_mode, transposed;
    lama_tensor x2;
    const void* w2_packed;
    const float* bias; /* [Cout] or NULL                                                 */
    int32_t act;
    lama_tensor resid; /* added after the activation; ptr NULL = none                    */
    lama_tensor y;
    int32_t batch;
    int32_t precision; /* LAMA_PREC_*                                                    */
    /* Range watch of LAMA_PREC_F16X3 (ignored by the other precisions; NULL = off).  The fp16 split represents |x| <= 65504
     * only: a kernel that meets a larger activation (or a NaN) while splitting x / x2 ORs 1 into *range_flag (device
     * uint32, zeroed by the caller); its output is then garbage and the caller re-runs with LAMA_PREC_BF16X3 (fp32 exponent
     * range, 16 mantissa bits) or LAMA_PREC_F32.  The weights are range-checked by the host at pack time instead. */
    uint32_t* range_flag;
    /* Optional fused second stage (NULL / ptr NULL = off): SpectralTransform.conv1 of the NEXT layer, ffc.py:128-133,145, i.e.
     *     fuse1_y = ReLU( W1 y + fuse1_bias ),  W1 [192, 384] 1x1,
     * computed in the epilogue of the launch that produces the 384-channel x_g state y (the global branch of an FFC layer:
     * 3x3 over x_l + fused 1x1 over x2, Cout = 384, fp32 tensors, LAMA_PREC_F16X3 / BF16X3, planes of at least one 128-pixel
     * tile per CU; anything else returns LAMA_ERR_UNSUPPORTED when fuse1_w is set -- the caller then launches conv1 itself).
     * fuse1_w = lama_conv2d_pack_weight of W1 (BatchNorm scale folded) whose INPUT channels were reordered with
     * lama_fuse1_channel_order (the accumulator layout of the producing kernel); fuse1_y [B,192,H,W] fp32. */
    const void* fuse1_w;
    const float* fuse1_bias;
    lama_tensor fuse1_y;
    /* LAMA_CONV_* bits.  LAMA_CONV_COOPERATIVE: another stream has work for this GPU while this launch runs (the spectral branch of the
     * same FFC layer): kernels that would fill every CU's register file (the local 3x3 conv: two waves per SIMD) take a geometry that
     * leaves room for the other stream's workgroups (one 4-wave workgroup per CU, taken when the launch has 224 ... 320 workgroups) -- a few
     * per cent slower alone, faster together.  Results equal the plain launch up to fp32 summation order. */
    int32_t flags;
} lama_conv2d_args;
#define LAMA_CONV_COOPERATIVE 1
/* (v109) bits 8..10: log2 of the number of SIBLING launches that run at the same time as this one -- the parts of a batch that the host runs as
 * parallel branches of one hipGraph (lama_amd: generator.split_batch).  A launch over 1 / 2^n of the batch then takes the kernel geometry the WHOLE
 * batch would take (one 12-wave workgroup per 128-pixel tile of the global branch, with the optional fused conv1; the all-of-K-in-one-wave spectral
 * GEMM; no split-K in the Winograd conv) instead of the geometry that lets a lone small launch fill the chip: n siblings fill it together.
 * Same results as the unflagged launch up to fp32 summation order -- and bit-identical to the launch over the whole batch. */
#define LAMA_CONV_SIBLINGS_SHIFT 8
#define LAMA_CONV_SIBLINGS_MASK (7 << LAMA_CONV_SIBLINGS_SHIFT)
#define LAMA_CONV_SIBLINGS_LOG2(flags) (((flags) & LAMA_CONV_SIBLINGS_MASK) >> LAMA_CONV_SIBLINGS_SHIFT)
/* (v108, lama_winograd_conv3x3_fwd only) launch the GEMM half only; the caller owes the output transform -- lama_winograd_out_fwd or
 * lama_rfft2_winograd_out_fwd with the SAME arguments and workspace -- before anything reads y */
#define LAMA_CONV_DEFER_OUT 2

/* order[k] = the input channel of conv1 that sits at packed K position k (0 <= k < 384) of lama_conv2d_args.fuse1_w */
void lama_fuse1_channel_order(int32_t* order);

int lama_version(void);
const char* lama_error_string(int code);

/* Packed-weight size in BYTES for a conv with the given geometry. */
int64_t lama_conv2d_packed_weight_bytes(int32_t cout, int32_t cin, int32_t kh, int32_t kw, int32_t stride,
                                        int32_t transposed, int32_t precision);
/* Repack reference-layout weights (Conv2d: [Cout,Cin,kh,kw]; ConvTranspose2d: [Cin,Cout,kh,kw], both
 * as stored in the checkpoint) into the kernel's K-major layout, multiplying
 * output channel o by scale[o] (folded BatchNorm gamma/sqrt(var+eps); NULL = 1). */
int lama_conv2d_pack_weight(void* stream, const float* w, const float* scale, int32_t cout, int32_t cin, int32_t kh,
                            int32_t kw, int32_t stride, int32_t transposed, int32_t precision, void* dst);
int lama_conv2d_fwd(void* stream, const lama_conv2d_args* args);

/* Winograd F(2x2, 3x3) form of a 3x3 / stride 1 / reflect-pad 1 convolution: the SAME function as lama_conv2d_fwd on such a layer
 * (FFC.convl2l + convg2l over the 512-channel bottleneck state -> the 128 local output channels, ffc.py:188-196,220, with the folded
 * BatchNorm + ReLU of FFC_BN_ACT, ffc.py:251-255, and the FFCResnetBlock residual, ffc.py:288), computed with 16 instead of 36
 * multiplies per 2 x 2 output tile and (output, input) channel pair -- the launch is bound by the power the matrix cores draw, so
 * fewer MFMA products is what shortens it.  Split precisions only (LAMA_PREC_F16X3 / BF16X3; the 3-term split is applied to the
 * transformed operands: 1.2e-5 max-abs end to end with every such conv of the generator in this form); fp32 tensors; Cin % 32 == 0,
 * Cout % 128 == 0, H >= 4, W >= 8 (v110: any plane size -- tile rows are cut into overlapping segments of 8 / 16 / 32 column quads and the
 * columns behind W - 1 inside the last quad are reflected in registers; W in {32, 64, 128, 256} with H a multiple of 512 / W keep the exact
 * geometry of v107).  Anything else: LAMA_ERR_UNSUPPORTED (use lama_conv2d_fwd).
 *   lama_winograd_pack_weight: Conv2d weights [Cout, Cin, 3, 3] -> U = G g G^T per channel pair, BatchNorm scale folded, (hi, lo)
 *     split, MFMA A-fragment order; lama_winograd_packed_weight_bytes bytes.
 *   lama_winograd_conv3x3_fwd: args as lama_conv2d_fwd (x, w_packed, bias, act, resid, y, batch, precision, range_flag; kh = kw = 3,
 *     stride = 1, pad = 1, pad_mode = LAMA_PAD_REFLECT, or (v108) LAMA_PAD_ZERO: the dgrad convs of the reverse pass); workspace = lama_winograd_workspace_bytes device bytes (the half-inverted
 *     transform-domain sums between its two launches).  The exact geometries want x, y, resid as 16-byte aligned pointers with batch strides
 *     that are multiples of 4 elements (else LAMA_ERR_UNSUPPORTED); the any-size geometry needs element alignment only.
 *   lama_winograd_supported: 1 when lama_winograd_conv3x3_fwd takes this (cout, cin, H, W, precision) -- every bound of the launch,
 *     including the 32-bit byte offsets inside one image (cin * H * W * 4 < 2^31), which the workspace query alone does not see.
 *   lama_winograd_preferred (v110): 1 when this form is expected to beat lama_conv2d_fwd for (batch, cout, cin, H, W) -- always for the exact
 *     geometries; for the any-size geometry a cost model of whole rounds of workgroups (one per CU) against the direct kernel's: a launch of a
 *     little more than 256 units takes two rounds and loses (1 x 135 x 240), one of a little less wins (1 x 168 x 168: 65 against 92 us).
 *   range_flag (LAMA_PREC_F16X3): NOT the same watch as lama_conv2d_fwd's.  It is kept on the row sums r of the input transform
 *     (|V| <= 2 max|r|) and is raised at 2 max|r| >= 65504, i.e. up to 2x EARLY against a real overflow of a transformed operand, and it
 *     is BLIND TO NaN inputs (fmaxf drops them).  Inside the generator the global-branch launch (lama_conv2d_fwd) reads the same state
 *     buffer in the same layer and raises the flag for NaN / |x| > 65504; a stand-alone caller that needs NaN detection must run its own
 *     check (or lama_conv2d_fwd) on x. */
int64_t lama_winograd_packed_weight_bytes(int32_t cout, int32_t cin, int32_t precision);
int32_t lama_winograd_supported(int32_t cout, int32_t cin, int32_t H, int32_t W, int32_t precision);
int32_t lama_winograd_preferred(int32_t batch, int32_t cout, int32_t cin, int32_t H, int32_t W, int32_t precision);
int lama_winograd_pack_weight(void* stream, const float* w, const float* scale, int32_t cout, int32_t cin, int32_t precision, void* dst);
size_t lama_winograd_workspace_bytes(int32_t batch, int32_t cout, int32_t H, int32_t W);
int lama_winograd_conv3x3_fwd(void* stream, const lama_conv2d_args* args, void* workspace, size_t workspace_bytes);
/* (v108) the second launch of lama_winograd_conv3x3_fwd on its own (after a call with LAMA_CONV_DEFER_OUT), and the same work inside the
 * rfft2 launch of the NEXT FFC layer (ffc.py:86 of layer l + 1 with ffc.py:220,251-255,288 of layer l): both are memory-bound and independent, and
 * the FFT workgroups leave the HBM idle while they transform.  64 x 64 fp32 planes only (LAMA_ERR_UNSUPPORTED otherwise, nothing launched). */
/* lama_fourier_unit_winograd_out_fwd: lama_fourier_unit_fwd whose first launch is lama_rfft2_winograd_out_fwd (or, where that is unsupported,
 * lama_winograd_out_fwd followed by the plain FourierUnit): what the host mirror calls for layer l + 1 when layer l deferred its transform. */
int lama_winograd_out_fwd(void* stream, const lama_conv2d_args* args, void* workspace, size_t workspace_bytes);
int lama_rfft2_winograd_out_fwd(void* stream, const lama_tensor* x, const lama_tensor* spec, int32_t batch, void* workspace, size_t workspace_bytes,
                                const lama_conv2d_args* wino_args, void* wino_workspace, size_t wino_workspace_bytes);
int lama_fourier_unit_winograd_out_fwd(void* stream, const lama_tensor* x, const void* w_packed, const float* bias, const lama_tensor* y,
                                       int32_t batch, int32_t add_input, int32_t precision, void* workspace, size_t workspace_bytes,
                                       uint32_t* range_flag, const lama_conv2d_args* wino_args, void* wino_workspace,
                                       size_t wino_workspace_bytes);

/* torch.fft.rfftn(x, dim=(-2,-1), norm='ortho') followed by the Re/Im channel interleave
 * (ffc.py:86-89): x [B,C,h,w] -> spec [B,2C,h,w/2+1], channel 2c = Re, 2c+1 = Im. */
int lama_rfft2_fwd(void* stream, const lama_tensor* x, const lama_tensor* spec, int32_t batch, void* workspace,
                   size_t workspace_bytes);
/* de-interleave + torch.fft.irfftn(s=(h,w), dim=(-2,-1), norm='ortho') on a NON-Hermitian spectrum
 * (ffc.py:103-108; complex inverse along h, then c2r along w ignoring Im of bins 0 and w/2), fused
 * with the `x + fu(x)` add of SpectralTransform.forward (ffc.py:161): y = resid + irfft2(spec).
 * resid may be absent.  y [B,C,h,w]. */
int lama_irfft2_fwd(void* stream, const lama_tensor* spec, const lama_tensor* resid, const lama_tensor* y,
                    int32_t batch, void* workspace, size_t workspace_bytes);
/* (v108) the same transforms with the ReLU derivative that follows them in the reverse pass of the FourierUnit (refinement.py:163 through
 * ffc.py:101 and ffc.py:131): out = transform(...) * [mask_y > 0], mask_y laid out like the output (spec for the forward transform, y for the
 * inverse one).  Only where the compile-time two-pass kernels run (fp32 planes of 256 x 256, 16-byte aligned views): LAMA_ERR_UNSUPPORTED
 * otherwise, with nothing launched -- the caller then runs the plain transform + lama_act_bwd. */
int lama_rfft2_masked_fwd(void* stream, const lama_tensor* x, const lama_tensor* spec, const lama_tensor* mask_y, int32_t batch,
                          void* workspace, size_t workspace_bytes);
int lama_irfft2_masked_fwd(void* stream, const lama_tensor* spec, const lama_tensor* resid, const lama_tensor* mask_y, const lama_tensor* y,
                           int32_t batch, void* workspace, size_t workspace_bytes);
/* scratch bytes lama_rfft2_fwd / lama_irfft2_fwd / lama_fourier_unit_fwd need for [B,C,h,w] */
size_t lama_fft_workspace_bytes(int32_t batch, int32_t C, int32_t h, int32_t w);

/* FourierUnit.forward (ffc.py:76-113) in one call: y = [x +] irfft2( relu( W' rfft2(x) + b ) ).
 * w_packed = lama_conv2d_pack_weight(conv_layer.weight [2C,2C,1,1], bn scale); bias = bn shift.
 * workspace >= lama_fourier_unit_workspace_bytes. add_input != 0 fuses SpectralTransform's x + fu(x).
 * range_flag: as in lama_conv2d_args (the spectrum's DC bin grows with sqrt(h*w): the first tensor to leave the fp16 range). */
size_t lama_fourier_unit_workspace_bytes(int32_t batch, int32_t C, int32_t h, int32_t w);
int lama_fourier_unit_fwd(void* stream, const lama_tensor* x, const void* w_packed, const float* bias,
                          const lama_tensor* y, int32_t batch, int32_t add_input, int32_t precision, void* workspace,
                          size_t workspace_bytes, uint32_t* range_flag);
/* (v109) the superset of lama_fourier_unit_fwd and lama_fourier_unit_winograd_out_fwd (wino_args NULL = no deferred output transform) with the
 * LAMA_CONV_* flags of its spectral 1x1 launch (ffc.py:100-101) -- LAMA_CONV_SIBLINGS_* when the FourierUnit runs on a part of the batch beside
 * its siblings. */
int lama_fourier_unit_ex_fwd(void* stream, const lama_tensor* x, const void* w_packed, const float* bias, const lama_tensor* y,
                             int32_t batch, int32_t add_input, int32_t precision, void* workspace, size_t workspace_bytes,
                             uint32_t* range_flag, int32_t flags, const lama_conv2d_args* wino_args, void* wino_workspace,
                             size_t wino_workspace_bytes);

/* masked_img = cat(img*(1-mask), mask), trainers/default.py:59,67-68.  img [B,3,H,W], mask [B,1,H,W] -> out [B,4,H,W] */
int lama_mask_compose_fwd(void* stream, const lama_tensor* image, const lama_tensor* mask, const lama_tensor* out,
                          int32_t batch);
/* inpainted = mask*pred + (1-mask)*img, trainers/default.py:71 */
int lama_blend_fwd(void* stream, const lama_tensor* image, const lama_tensor* mask, const lama_tensor* pred,
                   const lama_tensor* out, int32_t batch);
/* np.clip(x*255, 0, 255).astype(uint8) of the HWC-permuted, cropped result, bin/predict.py:86-92.
 * src [B,3,H,W] fp32 -> dst u8 [B, crop_h, crop_w, 3] (RGB). */
int lama_quantize_u8_hwc_fwd(void* stream, const lama_tensor* src, uint8_t* dst, int32_t batch, int32_t crop_h,
                             int32_t crop_w);

/* (v110) The two ends of the predict step on what is on DISK: u8 HWC image [B][Hp][Wp][3] + u8 mask [B][Hp][Wp] (device pointers), each image's
 * valid h x w region in the top-left corner of its slot, sizes = int32 [B][2] (h, w) on the device (NULL: every image fills its slot; h = 0: an
 * empty slot of a partial batch -> zeros).  The kernels do what the reference's host code does per image:
 *   image.astype('float32') / 255 (evaluation/data.py:12-20: the same IEEE fp32 division), symmetric padding bottom / right up to (Hp, Wp)
 *   (data.py:29-33, np.pad mode='symmetric': row h + i = row h - 1 - i), mask > 0 (bin/predict.py:84; binarize == 0: mask / 255),
 * so a step uploads 4 bytes per pixel instead of 16 and the host neither converts nor pads.
 *   lama_mask_compose_u8_fwd:   out [B,4,Hp,Wp] fp32 = cat(img * (1 - mask), mask)                      (trainers/default.py:59,67-68)
 *   lama_blend_quantize_u8_fwd: dst u8 [B][Hp][Wp][3] = clip((mask * pred + (1 - mask) * img) * 255, 0, 255) truncated   (default.py:71 + bin/predict.py:86-92)
 * Bit-identical to lama_mask_compose_fwd / lama_blend_fwd + lama_quantize_u8_hwc_fwd on the fp32 tensors the host would have built. */
int lama_mask_compose_u8_fwd(void* stream, const uint8_t* image_hwc, const uint8_t* mask, const int32_t* sizes, const lama_tensor* out,
                             int32_t batch, int32_t binarize);
int lama_blend_quantize_u8_fwd(void* stream, const uint8_t* image_hwc, const uint8_t* mask, const int32_t* sizes, const lama_tensor* pred,
                               uint8_t* dst_hwc, int32_t batch, int32_t binarize);

/* Stand-alone per-channel affine + activation: y = act(x*scale[c] + shift[c]) (scale/shift NULL = identity).
 * Used when a caller runs generator.model layer by layer and hits a bare nn.BatchNorm2d (eval) /
 * nn.ReLU / nn.Sigmoid (ffc.py:352-354,362-363); the fused forward never launches it. */
int lama_affine_act_fwd(void* stream, const lama_tensor* x, const float* scale, const float* shift, int32_t act,
                        const lama_tensor* y, int32_t batch);
/* Stand-alone nn.ReflectionPad2d(pad) (ffc.py:314,360); fused into the 7x7 convs on the normal path. */
int lama_reflect_pad_fwd(void* stream, const lama_tensor* x, int32_t pad, const lama_tensor* y, int32_t batch);

/* ------------------------------------------------------------------------------------------------
 * Feature refinement (saicinpainting/evaluation/refinement.py, `refine=True` of bin/predict.py:75-79): what is
 * not a convolution / FFT.  The dgrad of every conv / FourierUnit of generator.model[first_resblock:] reuses
 * lama_conv2d_fwd / lama_rfft2_fwd / lama_irfft2_fwd with transposed, flipped packed weights (lama_amd/backward.py);
 * autograd (refinement.py:163 loss.backward()) is replaced by that explicit reverse pass.
 * ------------------------------------------------------------------------------------------------ */
/* gout = g * act'(y) with the derivative expressed through the layer OUTPUT y: ReLU -> [y > 0], sigmoid -> y (1 - y),
 * tanh -> 1 - y^2, none -> 1.  Backward of the ReLU / Sigmoid of ffc.py:253-254,354,363 and of FourierUnit's ReLU (ffc.py:101). */
int lama_act_bwd(void* stream, const lama_tensor* g, const lama_tensor* y, int32_t act, const lama_tensor* gout, int32_t batch);
/* out = a + b: gradient joins of the residual / two-branch structure (ffc.py:161,220-223,288) */
int lama_add_fwd(void* stream, const lama_tensor* a, const lama_tensor* b, const lama_tensor* out, int32_t batch);
/* adjoint of nn.ReflectionPad2d(pad) / padding_mode='reflect' (ffc.py:188-196,314,360): gp [B,C,H+2p,W+2p] -> g [B,C,H,W]
 * (+ addend, optional): every padded position is added onto the input pixel it mirrors. */
int lama_reflect_pad_bwd(void* stream, const lama_tensor* gp, const lama_tensor* addend, int32_t pad, const lama_tensor* g,
                         int32_t batch);
/* (v108) the same adjoint fused with what the reverse pass does to its result: s = fold(gp) [+ add1] [+ add2]; g = s (optional, may be
 * add2 itself); gm = s * act'(mask_y) (optional).  add1: the 1x1 path into the same tensor (SpectralTransform.conv1, ffc.py:158), add2: the
 * identity path of the resnet block (ffc.py:288), mask_y: the taped output of the layer upstream (ffc.py:253-254).  Replaces autograd's
 * separate AddBackward / ReluBackward passes of refinement.py:163; at least one of g, gm. */
int lama_reflect_pad_bwd_fused(void* stream, const lama_tensor* gp, const lama_tensor* add1, const lama_tensor* add2, int32_t pad,
                               const lama_tensor* mask_y, int32_t act, const lama_tensor* g, const lama_tensor* gm, int32_t batch,
                               const float* ring);
/* (v108) the data gradient of a reflect-padded 3x3 conv (ffc.py:188-196) WITHOUT its padded plane: the interior H x W of the zero-padded
 * correlation is a zero-pad-1 conv of the output gradient (lama_conv2d_fwd / lama_winograd_conv3x3_fwd with LAMA_PAD_ZERO, pad 1) and the
 * one-pixel frame around it -- rows y = -1, H (x = -1..W), columns x = -1, W (y = 0..H-1) -- is computed here, exact fp32:
 *   ring [batch][cout][2 (W + 2) + 2 H] (top, bottom, left, right; lama_dgrad_ring_bytes),
 *   w_ring [4][cin][3][cout] fp32 = the dgrad weights w'[cout][cin][3][3] (flipped, transposed) as  top: w'[o][c][2][k], bottom: w'[o][c][0][k],
 *   left: w'[o][c][k][2], right: w'[o][c][k][0];  cout % 128 == 0, cin % 64 == 0.
 * lama_reflect_pad_bwd_fused with ring != NULL then takes gp = the interior [B,C,H,W] (pad must be 1). */
size_t lama_dgrad_ring_bytes(int32_t batch, int32_t cout, int32_t H, int32_t W);
int lama_dgrad_ring_fwd(void* stream, const lama_tensor* g, const float* w_ring, int32_t cout, float* ring, int32_t batch);
/* kornia.filters.gaussian_blur2d(x, (5,5), (1.0,1.0)) [border reflect] of the top-left crop [0:y.H, 0:y.W] of x
 * (refinement.py:24,52,149), and its adjoint (gx is zero outside the crop). */
int lama_gauss5_fwd(void* stream, const lama_tensor* x, const lama_tensor* y, int32_t batch);
int lama_gauss5_bwd(void* stream, const lama_tensor* gy, const lama_tensor* gx, int32_t batch);
/* F.interpolate(x, size=(y.H, y.W), mode='bilinear', align_corners=False) (refinement.py:25,53,55,200-201) and its adjoint. */
int lama_bilinear_fwd(void* stream, const lama_tensor* x, const lama_tensor* y, int32_t batch);
int lama_bilinear_bwd(void* stream, const lama_tensor* gy, const lama_tensor* gx, int32_t batch);
/* y = x >= thr ? 1 : 0  (mask[mask >= eps] = 1; mask[mask < eps] = 0: refinement.py:57-62,70-71,300-301) */
int lama_threshold_fwd(void* stream, const lama_tensor* x, float thr, const lama_tensor* y, int32_t batch);
/* kornia.morphology.erosion(x, se) with a flat structuring element se [kh,kw] (device floats, non-zero = member; origin =
 * centre; geodesic border: positions outside the image never lower the minimum), refinement.py:68 with the 15x15 ellipse. */
int lama_erode_fwd(void* stream, const lama_tensor* x, const float* se, int32_t kh, int32_t kw, float max_val, const lama_tensor* y,
                   int32_t batch);
/* Masked L1 terms of refinement.py:75-84 over the elements with mask < thr (select_ge = 0) or mask >= thr (select_ge = 1); a
 * 1-channel mask is broadcast over pred's channels (mask.repeat(1,3,1,1)).
 *   fwd: accum[0] += sum |pred - target|, accum[1] += number of selected elements   (two device doubles, zeroed by the caller)
 *   bwd: g = (accumulate ? g : 0) + scale * sign(pred - target) on the selected elements  (d mean|.| / d pred: scale = 1 / count) */
int lama_l1_masked_fwd(void* stream, const lama_tensor* pred, const lama_tensor* target, const lama_tensor* mask, float thr,
                       int32_t select_ge, double* accum, int32_t batch);
int lama_l1_masked_bwd(void* stream, const lama_tensor* pred, const lama_tensor* target, const lama_tensor* mask, float thr,
                       int32_t select_ge, float scale, int32_t accumulate, const lama_tensor* g, int32_t batch);
/* one torch.optim.Adam step (refinement.py:134,165: betas, eps as given, no weight decay / amsgrad) on a flat fp32 buffer;
 * step = 1, 2, ... */
int lama_adam_step(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                   float beta2, float eps, int32_t step);

/* ---- quality metrics --------------------------------------------------------------------------
 * SSIM.forward(img1, img2) with size_average=False (saicinpainting/evaluation/losses/ssim.py:18-71): out[i] = mean over (C, H, W)
 * of the SSIM map of image i; depthwise window_size x window_size gaussian window given as its normalised 1-D factor `window1d`
 * (HOST array of window_size floats, ssim.py:36-45), zero padding window_size / 2.  Odd window sizes up to 15.  workspace: device
 * scratch of lama_ssim_workspace_bytes() bytes.  Replaces the five F.conv2d(groups=channel) calls + the elementwise map. */
size_t lama_ssim_workspace_bytes(int32_t batch, int32_t C, int32_t H, int32_t W);
int lama_ssim_fwd(void* stream, const lama_tensor* img1, const lama_tensor* img2, int32_t batch, int32_t window_size,
                  const float* window1d, float* out, void* workspace, size_t workspace_bytes);

#ifdef __cplusplus
}
#endif
#endif /* LAMA_HIP_H */
