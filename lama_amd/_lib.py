"""ctypes binding of the C ABI declared in include/lama_hip.h.

``LamaLib()`` loads the in-tree ``lama_amd/lib/liblama_hip.so`` (built by ``lama_amd.build`` /
``__graft_entry__.build()``) and raises if it is missing: there is no CPU or eager-PyTorch fallback.
All entry points take raw device pointers (``tensor.data_ptr()``) and a ``hipStream_t``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3
PAD_ZERO, PAD_REFLECT = 0, 1
PREC_F32, PREC_BF16X3, PREC_F16X3, PREC_F16 = 0, 1, 2, 3
CONV_COOPERATIVE = 1
CONV_DEFER_OUT = 2          # lama_winograd_conv3x3_fwd: the GEMM launch only (include/lama_hip.h)
CONV_SIBLINGS_SHIFT = 8     # (v109) bits 8..10: log2 of the number of sibling launches that share the chip with this one
DT_F32, DT_F16 = 0, 1
ABI_VERSION = 110      # LAMA_HIP_VERSION of include/lama_hip.h
PREC_NAMES = {'f32': PREC_F32, 'bf16x3': PREC_BF16X3, 'f16x3': PREC_F16X3, 'f16': PREC_F16}

_DEFAULT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'liblama_hip.so')


class LamaError(RuntimeError):
    """``code``: the numeric return value of the C entry point (LAMA_ERR_* < 0, hipError_t > 0) or None for host-side errors."""

    def __init__(self, msg: str = '', code=None):
        super().__init__(msg)
        self.code = code


class LamaRangeError(LamaError):
    """A weight or an activation left the range of the fp16 split (|x| <= 65504): re-run with PREC_BF16X3 / PREC_F32."""

F16_MAX = 65504.0
ERR_BAD_ARG, ERR_UNSUPPORTED, ERR_WORKSPACE = -1, -2, -3      # include/lama_hip.h LAMA_ERR_*


class Tensor4(C.Structure):
    """struct lama_tensor"""
    _fields_ = [('ptr', C.c_void_p), ('batch_stride', C.c_int64), ('C', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('dtype', C.c_int32)]


class Conv2dArgs(C.Structure):
    """struct lama_conv2d_args"""
    _fields_ = [('x', Tensor4), ('w_packed', C.c_void_p),
                ('kh', C.c_int32), ('kw', C.c_int32), ('stride', C.c_int32), ('pad', C.c_int32),
                ('pad_mode', C.c_int32), ('transposed', C.c_int32),
                ('x2', Tensor4), ('w2_packed', C.c_void_p), ('bias', C.c_void_p), ('act', C.c_int32),
                ('resid', Tensor4), ('y', Tensor4), ('batch', C.c_int32), ('precision', C.c_int32),
                ('range_flag', C.c_void_p), ('fuse1_w', C.c_void_p), ('fuse1_bias', C.c_void_p), ('fuse1_y', Tensor4),
                ('flags', C.c_int32)]


def view(t: Optional[torch.Tensor], c0: int = 0, c: Optional[int] = None) -> Tensor4:
    """lama_tensor view of channels [c0, c0+c) of a contiguous fp32 / fp16 NCHW tensor (None -> absent)."""
    if t is None:
        return Tensor4(None, 0, 0, 0, 0, 0)
    if t.dtype not in (torch.float32, torch.float16) or t.dim() != 4 or not t.is_contiguous():
        raise LamaError(f'expected a contiguous fp32 / fp16 NCHW tensor, got {t.dtype} {tuple(t.shape)}')
    B, Ct, H, W = t.shape
    c = Ct - c0 if c is None else c
    if c0 < 0 or c <= 0 or c0 + c > Ct:
        raise LamaError('channel slice out of range')
    return Tensor4(t.data_ptr() + t.element_size() * c0 * H * W, Ct * H * W, c, H, W, DT_F16 if t.dtype == torch.float16 else DT_F32)


class LamaLib:
    def __init__(self, path: Optional[str] = None, host_emulated: bool = False):
        """``host_emulated``: the library is the x86 build of the kernel sources (tests/hipemu) and takes HOST pointers; the
        real library takes device pointers only and every wrapper below refuses CPU tensors."""
        self.host_emulated = host_emulated
        path = path or _DEFAULT                   # (no environment variable chooses the library: tools pass a path -- use_library())
        if not os.path.exists(path):
            raise LamaError(f'{path} not found: build it with `python -m lama_amd.build` '
                            f'(hipcc --offload-arch=gfx950); lama_amd has no fallback path')
        self.path = path
        L = self._l = C.CDLL(path)
        i32, i64, vp, sz = C.c_int32, C.c_int64, C.c_void_p, C.c_size_t
        T = C.POINTER(Tensor4)
        L.lama_version.restype = C.c_int
        L.lama_error_string.restype = C.c_char_p
        L.lama_error_string.argtypes = [C.c_int]
        L.lama_conv2d_packed_weight_bytes.restype = i64
        L.lama_conv2d_packed_weight_bytes.argtypes = [i32] * 7
        L.lama_conv2d_pack_weight.restype = C.c_int
        L.lama_conv2d_pack_weight.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
        L.lama_conv2d_fwd.restype = C.c_int
        L.lama_conv2d_fwd.argtypes = [vp, C.POINTER(Conv2dArgs)]
        L.lama_rfft2_fwd.restype = C.c_int
        L.lama_rfft2_fwd.argtypes = [vp, T, T, i32, vp, sz]
        L.lama_irfft2_fwd.restype = C.c_int
        L.lama_irfft2_fwd.argtypes = [vp, T, T, T, i32, vp, sz]
        L.lama_winograd_out_fwd.restype = C.c_int
        L.lama_winograd_out_fwd.argtypes = [vp, C.POINTER(Conv2dArgs), vp, sz]
        L.lama_rfft2_winograd_out_fwd.restype = C.c_int
        L.lama_rfft2_winograd_out_fwd.argtypes = [vp, T, T, i32, vp, sz, C.POINTER(Conv2dArgs), vp, sz]
        L.lama_rfft2_masked_fwd.restype = C.c_int
        L.lama_rfft2_masked_fwd.argtypes = [vp, T, T, T, i32, vp, sz]
        L.lama_irfft2_masked_fwd.restype = C.c_int
        L.lama_irfft2_masked_fwd.argtypes = [vp, T, T, T, T, i32, vp, sz]
        L.lama_fft_workspace_bytes.restype = sz
        L.lama_fft_workspace_bytes.argtypes = [i32] * 4
        L.lama_fourier_unit_workspace_bytes.restype = sz
        L.lama_fourier_unit_workspace_bytes.argtypes = [i32] * 4
        L.lama_fourier_unit_fwd.restype = C.c_int
        L.lama_fourier_unit_fwd.argtypes = [vp, T, vp, vp, T, i32, i32, i32, vp, sz, vp]
        L.lama_fourier_unit_winograd_out_fwd.restype = C.c_int
        L.lama_fourier_unit_winograd_out_fwd.argtypes = [vp, T, vp, vp, T, i32, i32, i32, vp, sz, vp, C.POINTER(Conv2dArgs), vp, sz]
        L.lama_fourier_unit_ex_fwd.restype = C.c_int
        L.lama_fourier_unit_ex_fwd.argtypes = [vp, T, vp, vp, T, i32, i32, i32, vp, sz, vp, i32, C.POINTER(Conv2dArgs), vp, sz]
        L.lama_mask_compose_fwd.restype = C.c_int
        L.lama_mask_compose_fwd.argtypes = [vp, T, T, T, i32]
        L.lama_blend_fwd.restype = C.c_int
        L.lama_blend_fwd.argtypes = [vp, T, T, T, T, i32]
        L.lama_quantize_u8_hwc_fwd.restype = C.c_int
        L.lama_quantize_u8_hwc_fwd.argtypes = [vp, T, vp, i32, i32, i32]
        L.lama_mask_compose_u8_fwd.restype = C.c_int
        L.lama_mask_compose_u8_fwd.argtypes = [vp, vp, vp, vp, T, i32, i32]
        L.lama_blend_quantize_u8_fwd.restype = C.c_int
        L.lama_blend_quantize_u8_fwd.argtypes = [vp, vp, vp, vp, T, vp, i32, i32]
        L.lama_affine_act_fwd.restype = C.c_int
        L.lama_affine_act_fwd.argtypes = [vp, T, vp, vp, i32, T, i32]
        L.lama_reflect_pad_fwd.restype = C.c_int
        L.lama_reflect_pad_fwd.argtypes = [vp, T, i32, T, i32]
        f32, dp = C.c_float, C.POINTER(C.c_double)
        for name, args in (('lama_act_bwd', [vp, T, T, i32, T, i32]), ('lama_add_fwd', [vp, T, T, T, i32]),
                           ('lama_reflect_pad_bwd', [vp, T, T, i32, T, i32]), ('lama_reflect_pad_bwd_fused', [vp, T, T, T, i32, T, i32, T, T, i32, vp]),
                           ('lama_dgrad_ring_fwd', [vp, T, vp, i32, vp, i32]),
                           ('lama_gauss5_fwd', [vp, T, T, i32]),
                           ('lama_gauss5_bwd', [vp, T, T, i32]), ('lama_bilinear_fwd', [vp, T, T, i32]),
                           ('lama_bilinear_bwd', [vp, T, T, i32]), ('lama_threshold_fwd', [vp, T, f32, T, i32]),
                           ('lama_erode_fwd', [vp, T, vp, i32, i32, f32, T, i32]),
                           ('lama_l1_masked_fwd', [vp, T, T, T, f32, i32, vp, i32]),
                           ('lama_l1_masked_bwd', [vp, T, T, T, f32, i32, f32, i32, T, i32]),
                           ('lama_adam_step', [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, i32])):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = C.c_int, args
        L.lama_dgrad_ring_bytes.restype, L.lama_dgrad_ring_bytes.argtypes = C.c_size_t, [i32, i32, i32, i32]
        L.lama_ssim_workspace_bytes.restype, L.lama_ssim_workspace_bytes.argtypes = C.c_size_t, [i32, i32, i32, i32]
        L.lama_ssim_fwd.restype, L.lama_ssim_fwd.argtypes = C.c_int, [vp, T, T, i32, i32, C.POINTER(C.c_float), vp, vp, C.c_size_t]
        L.lama_fuse1_channel_order.restype, L.lama_fuse1_channel_order.argtypes = None, [C.POINTER(C.c_int32)]
        L.lama_winograd_packed_weight_bytes.restype, L.lama_winograd_packed_weight_bytes.argtypes = C.c_int64, [i32, i32, i32]
        L.lama_winograd_pack_weight.restype, L.lama_winograd_pack_weight.argtypes = C.c_int, [vp, vp, vp, i32, i32, i32, vp]
        L.lama_winograd_workspace_bytes.restype, L.lama_winograd_workspace_bytes.argtypes = C.c_size_t, [i32, i32, i32, i32]
        L.lama_winograd_supported.restype, L.lama_winograd_supported.argtypes = i32, [i32, i32, i32, i32, i32]
        L.lama_winograd_preferred.restype, L.lama_winograd_preferred.argtypes = i32, [i32, i32, i32, i32, i32, i32]
        L.lama_winograd_conv3x3_fwd.restype, L.lama_winograd_conv3x3_fwd.argtypes = C.c_int, [vp, C.POINTER(Conv2dArgs), vp, C.c_size_t]
        del dp
        if L.lama_version() != ABI_VERSION:
            raise LamaError(f'{path}: ABI version {L.lama_version()} != {ABI_VERSION}')

    # -- helpers -------------------------------------------------------------------------------
    def check(self, rc: int, what: str):
        if rc != 0:
            raise LamaError(f'{what} failed: {self._l.lama_error_string(rc).decode()} (code {rc})', code=int(rc))

    @staticmethod
    def stream_of(t: torch.Tensor) -> int:
        return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0

    # -- conv ------------------------------------------------------------------------------------
    def pack_conv_weight(self, w: torch.Tensor, scale: Optional[torch.Tensor], stride: int = 1, transposed: bool = False,
                         precision: int = PREC_F32) -> torch.Tensor:
        """Reference-layout conv weight (+ folded BN scale) -> packed device buffer."""
        if not self.host_emulated and not w.is_cuda:
            raise LamaError('pack_conv_weight: the weights are on the CPU (e.g. after load_checkpoint(map_location="cpu")): move the '
                            'module to the GPU first (model.to("cuda")) -- the HIP kernels dereference device pointers only')
        w = w.contiguous().float()
        if scale is not None and scale.device != w.device:
            raise LamaError(f'pack_conv_weight: weight on {w.device} but BatchNorm scale on {scale.device}')
        if transposed:
            cin, cout, kh, kw = w.shape
        else:
            cout, cin, kh, kw = w.shape
        nbytes = self._l.lama_conv2d_packed_weight_bytes(cout, cin, kh, kw, stride, int(transposed), precision)
        if nbytes <= 0:
            raise LamaError(f'unsupported conv geometry {tuple(w.shape)} stride={stride} transposed={transposed}')
        if precision in (PREC_F16X3, PREC_F16) and w.numel():
            # the fp16 split of the WEIGHTS is checked here, once (activations: range_flag of lama_conv2d_args)
            ws = w if scale is None else w * (scale.float().view(1, -1, 1, 1) if transposed else scale.float().view(-1, 1, 1, 1))
            amax = float(ws.abs().max())
            if not amax <= F16_MAX:
                raise LamaRangeError(f'folded conv weight |w| max {amax:.3g} exceeds the fp16 split range ({F16_MAX}); use bf16x3 or f32')
        dst = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
        sc = None if scale is None else scale.contiguous().float()
        self.check(self._l.lama_conv2d_pack_weight(self.stream_of(w), w.data_ptr(), None if sc is None else sc.data_ptr(),
                                                   cout, cin, kh, kw, stride, int(transposed), precision, dst.data_ptr()),
                   'lama_conv2d_pack_weight')
        if w.is_cuda:
            torch.cuda.current_stream(w.device).synchronize()  # w / sc may be temporaries
        return dst

    def conv2d(self, x: Tensor4, w_packed: torch.Tensor, y: Tensor4, batch: int, k: int, stride: int = 1, pad: int = 0,
               pad_mode: int = PAD_REFLECT, transposed: bool = False, bias: Optional[torch.Tensor] = None,
               act: int = ACT_NONE, resid: Optional[Tensor4] = None, x2: Optional[Tensor4] = None,
               w2_packed: Optional[torch.Tensor] = None, precision: int = PREC_F32, stream: int = 0,
               range_flag: Optional[torch.Tensor] = None, fuse1: Optional[tuple] = None, cooperative: bool = False, siblings_log2: int = 0):
        """``fuse1`` = (packed conv1 weights with the channel order of fuse1_channel_order(), BatchNorm shift [192], x1 view): the NEXT
        layer's SpectralTransform.conv1 in the epilogue of this (global-branch) launch -- lama_conv2d_args.fuse1_*.  ``siblings_log2`` (v109):
        this launch covers 1 / 2^n of a batch whose other parts run beside it (LAMA_CONV_SIBLINGS_*)."""
        a = Conv2dArgs()
        a.x, a.w_packed = x, w_packed.data_ptr()
        a.kh = a.kw = k
        a.stride, a.pad, a.pad_mode, a.transposed = stride, pad, pad_mode, int(transposed)
        if x2 is not None:
            a.x2, a.w2_packed = x2, w2_packed.data_ptr()
        a.bias = None if bias is None else bias.data_ptr()
        a.act = act
        if resid is not None:
            a.resid = resid
        a.y, a.batch, a.precision = y, batch, precision
        a.range_flag = None if range_flag is None else range_flag.data_ptr()
        if fuse1 is not None:
            a.fuse1_w, a.fuse1_bias, a.fuse1_y = fuse1[0].data_ptr(), fuse1[1].data_ptr(), fuse1[2]
        a.flags = (CONV_COOPERATIVE if cooperative else 0) | (int(siblings_log2) << CONV_SIBLINGS_SHIFT)   # lama_conv2d_args.flags
        self.check(self._l.lama_conv2d_fwd(stream, C.byref(a)), 'lama_conv2d_fwd')

    # -- Winograd F(2x2, 3x3) form of the stride-1 3x3 reflect conv (lama_winograd_*) -----------------
    def winograd_supported(self, cout: int, cin: int, H: int, W: int, precision: int) -> bool:
        return (precision in (PREC_F16X3, PREC_BF16X3) and self._l.lama_winograd_packed_weight_bytes(cout, cin, precision) > 0
                and self._l.lama_winograd_supported(cout, cin, H, W, precision) == 1)

    def winograd_preferred(self, batch: int, cout: int, cin: int, H: int, W: int, precision: int) -> bool:
        """(v110) supported AND expected to beat the direct kernel at this batch (the any-size geometry pays for whole rounds of workgroups)."""
        return self.winograd_supported(cout, cin, H, W, precision) and self._l.lama_winograd_preferred(batch, cout, cin, H, W, precision) == 1

    def winograd_workspace_bytes(self, batch: int, cout: int, H: int, W: int) -> int:
        return int(self._l.lama_winograd_workspace_bytes(batch, cout, H, W))

    def pack_winograd_weight(self, w: torch.Tensor, scale: Optional[torch.Tensor], precision: int) -> torch.Tensor:
        """Conv2d weight [Cout, Cin, 3, 3] (+ folded BN scale) -> packed U = G g G^T fragments (lama_winograd_pack_weight)."""
        if not self.host_emulated and not w.is_cuda:
            raise LamaError('pack_winograd_weight: the weights are on the CPU: move the module to the GPU first')
        w = w.contiguous().float()
        cout, cin, kh, kw = w.shape
        nbytes = self._l.lama_winograd_packed_weight_bytes(cout, cin, precision) if (kh, kw) == (3, 3) else -1
        if nbytes <= 0:
            raise LamaError(f'unsupported Winograd geometry {tuple(w.shape)} / precision {precision}', code=ERR_UNSUPPORTED)
        if precision == PREC_F16X3 and w.numel():
            ws = w if scale is None else w * scale.float().view(-1, 1, 1, 1)
            amax = float(ws.abs().sum(dim=(2, 3)).max())        # |U| <= sum |g| (the rows of G have absolute sum <= 1.5, squared 2.25 > needed)
            if not 2.25 * amax <= F16_MAX:
                raise LamaRangeError(f'transformed conv weight bound {2.25 * amax:.3g} exceeds the fp16 split range ({F16_MAX}); use bf16x3')
        dst = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
        sc = None if scale is None else scale.contiguous().float()
        self.check(self._l.lama_winograd_pack_weight(self.stream_of(w), w.data_ptr(), None if sc is None else sc.data_ptr(), cout, cin,
                                                     precision, dst.data_ptr()), 'lama_winograd_pack_weight')
        if w.is_cuda:
            torch.cuda.current_stream(w.device).synchronize()
        return dst

    def winograd_conv3x3(self, x: Tensor4, w_packed: torch.Tensor, y: Tensor4, batch: int, ws: torch.Tensor, bias: Optional[torch.Tensor] = None,
                         act: int = ACT_NONE, resid: Optional[Tensor4] = None, precision: int = PREC_F16X3, stream: int = 0,
                         range_flag: Optional[torch.Tensor] = None, pad_mode: int = PAD_REFLECT, defer_out: bool = False, siblings_log2: int = 0):
        """``defer_out`` (v108): only the GEMM launch; returns the argument block the caller hands to ``winograd_out`` / ``rfft2_wino_out``
        (with the same workspace) before anything reads y."""
        a = Conv2dArgs()
        a.x, a.w_packed = x, w_packed.data_ptr()
        a.kh = a.kw = 3
        a.stride, a.pad, a.pad_mode, a.transposed = 1, 1, pad_mode, 0
        a.bias = None if bias is None else bias.data_ptr()
        a.act = act
        if resid is not None:
            a.resid = resid
        a.y, a.batch, a.precision = y, batch, precision
        a.range_flag = None if range_flag is None else range_flag.data_ptr()
        a.flags = (CONV_DEFER_OUT if defer_out else 0) | (int(siblings_log2) << CONV_SIBLINGS_SHIFT)
        self.check(self._l.lama_winograd_conv3x3_fwd(stream, C.byref(a), ws.data_ptr(), ws.numel() * ws.element_size()), 'lama_winograd_conv3x3_fwd')
        return a if defer_out else None

    def winograd_out(self, args: 'Conv2dArgs', ws: torch.Tensor, stream: int = 0):
        """The output transform a ``winograd_conv3x3(..., defer_out=True)`` call left out (lama_winograd_out_fwd)."""
        self.check(self._l.lama_winograd_out_fwd(stream, C.byref(args), ws.data_ptr(), ws.numel() * ws.element_size()), 'lama_winograd_out_fwd')

    def rfft2_wino_out(self, x: Tensor4, spec: Tensor4, batch: int, ws: Optional[torch.Tensor], wino_args: 'Conv2dArgs', wino_ws: torch.Tensor,
                       stream: int = 0):
        """rfft2 + the deferred Winograd output transform of the layer before in one launch (lama_rfft2_winograd_out_fwd; ERR_UNSUPPORTED where
        no kernel does it: the caller then runs the two separately)."""
        self.check(self._l.lama_rfft2_winograd_out_fwd(stream, C.byref(x), C.byref(spec), batch, None if ws is None else ws.data_ptr(),
                                                       0 if ws is None else ws.numel() * ws.element_size(), C.byref(wino_args), wino_ws.data_ptr(),
                                                       wino_ws.numel() * wino_ws.element_size()), 'lama_rfft2_winograd_out_fwd')

    def fuse1_channel_order(self) -> torch.Tensor:
        """Input-channel order of a conv1 weight that rides in another launch's epilogue (lama_fuse1_channel_order)."""
        buf = (C.c_int32 * 384)()
        self._l.lama_fuse1_channel_order(buf)
        return torch.tensor(list(buf), dtype=torch.long)

    # -- fft -------------------------------------------------------------------------------------
    def fft_workspace_bytes(self, b, c, h, w) -> int:
        return int(self._l.lama_fft_workspace_bytes(b, c, h, w))

    def rfft2(self, x: Tensor4, spec: Tensor4, batch: int, ws: Optional[torch.Tensor] = None, stream: int = 0, mask: Optional[Tensor4] = None):
        """``mask``: spec *= [mask > 0] in the same launch (lama_rfft2_masked_fwd; LamaError ERR_UNSUPPORTED where no kernel does it)."""
        if mask is not None:
            self.check(self._l.lama_rfft2_masked_fwd(stream, C.byref(x), C.byref(spec), C.byref(mask), batch,
                                                     None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * ws.element_size()),
                       'lama_rfft2_masked_fwd')
            return
        self.check(self._l.lama_rfft2_fwd(stream, C.byref(x), C.byref(spec), batch,
                                          None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * ws.element_size()),
                   'lama_rfft2_fwd')

    def irfft2(self, spec: Tensor4, resid: Optional[Tensor4], y: Tensor4, batch: int, ws: Optional[torch.Tensor] = None,
               stream: int = 0, mask: Optional[Tensor4] = None):
        if mask is not None:
            self.check(self._l.lama_irfft2_masked_fwd(stream, C.byref(spec), None if resid is None else C.byref(resid), C.byref(mask), C.byref(y), batch,
                                                      None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * ws.element_size()),
                       'lama_irfft2_masked_fwd')
            return
        self.check(self._l.lama_irfft2_fwd(stream, C.byref(spec), None if resid is None else C.byref(resid), C.byref(y), batch,
                                           None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * ws.element_size()),
                   'lama_irfft2_fwd')

    def fourier_unit_workspace_bytes(self, b, c, h, w) -> int:
        return int(self._l.lama_fourier_unit_workspace_bytes(b, c, h, w))

    def fourier_unit(self, x: Tensor4, w_packed: torch.Tensor, bias: torch.Tensor, y: Tensor4, batch: int, add_input: bool,
                     ws: torch.Tensor, precision: int = PREC_F32, stream: int = 0, range_flag: Optional[torch.Tensor] = None,
                     wino_out: Optional[tuple] = None, siblings_log2: int = 0):
        """``wino_out`` = (argument block of a ``winograd_conv3x3(..., defer_out=True)`` call, its workspace): that conv's output transform runs
        inside this FourierUnit's rfft2 launch (lama_fourier_unit_winograd_out_fwd, v108).  ``siblings_log2`` (v109): lama_fourier_unit_ex_fwd."""
        if siblings_log2:
            wa, wws = wino_out if wino_out is not None else (None, None)
            self.check(self._l.lama_fourier_unit_ex_fwd(stream, C.byref(x), w_packed.data_ptr(), bias.data_ptr(), C.byref(y), batch,
                                                        int(add_input), precision, ws.data_ptr(), ws.numel() * ws.element_size(),
                                                        None if range_flag is None else range_flag.data_ptr(), int(siblings_log2) << CONV_SIBLINGS_SHIFT,
                                                        None if wa is None else C.byref(wa), None if wws is None else wws.data_ptr(),
                                                        0 if wws is None else wws.numel() * wws.element_size()),
                       'lama_fourier_unit_ex_fwd')
            return
        if wino_out is not None:
            wa, wws = wino_out
            self.check(self._l.lama_fourier_unit_winograd_out_fwd(stream, C.byref(x), w_packed.data_ptr(), bias.data_ptr(), C.byref(y), batch,
                                                                  int(add_input), precision, ws.data_ptr(), ws.numel() * ws.element_size(),
                                                                  None if range_flag is None else range_flag.data_ptr(), C.byref(wa),
                                                                  wws.data_ptr(), wws.numel() * wws.element_size()),
                       'lama_fourier_unit_winograd_out_fwd')
            return
        self.check(self._l.lama_fourier_unit_fwd(stream, C.byref(x), w_packed.data_ptr(), bias.data_ptr(), C.byref(y), batch,
                                                 int(add_input), precision, ws.data_ptr(), ws.numel() * ws.element_size(),
                                                 None if range_flag is None else range_flag.data_ptr()),
                   'lama_fourier_unit_fwd')

    # -- elementwise -------------------------------------------------------------------------------
    def mask_compose(self, image: Tensor4, mask: Tensor4, out: Tensor4, batch: int, stream: int = 0):
        self.check(self._l.lama_mask_compose_fwd(stream, C.byref(image), C.byref(mask), C.byref(out), batch), 'lama_mask_compose_fwd')

    def blend(self, image: Tensor4, mask: Tensor4, pred: Tensor4, out: Tensor4, batch: int, stream: int = 0):
        self.check(self._l.lama_blend_fwd(stream, C.byref(image), C.byref(mask), C.byref(pred), C.byref(out), batch), 'lama_blend_fwd')

    def mask_compose_u8(self, image_hwc: torch.Tensor, mask: torch.Tensor, sizes, out: Tensor4, batch: int, binarize: bool = True, stream: int = 0):
        """(v110) u8 HWC image [B,Hp,Wp,3] + u8 mask [B,Hp,Wp] (+ int32 sizes [B,2] or None) -> the generator's input [B,4,Hp,Wp] fp32."""
        self.check(self._l.lama_mask_compose_u8_fwd(stream, image_hwc.data_ptr(), mask.data_ptr(), None if sizes is None else sizes.data_ptr(),
                                                    C.byref(out), batch, int(bool(binarize))), 'lama_mask_compose_u8_fwd')

    def blend_quantize_u8(self, image_hwc: torch.Tensor, mask: torch.Tensor, sizes, pred: Tensor4, dst: torch.Tensor, batch: int,
                          binarize: bool = True, stream: int = 0):
        """(v110) blend (default.py:71) + u8 HWC quantisation (bin/predict.py:86-92) of the generator's output against the u8 operands."""
        self.check(self._l.lama_blend_quantize_u8_fwd(stream, image_hwc.data_ptr(), mask.data_ptr(), None if sizes is None else sizes.data_ptr(),
                                                      C.byref(pred), dst.data_ptr(), batch, int(bool(binarize))), 'lama_blend_quantize_u8_fwd')

    def quantize_u8_hwc(self, src: Tensor4, dst: torch.Tensor, batch: int, crop_h: int, crop_w: int, stream: int = 0):
        self.check(self._l.lama_quantize_u8_hwc_fwd(stream, C.byref(src), dst.data_ptr(), batch, crop_h, crop_w), 'lama_quantize_u8_hwc_fwd')


    def affine_act(self, x: Tensor4, scale, shift, act: int, y: Tensor4, batch: int, stream: int = 0):
        self.check(self._l.lama_affine_act_fwd(stream, C.byref(x), None if scale is None else scale.data_ptr(),
                                               None if shift is None else shift.data_ptr(), act, C.byref(y), batch), 'lama_affine_act_fwd')

    def reflect_pad(self, x: Tensor4, pad: int, y: Tensor4, batch: int, stream: int = 0):
        self.check(self._l.lama_reflect_pad_fwd(stream, C.byref(x), pad, C.byref(y), batch), 'lama_reflect_pad_fwd')


    # -- refinement (include/lama_hip.h, last section) --------------------------------------------
    def act_bwd(self, g: Tensor4, y: Tensor4, act: int, gout: Tensor4, batch: int, stream: int = 0):
        self.check(self._l.lama_act_bwd(stream, C.byref(g), C.byref(y), act, C.byref(gout), batch), 'lama_act_bwd')

    def add(self, a: Tensor4, b: Tensor4, out: Tensor4, batch: int, stream: int = 0):
        self.check(self._l.lama_add_fwd(stream, C.byref(a), C.byref(b), C.byref(out), batch), 'lama_add_fwd')

    def reflect_pad_bwd_fused(self, gp: Tensor4, add1: Optional[Tensor4], add2: Optional[Tensor4], pad: int, mask_y: Optional[Tensor4], act: int,
                              g: Optional[Tensor4], gm: Optional[Tensor4], batch: int, stream: int = 0, ring: Optional[torch.Tensor] = None):
        """s = fold(gp) [+ add1] [+ add2]; g = s; gm = s * act'(mask_y) -- one pass (include/lama_hip.h, v108).  With ``ring`` (dgrad_ring), gp
        is the interior [B,C,H,W] of the padded plane."""
        r = lambda t: None if t is None else C.byref(t)    # noqa: E731
        self.check(self._l.lama_reflect_pad_bwd_fused(stream, C.byref(gp), r(add1), r(add2), pad, r(mask_y), act, r(g), r(gm), batch,
                                                      None if ring is None else ring.data_ptr()), 'lama_reflect_pad_bwd_fused')

    def dgrad_ring_bytes(self, b: int, cout: int, h: int, w: int) -> int:
        return int(self._l.lama_dgrad_ring_bytes(b, cout, h, w))

    @staticmethod
    def dgrad_ring_weight(w_dgrad: torch.Tensor) -> torch.Tensor:
        """w' [cout, cin, 3, 3] (the flipped, transposed weights of a 3x3 dgrad conv) -> [4][cin][3][cout] fp32 (include/lama_hip.h)."""
        wt = w_dgrad.detach().float().permute(1, 2, 3, 0)            # [cin][ky][kx][cout]
        return torch.stack([wt[:, 2], wt[:, 0], wt[:, :, 2], wt[:, :, 0]], 0).contiguous()

    def dgrad_ring(self, g: Tensor4, w_ring: torch.Tensor, cout: int, ring: torch.Tensor, batch: int, stream: int = 0):
        if ring.numel() * 4 < self.dgrad_ring_bytes(batch, cout, g.H, g.W) or tuple(w_ring.shape) != (4, g.C, 3, cout):
            raise LamaError('dgrad_ring: ring / weight buffer does not match the launch', ERR_BAD_ARG)
        self.check(self._l.lama_dgrad_ring_fwd(stream, C.byref(g), w_ring.data_ptr(), cout, ring.data_ptr(), batch), 'lama_dgrad_ring_fwd')

    def reflect_pad_bwd(self, gp: Tensor4, addend: Optional[Tensor4], pad: int, g: Tensor4, batch: int, stream: int = 0):
        self.check(self._l.lama_reflect_pad_bwd(stream, C.byref(gp), None if addend is None else C.byref(addend), pad, C.byref(g), batch),
                   'lama_reflect_pad_bwd')

    def ssim(self, img1: Tensor4, img2: Tensor4, batch: int, window1d, out: torch.Tensor, workspace: torch.Tensor, stream: int = 0):
        """Per-image SSIM (ssim.py:46-71, size_average=False) into out [batch] fp32; window1d = host sequence of the odd-length
        normalised 1-D gaussian; workspace = device buffer of at least ssim_workspace_bytes() bytes."""
        if not self.host_emulated and not (out.is_cuda and workspace.is_cuda):
            raise LamaError('ssim: out / workspace must be device tensors')
        w = (C.c_float * len(window1d))(*[float(v) for v in window1d])
        self.check(self._l.lama_ssim_fwd(stream, C.byref(img1), C.byref(img2), batch, len(window1d), w, out.data_ptr(), workspace.data_ptr(),
                                         workspace.numel() * workspace.element_size()), 'lama_ssim_fwd')

    def ssim_workspace_bytes(self, batch: int, c: int, h: int, w: int) -> int:
        return int(self._l.lama_ssim_workspace_bytes(batch, c, h, w))

    def gauss5(self, x: Tensor4, y: Tensor4, batch: int, stream: int = 0):
        self.check(self._l.lama_gauss5_fwd(stream, C.byref(x), C.byref(y), batch), 'lama_gauss5_fwd')

    def gauss5_bwd(self, gy: Tensor4, gx: Tensor4, batch: int, stream: int = 0):
        self.check(self._l.lama_gauss5_bwd(stream, C.byref(gy), C.byref(gx), batch), 'lama_gauss5_bwd')

    def bilinear(self, x: Tensor4, y: Tensor4, batch: int, stream: int = 0):
        self.check(self._l.lama_bilinear_fwd(stream, C.byref(x), C.byref(y), batch), 'lama_bilinear_fwd')

    def bilinear_bwd(self, gy: Tensor4, gx: Tensor4, batch: int, stream: int = 0):
        self.check(self._l.lama_bilinear_bwd(stream, C.byref(gy), C.byref(gx), batch), 'lama_bilinear_bwd')

    def threshold(self, x: Tensor4, thr: float, y: Tensor4, batch: int, stream: int = 0):
        self.check(self._l.lama_threshold_fwd(stream, C.byref(x), thr, C.byref(y), batch), 'lama_threshold_fwd')

    def erode(self, x: Tensor4, se: torch.Tensor, y: Tensor4, batch: int, max_val: float = 1e4, stream: int = 0):
        kh, kw = se.shape
        self.check(self._l.lama_erode_fwd(stream, C.byref(x), se.data_ptr(), kh, kw, max_val, C.byref(y), batch), 'lama_erode_fwd')

    def l1_masked(self, pred: Tensor4, target: Tensor4, mask: Tensor4, thr: float, select_ge: bool, accum: torch.Tensor, batch: int,
                  stream: int = 0):
        assert accum.dtype == torch.float64 and accum.numel() >= 2
        self.check(self._l.lama_l1_masked_fwd(stream, C.byref(pred), C.byref(target), C.byref(mask), thr, int(select_ge), accum.data_ptr(),
                                              batch), 'lama_l1_masked_fwd')

    def l1_masked_bwd(self, pred: Tensor4, target: Tensor4, mask: Tensor4, thr: float, select_ge: bool, scale: float, accumulate: bool,
                      g: Tensor4, batch: int, stream: int = 0):
        self.check(self._l.lama_l1_masked_bwd(stream, C.byref(pred), C.byref(target), C.byref(mask), thr, int(select_ge), scale,
                                              int(accumulate), C.byref(g), batch), 'lama_l1_masked_bwd')

    def adam_step(self, param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, lr: float, step: int,
                  beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, stream: int = 0):
        n = param.numel()
        for t_ in (param, grad, exp_avg, exp_avg_sq):
            if t_.dtype != torch.float32 or not t_.is_contiguous() or t_.numel() != n:
                raise LamaError('adam_step: contiguous fp32 tensors of one size expected')
        self.check(self._l.lama_adam_step(stream, param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), n, lr,
                                          beta1, beta2, eps, step), 'lama_adam_step')


_LIB: Optional[LamaLib] = None


def get_lib() -> LamaLib:
    """The process-wide HIP library (loaded on first use; raises LamaError if it was not built)."""
    global _LIB
    if _LIB is None:
        _LIB = LamaLib()
    return _LIB


def use_library(path: str) -> LamaLib:
    """Measurement tools only (tools/, bench.py --lib): make another build of the same ABI -- the profiling build, an A/B base -- the
    process-wide library.  Must be called before anything loaded the default one; the product never calls it and no environment variable
    selects a library."""
    global _LIB
    if _LIB is not None and os.path.abspath(_LIB.path) != os.path.abspath(path):
        raise LamaError(f'use_library({path}): {_LIB.path} is already loaded')
    if _LIB is None:
        _LIB = LamaLib(path)
    return _LIB
