"""Host-side mirror of ``saicinpainting/training/modules/ffc.py`` whose forward passes run on the
hand-written gfx950 kernels of liblama_hip.so.

Same class names, constructor signatures, sub-module names and therefore the same ``state_dict`` keys
as the reference (``generator.model.<i>....``), so a reference checkpoint loads
with ``load_state_dict`` unchanged and ``make_generator(kind='ffc_resnet')`` can return this class.
``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.ConvTranspose2d`` objects are used as PARAMETER CONTAINERS
only -- their ``forward`` is never called; all arithmetic goes through ``lama_amd._lib`` (C ABI in
``include/lama_hip.h``).  There is no eager/CPU fallback: tensors must live on a GPU and the shared
library must be built, otherwise ``LamaError`` is raised.

Inference only (eval-mode BatchNorm folded into the conv weights; reference ffc.py:60-61,101,
130-133,236-239,253-254).  Options no shipped config enables (LFU, SE, gates, spectral positional
encoding, spatial scaling, 3-D FFC, groups != 1, spatial-transform wrappers, out_ffc) raise
``NotImplementedError`` at construction.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import collections
import contextlib
import warnings

import torch
import torch.nn as nn

from . import _lib as L
from ._lib import LamaError, LamaRangeError

_ACT = {'relu': L.ACT_RELU, 'sigmoid': L.ACT_SIGMOID, 'tanh': L.ACT_TANH, None: L.ACT_NONE}


# ----------------------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------------------

def _bn_fold(bn: nn.BatchNorm2d) -> Tuple[torch.Tensor, torch.Tensor]:
    """eval-mode BatchNorm as y = x*scale + shift (SURVEY.md Appendix A folding recipe)."""
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    return scale, shift


def _act_code(layer) -> int:
    if layer is None or isinstance(layer, nn.Identity) or layer is nn.Identity:
        return L.ACT_NONE
    if isinstance(layer, nn.ReLU) or layer is nn.ReLU:
        return L.ACT_RELU
    if isinstance(layer, nn.Sigmoid) or layer is nn.Sigmoid:
        return L.ACT_SIGMOID
    if isinstance(layer, nn.Tanh) or layer is nn.Tanh:
        return L.ACT_TANH
    raise NotImplementedError(f'activation {layer} is not supported by the HIP epilogues')


def _adjacent(a: torch.Tensor, b: torch.Tensor) -> Optional[torch.Tensor]:
    """If a and b are channel slices [0:ca] and [ca:ca+cb] of ONE contiguous NCHW buffer, return that
    buffer as a [B, ca+cb, H, W] tensor (zero-copy), else None."""
    if a.dim() != 4 or b.dim() != 4 or a.shape[0] != b.shape[0] or a.shape[2:] != b.shape[2:]:
        return None
    B, ca, H, W = a.shape
    cb = b.shape[1]
    bs = (ca + cb) * H * W
    want = (bs, H * W, W, 1)
    if tuple(a.stride()) != want or tuple(b.stride()) != want:
        return None
    if a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr():
        return None
    if b.storage_offset() != a.storage_offset() + ca * H * W:
        return None
    return torch.as_strided(a, (B, ca + cb, H, W), want, a.storage_offset())


def _pair_buffer(x_l: torch.Tensor, x_g) -> Tuple[torch.Tensor, int, int]:
    """(buffer [B, cl+cg, H, W], cl, cg) holding x_l | x_g channel-contiguously."""
    if not torch.is_tensor(x_g):
        return x_l.contiguous(), x_l.shape[1], 0
    buf = _adjacent(x_l, x_g)
    if buf is None:
        buf = torch.cat([x_l, x_g], dim=1)
    return buf, x_l.shape[1], x_g.shape[1]


def _check_input(x: torch.Tensor):
    if not x.is_cuda:
        raise LamaError('lama_amd runs on an MI355X only: input tensor is on ' + str(x.device) +
                        ' (there is no CPU fallback; the CPU oracle lives in oracle/ for tests)')
    if x.dtype not in (torch.float32, torch.float16):
        raise LamaError(f'fp32 (or, with precision f16, fp16) activations expected, got {x.dtype}')


class SidePipe:
    """The second stream of a run of FFC layers, with NO join on the critical path.  Layer i's local conv needs layer
    i-1's outputs only, and layer i's global launch needs layer i-1's x_l and this layer's spectral branch -- not this layer's local
    conv.  So the local convs form one chain on ``stream`` (each waits for the previous layer's global launch), the spectral branches
    and global launches form the other chain on the caller's stream (each global launch waits for the PREVIOUS layer's local conv, an
    event recorded a whole layer earlier), and the two meet again only at ``join``.  A fork + join around every local conv costs
    5 + 6.6 us of cross-queue signalling per layer inside a hipGraph (profiles/r02_timeline_overlap_step.txt)."""

    def __init__(self, stream: 'torch.cuda.Stream'):
        self.stream = stream
        self.prev_local = None        # event: the last local conv launched on ``stream``

    def join(self):
        if self.prev_local is not None:
            torch.cuda.current_stream(self.stream.device).wait_stream(self.stream)
            self.prev_local = None


class _Exec:
    """Where kernels run: the loaded library and the stream to launch on.  The default is the in-tree
    gfx950 library on the current torch stream; tests inject the host-emulated build of the same
    sources (tests/hipemu) to exercise this host logic without a GPU.

    It also owns the RANGE WATCH of the fp16 split (lama_conv2d_args.range_flag): every public ``forward`` opens a
    ``range_scope``; the conv / FourierUnit launches inside it hand the per-device flag to the kernels, and the outermost
    scope reads it back when it closes (one 4-byte D2H per forward) and raises ``LamaRangeError`` if an activation beyond
    65504 (or a NaN) was split -- FFCResNetGenerator.forward then falls back to PREC_BF16X3."""

    def __init__(self, lib: Optional[L.LamaLib] = None):
        self._lib = lib
        self.injected = lib is not None
        self._flags = {}                  # one flag per DEVICE of this exec -- shared by every generator that runs on it (the default exec: all of them)
        self._deferred_since_read = {}    # device -> deferred scopes have run since the flag was last read
        self._sticky = {}                 # device -> a deferred report a self-checking scope had to clear from the device flag (kept for check_range)
        self._depth = 0
        self.no_fuse1 = set()      # (input shape, precision) at which the fused conv1 epilogue was refused (FFC.launch)
        self._cur_flag = None
        self.fuse1_min_tiles = 192        # conv1 of the next layer rides in the global launch from this many 128-pixel tiles on (FFC.launch)
        # log2 of the number of parts of a batch that run side by side as parallel branches of one hipGraph (FFCResNetGenerator.split_batch):
        # set while the launches of ONE part are issued; every conv / FourierUnit / Winograd launch carries it (LAMA_CONV_SIBLINGS_*, v109)
        self.siblings_log2 = 0
        self.winograd = True              # the local 3x3 conv of a two-branch FFC layer as Winograd F(2x2, 3x3) where the shape allows (wino_dev.inc)
        self.local_first = True           # capture order at the fork: the first successor of a hipGraph node stays on its queue
        self.cooperative_serial = False   # tests: the one-stream launch order with the overlapped order's kernel geometry (bit-equal results)

    @property
    def lib(self) -> L.LamaLib:
        if self._lib is None:
            self._lib = L.get_lib()
        return self._lib

    def stream(self, t: torch.Tensor) -> int:
        return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0

    def check(self, x: torch.Tensor):
        if not self.injected:
            _check_input(x)

    @contextlib.contextmanager
    def range_scope(self, t: torch.Tensor, precision: int, deferred: bool = False):
        """``deferred``: the outermost scope does NOT read the flag back (no host synchronisation: forwards can queue behind one another);
        the flag stays raised on the device -- the kernels only ever OR into it -- until ``check_range`` reads it at the caller's next
        synchronisation point (FFCResNetGenerator.defer_range_check).  The flags belong to the exec (one per device), not to a generator: two
        generators on the default exec share them, so ``check_range`` of either reports -- and clears -- what both have raised."""
        outer = self._depth == 0
        if outer and precision in (L.PREC_F16X3, L.PREC_F16):
            key = self._dev_key(t.device)
            if key not in self._flags:
                self._flags[key] = torch.zeros(1, dtype=torch.int32, device=t.device)
            elif not deferred and self._deferred_since_read.get(key):
                # a flag left raised by DEFERRED forwards nobody asked about yet (check_range) is not this forward's: a self-checking forward starts
                # from a clean flag -- but the report is not lost: it is read here (this scope synchronises at its end anyway) and kept
                # as a host-side sticky bit that the next check_range / range_flag_raised ORs in
                if int(self._flags[key].item()) != 0:
                    self._sticky[key] = True
                self._flags[key].zero_()
            self._deferred_since_read[key] = bool(deferred)
            self._cur_flag = self._flags[key]
        self._depth += 1
        try:
            yield
        except BaseException:
            self._depth -= 1
            if outer:
                self._cur_flag = None
            raise
        self._depth -= 1
        if outer:
            flag, self._cur_flag = self._cur_flag, None
            if flag is not None and not deferred and int(flag.item()) != 0:
                flag.zero_()
                raise LamaRangeError(_RANGE_MSG)

    def range_flag_raised(self, device=None) -> bool:
        """Read back AND clear (one 4-byte D2H per device = a host synchronisation with that device's current stream) the flag(s) that deferred
        scopes left on the device: True when a forward since the last read split an activation beyond 65504 (or a NaN)."""
        bad = False
        want = None if device is None else self._dev_key(device)
        for key, flag in self._flags.items():
            if want is not None and key != want:
                continue
            self._deferred_since_read[key] = False
            if int(flag.item()) != 0:
                flag.zero_()
                bad = True
            if self._sticky.pop(key, False):
                bad = True
        return bad

    @contextlib.contextmanager
    def scratch_range_scope(self, t: torch.Tensor, precision: int):
        """Forwards whose VALUES do not matter (tune_split times graphs on noise): whatever they raise is discarded, and a report that deferred
        forwards of the caller left pending before is preserved (read first, restored afterwards).  Host-synchronising."""
        key = self._dev_key(t.device)
        flag = self._flags.get(key)
        pending = bool(self._deferred_since_read.get(key))
        was = flag is not None and pending and int(flag.item()) != 0
        try:
            with self.range_scope(t, precision, deferred=True):
                yield
        finally:
            flag = self._flags.get(key)
            if flag is not None:
                flag.fill_(1 if was else 0)
            self._deferred_since_read[key] = pending

    @staticmethod
    def _dev_key(device) -> str:
        """'cuda' and 'cuda:<current>' are the same device: the flags are keyed by the indexed name (tensors always carry the index)."""
        d = torch.device(device)
        if d.type == 'cuda' and d.index is None:
            d = torch.device('cuda', torch.cuda.current_device())
        return str(d)

    def conv2d(self, *a, **kw):
        flag = self._cur_flag if kw.get('precision') in (L.PREC_F16X3, L.PREC_F16) else None
        if self.siblings_log2:
            kw['siblings_log2'] = self.siblings_log2
        self.lib.conv2d(*a, range_flag=flag, **kw)

    def fourier_unit(self, *a, precision: int = L.PREC_F32, stream: int = 0, wino_out=None):
        flag = self._cur_flag if precision in (L.PREC_F16X3, L.PREC_F16) else None
        self.lib.fourier_unit(*a, precision=precision, stream=stream, range_flag=flag, wino_out=wino_out, siblings_log2=self.siblings_log2)

    def winograd_conv3x3(self, *a, precision: int = L.PREC_F16X3, **kw):
        flag = self._cur_flag if precision == L.PREC_F16X3 else None
        if self.siblings_log2:
            kw['siblings_log2'] = self.siblings_log2
        return self.lib.winograd_conv3x3(*a, precision=precision, range_flag=flag, **kw)


_RANGE_MSG = ('an activation left the range of the fp16 split (|x| > 65504 or NaN) in this forward; '
              're-run with precision bf16x3 (fp32 exponent range) or f32')
_DEFAULT_EXEC = _Exec()


def _act_dtype(precision: int) -> torch.dtype:
    """Element type of the activation tensors in HBM: fp16 for PREC_F16 (BASELINE configs[2]), fp32 otherwise."""
    return torch.float16 if precision == L.PREC_F16 else torch.float32


class _HipModule(nn.Module):
    """Base: packed-weight cache that is dropped whenever parameters may have changed."""

    precision = L.PREC_F16X3     # fp32-accurate 3-term fp16 split on the 16-bit MFMA; PREC_F32 = exact fmaf-chain MFMA (5x slower)

    def __init__(self):
        super().__init__()
        self._packed = None
        self._exec = _DEFAULT_EXEC

    def _invalidate(self):
        for m in self.modules():
            if isinstance(m, _HipModule):
                m._packed = None

    def _apply(self, fn, *a, **kw):
        self._invalidate()
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self._invalidate()
        return super().load_state_dict(*a, **kw)

    def set_exec(self, ex: _Exec):
        for m in self.modules():
            if isinstance(m, _HipModule):
                m._exec = ex
                m._packed = None
        return self

    def set_precision(self, precision: int):
        for m in self.modules():
            if isinstance(m, _HipModule):
                m.precision = precision
        self._precision_hook(precision)
        self._invalidate()          # packed weights and captured graphs belong to the old precision
        return self

    def _precision_hook(self, precision: int):
        """Per-layer exceptions to a uniform precision (FFCResNetGenerator: the fp32 tail of LAMA_PREC_F16)."""

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError('lama_amd modules are inference-only (eval-mode BatchNorm is folded into the kernels)')
        return super().train(False)


# ----------------------------------------------------------------------------------------------------
# FourierUnit / SpectralTransform
# ----------------------------------------------------------------------------------------------------

class FourierUnit(_HipModule):
    """ffc.py:49-113.  y = irfft2(relu(bn(conv1x1(interleave(rfft2(x))))))."""

    def __init__(self, in_channels, out_channels, groups=1, spatial_scale_factor=None, spatial_scale_mode='bilinear',
                 spectral_pos_encoding=False, use_se=False, se_kwargs=None, ffc3d=False, fft_norm='ortho'):
        super().__init__()
        if groups != 1 or spatial_scale_factor is not None or spectral_pos_encoding or use_se or ffc3d or fft_norm != 'ortho':
            raise NotImplementedError('FourierUnit: only the configuration used by the shipped LaMa configs is implemented '
                                      '(groups=1, no spatial scaling / positional encoding / SE / 3-D, ortho norm)')
        if in_channels != out_channels:
            raise NotImplementedError('FourierUnit: in_channels != out_channels')
        self.groups = groups
        self.conv_layer = nn.Conv2d(in_channels * 2, out_channels * 2, kernel_size=1, stride=1, padding=0, groups=1, bias=False)
        self.bn = nn.BatchNorm2d(out_channels * 2)
        self.relu = nn.ReLU(inplace=True)
        self.fft_norm = fft_norm

    def _pack(self):
        if self._packed is None:
            scale, shift = _bn_fold(self.bn)
            self._packed = (self._exec.lib.pack_conv_weight(self.conv_layer.weight.detach(), scale, precision=self.precision),
                            shift.contiguous())
        return self._packed

    def run(self, x: L.Tensor4, y: L.Tensor4, batch: int, add_input: bool, ws: torch.Tensor, stream: int, wino_out=None):
        wp, shift = self._pack()
        self._exec.fourier_unit(x, wp, shift, y, batch, add_input, ws, precision=self.precision, stream=stream, wino_out=wino_out)

    def workspace(self, x: torch.Tensor) -> torch.Tensor:
        b, c, h, w = x.shape
        n = self._exec.lib.fourier_unit_workspace_bytes(b, c, h, w)
        return torch.empty(n // 4 + 1, dtype=torch.float32, device=x.device)      # sized for fp32 spectra; fp16 ones use half of it

    def forward(self, x: torch.Tensor, add_input: bool = False) -> torch.Tensor:
        self._exec.check(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        with self._exec.range_scope(x, self.precision):
            self.run(L.view(x), L.view(y), x.shape[0], add_input, self.workspace(x), self._exec.stream(x))
        return y


class SpectralTransform(_HipModule):
    """ffc.py:116-163 with stride 1 and LFU disabled: conv2(x1 + fu(x1)), x1 = relu(bn(conv1(x)))."""

    def __init__(self, in_channels, out_channels, stride=1, groups=1, enable_lfu=True, **fu_kwargs):
        super().__init__()
        if enable_lfu:
            raise NotImplementedError('SpectralTransform: enable_lfu=True is not used by any shipped LaMa config')
        if stride != 1 or groups != 1:
            raise NotImplementedError('SpectralTransform: stride 2 / groups != 1 are not implemented')
        self.enable_lfu = enable_lfu
        self.stride = stride
        self.downsample = nn.Identity()
        self.conv1 = nn.Sequential(nn.Conv2d(in_channels, out_channels // 2, kernel_size=1, groups=groups, bias=False),
                                   nn.BatchNorm2d(out_channels // 2), nn.ReLU(inplace=True))
        self.fu = FourierUnit(out_channels // 2, out_channels // 2, groups, **fu_kwargs)
        self.conv2 = nn.Conv2d(out_channels // 2, out_channels, kernel_size=1, groups=groups, bias=False)

    def _pack(self, out_scale: Optional[torch.Tensor] = None):
        """conv1 (+ its BN) and conv2; ``out_scale`` folds the enclosing FFC_BN_ACT.bn_g into conv2."""
        if self._packed is None:
            lib = self._exec.lib
            s1, b1 = _bn_fold(self.conv1[1])
            self._packed = dict(w1=lib.pack_conv_weight(self.conv1[0].weight.detach(), s1, precision=self.precision), b1=b1.contiguous(),
                                w2=lib.pack_conv_weight(self.conv2.weight.detach(), out_scale, precision=self.precision))
        return self._packed

    def run_front(self, xg: L.Tensor4, x1: torch.Tensor, t: torch.Tensor, ws: torch.Tensor, batch: int, stream: int,
                  x1_ready: bool = False, wino_out=None):
        """x1 = relu(bn(conv1(xg))); t = x1 + fu(x1).  (conv2 is fused by the caller.)  ``x1_ready``: the producer of xg already
        wrote x1 from its epilogue (fuse1_operands).  ``wino_out``: the deferred output transform of the previous layer's Winograd local
        conv, finished inside this FourierUnit's first launch (round 4)."""
        pk = self._packed
        if not x1_ready:
            self._exec.conv2d(xg, pk['w1'], L.view(x1), batch, 1, bias=pk['b1'], act=L.ACT_RELU, precision=self.precision, stream=stream)
        self.fu.run(L.view(x1), L.view(t), batch, True, ws, stream, wino_out=wino_out)

    def fuse1_operands(self, x1: torch.Tensor) -> Optional[tuple]:
        """(packed conv1 weights in the channel order of the producing kernel's accumulators, BatchNorm shift, x1 view) for the
        ``fuse1`` argument of the launch that PRODUCES this layer's x_g -- or None when conv1 cannot ride there (shape / precision)."""
        conv = self.conv1[0]
        if self.precision not in (L.PREC_F16X3, L.PREC_BF16X3) or conv.in_channels != 384 or conv.out_channels != 192 or x1.dtype != torch.float32:
            return None
        pk = self._packed
        if pk is None:
            return None
        if 'w1f' not in pk:
            lib = self._exec.lib
            s1, _ = _bn_fold(self.conv1[1])
            order = lib.fuse1_channel_order().to(conv.weight.device)
            pk['w1f'] = lib.pack_conv_weight(conv.weight.detach()[:, order].contiguous(), s1, precision=self.precision)
        return pk['w1f'], pk['b1'], L.view(x1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._exec.check(x)
        x = x.contiguous()
        b, _, h, w = x.shape
        half = self.conv2.in_channels
        with self._exec.range_scope(x, self.precision):
            if self._packed is None or self._packed.get('fused_scale') not in (None, 'none'):
                self._packed = None
                self._pack(None)
            x1 = torch.empty(b, half, h, w, device=x.device, dtype=x.dtype)
            t = torch.empty_like(x1)
            st = self._exec.stream(x)
            self.run_front(L.view(x), x1, t, self.fu.workspace(x1), b, st)
            y = torch.empty(b, self.conv2.out_channels, h, w, device=x.device, dtype=x.dtype)
            self._exec.conv2d(L.view(t), self._packed['w2'], L.view(y), b, 1, precision=self.precision, stream=st)
        return y


# ----------------------------------------------------------------------------------------------------
# FFC / FFC_BN_ACT / FFCResnetBlock
# ----------------------------------------------------------------------------------------------------

class FFC(_HipModule):
    """ffc.py:166-225 (parameter container; the arithmetic is launched by FFC_BN_ACT so that BatchNorm,
    activation and the residual fuse into the conv epilogues)."""

    def __init__(self, in_channels, out_channels, kernel_size, ratio_gin, ratio_gout, stride=1, padding=0, dilation=1,
                 groups=1, bias=False, enable_lfu=True, padding_type='reflect', gated=False, **spectral_kwargs):
        super().__init__()
        assert stride == 1 or stride == 2, "Stride should be 1 or 2."
        if dilation != 1 or groups != 1 or bias or gated:
            raise NotImplementedError('FFC: dilation/groups/bias/gated variants are not used by the shipped LaMa configs')
        if padding_type != 'reflect' and padding != 0:
            raise NotImplementedError('FFC: only reflect padding is implemented')
        self.stride, self.kernel_size, self.padding = stride, kernel_size, padding
        in_cg = int(in_channels * ratio_gin)
        in_cl = in_channels - in_cg
        out_cg = int(out_channels * ratio_gout)
        out_cl = out_channels - out_cg
        self.in_cl, self.in_cg, self.out_cl, self.out_cg = in_cl, in_cg, out_cl, out_cg
        self.ratio_gin, self.ratio_gout = ratio_gin, ratio_gout
        self.global_in_num = in_cg

        def conv(ci, co):
            return nn.Conv2d(ci, co, kernel_size, stride, padding, dilation, groups, bias, padding_mode=padding_type)

        self.convl2l = nn.Identity() if in_cl == 0 or out_cl == 0 else conv(in_cl, out_cl)
        self.convl2g = nn.Identity() if in_cl == 0 or out_cg == 0 else conv(in_cl, out_cg)
        self.convg2l = nn.Identity() if in_cg == 0 or out_cl == 0 else conv(in_cg, out_cl)
        self.convg2g = nn.Identity() if in_cg == 0 or out_cg == 0 else SpectralTransform(in_cg, out_cg, stride, 1, enable_lfu,
                                                                                         **spectral_kwargs)
        self.gated = gated
        self.gate = nn.Identity()
        if in_cl == 0:
            raise NotImplementedError('FFC with ratio_gin == 1 (no local input) is not implemented')
        if in_cg > 0 and (out_cl == 0 or out_cg == 0):
            raise NotImplementedError('FFC with a global input needs both local and global outputs (ratio_gout in (0,1))')
        if in_cg > 0 and (stride != 1 or kernel_size != 3):
            raise NotImplementedError('FFC with a global input is implemented for 3x3 stride-1 layers (the resnet blocks)')

    # -- packing -------------------------------------------------------------------------------------
    def pack(self, sl=None, bl=None, sg=None, bg=None):
        """Packed weights of the four branches; (sl, bl) / (sg, bg) = folded BatchNorm scale / shift of the local / global
        outputs (None = the bare FFC of ffc.py:205-225: no BatchNorm, no bias)."""
        f, lib, prec = self, self._exec.lib, self.precision
        pk = {}
        dev = f.convl2l.weight.device if isinstance(f.convl2l, nn.Conv2d) else f.convl2g.weight.device

        def cat_opt(ts, n):
            ts = [t if t is not None else None for t in ts]
            return None if any(t is None for t in ts) else torch.cat(ts, 0).contiguous()

        if f.in_cg == 0:
            # every output comes from x_l: ONE conv with [convl2l ; convl2g] stacked along Cout
            ws, ss, bs = [], [], []
            if f.out_cl:
                ws.append(f.convl2l.weight.detach()); ss.append(sl); bs.append(bl)
            if f.out_cg:
                ws.append(f.convl2g.weight.detach()); ss.append(sg); bs.append(bg)
            pk['w_all'] = lib.pack_conv_weight(torch.cat(ws, 0), cat_opt(ss, 0), stride=f.stride, precision=prec)
            pk['b_all'] = cat_opt(bs, 0)
        else:
            # local output: conv over the whole state buffer [x_l | x_g] with [convl2l , convg2l] stacked along Cin
            w_lout = torch.cat([f.convl2l.weight.detach(), f.convg2l.weight.detach()], dim=1)
            pk['w_lout'] = lib.pack_conv_weight(w_lout, sl, precision=prec)
            # ... and in Winograd F(2x2, 3x3) form (16 instead of 36 MFMA products per 2 x 2 tile; lama_winograd_*), used by launch() where the
            # plane shape allows; a transformed weight beyond the fp16 range keeps the direct kernel
            if (f.kernel_size == 3 and f.stride == 1 and f.padding == 1 and prec in (L.PREC_F16X3, L.PREC_BF16X3)
                    and w_lout.shape[0] % 128 == 0 and w_lout.shape[1] % 32 == 0):
                try:
                    pk['w_lout_wino'] = lib.pack_winograd_weight(w_lout, sl, prec)
                except LamaRangeError:
                    pass
            pk['b_l'] = None if bl is None else bl.contiguous()
            pk['w_l2g'] = lib.pack_conv_weight(f.convl2g.weight.detach(), sg, precision=prec)
            pk['b_g'] = None if bg is None else bg.contiguous()
            st = f.convg2g
            st._packed = None
            st._pack(sg)
            st._packed['fused_scale'] = id(sg) if sg is not None else 'none'
            st.fu._pack()
        del dev
        return pk

    # -- launch --------------------------------------------------------------------------------------
    def launch(self, pk: dict, act: int, src: torch.Tensor, dst: torch.Tensor, scratch: Optional[dict],
               resid: Optional[torch.Tensor] = None, extra_pad: int = 0, side: Optional['torch.cuda.Stream'] = None,
               x1_ready: bool = False, fuse_next: Optional['SpectralTransform'] = None) -> bool:
        """src [B, in_cl+in_cg, H, W] -> dst [B, out_cl+out_cg, Ho, Wo] (x_l | x_g channel-contiguous), as 1 launch (no global
        input) or 6 launches (conv1x1, rfft2, spectral conv1x1, irfft2+add, fused local conv, fused global conv).

        ``side``: optional second HIP stream, or a ``SidePipe`` (the local convs of a run of layers as a chain of their own: no
        join per layer).  With a plain stream the spectral branch (conv1 -> rfft2 -> spectral 1x1 -> irfft2, HBM/latency
        bound, small LDS footprint) then runs on it concurrently with the MFMA-bound local 3x3 conv of the main stream; the
        two join before the global conv that consumes both.  Works inside hipGraph capture (fork/join via events).

        ``x1_ready``: scratch['x1'] already holds conv1 of this layer (written by the previous layer's global launch).
        ``fuse_next``: the SpectralTransform of the NEXT layer: its conv1 rides in the epilogue of this layer's global launch
        (lama_conv2d_args.fuse1_*).  Returns True when it did (the next layer then passes x1_ready=True)."""
        f, ex, prec = self, self._exec, self.precision
        B = src.shape[0]
        st = ex.stream(src)
        pad = f.padding + extra_pad
        if f.in_cg == 0:
            ex.conv2d(L.view(src), pk['w_all'], L.view(dst), B, f.kernel_size, f.stride, pad, L.PAD_REFLECT, False, pk['b_all'],
                      act, None if resid is None else L.view(resid), precision=prec, stream=st)
            return False
        cl, cg, ocl, ocg = f.in_cl, f.in_cg, f.out_cl, f.out_cg
        spec = f.convg2g
        # (not with the join-free chain of local convs, SidePipe: consecutive layers' local convs may overlap and share scratch['wino'])
        wino = (ex.winograd and pad == 1 and 'w_lout_wino' in pk and scratch is not None and scratch.get('wino') is not None
                and src.dtype == torch.float32 and dst.dtype == torch.float32 and not isinstance(side, SidePipe))

        def local_conv(stream, cooperative):
            """out_xl = act(bn(convl2l(x_l) + convg2l(x_g))) [+ residual]: one 3x3 over the whole state buffer (ffc.py:220)."""
            rl = None if resid is None else L.view(resid, 0, ocl)
            if wino:
                try:
                    # one-stream plans (scratch['defer_out'], _build_plan): only the GEMM half now -- the output transform rides in the rfft2
                    # launch of the NEXT layer (or is flushed by the plan), where it fills the HBM-idle transform phase of the FFT workgroups
                    defer = bool(scratch.get('defer_out')) and side is None
                    pend = ex.winograd_conv3x3(L.view(src), pk['w_lout_wino'], L.view(dst, 0, ocl), B, scratch['wino'], pk['b_l'], act, rl,
                                               precision=prec, stream=stream, defer_out=defer)
                    if defer:
                        scratch['pending_out'] = (pend, scratch['wino'])
                    return
                except LamaError as e:      # a view the Winograd entry does not take (alignment, 32-bit offsets of very tall planes): the direct kernel
                    if e.code != L.ERR_UNSUPPORTED:
                        raise
                    scratch['wino'] = None
            ex.conv2d(L.view(src), pk['w_lout'], L.view(dst, 0, ocl), B, 3, 1, pad, L.PAD_REFLECT, False, pk['b_l'], act, rl, precision=prec,
                      stream=stream, cooperative=cooperative)

        if isinstance(side, SidePipe) and src.is_cuda:
            main = torch.cuda.current_stream(src.device)
            side.stream.wait_stream(main)               # the previous layer's global launch (x_g) -- and every reader of dst's x_l slice
            local_conv(side.stream.cuda_stream, True)
            done = torch.cuda.Event()
            done.record(side.stream)
            spec.run_front(L.view(src, cl, cg), scratch['x1'], scratch['t'], scratch['ws'], B, st, x1_ready)
            if side.prev_local is not None:
                main.wait_event(side.prev_local)        # the global launch reads x_l written by the previous layer's local conv, and
            side.prev_local = done                      # overwrites an x_g slice that local convs up to that one have read
        elif side is not None and src.is_cuda:
            main = torch.cuda.current_stream(src.device)
            side.wait_stream(main)                      # fork: src (and the scratch buffers' last readers) are ordered before
            if not ex.local_first:
                spec.run_front(L.view(src, cl, cg), scratch['x1'], scratch['t'], scratch['ws'], B, side.cuda_stream, x1_ready)
            local_conv(st, True)
            if ex.local_first:
                spec.run_front(L.view(src, cl, cg), scratch['x1'], scratch['t'], scratch['ws'], B, side.cuda_stream, x1_ready)
            main.wait_stream(side)                      # join: t is ready for the global conv
        else:
            spec.run_front(L.view(src, cl, cg), scratch['x1'], scratch['t'], scratch['ws'], B, st, x1_ready, wino_out=scratch.pop('pending_out', None))
            local_conv(st, ex.cooperative_serial)
        # the global branch; by the time it runs this layer's own x1 has been consumed (rfft2 and the x + fu(x) add are upstream of t)
        fuse1 = None
        shape_key = (tuple(src.shape), prec)
        # only when the global launch fills the chip (one 128-pixel tile per CU and more): on a few dozen tiles the epilogue GEMM runs on
        # a few dozen CUs while a launch of its own would use all of them (4 x 256^2: 587 -> 648 images/s with conv1 on its own)
        tiles = (B * ((dst.shape[2] * dst.shape[3] + 127) // 128)) << ex.siblings_log2     # (a part of a batch counts with its siblings)
        if (fuse_next is not None and f.kernel_size == 3 and f.stride == 1 and ocg == 384 and shape_key not in ex.no_fuse1
                and (tiles >= ex.fuse1_min_tiles or ex.injected)):
            fuse1 = fuse_next.fuse1_operands(scratch['x1'])
        gargs = (L.view(src, 0, cl), pk['w_l2g'], L.view(dst, ocl, ocg), B, 3, 1, pad, L.PAD_REFLECT, False, pk['b_g'], act,
                 None if resid is None else L.view(resid, ocl, ocg))
        gkw = dict(x2=L.view(scratch['t']), w2_packed=spec._packed['w2'], precision=prec, stream=st)
        if fuse1 is not None:
            try:
                ex.conv2d(*gargs, fuse1=fuse1, **gkw)
                return True
            except LamaError as e:
                if e.code != L.ERR_UNSUPPORTED:
                    raise
                ex.no_fuse1.add(shape_key)    # e.g. planes too small for the 12 x 1 launch: at this shape conv1 stays a launch of its own
        ex.conv2d(*gargs, **gkw)
        return False

    def out_shape(self, src_shape, extra_pad: int = 0):
        B, _, H, W = src_shape
        pad = self.padding + extra_pad
        Ho = (H + 2 * pad - self.kernel_size) // self.stride + 1
        Wo = (W + 2 * pad - self.kernel_size) // self.stride + 1
        return (B, self.out_cl + self.out_cg, Ho, Wo)

    def make_scratch(self, src_shape, device, alias_t: bool = False) -> Optional[dict]:
        """``alias_t``: t = x1 + fu(x1) is written over x1 (irfft2 reads the residual element where it writes; the global-branch launch reads t
        for the pixels of a tile before its epilogue writes the NEXT layer's x1 for the same pixels)."""
        if self.in_cg == 0:
            return None
        B, _, H, W = src_shape
        half = self.convg2g.conv2.in_channels
        x1 = torch.empty(B, half, H, W, device=device, dtype=_act_dtype(self.precision))
        sc = dict(x1=x1, t=x1 if alias_t else torch.empty_like(x1), ws=self.convg2g.fu.workspace(x1))
        lib = self._exec.lib
        if (self._exec.winograd and self.kernel_size == 3 and self.stride == 1 and self.padding == 1 and self.out_cl
                and lib.winograd_preferred(B, self.out_cl, self.in_cl + self.in_cg, H, W, self.precision)):
            sc['wino'] = torch.empty(lib.winograd_workspace_bytes(B, self.out_cl, H, W) // 4, device=device, dtype=torch.float32)
        return sc

    def forward(self, x):
        """The bare FFC of ffc.py:205-225 (no BatchNorm, no activation): (x_l, x_g) -> (out_xl, out_xg), same launches as the
        fused layer with unit scale, no bias and LAMA_ACT_NONE."""
        x_l, x_g = x if type(x) is tuple else (x, 0)
        self._exec.check(x_l)
        src, cl, cg = _pair_buffer(x_l, x_g)
        if cl != self.in_cl or cg != self.in_cg:
            raise LamaError(f'FFC expected ({self.in_cl},{self.in_cg}) local/global channels, got ({cl},{cg})')
        with self._exec.range_scope(src, self.precision):
            if self._packed is None or (self.in_cg and (self.convg2g._packed or {}).get('fused_scale') != 'none'):
                self._packed = self.pack()
            dst = torch.empty(self.out_shape(src.shape), device=src.device, dtype=_act_dtype(self.precision))
            self.launch(self._packed, L.ACT_NONE, src, dst, self.make_scratch(src.shape, src.device))
        return (dst[:, :self.out_cl] if self.out_cl else 0), (dst[:, self.out_cl:] if self.out_cg else 0)


class FFC_BN_ACT(_HipModule):
    """ffc.py:228-255: FFC + BatchNorm (local, global) + activation; BatchNorm(eval) is folded into the packed weights and the
    epilogue bias, the activation (and the residual add of the enclosing block) runs in the conv epilogues (FFC.launch)."""

    def __init__(self, in_channels, out_channels, kernel_size, ratio_gin, ratio_gout, stride=1, padding=0, dilation=1,
                 groups=1, bias=False, norm_layer=nn.BatchNorm2d, activation_layer=nn.Identity, padding_type='reflect',
                 enable_lfu=True, **kwargs):
        super().__init__()
        if norm_layer is not nn.BatchNorm2d:
            raise NotImplementedError('only BatchNorm2d (eval) is folded into the kernels')
        self.ffc = FFC(in_channels, out_channels, kernel_size, ratio_gin, ratio_gout, stride, padding, dilation, groups, bias,
                       enable_lfu, padding_type=padding_type, **kwargs)
        global_channels = int(out_channels * ratio_gout)
        self.bn_l = nn.Identity() if ratio_gout == 1 else nn.BatchNorm2d(out_channels - global_channels)
        self.bn_g = nn.Identity() if ratio_gout == 0 else nn.BatchNorm2d(global_channels)
        self.act_l = nn.Identity() if ratio_gout == 1 else activation_layer(inplace=True)
        self.act_g = nn.Identity() if ratio_gout == 0 else activation_layer(inplace=True)
        self._act = _act_code(activation_layer)

    # -- packing -------------------------------------------------------------------------------------
    def _pack(self):
        if self._packed is not None:
            return self._packed
        f = self.ffc
        sl, bl = _bn_fold(self.bn_l) if f.out_cl else (None, None)
        sg, bg = _bn_fold(self.bn_g) if f.out_cg else (None, None)
        self._packed = f.pack(sl, bl, sg, bg)
        self._packed['_sg'] = sg
        f._packed = None         # a bare FFC.forward packs its own (unscaled) set
        return self._packed

    # -- launch --------------------------------------------------------------------------------------
    def run(self, src: torch.Tensor, dst: torch.Tensor, scratch: Optional[dict], resid: Optional[torch.Tensor] = None,
            extra_pad: int = 0, side: Optional['torch.cuda.Stream'] = None, x1_ready: bool = False,
            fuse_next: Optional['FFC_BN_ACT'] = None) -> bool:
        f = self.ffc
        if f.in_cg and self._packed is not None and (f.convg2g._packed or {}).get('fused_scale') != id(self._packed.get('_sg')):
            self._packed = None   # a stand-alone SpectralTransform.forward / FFC.forward re-packed conv2 without bn_g
        nxt = None
        if fuse_next is not None and fuse_next.ffc.in_cg:
            fuse_next._pack()                                  # its conv1 weights must exist before they can ride along
            nxt = fuse_next.ffc.convg2g
        return f.launch(self._pack(), self._act, src, dst, scratch, resid, extra_pad, side, x1_ready, nxt)

    def out_shape(self, src_shape, extra_pad: int = 0):
        return self.ffc.out_shape(src_shape, extra_pad)

    def make_scratch(self, src_shape, device, alias_t: bool = False) -> Optional[dict]:
        return self.ffc.make_scratch(src_shape, device, alias_t)

    def forward(self, x, extra_pad: int = 0):
        x_l, x_g = x if type(x) is tuple else (x, 0)
        self._exec.check(x_l)
        src, cl, cg = _pair_buffer(x_l, x_g)
        if cl != self.ffc.in_cl or cg != self.ffc.in_cg:
            raise LamaError(f'FFC_BN_ACT expected ({self.ffc.in_cl},{self.ffc.in_cg}) local/global channels, got ({cl},{cg})')
        # a layer with a global output produces the (x_l | x_g) state of the resnet blocks: fp32 at every precision (see _build_plan)
        dst = torch.empty(self.out_shape(src.shape, extra_pad), device=src.device,
                          dtype=torch.float32 if self.ffc.out_cg else _act_dtype(self.precision))
        with self._exec.range_scope(src, self.precision):
            self.run(src, dst, self.make_scratch(src.shape, src.device), None, extra_pad)
        ocl, ocg = self.ffc.out_cl, self.ffc.out_cg
        return (dst[:, :ocl] if ocl else 0), (dst[:, ocl:] if ocg else 0)


class FFCResnetBlock(_HipModule):
    """ffc.py:258-292 (inline=False): (x_l, x_g) + conv2(conv1((x_l, x_g))); the add rides in conv2's epilogues."""

    def __init__(self, dim, padding_type, norm_layer, activation_layer=nn.ReLU, dilation=1, spatial_transform_kwargs=None,
                 inline=False, **conv_kwargs):
        super().__init__()
        if spatial_transform_kwargs is not None or inline or dilation != 1:
            raise NotImplementedError('FFCResnetBlock: spatial transforms / inline / dilation are not used by big-lama')
        self.conv1 = FFC_BN_ACT(dim, dim, kernel_size=3, padding=dilation, dilation=dilation, norm_layer=norm_layer,
                                activation_layer=activation_layer, padding_type=padding_type, **conv_kwargs)
        self.conv2 = FFC_BN_ACT(dim, dim, kernel_size=3, padding=dilation, dilation=dilation, norm_layer=norm_layer,
                                activation_layer=activation_layer, padding_type=padding_type, **conv_kwargs)
        self.inline = inline

    def run(self, src: torch.Tensor, tmp: torch.Tensor, dst: torch.Tensor, scratch: Optional[dict], side=None, x1_ready: bool = False,
            next_block: Optional['FFCResnetBlock'] = None, fuse: bool = False) -> bool:
        """``x1_ready`` / return value: conv1 of a layer's SpectralTransform may have been computed by the launch that produced the
        layer's input (FFC.launch); ``next_block``: the block whose first conv1 this block's last launch computes."""
        mid_ready = self.conv1.run(src, tmp, scratch, side=side, x1_ready=x1_ready, fuse_next=self.conv2 if fuse else None)
        return self.conv2.run(tmp, dst, scratch, resid=src, side=side, x1_ready=mid_ready,
                              fuse_next=None if next_block is None else next_block.conv1)

    def forward(self, x):
        x_l, x_g = x if type(x) is tuple else (x, 0)
        self._exec.check(x_l)
        src, cl, cg = _pair_buffer(x_l, x_g)
        tmp, dst = torch.empty_like(src, dtype=_act_dtype(self.precision)), torch.empty_like(src)   # the state keeps its element type
        with self._exec.range_scope(src, self.precision):
            self.run(src, tmp, dst, self.conv1.make_scratch(src.shape, src.device))
        return (dst[:, :cl], dst[:, cl:]) if cg else (dst, 0)


class ConcatTupleLayer(nn.Module):
    """ffc.py:295-302.  Free when x_l | x_g already share one buffer (always the case on the fused path)."""

    def forward(self, x):
        assert isinstance(x, tuple)
        x_l, x_g = x
        assert torch.is_tensor(x_l) or torch.is_tensor(x_g)
        if not torch.is_tensor(x_g):
            return x_l
        buf = _adjacent(x_l, x_g)
        return buf if buf is not None else torch.cat(x, dim=1)


# ----------------------------------------------------------------------------------------------------
# stand-alone leaf layers of generator.model (used when a caller runs the Sequential layer by layer)
# ----------------------------------------------------------------------------------------------------

class ReflectionPad2d(_HipModule):
    def __init__(self, padding: int):
        super().__init__()
        self.padding = int(padding)

    def forward(self, x):
        self._exec.check(x)
        x = x.contiguous()
        B, Cn, H, W = x.shape
        y = torch.empty(B, Cn, H + 2 * self.padding, W + 2 * self.padding, device=x.device, dtype=torch.float32)
        self._exec.lib.reflect_pad(L.view(x), self.padding, L.view(y), B, self._exec.stream(x))
        return y


class _Affine(_HipModule):
    def _launch(self, x, scale, shift, act):
        self._exec.check(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        self._exec.lib.affine_act(L.view(x), scale, shift, act, L.view(y), x.shape[0], self._exec.stream(x))
        return y


class BatchNorm2dEval(nn.BatchNorm2d, _Affine):
    """nn.BatchNorm2d parameter container whose (eval) forward is the HIP affine kernel."""

    def __init__(self, num_features):
        nn.BatchNorm2d.__init__(self, num_features)
        self._packed = None
        self._exec = _DEFAULT_EXEC

    def forward(self, x):
        scale, shift = _bn_fold(self)
        return self._launch(x, scale.contiguous(), shift.contiguous(), L.ACT_NONE)


class Activation(_Affine):
    def __init__(self, kind: str):
        super().__init__()
        self.kind = kind

    def forward(self, x):
        return self._launch(x, None, None, _ACT[self.kind])


class ConvTranspose2dUp(nn.ConvTranspose2d, _HipModule):
    """nn.ConvTranspose2d(k3, s2, p1, op1) parameter container (ffc.py:348-351) + HIP forward."""

    def __init__(self, cin, cout):
        nn.ConvTranspose2d.__init__(self, cin, cout, kernel_size=3, stride=2, padding=1, output_padding=1)
        self._packed = None
        self._exec = _DEFAULT_EXEC

    def run(self, src, dst, bn: Optional[nn.BatchNorm2d] = None, act: int = L.ACT_NONE):
        key = id(bn)
        if self._packed is None or self._packed[0] != key:
            if bn is not None:
                scale, shift = _bn_fold(bn)
                bias = self.bias.detach().float() * scale + shift
            else:
                scale, bias = None, self.bias.detach().float()
            self._packed = (key, self._exec.lib.pack_conv_weight(self.weight.detach(), scale, stride=2, transposed=True,
                                                                  precision=self.precision), bias.contiguous())
        _, wp, bias = self._packed
        self._exec.conv2d(L.view(src), wp, L.view(dst), src.shape[0], 3, 2, 1, L.PAD_ZERO, True, bias, act,
                              precision=self.precision, stream=self._exec.stream(src))

    def forward(self, x, bn=None, act=L.ACT_NONE):
        self._exec.check(x)
        x = x.contiguous()
        B, _, H, W = x.shape
        y = torch.empty(B, self.out_channels, 2 * H, 2 * W, device=x.device, dtype=_act_dtype(self.precision))
        with self._exec.range_scope(x, self.precision):
            self.run(x, y, bn, act)
        return y


class Conv2dOut(nn.Conv2d, _HipModule):
    """The 7x7 output conv (ffc.py:361), optionally fused with the preceding reflection pad and the
    output activation."""

    def __init__(self, cin, cout, kernel_size=7, padding=0):
        nn.Conv2d.__init__(self, cin, cout, kernel_size=kernel_size, padding=padding)
        self._packed = None
        self._exec = _DEFAULT_EXEC

    def run(self, src, dst, extra_pad=0, act=L.ACT_NONE):
        if self._packed is None:
            self._packed = (self._exec.lib.pack_conv_weight(self.weight.detach(), None, precision=self.precision),
                            self.bias.detach().float().contiguous())
        wp, bias = self._packed
        self._exec.conv2d(L.view(src), wp, L.view(dst), src.shape[0], self.kernel_size[0], 1, self.padding[0] + extra_pad,
                              L.PAD_REFLECT, False, bias, act, precision=self.precision, stream=self._exec.stream(src))

    def forward(self, x, extra_pad=0, act=L.ACT_NONE):
        self._exec.check(x)
        x = x.contiguous()
        B, _, H, W = x.shape
        k, p = self.kernel_size[0], self.padding[0] + extra_pad
        y = torch.empty(B, self.out_channels, H + 2 * p - k + 1, W + 2 * p - k + 1, device=x.device, dtype=torch.float32)
        with self._exec.range_scope(x, self.precision):
            self.run(x, y, extra_pad, act)
        return y


class LayerSequence(nn.Sequential):
    """``generator.model``: indexable / sliceable like the reference's nn.Sequential (refinement.py:270-289,
    predict_inner_features.py:56,84-86).  Its forward peephole-fuses the patterns the reference spells as
    separate layers: ReflectionPad2d -> conv, ConvTranspose2d -> BatchNorm2d -> ReLU, conv -> output act."""

    def forward(self, x):
        hip = next((m for m in self if isinstance(m, _HipModule)), None)
        first = x[0] if isinstance(x, tuple) else x
        if hip is None or not torch.is_tensor(first):
            return self._forward(x)
        with hip._exec.range_scope(first, hip.precision):     # ONE range-flag read-back for the whole slice
            return self._forward(x)

    def _forward(self, x):
        layers = list(self)
        i, n = 0, len(layers)
        while i < n:
            cur = layers[i]
            nxt = layers[i + 1] if i + 1 < n else None
            if isinstance(cur, ReflectionPad2d) and isinstance(nxt, FFC_BN_ACT) and torch.is_tensor(x):
                x = nxt(x, extra_pad=cur.padding); i += 2
            elif isinstance(cur, ReflectionPad2d) and isinstance(nxt, Conv2dOut):
                act = layers[i + 2] if i + 2 < n and isinstance(layers[i + 2], Activation) else None
                x = nxt(x, extra_pad=cur.padding, act=_ACT[act.kind] if act else L.ACT_NONE)
                i += 3 if act else 2
            elif isinstance(cur, ConvTranspose2dUp) and isinstance(nxt, BatchNorm2dEval) and i + 2 < n and \
                    isinstance(layers[i + 2], Activation):
                x = cur(x, bn=nxt, act=_ACT[layers[i + 2].kind]); i += 3
            else:
                x = cur(x); i += 1
        return x


# ----------------------------------------------------------------------------------------------------
# the generator
# ----------------------------------------------------------------------------------------------------

class FFCResNetGenerator(_HipModule):
    """ffc.py:305-367.  ``forward(x[B,input_nc,H,W]) -> [B,output_nc,H,W]``; ``.model`` has the reference's
    36-entry layout (big-lama) and state_dict keys.

    The fused forward keeps the bottleneck state (x_l | x_g) in one channel-contiguous buffer,
    ping-pongs three such buffers through the residual blocks, reuses one scratch set for every
    SpectralTransform, and (``use_graph=True``) replays the ~230 launches from a captured hipGraph.
    """

    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=3, n_blocks=9, norm_layer=nn.BatchNorm2d,
                 padding_type='reflect', activation_layer=nn.ReLU, up_norm_layer=nn.BatchNorm2d, up_activation=nn.ReLU(True),
                 init_conv_kwargs={}, downsample_conv_kwargs={}, resnet_conv_kwargs={}, spatial_transform_layers=None,
                 spatial_transform_kwargs={}, add_out_act=True, max_features=1024, out_ffc=False, out_ffc_kwargs={}):
        assert (n_blocks >= 0)
        super().__init__()
        if spatial_transform_layers is not None or out_ffc:
            raise NotImplementedError('spatial_transform_layers / out_ffc are not used by big-lama and not implemented')
        if norm_layer is not nn.BatchNorm2d or up_norm_layer is not nn.BatchNorm2d or padding_type != 'reflect':
            raise NotImplementedError('only BatchNorm2d + reflect padding (the big-lama configuration) are implemented')
        up_act = 'relu' if isinstance(up_activation, nn.ReLU) else None
        if up_act is None:
            raise NotImplementedError('up_activation must be ReLU')

        model: List[nn.Module] = [ReflectionPad2d(3),
                                  FFC_BN_ACT(input_nc, ngf, kernel_size=7, padding=0, norm_layer=norm_layer,
                                             activation_layer=activation_layer, **init_conv_kwargs)]
        for i in range(n_downsampling):                                       # ffc.py:320-332
            mult = 2 ** i
            if i == n_downsampling - 1:
                cur_conv_kwargs = dict(downsample_conv_kwargs)
                cur_conv_kwargs['ratio_gout'] = resnet_conv_kwargs.get('ratio_gin', 0)
            else:
                cur_conv_kwargs = downsample_conv_kwargs
            model += [FFC_BN_ACT(min(max_features, ngf * mult), min(max_features, ngf * mult * 2), kernel_size=3, stride=2,
                                 padding=1, norm_layer=norm_layer, activation_layer=activation_layer, **cur_conv_kwargs)]
        mult = 2 ** n_downsampling
        feats_num_bottleneck = min(max_features, ngf * mult)
        for i in range(n_blocks):                                             # ffc.py:338-343
            model += [FFCResnetBlock(feats_num_bottleneck, padding_type=padding_type, activation_layer=activation_layer,
                                     norm_layer=norm_layer, **resnet_conv_kwargs)]
        model += [ConcatTupleLayer()]
        for i in range(n_downsampling):                                       # ffc.py:348-354
            mult = 2 ** (n_downsampling - i)
            model += [ConvTranspose2dUp(min(max_features, ngf * mult), min(max_features, int(ngf * mult / 2))),
                      BatchNorm2dEval(min(max_features, int(ngf * mult / 2))), Activation(up_act)]
        model += [ReflectionPad2d(3), Conv2dOut(ngf, output_nc, kernel_size=7, padding=0)]
        if add_out_act:                                                       # ffc.py:362-363 + base.get_activation
            kind = 'tanh' if add_out_act is True else add_out_act
            if kind not in ('tanh', 'sigmoid'):
                raise ValueError(f'Unknown activation kind {kind}')
            model.append(Activation(kind))
        self.model = LayerSequence(*model)
        self.use_graph = False
        # spectral branch (conv1 -> rfft2 -> spectral 1x1 -> irfft2) on a second stream next to the MFMA-bound local 3x3 conv,
        # joined before the global conv (fused forward only; fork / join are captured into the hipGraph).  +3-4 % images/s.
        # Round 1 had to switch it off: FFT planes came out wrong when FFT and conv workgroups shared a CU.  Root cause (round 2,
        # DESIGN.md 4.3): packed-fp32 VALU instructions with an op_sel swizzle are corrupted by another kernel's MFMA on the same
        # SIMD; the library is now built without them (lama_amd/build.py) and 100 000 overlapped layer runs are bit-identical.
        self.overlap_streams = True
        # True: SpectralTransform.conv1 of every layer but the first rides in the epilogue of the launch that produces its input (the
        # global branch of the previous layer): 35 of 36 pointwise launches less per forward.  Worth -10 us per layer in
        # serial launch order; off by default since the second stream hides the stand-alone conv1 beside the cooperative local conv
        # (705 -> 719 images/s without it, profiles/r02_ab_overlap_and_fuse1_final.txt)
        self.fuse_conv1 = None            # None = by launch order (below), True / False = forced
        # with the Winograd local conv (FFC.launch) a plan runs on ONE stream with conv1 fused into the global launch (see _build_plan)
        self.serial_with_winograd = True
        self.serial_when_full = True      # round 6: ... and for direct-conv plans whose bottleneck launches fill the chip (see _build_plan)
        self.fuse_conv1_serial = True
        # the local convs of the residual blocks as a chain of their own on the second stream (SidePipe) instead of a fork + join per
        # layer.  Off: inside a hipGraph ROCm 7.2 spreads that topology over three queues and every edge becomes a ~10 us cross-queue
        # signal (727 -> 695 images/s, profiles/r02_ab_pipeline_local.txt); bit-identical results either way.
        self.pipeline_local = False
        # fp16-split range watch (lama_conv2d_args.range_flag): one 4-byte read-back per forward; when an activation beyond 65504
        # (or a NaN) was met the forward is repeated with the 3-term bf16 split (fp32 exponent range) and the generator stays
        # on it.  auto_fallback = False raises LamaRangeError instead.
        self.auto_fallback = True
        # True: ``forward`` does NOT read the flag back (no host synchronisation per forward: step k + 1 queues behind step k, and a serving
        # loop's launch thread never waits for the GPU); the flag stays raised on the device and the CALLER asks at its own synchronisation
        # point -- ``check_range()`` -- before it trusts the results produced since the last check.  predict.py (per bucket) and bench.py
        # (closing barrier of the timed region) run this way.  Default False: a plain ``generator(x)`` is self-checking like the reference.
        self.defer_range_check = False
        # LAMA_PREC_F16 (fp16 activations in HBM, BASELINE configs[2]): the layers BEHIND the resnet blocks -- the three ConvTranspose2d + BN + ReLU and
        # the 7x7 head -- keep fp32 tensors and the 3-term split.  Measured on the oracle with a rounding to fp16 wherever the path has one
        # (tools/fp16_by_tensor.py, 1 x 1024^2): every ONE of the three upsampled tensors in fp16 costs 5-9e-3 max-abs on its own (the features
        # are largest behind the 18 blocks).  On the GPU at 4 x 1024^2: 1.68e-2 -> 1.19e-2 max-abs for 218 -> 210 images/s.
        # False = the round-3 layout (everything between stem and head fp16).
        self.f16_fp32_tail = True
        # working set of a residual layer (8 x 512^2: three 67 MB state buffers + x1, t, two spectra, the Winograd partial sums = 337 MB against
        # 256 MB of Infinity Cache): the block output in place of its input, t = x1 + fu(x1) in place of x1
        self.inplace_residual = True
        self.alias_t = True
        self.alias_wino = True
        # the Winograd output transform of a residual layer inside the rfft2 launch of the next one (one-stream plans)
        self.defer_wino_out = True
        # activation-buffer sets (+ captured hipGraphs) per input shape: 2.2 GB at 8 x 512^2, so only the most recently used
        # ``max_plans`` shapes are kept (a directory of many image sizes would otherwise fill HBM)
        self.max_plans = 4
        # Parts of a batch as PARALLEL BRANCHES of the plan (round 5).  Every launch of the one-stream forward fills the chip with workgroups that run
        # the same phase at the same time, and between two launches the chip drains and refills (~190 boundaries per forward).  Two (four) parts of
        # the batch, each with its own buffers, issued on streams of their own inside the SAME captured hipGraph (kernel branches of a graph run
        # side by side on ROCm 7.2; two graphs do not) put one part's memory-bound launches beside the other's MFMA-bound ones and fill each
        # other's boundaries: 8 x 512^2 +3-5 %, 4 x 1024^2 +7-8 %, 16 x 512^2 in four parts +18 % (same box, bit-identical output:
        # profiles/r05_split_batch.txt).  Each part's launches tell the library that they share the chip (LAMA_CONV_SIBLINGS_*, v109) and get the
        # kernel geometry of the whole batch.  None = by shape (_split_parts: GPU, graph mode, every precision but exact fp32, from 256 bottleneck tiles on: four parts, two when 4 does not divide the batch),
        # 1 = off, 2 / 4 = forced.
        self.split_batch = None
        self.n_downsampling = n_downsampling
        # The rule's choice is VERIFIED once per input shape (graph mode): both the split and the one-part graph are captured and replayed alternately
        # (median of 9 each); the split plan is kept only if it is at least 2 % faster (split_margin; a tie keeps the one-part plan, so the decision is
        # the same from process to process) and the loser's buffers are released at once.  Parallel kernel branches are a property of the runtime: where they
        # are serialised -- rocprofv3's kernel trace does that (profiles/r05_overlap_under_rocprof.txt: 33 ms per replay instead of 9.4) -- the
        # quarter-size launches of a split plan would run one after the other on a quarter of the chip each; the check then keeps the one-part
        # plan.  ~100 ms once per shape.  False: trust the rule.
        self.verify_split = True
        self._assume_graph = False        # set by a caller that captures this generator's plain launches into a graph of its own (HostFedStep)
        self._split_ok = {}               # (shape, device) -> False where the check rejected the split plan
        self.split_decisions = {}         # (shape, device) -> what tune_split measured and kept (bench.py prints it)
        self.split_margin = 0.02          # the split plan must be this much faster (median) to be kept
        self.split_replays = 9            # timed replays of each graph in tune_split
        # False: ``forward`` returns the plan's own output buffer instead of a copy of it -- for callers that consume the result before this
        # generator's next forward of the same shape (DefaultInpaintingTrainingModule with keep_predicted_image = False: blend reads it at once)
        self.clone_output = True
        self._plans = collections.OrderedDict()
        super().train(False)

    # ------------------------------------------------------------------------------------------------
    def _precision_hook(self, precision: int):
        if precision != L.PREC_F16 or not getattr(self, 'f16_fp32_tail', False):
            return
        tail = False
        for lay in self.model:
            tail = tail or isinstance(lay, ConcatTupleLayer)
            if tail and isinstance(lay, _HipModule):
                lay.precision = L.PREC_F16X3             # fp32 tensors, fp32-accurate arithmetic

    def _invalidate(self):
        super()._invalidate()
        self._plans = collections.OrderedDict()
        self._split_ok = {}

    def _build_plan(self, shape, device, one_stream: bool = False):
        """Pre-allocate every activation buffer for an input shape and record the launch list.  ``one_stream``: never a second stream inside this
        plan (the parts of a split plan are branches of their own already: a fork + join per layer INSIDE a branch made hipStreamEndCapture
        segfault on ROCm 7.2)."""
        layers = list(self.model)
        steps = []
        bufs = {}
        cur_shape = tuple(shape)
        cur = 'in'
        i, n = 0, len(layers)
        pad_pending = 0

        adt = _act_dtype(self.precision)

        def new(name, shp, dtype=None):
            # PREC_F16: the activations between the stem's output and the head's input are fp16, EXCEPT the residual stream of the
            # resnet blocks (the (x_l | x_g) state that every block adds to: 18 fp16 roundings of it cost 2x the end-to-end error);
            # the image that leaves the head is fp32
            bufs[name] = torch.empty(shp, device=device, dtype=dtype or adt)
            return name

        scratch = None
        k = 0
        while i < n:
            lay = layers[i]
            if isinstance(lay, ReflectionPad2d):
                pad_pending = lay.padding; i += 1; continue
            if isinstance(lay, FFC_BN_ACT):
                shp = lay.out_shape(cur_shape, pad_pending)
                feeds_blocks = lay.ffc.out_cg > 0      # the layer that produces the first (x_l | x_g) state
                dst = new(f'a{k}', shp, torch.float32 if feeds_blocks else None); k += 1
                steps.append(('ffc', lay, cur, dst, pad_pending))
                cur, cur_shape, pad_pending = dst, shp, 0
            elif isinstance(lay, FFCResnetBlock):
                if 'rt' not in bufs:                   # first block: the scratch set of every SpectralTransform (None without a global branch)
                    scratch = lay.conv1.make_scratch(cur_shape, device, alias_t=self.alias_t)
                    new('rt', cur_shape)
                # the block's output IN PLACE of its input (round 4): the second layer reads the input only as its residual operand, element
                # by element where it writes (Winograd out kernel, global-branch epilogue), so the 18 blocks walk on TWO state buffers
                # instead of three: 67 MB less working set per layer at 8 x 512^2
                if self.inplace_residual and cur != 'in' and bufs[cur].dtype == torch.float32:
                    dst = cur
                else:
                    dst = 'rA' if cur != 'rA' else 'rB'
                    if dst not in bufs:
                        new(dst, cur_shape, torch.float32)
                steps.append(('res', lay, cur, 'rt', dst))
                cur = dst
            elif isinstance(lay, ConcatTupleLayer):
                pass
            elif isinstance(lay, ConvTranspose2dUp):
                bn = layers[i + 1] if i + 1 < n and isinstance(layers[i + 1], BatchNorm2dEval) else None
                act = layers[i + 2] if bn is not None and i + 2 < n and isinstance(layers[i + 2], Activation) else None
                B, _, H, W = cur_shape
                shp = (B, lay.out_channels, 2 * H, 2 * W)
                dst = new(f'a{k}', shp, _act_dtype(lay.precision)); k += 1
                if bn is not None and act is not None:
                    steps.append(('up', lay, cur, dst, bn, _ACT[act.kind])); i += 2
                else:
                    steps.append(('up', lay, cur, dst, None, L.ACT_NONE))
                cur, cur_shape = dst, shp
            elif isinstance(lay, Conv2dOut):
                act = layers[i + 1] if i + 1 < n and isinstance(layers[i + 1], Activation) else None
                B, _, H, W = cur_shape
                kk, p = lay.kernel_size[0], lay.padding[0] + pad_pending
                shp = (B, lay.out_channels, H + 2 * p - kk + 1, W + 2 * p - kk + 1)
                dst = new('out', shp, torch.float32)
                steps.append(('out', lay, cur, dst, pad_pending, _ACT[act.kind] if act else L.ACT_NONE))
                cur, cur_shape, pad_pending = dst, shp, 0
                if act:
                    i += 1
            else:
                raise LamaError(f'no fused plan for layer {type(lay).__name__}; run generator.model layer by layer')
            i += 1
        # Launch order.  Where the residual blocks' local conv takes the Winograd kernel (scratch['wino']) the plan is ONE stream: that
        # kernel holds a CU's whole register file (one wave per SIMD, 256 accumulator registers), so nothing of the spectral branch can run
        # beside it and a second stream buys only cross-queue signals; in a one-stream order conv1 of the next layer is cheapest in the
        # global launch's epilogue (round 3, same box: 716 / 714 / 720-724 images/s for two streams / one / one + fused conv1,
        # profiles/r03_ab_launch_order.txt).  Elsewhere (planes the Winograd kernel does not take) the round-2 order stays: the spectral
        # branch on a second stream beside the cooperative direct conv.
        # (the predicate FFC.launch uses: the workspace exists AND every residual layer holds Winograd-packed weights -- a transformed
        # weight beyond the fp16 range keeps the direct kernel, FFC.pack -- otherwise the direct local conv would run on one stream in
        # its non-cooperative geometry with nothing beside it)
        res_layers = [l for st_ in steps if st_[0] == 'res' for l in (st_[1].conv1, st_[1].conv2)]
        wino = bool(scratch and scratch.get('wino') is not None and self._exec.winograd and res_layers
                    and all('w_lout_wino' in l._pack() for l in res_layers))
        if scratch and scratch.get('wino') is not None and not wino:
            scratch['wino'] = None                   # 33 MB at 8 x 64 x 64 that no launch would read
        serial = wino and self.serial_with_winograd
        if not wino and not serial and self.serial_when_full and res_layers and scratch and scratch.get('x1') is not None:
            # round 6 (planes the Winograd launch does not take -- any width that is not 32 / 64 / 128 / 256): when the bottleneck launches fill the
            # chip by themselves (>= 200 tiles of 128 pixels) a second stream has nothing to run the spectral branch ON -- the mixed-radix FFT
            # workgroups hold a CU's whole LDS -- and one stream with conv1 in the global launch's epilogue wins (1080 x 1920: 13.7 -> 13.1 ms,
            # 1344^2: 12.8 -> 12.1; 1000 x 1504, 184 tiles: 11.9 -> 12.3, so it keeps its two streams; profiles/r06_odd_shapes.txt)
            B_, _, H_, W_ = scratch['x1'].shape
            serial = B_ * ((H_ * W_ + 127) // 128) >= 200
        if wino and serial and self.defer_wino_out:
            # the Winograd output transform of layer l rides in the rfft2 launch of layer l + 1 (FFC.launch): the partial sums P live until then,
            # i.e. while that rfft2 writes the FIRST spectrum -- P goes behind it, over the second spectrum (written by the spectral GEMM, after
            # that launch) and a tail: 211 + 7.5 MB touched per layer at 8 x 512^2
            B_, C1, H_, W_ = scratch['x1'].shape
            spec_elems = (((B_ * 2 * C1 * H_ * (W_ // 2 + 1) * 4) + 255) & ~255) // 4
            wn = scratch['wino']
            ws = torch.empty(max(scratch['ws'].numel(), spec_elems + wn.numel()), device=device, dtype=torch.float32)
            scratch['ws'], scratch['wino'] = ws, ws[spec_elems:spec_elems + wn.numel()]
            scratch['defer_out'] = True
        elif wino and serial and self.alias_wino:
            # one-stream order: the FourierUnit's spectra (rfft2 -> GEMM -> irfft2) are dead when the Winograd launches start and the
            # Winograd partial sums are dead when the next layer's rfft2 starts: one allocation for both (8 x 512^2: 33.5 of 52 MB shared)
            ws, wn = scratch['ws'], scratch['wino']
            if wn.numel() <= ws.numel():
                scratch['wino'] = ws[:wn.numel()]
            else:
                scratch['ws'] = wn[:ws.numel()]
        side = torch.cuda.Stream(device=device) if (self.overlap_streams and not serial and not one_stream and torch.device(device).type == 'cuda') else None
        return dict(steps=steps, bufs=bufs, scratch=scratch, out=cur, graph=None, static_in=None, side=side,
                    fuse=self.fuse_conv1 or (serial and self.fuse_conv1 is not False and self.fuse_conv1_serial))

    def _run_plan(self, plan, x):
        if 'parts' in plan:
            return self._run_split(plan, x)
        bufs = plan['bufs']

        def B(name):
            return x if name == 'in' else bufs[name]

        steps = plan['steps']
        x1_ready = False
        if plan['scratch']:
            plan['scratch'].pop('pending_out', None)        # (left behind by a run that raised)
        pipe = SidePipe(plan['side']) if (plan['side'] is not None and self.pipeline_local and x.is_cuda) else None
        for i, st in enumerate(steps):
            kind = st[0]
            if kind == 'ffc':
                _, lay, s, d, pad = st
                lay.run(B(s), B(d), plan['scratch'] if lay.ffc.in_cg else None, None, pad, side=plan['side'])
                x1_ready = False
            elif kind == 'res':
                _, lay, s, t, d = st
                fuse = plan.get('fuse', self.fuse_conv1)
                nxt = steps[i + 1][1] if fuse and i + 1 < len(steps) and steps[i + 1][0] == 'res' else None
                x1_ready = lay.run(B(s), B(t), B(d), plan['scratch'], side=pipe or plan['side'], x1_ready=x1_ready, next_block=nxt,
                                   fuse=fuse)
                if i + 1 == len(steps) or steps[i + 1][0] != 'res':
                    # (plan['scratch'] is None when the blocks have no global branch: resnet_conv_kwargs ratio_gin = ratio_gout = 0)
                    pend = plan['scratch'].pop('pending_out', None) if plan['scratch'] else None
                    if pend is not None:            # the last residual layer's output transform has no rfft2 launch to ride in
                        self._exec.lib.winograd_out(pend[0], pend[1], self._exec.stream(x))
                    if pipe is not None:
                        pipe.join()                 # the last local conv: everything downstream reads its x_l
            elif kind == 'up':
                _, lay, s, d, bn, act = st
                lay.run(B(s), B(d), bn, act)
            else:
                _, lay, s, d, pad, act = st
                lay.run(B(s), B(d), pad, act)
        return bufs[plan['out']]

    def drop_plan(self, shape, device) -> None:
        """Free the activation buffers / captured graph of one input shape (predict.py drops a bucket's plan when it is done)."""
        self._plans.pop(self._plan_key(shape, device), None)

    @staticmethod
    def _plan_key(shape, device):
        """Key of the plan cache and of the split verdicts: 'cuda' and 'cuda:<current>' are ONE device (predict / HostFedStep pass
        torch.device('cuda'), every forward sees x.device = 'cuda:0' -- two keys meant a verdict and a tuned plan no forward ever found)."""
        return (tuple(int(v) for v in shape), _Exec._dev_key(device))

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        self._exec.check(input)
        x = input.contiguous()
        try:
            with self._exec.range_scope(x, self.precision, deferred=self.defer_range_check):
                return self._forward(x)
        except LamaRangeError as e:
            if not self.auto_fallback or self.precision not in (L.PREC_F16X3, L.PREC_F16):
                raise
            warnings.warn(f'lama_amd: {e}; switching this generator to the 3-term bf16 split (PREC_BF16X3)')
            self.set_precision(L.PREC_BF16X3)
            return self.forward(input)

    def check_range(self, device=None, reduce=None) -> bool:
        """The other half of ``defer_range_check``: one 4-byte read-back (a host synchronisation) of the range flag the forwards since the last
        check have been OR-ing into.  Returns True when every one of them stayed inside the fp16 split's range.  Otherwise their results are NOT
        valid: with ``auto_fallback`` the generator switches to the 3-term bf16 split and False is returned -- the caller re-runs those inputs --
        without it LamaRangeError is raised.  ``reduce`` (multi-rank callers): bool -> bool, the OR over all ranks, called UNCONDITIONALLY (a
        collective), so that every rank switches precision and re-runs alike."""
        bad = self._exec.range_flag_raised(device)
        if reduce is not None:
            bad = bool(reduce(bad))
        if not bad:
            return True
        if not self.auto_fallback or self.precision not in (L.PREC_F16X3, L.PREC_F16):
            raise LamaRangeError(_RANGE_MSG)
        warnings.warn(f'lama_amd: {_RANGE_MSG}; switching this generator to the 3-term bf16 split (PREC_BF16X3): re-run the inputs since the last check')
        self.set_precision(L.PREC_BF16X3)
        return False

    def _split_parts(self, shape, device) -> int:
        """How many parallel parts the plan of this input shape has (``split_batch``)."""
        B = int(shape[0])
        n = self.split_batch
        if n is None:
            if torch.device(device).type != 'cuda' or self.precision not in (L.PREC_F16X3, L.PREC_BF16X3, L.PREC_F16):
                return 1
            if not (self.use_graph or self._assume_graph):
                return 1                                    # plain launches: four parts are four times the host calls (launch-bound), no gain
            h, w = int(shape[2]) >> self.n_downsampling, int(shape[3]) >> self.n_downsampling
            tiles = B * ((h * w + 127) // 128)              # 128-pixel tiles of the bottleneck: the parts TOGETHER must fill the chip (one
            n = 4 if tiles >= 256 else 1                    # 12-wave workgroup per tile and CU), so nothing splits below 256 tiles; four parts
            while n > 1 and B % n:                          # measured >= two at every size tried (profiles/r05_split_batch.txt)
                n //= 2
            if n > 1 and self._split_ok.get(self._plan_key(shape, device)) is False:
                return 1                                    # this runtime does not run the parts side by side (verify_split)
            return n
        n = int(n)
        if n < 1 or n & (n - 1) or n > 4 or B % n:
            raise LamaError(f'split_batch={n}: 1, 2 or 4 parts that divide the batch ({B})')
        return n

    def _build_split_plan(self, shape, device, n: int) -> dict:
        B = int(shape[0])
        h = B // n
        parts = [self._build_plan((h,) + tuple(shape[1:]), device, one_stream=True) for _ in range(n)]
        out_shape = (B,) + tuple(parts[0]['bufs'][parts[0]['out']].shape[1:])
        out_full = torch.empty(out_shape, device=device, dtype=parts[0]['bufs'][parts[0]['out']].dtype)
        for i, pl in enumerate(parts):
            pl['bufs'][pl['out']] = out_full[i * h:(i + 1) * h]       # the parts write their images side by side
        streams = [torch.cuda.Stream(device=device) for _ in range(n - 1)] if torch.device(device).type == 'cuda' else []
        return dict(parts=parts, part_batch=h, streams=streams, out_full=out_full, nsplit=n, graph=None, static_in=None, scratch=None, side=None)

    def _run_split(self, plan, x):
        ex, parts, h = self._exec, plan['parts'], plan['part_batch']
        main = torch.cuda.current_stream(x.device) if x.is_cuda else None
        keep = ex.siblings_log2
        ex.siblings_log2 = len(parts).bit_length() - 1
        try:
            for i in range(1, len(parts)):
                xi = x[i * h:(i + 1) * h]
                if main is not None:
                    s = plan['streams'][i - 1]
                    s.wait_stream(main)                      # fork (captured into the graph as an edge)
                    with torch.cuda.stream(s):
                        self._run_plan(parts[i], xi)
                else:
                    self._run_plan(parts[i], xi)
            self._run_plan(parts[0], x[:h])
            if main is not None:
                for s in plan['streams']:
                    main.wait_stream(s)                      # join
        finally:
            ex.siblings_log2 = keep
        return plan['out_full']

    def _plan_for(self, shape, device) -> dict:
        key = self._plan_key(shape, device)
        n = self._split_parts(shape, device)
        plan = self._plans.get(key)
        if plan is not None and plan.get('nsplit', 1) != n:
            plan = None                                      # split_batch changed since this plan was built
            del self._plans[key]
        if plan is None:
            while len(self._plans) >= max(1, self.max_plans):
                self._plans.popitem(last=False)
            plan = self._plans[key] = self._build_split_plan(shape, device, n) if n > 1 else self._build_plan(shape, device)
        else:
            self._plans.move_to_end(key)
        return plan

    def input_buffer(self, shape, device) -> torch.Tensor:
        """The tensor this shape's plan READS its input from (the captured hipGraph's static input when ``use_graph``): a caller that produces
        the generator's input itself (DefaultInpaintingTrainingModule: mask_compose) writes it here and passes it to ``forward``, which then
        skips its staging copy (33.5 MB read + written per step at 8 x 512^2).  Valid until the plan is dropped."""
        device = torch.device(device)
        plan = self._plan_for(shape, device)
        if plan['static_in'] is None:
            plan['static_in'] = torch.empty(tuple(shape), device=device, dtype=torch.float32)
        return plan['static_in']

    def _capture(self, plan, device):
        """Warm the plan up on a side stream (packs weights) and capture its launches into plan['graph']; plan['static_in'] holds the input."""
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):           # warm-up outside capture
            self._run_plan(plan, plan['static_in'])
        torch.cuda.current_stream(device).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            plan['static_out'] = self._run_plan(plan, plan['static_in'])
        plan['graph'] = g

    def tune_split(self, shape, device) -> int:
        """``verify_split``: decide ONCE per input shape whether the split plan the rule proposes really runs its parts side by side (see
        __init__).  Builds and captures both graphs on uniform-random input (under a scratch range flag: what noise raises is nobody's
        report), replays them alternately -- 9 timed replays each after a warm-up pair -- and keeps the split plan only when its MEDIAN is at
        least ``split_margin`` (2 %) below the one-part plan's: a tie is the one-part plan, so two processes on one box decide alike.  The
        loser's buffers and graph are released before this returns; the decision is recorded in ``split_decisions`` (and ``split_timing``).
        Host-synchronising; never called inside a stream capture."""
        device = torch.device(device)
        key = self._plan_key(shape, device)
        n = self._split_parts(shape, device)
        if (n <= 1 or self.split_batch is not None or not self.verify_split or key in self._split_ok or device.type != 'cuda'
                or torch.cuda.is_current_stream_capturing()):
            return n
        x = torch.rand(tuple(shape), device=device, dtype=torch.float32)
        cand, times = {}, {n: [], 1: []}
        with self._exec.scratch_range_scope(x, self.precision):
            for parts in (n, 1):
                plan = self._build_split_plan(shape, device, parts) if parts > 1 else self._build_plan(shape, device)
                plan['static_in'] = x
                self._capture(plan, device)
                cand[parts] = plan
            # alternate the two graphs (the first replays after an idle period run on ramping clocks)
            for rnd in range(1 + self.split_replays):
                for parts in (n, 1):
                    g = cand[parts]['graph']
                    if rnd == 0:
                        g.replay()                            # warm-up, untimed
                        continue
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record()
                    g.replay()
                    t1.record()
                    torch.cuda.synchronize(device)
                    times[parts].append(t0.elapsed_time(t1))
        med = {parts: sorted(ts)[len(ts) // 2] for parts, ts in times.items()}
        ok = med[n] <= (1.0 - self.split_margin) * med[1]
        self._split_ok[key] = ok
        self.split_decisions[key] = dict(parts=n, ms_split=round(med[n], 3), ms_one_part=round(med[1], 3), kept=n if ok else 1,
                                         margin=self.split_margin, replays=self.split_replays)
        self.split_timing = {key: self.split_decisions[key]}
        plan = cand.pop(n if ok else 1)           # (its captured graph and its input buffer -- the random tensor -- stay: forward stages into it,
        cand.clear()                              #  input_buffer hands it to a caller that writes its input in place); the loser is released now
        del x
        while len(self._plans) >= max(1, self.max_plans):
            self._plans.popitem(last=False)
        self._plans[key] = plan
        torch.cuda.empty_cache()
        return n if ok else 1

    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.use_graph and x.is_cuda and self.verify_split and self.split_batch is None:
            self.tune_split(x.shape, x.device)
        plan = self._plan_for(x.shape, x.device)
        # ``clone_output`` False: the caller consumes the result before this generator's next forward (the plan's output buffer is returned)
        fin = (lambda t: t.clone()) if self.clone_output else (lambda t: t)
        if not (self.use_graph and x.is_cuda):
            return fin(self._run_plan(plan, x))
        own = plan['static_in'] is not None and x.data_ptr() == plan['static_in'].data_ptr()       # written in place by the caller (input_buffer)
        if plan['graph'] is None:
            if not own:
                plan['static_in'] = torch.empty_like(x)
                plan['static_in'].copy_(x)
            self._capture(plan, x.device)
        if not own:
            plan['static_in'].copy_(x)
        plan['graph'].replay()
        return fin(plan['static_out'])
