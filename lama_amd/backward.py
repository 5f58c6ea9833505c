"""Explicit reverse pass (input gradients only) through ``generator.model[first_resblock:]`` on the HIP kernels.

Feature refinement (saicinpainting/evaluation/refinement.py:86-174) optimises the bottleneck features (z1, z2) with Adam; the
reference obtains d loss / d (z1, z2) from ``loss.backward()`` through 18 FFCResnetBlocks, three ConvTranspose2d + BN + ReLU and
the 7x7 head + sigmoid.  Here that autograd graph is replaced by a hand-written tape: the forward stores what the derivatives
need (every ReLU output, the post-ReLU spectra of the FourierUnits), and the reverse pass is a chain of the SAME kernels the
forward uses -- a dgrad is a convolution with transposed / flipped weights:

  * 3x3 reflect conv (ffc.py:188-196):  d xp = conv_zero_pad2(g, W^T flipped) on the (H+2) x (W+2) padded grid, then the adjoint of
    the reflection folds the border ring back (lama_reflect_pad_bwd);
  * 1x1 convs (ffc.py:128-140): conv1x1(g, W^T);
  * FourierUnit (ffc.py:76-113): y = irfft2(relu(W rfft2(x) + b)).  With D = diag(1, 2, .., 2, 1) over the kx bins (the weights of
    the half spectrum), adjoint(irfft2) = D rfft2 and adjoint(rfft2) = irfft2 D^-1; D commutes with the per-point channel GEMM and
    with the (positive) ReLU mask, so  dx = irfft2( W^T ( [relu output > 0] * rfft2(dy) ) )  -- exactly the forward kernels;
  * ConvTranspose2d(k3, s2, p1, op1) (ffc.py:348-351): dgrad = conv2d(g, W, stride 2, zero pad 1) with the SAME weight tensor;
  * BatchNorm(eval) is a per-channel scale: folded into the dgrad weights on their INPUT-channel axis; ReLU / sigmoid derivatives
    come from the stored outputs (lama_act_bwd).

Gradients are small (a mean over ~1e6 pixels): the reverse pass always runs on the 3-term bf16 split (fp32 exponent range) or on
exact fp32 -- never on the fp16 split, whose lo term underflows below ~1e-4.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import ffc as F
from ._lib import LamaError


def _tflip(w: torch.Tensor, in_scale: Optional[torch.Tensor]) -> torch.Tensor:
    """Conv weight [O, I, kh, kw] (+ BatchNorm scale per O) -> dgrad conv weight [I, O, kh, kw]: transposed, taps flipped, the
    scale on the (new) input-channel axis."""
    w = w.detach().float()
    if in_scale is not None:
        w = w * in_scale.view(-1, 1, 1, 1)
    return w.permute(1, 0, 2, 3).flip(2, 3).contiguous()


class _FFCLayerTape:
    """One FFC_BN_ACT layer of a resnet block (both inputs and both outputs present), forward with taping + reverse."""

    def __init__(self, lay: F.FFC_BN_ACT, ex: F._Exec, bwd_precision: int):
        self.lay, self.ex, self.bprec = lay, ex, bwd_precision
        f = lay.ffc
        if not (f.in_cg and f.out_cg and f.in_cl and f.out_cl and f.kernel_size == 3 and f.stride == 1):
            raise LamaError('backward: only the resnet-block FFC layers (3x3, stride 1, local + global) are supported')
        self._bw = None

    # -- forward ---------------------------------------------------------------------------------------------------
    def alloc(self, B, H, W, dev):
        f = self.lay.ffc
        half = f.convg2g.conv2.in_channels
        wf = W // 2 + 1
        return dict(out=torch.empty(B, f.out_cl + f.out_cg, H, W, device=dev), x1=torch.empty(B, half, H, W, device=dev),
                    s2=torch.empty(B, 2 * half, H, wf, device=dev))

    def forward(self, src: torch.Tensor, tape: dict, sh: dict):
        lay, ex = self.lay, self.ex
        f, prec = lay.ffc, lay.precision
        pk = lay._pack()
        spec = f.convg2g
        sp = spec._packed
        fuw, fub = spec.fu._pack()
        B = src.shape[0]
        st = ex.stream(src)
        cl, cg, ocl, ocg = f.in_cl, f.in_cg, f.out_cl, f.out_cg
        x1, s2, dst = tape['x1'], tape['s2'], tape['out']
        ex.conv2d(L.view(src, cl, cg), sp['w1'], L.view(x1), B, 1, bias=sp['b1'], act=L.ACT_RELU, precision=prec, stream=st)
        ex.lib.rfft2(L.view(x1), L.view(sh['s1']), B, sh['fftws'], st)
        ex.conv2d(L.view(sh['s1']), fuw, L.view(s2), B, 1, bias=fub, act=L.ACT_RELU, precision=prec, stream=st)
        ex.lib.irfft2(L.view(s2), L.view(x1), L.view(sh['t']), B, sh['fftws'], st)
        done = False
        if sh.get('wino') is not None and 'w_lout_wino' in pk and ex.winograd:
            # the local 3x3 as Winograd F(2x2, 3x3), as in the predict path (FFC.launch); shapes / views the entry does not take: direct kernel
            try:
                ex.winograd_conv3x3(L.view(src), pk['w_lout_wino'], L.view(dst, 0, ocl), B, sh['wino'], pk['b_l'], lay._act, None,
                                    precision=prec, stream=st)
                done = True
            except LamaError as e:
                if e.code != L.ERR_UNSUPPORTED:
                    raise
                sh['wino'] = None
        if not done:
            ex.conv2d(L.view(src), pk['w_lout'], L.view(dst, 0, ocl), B, 3, 1, 1, L.PAD_REFLECT, False, pk['b_l'], lay._act, None,
                      precision=prec, stream=st)
        ex.conv2d(L.view(src, 0, cl), pk['w_l2g'], L.view(dst, ocl, ocg), B, 3, 1, 1, L.PAD_REFLECT, False, pk['b_g'], lay._act, None,
                  x2=L.view(sh['t']), w2_packed=sp['w2'], precision=prec, stream=st)

    # -- reverse ---------------------------------------------------------------------------------------------------
    def _pack_bwd(self):
        if self._bw is not None:
            return self._bw
        lay, lib, prec = self.lay, self.ex.lib, self.bprec
        f = lay.ffc
        sl, _ = F._bn_fold(lay.bn_l)
        sg, _ = F._bn_fold(lay.bn_g)
        st = f.convg2g
        s1, _ = F._bn_fold(st.conv1[1])
        sfu, _ = F._bn_fold(st.fu.bn)
        bw = {}
        # d x_l <- [g_l | g_g] through convl2l / convl2g (one conv over the 512 gradient channels)
        wd_l = torch.cat([_tflip(f.convl2l.weight, sl), _tflip(f.convl2g.weight, sg)], dim=1)
        bw['wd_l'] = lib.pack_conv_weight(wd_l, None, precision=prec)
        # round 4: the same dgrad without its padded plane -- interior as Winograd F(2x2, 3x3) with zero padding, frame by lama_dgrad_ring_fwd
        # (the planes decide per call whether the Winograd entry takes them: _FFCLayerTape.backward)
        if self.ex.winograd and prec in (L.PREC_BF16X3, L.PREC_F16X3) and wd_l.shape[0] % 128 == 0 and wd_l.shape[1] % 32 == 0:
            try:
                bw['wd_l_wino'] = lib.pack_winograd_weight(wd_l, None, prec)
                bw['wr_l'] = lib.dgrad_ring_weight(wd_l).to(wd_l.device)
            except LamaError as e:
                if e.code != L.ERR_UNSUPPORTED:
                    raise
        bw['wd_g2l'] = lib.pack_conv_weight(_tflip(f.convg2l.weight, sl), None, precision=prec)          # d x_g <- g_l
        bw['wd_2'] = lib.pack_conv_weight(_tflip(st.conv2.weight, sg), None, precision=prec)               # d t <- g_g
        bw['wd_fu'] = lib.pack_conv_weight(_tflip(st.fu.conv_layer.weight, sfu), None, precision=prec)     # spectral 1x1, transposed
        bw['wd_1'] = lib.pack_conv_weight(_tflip(st.conv1[0].weight, s1), None, precision=prec)            # d x_g <- d x1
        self._bw = bw
        return bw

    def backward(self, gm: torch.Tensor, tape: dict, sh: dict, g_src: Optional[torch.Tensor], gm_src: Optional[torch.Tensor] = None,
                 identity: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None):
        """gm [B,512,H,W] = d loss / d (layer output) ALREADY multiplied by the layer's own activation derivative (the step upstream in
        the reverse order wrote it: round 4, lama_reflect_pad_bwd_fused) -> s = d loss / d (layer input) [+ identity];
        ``g_src`` <- s (may be ``identity`` itself), ``gm_src`` <- s * relu'(mask) for the layer whose taped output ``mask`` is."""
        lay, ex, prec = self.lay, self.ex, self.bprec
        f = lay.ffc
        bw = self._pack_bwd()
        B = gm.shape[0]
        st = ex.stream(gm)
        lib = ex.lib
        cl, cg, ocl, ocg = f.in_cl, f.in_cg, f.out_cl, f.out_cg
        v = lambda t, c0, n: None if t is None else L.view(t, c0, n)     # noqa: E731
        # local input
        done = False
        if 'wd_l_wino' in bw and sh.get('wino_b') is not None:
            # the frame kernel FIRST: it has bounds of its own (channels % 64, its LDS footprint, batch <= 65535) and refuses with
            # ERR_UNSUPPORTED before anything was launched -- then neither half of the pair runs and the direct conv + fold below takes the layer
            try:
                lib.dgrad_ring(L.view(gm), bw['wr_l'], cl, sh['ring_l'], B, st)
                lib.winograd_conv3x3(L.view(gm), bw['wd_l_wino'], L.view(sh['gi_l']), B, sh['wino_b'], None, L.ACT_NONE, None, precision=prec,
                                     stream=st, pad_mode=L.PAD_ZERO)
                done = True
            except LamaError as e:
                if e.code != L.ERR_UNSUPPORTED:
                    raise
                sh['wino_b'] = None
        if done:
            lib.reflect_pad_bwd_fused(L.view(sh['gi_l']), None, v(identity, 0, cl), 1, v(mask, 0, cl), L.ACT_RELU, v(g_src, 0, cl),
                                      v(gm_src, 0, cl), B, st, ring=sh['ring_l'])
        else:
            lib.conv2d(L.view(gm), bw['wd_l'], L.view(sh['gp_l']), B, 3, 1, 2, L.PAD_ZERO, False, None, L.ACT_NONE, precision=prec, stream=st)
            lib.reflect_pad_bwd_fused(L.view(sh['gp_l']), None, v(identity, 0, cl), 1, v(mask, 0, cl), L.ACT_RELU, v(g_src, 0, cl),
                                      v(gm_src, 0, cl), B, st)
        # spectral branch: t = x1 + fu(x1), out_g += conv2(t)
        lib.conv2d(L.view(gm, ocl, ocg), bw['wd_2'], L.view(sh['g_t']), B, 1, precision=prec, stream=st)
        # the two ReLU derivatives of the branch ride in the transforms where a kernel does that (256 x 256 planes: lama_*_masked_fwd, v108)
        fused = sh.get('fft_mask', True)
        if fused:
            try:
                lib.rfft2(L.view(sh['g_t']), L.view(sh['s3']), B, sh['fftws'], st, mask=L.view(tape['s2']))
            except LamaError as e:
                if e.code != L.ERR_UNSUPPORTED:
                    raise
                fused = sh['fft_mask'] = False
        if not fused:
            lib.rfft2(L.view(sh['g_t']), L.view(sh['s1']), B, sh['fftws'], st)
            lib.act_bwd(L.view(sh['s1']), L.view(tape['s2']), L.ACT_RELU, L.view(sh['s3']), B, st)
        lib.conv2d(L.view(sh['s3']), bw['wd_fu'], L.view(sh['s1']), B, 1, precision=prec, stream=st)
        if fused:
            lib.irfft2(L.view(sh['s1']), L.view(sh['g_t']), L.view(sh['g_x1']), B, sh['fftws'], st, mask=L.view(tape['x1']))
        else:
            lib.irfft2(L.view(sh['s1']), L.view(sh['g_t']), L.view(sh['g_x1']), B, sh['fftws'], st)          # + g_t: the identity path of t
            lib.act_bwd(L.view(sh['g_x1']), L.view(tape['x1']), L.ACT_RELU, L.view(sh['g_x1']), B, st)
        # global input: through convg2l (3x3) and through conv1 (1x1)
        lib.conv2d(L.view(gm, 0, ocl), bw['wd_g2l'], L.view(sh['gp_g']), B, 3, 1, 2, L.PAD_ZERO, False, None, L.ACT_NONE, precision=prec, stream=st)
        lib.conv2d(L.view(sh['g_x1']), bw['wd_1'], L.view(sh['g1']), B, 1, precision=prec, stream=st)
        lib.reflect_pad_bwd_fused(L.view(sh['gp_g']), L.view(sh['g1']), v(identity, cl, cg), 1, v(mask, cl, cg), L.ACT_RELU, v(g_src, cl, cg),
                                  v(gm_src, cl, cg), B, st)


class RearPass:
    """``generator.model[first:]`` (refinement.py:276-289 ``forward_rears``) as forward-with-tape + explicit reverse pass.

    ``forward(z)``: z = the (x_l | x_g) state [B, 512, h, w] after ``generator.model[:first]`` -> pred [B, 3, 8h, 8w].
    ``backward(g_pred)``: d loss / d pred -> d loss / d z (same layout as z).  Buffers are allocated once per input shape."""

    def __init__(self, generator: F.FFCResNetGenerator, first: int, bwd_precision: int = L.PREC_BF16X3):
        if bwd_precision == L.PREC_F16X3:
            raise LamaError('the reverse pass does not run on the fp16 split (gradients underflow its lo term): use bf16x3 or f32')
        if bwd_precision == L.PREC_F16 or generator.precision == L.PREC_F16:
            raise LamaError('refinement runs on fp32 activation tensors: set the generator to f16x3 / bf16x3 / f32 (the tape and the '
                            'gradients are fp32; LAMA_PREC_F16 is the plain predict path only)')
        self.gen, self.ex, self.bprec = generator, generator._exec, bwd_precision
        layers = list(generator.model)[first:]
        self.blocks: List[tuple] = []
        self.ups: List[tuple] = []
        self.head = None
        i, n = 0, len(layers)
        while i < n:
            lay = layers[i]
            if isinstance(lay, F.FFCResnetBlock):
                self.blocks.append((lay, _FFCLayerTape(lay.conv1, self.ex, bwd_precision), _FFCLayerTape(lay.conv2, self.ex, bwd_precision)))
            elif isinstance(lay, F.ConcatTupleLayer):
                pass
            elif isinstance(lay, F.ConvTranspose2dUp):
                if not (i + 2 < n and isinstance(layers[i + 1], F.BatchNorm2dEval) and isinstance(layers[i + 2], F.Activation)
                        and layers[i + 2].kind == 'relu'):
                    raise LamaError('backward: ConvTranspose2d must be followed by BatchNorm2d + ReLU (ffc.py:348-354)')
                self.ups.append((lay, layers[i + 1]))
                i += 2
            elif isinstance(lay, F.ReflectionPad2d) and i + 1 < n and isinstance(layers[i + 1], F.Conv2dOut):
                act = layers[i + 2] if i + 2 < n and isinstance(layers[i + 2], F.Activation) else None
                self.head = (lay.padding, layers[i + 1], F._ACT[act.kind] if act else L.ACT_NONE)
                i += 2 if act else 1
            else:
                raise LamaError(f'backward: no reverse pass for layer {type(lay).__name__}')
            i += 1
        if self.head is None or not self.blocks:
            raise LamaError('backward: generator.model[first:] must hold resnet blocks and the output head')
        self._plan = None
        self._bw_up: Dict[int, torch.Tensor] = {}
        self._bw_head = None
        self.debug: Optional[dict] = None      # tests: receives clones of the gradient at every stage boundary of backward()

    # ---------------------------------------------------------------------------------------------------------------
    def _build(self, z: torch.Tensor):
        B, Cn, H, W = z.shape
        dev = z.device
        f0 = self.blocks[0][0].conv1.ffc
        half = f0.convg2g.conv2.in_channels
        wf = W // 2 + 1
        e = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)    # noqa: E731
        nws = self.ex.lib.fft_workspace_bytes(B, half, H, W)
        lib = self.ex.lib
        nwino = 0
        if self.ex.winograd and lib.winograd_preferred(B, f0.out_cl, f0.in_cl + f0.in_cg, H, W, self.gen.precision):
            nwino = lib.winograd_workspace_bytes(B, f0.out_cl, H, W)
        sh = dict(s1=e(B, 2 * half, H, wf), s3=e(B, 2 * half, H, wf), t=e(B, half, H, W),
                  fftws=(e(nws // 4 + 1) if nws else None), wino=(e(nwino // 4) if nwino else None),
                  wino_b=None, gi_l=None, ring_l=None,
                  gp_l=e(B, f0.in_cl, H + 2, W + 2), gp_g=e(B, f0.in_cg, H + 2, W + 2), g_t=e(B, half, H, W), g_x1=e(B, half, H, W),
                  g1=e(B, f0.in_cg, H, W))
        if (self.ex.winograd and self.bprec in (L.PREC_BF16X3, L.PREC_F16X3) and f0.in_cl % 128 == 0
                and lib.winograd_preferred(B, f0.in_cl, f0.out_cl + f0.out_cg, H, W, self.bprec)):
            # the dgrad into x_l as Winograd interior + frame (its own workspace: gm of the layer is live while the forward one is not, but
            # the split-K factor differs with the channel counts)
            sh['wino_b'] = e(lib.winograd_workspace_bytes(B, f0.in_cl, H, W) // 4)
            sh['gi_l'] = e(B, f0.in_cl, H, W)
            sh['ring_l'] = e(lib.dgrad_ring_bytes(B, f0.in_cl, H, W) // 4)
        tapes = [dict(c1=t1.alloc(B, H, W, dev), c2=t2.alloc(B, H, W, dev)) for _, t1, t2 in self.blocks]
        state = [e(B, Cn, H, W) for _ in range(2)]                       # block outputs ping-pong (the reverse pass does not need them)
        gst = [e(B, Cn, H, W) for _ in range(3)]                         # g (in place), gm of the second / first layer of a block
        ups, h, w, c = [], H, W, Cn
        for up, bn in self.ups:
            h, w, c = 2 * h, 2 * w, up.out_channels
            ups.append(dict(y=e(B, c, h, w), g=e(B, c, h, w)))
        pad, conv, act = self.head
        pred = e(B, conv.out_channels, h, w)
        head = dict(pred=pred, g1=e(B, conv.out_channels, h, w), gp=e(B, conv.in_channels, h + 2 * pad, w + 2 * pad))
        self._plan = dict(key=(tuple(z.shape), str(dev)), sh=sh, tapes=tapes, state=state, gst=gst, ups=ups, head=head)

    def forward(self, z: torch.Tensor) -> torch.Tensor:
        self.ex.check(z)
        if z.dim() != 4 or not z.is_contiguous():
            raise LamaError('RearPass.forward: contiguous [B, C, h, w] state expected')
        if self._plan is None or self._plan['key'] != (tuple(z.shape), str(z.device)):
            self._build(z)
        p, ex = self._plan, self.ex
        lib = ex.lib
        B = z.shape[0]
        st = ex.stream(z)
        with ex.range_scope(z, self.gen.precision):
            cur = z
            for bi, (blk, t1, t2) in enumerate(self.blocks):
                tp = p['tapes'][bi]
                t1.forward(cur, tp['c1'], p['sh'])
                t2.forward(tp['c1']['out'], tp['c2'], p['sh'])
                nxt = p['state'][bi & 1]
                lib.add(L.view(tp['c2']['out']), L.view(cur), L.view(nxt), B, st)          # ffc.py:288 (kept apart: exact ReLU masks)
                cur = nxt
            for (up, bn), ub in zip(self.ups, p['ups']):
                up.run(cur, ub['y'], bn, L.ACT_RELU)
                cur = ub['y']
            pad, conv, act = self.head
            conv.run(cur, p['head']['pred'], pad, act)
        return p['head']['pred']

    # ---------------------------------------------------------------------------------------------------------------
    def _up_bwd_weight(self, k: int, up: F.ConvTranspose2dUp, bn) -> torch.Tensor:
        if k not in self._bw_up:
            scale, _ = F._bn_fold(bn)
            w = up.weight.detach().float() * scale.view(1, -1, 1, 1)          # [Cin, Cout, 3, 3]: conv weight [out = Cin, in = Cout]
            self._bw_up[k] = self.ex.lib.pack_conv_weight(w.contiguous(), None, stride=2, precision=self.bprec)
        return self._bw_up[k]

    def backward(self, g_pred: torch.Tensor) -> torch.Tensor:
        p, ex, prec = self._plan, self.ex, self.bprec
        if p is None:
            raise LamaError('RearPass.backward before forward')
        lib = ex.lib
        B = g_pred.shape[0]
        st = ex.stream(g_pred)
        pad, conv, act = self.head
        hd = p['head']
        if self._bw_head is None:
            self._bw_head = lib.pack_conv_weight(_tflip(conv.weight, None), None, precision=prec)
        lib.act_bwd(L.view(g_pred), L.view(hd['pred']), act, L.view(hd['g1']), B, st)
        k = conv.kernel_size[0]
        lib.conv2d(L.view(hd['g1']), self._bw_head, L.view(hd['gp']), B, k, 1, k - 1, L.PAD_ZERO, False, None, L.ACT_NONE, precision=prec, stream=st)
        g = p['ups'][-1]['g'] if self.ups else p['gst'][0]
        # the head's fold writes the gradient already multiplied by the ReLU mask of the last upsampling layer (one pass less over the
        # largest tensor of the network); with the debug taps on, the two steps stay apart
        masked = bool(self.ups) and self.debug is None
        if masked:
            lib.reflect_pad_bwd_fused(L.view(hd['gp']), None, None, pad, L.view(p['ups'][-1]['y']), L.ACT_RELU, None, L.view(g), B, st)
        else:
            lib.reflect_pad_bwd(L.view(hd['gp']), None, pad, L.view(g), B, st)
        if self.debug is not None:
            self.debug['head_in'] = g.clone()
        for ui in range(len(self.ups) - 1, -1, -1):
            up, bn = self.ups[ui]
            ub = p['ups'][ui]
            if not masked:
                lib.act_bwd(L.view(g), L.view(ub['y']), L.ACT_RELU, L.view(ub['g']), B, st)       # in place when g is ub['g']
            masked = False
            dst = p['ups'][ui - 1]['g'] if ui > 0 else p['gst'][0]
            lib.conv2d(L.view(ub['g']), self._up_bwd_weight(ui, up, bn), L.view(dst), B, 3, 2, 1, L.PAD_ZERO, False, None, L.ACT_NONE,
                       precision=prec, stream=st)
            g = dst
            if self.debug is not None:
                self.debug[f'up{ui}_in'] = g.clone()
        # g = d loss / d (state after the last block), kept in gst[0] and updated IN PLACE block by block (the identity path, ffc.py:288);
        # gm = g * relu'(output of the block's second layer): an act_bwd launch for the last block, after that the second output of the
        # fold that ends the reverse pass of the block downstream (M0 / M1: gm of the second layer, gm of the first layer)
        if g is not p['gst'][0]:
            raise LamaError('backward: the gradient of the block state must live in gst[0]')
        m0, m1 = p['gst'][1], p['gst'][2]
        nb = len(self.blocks)
        lib.act_bwd(L.view(g), L.view(p['tapes'][nb - 1]['c2']['out']), self.blocks[nb - 1][2].lay._act, L.view(m0), B, st)
        for bi in range(nb - 1, -1, -1):
            _, t1, t2 = self.blocks[bi]
            tp = p['tapes'][bi]
            if t1.lay._act != L.ACT_RELU or t2.lay._act != L.ACT_RELU:
                raise LamaError('backward: resnet-block layers with a ReLU activation expected (ffc.py:253-254)')
            t2.backward(m0, tp['c2'], p['sh'], None, m1, None, tp['c1']['out'])          # m1 = d / d (conv1 output) * relu'(conv1 output)
            up_mask = p['tapes'][bi - 1]['c2']['out'] if bi > 0 else None
            t1.backward(m1, tp['c1'], p['sh'], g, m0 if bi > 0 else None, g, up_mask)      # g += d / d (block input); m0 = g * relu'(...)
            if self.debug is not None:
                self.debug[f'block{bi}_in'] = g.clone()
        return g
