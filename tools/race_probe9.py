#!/usr/bin/env python3
"""Fifth round: next to the MFMA spinner (lama_debug_hog mode 3), which class of work goes wrong -- VALU, LDS + barrier, sincospif,
butterflies, fast transcendentals -- and how do the wrong FFT outputs look (which planes, how many elements, how far off)?
    LAMA_HIP_LIB=lama_amd/lib/liblama_hip_prof.so python tools/race_probe9.py <iterations>"""
import ctypes as C
import sys
import torch
sys.path.insert(0, '.')
from lama_amd import _lib as L

lib = L.get_lib()
hog = lib._l.lama_debug_hog
hog.restype, hog.argtypes = C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
probe = lib._l.lama_debug_probe
probe.restype, probe.argtypes = C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
G = 1536
torch.manual_seed(0)
src = torch.randn(G * 4096, device='cuda')
out = torch.empty_like(src)
hout = torch.empty(256 * 512, device='cuda')
main = torch.cuda.current_stream(); side = torch.cuda.Stream()
for name, mode in (('valu', 0), ('lds_transpose+barrier', 1), ('lds_shift+barrier', 5), ('sincospif', 2), ('radix8', 3), ('fast_trans', 4), ('v_pk_add_f32', 6), ('v_pk_mul_f32', 7), ('v_pk_fma_f32', 8), ('v_pk_add_f32 op_sel', 9), ('v_pk_mul_f32 sgpr op_sel_hi', 10)):
    lib.check(probe(main.cuda_stream, G, mode, src.data_ptr(), out.data_ptr()), 'probe')
    torch.cuda.synchronize()
    ref = out.clone()
    for hogmode, hname in ((-1, 'alone'), (0, 'valu_hog'), (3, 'mfma_hog')):
        bad, nel, mx = 0, 0, 0.0
        for it in range(N):
            out.fill_(7.0)
            side.wait_stream(main)
            if hogmode >= 0:
                lib.check(hog(main.cuda_stream, 256, hogmode, 1200, hout.data_ptr()), 'hog')
            lib.check(probe(side.cuda_stream, G, mode, src.data_ptr(), out.data_ptr()), 'probe')
            main.wait_stream(side)
            torch.cuda.synchronize()
            d = out != ref
            if bool(d.any()):
                bad += 1
                nel = max(nel, int(d.sum()))
                mx = max(mx, float((out - ref)[d].abs().max()))
        print(f'== probe {name} next to {hname}: wrong {bad} / {N}, up to {nel} elements, max |diff| {mx:.3e}', flush=True)

# the FFT kernels next to the MFMA spinner: what do the wrong outputs look like?
B, H, W = 8, 64, 64
wf = W // 2 + 1
x1 = torch.randn(B, 192, H, W, device='cuda')
s1 = torch.empty(B, 384, H, wf, device='cuda')
lib.rfft2(L.view(x1), L.view(s1), B, None, main.cuda_stream); torch.cuda.synchronize()
r1 = s1.clone()
for it in range(3):
    s1.fill_(7.0)
    side.wait_stream(main)
    lib.check(hog(main.cuda_stream, 256, 3, 1200, hout.data_ptr()), 'hog')
    lib.rfft2(L.view(x1), L.view(s1), B, None, side.cuda_stream)
    main.wait_stream(side)
    torch.cuda.synchronize()
    d = s1 != r1
    planes = d.view(B * 192, 2, H, wf).any(dim=3).any(dim=2).any(dim=1).nonzero().flatten()
    diff = (s1 - r1)[d].abs()
    i0 = d.nonzero()[0].tolist() if bool(d.any()) else None
    print(f'== rfft2 next to mfma_hog: {int(d.sum())} wrong elements in {planes.numel()} of {B * 192} planes; |diff| median {float(diff.median()) if diff.numel() else 0:.3e} '
          f'max {float(diff.max()) if diff.numel() else 0:.3e}; first planes {planes[:12].tolist()}; first wrong index {i0}: got {float(s1[tuple(i0)]) if i0 else None} ref {float(r1[tuple(i0)]) if i0 else None}',
          flush=True)
