#!/usr/bin/env python3
"""Fourth round (see race_probe5-7.py): WHICH co-resident activity breaks the FFT kernels?  A synthetic load (lama_debug_hog of the
profiling build: one 512-thread workgroup per CU) is started on the main stream, then rfft2(x1) and irfft2(s2) run on the side
stream on inputs that were complete before the fork, and their outputs are compared with a serial run.
    LAMA_HIP_LIB=lama_amd/lib/liblama_hip_prof.so python tools/race_probe8.py <iterations>"""
import ctypes as C
import sys
import torch
sys.path.insert(0, '.')
from lama_amd import _lib as L

lib = L.get_lib()
hog = lib._l.lama_debug_hog
hog.restype, hog.argtypes = C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
B, H, W = 8, 64, 64
wf = W // 2 + 1
torch.manual_seed(0)
x1 = torch.randn(B, 192, H, W, device='cuda')
s2 = torch.relu(torch.randn(B, 384, H, wf, device='cuda'))
s1 = torch.empty(B, 384, H, wf, device='cuda'); t = torch.empty(B, 192, H, W, device='cuda')
hout = torch.empty(256 * 512, device='cuda')
xa = torch.randn(B, 512, H, W, device='cuda'); ya = torch.empty(B, 128, H, W, device='cuda')
wa = lib.pack_conv_weight(torch.randn(128, 512, 3, 3, device='cuda') * 0.02, None, precision=L.PREC_F16X3)
main = torch.cuda.current_stream(); side = torch.cuda.Stream()


def ffts(s):
    lib.rfft2(L.view(x1), L.view(s1), B, None, s)
    lib.irfft2(L.view(s2), L.view(x1), L.view(t), B, None, s)


ffts(main.cuda_stream); torch.cuda.synchronize()
r1, rt = s1.clone(), t.clone()
# calibrate the hog length to ~150 us
for name, mode, iters in (('none', -1, 0), ('valu_spin', 0, 1500), ('lds_inbounds', 1, 500), ('barrier_loop', 2, 20000), ('mfma_loop', 3, 1200),
                          ('lds+barrier', 4, 500), ('lds_out_of_bounds', 5, 500), ('conv3x3_wreg', 100, 0)):
    bad1 = badt = 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(N):
        s1.fill_(7.0); t.fill_(7.0)
        side.wait_stream(main)
        if it == N - 1: ev0.record(main)
        if mode == 100:
            lib.conv2d(L.view(xa), wa, L.view(ya), B, 3, 1, 1, L.PAD_REFLECT, False, None, L.ACT_RELU, precision=L.PREC_F16X3, stream=main.cuda_stream)
        elif mode >= 0:
            lib.check(hog(main.cuda_stream, 256, mode, iters, hout.data_ptr()), 'hog')
        if it == N - 1: ev1.record(main)
        ffts(side.cuda_stream)
        main.wait_stream(side)
        torch.cuda.synchronize()
        bad1 += int(not torch.equal(s1, r1)); badt += int(not torch.equal(t, rt))
    print(f'== co-resident {name}: rfft2 wrong {bad1} / {N}, irfft2 wrong {badt} / {N}  (load kernel {ev0.elapsed_time(ev1) * 1e3:.0f} us)', flush=True)
