"""rfft2 / irfft2 through the C ABI against torch.fft at every plane length 2 .. 256 plus degenerate / large-prime pairs (432 planes; the test form is
tests/test_kernels_gpu.py::test_rfft2_irfft2_every_length_up_to_256): python tools/fft_sweep.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_kernels_emu import _spec_ref, _inv_ref
from lama_amd import _lib as L
lib = L.get_lib()
DEV='cuda'
st = torch.cuda.current_stream().cuda_stream
def run(h,w,B=1,Cn=2,seed=None):
    g = torch.Generator().manual_seed(h if seed is None else seed)
    x = torch.randn(B, Cn, h, w, generator=g)
    spec = torch.zeros(B, 2 * Cn, h, w // 2 + 1, device=DEV)
    ws = torch.zeros(max(lib.fft_workspace_bytes(B, Cn, h, w), 4) // 4, device=DEV)
    xd=x.to(DEV)
    lib.rfft2(L.view(xd), L.view(spec), B, ws, stream=st)
    e1 = float((spec.cpu() - _spec_ref(x)).abs().max())
    spec2 = torch.relu(torch.randn(B, 2 * Cn, h, w // 2 + 1, generator=g))
    resid = torch.randn(B, Cn, h, w, generator=g)
    y = torch.zeros(B, Cn, h, w, device=DEV)
    s2d, rd = spec2.to(DEV), resid.to(DEV)
    lib.irfft2(L.view(s2d), L.view(rd), L.view(y), B, ws, stream=st)
    e2 = float((y.cpu() - (resid + _inv_ref(spec2, h, w))).abs().max())
    return e1,e2
bad=[]
pairs=[(h, 2 + (h * 97) % 255) for h in range(2,257)]
pairs += [(h,w) for h in (1,2,3,4,5,6,7,8) for w in (2,3,4,8,12,16,28,49,64,98,100,128,196,200,250,256)]
pairs += [(w,h) for h in (2,3,4,5,6,7,8) for w in (28,49,98,196,200,250,256)]
for h,w in pairs:
    try:
        e1,e2=run(h,w)
    except Exception as ex:
        print('EXC',h,w,repr(ex)[:200]); continue
    if e1>2e-4 or e2>2e-4:
        bad.append((h,w,e1,e2)); print('BAD',h,w,e1,e2,flush=True)
print('n pairs',len(pairs),'bad',len(bad))
# repeatability of a bad one
for h,w,_,_ in bad[:5]:
    print('repeat',h,w,[run(h,w) for _ in range(3)])
