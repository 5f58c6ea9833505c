"""big-lama forward against the fp32 oracle at a list of padded shapes (first call and the replay of the captured plan): python tools/shape_sweep.py [BxHxW ...]
(default: 24 shapes across the kernels' geometry decisions; round 6, fourth session: all <= 1.4e-4).  The subset with a short oracle pass is
tests/test_generator_gpu.py::test_biglama_shape_sweep."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lama_oracle as O
from lama_amd.modules import make_generator
from lama_amd import _lib as L
cfg = O.BIG_LAMA
sd = O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64)
gen = make_generator(None, kind='ffc_resnet', **cfg)
gen.load_state_dict(sd, strict=True)
gen = gen.cuda().set_precision(L.PREC_F16X3)
shapes = [(1,16,24),(1,32,32),(1,32,64),(1,64,64),(2,72,104),(3,136,136),(1,200,328),(1,264,392),(2,312,120),(1,400,408),(1,512,520),(1,520,512),
          (1,1016,1032),(1,24,512),(1,512,24),(5,96,160),(1,1000,1504),(1,640,808),(1,808,640),(7,64,72),(1,88,1048),(9,40,48),(1,16,16),(2,1024,16)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for B,H,W in shapes:
    batch = O.make_synthetic_batch(B, H, W, seed=H*7+W)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    t=time.time()
    try:
        with torch.no_grad():
            ref = O.generator_forward(x, sd, cfg)
    except Exception as ex:
        print(B,H,W,'ORACLE EXC',repr(ex)[:150],flush=True); continue
    t1=time.time()-t
    try:
        y = gen(x.cuda()).cpu()
        y2 = gen(x.cuda()).cpu()     # second call: the captured plan
    except Exception as ex:
        print(B,H,W,'GPU EXC',repr(ex)[:300],flush=True); gen._plans.clear(); continue
    e=float((y-ref).abs().max()); e2=float((y2-ref).abs().max())
    print(f'{B}x{H}x{W}: max-abs {e:.3e} (replay {e2:.3e})  oracle {t1:.1f}s  {"BAD" if max(e,e2)>2e-4 else "ok"}',flush=True)
    gen._plans.clear()
