import sys; sys.path.insert(0,'.')
import torch
from lama_amd import _lib as L, ffc as F
from lama_amd.backward import RearPass
from lama_amd.modules import make_generator
from oracle import lama_oracle as O, refine_oracle as R
dev = sys.argv[1] if len(sys.argv) > 1 else 'emu'
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = dict(O.BIG_LAMA); cfg['n_blocks'] = 2
sd = O.make_synthetic_state_dict(cfg, seed=2, calib_hw=64)
gen = make_generator(None, kind='ffc_resnet', **cfg)
gen.load_state_dict(sd, strict=True)
if dev == 'emu':
    from tests.emu import emu_lib
    gen.set_exec(F._Exec(emu_lib())); D = 'cpu'
else:
    gen.cuda(); D = 'cuda'
gen.set_precision(L.PREC_F32)
fri = R.first_resblock_index(cfg)
batch = O.make_synthetic_batch(1, HW, HW, seed=6)
x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
with torch.no_grad():
    z1, z2 = O.run_layers(x, sd, cfg, 0, fri)
z1r, z2r = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
taps = {}
pred_ref = O.run_layers((z1r, z2r), sd, cfg, fri, None, taps=taps)
for k, v in taps.items():
    for t in (v if isinstance(v, tuple) else (v,)):
        if torch.is_tensor(t) and t.requires_grad: t.retain_grad()
gw = torch.randn(pred_ref.shape, generator=torch.Generator().manual_seed(7)) / pred_ref.numel()
(pred_ref * gw).sum().backward()
rear = RearPass(gen, fri, bwd_precision=L.PREC_F32)
rear.debug = {}
pred = rear.forward(torch.cat([z1, z2], 1).contiguous().to(D))
print('pred err', float((pred.cpu() - pred_ref.detach()).abs().max()))
g = rear.backward(gw.contiguous().to(D)).cpu()
plan = O.layer_plan(cfg)
def tapgrad(i):
    v = taps[i]
    return torch.cat([t.grad for t in v], 1) if isinstance(v, tuple) else v.grad
def rel(a, b):
    d = (a.cpu() - b).abs()
    return (f'max-rel {float(d.max()) / float(b.abs().max()):.2e}  l2-rel {float(d.norm() / b.norm()):.2e}  '
            f'elements off by > 1e-3 of max: {int((d > 1e-3 * b.abs().max()).sum())} of {b.numel()}')
kinds = [(i, plan[i]['kind']) for i in range(fri, len(plan))]
print(kinds)
# head input = output of last relu (index before reflpad)
idx_relu = [i for i, k in kinds if k == 'relu']
idx_concat = [i for i, k in kinds if k == 'concat'][0]
print('head_in', rel(rear.debug['head_in'], tapgrad(idx_relu[-1])))
ups_in = [idx_concat] + idx_relu[:-1]
for ui in (2, 1, 0):
    print(f'up{ui}_in', rel(rear.debug[f'up{ui}_in'], tapgrad(ups_in[ui])))
blocks = [i for i, k in kinds if k == 'resblock']
print('block1_in', rel(rear.debug['block1_in'], tapgrad(blocks[0])))
gref = torch.cat([z1r.grad, z2r.grad], 1)
print('block0_in (final)', rel(g, gref))
