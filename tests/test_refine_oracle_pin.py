"""Pin oracle/refine_oracle.py to the REFERENCE's own ``saicinpainting/evaluation/refinement.py:86-174,228-314``.

BUILD CONTAINER ONLY (skipped where /root/reference is absent, e.g. on the GPU box).  The reference module is imported
UNMODIFIED.  What it cannot import in this image is stubbed, and every stub is bound to a helper that is itself pinned by
known-answer vectors (tests/test_refine_helpers_known_answers.py):

  * ``kornia.filters.gaussian_blur2d``, ``kornia.geometry.transform.resize``, ``kornia.morphology.erosion`` -> the restated helpers
    of oracle/refine_oracle.py; ``cv2.getStructuringElement`` -> ``ellipse_kernel``; ``easydict`` -> dict subclass (never executed);
  * a package stub for ``saicinpainting.evaluation`` (its ``__init__`` pulls the evaluators and their downloads) whose ``__path__``
    is the real directory, so ``evaluation/data.py`` and ``evaluation/utils.py`` are the reference's own files;
  * ``torch.device('cuda:N')`` -> cpu inside the reference module only (it hard-codes cuda devices, refinement.py:265,277).

The generator the reference loop runs is the reference's own ``FFCResNetGenerator`` (reference factory, state dict loaded with
strict=True); the oracle side runs ``oracle.refine_oracle.refine_predict`` on the same state dict.  Control flow, pyramid, masks,
loss, Adam and the final blend must agree BIT FOR BIT (both sides execute the same torch primitives in the same order).
"""
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'saicinpainting')), reason='needs /root/reference (build container)')

_STUBBED = ['kornia', 'kornia.filters', 'kornia.geometry', 'kornia.geometry.transform', 'kornia.morphology', 'cv2', 'easydict',
            'pytorch_lightning', 'saicinpainting.evaluation']


class _TorchProxy:
    """``torch`` as the reference module sees it: everything forwarded, ``device(...)`` always the CPU."""

    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def device(*a, **kw):
        return torch.device('cpu')


@pytest.fixture(scope='module')
def ref_modules():
    from oracle import refine_oracle as R
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k.split('.')[0] in ('kornia', 'cv2', 'easydict', 'pytorch_lightning', 'saicinpainting')}
    for k in saved:
        del sys.modules[k]

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for a, v in attrs.items():
            setattr(m, a, v)
        sys.modules[name] = m
        return m

    def blur(x, kernel_size, sigma):
        assert tuple(kernel_size) == (5, 5) and tuple(sigma) == (1.0, 1.0)
        return R.gaussian_blur2d(x, 5, 1.0)

    def resize(x, size, interpolation='bilinear', align_corners=False):
        assert interpolation == 'bilinear' and align_corners is False
        return F.interpolate(x, size=size, mode='bilinear', align_corners=False)

    def get_se(shape, ksize):
        assert shape == 2 and tuple(ksize) == (15, 15)
        return R.ellipse_kernel(15).numpy().astype(np.uint8)

    k = mod('kornia')
    k.filters = mod('kornia.filters', gaussian_blur2d=blur)
    k.geometry = mod('kornia.geometry')
    k.geometry.transform = mod('kornia.geometry.transform', resize=resize,
                               rotate=lambda *a, **kw: (_ for _ in ()).throw(RuntimeError('stub')))
    k.morphology = mod('kornia.morphology', erosion=lambda x, kern: R.erosion(x, kern))
    mod('cv2', getStructuringElement=get_se, MORPH_ELLIPSE=2, INTER_AREA=3)
    mod('easydict', EasyDict=type('EasyDict', (dict,), {}))
    mod('pytorch_lightning', seed_everything=lambda *a, **kw: None)
    sys.path.insert(0, REF)
    try:
        import saicinpainting                                                        # the reference's (empty) top-level package
        ev = mod('saicinpainting.evaluation')
        ev.__path__ = [os.path.join(REF, 'saicinpainting', 'evaluation')]
        saicinpainting.evaluation = ev
        from saicinpainting.evaluation import refinement as ref_refinement           # UNMODIFIED reference file
        from saicinpainting.training.modules import make_generator
        assert os.path.realpath(ref_refinement.__file__).startswith(REF)
        ref_refinement.torch = _TorchProxy()
        ref_refinement.tqdm = lambda it, **kw: _Bar(it)
        yield ref_refinement, make_generator
    finally:
        sys.path.remove(REF)
        for k_ in list(sys.modules):
            if k_.split('.')[0] in ('kornia', 'cv2', 'easydict', 'pytorch_lightning', 'saicinpainting'):
                del sys.modules[k_]
        sys.modules.update({k_: v for k_, v in saved.items() if v is not None})


class _Bar:
    def __init__(self, it):
        self.it = it

    def __iter__(self):
        return iter(self.it)

    def set_description(self, *_a, **_kw):
        pass


class _Inpainter:
    """What refine_predict touches of DefaultInpaintingTrainingModule (refinement.py:262-289)."""
    training = False
    add_noise_kwargs = None
    concat_mask = True

    def __init__(self, generator):
        self.generator = generator


def _case(h, w, seed, gray_stroke=False):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(1, 3, max(h // 16, 2), max(w // 16, 2), generator=g)
    image = F.interpolate(low, size=(h, w), mode='bilinear', align_corners=False)
    image = (image + 0.05 * torch.rand(1, 3, h, w, generator=g)).clamp(0, 1)
    mask = torch.zeros(1, 1, h, w)
    mask[:, :, h // 5:h // 2, w // 4:(2 * w) // 3] = 1.0
    mask[:, :, (3 * h) // 5:(3 * h) // 5 + max(h // 40, 2), w // 8:(7 * w) // 8] = 0.5 if gray_stroke else 1.0
    return image, mask


# (h, w, kwargs of refine_predict): three scales; one scale boundary; a px-budget resize with a gray mask stroke
CASES = [
    (250, 300, dict(min_side=64, max_scales=3, px_budget=1800000), False),
    (256, 256, dict(min_side=128, max_scales=3, px_budget=1800000), False),
    (300, 420, dict(min_side=100, max_scales=3, px_budget=60000), True),
]


@pytest.mark.parametrize('h,w,kw,gray', CASES)
def test_refine_oracle_equals_reference_refinement(ref_modules, h, w, kw, gray):
    ref_refinement, make_generator = ref_modules
    from oracle import lama_oracle as O
    from oracle import refine_oracle as R
    cfg = O.small_config(ngf=8, n_blocks=3)
    sd = O.make_synthetic_state_dict(cfg, seed=11, calib_hw=32)
    gen = make_generator(None, kind='ffc_resnet', **cfg)
    res = gen.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    gen.eval()
    for p in gen.parameters():
        p.requires_grad_(False)                                                       # = LightningModule.freeze() (bin/predict.py:60)
    image, mask = _case(h, w, seed=h * 1000 + w, gray_stroke=gray)
    ph, pw = O.ceil_modulo(h, 8), O.ceil_modulo(w, 8)
    image_p = F.pad(image, (0, pw - w, 0, ph - h), mode='reflect')                   # the dataset hands padded tensors + unpad_to_size
    mask_p = F.pad(mask, (0, pw - w, 0, ph - h), mode='reflect')
    n_iters, lr = 6, 0.002
    batch = {'image': image_p.clone(), 'mask': mask_p.clone(), 'unpad_to_size': [torch.tensor([h]), torch.tensor([w])]}
    want = ref_refinement.refine_predict(batch, _Inpainter(gen), gpu_ids='0,', modulo=8, n_iters=n_iters, lr=lr, **kw)
    trace = []
    got = R.refine_predict(image_p.clone(), mask_p.clone(), (h, w), sd, cfg, modulo=8, n_iters=n_iters, lr=lr, trace=trace, **kw)
    assert len(trace) >= 2                                                            # the refinement loop ran, not only the plain forward
    assert all(len(t.get('loss', [])) == n_iters for t in trace[1:])
    assert got.shape == want.shape
    assert torch.equal(got, want), float((got - want).abs().max())
