"""Test helper: torch autograd through the ORACLE's rear layers with the ReLU masks of a RearPass tape.

ReLU' is discontinuous: two valid fp32 evaluations of the same 18-block network put a handful of pre-activations on different sides of 0 and
the gradients differ by ~5e-3 relative L2 for that reason alone (the oracle against itself with its input perturbed by 1e-7).  A strict check
of the explicit reverse pass therefore fixes the masks: every ``torch.relu`` of ``oracle.lama_oracle`` is replaced -- in the oracle's own call
order -- by ``x * mask`` with the mask the HIP forward recorded (tape output > 0).  Forward values change only at the flipped elements (by
their rounding-size pre-activation); the gradient is exactly the adjoint of the linearised network the HIP reverse pass claims to apply."""
import contextlib

import torch

from oracle import lama_oracle as O


def tape_masks(rear):
    """ReLU masks of a RearPass tape (after ``rear.forward``) in the order ``oracle.lama_oracle.run_layers`` calls ``torch.relu`` on
    ``generator.model[first:]``: per FFC layer x1, the post-ReLU spectrum, out_l, out_g (ffc.py:145,101,253,254); then one per upsampling layer."""
    p = rear._plan
    masks = []
    for (blk, t1, t2), tp in zip(rear.blocks, p['tapes']):
        for t, key in ((t1, 'c1'), (t2, 'c2')):
            ocl = t.lay.ffc.out_cl
            out = tp[key]['out']
            masks += [tp[key]['x1'] > 0, tp[key]['s2'] > 0, out[:, :ocl] > 0, out[:, ocl:] > 0]
    masks += [ub['y'] > 0 for ub in p['ups']]
    return [m.float().cpu() for m in masks]


@contextlib.contextmanager
def relu_masks(masks):
    """Inside: ``torch.relu(x)`` is ``x * masks.pop(0)`` (shape-checked); every mask must have been consumed on exit."""
    queue = list(masks)
    real = torch.relu

    def masked(x):
        m = queue.pop(0)
        assert m.shape == x.shape, (tuple(m.shape), tuple(x.shape))
        return x * m

    torch.relu = masked
    try:
        yield
        assert not queue, f'{len(queue)} masks left over'
    finally:
        torch.relu = real


def rear_gradient(z1, z2, sd, cfg, first, gw, masks):
    """(pred, d <gw, pred> / d (z1 | z2)) through oracle layers [first:] with the given ReLU masks."""
    a, b = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
    with relu_masks(masks):
        pred = O.run_layers((a, b), sd, cfg, first, None)
    (pred * gw).sum().backward()
    return pred.detach(), torch.cat([a.grad, b.grad], 1)
