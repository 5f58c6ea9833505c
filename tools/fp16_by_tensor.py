#!/usr/bin/env python3
"""CPU experiment (ORACLE only, test infrastructure): which STORED tensors of the LAMA_PREC_F16 path (BASELINE configs[2]: fp16 activations in HBM)
carry its end-to-end error (Next #7).  The fp32 oracle is run with roundings to fp16 inserted exactly where the HIP path stores an fp16
tensor (weights stay fp32: the path keeps hi + lo weight parts); every group below can be switched to fp32 storage:

    front   stem / down1 / down2 outputs (the input of down3 .. is read as fp16)
    blockin the READ of the fp32 residual stream by conv1 of every FFCResnetBlock (one matrix-core operand per activation: rounded while staged)
    mid     the (x_l | x_g) tensor between conv1 and conv2 of every FFCResnetBlock            (the residual stream itself is fp32 already)
    x1      SpectralTransform.conv1's output (input of the FourierUnit and of the residual add x1 + fu(x1))
    spec    the two spectra of the FourierUnit (rfft2 output / spectral 1x1 output)
    t       x1 + fu(x1), the input of SpectralTransform.conv2
    up      the outputs of the three ConvTranspose2d + BN + ReLU (up3 = the head's input); up1 / up2 / up3 select one of them

    python tools/fp16_by_tensor.py [res=1024] [batch=1]
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lama_oracle as O
torch.set_num_threads(int(os.environ.get('OMP_NUM_THREADS', '8')))
def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    bn = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cfg = O.BIG_LAMA
    sd = O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64)
    batch = O.make_synthetic_batch(bn, res, res, seed=12)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    UP = ['up1', 'up2', 'up3']
    BODY = ['front', 'blockin', 'mid', 'x1', 'spec', 't']
    with torch.no_grad():
        ref = O.generator_forward(x, sd, cfg)
        assert float((ref - O.generator_forward_fp16_storage(x, sd, cfg, half=())).abs().max()) == 0.0
        cases = [('all fp16 (the round-3 LAMA_PREC_F16 layout)', BODY + UP), ('tail fp32 (the layout since round 4)', BODY)] + \
                [(f'only {g}', [g]) for g in BODY + UP] + \
                [('all but up3 (head input fp32)', BODY + ['up1', 'up2']), ('all but up2, up3', BODY + ['up1']),
                 ('all but x1, t, spec', ['front', 'mid'] + UP), ('front + up only', ['front'] + UP)]
        for name, groups in cases:
            d = (O.generator_forward_fp16_storage(x, sd, cfg, half=groups) - ref).abs()
            print(f'{bn} x {res}^2  fp16 storage: {name:45s} max-abs {float(d.max()):.2e}  mean-abs {float(d.mean()):.2e}', flush=True)


if __name__ == '__main__':
    main()
