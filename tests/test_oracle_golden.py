"""Pin the CPU oracle (oracle/lama_oracle.py) against vectors produced by the reference's own
ffc.py classes (tests/golden/make_golden.py).  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import lama_oracle as O

TOL = 2e-5  # fp32, same torch primitives; differences are summation-order only


def _npz(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.fixture(scope='module')
def small(golden_dir):
    cfg = O.small_config(ngf=8, n_blocks=2)
    sd = O.make_synthetic_state_dict(cfg, seed=7, calib_hw=32)
    return cfg, sd, _npz(golden_dir, 'small_gen.npz')


def test_small_generator_weights_regenerate(small):
    cfg, sd, g = small
    cs = sum(float(v.double().sum()) for v in sd.values() if v.is_floating_point())
    assert abs(cs - float(g['sd_checksum'][0])) < 1e-3 * max(1.0, abs(cs))


@pytest.mark.parametrize('case', ['a', 'b', 'c'])
def test_small_generator_matches_reference(small, case):
    cfg, sd, g = small
    x = torch.from_numpy(g[f'{case}_x'])
    taps = {}
    with torch.no_grad():
        y = O.generator_forward(x, sd, cfg, taps=taps)
    assert np.abs(y.numpy() - g[f'{case}_y']).max() < TOL
    for i in (4, 5, 6):
        assert np.abs(taps[i][0].numpy() - g[f'{case}_tap{i}_l']).max() < 1e-4
        assert np.abs(taps[i][1].numpy() - g[f'{case}_tap{i}_g']).max() < 1e-4
    for i in (7, 10, 13):
        assert np.abs(taps[i].numpy() - g[f'{case}_tap{i}']).max() < 1e-4


@pytest.mark.parametrize('tag', ['e', 'o', 'p', 'q'])
def test_fourier_unit_matches_reference_and_f64_dft(golden_dir, tag):
    g = _npz(golden_dir, 'ffc_units.npz')
    sd = {k[len(f'fu_{tag}_sd_'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f'fu_{tag}_sd_')}
    sd = {'fu.' + k: v for k, v in sd.items()}
    x = torch.from_numpy(g[f'fu_{tag}_x'])
    with torch.no_grad():
        y = O.fourier_unit(x, sd, 'fu')
    assert np.abs(y.numpy() - g[f'fu_{tag}_y']).max() < TOL
    w = sd['fu.conv_layer.weight'][:, :, 0, 0].numpy()
    y64 = O.fourier_unit_f64_dft(x.numpy(), w, sd['fu.bn.weight'].numpy(), sd['fu.bn.bias'].numpy(),
                                 sd['fu.bn.running_mean'].numpy(), sd['fu.bn.running_var'].numpy())
    assert np.abs(y64 - g[f'fu_{tag}_y']).max() < TOL


def test_ffc_block_units_match_reference(golden_dir):
    g = _npz(golden_dir, 'ffc_units.npz')
    sd = {'b.' + k[len('blk_sd_'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('blk_sd_')}
    xl, xg = torch.from_numpy(g['blk_xl']), torch.from_numpy(g['blk_xg'])
    spec = dict(ratio_gin=0.75, ratio_gout=0.75)
    with torch.no_grad():
        yl, yg = O.ffc_resnet_block(xl, xg, sd, 'b', spec)
        l1, g1 = O.ffc_bn_act(xl, xg, sd, 'b.conv1', dict(k=3, stride=1, pad=1, **spec))
        st = O.spectral_transform(xg, sd, 'b.conv1.ffc.convg2g')
    for a, key in ((yl, 'blk_yl'), (yg, 'blk_yg'), (l1, 'blk_c1_l'), (g1, 'blk_c1_g'), (st, 'blk_st')):
        assert np.abs(a.numpy() - g[key]).max() < TOL


def test_biglama_shape_matches_reference(golden_dir):
    g = _npz(golden_dir, 'biglama_256.npz')
    cfg = O.BIG_LAMA
    spec = O.state_dict_spec(cfg)
    n_tensors = sum(5 if r == 'bn' else 1 for _, _, r in spec)
    assert n_tensors == 989
    sd = O.make_synthetic_state_dict(cfg, seed=0, calib_hw=64)
    assert len(sd) == 989
    nparams = sum(v.numel() for k, v in sd.items() if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked')))
    assert nparams == 50975875
    batch = O.make_synthetic_batch(1, 256, 256, seed=1234)
    x = torch.cat([batch['image'] * (1 - batch['mask']), batch['mask']], 1)
    taps = {}
    with torch.no_grad():
        y = O.generator_forward(x, sd, cfg, taps=taps)
    assert np.abs(y[:, :, ::8, ::8].numpy() - g['y_sample']).max() < 1e-4
    for i in (4, 5, 13, 22):
        assert np.abs(taps[i][0][:, ::16, ::16, ::16].numpy() - g[f'tap{i}_sample']).max() < 1e-3
        assert np.abs(taps[i][1][:, ::32, ::8, ::8].numpy() - g[f'tap{i}_g_sample']).max() < 1e-3
    # the fixture is a meaningful test signal: output spans (0,1)
    assert y.min() < 0.05 and y.max() > 0.95 and y.std() > 0.05


def test_predict_glue_matches_reference(golden_dir):
    g = _npz(golden_dir, 'predict_glue.npz')
    cfg = O.small_config(ngf=8, n_blocks=2)
    sd = O.make_synthetic_state_dict(cfg, seed=7, calib_hw=32)
    sd = {'generator.' + k: v for k, v in sd.items()}
    cur, u8 = O.predict_one(g['image'], g['mask'], sd, cfg)
    assert cur.shape == (37, 50, 3)
    assert np.abs(cur - g['inpainted']).max() < TOL
    assert np.abs(u8.astype(int) - g['u8'].astype(int)).max() <= 1


def test_ffc_units_at_biglama_channel_counts_match_reference(golden_dir):
    """VERDICT r4 Next #6: FourierUnit / SpectralTransform / FFC_BN_ACT / FFCResnetBlock at 512 = (128 | 384) channels on 64 x 64 planes
    (the bottleneck of BASELINE configs[1]) from the reference's own classes (tests/golden/make_golden_units512.py) against the oracle."""
    sys.path.insert(0, golden_dir)
    from make_golden_units512 import BLOCK, block_inputs, sample
    g = _npz(golden_dir, 'ffc_block512.npz')
    sd = O.make_synthetic_state_dict(O.BIG_LAMA, seed=0, calib_hw=64)
    bsd = {k: v for k, v in sd.items() if k.startswith(BLOCK + '.')}
    assert abs(sum(float(v.double().sum()) for v in bsd.values() if v.is_floating_point()) - float(g['sd_checksum'][0])) < 1e-6 * abs(float(g['sd_checksum'][0]))
    xl, xg = block_inputs()
    assert abs(float(xl.double().sum()) - g['x_checksum'][0]) < 1e-3 and abs(float(xg.double().sum()) - g['x_checksum'][1]) < 1e-3
    spec = dict(ratio_gin=0.75, ratio_gout=0.75)
    with torch.no_grad():
        yl, yg = O.ffc_resnet_block(xl, xg, sd, BLOCK, spec)
        l1, g1 = O.ffc_bn_act(xl, xg, sd, BLOCK + '.conv1', dict(k=3, stride=1, pad=1, **spec))
        st = O.spectral_transform(xg, sd, BLOCK + '.conv1.ffc.convg2g')
        fu = O.fourier_unit(xg[:, :192].contiguous(), sd, BLOCK + '.conv1.ffc.convg2g.fu')
    for a, key in ((yl, 'yl'), (yg, 'yg'), (l1, 'c1_l'), (g1, 'c1_g'), (st, 'st'), (fu, 'fu')):
        ref = g[key + '_sample']
        assert np.abs(sample(a) - ref).max() < 2e-5 * max(1.0, float(g[key + '_stat'][2])), key
