"""CRNN mask estimator against outputs of the REFERENCE'S OWN model / prepare_data / reshape_mask code
(tests/golden/crnn_ref.npz, produced by tests/golden/make_golden_crnn.py).  Runs on CPU (PyTorch)."""
import os

import numpy as np
import pytest
import torch

from disco_amd.dnn.crnn import CRNN, build_crnn, get_z_for_mask


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'crnn_ref.npz'))


def _model(gold, tag, n_ch):
    sd = {k[len(tag) + 4:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(f'{tag}_sd_')}
    return build_crnn(n_ch=n_ch, state_dict=sd)          # strict load: every reference key must exist here and vice versa


@pytest.mark.parametrize('tag,n_ch', [('sc', 1), ('mc', 4)])
def test_state_dict_compatible_and_window_forward(gold, tag, n_ch):
    model = _model(gold, tag, n_ch)
    x_in = torch.from_numpy(gold[f'{tag}_x_in'])
    with torch.no_grad():
        m = model(x_in).numpy()
    assert m.shape == gold[f'{tag}_m_stack'].shape == (5, 15, 257)
    assert np.abs(m - gold[f'{tag}_m_stack']).max() < 2e-6


@pytest.mark.parametrize('tag,n_ch', [('sc', 1), ('mc', 4)])
def test_sequence_path_equals_reference_mask(gold, tag, n_ch):
    """predict_masks (one convolution pass over the whole sequence, truncated GRU) == reshape_mask(model(prepare_data))."""
    model = _model(gold, tag, n_ch)
    chans = [np.abs(gold[f'{tag}_Y'])]
    if n_ch > 1:
        chans += [np.abs(z) for z in gold[f'{tag}_Z']]
    mag = torch.from_numpy(np.stack(chans)[None].transpose(0, 1, 3, 2).copy())          # (1, n_ch, T, F)
    mask = model.predict_masks(mag).numpy()[0]                                           # (T, F)
    ref = gold[f'{tag}_mask']                                                            # (F, T)
    assert mask.shape == ref.T.shape
    assert np.abs(mask - ref.T).max() < 3e-6
    # chunking over the batch does not change anything
    mag2 = torch.cat([mag, mag * 0.5, mag * 2.0])
    a = model.predict_masks(mag2, chunk=2).numpy()
    b = model.predict_masks(mag2, chunk=64).numpy()
    assert np.array_equal(a[0], mask) or np.abs(a[0] - mask).max() < 1e-6
    assert np.abs(a - b).max() < 1e-6


def test_prepare_data_equivalence(gold):
    """The windows the reference feeds the network are slices of the clipped, zero-padded magnitude sequence."""
    Y = gold['sc_Y']
    x = np.clip(np.abs(Y), 1e-6, 1e3)
    x = np.pad(x, ((0, 0), (10, 10)))
    for i in range(5):
        assert np.allclose(gold['sc_x_in'][i, 0], x[:, i:i + 21].T, atol=1e-7)
    assert int(gold['sc_lost']) == 6


def test_get_z_for_mask_matches_reference(gold):
    zs = np.arange(4)[:, None, None] * np.ones((4, 2, 3))
    zn = -np.arange(1, 5)[:, None, None] * np.ones((4, 2, 3))
    for k in range(4):
        assert np.array_equal(get_z_for_mask(zs, zn, k, 4, 'zs_hat'), gold[f'zfm_zs_hat_{k}'])
        assert np.array_equal(get_z_for_mask(zs, zn, k, 4, ['zs_hat', 'zn_hat']), gold[f'zfm_both_{k}'])
        t = get_z_for_mask(torch.from_numpy(zs), torch.from_numpy(zn), k, 4, 'zs_hat')
        assert np.array_equal(t.numpy(), gold[f'zfm_zs_hat_{k}'])


def test_shapes_of_the_architecture():
    m = CRNN(n_ch=1)
    assert (m.x_out, m.y_out, m.mid_frame()) == (15, 4, 7)
    keys = set(m.state_dict())
    assert {'cnn.model.0.weight', 'cnn.model.1.running_mean', 'cnn.model.6.bias', 'rnn.model.0.rnn_layer.weight_ih_l0',
            'ff.layers.0.weight'} <= keys


@pytest.mark.parametrize('tag,n_ch', [('sc', 1), ('mc', 4)])
@pytest.mark.parametrize('ftp,nt', [('last', None), ('mid', 'scale_to_unit_norm'), ('mid', 'scale_to_1'), ('last', 'center_and_scale')])
def test_frame_to_pred_last_and_normalisations(gold, golden_dir, tag, n_ch, ftp, nt):
    """prepare_data's other output-frame choice ('last': pad (17, 3), all 15 GRU steps, last output frame) and the three
    numpy normalisations, against the reference's own prepare_data / normalization / reshape_mask (crnn_variants_ref.npz)."""
    var = np.load(os.path.join(golden_dir, 'crnn_variants_ref.npz'))
    model = _model(gold, tag, n_ch)
    chans = [np.abs(gold[f'{tag}_Y'])]
    if n_ch > 1:
        chans += [np.abs(z) for z in gold[f'{tag}_Z']]
    mag = torch.from_numpy(np.stack(chans)[None].transpose(0, 1, 3, 2).copy())
    mask = model.predict_masks(mag, frame_to_pred=ftp, norm_type=nt).numpy()[0]
    ref = var[f'{tag}_{ftp}_{nt}']
    assert mask.shape == ref.T.shape
    assert np.abs(mask - ref.T).max() < 1e-5, np.abs(mask - ref.T).max()
    with pytest.raises(NotImplementedError):
        model.predict_masks(mag, norm_type='pcen')
    with pytest.raises(ValueError):
        model.predict_masks(mag, frame_to_pred='all')


def test_dnn_utils_normalization_vs_reference_golden(golden_dir):
    """disco_amd/dnn/utils.py:normalization against outputs of the reference's own function (dnn/utils.py:14-41)."""
    import os
    from disco_amd.dnn.utils import normalization
    g = np.load(os.path.join(golden_dir, 'dnn_normalization_ref.npz'))
    x = torch.from_numpy(g['x'])
    for nt in ('scale_to_unit_norm', 'scale_to_1', 'center_and_scale', 'none'):
        for axis in (0, 1, 2):
            got = normalization(x, None if nt == 'none' else nt, axis).numpy()
            ref = g[f'{nt}_axis{axis}']
            assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max(), (nt, axis)
    assert normalization(x, 'something_else') is x          # unknown types pass through, as in the reference
