"""Host-side logic of the reference call surface (disco_amd.speech_enhancement.tango.offline_tango) exercised without a GPU:
the package's library handle is swapped, FOR THESE TESTS ONLY, for the hipemu build of the same kernel sources.  Covers the
branches only a real call reaches: ragged channel counts (one-node shards), 'ivad' masks, CRNN masks through `mods`.
(The package itself never loads the emulator; the real parity runs are tests/test_gpu_reference_surface.py, -m gpu.)"""
import os

import numpy as np
import pytest

import emu_build
from disco_amd import _engines, _lib
from oracle import tango_oracle as to


def relerr(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


@pytest.fixture()
def emulated_package(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', emu_build.load_emu())
    _engines._cache.clear()
    yield
    _engines._cache.clear()


NAMES = ['yf', 'sf', 'nf', 'z_y', 'z_s', 'z_n', 'zn', 'masks_z', 'mask_w']


def test_ragged_nodes_against_reference_golden(emulated_package, golden_dir):
    from disco_amd.speech_enhancement.tango import offline_tango
    g = np.load(os.path.join(golden_dir, 'tango_ref_k3ragged.npz'))
    K = int(g['K'])
    y, s, n = ([g[f'{c}{k}'] for k in range(K)] for c in 'ysn')
    res = offline_tango(y, s, n, vads=['irm1', 'irm1'], mods=[None, None])
    o = to.as_reference_tuple(to.offline_tango_vec(y, s, n, vads=['irm1', 'irm1'], precision='f64', solver='eigh'))
    for i, nm in enumerate(NAMES):
        for k in range(K):
            assert res[i][k].shape == g[f'{nm}{k}'].shape
            assert relerr(res[i][k], o[i][k]) < (2e-5 if 'mask' in nm else 2e-4), (nm, k)
            if i < 7:
                assert relerr(res[i][k], g[f'{nm}{k}']) < 2e-2, (nm, k)          # the reference's own complex64 outputs


def test_ivad_masks_against_reference_golden(emulated_package, golden_dir):
    from disco_amd.speech_enhancement.tango import offline_tango
    g = np.load(os.path.join(golden_dir, 'ivad_ref.npz'))
    K = int(g['K'])
    y, s, n = ([g[f'{c}{k}'] for k in range(K)] for c in 'ysn')
    res = offline_tango(y, s, n, vads=['ivad', 'ivad'], mods=[None, None])
    for i, nm in enumerate(NAMES):
        for k in range(K):
            if 'mask' in nm:
                assert np.array_equal(res[i][k], g[f'{nm}{k}']), (nm, k)
            else:
                assert relerr(res[i][k], g[f'{nm}{k}']) < 1e-3, (nm, k)


@pytest.mark.parametrize('mode', ['local', None, 'distant', 'compressed', 'use_oracle_refs', 'use_oracle_zs', 'previous'])
def test_device_resident_reference_outputs_every_mode(emulated_package, golden_dir, mode):
    """disco_tango_reference (one library call for the nine outputs, every mask_for_z variant formed on the device) against the
    float64 oracle on a synthetic room and against the reference's own outputs on its golden scene."""
    from disco_amd import synth
    from disco_amd.speech_enhancement.tango import offline_tango
    y, s, n, _ = synth.make_room_numpy(8, K=3, M=2, L=7000)
    res = offline_tango(y, s, n, vads=['irm1', 'irm1'], mask_for_z=mode)
    o = to.as_reference_tuple(to.offline_tango_vec(y, s, n, vads=['irm1', 'irm1'], mask_for_z=mode, precision='f64', solver='eigh'))
    for i, nm in enumerate(NAMES):
        for k in range(3):
            assert relerr(res[i][k], o[i][k]) < (2e-5 if 'mask' in nm else 1e-4), (mode, nm, k, relerr(res[i][k], o[i][k]))
    if mode in ('local', None):
        return
    g = np.load(os.path.join(golden_dir, 'tango_ref_modes_k2m2.npz'))
    res = offline_tango([g['y0'], g['y1']], [g['s0'], g['s1']], [g['n0'], g['n1']], vads=['irm1', 'irm1'], mask_for_z=mode)
    for k in range(2):
        assert relerr(res[0][k], g[f'{mode}_yf{k}']) < 1e-2


def test_step1_only_variant_and_mixed_mask_types(emulated_package):
    """get_z_signals.offline_tango runs step 1 ONLY (steps = 1 of disco_tango_reference): a single model / mask type is enough,
    as in the reference (get_z_signals.py:277-281).  A step-2 mask type that differs from step 1's goes through a second mask
    computation and is passed in."""
    from disco_amd import synth
    from disco_amd.speech_enhancement import get_z_signals
    from disco_amd.speech_enhancement.tango import offline_tango
    y, s, n, _ = synth.make_room_numpy(7, K=2, M=2, L=6000)
    z_y, z_s, z_n, zn, masks_z = get_z_signals.offline_tango(y, s, n, vads='irm1')
    o = to.offline_tango_vec(y, s, n, vads=['irm1', 'irm1'], precision='f64', solver='eigh')
    for k in range(2):
        assert relerr(z_y[k], o['z_y'][k]) < 1e-4 and relerr(zn[k], o['zn'][k]) < 1e-4 and relerr(z_s[k], o['z_s'][k]) < 1e-4
        assert relerr(masks_z[k], o['masks_z'][k]) < 2e-5
    res = offline_tango(y, s, n, vads=['irm1', 'iam2'])
    o2 = to.as_reference_tuple(to.offline_tango_vec(y, s, n, vads=['irm1', 'iam2'], precision='f64', solver='eigh'))
    for i, nm in enumerate(NAMES):
        for k in range(2):
            assert relerr(res[i][k], o2[i][k]) < (5e-5 if 'mask' in nm else 1e-4), (nm, k, relerr(res[i][k], o2[i][k]))


def test_intern_filter_every_branch_against_reference_golden(emulated_package, golden_dir):
    """intern_filter's three branches ('gevd' rank 1, 'r1-mwf' -- the default --, 'mwf') against the reference's own outputs."""
    from disco_amd.se_utils.internal_formulas import intern_filter
    g = np.load(os.path.join(golden_dir, 'intern_filter_ref.npz'))
    seen = set()
    for i in range(int(g['n_cases'])):
        typ = str(g[f'c{i}_type'])
        seen.add(typ)
        kw = dict(type='gevd', rank=1) if typ == 'gevd' else dict(type=typ)
        w, (t1, si) = intern_filter(g[f'c{i}_Rxx'], g[f'c{i}_Rnn'], mu=1, **kw)
        assert relerr(w, g[f'c{i}_w']) < 2e-4 and relerr(t1, g[f'c{i}_t1']) < 2e-4, (i, typ, relerr(w, g[f'c{i}_w']))
    assert seen == {'gevd', 'r1-mwf', 'mwf'}


def test_long_reference_golden_scene_direct_1e4(emulated_package, golden_dir):
    """The reference's OWN outputs on a longer, well-conditioned scene (tests/golden/make_golden_long.py: 101 frames, 2 x 3
    microphones, every pencil well conditioned) against the HIP path DIRECTLY at the north star's 1e-4 -- no oracle in between."""
    from disco_amd.speech_enhancement.tango import offline_tango
    g = np.load(os.path.join(golden_dir, 'tango_ref_long.npz'))
    K = int(g['K'])
    y, s, n = ([g[f'{c}{k}'] for k in range(K)] for c in 'ysn')
    res = offline_tango(y, s, n, vads=['irm1', 'irm1'], mods=[None, None])
    names = ['yf', 'sf', 'nf', 'z_y', 'z_s', 'z_n', 'zn', 'masks_z', 'mask_w']
    for i, nm in enumerate(names):
        for k in range(K):
            if f'{nm}{k}' in g.files:
                e = relerr(res[i][k], g[f'{nm}{k}'])
                assert e < (2e-5 if 'mask' in nm else 1e-4), (nm, k, e)
