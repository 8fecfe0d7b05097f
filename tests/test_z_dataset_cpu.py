"""On-disk z dataset layout (get_z_signals.py:320-359): names, shapes, dtypes, skip rule -- host logic, no GPU."""
import os

import numpy as np

from disco_amd.speech_enhancement import z_dataset as zd


def test_directory_name():
    assert zd.get_directory_name([[0, 6]]) == '0-6'
    assert zd.get_directory_name([[3, 6], [5, 15]]) == '3-6_5-15'


def test_write_read_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    R, K, T, F = 2, 3, 5, 9
    z = (rng.standard_normal((R, K, T, F)) + 1j * rng.standard_normal((R, K, T, F))).astype(np.complex64)
    zn = (rng.standard_normal((R, K, T, F)) + 1j * rng.standard_normal((R, K, T, F))).astype(np.complex64)
    root = str(tmp_path)
    assert zd.write_z_dataset(root, [11001, 11002], 'ssn', z, zn) == [11001, 11002]
    f = os.path.join(root, 'raw', '0-6', 'zs_hat', '11002_ssn_Node-3.npy')          # the reference's file name
    assert os.path.isfile(f)
    a = np.load(f)
    assert a.shape == (F, T) and a.dtype == np.complex64 and np.array_equal(a, z[1, 2].T)
    m = zd.read_z(root, 11001, 'ssn', 0, kind='zn_hat', normed=True)
    assert m.dtype == np.float32 and np.allclose(m, np.abs(zn[0, 0].T))
    # second call: both rooms are skipped (last node's normed/abs/zn_hat file exists)
    assert zd.write_z_dataset(root, [11001, 11002], 'ssn', z, zn) == []
    assert zd.write_z_dataset(root, [11001, 11003], 'ssn', z, zn) == [11003]


def test_wav_and_mask_files(tmp_path):
    """tango.py:595-613 file names; 16-bit PCM round trip within half an LSB."""
    from disco_amd.speech_enhancement import results_io as rio
    rng = np.random.default_rng(1)
    K, L, T, F = 2, 4000, 6, 9
    sig = {'in_mix': 0.3 * rng.standard_normal((K, L)), 'out_mix': 0.2 * rng.standard_normal((K, L)), 'mid_z': 0.2 * rng.standard_normal((K, L))}
    sig['in_mix'][0, 10] = 3.0                                     # clipped like any PCM writer would
    mz = rng.random((K, T, F)).astype(np.float32)
    files = rio.write_room_results(str(tmp_path), 11001, 'ssn', sig, masks_z=mz, mask_w=mz)
    assert os.path.join(str(tmp_path), 'WAV', '11001', 'out_mix-ssn_Node-2.wav') in files
    assert os.path.join(str(tmp_path), 'MASK', '11001', 'step2_ssn_Node-1.npy') in files
    x, fs = rio.read_wav(os.path.join(str(tmp_path), 'WAV', '11001', 'in_mix-ssn_Node-1.wav'))
    assert fs == 16000 and len(x) == L
    assert np.max(np.abs(x - np.clip(sig['in_mix'][0], -1, 1 - 2.0 ** -15))) <= 0.5 / 32768 + 1e-9
    assert np.load(os.path.join(str(tmp_path), 'MASK', '11001', 'step1_ssn_Node-2.npy')).shape == (F, T)
