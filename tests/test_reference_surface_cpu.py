"""Host-side behaviour of the reference call surface that needs no GPU: argument validation mirrors the
reference's exceptions, the C-ABI library exports every symbol of include/disco_hip.h, and the package refuses
to run without the HIP library (there is no CPU path)."""
import ctypes
import os
import re

import numpy as np
import pytest

from disco_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_binding_agree():
    hdr = open(os.path.join(REPO, 'include', 'disco_hip.h')).read()
    declared = set(re.findall(r'\b(disco_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'disco_hip'}
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)


def test_library_exports_every_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip('libdisco_hip.so not built yet (run __graft_entry__.build())')
    lib = ctypes.CDLL(_lib.LIB_PATH)          # dlopen needs no device
    for name in _lib.PROTOTYPES:
        assert hasattr(lib, name), name
    _lib.bind(lib)
    assert b'gfx950' in lib.disco_version()


def test_cfg_struct_matches_header():
    assert ctypes.sizeof(_lib.DiscoCfg) == 16 * 4


def test_mask_type_errors_before_device():
    from disco_amd.engine import parse_mask_type
    from disco_amd.sigproc_utils import tf_mask
    assert parse_mask_type('irm2') == (0, 2) and parse_mask_type('ibm1') == (1, 1)
    with pytest.raises(ValueError):
        tf_mask(np.zeros((3, 3), complex), np.zeros((3, 3), complex), type='foo1')


def test_intern_filter_argument_errors():
    from disco_amd.se_utils.internal_formulas import intern_filter
    R = np.eye(3)
    with pytest.raises(AttributeError):
        intern_filter(R, R, type='nope')
    with pytest.raises(TypeError):
        intern_filter(R, R, type='gevd')               # default rank='Full', as in the reference
    with pytest.raises(TypeError):
        intern_filter(R, R, type='gevd', rank='full')  # the reference slices D[rank:, :] with it (internal_formulas.py:66-67)


def test_offline_tango_argument_errors():
    from disco_amd.speech_enhancement.tango import concatenate_signals, offline_tango
    y = np.zeros((2, 2, 4096), np.float32)
    with pytest.raises(ValueError):
        offline_tango(y, y, y, vads=['bad1', 'irm1'])
    with pytest.raises(ValueError):
        offline_tango(y, y, y, vads=['crnn', 'crnn'])            # DNN masks need models in `mods`
    with pytest.raises(NotImplementedError):
        offline_tango(y, y, y, vads=['rnn', 'rnn'])              # dnn/models/heymann.py is not shipped by the reference
    a = [np.ones((2, 3, 4)), 2 * np.ones((2, 3, 4))]
    z = [5 * np.ones((3, 4)), 7 * np.ones((3, 4))]
    c = concatenate_signals(a, z, 0)
    assert c.shape == (3, 3, 4) and c[2, 0, 0] == 7


def test_no_cpu_fallback(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    monkeypatch.setattr(_lib, '_lib', None)
    with pytest.raises(RuntimeError, match='no CPU path'):
        _lib.load()
