"""`tf_mask` with the reference's signature (disco_theque/sigproc_utils.py:58-86 == dnn/utils.py:44-71)."""
import numpy as np

from ._engines import get_engine
from .engine import parse_mask_type


def tf_mask(s, n, type='irm1', bin_thr=0):
    """Oracle TF mask from target / noise STFTs.  Raises ValueError for an unknown type (dnn/utils.py:69) and
    AssertionError on a shape mismatch (sigproc_utils.py:71), like the reference."""
    parse_mask_type(type)                                  # ValueError before touching the device
    s = np.asarray(s)
    n = np.asarray(n)
    assert s.shape == n.shape, 'Target and noise STFTs should have the same shape'
    eng = get_engine(rooms=1, nodes=1, mics=1, length=1024)
    m = eng.tf_mask(np.ascontiguousarray(s, dtype=np.complex64), np.ascontiguousarray(n, dtype=np.complex64),
                    type=type, bin_thr=float(bin_thr)).numpy()
    return m.astype(bool) if type.startswith('ibm') else m
