"""`intern_filter` with the reference's signature (disco_theque/se_utils/internal_formulas.py:31-81), on the MI355X.

Only the branch the hot path uses is implemented on the GPU: type='gevd', rank=1 (tango.py:367, 443).  The other
branches of the reference ('r1-mwf', 'mwf') are dead on the hot path and raise NotImplementedError;
an unknown type raises AttributeError and the default rank='Full' raises TypeError, exactly like the reference."""
import numpy as np

from .._engines import get_engine

eps = 2.220446049250313e-16      # internal_formulas.py:6
eta = 1e6                        # internal_formulas.py:7


def intern_filter(Rxx, Rnn, mu=1, type='r1-mwf', rank='Full'):
    """Returns (Wint, (t1, sort_index)).  Wint, t1: complex128 (P,) arrays (computed in float64 on the GPU, returned
    through complex64).  sort_index is None: the GPU solver extracts only the dominant generalized eigenpair."""
    if type in ('r1-mwf', 'mwf'):
        raise NotImplementedError("intern_filter: only type='gevd', rank=1 (the hot-path branch) runs on the GPU")
    if type != 'gevd':
        raise AttributeError('Unknown filter reference')                      # internal_formulas.py:79
    if not isinstance(rank, (int, np.integer)):
        raise TypeError("slice indices must be integers (rank='Full' is unusable with type='gevd', as in the reference)")
    if int(rank) != 1:
        raise NotImplementedError('intern_filter: only rank=1 runs on the GPU')
    Rxx = np.ascontiguousarray(Rxx, dtype=np.complex64)
    Rnn = np.ascontiguousarray(Rnn, dtype=np.complex64)
    assert Rxx.shape == Rnn.shape and Rxx.ndim == 2 and Rxx.shape[0] == Rxx.shape[1]
    eng = get_engine(rooms=1, nodes=1, mics=1, length=1024)
    w, t1 = eng.gevd_mwf_r1(Rxx[None], Rnn[None], mu=float(mu))
    return w.numpy()[0].astype(np.complex128), (t1.numpy()[0].astype(np.complex128), None)


def intern_filter_batched(Rxx, Rnn, mu=1):
    """(..., P, P) pencils -> w, t1 (..., P): the batched form the engine actually runs."""
    Rxx = np.ascontiguousarray(Rxx, dtype=np.complex64)
    Rnn = np.ascontiguousarray(Rnn, dtype=np.complex64)
    eng = get_engine(rooms=1, nodes=1, mics=1, length=1024)
    w, t1 = eng.gevd_mwf_r1(Rxx, Rnn, mu=float(mu))
    return w.numpy(), t1.numpy()
