"""NumPy restatement of the reference's mask and MWF-solve primitives.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows, line by line:
  * disco_theque/dnn/utils.py:44-71  (= disco_theque/sigproc_utils.py:58-86)   tf_mask
  * disco_theque/se_utils/internal_formulas.py:31-81                           intern_filter
  * disco_theque/se_utils/internal_formulas.py:84-103                          spatial_correlation_matrix
Pinned against the reference's own functions by tests/golden/make_golden.py
(fixtures tests/golden/intern_filter_*.npz, tf_mask_*.npz).
"""
import sys

import numpy as np
import scipy.linalg

EPS = sys.float_info.epsilon        # internal_formulas.py:6
ETA = 1e6                           # internal_formulas.py:7


def tf_mask(s, n, type='irm1', bin_thr=0):
    """dnn/utils.py:44-71.  s, n: STFTs of target and noise; returns the TF mask."""
    power = int(type[-1])
    if 'irm' in type:
        n_ = np.maximum(abs(n), EPS)
        xi = (abs(s) / n_) ** power
        m = xi / (1 + xi)
    elif 'ibm' in type:
        n_ = np.maximum(abs(n), EPS)
        xi = (abs(s) / n_) ** power
        m = (xi >= 10 ** (bin_thr / 10))            # math_utils.py:46-62 db2lin, power quantity
    elif 'iam' in type:
        m = (abs(s) / abs(s + n)) ** power
    else:
        raise ValueError('Unknown mask type. Should be "irmX", "ibmX" or "iamX"')
    return m


def intern_filter(Rxx, Rnn, mu=1, type='r1-mwf', rank='Full'):
    """internal_formulas.py:31-81, same branches, same dtype flow, same return structure."""
    P = np.shape(Rxx)[0]
    t1 = np.zeros(P)
    t1[0] = 1.0                                                     # :43  e1
    sort_index = None
    if type == 'r1-mwf':                                            # :45-54
        D, X = np.linalg.eig(Rxx)
        D = np.real(D)
        Dmax, maxind = D.max(), D.argmax()
        Rxx = np.outer(np.abs(Dmax) * X[:, maxind], np.conjugate(X[:, maxind]).T)
        Pm = np.linalg.lstsq(Rnn, Rxx, rcond=None)[0]
        Wint = 1 / (mu + np.trace(Pm)) * Pm[:, 0]
    elif type == 'gevd':                                            # :56-73
        D, Q = scipy.linalg.eig(Rxx, Rnn)
        D = np.maximum(D, EPS * np.ones(np.shape(D)))
        D = np.minimum(D, ETA * np.ones(np.shape(D)))
        sort_index = np.argsort(D)
        D = np.diag(D[sort_index[::-1]])
        Q = Q[:, sort_index[::-1]]
        if rank != 'full':
            D[rank:, :] = 0          # rank='Full' (the default) raises TypeError here, as in the reference (:66-67)
        Qi = np.linalg.inv(Q)
        Wint = np.matmul(Q, np.matmul(D, np.matmul(np.linalg.inv(D + mu * np.eye(len(D))), Qi)))[:, 0]
        t1 = Q[:, 0] * Qi[0, 0]
    elif type == 'mwf':                                             # :74-76
        Pm = np.linalg.lstsq(Rnn + Rxx, Rxx, rcond=None)[0]
        Wint = Pm[:, 0]
    else:
        raise AttributeError('Unknown filter reference')
    return Wint, (t1, sort_index)


def gevd_mwf_r1_hermitian(Rxx, Rnn, mu=1.0):
    """Batched float64 closed form of the live branch (type='gevd', rank=1) for Hermitian pencils.

    With Rnn = L L^H, C = L^-1 Rxx L^-H = V diag(d) V^H (d descending), the generalized
    eigenvectors are Q = L^-H V, Q^-1 = V^H L^H, hence
        w  = q0 * d0/(d0+mu) * (Q^-1)[0,0] = L^-H v0 * d0/(d0+mu) * L[0,0] * conj(v0[0])
        t1 = q0 * (Q^-1)[0,0]
    which is invariant to the scale/phase of q0 (the reference's LAPACK normalisation does not
    matter).  d0 is clamped to [EPS, ETA] as at internal_formulas.py:59-62.
    Rxx, Rnn: (..., P, P).  Returns w, t1 (..., P) complex128 and d0 (...,).
    """
    Rxx = np.asarray(Rxx, dtype=np.complex128)
    Rnn = np.asarray(Rnn, dtype=np.complex128)
    L = np.linalg.cholesky(Rnn)
    Li = np.linalg.inv(L)
    C = Li @ Rxx @ np.conjugate(np.swapaxes(Li, -1, -2))
    C = 0.5 * (C + np.conjugate(np.swapaxes(C, -1, -2)))
    d, V = np.linalg.eigh(C)
    d0 = np.clip(d[..., -1], EPS, ETA)
    v0 = V[..., :, -1]
    q0 = np.einsum('...ji,...j->...i', np.conjugate(Li), v0)        # L^-H v0
    g = (L[..., 0, 0] * np.conjugate(v0[..., 0]))[..., None]
    t1 = q0 * g
    w = t1 * (d0 / (d0 + mu))[..., None]
    return w, t1, d0


def spatial_correlation_matrix(Rxx, x, lambda_cor=0.95, M=None):
    """internal_formulas.py:84-103 (online smoothing; not on the shipped batch path)."""
    if M is None:
        return lambda_cor * Rxx + (1 - lambda_cor) * np.outer(x, np.conjugate(x).T)
    return lambda_cor * Rxx + M * (1 - lambda_cor) * np.outer(x, np.conjugate(x).T)


def vad_oracle_batch(x_, win_len=512, win_hop=256, thr=0.001, rat=2):
    """sigproc_utils.py:12-55 -- power-based oracle VAD, one decision per window, written back over the window's samples."""
    x = x_ - np.mean(x_)
    x2 = abs(x ** 2)
    thr_ = thr * np.quantile(x2, 0.99)
    vad_o = np.zeros(len(x2))
    for n in np.arange(int(np.ceil((len(x2) - win_len) / win_hop + 1))):
        lo, hi = n * win_hop, np.minimum(n * win_hop + win_len, len(x2))
        nb_va = np.sum(x2[lo:hi] > thr_)
        if nb_va >= int((hi - lo) / rat):                      # np.int in the reference (numpy 1.18)
            vad_o[lo:hi] = 1
    return vad_o


def ivad_mask(ts, shape, n_fft=512, hop=256):
    """get_mask's 'ivad' branch (tango.py:217-221): the VAD sampled every hop, tiled over frequency; float64 zeros beyond."""
    m = np.zeros(shape)
    vad = vad_oracle_batch(ts, win_len=n_fft, win_hop=hop)[::hop]
    m[:, :len(vad)] = np.tile(vad, (shape[0], 1))
    return m
