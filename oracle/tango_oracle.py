"""CPU oracle of the two-step Tango / DANSE MWF (`offline_tango`).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Two restatements of
disco_theque/speech_enhancement/tango.py:252-457:

  * `offline_tango_literal`  -- loop nest kept as in the reference (node x bin x frame loops of
    np.outer / np.inner, one intern_filter call per (node, bin)).  This is "the reference CPU path";
    it is what bench.py times as `cpu_baseline` (kind "port").
  * `offline_tango_vec`      -- the same arithmetic vectorised (einsum covariance, per-bin solve),
    in two precisions: 'ref32' follows the reference's dtype flow (complex64 STFT / covariance /
    LAPACK, complex128 filter, complex64 outputs), 'f64' keeps everything in float64/complex128
    with no intermediate rounding (the "exact" answer used to judge rounding noise).

Both are pinned against the reference's own code (tests/golden/make_golden.py ->
tests/golden/tango_*.npz) by tests/test_oracle_golden.py.

Reference line map
  tango.py:326-348  STFT of y,s,n per channel; mask at the reference mic; s_hat = m*Y, n_hat = (1-m)*Y
  tango.py:357-364  Rss_loc[f] = mean_t s_hat s_hat^H,  Rnn_loc[f] = mean_t n_hat n_hat^H
  tango.py:367-368  w_loc, t1 = intern_filter(Rss, Rnn, mu=1, 'gevd', rank=1)
  tango.py:369-374  z_y = w^H y, z_s = w^H s, z_n = w^H n  (z_gevd_* = t1^T s|n, unconjugated, never returned)
  tango.py:376      zn = Y[ref] - z_y
  tango.py:142-155  concatenate_signals: [Y_k ; m*z_j (j<k) ; m*z_j (j>k)]
  tango.py:382-409  step-2 inputs and masks; mask_for_z modes
  tango.py:411-450  step-2 covariance (P = M + K - 1), solve, apply
  tango.py:457      return (yf, sf, nf, z_y, z_s, z_n, zn, masks_z, mask_w)
"""
import copy

import numpy as np

from . import mwf_oracle as mo
from . import stft_oracle as so


def _as_node_list(x):
    return [np.asarray(xk) for xk in x]


def concatenate_signals(y, z, k, m=1):
    """tango.py:142-155."""
    return np.concatenate((y[k], m * np.array(z)[:k], m * np.array(z)[k + 1:]), axis=0)


def _oracle_mask(S, N, mask_type, ts=None, n_fft=512, hop=256):
    if mask_type[:-1] in ('irm', 'ibm', 'iam'):
        return mo.tf_mask(S, N, type=mask_type)
    if mask_type == 'ivad':                                    # tango.py:217-221, ts = s[node][0]
        return mo.ivad_mask(ts, np.shape(S), n_fft, hop)
    raise ValueError('Unknown value for `mask_type`')          # tango.py:223


def _cov_mean(V, ref32):
    """mean_t V[:, f, t] V[:, f, t]^H for every f.  V: (P, F, T) -> (F, P, P)."""
    Vt = np.transpose(V, (1, 2, 0))                                  # (F, T, P)
    if ref32:
        phi = Vt[:, :, :, None] * np.conjugate(Vt)[:, :, None, :]    # (F, T, P, P) complex64
        return np.mean(phi, axis=1)                                  # complex64 accumulate, like np.mean(np.array(phi_s_f), axis=0)
    return np.einsum('ftp,ftq->fpq', Vt, np.conjugate(Vt)) / Vt.shape[1]


def _solve_bins(Rss, Rnn, mu, solver):
    """One (w, t1) per frequency bin.  Rss, Rnn: (F, P, P)."""
    if solver == 'eigh':
        w, t1, _ = mo.gevd_mwf_r1_hermitian(Rss, Rnn, mu)
        return w, t1
    F, P, _ = Rss.shape
    w = np.zeros((F, P), np.complex128)
    t1 = np.zeros((F, P), np.complex128)
    for f in range(F):
        wf, (t1f, _) = mo.intern_filter(Rss[f], Rnn[f], mu=mu, type='gevd', rank=1)
        w[f], t1[f] = wf, t1f
    return w, t1


def offline_tango_vec(y, s, n, vads=('irm1', 'irm1'), mask_for_z='local', n_fft=512, hop=256,
                      ref_mics=None, mu=1, precision='f64', pad_mode='reflect', solver='eig',
                      masks=None, extra_iters=0):
    """Vectorised two-step Tango.  y, s, n: [node][channel] -> time (list or (K, M, L) array).

    masks: optional (masks_z, mask_w) lists of (F, T) arrays replacing the oracle masks (DNN stand-in).
    extra_iters: DANSE-style continuation (NOT in the reference, which is strictly two-step, tango.py:1-2): after
    step 2 every node re-compresses with the local part of its global filter, z_k <- w_glo,k[:M]^H y_k, and step 2 is
    run again (BASELINE.json configs[4]; SURVEY.md section 7 item 10 defines it; no reference parity exists).
    Returns a dict with every intermediate: Y,S,N STFTs, masks_z, mask_w, Rss_loc, Rnn_loc, w_loc, t1_loc,
    z_y, z_s, z_n, zn, Rss_glo, Rnn_glo, w_glo, t1_glo, yf, sf, nf (lists over nodes).
    """
    ref32 = (precision == 'ref32')
    cdt = np.complex64 if ref32 else np.complex128
    y, s, n = _as_node_list(y), _as_node_list(s), _as_node_list(n)
    K = len(y)
    ref_mics = [0] * K if ref_mics is None else list(ref_mics)
    # 'previous' = any other string in the reference: the final else of tango.py:428-429, remote rows = unmasked z_y
    MODES = ('local', None, 'distant', 'compressed', 'use_oracle_refs', 'use_oracle_zs', 'previous')
    if mask_for_z not in MODES:
        raise NotImplementedError(f'oracle implements mask_for_z in {MODES}')
    oracle_sigs = isinstance(mask_for_z, str) and 'use_oracle_' in mask_for_z          # tango.py:343

    Y = [so.stft(y[k], n_fft, hop, pad_mode, cdt) for k in range(K)]          # (M, F, T)
    S = [so.stft(s[k], n_fft, hop, pad_mode, cdt) for k in range(K)]
    N = [so.stft(n[k], n_fft, hop, pad_mode, cdt) for k in range(K)]

    out = dict(Y=Y, S=S, N=N)
    keys = ['masks_z', 'Rss_loc', 'Rnn_loc', 'w_loc', 't1_loc', 'z_y', 'z_s', 'z_n', 'zn',
            'mask_w', 'Rss_glo', 'Rnn_glo', 'w_glo', 't1_glo', 'yf', 'sf', 'nf']
    for key in keys:
        out[key] = [None] * K

    # ---- step 1 (tango.py:326-376)
    for k in range(K):
        r = ref_mics[k]
        m = masks[0][k] if masks is not None else _oracle_mask(S[k][r], N[k][r], vads[0], s[k][r], n_fft, hop)
        out['masks_z'][k] = m
        if oracle_sigs:                                                            # tango.py:343-345
            s_hat, n_hat = S[k], N[k]
        else:
            s_hat = m * Y[k]
            n_hat = (1 - m) * Y[k]
        Rss = _cov_mean(s_hat, ref32)
        Rnn = _cov_mean(n_hat, ref32)
        w, t1 = _solve_bins(Rss, Rnn, mu, solver)
        out['Rss_loc'][k], out['Rnn_loc'][k], out['w_loc'][k], out['t1_loc'][k] = Rss, Rnn, w, t1
        wc = np.conjugate(w)                                                   # (F, M)
        out['z_y'][k] = np.einsum('fm,mft->ft', wc, Y[k]).astype(cdt)
        out['z_s'][k] = np.einsum('fm,mft->ft', wc, S[k]).astype(cdt)
        out['z_n'][k] = np.einsum('fm,mft->ft', wc, N[k]).astype(cdt)
        out['zn'][k] = Y[k][r] - out['z_y'][k]

    for it in range(1 + extra_iters):
        if it > 0:
            if mask_for_z != 'local':
                raise NotImplementedError('extra_iters is defined for mask_for_z="local" only')
            for k in range(K):
                Mk = Y[k].shape[0]
                wl = np.conjugate(out['w_glo'][k][:, :Mk])
                out['z_y'][k] = np.einsum('fm,mft->ft', wl, Y[k]).astype(cdt)
                out['z_s'][k] = np.einsum('fm,mft->ft', wl, S[k]).astype(cdt)
                out['z_n'][k] = np.einsum('fm,mft->ft', wl, N[k]).astype(cdt)
        # ---- exchange + step 2 (tango.py:378-450)
        z_for_rs = copy.deepcopy(out['z_y'])
        z_for_rn = copy.deepcopy(out['z_y'])
        for k in range(K):
            if masks is not None:
                mw = masks[1][k]
            else:
                mw = _oracle_mask(S[k][0], N[k][0], vads[1], s[k][0], n_fft, hop)      # tango.py:391 (channel 0)
            out['mask_w'][k] = mw
        for k in range(K):                                                         # tango.py:396-409
            if mask_for_z == 'distant':
                z_for_rs[k] = z_for_rs[k] * out['mask_w'][k]
                z_for_rn[k] = z_for_rn[k] * (1 - out['mask_w'][k])
            elif mask_for_z == 'compressed':
                mc = _oracle_mask(out['z_s'][k], out['z_n'][k], vads[0])
                z_for_rs[k] = z_for_rs[k] * mc
                z_for_rn[k] = z_for_rn[k] * (1 - mc)
            elif mask_for_z == 'use_oracle_refs':
                z_for_rs[k] = S[k][ref_mics[k]]
                z_for_rn[k] = N[k][ref_mics[k]]
            elif mask_for_z == 'use_oracle_zs':
                z_for_rs[k] = out['z_s'][k]
                z_for_rn[k] = out['z_n'][k]
        s_hat_w = [out['mask_w'][k] * Y[k] for k in range(K)]
        n_hat_w = [(1 - out['mask_w'][k]) * Y[k] for k in range(K)]
        for k in range(K):
            if mask_for_z == 'local':
                ms, mn = out['mask_w'][k], 1 - out['mask_w'][k]
            elif mask_for_z is None:                                               # tango.py:419-422
                ms, mn = 1, 1
                z_for_rn = out['zn']
            else:                                                                  # 'previous' branch, tango.py:428-429
                ms, mn = 1, 1
            in_y = concatenate_signals(Y, out['z_y'], k)
            in_s = concatenate_signals(S, out['z_s'], k)
            in_n = concatenate_signals(N, out['z_n'], k)
            phi_s = concatenate_signals(s_hat_w, [np.asarray(a, dtype=cdt) for a in z_for_rs], k, ms)
            phi_n = concatenate_signals(n_hat_w, [np.asarray(a, dtype=cdt) for a in z_for_rn], k, mn)
            Rss = _cov_mean(phi_s.astype(cdt), ref32)
            Rnn = _cov_mean(phi_n.astype(cdt), ref32)
            w, t1 = _solve_bins(Rss, Rnn, mu, solver)
            out['Rss_glo'][k], out['Rnn_glo'][k], out['w_glo'][k], out['t1_glo'][k] = Rss, Rnn, w, t1
            wc = np.conjugate(w)
            out['yf'][k] = np.einsum('fp,pft->ft', wc, in_y).astype(cdt)
            out['sf'][k] = np.einsum('fp,pft->ft', wc, in_s).astype(cdt)
            out['nf'][k] = np.einsum('fp,pft->ft', wc, in_n).astype(cdt)
    return out


def as_reference_tuple(out):
    """The 9-tuple `offline_tango` returns (tango.py:457)."""
    return (out['yf'], out['sf'], out['nf'], out['z_y'], out['z_s'], out['z_n'], out['zn'],
            out['masks_z'], out['mask_w'])


def offline_tango_literal(y, s, n, vads=('irm1', 'irm1'), mask_for_z='local', n_fft=512, hop=256,
                          ref_mics=None, mu=1, pad_mode='reflect', step1_only=False):
    """The reference's loop nest, kept as loops (tango.py:326-457; get_z_signals.py:213-317 when
    step1_only).  complex64 storage, one intern_filter per (node, bin), np.outer per frame for the
    statistics, np.inner per frame for the filtering.  Slow by construction: this is the CPU baseline."""
    y, s, n = _as_node_list(y), _as_node_list(s), _as_node_list(n)
    K = len(y)
    ref_mics = [0] * K if ref_mics is None else list(ref_mics)
    Mk = [yk.shape[0] for yk in y]
    Pk = [Mk[k] + K - 1 for k in range(K)]
    F = n_fft // 2 + 1
    T = so.n_frames_of(y[0].shape[-1], hop)
    c64 = np.complex64
    Y = [None] * K
    S = [None] * K
    N = [None] * K
    s_hat_z, n_hat_z = [None] * K, [None] * K
    masks_z, mask_w = [None] * K, [None] * K
    z_y = [np.zeros((F, T), c64) for _ in range(K)]
    z_s = [np.zeros((F, T), c64) for _ in range(K)]
    z_n = [np.zeros((F, T), c64) for _ in range(K)]
    zn = [None] * K
    for k in range(K):
        Yk, Sk, Nk, sh, nh = [], [], [], [], []
        mask_z = None
        for c in range(Mk[k]):
            Yk.append(so.stft(y[k][c], n_fft, hop, pad_mode))
            Sk.append(so.stft(s[k][c], n_fft, hop, pad_mode))
            Nk.append(so.stft(n[k][c], n_fft, hop, pad_mode))
            if c == ref_mics[k]:
                mask_z = _oracle_mask(Sk[c], Nk[c], vads[0], s[k][c], n_fft, hop)   # ts = s_ (tango.py:340-341)
                masks_z[k] = mask_z
            sh.append(mask_z * Yk[c])
            nh.append((1 - mask_z) * Yk[c])
        Y[k], S[k], N[k] = np.array(Yk), np.array(Sk), np.array(Nk)
        s_hat_z[k], n_hat_z[k] = np.array(sh), np.array(nh)
        for f in range(F):
            phi_s = [None] * T
            phi_n = [None] * T
            for t in range(T):
                a = s_hat_z[k][:, f, t]
                b = n_hat_z[k][:, f, t]
                phi_s[t] = np.outer(a, np.conjugate(a).T)
                phi_n[t] = np.outer(b, np.conjugate(b).T)
            Rss = np.mean(np.array(phi_s), axis=0)
            Rnn = np.mean(np.array(phi_n), axis=0)
            w_loc, _ = mo.intern_filter(Rss, Rnn, mu=mu, type='gevd', rank=1)
            wc = np.conjugate(w_loc)
            for t in range(T):
                z_y[k][f, t] = np.inner(wc, Y[k][:, f, t])
                z_s[k][f, t] = np.inner(wc, S[k][:, f, t])
                z_n[k][f, t] = np.inner(wc, N[k][:, f, t])
        zn[k] = Y[k][ref_mics[k]] - z_y[k]
    if step1_only:
        return z_y, z_s, z_n, zn, masks_z

    yf = [np.zeros((F, T), c64) for _ in range(K)]
    sf = [np.zeros((F, T), c64) for _ in range(K)]
    nf = [np.zeros((F, T), c64) for _ in range(K)]
    z_for_rs, z_for_rn = copy.deepcopy(z_y), copy.deepcopy(z_y)
    in_y, in_s, in_n = [None] * K, [None] * K, [None] * K
    for k in range(K):
        in_y[k] = concatenate_signals(Y, z_y, k)
        in_s[k] = concatenate_signals(S, z_s, k)
        in_n[k] = concatenate_signals(N, z_n, k)
        mask_w[k] = _oracle_mask(S[k][0], N[k][0], vads[1], s[k][0], n_fft, hop)
    s_hat_w = [[] for _ in range(K)]
    n_hat_w = [[] for _ in range(K)]
    for k in range(K):
        for c in range(Mk[k]):
            s_hat_w[k].append(mask_w[k] * Y[k][c])
            n_hat_w[k].append((1 - mask_w[k]) * Y[k][c])
        if mask_for_z == 'local':
            ms, mn = mask_w[k], 1 - mask_w[k]
        elif mask_for_z is None:
            ms, mn = 1, 1
            z_for_rn = zn
        else:
            raise NotImplementedError(mask_for_z)
        phi_in_s = concatenate_signals(s_hat_w, z_for_rs, k, ms)
        phi_in_n = concatenate_signals(n_hat_w, z_for_rn, k, mn)
        for f in range(F):
            phi_s = [None] * T
            phi_n = [None] * T
            for t in range(T):
                a = phi_in_s[:, f, t]
                b = phi_in_n[:, f, t]
                phi_s[t] = np.outer(a, np.conjugate(a).T)
                phi_n[t] = np.outer(b, np.conjugate(b).T)
            Rss = np.mean(np.array(phi_s), axis=0)
            Rnn = np.mean(np.array(phi_n), axis=0)
            w_glo, _ = mo.intern_filter(Rss, Rnn, mu=mu, type='gevd', rank=1)
            wc = np.conjugate(w_glo)
            for t in range(T):
                yf[k][f, t] = np.inner(wc, in_y[k][:, f, t])
                sf[k][f, t] = np.inner(wc, in_s[k][:, f, t])
                nf[k][f, t] = np.inner(wc, in_n[k][:, f, t])
    return yf, sf, nf, z_y, z_s, z_n, zn, masks_z, mask_w
