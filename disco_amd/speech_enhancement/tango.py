"""`offline_tango` with the reference's call surface (disco_theque/speech_enhancement/tango.py:252-457), on the MI355X.

Same arguments, same 9-tuple of per-node lists of (F, T) arrays.  Everything numerical runs in libdisco_hip.so;
this file only orders the C-ABI calls the way the reference orders its loop nests.
"""
import numpy as np

from .._engines import get_engine

N_FFT = 512                      # tango.py:28
N_HOP = 256                      # tango.py:29
MASK_Z = 'local'                 # tango.py:36


def concatenate_signals(y, z, k, m=1):
    """tango.py:142-155 -- [y_k ; m*z_j (j<k) ; m*z_j (j>k)] (host helper, numpy)."""
    return np.concatenate((y[k], m * np.array(z)[:k], m * np.array(z)[k + 1:]), axis=0)


def _as_batch(x):
    """[node][channel] -> time lists or (K, M, L) arrays -> (1, K, M, L) float32."""
    nodes = [np.asarray(xk, dtype=np.float32) for xk in x]
    if len({xk.shape for xk in nodes}) != 1:
        raise NotImplementedError('disco_amd.offline_tango needs the same number of channels at every node '
                                  '(the reference allows ragged nb_ch; the batched GPU layout is uniform)')
    return np.ascontiguousarray(np.stack(nodes)[None])


def _mask_names(vads):
    if isinstance(vads, str):
        vads = [vads, vads]
    for v in vads[:2]:
        if v[:-1] not in ('irm', 'ibm', 'iam'):
            if 'rnn' in v or v == 'ivad':
                raise NotImplementedError(f"mask type '{v}': only the oracle TF masks are wired in this round (SURVEY 8f-1)")
            raise ValueError('Unknown value for `mask_type`')                 # tango.py:223
    return list(vads[:2])


def offline_tango_batched(y, s, n, vads='irm1', mods=None, mask_for_z=MASK_Z, z_sigs='zs_hat', n_fft=N_FFT,
                          pad_mode='reflect', ref_mic=0, mu=1.0):
    """y, s, n: (R, K, M, L) float32.  Returns a dict of device-computed arrays with a leading room axis, in the
    engine's frame-major layout (R, K, T, F): yf, sf, nf, z_y, z_s, z_n, zn, masks_z, mask_w."""
    vads = _mask_names(vads)
    MODES = ('local', None, 'distant', 'compressed', 'use_oracle_refs', 'use_oracle_zs')
    if mask_for_z not in MODES:
        raise NotImplementedError(f'mask_for_z must be one of {MODES}')       # 'use_oracle_sigs' is broken in the reference
    oracle_sigs = isinstance(mask_for_z, str) and 'use_oracle_' in mask_for_z
    y = np.ascontiguousarray(y, dtype=np.float32)
    s = np.ascontiguousarray(s, dtype=np.float32)
    n = np.ascontiguousarray(n, dtype=np.float32)
    R, K, M, L = y.shape
    eng = get_engine(rooms=R, nodes=K, mics=M, length=L, n_fft=n_fft, mask=vads[0], pad_mode=pad_mode, ref_mic=ref_mic,
                     mu=mu, staged_step2=True)
    T, F = eng.T, eng.F
    G = R * K
    Y = eng.stft(y.reshape(G, M, L)).reshape(R, K, T, F, M)
    S = eng.stft(s.reshape(G, M, L)).reshape(R, K, T, F, M)
    N = eng.stft(n.reshape(G, M, L)).reshape(R, K, T, F, M)
    Sh, Nh = S.numpy(), N.numpy()
    # masks at the reference mic (step 1, tango.py:338-342) and at channel 0 (step 2, tango.py:391)
    masks_z = eng.tf_mask(np.ascontiguousarray(Sh[..., ref_mic]), np.ascontiguousarray(Nh[..., ref_mic]), type=vads[0])
    same = (ref_mic == 0 and vads[1] == vads[0])
    mask_w = masks_z if same else eng.tf_mask(np.ascontiguousarray(Sh[..., 0]), np.ascontiguousarray(Nh[..., 0]), type=vads[1])
    mz = masks_z.numpy().astype(np.float32)
    mw = mz if same else mask_w.numpy().astype(np.float32)
    # step 1 (tango.py:357-376)
    if oracle_sigs:                                                            # statistics from the oracle images (tango.py:343-345)
        Rss, _ = eng.cov_masked(S, np.ones_like(mz))
        _, Rnn = eng.cov_masked(N, np.zeros_like(mz))
        w_loc, _ = eng.gevd_mwf_r1(Rss, Rnn, want_t1=False)
    else:
        eng.cov_masked(Y, mz)
        w_loc, _ = eng.gevd_mwf_r1_pending(M)
    z_y, z_s, z_n = eng.apply(Y, w_loc), eng.apply(S, w_loc), eng.apply(N, w_loc)
    zn = eng.noise_residual(Y, z_y)
    out = dict(masks_z=mz, mask_w=mw, z_y=z_y.numpy(), z_s=z_s.numpy(), z_n=z_n.numpy(), zn=zn.numpy())
    # exchange + step 2 (tango.py:378-450)
    if mask_for_z == 'local':
        eng.cov_masked(Y, mw, z_y, z_y, mask_remote=True)
    elif mask_for_z is None:                                                   # unmasked z / zn rows (tango.py:419-422)
        eng.cov_masked(Y, mw, z_y, zn, mask_remote=False)
    else:                                                                      # sender-side variants (tango.py:396-409): the
        zy = out['z_y']                                                        # remote rows are prepared on the host, then fed
        if mask_for_z == 'distant':                                            # to the same covariance kernel unmasked
            zs_rows, zn_rows = zy * mw, zy * (1 - mw)
        elif mask_for_z == 'compressed':
            mc = eng.tf_mask(out['z_s'], out['z_n'], type=vads[0]).numpy()
            zs_rows, zn_rows = zy * mc, zy * (1 - mc)
        elif mask_for_z == 'use_oracle_refs':
            zs_rows, zn_rows = np.ascontiguousarray(Sh[..., ref_mic]), np.ascontiguousarray(Nh[..., ref_mic])
        else:                                                                  # 'use_oracle_zs'
            zs_rows, zn_rows = out['z_s'], out['z_n']
        eng.cov_masked(Y, mw, zs_rows.astype(np.complex64), zn_rows.astype(np.complex64), mask_remote=False)
    w_glo, _ = eng.gevd_mwf_r1_pending(M + K - 1)
    out['yf'] = eng.apply(Y, w_glo, Z=z_y if K > 1 else None).numpy()
    out['sf'] = eng.apply(S, w_glo, Z=z_s if K > 1 else None).numpy()
    out['nf'] = eng.apply(N, w_glo, Z=z_n if K > 1 else None).numpy()
    return out


def offline_tango(y, s, n, vads='irm1', mods=None, mask_for_z=MASK_Z, z_sigs='zs_hat'):
    """Drop-in for the reference's `offline_tango`: returns
    (yf, sf, nf, z_y, z_s, z_n, zn, masks_z, mask_w), each a list over nodes of (F, T) arrays (tango.py:457)."""
    d = offline_tango_batched(_as_batch(y), _as_batch(s), _as_batch(n), vads=vads, mods=mods, mask_for_z=mask_for_z,
                              z_sigs=z_sigs)
    K = d['yf'].shape[1]
    names = ['yf', 'sf', 'nf', 'z_y', 'z_s', 'z_n', 'zn', 'masks_z', 'mask_w']
    return tuple([np.ascontiguousarray(d[nm][0, k].T) for k in range(K)] for nm in names)
