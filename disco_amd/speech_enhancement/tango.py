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
    """[node][channel] -> time lists or (K, M, L) arrays -> (1, K, M, L) float32; None when the nodes are ragged."""
    nodes = [np.asarray(xk, dtype=np.float32) for xk in x]
    if len({xk.shape for xk in nodes}) != 1:
        return None
    return np.ascontiguousarray(np.stack(nodes)[None])


def _mask_names(vads, mods=None):
    if isinstance(vads, str):
        vads = [vads, vads]
    mods = [None, None] if mods is None else list(mods)
    for i, v in enumerate(vads[:2]):
        if v[:-1] in ('irm', 'ibm', 'iam') or v == 'ivad':
            continue
        if v == 'crnn':                                                       # tango.py:209-215; models from dnn/crnn.py
            if mods[i] is None and not (i == 1 and vads[0] == 'crnn'):        # step 2 may re-use the step-1 mask (388-389)
                raise ValueError("vads='crnn' needs a model in `mods` (disco_amd.dnn.crnn.build_crnn)")
            continue
        if 'rnn' in v:
            raise NotImplementedError(f"mask type '{v}': the reference's RNN model file (dnn/models/heymann.py) is not shipped")
        raise ValueError('Unknown value for `mask_type`')                     # tango.py:223
    return list(vads[:2])


def _engine_mask_type(v):
    return 'irm1' if v in ('ivad', 'crnn') else v                             # the context's TF-mask type is unused for those


def _get_mask(eng, Sh_c, Nh_c, ts, vad, mod=None, Yh_c=None, z_rows=None):
    """get_mask (tango.py:189-225): TF mask of the channel's STFTs, the frame VAD of `ts`, or the CRNN's prediction from
    |Y| of that channel [+ |z| of the other nodes] (prepare_data / reshape_mask live in dnn/crnn.py:predict_masks).
    Sh_c, Nh_c, Yh_c (..., T, F) STFTs of one channel; ts (n_sig, L); z_rows (..., C-1, T, F) or None.  -> float32 (..., T, F)."""
    if vad == 'ivad':
        m = eng.mask_ivad(np.ascontiguousarray(ts, dtype=np.float32)).numpy()
        return m.reshape(Sh_c.shape)
    if vad == 'crnn':
        import torch
        T, F = Yh_c.shape[-2:]
        mag = np.abs(Yh_c).reshape(-1, 1, T, F)
        if z_rows is not None:
            mag = np.concatenate([mag, np.abs(z_rows).reshape(mag.shape[0], -1, T, F)], axis=1)
        par = next(mod.parameters())
        with torch.no_grad():
            m = mod.predict_masks(torch.from_numpy(np.ascontiguousarray(mag)).to(par.device, par.dtype))
        return m.float().cpu().numpy().reshape(Yh_c.shape)
    return eng.tf_mask(np.ascontiguousarray(Sh_c), np.ascontiguousarray(Nh_c), type=vad).numpy().astype(np.float32)


def offline_tango_batched(y, s, n, vads='irm1', mods=None, mask_for_z=MASK_Z, z_sigs='zs_hat', n_fft=N_FFT,
                          pad_mode='reflect', ref_mic=0, mu=1.0):
    """y, s, n: (R, K, M, L) float32.  Returns a dict of device-computed arrays with a leading room axis, in the
    engine's frame-major layout (R, K, T, F): yf, sf, nf, z_y, z_s, z_n, zn, masks_z, mask_w."""
    vads = _mask_names(vads, mods)
    mods = [None, None] if mods is None else list(mods)
    MODES = ('local', None, 'distant', 'compressed', 'use_oracle_refs', 'use_oracle_zs', 'previous')
    if mask_for_z not in MODES:
        raise NotImplementedError(f'mask_for_z must be one of {MODES}')       # 'use_oracle_sigs' is broken in the reference
    oracle_sigs = isinstance(mask_for_z, str) and 'use_oracle_' in mask_for_z
    y = np.ascontiguousarray(y, dtype=np.float32)
    s = np.ascontiguousarray(s, dtype=np.float32)
    n = np.ascontiguousarray(n, dtype=np.float32)
    R, K, M, L = y.shape
    eng = get_engine(rooms=R, nodes=K, mics=M, length=L, n_fft=n_fft, mask=_engine_mask_type(vads[0]), pad_mode=pad_mode,
                     ref_mic=ref_mic, mu=mu, staged_step2=True)
    T, F = eng.T, eng.F
    G = R * K
    Y = eng.stft(y.reshape(G, M, L)).reshape(R, K, T, F, M)
    S = eng.stft(s.reshape(G, M, L)).reshape(R, K, T, F, M)
    N = eng.stft(n.reshape(G, M, L)).reshape(R, K, T, F, M)
    Sh, Nh = S.numpy(), N.numpy()
    # masks at the reference mic (step 1, tango.py:338-342: ts = that channel's time signal) and at channel 0 (step 2,
    # tango.py:391-394: ts = s[node][0])
    Yh = Y.numpy() if 'crnn' in vads else None
    mz = _get_mask(eng, Sh[..., ref_mic], Nh[..., ref_mic], s[:, :, ref_mic].reshape(G, L), vads[0], mods[0],
                   None if Yh is None else Yh[..., ref_mic])
    if vads[1] != 'crnn':                                                      # a DNN step-2 mask needs z: after step 1
        same = (vads[1] == vads[0]) and ref_mic == 0
        mw = mz if same else _get_mask(eng, Sh[..., 0], Nh[..., 0], s[:, :, 0].reshape(G, L), vads[1])
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
    out = dict(masks_z=mz, z_y=z_y.numpy(), z_s=z_s.numpy(), z_n=z_n.numpy(), zn=zn.numpy())
    if vads[1] == 'crnn':
        if mods[1] is None:                                                    # same predicted mask as at step 1 (tango.py:388-389)
            mw = mz
        else:                                                                  # CRNN([|Y_k,0| ; |z_j|, j != k]) (tango.py:387-394)
            from ..dnn.crnn import get_z_for_mask
            rows = np.stack([np.stack([get_z_for_mask(out['z_y'][r], out['zn'][r], k, K, z_sigs) for k in range(K)])
                             for r in range(R)]) if K > 1 else None
            mw = _get_mask(eng, None, None, None, 'crnn', mods[1], Yh[..., 0], rows)
    out['mask_w'] = mw
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
            if vads[0] in ('ivad', 'crnn'):
                raise NotImplementedError("mask_for_z='compressed' needs a TF mask type at step 1 (the reference passes neither a time signal nor z to get_mask there, tango.py:403)")
            mc = eng.tf_mask(out['z_s'], out['z_n'], type=vads[0]).numpy()
            zs_rows, zn_rows = zy * mc, zy * (1 - mc)
        elif mask_for_z == 'use_oracle_refs':
            zs_rows, zn_rows = np.ascontiguousarray(Sh[..., ref_mic]), np.ascontiguousarray(Nh[..., ref_mic])
        elif mask_for_z == 'use_oracle_zs':
            zs_rows, zn_rows = out['z_s'], out['z_n']
        else:                                                                  # 'previous': the reference's final else
            zs_rows, zn_rows = zy, zy                                          # (tango.py:428-429), unmasked z_y in both
        eng.cov_masked(Y, mw, zs_rows.astype(np.complex64), zn_rows.astype(np.complex64), mask_remote=False)
    w_glo, _ = eng.gevd_mwf_r1_pending(M + K - 1)
    out['yf'] = eng.apply(Y, w_glo, Z=z_y if K > 1 else None).numpy()
    out['sf'] = eng.apply(S, w_glo, Z=z_s if K > 1 else None).numpy()
    out['nf'] = eng.apply(N, w_glo, Z=z_n if K > 1 else None).numpy()
    return out


def _offline_tango_ragged(y, s, n, vads, mask_for_z, n_fft=N_FFT, pad_mode='reflect', ref_mic=0, mu=1.0):
    """Nodes with DIFFERENT channel counts (the reference allows them: nb_ch is per node, tango.py:259-261, 286).  The
    batched layout is uniform, so every node runs as a one-node shard (`disco_set_node_shard(k, 1)`) of a K-node context
    with ITS OWN mic count; the remote rows of step 2 are the z of all K nodes, exactly the node-sharded data flow.
    Returns per-node lists of (T, F) arrays."""
    vads = _mask_names(vads, [1, 1])
    if 'crnn' in vads:
        raise NotImplementedError('DNN masks with ragged channel counts: run the nodes through disco_amd.dnn.inloop')
    MODES = ('local', None, 'distant', 'compressed', 'use_oracle_refs', 'use_oracle_zs', 'previous')
    if mask_for_z not in MODES:
        raise NotImplementedError(f'mask_for_z must be one of {MODES}')
    oracle_sigs = isinstance(mask_for_z, str) and 'use_oracle_' in mask_for_z
    y = [np.ascontiguousarray(a, dtype=np.float32) for a in y]
    s = [np.ascontiguousarray(a, dtype=np.float32) for a in s]
    n = [np.ascontiguousarray(a, dtype=np.float32) for a in n]
    K, L = len(y), y[0].shape[-1]
    assert all(a.shape[-1] == L for a in y + s + n), 'all channels must have the same number of samples'
    names = ['yf', 'sf', 'nf', 'z_y', 'z_s', 'z_n', 'zn', 'masks_z', 'mask_w']
    out = {nm: [None] * K for nm in names}
    st = []
    engines = set()
    try:
        for k in range(K):                                                     # step 1 (tango.py:326-376)
            M = y[k].shape[0]
            eng = get_engine(rooms=1, nodes=K, mics=M, length=L, n_fft=n_fft, mask=_engine_mask_type(vads[0]), pad_mode=pad_mode,
                             ref_mic=ref_mic, mu=mu, staged_step2=True)
            engines.add(eng)
            eng.set_node_shard(k, 1)
            T, F = eng.T, eng.F
            Y, S, N = (eng.stft(a[None]).reshape(1, 1, T, F, M) for a in (y[k], s[k], n[k]))
            Sh, Nh = S.numpy(), N.numpy()
            mz = _get_mask(eng, Sh[..., ref_mic], Nh[..., ref_mic], s[k][ref_mic][None], vads[0])
            same = (vads[1] == vads[0]) and ref_mic == 0
            mw = mz if same else _get_mask(eng, Sh[..., 0], Nh[..., 0], s[k][0][None], vads[1])
            if oracle_sigs:
                Rss, _ = eng.cov_masked(S, np.ones_like(mz))
                _, Rnn = eng.cov_masked(N, np.zeros_like(mz))
                w_loc, _ = eng.gevd_mwf_r1(Rss, Rnn, want_t1=False)
            else:
                eng.cov_masked(Y, mz)
                w_loc, _ = eng.gevd_mwf_r1_pending(M)
            z_y, z_s, z_n = (eng.apply(A, w_loc).numpy() for A in (Y, S, N))
            zn = eng.noise_residual(Y, z_y).numpy()
            for nm, v in (('z_y', z_y), ('z_s', z_s), ('z_n', z_n), ('zn', zn), ('masks_z', mz), ('mask_w', mw)):
                out[nm][k] = v[0, 0]
            st.append((eng, M, Y.numpy(), Sh, Nh, mw))
        Zy, Zs, Zn, ZN = (np.ascontiguousarray(np.stack(out[nm])[None]) for nm in ('z_y', 'z_s', 'z_n', 'zn'))
        MW = np.stack(out['mask_w'])[None]
        if mask_for_z == 'compressed':                                         # sender-side mask from (z_s, z_n), tango.py:402-405
            if vads[0] == 'ivad':
                raise NotImplementedError("mask_for_z='compressed' with 'ivad': the reference passes no time signal there and fails")
            MC = st[0][0].tf_mask(Zs[0], Zn[0], type=vads[0]).numpy()[None]
        ref_S = np.stack([t[3][0, 0, ..., ref_mic] for t in st])[None]
        ref_N = np.stack([t[4][0, 0, ..., ref_mic] for t in st])[None]
        for k in range(K):                                                     # exchange + step 2 (tango.py:378-450)
            eng, M, Yh, Sh, Nh, mw = st[k]
            eng.set_node_shard(k, 1)
            if mask_for_z == 'local':
                eng.cov_masked(Yh, mw, Zy, Zy, mask_remote=True)
            elif mask_for_z is None:
                eng.cov_masked(Yh, mw, Zy, ZN, mask_remote=False)
            else:
                if mask_for_z == 'distant':
                    zs_rows, zn_rows = Zy * MW, Zy * (1 - MW)
                elif mask_for_z == 'compressed':
                    zs_rows, zn_rows = Zy * MC, Zy * (1 - MC)
                elif mask_for_z == 'use_oracle_refs':
                    zs_rows, zn_rows = ref_S, ref_N
                elif mask_for_z == 'use_oracle_zs':
                    zs_rows, zn_rows = Zs, Zn
                else:                                                          # 'previous' (tango.py:428-429)
                    zs_rows, zn_rows = Zy, Zy
                eng.cov_masked(Yh, mw, np.ascontiguousarray(zs_rows, np.complex64), np.ascontiguousarray(zn_rows, np.complex64),
                               mask_remote=False)
            w_glo, _ = eng.gevd_mwf_r1_pending(M + K - 1)
            for nm, A, Z in (('yf', Yh, Zy), ('sf', Sh, Zs), ('nf', Nh, Zn)):
                out[nm][k] = eng.apply(A, w_glo, Z=Z if K > 1 else None).numpy()[0, 0]
    finally:
        for eng in engines:                                                    # the cached engines go back to "all nodes"
            eng.set_node_shard(0, K)
    return out


def offline_tango(y, s, n, vads='irm1', mods=None, mask_for_z=MASK_Z, z_sigs='zs_hat'):
    """Drop-in for the reference's `offline_tango`: returns
    (yf, sf, nf, z_y, z_s, z_n, zn, masks_z, mask_w), each a list over nodes of (F, T) arrays (tango.py:457).
    Nodes may have different channel counts, as in the reference."""
    names = ['yf', 'sf', 'nf', 'z_y', 'z_s', 'z_n', 'zn', 'masks_z', 'mask_w']
    yb, sb, nb = _as_batch(y), _as_batch(s), _as_batch(n)
    if yb is None or sb is None or nb is None:
        d = _offline_tango_ragged(y, s, n, vads, mask_for_z)                   # (DNN masks: uniform nodes only)
        return tuple([np.ascontiguousarray(v.T) for v in d[nm]] for nm in names)
    d = offline_tango_batched(yb, sb, nb, vads=vads, mods=mods, mask_for_z=mask_for_z, z_sigs=z_sigs)
    K = d['yf'].shape[1]
    return tuple([np.ascontiguousarray(d[nm][0, k].T) for k in range(K)] for nm in names)
