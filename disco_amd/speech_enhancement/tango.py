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
    mods = [None, None] if mods is None else list(mods) + [None] * max(0, 2 - len(mods))
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


def _time_mask(eng, vad, s_ch, n_ch, y_ch=None, mod=None, z_rows=None, n_fft=N_FFT, pad_mode='reflect'):
    """A mask the library does not compute by itself inside disco_tango_reference, as a device array (R, K, T, F):
    'ivad' (frame VAD of the target image's time signal, tango.py:217-221), a TF mask of ANOTHER type than the engine's
    (step 2 with vads[1] != vads[0]), or the CRNN's prediction from |STFT(y_ch)| [+ |z| of the other nodes]."""
    R, K, L = s_ch.shape
    if vad == 'ivad':
        return eng.mask_ivad(np.ascontiguousarray(s_ch.reshape(R * K, L))).reshape(R, K, eng.T, eng.F)
    if vad == 'crnn':
        import torch
        Yc = eng.stft(np.ascontiguousarray(y_ch.reshape(R * K, 1, L))).numpy().reshape(R * K, 1, eng.T, eng.F)
        mag = np.abs(Yc)
        if z_rows is not None:
            mag = np.concatenate([mag, np.abs(z_rows).reshape(R * K, -1, eng.T, eng.F)], axis=1)
        par = next(mod.parameters())
        with torch.no_grad():
            m = mod.predict_masks(torch.from_numpy(np.ascontiguousarray(mag)).to(par.device, par.dtype))
        return np.ascontiguousarray(m.float().cpu().numpy().reshape(R, K, eng.T, eng.F))
    other = get_engine(rooms=R, nodes=K, mics=1, length=L, n_fft=n_fft, mask=vad, pad_mode=pad_mode)
    return other.mask_oracle(np.ascontiguousarray(s_ch.reshape(R * K, L)), np.ascontiguousarray(n_ch.reshape(R * K, L))).numpy() \
        .reshape(R, K, eng.T, eng.F)


def offline_tango_batched(y, s, n, vads='irm1', mods=None, mask_for_z=MASK_Z, z_sigs='zs_hat', n_fft=N_FFT,
                          pad_mode='reflect', ref_mic=0, mu=1.0, steps=3):
    """y, s, n: (R, K, M, L) float32.  Returns a dict of device-computed arrays with a leading room axis, in the
    engine's frame-major layout (R, K, T, F): yf, sf, nf, z_y, z_s, z_n, zn, masks_z, mask_w (steps=1: the step-1 five).
    The whole path is ONE library call (disco_tango_reference: STFTs, masks, statistics, solves, every mask_for_z variant and
    the three filter passes stay on the device); only 'ivad' / DNN masks are prepared outside it and passed in."""
    vads = _mask_names(vads, mods)
    mods = [None, None] if mods is None else list(mods) + [None] * (2 - len(mods))
    MODES = ('local', None, 'distant', 'compressed', 'use_oracle_refs', 'use_oracle_zs', 'previous')
    if mask_for_z not in MODES:
        raise NotImplementedError(f'mask_for_z must be one of {MODES}')       # 'use_oracle_sigs' is broken in the reference
    y = np.ascontiguousarray(y, dtype=np.float32)
    s = np.ascontiguousarray(s, dtype=np.float32)
    n = np.ascontiguousarray(n, dtype=np.float32)
    R, K, M, L = y.shape
    if mask_for_z == 'compressed' and vads[0] in ('ivad', 'crnn') and steps & 2:
        raise NotImplementedError("mask_for_z='compressed' needs a TF mask type at step 1 (the reference passes neither a time signal nor z to get_mask there, tango.py:403)")
    eng = get_engine(rooms=R, nodes=K, mics=M, length=L, n_fft=n_fft, mask=_engine_mask_type(vads[0]), pad_mode=pad_mode,
                     ref_mic=ref_mic, mu=mu, staged_step2=True)
    tf = lambda v: v[:-1] in ('irm', 'ibm', 'iam')
    # masks the library cannot derive from (S, N) with the engine's own TF type (see _time_mask); None = computed inside
    mz = None if tf(vads[0]) else _time_mask(eng, vads[0], s[:, :, ref_mic], n[:, :, ref_mic], y[:, :, ref_mic], mods[0],
                                             n_fft=n_fft, pad_mode=pad_mode)
    yd, sd, nd = eng.to_device(y, np.float32)[1], eng.to_device(s, np.float32)[1], eng.to_device(n, np.float32)[1]
    need_z_for_mw = vads[1] == 'crnn' and mods[1] is not None
    if steps == 1 or need_z_for_mw:
        out = {k: v.numpy() for k, v in eng.tango_reference(yd, sd, nd, mask_z=mz, mask_for_z=mask_for_z, steps=1).items()}
        if steps == 1:
            return out
    if vads[1] == 'crnn':
        if mods[1] is None:                                                    # same predicted mask as at step 1 (tango.py:388-389)
            mw = mz
        else:                                                                  # CRNN([|Y_k,0| ; |z_j|, j != k]) (tango.py:387-394)
            from ..dnn.crnn import get_z_for_mask
            rows = np.stack([np.stack([get_z_for_mask(out['z_y'][r], out['zn'][r], k, K, z_sigs) for k in range(K)])
                             for r in range(R)]) if K > 1 else None
            mw = _time_mask(eng, 'crnn', s[:, :, 0], n[:, :, 0], y[:, :, 0], mods[1], rows, n_fft=n_fft, pad_mode=pad_mode)
    elif tf(vads[1]) and tf(vads[0]) and vads[1] == vads[0]:
        mw = None                                                              # the library's own tf_mask at channel 0
    elif vads[1] == vads[0] and ref_mic == 0:
        mw = mz                                                                # 'ivad' twice on the same channel
    else:
        mw = _time_mask(eng, vads[1], s[:, :, 0], n[:, :, 0], n_fft=n_fft, pad_mode=pad_mode)
    if need_z_for_mw:
        out.update({k: v.numpy() for k, v in eng.tango_reference(yd, sd, nd, mask_z=mz, mask_w=mw, mask_for_z=mask_for_z, steps=2).items()})
        return out
    return {k: v.numpy() for k, v in eng.tango_reference(yd, sd, nd, mask_z=mz, mask_w=mw, mask_for_z=mask_for_z, steps=3).items()}


def _offline_tango_ragged(y, s, n, vads, mask_for_z, n_fft=N_FFT, pad_mode='reflect', ref_mic=0, mu=1.0, steps=3):
    """Nodes with DIFFERENT channel counts (the reference allows them: nb_ch is per node, tango.py:259-261, 286).  The
    batched layout is uniform, so every node runs as a one-node shard (`disco_set_node_shard(k, 1)`) of a K-node context
    with ITS OWN mic count; the remote rows of step 2 are the z of all K nodes, exactly the node-sharded data flow.
    steps = 1: stop after step 1 (get_z_signals.py:213-317) -- no step-2 mask, statistics, solves or filter passes, and none of
    step 2's restrictions.  Returns per-node lists of (T, F) arrays."""
    vads = _mask_names(vads, [1, 1])
    if steps == 1:
        vads = [vads[0], vads[0]]
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
            mw = mz if (same or steps == 1) else _get_mask(eng, Sh[..., 0], Nh[..., 0], s[k][0][None], vads[1])
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
        if steps == 1:
            return out
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
