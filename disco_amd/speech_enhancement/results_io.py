"""The audio / mask files the reference writes after the path (disco_theque/speech_enhancement/tango.py:595-613), for batches:

    <root>/WAV/<i_rir>/in_mix-<noise>_Node-<k>.wav      y[k][0]          mixture at the node's first microphone
    <root>/WAV/<i_rir>/out_mix-<noise>_Node-<k>.wav     iSTFT(yf)        step-2 output
    <root>/WAV/<i_rir>/mid_z-<noise>_Node-<k>.wav       iSTFT(z_y)       compressed signal after step 1
    <root>/WAV/<i_rir>/in_noi- / out_noi- / in_tar- / out_tar-<noise>_Node-<k>.wav     n[k][0], iSTFT(nf), s[k][0], iSTFT(sf)
    <root>/MASK/<i_rir>/step1_<noise>_Node-<k>.npy, step2_...npy                        masks_z, mask_w   (F, T)

The reference writes the audio with `soundfile.write(path, data, fs)` (third-party, absent): for a '.wav' path and float
input that is 16-bit PCM, full scale at +-1.0.  Here the standard library's `wave` module writes the same container
(samples = round(clip(x, -1, 1 - 2^-15) * 32768)); bit-level equality with libsndfile's rounding is not pinned.
"""
import os
import wave

import numpy as np

FS = 16000
WAV_KINDS = ('in_mix', 'out_mix', 'mid_z', 'in_noi', 'out_noi', 'in_tar', 'out_tar')


def write_wav(path, x, fs=FS):
    """float signal in [-1, 1) -> 16-bit PCM mono WAV."""
    x = np.asarray(x, dtype=np.float64)
    pcm = np.round(np.clip(x, -1.0, 1.0 - 2.0 ** -15) * 32768.0).astype('<i2')
    with wave.open(path, 'wb') as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(int(fs))
        f.writeframes(pcm.tobytes())


def read_wav(path):
    with wave.open(path, 'rb') as f:
        assert f.getnchannels() == 1 and f.getsampwidth() == 2
        fs = f.getframerate()
        x = np.frombuffer(f.readframes(f.getnframes()), dtype='<i2').astype(np.float32) / 32768.0
    return x, fs


def write_room_results(root, i_rir, noise, signals, masks_z=None, mask_w=None, fs=FS):
    """signals: dict kind -> (K, L) array for kinds of WAV_KINDS (missing kinds are skipped); masks (K, T, F) engine layout
    (written transposed, (F, T), as the reference holds them).  Returns the list of files written."""
    wav_dir = os.path.join(root, 'WAV', str(i_rir))
    os.makedirs(wav_dir, exist_ok=True)
    written = []
    for kind in WAV_KINDS:
        if kind not in signals:
            continue
        arr = np.asarray(signals[kind])
        for k in range(arr.shape[0]):
            p = os.path.join(wav_dir, '{}-{}_Node-{}.wav'.format(kind, noise, k + 1))
            write_wav(p, arr[k], fs)
            written.append(p)
    if masks_z is not None or mask_w is not None:
        mdir = os.path.join(root, 'MASK', str(i_rir))
        os.makedirs(mdir, exist_ok=True)
        for name, m in (('step1', masks_z), ('step2', mask_w)):
            if m is None:
                continue
            m = np.asarray(m)
            for k in range(m.shape[0]):
                p = os.path.join(mdir, '{}_{}_Node-{}.npy'.format(name, noise, k + 1))
                np.save(p, np.ascontiguousarray(m[k].T))
                written.append(p)
    return written


# ---- the result pickles (tango.py:617-635) ------------------------------------------------------------------------------------
# Two dictionaries of per-node arrays per (room, noise): `results_tango_<rir>_<noise>.p` (step-2 output) and
# `results_mwf_<rir>_<noise>.p` (the compressed signal after step 1), same keys as the reference.  The level metrics (fw_snr,
# fw_sd: the reference's own disco_theque/metrics.py) are computed by disco_amd.metrics on the GPU; the keys the reference fills
# from mir_eval.separation.bss_eval_sources and pystoi.stoi (third-party, absent here) are present and hold NaN, so that code
# reading the pickles finds every key it expects.
RESULT_KEYS_TANGO = ('snr_in_raw', 'sar_cnv', 'sir_cnv', 'sdr_cnv', 'delta_stoi_cnv', 'delta_stoi_dry', 'snr_out', 'snr_in_cnv',
                     'snr_in_dry', 'fw_sd_cnv', 'fw_sd_dry', 'sar_dry', 'sir_dry', 'sdr_dry', 'sdr_in_cnv', 'sir_in_cnv',
                     'sdr_in_dry', 'sir_in_dry', 'sar_in_dry')
RESULT_KEYS_MWF = tuple('delta_stoi' if k == 'delta_stoi_cnv' else k for k in RESULT_KEYS_TANGO)
THIRD_PARTY_KEYS = ('sar_cnv', 'sir_cnv', 'sdr_cnv', 'delta_stoi_cnv', 'delta_stoi', 'delta_stoi_dry', 'sar_dry', 'sir_dry', 'sdr_dry',
                    'sdr_in_cnv', 'sir_in_cnv', 'sdr_in_dry', 'sir_in_dry', 'sar_in_dry')


def room_results(s_in, n_in, sf_t, nf_t, szf_t, nzf_t, rnd_snrs, s_dry=None, n_dry=None, fs=FS):
    """The two result dictionaries of one (room, noise) (tango.py:541-635, the parts that are the reference's own code).
    s_in, n_in (K, L): target / noise images at every node's first microphone; sf_t, nf_t (K, L): their step-2 outputs in time;
    szf_t, nzf_t (K, L): their compressed (step-1) versions in time; rnd_snrs: the drawn input SNRs; s_dry, n_dry (L,): the dry
    sources, or None (the `_dry` level metrics are then NaN too).  The first second is skipped as in the reference ([fs:])."""
    from .. import metrics as dm
    K = np.shape(s_in)[0]
    L = min(np.shape(a)[-1] for a in (s_in, n_in, sf_t, nf_t, szf_t, nzf_t) if a is not None)
    if s_dry is not None:
        L = min(L, len(s_dry), len(n_dry))
    cut = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float32)[..., fs:L])
    nan = np.full(K, np.nan)
    res = {k: nan.copy() for k in RESULT_KEYS_TANGO}
    resz = {k: nan.copy() for k in RESULT_KEYS_MWF}
    res['snr_in_raw'] = resz['snr_in_raw'] = rnd_snrs
    snr_in = np.asarray(dm.fw_snr(cut(s_in), cut(n_in), fs)[1])                        # tango.py:581
    res['snr_in_cnv'] = resz['snr_in_cnv'] = snr_in
    res['snr_out'] = np.asarray(dm.fw_snr(cut(sf_t), cut(nf_t), fs)[1])                # :580
    resz['snr_out'] = np.asarray(dm.fw_snr(cut(szf_t), cut(nzf_t), fs)[1])             # :584
    res['fw_sd_cnv'] = np.asarray(dm.fw_sd(cut(sf_t), cut(s_in), fs)[1])               # :590
    resz['fw_sd_cnv'] = np.asarray(dm.fw_sd(cut(szf_t), cut(s_in), fs)[1])             # :592
    if s_dry is not None:
        sd_, nd_ = cut(s_dry)[None], cut(n_dry)[None]
        dry = float(np.asarray(dm.fw_snr(sd_, nd_, fs)[1]).reshape(-1)[0])
        res['snr_in_dry'] = resz['snr_in_dry'] = np.full(K, dry)                       # :582, 586
        rep = np.repeat(sd_, K, axis=0)
        res['fw_sd_dry'] = np.asarray(dm.fw_sd(cut(sf_t), rep, fs)[1])                 # :591
        resz['fw_sd_dry'] = np.asarray(dm.fw_sd(cut(szf_t), rep, fs)[1])               # :593
    return res, resz


def write_result_pickles(root, i_rir, noise, res, resz):
    """<root>/OIM/results_tango_<i_rir>_<noise>.p and results_mwf_... (tango.py:634-635)."""
    import pickle
    d = os.path.join(root, 'OIM')
    os.makedirs(d, exist_ok=True)
    files = []
    for name, r in (('tango', res), ('mwf', resz)):
        p = os.path.join(d, 'results_{}_{}_{}.p'.format(name, i_rir, noise))
        with open(p, 'wb') as f:
            pickle.dump(r, f)
        files.append(p)
    return files
