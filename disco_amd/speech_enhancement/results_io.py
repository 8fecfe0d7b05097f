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
