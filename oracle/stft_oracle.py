"""NumPy restatement of the librosa STFT / iSTFT the reference calls.

TEST INFRASTRUCTURE (see oracle/__init__.py).

librosa is a third-party dependency of the reference that is neither vendored in
/root/reference nor pinned in its requirements.txt (era-appropriate: 0.7/0.8).
Call sites being restated:
  * disco_theque/speech_enhancement/tango.py:335-337   lb.core.stft(x, n_fft=512, hop_length=256, center=True)
  * disco_theque/speech_enhancement/tango.py:528-539   lb.core.istft(X, hop_length=256, win_length=512, center=True, length=L)
  * disco_theque/math_utils.py:134-152                 my_stft / my_istft (same parameters)

Published algorithm (librosa.core.spectrum.stft / istft):
  stft : window = scipy.signal.get_window('hann', n_fft, fftbins=True) (periodic Hann, float64);
         center=True pads n_fft//2 samples each side (mode 'reflect' for librosa < 0.10,
         'constant' for >= 0.10); frames start every `hop` samples, T = 1 + L // hop;
         X[:, t] = rfft(window * frame_t) (unscaled), stored as complex64.
  istft: frame_t = window * irfft(X[:, t]); overlap-add at t*hop into a float32 buffer of
         n_fft + hop*(T-1) samples; divide by the window sum-of-squares envelope where it exceeds
         `tiny`; drop n_fft//2 leading samples; zero-pad / trim to `length`.
"""
import numpy as np


def hann_periodic(n_fft):
    """scipy.signal.get_window('hann', n_fft, fftbins=True), float64."""
    k = np.arange(n_fft, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n_fft)


def n_frames_of(length, hop):
    """Number of centred frames librosa produces for a `length`-sample signal."""
    return 1 + length // hop


def stft(x, n_fft=512, hop=256, pad_mode='reflect', out_dtype=np.complex64):
    """Centred Hann STFT of a 1-D (or (..., L)) real signal -> (..., F, T).

    out_dtype=complex64 reproduces the reference's storage type (tango.py:335);
    out_dtype=complex128 keeps the float64 intermediate un-rounded (the "exact" oracle).
    """
    x = np.asarray(x)
    lead = x.shape[:-1]
    L = x.shape[-1]
    win = hann_periodic(n_fft)
    half = n_fft // 2
    padw = [(0, 0)] * len(lead) + [(half, half)]
    if pad_mode == 'reflect':
        xp = np.pad(x, padw, mode='reflect')
    elif pad_mode == 'constant':
        xp = np.pad(x, padw, mode='constant')
    else:
        raise ValueError('pad_mode must be "reflect" or "constant"')
    T = n_frames_of(L, hop)
    idx = (np.arange(T) * hop)[:, None] + np.arange(n_fft)[None, :]       # (T, n_fft)
    frames = xp[..., idx].astype(np.float64) * win                        # (..., T, n_fft)
    X = np.fft.rfft(frames, axis=-1)                                      # (..., T, F) complex128
    X = np.swapaxes(X, -1, -2)                                            # (..., F, T)
    return X.astype(out_dtype)


def window_sumsquare(n_frames, n_fft=512, hop=256, dtype=np.float32):
    """librosa.filters.window_sumsquare('hann', n_frames, hop, n_fft, n_fft, norm=None)."""
    n = n_fft + hop * (n_frames - 1)
    env = np.zeros(n, dtype=dtype)
    win_sq = (hann_periodic(n_fft) ** 2).astype(dtype)
    for t in range(n_frames):
        s = t * hop
        env[s:min(n, s + n_fft)] += win_sq[:max(0, min(n_fft, n - s))]
    return env


def istft(X, length, n_fft=512, hop=256, work_dtype=np.float32):
    """Inverse of `stft` (center=True, `length` given) for X of shape (..., F, T) -> (..., length).

    work_dtype=float32 mirrors librosa (output buffer and envelope in float32);
    float64 gives the un-rounded oracle.
    """
    X = np.asarray(X)
    lead = X.shape[:-2]
    F, T = X.shape[-2:]
    assert F == n_fft // 2 + 1
    n_frames = min(T, int(np.ceil((length + n_fft) / hop)))
    win = hann_periodic(n_fft)
    # numpy 1.18 (the reference's pin) runs pocketfft in double for any input; NumPy >= 2 would keep
    # complex64 in single precision, so promote explicitly.
    Xc = np.swapaxes(X[..., :n_frames], -1, -2).astype(np.complex128)
    frames = np.fft.irfft(Xc, n=n_fft, axis=-1) * win                     # (..., T, n_fft)
    n = n_fft + hop * (n_frames - 1)
    y = np.zeros(lead + (n,), dtype=work_dtype)
    for t in range(n_frames):
        y[..., t * hop:t * hop + n_fft] += frames[..., t, :].astype(work_dtype)
    env = window_sumsquare(n_frames, n_fft, hop, dtype=work_dtype)
    nz = env > np.finfo(work_dtype).tiny
    y[..., nz] /= env[nz]
    y = y[..., n_fft // 2:]
    if y.shape[-1] >= length:
        y = y[..., :length]
    else:
        y = np.concatenate([y, np.zeros(lead + (length - y.shape[-1],), dtype=work_dtype)], axis=-1)
    return y
