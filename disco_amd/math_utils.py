"""`my_stft` / `my_istft` with the reference's signatures (disco_theque/math_utils.py:134-152), on the MI355X."""
import numpy as np

from ._engines import get_engine

N_FFT, N_HOP = 512, 256


def my_stft(x):
    """librosa STFT with the reference's parameters (n_fft=512, hop=256, center=True).  x: (L,) -> (257, T) complex64."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if x.ndim != 1:
        raise ValueError('my_stft expects a 1-D time signal')
    eng = get_engine(rooms=1, nodes=1, mics=1, length=x.shape[0], n_fft=N_FFT)
    return eng.stft(x[None, None, :]).numpy()[0, :, :, 0].T.copy()


def my_istft(y, out_len):
    """librosa iSTFT with the reference's parameters.  y: (257, T) complex -> (out_len,) float32."""
    y = np.asarray(y)
    eng = get_engine(rooms=1, nodes=1, mics=1, length=int(out_len), n_fft=N_FFT)
    if y.shape != (eng.F, eng.T):
        raise ValueError(f'my_istft: expected a ({eng.F}, {eng.T}) STFT for out_len={out_len}, got {y.shape}')
    return eng.istft(np.ascontiguousarray(y.T[None], dtype=np.complex64)).numpy()[0]
