"""The restated librosa STFT/iSTFT (oracle/stft_oracle.py) against two independent implementations.

librosa is third-party, absent from /root/reference and unpinned there; the reference holds no test or
golden vector for it ("parity unpinned" by the reference).  torch.stft and scipy.signal.stft implement the
same documented transform (periodic Hann, centred, reflect padding) and are used as secondary oracles.
"""
import numpy as np
import pytest
import scipy.signal

from oracle import stft_oracle as so


@pytest.mark.parametrize('L', [4096, 5000, 160000 // 8])
@pytest.mark.parametrize('n_fft,hop', [(512, 256), (1024, 512)])
@pytest.mark.parametrize('pad_mode', ['reflect', 'constant'])
def test_stft_vs_torch(L, n_fft, hop, pad_mode):
    torch = pytest.importorskip('torch')
    rng = np.random.default_rng(L + n_fft)
    x = rng.standard_normal(L).astype(np.float32)
    X = so.stft(x, n_fft, hop, pad_mode, np.complex128)
    Xt = torch.stft(torch.from_numpy(x).double(), n_fft, hop, window=torch.hann_window(n_fft, periodic=True, dtype=torch.float64),
                    center=True, pad_mode=pad_mode, return_complex=True).numpy()
    assert X.shape == Xt.shape == (n_fft // 2 + 1, 1 + L // hop)
    assert np.abs(X - Xt).max() / np.abs(Xt).max() < 1e-12
    X32 = so.stft(x, n_fft, hop, pad_mode)
    assert X32.dtype == np.complex64
    assert np.abs(X32 - Xt).max() / np.abs(Xt).max() < 3e-7


def test_stft_vs_scipy_signal():
    rng = np.random.default_rng(5)
    L, n_fft, hop = 8192, 512, 256
    x = rng.standard_normal(L)
    X = so.stft(x, n_fft, hop, 'reflect', np.complex128)
    _, _, Z = scipy.signal.stft(x, window='hann', nperseg=n_fft, noverlap=n_fft - hop, boundary='even', padded=False)
    Z = Z * so.hann_periodic(n_fft).sum()
    assert Z.shape == X.shape
    assert np.abs(X - Z).max() / np.abs(Z).max() < 1e-12


@pytest.mark.parametrize('L', [4096, 5000, 4097])
@pytest.mark.parametrize('n_fft,hop', [(512, 256), (1024, 512)])
def test_istft_roundtrip_and_torch(L, n_fft, hop):
    torch = pytest.importorskip('torch')
    rng = np.random.default_rng(L)
    x = rng.standard_normal(L).astype(np.float32)
    X = so.stft(x, n_fft, hop)
    xr = so.istft(X, L, n_fft, hop)
    assert xr.dtype == np.float32 and xr.shape == (L,)
    assert np.abs(xr - x).max() < 5e-6
    xt = torch.istft(torch.from_numpy(X), n_fft, hop, window=torch.hann_window(n_fft, periodic=True), center=True, length=L).numpy()
    assert np.abs(xr - xt).max() < 5e-6
    # a modified spectrum (not the STFT of any signal) must also agree with torch
    Xm = X * (0.5 + rng.random(X.shape)).astype(np.float32)
    a = so.istft(Xm, L, n_fft, hop, work_dtype=np.float64)
    b = torch.istft(torch.from_numpy(Xm.astype(np.complex128)), n_fft, hop, window=torch.hann_window(n_fft, periodic=True, dtype=torch.float64),
                    center=True, length=L).numpy()
    assert np.abs(a - b).max() / np.abs(b).max() < 1e-10


def test_stft_batched_leading_dims():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 3, 3000)).astype(np.float32)
    X = so.stft(x)
    assert X.shape == (2, 3, 257, 12)
    assert np.array_equal(X[1, 2], so.stft(x[1, 2]))
    xr = so.istft(X, 3000)
    assert np.abs(xr - x).max() < 5e-6
