import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from disco_amd import _lib
from disco_amd.engine import Engine
from oracle import mwf_oracle as mo
import parity_checks as pc
lib = _lib.load()
eng = Engine(lib=lib, rooms=1, nodes=1, mics=1, length=1024)
rng = np.random.default_rng(11)
for P in (2, 4, 7, 15):
    n, T = 64, 6 * P + 5
    a = rng.standard_normal((n, P, 1)) + 1j * rng.standard_normal((n, P, 1))
    X = a * (rng.standard_normal((n, 1, T)) + 1j * rng.standard_normal((n, 1, T))) + 0.3 * (rng.standard_normal((n, P, T)) + 1j * rng.standard_normal((n, P, T)))
    Nn = rng.standard_normal((n, P, T)) + 1j * rng.standard_normal((n, P, T))
    Rxx = (X @ X.conj().transpose(0, 2, 1) / T).astype(np.complex64)
    Rnn = (Nn @ Nn.conj().transpose(0, 2, 1) / T).astype(np.complex64)
    w, t1 = eng.gevd_mwf_r1(Rxx, Rnn)
    wr, t1r, d0 = mo.gevd_mwf_r1_hermitian(Rxx, Rnn, 1.0)
    w, t1 = w.numpy(), t1.numpy()
    e = np.linalg.norm(w - wr, axis=-1) / np.linalg.norm(wr, axis=-1)
    print('P', P, 'worst', e.max(), 'median', np.median(e), 'nan', int(np.isnan(w.view(np.float32)).sum()), 'zero-ish', int((np.linalg.norm(w, axis=-1) < 1e-10).sum()), 't1 err', (np.linalg.norm(t1 - t1r, axis=-1) / np.linalg.norm(t1r, axis=-1)).max())
