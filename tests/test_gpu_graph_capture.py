"""The whole path as ONE hipGraph launch (include/disco_hip.h: disco_reserve).

After disco_reserve the whole-path entry points neither allocate nor synchronise: a call is a fixed sequence of kernel launches on
the caller's stream.  Here disco_mask_oracle + disco_tango_enhance (and the iterated scheme) are captured on a side stream with
torch.cuda.graph (hipStreamBeginCapture underneath), replayed on fresh inputs, and compared BIT FOR BIT with the eager calls and
at 1e-4 with the float64 oracle."""
import numpy as np
import pytest

import parity_checks as pc
from disco_amd import _lib, synth
from disco_amd.engine import Engine

pytestmark = pytest.mark.gpu


def _rooms(R, K, M, L, first_room):
    y, s, n = synth.make_rooms_numpy(R, K=K, M=M, L=L, first_room=first_room)
    return y.astype(np.float32), s.astype(np.float32), n.astype(np.float32)


def _oracle_time_outputs(y, s, n, n_fft, iters):
    """float64 oracle of the enhanced time signals of one room: (K, L)."""
    from oracle import stft_oracle as so
    from oracle import tango_oracle as to
    o = to.offline_tango_vec(y, s, n, vads=['irm1', 'irm1'], n_fft=n_fft, hop=n_fft // 2, precision='f64', solver='eigh',
                             extra_iters=iters - 1)
    return [so.istft(o['yf'][k], y.shape[-1], n_fft, n_fft // 2, work_dtype=np.float64) for k in range(y.shape[0])]


@pytest.mark.timeout(600)
@pytest.mark.parametrize('K,M,n_fft,iters,overlap', [(4, 4, 512, 1, 0), (1, 4, 512, 1, 0), (3, 2, 1024, 2, 0), (4, 4, 512, 1, 2), (3, 2, 1024, 2, 2)])
def test_whole_path_replayed_from_a_hip_graph(K, M, n_fft, iters, overlap):
    """overlap = 2: the overlapped form (two half-batches, the second on the context's side stream forked from / joined to the
    capturing stream with events) must capture and replay as well."""
    import torch
    lib = _lib.load()
    torch.cuda.set_device(0)
    dev = torch.device('cuda:0')
    R, L = 3, 24000
    eng = Engine(rooms=R, nodes=K, mics=M, length=L, n_fft=n_fft, lib=lib)
    eng.set_option('overlap_solves', overlap)
    eng.reserve(1)
    T, F, G = eng.T, eng.F, R * K
    y = torch.empty((R, K, M, L), dtype=torch.float32, device=dev)
    s_ref = torch.empty((R, K, L), dtype=torch.float32, device=dev)
    n_ref = torch.empty((R, K, L), dtype=torch.float32, device=dev)
    mask = torch.empty((R, K, T, F), dtype=torch.float32, device=dev)
    out = torch.empty((R, K, L), dtype=torch.float32, device=dev)

    def load(first_room):
        yh, sh, nh = _rooms(R, K, M, L, first_room)
        y.copy_(torch.from_numpy(yh))
        s_ref.copy_(torch.from_numpy(np.ascontiguousarray(sh[:, :, 0])))
        n_ref.copy_(torch.from_numpy(np.ascontiguousarray(nh[:, :, 0])))
        return yh, sh, nh

    def launch(stream):
        eng._chk(lib.disco_mask_oracle(eng.ctx, s_ref.data_ptr(), n_ref.data_ptr(), G, mask.data_ptr(), stream))
        if iters > 1:
            eng._chk(lib.disco_tango_enhance_iterated(eng.ctx, y.data_ptr(), mask.data_ptr(), mask.data_ptr(), iters, out.data_ptr(),
                                                      None, None, None, 0, stream))
        else:
            eng._chk(lib.disco_tango_enhance(eng.ctx, y.data_ptr(), mask.data_ptr(), mask.data_ptr(), out.data_ptr(), None, None,
                                             None, 0, stream))

    load(100)
    launch(None)                       # eager warm-up on the null stream (nothing left to allocate afterwards either way)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        launch(side.cuda_stream)
    torch.cuda.synchronize()

    for first_room in (7, 31):
        yh, sh, nh = load(first_room)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        got = out.cpu().numpy().copy()
        launch(None)
        torch.cuda.synchronize()
        eager = out.cpu().numpy()
        assert np.array_equal(got, eager), 'graph replay differs from the eager launch sequence'
        for r in range(R):
            ref = _oracle_time_outputs(yh[r], sh[r], nh[r], n_fft, iters)
            for k in range(K):
                assert pc.relerr(got[r, k], ref[k]) < 1e-4, (first_room, r, k)
