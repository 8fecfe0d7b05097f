"""Where does the C5 error of one room come from?  Rooms of a C5-shaped batch against the float64 oracle per node, for 1 and 2 step-2
iterations, for the room alone and inside the batch, for the wide-shape routes."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from disco_amd import _lib, synth
from disco_amd.engine import Engine
from oracle import stft_oracle as so, tango_oracle as to

rooms = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [199, 0]
K, M, N, L = 8, 8, 1024, 160000
lib = _lib.load()
dev = torch.device('cuda', 0)
res = {}
for room in rooms:
    y, s_ref, n_ref = synth.make_rooms_torch(1, K, M, L, first_room=room, device=dev, ref_only_sn=True)
    yh, sh, nh = y[0].cpu().numpy(), s_ref[0].cpu().numpy(), n_ref[0].cpu().numpy()
    s = np.zeros_like(yh); n = np.zeros_like(yh); s[:, 0] = sh; n[:, 0] = nh
    for iters in (1, 2):
        o = to.offline_tango_vec(yh, s, n, vads=['irm1', 'irm1'], n_fft=N, hop=N // 2, precision='f64', solver='eigh', extra_iters=iters - 1)
        refs = [so.istft(o['yf'][k], L, N, N // 2, work_dtype=np.float64) for k in range(K)]
        for route in ('dma', 'staged'):
            eng = Engine(rooms=1, nodes=K, mics=M, length=L, n_fft=N, lib=lib)
            eng.set_option('room_cov', 1 if route == 'dma' else 0)
            m = eng.mask_oracle(sh, nh).reshape(1, K, eng.T, eng.F)
            out, yf = eng.tango_enhance_iterated(yh[None], m, iters=iters)
            out = out.numpy()[0]; yfh = yf.numpy()[0]
            e_t = [float(np.linalg.norm(out[k] - refs[k]) / np.linalg.norm(refs[k])) for k in range(K)]
            e_f = [float(np.linalg.norm(yfh[k].T - o['yf'][k]) / np.linalg.norm(o['yf'][k])) for k in range(K)]
            # per-bin error of the worst node
            kw = int(np.argmax(e_f))
            pb = np.linalg.norm(yfh[kw].T - o['yf'][kw], axis=1) / np.linalg.norm(o['yf'][kw], axis=1)
            top = np.argsort(pb)[::-1][:5]
            Rn = np.asarray(o['Rnn_glo'][kw]); Rs = np.asarray(o['Rss_glo'][kw])
            info = []
            for f in top:
                import scipy.linalg as sl
                d = np.sort(sl.eigh(Rs[f], Rn[f], eigvals_only=True))[::-1]
                info.append((int(f), float(pb[f]), float(np.linalg.cond(Rn[f])), float(d[1] / d[0])))
            res[f'room{room}_it{iters}_{route}'] = {'time_err_per_node': e_t, 'yf_err_per_node': e_f, 'worst_node': kw, 'worst_bins(f,err,condRnn,gap)': info}
            print(room, iters, route, 'max time err %.2e' % max(e_t), 'worst node', kw, info[:3], flush=True)
json.dump(res, open('gpurun_out/dbg_c5_room.json', 'w'), indent=1)
