"""C4 (CRNN masks in the loop, random weights): why do some rooms of the batch sit far from the float64 oracle?  For a list of rooms: the HIP
path against the float64 oracle fed the SAME masks (bench.py's check), against it the oracle's own sensitivity -- the float64 oracle with every
mask value moved by one float32 rounding (x (1 +- 6e-8)), and the oracle in the reference's own dtype (precision='ref32': complex64 statistics,
the reference's eigh) -- and what the masks look like (share of exactly-saturated values, smallest (1 - m)).  Test / measurement tooling.
Usage: python tools/gpu/exp_c4_conditioning.py out.json rooms=69,43,114,32,82,81,24,0,41 | all [quick=1: no sensitivity runs] [scale=40]"""
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def job(args):
    import numpy as np
    room, yr, sr, nr, got, mz, mw, n_fft, quick = args
    from oracle import stft_oracle as so, tango_oracle as to
    L = yr.shape[-1]
    K = yr.shape[0]
    s = np.zeros_like(yr); n = np.zeros_like(yr)
    s[:, 0] = sr; n[:, 0] = nr

    def run(masks, precision='f64'):
        o = to.offline_tango_vec(yr, s, n, n_fft=n_fft, hop=n_fft // 2, precision=precision, solver='eigh' if precision == 'f64' else 'eig', masks=masks)
        return [so.istft(np.asarray(o['yf'][k]).astype(np.complex128), L, n_fft, n_fft // 2, work_dtype=np.float64) for k in range(K)], o
    masks = ([m for m in mz], [m for m in mw])
    ref, o = run(masks)
    rel = lambda a, b: max(float(np.linalg.norm(a[k] - b[k]) / np.linalg.norm(b[k])) for k in range(K))
    rng = np.random.default_rng(room)
    pert = tuple([np.clip(m.astype(np.float64) * (1.0 + 6e-8 * rng.choice([-1.0, 1.0], size=m.shape)), 0.0, 1.0) for m in ms] for ms in masks)
    out = {'hip_vs_f64_oracle': rel(got, ref)}
    if not quick:
        out['f64_oracle_masks_moved_one_float32_ulp'] = rel(run(pert)[0], ref)
        try:
            out['oracle_in_reference_dtype_vs_f64'] = rel(run(masks, 'ref32')[0], ref)
        except Exception as ex:                                  # (diagnostic only)
            out['oracle_in_reference_dtype_vs_f64'] = repr(ex)
    allm = np.concatenate([np.asarray(m).ravel() for ms in masks for m in ms])
    out['masks'] = {'share_exactly_1': float((allm == 1.0).mean()), 'share_exactly_0': float((allm == 0.0).mean()), 'share_above_0.999': float((allm > 0.999).mean()),
                    'share_below_0.001': float((allm < 0.001).mean()), 'min': float(allm.min()), 'max': float(allm.max())}
    # per (node, bin) of both steps: is there a frame at all in which the noise statistic gets weight?  sum_t (1 - m)^2 against sum_t m^2
    wn = np.stack([((1.0 - np.asarray(m, np.float64)) ** 2).sum(axis=-1) for ms in masks for m in ms])          # (2 K, F) (masks are (F, T)): both steps
    ws = np.stack([(np.asarray(m, np.float64) ** 2).sum(axis=-1) for ms in masks for m in ms])
    out['bins'] = {'min_noise_weight_sum': float(wn.min()), 'min_speech_weight_sum': float(ws.min()), 'bins_noise_weight_below_1e-6': int((wn < 1e-6).sum()),
                         'bins_speech_weight_below_1e-6': int((ws < 1e-6).sum()), 'bins': int(wn.size)}
    return room, out


def main():
    import numpy as np
    import torch
    from disco_amd import synth
    from disco_amd.engine import Engine
    from disco_amd.dnn.crnn import build_crnn
    from disco_amd.dnn.inloop import tango_enhance_dnn
    out_path = sys.argv[1]
    kv = dict(a.split('=') for a in sys.argv[2:])
    rooms = list(range(125)) if kv.get('rooms') == 'all' else [int(x) for x in kv.get('rooms', '69,43,114,32,82,81,24,0,41').split(',')]
    quick = int(kv.get('quick', 0))
    scale, clamp = float(kv.get('scale', 40)), float(kv.get('clamp', 0))
    R, K, M, N, L = 125, 4, 4, 512, 160000
    dev = torch.device('cuda:0')
    eng = Engine(rooms=R, nodes=K, mics=M, length=L, n_fft=N, device=0)
    y, s_ref, n_ref = synth.make_rooms_torch(R, K, M, L, first_room=0, device=dev, ref_only_sn=True)
    torch.manual_seed(0)
    model_z, model_w = build_crnn(1, device=dev), build_crnn(K, device=dev)
    with torch.no_grad():
        for mdl in (model_z, model_w):
            mdl.ff.layers[0].weight.mul_(scale)
    out, mz, mw = tango_enhance_dnn(eng, y, model_z, model_w, want_masks=True)
    torch.cuda.synchronize()
    tr = lambda m, r: [np.ascontiguousarray(m[r, k].cpu().numpy().T) for k in range(K)]        # (T, F) -> (F, T) per node
    jobs = [(r, y[r].cpu().numpy(), s_ref[r].cpu().numpy(), n_ref[r].cpu().numpy(), out[r].cpu().numpy(), tr(mz, r), tr(mw, r), N, quick) for r in rooms]
    res = {'scale': scale, 'rooms': {}}
    with ProcessPoolExecutor(max_workers=min(len(jobs), 32)) as pool:
        for room, o in pool.map(job, jobs):
            res['rooms'][str(room)] = o
            if len(rooms) <= 16:
                print(room, json.dumps(o), flush=True)
    json.dump(res, open(out_path, 'w'), indent=1)


if __name__ == '__main__':
    main()
