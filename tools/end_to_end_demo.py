#!/usr/bin/env python3
"""The whole chain on the GPU, scored: shoebox rooms -> disco_ism_rir -> dry signals -> disco_rir_convolve -> mixtures at 0-6 dB -> two-step MWF
(reference-output mode: yf, sf, nf) -> iSTFT -> the reference's metrics (disco_amd.metrics) -> z data set on disk.
What the reference does across gen_disco/convolve_signals.py, speech_enhancement/tango.py:main and get_z_signals.py:main,
on synthetic rooms (SURVEY 8d recipe).  Usage: tools/end_to_end_demo.py [rooms] [out_dir]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from disco_amd import metrics as gm
from disco_amd import synth
from disco_amd._engines import get_engine
from disco_amd.speech_enhancement import z_dataset as zd
from disco_amd.speech_enhancement.tango import offline_tango_batched

R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
out_dir = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp(prefix='disco_z_')
K, M, L, FS = 4, 4, 160000, 16000
dev = 'cuda'
torch.manual_seed(0)
t0 = time.perf_counter()
# ---- rooms: shoebox geometry of disco_amd/synth.py (SURVEY 8d ranges), RIRs by the image-source generator on the GPU
# (absorption from Sabine's formula for the drawn RT60, as gen_disco/room_setups.py does for its alpha)
dims = np.empty((R, 3), np.float32); absorb = np.empty(R, np.float32); snr_db = np.empty(R)
srcs = np.empty((R, 2, 3), np.float32); mics = np.empty((R, K * M, 3), np.float32)
for r in range(R):
    rng = np.random.default_rng(1234 + r)
    d3, beta, mic_pos, src_pos = synth._geometry(rng, K, M)
    dims[r], srcs[r], mics[r] = d3, src_pos, mic_pos.reshape(K * M, 3)
    vol, surf = d3.prod(), 2 * (d3[0] * d3[1] + d3[0] * d3[2] + d3[1] * d3[2])
    absorb[r] = min(0.95, 0.161 * vol / (surf * beta))
    snr_db[r] = rng.uniform(0, 6)
eng0 = get_engine(rooms=1, nodes=1, mics=1, length=1024)
rir = torch.from_numpy(eng0.ism_rir(dims, absorb, srcs, mics, max_order=20, rir_len=synth.RIR_TAPS).numpy()).to(dev)   # (R, 2, K*M, taps)
dry = torch.randn((R, 2, L), device=dev)
dry[:, 0] *= np.sqrt(synth.TARGET_VAR)
dry[:, 0, :FS] = 0                                                        # 1 s leading silence of the target
# ---- reverberation: every (room, source) against its K*M impulse responses
img = torch.empty((R * 2, K * M, L), device=dev)
p = lambda t: t.data_ptr()
eng0._chk(eng0.lib.disco_rir_convolve(eng0.ctx, p(dry.reshape(R * 2, L)), p(rir.reshape(R * 2, K * M, synth.RIR_TAPS).contiguous()),
                                      R * 2, K * M, L, synth.RIR_TAPS, p(img), L, None))
img = img.reshape(R, 2, K, M, L)
s_img, n_img = img[:, 0], img[:, 1]
ps = s_img[:, 0, 0, FS:].var(dim=-1); pn = n_img[:, 0, 0, FS:].var(dim=-1)
n_img = n_img * torch.sqrt(ps / (pn * torch.tensor(10 ** (snr_db / 10), device=dev, dtype=torch.float32))).view(R, 1, 1, 1)
y = s_img + n_img
torch.cuda.synchronize()
t_gen = time.perf_counter() - t0
# ---- the path, reference-output mode (filters applied to y, s and n)
t0 = time.perf_counter()
d = offline_tango_batched(y.cpu().numpy(), s_img.cpu().numpy(), n_img.cpu().numpy(), vads='irm1')
eng = get_engine(rooms=R, nodes=K, mics=M, length=L, staged_step2=True)
T, F = eng.T, eng.F
to_time = lambda a: torch.from_numpy(eng.istft(np.ascontiguousarray(a.reshape(R * K, T, F))).numpy()).to(dev)
yf_t, sf_t, nf_t, zs_t, zn_t = (to_time(d[k]) for k in ('yf', 'sf', 'nf', 'z_s', 'z_n'))
torch.cuda.synchronize()
t_path = time.perf_counter() - t0
# ---- scoring as tango.py:541-593 does (span [fs:], reference mic of every node), on the GPU
t0 = time.perf_counter()
s_ref = s_img[:, :, 0].reshape(R * K, L).contiguous(); n_ref = n_img[:, :, 0].reshape(R * K, L).contiguous()
y_ref = y[:, :, 0].reshape(R * K, L).contiguous()
res = {
    'snr_in_dB': gm.snr(s_ref, n_ref, start=FS), 'snr_out_step1_dB': gm.snr(zs_t, zn_t, start=FS),
    'snr_out_step2_dB': gm.snr(sf_t, nf_t, start=FS),
    'delta_snr_step2_dB': gm.delta_snr(sf_t, nf_t, s_ref, n_ref, start=FS),
    'fw_snr_in_dB': gm.fw_snr(s_ref, n_ref, FS, start=FS)[1], 'fw_snr_out_dB': gm.fw_snr(sf_t, nf_t, FS, start=FS)[1],
    'fw_sd_dB': gm.fw_sd(sf_t, s_ref, FS, start=FS)[1],
    'si_sdr_in_dB': gm.si_sdr(s_ref, y_ref, start=FS), 'si_sdr_out_dB': gm.si_sdr(s_ref, yf_t, start=FS),
}
t_score = time.perf_counter() - t0
written = zd.write_z_dataset(out_dir, list(range(11001, 11001 + R)), 'ssn', d['z_y'], d['zn'])
print(json.dumps({'rooms': R, 'nodes': K, 'mics': M, 'seconds': {'generate+reverberate': round(t_gen, 2), 'path(reference mode, host round trips)': round(t_path, 2),
                                                              'score': round(t_score, 3)},
                  'mean_over_rooms_and_nodes': {k: round(float(np.mean(v)), 2) for k, v in res.items()},
                  'z_dataset': {'dir': out_dir, 'rooms_written': len(written)}}))
