"""Seeded synthetic multi-room input generator (stand-in for the reference's pyroomacoustics data).

Follows SURVEY.md section 8(d).  The reference synthesises its rooms with pyroomacoustics'
image-source model (dataset_generation/gen_disco/convolve_signals.py:216-282, room ranges at :361-363,
target level at :405, SNR range tango.py:37); pyroomacoustics is not available here, so the same
parameter ranges drive a cheap statistical RIR: a direct-path tap plus an exponentially decaying
Gaussian tail.  Both the CPU oracle and the HIP path consume the arrays produced here, so parity
never depends on the generator -- only the workload shape does.

Two backends:
  * numpy  (default) -- reproducible per room from `default_rng(seed + room)`; used by tests/oracle.
  * torch  -- same recipe with the heavy parts (noise draws, FFT convolution) on the GPU, used by
    bench.py to build 1000-room batches in seconds.  Different random stream, same distribution.
"""
import numpy as np

FS = 16000
C_SOUND = 343.0
RIR_TAPS = 4096
TARGET_VAR = 10 ** (-23 / 10)          # convolve_signals.py:405
LEAD_SILENCE = FS                      # signal_setups.py:68 (1 s)
MIC_RADIUS = 0.05                      # convolve_signals.py:363 (d_mn)
TAIL_GAIN = 0.02


def _geometry(rng, K, M):
    """Room dims, node centres (>= 0.5 m apart), mic positions, two source positions."""
    dims = np.array([rng.uniform(3, 8), rng.uniform(3, 5), rng.uniform(2.5, 3)])
    beta = rng.uniform(0.3, 0.6)
    centres = []
    while len(centres) < K:
        c = np.array([rng.uniform(0.5, dims[0] - 0.5), rng.uniform(0.5, dims[1] - 0.5), rng.uniform(0.7, 1.5)])
        if all(np.linalg.norm(c - o) >= 0.5 for o in centres) or len(centres) > 50:
            centres.append(c)
    ang = 2 * np.pi * np.arange(M) / M
    ring = np.stack([MIC_RADIUS * np.cos(ang), MIC_RADIUS * np.sin(ang), np.zeros(M)], axis=1)
    mics = np.stack([c + ring for c in centres])                         # (K, M, 3)
    srcs = np.stack([np.array([rng.uniform(0.3, dims[0] - 0.3), rng.uniform(0.3, dims[1] - 0.3),
                               rng.uniform(1.0, 2.0)]) for _ in range(2)])  # target, noise
    return dims, beta, mics, srcs


def _rir_params(rng, K, M):
    dims, beta, mics, srcs = _geometry(rng, K, M)
    dist = np.linalg.norm(mics[None] - srcs[:, None, None, :], axis=-1)  # (2, K, M)
    dist = np.maximum(dist, 0.2)
    delay = np.round(dist / C_SOUND * FS).astype(np.int64)
    return beta, dist, delay


def make_room_numpy(room, K=4, M=4, L=160000, seed=1234):
    """One room -> y, s, n of shape (K, M, L) float32 and the SNR (dB) at node 0 / mic 0."""
    rng = np.random.default_rng(seed + room)
    beta, dist, delay = _rir_params(rng, K, M)
    t = np.arange(RIR_TAPS)
    rir = np.zeros((2, K, M, RIR_TAPS))
    for src in range(2):
        for k in range(K):
            for m in range(M):
                d = int(delay[src, k, m])
                tail = TAIL_GAIN * rng.standard_normal(RIR_TAPS) * np.exp(-6.9 * np.maximum(t - d, 0) / (beta * FS))
                tail[:d + 1] = 0.0
                h = tail
                if d < RIR_TAPS:
                    h[d] = 1.0 / dist[src, k, m]
                rir[src, k, m] = h
    dry_s = np.zeros(L)
    lead = min(LEAD_SILENCE, L // 8)
    dry_s[lead:] = np.sqrt(TARGET_VAR) * rng.standard_normal(L - lead)
    dry_n = rng.standard_normal(L)
    snr_db = rng.uniform(0, 6)
    nfft = 1 << int(np.ceil(np.log2(L + RIR_TAPS)))
    Hf = np.fft.rfft(rir, nfft, axis=-1)
    s_img = np.fft.irfft(np.fft.rfft(dry_s, nfft) * Hf[0], nfft, axis=-1)[..., :L]
    n_img = np.fft.irfft(np.fft.rfft(dry_n, nfft) * Hf[1], nfft, axis=-1)[..., :L]
    ps = np.var(s_img[0, 0, lead:])
    pn = np.var(n_img[0, 0, lead:])
    n_img *= np.sqrt(ps / (pn * 10 ** (snr_db / 10)))
    s32 = s_img.astype(np.float32)
    n32 = n_img.astype(np.float32)
    return s32 + n32, s32, n32, snr_db


def make_rooms_numpy(R, K=4, M=4, L=160000, seed=1234, first_room=0):
    y = np.empty((R, K, M, L), np.float32)
    s = np.empty_like(y)
    n = np.empty_like(y)
    for r in range(R):
        y[r], s[r], n[r], _ = make_room_numpy(first_room + r, K, M, L, seed)
    return y, s, n


def make_rooms_torch(R, K=4, M=4, L=160000, seed=1234, first_room=0, device='cuda', chunk=50,
                     ref_only_sn=True, engine=None):
    """Same recipe on the GPU.  Returns torch tensors y (R,K,M,L) and s, n -- restricted to channel 0
    of every node, shape (R,K,L), when ref_only_sn (all the oracle mask needs; saves 2/3 of the HBM).
    engine: an Engine on the same device -> the source images are formed by the library's own RIR convolution (disco_rir_convolve, the
    kernel that stands in for gen_disco/convolve_signals.py:160-163) instead of torch.fft; the two agree to float32 rounding
    (tests/test_gpu_parity.py::test_synth_rooms_through_rir_convolve)."""
    import torch
    g = torch.Generator(device=device)
    y = torch.empty((R, K, M, L), dtype=torch.float32, device=device)
    sn_shape = (R, K, L) if ref_only_sn else (R, K, M, L)
    s = torch.empty(sn_shape, dtype=torch.float32, device=device)
    n = torch.empty(sn_shape, dtype=torch.float32, device=device)
    nfft = 1 << int(np.ceil(np.log2(L + RIR_TAPS)))
    t = torch.arange(RIR_TAPS, device=device, dtype=torch.float32)
    lead = min(LEAD_SILENCE, L // 8)
    for r0 in range(0, R, chunk):
        rc = min(chunk, R - r0)
        beta = np.empty(rc)
        dist = np.empty((rc, 2, K, M))
        delay = np.empty((rc, 2, K, M), np.int64)
        snr = np.empty(rc)
        for i in range(rc):
            rng = np.random.default_rng(seed + first_room + r0 + i)
            beta[i], dist[i], delay[i] = _rir_params(rng, K, M)
            snr[i] = rng.uniform(0, 6)
        g.manual_seed(seed * 1000003 + first_room + r0)
        beta_t = torch.tensor(beta, device=device, dtype=torch.float32).view(rc, 1, 1, 1, 1)
        dist_t = torch.tensor(dist, device=device, dtype=torch.float32)
        delay_t = torch.tensor(delay, device=device)
        rel = t.view(1, 1, 1, 1, -1) - delay_t.unsqueeze(-1)
        rir = TAIL_GAIN * torch.randn((rc, 2, K, M, RIR_TAPS), device=device, generator=g)
        rir = rir * torch.exp(-6.9 * rel.clamp(min=0) / (beta_t * FS)) * (rel > 0)
        rir.scatter_(-1, delay_t.clamp(max=RIR_TAPS - 1).unsqueeze(-1), (1.0 / dist_t).unsqueeze(-1))
        dry_s = np.sqrt(TARGET_VAR) * torch.randn((rc, L), device=device, generator=g)
        dry_s[:, :lead] = 0
        dry_n = torch.randn((rc, L), device=device, generator=g)
        if engine is not None:
            s_img = torch.empty((rc, K, M, L), dtype=torch.float32, device=device)
            n_img = torch.empty((rc, K, M, L), dtype=torch.float32, device=device)
            for img, dry, src in ((s_img, dry_s, 0), (n_img, dry_n, 1)):
                h = rir[:, src].reshape(rc, K * M, RIR_TAPS).contiguous()
                d = dry.contiguous()
                engine._chk(engine.lib.disco_rir_convolve(engine.ctx, d.data_ptr(), h.data_ptr(), rc, K * M, L, RIR_TAPS, img.data_ptr(), L,
                                                          None))
            torch.cuda.synchronize()
            del rir
        else:
            Hf = torch.fft.rfft(rir, nfft, dim=-1)
            s_img = torch.fft.irfft(torch.fft.rfft(dry_s, nfft).view(rc, 1, 1, -1) * Hf[:, 0], nfft, dim=-1)[..., :L]
            n_img = torch.fft.irfft(torch.fft.rfft(dry_n, nfft).view(rc, 1, 1, -1) * Hf[:, 1], nfft, dim=-1)[..., :L]
            del Hf, rir
        ps = s_img[:, 0, 0, lead:].var(dim=-1)
        pn = n_img[:, 0, 0, lead:].var(dim=-1)
        gain = torch.sqrt(ps / (pn * torch.tensor(10 ** (snr / 10), device=device, dtype=torch.float32)))
        n_img = n_img * gain.view(rc, 1, 1, 1)
        y[r0:r0 + rc] = s_img + n_img
        if ref_only_sn:
            s[r0:r0 + rc] = s_img[:, :, 0]
            n[r0:r0 + rc] = n_img[:, :, 0]
        else:
            s[r0:r0 + rc] = s_img
            n[r0:r0 + rc] = n_img
        del s_img, n_img
    return y, s, n
