#!/usr/bin/env python3
"""Benchmark of the MI355X MWF hot path (driver contract: one JSON line on rank 0).

Workload (BASELINE.json configs[2], "C3"): `--rooms` concurrent rooms per GPU (default 1000), 4 nodes x 4 mics,
16 kHz, 10 s clips (L = 160000), 512-pt STFT / hop 256, oracle IRM mask, two-step Tango with the z exchange
kept on the GPU (all nodes of a room live on one device).  One "step" = the whole path over the
whole batch: oracle mask (2 STFTs/node) -> STFT -> covariance -> GEVD-MWF solve -> z -> exchange -> covariance
-> solve -> filter -> iSTFT, inputs and outputs resident in HBM.  metric = node-frames/s (1 node-frame = one
hop of all M mics of one node); x real-time = audio seconds per room / seconds per step.

Multi-GPU: rooms shard across ranks with no data-path collective (weak scaling: `--rooms` per GPU); the only
communication is the timing barrier / max-reduce the contract asks for.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK = 8.0e12          # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured achievable)


def b_alg(M, K, F, H):
    """algorithmic bytes per node-frame of the whole path ('enhanced' outputs)."""
    if K == 1:
        return 8 * M * H + 4 * F + 4 * H
    return 16 * M * H + 8 * F + 16 * (K - 1) * F + 8 * F + 4 * H


def kernel_alg_bytes(M, K, F, H):
    """Compulsory bytes per node-frame of each stage given its C-ABI contract (inputs once + outputs once);
    DESIGN.md section 'Kernels' derives them."""
    P2 = M + K - 1
    return {
        'mask_oracle': 2 * H * 4 + F * 4,                     # s_ref, n_ref hop samples in, mask out
        'stft': M * H * 4 + M * F * 8,                        # hop samples of M mics in, M*F bins out
        'stft_cov1': M * H * 4 + M * F * 8 + F * 4,           # samples + mask in, X out (covariances amortised over T)
        'cov1': M * F * 8 + F * 4,                            # X + mask in (covariances: amortised over T)
        'apply1': M * F * 8 + F * 8,                          # X in, z out
        'cov2': M * F * 8 + F * 4 + (K - 1) * F * 8,          # X + mask + remote z in
        'apply2': M * F * 8 + (K - 1) * F * 8 + F * 8,        # X + remote z in, yf out
        'step2_cov': M * F * 8 + F * 4,                       # X + mask in (z stays on chip)
        'step2_apply': M * F * 8 + F * 8,                     # X in, yf out
        'step2_apply_istft': M * F * 8 + H * 4,               # X in, hop samples out (yf stays on chip)
        'istft': F * 8 + H * 4,                               # yf in, hop samples out
    }


_WORKER = r"""
import sys, time
sys.path.insert(0, sys.argv[1])
room, K, M, L, start_at = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])
from disco_amd import synth
from oracle import tango_oracle as to
y, s, n, _ = synth.make_room_numpy(room, K=K, M=M, L=L)
late = time.time() > start_at
while time.time() < start_at:
    time.sleep(0.005)
t0 = time.time()
to.offline_tango_vec(y, s, n, vads=['irm1', 'irm1'], precision='f64', solver='eigh')
print(t0, time.time(), int(late))
"""


def cpu_vectorised_all_cores(K, M, L, hop=256, lead_seconds=20.0, limit_seconds=90.0):
    """the vectorised oracle on every host core at once -- one single-threaded process per core, one room
    each, all released at the same wall-clock instant; rate = rooms done / (last finish - common start).  Plain
    subprocesses with a hard time limit: a reported baseline must never be able to hang or fail the bench."""
    import subprocess
    procs = os.cpu_count() or 1
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
    start_at = time.time() + lead_seconds
    ps = []
    try:
        for r in range(procs):
            ps.append(subprocess.Popen([sys.executable, '-c', _WORKER, REPO, str(r), str(K), str(M), str(L), repr(start_at)],
                                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True))
        ends, late = [], 0
        for p in ps:
            out, _ = p.communicate(timeout=max(1.0, start_at + limit_seconds - time.time()))
            t0, t1, lt = out.split()
            ends.append(float(t1))
            late += int(lt)
        if late:
            return {'error': f'{late} of {procs} workers were not ready at the common start'}
        wall = max(ends) - start_at
    except Exception as e:
        for p in ps:
            if p.poll() is None:
                p.kill()
        return {'error': repr(e)}
    T = 1 + L // hop
    return {'value': procs * K * T / wall, 'unit': 'node-frames/s', 'cores': procs, 'seconds': round(wall, 2),
            'what': f'{procs} processes x 1 room each (OMP_NUM_THREADS=1), released together, '
                    'oracle/tango_oracle.py:offline_tango_vec'}


def cpu_baseline(K, M, L, seconds_hint=25.0):
    """The reference's CPU path (literal loop nest, oracle/tango_oracle.py:offline_tango_literal -- pinned
    bit-exact against the reference's own code) on ONE room of the same workload, one host core."""
    import numpy as np
    from disco_amd import synth
    from oracle import tango_oracle as to
    y, s, n, _ = synth.make_room_numpy(0, K=K, M=M, L=L)
    t0 = time.perf_counter()
    to.offline_tango_literal(y, s, n, vads=['irm1', 'irm1'])
    dt = time.perf_counter() - t0
    T = 1 + L // 256
    # the vectorised NumPy restatement of the same path (float64, batched eigh), same room, one process
    t0 = time.perf_counter()
    to.offline_tango_vec(y, s, n, vads=['irm1', 'irm1'], precision='f64', solver='eigh')
    dtv = time.perf_counter() - t0
    return {'value': K * T / dt, 'unit': 'node-frames/s', 'cores': 1, 'kind': 'port',
            'vectorised_numpy': {'value': K * T / dtv, 'unit': 'node-frames/s', 'seconds': round(dtv, 2),
                                 'what': 'oracle/tango_oracle.py:offline_tango_vec (float64, einsum covariances, batched eigh), same room'},
            'vectorised_numpy_all_cores': cpu_vectorised_all_cores(K, M, L),
            'sample': f'1 room ({K} nodes x {M} mics, {L} samples = {K * T} node-frames), literal reference loop nest '
                      f'(tango.py:326-457 restated, bit-exact vs reference), STFTs included, {dt:.1f} s on 1 of '
                      f'{os.cpu_count()} host cores',
            'x_realtime': (L / 16000.0) / dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--rooms', type=int, default=1000, help='rooms per GPU')
    ap.add_argument('--nodes', type=int, default=4)
    ap.add_argument('--mics', type=int, default=4)
    ap.add_argument('--length', type=int, default=160000)
    ap.add_argument('--n-fft', type=int, default=512)
    ap.add_argument('--mask', default='oracle', choices=['oracle', 'crnn'],
                    help="'crnn': BASELINE configs[3] -- randomly initialised CRNN mask estimators (PyTorch-ROCm) in the loop")
    ap.add_argument('--online-every', type=int, default=0,
                    help='> 0: time the ONLINE pipeline (SURVEY 8f-2) with a filter update every this many frames instead of '
                         'the batch path (not the headline metric; stage timing / roofline are skipped)')
    ap.add_argument('--iters', type=int, default=1,
                    help='> 1: the DANSE-style iterated scheme (BASELINE configs[4]; disco_tango_enhance_iterated, staged kernels)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-stage-timing', action='store_true')
    ap.add_argument('--pmc-calibrate', action='store_true', help='also run a 4 GiB device copy (known bytes) for PMC calibration')
    args = ap.parse_args()

    import numpy as np
    import torch
    from disco_amd import _lib, synth
    from disco_amd.engine import Engine

    from disco_amd import dist as dd
    rank, world, local_rank = dd.env_rank_world()
    if args.gpus != world and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE {world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: there is no CPU path')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        dist = dd.init('nccl', rank, world, device=dev)          # RCCL; used for the timing contract only

    R, K, M, Ls, N = args.rooms, args.nodes, args.mics, args.length, args.n_fft
    H, F = N // 2, N // 2 + 1
    lib = _lib.load()
    eng = Engine(rooms=R, nodes=K, mics=M, length=Ls, n_fft=N, device=local_rank, lib=lib)
    T = eng.T
    assert torch.cuda.current_stream().cuda_stream == 0, 'bench times the null stream the library launches on'

    # synthetic rooms, generated on the GPU (SURVEY 8d recipe); rank r owns rooms [r*R, (r+1)*R)
    first_room, _ = dd.room_range(rank, world, R)
    y, s_ref, n_ref = synth.make_rooms_torch(R, K, M, Ls, first_room=first_room, device=dev, ref_only_sn=True)
    mask = torch.empty((R, K, T, F), dtype=torch.float32, device=dev)
    out = torch.empty((R, K, Ls), dtype=torch.float32, device=dev)
    ws = torch.empty(eng.workspace_bytes(), dtype=torch.uint8, device=dev)
    G = R * K

    if args.mask == 'crnn':
        from disco_amd.dnn.crnn import build_crnn
        from disco_amd.dnn.inloop import tango_enhance_dnn
        torch.manual_seed(0)
        model_z = build_crnn(1, device=dev)
        model_w = build_crnn(K, device=dev) if K > 1 else None

    def step():
        if args.mask == 'crnn':
            out.copy_(tango_enhance_dnn(eng, y, model_z, model_w))
            return
        eng._chk(lib.disco_mask_oracle(eng.ctx, s_ref.data_ptr(), n_ref.data_ptr(), G, mask.data_ptr(), None))
        if args.online_every > 0:
            eng._chk(lib.disco_tango_online(eng.ctx, y.data_ptr(), mask.data_ptr(), mask.data_ptr(), 0.95, args.online_every,
                                            1e-3, out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), None))
            return
        if args.iters > 1:
            eng._chk(lib.disco_tango_enhance_iterated(eng.ctx, y.data_ptr(), mask.data_ptr(), mask.data_ptr(), args.iters,
                                                      out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), None))
            return
        eng._chk(lib.disco_tango_enhance(eng.ctx, y.data_ptr(), mask.data_ptr(), mask.data_ptr(), out.data_ptr(),
                                         None, None, ws.data_ptr(), ws.numel(), None))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.pmc_calibrate:
        src = torch.empty(1 << 30, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        dst.copy_(src)                      # 4 GiB read + 4 GiB written by one elementwise copy kernel
        torch.cuda.synchronize()
        del src, dst
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    value, dt = dd.whole_job_throughput(R * K * T * args.steps, dt, world, device=dev)
    assert bool(torch.isfinite(out).all())
    ms_per_step = 1e3 * dt / args.steps
    x_rt = (Ls / 16000.0) / (dt / args.steps)

    # ---- per-stage timing with HIP events on the launch stream (rank 0, N=1), for the roofline object
    roofline, stages = None, None
    if rank == 0 and not args.no_stage_timing and args.mask == 'oracle' and args.online_every == 0 and args.iters == 1:
        X = torch.empty((R, K, T, F, M), dtype=torch.complex64, device=dev)
        z = torch.empty((R, K, T, F), dtype=torch.complex64, device=dev)
        yf = torch.empty_like(z)
        P2 = M + K - 1
        Rss = torch.empty((R, K, F, P2, P2), dtype=torch.complex64, device=dev)
        Rnn = torch.empty_like(Rss)
        w = torch.empty((R, K, F, P2), dtype=torch.complex64, device=dev)
        p = lambda t: t.data_ptr()
        w2 = torch.empty_like(w)
        NUL = None
        # exactly the launches disco_tango_enhance makes, one stage per call (covariances stay as partial sums in
        # the context and feed the solver directly, as in the fused path)
        calls = [
            ('mask_oracle', lambda: lib.disco_mask_oracle(eng.ctx, p(s_ref), p(n_ref), G, p(mask), None)),
            ('stft_cov1', lambda: lib.disco_stft_cov_fused(eng.ctx, p(y), p(mask), p(X), NUL, NUL, None)),
            ('solve1', lambda: lib.disco_gevd_mwf_r1_pending(eng.ctx, 1.0, p(w), NUL, None)),
        ]
        if K > 1:
            calls += [
                ('step2_cov', lambda: lib.disco_step2_cov_fused_reuse(eng.ctx, p(X), p(mask), p(w), NUL, None)),
                ('solve2', lambda: lib.disco_gevd_mwf_r1_pending(eng.ctx, 1.0, p(w2), NUL, None)),
            ]
            if N == 512:
                calls += [('step2_apply_istft', lambda: lib.disco_step2_apply_istft_fused(eng.ctx, p(X), p(w), p(w2), p(out), None))]
            else:
                calls += [('step2_apply', lambda: lib.disco_step2_apply_fused(eng.ctx, p(X), p(w), p(w2), NUL, p(yf), None)),
                          ('istft', lambda: lib.disco_istft(eng.ctx, p(yf), G, p(out), None))]
        else:
            calls += [('apply1', lambda: lib.disco_apply(eng.ctx, p(X), NUL, p(w), M, 1, p(z), None)),
                      ('istft', lambda: lib.disco_istft(eng.ctx, p(z), G, p(out), None))]
        reps = max(2, min(args.steps, 5))
        acc = {name: 0.0 for name, _ in calls}
        for rep in range(reps + 1):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(calls) + 1)]
            evs[0].record()
            for i, (name, fn) in enumerate(calls):
                eng._chk(fn())
                evs[i + 1].record()
            torch.cuda.synchronize()
            if rep == 0:
                continue            # warm-up of the per-stage sequence
            for i, (name, _) in enumerate(calls):
                acc[name] += evs[i].elapsed_time(evs[i + 1]) / reps
        kab = kernel_alg_bytes(M, K, F, H)
        stages = {}
        for name, ms in acc.items():
            ent = {'ms': round(ms, 4)}
            if name in kab:
                ent['alg_bytes'] = kab[name] * R * K * T
                ent['GBps'] = round(ent['alg_bytes'] / (ms * 1e-3) / 1e9, 1)
            stages[name] = ent
        dom = max((n_ for n_ in acc if n_ in kab), key=lambda n_: acc[n_])
        achieved = stages[dom]['alg_bytes'] / (acc[dom] * 1e-3)
        traffic = None
        tfile = os.path.join(REPO, 'profiles', 'pmc_traffic.json')
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(dom, {}).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': round(achieved / 1e9, 1), 'peak': HBM_PEAK / 1e9,
                    'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK, 4), 'traffic': traffic,
                    'alg_bytes_per_launch': stages[dom]['alg_bytes'], 'avg_launch_ms': round(acc[dom], 4),
                    'pipeline': {'B_alg_per_node_frame': b_alg(M, K, F, H),
                                 'achieved_GBps': round(value / world * b_alg(M, K, F, H) / 1e9, 1),
                                 'frac': round(value / world * b_alg(M, K, F, H) / HBM_PEAK, 4)}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(K, M, Ls)

    # BASELINE.json config the shape corresponds to (C3 is the headline; the others are run by hand)
    shape = (K, M, N)
    cfg_name = ('C4' if args.mask == 'crnn' else 'C3') if shape == (4, 4, 512) else (
        'C2' if shape == (1, 4, 512) else ('C5-shaped' if shape == (8, 8, 1024) else 'custom'))
    if rank == 0:
        mask_desc = 'oracle irm1 mask' if args.mask == 'oracle' else 'CRNN masks in the loop (random weights, fp32 PyTorch-ROCm)'
        line = {
            'metric': 'STFT node-frames/s, whole MWF path (STFT->mask->cov->GEVD-MWF->z exchange->MWF->iSTFT)',
            'value': value, 'unit': 'node-frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'x_realtime': x_rt,
            'config': {'workload': f'{cfg_name}: {R} rooms/GPU x {K} nodes x {M} mics, 16 kHz, L={Ls}, '
                                   f'{N}-pt STFT hop {H}, {mask_desc}, two-step Tango (mask_for_z=local), outputs=enhanced'
                                   + (f', ONLINE mode lambda=0.95 update_every={args.online_every}' if args.online_every else '')
                                   + (f', {args.iters} step-2 iterations (DANSE-style)' if args.iters > 1 else ''),
                       'rooms_per_gpu': R, 'nodes': K, 'mics': M, 'length': Ls, 'n_fft': N, 'frames': T,
                       'parallelism': f'rooms sharded over {world} GPU(s), no data-path collective'},
            'roofline': roofline, 'cpu_baseline': cpu, 'stages': stages,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()                    # rank 0 may still be in its per-stage timing: leave together
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
