#!/usr/bin/env python3
"""Benchmark of the MI355X MWF hot path (driver contract: one JSON line on rank 0).

Workload: `--config` picks one of BASELINE.json's configurations (default C3, the one the headline metric is quoted on):
  C2  256 rooms x 1 node  x 4 mics, 512-pt STFT, oracle mask            (single node: step 1 + iSTFT)
  C3  1000 rooms x 4 nodes x 4 mics, 512-pt STFT, oracle mask, z exchange on the GPU
  C4  C3's shape with CRNN masks (PyTorch-ROCm) in the loop, 125 rooms per GPU (1000 rooms over 8 GPUs)
  C5  200 rooms x 8 nodes x 8 mics, 1024-pt STFT, 2 step-2 iterations (DANSE-style stress shape)
all at 16 kHz, 10 s clips (L = 160000); --rooms/--nodes/--mics/--n-fft/--iters/--mask override single fields.
One "step" = the whole path over the whole batch: oracle mask (2 STFTs/node) -> STFT -> covariance -> GEVD-MWF solve -> z ->
exchange -> covariance -> solve -> filter -> iSTFT, inputs and outputs resident in HBM.  metric = node-frames/s (1 node-frame =
one hop of all M mics of one node); x real-time = audio seconds per room / seconds per step.

Multi-GPU (`--gpus N`): one process per GPU.  Under torch.distributed.run (RANK/WORLD_SIZE in the environment) the ranks are
taken as given; started plainly with --gpus N > 1 the script launches its own N ranks (disco_amd/dist.py:launch_ranks).
  --shard rooms (default): rooms shard across ranks with NO data-path collective (weak scaling: `--rooms` per GPU); RCCL only
                           carries the timing barrier / max / sum of the contract.
  --shard nodes          : the nodes of every room are split over the ranks and z is exchanged with one RCCL all-gather per
                           step-2 iteration -- the exchange DISCO's algorithm performs (tango.py:378-386); the line then carries
                           the all-gather's bytes per rank and link rate.
After the timed region rank 0 checks sampled rooms of the LAST timed step against the float64 CPU oracle ("parity_sample";
a failure exits non-zero after printing the line).
"""
import argparse
import hashlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK = 8.0e12          # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured achievable)

CONFIGS = {
    'C2': dict(rooms=256, nodes=1, mics=4, n_fft=512, iters=1, mask='oracle'),
    'C3': dict(rooms=1000, nodes=4, mics=4, n_fft=512, iters=1, mask='oracle'),
    'C4': dict(rooms=125, nodes=4, mics=4, n_fft=512, iters=1, mask='crnn'),
    'C5': dict(rooms=200, nodes=8, mics=8, n_fft=1024, iters=2, mask='oracle'),
}


def b_alg(M, K, F, H, iters=1):
    """algorithmic bytes per node-frame of the whole path ('enhanced' outputs)."""
    if K == 1:
        return 8 * M * H + 4 * F + 4 * H
    base = 16 * M * H + 8 * F + 16 * (K - 1) * F + 8 * F + 4 * H
    return base + (iters - 1) * (8 * M * H + 4 * F + 16 * (K - 1) * F + 8 * F)


def kernel_alg_bytes(M, K, F, H):
    """Compulsory bytes per node-frame of each stage given its C-ABI contract (inputs once + outputs once);
    DESIGN.md section 'Kernels' derives them.  Keys are the library's stage names (disco_stage_report)."""
    return {
        'mask_oracle': 2 * H * 4 + F * 4,                     # s_ref, n_ref hop samples in, mask out
        'stft': M * H * 4 + M * F * 8,                        # hop samples of M mics in, M*F bins out
        'stft_cov1': M * H * 4 + M * F * 8 + F * 4,           # samples + mask in, X out (covariances amortised over T)
        'stft_cov1_nostore': M * H * 4 + F * 4,               # samples + mask in (X not materialised)
        'cov1': M * F * 8 + F * 4,                            # X + mask in (covariances: amortised over T)
        'apply1': M * F * 8 + F * 8,                          # X in, z out
        'cov2': M * F * 8 + F * 4 + (K - 1) * F * 8,          # X + mask + remote z in
        'room_cov2': M * F * 8 + F * 4 + F * 8,               # X + mask in, z out (all nodes of a room in one workgroup: remote z's stay on chip)
        'apply2': M * F * 8 + (K - 1) * F * 8 + F * 8,        # X + remote z in, yf out
        'step2_cov': M * F * 8 + F * 4,                       # X + mask in (z stays on chip)
        'step2_apply': M * F * 8 + F * 8,                     # X in, yf out
        'step2_apply_istft': M * F * 8 + H * 4,               # X in, hop samples out (yf stays on chip)
        'stft_apply_istft': M * H * 4 + H * 4,                # samples in, hop samples out (single node, nothing materialised)
        'step2_stft_apply_istft': M * H * 4 + H * 4,          # samples in, hop samples out (spectra re-transformed, z / yf on chip)
        'istft': F * 8 + H * 4,                               # yf in, hop samples out
    }


def csrc_digest():
    """Identity of the kernel sources a PMC traffic file was measured on (profiles/pmc_traffic*.json carry it)."""
    h = hashlib.sha256()
    d = os.path.join(REPO, 'disco_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


_WORKER = r"""
import sys, time
sys.path.insert(0, sys.argv[1])
room, K, M, L, start_at, n_fft = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]), int(sys.argv[7])
from disco_amd import synth
from oracle import tango_oracle as to
y, s, n, _ = synth.make_room_numpy(room, K=K, M=M, L=L)
late = time.time() > start_at
while time.time() < start_at:
    time.sleep(0.005)
t0 = time.time()
to.offline_tango_vec(y, s, n, vads=['irm1', 'irm1'], n_fft=n_fft, hop=n_fft // 2, precision='f64', solver='eigh')
print(t0, time.time(), int(late))
"""


def cpu_vectorised_all_cores(K, M, L, n_fft=512, lead_seconds=20.0, limit_seconds=90.0):
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
            ps.append(subprocess.Popen([sys.executable, '-c', _WORKER, REPO, str(r), str(K), str(M), str(L), repr(start_at), str(n_fft)],
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
    T = 1 + L // (n_fft // 2)
    return {'value': procs * K * T / wall, 'unit': 'node-frames/s', 'cores': procs, 'seconds': round(wall, 2),
            'what': f'{procs} processes x 1 room each (OMP_NUM_THREADS=1), released together, '
                    'oracle/tango_oracle.py:offline_tango_vec'}


def cpu_baseline(K, M, L, n_fft=512):
    """The reference's CPU path (literal loop nest, oracle/tango_oracle.py:offline_tango_literal -- pinned bit-exact against
    the reference's own code) on a bounded sample of the same workload, one host core: one room when that fits ~30 s, else
    the first nodes' worth the clock allows (the literal loop nest costs ~2 s per node at 4 mics, ~8x that at 8 mics / P = 15)."""
    from disco_amd import synth
    from oracle import tango_oracle as to
    hop = n_fft // 2
    T = 1 + L // hop
    Ls = L if K * M <= 16 else L // 8                     # C5-shaped rooms: an eighth of the clip keeps the sample bounded
    y, s, n, _ = synth.make_room_numpy(0, K=K, M=M, L=Ls)
    Ts = 1 + Ls // hop
    t0 = time.perf_counter()
    to.offline_tango_literal(y, s, n, vads=['irm1', 'irm1'], n_fft=n_fft, hop=hop)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    to.offline_tango_vec(y, s, n, vads=['irm1', 'irm1'], n_fft=n_fft, hop=hop, precision='f64', solver='eigh')
    dtv = time.perf_counter() - t0
    return {'value': K * Ts / dt, 'unit': 'node-frames/s', 'cores': 1, 'kind': 'port',
            'vectorised_numpy': {'value': K * Ts / dtv, 'unit': 'node-frames/s', 'seconds': round(dtv, 2),
                                 'what': 'oracle/tango_oracle.py:offline_tango_vec (float64, einsum covariances, batched eigh), same sample'},
            'vectorised_numpy_all_cores': cpu_vectorised_all_cores(K, M, L, n_fft) if K * M <= 16 else None,
            'sample': f'1 room ({K} nodes x {M} mics, {Ls} of {L} samples = {K * Ts} node-frames), literal reference loop nest '
                      f'(tango.py:326-457 restated, bit-exact vs reference), STFTs included, {dt:.1f} s on 1 of '
                      f'{os.cpu_count()} host cores',
            'x_realtime': (Ls / 16000.0) / dt}


def parity_sample(samples, n_fft, iters, tol=1e-4):
    """Sampled rooms of the batch the timed region just processed, against the float64 CPU oracle (test infrastructure used
    as the checker, never as the thing measured).  samples: [(room id, y (K,M,L), s_ref (K,L), n_ref (K,L), got (Kl,L), k0)]
    as host arrays; `got` are this rank's nodes [k0, k0+Kl) of the room's output."""
    import numpy as np
    from oracle import stft_oracle as so
    from oracle import tango_oracle as to
    worst, per_room = 0.0, {}
    for r, yr, sr, nr, got, k0 in samples:
        L = yr.shape[-1]
        s = np.zeros_like(yr)
        n = np.zeros_like(yr)
        s[:, 0] = sr                                      # the masks only look at the reference microphone (tango.py:338-342)
        n[:, 0] = nr
        o = to.offline_tango_vec(yr, s, n, vads=['irm1', 'irm1'], n_fft=n_fft, hop=n_fft // 2, precision='f64', solver='eigh',
                                 extra_iters=iters - 1)
        e = 0.0
        for kl in range(got.shape[0]):
            ref = so.istft(o['yf'][k0 + kl], L, n_fft, n_fft // 2, work_dtype=np.float64)
            e = max(e, float(np.linalg.norm(got[kl] - ref) / np.linalg.norm(ref)))
        per_room[int(r)] = e
        worst = max(worst, e)
    return {'rooms': [int(x[0]) for x in samples], 'worst_rel': worst, 'tol': tol, 'ok': bool(worst < tol),
            'per_room': per_room,
            'oracle': 'oracle/tango_oracle.py:offline_tango_vec(float64) + oracle/stft_oracle.py:istft, per (room, node) '
                      '||out - ref||_2 / ||ref||_2 on the output of the last timed step'}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='C3', choices=sorted(CONFIGS), help='BASELINE.json configuration (shape defaults)')
    ap.add_argument('--rooms', type=int, default=None, help='rooms per GPU')
    ap.add_argument('--nodes', type=int, default=None)
    ap.add_argument('--mics', type=int, default=None)
    ap.add_argument('--length', type=int, default=160000)
    ap.add_argument('--n-fft', type=int, default=None)
    ap.add_argument('--mask', default=None, choices=['oracle', 'crnn'],
                    help="'crnn': BASELINE configs[3] -- randomly initialised CRNN mask estimators (PyTorch-ROCm) in the loop")
    ap.add_argument('--online-every', type=int, default=0,
                    help='> 0: time the ONLINE pipeline (SURVEY 8f-2) with a filter update every this many frames instead of '
                         'the batch path (not the headline metric; roofline is skipped)')
    ap.add_argument('--iters', type=int, default=None,
                    help='> 1: the DANSE-style iterated scheme (BASELINE configs[4]; disco_tango_enhance_iterated)')
    ap.add_argument('--shard', default='rooms', choices=['rooms', 'nodes'],
                    help="'nodes': split the nodes of every room over the ranks, one RCCL all-gather of z per step-2 iteration")
    ap.add_argument('--graph', action='store_true',
                    help='capture one step (mask + whole path) into a hipGraph on a side stream and time its replays: one launch per '
                         'step instead of 6-20 (matters for small batches; oracle masks, room-sharded batch path only)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-stage-timing', action='store_true')
    ap.add_argument('--no-parity', action='store_true', help='skip the sampled-room oracle check after the timed region')
    ap.add_argument('--parity-rooms', type=int, default=3)
    ap.add_argument('--pmc-calibrate', action='store_true', help='also run a 4 GiB device copy (known bytes) for PMC calibration')
    ap.add_argument('--selftest-launch', action='store_true',
                    help='no GPU work: every rank joins a gloo group and rank 0 prints n_gpus (covers the self-launch path on CPU)')
    args = ap.parse_args(argv)
    cfg = CONFIGS[args.config]
    for k in ('rooms', 'nodes', 'mics', 'n_fft', 'iters', 'mask'):
        if getattr(args, k) is None:
            setattr(args, k, cfg[k])
    return args


def main(argv=None):
    args = parse_args(argv)
    from disco_amd import dist as dd
    rank, world, local_rank = dd.env_rank_world()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # started plainly: become the launcher of N ranks of this very command line (one process per GPU)
        if not args.selftest_launch:
            import torch
            have = torch.cuda.device_count()
            if have < args.gpus:
                raise SystemExit(f'--gpus {args.gpus} but only {have} GPU(s) are visible')
        raise SystemExit(dd.launch_ranks(os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv), args.gpus))
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE {world}: start with --gpus {world}, or plainly (no torchrun) to self-launch')

    if args.selftest_launch:
        dist = dd.init('gloo', rank, world)
        value, dt = dd.whole_job_throughput(1000.0 * (rank + 1), 0.5 + 0.1 * rank, world)
        if rank == 0:
            print(json.dumps({'selftest': 'launch', 'n_gpus': world, 'value': value, 'seconds': dt}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return 0

    import numpy as np
    import torch
    from disco_amd import _lib, synth
    from disco_amd.engine import Engine

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: there is no CPU path')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    node_sharded = args.shard == 'nodes'
    if world == 1 and node_sharded:
        os.environ.setdefault('MASTER_PORT', str(dd.free_port()))
    dist = dd.init('nccl', rank, world, device=dev) if (world > 1 or node_sharded) else None    # RCCL (one-rank group for --shard nodes)

    R, K, M, Ls, N = args.rooms, args.nodes, args.mics, args.length, args.n_fft
    H, F = N // 2, N // 2 + 1
    lib = _lib.load()
    if node_sharded and (args.mask != 'oracle' or args.online_every):
        raise SystemExit('--shard nodes runs the batch path with oracle masks')
    eng = Engine(rooms=R, nodes=K, mics=M, length=Ls, n_fft=N, device=local_rank, lib=lib)
    T = eng.T
    assert torch.cuda.current_stream().cuda_stream == 0, 'bench times the null stream the library launches on'

    want_parity = rank == 0 and not args.no_parity and args.mask == 'oracle' and args.online_every == 0
    n_s = max(1, min(args.parity_rooms, R))
    sample_rooms = sorted({int(round(i * (R - 1) / max(n_s - 1, 1))) for i in range(n_s)})
    if K * M > 16:
        sample_rooms = sample_rooms[:1]                   # a P = 15 room costs the float64 oracle ~1 min
    sample_in = {}

    # synthetic rooms, generated on the GPU (SURVEY 8d recipe)
    if node_sharded:
        from disco_amd import node_sharded as ns
        k0, Kl = ns.node_range(rank, world, K)
        eng.set_node_shard(k0, Kl)
        # every rank draws the same R rooms and keeps its own nodes (the generator is per room; slicing keeps it simple)
        y_all, s_all, n_all = synth.make_rooms_torch(R, K, M, Ls, first_room=0, device=dev, ref_only_sn=True)
        y = y_all[:, k0:k0 + Kl].contiguous()
        s_ref = s_all[:, k0:k0 + Kl].contiguous()
        n_ref = n_all[:, k0:k0 + Kl].contiguous()
        if want_parity:                                   # the oracle needs ALL nodes of a sampled room
            sample_in = {r: (y_all[r].cpu().numpy(), s_all[r].cpu().numpy(), n_all[r].cpu().numpy()) for r in sample_rooms}
        del y_all, s_all, n_all
        units_per_step = R * Kl * T
    else:
        k0, Kl = 0, K
        first_room, _ = dd.room_range(rank, world, R)     # rank r owns rooms [r*R, (r+1)*R)
        y, s_ref, n_ref = synth.make_rooms_torch(R, K, M, Ls, first_room=first_room, device=dev, ref_only_sn=True)
        if want_parity:
            sample_in = {r: (y[r].cpu().numpy(), s_ref[r].cpu().numpy(), n_ref[r].cpu().numpy()) for r in sample_rooms}
        units_per_step = R * K * T
    mask = torch.empty((R, Kl, T, F), dtype=torch.float32, device=dev)
    out = torch.empty((R, Kl, Ls), dtype=torch.float32, device=dev)
    ws = None if node_sharded else torch.empty(eng.workspace_bytes(), dtype=torch.uint8, device=dev)
    G = R * Kl

    if args.mask == 'crnn':
        from disco_amd.dnn.crnn import build_crnn
        from disco_amd.dnn.inloop import tango_enhance_dnn
        torch.manual_seed(0)
        model_z = build_crnn(1, device=dev)
        model_w = build_crnn(K, device=dev) if K > 1 else None

    gather_events = []                     # (start, stop) torch events around every all-gather of z (--shard nodes)

    def step():
        if args.mask == 'crnn':
            out.copy_(tango_enhance_dnn(eng, y, model_z, model_w))
            return
        eng._chk(lib.disco_mask_oracle(eng.ctx, s_ref.data_ptr(), n_ref.data_ptr(), G, mask.data_ptr(), None))
        if node_sharded:
            ns.tango_enhance_node_sharded_torch(eng, y, mask, mask, iters=args.iters, out=out, gather_events=gather_events)
            return
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

    eager_step = step
    if args.graph:
        if args.mask != 'oracle' or node_sharded or args.online_every:
            raise SystemExit('--graph captures the room-sharded batch path with oracle masks')
        eng.reserve(0)
        step()                              # nothing is left to allocate inside the captured calls
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=dev)
        hip_graph = torch.cuda.CUDAGraph()
        h = side.cuda_stream
        with torch.cuda.graph(hip_graph, stream=side):
            eng._chk(lib.disco_mask_oracle(eng.ctx, s_ref.data_ptr(), n_ref.data_ptr(), G, mask.data_ptr(), h))
            if args.iters > 1:
                eng._chk(lib.disco_tango_enhance_iterated(eng.ctx, y.data_ptr(), mask.data_ptr(), mask.data_ptr(), args.iters,
                                                          out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), h))
            else:
                eng._chk(lib.disco_tango_enhance(eng.ctx, y.data_ptr(), mask.data_ptr(), mask.data_ptr(), out.data_ptr(),
                                                 None, None, ws.data_ptr(), ws.numel(), h))
        torch.cuda.synchronize()
        step = hip_graph.replay             # replays on the current (null) stream, the one the barriers drain

    if args.pmc_calibrate:
        src = torch.empty(1 << 30, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        torch.neg(src, out=dst)             # 4 GiB read + 4 GiB written by ONE elementwise kernel (a plain copy_ goes to the DMA engines)
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
    value, dt = dd.whole_job_throughput(units_per_step * args.steps, dt, world, device=dev)
    assert bool(torch.isfinite(out).all())
    gather_ms = [a_.elapsed_time(b_) for a_, b_ in gather_events[-args.steps * args.iters:]] if gather_events else []
    ms_per_step = 1e3 * dt / args.steps
    x_rt = (Ls / 16000.0) / (dt / args.steps)

    # ---- sampled rooms of the last timed step against the CPU oracle (rank 0)
    parity = None
    if want_parity:
        parity = parity_sample([(r,) + sample_in[r] + (out[r].cpu().numpy(), k0) for r in sample_rooms], N, args.iters)

    # ---- per-stage timing with the library's own HIP events on the launch stream (rank 0), for the roofline object
    roofline, stages = None, None
    if rank == 0 and not args.no_stage_timing and args.mask == 'oracle' and not node_sharded:
        reps = max(2, min(args.steps, 5))
        eager_step()                            # the event objects are created inside the library: warm that path once
        torch.cuda.synchronize()
        # one report per step, and the MEDIAN over the steps: an event pair also spans whatever the host does between the two
        # records, and a single descheduled launch call (seen once: 85 ms inside one 7 ms stage) would otherwise own the mean
        per_rep = []
        for _ in range(reps):
            eng.stage_timing(True)
            eager_step()                        # (a captured graph carries no events: the stage pass always launches eagerly)
            per_rep.append(eng.stage_report())
        eng.stage_timing(False)
        kab = kernel_alg_bytes(M, K, F, H)
        stages = {}
        for name in per_rep[0]:
            ms_list = sorted(r_[name][0] for r_ in per_rep if name in r_)
            per_step = ms_list[len(ms_list) // 2] if len(ms_list) % 2 else 0.5 * (ms_list[len(ms_list) // 2 - 1] + ms_list[len(ms_list) // 2])
            launches = per_rep[0][name][1]
            ent = {'ms': round(per_step, 4), 'launches_per_step': float(launches), 'ms_min': round(ms_list[0], 4), 'ms_max': round(ms_list[-1], 4)}
            if name in kab:
                ent['alg_bytes'] = kab[name] * R * K * T * launches     # per step (all launches of the stage)
                ent['GBps'] = round(ent['alg_bytes'] / (per_step * 1e-3) / 1e9, 1)
            stages[name] = ent
        if args.online_every == 0:
            cand = [n_ for n_ in stages if 'alg_bytes' in stages[n_]]
            dom = max(cand, key=lambda n_: stages[n_]['ms'])
            lps = stages[dom]['launches_per_step']
            launch_ms = stages[dom]['ms'] / lps
            launch_bytes = stages[dom]['alg_bytes'] / lps
            achieved = launch_bytes / (launch_ms * 1e-3)
            traffic, traffic_note = None, None
            tfile = os.path.join(REPO, 'profiles', f'pmc_traffic_{args.config}.json')
            if os.path.exists(tfile):
                try:
                    tj = json.load(open(tfile))
                    traffic = tj.get(dom, {}).get('hbm_bytes_per_launch')
                    if tj.get('_csrc_digest') != csrc_digest():
                        traffic_note = (f'{os.path.basename(tfile)} was measured on other kernel sources (digest '
                                        f'{tj.get("_csrc_digest")} vs {csrc_digest()}): stale, shown for orientation only')
                except Exception:
                    traffic = None
            pipeline_b = b_alg(M, K, F, H, args.iters)
            roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': round(achieved / 1e9, 1), 'peak': HBM_PEAK / 1e9,
                        'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK, 4), 'traffic': traffic,
                        'alg_bytes_per_launch': launch_bytes, 'avg_launch_ms': round(launch_ms, 4),
                        'pipeline': {'B_alg_per_node_frame': pipeline_b,
                                     'achieved_GBps': round(value / world * pipeline_b / 1e9, 1),
                                     'frac': round(value / world * pipeline_b / HBM_PEAK, 4)}}
            if traffic_note:
                roofline['traffic_note'] = traffic_note

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(K, M, Ls, N)

    exchange = None
    if node_sharded:
        # one all-gather per step-2 iteration: every rank receives the other ranks' z (R * (K - Kl) * T * F complex64)
        per_gather = R * (K - Kl) * T * F * 8
        exchange = {'collective': 'all_gather_into_tensor (RCCL)', 'gathers_per_step': args.iters,
                    'bytes_received_per_rank_per_gather': per_gather,
                    'bytes_per_peer_link_per_gather': R * Kl * T * F * 8,
                    'ms_per_gather': (sum(gather_ms) / len(gather_ms)) if gather_ms else None,
                    'link_GBps': (R * Kl * T * F * 8 / (sum(gather_ms) / len(gather_ms) * 1e-3) / 1e9) if (gather_ms and world > 1) else None,
                    'note': 'xGMI is point-to-point: each of the W-1 peers sends its R*Kl*T*F*8-byte block over its own link'}

    if rank == 0:
        shape = (K, M, N, args.iters, args.mask)
        cfg_shape = CONFIGS[args.config]
        is_cfg = shape == (cfg_shape['nodes'], cfg_shape['mics'], cfg_shape['n_fft'], cfg_shape['iters'], cfg_shape['mask'])
        cfg_name = args.config if is_cfg else 'custom'
        mask_desc = 'oracle irm1 mask' if args.mask == 'oracle' else 'CRNN masks in the loop (random weights, PyTorch-ROCm)'
        par = (f'nodes of every room split over {world} GPU(s) ({Kl} per rank), one RCCL all-gather of z per step-2 iteration'
               if node_sharded else f'rooms sharded over {world} GPU(s), no data-path collective')
        line = {
            'metric': 'STFT node-frames/s, whole MWF path (STFT->mask->cov->GEVD-MWF->z exchange->MWF->iSTFT)',
            'value': value, 'unit': 'node-frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong' if node_sharded else 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'x_realtime': x_rt,
            'config': {'workload': f'{cfg_name}: {R} rooms{"" if node_sharded else "/GPU"} x {K} nodes x {M} mics, 16 kHz, L={Ls}, '
                                   f'{N}-pt STFT hop {H}, {mask_desc}, two-step Tango (mask_for_z=local), outputs=enhanced'
                                   + (f', ONLINE mode lambda=0.95 update_every={args.online_every}' if args.online_every else '')
                                   + (f', {args.iters} step-2 iterations (DANSE-style)' if args.iters > 1 else ''),
                       'launch': 'one hipGraph replay per step' if args.graph else 'eager kernel launches',
                       'rooms_per_gpu': R, 'nodes': K, 'mics': M, 'length': Ls, 'n_fft': N, 'frames': T,
                       'iters': args.iters, 'parallelism': par},
            'roofline': roofline, 'cpu_baseline': cpu, 'parity_sample': parity, 'stages': stages,
        }
        if exchange:
            line['exchange'] = exchange
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()                    # rank 0 may still be in its per-stage timing / parity check: leave together
        dist.destroy_process_group()
    if parity is not None and not parity['ok']:
        print(f'PARITY FAILURE: worst relative error {parity["worst_rel"]:.3e} >= {parity["tol"]}', file=sys.stderr)
        return 3
    return 0


if __name__ == '__main__':
    sys.exit(main())
