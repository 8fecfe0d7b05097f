"""Multi-GPU plumbing of the benchmark / batch driver: rooms shard across ranks with NO data-path collective
(rooms are independent, the reference runs one process per RIR, exp/ex1/loop_tango.sh:28-29).  The only
communication is the timing contract: barrier, max-over-ranks of the elapsed time, sum of the processed units."""
import os


def env_rank_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))


def room_range(rank, world, rooms_per_rank):
    """Weak scaling: every rank owns `rooms_per_rank` rooms; global room ids are contiguous per rank."""
    return rank * rooms_per_rank, (rank + 1) * rooms_per_rank


def split_rooms(total_rooms, world):
    """Strong-scaling helper: contiguous, balanced split of `total_rooms` over `world` ranks."""
    base, rem = divmod(total_rooms, world)
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((start, start + n))
        start += n
    return out


def init(backend, rank, world, device=None):
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29512')
    kw = {'device_id': device} if (device is not None and backend == 'nccl') else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def max_over_ranks(seconds, device='cpu'):
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device='cpu'):
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def whole_job_throughput(units_this_rank, seconds_this_rank, world, device='cpu'):
    """value = units processed by ALL ranks / max-over-ranks time (the bench contract)."""
    if world == 1:
        return units_this_rank / seconds_this_rank, seconds_this_rank
    tmax = max_over_ranks(seconds_this_rank, device)
    total = sum_over_ranks(float(units_this_rank), device)
    return total / tmax, tmax


def free_port():
    import socket
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def launch_ranks(script, argv, n, timeout=None):
    """Start `n` ranks of `python script argv...` on this node, one process per GPU, the way the reference starts one process
    per RIR (exp/ex1/loop_tango.sh:28-29): RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment,
    rendezvous on 127.0.0.1.  stdout / stderr are inherited (rank 0 prints the result line).  Returns the worst exit code;
    if one rank fails the others are terminated (a dead peer would otherwise leave them in a collective forever)."""
    import subprocess
    import sys
    import time
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: what RCCL needs on this host driver
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env))
    t_end = None if timeout is None else time.time() + timeout
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0:
                rc = rc or r
                for q in live:
                    q.terminate()
        if t_end is not None and time.time() > t_end:
            for q in live:
                q.kill()
            return rc or 124
        time.sleep(0.05)
    return rc
