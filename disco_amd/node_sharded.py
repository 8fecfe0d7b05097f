"""Node-sharded two-step MWF: the nodes of a room live on different GPUs and the compressed signals z are exchanged with
ONE all-gather between the two steps -- the communication DISCO's distributed algorithm actually performs
(tango.py:378-386; SURVEY.md 8e "finer sharding").  Rank q of W holds nodes [q*K/W, (q+1)*K/W) of every room.
The DANSE-style iterated scheme (BASELINE configs[4]) adds one more all-gather per extra iteration.

Two drivers over the same staged entry points: `tango_enhance_node_sharded` (numpy host, any all-gather callable) and
`tango_enhance_node_sharded_torch` (device-resident torch tensors, z gathered on the GPUs by torch.distributed --
backend 'nccl' = RCCL over xGMI -- with no host round trip).

The default deployment shards ROOMS (all nodes of a room on one GPU, no data-path collective, z exchanged on chip);
this mode exists for rooms whose nodes do not fit / are produced on different devices.  xGMI is point-to-point, so for
K <= 8 ranks the all-gather is bound by R_local * T * F * 8 bytes per link (1.29 MB per (room, node) at C3).
"""
import numpy as np


def node_range(rank, world, K):
    if K % world:
        raise ValueError(f'{K} nodes do not split evenly over {world} ranks')
    kl = K // world
    return rank * kl, kl


def _run(eng, y_local, mask_z_local, mask_w_local, iters, z_out, gather, yf_out=None, out=None, z_shape=None, X_out=None):
    """Shared data flow.  z_out / yf_out: caller-owned device arrays for this rank's z / yf (or None: DevBuf);
    gather(z_local) -> z of ALL nodes (R, K, T, F), numpy or device array."""
    if iters < 1:
        raise ValueError('iters must be >= 1')
    R, Kl, M = eng.R, eng.Kl, eng.M
    if hasattr(y_local, 'data_ptr'):
        y_sig = y_local.reshape(R * Kl, M, eng.Lsamp)
        assert y_sig.is_contiguous()
    else:
        y_sig = np.ascontiguousarray(y_local, dtype=np.float32).reshape(R * Kl, M, eng.Lsamp)
    X = eng.stft(y_sig, out=X_out).reshape(R, Kl, eng.T, eng.F, M)
    # step 1, local (tango.py:326-376)
    eng.cov_masked(X, mask_z_local, Rss_out=False)
    w_loc, _ = eng.gevd_mwf_r1_pending(M)
    z_all = w_glo = None
    for it in range(iters):
        z_loc = eng.apply(X, w_loc, out=z_out)
        # the exchange (tango.py:378-386): one all-gather of the compressed signals
        z_all = gather(z_loc)
        assert tuple(z_all.shape) == (z_shape or (R, eng.K, eng.T, eng.F))
        # step 2, local again (tango.py:411-450)
        if eng.K > 1:
            eng.cov_masked(X, mask_w_local, z_all, z_all, mask_remote=True, Rss_out=False)
        else:
            eng.cov_masked(X, mask_w_local, Rss_out=False)
        w_glo, _ = eng.gevd_mwf_r1_pending(M + eng.K - 1)
        if it + 1 < iters:
            # DANSE-style continuation (disco_tango_enhance_iterated): re-compress with the local part of the new filter
            w_loc = eng.filter_head(w_glo)
    yf = eng.apply(X, w_glo, Z=z_all if eng.K > 1 else None, out=yf_out)
    out = eng.istft(yf.reshape(R * Kl, eng.T, eng.F), out=out).reshape(R, Kl, eng.Lsamp)
    return out, yf, z_all


def tango_enhance_node_sharded(eng, y_local, mask_z_local, mask_w_local, all_gather_z, iters=1):
    """eng: Engine(rooms=R, nodes=K, ...) on which `set_node_shard(k0, Kl)` has been called.
    y_local (R, Kl, M, L), masks (R, Kl, T, F) -- this rank's nodes.
    all_gather_z: callable taking this rank's z as a numpy (R, Kl, T, F) complex64 array and returning the z of ALL nodes,
    (R, K, T, F), in global node order (torch.distributed all_gather over RCCL in production, gloo in the CPU test).
    iters > 1: the iterated scheme of disco_tango_enhance_iterated, one all-gather per iteration.
    Returns (out_local (R, Kl, L) DevBuf, yf_local DevBuf, z_all numpy -- the LAST exchanged z)."""
    def gather(z_loc):
        return np.ascontiguousarray(all_gather_z(z_loc.numpy()), dtype=np.complex64)
    return _run(eng, y_local, mask_z_local, mask_w_local, iters, None, gather)


def tango_enhance_node_sharded_torch(eng, y_local, mask_z_local, mask_w_local, group=None, iters=1, out=None, gather_events=None):
    """Device-resident variant: y_local (R, Kl, M, L) float32 and the masks (R, Kl, T, F) float32 are contiguous torch tensors
    on this rank's GPU; z stays on the GPUs and is all-gathered by torch.distributed over `group` (RCCL over xGMI; every
    rank holds the same number of nodes, rank order == node order).  The library launches on the null stream, which is
    torch's default stream, so the collective is ordered after the kernels that produce z and before those that read it.
    Per all-gather a rank sends R * Kl * T * F * 8 bytes to each peer (1.29 MB per (room, node) at C3).
    out: optional (R, Kl, L) float32 torch tensor to receive the time signals.  gather_events: optional list that receives a
    (start, stop) pair of torch.cuda events per all-gather (recorded on the current stream; the caller synchronises and reads).
    Returns (out_local (R, Kl, L) DevBuf or `out`, yf_local torch (R, Kl, T, F) complex64, z_all torch (R, K, T, F) complex64)."""
    import torch
    import torch.distributed as dist
    W = dist.get_world_size(group)
    R, Kl, K = eng.R, eng.Kl, eng.K
    if Kl * W != K:
        raise ValueError(f'{W} ranks x {Kl} nodes per rank != {K} nodes')
    dev = y_local.device
    shape = (R, Kl, eng.T, eng.F)
    z_loc = torch.empty(shape, dtype=torch.complex64, device=dev)
    yf = torch.empty(shape, dtype=torch.complex64, device=dev)
    parts = torch.empty((W,) + shape, dtype=torch.complex64, device=dev)
    # the spectra as a torch tensor too: torch's caching allocator hands the same block back every call, while a DevBuf is a
    # hipMalloc + hipFree of several GB per call (measured: 100 ms of a 113 ms step at 250 rooms)
    X = torch.empty((R * Kl, eng.T, eng.F, eng.M), dtype=torch.complex64, device=dev)

    timed = gather_events is not None and dev.type == 'cuda'

    def gather(z):
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        # concatenated-along-dim-0 form: the one every backend (RCCL, gloo) accepts
        if z.is_cuda and dist.get_backend(group) == 'gloo':
            # functional tests of the N-rank data flow on a box with ONE GPU (bench.py --dist-backend gloo --single-device): gloo gathers
            # host tensors only, so this path -- never the measured one -- stages the blocks through the host
            host = torch.empty((W * R, Kl, eng.T, eng.F, 2), dtype=torch.float32)
            dist.all_gather_into_tensor(host, torch.view_as_real(z).cpu(), group=group)
            torch.view_as_real(parts).view(W * R, Kl, eng.T, eng.F, 2).copy_(host)
        else:
            dist.all_gather_into_tensor(torch.view_as_real(parts).view(W * R, Kl, eng.T, eng.F, 2), torch.view_as_real(z), group=group)
        if timed:
            e1.record()
            gather_events.append((e0, e1))
        return parts              # rank-major [W][R][Kl][T][F]: consumed as it arrives (Engine.set_z_blocks), no transposing copy
    eng.set_z_blocks(Kl)
    try:
        out_, yf_, z_rank_major = _run(eng, y_local, mask_z_local, mask_w_local, iters, z_loc, gather, yf_out=yf, out=out, z_shape=(W, R, Kl, eng.T, eng.F), X_out=X)
    finally:
        eng.set_z_blocks(K)
    # the documented return value keeps global node order (R, K, T, F): a view-free copy made only for the caller's benefit
    return out_, yf_, z_rank_major.permute(1, 0, 2, 3, 4).reshape(R, K, eng.T, eng.F)


def torch_all_gather(world):
    """all_gather_z for torch.distributed (backend 'nccl' = RCCL on ROCm, or 'gloo'): contiguous node blocks per rank."""
    import torch
    import torch.distributed as dist

    def gather(z_local):
        t = torch.view_as_real(torch.from_numpy(np.ascontiguousarray(z_local)))          # (R, Kl, T, F, 2) float32
        if dist.get_backend() == 'nccl':
            t = t.cuda()
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        z = torch.cat(parts, dim=1)                                                     # rank order == node order
        return torch.view_as_complex(z.contiguous()).cpu().numpy()
    return gather
