"""Node-sharded two-step MWF: the nodes of a room live on different GPUs and the compressed signals z are exchanged with
ONE all-gather between the two steps -- the communication DISCO's distributed algorithm actually performs
(tango.py:378-386; SURVEY.md 8e "finer sharding").  Rank q of W holds nodes [q*K/W, (q+1)*K/W) of every room.

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


def tango_enhance_node_sharded(eng, y_local, mask_z_local, mask_w_local, all_gather_z):
    """eng: Engine(rooms=R, nodes=K, ...) on which `set_node_shard(k0, Kl)` has been called.
    y_local (R, Kl, M, L), masks (R, Kl, T, F) -- this rank's nodes.
    all_gather_z: callable taking this rank's z as a numpy (R, Kl, T, F) complex64 array and returning the z of ALL nodes,
    (R, K, T, F), in global node order (torch.distributed all_gather over RCCL in production, gloo in the CPU test).
    Returns (out_local (R, Kl, L) DevBuf, yf_local DevBuf, z_all numpy)."""
    R, Kl, M = eng.R, eng.Kl, eng.M
    X = eng.stft(np.ascontiguousarray(y_local, dtype=np.float32).reshape(R * Kl, M, eng.Lsamp)).reshape(R, Kl, eng.T, eng.F, M)
    # step 1, local (tango.py:326-376)
    eng.cov_masked(X, mask_z_local, Rss_out=False)
    w_loc, _ = eng.gevd_mwf_r1_pending(M)
    z_loc = eng.apply(X, w_loc)
    # the exchange (tango.py:378-386): one all-gather of the compressed signals
    z_all = np.ascontiguousarray(all_gather_z(z_loc.numpy()), dtype=np.complex64)
    assert z_all.shape == (R, eng.K, eng.T, eng.F)
    # step 2, local again (tango.py:411-450)
    if eng.K > 1:
        eng.cov_masked(X, mask_w_local, z_all, z_all, mask_remote=True, Rss_out=False)
    else:
        eng.cov_masked(X, mask_w_local, Rss_out=False)
    w_glo, _ = eng.gevd_mwf_r1_pending(M + eng.K - 1)
    yf = eng.apply(X, w_glo, Z=z_all if eng.K > 1 else None)
    out = eng.istft(yf.reshape(R * Kl, eng.T, eng.F)).reshape(R, Kl, eng.Lsamp)
    return out, yf, z_all


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
