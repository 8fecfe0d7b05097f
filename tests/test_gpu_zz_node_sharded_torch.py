"""Device-resident node-sharded driver on the MI355X (disco_amd/node_sharded.py:tango_enhance_node_sharded_torch): one-rank
RCCL group on the single GPU of the test box -- torch ROCm tensors as z / yf buffers, all_gather_into_tensor, the
iterated scheme with disco_filter_head -- against disco_tango_enhance_iterated.  The two-rank data flow is covered by
tests/test_sharding_gloo.py on CPU; 8-GPU runs are the driver's.  (File name sorts last: written when no GPU time was left
to try it, so it must not stand in front of the parity suite under `-x`.)"""
import pytest

import parity_checks as pc
from disco_amd import _lib
from disco_amd.engine import Engine

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(300)
def test_node_sharded_torch_one_rank_rccl():
    import torch
    lib = _lib.load()
    torch.cuda.set_device(0)
    errs = pc.check_node_sharded_torch_one_rank(lambda **cfg: Engine(lib=lib, **cfg), 'cuda:0', 'nccl', K=4, M=4, L=40000, iters=2)
    print(errs)
