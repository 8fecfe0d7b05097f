"""C4 (CRNN masks in the loop): where the time goes.  torch.profiler table of one step + the step time."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from disco_amd import _lib, synth
from disco_amd.engine import Engine
from disco_amd.dnn.crnn import build_crnn
from disco_amd.dnn.inloop import tango_enhance_dnn
R, K, M, L = int(sys.argv[1]) if len(sys.argv) > 1 else 125, 4, 4, 160000
dev = torch.device('cuda', 0)
eng = Engine(rooms=R, nodes=K, mics=M, length=L, lib=_lib.load())
y, _, _ = synth.make_rooms_torch(R, K, M, L, device=dev)
torch.manual_seed(0)
mz, mw = build_crnn(1, device=dev), build_crnn(K, device=dev)
for _ in range(2):
    out = tango_enhance_dnn(eng, y, mz, mw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    out = tango_enhance_dnn(eng, y, mz, mw)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f'rooms {R}: {dt * 1e3:.1f} ms/step = {10.0 / dt:.1f} x real-time; peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB')
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    out = tango_enhance_dnn(eng, y, mz, mw)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=18, max_name_column_width=70))
