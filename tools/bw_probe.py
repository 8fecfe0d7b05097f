import torch, time
dev='cuda'
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts=[]
    for _ in range(n):
        a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
N=5*1024**3  # elements float32 -> 20 GiB
x=torch.empty(N, dtype=torch.float32, device=dev)
ms=timeit(lambda: x.fill_(1.0)); print('fill 21.5GB ms',ms,'TB/s',N*4/ms/1e9)
y=torch.empty(N//2, dtype=torch.float32, device=dev); z=torch.empty(N//2, dtype=torch.float32, device=dev)
ms=timeit(lambda: z.copy_(y)); print('copy 10.7->10.7GB ms',ms,'TB/s total',N*4/ms/1e9)
ms=timeit(lambda: torch.sum(x)); print('sum read 21.5GB ms',ms,'TB/s',N*4/ms/1e9)
a=torch.empty(N//4, dtype=torch.float32, device=dev)
ms=timeit(lambda: torch.add(a,1.0,out=y[:N//4])); print('add r5.4 w5.4 ms',ms,'TB/s',N*2/ms/1e9)
# 1 read : 2 write like stft
w=torch.empty(N//4*2, dtype=torch.float32, device=dev)
ms=timeit(lambda: torch.cat([a,a],out=w)); print('cat r5.4(x2) w10.7 ms',ms)
