"""What the HBM takes per second as pure stores, pure loads and copies (torch kernels, 16 GiB buffers): the yardstick for the store-bound STFT
kernels (k_stft_pairs<1024>: 16.5 GB of stores in 4.8 ms).  Test / measurement tooling.   Usage: python tools/gpu/store_rate.py"""
import torch


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    dev = torch.device('cuda:0')
    n = 4 * 2 ** 30                                     # 4 Gi floats = 16 GiB
    x = torch.empty(n, dtype=torch.float32, device=dev)
    y = torch.empty(n, dtype=torch.float32, device=dev)
    gb = n * 4 / 1e9
    ms = timed(lambda: x.fill_(1.0));            print(f'fill   (store {gb:.1f} GB):          {ms:.3f} ms  {gb / ms:.2f} TB/s')
    ms = timed(lambda: x.zero_());               print(f'zero   (store {gb:.1f} GB):          {ms:.3f} ms  {gb / ms:.2f} TB/s')
    ms = timed(lambda: x.sum());                 print(f'sum    (load  {gb:.1f} GB):          {ms:.3f} ms  {gb / ms:.2f} TB/s')
    ms = timed(lambda: y.copy_(x));              print(f'copy   (load + store {2 * gb:.1f} GB): {ms:.3f} ms  {2 * gb / ms:.2f} TB/s')
    ms = timed(lambda: torch.add(x, 1.0, out=y)); print(f'add    (load + store {2 * gb:.1f} GB): {ms:.3f} ms  {2 * gb / ms:.2f} TB/s')
    h = n // 4
    ms = timed(lambda: torch.add(x[:h], x[h:2 * h], out=y[:h]))
    print(f'a + b  (2 loads + 1 store {3 * gb / 4:.1f} GB): {ms:.3f} ms  {3 * gb / 4 / ms:.2f} TB/s')


if __name__ == '__main__':
    main()
