"""What HBM sustains for the access mix of the memory-heavy layers: streams of 7 GB bf16 tensors through torch's
elementwise kernels (1 read + 1 write, 2 reads + 1 write, 2 reads + 2 writes)."""
import time

import torch

n = 36828480 * 96          # the largest activation tensor of a 64 x 300-frame forward
a = torch.randn(n, device="cuda", dtype=torch.bfloat16)
b = torch.randn(n, device="cuda", dtype=torch.bfloat16)
c = torch.empty_like(a)
d = torch.empty_like(a)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


gb = n * 2 / 1e9
t = timed(lambda: c.copy_(a)); print(f"copy        1R+1W: {2 * gb / t / 1e3:.2f} TB/s ({t * 1e3:.2f} ms)")
t = timed(lambda: torch.add(a, b, out=c)); print(f"add         2R+1W: {3 * gb / t / 1e3:.2f} TB/s ({t * 1e3:.2f} ms)")


def two_out():
    torch.add(a, b, out=c)
    torch.mul(a, b, out=d)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def two_out_streams():
    with torch.cuda.stream(s1):
        torch.add(a, b, out=c)
    with torch.cuda.stream(s2):
        torch.mul(a, b, out=d)


t = timed(two_out_streams); print(f"add || mul  4R+2W on two streams: {6 * gb / t / 1e3:.2f} TB/s ({t * 1e3:.2f} ms)")
t = timed(lambda: torch.sin(a, out=c)); print(f"sin         1R+1W: {2 * gb / t / 1e3:.2f} TB/s ({t * 1e3:.2f} ms)")
