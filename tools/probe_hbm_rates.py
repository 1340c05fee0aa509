"""What a plain streaming kernel gets from HBM on this box (torch elementwise kernels, 1 GiB operands, well beyond the 256 MB
Infinity Cache): the practical ceiling the HBM-bound layers (1x1 convs, stem, decode) should be read against."""
import torch, time
dev = torch.device("cuda:0")
n = 1 << 29           # halves: 1 GiB
a = torch.empty(n, dtype=torch.float16, device=dev).normal_()
b = torch.empty_like(a)
c = torch.empty_like(a)
def t(fn, nbytes, name, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:34s} {ms*1e3:8.1f} us  {nbytes/ms/1e9:7.2f} TB/s", flush=True)
t(lambda: b.copy_(a), 2 * n * 2, "copy (1 read + 1 write)")
t(lambda: torch.add(a, b, out=c), 3 * n * 2, "add (2 reads + 1 write)")
t(lambda: b.fill_(1.0), n * 2, "fill (write only)")
t(lambda: a.sum(), n * 2, "sum (read only)")
t(lambda: torch.mul(a, 2.0, out=b), 2 * n * 2, "scale (1 read + 1 write)")
for mb in (64, 128, 256, 512):
    m = mb * (1 << 19)
    x, y = a[:m], b[:m]
    t(lambda: y.copy_(x), 2 * m * 2, f"copy of {mb} MiB (in + out = {2*mb} MiB)")
