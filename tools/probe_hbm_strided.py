"""Does HBM care about the piece size?  Useful bytes/s when a [M, C] f16 map (C = 576: 1152-byte pixel rows, 472 MB) is read in
column blocks - 64-byte, 128-byte, 256-byte pieces per row, one block per kernel (what a K-step-sliced 1x1 conv does, the
blocks of a row microseconds apart) - against reading it whole."""
import torch
dev = torch.device("cuda:0")
M, C = 64 * 80 * 80, 576
x = torch.empty(M, C, dtype=torch.float16, device=dev).normal_()
def t(fn, nbytes, name, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:52s} {ms*1e3:8.1f} us  {nbytes/ms/1e9:7.2f} TB/s useful", flush=True)
big = torch.empty(1 << 28, dtype=torch.float16, device=dev)       # 512 MiB: flushes the 256 MB Infinity Cache between runs
for w in (32, 64, 128, 288, 576):
    outs = [torch.empty(M, w, dtype=torch.float16, device=dev) for _ in range(C // w)]
    def run():
        big.fill_(0)
        for i, o in enumerate(outs):
            torch.mul(x[:, i * w:(i + 1) * w], 2.0, out=o)
    def flush_only():
        big.fill_(0)
    # time run minus flush
    for _ in range(2): run()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(5): flush_only()
    e[1].record()
    for _ in range(5): run()
    e[2].record(); torch.cuda.synchronize()
    ms = (e[1].elapsed_time(e[2]) - e[0].elapsed_time(e[1])) / 5
    nbytes = 2 * M * C * 2
    print(f"column blocks of {w:3d} ch ({2*w:4d} B pieces), {C//w:2d} kernels: {ms*1e3:8.1f} us  {nbytes/ms/1e9:6.2f} TB/s (read + write)", flush=True)
