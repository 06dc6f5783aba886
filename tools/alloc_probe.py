"""Streaming rate of ten 1.7 GB allocations of one process (a copy of one half onto the other, three rounds): the rate
is a property of the allocation -- 4-5 % between them, stable per allocation.  python tools/alloc_probe.py"""
import torch, time
dev = torch.device("cuda", 0)
N = 1708919552 // 4
ts = [torch.empty(N, dtype=torch.float32, device=dev).normal_() for _ in range(10)]
h = N // 2
for rep in range(3):
    out = []
    for t in ts:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(4):
            t[:h].copy_(t[h:2 * h])
        e1.record()
        torch.cuda.synchronize()
        out.append(4 * 2 * h * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    print("rep", rep, " ".join("%.3f" % v for v in out), "TB/s")
print(["%x" % t.data_ptr() for t in ts])
