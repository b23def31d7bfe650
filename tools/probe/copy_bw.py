"""What a plain device copy / read-only reduction reaches on this GPU (reference point for the BatchNorm streaming kernels)."""
import torch
for mb in (32, 128, 411, 822):
    n = mb * (1 << 20) // 2
    x = torch.randn(n, device="cuda", dtype=torch.bfloat16)
    y = torch.empty_like(x)
    for name, fn, traffic in (("copy", lambda: y.copy_(x), 2), ("read (sum)", lambda: x.float().sum() if False else torch.sum(x, dtype=torch.float32), 1),
                              ("add 2R1W", lambda: torch.add(x, y, out=y), 3)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print(f"{mb:4d} MB {name:12s} {ms*1e3:8.1f} us  {traffic * mb * 1.048576 / ms:8.1f} GB/s")
for mb in (128, 411, 822):      # write-only
    n = mb * (1 << 20) // 2
    y = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): y.fill_(1.0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): y.fill_(1.0)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"{mb:4d} MB fill (write) {ms*1e3:8.1f} us  {mb * 1.048576 / ms:8.1f} GB/s")
