import torch, time
x = torch.empty(1280*1024*1024, dtype=torch.float64, device="cuda").fill_(1.0)   # 10.7 GB
y = torch.empty_like(x[: x.numel() // 2])
for name, fn, nbytes in (("read (sum)", lambda: x.sum(), x.numel()*8), ("copy (read+write)", lambda: y.copy_(x[: y.numel()]), 2*y.numel()*8)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print("%s: %.3f ms, %.2f TB/s" % (name, ms, nbytes / ms / 1e9))
