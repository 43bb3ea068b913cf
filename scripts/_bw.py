import torch, time
dev = torch.device("cuda:0")
for mb in (126, 252, 503, 1006):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, dtype=torch.float32, device=dev)
    y = torch.empty(n, dtype=torch.float32, device=dev)
    for name, fn, bytes_ in (("fill", lambda: x.fill_(1.0), mb * 1e6), ("copy", lambda: y.copy_(x), 2 * mb * 1e6), ("read (sum)", lambda: x.sum(), mb * 1e6)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 100
        print(f"{mb:5d} MB {name:10s} {us:8.1f} us  {bytes_ / us / 1e6:6.2f} TB/s")
