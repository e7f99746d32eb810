import torch, time
for mb in (64, 256, 1024):
    n = mb << 20
    a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    a.fill_(3)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): b.copy_(a)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 / 1e3
    print(f"copy {mb} MB: {t*1e6:.1f} us  {2*n/t/1e12:.2f} TB/s read+written")
    x = a.view(torch.int64)
    s.record()
    for _ in range(20): y = x + 1
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 / 1e3
    print(f"add  {mb} MB: {t*1e6:.1f} us  {2*n/t/1e12:.2f} TB/s read+written")
