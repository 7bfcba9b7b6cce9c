import torch, time
dev = torch.device("cuda", 0)
for mb in (256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    for _ in range(3): y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"copy {mb} MB: {ms:.3f} ms  {2*mb/1024/ms*1000/1000:.2f} TB/s (read+write)")
    # read-only reduction and write-only fill
    e0.record()
    for _ in range(10): s = x.sum()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"sum  {mb} MB: {ms:.3f} ms  {mb/1024/ms:.2f} TB/s (read)")
    e0.record()
    for _ in range(10): y.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"fill {mb} MB: {ms:.3f} ms  {mb/1024/ms:.2f} TB/s (write)")
