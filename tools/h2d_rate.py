import torch, time
for mb in (16, 64, 256, 1024):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for _ in range(2): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    reps = max(4, 4096 // mb)
    t0 = time.perf_counter()
    for _ in range(reps): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("pinned->device %5d MiB pieces: %.1f GB/s" % (mb, n * reps / dt / 1e9))
# two streams
n = 256 << 20
hs = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(2)]
ds = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(2)]
ss = [torch.cuda.Stream() for _ in range(2)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(8):
    for i in range(2):
        with torch.cuda.stream(ss[i]): ds[i].copy_(hs[i], non_blocking=True)
torch.cuda.synchronize()
print("two streams: %.1f GB/s" % (n * 16 / (time.perf_counter() - t0) / 1e9))
