"""raw pinned host-to-device copy rate of the box (torch): the ceiling the PCIe-inclusive leg of bench.py is read against
(56 GB/s measured; the leg reaches 31-46 GB/s depending on the box)"""
import torch, time
n = 512 << 20
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for chunk in (n, n // 8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(4):
        for o in range(0, n, chunk):
            d[o:o + chunk].copy_(h[o:o + chunk], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"pinned H2D, {chunk >> 20} MiB pieces: {4 * n / dt / 1e9:.1f} GB/s")
s2 = torch.cuda.Stream()
torch.cuda.synchronize()
t0 = time.perf_counter()
half = n // 2
for r in range(4):
    d[:half].copy_(h[:half], non_blocking=True)
    with torch.cuda.stream(s2):
        d[half:].copy_(h[half:], non_blocking=True)
torch.cuda.synchronize()
print(f"two streams: {4 * n / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
