#!/usr/bin/env python3
"""Host link rate on the GPU box: page-locked host memory <-> HBM with torch's copy engine calls (hipMemcpyAsync),
one stream and several streams at once, for the transfer sizes the crawler uses."""
import time
import torch

assert torch.cuda.is_available()
dev = torch.device("cuda:0")
for mb in (22, 90, 256, 1024):
    n = mb * (1 << 20)
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    for name, src, dst in (("H2D", h, d), ("D2H", d, h)):
        for _ in range(3):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        reps = max(4, 2048 // mb)
        t0 = time.perf_counter()
        for _ in range(reps):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name} {mb:5d} MiB x{reps}: {n * reps / dt / 1e9:6.1f} GB/s")
# three streams, 22 MiB each way at once (what three crawler workers do)
streams = [torch.cuda.Stream() for _ in range(3)]
hs = [torch.empty(22 << 20, dtype=torch.uint8).pin_memory() for _ in range(6)]
ds = [torch.empty(22 << 20, dtype=torch.uint8, device=dev) for _ in range(6)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            ds[i].copy_(hs[i], non_blocking=True)
            hs[3 + i].copy_(ds[3 + i], non_blocking=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"3 streams, H2D + D2H of 22 MiB each: {50 * 3 * (22 << 20) / dt / 1e9:.1f} GB/s each direction")


def concurrent(label, jobs, reps=40):
    """jobs: list of (stream index, direction); every job moves 22 MiB per repetition"""
    n = 22 << 20
    st = [torch.cuda.Stream() for _ in range(1 + max(j[0] for j in jobs))]
    hh = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in jobs]
    dd = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in jobs]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for k, (si, direction) in enumerate(jobs):
            with torch.cuda.stream(st[si]):
                if direction == "up":
                    dd[k].copy_(hh[k], non_blocking=True)
                else:
                    hh[k].copy_(dd[k], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    up = sum(1 for j in jobs if j[1] == "up") * reps * n / dt / 1e9
    down = sum(1 for j in jobs if j[1] == "down") * reps * n / dt / 1e9
    print(f"{label}: up {up:.1f} GB/s, down {down:.1f} GB/s")


concurrent("3 streams, uploads only", [(0, "up"), (1, "up"), (2, "up")])
concurrent("3 streams, downloads only", [(0, "down"), (1, "down"), (2, "down")])
concurrent("1 upload stream + 1 download stream", [(0, "up"), (1, "down")])
concurrent("one stream, upload then download", [(0, "up"), (0, "down")])
concurrent("2 streams, each upload + download", [(0, "up"), (0, "down"), (1, "up"), (1, "down")])
