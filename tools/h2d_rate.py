"""Pinned host → device copy rate of this box, 1..8 concurrent copy streams, nothing else running (GPU box).  Reference point
for bench.py's host-fed loop: the idle link does ≈57 GB/s here; under the pipeline's kernels the same copies get ≈32 GB/s."""
import torch, time
dev = torch.device("cuda:0")
n = 655_360_000  # int16 elements = 1.31 GB
host = torch.empty(n, dtype=torch.int16).pin_memory()
d = torch.empty(n, dtype=torch.int16, device=dev)
for parts in (1, 2, 4, 8):
    streams = [torch.cuda.Stream(dev) for _ in range(parts)]
    chunk = n // parts
    def go():
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                d[k * chunk:(k + 1) * chunk].copy_(host[k * chunk:(k + 1) * chunk], non_blocking=True)
    go(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): go()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(f"{parts} stream(s): {n * 2 / dt / 1e9:.1f} GB/s")
