"""Where the cold-start seconds go: hipMalloc of a large block, the FIRST kernel that writes it, the second one.
(torch is the plumbing: torch.empty -> hipMalloc through the caching allocator on a fresh process.)"""
import sys
import time

import torch

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 24.0
torch.cuda.init()
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    x = torch.empty(int(gb * (1 << 30)) // 8, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    x.fill_(1)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    x.fill_(2)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    y = x[: x.numel() // 2].sum()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print("block %d: %.0f GB  malloc %.1f ms, first fill %.1f ms, second fill %.1f ms, read half %.1f ms" % (rep, gb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
    del x, y
    torch.cuda.empty_cache()          # hipFree: the next block is a new hipMalloc
