import time, torch
torch.cuda.init(); torch.cuda.synchronize()
for gb in (1, 8, 32, 64):
    t=time.perf_counter(); x=torch.empty(gb<<30, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); t1=time.perf_counter()-t
    t=time.perf_counter(); x.zero_(); torch.cuda.synchronize(); t2=time.perf_counter()-t
    t=time.perf_counter(); x.zero_(); torch.cuda.synchronize(); t3=time.perf_counter()-t
    del x; torch.cuda.empty_cache(); torch.cuda.synchronize()
    print(f"{gb} GB: hipMalloc {t1*1e3:.1f} ms, first touch {t2*1e3:.1f} ms, second {t3*1e3:.1f} ms", flush=True)
