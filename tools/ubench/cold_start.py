import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from olavm_amd.backend import Backend
be = Backend(device=0, stream=torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda").manual_seed(1)
vals = torch.randint(0, 2**63 - 1, (94, 1 << 22), dtype=torch.int64, device="cuda", generator=g)
torch.cuda.synchronize()
def run(tag):
    t0 = time.perf_counter(); b = be.commit_dev(vals.data_ptr(), 94, 22); torch.cuda.synchronize(); dt = time.perf_counter() - t0; b.free(); print(tag, round(dt * 1e3, 1), "ms", flush=True)
run("cold"); run("warm"); run("warm")
be.trim(); run("after trim (cache released)"); run("warm")
