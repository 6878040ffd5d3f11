"""Times ola_permuted_cols_dev (operands resident) against the oracle's sequential restatement on the host.
    python tools/bench_lookup.py [log_n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from olavm_amd.backend import Backend
from tests import oracle_lib

be = Backend(device=0)
o = oracle_lib.load()
rng = np.random.default_rng(1)
for log_n in [int(a) for a in sys.argv[1:]] or [17, 20, 22, 24]:
    n = 1 << log_n
    a = rng.integers(0, n // 3, n).astype(np.uint64)
    b = np.arange(n, dtype=np.uint64)
    da, db = torch.from_numpy(a.view(np.int64)).cuda(), torch.from_numpy(b.view(np.int64)).cuda()
    di, dt = torch.empty_like(da), torch.empty_like(da)
    be.permuted_cols_dev(da.data_ptr(), db.data_ptr(), n, di.data_ptr(), dt.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        be.permuted_cols_dev(da.data_ptr(), db.data_ptr(), n, di.data_ptr(), dt.data_ptr())
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    oi, ot = o.permuted_cols(a, b)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    ok = np.array_equal(di.cpu().numpy().view(np.uint64), oi) and np.array_equal(dt.cpu().numpy().view(np.uint64), ot)
    print("n = 2^%d: device %.2f ms (%.0f M rows/s), oracle on one host core %.0f ms, identical: %s" % (log_n, gpu_ms, n / gpu_ms / 1e3, cpu_ms, ok), flush=True)
