import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from olavm_amd.backend import Backend, OLA_NTT_COSET_LDE, OLA_NTT_COSET_LDE_LEAF_ORDER
from tests.inputs import splitmix_columns
from tests import oracle_lib
o = oracle_lib.load()
be = Backend(device=0, stream=torch.cuda.current_stream().cuda_stream)
log_n, cols = int(sys.argv[1]), int(sys.argv[2])
n, N = 1 << log_n, 8 << log_n
x = splitmix_columns(torch, cols, n)
def leaf_op():
    out = torch.zeros((cols, N), dtype=torch.int64, device="cuda")
    be.ntt_dev(OLA_NTT_COSET_LDE_LEAF_ORDER, x.data_ptr(), out.data_ptr(), log_n, cols, shift=7, blowup_log=3)
    torch.cuda.synchronize()
    return out
def describe(name, a, b):
    d = (a != b)
    print(name, "differ:", int(d.sum()), "per column (first 8 with any):", [(c, int(d[c].sum())) for c in range(cols) if bool(d[c].any())][:8],
          "per coset:", [int(d[:, k * n:(k + 1) * n].sum()) for k in range(8)], flush=True)
l1, l2 = leaf_op(), leaf_op()
describe("leaf op run 1 vs run 2", l1, l2)
nat = torch.zeros((cols, N), dtype=torch.int64, device="cuda")
scratch = torch.zeros_like(nat)
be.ntt_dev(OLA_NTT_COSET_LDE, x.data_ptr(), nat.data_ptr(), log_n, cols, shift=7, blowup_log=3, scratch_ptr=scratch.data_ptr())
torch.cuda.synchronize()
describe("leaf op vs scratch of the natural op", l1, scratch)
idx = torch.arange(N, device="cuda")
def brev(t, bits):
    r = torch.zeros_like(t)
    for b in range(bits):
        r |= ((t >> b) & 1) << (bits - 1 - b)
    return r
perm = brev(idx, log_n + 3)
for c in (0, 4, 5, 7, cols - 1):
    want = o.evaluate_poly_with_offset(x[c].cpu().numpy().view(np.uint64), 7, 8)
    w = torch.from_numpy(want.view(np.int64)).cuda()[perm]
    print("column", c, "vs oracle: leaf run 1", bool(torch.equal(l1[c], w)), " leaf run 2", bool(torch.equal(l2[c], w)), " natural-op scratch", bool(torch.equal(scratch[c], w)),
          " natural out", bool(torch.equal(nat[c][perm], w)), flush=True)
l3 = leaf_op()
describe("leaf op run 3 vs run 2", l3, l2)
