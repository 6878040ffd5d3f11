"""Workload for tools/pmc_sq.sh: two forward NTTs and one x8 leaf-order LDE of 94 x 2^22 -- every T-form pass kernel of the 2^22
transforms (strided without / with load multipliers, closing natural-order, closing bit-reversed) under the SQ counters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from olavm_amd.backend import Backend, OLA_NTT_EVALUATE, OLA_NTT_COSET_LDE_LEAF_ORDER
be = Backend(device=0)
g = torch.Generator(device="cuda").manual_seed(1)
data = torch.randint(0, 2**63 - 1, (94, 1 << 22), dtype=torch.int64, device="cuda", generator=g)
out, scratch = torch.empty_like(data), torch.empty_like(data)
lde = torch.empty((94, 8 << 22), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for _ in range(2):
    be.ntt_dev(OLA_NTT_EVALUATE, data.data_ptr(), out.data_ptr(), 22, 94, scratch_ptr=scratch.data_ptr())
be.ntt_dev(OLA_NTT_COSET_LDE_LEAF_ORDER, data.data_ptr(), lde.data_ptr(), 22, 94, shift=7, blowup_log=3)
torch.cuda.synchronize()
be.close()
