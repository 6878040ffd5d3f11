"""Workload for tools/pmc.sh: 3 forward NTTs of the bench shape (94 x 2^22) and one 94 x 2^19 commitment whose
canonicalize_kernel (one 8-byte read and write per element) calibrates FETCH_SIZE / WRITE_SIZE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from olavm_amd.backend import Backend, OLA_NTT_EVALUATE
be = Backend(device=0)
g = torch.Generator(device="cuda").manual_seed(1)
data = torch.randint(0, 2**63 - 1, (94, 1 << 22), dtype=torch.int64, device="cuda", generator=g)
out, scratch = torch.empty_like(data), torch.empty_like(data)
torch.cuda.synchronize()      # the library runs on a stream of its own
for _ in range(3):
    be.ntt_dev(OLA_NTT_EVALUATE, data.data_ptr(), out.data_ptr(), 22, 94, scratch_ptr=scratch.data_ptr())
torch.cuda.synchronize()
b = be.commit_dev(data.data_ptr(), 94, 19)      # 2^19: its pass kernels (6 + 6 + 7 bits) are not the 2^22 transform's (7 + 7 + 8)
torch.cuda.synchronize()
b.free()
be.close()
