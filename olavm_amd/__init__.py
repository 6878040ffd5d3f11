"""olavm_amd -- MI355X-native Goldilocks STARK proving backend for OlaVM (hot path behind
circuits::stark::prover::prove_with_traces).  The product is the HIP library `lib/libola_gpu.so`
(sources in csrc/, C ABI in include/ola_gpu.h); this package is the thin ctypes plumbing used by the
tests and bench.py.  There is deliberately no CPU fallback: importing `olavm_amd.backend` without the
built library, or creating a context without a GPU, fails loudly."""
from .backend import Backend, Batch, Challenger, OlaGpuError, load_library, lib_path  # noqa: F401
