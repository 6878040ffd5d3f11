"""the standalone-NTT matrix on one MI355X.

log_n in {20..24} x batch in {1, 32, 94} x {forward NTT (evaluate_poly), inverse NTT (interpolate_poly), coset LDE x8
(evaluate_poly_with_offset, shift 7; in commitment-leaf order as the prover uses it, and in natural order)}, operands resident in HBM, through the C ABI (ola_ntt_batch_dev).  Inputs: a
splitmix64 stream per column (seed 0x01A5EED + column), reduced to canonical form; plus the adversarial tiling
{0, 1, p-1, 2^32-1, 2^32, 0xFFFFFFFF00000000}.  Algorithmic bytes: 16*n*B for NTT / iNTT, 72*n*B for the LDE.

Full-size parity properties (size-independent, bit-exact): interpolate(evaluate(x)) == x, and
coset_interpolate_{8n}(coset_lde_8(c)) == c || 0...0 -- the second one inverts the whole low-degree extension with a
transform of a different size and a different pass structure.  Also measures a device-to-device copy of the largest
batch (the practical HBM ceiling next to the 8 TB/s spec figure).

    python tools/bench_ntt_matrix.py [--log-n 20 21 22 23 24] [--cols 1 32 94] [--reps 5] [--out gpurun_out/ntt_matrix.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

HBM_PEAK = 8.0e12
from tests.inputs import P, s64 as _s64, splitmix_columns, adversarial_columns  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, nargs="+", default=[20, 21, 22, 23, 24])
    ap.add_argument("--cols", type=int, nargs="+", default=[1, 32, 94])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default="gpurun_out/ntt_matrix.json")
    ap.add_argument("--check-limit-gb", type=float, default=230.0, help="skip a property check that would need more HBM than this")
    args = ap.parse_args()

    import torch
    from olavm_amd.backend import (Backend, OLA_NTT_EVALUATE, OLA_NTT_INTERPOLATE, OLA_NTT_COSET_LDE, OLA_NTT_COSET_INTERPOLATE,
                                   OLA_NTT_COSET_LDE_LEAF_ORDER)
    stream = torch.cuda.current_stream()
    be = Backend(device=0, stream=stream.cuda_stream)
    rows, checks = [], []

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps

    for log_n in args.log_n:
        n = 1 << log_n
        for cols in args.cols:
            data = splitmix_columns(torch, cols, n)
            torch.cuda.empty_cache()
            out = torch.empty_like(data)
            scratch = torch.empty_like(data)
            lde = torch.empty((cols, 8 * n), dtype=torch.int64, device="cuda")
            lde_scratch = torch.empty_like(lde)
            torch.cuda.synchronize()          # the library's stream is not torch's: inputs must be complete before it is called
            ops = [("ntt", OLA_NTT_EVALUATE, data, out, scratch, log_n, 0, 16),
                   ("intt", OLA_NTT_INTERPOLATE, data, out, scratch, log_n, 0, 16),
                   ("coset_lde8_leaf_order", OLA_NTT_COSET_LDE_LEAF_ORDER, data, lde, lde_scratch, log_n, 3, 72),   # what commitments use
                   ("coset_lde8_natural", OLA_NTT_COSET_LDE, data, lde, lde_scratch, log_n, 3, 72)]
            for name, op, src, dst, scr, ln, blow, bytes_per in ops:
                ms = timed(lambda: be.ntt_dev(op, src.data_ptr(), dst.data_ptr(), ln, cols, shift=7, blowup_log=blow, scratch_ptr=scr.data_ptr()))
                alg = bytes_per * n * cols
                rows.append({"op": name, "log_n": log_n, "cols": cols, "ms": round(ms, 4), "algorithmic_bytes": alg,
                             "achieved_GBps": round(alg / ms / 1e6, 1), "frac_of_hbm_peak": round(alg / (ms * 1e-3) / HBM_PEAK, 4)})
                print(json.dumps(rows[-1]), flush=True)
            # ---- full-size properties
            need_gb = (3 * 8 * n + 4 * n) * cols * 8 / 1e9
            lde_check = need_gb < args.check_limit_gb
            if not lde_check:
                del lde, lde_scratch
                lde = lde_scratch = None
                torch.cuda.empty_cache()
            for label, x in (("splitmix64", data), ("adversarial", adversarial_columns(torch, cols, n))):
                torch.cuda.synchronize()
                be.ntt_dev(OLA_NTT_EVALUATE, x.data_ptr(), out.data_ptr(), log_n, cols, scratch_ptr=scratch.data_ptr())
                back = torch.empty_like(x)
                be.ntt_dev(OLA_NTT_INTERPOLATE, out.data_ptr(), back.data_ptr(), log_n, cols, scratch_ptr=scratch.data_ptr())
                torch.cuda.synchronize()
                canon = torch.where((x < 0) & (x >= _s64(P)), x - _s64(P), x)
                ok = bool(torch.equal(back, canon))
                checks.append({"property": "interpolate(evaluate(x)) == x", "inputs": label, "log_n": log_n, "cols": cols, "ok": ok})
                del back
                if lde_check:
                    be.ntt_dev(OLA_NTT_COSET_LDE, x.data_ptr(), lde.data_ptr(), log_n, cols, shift=7, blowup_log=3, scratch_ptr=lde_scratch.data_ptr())
                    coeffs = torch.empty_like(lde)
                    torch.cuda.synchronize()
                    be.ntt_dev(OLA_NTT_COSET_INTERPOLATE, lde.data_ptr(), coeffs.data_ptr(), log_n + 3, cols, shift=7, scratch_ptr=lde_scratch.data_ptr())
                    torch.cuda.synchronize()
                    ok = bool(torch.equal(coeffs[:, :n], canon)) and not bool(coeffs[:, n:].any())
                    checks.append({"property": "coset_interpolate_8n(coset_lde_8(c)) == c || 0", "inputs": label, "log_n": log_n, "cols": cols, "ok": ok})
                    del coeffs
                print(json.dumps(checks[-1]), flush=True)
            del data, out, scratch, lde, lde_scratch, ops, x, canon, src, dst, scr
            torch.cuda.empty_cache()
            be.trim()

    # ---- device-to-device copy of 94 x 2^24 (12.6 GB read + 12.6 GB written)
    a = torch.empty((94, 1 << 24), dtype=torch.int64, device="cuda").fill_(3)
    b = torch.empty_like(a)
    ms = timed(lambda: b.copy_(a))
    copy = {"d2d_copy_bytes": 2 * a.numel() * 8, "ms": round(ms, 3), "GBps": round(2 * a.numel() * 8 / ms / 1e6, 1)}
    print(json.dumps(copy), flush=True)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump({"device": torch.cuda.get_device_name(0), "reps": args.reps, "hbm_peak_Bps": HBM_PEAK, "rows": rows, "checks": checks,
                   "d2d_copy": copy, "all_checks_ok": all(c["ok"] for c in checks), "timestamp": time.strftime("%Y-%m-%d %H:%M:%S")}, f, indent=1)
    print("all checks ok" if all(c["ok"] for c in checks) else "CHECK FAILURES")
    if not all(c["ok"] for c in checks):
        raise SystemExit(1)


if __name__ == "__main__":
    main()
