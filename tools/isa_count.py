#!/usr/bin/env python3
"""Static instruction mix of the gfx950 kernels in a HIP source: compiles the device side to assembly
(hipcc --offload-device-only -S) and counts, per kernel, the VALU instructions by issue class, the scalar, LDS and
memory instructions, the wait states, registers and LDS bytes.  Loops are counted ONCE (static count): the transform and
hash kernels this is used on are fully unrolled apart from their column loop.

Issue classes are the ones tools/ubench/valu_rates.hip measured on the MI355X (profiles/r02_valu_rates_waves.txt, 8 waves per
SIMD): "fast" = 32-bit VOP2-style moves / adds / logic / right shifts (2.4 - 2.9 cycles per wave64 instruction), "slow" =
everything else on the VALU (64-bit adds, carries, multiplies, multiply-adds, left shifts, VOP3 three-operand forms, compares,
selects: 4.2 - 4.8 cycles).  `cycles` = 2.6 * fast + 4.6 * slow is the issue-time model used in DESIGN.md.

    python tools/isa_count.py olavm_amd/csrc/ola_gpu.hip --filter ntt2_pass_kernel [--per N] [-D NAME=VAL ...]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

FAST = {
    "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32",
    "v_ashrrev_i32", "v_lshrrev_b32", "v_fma_f32", "v_add_f32", "v_mul_f32", "v_max_u32", "v_min_u32", "v_max_i32", "v_min_i32",
    "v_xnor_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_accvgpr_mov_b32",
}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_asm(src, defines, extra):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--offload-device-only", "-S", "-o", out, src,
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "olavm_amd", "csrc")] + ["-D" + d for d in defines] + extra
    subprocess.check_call(cmd)
    return out


def parse(path):
    kernels, cur, name = {}, None, None
    meta = {}
    for line in open(path):
        s = line.strip()
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", s)
        if m and not s.startswith(".L") and not s.startswith("BB"):
            name = m.group(1)
            cur = kernels.setdefault(name, {"fast": 0, "slow": 0, "salu": 0, "lds": 0, "vmem": 0, "smem": 0, "nop_states": 0, "waitcnt": 0,
                                            "other": 0, "ops": {}})
            continue
        if s.startswith(".amdhsa_kernel"):
            mk = s.split()[1]
            meta[mk] = {}
            cur_meta = meta[mk]
            continue
        if s.startswith(".amdhsa_next_free_vgpr") and meta:
            cur_meta["vgpr"] = int(s.split()[1])
        if s.startswith(".amdhsa_accum_offset") and meta:
            cur_meta["accum_offset"] = int(s.split()[1])
        if s.startswith(".amdhsa_group_segment_fixed_size") and meta:
            cur_meta["lds"] = int(s.split()[1])
        if s.startswith(".amdhsa_private_segment_fixed_size") and meta:
            cur_meta["scratch"] = int(s.split()[1])
        if cur is None or not s or s.startswith(".") or s.startswith(";") or s.startswith("//"):
            continue
        if s.endswith(":"):
            continue
        op = s.split()[0]
        base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)
        if op.startswith("v_"):
            if base in FAST and not op.endswith("_sdwa"):
                cur["fast"] += 1
            else:
                cur["slow"] += 1
            cur["ops"][base] = cur["ops"].get(base, 0) + 1
        elif op == "s_nop":
            cur["nop_states"] += int(s.split()[1]) + 1
        elif op == "s_waitcnt":
            cur["waitcnt"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            cur["smem"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
            cur["vmem"] += 1
            if op.startswith("scratch_"):
                cur["ops"]["scratch"] = cur["ops"].get("scratch", 0) + 1
        else:
            cur["other"] += 1
    return kernels, meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("--filter", default="")
    ap.add_argument("--per", type=float, default=0, help="divide the counts by this many elements (e.g. 16 * columns-per-loop)")
    ap.add_argument("-D", action="append", default=[])
    ap.add_argument("--top", type=int, default=0, help="print the N most frequent VALU opcodes")
    ap.add_argument("--asm", default="", help="read this assembly file instead of compiling")
    ap.add_argument("extra", nargs="*")
    a = ap.parse_args()
    path = a.asm or compile_asm(a.src, a.D, a.extra)
    if not a.asm:
        print("asm:", path, file=sys.stderr)
    kernels, meta = parse(path)
    for name, k in kernels.items():
        if a.filter and a.filter not in name:
            continue
        valu = k["fast"] + k["slow"]
        if valu == 0:
            continue
        cyc = 2.6 * k["fast"] + 4.6 * k["slow"]
        m = meta.get(name, {})
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        line = f"{dem[:110]}\n   VALU {valu} (fast {k['fast']}, slow {k['slow']}; model {cyc:.0f} cycles)  SALU {k['salu']}  LDS {k['lds']}  VMEM {k['vmem']}  SMEM {k['smem']}  s_nop states {k['nop_states']}  waitcnt {k['waitcnt']}"
        line += f"\n   vgpr {m.get('vgpr')} (arch {m.get('accum_offset')})  lds {m.get('lds')} B  scratch {m.get('scratch')} B"
        if a.per:
            line += f"\n   per element (/{a.per:g}): VALU {valu / a.per:.1f}  fast {k['fast'] / a.per:.1f}  slow {k['slow'] / a.per:.1f}  model cycles {cyc / a.per:.1f}"
        print(line)
        if a.top:
            for op, n in sorted(k["ops"].items(), key=lambda t: -t[1])[:a.top]:
                print(f"      {op:28s} {n}")
    if not a.asm:
        print("asm:", path, file=sys.stderr)


if __name__ == "__main__":
    main()
