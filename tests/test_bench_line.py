"""The line bench.py prints last is what the driver parses (round 5's grew to 22 KB and was lost: BENCH_r05.json `parsed: null`).
These tests build it from a canned full result -- round 5's own 22 KB record, profiles/r05_bench_final.json, plus the blocks added
since -- and hold it to the contract: one line, under 4096 bytes, the required keys, numbers that agree with each other."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")


def canned():
    d = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_final.json")))
    # what round 6 adds to the full record
    d["roofline"].update({"bound_short": "hbm by contract; VALU issue binds (64-bit modmul on a 32-bit pipe)", "kernel_short": "ntt2t_pass_kernel<8,2,false,8,2>",
                          "valu_insts_per_element": 305.0, "valu_floor_ms": 3.058, "frac_of_floor": 0.76,
                          "per_pass_avg_ms": {"ntt2t_pass_kernel<7,0,false,8,0>": 1.25, "ntt2t_pass_kernel<7,0,false,8,1>": 1.31, "ntt2t_pass_kernel<8,2,false,8,2>": 1.63},
                          "dominant": {"whole_transform_ms": 4.02, "whole_transform_frac": 0.196}})
    d["cpu_baseline"]["prove"]["log_n"] = 17
    d["cpu_baseline"]["prove_small"] = dict(d["cpu_baseline"]["prove"], log_n=14)
    d["start"] = {"init_ms": 880.0, "early_hook_excess_over_warm": 1.1, "lazy_excess_over_warm": 2.4}
    return d


def test_compact_line_is_one_short_line_with_the_contract_keys():
    full = canned()
    assert len(json.dumps(full)) > 20000              # the record that broke the driver's parser
    line = json.dumps(bench.compact_line(full), separators=(",", ":"))
    assert "\n" not in line and len(line) < bench.LINE_LIMIT == 4096, len(line)
    c = json.loads(line)
    for k in REQUIRED:
        assert k in c, k
    assert c["metric"] == "goldilocks_ntt_throughput" and c["unit"] == "GB/s" and c["dtype"] == "u64" and c["vs_baseline"] is None
    assert set(c["config"]) >= {"workload", "log_n", "columns_per_gpu", "parallelism"} and "model" not in c["config"]
    r = c["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "valu_floor_ms", "frac_of_floor"):
        assert k in r, k
    assert r["bound"].startswith("hbm") and "VALU" in r["bound"] and len(r["bound"]) <= 80 and len(r["kernel"]) <= 80
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["peak"] == 8000.0
    b = c["cpu_baseline"]
    assert set(b) >= {"value", "unit", "cores", "kind", "sample", "prove", "prove_small"} and b["kind"] == "port" and len(b["sample"]) <= 80
    assert b["prove"]["log_n"] == 17 and b["prove_small"]["log_n"] == 14
    assert set(c["proofs"]) >= {"poseidon_2p22", "blake3_2p22", "real_poseidon_2p22", "real_blake3_2p22", "readme_fibo_blake3", "config4", "2p24"}
    assert all(isinstance(v, (int, float)) for v in c["proofs"].values()) and c["proofs_verified"] is True
    assert c["roofline_lde"]["frac"] == full["roofline_lde"]["frac"]
    # every string in the line is short: prose lives in bench_details.json
    def strings(x):
        if isinstance(x, dict):
            for v in x.values():
                yield from strings(v)
        elif isinstance(x, str):
            yield x
    assert max(len(t) for t in strings(c)) <= 100
    # the timing the driver cross-checks: steps x ms_per_step is the timed region, and the value follows from it
    assert 0 < c["ms_per_step"] * c["steps"] < 60_000
    alg = 16.0 * (1 << c["config"]["log_n"]) * c["config"]["columns_per_gpu"] * c["n_gpus"]
    assert abs(c["value"] - alg / (c["ms_per_step"] * 1e-3) / 1e9) / c["value"] < 0.01


def test_emit_prints_the_compact_line_last_and_writes_the_details(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "DETAILS_FILE", str(tmp_path / "bench_details.json"))
    full = canned()
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(full)
    lines = buf.getvalue().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096 and json.loads(lines[0])["details"] == "bench_details.json"
    assert json.load(open(tmp_path / "bench_details.json")) == full          # nothing is lost: the long record is in the side file


def test_a_result_with_failed_extras_still_yields_a_parseable_line():
    full = canned()
    for k in ("prove", "prove_real_execution", "readme_fibo_loop_blake3", "config4_poseidon_heavy", "prove_2p24_rows"):
        full[k] = {"error": "x" * 300}
    full["cpu_baseline"]["prove"] = {"error": "y" * 300}
    full["roofline_lde"] = {"error": "z" * 300}
    line = json.dumps(bench.compact_line(full), separators=(",", ":"))
    c = json.loads(line)
    assert len(line) < 4096 and "proofs" not in c and "roofline_lde" not in c and len(c["cpu_baseline"]["prove"]["error"]) <= 80
    for k in REQUIRED:
        assert k in c


def test_valu_floor_is_the_stated_product():
    # 305 instructions per element x 94 x 2^22 elements / 64 lanes x 4.0 cycles / (256 x 4 x 2.4e9 SIMD-cycles/s)
    ms = bench.valu_floor_ms(305.0, 94 * (1 << 22))
    assert abs(ms - 305.0 * 94 * (1 << 22) / 64 * 4.0 / (256 * 4 * 2.4e9) * 1e3) < 1e-9 and 3.0 < ms < 3.1
