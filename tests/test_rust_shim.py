"""The Rust side of the boundary, checked mechanically (no Rust toolchain in this image, so nothing here is compiled):

  (i)   integration/rust/ola_gpu_sys.rs declares exactly the exports of include/ola_gpu.h -- same names, same arity, and per
        argument the same kind: pointer depth, constness of what is pointed to, integer width -- and the same struct layouts and
        constants;
  (ii)  integration/patches/0001-feature-hip.patch is what tools/make_hip_patch.py produces from /root/reference and
        integration/rust/*, and `git apply --check` accepts it on a copy of the reference's files;
  (iii) every reference location the header cites (file:line) exists, and where the comment names the function or type that
        lives there, that name is found at the cited lines.

(ii) and (iii) need /root/reference (present in the build container, not on the GPU box): skipped without it.
"""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HEADER = os.path.join(ROOT, "include", "ola_gpu.h")
SYS_RS = os.path.join(ROOT, "integration", "rust", "ola_gpu_sys.rs")
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")


def _strip_c_comments(s):
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    return re.sub(r"//[^\n]*", "", s)


C_INT = {"uint64_t": ("u", 64), "uint32_t": ("u", 32), "int32_t": ("i", 32), "uint8_t": ("u", 8), "size_t": ("u", "size"), "double": ("f", 64),
         "char": ("char", 8), "void": ("void", 0)}
RS_INT = {"u64": ("u", 64), "u32": ("u", 32), "i32": ("i", 32), "u8": ("u", 8), "usize": ("u", "size"), "f64": ("f", 64), "c_char": ("char", 8),
          "c_void": ("void", 0)}


def _c_kind(t):
    """C type -> (base, [constness of each pointer level, innermost first])"""
    t = " ".join(t.split())
    if t == "ola_all_gather_fn":
        return ("fnptr", [])
    m = re.match(r"^(const )?(\w+)\s*((?:\*\s*(?:const\s*)?)*)$", t)
    assert m, t
    base = C_INT.get(m.group(2), ("struct", m.group(2)))
    stars = re.findall(r"\*\s*(const)?", m.group(3))
    consts = []
    for i in range(len(stars)):
        consts.append(bool(m.group(1)) if i == 0 else bool(stars[i - 1]))
    return (base, consts)


def _rs_kind(t):
    t = " ".join(t.split())
    if t == "OlaAllGatherFn":
        return ("fnptr", [])
    consts = []
    while True:
        m = re.match(r"^\*(const|mut) (.*)$", t)
        if not m:
            break
        consts.append(m.group(1) == "const")
        t = m.group(2)
    consts.reverse()          # innermost first
    return (RS_INT.get(t, ("struct", t)), consts)


def c_prototypes():
    s = _strip_c_comments(open(HEADER).read())
    out = {}
    for ret, name, args in re.findall(r"\n\s*((?:const\s+)?[A-Za-z_][\w\s\*]*?)\b(ola_\w+)\s*\(([^;{]*?)\)\s*;", s, flags=re.S):
        kinds = []
        args = " ".join(args.split())
        if args != "void":
            for a in args.split(","):
                m = re.match(r"^(.*?)(\w+)\s*(\[\d*\])?$", a.strip())
                ty = m.group(1).strip() + ("*" if m.group(3) else "")        # an array parameter is a pointer
                kinds.append(_c_kind(ty))
        out[name] = (kinds, _c_kind(ret.strip()))
    return out


def rs_prototypes():
    s = re.sub(r"//[^\n]*", "", open(SYS_RS).read())
    block = re.search(r'extern "C" \{(.*?)\n\}', s, flags=re.S).group(1)
    out = {}
    for name, args, ret in re.findall(r"pub fn (\w+)\s*\((.*?)\)\s*->\s*([^;]+);", block, flags=re.S):
        kinds = []
        args = " ".join(args.split())
        if args:
            for a in args.split(","):
                kinds.append(_rs_kind(a.split(":", 1)[1].strip()))
        out[name] = (kinds, _rs_kind(ret.strip()))
    return out


def test_extern_block_matches_the_header():
    c, rs = c_prototypes(), rs_prototypes()
    assert len(c) >= 52
    assert sorted(c) == sorted(rs), (sorted(set(c) - set(rs)), sorted(set(rs) - set(c)))
    for name in c:
        (ck, cr), (rk, rr) = c[name], rs[name]
        assert len(ck) == len(rk), (name, "arity", len(ck), len(rk))
        for i, (a, b) in enumerate(zip(ck, rk)):
            assert a == b, (name, "argument %d" % i, a, b)
        assert cr == rr, (name, "return type", cr, rr)


def test_structs_and_constants_match_the_header():
    h = _strip_c_comments(open(HEADER).read())
    rs = open(SYS_RS).read()
    for struct in ("OlaGpuConfig", "OlaChallenger", "OlaScopeTime", "OlaPassTime"):
        cbody = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), h, flags=re.S).group(1)
        cf = []
        for ty, nm, arr in re.findall(r"([\w\s\*]+?)\s*\b(\w+)\s*(\[\d+\])?\s*;", cbody):
            k = _c_kind(" ".join(ty.split()))
            cf.append((nm, k, int(arr[1:-1]) if arr else 0))
        rbody = re.search(r"pub struct %s \{(.*?)\n\}" % struct, rs, flags=re.S).group(1)
        rf = []
        for nm, ty in re.findall(r"pub (\w+):\s*([^,\n]+),", rbody):
            m = re.match(r"^\[(\w+); (\d+)\]$", ty.strip())
            rf.append((nm, _rs_kind(m.group(1) if m else ty.strip()), int(m.group(2)) if m else 0))
        assert cf == rf, (struct, cf, rf)
    for name, val in re.findall(r"#define (OLA_\w+)\s+\(?(-?\d+)u?\)?", h):
        if name == "OLA_GPU_H" or name.startswith("OLA_PHASE_") and name != "OLA_PHASE_COUNT":
            continue
        m = re.search(r"pub const %s: \w+ = (-?\d+);" % name, rs)
        assert m, name + " is not declared in ola_gpu_sys.rs"
        assert int(m.group(1)) == int(val), (name, m.group(1), val)


@needs_ref
def test_patch_is_current_and_applies_to_the_reference(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_hip_patch as mk
    want = mk.make(REF)
    assert open(mk.PATCH).read() == want, "integration/patches/0001-feature-hip.patch is stale: run tools/make_hip_patch.py"
    for f in mk.EDITED:
        dst = tmp_path / f
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(os.path.join(REF, f), dst)
    r = subprocess.run(["git", "apply", "--check", "-p1", mk.PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(["git", "apply", "-p1", mk.PATCH], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for dst, src in mk.ADDED.items():
        assert (tmp_path / dst).read_text() == open(os.path.join(ROOT, "integration", "rust", src)).read()
    prover = (tmp_path / "circuits/src/stark/prover.rs").read_text()
    assert prover.count('#[cfg(feature = "hip")]') == 1 and "prove_with_traces_hip::<F, C, D>" in prover
    # the backend starts up where the reference starts its own GPU state: the line after init_gpu() in OlaStark::default()
    # (ola_stark.rs:47), i.e. before prove() generates the traces (client/src/main.rs:191-200)
    ref_lines = open(os.path.join(REF, "circuits/src/stark/ola_stark.rs")).read().split("\n")
    assert ref_lines[46].strip() == "plonky2::field::cfft::ntt::init_gpu();" and "fn default() -> Self" in ref_lines[45]
    stark = (tmp_path / "circuits/src/stark/ola_stark.rs").read_text().split("\n")
    assert stark[46].strip() == "plonky2::field::cfft::ntt::init_gpu();" and stark[47].strip() == '#[cfg(feature = "hip")]'
    assert stark[48].strip() == "super::hip_prover::init_early();"
    shim0 = open(os.path.join(ROOT, "integration", "rust", "hip_prover.rs")).read()
    assert "pub fn init_early()" in shim0 and "ola_gpu_warmup(-1, OLA_WARMUP_PINNED_RING, words.as_ptr(), words.len())" in shim0
    # the cached context is keyed by the configuration it was created with, and column counts are checked against the AIR set
    assert "*old_key != key" in shim0 and "ola_table_shape(c, words.as_ptr(), words.len(), t as u32" in shim0
    client = open(os.path.join(REF, "client/src/main.rs")).read().split("\n")
    assert "OlaStark::<F, D>::default()" in client[192] and "prove::<F, C, D>(" in client[194]
    # the caller's TimingTree goes in (no `let _ = timing;` any more), and the one method the shim needs exists after the patch
    hip_branch = prover[prover.index('#[cfg(feature = "hip")]'):prover.index("let rate_bits = config.fri_config.rate_bits;")]
    assert "timing," in hip_branch and "let _ = timing" not in hip_branch
    timing_rs = (tmp_path / "plonky2/plonky2/src/util/timing.rs").read_text()
    assert timing_rs.count("pub fn record(") == 2            # with and without the `timing` feature, like push / pop
    shim = open(os.path.join(ROOT, "integration", "rust", "hip_prover.rs")).read()
    assert "timing.record(" in shim and "timing: &mut TimingTree" in shim and "ola_prove_with_traces_cols(" in shim
    assert "c.values.as_ptr() as *const u64" in shim and "flat_map" not in shim          # one pointer per column, no gather
    # what the shim calls in the reference exists there under those names
    for path, needle in (("circuits/src/stark/serialization.rs", "pub fn read_all_proof<"), ("circuits/src/stark/serialization.rs", "pub fn new(buffer: Vec<u8>)"),
                         ("circuits/src/builtins/bitwise/bitwise_stark.rs", "pub fn get_compress_challenge(&self) -> Option<F>"),
                         ("circuits/src/program/program_stark.rs", "pub fn get_compress_challenge(&self) -> Option<F>"),
                         ("circuits/src/stark/ola_stark.rs", "Bitwise = 2,"), ("circuits/src/stark/ola_stark.rs", "Program = 10,"),
                         ("circuits/src/stark/ola_stark.rs", "pub(crate) const NUM_TABLES: usize = 12;"), ("plonky2/field/src/types.rs", "fn to_noncanonical_u64(&self) -> u64;"),
                         ("plonky2/plonky2/src/fri/mod.rs", "pub proof_of_work_bits: u32,"), ("circuits/src/stark/proof.rs", "pub public_values: PublicValues,"),
                         ("plonky2/plonky2/src/fri/mod.rs", "pub reduction_strategy: FriReductionStrategy,"),
                         ("plonky2/plonky2/src/fri/reduction_strategies.rs", "ConstantArityBits(usize, usize),"), ("circuits/src/stark/config.rs", "pub num_challenges: usize,"),
                         ("plonky2/field/src/polynomial/mod.rs", "pub values: Vec<F>,"), ("plonky2/field/src/goldilocks_field.rs", "#[repr(transparent)]"),
                         ("plonky2/plonky2/src/util/timing.rs", "children: Vec<TimingTree>,"), ("plonky2/plonky2/src/util/timing.rs", "exit_time: Option<Instant>,")):
        assert needle in open(os.path.join(REF, path)).read(), (path, needle)


@needs_ref
def test_scope_names_handed_to_the_timing_tree_are_the_references():
    """"emit the same scope names ... so numbers line up 1:1".  The names the library marks as the reference's
    (ScopeLog::is_reference_scope, olavm_amd/csrc/device_ctx.h) are exactly the `timed!` names of the path -- prover.rs:111-553,
    fri/oracle.rs:56-90,221-225, fri/prover.rs:41-58 -- minus "transpose LDEs" (oracle.rs:84; the LDE is produced in leaf order, there is
    no transpose to time); the formatted one is matched by its prefix."""
    ref = set()
    for path, lo, hi in (("circuits/src/stark/prover.rs", 111, 553), ("plonky2/plonky2/src/fri/oracle.rs", 56, 90), ("plonky2/plonky2/src/fri/oracle.rs", 221, 225),
                         ("plonky2/plonky2/src/fri/prover.rs", 41, 58)):
        text = "".join(open(os.path.join(REF, path)).readlines()[lo - 1:hi])
        for m in re.finditer(r'timed!\(\s*timing,\s*(?:&format!\()?"([^"]+)"', text):
            ref.add(m.group(1))
    assert len(ref) == 14 and "transpose LDEs" in ref and "perform final FFT {}" in ref
    src = open(os.path.join(ROOT, "olavm_amd", "csrc", "device_ctx.h")).read()
    body = src[src.index("static bool is_reference_scope"):src.index("struct DeviceCtx;")]
    ours = set(re.findall(r'"([^"]+)"', body[body.index("names[]"):body.index("for (const char* s : names)")]))
    prefix = re.search(r'n\.rfind\("([^"]+)", 0\) == 0', body).group(1)
    assert ours | {prefix + "{}"} == ref - {"transpose LDEs"}, (ours, ref)
    # and every scope the library opens under one of those names uses the same spelling (PhaseTimer call sites)
    opened = set()
    for f in ("stark.hip", "batch.hip", "fri.hip"):
        for m in re.finditer(r'PhaseTimer\w*\s*\w*\(ctx,\s*(?:nperm > 0 \? )?"\s*([^"]+)"', open(os.path.join(ROOT, "olavm_amd", "csrc", f)).read()):
            opened.add(m.group(1).strip())
    assert (ref - {"transpose LDEs", "perform final FFT {}"}) <= opened, (ref - opened)
    assert any(o.startswith(prefix.strip()) for o in opened)


def _resolve(path):
    """A cited path (possibly abbreviated: `prover.rs`, `fri/oracle.rs`) -> the reference file it means."""
    hits = []
    for base in ("", "circuits/src/stark/", "circuits/src/", "plonky2/plonky2/src/", "plonky2/plonky2/src/hash/", "plonky2/field/src/", "plonky2/"):
        p = os.path.join(REF, base, path)
        if os.path.isfile(p) and p not in hits:
            hits.append(p)
    return hits


@needs_ref
def test_every_reference_location_the_header_cites_exists():
    text = open(HEADER).read()
    cites = re.findall(r"([A-Za-z_0-9/]+\.rs):(\d[\d,\-]*\d|\d)", text)
    assert len(cites) >= 55
    for path, ranges in cites:
        files = _resolve(path)
        assert files, "cited file not found in the reference: " + path
        ok = False
        for f in files:
            n = sum(1 for _ in open(f, errors="replace"))
            if all(int(x) <= n for r in ranges.split(",") for x in r.split("-")):
                ok = True
        assert ok, f"{path}:{ranges} points past the end of the file"
    # named citations: `name` (file.rs:a-b) -- the name must occur at the cited lines (a few lines of slack for attributes / docs)
    named = re.findall(r"([A-Za-z_][\w:]*(?:<[\w, ]*>)?)`?\s*\(((?:[A-Za-z_0-9/]+\.rs:[\d,\-]+(?:,\s*|\s+and\s+)?)+)[;)]", text)
    checked = 0
    for name, cite in named:
        ident = re.sub(r"<.*>", "", name).split("::")[-1]
        # only what is spelled like a Rust item: a path, a snake_case name with an underscore, or a CamelCase type
        if not ("::" in name or re.match(r"^[a-z]+(_[a-z0-9]+)+$", ident) or re.match(r"^[A-Z][a-z0-9]+([A-Z][a-z0-9]*)+$", ident)):
            continue
        found_file = False
        for path, ranges in re.findall(r"([A-Za-z_0-9/]+\.rs):([\d,\-]+)", cite):
            for f in _resolve(path):
                lines = open(f, errors="replace").read().split("\n")
                for r in ranges.strip(",").split(","):
                    ab = [int(x) for x in r.split("-") if x]
                    a, b = ab[0], ab[-1]
                    if re.search(r"\b%s\b" % re.escape(ident), "\n".join(lines[max(0, a - 4):b + 3])):
                        found_file = True
        if any(_resolve(p) and re.search(r"\b%s\b" % re.escape(ident), open(_resolve(p)[0], errors="replace").read()) for p, _ in re.findall(r"([A-Za-z_0-9/]+\.rs):([\d,\-]+)", cite)):
            # the identifier is a thing of that file: then it has to be AT the cited lines
            assert found_file, f"`{name}` is not at {cite.strip()} any more"
            checked += 1
    assert checked >= 8, checked
