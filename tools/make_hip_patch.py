#!/usr/bin/env python3
"""integration/patches/0001-feature-hip.patch from the reference tree and integration/rust/*: the patch a maintainer applies to
Sin7Y/olavm to put the MI355X backend behind `circuits::stark::prover::prove_with_traces` (cargo feature `hip` on the `circuits`
crate; the executor / client path is untouched).

    python tools/make_hip_patch.py [--reference /root/reference] [--check]      # --check: compare with the committed patch

The patch adds circuits/build.rs, circuits/src/stark/{ola_gpu_sys,hip_prover}.rs (verbatim copies of integration/rust/*) and edits
five files: circuits/Cargo.toml (feature + build script), circuits/src/stark/mod.rs (two `mod` lines),
circuits/src/stark/prover.rs (the `#[cfg(feature = "hip")]` branch at the top of prove_with_traces, prover.rs:79-105),
circuits/src/stark/ola_stark.rs (the early start-up next to the reference's own init_gpu(), ola_stark.rs:47) and
plonky2/plonky2/src/util/timing.rs (`TimingTree::record`: the GPU's scope times enter the caller's tree through it).
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "integration", "patches", "0001-feature-hip.patch")
EDITED = ["circuits/Cargo.toml", "circuits/src/stark/mod.rs", "circuits/src/stark/prover.rs", "circuits/src/stark/ola_stark.rs",
          "plonky2/plonky2/src/util/timing.rs"]
ADDED = {"circuits/build.rs": "build.rs", "circuits/src/stark/ola_gpu_sys.rs": "ola_gpu_sys.rs", "circuits/src/stark/hip_prover.rs": "hip_prover.rs"}


def replace_once(text, old, new, what):
    if text.count(old) != 1:
        raise SystemExit(f"make_hip_patch: the reference no longer has exactly one `{what}` anchor (found {text.count(old)})")
    return text.replace(old, new)


def edit(path, text):
    if path == "circuits/Cargo.toml":
        text = replace_once(text, 'edition = "2021"\n', 'edition = "2021"\nbuild = "build.rs"\n', "edition line")
        return replace_once(text, "benchmark = []\n", "benchmark = []\n# prove_with_traces on an MI355X through libola_gpu.so (see circuits/build.rs "
                            "for the environment it needs)\nhip = []\n", "benchmark feature")
    if path == "circuits/src/stark/mod.rs":
        text = replace_once(text, "mod get_challenges;\n", 'mod get_challenges;\n#[cfg(feature = "hip")]\npub mod hip_prover;\n', "mod get_challenges")
        return replace_once(text, "pub mod ola_stark;\n", '#[cfg(feature = "hip")]\npub mod ola_gpu_sys;\npub mod ola_stark;\n', "mod ola_stark")
    if path == "circuits/src/stark/prover.rs":
        old = ("    [(); ProgChunkStark::<F, D>::COLUMNS]:,\n{\n    let rate_bits = config.fri_config.rate_bits;\n"
               "    let cap_height = config.fri_config.cap_height;\n\n    let mut twiddle_map = BTreeMap::new();\n")
        new = ("    [(); ProgChunkStark::<F, D>::COLUMNS]:,\n{\n"
               "    // The MI355X backend proves all twelve tables in one call and returns the AllProof in the wire format of\n"
               "    // serialization.rs; everything below is the CPU prover.\n"
               '    #[cfg(feature = "hip")]\n    {\n'
               "        return super::hip_prover::prove_with_traces_hip::<F, C, D>(\n            ola_stark,\n            config,\n"
               "            &trace_poly_values,\n            public_values,\n            timing,\n        );\n    }\n\n"
               "    let rate_bits = config.fri_config.rate_bits;\n    let cap_height = config.fri_config.cap_height;\n\n"
               "    let mut twiddle_map = BTreeMap::new();\n")
        return replace_once(text, old, new, "head of prove_with_traces")
    if path == "circuits/src/stark/ola_stark.rs":
        # ola_stark.rs:47: where the reference brings ITS GPU state up (init_gpu, cfft/ntt/mod.rs:53-99), before prove() generates
        # the traces (client/src/main.rs:191-200) -- the backend's start-up goes to the same place and overlaps the trace generation
        return replace_once(text, "        plonky2::field::cfft::ntt::init_gpu();\n",
                            "        plonky2::field::cfft::ntt::init_gpu();\n"
                            '        #[cfg(feature = "hip")]\n        super::hip_prover::init_early();\n', "init_gpu() call in OlaStark::default()")
    if path == "plonky2/plonky2/src/util/timing.rs":
        # TimingTree's fields are private and push / pop read the host's clock: a scope that ran on the GPU and is already over
        # needs one method that appends a closed node with given times
        old = "    #[cfg(feature = \"timing\")]\n    fn duration(&self) -> Duration {\n"
        new = ('    /// A scope that ran elsewhere (on a GPU) and is over: appended `depth` levels below the deepest open\n'
               '    /// scope -- following the scopes recorded last -- with the times given.  Scopes arrive parents first.\n'
               '    #[cfg(feature = "timing")]\n'
               '    pub fn record(&mut self, depth: usize, ctx: &str, mut level: log::Level, enter_time: Instant, duration: Duration) {\n'
               '        assert!(self.is_open());\n\n'
               '        level = level.max(self.level);\n\n'
               '        if let Some(last_child) = self.children.last_mut() {\n'
               '            if last_child.is_open() {\n'
               '                last_child.record(depth, ctx, level, enter_time, duration);\n'
               '                return;\n'
               '            }\n'
               '        }\n\n'
               '        let mut node = self;\n'
               '        for _ in 0..depth {\n'
               '            node = node.children.last_mut().expect("recorded scopes arrive parents first");\n'
               '        }\n'
               '        node.children.push(TimingTree {\n'
               '            name: ctx.to_string(),\n'
               '            level,\n'
               '            enter_time,\n'
               '            exit_time: Some(enter_time + duration),\n'
               '            children: vec![],\n'
               '        })\n'
               '    }\n\n'
               '    #[cfg(not(feature = "timing"))]\n'
               '    pub fn record(\n'
               '        &mut self,\n'
               '        _depth: usize,\n'
               '        _ctx: &str,\n'
               '        _level: log::Level,\n'
               '        _enter_time: std::time::Instant,\n'
               '        _duration: std::time::Duration,\n'
               '    ) {\n'
               '    }\n\n') + old
        return replace_once(text, old, new, "TimingTree::duration")
    raise KeyError(path)


def make(reference):
    tmp = tempfile.mkdtemp(prefix="hip_patch_")
    try:
        for side in ("a", "b"):
            for f in EDITED:
                dst = os.path.join(tmp, side, f)
                os.makedirs(os.path.dirname(dst), exist_ok=True)
                shutil.copy(os.path.join(reference, f), dst)
        for f in EDITED:
            p = os.path.join(tmp, "b", f)
            text = edit(f, open(p).read())
            open(p, "w").write(text)
        for dst, src in ADDED.items():
            shutil.copy(os.path.join(ROOT, "integration", "rust", src), os.path.join(tmp, "b", dst))
        out = subprocess.run(["diff", "-ruN", "a", "b"], cwd=tmp, capture_output=True, text=True).stdout
        # no timestamps in the headers: the patch is a function of the two trees
        return re.sub(r"^(---|\+\+\+) (\S+)\t.*$", r"\1 \2", out, flags=re.M)
    finally:
        shutil.rmtree(tmp)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    text = make(a.reference)
    if a.check:
        if open(PATCH).read() != text:
            raise SystemExit("integration/patches/0001-feature-hip.patch is stale: run tools/make_hip_patch.py")
        print("patch is up to date")
        return
    open(PATCH, "w").write(text)
    print("wrote", PATCH, f"({text.count(chr(10))} lines)")


if __name__ == "__main__":
    main()
