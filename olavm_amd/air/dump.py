"""Writes the AIR-set blob of the reference's 12-table OlaStark (what `ola_prove_with_traces` takes as `airset`) to a file of
little-endian u64 words, for a host binding that embeds it.  usage: python -m olavm_amd.air.dump out.bin"""
import sys

from . import ola_tables


def main(argv):
    path = argv[1] if len(argv) > 1 else "ola_airset.bin"
    blob = ola_tables.ola_stark().blob()
    blob.astype("<u8").tofile(path)
    print("%s: %d words, %d tables, %d cross-table lookups" % (path, blob.size, int(blob[2]), int(blob[3])))


if __name__ == "__main__":
    main(sys.argv)


def columns_header():
    """C++ header with every column index / table width of olavm_amd/air/ola_tables.py (ranges as NAME_START / NAME_END),
    the opcode bit positions and the memory-region constants -- what the native trace generator
    (olavm_amd/csrc/host/tracegen.cpp) shares with the Python table descriptions.  Written at build time."""
    from . import ola_tables as T
    out = ["// generated from olavm_amd/air/ola_tables.py by olavm_amd.air.dump.columns_header() -- do not edit",
           "#pragma once", "#include <cstdint>", "namespace olacols {"]
    for name in sorted(n for n in dir(T) if n.isupper()):
        v = getattr(T, name)
        if isinstance(v, bool):
            continue
        if isinstance(v, int):
            out.append("constexpr uint64_t %s = %dull;" % (name, v % (1 << 64)))
        elif isinstance(v, range):
            out.append("constexpr uint64_t %s_START = %dull, %s_END = %dull;" % (name, v.start, name, v.stop))
    for op, sh in sorted(T.OPCODE_SHIFT.items()):
        out.append("constexpr uint32_t OP_%s = %d;" % (op, sh))
    out.append("}  // namespace olacols")
    return "\n".join(out) + "\n"
