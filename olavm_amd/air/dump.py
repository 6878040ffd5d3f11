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
