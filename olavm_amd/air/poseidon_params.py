"""Poseidon-Goldilocks parameter tables for the Poseidon AIR, parsed from the generated header
include/ola_poseidon_constants.h (tools/gen_poseidon_tables.py).  The partial-round tables are our own factorisation;
any valid factorisation yields the same constraint POLYNOMIALS (lane 0 is untouched by the change of basis and the
final state coincides), see DESIGN.md."""
import os
import re

_HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "include", "ola_poseidon_constants.h")


def _arr(text, name):
    m = re.search(name + r"\[\d+\]\s*=\s*\{(.*?)\};", text, re.S)
    return [int(x.rstrip("ul"), 16) if x.startswith("0x") else int(x) for x in re.findall(r"0x[0-9a-fA-F]+|\b\d+\b", m.group(1))]


_t = open(_HDR).read()
RC = _arr(_t, "OLA_POSEIDON_RC")                  # 360
MDS_CIRC = _arr(_t, "OLA_POSEIDON_MDS_CIRC")      # 12
MDS_DIAG = _arr(_t, "OLA_POSEIDON_MDS_DIAG")      # 12
FAST_FIRST_C = _arr(_t, "OLA_POSEIDON_FAST_FIRST_C")  # 12
FAST_POST_C = _arr(_t, "OLA_POSEIDON_FAST_POST_C")    # 22
FAST_VHAT = _arr(_t, "OLA_POSEIDON_FAST_VHAT")        # 22 x 11
FAST_W = _arr(_t, "OLA_POSEIDON_FAST_W")              # 22 x 11
FAST_INIT = _arr(_t, "OLA_POSEIDON_FAST_INIT")        # 11 x 11 (row-major, y[r] = sum_c INIT[r][c] x[c])
assert len(RC) == 360 and len(FAST_VHAT) == 242 and len(FAST_INIT) == 121
