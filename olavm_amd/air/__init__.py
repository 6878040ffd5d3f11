"""AIR descriptions as data: a small expression DSL that mirrors the reference's `eval_packed_generic` bodies and
compiles them to a flat u64 "AIR-set blob" interpreted by the HIP quotient kernel (and, independently, by the oracle)."""
from .dsl import AirTable, AirSet, Col, CrossTableLookup, TableWithColumns, P  # noqa: F401
