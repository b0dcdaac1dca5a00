#!/usr/bin/env python3
"""Audit of the tiled kernel's accumulator file (fusion_tile_acc.inc) by hand:  python tools/check_acc_registers.py
The same audit runs inside cudadepthmapintegration_amd.build.build() whenever fusion_tile.hip is recompiled and in
tests/test_abi.py; see build.audit_accumulator_registers for what it checks."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cudadepthmapintegration_amd import build  # noqa: E402

if __name__ == "__main__":
    n = build.run_accumulator_audit(verbose=True)
    print(f"checked {n} fuse_tile_kernel instantiations, 0 violations")
