"""Soak run of tests/fuzz_cases.py::fuzz_allfocus (the pytest suite runs 40 cases with a fixed seed: tests/test_gpu_fuzz.py).
usage: python tools/fuzz_allfocus.py [cases] [seed]"""
import os
import sys
sys.path.insert(0, ".")
sys.path.insert(0, os.path.join(".", "tests"))
import fuzz_cases
import lfinterpolator_amd as L
from oracle import lfi_oracle_c as oc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bad = fuzz_cases.fuzz_allfocus(L, oc, n_cases, seed, log=lambda s: print(s, flush=True))
for b in bad:
    print("MISMATCH", b)
print("done:", n_cases, "cases,", len(bad), "mismatches")
sys.exit(1 if bad else 0)
