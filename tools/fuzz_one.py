#!/usr/bin/env python3
"""One seed of fuzz_sequence.py / fuzz_parity.py, a few times over (is a failure reproducible?):
    python tools/fuzz_one.py sequence|parity SEED [repeats]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
mod = importlib.import_module("fuzz_" + sys.argv[1])
seed = int(sys.argv[2])
for k in range(int(sys.argv[3]) if len(sys.argv) > 3 else 5):
    print(k, mod.one_case(seed), flush=True)
