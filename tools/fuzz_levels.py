"""Randomised level-jump programmes through the packed N = 4096 kernels, many seeds (the committed tests run the first few): every row against the oracle at the plain bar.  python tools/fuzz_levels.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import test_gpu_dynamic_range as T
from oracle import pyoracle as po
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for seed in range(10, 10 + N):
    try:
        T.test_random_level_programmes_every_row_at_its_own_bar(po, seed)
    except AssertionError as e:
        bad += 1; print("FAIL ms", seed, str(e)[:300])
for seed in range(4, 4 + max(N // 4, 1)):
    try:
        T.test_random_level_programmes_pair_kernel(po, seed)
    except AssertionError as e:
        bad += 1; print("FAIL pair", seed, str(e)[:300])
print("failures", bad)
