#!/usr/bin/env python3
"""Per-class kernel time against the number of windows in flight (minifam profiles, 3 kb reads):
what one more wavefront per SIMD costs each of its neighbours."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import deciphon_amd
from dcp_testlib import GOLDEN

eng = deciphon_amd.Engine(0)
eng.load_dcp(os.path.join(GOLDEN, "minifam.dcp"))
eng.commit()
eng.set_mode(True, False)
rng = np.random.default_rng(1)
reads = [rng.integers(0, 4, 3000).astype(np.uint8) for _ in range(5120)]
eng.set_sequences(reads)
for prof in (0, 1):
    K = eng.core_size(prof)
    for n in (512, 1024, 1536, 2048, 3072, 4096, 5120):
        wins = np.array([(prof, s, 0, 3000) for s in range(n)], dtype=np.int32)
        eng.stage(wins)
        eng.run_staged(2)
        ms, cells = eng.run_staged(10)
        print(f"K={K} windows={n:5d} ({n / 1024:.1f} per SIMD)  {ms / 10:7.3f} ms  {cells / (ms / 10 * 1e-3) / 1e9:7.1f} GCUPS", flush=True)
