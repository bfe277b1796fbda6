#!/usr/bin/env python3
"""How much of the bench step is memory latency?  The same windows once with random reads and once
with poly-A reads: identical instruction stream, but every emission-row load of the poly-A run hits
the same five rows (L1-resident)."""
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
for name, reads in (("random", [rng.integers(0, 4, 3000).astype(np.uint8) for _ in range(1000)]),
                    ("poly-A", [np.zeros(3000, np.uint8) for _ in range(1000)])):
    eng.set_sequences(reads)
    wins = np.array([(p, s, 0, 3000) for p in range(3) for s in range(1000)], dtype=np.int32)
    eng.stage(wins)
    eng.run_staged(3)
    ms, cells = eng.run_staged(20)
    print(f"{name:8s} {ms / 20:7.3f} ms/step  {cells / (ms / 20 * 1e-3) / 1e9:7.1f} GCUPS")
