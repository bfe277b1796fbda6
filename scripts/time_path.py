#!/usr/bin/env python3
"""Times the path pass (viterbi_path + trellis_unzip) on the planted-domain windows of the
bench workload: wall time of Engine.path, i.e. kernels + device unzip + D2H of the steps."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import bench
import deciphon_amd
from dcp_testlib import GOLDEN
from oracle.dcp_reader import read_dcp

db = read_dcp(os.path.join(GOLDEN, "minifam.dcp"))
from deciphon_amd import synth
reads = synth.synth_reads(1000, 3000, [p.consensus for p in db.proteins], bench.SEED)
eng = deciphon_amd.Engine(0)
eng.load_dcp(os.path.join(GOLDEN, "minifam.dcp"))
eng.commit()
eng.set_sequences(reads)
eng.set_mode(True, False)
wins = [(p, s, 0, 3000) for p in range(3) for s in range(1000)]
nul, alt = eng.cost(wins)
lrt = -2.0 * ((-nul) - (-alt))
hits = [wins[i] for i in np.nonzero(lrt >= 0)[0]]
for n in (1, len(hits), 10 * len(hits)):
    sel = (hits * 10)[:n]
    eng.path(sel[:1])
    lib, h = eng.lib, eng.h
    arr = (deciphon_amd.Window * n)(*[deciphon_amd.Window(*w) for w in sel])
    t0 = time.perf_counter()
    rc = lib.dcp_hip_path(h, n, arr)
    dt = time.perf_counter() - t0
    assert rc == 0
    cells = sum(eng.core_size(w[0]) * (w[3] - w[2]) for w in sel)
    print(f"path pass: {n:5d} windows  {dt * 1e3:8.2f} ms  {cells / dt / 1e9:7.2f} GCUPS")
