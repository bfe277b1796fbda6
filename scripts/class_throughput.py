#!/usr/bin/env python3
"""Cost-pass throughput per kernel class: one synthetic profile of a given K against enough
reads to fill the GPU (throughput mode)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import deciphon_amd
from dcp_testlib import random_seq, synth_profile

rng = np.random.default_rng(3)
eng = deciphon_amd.Engine(0)
L = 1000
reads = [random_seq(rng, L) for _ in range(16384)]
eng.set_sequences(reads)
eng.set_mode(True, False)
def real_like(K):
    """A profile with the transition/emission structure of real Pfam models: the minifam
    profiles (K = 173, 241, 162) concatenated and cut/tiled to K positions."""
    from dcp_testlib import GOLDEN, oracle
    from oracle.dcp_reader import read_dcp
    from oracle.pyoracle import Profile

    orc = oracle()
    ps = [orc.setup_profile(q) for q in read_dcp(os.path.join(GOLDEN, "minifam.dcp")).proteins]
    trans = np.concatenate([q.trans for q in ps], axis=1)
    match = np.concatenate([q.match for q in ps], axis=1)
    reps = (K + trans.shape[1] - 1) // trans.shape[1]
    trans = np.tile(trans, (1, reps))[:, :K].copy()
    match = np.tile(match, (1, reps))[:, :K].copy()
    inf = np.float32(np.inf)
    for t in (1, 3, 4, 6, 7):
        col = trans[t]
        col[np.isinf(col)] = np.float32(3.0)  # interior joins: a finite transition instead of the model start
        col[0] = inf
    for t in (2, 5):
        col = trans[t]
        col[np.isinf(col)] = np.float32(3.0)
        col[K - 1] = inf
    return Profile(K, np.ascontiguousarray(trans), np.ascontiguousarray(match), ps[0].null, ps[0].bg, f"REAL{K}")


args = sys.argv[1:]
real = "--real" in args
args = [a for a in args if a != "--real"]
for K in [int(a) for a in (args or "3 16 32 64 100 128 173 192 241 256 400 512 900 1024 1800 2048 4096".split())]:
    eng.clear_profiles()
    p = real_like(K) if real else synth_profile(rng, K)
    eng.add_profile(p.K, p.trans, p.match, p.null, p.bg)
    eng.commit()
    # enough windows for about ten generations of wavefronts whatever the kernel (short profiles run several
    # windows per wavefront): windows at a few offsets of every read when the reads alone are too few
    want = int(max(512, min(4.0e9 / (K * L), 2.0e6)))
    per_read = max(1, (want + len(reads) - 1) // len(reads))
    nreads = min(len(reads), want)
    wins = np.array([(0, s, 7 * j, 7 * j + L - 7 * per_read) for j in range(per_read) for s in range(nreads)], np.int32)
    eng.stage(wins)
    eng.run_staged(1)
    ms, cells = eng.run_staged(3)
    print(f"K={K:5d}  windows={len(wins):7d}  {ms / 3:8.2f} ms  {cells / (ms / 3 * 1e-3) / 1e9:7.1f} GCUPS")
