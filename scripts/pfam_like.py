#!/usr/bin/env python3
"""Throughput-mode check on a Pfam-shaped synthetic database (BASELINE configs[3] in small):
P profiles with a Pfam-like length distribution (log-normal, median ~150, tail to 2000) built
directly in DP-cost space, R reads of N nt, one window per pair, cost pass only."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import deciphon_amd
from dcp_testlib import bits, oracle, random_seq, synth_profile

ap = argparse.ArgumentParser()
ap.add_argument("--profiles", type=int, default=400)
ap.add_argument("--reads", type=int, default=1000)
ap.add_argument("--read-len", type=int, default=1000)
ap.add_argument("--check", type=int, default=12)
args = ap.parse_args()

rng = np.random.default_rng(2025)
Ks = np.clip(np.exp(rng.normal(np.log(150), 0.7, size=args.profiles)).astype(int), 8, 2200)
eng = deciphon_amd.Engine(0)
t0 = time.perf_counter()
profs = []
for K in Ks:
    p = synth_profile(rng, int(K))
    profs.append(p)
    eng.add_profile(p.K, p.trans, p.match, p.null, p.bg)
eng.commit()
t1 = time.perf_counter()
reads = [random_seq(rng, args.read_len) for _ in range(args.reads)]
eng.set_sequences(reads)
eng.set_mode(True, False)
wins = [(p, s, 0, args.read_len) for p in range(len(profs)) for s in range(len(reads))]
t2 = time.perf_counter()
eng.stage(wins)
t3 = time.perf_counter()
eng.run_staged(1)
ms, cells = eng.run_staged(3)
print(f"profiles={len(profs)} (K: min {Ks.min()} median {int(np.median(Ks))} max {Ks.max()}, sum {Ks.sum()}) "
      f"reads={len(reads)}x{args.read_len} windows={len(wins)}")
print(f"build+commit {t1 - t0:.1f} s, stage {t3 - t2:.2f} s, cost pass {ms / 3:.1f} ms/step -> "
      f"{cells / (ms / 3 * 1e-3) / 1e9:.1f} GCUPS")
nul, alt = eng.fetch_staged()
orc = oracle()
xt = orc.xtrans(max(args.read_len // 3, 1), True, False)
for i in rng.choice(len(wins), size=args.check, replace=False):
    p, s, _, _ = wins[i]
    assert bits(nul[i]) == bits(orc.null(profs[p], xt, reads[s])), wins[i]
    assert bits(alt[i]) == bits(orc.cost(profs[p], xt, reads[s])), (wins[i], profs[p].K)
print(f"{args.check} random windows bit-exact against the oracle")
