#!/usr/bin/env python3
"""Randomised soak of the GPU path against the oracle: random core sizes over every kernel class
(1..16383), continuous / quantised tables, delete runs that cross wavefronts and strips, short
windows inside longer reads; scores bit for bit, paths step for step.  scripts/soak.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import deciphon_amd
from dcp_testlib import bits, oracle, random_seq, synth_profile

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
orc = oracle()
eng = deciphon_amd.Engine(0)
edges = [1, 2, 3, 12, 13, 28, 29, 60, 61, 63, 64, 65, 93, 94, 124, 125, 128, 129, 192, 193, 256, 257, 320, 321, 384, 385, 448,
         449, 512, 513, 640, 641, 768, 769, 1024, 1025, 1536, 1537, 2048, 2049, 4096, 4097, 6144, 6145, 8192]
t0 = time.time()
rounds = windows = redone = 0
while time.time() - t0 < budget:
    nprof = int(rng.integers(3, 9))
    profs = []
    for _ in range(nprof):
        u = rng.random()
        K = int(rng.choice(edges)) if u < 0.4 else int(np.exp(rng.uniform(0, np.log(5000)))) if u < 0.9 else int(rng.integers(4097, 16384))
        quant = [None, None, 0.5, 2.0, 8.0][int(rng.integers(0, 5))]
        p = synth_profile(rng, K, quant, float(rng.choice([0, 0.02, 0.2])))
        if K > 8 and rng.random() < 0.4:  # nearly free delete runs
            p.trans[7, 1:] = np.float32(rng.choice([0.0, 0.01, 0.25]))
            p.trans[3, 1:] = np.float32(rng.choice([0.0, 0.02, 0.5]))
            p.match[:, int(rng.integers(0, K)):] += np.float32(rng.choice([5.0, 30.0]))
        profs.append((p, quant))
    reads = [random_seq(rng, int(rng.integers(1, 90))) for _ in range(int(rng.integers(2, 6)))]
    mh, h3 = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    eng.clear_profiles()
    for p, _ in profs:
        eng.add_profile(p.K, p.trans, p.match, p.null, p.bg)
    eng.commit()
    eng.set_sequences(reads)
    eng.set_mode(mh, h3)
    wins = []
    for pi in range(nprof):
        for si, r in enumerate(reads):
            a = int(rng.integers(0, len(r)))
            b = int(rng.integers(a + 1, len(r) + 1))
            wins.append((pi, si, a, b) if rng.random() < 0.5 else (pi, si, 0, len(r)))
    nul, alt = eng.cost(wins)
    want_trellis = rng.random() < 0.3
    try:
        paths = eng.path(wins, trellis=want_trellis)
    except deciphon_amd.HipError as e:
        print(f"ERROR seed={seed} round={rounds} mh={mh} h3={h3}: {e}", flush=True)
        for w in wins:
            try:
                eng.path([w], trellis=False)
            except deciphon_amd.HipError as e2:
                p, quant = profs[w[0]]
                dd = p.trans[7, 1:3].tolist() if p.K > 2 else None
                print(f"  window {w} K={p.K} quant={quant} DD={dd} L={w[3] - w[2]}: {e2}", flush=True)
        sys.exit(1)
    redone += eng.path_redone
    for i, (pi, si, a, b) in enumerate(wins):
        p, quant = profs[pi]
        seq = np.ascontiguousarray(reads[si][a:b])
        xt = orc.xtrans(max(len(seq) // 3, 1), mh, h3)
        ok = bits(nul[i]) == bits(orc.null(p, xt, seq)) and bits(alt[i]) == bits(orc.cost(p, xt, seq))
        score, xo, no = orc.path(p, xt, seq)
        ok = ok and bits(paths[i]["score"]) == bits(score)
        if want_trellis:  # the packed trellis: the literal kernel, or the row replay beyond 4096
            ok = ok and np.array_equal(paths[i]["xnodes"], xo) and np.array_equal(paths[i]["nodes"], no)
        if np.isfinite(score):
            try:
                ids, sizes = orc.unzip(p.K, len(seq), xo, no)
            except RuntimeError as e:
                print(f"ORACLE UNZIP FAILED seed={seed} round={rounds} window={wins[i]} K={p.K} quant={quant} mh={mh} h3={h3} "
                      f"score={score} gpu_steps={len(paths[i]['state_ids'])} gpu_score={paths[i]['score']}: {e}", flush=True)
                print("  xnodes", xo.tolist()[:12], "nodes", no.tolist()[:24], flush=True)
                sys.exit(1)
            ok = ok and np.array_equal(paths[i]["state_ids"], ids) and np.array_equal(paths[i]["seqsizes"], sizes)
        else:  # no finite path: nothing to walk (the reference stops at the non-finite lrt)
            ok = ok and len(paths[i]["state_ids"]) == 0
        if not ok:
            print(f"MISMATCH seed={seed} round={rounds} window={wins[i]} K={p.K} quant={quant} mh={mh} h3={h3}", flush=True)
            sys.exit(1)
    rounds += 1
    windows += len(wins)
    if rounds % 10 == 0:
        print(f"  {rounds} rounds, {windows} windows, {redone} redone literally, {time.time() - t0:.0f} s", flush=True)
print(f"soak ok: seed {seed}, {rounds} rounds, {windows} windows ({redone} through the literal pass), {time.time() - t0:.0f} s")
