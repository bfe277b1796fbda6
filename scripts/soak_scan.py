#!/usr/bin/env python3
"""Randomised soak of the outer API (dcp_scan_setup / dcp_scan_run) against the oracle-driven
restatement of thread_run: random small databases, reads of random length (several chained windows
per pair for the short profiles), both window modes, sometimes scanned as partitions.
scripts/soak_scan.py [seconds] [seed]; writes progress lines."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from dcp_synth import random_protein, write_dcp
from dcp_testlib import oracle
from deciphon_amd.scan import Batch, Scan, Sequence
from oracle.dcp_reader import read_dcp
from test_gpu_scan import oracle_scan

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
orc = oracle()
tmp = tempfile.mkdtemp()
t0 = time.time()
it = nrows = 0
from dcp_testlib import GOLDEN, read_fasta
CONS = [t for _, t in read_fasta(os.path.join(GOLDEN, "consensus.fna"))]
while time.time() - t0 < budget:
    if it % 2 == 1:
        # the reference's own database with mutated copies of its consensus reads planted at random
        # places: many hits, chained windows, hits that move the next window's start
        dcp = os.path.join(GOLDEN, "minifam.dcp")
        reads = []
        for sid in range(int(rng.integers(1, 4))):
            n = int(rng.choice([300, 900, 2500, 9000, 14000]))
            text = list(rng.choice(list("ACGT"), size=n))
            for _ in range(int(rng.integers(1, 6))):
                rate = float(rng.choice([0.0, 0.05, 0.15]))
                dom = [ch if rng.random() > rate else "ACGT"[rng.integers(0, 4)] for ch in CONS[int(rng.integers(0, 3))]]
                if rng.random() < 0.3:
                    dom = dom[: int(rng.integers(30, len(dom)))]
                if len(dom) < n:
                    at = int(rng.integers(0, n - len(dom) + 1))
                    text[at : at + len(dom)] = dom
            reads.append((2000 + sid, "".join(text)))
        mh, h3 = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        want = oracle_scan(orc, read_dcp(dcp), reads, mh, h3)
        nparts = int(rng.choice([1, 2, 3]))
        rows = []
        for part in range(nparts):
            batch = Batch()
            for sid, text in reads:
                batch.add(Sequence(sid, f"seq{sid}", text))
            kw = {} if nparts == 1 else {"partition": (0, part, nparts)}
            with Scan(dcp, 0, 1, mh, h3, False, **kw) as scan:
                scan.run(os.path.join(tmp, f"prodm{it}_{part}"), batch)
                rows += scan.products()
        if rows != want:
            print(f"MISMATCH (minifam) seed={seed} iteration={it} reads={[len(t) for _, t in reads]} mh={mh} h3={h3} "
                  f"nparts={nparts}: {len(rows)} rows against {len(want)}", flush=True)
            sys.exit(1)
        it += 1
        nrows += len(rows)
        if it % 10 == 0:
            print(f"  {it} scans, {nrows} product rows, {time.time() - t0:.0f} s", flush=True)
        continue
    nprof = int(rng.integers(1, 6))
    Ks = [int(rng.choice([2, 3, 5, 9, 17, 40, 70, 130, 200, 300])) for _ in range(nprof)]
    dcp = os.path.join(tmp, f"db{it}.dcp")
    prots = [random_protein(rng, K, f"SOAK{it}_{i}.1") for i, K in enumerate(Ks)]
    for p in prots:  # peaked codon preferences, so that a planted domain scores well above the null model
        w = np.exp(8.0 * rng.random((p.core_size + 1, 64)))
        p.emission[:, 20:84] = (np.log(w / w.sum(axis=1, keepdims=True)) + np.log(0.2)).astype(np.float32)
        p.emission[p.core_size] = p.emission[p.core_size - 1]
    write_dcp(dcp, prots)
    # a profile's favourite codons in a row: planted (with 5 % substitutions) so that windows do hit
    def favourite(p):
        codes = np.argmax(p.emission[: p.core_size, 20:84], axis=1)
        return "".join("ACGT"[c // 16] + "ACGT"[c // 4 % 4] + "ACGT"[c % 4] for c in codes)
    reads = []
    for sid in range(int(rng.integers(1, 5))):
        n = int(rng.choice([1, 2, 7, 60, 400, 1500, 4000]))
        text = list(rng.choice(list("ACGT"), size=n))
        for _ in range(int(rng.integers(0, 4))):
            dom = [ch if rng.random() > 0.05 else "ACGT"[rng.integers(0, 4)] for ch in favourite(prots[int(rng.integers(0, nprof))])]
            if len(dom) < n:
                at = int(rng.integers(0, n - len(dom) + 1))
                text[at : at + len(dom)] = dom
        reads.append((1000 + sid, "".join(text)))
    mh, h3 = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    want = oracle_scan(orc, read_dcp(dcp), reads, mh, h3)
    nparts = int(rng.choice([1, 1, 2, 3]))
    rows = []
    for part in range(nparts):
        batch = Batch()
        for sid, text in reads:
            batch.add(Sequence(sid, f"seq{sid}", text))
        kw = {} if nparts == 1 else {"partition": (0, part, nparts)}
        out = os.path.join(tmp, f"prod{it}_{part}")
        with Scan(dcp, 0, 1, mh, h3, False, **kw) as scan:
            scan.run(out, batch)
            rows += scan.products()
    if rows != want:
        print(f"MISMATCH seed={seed} iteration={it} Ks={Ks} reads={[len(t) for _, t in reads]} mh={mh} h3={h3} "
              f"nparts={nparts}: {len(rows)} rows against {len(want)}", flush=True)
        for a, b in zip(rows, want):
            if a != b:
                print("  got ", a[:160])
                print("  want", b[:160])
                break
        sys.exit(1)
    it += 1
    nrows += len(rows)
    os.remove(dcp)
    if it % 10 == 0:
        print(f"  {it} scans, {nrows} product rows, {time.time() - t0:.0f} s", flush=True)
print(f"scan soak ok: seed {seed}, {it} scans, {nrows} product rows, {time.time() - t0:.0f} s")
