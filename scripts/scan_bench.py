#!/usr/bin/env python3
"""Whole-scan timing through the reference's outer API (dcp_scan_setup / dcp_scan_run) on a
synthetic database: everything a scan does -- .dcp ingest, read encoding, rounds of chained
windows, cost pass, path pass for the hits, unzip, products.tsv."""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from dcp_synth import random_protein, write_dcp
from deciphon_amd import host
from deciphon_amd.scan import Batch, Scan, Sequence

ap = argparse.ArgumentParser()
ap.add_argument("--profiles", type=int, default=200)
ap.add_argument("--reads", type=int, default=300)
ap.add_argument("--read-len", type=int, default=20000)
args = ap.parse_args()

rng = np.random.default_rng(11)
Ks = np.clip(np.exp(rng.normal(np.log(150), 0.6, size=args.profiles)).astype(int), 10, 1500)
tmp = tempfile.mkdtemp()
dcp = os.path.join(tmp, "synth.dcp")
t0 = time.perf_counter()
write_dcp(dcp, [random_protein(rng, int(K), f"SYN{i:05d}.1") for i, K in enumerate(Ks)])
t1 = time.perf_counter()
batch = Batch()
for i in range(args.reads):
    batch.add(Sequence(i, f"read{i}", "".join(rng.choice(list("ACGT"), size=args.read_len))))
t2 = time.perf_counter()
scan = Scan(dcp, 0, 1, True, False, False)
t3 = time.perf_counter()
scan.run(os.path.join(tmp, "prod"), batch)
t4 = time.perf_counter()
rows = scan.products()
cells = 0
nwin = 0
for K in Ks:
    it = host.WindowIter(args.read_len, int(K))
    while (w := it.next()) is not None:
        cells += int(K) * (w[2] - w[1]) * args.reads
        nwin += args.reads
print(f"db: {args.profiles} profiles (sum K {Ks.sum()}), {os.path.getsize(dcp) / 1e6:.0f} MB written in {t1 - t0:.1f} s")
print(f"scan setup (ingest + H2D) {t3 - t2:.2f} s; scan run {t4 - t3:.2f} s for {nwin} windows "
      f"(no-hit window chain), {len(rows)} product rows -> {cells / (t4 - t3) / 1e9:.1f} GCUPS whole-scan")
