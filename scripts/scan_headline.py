#!/usr/bin/env python3
"""Whole-scan timing on bench.py's headline workload through the reference's outer API: the Pfam-shaped
database written as a .dcp, dcp_scan_setup (ingest + H2D), dcp_scan_run (read encoding, rounds of chained windows,
cost pass, LRT filter, path pass of the hits, unzip, decoding, products.tsv).  DECIPHON_HIP_TIMING=1 prints the
phases of the run."""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from deciphon_amd import synth
from deciphon_amd.scan import Batch, Scan, Sequence

ap = argparse.ArgumentParser()
ap.add_argument("--profiles", type=int, default=400)
ap.add_argument("--reads", type=int, default=500)
ap.add_argument("--read-len", type=int, default=10000)
ap.add_argument("--repeat", type=int, default=2)
ap.add_argument("--no-callback", action="store_true", help="no progress callback (a C caller's NULL)")
args = ap.parse_args()

seeds = synth.load_seeds(bench.SEED_DB)
Ks = synth.pfam_like_lengths(args.profiles, bench.SEED)
prots = synth.pfam_like_database(seeds, args.profiles, bench.SEED, lengths=Ks)
tmp = tempfile.mkdtemp()
dcp = os.path.join(tmp, "pfam_like.dcp")
t0 = time.perf_counter()
synth.write_dcp(dcp, prots, 0.01, False, False)
t1 = time.perf_counter()
stride = max(1, len(Ks) // 48)
cons = [prots[i]["consensus"] for i in range(0, len(Ks), stride)]
reads = synth.synth_reads(args.reads, args.read_len, cons, bench.SEED)
batch = Batch()
for i, r in enumerate(reads):
    batch.add(Sequence(i, f"read{i}", "".join("ACGT"[v] for v in r)))
wins = bench.all_windows(Ks, len(reads), args.read_len)
cells = float((Ks[wins[:, 0]].astype(np.float64) * (wins[:, 3] - wins[:, 2])).sum())
t2 = time.perf_counter()
scan = Scan(dcp, 0, 1, True, False, False, progress_callback=not args.no_callback)
t3 = time.perf_counter()
print(f"db: {args.profiles} profiles (sum K {int(Ks.sum())}), {os.path.getsize(dcp) / 1e6:.0f} MB written in {t1 - t0:.1f} s; "
      f"setup (ingest + H2D) {t3 - t2:.2f} s")
for rep in range(args.repeat):
    ta = time.perf_counter()
    scan.run(os.path.join(tmp, f"prod{rep}"), batch)
    tb = time.perf_counter()
    rows = scan.products()
    print(f"scan run {rep}: {tb - ta:.3f} s, {len(wins)} windows of the no-hit chain ({cells:.3e} cells), "
          f"{len(rows)} product rows -> {cells / (tb - ta) / 1e9:.1f} GCUPS whole-scan")
