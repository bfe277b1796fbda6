#!/usr/bin/env python3
"""A scan over all the GPUs of a node: torchrun --nproc-per-node N scripts/scan_multi_gpu.py db.dcp reads.fna outdir
(one process per GPU, contiguous profile partitions, rows gathered in partition order on rank 0)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deciphon_amd import dist

dbfile, fasta, outdir = sys.argv[1:4]
seqs, name, chunks = [], None, []
for line in open(fasta):
    line = line.strip()
    if line.startswith(">"):
        if name is not None:
            seqs.append((len(seqs), name, "".join(chunks)))
        name, chunks = line[1:].split()[0], []
    elif line:
        chunks.append(line)
if name is not None:
    seqs.append((len(seqs), name, "".join(chunks)))
os.makedirs(outdir, exist_ok=True)
rows = dist.scan_partitioned(dbfile, seqs, outdir)
rank, _, world = dist.env_rank()
if rank == 0:
    print(f"{len(rows)} product rows from {world} partition(s) -> {os.path.join(outdir, 'products.tsv')}")
