#!/usr/bin/env python3
"""bench.py -- GCUPS of the Viterbi scan path on MI355X.

One "step" = one pass of the cost kernels (viterbi_null + viterbi_cost, what every window
pays: c-core/thread.c:114-117) over one batch of windows already resident in HBM.

Workload at N=1 (the one BASELINE.json's metric is quoted on: "Pfam x 10 kb reads"): a
Pfam-shaped pressed-profile set -- P profiles whose lengths follow Pfam-A's (log-normal, median
140, mean ~173, tail to 2500) and whose tables are node runs of the reference's minifam profiles
(SURVEY 8d config 4's fallback: Pfam-A itself is not available offline; deciphon_amd/synth.py) --
written as a `.dcp` file in the current writer's encoding and ingested through dcp_hip_load_dcp,
against R synthetic 10 kb reads (iid ACGT, every 10th with a planted error-bearing domain),
every window of every (profile, read) pair as c-core/window.c cuts them (window = 50 K nt,
overlap 4K - 1).  P and R are sized so that a step stays below a second.  With N ranks the
PROFILES are sharded, as north_star says: the database has N x P profiles, rank i owns the i-th
contiguous partition (boundaries balanced by core size), reads are replicated, there is no
data-path collective; the hit records are gathered with RCCL after the timed region.  Per-GPU
work is fixed as N grows: weak scaling.

`python bench.py --gpus N` with N > 1 and no torchrun environment starts its own N ranks: a CHILD
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py ...` (before this process has
imported torch or touched HIP), whose one JSON line and exit code it relays.

Prints ONE JSON line (rank 0): metric/value/unit per BASELINE.json, plus
  roofline     -- the binding roofline of the cost kernels is VALU issue:
                  wave-level VALU instructions of one step (PMC pass of this same workload and
                  kernel source, profiles/*_traffic.json) / HIP-event time of one step, against
                  the chip's nominal VALU issue peak (`peak`, `frac`) and against the best
                  instruction stream measured on this GPU (`peak_measured`, `frac_of_measured`); HBM
                  figures (measured traffic, algorithmic bytes) ride along as secondary fields
  cpu_baseline -- the reference's own viterbi.c (oracle/_ref, kind "reference") or the oracle
                  restatement (kind "port") on a bounded sample of the same workload: a sweep over
                  thread counts up to what this process may use (affinity, cgroup quota); value = best
  config.end_to_end -- the whole scan on the same workload through the reference's outer API, wall time
                  as SURVEY 8d defines it (first H2D of the reads to the last product row on the host:
                  reads H2D + encode, rounds of chained windows, cost pass, LRT filter, path pass, unzip,
                  decoding, products.tsv), N = 1 only
  config.secondary -- BASELINE configs[1] (minifam x 1000 synthetic 3 kb reads), same engine
  config.large_db -- the cost pass over a database of 5000 profiles (5.3 GB file, 6 GB of tables) x 100 reads
                  (--large-db P; 0 skips it)
"""
import argparse
import hashlib
import json
import math
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_ISSUE_PEAK = 256 * 4 * 2.4 / 2.0  # G wave64-VALU instr/s: 256 CUs x 4 SIMD-32s, 2.4 GHz, 2 cycles each
VALU_MEASURED_PEAK = 848.0  # best sustained stream on this GPU: 2 v_add_f32 per v_min3_f32, 8 waves per SIMD
BYTES_PER_CELL = 20.0  # five fp32 match-emission operands per DP cell (cost pass)
SEED = 20250310
SEED_DB = os.path.join(ROOT, "tests", "golden", "minifam.dcp")  # the reference's committed fixture (data)


def kernel_source_hash():
    """Identifies the kernel source a PMC summary was taken on (instruction counts go stale with it)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "deciphon_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def window_chain(read_len, K):
    """The windows of one (profile, read) pair when no window hits (c-core/window.c:7-37)."""
    from deciphon_amd import host

    it, out = host.WindowIter(read_len, K), []
    while (w := it.next()) is not None:
        out.append((w[1], w[2]))
    return out


def all_windows(Ks, nreads, read_len):
    """int32 [n][4] = (profile, read, start, stop): every window of every pair."""
    chains = {}
    parts = []
    for p, K in enumerate(Ks):
        ch = chains.get(int(K))
        if ch is None:
            ch = chains[int(K)] = np.array(window_chain(read_len, int(K)), np.int32).reshape(-1, 2)
        w = np.empty((nreads, len(ch), 4), np.int32)
        w[:, :, 0] = p
        w[:, :, 1] = np.arange(nreads, dtype=np.int32)[:, None]
        w[:, :, 2:] = ch[None]
        parts.append(w.reshape(-1, 4))
    return np.ascontiguousarray(np.concatenate(parts))


def scratch_dir():
    """Where the synthetic databases are written: DECIPHON_BENCH_TMP, else the system's temporary directory."""
    return tempfile.mkdtemp(prefix="dcp_bench_", dir=os.environ.get("DECIPHON_BENCH_TMP") or None)


def pfam_reads(seeds, Ks_all, nreads, read_len):
    """Replicated on every rank: planted domains come from profiles spread over the WHOLE database."""
    from deciphon_amd import synth

    stride = max(1, len(Ks_all) // 48)
    cons = [synth.pfam_like_database(seeds, 1, SEED, first=i, lengths=Ks_all[i : i + 1])[0]["consensus"]
            for i in range(0, len(Ks_all), stride)]
    return synth.synth_reads(nreads, read_len, cons, SEED)


def pfam_workload(eng, args, rank, world, tmp):
    """Writes this rank's partition of the Pfam-shaped database as a .dcp (streamed: one protein alive at a time),
    ingests it (dcp_hip_load_dcp) and loads the (replicated) reads."""
    from deciphon_amd import host, synth

    seeds = synth.load_seeds(SEED_DB)
    Ks_all = synth.pfam_like_lengths(args.profiles * world, SEED)
    bounds = host.partition_bounds(Ks_all, world, balanced=True)
    first, last = int(bounds[rank]), int(bounds[rank + 1])
    Ks = Ks_all[first:last]
    dcp = os.path.join(tmp, f"pfam_like_{rank}.dcp")
    t0 = time.perf_counter()
    synth.write_dcp(dcp, synth.iter_pfam_like(seeds, last - first, SEED, first=first, lengths=Ks), 0.01, False, False)
    t1 = time.perf_counter()
    eng.load_dcp(dcp)
    eng.commit()
    t2 = time.perf_counter()
    reads = pfam_reads(seeds, Ks_all, args.reads, args.read_len)
    eng.set_sequences(reads)
    eng.set_mode(True, False)
    wins = all_windows(Ks, len(reads), args.read_len)
    desc = (f"Pfam-shaped synthetic profile set ({args.profiles} profiles per GPU, lengths log-normal median 140 "
            f"[{int(Ks.min())}..{int(Ks.max())}], sum K = {int(Ks.sum())}, tables = minifam node runs) x {args.reads} "
            f"synthetic {args.read_len} nt reads, all {len(wins)} windows of c-core/window.c, viterbi_null+viterbi_cost")
    db = {"file_bytes": os.path.getsize(dcp), "write_s": t1 - t0, "ingest_s": t2 - t1, "staging_chunks": eng.load_chunks,
          "path": dcp}
    return seeds, Ks_all, reads, wins, desc, (first, last), db


def minifam_workload(eng, nreads, read_len):
    from deciphon_amd import synth

    seeds = synth.load_seeds(SEED_DB)
    eng.load_dcp(SEED_DB)
    eng.commit()
    reads = synth.synth_reads(nreads, read_len, [s["consensus"] for s in seeds], SEED)
    eng.set_sequences(reads)
    eng.set_mode(True, False)
    wins = np.array([(p, s, 0, read_len) for p in range(len(seeds)) for s in range(nreads)], np.int32)
    return seeds, reads, wins


def host_cpus():
    """What this process may use: logical and physical CPUs of the machine, its affinity mask, and the CPU-time quota
    of its cgroup (v2 cpu.max, v1 cpu.cfs_quota_us) in CPUs -- a container with a quota of 16 CPUs on a 256-thread
    machine runs 256 busy threads at a sixteenth of their speed."""
    logical = os.cpu_count() or 1
    phys = set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = logical
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    usable = affinity if quota is None else max(1, min(affinity, int(math.ceil(quota))))
    return {"logical": logical, "physical": (len(phys) or None), "affinity": affinity, "cgroup_quota_cpus": quota,
            "usable": usable}


def cpu_baseline(seeds, Ks_all, reads, read_len, budget_s=22.0):
    """The reference's per-window work (viterbi_null + viterbi_cost, c-core/thread.c:114-117) on the host:
    six profiles at the 10/30/50/70/90/98 % quantiles of the workload's core sizes, each against windows of the first
    reads, one OpenMP thread per struct viterbi (as c-core/scan.c:188-208), at several thread counts up to (and one
    step beyond) the CPUs this process may use; every point is reported, the best is the value.  Each point takes
    about budget_s / points of wall time (calibrated by a first pass)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from deciphon_amd import synth
    from dcp_testlib import oracle, reflib  # TEST INFRASTRUCTURE: the checker, timed here as the baseline

    orc, ref = oracle(), reflib()
    cpus = host_cpus()
    order = np.argsort(Ks_all, kind="stable")
    picks = [int(order[int(q * (len(order) - 1))]) for q in (0.10, 0.30, 0.50, 0.70, 0.90, 0.98)]
    profs = []
    for i in picks:
        p = synth.pfam_like_database(seeds, 1, SEED, first=i, lengths=Ks_all[i : i + 1])[0]
        profs.append(orc.setup_profile(SimpleNamespace(**p)))
    Kpicks = [p.K for p in profs]

    def sample(prof, nwin):
        chain = window_chain(read_len, prof.K)
        seqs = []
        for r in reads:
            for a, b in chain:
                seqs.append(np.ascontiguousarray(r[a:b]))
            if len(seqs) >= nwin:
                break
        seqs = seqs[:nwin]
        off = np.zeros(len(seqs) + 1, np.int64)
        np.cumsum([len(s) for s in seqs], out=off[1:])
        xts = np.stack([orc.xtrans(max(len(s) // 3, 1), True, False) for s in seqs])
        return np.concatenate(seqs), off, xts, float(prof.K) * float(off[-1])

    if ref is None:  # the scalar restatement, one thread: a handful of windows
        t0, cells = time.time(), 0.0
        for prof in profs:
            nt, off, xts, c = sample(prof, 2)
            for j in range(2):
                s = nt[off[j] : off[j + 1]]
                orc.null(prof, xts[j], s)
                orc.cost(prof, xts[j], s)
            cells += c
        secs = time.time() - t0
        return {"value": cells / secs / 1e9, "unit": "GCUPS", "cores": 1, "kind": "port", "host_cpus": cpus,
                "sample": f"profiles of K = {Kpicks} x 2 windows each, oracle restatement, {secs:.1f} s on 1 thread"}

    u = cpus["usable"]
    counts = sorted({1, max(1, u // 2), u, min(2 * u, cpus["logical"]), min(4 * u, cpus["logical"])})
    per_point = budget_s / len(counts)
    sweep = []
    for T in counts:
        cells = secs = 0.0
        for prof in profs:
            nt, off, xts, c = sample(prof, max(2 * T, 4))  # two windows per thread: the dynamic schedule evens them out
            s1, _ = ref.bench(prof, xts, nt, off, T, 1)
            repeat = max(1, int(per_point / len(profs) / max(s1, 1e-3)) - 1)
            s2, _ = ref.bench(prof, xts, nt, off, T, repeat)
            cells += c * (1 + repeat)
            secs += s1 + s2
        sweep.append({"threads": T, "gcups": cells / secs / 1e9, "seconds": secs})
    best = max(sweep, key=lambda p: p["gcups"])
    return {"value": best["gcups"], "unit": "GCUPS", "cores": best["threads"], "kind": "reference", "sweep": sweep,
            "host_cpus": cpus,
            "sample": f"profiles of K = {Kpicks} (10/30/50/70/90/98 % quantiles of the workload) x two windows per thread "
                      f"of the first reads, viterbi_null+viterbi_cost per window (the reference's AVX2 viterbi.c, OpenMP, "
                      f"passive waits), {sum(p['seconds'] for p in sweep):.1f} s over thread counts "
                      f"{[p['threads'] for p in sweep]}; this process may use {u} CPUs (affinity {cpus['affinity']}, cgroup "
                      f"quota {cpus['cgroup_quota_cpus']}) of {cpus['logical']} logical / {cpus['physical']} physical"}


def end_to_end(dcp, reads, cells, device, runs=3):
    """The whole scan through include/deciphon.h (what python-core's Scan binds): dcp_scan_setup once, dcp_scan_run
    `runs` times (the first warms allocations up and is reported apart)."""
    from deciphon_amd.scan import Batch, Scan, Sequence

    os.environ.setdefault("DECIPHON_HIP_DEVICE", str(device))
    batch = Batch()
    for i, r in enumerate(reads):
        batch.add(Sequence(i, f"read{i}", "".join("ACGT"[v] for v in r)))
    out_dir = os.path.join(os.path.dirname(dcp), "products")
    t0 = time.perf_counter()
    scan = Scan(dcp, 0, 1, True, False, False)
    setup_s = time.perf_counter() - t0
    walls, phases, rows = [], None, 0
    for rep in range(runs):
        t0 = time.perf_counter()
        scan.run(out_dir, batch)
        walls.append(time.perf_counter() - t0)
        phases = scan.last_timing()
        rows = len(scan.products())
    scan.free()
    timed = walls[1:] if len(walls) > 1 else walls
    s = sum(timed) / len(timed)
    return {"gcups": cells / s / 1e9, "s_per_scan": s, "first_scan_s": walls[0], "scans_timed": len(timed),
            "setup_s": setup_s, "product_rows": rows, "phases_last_scan": phases,
            "cells": cells,
            "note": "dcp_scan_run wall time (host clock around the call): reads H2D + encode, rounds of chained windows, "
                    "cost pass + LRT filter, path pass + unzip of the hits, quasi-codon decoding, products.tsv; cells = the "
                    "windows of the no-hit chains (a hit moves the next window, c-core/window.c:21-31), as the cost-pass value "
                    "counts them; setup_s = dcp_scan_setup (open, ingest, H2D), once per database"}


def large_db(args, device, tmp):
    """The cost pass over a database whose tables exceed 4 GB (Pfam-A is ~2e4 profiles, ~20 GB of tables)."""
    import deciphon_amd
    from deciphon_amd import synth

    seeds = synth.load_seeds(SEED_DB)
    Ks = synth.pfam_like_lengths(args.large_db, SEED + 1)
    dcp = os.path.join(tmp, "pfam_like_large.dcp")
    t0 = time.perf_counter()
    synth.write_dcp(dcp, synth.iter_pfam_like(seeds, len(Ks), SEED + 1, lengths=Ks), 0.01, False, False)
    t1 = time.perf_counter()
    with deciphon_amd.Engine(device) as eng:
        eng.load_dcp(dcp)
        eng.commit()
        t2 = time.perf_counter()
        size = os.path.getsize(dcp)
        os.unlink(dcp)
        reads = pfam_reads(seeds, Ks, args.large_db_reads, args.read_len)
        eng.set_sequences(reads)
        eng.set_mode(True, False)
        wins = all_windows(Ks, len(reads), args.read_len)
        eng.stage(wins)
        eng.run_staged(1)
        ms, cells = eng.run_staged(args.large_db_steps)
        pool = eng.pool_bytes
    return {"profiles": int(len(Ks)), "sum_K": int(Ks.sum()), "reads": len(reads), "windows": int(len(wins)),
            "file_bytes": size, "pool_bytes": pool, "write_s": t1 - t0, "ingest_s": t2 - t1,
            "ingest_GBps": size / (t2 - t1) / 1e9, "cells_per_step": cells, "ms_per_step": ms / args.large_db_steps,
            "value": cells * args.large_db_steps / (ms * 1e-3) / 1e9, "unit": "GCUPS", "steps": args.large_db_steps}


def measured_counters(workload_key):
    """PMC summary of this workload on this kernel source (scripts/profile_bench.sh -> profiles/*_traffic.json:
    FETCH_SIZE / WRITE_SIZE / SQ_INSTS_* from separate --pmc passes).  None when no summary matches."""
    import glob

    want = kernel_source_hash()
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        t = json.load(open(f))
        if t.get("workload_key") != workload_key:
            continue
        if t.get("kernel_source_hash") == want:
            return t, os.path.basename(f), None
        stale = stale or os.path.basename(f)
    return None, None, stale


def spawn_ranks(args):
    """--gpus N without a torchrun environment: N ranks as a child process (never an exec, and nothing here has
    touched HIP yet).  Relays the child's one JSON line and exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(r.stdout)
    sys.stdout.flush()
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=("pfam", "minifam"), default="pfam")
    ap.add_argument("--profiles", type=int, default=400, help="profiles per GPU (pfam workload)")
    ap.add_argument("--reads", type=int, default=None)
    ap.add_argument("--read-len", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--large-db", type=int, default=5000, metavar="P",
                    help="also time the cost pass over a database of P profiles (5000 = 6 GB of tables; 0 = skip)")
    ap.add_argument("--large-db-reads", type=int, default=100)
    ap.add_argument("--large-db-steps", type=int, default=2)
    ap.add_argument("--profile", action="store_true",
                    help="only warmup + timed launches (for rocprofv3: every dispatch is one step's)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be at least 1")
    if args.reads is None:
        args.reads = 500 if args.workload == "pfam" else 1000
    if args.read_len is None:
        args.read_len = 10000 if args.workload == "pfam" else 3000

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # the cpu_baseline leg: idle OpenMP threads sleep, not spin

    import torch

    import deciphon_amd
    from deciphon_amd import dist as ddist

    rank, local_rank, world = ddist.init_process_group("cuda")
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        sys.exit(2)
    dist = torch.distributed if world > 1 else None
    gloo = os.environ.get("DECIPHON_DIST_BACKEND") == "gloo"
    ndev = max(1, torch.cuda.device_count())
    if world > ndev and not gloo:
        print(f"bench.py: {world} ranks but {ndev} GPU(s): one rank per GPU (DECIPHON_DIST_BACKEND=gloo rehearses "
              f"more ranks than GPUs)", file=sys.stderr)
        sys.exit(2)
    # one rank per GPU; the modulo only matters when a multi-rank run is rehearsed on fewer GPUs
    # (DECIPHON_DIST_BACKEND=gloo), where ranks share a device
    local_rank %= ndev
    dev = f"cuda:{local_rank}"
    dev_coll = "cpu" if gloo else dev

    tmp = scratch_dir()
    try:
        run(args, torch, deciphon_amd, ddist, dist, rank, local_rank, world, dev, dev_coll, tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run(args, torch, deciphon_amd, ddist, dist, rank, local_rank, world, dev, dev_coll, tmp):
    eng = deciphon_amd.Engine(local_rank)
    db = None
    if args.workload == "pfam":
        seeds, Ks_all, reads, wins, desc, (first, last), db = pfam_workload(eng, args, rank, world, tmp)
        parallelism = (f"profiles sharded over {world} GPU(s) in contiguous partitions balanced by core size "
                       f"(rank 0 owns {first}..{last - 1}), reads replicated, no data-path collective")
        workload_key = f"pfam:{args.profiles}x{args.reads}x{args.read_len}"
    else:
        seeds, reads, wins = minifam_workload(eng, args.reads, args.read_len)
        desc = (f"minifam.dcp (K=173,241,162) x {args.reads} synthetic {args.read_len} nt reads, one window per "
                f"pair, viterbi_null+viterbi_cost")
        parallelism = "replicas: every GPU scores the same windows"
        workload_key = f"minifam:{args.reads}x{args.read_len}"
    eng.stage(wins)  # inputs resident in HBM before the timed region

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    rccl_ranks = None
    if dist is not None:
        # the collective library is up before the timed region: one all_reduce of ones on the collective device
        # (device tensors under nccl = RCCL) counts the ranks it actually connects
        one = torch.ones(1, dtype=torch.float32, device=dev_coll)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        rccl_ranks = int(round(float(one.item())))

    eng.run_staged(args.warmup)
    barrier()
    t0 = time.perf_counter()
    ms, cells = eng.run_staged(args.steps)  # returns when the last launch has finished
    barrier()
    wall = time.perf_counter() - t0

    t = torch.tensor([wall], dtype=torch.float64, device=dev_coll)
    c = torch.tensor([cells], dtype=torch.float64, device=dev_coll)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    t_max, total_cells = float(t.item()), float(c.item())

    host_gcups = None
    all_rows = []
    if not args.profile:
        # outside the timed region: the same step through the host-buffer boundary (reads H2D + encode,
        # window list H2D, kernels, scores D2H) -- the PCIe-inclusive rate DESIGN.md quotes
        t1 = time.perf_counter()
        eng.set_sequences(reads)
        nul, alt = eng.cost(wins)
        host_gcups = cells / (time.perf_counter() - t1) / 1e9
        # the path's only exchange, after the timed region: hit records gathered over RCCL
        lrt = -2.0 * ((-nul) - (-alt))
        rows = [f"{rank}\t{wins[i][0]}\t{wins[i][1]}\t{wins[i][2]}\t{lrt[i]:.1f}" for i in np.nonzero(lrt >= 0)[0]]
        all_rows = ddist.gather_rows(rows, dev_coll)

    if rank != 0:
        return
    gcups = total_cells * args.steps / t_max / 1e9
    kernel_ms = ms / args.steps
    algo_gbps = (cells * BYTES_PER_CELL) / (kernel_ms * 1e-3) / 1e9
    pmc, pmc_src, stale = measured_counters(workload_key)
    roof = {"bound": "valu_issue", "achieved": None, "peak": VALU_ISSUE_PEAK, "unit": "G wave-instr/s",
            "frac": None, "peak_measured": VALU_MEASURED_PEAK, "frac_of_measured": None, "traffic": None,
            "kernel_ms_per_step": kernel_ms,
            "kernels": "all cost kernels of one step, concurrent on their own streams and timed together with HIP events "
                       "on the engine's stream: dcp_cost_kernel<Q,W> (one window per wavefront or workgroup, one launch per "
                       "class present, plus the narrow variants of classes 4..6), dcp_cost_pack_kernel<Q,S> / "
                       "dcp_cost_pack_lds_kernel (several windows of a short profile per wavefront), dcp_cost_fused_kernel "
                       "(small mixed launches)",
            "hbm": {"algorithmic_bytes_per_step": cells * BYTES_PER_CELL, "algorithmic_GBps": algo_gbps,
                    "peak_GBps": HBM_PEAK_GBPS,
                    "note": "20 B/cell (SURVEY 8d) are re-read from L2, not HBM: algorithmic_GBps / peak can "
                            "exceed 1 and bounds nothing; traffic = measured HBM bytes per step"},
            "note": "peak = nominal: 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md); "
                    "peak_measured = the best sustained instruction stream measured on this GPU (2 v_add_f32 per "
                    "v_min3_f32 -- the DP's mix -- at 8 wavefronts per SIMD: profiles/r02_valu_rates.txt, "
                    "scripts/valu_rates.hip); no stream issues at the nominal 2 cycles"}
    if pmc is not None:
        valu = pmc["sq_insts_valu_per_step"]
        ach = valu / (kernel_ms * 1e-3) / 1e9
        roof.update(achieved=ach, frac=ach / VALU_ISSUE_PEAK, frac_of_measured=ach / VALU_MEASURED_PEAK,
                    traffic=pmc.get("hbm_bytes_per_step"),
                    valu_insts_per_step=valu, counters_source=pmc_src,
                    valu_insts_per_cell=valu * 64.0 / cells,
                    measured_stream_rates={
                        "unit": "G wave-instr/s", "source": "profiles/r02_valu_rates.txt (scripts/valu_rates.hip)",
                        "mix_2add_1min3_8_waves_per_simd": 848.0, "mix_3_waves_per_simd": 729.0,
                        "mix_2_waves_per_simd": 640.0, "v_add_f32_alone": 735.0, "v_min3_f32_alone": 546.0,
                        "note": "the cost kernels run 2-4 wavefronts per SIMD"})
        if pmc.get("hbm_bytes_per_step"):
            roof["hbm"]["achieved_GBps"] = pmc["hbm_bytes_per_step"] / (kernel_ms * 1e-3) / 1e9
            roof["hbm"]["frac"] = roof["hbm"]["achieved_GBps"] / HBM_PEAK_GBPS
    else:
        roof["note"] += ("; no PMC summary under profiles/ matches this workload and kernel source"
                         + (f" ({stale} was taken on other kernel source)" if stale else "")
                         + ": run scripts/profile_bench.sh")
    out = {
        "metric": "GCUPS (Viterbi DP cell updates/sec)", "value": gcups, "unit": "GCUPS",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_max / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": desc, "workload_key": workload_key, "kernel_source_hash": kernel_source_hash(),
                   "windows_per_gpu": int(len(wins)), "cells_per_step_per_gpu": cells,
                   "hits_gathered": len(all_rows), "pcie_inclusive_gcups_per_gpu": host_gcups,
                   "parallelism": parallelism,
                   "collective": None if dist is None else {
                       "backend": dist.get_backend(), "library": "gloo (rehearsal)" if dev_coll == "cpu" else "RCCL",
                       "rccl_ranks": rccl_ranks, "world_size": dist.get_world_size()}},
        "roofline": roof,
    }
    if db is not None:
        out["config"]["database"] = {k: v for k, v in db.items() if k != "path"}
        out["config"]["database"]["ingest_GBps"] = db["file_bytes"] / db["ingest_s"] / 1e9
    if world == 1 and not args.profile and not args.no_end_to_end and args.workload == "pfam":
        eng.close()  # the scan owns its engines; the bench's tables go first
        out["config"]["end_to_end"] = end_to_end(db["path"], reads, cells, local_rank)
        out["config"]["end_to_end"]["frac_of_value"] = out["config"]["end_to_end"]["gcups"] / gcups
    if world == 1 and not args.profile and not args.no_secondary and args.workload == "pfam":
        # BASELINE configs[1], same engine class, same entry points
        with deciphon_amd.Engine(local_rank) as e2:
            _, _, w2 = minifam_workload(e2, 1000, 3000)
            e2.stage(w2)
            e2.run_staged(3)
            ms2, cells2 = e2.run_staged(20)
            out["config"]["secondary"] = {
                "workload": "minifam.dcp (K=173,241,162) x 1000 synthetic 3000 nt reads, one window per pair "
                            "(BASELINE configs[1])",
                "value": cells2 * 20 / (ms2 * 1e-3) / 1e9, "unit": "GCUPS", "ms_per_step": ms2 / 20}
    if world == 1 and not args.profile and args.large_db > 0 and args.workload == "pfam":
        eng.close()
        try:  # 5 GB of scratch file and 6 GB of HBM more than the line needs: a box that lacks them still gets its line
            out["config"]["large_db"] = large_db(args, local_rank, tmp)
            out["config"]["large_db"]["frac_of_value"] = out["config"]["large_db"]["value"] / gcups
        except (OSError, MemoryError, RuntimeError) as e:
            out["config"]["large_db"] = {"error": f"{type(e).__name__}: {e}"}
    if not args.no_cpu_baseline and not args.profile and world == 1:
        out["cpu_baseline"] = cpu_baseline(seeds, Ks_all, reads, args.read_len) if args.workload == "pfam" else None
        if out["cpu_baseline"]:
            out["config"]["vs_cpu_baseline"] = gcups / out["cpu_baseline"]["value"]
        else:
            del out["cpu_baseline"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
