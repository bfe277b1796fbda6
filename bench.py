#!/usr/bin/env python3
"""bench.py -- GCUPS of the Viterbi scan path on MI355X.

One "step" = one pass of the cost kernels (viterbi_null + viterbi_cost, what every window
pays: c-core/thread.c:114-117) over one batch of windows already resident in HBM.

Workload at N=1 (the one BASELINE.json's metric is quoted on: "Pfam x 10 kb reads"): a
Pfam-shaped pressed-profile set -- P profiles whose lengths follow Pfam-A's (log-normal, median
140, mean ~173, tail to 2500) and whose tables are node runs of the reference's minifam profiles
(SURVEY 8d config 4's fallback: Pfam-A itself is not available offline; deciphon_amd/synth.py) --
against R synthetic 10 kb reads (iid ACGT, every 10th with a planted error-bearing domain),
every window of every (profile, read) pair as c-core/window.c cuts them (window = 50 K nt,
overlap 4K - 1).  P and R are sized so that a step stays below a second.  With N ranks the
PROFILES are sharded, as north_star says: the database has N x P profiles, rank i owns the i-th
contiguous partition (boundaries balanced by core size), reads are replicated, there is no
data-path collective; the hit records are gathered with RCCL after the timed region.  Per-GPU
work is fixed as N grows: weak scaling.

Prints ONE JSON line (rank 0): metric/value/unit per BASELINE.json, plus
  roofline     -- the binding roofline of the cost kernels is VALU issue:
                  wave-level VALU instructions of one step (PMC pass of this same workload and
                  kernel source, profiles/*_traffic.json) / HIP-event time of one step, against
                  the chip's VALU issue peak; HBM figures (measured traffic, algorithmic bytes)
                  ride along as secondary fields
  cpu_baseline -- the reference's own viterbi.c (oracle/_ref, kind "reference") or the oracle
                  restatement (kind "port") on a bounded sample of the same workload, on all
                  host cores (count stated)
  config.secondary -- BASELINE configs[1] (minifam x 1000 synthetic 3 kb reads), same engine
"""
import argparse
import hashlib
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_ISSUE_PEAK = 256 * 4 * 2.4 / 2.0  # G wave64-VALU instr/s: 256 CUs x 4 SIMD-32s, 2.4 GHz, 2 cycles each
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


def pfam_workload(eng, args, rank, world):
    """Loads this rank's partition of the Pfam-shaped database and the (replicated) reads."""
    from deciphon_amd import host, synth

    seeds = synth.load_seeds(SEED_DB)
    Ks_all = synth.pfam_like_lengths(args.profiles * world, SEED)
    bounds = host.partition_bounds(Ks_all, world, balanced=True)
    first, last = int(bounds[rank]), int(bounds[rank + 1])
    proteins = synth.pfam_like_database(seeds, last - first, SEED, first=first, lengths=Ks_all[first:last])
    for p in proteins:
        eng.add_protein(p["core_size"], p["trans"], p["emission"], p["BMk"], p["null_emission"], p["bg_emission"])
    eng.commit()
    # planted domains come from profiles spread over the WHOLE database, so every rank sees the same reads
    stride = max(1, len(Ks_all) // 48)
    cons = [synth.pfam_like_database(seeds, 1, SEED, first=i, lengths=Ks_all[i : i + 1])[0]["consensus"]
            for i in range(0, len(Ks_all), stride)]
    reads = synth.synth_reads(args.reads, args.read_len, cons, SEED)
    eng.set_sequences(reads)
    eng.set_mode(True, False)
    Ks = Ks_all[first:last]
    wins = all_windows(Ks, len(reads), args.read_len)
    desc = (f"Pfam-shaped synthetic profile set ({args.profiles} profiles per GPU, lengths log-normal median 140 "
            f"[{int(Ks.min())}..{int(Ks.max())}], sum K = {int(Ks.sum())}, tables = minifam node runs) x {args.reads} "
            f"synthetic {args.read_len} nt reads, all {len(wins)} windows of c-core/window.c, viterbi_null+viterbi_cost")
    return proteins, reads, wins, desc, (first, last)


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


def host_cores():
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
    return logical, (len(phys) or None)


def cpu_baseline(proteins, reads, read_len, budget_s=18.0):
    """The reference's per-window work (viterbi_null + viterbi_cost, c-core/thread.c:114-117) on the host:
    six profiles at the 10/30/50/70/90/98 % quantiles of the workload's core sizes, each against the
    windows of the first reads, one OpenMP thread per logical core (one struct viterbi each, as
    c-core/scan.c:188-208), repeated until about budget_s seconds of work (calibrated by a first run)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dcp_testlib import oracle, reflib  # TEST INFRASTRUCTURE: the checker, timed here as the baseline

    orc, ref = oracle(), reflib()
    logical, physical = host_cores()
    order = np.argsort([p["core_size"] for p in proteins])
    picks = [proteins[int(order[int(q * (len(order) - 1))])] for q in (0.10, 0.30, 0.50, 0.70, 0.90, 0.98)]
    total_cells = total_secs = 0.0
    nread = min(len(reads), max(2 * logical, 8))
    for p in picks:
        prof = orc.setup_profile(SimpleNamespace(**p))
        chain = window_chain(read_len, prof.K)
        seqs = [np.ascontiguousarray(r[a:b]) for r in reads[:nread] for a, b in chain]
        off = np.zeros(len(seqs) + 1, np.int64)
        np.cumsum([len(s) for s in seqs], out=off[1:])
        nt = np.concatenate(seqs)
        xts = np.stack([orc.xtrans(max(len(s) // 3, 1), True, False) for s in seqs])
        cells = float(prof.K) * float(off[-1])
        if ref is not None:
            secs, _ = ref.bench(prof, xts, nt, off, logical, 1)
            repeat = max(1, int(budget_s / len(picks) / max(secs, 1e-3)))
            secs, _ = ref.bench(prof, xts, nt, off, logical, repeat)
            total_cells += cells * repeat
            total_secs += secs
        else:  # the scalar restatement, one thread: a handful of windows
            t0 = time.time()
            for s, xt in list(zip(seqs, xts))[:2]:
                orc.null(prof, xt, s)
                orc.cost(prof, xt, s)
            total_secs += time.time() - t0
            total_cells += float(prof.K) * float(sum(len(s) for s in seqs[:2]))
    kind = "reference" if ref is not None else "port"
    used = logical if ref is not None else 1
    return {"value": total_cells / total_secs / 1e9, "unit": "GCUPS", "cores": used, "kind": kind,
            "logical_cores": logical, "physical_cores": physical,
            "sample": f"profiles of K = {[p['core_size'] for p in picks]} (10/30/50/70/90/98 % quantiles of the "
                      f"workload) x the windows of the first {nread} reads, viterbi_null+viterbi_cost per window, "
                      f"{total_secs:.1f} s on {used} host thread(s)"}


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
    ap.add_argument("--profile", action="store_true",
                    help="only warmup + timed launches (for rocprofv3: every dispatch is one step's)")
    args = ap.parse_args()
    if args.reads is None:
        args.reads = 500 if args.workload == "pfam" else 1000
    if args.read_len is None:
        args.read_len = 10000 if args.workload == "pfam" else 3000

    import torch

    import deciphon_amd
    from deciphon_amd import dist as ddist

    rank, local_rank, world = ddist.init_process_group("cuda")
    dist = torch.distributed if world > 1 else None
    # one rank per GPU; the modulo only matters when a multi-rank run is rehearsed on fewer GPUs
    # (DECIPHON_DIST_BACKEND=gloo), where ranks share a device
    local_rank %= max(1, torch.cuda.device_count())
    dev = f"cuda:{local_rank}"
    dev_coll = "cpu" if os.environ.get("DECIPHON_DIST_BACKEND") == "gloo" else dev

    eng = deciphon_amd.Engine(local_rank)
    if args.workload == "pfam":
        proteins, reads, wins, desc, (first, last) = pfam_workload(eng, args, rank, world)
        parallelism = (f"profiles sharded over {world} GPU(s) in contiguous partitions balanced by core size "
                       f"(rank 0 owns {first}..{last - 1}), reads replicated, no data-path collective")
        workload_key = f"pfam:{args.profiles}x{args.reads}x{args.read_len}"
    else:
        proteins, reads, wins = minifam_workload(eng, args.reads, args.read_len)
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

    if rank == 0:
        gcups = total_cells * args.steps / t_max / 1e9
        kernel_ms = ms / args.steps
        algo_gbps = (cells * BYTES_PER_CELL) / (kernel_ms * 1e-3) / 1e9
        pmc, pmc_src, stale = measured_counters(workload_key)
        roof = {"bound": "valu_issue", "achieved": None, "peak": VALU_ISSUE_PEAK, "unit": "G wave-instr/s",
                "frac": None, "traffic": None, "kernel_ms_per_step": kernel_ms,
                "kernels": "the cost kernels of one step (dcp_cost_kernel<Q,W>, one launch per kernel class present, "
                           "concurrent on their own streams), timed together with HIP events on the engine's stream",
                "hbm": {"algorithmic_bytes_per_step": cells * BYTES_PER_CELL, "algorithmic_GBps": algo_gbps,
                        "peak_GBps": HBM_PEAK_GBPS,
                        "note": "20 B/cell (SURVEY 8d) are re-read from L2, not HBM: algorithmic_GBps / peak can "
                                "exceed 1 and bounds nothing; traffic = measured HBM bytes per step"},
                "note": "peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md)"}
        if pmc is not None:
            valu = pmc["sq_insts_valu_per_step"]
            ach = valu / (kernel_ms * 1e-3) / 1e9
            roof.update(achieved=ach, frac=ach / VALU_ISSUE_PEAK, traffic=pmc.get("hbm_bytes_per_step"),
                        valu_insts_per_step=valu, counters_source=pmc_src,
                        valu_insts_per_cell=valu * 64.0 / cells,
                        measured_stream_rates={
                            "unit": "G wave-instr/s", "source": "profiles/r02_valu_rates.txt (scripts/valu_rates.hip)",
                            "mix_2add_1min3_8_waves_per_simd": 848.0, "mix_3_waves_per_simd": 729.0,
                            "mix_2_waves_per_simd": 640.0, "v_add_f32_alone": 735.0, "v_min3_f32_alone": 546.0,
                            "frac_of_mix_at_8_waves": ach / 848.0,
                            "note": "no instruction stream measured on this GPU issues at the nominal 2 cycles per wave64; "
                                    "the cost kernels run 2-4 wavefronts per SIMD"})
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
                       "parallelism": parallelism},
            "roofline": roof,
        }
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
        if not args.no_cpu_baseline and not args.profile and world == 1:
            out["cpu_baseline"] = cpu_baseline(proteins, reads, args.read_len)
            out["config"]["vs_cpu_baseline"] = gcups / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
