#!/usr/bin/env python3
"""bench.py -- GCUPS of the Viterbi scan path on MI355X.

One "step" = one pass of the cost kernels (viterbi_null + viterbi_cost, what every
window pays: c-core/thread.c:114-117) over one batch of synthetic windows that is
already resident in HBM.  Workload at N=1: BASELINE.json configs[1] -- the three
minifam profiles (K = 173/241/162) x 1000 synthetic 3 kb reads (iid ACGT, seed
20250310+i, 10 % with a planted error-bearing domain), one window per pair.  With N
ranks the reads are sharded over ranks (weak scaling: every rank scores 1000 reads of
its own against the profiles), no data-path collective; hit records are gathered
with RCCL after the timed region.

Prints ONE JSON line (rank 0): metric/value/unit per BASELINE.json, plus
  roofline     -- algorithmic bytes (20 B per DP cell) / measured kernel time
  cpu_baseline -- the reference's own viterbi.c (oracle/_ref, kind "reference") or the
                  oracle restatement (kind "port") on a bounded sample, host cores stated
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_ISSUE_PEAK = 256 * 4 * 2.4 / 2.0  # G wave64-VALU instr/s: 256 CUs x 4 SIMD-32s, 2.4 GHz, 2 cycles each
BYTES_PER_CELL = 20.0  # five fp32 match-emission operands per DP cell (cost pass)
SEED = 20250310


def synth_reads(nreads, length, consensus, rank=0):
    """iid-uniform ACGT reads; every 10th carries one planted domain: a profile consensus
    back-translated with a fixed codon per amino acid, then 10 % substitutions, 3 %
    insertions, 3 % deletions."""
    codon = {a: c for a, c in zip("ACDEFGHIKLMNPQRSTVWY",
                                  ["GCT", "TGT", "GAT", "GAA", "TTT", "GGT", "CAT", "ATT", "AAA", "CTG", "ATG",
                                   "AAT", "CCT", "CAA", "CGT", "TCT", "ACT", "GTT", "TGG", "TAT"])}
    lut = {"A": 0, "C": 1, "G": 2, "T": 3}
    reads = []
    for i in range(nreads):
        rng = np.random.default_rng(SEED + rank * nreads + i)
        r = rng.integers(0, 4, size=length).astype(np.uint8)
        if i % 10 == 0 and consensus:
            cons = consensus[i // 10 % len(consensus)]
            dom = np.array([lut[ch] for a in cons for ch in codon.get(a.upper(), "GCT")], dtype=np.uint8)
            out = []
            for b in dom:
                u = rng.random()
                if u < 0.03:
                    continue
                if u < 0.06:
                    out.append(rng.integers(0, 4))
                out.append(rng.integers(0, 4) if rng.random() < 0.10 else b)
            dom = np.array(out, dtype=np.uint8)[: max(1, length - 10)]
            at = int(rng.integers(0, length - len(dom) + 1))
            r[at : at + len(dom)] = dom
        reads.append(r)
    return reads


def cpu_baseline(db, reads, cores, budget_s=15.0):
    """The reference's per-window work (viterbi_null + viterbi_cost) on the host, on a bounded
    sample of the same workload: profile 0 against the first reads, repeated until about
    budget_s seconds of work (calibrated by a short first run)."""
    from dcp_testlib import oracle, reflib

    orc = oracle()
    ref = reflib()
    prof = orc.setup_profile(db.proteins[0])
    if ref is not None:
        n = min(len(reads), 8 * cores)
        sample = reads[:n]
        xts = np.stack([orc.xtrans(max(len(r) // 3, 1), True, False) for r in sample])
        off = np.zeros(n + 1, np.int64)
        np.cumsum([len(r) for r in sample], out=off[1:])
        nt = np.concatenate(sample)
        secs, _ = ref.bench(prof, xts, nt, off, cores, 1)  # calibration
        repeat = max(1, int(budget_s / max(secs, 1e-3)))
        secs, _ = ref.bench(prof, xts, nt, off, cores, repeat)
        kind, used = "reference", cores
        # the same engine on ONE host thread (SURVEY 8d asks for both), a few seconds of it
        n1 = min(n, 8)
        s1, _ = ref.bench(prof, xts[:n1], nt[: off[n1]], off[: n1 + 1], 1, 1)
        r1 = max(1, int(3.0 / max(s1, 1e-3)))
        s1, _ = ref.bench(prof, xts[:n1], nt[: off[n1]], off[: n1 + 1], 1, r1)
        one_thread = float(prof.K) * float(off[n1]) * r1 / s1 / 1e9
    else:
        n, repeat = min(len(reads), 4), 1
        sample = reads[:n]
        t0 = time.time()
        for r in sample:
            xt = orc.xtrans(max(len(r) // 3, 1), True, False)
            orc.null(prof, xt, r)
            orc.cost(prof, xt, r)
        secs = time.time() - t0
        kind, used = "port", 1
        one_thread = None
    cells = float(prof.K) * float(sum(len(r) for r in sample)) * repeat
    return {"value": cells / secs / 1e9, "unit": "GCUPS", "cores": used, "kind": kind,
            "value_one_thread": one_thread,
            "sample": f"profile 0 (K={prof.K}) x the first {n} reads x {repeat} repeats, "
                      f"viterbi_null+viterbi_cost per window, {secs:.1f} s on {used} host thread(s)"}


def measured_traffic():
    """HBM bytes per step and VALU instructions per step from the newest committed PMC summary
    (scripts/profile_bench.sh writes profiles/*_traffic.json: FETCH_SIZE / WRITE_SIZE / SQ_INSTS_*
    collected in separate --pmc passes)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None, None, None
    t = json.load(open(files[-1]))
    return t.get("hbm_bytes_per_step"), os.path.basename(files[-1]), t.get("sq_insts_valu_per_step")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=1000)
    ap.add_argument("--read-len", type=int, default=3000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    import deciphon_amd
    from deciphon_amd import dist as ddist
    from dcp_testlib import GOLDEN
    from oracle.dcp_reader import read_dcp

    rank, local_rank, world = ddist.init_process_group("cuda")
    dist = torch.distributed if world > 1 else None
    # one rank per GPU; the modulo only matters when a multi-rank run is rehearsed on fewer GPUs
    # (DECIPHON_DIST_BACKEND=gloo), where ranks share a device
    local_rank %= max(1, torch.cuda.device_count())
    dev = f"cuda:{local_rank}"
    if os.environ.get("DECIPHON_DIST_BACKEND") == "gloo":
        dev_coll = "cpu"
    else:
        dev_coll = dev

    dcp = os.path.join(GOLDEN, "minifam.dcp")
    db = read_dcp(dcp)
    consensus = [p.consensus for p in db.proteins]
    reads = synth_reads(args.reads, args.read_len, consensus, rank)  # every rank scores its own reads

    eng = deciphon_amd.Engine(local_rank)
    eng.load_dcp(dcp)
    eng.commit()
    eng.set_sequences(reads)
    eng.set_mode(True, False)
    nprof = eng.num_profiles
    wins = np.array([(p, s, 0, len(reads[s])) for p in range(nprof) for s in range(len(reads))], dtype=np.int32)
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

    # outside the timed region: the same step through the host-buffer boundary (reads H2D + encode,
    # window list H2D, kernels, scores D2H) -- the PCIe-inclusive rate DESIGN.md quotes
    t1 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        eng.set_sequences(reads)
        eng.cost(wins)
    host_gcups = cells * reps / (time.perf_counter() - t1) / 1e9
    eng.stage(wins)
    eng.run_staged(1)

    # the path's only exchange, after the timed region: hit records gathered over RCCL
    nul, alt = eng.fetch_staged()
    lrt = -2.0 * ((-nul) - (-alt))
    rows = [f"{rank}\t{wins[i][0]}\t{wins[i][1]}\t{lrt[i]:.1f}" for i in np.nonzero(lrt >= 0)[0]]
    all_rows = ddist.gather_rows(rows, dev_coll)

    if rank == 0:
        gcups = total_cells * args.steps / t_max / 1e9
        kernel_ms = ms / args.steps
        per_gpu_gbps = (cells * BYTES_PER_CELL) / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src, valu_insts = measured_traffic()
        # what actually binds: a SIMD-32 takes 2 cycles per wave64 VALU
        # instruction (MI355X_MICROARCH.md) -> 1024 SIMDs x 2.4 GHz / 2 = 1228.8 G instr/s; counted instructions (PMC pass of the
        # same workload) / measured kernel time of this run
        issue = None
        if valu_insts and args.reads == 1000 and args.read_len == 3000:
            ach = valu_insts / (kernel_ms * 1e-3) / 1e9
            issue = {"bound": "valu_issue", "achieved": ach, "peak": VALU_ISSUE_PEAK, "unit": "G wave-instr/s",
                     "frac": ach / VALU_ISSUE_PEAK, "valu_insts_per_step": valu_insts}
        out = {
            "metric": "GCUPS (Viterbi DP cell updates/sec)", "value": gcups, "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"minifam.dcp (K=173,241,162) x {args.reads} synthetic {args.read_len} nt reads "
                                   f"per GPU, one window per pair, viterbi_null+viterbi_cost",
                       "profiles": nprof, "reads_per_gpu": args.reads, "read_len": args.read_len,
                       "hits_gathered": len(all_rows),
                       "pcie_inclusive_gcups_per_gpu": host_gcups,
                       "parallelism": f"reads sharded over {world} GPU(s), no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": per_gpu_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": per_gpu_gbps / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel_ms_per_step": kernel_ms, "traffic_source": traffic_src, "issue": issue,
                         "note": "achieved = 20 B/cell (SURVEY 8d) x cells of one step / HIP-event time of one "
                                 "step on the engine's stream; operands are re-read from L2, so frac can exceed "
                                 "the HBM share (traffic = measured HBM bytes per step); the binding limit is "
                                 "instruction issue, see DESIGN.md section 5"},
        }
        if not args.no_cpu_baseline and world == 1:  # a reported baseline, timed once (N = 1) on the host cores
            cores = min(os.cpu_count() or 1, 16)
            out["cpu_baseline"] = cpu_baseline(db, reads, cores)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
