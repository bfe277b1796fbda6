"""GPU: bench.py's contract -- one JSON line with the agreed keys -- for one rank, and the multi-rank
path (barriers, max-over-ranks time, whole-job cells, gathered hit records) rehearsed with two ranks
that share this box's one GPU (DECIPHON_DIST_BACKEND=gloo; on a real node each rank has its own GPU
and the collectives are RCCL)."""
import json
import os
import subprocess
import sys

import pytest

from dcp_testlib import ROOT, bits

pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_one_rank_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1"],
                       capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0, r.stderr
    d = _line(r.stdout)
    assert KEYS <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["unit"] == "GCUPS" and d["dtype"] == "f32"
    assert d["config"]["workload"].startswith("Pfam-shaped") and d["config"]["workload_key"] == "pfam:400x500x10000"
    assert d["ms_per_step"] < 1000.0  # the workload is sized to keep a step below a second
    # the CPU leg: a sweep over thread counts no larger than twice^2 what the process may use; value = its best point
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port")
    if cb["kind"] == "reference":
        pts = {p["threads"]: p["gcups"] for p in cb["sweep"]}
        assert 1 in pts and cb["host_cpus"]["usable"] in pts and len(pts) >= 3
        assert cb["value"] == max(pts.values()) and pts[cb["cores"]] == cb["value"]
        assert cb["value"] >= pts[1] and cb["host_cpus"]["usable"] <= cb["host_cpus"]["affinity"]
    assert abs(d["config"]["vs_cpu_baseline"] - d["value"] / cb["value"]) < 1e-6 * d["config"]["vs_cpu_baseline"]
    r = d["roofline"]
    assert r["bound"] == "valu_issue" and r["peak"] == 1228.8 and r["peak_measured"] == 848.0
    if r["frac"] is not None:  # a PMC summary of this kernel source is committed
        assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert r["frac"] < r["frac_of_measured"] <= 1.0
        assert r["traffic"] > 0 and r["hbm"]["frac"] <= 1.0
    assert d["config"]["hits_gathered"] >= 50  # every 10th read carries a planted domain
    assert d["config"]["secondary"]["value"] > 100  # BASELINE configs[1] rides along
    # a database beyond 4 GB of tables rides along: within a tenth of the headline's rate
    big = d["config"]["large_db"]
    assert "error" not in big and big["pool_bytes"] > 2**32 and big["profiles"] == 5000
    assert 0.85 * d["value"] < big["value"] < 1.1 * d["value"]
    # the database went through the .dcp ingest path
    assert d["config"]["database"]["file_bytes"] > 300e6 and d["config"]["database"]["staging_chunks"] >= 1
    # the whole scan on the same workload (SURVEY 8d's wall definition): slower than the kernels alone, not absurdly so
    e = d["config"]["end_to_end"]
    assert e["cells"] == d["config"]["cells_per_step_per_gpu"] and e["product_rows"] >= 50
    assert 0.3 * d["value"] < e["gcups"] < d["value"]
    ph = e["phases_last_scan"]
    assert ph["windows"] >= d["config"]["windows_per_gpu"] - 1 and ph["path_passes"] >= e["product_rows"]
    parts = sum(ph[k] for k in ("reads_h2d_encode_s", "window_bookkeeping_s", "cost_pass_s", "path_pass_s",
                                "rows_decode_s", "products_tsv_s"))
    assert abs(parts - ph["total_s"]) < 0.02 * ph["total_s"] + 1e-3


def test_gpus_flag_starts_its_own_ranks():
    """`bench.py --gpus 2` with no torchrun around it: the script starts the two ranks itself (a child
    torch.distributed.run) and relays their line.  Two ranks share this box's GPU, hence gloo."""
    env = dict(os.environ, DECIPHON_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--profiles", "60", "--reads", "40"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["collective"]["rccl_ranks"] == 2 and d["config"]["collective"]["world_size"] == 2
    # a launcher that starts another number of ranks than --gpus says is refused, not silently accepted
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "1", "--warmup", "0", "--profiles", "20", "--reads", "10"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "--gpus 1 but the launcher started 2" in r.stderr


def test_bench_step_scores_match_the_oracle(orc, tmp_path):
    """The bench's own step -- the headline database through the .dcp ingest, ALL 472 500 windows staged and launched as
    bench.py launches them (XCD-aware mapping, packs, narrow classes, every stream) -- with a stratified sample of its
    scores compared bit for bit with the oracle: windows of the shortest, the longest and evenly spread profiles
    (every kernel class the workload holds), first and last window of a chain, planted and plain reads."""
    import types

    import numpy as np

    import deciphon_amd
    from deciphon_amd import synth

    sys.path.insert(0, ROOT)
    import bench

    args = types.SimpleNamespace(profiles=400, reads=500, read_len=10000)
    with deciphon_amd.Engine(0) as eng:
        seeds, Ks, reads, wins, _, _, _ = bench.pfam_workload(eng, args, 0, 1, str(tmp_path))
        eng.stage(wins)
        eng.run_staged(1)
        nul, alt = eng.fetch_staged()
    order = np.argsort(Ks, kind="stable")
    picks = sorted({int(order[int(q * (len(order) - 1))]) for q in np.linspace(0.0, 1.0, 26)})
    sample = []
    for n, p in enumerate(picks):
        mine = np.nonzero(wins[:, 0] == p)[0]
        # a planted read (every 10th) and a plain one; first window of one chain, last window of the other
        for read, which in ((10 * (n % 40), 0), (10 * (n % 40) + 3, -1)):
            sample.append(int(mine[wins[mine, 1] == read][which]))
    assert len(sample) >= 50
    classes = set()
    for i in sample:
        p, r, a, b = (int(v) for v in wins[i])
        prot = synth.pfam_like_database(seeds, 1, bench.SEED, first=p, lengths=Ks[p : p + 1])[0]
        prof = orc.setup_profile(types.SimpleNamespace(**prot))
        seq = np.ascontiguousarray(reads[r][a:b])
        xt = orc.xtrans(max(len(seq) // 3, 1), True, False)
        assert bits(nul[i]) == bits(orc.null(prof, xt, seq)), (p, r, a, b)
        assert bits(alt[i]) == bits(orc.cost(prof, xt, seq)), (p, r, a, b)
        classes.add((prof.K + 63) // 64)
    assert len(classes) >= 6  # positions per lane 1 .. 11: packs, single-wave and two-wave classes


def test_two_ranks_line():
    env = dict(os.environ, DECIPHON_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29545", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "1", "--profiles", "120", "--reads", "100"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = _line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert "profiles sharded over 2 GPU(s)" in d["config"]["parallelism"] and "cpu_baseline" not in d
    assert d["config"]["collective"]["rccl_ranks"] == 2
    # the two partitions together hold the 240 profiles: whole-job cells = both ranks' cells
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--profiles",
                          "240", "--reads", "100", "--profile"], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr
    whole = _line(one.stdout)["config"]["cells_per_step_per_gpu"]
    assert abs(d["value"] * d["ms_per_step"] * 1e6 - whole) <= 1e-6 * whole
    assert d["config"]["hits_gathered"] >= 1
