"""GPU: bench.py's contract -- one JSON line with the agreed keys -- for one rank, and the multi-rank
path (barriers, max-over-ranks time, whole-job cells, gathered hit records) rehearsed with two ranks
that share this box's one GPU (DECIPHON_DIST_BACKEND=gloo; on a real node each rank has its own GPU
and the collectives are RCCL)."""
import json
import os
import subprocess
import sys

import pytest

from dcp_testlib import ROOT

pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_one_rank_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    d = _line(r.stdout)
    assert KEYS <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["unit"] == "GCUPS" and d["dtype"] == "f32"
    assert d["config"]["workload"].startswith("Pfam-shaped") and d["config"]["workload_key"] == "pfam:400x500x10000"
    assert d["ms_per_step"] < 1000.0  # the workload is sized to keep a step below a second
    assert d["value"] > 50 * d["cpu_baseline"]["value"]  # north star: >= 50x the reference's CPU path
    assert d["cpu_baseline"]["cores"] == (os.cpu_count() if d["cpu_baseline"]["kind"] == "reference" else 1)
    r = d["roofline"]
    assert r["bound"] == "valu_issue" and r["peak"] == 1228.8
    if r["frac"] is not None:  # a PMC summary of this kernel source is committed
        assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert r["traffic"] > 0 and r["hbm"]["frac"] <= 1.0
    assert d["config"]["hits_gathered"] >= 50  # every 10th read carries a planted domain
    assert d["config"]["secondary"]["value"] > 100  # BASELINE configs[1] rides along


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
    # the two partitions together hold the 240 profiles: whole-job cells = both ranks' cells
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--profiles",
                          "240", "--reads", "100", "--profile"], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr
    whole = _line(one.stdout)["config"]["cells_per_step_per_gpu"]
    assert abs(d["value"] * d["ms_per_step"] * 1e6 - whole) <= 1e-6 * whole
    assert d["config"]["hits_gathered"] >= 1
