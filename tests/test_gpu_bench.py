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
    assert d["value"] > 50 * d["cpu_baseline"]["value"]  # north star: >= 50x the reference's CPU path
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / 8000.0) < 1e-9
    assert d["config"]["hits_gathered"] == 100  # every 10th read carries a planted domain


def test_two_ranks_line():
    env = dict(os.environ, DECIPHON_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29545", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "5", "--warmup", "1"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = _line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["hits_gathered"] == 200 and "cpu_baseline" not in d
