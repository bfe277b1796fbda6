"""CPU, world_size 2 over gloo: the N > 1 plumbing of the scan path -- contiguous partitions,
read sharding and the rank-ordered gather of product rows (the path's only exchange)."""
import os
import socket
import subprocess
import sys
import textwrap

from deciphon_amd import dist as ddist
from deciphon_amd import host
from dcp_testlib import ROOT


def test_partition_bounds_follow_partition_size():
    for n, k in ((3, 2), (20000, 8), (5, 9), (7, 7)):
        b = ddist.partition_bounds(n, k)
        assert [c for _, c in b] == [host.partition_size(n, k, i) for i in range(k)]
        assert b[0][0] == 0 and all(b[i][0] + b[i][1] == b[i + 1][0] for i in range(k - 1))
        assert sum(c for _, c in b) == n
    items = list(range(10))
    assert sum((ddist.shard(items, r, 3) for r in range(3)), []) == items


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    from deciphon_amd import dist as ddist
    rank, local_rank, world = ddist.init_process_group("cpu")
    assert world == 2 and dist.get_backend() == "gloo"
    profiles = ["PF%05d" % i for i in range(5)]
    mine = ddist.shard(profiles, rank, world)            # contiguous profile partition of this rank
    rows = ["%d\\t%s\\trow of rank %d" % (i, p, rank) for i, p in enumerate(mine)]
    if rank == 1:
        rows = []                                         # a rank without hits
    allrows = ddist.gather_rows(rows, "cpu")
    want = ["%d\\t%s\\trow of rank 0" % (i, p) for i, p in enumerate(ddist.shard(profiles, 0, world))]
    assert allrows == want, allrows
    rows2 = ["r%d-%d" % (rank, i) for i in range(rank + 1)]
    assert ddist.gather_rows(rows2, "cpu") == ["r0-0", "r1-0", "r1-1"]
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join({outdir!r}, "rank%d.ok" % rank), "w").write("ok")
""")


def test_two_ranks_gather_rows_in_rank_order(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, outdir=str(tmp_path)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()
