"""Multi-GPU plumbing of the scan path: one process per GPU (torch.distributed; backend
"nccl" is RCCL on ROCm, "gloo" on CPU for tests).  The DP itself needs no collective:
profiles are split into contiguous partitions (c-core/partition_size.c:13-16) or reads are
sharded, and only product rows / hit records are gathered at the end, in rank order -- which
is the reference's partition order (c-core/product.c:63-81)."""
from __future__ import annotations

import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(
        os.environ.get("WORLD_SIZE", "1"))


def partition_bounds(nelems: int, nparts: int):
    """[(first, count)] of every partition: ceil((N - i) / k) elements each, contiguous."""
    out, first = [], 0
    for i in range(nparts):
        size = (max(0, nelems - i) + nparts - 1) // nparts
        out.append((first, size))
        first += size
    return out


def shard(items, rank: int, world: int):
    """Contiguous shard `rank` of `items` with the same partition rule."""
    first, count = partition_bounds(len(items), world)[rank]
    return items[first : first + count]


def init_process_group(device=None):
    """Initialises torch.distributed from the torchrun environment (no-op for one process)."""
    import torch
    import torch.distributed as dist

    rank, local_rank, world = env_rank()
    if world <= 1 or dist.is_initialized():
        return rank, local_rank, world
    # DECIPHON_DIST_BACKEND=gloo: rehearsal of a multi-rank job on a box with fewer GPUs than ranks
    backend = os.environ.get("DECIPHON_DIST_BACKEND")
    if device is not None and str(device).startswith("cuda") and backend != "gloo":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group("gloo")
    return rank, local_rank, world


def gather_rows(rows, device="cpu"):
    """All ranks' product rows, concatenated in rank order, on every rank.  Two collectives:
    all_gather of the byte counts, then all_gather of the padded byte tensors."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(rows)
    world = dist.get_world_size()
    blob = ("\n".join(rows)).encode()
    n = torch.tensor([len(blob), len(rows)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    cap = max(1, max(int(s[0]) for s in sizes))
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    if blob:
        buf[: len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    out = []
    for s, b in zip(sizes, bufs):
        nbytes, nrows = int(s[0]), int(s[1])
        if nrows:
            out += bytes(b[:nbytes].cpu().numpy()).decode().split("\n")
    return out


PRODUCTS_HEADER = "sequence\twindow\twindow_start\twindow_stop\thit\thit_start\thit_stop\tprofile\tabc\tlrt\tevalue\tmatch"


def scan_partitioned(dbfile, sequences, product_dir, multi_hits: bool = True, hmmer3_compat: bool = False,
                     backend: str | None = None, balanced: bool = True):
    """One scan over all GPUs of the job (run under torchrun, one process per GPU): rank i scans the
    i-th contiguous profile partition (c-core/partition_size.c:13-16 -- what thread i of
    c-core/scan.c:188-208 would take) against all the reads; the product rows are gathered in
    rank order, which is the reference's row order (c-core/product.c:63-81), and rank 0 writes
    product_dir/products.tsv.  sequences: [(id, name, text)].  Returns all rows on every rank.
    balanced (default): the partition boundaries balance the sum of core sizes -- DP cells go with K -- instead
    of the number of profiles; contiguous and in order either way, so the row order is the same.

    backend: "nccl" (RCCL; the default when every rank has its own GPU) or "gloo" (also lets several
    ranks share one GPU, as the tests on a one-GPU box do)."""
    import torch
    import torch.distributed as dist

    from .scan import Batch, Scan, Sequence

    rank, local_rank, world = env_rank()
    ndev = max(1, torch.cuda.device_count())
    backend = backend or os.environ.get("DECIPHON_DIST_BACKEND") or ("nccl" if ndev >= world else "gloo")
    device = local_rank % ndev
    if world > 1 and not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")
    batch = Batch()
    for sid, name, text in sequences:
        batch.add(Sequence(sid, name, text))
    part_dir = os.path.join(str(product_dir), f"part{rank}")
    with Scan(dbfile, 0, 1, multi_hits, hmmer3_compat, False, partition=(device, rank, world), balanced=balanced) as scan:
        scan.run(part_dir, batch)
        rows = scan.products()
    rows = gather_rows(rows, f"cuda:{device}" if backend == "nccl" and world > 1 else "cpu")
    if rank == 0:
        with open(os.path.join(str(product_dir), "products.tsv"), "w") as f:
            f.write(PRODUCTS_HEADER + "\n")
            for r in rows:
                f.write(r + "\n")
    if world > 1:
        dist.barrier()
    return rows
