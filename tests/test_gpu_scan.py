"""GPU: the reference's outer API (dcp_scan_* / dcp_batch_*, include/deciphon.h) through the
Python mirror of python-core's Scan/Batch classes -- the shape of python-core/tests/test_scan.py
and c-core/test_scan.c, test_window.c.  Expected rows come from the reference's committed
products.tsv (every column but the HMMER e-value, which is out of scope) and from an oracle-driven
restatement of thread_run."""
import os

import numpy as np
import pytest

import dcp_testlib
from dcp_testlib import GOLDEN, read_fasta
from oracle.dcp_reader import read_dcp

pytestmark = pytest.mark.gpu

DCP = os.path.join(GOLDEN, "minifam.dcp")


def oracle_scan(orc, db, reads, multi_hits, hmmer3_compat):
    return dcp_testlib.oracle_scan(orc, db.proteins, reads, multi_hits, hmmer3_compat)


def run_scan(tmp_path, reads, multi_hits=True, hmmer3_compat=False, dbfile=DCP, **kw):
    from deciphon_amd.scan import Batch, Scan, Sequence

    batch = Batch()
    for sid, text in reads:
        batch.add(Sequence(sid, f"seq{sid}", text))
    with Scan(dbfile, 0, 1, multi_hits, hmmer3_compat, False, **kw) as scan:
        scan.run(tmp_path, batch)
        assert scan.progress() == 100
        rows = scan.products()
    lines = open(os.path.join(tmp_path, "products.tsv")).read().splitlines()
    assert lines[0].split("\t") == ["sequence", "window", "window_start", "window_stop", "hit", "hit_start",
                                    "hit_stop", "profile", "abc", "lrt", "evalue", "match"]
    assert lines[1:] == rows
    return rows


def test_consensus_scan_matches_reference_products(tmp_path, orc):
    """control/tests/files/consensus.fna x minifam.dcp == control/tests/files/snap.dcs."""
    reads = [(i, s) for i, (_, s) in enumerate(read_fasta(os.path.join(GOLDEN, "consensus.fna")))]
    rows = run_scan(str(tmp_path), reads)
    gold = [ln.rstrip("\n").split("\t") for ln in open(os.path.join(GOLDEN, "products.tsv"))][1:]
    assert len(rows) == len(gold) == 3
    for got, want in zip(rows, gold):
        g = got.split("\t")
        assert g[:10] == want[:10]  # sequence .. lrt, byte for byte
        assert g[10] == "nan"       # evalue: HMMER (h3daemon) is out of scope
        assert g[11] == want[11]    # match: nucleotides, state, codon and amino of every step, byte for byte


@pytest.mark.parametrize("multi_hits,hmmer3_compat", [(True, False), (False, False), (True, True), (False, True)])
def test_all_reads_all_modes_against_oracle_scan(tmp_path, orc, multi_hits, hmmer3_compat):
    """The 8 reads of c-core/test_consensus.h x the 4 mode combinations of c-core/test_scan.c:15-16."""
    named = read_fasta(os.path.join(GOLDEN, "consensus.fna")) + read_fasta(os.path.join(GOLDEN, "consensus_multi.fna"))
    reads = [(i + 1, s) for i, (_, s) in enumerate(named)]
    rows = run_scan(str(tmp_path), reads, multi_hits, hmmer3_compat)
    assert rows == oracle_scan(orc, read_dcp(DCP), reads, multi_hits, hmmer3_compat)
    assert len(rows) >= 8


def test_sliding_windows_chain_like_test_window(tmp_path, orc):
    """c-core/test_window.c:25-37: a long read made of consensus copies with 70 % of the
    positions randomised, scanned through chained windows (window = 50*K nt)."""
    rng = np.random.default_rng(31)
    cons = read_fasta(os.path.join(GOLDEN, "consensus.fna"))[0][1]
    text = list((cons * 40)[:20000])
    for i in range(len(text)):
        if rng.random() < 0.7:
            text[i] = "ACGT"[rng.integers(0, 4)]
    # keep a few intact copies so that some windows do hit
    for at in (1500, 9000, 16500):
        text[at : at + len(cons)] = cons
    reads = [(7, "".join(text)), (8, cons[:100]), (9, "ACGTN" * 50)]
    rows = run_scan(str(tmp_path), reads)
    want = oracle_scan(orc, read_dcp(DCP), reads, True, False)
    assert rows == want
    assert any(r.split("\t")[1] != "0" for r in rows)  # a hit in a later window
    # dcp_scan_run scores every pair's no-hit chain in one launch and lets only the pairs that hit walk their real
    # chains; with nothing speculated -- every pair round by round -- the file is the same
    os.environ["DECIPHON_HIP_SPECULATE"] = "0"
    try:
        assert run_scan(str(tmp_path / "rounds"), reads) == rows
    finally:
        del os.environ["DECIPHON_HIP_SPECULATE"]


def test_long_reads_with_error_bearing_domains_like_config5(tmp_path, orc):
    """SURVEY 8(d) config 5 in small: 50 kb reads with planted domains carrying 12 % errors
    (8 % substitutions, 2 % insertions, 2 % deletions), about 7 chained windows per
    (profile, read) pair; every product row (window chain, hit span, lrt, every step of every
    path) against the oracle-driven restatement of thread_run."""
    rng = np.random.default_rng(59)
    cons = [t for _, t in read_fasta(os.path.join(GOLDEN, "consensus.fna"))]
    reads = []
    for sid in range(3):
        text = ["ACGT"[i] for i in rng.integers(0, 4, size=50000)]
        for at in sorted(rng.integers(0, 48000, size=7)):
            dom = []
            for ch in cons[int(rng.integers(0, len(cons)))]:
                u = rng.random()
                if u < 0.02:
                    continue
                if u < 0.04:
                    dom.append("ACGT"[rng.integers(0, 4)])
                dom.append("ACGT"[rng.integers(0, 4)] if rng.random() < 0.08 else ch)
            text[at : at + len(dom)] = dom
        reads.append((100 + sid, "".join(text[:50000])))
    rows = run_scan(str(tmp_path), reads)
    want = oracle_scan(orc, read_dcp(DCP), reads, True, False)
    assert rows == want
    assert len(rows) >= 10 and len({r.split("\t")[1] for r in rows}) >= 4  # hits in several windows of the chains


def test_partitions_concatenate_to_the_whole_scan(tmp_path, orc):
    """Contiguous profile partitions (c-core/partition_size.c) scanned separately give, in
    partition order, exactly the rows of the unpartitioned scan (c-core/product.c:63-81)."""
    named = read_fasta(os.path.join(GOLDEN, "consensus.fna"))
    reads = [(i, s) for i, (_, s) in enumerate(named)]
    whole = run_scan(str(tmp_path / "w"), reads)
    parts = []
    for idx in range(2):
        parts += run_scan(str(tmp_path / f"p{idx}"), reads, partition=(0, idx, 2))
    assert parts == whole
    more = []
    for idx in range(5):  # more partitions than profiles: the surplus ones are empty
        d = tmp_path / f"q{idx}"
        more += run_scan(str(d), reads, partition=(0, idx, 5)) if idx < 3 else []
    assert more == whole


def test_errors_and_interrupt(tmp_path):
    from deciphon_amd.scan import Batch, DeciphonError, Scan, Sequence

    with pytest.raises(DeciphonError) as e:
        Scan(str(tmp_path / "nope.dcp"), 0, 1, True, False, False)
    assert e.value.code == 21  # DCP_EOPENDB
    b = Batch()
    with pytest.raises(DeciphonError) as e:
        b.add(Sequence(1, "bad", "ACGTU"))
    assert e.value.code == 74  # DCP_ENUCLTSEQTU
    b.add(Sequence(2, "rna", "ACGUACGU"))
    with Scan(DCP, 0, 1, True, False, False) as scan:
        with pytest.raises(DeciphonError) as e:
            scan.run(str(tmp_path), b)
        assert e.value.code == 72  # DCP_EDBDNASEQRNA
    cons = read_fasta(os.path.join(GOLDEN, "consensus.fna"))[0][1]
    b = Batch()
    b.add(Sequence(3, "ok", cons))
    calls = []
    scan = Scan(DCP, 0, 1, True, False, False, on_window=lambda: calls.append(1))
    with scan:
        scan.run(str(tmp_path), b)
        assert len(calls) == 3  # one callback per window: 3 profiles x 1 read x 1 window


def test_two_ranks_scan_their_partitions_and_gather(tmp_path):
    """deciphon_amd.dist.scan_partitioned under torchrun with 2 ranks (both on this box's one GPU,
    rows gathered over gloo): the gathered products.tsv equals the single-process scan's."""
    import subprocess
    import sys

    from dcp_testlib import ROOT

    reads = read_fasta(os.path.join(GOLDEN, "consensus_multi.fna"))
    fasta = tmp_path / "reads.fna"
    fasta.write_text("".join(f">r{i}\n{t}\n" for i, (_, t) in enumerate(reads)))
    single = run_scan(str(tmp_path / "single"), [(i, t) for i, (_, t) in enumerate(reads)])
    env = dict(os.environ, DECIPHON_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533",
                        os.path.join(ROOT, "scripts", "scan_multi_gpu.py"), DCP, str(fasta), str(tmp_path / "multi")],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = (tmp_path / "multi" / "products.tsv").read_text().splitlines()
    assert lines[1:] == single and len(single) > 0


def test_speculative_chains_with_hits_all_along_long_reads(tmp_path, orc):
    """dcp_scan_run scores every pair's NO-hit window chain in one batch and lets the pairs that did hit walk their real
    chains: the windows after a hit keep their speculated scores only while they are the same windows (c-core/window.c:21-31
    moves the next window's start by the hit's end unless that lies more than 4 K before the window's end), the others are
    scored again in later rounds.  Short real-structured profiles (windows of 1500...8650 nt) against 30 kb reads that
    carry a domain every ~700 nt, so that hits fall at every offset of their windows, many pairs hit several times in a
    row and some windows must be re-scored -- every row against the oracle-driven thread_run, and the same file with
    nothing speculated (every pair round by round)."""
    from deciphon_amd import synth
    from deciphon_amd.scan import Batch, Scan, Sequence

    seeds = synth.load_seeds(DCP)
    Ks = (30, 45, 60, 93, 124, 173)
    prots = [synth.tile_protein(seeds, K, 29 * i, f"SP{K}") for i, K in enumerate(Ks)]
    dcp = str(tmp_path / "short.dcp")
    synth.write_dcp(dcp, prots, 0.01, False, False)
    rng = np.random.default_rng(2025)
    reads = []
    for sid in range(3):
        x = rng.integers(0, 4, size=30000).astype(np.uint8)
        at = int(rng.integers(0, 300))
        while at < 29000:
            p = prots[int(rng.integers(0, len(prots)))]
            dom = synth.mutate(synth.back_translate(p["consensus"]), rng, 0.03, 0.01, 0.01)
            dom = dom[: 30000 - at]
            x[at : at + len(dom)] = dom
            at += len(dom) + int(rng.integers(100, 900))
        reads.append((sid + 1, "".join("ACGT"[v] for v in x)))
    batch = Batch()
    for sid, text in reads:
        batch.add(Sequence(sid, f"r{sid}", text))
    with Scan(dcp, 0, 1, True, False, False) as scan:
        scan.run(str(tmp_path / "spec"), batch)
        rows = scan.products()
        timing = scan.last_timing()
    want = oracle_scan(orc, read_dcp(dcp), reads, True, False)
    assert rows == want
    assert len(rows) >= 60 and len({r.split("\t")[1] for r in rows}) >= 8  # hits in many windows of the chains
    assert timing["rounds"] >= 3 and timing["path_passes"] >= len(rows)  # windows were scored again after hits
    os.environ["DECIPHON_HIP_SPECULATE"] = "0"
    try:
        assert run_scan(str(tmp_path / "rounds"), reads, dbfile=dcp) == rows
    finally:
        del os.environ["DECIPHON_HIP_SPECULATE"]
    # ... and with many small cost batches in flight (chunks of 2e6 cells)
    os.environ["DECIPHON_HIP_CHUNK_CELLS"] = "2e6"
    try:
        assert run_scan(str(tmp_path / "chunks"), reads, dbfile=dcp) == rows
    finally:
        del os.environ["DECIPHON_HIP_CHUNK_CELLS"]
