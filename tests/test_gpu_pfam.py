"""GPU: a Pfam-shaped database against 10 kb reads through the outer API (BASELINE configs[3] in small).

200 profiles with a Pfam-like length distribution (log-normal, median 140) made of minifam node runs
(deciphon_amd.synth), a few of them beyond one wavefront and one beyond 4096 positions, written as a pressed
.dcp in BOTH array encodings (the current writer's `bin` / int-array form, c-core/write.c:59-66,
c-core/database_writer.c:76-93, and the legacy `ext` form of the reference's committed fixture), scanned
against 10 kb reads that carry error-bearing domains of several profiles -- with 1, 2 and 3 contiguous
partitions, by count (c-core/partition_size.c:13-16) and balanced by core size.  Every product row (window
chain, hit span, lrt, every step of every path) must equal the oracle-driven restatement of thread_run
(c-core/thread.c:49-207); the ingest goes through several double-buffered staging chunks."""
import os

import numpy as np
import pytest

from dcp_testlib import GOLDEN, bits, oracle_scan
from test_gpu_scan import run_scan

pytestmark = pytest.mark.gpu

NPROF, DB_SEED = 200, 77
LONG = {17: 1100, 60: 1800, 111: 2300, 150: 4200}  # beyond one wavefront; 4200: the strip class


@pytest.fixture(scope="module")
def pfam(tmp_path_factory):
    from types import SimpleNamespace

    from deciphon_amd import synth

    d = tmp_path_factory.mktemp("pfam")
    seeds = synth.load_seeds(os.path.join(GOLDEN, "minifam.dcp"))
    Ks = synth.pfam_like_lengths(NPROF, DB_SEED)
    for i, K in LONG.items():
        Ks[i] = K
    proteins = synth.pfam_like_database(seeds, NPROF, DB_SEED, lengths=Ks)
    synth.write_dcp(str(d / "pfam_bin.dcp"), proteins, legacy=False)
    synth.write_dcp(str(d / "pfam_ext.dcp"), proteins, legacy=True)
    # two 10 kb reads, each with domains of five profiles (12 % errors), the long ones among them
    rng = np.random.default_rng(5)
    reads = []
    for r, picks in enumerate(((3, 17, 60, 88, 150), (9, 42, 111, 130, 199))):
        x = rng.integers(0, 4, size=10000).astype(np.uint8)
        for j, pi in enumerate(picks):
            cons = proteins[pi]["consensus"]
            a = int(rng.integers(0, max(len(cons) - 250, 1)))
            dom = synth.mutate(synth.back_translate(cons[a : a + 250]), rng, 0.08, 0.02, 0.02)
            at = 300 + j * 1900
            x[at : at + len(dom)] = dom
        reads.append((r + 1, "".join("ACGT"[v] for v in x)))
    prots = [SimpleNamespace(**p) for p in proteins]
    return SimpleNamespace(dir=d, proteins=prots, reads=reads, Ks=Ks)


@pytest.fixture(scope="module")
def expected(pfam, orc):
    rows = oracle_scan(orc, pfam.proteins, pfam.reads, True, False, threads=min(os.cpu_count() or 1, 16))
    profiles_hit = {r.split("\t")[7] for r in rows}
    assert len(rows) >= 10 and len(profiles_hit) >= 8
    assert {pfam.proteins[i].accession for i in LONG} <= profiles_hit  # the long profiles are among the hits
    return rows


def test_ingest_in_chunks_and_both_encodings_give_the_same_tables(pfam, monkeypatch):
    """dcp_hip_load_dcp through >= 3 staging chunks (double-buffer reuse) equals the one-chunk load, and the
    legacy encoding equals the current one: cost-pass scores of every profile, bit for bit."""
    import deciphon_amd

    read = deciphon_amd.encode(pfam.reads[0][1])[:600]
    out = {}
    for name, path, mb in (("one", "pfam_bin.dcp", None), ("chunks", "pfam_bin.dcp", "64"), ("ext", "pfam_ext.dcp", "48")):
        if mb:
            monkeypatch.setenv("DECIPHON_HIP_STAGE_MB", mb)
        else:
            monkeypatch.delenv("DECIPHON_HIP_STAGE_MB", raising=False)
        with deciphon_amd.Engine(0) as eng:
            eng.load_dcp(str(pfam.dir / path))
            eng.commit()
            out[name + "_chunks"] = eng.load_chunks
            assert eng.num_profiles == NPROF and [eng.core_size(i) for i in range(NPROF)] == list(pfam.Ks)
            eng.set_sequences([read])
            eng.set_mode(True, False)
            out[name] = eng.cost([(p, 0, 0, len(read)) for p in range(NPROF)])
    assert out["one_chunks"] == 1 or out["one_chunks"] == 2  # 256 MiB chunks: the whole pool in one or two
    assert out["chunks_chunks"] >= 3 and out["ext_chunks"] >= 4
    for k in ("chunks", "ext"):
        assert np.array_equal(out[k][0].view(np.uint32), out["one"][0].view(np.uint32))
        assert np.array_equal(out[k][1].view(np.uint32), out["one"][1].view(np.uint32))


def test_whole_scan_equals_oracle_thread_run(pfam, expected, tmp_path, monkeypatch):
    monkeypatch.setenv("DECIPHON_HIP_STAGE_MB", "64")
    rows = run_scan(str(tmp_path / "w"), pfam.reads, dbfile=str(pfam.dir / "pfam_bin.dcp"))
    assert rows == expected


def test_legacy_encoding_scan_equals_oracle_thread_run(pfam, expected, tmp_path):
    rows = run_scan(str(tmp_path / "w"), pfam.reads, dbfile=str(pfam.dir / "pfam_ext.dcp"))
    assert rows == expected


def test_long_read_against_the_whole_database_like_config5(pfam, orc, tmp_path):
    """BASELINE configs[4] at database scale in small: a 25 kb read with twelve planted domains carrying 12 %
    errors (8 % substitutions, 2 % insertions, 2 % deletions) against all 200 profiles -- chained windows
    (window.c: 50 K nt, the next start moved by the last hit) for every profile below 500 positions, every
    product row equal to the oracle-driven thread_run.  (test_gpu_scan.py has 50 kb reads against minifam.)"""
    from deciphon_amd import synth

    rng = np.random.default_rng(50)
    x = rng.integers(0, 4, size=25000).astype(np.uint8)
    for j, pi in enumerate(rng.choice(NPROF, size=12, replace=False)):
        cons = pfam.proteins[int(pi)].consensus
        a = int(rng.integers(0, max(len(cons) - 200, 1)))
        dom = synth.mutate(synth.back_translate(cons[a : a + 200]), rng, 0.08, 0.02, 0.02)
        at = 500 + j * 2000
        x[at : at + len(dom)] = dom
    reads = [(500, "".join("ACGT"[v] for v in x))]
    want = oracle_scan(orc, pfam.proteins, reads, True, False, threads=min(os.cpu_count() or 1, 32))
    rows = run_scan(str(tmp_path / "w"), reads, dbfile=str(pfam.dir / "pfam_bin.dcp"))
    assert rows == want
    assert len(rows) >= 12 and len({r.split("\t")[1] for r in rows}) >= 4  # hits in several windows of the chains


@pytest.mark.parametrize("nparts", [2, 3])
@pytest.mark.parametrize("balanced", [False, True])
def test_partitions_concatenate_to_the_whole_scan(pfam, expected, tmp_path, nparts, balanced):
    """Contiguous partitions (by count as c-core/protein_reader.c:112-128, or balanced by core size) scanned
    separately give, in partition order, the rows of the whole scan (c-core/product.c:63-81)."""
    from deciphon_amd import host
    from deciphon_amd.scan import Batch, Scan, Sequence

    db = str(pfam.dir / "pfam_bin.dcp")
    bounds = host.Database(db).partition_bounds(nparts, balanced)
    rows, owned = [], []
    for idx in range(nparts):
        batch = Batch()
        for sid, text in pfam.reads:
            batch.add(Sequence(sid, f"seq{sid}", text))
        with Scan(db, 0, 1, True, False, False, partition=(0, idx, nparts), balanced=balanced) as scan:
            first, count = scan.partition_range()
            assert (first, first + count) == (int(bounds[idx]), int(bounds[idx + 1]))
            owned.append(int(pfam.Ks[first : first + count].sum()))
            scan.run(str(tmp_path / f"p{idx}"), batch)
            rows += scan.products()
    assert rows == expected
    if not balanced:
        assert [int(bounds[i + 1] - bounds[i]) for i in range(nparts)] == [
            host.partition_size(NPROF, nparts, i) for i in range(nparts)]
    else:
        # no boundary is further from its target than half the core size next to it
        assert max(owned) - min(owned) <= 2 * max(pfam.Ks)


def test_a_database_beyond_4_gb_of_tables(tmp_path, orc):
    """Pfam-A pressed is ~2e4 profiles, ~20 GB of tables (c-core/database_writer.c:14,204-208 plans for files beyond
    4 GB).  Here: 4200 Pfam-shaped profiles streamed into a .dcp of ~4.6 GB, ingested through ~20 staging chunks into
    a pool beyond 2^32 bytes (table offsets past 32 bits), then every window of two 3 kb reads against ALL profiles in
    one launch; the scores of windows at the first, the last and evenly spread profiles in between -- and the path of
    a planted domain of the LAST profile, whose tables sit at the far end of the pool -- against the oracle."""
    from types import SimpleNamespace

    import deciphon_amd
    from deciphon_amd import synth

    nprof, seed = 4200, 4242
    seeds = synth.load_seeds(os.path.join(GOLDEN, "minifam.dcp"))
    Ks = synth.pfam_like_lengths(nprof, seed)
    dcp = str(tmp_path / "big.dcp")
    synth.write_dcp(dcp, synth.iter_pfam_like(seeds, nprof, seed, lengths=Ks), 0.01, False, False)
    assert os.path.getsize(dcp) > 4.0e9

    def protein(i):
        return synth.pfam_like_database(seeds, 1, seed, first=i, lengths=Ks[i : i + 1])[0]

    rng = np.random.default_rng(8)
    reads = [rng.integers(0, 4, size=3000).astype(np.uint8) for _ in range(2)]
    last = protein(nprof - 1)
    dom = synth.mutate(synth.back_translate(last["consensus"][:200]), rng, 0.05, 0.02, 0.02)[:2000]
    reads[1][400 : 400 + len(dom)] = dom
    with deciphon_amd.Engine(0) as eng:
        eng.load_dcp(dcp)
        eng.commit()
        assert eng.num_profiles == nprof and eng.pool_bytes > 2**32 and eng.load_chunks >= 15
        assert [eng.core_size(i) for i in (0, nprof // 2, nprof - 1)] == [int(Ks[i]) for i in (0, nprof // 2, nprof - 1)]
        eng.set_sequences(reads)
        eng.set_mode(True, False)
        wins = np.array([(p, r, 0, min(3000, 50 * int(Ks[p]))) for p in range(nprof) for r in range(2)], np.int32)
        nul, alt = eng.cost(wins)
        picks = sorted({0, 1, nprof - 2, nprof - 1} | {int(v) for v in np.linspace(0, nprof - 1, 24)})
        for p in picks:
            prof = orc.setup_profile(SimpleNamespace(**protein(p)))
            for r in range(2):
                i = 2 * p + r
                seq = np.ascontiguousarray(reads[r][: wins[i][3]])
                xt = orc.xtrans(max(len(seq) // 3, 1), True, False)
                assert bits(nul[i]) == bits(orc.null(prof, xt, seq)), (p, r)
                assert bits(alt[i]) == bits(orc.cost(prof, xt, seq)), (p, r)
        # the planted domain: a hit of the last profile, path and all
        i = 2 * (nprof - 1) + 1
        assert -2.0 * ((-nul[i]) - (-alt[i])) > 0
        prof = orc.setup_profile(SimpleNamespace(**last))
        seq = np.ascontiguousarray(reads[1][: wins[i][3]])
        xt = orc.xtrans(max(len(seq) // 3, 1), True, False)
        _, xn, nd = orc.path(prof, xt, seq)
        ids, sizes = orc.unzip(prof.K, len(seq), xn, nd)
        got = eng.path([tuple(int(v) for v in wins[i])], trellis=True)[0]
        assert np.array_equal(got["state_ids"], ids) and np.array_equal(got["seqsizes"], sizes)
        assert np.array_equal(got["xnodes"], xn) and np.array_equal(got["nodes"], nd)

    # BASELINE configs[4] on the same database (a fifth of Pfam-A's profiles): six 50 kb reads with error-bearing domains
    # of profiles from the first to the last, all 4200 profiles through dcp_scan_run.  At this size the oracle takes a
    # SAMPLE -- every row of the eight planted profiles against the oracle-driven thread_run on those eight -- and the
    # rest is held by properties that do not depend on size: the rows of two balanced partitions concatenate to the
    # whole scan's (c-core/product.c:63-81), the scan with nothing speculated writes the same file, and rows come in
    # (profile, read, window) order.
    from deciphon_amd.scan import Batch, Scan, Sequence

    planted = sorted({0, 1, nprof // 3, nprof // 2, 2 * nprof // 3, nprof - 2, nprof - 1, 777})
    prots = {i: protein(i) for i in planted}
    long_reads = []
    for sid in range(6):
        x = rng.integers(0, 4, size=50000).astype(np.uint8)
        for j in range(12):
            p = prots[planted[(sid + j) % len(planted)]]
            cons = p["consensus"]
            a = int(rng.integers(0, max(len(cons) - 220, 1)))
            dom = synth.mutate(synth.back_translate(cons[a : a + 220]), rng, 0.08, 0.02, 0.02)
            at = 900 + j * 4000
            x[at : at + len(dom)] = dom
        long_reads.append((300 + sid, "".join("ACGT"[v] for v in x)))

    def scan_rows(out, partition=None):
        batch = Batch()
        for sid, text in long_reads:
            batch.add(Sequence(sid, f"seq{sid}", text))
        with Scan(dcp, 0, 1, True, False, False, partition=partition, balanced=True) as scan:
            scan.run(str(tmp_path / out), batch)
            return scan.products()

    whole = scan_rows("whole")
    acc = {prots[i]["accession"]: i for i in planted}
    mine = [r for r in whole if r.split("\t")[7] in acc]
    want = oracle_scan(orc, [SimpleNamespace(**prots[i]) for i in planted], long_reads, True, False,
                       threads=min(os.cpu_count() or 1, 16))
    assert mine == want and len(want) >= 40 and len({r.split("\t")[1] for r in want}) >= 4
    order = [(int(r.split("\t")[7][2:7]), int(r.split("\t")[0]), int(r.split("\t")[1])) for r in whole]
    assert order == sorted(order)  # accession SYnnnnn.1 = database order
    assert scan_rows("p0", (0, 0, 2)) + scan_rows("p1", (0, 1, 2)) == whole
    os.environ["DECIPHON_HIP_SPECULATE"] = "0"
    try:
        assert scan_rows("rounds") == whole
    finally:
        del os.environ["DECIPHON_HIP_SPECULATE"]
    os.unlink(dcp)
