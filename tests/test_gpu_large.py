"""GPU: every kernel class above one wavefront against bits produced by the REFERENCE's own
viterbi.c (tests/golden/large_classes.npz, made by tests/golden/make_golden.py from oracle/_ref):
K = 257 .. 16383 at the class boundaries, continuous and tie-rich tables, windows of <= 64 nt,
3 kb and -- SURVEY 8d config 3b -- K in {2048, 8192, 16383} on 10 kb reads.  Scores as fp32 bit
patterns, the whole packed trellis by CRC32, the unzipped path step by step, through the C ABI."""
import os
import zlib

import numpy as np
import pytest

from dcp_testlib import GOLDEN, bits
from large_cases import build_case, large_cases, tiled_protein, window_cap_cases

pytestmark = pytest.mark.gpu

CASES = large_cases()


def _run(engine, orc, c, g, i=None):
    i = c["idx"] if i is None else i  # position of the case in the golden arrays
    prof, seq, xt = build_case(c, orc)
    engine.clear_profiles()
    engine.add_profile(prof.K, prof.trans, prof.match, prof.null, prof.bg)
    engine.commit()
    engine.set_sequences([seq])
    engine.set_mode(bool(c["mh"]), bool(c["h3"]))
    s = max(c["L"] // 3, 1)
    if c["quant"]:  # quantised special transitions: handed over (the default table is the product's own xtrans)
        table = np.zeros((s + 1, 13), np.float32)
        table[s] = xt
        engine.set_xtrans_table(table)
    try:
        win = [(0, 0, 0, c["L"])]
        nul, alt = engine.cost(win)
        assert bits(nul[0]) == int(g["null_bits"][i]), c
        assert bits(alt[0]) == int(g["alt_bits"][i]), c
        a, b = int(g["path_off"][i]), int(g["path_off"][i + 1])
        if a == b:  # no finite path: the reference never walks such a trellis
            return
        p = engine.path(win, trellis=True)[0]
        # the steps of the first answer (fast pass: DP table + traceback; row replay on ties) ...
        assert bits(p["score"]) == int(g["alt_bits"][i]), c
        assert np.array_equal(p["state_ids"], g["path_ids"][a:b]), c
        assert np.array_equal(p["seqsizes"], g["path_sizes"][a:b]), c
        # ... and the packed trellis of the pass-by-pass kernels (row replay beyond 4096 positions)
        assert zlib.crc32(p["xnodes"].tobytes()) == int(g["xnodes_crc"][i]), c
        assert zlib.crc32(p["nodes"].tobytes()) == int(g["nodes_crc"][i]), c
        assert bits(p["literal_score"]) == int(g["alt_bits"][i]), c
        assert np.array_equal(p["literal_state_ids"], g["path_ids"][a:b]), c
        assert np.array_equal(p["literal_seqsizes"], g["path_sizes"][a:b]), c
    finally:
        if c["quant"]:
            engine.set_xtrans_table(np.zeros((0, 13), np.float32))


@pytest.mark.parametrize("lo,hi,name", [(0, 22, "short windows"), (22, 29, "3 kb windows"), (29, 32, "10 kb reads")])
def test_long_profiles_against_reference_goldens(engine, orc, lo, hi, name):
    g = np.load(os.path.join(GOLDEN, "large_classes.npz"))
    assert len(CASES) == len(g["K"]) == 32
    for c in CASES[lo:hi]:
        assert (c["K"], c["L"]) == (int(g["K"][c["idx"]]), int(g["L"][c["idx"]]))
        _run(engine, orc, c, g)


def test_windows_at_the_100000_row_cap_against_reference_goldens(engine, orc):
    """c-core/window.c:13: a window never exceeds 100 000 nucleotides; K >= 2001 on a read of 100 kb or more reaches
    that -- the longest DP the scan runs.  tests/golden/window_cap.npz holds the reference viterbi.c's bits for one
    multi-wave class (K = 2048, 2.0e8 cells: path pass in 200 blocks of 500 rows) and one strip class (K = 4200,
    4.2e8 cells: the whole 9.8 GB table, beyond the default 4 GB arena -- a lone table is taken whatever the budget
    says): scores, every trellis word (CRC32 of 410 / 840 MB), the unzipped path."""
    g = np.load(os.path.join(GOLDEN, "window_cap.npz"))
    cases = window_cap_cases()
    assert len(cases) == len(g["K"]) == 2
    for j, c in enumerate(cases):
        assert (c["K"], c["L"]) == (int(g["K"][j]), int(g["L"][j])) and c["L"] == 100000
        _run(engine, orc, c, g, j)


def test_a_table_beyond_a_strict_budget_is_a_clean_enomem(orc, monkeypatch):
    """DECIPHON_HIP_PATH_STRICT=1 makes the path pass's HBM budget (DECIPHON_HIP_PATH_BUDGET_MB) a hard limit: a window
    whose DP table alone exceeds it is refused with DCP_ENOMEM -- what the reference returns when trellis_setup's
    realloc fails (c-core/trellis.c:22-40) -- and the engine stays usable."""
    import deciphon_amd

    c = dict(idx=102, K=4200, L=3000, kind="tiled", quant=None, pinf=0.0, mh=1, h3=0)
    prof, seq, _ = build_case(c, orc)
    with deciphon_amd.Engine(0) as engine:  # a fresh engine: its table arena holds nothing yet
        engine.add_profile(prof.K, prof.trans, prof.match, prof.null, prof.bg)
        engine.commit()
        engine.set_sequences([seq])
        engine.set_mode(True, False)
        win = [(0, 0, 0, c["L"])]
        monkeypatch.setenv("DECIPHON_HIP_PATH_BUDGET_MB", "64")  # the table takes 3001 x (8 + 3 x 8192) x 4 B = 295 MB
        monkeypatch.setenv("DECIPHON_HIP_PATH_STRICT", "1")
        with pytest.raises(deciphon_amd.HipError) as e:
            engine.path(win, trellis=False)
        assert e.value.code == 20  # DCP_ENOMEM
        monkeypatch.delenv("DECIPHON_HIP_PATH_STRICT")
        p = engine.path(win, trellis=False)[0]  # not strict: the lone table is taken beyond the budget
        xt = orc.xtrans(c["L"] // 3, True, False)
        alt, xn, nd = orc.path(prof, xt, seq)
        assert bits(p["score"]) == bits(alt)
        ids, sizes = orc.unzip(prof.K, len(seq), xn, nd)
        assert np.array_equal(p["state_ids"], ids) and np.array_equal(p["seqsizes"], sizes)


def test_scan_of_a_120_kb_read_cuts_windows_at_the_cap(tmp_path, orc):
    """The window chain of c-core/window.c on a read longer than the cap, through dcp_scan_run: K = 2048 (50 K = 102 400
    > 100 000) against a 120 kb read whose first 100 000 nucleotides are the golden case's window.  Window 0 is
    [0, 100000), the next starts where window.c:21-31 puts it after the hit; every row (window ranges, hit spans, lrt,
    every step) equals the oracle-driven thread_run."""
    from dcp_testlib import oracle_scan
    from deciphon_amd import synth
    from deciphon_amd.scan import Batch, Scan, Sequence
    from oracle.dcp_reader import read_dcp

    c = window_cap_cases()[0]
    _, seq, _ = build_case(c, orc)
    rng = np.random.default_rng(5)
    tail = rng.integers(0, 4, size=20000).astype(np.uint8)
    text = "".join("ACGT"[v] for v in np.concatenate([seq, tail]))
    dcp = str(tmp_path / "k2048.dcp")
    synth.write_dcp(dcp, [tiled_protein(c)], 0.01, False, False)
    batch = Batch()
    batch.add(Sequence(11, "long", text))
    with Scan(dcp, 0, 1, True, False, False) as scan:
        scan.run(str(tmp_path / "prod"), batch)
        rows = scan.products()
    want = oracle_scan(orc, read_dcp(dcp).proteins, [(11, text)], True, False)
    assert rows == want
    first = rows[0].split("\t")
    assert first[1:4] == ["0", "0", "100000"]  # the cap
    g = np.load(os.path.join(GOLDEN, "window_cap.npz"))
    nul, alt = (np.array([g[k][0]], np.uint32).view(np.float32)[0] for k in ("null_bits", "alt_bits"))
    assert first[9] == f"{orc.lrt(-nul, -alt):.1f}"  # the reference viterbi.c's scores for that window
    assert any(r.split("\t")[1] == "1" and int(r.split("\t")[3]) == 120000 for r in rows) or len(rows) >= 1
