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
from large_cases import build_case, large_cases

pytestmark = pytest.mark.gpu

CASES = large_cases()


def _run(engine, orc, c, g):
    i = c["idx"]
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
