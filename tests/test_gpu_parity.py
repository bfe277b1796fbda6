"""GPU parity: the HIP path, called through the C ABI (include/deciphon_hip.h), against
the CPU oracle on the same inputs and against goldens produced by the reference's own
viterbi.c (tests/golden/make_golden.py).  Bit-exact: scores as fp32 bit patterns, every
trellis word, every path step."""
import os
import sys
import zlib

import numpy as np
import pytest

from dcp_testlib import GOLDEN, bits, random_seq, read_fasta, synth_profile

sys.path.insert(0, GOLDEN)
from make_golden import MODES, synth_case_params, synth_xt  # noqa: E402

pytestmark = pytest.mark.gpu


def _reads():
    return read_fasta(os.path.join(GOLDEN, "consensus.fna")) + read_fasta(os.path.join(GOLDEN, "consensus_multi.fna"))


def test_device_present():
    import deciphon_amd

    assert deciphon_amd.device_count() >= 1


def test_synthetic_tie_rich_cases(engine, orc):
    """240 small random profiles with costs quantised to multiples of 0.5..8 (exact fp32
    ties between distinct candidates everywhere), K = 2..256, both window modes."""
    g = np.load(os.path.join(GOLDEN, "synth_ties.npz"))
    rng = np.random.default_rng(int(g["seed"]))
    cases = []
    for it in range(int(g["ncase"])):
        K, L, quant, pinf, mh, h3 = synth_case_params(rng, it)
        prof = synth_profile(rng, K, quant, pinf)
        seq = random_seq(rng, L)
        cases.append((it, K, L, quant, mh, h3, prof, seq))
        assert K == int(g["K"][it]) and L == int(g["L"][it])
    groups = {}
    for c in cases:
        groups.setdefault((c[3], c[4], c[5]), []).append(c)
    for (quant, mh, h3), grp in groups.items():
        engine.clear_profiles()
        for c in grp:
            engine.add_profile(c[6].K, c[6].trans, c[6].match, c[6].null, c[6].bg)
        engine.commit()
        engine.set_sequences([c[7] for c in grp])
        engine.set_mode(bool(mh), bool(h3))
        smax = max(max(c[2] // 3, 1) for c in grp)
        table = np.zeros((smax + 1, 13), np.float32)
        for s in range(1, smax + 1):
            table[s] = synth_xt(orc, 3 * s, mh, h3, quant)
        engine.set_xtrans_table(table)
        wins = [(i, i, 0, c[2]) for i, c in enumerate(grp)]
        nul, alt = engine.cost(wins)
        paths = engine.path(wins)
        for i, c in enumerate(grp):
            it, K, L, _, _, _, prof, seq = c
            xt = synth_xt(orc, L, mh, h3, quant)
            assert bits(nul[i]) == bits(orc.null(prof, xt, seq)) == int(g["null_bits"][it]), (it, K, L)
            assert bits(alt[i]) == bits(orc.cost(prof, xt, seq)) == int(g["alt_bits"][it]), (it, K, L)
            score, xo, no = orc.path(prof, xt, seq)
            assert bits(paths[i]["score"]) == bits(score), (it, K, L)
            assert np.array_equal(paths[i]["xnodes"], xo), (it, K, L)
            assert np.array_equal(paths[i]["nodes"], no), (it, K, L)
            assert zlib.crc32(paths[i]["xnodes"].tobytes()) == int(g["xnodes_crc"][it])
            assert zlib.crc32(paths[i]["nodes"].tobytes()) == int(g["nodes_crc"][it])
            ids, sizes = orc.unzip(K, L, xo, no)
            assert np.array_equal(paths[i]["state_ids"], ids) and np.array_equal(paths[i]["seqsizes"], sizes)
    engine.set_xtrans_table(np.zeros((0, 13), np.float32))


def test_minifam_consensus_against_reference_goldens(engine, orc):
    """BASELINE config 1: minifam.dcp x the 8 consensus reads x the 4 mode combinations of
    c-core/test_scan.c:15-16; expected values come from the reference's own viterbi.c."""
    import deciphon_amd

    g = np.load(os.path.join(GOLDEN, "minifam_consensus.npz"))
    engine.clear_profiles()
    engine.load_dcp(os.path.join(GOLDEN, "minifam.dcp"))
    engine.commit()
    assert engine.num_profiles == 3
    assert [engine.core_size(i) for i in range(3)] == [173, 241, 162]
    assert [engine.accession(i) for i in range(3)] == ["PF00742.20", "PF00696.29", "PF16620.6"]
    reads = [deciphon_amd.encode(s) for _, s in _reads()]
    engine.set_sequences(reads)
    for mh, h3 in MODES:
        engine.set_mode(bool(mh), bool(h3))
        sel = np.nonzero((g["multi_hits"] == mh) & (g["hmmer3_compat"] == h3))[0]
        wins = [(int(g["profile"][j]), int(g["read"][j]), 0, len(reads[int(g["read"][j])])) for j in sel]
        nul, alt = engine.cost(wins)
        for i, j in enumerate(sel):
            assert bits(nul[i]) == int(g["null_bits"][j]), (mh, h3, wins[i])
            assert bits(alt[i]) == int(g["alt_bits"][j]), (mh, h3, wins[i])
        hit = [i for i, j in enumerate(sel) if np.isfinite(g["lrt"][j]) and g["lrt"][j] >= 0]
        paths = engine.path([wins[i] for i in hit])
        for p, i in zip(paths, hit):
            j = sel[i]
            assert bits(p["score"]) == int(g["alt_bits"][j])
            assert zlib.crc32(p["xnodes"].tobytes()) == int(g["xnodes_crc"][j]), (mh, h3, wins[i])
            assert zlib.crc32(p["nodes"].tobytes()) == int(g["nodes_crc"][j]), (mh, h3, wins[i])
            a, b = int(g["path_off"][j]), int(g["path_off"][j + 1])
            assert np.array_equal(p["state_ids"], g["path_ids"][a:b])
            assert np.array_equal(p["seqsizes"], g["path_sizes"][a:b])


def test_products_tsv_of_the_reference(engine, orc):
    """The reference's committed scan output (control/tests/files/snap.dcs): lrt to the
    printed precision, window/hit ranges and the state name + subsequence of every match."""
    import deciphon_amd

    engine.clear_profiles()
    engine.load_dcp(os.path.join(GOLDEN, "minifam.dcp"))
    engine.commit()
    named = read_fasta(os.path.join(GOLDEN, "consensus.fna"))
    reads = [deciphon_amd.encode(s) for _, s in named]
    engine.set_sequences(reads)
    engine.set_mode(True, False)
    acc = {engine.accession(i): i for i in range(engine.num_profiles)}
    rows = [line.rstrip("\n").split("\t") for line in open(os.path.join(GOLDEN, "products.tsv"))][1:]
    assert len(rows) == 3
    for row in rows:
        seq_id, win, wstart, wstop, hit, hstart, hstop, profile, abc, lrt, _evalue, match = row
        si, pi = int(seq_id), acc[profile]
        w = (pi, si, int(wstart), int(wstop))
        nul, alt = engine.cost([w])
        assert f"{orc.lrt(-nul[0], -alt[0]):.1f}" == lrt
        p = engine.path([w])[0]
        h, _last = orc.hits(p["state_ids"], p["seqsizes"])
        assert h is not None and (h[0], h[1]) == (int(hstart), int(hstop))
        pos, got = h[0], []
        for st, sz in zip(p["state_ids"][h[2] : h[3]], p["seqsizes"][h[2] : h[3]]):
            got.append((named[si][1][int(wstart) + pos : int(wstart) + pos + sz], orc.state_name(st)))
            pos += sz
        want = [tuple(m.split(",")[:2]) for m in match.split(";")]
        assert got == want


def test_one_position_per_lane_less_than_the_class_layout(engine, orc, monkeypatch):
    """K <= 320 / 448 / 640 run (5,1) / (7,1) / (10,1) -- ONE wavefront of ten positions per lane -- on the tables of
    (6,1) / (8,1) / (6,2) (dcp_launch_cost_narrow): the oracle's bits, and the bits of the class's own shape
    (DECIPHON_HIP_NARROW=0)."""
    rng = np.random.default_rng(404)
    Ks = (257, 300, 320, 321, 384, 385, 448, 449, 512, 513, 600, 640, 641, 768)
    profs = [synth_profile(rng, K, [None, 2.0][i % 2], [0.0, 0.05][i % 2]) for i, K in enumerate(Ks)]
    for p in profs[::3]:  # cheap delete runs: carries across lanes and across the two wavefronts
        p.trans[7, 1:] = np.float32(0.01)
        p.trans[3, 1:] = np.float32(0.02)
    engine.clear_profiles()
    for p in profs:
        engine.add_profile(p.K, p.trans, p.match, p.null, p.bg)
    engine.commit()
    reads = [random_seq(rng, n) for n in (3, 40, 181, 400)]
    engine.set_sequences(reads)
    engine.set_mode(True, False)
    wins = [(pi, si, 0, len(r)) for pi in range(len(profs)) for si, r in enumerate(reads)]
    monkeypatch.delenv("DECIPHON_HIP_NARROW", raising=False)
    nul, alt = engine.cost(wins)
    monkeypatch.setenv("DECIPHON_HIP_NARROW", "0")
    nul0, alt0 = engine.cost(wins)
    monkeypatch.delenv("DECIPHON_HIP_NARROW")
    assert np.array_equal(nul.view(np.uint32), nul0.view(np.uint32))
    assert np.array_equal(alt.view(np.uint32), alt0.view(np.uint32))
    for i, (pi, si, a, b) in enumerate(wins):
        xt = orc.xtrans(max((b - a) // 3, 1), True, False)
        assert bits(nul[i]) == bits(orc.null(profs[pi], xt, reads[si])), wins[i]
        assert bits(alt[i]) == bits(orc.cost(profs[pi], xt, reads[si])), wins[i]


def test_windows_inside_reads_and_ragged_batch(engine, orc):
    """Windows that start mid-read (the t-mers before the window start must not leak in),
    lengths 1..5 (fewer than five emission lengths), and profiles of every Q class in one call."""
    rng = np.random.default_rng(99)
    profs = [synth_profile(rng, K) for K in (2, 3, 64, 65, 128, 130, 192, 200, 256)]
    engine.clear_profiles()
    for p in profs:
        engine.add_profile(p.K, p.trans, p.match, p.null, p.bg)
    engine.commit()
    reads = [random_seq(rng, n) for n in (1, 2, 5, 17, 200, 333)]
    engine.set_sequences(reads)
    engine.set_mode(True, False)
    wins = []
    for pi in range(len(profs)):
        for si, r in enumerate(reads):
            n = len(r)
            wins.append((pi, si, 0, n))
            for _ in range(3):
                a = int(rng.integers(0, n))
                b = int(rng.integers(a + 1, n + 1))
                wins.append((pi, si, a, b))
            for L in (1, 2, 3, 4, 5):
                if n >= L:
                    wins.append((pi, si, n - L, n))
    nul, alt = engine.cost(wins)
    paths = engine.path(wins)
    for i, (pi, si, a, b) in enumerate(wins):
        seq = np.ascontiguousarray(reads[si][a:b])
        xt = orc.xtrans(max((b - a) // 3, 1), True, False)
        assert bits(nul[i]) == bits(orc.null(profs[pi], xt, seq)), wins[i]
        assert bits(alt[i]) == bits(orc.cost(profs[pi], xt, seq)), wins[i]
        score, xo, no = orc.path(profs[pi], xt, seq)
        assert bits(paths[i]["score"]) == bits(score)
        assert np.array_equal(paths[i]["xnodes"], xo) and np.array_equal(paths[i]["nodes"], no), wins[i]


def test_long_profiles_multi_wave_kernels(engine, orc):
    """K = 257..4096 run as workgroups of 2..16 wavefronts that exchange boundary values
    through LDS; includes delete runs crossing wave boundaries and exact ties."""
    rng = np.random.default_rng(23)
    profs, quants = [], []
    for it, K in enumerate((257, 300, 511, 512, 513, 700, 1024, 1025, 1500, 2048, 2049, 3000, 4096, 260, 1030, 2100)):
        quant = [None, 1.0, 4.0][it % 3]
        p = synth_profile(rng, K, quant, [0, 0.05][it % 2])
        if it % 4 == 0:
            p.trans[7, 1:] = np.float32(0.01)
            p.trans[3, 1:] = np.float32(0.02)
            p.trans[1, 1:] = np.float32(9.0)
            p.match[:, K // 3:] += np.float32(30.0)
        profs.append(p)
        quants.append(quant)
    engine.clear_profiles()
    for p in profs:
        engine.add_profile(p.K, p.trans, p.match, p.null, p.bg)
    engine.commit()
    reads = [random_seq(rng, n) for n in (1, 4, 5, 9, 33, 64)]
    engine.set_sequences(reads)
    engine.set_mode(True, False)
    wins = [(pi, si, 0, len(r)) for pi in range(len(profs)) for si, r in enumerate(reads)]
    nul, alt = engine.cost(wins)
    paths = engine.path(wins)
    for i, (pi, si, a, b) in enumerate(wins):
        seq = reads[si]
        xt = orc.xtrans(max(len(seq) // 3, 1), True, False)
        assert bits(nul[i]) == bits(orc.null(profs[pi], xt, seq)), wins[i]
        assert bits(alt[i]) == bits(orc.cost(profs[pi], xt, seq)), (wins[i], profs[pi].K)
        score, xo, no = orc.path(profs[pi], xt, seq)
        assert bits(paths[i]["score"]) == bits(score), (wins[i], profs[pi].K)
        assert np.array_equal(paths[i]["xnodes"], xo), (wins[i], profs[pi].K)
        assert np.array_equal(paths[i]["nodes"], no), (wins[i], profs[pi].K)
        # the steps come from the fast pass (cost kernel of the class with the DP table kept + traceback)
        ids, sizes = orc.unzip(profs[pi].K, len(seq), xo, no)
        assert np.array_equal(paths[i]["state_ids"], ids) and np.array_equal(paths[i]["seqsizes"], sizes), wins[i]
    with pytest.raises(Exception):
        big = synth_profile(rng, 16384)  # state ids keep 14 bits for k + 1
        engine.add_profile(big.K, big.trans, big.match, big.null, big.bg)


def test_profiles_beyond_4096_strip_by_strip(engine, orc):
    """K = 4097..16383: one workgroup walks every row in strips of 2048 positions with the
    folded rows in HBM (StripWave).  Scores, the fast path pass and the packed trellis (replayed
    from the DP table) against the oracle."""
    import deciphon_amd

    rng = np.random.default_rng(53)
    Ks = (4097, 6000, 8192, 12289, 16383)
    profs = []
    for i, K in enumerate(Ks):
        p = synth_profile(rng, K, None, [0, 0.02][i % 2])
        if i % 2 == 0:  # delete runs that cross strips
            p.trans[7, 1:] = np.float32(0.01)
            p.trans[3, 1:] = np.float32(0.02)
            p.match[:, K // 2:] += np.float32(20.0)
        profs.append(p)
    small = synth_profile(rng, 200, None, 0.0)
    seqs = [random_seq(rng, int(n)) for n in (1, 7, 23, 40)]
    engine.clear_profiles()
    for p in profs + [small]:
        engine.add_profile(p.K, p.trans, p.match, p.null, p.bg)
    engine.commit()
    engine.set_sequences(seqs)
    engine.set_mode(True, False)
    wins = [(pi, si, 0, len(seqs[si])) for pi in range(len(profs) + 1) for si in range(len(seqs))]
    wins.append((1, 3, 5, 30))
    nul, alt = engine.cost(wins)
    allp = profs + [small]
    for i, (pi, si, a, b) in enumerate(wins):
        seq = np.ascontiguousarray(seqs[si][a:b])
        xt = orc.xtrans(max(len(seq) // 3, 1), True, False)
        assert bits(nul[i]) == bits(orc.null(allp[pi], xt, seq)), wins[i]
        assert bits(alt[i]) == bits(orc.cost(allp[pi], xt, seq)), (wins[i], allp[pi].K)
    paths = engine.path(wins, trellis=False)
    assert engine.path_redone == 0
    for i, (pi, si, a, b) in enumerate(wins):
        seq = np.ascontiguousarray(seqs[si][a:b])
        xt = orc.xtrans(max(len(seq) // 3, 1), True, False)
        score, xo, no = orc.path(allp[pi], xt, seq)
        ids, sizes = orc.unzip(allp[pi].K, len(seq), xo, no)
        assert bits(paths[i]["score"]) == bits(score)
        assert np.array_equal(paths[i]["state_ids"], ids) and np.array_equal(paths[i]["seqsizes"], sizes), wins[i]
    # the packed trellis (and exact ties): replayed row by row from the DP table (row_replay.h)
    sel = [wins[i] for i in (0, 3, 5, 10, 14, 19, len(wins) - 1)]
    full = engine.path(sel, trellis=True)
    for (pi, si, a, b), r in zip(sel, full):
        seq = np.ascontiguousarray(seqs[si][a:b])
        xt = orc.xtrans(max(len(seq) // 3, 1), True, False)
        score, xo, no = orc.path(allp[pi], xt, seq)
        assert bits(r["literal_score"]) == bits(score)
        assert np.array_equal(r["xnodes"], xo) and np.array_equal(r["nodes"], no), (pi, si)
        ids, sizes = orc.unzip(allp[pi].K, len(seq), xo, no)
        assert np.array_equal(r["literal_state_ids"], ids) and np.array_equal(r["literal_seqsizes"], sizes)


def test_ties_beyond_4096_go_through_the_row_replay(engine, orc):
    """Quantised tables on K = 5000 and 9000: the fast pass gives up on exact ties and the
    windows are redone from the DP table row by row -- same steps as the oracle."""
    rng = np.random.default_rng(61)
    profs = [synth_profile(rng, K, 2.0, 0.02) for K in (5000, 9000)]
    seqs = [random_seq(rng, int(n)) for n in (6, 21, 33)]
    engine.clear_profiles()
    for p in profs:
        engine.add_profile(p.K, p.trans, p.match, p.null, p.bg)
    engine.commit()
    engine.set_sequences(seqs)
    engine.set_mode(True, False)
    smax = max(max(len(s) // 3, 1) for s in seqs)
    table = np.zeros((smax + 1, 13), np.float32)
    for s in range(1, smax + 1):
        table[s] = synth_xt(orc, 3 * s, 1, 0, 2.0)
    engine.set_xtrans_table(table)
    wins = [(pi, si, 0, len(seqs[si])) for pi in range(len(profs)) for si in range(len(seqs))]
    res = engine.path(wins, trellis=False)
    assert engine.path_redone > 0
    for (pi, si, a, b), r in zip(wins, res):
        xt = synth_xt(orc, len(seqs[si]), 1, 0, 2.0)
        score, xo, no = orc.path(profs[pi], xt, seqs[si])
        ids, sizes = orc.unzip(profs[pi].K, len(seqs[si]), xo, no)
        assert bits(r["score"]) == bits(score)
        assert np.array_equal(r["state_ids"], ids) and np.array_equal(r["seqsizes"], sizes), (pi, si)
    engine.set_xtrans_table(np.zeros((0, 13), np.float32))


def test_empty_and_invalid_calls(engine):
    import deciphon_amd

    nul, alt = engine.cost([])
    assert len(nul) == 0 and len(alt) == 0
    with pytest.raises(deciphon_amd.HipError) as e:
        engine.cost([(10 ** 6, 0, 0, 1)])
    assert e.value.code == 8  # DCP_EFUNCUSE
    with pytest.raises(deciphon_amd.HipError) as e:
        engine.cost([(0, 0, 0, 0)])
    assert e.value.code == 11  # DCP_EZEROSEQ
    with pytest.raises(deciphon_amd.HipError) as e:
        engine.cost([(0, 0, 0, 10 ** 7)])
    assert e.value.code == 8
    # costs are -log-probabilities: a negative delete cost would break E = min M and is refused
    bad = synth_profile(np.random.default_rng(1), 40)
    bad.trans[7, 5] = np.float32(-0.5)
    with pytest.raises(deciphon_amd.HipError) as e:
        engine.add_profile(bad.K, bad.trans, bad.match, bad.null, bad.bg)
    assert e.value.code == 8


def test_full_size_config2_properties(engine, orc):
    """BASELINE configs[1] at full size (minifam x 1000 synthetic 3 kb reads = the bench
    workload): a random sample of windows against the oracle, bit-identical results across
    two runs and across the staged (bench) entry points, and the planted domains are found."""
    import bench
    from deciphon_amd import synth
    from oracle.dcp_reader import read_dcp

    db = read_dcp(os.path.join(GOLDEN, "minifam.dcp"))
    reads = synth.synth_reads(1000, 3000, [p.consensus for p in db.proteins], bench.SEED)
    engine.clear_profiles()
    engine.load_dcp(os.path.join(GOLDEN, "minifam.dcp"))
    engine.commit()
    engine.set_sequences(reads)
    engine.set_mode(True, False)
    wins = [(p, s, 0, 3000) for p in range(3) for s in range(1000)]
    nul, alt = engine.cost(wins)
    nul2, alt2 = engine.cost(wins)
    assert np.array_equal(nul.view(np.uint32), nul2.view(np.uint32))
    assert np.array_equal(alt.view(np.uint32), alt2.view(np.uint32))
    engine.stage(wins)
    engine.run_staged(2)
    nul3, alt3 = engine.fetch_staged()
    assert np.array_equal(nul.view(np.uint32), nul3.view(np.uint32))
    assert np.array_equal(alt.view(np.uint32), alt3.view(np.uint32))
    # the null model does not depend on the profile (minifam shares one null table)
    assert np.array_equal(nul[:1000].view(np.uint32), nul[1000:2000].view(np.uint32))
    rng = np.random.default_rng(5)
    profs = [orc.setup_profile(p) for p in db.proteins]
    xt = orc.xtrans(1000, True, False)
    for i in rng.choice(len(wins), size=24, replace=False):
        p, s, _, _ = wins[i]
        assert bits(nul[i]) == bits(orc.null(profs[p], xt, reads[s]))
        assert bits(alt[i]) == bits(orc.cost(profs[p], xt, reads[s]))
    lrt = -2.0 * ((-nul) - (-alt))
    planted = [(s // 10 % 3, s) for s in range(0, 1000, 10)]  # synth_reads: read s carries profile (s/10)%3
    found = sum(lrt[p * 1000 + s] > 0 for p, s in planted)
    assert found >= 90, found
    others = np.ones(3000, bool)
    for p, s in planted:
        others[p * 1000 + s] = False
    assert (lrt[others] > 0).mean() < 0.02


def test_tiny_profile_many_windows_like_massive(engine, orc):
    """BASELINE configs[2]: massive.hmm is a K = 3 profile (c-core/massive.hmm:2-5); a 10 kb
    read gives 72 chained windows of 150 nt.  A synthetic K = 3 profile stands in (pressing
    the HMM needs the absent imm library)."""
    from deciphon_amd import host

    rng = np.random.default_rng(8)
    prof = synth_profile(rng, 3)
    engine.clear_profiles()
    engine.add_profile(3, prof.trans, prof.match, prof.null, prof.bg)
    engine.commit()
    reads = [random_seq(rng, 10000) for _ in range(20)] + [random_seq(rng, n) for n in range(1, 22)]
    engine.set_sequences(reads)
    engine.set_mode(True, False)
    wins = []
    for s, r in enumerate(reads):
        it = host.WindowIter(len(r), 3)
        while (w := it.next()) is not None:
            wins.append((0, s, w[1], w[2]))
    assert len(wins) > 20 * 70
    nul, alt = engine.cost(wins)
    for i in list(rng.choice(len(wins), size=200, replace=False)) + list(range(len(wins) - 21, len(wins))):
        _, s, a, b = wins[i]
        seq = np.ascontiguousarray(reads[s][a:b])
        xt = orc.xtrans(max((b - a) // 3, 1), True, False)
        assert bits(nul[i]) == bits(orc.null(prof, xt, seq)) and bits(alt[i]) == bits(orc.cost(prof, xt, seq))


def test_device_unzip_and_host_fallback(engine, orc, monkeypatch):
    """trellis_unzip runs on the GPU; a path that does not fit its step buffer falls back to
    the host unzip of the fetched trellis.  Both must give the reference's steps."""
    import deciphon_amd

    engine.clear_profiles()
    engine.load_dcp(os.path.join(GOLDEN, "minifam.dcp"))
    engine.commit()
    reads = [deciphon_amd.encode(s) for _, s in _reads()]
    engine.set_sequences(reads)
    engine.set_mode(True, False)
    wins = [(0, 0, 0, len(reads[0])), (1, 1, 0, len(reads[1])), (0, 3, 0, len(reads[3])), (2, 2, 10, 400)]
    normal = engine.path(wins)
    monkeypatch.setenv("DECIPHON_HIP_UNZIP_CAP", "50")  # far too small: every path overflows
    fallback = engine.path(wins)
    monkeypatch.delenv("DECIPHON_HIP_UNZIP_CAP")
    from oracle.dcp_reader import read_dcp

    db = read_dcp(os.path.join(GOLDEN, "minifam.dcp"))
    for (p, s, a, b), x, y in zip(wins, normal, fallback):
        prof = orc.setup_profile(db.proteins[p])
        seq = np.ascontiguousarray(reads[s][a:b])
        xt = orc.xtrans(max(len(seq) // 3, 1), True, False)
        _, xo, no = orc.path(prof, xt, seq)
        ids, sizes = orc.unzip(prof.K, len(seq), xo, no)
        for r in (x, y):
            assert np.array_equal(r["state_ids"], ids) and np.array_equal(r["seqsizes"], sizes)
            assert np.array_equal(r["xnodes"], xo) and np.array_equal(r["nodes"], no)


def test_fast_path_pass_equals_literal_pass(engine, orc):
    """The path pass has two implementations: traceback from the stored DP values (default)
    and the literal pass-by-pass kernel (on demand / on ties).  Same steps, same score; the
    fast one may only give up (and be redone literally) where an exact tie needs pass order."""
    rng = np.random.default_rng(41)
    profs, seqs, quants = [], [], []
    for it in range(60):
        K = int(rng.choice([2, 3, 9, 33, 64, 65, 100, 173, 192, 241, 256, 300, 600, 1100]))
        quant = [None, None, None, 0.5, 2.0, 8.0][it % 6]
        profs.append(synth_profile(rng, K, quant, [0, 0.05, 0.3][it % 3]))
        seqs.append(random_seq(rng, int(rng.integers(1, 70))))
        quants.append(quant)
    engine.clear_profiles()
    for p in profs:
        engine.add_profile(p.K, p.trans, p.match, p.null, p.bg)
    engine.commit()
    engine.set_sequences(seqs)
    engine.set_mode(True, False)
    wins = [(i, i, 0, len(seqs[i])) for i in range(len(profs))]
    res = engine.path(wins)
    assert 0 < engine.path_redone < len(wins)  # quantised tables tie, continuous ones do not
    for i, r in enumerate(res):
        xt = orc.xtrans(max(len(seqs[i]) // 3, 1), True, False)
        score, xo, no = orc.path(profs[i], xt, seqs[i])
        ids, sizes = orc.unzip(profs[i].K, len(seqs[i]), xo, no)
        assert np.array_equal(r["state_ids"], ids) and np.array_equal(r["seqsizes"], sizes), (i, profs[i].K, quants[i])
        assert np.array_equal(r["literal_state_ids"], ids) and np.array_equal(r["literal_seqsizes"], sizes)
        assert bits(r["score"]) == bits(score) == bits(r["literal_score"])
        assert np.array_equal(r["xnodes"], xo) and np.array_equal(r["nodes"], no)
    untied = [i for i, q in enumerate(quants) if q is None]
    fast = engine.path([wins[i] for i in untied], trellis=False)
    assert engine.path_redone == 0  # continuous costs: the fast pass alone
    # dcp_hip_path_steps_packed: the same steps without the copy, a word each (state id | emission length << 16) --
    # of the fast pass here, of a mix of both passes (literal redo's among them) after the next call
    for j, r in enumerate(fast):
        pk = engine.path_steps_packed(j)
        assert np.array_equal(pk & 0xFFFF, r["state_ids"].astype(np.uint32)) and np.array_equal(pk >> 16, r["seqsizes"].astype(np.uint32))
    mixed = engine.path(wins, trellis=False)
    assert engine.path_redone > 0
    for j, r in enumerate(mixed):
        pk = engine.path_steps_packed(j)
        assert np.array_equal(pk & 0xFFFF, r["state_ids"].astype(np.uint32)) and np.array_equal(pk >> 16, r["seqsizes"].astype(np.uint32))
        assert np.array_equal(r["state_ids"], res[j]["state_ids"]) and np.array_equal(r["seqsizes"], res[j]["seqsizes"])


def test_path_pass_slices_by_table_memory(engine, orc, monkeypatch):
    """dcp_hip_path cuts a large request into slices whose DP tables fit the HBM budget;
    the slicing must not show in the results (forced here with a 1 MB budget)."""
    rng = np.random.default_rng(43)
    profs = [synth_profile(rng, K, None, 0.05) for K in (40, 173, 300, 700)]
    seqs = [random_seq(rng, int(rng.integers(150, 400))) for _ in range(12)]
    engine.clear_profiles()
    for p in profs:
        engine.add_profile(p.K, p.trans, p.match, p.null, p.bg)
    engine.commit()
    engine.set_sequences(seqs)
    engine.set_mode(True, False)
    wins = [(p, s, 0, len(seqs[s])) for s in range(len(seqs)) for p in range(len(profs))]
    whole = engine.path(wins, trellis=False)
    monkeypatch.setenv("DECIPHON_HIP_PATH_BUDGET_MB", "1")
    sliced = engine.path(wins, trellis=False)
    monkeypatch.delenv("DECIPHON_HIP_PATH_BUDGET_MB")
    for (p, s, _, _), a, b in zip(wins, whole, sliced):
        xt = orc.xtrans(max(len(seqs[s]) // 3, 1), True, False)
        score, xo, no = orc.path(profs[p], xt, seqs[s])
        ids, sizes = orc.unzip(profs[p].K, len(seqs[s]), xo, no)
        for r in (a, b):
            assert np.array_equal(r["state_ids"], ids) and np.array_equal(r["seqsizes"], sizes)
            assert bits(r["score"]) == bits(score)


def test_path_pass_does_not_depend_on_call_history(engine, orc):
    """The DP tables of the fast path pass live in an arena that is reused call after call:
    every table cell the traceback reads (row 0 included) must have been written by THIS
    call.  Tie-rich windows in changing subsets and orders over the same arena; a stale
    row 0 shows as a different (equal-cost) path."""
    rng = np.random.default_rng(47)
    profs = [synth_profile(rng, int(K), 2.0, 0.05) for K in (3, 193, 256, 33, 65, 17, 4, 128, 15, 100, 63, 31)]
    seqs = [random_seq(rng, int(rng.integers(5, 64))) for _ in profs]
    engine.clear_profiles()
    for p in profs:
        engine.add_profile(p.K, p.trans, p.match, p.null, p.bg)
    engine.commit()
    engine.set_sequences(seqs)
    engine.set_mode(True, True)
    smax = max(max(len(s) // 3, 1) for s in seqs)
    table = np.zeros((smax + 1, 13), np.float32)
    for s in range(1, smax + 1):
        table[s] = synth_xt(orc, 3 * s, 1, 1, 2.0)
    engine.set_xtrans_table(table)
    want = []
    for p, s in zip(profs, seqs):
        _, xo, no = orc.path(p, synth_xt(orc, len(s), 1, 1, 2.0), s)
        want.append(orc.unzip(p.K, len(s), xo, no))
    n = len(profs)
    orders = [list(range(n)), [8], list(range(n))[::-1], [7, 8], [8, 1], list(range(0, n, 2)), list(range(n))]
    for it in range(6):
        orders.append([int(i) for i in rng.permutation(n)[: int(rng.integers(1, n + 1))]])
    for order in orders:
        res = engine.path([(i, i, 0, len(seqs[i])) for i in order], trellis=False)
        for i, r in zip(order, res):
            assert np.array_equal(r["state_ids"], want[i][0]) and np.array_equal(r["seqsizes"], want[i][1]), (order, i)
    engine.set_xtrans_table(np.zeros((0, 13), np.float32))


def test_window_without_any_finite_path(engine, orc):
    """viterbi_cost = +inf (every way into the core is closed): the reference never walks such a
    trellis (c-core/thread.c:118-121 stops at the non-finite lrt); dcp_hip_path gives 0 steps and
    score +inf for it and is not disturbed in the windows around it."""
    rng = np.random.default_rng(67)
    closed = synth_profile(rng, 40, None, 0.0)
    closed.trans[0, :] = np.float32(np.inf)  # BM: no entry
    normal = synth_profile(rng, 40, None, 0.0)
    seqs = [random_seq(rng, 30), random_seq(rng, 12)]
    engine.clear_profiles()
    for p in (normal, closed):
        engine.add_profile(p.K, p.trans, p.match, p.null, p.bg)
    engine.commit()
    engine.set_sequences(seqs)
    engine.set_mode(True, False)
    wins = [(0, 0, 0, 30), (1, 0, 0, 30), (1, 1, 0, 12), (0, 1, 0, 12)]
    nul, alt = engine.cost(wins)
    assert np.isinf(alt[1]) and np.isinf(alt[2]) and np.isfinite(alt[0]) and np.isfinite(alt[3])
    res = engine.path(wins, trellis=False)
    for i, (pi, si, a, b) in enumerate(wins):
        xt = orc.xtrans(max(len(seqs[si]) // 3, 1), True, False)
        score, xo, no = orc.path((normal, closed)[pi], xt, seqs[si])
        assert bits(res[i]["score"]) == bits(score)
        if pi == 1:
            assert len(res[i]["state_ids"]) == 0 and np.isinf(res[i]["score"])
        else:
            ids, sizes = orc.unzip(40, len(seqs[si]), xo, no)
            assert np.array_equal(res[i]["state_ids"], ids) and np.array_equal(res[i]["seqsizes"], sizes)
