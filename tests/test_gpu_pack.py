"""GPU: the cost pass of short profiles, several windows per wavefront (deciphon_amd/csrc/viterbi_pack.h),
through the C ABI: against the oracle, and bit for bit against the one-window-per-wavefront kernels
(DECIPHON_HIP_PACK=0) on the same inputs -- every shape (groups of 4, 8, 16, 32 lanes), packs that are
full, partly filled and of mixed window lengths, tie-rich tables, and BASELINE configs[2] (massive.hmm is
a K = 3 profile, c-core/massive.hmm:2-5) at its full size: 1000 x 10 kb reads, 72 chained windows each."""
import numpy as np
import pytest

from dcp_testlib import bits, random_seq, synth_profile

pytestmark = pytest.mark.gpu

# (K, shape it lands in): capacities 3, 6, 12, 14, 28, 45, 60, 93, 124
KS = (1, 2, 3, 4, 6, 7, 12, 13, 14, 20, 28, 29, 45, 46, 60, 61, 64, 65, 93, 94, 100, 124)


def _both(engine, wins, monkeypatch):
    """packed (four-lane groups with their table in LDS) == packed with every table in global memory == one
    window per wavefront, bit for bit"""
    monkeypatch.delenv("DECIPHON_HIP_PACK", raising=False)
    monkeypatch.delenv("DECIPHON_HIP_PACK_LDS", raising=False)
    packed = engine.cost(wins)
    monkeypatch.setenv("DECIPHON_HIP_PACK_LDS", "0")
    nolds = engine.cost(wins)
    monkeypatch.delenv("DECIPHON_HIP_PACK_LDS")
    monkeypatch.setenv("DECIPHON_HIP_PACK", "0")
    plain = engine.cost(wins)
    monkeypatch.delenv("DECIPHON_HIP_PACK")
    for other in (nolds, plain):
        assert np.array_equal(packed[0].view(np.uint32), other[0].view(np.uint32))
        assert np.array_equal(packed[1].view(np.uint32), other[1].view(np.uint32))
    return packed


@pytest.mark.parametrize("quant", [None, 2.0])
def test_every_shape_against_oracle_and_plain_kernels(engine, orc, monkeypatch, quant):
    rng = np.random.default_rng(17 if quant else 16)
    profs = [synth_profile(rng, K, quant, [0.0, 0.05][i % 2]) for i, K in enumerate(KS)]
    profs.append(synth_profile(rng, 125, quant))  # just beyond the largest shape: stays on its own wavefront
    engine.clear_profiles()
    for p in profs:
        engine.add_profile(p.K, p.trans, p.match, p.null, p.bg)
    engine.commit()
    reads = [random_seq(rng, int(n)) for n in (1, 2, 5, 9, 33, 150, 151, 400, 37, 64, 90, 17)] + \
            [random_seq(rng, 120) for _ in range(21)]
    engine.set_sequences(reads)
    engine.set_mode(True, False)
    if quant:
        smax = max(len(r) // 3 for r in reads) + 1
        table = np.zeros((smax + 1, 13), np.float32)
        for s in range(1, smax + 1):
            table[s] = (np.round(orc.xtrans(s, True, False) / quant) * quant).astype(np.float32)
        engine.set_xtrans_table(table)
    wins = []
    for pi in range(len(profs)):
        for si, r in enumerate(reads):  # 33 windows per profile: no shape's G divides it, and lengths are mixed
            wins.append((pi, si, 0, len(r)))
        wins.append((pi, 7, 13, 391))  # a window inside a read
    try:
        nul, alt = _both(engine, wins, monkeypatch)
        for i, (pi, si, a, b) in enumerate(wins):
            seq = np.ascontiguousarray(reads[si][a:b])
            xt = orc.xtrans(max(len(seq) // 3, 1), True, False)
            if quant:
                xt = (np.round(xt / quant) * quant).astype(np.float32)
            assert bits(nul[i]) == bits(orc.null(profs[pi], xt, seq)), (profs[pi].K, wins[i])
            assert bits(alt[i]) == bits(orc.cost(profs[pi], xt, seq)), (profs[pi].K, wins[i])
    finally:
        engine.set_xtrans_table(np.zeros((0, 13), np.float32))


def test_massive_like_at_full_size(engine, orc, monkeypatch):
    """BASELINE configs[2]: a K = 3 profile x 1000 reads of 10 kb, every window of c-core/window.c
    (72 per read, 150 nt each): packed == plain on all 72 000 windows, a sample against the oracle."""
    from deciphon_amd import host

    rng = np.random.default_rng(8)
    prof = synth_profile(rng, 3)
    engine.clear_profiles()
    engine.add_profile(3, prof.trans, prof.match, prof.null, prof.bg)
    engine.commit()
    reads = [random_seq(rng, 10000) for _ in range(1000)]
    engine.set_sequences(reads)
    engine.set_mode(True, False)
    chain = []
    it = host.WindowIter(10000, 3)
    while (w := it.next()) is not None:
        chain.append((w[1], w[2]))
    assert len(chain) == 72
    wins = np.array([(0, s, a, b) for s in range(len(reads)) for a, b in chain], np.int32)
    nul, alt = _both(engine, wins, monkeypatch)
    for i in rng.choice(len(wins), size=300, replace=False):
        _, s, a, b = (int(v) for v in wins[i])
        seq = np.ascontiguousarray(reads[s][a:b])
        xt = orc.xtrans(max((b - a) // 3, 1), True, False)
        assert bits(nul[i]) == bits(orc.null(prof, xt, seq)) and bits(alt[i]) == bits(orc.cost(prof, xt, seq))


def test_many_reads_beyond_the_grid_y_limit(engine, orc):
    """70 000 short reads in one batch (the y extent of a grid stops at 65 535): codes and scores of reads on
    both sides of that index against the oracle."""
    rng = np.random.default_rng(9)
    prof = synth_profile(rng, 16)
    engine.clear_profiles()
    engine.add_profile(16, prof.trans, prof.match, prof.null, prof.bg)
    engine.commit()
    reads = [random_seq(rng, 30) for _ in range(70000)]
    engine.set_sequences(reads)
    engine.set_mode(True, False)
    pick = [0, 1, 65534, 65535, 65536, 65537, 69999] + [int(v) for v in rng.integers(0, 70000, size=40)]
    nul, alt = engine.cost([(0, s, 0, 30) for s in pick])
    xt = orc.xtrans(10, True, False)
    for i, s in enumerate(pick):
        assert bits(nul[i]) == bits(orc.null(prof, xt, reads[s])) and bits(alt[i]) == bits(orc.cost(prof, xt, reads[s])), s


def test_lrt_filter_on_the_device(engine, orc):
    """dcp_hip_cost_hits: process_window's filter (c-core/thread.c:118-121) as a kernel -- the same windows, in
    the same order, with the same lrt bits as filtering dcp_hip_cost's scores on the host (c-core/lrt.h:6-9)."""
    import os

    import deciphon_amd
    from deciphon_amd import host, synth
    from dcp_testlib import GOLDEN

    engine.clear_profiles()
    engine.load_dcp(os.path.join(GOLDEN, "minifam.dcp"))
    rng = np.random.default_rng(21)
    dead = synth_profile(rng, 40)
    dead.trans[0, :] = np.float32(np.inf)  # no way into the core: viterbi_cost = +inf, lrt = -inf
    engine.add_profile(dead.K, dead.trans, dead.match, dead.null, dead.bg)
    engine.commit()
    seeds = synth.load_seeds(os.path.join(GOLDEN, "minifam.dcp"))
    reads = synth.synth_reads(300, 1500, [s["consensus"] for s in seeds], 77, planted_every=3)
    engine.set_sequences(reads)
    engine.set_mode(True, False)
    wins = np.array([(p, s, 0, 1500) for p in range(4) for s in range(len(reads))], np.int32)
    nul, alt = engine.cost(wins)
    lrt = np.array([host.lrt(-a, -b) for a, b in zip(nul, alt)], np.float32)
    keep = np.nonzero(np.isfinite(lrt) & (lrt >= 0))[0]
    idx, got = engine.cost_hits(wins)
    assert 50 <= len(keep) < len(wins) // 2 and not np.isfinite(lrt[3 * len(reads):]).any()
    assert np.array_equal(idx, keep.astype(np.int32))
    assert np.array_equal(got.view(np.uint32), lrt[keep].view(np.uint32))
    assert engine.cost_hits(wins[:0])[0].size == 0


@pytest.mark.parametrize("mode", ["auto", "one launch per class", "three blocks side by side", "a launch per block"])
@pytest.mark.parametrize("rows", ["50", "500", "0"])
def test_fast_path_pass_in_blocks(engine, orc, monkeypatch, rows, mode):
    """The fast path pass holds a window's DP table a block at a time (checkpoints of the folded ring every
    B rows, blocks recomputed from the last to the first, the traceback resumed from block to block): the
    same steps as the oracle's trellis_unzip with B = 50 (dozens of blocks), the default 500 and 0 (whole
    tables), for every single-wave and multi-wave class, on windows with planted error-bearing domains --
    in each of the forms the engine chooses between by the HBM it has: as many blocks of a window side by side as the
    table arena holds (auto: all of them here), one workgroup walking its window's blocks in one launch
    (dcp_path_blocks_kernel), groups of three blocks, and a launch per block and phase."""
    import os

    from dcp_testlib import GOLDEN
    from deciphon_amd import synth
    from oracle.dcp_reader import Protein

    monkeypatch.setenv("DECIPHON_HIP_CKPT_ROWS", rows)
    if mode == "one launch per class":
        monkeypatch.setenv("DECIPHON_HIP_PATH_GROUP", "1")
    elif mode == "three blocks side by side":
        monkeypatch.setenv("DECIPHON_HIP_PATH_GROUP", "3")
    elif mode == "a launch per block":
        monkeypatch.setenv("DECIPHON_HIP_PATH_GROUP", "1")
        monkeypatch.setenv("DECIPHON_HIP_PATH_FUSED", "0")
    seeds = synth.load_seeds(os.path.join(GOLDEN, "minifam.dcp"))
    Ks = (40, 100, 173, 250, 300, 500, 700, 1000, 1500)
    prots = [synth.tile_protein(seeds, K, 11 * i, f"T{K}") for i, K in enumerate(Ks)]
    engine.clear_profiles()
    for p in prots:
        engine.add_protein(p["core_size"], p["trans"], p["emission"], p["BMk"], p["null_emission"], p["bg_emission"])
    engine.commit()
    rng = np.random.default_rng(int(rows) + 3)
    reads = []
    for p in prots:
        x = rng.integers(0, 4, size=1700).astype(np.uint8)
        for at in (100, 900):
            a = int(rng.integers(0, max(len(p["consensus"]) - 150, 1)))
            dom = synth.mutate(synth.back_translate(p["consensus"][a : a + 150]), rng, 0.06, 0.02, 0.02)
            x[at : at + len(dom)] = dom
        reads.append(x)
    engine.set_sequences(reads)
    engine.set_mode(True, False)
    wins = [(i, i, 0, 1700) for i in range(len(prots))] + [(i, i, 37, 37 + 555) for i in range(len(prots))]
    res = engine.path(wins, trellis=False)
    assert engine.path_redone == 0
    for (pi, si, a, b), r in zip(wins, res):
        p = prots[pi]
        prof = orc.setup_profile(Protein(p["accession"], 1, p["consensus"], p["core_size"], p["null_emission"],
                                         p["bg_emission"], p["trans"], p["emission"], p["BMk"]))
        seq = np.ascontiguousarray(reads[si][a:b])
        xt = orc.xtrans(max(len(seq) // 3, 1), True, False)
        score, xo, no = orc.path(prof, xt, seq)
        ids, sizes = orc.unzip(prof.K, len(seq), xo, no)
        assert bits(r["score"]) == bits(score), (rows, p["core_size"], a, b)
        assert np.array_equal(r["state_ids"], ids) and np.array_equal(r["seqsizes"], sizes), (rows, p["core_size"], a, b)


def test_two_cost_batches_in_flight_and_a_path_pass_between(orc):
    """dcp_hip_cost_hits_begin / _end: two batches outstanding at once (the second queued behind the first on the same
    kernel streams, its window lists in buffers of its own), a path pass between a _begin and its _end (it has buffers
    and streams of its own too), every other call refused meanwhile -- and the hits of each batch equal to those of the
    one-call form.  On a FRESH engine and process state: the first use of every buffer set."""
    import time

    import deciphon_amd
    from dcp_testlib import random_seq, synth_profile

    rng = np.random.default_rng(99)
    profs = [synth_profile(rng, K, None, 0.02) for K in (12, 60, 173, 300, 640)]
    reads = [random_seq(rng, 1500) for _ in range(400)]
    for r in reads[::7]:  # domains that hit: the consensus-free way -- copy a stretch the model likes
        r[100:400] = reads[0][100:400]
    with deciphon_amd.Engine(0) as eng:
        for p in profs:
            eng.add_profile(p.K, p.trans, p.match, p.null, p.bg)
        eng.commit()
        eng.set_sequences(reads)
        eng.set_mode(True, False)
        small = np.array([(p, s, 0, 1500) for p in range(2) for s in range(40)], np.int32)
        big = np.array([(p, s, a, a + 1200) for p in range(5) for s in range(400) for a in (0, 150, 300)], np.int32)
        eng.cost_hits_begin(small)
        eng.cost_hits_begin(big)  # while the first is in flight
        with pytest.raises(deciphon_amd.HipError) as e:
            eng.cost_hits_begin(small)  # a third: refused
        assert e.value.code == 8
        with pytest.raises(deciphon_amd.HipError):
            eng.set_sequences(reads)  # anything that would pull the data from under the kernels: refused
        paths = eng.path([tuple(int(v) for v in small[3])], trellis=False)  # allowed: its own buffers and streams
        t0 = time.perf_counter()
        got_small = eng.cost_hits_end()
        got_big = eng.cost_hits_end()
        waited = time.perf_counter() - t0
        with pytest.raises(deciphon_amd.HipError):
            eng.cost_hits_end()  # nothing outstanding
        want_small, want_big = eng.cost_hits(small), eng.cost_hits(big)
        for got, want in ((got_small, want_small), (got_big, want_big)):
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
        nul, alt = eng.cost(big)
        lrt = (-2.0 * ((-nul) - (-alt))).astype(np.float32)
        keep = np.nonzero(np.isfinite(lrt) & (lrt >= 0))[0]
        assert np.array_equal(got_big[0], keep.astype(np.int32)) and len(keep) > 0
        xt = orc.xtrans(1500 // 3, True, False)
        p, s = int(small[3][0]), int(small[3][1])
        assert bits(paths[0]["score"]) == bits(orc.cost(profs[p], xt, reads[s]))
        print(f"two batches ({len(small)} + {len(big)} windows): ends waited {waited * 1e3:.1f} ms")
