"""CPU: the oracle restatement (oracle/dcp_oracle.c) against every golden the reference
holds for this path, and -- where oracle/_ref exists -- against the reference's own
viterbi.c run live.  This is what pins the oracle."""
import os
import sys
import zlib

import numpy as np
import pytest

from dcp_testlib import GOLDEN, bits, random_seq, read_fasta, reflib, synth_profile

sys.path.insert(0, GOLDEN)
from make_golden import MODES, synth_case_params, synth_xt  # noqa: E402
from oracle.dcp_reader import read_dcp  # noqa: E402


@pytest.fixture(scope="module")
def minifam(orc):
    db = read_dcp(os.path.join(GOLDEN, "minifam.dcp"))
    return db, [orc.setup_profile(p) for p in db.proteins]


@pytest.fixture(scope="module")
def reads(orc):
    r = read_fasta(os.path.join(GOLDEN, "consensus.fna")) + read_fasta(os.path.join(GOLDEN, "consensus_multi.fna"))
    return [(n, s, orc.encode(s)) for n, s in r]


def test_fixture_facts(minifam):
    """header and protein facts of control/tests/files/minifam.dcp."""
    db, _ = minifam
    assert db.header["magic_number"] == 0xC6F1 and db.header["version"] == 1 and db.header["entry_dist"] == 2
    assert abs(db.header["epsilon"] - 0.01) < 1e-9 and db.header["has_ga"] is True
    assert db.protein_sizes == [1063919, 1474979, 997423]
    assert [p.core_size for p in db.proteins] == [173, 241, 162]
    assert [p.accession for p in db.proteins] == ["PF00742.20", "PF00696.29", "PF16620.6"]


def test_code_indexing(orc):
    """imm_eseq_get: off[len] + sum idx*4^(len-1-i)."""
    seq = orc.encode("ACGTTGCA")
    assert orc.code(seq, 0, 1) == 0 and orc.code(seq, 3, 1) == 3
    assert orc.code(seq, 0, 2) == 4 + 0 * 4 + 1
    assert orc.code(seq, 1, 3) == 20 + 1 * 16 + 2 * 4 + 3
    assert orc.code(seq, 0, 5) == 340 + ((((0 * 4 + 1) * 4 + 2) * 4 + 3) * 4 + 3)
    assert orc.code(orc.encode("TTTTT"), 0, 5) == 1363


def test_survey_probe_values(orc, minifam, reads):
    """The values the survey measured with the reference's unmodified viterbi.c."""
    _, profs = minifam
    want = {(0, 0): (-691.526245, -545.702820, 291.6469, 177), (1, 1): (-975.136719, -800.470520, 349.3324, 245),
            (2, 2): (-667.955017, -487.766724, 360.3766, 166)}
    for (pi, ri), (nul, alt, lrt, nsteps) in want.items():
        x = reads[ri][2]
        xt = orc.xtrans(max(len(x) // 3, 1), True, False)
        n, c = orc.null(profs[pi], xt, x), orc.cost(profs[pi], xt, x)
        assert abs(-n - nul) < 1e-4 and abs(-c - alt) < 1e-4 and abs(orc.lrt(-n, -c) - lrt) < 1e-3
        _, xn, nd = orc.path(profs[pi], xt, x)
        ids, _sizes = orc.unzip(profs[pi].K, len(x), xn, nd)
        assert len(ids) == nsteps
    multi = {3: (584.4653, 286.2673, 356), 4: (585.3613, 286.7153, 352), 5: (552.3955, 283.0732, 357),
             6: (491.9983, 283.1218, 359), 7: (535.7029, 284.8184, 360)}
    for ri, (lrt_mh, lrt_sh, nsteps) in multi.items():
        x = reads[ri][2]
        for mh, want_lrt in ((True, lrt_mh), (False, lrt_sh)):
            xt = orc.xtrans(max(len(x) // 3, 1), mh, False)
            lrt = orc.lrt(-orc.null(profs[0], xt, x), -orc.cost(profs[0], xt, x))
            assert abs(lrt - want_lrt) < 2e-3, (ri, mh, lrt)
        xt = orc.xtrans(max(len(x) // 3, 1), True, False)
        _, xn, nd = orc.path(profs[0], xt, x)
        ids, _ = orc.unzip(profs[0].K, len(x), xn, nd)
        assert len(ids) == nsteps


def test_goldens_from_reference_viterbi(orc, minifam, reads):
    """Every score bit, trellis CRC and path of tests/golden/minifam_consensus.npz
    (written by the reference's own c-core/viterbi.c, oracle/_ref)."""
    _, profs = minifam
    g = np.load(os.path.join(GOLDEN, "minifam_consensus.npz"))
    assert len(g["profile"]) == 3 * 8 * len(MODES)
    for j in range(len(g["profile"])):
        prof, x = profs[int(g["profile"][j])], reads[int(g["read"][j])][2]
        xt = orc.xtrans(max(len(x) // 3, 1), bool(g["multi_hits"][j]), bool(g["hmmer3_compat"][j]))
        assert bits(orc.null(prof, xt, x)) == int(g["null_bits"][j])
        assert bits(orc.cost(prof, xt, x)) == int(g["alt_bits"][j])
        if np.isfinite(g["lrt"][j]) and g["lrt"][j] >= 0:
            score, xn, nd = orc.path(prof, xt, x)
            assert bits(score) == int(g["alt_bits"][j])
            assert zlib.crc32(xn.tobytes()) == int(g["xnodes_crc"][j])
            assert zlib.crc32(nd.tobytes()) == int(g["nodes_crc"][j])
            ids, sizes = orc.unzip(prof.K, len(x), xn, nd)
            a, b = int(g["path_off"][j]), int(g["path_off"][j + 1])
            assert np.array_equal(ids, g["path_ids"][a:b]) and np.array_equal(sizes, g["path_sizes"][a:b])


def test_synthetic_tie_goldens(orc):
    g = np.load(os.path.join(GOLDEN, "synth_ties.npz"))
    rng = np.random.default_rng(int(g["seed"]))
    for it in range(int(g["ncase"])):
        K, L, quant, pinf, mh, h3 = synth_case_params(rng, it)
        prof = synth_profile(rng, K, quant, pinf)
        seq = random_seq(rng, L)
        xt = synth_xt(orc, L, mh, h3, quant)
        assert bits(orc.null(prof, xt, seq)) == int(g["null_bits"][it])
        assert bits(orc.cost(prof, xt, seq)) == int(g["alt_bits"][it])
        _, xn, nd = orc.path(prof, xt, seq)
        assert zlib.crc32(xn.tobytes()) == int(g["xnodes_crc"][it]), (it, K, L)
        assert zlib.crc32(nd.tobytes()) == int(g["nodes_crc"][it]), (it, K, L)


def test_long_profile_goldens(orc):
    """tests/large_cases.py: K = 257 .. 16383 (every multi-wave kernel class and the strip class at their
    boundary sizes), windows up to 10 kb, continuous and tie-rich tables.  The expected bits were produced
    by the reference's own viterbi.c (oracle/_ref via make_golden.py); the restatement must reproduce
    scores, every trellis word (CRC32 of the whole trellis) and the unzipped path.  The path itself is
    re-priced step by step (transitions and emissions it names, in double): it must be a legal path of the
    model whose total is the Viterbi score -- a check of trellis_unzip's output (state ids, emission
    sizes, N/J/C/I/D steps included) that does not go through any back-pointer."""
    from large_cases import build_case, large_cases, path_cost

    g = np.load(os.path.join(GOLDEN, "large_classes.npz"))
    cases = large_cases()
    assert len(cases) == len(g["K"])

    def one(c):
        i = c["idx"]
        assert (c["K"], c["L"]) == (int(g["K"][i]), int(g["L"][i]))
        prof, seq, xt = build_case(c, orc)
        assert bits(orc.null(prof, xt, seq)) == int(g["null_bits"][i]), i
        alt, xn, nd = orc.path(prof, xt, seq)  # the same DP as orc.cost, pointers kept
        assert bits(alt) == int(g["alt_bits"][i]), (i, c)
        assert zlib.crc32(xn.tobytes()) == int(g["xnodes_crc"][i]), (i, c)
        assert zlib.crc32(nd.tobytes()) == int(g["nodes_crc"][i]), (i, c)
        a, b = int(g["path_off"][i]), int(g["path_off"][i + 1])
        if not np.isfinite(alt):
            assert a == b
            return set()
        ids, sizes = orc.unzip(prof.K, len(seq), xn, nd)
        assert np.array_equal(ids, g["path_ids"][a:b]) and np.array_equal(sizes, g["path_sizes"][a:b]), i
        total = path_cost(orc, prof, xt, seq, ids, sizes)
        assert abs(total - float(alt)) <= 1e-4 * max(abs(float(alt)), 1.0), (i, total, float(alt))
        return {int(s) >> 14 if int(s) >> 14 < 3 else int(s) & 0x3FFF for s in ids}

    # the cases are independent and the oracle is C behind ctypes (the GIL is released): a few at a time
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(min(os.cpu_count() or 1, 6)) as ex:
        kinds = set().union(*ex.map(one, cases))
    assert kinds >= {0, 1, 2, 3, 4, 5, 6, 7, 8, 9}  # M, I, D and S, N, B, E, J, C, T all occur on the paths


def test_window_cap_golden(orc):
    """c-core/window.c:13's longest window, 100 000 rows, at K = 2048: the restatement against the reference
    viterbi.c's bits (tests/golden/window_cap.npz) -- scores, CRC32 of the 410 MB trellis, the unzipped path, the
    path re-priced.  The strip-class case of the same file (K = 4200, twice the cells) reproduces too (checked when
    the golden was made: 85 s of this scalar code); it is left to the -m gpu tests, which compare the HIP path
    with the reference bits directly."""
    from large_cases import build_case, path_cost, window_cap_cases

    g = np.load(os.path.join(GOLDEN, "window_cap.npz"))
    c = window_cap_cases()[0]
    assert (c["K"], c["L"]) == (int(g["K"][0]), int(g["L"][0])) == (2048, 100000)
    prof, seq, xt = build_case(c, orc)
    assert bits(orc.null(prof, xt, seq)) == int(g["null_bits"][0])
    alt, xn, nd = orc.path(prof, xt, seq)
    assert bits(alt) == int(g["alt_bits"][0])
    assert zlib.crc32(xn.tobytes()) == int(g["xnodes_crc"][0]) and zlib.crc32(nd.tobytes()) == int(g["nodes_crc"][0])
    ids, sizes = orc.unzip(prof.K, len(seq), xn, nd)
    a, b = int(g["path_off"][0]), int(g["path_off"][1])
    assert np.array_equal(ids, g["path_ids"][a:b]) and np.array_equal(sizes, g["path_sizes"][a:b])
    total = path_cost(orc, prof, xt, seq, ids, sizes)
    assert abs(total - float(alt)) <= 1e-4 * abs(float(alt))


def test_products_tsv_golden(orc, minifam, reads):
    """The reference's committed scan result (control/tests/files/snap.dcs): windows, hit
    spans, lrt as printed, and the (subsequence, state) pairs of the match column."""
    _, profs = minifam
    acc = {p.accession: i for i, p in enumerate(profs)}
    rows = [ln.rstrip("\n").split("\t") for ln in open(os.path.join(GOLDEN, "products.tsv"))]
    assert rows[0] == ["sequence", "window", "window_start", "window_stop", "hit", "hit_start", "hit_stop",
                       "profile", "abc", "lrt", "evalue", "match"]
    for row in rows[1:]:
        si, pi = int(row[0]), acc[row[7]]
        name, text, x = reads[si]
        wins = orc.windows(len(x), profs[pi].K)
        assert wins == [(int(row[1]), int(row[2]), int(row[3]))]
        xt = orc.xtrans(max(len(x) // 3, 1), True, False)
        lrt = orc.lrt(-orc.null(profs[pi], xt, x), -orc.cost(profs[pi], xt, x))
        assert f"{lrt:.1f}" == row[9]
        _, xn, nd = orc.path(profs[pi], xt, x)
        ids, sizes = orc.unzip(profs[pi].K, len(x), xn, nd)
        hit, last = orc.hits(ids, sizes)
        assert (hit[0], hit[1]) == (int(row[5]), int(row[6])) and last == hit[1] - 1
        pos, got = hit[0], []
        for st, sz in zip(ids[hit[2] : hit[3]], sizes[hit[2] : hit[3]]):
            got.append((text[pos : pos + sz], orc.state_name(st)))
            pos += sz
        assert got == [tuple(m.split(",")[:2]) for m in row[11].split(";")]


def test_live_against_reference_viterbi(orc):
    """Where oracle/_ref is present: fresh random tie-rich cases against the reference itself."""
    ref = reflib()
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(4242)
    for it in range(330):
        # the last cases: sizes of every multi-wave kernel class and of the strip class
        K = int(rng.choice([2, 3, 5, 8, 9, 17, 33, 64, 100, 173])) if it < 300 else \
            int(rng.choice([257, 385, 600, 1000, 1537, 2049, 3000, 4097, 6000, 9000]))
        quant = [None, 1.0, 4.0][it % 3]
        prof = synth_profile(rng, K, quant, [0.0, 0.2][it % 2])
        seq = random_seq(rng, int(rng.integers(1, 80)))
        xt = synth_xt(orc, len(seq), it % 2, (it // 2) % 2, quant)
        ref.setup(prof)
        assert bits(orc.null(prof, xt, seq)) == bits(ref.null(xt, seq))
        assert bits(orc.cost(prof, xt, seq)) == bits(ref.cost(xt, seq))
        _, xn, nd = orc.path(prof, xt, seq)
        rx, rn = ref.path(xt, seq)
        assert np.array_equal(xn, rx) and np.array_equal(nd, rn), (it, K)


def test_reference_row0_leak_is_off_path(orc, minifam, reads):
    """The reference never clears DP row 0 between runs (c-core/viterbi.c:471-473 only sets S
    and B), so viterbi_path after viterbi_cost on the same struct sees the previous run's last
    row as row 0.  On the goldens that changes trellis words off the optimal path only: scores
    and unzipped paths equal the history-free run this repository defines parity on."""
    ref = reflib()
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    _, profs = minifam
    differs = 0
    for pi, ri in ((0, 0), (1, 1), (2, 2), (0, 3), (0, 7)):
        prof, x = profs[pi], reads[ri][2]
        xt = orc.xtrans(max(len(x) // 3, 1), True, False)
        ref.setup(prof)
        c_fresh = ref.cost(xt, x, fresh=True)
        xn1, nd1 = ref.path(xt, x, fresh=False)  # the reference's real call sequence
        xn0, nd0 = ref.path(xt, x, fresh=True)
        differs += int(not (np.array_equal(xn0, xn1) and np.array_equal(nd0, nd1)))
        assert bits(c_fresh) == bits(orc.cost(prof, xt, x))
        a, b = orc.unzip(prof.K, len(x), xn0, nd0), orc.unzip(prof.K, len(x), xn1, nd1)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert differs > 0  # the leak is real


def test_windows_and_partitions(orc):
    # c-core/window.c: K=173, 10 kb read -> [0,8650), [7959,10000)  
    assert orc.windows(10000, 173) == [(0, 0, 8650), (1, 7959, 10000)]
    # a hit reported at window-relative position 8000 moves the next start past it
    assert orc.windows(10000, 173, lambda idx: 8000 if idx == 0 else None)[1] == (1, 8001, 10000)
    assert orc.windows(100, 3) == [(0, 0, 100)] or len(orc.windows(100, 3)) >= 1
    w = orc.windows(1000, 3)
    assert w[0] == (0, 0, 150) and w[1] == (1, 139, 289) and w[-1][2] == 1000
    # c-core/partition_size.c: ceil((N - i) / k)
    assert [orc.partition_size(10, 3, i) for i in range(3)] == [4, 3, 3]
    assert sum(orc.partition_size(20000, 8, i) for i in range(8)) == 20000


def test_encode_and_disambiguate(orc):
    assert list(orc.encode("acgtACGT")) == [0, 1, 2, 3, 0, 1, 2, 3]
    assert list(orc.encode("ACGU")) == [0, 1, 2, 3]
    # R = A|G -> the more frequent one in this sequence; ties keep the first listed
    assert list(orc.encode("GGAR")) == [2, 2, 0, 2]
    assert list(orc.encode("AGR")) == [0, 2, 0]
    assert list(orc.encode("CCTN")) == [1, 1, 3, 1]
    with pytest.raises(ValueError):
        orc.encode("ACGTU")  # DCP_ENUCLTSEQTU
    with pytest.raises(ValueError):
        orc.encode("ACG-T")  # DCP_ESEQABC
