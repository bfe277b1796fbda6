"""CPU: the kernel logic itself (deciphon_amd/csrc/viterbi_body.h, unchanged) instantiated
on a lock-step 64-lane emulator (tests/emul/) and compared bit for bit with the oracle.
This checks striping, the folded recurrences of the cost pass, the lazy D->D carries and
the E tie rule without a GPU; the DPP/readlane lowering itself is only tested with -m gpu."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from dcp_testlib import GOLDEN, ROOT, bits, choose_qw, code_rows, pack_profile, random_seq, read_fasta, synth_profile
from oracle.dcp_reader import read_dcp


@pytest.fixture(scope="module")
def em():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emul")], check=True)
    return C.CDLL(os.path.join(ROOT, "tests", "emul", "libdcp_emul.so"))


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def run_cost(em, prof, xt, seq):
    pool, pd = pack_profile(prof)
    rows = code_rows(seq)
    xt16 = np.zeros(16, np.float32)
    xt16[:13] = xt
    out = np.zeros(2, np.float32)
    assert em.emul_cost(_vp(pool), C.byref(pd), _vp(rows), len(seq), _vp(xt16), _vp(out)) == 0
    return out


def run_path(em, prof, xt, seq):
    pool, pd = pack_profile(prof, *choose_qw(prof.K, path=True))
    rows = code_rows(seq)
    xt16 = np.zeros(16, np.float32)
    xt16[:13] = xt
    L = len(seq)
    xn = np.zeros(L + 1, np.uint32)
    nd = np.zeros((L + 1) * prof.K, np.uint16)
    sc = C.c_float(0)
    assert em.emul_path(_vp(pool), C.byref(pd), _vp(rows), L, _vp(xt16), _vp(xn), _vp(nd), C.byref(sc)) == 0
    return np.float32(sc.value), xn, nd


def test_code_rows_helper(orc):
    seq = random_seq(np.random.default_rng(0), 40)
    rows = code_rows(seq)
    for r in range(1, 41):
        for t in range(1, 6):
            if r - t >= 0:
                assert rows[r, t - 1] == orc.code(seq, r - t, t)


def test_tie_rich_random_cases(em, orc):
    rng = np.random.default_rng(7)
    for it in range(400):
        K = int(rng.choice([2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 33, 63, 64, 65, 100, 128, 129, 173, 192, 193, 241, 256]))
        quant = [None, 0.5, 1.0, 2.0, 4.0, 8.0][it % 6]
        prof = synth_profile(rng, K, quant, [0, 0.05, 0.3][it % 3])
        seq = random_seq(rng, int(rng.integers(1, 40)))
        xt = orc.xtrans(max(len(seq) // 3, 1), it % 2, (it // 2) % 2)
        if quant:
            xt = (np.round(xt / quant) * quant).astype(np.float32)
        out = run_cost(em, prof, xt, seq)
        assert bits(out[0]) == bits(orc.null(prof, xt, seq)), (it, K)
        assert bits(out[1]) == bits(orc.cost(prof, xt, seq)), (it, K)
        score, xn, nd = run_path(em, prof, xt, seq)
        s_o, xo, no = orc.path(prof, xt, seq)
        assert bits(score) == bits(s_o)
        assert np.array_equal(xn, xo) and np.array_equal(nd, no), (it, K, quant)


def test_one_position_per_lane_less_on_the_class_layout(em, orc):
    """dcp_launch_cost_narrow: K <= 320 / 448 / 640 run as (5,1) / (7,1) / (10,1) on the tables padded for
    (6,1) / (8,1) / (6,2) -- the same bits as the oracle (and therefore as the class's own shape)."""
    rng = np.random.default_rng(21)
    cases = [(257, 6, 1, 5, 1), (300, 6, 1, 5, 1), (320, 6, 1, 5, 1), (385, 8, 1, 7, 1), (448, 8, 1, 7, 1),
             # one wavefront of 10 positions per lane on the two-wave layout of 768 columns
             (513, 6, 2, 10, 1), (601, 6, 2, 10, 1), (640, 6, 2, 10, 1)]
    for it, (K, Q, W, QA, WA) in enumerate(cases * 2):
        quant = [None, 2.0][it % 2]
        prof = synth_profile(rng, K, quant, [0, 0.05][it % 2])
        if it >= len(cases):  # cheap delete runs: the lazy loop carries across lanes (and waves)
            prof.trans[7, 1:] = np.float32(0.01)
            prof.trans[3, 1:] = np.float32(0.02)
        seq = random_seq(rng, int(rng.integers(20, 70)))
        xt = orc.xtrans(max(len(seq) // 3, 1), True, False)
        if quant:
            xt = (np.round(xt / quant) * quant).astype(np.float32)
        pool, pd = pack_profile(prof, Q, W)
        pd.Q = QA
        pd.W = WA
        rows = code_rows(seq)
        xt16 = np.zeros(16, np.float32)
        xt16[:13] = xt
        out = np.zeros(2, np.float32)
        assert em.emul_cost(_vp(pool), C.byref(pd), _vp(rows), len(seq), _vp(xt16), _vp(out)) == 0
        assert bits(out[0]) == bits(orc.null(prof, xt, seq)), (K, QA)
        assert bits(out[1]) == bits(orc.cost(prof, xt, seq)), (K, QA)


def test_long_delete_runs_cross_many_lanes(em, orc):
    """Cheap D->D and expensive everything else: delete runs span dozens of lanes, so the
    lazy carry loop must iterate many times (worst case of c-core/viterbi.c:569-580)."""
    rng = np.random.default_rng(3)
    for K in (64, 130, 256):
        prof = synth_profile(rng, K)
        prof.trans[7, 1:] = np.float32(0.01)  # DD
        prof.trans[3, 1:] = np.float32(0.02)  # MD
        prof.trans[1, 1:] = np.float32(9.0)   # MM
        prof.match[:, K // 2:] += np.float32(30.0)
        seq = random_seq(rng, 25)
        xt = orc.xtrans(8, True, False)
        out = run_cost(em, prof, xt, seq)
        assert bits(out[1]) == bits(orc.cost(prof, xt, seq))
        score, xn, nd = run_path(em, prof, xt, seq)
        s_o, xo, no = orc.path(prof, xt, seq)
        assert bits(score) == bits(s_o) and np.array_equal(xn, xo) and np.array_equal(nd, no)
        assert ((no >> 5) & 1).sum() > K  # plenty of D<-D pointers were taken


def test_long_profiles_as_multi_wave_groups(em, orc):
    """K > 256: W = 2..16 wavefronts per problem, emulated as one 64*W-lane vector.  Checks the
    group-level logic (what is exchanged, in which order); the LDS/barrier lowering is GPU-only."""
    rng = np.random.default_rng(17)
    fallback, rows = [0, 0], [0, 0]
    for it in range(40):
        K = int(rng.choice([257, 300, 511, 512, 513, 700, 1024, 1025, 1500, 2048, 2049, 3000, 4096]))
        quant = [None, 1.0, 4.0][it % 3]
        prof = synth_profile(rng, K, quant, [0, 0.05][it % 2])
        if it % 5 == 0:  # long delete runs that cross wave boundaries
            prof.trans[7, 1:] = np.float32(0.01)
            prof.trans[3, 1:] = np.float32(0.02)
            prof.trans[1, 1:] = np.float32(9.0)
            prof.match[:, K // 3:] += np.float32(30.0)
        seq = random_seq(rng, int(rng.integers(1, 12)))
        xt = orc.xtrans(max(len(seq) // 3, 1), it % 2, 0)
        if quant:
            xt = (np.round(xt / quant) * quant).astype(np.float32)
        em.emul_fallback_rows.restype = C.c_long
        em.emul_fallback_rows()
        out = run_cost(em, prof, xt, seq)
        assert bits(out[0]) == bits(orc.null(prof, xt, seq)) and bits(out[1]) == bits(orc.cost(prof, xt, seq)), (it, K)
        # rows whose waves exchange boundary values once (the published D of every wave provably
        # final) vs rows that fall back to exchanging until stable: nearly free delete runs force
        # the fallback, ordinary tables never need it -- both must give the reference's bits
        fallback[it % 5 == 0] += em.emul_fallback_rows()
        rows[it % 5 == 0] += len(seq)
        score, xn, nd = run_path(em, prof, xt, seq)
        s_o, xo, no = orc.path(prof, xt, seq)
        assert bits(score) == bits(s_o) and np.array_equal(xn, xo) and np.array_equal(nd, no), (it, K)
    assert fallback[0] == 0 and 0 < fallback[1] <= rows[1], (fallback, rows)


def test_minifam_consensus_pairs(em, orc):
    db = read_dcp(os.path.join(GOLDEN, "minifam.dcp"))
    reads = [orc.encode(s) for _, s in read_fasta(os.path.join(GOLDEN, "consensus.fna"))]
    for pi, ri in ((0, 0), (1, 1), (2, 2), (0, 1)):
        prof = orc.setup_profile(db.proteins[pi])
        seq = reads[ri]
        xt = orc.xtrans(max(len(seq) // 3, 1), True, False)
        out = run_cost(em, prof, xt, seq)
        assert bits(out[0]) == bits(orc.null(prof, xt, seq)) and bits(out[1]) == bits(orc.cost(prof, xt, seq))
    prof, seq = orc.setup_profile(db.proteins[2]), reads[2]
    xt = orc.xtrans(max(len(seq) // 3, 1), True, False)
    score, xn, nd = run_path(em, prof, xt, seq)
    s_o, xo, no = orc.path(prof, xt, seq)
    assert bits(score) == bits(s_o) and np.array_equal(xn, xo) and np.array_equal(nd, no)


def test_strip_mined_profiles(em, orc):
    """StripWave: a profile wider than one workgroup is walked strip by strip with the ring of
    folded rows in memory, B applied when a row is used, and the previous strip standing in
    front of wave 0.  Small strips here (64..512 positions) so that every boundary case is
    hit in a few rows; scores and the traceback over the stored table against the oracle."""
    rng = np.random.default_rng(23)
    em.emul_fallback_rows.restype = C.c_long
    em.emul_traceback.restype = C.c_int
    shapes = [(1, 1, 3), (2, 1, 2), (1, 2, 3), (2, 2, 2), (4, 2, 2), (1, 4, 2), (1, 1, 5), (2, 2, 3)]
    nfallback = 0
    for it in range(64):
        Q, W, S = shapes[it % len(shapes)]
        KS = 64 * Q * W
        K = int(rng.integers(KS * (S - 1) + 1, KS * S + 1))
        if it % 7 == 0:
            K = KS * S  # no padding at all
        if it % 11 == 0:
            K = KS * (S - 1) + 1  # one position in the last strip
        quant = [None, 1.0, None, 4.0][it % 4]
        prof = synth_profile(rng, K, quant, [0, 0.05][it % 2])
        if it % 3 == 0:  # delete runs that cross waves and strips
            prof.trans[7, 1:] = np.float32(0.01)
            prof.trans[3, 1:] = np.float32(0.02)
            prof.trans[1, 1:] = np.float32(9.0)
            prof.match[:, K // 3:] += np.float32(30.0)
        seq = random_seq(rng, int(rng.integers(1, 24)))
        L = len(seq)
        xt = orc.xtrans(max(L // 3, 1), it % 2, (it // 2) % 2)
        if quant:
            xt = (np.round(xt / quant) * quant).astype(np.float32)
        pool, pd = pack_profile(prof, Q, W, S)
        rows = code_rows(seq)
        xt16 = np.zeros(16, np.float32)
        xt16[:13] = xt
        ring = np.full(10 * pd.Kp, np.nan, np.float32)
        out = np.zeros(2, np.float32)
        em.emul_fallback_rows()
        assert em.emul_strip_cost(_vp(pool), C.byref(pd), _vp(rows), L, _vp(xt16), _vp(out), _vp(ring), None, None) == 0
        nfallback += em.emul_fallback_rows()
        assert bits(out[0]) == bits(orc.null(prof, xt, seq)), (it, K, Q, W, S)
        assert bits(out[1]) == bits(orc.cost(prof, xt, seq)), (it, K, Q, W, S)
        # the same with the DP table kept, then the traceback of the fast path pass
        cells = np.full((L + 1) * 3 * pd.Kp, np.nan, np.float32)
        sp = np.full((L + 1) * 8, np.nan, np.float32)
        out2 = np.zeros(2, np.float32)
        assert em.emul_strip_cost(_vp(pool), C.byref(pd), _vp(rows), L, _vp(xt16), _vp(out2), _vp(ring), _vp(cells), _vp(sp)) == 0
        assert bits(out2[1]) == bits(out[1])
        cap = 2 * L + 2 * K + 64
        buf = np.zeros(cap, np.uint32)
        n = em.emul_traceback(_vp(pool), C.byref(pd), _vp(rows), L, _vp(xt16), _vp(cells), _vp(sp), _vp(buf), C.c_long(cap))
        if n > 0:  # n < 0: an exact tie the values cannot resolve (the literal pass takes over)
            _, xo, no = orc.path(prof, xt, seq)
            ids, sizes = orc.unzip(K, L, xo, no)
            w = buf[cap - n:]
            assert np.array_equal(w & 0xFFFF, ids.astype(np.uint32)) and np.array_equal(w >> 16, sizes.astype(np.uint32)), (it, K)
        else:
            assert n == -2, (it, K, n)
    assert nfallback > 0


def test_rows_replayed_from_the_table_give_the_reference_trellis(em, orc):
    """row_replay.h: with the DP table of the fast path pass, the trellis words of every row
    can be recomputed on their own, pass by pass with the reference's strict-< updates.  Tie-rich
    tables (every cost a multiple of 0.5..8), all kernel shapes incl. strips: every xnode and
    node word against the oracle."""
    rng = np.random.default_rng(29)
    for it in range(150):
        strips = 1
        if it % 5 == 4:
            Q, W, strips = [(1, 1, 3), (2, 2, 2), (1, 2, 3)][it % 3]
            K = int(rng.integers(64 * Q * W * (strips - 1) + 1, 64 * Q * W * strips + 1))
        else:
            K = int(rng.choice([2, 3, 5, 8, 9, 16, 17, 33, 64, 65, 100, 173, 241, 256, 300, 400, 600]))
            Q, W = choose_qw(K)
        quant = [None, 0.5, 1.0, 2.0, 4.0, 8.0][it % 6]
        prof = synth_profile(rng, K, quant, [0, 0.05, 0.3][it % 3])
        seq = random_seq(rng, int(rng.integers(1, 40)))
        L = len(seq)
        xt = orc.xtrans(max(L // 3, 1), it % 2, (it // 2) % 2)
        if quant:
            xt = (np.round(xt / quant) * quant).astype(np.float32)
        pool, pd = pack_profile(prof, Q, W, strips)
        rows = code_rows(seq)
        xt16 = np.zeros(16, np.float32)
        xt16[:13] = xt
        out = np.zeros(2, np.float32)
        cells = np.full((L + 1) * 3 * pd.Kp, np.nan, np.float32)
        sp = np.full((L + 1) * 8, np.nan, np.float32)
        if strips > 1:
            ring = np.zeros(10 * pd.Kp, np.float32)
            assert em.emul_strip_cost(_vp(pool), C.byref(pd), _vp(rows), L, _vp(xt16), _vp(out), _vp(ring), _vp(cells), _vp(sp)) == 0
        else:
            assert em.emul_cost_store(_vp(pool), C.byref(pd), _vp(rows), L, _vp(xt16), _vp(out), _vp(cells), _vp(sp)) == 0
        xn = np.full(L + 1, 0xFFFFFFFF, np.uint32)
        nd = np.full((L + 1) * K, 0xFFFF, np.uint16)
        assert em.emul_replay(_vp(pool), C.byref(pd), _vp(rows), L, _vp(xt16), _vp(cells), _vp(sp), _vp(xn), _vp(nd)) == 0
        score, xo, no = orc.path(prof, xt, seq)
        assert bits(out[1]) == bits(score)
        assert np.array_equal(xn, xo), (it, K, quant)
        assert np.array_equal(nd, no), (it, K, quant)


class Pack(C.Structure):
    """struct DcpPack (deciphon_amd/csrc/dcp_types.h)."""

    _fields_ = [("profile", C.c_int32), ("Lmax", C.c_int32), ("L", C.c_int32 * 16), ("xt_row", C.c_int32 * 16),
                ("out", C.c_int32 * 16), ("code_row", C.c_uint32 * 16)]


PACK_SHAPES = ((4, 1), (4, 2), (4, 4), (8, 2), (8, 4), (16, 2), (16, 3), (16, 4), (32, 2), (32, 3), (32, 4), (32, 6),
               (32, 8))  # (S, Q)


def run_pack(em, orc, prof, S, Q, seqs, mh=True, h3=False, quant=None):
    """Windows `seqs` (at most 64 / S) of one profile through PackWave<Q, S>; -> float32 [n][2]."""
    from dcp_testlib import choose_qw

    G = 64 // S
    assert len(seqs) <= G and prof.K <= (S - 1) * Q
    pool, pd = pack_profile(prof, *choose_qw(prof.K))  # the layout of the class the profile belongs to
    rows, first = [], []
    for s in seqs:
        first.append(sum(len(r) for r in rows))
        rows.append(code_rows(s))
    rows = np.ascontiguousarray(np.concatenate(rows))
    smax = max(max(len(s) // 3, 1) for s in seqs)
    xt = np.zeros((smax + 1, 16), np.float32)
    for s in range(1, smax + 1):
        v = orc.xtrans(s, mh, h3)
        xt[s, :13] = (np.round(v / quant) * quant).astype(np.float32) if quant else v
    pk = Pack()
    pk.profile, pk.Lmax = 0, max(len(s) for s in seqs)
    out = np.full((len(seqs) + 3, 2), np.float32(-7.0), np.float32)  # slots beyond the windows must stay untouched
    for g, s in enumerate(seqs):
        pk.L[g], pk.xt_row[g], pk.out[g], pk.code_row[g] = len(s), max(len(s) // 3, 1), g, first[g]
    assert em.emul_cost_pack(Q, S, _vp(pool), C.byref(pd), _vp(rows), len(rows), _vp(xt), C.byref(pk), _vp(out)) == 0
    assert np.all(out[len(seqs):] == np.float32(-7.0))
    if Q <= 4:  # the same windows with the short emission lengths' rows read from the LDS copy: the same bits
        out2 = np.full_like(out, np.float32(-7.0))
        assert em.emul_cost_pack_lds(Q, S, _vp(pool), C.byref(pd), _vp(rows), len(rows), _vp(xt), C.byref(pk), _vp(out2)) == 0
        assert np.array_equal(out.view(np.uint32), out2.view(np.uint32))
    return out[: len(seqs)], xt


def test_several_windows_per_wavefront(em, orc):
    """PackWave: groups of 4..32 lanes each run their own window of one profile -- full and partly filled
    packs, windows of different lengths in one pack (each group captures its results at its own last row),
    continuous and tie-rich tables, core sizes up to each shape's capacity (S - 1) * Q."""
    rng = np.random.default_rng(11)
    for it in range(220):
        S, Q = PACK_SHAPES[it % len(PACK_SHAPES)]
        cap = (S - 1) * Q
        K = int(rng.choice([1, 2, 3, cap // 2 + 1, cap - 1, cap])) if it % 3 else cap
        K = max(1, min(K, cap))
        if S * Q > 64 * choose_qw(K)[0]:  # the shape reads S * Q columns: never chosen for so short a row
            K = cap
        quant = [None, None, 1.0, 4.0][it % 4]
        prof = synth_profile(rng, K, quant, [0, 0.05, 0.3][it % 3])
        G = 64 // S
        n = G if it % 2 == 0 else int(rng.integers(1, G + 1))
        same = it % 5 == 0
        L0 = int(rng.integers(1, 48))
        seqs = [random_seq(rng, L0 if same else int(rng.integers(1, 48))) for _ in range(n)]
        out, xt = run_pack(em, orc, prof, S, Q, seqs, it % 2 == 0, it % 4 == 1, quant)
        for g, s in enumerate(seqs):
            x = np.ascontiguousarray(xt[max(len(s) // 3, 1), :13])
            assert bits(out[g, 0]) == bits(orc.null(prof, x, s)), (it, S, Q, K, g, len(s))
            assert bits(out[g, 1]) == bits(orc.cost(prof, x, s)), (it, S, Q, K, g, len(s))


def test_packed_windows_with_long_delete_runs(em, orc):
    """Nearly free D->D runs cross many lanes -- and must stop at the separator lane of the next group."""
    rng = np.random.default_rng(12)
    for S, Q in PACK_SHAPES:
        K = (S - 1) * Q
        prof = synth_profile(rng, K)
        if K > 1:
            prof.trans[7, 1:] = np.float32(0.01)  # DD
            prof.trans[3, 1:] = np.float32(0.02)  # MD
            prof.match[:, K // 3:] += np.float32(30.0)
        seqs = [random_seq(rng, int(rng.integers(20, 60))) for _ in range(64 // S)]
        out, xt = run_pack(em, orc, prof, S, Q, seqs)
        for g, s in enumerate(seqs):
            x = np.ascontiguousarray(xt[max(len(s) // 3, 1), :13])
            assert bits(out[g, 0]) == bits(orc.null(prof, x, s)) and bits(out[g, 1]) == bits(orc.cost(prof, x, s)), (S, Q, g)


def test_fast_path_pass_in_blocks(em, orc):
    """The DP table a block at a time (dcp_types.h): checkpoints of the folded ring every B rows, every block
    recomputed from its checkpoint into a table of B + 6 rows (never-written slots hold NaN), the traceback
    resumed from block to block -- the same steps as trellis_unzip on the oracle's trellis, for windows of one
    to a dozen blocks, block boundaries in and beside insert, delete and special-state runs."""
    em.emul_path_blocks.restype = C.c_int
    rng = np.random.default_rng(41)
    multi = ties = 0
    for it in range(160):
        K = int(rng.choice([3, 17, 64, 65, 100, 173, 241, 256, 300, 400, 600, 1000]))
        prof = synth_profile(rng, K, None, [0, 0.05][it % 2])
        if it % 4 == 0:  # long delete runs
            prof.trans[7, 1:] = np.float32(0.01)
            prof.trans[3, 1:] = np.float32(0.02)
        B = [5, 10, 15, 20, 0][it % 5]
        L = int(rng.integers(1, 70))
        seq = random_seq(rng, L)
        xt = orc.xtrans(max(L // 3, 1), it % 2, (it // 2) % 2)
        pool, pd = pack_profile(prof)
        rows = code_rows(seq)
        xt16 = np.zeros(16, np.float32)
        xt16[:13] = xt
        cap = 2 * L + 2 * K + 64
        buf = np.zeros(cap, np.uint32)
        score = C.c_float(0)
        n = em.emul_path_blocks(_vp(pool), C.byref(pd), _vp(rows), L, _vp(xt16), B, _vp(buf), C.c_long(cap), C.byref(score))
        s_o, xo, no = orc.path(prof, xt, seq)
        assert bits(score.value) == bits(s_o), (it, K, L, B)
        if not np.isfinite(s_o):
            continue
        if n == -2:  # an exact tie the values cannot resolve (the literal pass takes over): rare with these tables
            ties += 1
            continue
        assert n > 0, (it, K, L, B, n)
        ids, sizes = orc.unzip(K, L, xo, no)
        w = buf[cap - n:]
        assert np.array_equal(w & 0xFFFF, ids.astype(np.uint32)) and np.array_equal(w >> 16, sizes.astype(np.uint32)), (it, K, L, B)
        multi += B > 0 and L > B + 5
    assert multi > 60 and ties < 8
