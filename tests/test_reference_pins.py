"""CPU: pieces of the path that the REFERENCE's own code can check directly.  oracle/Makefile `ref` compiles,
unmodified and where they lie under /root/reference, every c-core file of the path that needs nothing but libc and
the reference's own headers: viterbi.c (the DP, used by test_oracle.py), error.c, partition_size.c, state.c,
disambiguate.c, uppercase.c.  Here the product's host code (include/deciphon_host.h) and the oracle restatement
are compared with those functions themselves -- not with a restatement of them."""
import ctypes as C

import numpy as np
import pytest

import deciphon_amd
from deciphon_amd import host
from dcp_testlib import random_seq, reflib, synth_profile


@pytest.fixture(scope="module")
def ref():
    r = reflib()
    if r is None:
        pytest.skip("oracle/_ref was not built (no /root/reference here and no prebuilt library)")
    L = r.lib
    L.partition_size.argtypes = [C.c_long, C.c_long, C.c_long]
    L.partition_size.restype = C.c_long
    L.state_name.argtypes = [C.c_int, C.c_char_p]
    for f in ("state_is_mute", "state_is_core", "state_is_start", "state_is_end"):
        getattr(L, f).argtypes = [C.c_int]
        getattr(L, f).restype = C.c_bool
    for f in ("state_make_match_id", "state_make_insert_id", "state_make_delete_id", "state_core_idx"):
        getattr(L, f).argtypes = [C.c_int]
        getattr(L, f).restype = C.c_int
    L.state_make_end.restype = C.c_int
    L.disambiguate.argtypes = [C.c_int, C.c_char_p]
    L.uppercase.argtypes = [C.c_size_t, C.c_char_p]
    L.dcp_error_string.argtypes = [C.c_int]
    L.dcp_error_string.restype = C.c_char_p
    return L


def test_partition_size_is_the_references(ref, orc):
    """c-core/partition_size.c:13-16 itself against dcp_partition_size and the by-count partition bounds
    (c-core/protein_reader.c:112-128 sums the same sizes)."""
    rng = np.random.default_rng(3)
    cases = [(3, 2), (20000, 8), (7, 7), (5, 9), (0, 4), (1, 1), (19632, 8), (19632, 3)]
    cases += [(int(rng.integers(0, 50000)), int(rng.integers(1, 130))) for _ in range(200)]
    for n, parts in cases:
        want = [ref.partition_size(n, parts, i) for i in range(parts)]
        assert [host.partition_size(n, parts, i) for i in range(parts)] == want, (n, parts)
        assert [orc.partition_size(n, parts, i) for i in range(parts)] == want, (n, parts)
        if n > 0:
            f = host.partition_bounds(np.ones(n, np.int32), parts, False)
            assert [int(f[i + 1] - f[i]) for i in range(parts)] == want, (n, parts)
            assert int(f[-1]) == n == sum(want)


def _ref_name(ref, sid):
    buf = C.create_string_buffer(16)
    assert ref.state_name(sid, buf) == 0
    return buf.value.decode()


def test_state_ids_are_the_references(ref, orc):
    """c-core/state.c:25,92-96 (+ state.h:9-25): every id the unzip can emit -- M/I/D of every core index up to
    the 14-bit limit and the specials -- is named, classified and built the same way."""
    for k in list(range(0, 300)) + [4095, 4096, 8191, 16382]:
        for make, tag in ((ref.state_make_match_id, 0), (ref.state_make_insert_id, 1), (ref.state_make_delete_id, 2)):
            sid = make(k)
            assert sid == (tag << 14) | (k + 1)
            assert ref.state_core_idx(sid) == k
            name = _ref_name(ref, sid)
            assert host.state_name(sid) == name == orc.state_name(sid)
            assert host.state_is_mute(sid) == bool(ref.state_is_mute(sid))
    for n in range(3, 10):  # S, N, B, E, J, C, T
        sid = 0xC000 | n
        assert host.state_name(sid) == _ref_name(ref, sid) == orc.state_name(sid)
        assert host.state_is_mute(sid) == bool(ref.state_is_mute(sid))
    assert ref.state_make_end() == 0xC000 | 9


def test_unzipped_paths_hold_reference_state_ids(ref, orc):
    """The ids of real unzipped paths (oracle trellis -> product's dcp_unzip) re-made by the reference's makers."""
    rng = np.random.default_rng(21)
    for it in range(60):
        K = int(rng.choice([2, 7, 33, 100, 257]))
        prof = synth_profile(rng, K, [None, 1.0][it % 2])
        seq = random_seq(rng, int(rng.integers(1, 120)))
        xt = orc.xtrans(max(len(seq) // 3, 1), bool(it % 2), False)
        _, xn, nd = orc.path(prof, xt, seq)
        ids, sizes = host.unzip(K, len(seq), xn, nd)
        for sid in ids:
            sid = int(sid)
            name = _ref_name(ref, sid)
            assert name == host.state_name(sid)
            if name[0] in "MID":
                k = int(name[1:]) - 1
                make = {"M": ref.state_make_match_id, "I": ref.state_make_insert_id, "D": ref.state_make_delete_id}[name[0]]
                assert make(k) == sid and 0 <= k < K
        assert int(ids[-1]) == ref.state_make_end()


def _ref_encode(ref, text):
    """sequence_setup's first steps (c-core/sequence.c:20-24): uppercase, then disambiguate; then symbol indices
    (A, C, G, T|U = 0..3: imm's DNA / RNA alphabets)."""
    raw = text.encode("latin-1")
    buf = C.create_string_buffer(raw, len(raw) + 1)
    ref.uppercase(len(raw), buf)
    rc = ref.disambiguate(len(raw), buf)
    return rc, buf.raw[: len(raw)].decode("latin-1")


def test_read_encoding_is_the_references(ref, orc):
    """c-core/uppercase.c + c-core/disambiguate.c themselves against dcp_encode_sequence (and the oracle's)."""
    rng = np.random.default_rng(77)
    dna = "ACGTacgtRYMKSWHBVDNXrymkswhbvdnx"
    rna = "ACGUacguRYMKSWHBVDNXrymkswhbvdnx"
    cases = ["A", "n", "NNNN", "RYRY", "ryKM", "UUUy", "TTTy", "cGGs", "SSSS", "WWWW", "HBVD", "XN"]
    for letters in (dna, rna, "ACGTN", "acgun"):
        cases += ["".join(rng.choice(list(letters), size=int(rng.integers(1, 80)))) for _ in range(300)]
    for s in cases:
        rc, text = _ref_encode(ref, s)
        assert rc == 0, s
        assert set(text) <= set("ACGTU"), (s, text)
        want = np.array(["ACGTU".index(c) if c != "U" else 3 for c in text], np.uint8)
        assert np.array_equal(deciphon_amd.encode(s), want), (s, text)
        assert np.array_equal(orc.encode(s), want), (s, text)
    # T and U in one read: the reference refuses (DCP_ENUCLTSEQTU), so does the product
    import os

    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(2)
    os.dup2(devnull, 2)  # error() logs at the raise site (c-core/error.c:102-121)
    try:
        rc, _ = _ref_encode(ref, "ACGTU")
    finally:
        os.dup2(saved, 2)
        os.close(devnull)
        os.close(saved)
    assert rc == 74
    with pytest.raises(deciphon_amd.HipError) as e:
        deciphon_amd.encode("ACGTU")
    assert e.value.code == rc


def test_error_strings_are_the_references(ref):
    """c-core/error.c's table itself (dcp_error_string) against the product's, for every code and beyond."""
    holes = []
    for code in range(-2, 125):
        want = ref.dcp_error_string(code)
        if want is None:  # a hole in the reference's designated-initialiser table: it returns NULL there
            holes.append(code)
            assert deciphon_amd.error_string(code) == ""  # the product returns an empty string, never NULL
        else:
            assert deciphon_amd.error_string(code) == want.decode(), code
    assert holes == [62]
