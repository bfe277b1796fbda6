"""CPU: the product's host-side code (deciphon_amd/csrc/host_logic.cpp, dcp_db.cpp) through
its C ABI (include/deciphon_host.h, the no-GPU part of deciphon_hip.h) against the oracle
and the independent Python .dcp reader.  No compute call needs a GPU here."""
import os

import numpy as np
import pytest

import deciphon_amd
from deciphon_amd import host
from dcp_testlib import GOLDEN, ROOT, random_seq, read_fasta, synth_profile
from oracle.dcp_reader import read_dcp


def test_xtrans_bit_exact_for_every_length(orc):
    for S in list(range(1, 3000)) + [9999, 20000, 33333]:
        for mh in (False, True):
            for h3 in (False, True):
                a, b = deciphon_amd.xtrans(S, mh, h3), orc.xtrans(S, mh, h3)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (S, mh, h3)


def test_encode_matches_oracle(orc):
    rng = np.random.default_rng(5)
    letters = "ACGTacgtRYMKSWHBVDNXrymkswhbvdnx"
    for _ in range(300):
        s = "".join(rng.choice(list(letters), size=int(rng.integers(1, 60))))
        assert np.array_equal(deciphon_amd.encode(s), orc.encode(s)), s
    s = "".join(rng.choice(list("ACGUnry"), size=50))
    assert np.array_equal(deciphon_amd.encode(s), orc.encode(s))
    with pytest.raises(deciphon_amd.HipError) as e:
        deciphon_amd.encode("ACGTU")
    assert e.value.code == 74  # DCP_ENUCLTSEQTU
    with pytest.raises(deciphon_amd.HipError) as e:
        deciphon_amd.encode("AC-GT")
    assert e.value.code == 57  # DCP_ESEQABC


def test_dcp_reader_against_python_reader():
    path = os.path.join(GOLDEN, "minifam.dcp")
    ref = read_dcp(path)
    db = host.Database(path)
    assert len(db) == 3 and abs(db.epsilon - 0.01) < 1e-9 and db.entry_dist == 2 and db.has_ga
    off = [db.offset(i) for i in range(4)]
    assert [off[i + 1] - off[i] for i in range(3)] == ref.protein_sizes
    assert off[3] == os.path.getsize(path)  # parsed to exact EOF
    for i, want in enumerate(ref.proteins):
        got = db.protein(i)
        assert got["core_size"] == want.core_size and got["accession"] == want.accession
        assert got["consensus"] == want.consensus
        for key, arr in (("trans", want.trans), ("emission", want.emission), ("BMk", want.BMk),
                         ("null_emission", want.null_emission), ("bg_emission", want.bg_emission)):
            assert np.array_equal(got[key].view(np.uint32), arr.view(np.uint32)), (i, key)
    db.close()


def test_dcp_reader_current_writer_encoding(tmp_path):
    """The current writer emits `bin` + native floats and an int array of sizes
    (c-core/write.c:59-66, c-core/database_writer.c:76-93); re-encode the fixture that way."""
    import msgpack

    src = read_dcp(os.path.join(GOLDEN, "minifam.dcp"))

    def f32(a):
        return np.ascontiguousarray(a, "<f4").tobytes()

    def nuclt():
        return [f32(np.zeros(4, np.float32)), f32(np.zeros(125, np.float32))]

    packer = msgpack.Packer(use_bin_type=True, use_single_float=True)
    prots = []
    for p in src.proteins:
        K = p.core_size
        nodes = b"".join(
            packer.pack("nuclt_dist") + packer.pack(nuclt()) + packer.pack("trans") + packer.pack(f32(p.trans[i]))
            + packer.pack("emission") + packer.pack(f32(p.emission[i])) for i in range(K + 1))
        body = (packer.pack_map_header(10) + packer.pack("accession") + packer.pack(p.accession)
                + packer.pack("gencode") + packer.pack(p.gencode) + packer.pack("consensus")
                + packer.pack(p.consensus) + packer.pack("core_size") + packer.pack(K)
                + packer.pack("null_nuclt_dist") + packer.pack(nuclt()) + packer.pack("null_emission")
                + packer.pack(f32(p.null_emission)) + packer.pack("bg_nuclt_dist") + packer.pack(nuclt())
                + packer.pack("bg_emission") + packer.pack(f32(p.bg_emission)) + packer.pack("nodes")
                + packer.pack_map_header((K + 1) * 3) + nodes + packer.pack("BMk") + packer.pack(f32(p.BMk)))
        prots.append(body)
    abc = {"symbols": "ACGT", "idx": b"\0" * 94, "any_symbol_id": 55, "typeid": 4}
    header = (packer.pack_map_header(8) + packer.pack("magic_number") + packer.pack(0xC6F1)
              + packer.pack("version") + packer.pack(1) + packer.pack("entry_dist") + packer.pack(2)
              + packer.pack("epsilon") + packer.pack(0.01) + packer.pack("abc") + packer.pack(abc)
              + packer.pack("amino") + packer.pack(dict(abc, symbols="ACDEFGHIKLMNPQRSTVWY", typeid=2))
              + packer.pack("has_ga") + packer.pack(True) + packer.pack("protein_sizes")
              + packer.pack([len(b) for b in prots]))
    blob = (packer.pack_map_header(2) + packer.pack("header") + header + packer.pack("proteins")
            + packer.pack_array_header(len(prots)) + b"".join(prots))
    path = tmp_path / "current.dcp"
    path.write_bytes(blob)
    db = host.Database(str(path))
    assert len(db) == 3
    for i, want in enumerate(src.proteins):
        got = db.protein(i)
        assert got["accession"] == want.accession
        assert np.array_equal(got["emission"].view(np.uint32), want.emission.view(np.uint32))
        assert np.array_equal(got["trans"].view(np.uint32), want.trans.view(np.uint32))
    again = read_dcp(str(path))  # the Python reader accepts it as well
    assert [p.core_size for p in again.proteins] == [173, 241, 162]


def test_dcp_reader_rejects_bad_files(tmp_path):
    p = tmp_path / "bad.dcp"
    p.write_bytes(b"\x82\xa6header\x88\xacmagic_number\xcd\x12\x34")
    with pytest.raises(deciphon_amd.HipError) as e:
        host.Database(str(p))
    assert e.value.code == 69  # DCP_ENOTDBFILE
    with pytest.raises(deciphon_amd.HipError) as e:
        host.Database(str(tmp_path / "missing.dcp"))
    assert e.value.code == 21  # DCP_EOPENDB
    raw = open(os.path.join(GOLDEN, "minifam.dcp"), "rb").read()
    q = tmp_path / "trunc.dcp"
    q.write_bytes(raw[:500000])
    with pytest.raises(deciphon_amd.HipError) as e:
        host.Database(str(q))
    assert e.value.code == 3  # DCP_EFDATA


def test_windows_match_oracle(orc):
    rng = np.random.default_rng(11)
    for _ in range(200):
        K = int(rng.integers(1, 400))
        n = int(rng.integers(1, 60000))
        hits = {}
        it = host.WindowIter(n, K)
        got = []
        while (w := it.next()) is not None:
            got.append(w)
            if rng.random() < 0.4:
                hits[w[0]] = int(rng.integers(0, w[2] - w[1]))
                it.set_last_hit_position(hits[w[0]])
        assert got == orc.windows(n, K, lambda idx: hits.get(idx))


def test_unzip_and_hits_match_oracle(orc):
    rng = np.random.default_rng(12)
    for it in range(150):
        K = int(rng.choice([2, 3, 9, 33, 100]))
        prof = synth_profile(rng, K, [None, 1.0][it % 2])
        seq = random_seq(rng, int(rng.integers(1, 90)))
        xt = orc.xtrans(max(len(seq) // 3, 1), bool(it % 2), False)
        _, xn, nd = orc.path(prof, xt, seq)
        want = orc.unzip(K, len(seq), xn, nd)
        got = host.unzip(K, len(seq), xn, nd)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        h, last = orc.hits(want[0], want[1])
        g = host.path_hit(got[0], got[1])
        assert (g is None) == (h is None)
        if h is not None:
            assert g[:4] == h and g[4] == last
        for st in got[0]:
            assert host.state_name(st) == orc.state_name(st)
    assert host.state_is_mute(0xC000 | 3) and not host.state_is_mute(0xC000 | 4) and host.state_is_mute(2 << 14 | 5)
    assert host.lrt(-691.526245, -545.70282) == orc.lrt(-691.526245, -545.70282)


def test_partition_size():
    for n, k in ((3, 2), (20000, 8), (7, 7), (5, 9), (0, 4)):
        sizes = [host.partition_size(n, k, i) for i in range(k)]
        assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
        assert sizes == sorted(sizes, reverse=True)


def test_error_strings():
    assert deciphon_amd.error_string(20) == "not enough memory"
    assert deciphon_amd.error_string(57).startswith("invalid sequence letter")
    assert deciphon_amd.error_string(80) == "invalid number of proteins"
    assert deciphon_amd.error_string(999) == "unknown error #999"


def test_no_cpu_fallback():
    """On a machine without a GPU the engine refuses to exist instead of computing on the host."""
    if deciphon_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(deciphon_amd.HipError) as e:
        deciphon_amd.Engine(0)
    assert e.value.code == 8  # DCP_EFUNCUSE


def test_host_reader_under_sanitizers(tmp_path):
    """The C++ host side that never touches the GPU (.dcp reader, windows, partitions) built
    with -fsanitize=address,undefined and run over the golden database, truncated copies and
    200 randomly corrupted copies: every one must parse or fail with an error code."""
    import subprocess

    csrc = os.path.join(ROOT, "deciphon_amd", "csrc")
    exe = str(tmp_path / "host_sanitize")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-fno-omit-frame-pointer", "-I", csrc, "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "host_sanitize.cpp"), os.path.join(csrc, "host_logic.cpp"),
                    os.path.join(csrc, "dcp_db.cpp"), os.path.join(csrc, "host_capi.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe, os.path.join(GOLDEN, "minifam.dcp"), str(tmp_path / "cut.dcp")], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fuzz done" in r.stdout and "K=173 acc=PF00742.20 rc=0" in r.stdout
    assert "window 1 [7959,10000)" in r.stdout  # c-core/window.c on a 10 kb read, K = 173


def test_both_dcp_encodings_read_identically(tmp_path):
    """The golden database (legacy `ext` encoding) re-encoded by deciphon_amd.synth.write_dcp in the CURRENT
    writer's form (f32 arrays as `bin`, protein_sizes as an int array: c-core/write.c:59-66,
    c-core/database_writer.c:76-93; read by c-core/read.c:118-132, c-core/database_reader.c:103-130) and in
    the legacy form again: the product's reader returns bit-identical proteins from all three files, and so
    does the independent Python reader.  (No reference-held fixture exists in the current encoding: that
    form is pinned to the reference's writer SOURCE only -- parity unpinned beyond it.)"""
    from deciphon_amd import host, synth
    from oracle.dcp_reader import read_dcp

    seeds = synth.load_seeds(os.path.join(GOLDEN, "minifam.dcp"))
    assert [s["core_size"] for s in seeds] == [173, 241, 162]
    for legacy in (False, True):
        path = str(tmp_path / f"re{int(legacy)}.dcp")
        synth.write_dcp(path, seeds, legacy=legacy)
        db, py = host.Database(path), read_dcp(path)
        assert len(db) == 3 and list(db.core_sizes()) == [173, 241, 162]
        for i, s in enumerate(seeds):
            p = db.protein(i)
            assert (p["accession"], p["consensus"]) == (s["accession"], s["consensus"])
            for k in ("trans", "emission", "BMk", "null_emission", "bg_emission"):
                assert np.array_equal(p[k].view(np.uint32), s[k].view(np.uint32)), (legacy, i, k)
                assert np.array_equal(getattr(py.proteins[i], k).view(np.uint32).ravel(), s[k].view(np.uint32).ravel())
        db.close()


def test_partition_bounds_by_count_and_balanced():
    """By count: the reference's rule (c-core/partition_size.c:13-16 summed as c-core/protein_reader.c:112-128
    does).  Balanced: contiguous, in order, and on a Pfam-shaped database (2*10^4 lengths from the log-normal
    of deciphon_amd.synth) no GPU of 2..8 gets more than 1.05 x the mean sum of core sizes."""
    from dcp_testlib import oracle
    from deciphon_amd import host, synth

    orc = oracle()
    for n in (0, 1, 3, 10, 20000):
        for parts in (1, 2, 3, 8, 128):
            f = host.partition_bounds(np.ones(n, np.int32), parts, False)
            assert [int(f[i + 1] - f[i]) for i in range(parts)] == [orc.partition_size(n, parts, i) for i in range(parts)]
    K = synth.pfam_like_lengths(20000, 3)
    for parts in (2, 3, 4, 8):
        f = host.partition_bounds(K, parts, True)
        assert f[0] == 0 and f[-1] == len(K) and np.all(np.diff(f) >= 0)
        sums = np.array([K[f[i] : f[i + 1]].sum() for i in range(parts)], np.float64)
        assert sums.max() / sums.mean() <= 1.05
    # a sorted database (longest profiles last) is where splitting by count goes wrong and by size does not
    Ks = np.sort(K)
    by_count = host.partition_bounds(Ks, 8, False)
    by_size = host.partition_bounds(Ks, 8, True)
    s0 = np.array([Ks[by_count[i] : by_count[i + 1]].sum() for i in range(8)], np.float64)
    s1 = np.array([Ks[by_size[i] : by_size[i + 1]].sum() for i in range(8)], np.float64)
    assert s0.max() / s0.mean() > 2.0 and s1.max() / s1.mean() <= 1.05
    assert list(host.partition_bounds(np.array([5, 5, 5], np.int32), 5, True)) == [0, 1, 1, 2, 2, 3]
