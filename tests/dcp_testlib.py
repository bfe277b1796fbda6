"""Shared helpers of the test-suite: synthetic profiles/reads, the padded HBM layout
of deciphon_amd/csrc/dcp_types.h restated in numpy, and loaders for the fixtures."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle.pyoracle import Oracle, Profile, RefLib  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
TABLE_SIZE = 1364
INF = np.float32(np.inf)
CODE_OFF = (0, 4, 20, 84, 340)


class ProfileDev(C.Structure):
    """struct DcpProfileDev (deciphon_amd/csrc/dcp_types.h)."""

    _fields_ = [("K", C.c_int32), ("Kp", C.c_int32), ("Q", C.c_int32), ("W", C.c_int32),
                ("rows_off", C.c_int64), ("trans_off", C.c_int64), ("pad0", C.c_int64), ("pad1", C.c_int64)]


ROW_HDR = 4  # DCP_ROW_HDR: {null[c], bg[c], 0, 0} in front of every emission row


def synth_profile(rng, K: int, quant=None, pinf: float = 0.0) -> Profile:
    """Random profile in DP-cost space with the +inf structure protein_setup_viterbi
    (c-core/protein.c:363-383) guarantees.  `quant` rounds every cost to a multiple of
    it, which makes exact fp32 ties between distinct candidates common."""

    def costs(shape, scale):
        x = rng.random(shape).astype(np.float32) * np.float32(scale)
        if quant:
            x = np.round(x / quant) * quant
        x = x.astype(np.float32)
        if pinf:
            x[rng.random(shape) < pinf] = INF
        return x

    trans = costs((8, K), 6.0)
    trans[[1, 3, 4, 6, 7], 0] = INF  # MM MD IM DM DD of position 0
    trans[[2, 5], K - 1] = INF  # MI II of position K-1
    match = costs((TABLE_SIZE, K), 12.0)
    null = costs((TABLE_SIZE,), 8.0)
    bg = costs((TABLE_SIZE,), 8.0)
    return Profile(K, np.ascontiguousarray(trans), np.ascontiguousarray(match), null, bg, f"SYN{K}")


def choose_qw(K: int, path: bool = False):
    """(positions per lane, waves per problem) of the cost kernels -- or, path=True, of the
    pass-by-pass path kernel, which keeps at most 4 positions per lane and runs (3, 2W) where
    the cost kernel runs (6, W) / (8, W) on the same padded layout
    (deciphon_amd/csrc/viterbi_kernels.hip: dcp_class_of / dcp_class_shape / dcp_launch_path)."""
    if K <= 256:
        return (2 if 60 < K <= 64 else max(1, (K + 63) // 64)), 1  # 61..64: the 128-column layout (dcp_class_of)
    shapes = ((3, 2), (4, 2), (3, 4), (4, 4), (3, 8), (4, 8), (4, 16)) if path else \
             ((6, 1), (8, 1), (6, 2), (4, 4), (6, 4), (8, 4), (8, 8))
    for Q, W in shapes:
        if K <= 64 * Q * W:
            return Q, W
    raise ValueError("core size beyond 4096: the strip class (pack_profile(..., strips=))")


def pack_profile(prof: Profile, Q: int | None = None, W: int | None = None, strips: int = 1):
    """-> (pool float32[...], ProfileDev) in the padded layout the kernels read
    (strips > 1: the StripWave layout, Kp = strips * 64 * Q * W)."""
    K = prof.K
    if Q is None or W is None:
        Q, W = choose_qw(K)
    Kp = 64 * Q * W * strips
    assert K <= Kp
    rows = np.full((TABLE_SIZE, ROW_HDR + Kp), INF, dtype=np.float32)
    rows[:, 0] = prof.null
    rows[:, 1] = prof.bg
    rows[:, 2:ROW_HDR] = 0
    rows[:, ROW_HDR : ROW_HDR + K] = prof.match
    trans = np.full((8, Kp), INF, dtype=np.float32)
    trans[:, :K] = prof.trans
    pool = np.concatenate([rows.ravel(), trans.ravel()]).astype(np.float32)
    pd = ProfileDev(K, Kp, Q, W, 0, rows.size, 0, 0)
    return pool, pd


def code_rows(seq: np.ndarray) -> np.ndarray:
    """DcpCodeRow[len+1]: row r holds the codes of the t-mers covering r-t..r-1."""
    n = len(seq)
    rows = np.zeros((n + 1, 8), dtype=np.uint32)
    s = seq.astype(np.int64)
    for t in range(1, 6):
        if n < t:
            break
        idx = np.zeros(n - t + 1, dtype=np.int64)
        for i in range(t):
            idx = idx * 4 + s[i : n - t + 1 + i]
        rows[t:, t - 1] = CODE_OFF[t - 1] + idx
    return rows


def random_seq(rng, n: int) -> np.ndarray:
    return rng.integers(0, 4, size=n).astype(np.uint8)


def read_fasta(path: str):
    out, name, buf = [], None, []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                if name is not None:
                    out.append((name, "".join(buf)))
                name, buf = line[1:], []
            elif line:
                buf.append(line)
    if name is not None:
        out.append((name, "".join(buf)))
    return out


_oracle = None


def oracle() -> Oracle:
    global _oracle
    if _oracle is None:
        so = os.path.join(ROOT, "oracle", "libdcp_oracle.so")
        src = os.path.join(ROOT, "oracle", "dcp_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
        _oracle = Oracle()
    return _oracle


def reflib():
    """The reference's own viterbi.c (oracle/_ref), or None where it was not built."""
    if not RefLib.available() and os.path.isdir("/root/reference/c-core"):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    return RefLib() if RefLib.available() else None


def bits(x) -> int:
    return int(np.float32(x).view(np.uint32))


def oracle_scan(orc, proteins, reads, multi_hits, hmmer3_compat, threads: int = 1, epsilon: float = 0.01):
    """thread_run + process_window (c-core/thread.c:49-207) on the CPU oracle, without HMMER: the rows of
    products.tsv in the reference's order, codon and amino fields from oracle.pydecode (write_match,
    c-core/product_thread.c:112-148).  proteins: objects with the fields of oracle.dcp_reader.Protein;
    reads: [(id, text)].  threads > 1 spreads the proteins over a thread pool (ctypes drops the GIL)."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor

    from oracle import pydecode

    encoded = [(sid, text, orc.encode(text)) for sid, text in reads]

    def one(prot):
        rows = []
        prof = orc.setup_profile(prot)
        for sid, text, x in encoded:
            w = orc.lib.orc_window_setup(len(x), prof.K)
            while orc.lib.orc_window_next(C.byref(w)):
                seq = np.ascontiguousarray(x[w.start : w.stop])
                xt = orc.xtrans(max(len(seq) // 3, 1), multi_hits, hmmer3_compat)
                lrt = orc.lrt(-orc.null(prof, xt, seq), -orc.cost(prof, xt, seq))
                if not np.isfinite(lrt) or lrt < 0:
                    continue
                _, xn, nd = orc.path(prof, xt, seq)
                ids, sizes = orc.unzip(prof.K, len(seq), xn, nd)
                hit, last = orc.hits(ids, sizes)
                if hit is None:
                    continue
                w.last_hit_pos = last
                pos, cells = hit[0], []
                for st, sz in zip(ids[hit[2] : hit[3]], sizes[hit[2] : hit[3]]):
                    st, sz = int(st), int(sz)
                    dec = ","
                    if sz:  # an emitting state: insert -> background, match -> its node, N / J / C -> null model
                        kind, k = st >> 14, (st & 0x3FFF) - 1
                        entry = 1 if kind == 1 else 2 + k if kind == 0 else 0
                        codon, amino = pydecode.decode(epsilon, prot.nucltp[entry], prot.codonm[entry],
                                                       seq[pos : pos + sz])
                        dec = "".join("ACGT"[v] for v in codon) + "," + amino
                    cells.append(f"{text[w.start + pos : w.start + pos + sz]},{orc.state_name(st)},{dec}")
                    pos += sz
                rows.append(f"{sid}\t{w.idx}\t{w.start}\t{w.stop}\t0\t{hit[0]}\t{hit[1]}\t{prot.accession}\tdna\t"
                            f"{lrt:.1f}\tnan\t" + ";".join(cells))
        return rows

    if threads <= 1:
        parts = [one(p) for p in proteins]
    else:
        with ThreadPoolExecutor(threads) as ex:
            parts = list(ex.map(one, proteins))
    return [r for part in parts for r in part]
