"""Synthetic workloads in the reference's own data formats: Pfam-shaped pressed databases
built by resampling the nodes of a seed database, reads with planted error-bearing domains,
and a `.dcp` writer for both array encodings found in the wild.

Pfam-A itself is not available offline (SURVEY 8d config 4 names this fallback): a profile
here is a chain of node runs cut out of the seed proteins (control/tests/files/minifam.dcp in
the tests and the benchmark), so its emission tables, transitions and entry costs have the
values and the structure of real pressed profiles; only the lengths are drawn, from a
log-normal fitted to Pfam-A's model lengths (median 140, mean ~173, tail to 2500).

`.dcp` schema: c-core/database_writer.c:95-193 (header, protein_sizes), c-core/protein.c:234-281
(protein_pack), c-core/write.c:59-66 (f32 arrays as `bin`); the legacy encoding is the one of
the reference's committed fixture (f32 arrays as big-endian `ext` type 8, protein_sizes as one
`ext` type 6: SURVEY Appendix A).  No third-party msgpack module: the few MessagePack forms the
format needs are emitted here.
"""
from __future__ import annotations

import struct

import numpy as np

TABLE_SIZE = 1364
AMINO = "ACDEFGHIKLMNPQRSTVWY"
# one fixed codon per amino acid, to back-translate a consensus into a plantable domain
CODON = dict(zip(AMINO, ["GCT", "TGT", "GAT", "GAA", "TTT", "GGT", "CAT", "ATT", "AAA", "CTG", "ATG", "AAT", "CCT",
                         "CAA", "CGT", "TCT", "ACT", "GTT", "TGG", "TAT"]))
_NT = {"A": 0, "C": 1, "G": 2, "T": 3}


# ---- profiles ------------------------------------------------------------------------------

def pfam_like_lengths(n: int, seed: int, lo: int = 10, hi: int = 2500) -> np.ndarray:
    """n core sizes from a log-normal (median 140, sigma 0.65: mean ~173), clipped to [lo, hi]."""
    rng = np.random.default_rng(seed)
    return np.clip(np.exp(rng.normal(np.log(140.0), 0.65, size=n)).astype(np.int64), lo, hi)


def _finish(K, src, seeds, accession):
    """src[K] = (seed protein, node) per position -> a protein dict in protein_unpack's layout."""
    src = np.asarray(src, np.int64).reshape(K, 2)
    P, I = src[:, 0], src[:, 1]
    emission = np.empty((K + 1, TABLE_SIZE), np.float32)
    trans = np.empty((K + 1, 7), np.float32)
    BMk = np.empty(K, np.float32)
    nucltp = np.zeros((K + 3, 4), np.float32)    # 0 = null, 1 = background, 2 + n = node n
    codonm = np.zeros((K + 3, 125), np.float32)
    s0 = seeds[int(P[0])]
    if "nucltp" in s0:
        nucltp[:2], codonm[:2] = s0["nucltp"][:2], s0["codonm"][:2]
    cons = np.full(K, "x", dtype="<U1")
    for p, s in enumerate(seeds):  # one gather per seed protein, not one copy per node
        at = np.nonzero(P == p)[0]
        if not len(at):
            continue
        i = I[at]
        emission[at] = s["emission"][i]
        trans[at] = s["trans"][i]
        BMk[at] = s["BMk"][i]
        if "nucltp" in s:
            nucltp[2 + at], codonm[2 + at] = s["nucltp"][2 + i], s["codonm"][2 + i]
        c = np.array(list(s["consensus"]), dtype="<U1")
        ok = i < len(c)
        cons[at[ok]] = c[i[ok]]
    nucltp[2 + K], codonm[2 + K] = nucltp[1 + K], codonm[1 + K]
    # the end of a model as the reference builds it (c-core/model.c; visible in any pressed profile):
    # node K duplicates node K-1, whose MD and DD are impossible and whose DM is certain
    emission[K] = emission[K - 1]
    trans[K - 1, 2] = trans[K - 1, 6] = -np.inf
    trans[K - 1, 5] = 0.0
    trans[K] = trans[K - 1]
    return dict(core_size=K, accession=accession, gencode=int(s0.get("gencode", 1)), consensus="".join(cons),
                trans=trans, emission=emission, BMk=BMk, null_emission=np.array(s0["null_emission"], np.float32),
                bg_emission=np.array(s0["bg_emission"], np.float32), nucltp=nucltp, codonm=codonm)


def resample_protein(seeds, K: int, rng, accession: str, mean_run: int = 30) -> dict:
    """A protein of K nodes made of runs of consecutive interior nodes of the seed proteins."""
    src = []
    while len(src) < K:
        p = int(rng.integers(0, len(seeds)))
        Ks = seeds[p]["core_size"]
        run = int(min(rng.geometric(1.0 / mean_run), K - len(src), max(Ks - 2, 1)))
        i0 = int(rng.integers(0, max(Ks - 1 - run, 0) + 1))
        src += [(p, i0 + j) for j in range(run)]
    return _finish(K, src, seeds, accession)


def tile_protein(seeds, K: int, offset: int, accession: str) -> dict:
    """A protein of K nodes: the interior nodes of the seed proteins one after the other, cyclically,
    starting `offset` nodes in (deterministic: the long profiles of SURVEY 8d config 3b)."""
    ring = [(p, i) for p, s in enumerate(seeds) for i in range(s["core_size"] - 1)]
    src = [ring[(offset + k) % len(ring)] for k in range(K)]
    return _finish(K, src, seeds, accession)


def load_seeds(path: str):
    """Every protein of a pressed database, through the product's own reader."""
    from .host import Database

    db = Database(path)
    seeds = [db.protein(i) for i in range(len(db))]
    db.close()
    return seeds


def pfam_like_database(seeds, n: int, seed: int, first: int = 0, lengths=None):
    """Proteins first..first+n-1 of the (conceptually endless) Pfam-shaped database `seed`: protein i
    depends on (seed, i) only, so partitions can be generated independently of each other."""
    return list(iter_pfam_like(seeds, n, seed, first, lengths))


# ---- reads ---------------------------------------------------------------------------------

def back_translate(consensus: str) -> np.ndarray:
    return np.array([_NT[ch] for a in consensus for ch in CODON.get(a.upper(), "GCT")], dtype=np.uint8)


def mutate(dom: np.ndarray, rng, sub: float, ins: float, dele: float) -> np.ndarray:
    out = []
    for b in dom:
        u = rng.random()
        if u < dele:
            continue
        if u < dele + ins:
            out.append(rng.integers(0, 4))
        out.append(rng.integers(0, 4) if rng.random() < sub else b)
    return np.array(out, dtype=np.uint8)


def synth_reads(nreads: int, length: int, consensus, seed: int, planted_every: int = 10, sub: float = 0.10,
                ins: float = 0.03, dele: float = 0.03, first: int = 0):
    """iid-uniform ACGT reads (read i depends on (seed, first + i) only); every `planted_every`-th carries
    one planted domain: a profile consensus (at most 400 aa of it) back-translated, with substitutions,
    insertions and deletions (10 % / 3 % / 3 %)."""
    reads = []
    for j in range(nreads):
        i = first + j
        rng = np.random.default_rng([seed, 7, i])
        r = rng.integers(0, 4, size=length).astype(np.uint8)
        if planted_every and i % planted_every == 0 and consensus:
            cons = consensus[(i // planted_every) % len(consensus)]
            if len(cons) > 400:
                a = int(rng.integers(0, len(cons) - 400))
                cons = cons[a : a + 400]
            dom = mutate(back_translate(cons), rng, sub, ins, dele)[: max(1, length - 10)]
            at = int(rng.integers(0, length - len(dom) + 1))
            r[at : at + len(dom)] = dom
        reads.append(r)
    return reads


# ---- .dcp writer ---------------------------------------------------------------------------

def _u(n: int) -> bytes:  # MessagePack unsigned int
    if n < 128:
        return bytes([n])
    if n < 1 << 8:
        return b"\xcc" + struct.pack(">B", n)
    if n < 1 << 16:
        return b"\xcd" + struct.pack(">H", n)
    if n < 1 << 32:
        return b"\xce" + struct.pack(">I", n)
    return b"\xcf" + struct.pack(">Q", n)


def _s(text: str) -> bytes:
    b = text.encode()
    if len(b) < 32:
        return bytes([0xA0 | len(b)]) + b
    if len(b) < 1 << 8:
        return b"\xd9" + struct.pack(">B", len(b)) + b
    if len(b) < 1 << 16:
        return b"\xda" + struct.pack(">H", len(b)) + b
    return b"\xdb" + struct.pack(">I", len(b)) + b


def _map(n: int) -> bytes:
    return bytes([0x80 | n]) if n < 16 else (b"\xde" + struct.pack(">H", n) if n < 1 << 16 else b"\xdf" + struct.pack(">I", n))


def _arr(n: int) -> bytes:
    return bytes([0x90 | n]) if n < 16 else (b"\xdc" + struct.pack(">H", n) if n < 1 << 16 else b"\xdd" + struct.pack(">I", n))


def _bin(b: bytes) -> bytes:
    if len(b) < 1 << 8:
        return b"\xc4" + struct.pack(">B", len(b)) + b
    if len(b) < 1 << 16:
        return b"\xc5" + struct.pack(">H", len(b)) + b
    return b"\xc6" + struct.pack(">I", len(b)) + b


def _ext(code: int, b: bytes) -> bytes:
    fix = {1: 0xD4, 2: 0xD5, 4: 0xD6, 8: 0xD7, 16: 0xD8}
    if len(b) in fix:
        return bytes([fix[len(b)], code]) + b
    if len(b) < 1 << 8:
        return b"\xc7" + struct.pack(">B", len(b)) + bytes([code]) + b
    if len(b) < 1 << 16:
        return b"\xc8" + struct.pack(">H", len(b)) + bytes([code]) + b
    return b"\xc9" + struct.pack(">I", len(b)) + bytes([code]) + b


def _f32(a, legacy: bool) -> bytes:
    if legacy:
        return _ext(8, np.ascontiguousarray(a, ">f4").tobytes())
    return _bin(np.ascontiguousarray(a, "<f4").tobytes())


def _f32_head(n: int, legacy: bool) -> bytes:
    """The MessagePack header _f32 puts in front of n floats."""
    return _f32(np.zeros(n, np.float32), legacy)[: -4 * n]


def pack_protein(p: dict, legacy: bool = False) -> bytes:
    """protein_pack, c-core/protein.c:234-281.  The K + 1 node records all have the same byte layout, so they are
    laid out as ONE [K + 1][node bytes] array: constant key / header bytes and four float payloads per row."""
    K = int(p["core_size"])
    nucltp = p.get("nucltp", np.zeros((K + 3, 4), np.float32))
    codonm = p.get("codonm", np.zeros((K + 3, 125), np.float32))
    dt = ">f4" if legacy else "<f4"

    def nuclt(j):  # nuclt_dist_pack, c-core/nuclt_dist.c:13-20
        return _arr(2) + _f32(nucltp[j], legacy) + _f32(codonm[j], legacy)

    pieces = [(_s("nuclt_dist") + _arr(2) + _f32_head(4, legacy), nucltp[2 : K + 3]),
              (_f32_head(125, legacy), codonm[2 : K + 3]),
              (_s("trans") + _f32_head(7, legacy), p["trans"][: K + 1]),
              (_s("emission") + _f32_head(TABLE_SIZE, legacy), p["emission"][: K + 1])]
    width = sum(len(h) + 4 * a.shape[1] for h, a in pieces)
    rec = np.empty((K + 1, width), np.uint8)
    at = 0
    for h, a in pieces:
        rec[:, at : at + len(h)] = np.frombuffer(h, np.uint8)
        at += len(h)
        n = 4 * a.shape[1]
        rec[:, at : at + n] = np.ascontiguousarray(a, dt).view(np.uint8).reshape(K + 1, n)
        at += n
    nodes = rec.tobytes()
    return (_map(10) + _s("accession") + _s(p["accession"]) + _s("gencode") + _u(int(p.get("gencode", 1)))
            + _s("consensus") + _s(p["consensus"]) + _s("core_size") + _u(K) + _s("null_nuclt_dist") + nuclt(0)
            + _s("null_emission") + _f32(p["null_emission"], legacy) + _s("bg_nuclt_dist") + nuclt(1)
            + _s("bg_emission") + _f32(p["bg_emission"], legacy) + _s("nodes") + _map((K + 1) * 3) + nodes
            + _s("BMk") + _f32(p["BMk"], legacy))


def write_dcp(path: str, proteins, epsilon: float = 0.01, legacy: bool = False, rna: bool = False) -> list:
    """proteins: an iterable of dicts in the layout of deciphon_amd.host.Database.protein.  Each is packed, appended
    to a temporary file and dropped (the header, which comes first, needs every record's size), so a database far
    larger than memory can be streamed from a generator -- as the reference's own writer does with its temporary
    chunk files (c-core/database_writer.c:14,204-208).  Returns the records' byte sizes."""
    import os
    import shutil

    sizes = []
    body = path + ".proteins.tmp"
    with open(body, "wb") as f:
        for p in proteins:
            b = pack_protein(p, legacy)
            sizes.append(len(b))
            f.write(b)
    abc = (_map(4) + _s("symbols") + _s("ACGU" if rna else "ACGT") + _s("idx") + _ext(1, b"\0" * 94)
           + _s("any_symbol_id") + _u(55) + _s("typeid") + _u(5 if rna else 4))
    amino = (_map(4) + _s("symbols") + _s(AMINO) + _s("idx") + _ext(1, b"\0" * 94) + _s("any_symbol_id") + _u(55)
             + _s("typeid") + _u(2))
    if legacy:
        psz = _ext(6, np.array(sizes, ">u4").tobytes())
    else:
        psz = _arr(len(sizes)) + b"".join(_u(n) for n in sizes)
    header = (_map(8) + _s("magic_number") + _u(0xC6F1) + _s("version") + _u(1) + _s("entry_dist") + _u(2)
              + _s("epsilon") + b"\xca" + struct.pack(">f", float(epsilon)) + _s("abc") + abc + _s("amino") + amino
              + _s("has_ga") + b"\xc3" + _s("protein_sizes") + psz)
    try:
        with open(path, "wb") as f, open(body, "rb") as g:
            f.write(_map(2) + _s("header") + header + _s("proteins") + _arr(len(sizes)))
            shutil.copyfileobj(g, f, 64 << 20)
    finally:
        os.unlink(body)
    return sizes


def iter_pfam_like(seeds, n: int, seed: int, first: int = 0, lengths=None):
    """pfam_like_database as a generator: one protein alive at a time."""
    Ks = pfam_like_lengths(first + n, seed)[first:] if lengths is None else lengths
    for j, K in enumerate(Ks):
        i = first + j
        yield resample_protein(seeds, int(K), np.random.default_rng([seed, i]), f"SY{i:05d}.1")
