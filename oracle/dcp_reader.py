"""TEST INFRASTRUCTURE ONLY -- independent Python reader of pressed `.dcp` databases.

Schema after the reference's writer/reader pair: c-core/database_writer.c:95-149,
c-core/database_reader.c:26-80, c-core/protein.c:234-351.
Accepts both numeric-array encodings (current `bin` native-endian; the legacy
big-endian `ext` types 8 / 6 of the committed fixture control/tests/files/minifam.dcp).
It cross-checks the product's own C++ reader (deciphon_amd/csrc/dcp_db.cpp) in
tests/; the product never imports this module.
"""
from __future__ import annotations

import dataclasses

import msgpack
import numpy as np

TABLE_SIZE = 1364  # c-core/protein_node_size.h:4-9
TRANS_SIZE = 7  # c-core/trans.h:8-27 : MM MI MD IM II DM DD


@dataclasses.dataclass
class Protein:
    accession: str
    gencode: int
    consensus: str
    core_size: int
    null_emission: np.ndarray  # [1364] log-probs
    bg_emission: np.ndarray  # [1364]
    trans: np.ndarray  # [(K+1), 7]
    emission: np.ndarray  # [(K+1), 1364] node-major
    BMk: np.ndarray  # [K]
    nucltp: np.ndarray = None  # [(K+3), 4]: 0 = null, 1 = background, 2 + n = node n (decoder_setup, c-core/decoder.c:21-36)
    codonm: np.ndarray = None  # [(K+3), 125]


@dataclasses.dataclass
class Database:
    header: dict
    protein_sizes: list
    proteins: list


def _f32(x) -> np.ndarray:
    if isinstance(x, msgpack.ExtType):
        if x.code == 8:  # legacy: big-endian f32
            return np.frombuffer(x.data, dtype=">f4").astype(np.float32)
        raise ValueError(f"unexpected ext type {x.code} for f32 array")
    if isinstance(x, (bytes, bytearray)):  # current writer: c-core/write.c:59-66
        return np.frombuffer(x, dtype="<f4").astype(np.float32)
    raise ValueError(f"unexpected f32 array encoding {type(x)}")


def _sizes(x) -> list:
    if isinstance(x, msgpack.ExtType):
        if x.code == 6:  # legacy: big-endian u32
            return [int(v) for v in np.frombuffer(x.data, dtype=">u4")]
        raise ValueError(f"unexpected ext type {x.code} for protein_sizes")
    return [int(v) for v in x]


def read_dcp(path: str) -> Database:
    with open(path, "rb") as f:
        raw = f.read()
    # The "nodes" map repeats its three keys K+1 times, so maps are kept as
    # pair lists instead of dicts.
    top = msgpack.unpackb(raw, raw=False, strict_map_key=False, object_pairs_hook=list)
    assert [k for k, _ in top] == ["header", "proteins"], "not a deciphon database"
    header_pairs, proteins_raw = top[0][1], top[1][1]
    header = {}
    for k, v in header_pairs:
        header[k] = v
    if header["magic_number"] != 0xC6F1:  # c-core/magic_number.h:4
        raise ValueError("not a database file")
    if header["version"] != 1:  # c-core/database_version.h:4
        raise ValueError("unsupported database version")
    sizes = _sizes(header["protein_sizes"])

    proteins = []
    for pairs in proteins_raw:
        keys = [k for k, _ in pairs]
        assert keys == [
            "accession", "gencode", "consensus", "core_size", "null_nuclt_dist",
            "null_emission", "bg_nuclt_dist", "bg_emission", "nodes", "BMk",
        ], keys
        d = dict(pairs)
        K = int(d["core_size"])
        nodes = d["nodes"]
        assert len(nodes) == (K + 1) * 3
        trans = np.empty((K + 1, TRANS_SIZE), dtype=np.float32)
        emission = np.empty((K + 1, TABLE_SIZE), dtype=np.float32)
        nucltp = np.empty((K + 3, 4), dtype=np.float32)
        codonm = np.empty((K + 3, 125), dtype=np.float32)
        for j, key in enumerate(("null_nuclt_dist", "bg_nuclt_dist")):
            nucltp[j], codonm[j] = _f32(d[key][0]), _f32(d[key][1])
        for i in range(K + 1):
            (k0, nd), (k1, t), (k2, e) = nodes[3 * i : 3 * i + 3]
            assert (k0, k1, k2) == ("nuclt_dist", "trans", "emission")
            trans[i] = _f32(t)
            emission[i] = _f32(e)
            nucltp[2 + i], codonm[2 + i] = _f32(nd[0]), _f32(nd[1])
        proteins.append(
            Protein(
                accession=d["accession"],
                gencode=int(d["gencode"]),
                consensus=d["consensus"],
                core_size=K,
                null_emission=_f32(d["null_emission"]),
                bg_emission=_f32(d["bg_emission"]),
                trans=trans,
                emission=emission,
                BMk=_f32(d["BMk"]),
                nucltp=nucltp,
                codonm=codonm,
            )
        )
    assert len(proteins) == len(sizes)
    return Database(header=header, protein_sizes=sizes, proteins=proteins)
