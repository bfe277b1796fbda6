"""Synthetic pressed databases for tests and scripts: random proteins in log-probability
space (what protein_unpack yields) written in the CURRENT writer's encoding (`bin` + native
floats, int array of sizes: c-core/write.c:59-66, c-core/database_writer.c:76-193)."""
from __future__ import annotations

import msgpack
import numpy as np

from oracle.dcp_reader import Protein

TABLE_SIZE = 1364
AMINO = "ACDEFGHIKLMNPQRSTVWY"


def _lprobs(rng, shape, axis=-1, zero_frac=0.0):
    p = rng.random(shape).astype(np.float64) + 1e-3
    if zero_frac:
        p[rng.random(shape) < zero_frac] = 0.0
        p.reshape(-1)[0] = 0.5
    p = p / p.sum(axis=axis, keepdims=True)
    with np.errstate(divide="ignore"):
        return np.log(p).astype(np.float32)


def random_protein(rng, K: int, accession: str) -> Protein:
    """Emission tables normalised per code length; transitions normalised per state group
    (MM+MI+MD, IM+II, DM+DD), with the end-of-model structure of c-core/protein.c:148-160."""
    emission = np.empty((K + 1, TABLE_SIZE), np.float32)
    for lo, hi in ((0, 4), (4, 20), (20, 84), (84, 340), (340, 1364)):
        emission[:, lo:hi] = _lprobs(rng, (K + 1, hi - lo)) + np.float32(np.log(0.2))
    emission[K] = emission[K - 1]
    trans = np.empty((K + 1, 7), np.float32)
    trans[:, 0:3] = _lprobs(rng, (K + 1, 3))
    trans[:, 3:5] = _lprobs(rng, (K + 1, 2))
    trans[:, 5:7] = _lprobs(rng, (K + 1, 2))
    trans[K - 1 :, 2] = -np.inf  # MD of the last nodes
    trans[K - 1 :, 6] = -np.inf  # DD
    trans[K - 1 :, 5] = 0.0
    null = np.empty(TABLE_SIZE, np.float32)
    bg = np.empty(TABLE_SIZE, np.float32)
    for lo, hi in ((0, 4), (4, 20), (20, 84), (84, 340), (340, 1364)):
        null[lo:hi] = _lprobs(rng, (hi - lo,)) + np.float32(np.log(0.2))
        bg[lo:hi] = _lprobs(rng, (hi - lo,)) + np.float32(np.log(0.2))
    BMk = _lprobs(rng, (K,))
    consensus = "".join(rng.choice(list(AMINO), size=K))
    return Protein(accession, 1, consensus, K, null, bg, trans, emission, BMk)


def write_dcp(path: str, proteins, epsilon: float = 0.01) -> None:
    packer = msgpack.Packer(use_bin_type=True, use_single_float=True)

    def f32(a):
        return np.ascontiguousarray(a, "<f4").tobytes()

    nuclt = [f32(np.zeros(4, np.float32)), f32(np.zeros(125, np.float32))]
    blobs = []
    for p in proteins:
        K = p.core_size
        nodes = b"".join(
            packer.pack("nuclt_dist") + packer.pack(nuclt) + packer.pack("trans") + packer.pack(f32(p.trans[i]))
            + packer.pack("emission") + packer.pack(f32(p.emission[i])) for i in range(K + 1))
        blobs.append(
            packer.pack_map_header(10) + packer.pack("accession") + packer.pack(p.accession)
            + packer.pack("gencode") + packer.pack(p.gencode) + packer.pack("consensus") + packer.pack(p.consensus)
            + packer.pack("core_size") + packer.pack(K) + packer.pack("null_nuclt_dist") + packer.pack(nuclt)
            + packer.pack("null_emission") + packer.pack(f32(p.null_emission)) + packer.pack("bg_nuclt_dist")
            + packer.pack(nuclt) + packer.pack("bg_emission") + packer.pack(f32(p.bg_emission))
            + packer.pack("nodes") + packer.pack_map_header((K + 1) * 3) + nodes + packer.pack("BMk")
            + packer.pack(f32(p.BMk)))
    abc = {"symbols": "ACGT", "idx": b"\0" * 94, "any_symbol_id": 55, "typeid": 4}
    header = (packer.pack_map_header(8) + packer.pack("magic_number") + packer.pack(0xC6F1) + packer.pack("version")
              + packer.pack(1) + packer.pack("entry_dist") + packer.pack(2) + packer.pack("epsilon")
              + packer.pack(float(epsilon)) + packer.pack("abc") + packer.pack(abc) + packer.pack("amino")
              + packer.pack(dict(abc, symbols=AMINO, typeid=2)) + packer.pack("has_ga") + packer.pack(True)
              + packer.pack("protein_sizes") + packer.pack([len(b) for b in blobs]))
    with open(path, "wb") as f:
        f.write(packer.pack_map_header(2) + packer.pack("header") + header + packer.pack("proteins")
                + packer.pack_array_header(len(blobs)))
        for b in blobs:
            f.write(b)
