"""ctypes binding of include/deciphon_host.h: the .dcp reader and the scalar
bookkeeping of process_window (windows, trellis_unzip, hit spans).  No GPU needed."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .hip import HipError, load_library

TABLE_SIZE = 1364


class _Window(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("core_size", "seq_size", "start", "stop", "idx", "last_hit_pos")]


def _lib():
    L = load_library()
    if getattr(L, "_host_ready", False):
        return L
    vp, i32 = C.c_void_p, C.c_int
    L.dcp_db_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.dcp_db_close.argtypes = [vp]
    L.dcp_db_close.restype = None
    L.dcp_db_num_proteins.argtypes = [vp]
    L.dcp_db_epsilon.argtypes = [vp]
    L.dcp_db_epsilon.restype = C.c_float
    L.dcp_db_entry_dist.argtypes = [vp]
    L.dcp_db_has_ga.argtypes = [vp]
    L.dcp_db_protein_offset.argtypes = [vp, i32]
    L.dcp_db_protein_offset.restype = C.c_int64
    L.dcp_db_protein_core_size.argtypes = [vp, i32, C.POINTER(i32)]
    L.dcp_db_read_protein.argtypes = [vp, i32, vp, vp, vp, vp, vp, C.c_char_p, C.c_char_p]
    L.dcp_db_read_nuclt_dist.argtypes = [vp, i32, vp, vp, C.POINTER(i32)]
    L.dcp_decode_quasi_codon.argtypes = [C.c_float, vp, vp, vp, i32, vp]
    L.dcp_gencode_amino_of.argtypes = [i32, vp]
    L.dcp_gencode_amino_of.restype = C.c_char
    L.dcp_db_core_sizes.argtypes = [vp, vp]
    L.dcp_db_partition_bounds.argtypes = [vp, i32, i32, vp]
    L.dcp_partition_bounds_of.argtypes = [i32, vp, i32, i32, vp]
    L.dcp_partition_size.argtypes = [C.c_long, C.c_long, C.c_long]
    L.dcp_partition_size.restype = C.c_long
    L.dcp_window_setup.argtypes = [C.POINTER(_Window), i32, i32]
    L.dcp_window_setup.restype = None
    L.dcp_window_next.argtypes = [C.POINTER(_Window)]
    L.dcp_trellis_unzip.argtypes = [i32, i32, vp, vp, i32, vp, vp, C.POINTER(i32)]
    L.dcp_path_hit.argtypes = [i32, vp, vp, vp]
    L.dcp_state_name_of.argtypes = [i32, C.c_char_p]
    L.dcp_state_name_of.restype = None
    L.dcp_state_is_mute_id.argtypes = [i32]
    L.dcp_lrt_of.argtypes = [C.c_float, C.c_float]
    L.dcp_lrt_of.restype = C.c_float
    L._host_ready = True
    return L


class Database:
    """A pressed .dcp file (replaces database_reader + protein_reader + protein_unpack)."""

    def __init__(self, path: str):
        self.lib = _lib()
        h = C.c_void_p()
        rc = self.lib.dcp_db_open(os.fsencode(path), C.byref(h))
        if rc:
            raise HipError(rc, path)
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.dcp_db_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.lib.dcp_db_num_proteins(self.h)

    @property
    def epsilon(self) -> float:
        return float(self.lib.dcp_db_epsilon(self.h))

    @property
    def entry_dist(self) -> int:
        return self.lib.dcp_db_entry_dist(self.h)

    @property
    def has_ga(self) -> bool:
        return bool(self.lib.dcp_db_has_ga(self.h))

    def offset(self, i: int) -> int:
        return int(self.lib.dcp_db_protein_offset(self.h, i))

    def core_sizes(self) -> np.ndarray:
        K = np.zeros(len(self), np.int32)
        if rc := self.lib.dcp_db_core_sizes(self.h, K.ctypes.data_as(C.c_void_p)):
            raise HipError(rc)
        return K

    def partition_bounds(self, nparts: int, balanced: bool = False) -> np.ndarray:
        """first[nparts + 1]: partition p = proteins first[p] .. first[p+1]-1 (c-core/protein_reader.c:112-128,
        or boundaries that balance the sum of core sizes)."""
        first = np.zeros(nparts + 1, np.int32)
        if rc := self.lib.dcp_db_partition_bounds(self.h, nparts, int(balanced), first.ctypes.data_as(C.c_void_p)):
            raise HipError(rc)
        return first

    def protein(self, i: int) -> dict:
        k = C.c_int(0)
        rc = self.lib.dcp_db_protein_core_size(self.h, i, C.byref(k))
        if rc:
            raise HipError(rc)
        K = k.value
        trans = np.zeros((K + 1, 7), np.float32)
        emission = np.zeros((K + 1, TABLE_SIZE), np.float32)
        BMk = np.zeros(K, np.float32)
        null = np.zeros(TABLE_SIZE, np.float32)
        bg = np.zeros(TABLE_SIZE, np.float32)
        acc = C.create_string_buffer(32)
        cons = C.create_string_buffer(K + 1)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        rc = self.lib.dcp_db_read_protein(self.h, i, p(trans), p(emission), p(BMk), p(null), p(bg), acc, cons)
        if rc:
            raise HipError(rc)
        nucltp = np.zeros((K + 3, 4), np.float32)
        codonm = np.zeros((K + 3, 125), np.float32)
        gencode = C.c_int(0)
        rc = self.lib.dcp_db_read_nuclt_dist(self.h, i, p(nucltp), p(codonm), C.byref(gencode))
        if rc:
            raise HipError(rc)
        # nucltp / codonm: entry 0 = null model, 1 = background, 2 + n = node n (what decoder_setup takes)
        return dict(core_size=K, accession=acc.value.decode(), consensus=cons.value.decode(), trans=trans,
                    emission=emission, BMk=BMk, null_emission=null, bg_emission=bg, gencode=gencode.value,
                    nucltp=nucltp, codonm=codonm)


def partition_bounds(core_sizes, nparts: int, balanced: bool = False) -> np.ndarray:
    """first[nparts + 1] for profiles of the given core sizes (see Database.partition_bounds)."""
    K = np.ascontiguousarray(core_sizes, np.int32)
    first = np.zeros(nparts + 1, np.int32)
    if rc := _lib().dcp_partition_bounds_of(len(K), K.ctypes.data_as(C.c_void_p), nparts, int(balanced),
                                            first.ctypes.data_as(C.c_void_p)):
        raise HipError(rc)
    return first


def partition_size(nelems: int, nparts: int, idx: int) -> int:
    return int(_lib().dcp_partition_size(nelems, nparts, idx))


class WindowIter:
    """window_setup / window_next / window_set_last_hit_position (c-core/window.c)."""

    def __init__(self, seq_size: int, core_size: int):
        self.lib = _lib()
        self.w = _Window()
        self.lib.dcp_window_setup(C.byref(self.w), seq_size, core_size)

    def next(self):
        if not self.lib.dcp_window_next(C.byref(self.w)):
            return None
        return self.w.idx, self.w.start, self.w.stop

    def set_last_hit_position(self, pos: int) -> None:
        self.w.last_hit_pos = pos


def unzip(K: int, L: int, xnodes: np.ndarray, nodes: np.ndarray):
    lib = _lib()
    xnodes = np.ascontiguousarray(xnodes, np.uint32)
    nodes = np.ascontiguousarray(nodes, np.uint16)
    cap = 2 * L + 2 * K + 16
    while True:
        ids = np.zeros(cap, np.int32)
        sizes = np.zeros(cap, np.int32)
        n = C.c_int(0)
        rc = lib.dcp_trellis_unzip(K, L, xnodes.ctypes.data_as(C.c_void_p), nodes.ctypes.data_as(C.c_void_p), cap,
                                   ids.ctypes.data_as(C.c_void_p), sizes.ctypes.data_as(C.c_void_p), C.byref(n))
        if rc == 20 and n.value > cap:  # DCP_ENOMEM: buffer too small, size is known now
            cap = n.value
            continue
        if rc:
            raise HipError(rc)
        return ids[: n.value].copy(), sizes[: n.value].copy()


def path_hit(state_ids, seqsizes):
    """-> (hit_start, hit_stop, begin_step, end_step, last_hit_pos) or None (c-core/thread.c:130-166)."""
    ids = np.ascontiguousarray(state_ids, np.int32)
    sizes = np.ascontiguousarray(seqsizes, np.int32)
    out = np.zeros(5, np.int32)
    ok = _lib().dcp_path_hit(len(ids), ids.ctypes.data_as(C.c_void_p), sizes.ctypes.data_as(C.c_void_p),
                             out.ctypes.data_as(C.c_void_p))
    return tuple(int(v) for v in out) if ok else None


def state_name(state_id: int) -> str:
    buf = C.create_string_buffer(8)
    _lib().dcp_state_name_of(int(state_id), buf)
    return buf.value.decode()


def state_is_mute(state_id: int) -> bool:
    return bool(_lib().dcp_state_is_mute_id(int(state_id)))


def lrt(null_loglik, alt_loglik) -> np.float32:
    return np.float32(_lib().dcp_lrt_of(float(null_loglik), float(alt_loglik)))


def decode_quasi_codon(epsilon: float, nucltp, codonm, nt) -> np.ndarray:
    """decoder_decode (c-core/decoder.c:38-58): codon (3 nucleotide indices) behind the 1..5 nucleotides nt."""
    a = np.ascontiguousarray(nucltp, np.float32)
    b = np.ascontiguousarray(codonm, np.float32)
    z = np.ascontiguousarray(nt, np.uint8)
    out = np.zeros(3, np.uint8)
    rc = _lib().dcp_decode_quasi_codon(float(epsilon), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                       z.ctypes.data_as(C.c_void_p), len(z), out.ctypes.data_as(C.c_void_p))
    if rc:
        raise HipError(rc)
    return out


def gencode_amino(gencode_id: int, codon) -> str:
    c = np.ascontiguousarray(codon, np.uint8)
    r = _lib().dcp_gencode_amino_of(int(gencode_id), c.ctypes.data_as(C.c_void_p))
    return r.decode() if r != b"\0" else ""
