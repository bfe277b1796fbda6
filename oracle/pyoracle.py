"""TEST INFRASTRUCTURE ONLY -- ctypes bindings of the parity oracle.

`Oracle`  : oracle/libdcp_oracle.so, our scalar C restatement (dcp_oracle.c).
`RefLib`  : oracle/_ref/libdcp_ref.so, the reference's own c-core/viterbi.c
            compiled unmodified (oracle/Makefile `ref`), when present.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libdcp_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libdcp_ref.so")
REFERENCE_SRC = "/root/reference/c-core"

TABLE_SIZE = 1364
NUM_TRANS = 8
NUM_XTRANS = 13

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(dtype=np.uint16, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build(ref: bool = True) -> None:
    """Compile the oracle (always) and oracle/_ref (only where the reference tree exists)."""
    subprocess.run(["make", "-s", "-C", HERE], check=True)
    if ref and os.path.isdir(REFERENCE_SRC):
        subprocess.run(["make", "-s", "-C", HERE, "ref"], check=True)


@dataclasses.dataclass
class Profile:
    """A profile in DP-parameter space (costs), see dcp_oracle.h."""

    K: int
    trans: np.ndarray  # [8, K] BM MM MI MD IM II DM DD
    match: np.ndarray  # [1364, K]
    null: np.ndarray  # [1364]
    bg: np.ndarray  # [1364]
    accession: str = ""


class _Window(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("core_size", "seq_size", "start", "stop", "idx", "last_hit_pos")]


class Oracle:
    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            build(ref=False)
        L = self.lib = C.CDLL(path)
        L.orc_code.argtypes = [u8p, C.c_int, C.c_int]
        L.orc_xtrans.argtypes = [C.c_int, C.c_int, C.c_int, f32p]
        L.orc_setup_profile.argtypes = [C.c_int, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p]
        L.orc_null.argtypes = [f32p, C.c_float, u8p, C.c_int]
        L.orc_null.restype = C.c_float
        L.orc_cost.argtypes = [C.c_int, f32p, f32p, f32p, f32p, f32p, u8p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_cost.restype = C.c_float
        L.orc_unzip.argtypes = [C.c_int, C.c_int, u32p, u16p, i32p, i32p, C.c_int]
        L.orc_lrt.argtypes = [C.c_float, C.c_float]
        L.orc_lrt.restype = C.c_float
        L.orc_window_setup.argtypes = [C.c_int, C.c_int]
        L.orc_window_setup.restype = _Window
        L.orc_window_next.argtypes = [C.POINTER(_Window)]
        L.orc_hits.argtypes = [i32p, i32p, C.c_int, i32p, C.c_int, C.POINTER(C.c_int)]
        L.orc_encode.argtypes = [C.c_char_p, C.c_int, u8p]
        L.orc_state_name.argtypes = [C.c_int, C.c_char_p]
        L.orc_partition_size.argtypes = [C.c_long, C.c_long, C.c_long]
        L.orc_partition_size.restype = C.c_long

    def code(self, seq: np.ndarray, pos: int, length: int) -> int:
        return self.lib.orc_code(seq, pos, length)

    def xtrans(self, seq_size: int, multi_hits: bool, hmmer3_compat: bool) -> np.ndarray:
        xt = np.empty(NUM_XTRANS, dtype=np.float32)
        self.lib.orc_xtrans(seq_size, int(multi_hits), int(hmmer3_compat), xt)
        return xt

    def setup_profile(self, protein) -> Profile:
        """protein: oracle.dcp_reader.Protein (log-probs, node-major)."""
        K = protein.core_size
        trans = np.empty((NUM_TRANS, K), dtype=np.float32)
        match = np.empty((TABLE_SIZE, K), dtype=np.float32)
        null = np.empty(TABLE_SIZE, dtype=np.float32)
        bg = np.empty(TABLE_SIZE, dtype=np.float32)
        self.lib.orc_setup_profile(
            K, np.ascontiguousarray(protein.trans), np.ascontiguousarray(protein.emission),
            np.ascontiguousarray(protein.BMk), np.ascontiguousarray(protein.null_emission),
            np.ascontiguousarray(protein.bg_emission), trans, match, null, bg)
        return Profile(K, trans, match, null, bg, protein.accession)

    def null(self, prof: Profile, xt: np.ndarray, seq: np.ndarray) -> np.float32:
        return np.float32(self.lib.orc_null(prof.null, float(xt[0]), seq, len(seq)))

    def cost(self, prof: Profile, xt: np.ndarray, seq: np.ndarray, ref_lanes: int = 8) -> np.float32:
        return np.float32(self.lib.orc_cost(prof.K, prof.trans, prof.match, prof.null, prof.bg, xt, seq,
                                            len(seq), ref_lanes, None, None))

    def path(self, prof: Profile, xt: np.ndarray, seq: np.ndarray, ref_lanes: int = 8):
        L = len(seq)
        xnodes = np.zeros(L + 1, dtype=np.uint32)
        nodes = np.zeros((L + 1) * prof.K, dtype=np.uint16)
        score = self.lib.orc_cost(prof.K, prof.trans, prof.match, prof.null, prof.bg, xt, seq, L, ref_lanes,
                                  xnodes.ctypes.data_as(C.c_void_p), nodes.ctypes.data_as(C.c_void_p))
        return np.float32(score), xnodes, nodes

    def unzip(self, K: int, L: int, xnodes: np.ndarray, nodes: np.ndarray):
        # a path has at most L emitting steps, but every domain of a multi-hit path may run through up
        # to K mute delete states: (L + 1) * (K + 4) bounds it; start small and grow on overflow (-1)
        cap, limit = 2 * L + 2 * K + 64, (L + 1) * (K + 4) + 8
        while True:
            ids = np.empty(cap, dtype=np.int32)
            sizes = np.empty(cap, dtype=np.int32)
            n = self.lib.orc_unzip(K, L, xnodes, nodes, ids, sizes, cap)
            if n == -1 and cap < limit:
                cap = min(4 * cap, limit)
                continue
            if n < 0:
                raise RuntimeError(f"orc_unzip failed ({n})")
            return ids[:n].copy(), sizes[:n].copy()

    def lrt(self, null_loglik, alt_loglik) -> np.float32:
        return np.float32(self.lib.orc_lrt(float(null_loglik), float(alt_loglik)))

    def windows(self, seq_size: int, core_size: int, last_hit_positions=None):
        """Window ranges of c-core/window.c; `last_hit_positions(idx)` may return the
        position the previous window's hit reported (or None to leave it sticky)."""
        w = self.lib.orc_window_setup(seq_size, core_size)
        out = []
        while self.lib.orc_window_next(C.byref(w)):
            out.append((w.idx, w.start, w.stop))
            if last_hit_positions is not None:
                p = last_hit_positions(w.idx)
                if p is not None:
                    w.last_hit_pos = p
        return out

    def hits(self, ids: np.ndarray, sizes: np.ndarray):
        out = np.zeros(4, dtype=np.int32)
        last = C.c_int(-1)
        n = self.lib.orc_hits(np.ascontiguousarray(ids, dtype=np.int32),
                              np.ascontiguousarray(sizes, dtype=np.int32), len(ids), out, 1, C.byref(last))
        return (tuple(int(v) for v in out) if n else None), last.value

    def encode(self, data: str) -> np.ndarray:
        raw = data.encode()
        out = np.zeros(len(raw), dtype=np.uint8)
        rc = self.lib.orc_encode(raw, len(raw), out)
        if rc:
            raise ValueError(f"orc_encode error {rc}")
        return out

    def state_name(self, state_id: int) -> str:
        buf = C.create_string_buffer(16)
        self.lib.orc_state_name(int(state_id), buf)
        return buf.value.decode()

    def partition_size(self, nelems: int, nparts: int, idx: int) -> int:
        return int(self.lib.orc_partition_size(nelems, nparts, idx))


class RefLib:
    """The reference's own viterbi.c behind a flat-array driver (oracle/ref_glue.c)."""

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_SO)

    def __init__(self, path: str = REF_SO):
        L = self.lib = C.CDLL(path)
        L.ref_new.restype = C.c_void_p
        L.ref_del.argtypes = [C.c_void_p]
        L.ref_setup.argtypes = [C.c_void_p, C.c_int, f32p, f32p, f32p, f32p]
        L.ref_set_xtrans.argtypes = [C.c_void_p, f32p]
        L.ref_fresh.argtypes = [C.c_void_p]
        L.ref_null.argtypes = [C.c_void_p, u8p, C.c_int]
        L.ref_null.restype = C.c_float
        L.ref_cost.argtypes = [C.c_void_p, u8p, C.c_int]
        L.ref_cost.restype = C.c_float
        L.ref_path.argtypes = [C.c_void_p, u8p, C.c_int, u32p, u16p]
        L.ref_bench.argtypes = [C.c_int, f32p, f32p, f32p, f32p, f32p, u8p, i64p, C.c_int, C.c_int, C.c_int, f32p]
        L.ref_bench.restype = C.c_double
        self.h = C.c_void_p(L.ref_new())
        self.K = 0

    def __del__(self):
        try:
            self.lib.ref_del(self.h)
        except Exception:
            pass

    def setup(self, prof: Profile) -> None:
        self._prof = prof  # ref_glue.c borrows the arrays
        rc = self.lib.ref_setup(self.h, prof.K, prof.trans, prof.match, prof.null, prof.bg)
        assert rc == 0
        self.K = prof.K

    def null(self, xt: np.ndarray, seq: np.ndarray) -> np.float32:
        self.lib.ref_set_xtrans(self.h, xt)
        return np.float32(self.lib.ref_null(self.h, seq, len(seq)))

    def cost(self, xt: np.ndarray, seq: np.ndarray, fresh: bool = True) -> np.float32:
        """fresh=True: run on a just-set-up struct viterbi (history-free);
        fresh=False: the reference's real call sequence, row 0 keeps the previous run's last row."""
        self.lib.ref_set_xtrans(self.h, xt)
        if fresh:
            assert self.lib.ref_fresh(self.h) == 0
        return np.float32(self.lib.ref_cost(self.h, seq, len(seq)))

    def path(self, xt: np.ndarray, seq: np.ndarray, fresh: bool = True):
        L = len(seq)
        xnodes = np.zeros(L + 1, dtype=np.uint32)
        nodes = np.zeros((L + 1) * self.K, dtype=np.uint16)
        self.lib.ref_set_xtrans(self.h, xt)
        if fresh:
            assert self.lib.ref_fresh(self.h) == 0
        rc = self.lib.ref_path(self.h, seq, L, xnodes, nodes)
        assert rc == 0
        return xnodes, nodes

    def bench(self, prof: Profile, xts: np.ndarray, seqs: np.ndarray, offsets: np.ndarray, nthreads: int,
              repeat: int = 1):
        nprob = len(offsets) - 1
        out = np.zeros(2 * nprob, dtype=np.float32)
        secs = self.lib.ref_bench(prof.K, prof.trans, prof.match, prof.null, prof.bg,
                                  np.ascontiguousarray(xts, dtype=np.float32), seqs,
                                  np.ascontiguousarray(offsets, dtype=np.int64), nprob, nthreads, repeat, out)
        return secs, out.reshape(nprob, 2)
