"""ctypes binding of include/deciphon_hip.h (the batched operator ABI)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TABLE_SIZE = 1364
NUM_TRANS = 8
NUM_XTRANS = 13


class HipError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        self.code = code
        msg = error_string(code)
        super().__init__(f"deciphon error {code}: {msg}" + (f" ({detail})" if detail else ""))


class Window(C.Structure):
    """struct dcp_hip_window: [start, stop) of sequence `seq` against `profile`."""

    _fields_ = [("profile", C.c_int32), ("seq", C.c_int32), ("start", C.c_int32), ("stop", C.c_int32)]


def library_path() -> str:
    # DECIPHON_HIP_LIBDIR: a build of the same library elsewhere (kernel experiments: make OUT=<dir> ...)
    return os.path.join(os.environ.get("DECIPHON_HIP_LIBDIR") or os.path.join(_HERE, "lib"), "libdeciphon_hip.so")


def load_library() -> C.CDLL:
    """Loads the HIP library; raises (never falls back) when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or make -C deciphon_amd/csrc); there is no CPU fallback")
    L = C.CDLL(path)
    vp, i32, f32p = C.c_void_p, C.c_int, C.POINTER(C.c_float)
    L.dcp_hip_device_count.restype = i32
    L.dcp_hip_new.argtypes = [i32]
    L.dcp_hip_new.restype = vp
    L.dcp_hip_del.argtypes = [vp]
    L.dcp_hip_del.restype = None
    L.dcp_hip_strerror.argtypes = [vp]
    L.dcp_hip_strerror.restype = C.c_char_p
    L.dcp_hip_add_profile.argtypes = [vp, i32, vp, vp, vp, vp, C.POINTER(i32)]
    L.dcp_hip_add_protein.argtypes = [vp, i32, vp, vp, vp, vp, vp, C.POINTER(i32)]
    L.dcp_hip_load_dcp.argtypes = [vp, C.c_char_p, i32, i32]
    L.dcp_hip_num_profiles.argtypes = [vp]
    L.dcp_hip_load_chunks.argtypes = [vp]
    L.dcp_hip_pool_bytes.argtypes = [vp]
    L.dcp_hip_pool_bytes.restype = C.c_int64
    L.dcp_hip_profile_core_size.argtypes = [vp, i32]
    L.dcp_hip_profile_accession.argtypes = [vp, i32]
    L.dcp_hip_profile_accession.restype = C.c_char_p
    L.dcp_hip_commit_profiles.argtypes = [vp]
    L.dcp_hip_clear_profiles.argtypes = [vp]
    L.dcp_hip_clear_profiles.restype = None
    L.dcp_hip_encode.argtypes = [C.c_char_p, C.c_int64, vp]
    L.dcp_hip_set_sequences.argtypes = [vp, i32, vp, vp]
    L.dcp_hip_set_mode.argtypes = [vp, i32, i32]
    L.dcp_hip_set_xtrans_table.argtypes = [vp, i32, vp]
    L.dcp_hip_xtrans.argtypes = [i32, i32, i32, vp]
    L.dcp_hip_xtrans.restype = None
    L.dcp_hip_cost.argtypes = [vp, i32, vp, vp, vp]
    L.dcp_hip_cost_hits.argtypes = [vp, i32, vp, C.POINTER(i32), vp, vp]
    L.dcp_hip_cost_hits_begin.argtypes = [vp, i32, vp]
    L.dcp_hip_cost_hits_end.argtypes = [vp, C.POINTER(i32), vp, vp]
    L.dcp_hip_cost_bench.argtypes = [vp, i32, vp, i32, i32, f32p, C.POINTER(C.c_double), vp, vp]
    L.dcp_hip_stage.argtypes = [vp, i32, vp]
    L.dcp_hip_run_staged.argtypes = [vp, i32, f32p, C.POINTER(C.c_double)]
    L.dcp_hip_fetch_staged.argtypes = [vp, vp, vp]
    L.dcp_hip_path.argtypes = [vp, i32, vp]
    L.dcp_hip_path_redone.argtypes = [vp]
    L.dcp_hip_path_reserve.argtypes = [vp, C.c_int64]
    L.dcp_hip_path_nsteps.argtypes = [vp, i32]
    L.dcp_hip_path_steps.argtypes = [vp, i32, vp, vp]
    L.dcp_hip_path_steps_packed.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(C.c_int32)]
    L.dcp_hip_path_trellis.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp)]
    L.dcp_hip_path_score.argtypes = [vp, i32]
    L.dcp_hip_path_score.restype = C.c_float
    L.dcp_error_string.argtypes = [i32]
    L.dcp_error_string.restype = C.c_char_p
    _LIB = L
    return L


def error_string(code: int) -> str:
    try:
        s = load_library().dcp_error_string(int(code))
    except ImportError:
        return "?"
    return s.decode() if s else ""


def device_count() -> int:
    return int(load_library().dcp_hip_device_count())


def encode(data: str) -> np.ndarray:
    """dcp_batch_add's normalisation of one nucleotide string -> indices 0..3."""
    raw = data.encode()
    out = np.zeros(len(raw), dtype=np.uint8)
    rc = load_library().dcp_hip_encode(raw, len(raw), out.ctypes.data_as(C.c_void_p))
    if rc:
        raise HipError(rc)
    return out


def xtrans(seq_size: int, multi_hits: bool, hmmer3_compat: bool) -> np.ndarray:
    out = np.zeros(NUM_XTRANS, dtype=np.float32)
    load_library().dcp_hip_xtrans(int(seq_size), int(multi_hits), int(hmmer3_compat), out.ctypes.data_as(C.c_void_p))
    return out


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class Engine:
    """One GPU's worth of resident profiles and reads (struct dcp_hip)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        self.h = self.lib.dcp_hip_new(int(device))
        if not self.h:
            raise HipError(8, f"no usable HIP device {device}; there is no CPU fallback")
        self.device = int(device)
        self._seq_lens = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.dcp_hip_del(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc:
            raise HipError(rc, self.lib.dcp_hip_strerror(self.h).decode())

    # ---- profiles -------------------------------------------------------------
    def add_profile(self, K: int, trans, match, null, bg) -> int:
        trans, match, null, bg = _f32(trans), _f32(match), _f32(null), _f32(bg)
        assert trans.shape == (NUM_TRANS, K) and match.shape == (TABLE_SIZE, K)
        assert null.shape == (TABLE_SIZE,) and bg.shape == (TABLE_SIZE,)
        idx = C.c_int(-1)
        self._check(self.lib.dcp_hip_add_profile(self.h, K, _p(trans), _p(match), _p(null), _p(bg), C.byref(idx)))
        return idx.value

    def add_protein(self, K: int, node_trans, node_emission, BMk, null_lprob, bg_lprob) -> int:
        a = [_f32(v) for v in (node_trans, node_emission, BMk, null_lprob, bg_lprob)]
        assert a[0].shape == (K + 1, 7) and a[1].shape == (K + 1, TABLE_SIZE) and a[2].shape == (K,)
        idx = C.c_int(-1)
        self._check(self.lib.dcp_hip_add_protein(self.h, K, *[_p(v) for v in a], C.byref(idx)))
        return idx.value

    def load_dcp(self, path: str, first: int = 0, count: int = -1) -> None:
        self._check(self.lib.dcp_hip_load_dcp(self.h, os.fsencode(path), first, count))

    @property
    def load_chunks(self) -> int:
        """Staging chunks the last load_dcp went through."""
        return self.lib.dcp_hip_load_chunks(self.h)

    @property
    def pool_bytes(self) -> int:
        """HBM bytes of the resident profile tables."""
        return int(self.lib.dcp_hip_pool_bytes(self.h))

    @property
    def num_profiles(self) -> int:
        return self.lib.dcp_hip_num_profiles(self.h)

    def core_size(self, i: int) -> int:
        return self.lib.dcp_hip_profile_core_size(self.h, i)

    def accession(self, i: int) -> str:
        s = self.lib.dcp_hip_profile_accession(self.h, i)
        return s.decode() if s else ""

    def commit(self) -> None:
        self._check(self.lib.dcp_hip_commit_profiles(self.h))

    def clear_profiles(self) -> None:
        self.lib.dcp_hip_clear_profiles(self.h)

    # ---- sequences ------------------------------------------------------------
    def set_sequences(self, seqs) -> None:
        """seqs: list of uint8 arrays of nucleotide indices 0..3."""
        seqs = [np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
        off = np.zeros(len(seqs) + 1, dtype=np.int64)
        np.cumsum([len(s) for s in seqs], out=off[1:])
        nt = np.concatenate(seqs) if seqs else np.zeros(0, dtype=np.uint8)
        nt = np.ascontiguousarray(nt)
        self._check(self.lib.dcp_hip_set_sequences(self.h, len(seqs), _p(nt), _p(off)))
        self._seq_lens = [len(s) for s in seqs]

    def set_mode(self, multi_hits: bool = True, hmmer3_compat: bool = False) -> None:
        self._check(self.lib.dcp_hip_set_mode(self.h, int(multi_hits), int(hmmer3_compat)))

    def set_xtrans_table(self, table) -> None:
        """table[rows][13]: caller-supplied special transitions for amino lengths 0..rows-1."""
        t = _f32(table).reshape(-1, NUM_XTRANS)
        self._check(self.lib.dcp_hip_set_xtrans_table(self.h, t.shape[0], _p(t)))

    # ---- the DP ---------------------------------------------------------------
    @staticmethod
    def _windows(windows):
        if isinstance(windows, np.ndarray):  # int32 [n][4] = (profile, seq, start, stop): no conversion
            a = np.ascontiguousarray(windows, dtype=np.int32).reshape(-1, 4)
            ptr = _p(a)
            ptr._keep = a
            return len(a), ptr
        n = len(windows)
        arr = (Window * max(n, 1))()
        for i, w in enumerate(windows):
            arr[i] = w if isinstance(w, Window) else Window(*w)
        return n, arr

    def cost(self, windows):
        """-> (null_cost[n], alt_cost[n]): viterbi_null / viterbi_cost of every window."""
        n, arr = self._windows(windows)
        nul = np.zeros(n, dtype=np.float32)
        alt = np.zeros(n, dtype=np.float32)
        self._check(self.lib.dcp_hip_cost(self.h, n, arr, _p(nul), _p(alt)))
        return nul, alt

    def cost_hits(self, windows):
        """-> (indices, lrt) of the windows whose lrt is finite and >= 0: the cost pass followed by
        process_window's filter (c-core/thread.c:118-121), both on the device."""
        n, arr = self._windows(windows)
        idx = np.zeros(max(n, 1), dtype=np.int32)
        lrt = np.zeros(max(n, 1), dtype=np.float32)
        nh = C.c_int(0)
        self._check(self.lib.dcp_hip_cost_hits(self.h, n, arr, C.byref(nh), _p(idx), _p(lrt)))
        return idx[: nh.value].copy(), lrt[: nh.value].copy()

    def cost_hits_begin(self, windows) -> None:
        """The first half of cost_hits: stages the windows, enqueues the kernels and returns while the GPU works.  Up to
        two batches may be outstanding; cost_hits_end() delivers the oldest."""
        n, arr = self._windows(windows)
        self._check(self.lib.dcp_hip_cost_hits_begin(self.h, n, arr))
        self._pending = getattr(self, "_pending", []) + [n]

    def cost_hits_end(self):
        n = self._pending.pop(0) if getattr(self, "_pending", None) else 0
        idx = np.zeros(max(n, 1), dtype=np.int32)
        lrt = np.zeros(max(n, 1), dtype=np.float32)
        nh = C.c_int(0)
        self._check(self.lib.dcp_hip_cost_hits_end(self.h, C.byref(nh), _p(idx), _p(lrt)))
        return idx[: nh.value].copy(), lrt[: nh.value].copy()

    def cost_bench(self, windows, warmup: int, reps: int):
        """-> (ms per launch, cells per launch, null_cost, alt_cost), timed with HIP events."""
        n, arr = self._windows(windows)
        nul = np.zeros(n, dtype=np.float32)
        alt = np.zeros(n, dtype=np.float32)
        ms = C.c_float(0)
        cells = C.c_double(0)
        self._check(self.lib.dcp_hip_cost_bench(self.h, n, arr, warmup, reps, C.byref(ms), C.byref(cells), _p(nul),
                                                _p(alt)))
        return ms.value, cells.value, nul, alt

    def stage(self, windows) -> None:
        """Copies the window list to HBM for run_staged() (measurement)."""
        n, arr = self._windows(windows)
        self._staged_n = n
        self._check(self.lib.dcp_hip_stage(self.h, n, arr))

    def run_staged(self, reps: int):
        """-> (HIP-event ms of `reps` cost-pass launches together, DP cells of one launch)."""
        ms = C.c_float(0)
        cells = C.c_double(0)
        self._check(self.lib.dcp_hip_run_staged(self.h, reps, C.byref(ms), C.byref(cells)))
        return ms.value, cells.value

    def fetch_staged(self):
        nul = np.zeros(self._staged_n, dtype=np.float32)
        alt = np.zeros(self._staged_n, dtype=np.float32)
        self._check(self.lib.dcp_hip_fetch_staged(self.h, _p(nul), _p(alt)))
        return nul, alt

    def path_reserve(self, nbytes: int):
        """Set HBM aside for the path pass's DP tables now (its clearing overlaps what follows)."""
        self._check(self.lib.dcp_hip_path_reserve(self.h, int(nbytes)))

    def path_steps_packed(self, i: int) -> np.ndarray:
        """Window i's steps of the last path() as the engine holds them (state id | emission length << 16), copied."""
        p, n = C.c_void_p(), C.c_int32(0)
        self._check(self.lib.dcp_hip_path_steps_packed(self.h, i, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n.value,)).copy()

    def path(self, windows, trellis: bool = True):
        """viterbi_path + trellis_unzip -> list of dicts(score, state_ids, seqsizes[, xnodes, nodes]).
        The steps are read first (they come from the fast pass); trellis=True then asks for the
        packed back-pointers too, which makes the library run the literal pass for the batch."""
        n, arr = self._windows(windows)
        self._check(self.lib.dcp_hip_path(self.h, n, arr))
        self.path_redone = self.lib.dcp_hip_path_redone(self.h)
        out = []
        for i in range(n):
            ns = self.lib.dcp_hip_path_nsteps(self.h, i)
            ids = np.zeros(ns, dtype=np.int32)
            sizes = np.zeros(ns, dtype=np.int32)
            self._check(self.lib.dcp_hip_path_steps(self.h, i, _p(ids), _p(sizes)))
            out.append(dict(score=np.float32(self.lib.dcp_hip_path_score(self.h, i)), state_ids=ids, seqsizes=sizes))
        if not trellis:
            return out
        for i in range(n):
            xn, nd = C.c_void_p(), C.c_void_p()
            self._check(self.lib.dcp_hip_path_trellis(self.h, i, C.byref(xn), C.byref(nd)))
            if isinstance(windows, np.ndarray):  # arr is then a bare pointer to the int32 [n][4] array
                prof, _, start, stop = (int(v) for v in np.asarray(windows, dtype=np.int32).reshape(-1, 4)[i])
            else:
                prof, start, stop = arr[i].profile, arr[i].start, arr[i].stop
            L = stop - start
            K = self.core_size(prof)
            out[i]["xnodes"] = np.ctypeslib.as_array(C.cast(xn, C.POINTER(C.c_uint32)), shape=(L + 1,)).copy()
            out[i]["nodes"] = np.ctypeslib.as_array(C.cast(nd, C.POINTER(C.c_uint16)), shape=((L + 1) * K,)).copy()
            # the literal pass has replaced the steps: they must be the very same path
            ns = self.lib.dcp_hip_path_nsteps(self.h, i)
            ids = np.zeros(ns, dtype=np.int32)
            sizes = np.zeros(ns, dtype=np.int32)
            self._check(self.lib.dcp_hip_path_steps(self.h, i, _p(ids), _p(sizes)))
            out[i]["literal_state_ids"], out[i]["literal_seqsizes"] = ids, sizes
            out[i]["literal_score"] = np.float32(self.lib.dcp_hip_path_score(self.h, i))
        return out
