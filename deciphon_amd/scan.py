"""Python mirror of python-core's wrapper classes over the C API of include/deciphon.h:
`Scan` (python-core/deciphon_core/scan.py:23-77), `Batch` (batch.py:9-30), `Sequence`
(sequence.py) and `DeciphonError` (error.py), over ctypes instead of CFFI."""
from __future__ import annotations

import ctypes as C
import dataclasses
import os

from .hip import HipError as DeciphonError
from .hip import load_library

__all__ = ["Scan", "Batch", "Sequence", "DeciphonError"]

_CALLBACK = C.CFUNCTYPE(None, C.c_void_p)


def _lib():
    L = load_library()
    if getattr(L, "_scan_ready", False):
        return L
    vp, i32 = C.c_void_p, C.c_int
    L.dcp_scan_new.restype = vp
    L.dcp_scan_del.argtypes = [vp]
    L.dcp_scan_del.restype = None
    L.dcp_scan_setup.argtypes = [vp, C.c_char_p, i32, i32, C.c_bool, C.c_bool, C.c_bool, _CALLBACK, vp]
    L.dcp_scan_setup_partition.argtypes = [vp, C.c_char_p, i32, i32, i32, C.c_bool, C.c_bool, _CALLBACK, vp]
    L.dcp_scan_setup_partition_balanced.argtypes = L.dcp_scan_setup_partition.argtypes
    L.dcp_scan_partition_range.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.dcp_scan_run.argtypes = [vp, vp, C.c_char_p]
    L.dcp_scan_interrupt.argtypes = [vp]
    L.dcp_scan_interrupt.restype = None
    L.dcp_scan_progress.argtypes = [vp]
    L.dcp_scan_num_products.argtypes = [vp]
    L.dcp_scan_num_products.restype = C.c_long
    L.dcp_scan_product.argtypes = [vp, C.c_long]
    L.dcp_scan_product.restype = C.c_char_p
    L.dcp_scan_last_timing.argtypes = [vp, C.POINTER(C.c_double), i32]
    L.dcp_batch_new.restype = vp
    L.dcp_batch_del.argtypes = [vp]
    L.dcp_batch_del.restype = None
    L.dcp_batch_add.argtypes = [vp, C.c_long, C.c_char_p, C.c_char_p]
    L.dcp_batch_reset.argtypes = [vp]
    L.dcp_batch_reset.restype = None
    L._scan_ready = True
    return L


@dataclasses.dataclass
class Sequence:
    id: int
    name: str
    data: str


class Batch:
    def __init__(self):
        self._lib = _lib()
        self._cbatch = self._lib.dcp_batch_new()
        if not self._cbatch:
            raise MemoryError()

    def add(self, sequence: Sequence):
        if rc := self._lib.dcp_batch_add(self._cbatch, sequence.id, sequence.name.encode(), sequence.data.encode()):
            raise DeciphonError(rc)

    def reset(self):
        self._lib.dcp_batch_reset(self._cbatch)

    @property
    def cdata(self):
        return self._cbatch

    def __del__(self):
        if getattr(self, "_cbatch", None):
            self._lib.dcp_batch_del(self._cbatch)
            self._cbatch = None


class Scan:
    """Same constructor and methods as python-core's Scan.  `partition=(device, index, nparts)`
    (not in the reference) makes this scan own one contiguous profile partition on one GPU;
    `balanced=True` puts the partition boundaries where they balance the sum of core sizes."""

    def __init__(self, dbfile, port: int = 0, num_threads: int = 1, multi_hits: bool = True,
                 hmmer3_compat: bool = False, cache: bool = False, partition=None, on_window=None,
                 balanced: bool = False, progress_callback: bool = True):
        self._lib = _lib()
        self._cscan = self._lib.dcp_scan_new()
        if not self._cscan:
            raise MemoryError()
        self.interrupted = False
        self._on_window = on_window

        def _cb(_userdata):
            if self._on_window is not None:
                try:
                    self._on_window()
                except BaseException:
                    self.interrupt()  # python-core/deciphon_core/scan.py:12-15
                    raise

        # python-core always hands the library a callback (an empty function: it lets Python raise KeyboardInterrupt
        # between windows); progress_callback=False hands it none, as a C caller may (c-core/test_scan.c:36-41)
        self._cb = _CALLBACK(_cb) if progress_callback else C.cast(None, _CALLBACK)
        path = os.fsencode(getattr(dbfile, "path", dbfile))
        if partition is None:
            rc = self._lib.dcp_scan_setup(self._cscan, path, port, num_threads, multi_hits, hmmer3_compat, cache,
                                          self._cb, None)
        else:
            device, index, nparts = partition
            setup = self._lib.dcp_scan_setup_partition_balanced if balanced else self._lib.dcp_scan_setup_partition
            rc = setup(self._cscan, path, device, index, nparts, multi_hits, hmmer3_compat, self._cb, None)
        if rc:
            self.free()
            raise DeciphonError(rc)

    def run(self, snap, batch: Batch):
        self.interrupted = False
        basedir = getattr(snap, "basedir", snap)
        if rc := self._lib.dcp_scan_run(self._cscan, batch.cdata, str(basedir).encode()):
            raise DeciphonError(rc)

    def partition_range(self):
        """(first profile, number of profiles) of the database this scan owns."""
        first, count = C.c_int(0), C.c_int(0)
        if rc := self._lib.dcp_scan_partition_range(self._cscan, C.byref(first), C.byref(count)):
            raise DeciphonError(rc)
        return first.value, count.value

    def products(self):
        n = self._lib.dcp_scan_num_products(self._cscan)
        return [self._lib.dcp_scan_product(self._cscan, i).decode() for i in range(n)]

    def last_timing(self) -> dict:
        """Where the wall time of the last run went (dcp_scan_last_timing): seconds per phase and counts."""
        keys = ("total_s", "reads_h2d_encode_s", "window_bookkeeping_s", "cost_pass_s", "path_pass_s", "rows_decode_s",
                "products_tsv_s", "rounds", "windows", "path_passes")
        buf = (C.c_double * len(keys))()
        n = self._lib.dcp_scan_last_timing(self._cscan, buf, len(keys))
        assert n == len(keys)
        return {k: (int(buf[i]) if i >= 7 else float(buf[i])) for i, k in enumerate(keys)}

    def interrupt(self):
        self.interrupted = True
        self._lib.dcp_scan_interrupt(self._cscan)

    def progress(self) -> int:
        return self._lib.dcp_scan_progress(self._cscan)

    def free(self):
        if getattr(self, "_cscan", None):
            self._lib.dcp_scan_del(self._cscan)
            self._cscan = None

    def __del__(self):
        self.free()

    def __enter__(self):
        return self

    def __exit__(self, *_):
        self.free()
