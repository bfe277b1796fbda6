"""GPU: the per-problem drop-in for c-core/viterbi.h (include/dcp_viterbi.h), driven the way
protein_setup_viterbi + process_window drive the reference: setters, then
viterbi_null / viterbi_cost / viterbi_path with a code callback, then the trellis."""
import ctypes as C

import numpy as np
import pytest

import deciphon_amd
from dcp_testlib import bits, random_seq, synth_profile

pytestmark = pytest.mark.gpu

CODE_FN = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.c_void_p)


class Trellis(C.Structure):
    _fields_ = [("core_size", C.c_int), ("xnodes", C.POINTER(C.c_uint32)), ("nodes", C.POINTER(C.c_uint16)),
                ("xnode", C.POINTER(C.c_uint32)), ("node", C.POINTER(C.c_uint16))]


@pytest.fixture(scope="module")
def lib():
    L = deciphon_amd.load_library()
    L.viterbi_new.restype = C.c_void_p
    L.viterbi_del.argtypes = [C.c_void_p]
    L.viterbi_setup.argtypes = [C.c_void_p, C.c_int]
    L.viterbi_set_extr_trans.argtypes = [C.c_void_p, C.c_int, C.c_float]
    L.viterbi_set_core_trans.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int]
    L.viterbi_set_null.argtypes = [C.c_void_p, C.c_float, C.c_int]
    L.viterbi_set_background.argtypes = [C.c_void_p, C.c_float, C.c_int]
    L.viterbi_set_match.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int]
    for f in (L.viterbi_null, L.viterbi_cost):
        f.argtypes = [C.c_void_p, C.c_int, CODE_FN, C.c_void_p]
        f.restype = C.c_float
    L.viterbi_path.argtypes = [C.c_void_p, C.c_int, CODE_FN, C.c_void_p]
    L.viterbi_trellis.argtypes = [C.c_void_p]
    L.viterbi_trellis.restype = C.POINTER(Trellis)
    return L


def load(lib, v, prof, xt):
    assert lib.viterbi_setup(v, prof.K) == 0
    for i in range(13):
        lib.viterbi_set_extr_trans(v, i, float(xt[i]))
    for tid in range(8):
        for k in range(prof.K):
            lib.viterbi_set_core_trans(v, tid, float(prof.trans[tid, k]), k)
    for c in range(1364):
        lib.viterbi_set_null(v, float(prof.null[c]), c)
        lib.viterbi_set_background(v, float(prof.bg[c]), c)
        for k in range(prof.K):
            lib.viterbi_set_match(v, float(prof.match[c, k]), k, c)


def test_setters_and_runs_match_oracle(lib, orc):
    assert lib.viterbi_table_size() == 1364
    v = lib.viterbi_new()
    assert v
    rng = np.random.default_rng(77)
    for it, K in enumerate((2, 7, 40)):
        quant = [None, 2.0, 1.0][it]
        prof = synth_profile(rng, K, quant)
        seq = random_seq(rng, 30 + it)
        xt = orc.xtrans(max(len(seq) // 3, 1), True, False)
        if quant:
            xt = (np.round(xt / quant) * quant).astype(np.float32)
        load(lib, v, prof, xt)
        fn = CODE_FN(lambda pos, n, _arg: orc.code(seq, pos, n))
        L = len(seq)
        assert bits(lib.viterbi_null(v, L, fn, None)) == bits(orc.null(prof, xt, seq))
        assert bits(lib.viterbi_cost(v, L, fn, None)) == bits(orc.cost(prof, xt, seq))
        assert lib.viterbi_path(v, L, fn, None) == 0
        tr = lib.viterbi_trellis(v).contents
        assert tr.core_size == K
        xn = np.ctypeslib.as_array(tr.xnodes, shape=(L + 1,))
        nd = np.ctypeslib.as_array(tr.nodes, shape=((L + 1) * K,))
        _, xo, no = orc.path(prof, xt, seq)
        assert np.array_equal(xn, xo) and np.array_equal(nd, no)
        # a second run on the same struct gives the same answer (no row-0 history)
        assert bits(lib.viterbi_cost(v, L, fn, None)) == bits(orc.cost(prof, xt, seq))
    # L = 0: R[0] = -RR and an untouched T (c-core/viterbi.c:599,703)
    assert lib.viterbi_null(v, 0, fn, None) == -float(xt[0])
    assert np.isinf(lib.viterbi_cost(v, 0, fn, None))
    # a callback that does not spell a sequence is refused
    bad = CODE_FN(lambda pos, n, _arg: 1363 if n == 5 else orc.code(seq, pos, n))
    assert np.isnan(lib.viterbi_cost(v, len(seq), bad, None))
    assert lib.viterbi_path(v, len(seq), bad, None) == 8  # DCP_EFUNCUSE
    lib.viterbi_del(v)
