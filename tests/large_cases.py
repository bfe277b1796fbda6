"""The long-profile parity cases: one list shared by tests/golden/make_golden.py (which runs the
reference's own viterbi.c on them, oracle/_ref), the CPU tests (oracle restatement against those
bits) and the -m gpu tests (HIP path against those bits).  Inputs are re-derived from the case
parameters alone, so only outputs are committed (tests/golden/large_classes.npz).

Every kernel class above one wavefront is covered at its boundary sizes: K = 257 ... 4096 (multi-wave
groups), 4097 ... 16383 (strip class), with continuous and quantised (tie-rich) tables on short
windows, real-structured tables (minifam nodes tiled) on 3 kb windows, and
K in {2048, 8192, 16383} on 10 kb reads."""
from __future__ import annotations

import os

import numpy as np

from dcp_testlib import GOLDEN, random_seq, synth_profile

LARGE_KS = (257, 384, 512, 768, 1024, 1536, 2048, 4096, 4097, 8192, 16383)


def large_cases():
    cases = []

    def add(**kw):
        kw["idx"] = len(cases)
        cases.append(kw)

    for i, K in enumerate(LARGE_KS):
        add(K=K, L=24 + (7 * i) % 40, kind="synth", quant=None, pinf=0.0, mh=1, h3=0)
        add(K=K, L=20 + (11 * i) % 44, kind="synth", quant=[2.0, 1.0, 4.0][i % 3], pinf=[0.05, 0.0, 0.3][i % 3],
            mh=i % 2, h3=(i // 2) % 2)
    for K in (257, 768, 1536, 4096, 4097):
        add(K=K, L=3000, kind="tiled", quant=None, pinf=0.0, mh=1, h3=0)
    add(K=512, L=3000, kind="synth", quant=1.0, pinf=0.0, mh=1, h3=0)  # ties row after row, thousands of rows
    add(K=1024, L=3000, kind="tiled", quant=None, pinf=0.0, mh=0, h3=1)
    for K in (2048, 8192, 16383):  # SURVEY 8d config 3b at its stated read length
        add(K=K, L=10000, kind="tiled", quant=None, pinf=0.0, mh=1, h3=0)
    return cases


def window_cap_cases():
    """c-core/window.c:13: a window is min(50 K, 100 000) nucleotides, so K >= 2001 and a read of 100 kb or more give
    windows of 100 000 rows -- the longest DP the scan ever runs.  One multi-wave class (K = 2048) and one strip class
    (K = 4200), planted domains; idx continues large_cases()' numbering (it seeds the inputs)."""
    return [dict(idx=100, K=2048, L=100000, kind="tiled", quant=None, pinf=0.0, mh=1, h3=0),
            dict(idx=101, K=4200, L=100000, kind="tiled", quant=None, pinf=0.0, mh=1, h3=0)]


_seeds = None


def seeds():
    global _seeds
    if _seeds is None:
        from deciphon_amd.synth import load_seeds

        _seeds = load_seeds(os.path.join(GOLDEN, "minifam.dcp"))
    return _seeds


def tiled_protein(case) -> dict:
    """The protein (deciphon_amd.synth layout: what a .dcp holds) behind a "tiled" case."""
    from deciphon_amd import synth

    return synth.tile_protein(seeds(), case["K"], 37 * case["idx"], f"TILE{case['K']}")


def build_case(case, orc):
    """-> (Profile in DP-cost space, read uint8[L], xt float32[13])"""
    from deciphon_amd import synth
    from oracle.dcp_reader import Protein

    rng = np.random.default_rng([20250310, case["idx"]])
    K, L, quant = case["K"], case["L"], case["quant"]
    if case["kind"] == "synth":
        prof = synth_profile(rng, K, quant, case["pinf"])
        seq = random_seq(rng, L)
    else:
        p = tiled_protein(case)
        prof = orc.setup_profile(Protein(p["accession"], 1, p["consensus"], K, p["null_emission"], p["bg_emission"],
                                         p["trans"], p["emission"], p["BMk"]))
        seq = random_seq(rng, L)
        # two planted domains with 5 % substitutions and 2 % + 2 % indels: a multi-hit path through
        # M, I and D states with N, J and C stretches around it
        cons = p["consensus"]
        for frac in (0.15, 0.6):
            a = int(rng.integers(0, max(len(cons) - 300, 1)))
            dom = synth.mutate(synth.back_translate(cons[a : a + 300]), rng, 0.05, 0.02, 0.02)[: L // 3]
            at = int(frac * L)
            seq[at : at + len(dom)] = dom[: L - at]
    xt = orc.xtrans(max(L // 3, 1), case["mh"], case["h3"])
    if quant:
        xt = (np.round(xt / quant) * quant).astype(np.float32)
    return prof, seq, xt


def path_cost(orc, prof, xt, seq, ids, sizes) -> float:
    """Cost of walking the unzipped path: every transition and emission it names, summed in double.
    Independent of any back-pointer: it checks that the steps are a legal path of the model
    (c-core/state.h ids, c-core/viterbi.c:492-586 transitions) whose total is the Viterbi optimum."""
    RR, SN, NN, SB, NB, EB, JB, EJ, JJ, EC, CC, ET, CT = (float(v) for v in xt)
    BM, MM, MI, MD, IM, II, DM, DD = range(8)
    tr = prof.trans.astype(np.float64)
    total, pos = 0.0, 0
    prev = None
    for sid, sz in zip(ids, sizes):
        sid, sz = int(sid), int(sz)
        kind, k = sid >> 14, (sid & 0x3FFF) - 1
        name = {3: {3: "S", 4: "N", 5: "B", 6: "E", 7: "J", 8: "C", 9: "T"}.get(sid & 0x3FFF)}.get(kind) or "MID"[kind]
        code = orc.code(seq, pos, sz) if sz else None
        if prev is not None:
            pn, pk = prev
            if name == "N":
                total += (SN if pn == "S" else NN)
            elif name == "B":
                total += {"S": SB, "N": NB, "E": EB, "J": JB}[pn]
            elif name == "J":
                total += (EJ if pn == "E" else JJ)
            elif name == "C":
                total += (EC if pn == "E" else CC)
            elif name == "T":
                total += (ET if pn == "E" else CT)
            elif name == "E":
                assert pn in "MD"
            elif name == "M":
                total += tr[BM, k] if pn == "B" else tr[{"M": MM, "I": IM, "D": DM}[pn], k]
                assert pn == "B" or pk == k - 1
            elif name == "I":
                total += tr[{"M": MI, "I": II}[pn], k]
                assert pk == k
            elif name == "D":
                total += tr[{"M": MD, "D": DD}[pn], k]
                assert pk == k - 1
        if sz:
            if name in "NJC":
                total += float(prof.null[code])
            elif name == "M":
                total += float(prof.match[code, k])
            elif name == "I":
                total += float(prof.bg[code])
            else:
                raise AssertionError(f"{name} emits")
        prev = (name, k)
        pos += sz
    assert pos == len(seq) and prev[0] == "T"
    return total
