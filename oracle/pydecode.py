"""TEST INFRASTRUCTURE ONLY -- quasi-codon model of a Deciphon state, restated for the parity tests.

decoder_decode (c-core/decoder.c:38-58) calls third-party imm's imm_frame_cond_decode, whose source is not in
the reference tree (imm is cloned at HEAD by the reference's CI: unpinned).  What IS in the tree are pressed
emission tables computed by the same model (control/tests/files/minifam.dcp): `emission_lprob` below is the
marginal form of the model -- P(z) of emitting the 1..5 nucleotides z from a state with nucleotide distribution
p and codon marginals M (5 x 5 x 5, index 4 = any) under per-base error rate e -- and tests/test_decoder.py shows
that it reproduces every entry of those tables to fp32 rounding.  `decode` is the same formula with M replaced by
the indicator of one codon x, i.e. P(z | x), maximised over the 64 codons weighted by M[x]: the restatement of
imm_frame_cond_decode.  Pinned by the tables (the likelihood) and by the reference's committed products.tsv
(three hits of exact codons); the tie rule and anything else of imm's decode is parity unpinned.
"""
from __future__ import annotations

import itertools

import numpy as np

ANY = 4
# NCBI translation table 1 (imm_gencode, third-party), amino acids in TCAG order of the codon positions
GENCODE1 = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
_TCAG = {0: 2, 1: 1, 2: 3, 3: 0}  # A, C, G, T -> position in T, C, A, G


def emission_prob(e: float, p, M, z) -> float:
    """P(z): probability (not log) that the state emits the nucleotide indices z (1..5 of them)."""
    f, n = 1.0 - e, len(z)

    def del1(a, b):  # codons with one base deleted that read (a, b)
        return M[ANY, a, b] + M[a, ANY, b] + M[a, b, ANY]

    def one(a):  # codons holding base a somewhere, two bases deleted
        return M[a, ANY, ANY] + M[ANY, a, ANY] + M[ANY, ANY, a]

    if n == 1:
        return e * e * f * f / 3 * one(z[0])
    if n == 2:
        return 2 * e * f ** 3 / 3 * del1(z[0], z[1]) + e ** 3 * f / 3 * (p[z[0]] * one(z[1]) + p[z[1]] * one(z[0]))
    if n == 3:
        v = f ** 4 * M[z[0], z[1], z[2]]
        v += 4 * e * e * f * f / 9 * (p[z[0]] * del1(z[1], z[2]) + p[z[1]] * del1(z[0], z[2]) + p[z[2]] * del1(z[0], z[1]))
        return v + e ** 4 * p[z[0]] * p[z[1]] * p[z[2]]
    if n == 4:
        a = sum(p[z[j]] * M[tuple(z[t] for t in range(4) if t != j)] for j in range(4))
        b = 0.0
        for i, j in itertools.combinations(range(4), 2):
            r = [z[t] for t in range(4) if t not in (i, j)]
            b += p[z[i]] * p[z[j]] * del1(r[0], r[1])
        return e * f ** 3 / 2 * a + e ** 3 * f / 9 * b
    if n == 5:
        v = 0.0
        for i, j in itertools.combinations(range(5), 2):
            v += p[z[i]] * p[z[j]] * M[tuple(z[t] for t in range(5) if t not in (i, j))]
        return e * e * f * f / 10 * v
    raise ValueError("1..5 nucleotides")


def decode(e: float, nucltp, codonm, z):
    """-> (codon as 3 nucleotide indices, amino acid under translation table 1)."""
    p = np.exp(np.asarray(nucltp, np.float64))
    M = np.exp(np.asarray(codonm, np.float64)).reshape(5, 5, 5)
    best, arg = 0.0, None
    for x in itertools.product(range(4), repeat=3):
        ind = np.zeros((5, 5, 5))
        for a in (x[0], ANY):
            for b in (x[1], ANY):
                for c in (x[2], ANY):
                    ind[a, b, c] = 1.0
        joint = M[x] * emission_prob(e, p, ind, list(z))
        if joint > best:
            best, arg = joint, x
    if arg is None:
        raise ValueError("no codon has positive probability")
    return arg, GENCODE1[_TCAG[arg[0]] * 16 + _TCAG[arg[1]] * 4 + _TCAG[arg[2]]]
