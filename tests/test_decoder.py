"""CPU: quasi-codon decoding (c-core/decoder.c:38-58, c-core/match.c:66-89).

The arithmetic lives in third-party imm (imm_frame_cond_decode), absent from the reference tree and unpinned in
its CI.  Three pins, from strongest to weakest:
  1. the quasi-codon MODEL: its marginal form (oracle/pydecode.py) reproduces the emission tables of the
     reference's own pressed database, every one of the 1364 codes of every node checked;
  2. the product's decoder (csrc/host_logic.cpp, closed forms) equals the oracle's (the marginal formula on an
     indicator codon distribution) on random quasi-codons of every length;
  3. the reference's committed products.tsv: (quasi-codon, codon, amino) of all 582 steps of its three hits.
What imm does on exact ties, and its fp32 log-space rounding near ties, is parity unpinned."""
import os

import numpy as np
import pytest

from dcp_testlib import GOLDEN
from oracle import pydecode

CODE_OFF = (0, 4, 20, 84, 340)


@pytest.fixture(scope="module")
def seeds():
    from deciphon_amd import synth

    return synth.load_seeds(os.path.join(GOLDEN, "minifam.dcp"))


def _codes():
    import itertools

    for n in range(1, 6):
        for i, z in enumerate(itertools.product(range(4), repeat=n)):
            yield CODE_OFF[n - 1] + i, list(z)


def test_model_reproduces_the_pressed_emission_tables(seeds):
    eps = 0.01  # header.epsilon of the fixture
    checked = 0
    for s in seeds:
        K = s["core_size"]
        for n in (0, 1, K // 2, K - 1):
            p = np.exp(s["nucltp"][2 + n].astype(np.float64))
            M = np.exp(s["codonm"][2 + n].astype(np.float64)).reshape(5, 5, 5)
            table = s["emission"][n].astype(np.float64)
            for code, z in _codes():
                want = table[code]
                got = np.log(pydecode.emission_prob(eps, p, M, z))
                assert abs(got - want) <= 2e-5 * max(1.0, abs(want)), (s["accession"], n, z, got, want)
                checked += 1
    assert checked == 3 * 4 * 1364
    # the null and background tables come from the same model
    s = seeds[0]
    for entry, table in ((0, s["null_emission"]), (1, s["bg_emission"])):
        p = np.exp(s["nucltp"][entry].astype(np.float64))
        M = np.exp(s["codonm"][entry].astype(np.float64)).reshape(5, 5, 5)
        for code, z in list(_codes())[::7]:
            assert abs(np.log(pydecode.emission_prob(eps, p, M, z)) - float(table[code])) <= 2e-5 * max(1.0, abs(float(table[code])))


def test_product_decoder_equals_the_oracle(seeds):
    from deciphon_amd import host

    rng = np.random.default_rng(4)
    for it in range(400):
        s = seeds[it % 3]
        entry = int(rng.integers(0, s["core_size"] + 2))  # null, background and nodes alike
        z = rng.integers(0, 4, size=int(rng.integers(1, 6))).astype(np.uint8)
        codon = host.decode_quasi_codon(0.01, s["nucltp"][entry], s["codonm"][entry], z)
        want, amino = pydecode.decode(0.01, s["nucltp"][entry], s["codonm"][entry], z)
        assert tuple(int(v) for v in codon) == want, (it, entry, z)
        assert host.gencode_amino(1, codon) == amino
    assert host.gencode_amino(999, [0, 0, 0]) == ""  # unknown translation table
    assert host.gencode_amino(11, [0, 3, 2]) == "M" and host.gencode_amino(4, [3, 2, 0]) == "W"  # ATG; TGA in table 4


def test_reference_products_triples(seeds):
    """control/tests/files/snap.dcs: every (quasi-codon, state, codon, amino) of the three golden hits."""
    from deciphon_amd import host

    acc = {s["accession"]: s for s in seeds}
    nt = {"A": 0, "C": 1, "G": 2, "T": 3}
    rows = [ln.rstrip("\n").split("\t") for ln in open(os.path.join(GOLDEN, "products.tsv"))][1:]
    n = 0
    for row in rows:
        s = acc[row[7]]
        for cell in row[11].split(";"):
            frag, state, codon, amino = cell.split(",")
            if state in ("B", "E"):
                assert (frag, codon, amino) == ("", "", "")
                continue
            assert state[0] == "M"
            k = int(state[1:]) - 1
            got = host.decode_quasi_codon(0.01, s["nucltp"][2 + k], s["codonm"][2 + k], [nt[c] for c in frag])
            assert "".join("ACGT"[v] for v in got) == codon and host.gencode_amino(s["gencode"], got) == amino, cell
            n += 1
    assert n == 173 + 241 + 162
