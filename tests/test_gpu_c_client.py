"""GPU: a plain C program (tests/c/scan_client.c, the shape of c-core/test_scan.c) compiled with
gcc against include/deciphon.h and linked to libdeciphon_hip.so."""
import os
import subprocess

import pytest

import deciphon_amd
from dcp_testlib import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def test_c_client_links_and_scans(tmp_path):
    exe = tmp_path / "scan_client"
    libdir = os.path.dirname(deciphon_amd.library_path())
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "scan_client.c"), "-o", str(exe), "-L", libdir, "-ldeciphon_hip",
                    f"-Wl,-rpath,{libdir}"], check=True)
    out = subprocess.run([str(exe), os.path.join(GOLDEN, "minifam.dcp"), os.path.join(GOLDEN, "consensus.fna"),
                          str(tmp_path / "prod")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip() == "rows=3 progress=100 windows=18"  # 3 profiles x 3 reads x 1 window, run twice
    rows = open(tmp_path / "prod" / "products.tsv").read().splitlines()
    gold = open(os.path.join(GOLDEN, "products.tsv")).read().splitlines()
    assert len(rows) == len(gold) == 4
    for got, want in zip(rows[1:], gold[1:]):
        assert got.split("\t")[:10] == want.split("\t")[:10]
    bad = subprocess.run([str(exe), str(tmp_path / "missing.dcp"), os.path.join(GOLDEN, "consensus.fna"),
                          str(tmp_path / "p2")], capture_output=True, text=True)
    assert bad.returncode == 1 and "failed to open DB file" in bad.stderr
