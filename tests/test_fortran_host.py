"""The Fortran-2003 host (sbdart_amd/fortran): ISO_C_BINDING shim + program sbdart_amd.

CPU: it builds with amdflang, its spectral-grid logic (setfilt's grid size + wllimits,
spectra.f:3370-3384, drt.f:1657-1740) reproduces the band edges of the reference-captured
records exactly, and it fails loudly without a GPU.
GPU: with the optical properties of the reference's own test runs (TestRuns/test_runs)
it reproduces the reference's stdout (sbchk.1, sbchk.2, every third run of sbchk.4, sbchk.5) at print
precision.
"""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

BIN = os.path.join(ROOT, "sbdart_amd", "bin")
HOST = os.path.join(BIN, "sbdart_amd")
FLANG = "/opt/rocm/bin/amdflang"

needs_flang = pytest.mark.skipif(not os.path.exists(FLANG), reason="amdflang not installed")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "sbdart_amd", "csrc"), "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "sbdart_amd", "fortran"), "-s"])


@needs_flang
def test_host_builds_and_grid_matches_reference_records():
    _build()
    for name, args in (("sbchk1", (".25", "1.0", ".005")), ("sbchk3", ("4", "20", "-.01")),
                       ("cfgD_nstr32_50ly", (".25", "100", "20")), ("sbchk2", (".55", ".55", "0"))):
        out = subprocess.check_output([os.path.join(BIN, "sbd_grid_selftest"),
                                       os.path.join(GOLDEN, name + ".sbdrec"), *args], text=True).split()
        assert float(out[2]) == 0.0, (name, out)
    assert subprocess.check_output([os.path.join(BIN, "sbd_grid_selftest"), os.path.join(GOLDEN, "sbchk1.sbdrec"),
                                    ".25", "1.0", ".005"], text=True).split()[0] == "151"


@needs_flang
def test_host_prints_namelist_without_input_and_fails_without_gpu(tmp_path):
    import torch
    _build()
    r = subprocess.run([HOST], cwd=tmp_path, capture_output=True, text=True)
    assert "&INPUT" in r.stdout.upper() and "WLINF" in r.stdout.upper()   # drt.f:228-231
    if not torch.cuda.is_available():
        (tmp_path / "INPUT").write_text(" &INPUT\n idatm=4, isat=0, wlinf=.25, wlsup=1.0, wlinc=.005, iout=1,\n /\n")
        env = dict(os.environ, SBD_OPTICS=os.path.join(GOLDEN, "sbchk1.sbdrec"))
        r = subprocess.run([HOST], cwd=tmp_path, env=env, capture_output=True, text=True)
        assert r.returncode != 0 and "no usable HIP device" in r.stderr


def _tokens(text):
    out = []
    for tok in text.split():
        try:
            out.append(float(tok))
        except ValueError:
            out.append(tok)
    return out


def _compare_stdout(got, want):
    """SURVEY.md section 4 rule: equal after rounding to 5 significant digits, with an
    absolute floor (1e-6 x the largest magnitude of the file) for analytically-zero fields."""
    g, w = _tokens(got), _tokens(want)
    assert len(g) == len(w), (len(g), len(w))
    nums = [abs(x) for x in w if isinstance(x, float)]
    floor = 1e-6 * max(nums)
    for a, b in zip(g, w):
        if isinstance(b, str):
            assert a == b
        else:
            assert abs(a - b) <= 2e-4 * abs(b) + floor or abs(a - b) <= floor, (a, b)


def _runs(recs):
    """Split a concatenated record list into sbdart runs (a run restarts at iwl=1, kd=1)."""
    runs, cur = [], []
    for r in recs:
        if r.iwl == 1 and r.kd == 1 and cur and not (cur[-1].iwl == 1 and cur[-1].kd < r.kd):
            runs.append(cur)
            cur = []
        cur.append(r)
    runs.append(cur)
    return runs


@pytest.mark.gpu
@needs_flang
@pytest.mark.parametrize("case", ["sbchk1", "sbchk2", "sbchk4", "sbchk5"])
def test_host_reproduces_reference_stdout(case, tmp_path):
    from sbdart_amd.records import read_records, write_records
    _build()
    man = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))[case]
    recs = read_records(os.path.join(GOLDEN, case + ".sbdrec"))
    runs = _runs(recs)
    assert len(runs) == len(man["namelists"]), (len(runs), len(man["namelists"]))
    got = ""
    for i, (nl, rr) in enumerate(zip(man["namelists"], runs)):
        d = tmp_path / f"run{i}"
        d.mkdir()
        (d / "INPUT").write_text("\n &INPUT\n" + nl + "\n /\n")
        write_records(str(d / "OPTICS.sbdrec"), [r.inputs_only() for r in rr], with_out=False)
        p = subprocess.run([HOST], cwd=d, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        got += p.stdout
    want = open(os.path.join(GOLDEN, case + ".stdout")).read()
    _compare_stdout(got, want)
