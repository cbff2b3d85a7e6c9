"""The reference's own formulation of the band system (sbdart_amd/csrc/sbd_refband.hpp) compiled for the HOST,
sbd_band_rcond_host, against the oracle's SGBCO estimate -- BIT FOR BIT.  The oracle is pinned on the reference: for the
records of tests/golden/illcond/reference_warnings.* the reference executable itself wrote (or did not write)
SBDART_WARNING.02, and 1 + RCOND == 1 must say the same.  The device kernel (band_rcond_kernel) runs this source; its own
test is tests/test_gpu_parity.py::test_reference_warning_fixtures.  No GPU involved here."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _host_rcond(L, r, mazim=0):
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    dt, w, pm = (np.ascontiguousarray(x, dtype=np.float64) for x in (r.dtauc, r.ssalb, r.pmom))
    out = C.c_double(0.0)
    rc = L.sbd_band_rcond_host(r.nlyr, r.nstr, r.nmom, mazim, int(r.plank), float(r.albedo), vp(dt), vp(w), vp(pm), C.byref(out))
    assert rc == 0, rc
    return out.value


def _oracle_rcond(rec):
    import pyoracle
    lib = pyoracle.lib()
    lib.sbdo_last_rcond.restype = C.c_double
    lib.sbdo_last_rcond.argtypes = [C.c_int]
    o = pyoracle.disort(rec)
    return lib.sbdo_last_rcond(0), o["status"]


def test_reference_warning_fixtures_on_the_host():
    from sbdart_amd import _lib
    from sbdart_amd.records import read_records
    L = _lib.load()
    recs = read_records(os.path.join(GOLDEN, "illcond", "reference_warnings.sbdrec"))
    meta = json.load(open(os.path.join(GOLDEN, "illcond", "reference_warnings.json")))["records"]
    assert len(recs) == len(meta) > 100
    npos = nneg = 0
    for r, m in zip(recs, meta):
        want, st = _oracle_rcond(r)
        got = _host_rcond(L, r)
        assert got == want or (got != got and want != want), (m["family"], got, want)        # the same bits
        ref_warns_2 = 2 in m["reference_warnings"]
        assert (1.0 + got == 1.0) == ref_warns_2 == bool(st & 1), (m, got)
        npos += ref_warns_2
        nneg += (not ref_warns_2) and m["family"].startswith("band")
    assert npos >= 30 and nneg >= 4, (npos, nneg)          # the reference raised errmsg 2 in these, and narrowly did not in those


def test_oracle_is_pinned_on_the_reference_for_every_fixture():
    """The labels and outputs in reference_warnings.* come from the reference EXECUTABLE (one run per record, its
    SBDART_WARNING.NN files).  The C oracle must raise exactly those warnings (errmsg 2 / 3 / 4 <-> bits 1 / 2 / 4) and
    return the reference's fluxes bit for bit -- also where the band system is singular to working precision."""
    import pyoracle
    from sbdart_amd.records import read_records
    recs = read_records(os.path.join(GOLDEN, "illcond", "reference_warnings.sbdrec"))
    meta = json.load(open(os.path.join(GOLDEN, "illcond", "reference_warnings.json")))["records"]
    seen = {2: 0, 3: 0}
    for r, m in zip(recs, meta):
        o = pyoracle.disort(r)
        w = m["reference_warnings"]
        want = (1 if 2 in w else 0) | (2 if 3 in w else 0) | (4 if 4 in w else 0)
        assert (o["status"] & 7) == want, (m, o["status"])
        for f in ("rfldir", "rfldn", "flup", "dfdt", "uavg"):
            assert np.array_equal(o[f], getattr(r, f), equal_nan=True), (m["family"], f)
        for k in seen:
            seen[k] += k in w
    assert seen[2] >= 30 and seen[3] >= 60, seen


@pytest.mark.parametrize("name", ["sbchk1", "cfgB_sw_nstr16", "cfgD_nstr32_50ly", "cfg3_lw_nstr16_cloud", "conservative_thermal",
                                  "illcond/nstr40_next_to_conservative", "illcond/thin65_thermal"])
def test_golden_records_on_the_host(name):
    """Ordinary and ill-conditioned reference records, NSTR 4..40, LYRCUT on and off: the same estimate, bit for bit."""
    from sbdart_amd import _lib
    from sbdart_amd.records import read_records
    L = _lib.load()
    recs = [r for r in read_records(os.path.join(GOLDEN, name + ".sbdrec")) if r.lamber and not r.ibcnd][:10]
    assert recs
    for r in recs:
        want, st = _oracle_rcond(r)
        if st & 0x38:
            continue
        got = _host_rcond(L, r)
        assert got == want, (name, r.nstr, r.nlyr, got, want)
