"""GPU box: one INPUT of the end-to-end fuzz under the host's path switches (default: compact form + gas on the device;
SBD_HOST_GAS=1; SBD_NO_MIX=1) against the reference's stdout -- which path a difference belongs to.
usage: python tools/repro_case.py "namelist text" [files]   (files: take tests/test_band_model.py's USER_FILES)"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
from test_fortran_host import run_reference_and_host, _compare_stdout
nl = sys.argv[1]
files = None
if len(sys.argv) > 2:
    from test_band_model import USER_FILES
    files = USER_FILES
for name, env in (("default", {}), ("SBD_HOST_GAS=1", {"SBD_HOST_GAS": "1"}), ("SBD_NO_MIX=1", {"SBD_NO_MIX": "1"}),
                  ("SBD_FORCE_EIG_FALLBACK=1", {"SBD_FORCE_EIG_FALLBACK": "1"}), ("SBD_BAND_V1=1", {"SBD_BAND_V1": "1"}),
                  ("SBD_LAYER_V1=1", {"SBD_LAYER_V1": "1"})):
    with tempfile.TemporaryDirectory() as d:
        try:
            ref, got, cap = run_reference_and_host(nl, d, from_input=True, files=files, host_env=env)
        except AssertionError as e:
            print(name, "host failed:", str(e)[:300]); continue
        try:
            off = _compare_stdout(got, ref)
            print(name, "ok", len(ref.split()), "tokens,", off, "off by one")
        except AssertionError as e:
            nan = sum(1 for t in got.split() if "NaN" in t)
            print(name, "DIFFERS:", str(e)[:200], "| NaN tokens in host output:", nan, "of", len(got.split()))
