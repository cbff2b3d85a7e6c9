"""Error ratchet for the GPU parity tests ("gates are loose against what is measured, and nothing
ratchets").  Every gated comparison reports its worst normalised error under a stable key; the tracked file
tests/golden/measured_errors.json holds what the committed kernels measured on an MI355X.  A test passes only if

    error <= absolute gate of the test                       (as before)
    error <= max(10 x recorded error, FLOOR)                 (the ratchet: a regression of >10x fails)

A key with no record FAILS (so a new comparison cannot slip in ungated) unless SBD_RATCHET_UPDATE=1, in which case
the run writes what it measured to gpurun_out/measured_errors.json (merged back from the GPU box by gpurun) and the
builder copies it over the tracked file:   tools/ratchet_update.sh.
Normalisation: |gpu - ref| / (max|column| + 1e-6 max|record|) -- dimensionless, independent of the gate's value.
"""
import json
import os

from conftest import GOLDEN, ROOT

RECORD = os.path.join(GOLDEN, "measured_errors.json")
UPDATE = os.environ.get("SBD_RATCHET_UPDATE") == "1"
OUT = os.path.join(ROOT, "gpurun_out", "measured_errors.json")
FLOOR = 2e-13          # below this a factor 10 is rounding weather, not a regression
FACTOR = 10.0

_recorded = json.load(open(RECORD)) if os.path.exists(RECORD) else {}
_measured = {}


def normalised(err, scale, recmax):
    return float(err) / (float(scale) + 1e-6 * float(recmax) + 1e-300)


def ratchet(key, value):
    """Report `value` (worst normalised error of one comparison) under `key`; assert the ratchet."""
    value = float(value)
    _measured[key] = max(value, _measured.get(key, 0.0))
    if UPDATE:
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        old = json.load(open(OUT)) if os.path.exists(OUT) else {}
        old[key] = _measured[key]
        with open(OUT, "w") as f:
            json.dump(old, f, indent=0, sort_keys=True)
        return
    assert key in _recorded, f"no recorded error for {key!r}: run the GPU suite with SBD_RATCHET_UPDATE=1 and commit"
    allowed = max(FACTOR * _recorded[key], FLOOR)
    assert value <= allowed, (f"{key}: error {value:.3e} is more than {FACTOR:g}x the recorded {_recorded[key]:.3e} "
                              f"(tests/golden/measured_errors.json)")
