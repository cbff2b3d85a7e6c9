#!/usr/bin/env python3
"""Merge the errors a GPU run measured in update mode (SBD_RATCHET_UPDATE=1 -> gpurun_out/measured_errors.json) into the
tracked record tests/golden/measured_errors.json: keys the record does not hold yet are added; with --all every
measured key replaces the record's value (after a kernel change that moved the errors on purpose)."""
import json, os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
rec_p = os.path.join(root, "tests", "golden", "measured_errors.json")
new_p = os.path.join(root, "gpurun_out", "measured_errors.json")
rec, new = json.load(open(rec_p)), json.load(open(new_p))
n = 0
for k, v in new.items():
    if k not in rec or "--all" in sys.argv:
        n += rec.get(k) != v
        rec[k] = v
json.dump(rec, open(rec_p, "w"), indent=0, sort_keys=True)
print(f"{n} key(s) written, {len(rec)} in the record")
