#!/usr/bin/env python3
"""Fixtures from what the reference's AUTHORS shipped (build container only: reads /root/reference).

  * TestRuns/sbchk.1-5            -- the golden outputs `TestRuns/test_runs` is diffed against (test_runs:6);
  * RunRT/RUNS/<name>.sbd (12)     -- captured sweeps of the GUI: a command block, `_DATA_`, the tokens the runs printed.
    The 12 are the ones SURVEY.md section 4 found replayable (sbchk1-5 duplicate TestRuns' inputs as command blocks).

Both kinds are DATA files of the reference (inputs and expected outputs); they are stored as they are, gzip'ed
(mtime 0: reproducible bytes).  tests/test_shipped_goldens.py replays them.  Nothing here is reference source text:
the bash script that wrote sbchk.N is NOT copied -- the runs' inputs come from the sbchkN.sbd command blocks."""
import gzip
import json
import os

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
SWEEPS = ["sbchk1", "sbchk2", "sbchk3", "sbchk4", "sbchk5", "sza_tcloud", "tcloud_albcon_sza_wlinf",
          "tcloud_nre_albcon_10", "tcloud_nre_sza_albcon", "tcloud_sza_iout_11", "test", "wlinf_iout_11"]


def put(name, data):
    with open(os.path.join(OUT, name + ".gz"), "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0, filename="") as g:
            g.write(data)


def main():
    manifest = {}
    for n in range(1, 6):
        raw = open(os.path.join(REF, "TestRuns", f"sbchk.{n}"), "rb").read()
        put(f"sbchk.{n}", raw)
        manifest[f"sbchk.{n}"] = {"source": f"TestRuns/sbchk.{n}", "tokens": len(raw.split())}
    for s in SWEEPS:
        raw = open(os.path.join(REF, "RunRT", "RUNS", s + ".sbd"), "rb").read()
        put(s + ".sbd", raw)
        data = raw.decode().split("_DATA_", 1)[1]
        manifest[s + ".sbd"] = {"source": f"RunRT/RUNS/{s}.sbd", "tokens": len(data.split())}
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(sum(v["tokens"] for k, v in manifest.items() if k.endswith(".sbd")), "sweep tokens,",
          sum(v["tokens"] for k, v in manifest.items() if not k.endswith(".sbd")), "TestRuns tokens")


if __name__ == "__main__":
    main()
