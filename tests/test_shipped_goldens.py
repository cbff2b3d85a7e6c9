"""Parity against what the reference's AUTHORS shipped (VERDICT r03 "missing #3"): TestRuns/sbchk.1-5 and the twelve
replayable sweeps of RunRT/RUNS/*.sbd, stored as they are under tests/golden/shipped/ (make_shipped.py).

Every sweep's command block is expanded with RunRT's rules (sbdart_amd/sweep.py), every run is made from its INPUT
alone, and the printed tokens are compared with the shipped `_DATA_` block by the rule of SURVEY.md section 4: the
same printed number (5 digits), fields below 1e-6 of the file's largest magnitude exempt (cancellation noise that
differs between compilers), at most 0.2 % of the tokens one unit off in the last printed digit.

  * CPU leg (`-m "not gpu"`): the reference compiled here (oracle/_ref/sbdart_ref) -- closes the chain
    GPU engine -> oracle -> amdflang build -> what the authors' own build printed;
  * GPU leg (`-m gpu`): sbdart_amd in batch mode (one process for the sweep's hundreds of runs).
The counts go to gpurun_out/shipped_tokens_<leg>.json (copied to profiles/ per round)."""
import gzip
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

from conftest import GOLDEN, REF_DIR, ROOT, have_ref
from sbdart_amd.sweep import Sweep, run_directories
from test_fortran_host import _compare_stdout

SHIPPED = os.path.join(GOLDEN, "shipped")
MANIFEST = json.load(open(os.path.join(SHIPPED, "MANIFEST.json")))
SWEEPS = sorted(k[:-4] for k in MANIFEST if k.endswith(".sbd"))
HOST = os.path.join(ROOT, "sbdart_amd", "bin", "sbdart_amd")


def shipped(name):
    return gzip.open(os.path.join(SHIPPED, name + ".gz"), "rt").read()


def command_and_data(name):
    block, data = shipped(name + ".sbd").split("_DATA_", 1)
    return block, data


def test_fixtures_are_what_the_manifest_says():
    assert len(SWEEPS) == 12
    total = 0
    for s in SWEEPS:
        block, data = command_and_data(s)
        assert len(data.split()) == MANIFEST[s + ".sbd"]["tokens"]
        total += len(data.split())
        assert len(Sweep(block)) >= 1
    assert total == 79796
    for n in range(1, 6):
        assert len(shipped(f"sbchk.{n}").split()) == MANIFEST[f"sbchk.{n}"]["tokens"]


def _record(leg, name, ntok, off):
    try:
        out = os.path.join(ROOT, "gpurun_out", f"shipped_tokens_{leg}.json")      # (a file per leg: they run on different boxes)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        d = json.load(open(out)) if os.path.exists(out) else {}
        d[name] = {"tokens": ntok, "one_unit_off": off}
        json.dump(d, open(out, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def _one_process_per_run(exe, sweep, workdir):
    """The reference's way: cwd = the run's directory, INPUT in it, text on stdout.  Runs side by side (each in its own
    directory: the reference's files are cwd-relative)."""
    def one(it):
        d = os.path.join(workdir, f"run{it:04d}")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "INPUT"), "w") as f:
            f.write("\n &INPUT\n" + sweep.inputs(it)[0] + " /\n")
        p = subprocess.run([exe], cwd=d, capture_output=True, text=True)
        assert p.returncode == 0, (d, p.stderr[-500:])
        return p.stdout
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        return list(ex.map(one, range(len(sweep))))


def _sorted_records(text, width):
    tok = text.split()
    recs = [tok[i:i + width] for i in range(0, len(tok), width)]
    return " ".join(" ".join(r) for r in sorted(recs, key=lambda r: [float(x) for x in r]))


def _check_sweep(leg, name, outs):
    _, data = command_and_data(name)
    got = "".join(outs)
    off = _compare_stdout(got, data)
    _record(leg, name + ".sbd", len(data.split()), off)
    # ... and TestRuns' own golden file for the five examples (the same runs; example 4 in another loop order:
    # its one-line records are matched as sorted lists)
    if name.startswith("sbchk"):
        want = shipped("sbchk." + name[-1])
        if name == "sbchk4":
            got, want = _sorted_records(got, 9), _sorted_records(want, 9)
        off = _compare_stdout(got, want)
        _record(leg, "sbchk." + name[-1], len(want.split()), off)


@pytest.mark.parametrize("name", SWEEPS)
def test_reference_build_reproduces_the_shipped_outputs(name, tmp_path):
    """CPU: the reference compiled here with amdflang prints what the authors' build printed."""
    if not have_ref("sbdart_ref"):
        pytest.skip("oracle/_ref/sbdart_ref not built (oracle/build_ref.sh needs /root/reference)")
    block, _ = command_and_data(name)
    outs = _one_process_per_run(os.path.join(REF_DIR, "sbdart_ref"), Sweep(block), str(tmp_path))
    _check_sweep("reference_build", name, outs)


@pytest.mark.gpu
@pytest.mark.parametrize("name", SWEEPS)
def test_engine_reproduces_the_shipped_outputs(name, tmp_path):
    """GPU: every run of the shipped sweep from its INPUT alone, through sbdart_amd --batch (one process)."""
    from test_fortran_host import _build
    _build()
    block, _ = command_and_data(name)
    outs = Sweep(block).run_batch(HOST, str(tmp_path))
    _check_sweep("engine", name, outs)
