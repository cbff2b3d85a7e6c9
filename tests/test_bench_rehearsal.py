"""The N > 1 bench line rehearsed on ONE GPU (VERDICT r05 "next" #4): `bench.py --gpus 2 --share-device` runs both ranks
on cuda:0 and reduces over gloo, so that every piece of bookkeeping an 8-GPU SCALE run needs -- shards, the MAX-reduced
clock, the sum over ranks, `roofline`, `cpu_baseline`, `flux_rmse_vs_cpu` -- has executed before the driver's first real
multi-GPU run.  The line says `"rehearsal": true`: it is not a multi-GPU measurement and bench.py refuses the flag unless
this test's environment variable is set."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None, timeout=900):
    env = dict(os.environ, **(env_extra or {}))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    return p


def test_share_device_is_refused_without_the_test_switch():
    env = {k: v for k, v in os.environ.items() if k != "SBD_BENCH_SHARE_DEVICE_TEST"}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "share-device" in (p.stderr + p.stdout)


def test_strong_scaling_sizes_itself():
    """--scaling strong without --nwl: 49 152 points PER RANK of the one sweep (a bench-size batch on every GPU)."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.default_nwl("weak", 8) == 49152 and bench.default_nwl("strong", 8) == 8 * 49152
    assert bench.default_nwl("strong", 1) == 49152


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_two_rank_line_on_one_gpu(scaling):
    nwl = 6144 if scaling == "strong" else 3072
    p = _run(["--gpus", "2", "--share-device", "--scaling", scaling, "--nwl", str(nwl), "--steps", "2", "--warmup", "1",
              "--no-side-lines", "--cpu-baseline-seconds", "2"], {"SBD_BENCH_SHARE_DEVICE_TEST": "1"})
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]              # rank 0 prints, nobody else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rehearsal"] is True and d["scaling"] == scaling
    assert d["metric"].startswith("spectral-points/sec") and d["unit"] == "spectral-points/s" and d["dtype"] == "f64"
    sh = d["config"]["shards"]
    assert [s["rank"] for s in sh] == [0, 1]
    if scaling == "strong":
        # ONE sweep cut between spectral points: disjoint, covering, balanced
        assert sh[0]["points"][0] == 0 and sh[0]["points"][1] == sh[1]["points"][0] and sh[1]["points"][1] == nwl
        assert abs((sh[0]["points"][1] - sh[0]["points"][0]) - (sh[1]["points"][1] - sh[1]["points"][0])) <= 1
        assert d["config"]["nwl_total"] == nwl
    else:
        assert all(s["points"] == [0, nwl] for s in sh) and d["config"]["nwl_total"] == 2 * nwl
    assert d["config"]["solves_total"] == sum(s["solves"] for s in sh)
    # the clock is the slowest rank's, the rate is the whole job's points over it
    slowest = max(s["timed_s"] for s in sh)
    assert abs(d["ms_per_step"] - 1e3 * slowest / d["steps"]) <= 1e-6 * d["ms_per_step"] + 1e-9
    assert abs(d["value"] - d["config"]["nwl_total"] * d["steps"] / slowest) <= 1e-9 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and r["kernel"].startswith("void sbd::") and "kernel" in r["kernel"]
    c = d["cpu_baseline"]
    assert c is not None and c["value"] > 0 and c["cores"] == 1 and c["kind"] in ("reference", "port") and "rank 0" in c["sample"]
    f = d["flux_rmse_vs_cpu"]
    assert f["solves"] > 100 and f["per_solve_max_abs"] < 1e-9
    assert d["nonzero_status"] == 0
