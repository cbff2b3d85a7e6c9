"""`sbdart_amd --serve` + the `sbdart` client (VERDICT r05 "next" #3c): the harnesses of the reference launch an executable
named `sbdart` once per run, in the run's directory, and read its stdout (RunRT/RunRT.py:2021-2044, TestRuns/test_runs:31-145).
A process of its own pays the HIP runtime's start-up for every run; the client hands the run -- its directory and its own
file descriptors 1 and 2 -- to a resident server instead.  The harness stays unchanged.

CPU tests: the protocol (descriptor passing, working directory, exit code, auto-start, the fall-back to a process of its
own, the idle exit) on runs that need no GPU (IOUT = 2: the band model's gas-depth report; a missing INPUT).
GPU tests: TestRuns' five examples (180 runs) through the client, byte for byte what `sbdart_amd --batch` prints and token
for token what the authors shipped; a run too large for a file of work items made whole in the server (phase 3)."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT

BIN = os.path.join(ROOT, "sbdart_amd", "bin")
HOST, CLIENT = os.path.join(BIN, "sbdart_amd"), os.path.join(BIN, "sbdart")


def _build():
    from test_fortran_host import _build as b
    b()
    assert os.access(CLIENT, os.X_OK), "sbdart client not built (make -C sbdart_amd/fortran)"


def _mkrun(d, body):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "INPUT"), "w") as f:
        f.write("\n &INPUT\n" + body + "\n /\n")
    return d


class Server:
    def __init__(self, sock, idle_s=60, env=None):
        self.sock = sock
        e = dict(os.environ, SBDART_AMD_IDLE_S=str(idle_s), **(env or {}))
        self.p = subprocess.Popen([HOST, "--serve", sock], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        for _ in range(3000):
            if os.path.exists(sock) or self.p.poll() is not None:
                break
            time.sleep(0.01)
        assert os.path.exists(sock), self.p.communicate()

    def stop(self):
        self.p.terminate()
        try:
            self.p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            self.p.kill()


GAS_REPORT = "idatm=4, wlinf=.5, wlsup=.6, wlinc=.05, iout=2"


def test_client_and_server_speak(tmp_path):
    _build()
    sock = str(tmp_path / "sock")
    d1 = _mkrun(str(tmp_path / "a"), GAS_REPORT)
    d2 = _mkrun(str(tmp_path / "b"), "idatm=2, wlinf=.3, wlsup=.4, wlinc=.02, iout=2")
    alone = [subprocess.run([HOST], cwd=d, capture_output=True, text=True) for d in (d1, d2)]
    srv = Server(sock)
    try:
        env = dict(os.environ, SBDART_AMD_SOCKET=sock, SBDART_AMD_NO_AUTOSTART="1")
        for rep in range(3):
            for d, a in zip((d1, d2), alone):
                p = subprocess.run([CLIENT], cwd=d, env=env, capture_output=True, text=True)
                assert p.returncode == 0 and p.stdout == a.stdout and len(p.stdout.split()) > 10   # the client's OWN stdout received it
        # a directory without INPUT: what a process of its own prints (the reference lists the namelist's defaults, drt.f:228-231)
        e = str(tmp_path / "empty")
        os.makedirs(e)
        a = subprocess.run([HOST], cwd=e, capture_output=True, text=True)
        p = subprocess.run([CLIENT], cwd=e, env=env, capture_output=True, text=True)
        assert p.stdout == a.stdout
        # the server is still there and still right after that
        p = subprocess.run([CLIENT], cwd=d1, env=env, capture_output=True, text=True)
        assert p.stdout == alone[0].stdout
        assert srv.p.poll() is None
    finally:
        srv.stop()


def test_client_starts_its_server_and_the_server_leaves_when_idle(tmp_path):
    _build()
    sock = str(tmp_path / "s" / "sock")
    os.makedirs(os.path.dirname(sock))
    d = _mkrun(str(tmp_path / "a"), GAS_REPORT)
    want = subprocess.run([HOST], cwd=d, capture_output=True, text=True).stdout
    env = dict(os.environ, SBDART_AMD_SOCKET=sock, SBDART_AMD_IDLE_S="2")
    p = subprocess.run([CLIENT], cwd=d, env=env, capture_output=True, text=True, timeout=120)     # nobody listens: the client starts one
    assert p.returncode == 0 and p.stdout == want
    assert os.path.exists(sock)
    t0 = time.perf_counter()
    p = subprocess.run([CLIENT], cwd=d, env=env, capture_output=True, text=True)                  # ... and the next run finds it
    assert p.stdout == want and time.perf_counter() - t0 < 2.0
    for _ in range(100):                                                                           # idle for 2 s: gone, socket removed
        if not os.path.exists(sock):
            break
        time.sleep(0.1)
    assert not os.path.exists(sock)


def test_without_a_server_the_client_is_the_executable(tmp_path):
    _build()
    d = _mkrun(str(tmp_path / "a"), GAS_REPORT)
    want = subprocess.run([HOST], cwd=d, capture_output=True, text=True).stdout
    env = dict(os.environ, SBDART_AMD_SOCKET=str(tmp_path / "nobody"), SBDART_AMD_NO_AUTOSTART="1")
    p = subprocess.run([CLIENT], cwd=d, env=env, capture_output=True, text=True)
    assert p.returncode == 0 and p.stdout == want


def test_a_second_server_on_the_same_socket_is_refused(tmp_path):
    _build()
    sock = str(tmp_path / "sock")
    srv = Server(sock)
    try:
        p = subprocess.run([HOST, "--serve", sock], capture_output=True, text=True, timeout=60)
        assert p.returncode != 0 and "another server" in (p.stderr + p.stdout)
        assert os.path.exists(sock) and srv.p.poll() is None
    finally:
        srv.stop()


@pytest.mark.gpu
def test_testruns_through_the_client(tmp_path):
    """TestRuns' five examples -- 180 runs launched ONE PROCESS PER RUN like test_runs does, the process being the client --
    give byte for byte the texts of `sbdart_amd --batch` and, token for token, the authors' sbchk.1-5."""
    _build()
    from sbdart_amd.sweep import Sweep
    from test_shipped_goldens import _check_sweep, command_and_data
    sock = str(tmp_path / "sock")
    srv = Server(sock, env={"SBD_TIMING": "1"})
    env = dict(os.environ, SBDART_AMD_SOCKET=sock, SBDART_AMD_NO_AUTOSTART="1")
    report = {}
    try:
        nrun, t_all = 0, 0.0
        for k in range(1, 6):
            block, _ = command_and_data(f"sbchk{k}")
            sw = Sweep(block)
            t0 = time.perf_counter()
            outs = sw.run(CLIENT, str(tmp_path / f"c{k}"), env=env)
            dt = time.perf_counter() - t0
            ref = sw.run_batch(HOST, str(tmp_path / f"b{k}"))
            assert outs == ref, f"sbchk{k}"
            _check_sweep("engine_served", f"sbchk{k}", outs)
            report[f"sbchk{k}"] = {"runs": len(sw), "seconds": dt, "ms_per_run": 1e3 * dt / len(sw)}
            nrun += len(sw)
            t_all += dt
        report["testruns_180"] = {"runs": nrun, "seconds": t_all, "ms_per_run": 1e3 * t_all / nrun,
                                  "how": "one `sbdart` client process per run, sequentially, as TestRuns/test_runs launches them"}
        assert nrun == 180
    finally:
        srv.stop()
        report["server_stderr_tail"] = (srv.p.stderr.read() or "")[-600:]
    print(json.dumps(report))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(report, open(os.path.join(ROOT, "gpurun_out", "serve_testruns.json"), "w"), indent=1)
    except OSError:
        pass


@pytest.mark.gpu
def test_a_large_run_is_made_whole_in_the_server(tmp_path):
    """A run whose work items would not fit a file (SBDART_AMD_BIG_MB, here 1 MB) comes back from the forked first phase with
    exit code 77 and is made whole in the server (compact form, gas terms on the device): the text of a process of its own."""
    _build()
    sock = str(tmp_path / "sock")
    d = _mkrun(str(tmp_path / "big"), "idatm=6 isat=0 wlinf=.25 wlsup=4.0 wlinc=.0005 nstr=16 iout=10")
    # ... a run of middle size (751 wavelengths: 10 MB of work items) is REHEARSED in the forked child -- exit code 78 -- and
    # made whole in the server too; a single wavelength goes through the file of work items
    mid = _mkrun(str(tmp_path / "mid"), "idatm=6 isat=0 wlinf=.25 wlsup=4.0 wlinc=.005 nstr=16 iout=10")
    small = _mkrun(str(tmp_path / "small"), "idatm=4 wlinf=.55 wlsup=.55 iout=10 sza=30")
    want = [subprocess.run([HOST], cwd=x, capture_output=True, text=True) for x in (d, mid, small)]
    assert all(w.returncode == 0 for w in want)
    srv = Server(sock, env={"SBDART_AMD_BIG_MB": "20"})
    try:
        env = dict(os.environ, SBDART_AMD_SOCKET=sock, SBDART_AMD_NO_AUTOSTART="1")
        for x, w in list(zip((d, mid, small), want)) * 2:
            p = subprocess.run([CLIENT], cwd=x, env=env, capture_output=True, text=True)
            assert p.returncode == 0 and p.stdout == w.stdout, (x, p.stderr[-500:])
        assert not os.path.exists(os.path.join(d, ".sbd_items"))
    finally:
        srv.stop()


def test_the_runtime_sees_only_the_devices_of_the_run(tmp_path):
    """SBD_DEVICES -> ROCR_VISIBLE_DEVICES before the first HIP call (VERDICT r05 #3b: on an 8-GPU node the runtime brings up
    eight agents for a run that uses one), the ordinals renumbered; the user's own restriction of the runtime is respected."""
    _build()
    d = _mkrun(str(tmp_path / "a"), GAS_REPORT)
    base = {k: v for k, v in os.environ.items() if k not in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "SBD_DEVICES")}

    def seen(**extra):
        p = subprocess.run([HOST], cwd=d, env=dict(base, SBD_SHOW_DEVICES="1", **extra), capture_output=True, text=True)
        line = [ln for ln in p.stderr.splitlines() if ln.startswith("sbdart_amd: devices:")][0]
        return line.split("devices: ")[1]

    assert seen() == "ROCR_VISIBLE_DEVICES=0 SBD_DEVICES=0"
    assert seen(SBD_DEVICES="2") == "ROCR_VISIBLE_DEVICES=2 SBD_DEVICES=0"
    assert seen(SBD_DEVICES="1,3,6") == "ROCR_VISIBLE_DEVICES=1,3,6 SBD_DEVICES=0,1,2"
    assert seen(SBD_DEVICES="all") == "ROCR_VISIBLE_DEVICES=(unset) SBD_DEVICES=all"
    assert seen(SBD_DEVICES="2", ROCR_VISIBLE_DEVICES="4,5,6") == "ROCR_VISIBLE_DEVICES=4,5,6 SBD_DEVICES=2"
    assert seen(SBD_DEVICES="1", HIP_VISIBLE_DEVICES="3,4") == "ROCR_VISIBLE_DEVICES=(unset) SBD_DEVICES=1"
