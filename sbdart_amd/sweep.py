"""Python-3 sweep runner: the RunRT command-block semantics over the `sbdart` process boundary.

What RunRT's GenInput (RunRT/GenInput.py:43-147, Python 2.7, not importable) does with a command
block, restated for Python 3 and for any executable that honours the reference's outer contract
(cwd holds INPUT, result text on stdout -- `sbdart_amd/bin/sbdart_amd` or the reference itself):

    TCLOUD=0;10;100       a `;` list makes a cycle: the FIRST cycle varies fastest
    WLINF=0.5;0.8         second cycle
    WLSUP=0.5;0.8 &       trailing `&`: covariant -- steps together with the cycle above it
    ALBCON=0.5            no `;`: constant of every run        # comments start with '#'

and with the text that comes back (RtReader.py's token layouts): IOUT=10 one line per run
(WLINF WLSUP FFEW TOPDN TOPUP TOPDIR BOTDN BOTUP BOTDIR), IOUT=1 a '"tbf' block of NWL rows
(WL FFV TOPDN TOPUP TOPDIR BOTDN BOTUP BOTDIR), IOUT=11 NZ rows (Z P FXDN FXUP FXDIR DFDZ HEAT).
"""
from __future__ import annotations

import os
import subprocess
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple


class SweepError(ValueError):
    pass


class Sweep:
    def __init__(self, command_block: str):
        self.cycles: List["OrderedDict[str, List[str]]"] = []   # one dict per nested cycle, leading variable first
        self.constants: "OrderedDict[str, str]" = OrderedDict()
        self.iout = 10
        for raw in command_block.split("\n"):
            line = raw.split("#")[0].strip()
            if not line or "=" not in line:
                continue
            name, rhs = (x.strip() for x in line.split("=", 1))
            covariant = rhs.endswith("&")
            if covariant:
                rhs = rhs[:-1].strip()
            if name.upper() == "IOUT":
                self.iout = int(rhs.split(";")[0])
                rhs = rhs.split(";")[0]            # one output format per sweep
            if ";" in rhs:
                values = [v.strip().replace(" ", "_") for v in rhs.split(";")]
                if covariant:
                    if not self.cycles:
                        raise SweepError(f"covariant variable {name} has no cycle to follow")
                    if len(values) != len(next(iter(self.cycles[-1].values()))):
                        raise SweepError(f"Number of elements in covariant variable, {name}, is incorrect.")
                    self.cycles[-1][name] = values
                else:
                    self.cycles.append(OrderedDict([(name, values)]))
            else:
                self.constants[name] = rhs

    @property
    def shape(self) -> Tuple[int, ...]:
        return tuple(len(next(iter(c.values()))) for c in self.cycles)

    def __len__(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n

    def inputs(self, iteration: int) -> Tuple[str, List[str]]:
        """(namelist body of this iteration, leading "NAME=value" of every cycle); iteration 0..len-1,
        first cycle fastest."""
        if not 0 <= iteration < len(self):
            raise IndexError(iteration)
        body, lead = "", []
        it = iteration
        for cyc, n in zip(self.cycles, self.shape):
            i = it % n
            it //= n
            for k, (name, values) in enumerate(cyc.items()):
                if name[0].isalpha():
                    body += f"{name}={values[i]}\n"
                if k == 0:
                    lead.append(f"{name}={values[i]}")
        for name, v in self.constants.items():
            body += f"{name}={v}\n"
        return body, lead

    def run(self, exe: str, workdir: str, env: Optional[Dict[str, str]] = None,
            prepare=None) -> List[str]:
        """Run every iteration in its own directory workdir/run%04d (the reference's cwd-relative files
        make concurrent runs in one directory impossible); returns the stdout texts in iteration order.
        `prepare(directory, namelist_body)` may drop extra files (e.g. an optics file) beside INPUT."""
        outs = []
        for it in range(len(self)):
            d = os.path.join(workdir, f"run{it:04d}")
            os.makedirs(d, exist_ok=True)
            body, _ = self.inputs(it)
            with open(os.path.join(d, "INPUT"), "w") as f:
                f.write("\n &INPUT\n" + body + " /\n")
            if prepare is not None:
                prepare(d, body)
            p = subprocess.run([exe], cwd=d, env=env, capture_output=True, text=True)
            if p.returncode != 0:
                raise RuntimeError(f"{exe} failed in {d}: {p.stderr.strip()}")
            outs.append(p.stdout)
        return outs


    def run_batch(self, exe: str, workdir: str, env: Optional[Dict[str, str]] = None, prepare=None) -> List[str]:
        """The same runs through ONE process: `exe --batch LIST` (sbdart_amd's batch mode: the GPU runtime comes up
        once, engines are kept and found again by configuration, phase 1 of the runs -- INPUT to work items -- in a
        pool of forked children).  Every run keeps its own directory and finds its text in SBDART.stdout there;
        returns those texts in iteration order, byte for byte what `run` returns."""
        dirs = []
        for it in range(len(self)):
            d = os.path.join(workdir, f"run{it:04d}")
            os.makedirs(d, exist_ok=True)
            body, _ = self.inputs(it)
            with open(os.path.join(d, "INPUT"), "w") as f:
                f.write("\n &INPUT\n" + body + " /\n")
            if prepare is not None:
                prepare(d, body)
            dirs.append(os.path.abspath(d))
        return run_directories(exe, dirs, workdir, env)


def run_directories(exe: str, dirs: Sequence[str], workdir: str, env: Optional[Dict[str, str]] = None) -> List[str]:
    """`exe --batch LIST` over run directories that already hold their INPUT; the texts of SBDART.stdout."""
    lst = os.path.join(workdir, "batch.list")
    with open(lst, "w") as f:
        f.write("\n".join(dirs) + "\n")
    for d in dirs:
        for name in ("SBDART.stdout", "SBDART.stderr"):
            if os.path.exists(os.path.join(d, name)):
                os.remove(os.path.join(d, name))
    p = subprocess.run([exe, "--batch", lst], cwd=workdir, env=env, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError(f"{exe} --batch failed with code {p.returncode}: {p.stderr.strip()[-2000:]} {p.stdout.strip()[-500:]}")
    outs = []
    for d in dirs:
        with open(os.path.join(d, "SBDART.stdout")) as f:
            outs.append(f.read())
    return outs


FLUX_KEYS = ("TOPDN", "TOPUP", "TOPDIR", "BOTDN", "BOTUP", "BOTDIR")


def parse_iout10(text: str) -> Dict[str, float]:
    t = [float(x) for x in text.split()]
    if len(t) < 9:
        raise SweepError("IOUT=10 record needs 9 numbers")
    out = {"WLINF": t[0], "WLSUP": t[1], "FFEW": t[2]}
    out.update(zip(FLUX_KEYS, t[3:9]))
    out["TOPFLUX"] = out["TOPDN"] - out["TOPUP"]          # the derived keys RtReader adds (RtReader.py:115-121)
    out["BOTFLUX"] = out["BOTDN"] - out["BOTUP"]
    out["ABSORPTION"] = out["TOPFLUX"] - out["BOTFLUX"]
    return out


def parse_iout1(text: str) -> Dict[str, List[float]]:
    tok = text.split()
    if not tok or tok[0] != '"tbf':
        raise SweepError('IOUT=1 output starts with "tbf')
    nwl = int(tok[1])
    v = [float(x) for x in tok[2:2 + 8 * nwl]]
    cols = {k: v[i::8] for i, k in enumerate(("WL", "FFV") + FLUX_KEYS)}
    cols["TOPFLUX"] = [a - b for a, b in zip(cols["TOPDN"], cols["TOPUP"])]
    cols["BOTFLUX"] = [a - b for a, b in zip(cols["BOTDN"], cols["BOTUP"])]
    cols["ABSORPTION"] = [a - b for a, b in zip(cols["TOPFLUX"], cols["BOTFLUX"])]
    return cols


def parse_iout11(text: str) -> Dict[str, List[float]]:
    tok = text.split()
    nz = int(tok[0])
    v = [float(x) for x in tok[2:2 + 7 * nz]]
    cols = {k: v[i::7] for i, k in enumerate(("Z", "P", "FXDN", "FXUP", "FXDIR", "DFDZ", "HEAT"))}
    cols["PHIDW"] = float(tok[1])
    return cols
