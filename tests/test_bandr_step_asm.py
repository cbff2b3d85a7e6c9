"""The generated sub-step of band_rows_kernel (tools/gen_bandr_step.py -> sbd_bandr_step.inc), checked on the CPU by
walking its control flow: for every NSTR/2 = 17..20, every sub-step J and both kinds of layer (with and without the
columns of x_lc+1), the instructions that would execute are followed from the computed jump's table entry to the end
label, and what they do is compared with what the kernel needs --

  * exactly the columns right of J are updated, each once, in order, then the right-hand side;
  * every FMA takes its SGPR pair from the two v_readfirstlane of the SAME column's register halves, fetched while EXEC
    was narrowed to the pivot lane, and runs with EXEC restored;
  * at least two wait states lie between the VALU write of an SGPR and its VALU read (the hazard the compiler would
    pad, but does not see inside inline asm);
  * the jump arithmetic matches the table: 4-byte entries behind three 4-byte instructions for the update sequence,
    8-byte cases for the column pick.
No GPU, no compiler: the generator's emit() is imported and its text parsed."""
import importlib.util
import os
import re

import pytest

from conftest import ROOT

_spec = importlib.util.spec_from_file_location("gen_bandr_step", os.path.join(ROOT, "tools", "gen_bandr_step.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)


def _asm_blocks(text):
    """The two asm volatile(...) instruction lists of one BandRowsStep<NN>: pick, run."""
    blocks = []
    for m in re.finditer(r"asm volatile\(\n(.*?)\n\s*:", text, re.S):
        blocks.append([ln.strip()[1:-3] for ln in m.group(1).splitlines() if ln.strip().startswith('"')])
    assert len(blocks) == 2
    return blocks


def _walk(lines, J, last, n):
    labels = {ln[:-1]: i for i, ln in enumerate(lines) if ln.endswith(":")}
    assert lines[:6] == ["s_lshl_b32 s96, %[J], 2", "s_add_i32 s96, s96, 12", "s_getpc_b64 s[98:99]", "s_add_u32 s98, s98, s96",
                         "s_addc_u32 s99, s99, 0", "s_setpc_b64 s[98:99]"]
    # s_getpc returns the address of the instruction behind it: three 4-byte scalar instructions (12) to the table, whose
    # entries are one 4-byte s_branch each (J << 2)
    table = lines[6:6 + n]
    assert all(t.startswith("s_branch ") for t in table)
    pc = 6 + J
    exec_narrow = False
    sgpr = {}                      # number -> (vgpr number it was read from, step of the write)
    step = 0
    done = []
    scc = False
    for _ in range(100000):
        ln = lines[pc]
        if ln.endswith(":"):
            if ln == ".Lbr_end_%=:":
                return done
            pc += 1
            continue
        step += 1
        op = ln.split()[0]
        if op == "s_branch":
            pc = labels[ln.split()[1]]
            continue
        if op == "s_cbranch_scc1":
            pc = labels[ln.split()[1]] if scc else pc + 1
            continue
        if op == "s_cmp_lg_u32":
            assert ln == "s_cmp_lg_u32 %[last], 0"
            scc = last != 0
        elif op == "s_nop":
            step += int(ln.split()[1])            # s_nop N: N + 1 wait states
        elif op == "s_mov_b64":
            assert ln in ("s_mov_b64 exec, %[mask]", "s_mov_b64 exec, -1"), ln
            exec_narrow = ln.endswith("%[mask]")
        elif op == "v_readfirstlane_b32":
            m = re.fullmatch(r"v_readfirstlane_b32 s(\d+), v(\d+)", ln)
            assert m and exec_narrow, (ln, "EXEC must be narrowed to the pivot lane")
            sgpr[int(m.group(1))] = (int(m.group(2)), step)
        elif op == "v_fmac_f64_e32":
            m = re.fullmatch(r"v_fmac_f64_e32 v\[(\d+):(\d+)\], s\[(\d+):(\d+)\], %\[m\]", ln)
            assert m and not exec_narrow, (ln, "the update runs on every lane")
            v0, v1, s0, s1 = (int(x) for x in m.groups())
            assert v1 == v0 + 1 and s1 == s0 + 1 and s0 % 2 == 0 and v0 % 2 == 0
            assert sgpr[s0][0] == v0 and sgpr[s1][0] == v1, (J, ln, "the pair holds another column's element")
            assert step - max(sgpr[s0][1], sgpr[s1][1]) - 1 >= 2, (J, ln, "VALU write of an SGPR -> VALU read: two wait states")
            done.append(v0)
        else:
            raise AssertionError("unexpected instruction " + ln)
        pc += 1
    raise AssertionError("no end label reached")


@pytest.mark.parametrize("nn", [17, 18, 19, 20])
def test_update_sequence_of_every_sub_step(nn):
    n = 2 * nn
    pick, run = _asm_blocks(gen.emit(nn))
    col = lambda c: gen.vreg(c)                       # register pair of column c (cur, nxt, then the right-hand side)
    for last in (0, 1):
        for J in range(n):
            want = [col(c) for c in range(J + 1, n if last else 2 * n)] + [col(2 * n)]
            assert _walk(run, J, last, n) == want, (nn, J, last)
    # the column pick: 8-byte cases (v_mov_b64_e32 + s_branch) behind three scalar instructions
    assert pick[:6] == ["s_lshl_b32 s96, %1, 3", "s_add_i32 s96, s96, 12", "s_getpc_b64 s[98:99]", "s_add_u32 s98, s98, s96",
                        "s_addc_u32 s99, s99, 0", "s_setpc_b64 s[98:99]"]
    cases = pick[6:-1]
    assert len(cases) == 2 * n and pick[-1] == ".Lbk_end_%=:"
    for c in range(n):
        assert cases[2 * c] == f"v_mov_b64_e32 %0, v[{col(c)}:{col(c) + 1}]" and cases[2 * c + 1] == "s_branch .Lbk_end_%="


def test_register_plan_fits_two_waves_per_simd():
    """cur, nxt and the right-hand side on fixed registers v[VBASE ...]: below 256 at NSTR 40, clear of the SGPRs the
    groups use (s[SBASE, SBASE + 2 G)) and of the jump's scratch (s96, s98, s99)."""
    top = gen.vreg(2 * 40) + 1
    assert gen.VBASE % 2 == 0 and top < 256
    assert gen.SBASE % 2 == 0 and gen.SBASE + 2 * gen.G <= 96
