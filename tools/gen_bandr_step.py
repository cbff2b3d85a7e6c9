#!/usr/bin/env python3
"""Generates sbdart_amd/csrc/sbd_bandr_step.inc: the elimination sub-step of band_rows_kernel (sbd_bandr.hpp) as
inline asm, one pair of blocks per NSTR/2 = 17..20.

The kernel keeps a matrix row per lane: cur[n] (columns of x_lc) and nxt[n] (columns of x_lc+1) in registers.  Sub-step
J of a layer adds m * (pivot row's element) to every column right of J: columns J+1..n-1 of cur, all of nxt, and the
right-hand side -- a SUFFIX of one fixed sequence "column 1, 2, .., 2n-1, b".  So the sequence is written once and
sub-step J jumps into it (s_getpc/s_setpc into a table of branches): 3 instructions per column (two v_readlane of the
pivot lane's element, one v_fmac_f64 in place), the two v_readlane of column c+1 issued before the FMA of column c on
alternating SGPR pairs, so that the two wait states between a VALU write of an SGPR and its VALU read are filled with
work.  Written in C++ (J through a tree of scalar branches, static register indices) the compiler's PHI copies cost 40
v_mov_b64 per sub-step and out-of-place FMAs a second set of 80 registers (2 waves per SIMD need <= 256); same cure
as tools/gen_band1_take.py.

BandRowsStep<NN>::pick(cur, J, ak): ak = cur[J], J wave-uniform.
BandRowsStep<NN>::run(cur, nxt, b, m, P, J, last): the updates; P = pivot lane, last != 0: no x_lc+1 (last layer).
Run:  python tools/gen_bandr_step.py   (make runs it; the output is not kept in the repository)."""
import os

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sbdart_amd", "csrc", "sbd_bandr_step.inc")
PAIR = ("s[92:93]", "s[94:95]")
PLO = ("s92", "s94")
PHI = ("s93", "s95")
VBASE = 64       # cur[c] lives in v[VBASE+2c : +1], nxt[c] behind cur, then b: fixed registers, because AMDGPU inline asm
                 # cannot name the halves of a 64-bit operand (v_readlane_b32 needs them) -- "+{v[a:b]}" constraints


def vreg(i):
    return VBASE + 2 * i


def emit(nn):
    n = 2 * nn
    name = [f"v[{vreg(i)}:{vreg(i) + 1}]" for i in range(2 * n + 1)]     # cur, nxt, b
    B = 2 * n
    m, P, J, last = "%[m]", "%[P]", "%[J]", "%[last]"

    def rl(i, pair):
        return [f"v_readlane_b32 {PLO[pair]}, v{vreg(i)}, {P}", f"v_readlane_b32 {PHI[pair]}, v{vreg(i) + 1}, {P}"]

    L = [f"s_lshl_b32 s96, {J}, 2",
         "s_add_i32 s96, s96, 12",                 # bytes from the getpc result to the table
         "s_getpc_b64 s[98:99]",
         "s_add_u32 s98, s98, s96",
         "s_addc_u32 s99, s99, 0",
         "s_setpc_b64 s[98:99]"]
    for j in range(n):                             # sub-step j starts at column j+1
        L.append(f"s_branch .Lbr_p{j + 1}_%=" if j + 1 < n else "s_branch .Lbr_bound_%=")
    for c in range(1, n):                          # x_lc
        L.append(f".Lbr_b{c}_%=:")
        if c + 1 < n:
            L += rl(c + 1, (c + 1) & 1)
        else:
            L.append("s_nop 0")                    # (no v_readlane left to fill the second wait state)
        L.append(f"v_fmac_f64_e32 {name[c]}, {PAIR[c & 1]}, {m}")
    L.append(".Lbr_bound_%=:")
    L += [f"s_cmp_lg_u32 {last}, 0", "s_cbranch_scc1 .Lbr_rhs_%="]
    L += rl(n, n & 1)
    for c in range(n, 2 * n):                      # x_lc+1
        if c + 1 < 2 * n:
            L += rl(c + 1, (c + 1) & 1)
        else:
            L.append("s_nop 0")
        L.append(f"v_fmac_f64_e32 {name[c]}, {PAIR[c & 1]}, {m}")
    L.append(".Lbr_rhs_%=:")
    L += rl(B, 0) + ["s_nop 1", f"v_fmac_f64_e32 {name[B]}, {PAIR[0]}, {m}", "s_branch .Lbr_end_%="]
    for c in range(1, n):                          # entries: the first column's element, then into the sequence
        L.append(f".Lbr_p{c}_%=:")
        L += rl(c, c & 1)
        L += ["s_nop 1", f"s_branch .Lbr_b{c}_%="]
    L.append(".Lbr_end_%=:")
    text = "".join(f'        "{x}\\n"\n' for x in L)
    io = [f'"+{{{name[c]}}}"(cur[{c}])' for c in range(n)] + [f'"+{{{name[n + c]}}}"(nxt[{c}])' for c in range(n)] + [f'"+{{{name[B]}}}"(b)']
    run = (f"    SBD_DEVICE static void run(double (&cur)[{n}], double (&nxt)[{n}], double &b, double m, int P, int J, int last)\n    {{\n"
           f"        asm volatile(\n{text}"
           f"        : {', '.join(io)}\n        : [m] \"v\"(m), [P] \"s\"(P), [J] \"s\"(J), [last] \"s\"(last)\n"
           f"        : \"s92\", \"s93\", \"s94\", \"s95\", \"s96\", \"s98\", \"s99\", \"scc\");\n    }}\n")
    # ak = cur[J]
    K = [f"s_lshl_b32 s96, {'%1'}, 3", "s_add_i32 s96, s96, 12", "s_getpc_b64 s[98:99]", "s_add_u32 s98, s98, s96",
         "s_addc_u32 s99, s99, 0", "s_setpc_b64 s[98:99]"]
    for c in range(n):
        K += [f"v_mov_b64_e32 %0, {name[c]}", "s_branch .Lbk_end_%="]
    K.append(".Lbk_end_%=:")
    ktext = "".join(f'        "{x}\\n"\n' for x in K)
    kin = [f'"{{{name[c]}}}"(cur[{c}])' for c in range(n)]
    pick = (f"    SBD_DEVICE static void pick(const double (&cur)[{n}], int J, double &ak)\n    {{\n"
            f"        asm volatile(\n{ktext}"
            f"        : \"=&v\"(ak)\n        : \"s\"(J), {', '.join(kin)}\n"
            f"        : \"s96\", \"s98\", \"s99\", \"scc\");\n    }}\n")
    return f"template <>\nstruct BandRowsStep<{nn}> {{\n{pick}{run}}};\n"


def main():
    parts = ["// GENERATED by tools/gen_bandr_step.py -- do not edit.  See that file for the why.\n",
             "template <int NN> struct BandRowsStep;\n"]
    for nn in range(17, 21):
        parts.append(emit(nn))
    with open(OUT, "w") as f:
        f.write("".join(parts))
    print("wrote", os.path.normpath(OUT), sum(len(p) for p in parts), "bytes")


if __name__ == "__main__":
    main()
