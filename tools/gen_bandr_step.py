#!/usr/bin/env python3
"""Generates sbdart_amd/csrc/sbd_bandr_step.inc: the elimination sub-step of band_rows_kernel (sbd_bandr.hpp) as
inline asm, one pair of blocks per NSTR/2 = 17..20.

The kernel keeps a matrix row per lane: cur[n] (columns of x_lc) and nxt[n] (columns of x_lc+1) in registers.  Sub-step
J of a layer adds m * (pivot row's element) to every column right of J: columns J+1..n-1 of cur, all of nxt, and the
right-hand side -- a SUFFIX of one fixed sequence "column 1, 2, .., 2n-1, b".  So the sequence is written once, in
place on fixed registers, and sub-step J jumps into it (s_getpc/s_setpc into a table of branches).  A column costs
three instructions: two v_readfirstlane_b32 of the pivot lane's element into an SGPR pair, one v_fmac_f64 with that
pair as an operand.  The elements of G = 16 columns are fetched together with EXEC narrowed to the pivot lane
(v_readfirstlane then reads that lane): measured on gfx950 (tools/microbench/valu_rates.hip, two waves per SIMD), a
v_readlane_b32 with an SGPR lane select takes 1.7 issue slots of 4 cycles, v_readfirstlane_b32 -- like a constant or M0
lane select -- one, an FMA one; fetching a group ahead also keeps the two wait states between a VALU write of an SGPR
and its VALU read filled.  A sub-step that starts inside a group enters through a stub that fetches the rest of the
group.  Written in C++ (J through a tree of scalar branches, static register indices) the compiler's PHI copies cost
40 v_mov_b64 per sub-step and out-of-place FMAs a second set of 80 registers (2 waves per SIMD need <= 256); same
cure as tools/gen_band1_take.py.

BandRowsStep<NN>::pick(cur, J, ak): ak = cur[J], J wave-uniform.
BandRowsStep<NN>::run(cur, nxt, b, m, mask, J, last): the updates; mask = 1 << pivot lane, last != 0: no x_lc+1 (the
last layer).  EXEC must be all ones on entry (the kernel's sub-step loop is wave-uniform) and is on exit.
Run:  python tools/gen_bandr_step.py   (make runs it; the output is not kept in the repository)."""
import os

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sbdart_amd", "csrc", "sbd_bandr_step.inc")
G = 16           # columns per group: the pivot lane's elements of a group are fetched with EXEC narrowed to that lane
SBASE = 60       # ... into s[SBASE : SBASE + 2 G)
VBASE = 64       # cur[c] lives in v[VBASE+2c : +1], nxt[c] behind cur, then b: fixed registers, because AMDGPU inline asm
                 # cannot name the halves of a 64-bit operand (v_readfirstlane_b32 needs them) -- "+{v[a:b]}" constraints


def vreg(i):
    return VBASE + 2 * i


def emit(nn):
    n = 2 * nn
    name = [f"v[{vreg(i)}:{vreg(i) + 1}]" for i in range(2 * n + 1)]     # cur, nxt, b
    B = 2 * n
    m, J, last, mask = "%[m]", "%[J]", "%[last]", "%[mask]"

    def pair(i):                                   # SGPR pair of the i-th column of a group
        return f"s[{SBASE + 2 * i}:{SBASE + 2 * i + 1}]"

    def fetch(cols):                               # the pivot lane's elements of these columns, into the group's pairs
        out = [f"s_mov_b64 exec, {mask}"]
        for c in cols:
            i = c2i[c]
            out += [f"v_readfirstlane_b32 s{SBASE + 2 * i}, v{vreg(c)}", f"v_readfirstlane_b32 s{SBASE + 2 * i + 1}, v{vreg(c) + 1}"]
        out.append("s_mov_b64 exec, -1")
        return out

    # groups of G columns: x_lc's columns 1..n-1, then x_lc+1's n..2n-1
    groups = [list(range(g, min(g + G, n))) for g in range(1, n, G)] + [list(range(g, min(g + G, 2 * n))) for g in range(n, 2 * n, G)]
    c2i = {c: k for grp in groups for k, c in enumerate(grp)}
    L = [f"s_lshl_b32 s96, {J}, 2",
         "s_add_i32 s96, s96, 12",                 # bytes from the getpc result to the table
         "s_getpc_b64 s[98:99]",
         "s_add_u32 s98, s98, s96",
         "s_addc_u32 s99, s99, 0",
         "s_setpc_b64 s[98:99]"]
    for j in range(n):                             # sub-step j starts at column j+1
        L.append(f"s_branch .Lbr_p{j + 1}_%=" if j + 1 < n else "s_branch .Lbr_bound_%=")
    for grp in groups:
        if grp[0] == n:
            L.append(".Lbr_bound_%=:")
            L += [f"s_cmp_lg_u32 {last}, 0", "s_cbranch_scc1 .Lbr_rhs_%="]
        elif grp[0] < n:
            L.append(f".Lbr_p{grp[0]}_%=:")       # (entering at the first column of a group needs no stub)
        L += fetch(grp)
        if len(grp) == 1:
            L.append("s_nop 1")
        for c in grp:
            if c < n:
                L.append(f".Lbr_b{c}_%=:")
            L.append(f"v_fmac_f64_e32 {name[c]}, {pair(c2i[c])}, {m}")
    L.append(".Lbr_rhs_%=:")
    L += [f"s_mov_b64 exec, {mask}", f"v_readfirstlane_b32 s{SBASE}, v{vreg(B)}", f"v_readfirstlane_b32 s{SBASE + 1}, v{vreg(B) + 1}",
          "s_mov_b64 exec, -1", "s_nop 1", f"v_fmac_f64_e32 {name[B]}, {pair(0)}, {m}", "s_branch .Lbr_end_%="]
    for grp in groups:                             # entries inside a group: the rest of the group's elements, then into the sequence
        if grp[0] >= n:
            break
        for c in grp[1:]:
            L.append(f".Lbr_p{c}_%=:")
            L += fetch([x for x in grp if x >= c])
            L += ["s_nop 1", f"s_branch .Lbr_b{c}_%="]
    L.append(".Lbr_end_%=:")
    text = "".join(f'        "{x}\\n"\n' for x in L)
    io = [f'"+{{{name[c]}}}"(cur[{c}])' for c in range(n)] + [f'"+{{{name[n + c]}}}"(nxt[{c}])' for c in range(n)] + [f'"+{{{name[B]}}}"(b)']
    run = (f"    SBD_DEVICE static void run(double (&cur)[{n}], double (&nxt)[{n}], double &b, double m, unsigned long long mask, int J, int last)\n    {{\n"
           f"        asm volatile(\n{text}"
           f"        : {', '.join(io)}\n        : [m] \"v\"(m), [mask] \"s\"(mask), [J] \"s\"(J), [last] \"s\"(last)\n"
           f"        : {', '.join(chr(34) + 's' + str(SBASE + q) + chr(34) for q in range(2 * G))}, \"s96\", \"s98\", \"s99\", \"scc\");\n    }}\n")
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
