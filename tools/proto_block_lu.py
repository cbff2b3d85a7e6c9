#!/usr/bin/env python3
"""Numerical prototype (CPU, numpy) for the band kernel's layer-block elimination variants.

Assembles SETMTX's boundary-value matrix (disort.f:2702-2994) in block form from the oracle's GC/KK of a
record, takes b = A @ LL_oracle (long double) so that the exact solution is known, and solves it
  (a) by layer-block LU with partial pivoting + ordinary back-substitution (what band4/backsolve4 do),
  (b) by the same elimination finished inside each block (Gauss-Jordan: U0^-1 [U1 | y] stored only),
  (c) like (b) but without any stored factor: x_1 from the running product of the -U1' blocks.
Prints the worst error of each against the known solution, relative to max|LL|.
Developer tool: imports the oracle, never part of the product path.
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402
from sbdart_amd.records import read_records  # noqa: E402


def blocks(rec, dbg):
    n, L = rec.nstr, rec.nlyr
    nn = n // 2
    gc, kk = dbg["gc"], dbg["kk"]            # [lc][i][j], [lc][j]
    dither = pyoracle.lib().sbdo_dither()
    ss = np.where(rec.ssalb == 1.0, 1.0 - dither, rec.ssalb)
    f = rec.pmom[:, n] if rec.pmom.shape[1] > n else np.zeros(L)
    dt = np.maximum(rec.dtauc, 0.0)
    dtaucp = (1.0 - f * ss) * dt
    # ncut (no LYRCUT when plank)
    ncut = L
    if not rec.plank and L > 1:
        abscum = np.cumsum((1.0 - ss) * dt)
        hit = np.nonzero(abscum >= 10.0)[0]
        if len(hit):
            ncut = int(hit[0]) + 1
    ek = np.exp(kk[:, :nn] * dtaucp[:, None])      # EK(iq, lc), iq <= nn (kk < 0)
    cmu, cwt = np.zeros(nn), np.zeros(nn)
    pyoracle.lib().sbdo_qgausn(nn, pyoracle._p(cmu), pyoracle._p(cwt))
    top = gc[0][nn - 1::-1, :].copy()                # rows GC(nn+1-r, j, 1), r = 1..nn
    top[:, :nn] *= ek[0][None, :]                    # exp(KK(j,1) taucpr(1)) for j <= nn
    A, B = [], []
    for lc in range(ncut - 1):
        fa = np.ones(n)
        fa[nn:] = ek[lc][::-1]                       # EK(n+1-j, lc), j > nn
        fb = np.ones(n)
        fb[:nn] = ek[lc + 1]                         # EK(j, lc+1), j <= nn
        A.append(gc[lc] * fa[None, :])
        B.append(-gc[lc + 1] * fb[None, :])
    g = gc[ncut - 1]
    refl = (ncut == L)
    sb = np.zeros(n)
    if refl:
        for k in range(nn):
            sb += cwt[k] * cmu[k] * rec.albedo * g[nn - 1 - k, :]
    bot = g[nn:, :] - 2.0 * sb[None, :]
    fbot = np.ones(n)
    fbot[nn:] = ek[ncut - 1][::-1]
    bot = bot * fbot[None, :]
    return top, A, B, bot, ncut


def dense(top, A, B, bot, n):
    nn = n // 2
    ncut = len(A) + 1
    N = n * ncut
    M = np.zeros((N, N))
    M[:nn, :n] = top
    for lc in range(ncut - 1):
        r0 = nn + lc * n
        M[r0:r0 + n, lc * n:(lc + 1) * n] = A[lc]
        M[r0:r0 + n, (lc + 1) * n:(lc + 2) * n] = B[lc]
    M[N - nn:, N - n:] = bot
    return M


def adjoint_top(top, A, B, bot, b, n, c):
    """f = c^T x_1 with c's row riding through the elimination (never a pivot candidate)."""
    nn = n // 2
    ncut = len(A) + 1
    carry = np.hstack([top, np.zeros((nn, n)), b[:nn, None]])
    F = np.hstack([c, np.zeros(n), [0.0]])
    for lc in range(ncut):
        if lc < ncut - 1:
            rows = np.hstack([A[lc], B[lc], b[nn + lc * n: nn + (lc + 1) * n, None]])
        else:
            rows = np.hstack([bot, np.zeros((nn, n)), b[len(b) - nn:, None]])
        W = np.vstack([carry, rows])
        live = list(range(W.shape[0]))
        for J in range(n):
            p = max(live, key=lambda r: abs(W[r, J]))
            live.remove(p)
            piv = W[p, J]
            for r in live:
                m = W[r, J] / piv
                W[r, J:] -= m * W[p, J:]
                W[r, J] = 0.0
            m = F[J] / piv
            F[J:] -= m * W[p, J:]
            F[J] = 0.0
        if lc < ncut - 1:
            carry = np.hstack([W[live][:, n:2 * n], np.zeros((nn, n)), W[live][:, 2 * n:2 * n + 1]])
            F = np.hstack([F[n:2 * n], np.zeros(n), F[2 * n:]])
    return -F[2 * n]


def block_solve(top, A, B, bot, b, n, mode):
    """mode 'lu': store U0,U1,y ; 'gj': store U1' = U0^-1 U1, y' ; 'prod': x_1 via running product."""
    nn = n // 2
    ncut = len(A) + 1
    carry = np.hstack([top, np.zeros((nn, n)), b[:nn, None]])
    store = []
    P = np.eye(n)
    s = np.zeros(n)
    for lc in range(ncut):
        if lc < ncut - 1:
            rows = np.hstack([A[lc], B[lc], b[nn + lc * n: nn + (lc + 1) * n, None]])
        else:
            rows = np.hstack([bot, np.zeros((nn, n)), b[len(b) - nn:, None]])
        W = np.vstack([carry, rows])
        live = list(range(W.shape[0]))
        retired = []
        for J in range(n):
            p = max(live, key=lambda r: abs(W[r, J]))
            live.remove(p)
            piv = W[p, J]
            for r in live:
                m = W[r, J] / piv
                W[r, J:] -= m * W[p, J:]
                W[r, J] = 0.0
            retired.append(p)
        U = W[retired]
        U0, U1, y = U[:, :n], U[:, n:2 * n], U[:, 2 * n]
        if mode == "lu":
            store.append((U0.copy(), U1.copy(), y.copy()))
        else:
            # finish inside the block: rows scaled by 1/pivot, upward eliminations
            R = np.hstack([U1, y[:, None]])
            U0 = U0.copy()
            for J in range(n - 1, -1, -1):
                R[J] /= U0[J, J]
                for i in range(J):
                    R[i] -= U0[i, J] * R[J]
            store.append((R[:, :n].copy(), R[:, n].copy()))
        if lc < ncut - 1:
            carry = np.hstack([W[live][:, n:2 * n], np.zeros((nn, n)), W[live][:, 2 * n:2 * n + 1]])
    x = np.zeros((ncut, n))
    if mode == "lu":
        xn = np.zeros(n)
        for lc in range(ncut - 1, -1, -1):
            U0, U1, y = store[lc]
            r = y - U1 @ xn
            xq = np.zeros(n)
            for J in range(n - 1, -1, -1):
                xq[J] = (r[J] - U0[J, J + 1:] @ xq[J + 1:]) / U0[J, J]
            x[lc] = xq
            xn = xq
    elif mode == "gj":
        xn = np.zeros(n)
        for lc in range(ncut - 1, -1, -1):
            M, y = store[lc]
            x[lc] = y - M @ xn
            xn = x[lc]
    else:
        # x_ncut directly, x_1 = sum_k P_{k-1} y'_k + P_{ncut-1} x_ncut with P_k = prod_{j<=k} (-U1'_j)
        x[:] = np.nan
        x[ncut - 1] = store[ncut - 1][1]
        P = np.eye(n)
        s = np.zeros(n)
        for lc in range(ncut - 1):
            M, y = store[lc]
            s = s + P @ y
            P = -(P @ M)
        x[0] = s + P @ x[ncut - 1] if ncut > 1 else x[0]
    return x


def main():
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.sbdrec"))) + \
        sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "illcond", "*.sbdrec")))
    only = sys.argv[1:] or None
    for fn in files:
        name = os.path.basename(fn)
        if only and not any(o in name for o in only):
            continue
        recs = read_records(fn)
        if recs[0].nstr > 16:
            continue
        worst = dict(lu=0.0, gj=0.0, prod=0.0, prod_top=0.0, lu_top=0.0, gj_top=0.0, adj_f=0.0, lu_f=0.0)
        pmax = 0.0
        step = max(1, len(recs) // 40)
        for rec in recs[::step]:
            o = pyoracle.disort(rec, debug_mode=0)
            if o["status"] & (8 | 16 | 32):
                continue
            dbg = o["dbg"]
            n = rec.nstr
            top, A, B, bot, ncut = blocks(rec, dbg)
            M = dense(top, A, B, bot, n)
            ll = dbg["ll"][:ncut].reshape(-1)
            b = np.asarray(M.astype(np.longdouble) @ ll.astype(np.longdouble), dtype=np.float64)
            scale = np.abs(ll).max() or 1.0
            rng = np.random.default_rng(1)
            c = rng.uniform(0.1, 1.0, n)
            fref = float(c.astype(np.longdouble) @ ll[:n].astype(np.longdouble))
            fscale = float(np.abs(c) @ np.abs(ll[:n])) or 1.0
            worst["adj_f"] = max(worst["adj_f"], abs(adjoint_top(top, A, B, bot, b, n, c) - fref) / fscale)
            xlu = block_solve(top, A, B, bot, b, n, "lu")
            worst["lu_f"] = max(worst["lu_f"], abs(c @ xlu[0] - fref) / fscale)
            for mode in ("lu", "gj", "prod"):
                x = block_solve(top, A, B, bot, b, n, mode)
                ref = ll.reshape(ncut, n)
                if mode != "prod":
                    worst[mode] = max(worst[mode], np.abs(x - ref).max() / scale)
                    worst[mode + "_top"] = max(worst[mode + "_top"], np.abs(x[0] - ref[0]).max() / scale)
                else:
                    worst["prod"] = max(worst["prod"], np.abs(x[ncut - 1] - ref[ncut - 1]).max() / scale)
                    worst["prod_top"] = max(worst["prod_top"], np.abs(x[0] - ref[0]).max() / scale)
        print(f"{name:32s} n={recs[0].nstr:2d} " + " ".join(f"{k}={v:.1e}" for k, v in worst.items()))


if __name__ == "__main__":
    main()
