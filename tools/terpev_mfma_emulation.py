#!/usr/bin/env python3
"""CPU emulation of sbdart_amd/csrc/experimental/sbd_terpev.hpp: the kernel's lane-level indexing with
v_mfma_f64_16x16x4_f64 emulated from its documented operand layouts (A: lane l holds A[l % 16][l / 16], B: B[l / 16][l % 16],
C/D register r of lane l: row l / 16 + 4 r, column l % 16) against TERPEV as the layer kernel forms it (sbd_layer2.hpp).
Checks the algebra (parity split, S = E11 + E21 / D = E11 - E21, GU's two columns per eigenvalue) and the chaining of the
first product's C/D registers as the second one's B operands.  Not a test of the compiled kernel (it never ran)."""
import numpy as np

NN, N, NUMU = 16, 32, 20
LANES = np.arange(64)
J, Q = LANES & 15, LANES >> 4


def mfma(a, b, c):
    """a, b: [64]; c: [64, 4] -> d [64, 4]"""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    A[J, Q] = a
    B[Q, J] = b
    full = A @ B                                   # [16 rows][16 cols]
    d = c.copy()
    for r in range(4):
        d[:, r] += full[Q + 4 * r, J]
    return d


def kernel(m, ylmc, cwt, ylmu, e11, e21, half_gl):
    """ylmc [jq][l], ylmu [iu][l], e11 / e21 [jq][j], half_gl [l] = 1/2 GL(l)  ->  GU [column][iu]"""
    gu = np.full((N, NUMU), np.nan)
    yw = np.zeros((2, 4, 64)); yu = np.zeros((2, 2, 4, 64)); l2 = np.zeros((2, 4, 64), dtype=int)
    for par in range(2):
        l1 = m + par + 2 * J
        for s in range(4):
            jq = 4 * s + Q
            yw[par, s] = np.where(l1 < N, ylmc[jq, np.minimum(l1, N - 1)] * cwt[jq], 0.0)
        for r in range(4):
            l = m + par + 2 * (4 * r + Q)
            l2[par, r] = l
            for tile in range(2):
                iu = 16 * tile + J
                ok = (l < N) & (iu < NUMU)
                yu[tile, par, r] = np.where(ok, ylmu[np.minimum(iu, NUMU - 1), np.minimum(l, N - 1)], 0.0)
    te = np.zeros((64, 4)); to = np.zeros((64, 4))
    for s in range(4):
        jq = 4 * s + Q
        te = mfma(yw[0, s], e11[jq, J] + e21[jq, J], te)
        to = mfma(yw[1, s], e11[jq, J] - e21[jq, J], to)
    for tile in range(2):
        pe = np.zeros((64, 4)); po = np.zeros((64, 4))
        for r in range(4):
            ae = np.where(l2[0, r] < N, half_gl[np.minimum(l2[0, r], N - 1)], 0.0) * yu[tile, 0, r]
            ao = np.where(l2[1, r] < N, half_gl[np.minimum(l2[1, r], N - 1)], 0.0) * yu[tile, 1, r]
            pe = mfma(ae, te[:, r], pe)
            po = mfma(ao, to[:, r], po)
        for r in range(4):
            iu = 16 * tile + Q + 4 * r
            for lane in range(64):
                if iu[lane] < NUMU:
                    gu[J[lane] + NN, iu[lane]] = pe[lane, r] + po[lane, r]
                    gu[NN - 1 - J[lane], iu[lane]] = po[lane, r] - pe[lane, r]
    return gu


def layer_kernel_terpev(m, ylmc, cwt, ylmu, e11, e21, half_gl):
    gu = np.zeros((N, NUMU))
    for me in range(1, NN + 1):
        a, b = e11[:, me - 1], e21[:, me - 1]
        wkp = np.zeros(N); wkn = np.zeros(N)
        for l in range(m, N):
            y = ylmc[:NN, l] * cwt[:NN]
            if (l - m) % 2 == 0:
                sp = np.sum(y * a + y * b); sn = np.sum(y * (-b) + y * (-a))
            else:
                sp = np.sum(y * a - y * b); sn = np.sum(y * (-b) + (-y) * (-a))
            wkp[l] = half_gl[l] * sp; wkn[l] = half_gl[l] * sn
        for iu in range(NUMU):
            gu[me + NN - 1, iu] = np.sum(wkp * ylmu[iu])
            gu[NN - me, iu] = np.sum(wkn * ylmu[iu])
    return gu


if __name__ == "__main__":
    rng = np.random.default_rng(7)
    worst = 0.0
    for m in (0, 1, 2, 5, 16, 30, 31):
        ylmc = rng.normal(size=(N, N)); cwt = rng.uniform(.01, .2, size=N); ylmu = rng.normal(size=(NUMU, N))
        e11 = rng.normal(size=(NN, NN)); e21 = rng.normal(size=(NN, NN)); half_gl = rng.normal(size=N)
        got = kernel(m, ylmc, cwt, ylmu, e11, e21, half_gl)
        want = layer_kernel_terpev(m, ylmc, cwt, ylmu, e11, e21, half_gl)
        assert np.isfinite(got).all()
        err = float(np.abs(got - want).max() / np.abs(want).max())
        worst = max(worst, err)
        print(f"mode {m:2d}: worst |difference| / max = {err:.2e}")
    assert worst < 1e-13
    print("emulation agrees with the layer kernel's TERPEV")
