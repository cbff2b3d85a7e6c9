"""The algebra of the layer kernel's Jacobi meetings (sbdart_amd/csrc/sbd_layer2.hpp, `meet`), restated in numpy:
columns are stored as w_j^1/2 x (true column) with the squared scale w_j beside them, a pair is rotated by
p' = p - tau q, q' = q + tau' p with tau, tau' from one quadratic that holds no square root of the scales, and the
scales leave after the last sweep.  Checked here, on the CPU: the stored columns converge to an orthogonal set whose
norms over the scales are B's singular values and whose accumulated transformation is orthogonal x diagonal -- and
that ONE common tangent for both columns (the first attempt, caught by the sbchk5 stdout test on the GPU) is not."""
import numpy as np


def sweeps(B, scaled=True, common_tangent=False, nsweep=12):
    n = B.shape[1]
    b = B.copy()
    V = np.eye(n)                      # accumulated transformation of the STORED columns: b = B V
    w = np.ones(n)
    for _ in range(nsweep):
        for p in range(n):
            for q in range(p + 1, n):
                a, bb, g = b[:, p] @ b[:, p], b[:, q] @ b[:, q], b[:, p] @ b[:, q]
                if g * g <= (2.2e-16) ** 2 * a * bb:
                    continue
                if not scaled:
                    d = bb - a
                    t = (2 * g if d >= 0 else -2 * g) / (abs(d) + np.hypot(d, 2 * g))
                    c = 1 / np.sqrt(1 + t * t)
                    G = np.array([[c, c * t], [-c * t, c]])
                elif common_tangent:
                    d = bb - a
                    t = (2 * g if d >= 0 else -2 * g) / (abs(d) + np.hypot(d, 2 * g))
                    G = np.array([[1.0, t], [-t, 1.0]])
                    w[p] *= 1 + t * t; w[q] *= 1 + t * t
                else:
                    e = w[p] * bb - w[q] * a
                    ww = w[p] * w[q]
                    u = 2 * g / (abs(e) + np.sqrt(e * e + 4 * g * g * ww))
                    sg = 1.0 if e >= 0 else -1.0
                    tau, taup = sg * u * w[p], sg * u * w[q]
                    G = np.array([[1.0, taup], [-tau, 1.0]])
                    q1 = 1 + u * u * ww
                    w[p] *= q1; w[q] *= q1
                b[:, [p, q]] = b[:, [p, q]] @ G
                V[:, [p, q]] = V[:, [p, q]] @ G
    return b, V, w


def _cases():
    rng = np.random.default_rng(7)
    for n in (4, 8, 16):
        for _ in range(5):
            A = rng.standard_normal((n, n))
            S = A @ A.T + n * np.eye(n)
            L = np.linalg.cholesky(S)
            A2 = rng.standard_normal((n, n))
            C = np.linalg.cholesky(A2 @ A2.T + n * np.eye(n))
            yield C.T @ L                                        # the kernel's B = C^T L


def test_scaled_rotations_are_rotations_of_the_true_columns():
    for B in _cases():
        n = B.shape[1]
        sv = np.sort(np.linalg.svd(B, compute_uv=False))
        b, V, w = sweeps(B, scaled=True)
        true = b / np.sqrt(w)                                   # the scales leave once, after the last sweep
        gram = true.T @ true
        off = gram - np.diag(np.diag(gram))
        assert np.abs(off).max() <= 1e-13 * np.abs(gram).max()
        assert np.allclose(np.sort(np.sqrt(np.diag(gram))), sv, rtol=1e-13, atol=0)
        Vt = V / np.sqrt(w)                                     # orthogonal x diagonal: V D^-1 is orthogonal
        assert np.abs(Vt.T @ Vt - np.eye(n)).max() <= 1e-13
        # ... and the same answer as the normalised rotations
        b0, V0, _ = sweeps(B, scaled=False)
        assert np.allclose(np.sort(np.linalg.norm(b0, axis=0)), np.sort(np.linalg.norm(true, axis=0)), rtol=1e-13)


def test_one_common_tangent_for_both_columns_is_not_a_rotation():
    B = next(iter(_cases()))
    n = B.shape[1]
    sv = np.sort(np.linalg.svd(B, compute_uv=False))
    b, V, w = sweeps(B, scaled=True, common_tangent=True)
    true = b / np.sqrt(w)
    gram = true.T @ true
    assert np.abs(gram - np.diag(np.diag(gram))).max() <= 1e-12 * np.abs(gram).max()      # the columns DO come out orthogonal
    assert not np.allclose(np.sort(np.sqrt(np.diag(gram))), sv, rtol=1e-6, atol=0)         # but they are not B's singular pairs
