"""The design of the experimental TERPEV kernel (sbdart_amd/csrc/experimental/sbd_terpev.hpp: written in round 6, never run --
GPU access closed) replayed on the CPU: its lane-level indexing with v_mfma_f64_16x16x4_f64 emulated from the documented
operand layouts must give the layer kernel's TERPEV (tools/terpev_mfma_emulation.py).  Not a test of compiled code."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("mode", [0, 1, 7, 16, 31])
def test_emulated_matrix_core_terpev_equals_the_layer_kernels(mode):
    import terpev_mfma_emulation as E
    rng = np.random.default_rng(100 + mode)
    ylmc = rng.normal(size=(E.N, E.N)); cwt = rng.uniform(.01, .2, size=E.N); ylmu = rng.normal(size=(E.NUMU, E.N))
    e11 = rng.normal(size=(E.NN, E.NN)); e21 = rng.normal(size=(E.NN, E.NN)); half_gl = rng.normal(size=E.N)
    got = E.kernel(mode, ylmc, cwt, ylmu, e11, e21, half_gl)
    want = E.layer_kernel_terpev(mode, ylmc, cwt, ylmu, e11, e21, half_gl)
    assert np.isfinite(got).all()
    assert float(np.abs(got - want).max()) <= 1e-13 * float(np.abs(want).max())
