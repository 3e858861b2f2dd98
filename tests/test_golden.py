"""Golden vectors (tests/golden/oracle_vectors.npz): outputs of this repository's own CPU oracle, NOT of the reference
(parity unpinned -- see tests/golden/make_golden.py).  CPU tier: the oracle still reproduces them.  GPU tier: the HIP path
reproduces them without touching the oracle."""
import os

import numpy as np
import pytest

import support as S

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz"))


def test_oracle_reproduces_golden(oracle):
    f, A, B = oracle.flow_map_lin(G["flow_x"], G["flow_u"])
    assert np.allclose(f, G["flow_f"], rtol=1e-12, atol=1e-12) and np.allclose(A, G["flow_A"], rtol=1e-11, atol=1e-12) and np.allclose(B, G["flow_B"], rtol=1e-11, atol=1e-12)
    tt, ts = np.zeros(1), G["mpc_target"][None, :].copy()
    for i in range(2):
        r = oracle.mpc_solve(8, 0.0, G["mpc_x0"][i], tt, ts, int(G["mpc_nev"]), G["mpc_ev"], G["mpc_md"])
        assert np.array_equal(r["mode"], G["mpc_mode"][i])
        assert np.allclose(r["X"], G["mpc_X"][i], rtol=1e-9, atol=1e-10) and np.allclose(r["U"], G["mpc_U"][i], rtol=1e-9, atol=1e-9)
    for i in range(len(G["wbc_mode"])):
        st, out, _ = oracle.wbc_update(G["wbc_xd"][i], G["wbc_u"][i], G["wbc_rbd"][i], int(G["wbc_mode"][i]), 0.002, float(G["wbc_time"][i]), G["wbc_il"][i])
        assert st == 0 and np.abs(out - G["wbc_out"][i]).max() <= 1e-7 * max(1.0, np.abs(G["wbc_out"][i]).max())


@pytest.mark.gpu
def test_hip_reproduces_golden(interface):
    import gpu_harness as H
    B, N = 2, 8
    sol = H.make_solver(interface, 4, N)
    tt = np.zeros((B, 1)); ts = np.tile(G["mpc_target"], (B, 1, 1)).copy()
    nev = int(G["mpc_nev"])
    mb = H.MpcBatch(G["mpc_x0"], tt, ts, np.full(B, nev, dtype=np.int32), np.tile(G["mpc_ev"], (B, 1)), np.tile(G["mpc_md"], (B, 1)), N)
    sol.mpc(mb.args)
    r = mb.results()
    assert np.array_equal(r["mode"], G["mpc_mode"])                                                     # contact modes bit exact
    assert np.abs(r["X"] - G["mpc_X"]).max() <= 1e-6 * max(1.0, np.abs(G["mpc_X"]).max())
    assert np.abs(r["U"] - G["mpc_U"]).max() <= 1e-6 * max(1.0, np.abs(G["mpc_U"]).max())
    nb = len(G["wbc_mode"])
    wb = H.WbcBatch(G["wbc_rbd"], np.full(nb, 0.002), G["wbc_time"], G["wbc_il"], G["wbc_xd"], G["wbc_u"], G["wbc_mode"])
    sol.wbc(wb.args)
    w = wb.results()
    for i in range(nb):
        assert np.abs(w["out"][i][36:] - G["wbc_out"][i][36:]).max() <= 1e-6 * max(1.0, np.abs(G["wbc_out"][i][36:]).max())
