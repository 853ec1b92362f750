"""The two independent CPU formulations must agree: oracle/pqp_oracle.py (full quasi-definite KKT,
sparse LU) vs oracle/pqp_oracle.c (reduced SPD band under the per-waypoint interleave, banded Cholesky)."""
import numpy as np
import pytest
import scipy.sparse as sp

import pqp_oracle as O
import pqp_oracle_c as OC
from path_optimizer_2_amd.synth import make_batch


@pytest.mark.parametrize("n,profile", [(8, "uniform"), (80, "uniform"), (120, "varied")])
def test_c_assembly_equals_python_assembly(n, profile):
    import ctypes as C
    b = make_batch(2, n, profile)
    lib = OC.load()
    rng = np.random.default_rng(n)
    for q in range(2):
        lin = O.first_linearization(b["ref"][q]) + (rng.normal(scale=[0.2, 0.04, 0.01], size=(n, 3)) if q else 0.0)
        lin = np.ascontiguousarray(lin)
        Pd, A, lo, up, sz = O.assemble_path_qp(b["ref"][q], lin, b["bounds"][q], b["scal"][q])
        nnz = 17 * n - 5
        ri = np.zeros(nnz, dtype=np.int32); ci = np.zeros(nnz, dtype=np.int32); av = np.zeros(nnz)
        pd = np.zeros(sz["vars"]); l = np.zeros(sz["cons"]); u = np.zeros(sz["cons"])
        prm = OC.params()
        got = lib.pqo_assemble(C.byref(prm), n, OC._vp(np.ascontiguousarray(b["ref"][q])), OC._vp(lin), OC._vp(np.ascontiguousarray(b["bounds"][q])),
                               OC._vp(np.ascontiguousarray(b["scal"][q])), OC._vp(ri), OC._vp(ci), OC._vp(av), OC._vp(pd), OC._vp(l), OC._vp(u))
        assert got == nnz
        Ac = sp.coo_matrix((av, (ri, ci)), shape=A.shape).toarray()
        np.testing.assert_allclose(Ac, A, rtol=1e-14, atol=1e-16)
        np.testing.assert_array_equal(pd, Pd)
        np.testing.assert_allclose(l, lo, rtol=1e-14, atol=1e-16)
        np.testing.assert_allclose(u, up, rtol=1e-14, atol=1e-16)


@pytest.mark.parametrize("n,profile", [(80, "uniform"), (120, "varied")])
def test_c_and_python_admm_agree(n, profile):
    b = make_batch(3, n, profile)
    for eps in (2e-3, 1e-6):
        for q in range(3):
            rp = O.solve_path(b["ref"][q], b["bounds"][q], b["scal"][q], st=O.OsqpSettings(eps_abs=eps, eps_rel=eps))
            rc = OC.solve_path(OC.params(eps_abs=eps, eps_rel=eps), b["ref"][q], b["bounds"][q], b["scal"][q])
            assert rc["ok"]
            assert [int(v) for v in rc["iters"]] == [r["iters"] for r in rp]        # same algorithm, same stopping check
            assert np.abs(rc["x"] - rp[-1]["x"]).max() < 1e-8
            assert np.abs(rc["out"] - rp[-1]["out"]).max() < 1e-8


def test_c_converged_solution_passes_the_certificate():
    b = make_batch(2, 80)
    for q in range(2):
        rc = OC.solve_path(OC.params(eps_abs=1e-10, eps_rel=1e-10, max_iter=100000), b["ref"][q], b["bounds"][q], b["scal"][q], passes=0)
        Pd, A, lo, up, sz = O.assemble_path_qp(b["ref"][q], O.first_linearization(b["ref"][q]), b["bounds"][q], b["scal"][q])
        cert = O.kkt_certificate(sp.diags(Pd), np.zeros(sz["vars"]), A, lo, up, rc["x"], rc["y"])
        assert cert["pri"] < 1e-8 and cert["stat"] < 1e-8 and cert["comp"] < 1e-8, cert


def test_batch_driver_matches_single():
    b = make_batch(6, 40, "varied")
    prm = OC.params(eps_abs=1e-5, eps_rel=1e-5)
    r = OC.solve_batch(prm, b["ref"], b["bounds"], b["scal"], threads=2)
    assert r["solved"] == 6
    for q in range(6):
        s = OC.solve_path(prm, b["ref"][q], b["bounds"][q], b["scal"][q])
        np.testing.assert_array_equal(r["out"][q], s["out"])


def test_dense_assembly_mode_gives_the_same_qp_and_solution():
    """The reference-faithful assembly of the CPU baseline (dense cons x vars fill + full scan, base_solver.cpp:122,145,159,210) is the
    same QP as the direct structural fill: identical paths and iteration counts."""
    b = make_batch(6, 40, "varied")
    prm = OC.params(eps_abs=1e-5, eps_rel=1e-5)
    direct = OC.solve_batch(prm, b["ref"], b["bounds"], b["scal"])
    OC.load().pqo_set_dense_assembly(1)
    try:
        dense = OC.solve_batch(prm, b["ref"], b["bounds"], b["scal"])
    finally:
        OC.load().pqo_set_dense_assembly(0)
    np.testing.assert_array_equal(dense["out"], direct["out"])
    np.testing.assert_array_equal(dense["iters"], direct["iters"])
    assert dense["solved"] == direct["solved"] == 6
