"""The generic banded-QP ADMM core (pqp_banded_qp.hpp) on the host emulator against the oracle, for the three
smoother QPs of SURVEY.md §8a rows S1-S3 (reference assembly restated in oracle/pqp_oracle.py)."""
import numpy as np
import pytest
import scipy.sparse as sp

import banded_util as B
import emu_util as E
import pqp_oracle as O
from smoother_cases import post_inputs, tension_inputs

TIGHT = O.OsqpSettings(eps_abs=1e-9, eps_rel=1e-9, max_iter=100000)


def _check(P, q, A, lo, up, pos, bw_expected, polish):
    b = B.to_banded(P, q, A, lo, up, pos)
    assert b["bw"] == bw_expected
    ref = O.osqp_admm(sp.csc_matrix(P), q, A, lo, up, TIGHT)
    assert ref["status"] == "solved"
    if polish:
        prm = E.params(eps_abs=1e-3, eps_rel=1e-3, polish=1, polish_every=25, adaptive_rho_interval=25)
    else:
        prm = E.params(eps_abs=1e-8, eps_rel=1e-8, max_iter=50000)
    r = B.emu_solve(prm, b)
    assert r["status"] == 1
    x = r["x"][pos]                      # back to the reference variable order
    assert np.abs(x - ref["x"]).max() < 1e-5
    cert = O.kkt_certificate(sp.csc_matrix(P), q, A, lo, up, x, r["y"])
    assert cert["pri"] < 1e-5 and cert["stat"] < 1e-5 and cert["comp"] < 1e-5, cert
    return r, ref


@pytest.mark.parametrize("polish", [False, True])
def test_tension2_qp(polish):
    n = 24
    x, y, ang, k, s, _ = tension_inputs(n, seed=1)
    P, q, A, lo, up = O.assemble_tension2(x, y, ang, k, s)
    _check(P, q, A, lo, up, B.interleave_tension2(n), 4, polish)


@pytest.mark.parametrize("polish", [False, True])
def test_tension_qp(polish):
    n = 20
    x, y, ang, k, s, cl = tension_inputs(n, seed=2)
    P, q, A, lo, up = O.assemble_tension(x, y, ang, cl)
    _check(P, q, A, lo, up, B.interleave3(n), 9, polish)


@pytest.mark.parametrize("polish", [False, True])
def test_post_smooth_qp(polish):
    m = 18
    s, lb, ub, l0 = post_inputs(m, seed=3)
    P, q, A, lo, up = O.assemble_post(s, list(zip(lb, ub)), l0)
    _check(P, q, A, lo, up, B.interleave3(m), 3, polish)


def test_plain_admm_matches_oracle_iteration_count():
    n = 16
    x, y, ang, k, s, _ = tension_inputs(n, seed=4)
    P, q, A, lo, up = O.assemble_tension2(x, y, ang, k, s)
    b = B.to_banded(P, q, A, lo, up, B.interleave_tension2(n))
    for eps in (1e-3, 1e-6):       # OSQP default 1e-3 is what the reference's smoothers run
        ref = O.osqp_admm(sp.csc_matrix(P), q, A, lo, up, O.OsqpSettings(eps_abs=eps, eps_rel=eps))
        r = B.emu_solve(E.params(eps_abs=eps, eps_rel=eps), b)
        assert r["status"] == 1 and r["iters"] == ref["iters"]


def test_equality_constrained_qp_is_solved_directly():
    """polish = 2: TensionSmoother2's QP has no inequality rows (every row of tension_smoother_2.cpp:119-145 has l == u), so it is ONE KKT
    system: solved at iteration 0 (no ADMM iterations), KKT-verified, the exact optimum.  A QP with inequality rows runs the plain
    ADMM under the same setting."""
    n = 24
    x, y, ang, k, s, _ = tension_inputs(n, seed=1)
    P, q, A, lo, up = O.assemble_tension2(x, y, ang, k, s)
    assert (lo == up).all()
    pos = B.interleave_tension2(n)
    b = B.to_banded(P, q, A, lo, up, pos)
    ref = O.osqp_admm(sp.csc_matrix(P), q, A, lo, up, TIGHT)
    for mode in (2, 1):
        r = B.emu_solve(E.params(eps_abs=1e-3, eps_rel=1e-3, polish=mode, polish_every=25), b)
        assert r["status"] == 1 and r["iters"] == 0 and r["info"][4] == 1            # solved, no ADMM iteration, polished
    assert np.abs(r["x"][pos] - ref["x"]).max() < 1e-6
    cert = O.kkt_certificate(sp.csc_matrix(P), q, A, lo, up, r["x"][pos], r["y"])
    assert cert["pri"] < 1e-6 and cert["stat"] < 1e-6, cert
    # postSmooth has box rows: plain ADMM, the reference setting's iteration count
    m = 18
    s2, lb, ub, l0 = post_inputs(m, seed=3)
    P, q, A, lo, up = O.assemble_post(s2, list(zip(lb, ub)), l0)
    b = B.to_banded(P, q, A, lo, up, B.interleave3(m))
    r2 = B.emu_solve(E.params(eps_abs=1e-3, eps_rel=1e-3, polish=2), b)
    r0 = B.emu_solve(E.params(eps_abs=1e-3, eps_rel=1e-3), b)
    assert r2["status"] == 1 and r2["iters"] == r0["iters"] and r2["info"][4] == 0 and np.array_equal(r2["x"], r0["x"])
