"""Helpers of the HiGHS pins (tests/test_highs_pin.py, tests/test_gpu_highs_pin.py): a solver's output record against HiGHS on the reference's QP."""
import os
import sys

import numpy as np
import scipy.sparse as sp

import pqp_oracle as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import highs_qp as H


def qp_point(o, A, lo, up, sz):
    """the kernel's output record as a point of the reference's QP: states, controls, and the slack each collision row needs (a slack is a column of its own
    with one entry, 1, in its row - whatever the constraint mode, base_solver.cpp:192-205)"""
    n = o.shape[0]
    x = np.zeros(sz["vars"])
    x[0:3 * n:3] = o[:, 3]; x[1:3 * n:3] = o[:, 4]; x[2:3 * n:3] = o[:, 5]; x[3 * n:4 * n - 1] = o[:-1, 6]
    rows = A @ x
    Ac = sp.csc_matrix(A)
    for j in range(sz["state"] + sz["control"], sz["vars"]):
        (r,) = Ac.indices[Ac.indptr[j]:Ac.indptr[j + 1]]
        x[j] = np.clip(rows[r], lo[r], up[r]) - rows[r]
    return x


def against_highs(ref, lin, bounds, scal, o, prm=None):
    Pd, A, lo, up, sz = O.assemble_path_qp(ref, lin, bounds, scal, prm)
    A = sp.csr_matrix(A)
    xh, _, _ = H.solve_qp(Pd, np.zeros(sz["vars"]), A, lo, up)
    x = qp_point(o, A, lo, up, sz)
    Ax = A @ x
    assert np.maximum(lo - Ax, Ax - up).max() < 1e-7                                       # the kernel's point holds every row
    f = lambda z: 0.5 * np.sum(Pd * z * z)
    assert abs(f(x) - f(xh)) < 1e-7 * max(1.0, f(xh)), (f(x), f(xh))                       # and has HiGHS's optimal value
    out = O.unpack_path(xh, ref)
    d = np.abs(out - o)
    assert d[:, 4:7].max() < 2e-5, d[:, 4:7].max()                                         # heading offset, curvature, curvature rate
    assert d[:, 3].max() < 3e-4 and d[:, 0:2].max() < 3e-4, d[:, 3].max()                  # lateral offset (and x, y): the flat direction under HiGHS's 1e-7 l^2
