"""GPU parity of the three reference-line smoothing QPs through the C ABI:
assemble kernel + banded ADMM core + finish kernel against the oracle's restatement of the reference assembly
(oracle/pqp_oracle.py: assemble_tension2 / assemble_tension / assemble_post) solved to convergence."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.sparse.linalg import spsolve

import pqp_oracle as O
from path_optimizer_2_amd import capi
from smoother_cases import post_inputs, post_reduced_kkt as _post_reduced_kkt, tension_inputs, tension_kkt_certificate as _tension_kkt_certificate

pytestmark = pytest.mark.gpu
TIGHT = O.OsqpSettings(eps_abs=1e-9, eps_rel=1e-9, max_iter=100000)


def _polished():
    return capi.default_params(eps_abs=1e-3, eps_rel=1e-3, polish=1, polish_every=25, adaptive_rho_interval=25)


def _chord(x, y):
    return np.concatenate([[0.0], np.cumsum(np.hypot(np.diff(x), np.diff(y)))])


@pytest.mark.parametrize("n,batch", [(24, 5), (80, 3), (200, 2)])
def test_tension2(hip_lib, n, batch):
    cases = [tension_inputs(n, seed=10 + b) for b in range(batch)]
    arr = [np.stack([c[k] for c in cases]) for k in range(5)]
    h = capi.Handle(_polished(), max_batch=batch, max_n=n)
    r = h.smooth_tension2(arr[0], arr[1], arr[2], arr[3], arr[4])
    assert (r["status"] == 1).all()
    for b in range(batch):
        x, y, ang, k, s, _ = cases[b]
        P, q, A, lo, up = O.assemble_tension2(x, y, ang, k, s)
        ref = O.osqp_admm(sp.csc_matrix(P), q, A, lo, up, TIGHT)
        assert np.abs(r["x"][b] - ref["x"][:n]).max() < 1e-5 and np.abs(r["y"][b] - ref["x"][n:2 * n]).max() < 1e-5
        np.testing.assert_allclose(r["s"][b], _chord(r["x"][b], r["y"][b]), atol=1e-12)       # tension_smoother_2.cpp:61-70
    # the reference's own setting (OSQP default eps 1e-3, no polish): same ADMM, same stopping check as the oracle
    h2 = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3), max_batch=batch, max_n=n)
    r2 = h2.smooth_tension2(arr[0], arr[1], arr[2], arr[3], arr[4])
    for b in range(batch):
        x, y, ang, k, s, _ = cases[b]
        P, q, A, lo, up = O.assemble_tension2(x, y, ang, k, s)
        ref = O.osqp_admm(sp.csc_matrix(P), q, A, lo, up, O.OsqpSettings(eps_abs=1e-3, eps_rel=1e-3))
        assert abs(int(r2["iters"][b]) - ref["iters"]) <= 25
        assert np.abs(r2["x"][b] - ref["x"][:n]).max() < 1e-6
    h.close(); h2.close()


@pytest.mark.parametrize("n,batch,scaling", [(24, 5, 10), (80, 3, 0), (200, 2, 0)])
def test_tension2_is_solved_directly_when_polish_is_on(hip_lib, n, batch, scaling):
    """TensionSmoother2's QP has no inequality rows (tension_smoother_2.cpp:119-145: l == u in every row): with polish = 2 (and 1) it is
    solved exactly - no ADMM iteration, iters = 0 - by tension2_exact_kernel's Riccati sweep (whatever `scaling` says: nothing is
    equilibrated); bench.py --config 4 runs it this way."""
    cases = [tension_inputs(n, seed=50 + b) for b in range(batch)]
    arr = [np.stack([c[k] for c in cases]) for k in range(5)]
    for mode in (2, 1):
        h = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3, polish=mode, polish_every=25, scaling=scaling), max_batch=batch, max_n=n)
        r = h.smooth_tension2(arr[0], arr[1], arr[2], arr[3], arr[4])
        assert (r["status"] == 1).all() and (r["iters"] == 0).all()
        for b in range(batch):
            x, y, ang, k, s, _ = cases[b]
            P, q, A, lo, up = O.assemble_tension2(x, y, ang, k, s)
            assert (lo == up).all()
            ref = O.osqp_admm(sp.csc_matrix(P), q, A, lo, up, TIGHT)
            assert np.abs(r["x"][b] - ref["x"][:n]).max() < 1e-5 and np.abs(r["y"][b] - ref["x"][n:2 * n]).max() < 1e-5
        h.close()


@pytest.mark.parametrize("n,batch", [(20, 4), (80, 2), (150, 1)])
def test_tension(hip_lib, n, batch):
    cases = [tension_inputs(n, seed=20 + b) for b in range(batch)]
    x, y, ang, cl = (np.stack([c[k] for c in cases]) for k in (0, 1, 2, 5))
    h = capi.Handle(_polished(), max_batch=batch, max_n=n)
    r = h.smooth_tension(x, y, ang, cl)
    assert (r["status"] == 1).all()
    for b in range(batch):
        P, q, A, lo, up = O.assemble_tension(x[b], y[b], ang[b], cl[b])
        ref = O.osqp_admm(sp.csc_matrix(P), q, A, lo, up, O.OsqpSettings(eps_abs=1e-11, eps_rel=1e-11, max_iter=400000))
        # this QP is ill-conditioned (P is singular in d, third-difference weight 50, coordinates ~70 m): both the polished
        # GPU point and the ADMM oracle sit on a ~1e-5 round-off floor; the parity bar is 1e-4
        assert np.abs(r["x"][b] - ref["x"][:n]).max() < 5e-5 and np.abs(r["y"][b] - ref["x"][n:2 * n]).max() < 5e-5
    h.close()


def test_tension_sizes_beyond_the_9x9_formulation(hip_lib):
    """750 variables in 9 x 9 blocks need more LDS than a CU has: from there on every handle - the reference's ADMM setting included - gets
    the exact kernel.  A handle that asks for exact optima (polish = 1) solves the same QP as a box QP in the lateral shifts, one wavefront per
    scenario (tension_exact_kernel), for up to 1024 points; checked by the KKT conditions of the oracle's matrices, no solver involved."""
    for n, seeds in ((250, (1, 2, 3)), (384, (4,)), (500, (7, 8)), (700, (9,)), (1000, (10, 11)), (130, (5, 6))):         # (500: eight points per lane, round 4; 700 / 1000: twelve / sixteen, round 5)
        cases = [tension_inputs(n, seed=sd) for sd in seeds]
        x, y, ang, cl = (np.stack([c[k] for c in cases]) for k in (0, 1, 2, 5))
        if n >= 250:
            # the reference's own setting (OSQP defaults, eps 1e-3: tension_smoother.cpp:61-65) beyond the 9 x 9 core's LDS capacity: the exact
            # kernel's optimum (zero residuals: solved at any eps), where round 2 returned PQP_ERR_CAPACITY
            h = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3), max_batch=len(seeds), max_n=n)
            rr = h.smooth_tension(x, y, ang, cl)
            assert (rr["status"] == 1).all()
            for b in range(len(seeds)):
                assert _tension_kkt_certificate(x[b], y[b], ang[b], cl[b], rr["x"][b], rr["y"][b]) < 1e-6, (n, b)
            h.close()
        h = capi.Handle(_polished(), max_batch=len(seeds), max_n=n)
        r = h.smooth_tension(x, y, ang, cl)
        assert (r["status"] == 1).all() and (r["iters"] == 0).all()
        for b in range(len(seeds)):
            assert _tension_kkt_certificate(x[b], y[b], ang[b], cl[b], r["x"][b], r["y"][b]) < 1e-6, (n, b)
            np.testing.assert_allclose(r["s"][b], _chord(r["x"][b], r["y"][b]), atol=1e-11)
        h.close()
    # lines on which the active-set rounds cycle with the cautious threshold of 0.5 (1 in 1000, tools/active_set_sweep.py): single moves end them
    for n2, sd in ((24, 1877), (48, 1960), (80, 1325)):
        c2 = tension_inputs(n2, seed=sd)
        h = capi.Handle(_polished(), max_batch=1, max_n=n2)
        r2 = h.smooth_tension(c2[0][None], c2[1][None], c2[2][None], c2[5][None])
        assert r2["status"][0] == 1 and _tension_kkt_certificate(c2[0], c2[1], c2[2], c2[5], r2["x"][0], r2["y"][0]) < 1e-6, (n2, sd)
        h.close()
    # the certificate is not vacuous: a point moved off the optimum fails it by orders of magnitude
    bad = _tension_kkt_certificate(x[0], y[0], ang[0], cl[0], r["x"][0] + 1e-3 * np.cos(ang[0] + np.pi / 2) * np.sin(np.arange(n)), r["y"][0] + 1e-3 * np.sin(ang[0] + np.pi / 2) * np.sin(np.arange(n)))
    assert bad > 1e-3


def test_tension_lines_of_any_length(hip_lib):
    """Beyond 1024 points (the reference has no cap: tension_smoother.cpp:49-100; a point per metre of raw reference) the exact kernel keeps its
    arrays in HBM (tension_exact_kernel<0>, SmHbm) instead of registers: the same iteration, checked by the KKT conditions of the oracle's
    matrices - both handle settings, a point count per line across the 64-point chunks, and the previous cycle's active set carried."""
    for n, seeds in ((1025, (21,)), (1600, (22, 23)), (2500, (24,))):
        cases = [tension_inputs(n, seed=sd) for sd in seeds]
        x, y, ang, cl = (np.stack([c[k] for c in cases]) for k in (0, 1, 2, 5))
        for prm in (_polished(), capi.default_params(eps_abs=1e-3, eps_rel=1e-3)):
            h = capi.Handle(prm, max_batch=len(seeds), max_n=n)
            r = h.smooth_tension(x, y, ang, cl, info=True)
            assert (r["status"] == 1).all() and (r["iters"] == 0).all()
            assert (r["info"][:, 5] <= 40).all(), r["info"][:, 5]               # factorisations: 10-12 interior iterations + 1-3 rounds at every size
            for b in range(len(seeds)):
                assert _tension_kkt_certificate(x[b], y[b], ang[b], cl[b], r["x"][b], r["y"][b]) < 1e-6, (n, b)
                np.testing.assert_allclose(r["s"][b], _chord(r["x"][b], r["y"][b]), atol=1e-10)
            h.close()
    # a point count per line: the chunk boundaries (1088 = 17 * 64), short lines in a long batch, the tail repeats the last point
    n = 1700
    counts = np.array([1700, 1087, 1088, 1089, 5, 64, 1025, 1699], dtype=np.int32)
    B = len(counts)
    x = np.zeros((B, n)); y = np.zeros((B, n)); ang = np.zeros((B, n)); cl = np.ones((B, n))
    for b, c in enumerate(counts):
        cx, cy, ca, _, _, cc = tension_inputs(int(c), seed=300 + b)
        x[b, :c], y[b, :c], ang[b, :c], cl[b, :c] = cx, cy, ca, cc
    h = capi.Handle(_polished(), max_batch=B, max_n=n)
    h.set_option(capi.OPT_CARRY_CYCLES, 1)
    first = h.smooth_tension_var(x, y, ang, cl, counts)
    again = h.smooth_tension_var(x, y, ang, cl, counts)               # from the carried set: the same optimum in one or two rounds
    h.close()
    for r in (first, again):
        assert (r["status"] == 1).all()
        for b, c in enumerate(counts):
            assert _tension_kkt_certificate(x[b, :c], y[b, :c], ang[b, :c], cl[b, :c], r["x"][b, :c], r["y"][b, :c]) < 1e-6, b
            assert np.all(r["x"][b, c:] == r["x"][b, c - 1]) and np.all(r["s"][b, c:] == r["s"][b, c - 1])
    assert np.abs(first["x"] - again["x"]).max() < 1e-7


def test_exact_tension_kernel_is_bounded_from_the_cold_start(hip_lib):
    """The first active set comes from interior-point iterations on the box QP (round 3): a launch lasts as long as its slowest line, and no line
    of these batches needs more than 18 factorisations (round 2's rounds from OSQP's cold-start rule: up to 39 / 67 / 52 at 48 / 80 / 200 points).
    The counts are those of the numpy restatement (tools/active_set_sweep.py); the optimum is checked by the KKT certificate of the oracle's matrices."""
    for n in (48, 80, 200):
        cases = [tension_inputs(n, seed=1000 + b) for b in range(32)]
        x, y, ang, cl = (np.stack([c[k] for c in cases]) for k in (0, 1, 2, 5))
        h = capi.Handle(_polished(), max_batch=len(cases), max_n=n)
        r = h.smooth_tension(x, y, ang, cl, info=True)
        h.close()
        assert (r["status"] == 1).all()
        fac, ipm = r["info"][:, 5], r["info"][:, 3]
        assert fac.max() <= 18 and ipm.max() <= 15 and (fac - ipm).max() <= 4, (n, fac.max(), ipm.max())
        for b in (0, 7, 31):
            assert _tension_kkt_certificate(x[b], y[b], ang[b], cl[b], r["x"][b], r["y"][b]) < 1e-6, (n, b)


def test_exact_tension_kernel_on_hostile_clearances_and_rough_lines(hip_lib):
    """The interior start on inputs it was not tuned on: every box a tenth of a micrometre wide, widths spread over six decades, a third of them
    narrower than a millimetre, clearances far beyond the 2 m cap, raw lines with half a metre of noise.  Every line ends at the KKT point of the
    oracle's matrices; none needs more than 45 factorisations (the numpy restatement: at most 36 over these patterns)."""
    n = 80
    for pat in ("tiny", "log", "mixed", "huge", "noisy"):
        cases = []
        for b in range(16):
            x, y, ang, _, _, cl = tension_inputs(n, seed=2000 + b)
            rng = np.random.default_rng(2007 + b)
            if pat == "tiny": cl = np.full(n, 1e-7)
            elif pat == "log": cl = 10 ** rng.uniform(-6, 0.3, n)
            elif pat == "mixed": cl = np.where(rng.uniform(size=n) < 0.3, 10 ** rng.uniform(-9, -3, n), rng.uniform(0.3, 3, n))
            elif pat == "huge": cl = np.full(n, 50.0)
            else: x = x + rng.normal(scale=0.5, size=n); y = y + rng.normal(scale=0.5, size=n)
            cases.append((x, y, ang, cl))
        x, y, ang, cl = (np.stack([c[k] for c in cases]) for k in range(4))
        h = capi.Handle(_polished(), max_batch=len(cases), max_n=n)
        r = h.smooth_tension(x, y, ang, cl, info=True)
        h.close()
        assert (r["status"] == 1).all(), pat
        assert r["info"][:, 5].max() <= 45, (pat, r["info"][:, 5].max())
        for b in range(len(cases)):
            assert _tension_kkt_certificate(x[b], y[b], ang[b], cl[b], r["x"][b], r["y"][b]) < 1e-6, (pat, b)


def test_exact_smoother_kernels_carry_the_previous_cycle_s_active_set(hip_lib):
    """PQP_OPT_CARRY_CYCLES on the exact TensionSmoother / postSmooth kernels: a line that moved a little since the previous solve starts its
    active-set rounds from the set its slot ended with.  The same optimum (the KKT certificate of the oracle's matrices; equal to the cold solve's),
    a fraction of the factorisations; another shape on the handle starts cold."""
    n = 80
    cases = [tension_inputs(n, seed=40 + b) for b in range(8)]
    x, y, ang, cl = (np.stack([c[k] for c in cases]) for k in (0, 1, 2, 5))
    hc = capi.Handle(_polished(), max_batch=8, max_n=n); hk = capi.Handle(_polished(), max_batch=8, max_n=n)
    hc.set_option(capi.OPT_CARRY_CYCLES, 1)
    rng = np.random.default_rng(3)
    for v in range(3):
        sh = rng.uniform(-0.03, 0.03, x.shape) if v else 0.0
        xv, yv, clv = x + sh * np.cos(ang + np.pi / 2), y + sh * np.sin(ang + np.pi / 2), cl * (1 + (rng.uniform(-0.05, 0.05, (8, 1)) if v else 0.0))
        rc, rk = hc.smooth_tension(xv, yv, ang, clv, info=True), hk.smooth_tension(xv, yv, ang, clv, info=True)
        assert (rc["status"] == 1).all() and (rk["status"] == 1).all()
        for b in range(8):
            assert _tension_kkt_certificate(xv[b], yv[b], ang[b], clv[b], rc["x"][b], rc["y"][b]) < 1e-6, (v, b)
        assert np.abs(rc["x"] - rk["x"]).max() < 1e-7 and np.abs(rc["y"] - rk["y"]).max() < 1e-7
        if v:
            assert rc["info"][:, 5].mean() < 0.5 * rk["info"][:, 5].mean(), (v, rc["info"][:, 5].mean(), rk["info"][:, 5].mean())
    c2 = [tension_inputs(48, seed=60 + b) for b in range(8)]
    a2 = [np.stack([c[k] for c in c2]) for k in (0, 1, 2, 5)]
    np.testing.assert_array_equal(hc.smooth_tension(*a2)["x"], hk.smooth_tension(*a2)["x"])
    hc.close(); hk.close()


@pytest.mark.parametrize("m,batch", [(18, 4), (60, 2), (150, 1)])
def test_post_smooth(hip_lib, m, batch):
    cases = [post_inputs(m, seed=30 + b) for b in range(batch)]
    s = np.stack([c[0] for c in cases]); lb = np.stack([c[1] for c in cases]); ub = np.stack([c[2] for c in cases])
    l0 = np.array([c[3] for c in cases])
    h = capi.Handle(_polished(), max_batch=batch, max_n=m)
    r = h.post_smooth(s, lb, ub, l0)
    assert (r["status"] == 1).all()
    for b in range(batch):
        P, q, A, lo, up = O.assemble_post(s[b], list(zip(lb[b], ub[b])), l0[b])
        ref = O.osqp_admm(sp.csc_matrix(P), q, A, lo, up, TIGHT)
        assert np.abs(r["l"][b] - ref["x"][:m]).max() < 1e-5
        assert abs(r["l"][b][0] - l0[b]) < 1e-7                   # the KKT acceptance tolerance (polish_tol)
        assert (r["l"][b][1:] >= lb[b][1:] - 1e-7).all() and (r["l"][b][1:] <= ub[b][1:] + 1e-7).all()
    h.close()


def test_tension_with_a_point_count_per_scenario(hip_lib):
    """pqp_smooth_tension_var_device: scenarios of 20..60 points in one launch of the 60-point pattern (the shorter ones padded with
    decoupled dummies) against the oracle's assembly of each scenario at its own size - the difference windows of the cost and the
    end point's +-0.5 m box (tension_smoother.cpp:108-124,161-162) must end at the scenario's last point."""
    counts = np.array([60, 20, 37, 4, 59], dtype=np.int32)
    n, B = 60, len(counts)
    cases = [tension_inputs(int(c), seed=70 + b) for b, c in enumerate(counts)]
    pad = lambda k: np.stack([np.concatenate([c[k], np.full(n - len(c[k]), np.nan)]) for c in cases])      # the padding is never read
    h = capi.Handle(_polished(), max_batch=B, max_n=n)
    r = h.smooth_tension_var(pad(0), pad(1), pad(2), pad(5), counts)
    assert (r["status"] == 1).all()
    for b, c in enumerate(counts):
        x, y, ang, _, _, cl = cases[b]
        P, q, A, lo, up = O.assemble_tension(x, y, ang, cl)
        ref = O.osqp_admm(sp.csc_matrix(P), q, A, lo, up, O.OsqpSettings(eps_abs=1e-11, eps_rel=1e-11, max_iter=400000))
        assert np.abs(r["x"][b, :c] - ref["x"][:c]).max() < 5e-5 and np.abs(r["y"][b, :c] - ref["x"][c:2 * c]).max() < 5e-5
        np.testing.assert_allclose(r["s"][b, :c], _chord(r["x"][b, :c], r["y"][b, :c]), atol=1e-12)
        assert np.all(r["x"][b, c:] == r["x"][b, c - 1]) and np.all(r["s"][b, c:] == r["s"][b, c - 1])          # the tail repeats the last point
    # and the same numbers as the launch of one scenario at its own size
    one = h.smooth_tension(*(cases[2][k][None] for k in (0, 1, 2, 5)))
    assert np.abs(one["x"][0] - r["x"][2, :37]).max() < 1e-6 and np.abs(one["y"][0] - r["y"][2, :37]).max() < 1e-6
    h.close()


def test_post_smooth_exact_kernel(hip_lib):
    """Handles that ask for exact optima (polish = 1) solve postSmooth's QP as a box QP in the offsets, one wavefront per scenario, for
    corridors (post_exact_kernel; here up to 64 layers, one per lane): ragged batch, tight and pinned boxes, 64 layers; checked by the KKT conditions
    of the oracle's matrices (no solver) and, for a few, against the oracle's ADMM run to 1e-9."""
    rng = np.random.default_rng(3)
    B, m = 96, 64
    counts = rng.integers(4, m + 1, size=B).astype(np.int32); counts[:4] = [64, 4, 5, 63]
    s = np.zeros((B, m)); lb = np.zeros((B, m)); ub = np.zeros((B, m)); l0 = np.zeros(B)
    for b in range(B):
        c = int(counts[b])
        sb, lbb, ubb, v = post_inputs(c, seed=200 + b)
        if b % 5 == 1:
            half = rng.uniform(0.02, 0.15, size=c); mid = 0.5 * (lbb + ubb); lbb, ubb = mid - half, mid + half        # tight corridor: most boxes active
        if b % 7 == 2:
            k = int(rng.integers(1, c)); ubb[k] = lbb[k]                                                               # an interior layer pinned
        if b % 11 == 3:
            sb = np.concatenate([[0.0], np.cumsum(rng.uniform(0.8, 2.2, size=c - 1))])                               # uneven layer spacing
        s[b, :c], lb[b, :c], ub[b, :c], l0[b] = sb, lbb, ubb, v
        s[b, c:] = np.nan; lb[b, c:] = np.nan; ub[b, c:] = np.nan                                                       # never read
    h = capi.Handle(_polished(), max_batch=B, max_n=m)
    r = h.post_smooth_var(s, lb, ub, l0, counts, info=True)
    assert (r["status"] == 1).all() and (r["iters"] == 0).all()
    worst = 0.0
    for b in range(B):
        c = int(counts[b])
        worst = max(worst, _post_reduced_kkt(s[b, :c], lb[b, :c], ub[b, :c], l0[b], r["l"][b, :c]))
        assert np.all(r["l"][b, c:] == 0.0)
    assert worst < 5e-7, worst
    assert r["info"][:, 5].max() <= 40 and r["info"][:, 5].mean() < 8           # active-set rounds: a handful
    for b in (0, 1, 6, 9, 14):
        c = int(counts[b])
        P, q, A, lo, up = O.assemble_post(s[b, :c], list(zip(lb[b, :c], ub[b, :c])), l0[b])
        ref = O.osqp_admm(sp.csc_matrix(P), q, A, lo, up, TIGHT)
        assert np.abs(r["l"][b, :c] - ref["x"][:c]).max() < 1e-5, b
    # the certificate is not vacuous
    assert _post_reduced_kkt(s[0, :64], lb[0, :64], ub[0, :64], l0[0], r["l"][0, :64] + 1e-4 * np.sin(np.arange(64))) > 1e-5
    # an inverted box has no feasible point: OSQP's verdict
    lb2 = lb.copy(); lb2[5, 2] = ub[5, 2] + 0.1
    r2 = h.post_smooth_var(s, lb2, ub, l0, counts)
    assert r2["status"][5] == 4 and (np.delete(r2["status"], 5) == 1).all()
    # two layers at the same abscissa make the difference rows singular: NUMERICAL, zeros, the neighbours in the batch untouched
    s3 = s.copy(); s3[7, 3] = s3[7, 2]
    r3 = h.post_smooth_var(s3, lb, ub, l0, counts)
    assert r3["status"][7] == 3 and np.all(r3["l"][7] == 0.0) and (np.delete(r3["status"], 7) == 1).all()
    h.close()


@pytest.mark.parametrize("m", [100, 200, 341, 500, 700, 1000])
def test_post_smooth_exact_kernel_on_long_corridors(hip_lib, m):
    """More than 64 layers: two, four, six, eight, twelve or sixteen layers per lane of the same wavefront (post_exact_kernel<K>); ragged counts across the slot
    boundaries (63, 64, 65, 128, 129 ...), the KKT certificate of the oracle's matrices, and the generic core - the reference's
    formulation, ADMM run to 1e-9 on the device - on one of them."""
    rng = np.random.default_rng(m)
    counts = np.array([m, 63, 64, 65, 4, min(m, 128), min(m, 129), m - 1, min(m, 193), (m + 64) // 2], dtype=np.int32)
    B = len(counts)
    s = np.full((B, m), np.nan); lb = np.full((B, m), np.nan); ub = np.full((B, m), np.nan); l0 = np.zeros(B)
    for b in range(B):
        c = int(counts[b])
        sb, lbb, ubb, v = post_inputs(c, seed=500 + 7 * b + m)
        if b % 3 == 1:
            half = rng.uniform(0.02, 0.15, size=c); mid = 0.5 * (lbb + ubb); lbb, ubb = mid - half, mid + half
        s[b, :c], lb[b, :c], ub[b, :c], l0[b] = sb, lbb, ubb, v
    h = capi.Handle(_polished(), max_batch=B, max_n=m)
    r = h.post_smooth_var(s, lb, ub, l0, counts, info=True)
    assert (r["status"] == 1).all() and (r["iters"] == 0).all()
    for b in range(B):
        c = int(counts[b])
        assert _post_reduced_kkt(s[b, :c], lb[b, :c], ub[b, :c], l0[b], r["l"][b, :c]) < 5e-7, b
        assert np.all(r["l"][b, c:] == 0.0)
    h.close()
    b, c = 3, 65
    g = capi.Handle(capi.default_params(eps_abs=1e-9, eps_rel=1e-9, max_iter=200000, adaptive_rho_interval=25), max_batch=1, max_n=c)
    want = g.post_smooth(s[b:b + 1, :c], lb[b:b + 1, :c], ub[b:b + 1, :c], l0[b:b + 1])
    assert want["status"][0] == 1 and want["iters"][0] > 0
    assert np.abs(want["l"][0] - r["l"][b, :c]).max() < 1e-6
    g.close()


def test_post_smooth_corridors_of_any_length(hip_lib):
    """Beyond 1024 layers (reference_path_smoother.cpp:526-580 has no cap) post_exact_kernel<0> keeps its arrays in HBM; beyond what the generic
    core holds (3 m variables on at most 1024 lanes) also a handle in the reference's ADMM setting gets the exact optimum."""
    for m in (1025, 2200):
        counts = np.array([m, 1024, m - 1, 4, 1088 if m > 1088 else 65, (m + 1024) // 2], dtype=np.int32)
        B = len(counts)
        rng = np.random.default_rng(m)
        s = np.full((B, m), np.nan); lb = np.full((B, m), np.nan); ub = np.full((B, m), np.nan); l0 = np.zeros(B)
        for b in range(B):
            c = int(counts[b])
            sb, lbb, ubb, v = post_inputs(c, seed=900 + 7 * b + m)
            if b % 3 == 1:
                half = rng.uniform(0.02, 0.15, size=c); mid = 0.5 * (lbb + ubb); lbb, ubb = mid - half, mid + half
            s[b, :c], lb[b, :c], ub[b, :c], l0[b] = sb, lbb, ubb, v
        for prm in (_polished(), capi.default_params(eps_abs=1e-3, eps_rel=1e-3)):
            h = capi.Handle(prm, max_batch=B, max_n=m)
            r = h.post_smooth_var(s, lb, ub, l0, counts, info=True)
            h.close()
            assert (r["status"] == 1).all() and (r["iters"] == 0).all()
            for b in range(B):
                c = int(counts[b])
                assert _post_reduced_kkt(s[b, :c], lb[b, :c], ub[b, :c], l0[b], r["l"][b, :c]) < 5e-7, (m, b)
                assert np.all(r["l"][b, c:] == 0.0)
    # hostile corridors end as they do in registers: abscissae that do not increase -> NUMERICAL, an inverted box -> PRIMAL_INFEASIBLE, zero offsets
    m = 1300
    sb, lbb, ubb, v = post_inputs(m, seed=5)
    s2 = np.stack([sb, sb, sb]); lb2 = np.stack([lbb, lbb, lbb]); ub2 = np.stack([ubb, ubb, ubb])
    s2[1, 1200] = s2[1, 1199]
    lb2[2, 1100], ub2[2, 1100] = 0.5, -0.5
    h = capi.Handle(_polished(), max_batch=3, max_n=m)
    r = h.post_smooth(s2, lb2, ub2, np.full(3, v))
    h.close()
    assert list(r["status"]) == [1, 3, 4] and np.all(r["l"][1:] == 0.0)


def test_tension2_beyond_the_generic_core_in_the_reference_setting(hip_lib):
    """TensionSmoother2's 4 n variables fill the generic core's 1024 lanes at 256 points; tension_smoother_2.cpp:20-72 has no cap.  Beyond it a
    handle in the reference's ADMM setting gets the Riccati sweep's optimum (zero residuals: solved at any eps) instead of PQP_ERR_CAPACITY."""
    n = 700
    cases = [tension_inputs(n, seed=40 + b) for b in range(2)]
    arr = [np.stack([c[k] for c in cases]) for k in range(5)]
    h = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3), max_batch=2, max_n=n)
    r = h.smooth_tension2(arr[0], arr[1], arr[2], arr[3], arr[4])
    h.close()
    g = capi.Handle(_polished(), max_batch=2, max_n=n)
    want = g.smooth_tension2(arr[0], arr[1], arr[2], arr[3], arr[4])
    g.close()
    assert (r["status"] == 1).all() and (want["status"] == 1).all()
    assert np.array_equal(r["x"], want["x"]) and np.array_equal(r["y"], want["y"]) and np.array_equal(r["s"], want["s"])
    for b in range(2):
        x, y, ang, k, s, _ = cases[b]
        P, q, A, lo, up = O.assemble_tension2(x, y, ang, k, s)
        # equality rows only: the optimum is the solution of the KKT system of the oracle's matrices
        nv, nc = P.shape[0], A.shape[0]
        kkt = sp.bmat([[sp.csc_matrix(P), sp.csc_matrix(A).T], [sp.csc_matrix(A), None]], format="csc")
        sol = spsolve(kkt, np.concatenate([-q, lo]))
        assert np.abs(lo - up).max() == 0.0
        np.testing.assert_allclose(r["x"][b], sol[:n], atol=1e-7)
        np.testing.assert_allclose(r["y"][b], sol[n:2 * n], atol=1e-7)


def test_tension2_exact_kernel_with_a_point_count_per_scenario(hip_lib):
    """Handles with polish != 0 solve TensionSmoother2's equality-constrained QP by a Riccati sweep, one lane per scenario
    (tension2_exact_kernel): a ragged batch (3..256 points, uneven spacing) against the oracle's assembly of each scenario at its own size
    solved to 1e-9, the chord lengths, the repeated tail - and against the generic core's direct KKT solve of the 4n-variable formulation
    (polish = 0 handles keep it; here: the reference's ADMM run to 1e-9 on the device)."""
    rng = np.random.default_rng(8)
    counts = np.array([256, 3, 4, 5, 17, 64, 65, 130, 200, 255], dtype=np.int32)
    n, B = 256, len(counts)
    arr = [np.full((B, n), np.nan) for _ in range(5)]
    cases = []
    for b, c in enumerate(counts):
        x, y, ang, k, s, _ = tension_inputs(int(c), seed=300 + b, ds=1.0 if b % 2 else 0.7 + 0.1 * b)
        cases.append((x, y, ang, k, s))
        for j, a in enumerate((x, y, ang, k, s)):
            arr[j][b, :c] = a
    h = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3, polish=2), max_batch=B, max_n=n)
    r = h.smooth_tension2_var(*arr, counts)
    assert (r["status"] == 1).all() and (r["iters"] == 0).all()
    for b, c in enumerate(counts):
        x, y, ang, k, s = cases[b]
        P, q, A, lo, up = O.assemble_tension2(x, y, ang, k, s)
        K = np.block([[P, A.T], [A, np.zeros((A.shape[0], A.shape[0]))]])               # every row is an equality: the optimum is one linear system
        sol = np.linalg.solve(K, np.r_[-q, lo])
        assert np.abs(r["x"][b, :c] - sol[:c]).max() < 1e-7 and np.abs(r["y"][b, :c] - sol[c:2 * c]).max() < 1e-7, (b, c)
        np.testing.assert_allclose(r["s"][b, :c], _chord(r["x"][b, :c], r["y"][b, :c]), atol=1e-11)
        assert np.all(r["x"][b, c:] == r["x"][b, c - 1]) and np.all(r["y"][b, c:] == r["y"][b, c - 1]) and np.all(r["s"][b, c:] == r["s"][b, c - 1])
    h.close()
    # the reference's formulation on the generic core, run to convergence, lands on the same line
    x, y, ang, k, s = cases[4]
    g = capi.Handle(capi.default_params(eps_abs=1e-9, eps_rel=1e-9, max_iter=20000), max_batch=1, max_n=17)
    want = g.smooth_tension2(x[None], y[None], ang[None], k[None], s[None])
    assert want["status"][0] == 1 and want["iters"][0] > 0
    assert np.abs(want["x"][0] - r["x"][4, :17]).max() < 1e-6 and np.abs(want["y"][0] - r["y"][4, :17]).max() < 1e-6
    g.close()


def test_golden_smoother_fixtures_through_the_hip_path(hip_lib):
    """Committed golden vectors of the three smoothing QPs (tests/golden/smoothers.npz, make_golden.py): the exact kernels land on the
    stored optima, the reference's ADMM setting (eps 1e-3) within its own tolerance of them."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "smoothers.npz"))
    for tag in ("a", "b"):
        x, y, ang, k, s, cl = (g[f"{tag}_{key}"][None] for key in ("x", "y", "angle", "k", "s", "clearance"))
        n = x.shape[1]
        h = capi.Handle(_polished(), max_batch=1, max_n=n)
        r = h.smooth_tension2(x, y, ang, k, s)
        assert r["status"][0] == 1 and np.abs(r["x"][0] - g[f"{tag}_t2_x"]).max() < 1e-7 and np.abs(r["y"][0] - g[f"{tag}_t2_y"]).max() < 1e-7
        r = h.smooth_tension(x, y, ang, cl)
        assert r["status"][0] == 1 and np.abs(r["x"][0] - g[f"{tag}_t_x"]).max() < 5e-5 and np.abs(r["y"][0] - g[f"{tag}_t_y"]).max() < 5e-5
        h.close()
        h = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3), max_batch=1, max_n=n)          # the reference's setting
        r = h.smooth_tension2(x, y, ang, k, s)
        assert r["status"][0] == 1 and r["iters"][0] > 0 and np.abs(r["x"][0] - g[f"{tag}_t2_x"]).max() < 2e-2
        r = h.smooth_tension(x, y, ang, cl)
        # (TensionSmoother's QP is ill-conditioned: OSQP's eps 1e-3 residual test passes decimetres away from the optimum at the far end of the line)
        assert r["status"][0] == 1 and r["iters"][0] > 0 and np.abs(r["x"][0] - g[f"{tag}_t_x"]).max() < 0.5
        h.close()
    for tag in ("c", "d"):
        s, lb, ub, l0 = g[f"{tag}_s"][None], g[f"{tag}_lb"][None], g[f"{tag}_ub"][None], np.array([float(g[f"{tag}_l0"])])
        h = capi.Handle(_polished(), max_batch=1, max_n=s.shape[1])
        r = h.post_smooth(s, lb, ub, l0)
        assert r["status"][0] == 1 and np.abs(r["l"][0] - g[f"{tag}_l"]).max() < 1e-6
        h.close()
        h = capi.Handle(capi.default_params(eps_abs=1e-3, eps_rel=1e-3), max_batch=1, max_n=s.shape[1])
        r = h.post_smooth(s, lb, ub, l0)
        assert r["status"][0] == 1 and r["iters"][0] > 0 and np.abs(r["l"][0] - g[f"{tag}_l"]).max() < 2e-2
        h.close()
