"""The lane-per-QP path-QP solver (csrc/pqp_path_lq.hpp: the path QP as a linear-quadratic control problem, interior-point rounds +
active-set rounds) compiled for the HOST and checked against the oracle; tests/test_gpu_stream.py then checks the real kernel through the
C ABI.  The oracle here is the checker, never the thing tested."""
import numpy as np
import pytest

import pqp_oracle as O
import pqp_oracle_c as OC
from path_optimizer_2_amd.synth import make_batch
import lq_emu_util as E

TIGHT_C = dict(eps_abs=1e-9, eps_rel=1e-9, max_iter=40000)
TIGHT = O.OsqpSettings(eps_abs=1e-9, eps_rel=1e-9, max_iter=40000)


@pytest.mark.parametrize("n,profile,batch,seed", [(80, "uniform", 192, None), (120, "varied", 96, 2), (200, "uniform", 48, 4), (9, "varied", 64, 1), (300, "varied", 24, 8)])
def test_every_path_is_the_converged_oracle_s(n, profile, batch, seed):
    b = make_batch(batch, n, profile) if seed is None else make_batch(batch, n, profile, seed=seed)
    r = E.solve(b["ref"], b["bounds"], b["scal"], passes=1)
    assert (r["status"] == 1).all()
    assert (r["info"][:, 4] == 2).all()                      # both passes ended in an active-set round that changed nothing: the KKT test
    k = min(batch, 32)
    want = OC.solve_batch(OC.params(**TIGHT_C), b["ref"][:k], b["bounds"][:k], b["scal"][:k], passes=1)["out"]
    err = np.abs(r["out"][:k, :, 3:5] - want[:, :, 3:5]).max(axis=(1, 2))
    assert err.max() < 2e-5 and np.median(err) < 1e-6, (err.max(), np.median(err))
    assert np.abs(r["out"][:k] - want).max() < 1e-4


def kkt_certificate_of_the_last_pass(ref, bounds, scal, first, final):
    """Solver-independent: the returned (l, psi, kappa, kappa') with the slacks and multipliers it implies satisfies the KKT conditions of
    the assembled QP of the LAST pass (oracle assembly around the first pass's optimum, reference numbering)."""
    import scipy.sparse as sp
    n = ref.shape[0]
    Pd, A, lo, up, sz = O.assemble_path_qp(ref, first[:, 3:6], bounds, scal)
    o = final
    x = np.zeros(sz["vars"])
    x[0:3 * n:3] = o[:, 3]; x[1:3 * n:3] = o[:, 4]; x[2:3 * n:3] = o[:, 5]; x[3 * n:4 * n - 1] = o[:-1, 6]
    # slacks: what the collision rows need beyond their box
    A = sp.csr_matrix(A)
    rows = A @ x
    for i in range(n):
        for j in range(2):
            r_ = 4 * n + 2 * i + j
            x[4 * n - 1 + 2 * i + j] = np.clip(rows[r_], lo[r_], up[r_]) - rows[r_]
    ax = A @ x
    assert np.maximum(lo - ax, ax - up).max() < 1e-8                      # primal feasibility
    # multipliers: slack rows y = -w_s s; the rest from stationarity by least squares on the rows active at x
    g = Pd * x
    act = (np.abs(ax - lo) < 1e-7) | (np.abs(ax - up) < 1e-7)
    At = A[act].T.toarray()
    y_act, *_ = np.linalg.lstsq(At, -g, rcond=None)
    assert np.abs(At @ y_act + g).max() < 1e-6                            # stationarity
    y = np.zeros(sz["cons"]); y[act] = y_act
    ineq = act & (up - lo > 1e-9)
    assert (y[ineq & (np.abs(ax - up) < 1e-7)] > -1e-6).all() and (y[ineq & (np.abs(ax - lo) < 1e-7)] < 1e-6).all()   # dual signs


def test_kkt_certificate_of_the_returned_point():
    b = make_batch(6, 60, "varied", seed=12)
    r0 = E.solve(b["ref"], b["bounds"], b["scal"], passes=0)
    r1 = E.solve(b["ref"], b["bounds"], b["scal"], passes=1)
    for q in range(6):
        kkt_certificate_of_the_last_pass(b["ref"][q], b["bounds"][q], b["scal"][q], r0["out"][q], r1["out"][q])


def test_sorted_launch_starts_the_second_pass_from_the_first_pass_s_set():
    """In a launch whose wavefronts are sorted by their phase counts (Args::order) the re-linearised pass first tries the previous pass's active set on the
    new transition rows (lq::kDirectRounds rounds); a confirmed set IS the KKT test, an unconfirmed one falls back to the interior-point rounds started
    from the kept-aside optimum - with the same iteration counts as an unsorted launch.  Same statuses, the same optimum within the rounds' tolerances."""
    fallbacks = 0
    for n, profile, batch, seed in ((80, "uniform", 1024, 21), (200, "varied", 256, 22), (9, "varied", 256, 23), (300, "varied", 96, 24)):
        b = make_batch(batch, n, profile, seed=seed)
        r0 = E.solve(b["ref"], b["bounds"], b["scal"], passes=1)
        r1 = E.solve(b["ref"], b["bounds"], b["scal"], passes=1, sorted_launch=True)
        assert (r0["status"] == r1["status"]).all()
        ok = r0["status"] == 1
        assert np.abs(r0["out"][ok] - r1["out"][ok])[:, :, 3:6].max() < 5e-7
        # the first pass is the same computation
        assert (r0["info"][:, 2] == r1["info"][:, 2]).all() and (r0["info"][:, 5] == r1["info"][:, 5]).all()
        it2_0, it2_1 = r0["info"][ok, 3] - r0["info"][ok, 2], r1["info"][ok, 3] - r1["info"][ok, 2]
        hit = it2_1 == 0
        assert hit.mean() > (0.9 if n <= 120 else 0.5), hit.mean()            # most sets survive the re-linearisation
        # a fallback is the unsorted launch's second pass (same start, same iterations) behind the rounds that did not confirm
        assert (it2_1[~hit] == it2_0[~hit]).all()
        fallbacks += int((~hit).sum())
        k = min(batch, 16)
        want = OC.solve_batch(OC.params(**TIGHT_C), b["ref"][:k], b["bounds"][:k], b["scal"][:k], passes=1)["out"]
        sel = ok[:k]
        assert np.abs(r1["out"][:k][sel] - want[sel])[:, :, 3:5].max() < 2e-5
    assert fallbacks >= 20          # (the fallback was exercised)


def test_the_chunk_layout_is_the_same_solver():
    """lq::ChunkWs ([chunk][lane][16 bytes], whole-chunk stores: the workspace of the device kernel's staged form) against lq::StridedWs ([field][lane]): other
    addresses, the same arithmetic on the same values - the same counts for every QP, ragged paths and sorted launches included, and the same paths up to what
    the compiler's choice of fused multiply-adds in the two instantiations moves (a few 1e-9)."""
    for n, profile, batch, seed in ((80, "uniform", 256, 31), (37, "varied", 128, 32), (200, "varied", 48, 33)):
        b = make_batch(batch, n, profile, seed=seed)
        n_of = np.full(batch, n, dtype=np.int32); n_of[::3] = max(n - 11, 2); n_of[1] = 2
        for srt in (False, True):
            r0 = E.solve(b["ref"], b["bounds"], b["scal"], passes=1, n_of=n_of, sorted_launch=srt)
            r1 = E.solve(b["ref"], b["bounds"], b["scal"], passes=1, n_of=n_of, sorted_launch=srt, chunk_layout=True)
            assert np.array_equal(r0["status"], r1["status"]) and np.array_equal(r0["info"][:, 2:], r1["info"][:, 2:])
            assert np.abs(r0["out"] - r1["out"]).max() < 5e-8


def test_kkt_certificate_of_a_sorted_launch_s_point():
    b = make_batch(6, 60, "varied", seed=12)
    r0 = E.solve(b["ref"], b["bounds"], b["scal"], passes=0)
    r1 = E.solve(b["ref"], b["bounds"], b["scal"], passes=1, sorted_launch=True)
    for q in range(6):
        kkt_certificate_of_the_last_pass(b["ref"][q], b["bounds"][q], b["scal"][q], r0["out"][q], r1["out"][q])


# QPs on which the plain active-set rounds cycle - from the interior point of every attempt - until the rounds run out (found by
# tools/lq_robustness_sweep.py-style sweeps of 16 384 ... 131 072 QPs per size; none at 300 waypoints or fewer): (n, profile, seed, QP)
CYCLING = [(512, "varied", None, 5742), (1000, "varied", 3001, 1450), (1000, "uniform", 3007, 3912)]


@pytest.mark.parametrize("n,profile,seed,qp", CYCLING)
def test_guarded_rounds_end_the_cycles_of_long_paths(n, profile, seed, qp):
    """forward_set<GUARDED>: the last attempt's rounds beyond the usual twelve step only as far as the piecewise quadratic objective
    falls.  These QPs ended PQP_STATUS_MAX_ITER before (profiles/r05t_crossover_long_paths.txt: 16 383 of 16 384 at 512 waypoints); now
    SOLVED, and what SOLVED returns passes the KKT conditions of the assembled QP."""
    kw = {} if seed is None else {"seed": seed}
    b = make_batch(1, n, profile, first_qp=qp, **kw)
    r0 = E.solve(b["ref"], b["bounds"], b["scal"], passes=0)
    r1 = E.solve(b["ref"], b["bounds"], b["scal"], passes=1)
    assert r1["status"][0] == 1 and r1["info"][0, 4] == 2
    assert r1["info"][0, 7] > 36                      # (it did take the guarded rounds: more than three attempts' worth of plain ones)
    kkt_certificate_of_the_last_pass(b["ref"][0], b["bounds"][0], b["scal"][0], r0["out"][0], r1["out"][0])


def test_first_solve_only_given_linearisation_point_and_two_step_route():
    n = 50
    b = make_batch(8, n, "varied", seed=21)
    r0 = E.solve(b["ref"], b["bounds"], b["scal"], passes=0)
    for q in range(4):
        want = O.solve_path(b["ref"][q], b["bounds"][q], b["scal"][q], st=TIGHT, passes=0)[-1]["out"]
        assert np.abs(r0["out"][q][:, 3:6] - want[:, 3:6]).max() < 1e-6
    r1 = E.solve(b["ref"], b["bounds"], b["scal"], passes=0, lin=np.ascontiguousarray(r0["out"][:, :, 3:6]))
    r01 = E.solve(b["ref"], b["bounds"], b["scal"], passes=1)
    assert np.abs(r1["out"] - r01["out"]).max() < 1e-7
    rng = np.random.default_rng(1)
    lin = np.stack([O.first_linearization(b["ref"][q]) for q in range(2)]) + rng.normal(scale=[0.1, 0.02, 0.005], size=(2, n, 3))
    r = E.solve(b["ref"][:2], b["bounds"][:2], b["scal"][:2], passes=0, lin=lin)
    for q in range(2):
        want = O.solve_path(b["ref"][q], b["bounds"][q], b["scal"][q], st=TIGHT, passes=0, lin0=lin[q])[-1]["out"]
        assert np.abs(r["out"][q][:, 3:6] - want[:, 3:6]).max() < 1e-6


def test_rough_constraints_mode():
    n = 70
    b = make_batch(4, n)
    b["bounds"][:, :, 4] -= 0.15; b["bounds"][:, :, 5] += 0.1
    oprm = O.PathQpParams(rough_constraints_far_away=True, precise_planning_length=9.0)
    r = E.solve(b["ref"], b["bounds"], b["scal"], passes=1, prm=E.production(rough_constraints_far_away=1, precise_planning_length=9.0))
    for q in range(4):
        want = O.solve_path(b["ref"][q], b["bounds"][q], b["scal"][q], prm=oprm, st=TIGHT, passes=1)[-1]["out"]
        assert r["status"][q] == 1
        assert np.abs(r["out"][q][:, 3:6] - want[:, 3:6]).max() < 2e-6


def test_waypoint_count_per_qp_blocked_road_and_infeasible_start():
    n = 64
    b = make_batch(10, n, "varied", seed=5)
    n_of = np.array([64, 2, 1, 33, 40, 64, 17, 3, 50, 64], dtype=np.int32)
    b["scal"][n_of < n, 4] = 1.0             # a road cut short is blocked (ReferencePath::isBlocked(): no end-heading row, base_solver.cpp:254)
    b["scal"][5, 2] = 0.9                    # start curvature outside its box
    b["scal"][7, 4] = 0.0                    # ... and a 3-waypoint road that keeps its end-heading row: the heading is out of the controls' reach
    r = E.solve(b["ref"], b["bounds"], b["scal"], passes=1, n_of=n_of)
    assert r["status"][2] == 0 and r["status"][5] == 4 and r["status"][7] == 4
    assert r["iters"][7] < 40                # (no feasible point: seen by the stalled steps, not by running into the iteration limit)
    assert not OC.solve_path(OC.params(eps_abs=1e-6, eps_rel=1e-6, max_iter=4000), b["ref"][7, :3], b["bounds"][7, :3], b["scal"][7], passes=0)["ok"]
    for q in (0, 1, 3, 4, 6, 8, 9):
        m = int(n_of[q])
        assert r["status"][q] == 1
        want = OC.solve_path(OC.params(**TIGHT_C), b["ref"][q, :m], b["bounds"][q, :m], b["scal"][q], passes=1)["out"]
        assert np.abs(r["out"][q, :m, 3:5] - want[:, 3:5]).max() < 2e-5, q
        assert (r["out"][q, m:] == 0).all()
        alone = E.solve(b["ref"][q:q + 1, :m], b["bounds"][q:q + 1, :m], b["scal"][q:q + 1], passes=1)
        assert np.array_equal(alone["out"][0], r["out"][q, :m])              # a truncated QP is bit for bit the QP solved alone


def test_narrow_and_degenerate_corridors():
    """Corridors narrower than the 0.1 m minimum clearance keep their width (getSoftBounds, base_solver.cpp:290-295); a zero-width box is
    an equality row; the optimum is still the oracle's."""
    n = 40
    b = make_batch(6, n, seed=30)
    b["bounds"][0, 10:14, 0:4] = np.array([-0.02, 0.03, -0.02, 0.03])         # 5 cm wide
    b["bounds"][1, 20, 0:2] = np.array([0.4, 0.4])                            # zero width on the front circle
    b["bounds"][2, 5:30, 1] = 0.2; b["bounds"][2, 5:30, 3] = 0.2              # a long wall on the left
    r = E.solve(b["ref"], b["bounds"], b["scal"], passes=1)
    assert (r["status"] == 1).all()
    for q in range(3):
        want = OC.solve_path(OC.params(**TIGHT_C), b["ref"][q], b["bounds"][q], b["scal"][q], passes=1)["out"]
        assert np.abs(r["out"][q][:, 3:5] - want[:, 3:5]).max() < 2e-5, q


def test_fuzzed_scenarios_are_solved_right_or_reported_unsolved():
    """Corridors 0.1-3 m wide around a wandering centre, start states up to 2 m / 0.5 rad off the line, steering limits of 8-20 degrees,
    blocked roads, end headings up to 3 rad off, 2-130 waypoints.  Whatever is reported SOLVED is the oracle's path; what the oracle
    solves is solved here too unless the scenario is one of the end-heading U-turns (status MAX_ITER, reported after 30 iterations)."""
    rng = np.random.default_rng(0)
    prm = OC.params(**TIGHT_C)
    solved = wrong = missed = 0
    for trial in range(10):
        n = int(rng.choice([2, 3, 5, 17, 40, 80, 130]))
        B = 24
        b = make_batch(B, n, str(rng.choice(["uniform", "varied"])), seed=int(rng.integers(1, 1 << 30)))
        mode = trial % 5
        if mode == 1:
            w = rng.uniform(0.05, 1.5, (B, 1)); c = rng.uniform(-0.5, 0.5, (B, n))
            b["bounds"][:, :, 0] = c - w; b["bounds"][:, :, 1] = c + w; b["bounds"][:, :, 2] = c - w * rng.uniform(0.5, 1.5, (B, 1)); b["bounds"][:, :, 3] = c + w
        if mode == 2:
            b["scal"][:, 0] = rng.uniform(-2, 2, B); b["scal"][:, 1] = rng.uniform(-0.5, 0.5, B)
        if mode == 3:
            b["scal"][:, 5] = np.deg2rad(rng.uniform(8, 20, B)); b["scal"][:, 4] = rng.integers(0, 2, B)
        if mode == 4:
            b["scal"][:, 3] = rng.uniform(-3, 3, B)
        r = E.solve(b["ref"], b["bounds"], b["scal"], passes=1)
        for q in range(B):
            o = OC.solve_path(prm, b["ref"][q], b["bounds"][q], b["scal"][q], passes=1)
            if r["status"][q] == 1:
                solved += 1
                if o["ok"] and np.abs(o["out"][:, 3:5] - r["out"][q][:, 3:5]).max() > 1e-4:
                    wrong += 1
            elif o["ok"] and mode != 4:
                missed += 1
            assert r["iters"][q] <= 260                         # (two passes of at most 100 + the retry: nobody iterates for ever)
    assert wrong == 0 and missed == 0, (wrong, missed)
    assert solved > 150


def test_carrying_the_previous_cycle_s_optimum_changes_the_iterations_not_the_result():
    """PQP_OPT_CARRY_CYCLES: the first pass of QP k starts from what QP k's slot held after the previous solve (the same scenario one planning
    cycle earlier, synth.jitter_batch).  Same optimum (to the KKT test's 1e-7), a third fewer sweeps; a slot whose new QP has nothing to do with
    its old one is still solved."""
    from path_optimizer_2_amd.synth import jitter_batch
    B, n = 256, 80
    host = make_batch(B, n)
    c = E.Carried(B, n)
    sweeps = []
    for v in range(3):
        hv = jitter_batch(host, v)
        r = c.solve(host["ref"], hv["bounds"], hv["scal"])
        cold = E.solve(host["ref"], hv["bounds"], hv["scal"])
        assert (r["status"] == 1).all()
        assert np.abs(r["out"] - cold["out"]).max() < 5e-7
        sweeps.append((r["info"][:, 6].mean(), cold["info"][:, 6].mean()))
    assert sweeps[0][0] == sweeps[0][1]                       # nothing to carry in the first cycle
    assert sweeps[1][0] < 0.75 * sweeps[1][1] and sweeps[2][0] < 0.75 * sweeps[2][1]
    other = make_batch(B, n, "varied", seed=99)               # unrelated scenarios in the same slots
    r = c.solve(other["ref"], other["bounds"], other["scal"])
    cold = E.solve(other["ref"], other["bounds"], other["scal"])
    assert (r["status"] == 1).all() and np.abs(r["out"] - cold["out"]).max() < 5e-7


def test_golden_fixtures_by_a_route_without_admm():
    """tests/golden/path_n8, path_n80: `out_star` = the converged optimum of the QP linearised around `lin` (first solve only), `path_out` =
    the whole optimizePath.  The fixtures were generated by the oracle's ADMM; the lane-per-QP solver reaches the same points by interior-point
    + Riccati + active-set rounds - no iteration, factorisation or scaling in common with it."""
    import os
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name in ("path_n8", "path_n80"):
        g = np.load(os.path.join(root, name + ".npz"))
        r0 = E.solve(g["ref"], g["bounds"], g["scal"], passes=0, lin=g["lin"])
        r1 = E.solve(g["ref"], g["bounds"], g["scal"], passes=1)
        assert (r0["status"] == 1).all() and (r1["status"] == 1).all()
        assert np.abs(r0["out"] - g["out_star"]).max() < 1e-6, name
        assert np.abs(r1["out"][:, :, 3:5] - g["path_out"][:, :, 3:5]).max() < 1e-6, name


def test_start_curvature_outside_its_box_by_less_than_the_tolerance():
    """The one scenario of the robustness sweeps whose start curvature lies 3.8e-5 outside the curvature box: OSQP at eps 1e-4 - the reference - calls it
    solved, with a point that misses the row by that little.  Both solvers project the start state onto the box when the violation is within OSQP's primal
    tolerance and solve that QP exactly (tests/test_lane_emulation.py::test_a_start_curvature_outside_its_box_by_less_than_the_tolerance_is_projected): SOLVED,
    the same path; a start curvature outside the box by more than the tolerance stays PRIMAL_INFEASIBLE."""
    import emu_util as LANE
    b = make_batch(1, 80, "varied", seed=1007, first_qp=6640)
    r = E.solve(b["ref"], b["bounds"], b["scal"])
    assert r["status"][0] == 1 and r["info"][0, 4] == 2
    o = LANE.solve(LANE.production(), b["ref"], b["bounds"], b["scal"], passes=1)
    assert o["status"][0] == 1 and np.abs(r["out"][0][:, 3:5] - o["out"][0][:, 3:5]).max() < 1e-6
    sc = b["scal"].copy()
    sc[0, 2] = 0.9                                   # far outside: no point satisfies the rows
    assert E.solve(b["ref"], b["bounds"], sc)["status"][0] == 4


@pytest.mark.parametrize("what", ["nan_s", "nan_bound", "inf_start", "equal_s", "nan_pose"])
def test_a_scenario_that_is_not_a_number_ends_numerical_with_a_zero_record(what):
    """the lane-per-QP solver's input check (first preparation sweep): PQP_STATUS_NUMERICAL, zero record, no interior-point iteration on NaNs"""
    from path_optimizer_2_amd.synth import make_batch
    b = make_batch(3, 40, seed=31)
    clean = E.solve(b["ref"], b["bounds"], b["scal"], passes=1)
    h = {k: v.copy() for k, v in b.items()}
    if what == "nan_s": h["ref"][1, 10, 0] = np.nan
    if what == "nan_bound": h["bounds"][1, 5, 0] = np.nan
    if what == "inf_start": h["scal"][1, 1] = np.inf
    if what == "equal_s": h["ref"][1, 30, 0] = h["ref"][1, 29, 0]
    if what == "nan_pose": h["ref"][1, 39, 4] = np.nan
    r = E.solve(h["ref"], h["bounds"], h["scal"], passes=1)
    assert r["status"][1] == 3 and r["iters"][1] == 0 and np.all(r["out"][1] == 0.0)
    for q in (0, 2):
        assert r["status"][q] == 1 and np.array_equal(r["out"][q], clean["out"][q])
