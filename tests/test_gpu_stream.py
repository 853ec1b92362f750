"""The lane-per-QP path-QP kernel (path_stream_kernel, csrc/pqp_path_lq.hpp; PQP_OPT_STREAM_BATCH) on the GPU, through the C ABI, against
the converged C oracle, against the lane-per-waypoint kernel, and - at BASELINE configs[3]'s full size - by properties."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _handle(capi, batch, n, stream=True, **over):
    h = capi.Handle(capi.production_params(**over), device=0, max_batch=batch, max_n=n)
    h.set_option(capi.OPT_STORE_WARM, 0)
    h.set_option(capi.OPT_STREAM_BATCH, 1 if stream else 0)
    return h


def _oracle(b, k, n_of=None):
    import pqp_oracle_c as OC
    prm = OC.params(eps_abs=1e-9, eps_rel=1e-9, max_iter=40000)
    if n_of is None:
        return OC.solve_batch(prm, b["ref"][:k], b["bounds"][:k], b["scal"][:k], passes=1)["out"]
    outs = []
    for q in range(k):
        m = int(n_of[q])
        outs.append(OC.solve_path(prm, b["ref"][q, :m], b["bounds"][q, :m], b["scal"][q], passes=1)["out"])
    return outs


@pytest.mark.parametrize("n,profile,batch", [(80, "uniform", 256), (120, "varied", 192), (200, "uniform", 128), (35, "varied", 130)])
def test_stream_kernel_against_the_converged_oracle(hip_lib, n, profile, batch):
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.synth import make_batch
    b = make_batch(batch, n, profile, seed=11)
    h = _handle(capi, batch, n)
    r = h.solve(b["ref"], b["bounds"], b["scal"], passes=1)
    h.close()
    assert (r["status"] == 1).all()
    want = _oracle(b, 48)
    err = np.abs(r["out"][:48, :, 3:5] - want[:, :, 3:5]).max(axis=(1, 2))
    assert err.max() < 1e-4 and np.median(err) < 1e-6, (err.max(), np.median(err))
    assert np.abs(r["out"][:48] - want).max() < 1e-4          # x, y, heading, kappa, kappa' too
    assert (r["info"][:, 4] == 2).all()                       # both passes verified by an active-set round


def test_stream_kernel_equals_the_lane_per_waypoint_kernel(hip_lib):
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.synth import make_batch
    batch, n = 1024, 80
    b = make_batch(batch, n)
    ha, hb = _handle(capi, batch, n, stream=False), _handle(capi, batch, n, stream=True)
    ra = ha.solve(b["ref"], b["bounds"], b["scal"], passes=1)
    rb = hb.solve(b["ref"], b["bounds"], b["scal"], passes=1)
    ha.close(); hb.close()
    assert (ra["status"] == 1).all() and (rb["status"] == 1).all()
    d = np.abs(ra["out"][:, :, 3:5] - rb["out"][:, :, 3:5]).max(axis=(1, 2))
    assert d.max() < 2e-5 and np.median(d) < 1e-7, (d.max(), np.median(d))


def test_stream_kernel_carries_the_previous_cycle(hip_lib):
    """PQP_OPT_CARRY_CYCLES on the device: the same scenarios one planning cycle later start from their slots' previous optima - same paths as the
    cold solve, fewer sweeps; off by default; a change of shape starts cold."""
    import torch
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.synth import jitter_batch, make_batch
    batch, n = 1000, 80
    host = make_batch(batch, n, seed=8)
    dev = torch.device("cuda", 0)
    hc, hk = _handle(capi, batch, n), _handle(capi, batch, n)
    hc.set_option(capi.OPT_CARRY_CYCLES, 1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ref = t(host["ref"])
    for v in range(3):
        hv = jitter_batch(host, v, seed=8)
        bounds, scal = t(hv["bounds"]), t(hv["scal"])
        res = []
        for h in (hc, hk):
            out = torch.zeros((batch, n, 7), dtype=torch.float64, device=dev); st = torch.zeros(batch, dtype=torch.int32, device=dev)
            info = torch.zeros((batch, 8), dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            h.solve_device(batch, n, ref, bounds, scal, out, passes=1, status=st, info=info)
            h.sync()
            assert (st.cpu().numpy() == 1).all()
            res.append((out.cpu().numpy(), info.cpu().numpy()[:, 6].mean()))
        assert np.abs(res[0][0] - res[1][0]).max() < 5e-7
        if v == 0:
            assert res[0][1] == res[1][1]
        else:
            assert res[0][1] < 0.75 * res[1][1], (v, res[0][1], res[1][1])
    b2 = make_batch(batch, 64, seed=9)                           # another shape on the same handle: nothing to carry
    r2 = hc.solve(b2["ref"], b2["bounds"], b2["scal"], passes=1)
    k2 = hk.solve(b2["ref"], b2["bounds"], b2["scal"], passes=1)
    assert np.array_equal(r2["out"], k2["out"])
    hc.close(); hk.close()


def test_stream_kernel_first_solve_only_and_a_given_linearisation_point(hip_lib):
    """passes = 0 is BaseSolver::solve alone; lin != NULL is updateProblemFormulationAndSolve's QP solved cold."""
    import pqp_oracle_c as OC
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.synth import make_batch
    batch, n = 96, 80
    b = make_batch(batch, n, seed=5)
    h = _handle(capi, batch, n)
    r0 = h.solve(b["ref"], b["bounds"], b["scal"], passes=0)
    lin = np.ascontiguousarray(r0["out"][:, :, 3:6])
    r1 = h.solve(b["ref"], b["bounds"], b["scal"], passes=0, lin=lin)
    r01 = h.solve(b["ref"], b["bounds"], b["scal"], passes=1)
    h.close()
    prm = OC.params(eps_abs=1e-9, eps_rel=1e-9, max_iter=40000)
    for q in range(12):
        w0 = OC.solve_path(prm, b["ref"][q], b["bounds"][q], b["scal"][q], passes=0)["out"]
        assert np.abs(r0["out"][q, :, 3:5] - w0[:, 3:5]).max() < 1e-5
    assert np.abs(r1["out"] - r01["out"]).max() < 1e-7           # the two-step route gives the fused launch's paths


def test_stream_kernel_with_a_waypoint_count_per_qp_and_an_infeasible_start(hip_lib):
    import torch
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.synth import make_batch
    batch, n = 70, 96
    b = make_batch(batch, n, "varied", seed=3)
    rng = np.random.default_rng(1)
    n_of = rng.integers(2, n + 1, batch).astype(np.int32)
    n_of[0] = n; n_of[1] = 2; n_of[2] = 1                        # full, the shortest QP, nothing to optimise
    b["scal"][n_of < n, 4] = 1.0                                 # a road cut short is blocked: no end-heading row (base_solver.cpp:254)
    b["scal"][5, 2] = 1.0                                        # a start curvature outside its box: no feasible point
    h = _handle(capi, batch, n)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ref, bounds, scal, cnt = t(b["ref"]), t(b["bounds"]), t(b["scal"]), t(n_of)
    out = torch.zeros((batch, n, 7), dtype=torch.float64, device=dev)
    st = torch.zeros(batch, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    h.solve_var_device(batch, n, cnt, ref, bounds, scal, out, passes=1, status=st)
    h.sync()
    out, st = out.cpu().numpy(), st.cpu().numpy()
    h.close()
    assert st[2] == 0 and st[5] == 4                             # PQP_STATUS_UNSOLVED, PQP_STATUS_PRIMAL_INFEASIBLE
    ok = np.ones(batch, bool); ok[[2, 5]] = False
    assert (st[ok] == 1).all()
    want = _oracle(b, 16, n_of)
    for q in range(16):
        if ok[q]:
            m = int(n_of[q])
            assert np.abs(out[q, :m, 3:5] - want[q][:, 3:5]).max() < 2e-5, q
            assert (out[q, m:] == 0).all()                       # rows beyond a QP's own count are not written


def test_stream_kernel_rough_constraints_far_away(hip_lib):
    """base_solver.cpp:25-34,201-205: beyond precise_planning_length one collision row per waypoint on the centre circle."""
    import pqp_oracle as O
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.synth import make_batch
    batch, n = 64, 80
    b = make_batch(batch, n, seed=9)
    b["bounds"][:, :, 4] -= 0.15; b["bounds"][:, :, 5] += 0.1   # a centre box of its own
    h = _handle(capi, batch, n, rough_constraints_far_away=1, precise_planning_length=12.0)
    r = h.solve(b["ref"], b["bounds"], b["scal"], passes=1)
    h.close()
    assert (r["status"] == 1).all()
    prm = O.PathQpParams(rough_constraints_far_away=True, precise_planning_length=12.0)
    st = O.OsqpSettings(eps_abs=1e-9, eps_rel=1e-9, max_iter=40000)
    for q in range(4):
        want = O.solve_path(b["ref"][q], b["bounds"][q], b["scal"][q], prm=prm, st=st)[-1]["out"]
        assert np.abs(r["out"][q, :, 3:5] - want[:, 3:5]).max() < 2e-5, q


@pytest.mark.gpu
@pytest.mark.parametrize("n,profile,batch", [(80, "uniform", 4096), (120, "varied", 2048), (300, "varied", 192), (9, "varied", 1000)])
def test_the_two_forms_of_the_kernel_agree(hip_lib, n, profile, batch):
    """PQP_OPT_STREAM_STAGED: the [field][lane] workspace with the register prefetch (launches that fill the chip) against the [chunk][lane] workspace whose records
    are staged in LDS two waypoints ahead by LDS-direct loads (launches that leave SIMDs idle) - the same solver over other addresses: every QP's status and
    counts equal, the paths equal up to the fused multiply-adds the compiler picks in the two instantiations; ragged paths; run twice (a record read before its
    copy has landed would show as a run-to-run difference)."""
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.synth import make_batch
    b = make_batch(batch, n, profile, seed=41)
    n_of = np.full(batch, n, dtype=np.int32)
    n_of[::5] = max(n - 7, 2); n_of[2::9] = max(n // 2, 2); n_of[1] = 2
    b["scal"][n_of < n, 4] = 1.0
    res = {}
    for form in (0, 1):
        h = _handle(capi, batch, n)
        h.set_option(capi.OPT_STREAM_STAGED, form)
        runs = [h.solve_var(n_of, b["ref"], b["bounds"], b["scal"], passes=1) for _ in range(2)]
        assert h.last_path_kernel() == capi.KERNEL_LANE_PER_QP
        h.close()
        assert np.array_equal(runs[0]["out"], runs[1]["out"]) and np.array_equal(runs[0]["info"], runs[1]["info"]), form
        res[form] = runs[0]
    assert np.array_equal(res[0]["status"], res[1]["status"])
    same = (res[0]["info"][:, 2:8] == res[1]["info"][:, 2:8]).all(axis=1)
    assert same.mean() > 0.99, same.mean()                 # (a borderline row may flip a count between two compilations of the arithmetic)
    ok = (res[0]["status"] == 1) & same
    d = 0.0
    for q in np.nonzero(ok)[0]:
        m = int(n_of[q])
        d = max(d, float(np.abs(res[0]["out"][q, :m] - res[1]["out"][q, :m]).max()))
    assert d < 1e-6, d                                       # (1.7e-7 seen at 120 and 300 waypoints: the flat direction of the lateral offsets, weight_l = 0)


def test_stream_kernel_on_the_whole_of_configs_3(hip_lib):
    """BASELINE configs[3] at its full 65 536 QPs of 80 waypoints on ONE GPU: every QP verified, deterministic run to run, and a sample
    of the paths against the converged oracle."""
    import torch
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.synth import make_batch
    batch, n = 65536, 80
    b = make_batch(batch, n)
    dev = torch.device("cuda", 0)
    ref, bounds, scal = (torch.from_numpy(b[k]).to(dev) for k in ("ref", "bounds", "scal"))
    h = _handle(capi, batch, n)
    shas = []
    for _ in range(2):
        out = torch.zeros((batch, n, 7), dtype=torch.float64, device=dev)
        st = torch.zeros(batch, dtype=torch.int32, device=dev)
        info = torch.zeros((batch, 8), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        h.solve_device(batch, n, ref, bounds, scal, out, passes=1, status=st, info=info)
        h.sync()
        o = out.cpu().numpy()
        shas.append(hashlib.sha1(o.tobytes()).hexdigest())
    h.close()
    assert (st.cpu().numpy() == 1).all()
    assert shas[0] == shas[1]
    # PQP_OPT_ORDER_BY_COST on this kernel (round 5): from the second solve on the wavefronts hold QPs that ran the same phases in the previous one - also on a
    # jittered batch, whose map comes from another batch's counts - every QP solved exactly once, and the wavefronts' lock-step phase maxima come down
    # from ~25 to ~17 per wavefront.  Round 6: in such a sorted launch the re-linearised pass first tries the first pass's active set (lq::kDirectRounds
    # active-set rounds before the interior-point rounds), so a sorted launch's path is the same optimum through other arithmetic: within the rounds'
    # tolerances of the unsorted launch's, bit for bit the same from one sorted launch to the next (a QP's arithmetic does not depend on its slot), and
    # ~98 % of the QPs end their second pass without an interior-point iteration.
    from path_optimizer_2_amd.synth import jitter_batch
    ho = _handle(capi, batch, n)
    ho.set_option(capi.OPT_ORDER_BY_COST, 1)
    hv = jitter_batch(b, 1)
    bounds_v, scal_v = torch.from_numpy(hv["bounds"]).to(dev), torch.from_numpy(hv["scal"]).to(dev)
    sorted_shas = []
    for k, (bb, ss) in enumerate(((bounds, scal), (bounds, scal), (bounds_v, scal_v), (bounds, scal))):
        out.zero_(); st.zero_(); info.zero_()
        ho.solve_device(batch, n, ref, bb, ss, out, passes=1, status=st, info=info)
        ho.sync()
        assert (st.cpu().numpy() == 1).all(), k
        ok = out.cpu().numpy()
        inf = info.cpu().numpy()
        if k == 0:          # no map yet: the unsorted launch
            assert hashlib.sha1(ok.tobytes()).hexdigest() == shas[0]
            assert (inf[:, 3] - inf[:, 2] > 0).all()
        elif k in (1, 3):
            sorted_shas.append(hashlib.sha1(ok.tobytes()).hexdigest())
            assert np.abs(ok - o)[:, :, 3:6].max() < 5e-7
            assert np.abs(ok - o)[:, :, 0:2].max() < 5e-7
            hit = inf[:, 3] - inf[:, 2] == 0
            assert 0.95 < hit.mean() < 1.0, hit.mean()
    assert sorted_shas[0] == sorted_shas[1]
    ho.close()
    # properties of an optimum that need no oracle: start state, curvature box, end box, x / y consistent with l
    assert np.abs(o[:, 0, 3] - b["scal"][:, 0]).max() < 1e-12 and np.abs(o[:, 0, 5] - b["scal"][:, 2]).max() < 1e-12
    assert np.abs(o[:, :, 5]).max() <= np.tan(35 * np.pi / 180) / 2.5 + 1e-8
    assert np.abs(o[:, -1, 3]).max() <= 1.0 + 1e-8
    nx = b["ref"][:, :, 3] + o[:, :, 3] * np.cos(b["ref"][:, :, 2] + np.pi / 2)
    assert np.abs(o[:, :, 0] - nx).max() < 1e-9
    idx = np.linspace(0, batch - 1, 24).astype(int)
    want = _oracle({k: v[idx] for k, v in b.items()}, 24)
    assert np.abs(o[idx][:, :, 3:5] - want[:, :, 3:5]).max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n,profile", [(200, "varied"), (120, "varied")])
def test_sorted_launches_of_other_shapes(hip_lib, n, profile):
    """The re-linearised pass of a sorted launch (PQP_OPT_ORDER_BY_COST, 768 wavefronts or more) on longer, varied and ragged paths: fewer sets survive the
    re-linearisation (93 % at 120 waypoints, 86 % at 200), the others fall back to the interior-point rounds from the kept-aside optimum.  Same
    statuses as the unsorted launch, the same optimum, and against the device emulation's counts QP by QP on a sample."""
    import torch
    import lq_emu_util as E
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.synth import make_batch
    batch = 49152
    b = make_batch(batch, n, profile, seed=31)
    n_of = np.full(batch, n, dtype=np.int32)
    n_of[::7] = n - 13; n_of[3::11] = max(n // 2, 2)
    b["scal"][n_of < n, 4] = 1.0                                 # a road cut short is blocked: no end-heading row (base_solver.cpp:254)
    dev = torch.device("cuda", 0)
    ref, bounds, scal, nof = (torch.from_numpy(v).to(dev) for v in (b["ref"], b["bounds"], b["scal"], n_of))
    h = _handle(capi, batch, n)
    h.set_option(capi.OPT_ORDER_BY_COST, 1)
    res = []
    for k in range(3):
        out = torch.zeros((batch, n, 7), dtype=torch.float64, device=dev)
        st = torch.zeros(batch, dtype=torch.int32, device=dev)
        info = torch.zeros((batch, 8), dtype=torch.float64, device=dev)
        h.solve_var_device(batch, n, nof, ref, bounds, scal, out, passes=1, status=st, info=info)
        h.sync()
        res.append((out.cpu().numpy(), st.cpu().numpy(), info.cpu().numpy()))
    h.close()
    (o0, s0, i0), (o1, s1, i1), (o2, s2, i2) = res
    assert (s0 == s1).all() and (s1 == s2).all() and (s0 == 1).mean() > 0.999
    ok = s0 == 1
    assert np.abs(o1[ok] - o0[ok])[:, :, 3:6].max() < 5e-7
    assert (o1 == o2).all() and (i1 == i2).all()
    assert (i0[ok, 3] - i0[ok, 2] > 0).all()                    # the unsorted launch: every second pass ran interior-point iterations
    hit = i1[ok, 3] - i1[ok, 2] == 0
    assert 0.75 < hit.mean() < 0.99, hit.mean()
    # the same source on the host, sorted launch: the same counts per QP
    k = 256
    r = E.solve(b["ref"][:k], b["bounds"][:k], b["scal"][:k], passes=1, n_of=n_of[:k], sorted_launch=True)
    assert (r["status"] == s1[:k]).all()
    same = (r["info"][:, 2:8] == i1[:k, 2:8]).all(axis=1)
    assert same.mean() > 0.97, same.mean()                      # (FMA contraction differs between the two compilers: a borderline row may flip a count)
    sel = r["status"] == 1
    assert np.abs(r["out"][sel] - o1[:k][sel])[:, :, 3:6].max() < 5e-7


def test_paths_of_more_than_512_waypoints(hip_lib):
    """The reference has no size limit (an 80 m line at 0.15 m spacing: 530 waypoints, reference_path_impl.cpp:321-336).  Beyond the
    lane-per-waypoint kernel's 512 lanes pqp_path_solve runs the lane-per-QP kernel whatever the batch."""
    import pqp_oracle_c as OC
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.synth import make_batch
    batch, n = 24, 700
    b = make_batch(batch, n, "varied", seed=17)
    h = capi.Handle(capi.production_params(), device=0, max_batch=batch, max_n=n)       # (default options: warm state on, no stream threshold touched)
    r = h.solve(b["ref"], b["bounds"], b["scal"], passes=1)
    assert (r["status"] == 1).all()
    prm = OC.params(eps_abs=1e-9, eps_rel=1e-9, max_iter=60000)
    for q in range(3):
        want = OC.solve_path(prm, b["ref"][q], b["bounds"][q], b["scal"][q], passes=1)["out"]
        assert np.abs(r["out"][q][:, 3:5] - want[:, 3:5]).max() < 1e-4, q
    # BaseSolver::solve, then updateProblemFormulationAndSolve (warm == 1 with the first solution as `lin`): beyond 512 waypoints the second
    # QP is solved cold - the same optimum as the fused launch's
    r0 = h.solve(b["ref"], b["bounds"], b["scal"], passes=0)
    r1 = h.solve(b["ref"], b["bounds"], b["scal"], passes=0, warm=True, lin=np.ascontiguousarray(r0["out"][:, :, 3:6]))
    # (two routes to one optimum - interior-point rounds from the first pass's optimum, or cold - end in active-set rounds whose tests have 1e-7 of slack: 1.1e-7 on one of these QPs)
    assert (r1["status"] == 1).all() and np.abs(r1["out"] - r["out"]).max() < 5e-7, np.abs(r1["out"] - r["out"]).max()
    with pytest.raises(capi.PqpError):
        h.get_solution(batch, n)                                                         # (no OSQP-style workspace behind that kernel)
    h.close()
    h = capi.Handle(capi.default_params(), device=0, max_batch=batch, max_n=n)          # the reference's ADMM setting (what the BaseSolver shim runs)
    rd = h.solve(b["ref"], b["bounds"], b["scal"], passes=1)
    h.close()
    assert (rd["status"] == 1).all() and np.array_equal(rd["out"], r["out"])             # exact optima there too: they meet eps 2e-3


@pytest.mark.gpu
def test_long_paths_whose_active_set_rounds_cycle(hip_lib):
    """The QPs of tests/test_lq_emulation.py::CYCLING on the device: the plain active-set rounds of all three attempts cycle on them; the
    guarded rounds of the last attempt (pqp_path_lq.hpp forward_set<true>: steps as long as the piecewise quadratic objective falls along
    them) end at the optimum - SOLVED, and the returned path passes the KKT conditions of the assembled QP of the last pass."""
    from path_optimizer_2_amd import capi
    from path_optimizer_2_amd.synth import make_batch
    from test_lq_emulation import CYCLING, kkt_certificate_of_the_last_pass
    for n, profile, seed, qp in CYCLING:
        kw = {} if seed is None else {"seed": seed}
        # the QP among 63 neighbours: one wavefront, its other lanes on QPs of ordinary cost
        b = make_batch(64, n, profile, first_qp=qp - 10, **kw)
        h = capi.Handle(capi.production_params(), device=0, max_batch=64, max_n=n)
        h.set_option(capi.OPT_STORE_WARM, 0)
        h.set_option(capi.OPT_STREAM_BATCH, 1)
        r0 = h.solve(b["ref"], b["bounds"], b["scal"], passes=0)
        r1 = h.solve(b["ref"], b["bounds"], b["scal"], passes=1)
        assert h.last_path_kernel() == capi.KERNEL_LANE_PER_QP
        h.close()
        assert (r1["status"] == 1).all() and (r1["info"][:, 4] == 2).all()
        assert r1["info"][10, 7] > 36, r1["info"][10]
        kkt_certificate_of_the_last_pass(b["ref"][10], b["bounds"][10], b["scal"][10], r0["out"][10], r1["out"][10])
