"""Generates the golden fixtures of tests/golden/*.npz from the oracle (oracle/pqp_oracle.py) and the synthetic
generator.  The reference holds no vectors for this path, so these are produced here, by the
build's own restatement, and pinned by the KKT certificate stored next to them.
Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pqp_oracle as O  # noqa: E402
from path_optimizer_2_amd.synth import make_batch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make(name, n, batch, profile):
    b = make_batch(batch, n, profile)
    rng = np.random.default_rng(n)
    rows, cols, colptr, pcols = O.structural_pattern(n, n)
    st = O.OsqpSettings(eps_abs=1e-10, eps_rel=1e-10, max_iter=100000)
    lin = np.zeros((batch, n, 3)); a_val = []; lower = []; upper = []; xs = []; ys = []; outs = []; cert = []
    for q in range(batch):
        lin[q] = O.first_linearization(b["ref"][q])
        if q % 2 == 1:
            lin[q] += rng.normal(scale=[0.2, 0.04, 0.01], size=(n, 3))
        Pd, A, lo, up, sz = O.assemble_path_qp(b["ref"][q], lin[q], b["bounds"][q], b["scal"][q])
        r = O.osqp_admm(sp.diags(Pd), np.zeros(sz["vars"]), A, lo, up, st)
        assert r["status"] == "solved"
        c = O.kkt_certificate(sp.diags(Pd), np.zeros(sz["vars"]), A, lo, up, r["x"], r["y"])
        a_val.append(A[rows, cols]); lower.append(lo); upper.append(up); xs.append(r["x"]); ys.append(r["y"])
        outs.append(O.unpack_path(r["x"], b["ref"][q])); cert.append([c["pri"], c["stat"], c["comp"]])
    # the two-pass pipeline output (optimizePath) for the un-perturbed start
    path_out = np.stack([O.solve_path(b["ref"][q], b["bounds"][q], b["scal"][q], st=st)[-1]["out"] for q in range(batch)])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), ref=b["ref"], bounds=b["bounds"], scal=b["scal"], lin=lin,
                        rows=rows, cols=cols, colptr=colptr, pcols=pcols, a_val=np.array(a_val), lower=np.array(lower),
                        upper=np.array(upper), x_star=np.array(xs), y_star=np.array(ys), out_star=np.array(outs),
                        cert=np.array(cert), path_out=path_out)
    print(name, "cert max", np.max(cert))


def make_scene_fixture(name, seed, length):
    """The chain around the QPs: knots -> tk::spline coefficients -> reference states -> corridor bounds, and the
    DP corridor search on the same scene.  The distance layer is stored as float16-safe float32 cropped to the part the line uses."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import corridor_oracle as K
    from path_optimizer_2_amd.synth import make_scene
    sc = make_scene(seed=seed, n=10, length=(50.0, 30.0), n_obstacles=30)
    sx = K.spline_fit(sc["knots_s"], sc["knots_x"]); sy = K.spline_fit(sc["knots_s"], sc["knots_y"])
    g = K.GridGeom(sc["rows"], sc["cols"], sc["resolution"], sc["length"][0], sc["length"][1], sc["pos"][0], sc["pos"][1])
    tab, ext = K.pack_spline(sx, sy)
    ref = K.build_reference_from_spline(sx, sy, length)
    start = np.array([ref[0, 3] + 0.2, ref[0, 4] + 0.4, ref[0, 2] + 0.03])
    init_err = np.array(K.process_init_state(sx, sy, *start))
    bounds, n_valid, blocked = K.update_bounds_improved(ref, sx, sy, sc["dist"], g)
    dp = K.graph_search_dp(sx, sy, length, tuple(start), sc["dist"], g)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), knots_s=sc["knots_s"], knots_x=sc["knots_x"], knots_y=sc["knots_y"], spline=tab,
                        spline_ext=ext, dist=sc["dist"], geom=np.array([g.rows, g.cols, g.resolution, g.length_x, g.length_y, g.pos_x, g.pos_y]),
                        length=length, start=start, ref=ref, init_err=init_err, bounds=bounds, n_valid=n_valid,
                        blocked_row=np.array(blocked if blocked is not None else []), dp_layers_s=dp["layers_s"], dp_lb=dp["lb"], dp_ub=dp["ub"],
                        dp_vehicle_l=dp["vehicle_l"])
    print(name, "states", len(ref), "usable", n_valid, "dp layers", len(dp["layers_s"]))


def make_line_fixture(name, seed):
    """The head of ReferencePathSmoother::solve: input points -> bSpline -> spline -> segmentRawReference, the tail of postSmooth
    (offsets -> points) on that spline, and setReferencePathLength for a target beside the line."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import corridor_oracle as K
    rng = np.random.default_rng(seed)
    n = 9
    pts = np.cumsum(np.column_stack([np.full(n, 3.7), rng.uniform(-1.2, 1.2, n)]), axis=0)
    x, y, s = K.bspline_resample(pts)
    sx, sy = K.spline_fit(s, x), K.spline_fit(s, y)
    tab, ext = K.pack_spline(sx, sy)
    seg = K.segment_raw_reference(sx, sy, float(s[-1]))
    at_s = np.linspace(0.5, float(s[-1]) - 0.5, 12)
    off = rng.uniform(-1.0, 1.0, 12)
    op = K.offsets_to_points(sx, sy, at_s, off)
    tx, ty = K.spline_eval(sx, 17.0) + 0.4, K.spline_eval(sy, 17.0) - 0.9
    np.savez_compressed(os.path.join(HERE, name + ".npz"), points=pts, raw_x=x, raw_y=y, raw_s=s, spline=tab, spline_ext=ext,
                        seg_x=seg[0], seg_y=seg[1], seg_s=seg[2], seg_angle=seg[3], seg_k=seg[4], at_s=at_s, offsets=off,
                        off_x=op[0], off_y=op[1], off_s=op[2], target=np.array([tx, ty, 0.0]),
                        cut_length=K.reference_length(sx, sy, float(s[-1]), tx, ty))
    print(name, "raw points", len(x), "samples", len(seg[2]))


def make_smoother_fixture(name):
    """The three smoothing QPs: inputs and the optimum of the oracle's assembly of the reference's formulation -
    TensionSmoother2's from the dense KKT system (it has equality rows only), TensionSmoother's and postSmooth's from the oracle's ADMM
    run to 1e-11 / 1e-10."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from smoother_cases import post_inputs, tension_inputs
    out = {}
    for tag, n, seed in (("a", 24, 901), ("b", 60, 902)):
        x, y, ang, k, s, cl = tension_inputs(n, seed=seed)
        P, q, A, lo, up = O.assemble_tension2(x, y, ang, k, s)
        sol = np.linalg.solve(np.block([[P, A.T], [A, np.zeros((A.shape[0], A.shape[0]))]]), np.r_[-q, lo])
        P2, q2, A2, lo2, up2 = O.assemble_tension(x, y, ang, cl)
        r2 = O.osqp_admm(sp.csc_matrix(P2), q2, A2, lo2, up2, O.OsqpSettings(eps_abs=1e-11, eps_rel=1e-11, max_iter=800000))
        assert r2["status"] == "solved"
        for key, val in (("x", x), ("y", y), ("angle", ang), ("k", k), ("s", s), ("clearance", cl), ("t2_x", sol[:n]), ("t2_y", sol[n:2 * n]),
                         ("t_x", r2["x"][:n]), ("t_y", r2["x"][n:2 * n])):
            out[f"{tag}_{key}"] = val
    for tag, m, seed in (("c", 18, 903), ("d", 60, 904)):
        s, lb, ub, l0 = post_inputs(m, seed=seed)
        P, q, A, lo, up = O.assemble_post(s, list(zip(lb, ub)), l0)
        r = O.osqp_admm(sp.csc_matrix(P), q, A, lo, up, O.OsqpSettings(eps_abs=1e-10, eps_rel=1e-10, max_iter=400000))
        assert r["status"] == "solved"
        for key, val in (("s", s), ("lb", lb), ("ub", ub), ("l0", np.array(l0)), ("l", r["x"][:m])):
            out[f"{tag}_{key}"] = val
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, sorted(out)[:6], "...")


if __name__ == "__main__":
    if "smoothers" in sys.argv[1:]:
        make_smoother_fixture("smoothers"); sys.exit(0)
    make_smoother_fixture("smoothers")
    make_line_fixture("line_a", 3)
    make("path_n8", 8, 4, "varied")
    make("path_n80", 80, 4, "uniform")
    make_scene_fixture("scene_a", 0, 24.0)
    make_scene_fixture("scene_b", 5, 30.0)
