"""Scene inputs for the bench tools, built through the PRODUCT path only (pqp_spline_fit, pqp_reference_states): the tools
must not depend on oracle/."""
import numpy as np
from path_optimizer_2_amd import capi
from path_optimizer_2_amd.synth import make_scene


def build(h, seeds, n_states, **scene_kw):
    """For every seed: a scene (distance layer + knots); splines fitted and `n_states` reference states sampled at the scene's
    fixed spacing by the engine.  Returns dict(dist [S][rows][cols], tab [S][9][m], ext [S][4], ref [S][n][5], geom, scenes)."""
    scenes = [make_scene(seed=s, n=n_states, **scene_kw) for s in seeds]
    ks = np.stack([sc["knots_s"] for sc in scenes]); kx = np.stack([sc["knots_x"] for sc in scenes]); ky = np.stack([sc["knots_y"] for sc in scenes])
    tab, ext = h.spline_fit(ks, kx, ky)
    sp = scenes[0]["spacing"]
    # fixed spacing: 20 additions of 0.3 give 6.000000000000001, so ask for half a step more than the last state needs
    ref, count, _ = h.reference_states(tab, ext, np.full(len(scenes), (n_states - 1) * sp + 0.5 * sp), n_states, ds_small=sp, ds_large=sp, dynamic=False)
    assert (count >= n_states).all(), count
    sc0 = scenes[0]
    geom = capi.PqpGridGeometry(sc0["rows"], sc0["cols"], sc0["resolution"], sc0["length"][0], sc0["length"][1], sc0["pos"][0], sc0["pos"][1])
    return dict(dist=np.stack([sc["dist"] for sc in scenes]), tab=tab, ext=ext, ref=ref, geom=geom, scenes=scenes)
