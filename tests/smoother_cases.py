"""Synthetic inputs for the three smoother QPs (shared by CPU and GPU tests)."""
import numpy as np


def tension_inputs(n, seed=0, ds=1.0):
    """A noisy curved polyline re-sampled at ~1 m (what segmentRawReference hands to osqpSmooth,
    reference_path_smoother.cpp:47-105): x, y, heading, curvature, arclength lists."""
    rng = np.random.default_rng(seed)
    s = np.arange(n) * ds
    k = 0.05 * np.sin(s / 9.0 + rng.uniform(0, 6.28)) + rng.uniform(-0.01, 0.01)
    ang = np.concatenate([[0.3], 0.3 + np.cumsum(0.5 * (k[1:] + k[:-1]) * ds)])
    x = np.concatenate([[1.0], 1.0 + np.cumsum(np.cos(0.5 * (ang[1:] + ang[:-1])) * ds)]) + rng.normal(scale=0.05, size=n)
    y = np.concatenate([[-2.0], -2.0 + np.cumsum(np.sin(0.5 * (ang[1:] + ang[:-1])) * ds)]) + rng.normal(scale=0.05, size=n)
    clearance = rng.uniform(0.3, 3.0, size=n)
    return x, y, ang, k, s, clearance


def post_inputs(m, seed=0):
    rng = np.random.default_rng(seed)
    s = np.arange(m) * 1.5
    centre = 0.8 * np.sin(s / 11.0 + rng.uniform(0, 6.28))
    half = rng.uniform(0.4, 1.5, size=m)
    lb, ub = centre - half, centre + half
    return s, lb, ub, float(centre[0] + rng.uniform(-0.2, 0.2))
