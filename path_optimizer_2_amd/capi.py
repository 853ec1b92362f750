"""ctypes binding of the C ABI in include/pqp.h (libpqp_hip.so).

Python is plumbing here (tests, bench, smoke); the product is the shared library.  There is no CPU
fallback: if the library or a HIP device is missing, loading / pqp_create fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpqp_hip.so")


class PqpParams(C.Structure):
    _fields_ = [
        ("front_length", C.c_double), ("rear_length", C.c_double), ("wheel_base", C.c_double),
        ("expected_safety_margin", C.c_double), ("precise_planning_length", C.c_double),
        ("constraint_end_heading", C.c_int32), ("rough_constraints_far_away", C.c_int32),
        ("weight_l", C.c_double), ("weight_kappa", C.c_double), ("weight_dkappa", C.c_double),
        ("weight_slack", C.c_double), ("end_l_bound", C.c_double), ("end_psi_tol", C.c_double),
        ("end_psi_max", C.c_double), ("min_clearance", C.c_double),
        ("eps_abs", C.c_double), ("eps_rel", C.c_double), ("rho", C.c_double), ("sigma", C.c_double),
        ("alpha", C.c_double), ("max_iter", C.c_int32), ("scaling", C.c_int32),
        ("adaptive_rho", C.c_int32), ("adaptive_rho_interval", C.c_int32),
        ("adaptive_rho_tolerance", C.c_double), ("check_termination", C.c_int32), ("polish", C.c_int32),
        ("polish_refine_iter", C.c_int32), ("polish_every", C.c_int32), ("polish_warm_set", C.c_int32), ("polish_max_rounds", C.c_int32),
        ("polish_reseed", C.c_int32), ("polish_diverge", C.c_int32), ("polish_delta", C.c_double), ("polish_tol", C.c_double),
        ("polish_reseed_factor", C.c_double), ("eps_prim_inf", C.c_double), ("polish_patience", C.c_int32), ("prim_inf_after", C.c_int32),
        ("polish_lazy", C.c_int32), ("polish_final_refine", C.c_int32),
        ("tension2_deviation_weight", C.c_double), ("tension2_curvature_weight", C.c_double),
        ("tension2_curvature_rate_weight", C.c_double), ("cartesian_curvature_weight", C.c_double),
        ("cartesian_curvature_rate_weight", C.c_double), ("cartesian_deviation_weight", C.c_double),
    ]


class PqpGridGeometry(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("resolution", C.c_double), ("length_x", C.c_double), ("length_y", C.c_double),
                ("pos_x", C.c_double), ("pos_y", C.c_double)]


class PqpCorridorParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("front_length", "rear_length", "car_width", "safety_margin", "epsilon", "search_radius", "delta_s",
                                         "smaller_ds", "search_range", "min_space", "projection_window")]


class PqpDpParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("lateral_range", "longitudinal_spacing", "lateral_spacing", "car_width")]


class PqpChainConfig(C.Structure):
    _fields_ = [("raw_max", C.c_int32), ("sample_max", C.c_int32), ("layer_max", C.c_int32), ("n_max", C.c_int32), ("output_spacing", C.c_double),
                ("dynamic_segmentation", C.c_int32), ("max_steering_angle", C.c_double), ("smoothed_length_margin", C.c_double),
                ("corridor", PqpCorridorParams), ("dp", PqpDpParams), ("smoothing_method", C.c_int32)]


class PqpSizes(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("n", "state", "control", "precise", "slack", "vars", "cons", "nnz_a", "nnz_p")]


EXPORTS = [
    "pqp_default_params", "pqp_production_params", "pqp_last_error", "pqp_version", "pqp_create", "pqp_destroy", "pqp_set_params", "pqp_set_option",
    "pqp_constrain_angle_device", "pqp_stream_wait", "pqp_mark", "pqp_wait_mark", "pqp_get_stream",
    "pqp_chain_default_config", "pqp_optimize_path_device", "pqp_clearance_device", "pqp_smooth_tension2_var_device", "pqp_smooth_tension_var_device", "pqp_post_smooth_var_device", "pqp_spline_fit_var_device",
    "pqp_shard_range", "pqp_multi_create", "pqp_multi_destroy", "pqp_multi_shards", "pqp_multi_handle", "pqp_multi_set_option", "pqp_multi_path_solve", "pqp_multi_gather_paths", "pqp_multi_gather_ranks", "pqp_sync", "pqp_path_sizes", "pqp_path_pattern", "pqp_path_assemble",
    "pqp_path_assemble_device", "pqp_path_solve", "pqp_path_solve_device", "pqp_path_solve_var_device", "pqp_path_solve_var", "pqp_path_get_solution",
    "pqp_last_kernel_ms", "pqp_last_path_kernel", "pqp_stream_batch_default", "pqp_kernel_ms_history", "pqp_smooth_tension2", "pqp_smooth_tension2_device", "pqp_smooth_tension", "pqp_smooth_tension_device",
    "pqp_post_smooth", "pqp_post_smooth_device", "pqp_corridor_default_params", "pqp_corridor_bounds", "pqp_corridor_bounds_device",
    "pqp_reference_states", "pqp_reference_states_device", "pqp_spline_fit", "pqp_spline_fit_device", "pqp_dp_default_params",
    "pqp_dp_corridor", "pqp_dp_corridor_device", "pqp_segment_raw_reference", "pqp_segment_raw_reference_device", "pqp_bspline_resample", "pqp_bspline_resample_device", "pqp_reference_length", "pqp_reference_length_device", "pqp_offsets_to_points", "pqp_offsets_to_points_device",
]

_lib = None


def load_library(path=None, with_torch=None):
    """dlopen libpqp_hip.so and declare the prototypes.  Raises OSError if it was not built.
    with_torch: import torch BEFORE the library is mapped (see below).  None = yes unless PQP_CAPI_TORCH=0: the safe default for a process that
    may use torch later; a torch-free user of the C ABI passes False (or sets PQP_CAPI_TORCH=0) and skips torch's multi-second import."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("PQP_LIB") or LIB_PATH        # PQP_LIB: an alternative build of the library (experiments)
    if not os.path.exists(path):
        raise OSError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # A process that also uses torch must load torch's HIP runtime FIRST: the wheel bundles its own libamdhip64 under the soname this library needs
    # too, and whichever copy is mapped first serves both - with /opt/rocm's mapped first, torch finds no device ("No HIP GPUs are available").
    # (The device-resident entry points of this binding use torch as memory plumbing; the library itself does not depend on it.)
    import sys
    if with_torch is None:
        with_torch = os.environ.get("PQP_CAPI_TORCH", "1") != "0"
    if with_torch or "torch" in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(path)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
    lib.pqp_default_params.argtypes = [C.POINTER(PqpParams)]
    lib.pqp_default_params.restype = None
    lib.pqp_production_params.argtypes = [C.POINTER(PqpParams)]
    lib.pqp_production_params.restype = None
    lib.pqp_last_error.restype = C.c_char_p
    lib.pqp_version.restype = C.c_char_p
    lib.pqp_create.argtypes = [C.POINTER(vp), C.POINTER(PqpParams), C.c_int, C.c_int, C.c_int]
    lib.pqp_destroy.argtypes = [vp]
    lib.pqp_set_params.argtypes = [vp, C.POINTER(PqpParams)]
    lib.pqp_set_option.argtypes = [vp, C.c_int, C.c_int]
    lib.pqp_stream_wait.argtypes = [vp, vp]
    lib.pqp_mark.argtypes = [vp, C.c_int]
    lib.pqp_constrain_angle_device.argtypes = [vp, C.c_int, vp, vp]
    lib.pqp_wait_mark.argtypes = [vp, vp, C.c_int]
    lib.pqp_get_stream.argtypes = [vp, C.POINTER(vp)]
    lib.pqp_chain_default_config.argtypes = [C.POINTER(PqpChainConfig)]
    lib.pqp_chain_default_config.restype = None
    lib.pqp_optimize_path_device.argtypes = [vp, vp, C.POINTER(PqpChainConfig), C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.POINTER(PqpGridGeometry),
                                             vp, vp, vp, vp, vp, vp]
    lib.pqp_clearance_device.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, C.POINTER(PqpGridGeometry), vp]
    lib.pqp_smooth_tension2_var_device.argtypes = [vp, C.c_int, C.c_int] + [vp] * 12
    lib.pqp_smooth_tension_var_device.argtypes = [vp, C.c_int, C.c_int] + [vp] * 11
    lib.pqp_post_smooth_var_device.argtypes = [vp, C.c_int, C.c_int] + [vp] * 9
    lib.pqp_spline_fit_var_device.argtypes = [vp, C.c_int, C.c_int] + [vp] * 6
    lib.pqp_shard_range.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip]
    lib.pqp_shard_range.restype = None
    lib.pqp_multi_create.argtypes = [C.POINTER(vp), C.POINTER(PqpParams), C.c_int, ip, C.c_int, C.c_int]
    lib.pqp_multi_destroy.argtypes = [vp]
    lib.pqp_multi_shards.argtypes = [vp]
    lib.pqp_multi_handle.argtypes = [vp, C.c_int]
    lib.pqp_multi_handle.restype = vp
    lib.pqp_multi_set_option.argtypes = [vp, C.c_int, C.c_int]
    lib.pqp_multi_path_solve.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp]
    lib.pqp_multi_gather_paths.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    lib.pqp_multi_gather_ranks.argtypes = [vp]
    lib.pqp_sync.argtypes = [vp]
    lib.pqp_path_sizes.argtypes = [C.POINTER(PqpParams), C.c_int, vp, C.POINTER(PqpSizes)]
    lib.pqp_path_pattern.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    for name in ("pqp_path_assemble", "pqp_path_assemble_device"):
        getattr(lib, name).argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    for name in ("pqp_path_solve", "pqp_path_solve_device"):
        getattr(lib, name).argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.pqp_path_get_solution.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.pqp_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.pqp_last_path_kernel.argtypes = [vp]
    lib.pqp_kernel_ms_history.argtypes = [vp, vp, C.c_int]
    lib.pqp_smooth_tension2.argtypes = [vp, C.c_int, C.c_int] + [vp] * 10
    lib.pqp_smooth_tension2_device.argtypes = [vp, C.c_int, C.c_int] + [vp] * 11
    lib.pqp_smooth_tension.argtypes = [vp, C.c_int, C.c_int] + [vp] * 9
    lib.pqp_smooth_tension_device.argtypes = [vp, C.c_int, C.c_int] + [vp] * 10
    lib.pqp_post_smooth.argtypes = [vp, C.c_int, C.c_int] + [vp] * 7
    lib.pqp_post_smooth_device.argtypes = [vp, C.c_int, C.c_int] + [vp] * 8
    lib.pqp_corridor_default_params.argtypes = [C.POINTER(PqpCorridorParams)]
    lib.pqp_corridor_default_params.restype = None
    lib.pqp_corridor_bounds_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.POINTER(PqpGridGeometry),
                                               C.POINTER(PqpCorridorParams), vp, vp]
    lib.pqp_corridor_bounds.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, C.POINTER(PqpGridGeometry),
                                        C.POINTER(PqpCorridorParams), vp, vp]
    for name in ("pqp_reference_states", "pqp_reference_states_device"):
        getattr(lib, name).argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_double, C.c_double, C.c_int, vp, vp, vp]
    for name in ("pqp_offsets_to_points", "pqp_offsets_to_points_device"):
        getattr(lib, name).argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    for name in ("pqp_reference_length", "pqp_reference_length_device"):
        getattr(lib, name).argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    for name in ("pqp_bspline_resample", "pqp_bspline_resample_device"):
        getattr(lib, name).argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    for name in ("pqp_segment_raw_reference", "pqp_segment_raw_reference_device"):
        getattr(lib, name).argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_double, vp, vp, vp, vp, vp, vp]
    for name in ("pqp_spline_fit", "pqp_spline_fit_device"):
        getattr(lib, name).argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    lib.pqp_dp_default_params.argtypes = [C.POINTER(PqpDpParams)]
    lib.pqp_dp_default_params.restype = None
    lib.pqp_dp_corridor_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.POINTER(PqpGridGeometry), C.POINTER(PqpDpParams),
                                           vp, vp, vp, vp, vp]
    lib.pqp_dp_corridor.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, vp, C.POINTER(PqpGridGeometry), C.POINTER(PqpDpParams),
                                    vp, vp, vp, vp, vp]
    for name in ("pqp_path_solve_var_device", "pqp_path_solve_var"):
        getattr(lib, name).argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    if path == LIB_PATH:
        _lib = lib
    return lib


def default_params(lib=None, **over):
    lib = lib or load_library()
    p = PqpParams()
    lib.pqp_default_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def production_params(lib=None, **over):
    """pqp_production_params: the defaults + the engine's production solver setting (1e-4 + KKT-verified polish)."""
    lib = lib or load_library()
    p = PqpParams()
    lib.pqp_production_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(a))      # raw device pointer (e.g. torch.Tensor.data_ptr())


OPT_STORE_WARM, OPT_ORDER_BY_COST, OPT_RESERVE_CUS, OPT_STREAM_BATCH, OPT_CARRY_CYCLES, OPT_CHAIN_GRAPH, OPT_STREAM_STAGED = 1, 2, 3, 4, 5, 6, 7
def stream_batch_default(n, lib=None):
    """pqp_stream_batch_default: from how many QPs of n waypoints on a cold call runs on the lane-per-QP kernel (PQP_OPT_STREAM_BATCH's default)."""
    lib = lib or load_library()
    lib.pqp_stream_batch_default.restype = C.c_int
    return int(lib.pqp_stream_batch_default(C.c_int(n)))


def path_interval(value, n):
    """What pqp_params.adaptive_rho_interval / check_termination / polish_every mean for paths of n waypoints: negative values (the production
    setting) stand for "by path length" (csrc/pqp_defaults.hpp path_interval: 5 iterations up to 90 waypoints, 8 beyond)."""
    return value if value >= 0 else (5 if n <= 90 else 8)


KERNEL_NONE, KERNEL_LANE_PER_WAYPOINT, KERNEL_LANE_PER_QP = 0, 1, 2      # pqp_last_path_kernel
SMOOTHING_TENSION2, SMOOTHING_TENSION = 0, 1


class MultiHandle:
    """pqp_multi: one handle + one host thread per shard of the batch (several GPUs of one node; devices may repeat)."""

    def __init__(self, params=None, devices=(0,), max_batch_per_shard=1024, max_n=128):
        self.lib = load_library(with_torch=True)
        self.params = params or default_params(self.lib)
        self._m = C.c_void_p()
        devs = (C.c_int32 * len(devices))(*devices)
        rc = self.lib.pqp_multi_create(C.byref(self._m), C.byref(self.params), len(devices), devs, max_batch_per_shard, max_n)
        if rc != 0:
            raise PqpError(f"pqp error {rc}: {self.lib.pqp_last_error().decode()}")

    def close(self):
        if self._m:
            self.lib.pqp_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, option, value):
        if self.lib.pqp_multi_set_option(self._m, int(option), int(value)) != 0:
            raise PqpError(self.lib.pqp_last_error().decode())

    def solve(self, ref, bounds, scal, lin=None, passes=1, n_of=None):
        ref = np.ascontiguousarray(ref, dtype=np.float64); bounds = np.ascontiguousarray(bounds, dtype=np.float64)
        scal = np.ascontiguousarray(scal, dtype=np.float64)
        lin = None if lin is None else np.ascontiguousarray(lin, dtype=np.float64)
        counts = None if n_of is None else np.ascontiguousarray(n_of, dtype=np.int32)
        batch, n = ref.shape[0], ref.shape[1]
        out = np.zeros((batch, n, 7)); status = np.zeros(batch, dtype=np.int32); iters = np.zeros(batch, dtype=np.int32); info = np.zeros((batch, 8))
        rc = self.lib.pqp_multi_path_solve(self._m, batch, n, _ptr(counts), _ptr(ref), _ptr(lin), _ptr(bounds), _ptr(scal), passes, _ptr(out),
                                           _ptr(status), _ptr(iters), _ptr(info))
        if rc != 0:
            raise PqpError(f"pqp error {rc}: {self.lib.pqp_last_error().decode()}")
        self._last = (batch, n)
        return dict(out=out, status=status, iters=iters, info=info)

    def gather_paths(self, devices):
        """pqp_multi_gather_paths after solve(): one torch tensor [batch][n][7] per shard, on the shard's GPU (`devices`: the list given at creation),
        every one holding the whole batch - gathered over RCCL, nothing through the host."""
        import torch
        batch, n = self._last
        outs = [torch.zeros((batch, n, 7), dtype=torch.float64, device=torch.device("cuda", int(d))) for d in devices]
        for d in devices:
            torch.cuda.synchronize(int(d))
        ptrs = (C.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
        rc = self.lib.pqp_multi_gather_paths(self._m, batch, n, ptrs)
        if rc != 0:
            raise PqpError(f"pqp error {rc}: {self.lib.pqp_last_error().decode()}")
        return outs

    def gather_ranks(self):
        """ncclCommCount of the gather's communicator (0 before the first gather_paths)."""
        return int(self.lib.pqp_multi_gather_ranks(self._m))


class PqpError(RuntimeError):
    pass


class Handle:
    """Thin RAII wrapper over pqp_handle (one per GPU)."""

    def __init__(self, params=None, device=0, max_batch=1024, max_n=128):
        self.lib = load_library(with_torch=True)
        self.params = params or default_params(self.lib)
        self._h = C.c_void_p()
        self.device = int(device)          # the torch helpers below allocate and synchronise on the handle's own GPU
        self._check(self.lib.pqp_create(C.byref(self._h), C.byref(self.params), device, max_batch, max_n))

    def _check(self, rc):
        if rc != 0:
            raise PqpError(f"pqp error {rc}: {self.lib.pqp_last_error().decode()}")

    def close(self):
        if self._h:
            self.lib.pqp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, params):
        self.params = params
        self._check(self.lib.pqp_set_params(self._h, C.byref(params)))

    def sync(self):
        self._check(self.lib.pqp_sync(self._h))

    def set_option(self, option, value):
        """PQP_OPT_STORE_WARM = 1, PQP_OPT_ORDER_BY_COST = 2 (include/pqp.h)."""
        self._check(self.lib.pqp_set_option(self._h, int(option), int(value)))

    def mark(self, slot):
        self._check(self.lib.pqp_mark(self._h, slot))

    def constrain_angle(self, angles):
        """pqp_constrain_angle_device on a host array (torch as the memory plumbing)."""
        import torch
        dev = torch.device("cuda", self.device)
        a = torch.from_numpy(np.ascontiguousarray(angles, dtype=np.float64).ravel()).to(dev)
        o = torch.empty_like(a)
        torch.cuda.synchronize(dev)
        self._check(self.lib.pqp_constrain_angle_device(self._h, a.numel(), C.c_void_p(a.data_ptr()), C.c_void_p(o.data_ptr())))
        self.sync()
        return o.cpu().numpy().reshape(np.shape(angles))

    def wait_mark(self, other, slot):
        """Everything enqueued on this handle from now on waits for `other`'s mark `slot` (pqp_wait_mark)."""
        self._check(self.lib.pqp_wait_mark(self._h, other._h, slot))

    def wait_stream(self, hip_stream=None):
        """Order the handle's coming launches after what is enqueued on `hip_stream` now (None: the default stream; for torch tensors
        pass torch.cuda.current_stream().cuda_stream)."""
        self._check(self.lib.pqp_stream_wait(self._h, C.c_void_p(hip_stream or 0)))

    def stream(self):
        s = C.c_void_p()
        self._check(self.lib.pqp_get_stream(self._h, C.byref(s)))
        return s.value

    def chain_config(self, **over):
        c = PqpChainConfig()
        self.lib.pqp_chain_default_config(C.byref(c))
        for k, v in over.items():
            setattr(c, k, v)
        return c

    def optimize_path(self, points, n_points, start, target, dist, geom, map_of=None, smoother=None, cfg=None, start_k=None):
        """pqp_optimize_path_device with torch as the memory plumbing: host arrays in, device-resident chain, host arrays out.
        points [B][p_max][2], n_points [B], start / target [B][3], dist [n_maps][rows][cols] float32.  smoother: the handle the two
        smoother QPs run on (None: this one).  Returns dict(out [B][n_max][7], n_out, status, stage, iters)."""
        import torch
        dev = torch.device("cuda", self.device)
        cfg = cfg or self.chain_config()
        t = lambda a, dt: None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
        dist = np.asarray(dist, dtype=np.float32)
        if dist.ndim == 2:
            dist = dist[None]
        d_dist = t(np.transpose(dist, (0, 2, 1)), np.float32)              # the ABI's column-major layer
        B, p_max = points.shape[0], points.shape[1]
        d_pts, d_np, d_st, d_tg = t(points, np.float64), t(n_points, np.int32), t(start, np.float64), t(target, np.float64)
        d_map, d_k = t(map_of, np.int32), t(start_k, np.float64)
        out = torch.zeros((B, cfg.n_max, 7), dtype=torch.float64, device=dev)
        ints = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(4)]
        torch.cuda.synchronize(dev)
        p = lambda x: None if x is None else C.c_void_p(x.data_ptr())
        self._check(self.lib.pqp_optimize_path_device(self._h, smoother._h if smoother is not None else None, C.byref(cfg), B, p_max, p(d_pts), p(d_np),
                                                      p(d_st), p(d_tg), p(d_dist), p(d_map), C.byref(geom), p(d_k), p(out), p(ints[0]), p(ints[1]),
                                                      p(ints[2]), p(ints[3])))
        self.sync()
        if smoother is not None:
            smoother.sync()
        return dict(out=out.cpu().numpy(), n_out=ints[0].cpu().numpy(), status=ints[1].cpu().numpy(), stage=ints[2].cpu().numpy(), iters=ints[3].cpu().numpy())

    def corridor_params(self, **over):
        p = PqpCorridorParams()
        self.lib.pqp_corridor_default_params(C.byref(p))
        for k, v in over.items():
            setattr(p, k, v)
        return p

    def dp_corridor(self, spline, spline_ext, length, start, dist, geom, max_layers=128, map_of=None, prm=None):
        """pqp_dp_corridor (host arrays) -> (layers_s [B][max_layers], lb, ub, count [B], vehicle_l [B])."""
        spline = np.ascontiguousarray(spline, dtype=np.float64); spline_ext = np.ascontiguousarray(spline_ext, dtype=np.float64)
        length = np.ascontiguousarray(length, dtype=np.float64); start = np.ascontiguousarray(start, dtype=np.float64)
        dist = np.asarray(dist, dtype=np.float32)
        if dist.ndim == 2:
            dist = dist[None]
        dist_cm = np.ascontiguousarray(np.transpose(dist, (0, 2, 1)))
        B, m = spline.shape[0], spline.shape[2]
        ls = np.zeros((B, max_layers)); lb = np.zeros((B, max_layers)); ub = np.zeros((B, max_layers))
        count = np.zeros(B, dtype=np.int32); vl = np.zeros(B)
        mo = None if map_of is None else np.ascontiguousarray(map_of, dtype=np.int32)
        if prm is None:
            prm = PqpDpParams()
            self.lib.pqp_dp_default_params(C.byref(prm))
        self._check(self.lib.pqp_dp_corridor(self._h, B, m, max_layers, _ptr(spline), _ptr(spline_ext), _ptr(length), _ptr(start), _ptr(dist_cm),
                                             dist.shape[0], _ptr(mo), C.byref(geom), C.byref(prm), _ptr(ls), _ptr(lb), _ptr(ub), _ptr(count), _ptr(vl)))
        return ls, lb, ub, count, vl

    def spline_fit(self, s, x, y):
        """pqp_spline_fit (host arrays [B][m]) -> (spline [B][9][m], spline_ext [B][4])."""
        s = np.ascontiguousarray(s, dtype=np.float64); x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        B, m = s.shape
        tab = np.zeros((B, 9, m)); ext = np.zeros((B, 4))
        self._check(self.lib.pqp_spline_fit(self._h, B, m, _ptr(s), _ptr(x), _ptr(y), _ptr(tab), _ptr(ext)))
        return tab, ext

    def reference_states(self, spline, spline_ext, max_s, n_max, start=None, ds_small=0.15, ds_large=0.3, dynamic=True):
        """pqp_reference_states (host arrays): spline [B][9][m], spline_ext [B][4], max_s [B], start [B][3] or None.
        Returns (ref [B][n_max][5], count [B], init_err [B][2] or None)."""
        spline = np.ascontiguousarray(spline, dtype=np.float64)
        spline_ext = np.ascontiguousarray(spline_ext, dtype=np.float64)
        max_s = np.ascontiguousarray(max_s, dtype=np.float64)
        B, m = spline.shape[0], spline.shape[2]
        st = None if start is None else np.ascontiguousarray(start, dtype=np.float64)
        ref = np.zeros((B, n_max, 5)); count = np.zeros(B, dtype=np.int32)
        err = None if start is None else np.zeros((B, 2))
        self._check(self.lib.pqp_reference_states(self._h, B, n_max, m, _ptr(spline), _ptr(spline_ext), _ptr(max_s), _ptr(st), ds_small,
                                                  ds_large, 1 if dynamic else 0, _ptr(ref), _ptr(count), _ptr(err)))
        return ref, count, err

    def offsets_to_points(self, spline, spline_ext, at_s, l, m_of=None):
        """pqp_offsets_to_points (host arrays): spline [B][9][m_spline], at_s, l [B][m], m_of [B] or None -> (x, y, s) [B][m]."""
        spline = np.ascontiguousarray(spline, dtype=np.float64); spline_ext = np.ascontiguousarray(spline_ext, dtype=np.float64)
        at_s = np.ascontiguousarray(at_s, dtype=np.float64); l = np.ascontiguousarray(l, dtype=np.float64)
        mo = None if m_of is None else np.ascontiguousarray(m_of, dtype=np.int32)
        B, ms, m = spline.shape[0], spline.shape[2], at_s.shape[1]
        x = np.zeros((B, m)); y = np.zeros((B, m)); s = np.zeros((B, m))
        self._check(self.lib.pqp_offsets_to_points(self._h, B, ms, m, _ptr(spline), _ptr(spline_ext), _ptr(at_s), _ptr(l), _ptr(mo), _ptr(x), _ptr(y), _ptr(s)))
        return x, y, s

    def reference_length(self, spline, spline_ext, length, target):
        """pqp_reference_length (host arrays): spline [B][9][m], spline_ext [B][4], length [B], target [B][3] -> length_out [B]."""
        spline = np.ascontiguousarray(spline, dtype=np.float64); spline_ext = np.ascontiguousarray(spline_ext, dtype=np.float64)
        length = np.ascontiguousarray(length, dtype=np.float64); target = np.ascontiguousarray(target, dtype=np.float64)
        B, m = spline.shape[0], spline.shape[2]
        out = np.zeros(B)
        self._check(self.lib.pqp_reference_length(self._h, B, m, _ptr(spline), _ptr(spline_ext), _ptr(length), _ptr(target), _ptr(out)))
        return out

    def bspline_resample(self, points, n_points, n_max):
        """pqp_bspline_resample (host arrays): points [B][p_max][2], n_points [B].  Returns dict(x, y, s: [B][n_max], count [B])."""
        points = np.ascontiguousarray(points, dtype=np.float64)
        n_points = np.ascontiguousarray(n_points, dtype=np.int32)
        B, p_max = points.shape[0], points.shape[1]
        o = {k: np.zeros((B, n_max)) for k in ("x", "y", "s")}
        count = np.zeros(B, dtype=np.int32)
        self._check(self.lib.pqp_bspline_resample(self._h, B, p_max, n_max, _ptr(points), _ptr(n_points), _ptr(o["x"]), _ptr(o["y"]), _ptr(o["s"]),
                                                  _ptr(count)))
        o["count"] = count
        return o

    def segment_raw_reference(self, spline, spline_ext, max_s, n_max, delta_s=1.0):
        """pqp_segment_raw_reference (host arrays): spline [B][9][m], spline_ext [B][4], max_s [B].
        Returns dict(x, y, s, angle, k: [B][n_max], count [B]) - the smoother QPs' input lists."""
        spline = np.ascontiguousarray(spline, dtype=np.float64)
        spline_ext = np.ascontiguousarray(spline_ext, dtype=np.float64)
        max_s = np.ascontiguousarray(max_s, dtype=np.float64)
        B, m = spline.shape[0], spline.shape[2]
        o = {k: np.zeros((B, n_max)) for k in ("x", "y", "s", "angle", "k")}
        count = np.zeros(B, dtype=np.int32)
        self._check(self.lib.pqp_segment_raw_reference(self._h, B, n_max, m, _ptr(spline), _ptr(spline_ext), _ptr(max_s), delta_s,
                                                       _ptr(o["x"]), _ptr(o["y"]), _ptr(o["s"]), _ptr(o["angle"]), _ptr(o["k"]), _ptr(count)))
        o["count"] = count
        return o

    def corridor_bounds(self, ref, spline, spline_ext, dist, geom, map_of=None, prm=None, n_of=None):
        """pqp_corridor_bounds (host arrays): ref [B][n][5], spline [B][9][m], spline_ext [B][4], dist [n_maps][rows][cols] float32
        (converted to the ABI's column-major order here), geom = PqpGridGeometry.  Returns (bounds [B][n][6], n_valid [B])."""
        ref = np.ascontiguousarray(ref, dtype=np.float64)
        spline = np.ascontiguousarray(spline, dtype=np.float64)
        spline_ext = np.ascontiguousarray(spline_ext, dtype=np.float64)
        dist = np.asarray(dist, dtype=np.float32)
        if dist.ndim == 2:
            dist = dist[None]
        dist_cm = np.ascontiguousarray(np.transpose(dist, (0, 2, 1)))        # [n_maps][cols][rows]
        B, n = ref.shape[0], ref.shape[1]
        m = spline.shape[2]
        bounds = np.zeros((B, n, 6))
        n_valid = np.zeros(B, dtype=np.int32)
        mo = None if map_of is None else np.ascontiguousarray(map_of, dtype=np.int32)
        prm = prm or self.corridor_params()
        no = None if n_of is None else np.ascontiguousarray(n_of, dtype=np.int32)
        self._check(self.lib.pqp_corridor_bounds(self._h, B, n, m, _ptr(ref), _ptr(no), _ptr(spline), _ptr(spline_ext), _ptr(dist_cm), dist.shape[0],
                                                 _ptr(mo), C.byref(geom), C.byref(prm), _ptr(bounds), _ptr(n_valid)))
        return bounds, n_valid

    def sizes(self, n, s=None):
        out = PqpSizes()
        self._check(self.lib.pqp_path_sizes(C.byref(self.params), n, _ptr(s), C.byref(out)))
        return {k: getattr(out, k) for k, _ in PqpSizes._fields_}

    def pattern(self, n, precise=None):
        precise = n if precise is None else precise
        nv = 3 * n + n - 1 + precise + n
        nnz_a = 3 * n + 7 * (n - 1) + n + 6 * precise + 2 * (n - precise) + 2
        rows = np.zeros(nnz_a, dtype=np.int32)
        colptr = np.zeros(nv + 1, dtype=np.int32)
        pcols = np.zeros(n + n - 1 + precise + n, dtype=np.int32)
        self._check(self.lib.pqp_path_pattern(self._h, n, precise, _ptr(rows), _ptr(colptr), _ptr(pcols)))
        return rows, colptr, pcols

    def assemble(self, ref, lin, bounds, scal, precise=None):
        batch, n = ref.shape[0], ref.shape[1]
        precise = n if precise is None else precise
        nnz_a = 3 * n + 7 * (n - 1) + n + 6 * precise + 2 * (n - precise) + 2
        nnz_p = n + n - 1 + precise + n
        cons = 4 * n + precise + n + 2
        a_val = np.zeros((batch, nnz_a)); p_val = np.zeros((batch, nnz_p))
        lo = np.zeros((batch, cons)); up = np.zeros((batch, cons))
        self._check(self.lib.pqp_path_assemble(self._h, batch, n, precise, _ptr(ref), _ptr(lin), _ptr(bounds),
                                               _ptr(scal), _ptr(a_val), _ptr(p_val), _ptr(lo), _ptr(up)))
        return a_val, p_val, lo, up

    def solve(self, ref, bounds, scal, lin=None, passes=1, warm=False):
        """Host-array convenience: returns dict(out, status, iters, info)."""
        batch, n = ref.shape[0], ref.shape[1]
        out = np.zeros((batch, n, 7)); status = np.zeros(batch, dtype=np.int32)
        iters = np.zeros(batch, dtype=np.int32); info = np.zeros((batch, 8))
        self._check(self.lib.pqp_path_solve(self._h, batch, n, _ptr(ref), _ptr(lin), _ptr(bounds), _ptr(scal),
                                            passes, 1 if warm else 0, _ptr(out), _ptr(status), _ptr(iters), _ptr(info)))
        return dict(out=out, status=status, iters=iters, info=info)

    def solve_var(self, n_of, ref, bounds, scal, lin=None, passes=1, warm=False):
        """pqp_path_solve_var (host arrays): a waypoint count per QP, arrays of stride n_max = ref.shape[1]."""
        batch, n = ref.shape[0], ref.shape[1]
        counts = np.ascontiguousarray(n_of, dtype=np.int32)
        out = np.zeros((batch, n, 7)); status = np.zeros(batch, dtype=np.int32)
        iters = np.zeros(batch, dtype=np.int32); info = np.zeros((batch, 8))
        self._check(self.lib.pqp_path_solve_var(self._h, batch, n, _ptr(counts), _ptr(ref), _ptr(lin), _ptr(bounds), _ptr(scal),
                                                passes, 1 if warm else 0, _ptr(out), _ptr(status), _ptr(iters), _ptr(info)))
        return dict(out=out, status=status, iters=iters, info=info)

    def solve_device(self, batch, n, ref, bounds, scal, out, lin=None, passes=1, warm=False, status=None,
                     iters=None, info=None):
        """Device pointers (ints or objects with data_ptr()); asynchronous on the handle's stream."""
        def dp(x):
            if x is None:
                return None
            return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        self._check(self.lib.pqp_path_solve_device(self._h, batch, n, dp(ref), dp(lin), dp(bounds), dp(scal), passes,
                                                   1 if warm else 0, dp(out), dp(status), dp(iters), dp(info)))

    def solve_var_device(self, batch, n_max, n_of, ref, bounds, scal, out, lin=None, passes=1, warm=False, status=None, iters=None, info=None):
        """pqp_path_solve_var_device: like solve_device with a waypoint count per QP (n_of: device int32 [batch])."""
        def dp(x):
            if x is None:
                return None
            return C.c_void_p(x.data_ptr() if hasattr(x, "data_ptr") else int(x))
        self._check(self.lib.pqp_path_solve_var_device(self._h, batch, n_max, dp(n_of), dp(ref), dp(lin), dp(bounds), dp(scal), passes,
                                                       1 if warm else 0, dp(out), dp(status), dp(iters), dp(info)))

    def get_solution(self, batch, n, precise=None):
        precise = n if precise is None else precise
        nv = 3 * n + n - 1 + precise + n
        nc = 4 * n + precise + n + 2
        x = np.zeros((batch, nv)); y = np.zeros((batch, nc))
        self._check(self.lib.pqp_path_get_solution(self._h, batch, n, precise, _ptr(x), _ptr(y)))
        return x, y

    # ---- smoother QPs (host arrays [batch][n]) ----
    def smooth_tension2(self, x, y, angle, k, s):
        B, n = x.shape
        ox = np.zeros((B, n)); oy = np.zeros((B, n)); os_ = np.zeros((B, n)); st = np.zeros(B, dtype=np.int32); it = np.zeros(B, dtype=np.int32)
        c = np.ascontiguousarray
        self._check(self.lib.pqp_smooth_tension2(self._h, B, n, _ptr(c(x)), _ptr(c(y)), _ptr(c(angle)), _ptr(c(k)), _ptr(c(s)), _ptr(ox), _ptr(oy),
                                                 _ptr(os_), _ptr(st), _ptr(it)))
        return dict(x=ox, y=oy, s=os_, status=st, iters=it)

    def smooth_tension(self, x, y, angle, clearance, info=False):
        B, n = x.shape
        ox = np.zeros((B, n)); oy = np.zeros((B, n)); os_ = np.zeros((B, n)); st = np.zeros(B, dtype=np.int32); it = np.zeros(B, dtype=np.int32)
        c = np.ascontiguousarray
        if not info:
            self._check(self.lib.pqp_smooth_tension(self._h, B, n, _ptr(c(x)), _ptr(c(y)), _ptr(c(angle)), _ptr(c(clearance)), _ptr(ox), _ptr(oy),
                                                    _ptr(os_), _ptr(st), _ptr(it)))
            return dict(x=ox, y=oy, s=os_, status=st, iters=it)
        # with the info rows (factorisations in [5]): the device entry point, torch as the memory plumbing
        import torch
        dev = torch.device("cuda", self.device)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        d = [t(a) for a in (x, y, angle, clearance)]
        o = [torch.zeros((B, n), dtype=torch.float64, device=dev) for _ in range(3)]
        dst, dit = (torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2))
        dinf = torch.zeros((B, 8), dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        p = lambda a: C.c_void_p(a.data_ptr())
        self._check(self.lib.pqp_smooth_tension_device(self._h, B, n, p(d[0]), p(d[1]), p(d[2]), p(d[3]), p(o[0]), p(o[1]), p(o[2]), p(dst), p(dit), p(dinf)))
        self.sync()
        return dict(x=o[0].cpu().numpy(), y=o[1].cpu().numpy(), s=o[2].cpu().numpy(), status=dst.cpu().numpy(), iters=dit.cpu().numpy(), info=dinf.cpu().numpy())

    def smooth_tension2_var(self, x, y, angle, k, s, n_of):
        """pqp_smooth_tension2_var_device (torch as the memory plumbing): lists [B][n_max], n_of [B] points per scenario."""
        import torch
        dev = torch.device("cuda", self.device)
        B, n = x.shape
        t = lambda a, dt=np.float64: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
        d = [t(a) for a in (x, y, angle, k, s)]
        d_n = t(n_of, np.int32)
        o = [torch.zeros((B, n), dtype=torch.float64, device=dev) for _ in range(3)]
        st, it = (torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2))
        torch.cuda.synchronize(dev)
        p = lambda a: C.c_void_p(a.data_ptr())
        self._check(self.lib.pqp_smooth_tension2_var_device(self._h, B, n, p(d_n), p(d[0]), p(d[1]), p(d[2]), p(d[3]), p(d[4]), p(o[0]), p(o[1]), p(o[2]), p(st), p(it), None))
        self.sync()
        return dict(x=o[0].cpu().numpy(), y=o[1].cpu().numpy(), s=o[2].cpu().numpy(), status=st.cpu().numpy(), iters=it.cpu().numpy())

    def smooth_tension_var(self, x, y, angle, clearance, n_of):
        """pqp_smooth_tension_var_device (torch as the memory plumbing): lists [B][n_max], n_of [B] points per scenario."""
        import torch
        dev = torch.device("cuda", self.device)
        B, n = x.shape
        t = lambda a, dt=np.float64: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
        d = [t(a) for a in (x, y, angle, clearance)]
        d_n = t(n_of, np.int32)
        o = [torch.zeros((B, n), dtype=torch.float64, device=dev) for _ in range(3)]
        st, it = (torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2))
        torch.cuda.synchronize(dev)
        p = lambda a: C.c_void_p(a.data_ptr())
        self._check(self.lib.pqp_smooth_tension_var_device(self._h, B, n, p(d_n), p(d[0]), p(d[1]), p(d[2]), p(d[3]), p(o[0]), p(o[1]), p(o[2]), p(st), p(it), None))
        self.sync()
        return dict(x=o[0].cpu().numpy(), y=o[1].cpu().numpy(), s=o[2].cpu().numpy(), status=st.cpu().numpy(), iters=it.cpu().numpy())

    def post_smooth(self, layers_s, lb, ub, vehicle_l):
        B, m = layers_s.shape
        ol = np.zeros((B, m)); st = np.zeros(B, dtype=np.int32); it = np.zeros(B, dtype=np.int32)
        c = np.ascontiguousarray
        self._check(self.lib.pqp_post_smooth(self._h, B, m, _ptr(c(layers_s)), _ptr(c(lb)), _ptr(c(ub)), _ptr(c(vehicle_l)), _ptr(ol), _ptr(st), _ptr(it)))
        return dict(l=ol, status=st, iters=it)

    def post_smooth_var(self, layers_s, lb, ub, vehicle_l, m_of, info=False):
        """pqp_post_smooth_var_device (torch as the memory plumbing): lists [B][m_max], m_of [B] layers per scenario."""
        import torch
        dev = torch.device("cuda", self.device)
        B, m = layers_s.shape
        t = lambda a, dt=np.float64: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
        d = [t(a) for a in (layers_s, lb, ub, vehicle_l)]
        d_m = t(m_of, np.int32)
        ol = torch.zeros((B, m), dtype=torch.float64, device=dev)
        st, it = (torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2))
        inf = torch.zeros((B, 8), dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        p = lambda a: C.c_void_p(a.data_ptr())
        self._check(self.lib.pqp_post_smooth_var_device(self._h, B, m, p(d_m), p(d[0]), p(d[1]), p(d[2]), p(d[3]), p(ol), p(st), p(it), p(inf) if info else None))
        self.sync()
        r = dict(l=ol.cpu().numpy(), status=st.cpu().numpy(), iters=it.cpu().numpy())
        if info:
            r["info"] = inf.cpu().numpy()
        return r

    def kernel_ms_history(self, count):
        """HIP-event durations of the last `count` launches of this handle (oldest first)."""
        ms = np.zeros(count, dtype=np.float32)
        self._check(self.lib.pqp_kernel_ms_history(self._h, _ptr(ms), count))
        return ms

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(self.lib.pqp_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def last_path_kernel(self):
        """pqp_last_path_kernel: 1 = lane-per-waypoint kernel, 2 = lane-per-QP kernel served the last solve (0: none yet)."""
        return int(self.lib.pqp_last_path_kernel(self._h))
