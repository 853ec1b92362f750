"""Build + drive tests/emu/liblane_emu.so: the device algorithm source compiled for the host (test infrastructure)."""
import ctypes as C
import os
import subprocess

import numpy as np

from path_optimizer_2_amd.capi import PqpParams

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "emu", "lane_emu.cpp")
# PQP_EMU_DIET=1: the emulation of the register-diet contexts (pass constants in the shared-memory array, Ruiz vectors parked)
DIET = os.environ.get("PQP_EMU_DIET", "0") == "1"
LIB = os.path.join(HERE, "emu", "liblane_emu_diet.so" if DIET else "liblane_emu.so")
# PQP_SANITIZED_LIBS=<dir>: load a prebuilt (-fsanitize=address,undefined) library from there instead (tools/sanitize_cpu.sh)
if os.environ.get("PQP_SANITIZED_LIBS"):
    LIB = os.path.join(os.environ["PQP_SANITIZED_LIBS"], "liblane_emu.so")
_DEPS = [SRC, os.path.join(ROOT, "path_optimizer_2_amd", "csrc", "pqp_path_lane.hpp"), os.path.join(ROOT, "path_optimizer_2_amd", "csrc", "pqp_banded_qp.hpp"),
         os.path.join(ROOT, "path_optimizer_2_amd", "csrc", "pqp_defaults.hpp"), os.path.join(ROOT, "include", "pqp.h")]
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.environ.get("PQP_SANITIZED_LIBS") and (not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in _DEPS)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", f"-DPQP_EMU_DIET={1 if DIET else 0}", "-o", LIB, SRC], check=True)
    _lib = C.CDLL(LIB)
    return _lib


def params(**over):
    p = PqpParams()
    load().pqp_emu_default_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def production(**over):
    p = PqpParams()
    load().pqp_emu_production_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def solve(prm, ref, bounds, scal, lin=None, passes=1, n_of=None):
    lib = load()
    B, n = ref.shape[0], ref.shape[1]
    vp = lambda a: None if a is None else np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    out = np.zeros((B, n, 7)); st = np.zeros(B, dtype=np.int32); it = np.zeros(B, dtype=np.int32); info = np.zeros((B, 8))
    wx = np.zeros((B, n, 6)); wy = np.zeros((B, n, 6)); wye = np.zeros((B, 2)); wrho = np.zeros(B)
    ref = np.ascontiguousarray(ref); bounds = np.ascontiguousarray(bounds); scal = np.ascontiguousarray(scal)
    lin_c = None if lin is None else np.ascontiguousarray(lin)
    n_of_c = None if n_of is None else np.ascontiguousarray(n_of, dtype=np.int32)
    lib.pqp_emu_set_counts(None if n_of_c is None else n_of_c.ctypes.data_as(C.c_void_p))
    lib.pqp_emu_path_solve(C.byref(prm), B, n, vp(ref), vp(lin_c), vp(bounds), vp(scal), passes, 0, vp(out), vp(st), vp(it),
                           vp(info), vp(wx), vp(wy), vp(wye), vp(wrho))
    lib.pqp_emu_set_counts(None)
    return dict(out=out, status=st, iters=it, info=info, wx=wx, wy=wy, wye=wye)


def solve_carrying_tails(prm, ref, bounds, scal, prev, prev_bins, threshold_bin, k=8):
    """The next planning cycle of a handle with PQP_OPT_CARRY_CYCLES = k >= 2, emulated: `prev` = the previous solve()'s result (its final iterates are the
    warm state), prev_bins[q] = the QPs' cost bins in that launch, threshold_bin = the bin from which on a QP is one of the expensive ones.  Returns
    the result and the cost keys the run left."""
    lib = load()
    B, n = ref.shape[0], ref.shape[1]
    vp = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    out = np.zeros((B, n, 7)); st = np.zeros(B, dtype=np.int32); it = np.zeros(B, dtype=np.int32); info = np.zeros((B, 8))
    wx, wy, wye = prev["wx"].copy(), prev["wy"].copy(), prev["wye"].copy()
    wrho = np.full(B, 0.1)
    keys = (np.asarray(prev_bins, dtype=np.int64) << 24).astype(np.uint32).view(np.int32).copy()       # as the kernel stores them: bin << 24 in an int32
    hist = np.zeros(258, dtype=np.int32); hist[257] = threshold_bin
    lib.pqp_emu_set_carry(keys.ctypes.data_as(C.c_void_p), hist.ctypes.data_as(C.c_void_p), k)
    lib.pqp_emu_path_solve(C.byref(prm), B, n, vp(ref), None, vp(bounds), vp(scal), 1, 1, vp(out), vp(st), vp(it), vp(info), vp(wx), vp(wy), vp(wye), vp(wrho))
    lib.pqp_emu_set_carry(None, None, 0)
    return dict(out=out, status=st, iters=it, info=info), keys


def to_reference_order(wx, wy, wye, n, precise=None):
    """Lane layout of one QP ([n][6] primal, [n][6] dual, [2] end-row duals) -> the reference numbering."""
    precise = n if precise is None else precise
    x = np.zeros(3 * n + n - 1 + precise + n)
    y = np.zeros(4 * n + precise + n + 2)
    for i in range(n):
        x[3 * i:3 * i + 3] = wx[i, :3]
        if i > 0:
            x[3 * n + i - 1] = wx[i, 3]
        y[3 * i:3 * i + 3] = wy[i, :3]
        y[3 * n + i] = wy[i, 3]
        if i < precise:
            x[4 * n - 1 + 2 * i] = wx[i, 4]; x[4 * n - 1 + 2 * i + 1] = wx[i, 5]
            y[4 * n + 2 * i] = wy[i, 4]; y[4 * n + 2 * i + 1] = wy[i, 5]
        else:
            x[4 * n - 1 + 2 * precise + (i - precise)] = wx[i, 4]
            y[4 * n + 2 * precise + (i - precise)] = wy[i, 4]
    y[-2:] = wye
    return x, y
