"""What of the reference compiles from its own std-only sources (oracle/_ref/libref_types.so: constrainAngle of include/tools/tools.hpp:24-35,
the POD types of include/data_struct/data_struct.hpp:14-32,74-93, VehicleState of src/data_struct/vehicle_state_frenet.cpp) against the
restatements: constrain_angle of both oracles and of the HIP kernels bit for bit, include/pqp_types.hpp field by field."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TYPES = os.path.join(ROOT, "oracle", "_ref", "libref_types.so")
needs_ref = pytest.mark.skipif(not os.path.exists(REF_TYPES), reason="oracle/_ref not built (needs /root/reference: `make -C oracle ref`)")


def _ref():
    lib = C.CDLL(REF_TYPES)
    lib.ref_constrain_angle.argtypes = [C.c_double]
    lib.ref_constrain_angle.restype = C.c_double
    return lib


def _angles():
    rng = np.random.default_rng(7)
    pi = math.pi
    edge = [0.0, pi, -pi, np.nextafter(pi, 4.0), np.nextafter(-pi, -4.0), np.nextafter(pi, 0.0), 2 * pi, -2 * pi, 3 * pi, -3 * pi,
            np.nextafter(3 * pi, 10.0), 5 * pi - 1e-12, -7 * pi + 1e-12, 1e-300, -1e-300, 100.0, -100.0]
    return np.concatenate([np.array(edge), rng.uniform(-40.0, 40.0, 4000), rng.uniform(-pi, pi, 1000) + 2 * pi * rng.integers(-5, 6, 1000)])


@needs_ref
def test_constrain_angle_of_both_oracles_is_the_reference_template_bit_for_bit():
    import pqp_oracle as O
    import __graft_entry__ as g
    g.build_oracle()
    oc = C.CDLL(os.path.join(ROOT, "oracle", "libpqp_oracle.so"))
    oc.pqo_constrain_angle.argtypes = [C.c_double]
    oc.pqo_constrain_angle.restype = C.c_double
    ref = _ref()
    for a in _angles():
        want = ref.ref_constrain_angle(float(a))
        assert -math.pi <= want <= math.pi
        assert O.constrain_angle(float(a)) == want, a
        assert oc.pqo_constrain_angle(float(a)) == want, a


@needs_ref
def test_pqp_types_have_the_reference_layouts_and_semantics():
    ref = _ref()
    buf = (C.c_int * 64)()
    cnt = ref.ref_type_layout(buf, 64)
    want_layout = [int(buf[i]) for i in range(cnt)]
    a7 = (C.c_double * 7)(1.5, -2.5, 0.25, 0.125, 7.0, 3.0, -1.0)
    o8 = (C.c_double * 8)()
    ref.ref_state_ctor(a7, o8)
    o6 = (C.c_double * 6)()
    ref.ref_vehicle_state.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)] + [C.c_double] * 4 + [C.POINTER(C.c_double)]
    ref.ref_vehicle_state((C.c_double * 4)(0.5, 0.75, -0.5, 0.0625), (C.c_double * 4)(20.0, 3.0, 0.375, -0.03125), 0.4, -0.1, -0.2, 0.05, o6)
    exe = os.path.join(ROOT, "tests", "cpp", "types_layout")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wno-invalid-offsetof", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "types_layout.cpp")], check=True)
    lines = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines()
    got = {ln.split()[0]: ln.split()[1:] for ln in lines}
    assert [int(x) for x in got["layout"]] == want_layout                      # sizeof / offsetof of State, SlState, SingleBound, VehicleStateBound
    assert want_layout[0] == 64 and want_layout[9] == 80 and want_layout[18] == 120
    assert [float(x) for x in got["state"]] == list(o8)                         # State(x, y, heading, k, s, v, a): d_k stays 0
    assert [float(x) for x in got["vehicle"]] == list(o6)
    # SingleBound::set takes {ub, lb} in that order (data_struct.hpp:82-88): what the shim's bounds[n][6] = (lb, ub) per circle undoes
    o5 = (C.c_double * 5)()
    ref.ref_single_bound_set((C.c_double * 2)(1.75, -2.25), (C.c_double * 3)(4.0, 5.0, 0.5), o5)
    assert list(o5) == [1.75, -2.25, 4.0, 5.0, 0.5]


@needs_ref
@pytest.mark.gpu
def test_constrain_angle_of_the_hip_kernels_is_the_reference_template_bit_for_bit(hip_lib):
    from path_optimizer_2_amd import capi
    ref = _ref()
    a = _angles()
    h = capi.Handle(capi.default_params(), device=0, max_batch=1, max_n=8)
    got = h.constrain_angle(a)
    want = np.array([ref.ref_constrain_angle(float(x)) for x in a])
    h.close()
    assert np.array_equal(got, want)
