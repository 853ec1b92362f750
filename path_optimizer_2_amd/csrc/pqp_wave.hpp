// pqp_wave.hpp - wavefront-level device helpers shared by the translation units of libpqp_hip.so (gfx950, wave64): wave-uniform values,
// DPP reductions.  Device code only.
#pragma once
#include <hip/hip_runtime.h>

namespace pqp {

// A value that is the same in every lane, told to the compiler: what is derived from it - the control state of PathQp::run - then
// branches with s_cbranch instead of exec-mask bookkeeping (v_cndmask per state variable per branch; +4.4 %: profiles/r02l_uniform_control.txt).
__device__ __forceinline__ double uniform(double x) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
__device__ __forceinline__ bool uniform(bool x) { return __builtin_amdgcn_readfirstlane((int)x) != 0; }

// wave / workgroup reductions shared by the hot and the cold context
// One step of a wavefront max-reduction in the VALU (DPP: data-parallel primitives move a value between lanes inside the instruction,
// no LDS round trip as with __shfl): x = max(x, x of the lane CTRL selects); lanes without a source keep their value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max_step(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return fmax(x, __hiloint2double(ohi, olo));
}
// max over the 64 lanes of a wavefront, the same value in every lane (and known to the compiler as wave-uniform)
__device__ __forceinline__ double wave_max(double x) {
    x = dpp_max_step<0x111, 0xf>(x);      // row_shr:1
    x = dpp_max_step<0x112, 0xf>(x);      // row_shr:2
    x = dpp_max_step<0x114, 0xf>(x);      // row_shr:4
    x = dpp_max_step<0x118, 0xf>(x);      // row_shr:8      -> lane 15 of every row of 16: the row's max
    x = dpp_max_step<0x142, 0xa>(x);      // row_bcast:15   -> lanes 31, 63: max of rows 0-1, 2-3
    x = dpp_max_step<0x143, 0xc>(x);      // row_bcast:31   -> lane 63: max of the wavefront
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}

// the same for a sum (lanes without a source add 0)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_sum_step(double x) {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const int olo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    const int ohi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return x + __hiloint2double(ohi, olo);
}
__device__ __forceinline__ double wave_sum(double x) {
    x = dpp_sum_step<0x111, 0xf>(x);
    x = dpp_sum_step<0x112, 0xf>(x);
    x = dpp_sum_step<0x114, 0xf>(x);
    x = dpp_sum_step<0x118, 0xf>(x);
    x = dpp_sum_step<0x142, 0xa>(x);
    x = dpp_sum_step<0x143, 0xc>(x);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}

// Up to this many wavefronts per QP the polish save area (and the parked Ruiz vectors) live in LDS; beyond, in the workgroup slot's global
// memory.  4 is what fits (256 lanes: 157 KB).
constexpr int kSaveLdsMaxNw = 4;

}  // namespace pqp
