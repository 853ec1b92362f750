// pqp_path_lq_abi.hpp - what the host side (pqp_kernels.hip) and the kernel (pqp_path_stream.hip, pqp_path_lq.hpp) of the lane-per-QP path
// solver share: the kernel's argument block and the layout of its workspace.
#pragma once
#include <stdint.h>

#include "../../include/pqp.h"

namespace pqp {
namespace lq {

// Workspace: per waypoint and lane kFieldsD doubles followed by kFieldsF floats, in one of two layouts chosen per launch (Args::staged):
//   [field][lane] (lq::StridedWs): element (waypoint i, field f) of lane j of a wavefront's block sits at block[(i * kBlockDoubles + f) * 64 + j] (doubles) resp.
//     ((float*)(block + (i * kBlockDoubles + kFieldsD) * 64))[f * 64 + j] - every load / store instruction of a wavefront is one contiguous 512 (256) bytes;
//   [chunk][lane][16 bytes] (lq::ChunkWs): the fields in 16-byte CHUNKS (two doubles / four floats) that stay together - chunk c of (waypoint i, lane j) sits at
//     block[((i * kBlockChunks + c) * 64 + j) * 2 .. + 1], double field f is part f & 1 of chunk f >> 1, float field f is float f & 3 of chunk
//     kFieldsD / 2 + (f >> 2).  A lane's share of a chunk is what an LDS-direct load (global_load_lds_dwordx4) copies per lane, whatever waypoint each lane is at.
// fp64: everything an ACTIVE-SET round reads or writes (problem data, gains, the point) - those rounds return the result.  fp32: what only
// the interior-point rounds exchange between their sweeps (slacks, multipliers, row steps): they only have to predict the active set.
enum FieldD {
    D_M00 = 0, D_M01, D_M10, D_M11, D_M12, D_C0, D_C1, D_DS,      // transition i -> i + 1 (i < n - 1)
    D_LOF, D_UPF, D_LOR, D_UPR,                                    // soft boxes of the collision rows (rear off: up = +inf)
    D_K0, D_K1, D_K2, D_KK,                                        // feedback law u_i = -K x_i - k of the last backward sweep
    D_X0, D_X1, D_X2,                                              // the point of the last active-set round
    D_GK,                                                          // interior-point rounds: value of the kappa row (its residual needs all digits)
    D_ACT, D_LAM,                                                  // active-set rounds: the three rows' states packed as f + 3 r + 9 k + 13; multiplier of the kappa row
    kFieldsD
};
enum FieldF {
    S_TLF = 0, S_TUF, S_ZLF, S_ZUF, S_TLR, S_TUR, S_ZLR, S_ZUR, S_TLK, S_TUK, S_ZLK, S_ZUK,     // slacks and multipliers of the three rows: one 16-byte chunk per row
    S_DGF, S_DGR, S_DGK,                                           // row steps of the last interior-point roll-out
    S_PAD,      // (the gains stay fp64 in every round: an fp32 gain times a state of order 1 is 1e-8 of noise in a row value, more than the slack
                // of a tightly active row near the end of the interior-point rounds - the steps then shrink to nothing)
    kFieldsF
};
constexpr int kBlockDoubles = kFieldsD + kFieldsF / 2;            // 30 doubles = 240 bytes per waypoint and QP
constexpr int kBlockChunks = kBlockDoubles / 2;                   // 15 chunks of 16 bytes
static_assert(kFieldsD % 2 == 0 && kFieldsF % 4 == 0, "the fields fill whole 16-byte chunks");

// phase key of a QP: interior-point iterations of the first pass (5 bits), active-set rounds of the first pass (3), iterations (4) and rounds (3) of the
// re-linearised pass - what a wavefront runs in lock-step, most significant first
constexpr int kOrderBins = 1 << 15;

struct Args {
    int batch, n, passes;
    const int32_t* n_of;        // [batch] or nullptr
    const double* ref;          // [batch][n][5]
    const double* lin;          // [batch][n][3] or nullptr
    const double* bounds;       // [batch][n][6]
    const double* scal;         // [batch][6]
    double* out;                // [batch][n][7]
    int32_t* status;            // [batch] or nullptr
    int32_t* iters;             // [batch] or nullptr: interior-point iterations over all passes
    double* info;               // [batch][PQP_INFO_STRIDE] or nullptr
    double* ws;                 // [ceil(batch / 64)][n][kBlockDoubles][64]
    // PQP_OPT_ORDER_BY_COST on this kernel (batches that fill the chip): slot -> QP map of THIS launch (wavefronts of QPs that ran the same interior-point
    // iterations / active-set rounds per pass in the previous solve: the 64 lanes of a wavefront run the maxima of their phases in lock-step) or nullptr;
    // what the launch records for the next one: every QP's phase key, the key histogram (+ [kOrderBins]: wavefronts finished), and the next map -
    // written by the last wavefront to finish (pqp_path_stream.hip)
    const int32_t* order;       // [batch] or nullptr.  The solver also asks WHETHER the launch is sorted: its re-linearised passes then begin with active-set rounds on the
                                // previous pass's set (lq::kDirectRounds, pqp_path_lq.hpp) - the phase key's third field keeps the QPs that fall back in wavefronts of their own
    int32_t* key_out;           // [batch] or nullptr
    int32_t* hist;              // [kOrderBins + 1]
    int32_t* order_next;        // [batch]
    int staged;                 // 1: the [chunk][lane] layout, the sweeps' records staged in LDS two waypoints ahead (launches that leave SIMDs idle: pqp_kernels.hip)
    int carry;                  // 1: the workspace still holds what this very launch shape left there last time (PQP_OPT_CARRY_CYCLES): a QP's first pass
                                // starts its interior-point rounds from its slot's previous optimum - the same scenario one planning cycle earlier
    pqp_params prm;
};

}  // namespace lq
}  // namespace pqp
