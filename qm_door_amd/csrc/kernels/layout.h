// HBM layout of the horizon-stacked projected LQ stages handed from lq_node_kernel to riccati_kernel.
//
// One record per (instance, node), fp64, every matrix row-major with the *lane index as the fast dimension*, so a
// wavefront whose lane c owns column c writes/reads whole rows as contiguous 240-byte runs (coalesced).
//   m~ = 30 - nc projected inputs (14..18 for this robot), padded to MT = 18 columns.
// Algorithmic bytes per node (what DESIGN.md's roofline uses): the record is written once and read once.
#pragma once

namespace qmk {

constexpr int NX = 30, NU = 30, MT = 18, NCMAX = 16;
constexpr int OFF_AT = 0;                    // A~   [30][30]
constexpr int OFF_BT = OFF_AT + 900;         // B~   [30][MT]
constexpr int OFF_QT = OFF_BT + 30 * MT;     // Q~   [30][30]
constexpr int OFF_PT = OFF_QT + 900;         // P~   [MT][30]
constexpr int OFF_RT = OFF_PT + MT * 30;     // R~   [MT][MT]
constexpr int OFF_bt = OFF_RT + MT * MT;     // b~   [30]
constexpr int OFF_qt = OFF_bt + 30;          // q~   [30]
constexpr int OFF_rt = OFF_qt + 30;          // r~   [MT]
constexpr int OFF_PX = OFF_rt + MT + 2;      // Px   [30][30]   (du = Pe + Px dx + Pu du~); rows 0..11 (force inputs) are structurally zero: NOT written by
                                             //      lq_node_kernel and NOT read by the forward sweep / the DDP rollout (2.9 KB per stage each way)
constexpr int OFF_PU = OFF_PX + 900;         // Pu   [30][MT]
constexpr int OFF_PE = OFF_PU + 30 * MT;     // Pe   [30]
constexpr int STAGE_DOUBLES = OFF_PE + 30 + 6;  // 4760, multiple of 8
static_assert(STAGE_DOUBLES % 8 == 0, "stage records stay 64-byte aligned");

// feedback record written by the backward sweep for the forward sweep: K [MT][30], k [MT]
constexpr int OFF_KFB = 0, OFF_kff = MT * 30;
constexpr int GAIN_DOUBLES = MT * 30 + MT + 2;  // 560

// per-node metrics: dt*cost, dt*|defect|^2, dt*|eq|^2, armijo contribution (filled by the forward sweep)
constexpr int NODE_METRICS = 4;

// optional debug dump of the un-projected LQ blocks (qmgpu_debug_get_lq): A B b Q R q r C D e
constexpr int DBG_A = 0, DBG_B = 900, DBG_b = 1800, DBG_Q = 1830, DBG_R = 2730, DBG_q = 3630, DBG_r = 3660, DBG_C = 3690, DBG_D = DBG_C + 16 * 30,
              DBG_e = DBG_D + 16 * 30, DBG_DOUBLES = DBG_e + 16 + 2;  // 4668

}  // namespace qmk
