// HBM layout of the horizon-stacked projected LQ stages handed from lq_node_kernel to riccati_kernel.
//
// One record per (instance, node), fp64, every matrix row-major with the *lane index as the fast dimension*, so a
// wavefront whose lane c owns column c writes/reads whole rows as contiguous 240-byte runs (coalesced).
//   m~ = 30 - nc projected inputs (14..18 for this robot), padded to MT = 18 columns.
// Algorithmic bytes per node (what DESIGN.md's roofline uses): the record is written once and read once.
#pragma once

namespace qmk {

constexpr int NX = 30, NU = 30, MT = 18, NCMAX = 16;
constexpr int OFF_AT = 0;                    // rows 0..11: A~ [12][30] (dense rows of the RK2 map); rows 12..29: Px rows 12..29 (see below)
constexpr int OFF_BT = OFF_AT + 900;         // rows 0..11: B~ [12][MT];                               rows 12..29: Pu rows 12..29
constexpr int OFF_QT = OFF_BT + 30 * MT;     // Q~   [30][30]
constexpr int OFF_PT = OFF_QT + 900;         // P~   [MT][30]
constexpr int OFF_RT = OFF_PT + MT * 30;     // R~   [MT][MT]
constexpr int OFF_bt = OFF_RT + MT * MT;     // b~   [30]
constexpr int OFF_qt = OFF_bt + 30;          // q~   [30]
constexpr int OFF_rt = OFF_qt + 30;          // r~   [MT]
constexpr int OFF_DTPREV = OFF_rt + MT;      // step of node k - 1 and of node k + 1: what the consumer needs to form the joint rows of the record it stages NEXT
constexpr int OFF_DTNEXT = OFF_DTPREV + 1;   // (backward / forward order) arrives with the record it is processing -- no load of its own on the sweep
constexpr int OFF_TAIL = OFF_rt + MT + 2;    // end of what the backward sweep reads (3284)
// The joint rows (12..29) of the projected dynamics are not data of their own: x_j+ = x_j + dt v_j exactly, so
//   A~[i][:] = e_i + dt Px[i][:],   B~[i][:] = dt Pu[i][:]      (i >= 12; du = Pe + Px dx + Pu du~)
// and the record holds Px / Pu THERE, once; riccati_kernel forms the A~ / B~ rows while the staged copy lands in LDS (jointRowsToDynamics).
// Rows 0..11 (force inputs) of Px are structurally zero and rows 0..11 of Pu are unit vectors on the free stance forces in foot order (zero rows for a swing
// foot, whose force Pe pins): neither exists in the record; the consumers form them from the node's contact mode, which rides in the record's tail.
// 6.9 + 1.7 KB per stage less written by lq_node_kernel and less read by the forward sweep than with separate A~ B~ / Px Pu copies (round 3).
constexpr int OFF_PE = OFF_TAIL;             // Pe   [30]
constexpr int OFF_MODE = OFF_PE + 30;        // contact mode of the node (as a real)
constexpr int OFF_DT = OFF_MODE + 1;         // step of the node itself (the forward sweep multiplies its joint rows by it)
constexpr int STAGE_DOUBLES = OFF_MODE + 2 + 4;  // 3320, multiple of 8
// offsets of row i >= 12 of Px and of Pu
constexpr int offPxRow(int i) { return OFF_AT + i * 30; }
constexpr int offPuRow(int i) { return OFF_BT + i * MT; }
// column of Pu that carries the unit entry of force input i < 12, or -1 (swing foot): stance feet in foot order, three columns each (lq_node_kernel: puColOf)
constexpr int puColumnOfForce(int mode, int i) {
  int nb = 0, col = -1;
  for (int leg = 0; leg < 4; ++leg) { const int st = (mode >> (3 - leg)) & 1; if (st && i / 3 == leg) col = 3 * nb + i % 3; nb += st; }
  return col;
}
static_assert(STAGE_DOUBLES % 8 == 0, "stage records stay 64-byte aligned");

// feedback record written by the backward sweep for the forward sweep: K [MT][30], k [MT]
constexpr int OFF_KFB = 0, OFF_kff = MT * 30;
constexpr int GAIN_DOUBLES = MT * 30 + MT + 2;  // 560

// the input-weight buffer R' [30][30] is followed by constants of the relaxed log barriers that input_weight_kernel derives once per settings update
// (every node evaluation used to recompute them: two logarithms per barrier value, half of the four values per joint limit):
//   log(delta) of the joint-position, joint-velocity and friction-cone barrier; value(-lower) + value(upper) per arm joint for positions and velocities
constexpr int QM_RW_DERIVED = 900, QM_BC_LOGD_POS = 0, QM_BC_LOGD_VEL = 1, QM_BC_LOGD_FRIC = 2, QM_BC_POS0 = 4, QM_BC_VEL0 = 10, QM_RW_DOUBLES = 916;

// per-node metrics: dt*cost, dt*|defect|^2, dt*|eq|^2, armijo contribution (filled by the forward sweep)
constexpr int NODE_METRICS = 4;

// optional debug dump of the un-projected LQ blocks (qmgpu_debug_get_lq): A B b Q R q r C D e
constexpr int DBG_A = 0, DBG_B = 900, DBG_b = 1800, DBG_Q = 1830, DBG_R = 2730, DBG_q = 3630, DBG_r = 3660, DBG_C = 3690, DBG_D = DBG_C + 16 * 30,
              DBG_e = DBG_D + 16 * 30, DBG_DOUBLES = DBG_e + 16 + 2;  // 4668

}  // namespace qmk
