// frontend_kernel -- what sits immediately before the MPC call in a control cycle, one thread per robot instance:
//   rbdState[55] -> centroidal state x[30] (QMController::updateStateEstimation, qm_controllers/src/QMController.cpp:239-244) and
//   command -> two-knot TargetTrajectories (qm_controllers/src/QmTargetTrajectoriesPublisher_node.cpp:59-254).
// The normalised centroidal momentum comes from the same streaming tree sweep as the flow map, run in the opposite direction:
// the measured base twist is given, h is the unknown (closeSweep solves for the twist given h).
#pragma once
#include "../../../include/qmgpu.h"
#include "model_dev.h"

namespace qmk {

struct FrontendArgs {
  const qmgpu_problem* P;
  qmgpu_frontend_args a;
};

__device__ inline void centroidalStateFromRbd(const qmgpu_model& md, const double* rbd, double* x) {
  // Pinocchio coordinates: q = [p(3), zyx(3), qj(18)]; base twist in world axes straight from the estimator (rbd[24:30])
  const double yaw = rbd[0], pitch = rbd[1], roll = rbd[2];
  ChainState<double> base;
  double sz, cz, sy, cy;
  baseRotation(yaw, pitch, roll, base.R, sz, cz, sy, cy);
  Accum<double> acc;
  accumulateBody(md, 0, base, acc);
  {
    ChainState<double> s = base;
    for (int a = 0; a < 6; ++a) bodyStep(md, 13 + a, rbd[6 + 12 + a], rbd[30 + 12 + a], s, acc);
  }
  for (int leg = 0; leg < 4; ++leg) {
    ChainState<double> s = base;
    for (int j = 0; j < 3; ++j) bodyStep(md, 1 + 3 * leg + j, rbd[6 + 3 * leg + j], rbd[30 + 3 * leg + j], s, acc);
  }
  const double m = md.total_mass, im = 1.0 / m;
  const Vec3<double> cm = scale(im, acc.M1);
  const double cc = dot(cm, cm);
  Sym3<double> Ic;
  Ic.xx = acc.Io.xx - m * (cc - cm.x * cm.x); Ic.yy = acc.Io.yy - m * (cc - cm.y * cm.y); Ic.zz = acc.Io.zz - m * (cc - cm.z * cm.z);
  Ic.xy = acc.Io.xy + m * (cm.x * cm.y); Ic.xz = acc.Io.xz + m * (cm.x * cm.z); Ic.yz = acc.Io.yz + m * (cm.y * cm.z);
  const Vec3<double> om(rbd[24], rbd[25], rbd[26]), dp(rbd[27], rbd[28], rbd[29]);
  const Vec3<double> haC = acc.ha - cross(cm, acc.hl);
  const Vec3<double> hl = acc.hl + scale(m, dp + cross(om, cm));
  const Vec3<double> ha = haC + mul(Ic, om);
  x[0] = im * hl.x; x[1] = im * hl.y; x[2] = im * hl.z; x[3] = im * ha.x; x[4] = im * ha.y; x[5] = im * ha.z;
  x[6] = rbd[3]; x[7] = rbd[4]; x[8] = rbd[5]; x[9] = yaw; x[10] = pitch; x[11] = roll;
  for (int j = 0; j < 18; ++j) x[12 + j] = rbd[6 + j];
}

__device__ inline void quatRotate(const double* q /*xyzw*/, const double* v, double* o) {
  // R(q) v with R from the unit quaternion (Eigen::Quaterniond::toRotationMatrix)
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double r00 = 1 - 2 * (y * y + z * z), r01 = 2 * (x * y - w * z), r02 = 2 * (x * z + w * y);
  const double r10 = 2 * (x * y + w * z), r11 = 1 - 2 * (x * x + z * z), r12 = 2 * (y * z - w * x);
  const double r20 = 2 * (x * z - w * y), r21 = 2 * (y * z + w * x), r22 = 1 - 2 * (x * x + y * y);
  o[0] = r00 * v[0] + r01 * v[1] + r02 * v[2]; o[1] = r10 * v[0] + r11 * v[1] + r12 * v[2]; o[2] = r20 * v[0] + r21 * v[1] + r22 * v[2];
}

__global__ void __launch_bounds__(64) frontend_kernel(FrontendArgs fa) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const qmgpu_frontend_args& a = fa.a;
  if (i >= a.batch) return;
  const qmgpu_model& md = fa.P->model;
  const qmgpu_settings& st = fa.P->settings;
  const double* rbd = a.rbd_measured + size_t(i) * 55;
  double x[30];
  centroidalStateFromRbd(md, rbd, x);
  if (a.yaw_last) {  // angles::shortest_angular_distance (QMController.cpp:243)
    const double yl = a.yaw_last[i];
    double dyaw = fmod(x[9] - yl + 3.14159265358979323846, 6.28318530717958647692);
    if (dyaw < 0.0) dyaw += 6.28318530717958647692;
    x[9] = yl + (dyaw - 3.14159265358979323846);
  }
  double* xo = a.x0 + size_t(i) * 30;
  for (int k = 0; k < 30; ++k) xo[k] = x[k];

  const double t = a.time[i];
  const double T = st.time_horizon;                       // TIME_TO_TARGET = mpc.timeHorizon (:275)
  const double zRef = st.com_height + (a.feet_height ? a.feet_height[i] : 0.0);
  const int kind = a.command_kind[i];
  const double* cmd = a.command + size_t(i) * 7;
  double* lastEe = a.last_ee_target + size_t(i) * 7;
  const double* eeCur = rbd + 48;                          // EE position + quaternion xyzw (StateEstimateBase.cpp:101-102)
  const double* baseCur = x + 6;                           // observation.state.segment<6>(6)
  double s0[37], s1[37], tReach = t + T;
  for (int k = 0; k < 37; ++k) { s0[k] = 0.0; s1[k] = 0.0; }
  for (int j = 0; j < 18; ++j) { s0[12 + j] = st.default_joint_state[j]; s1[12 + j] = st.default_joint_state[j]; }
  // first knot of targetPoseToTargetTrajectories (:69-76): current base pose with z = comHeight + feet, pitch = roll = 0
  s0[6] = baseCur[0]; s0[7] = baseCur[1]; s0[8] = zRef; s0[9] = baseCur[3]; s0[10] = 0.0; s0[11] = 0.0;
  if (kind == 1) {
    // cmdVelToTargetTrajectories (:89-129)
    ChainState<double> b; double sz, cz, sy, cy;
    baseRotation(baseCur[3], baseCur[4], baseCur[5], b.R, sz, cz, sy, cy);
    const Vec3<double> vr = mul(b.R, cmd[0], cmd[1], cmd[2]);
    s1[6] = baseCur[0] + vr.x * T; s1[7] = baseCur[1] + vr.y * T; s1[8] = zRef; s1[9] = baseCur[3] + cmd[3] * T; s1[10] = 0.0; s1[11] = 0.0;
    const double dx = lastEe[0] - eeCur[0], dy = lastEe[1] - eeCur[1], dz = lastEe[2] - eeCur[2];
    if (sqrt(dx * dx + dy * dy + dz * dz) > 0.1) { lastEe[0] = eeCur[0]; lastEe[1] = eeCur[1]; lastEe[2] = eeCur[2]; }
    for (int k = 0; k < 7; ++k) { s0[30 + k] = lastEe[k]; s1[30 + k] = lastEe[k]; }   // eeStateLast.state = EeTargetPose (:119-120)
    s0[0] = vr.x; s0[1] = vr.y; s0[2] = vr.z; s1[0] = vr.x; s1[1] = vr.y; s1[2] = vr.z;   // (:125-126)
  } else if (kind == 2) {
    // EeCmdVelToTargetTrajectories (:134-188)
    const double qi[4] = {0.0, 0.0, -sin(baseCur[3] / 2), cos(baseCur[3] / 2)};   // quat_init^T (inverse of the yaw-only rotation)
    double tmp[3], vr[3];
    quatRotate(qi, cmd, tmp);
    quatRotate(eeCur + 3, tmp, vr);
    double ee[7];
    for (int k = 0; k < 7; ++k) ee[k] = eeCur[k];
    ee[0] = eeCur[0] + vr[0] * T; ee[1] = eeCur[1] + vr[1] * T; ee[2] = lastEe[2]; ee[3] = lastEe[3]; ee[4] = lastEe[4];
    ee[5] = eeCur[5] + sin(vr[2] * T / 2); ee[6] = eeCur[6] + cos(vr[2] * T / 2);
    const double yaw = atan2(2.0 * (ee[6] * ee[5] + ee[3] * ee[4]), 1.0 - 2.0 * (ee[4] * ee[4] + ee[5] * ee[5]));
    s1[6] = ee[0] - a.arm_dist * cos(baseCur[3]); s1[7] = ee[1] - a.arm_dist * sin(baseCur[3]); s1[8] = zRef; s1[9] = yaw; s1[10] = 0.0; s1[11] = 0.0;
    for (int k = 0; k < 7; ++k) { s0[30 + k] = eeCur[k]; s1[30 + k] = ee[k]; }
  } else if (kind == 3) {
    // EEgoalPoseToTargetTrajectories (:195-238); the callback then stores the goal as lastEeTarget_ (:253)
    const double yaw = atan2(2.0 * (cmd[6] * cmd[5] + cmd[3] * cmd[4]), 1.0 - 2.0 * (cmd[4] * cmd[4] + cmd[5] * cmd[5]));
    s1[6] = cmd[0] - a.arm_dist * cos(yaw); s1[7] = cmd[1] - a.arm_dist * sin(yaw); s1[8] = zRef; s1[9] = yaw; s1[10] = 0.0; s1[11] = 0.0;
    const Vec3<double> od = quaternionDistance(eeCur + 3, cmd + 3);
    const double disp = sqrt((cmd[0] - eeCur[0]) * (cmd[0] - eeCur[0]) + (cmd[1] - eeCur[1]) * (cmd[1] - eeCur[1]) + (cmd[2] - eeCur[2]) * (cmd[2] - eeCur[2]));
    const double rot = sqrt(od.x * od.x + od.y * od.y + od.z * od.z);
    tReach = t + fmax(rot / st.target_rotation_velocity, disp / st.target_displacement_velocity);   // estimateTimeToTarget (:40-57)
    for (int k = 0; k < 7; ++k) { s0[30 + k] = eeCur[k]; s1[30 + k] = cmd[k]; lastEe[k] = cmd[k]; }
  } else {
    // QMController::starting (QMController.cpp:107-113): hold the measured state, arm at its initial configuration, EE at the
    // spawn pose; a single knot upstream -- emitted twice here so that every kind has two knots
    for (int k = 0; k < 24; ++k) s0[k] = x[k];
    for (int k = 0; k < 6; ++k) s0[24 + k] = st.initial_state[24 + k];
    s0[30] = a.start_x + a.arm_dist * cos(a.start_psi); s0[31] = a.start_y + a.arm_dist * sin(a.start_psi); s0[32] = st.com_height + rbd[5];
    s0[33] = 0.0; s0[34] = 0.0; s0[35] = sin(a.start_psi / 2); s0[36] = cos(a.start_psi / 2);
    for (int k = 0; k < 37; ++k) s1[k] = s0[k];
  }
  a.target_times[size_t(i) * 2] = t; a.target_times[size_t(i) * 2 + 1] = tReach;
  double* ts = a.target_states + size_t(i) * 74;
  for (int k = 0; k < 37; ++k) { ts[k] = s0[k]; ts[37 + k] = s1[k]; }
}

// ---- gait front end: one thread per instance tiles its template exactly as host_config.cpp: qmgpu_tile_gait does (same operations in the same
// order, multiply-add contraction off, so the event times are bit-identical to the host's)
struct GaitArgs {
  int batch, numTemplates;
  const qmgpu_gait* templates;   // device copy
  const int* gaitIndex; const int* prevMode; double transitionStance; const double* tPhase0; const double* tBegin; const double* tEnd;
  int* numEvents; double* eventTimes; int* modes; int* status;
};
__global__ void __launch_bounds__(64) gait_schedule_kernel(GaitArgs a) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.batch) return;
  double* ev = a.eventTimes + size_t(i) * QMGPU_MAX_EVENTS;
  int* md = a.modes + size_t(i) * (QMGPU_MAX_EVENTS + 1);
  auto stanceOnly = [&](int st) {
    a.numEvents[i] = 0;
    for (int k = 0; k < QMGPU_MAX_EVENTS; ++k) ev[k] = 1e300;
    for (int k = 0; k <= QMGPU_MAX_EVENTS; ++k) md[k] = 15;
    if (a.status) a.status[i] = st;
  };
  const int gi = a.gaitIndex[i];
  if (gi < 0 || gi >= a.numTemplates) { stanceOnly(QMGPU_ERR_INVALID_ARGUMENT); return; }
  const qmgpu_gait& g = a.templates[gi];
  if (g.num_modes < 1 || g.num_modes > QMGPU_MAX_EVENTS) { stanceOnly(QMGPU_ERR_INVALID_ARGUMENT); return; }   // before switching_times[num_modes] is touched
  const double period = g.switching_times[g.num_modes] - g.switching_times[0];
  if (!(period > 0.0) || !(period < 1e300)) { stanceOnly(QMGPU_ERR_INVALID_ARGUMENT); return; }
  const int prevMode = a.prevMode ? a.prevMode[i] : 15;
  if (prevMode < 0 || prevMode > 15) { stanceOnly(QMGPU_ERR_INVALID_ARGUMENT); return; }
  const double tSwitch = a.tPhase0[i], tBegin = a.tBegin[i], tEnd = a.tEnd[i];
  if (!(fabs(tSwitch) < 1e300) || !(fabs(tBegin) < 1e300) || !(fabs(tEnd) < 1e300)) { stanceOnly(QMGPU_ERR_INVALID_ARGUMENT); return; }   // NaN / infinite times
  const bool transition = prevMode != 15 && prevMode != g.modes[0] && a.transitionStance > 0.0;
  const double tPhase0 = transition ? tSwitch + a.transitionStance : tSwitch;
  double start = tPhase0;
  if (tBegin > tPhase0) {
    const double cycles = floor((tBegin - tPhase0) / period);
    start = tPhase0 + (cycles >= 1.0 ? cycles - 1.0 : 0.0) * period;
  }
  // events are produced in order and merged on the fly: an event whose mode equals the last kept mode disappears
  int n = 0, lastMode = (start > tPhase0) ? 15 : prevMode;
  bool overflow = false;
  auto push = [&](double t, int modeAfter) {
    if (modeAfter == lastMode) return;
    if (n < QMGPU_MAX_EVENTS) { ev[n] = t; md[n + 1] = modeAfter; }
    else overflow = true;
    ++n; lastMode = modeAfter;
  };
  md[0] = lastMode;
  if (transition && start == tPhase0) push(tSwitch, 15);
  double t = start;
  double evTime = start;
  // At most QMGPU_MAX_EVENTS + 2 cycles: a template with a mode change yields an event per cycle (more cycles overflow anyway), one without
  // changes yields none however often it is tiled -- so the bound changes no result, it only ends the loop (as host_config.cpp: qmgpu_switch_gait).
  for (int cycle = 0; t < tEnd && !overflow && cycle < QMGPU_MAX_EVENTS + 2; ++cycle) {
    for (int m = 0; m < g.num_modes; ++m) {
      push(evTime, g.modes[m]);
      t += g.switching_times[m + 1] - g.switching_times[m];
      evTime = t;
    }
  }
  // bounded loop ended before tEnd without overflowing: a template without mode changes -- the final phase starts after the last WHOLE cycle (host_config.cpp)
  if (t < tEnd && !overflow) { const double period = g.switching_times[g.num_modes] - g.switching_times[0]; evTime = t + ceil((tEnd - t) / period) * period; }
  push(evTime, 15);   // default final phase
  if (overflow) { stanceOnly(QMGPU_ERR_CAPACITY); return; }
  a.numEvents[i] = n;
  for (int k = n; k < QMGPU_MAX_EVENTS; ++k) ev[k] = 1e300;
  for (int k = n + 1; k <= QMGPU_MAX_EVENTS; ++k) md[k] = 15;
  if (a.status) a.status[i] = QMGPU_OK;
}

}  // namespace qmk
