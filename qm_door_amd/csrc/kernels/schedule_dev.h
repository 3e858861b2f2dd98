// Device-side gait / reference evaluation at one shooting-node time (wave-uniform scalar work).
//
// Mirrors: contact flags from the mode schedule (upstream SwitchedModelReferenceManager::getContactFlags, used at
// qm_interface/src/constraint/NormalVelocityConstraintCppAd.cpp:37-39 and LeggedRobotQuadraticTrackingCost.h:36),
// the swing z reference (QMPreComputation.cpp:56-66 -> upstream SwingTrajectoryPlanner, task.info:24-31), and the
// target interpolation (LeggedRobotQuadraticTrackingCost.h:37, EndEffectorConstraint.cpp:80-113).
#pragma once
#include "problem_r.h"
#include "gpu_rt.h"

namespace qmk {

struct Schedule {
  int numEvents;
  const real* eventTimes;  // [MAX_EVENTS]
  const int* modes;          // [MAX_EVENTS+1]
};

// Continuous-time lookup (policy evaluation): std::lower_bound on the event times (upstream ModeSchedule::modeAtTime ->
// lookup::findIndexInTimeArray): an event time itself still belongs to the phase before it.
__device__ __forceinline__ int phaseAt(const Schedule& s, real t) {
  int i = 0;
  while (i < s.numEvents && s.eventTimes[i] < t) ++i;
  return i;
}
// Mode lookup of a SHOOTING NODE: a node placed exactly on an event time is upstream's PostEvent node (evaluated at
// t + weakEpsilon by ocs2_sqp's getIntervalStart), so it takes the mode that STARTS there: std::upper_bound.
__device__ __forceinline__ int nodePhaseAt(const Schedule& s, real t) {
  int i = 0;
  while (i < s.numEvents && s.eventTimes[i] <= t) ++i;
  return i;
}
__device__ __forceinline__ bool contactOf(int mode, int leg) { return (mode >> (3 - leg)) & 1; }

// Hermite cubic in normalised time (upstream CubicSpline): value and time derivative at t.
__device__ __forceinline__ void cubic(real t0, real p0, real v0, real t1, real p1, real v1, real t, real& pos, real& vel) {
  const real dt = t1 - t0, dp = p1 - p0, dv = v1 - v0;
  const real c1 = v0 * dt, c2 = -(3.0_r * v0 + dv) * dt + 3.0_r * dp, c3 = (2.0_r * v0 + dv) * dt - 2.0_r * dp;
  const real tn = (t - t0) / dt;
  pos = ((c3 * tn + c2) * tn + c1) * tn + p0;
  vel = ((3.0_r * c3 * tn + 2.0_r * c2) * tn + c1) / dt;
}

__device__ inline void swingReference(const SettingsR& st, const Schedule& s, int leg, real t, int phase, real& zpos, real& zvel) {
  const int numPhases = s.numEvents + 1;
  int startIdx = -1;
  for (int ip = phase - 1; ip >= 0; --ip) if (contactOf(s.modes[ip], leg)) { startIdx = ip; break; }
  int finalIdx = numPhases - 1;
  for (int ip = phase + 1; ip < numPhases; ++ip) if (contactOf(s.modes[ip], leg)) { finalIdx = ip - 1; break; }
  const real tStart = (startIdx >= 0) ? s.eventTimes[startIdx] : ((s.numEvents > 0 ? s.eventTimes[0] : t) - st.touchdown_after_horizon);
  const real tFinal = (finalIdx < numPhases - 1) ? s.eventTimes[finalIdx] : ((s.numEvents > 0 ? s.eventTimes[s.numEvents - 1] : t) + st.touchdown_after_horizon);
  const real scaling = fmin(1.0_r, (tFinal - tStart) / st.swing_time_scale);
  const real tMid = 0.5_r * (tStart + tFinal), midHeight = scaling * st.swing_height;
  if (t < tMid) cubic(tStart, 0.0_r, scaling * st.liftoff_velocity, tMid, midHeight, 0.0_r, t, zpos, zvel);
  else cubic(tMid, midHeight, 0.0_r, tFinal, 0.0_r, scaling * st.touchdown_velocity, t, zpos, zvel);
}

// (index, alpha): alpha is the weight of the LEFT knot (upstream LinearInterpolation::timeSegment)
__device__ __forceinline__ void timeSegment(const real* times, int K, real t, int& index, real& alpha) {
  if (K <= 1) { index = 0; alpha = 1.0_r; return; }
  int lb = 0;
  while (lb < K && times[lb] < t) ++lb;
  const int interval = lb - 1, last = K - 1;
  if (interval < 0) { index = 0; alpha = 1.0_r; }
  else if (interval >= last) { index = max(last - 1, 0); alpha = 0.0_r; }
  else {
    const real len = times[interval + 1] - times[interval];
    index = interval;
    alpha = (len > 2.0_r * REAL_EPS) ? (times[interval + 1] - t) / len : 1.0_r;
  }
}

// The same for a WHOLE WAVEFRONT and a wavefront-uniform t: the lower bound is the number of entries below t (the grid is non-decreasing), counted with one
// ballot per 64 entries instead of the dependent chain of loads of the scan above (one memory round trip per entry: a resampling pass over a 100-node
// horizon walked ~50 entries per node).  Index and alpha as timeSegment, bit for bit.
__device__ __forceinline__ int gridCountBelow(const real* times, int K, real t, int lane) {   // entries of a non-decreasing grid below t (a whole wavefront, t uniform)
  int lb = 0;
  for (int base = 0; base < K; base += 64) {
    const int i = base + lane;
    const bool below = i < K && times[i < K ? i : 0] < t;
    const unsigned long long m = qmBallot(below);
    lb += qmPopCount(m);
    if (m != ~0ull) break;   // nothing below t behind the first entry that is not
  }
  return lb;
}
__device__ __forceinline__ void timeSegmentWave(const real* times, int K, real t, int lane, int& index, real& alpha) {
  if (K <= 1) { index = 0; alpha = 1.0_r; return; }
  const int lb = gridCountBelow(times, K, t, lane);
  const int interval = lb - 1, last = K - 1;
  if (interval < 0) { index = 0; alpha = 1.0_r; }
  else if (interval >= last) { index = max(last - 1, 0); alpha = 0.0_r; }
  else {
    const real len = times[interval + 1] - times[interval];
    index = interval;
    alpha = (len > 2.0_r * REAL_EPS) ? (times[interval + 1] - t) / len : 1.0_r;
  }
}

// End-effector reference pose at t: position lerp, Eigen-style slerp from the left knot by (1 - alpha).
__device__ inline void eeReference(const real* times, const real* states, int K, real t, real pos[3], real quat[4]) {
  int idx; real alpha;
  timeSegment(times, K, t, idx, alpha);
  const real* lhs = states + size_t(idx) * QMGPU_NTARGET;
  if (K <= 1) {
    for (int i = 0; i < 3; ++i) pos[i] = lhs[30 + i];
    for (int i = 0; i < 4; ++i) quat[i] = lhs[33 + i];
    return;
  }
  const real* rhs = lhs + QMGPU_NTARGET;
  for (int i = 0; i < 3; ++i) pos[i] = alpha * lhs[30 + i] + (1.0_r - alpha) * rhs[30 + i];
  const real tt = 1.0_r - alpha;
  const real* a = lhs + 33; const real* b = rhs + 33;
  const real d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  const real absD = fabs(d);
  real s0, s1;
  if (absD >= 1.0_r - REAL_EPS) { s0 = 1.0_r - tt; s1 = tt; }
  else { const real theta = acos(absD), sinTheta = sin(theta); s0 = sin((1.0_r - tt) * theta) / sinTheta; s1 = sin(tt * theta) / sinTheta; }
  if (d < 0) s1 = -s1;
  for (int i = 0; i < 4; ++i) quat[i] = s0 * a[i] + s1 * b[i];
}
// Force tracking (own formulation, qmgpu_settings::ee_contact_stiffness): end-effector force reference and anchor of the compliant
// environment at t, interpolated linearly between the target knots; contact [K][6] = f_ref (3), p_env (3)
__device__ inline void eeContactReference(const real* times, const real* contact, int K, real t, real fref[3], real env[3]) {
  int idx; real alpha;
  timeSegment(times, K, t, idx, alpha);
  const real* lhs = contact + size_t(idx) * 6;
  const real* rhs = K > 1 ? lhs + 6 : lhs;
  for (int i = 0; i < 3; ++i) { fref[i] = alpha * lhs[i] + (1.0_r - alpha) * rhs[i]; env[i] = alpha * lhs[3 + i] + (1.0_r - alpha) * rhs[3 + i]; }
}
// state reference component i (i < 30) at the segment already located
__device__ __forceinline__ real xReference(const real* states, int K, int idx, real alpha, int i) {
  const real* lhs = states + size_t(idx) * QMGPU_NTARGET;
  if (K <= 1) return lhs[i];
  return alpha * lhs[i] + (1.0_r - alpha) * lhs[QMGPU_NTARGET + i];
}

// upstream RelaxedBarrierPenalty (settings task.info:291-316)
struct Barrier {
  real mu, delta;
  __device__ __forceinline__ real value(real h) const { const real q = (h - 2.0_r * delta) / delta; return h > delta ? -mu * log(h) : mu * (-log(delta) + 0.5_r * q * q - 0.5_r); }
  __device__ __forceinline__ real d1(real h) const { return h > delta ? -mu / h : mu * ((h - 2.0_r * delta) / (delta * delta)); }
  __device__ __forceinline__ real d2(real h) const { return h > delta ? mu / (h * h) : mu / (delta * delta); }
  // the same three numbers at once, log(delta) given (layout.h: QM_BC_*): one logarithm and one division in the logarithmic branch (value alone computed two
  // logarithms and a division, d1 and d2 three more divisions); the quadratic branch is a real branch -- no lane is in it on a healthy iterate
  __device__ __forceinline__ void eval(real h, real logDelta, real& val, real& g, real& hs) const {
    if (h > delta) { const real r = 1.0_r / h; val = -mu * log(h); g = -mu * r; hs = mu * r * r; }
    else { const real q = (h - 2.0_r * delta) / delta; val = mu * (-logDelta + 0.5_r * q * q - 0.5_r); g = mu * ((h - 2.0_r * delta) / (delta * delta)); hs = mu / (delta * delta); }
  }
  __device__ __forceinline__ real valueL(real h, real logDelta) const {
    if (h > delta) return -mu * log(h);
    const real q = (h - 2.0_r * delta) / delta;
    return mu * (-logDelta + 0.5_r * q * q - 0.5_r);
  }
};

}  // namespace qmk
