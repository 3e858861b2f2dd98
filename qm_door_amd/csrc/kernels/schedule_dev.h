// Device-side gait / reference evaluation at one shooting-node time (wave-uniform scalar work).
//
// Mirrors: contact flags from the mode schedule (upstream SwitchedModelReferenceManager::getContactFlags, used at
// qm_interface/src/constraint/NormalVelocityConstraintCppAd.cpp:37-39 and LeggedRobotQuadraticTrackingCost.h:36),
// the swing z reference (QMPreComputation.cpp:56-66 -> upstream SwingTrajectoryPlanner, task.info:24-31), and the
// target interpolation (LeggedRobotQuadraticTrackingCost.h:37, EndEffectorConstraint.cpp:80-113).
#pragma once
#include "../../../include/qmgpu.h"
#include "gpu_rt.h"

namespace qmk {

struct Schedule {
  int numEvents;
  const double* eventTimes;  // [MAX_EVENTS]
  const int* modes;          // [MAX_EVENTS+1]
};

// Continuous-time lookup (policy evaluation): std::lower_bound on the event times (upstream ModeSchedule::modeAtTime ->
// lookup::findIndexInTimeArray): an event time itself still belongs to the phase before it.
__device__ __forceinline__ int phaseAt(const Schedule& s, double t) {
  int i = 0;
  while (i < s.numEvents && s.eventTimes[i] < t) ++i;
  return i;
}
// Mode lookup of a SHOOTING NODE: a node placed exactly on an event time is upstream's PostEvent node (evaluated at
// t + weakEpsilon by ocs2_sqp's getIntervalStart), so it takes the mode that STARTS there: std::upper_bound.
__device__ __forceinline__ int nodePhaseAt(const Schedule& s, double t) {
  int i = 0;
  while (i < s.numEvents && s.eventTimes[i] <= t) ++i;
  return i;
}
__device__ __forceinline__ bool contactOf(int mode, int leg) { return (mode >> (3 - leg)) & 1; }

// Hermite cubic in normalised time (upstream CubicSpline): value and time derivative at t.
__device__ __forceinline__ void cubic(double t0, double p0, double v0, double t1, double p1, double v1, double t, double& pos, double& vel) {
  const double dt = t1 - t0, dp = p1 - p0, dv = v1 - v0;
  const double c1 = v0 * dt, c2 = -(3.0 * v0 + dv) * dt + 3.0 * dp, c3 = (2.0 * v0 + dv) * dt - 2.0 * dp;
  const double tn = (t - t0) / dt;
  pos = ((c3 * tn + c2) * tn + c1) * tn + p0;
  vel = ((3.0 * c3 * tn + 2.0 * c2) * tn + c1) / dt;
}

__device__ inline void swingReference(const qmgpu_settings& st, const Schedule& s, int leg, double t, int phase, double& zpos, double& zvel) {
  const int numPhases = s.numEvents + 1;
  int startIdx = -1;
  for (int ip = phase - 1; ip >= 0; --ip) if (contactOf(s.modes[ip], leg)) { startIdx = ip; break; }
  int finalIdx = numPhases - 1;
  for (int ip = phase + 1; ip < numPhases; ++ip) if (contactOf(s.modes[ip], leg)) { finalIdx = ip - 1; break; }
  const double tStart = (startIdx >= 0) ? s.eventTimes[startIdx] : ((s.numEvents > 0 ? s.eventTimes[0] : t) - st.touchdown_after_horizon);
  const double tFinal = (finalIdx < numPhases - 1) ? s.eventTimes[finalIdx] : ((s.numEvents > 0 ? s.eventTimes[s.numEvents - 1] : t) + st.touchdown_after_horizon);
  const double scaling = fmin(1.0, (tFinal - tStart) / st.swing_time_scale);
  const double tMid = 0.5 * (tStart + tFinal), midHeight = scaling * st.swing_height;
  if (t < tMid) cubic(tStart, 0.0, scaling * st.liftoff_velocity, tMid, midHeight, 0.0, t, zpos, zvel);
  else cubic(tMid, midHeight, 0.0, tFinal, 0.0, scaling * st.touchdown_velocity, t, zpos, zvel);
}

// (index, alpha): alpha is the weight of the LEFT knot (upstream LinearInterpolation::timeSegment)
__device__ __forceinline__ void timeSegment(const double* times, int K, double t, int& index, double& alpha) {
  if (K <= 1) { index = 0; alpha = 1.0; return; }
  int lb = 0;
  while (lb < K && times[lb] < t) ++lb;
  const int interval = lb - 1, last = K - 1;
  if (interval < 0) { index = 0; alpha = 1.0; }
  else if (interval >= last) { index = max(last - 1, 0); alpha = 0.0; }
  else {
    const double len = times[interval + 1] - times[interval];
    index = interval;
    alpha = (len > 4.440892098500626e-16) ? (times[interval + 1] - t) / len : 1.0;
  }
}

// End-effector reference pose at t: position lerp, Eigen-style slerp from the left knot by (1 - alpha).
__device__ inline void eeReference(const double* times, const double* states, int K, double t, double pos[3], double quat[4]) {
  int idx; double alpha;
  timeSegment(times, K, t, idx, alpha);
  const double* lhs = states + size_t(idx) * QMGPU_NTARGET;
  if (K <= 1) {
    for (int i = 0; i < 3; ++i) pos[i] = lhs[30 + i];
    for (int i = 0; i < 4; ++i) quat[i] = lhs[33 + i];
    return;
  }
  const double* rhs = lhs + QMGPU_NTARGET;
  for (int i = 0; i < 3; ++i) pos[i] = alpha * lhs[30 + i] + (1.0 - alpha) * rhs[30 + i];
  const double tt = 1.0 - alpha;
  const double* a = lhs + 33; const double* b = rhs + 33;
  const double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  const double absD = fabs(d);
  double s0, s1;
  if (absD >= 1.0 - 2.220446049250313e-16) { s0 = 1.0 - tt; s1 = tt; }
  else { const double theta = acos(absD), sinTheta = sin(theta); s0 = sin((1.0 - tt) * theta) / sinTheta; s1 = sin(tt * theta) / sinTheta; }
  if (d < 0) s1 = -s1;
  for (int i = 0; i < 4; ++i) quat[i] = s0 * a[i] + s1 * b[i];
}
// state reference component i (i < 30) at the segment already located
__device__ __forceinline__ double xReference(const double* states, int K, int idx, double alpha, int i) {
  const double* lhs = states + size_t(idx) * QMGPU_NTARGET;
  if (K <= 1) return lhs[i];
  return alpha * lhs[i] + (1.0 - alpha) * lhs[QMGPU_NTARGET + i];
}

// upstream RelaxedBarrierPenalty (settings task.info:291-316)
struct Barrier {
  double mu, delta;
  __device__ __forceinline__ double value(double h) const { const double q = (h - 2.0 * delta) / delta; return h > delta ? -mu * log(h) : mu * (-log(delta) + 0.5 * q * q - 0.5); }
  __device__ __forceinline__ double d1(double h) const { return h > delta ? -mu / h : mu * ((h - 2.0 * delta) / (delta * delta)); }
  __device__ __forceinline__ double d2(double h) const { return h > delta ? mu / (h * h) : mu / (delta * delta); }
};

}  // namespace qmk
