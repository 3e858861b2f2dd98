// qmo_mpc.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; parity unpinned, see qmo_core.h).
//
// One multiple-shooting SQP iteration for the AlienGo+Z1 centroidal OCP, restating
//   - the problem assembled in qm_interface/src/QMInterface.cpp:79-142 (cost, soft constraints,
//     per-foot equality constraints, initializer) and
//   - upstream ocs2_sqp::SqpSolver::runImpl as configured by qm_controllers/config/task.info:76-93
//     (RK2 sensitivity discretisation, dt-scaled cost, QR constraint projection, Riccati solve of the
//     equality-free QP, filter line search).
#pragma once
#include <memory>
#include <thread>
#include "qmo_model.h"

namespace qmo {

using D60 = Dual<60>;

// ------------------------------------------------------------------------------------------------ modes / schedule
// upstream ocs2_legged_robot MotionPhaseDefinition.h: mode = 8*LF + 4*RF + 2*LH + RH
inline void modeToContactFlags(int mode, bool flags[4]) { flags[0] = mode & 8; flags[1] = mode & 4; flags[2] = mode & 2; flags[3] = mode & 1; }

struct ModeSchedule {
  int numEvents = 0;
  const double* eventTimes = nullptr;  // [numEvents]
  const int32_t* modes = nullptr;      // [numEvents+1]
  // upstream ModeSchedule::modeAtTime -> lookup::findIndexInTimeArray == std::lower_bound
  int phaseAt(double t) const { int i = 0; while (i < numEvents && eventTimes[i] < t) ++i; return i; }
  int modeAt(double t) const { return modes[phaseAt(t)]; }
  // shooting nodes: a node on an event time is upstream's PostEvent node (interval start = t + weakEpsilon) and takes the
  // mode that starts there (std::upper_bound)
  int nodePhaseAt(double t) const { int i = 0; while (i < numEvents && eventTimes[i] <= t) ++i; return i; }
  int nodeModeAt(double t) const { return modes[nodePhaseAt(t)]; }
};

// upstream ocs2_legged_robot CubicSpline (normalised-time Hermite cubic)
struct CubicSpline {
  double t0, t1, dt, c0, c1, c2, c3;
  CubicSpline(double ts, double ps, double vs, double te, double pe, double ve) {
    t0 = ts; t1 = te; dt = te - ts;
    const double dp = pe - ps, dv = ve - vs;
    c0 = ps;
    c1 = vs * dt;
    c2 = -(3.0 * vs + dv) * dt + 3.0 * dp;
    c3 = (2.0 * vs + dv) * dt - 2.0 * dp;
  }
  double position(double t) const { const double tn = (t - t0) / dt; return c3 * tn * tn * tn + c2 * tn * tn + c1 * tn + c0; }
  double velocity(double t) const { const double tn = (t - t0) / dt; return (3.0 * c3 * tn * tn + 2.0 * c2 * tn + c1) / dt; }
};

// upstream SwingTrajectoryPlanner::update + SplineCpg for flat terrain (height 0), queried at one time.
// Settings: task.info:24-31.  Returns z position / z velocity reference of foot `leg` at time t.
inline void swingReference(const qmgpu_settings& st, const ModeSchedule& ms, int leg, double t, double* zpos, double* zvel) {
  const int numPhases = ms.numEvents + 1;
  const int p = ms.nodePhaseAt(t);
  auto inContact = [&](int phase) { bool f[4]; modeToContactFlags(ms.modes[phase], f); return f[leg]; };
  if (inContact(p)) { *zpos = 0.0; *zvel = 0.0; return; }
  int startIdx = -1;
  for (int ip = p - 1; ip >= 0; --ip) if (inContact(ip)) { startIdx = ip; break; }
  int finalIdx = numPhases - 1;
  for (int ip = p + 1; ip < numPhases; ++ip) if (inContact(ip)) { finalIdx = ip - 1; break; }
  // upstream throws when lift-off / touch-down is undefined; this restatement extends the swing by
  // touchdownAfterHorizon instead (documented deviation; never hit by schedules from the gait tiler).
  const double tStart = (startIdx >= 0) ? ms.eventTimes[startIdx] : ((ms.numEvents > 0 ? ms.eventTimes[0] : t) - st.touchdown_after_horizon);
  const double tFinal = (finalIdx < numPhases - 1) ? ms.eventTimes[finalIdx] : ((ms.numEvents > 0 ? ms.eventTimes[ms.numEvents - 1] : t) + st.touchdown_after_horizon);
  const double scaling = std::min(1.0, (tFinal - tStart) / st.swing_time_scale);
  const double tMid = 0.5 * (tStart + tFinal);
  const double midHeight = scaling * st.swing_height;
  if (t < tMid) {
    CubicSpline s(tStart, 0.0, scaling * st.liftoff_velocity, tMid, midHeight, 0.0);
    *zpos = s.position(t); *zvel = s.velocity(t);
  } else {
    CubicSpline s(tMid, midHeight, 0.0, tFinal, 0.0, scaling * st.touchdown_velocity);
    *zpos = s.position(t); *zvel = s.velocity(t);
  }
}

// ------------------------------------------------------------------------------------------------ references
struct Target {
  int K = 0;
  const double* times = nullptr;   // [K]
  const double* states = nullptr;  // [K][37]
  const double* contact = nullptr; // [K][6] or null: end-effector force reference (3) and environment anchor (3) per knot (force tracking)
};
// upstream LinearInterpolation::timeSegment: (index, alpha = weight of the LEFT knot)
inline void timeSegment(const double* times, int K, double t, int* index, double* alpha) {
  if (K <= 1) { *index = 0; *alpha = 1.0; return; }
  int lb = 0; while (lb < K && times[lb] < t) ++lb;
  const int interval = lb - 1;
  const int last = K - 1;
  if (interval >= 0) {
    if (interval < last) {
      const double len = times[interval + 1] - times[interval];
      *index = interval;
      *alpha = (len > 2.0 * 2.220446049250313e-16) ? (times[interval + 1] - t) / len : 1.0;
    } else { *index = std::max(last - 1, 0); *alpha = 0.0; }
  } else { *index = 0; *alpha = 1.0; }
}
// x_ref(t) = TargetTrajectories::getDesiredState(t).head(30) (LeggedRobotQuadraticTrackingCost.h:37);
// EE pose per EndEffectorConstraint::interpolateEndEffectorPose (EndEffectorConstraint.cpp:80-113): position lerp,
// Eigen slerp from the left knot with parameter (1 - alpha).
inline void referenceAt(const Target& tg, double t, double xref[30], double eePos[3], double eeQuat[4]) {
  int idx; double alpha;
  timeSegment(tg.times, tg.K, t, &idx, &alpha);
  const double* lhs = tg.states + size_t(idx) * 37;
  if (tg.K <= 1) {
    for (int i = 0; i < 30; ++i) xref[i] = lhs[i];
    for (int i = 0; i < 3; ++i) eePos[i] = lhs[30 + i];
    for (int i = 0; i < 4; ++i) eeQuat[i] = lhs[33 + i];
    return;
  }
  const double* rhs = lhs + 37;
  for (int i = 0; i < 30; ++i) xref[i] = alpha * lhs[i] + (1.0 - alpha) * rhs[i];
  for (int i = 0; i < 3; ++i) eePos[i] = alpha * lhs[30 + i] + (1.0 - alpha) * rhs[30 + i];
  // Eigen::QuaternionBase::slerp(t, other)
  const double tt = 1.0 - alpha;
  const double* a = lhs + 33; const double* b = rhs + 33;
  const double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  const double absD = std::fabs(d);
  double s0, s1;
  if (absD >= 1.0 - 2.220446049250313e-16) { s0 = 1.0 - tt; s1 = tt; }
  else {
    const double theta = std::acos(absD), sinTheta = std::sin(theta);
    s0 = std::sin((1.0 - tt) * theta) / sinTheta;
    s1 = std::sin(tt * theta) / sinTheta;
  }
  if (d < 0) s1 = -s1;
  for (int i = 0; i < 4; ++i) eeQuat[i] = s0 * a[i] + s1 * b[i];
}

// upstream ocs2_legged_robot weightCompensatingInput
inline void weightCompensatingInput(const qmgpu_problem& P, int mode, double u[30]) {
  bool fl[4]; modeToContactFlags(mode, fl);
  int n = 0; for (bool f : fl) n += f;
  for (int i = 0; i < 30; ++i) u[i] = 0.0;
  if (n > 0) for (int c = 0; c < 4; ++c) if (fl[c]) u[3 * c + 2] = P.model.total_mass * P.settings.gravity / n;
}

// ------------------------------------------------------------------------------------------------ penalties
// upstream RelaxedBarrierPenalty
// end-effector contact of the node at time t (references interpolated linearly like every other one); installs it for the flow map
// evaluations of the calling thread and removes it again at the end of the scope
struct ContactScope {
  ContactScope(const qmgpu_settings& st, const Target& tg, double t) {
    EeContact c;
    if (tg.contact && st.ee_contact_stiffness != 0.0) {
      int idx; double alpha;
      timeSegment(tg.times, tg.K, t, &idx, &alpha);
      const double* lhs = tg.contact + size_t(idx) * 6;
      const double* rhs = tg.K > 1 ? lhs + 6 : lhs;
      c.K = st.ee_contact_stiffness;
      for (int a = 0; a < 3; ++a) { c.fref[a] = alpha * lhs[a] + (1.0 - alpha) * rhs[a]; c.env[a] = alpha * lhs[3 + a] + (1.0 - alpha) * rhs[3 + a]; }
    }
    g_eeContact = c;
  }
  ~ContactScope() { g_eeContact = EeContact(); }
};

struct Barrier {
  double mu, delta;
  double value(double h) const { return h > delta ? -mu * std::log(h) : mu * (-std::log(delta) + 0.5 * ((h - 2.0 * delta) / delta) * ((h - 2.0 * delta) / delta) - 0.5); }
  double d1(double h) const { return h > delta ? -mu / h : mu * ((h - 2.0 * delta) / (delta * delta)); }
  double d2(double h) const { return h > delta ? mu / (h * h) : mu / (delta * delta); }
};

// ------------------------------------------------------------------------------------------------ dynamics linearisation
inline void flowMapLinearization(const qmgpu_problem& P, const double* x, const double* u, double* f, Mat& A, Mat& B) {
  static thread_local D60 xd[30], ud[30], fd[30];
  for (int i = 0; i < 30; ++i) { xd[i] = D60(x[i]); xd[i].d[i] = 1.0; ud[i] = D60(u[i]); ud[i].d[30 + i] = 1.0; }
  flowMapD60(P.model, P.settings.gravity, xd, ud, fd, nullptr);
  A = Mat(30, 30); B = Mat(30, 30);
  for (int i = 0; i < 30; ++i) { f[i] = fd[i].v; for (int j = 0; j < 30; ++j) { A(i, j) = fd[i].d[j]; B(i, j) = fd[i].d[30 + j]; } }
}

// Input weight R' (QMInterface.cpp:274-299): the 12x12 leg joint-velocity block is J^T R_task J with J the
// feet Jacobian (contact order) w.r.t. the 12 leg joints at the initial state.
inline Mat inputWeight(const qmgpu_problem& P) {
  Kin<double> k;
  forwardKinematics<double>(P.model, P.settings.initial_state + 6, k);
  Mat J(12, 12);
  for (int c = 0; c < 4; ++c)
    for (int j = 0; j < 12; ++j) { V3<double> lin, ang; pointJacobianColumn(P.model, k, P.model.foot_body[c], k.foot[c], 6 + j, lin, ang); for (int a = 0; a < 3; ++a) J(3 * c + a, j) = lin[a]; }
  Mat Rt = Mat::from(P.settings.R_task, 30, 30);
  Mat R = Rt;
  setBlock(R, 12, 12, T(J) * block(Rt, 12, 12, 12, 12) * J);
  return R;
}

// ------------------------------------------------------------------------------------------------ one node
struct NodeLQ {
  // dynamics x+ ~ A dx + B du + b ; cost (already dt-scaled) ; equality constraints C dx + D du + e = 0
  Mat A, B, Q, R, Pm, C, D;
  Vec b, q, r, e;
  double cost = 0.0;
  int nc = 0;
  // projection du = Pe + Px dx + Pu dut
  Mat Px, Pu; Vec Pe;
  // projected
  Mat At, Bt, Qt, Rt, Pt; Vec bt, qt, rt;
};

struct NodeMetrics { double cost = 0, dynViolationSSE = 0, eqViolationSSE = 0; };

struct Problem {
  const qmgpu_problem* P;
  Mat Rw;  // R'
  ModeSchedule ms;
  Target tg;
  int nodeThreads = 1;  // worker threads over the shooting nodes (task.info:78 nThreads = 3 in the reference); the results do not depend on it
};

// f(k) for k in [0, count), node k on thread k % threads (upstream's SqpSolver distributes the nodes over its thread pool the same way)
template <class F> inline void forEachNode(int count, int threads, F&& f) {
  if (threads <= 1) { for (int k = 0; k < count; ++k) f(k); return; }
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) pool.emplace_back([&, t]() { for (int k = t; k < count; k += threads) f(k); });
  for (auto& th : pool) th.join();
}

// value of every node term at (t,x,u): cost (unscaled), equality constraint vector
inline double nodeCost(const Problem& pr, double t, const double* x, const double* u, int mode, bool terminal, std::vector<double>* eq,
                       const FlowAux<double>& aux) {
  const qmgpu_settings& st = pr.P->settings;
  double xref[30], eePos[3], eeQuat[4];
  referenceAt(pr.tg, t, xref, eePos, eeQuat);
  double cost = 0.0;
  // end-effector soft constraint (QMInterface.cpp:103-104,147-172)
  {
    double qee[4]; matrixToQuaternion(aux.eeRot, qee);
    const V3<double> od = quaternionDistance(qee, eeQuat);
    const double muP = terminal ? st.ee_final_mu_position : st.ee_mu_position;
    const double muO = terminal ? st.ee_final_mu_orientation : st.ee_mu_orientation;
    for (int a = 0; a < 3; ++a) { const double hp = aux.eePos[a] - eePos[a]; cost += 0.5 * muP * hp * hp + 0.5 * muO * od[a] * od[a]; }
  }
  if (terminal) return cost;
  if (g_eeContact.K != 0.0)   // end-effector force soft constraint 1/2 mu_f |f_e - f_ref|^2 (force tracking, own formulation)
    for (int a = 0; a < 3; ++a) { const double hf = -g_eeContact.K * (aux.eePos[a] - g_eeContact.env[a]) - g_eeContact.fref[a]; cost += 0.5 * st.ee_force_mu * hf * hf; }
  bool fl[4]; modeToContactFlags(mode, fl);
  double unom[30]; weightCompensatingInput(*pr.P, mode, unom);
  Vec dx(30), du(30);
  for (int i = 0; i < 30; ++i) { dx[i] = x[i] - xref[i]; du[i] = u[i] - unom[i]; }
  const Mat Q = Mat::from(st.Q, 30, 30);
  cost += 0.5 * dot(dx, Q * dx) + 0.5 * dot(du, pr.Rw * du);
  // arm joint position / velocity soft box (QMInterface.cpp:177-259), with the offset of initializeOffset(0,0,0)
  const Barrier bp{st.joint_pos_barrier_mu, st.joint_pos_barrier_delta}, bv{st.joint_vel_barrier_mu, st.joint_vel_barrier_delta};
  for (int i = 0; i < 6; ++i) {
    const double lo = pr.P->model.q_lower[12 + i], up = pr.P->model.q_upper[12 + i];
    cost += bp.value(x[24 + i] - lo) + bp.value(up - x[24 + i]) - (bp.value(0.0 - lo) + bp.value(up - 0.0));
    cost += bv.value(u[24 + i] - st.arm_vel_lower[i]) + bv.value(st.arm_vel_upper[i] - u[24 + i]) - (bv.value(0.0 - st.arm_vel_lower[i]) + bv.value(st.arm_vel_upper[i] - 0.0));
  }
  // friction cone soft constraint on stance feet (QMInterface.cpp:116-121,344-358)
  const Barrier bf{st.friction_barrier_mu, st.friction_barrier_delta};
  for (int c = 0; c < 4; ++c) if (fl[c]) {
    const double fx = u[3 * c], fy = u[3 * c + 1], fz = u[3 * c + 2];
    const double h = st.friction_coefficient * fz - std::sqrt(fx * fx + fy * fy + st.friction_regularization);
    cost += bf.value(h);
  }
  if (eq) {
    eq->clear();
    double zp, zv;
    for (int c = 0; c < 4; ++c) {
      if (!fl[c]) for (int a = 0; a < 3; ++a) eq->push_back(u[3 * c + a]);                // zeroForce  (QMInterface.cpp:123-124)
      if (fl[c]) for (int a = 0; a < 3; ++a) eq->push_back(aux.footVel[c][a] + (a == 2 ? st.position_error_gain * aux.footPos[c][2] : 0.0));   // zeroVelocity (QMInterface.cpp:126,324-339; Ax(2,2) = positionErrorGain)
      if (!fl[c]) {                                                                      // normalVelocity (QMPreComputation.cpp:56-66)
        swingReference(st, pr.ms, c, t, &zp, &zv);
        eq->push_back(aux.footVel[c][2] - zv + st.position_error_gain * (aux.footPos[c][2] - zp));
      }
    }
  }
  return cost;
}

// RK2 (Heun) step, upstream ocs2 rk2Discretization: x+ = x + dt/2 (k1 + k2), k2 = f(x + dt k1, u)
inline void rk2Step(const qmgpu_problem& P, double dt, const double* x, const double* u, double* xn) {
  double k1[30], k2[30], x2[30];
  flowMap<double>(P.model, P.settings.gravity, x, u, k1);
  for (int i = 0; i < 30; ++i) x2[i] = x[i] + dt * k1[i];
  flowMap<double>(P.model, P.settings.gravity, x2, u, k2);
  for (int i = 0; i < 30; ++i) xn[i] = x[i] + 0.5 * dt * (k1[i] + k2[i]);
}

inline NodeMetrics nodeMetrics(const Problem& pr, double t, double dt, const double* x, const double* u, const double* xnext, bool terminal) {
  NodeMetrics m;
  const ContactScope contact(pr.P->settings, pr.tg, terminal ? 1e300 : t);
  if (terminal) g_eeContact = EeContact();   // the contact acts on the intermediate nodes only
  const int mode = pr.ms.nodeModeAt(t);
  static thread_local double f[30];
  FlowAux<double> aux;
  double uz[30] = {0};
  flowMap<double>(pr.P->model, pr.P->settings.gravity, x, terminal ? uz : u, f, &aux);
  std::vector<double> eq;
  const double c = nodeCost(pr, t, x, u, mode, terminal, terminal ? nullptr : &eq, aux);
  if (terminal) { m.cost = c; return m; }
  m.cost = dt * c;
  double xn[30];
  rk2Step(*pr.P, dt, x, u, xn);
  for (int i = 0; i < 30; ++i) m.dynViolationSSE += dt * (xn[i] - xnext[i]) * (xn[i] - xnext[i]);
  for (double v : eq) m.eqViolationSSE += dt * v * v;
  return m;
}

// LQ approximation of one node (setupIntermediateNode / setupTerminalNode of upstream ocs2_sqp)
inline void nodeLQ(const Problem& pr, double t, double dt, const double* x, const double* u, const double* xnext, bool terminal, NodeLQ& o) {
  const qmgpu_problem& P = *pr.P;
  const qmgpu_settings& st = P.settings;
  const int mode = pr.ms.nodeModeAt(t);
  bool fl[4]; modeToContactFlags(mode, fl);
  double xref[30], eePosRef[3], eeQuatRef[4];
  referenceAt(pr.tg, t, xref, eePosRef, eeQuatRef);
  const ContactScope contact(st, pr.tg, t);
  if (terminal) g_eeContact = EeContact();

  // ---- kinematic quantities with derivatives (dual numbers over [x;u])
  static thread_local D60 xd[30], ud[30], fd[30];
  for (int i = 0; i < 30; ++i) { xd[i] = D60(x[i]); xd[i].d[i] = 1.0; ud[i] = D60(terminal ? 0.0 : u[i]); ud[i].d[30 + i] = 1.0; }
  FlowAux<D60> aux;
  flowMapD60(P.model, st.gravity, xd, ud, fd, &aux);

  o.Q = Mat(30, 30); o.q = Vec(30, 0.0); o.R = Mat(30, 30); o.r = Vec(30, 0.0); o.Pm = Mat(30, 30);
  o.cost = 0.0;
  // ---- EE soft constraint, Gauss-Newton (StateSoftConstraint + QuadraticPenalty)
  {
    D60 qee[4]; matrixToQuaternion(aux.eeRot, qee);
    const V3<D60> od = quaternionDistance(qee, eeQuatRef);
    const double muP = terminal ? st.ee_final_mu_position : st.ee_mu_position;
    const double muO = terminal ? st.ee_final_mu_orientation : st.ee_mu_orientation;
    for (int a = 0; a < 6; ++a) {
      const D60 h = a < 3 ? (aux.eePos[a] - D60(eePosRef[a])) : od[a - 3];
      const double mu = a < 3 ? muP : muO;
      o.cost += 0.5 * mu * h.v * h.v;
      for (int i = 0; i < 30; ++i) { o.q[i] += mu * h.v * h.d[i]; for (int j = 0; j < 30; ++j) o.Q(i, j) += mu * h.d[i] * h.d[j]; }
    }
  }
  if (terminal) { o.nc = 0; return; }
  if (g_eeContact.K != 0.0) {   // end-effector force soft constraint, Gauss-Newton (state only)
    for (int a = 0; a < 3; ++a) {
      const D60 h = (-g_eeContact.K) * (aux.eePos[a] - D60(g_eeContact.env[a])) - D60(g_eeContact.fref[a]);
      o.cost += 0.5 * st.ee_force_mu * h.v * h.v;
      for (int i = 0; i < 30; ++i) { o.q[i] += st.ee_force_mu * h.v * h.d[i]; for (int j = 0; j < 30; ++j) o.Q(i, j) += st.ee_force_mu * h.d[i] * h.d[j]; }
    }
  }

  // ---- tracking cost
  double unom[30]; weightCompensatingInput(P, mode, unom);
  Vec dx(30), du(30);
  for (int i = 0; i < 30; ++i) { dx[i] = x[i] - xref[i]; du[i] = u[i] - unom[i]; }
  const Mat Qw = Mat::from(st.Q, 30, 30);
  o.Q = o.Q + Qw; o.R = o.R + pr.Rw;
  { const Vec Qdx = Qw * dx, Rdu = pr.Rw * du; o.q = o.q + Qdx; o.r = o.r + Rdu; o.cost += 0.5 * dot(dx, Qdx) + 0.5 * dot(du, Rdu); }
  // ---- arm joint limits
  const Barrier bp{st.joint_pos_barrier_mu, st.joint_pos_barrier_delta}, bv{st.joint_vel_barrier_mu, st.joint_vel_barrier_delta};
  for (int i = 0; i < 6; ++i) {
    const double lo = P.model.q_lower[12 + i], up = P.model.q_upper[12 + i];
    const double hl = x[24 + i] - lo, hu = up - x[24 + i];
    o.cost += bp.value(hl) + bp.value(hu) - (bp.value(-lo) + bp.value(up));
    o.q[24 + i] += bp.d1(hl) - bp.d1(hu);
    o.Q(24 + i, 24 + i) += bp.d2(hl) + bp.d2(hu);
    const double vl = u[24 + i] - st.arm_vel_lower[i], vu = st.arm_vel_upper[i] - u[24 + i];
    o.cost += bv.value(vl) + bv.value(vu) - (bv.value(-st.arm_vel_lower[i]) + bv.value(st.arm_vel_upper[i]));
    o.r[24 + i] += bv.d1(vl) - bv.d1(vu);
    o.R(24 + i, 24 + i) += bv.d2(vl) + bv.d2(vu);
  }
  // ---- friction cone (upstream FrictionConeConstraint, quadratic order, Config defaults regularization 25, hessianDiagonalShift 1e-6)
  const Barrier bf{st.friction_barrier_mu, st.friction_barrier_delta};
  for (int c = 0; c < 4; ++c) if (fl[c]) {
    const double fx = u[3 * c], fy = u[3 * c + 1], fz = u[3 * c + 2];
    const double F = std::sqrt(fx * fx + fy * fy + st.friction_regularization);
    const double h = st.friction_coefficient * fz - F;
    const double g[3] = {-fx / F, -fy / F, st.friction_coefficient};
    const double F3 = F * F * F;
    double H[3][3] = {{-(fy * fy + st.friction_regularization) / F3, fx * fy / F3, 0}, {fx * fy / F3, -(fx * fx + st.friction_regularization) / F3, 0}, {0, 0, 0}};
    for (int a = 0; a < 3; ++a) H[a][a] -= st.friction_hessian_shift;
    o.cost += bf.value(h);
    for (int a = 0; a < 3; ++a) {
      o.r[3 * c + a] += bf.d1(h) * g[a];
      for (int bq = 0; bq < 3; ++bq) o.R(3 * c + a, 3 * c + bq) += bf.d2(h) * g[a] * g[bq] + bf.d1(h) * H[a][bq];
    }
  }
  // ---- equality constraints, insertion order of QMInterface.cpp:116-131
  std::vector<std::vector<double>> rowsC, rowsD; std::vector<double> ev;
  auto pushDual = [&](const D60& h) { std::vector<double> c(30), d(30); for (int i = 0; i < 30; ++i) { c[i] = h.d[i]; d[i] = h.d[30 + i]; } rowsC.push_back(c); rowsD.push_back(d); ev.push_back(h.v); };
  for (int c = 0; c < 4; ++c) {
    if (!fl[c]) for (int a = 0; a < 3; ++a) pushDual(ud[3 * c + a]);
    if (fl[c]) for (int a = 0; a < 3; ++a) pushDual(a == 2 ? aux.footVel[c][2] + st.position_error_gain * aux.footPos[c][2] : aux.footVel[c][a]);
    if (!fl[c]) { double zp, zv; swingReference(st, pr.ms, c, t, &zp, &zv); pushDual(aux.footVel[c][2] - D60(zv) + st.position_error_gain * (aux.footPos[c][2] - D60(zp))); }
  }
  o.nc = int(ev.size());
  o.C = Mat(o.nc, 30); o.D = Mat(o.nc, 30); o.e = ev;
  for (int i = 0; i < o.nc; ++i) for (int j = 0; j < 30; ++j) { o.C(i, j) = rowsC[i][j]; o.D(i, j) = rowsD[i][j]; }

  // ---- dynamics: upstream rk2SensitivityDiscretization
  {
    double k1[30], k2[30], x2[30];
    Mat A1(30, 30), B1(30, 30), A2, B2;
    // first stage: the evaluation at (x, u) above already carries all sixty directions
    for (int i = 0; i < 30; ++i) { k1[i] = fd[i].v; for (int j = 0; j < 30; ++j) { A1(i, j) = fd[i].d[j]; B1(i, j) = fd[i].d[30 + j]; } }
    for (int i = 0; i < 30; ++i) x2[i] = x[i] + dt * k1[i];
    flowMapLinearization(P, x2, u, k2, A2, B2);
    B2 = B2 + dt * (A2 * B1);
    A2 = A2 + dt * (A2 * A1);
    o.A = 0.5 * dt * A1 + 0.5 * dt * A2;
    for (int i = 0; i < 30; ++i) o.A(i, i) += 1.0;
    o.B = 0.5 * dt * B1 + 0.5 * dt * B2;
    o.b = Vec(30);
    for (int i = 0; i < 30; ++i) o.b[i] = x[i] + 0.5 * dt * (k1[i] + k2[i]) - xnext[i];
  }
  // ---- scale cost by dt
  o.cost *= dt; o.Q = dt * o.Q; o.R = dt * o.R; o.q = dt * o.q; o.r = dt * o.r;
}

// upstream LinearAlgebra::qrConstraintProjection + multiple_shooting::projectTranscription / changeOfInputVariables
inline void projectNode(NodeLQ& o) {
  const int m = 30, nc = o.nc;
  if (nc == 0) {
    o.Pu = Mat::identity(m); o.Px = Mat(m, 30); o.Pe = Vec(m, 0.0);
  } else {
    Mat Qh, Rh;
    householderQR(T(o.D), Qh, Rh);  // D^T = Q [R1; 0]
    const Mat Q1 = block(Qh, 0, 0, m, nc);
    o.Pu = block(Qh, 0, nc, m, m - nc);
    // pseudoInverse^T = Q1 R1^-T ... solve R1^T Y = [C e]  then  Px = -Q1 Y
    Mat rhs(nc, 31);
    for (int i = 0; i < nc; ++i) { for (int j = 0; j < 30; ++j) rhs(i, j) = o.C(i, j); rhs(i, 30) = o.e[i]; }
    for (int col = 0; col < 31; ++col)
      for (int i = 0; i < nc; ++i) { double s = rhs(i, col); for (int k = 0; k < i; ++k) s -= Rh(k, i) * rhs(k, col); rhs(i, col) = s / Rh(i, i); }
    const Mat PxPe = -1.0 * (Q1 * rhs);
    o.Px = block(PxPe, 0, 0, m, 30);
    o.Pe = Vec(m); for (int i = 0; i < m; ++i) o.Pe[i] = PxPe(i, 30);
  }
  // dynamics
  o.At = o.A + o.B * o.Px;
  o.bt = o.b + o.B * o.Pe;
  o.Bt = o.B * o.Pu;
  // cost (P = 0 before projection for this problem but kept general)
  const Vec RPe = o.R * o.Pe;
  const Mat RPx = o.R * o.Px;
  const Mat PuT = T(o.Pu), PxT = T(o.Px);
  o.qt = o.q + PxT * (o.r + RPe) + tmul(o.Pm, o.Pe);
  o.rt = PuT * (o.r + RPe);
  o.Qt = o.Q + PxT * o.Pm + T(o.Pm) * o.Px + PxT * RPx;
  o.Pt = PuT * (o.Pm + RPx);
  o.Rt = PuT * o.R * o.Pu;
}

struct SqpResult {
  std::vector<double> X, U;  // new iterate
  double merit0 = 0, viol0 = 0, merit1 = 0, viol1 = 0, alpha = 0, armijo = 0;
  int stepType = 0;  // 0 unknown, 1 constraint, 2 dual, 3 cost, 4 zero
  int status = 0;
  double dxNorm = 0, duNorm = 0;  // l2 norms of the step directions over the whole horizon (upstream trajectoryNorm)
};

// upstream SqpSolver::checkConvergence: 0 not converged, 1 iteration limit, 2 step size, 3 metrics, 4 primal step
inline int sqpConvergence(const qmgpu_settings& st, int iteration, const SqpResult& r) {
  if (iteration + 1 >= st.sqp_iterations) return 1;
  if (r.alpha < st.alpha_min) return 2;
  if (std::fabs(r.merit1 - r.merit0) < st.cost_tol && r.viol1 < st.g_min) return 3;
  if (r.alpha * r.dxNorm < st.delta_tol && r.alpha * r.duNorm < st.delta_tol) return 4;
  return 0;
}

// what the DDP variant takes from the LQ approximation + Riccati recursion of one iterate
struct RiccatiExport { std::vector<NodeLQ> lq; std::vector<Mat> K; std::vector<Vec> k; double armijo = 0.0; int status = 0; };

// One SQP iteration over the grid tgrid[0..N]; X [(N+1)*30], U [N*30] hold the initial guess.  With `riccatiOut` the function stops after the
// Riccati recursion and hands back the projected stages and gains (no line search).
inline SqpResult sqpIteration(const Problem& pr, int N, const double* tgrid, const double* x0, const std::vector<double>& X, const std::vector<double>& U,
                              bool lineSearch, std::vector<NodeLQ>* keepLQ = nullptr, RiccatiExport* riccatiOut = nullptr) {
  const qmgpu_settings& st = pr.P->settings;
  std::vector<NodeLQ> lq(N + 1);
  std::unique_ptr<PhaseTimer> phase(new PhaseTimer(PH_LQ));
  forEachNode(N, pr.nodeThreads, [&](int k) { nodeLQ(pr, tgrid[k], tgrid[k + 1] - tgrid[k], &X[k * 30], &U[k * 30], &X[(k + 1) * 30], false, lq[k]); projectNode(lq[k]); });
  nodeLQ(pr, tgrid[N], 0.0, &X[N * 30], nullptr, nullptr, true, lq[N]);

  phase.reset(); phase.reset(new PhaseTimer(PH_RICCATI));
  // ---- Riccati backward (the equality-free OCP-QP HPIPM solves with one factorisation)
  std::vector<Mat> Kfb(N); std::vector<Vec> kff(N);
  Mat S = lq[N].Q; Vec s = lq[N].q;
  for (int k = N - 1; k >= 0; --k) {
    const NodeLQ& n = lq[k];
    const Mat BtS = T(n.Bt) * S;
    Mat H = n.Rt + BtS * n.Bt;
    const Mat G = n.Pt + BtS * n.At;
    const Vec Sb = S * n.bt;
    const Vec g = n.rt + tmul(n.Bt, s + Sb);
    Mat L = H;
    if (!cholesky(L)) { SqpResult r; r.status = 1; r.X = X; r.U = U; if (riccatiOut) riccatiOut->status = 1; return r; }
    Mat Kk = -1.0 * cholSolve(L, G);
    Vec kk = g; cholSolve(L, kk); kk = -1.0 * kk;
    const Mat AtS = T(n.At) * S;
    Mat Sn = n.Qt + AtS * n.At + T(G) * Kk;
    for (int i = 0; i < 30; ++i) for (int j = i + 1; j < 30; ++j) { const double a = 0.5 * (Sn(i, j) + Sn(j, i)); Sn(i, j) = Sn(j, i) = a; }
    const Vec sn = n.qt + tmul(n.At, s + Sb) + tmul(G, kk);
    S = Sn; s = sn; Kfb[k] = Kk; kff[k] = kk;
  }
  // ---- forward
  std::vector<double> dX((N + 1) * 30, 0.0), dU(N * 30, 0.0);
  Vec dx(30); for (int i = 0; i < 30; ++i) dx[i] = x0[i] - X[i];
  double armijo = 0.0;
  for (int k = 0; k < N; ++k) {
    const NodeLQ& n = lq[k];
    for (int i = 0; i < 30; ++i) dX[k * 30 + i] = dx[i];
    const Vec dut = Kfb[k] * dx + kff[k];
    armijo += dot(n.qt, dx) + dot(n.rt, dut);
    const Vec du = n.Pe + n.Px * dx + n.Pu * dut;
    for (int i = 0; i < 30; ++i) dU[k * 30 + i] = du[i];
    dx = n.At * dx + n.Bt * dut + n.bt;
  }
  for (int i = 0; i < 30; ++i) dX[N * 30 + i] = dx[i];
  armijo += dot(lq[N].q, dx);
  if (riccatiOut) { riccatiOut->lq = lq; riccatiOut->K = Kfb; riccatiOut->k = kff; riccatiOut->armijo = armijo; SqpResult r; r.armijo = armijo; r.X = X; r.U = U; return r; }

  phase.reset(); phase.reset(new PhaseTimer(PH_LINESEARCH));
  // ---- performance of the baseline and filter line search (upstream FilterLinesearch::acceptStep, SqpSolver::takeStep)
  auto performance = [&](const std::vector<double>& Xn, const std::vector<double>& Un, double& merit, double& viol) {
    double cost = 0, dyn = 0, eq = 0;
    for (int i = 0; i < 30; ++i) dyn += (x0[i] - Xn[i]) * (x0[i] - Xn[i]);  // initial-state gap
    std::vector<NodeMetrics> nm(N);
    forEachNode(N, pr.nodeThreads, [&](int k) { nm[k] = nodeMetrics(pr, tgrid[k], tgrid[k + 1] - tgrid[k], &Xn[k * 30], &Un[k * 30], &Xn[(k + 1) * 30], false); });
    for (int k = 0; k < N; ++k) { cost += nm[k].cost; dyn += nm[k].dynViolationSSE; eq += nm[k].eqViolationSSE; }   // summed in node order: independent of the thread count
    cost += nodeMetrics(pr, tgrid[N], 0.0, &Xn[N * 30], nullptr, nullptr, true).cost;
    merit = cost; viol = std::sqrt(dyn + eq);
  };
  SqpResult res; res.armijo = armijo;
  { double sx = 0, su = 0; for (double v : dX) sx += v * v; for (double v : dU) su += v * v; res.dxNorm = std::sqrt(sx); res.duNorm = std::sqrt(su); }
  performance(X, U, res.merit0, res.viol0);
  double alpha = 1.0;
  std::vector<double> Xn(X.size()), Un(U.size());
  do {
    for (size_t i = 0; i < X.size(); ++i) Xn[i] = X[i] + alpha * dX[i];
    for (size_t i = 0; i < U.size(); ++i) Un[i] = U[i] + alpha * dU[i];
    performance(Xn, Un, res.merit1, res.viol1);
    bool accepted;
    if (!lineSearch) { accepted = true; res.stepType = 0; }
    else if (res.viol1 > st.g_max) { accepted = res.viol1 < (1.0 - st.gamma_c) * res.viol0; res.stepType = 1; }
    else if (res.viol1 < st.g_min && res.viol0 < st.g_min && alpha * armijo < 0.0) { accepted = res.merit1 < res.merit0 + st.armijo_factor * alpha * armijo; res.stepType = 3; }
    else { accepted = res.merit1 < res.merit0 - st.gamma_c * res.viol0 || res.viol1 < (1.0 - st.gamma_c) * res.viol0; res.stepType = 2; }
    if (accepted) { res.alpha = alpha; res.X = Xn; res.U = Un; if (keepLQ) *keepLQ = lq; return res; }
    alpha *= st.alpha_decay;
  } while (alpha >= st.alpha_min);
  res.alpha = 0.0; res.stepType = 4; res.X = X; res.U = U; res.merit1 = res.merit0; res.viol1 = res.viol0;
  if (keepLQ) *keepLQ = lq;
  return res;
}

// ------------------------------------------------------------------------------------------------ DDP variant
// Single-shooting DDP of the family the reference's task file configures in ddp{} (task.info:34-72, never instantiated by the reference):
// own restatement, see include/qmgpu.h QMGPU_ALG_DDP for the stated deviations from upstream's SLQ (RK2 rollout on the shooting grid,
// discrete-time backward pass).
struct DdpResult { std::vector<double> X, U; double merit0 = 0, eq0 = 0, merit1 = 0, eq1 = 0, alpha = 0, armijo = 0; int trials = 0, status = 0; };

inline DdpResult ddpIteration(const Problem& pr, int N, const double* tgrid, const double* x0, const std::vector<double>& Uinit, const double* Xwarm = nullptr) {
  const qmgpu_settings& st = pr.P->settings;
  DdpResult res;
  // 1. nominal trajectory: the warm start as it is (states and inputs of the previous solve; its defects are part of the linearisation, as in
  //    Gauss-Newton multiple shooting), or the open-loop rollout of the given inputs from x0 when no states are given
  std::vector<double> X((N + 1) * 30), U = Uinit;
  for (int i = 0; i < 30; ++i) X[i] = x0[i];
  if (Xwarm) { for (int k = 1; k <= N; ++k) for (int i = 0; i < 30; ++i) X[k * 30 + i] = Xwarm[k * 30 + i]; }
  else for (int k = 0; k < N; ++k) { const ContactScope contact(st, pr.tg, tgrid[k]); rk2Step(*pr.P, tgrid[k + 1] - tgrid[k], &X[k * 30], &U[k * 30], &X[(k + 1) * 30]); }
  // 2. LQ approximation along it + projected Riccati recursion
  RiccatiExport rx;
  sqpIteration(pr, N, tgrid, x0, X, U, false, nullptr, &rx);
  res.armijo = rx.armijo; res.status = rx.status; res.X = X; res.U = U;
  // merit of a trajectory: dt-scaled costs + penalty * dt (|eq|^2 + |defect|^2); a rolled-out trajectory has no defect, a warm start may
  auto meritOf = [&](const std::vector<double>& Xn, const std::vector<double>& Un, double& eqOut) {
    double cost = 0, eq = 0;
    for (int k = 0; k < N; ++k) { const NodeMetrics m = nodeMetrics(pr, tgrid[k], tgrid[k + 1] - tgrid[k], &Xn[k * 30], &Un[k * 30], &Xn[(k + 1) * 30], false); cost += m.cost; eq += m.eqViolationSSE + m.dynViolationSSE; }
    cost += nodeMetrics(pr, tgrid[N], 0.0, &Xn[N * 30], nullptr, nullptr, true).cost;
    eqOut = eq;
    return cost + st.ddp_constraint_penalty * eq;
  };
  res.merit0 = meritOf(X, U, res.eq0); res.merit1 = res.merit0; res.eq1 = res.eq0;
  if (rx.status != 0) return res;
  // 3. policy rollouts, first step length that passes the Armijo test
  for (double alpha = st.ddp_max_step; alpha >= st.ddp_min_step && res.trials < 8; alpha *= 0.5) {
    ++res.trials;
    std::vector<double> Xn((N + 1) * 30), Un(N * 30);
    for (int i = 0; i < 30; ++i) Xn[i] = x0[i];
    for (int k = 0; k < N; ++k) {
      const NodeLQ& n = rx.lq[k];
      Vec dx(30); for (int i = 0; i < 30; ++i) dx[i] = Xn[k * 30 + i] - X[k * 30 + i];
      const Vec dut = rx.K[k] * dx + alpha * rx.k[k];
      const Vec du = alpha * n.Pe + n.Px * dx + n.Pu * dut;
      for (int i = 0; i < 30; ++i) Un[k * 30 + i] = U[k * 30 + i] + du[i];
      const ContactScope contact(st, pr.tg, tgrid[k]);
      rk2Step(*pr.P, tgrid[k + 1] - tgrid[k], &Xn[k * 30], &Un[k * 30], &Xn[(k + 1) * 30]);
    }
    double eq1;
    const double m1 = meritOf(Xn, Un, eq1);
    if (m1 == m1 && m1 <= res.merit0 - st.armijo_factor * alpha * std::fabs(rx.armijo)) { res.alpha = alpha; res.merit1 = m1; res.eq1 = eq1; res.X = Xn; res.U = Un; return res; }
  }
  return res;
}

}  // namespace qmo
