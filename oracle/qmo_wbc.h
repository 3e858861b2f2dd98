// qmo_wbc.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; parity unpinned, see qmo_core.h).
//
// Whole-body controller: restates qm_wbc/src/WbcBase.cpp (model update + 13 task builders),
// qm_wbc/include/qm_wbc/Task.h, qm_wbc/src/HoQp.cpp (hierarchical null-space QP cascade) and
// qm_wbc/src/Hierarchical{Wbc,MpcWbc}.cpp.  The QP itself (qpOASES in the reference, HoQp.cpp:135-150,
// un-vendored) is solved by a dense primal-dual interior point method: the reference fixes only the QP
// data and that the unique minimiser is returned, not the algorithm.
#pragma once
#include "qmo_mpc.h"

namespace qmo {

using D1 = Dual<1>;

// ------------------------------------------------------------------------------------------------ rotations (upstream ocs2_robotic_tools)
inline M3<double> rotZyx(const double e[3]) { return axisRotation<double>(2, e[0]) * axisRotation<double>(1, e[1]) * axisRotation<double>(0, e[2]); }
// getEulerAnglesZyxDerivativesFromGlobalAngularVelocity
inline void eulerRatesFromGlobalAngVel(const double e[3], const double w[3], double out[3]) {
  const double sz = std::sin(e[0]), cz = std::cos(e[0]), sy = std::sin(e[1]), cy = std::cos(e[1]);
  const double tmp = cz * w[0] / cy + sz * w[1] / cy;
  out[0] = sy * tmp + w[2]; out[1] = -sz * w[0] + cz * w[1]; out[2] = tmp;
}
// getGlobalAngularVelocityFromEulerAnglesZyxDerivatives
inline void globalAngVelFromEulerRates(const double e[3], const double de[3], double w[3]) {
  const double sz = std::sin(e[0]), cz = std::cos(e[0]), sy = std::sin(e[1]), cy = std::cos(e[1]);
  w[0] = -sz * de[1] + cy * cz * de[2];
  w[1] = cz * de[1] + cy * sz * de[2];
  w[2] = de[0] - sy * de[2];
}
// getGlobalAngularAccelerationFromEulerAnglesZyxDerivatives: d/dt of the map above
inline void globalAngAccFromEulerRates(const double e[3], const double de[3], const double dde[3], double a[3]) {
  const double sz = std::sin(e[0]), cz = std::cos(e[0]), sy = std::sin(e[1]), cy = std::cos(e[1]);
  const double sz_t = cz * de[0], cz_t = -sz * de[0], sy_t = cy * de[1], cy_t = -sy * de[1];
  a[0] = -sz * dde[1] + cy * cz * dde[2] - sz_t * de[1] + (cy_t * cz + cy * cz_t) * de[2];
  a[1] = cz * dde[1] + cy * sz * dde[2] + cz_t * de[1] + (cy_t * sz + cy * sz_t) * de[2];
  a[2] = dde[0] - sy * dde[2] - sy_t * de[2];
}
// rotationMatrixToRotationVector / rotationErrorInWorld(lhs, rhs) = log(lhs * rhs^T)
inline V3<double> rotationVector(const M3<double>& R) {
  const double tr = R.m[0][0] + R.m[1][1] + R.m[2][2];
  const V3<double> skew(R.m[2][1] - R.m[1][2], R.m[0][2] - R.m[2][0], R.m[1][0] - R.m[0][1]);
  const double c = std::max(-1.0, std::min(1.0, 0.5 * (tr - 1.0)));
  const double theta = std::acos(c);
  double k;
  if (theta < 1e-4) k = 0.5 + theta * theta / 12.0;  // series of theta / (2 sin theta)
  else k = 0.5 * theta / std::sin(theta);
  return k * skew;
}
inline V3<double> rotationErrorInWorld(const M3<double>& lhs, const M3<double>& rhs) { return rotationVector(lhs * transpose(rhs)); }

// ------------------------------------------------------------------------------------------------ model update
struct WbcModel {
  Vec qM, vM, qD, vD, baseAccDesired;
  Mat M, J, dJ, baseJ, baseDJ, armJ, armDJ;
  Vec nle;
  V3<double> footPosM[4], footVelM[4], footPosD[4], footVelD[4];
  V3<double> eePosM, eeVelM, eePosD, eeVelD, eeAngVelM, eeAngVelD;
  M3<double> eeRotM, eeRotD;
};

// 6 x 24 LOCAL_WORLD_ALIGNED frame Jacobian and its time variation along v (dual number in time).
inline void frameJacobian(const qmgpu_model& md, const double* q, const double* v, int body, const double off[3], Mat& J, Mat& dJ) {
  D1 qd[NV];
  for (int i = 0; i < NV; ++i) { qd[i] = D1(q[i]); qd[i].d[0] = v[i]; }
  Kin<D1> k;
  forwardKinematics<D1>(md, qd, k);
  const V3<D1> r = k.p[body] + k.R[body] * V3<D1>(D1(off[0]), D1(off[1]), D1(off[2]));
  J = Mat(6, NV); dJ = Mat(6, NV);
  for (int d = 0; d < NV; ++d) {
    V3<D1> lin, ang;
    pointJacobianColumn(md, k, body, r, d, lin, ang);
    for (int a = 0; a < 3; ++a) { J(a, d) = lin[a].v; dJ(a, d) = lin[a].d[0]; J(3 + a, d) = ang[a].v; dJ(3 + a, d) = ang[a].d[0]; }
  }
}

inline void wbcUpdateMeasured(const qmgpu_problem& P, const double* rbd, WbcModel& w) {
  const qmgpu_model& md = P.model;
  w.qM = Vec(NV, 0.0); w.vM = Vec(NV, 0.0);
  // WbcBase.cpp:150-156
  for (int i = 0; i < 3; ++i) { w.qM[i] = rbd[3 + i]; w.qM[3 + i] = rbd[i]; }
  for (int j = 0; j < NJ; ++j) w.qM[6 + j] = rbd[6 + j];
  for (int i = 0; i < 3; ++i) w.vM[i] = rbd[NV + 3 + i];
  eulerRatesFromGlobalAngVel(&w.qM[3], rbd + NV, &w.vM[3]);
  for (int j = 0; j < NJ; ++j) w.vM[6 + j] = rbd[NV + 6 + j];

  // time-dual kinematics: q(t) = q + t v
  D1 qd[NV];
  for (int i = 0; i < NV; ++i) { qd[i] = D1(w.qM[i]); qd[i].d[0] = w.vM[i]; }
  Kin<D1> k;
  forwardKinematics<D1>(md, qd, k);
  // mass matrix (crba, symmetrised: WbcBase.cpp:165-167) and nonlinear effects (WbcBase.cpp:170) by projecting
  // the inertial + gravity wrench of every body on its Jacobian (d'Alembert), bias accelerations from the dual part
  w.M = Mat(NV, NV); w.nle = Vec(NV, 0.0);
  for (int b = 0; b < NB; ++b) {
    V3<D1> lin[NV], ang[NV];
    V3<D1> vc, om;
    for (int d = 0; d < NV; ++d) { pointJacobianColumn(md, k, b, k.com[b], d, lin[d], ang[d]); vc = vc + D1(w.vM[d]) * lin[d]; om = om + D1(w.vM[d]) * ang[d]; }
    // bias accelerations: d/dt (J(q(t)) v) with v constant
    const V3<double> ac(vc.x.d[0], vc.y.d[0], vc.z.d[0]), al(om.x.d[0], om.y.d[0], om.z.d[0]);
    const V3<double> omv(om.x.v, om.y.v, om.z.v);
    M3<double> Iw; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Iw.m[i][j] = k.Iw[b].m[i][j].v;
    const V3<double> force = md.mass[b] * (ac + V3<double>(0, 0, P.settings.gravity));
    const V3<double> torque = Iw * al + cross(omv, Iw * omv);
    for (int i = 0; i < NV; ++i) {
      const V3<double> li(lin[i].x.v, lin[i].y.v, lin[i].z.v), ai(ang[i].x.v, ang[i].y.v, ang[i].z.v);
      w.nle[i] += dot(li, force) + dot(ai, torque);
      const V3<double> Iai = Iw * ai;
      for (int j = 0; j < NV; ++j) {
        const V3<double> lj(lin[j].x.v, lin[j].y.v, lin[j].z.v), aj(ang[j].x.v, ang[j].y.v, ang[j].z.v);
        w.M(i, j) += md.mass[b] * dot(li, lj) + dot(Iai, aj);
      }
    }
  }
  // feet Jacobians (contact order) and their time variation (WbcBase.cpp:171-187)
  w.J = Mat(12, NV); w.dJ = Mat(12, NV);
  for (int c = 0; c < 4; ++c) {
    Mat Jf, dJf;
    frameJacobian(md, w.qM.data(), w.vM.data(), md.foot_body[c], md.foot_offset[c], Jf, dJf);
    for (int a = 0; a < 3; ++a) for (int d = 0; d < NV; ++d) { w.J(3 * c + a, d) = Jf(a, d); w.dJ(3 * c + a, d) = dJf(a, d); }
    V3<double> vel;
    for (int a = 0; a < 3; ++a) { double s = 0; for (int d = 0; d < NV; ++d) s += Jf(a, d) * w.vM[d]; vel[a] = s; }
    w.footVelM[c] = vel;
    w.footPosM[c] = V3<double>(k.foot[c].x.v, k.foot[c].y.v, k.foot[c].z.v);
  }
  const double zero[3] = {0, 0, 0};
  frameJacobian(md, w.qM.data(), w.vM.data(), 0, zero, w.baseJ, w.baseDJ);                      // WbcBase.cpp:190-196
  frameJacobian(md, w.qM.data(), w.vM.data(), md.ee_body, md.ee_offset, w.armJ, w.armDJ);       // WbcBase.cpp:199-202
  w.eePosM = V3<double>(k.ee.x.v, k.ee.y.v, k.ee.z.v);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) w.eeRotM.m[i][j] = k.Ree.m[i][j].v;
  for (int a = 0; a < 3; ++a) { double sl = 0, sa = 0; for (int d = 0; d < NV; ++d) { sl += w.armJ(a, d) * w.vM[d]; sa += w.armJ(3 + a, d) * w.vM[d]; } w.eeVelM[a] = sl; w.eeAngVelM[a] = sa; }
}

inline void wbcUpdateDesired(const qmgpu_problem& P, const double* xDes, const double* uDes, double* inputLast, double period, WbcModel& w, const double* eeForce = nullptr) {
  const qmgpu_model& md = P.model;
  w.qD = Vec(xDes + 6, xDes + 30);
  // v_des from the centroidal map (WbcBase.cpp:217-219)
  Kin<double> k;
  forwardKinematics<double>(md, w.qD.data(), k);
  static thread_local double A[6][NV];
  centroidalMomentumMatrix(md, k, A);
  double vb[6];
  baseVelocityFromMomentum(md, A, xDes, uDes + 12, vb);
  w.vD = Vec(NV);
  for (int a = 0; a < 6; ++a) w.vD[a] = vb[a];
  for (int j = 0; j < NJ; ++j) w.vD[6 + j] = uDes[12 + j];
  // joint accelerations by finite difference of the commanded joint velocities (WbcBase.cpp:224-225, stateful)
  Vec jointAccel(NJ);
  for (int j = 0; j < NJ; ++j) jointAccel[j] = (uDes[12 + j] - inputLast[12 + j]) / period;
  for (int i = 0; i < 30; ++i) inputLast[i] = uDes[i];
  // dA/dt * v (pinocchio::dccrba) from the time-dual of A_G along v_des
  D1 qd[NV];
  for (int i = 0; i < NV; ++i) { qd[i] = D1(w.qD[i]); qd[i].d[0] = w.vD[i]; }
  Kin<D1> kd;
  forwardKinematics<D1>(md, qd, kd);
  static thread_local D1 Ad[6][NV];
  centroidalMomentumMatrix(md, kd, Ad);
  // m * normalized momentum rate (WbcBase.cpp:232) - Adot v - Aj qdd_j (WbcBase.cpp:233-234)
  double rate[6] = {0, 0, -md.total_mass * P.settings.gravity, 0, 0, 0};
  for (int c = 0; c < 4; ++c) {
    const V3<double> f(uDes[3 * c], uDes[3 * c + 1], uDes[3 * c + 2]);
    const V3<double> t = cross(k.foot[c] - k.comTotal, f);
    for (int a = 0; a < 3; ++a) { rate[a] += f[a]; rate[3 + a] += t[a]; }
  }
  if (eeForce) {   // force tracking (own formulation): the external end-effector force is part of the desired momentum rate
    const V3<double> f(eeForce[0], eeForce[1], eeForce[2]);
    const V3<double> t = cross(k.ee - k.comTotal, f);
    for (int a = 0; a < 3; ++a) { rate[a] += f[a]; rate[3 + a] += t[a]; }
  }
  for (int a = 0; a < 6; ++a) {
    for (int d = 0; d < NV; ++d) rate[a] -= Ad[a][d].d[0] * w.vD[d];
    for (int j = 0; j < NJ; ++j) rate[a] -= A[a][6 + j] * jointAccel[j];
  }
  // AbInv * rate (same block inverse)
  {
    M3<double> Ab22, Ab12;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Ab22.m[i][j] = A[3 + i][3 + j]; Ab12.m[i][j] = A[i][3 + j]; }
    const V3<double> wv = inverse3(Ab22) * V3<double>(rate[3], rate[4], rate[5]);
    const V3<double> t = Ab12 * wv;
    w.baseAccDesired = Vec(6);
    for (int a = 0; a < 3; ++a) { w.baseAccDesired[a] = (rate[a] - t[a]) / md.total_mass; w.baseAccDesired[3 + a] = wv[a]; }
  }
  // desired feet / EE kinematics (forwardKinematics(qDesired, vDesired), WbcBase.cpp:221)
  for (int c = 0; c < 4; ++c) {
    w.footPosD[c] = k.foot[c];
    V3<double> vel;
    for (int d = 0; d < NV; ++d) { V3<double> lin, ang; pointJacobianColumn(md, k, md.foot_body[c], k.foot[c], d, lin, ang); vel = vel + w.vD[d] * lin; }
    w.footVelD[c] = vel;
  }
  w.eePosD = k.ee; w.eeRotD = k.Ree;
  V3<double> vel, angv;
  for (int d = 0; d < NV; ++d) { V3<double> lin, ang; pointJacobianColumn(md, k, md.ee_body, k.ee, d, lin, ang); vel = vel + w.vD[d] * lin; angv = angv + w.vD[d] * ang; }
  w.eeVelD = vel; w.eeAngVelD = angv;
}

// ------------------------------------------------------------------------------------------------ tasks (Task.h)
struct Task {
  Mat a, d; Vec b, f;
  Task() = default;
  Task(Mat a_, Vec b_, Mat d_, Vec f_) : a(std::move(a_)), d(std::move(d_)), b(std::move(b_)), f(std::move(f_)) {}
  Task operator+(const Task& r) const { return Task(vstack(a, r.a), vcat(b, r.b), vstack(d, r.d), vcat(f, r.f)); }
  Task operator*(double s) const { Task t = *this; t.a = s * t.a; t.b = s * t.b; t.d = s * t.d; t.f = s * t.f; return t; }
};

struct WbcTasks {
  const qmgpu_problem& P; const WbcModel& w; bool contact[4]; int numContacts;
  static constexpr int ND = QMGPU_NWBC_DEC;
  WbcTasks(const qmgpu_problem& P_, const WbcModel& w_, int mode) : P(P_), w(w_) { modeToContactFlags(mode, contact); numContacts = 0; for (bool c : contact) numContacts += c; }
  double Jv(const Mat& J, int row) const { double s = 0; for (int d = 0; d < NV; ++d) s += J(row, d) * w.vM[d]; return s; }

  Task floatingBaseEom() const {  // WbcBase.cpp:370-388
    Mat a(6, ND); Vec b(6);
    for (int i = 0; i < 6; ++i) { for (int j = 0; j < NV; ++j) a(i, j) = w.M(i, j); for (int j = 0; j < 12; ++j) a(i, NV + j) = -w.J(j, i); b[i] = -w.nle[i]; }
    return Task(a, b, Mat(), Vec());
  }
  Task torqueLimits() const {  // WbcBase.cpp:392-415 (LF leg limits reused for every leg: WbcBase.cpp:599-600)
    Mat d(2 * NJ, ND); Vec f(2 * NJ);
    for (int i = 0; i < NJ; ++i) {
      for (int j = 0; j < NV; ++j) { d(i, j) = w.M(6 + i, j); d(NJ + i, j) = -w.M(6 + i, j); }
      for (int j = 0; j < 12; ++j) { d(i, NV + j) = -w.J(j, 6 + i); d(NJ + i, NV + j) = w.J(j, 6 + i); }
      const double lim = i < 12 ? P.model.effort_limit[i % 3] : P.model.effort_limit[i];
      f[i] = lim - w.nle[6 + i]; f[NJ + i] = lim + w.nle[6 + i];
    }
    return Task(Mat(), Vec(), d, f);
  }
  Task noContactMotion() const {  // WbcBase.cpp:418-433
    Mat a(3 * numContacts, ND); Vec b(3 * numContacts); int j = 0;
    for (int c = 0; c < 4; ++c) if (contact[c]) { for (int r = 0; r < 3; ++r) { for (int d = 0; d < NV; ++d) a(3 * j + r, d) = w.J(3 * c + r, d); b[3 * j + r] = -Jv(w.dJ, 3 * c + r); } ++j; }
    return Task(a, b, Mat(), Vec());
  }
  Task frictionCone() const {  // WbcBase.cpp:439-469 (keeps the 3*n_sw all-zero inequality rows)
    const int nsw = 4 - numContacts; const double mu = P.settings.wbc_friction_coefficient;
    Mat a(3 * nsw, ND); int j = 0;
    for (int c = 0; c < 4; ++c) if (!contact[c]) { for (int r = 0; r < 3; ++r) a(3 * j + r, NV + 3 * c + r) = 1.0; ++j; }
    Vec b(3 * nsw, 0.0);
    const double pyr[5][3] = {{0, 0, -1}, {1, 0, -mu}, {-1, 0, -mu}, {0, 1, -mu}, {0, -1, -mu}};
    Mat d(5 * numContacts + 3 * nsw, ND); j = 0;
    for (int c = 0; c < 4; ++c) if (contact[c]) { for (int r = 0; r < 5; ++r) for (int q = 0; q < 3; ++q) d(5 * j + r, NV + 3 * c + q) = pyr[r][q]; ++j; }
    return Task(a, b, d, Vec(d.r, 0.0));
  }
  Task baseHeight() const {  // WbcBase.cpp:308-320
    Mat a(1, ND); a(0, 2) = 1.0;
    Vec b(1); b[0] = w.baseAccDesired[2] + P.settings.kp_base_height * (w.qD[2] - w.qM[2]) + P.settings.kd_base_height * (w.vD[2] - w.vM[2]);
    return Task(a, b, Mat(), Vec());
  }
  Task baseLinear() const {  // WbcBase.cpp:240-252
    Mat a(2, ND); a(0, 0) = 1.0; a(1, 1) = 1.0; Vec b(2);
    for (int i = 0; i < 2; ++i) b[i] = w.baseAccDesired[i] + P.settings.kp_base_linear * (w.qD[i] - w.qM[i]) + P.settings.kd_base_linear * (w.vD[i] - w.vM[i]);
    return Task(a, b, Mat(), Vec());
  }
  Task baseXYLinearAccel() const {  // WbcBase.cpp:255-267 (defined, unused by either controller)
    Mat a(2, ND); a(0, 0) = 1.0; a(1, 1) = 1.0; Vec b(2); b[0] = w.baseAccDesired[0]; b[1] = w.baseAccDesired[1];
    return Task(a, b, Mat(), Vec());
  }
  Task baseAngular() const {  // WbcBase.cpp:270-305
    Mat a(3, ND); Vec b(3);
    for (int r = 0; r < 3; ++r) for (int d = 0; d < NV; ++d) a(r, d) = w.baseJ(3 + r, d);
    const double* eul = &w.qM[3];
    double wM[3], wD[3], acc[3];
    globalAngVelFromEulerRates(eul, &w.vM[3], wM);
    globalAngVelFromEulerRates(eul, &w.vD[3], wD);
    const V3<double> err = rotationErrorInWorld(rotZyx(&w.qD[3]), rotZyx(eul));
    globalAngAccFromEulerRates(eul, &w.vD[3], &w.baseAccDesired[3], acc);
    for (int r = 0; r < 3; ++r) b[r] = acc[r] + P.settings.kp_base_angular * err[r] + P.settings.kd_base_angular * (wD[r] - wM[r]) - Jv(w.baseDJ, 3 + r);
    return Task(a, b, Mat(), Vec());
  }
  Task swingLeg() const {  // WbcBase.cpp:323-346
    const int nsw = 4 - numContacts; Mat a(3 * nsw, ND); Vec b(3 * nsw); int j = 0;
    for (int c = 0; c < 4; ++c) if (!contact[c]) {
      for (int r = 0; r < 3; ++r) {
        const double acc = P.settings.kp_swing * (w.footPosD[c][r] - w.footPosM[c][r]) + P.settings.kd_swing * (w.footVelD[c][r] - w.footVelM[c][r]);
        for (int d = 0; d < NV; ++d) a(3 * j + r, d) = w.J(3 * c + r, d);
        b[3 * j + r] = acc - Jv(w.dJ, 3 * c + r);
      }
      ++j;
    }
    return Task(a, b, Mat(), Vec());
  }
  Task armJointNominalTracking() const {  // WbcBase.cpp:471-497
    Mat a(6, ND); Vec b(6);
    for (int i = 0; i < 6; ++i) { a(i, NV - 6 + i) = 1.0; b[i] = P.settings.kp_arm_joint[i] * (w.qD[NV - 6 + i] - w.qM[NV - 6 + i]) + P.settings.kd_arm_joint[i] * (w.vD[NV - 6 + i] - w.vM[NV - 6 + i]); }
    return Task(a, b, Mat(), Vec());
  }
  Task eeLinear() const {  // WbcBase.cpp:499-524
    Mat a(3, ND); Vec b(3);
    for (int r = 0; r < 3; ++r) {
      for (int d = 0; d < NV; ++d) a(r, d) = w.armJ(r, d);
      b[r] = P.settings.kp_ee_linear[r] * (w.eePosD[r] - w.eePosM[r]) + P.settings.kd_ee_linear[r] * (w.eeVelD[r] - w.eeVelM[r]) - Jv(w.armDJ, r);
    }
    return Task(a, b, Mat(), Vec());
  }
  Task eeAngular() const {  // WbcBase.cpp:526-563: columns 3..5 of A and of dJ zeroed, desired angular velocity unused
    Mat a(3, ND); Vec b(3);
    const V3<double> err = rotationErrorInWorld(w.eeRotD, w.eeRotM);
    for (int r = 0; r < 3; ++r) {
      double djv = 0;
      for (int d = 0; d < NV; ++d) { if (d >= 3 && d < 6) continue; a(r, d) = w.armJ(3 + r, d); djv += w.armDJ(3 + r, d) * w.vM[d]; }
      b[r] = P.settings.kp_ee_angular[r] * err[r] + P.settings.kd_ee_angular[r] * (-w.eeAngVelM[r]) - djv;
    }
    return Task(a, b, Mat(), Vec());
  }
  Task contactForce(const double* uDes) const {  // WbcBase.cpp:566-578
    Mat a(12, ND); Vec b(12);
    for (int i = 0; i < 12; ++i) { a(i, NV + i) = 1.0; b[i] = uDes[i]; }
    return Task(a, b, Mat(), Vec());
  }
};

// ================================================================================================ the QP of one HoQp level (HoQp.cpp:60-150)
// The reference hands   min 1/2 [z;v]' blkdiag(AZ'AZ + 1e-12 I, I) [z;v] + c'[z;v]   s.t.  D_inh Z z <= f_inh - D_inh x_prev + v_prev,  D_own Z z - v <= f_own - D_own x_prev,  v >= 0
// to qpOASES (an online active-set method: cold start, nWSR = 100, HoQp.cpp:136-149; un-vendored).  Restated here with the slack block eliminated,
//   min 1/2 z'G z + g'z + 1/2 sum_{i own} max(0, D_i z - f_i)^2     s.t.   D_i z <= f_i  (i inherited; f_i >= 0: z = 0 is feasible),
// which is exact (v = max(0, D z - f) at the optimum), and solved by a method of the same class as the reference's:
//   a PRIMAL ACTIVE-SET method ends every level -- a working set, one row changes per iteration, the iterate stays feasible, and the point returned satisfies the KKT
//   conditions on its working set: THE minimiser (unique up to the directions neither a task nor a pinned row sees, which the next level decides);
//   an INTERIOR POINT (Mehrotra) runs in front of it on the levels with inherited rows, only to hand it a starting point and a working set that are usually
//   right already (one active-set iteration then ends the level; cold, from z = 0, it would take one iteration per active row and a zero-length step per
//   zero-margin row the first direction happens to cross -- up to 55 on three-leg stances).  Nothing the interior point returns is final.
// Rows a higher level left STRONGLY active are equalities for every level below (see eliminateImpliedEqualities): they are removed from the problem exactly,
// by a change of variables, before anything else -- the "cone without interior" a low level inherits is solved on its face.
constexpr double kLowerLevelStart = 0.5;     // starting slacks / multipliers of the interior point in units of sqrt(scale): the duality measure starts at 0.25 scale whatever the size of the level's
                                             // gradients and margins.  (Until round 5 the constant 300, tuned on the bench batch whose scale is 7e5; a moving trot's levels have scales of 1e2 .. 1e4 and spent
                                             // four of nine iterations bringing the measure down to where it should have started: 9.3 -> 5.9 passes per tick there, 8.0 on the bench batch either way.  The
                                             // slowest instances are as slow as before: it pays since the WBC launch shares the device with the next cycle's node kernels)
constexpr double kStagnationMu = 1e-10;
constexpr double kEps = 2.220446049250313e-16;
constexpr double kMinNormCheap = 1e4;
constexpr int kIpmMinVariables = 8;
constexpr double kAsLamTol = 8.0;            // a multiplier counts once it exceeds this many roundings of the gradient it balances
constexpr int kAsMaxWorkingSetChanges = 100; // nWSR of HoQp.cpp:141
constexpr int kHeldFormMaxIterations = 4;    // the held-variable form of a level with own rows (solveLevel) is given up beyond this many iterations: status 6, the interior point takes over (= QP_HELD_CAP of the kernels)
constexpr int kAsMaxPinned = 28;             // = QP_KMAX of the kernels (qp_dev.h): rows the small system of the pinned set holds
// Experiment knobs (qmo_set_experiment; defaults = the product's algorithm).  inline: one copy whatever the number of translation units; written between batches only.
inline double g_expLowerLevelStart = kLowerLevelStart;   // another starting value of the interior point = another path to the same vertex (tests: the result must not depend on it)
inline int g_expNoMinNormStart = 0;                      // 1: the first level without its minimum-norm start (tests: same torques)
inline int g_expNoInteriorPoint = 0;                     // 1: the active-set method alone, cold from z = 0 on every level (tests: same vertex)
inline int g_expGuessOrder = 1;
inline int g_expLiteralRegMaxN = 12;                     // levels of at most this many variables keep HoQp's 1e-12 I IN the factorised matrix and the gradient (LevelQp::lit); 0: the limit everywhere, as until round 5
inline int g_expCanonicalFirst = 1;                      // HierarchicalMpcWbc takes the canonical representative at every level from the first pass on (0: only when directions are left over at the end, as until round 6)
inline int g_expOwnInteriorPoint = 3;                    // a level with own rows whose held-variable form is rejected (torque limits that cannot hold) runs the interior point, own rows as penalised slacks, in front of its active-set method (0: cold from z = 0, as until round 6: 40-46 changes on diverged robots)
inline double g_expIpmStartDelta = 0.0;                  // experiment, off: inherited rows of the interior point started on their margins (interiorPointPhase has what it did)
inline int g_expNoWarmStart = 0;                         // 1: the working set carried from the previous tick (wbcUpdate: ws) is ignored -- every level cold (tests: same torques)
inline int g_expTrace = 0;                               // per-iteration trace on stderr

struct QpStats { int ipmIterations = 0, iterations = 0, adds = 0, drops = 0, zeroSteps = 0, innerSteps = 0, eliminated = 0, status = 0; bool minNorm = false, warmTried = false, warmRefuted = false; };   // status: 0 ok | 1 working-set changes exhausted | 2 numerical failure | 3 final check failed

// Cholesky that leaves out the directions without curvature (pivot <= floorv): row / column replaced by the identity, the right-hand side entry by zero, so their step is exactly zero.
// A pivot is a difference, A_jj - sum_k L_jk^2, rounded relative to A_jj: it counts as curvature once it exceeds floorAbs + floorRel (j + 1) A_jj.
inline bool choleskyExcluding(Mat& A, double floorAbs, double floorRel, std::vector<char>& excluded, const std::vector<char>* forced = nullptr) {
  const int n = A.r;
  excluded.assign(n, 0);
  for (int j = 0; j < n; ++j) {
    double d = A(j, j);
    const double floorv = floorAbs + floorRel * double(j + 1) * A(j, j);
    for (int k = 0; k < j; ++k) d -= A(j, k) * A(j, k);
    if (!(d == d)) return false;
    if (!(d > floorv) || (forced && (*forced)[j])) { excluded[j] = 1; for (int k = 0; k < j; ++k) A(j, k) = 0.0; A(j, j) = 1.0; for (int i = j + 1; i < n; ++i) A(i, j) = 0.0; for (int i = 0; i < j; ++i) A(i, j) = 0.0; continue; }
    d = std::sqrt(d);
    A(j, j) = d;
    for (int i = j + 1; i < n; ++i) { double t = A(i, j); for (int k = 0; k < j; ++k) t -= A(i, k) * A(j, k); A(i, j) = t / d; }
    for (int i = 0; i < j; ++i) A(i, j) = 0.0;
  }
  return true;
}
inline void cholSolveExcluding(const Mat& L, const std::vector<char>& excluded, Vec& b) { for (size_t j = 0; j < b.size(); ++j) if (excluded[j]) b[j] = 0.0; cholSolve(L, b); }

// The reduced problem of one level.  G0 = (A Z)'(A Z) WITHOUT HoQp's 1e-12 I (reg, HoQp.cpp:66).  The regulariser is neither part of the factorised matrices nor of
// the gradients: in a direction no task sees it would be the ONLY curvature, 1e-12 z -- the step there would be the rounding of the gradient divided by 1e-12, and a row
// pinned across such a direction would show a multiplier of that size (positive: an equality for the levels below) although the level's cost does not depend on the row
// at all.  The limit reg -> 0 is taken instead: a direction whose pivot does not exceed 10 reg plus its own rounding counts as having no curvature and stays where it is
// (choleskyExcluding), and where directions are left over after the last level the canonical representative is taken (wbcUpdate).
struct LevelQp {
  Mat G0; Vec g; Mat D; Vec f; int mOwn = 0; double reg = 0.0;     // rows [0, mOwn) of D are the level's own (soft), the rest inherited (hard)
  Mat AZ; Vec rhat;      // the level's task in its variables, cost 1/2 |AZ z + rhat|^2 (G0 = AZ'AZ, g = AZ'rhat); empty for a generic QP
  int n() const { return G0.r; } int m() const { return D.r; }
  // gradient of the smooth part.  In residual form when the task is known: AZ'(AZ z + rhat) is rounded relative to the RESIDUAL, G0 z + g relative to |G0| |z| -- two
  // orders of magnitude worse where a weakly seen direction carries a large z (the arm accelerations of HierarchicalMpcWbc, 1e4 rad/s^2 through singular values of 1e-5)
  // lit: the regulariser is kept LITERALLY -- in the factorised matrix and in the gradient -- instead of in the limit.  Round 6 (tools/hoqp_exact.py): the
  // 50-digit solution of the reference's own level QPs shows what the limit costs where a level sees a direction only WEAKLY: HierarchicalMpcWbc's last level (the contact-force
  // task in the six variables the levels above left) sees one combination of arm accelerations through a singular value of ~2e-7, curvature 5e-14 -- under the exclusion floor, so
  // the direction stayed where it was, while its gradient (2e-6: the task's residual is O(10)) is eight orders above its rounding and the reference's answer, -g / (c + 1e-12) ~ 1e6,
  // runs into a torque limit: arm torques off by 100 %, leg torques by 1e-5 .. 1e-2 on those ticks.  With 1e-12 on the diagonal the pivot is c + 1e-12, resolved as long as the
  // level's own rounding, 16 eps n max(K_jj), stays below 1e-12: true for the small, well-scaled last levels (n <= 12, K_jj ~ 1: the contact-force level of every gait and
  // controller, 2 .. 12 variables; against the exact solution the start-up branch's went from 6.6e-9 to 6.5e-13, three-leg stances of HierarchicalMpcWbc from 1.2e-7 to 2.5e-11
  // when the bound went from 8 to 12), NOT for the 18-variable level (K_jj ~ 2e3: its
  // rounding is 1e-10, and a direction it does not see at all would carry the rounding of the gradient divided by 1e-12) -- so the literal form is taken for levels of at most
  // kLiteralRegMaxN variables whose variables are still the reference's z (no implied-equality change of variables), the limit elsewhere.
  bool lit = false;
  bool literal() const { return lit && reg > 0.0; }
  double floorAbs() const { return literal() ? 0.0 : 10.0 * reg; }
  Mat hessian() const { Mat K = G0; if (literal()) for (int i = 0; i < K.r; ++i) K(i, i) += reg; return K; }
  Vec costGradient(const Vec& z) const { Vec gr = AZ.r > 0 ? tmul(AZ, AZ * z + rhat) : G0 * z + g; if (literal()) for (size_t i = 0; i < gr.size(); ++i) gr[i] += reg * z[i]; return gr; }
  // what one rounding of that gradient amounts to, component by component: the residual AZ z + rhat carries eps (|AZ| |z| + |rhat|) -- whatever its own size, it is a
  // difference -- and the product with AZ' passes that on: the largest component of eps |AZ|'(|AZ| |z| + |rhat|).  (The generic form: eps (|G0| |z| + |g|), passed in.)
  double gradientRounding(const Vec& z, double generic) const {
    if (AZ.r == 0) return generic;
    Vec a(AZ.r, 0.0);
    for (int r = 0; r < AZ.r; ++r) { double t = std::fabs(rhat[r]); for (int c = 0; c < AZ.c; ++c) t += std::fabs(AZ(r, c)) * std::fabs(z[c]); a[r] = t; }
    double worst = 0.0;
    for (int c = 0; c < AZ.c; ++c) { double t = 0.0; for (int r = 0; r < AZ.r; ++r) t += std::fabs(AZ(r, c)) * a[r]; worst = std::max(worst, t); }
    return std::max(worst, 1e-300);
  }
};
struct LevelWork {       // what the two phases share
  std::vector<char> on;  // rows that take part (not identically zero, not eliminated)
  Vec dn, wP;            // largest entry of a row; augmentation weight of a pinned row
  double hmax = 0.0, scale = 1.0;
};
inline LevelWork prepareLevel(const LevelQp& q) {
  LevelWork w;
  const int n = q.n(), m = q.m();
  w.on.assign(m, 0); w.dn.assign(m, 0.0); w.wP.assign(m, 0.0);
  for (int i = 0; i < n; ++i) w.hmax = std::max(w.hmax, q.G0(i, i));
  for (double v : q.g) w.scale = std::max(w.scale, std::fabs(v));
  for (int i = 0; i < m; ++i) {
    double d2 = 0.0;
    for (int j = 0; j < n; ++j) { w.dn[i] = std::max(w.dn[i], std::fabs(q.D(i, j))); d2 += q.D(i, j) * q.D(i, j); }
    w.on[i] = w.dn[i] > 0.0;     // rows that vanish identically carry no information (WbcBase.cpp:458 creates them)
    if (w.on[i]) { w.scale = std::max(w.scale, std::fabs(q.f[i])); }
  }
  // Augmentation weight of a pinned row: the largest curvature of the level's cost per unit of the row's squared norm -- along the normal of every pinned row the
  // factorised matrix is then at least as stiff as the cost is anywhere, so T = L^-1 D_P' and S = T'T stay O(1) however weakly the cost itself sees the directions the
  // row acts in (a cone row across force directions next to swing-leg tasks weighted x100: with the cost's own curvature along the normal S_jj reaches 1e10 and the
  // products T mu lose the digits that keep the row on its bound), and the rounding it adds, eps hmax, is the rounding G = AZ'AZ carries anyway.
  for (int i = 0; i < m; ++i) if (w.on[i]) {
    double d2 = 0.0;
    for (int j = 0; j < n; ++j) d2 += q.D(i, j) * q.D(i, j);
    w.wP[i] = std::max(1.0, w.hmax) / d2;
  }
  return w;
}

// ------------------------------------------------------------------------------------------------ phase 1: interior point on the inherited rows (a starting point, nothing more)
// Mehrotra predictor-corrector on   min 1/2 z'Gz + g'z  s.t.  D z + s = f, s >= 0   from z = 0, slacks max(sigma, f), multipliers sigma = sigma0 sqrt(scale).  Runs until the working set can
// plausibly be read off the iterate (duality measure <= 1e-6 scale with residuals to match), until it has converged or stagnates at the rounding floor of its normal
// equations, or until a step loses all accuracy (the previous iterate is handed over).  Returns the iterations used.
struct IpmPoint { Vec z, s, lam; bool usable = false, ownRows = false; };
constexpr double kIpmHandOverMu = 1e-8;     // duality measure (x scale) at which the working set is read off the iterate; x 1e-2 for each of the (at most two) resumptions.  (1e-6 saves 7 % of the passes and is NOT
                                            //  taken: a direction at the exclusion floor of the factorisation stays where the interior point left it, and at 1e-6 that is up to 0.7 of the torques away from the minimiser, at 1e-8 5e-4.)
// ownRows (round 6): the level's own rows take part as what they are in the reference's QP -- D z - v <= f with 1/2 v'v in the cost; v = lam at the optimum, so the row reads
// D z + s - lam = f, its weight in the normal equations is lam / (s + lam) <= 1 and its multiplier IS its violation.
inline int interiorPointPhase(const LevelQp& q, const LevelWork& w, double sigma0, IpmPoint& pt, double muTarget = kIpmHandOverMu, bool resume = false, int itStart = 0, bool ownRows = false) {
  const int n = q.n(), m = q.m();
  std::vector<int> rows; for (int i = ownRows ? 0 : q.mOwn; i < m; ++i) if (w.on[i]) rows.push_back(i);
  const int mr = int(rows.size());
  const IpmPoint from = pt;
  pt.z.assign(n, 0.0); pt.s.assign(m, 0.0); pt.lam.assign(m, 0.0); pt.usable = false; pt.ownRows = ownRows;
  if (mr == 0) return 0;
  const Mat& G = q.G0;       // (without HoQp's regulariser, as in the active-set phase: directions it alone would carry are left out of the factorisation)
  Mat D(mr, n); Vec f(mr);
  for (int r = 0; r < mr; ++r) { for (int j = 0; j < n; ++j) D(r, j) = q.D(rows[r], j); f[r] = q.f[rows[r]]; }
  const double scale = w.scale, sigma = sigma0 * std::sqrt(w.scale);
  Vec z(n, 0.0), s(mr), lam(mr, sigma);
  std::vector<char> soft(mr, 0); for (int r = 0; r < mr; ++r) soft[r] = rows[r] < q.mOwn;
  if (g_expIpmStartDelta > 0.0 && !ownRows) {
    // Experiment of round 6, measured and NOT kept (default 0 = off): inherited rows started ON their margins and centred, s = max(f, delta |d|_inf), lam = sigma^2 / s, instead of
    // s = lam = sigma (1.5e3 next to margins of 0-90: a start 1.4e3 outside every row, from which the diverged robots of the bench's steady-state leg crawl for ten iterations
    // at step lengths of 1-5 %).  delta = 10: second-level iterations on those robots 13.5 -> 9.3 in the mean, 23 -> 13 at most, regular closed-loop ticks 5.4 -> 5.2, their
    // ticks on the GPU 0.82 -> 0.66 ms -- but another path to the vertex: ONE tick of the static walk's 76,800 (HierarchicalWbc, where every tick had been within 1e-6) came out
    // 6e-3 apart between GPU and this restatement, on a tick whose torques move by 5e-3 under 1e-9 input noise on this side alone, and one instance of the 120-step bench leg
    // took 47 passes.  Parity on every tick is worth more than 0.15 ms on three robots.
    for (int i = 0; i < mr; ++i) { s[i] = std::max(f[i], g_expIpmStartDelta * w.dn[rows[i]]); lam[i] = sigma * sigma / s[i]; }
  } else
  for (int i = 0; i < mr; ++i) {
    if (!soft[i]) s[i] = std::max(sigma, f[i]);
    else if (f[i] >= 0.0) s[i] = f[i] + sigma;             // s - lam = f at z = 0: the row's equation holds from the start
    else { s[i] = sigma; lam[i] = sigma - f[i]; }
  }
  if (resume) { z = from.z; for (int r = 0; r < mr; ++r) { s[r] = from.s[rows[r]]; lam[r] = from.lam[rows[r]]; } }     // (the guess its last iterate gave was refuted: on from there)
  Vec zPrev = z, sPrev = s, lamPrev = lam;
  double nrdPrev = 0.0, muPrev = 0.0;
  auto handOver = [&](const Vec& zz, const Vec& ss, const Vec& ll) { pt.z = zz; for (int r = 0; r < mr; ++r) { pt.s[rows[r]] = ss[r]; pt.lam[rows[r]] = ll[r]; } pt.usable = true; };
  int it = itStart;
  for (; it < 40; ++it) {
    Vec rd = G * z + q.g + tmul(D, lam);
    if (q.literal()) for (int i = 0; i < n; ++i) rd[i] += q.reg * z[i];
    Vec rp = D * z + s - f;
    for (int i = 0; i < mr; ++i) if (soft[i]) rp[i] -= lam[i];
    const double mu = dot(s, lam) / mr;
    double nrd = 0, nrp = 0; for (double v : rd) nrd = std::max(nrd, std::fabs(v)); for (double v : rp) nrp = std::max(nrp, std::fabs(v));
    // a late Newton step of a degenerate problem (barrier weights ~1e18) can lose all accuracy: the previous iterate is what the active-set method starts from
    if (it > itStart && (!(nrd == nrd) || !(mu == mu) || nrd > 100.0 * std::max(nrdPrev, 1e-9 * scale))) { handOver(zPrev, sPrev, lamPrev); return it; }
    if (nrd <= 1e-4 * scale && nrp <= 1e-9 * scale && mu <= muTarget * scale) { handOver(z, s, lam); return it; }     // the working set can be read: over to the active-set method, for good
    if (it > itStart && mu > 0.5 * muPrev && mu <= kStagnationMu * scale) { handOver(z, s, lam); return it; }          // stagnation at the rounding floor
    zPrev = z; sPrev = s; lamPrev = lam; nrdPrev = nrd; muPrev = mu;
    Mat K = q.hessian();
    for (int r = 0; r < mr; ++r) { const double wr = soft[r] ? lam[r] / (s[r] + lam[r]) : lam[r] / s[r]; for (int i = 0; i < n; ++i) { const double wi = wr * D(r, i); if (wi == 0.0) continue; for (int j = 0; j < n; ++j) K(i, j) += wi * D(r, j); } }
    std::vector<char> excluded;
    if (!choleskyExcluding(K, q.floorAbs(), 16.0 * kEps, excluded)) { handOver(zPrev, sPrev, lamPrev); return it; }
    auto solve = [&](const Vec& rc, Vec& dz, Vec& ds, Vec& dl) {
      Vec t(mr); for (int i = 0; i < mr; ++i) t[i] = (lam[i] * rp[i] - rc[i]) / (soft[i] ? s[i] + lam[i] : s[i]);
      dz = -1.0 * (rd + tmul(D, t));
      cholSolveExcluding(K, excluded, dz);
      const Vec Ddz = D * dz;
      ds.resize(mr); dl.resize(mr);
      for (int i = 0; i < mr; ++i) {
        if (soft[i]) { dl[i] = (lam[i] * (Ddz[i] + rp[i]) - rc[i]) / (s[i] + lam[i]); ds[i] = (-rp[i] - Ddz[i]) + dl[i]; }
        else { ds[i] = -rp[i] - Ddz[i]; dl[i] = (-rc[i] - lam[i] * ds[i]) / s[i]; }
      }
    };
    auto maxStep = [&](const Vec& ds, const Vec& dl) { double a = 1.0; for (int i = 0; i < mr; ++i) { if (ds[i] < 0) a = std::min(a, -s[i] / ds[i]); if (dl[i] < 0) a = std::min(a, -lam[i] / dl[i]); } return a; };
    Vec rc(mr), dz, ds, dl;
    for (int i = 0; i < mr; ++i) rc[i] = s[i] * lam[i];
    solve(rc, dz, ds, dl);
    const double aAff = maxStep(ds, dl);
    double muAff = 0; for (int i = 0; i < mr; ++i) muAff += (s[i] + aAff * ds[i]) * (lam[i] + aAff * dl[i]); muAff /= mr;
    const double sig = std::pow(muAff / mu, 3.0);
    const double cw = std::min(1.0, 4.0 * aAff);
    for (int i = 0; i < mr; ++i) rc[i] = s[i] * lam[i] + cw * ds[i] * dl[i] - sig * mu;
    solve(rc, dz, ds, dl);
    const double tau = std::max(0.995, 1.0 - mu);
    const double a = std::min(1.0, tau * maxStep(ds, dl));
    for (int i = 0; i < n; ++i) z[i] += a * dz[i];
    for (int i = 0; i < mr; ++i) { s[i] += a * ds[i]; lam[i] += a * dl[i]; }
  }
  handOver(z, s, lam);
  return it;
}

// ------------------------------------------------------------------------------------------------ phase 2: primal active-set method
// States of a row: I inactive (D z < f; own rows: v = 0) | P pinned (D z = f; an own row has v = 0 as well, which is the optimum only if its multiplier vanishes) |
// V violated (own rows only: v = D z - f > 0, the row is an exact quadratic penalty).
//   iteration:  the minimiser on the working set from the current point, EXACTLY, by the range-space method on an augmented Hessian of the SAME size as the cost's:
//                 K = G + sum_P w_j d_j d_j' + sum_V d d',   w_j = (curvature of the cost along d_j) / |d_j|^2         (Cholesky K = L L'; directions without curvature left out)
//                 -- on the working set the added terms are constant, the minimiser is unchanged, and a pinned row across directions the cost does not see gives them the
//                 curvature its own normal has; nothing is multiplied by a large penalty, so a weakly curved direction (singular values of 1e-5 next to swing-leg tasks
//                 weighted x100 are routine) keeps its digits --
//                 T = L^-1 D_P',  S = T'T,   u = L^-1 (-grad - D_P' W r_P),   S mu = T'u + r_P,   p = L^-T (u - T mu):   D_P (z + p) = f_P, mu = the multipliers.
//                 (rows of the working set that are combinations of others show up as vanishing pivots of S and are skipped: their multiplier is zero)
//               step cut at the first row that changes sign (I: reaches its bound; V: its violation returns to zero): that row is pinned -- own rows too: whether it ends
//                 violated, inactive or exactly on its bound (reached through a direction without curvature) is decided by its multiplier
//               full step: one or two more passes of the same solve from the new point (iterative refinement: the gradient is evaluated in residual form, exact to the
//                 rounding of the residual), then the pinned row with the most negative multiplier is released (own rows: largest |multiplier|, to I or V by its sign);
//                 none: done
// Ties (several rows at the same step length: zero-margin rows of a degenerate vertex) go to the smallest row index.  A multiplier counts once lam_j |d_j| stands clear of
// the rounding of the gradient it balances, kAsLamTol eps (hmax |z| + scale).  A row released and pinned again by a zero-length step is not released again before the
// point has moved.
// start: z = 0 with the own rows whose bound is zero pinned (or, in the held-variable form of solveLevel, their variables held), or the interior point's iterate with
// its GUESS of the working set: the rows with multiplier > slack and the rows the iterate violates.  The guessed rows are still off their bounds by their slacks; the
// first step is meant to bring them there.  Rows of the guess that are combinations of other pinned rows leave first (the small system orders the rows already on
// their bounds in front, so that a dependency shows on a guessed row); a row the guessed step reaches only at its very end is not in its way; and if another row cuts
// the step short before anything has moved, the guess is refuted and solveLevel lets the interior point go on (at most twice; after that the step is taken as far as
// it goes and the row that cut it is pinned).  The first full step puts every pinned row on its bound; from there on the method is the textbook one.
// warmMask (inherited rows only; bit i = row i): the working set the PREVIOUS tick of this robot ended this level with, taken as the guess from z = 0 -- same rules as for
// the interior point's guess (dependent rows leave first, the first step is only taken in full), and a first step that any row cuts short refutes it, also when the
// guess was "no row is active": the level then starts over the cold way.
inline QpStats activeSetPhase(const LevelQp& q, const LevelWork& w, const IpmPoint* start, Vec& z, Vec& lamOut, std::vector<char>& stateOut, const std::vector<char>* fixedVars = nullptr, bool* guessRefuted = nullptr,
                              const uint64_t* warmMask = nullptr, const Vec* warmZ = nullptr, int iterationCap = kAsMaxWorkingSetChanges) {
  enum { I = 0, P = 1, V = 2 };
  QpStats st;
  const int n = q.n(), m = q.m(), mOwn = q.mOwn;
  const Mat& D = q.D; const Vec& f = q.f;
  const double scale = w.scale, tol = 1e-9 * scale, hmax = w.hmax;
  std::vector<char> state(m, I), stuck(m, 0), guess(m, 0);   // guess: pinned on the interior point's word, not yet brought to its bound
  Vec lam(m, 0.0);
  z.assign(n, 0.0);
  for (int i = 0; i < mOwn; ++i) if (w.on[i]) state[i] = f[i] < -tol ? V : (f[i] <= tol ? P : I);
  // (fixedVars: the variables the zero-bound own rows act on are held at zero instead of the rows being pinned -- solveLevel, the first attempt on a level with own rows)
  std::vector<char> rowOn = w.on;
  if (fixedVars) for (int i = 0; i < mOwn; ++i) if (rowOn[i] && state[i] == P) { bool inside = true; for (int j = 0; j < n; ++j) if (D(i, j) != 0.0 && !(*fixedVars)[j]) inside = false; if (inside) { rowOn[i] = 0; state[i] = I; } }
  if (start && start->usable) {
    z = start->z;
    const Vec Dz = D * z;
    for (int i = mOwn; i < m; ++i) if (rowOn[i] && (start->lam[i] > 1.0 * start->s[i] || Dz[i] - f[i] > 0.0)) { state[i] = P; guess[i] = 1; }
    // own rows behind the interior point: on the side of their bound the iterate has them on (no guess: a violated row is a penalty wherever it stands)
    if (start->ownRows) for (int i = 0; i < mOwn; ++i) if (rowOn[i]) { const double rr = Dz[i] - f[i]; state[i] = rr > tol ? V : (rr >= -tol ? P : I); }
  }
  if (warmMask) for (int i = mOwn; i < m && i < 64; ++i) if (rowOn[i] && ((*warmMask >> i) & 1ull)) { state[i] = P; guess[i] = 1; }
  if (warmMask && warmZ) z = *warmZ;      // (the previous tick's solution, scaled back into the rows by solveLevel)
  int lastReleased = -1, fullSteps = 0, guard = 0;
  for (;; ++st.iterations) {
    if (st.iterations > iterationCap && iterationCap < kAsMaxWorkingSetChanges) { st.status = 6; break; }      // (the held-variable form given up: solveLevel)
    if (st.iterations > kAsMaxWorkingSetChanges || ++guard > 4 * kAsMaxWorkingSetChanges) { st.status = 1; break; }      // (guard: every trip of the loop counts, also those that do not change the working set)
    Mat K = q.hessian();       // (without HoQp's regulariser: a direction it alone would carry counts as having no curvature, below)
    std::vector<int> pin;      // the pinned rows, those already on their bounds first: a dependency then shows on a row of the guess, never on a row the ratio test pinned
    for (int r = 0; r < m; ++r) if (rowOn[r] && state[r] == P && !guess[r]) pin.push_back(r);
    { // (the guessed rows by decreasing multiplier estimate of the interior point, lam |d|: of two guessed rows that depend on each other -- the two sides of a friction
      //  pyramid at its apex -- the one with the smaller estimate then shows the vanishing pivot and leaves; in index order it was the later one, the wrong one half of
      //  the time: released with a negative multiplier one iteration later, the other pinned by a zero-length step after that -- two factorisations for nothing)
      std::vector<int> gs;
      for (int r = 0; r < m; ++r) if (rowOn[r] && state[r] == P && guess[r]) gs.push_back(r);
      if (start && start->usable && g_expGuessOrder) std::stable_sort(gs.begin(), gs.end(), [&](int a, int b) { return start->lam[a] * w.dn[a] > start->lam[b] * w.dn[b]; });
      for (int r : gs) pin.push_back(r);
      // (the kernels' small system holds kAsMaxPinned rows; of a larger guess -- the interior point of a degenerate level: dozens of zero-margin rows with multiplier above
      //  slack -- the rows with the smallest estimates stay out, on both sides; the ratio test meets them again if the step crosses them)
      while (int(pin.size()) > kAsMaxPinned && guess[pin.back()]) { state[pin.back()] = I; guess[pin.back()] = 0; pin.pop_back(); }
    }
    for (int r = 0; r < m; ++r) {
      if (!rowOn[r] || state[r] == I) continue;
      const double wr = state[r] == P ? w.wP[r] : 1.0;
      for (int i = 0; i < n; ++i) { const double wi = wr * D(r, i); if (wi == 0.0) continue; for (int j = 0; j < n; ++j) K(i, j) += wi * D(r, j); }
    }
    const int k = int(pin.size());
    // a direction has no curvature when its pivot does not stand clear of the rounding of the matrix being factorised, or of HoQp's regulariser (x10: the reference's
    // 1e-12 I decides the directions whose curvature is comparable with it by a blend of task and minimum norm; here they count as unseen by the task)
    std::vector<char> excluded, dependent;
    if (!choleskyExcluding(K, q.floorAbs(), 16.0 * kEps, excluded, fixedVars)) { st.status = 2; break; }
    auto forward = [&](Vec b) { for (int i = 0; i < n; ++i) { if (excluded[i]) { b[i] = 0.0; continue; } double t = b[i]; for (int c = 0; c < i; ++c) t -= K(i, c) * b[c]; b[i] = t / K(i, i); } return b; };
    auto backward = [&](Vec b) { for (int i = n - 1; i >= 0; --i) { if (excluded[i]) { b[i] = 0.0; continue; } double t = b[i]; for (int c = i + 1; c < n; ++c) t -= K(c, i) * b[c]; b[i] = t / K(i, i); } return b; };
    Mat Tm(n, k), S(k, k);
    for (int j = 0; j < k; ++j) { Vec d(n); for (int c = 0; c < n; ++c) d[c] = D(pin[j], c); const Vec t = forward(d); for (int c = 0; c < n; ++c) Tm(c, j) = t[c]; }
    for (int a = 0; a < k; ++a) for (int b2 = 0; b2 < k; ++b2) { double t = 0.0; for (int c = 0; c < n; ++c) t += Tm(c, a) * Tm(c, b2); S(a, b2) = t; }
    if (k > 0) {   // dependent rows: pivot lost against the row's own diagonal entry (relative test, row by row: S_jj spans twelve orders of magnitude between cone and torque rows)
      Vec sdiag(k); for (int j = 0; j < k; ++j) sdiag[j] = S(j, j);
      dependent.assign(k, 0);
      for (int j = 0; j < k; ++j) {
        double d = S(j, j);
        for (int c = 0; c < j; ++c) d -= S(j, c) * S(j, c);
        if (!(d > 1e-11 * sdiag[j])) { dependent[j] = 1; for (int c = 0; c < j; ++c) S(j, c) = 0.0; S(j, j) = 1.0; for (int i = j + 1; i < k; ++i) S(i, j) = 0.0; continue; }
        d = std::sqrt(d); S(j, j) = d;
        for (int i = j + 1; i < k; ++i) { double t = S(i, j); for (int c = 0; c < j; ++c) t -= S(i, c) * S(j, c); S(i, j) = t / d; }
      }
    }
    // A pinned row that is a combination of other pinned rows is skipped by the solve (no pivot in S).  That is safe while the whole pinned set sits ON its bounds -- the
    // step keeps the others there and the combination with them.  The interior point's guess is different: its rows are still off their bounds by their slacks, the
    // first step is meant to bring them there, and that step is only taken if it can be taken in full (below).  A guess with a dependent row, or one whose step another
    // row cuts short, is dropped: the off-bound rows go back to inactive and the method continues from the same (feasible) point with what is left.
    double zmax0 = 1.0; for (double v : z) zmax0 = std::max(zmax0, std::fabs(v));
    bool offBound = false, anyDep = false;
    for (int j = 0; j < k; ++j) { offBound = offBound || guess[pin[j]]; anyDep = anyDep || dependent[j]; }
    // (a guess with dependent rows: those leave first -- the ratio test meets them again if the step crosses them)
    bool depGuess = false; for (int j = 0; j < k; ++j) depGuess = depGuess || (dependent[j] && guess[pin[j]]);     // (a dependency among rows already on their bounds is harmless: skipped by the solve)
    if (offBound && depGuess) { for (int j = 0; j < k; ++j) if (dependent[j] && guess[pin[j]]) { state[pin[j]] = I; guess[pin[j]] = 0; } fullSteps = 0; if (g_expTrace) fprintf(stderr, "  AS it %d: dependent rows of the guess leave the working set\n", st.iterations); --st.iterations; continue; }
    // one pass of the solve from zz: the step p and the multipliers mu of the pinned rows at zz + p
    auto solvePass = [&](const Vec& zz, Vec& pOut, Vec& muOut) {
      const Vec Dz = D * zz;
      Vec t(m, 0.0);
      for (int r = 0; r < m; ++r) if (rowOn[r]) { if (state[r] == P) t[r] = w.wP[r] * (Dz[r] - f[r]); else if (state[r] == V) t[r] = Dz[r] - f[r]; }
      const Vec u = forward(-1.0 * (q.costGradient(zz) + tmul(D, t)));
      Vec mu(k, 0.0);
      for (int j = 0; j < k; ++j) { double tj = Dz[pin[j]] - f[pin[j]]; for (int c = 0; c < n; ++c) tj += Tm(c, j) * u[c]; mu[j] = tj; }
      for (int j = 0; j < k; ++j) { if (dependent[j]) { mu[j] = 0.0; continue; } double tj = mu[j]; for (int c = 0; c < j; ++c) tj -= S(j, c) * mu[c]; mu[j] = tj / S(j, j); }
      for (int j = k - 1; j >= 0; --j) { if (dependent[j]) { mu[j] = 0.0; continue; } double tj = mu[j]; for (int c = j + 1; c < k; ++c) tj -= S(c, j) * mu[c]; mu[j] = tj / S(j, j); }
      Vec v = u;
      for (int c = 0; c < n; ++c) for (int j = 0; j < k; ++j) v[c] -= Tm(c, j) * mu[j];
      pOut = backward(v); muOut = mu;
      ++st.innerSteps;
    };
    Vec p, mu;
    solvePass(z, p, mu);
    bool finite = true; for (double v : p) finite = finite && v == v;
    if (!finite) { st.status = 2; break; }
    double pmax = 0.0; for (double v : p) pmax = std::max(pmax, std::fabs(v));
    const Vec Dz = D * z, Dp = D * p;
    // what the pinned rows let through (rounding; a dependent row that was skipped): a row that is a combination of pinned rows shows a step component of that size
    // and must not be taken for a blocking row
    double leak = 0.0;
    for (int i = 0; i < m; ++i) if (rowOn[i] && state[i] == P) leak = std::max(leak, std::fabs(Dp[i] + (Dz[i] - f[i])) / w.dn[i]);   // (a pinned row not yet on its bound -- the interior point's guess -- is meant to get there: D p = -(D z - f))
    // first sign change along the step
    double alpha = 1.0; int block = -1;
    for (int i = 0; i < m; ++i) {
      if (!rowOn[i] || state[i] == P) continue;
      const double epsP = std::max(1e-13 * std::max(1.0, pmax), 1e3 * leak) * w.dn[i];
      double a;
      if (state[i] == I) { if (!(Dp[i] > epsP)) continue; a = std::max(0.0, f[i] - Dz[i]) / Dp[i]; }
      else { if (!(Dp[i] < -epsP)) continue; a = std::max(0.0, Dz[i] - f[i]) / -Dp[i]; }
      if (a < alpha) { alpha = a; block = i; }      // strictly smaller: ties keep the smallest index
    }
    if (g_expTrace) { int nV = 0, nEx = 0, nDep = 0; for (int i = 0; i < m; ++i) nV += rowOn[i] && state[i] == V; for (char e : excluded) nEx += e; for (char e : dependent) nDep += e;
      fprintf(stderr, "  AS it %d n %d P %d (dependent %d) V %d excl %d pmax %.3e alpha %.3e block %d leak %.1e\n", st.iterations, n, k, nDep, nV, nEx, pmax, alpha, block, leak); }
    if (block >= 0 && offBound && alpha >= 1.0 - 1e-9) block = -1;      // (a row the guessed step reaches at its very end is not in its way)
    if (block >= 0 && fullSteps > 0 && pmax <= 1e-9 * zmax0) block = -1;  // (a refinement correction at rounding size changes no row's side: a row that ends exactly ON its bound -- a violated own row whose violation the level removes -- would otherwise be pinned or not by the last bit)
    // the step that was to bring the guessed rows onto their bounds is cut short by another row: the guess is wrong.  Nothing has moved yet: the caller may let the
    // interior point go on from its iterate and read the working set again (solveLevel; at most twice -- after that the step is taken as far as it goes)
    if (block >= 0 && (offBound || warmMask) && guessRefuted && st.adds == 0 && st.drops == 0) { *guessRefuted = true; st.status = 6; break; }
    if (block >= 0) {
      const bool moved = alpha * pmax > 1e-13 * zmax0;       // a step that does not move the point beyond its rounding counts as zero-length
      for (int i = 0; i < n; ++i) z[i] += alpha * p[i];
      if (moved) std::fill(stuck.begin(), stuck.end(), 0); else if (block == lastReleased) stuck[block] = 1;
      lastReleased = -1;
      state[block] = P; fullSteps = 0;
      ++st.adds; if (!moved) ++st.zeroSteps;
      continue;
    }
    for (int i = 0; i < n; ++i) z[i] += p[i];
    std::fill(guess.begin(), guess.end(), 0);     // a full step: every pinned row is on its bound now
    // refinement: the same working set once more from the new point (the gradient is re-evaluated exactly, the ratio test applies again) until the correction is
    // rounding -- at most three full steps in a row
    {
      double zmax = 1.0; for (double v : z) zmax = std::max(zmax, std::fabs(v));
      // (a correction that is itself small -- the step from the interior point's iterate -- is a Newton step on a quadratic: exact up to the rounding of the solve, which
      //  scales with the correction; only a step that moved the point by more than 1e-4 of its size is refined.  Refining every full step was measured in round 5 on
      //  HierarchicalMpcWbc's closed loop: the same 27 of 256,000 ticks deviate by the same amounts, so it is not paid for)
      if (pmax > 1e-13 * zmax && (pmax > 1e-4 * zmax || fullSteps > 0) && fullSteps < 3) { ++fullSteps; --st.iterations; if (g_expTrace) fprintf(stderr, "    full step %d on this working set: correction %.3e (zmax %.3e)\n", fullSteps, pmax, zmax); continue; }
    }
    std::fill(lam.begin(), lam.end(), 0.0);
    for (int j = 0; j < k; ++j) lam[pin[j]] = mu[j];
    double zmaxF = 1.0; for (double v : z) zmaxF = std::max(zmaxF, std::fabs(v));
    const double gradNoise = kAsLamTol * kEps * q.gradientRounding(z, hmax * zmaxF + scale);
    int rel = -1; double worst = 1.0;
    for (int i = 0; i < m; ++i) {
      if (!rowOn[i] || state[i] != P || stuck[i]) continue;
      const double bad = (i < mOwn ? std::fabs(lam[i]) : -lam[i]) * w.dn[i] / gradNoise;
      if (bad > worst) { worst = bad; rel = i; }
    }
    if (g_expTrace) { fprintf(stderr, "    full step: release %d (noise %.1e; hmax %.2e zmax %.2e scale %.2e)  pinned:", rel, gradNoise, hmax, zmaxF, scale); for (int i = 0; i < m; ++i) if (rowOn[i] && state[i] == P) fprintf(stderr, " %d:%.2e", i, lam[i]); fprintf(stderr, "\n"); }
    if (rel < 0) {
      // the point satisfies the KKT conditions on its working set; the bounds themselves once more (partial steps accumulate rounding)
      const Vec Df = D * z;
      for (int i = 0; i < m; ++i) if (rowOn[i] && (i >= mOwn || state[i] != V) && !(Df[i] - f[i] <= tol)) { st.status = 3; if (g_expTrace) fprintf(stderr, "    VIOLATED row %d state %d by %.3e (tol %.1e)\n", i, int(state[i]), Df[i] - f[i], tol); }
      // held variables: the cost must not want them moved (else the zero-bound rows have to be treated as rows: status 5, solveLevel starts over)
      if (fixedVars) {
        const Vec Dzf = D * z;
        Vec t(m, 0.0);
        for (int r = 0; r < m; ++r) if (rowOn[r]) { if (state[r] == P) t[r] = lam[r]; else if (state[r] == V) t[r] = Dzf[r] - f[r]; }
        const Vec gfull = q.costGradient(z) + tmul(D, t);
        for (int j = 0; j < n; ++j) if ((*fixedVars)[j] && std::fabs(gfull[j]) > gradNoise) st.status = 5;
      }
      // strongly active rows: pinned with a multiplier that counts, or violated
      // (a violated own row's multiplier is its violation; one that ends on its bound is not strongly active)
      { const Vec Dv = D * z; for (int i = 0; i < m; ++i) if (rowOn[i] && state[i] == V) lam[i] = Dv[i] - f[i]; }
      for (int i = 0; i < m; ++i) if (rowOn[i] && state[i] != I && !(lam[i] * w.dn[i] > gradNoise)) lam[i] = 0.0;
      break;
    }
    state[rel] = (rel < mOwn && lam[rel] > 0.0) ? V : I; lam[rel] = 0.0; lastReleased = rel; fullSteps = 0;
    ++st.drops;
  }
  lamOut = lam; stateOut = state;
  return st;
}

// ------------------------------------------------------------------------------------------------ implied equalities
// A row that a higher level left active with a positive multiplier (pinned, or an own row that ended violated) is an EQUALITY for every level below: stationarity of
// that level, G z + g + sum_j lam_j d_j = 0, projected on the null space N the next level moves in (N'(G z + g) = 0) gives sum_j lam_j (d_j N) = 0 with lam_j > 0 --
// the restricted rows are positively dependent, and every feasible direction w (d_j N w <= 0 for all j) has d_j N w = 0 for each of them.  The k-th row of such a set
// is a combination of the others: an inequality solver meets them as a cone without interior, discovers them row by row through zero-length steps and over-pins the
// dependent one on a rounding-size step component; an interior point has nothing to work in.  They are removed EXACTLY instead: z = N_E w with N_E the kernel of the
// stacked equality rows (full-pivot LU, rank revealing: dependent rows cost nothing), the level is solved in w.
inline Mat eliminateImpliedEqualities(LevelQp& q, const std::vector<char>& eq) {
  const int n = q.n(), m = q.m();
  std::vector<int> E; for (int i = q.mOwn; i < m; ++i) if (eq[i]) { bool nz = false; for (int j = 0; j < n; ++j) nz = nz || q.D(i, j) != 0.0; if (nz) E.push_back(i); }
  if (E.empty()) return Mat();
  Mat DE(int(E.size()), n);
  for (size_t r = 0; r < E.size(); ++r) for (int j = 0; j < n; ++j) DE(int(r), j) = q.D(E[r], j);
  const Mat N = kernelFullPivLU(DE);
  LevelQp r; r.mOwn = q.mOwn; r.reg = q.reg; r.lit = q.lit; r.f = q.f;
  r.G0 = T(N) * (q.G0 * N); r.g = tmul(N, q.g); r.D = q.D * N;
  if (q.AZ.r > 0) { r.AZ = q.AZ * N; r.rhat = q.rhat; r.G0 = T(r.AZ) * r.AZ; r.g = tmul(r.AZ, r.rhat); }
  // rows that are combinations of the eliminated ones vanish up to rounding in the new variables: they stay tight, and carry no information
  for (int i = 0; i < m; ++i) {
    double dOld = 0.0, dNew = 0.0;
    for (int j = 0; j < n; ++j) dOld = std::max(dOld, std::fabs(q.D(i, j)));
    for (int j = 0; j < r.D.c; ++j) dNew = std::max(dNew, std::fabs(r.D(i, j)));
    if (eq[i] || !(dNew > 1e-12 * dOld)) for (int j = 0; j < r.D.c; ++j) r.D(i, j) = 0.0;
  }
  q = r;
  return N;
}

// One level: z (decision variables of the level as the reference counts them), and the rows that are strongly active at the solution (eq, in / out).
// warm (in / out, may be null): bit 63 = valid, bits 0..55 = the rows this level ended pinned at the previous tick of the same robot in the same contact mode (wbcUpdate
// keeps one word per solve).  A valid word is tried first, from z = 0 (activeSetPhase: warmMask); refuted, the level is solved the cold way.  Any path ends at the same vertex.
// warmZ (in / out, may be null; with warm only): the level's solution z of that tick -- a level's cost leaves directions unseen (its Hessian is (A Z)'(A Z), rank < n: the next level
// decides them) and the minimiser reached from z = 0 through the seen directions alone is usually OUTSIDE the inherited rows while minimisers inside them exist (which is
// what the interior point finds); the previous tick's minimiser, a tick later, is still one of those up to the tick's change.  It is scaled back by t <= 1 until every
// row holds (z = 0 is feasible and the rows are convex), the carried rows are the guess, the first step must go through in full -- else the cold path.
inline QpStats solveLevel(LevelQp q, std::vector<char>& eq, Vec& z, uint64_t* warm = nullptr, double* warmZ = nullptr) {
  const int nFull = q.n(), m = q.m();
  eq.resize(m, 0);
  QpStats st;
  z.assign(nFull, 0.0);
  if (nFull == 0) return st;
  const Mat N = eliminateImpliedEqualities(q, eq);
  const bool reduced = N.r > 0;
  if (reduced) q.lit = false;      // (in the new variables w the regulariser would be 1e-12 N'N, not 1e-12 I: the limit is taken there)
  if (reduced) st.eliminated = nFull - N.c;
  if (reduced && N.c == 0) { if (warm) *warm = 0; return st; }          // the equalities leave nothing to decide: z = 0
  const LevelWork w = prepareLevel(q);
  IpmPoint pt;
  bool hard = false; for (int i = q.mOwn; i < m; ++i) hard = hard || w.on[i];
  // (levels of at most kIpmMinVariables variables -- the contact-force level after a trot's first two levels: 2 to 8 -- go without: cold, the active-set method needs
  //  2 iterations on average there and 14 at most, against 9.6 / 20 passes with the interior point in front: measured on 40k closed-loop ticks)
  const bool useIpm = hard && q.mOwn == 0 && q.n() > kIpmMinVariables && !g_expNoInteriorPoint;
  int ipmIt = 0;
  Vec zw, lam; std::vector<char> state;
  bool solved = false;
  // (Tried and NOT kept: the level's unconstrained minimiser first -- one factorisation from z = 0, done if no row is in its way: 22 % of a trot's ticks.  A direction whose
  //  curvature sits at the exclusion floor of the factorisation is then either solved for or left at ZERO, the full size of its component apart; behind the interior point it is
  //  left at the interior point's iterate, which is next to the minimiser either way.  HierarchicalMpcWbc's closed loop went from 5e-4 to 0.8 in its worst tick.)
  bool warmTried = false, warmRefuted = false;
  if (warm && (*warm >> 63) && q.mOwn == 0 && !g_expNoWarmStart) {
    const uint64_t mask = *warm & ((1ull << 56) - 1ull);
    warmTried = true;
    Vec z0(q.n(), 0.0);
    const bool haveZ = warmZ && ((*warm >> 62) & 1ull) && !reduced;
    if (haveZ) {
      for (int j = 0; j < q.n(); ++j) z0[j] = warmZ[j];
      const Vec Dz = q.D * z0;
      double t = 1.0;
      // (rows of the carried set may start beyond their bound -- the first step brings them onto it, as it does the rows an interior point's iterate violates; every
      //  other row must hold: f >= 0, the ratio is in [0, 1), and t depends continuously on the data)
      for (int i = 0; i < m; ++i) if (w.on[i] && !((mask >> i) & 1ull) && Dz[i] > q.f[i]) t = std::min(t, q.f[i] / Dz[i]);
      bool fin = t == t; for (double v : z0) fin = fin && v == v;
      for (double& v : z0) v = fin ? v * t : 0.0;
    }
    st = activeSetPhase(q, w, nullptr, zw, lam, state, nullptr, &warmRefuted, &mask, haveZ ? &z0 : nullptr);
    solved = !warmRefuted;
    if (g_expTrace) fprintf(stderr, "  working set of the previous tick %014llx: %s\n", (unsigned long long)mask, solved ? "taken" : "refuted");
  }
  if (!solved && useIpm) ipmIt = interiorPointPhase(q, w, g_expLowerLevelStart, pt);
  // A level with own rows whose bound is zero (the friction rows of the first level: every one of them acts on the contact forces only, 0 <= 0 at z = 0): pinning
  // them all holds those forces at zero.  Tried in that form first -- the variables held, the rows left out: no working set to carry, one factorisation and one solve away
  // from the limits -- and kept if, at the end, the cost does not want the held variables moved (the first level's cost vanishes at its minimiser: it never does, unless
  // torque limits are violated).  Otherwise the level is solved again with those rows as rows.
  // A level with own rows only and a task the variables can meet exactly (the first level: equations of motion and contact rows, 18 independent rows in 36 variables):
  // away from the limits its minimisers are the solutions of A Z z = -rhat, and the one the reference's 1e-12 I selects is the one of smallest norm,
  //   z = (A Z)' y,   (A Z)(A Z)' y = -rhat        (a Cholesky of the size of the TASK: 18 instead of 36)
  // -- contact forces that carry the robot and small accelerations, inside the friction cones and the torque limits in every regular tick.  Taken if it is: every own
  // row strictly satisfied (then no row is active, the point is the level's minimiser AND its canonical representative: wbcUpdate needs no completion here).
  if (q.mOwn > 0 && q.mOwn == m && q.AZ.r > 0 && q.AZ.r <= q.n() && !g_expNoMinNormStart) {
    const int r = q.AZ.r, n = q.n();
    Mat Gd(r, r);
    // (weighted norm: the variables the zero-bound rows act on -- the contact forces under their cones -- are cheap, kMinNormCheap x, so that forces carry the robot and
    //  the accelerations stay small: the unweighted point, with 1 N costing as much as 1 rad/s^2, leaves the cones)
    Vec winv(n, 1.0);
    for (int i = 0; i < q.mOwn; ++i) if (w.on[i] && q.f[i] == 0.0) for (int c = 0; c < n; ++c) if (q.D(i, c) != 0.0) winv[c] = kMinNormCheap;
    for (int a = 0; a < r; ++a) for (int b = 0; b < r; ++b) { double t = 0.0; for (int c = 0; c < n; ++c) t += q.AZ(a, c) * winv[c] * q.AZ(b, c); Gd(a, b) = t; }
    double dmax = 0.0; for (int a = 0; a < r; ++a) dmax = std::max(dmax, Gd(a, a));
    bool ok = true;
    for (int j = 0; j < r && ok; ++j) {     // plain Cholesky; a pivot lost against the diagonal (dependent task rows) ends the attempt
      double d = Gd(j, j);
      for (int k = 0; k < j; ++k) d -= Gd(j, k) * Gd(j, k);
      if (!(d > 1e-10 * dmax)) { ok = false; break; }
      d = std::sqrt(d); Gd(j, j) = d;
      for (int i = j + 1; i < r; ++i) { double t = Gd(i, j); for (int k = 0; k < j; ++k) t -= Gd(i, k) * Gd(j, k); Gd(i, j) = t / d; }
      for (int i = 0; i < j; ++i) Gd(i, j) = 0.0;
    }
    if (ok) {
      Vec y = -1.0 * q.rhat;
      cholSolve(Gd, y);
      zw = tmul(q.AZ, y);
      for (int c = 0; c < n; ++c) zw[c] *= winv[c];
      const Vec res = q.AZ * zw + q.rhat, Dz = q.D * zw;
      double resmax = 0.0, rscale = 1.0; for (int a = 0; a < r; ++a) { resmax = std::max(resmax, std::fabs(res[a])); rscale = std::max(rscale, std::fabs(q.rhat[a])); }
      ok = resmax <= 1e-9 * rscale;
      if (g_expTrace) { fprintf(stderr, "    min-norm: residual %.2e (scale %.2e) zmax %.2e; violated rows:", resmax, rscale, [&]{ double zm = 0; for (double v : zw) zm = std::max(zm, std::fabs(v)); return zm; }()); for (int i = 0; i < m; ++i) if (w.on[i] && !(Dz[i] - q.f[i] <= 0.0)) fprintf(stderr, " %d:%.2e", i, Dz[i] - q.f[i]); fprintf(stderr, "\n"); }
      for (int i = 0; i < m && ok; ++i) if (w.on[i] && !(Dz[i] - q.f[i] <= 0.0)) ok = false;
      for (double v : zw) ok = ok && v == v;
    }
    if (ok) { solved = true; st = QpStats(); st.minNorm = true; lam.assign(m, 0.0); state.assign(m, 0); }
    // (Measured in round 6 and NOT kept: the active-set method started FROM this point when it violates limits -- origin shifted, the violated rows start violated -- instead
    //  of from z = 0: on the eleven slowest ticks of the bench's steady-state leg, robots whose torque limits cannot hold, 58-72 working-set changes instead of 40-46.)
    if (g_expTrace) fprintf(stderr, "  minimum-norm start of the level: %s\n", ok ? "taken" : "rejected");
  }
  if (!solved && q.mOwn > 0 && q.mOwn == m && g_expOwnInteriorPoint == 2) {
    ipmIt = interiorPointPhase(q, w, g_expLowerLevelStart, pt, kIpmHandOverMu, false, 0, true);
  } else if (!solved && q.mOwn > 0) {
    std::vector<char> fixedVars(q.n(), 0); bool any = false;
    for (int i = 0; i < q.mOwn; ++i) if (w.on[i] && std::fabs(q.f[i]) <= 1e-9 * w.scale) for (int j = 0; j < q.n(); ++j) if (q.D(i, j) != 0.0) { fixedVars[j] = 1; any = true; }
    if (any) {
      st = activeSetPhase(q, w, nullptr, zw, lam, state, &fixedVars, nullptr, nullptr, nullptr, g_expOwnInteriorPoint == 3 ? kHeldFormMaxIterations : kAsMaxWorkingSetChanges);
      solved = st.status == 0;
      if (!solved && g_expTrace) fprintf(stderr, "  held-variable form rejected (status %d): the zero-bound rows as rows\n", st.status);
      // (round 6) rejected means the cost wants the held forces moved: limits are violated wherever the task is met.  From z = 0 the active-set method then changes one row per
      // iteration -- 40-46 of them on robots whose plan has diverged, the tail of the bench's steady-state leg --; the interior point moves all rows at once and hands over
      // the side of its bound every row ends on.
      if (!solved && (g_expOwnInteriorPoint == 1 || g_expOwnInteriorPoint == 3) && q.mOwn == m) ipmIt = interiorPointPhase(q, w, g_expLowerLevelStart, pt, kIpmHandOverMu, false, 0, true);
      if (solved) for (int i = 0; i < q.mOwn; ++i) if (w.on[i] && std::fabs(q.f[i]) <= 1e-9 * w.scale && state[i] == 0) { state[i] = 1; lam[i] = 0.0; }   // (reported as pinned with a vanishing multiplier: what they are)
    }
  }
  if (!solved) {
    double muTarget = kIpmHandOverMu;
    for (int resumed = 0;; ++resumed) {
      bool refuted = false;
      st = activeSetPhase(q, w, &pt, zw, lam, state, nullptr, (useIpm && pt.usable && resumed < 2) ? &refuted : nullptr);
      if (!refuted) break;
      muTarget *= 1e-2;
      if (g_expTrace) fprintf(stderr, "  guess refuted: interior point resumed (target %.0e)\n", muTarget);
      ipmIt = interiorPointPhase(q, w, g_expLowerLevelStart, pt, muTarget, true, ipmIt);
    }
  }
  st.ipmIterations = ipmIt; st.eliminated = reduced ? nFull - N.c : 0; st.warmTried = warmTried; st.warmRefuted = warmRefuted;
  if (warm) {
    uint64_t mask = 0;
    if (st.status == 0 && q.mOwn == 0) { mask = 1ull << 63; for (int i = 0; i < m && i < 56; ++i) if (w.on[i] && !state.empty() && state[i] == 1) mask |= 1ull << i; }
    if (mask && warmZ && !reduced) { mask |= 1ull << 62; for (int j = 0; j < q.n(); ++j) warmZ[j] = zw[j]; }      // (bit 62: the solution travels with the rows)
    *warm = mask;
  }
  if (st.status == 2) zw.assign(q.n(), 0.0);      // numerical failure: the level is skipped (x stays the higher priorities' solution) and flagged
  z = reduced ? N * zw : zw;
  for (int i = 0; i < m; ++i) if (w.on[i]) eq[i] = eq[i] || (state[i] != 0 && lam[i] > 0.0);
  return st;
}

// generic convex QP  min 1/2 z'Hz + c'z  s.t.  D z <= f  (tests: KKT residuals, enumeration): every row hard, feasible start required at z = 0 unless the interior point finds one
inline QpStats solveQpGeneric(const Mat& H, const Vec& c, const Mat& D, const Vec& f, Vec& z) {
  LevelQp q; q.G0 = H; q.g = c; q.D = D; q.f = f; q.mOwn = 0; q.reg = 0.0;
  std::vector<char> eq;
  return solveLevel(q, eq, z);
}

// ------------------------------------------------------------------------------------------------ HoQp (HoQp.cpp:12-158)
struct HoQp {
  Task task, stackedTasksPrev, stackedTasks;
  bool hasEq = false, hasIneq = false;
  int numSlack = 0, numDec = 0, numPrevSlack = 0;
  Mat Zprev, Z, Hm, Dm, aZ;
  Vec slackPrev, xPrev, cv, fv, stackedSlack, slackSol, decSol;
  QpStats stats, completionStats;
  bool completed = false;
  std::vector<char> stackedEq;   // per row of stackedTasks.d: strongly active at this level's solution or above -> an equality for the levels below
  std::vector<int> sel, selNext; // the coordinates of x that ARE the level's decision variables: z = (x - xPrev)[sel] (the kernel bases carry an identity block, kernelFullPivLU); selNext: the next level's
  int qpIters = 0, attempts = 0, polished = 0;   // (diagnostics kept for the tests: attempts is always 0 -- no relaxed re-solve exists any more; polished = the level ended at a verified vertex)

  HoQp(const Task& t, const HoQp* higher, bool canonical = false, uint64_t* warmMain = nullptr, uint64_t* warmCanon = nullptr, double* warmZ = nullptr, int warmZCap = 0) : task(t) {
    // initVars
    numSlack = task.d.r; hasEq = task.a.r > 0; hasIneq = numSlack > 0;
    if (higher) { Zprev = higher->Z; stackedTasksPrev = higher->stackedTasks; slackPrev = higher->stackedSlack; xPrev = higher->solution(); numPrevSlack = higher->stackedTasks.d.r; numDec = Zprev.c; }
    else { numDec = std::max(task.a.c, task.d.c); stackedTasksPrev = Task(Mat(0, numDec), Vec(), Mat(0, numDec), Vec()); Zprev = Mat::identity(numDec); xPrev = Vec(numDec, 0.0); numPrevSlack = 0; }
    stackedTasks = task + stackedTasksPrev;
    const int nz = numDec + numSlack;
    // buildHMatrix
    Hm = Mat(nz, nz);
    Mat zz0(numDec, numDec);
    if (hasEq) { aZ = task.a * Zprev; zz0 = T(aZ) * aZ; Mat zz = zz0; for (int i = 0; i < numDec; ++i) zz(i, i) += 1e-12; setBlock(Hm, 0, 0, zz); }
    for (int i = 0; i < numSlack; ++i) Hm(numDec + i, numDec + i) = 1.0;
    // buildCVector
    cv = Vec(nz, 0.0);
    if (hasEq) { const Vec t2 = tmul(aZ, task.a * xPrev - task.b); for (int i = 0; i < numDec; ++i) cv[i] = t2[i]; }
    // buildDMatrix / buildFVector
    const int rows = 2 * numSlack + numPrevSlack;
    Dm = Mat(rows, nz); fv = Vec(rows, 0.0);
    for (int i = 0; i < numSlack; ++i) Dm(i, numDec + i) = -1.0;
    if (numPrevSlack > 0) {
      const Mat dz = stackedTasksPrev.d * Zprev;
      setBlock(Dm, numSlack, 0, dz);
      const Vec dx = stackedTasksPrev.d * xPrev;
      // x_prev satisfies the inherited rows with its slack, so this margin is >= 0 in exact arithmetic; rounding can leave it at -1e-13.  Clamp at zero.
      for (int i = 0; i < numPrevSlack; ++i) fv[numSlack + i] = std::max(0.0, stackedTasksPrev.f[i] - dx[i] + slackPrev[i]);
    }
    if (hasIneq) {
      const Mat dz = task.d * Zprev;
      setBlock(Dm, numSlack + numPrevSlack, 0, dz);
      const Vec dx = task.d * xPrev;
      for (int i = 0; i < numSlack; ++i) { Dm(numSlack + numPrevSlack + i, numDec + i) = -1.0; fv[numSlack + numPrevSlack + i] = task.f[i] - dx[i]; }
    }
    // solveProblem: the reduced problem (slack block eliminated), own rows first, then the inherited ones
    Vec sol(nz, 0.0);
    if (numDec > 0) {
      const int mAll = numSlack + numPrevSlack;
      LevelQp q; q.G0 = zz0; q.g = Vec(cv.begin(), cv.begin() + numDec); if (hasEq) { q.AZ = aZ; q.rhat = task.a * xPrev - task.b; } q.D = Mat(mAll, numDec); q.f = Vec(mAll, 0.0); q.mOwn = numSlack; q.reg = hasEq ? 1e-12 : 0.0; q.lit = hasEq && numSlack == 0 && numDec <= g_expLiteralRegMaxN;
      for (int i = 0; i < numSlack; ++i) { for (int j = 0; j < numDec; ++j) q.D(i, j) = Dm(numSlack + numPrevSlack + i, j); q.f[i] = fv[numSlack + numPrevSlack + i]; }
      for (int i = 0; i < numPrevSlack; ++i) { for (int j = 0; j < numDec; ++j) q.D(numSlack + i, j) = Dm(numSlack + i, j); q.f[numSlack + i] = fv[numSlack + i]; }
      std::vector<char> eqr(mAll, 0);
      if (higher) for (int i = 0; i < numPrevSlack; ++i) eqr[numSlack + i] = higher->stackedEq[i];
      if (g_expTrace) fprintf(stderr, "level: numDec %d own %d inherited %d\n", numDec, numSlack, numPrevSlack);
      Vec zr;
      stats = solveLevel(q, eqr, zr, warmMain, numDec <= warmZCap ? warmZ : nullptr);
      stackedEq = eqr;      // rows ordered as stackedTasks.d = [own; inherited] (Task::operator+)
      if (g_expTrace) fprintf(stderr, " -> status %d eliminated %d ipm %d iterations %d adds %d drops %d zero steps %d inner %d\n", stats.status, stats.eliminated, stats.ipmIterations, stats.iterations, stats.adds, stats.drops, stats.zeroSteps, stats.innerSteps);
      qpIters = stats.status ? -1 : std::min(stats.ipmIterations + stats.iterations, 59);
      polished = stats.status == 0;
      for (int i = 0; i < numDec; ++i) sol[i] = zr[i];
    } else stackedEq.assign(numSlack + numPrevSlack, 0);
    decSol = Vec(sol.begin(), sol.begin() + numDec); slackSol = Vec(sol.begin() + numDec, sol.end());
    // the slack variables of the level: v = max(0, D z - f), exactly
    if (hasIneq) {
      const Vec dz = (task.d * Zprev) * decSol;
      const Vec dx = task.d * xPrev;
      for (int i = 0; i < numSlack; ++i) slackSol[i] = std::max(0.0, dz[i] + dx[i] - task.f[i]);
    }
    // buildZMatrix
    if (higher) sel = higher->selNext; else { sel.resize(numDec); for (int j = 0; j < numDec; ++j) sel[j] = j; }
    if (hasEq) { std::vector<int> freeCols; const Mat ker = kernelFullPivLU(aZ, nullptr, &freeCols); Z = Zprev * ker; for (int j : freeCols) selNext.push_back(sel[j]); } else { Z = Zprev; selNext = sel; }
    // stackSlackSolutions
    stackedSlack = higher ? vcat(higher->stackedSlack, slackSol) : slackSol;
    // Canonical representative (wbcUpdate): among the minimisers of this level -- z* + ker w inside the inherited rows and, for the level's own rows, inside the slack
    // they ended with -- the one of smallest norm in the level's own variables: what HoQp's 1e-12 I (HoQp.cpp:66) selects, in the limit 1e-12 -> 0.  It only moves the
    // point the next level starts from: the next level still decides every direction of ker, and the rows pinned here are nobody's equalities.
    if (canonical && hasEq && Z.c > 0 && numDec > 0) {
      const Mat ker = kernelFullPivLU(aZ);
      const Vec xL = solution();
      const int mAll = stackedTasks.d.r;
      LevelQp q; q.AZ = ker; q.rhat = decSol; q.G0 = T(ker) * ker; q.g = tmul(ker, decSol); q.mOwn = 0; q.reg = 0.0;
      q.D = stackedTasks.d * Z; q.f = Vec(mAll, 0.0);
      const Vec dx = stackedTasks.d * xL;
      for (int i = 0; i < mAll; ++i) { const double slack = i < numSlack ? slackSol[i] : slackPrev[i - numSlack]; q.f[i] = std::max(0.0, stackedTasks.f[i] - dx[i] + slack); }   // (rows ordered [own; inherited], Task::operator+)
      // (the rows strongly active at this level's solution are equalities on its solution set as well -- same argument as for the levels below -- and are eliminated
      //  the same way; what the completion itself pins is not passed on)
      std::vector<char> eqc = stackedEq;
      Vec wv;
      if (g_expTrace) fprintf(stderr, "completion: %d free directions\n", ker.c);
      completionStats = solveLevel(q, eqc, wv, warmCanon);
      completed = true;
      if (completionStats.status == 0 || completionStats.status == 1) { const Vec dzv = ker * wv; for (int i = 0; i < numDec; ++i) decSol[i] += dzv[i]; }
    }
  }
  Vec solution() const { return numDec > 0 ? xPrev + Zprev * decSol : xPrev; }
};

// HierarchicalWbc::update (variant 0) / HierarchicalMpcWbc::update (variant 1); returns [x(36); tau(18)]
inline int wbcUpdate(const qmgpu_problem& P, int variant, const double* xDes, const double* uDes, const double* rbd, int mode, double period, double time,
                     double* inputLast, double out[54], WbcModel* modelOut = nullptr, const double* eeForce = nullptr, int32_t* diag /*[8]: attempts, iterations per level*/ = nullptr,
                     uint64_t* ws /*[QMGPU_WBC_STATE_WORDS] in / out or null: the solver state carried from tick to tick (qmgpu_wbc_args::working_set)*/ = nullptr) {
  WbcModel w;
  std::unique_ptr<PhaseTimer> phase(new PhaseTimer(PH_WBC_MODEL));
  wbcUpdateMeasured(P, rbd, w);
  // force tracking (own formulation): M qdd + nle = S^T tau + Jc^T F + Jee^T f_e  <=>  nle <- nle - Jee^T f_e in the equations of
  // motion task, the torque limits and the torque recovery
  if (eeForce) for (int d = 0; d < NV; ++d) for (int a = 0; a < 3; ++a) w.nle[d] -= w.armJ(a, d) * eeForce[a];
  wbcUpdateDesired(P, xDes, uDes, inputLast, period, w, eeForce);
  WbcTasks tk(P, w, mode);
  const Task task0 = tk.floatingBaseEom() + tk.torqueLimits() + tk.noContactMotion() + tk.frictionCone();
  Task task1, task2;
  if (variant == 0) {
    task1 = (time < 10.0) ? tk.armJointNominalTracking() : (tk.baseHeight() + tk.baseAngular() + tk.eeLinear() + tk.eeAngular() + tk.swingLeg() * 100.0);
    task2 = tk.contactForce(uDes) + tk.baseLinear();
  } else {
    task1 = tk.baseHeight() + tk.baseAngular() + tk.baseLinear() + tk.swingLeg() * 100.0;
    task2 = tk.contactForce(uDes);
  }
  phase.reset(); phase.reset(new PhaseTimer(PH_WBC_QP));
  // Directions no task sees are fixed in the reference by HoQp's 1e-12 I alone (HoQp.cpp:66): every level returns, among its minimisers, the one of smallest norm IN ITS
  // OWN VARIABLES z -- the coordinates of the kernel basis the levels above left.  Where the LAST level decides everything that is left (every gait of gait.info once
  // the start-up branch is over) that choice is invisible: the next level re-decides the same directions, and the cascade runs without it.  Where directions are left
  // over at the end (swing legs during the start-up branch, flight), the result is the minimum-norm point of the last level RELATIVE TO the point the level above
  // returned, which is itself only defined by the same rule: the cascade is then run again with the canonical representative taken at every level (HoQp, canonical).
  // Taken in the limit 1e-12 -> 0 (with the regulariser itself those directions carry the rounding of the gradient divided by 1e-12).
  std::unique_ptr<HoQp> h0, h1, h2;
  const HoQp* last = nullptr;
  // HierarchicalMpcWbc gives the arm no task: its last level (contact forces) sees one combination of arm accelerations through a singular value of ~2e-7 -- curvature 5e-14, less
  // than the regulariser's 1e-12 -- so WHERE the level above left that direction is visible in the answer: the reference's regulariser pulls the last level's z to zero there, i.e.
  // back to the point the level above returned, which its own regulariser had made the minimum-norm one.  That controller therefore takes the canonical representative at every
  // level from the first pass on (measured against the 50-digit solution of the reference's QPs, tools/hoqp_exact.py: leg torques median 2.9e-8 -> 1.3e-10, and the answer stops
  // depending on the path to the vertex: 23 of 512 stress instances above 1e-9 -> 1).  HierarchicalWbc's last level sees everything that is left strongly; it keeps the rule above.
  bool canonical = g_expCanonicalFirst && variant == 1;
  // The working sets of the previous tick (one word per solve: [1 + 6 pass + 2 level + completion]) are guesses for this one as long as the rows mean the same thing:
  // same contact mode, controller and task set (word 0); anything else starts cold.  Words 13 / 14: passes of every solve of this tick (a byte each, bit 7 = guess refuted).
  if (ws) {
    const uint64_t key = (1ull << 63) | uint64_t(mode & 15) | (uint64_t(variant & 1) << 8) | (uint64_t(variant == 0 && time < 10.0) << 9);
    if (ws[0] != key) { for (int i = 1; i < QMGPU_WBC_STATE_WORDS; ++i) ws[i] = 0; ws[0] = key; }
    ws[13] = ws[14] = 0;
  }
  auto slot = [&](int pass, int level, int completion) { return ws ? ws + 1 + 6 * pass + 2 * level + completion : nullptr; };
  auto count = [&](int pass, int level, int completion, const QpStats& s) {
    if (!ws) return;
    const uint64_t b = uint64_t(std::min(s.ipmIterations + s.iterations, 127)) | (s.warmRefuted ? 128ull : 0ull);
    ws[13 + pass] |= b << (8 * (2 * level + completion));
  };
  for (int pass = 0; pass < 2; ++pass) {
    h2.reset();
    h0.reset(new HoQp(task0, nullptr, canonical, slot(pass, 0, 0), slot(pass, 0, 1)));
    // (words 16..33 / 34..41: the solutions of the second and third level of pass 0 travel with their rows -- solveLevel: warmZ)
    h1.reset(new HoQp(task1, h0.get(), canonical, slot(pass, 1, 0), slot(pass, 1, 1), (ws && pass == 0) ? reinterpret_cast<double*>(ws + 16) : nullptr, 18));
    if (h1->Z.c > 0) h2.reset(new HoQp(task2, h1.get(), canonical, slot(pass, 2, 0), slot(pass, 2, 1), (ws && pass == 0) ? reinterpret_cast<double*>(ws + 34) : nullptr, 8));     // FLY: level 2 has no decision variables left -> skip
    last = h2 ? h2.get() : h1.get();
    { const HoQp* lvp[3] = {h0.get(), h1.get(), h2.get()}; for (int l = 0; l < 3; ++l) if (lvp[l]) { count(pass, l, 0, lvp[l]->stats); if (lvp[l]->completed) count(pass, l, 1, lvp[l]->completionStats); } }
    if (canonical || !(last->Z.c > 0 && last->numDec > 0)) break;
    canonical = true;
  }
  const HoQp* lv[3] = {h0.get(), h1.get(), h2.get()};
  int status = 0;
  if (diag) for (int i = 0; i < 8; ++i) diag[i] = 0;
  for (int l = 0; l < 3; ++l) if (lv[l]) {
    status |= lv[l]->stats.status ? (1 << l) : 0;
    if (lv[l]->completed && lv[l]->completionStats.status) status |= 8;
    if (diag) { diag[l] = lv[l]->polished + 10 * std::min(lv[l]->stats.iterations, 99) + 1000 * std::min(lv[l]->stats.drops, 99); diag[4 + l] = lv[l]->qpIters; }
  }
  if (diag && last->completed) { diag[3] = (last->completionStats.status == 0) + 10 * std::min(last->completionStats.iterations, 99); diag[7] = std::min(last->completionStats.ipmIterations + last->completionStats.iterations, 59); }
  const Vec x = last->solution();
  // updateCmd (WbcBase.cpp:580-595)
  for (int i = 0; i < 36; ++i) out[i] = x[i];
  for (int i = 0; i < NJ; ++i) {
    double s = w.nle[6 + i];
    for (int j = 0; j < NV; ++j) s += w.M(6 + i, j) * x[j];
    for (int j = 0; j < 12; ++j) s -= w.J(j, 6 + i) * x[NV + j];
    out[36 + i] = s;
  }
  if (modelOut) *modelOut = w;
  return status;
}

}  // namespace qmo
