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

// Cholesky with a pivot floor (1e-13 x the largest diagonal entry of the cost Hessian, NOT of the barrier-weighted matrix): directions no task and no inequality row touches have (numerically) zero curvature;
// flooring the pivot leaves them where they are (their right-hand side is zero) instead of dividing by roundoff.
inline bool choleskyFloored(Mat& A, double floorv) {
  const int n = A.r;
  for (int j = 0; j < n; ++j) {
    double d = A(j, j);
    for (int k = 0; k < j; ++k) d -= A(j, k) * A(j, k);
    if (!(d > floorv)) d = floorv;
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A(j, j) = d;
    for (int i = j + 1; i < n; ++i) { double t = A(i, j); for (int k = 0; k < j; ++k) t -= A(i, k) * A(j, k); A(i, j) = t / d; }
    for (int i = 0; i < j; ++i) A(i, j) = 0.0;
  }
  return true;
}

// ------------------------------------------------------------------------------------------------ dense convex QP: min 1/2 z'Hz + c'z  s.t.  D z <= f
// Mehrotra predictor-corrector primal-dual interior point.  Returns iterations used, negative on failure.
// sigma0: starting value of the slacks (floor) and multipliers.  1 for the top level; kLowerLevelStart below it, where the cost gradients are
// ~1e4 and a unit start spends up to fifteen iterations on steps of a few per cent before the duality measure starts to fall (slowest of the
// 256 bench instances: 45 -> 31 iterations per update; mean 32.8 -> 24.2).  A constant, so that both implementations start identically
// whatever null-space basis they use; <= 0 selects sqrt(scale) (second / third attempts, see HoQp).
constexpr double kLowerLevelStart = 300.0;
constexpr double kStagnationMu = 1e-10;
constexpr bool kPolishAdd = false;       // adding violated rows to the guess (ipm_dev.h: QM_IPM_POLISH_ADD)
constexpr int kEarlyTriesOwn = 4; constexpr double kEarlyMuOwn = 1e-2, kEarlyNrpOwn = 1e-2, kEarlyNrdOwn = 1e-1, kEarlyDropOwn = 0.1;   // = QM_IPM_EARLY_*_OWN (ipm_dev.h)
constexpr bool kZeroTryOwn = true;       // = QM_IPM_ZERO_TRY_OWN of the kernels (ipm_dev.h)
constexpr int kPolishCorrections = 4;    // releases + additions per polish attempt (ipm_dev.h: QM_IPM_POLISH_CORRECTIONS)   // = QM_IPM_STAGNATION_MU of the kernels (ipm_dev.h)
// diagnostics of the last solveQpIpm call of this thread: 1 = the returned point is a polished (exact) vertex, 0 = the interior-point iterate stands
static thread_local int g_ipmPolished = 0;
static thread_local int g_ipmExit = 0; static thread_local double g_ipmExitMu = 0, g_ipmExitNrd = 0;   // TEMP diagnostics
// Experiment knobs (qmo_set_experiment; defaults = the product's algorithm).  lowerLevelStart: the interior point's starting slacks / multipliers below the
// top level -- a second value gives the ALGORITHMIC sensitivity of an instance (how far the oracle's own torques move when only the path of the
// interior point changes; tests/test_gpu_wbc.py).  orthonormalNullSpace: Gram-Schmidt on the kernel basis, the round-3 kernels' basis, which
// reproduces their round-4 failures inside the oracle (profiles/r04_notes.md section 1).
static double g_expLowerLevelStart = kLowerLevelStart;
static int g_expOrthonormalNullSpace = 0;
static int g_expNoZeroTry = 0;   // the first level without its zero try (tests: the shortcut must not change the result)
inline int solveQpIpm(const Mat& H, const Vec& c, const Mat& Din, const Vec& fin, Vec& z, int maxIter = 40, double* kktRes = nullptr, double sigma0 = 1.0,
                      bool activeSetCorrection = false) {
  const int n = H.r;
  // rows that are identically zero carry no information (the reference's friction task creates them, WbcBase.cpp:458)
  std::vector<int> keep;
  for (int i = 0; i < Din.r; ++i) { bool nz = false; for (int j = 0; j < n; ++j) if (Din(i, j) != 0.0) { nz = true; break; } if (nz) keep.push_back(i); }
  const int m = int(keep.size());
  Mat D(m, n); Vec f(m);
  for (int i = 0; i < m; ++i) { for (int j = 0; j < n; ++j) D(i, j) = Din(keep[i], j); f[i] = fin[keep[i]]; }
  z.assign(n, 0.0);
  if (m == 0) {
    Mat L = H; if (!cholesky(L)) return -1;
    z = -1.0 * c; cholSolve(L, z); return 0;
  }
  double pivotFloor = 0.0;
  for (int i = 0; i < n; ++i) pivotFloor = std::max(pivotFloor, 1e-13 * H(i, i));
  double scale = 1.0; for (double v : c) scale = std::max(scale, std::fabs(v)); for (double v : f) scale = std::max(scale, std::fabs(v));
  // starting point: slacks max(sigma, f - D z), multipliers sigma (sigma0, or sqrt(scale) for the second / third attempt, see HoQp)
  const double sigma = sigma0 > 0.0 ? sigma0 : std::sqrt(scale);
  Vec s(m), lam(m, sigma);
  { const Vec Dz = D * z; for (int i = 0; i < m; ++i) s[i] = std::max(sigma, f[i] - Dz[i]); }
  // ---- active-set polish.  The normal-equation interior point stalls at a dual residual of ~1e-7 * scale (barrier weights ~1e14);
  // an active-set solver like qpOASES returns the vertex itself.  With the active set read off the final iterate (multiplier larger
  // than slack) the equality-constrained QP is solved by a few augmented-Lagrangian Newton steps from the interior-point solution:
  //   grad = H z + c + D_A' (lam_A + rho r_A),  r = D z - f ;   (H + rho D_A' D_A) dz = -grad ;   lam_A += rho r_A(z + dz).
  // The result is kept only if it is primal feasible and its multipliers are non-negative (else the interior-point iterate stands).
  // activeSetCorrection (levels without slack variables of their own, i.e. every level below the first): the multiplier estimates after
  // the FIRST step already tell whether the guess was right (rho is large: they agree with the final ones to two digits).  If some are
  // negative while the point is feasible, those rows are released and the polish starts again from the interior-point iterate (once);
  // if the guess is still wrong after that the attempt is abandoned at once instead of after two more steps and the check.  On the
  // bench set the slowest instance read one weakly active row too many at both early attempts and went on for five more interior-point
  // iterations (12 + three polishes = 24 passes of its second level; now 14).
  // zero = the try BEFORE the first interior-point iteration (a level with slack variables of its own, kZeroTryOwn below): the guess is "no limit binds" --
  // the working set holds the rows with a zero right-hand side only (v >= 0 of every slack variable, and the friction rows of a swing leg, 0 <= 0), the
  // multiplier estimates start from zero.
  auto tryPolish = [&](bool early, bool zero = false) -> bool {
    std::vector<int> act;
    for (int i = 0; i < m; ++i) if (zero ? (f[i] <= 1e-9 * scale) : (lam[i] > s[i])) act.push_back(i);
    double hmax = 0.0; for (int i = 0; i < n; ++i) hmax = std::max(hmax, H(i, i));
    const double rho = 1e6 * std::max(1.0, hmax);
    // Active-set correction loop (levels without slack variables of their own): the estimates after the FIRST step of an attempt decide.  Negative
    // multipliers at a feasible point -> those rows are released; rows outside the guess that the step violates (weakly active rows whose slack and
    // multiplier both vanish: the interior point cannot classify them) -> they are added; either way the polish starts again from the interior-point
    // iterate, at most kPolishCorrections times per attempt (round 3: one release, no add -- with robots in motion 14-28 % of the level-1 polishes
    // were then rejected and the interior-point iterate, accurate to its tolerances only, stood).
    for (int corrections = 0;; ) {
      Mat K = H;
      for (int r : act) for (int i = 0; i < n; ++i) { const double wi = rho * D(r, i); if (wi == 0.0) continue; for (int j = 0; j < n; ++j) K(i, j) += wi * D(r, j); }
      if (!choleskyFloored(K, pivotFloor)) return false;
      Vec zp = z, lp(m, 0.0);
      for (int r : act) lp[r] = zero ? 0.0 : lam[r];
      bool again = false;
      // zero try with nothing pinned but simple bounds (v_i >= 0: rows with a single entry, decoupled from everything else as long as no constraint row is in the
      // working set): the first Newton step on the quadratic is the minimiser, the further steps would only repeat it
      bool boundsOnly = zero;
      if (zero) for (int r : act) { int nnz = 0; for (int j = 0; j < n; ++j) nnz += D(r, j) != 0.0; if (nnz > 1) { boundsOnly = false; break; } }
      const int nSteps = boundsOnly ? 1 : 3;
      for (int step = 0; step < nSteps; ++step) {
        const Vec Dz = D * zp;
        Vec t(m, 0.0);
        for (int r : act) t[r] = lp[r] + rho * (Dz[r] - f[r]);
        Vec dz = -1.0 * (H * zp + c + tmul(D, t));
        cholSolve(K, dz);
        for (int i = 0; i < n; ++i) zp[i] += dz[i];
        const Vec Dz2 = D * zp;
        for (int r : act) lp[r] += rho * (Dz2[r] - f[r]);
        if (activeSetCorrection && step == 0) {
          double viol = -1e300, lmin = 0.0;
          for (int i = 0; i < m; ++i) viol = std::max(viol, Dz2[i] - f[i]);
          for (int r : act) lmin = std::min(lmin, lp[r]);
          if (corrections < kPolishCorrections && lmin < -1e-9 * scale && viol <= 1e-6 * scale) {
            std::vector<int> kept;
            for (int r : act) if (!(lp[r] < 0.0)) kept.push_back(r);
            act.swap(kept); ++corrections; again = true;
            break;
          }
          if (kPolishAdd && !early && corrections < kPolishCorrections && viol > 1e-8 * scale && viol <= 0.1 * scale) {   // (only once the interior point has converged)
            std::vector<char> in(m, 0); for (int r : act) in[r] = 1;
            for (int i = 0; i < m; ++i) if (!in[i] && Dz2[i] - f[i] > 1e-9 * scale) in[i] = 1;
            act.clear(); for (int i = 0; i < m; ++i) if (in[i]) act.push_back(i);
            ++corrections; again = true;
            break;
          }
          if (!(viol <= 1e-8 * scale) || !(lmin >= -1e-8 * scale)) return false;
        }
      }
      if (again) continue;
      const Vec Dz = D * zp;
      bool ok = true;
      for (int i = 0; i < m; ++i) if (!(Dz[i] - f[i] <= 1e-9 * scale)) ok = false;
      for (int r : act) if (!(lp[r] >= -1e-9 * scale)) ok = false;
      for (double v : zp) if (!(v == v)) ok = false;
      if (ok) { z = zp; return true; }
      return false;
    }
  };
  // The first level (equations of motion, torque limits, friction cones; slack variables of its own): away from the limits NO inequality row is active and
  // the level is an equality-constrained least-squares problem -- one factorisation instead of two interior-point iterations and a polish.  Tried first;
  // accepted (all rows feasible, all multipliers of the working set >= -1e-9 scale: then it IS the solution of the strictly convex QP) in every instance of
  // the bench, moving, closed-loop and 2 x 2048 stress sets (profiles/r04_notes.md section 7); a rejected try leaves z, s, lam untouched.
  if (kZeroTryOwn && !g_expNoZeroTry && !activeSetCorrection && tryPolish(true, true)) { g_ipmPolished = 1; return 0; }
  int it = 0;
  int earlyTries = 0; double lastTryMu = 1e300;
  Vec zPrev = z, sPrev = s, lamPrev = lam;
  double nrdPrev = 0.0, muPrev = 0.0;
  for (; it < maxIter; ++it) {
    const Vec rd = H * z + c + tmul(D, lam);
    Vec rp = D * z + s - f;
    double mu = dot(s, lam) / m;
    double nrd = 0, nrp = 0; for (double v : rd) nrd = std::max(nrd, std::fabs(v)); for (double v : rp) nrp = std::max(nrp, std::fabs(v));
    // Late iterations of degenerate problems (rows active with a zero multiplier) push the barrier weights to ~1e18 and the
    // Newton step can lose all accuracy.  A step that blows the dual residual up (or produces NaN) is rejected: the previous
    // iterate is returned, as converged if its complementarity was already <= 1e-8 * scale, flagged otherwise.
    if (it > 0 && (!(nrd == nrd) || !(mu == mu) || nrd > 100.0 * std::max(nrdPrev, 1e-9 * scale))) {
      z = zPrev; s = sPrev; lam = lamPrev;
      if (kktRes) *kktRes = std::max(nrdPrev, muPrev);
      if (!(muPrev <= 1e-8 * scale)) return -3;
      g_ipmExit = 3; g_ipmExitMu = muPrev / scale; g_ipmExitNrd = nrdPrev / scale;
      break;   // accepted as converged: polished below like any other final iterate
    }
    if (kktRes) *kktRes = std::max(nrd, std::max(nrp, mu));
    // primal feasibility and complementarity tight; the dual residual tolerance is looser (see above)
    if (nrd <= 1e-7 * scale && nrp <= 1e-9 * scale && mu <= 1e-12 * scale) { g_ipmExit = 1; g_ipmExitMu = mu / scale; g_ipmExitNrd = nrd / scale; break; }
    // the polish is first tried as soon as the active set can plausibly be read off (mu <= 1e-6 scale; at most twice, the second time
    // only after the complementarity has dropped another 100x): an accepted vertex is exact whatever iterate it started from; a
    // rejected one leaves z, s, lam untouched and the interior point goes on
    // A level WITH slack variables of its own (the first one: torque limits and friction cones soften the equations of motion, and away from the limits
    // no row is active) is tried much earlier: from mu <= 1e-2 scale, up to four times, after every 10x drop.  Measured on the four parity sets
    // (profiles/r04_notes.md section 7): the first attempt, after two interior-point iterations instead of four, is accepted in every instance of
    // the bench and closed-loop sets and in 86 % of the stress instances (2.5 iterations on average instead of 3.9).
    const bool ownSlack = !activeSetCorrection;
    if (earlyTries < (ownSlack ? kEarlyTriesOwn : 2) && nrd <= (ownSlack ? kEarlyNrdOwn : 1e-4) * scale && nrp <= (ownSlack ? kEarlyNrpOwn : 1e-6) * scale &&
        mu <= (ownSlack ? kEarlyMuOwn : 1e-6) * scale && mu <= (ownSlack ? kEarlyDropOwn : 0.01) * lastTryMu) {
      ++earlyTries; lastTryMu = mu;
      if (tryPolish(true)) { g_ipmPolished = 1; return it; }
    }
    // stagnation: complementarity no longer halves although it is already small (round-off floor of the normal equations) -- stop
    // here instead of iterating into the divergence that follows; the polish finishes the job
    // (1e-10: round 3 stopped at mu <= 1e-6 * scale, i.e. after one slow iteration at an iterate whose active set cannot be read yet -- with robots in
    //  motion 14-28 % of the level-1 solves left that way, unpolished, 1e-5 .. 1e-1 off in the weakly weighted task directions)
    if (it > 0 && mu > 0.5 * muPrev && mu <= kStagnationMu * scale && nrp <= 1e-9 * scale && nrd <= 1e-7 * scale) { g_ipmExit = 2; g_ipmExitMu = mu / scale; g_ipmExitNrd = nrd / scale; break; }
    zPrev = z; sPrev = s; lamPrev = lam; nrdPrev = nrd; muPrev = mu;
    Mat K = H;
    for (int r = 0; r < m; ++r) { const double w = lam[r] / s[r]; for (int i = 0; i < n; ++i) { const double wi = w * D(r, i); if (wi == 0.0) continue; for (int j = 0; j < n; ++j) K(i, j) += wi * D(r, j); } }
    if (!choleskyFloored(K, pivotFloor)) return -2;
    auto solve = [&](const Vec& rc, Vec& dz, Vec& ds, Vec& dl) {
      Vec t(m); for (int i = 0; i < m; ++i) t[i] = (lam[i] * rp[i] - rc[i]) / s[i];
      dz = -1.0 * (rd + tmul(D, t));
      cholSolve(K, dz);
      const Vec Ddz = D * dz;
      ds.resize(m); dl.resize(m);
      for (int i = 0; i < m; ++i) { ds[i] = -rp[i] - Ddz[i]; dl[i] = (-rc[i] - lam[i] * ds[i]) / s[i]; }
    };
    auto maxStep = [&](const Vec& ds, const Vec& dl) { double a = 1.0; for (int i = 0; i < m; ++i) { if (ds[i] < 0) a = std::min(a, -s[i] / ds[i]); if (dl[i] < 0) a = std::min(a, -lam[i] / dl[i]); } return a; };
    Vec rc(m), dz, ds, dl;
    for (int i = 0; i < m; ++i) rc[i] = s[i] * lam[i];
    solve(rc, dz, ds, dl);
    const double aAff = maxStep(ds, dl);
    double muAff = 0; for (int i = 0; i < m; ++i) muAff += (s[i] + aAff * ds[i]) * (lam[i] + aAff * dl[i]); muAff /= m;
    const double sigma = std::pow(muAff / mu, 3.0);
    const double cw = std::min(1.0, 4.0 * aAff);
    for (int i = 0; i < m; ++i) rc[i] = s[i] * lam[i] + cw * ds[i] * dl[i] - sigma * mu;
    solve(rc, dz, ds, dl);
    const double tau = std::max(0.995, 1.0 - mu);
    const double a = std::min(1.0, tau * maxStep(ds, dl));
    for (int i = 0; i < n; ++i) z[i] += a * dz[i];
    for (int i = 0; i < m; ++i) { s[i] += a * ds[i]; lam[i] += a * dl[i]; }
  }
  if (it >= maxIter) return -4;   // iteration cap: a failure like the others (HoQp retries from a different starting point)
  g_ipmPolished = tryPolish(false) ? 1 : 0;
  if (!g_ipmPolished && getenv("QMO_DEBUG_EXIT")) fprintf(stderr, "UNPOLISHED n %d exit %d mu/s %.1e nrd/s %.1e it %d\n", n, g_ipmExit, g_ipmExitMu, g_ipmExitNrd, it);
  return it;
}

// ------------------------------------------------------------------------------------------------ HoQp (HoQp.cpp:12-158)
constexpr double kInheritedMargin = 1e-5;   // second-attempt relaxation of inherited inequality rows (see HoQp)
struct HoQp {
  Task task, stackedTasksPrev, stackedTasks;
  bool hasEq = false, hasIneq = false;
  int numSlack = 0, numDec = 0, numPrevSlack = 0;
  Mat Zprev, Z, Hm, Dm;
  Vec slackPrev, xPrev, cv, fv, stackedSlack, slackSol, decSol;
  int qpIters = 0, attempts = 0, polished = 0;   // attempts: 0 = the first solve converged, 1 / 2 = the relaxed re-solves, 3 = level skipped

  HoQp(const Task& t, const HoQp* higher) : task(t) {
    // initVars
    numSlack = task.d.r; hasEq = task.a.r > 0; hasIneq = numSlack > 0;
    if (higher) { Zprev = higher->Z; stackedTasksPrev = higher->stackedTasks; slackPrev = higher->stackedSlack; xPrev = higher->solution(); numPrevSlack = higher->stackedTasks.d.r; numDec = Zprev.c; }
    else { numDec = std::max(task.a.c, task.d.c); stackedTasksPrev = Task(Mat(0, numDec), Vec(), Mat(0, numDec), Vec()); Zprev = Mat::identity(numDec); xPrev = Vec(numDec, 0.0); numPrevSlack = 0; }
    stackedTasks = task + stackedTasksPrev;
    const int nz = numDec + numSlack;
    // buildHMatrix
    Hm = Mat(nz, nz);
    Mat aZ;
    if (hasEq) { aZ = task.a * Zprev; Mat zz = T(aZ) * aZ; for (int i = 0; i < numDec; ++i) zz(i, i) += 1e-12; setBlock(Hm, 0, 0, zz); }
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
      // x_prev satisfies the inherited rows with its slack, so this margin is >= 0 in exact arithmetic; rounding (and the 1e-9 * scale
      // feasibility tolerance of the polished vertex) can leave it at -1e-8, which a lower level whose null space no longer sees the
      // row cannot repair.  Clamp at zero.
      for (int i = 0; i < numPrevSlack; ++i) fv[numSlack + i] = std::max(0.0, stackedTasksPrev.f[i] - dx[i] + slackPrev[i]);
    }
    if (hasIneq) {
      const Mat dz = task.d * Zprev;
      setBlock(Dm, numSlack + numPrevSlack, 0, dz);
      const Vec dx = task.d * xPrev;
      for (int i = 0; i < numSlack; ++i) { Dm(numSlack + numPrevSlack + i, numDec + i) = -1.0; fv[numSlack + numPrevSlack + i] = task.f[i] - dx[i]; }
    }
    // solveProblem
    Vec sol;
    // A degenerate low-priority level -- more inherited rows active at x_prev than the remaining null space has dimensions, so that
    // {z : D_prev Z z <= margin} has no interior (three-leg stance: 5 free directions, 7 active rows) -- stalls the interior point.
    // Further attempts: every inherited row gets a margin of at least kInheritedMargin (1e-5 N / Nm: 3e-7 of the limits it bounds), then
    // 100x that, and the iteration starts from slacks / multipliers of O(sqrt(scale)).  If that fails too the level is skipped (z = 0: x stays the
    // higher priorities' solution) and the failure is reported.  2 x 2048 random configurations: 6 second attempts, no skip.
    if (nz > 0) {
      g_ipmPolished = 0;
      qpIters = solveQpIpm(Hm, cv, Dm, fv, sol, 40, nullptr, higher ? g_expLowerLevelStart : 1.0, numSlack == 0);
      for (int attempt = 1; attempt <= 2 && qpIters < 0; ++attempt) {
        Vec fr = fv;
        const double margin = attempt == 1 ? kInheritedMargin : 100.0 * kInheritedMargin;   // 1e-5, then 1e-3
        for (int i = 0; i < numPrevSlack; ++i) fr[numSlack + i] = std::max(margin, fr[numSlack + i]);
        g_ipmPolished = 0;
        qpIters = solveQpIpm(Hm, cv, Dm, fr, sol, 40, nullptr, -1.0, numSlack == 0);
        attempts = attempt;
      }
      if (qpIters < 0) { sol.assign(nz, 0.0); attempts = 3; }
      polished = g_ipmPolished;
    } else sol.clear();
    decSol = Vec(sol.begin(), sol.begin() + numDec); slackSol = Vec(sol.begin() + numDec, sol.end());
    // An interior point method leaves the slacks of inactive rows at O(sqrt(mu)) (v = 0 and its multiplier = 0 is a degenerate
    // complementarity pair).  The exact minimiser, which an active-set solver like qpOASES returns, has v = max(0, D z - f):
    // restore it from the decision variables, which are not affected.
    if (hasIneq) {
      const Vec dz = (task.d * Zprev) * decSol;
      const Vec dx = task.d * xPrev;
      for (int i = 0; i < numSlack; ++i) slackSol[i] = std::max(0.0, dz[i] + dx[i] - task.f[i]);
    }
    // buildZMatrix
    if (hasEq) Z = Zprev * kernelFullPivLU(aZ); else Z = Zprev;
    if (g_expOrthonormalNullSpace && Z.c > 0) {   // EXPERIMENT: orthonormal columns (modified Gram-Schmidt, twice)
      for (int pass = 0; pass < 2; ++pass)
        for (int j = 0; j < Z.c; ++j) {
          for (int k = 0; k < j; ++k) { double d = 0; for (int i = 0; i < Z.r; ++i) d += Z(i, k) * Z(i, j); for (int i = 0; i < Z.r; ++i) Z(i, j) -= d * Z(i, k); }
          double nn = 0; for (int i = 0; i < Z.r; ++i) nn += Z(i, j) * Z(i, j); nn = std::sqrt(nn); for (int i = 0; i < Z.r; ++i) Z(i, j) /= nn;
        }
    }
    // stackSlackSolutions
    stackedSlack = higher ? vcat(higher->stackedSlack, slackSol) : slackSol;
  }
  Vec solution() const { return numDec > 0 ? xPrev + Zprev * decSol : xPrev; }
};

// HierarchicalWbc::update (variant 0) / HierarchicalMpcWbc::update (variant 1); returns [x(36); tau(18)]
inline int wbcUpdate(const qmgpu_problem& P, int variant, const double* xDes, const double* uDes, const double* rbd, int mode, double period, double time,
                     double* inputLast, double out[54], WbcModel* modelOut = nullptr, const double* eeForce = nullptr, int32_t* diag /*[8]: attempts, iterations per level*/ = nullptr) {
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
  HoQp h0(task0, nullptr);
  HoQp h1(task1, &h0);
  Vec x;
  int status = (h0.qpIters < 0 || h0.qpIters >= 60 ? 1 : 0) | (h1.qpIters < 0 || h1.qpIters >= 60 ? 2 : 0);
  if (diag) { for (int i = 0; i < 8; ++i) diag[i] = 0; diag[0] = h0.attempts + 10 * h0.polished; diag[1] = h1.attempts + 10 * h1.polished; diag[4] = h0.qpIters; diag[5] = h1.qpIters; }
  if (h1.Z.c > 0) {
    HoQp h2(task2, &h1); x = h2.solution(); status |= (h2.qpIters < 0 || h2.qpIters >= 60 ? 4 : 0);
    if (diag) { diag[2] = h2.attempts + 10 * h2.polished; diag[6] = h2.qpIters; }
    if (h2.Z.c > 0) {
      // Directions no task sees (the arm accelerations of HierarchicalMpcWbc) are fixed in the reference only by HoQp's 1e-12
      // regulariser and qpOASES' internal regularisation, i.e. "small".  Defined here as the minimum-norm completion: one more
      // level with the task x = 0.
      HoQp h3(Task(Mat::identity(36), Vec(36, 0.0), Mat(), Vec()), &h2); x = h3.solution(); status |= (h3.qpIters < 0 || h3.qpIters >= 60 ? 8 : 0);
      if (diag) { diag[3] = h3.attempts + 10 * h3.polished; diag[7] = h3.qpIters; }
    }
  } else x = h1.solution();  // FLY: level 2 has no decision variables left -> skip
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
