// qmo_model.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; parity unpinned, see qmo_core.h).
//
// Rigid-body quantities of the 24-DoF AlienGo+Z1 model, written the "textbook" way (world-frame
// geometric Jacobians summed over bodies) and templated on the scalar so that derivatives come from
// forward-mode dual numbers, as the reference obtains them from CppAD.
//
// Restates (upstream, not vendored): pinocchio::forwardKinematics / computeJointJacobians /
// computeCentroidalMap / crba / nonLinearEffects / getFrameJacobian[TimeVariation] / dccrba as they are
// called from qm_interface/src/QMPreComputation.cpp:73-88 and qm_wbc/src/WbcBase.cpp:146-238, for the model
// OCS2 builds in centroidal_model::createPinocchioInterface (composite Translation + SphericalZYX root,
// generalized velocity v = [dp_world, d(yaw,pitch,roll), dq_joint]).
#pragma once
#include "../include/qmgpu.h"
#include "qmo_core.h"

namespace qmo {

constexpr int NB = QMGPU_NB, NV = QMGPU_NV, NJ = QMGPU_NJ, NX = QMGPU_NX, NU = QMGPU_NU, NCT = QMGPU_NC;

template <class S> struct V3 {
  S x, y, z;
  V3() : x(0.0), y(0.0), z(0.0) {}
  V3(S a, S b, S c) : x(a), y(b), z(c) {}
  S& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  const S& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
template <class S> V3<S> operator+(const V3<S>& a, const V3<S>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class S> V3<S> operator-(const V3<S>& a, const V3<S>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class S> V3<S> operator*(const S& s, const V3<S>& a) { return {s * a.x, s * a.y, s * a.z}; }
template <class S> V3<S> cross(const V3<S>& a, const V3<S>& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <class S> S dot(const V3<S>& a, const V3<S>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

template <class S> struct M3 {
  S m[3][3];
  M3() { for (auto& r : m) for (auto& e : r) e = S(0.0); }
  static M3 identity() { M3 r; r.m[0][0] = r.m[1][1] = r.m[2][2] = S(1.0); return r; }
};
template <class S> M3<S> operator*(const M3<S>& a, const M3<S>& b) { M3<S> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { S s(0.0); for (int k = 0; k < 3; ++k) s += a.m[i][k] * b.m[k][j]; r.m[i][j] = s; } return r; }
template <class S> V3<S> operator*(const M3<S>& a, const V3<S>& b) { return {a.m[0][0] * b.x + a.m[0][1] * b.y + a.m[0][2] * b.z, a.m[1][0] * b.x + a.m[1][1] * b.y + a.m[1][2] * b.z, a.m[2][0] * b.x + a.m[2][1] * b.y + a.m[2][2] * b.z}; }
template <class S> M3<S> transpose(const M3<S>& a) { M3<S> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i]; return r; }
template <class S> M3<S> operator+(const M3<S>& a, const M3<S>& b) { M3<S> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] + b.m[i][j]; return r; }
template <class S> M3<S> axisRotation(int axis, const S& angle) {
  const S c = cos(angle), s = sin(angle);
  M3<S> r = M3<S>::identity();
  const int a = (axis + 1) % 3, b = (axis + 2) % 3;
  r.m[a][a] = c; r.m[a][b] = S(0.0) - s; r.m[b][a] = s; r.m[b][b] = c;
  return r;
}
template <class S> M3<S> inverse3(const M3<S>& a) {
  M3<S> r;
  const S c00 = a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1];
  const S c01 = a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2];
  const S c02 = a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0];
  const S det = a.m[0][0] * c00 + a.m[0][1] * c01 + a.m[0][2] * c02;
  const S id = S(1.0) / det;
  r.m[0][0] = c00 * id; r.m[1][0] = c01 * id; r.m[2][0] = c02 * id;
  r.m[0][1] = (a.m[0][2] * a.m[2][1] - a.m[0][1] * a.m[2][2]) * id;
  r.m[1][1] = (a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0]) * id;
  r.m[2][1] = (a.m[0][1] * a.m[2][0] - a.m[0][0] * a.m[2][1]) * id;
  r.m[0][2] = (a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1]) * id;
  r.m[1][2] = (a.m[0][2] * a.m[1][0] - a.m[0][0] * a.m[1][2]) * id;
  r.m[2][2] = (a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0]) * id;
  return r;
}

// True when generalized velocity k moves body `body`.
inline bool dofMovesBody(const qmgpu_model& md, int k, int body) {
  if (k < 6) return true;
  const int jb = k - 5;  // body carried by joint k
  for (int b = body; b >= 0; b = md.parent[b]) if (b == jb) return true;
  return false;
}

template <class S> struct Kin {
  M3<S> R[NB];
  V3<S> p[NB];
  V3<S> axis[NV];    // world axis of DoF k (for k<3: translation direction)
  V3<S> origin[NV];  // a point on the axis (k>=3)
  V3<S> com[NB];     // world
  M3<S> Iw[NB];      // world-aligned inertia about the body com
  V3<S> foot[NCT];
  V3<S> ee;
  M3<S> Ree;
  V3<S> comTotal;
};

template <class S> void forwardKinematics(const qmgpu_model& md, const S* q, Kin<S>& k) {
  const M3<S> Rz = axisRotation<S>(2, q[3]), Ry = axisRotation<S>(1, q[4]), Rx = axisRotation<S>(0, q[5]);
  k.R[0] = Rz * Ry * Rx;
  k.p[0] = V3<S>(q[0], q[1], q[2]);
  for (int a = 0; a < 3; ++a) { V3<S> e; e[a] = S(1.0); k.axis[a] = e; k.origin[a] = V3<S>(); }
  k.axis[3] = V3<S>(S(0.0), S(0.0), S(1.0));
  k.axis[4] = Rz * V3<S>(S(0.0), S(1.0), S(0.0));
  k.axis[5] = (Rz * Ry) * V3<S>(S(1.0), S(0.0), S(0.0));
  for (int a = 3; a < 6; ++a) k.origin[a] = k.p[0];
  for (int b = 1; b < NB; ++b) {
    const int par = md.parent[b];
    const V3<S> off(S(md.joint_offset[b][0]), S(md.joint_offset[b][1]), S(md.joint_offset[b][2]));
    k.p[b] = k.p[par] + k.R[par] * off;
    V3<S> e; e[md.axis[b]] = S(1.0);
    k.axis[5 + b] = k.R[par] * e;
    k.origin[5 + b] = k.p[b];
    k.R[b] = k.R[par] * axisRotation<S>(md.axis[b], q[5 + b]);
  }
  V3<S> msum;
  for (int b = 0; b < NB; ++b) {
    k.com[b] = k.p[b] + k.R[b] * V3<S>(S(md.com[b][0]), S(md.com[b][1]), S(md.com[b][2]));
    M3<S> I;
    const double* in = md.inertia[b];
    I.m[0][0] = S(in[0]); I.m[0][1] = I.m[1][0] = S(in[1]); I.m[0][2] = I.m[2][0] = S(in[2]);
    I.m[1][1] = S(in[3]); I.m[1][2] = I.m[2][1] = S(in[4]); I.m[2][2] = S(in[5]);
    k.Iw[b] = k.R[b] * I * transpose(k.R[b]);
    msum = msum + S(md.mass[b]) * k.com[b];
  }
  k.comTotal = S(1.0 / md.total_mass) * msum;
  for (int c = 0; c < NCT; ++c) {
    const int b = md.foot_body[c];
    k.foot[c] = k.p[b] + k.R[b] * V3<S>(S(md.foot_offset[c][0]), S(md.foot_offset[c][1]), S(md.foot_offset[c][2]));
  }
  k.ee = k.p[md.ee_body] + k.R[md.ee_body] * V3<S>(S(md.ee_offset[0]), S(md.ee_offset[1]), S(md.ee_offset[2]));
  k.Ree = k.R[md.ee_body];
}

// Column k of the world-frame (LOCAL_WORLD_ALIGNED) Jacobian of a point r rigidly attached to `body`.
template <class S> void pointJacobianColumn(const qmgpu_model& md, const Kin<S>& k, int body, const V3<S>& r, int dof, V3<S>& lin, V3<S>& ang) {
  lin = V3<S>(); ang = V3<S>();
  if (!dofMovesBody(md, dof, body)) return;
  if (dof < 3) { lin = k.axis[dof]; return; }
  ang = k.axis[dof];
  lin = cross(k.axis[dof], r - k.origin[dof]);
}

// Centroidal momentum matrix A_G (6 x 24), momentum about the total com in world axes.
template <class S> void centroidalMomentumMatrix(const qmgpu_model& md, const Kin<S>& k, S A[6][NV]) {
  for (int d = 0; d < NV; ++d) {
    V3<S> hl, ha;
    for (int b = 0; b < NB; ++b) {
      V3<S> lin, ang;
      pointJacobianColumn(md, k, b, k.com[b], d, lin, ang);
      const V3<S> ml = S(md.mass[b]) * lin;
      hl = hl + ml;
      ha = ha + cross(k.com[b] - k.comTotal, ml) + k.Iw[b] * ang;
    }
    for (int a = 0; a < 3; ++a) { A[a][d] = hl[a]; A[3 + a][d] = ha[a]; }
  }
}

// Base velocity from normalized momentum: v_b = Ab^-1 (m*h - Aj*vj), with the block inverse OCS2 uses
// (upstream computeFloatingBaseCentroidalMomentumMatrixInverse).
template <class S> void baseVelocityFromMomentum(const qmgpu_model& md, const S A[6][NV], const S* hnorm, const S* vj, S vb[6]) {
  S rhs[6];
  for (int a = 0; a < 6; ++a) { S s = S(md.total_mass) * hnorm[a]; for (int j = 0; j < NJ; ++j) s -= A[a][6 + j] * vj[j]; rhs[a] = s; }
  M3<S> Ab22, Ab12;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { Ab22.m[i][j] = A[3 + i][3 + j]; Ab12.m[i][j] = A[i][3 + j]; }
  const M3<S> inv22 = inverse3(Ab22);
  const V3<S> w = inv22 * V3<S>(rhs[3], rhs[4], rhs[5]);
  const V3<S> t = Ab12 * w;
  const S im = S(1.0 / md.total_mass);
  for (int a = 0; a < 3; ++a) { vb[a] = im * (rhs[a] - t[a]); vb[3 + a] = w[a]; }
}

// Centroidal flow map (upstream PinocchioCentroidalDynamics::getValue, driven from
// qm_interface/src/dynamics/QMDynamicsAD.cpp:22-26).  Optionally returns foot velocities / EE pose.
// Force tracking (OWN FORMULATION, include/qmgpu.h qmgpu_settings::ee_contact_stiffness): the arm end-effector touches a compliant
// environment anchored at env, f_e = -K (p_ee - env) acts on the centroidal dynamics.  The node-level functions of qmo_mpc.h set the
// contact of the node they evaluate (thread local) before they call the flow map; K = 0 = no contact.
struct EeContact { double K = 0.0; double env[3] = {0.0, 0.0, 0.0}; double fref[3] = {0.0, 0.0, 0.0}; };
inline thread_local EeContact g_eeContact;

template <class S> struct FlowAux {
  V3<S> footPos[NCT], footVel[NCT];
  V3<S> eePos;
  M3<S> eeRot;
};
template <class S> void flowMap(const qmgpu_model& md, double gravity, const S* x, const S* u, S* f, FlowAux<S>* aux = nullptr) {
  Kin<S> k;
  forwardKinematics<S>(md, x + 6, k);
  static thread_local S A[6][NV];
  centroidalMomentumMatrix(md, k, A);
  S vb[6];
  baseVelocityFromMomentum(md, A, x, u + 12, vb);
  V3<S> fl(S(0.0), S(0.0), S(-gravity)), fa;
  const S im = S(1.0 / md.total_mass);
  for (int c = 0; c < NCT; ++c) {
    const V3<S> fc(u[3 * c], u[3 * c + 1], u[3 * c + 2]);
    fl = fl + im * fc;
    fa = fa + im * cross(k.foot[c] - k.comTotal, fc);
  }
  if (g_eeContact.K != 0.0) {
    const V3<S> fe = S(-g_eeContact.K) * (k.ee - V3<S>(S(g_eeContact.env[0]), S(g_eeContact.env[1]), S(g_eeContact.env[2])));
    fl = fl + im * fe;
    fa = fa + im * cross(k.ee - k.comTotal, fe);
  }
  for (int a = 0; a < 3; ++a) { f[a] = fl[a]; f[3 + a] = fa[a]; }
  for (int a = 0; a < 6; ++a) f[6 + a] = vb[a];
  for (int j = 0; j < NJ; ++j) f[12 + j] = u[12 + j];
  if (aux) {
    S v[NV];
    for (int a = 0; a < 6; ++a) v[a] = vb[a];
    for (int j = 0; j < NJ; ++j) v[6 + j] = u[12 + j];
    for (int c = 0; c < NCT; ++c) {
      aux->footPos[c] = k.foot[c];
      V3<S> vel;
      for (int d = 0; d < NV; ++d) { V3<S> lin, ang; pointJacobianColumn(md, k, md.foot_body[c], k.foot[c], d, lin, ang); vel = vel + v[d] * lin; }
      aux->footVel[c] = vel;
    }
    aux->eePos = k.ee;
    aux->eeRot = k.Ree;
  }
}

// Eigen::Quaternion(Matrix3) construction (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl),
// which ocs2::matrixToQuaternion forwards to.  Returns (x, y, z, w).
template <class S> void matrixToQuaternion(const M3<S>& R, S q[4]) {
  S t = R.m[0][0] + R.m[1][1] + R.m[2][2];
  if (t > 0.0) {
    t = sqrt(t + S(1.0));
    q[3] = S(0.5) * t;
    t = S(0.5) / t;
    q[0] = (R.m[2][1] - R.m[1][2]) * t;
    q[1] = (R.m[0][2] - R.m[2][0]) * t;
    q[2] = (R.m[1][0] - R.m[0][1]) * t;
  } else {
    int i = 0;
    if (R.m[1][1] > R.m[0][0]) i = 1;
    if (R.m[2][2] > R.m[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(R.m[i][i] - R.m[j][j] - R.m[k][k] + S(1.0));
    q[i] = S(0.5) * t;
    t = S(0.5) / t;
    q[3] = (R.m[k][j] - R.m[j][k]) * t;
    q[j] = (R.m[j][i] + R.m[i][j]) * t;
    q[k] = (R.m[k][i] + R.m[i][k]) * t;
  }
}

// ocs2::quaternionDistance(q, qRef) (upstream ocs2_robotic_tools RotationTransforms.h); quaternions (x,y,z,w).
template <class S> V3<S> quaternionDistance(const S q[4], const double qRef[4]) {
  const V3<S> qv(q[0], q[1], q[2]);
  const V3<S> rv = V3<S>(S(qRef[0]), S(qRef[1]), S(qRef[2]));
  return q[3] * rv - S(qRef[3]) * qv + cross(qv, rv);
}

}  // namespace qmo

namespace qmo {

// ------------------------------------------------------------------------------------------------ structured derivatives (timing-grade build)
// The same flow map and auxiliary kinematics as flowMap<Dual<60>>, by a second route: the map is LINEAR in the momentum, the joint
// rates and the contact forces and does not depend on the base position, so only the 21 configuration coordinates (zyx, q_j) are
// carried as dual directions; the 39 remaining columns are closed forms in VALUES the evaluation already holds (A_b^-1, A_j, the
// point Jacobians, (p_i - p_com) x).  tests/test_oracle_invariants.py checks that both routes give the same 60 columns; the
// -O3 -march=native baseline build (Makefile target fast) uses this one.
using D21 = Dual<21>;
using D60m = Dual<60>;

inline D60m liftToD60(const D21& s, const double hn[6], const double p[3], const double F[12], const double vj[18]) {
  D60m r(s.v);
  for (int k = 0; k < 6; ++k) r.d[k] = hn ? hn[k] : 0.0;
  for (int a = 0; a < 3; ++a) { r.d[6 + a] = p ? p[a] : 0.0; r.d[9 + a] = s.d[a]; }
  for (int j = 0; j < 18; ++j) { r.d[12 + j] = s.d[3 + j]; r.d[42 + j] = vj ? vj[j] : 0.0; }
  for (int i = 0; i < 12; ++i) r.d[30 + i] = F ? F[i] : 0.0;
  return r;
}

inline void flowMapStructured(const qmgpu_model& md, double gravity, const double* x, const double* u, D60m* f, FlowAux<D60m>* aux) {
  D21 q[NV];
  for (int a = 0; a < 3; ++a) { q[a] = D21(x[6 + a]); q[3 + a] = D21(x[9 + a]); q[3 + a].d[a] = 1.0; }
  for (int j = 0; j < NJ; ++j) { q[6 + j] = D21(x[12 + j]); q[6 + j].d[3 + j] = 1.0; }
  Kin<D21> k;
  forwardKinematics<D21>(md, q, k);
  static thread_local D21 A[6][NV];
  centroidalMomentumMatrix(md, k, A);
  D21 hn[6], vj[NJ], vb[6];
  for (int a = 0; a < 6; ++a) hn[a] = D21(x[a]);
  for (int j = 0; j < NJ; ++j) vj[j] = D21(u[12 + j]);
  baseVelocityFromMomentum(md, A, hn, vj, vb);
  // value of A and the closed-form columns of the base velocity: d vb / d hn_k = m Ab^-1 e_k, d vb / d vj_j = -Ab^-1 Aj e_j
  static thread_local double Av[6][NV];
  for (int a = 0; a < 6; ++a) for (int d = 0; d < NV; ++d) Av[a][d] = A[a][d].v;
  double vbHn[6][6], vbVj[6][NJ];
  {
    double e[6], zero18[NJ] = {0}, col[6], ev[NJ], zero6[6] = {0};
    for (int kk = 0; kk < 6; ++kk) { for (int a = 0; a < 6; ++a) e[a] = a == kk ? 1.0 : 0.0; baseVelocityFromMomentum<double>(md, Av, e, zero18, col); for (int a = 0; a < 6; ++a) vbHn[a][kk] = col[a]; }
    for (int j = 0; j < NJ; ++j) { for (int i = 0; i < NJ; ++i) ev[i] = i == j ? 1.0 : 0.0; baseVelocityFromMomentum<double>(md, Av, zero6, ev, col); for (int a = 0; a < 6; ++a) vbVj[a][j] = col[a]; }
  }
  const double im = 1.0 / md.total_mass;
  // momentum rates
  V3<D21> fl(D21(0.0), D21(0.0), D21(-gravity)), fa;
  double flF[3][12] = {{0}}, faF[3][12] = {{0}}, flP[3][3] = {{0}}, faP[3][3] = {{0}};
  for (int c = 0; c < NCT; ++c) {
    const V3<D21> fc(D21(u[3 * c]), D21(u[3 * c + 1]), D21(u[3 * c + 2]));
    const V3<D21> arm = k.foot[c] - k.comTotal;
    fl = fl + D21(im) * fc;
    fa = fa + D21(im) * cross(arm, fc);
    for (int a = 0; a < 3; ++a) {
      V3<double> e; e[a] = 1.0;
      const V3<double> t = cross(V3<double>(arm.x.v, arm.y.v, arm.z.v), e);
      flF[a][3 * c + a] = im;
      for (int r = 0; r < 3; ++r) faF[r][3 * c + a] = im * t[r];
    }
  }
  if (g_eeContact.K != 0.0) {   // force tracking: f_e = -K (p_ee - env), linear in the base position
    const V3<D21> fe = D21(-g_eeContact.K) * (k.ee - V3<D21>(D21(g_eeContact.env[0]), D21(g_eeContact.env[1]), D21(g_eeContact.env[2])));
    const V3<D21> arm = k.ee - k.comTotal;
    fl = fl + D21(im) * fe;
    fa = fa + D21(im) * cross(arm, fe);
    for (int a = 0; a < 3; ++a) {
      V3<double> e; e[a] = -g_eeContact.K;
      const V3<double> t = cross(V3<double>(arm.x.v, arm.y.v, arm.z.v), e);
      flP[a][a] = -g_eeContact.K * im;
      for (int r = 0; r < 3; ++r) faP[r][a] = im * t[r];
    }
  }
  for (int a = 0; a < 3; ++a) { f[a] = liftToD60(fl[a], nullptr, flP[a], flF[a], nullptr); f[3 + a] = liftToD60(fa[a], nullptr, faP[a], faF[a], nullptr); }
  for (int a = 0; a < 6; ++a) f[6 + a] = liftToD60(vb[a], vbHn[a], nullptr, nullptr, vbVj[a]);
  for (int j = 0; j < NJ; ++j) { f[12 + j] = D60m(u[12 + j]); f[12 + j].d[42 + j] = 1.0; }
  if (aux) {
    D21 v[NV];
    for (int a = 0; a < 6; ++a) v[a] = vb[a];
    for (int j = 0; j < NJ; ++j) v[6 + j] = vj[j];
    const double pId[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int c = 0; c < NCT; ++c) {
      V3<D21> vel;
      double velHn[3][6] = {{0}}, velVj[3][NJ] = {{0}};
      for (int d = 0; d < NV; ++d) {
        V3<D21> lin, ang;
        pointJacobianColumn(md, k, md.foot_body[c], k.foot[c], d, lin, ang);
        vel = vel + v[d] * lin;
        for (int r = 0; r < 3; ++r) {
          const double l = lin[r].v;
          if (d < 6) { for (int kk = 0; kk < 6; ++kk) velHn[r][kk] += l * vbHn[d][kk]; for (int j = 0; j < NJ; ++j) velVj[r][j] += l * vbVj[d][j]; }
          else velVj[r][d - 6] += l;
        }
      }
      for (int r = 0; r < 3; ++r) { aux->footPos[c][r] = liftToD60(k.foot[c][r], nullptr, pId[r], nullptr, nullptr); aux->footVel[c][r] = liftToD60(vel[r], velHn[r], nullptr, nullptr, velVj[r]); }
    }
    for (int r = 0; r < 3; ++r) aux->eePos[r] = liftToD60(k.ee[r], nullptr, pId[r], nullptr, nullptr);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) aux->eeRot.m[i][j] = liftToD60(k.Ree.m[i][j], nullptr, nullptr, nullptr, nullptr);
  }
}

// the derivative route of this build: flowMap<Dual<60>> (checker) or the structured one (timing-grade, -DQMO_FAST)
inline void flowMapD60(const qmgpu_model& md, double gravity, const D60m* xd, const D60m* ud, D60m* fd, FlowAux<D60m>* aux) {
#ifdef QMO_FAST
  double x[30], u[30];
  for (int i = 0; i < 30; ++i) { x[i] = xd[i].v; u[i] = ud[i].v; }
  flowMapStructured(md, gravity, x, u, fd, aux);
#else
  flowMap<D60m>(md, gravity, xd, ud, fd, aux);
#endif
}

}  // namespace qmo
