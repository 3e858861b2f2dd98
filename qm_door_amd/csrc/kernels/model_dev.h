// Centroidal dynamics of the 24-DoF AlienGo+Z1 model as ONE streaming sweep over the kinematic tree.
//
// What it replaces: the CppAD-generated flow map of upstream PinocchioCentroidalDynamicsAD driven from
// qm_interface/src/dynamics/QMDynamicsAD.cpp:22-33 and the Pinocchio passes of
// qm_interface/src/QMPreComputation.cpp:73-88 (forwardKinematics, computeJointJacobians, computeCentroidalMap, ...).
//
// Formulation (differs on purpose from the oracle's per-column Jacobian sums; both are the same mathematics):
//   * every joint origin has rpy = 0 and a coordinate axis, so a joint only mixes two columns of the parent rotation
//   * positions are kept relative to the base origin in world axes (the dynamics is translation invariant)
//   * the sweep accumulates  sum m c,  inertia about the base origin,  and the momentum produced by the joint
//     velocities alone; the 6x6 base block of the centroidal momentum matrix is then inverted in closed form:
//         omega = I_c^-1 (m h_ang - h_ang^J) ,  dp = (m h_lin - h_lin^J)/m - omega x c ,  d(zyx) = T(zyx)^-1 omega
//     which is exactly OCS2's block inverse of A_b (upstream computeFloatingBaseCentroidalMomentumMatrixInverse).
//   * bodies are visited in index order and every parent is either the previous body or the base (checked at
//     qmgpu_create), so only the current chain state lives in registers.
// Templated on the scalar: T = Du (lane-tangent forward mode, see du.h) or T = double.
#pragma once
#include "problem_r.h"
#include "du.h"

namespace qmk {

template <class T> struct ChainState {
  Mat3<T> R;
  Vec3<T> r, w, vo;  // origin (relative to base origin), angular / origin velocity due to joint rates only
};
template <class T> struct Accum {
  Vec3<T> M1, hl, ha;
  Sym3<T> Io;
};

template <class T> __device__ __forceinline__ void accumulateBody(const ModelR& md, int b, const ChainState<T>& s, Accum<T>& acc) {
  const real m = md.mass[b];
  const Vec3<T> lc = mul(s.R, md.com[b][0], md.com[b][1], md.com[b][2]);
  const Vec3<T> c = s.r + lc;
  const Vec3<T> vc = s.vo + cross(s.w, lc);
  // world inertia about the body com: R I R^T
  const real ixx = md.inertia[b][0], ixy = md.inertia[b][1], ixz = md.inertia[b][2], iyy = md.inertia[b][3], iyz = md.inertia[b][4], izz = md.inertia[b][5];
  const Vec3<T> a0 = scale(ixx, s.R.c0) + scale(ixy, s.R.c1) + scale(ixz, s.R.c2);  // (R I) column 0
  const Vec3<T> a1 = scale(ixy, s.R.c0) + scale(iyy, s.R.c1) + scale(iyz, s.R.c2);
  const Vec3<T> a2 = scale(ixz, s.R.c0) + scale(iyz, s.R.c1) + scale(izz, s.R.c2);
  Sym3<T> Iw;
  Iw.xx = a0.x * s.R.c0.x + a1.x * s.R.c1.x + a2.x * s.R.c2.x;
  Iw.xy = a0.x * s.R.c0.y + a1.x * s.R.c1.y + a2.x * s.R.c2.y;
  Iw.xz = a0.x * s.R.c0.z + a1.x * s.R.c1.z + a2.x * s.R.c2.z;
  Iw.yy = a0.y * s.R.c0.y + a1.y * s.R.c1.y + a2.y * s.R.c2.y;
  Iw.yz = a0.y * s.R.c0.z + a1.y * s.R.c1.z + a2.y * s.R.c2.z;
  Iw.zz = a0.z * s.R.c0.z + a1.z * s.R.c1.z + a2.z * s.R.c2.z;
  acc.M1 = acc.M1 + scale(m, c);
  const T cc = dot(c, c);
  acc.Io.xx = acc.Io.xx + Iw.xx + m * (cc - c.x * c.x);
  acc.Io.yy = acc.Io.yy + Iw.yy + m * (cc - c.y * c.y);
  acc.Io.zz = acc.Io.zz + Iw.zz + m * (cc - c.z * c.z);
  acc.Io.xy = acc.Io.xy + Iw.xy - m * (c.x * c.y);
  acc.Io.xz = acc.Io.xz + Iw.xz - m * (c.x * c.z);
  acc.Io.yz = acc.Io.yz + Iw.yz - m * (c.y * c.z);
  const Vec3<T> mv = scale(m, vc);
  acc.hl = acc.hl + mv;
  acc.ha = acc.ha + mul(Iw, s.w) + cross(c, mv);
}

// Advance the chain state from the parent of body b to body b (joint angle q, joint rate qd), then accumulate.
template <class T> __device__ __forceinline__ void bodyStep(const ModelR& md, int b, T q, T qd, ChainState<T>& s, Accum<T>& acc) {
  const Vec3<T> off = mul(s.R, md.joint_offset[b][0], md.joint_offset[b][1], md.joint_offset[b][2]);
  s.vo = s.vo + cross(s.w, off);
  s.r = s.r + off;
  T sn, cs;
  sincosT(q, sn, cs);
  const int axis = md.axis[b];
  if (axis == 0) {
    s.w = s.w + qd * s.R.c0;
    const Vec3<T> n1 = cs * s.R.c1 + sn * s.R.c2, n2 = cs * s.R.c2 - sn * s.R.c1;
    s.R.c1 = n1; s.R.c2 = n2;
  } else if (axis == 1) {
    s.w = s.w + qd * s.R.c1;
    const Vec3<T> n2 = cs * s.R.c2 + sn * s.R.c0, n0 = cs * s.R.c0 - sn * s.R.c2;
    s.R.c2 = n2; s.R.c0 = n0;
  } else {
    s.w = s.w + qd * s.R.c2;
    const Vec3<T> n0 = cs * s.R.c0 + sn * s.R.c1, n1 = cs * s.R.c1 - sn * s.R.c0;
    s.R.c0 = n0; s.R.c1 = n1;
  }
  accumulateBody(md, b, s, acc);
}

// Base rotation Rz(yaw) Ry(pitch) Rx(roll) by columns; also returns sin/cos of yaw and pitch for the Euler-rate map.
template <class T> __device__ __forceinline__ void baseRotation(T yaw, T pitch, T roll, Mat3<T>& R, T& sz, T& cz, T& sy, T& cy) {
  T sx, cx;
  sincosT(yaw, sz, cz); sincosT(pitch, sy, cy); sincosT(roll, sx, cx);
  R.c0 = Vec3<T>(cz * cy, sz * cy, T(0.0_r) - sy);
  R.c1 = Vec3<T>(cz * sy * sx - sz * cx, sz * sy * sx + cz * cx, cy * sx);
  R.c2 = Vec3<T>(cz * sy * cx + sz * sx, sz * sy * cx - cz * sx, cy * cx);
}

template <class A, class B> __device__ __forceinline__ Vec3<ProdT<A, B>> solveSym3(const Sym3<A>& S, Vec3<B> b) {
  const A c00 = S.yy * S.zz - S.yz * S.yz;
  const A c01 = S.yz * S.xz - S.xy * S.zz;
  const A c02 = S.xy * S.yz - S.yy * S.xz;
  const A det = S.xx * c00 + S.xy * c01 + S.xz * c02;
  const A c11 = S.xx * S.zz - S.xz * S.xz;
  const A c12 = S.xy * S.xz - S.xx * S.yz;
  const A c22 = S.xx * S.yy - S.xy * S.xy;
  const A id = 1.0_r / det;
  return Vec3<ProdT<A, B>>((c00 * b.x + c01 * b.y + c02 * b.z) * id, (c01 * b.x + c11 * b.y + c12 * b.z) * id, (c02 * b.x + c12 * b.y + c22 * b.z) * id);
}

// Closes the sweep: given the accumulators, the normalized momentum hn (6), the summed contact force and its torque about
// the base origin, produce f[0..11] = [d(h_lin/m), d(h_ang/m), dp_base, d(zyx)] and the base twist (dp, omega) + com.
// Three scalar types (du.h): P for quantities that depend on the configuration only, V for quantities that are also linear in a
// velocity-like argument (joint rates, momentum), F for the contact forces.  Plain evaluation: P = V = F = double.
template <class P, class V> struct Accum2 {
  Vec3<P> M1;
  Vec3<V> hl, ha;
  Sym3<P> Io;
};
template <class P, class V> struct BaseMotion2 {
  Vec3<V> dp, omega;
  Vec3<P> com;
};
template <class P, class V, class F> struct FlowOut {
  ProdT<real, F> lin[3];   // d(h_lin / m)
  ProdT<P, F> ang[3];        // d(h_ang / m)
  V kin[6];                  // base position rates, Euler ZYX rates
};
template <class P, class V, class F>
__device__ __forceinline__ void closeSweep2(const ModelR& md, real gravity, const Accum2<P, V>& acc, const V hn[6], Vec3<F> fsum, Vec3<ProdT<P, F>> tsum, P sz, P cz, P sy, P cy,
                                            FlowOut<P, V, F>& f, BaseMotion2<P, V>& bm) {
  const real m = md.total_mass, im = 1.0_r / md.total_mass;
  const Vec3<P> cm = scale(im, acc.M1);
  const P cc = dot(cm, cm);
  Sym3<P> Ic;
  Ic.xx = acc.Io.xx - m * (cc - cm.x * cm.x);
  Ic.yy = acc.Io.yy - m * (cc - cm.y * cm.y);
  Ic.zz = acc.Io.zz - m * (cc - cm.z * cm.z);
  Ic.xy = acc.Io.xy + m * (cm.x * cm.y);
  Ic.xz = acc.Io.xz + m * (cm.x * cm.z);
  Ic.yz = acc.Io.yz + m * (cm.y * cm.z);
  const Vec3<V> haC = acc.ha - cross(cm, acc.hl);
  const Vec3<V> rhsL = Vec3<V>(m * hn[0], m * hn[1], m * hn[2]) - acc.hl;
  const Vec3<V> rhsA = Vec3<V>(m * hn[3], m * hn[4], m * hn[5]) - haC;
  const Vec3<V> om = solveSym3(Ic, rhsA);
  const Vec3<V> dp = scale(im, rhsL) - cross(om, cm);
  const V tmp = (cz * om.x + sz * om.y) / cy;
  f.lin[0] = im * fsum.x; f.lin[1] = im * fsum.y; f.lin[2] = im * fsum.z - gravity;
  const Vec3<ProdT<P, F>> ta = scale(im, tsum - cross(cm, fsum));
  f.ang[0] = ta.x; f.ang[1] = ta.y; f.ang[2] = ta.z;
  f.kin[0] = dp.x; f.kin[1] = dp.y; f.kin[2] = dp.z;
  f.kin[3] = sy * tmp + om.z; f.kin[4] = cz * om.y - sz * om.x; f.kin[5] = tmp;
  bm.dp = dp; bm.omega = om; bm.com = cm;
}
template <class T> using BaseMotion = BaseMotion2<T, T>;
template <class T>
__device__ __forceinline__ void closeSweep(const ModelR& md, real gravity, const Accum<T>& acc, const T hn[6], Vec3<T> fsum, Vec3<T> tsum, T sz, T cz, T sy, T cy,
                                           T f[12], BaseMotion<T>& bm) {
  Accum2<T, T> a2;
  a2.M1 = acc.M1; a2.hl = acc.hl; a2.ha = acc.ha; a2.Io = acc.Io;
  FlowOut<T, T, T> o;
  closeSweep2<T, T, T>(md, gravity, a2, hn, fsum, tsum, sz, cz, sy, cy, o, bm);
#pragma unroll
  for (int i = 0; i < 3; ++i) { f[i] = o.lin[i]; f[3 + i] = o.ang[i]; }
#pragma unroll
  for (int i = 0; i < 6; ++i) f[6 + i] = o.kin[i];
}

// Eigen::Quaternion(Matrix3) (what ocs2::matrixToQuaternion forwards to); q = (x, y, z, w).  Branches on primal values only.
template <class T> __device__ __forceinline__ void matrixToQuaternion(const Mat3<T>& R, T q[4]) {
  // R(i,j): row i of column j
  const T r00 = R.c0.x, r10 = R.c0.y, r20 = R.c0.z, r01 = R.c1.x, r11 = R.c1.y, r21 = R.c1.z, r02 = R.c2.x, r12 = R.c2.y, r22 = R.c2.z;
  T t = r00 + r11 + r22;
  if (val(t) > 0.0_r) {
    t = sqrtT(t + 1.0_r);
    q[3] = 0.5_r * t;
    t = 0.5_r / t;
    q[0] = (r21 - r12) * t; q[1] = (r02 - r20) * t; q[2] = (r10 - r01) * t;
  } else if (val(r00) >= val(r11) && val(r00) >= val(r22)) {  // i = 0, j = 1, k = 2
    t = sqrtT(r00 - r11 - r22 + 1.0_r);
    q[0] = 0.5_r * t; t = 0.5_r / t;
    q[3] = (r21 - r12) * t; q[1] = (r10 + r01) * t; q[2] = (r20 + r02) * t;
  } else if (val(r11) > val(r00) && val(r11) >= val(r22)) {  // i = 1, j = 2, k = 0
    t = sqrtT(r11 - r22 - r00 + 1.0_r);
    q[1] = 0.5_r * t; t = 0.5_r / t;
    q[3] = (r02 - r20) * t; q[2] = (r21 + r12) * t; q[0] = (r01 + r10) * t;
  } else {  // i = 2, j = 0, k = 1
    t = sqrtT(r22 - r00 - r11 + 1.0_r);
    q[2] = 0.5_r * t; t = 0.5_r / t;
    q[3] = (r10 - r01) * t; q[0] = (r02 + r20) * t; q[1] = (r12 + r21) * t;
  }
}

// ocs2::quaternionDistance(q, qRef) = q.w qRef.vec - qRef.w q.vec + q.vec x qRef.vec
template <class T> __device__ __forceinline__ Vec3<T> quaternionDistance(const T q[4], const real r[4]) {
  const Vec3<T> qv(q[0], q[1], q[2]);
  const Vec3<T> c(q[1] * r[2] - q[2] * r[1], q[2] * r[0] - q[0] * r[2], q[0] * r[1] - q[1] * r[0]);
  return Vec3<T>(q[3] * r[0] - r[3] * qv.x + c.x, q[3] * r[1] - r[3] * qv.y + c.y, q[3] * r[2] - r[3] * qv.z + c.z);
}

}  // namespace qmk
