// The tree sweep shared by the LQ kernel (T = Du, lane-tangent derivatives) and the line-search kernel (T = double).
// Specialised to the AlienGo+Z1 topology (checked in qmgpu_create): bodies 1..12 are four 3-joint leg chains hanging off
// the base in the order LF, LH, RF, RH, bodies 13..18 the 6-joint arm chain; feet sit on bodies 3/6/9/12, the
// end-effector frame on body 18.
#pragma once
#include "model_dev.h"

namespace qmk {

// ---- leaf-to-root accumulation of one kinematic chain in JOINT-LOCAL frames ------------------------------------------------------
// Every joint origin has rpy = 0 and a coordinate axis, so crossing joint b is an elementary rotation E(q_b) about one axis plus a
// constant offset: rotating the subtree's composite quantities costs a 2-D rotation per vector and a 2x2 similarity for the inertia,
// instead of the general R I R^T per body of a root-to-leaf sweep in world axes (less than half the fp64 instructions).  Momentum is
// linear in the joint rates, so the momentum a joint contributes is just (composite inertia of its subtree) x (joint twist) -- no
// forward velocity pass is needed: one backward pass per chain yields mass moments, inertia, joint-induced momentum and the
// position / velocity (/ orientation) of the frame at the chain's tip, all in the frame of the chain's root and about its origin.
// Scalar types as in model_dev.h: P = configuration-only quantities, V = quantities also linear in the joint rates / momenta.
template <class P, class V> struct ChainAcc {
  real M;            // composite mass
  Vec3<P> h;           // first moment  sum m c
  Sym3<P> I;           // inertia about the current origin
  Vec3<V> l, k;        // momentum caused by the joint rates of the subtree (k about the current origin)
  Vec3<P> p;           // tip frame: position
  Vec3<V> vf;          // tip frame: velocity caused by the joint rates
};

template <class C, class T> __device__ __forceinline__ void rotAxis(int axis, C cs, C sn, Vec3<T>& v) {
  if (axis == 0) { const T a = v.y, b = v.z; v.y = cs * a - sn * b; v.z = sn * a + cs * b; }
  else if (axis == 1) { const T a = v.z, b = v.x; v.z = cs * a - sn * b; v.x = sn * a + cs * b; }
  else { const T a = v.x, b = v.y; v.x = cs * a - sn * b; v.y = sn * a + cs * b; }
}
template <class T> __device__ __forceinline__ Vec3<T> axisCross(int axis, const Vec3<T>& v) {  // e_axis x v
  if (axis == 0) return Vec3<T>(T(0.0_r), T(0.0_r) - v.z, v.y);
  if (axis == 1) return Vec3<T>(v.z, T(0.0_r), T(0.0_r) - v.x);
  return Vec3<T>(T(0.0_r) - v.y, v.x, T(0.0_r));
}
template <class T> __device__ __forceinline__ Vec3<T> symColumn(int axis, const Sym3<T>& S) {
  if (axis == 0) return Vec3<T>(S.xx, S.xy, S.xz);
  if (axis == 1) return Vec3<T>(S.xy, S.yy, S.yz);
  return Vec3<T>(S.xz, S.yz, S.zz);
}
// E S E^T for the rotation E about `axis`; (i, j) is the rotated plane, k the axis: entries II JJ IJ IK JK (KK unchanged)
template <class T> __device__ __forceinline__ void rotPlane(T c2, T s2, T cs, T sn, T& II, T& JJ, T& IJ, T& IK, T& JK) {
  const T d = 0.5_r * (II - JJ), m = 0.5_r * (II + JJ);
  const T t = d * c2 - IJ * s2;
  const T nij = d * s2 + IJ * c2;
  II = m + t; JJ = m - t; IJ = nij;
  const T a = IK, b = JK;
  IK = cs * a - sn * b; JK = sn * a + cs * b;
}
template <class T> __device__ __forceinline__ void rotSym(int axis, T cs, T sn, Sym3<T>& S) {
  const T c2 = cs * cs - sn * sn, s2 = 2.0_r * (cs * sn);
  if (axis == 0) rotPlane(c2, s2, cs, sn, S.yy, S.zz, S.yz, S.xy, S.xz);        // plane (y, z), axis x
  else if (axis == 1) rotPlane(c2, s2, cs, sn, S.zz, S.xx, S.xz, S.yz, S.xy);   // plane (z, x), axis y
  else rotPlane(c2, s2, cs, sn, S.xx, S.yy, S.xy, S.xz, S.yz);                  // plane (x, y), axis z
}

// add body b (its own frame = the current frame) to the composite: constants only
template <class P, class V> __device__ __forceinline__ void addBody(const ModelR& md, int b, ChainAcc<P, V>& c) {
  const real m = md.mass[b], cx = md.com[b][0], cy = md.com[b][1], cz = md.com[b][2];
  const real* in = md.inertia[b];
  c.M += m;
  c.h = c.h + Vec3<P>(P(m * cx), P(m * cy), P(m * cz));
  c.I.xx = c.I.xx + (in[0] + m * (cy * cy + cz * cz)); c.I.yy = c.I.yy + (in[3] + m * (cx * cx + cz * cz)); c.I.zz = c.I.zz + (in[5] + m * (cx * cx + cy * cy));
  c.I.xy = c.I.xy + (in[1] - m * cx * cy); c.I.xz = c.I.xz + (in[2] - m * cx * cz); c.I.yz = c.I.yz + (in[4] - m * cy * cz);
}

// cross joint b towards its parent: joint-rate contributions, rotation E(q), shift of the origin by the constant offset
// joint axes of the AlienGo+Z1 tree (checked against the loaded model in qmgpu_create): compile-time constants, so that the
// three-way axis selections below fold away and the sweep is straight-line code whose model-constant loads can be issued early
__device__ constexpr int LEG_AXIS[3] = {0, 1, 1};
__device__ constexpr int ARM_AXIS[6] = {2, 1, 1, 1, 2, 0};

template <class P, class V, class RotExtra> __device__ __forceinline__ void crossJoint(const ModelR& md, int b, int axis, P q, V qd, ChainAcc<P, V>& c, RotExtra&& rotExtra) {
  c.l = c.l + qd * axisCross(axis, c.h);
  c.k = c.k + qd * symColumn(axis, c.I);
  c.vf = c.vf + qd * axisCross(axis, c.p);
  P sn, cs;
  sincosT(q, sn, cs);
  rotAxis(axis, cs, sn, c.h); rotAxis(axis, cs, sn, c.l); rotAxis(axis, cs, sn, c.k); rotAxis(axis, cs, sn, c.p); rotAxis(axis, cs, sn, c.vf);
  rotSym(axis, cs, sn, c.I);
  rotExtra(axis, cs, sn);
  const real ox = md.joint_offset[b][0], oy = md.joint_offset[b][1], oz = md.joint_offset[b][2], M = c.M;
  // k about the new origin, inertia about the new origin (uses h about the old one), then h and the tip position
  c.k = c.k + Vec3<V>(oy * c.l.z - oz * c.l.y, oz * c.l.x - ox * c.l.z, ox * c.l.y - oy * c.l.x);
  c.I.xx = c.I.xx + (2.0_r * (oy * c.h.y + oz * c.h.z) + M * (oy * oy + oz * oz));
  c.I.yy = c.I.yy + (2.0_r * (ox * c.h.x + oz * c.h.z) + M * (ox * ox + oz * oz));
  c.I.zz = c.I.zz + (2.0_r * (ox * c.h.x + oy * c.h.y) + M * (ox * ox + oy * oy));
  c.I.xy = c.I.xy - ((ox * c.h.y + oy * c.h.x) + M * ox * oy);
  c.I.xz = c.I.xz - ((ox * c.h.z + oz * c.h.x) + M * ox * oz);
  c.I.yz = c.I.yz - ((oy * c.h.z + oz * c.h.y) + M * oy * oz);
  c.h = c.h + Vec3<P>(P(M * ox), P(M * oy), P(M * oz));
  c.p = c.p + Vec3<P>(P(ox), P(oy), P(oz));
}

template <class T> __device__ __forceinline__ Sym3<T> similarity(const Mat3<T>& R, const Sym3<T>& S) {  // R S R^T
  const Vec3<T> a0 = S.xx * R.c0 + S.xy * R.c1 + S.xz * R.c2;   // (R S) column 0
  const Vec3<T> a1 = S.xy * R.c0 + S.yy * R.c1 + S.yz * R.c2;
  const Vec3<T> a2 = S.xz * R.c0 + S.yz * R.c1 + S.zz * R.c2;
  Sym3<T> W;
  W.xx = a0.x * R.c0.x + a1.x * R.c1.x + a2.x * R.c2.x;
  W.xy = a0.x * R.c0.y + a1.x * R.c1.y + a2.x * R.c2.y;
  W.xz = a0.x * R.c0.z + a1.x * R.c1.z + a2.x * R.c2.z;
  W.yy = a0.y * R.c0.y + a1.y * R.c1.y + a2.y * R.c2.y;
  W.yz = a0.y * R.c0.z + a1.y * R.c1.z + a2.y * R.c2.z;
  W.zz = a0.z * R.c0.z + a1.z * R.c1.z + a2.z * R.c2.z;
  return W;
}

// In must provide: V hn(i) i<6 ; P euler(i) i<3 ; P q(j), V qd(j) j<18 (joint order) ; Vec3<F> force(c) c<4 (contact order)
// onEE(r_ee_rel_base, R_ee) is called once and returns the external force acting on the end-effector (Vec3<F>, zero without force
// tracking); onFoot(c, r_rel_base, v_joint_only) is called four times (world axes, relative to the base origin).
template <class P, class V, class F, class In, class FootFn, class EeFn>
__device__ __forceinline__ void centroidalSweep2(const ModelR& md, real gravity, const In& in, FootFn&& onFoot, EeFn&& onEE, FlowOut<P, V, F>& f, BaseMotion2<P, V>& bm) {
  P sz, cz, sy, cy;
  Mat3<P> R0;
  baseRotation(in.euler(0), in.euler(1), in.euler(2), R0, sz, cz, sy, cy);
  // totals in the base frame, about the base origin; start with the base body itself
  ChainAcc<P, V> tot;
  tot.M = 0.0_r;
  addBody(md, 0, tot);
  Vec3<F> fsum;
  Vec3<ProdT<P, F>> tsum;
  auto absorb = [&](const ChainAcc<P, V>& c) {
    tot.M += c.M; tot.h = tot.h + c.h; tot.l = tot.l + c.l; tot.k = tot.k + c.k;
    tot.I.xx = tot.I.xx + c.I.xx; tot.I.xy = tot.I.xy + c.I.xy; tot.I.xz = tot.I.xz + c.I.xz; tot.I.yy = tot.I.yy + c.I.yy; tot.I.yz = tot.I.yz + c.I.yz; tot.I.zz = tot.I.zz + c.I.zz;
  };
  {  // arm: bodies 18 .. 13, tip = end-effector frame (its orientation is carried along)
    ChainAcc<P, V> c;
    c.M = 0.0_r;
    c.p = Vec3<P>(P(md.ee_offset[0]), P(md.ee_offset[1]), P(md.ee_offset[2]));
    Mat3<P> Re;
    Re.c0 = Vec3<P>(P(1.0_r), P(0.0_r), P(0.0_r)); Re.c1 = Vec3<P>(P(0.0_r), P(1.0_r), P(0.0_r)); Re.c2 = Vec3<P>(P(0.0_r), P(0.0_r), P(1.0_r));
#pragma unroll
    for (int a = 5; a >= 0; --a) {
      addBody(md, 13 + a, c);
      crossJoint(md, 13 + a, ARM_AXIS[a], in.q(12 + a), in.qd(12 + a), c, [&](int axis, P cs, P sn) { rotAxis(axis, cs, sn, Re.c0); rotAxis(axis, cs, sn, Re.c1); rotAxis(axis, cs, sn, Re.c2); });
    }
    absorb(c);
    Mat3<P> Rw;
    Rw.c0 = mul(R0, Re.c0); Rw.c1 = mul(R0, Re.c1); Rw.c2 = mul(R0, Re.c2);
    const Vec3<P> rEE = mul(R0, c.p);
    const Vec3<F> fe = onEE(rEE, Rw);
    fsum = fe;
    tsum = cross(rEE, fe);
  }
#pragma unroll 1
  for (int leg = 0; leg < 4; ++leg) {
    int cft = 0;
    for (int k = 1; k < 4; ++k) if (md.foot_body[k] == 3 + 3 * leg) cft = k;
    ChainAcc<P, V> c;
    c.M = 0.0_r;
    c.p = Vec3<P>(P(md.foot_offset[cft][0]), P(md.foot_offset[cft][1]), P(md.foot_offset[cft][2]));
#pragma unroll
    for (int j = 2; j >= 0; --j) {
      addBody(md, 1 + 3 * leg + j, c);
      crossJoint(md, 1 + 3 * leg + j, LEG_AXIS[j], in.q(3 * leg + j), in.qd(3 * leg + j), c, [](int, P, P) {});
    }
    absorb(c);
    const Vec3<P> r = mul(R0, c.p);
    const Vec3<V> v = mul(R0, c.vf);
    const Vec3<F> Fc = in.force(cft);
    fsum = fsum + Fc;
    tsum = tsum + cross(r, Fc);
    onFoot(cft, r, v);
  }
  Accum2<P, V> acc;
  acc.M1 = mul(R0, tot.h);
  acc.hl = mul(R0, tot.l);
  acc.ha = mul(R0, tot.k);
  acc.Io = similarity(R0, tot.I);
  V hn[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) hn[i] = in.hn(i);
  closeSweep2<P, V, F>(md, gravity, acc, hn, fsum, tsum, sz, cz, sy, cy, f, bm);
}
// ---- the sweep of ad_node_kernel, round 4: every lane walks ONLY the kinematic chain of its own direction -------------------------------------
// A joint tangent is non-zero in its own chain only (the chains are accumulated in the base frame; the base Euler angles enter afterwards through R0), so of
// the 18 joint crossings centroidalSweep2 makes in every lane, 15 (leg lanes) or 12 (arm lanes) multiply zero tangents, and the three base-angle lanes need no
// chain tangent at all.  Here the arm lanes (dd 15..20) walk the arm, every other lane ONE leg ((dd - 3) / 3; the base-angle lanes leg 0, as passengers): two
// divergent regions of 6 and 3 crossings instead of 18 in every lane.  The PRIMAL composites of the five chains are published through LDS by one lane each
// (`pub`: this node's 4 x 22 + 31 doubles; the publishing lane executes the same instruction stream on the same inputs as every other lane of its chain, so the
// published values are the ones those lanes hold) and the totals are the published primal parts plus the lane's own tangents.
constexpr int SWEEP_PUB_LEG = 22, SWEEP_PUB_ARM = 31, SWEEP_PUB_NODE = 4 * SWEEP_PUB_LEG + SWEEP_PUB_ARM;
template <class In, class FootFn, class EeFn>
__device__ __forceinline__ void centroidalSweepOwnChain(const ModelR& md, real gravity, const In& in, int dd, real* pub, FootFn&& onFoot, EeFn&& onEE, FlowOut<Du, Du3, Du3>& f,
                                                        BaseMotion2<Du, Du3>& bm) {
  using P = Du; using V = Du3;
  P sz, cz, sy, cy;
  Mat3<P> R0;
  baseRotation(in.euler(0), in.euler(1), in.euler(2), R0, sz, cz, sy, cy);
  const bool isArm = dd >= 15, isLeg = dd >= 3 && dd < 15;
  const int myLeg = isLeg ? (dd - 3) / 3 : 0;
  ChainAcc<P, V> c;
  c.M = 0.0_r;
  Mat3<P> Re;
  Re.c0 = Vec3<P>(P(1.0_r), P(0.0_r), P(0.0_r)); Re.c1 = Vec3<P>(P(0.0_r), P(1.0_r), P(0.0_r)); Re.c2 = Vec3<P>(P(0.0_r), P(0.0_r), P(1.0_r));
  int cftMine = 0;
  if (isArm) {
    c.p = Vec3<P>(P(md.ee_offset[0]), P(md.ee_offset[1]), P(md.ee_offset[2]));
#pragma unroll
    for (int a = 5; a >= 0; --a) {
      addBody(md, 13 + a, c);
      crossJoint(md, 13 + a, ARM_AXIS[a], in.q(12 + a), in.qd(12 + a), c, [&](int axis, P cs, P sn) { rotAxis(axis, cs, sn, Re.c0); rotAxis(axis, cs, sn, Re.c1); rotAxis(axis, cs, sn, Re.c2); });
    }
  } else {
    for (int k = 1; k < 4; ++k) if (md.foot_body[k] == 3 + 3 * myLeg) cftMine = k;
    c.p = Vec3<P>(P(md.foot_offset[cftMine][0]), P(md.foot_offset[cftMine][1]), P(md.foot_offset[cftMine][2]));
#pragma unroll
    for (int j = 2; j >= 0; --j) {
      addBody(md, 1 + 3 * myLeg + j, c);
      crossJoint(md, 1 + 3 * myLeg + j, LEG_AXIS[j], in.q(3 * myLeg + j), in.qd(3 * myLeg + j), c, [](int, P, P) {});
    }
  }
  // publish the primal composites: the first lane of each chain (legs dd = 3 + 3 leg, arm dd = 15)
  if ((isLeg && dd == 3 + 3 * myLeg) || dd == 15) {
    real* p = pub + (isArm ? 4 * SWEEP_PUB_LEG : myLeg * SWEEP_PUB_LEG);
    p[0] = c.M; p[1] = c.h.x.v; p[2] = c.h.y.v; p[3] = c.h.z.v;
    p[4] = c.I.xx.v; p[5] = c.I.xy.v; p[6] = c.I.xz.v; p[7] = c.I.yy.v; p[8] = c.I.yz.v; p[9] = c.I.zz.v;
    p[10] = c.l.x.v; p[11] = c.l.y.v; p[12] = c.l.z.v; p[13] = c.k.x.v; p[14] = c.k.y.v; p[15] = c.k.z.v;
    p[16] = c.p.x.v; p[17] = c.p.y.v; p[18] = c.p.z.v; p[19] = c.vf.x.v; p[20] = c.vf.y.v; p[21] = c.vf.z.v;
    if (isArm) { p[22] = Re.c0.x.v; p[23] = Re.c0.y.v; p[24] = Re.c0.z.v; p[25] = Re.c1.x.v; p[26] = Re.c1.y.v; p[27] = Re.c1.z.v; p[28] = Re.c2.x.v; p[29] = Re.c2.y.v; p[30] = Re.c2.z.v; }
  }
  QM_WAVE_SYNC();
  // the lane's own tangents (zero for the base-angle lanes: their chain carried none)
  const real on = (isArm || isLeg) ? 1.0_r : 0.0_r;
  // totals in the base frame, about the base origin: the base body + the five published chains (values) + this lane's tangents
  ChainAcc<P, V> tot;
  tot.M = 0.0_r;
  addBody(md, 0, tot);
  {
    real s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = 0.0_r;
#pragma unroll
    for (int ch = 0; ch < 5; ++ch) {
      const real* p = pub + ch * SWEEP_PUB_LEG;
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] += p[i];
    }
    tot.M += s[0];
    tot.h = tot.h + Vec3<P>(P(s[1], on * c.h.x.d), P(s[2], on * c.h.y.d), P(s[3], on * c.h.z.d));
    tot.I.xx = tot.I.xx + P(s[4], on * c.I.xx.d); tot.I.xy = tot.I.xy + P(s[5], on * c.I.xy.d); tot.I.xz = tot.I.xz + P(s[6], on * c.I.xz.d);
    tot.I.yy = tot.I.yy + P(s[7], on * c.I.yy.d); tot.I.yz = tot.I.yz + P(s[8], on * c.I.yz.d); tot.I.zz = tot.I.zz + P(s[9], on * c.I.zz.d);
    tot.l = tot.l + Vec3<V>(V(s[10], on * c.l.x.d, on * c.l.x.e), V(s[11], on * c.l.y.d, on * c.l.y.e), V(s[12], on * c.l.z.d, on * c.l.z.e));
    tot.k = tot.k + Vec3<V>(V(s[13], on * c.k.x.d, on * c.k.x.e), V(s[14], on * c.k.y.d, on * c.k.y.e), V(s[15], on * c.k.z.d, on * c.k.z.e));
  }
  Vec3<Du3> fsum;
  Vec3<Du3> tsum;
  {  // arm tip = end-effector frame
    const real* p = pub + 4 * SWEEP_PUB_LEG;
    const real oa = isArm ? 1.0_r : 0.0_r;
    const Vec3<P> pe(P(p[16], oa * c.p.x.d), P(p[17], oa * c.p.y.d), P(p[18], oa * c.p.z.d));
    Mat3<P> Rl;
    Rl.c0 = Vec3<P>(P(p[22], oa * Re.c0.x.d), P(p[23], oa * Re.c0.y.d), P(p[24], oa * Re.c0.z.d));
    Rl.c1 = Vec3<P>(P(p[25], oa * Re.c1.x.d), P(p[26], oa * Re.c1.y.d), P(p[27], oa * Re.c1.z.d));
    Rl.c2 = Vec3<P>(P(p[28], oa * Re.c2.x.d), P(p[29], oa * Re.c2.y.d), P(p[30], oa * Re.c2.z.d));
    Mat3<P> Rw;
    Rw.c0 = mul(R0, Rl.c0); Rw.c1 = mul(R0, Rl.c1); Rw.c2 = mul(R0, Rl.c2);
    const Vec3<P> rEE = mul(R0, pe);
    const Vec3<Du3> fe = onEE(rEE, Rw);
    fsum = fe;
    tsum = cross(rEE, fe);
  }
#pragma unroll 1
  for (int leg = 0; leg < 4; ++leg) {
    int cft = 0;
    for (int k = 1; k < 4; ++k) if (md.foot_body[k] == 3 + 3 * leg) cft = k;
    const real* p = pub + leg * SWEEP_PUB_LEG;
    const real ol = (isLeg && myLeg == leg) ? 1.0_r : 0.0_r;
    const Vec3<P> pf(P(p[16], ol * c.p.x.d), P(p[17], ol * c.p.y.d), P(p[18], ol * c.p.z.d));
    const Vec3<V> vl(V(p[19], ol * c.vf.x.d, ol * c.vf.x.e), V(p[20], ol * c.vf.y.d, ol * c.vf.y.e), V(p[21], ol * c.vf.z.d, ol * c.vf.z.e));
    const Vec3<P> r = mul(R0, pf);
    const Vec3<V> v = mul(R0, vl);
    const Vec3<Du3> Fc = in.force(cft);
    fsum = fsum + Fc;
    tsum = tsum + cross(r, Fc);
    onFoot(cft, r, v);
  }
  Accum2<P, V> acc;
  acc.M1 = mul(R0, tot.h);
  acc.hl = mul(R0, tot.l);
  acc.ha = mul(R0, tot.k);
  acc.Io = similarity(R0, tot.I);
  V hn[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) hn[i] = in.hn(i);
  closeSweep2<P, V, Du3>(md, gravity, acc, hn, fsum, tsum, sz, cz, sy, cy, f, bm);
}

// single scalar type (plain evaluation with T = double; T = Du differentiates along one direction per lane through everything)
template <class T, class In, class FootFn, class EeFn>
__device__ __forceinline__ void centroidalSweep(const ModelR& md, real gravity, const In& in, FootFn&& onFoot, EeFn&& onEE, T f[12], BaseMotion<T>& bm) {
  FlowOut<T, T, T> o;
  centroidalSweep2<T, T, T>(md, gravity, in, onFoot, onEE, o, bm);
#pragma unroll
  for (int i = 0; i < 3; ++i) { f[i] = o.lin[i]; f[3 + i] = o.ang[i]; }
#pragma unroll
  for (int i = 0; i < 6; ++i) f[6 + i] = o.kin[i];
}

// four feet held in named registers (a register array indexed by a runtime contact index would be demoted to scratch)
template <class T> struct Feet {
  Vec3<T> r0, r1, r2, r3, v0, v1, v2, v3;
  __device__ __forceinline__ void set(int c, Vec3<T> r, Vec3<T> v) {
    if (c == 0) { r0 = r; v0 = v; } else if (c == 1) { r1 = r; v1 = v; } else if (c == 2) { r2 = r; v2 = v; } else { r3 = r; v3 = v; }
  }
  __device__ __forceinline__ Vec3<T> r(int c) const { return c == 0 ? r0 : (c == 1 ? r1 : (c == 2 ? r2 : r3)); }
  __device__ __forceinline__ Vec3<T> v(int c) const { return c == 0 ? v0 : (c == 1 ? v1 : (c == 2 ? v2 : v3)); }
};

}  // namespace qmk
