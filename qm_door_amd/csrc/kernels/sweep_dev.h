// The tree sweep shared by the LQ kernel (T = Du, lane-tangent derivatives) and the line-search kernel (T = double).
// Specialised to the AlienGo+Z1 topology (checked in qmgpu_create): bodies 1..12 are four 3-joint leg chains hanging off
// the base in the order LF, LH, RF, RH, bodies 13..18 the 6-joint arm chain; feet sit on bodies 3/6/9/12, the
// end-effector frame on body 18.
#pragma once
#include "model_dev.h"

namespace qmk {

// In must provide: T hn(i) i<6 ; T euler(i) i<3 ; T q(j), T qd(j) j<18 (joint order) ; Vec3<T> force(c) c<4 (contact order)
// onEE(r_ee_rel_base, R_ee) is called once, onFoot(c, r_rel_base, v_joint_only, force) four times.
template <class T, class In, class FootFn, class EeFn>
__device__ __forceinline__ void centroidalSweep(const qmgpu_model& md, double gravity, const In& in, FootFn&& onFoot, EeFn&& onEE, T f[12], BaseMotion<T>& bm) {
  T sz, cz, sy, cy;
  ChainState<T> base;
  baseRotation(in.euler(0), in.euler(1), in.euler(2), base.R, sz, cz, sy, cy);
  Accum<T> acc;
  accumulateBody(md, 0, base, acc);
  {
    ChainState<T> s = base;
#pragma unroll 1
    for (int a = 0; a < 6; ++a) bodyStep(md, 13 + a, in.q(12 + a), in.qd(12 + a), s, acc);
    onEE(s.r + mul(s.R, md.ee_offset[0], md.ee_offset[1], md.ee_offset[2]), s.R);
  }
  Vec3<T> fsum, tsum;
#pragma unroll 1
  for (int leg = 0; leg < 4; ++leg) {
    ChainState<T> s = base;
#pragma unroll
    for (int j = 0; j < 3; ++j) bodyStep(md, 1 + 3 * leg + j, in.q(3 * leg + j), in.qd(3 * leg + j), s, acc);
    int c = 0;
    for (int k = 1; k < 4; ++k) if (md.foot_body[k] == 3 + 3 * leg) c = k;
    const Vec3<T> lo = mul(s.R, md.foot_offset[c][0], md.foot_offset[c][1], md.foot_offset[c][2]);
    const Vec3<T> r = s.r + lo, v = s.vo + cross(s.w, lo);
    const Vec3<T> F = in.force(c);
    fsum = fsum + F;
    tsum = tsum + cross(r, F);
    onFoot(c, r, v);
  }
  T hn[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) hn[i] = in.hn(i);
  closeSweep(md, gravity, acc, hn, fsum, tsum, sz, cz, sy, cy, f, bm);
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
