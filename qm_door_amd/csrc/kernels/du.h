// Lane-tangent dual numbers for CDNA4 wavefronts.
//
// A wave evaluates a function ONCE in its primal part (`v`, identical in all 64 lanes, so branches on it
// stay wave-uniform) while every lane carries the directional derivative `d` along its own seed direction:
// lane l < 30 differentiates w.r.t. x[l], lane 30 <= l < 60 w.r.t. u[l-30].  After one pass lane l holds
// column l of the Jacobian [df/dx | df/du] -- the 60 Jacobian columns the reference obtains from CppAD
// (qm_interface/src/dynamics/QMDynamicsAD.cpp:30-33) come out of a single SIMD sweep, 60/64 lanes busy.
//
// The same templated device code runs with T = double (one shooting node per lane, value only) in the
// line-search kernel.
#pragma once
#include "gpu_rt.h"

namespace qmk {

struct Du {
  real v, d;
  __device__ __forceinline__ Du() : v(0.0_r), d(0.0_r) {}
  __device__ __forceinline__ Du(real a) : v(a), d(0.0_r) {}  // NOLINT implicit
  __device__ __forceinline__ Du(real a, real b) : v(a), d(b) {}
};
__device__ __forceinline__ Du operator+(Du a, Du b) { return Du(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ Du operator-(Du a, Du b) { return Du(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ Du operator-(Du a) { return Du(-a.v, -a.d); }
__device__ __forceinline__ Du operator*(Du a, Du b) { return Du(a.v * b.v, fma(a.v, b.d, a.d * b.v)); }
__device__ __forceinline__ Du operator*(real a, Du b) { return Du(a * b.v, a * b.d); }
__device__ __forceinline__ Du operator*(Du b, real a) { return Du(a * b.v, a * b.d); }
__device__ __forceinline__ Du operator+(Du a, real b) { return Du(a.v + b, a.d); }
__device__ __forceinline__ Du operator+(real b, Du a) { return Du(a.v + b, a.d); }
__device__ __forceinline__ Du operator-(Du a, real b) { return Du(a.v - b, a.d); }
__device__ __forceinline__ Du operator-(real b, Du a) { return Du(b - a.v, -a.d); }
__device__ __forceinline__ Du operator/(Du a, Du b) { const real q = a.v / b.v; return Du(q, (a.d - q * b.d) / b.v); }
__device__ __forceinline__ Du operator/(Du a, real b) { const real r = 1.0_r / b; return Du(a.v * r, a.d * r); }
__device__ __forceinline__ Du operator/(real a, Du b) { const real q = a / b.v; return Du(q, -q * b.d / b.v); }
__device__ __forceinline__ Du& operator+=(Du& a, Du b) { a.v += b.v; a.d += b.d; return a; }
__device__ __forceinline__ Du& operator-=(Du& a, Du b) { a.v -= b.v; a.d -= b.d; return a; }

// Velocity-type dual number for the structured Jacobian of ad_node_kernel.  The flow map is LINEAR in the joint velocities, the
// momenta and the contact forces: only the 21 configuration directions (Euler angles, joint angles) need a tangent through the tree
// sweep.  A quantity that depends on the configuration only is a Du (value, d = tangent along the lane's configuration direction); a
// quantity that also depends linearly on a velocity-like argument (joint rates, momentum) is a Du3 whose third slot e carries the
// tangent along the lane's VELOCITY direction -- that tangent only ever multiplies configuration VALUES, never configuration
// tangents, so one lane differentiates along two directions at once for 5/3 of the arithmetic of one.  Du3 * Du3 is deliberately
// not defined: a product of two velocity-type quantities would mean the map is not linear in them.
struct Du3 {
  real v, d, e;
  __device__ __forceinline__ Du3() : v(0.0_r), d(0.0_r), e(0.0_r) {}
  __device__ __forceinline__ Du3(real a) : v(a), d(0.0_r), e(0.0_r) {}  // NOLINT implicit
  __device__ __forceinline__ Du3(Du a) : v(a.v), d(a.d), e(0.0_r) {}    // NOLINT implicit
  __device__ __forceinline__ Du3(real a, real b, real c) : v(a), d(b), e(c) {}
};
__device__ __forceinline__ Du3 operator+(Du3 a, Du3 b) { return Du3(a.v + b.v, a.d + b.d, a.e + b.e); }
__device__ __forceinline__ Du3 operator-(Du3 a, Du3 b) { return Du3(a.v - b.v, a.d - b.d, a.e - b.e); }
__device__ __forceinline__ Du3 operator-(Du3 a) { return Du3(-a.v, -a.d, -a.e); }
__device__ __forceinline__ Du3 operator+(Du3 a, Du b) { return Du3(a.v + b.v, a.d + b.d, a.e); }
__device__ __forceinline__ Du3 operator+(Du b, Du3 a) { return Du3(a.v + b.v, a.d + b.d, a.e); }
__device__ __forceinline__ Du3 operator-(Du3 a, Du b) { return Du3(a.v - b.v, a.d - b.d, a.e); }
__device__ __forceinline__ Du3 operator-(Du b, Du3 a) { return Du3(b.v - a.v, b.d - a.d, -a.e); }
__device__ __forceinline__ Du3 operator+(Du3 a, real b) { return Du3(a.v + b, a.d, a.e); }
__device__ __forceinline__ Du3 operator+(real b, Du3 a) { return Du3(a.v + b, a.d, a.e); }
__device__ __forceinline__ Du3 operator-(Du3 a, real b) { return Du3(a.v - b, a.d, a.e); }
__device__ __forceinline__ Du3 operator-(real b, Du3 a) { return Du3(b - a.v, -a.d, -a.e); }
__device__ __forceinline__ Du3 operator*(Du a, Du3 b) { return Du3(a.v * b.v, fma(a.v, b.d, a.d * b.v), a.v * b.e); }
__device__ __forceinline__ Du3 operator*(Du3 b, Du a) { return Du3(a.v * b.v, fma(a.v, b.d, a.d * b.v), a.v * b.e); }
__device__ __forceinline__ Du3 operator*(real a, Du3 b) { return Du3(a * b.v, a * b.d, a * b.e); }
__device__ __forceinline__ Du3 operator*(Du3 b, real a) { return Du3(a * b.v, a * b.d, a * b.e); }
__device__ __forceinline__ Du3 operator/(Du3 a, Du b) { const real r = 1.0_r / b.v, q = a.v * r; return Du3(q, (a.d - q * b.d) * r, a.e * r); }
__device__ __forceinline__ Du3 operator/(Du3 a, real b) { const real r = 1.0_r / b; return Du3(a.v * r, a.d * r, a.e * r); }
__device__ __forceinline__ Du3& operator+=(Du3& a, Du3 b) { a.v += b.v; a.d += b.d; a.e += b.e; return a; }
__device__ __forceinline__ real val(Du3 a) { return a.v; }

// sin and cos of a joint / Euler angle (|a| of a few radians): two-term Cody-Waite reduction by pi/2 and the classic minimax
// kernels on [-pi/4, pi/4] (coefficients of the public-domain fdlibm k_sin.c / k_cos.c).  ~35 fp64 instructions instead of the
// ~170 of the general-range library sincos, error < 1 ulp for |a| < 1e5 -- 42 of them sit on every node's tree sweep.
__device__ __forceinline__ void qmSinCos(double a, double& s, double& c) {
  const double kf = rint(a * 6.36619772367581382433e-01);   // a * 2/pi
  const double r = fma(-kf, 6.07710050650619224932e-11, fma(-kf, 1.57079632673412561417e+00, a));
  const double z = r * r;
  const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  const double sr = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
  const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const double cr = 1.0 - (0.5 * z - z * pc);
  const int q = int(kf) & 3;
  const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}
// fp32 build: the library's single-precision pair (the fp64 coefficients above do not apply)
__device__ __forceinline__ void qmSinCos(float a, float& s, float& c) { s = sinf(a); c = cosf(a); }
__device__ __forceinline__ void sincosT(real a, real& s, real& c) { qmSinCos(a, s, c); }
__device__ __forceinline__ void sincosT(Du a, Du& s, Du& c) { real sv, cv; qmSinCos(a.v, sv, cv); s = Du(sv, cv * a.d); c = Du(cv, -sv * a.d); }
__device__ __forceinline__ real sqrtT(real a) { return sqrt(a); }
__device__ __forceinline__ Du sqrtT(Du a) { const real r = sqrt(a.v); return Du(r, 0.5_r * a.d / r); }
__device__ __forceinline__ real val(real a) { return a; }
__device__ __forceinline__ real val(Du a) { return a.v; }
// fused multiply-add helpers: r = a*b + c
__device__ __forceinline__ real fmaT(real a, real b, real c) { return fma(a, b, c); }
__device__ __forceinline__ Du fmaT(Du a, Du b, Du c) { return Du(fma(a.v, b.v, c.v), fma(a.v, b.d, fma(a.d, b.v, c.d))); }
__device__ __forceinline__ Du fmaT(real a, Du b, Du c) { return Du(fma(a, b.v, c.v), fma(a, b.d, c.d)); }

// result types of mixed arithmetic (double / Du / Du3)
template <class A, class B> using ProdT = decltype(A() * B());
template <class A, class B> using SumT = decltype(A() + B());

template <class T> struct Vec3 {
  T x, y, z;
  __device__ __forceinline__ Vec3() : x(0.0_r), y(0.0_r), z(0.0_r) {}
  __device__ __forceinline__ Vec3(T a, T b, T c) : x(a), y(b), z(c) {}
  template <class U> __device__ __forceinline__ Vec3(const Vec3<U>& o) : x(o.x), y(o.y), z(o.z) {}   // NOLINT implicit promotion (double -> Du -> Du3)
};
template <class A, class B> __device__ __forceinline__ Vec3<SumT<A, B>> operator+(Vec3<A> a, Vec3<B> b) { return Vec3<SumT<A, B>>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class A, class B> __device__ __forceinline__ Vec3<SumT<A, B>> operator-(Vec3<A> a, Vec3<B> b) { return Vec3<SumT<A, B>>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class A, class B> __device__ __forceinline__ Vec3<ProdT<A, B>> operator*(A s, Vec3<B> a) { return Vec3<ProdT<A, B>>(s * a.x, s * a.y, s * a.z); }
template <class T> __device__ __forceinline__ Vec3<T> scale(real s, Vec3<T> a) { return Vec3<T>(s * a.x, s * a.y, s * a.z); }
template <class A, class B> __device__ __forceinline__ Vec3<ProdT<A, B>> cross(Vec3<A> a, Vec3<B> b) {
  return Vec3<ProdT<A, B>>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <class A, class B> __device__ __forceinline__ ProdT<A, B> dot(Vec3<A> a, Vec3<B> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// 3x3 matrix stored by columns (a joint rotation about a body axis only mixes two columns)
template <class T> struct Mat3 {
  Vec3<T> c0, c1, c2;
};
template <class T> __device__ __forceinline__ Vec3<T> mul(const Mat3<T>& R, real x, real y, real z) { return scale(x, R.c0) + scale(y, R.c1) + scale(z, R.c2); }
template <class A, class B> __device__ __forceinline__ Vec3<ProdT<A, B>> mul(const Mat3<A>& R, Vec3<B> v) { return v.x * R.c0 + v.y * R.c1 + v.z * R.c2; }
template <class A, class B> __device__ __forceinline__ Vec3<ProdT<A, B>> mulT(const Mat3<A>& R, Vec3<B> v) { return Vec3<ProdT<A, B>>(dot(R.c0, v), dot(R.c1, v), dot(R.c2, v)); }

// symmetric 3x3: xx xy xz yy yz zz
template <class T> struct Sym3 {
  T xx, xy, xz, yy, yz, zz;
  __device__ __forceinline__ Sym3() : xx(0.0_r), xy(0.0_r), xz(0.0_r), yy(0.0_r), yz(0.0_r), zz(0.0_r) {}
};
template <class A, class B> __device__ __forceinline__ Vec3<ProdT<A, B>> mul(const Sym3<A>& S, Vec3<B> v) {
  return Vec3<ProdT<A, B>>(S.xx * v.x + S.xy * v.y + S.xz * v.z, S.xy * v.x + S.yy * v.y + S.yz * v.z, S.xz * v.x + S.yz * v.y + S.zz * v.z);
}

}  // namespace qmk
