// Arithmetic type of the MPC kernels.  The kernel sources are written once in terms of `real`; the library holds two builds of them:
//   qmgpu_api.hip     real = double  (namespace qmk)    -- the reference's own arithmetic (ocs2::scalar_t = double), every kernel
//   qmgpu_mpc32.hip   real = float   (namespace qmk32)  -- the MPC kernels only (BASELINE.json configs[4]: fp32-vs-fp64 sweep)
// selected per handle with qmgpu_create_ex(..., dtype).  Floating literals carry the suffix _r (a `real` constant), so that no fp64
// arithmetic sneaks into the fp32 build through a promoted literal.
#pragma once

#ifndef QM_REAL
#define QM_REAL double
#define QM_REAL_IS_DOUBLE 1
#endif

namespace qmk {
using real = QM_REAL;
constexpr real operator""_r(long double v) { return real(v); }
constexpr real operator""_r(unsigned long long v) { return real(v); }
// machine epsilon of the arithmetic type (upstream tests against std::numeric_limits<scalar_t>::epsilon())
constexpr real REAL_PIVOT_MIN = sizeof(real) == 8 ? real(1e-200) : real(1e-30);   // a Cholesky pivot below this is a failed factorisation
constexpr real REAL_EPS = sizeof(real) == 8 ? real(2.220446049250313e-16) : real(1.1920929e-07);
}  // namespace qmk
