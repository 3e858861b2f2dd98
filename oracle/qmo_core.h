// qmo_core.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).  Not part of the product; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it.
//
// PARITY UNPINNED: the reference (danisotelo/qm_door) ships no tests or golden vectors and its
// arithmetic lives in un-vendored OCS2 / Pinocchio / HPIPM / qpOASES / CppAD that cannot be built
// in this image.  This oracle is a plain fp64 restatement of the published
// algorithms, anchored on the reference's call sites, and pinned by the invariant suite in tests/.
//
// Small dense linear algebra + forward-mode dual numbers (the oracle's stand-in for CppAD, which the
// reference uses for every derivative: qm_interface/src/dynamics/QMDynamicsAD.cpp:15-33,
// qm_interface/src/QMInterface.cpp:363-379).
#pragma once
#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace qmo {

// ------------------------------------------------------------------------------------------------ phase timers (timing-grade build only)
// Where a cycle's time goes on the CPU: LQ approximation + projection, Riccati, line search, WBC model,
// WBC QPs.  Thread-seconds summed over all threads, read and cleared by qmo_time_split.
enum Phase { PH_LQ = 0, PH_RICCATI, PH_LINESEARCH, PH_WBC_MODEL, PH_WBC_QP, PH_COUNT };
inline double g_phaseSeconds[PH_COUNT] = {0, 0, 0, 0, 0};   // summed over all threads (a handful of additions per cycle)
inline std::mutex g_phaseMutex;
struct PhaseTimer {
#ifdef QMO_FAST
  int ph; std::chrono::steady_clock::time_point t0;
  explicit PhaseTimer(int p) : ph(p), t0(std::chrono::steady_clock::now()) {}
  ~PhaseTimer() {
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::lock_guard<std::mutex> lock(g_phaseMutex);
    g_phaseSeconds[ph] += s;
  }
#else
  explicit PhaseTimer(int) {}
#endif
};

// ------------------------------------------------------------------------------------------------ dual numbers
template <int N>
struct Dual {
  double v;
  double d[N];
  Dual() : v(0.0) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
  Dual(double x) : v(x) { for (int i = 0; i < N; ++i) d[i] = 0.0; }  // NOLINT implicit
};
template <int N> inline Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.v * b.d[i] + a.d[i] * b.v; return r; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v / b.v; for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v; return r; }
template <int N> inline Dual<N> operator+(const Dual<N>& a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N> inline Dual<N> operator+(double b, const Dual<N>& a) { return a + b; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> inline Dual<N> operator-(double b, const Dual<N>& a) { return (-a) + b; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, double b) { Dual<N> r; r.v = a.v * b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r; }
template <int N> inline Dual<N> operator*(double b, const Dual<N>& a) { return a * b; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, double b) { return a * (1.0 / b); }
template <int N> inline Dual<N> operator/(double a, const Dual<N>& b) { return Dual<N>(a) / b; }
template <int N> inline Dual<N>& operator+=(Dual<N>& a, const Dual<N>& b) { a = a + b; return a; }
template <int N> inline Dual<N>& operator-=(Dual<N>& a, const Dual<N>& b) { a = a - b; return a; }
template <int N> inline Dual<N>& operator*=(Dual<N>& a, const Dual<N>& b) { a = a * b; return a; }
template <int N> inline bool operator>(const Dual<N>& a, const Dual<N>& b) { return a.v > b.v; }
template <int N> inline bool operator<(const Dual<N>& a, const Dual<N>& b) { return a.v < b.v; }
template <int N> inline bool operator>(const Dual<N>& a, double b) { return a.v > b; }
template <int N> inline bool operator<(const Dual<N>& a, double b) { return a.v < b; }
template <int N> inline bool operator>=(const Dual<N>& a, double b) { return a.v >= b; }
template <int N> inline bool operator<=(const Dual<N>& a, double b) { return a.v <= b; }
template <int N> inline Dual<N> sin(const Dual<N>& a) { Dual<N> r; r.v = std::sin(a.v); double c = std::cos(a.v); for (int i = 0; i < N; ++i) r.d[i] = c * a.d[i]; return r; }
template <int N> inline Dual<N> cos(const Dual<N>& a) { Dual<N> r; r.v = std::cos(a.v); double s = -std::sin(a.v); for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i]; return r; }
template <int N> inline Dual<N> sqrt(const Dual<N>& a) { Dual<N> r; r.v = std::sqrt(a.v); double s = 0.5 / r.v; for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i]; return r; }
template <int N> inline Dual<N> log(const Dual<N>& a) { Dual<N> r; r.v = std::log(a.v); double s = 1.0 / a.v; for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i]; return r; }
template <int N> inline Dual<N> acos(const Dual<N>& a) { Dual<N> r; r.v = std::acos(a.v); double s = -1.0 / std::sqrt(1.0 - a.v * a.v); for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i]; return r; }
inline double value(double a) { return a; }
template <int N> inline double value(const Dual<N>& a) { return a.v; }
using std::sin; using std::cos; using std::sqrt; using std::log; using std::acos;

// ------------------------------------------------------------------------------------------------ dense matrices
struct Mat {
  int r = 0, c = 0;
  std::vector<double> a;  // row major
  Mat() = default;
  Mat(int r_, int c_) : r(r_), c(c_), a(size_t(r_) * c_, 0.0) {}
  double& operator()(int i, int j) { return a[size_t(i) * c + j]; }
  double operator()(int i, int j) const { return a[size_t(i) * c + j]; }
  static Mat identity(int n) { Mat m(n, n); for (int i = 0; i < n; ++i) m(i, i) = 1.0; return m; }
  static Mat from(const double* p, int r_, int c_) { Mat m(r_, c_); std::memcpy(m.a.data(), p, sizeof(double) * r_ * c_); return m; }
  void to(double* p) const { std::memcpy(p, a.data(), sizeof(double) * r * c); }
};
using Vec = std::vector<double>;

inline Mat T(const Mat& m) { Mat t(m.c, m.r); for (int i = 0; i < m.r; ++i) for (int j = 0; j < m.c; ++j) t(j, i) = m(i, j); return t; }
inline Mat operator*(const Mat& x, const Mat& y) {
  assert(x.c == y.r);
  Mat z(x.r, y.c);
  for (int i = 0; i < x.r; ++i)
    for (int k = 0; k < x.c; ++k) { double s = x(i, k); if (s == 0.0) continue; for (int j = 0; j < y.c; ++j) z(i, j) += s * y(k, j); }
  return z;
}
inline Mat operator+(const Mat& x, const Mat& y) { assert(x.r == y.r && x.c == y.c); Mat z = x; for (size_t i = 0; i < z.a.size(); ++i) z.a[i] += y.a[i]; return z; }
inline Mat operator-(const Mat& x, const Mat& y) { assert(x.r == y.r && x.c == y.c); Mat z = x; for (size_t i = 0; i < z.a.size(); ++i) z.a[i] -= y.a[i]; return z; }
inline Mat operator*(double s, const Mat& x) { Mat z = x; for (auto& v : z.a) v *= s; return z; }
inline Vec operator*(const Mat& x, const Vec& y) { assert(x.c == int(y.size())); Vec z(x.r, 0.0); for (int i = 0; i < x.r; ++i) { double s = 0; for (int j = 0; j < x.c; ++j) s += x(i, j) * y[j]; z[i] = s; } return z; }
inline Vec tmul(const Mat& x, const Vec& y) { assert(x.r == int(y.size())); Vec z(x.c, 0.0); for (int i = 0; i < x.r; ++i) for (int j = 0; j < x.c; ++j) z[j] += x(i, j) * y[i]; return z; }
inline Vec operator+(const Vec& x, const Vec& y) { Vec z = x; for (size_t i = 0; i < z.size(); ++i) z[i] += y[i]; return z; }
inline Vec operator-(const Vec& x, const Vec& y) { Vec z = x; for (size_t i = 0; i < z.size(); ++i) z[i] -= y[i]; return z; }
inline Vec operator*(double s, const Vec& x) { Vec z = x; for (auto& v : z) v *= s; return z; }
inline double dot(const Vec& x, const Vec& y) { double s = 0; for (size_t i = 0; i < x.size(); ++i) s += x[i] * y[i]; return s; }
inline Mat block(const Mat& m, int i0, int j0, int nr, int nc) { Mat b(nr, nc); for (int i = 0; i < nr; ++i) for (int j = 0; j < nc; ++j) b(i, j) = m(i0 + i, j0 + j); return b; }
inline void setBlock(Mat& m, int i0, int j0, const Mat& b) { for (int i = 0; i < b.r; ++i) for (int j = 0; j < b.c; ++j) m(i0 + i, j0 + j) = b(i, j); }
inline Mat vstack(const Mat& x, const Mat& y) {
  if (x.r == 0 && x.c == 0) return y;
  if (y.r == 0 && y.c == 0) return x;
  assert(x.c == y.c);
  Mat z(x.r + y.r, x.c);
  std::copy(x.a.begin(), x.a.end(), z.a.begin());
  std::copy(y.a.begin(), y.a.end(), z.a.begin() + x.a.size());
  return z;
}
inline Vec vcat(const Vec& x, const Vec& y) { Vec z = x; z.insert(z.end(), y.begin(), y.end()); return z; }

// Cholesky (lower). Returns false when a pivot is not positive.
inline bool cholesky(Mat& A) {
  const int n = A.r;
  for (int j = 0; j < n; ++j) {
    double d = A(j, j);
    for (int k = 0; k < j; ++k) d -= A(j, k) * A(j, k);
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A(j, j) = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A(i, j);
      for (int k = 0; k < j; ++k) s -= A(i, k) * A(j, k);
      A(i, j) = s / d;
    }
    for (int i = 0; i < j; ++i) A(i, j) = 0.0;
  }
  return true;
}
inline void cholSolve(const Mat& L, Vec& b) {
  const int n = L.r;
  for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L(i, k) * b[k]; b[i] = s / L(i, i); }
  for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= L(k, i) * b[k]; b[i] = s / L(i, i); }
}
inline Mat cholSolve(const Mat& L, const Mat& B) {
  Mat X = B;
  for (int j = 0; j < B.c; ++j) { Vec col(B.r); for (int i = 0; i < B.r; ++i) col[i] = B(i, j); cholSolve(L, col); for (int i = 0; i < B.r; ++i) X(i, j) = col[i]; }
  return X;
}

// Partial pivoting LU solve of a square system (small sizes only).
inline bool luSolve(Mat A, Vec& b) {
  const int n = A.r;
  for (int k = 0; k < n; ++k) {
    int p = k; double best = std::fabs(A(k, k));
    for (int i = k + 1; i < n; ++i) if (std::fabs(A(i, k)) > best) { best = std::fabs(A(i, k)); p = i; }
    if (best == 0.0) return false;
    if (p != k) { for (int j = 0; j < n; ++j) std::swap(A(k, j), A(p, j)); std::swap(b[k], b[p]); }
    for (int i = k + 1; i < n; ++i) { double f = A(i, k) / A(k, k); if (f == 0.0) continue; for (int j = k; j < n; ++j) A(i, j) -= f * A(k, j); b[i] -= f * b[k]; }
  }
  for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int j = i + 1; j < n; ++j) s -= A(i, j) * b[j]; b[i] = s / A(i, i); }
  return true;
}

// Householder QR of an m x n matrix (m >= n): returns full Q (m x m) and R (upper, in the top n rows of Rout).
// Mirrors Eigen::HouseholderQR as used by OCS2's qrConstraintProjection (upstream ocs2_core LinearAlgebra).
inline void householderQR(const Mat& A, Mat& Q, Mat& Rout) {
  const int m = A.r, n = A.c;
  Rout = A;
  Q = Mat::identity(m);
  for (int k = 0; k < std::min(m - 1, n); ++k) {
    double norm = 0; for (int i = k; i < m; ++i) norm += Rout(i, k) * Rout(i, k);
    norm = std::sqrt(norm);
    if (norm == 0.0) continue;
    const double alpha = Rout(k, k) > 0 ? -norm : norm;
    Vec v(m, 0.0);
    for (int i = k; i < m; ++i) v[i] = Rout(i, k);
    v[k] -= alpha;
    double vn = 0; for (int i = k; i < m; ++i) vn += v[i] * v[i];
    if (vn == 0.0) continue;
    for (int j = 0; j < n; ++j) { double s = 0; for (int i = k; i < m; ++i) s += v[i] * Rout(i, j); s *= 2.0 / vn; for (int i = k; i < m; ++i) Rout(i, j) -= s * v[i]; }
    for (int j = 0; j < m; ++j) { double s = 0; for (int i = k; i < m; ++i) s += Q(j, i) * v[i]; s *= 2.0 / vn; for (int i = k; i < m; ++i) Q(j, i) -= s * v[i]; }
  }
}

// Null-space basis of A (r x c) by full-pivot LU, the construction of Eigen's FullPivLU::kernel()
// (reference call site: qm_wbc/src/HoQp.cpp:129).  Returns c x (c - rank); freeOut: the column of A each kernel vector carries its 1 on.
// Pivot search as Eigen 3.3's (upstream: FullPivLU::computeInPlace takes bottomRightCorner(...).cwiseAbs().maxCoeff(&row, &col), a scalar visitor that walks the
// column-major corner column by column and keeps the FIRST strict maximum): ties go to the smallest column position, then the smallest row position.  The level
// tasks carry unit rows, exact ties are the rule; the basis, and with it the coordinates the minimum-norm completion is taken in, depends on this order.
// (no fused multiply-add in the elimination, in EITHER build of the oracle: the pivot search compares the updated entries for equality of magnitude --
// the level tasks carry unit rows, exact ties are the rule -- and the kernels' wbc_kernel.h takes the same decisions with the same roundings)
#if defined(__clang__)
#error "qmo_core.h: kernelFullPivLU relies on GCC's optimize(\"fp-contract=off\") attribute; build the oracle with g++ (oracle/Makefile) or add -ffp-contract=off for this compiler"
#endif
__attribute__((optimize("fp-contract=off"), noinline)) inline Mat kernelFullPivLU(const Mat& Ain, int* rankOut = nullptr, std::vector<int>* freeOut = nullptr, std::vector<int>* pivSeqOut = nullptr /* (row, column) POSITIONS of every pivot at the time it was chosen */) {
  Mat A = Ain;
  const int rows = A.r, cols = A.c, size = std::min(rows, cols);
  std::vector<int> colPerm(cols);
  for (int j = 0; j < cols; ++j) colPerm[j] = j;
  double maxPivot = 0.0;
  std::vector<double> pivots;
  int nonzero = 0;
  for (int k = 0; k < size; ++k) {
    int pr = k, pc = k; double best = 0.0;
    for (int j = k; j < cols; ++j) for (int i = k; i < rows; ++i) if (std::fabs(A(i, j)) > best) { best = std::fabs(A(i, j)); pr = i; pc = j; }
    if (best == 0.0) break;
    if (pivSeqOut) { pivSeqOut->push_back(pr); pivSeqOut->push_back(pc); }
    maxPivot = std::max(maxPivot, best);
    if (pr != k) for (int j = 0; j < cols; ++j) std::swap(A(k, j), A(pr, j));
    if (pc != k) { for (int i = 0; i < rows; ++i) std::swap(A(i, k), A(i, pc)); std::swap(colPerm[k], colPerm[pc]); }
    for (int i = k + 1; i < rows; ++i) { double f = A(i, k) / A(k, k); A(i, k) = f; for (int j = k + 1; j < cols; ++j) A(i, j) -= f * A(k, j); }
    ++nonzero;
  }
  // Eigen's default threshold: eps * diagonal size
  const double thresh = maxPivot * 2.220446049250313e-16 * double(size);
  int rank = 0;
  for (int k = 0; k < nonzero; ++k) if (std::fabs(A(k, k)) > thresh) ++rank;
  if (rankOut) *rankOut = rank;
  const int dimker = cols - rank;
  Mat ker(cols, std::max(dimker, 0));
  if (freeOut) freeOut->clear();
  if (dimker <= 0) return ker;
  // collect pivot columns (in the permuted ordering) that pass the threshold
  std::vector<int> piv;
  for (int k = 0; k < nonzero; ++k) if (std::fabs(A(k, k)) > thresh) piv.push_back(k);
  // permuted U: m = U(piv rows, [piv cols | free cols]) ; solve U11 * X = -U12
  std::vector<int> freeCols;
  { std::vector<char> isPiv(cols, 0); for (int k : piv) isPiv[k] = 1; for (int j = 0; j < cols; ++j) if (!isPiv[j]) freeCols.push_back(j); }
  Mat U11(rank, rank), U12(rank, dimker);
  for (int i = 0; i < rank; ++i) {
    for (int j = 0; j < rank; ++j) U11(i, j) = (piv[j] >= piv[i]) ? A(piv[i], piv[j]) : 0.0;
    for (int j = 0; j < dimker; ++j) U12(i, j) = (freeCols[j] >= piv[i]) ? A(piv[i], freeCols[j]) : 0.0;
  }
  Mat X(rank, dimker);
  for (int j = 0; j < dimker; ++j)
    for (int i = rank - 1; i >= 0; --i) { double s = -U12(i, j); for (int k = i + 1; k < rank; ++k) s -= U11(i, k) * X(k, j); X(i, j) = s / U11(i, i); }
  for (int j = 0; j < dimker; ++j) {
    for (int i = 0; i < rank; ++i) ker(colPerm[piv[i]], j) = X(i, j);
    ker(colPerm[freeCols[j]], j) = 1.0;
    if (freeOut) freeOut->push_back(colPerm[freeCols[j]]);
  }
  return ker;
}

}  // namespace qmo
