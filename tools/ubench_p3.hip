// Stand-alone timing of the dense factorisation step of riccati_kernel (P3): one wavefront, [H | G g] (18 + 31 columns) from LDS,
// variants of the multiplier broadcast and of the reciprocal square root.  Each variant is checked against a host Cholesky.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_p3 tools/ubench_p3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <utility>
constexpr int MT = 18, NG = 31, LDS_Y = 80, LDS_W = 48, REPS = 50;
__device__ __forceinline__ double rl(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double fastRsqrt(double x) {     // v_rsq_f64 + one third-order correction (what ocml does, minus the special cases)
  const double y = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(-x * y, y, 1.0);
  return __builtin_fma(y * e, __builtin_fma(e, 0.375, 0.5), y);
}
__device__ __forceinline__ long long tick(double& a) { long long t; asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(a)); return t; }

template <int R, bool FIRST> __device__ __forceinline__ void fmacRowBcast(double& acc, double bc, double m) {   // acc += bc[lane r of my row of 16] * m
  if (FIRST) asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bc), "v"(m), "n"(R));
  else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bc), "v"(m), "n"(R));
}
__device__ __forceinline__ double replicateRow0(double v, int addr) {   // value of lane (lane & 15), in every lane
  return __hiloint2double(__builtin_amdgcn_ds_bpermute(addr, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)));
}
template <int J, int R, int NT> struct RowUpd {
  static __device__ __forceinline__ void run(double* col, double bc, double ncj) {
    if constexpr (R < NT && R < 16) { fmacRowBcast<R, R == J + 2>(col[R], bc, ncj); RowUpd<J, R + 1, NT>::run(col, bc, ncj); }
  }
};
// V = 0: readlane multipliers + ocml rsqrt (the kernel today); 1: readlane + fast rsqrt; 2: LDS broadcast multipliers + fast rsqrt
// NT = compile-time number of pivots (rows >= NT are identity)
template <int V, int NT> __global__ void p3(const double* Tin, double* Wout, double* LTout, long long* ticks) {
  __shared__ double T[32 * LDS_Y];
  __shared__ double W[20 * LDS_W];
  __shared__ double LT[20 * 20 + 20];
  __shared__ __attribute__((aligned(16))) double rowb[2][32];
  const int lane = threadIdx.x;
  for (int e = lane; e < 32 * LDS_Y; e += 64) T[e] = Tin[e];
  __syncthreads();
  const bool isH = lane < MT, isG = lane >= MT && lane < MT + NG;
  const int c = isH ? lane : (isG ? lane - MT : 0);
  long long total = 0;
  double keep = 0;
  for (int rep = 0; rep < REPS; ++rep) {
    double col[MT], hv[MT];
    double dummy = keep;
    const long long t0 = tick(dummy);
#pragma unroll
    for (int r = 0; r < NT; ++r) hv[r] = T[r * LDS_Y + (isH ? 32 + c : c)];
#pragma unroll
    for (int r = 0; r < NT; ++r) asm volatile("" : "+v"(hv[r]));
#pragma unroll
    for (int r = 0; r < NT; ++r) col[r] = (isH ? c < NT : isG) ? hv[r] : ((isH && r == c) ? 1.0 : 0.0);
    double inv, invd[MT];
    {
      const double piv = rl(col[0], 0);
      inv = V == 0 ? rsqrt(piv > 0 ? piv : 1.0) : fastRsqrt(piv > 0 ? piv : 1.0);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      col[j] *= inv; invd[j] = inv;
      if constexpr (V == 2) {
        if (j + 1 < NT) {
          if (isH) rowb[j & 1][lane] = col[j];             // the scaled pivot row: multipliers of this step, by lane
          col[j + 1] -= rl(col[j], j + 1) * col[j];
          const double piv = rl(col[j + 1], j + 1);
          inv = fastRsqrt(piv > 0 ? piv : 1.0);
          // rows j + 2 .. NT - 1: multipliers from LDS, two per ds_read_b128 (same address in every lane)
          const double2* rb = reinterpret_cast<const double2*>(&rowb[j & 1][0]);
#pragma unroll
          for (int r = (j + 2) & ~1; r < NT; r += 2) {
            const double2 m = rb[r >> 1];
            if (r >= j + 2) col[r] -= m.x * col[j];
            if (r + 1 < NT) col[r + 1] -= m.y * col[j];
          }
        }
      } else if constexpr (V == 3) {
        // multipliers of the rows j + 2 .. 15 by DPP row_newbcast from a copy of the pivot row's H lanes replicated into every row of 16
        // lanes: ONE v_fmac_f64_dpp per row update; row j + 1 (on the pivot chain) and rows 16, 17 by v_readlane as before
        const double ncj = -col[j];
        const double bc = replicateRow0(col[j], (lane & 15) * 4);
        if (j + 1 < NT) {
          col[j + 1] -= rl(col[j], j + 1) * col[j];
          const double piv = rl(col[j + 1], j + 1);
          inv = fastRsqrt(piv > 0 ? piv : 1.0);
        }
        [&]<int... JJ>(std::integer_sequence<int, JJ...>) { ((JJ == j ? RowUpd<JJ, JJ + 2, NT>::run(col, bc, ncj) : void()), ...); }(std::make_integer_sequence<int, NT>{});
#pragma unroll
        for (int r = (j + 2 > 16 ? j + 2 : 16); r < NT; ++r) col[r] -= rl(col[j], r) * col[j];
      } else {
        if (j + 1 < NT) {
          col[j + 1] -= rl(col[j], j + 1) * col[j];
          const double piv = rl(col[j + 1], j + 1);
          inv = V == 0 ? rsqrt(piv > 0 ? piv : 1.0) : fastRsqrt(piv > 0 ? piv : 1.0);
        }
#pragma unroll
        for (int r = j + 2; r < NT; ++r) col[r] -= rl(col[j], r) * col[j];
      }
    }
    if (isH) {
#pragma unroll
      for (int r = 0; r < NT; ++r) LT[r * 20 + c] = col[r];
      double mine = 1.0;
#pragma unroll
      for (int r = 0; r < NT; ++r) if (r == c) mine = invd[r];
      LT[400 + c] = mine;
    } else if (isG) {
#pragma unroll
      for (int r = 0; r < NT; ++r) W[r * LDS_W + c] = col[r];
    }
    double d2 = col[0];
    const long long t1 = tick(d2);
    total += t1 - t0; keep += d2 * 1e-300;
    __syncthreads();
  }
  if (lane == 0) ticks[0] = total / REPS;
  __syncthreads();
  for (int e = lane; e < 20 * LDS_W; e += 64) Wout[e] = W[e];
  for (int e = lane; e < 420; e += 64) LTout[e] = LT[e];
  if (keep == 12345.0) Wout[0] = keep;
}


typedef double d4 __attribute__((ext_vector_type(4)));
// Blocked variant: the matrix stays in MFMA accumulator layout (lane (n, h), register r <-> row h + 4 r, column n of a 16 x 16 tile);
// per block of 4 pivots: the 4 x 4 diagonal block is broadcast and factorised/inverted redundantly in every lane, the panel rows become
// Linv * panel by one MFMA per column tile (the accumulator registers of rows jb..jb+3 ARE the B operand), and the trailing update is one
// MFMA per tile with A = -(panel rows at the H columns) (lane-local again).
template <int NR, int NC> __global__ void p3b(const double* Tin, double* Wout, double* LTout, long long* ticks, int nt) {
  __shared__ double T[32 * LDS_Y];
  __shared__ double W[20 * LDS_W];
  __shared__ double LT[20 * 20 + 20];
  const int lane = threadIdx.x, n = lane & 15, h = lane >> 4;
  for (int e = lane; e < 32 * LDS_Y; e += 64) T[e] = Tin[e];
  __syncthreads();
  long long total = 0;
  double keep = 0;
  int status = 0;
  for (int rep = 0; rep < REPS; ++rep) {
    double dummy = keep;
    const long long t0 = tick(dummy);
    d4 t[NR][NC];
#pragma unroll
    for (int R = 0; R < NR; ++R)
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * R + h + 4 * r, col = 16 * c + n;
          double v = T[i * LDS_Y + col];
          asm volatile("" : "+v"(v));
          const bool isHc = c >= 2;
          const int hc = col - 32;
          const bool live = i < nt && (!isHc || hc < nt);
          t[R][c][r] = live ? v : ((isHc && hc == i) ? 1.0 : 0.0);
        }
    constexpr int NB = NR == 1 ? 4 : 5;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int jb = 4 * b, R = jb >> 4, r = (jb & 15) >> 2;
      // ---- diagonal block, upper triangle D[k][k2], k2 >= k: lane (n = (jb + k2) % 16, h = k) of H tile 2 + (jb + k2) / 16
      double D[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int k2 = k; k2 < 4; ++k2) D[k][k2] = rl(t[R][2 + ((jb + k2) >> 4)][r], 16 * k + ((jb + k2) & 15));
      // ---- Cholesky of the block (lower L, row-wise) and its inverse M = L^-1, the same in every lane
      auto piv = [&](double d) { const bool ok = d > 1e-200; if (!ok) status = 1; return fastRsqrt(ok ? d : 1.0); };
      const double i0 = piv(D[0][0]);
      const double l10 = D[0][1] * i0, l20 = D[0][2] * i0, l30 = D[0][3] * i0;
      const double i1 = piv(D[1][1] - l10 * l10);
      const double l21 = (D[1][2] - l20 * l10) * i1, l31 = (D[1][3] - l30 * l10) * i1;
      const double i2 = piv(D[2][2] - l20 * l20 - l21 * l21);
      const double l32 = (D[2][3] - l30 * l20 - l31 * l21) * i2;
      const double i3 = piv(D[3][3] - l30 * l30 - l31 * l31 - l32 * l32);
      const double m10 = -l10 * i0 * i1;
      const double m21 = -l21 * i1 * i2;
      const double m32 = -l32 * i2 * i3;
      const double m20 = -(l20 * i0 + l21 * m10) * i2;
      const double m31 = -(l31 * i1 + l32 * m21) * i3;
      const double m30 = -(l30 * i0 + l31 * m10 + l32 * m20) * i3;
      // ---- A operand of the panel product: lane (m = n, k = h) supplies M[m][k] (m < 4)
      double a1 = 0.0;
      a1 = (n == 0 && h == 0) ? i0 : a1;
      a1 = (n == 1 && h == 0) ? m10 : a1; a1 = (n == 1 && h == 1) ? i1 : a1;
      a1 = (n == 2 && h == 0) ? m20 : a1; a1 = (n == 2 && h == 1) ? m21 : a1; a1 = (n == 2 && h == 2) ? i2 : a1;
      a1 = (n == 3 && h == 0) ? m30 : a1; a1 = (n == 3 && h == 1) ? m31 : a1; a1 = (n == 3 && h == 2) ? m32 : a1; a1 = (n == 3 && h == 3) ? i3 : a1;
      // ---- panel rows <- M * panel rows
      d4 pc[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) { const d4 z = {0, 0, 0, 0}; pc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, t[R][c][r], z, 0, 0, 0); }
#pragma unroll
      for (int c = 0; c < NC; ++c) t[R][c][r] = pc[c][0];
      // ---- trailing update of the rows below the panel
#pragma unroll
      for (int R2 = R; R2 < NR; ++R2) {
        if (16 * R2 + 15 <= jb + 3) continue;
        if (16 * R2 >= 4 * NB) continue;                       // rows beyond the last block stay identity
        const double u = t[R][2 + R2][r];                      // U[k = h][H column 16 R2 + n], lane-local
        const double a2 = (16 * R2 + n > jb + 3) ? -u : 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) t[R2][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, t[R][c][r], t[R2][c], 0, 0, 0);
      }
    }
    // ---- W = rows of the [G g] tiles, L^T = rows of the H tiles
#pragma unroll
    for (int R = 0; R < NR; ++R)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * R + h + 4 * r;
        if (i < 20) {
          W[i * LDS_W + n] = t[R][0][r]; W[i * LDS_W + 16 + n] = t[R][1][r];
#pragma unroll
          for (int c = 2; c < NC; ++c) { const int hc = 16 * (c - 2) + n; if (hc < 20) LT[i * 20 + hc] = t[R][c][r]; }
        }
      }
    double d2 = t[0][0][0];
    const long long t1 = tick(d2);
    total += t1 - t0; keep += d2 * 1e-300;
    __syncthreads();
  }
  if (lane == 0) ticks[0] = total / REPS + (status ? 1000000 : 0);
  __syncthreads();
  for (int e = lane; e < 20 * LDS_W; e += 64) Wout[e] = W[e];
  for (int e = lane; e < 420; e += 64) LTout[e] = LT[e];
  if (keep == 12345.0) Wout[0] = keep;
}

template <int NR, int NC> void runb(const char* name, int NT, const std::vector<double>& Lref, const std::vector<double>& Wref, double* dT, double* dW, double* dL, long long* dt) {
  for (int i = 0; i < 2; ++i) { hipLaunchKernelGGL((p3b<NR, NC>), dim3(1), dim3(64), 0, 0, dT, dW, dL, dt, NT); (void)hipDeviceSynchronize(); }
  std::vector<double> W(20 * LDS_W), L(420); long long t;
  (void)hipMemcpy(W.data(), dW, W.size() * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(L.data(), dL, L.size() * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
  double eL = 0, eW = 0;
  for (int r = 0; r < NT; ++r) { for (int c = r; c < NT; ++c) eL = fmax(eL, fabs(L[r * 20 + c] - Lref[c * MT + r])); for (int c = 0; c < NG; ++c) eW = fmax(eW, fabs(W[r * LDS_W + c] - Wref[r * NG + c])); }
  printf("%-44s NT=%2d  %6lld cycles   err L %.1e  W %.1e\n", name, NT, t, eL, eW);
}

template <int V, int NT> void run(const char* name, const std::vector<double>& T, const std::vector<double>& Lref, const std::vector<double>& Wref, double* dT, double* dW, double* dL, long long* dt) {
  for (int i = 0; i < 2; ++i) { hipLaunchKernelGGL((p3<V, NT>), dim3(1), dim3(64), 0, 0, dT, dW, dL, dt); (void)hipDeviceSynchronize(); }
  std::vector<double> W(20 * LDS_W), L(420); long long t;
  (void)hipMemcpy(W.data(), dW, W.size() * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(L.data(), dL, L.size() * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
  double eL = 0, eW = 0;
  for (int r = 0; r < NT; ++r) { for (int c = r; c < NT; ++c) eL = fmax(eL, fabs(L[r * 20 + c] - Lref[c * MT + r])); for (int c = 0; c < NG; ++c) eW = fmax(eW, fabs(W[r * LDS_W + c] - Wref[r * NG + c])); }
  printf("%-44s NT=%2d  %6lld cycles   err L %.1e  W %.1e\n", name, NT, t, eL, eW);
}

int main() {
  // SPD H (NT x NT leading block used), random G
  std::vector<double> T(32 * LDS_Y, 0.0), H(MT * MT), G(MT * NG);
  unsigned s = 12345; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return double(s >> 8) / (1 << 24) - 0.5; };
  std::vector<double> A(MT * MT); for (auto& a : A) a = rnd();
  for (int i = 0; i < MT; ++i) for (int j = 0; j < MT; ++j) { double v = i == j ? 2.0 : 0.0; for (int k = 0; k < MT; ++k) v += A[i * MT + k] * A[j * MT + k]; H[i * MT + j] = v; }
  for (auto& g : G) g = rnd();
  for (int r = 0; r < MT; ++r) { for (int c = 0; c < MT; ++c) T[r * LDS_Y + 32 + c] = H[r * MT + c]; for (int c = 0; c < NG; ++c) T[r * LDS_Y + c] = G[r * NG + c]; }
  double *dT, *dW, *dL; long long* dt;
  (void)hipMalloc(&dT, T.size() * 8); (void)hipMalloc(&dW, 20 * LDS_W * 8); (void)hipMalloc(&dL, 420 * 8); (void)hipMalloc(&dt, 8);
  (void)hipMemcpy(dT, T.data(), T.size() * 8, hipMemcpyHostToDevice);
  auto ref = [&](int NT, std::vector<double>& L, std::vector<double>& W) {
    L.assign(MT * MT, 0.0); W.assign(MT * NG, 0.0);
    for (int j = 0; j < NT; ++j) {
      double d = H[j * MT + j]; for (int k = 0; k < j; ++k) d -= L[j * MT + k] * L[j * MT + k]; L[j * MT + j] = sqrt(d);
      for (int i = j + 1; i < NT; ++i) { double v = H[i * MT + j]; for (int k = 0; k < j; ++k) v -= L[i * MT + k] * L[j * MT + k]; L[i * MT + j] = v / L[j * MT + j]; }
    }
    for (int c = 0; c < NG; ++c) for (int r = 0; r < NT; ++r) { double v = G[r * NG + c]; for (int k = 0; k < r; ++k) v -= L[r * MT + k] * W[k * NG + c]; W[r * NG + c] = v / L[r * MT + r]; }
  };
  std::vector<double> L, W;
  ref(18, L, W);
  run<0, 18>("readlane multipliers, ocml rsqrt", T, L, W, dT, dW, dL, dt);
  run<1, 18>("readlane multipliers, rsq + one correction", T, L, W, dT, dW, dL, dt);
  run<3, 18>("DPP row_newbcast multipliers", T, L, W, dT, dW, dL, dt);
  ref(16, L, W);
  run<0, 16>("readlane multipliers, ocml rsqrt", T, L, W, dT, dW, dL, dt);
  run<1, 16>("readlane multipliers, rsq + one correction", T, L, W, dT, dW, dL, dt);
  run<3, 16>("DPP row_newbcast multipliers", T, L, W, dT, dW, dL, dt);
  ref(14, L, W);
  runb<1, 3>("blocked MFMA, 1 row tile x 3 column tiles", 14, L, W, dT, dW, dL, dt);
  ref(16, L, W);
  runb<1, 3>("blocked MFMA, 1 row tile x 3 column tiles", 16, L, W, dT, dW, dL, dt);
  runb<2, 4>("blocked MFMA, 2 row tiles x 4 column tiles", 16, L, W, dT, dW, dL, dt);
  ref(17, L, W);
  runb<2, 4>("blocked MFMA, 2 row tiles x 4 column tiles", 17, L, W, dT, dW, dL, dt);
  ref(18, L, W);
  runb<2, 4>("blocked MFMA, 2 row tiles x 4 column tiles", 18, L, W, dT, dW, dL, dt);
  return 0;
}
