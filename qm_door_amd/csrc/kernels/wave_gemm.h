// wave_gemm.h -- C (M x N, row major in LDS) = op(A) B for one wavefront on v_mfma_f64_16x16x4_f64, arbitrary (runtime) sizes.
// Operands are read straight from LDS with predicated (zero-padded) loads, three k steps in flight per tile.  Used by the WBC
// kernel for the reduced data of a level (A Z, D Z, (A Z)^T (A Z), Z N); the fixed-size products of the MPC kernels have their
// own hand-laid tile loops.
#pragma once
#include "gpu_rt.h"

namespace qmk {

// TA = false: op(A)[i][k] = A[i * lda + k];  TA = true: op(A)[i][k] = A[k * lda + i]
template <bool TA, class Epilogue>
__device__ inline void waveGemm(const double* A, int lda, const double* B, int ldb, int M, int N, int K, int lane, double* scratch, Epilogue&& store) {
  const int l16 = lane & 15, h = lane >> 4;
  const int tilesM = (M + 15) >> 4, tilesN = (N + 15) >> 4, kSteps = (K + 3) >> 2;
#pragma unroll 1
  for (int tm = 0; tm < tilesM; ++tm) {
#pragma unroll 1
    for (int tn = 0; tn < tilesN; ++tn) {
      const int ia = tm * 16 + l16, jb = tn * 16 + l16;
      const bool iok = ia < M, jok = jb < N;
      const int iac = iok ? ia : 0, jbc = jok ? jb : 0;
      QmAcc c;
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = 0.0;
#pragma unroll 1
      for (int k0 = 0; k0 < kSteps; k0 += 3) {
        double av[3], bv[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int kk = 4 * (k0 + u) + h;
          const bool kok = kk < K;
          const int kc = kok ? kk : 0;
          const double ar = TA ? A[kc * lda + iac] : A[iac * lda + kc];
          const double br = B[kc * ldb + jbc];
          av[u] = (iok && kok) ? ar : 0.0;
          bv[u] = (jok && kok) ? br : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) qmMfma(c, av[u], bv[u], scratch);   // steps beyond kSteps multiply zeros
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = tm * 16 + h + 4 * r;
        if (i < M && jok) store(i, jb, c[r]);
      }
    }
  }
}

// The same product with a plain destination (D[i * ldd + j] = sum + diagAdd on the diagonal), restricted to the 16 x 16 tiles t with
// t % 4 == wave (wave < 0: all tiles): the share of one of the four wavefronts of wbc_kernel's fork-join.
template <bool TA>
__device__ inline void waveGemmTiles(const double* A, int lda, const double* B, int ldb, int M, int N, int K, double* D, int ldd, double diagAdd, int wave, int lane, double* scratch) {
  const int l16 = lane & 15, h = lane >> 4;
  const int tilesM = (M + 15) >> 4, tilesN = (N + 15) >> 4, kSteps = (K + 3) >> 2;
  int t = 0;
#pragma unroll 1
  for (int tm = 0; tm < tilesM; ++tm) {
#pragma unroll 1
    for (int tn = 0; tn < tilesN; ++tn, ++t) {
      if (wave >= 0 && (t & 3) != wave) continue;
      const int ia = tm * 16 + l16, jb = tn * 16 + l16;
      const bool iok = ia < M, jok = jb < N;
      const int iac = iok ? ia : 0, jbc = jok ? jb : 0;
      QmAcc c;
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = 0.0;
#pragma unroll 1
      for (int k0 = 0; k0 < kSteps; k0 += 3) {
        double av[3], bv[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int kk = 4 * (k0 + u) + h;
          const bool kok = kk < K;
          const int kc = kok ? kk : 0;
          const double ar = TA ? A[kc * lda + iac] : A[iac * lda + kc];
          const double br = B[kc * ldb + jbc];
          av[u] = (iok && kok) ? ar : 0.0;
          bv[u] = (jok && kok) ? br : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) qmMfma(c, av[u], bv[u], scratch);   // steps beyond kSteps multiply zeros
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = tm * 16 + h + 4 * r;
        if (i < M && jok) D[i * ldd + jb] = c[r] + (i == jb ? diagAdd : 0.0);
      }
    }
  }
}

}  // namespace qmk
