// Micro-probe for v_mfma_f64_16x16x4_f64 on gfx950: operand/result lane layout, dependent-issue latency, sustained rate.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma.hip -o gpurun_out/probe_mfma && gpurun_out/probe_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + l / 16];   // A[i][k], i = l%16, k = l/16   (assumed)
  const double b = B[(l / 16) * 16 + l % 16];  // B[k][j], k = l/16, j = l%16   (assumed)
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}

template <int CHAINS>
__global__ void rate_kernel(double* out, int iters, long long* cycles) {
  d4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = d4{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cycles = t1 - t0;
}

// one lone wave: cost of a wave-uniform (broadcast) LDS read feeding an fp64 FMA, b64 vs b128
template <int WIDTH>
__global__ void lds_bcast_kernel(double* out, int iters, long long* cycles) {
  __shared__ double buf[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) buf[i] = 1.0 + i * 1e-6;
  __syncthreads();
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const double x = 1.0 + threadIdx.x * 1e-3;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    const double* row = buf + (it & 31) * 64;
#pragma unroll
    for (int j = 0; j < 32; j += WIDTH) {
      if (WIDTH == 2) {
        const double2 v = *reinterpret_cast<const double2*>(row + j);
        acc[j & 7] += v.x * x; acc[(j + 1) & 7] += v.y * x;
      } else {
        acc[j & 7] += row[j] * x;
      }
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int c = 0; c < 8; ++c) s += acc[c];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) *cycles = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  std::vector<double> A(64), B(64), D(256);
  for (int i = 0; i < 64; ++i) { A[i] = 1 + (rand() % 1000) * 1e-3; B[i] = 1 + (rand() % 1000) * 1e-3; }
  double *dA, *dB, *dD; long long* dC;
  CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 1 << 24)); CK(hipMalloc(&dC, 8));
  CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
  layout_kernel<<<1, 64>>>(dA, dB, dD);
  CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
  double P[16][16];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { P[i][j] = 0; for (int k = 0; k < 4; ++k) P[i][j] += A[i * 4 + k] * B[k * 16 + j]; }
  int okA = 1, okB = 1, okAny = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const double v = D[l * 4 + r];
    int fi = -1, fj = -1;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (fabs(P[i][j] - v) < 1e-12) { fi = i; fj = j; }
    if (fi < 0) okAny = 0;
    if (!(fi == 4 * (l / 16) + r && fj == l % 16)) okA = 0;
    if (!(fi == (l / 16) + 4 * r && fj == l % 16)) okB = 0;
    if (l % 16 == 1 && r < 4) printf("lane %2d reg %d -> D[%d][%d]\n", l, r, fi, fj);
  }
  printf("LAYOUT operands A[i=l%%16][k=l/16], B[k=l/16][j=l%%16]: all results found=%d ; D[4*(l/16)+r][l%%16]=%d ; D[(l/16)+4r][l%%16]=%d\n", okAny, okA, okB);

  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  long long cyc; float ms;
  const int iters = 20000;
  rate_kernel<1><<<1, 64>>>(dD, iters, dC); CK(hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost));
  printf("DEPENDENT chain: %.1f clock64 ticks per mfma_f64_16x16x4\n", double(cyc) / iters);
  rate_kernel<4><<<1, 64>>>(dD, iters, dC); CK(hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost));
  printf("4 independent chains, one wave: %.1f ticks per mfma\n", double(cyc) / iters / 4);
  for (int wpb : {256, 512, 1024}) {
    const int blocks = 256 * 8;
    rate_kernel<4><<<blocks, wpb>>>(dD, 100, dC);
    CK(hipEventRecord(e0)); rate_kernel<4><<<blocks, wpb>>>(dD, iters / 10, dC); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = double(blocks) * (wpb / 64) * (iters / 10) * 4 * 2.0 * 16 * 16 * 4;
    printf("SUSTAINED %d blocks x %d threads: %.2f TFLOP/s fp64 (%.3f ms)\n", blocks, wpb, flops / (ms * 1e-3) / 1e12, ms);
  }
  {
    // calibrate clock64 ticks against wall time
    CK(hipEventRecord(e0)); rate_kernel<1><<<1, 64>>>(dD, 200000, dC); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost));
    printf("CLOCK64 %.1f MHz (ticks %lld in %.3f ms)\n", cyc / (ms * 1e3), cyc, ms);
  }
  lds_bcast_kernel<1><<<1, 64>>>(dD, 2000, dC); CK(hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost));
  printf("LDS broadcast b64 + fma, lone wave: %.1f ticks per (read+fma)\n", double(cyc) / 2000 / 32);
  lds_bcast_kernel<2><<<1, 64>>>(dD, 2000, dC); CK(hipMemcpy(&cyc, dC, 8, hipMemcpyDeviceToHost));
  printf("LDS broadcast b128 + 2 fma, lone wave: %.1f ticks per fma\n", double(cyc) / 2000 / 32);
  return 0;
}
