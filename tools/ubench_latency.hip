// Instruction timing probes for the latency-chain kernels (riccati P3, WBC interior point): s_memtime around unrolled chains, one wavefront.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_latency tools/ubench_latency.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define REP 256
__device__ __forceinline__ long long tick(double& a) { long long t; asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(a)); return t; }
__device__ __forceinline__ long long tick8(double* z) { long long t; asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7])); return t; }
__global__ void probes(double* out, long long* ticks, double seed) {
  __shared__ double lds[256];
  const int lane = threadIdx.x;
  lds[lane] = seed + lane; lds[lane + 64] = seed * 2 + lane;
  __syncthreads();
  double x = seed + lane * 1e-9, y = seed * 0.5, acc = 0;
  long long t0, t1;
  // 0: dependent v_fma_f64
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) x = __builtin_fma(x, y, 1e-3);
  asm volatile("" : "+v"(x));
  t1 = tick(x); if (lane == 0) ticks[0] = t1 - t0; acc += x;
  // 1: 8 independent v_fma_f64 chains
  double z[8]; for (int k = 0; k < 8; ++k) z[k] = seed + k + lane;
  t0 = tick8(z);
#pragma unroll
  for (int i = 0; i < REP / 8; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = __builtin_fma(z[k], y, 1e-3);
  for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(z[k]));
  t1 = tick8(z); if (lane == 0) ticks[1] = t1 - t0; for (int k = 0; k < 8; ++k) acc += z[k];
  // 2: dependent chain  readlane(lo,hi) -> fma with the SGPR pair -> readlane ...
  x = seed + lane;
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), 3), hi = __builtin_amdgcn_readlane(__double2hiint(x), 3);
    x = __builtin_fma(__hiloint2double(hi, lo), 1e-3, x);
  }
  asm volatile("" : "+v"(x));
  t1 = tick(x); if (lane == 0) ticks[2] = t1 - t0; acc += x;
  // 3: independent row updates as in P3: 8 rows, multiplier read from a fixed register by readlane (not on a chain)
  double piv = seed + lane;
  t0 = tick8(z);
#pragma unroll
  for (int i = 0; i < REP / 8; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int lo = __builtin_amdgcn_readlane(__double2loint(piv), k + 1), hi = __builtin_amdgcn_readlane(__double2hiint(piv), k + 1);
      z[k] = __builtin_fma(-__hiloint2double(hi, lo), piv, z[k]);
    }
  for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(z[k]));
  t1 = tick8(z); if (lane == 0) ticks[3] = t1 - t0; for (int k = 0; k < 8; ++k) acc += z[k];
  // 4: dependent v_rsq_f64
  x = seed + 2.0 + lane;
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) x = __builtin_amdgcn_rsq(x) + 2.0;
  asm volatile("" : "+v"(x));
  t1 = tick(x); if (lane == 0) ticks[4] = t1 - t0; acc += x;
  // 5: dependent ocml rsqrt
  x = seed + 2.0 + lane;
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) x = rsqrt(x) + 2.0;
  asm volatile("" : "+v"(x));
  t1 = tick(x); if (lane == 0) ticks[5] = t1 - t0; acc += x;
  // 6: dependent fp64 MFMA on one accumulator
  d4 c = {0, 0, 0, 0};
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, c, 0, 0, 0);
  asm volatile("" : "+v"(c)); x += c[0];
  t1 = tick(x); if (lane == 0) ticks[6] = t1 - t0; acc += c[0] + c[1] + c[2] + c[3];
  // 7: fp64 MFMA on 4 independent accumulators
  d4 cc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < REP / 4; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) cc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, cc[k], 0, 0, 0);
  for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(cc[k]));
  x += cc[0][0] + cc[1][0] + cc[2][0] + cc[3][0]; t1 = tick(x); if (lane == 0) ticks[7] = t1 - t0; for (int k = 0; k < 4; ++k) acc += cc[k][0] + cc[k][3];
  // 8: dependent LDS round trip: ds_write_b64 -> ds_read_b64 (broadcast address) -> fma
  x = seed + lane;
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) {
    lds[128 + lane] = x;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    const double m = *(volatile double*)&lds[128 + 5];
    x = __builtin_fma(m, 1e-3, x);
  }
  asm volatile("" : "+v"(x));
  t1 = tick(x); if (lane == 0) ticks[8] = t1 - t0; acc += x;
  // 9: dependent ds_read_b64 pointer chase (read only)
  int idx = lane & 63;
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) { const double v = *(volatile double*)&lds[idx]; idx = (int(v) + i) & 63; }
  x += idx; t1 = tick(x); if (lane == 0) ticks[9] = t1 - t0; acc += idx;
  // 10: independent broadcast ds_read_b64 + fma (8 rows), multipliers from LDS instead of readlane
  t0 = tick8(z);
#pragma unroll
  for (int i = 0; i < REP / 8; ++i) {
    double m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = lds[(i * 8 + k) & 127];
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = __builtin_fma(-m[k], piv, z[k]);
  }
  for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(z[k]));
  t1 = tick8(z); if (lane == 0) ticks[10] = t1 - t0; for (int k = 0; k < 8; ++k) acc += z[k];
  // 11: dependent v_rcp_f64
  x = seed + 2.0 + lane;
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) x = __builtin_amdgcn_rcp(x) + 2.0;
  asm volatile("" : "+v"(x));
  t1 = tick(x); if (lane == 0) ticks[11] = t1 - t0; acc += x;
  // 12: dependent v_mul_f64
  x = seed + lane * 1e-9;
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) x = x * y;
  asm volatile("" : "+v"(x));
  t1 = tick(x); if (lane == 0) ticks[12] = t1 - t0; acc += x;
  // 13: DPP row_newbcast on fp64 fma? (64-bit DPP): use update_dpp on both words + fma
  x = seed + lane;
  t0 = tick(x);
#pragma unroll
  for (int i = 0; i < REP; ++i) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x150 + 3, 0xF, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x150 + 3, 0xF, 0xF, false);
    x = __builtin_fma(__hiloint2double(hi, lo), 1e-3, x);
  }
  asm volatile("" : "+v"(x));
  t1 = tick(x); if (lane == 0) ticks[13] = t1 - t0; acc += x;
  // 14: barrier cost with 4 waves: measured in the 256-thread launch below (ticks[14])
  t0 = clock64();
#pragma unroll
  for (int i = 0; i < 64; ++i) __syncthreads();
  t1 = clock64(); if (lane == 0) ticks[14] = t1 - t0;
  // 15: clock64 overhead
  t0 = clock64();
#pragma unroll
  for (int i = 0; i < 64; ++i) { long long t = clock64(); asm volatile("" : "+s"(t)); }
  t1 = clock64(); if (lane == 0) ticks[15] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  double* out; long long* ticks;
  hipMalloc(&out, 1024 * sizeof(double)); hipMalloc(&ticks, 16 * sizeof(long long));
  const char* names[16] = {"dep v_fma_f64", "indep v_fma_f64 x8", "dep readlane2+fma(sgpr)", "indep readlane2+fma x8", "dep v_rsq_f64(+add)", "dep ocml rsqrt(+add)",
                           "dep mfma_f64_16x16x4", "indep mfma_f64 x4", "dep ds_write+ds_read bcast+fma", "dep ds_read_b64", "indep ds_read bcast+fma x8",
                           "dep v_rcp_f64(+add)", "dep v_mul_f64", "dep dpp row_newbcast2+fma", "__syncthreads x64", "clock64 x64"};
  for (int threads : {64, 256}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(probes, dim3(1), dim3(threads), 0, 0, out, ticks, 1.0 + 1e-6 * rep);
      hipDeviceSynchronize();
    }
    long long h[16]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    printf("--- %d threads (wave 0, cycles per operation)\n", threads);
    for (int i = 0; i < 16; ++i) printf("%-34s %8.1f\n", names[i], double(h[i]) / (i >= 14 ? 64 : REP));
  }
  return 0;
}
