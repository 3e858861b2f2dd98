// Single include point for the GPU runtime.  The product is built by hipcc for gfx950 only; the
// QMGPU_HOST_EMULATION branch exists solely so tests/emu can run the same kernel sources on host
// threads in the GPU-less build container (see tests/emu/simt_emu.h) -- it is never part of libqmgpu.so.
#pragma once
#ifdef QMGPU_HOST_EMULATION
#include "simt_emu.h"
#else
#include <hip/hip_runtime.h>
// Lanes of a wavefront execute in lockstep and LDS operations of one wavefront complete in issue order, so an intra-wave LDS
// hand-off needs no hardware barrier -- only a compiler scheduling fence.
#define QM_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
// butterfly exchange inside one wavefront (DPP / ds_swizzle); `scratch` (64 doubles of LDS) is only used by the host emulation
__device__ __forceinline__ double qmShflXor(double v, int mask, double* scratch) { (void)scratch; return __shfl_xor(v, mask, 64); }
// value of lane `src` (wave-uniform, compile-time constant after unrolling) broadcast through an SGPR pair: v_readlane_b32 x 2
__device__ __forceinline__ double qmReadLane(double v, int src, double* scratch) {
  (void)scratch;
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// One v_mfma_f64_16x16x4_f64: C[16x16] += A[16x4] B[4x16].  Operand / result layout measured on gfx950 (tools/probe_mfma.hip):
//   lane l supplies a = A[l % 16][l / 16] and b = B[l / 16][l % 16]; accumulator register r of lane l is C[l / 16 + 4 r][l % 16].
typedef double QmAcc __attribute__((ext_vector_type(4)));
typedef double QmD2 __attribute__((ext_vector_type(2)));   // 16-byte load/store unit (a native vector: stays in registers, unlike HIP's double2 struct)
__device__ __forceinline__ void qmMfma(QmAcc& c, double a, double b, double* scratch) { (void)scratch; c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double qmRsqrt(double x) { return rsqrt(x); }
// upper-triangle tile set of a symmetric product: acc[(ti,tj), ti <= tj] += A_ti B_tj for one k step of 4
template <int TP> __device__ __forceinline__ void qmMfmaUpper(QmAcc* acc, const double* a, const double* b, double* scratch) {
  int t = 0;
#pragma unroll
  for (int ti = 0; ti < TP; ++ti)
#pragma unroll
    for (int tj = ti; tj < TP; ++tj, ++t) qmMfma(acc[t], a[ti], b[tj], scratch);
}
// "every lane's value, addressable by (compile-time) lane index": v_readlane at the point of use
struct QmGather {
  double v;
  __device__ __forceinline__ double get(int src) const { return qmReadLane(v, src, nullptr); }
};
__device__ __forceinline__ QmGather qmGather(double v, double* scratch) { (void)scratch; return QmGather{v}; }
// wavefront all-reduces (butterfly: every lane ends with the same value, bit for bit -- the operation is commutative and both
// partners of a step combine the same pair).  Steps 1, 2 (quad permutes), 4 (row_half_mirror: quads are uniform by then) and 8
// (row_mirror) are DPP moves, steps 16 and 32 the row / half-wave swaps of gfx950 (v_permlane16_swap, v_permlane32_swap): ~25
// VALU instructions, no LDS round trip.  The __shfl_xor butterfly compiles to twelve ds_bpermute_b32 on a dependent chain; the
// interior point of the WBC runs about ten all-reduces per iteration.
template <int CTRL> __device__ __forceinline__ double qmDppMove(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double qmRowXor16(double v, bool oddRow) {    // value of lane ^ 16
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(int(oddRow ? r1[0] : r1[1]), int(oddRow ? r0[0] : r0[1]));
}
__device__ __forceinline__ double qmHalfXor32(double v, bool upper) {   // value of lane ^ 32
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(int(upper ? r1[0] : r1[1]), int(upper ? r0[0] : r0[1]));
}
template <class Op> __device__ __forceinline__ double qmAllReduce(double v, Op op) {
  const unsigned lane = threadIdx.x;
  v = op(v, qmDppMove<0xB1>(v));    // quad_perm [1,0,3,2]
  v = op(v, qmDppMove<0x4E>(v));    // quad_perm [2,3,0,1]
  v = op(v, qmDppMove<0x141>(v));   // row_half_mirror
  v = op(v, qmDppMove<0x140>(v));   // row_mirror
  v = op(v, qmRowXor16(v, (lane >> 4) & 1));
  v = op(v, qmHalfXor32(v, (lane & 32) != 0));
  return v;
}
__device__ __forceinline__ double qmAllSum(double v, double* scratch) { (void)scratch; return qmAllReduce(v, [](double a, double b) { return a + b; }); }
__device__ __forceinline__ double qmAllMax(double v, double* scratch) { (void)scratch; return qmAllReduce(v, [](double a, double b) { return fmax(a, b); }); }
__device__ __forceinline__ double qmAllMin(double v, double* scratch) { (void)scratch; return qmAllReduce(v, [](double a, double b) { return fmin(a, b); }); }
// Workgroup barrier that orders LDS traffic only: waits for this wavefront's LDS operations, then s_barrier.  Unlike
// __syncthreads() it does not drain outstanding global loads (vmcnt), so a register-staged prefetch of the next stage stays in
// flight across the barriers of the current one.  Use only where the data exchanged through the barrier lives in LDS.
#define QM_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// kernels that can only ever have one wavefront per SIMD (LDS bound): let the register allocator use the whole file instead of
// spilling to stay under the two-waves budget
#define QM_ONE_WAVE_PER_SIMD __attribute__((amdgpu_waves_per_eu(1, 1)))
// scheduling fence: keeps the compiler from hoisting a long run of v_readlane broadcasts (two SGPRs each) ahead of their uses
#define QM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// a per-lane integer the optimiser cannot see through: values derived from it (the sixty lane == k ? 1 : 0 seeds of the AD sweep) are not
// hoisted out of the loop it is refreshed in
__device__ __forceinline__ int qmOpaqueLane(int v) { asm volatile("" : "+v"(v)); return v; }
#define QM_POISON_LDS(ptr, count)   // host emulation only: fills LDS with NaN at kernel start
#define QM_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#define QM_LAUNCH_DYN(kernel, grid, block, shmemBytes, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmemBytes, stream, __VA_ARGS__)
// dynamic LDS (keeps the base 16-byte aligned: no static __shared__ may precede it in the same kernel)
#define QM_DYNAMIC_LDS(name) extern __shared__ __attribute__((aligned(16))) double name[]
#define QM_ALLOW_DYNAMIC_LDS(kernel, bytes) hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
#endif
