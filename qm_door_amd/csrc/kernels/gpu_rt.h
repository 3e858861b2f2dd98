// Single include point for the GPU runtime.  The product is built by hipcc for gfx950 only; the
// QMGPU_HOST_EMULATION branch exists solely so tests/emu can run the same kernel sources on host
// threads in the GPU-less build container (see tests/emu/simt_emu.h) -- it is never part of libqmgpu.so.
#pragma once
#ifdef QMGPU_HOST_EMULATION
#include "real.h"
#include "simt_emu.h"
#else
#include <hip/hip_runtime.h>
#include "real.h"
namespace qmk {
// Intra-wavefront LDS hand-off (lane a writes, lane b of the SAME wavefront reads).  Hardware: the lanes of a wavefront execute one
// instruction stream and the LDS operations of one wavefront complete in issue order, so no s_barrier and no s_waitcnt is needed.
// Compiler: the hand-off is a release by the writer and an acquire by the reader at wavefront scope, and it is stated as such --
//   fence release (wavefront) ; llvm.amdgcn.wave.barrier ; fence acquire (wavefront)
// The fences emit no instruction (SIMemoryLegalizer drops wavefront-scope fences: nothing to wait for), but at the IR level they are
// what forbids moving, merging or forwarding LDS accesses across the hand-off; the wave barrier keeps the machine scheduler from
// moving anything across it.  Round 2 used the bare wave barrier, whose memory semantics depend on the LLVM version (IntrNoMem until
// 2023; this ROCm 7.2 compiler declares it without a memory attribute, i.e. already as a clobber -- DESIGN.md section 4.7).
// -DQM_WAVE_SYNC_BARE restores the bare form (tools/wbc_variants.py builds it as one of the variants that must agree).
#ifdef QM_WAVE_SYNC_BARE
#define QM_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define QM_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif
// ---- a register value as 32-bit words (v_readlane / DPP / permlane move 32 bits): one word for fp32, two for fp64
__device__ __forceinline__ double qmFromWords(int lo, int hi, double) { return __hiloint2double(hi, lo); }
__device__ __forceinline__ float qmFromWords(int lo, int, float) { return __int_as_float(lo); }
__device__ __forceinline__ int qmLoWord(double v) { return __double2loint(v); }
__device__ __forceinline__ int qmHiWord(double v) { return __double2hiint(v); }
__device__ __forceinline__ int qmLoWord(float v) { return __float_as_int(v); }
__device__ __forceinline__ int qmHiWord(float) { return 0; }
// butterfly exchange inside one wavefront; the trailing pointer argument of the cross-lane primitives is unused on the GPU (the host
// emulation of tests/emu once exchanged through it)
__device__ __forceinline__ real qmShflXor(real v, int mask, real* = nullptr) { return __shfl_xor(v, mask, 64); }
// value of lane `src` (wave-uniform, compile-time constant after unrolling) broadcast through SGPRs: v_readlane_b32 per word
__device__ __forceinline__ real qmReadLane(real v, int src, real* = nullptr) {
  const int lo = __builtin_amdgcn_readlane(qmLoWord(v), src);
  const int hi = sizeof(real) == 8 ? __builtin_amdgcn_readlane(qmHiWord(v), src) : 0;
  return qmFromWords(lo, hi, real());
}
__device__ __forceinline__ int qmReadLaneInt(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
// the lanes of the wavefront where `p` holds, as a mask (v_cmp into a scalar pair), and the lowest set bit of such a mask (s_ff1)
__device__ __forceinline__ unsigned long long qmBallot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int qmFirstBit(unsigned long long m) { return __builtin_ctzll(m); }
__device__ __forceinline__ int qmPopCount(unsigned long long m) { return __builtin_popcountll(m); }
// One 16x16x4 matrix-core instruction: C[16x16] += A[16x4] B[4x16]  (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32).
// Operand layout (both types; measured on gfx950 for fp64, tools/probe_mfma.hip): lane l supplies a = A[l % 16][l / 16] and
// b = B[l / 16][l % 16].  The accumulators differ: register r of lane l is
//     fp64: C[l / 16 + 4 r][l % 16]            fp32: C[4 (l / 16) + r][l % 16].
// The kernels are written for the fp64 map.  The fp32 build keeps every initialisation / store of an accumulator unchanged and
// instead permutes the ROWS OF A it feeds: the lane that supplies physical row p = l % 16 loads logical row qmARow(p) =
// p / 4 + 4 (p % 4), so that physical row 4 h + r of the result is logical row h + 4 r -- the fp64 map.
typedef double QmAccD __attribute__((ext_vector_type(4)));
typedef float QmAccF __attribute__((ext_vector_type(4)));
template <class T> struct QmAccOf { typedef QmAccD type; };
template <> struct QmAccOf<float> { typedef QmAccF type; };
typedef QmAccOf<real>::type QmAcc;
typedef real QmD2 __attribute__((ext_vector_type(2)));   // load/store unit of the register-staged prefetch (a native vector: stays in registers, unlike HIP's double2 struct)
__device__ __forceinline__ void qmMfma(QmAccD& c, double a, double b, double* = nullptr) { c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void qmMfma(QmAccF& c, float a, float b, float* = nullptr) { c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ constexpr int qmARow(int p) { return sizeof(real) == 8 ? p : (p >> 2) + 4 * (p & 3); }
__device__ __forceinline__ double qmRsqrt(double x) { return rsqrt(x); }
__device__ __forceinline__ float qmRsqrt(float x) { return rsqrtf(x); }
// a * b and a - b rounded on their own: the optimiser may not fuse them into one multiply-add (results that feed exact comparisons shared with the host oracle)
__device__ __forceinline__ double qmMulNoFma(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double qmSubNoFma(double a, double b) { return __dsub_rn(a, b); }
// 1 / sqrt(x) for a positive, normal x on a dependent chain: v_rsq_f64 (~2^-26) and one third-order correction -- the arithmetic of the
// library routine without its scaling of denormals and its special-case selects (5 dependent instructions instead of 9)
__device__ __forceinline__ double qmRsqrtPos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(-x * y, y, 1.0);
  return __builtin_fma(y * e, __builtin_fma(e, 0.375, 0.5), y);
}
__device__ __forceinline__ float qmRsqrtPos(float x) { return __builtin_amdgcn_rsqf(x); }
// 1 / x for a positive, normal x on a dependent chain: v_rcp_f64 and two Newton steps (5 dependent instructions; a full IEEE division is 13 with its scaling)
__device__ __forceinline__ double qmRcpPos(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
  return __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
}
__device__ __forceinline__ float qmRcpPos(float x) { return __builtin_amdgcn_rcpf(x); }
// acc += (bc of lane R of this lane's row of 16 lanes) * m in ONE instruction (DPP row_newbcast, legal on 64-bit operands since gfx90a):
// the multiplier broadcast of a row operation without the v_readlane pair + wait state + separate multiply-add.  FIRST puts the two wait
// states a DPP source needs after a VALU write in front (the hazard recogniser does not look into inline assembly).
template <int R, bool FIRST> __device__ __forceinline__ void qmFmacRowBcast(double& acc, double bc, double m, double* = nullptr) {
  if (FIRST) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bc), "v"(m), "n"(R));
  else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bc), "v"(m), "n"(R));
}
template <int R, bool FIRST> __device__ __forceinline__ void qmFmacRowBcast(float& acc, float bc, float m, float* = nullptr) {
  if (FIRST) asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bc), "v"(m), "n"(R));
  else asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bc), "v"(m), "n"(R));
}
// the value of lane 16 G + (lane & 15) in every lane: one row of 16 lanes replicated into all four (ds_bpermute: the LDS crossbar, no LDS memory)
template <int G> __device__ __forceinline__ real qmReplicateRow(real v, real* = nullptr) {
  const int addr = (16 * G + (int(threadIdx.x) & 15)) * 4;
  const int lo = __builtin_amdgcn_ds_bpermute(addr, qmLoWord(v));
  const int hi = sizeof(real) == 8 ? __builtin_amdgcn_ds_bpermute(addr, qmHiWord(v)) : 0;
  return qmFromWords(lo, hi, real());
}
__device__ __forceinline__ real qmReplicateRow0(real v, real* s = nullptr) { return qmReplicateRow<0>(v, s); }
// pointer to LDS that keeps its address space through a function call (a generic pointer to LDS compiles to flat loads)
#define QM_LDS_CONST_PTR(T) const T __attribute__((address_space(3)))*
#define QM_TO_LDS_PTR(T, p) ((const T __attribute__((address_space(3)))*)(p))
// Read-only problem data (model, settings, constant weight matrices) seen through the CONSTANT address space.  A kernel that has stored to global memory can
// no longer prove that a later load from a plain global pointer is unclobbered, so the backend fetches even wave-uniform constants with VECTOR loads
// (ad_node_kernel: 353 global_load per wavefront and not one s_load in its sweep loop, each waited for on a one-wavefront-per-SIMD chain; round 3, PMC:
// SQ_INSTS_VMEM_RD 3.0e6 vs SQ_INSTS_SMEM 6.9e4 per launch).  Loads from address space 4 are invariant by definition: wave-uniform addresses become
// s_load_* through the scalar cache again.  Only for memory nothing writes while the kernel runs (the qmgpu_problem copy, R').
// (The pointer passes through an empty asm in scalar registers: a plain generic -> constant -> generic cast pair is folded away before the
// address-space inference sees it.)
template <class T> __device__ __forceinline__ const T* qmConstantPtr(const T* p) {
  const T __attribute__((address_space(4)))* p4 = (const T __attribute__((address_space(4)))*)(p);
  asm("" : "+s"(p4));
  return (const T*)p4;
}
#define QM_CONSTANT_REF(T, lvalue) (*qmk::qmConstantPtr<T>(&(lvalue)))
#define QM_CONSTANT_PTR(T, ptr) (qmk::qmConstantPtr<T>(ptr))
// streaming store: data written once and not read again by this kernel (goes out with the non-temporal cache policy)
#define QM_STREAM_STORE(ptr, value) __builtin_nontemporal_store((value), (ptr))
#define QM_STREAM_LOAD(ptr) __builtin_nontemporal_load(ptr)   // read once, never again by this CU
// a load that is served by the L2 and does not allocate in the CU's vector L1 (agent-scope relaxed atomic load: global_load ... sc1)
#define QM_L2_LOAD(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// keeps a value in a register at this point: loads placed before it stay unconditional (the compiler otherwise sinks an LDS read
// into the select that consumes it and pays the LDS latency once per branch)
#define QM_KEEP(x) asm volatile("" : "+v"(x))
// LDS pointer whose value the compiler may not fold: accesses through it use ONE address register plus the immediate offset of the
// instruction.  With the address known at compile time (dynamic LDS starts at a link-time constant) a fully unrolled loop materialises a
// separate address per access in a scalar register, spills them into vector lanes and pays v_readlane + v_mov per LDS instruction.
#define QM_OPAQUE_LDS(T, name, p) T __attribute__((address_space(3)))* name = (T __attribute__((address_space(3)))*)(p); asm volatile("" : "+v"(name))
// upper-triangle tile set of a symmetric product: acc[(ti,tj), ti <= tj] += A_ti B_tj for one k step of 4
template <int TP> __device__ __forceinline__ void qmMfmaUpper(QmAcc* acc, const real* a, const real* b, real* = nullptr) {
  int t = 0;
#pragma unroll
  for (int ti = 0; ti < TP; ++ti)
#pragma unroll
    for (int tj = ti; tj < TP; ++tj, ++t) qmMfma(acc[t], a[ti], b[tj]);
}
// "every lane's value, addressable by (compile-time) lane index": v_readlane at the point of use
struct QmGather {
  real v;
  __device__ __forceinline__ real get(int src) const { return qmReadLane(v, src); }
};
__device__ __forceinline__ QmGather qmGather(real v, real* = nullptr) { return QmGather{v}; }
// wavefront all-reduces (butterfly: every lane ends with the same value, bit for bit -- the operation is commutative and both
// partners of a step combine the same pair).  Steps 1, 2 (quad permutes), 4 (row_half_mirror: quads are uniform by then) and 8
// (row_mirror) are DPP moves, steps 16 and 32 the row / half-wave swaps of gfx950 (v_permlane16_swap, v_permlane32_swap): ~25
// VALU instructions, no LDS round trip.  The __shfl_xor butterfly compiles to twelve ds_bpermute_b32 on a dependent chain; the
// interior point of the WBC runs about ten all-reduces per iteration.
template <int CTRL> __device__ __forceinline__ real qmDppMove(real v) {
  int lo = qmLoWord(v), hi = qmHiWord(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  if (sizeof(real) == 8) hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return qmFromWords(lo, hi, real());
}
__device__ __forceinline__ real qmRowXor16(real v, bool oddRow) {    // value of lane ^ 16
  const unsigned lo = qmLoWord(v), hi = qmHiWord(v);
  const auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return qmFromWords(int(oddRow ? r0[0] : r0[1]), int(oddRow ? r1[0] : r1[1]), real());
}
__device__ __forceinline__ real qmHalfXor32(real v, bool upper) {   // value of lane ^ 32
  const unsigned lo = qmLoWord(v), hi = qmHiWord(v);
  const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return qmFromWords(int(upper ? r0[0] : r0[1]), int(upper ? r1[0] : r1[1]), real());
}
template <class Op> __device__ __forceinline__ real qmAllReduce(real v, Op op) {
  const unsigned lane = threadIdx.x;
  v = op(v, qmDppMove<0xB1>(v));    // quad_perm [1,0,3,2]
  v = op(v, qmDppMove<0x4E>(v));    // quad_perm [2,3,0,1]
  v = op(v, qmDppMove<0x141>(v));   // row_half_mirror
  v = op(v, qmDppMove<0x140>(v));   // row_mirror
  v = op(v, qmRowXor16(v, (lane >> 4) & 1));
  v = op(v, qmHalfXor32(v, (lane & 32) != 0));
  return v;
}
__device__ __forceinline__ real qmAllSum(real v, real* = nullptr) { return qmAllReduce(v, [](real a, real b) { return a + b; }); }
__device__ __forceinline__ real qmAllMax(real v, real* = nullptr) { return qmAllReduce(v, [](real a, real b) { return fmax(a, b); }); }
__device__ __forceinline__ real qmAllMin(real v, real* = nullptr) { return qmAllReduce(v, [](real a, real b) { return fmin(a, b); }); }
}  // namespace qmk
using qmk::QmAcc; using qmk::QmD2; using qmk::QmGather;
// Workgroup barrier that orders LDS traffic only: waits for this wavefront's LDS operations, then s_barrier.  Unlike
// __syncthreads() it does not drain outstanding global loads (vmcnt), so a register-staged prefetch of the next stage stays in
// flight across the barriers of the current one.  Use only where the data exchanged through the barrier lives in LDS.
#define QM_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// kernels that can only ever have one wavefront per SIMD (LDS bound): let the register allocator use the whole file instead of
// spilling to stay under the two-waves budget
#define QM_ONE_WAVE_PER_SIMD __attribute__((amdgpu_waves_per_eu(1, 1)))
// scheduling fence: keeps the compiler from hoisting a long run of v_readlane broadcasts (two SGPRs each) ahead of their uses
#define QM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// Phase clocks of the profiling build (-DQM_RICCATI_TIMING, tools/riccati_phase_probe.py): s_memtime deltas summed per slot in (scalar)
// registers, written once at the end into a device symbol.  The product build compiles every QM_TICK to nothing.
#ifdef QM_RICCATI_TIMING
#define QM_AD_WG_CLOCKS 9216
namespace qmk { __device__ unsigned long long qmRiccatiTicks[2048]; __device__ unsigned long long qmAdWgClock[2 * QM_AD_WG_CLOCKS]; }
#define QM_TICK_DECL unsigned long long qmT = clock64(), qmTs[24] = {}
#define QM_TICK(slot) do { const unsigned long long n_ = clock64(); qmTs[slot] += n_ - qmT; qmT = n_; } while (0)
#define QM_TICK_FLUSH(base, cond) do { if (cond) for (int i_ = 0; i_ < 24; ++i_) qmk::qmRiccatiTicks[(base) + i_] += qmTs[i_]; } while (0)
#else
#define QM_TICK_DECL
#define QM_TICK(slot)
#define QM_TICK_FLUSH(base, cond)
#endif
// a per-lane integer the optimiser cannot see through: values derived from it (the lane == k ? 1 : 0 seeds of the AD sweep) are not
// hoisted out of the loop it is refreshed in
__device__ __forceinline__ int qmOpaqueLane(int v) { asm volatile("" : "+v"(v)); return v; }
#define QM_POISON_LDS(ptr, count)   // host emulation only: fills LDS with NaN at kernel start
#define QM_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#define QM_LAUNCH_DYN(kernel, grid, block, shmemBytes, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmemBytes, stream, __VA_ARGS__)
// dynamic LDS (keeps the base 16-byte aligned: no static __shared__ may precede it in the same kernel)
#define QM_DYNAMIC_LDS(name) extern __shared__ __attribute__((aligned(16))) qmk::real name[]
#define QM_ALLOW_DYNAMIC_LDS(kernel, bytes) hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
#endif
