// simt_emu.h -- TEST-ONLY host emulation of the tiny subset of the HIP programming model our kernels use.
//
// Purpose: this build container has no GPU, and a gpurun round trip takes minutes.  Compiling the *same* kernel
// sources with -DQMGPU_HOST_EMULATION (g++ -std=c++20) runs every workgroup as blockDim.x host threads joined by a
// barrier at __syncthreads(), one workgroup at a time, so the kernel logic (LDS hand-offs, lane roles, index math)
// can be debugged against the oracle in seconds.  It is NOT a product path: nothing in qm_door_amd/ loads the
// emulation library, libqmgpu.so is built by hipcc only and refuses to run without a HIP device.
#pragma once
#include <barrier>
#include <limits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x = 1, y = 1, z = 1;
  dim3() = default;
  dim3(unsigned a, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct alignas(16) double2 { double x, y; };
inline thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
inline std::barrier<>* g_emuBarrier = nullptr;
inline std::vector<std::unique_ptr<std::barrier<>>> g_emuWaveBarriers;  // one per 64-lane wavefront of the running workgroup

#define __global__
#define QM_ONE_WAVE_PER_SIMD
// LDS is not zeroed between workgroups on the GPU: the emulation poisons it so that a read of never-written LDS that reaches the
// arithmetic (a zero-padded tile operand, say) turns the results into NaN instead of passing by luck
#define QM_POISON_LDS(ptr, count) do { if (threadIdx.x == 0) for (int i_ = 0; i_ < int(count); ++i_) (ptr)[i_] = std::numeric_limits<qmk::real>::quiet_NaN(); __syncthreads(); } while (0)
inline int qmOpaqueLane(int v) { return v; }
inline void __builtin_amdgcn_s_setprio(int) {}
#define QM_TICK_DECL
#define QM_TICK(slot)
#define QM_TICK_FLUSH(base, cond)
#define QM_STREAM_STORE(ptr, value) (*(ptr) = (value))
#define QM_STREAM_LOAD(ptr) (*(ptr))
#define QM_L2_LOAD(ptr) (*(ptr))
#define QM_CONSTANT_REF(T, lvalue) (lvalue)
#define QM_CONSTANT_PTR(T, ptr) (ptr)
#define QM_SCHED_FENCE()
#define QM_LDS_BARRIER() __syncthreads()
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__
inline void __syncthreads() { g_emuBarrier->arrive_and_wait(); }
// lanes of one wavefront run in lockstep on the GPU; the emulation needs a real rendezvous wherever a kernel relies on that
#define QM_WAVE_SYNC() g_emuWaveBarriers[(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)) / 64]->arrive_and_wait()
inline void sincos(double a, double* s, double* c) { *s = std::sin(a); *c = std::cos(a); }
inline float sinf(float a) { return std::sin(a); }
inline float cosf(float a) { return std::cos(a); }
using std::acos; using std::fabs; using std::fma; using std::fmax; using std::fmin; using std::log; using std::sqrt; using std::sin; using std::cos;
using std::max; using std::min;

// ---- wavefront exchange primitives.  The lanes of a wavefront exchange values through a buffer the EMULATION owns (one per
// wavefront of the running workgroup: two 128-double halves used alternately, so one barrier per exchange suffices -- a lane can
// only be one exchange ahead of the slowest lane of its wavefront).  The `scratch` argument of the primitives is ignored: product
// kernels need no LDS for these operations (they are register / DPP / v_readlane instructions on the GPU).
inline thread_local unsigned g_emuXchg = 0;
constexpr int kEmuWaveScratch = 4096;
inline std::vector<std::unique_ptr<double[]>> g_emuWaveScratch;
inline double* emuWaveScratch() { return g_emuWaveScratch[(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)) / 64].get(); }
inline double* emuXchgBuf(double*) { return emuWaveScratch() + 128 * ((g_emuXchg++) & 1u); }

// The primitives are templates over the arithmetic type T (double for the fp64 build, float for the fp32 build of the MPC kernels); the
// exchange buffers hold doubles either way (every float is exactly representable).
template <class T> inline T qmShflXor(T v, int mask, T* scratch = nullptr) {
  const unsigned lane = threadIdx.x & 63u;
  double* buf = emuXchgBuf(nullptr); (void)scratch;
  buf[lane] = double(v);
  QM_WAVE_SYNC();
  return T(buf[lane ^ unsigned(mask)]);
}

template <int R, bool FIRST, class T> inline void qmFmacRowBcast(T& acc, T bc, T m, T* = nullptr) {   // v_fmac_*_dpp row_newbcast:R
  const unsigned lane = threadIdx.x & 63u;
  double* buf = emuXchgBuf(nullptr);
  buf[lane] = double(bc);
  QM_WAVE_SYNC();
  acc += T(buf[(lane & ~15u) + unsigned(R)]) * m;
}
template <int G, class T> inline T qmReplicateRow(T v, T* = nullptr) {   // ds_bpermute with address 16 G + (lane & 15)
  const unsigned lane = threadIdx.x & 63u;
  double* buf = emuXchgBuf(nullptr);
  buf[lane] = double(v);
  QM_WAVE_SYNC();
  return T(buf[16u * unsigned(G) + (lane & 15u)]);
}
template <class T> inline T qmReplicateRow0(T v, T* s = nullptr) { return qmReplicateRow<0>(v, s); }
template <class T> inline T qmHalfXor32(T v, bool) { return qmShflXor(v, 32); }   // v_permlane32_swap
inline int qmReadLaneInt(int v, int src);
template <class T> inline T qmRowXor16(T v, bool) { return qmShflXor(v, 16); }    // v_permlane16_swap

template <class T> inline T qmReadLane(T v, int src, T* scratch = nullptr) {
  const unsigned lane = threadIdx.x & 63u;
  double* buf = emuXchgBuf(nullptr); (void)scratch;
  buf[lane] = double(v);
  QM_WAVE_SYNC();
  return T(buf[unsigned(src) & 63u]);
}
inline int qmReadLaneInt(int v, int src) { return qmReadLane<int>(v, src); }
inline unsigned long long qmBallot(bool p) {
  double* buf = emuXchgBuf(nullptr);
  buf[threadIdx.x & 63u] = p ? 1.0 : 0.0;
  QM_WAVE_SYNC();
  unsigned long long m = 0;
  for (int i = 0; i < 64; ++i) if (buf[i] != 0.0) m |= 1ull << i;
  return m;
}
inline int qmFirstBit(unsigned long long m) { return __builtin_ctzll(m); }
inline int qmPopCount(unsigned long long m) { return __builtin_popcountll(m); }

template <class T> struct QmGatherT {
  T vals[64];
  T get(int src) const { return vals[src & 63]; }
};
template <class T> inline QmGatherT<T> qmGather(T v, T* scratch = nullptr) {
  const unsigned lane = threadIdx.x & 63u;
  double* buf = emuXchgBuf(nullptr); (void)scratch;
  buf[lane] = double(v);
  QM_WAVE_SYNC();
  QmGatherT<T> g;
  for (int i = 0; i < 64; ++i) g.vals[i] = T(buf[i]);
  return g;
}
template <class T, class Op> inline T emuButterfly(T v, Op op) {
  const unsigned lane = threadIdx.x & 63u;
  QmGatherT<T> g = qmGather(v);
  T cur[64], nxt[64];
  for (int i = 0; i < 64; ++i) cur[i] = g.vals[i];
  for (int m = 1; m <= 32; m <<= 1) { for (int i = 0; i < 64; ++i) nxt[i] = op(cur[i], cur[i ^ m]);   // same pairing order as the DPP butterfly of gpu_rt.h
    for (int i = 0; i < 64; ++i) cur[i] = nxt[i]; }
  return cur[lane];
}
template <class T> inline T qmAllSum(T v, T* = nullptr) { return emuButterfly(v, [](T a, T b) { return a + b; }); }
template <class T> inline T qmAllMax(T v, T* = nullptr) { return emuButterfly(v, [](T a, T b) { return std::fmax(a, b); }); }
template <class T> inline T qmAllMin(T v, T* = nullptr) { return emuButterfly(v, [](T a, T b) { return std::fmin(a, b); }); }

// 16x16x4 matrix-core instruction on host threads: lane l supplies a = A[l % 16][l / 16], b = B[l / 16][l % 16]; the accumulator maps of the
// HARDWARE are emulated (measured on gfx950, tools/probe_mfma.hip; cdna4 ISA): register r of lane l is C[l / 16 + 4 r][l % 16] for fp64 and
// C[4 (l / 16) + r][l % 16] for fp32 -- so the fp32 build's permutation of the A rows (qmARow, gpu_rt.h) is exercised on the CPU tier too.
template <class T> struct QmAccT { T v[4]; T& operator[](int i) { return v[i]; } const T& operator[](int i) const { return v[i]; } };
template <class T> struct QmD2T { T x, y; };
template <class T> inline void emuMfmaTile(QmAccT<T>& c, const double* A, const double* B, unsigned lane) {
  const unsigned j = lane & 15u, h = lane >> 4;
  for (unsigned r = 0; r < 4; ++r) {
    const unsigned i = sizeof(T) == 8 ? h + 4 * r : 4 * h + r;
    T acc = c.v[r];
    for (unsigned k = 0; k < 4; ++k) acc += T(A[k * 16 + i]) * T(B[k * 16 + j]);
    c.v[r] = acc;
  }
}
template <class T> inline void qmMfma(QmAccT<T>& c, T a, T b, T* scratch = nullptr) {
  const unsigned lane = threadIdx.x & 63u;
  double* buf = emuXchgBuf(nullptr); (void)scratch;
  buf[lane] = double(a); buf[64 + lane] = double(b);
  QM_WAVE_SYNC();
  emuMfmaTile(c, buf, buf + 64, lane);
}
// all upper-triangle tiles of one k step with a single exchange (own region behind the 256 doubles of the plain exchanges)
template <int TP, class T> inline void qmMfmaUpper(QmAccT<T>* acc, const T* a, const T* b, T* scratch = nullptr) {
  const unsigned lane = threadIdx.x & 63u;
  (void)scratch;
  double* buf = emuWaveScratch() + 256 + (TP * 128) * ((g_emuXchg++) & 1u);
  for (int t = 0; t < TP; ++t) { buf[t * 128 + lane] = double(a[t]); buf[t * 128 + 64 + lane] = double(b[t]); }
  QM_WAVE_SYNC();
  int t = 0;
  for (int ti = 0; ti < TP; ++ti)
    for (int tj = ti; tj < TP; ++tj, ++t) emuMfmaTile(acc[t], buf + ti * 128, buf + tj * 128 + 64, lane);
}
inline double __longlong_as_double(long long v) { double d; std::memcpy(&d, &v, 8); return d; }
inline double qmRsqrt(double x) { return 1.0 / std::sqrt(x); }
inline float qmRsqrt(float x) { return 1.0f / std::sqrt(x); }
inline double qmRsqrtPos(double x) { return 1.0 / std::sqrt(x); }
inline double qmMulNoFma(double a, double b) { volatile double p = a * b; return p; }
inline double qmSubNoFma(double a, double b) { volatile double d = a - b; return d; }
inline float qmRsqrtPos(float x) { return 1.0f / std::sqrt(x); }
inline double qmRcpPos(double x) { return 1.0 / x; }
inline float qmRcpPos(float x) { return 1.0f / x; }
#define QM_KEEP(x) (void)(x)
#define QM_OPAQUE_LDS(T, name, p) T* name = (p)
#define QM_LDS_CONST_PTR(T) const T*
#define QM_TO_LDS_PTR(T, p) ((const T*)(p))
namespace qmk {
using QmAcc = QmAccT<real>;
using QmD2 = QmD2T<real>;
using QmGather = QmGatherT<real>;
constexpr int qmARow(int p) { return sizeof(real) == 8 ? p : (p >> 2) + 4 * (p & 3); }   // as gpu_rt.h
}  // namespace qmk
using qmk::QmAcc; using qmk::QmD2; using qmk::QmGather; using qmk::qmARow;

template <class F> void emuLaunch(F&& body, dim3 grid, dim3 block) {
  const unsigned nt = block.x * block.y * block.z;
  std::barrier<> bar(nt);
  g_emuBarrier = &bar;
  g_emuWaveBarriers.clear();
  g_emuWaveScratch.clear();
  for (unsigned w = 0; w < (nt + 63) / 64; ++w) g_emuWaveScratch.emplace_back(new double[kEmuWaveScratch]);
  for (unsigned w = 0; w < (nt + 63) / 64; ++w) g_emuWaveBarriers.emplace_back(std::make_unique<std::barrier<>>(std::min(64u, nt - 64 * w)));
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&, t]() {
      blockDim = block; gridDim = grid;
      threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
      for (unsigned b = 0; b < grid.x * grid.y * grid.z; ++b) {
        blockIdx = dim3(b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y));
        body();
        bar.arrive_and_wait();  // all lanes leave the workgroup before the (static) LDS is reused
      }
    });
  for (auto& t : th) t.join();
  g_emuBarrier = nullptr;
}

// ---- runtime API subset -------------------------------------------------------------------------------------------------
using hipError_t = int;
using hipStream_t = void*;
using hipEvent_t = double*;
constexpr int hipSuccess = 0;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline const char* hipGetErrorString(hipError_t) { return "emulation"; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount };
// compute units of the emulated device: QMGPU_EMU_CUS (default 256) -- a test sets it below its batch to reach the launch shapes chosen for batch > CUs
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { const char* e = std::getenv("QMGPU_EMU_CUS"); *v = e ? std::atoi(e) : 256; return 0; }
// device memory is not zeroed by the driver: the emulation fills it with 0xFF (NaN as double, -1 as int32) so that reads of scratch
// that no kernel wrote show up in the results
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); if (*p) std::memset(*p, 0xFF, n ? n : 1); return *p ? 0 : 1; }
inline hipError_t hipFree(void* p) { std::free(p); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return 0; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return 0; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return 0; }
constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2;
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return 0; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* l, int* g) { *l = 0; *g = 0; return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }   // (launches are synchronous on the host: every event has happened)
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new double(0); return 0; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new double(0); return 0; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
#define QM_LAUNCH(kernel, grid, block, stream, ...) emuLaunch([&]() { kernel(__VA_ARGS__); }, dim3(grid), dim3(block))
#define QM_LAUNCH_DYN(kernel, grid, block, shmemBytes, stream, ...) emuLaunch([&]() { kernel(__VA_ARGS__); }, dim3(grid), dim3(block))
static qmk::real g_emuDynamicLds[20480];   /* the CU's whole LDS at fp64; one workgroup runs at a time, and a kernel and the functions it calls see the same array (extern __shared__ on the GPU) */
#define QM_DYNAMIC_LDS(name) qmk::real* const name = g_emuDynamicLds
#define QM_ALLOW_DYNAMIC_LDS(kernel, bytes) 0
