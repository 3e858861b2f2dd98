// Interface between qmgpu_api.hip (fp64 build of the kernels, namespace qmk) and qmgpu_mpc32.hip (the MPC kernels built a second time
// with real = float, namespace qmk32).  Plain declarations only: nothing here depends on the arithmetic type.
#pragma once
#ifdef QMGPU_HOST_EMULATION
#include "kernels/gpu_rt.h"   // the emulation's stand-ins for hipStream_t / hipEvent_t
#else
#include <hip/hip_runtime_api.h>
#endif

#include <cstddef>
#include <functional>

#include "../../include/qmgpu.h"

namespace qmk32 {

struct Mpc32;
// (count, element size, is scratch) -> device memory owned by the handle
using RawAlloc = std::function<void*(size_t, size_t, bool)>;

// nullptr on failure (HIP error); the object itself is host memory, its device buffers belong to the handle's allocation list
Mpc32* create(const qmgpu_problem& problem, int maxBatch, int maxNodes, hipStream_t stream, const RawAlloc& alloc);
void destroy(Mpc32* p);
bool updateProblem(Mpc32* p, const qmgpu_problem& problem, hipStream_t stream);
// One MPC call in fp32: the caller's fp64 device arrays are converted to fp32 staging, the kernel chain of kernels/mpc_pipeline.h
// runs in fp32, the results are converted back into the caller's fp64 arrays.  ev: optional timing events (qmgpu_api.hip).
bool enqueue(Mpc32* p, hipStream_t stream, const qmgpu_mpc_args* a, double dt, int iterations, int ddpTrials, hipEvent_t* ev);

}  // namespace qmk32
