// linesearch_kernel and ddp_rollout_kernel in a translation unit of their own: compiled with LLVM's interprocedural register allocation ON (qm_door_amd/build.py), while qmgpu_api.hip --
// which only declares the kernel (QM_LS_EXTERN) and launches it -- is compiled with it off (wbc_kernel).  With the allocation visible across
// the call, the node evaluation the kernel calls (nodePerformance, 65 KB of code, not inlined) no longer saves and restores 156 callee-saved registers per lane
// through scratch memory: 1,248 B per lane, 0.16 GB of HBM traffic per step, 14 % of the kernel's time.  Same sources, same arithmetic: the `ipra` build variant of the
// whole library (tools/wbc_variants.py) has to agree with the product bit for bit in tests/test_gpu_two_builds.py.
#define QM_LS_UNIT
#include "kernels/linesearch_kernel.h"
#include "kernels/ddp_kernel.h"   // ddp_rollout_kernel calls the same node evaluation
