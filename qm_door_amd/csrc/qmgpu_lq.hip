// lq_node_kernel in a translation unit of its own, compiled at -O2 (qm_door_amd/build.py; qmgpu_api.hip only declares the kernel, QM_LQ_EXTERN).  Measured, not
// derived: the same source is 2.7 % faster at -O2 than at -O3 (0.456 -> 0.443 ms per launch on the bench workload, profiles/r04i_variant_timing.txt: the whole library at
// -O2, this kernel alone at -O2, this kernel alone at -O3 in its own unit -- the gain follows the flag, not the unit), with bit-identical results.  The other kernels do
// not move with the flag, and wbc_kernel's history says to leave their compilation alone.
#define QM_LQ_UNIT
#include "kernels/lq_kernel.h"
