// placeholder (real kernel follows)
#pragma once
#include "gpu_rt.h"
#include "../../../include/qmgpu.h"
namespace qmk {
constexpr int WBC_SCRATCH_DOUBLES = 8;
struct WbcArgs { const qmgpu_problem* P; int batch, variant; const double* xDes; const double* uDes; const double* rbd; const int* mode; const double* period; const double* time; double* inputLast; double* out; int* status; double* scratch; };
__global__ void wbc_kernel(WbcArgs a) {}
}
