// tools/ubench/hy_sens2_probe.hip -- compile-time probe (tools/kres_one.sh tools/ubench/hy_sens2_probe.hip [-DLANES=6 -DBLK=256 -DCOMP=true]):
// registers / scratch / LDS of hychem_sens2_kernel for a lanes-per-trajectory choice, in seconds instead of the whole library's 100 s
#include <hip/hip_runtime.h>
#include "hychem_sens2_kernel.hpp"
#ifndef LANES
#define LANES 6
#endif
#ifndef BLK
#define BLK 256
#endif
#ifndef COMP
#define COMP false
#endif
namespace crnn {
__device__ unsigned g_bounds[2];
template __global__ void hychem_sens2_kernel<9, 10, LANES, BLK, COMP>(const SolveParams, const double *, const HyParams, const HySensParams);
}
