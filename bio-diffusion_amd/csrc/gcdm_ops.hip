// libgcdm_ops.so -- the module-level operators (forward + backward) behind plug point 3, the non-production configurations and the
// training objective.  One translation unit, independent of libgcdm_hip.so (the fused sampling path); C ABI in include/gcdm_ops.h.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o bio-diffusion_amd/libgcdm_ops.so bio-diffusion_amd/csrc/gcdm_ops.hip
#include "gcdm_ops.hip.h"
#include "../../include/gcdm_ops.h"
