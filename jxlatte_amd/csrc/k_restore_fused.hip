// Fused restoration + colour tile kernel (Gab -> EPF -> XYB -> transfer) -- placeholder until the
// LDS-tiled version lands: reports "not covered" so the host uses the stage kernels of k_restore.hip.
#include "jxl_internal.h"
namespace jxl {
bool launch_restore_fused(const float* const in[3], void* const out[3], int h, int w, const int32_t* hf_mul,
                          const int32_t* sharpness, const RestoreParams& p, hipStream_t s) {
    (void)in; (void)out; (void)h; (void)w; (void)hf_mul; (void)sharpness; (void)p; (void)s;
    return false;
}
}  // namespace jxl
