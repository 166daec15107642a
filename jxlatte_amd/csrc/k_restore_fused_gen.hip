// Fused restoration kernel (restore_fused_body.h), the run-time-generic output sink: any transfer function / output format the
// boundary accepts (float planes behind a transfer function, planar u8 / u16 / int32, the double-precision pow forms ...). The
// formats a PNG writer and the u16 HDR path use have compile-time sinks of their own (k_restore_fused_q.hip).
#include "restore_fused_body.h"

namespace jxl {

// single: one frame (batch arguments null); else the n frames of a batch launch
void launch_restore_fused_gen(const FusedArgs* single, const FusedArgs* host_args, const FusedArgs* dev_args, int n, hipStream_t s) {
    if (single) launch_fused_sk<SK_GENERIC>(*single, s);
    else launch_fused_batch_sk<SK_GENERIC>(host_args, dev_args, n, s);
}

}  // namespace jxl
