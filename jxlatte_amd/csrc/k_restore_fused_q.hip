// Fused restoration kernel (restore_fused_body.h), output sinks with transfer function AND sample format fixed at compile time
// (restore_sink.h, SinkKind): PQ -> u16 planes (BASELINE config C4) and the three interleaved PNG formats (PNGWriter.java:65,
// 105-111, 191-203: sRGB 8 / 16 bit, PQ 16 bit). Same integers as the generic sink (the threshold tables of jxl_fastpow.h);
// what goes away is the per-sample choice of transfer and format (uniform compares and branches around every store: SALU 4.3x
// the float-plane variant, r3 counters) and the narrow stores (u16 formats leave as 8 / 16-byte stores).
#include "restore_fused_body.h"

namespace jxl {

namespace {
template <int SK>
void launch_any(const FusedArgs* single, const FusedArgs* host_args, const FusedArgs* dev_args, int n, hipStream_t s) {
    if (single) launch_fused_sk<SK>(*single, s);
    else launch_fused_batch_sk<SK>(host_args, dev_args, n, s);
}
}  // namespace

void launch_restore_fused_q(int sk, const FusedArgs* single, const FusedArgs* host_args, const FusedArgs* dev_args, int n, hipStream_t s) {
    switch (sk) {
    case SK_PQ_U16: launch_any<SK_PQ_U16>(single, host_args, dev_args, n, s); break;
    case SK_PQ_RGB16: launch_any<SK_PQ_RGB16>(single, host_args, dev_args, n, s); break;
    case SK_SRGB_RGB8: launch_any<SK_SRGB_RGB8>(single, host_args, dev_args, n, s); break;
    case SK_SRGB_RGB16: launch_any<SK_SRGB_RGB16>(single, host_args, dev_args, n, s); break;
    default: launch_restore_fused_gen(single, host_args, dev_args, n, s); break;
    }
}

}  // namespace jxl
