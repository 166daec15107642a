// Fused restoration + colour tile kernel for gfx950 (kernel template: restore_fused_body.h): the float-plane instantiations -- the
// hot variants of the 4K / batch workload -- and the entry points. The other output sinks are instantiated in
// k_restore_fused_gen.hip (run-time-generic) and k_restore_fused_q.hip (quantised PNG / u16 formats fixed at compile time).
//
// Replaces Frame.java:505-679, OpsinInverseMatrix.java:105-142, JXLImage.java:244-258, ImageBuffer.java:129-147,
// PNGWriter.java:65,105-111 (see the header).
#include "restore_fused_body.h"

namespace jxl {

bool fill_restore_fused_args(const float* const in[3], void* const out[3], int h, int w, const int32_t* hf_mul,
                             const int32_t* sharpness, const RestoreParams& p, FusedArgs& a) {
    if (w < 8 || h < 8) return false;  // mirror fix-up assumes at most one reflection within the halo
    if (p.epf_iters > 0 && (!hf_mul || !sharpness)) return false;
    for (int c = 0; c < 3; c++) {
        a.in[c] = in[c];
        a.out[c] = out[c];
    }
    a.hf_mul = hf_mul;
    a.sharpness = sharpness;
    a.W = w;
    a.H = h;
    a.bw = (w + 7) >> 3;
    a.p = p;
    return true;
}


// which kernel instantiation the arguments select: Gaborish on/off, EPF iterations, sink kind
// (epf_iters 0..3, or 4 = the 13-tap iteration alone of the two-launch form: three bits)
int restore_fused_variant(const FusedArgs& a) { return (a.p.gab ? 64 : 0) | (a.p.epf_iters & 7) << 3 | sink_kind_of(a.p); }

// host_args: the n frames' argument blocks (all of one variant), dev_args: the same blocks in device memory
void launch_restore_fused_batch(const FusedArgs* host_args, const FusedArgs* dev_args, int n, hipStream_t s) {
    if (n <= 0) return;
    const int sk = sink_kind_of(host_args[0].p);
    if (sk == SK_PLAIN) launch_fused_batch_sk<SK_PLAIN>(host_args, dev_args, n, s);
    else if (sk == SK_GENERIC) launch_restore_fused_gen(nullptr, host_args, dev_args, n, s);
    else launch_restore_fused_q(sk, nullptr, host_args, dev_args, n, s);
}

bool launch_restore_fused(const float* const in[3], void* const out[3], int h, int w, const int32_t* hf_mul,
                          const int32_t* sharpness, const RestoreParams& p, hipStream_t s) {
    FusedArgs a;
    if (!fill_restore_fused_args(in, out, h, w, hf_mul, sharpness, p, a)) return false;
    const int sk = sink_kind_of(p);
    if (sk == SK_PLAIN) {
        // 4x1 patches on 512 threads everywhere: twice the waves per CU of 4x2 patches for the same LDS footprint. The
        // 3-iteration variant used to run 4x2 patches (fewer tap loads for its 13-tap first iteration) but spilled 169 VGPRs
        // there: 841 us per 4K frame against 259 us with 4x1 patches and a 128-register budget (JXL_RESTORE_PH=2 selects 4x2,
        // float planes only)
        static const int ph_env = getenv("JXL_RESTORE_PH") ? atoi(getenv("JXL_RESTORE_PH")) : 0;
        if (ph_env == 2 && p.epf_iters <= 3) {  // (4 = the 13-tap iteration alone: 4x1 patches only)
            const int it = p.epf_iters;
            if (p.gab) {
                if (it == 0) launch_tph<true, 0, SK_PLAIN, 2>(a, s);
                else if (it == 1) launch_tph<true, 1, SK_PLAIN, 2>(a, s);
                else if (it == 2) launch_tph<true, 2, SK_PLAIN, 2>(a, s);
                else launch_tph<true, 3, SK_PLAIN, 2>(a, s);
            } else {
                if (it == 0) launch_tph<false, 0, SK_PLAIN, 2>(a, s);
                else if (it == 1) launch_tph<false, 1, SK_PLAIN, 2>(a, s);
                else if (it == 2) launch_tph<false, 2, SK_PLAIN, 2>(a, s);
                else launch_tph<false, 3, SK_PLAIN, 2>(a, s);
            }
        } else {
            launch_fused_sk<SK_PLAIN>(a, s);
        }
    } else if (sk == SK_GENERIC) {
        launch_restore_fused_gen(&a, nullptr, nullptr, 0, s);
    } else {
        launch_restore_fused_q(sk, &a, nullptr, nullptr, 0, s);
    }
    return true;
}

}  // namespace jxl
