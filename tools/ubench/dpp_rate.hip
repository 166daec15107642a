// Issue rate of DPP-modified f32 VALU (wave_shr:1 / wave_shl:1 / row_shr:1) against the plain forms, and of the division
// helpers, on gfx950: dependency-free instructions, all CUs busy.
// build: hipcc --offload-arch=gfx950 -O3 -o dpp_rate dpp_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 4096;
#define BODY8(INS)                                                                                                      \
    asm volatile(INS(0, 1) INS(1, 2) INS(2, 3) INS(3, 4) INS(4, 5) INS(5, 6) INS(6, 7) INS(7, 0)                        \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)                      \
                 : "v"(a), "v"(b));
#define I_PLAIN(d, s) "v_add_f32 %" #d ", %" #s ", %" #d "\n"
#define I_WSHR(d, s) "v_add_f32_dpp %" #d ", %" #s ", %" #d " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_WSHL(d, s) "v_add_f32_dpp %" #d ", %" #s ", %" #d " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_RSHR(d, s) "v_add_f32_dpp %" #d ", %" #s ", %" #d " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_MOVW(d, s) "v_mov_b32_dpp %" #d ", %" #s " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_ABS(d, s) "v_mul_f32_e64 %" #d ", |%" #s "|, %8\n"
#define I_RCP(d, s) "v_rcp_f32 %" #d ", %" #s "\n"
#define I_DSC(d, s) "v_div_scale_f32 %" #d ", vcc, %" #s ", %8, %" #s "\n"
#define I_FIX(d, s) "v_div_fixup_f32 %" #d ", %" #s ", %8, %9\n"
#define I_FMAS(d, s) "v_div_fmas_f32 %" #d ", %" #s ", %8, %9\n"
#define I_CND(d, s) "v_cmp_lt_f32 vcc, %" #s ", %8\n v_cndmask_b32 %" #d ", %" #s ", %9, vcc\n"
#define I_MAX(d, s) "v_max_f32 %" #d ", %" #s ", %8\n"
#define KERNEL(NAME, INS)                                                                                               \
    __global__ __launch_bounds__(256) void NAME(float* out, float a, float b) {                                         \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        for (int i = 0; i < ITERS; i++) { BODY8(INS) }                                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                                    \
    }
KERNEL(k_plain, I_PLAIN)
KERNEL(k_wshr, I_WSHR)
KERNEL(k_wshl, I_WSHL)
KERNEL(k_rshr, I_RSHR)
KERNEL(k_movw, I_MOVW)
KERNEL(k_abs, I_ABS)
KERNEL(k_rcp, I_RCP)
KERNEL(k_dsc, I_DSC)
KERNEL(k_fix, I_FIX)
KERNEL(k_fmas, I_FMAS)
KERNEL(k_cnd, I_CND)
KERNEL(k_max, I_MAX)
typedef void (*kfn)(float*, float, float);
int main() {
    float* d; hipMalloc(&d, 256 * 2048 * 8 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char* n; kfn f; int per; } ks[] = {{"v_add_f32", k_plain, 1}, {"v_add_f32_dpp wave_shr:1", k_wshr, 1}, {"v_add_f32_dpp wave_shl:1", k_wshl, 1},
        {"v_add_f32_dpp row_shr:1", k_rshr, 1}, {"v_mov_b32_dpp wave_shr:1", k_movw, 1}, {"v_mul_f32 |x|", k_abs, 1}, {"v_rcp_f32", k_rcp, 1},
        {"v_div_scale_f32", k_dsc, 1}, {"v_div_fixup_f32", k_fix, 1}, {"v_div_fmas_f32", k_fmas, 1}, {"v_cmp+v_cndmask", k_cnd, 2}, {"v_max_f32", k_max, 1}};
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int blocks = 256 * waves_per_simd;
        for (auto& k : ks) {
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 1.0001f);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double insts_per_simd = (double)ITERS * 8 * k.per * waves_per_simd;
            printf("%-28s waves/SIMD %d: %.3f ms -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", k.n, waves_per_simd, ms,
                   ms * 1e6 / insts_per_simd * 2.4);
        }
    }
    return 0;
}
