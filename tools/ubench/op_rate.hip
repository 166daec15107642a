// Issue rate of individual gfx950 VALU opcodes (dependency-free streams, all CUs busy): which ones run at the 2-cycle
// rate of v_add_f32 / v_mul_f32 and which at 4 or 8 cycles per wave64 instruction.
// build: hipcc --offload-arch=gfx950 -O3 -o op_rate op_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 2048;
#define BODY8(INS)                                                                                                      \
    asm volatile(INS(0, 1) INS(1, 2) INS(2, 3) INS(3, 4) INS(4, 5) INS(5, 6) INS(6, 7) INS(7, 0)                        \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)                      \
                 : "v"(a), "v"(b), "s"(sa)                                                                              \
                 : "vcc");
#define OP2(NAME, MN) \
    static const char* n_##NAME = MN;
#define KERNEL(NAME, INS)                                                                                               \
    __global__ __launch_bounds__(256) void NAME(float* out, float a, float b, float sa) {                               \
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        for (int i = 0; i < ITERS; i++) { BODY8(INS) }                                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                                    \
    }
#define D(d) "%" #d
#define I_ADD(d, s) "v_add_f32 %" #d ", %" #s ", %" #d "\n"
#define I_SUB(d, s) "v_sub_f32 %" #d ", %" #s ", %" #d "\n"
#define I_MULS(d, s) "v_mul_f32 %" #d ", %10, %" #s "\n"
#define I_MULL(d, s) "v_mul_f32 %" #d ", 0x3f8ccccd, %" #s "\n"
#define I_FMA(d, s) "v_fma_f32 %" #d ", %" #s ", %8, %9\n"
#define I_FMAC(d, s) "v_fmac_f32 %" #d ", %" #s ", %8\n"
#define I_MOV(d, s) "v_mov_b32 %" #d ", %" #s "\n"
#define I_AND(d, s) "v_and_b32 %" #d ", %" #s ", %8\n"
#define I_XOR(d, s) "v_xor_b32 %" #d ", %" #s ", %8\n"
#define I_ADDU(d, s) "v_add_u32 %" #d ", %" #s ", %8\n"
#define I_LSHL(d, s) "v_lshlrev_b32 %" #d ", 3, %" #s "\n"
#define I_ASHR(d, s) "v_ashrrev_i32 %" #d ", 31, %" #s "\n"
#define I_MIN(d, s) "v_min_f32 %" #d ", %" #s ", %8\n"
#define I_MAX(d, s) "v_max_f32 %" #d ", %" #s ", %8\n"
#define I_MED3(d, s) "v_med3_f32 %" #d ", %" #s ", %8, %9\n"
#define I_MAX3(d, s) "v_max3_f32 %" #d ", %" #s ", %8, %9\n"
#define I_CVTF(d, s) "v_cvt_f32_i32 %" #d ", %" #s "\n"
#define I_CVTI(d, s) "v_cvt_i32_f32 %" #d ", %" #s "\n"
#define I_MULLO(d, s) "v_mul_lo_u32 %" #d ", %" #s ", %8\n"
#define I_MAD24(d, s) "v_mad_u32_u24 %" #d ", %" #s ", %8, %9\n"
#define I_CMP(d, s) "v_cmp_lt_f32 vcc, %" #s ", %8\n"
#define I_CND(d, s) "v_cndmask_b32 %" #d ", %" #s ", %8, vcc\n"
#define I_BFI(d, s) "v_bfi_b32 %" #d ", %" #s ", %8, %9\n"
#define I_PKADD(d, s) "v_pk_add_f32 %" #d ", %" #s ", %" #s "\n"
#define I_ADD3(d, s) "v_add_f32_e64 %" #d ", %" #s ", |%" #d "|\n"
#define I_MULNEG(d, s) "v_mul_f32_e64 %" #d ", -%" #s ", %8\n"
#define I_PERM(d, s) "v_perm_b32 %" #d ", %" #s ", %8, %9\n"
#define I_LSHLADD(d, s) "v_lshl_add_u32 %" #d ", %" #s ", 2, %8\n"
#define I_ADD3U(d, s) "v_add3_u32 %" #d ", %" #s ", %8, %9\n"
#define I_RDL(d, s) "v_readlane_b32 s20, %" #s ", 3\n"
KERNEL(k_add, I_ADD) KERNEL(k_sub, I_SUB) KERNEL(k_muls, I_MULS) KERNEL(k_mull, I_MULL) KERNEL(k_fma, I_FMA) KERNEL(k_fmac, I_FMAC)
KERNEL(k_mov, I_MOV) KERNEL(k_and, I_AND) KERNEL(k_xor, I_XOR) KERNEL(k_addu, I_ADDU) KERNEL(k_lshl, I_LSHL) KERNEL(k_ashr, I_ASHR)
KERNEL(k_min, I_MIN) KERNEL(k_max, I_MAX) KERNEL(k_med3, I_MED3) KERNEL(k_max3, I_MAX3) KERNEL(k_cvtf, I_CVTF) KERNEL(k_cvti, I_CVTI)
KERNEL(k_mullo, I_MULLO) KERNEL(k_mad24, I_MAD24) KERNEL(k_cmp, I_CMP) KERNEL(k_cnd, I_CND) KERNEL(k_bfi, I_BFI)
KERNEL(k_add3, I_ADD3) KERNEL(k_mulneg, I_MULNEG) KERNEL(k_perm, I_PERM) KERNEL(k_lshladd, I_LSHLADD) KERNEL(k_add3u, I_ADD3U)
typedef void (*kfn)(float*, float, float, float);
int main() {
    float* d; (void)hipMalloc(&d, 256 * 2048 * 8 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char* n; kfn f; } ks[] = {{"v_add_f32", k_add}, {"v_sub_f32", k_sub}, {"v_mul_f32 sgpr", k_muls}, {"v_mul_f32 literal", k_mull},
        {"v_fma_f32", k_fma}, {"v_fmac_f32", k_fmac}, {"v_mov_b32", k_mov}, {"v_and_b32", k_and}, {"v_xor_b32", k_xor}, {"v_add_u32", k_addu},
        {"v_lshlrev_b32", k_lshl}, {"v_ashrrev_i32", k_ashr}, {"v_min_f32", k_min}, {"v_max_f32", k_max}, {"v_med3_f32", k_med3}, {"v_max3_f32", k_max3},
        {"v_cvt_f32_i32", k_cvtf}, {"v_cvt_i32_f32", k_cvti}, {"v_mul_lo_u32", k_mullo}, {"v_mad_u32_u24", k_mad24}, {"v_cmp_lt_f32", k_cmp},
        {"v_cndmask_b32", k_cnd}, {"v_bfi_b32", k_bfi}, {"v_add_f32_e64 |x|", k_add3}, {"v_mul_f32_e64 -x", k_mulneg}, {"v_perm_b32", k_perm},
        {"v_lshl_add_u32", k_lshladd}, {"v_add3_u32", k_add3u}};
    for (int waves_per_simd : {2, 4}) {
        const int blocks = 256 * waves_per_simd;
        for (auto& k : ks) {
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 1.0001f, 1.00001f);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double insts_per_simd = (double)ITERS * 8 * waves_per_simd;
            printf("%-22s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", k.n, waves_per_simd, ms * 1e6 / insts_per_simd * 2.4);
        }
    }
    return 0;
}
