// Issue rate of DEPENDENT f32 VALU chains on gfx950: K independent chains per wave (K = 1, 2, 4, 8), W waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o dep_rate dep_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 2048;
template <int K>
__global__ __launch_bounds__(256) void k_dep(float* out, float a, float b) {
    float x[8];
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
    for (int i = 0; i < ITERS; i++) {
        // 8 instructions per iteration, arranged as K chains (chain j = registers j, j+K, ... all feeding x[j])
        if (K == 1) asm volatile("v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %2\n v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %2\n v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %2\n v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %2\n" : "+v"(x[0]) : "v"(a), "v"(b));
        if (K == 2) asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n v_mul_f32 %0, %0, %3\n v_mul_f32 %1, %1, %3\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n v_mul_f32 %0, %0, %3\n v_mul_f32 %1, %1, %3\n" : "+v"(x[0]), "+v"(x[1]) : "v"(a), "v"(b));
        if (K == 4) asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n v_mul_f32 %0, %0, %5\n v_mul_f32 %1, %1, %5\n v_mul_f32 %2, %2, %5\n v_mul_f32 %3, %3, %5\n" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "v"(a), "v"(b));
        if (K == 8) asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_mul_f32 %4, %4, %9\n v_mul_f32 %5, %5, %9\n v_mul_f32 %6, %6, %9\n v_mul_f32 %7, %7, %9\n" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b));
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// the same with a DPP move feeding each add (pattern of the streaming kernel: mov_dpp + dependent add)
__global__ __launch_bounds__(256) void k_dppdep(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, t0, t1;
    for (int i = 0; i < ITERS; i++) {
        asm volatile("v_mov_b32_dpp %2, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                     "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n"
                     "s_nop 1\n v_mov_b32_dpp %2, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                     "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n s_nop 1\n"
                     : "+v"(x0), "+v"(x1), "=&v"(t0), "=&v"(t1) : "v"(a), "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1;
}
typedef void (*kfn)(float*, float, float);
int main() {
    float* d; (void)hipMalloc(&d, 256 * 2048 * 8 * sizeof(float));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char* n; kfn f; } ks[] = {{"1 chain", k_dep<1>}, {"2 chains", k_dep<2>}, {"4 chains", k_dep<4>}, {"8 chains", k_dep<8>}, {"2 chains, dpp mov + add", k_dppdep}};
    for (int waves_per_simd : {1, 2, 3, 4, 8}) {
        const int blocks = 256 * waves_per_simd;
        for (auto& k : ks) {
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 1.0001f);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const double insts_per_simd = (double)ITERS * 8 * waves_per_simd;
            printf("%-26s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD, %.2f per wave (2.4 GHz)\n", k.n, waves_per_simd,
                   ms * 1e6 / insts_per_simd * 2.4, ms * 1e6 / (ITERS * 8.0) * 2.4);
        }
    }
    return 0;
}
