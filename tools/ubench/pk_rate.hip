// Issue rate of packed vs scalar f32 VALU on gfx950: N dependent-free instructions per wave, all CUs busy.
// build: hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;
__global__ __launch_bounds__(256) void k_scalar(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; i++) {
        asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                     "v_mul_f32 %4, %4, %9\n v_mul_f32 %5, %5, %9\n v_mul_f32 %6, %6, %9\n v_mul_f32 %7, %7, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ __launch_bounds__(256) void k_packed(float* out, float a, float b) {
    v2f x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    v2f aa = {a, a}, bb = {b, b};
    for (int i = 0; i < ITERS; i++) {
        asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                     "v_pk_mul_f32 %4, %4, %9\n v_pk_mul_f32 %5, %5, %9\n v_pk_mul_f32 %6, %6, %9\n v_pk_mul_f32 %7, %7, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(aa), "v"(bb));
    }
    v2f s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
int main() {
    float* d; hipMalloc(&d, 256 * 2048 * 8 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
        for (int which = 0; which < 2; which++) {
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 1.0001f);
                else hipLaunchKernelGGL(k_packed, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 1.0001f);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double insts_per_simd = (double)ITERS * 8 * waves_per_simd;  // wave-instructions issued per SIMD
            printf("%s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", which ? "packed" : "scalar",
                   waves_per_simd, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
        }
    }
    return 0;
}
