// PCIe duplex ceiling for the streaming boundary: host->device reads and device->host writes by KERNELS through page-locked
// aliases (what jxl_vardct_commit_coeffs_i16 / read_output_begin queue since r5), alone and together, against the runtime's
// copies (SDMA).   hipcc --offload-arch=gfx950 -O3 tools/ubench/pcie_duplex.hip -o tools/ubench/pcie_duplex
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy(const v4i* __restrict__ s, v4i* __restrict__ d, size_t n) {
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const size_t MB = argc > 1 ? atoi(argv[1]) : 64, bytes = MB << 20, n = bytes / 16;
    const int grid = argc > 2 ? atoi(argv[2]) : 128, reps = 20;
    void *hin, *hout, *din, *dout, *ain, *aout;
    CK(hipHostMalloc(&hin, bytes, hipHostMallocDefault)); CK(hipHostMalloc(&hout, bytes, hipHostMallocDefault));
    CK(hipMalloc(&din, bytes)); CK(hipMalloc(&dout, bytes));
    CK(hipHostGetDevicePointer(&ain, hin, 0)); CK(hipHostGetDevicePointer(&aout, hout, 0));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    auto run = [&](const char* what, int mode) {
        double best = 1e9;
        for (int w = 0; w < 3; w++) {
            const double a = now();
            for (int r = 0; r < reps; r++) {
                if (mode & 1) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, s1, (const v4i*)ain, (v4i*)din, n);
                if (mode & 2) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, s2, (const v4i*)dout, (v4i*)aout, n);
                if (mode & 4) (void)hipMemcpyAsync(din, hin, bytes, hipMemcpyHostToDevice, s1);
                if (mode & 8) (void)hipMemcpyAsync(hout, dout, bytes, hipMemcpyDeviceToHost, s2);
            }
            (void)hipStreamSynchronize(s1); (void)hipStreamSynchronize(s2);
            best = std::min(best, now() - a);
        }
        const int dirs = ((mode & 5) ? 1 : 0) + ((mode & 10) ? 1 : 0);
        printf("%-44s %7.2f GB/s per direction, %7.2f GB/s total\n", what, bytes * reps / best / 1e9, dirs * bytes * reps / best / 1e9);
    };
    printf("%zu MB per transfer, kernel grid %d x 256\n", MB, grid);
    run("kernel reads host (H2D)", 1);
    run("kernel writes host (D2H)", 2);
    run("kernel H2D + kernel D2H together", 3);
    run("runtime copy H2D", 4);
    run("runtime copy D2H", 8);
    run("runtime H2D + runtime D2H together", 12);
    run("kernel H2D + runtime D2H together", 9);
    run("runtime H2D + kernel D2H together", 6);
    return 0;
}
