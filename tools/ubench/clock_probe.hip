// What does the shader clock do under load? s_memtime (shader clock counter) against s_memrealtime (100 MHz constant) in a probe
// workgroup, alone and next to a chip-filling VALU / LDS load.   hipcc --offload-arch=gfx950 -O2 clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k_probe(uint64_t* out, int spin) {
    const uint64_t c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    float a = threadIdx.x;
    for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;
    const uint64_t c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
    if (a == 12345.f) out[2] = 1;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_load(float* sink, int iters) {
    __shared__ float lds[4096];
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    lds[threadIdx.x] = a0;
    __syncthreads();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // independent multiplies and adds, no FMA (the library's instruction mix)
            a0 = a0 * 1.0001f; a1 = a1 + 0.5f; a2 = a2 * 0.9999f; a3 = a3 + 0.25f;
            a4 = a4 * 1.0001f; a5 = a5 + 0.5f; a6 = a6 * 0.9999f; a7 = a7 + 0.25f;
        } else {          // LDS reads mixed in
            a0 = a0 * 1.0001f + lds[(threadIdx.x + i) & 4095];
            a1 = a1 + lds[(threadIdx.x * 3 + i) & 4095];
            a2 = a2 * 0.9999f; a3 = a3 + 0.25f;
        }
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) sink[0] = 1;
}

int main() {
    uint64_t* d;
    float* sink;
    hipMalloc(&d, 64);
    hipMalloc(&sink, 64);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    uint64_t h[2];
    auto probe = [&](const char* what) {
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, sb, d, 200000);
            hipStreamSynchronize(sb);
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("%-34s memtime ticks %9llu  realtime ticks (100 MHz) %7llu  -> memtime runs at %.1f MHz\n", what,
                   (unsigned long long)h[0], (unsigned long long)h[1], (double)h[0] / ((double)h[1] / 100.0));
        }
    };
    probe("idle chip");
    for (int mode = 0; mode < 2; mode++)
        for (int wgs : {256, 1024, 4096}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            const int iters = 400000;
            hipEventRecord(e0, sa);
            if (mode == 0) hipLaunchKernelGGL(k_load<0>, dim3(wgs), dim3(256), 0, sa, sink, iters * 2048 / wgs);
            else hipLaunchKernelGGL(k_load<1>, dim3(wgs), dim3(256), 0, sa, sink, iters * 2048 / wgs / 2);
            hipEventRecord(e1, sa);
            char what[64];
            snprintf(what, sizeof what, "%s load, %d workgroups", mode ? "VALU+LDS" : "VALU", wgs);
            probe(what);
            hipStreamSynchronize(sa);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double insts = (mode == 0 ? 8.0 : 4.0) * (double)(iters * 2048 / wgs / (mode ? 2 : 1)) * wgs * 4;  // wave-instructions
            printf("    load kernel %.2f ms: %.0f G VALU wave-inst/s (+ loop overhead)\n", ms, insts / ms / 1e6);
        }
    return 0;
}
