// Memory-system ceiling of the IDCT stage's access pattern: int32 coefficient planes [H][W] -> float planes [H][W], 3 channels, moved in
// "items" of 8 rows x 256 columns (32 DCT8 blocks) per workgroup iteration with 16-byte lane accesses, by a persistent grid -- and, for
// comparison, the same bytes as one linear stream.     hipcc --offload-arch=gfx950 -O3 tile_copy.hip -o tile_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int ROWS, int COLS>  // item = ROWS x COLS samples of each channel; 256 threads
__global__ __launch_bounds__(256) void k_tiles(const int* __restrict__ c0, const int* __restrict__ c1, const int* __restrict__ c2, float* o0,
                                               float* o1, float* o2, int W, int H, int n_items, int prefetch) {
    const int per_row = W / COLS;
    constexpr int V = ROWS * COLS / 4 / 256;  // 16-byte vectors per thread and channel
    v4i cur[3][V], nxt[3][V];
    auto load = [&](int item, v4i (&r)[3][V]) {
        const int ty = item / per_row, tx = item - ty * per_row;
#pragma unroll
        for (int v = 0; v < V; v++) {
            const int i = threadIdx.x + v * 256, y = i / (COLS / 4), x = (i - y * (COLS / 4)) * 4;
            const size_t off = (size_t)(ty * ROWS + y) * W + tx * COLS + x;
            r[0][v] = *(const v4i*)(c0 + off);
            r[1][v] = *(const v4i*)(c1 + off);
            r[2][v] = *(const v4i*)(c2 + off);
        }
    };
    int item = blockIdx.x;
    if (item < n_items) load(item, cur);
    for (; item < n_items; item += gridDim.x) {
        const int nx = item + gridDim.x;
        if (prefetch && nx < n_items) load(nx, nxt);
        const int ty = item / per_row, tx = item - ty * per_row;
#pragma unroll
        for (int v = 0; v < V; v++) {
            const int i = threadIdx.x + v * 256, y = i / (COLS / 4), x = (i - y * (COLS / 4)) * 4;
            const size_t off = (size_t)(ty * ROWS + y) * W + tx * COLS + x;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                v4f f = {(float)cur[c][v].x, (float)cur[c][v].y, (float)cur[c][v].z, (float)cur[c][v].w};
                *(v4f*)((c == 0 ? o0 : c == 1 ? o1 : o2) + off) = f;
            }
        }
        if (prefetch) {
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int v = 0; v < V; v++) cur[c][v] = nxt[c][v];
        } else if (nx < n_items) load(nx, cur);
    }
}

__global__ __launch_bounds__(256) void k_linear(const v4i* __restrict__ c0, const v4i* __restrict__ c1, const v4i* __restrict__ c2, v4f* o0, v4f* o1,
                                                v4f* o2, size_t n4) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const v4i a = c0[i], b = c1[i], c = c2[i];
        o0[i] = v4f{(float)a.x, (float)a.y, (float)a.z, (float)a.w};
        o1[i] = v4f{(float)b.x, (float)b.y, (float)b.z, (float)b.w};
        o2[i] = v4f{(float)c.x, (float)c.y, (float)c.z, (float)c.w};
    }
}

int main() {
    const int W = 3840, H = 2160;
    const size_t n = (size_t)W * H;
    int* c[3];
    float* o[3];
    for (int i = 0; i < 3; i++) {
        hipMalloc(&c[i], n * 4);
        hipMalloc(&o[i], n * 4);
        hipMemset(c[i], 1, n * 4);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto timeit = [&](const char* what, auto launch) {
        for (int i = 0; i < 5; i++) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 50; i++) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %7.1f us  %5.2f TB/s\n", what, ms * 20, 24.0 * n / (ms / 50 * 1e-3) / 1e12);
    };
    for (int grid : {256, 768, 2048, 8192})
        timeit(("linear stream, grid " + std::to_string(grid)).c_str(),
               [&] { hipLaunchKernelGGL(k_linear, dim3(grid), dim3(256), 0, 0, (v4i*)c[0], (v4i*)c[1], (v4i*)c[2], (v4f*)o[0], (v4f*)o[1], (v4f*)o[2], n / 4); });
    for (int pf : {0, 1})
        for (int grid : {256, 768, 1024, 2048}) {
            char w[96];
            snprintf(w, sizeof w, "items 8 x 256, grid %d, %s", grid, pf ? "next item prefetched" : "no prefetch");
            timeit(w, [&] { hipLaunchKernelGGL((k_tiles<8, 256>), dim3(grid), dim3(256), 0, 0, c[0], c[1], c[2], o[0], o[1], o[2], W, H, (W / 256) * (H / 8), pf); });
            snprintf(w, sizeof w, "items 16 x 128, grid %d, %s", grid, pf ? "next item prefetched" : "no prefetch");
            timeit(w, [&] { hipLaunchKernelGGL((k_tiles<16, 128>), dim3(grid), dim3(256), 0, 0, c[0], c[1], c[2], o[0], o[1], o[2], W, H, (W / 128) * (H / 16), pf); });
            snprintf(w, sizeof w, "items 32 x 64, grid %d, %s", grid, pf ? "next item prefetched" : "no prefetch");
            timeit(w, [&] { hipLaunchKernelGGL((k_tiles<32, 64>), dim3(grid), dim3(256), 0, 0, c[0], c[1], c[2], o[0], o[1], o[2], W, H, (W / 64) * (H / 32), pf); });
        }
    return 0;
}
