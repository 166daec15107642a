// Modular inverse transforms on gfx950: inverse Squeeze lifting steps, RCT, int -> float.
//
// Replaces (J/ = java/com/traneptora/jxlatte/):
//   J/frame/modular/ModularChannel.java:23-47    tendency
//   J/frame/modular/ModularChannel.java:361-413  inverseHorizontalSqueeze / inverseVerticalSqueeze
//   J/frame/modular/ModularStream.java:255-326   RCT
//   J/frame/Frame.java:430-455                   modular ints -> frame buffer
//
// int32 arithmetic wraps (Java): all adds/muls are done in uint32; `/` truncates toward zero.
// The squeeze recurrence is serial along the squeeze axis (left = previously OUTPUT odd sample feeds
// the non-linear tendency()), so parallelism is rows x channels (H) or columns x channels (V).
//   V step: lane = column, rows walked in order: every load/store is a coalesced row segment.
//   H step: lane = row. A wave owns 64 rows; 64-column chunks of avg/res are staged through LDS so
//           that global traffic stays row-contiguous while each lane walks its own row.
#include "jxl_internal.h"

namespace jxl {

__device__ __forceinline__ int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__device__ __forceinline__ int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }

// ModularChannel.tendency (ModularChannel.java:23-47)
__device__ __forceinline__ int32_t tendency(int32_t a, int32_t b, int32_t c) {
    if (a >= b && b >= c) {
        int32_t x = wadd(wsub(wsub(wmul(4, a), wmul(3, c)), b), 6) / 12;
        const int32_t d = wmul(2, wsub(a, b));
        const int32_t e = wmul(2, wsub(b, c));
        if (wsub(x, (x & 1)) > d) x = wadd(d, 1);
        if (wadd(x, (x & 1)) > e) x = e;
        return x;
    }
    if (a <= b && b <= c) {
        int32_t x = wsub(wsub(wsub(wmul(4, a), wmul(3, c)), b), 6) / 12;
        const int32_t d = wmul(2, wsub(a, b));
        const int32_t e = wmul(2, wsub(b, c));
        if (wadd(x, (x & 1)) < d) x = wsub(d, 1);
        if (wsub(x, (x & 1)) < e) x = e;
        return x;
    }
    return 0;
}

// inverseVerticalSqueeze (ModularChannel.java:389-413): lane = column
__global__ __launch_bounds__(64) void k_inv_vsqueeze(const int32_t* __restrict__ avg, int ah, const int32_t* __restrict__ res,
                                                     int rh, int w, int32_t* __restrict__ out) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= w) return;
    int32_t top = 0;
    int32_t a = rh > 0 ? avg[x] : 0;
    for (int y = 0; y < rh; y++) {
        const int32_t residu = res[(int64_t)y * w + x];
        const int32_t nextAvg = y + 1 < ah ? avg[(int64_t)(y + 1) * w + x] : a;
        const int32_t t = y > 0 ? top : a;
        const int32_t diff = wadd(residu, tendency(t, a, nextAvg));
        const int32_t first = wadd(a, diff / 2);
        const int32_t second = wsub(first, diff);
        out[(int64_t)(2 * y) * w + x] = first;
        out[(int64_t)(2 * y + 1) * w + x] = second;
        top = second;
        a = nextAvg;
    }
    if (ah > rh) out[(int64_t)(2 * rh) * w + x] = avg[(int64_t)rh * w + x];
}

// inverseHorizontalSqueeze (ModularChannel.java:361-387): lane = row, LDS-staged 64x64 chunks
__global__ __launch_bounds__(64) void k_inv_hsqueeze(const int32_t* __restrict__ avg, int aw, const int32_t* __restrict__ res,
                                                     int rw, int h, int32_t* __restrict__ out) {
    __shared__ int32_t sA[64 * 65];  // avg chunk [row][col]; overwritten in place by the even outputs
    __shared__ int32_t sR[64 * 65];  // res chunk [row][col]; overwritten in place by the odd outputs
    const int lane = threadIdx.x;
    const int y0 = blockIdx.x * 64;
    const int rows = min(64, h - y0);
    const int ow = aw + rw;
    int32_t left = 0;
    int32_t a_next_chunk = 0;  // avg[x0 + 64] look-ahead of the lane's own row
    for (int x0 = 0; x0 < rw; x0 += 64) {
        const int cols = min(64, rw - x0);
        __syncthreads();
        for (int r = 0; r < rows; r++) {
            if (lane < cols) {
                sA[r * 65 + lane] = avg[(int64_t)(y0 + r) * aw + x0 + lane];
                sR[r * 65 + lane] = res[(int64_t)(y0 + r) * rw + x0 + lane];
            }
        }
        __syncthreads();
        if (lane < rows) {
            const int64_t rowA = (int64_t)(y0 + lane) * aw;
            // avg[x0 + cols] (first avg of the next chunk, or the odd tail) if it exists
            const bool has_next = x0 + cols < aw;
            a_next_chunk = has_next ? avg[rowA + x0 + cols] : 0;
            for (int i = 0; i < cols; i++) {
                const int x = x0 + i;
                const int32_t a = sA[lane * 65 + i];
                const int32_t residu = sR[lane * 65 + i];
                int32_t nextAvg;
                if (i + 1 < cols) nextAvg = sA[lane * 65 + i + 1];
                else nextAvg = has_next ? a_next_chunk : a;  // x + 1 < orig.width ? orig[x+1] : avg
                const int32_t l = x > 0 ? left : a;
                const int32_t diff = wadd(residu, tendency(l, a, nextAvg));
                const int32_t first = wadd(a, diff / 2);
                const int32_t second = wsub(first, diff);
                sA[lane * 65 + i] = first;   // a, residu of column i are consumed; column i+1 is still intact
                sR[lane * 65 + i] = second;
                left = second;
            }
        }
        __syncthreads();
        for (int r = 0; r < rows; r++) {
            const int64_t ro = (int64_t)(y0 + r) * ow + 2 * x0;
            const int j0 = lane, j1 = lane + 64;
            if (j0 < 2 * cols) out[ro + j0] = (j0 & 1) ? sR[r * 65 + (j0 >> 1)] : sA[r * 65 + (j0 >> 1)];
            if (j1 < 2 * cols) out[ro + j1] = (j1 & 1) ? sR[r * 65 + (j1 >> 1)] : sA[r * 65 + (j1 >> 1)];
        }
    }
    if (aw > rw && lane < rows) out[(int64_t)(y0 + lane) * ow + 2 * rw] = avg[(int64_t)(y0 + lane) * aw + rw];
}

void launch_inv_hsqueeze(const int32_t* avg, int aw, const int32_t* res, int rw, int h, int32_t* out, hipStream_t s) {
    if (h <= 0 || aw + rw <= 0) return;
    hipLaunchKernelGGL(k_inv_hsqueeze, dim3((h + 63) / 64), dim3(64), 0, s, avg, aw, res, rw, h, out);
}

void launch_inv_vsqueeze(const int32_t* avg, int ah, const int32_t* res, int rh, int w, int32_t* out, hipStream_t s) {
    if (w <= 0 || ah + rh <= 0) return;
    hipLaunchKernelGGL(k_inv_vsqueeze, dim3((w + 63) / 64), dim3(64), 0, s, avg, ah, res, rh, w, out);
}

// ModularStream.java:270-324; the channel permutation (:325-326) is applied by the host as pointer shuffling
__global__ __launch_bounds__(256) void k_rct(int32_t* v0, int32_t* v1, int32_t* v2, int64_t n, int type) {
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        switch (type) {
        case 1: v2[i] = wadd(v2[i], v0[i]); break;
        case 2: v1[i] = wadd(v1[i], v0[i]); break;
        case 3: { const int32_t a = v0[i]; v2[i] = wadd(v2[i], a); v1[i] = wadd(v1[i], a); break; }
        case 4: v1[i] = wadd(v1[i], wadd(v0[i], v2[i]) >> 1); break;
        case 5: { const int32_t a = v0[i]; const int32_t ac = wadd(a, v2[i]); v1[i] = wadd(v1[i], wadd(a, ac) >> 1); v2[i] = ac; break; }
        case 6: {
            const int32_t b = v1[i], c = v2[i];
            const int32_t tmp = wsub(v0[i], c >> 1);
            const int32_t f = wsub(tmp, b >> 1);
            v0[i] = wadd(f, b);
            v1[i] = wadd(c, tmp);
            v2[i] = f;
            break;
        }
        default: break;
        }
    }
}

void launch_rct(int32_t* v0, int32_t* v1, int32_t* v2, int64_t n, int type, hipStream_t s) {
    if (n <= 0 || type == 0) return;
    int grid = (int)((n + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_rct, dim3(grid), dim3(256), 0, s, v0, v1, v2, n, type);
}

// Frame.java:437-448: out = scale * (a [+ b]) with the int sum wrapping, int -> float conversion first
__global__ __launch_bounds__(256) void k_modular_to_float(const int32_t* a, const int32_t* b, int64_t n, float scale,
                                                          float* out) {
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = b ? scale * (float)wadd(a[i], b[i]) : scale * (float)a[i];
}

void launch_modular_to_float(const int32_t* a, const int32_t* b, int64_t n, float scale, float* out, hipStream_t s) {
    if (n <= 0) return;
    int grid = (int)((n + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_modular_to_float, dim3(grid), dim3(256), 0, s, a, b, n, scale, out);
}

}  // namespace jxl
