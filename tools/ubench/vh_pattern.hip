// Memory-system ceiling of the fused squeeze kernel's access pattern (k_modular_vh.hip): one wave = 64 output rows x a segment,
// walked in chunks; per chunk it reads row PIECES of P bytes from three planes (41 + 40 rows of the V inputs, 64 rows of the H
// residuals) and writes 64 row pieces of 2P bytes. No arithmetic. What does the piece size cost?
//   hipcc --offload-arch=gfx950 -O3 vh_pattern.hip -o vh_pattern && ./vh_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// P = piece bytes (64, 128, 256); lanes of one load instruction cover 256 / P rows x P / 4 columns
template <int P, bool LOADS, bool STORES, int VSHIFT = 0>
__global__ __launch_bounds__(64) void k_pat(const int* __restrict__ va, const int* __restrict__ vb, const int* __restrict__ hb, int* __restrict__ o,
                                            int w, int rw, int ow, int nseg, int seg_chunks, int n_tiles) {
    const int per = (n_tiles + 7) >> 3;
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (t >= n_tiles) return;
    const int st = t / nseg, sg = t - st * nseg;
    constexpr int CPR = P / 4, RPI = 64 / CPR;  // columns per row piece, rows per instruction
    const int lane = threadIdx.x, lr = lane / CPR, lc = lane % CPR;
    int acc = 0;
    constexpr int NA = (41 + RPI - 1) / RPI, NB = (40 + RPI - 1) / RPI, NH = 64 / RPI;
    for (int k = 0; k < seg_chunks; k++) {
        const int c0 = (sg * seg_chunks + k) * CPR;  // first column (in V-input samples) of the chunk
        if (LOADS) {
            // V inputs: rows 32 st - 8 .. 32 st + 32 of va (41) and .. + 31 of vb (40); all loads of the chunk issued back to back
            int ta[NA], tb[NB], th[NH];
            const int y0 = 32 * st + 8 + lr;  // (planes are allocated with slack: no row clamps)
#pragma unroll
            for (int i = 0; i < NA; i++) ta[i] = va[(size_t)(y0 + i * RPI) * w + c0 + lc + VSHIFT];
#pragma unroll
            for (int i = 0; i < NB; i++) tb[i] = vb[(size_t)(y0 + i * RPI) * w + c0 + lc + VSHIFT];
#pragma unroll
            for (int i = 0; i < NH; i++) th[i] = hb[(size_t)(64 * st + lr + i * RPI) * rw + c0 + lc];
#pragma unroll
            for (int i = 0; i < NA; i++) acc ^= ta[i];
#pragma unroll
            for (int i = 0; i < NB; i++) acc ^= tb[i];
#pragma unroll
            for (int i = 0; i < NH; i++) acc ^= th[i];
        }
        if (STORES) {
#pragma unroll
            for (int i = 0; i < NH; i++) {
                const int r = lr + i * RPI;
                o[(size_t)(64 * st + r) * ow + 2 * c0 + lc] = acc + r;
                o[(size_t)(64 * st + r) * ow + 2 * c0 + CPR + lc] = acc - r;
            }
        }
    }
    if (acc == 0x12345678) o[0] = acc;
}

template <int P, bool L, bool S, int VS = 0>
float run(const int* va, const int* vb, const int* hb, int* o, int w, int rw, int ow, int H, int seg_px) {
    const int cpr = P / 4, seg_chunks = seg_px / cpr, nseg = rw / seg_px, nst = H / 64, n_tiles = nst * nseg;
    const int per = (n_tiles + 7) / 8;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_pat<P, L, S, VS>), dim3(8 * per), dim3(64), 0, 0, va, vb, hb, o, w, rw, ow, nseg, seg_chunks, n_tiles);
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k_pat<P, L, S, VS>), dim3(8 * per), dim3(64), 0, 0, va, vb, hb, o, w, rw, ow, nseg, seg_chunks, n_tiles);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 5 * 1e3f;
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
// the real kernel's forms: V rows by quarter (lane = (q, j): rows 8 q + i), optional +1 column shift, H residuals and outputs as
// 16-byte accesses (lane = (row, piece)), optional buffer descriptors
template <bool SHIFT, bool X4, bool BUF, bool LOADS, bool STORES>
__global__ __launch_bounds__(64) void k_pat2(const int* __restrict__ va, const int* __restrict__ vb, const int* __restrict__ hb, int* __restrict__ o,
                                             int w, int rw, int ow, int nseg, int seg_chunks, int n_tiles, int H) {
    const int per = (n_tiles + 7) >> 3;
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (t >= n_tiles) return;
    const int st = t / nseg, sg = t - st * nseg;
    const int lane = threadIdx.x, q = lane >> 4, j = lane & 15;
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(va), 0, (H / 2 + 64) * w * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(vb), 0, (H / 2 + 64) * w * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(hb), 0, H * rw * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(o, 0, (unsigned)H * ow * 4, 0x00020000);
    int acc = 0;
    for (int k = 0; k < seg_chunks; k++) {
        const int c0 = (sg * seg_chunks + k) * 16;
        if (LOADS) {
            int ta[8], tb[8], tw[16];
            i32x4 th[4];
            const int cc = c0 + (SHIFT ? 1 : 0) + j;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const size_t off = (size_t)(32 * st + 8 + 8 * q + i) * w + cc;
                ta[i] = BUF ? __builtin_amdgcn_raw_buffer_load_b32(ra, (int)(off * 4), 0, 0) : va[off];
                tb[i] = BUF ? __builtin_amdgcn_raw_buffer_load_b32(rb, (int)(off * 4), 0, 0) : vb[off];
            }
            if (q == 0) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const size_t off = (size_t)(32 * st + i) * w + cc;
                    tw[i] = BUF ? __builtin_amdgcn_raw_buffer_load_b32(ra, (int)(off * 4), 0, 0) : va[off];
                    tw[8 + i] = BUF ? __builtin_amdgcn_raw_buffer_load_b32(rb, (int)(off * 4), 0, 0) : vb[off];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) tw[i] = 0;
            }
            if (X4) {
#pragma unroll
                for (int ps = 0; ps < 4; ps++) {
                    const size_t off = (size_t)(64 * st + 16 * ps + (lane >> 2)) * rw + c0 + 4 * (lane & 3);
                    th[ps] = BUF ? __builtin_amdgcn_raw_buffer_load_b128(rh, (int)(off * 4), 0, 0) : *(const i32x4*)(hb + off);
                }
            } else {
#pragma unroll
                for (int ps = 0; ps < 4; ps++)
#pragma unroll
                    for (int e = 0; e < 4; e++) th[ps][e] = hb[(size_t)(64 * st + 16 * ps + 4 * e + q) * rw + c0 + j];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) acc ^= ta[i] ^ tb[i];
#pragma unroll
            for (int i = 0; i < 16; i++) acc ^= tw[i];
#pragma unroll
            for (int ps = 0; ps < 4; ps++) acc ^= th[ps][0] ^ th[ps][1] ^ th[ps][2] ^ th[ps][3];
        }
        if (STORES) {
            if (X4) {
#pragma unroll
                for (int ps = 0; ps < 8; ps++) {
                    const size_t off = (size_t)(64 * st + 8 * ps + (lane >> 3)) * ow + 2 * c0 + 4 * (lane & 7);
                    i32x4 v = {acc, acc + 1, acc + 2, acc + ps};
                    if (BUF) __builtin_amdgcn_raw_buffer_store_b128(v, ro, (int)(off * 4), 0, 0);
                    else *(i32x4*)(o + off) = v;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int r = q + 4 * i;
                    o[(size_t)(64 * st + r) * ow + 2 * c0 + j] = acc + r;
                    o[(size_t)(64 * st + r) * ow + 2 * c0 + 16 + j] = acc - r;
                }
            }
        }
    }
    if (acc == 0x12345678) o[0] = acc;
}

template <bool SHIFT, bool X4, bool BUF, bool L, bool S>
float run2(const int* va, const int* vb, const int* hb, int* o, int w, int rw, int ow, int H, int seg_px) {
    const int seg_chunks = seg_px / 16, nseg = rw / seg_px, nst = H / 64, n_tiles = nst * nseg;
    const int per = (n_tiles + 7) / 8;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_pat2<SHIFT, X4, BUF, L, S>), dim3(8 * per), dim3(64), 0, 0, va, vb, hb, o, w, rw, ow, nseg, seg_chunks, n_tiles, H);
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k_pat2<SHIFT, X4, BUF, L, S>), dim3(8 * per), dim3(64), 0, 0, va, vb, hb, o, w, rw, ow, nseg, seg_chunks, n_tiles, H);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 5 * 1e3f;
}

// candidate form: V inputs as 16-byte loads of 16 consecutive rows per instruction (lane = (row, piece)), sector-aligned; H
// residuals and outputs as 16-byte accesses; SSHIFT = the output pieces start two samples early (8 bytes before a line)
template <bool SSHIFT, bool LOADS, bool STORES>
__global__ __launch_bounds__(64) void k_pat3(const int* __restrict__ va, const int* __restrict__ vb, const int* __restrict__ hb, int* __restrict__ o,
                                             int w, int rw, int ow, int nseg, int seg_chunks, int n_tiles, int H) {
    const int per = (n_tiles + 7) >> 3;
    const int t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (t >= n_tiles) return;
    const int st = t / nseg, sg = t - st * nseg;
    const int lane = threadIdx.x;
    int acc = 0;
    for (int k = 0; k < seg_chunks; k++) {
        const int c0 = (sg * seg_chunks + k) * 16;
        if (LOADS) {
            i32x4 ta[3], tb[3], th[4];
#pragma unroll
            for (int m = 0; m < 3; m++) {
                const int r = min(16 * m + (lane >> 2), 40);
                const size_t off = (size_t)(32 * st + r) * w + c0 + 4 * (lane & 3);
                ta[m] = *(const i32x4*)(va + off);
                tb[m] = *(const i32x4*)(vb + off);
            }
#pragma unroll
            for (int ps = 0; ps < 4; ps++) th[ps] = *(const i32x4*)(hb + (size_t)(64 * st + 16 * ps + (lane >> 2)) * rw + c0 + 4 * (lane & 3));
#pragma unroll
            for (int m = 0; m < 3; m++) acc ^= ta[m][0] ^ ta[m][1] ^ ta[m][2] ^ ta[m][3] ^ tb[m][0] ^ tb[m][1] ^ tb[m][2] ^ tb[m][3];
#pragma unroll
            for (int ps = 0; ps < 4; ps++) acc ^= th[ps][0] ^ th[ps][1] ^ th[ps][2] ^ th[ps][3];
        }
        if (STORES) {
#pragma unroll
            for (int ps = 0; ps < 8; ps++) {
                const size_t off = (size_t)(64 * st + 8 * ps + (lane >> 3)) * ow + 2 * c0 + 4 * (lane & 7) + (SSHIFT ? 30 : 0);
                struct __attribute__((packed, aligned(4))) u4 { int v[4]; } v = {{acc, acc + 1, acc + 2, acc + ps}};
                *(u4*)(o + off) = v;
            }
        }
    }
    if (acc == 0x12345678) o[0] = acc;
}
template <bool SS, bool L, bool S>
float run3(const int* va, const int* vb, const int* hb, int* o, int w, int rw, int ow, int H, int seg_px) {
    const int seg_chunks = seg_px / 16, nseg = rw / seg_px, nst = H / 64 - 1, n_tiles = nst * nseg;
    const int per = (n_tiles + 7) / 8;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_pat3<SS, L, S>), dim3(8 * per), dim3(64), 0, 0, va, vb, hb, o, w, rw, ow, nseg, seg_chunks, n_tiles, H);
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k_pat3<SS, L, S>), dim3(8 * per), dim3(64), 0, 0, va, vb, hb, o, w, rw, ow, nseg, seg_chunks, n_tiles, H);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / 5 * 1e3f;
}

int main() {
    // the chroma pair of an 8K image, both channels as one plane of double height: V planes 4320 x 3840, H residuals and output 8640 rows
    const int w = 3840, rw = 3840, ow = 7680, H = 8640;
    int *va, *vb, *hb, *o;
    CK(hipMalloc(&va, (size_t)(H / 2 + 64) * w * 4)); CK(hipMalloc(&vb, (size_t)(H / 2 + 64) * w * 4));
    CK(hipMalloc(&hb, (size_t)H * rw * 4)); CK(hipMalloc(&o, (size_t)H * ow * 4));
    CK(hipMemset(va, 1, (size_t)(H / 2 + 64) * w * 4)); CK(hipMemset(vb, 2, (size_t)(H / 2 + 64) * w * 4)); CK(hipMemset(hb, 3, (size_t)H * rw * 4));
    const double mb_in = ((double)H / 2 * w * 2 + (double)H * rw) * 4 / 1e6, mb_out = (double)H * ow * 4 / 1e6;
    printf("unique bytes: in %.0f MB, out %.0f MB\n", mb_in, mb_out);
    for (int seg : {64, 128, 256, 512}) {
        float t;
        t = run<64, true, false>(va, vb, hb, o, w, rw, ow, H, seg);   printf("seg %3d px  P= 64 loads only  %7.1f us  %.2f TB/s\n", seg, t, mb_in / t);
        t = run<128, true, false>(va, vb, hb, o, w, rw, ow, H, seg);  printf("seg %3d px  P=128 loads only  %7.1f us  %.2f TB/s\n", seg, t, mb_in / t);
        t = run<256, true, false>(va, vb, hb, o, w, rw, ow, H, seg);  printf("seg %3d px  P=256 loads only  %7.1f us  %.2f TB/s\n", seg, t, mb_in / t);
        t = run<64, false, true>(va, vb, hb, o, w, rw, ow, H, seg);   printf("seg %3d px  P= 64 stores only %7.1f us  %.2f TB/s\n", seg, t, mb_out / t);
        t = run<128, false, true>(va, vb, hb, o, w, rw, ow, H, seg);  printf("seg %3d px  P=128 stores only %7.1f us  %.2f TB/s\n", seg, t, mb_out / t);
        t = run<64, true, true>(va, vb, hb, o, w, rw, ow, H, seg);    printf("seg %3d px  P= 64 both        %7.1f us  %.2f TB/s\n", seg, t, (mb_in + mb_out) / t);
        t = run<128, true, true>(va, vb, hb, o, w, rw, ow, H, seg);   printf("seg %3d px  P=128 both        %7.1f us  %.2f TB/s\n", seg, t, (mb_in + mb_out) / t);
        t = run<256, true, true>(va, vb, hb, o, w, rw, ow, H, seg);   printf("seg %3d px  P=256 both        %7.1f us  %.2f TB/s\n", seg, t, (mb_in + mb_out) / t);
    }
    printf("\nV pieces start one sample late (the fused kernel's skew), consecutive-row form\n");
    for (int seg : {128, 256}) {
        float t;
        t = run<64, true, false, 1>(va, vb, hb, o, w, rw, ow, H, seg);  printf("seg %3d px  P= 64 vshift loads %6.1f us", seg, t);
        t = run<64, true, true, 1>(va, vb, hb, o, w, rw, ow, H, seg);   printf("  both %6.1f us  %.2f TB/s\n", t, (mb_in + mb_out) / t);
        t = run<128, true, false, 1>(va, vb, hb, o, w, rw, ow, H, seg); printf("seg %3d px  P=128 vshift loads %6.1f us", seg, t);
        t = run<128, true, true, 1>(va, vb, hb, o, w, rw, ow, H, seg);  printf("  both %6.1f us  %.2f TB/s\n", t, (mb_in + mb_out) / t);
        t = run<256, true, false, 1>(va, vb, hb, o, w, rw, ow, H, seg); printf("seg %3d px  P=256 vshift loads %6.1f us", seg, t);
        t = run<256, true, true, 1>(va, vb, hb, o, w, rw, ow, H, seg);  printf("  both %6.1f us  %.2f TB/s\n", t, (mb_in + mb_out) / t);
    }
    printf("\nthe real kernel's forms (quarter rows; shift = V pieces start one sample late; x4 = 16-byte H loads and stores; buf = buffer descriptors)\n");
    for (int seg : {64, 128}) {
        float t;
#define RUN(SH, X, B) \
        t = run2<SH, X, B, true, false>(va, vb, hb, o, w, rw, ow, H, seg); printf("seg %3d shift %d x4 %d buf %d  loads %6.1f us", seg, SH, X, B, t); \
        t = run2<SH, X, B, false, true>(va, vb, hb, o, w, rw, ow, H, seg); printf("  stores %6.1f us", t); \
        t = run2<SH, X, B, true, true>(va, vb, hb, o, w, rw, ow, H, seg); printf("  both %6.1f us  %.2f TB/s\n", t, (mb_in + mb_out) / t);
        RUN(false, false, false) RUN(true, false, false) RUN(false, true, false) RUN(true, true, false) RUN(true, true, true)
    }
    printf("\ncandidate: V inputs as 16-byte loads of consecutive rows, sector-aligned; sshift = output pieces start 8 bytes before a line\n");
    for (int seg : {64, 128, 256}) {
        float t;
        t = run3<false, true, false>(va, vb, hb, o, w, rw, ow, H, seg); printf("seg %3d  loads %6.1f us", seg, t);
        t = run3<false, false, true>(va, vb, hb, o, w, rw, ow, H, seg); printf("  stores %6.1f us", t);
        t = run3<true, false, true>(va, vb, hb, o, w, rw, ow, H, seg); printf("  stores shifted %6.1f us", t);
        t = run3<false, true, true>(va, vb, hb, o, w, rw, ow, H, seg); printf("  both %6.1f us  %.2f TB/s", t, (mb_in + mb_out) / t);
        t = run3<true, true, true>(va, vb, hb, o, w, rw, ow, H, seg); printf("  both, stores shifted %6.1f us  %.2f TB/s\n", t, (mb_in + mb_out) / t);
    }
    return 0;
}
