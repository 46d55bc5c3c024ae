// Microbenchmark: cost of a wavefront-level global load on gfx950 as a function of width, alignment
// and address pattern (the tabulation kernel's gathers are unaligned 4/8/16-byte-per-lane windows of
// ~80 bytes at 6 or so distinct places).  24 wavefronts per CU, data L2-resident (8 MB buffer), N
// independent loads per wavefront with 8 in flight; reports ns per wavefront-load per CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_loads.hip -o /tmp/ubench_loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 __attribute__((aligned(1))) u32_u;
typedef u32x2 __attribute__((aligned(1))) u32x2_u;
typedef u32x4 __attribute__((aligned(1))) u32x4_u;
struct __attribute__((packed, aligned(4))) u32x3_a { u32 x, y, z; };

// pattern: 0 = contiguous lanes (lane * W), 1 = segments of `seglanes` lanes at pseudo-random bases
template <typename T>
__global__ __launch_bounds__(768) void k(const uint8_t *buf, u32 mask, int iters, int misalign, int pattern,
                                         int seglanes, u32 *out) {
    const int lane = threadIdx.x & 63;
    const u32 gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    u32 acc = 0;
    u32 h = gw * 2654435761u + 12345u;
    const int W = sizeof(T) == 12 ? 8 : sizeof(T);   // the 12-byte type reads 8-byte-strided windows + 4
    for (int i = 0; i < iters; i += 8) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            h = h * 1664525u + 1013904223u;
            u32 off;
            if (pattern == 0) {
                off = ((h >> 4) & mask & ~1023u) + lane * W + misalign;
            } else if (pattern == 2) {
                // tabulation-like reference gather: R records at random bases, two 80-byte windows per
                // record that overlap by 40 bytes (left lanes then right lanes), 8 bytes per lane
                const int seg = lane / 10, l = lane - seg * 10, rec = seg >> 1, side = seg & 1;
                u32 hs = (h ^ (rec * 0x9E3779B9u)) * 2246822519u;
                off = ((hs >> 4) & mask & ~63u) + (hs & 63u) * (misalign ? 1 : 0) + side * 40 + l * W;
            } else if (pattern == 3) {
                // tabulation-like SEQ gather: the same windows over consecutive 100-byte records
                const int seg = lane / 10, l = lane - seg * 10, rec = seg >> 1, side = seg & 1;
                off = ((h >> 4) & mask & ~1023u) + misalign + rec * 100 + side * 40 + l * W;
            } else if (pattern == 5 || pattern == 6) {
                // the packed kernel's step: six records x two windows of five lanes (sixteen 4-bit bases = 8 bytes per lane,
                // read as 12), the right window 15 bytes behind the left (100-base reads, --length 70 --around 10);
                // 5 = reference-like (records at random bases), 6 = SEQ-like (consecutive 50-byte records)
                const int seg = lane / 5, l = lane - seg * 5, rec = seg >> 1, side = seg & 1;
                if (seg >= 12) { off = 0; }
                else if (pattern == 5) {
                    u32 hs = (h ^ (rec * 0x9E3779B9u)) * 2246822519u;
                    off = ((hs >> 4) & mask & ~63u) + ((hs & 63u) & ~3u) + side * 12 + l * 8;
                } else off = ((h >> 4) & mask & ~1023u) + rec * 52 + side * 12 + l * 8;
            } else if (pattern == 7 || pattern == 8) {
                // the union of a record's two windows as one stretch of four lanes x 16 bytes, sixteen records per load
                const int rec = lane >> 2, l = lane & 3;
                if (pattern == 7) {
                    u32 hs = (h ^ (rec * 0x9E3779B9u)) * 2246822519u;
                    off = ((hs >> 4) & mask & ~63u) + ((hs & 63u) & ~3u) + l * 16;
                } else off = ((h >> 4) & mask & ~1023u) + rec * 52 + l * 16;
            } else if (pattern == 4) {
                // union windows: one contiguous run of `seglanes` lanes per record at a random base
                const int seg = lane / seglanes, l = lane - seg * seglanes;
                u32 hs = (h ^ (seg * 0x9E3779B9u)) * 2246822519u;
                off = ((hs >> 4) & mask & ~63u) + (hs & 63u) * (misalign ? 1 : 0) + l * W;
            } else {
                // each segment of `seglanes` lanes reads a contiguous window at its own random base
                const int seg = lane / seglanes, l = lane - seg * seglanes;
                u32 hs = (h ^ (seg * 0x9E3779B9u)) * 2246822519u;
                off = ((hs >> 4) & mask & ~63u) + l * W + misalign + (hs & 3u) * (misalign ? 1 : 0);
            }
            v[u] = *(const T *)(buf + off);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if constexpr (sizeof(T) == 4) acc += v[u];
            else if constexpr (sizeof(T) == 8) acc += v[u].x ^ v[u].y;
            else if constexpr (sizeof(T) == 12) acc += v[u].x ^ v[u].y ^ v[u].z;
            else acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x12345678u) out[gw] = acc;
}

template <typename T>
static double run(const uint8_t *buf, u32 mask, int misalign, int pattern, int seglanes, u32 *out) {
    const int iters = 4096, grid = 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<T>, dim3(grid), dim3(768), 0, 0, buf, mask, 64, misalign, pattern, seglanes, out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<T>, dim3(grid), dim3(768), 0, 0, buf, mask, iters, misalign, pattern, seglanes, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-loads per CU: grid * 12 * iters / 256
    const double loads_per_cu = (double)grid * 12 * iters / 256.0;
    return ms * 1e6 / loads_per_cu;  // ns per wave-load per CU
}

// unaligned LDS reads: correctness and cost of ds_read_b64 at byte offsets
__global__ __launch_bounds__(768) void lds_k(int off, int iters, u32 *out, int check) {
    extern __shared__ __attribute__((aligned(16))) uint8_t l[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) l[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 acc = 0;
    const uint8_t *base = l + wave * 1024 + off;
    u32 o = lane * 8;
    for (int i = 0; i < iters; i++) {
        asm volatile("" : "+v"(o));   // keep the read in the loop
        const u32x2 v = *(const u32x2_u *)(base + o);
        acc += v.x ^ (v.y * 3u);
    }
    if (check) {
        const u32x2 v = *(const u32x2_u *)(base + lane * 8);
        u32 ok = 1;
        for (int b = 0; b < 8; b++) {
            const u32 want = (uint8_t)((wave * 1024 + off + lane * 8 + b) * 7 + 3);
            const u32 got = ((b < 4 ? v.x : v.y) >> (8 * (b & 3))) & 0xFF;
            if (want != got) ok = 0;
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = ok;
    } else if (acc == 0x12345678u) out[0] = acc;
}
static void lds_test(u32 *out) {
    for (int off = 0; off < 8; off++) {
        hipLaunchKernelGGL(lds_k, dim3(1), dim3(768), 16384, 0, off, 1, out, 1);
        std::vector<u32> h(768);
        hipMemcpy(h.data(), out, 768 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (u32 v : h) bad += v != 1;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(lds_k, dim3(512), dim3(768), 16384, 0, off, 20000, out, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("lds_read_b64 byte offset %d: %s, %.2f ns per wave-read per CU\n", off, bad ? "WRONG" : "ok",
               ms * 1e6 / (512.0 * 12 * 20000 / 256));
    }
}

int main(int argc, char **argv) {
    if (argc > 1) { u32 *o2; hipMalloc(&o2, 1 << 20); lds_test(o2); return 0; }
    const size_t n = 8u << 20;
    uint8_t *buf; u32 *out;
    hipMalloc(&buf, n + 4096); hipMalloc(&out, 1 << 20);
    hipMemset(buf, 1, n + 4096);
    const u32 mask = (u32)(n - 1);
    printf("pattern width misalign seglanes ns_per_waveload_per_CU bytes_per_ns_per_CU\n");
    for (int pattern = 0; pattern < 2; pattern++)
        for (int mis = 0; mis < 4; mis += (mis == 0 ? 1 : 2)) {
            for (int seg : {64, 20, 10, 5}) {
                if (pattern == 0 && seg != 64) continue;
                if (pattern == 1 && seg == 64) continue;
                double t4 = run<u32_u>(buf, mask, mis, pattern, seg, out);
                double t8 = run<u32x2_u>(buf, mask, mis, pattern, seg, out);
                double t16 = run<u32x4_u>(buf, mask, mis, pattern, seg, out);
                printf("%d 4 %d %d %.2f %.1f\n", pattern, mis, seg, t4, 256.0 / t4);
                printf("%d 8 %d %d %.2f %.1f\n", pattern, mis, seg, t8, 512.0 / t8);
                printf("%d 16 %d %d %.2f %.1f\n", pattern, mis, seg, t16, 1024.0 / t16);
            }
        }
    printf("# pattern 2 (ref-like, 3 records x 2 overlapping windows), 3 (seq-like), 4 (union window per record)\n");
    for (int mis = 0; mis < 2; mis++) {
        printf("2 8 %d 10 %.2f\n", mis, run<u32x2_u>(buf, mask, mis, 2, 10, out));
        printf("3 8 %d 10 %.2f\n", mis, run<u32x2_u>(buf, mask, mis, 3, 10, out));
        printf("4 8 %d 15 %.2f\n", mis, run<u32x2_u>(buf, mask, mis, 4, 15, out));
        printf("4 16 %d 8 %.2f\n", mis, run<u32x4_u>(buf, mask, mis, 4, 8, out));
        printf("4 16 %d 16 %.2f\n", mis, run<u32x4_u>(buf, mask, mis, 4, 16, out));
        printf("4 4 %d 32 %.2f\n", mis, run<u32_u>(buf, mask, mis, 4, 32, out));
    }
    printf("# the packed kernel's step (6 records x 2 windows x 5 lanes, 12-byte loads): reference-like %.2f  SEQ-like %.2f ns per wave-load and CU (6 records each)\n",
           run<u32x3_a>(buf, mask, 0, 5, 5, out), run<u32x3_a>(buf, mask, 0, 6, 5, out));
    printf("# union windows (16 records x 4 lanes x 16 bytes): reference-like %.2f  SEQ-like %.2f ns per wave-load and CU (16 records each)\n",
           run<u32x4_u>(buf, mask, 0, 7, 4, out), run<u32x4_u>(buf, mask, 0, 8, 4, out));
    printf("# alignment sweep, 8 bytes per lane: pattern 2 (ref-like) and 3 (seq-like) at byte phase 0..4\n");
    for (int mis = 0; mis <= 4; mis++)
        printf("phase %d: seq-like %.2f\n", mis, run<u32x2_u>(buf, mask, mis, 3, 10, out));
    printf("# dword-aligned 12-byte loads at 8-byte lane stride (aligned-down window + one dword)\n");
    printf("x3 aligned: seq-like %.2f  ref-like %.2f\n", run<u32x3_a>(buf, mask, 0, 3, 10, out), run<u32x3_a>(buf, mask, 0, 2, 10, out));
    printf("x2 aligned: seq-like %.2f  ref-like %.2f\n", run<u32x2_u>(buf, mask, 0, 3, 10, out), run<u32x2_u>(buf, mask, 0, 2, 10, out));
    printf("x2 unaligned: seq-like %.2f  ref-like %.2f\n", run<u32x2_u>(buf, mask, 1, 3, 10, out), run<u32x2_u>(buf, mask, 1, 2, 10, out));
    return 0;
}
