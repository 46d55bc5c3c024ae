// How the memory system prices a random 64-byte window (the reference window of a record of an unsorted batch over a 3 Gb
// genome: 1.5 GB of 4-bit codes) by where it lies in its 128-byte line: anywhere (it straddles two lines 47 % of the time),
// or never straddling (as a second copy of the reference, 64 bytes out of phase, would allow).  8 lanes x 8 bytes per window,
// eight windows per wavefront and round, four rounds in flight.
// Build: hipcc --offload-arch=gfx950 -O3 tools/experiments/randwin.hip -o tools/bin/randwin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;
__device__ __forceinline__ u32 hash(u32 x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int MODE, int WBYTES>
__global__ __launch_bounds__(512) void k(const unsigned char *buf, u64 bytes, u32 rounds, u64 *out) {
    const u32 lane = threadIdx.x & 63, gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u32 win = lane >> 3, part = lane & 7;
    u64 acc = 0;
    const u64 lines = bytes / 128 - 2;
    for (u32 r = 0; r < rounds; r += 4) {
        uint2 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u32 h = hash((gw * rounds + r + q) * 8u + win), h2 = hash(h ^ 0x9e3779b9u);
            u64 off = (u64)(h % (u32)lines) * 128;
            if (MODE == 0) off += (h2 % 128u) & ~3u;              // anywhere (dword-aligned)
            else if (MODE == 1) off += (h2 % (u32)(128 - WBYTES + 4)) & ~3u;    // inside one line
            else {
                // two copies, the second 2 GiB + 64 bytes behind the first: anywhere, read from the copy in which it lies in one line
                const u32 o = (h2 % 128u) & ~3u;
                off += o;
                if (o + WBYTES > 128u) off += (1ull << 31) + 64;
                if (MODE == 3) { const u32 o32 = (u32)off; if (part * 8 < WBYTES) v[q] = *(const uint2 *)(buf + (o32 + part * 8)); else v[q] = make_uint2(0, 0); continue; }
            }
            if (part * 8 < WBYTES) v[q] = *(const uint2 *)(buf + off + part * 8); else v[q] = make_uint2(0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) acc += v[q].x + v[q].y;
    }
    if (acc == 0x1234567u) out[0] = acc;
}
int main(int argc, char **argv) {
    const u64 bytes = 1500ull << 20;
    unsigned char *buf; u64 *out;
    hipMalloc(&buf, bytes + (1ull << 31) + 4096); hipMalloc(&out, 8);
    hipMemset(buf, 1, bytes);
    hipMemset(buf + (1ull << 31), 1, bytes + 4096);
    const u32 rounds = 256, blocks = 512 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char *name, int wbytes) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, buf, bytes, rounds, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, buf, bytes, rounds, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double wins = (double)blocks * 8 * rounds * 8;
        printf("%-44s %.3f ms  %.2f G windows/s  %.2f TB/s useful (%d B)\n", name, ms, wins / ms / 1e6, wins * wbytes / ms / 1e9, wbytes);
    };
    run(k<0, 64>, "64-byte windows anywhere", 64);
    run(k<1, 64>, "64-byte windows inside one 128-byte line", 64);
    run(k<2, 64>, "64-byte windows, two copies (64-bit address)", 64);
    run(k<3, 64>, "64-byte windows, two copies (32-bit offset)", 64);
    run(k<0, 32>, "32-byte windows anywhere (a 2-bit reference)", 32);
    run(k<1, 32>, "32-byte windows inside one line", 32);
    return 0;
}
