// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 for this kernel's access shapes
// (VERDICT r1 item 4).  Each kernel streams a buffer of a known size, well beyond the 256 MiB Infinity Cache, exactly
// once; run under `rocprofv3 --pmc FETCH_SIZE` (and, separately, `--pmc WRITE_SIZE`) and compare the counter with
// the bytes printed here.
//   read16   16 bytes per lane, contiguous (global_load_dwordx4): the guide's reference pattern (counter = 1/2)
//   read12   12 bytes per lane at an 8-byte lane stride, dword-aligned (global_load_dwordx3): the window loads of
//            tabulate_kernel (every lane re-reads the last dword of its neighbour's window)
//   read12r  the same 12-byte loads, records of 100 bytes at pseudo-random order (a gather like the SEQ windows)
//   write16  16 bytes per lane, contiguous stores
// Build: hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o /tmp/calib_fetch ; usage: calib_fetch [GiB]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) u32x3 { u32 x, y, z; };

__global__ void read16(const u32x4 *__restrict__ in, size_t n16, u32 *out) {
    u32 acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read12(const uint8_t *__restrict__ in, size_t n8, u32 *out) {
    u32 acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const u32x3 v = *(const u32x3 *)(in + i * 8);
        acc ^= v.x ^ v.y ^ v.z;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// records of 128 bytes visited in a scrambled order, ten 12-byte loads at an 8-byte stride per record (bytes 0..84)
__global__ void read12r(const uint8_t *__restrict__ in, size_t nrec, u32 *out) {
    u32 acc = 0;
    const size_t lanes = (size_t)gridDim.x * blockDim.x;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < nrec * 10; t += lanes) {
        const size_t rec = t / 10, l = t - rec * 10;
        const size_t scr = (rec * 2654435761ull) % nrec;       // (nrec odd: a permutation)
        const u32x3 v = *(const u32x3 *)(in + scr * 128 + l * 8);
        acc ^= v.x ^ v.y ^ v.z;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void write16(u32x4 *__restrict__ outp, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        u32x4 v = {(u32)i, 1u, 2u, 3u};
        outp[i] = v;
    }
}

int main(int argc, char **argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 2.0;
    const size_t bytes = (size_t)(gib * (1ull << 30)) / 4096 * 4096;
    uint8_t *buf;
    u32 *out;
    if (hipMalloc((void **)&buf, bytes + 4096) != hipSuccess || hipMalloc((void **)&out, 64) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes + 4096);
    hipDeviceSynchronize();
    const int grid = 256 * 8, block = 256;
    size_t nrec = bytes / 128;
    if (nrec % 2 == 0) nrec--;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(read16, dim3(grid), dim3(block), 0, 0, (const u32x4 *)buf, bytes / 16, out);
        hipLaunchKernelGGL(read12, dim3(grid), dim3(block), 0, 0, buf, bytes / 8, out);
        hipLaunchKernelGGL(read12r, dim3(grid), dim3(block), 0, 0, buf, nrec, out);
        hipLaunchKernelGGL(write16, dim3(grid), dim3(block), 0, 0, (u32x4 *)buf, bytes / 16);
        hipDeviceSynchronize();
    }
    printf("{\"read16_bytes\": %zu, \"read12_bytes\": %zu, \"read12r_unique_bytes\": %zu, \"read12r_lines128_bytes\": %zu, \"write16_bytes\": %zu}\n",
           bytes, bytes + 4, nrec * 84, nrec * 128, bytes);
    return 0;
}
