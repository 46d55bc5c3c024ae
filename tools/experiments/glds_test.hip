// LDS-DMA semantics on gfx950 as the packed kernels use them (phase-1 columns prefetched into the LDS): the saddr + 32-bit
// voffset form of global_load_lds_dword / _ushort, M0 = wave-uniform LDS byte address, lane i lands at M0 + 4 i; a ushort is
// zero-extended to a dword.  Build: hipcc --offload-arch=gfx950 -O3 tools/experiments/glds_test.hip -o tools/bin/glds_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
typedef unsigned short u16;
extern __shared__ u32 lds[];
__device__ __forceinline__ void glds_b32(const void *sbase, u32 voff, u32 lds_dst) {
    u32 keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds_u16(const void *sbase, u32 voff, u32 lds_dst) {
    u32 keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_ushort %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__global__ void k(const u32 *a, const u16 *b, u32 *out, int n) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0xDEADBEEFu;
    __syncthreads();
    const u32 base = (u32)__builtin_amdgcn_readfirstlane(40000 + wave * 1024);    // (beyond 64 KiB of the dynamic LDS: M0 holds more than 16 bits?)
    typedef __attribute__((address_space(3))) u32 lds_u1;
    const u32 idx = (u32)(blockIdx.x * blockDim.x + threadIdx.x) % (u32)n;
    glds_b32(a, idx * 4u, base);
    glds_u16(b, idx * 2u, base + 256u);
    // only even lanes: where does lane 2 j land?
    if (!(lane & 1)) glds_b32(a, idx * 4u, base + 512u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const u32 v0 = *(lds_u1 *)(base + 4u * lane), v1 = *(lds_u1 *)(base + 256u + 4u * lane), v2 = *(lds_u1 *)(base + 512u + 4u * lane);
    u32 *o = out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 3;
    o[0] = v0; o[1] = v1; o[2] = v2;
}
int main() {
    const int n = 1 << 20, threads = 256, blocks = 8;
    std::vector<u32> ha(n); std::vector<u16> hb(n);
    for (int i = 0; i < n; i++) { ha[i] = 0x10000000u + i; hb[i] = (u16)(i * 7 + 0x8001); }
    u32 *a, *out; u16 *b;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 2); hipMalloc(&out, blocks * threads * 12);
    hipMemcpy(a, ha.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), n * 2, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 * 2);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 65536 * 2, 0, a, b, out, n);
    std::vector<u32> ho(blocks * threads * 3);
    hipError_t e = hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    printf("launch: %s\n", hipGetErrorString(e));
    int bad0 = 0, bad1 = 0, bad2 = 0;
    for (int t = 0; t < blocks * threads; t++) {
        if (ho[3 * t] != ha[t]) bad0++;
        if (ho[3 * t + 1] != (u32)hb[t]) bad1++;
        const u32 want2 = (t & 1) ? 0xDEADBEEFu : ha[t];
        if (ho[3 * t + 2] != want2) bad2++;
    }
    printf("dword: %d bad; ushort zero-extended: %d bad; exec-masked lanes keep their own slot: %d bad\n", bad0, bad1, bad2);
    for (int t = 0; t < 6; t++) printf("  t%d: %08x %08x %08x (want %08x %08x)\n", t, ho[3 * t], ho[3 * t + 1], ho[3 * t + 2], ha[t], (u32)hb[t]);
    return bad0 || bad1 || bad2;
}
