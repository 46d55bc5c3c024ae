// gfx950 (MI355X / CDNA4) kernels of the damage-tabulation engine.
//
// Work decomposition (integer/byte histogram work — HBM/LDS bound, no MFMA):
//   * a wavefront (64 lanes) takes tiles of 63 consecutive records, handed out on demand within the two blocks of a CU;
//   * phase 1 of the tile loop, lane per record: coalesced SoA loads of the per-record columns, flag filter
//     (reader.py:121-132), and everything a record needs whose CIGAR is one match operation, alone or between soft
//     clips ([S] M [S]: nine records in ten of an aDNA library): soft-clip update (statistics.py:37-51),
//     fragment-length update (statistics.py:117-126), the 16-byte staging entry of the record;
//   * every other kept record — indels, N / P operations, hard clips, contig edges, anything odd or wrong — is handed,
//     index and columns, to the *general pass*: 64 such records at a time, lane per record, the full CIGAR scan
//     (pysam's query_alignment_start / _end, htslib's bam_endpos, align.parse_cigar), error checks and
//     classification into complete / partial / single-insertion / single-deletion entries (appended to the
//     wavefront's lists in global memory) and the few that walk their CIGAR column by column;
//   * phase 2, the steps: one lane owns eight consecutive bytes of a record's window (flank and columns merged), so
//     a wavefront step counts R = 64 / G records at once (3 at --length 70 --around 10); the entry of a record is
//     read back per slot.  The loads of step k+4 are issued before step k is counted (PIPE_DEPTH = 4 register sets; 3 with --min-basequal)
//     so that the gather latency of the resident genome is hidden.  The complete records of a tile are counted at
//     once; the entries of the lists behind the tile loop, in dense runs of one kind each;
//   * the common outcome (read base == reference base, or an A/C/G/T flank base) is one
//     conflict-free ds_add_u32 per byte into a (lane, byte)-indexed LDS table; everything else
//     (substitutions, N, masked columns) is queued and counted 64 events at a time into the MIS/CMP
//     tables;
//   * at block end the LDS image is stored to a per-block slot and a second kernel sums the
//     slots into the u64 accumulators (no global atomics on the hot path).
//   * tabulate_kernel<.., PK> — the packed kernels, what the benchmark and the command line run: the SEQ column and the
//     reference as 4-bit one-hot codes, sixteen bases per lane, plain matches in bit-sliced register counters (see the
//     template's comments).  Round 6: ONE block of 1024 threads per CU (MdxPkConfig); the phase-1 columns and the second
//     round trip of a wavefront's NEXT tile prefetched into an area of the LDS by LDS-DMA loads (pfl_*: asm statements the
//     compiler does not count — and what follows from that for the waits); kernel arguments through the constant address
//     space (karg_p); phase 1 makes the entries of single-indel records itself (SIP); a large genome's reference twice, half
//     a 128-byte line out of phase (MdxTabArgs::ref2);
//   * tabulate_kernel<.., RS> is the same kernel with the quality rescaling of mapdamage/rescale.py fused in for the
//     records of its own tile loop (mdx_tabulate_rescale_device, BASELINE configs[4]): routing and the quality copy in
//     phase 1, the rescaled columns — they are mismatches, hence events — through drain_all, one 1024-thread block per
//     CU (see MdxFuse in mdx_internal.h and the RS blocks below).
// Counting is done in *reference orientation* (left-/right-anchored, no complementing); the
// strand step of main.py:200-205 (reverse-complement + flank swap) becomes a fixed permutation
// applied once by finalize_kernel.  The oracle (oracle/mdx_oracle.c) builds and reverses the
// strings literally instead, so the two share no derivation.
#include "mdx_internal.h"

#include <type_traits>

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "gfx950 only: the s_waitcnt immediates (0xC07F = lgkmcnt(0), 0x0F70 = vmcnt(0)) and the inline assembly below are gfx9 encodings"
#endif

typedef uint8_t u8;
typedef int8_t i8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 __attribute__((aligned(1))) u32x2_u;
typedef u32x2 __attribute__((aligned(4))) u32x2_a4;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_u;
typedef u32 __attribute__((aligned(1))) u32_u;
typedef u16 __attribute__((aligned(1))) u16_u;
struct __attribute__((packed, aligned(4))) u32x3 { u32 x, y, z; };   // global_load_dwordx3
typedef u32 u32v3 __attribute__((ext_vector_type(3)));
typedef u32v3 __attribute__((aligned(4))) u32v3_u;
// MDX_NT=1: the operands streamed once (SEQ, qualities, the per-record columns) loaded non-temporal, so that their lines
// are the first to leave the XCD's L2 in favour of the reference (two random windows per record: 58 % of what crosses
// the fabric for the survey's genome).  Measured (r03c): L2 misses -2 %, kernel time unchanged on the 10 Mb genome and
// 7 % worse on a 3 Gb one (the loads skip the vector L1) — off.
#ifndef MDX_NT
#define MDX_NT 0
#endif

// The quality copy of the fused kernel (16-byte units at 16-byte-aligned addresses): the copy is written once and not
// read again by this kernel: MDX_CP_NT & 1 stores it non-temporal, so that it does not take the reference's place in the
// L2, MDX_CP_NT & 2 loads the source likewise — measured: 3.43 ms either way against 3.44 (off).
#ifndef MDX_CP_NT
#define MDX_CP_NT 0
#endif
typedef u32x4 __attribute__((aligned(16))) u32x4_a;
__device__ __forceinline__ void cp_store(u8 *p, const u32x4 v) {
#if MDX_CP_NT & 1
    __builtin_nontemporal_store(v, (u32x4_a *)p);
#else
    *(u32x4_a *)p = v;
#endif
}
__device__ __forceinline__ u32x4 cp_load(const u8 *p) {
#if MDX_CP_NT & 2
    return __builtin_nontemporal_load((const u32x4_a *)p);
#else
    return *(const u32x4_a *)p;
#endif
}
__device__ __forceinline__ u32x3 ld12_stream(const u8 *p) {
    u32x3 r;
#if MDX_NT
    const u32v3 t = __builtin_nontemporal_load((const u32v3_u *)p);
    r.x = t.x; r.y = t.y; r.z = t.z;
#else
    r = *(const u32x3 *)p;
#endif
    return r;
}
typedef unsigned long long u64;
typedef u64 __attribute__((aligned(1))) u64_u;
typedef long long i64;

#ifndef MDX_BLOCK
#define MDX_BLOCK 768                   // 12 wavefronts; two blocks per CU share the 160 KiB LDS
#endif
#ifndef MDX_WPS
#define MDX_WPS 6                       // wavefronts per SIMD the register budget is sized for
#endif
#ifndef PIPE_DEPTH
#define PIPE_DEPTH 4                    // wavefront steps in flight of the complete runs (register sets of the load pipeline)
#endif
#ifndef MDX_ENT_AHEAD
#define MDX_ENT_AHEAD 0                 // staging entries read one fill ahead
#endif
#ifndef MDX_PK_ENT_AHEAD
#define MDX_PK_ENT_AHEAD 0              // ... in the packed kernels (measured: +-1 % on every workload — the other wavefronts cover that wait)
#endif
#ifndef MDX_PD_G
#define MDX_PD_G 2                      // steps in flight of the single-indel runs
#endif
#ifndef MDX_PD_P
#define MDX_PD_P 3                      // ... and of the runs of partial entries
#endif
#ifndef MDX_QPREFETCH
#define MDX_QPREFETCH 1                 // MASK: the quality windows requested with the other two, PIPE_DEPTH steps ahead
#endif
#ifndef MDX_PK_FASTP
#define MDX_PK_FASTP 1                  // the partial steps of the plain packed kernels: see FIDP
#endif
#ifndef MDX_PK_SIP
#define MDX_PK_SIP 1                    // the packed kernels' phase 1 makes the entries of single-indel records itself (see SIP)
#endif
#ifndef MDX_PK_FASTIDX
#define MDX_PK_FASTIDX 1                // the packed kernels' complete steps: the staging entry's LDS address as add + min (a slot past its strand's
#endif                                  // last entry reads that last entry and is masked out), the event's place from two shift-adds
#ifndef MDX_PK_PD
#define MDX_PK_PD 4                     // ... and of the packed kernel's complete runs
#endif
#ifndef MDX_PK_PD_P
#define MDX_PK_PD_P 4                   // ... and of its runs of partial entries (four: its groups are added in pairs)
#endif
#ifndef MDX_PK_PTILE
#define MDX_PK_PTILE 1                  // the packed kernels count a tile's partial records in the tile loop (0: through the wavefront's list behind it; the fused one always does)
#endif
#ifndef MDX_PKF_PTILE
#define MDX_PKF_PTILE 0                 // ... and the packed fused kernel: no — through its list, rescaled by the list's passes (the run in the
                                        // tile loop cost the kernel's registers 3 % on config 5, which has no such record)
#endif
#ifndef MDX_PKM_PD
#define MDX_PKM_PD 3                    // ... and with --min-basequal
#endif
#ifndef MDX_PK_PD_ML
#define MDX_PK_PD_ML 4                  // ... and of the launches over several libraries (a pool's library, its records' place
#endif                                  // in the batch ordered by library and the pools that share its tiles want registers too)
#ifndef MDX_PKM_PD_ML
#define MDX_PKM_PD_ML 3                 // ... with --min-basequal (rounds and a second set of planes; 16 M records over 8 libraries at -Q 20, x the
                                        // one-library masked launch: one step in flight 1.20, two 1.07, three — 2 registers spilled — 1.02)
#endif
#ifndef MDX_PD_P_ML
#define MDX_PD_P_ML 2                   // ... their runs of partial entries (three: 4 registers spilled; two: none)
#endif
#ifndef MDX_PD_G_M
#define MDX_PD_G_M 2                    // ... of the masked one-library kernel and the fused one (their registers)
#endif
#ifndef MDX_PD_G_ML
#define MDX_PD_G_ML 1                   // ... and of single-indel entries
#endif
#ifndef MDX_PREFIX
#define MDX_PREFIX 1                    // plain prefixes of gapped records through the fast step
#endif
#define EVQ_CAP 64                      // rare-event queue capacity per wavefront
#define EVQ_BYTES (EVQ_CAP * 20)        // per wavefront: S[64] u32x2 | R[64] u32x2 | W[64] u32
// The fused kernel (one 1024-thread block per CU: the LDS has the room) holds the events of a whole run, as a rule: it never
// drains inside one — the drain of a fused record's event may touch global memory (rs_event), and with a store or an atomic
// possibly pending every wait of the pipelined loop is a vmcnt(0) — but stops the run, drains, and takes it up again
// (MdxFuse::qcap events per wavefront: as many as the image has room for, mdx_k_fuse_qcap)
#define MDX_PK_EVQ_BYTES (MDX_PK_QCAP * 20)   // the packed kernel's: {read 8 B, reference 8 B}[QCAP] | W[QCAP]
#define MDX_PK_TAB_BYTES 1024                 // ... and per block: the lanes' read-column masks (64 x 8 B), the symbol-pair table (256 x 2 B)
#define COL_S 24
#define ERR_BAD_READ 6
// symbol classes on the device: 0..3 = A,C,T,G ((ascii >> 1) & 3), 4 = '-', 5 = anything else
#define SYM_GAP 4
#define SYM_OTHER 5
// resident reference bytes: upper-case ASCII for A,C,G,T; negative (as int8) for the rest, so
// that a zero-extended read byte can never compare equal to a sign-extended invalid one
#define REF_GAP 0x84
#define REF_OTHER 0x85

// final column (mapdamage/seq.py:6-30 order, "Total" removed) of substitution ref>read,
// indexed [ref class][read class] with classes A,C,T,G,-; 31 = none
__constant__ u8 c_col[25] = {
    /* A> */ 31, 8, 9, 6, 16,
    /* C> */ 11, 31, 5, 10, 18,
    /* T> */ 13, 7, 31, 12, 17,
    /* G> */ 4, 14, 15, 31, 19,
    /* -> */ 20, 22, 21, 23, 31};
// raw MIS columns whose reference symbol is class k (three substitutions and the deletion)
__constant__ u8 c_refcols[16] = {8, 9, 6, 16, 11, 5, 10, 18, 13, 7, 12, 17, 4, 14, 15, 19};
// final column seen from the reverse strand (complement both symbols), for columns >= 4
__constant__ u8 c_comp_col[25] = {3, 2, 1, 0, 5, 4, 7, 6, 12, 13, 14, 15, 8,
                                  9, 10, 11, 17, 16, 19, 18, 21, 20, 23, 22, 24};

int mdx_k_block_threads() { return MDX_BLOCK; }
// LDS image: [tables w_total words, padded to 16 B][staging, 12 x mdx_stage_entries x 16 B][event queues, 12 x EVQ_BYTES]
int mdx_k_stage_off(const MdxDims &d) { return (int)((d.w_total + 3) / 4 * 4); }
int mdx_k_queue_off(const MdxDims &d) { return mdx_k_stage_off(d) + (MDX_BLOCK / 64) * mdx_stage_entries(d) * 4; }
// ... [byte-mask table: 9 x u64, entry n = the low n bytes set; PK: 17 x u64, entry n = the low n nibbles set]
#define LT_BYTES 136
size_t mdx_k_lds_bytes(const MdxDims &d) { return (size_t)mdx_k_queue_off(d) * 4 + (size_t)(MDX_BLOCK / 64) * EVQ_BYTES + LT_BYTES; }
// The fused tabulate + rescale kernel (tabulate_kernel<.., RS>): one block of 1024 threads per CU — 16 wavefronts with
// 128 registers each instead of 24 with 80 (measured with the plain kernel: +3 % on config 3) — because its image does
// not fit twice: behind the plain image [second TC table, 256-byte aligned][4 words][lookup table][terms]
#ifndef MDX_FUSE_BLOCK
#define MDX_FUSE_BLOCK 1024
#endif
#ifndef MDX_FUSE_WPS
#define MDX_FUSE_WPS 4
#endif
#ifndef MDX_FUSE_CPU
#define MDX_FUSE_CPU 2                  // 16-byte units per lane of the first pass of a tile's quality copy (measured: 2 3.64 ms, 4 3.74, 7 4.18 — the registers)
#endif
#ifndef MDX_PKF_CPU
#define MDX_PKF_CPU 1                   // ... of the packed fused kernel
#endif
#ifndef MDX_FUSE_PD
#define MDX_FUSE_PD 4                   // steps in flight of the complete runs in the fused kernel
#endif
#ifndef MDX_FUSE_CP2
#define MDX_FUSE_CP2 1                  // the passes of the quality copy behind the first: two units per lane and round trip
#endif
#define MDX_FUSE_RSQ 192                // per wavefront: transitions of fused records waiting for their qualities (8 bytes each)
#define MDX_FUSE_MRM 72                 // per wavefront: one 64-bit word per staging entry (the MR terms of its record)
int mdx_k_fuse_block_threads() { return MDX_FUSE_BLOCK; }
// The packed kernel (tabulate_kernel<.., PK>): two blocks of 512 threads per CU — 16 wavefronts with 128 registers each: its
// sixteen-base lanes need half the instructions per record and half the wavefronts to keep the units busy, and its bit-sliced
// counters want the registers (at 80 the hot loop spills)
#ifndef MDX_PK_BLOCK
#define MDX_PK_BLOCK 1024               // the largest block of the packed kernels (their launch bound): see MdxPkConfig
#endif
#ifndef MDX_PK_DEFER
#define MDX_PK_DEFER 1                  // the plain packed kernel adds its groups of four steps in pairs (tabulate_kernel: HS)
#endif
#ifndef MDX_PK_STEAL
#define MDX_PK_STEAL 2                    // pools a wavefront of the packed kernels asks for tiles once its own is empty (0: none)
#endif
#ifndef MDX_PK_DEFER_ML
#define MDX_PK_DEFER_ML 1               // ... the kernel of the launches over several libraries too (round 6: it has the registers)
#endif
#ifndef MDX_PK_WPS
#define MDX_PK_WPS 4
#endif
int mdx_k_pk_blocks_per_cu(int threads) { return MDX_PK_WPS * 256 / threads; }
int mdx_k_pk_queue_off(const MdxDims &d, int threads) { return mdx_k_stage_off(d) + (threads / 64) * mdx_stage_entries(d) * 4; }
// (behind the queues and the two tables, 16-byte aligned)
int mdx_k_pk_pfl_off(const MdxDims &d, int threads) {
    return (int)((((size_t)mdx_k_pk_queue_off(d, threads) * 4 + (size_t)(threads / 64) * MDX_PK_EVQ_BYTES + LT_BYTES + MDX_PK_TAB_BYTES + 15) & ~(size_t)15) / 4);
}
size_t mdx_k_pk_lds_bytes(const MdxDims &d, const MdxPkConfig &k) {
    if (k.pfl) return (size_t)mdx_k_pk_pfl_off(d, k.threads) * 4 + (size_t)(k.threads / 64) * MDX_PFL_WAVE_BYTES;
    return (size_t)mdx_k_pk_queue_off(d, k.threads) * 4 + (size_t)(k.threads / 64) * MDX_PK_EVQ_BYTES + LT_BYTES + MDX_PK_TAB_BYTES;
}
// The largest block whose image fits the LDS: 1024 threads at the defaults (--length 70: 153 KB), 512 or 256 where the tables
// of a long --length leave less room (a block of 256 — 25 KB beside the tables — fits wherever the ASCII kernel's image does).
// (MDX_PK_THREADS=256|512|1024: A/B runs and the tests of the smaller blocks)
MdxPkConfig mdx_k_pk_config(const MdxDims &d, size_t lds_limit) {
    static const int force_threads = [] { const char *e = getenv("MDX_PK_THREADS"); return e && *e ? atoi(e) : 0; }();
    MdxPkConfig k;
    k.threads = MDX_PK_BLOCK; k.pfl = 1;
    if (force_threads == 256 || force_threads == 512 || force_threads == 1024) k.threads = force_threads;
    if (k.threads > MDX_PK_BLOCK) k.threads = MDX_PK_BLOCK;
    while (k.threads > 256 && mdx_k_pk_lds_bytes(d, k) > lds_limit) k.threads /= 2;
    return k;
}
int mdx_k_fuse_queue_off(const MdxDims &d) { return mdx_k_stage_off(d) + (MDX_FUSE_BLOCK / 64) * mdx_stage_entries(d) * 4; }
int mdx_k_fuse_tcb_off(const MdxDims &d, int qcap) {
    const size_t end = (size_t)mdx_k_fuse_queue_off(d) * 4 + (size_t)(MDX_FUSE_BLOCK / 64) * (size_t)qcap * 20 + LT_BYTES;
    return (int)(((end + 255) & ~(size_t)255) / 4);
}
size_t mdx_k_fuse_lds_bytes(const MdxDims &d, int npos, int qcap) {
    return (size_t)mdx_k_fuse_tcb_off(d, qcap) * 4 + (size_t)d.nlib * d.w_tc * 4 + 16 + (size_t)((2 * npos * 94 + 15) & ~15) + (size_t)2 * npos * 8 +
           (size_t)(MDX_FUSE_BLOCK / 64) * (MDX_FUSE_MRM * 8 + MDX_FUSE_RSQ * 8 + 8);
}

// The packed fused kernel (tabulate_kernel<.., RS, PK>): the packed kernel's steps in the fused kernel's frame — one block
// of 1024 threads per CU (16 wavefronts with 128 registers, like the packed kernel's two blocks of 512: the rescale model,
// the MR words and the transition lists do not fit the LDS twice).  No second TC table: the image behind the packed
// kernel's [tables][staging][event queues][mask tables] is, 256-byte aligned, [4 words][lookup table][terms][MR words]
// [transition lists] (MdxFuse::tcb_off = the word offset of the four words).
int mdx_k_pkf_tcb_off(const MdxDims &d) {
    const size_t end = (size_t)mdx_k_fuse_queue_off(d) * 4 + (size_t)(MDX_FUSE_BLOCK / 64) * MDX_PK_EVQ_BYTES + LT_BYTES + MDX_PK_TAB_BYTES;
    return (int)(((end + 255) & ~(size_t)255) / 4);
}
size_t mdx_k_pkf_lds_bytes(const MdxDims &d, int npos) {
    return (size_t)mdx_k_pkf_tcb_off(d) * 4 + 16 + (size_t)((2 * npos * 94 + 15) & ~15) + (size_t)2 * npos * 8 +
           (size_t)(MDX_FUSE_BLOCK / 64) * (MDX_FUSE_MRM * 8 + MDX_FUSE_RSQ * 8 + 8);
}

// read byte -> class; accepted only if it is exactly the upper-case letter
// ("nt in 'ACGT-'", statistics.py:27)
__device__ __forceinline__ int classify_read(u32 ch) {
    const u32 k = (ch >> 1) & 3u;
    const u32 recon = (0x47544341u >> (k * 8)) & 0xFFu;  // 'A','C','T','G'
    return ch == recon ? (int)k : (ch == (u32)'-' ? SYM_GAP : SYM_OTHER);
}
// resident reference byte (sign-extended) -> class
__device__ __forceinline__ int classify_ref(int rch) {
    return rch < 0 ? (rch & 0xF) : ((rch >> 1) & 3);
}

__global__ void encode_ref_kernel(const u8 *__restrict__ in, u8 *__restrict__ out, i64 n) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        u32 ch = in[i];
        if (ch >= 'a' && ch <= 'z') ch -= 32;  // .upper() of main.py:180 / align.py:32-33
        const int c = classify_read(ch);
        out[i] = c < 4 ? (u8)ch : (c == SYM_GAP ? (u8)REF_GAP : (u8)REF_OTHER);
    }
}

void mdx_k_encode_ref(const u8 *d_ascii, u8 *d_codes, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(encode_ref_kernel, dim3(grid), dim3(256), 0, s, d_ascii, d_codes, (i64)n);
}

template <bool USE_LDS>
__device__ __forceinline__ void bump(u32 *lds, u64 *raw, int idx) {
    if (USE_LDS) atomicAdd(&lds[idx], 1u);
    else atomicAdd(&raw[idx], 1ull);
}
// TC increments of the fast path, the eight bytes of a lane in one block.  LDS byte address of byte j =
// base + ((r >> (8 j + 1)) & 3) * 2048 + 256 j (base = TC[library][strand] + 4 * lane): the base class of
// the reference byte selects the 2 KiB plane.  Hand-written (v_bfe_u32, v_lshl_add_u32, ds_add_u32): the
// compiler's own sequence is shift + and + add3 per byte.  The ds_adds are invisible to the compiler's
// lgkmcnt bookkeeping, which only makes its waits more conservative (LDS operations complete in order);
// nothing reads TC before the final barrier, which is preceded by an explicit s_waitcnt.
// One increment for all eight bytes: a byte of a complete record that is not a task has a table word of its
// own that no task maps to, so counting it is harmless; the steps of the other records zero such bytes (plane A)
// and account for them in DMP
__device__ __forceinline__ void tc_bump8_all(u32 r_lo, u32 r_hi, u32 base_bytes, u32 dd) {
    u32 t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile(
        "v_bfe_u32 %0, %8, 1, 2\n\t"
        "v_bfe_u32 %1, %8, 9, 2\n\t"
        "v_bfe_u32 %2, %8, 17, 2\n\t"
        "v_bfe_u32 %3, %8, 25, 2\n\t"
        "v_bfe_u32 %4, %9, 1, 2\n\t"
        "v_bfe_u32 %5, %9, 9, 2\n\t"
        "v_bfe_u32 %6, %9, 17, 2\n\t"
        "v_bfe_u32 %7, %9, 25, 2\n\t"
        "v_lshl_add_u32 %0, %0, 11, %10\n\t"
        "v_lshl_add_u32 %1, %1, 11, %10\n\t"
        "v_lshl_add_u32 %2, %2, 11, %10\n\t"
        "v_lshl_add_u32 %3, %3, 11, %10\n\t"
        "ds_add_u32 %0, %11\n\t"
        "ds_add_u32 %1, %11 offset:256\n\t"
        "v_lshl_add_u32 %4, %4, 11, %10\n\t"
        "v_lshl_add_u32 %5, %5, 11, %10\n\t"
        "ds_add_u32 %2, %11 offset:512\n\t"
        "ds_add_u32 %3, %11 offset:768\n\t"
        "v_lshl_add_u32 %6, %6, 11, %10\n\t"
        "v_lshl_add_u32 %7, %7, 11, %10\n\t"
        "ds_add_u32 %4, %11 offset:1024\n\t"
        "ds_add_u32 %5, %11 offset:1280\n\t"
        "ds_add_u32 %6, %11 offset:1536\n\t"
        "ds_add_u32 %7, %11 offset:1792"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "v"(r_lo), "v"(r_hi), "v"(base_bytes), "v"(dd)
        : "memory");
}
// (pk & 0x3ff00) | lane4: TC base of the record (256-byte aligned) + the lane's word
__device__ __forceinline__ u32 tc_base(u32 pk, u32 lane4) {
    u32 r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(0x3FF00u), "v"(pk), "v"(lane4));
    return r;
}

template <bool USE_LDS>
__device__ __forceinline__ void bump_n(u32 *lds, u64 *raw, int idx, u32 n) {
    if (USE_LDS) atomicAdd(&lds[idx], n);
    else atomicAdd(&raw[idx], (u64)n);
}

__device__ __forceinline__ void flag_error(u64 *err, i64 read, int code) {
    atomicMin(err, ((u64)read << 8) | (u64)code);
}

__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ int mbcnt64(u64 m, int base) {
    return (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, (u32)base));
}
// Rescaling in patch mode (MdxFuse::patch / MdxRescaleArgs::patch): a quality byte that changes (rescale.py:228-246) becomes
// an entry of the launch's list — index of the byte in the column | new Phred << 32 — instead of a store into a copy of the
// column.  The lanes that have one at the same time append together: one atomic for all of them (called under divergence it
// covers the active lanes — the ballot's).
// the kernel arguments where they lie — in the constant address space: what is read through such a pointer comes by a scalar
// load at the point of use (through a generic pointer it would be a vector load and a round trip)
typedef const __attribute__((address_space(4))) MdxTabArgs *karg_p;

// LDS-DMA (global_load_lds_*, gfx950): lane i's element lands in the LDS at the wave-uniform byte address in M0 + 4 i — a
// ushort zero-extended to a dword — and takes no register on the way (tools/experiments/glds_test.hip holds the semantics
// against the hardware).  The packed kernels prefetch the phase-1 columns of a wavefront's NEXT tile this way.  Written as asm
// statements: beside the builtin the compiler waits for vmcnt(0) at the next use of any ordinary load — in the middle of the
// pipelined runs.  What follows from that: the loads are absent from the compiler's count of outstanding loads (its counted
// waits can only wait longer for them, never too short: loads return in order), and their data is waited for by hand — an asm
// s_waitcnt vmcnt(0) with a memory clobber in front of the first read of the area.  M0 is the compiler's: saved and restored.
// (offsets: bytes from the column's base, 32 bits; the bases: kernel arguments — scalar registers never written by the VALU)
template <bool LIB>
__device__ __forceinline__ void pfl_dma_cols(const u32 dst, const u32 off2, const u32 off4, const u32 offo, const u16 *flag, const u16 *lib,
                                             const int32_t *tid, const int32_t *pos, const int32_t *tlen, const u32 *co, const u32 *so) {
    u32 keep;
    if (LIB)
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_ushort %1, %5\n\t"
                     "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_ushort %1, %6\n\t"
                     "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %2, %7\n\t"
                     "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %2, %8\n\t"
                     "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %2, %9\n\t"
                     "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %3, %10\n\t"
                     "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %3, %11\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off2), "v"(off4), "v"(offo), "s"(dst), "s"(flag), "s"(lib), "s"(tid), "s"(pos), "s"(tlen), "s"(co), "s"(so)
                     : "memory", "scc");
    else
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_ushort %1, %5\n\t"
                     "s_add_u32 m0, m0, 0x200\n\ts_nop 0\n\tglobal_load_lds_dword %2, %6\n\t"
                     "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %2, %7\n\t"
                     "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %2, %8\n\t"
                     "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %3, %9\n\t"
                     "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %3, %10\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off2), "v"(off4), "v"(offo), "s"(dst), "s"(flag), "s"(tid), "s"(pos), "s"(tlen), "s"(co), "s"(so)
                     : "memory", "scc");
}
// ... and the second round trip of a tile: up to three operations of the record's CIGAR and the bounds of its contig (the low
// words of contig_off[tid] and [tid + 1]), by the lanes that want them
__device__ __forceinline__ void pfl_dma_rt2(const u32 dst, const u32 o0, const u32 o1, const u32 o2, const u32 oc, const u32 oc1,
                                            const u32 *cigar, const int64_t *contig_off) {
    u32 keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dword %1, %7\n\t"
                 "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %2, %7\n\t"
                 "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %3, %7\n\t"
                 "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %4, %8\n\t"
                 "s_add_u32 m0, m0, 0x100\n\ts_nop 0\n\tglobal_load_lds_dword %5, %8\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(o0), "v"(o1), "v"(o2), "v"(oc), "v"(oc1), "s"(dst), "s"(cigar), "s"(contig_off)
                 : "memory", "scc");
}

// stores the compiler does not count either (the records a tile's phase 1 leaves to the general pass): base a wave-uniform
// pointer, off a byte offset.  (s_nop 4 opening every statement that hands a scalar register pair to a memory instruction:
// the pair may come fresh from a v_readlane — a spilled scalar — and a vector memory instruction reads it five states late;
// the compiler pads its own instructions, not an asm statement's)
__device__ __forceinline__ void pfl_store_b32(const void *base, const u32 off, const u32 v) {
    asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2" :: "v"(off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void pfl_store_b128(const void *base, const u32 off, const u32x4 v) {
    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(off), "v"(v), "s"(base) : "memory");
}

__device__ __forceinline__ void patch_put(unsigned long long *__restrict__ patch0, unsigned long long *__restrict__ n_patch0, long long cap,
                                          int parts, bool on, u32 idx, u32 newq) {
    const u64 m = __ballot(on);
    if (m == 0) return;
    // (the block's part of the list: a counter per part — one list for the whole launch is one address all wavefronts queue at)
    const u32 part = blockIdx.x & (u32)(parts - 1);
    unsigned long long *__restrict__ patch = patch0 + (size_t)part * (size_t)cap, *__restrict__ n_patch = n_patch0 + part;
    const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
    u64 base = 0;
    if ((int)(threadIdx.x & 63u) == leader) base = atomicAdd(n_patch, (unsigned long long)__popcll(m));
    const u64 b = (u64)(u32)rl((int)(u32)base, leader) | ((u64)(u32)rl((int)(u32)(base >> 32), leader) << 32);
    if (on) {
        const u64 at = b + (u64)mbcnt64(m, 0);
        if ((long long)at < cap) patch[at] = (u64)idx | ((u64)newq << 32);
    }
}

// bytes [lo, hi) of a 64-bit word, the range clamped to [0, 8)
__device__ __forceinline__ u64 byte_range(int lo, int hi) {
    lo = lo < 0 ? 0 : lo;
    hi = hi > 8 ? 8 : hi;
    if (hi <= lo) return 0ull;
    const u64 upto = hi >= 8 ? ~0ull : ((1ull << (8 * hi)) - 1ull);
    return upto & ~((1ull << (8 * lo)) - 1ull);
}
// static byte masks of the lane (side, 8 m) of a record (MdxDims): vm = the byte is a task of a complete
// record, em = the byte is a read column (its read byte is compared; a flank byte is only classified)
__device__ __forceinline__ void lane_masks(const MdxDims &d, int side, int m8, u64 &vm, u64 &em) {
    if (!side) {
        vm = byte_range(0, d.A + d.L - m8);
        em = vm & byte_range(d.A - m8, 8);
    } else {
        vm = byte_range(m8 + 8 - d.A - d.L, 8);
        em = vm & byte_range(0, m8 + 8 - d.A);
    }
}
// 8 flag bits -> bit 7 of the corresponding bytes of a 64-bit word
__device__ __forceinline__ u64 spread_bits(u32 m) {
    const u32 lo = (((m & 0xFu) * 0x00204081u) & 0x01010101u) << 7;
    const u32 hi = ((((m >> 4) & 0xFu) * 0x00204081u) & 0x01010101u) << 7;
    return (u64)lo | ((u64)hi << 32);
}
// bit 7 of the four bytes of v -> 4 flag bits
__device__ __forceinline__ u32 gather_bits(u32 v) { return (((v >> 7) & 0x01010101u) * 0x00204081u >> 21) & 0xFu; }

// record descriptor word w1 (one int per record)
#define D_REV 1
#define D_SIMPLE 2
#define D_HASQ 4
#define D_FULL 8        // plain-match record with every task present (nq >= L, both flanks complete)
#define D_PRE 32        // gapped record whose first / last match run is counted by the fast path
#define D_ONE 64        // ... and the rest of it too: [H][S] M {I|D} M [S][H], nothing left for the CIGAR walk
// staging entry word w / event word
#define PK_ONE 0x00080000u     // entry of a D_ONE record
#define D_NB_SHIFT 8    // nbefore, 8 bits
#define D_NA_SHIFT 16   // nafter, 8 bits

// (final column - 4) of substitution ref>read for classes A,C,T,G, 4 bits per [ref][read] entry
#define SUB_LUT 0x0ba0803961072540ull
//   A>: -,4,5,2   C>: 7,-,1,6   T>: 9,3,-,8   G>: 0,10,11,-   (entry index = ref * 4 + read)

// MIS column of the pair (r, s), classes A,C,T,G,- with r != s, without a table load: substitutions from
// SUB_LUT, deletions r>- = 16 + {0,2,1,3}[r], insertions ->s = 20 + {0,2,1,3}[s] (seq.py:6-30 order)
__device__ __forceinline__ int mis_col(int r, int s) {
    if (r < 4 && s < 4) return 4 + (int)((SUB_LUT >> (4 * (r * 4 + s))) & 15ull);
    asm volatile("" ::: "memory");   // the rare case stays a branch (not if-converted into the common path)
    const int k = r < 4 ? r : s;
    return (r < 4 ? 16 : 20) + (((k & 1) << 1) | (k >> 1));
}

// Rare path of a plain-match record column: (ch, rch) is not a plain match.  The read base goes to
// CMP (statistics.py:75-83); a substitution/indel goes to its MIS column only — the accompanying
// reference-base count of statistics.py:30 is derived by finalize_kernel (mis[r] = matches +
// sum of the r>x columns).
template <bool USE_LDS>
__device__ __forceinline__ void rare_column(u32 *lds, u64 *raw, int b_mis, int b_cmp, int L, int side,
                                            int p, int pc, u32 ch, int rch, bool masked) {
    // p: misincorporation position (column), pc: composition position (query index; differs behind a deletion)
    const int s = classify_read(ch);
    const int sp = (side ? L : 0) + p;      // (no 32-bit multiplies on this path: they run at a quarter of the rate)
    if (s < 4) bump<USE_LDS>(lds, raw, b_cmp + ((side ? L : 0) + pc) * 4 + s);
    if (!masked && s <= SYM_GAP) {
        const int r = classify_ref(rch);
        if (r <= SYM_GAP && r != s) {  // statistics.py:26-35
            bump<USE_LDS>(lds, raw, b_mis + __mul24(sp, 25) + mis_col(r, s));
        }
    }
}

enum { STEP_C = 0, STEP_P = 1, STEP_GI = 2, STEP_GD = 3 };   // kinds of fast-path steps (tabulate_kernel::count)

// Eight optimistic increments by position: word (base_bytes / 4) + (STRIDE / 4) j + class of reference byte j gets
// bit 0 of byte j of the mask (MIS rows: STRIDE 100, CMP rows: STRIDE 16).  LDS only.
template <int STRIDE>
__device__ __forceinline__ void direct8(u32 *lds, u32 r_lo, u32 r_hi, u32 base_bytes, u32 m_lo, u32 m_hi) {
    u32 *const row = lds + (base_bytes >> 2);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const u32 r = j < 4 ? r_lo : r_hi, m = j < 4 ? m_lo : m_hi;
        const u32 k = (r >> (8 * (j & 3) + 1)) & 3u;
        atomicAdd(&row[j * (STRIDE / 4) + k], (m >> (8 * (j & 3))) & 1u);
    }
}

// element idx of a column, the byte offset computed in 32 bits (batches hold fewer than 2^30 records)
template <class T>
__device__ __forceinline__ T ld32(const T *base, u32 idx) {
#if MDX_NT
    return __builtin_nontemporal_load((const T *)((const char *)base + (size_t)(idx * (u32)sizeof(T))));
#else
    return *(const T *)((const char *)base + (size_t)(idx * (u32)sizeof(T)));
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// The packed form (tabulate_kernel<.., PK>): SEQ and the resident reference as 4-bit codes (MDX_SEQ_4BIT, include/mdx.h:
// 1 = A, 2 = C, 4 = T, 8 = G — bit k = symbol class k — and 0 for anything else; 15 = the gap symbol, in registers and
// events only), two bases per byte, low nibble first.  A lane owns SIXTEEN consecutive bases of a record's window — two
// dwords: an aligned dwordx3 load and two v_alignbit_b32 per operand — so a record takes G4 = 2 ceil((A + L) / 16) lanes
// (10 at the defaults) and a step counts six records with the vector-memory instructions that count three in the ASCII
// kernel (every such instruction costs the texture addresser its 16+ cycles whatever it loads).  What changes most is the
// counting: a code is one-hot, so the sixteen reference nibbles of a lane *are* the 64 increments of its step — bit
// 4 j + k set: base k at the lane's nibble j — and they are added into bit-sliced counters held in registers (eight
// planes of two dwords: bit b of plane i = bit i of counter b), four steps' worth with two v_bitop3_b32 per carry-save
// adder.  The slots of a step are tied to a strand — [0, H4) forward records, [H4, 2 H4) reverse ones; the entries of a
// run are sorted by strand — so one set of planes does.  No LDS update per plain match; the planes are folded into the
// block's TC table ([base][64 j + lane]) every 255 steps at most, and the table reaches the block's partial slot in the
// ASCII kernel's layout.  A nibble that is not a task is zeroed and counts nothing (no DMP correction); an event undoes
// the count of its reference nibble in TC like the ASCII kernel's.  Events are queued per half lane (one dword of each
// string); nothing is drained inside a run — what does not fit the LDS queue goes to the wavefront's overflow list.
#define SYM4_GAP 15u
// 4-bit code -> symbol class (A, C, T, G = 0..3; '-' = 4; anything else 5)
__device__ __forceinline__ int cls4(u32 nib) {
    const bool one = nib != 0u && (nib & (nib - 1u)) == 0u;
    return one ? __ffs((int)nib) - 1 : (nib == SYM4_GAP ? SYM_GAP : SYM_OTHER);
}
// --min-basequal folded into the column (MDX_SEQ_4BITQ): a symbol whose quality is below the threshold is stored as the
// complement of its code — three bits set: 14, 13, 11, 7 for A, C, T, G; all four for a symbol that is no base (its mask
// matters too: behind an N operation align.py:65-71 masks the reference symbol of the same LEFT index, which pairs with
// another read symbol from the right end).  The code the nibble stands for, and whether it is masked:
__device__ __forceinline__ u32 unmask4(u32 nib, bool &masked) {
    masked = __builtin_popcount(nib) >= 3;
    return masked ? nib ^ 15u : nib;
}
// nibbles [lo, hi) of a 64-bit word, the range clamped to [0, 16)
__device__ __forceinline__ u64 nibble_range16(int lo, int hi) {
    lo = lo < 0 ? 0 : lo;
    hi = hi > 16 ? 16 : hi;
    if (hi <= lo) return 0ull;
    const u64 upto = hi >= 16 ? ~0ull : ((1ull << (4 * hi)) - 1ull);
    return upto & ~((1ull << (4 * lo)) - 1ull);
}
// lane_masks for the packed kernel's lanes of sixteen nibbles (side, first window nibble m16): vm = the nibble is a task of
// a complete record, em = it is a read column
__device__ __forceinline__ void lane_masks16(const MdxDims &d, int side, int m16, u64 &vm, u64 &em) {
    if (!side) {
        vm = nibble_range16(0, d.A + d.L - m16);
        em = vm & nibble_range16(d.A - m16, 16);
    } else {
        vm = nibble_range16(m16 + 16 - d.A - d.L, 16);
        em = vm & nibble_range16(0, m16 + 16 - d.A);
    }
}
// eight bits -> eight nibbles (bit j -> all four bits of nibble j)
__device__ __forceinline__ u32 spread8(u32 b) {
    u32 x = b & 0xFFu;
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return (x << 4) - x;
}
// carry-save adder over 32 one-bit columns: p + a + b = p' + 2 c
// (written out: the carry first, then the sum over p itself — left to the scheduler the sum comes first into a register of
// its own and every plane is copied back at the end of the loop body: 37 v_mov per group of four steps)
__device__ __forceinline__ void bs_csa(u32 &p, const u32 a, const u32 b, u32 &c) {
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe8\n\tv_bitop3_b32 %1, %1, %2, %3 bitop3:0x96"    // majority; p ^ a ^ b
        : "=&v"(c), "+v"(p) : "v"(a), "v"(b));
}
// half adder: p + a = p' + 2 c
__device__ __forceinline__ void bs_ha(u32 &p, const u32 a, u32 &c) {
    asm("v_and_b32 %0, %1, %2\n\tv_xor_b32 %1, %1, %2" : "=&v"(c), "+v"(p) : "v"(a));
}
// c, a word of weight 2^from, into the planes from `from` upwards
template <int FROM>
__device__ __forceinline__ void bs_ripple(u32 (&p)[8], u32 c) {
#pragma unroll
    for (int i = FROM; i < 8; i++) {
        u32 t;
        bs_ha(p[i], c, t);
        c = t;
    }
}
// the words of a group of N steps into one set of planes: (x0, x1) -> plane 0 and a carry, (x2, x3) -> plane 0 and another,
// the two carries -> plane 1 and one of weight 4, which ripples upwards
template <int N>
__device__ __forceinline__ void bs_add_group(u32 (&pl)[8], const u32 (&x)[4]) {
    if (N == 1) bs_ripple<0>(pl, x[0]);
    else if (N == 2) { u32 c; bs_csa(pl[0], x[0], x[1], c); bs_ripple<1>(pl, c); }
    else {
        u32 c0, c1, c2;
        bs_csa(pl[0], x[0], x[1], c0);
        if (N == 3) bs_ha(pl[0], x[2], c1);
        else bs_csa(pl[0], x[2], x[3], c1);
        bs_csa(pl[1], c0, c1, c2);
        bs_ripple<2>(pl, c2);
    }
}
// ... the same for four words, the carry of weight 4 handed out instead of rippled (tabulate_kernel's paired groups)
__device__ __forceinline__ void bs_group4(u32 (&pl)[8], const u32 (&x)[4], u32 &c2) {
    u32 c0, c1;
    bs_csa(pl[0], x[0], x[1], c0);
    bs_csa(pl[0], x[2], x[3], c1);
    bs_csa(pl[1], c0, c1, c2);
}
// resident reference bytes (encode_ref_kernel's, guard bands included) -> 4-bit codes, eight bases per thread and step
__device__ __forceinline__ u32 code4_of_ref(u32 b) {
    // 'A' 0x41, 'C' 0x43, 'T' 0x54, 'G' 0x47: class (b >> 1) & 3; 0x84 / 0x85: nothing
    return (b & 0x80u) ? 0u : 1u << ((b >> 1) & 3u);
}
__global__ void encode_ref4_kernel(const u8 *__restrict__ in, u8 *__restrict__ out, i64 n2) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (; i < n2; i += stride) out[i] = (u8)(code4_of_ref(in[2 * i]) | (code4_of_ref(in[2 * i + 1]) << 4));
}
void mdx_k_encode_ref4(const u8 *d_codes, u8 *d_ref4, int64_t n, hipStream_t s) {
    const i64 n2 = n / 2;
    if (n2 <= 0) return;
    const int grid = (int)((n2 + 255) / 256 < 8192 ? (n2 + 255) / 256 : 8192);
    hipLaunchKernelGGL(encode_ref4_kernel, dim3(grid), dim3(256), 0, s, d_codes, d_ref4, n2);
}
// a read symbol -> its code: exactly 'A', 'C', 'G', 'T' (statistics.py:27, 101), anything else 0
__device__ __forceinline__ u32 code4_of_read(u32 ch) {
    const int c = classify_read(ch);
    return c < 4 ? 1u << c : 0u;
}
__global__ void pack_seq_kernel(const u8 *__restrict__ in, u8 *__restrict__ out, i64 n) {
    const i64 n2 = (n + 1) / 2;
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (; i < n2; i += stride) {
        const u32 lo = code4_of_read(in[2 * i]), hi = 2 * i + 1 < n ? code4_of_read(in[2 * i + 1]) : 0u;
        out[i] = (u8)(lo | (hi << 4));
    }
}
// MDX_FLAG_HAS_QUAL (include/mdx.h) on the records of a resident batch that have qualities: at least one base, and a first
// quality byte that is not 0xFF (main.py:185, rescale.py:306)
__global__ void mark_has_qual_kernel(u16 *__restrict__ flag, const u32 *__restrict__ seq_off, const u8 *__restrict__ qual, i64 n) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const u32 so = seq_off[i];
        if (seq_off[i + 1] != so && qual[so] != 0xFF) flag[i] = (u16)(flag[i] | 0x4000u);
    }
}
void mdx_k_mark_has_qual(uint16_t *d_flag, const uint32_t *d_seq_off, const uint8_t *d_qual, int64_t n, hipStream_t s) {
    if (n <= 0 || !d_qual) return;
    const int grid = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(mark_has_qual_kernel, dim3(grid), dim3(256), 0, s, (u16 *)d_flag, d_seq_off, d_qual, (i64)n);
}
void mdx_k_pack_seq(const u8 *d_ascii, u8 *d_packed, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    const i64 n2 = (n + 1) / 2;
    const int grid = (int)((n2 + 255) / 256 < 16384 ? (n2 + 255) / 256 : 16384);
    hipLaunchKernelGGL(pack_seq_kernel, dim3(grid), dim3(256), 0, s, d_ascii, d_packed, (i64)n);
}
// ... and back (for the kernels that read ASCII: --min-basequal, rescaling, the generic path): a code that is not a
// base becomes 'N' — which is what every such symbol is to the reference
__global__ void unpack_seq_kernel(const u8 *__restrict__ in, u8 *__restrict__ out, i64 n) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        // (MDX_SEQ_4BITQ: a masked base is the complement of its code — the base it is; the qualities say the rest)
        bool m;
        const u32 nib = unmask4((in[i >> 1] >> (4 * (i & 1))) & 15u, m);
        const int c = cls4(nib);
        out[i] = c < 4 ? (u8)((0x47544341u >> (8 * c)) & 0xFFu) : (u8)'N';
    }
}
void mdx_k_unpack_seq(const u8 *d_packed, u8 *d_ascii, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    const int grid = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(unpack_seq_kernel, dim3(grid), dim3(256), 0, s, d_packed, d_ascii, (i64)n);
}

// FAST: the 8-byte-lane path for plain records (MdxDims::fast_ok(), reference shorter than 4 GiB);
// otherwise every record takes the generic CIGAR walk.
// RS: the fused tabulate + rescale launch (MdxFuse; mapdamage/rescale.py:195-365 for the records of the tile loop)
// PK: the packed form — 4-bit SEQ and reference, bit-sliced counting (see above); the LDS image holds one library
// ML: several libraries in one launch of the packed kernel (reader.py:47-50, statistics.py:12-20: the tables are keyed by
//     library).  The batch arrives ordered by library (mdx_libsort.hip: MdxTabArgs::perm, ::lib_start — the flag filter
//     applied on the way) and every pool of blocks counts ONE library: the pools are dealt to the libraries in proportion to
//     their tiles on the device (ml_plan_kernel, MdxTabArgs::ml_plan), a library's pools share its tiles among themselves as
//     the pools of a one-library launch share the batch's, and the reduction adds a block's image to the tables of its pool's
//     library.  (Until the end of round 5 every block went through all libraries, an epoch each: 8 libraries 1.27 x the
//     one-library launch; now 1.03 x.)
template <bool USE_LDS, bool MASK, bool FAST, bool RS = false, bool PK = false, bool ML = false>
__global__ __launch_bounds__(RS ? MDX_FUSE_BLOCK : (PK ? MDX_PK_BLOCK : MDX_BLOCK), RS ? MDX_FUSE_WPS : (PK ? MDX_PK_WPS : MDX_WPS)) void tabulate_kernel(MdxTabArgs a) {
    static_assert(!RS || (USE_LDS && FAST && !MASK), "the fused kernel is the unmasked fast LDS kernel");
    static_assert(!PK || (USE_LDS && FAST), "the packed kernel is the fast LDS kernel (plain, with the fused rescaling, or with --min-basequal)");
    static_assert(!ML || (PK && !RS), "a library per pool: the packed kernels (plain and --min-basequal)");
    // (the packed kernels: the launch's own block, 512 or 1024 threads — MdxPkConfig)
    const int BLOCK = RS ? MDX_FUSE_BLOCK : (PK ? (int)blockDim.x : MDX_BLOCK);
    // pfl: the phase-1 columns of a tile come out of the wavefront's prefetch area in the LDS (see the tile loop) — the packed
    // kernels but the fused one (its image has no room for the areas)
    constexpr bool PFLC = PK && !RS;
    constexpr bool pfl = PFLC;
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const MdxDims d = a.dims;
    const int L = d.L, A = d.A;
    const int lane = threadIdx.x & 63;
    // (wave-uniform, and told so: the staging and queue addresses derived from it live in scalar registers)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int waves_per_block = BLOCK / 64;
    const u32 gwave = blockIdx.x * waves_per_block + wave;
    const u32 nwaves = gridDim.x * waves_per_block;
    u64 *raw = a.raw;
#ifdef MDX_WAVE_CLK       // (instrumented builds: the wavefront's clock at its start, behind its tile loop and at its end)
    if (a.dbg_clk && lane == 0) a.dbg_clk[3 * (size_t)gwave] = wall_clock64();
#endif
#ifdef MDX_PHASE_CLK      // (instrumented builds, tools/experiments/phase_clk.py: the wavefront's shader-clock ticks by part of the kernel)
    u32 ph_acc[16] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    int ph_cur = 0;
    u64 ph_t0 = __builtin_amdgcn_s_memtime();
#define MDX_PH(id) do { const u64 t_ = __builtin_amdgcn_s_memtime(); const u32 dt_ = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(t_ - ph_t0)); \
        _Pragma("unroll") for (int q_ = 0; q_ < 16; q_++) ph_acc[q_] += ph_cur == q_ ? dt_ : 0u; ph_t0 = t_; ph_cur = (id); } while (0)
#define MDX_PH_IN(id) const int ph_prev_ = ph_cur; MDX_PH(id)
#define MDX_PH_OUT() MDX_PH(ph_prev_)
#define MDX_PH_IMPL(x) x##_impl
#else
#define MDX_PH_IMPL(x) x
#define MDX_PH(id) do { } while (0)
#define MDX_PH_IN(id) do { } while (0)
#define MDX_PH_OUT() do { } while (0)
#endif

    // nine-entry LDS table of byte masks (entry n = the low n bytes of a 64-bit word set): the per-record byte masks
    // of the partial steps are two or three lookups instead of 64-bit shifts
    const int QCAP = RS ? a.rs.qcap : EVQ_CAP;      // events of the ASCII kernels' queue (20 bytes each)
    u64 *const ltab = (u64 *)((u8 *)(lds + a.queue_off) + (BLOCK / 64) * (PK ? MDX_PK_EVQ_BYTES : QCAP * 20));
    // RS: the fused records count into a TC table of their own (the reference bases of their columns are part of the
    // rescale summary, rescale.py:142-143), added to the first one at block end; behind it four words for those counts,
    // the lookup table and the terms of the model
    const int rs_npos = RS ? 1 + a.rs.len5p + a.rs.len3p : 0;
    const int rs_ncnt = 752 + 2 * rs_npos * 94;
    // (PK: no second table — the packed steps count the reference bases of the fused records' columns themselves, see count16)
    u32 *const rs_cnt = lds + (RS ? a.rs.tcb_off + (PK ? 0 : d.nlib * d.w_tc) : 0);
    const u8 *const l_lut = (const u8 *)(rs_cnt + 4);
    const double *const l_term = (const double *)(l_lut + ((2 * rs_npos * 94 + 15) & ~15));
    // ... and per wavefront one 64-bit word per staging entry: bit sub * npos + key = the record has a rescaled column of
    // that kind (drain_all sets them; the MR sum is formed from them, in the reference's order, when the run is over)
    u64 *const mrm = (u64 *)(l_term + 2 * rs_npos) + (RS ? wave * MDX_FUSE_MRM : 0);
    // ... and a list of the transitions drain_all has found in fused records: {byte offset of the column in the quality
    // column, summary index | rescaled << 15}; their qualities are fetched together when the run is over (rsq_flush) —
    // a load inside drain_all would stall the wavefront once per drain
    u32x2 *const rsq = (u32x2 *)((u64 *)(l_term + 2 * rs_npos) + (BLOCK / 64) * MDX_FUSE_MRM) + (RS ? wave * (MDX_FUSE_RSQ + 1) : 0);
    u32 *const rsq_cnt = (u32 *)(rsq + MDX_FUSE_RSQ);
    if (RS && lane == 0) *rsq_cnt = 0u;
    int bcA = 0, bcC = 0, bcG = 0, bcT = 0;   // RS: reference bases (read orientation) this lane has counted itself
    if (USE_LDS) {
        for (i64 i = threadIdx.x; i < d.w_total; i += BLOCK) lds[i] = 0;
        if (FAST && !PK && threadIdx.x < 9) ltab[threadIdx.x] = threadIdx.x >= 8 ? ~0ull : ((1ull << (8 * threadIdx.x)) - 1ull);
        // (PK: seventeen 64-bit masks, entry n = the low n nibbles set)
        if (PK && threadIdx.x < 17) ltab[threadIdx.x] = threadIdx.x >= 16 ? ~0ull : ((1ull << (4 * threadIdx.x)) - 1ull);
        // (PK: what the drain wants to know of a pair of nibbles, reference << 4 | read: [2:0] class of the reference symbol,
        // [5:3] of the read symbol, [10:6] the MIS column of the pair — 31: none, statistics.py:26-35)
        if (PK && threadIdx.x < 256) {
            const int rc = cls4(threadIdx.x >> 4), sc = cls4(threadIdx.x & 15u);
            const int col = (sc <= SYM_GAP && rc <= SYM_GAP && rc != sc) ? mis_col(rc, sc) : 31;
            ((u16 *)(ltab + 17 + 64))[threadIdx.x] = (u16)(rc | (sc << 3) | (col << 6));
        }
        if (RS) {
            for (int i = threadIdx.x; i < (PK ? 0 : d.nlib * d.w_tc) + 4; i += BLOCK) lds[a.rs.tcb_off + i] = 0;
            for (int i = threadIdx.x; i < 2 * rs_npos * 94; i += BLOCK) ((u8 *)(rs_cnt + 4))[i] = a.rs.lut[i];
            for (int i = threadIdx.x; i < 2 * rs_npos; i += BLOCK) ((double *)(l_lut + ((2 * rs_npos * 94 + 15) & ~15)))[i] = a.rs.term[i];
            // the block's summary row: zeroed here, counted into with global atomics by this block alone
            u32 *const row = a.rs.subs_part + (size_t)blockIdx.x * rs_ncnt;
            for (int i = threadIdx.x; i < rs_ncnt; i += BLOCK) row[i] = 0;
            __threadfence();
        }
        __syncthreads();
    }

    // Window loads are dword-aligned: a lane fetches the 12 bytes at (address & ~3) and funnel-shifts its
    // 8-byte window out of them (v_alignbyte_b32).  Measured on gfx950 (tools/ubench_loads.hip): a wavefront
    // load whose lane addresses are not dword-aligned costs the texture-address unit about twice as much
    // (14.1 vs 8.4 ns for the SEQ pattern, 16.2 vs 12.6 ns for the reference pattern), and that unit is
    // what bounds this kernel.  The bases are aligned down here, their phase goes into the lane offsets.
    // (PK: offsets count bases — nibbles —, so the phase of the SEQ column's base address counts twice; the 4-bit
    // reference is the library's own allocation, dword-aligned)
    const u32 ph_ref = PK ? 0u : (u32)((size_t)(a.ref - 256) & 3), ph_seq = (PK ? 2u : 1u) * (u32)((size_t)a.seq & 3),
              ph_qual = MASK ? (u32)((size_t)a.qual & 3) : 0u;
    const u8 *const refW = PK ? a.ref4 : a.ref - 256 - ph_ref;  // start of the guard band (dword-aligned): window offsets are >= 0
    const u8 *const seqW = a.seq - ((size_t)a.seq & 3);
    const u8 *const qualW = MASK ? a.qual - ph_qual : nullptr;

    // Per-lane constants of the fast path (MdxDims): slot g = lane / G holds one record of the step; within
    // the slot, lanes [0, nl8) are the left side, [nl8, 2 nl8) the right side; lane (side, m) owns the bytes
    //   reference: refW[rfL + c_ro + (side ? nq : 0)]     refW = ref - 256, rfL = rbase - A + 256
    //   SEQ:       seq[sq + c_so + (side ? nq : 0)]
    // Lanes beyond R * G idle (they shadow lane 0 and are masked out of every count).
    int c_slot = 0, c_side = 0, c_m8 = 0;
    u32 c_ro = 0, c_so = (u32)(-A), c_cm = 0;
    u32 c_vm_lo = 0, c_vm_hi = 0, c_em_lo = 0, c_em_hi = 0;  // em is a subset of vm
    int p_strand = 0;      // PK: the strand of this lane's slot
    u32 c_evw = 0;         // PK: the lane's part of an event word (see qE)
    if (FAST && PK) {
        // the packed kernel's lanes: slot g = lane / G4 — [0, H4) forward-strand records, [H4, 2 H4) reverse-strand ones —,
        // within the slot lanes [0, nl16) are the left side, [nl16, 2 nl16) the right side; lane (side, m) owns sixteen bases
        const int g = lane / d.G4, ll = lane - g * d.G4;
        if (g < 2 * d.H4) {
            p_strand = g >= d.H4;
            c_slot = g - p_strand * d.H4;
            c_side = ll >= d.nl16;
            c_m8 = 16 * (ll - c_side * d.nl16);
            if (c_side) { c_ro = (u32)(2 * A - 16 - c_m8); c_cm = 0x7FFFu; }
            else c_ro = (u32)c_m8;
            u64 vm, em;
            lane_masks16(d, c_side, c_m8, vm, em);
            c_vm_lo = (u32)vm; c_vm_hi = (u32)(vm >> 32);
            c_em_lo = (u32)em; c_em_hi = (u32)(em >> 32);
        }
        c_evw = (u32)lane | ((u32)c_side << 6) | ((u32)(c_m8 >> 4) << 7) | ((u32)p_strand << 11);
        // (the read-column masks by lane, for the drain; every wavefront writes the same words and reads them after its own write)
        ltab[17 + lane] = (u64)c_em_lo | ((u64)c_em_hi << 32);
    }
    if (FAST && !PK) {
        const int g = lane / d.G, ll = lane - g * d.G;
        if (g < d.R) {
            c_slot = g;
            c_side = ll >= d.nl8;
            c_m8 = 8 * (ll - c_side * d.nl8);
            if (c_side) {
                c_ro = (u32)(2 * A - 8 - c_m8); c_so = (u32)(A - 8 - c_m8); c_cm = 0x7FFFu;
            } else {
                c_ro = (u32)c_m8; c_so = (u32)(c_m8 - A);
            }
            u64 vm, em;
            lane_masks(d, c_side, c_m8, vm, em);
            c_vm_lo = (u32)vm; c_vm_hi = (u32)(vm >> 32);
            c_em_lo = (u32)em; c_em_hi = (u32)(em >> 32);
        }
    }
    c_ro += ph_ref;   // the phases of the aligned-down bases
    const u32 c_qo = c_so + ph_qual;
    c_so += ph_seq;
    // PK: c_so - c_ro, the same in every lane (the staging entries carry SEQ offset - reference offset + this)
    const u32 pk_dso = ph_seq - (u32)A;
    const u32 c_hivm_lo = c_vm_lo & 0x80808080u, c_hivm_hi = c_vm_hi & 0x80808080u;
    // (the lane field of an event word is c_lane4 << 16: no register of its own across the hot loop)
    const u32 c_lane4 = (u32)lane << 2;     // byte offset of word `lane` (the dynamic LDS starts at address 0)

    // Record staging of the fast path (wave-private, LDS): phase 1 writes one 16-byte entry per plain record,
    // complete records first, {rfL, sq, nq | nbefore << 16 | nafter << 24,
    //                          TC base bytes [17:8] | library [29:24] | has qualities [30] | reverse strand [31]};
    // phase 2 reads the entry of its slot with one ds_read_b128 (no v_readlane broadcast).
    uint4 *const stg = (uint4 *)(lds + a.stage_off) + wave * mdx_stage_entries(d);
    // Rare-event queue (wave-private, LDS): the lanes holding a byte that is not a plain match append
    // {read 8 bytes, reference 8 bytes, record word | lane << 18 | masked-quality flags [7:0]}; the queue is
    // drained completely, 64 events in parallel, whenever the next step might not fit.
    u32x2 *const qS = (u32x2 *)((u8 *)(lds + a.queue_off) + wave * (QCAP * 20));
    u32x2 *const qR = qS + QCAP;
    u32 *const qW = (u32 *)(qR + QCAP);
    // RS: a step that finds the queue too full for a step's worth of events does nothing and says so (count()); the run
    // stops behind it, drains outside its pipelined loop and starts again from that step (run())
    bool rs_ovf = false;
    int rs_kredo = 0;
    int qcount = 0;
    // PK: an event is a lane (sixteen nibbles) with a read column that is not a plain match, five words: its read nibbles, its
    // reference nibbles (as the step saw them: outside the step's tasks both are zero), and [5:0] lane | [6] side |
    // [10:7] the lane's first window nibble / 16 | [11] reverse strand | [12] single-deletion entry, then [15:13] = deleted
    // bases g and [20:16] = first nibble of the lane behind the deletion (right side: the nibbles below it).
    // {S, R}[QCAP] | W[QCAP] in the wavefront's event area (MDX_PK_QCAP events).  The queue is drained behind a run once
    // 64 events wait, and inside one only when a step's events would not fit (a tile over a stretch of mismatches).
    // Nothing in a run stores to global memory: with a store possibly pending the compiler waits for vmcnt(0) at the head
    // of the pipelined loop — loads and stores share the counter and may retire out of order — instead of for the oldest
    // step's loads only.
    uint4 *const qQ = (uint4 *)(lds + a.queue_off + (PK ? wave * (MDX_PK_EVQ_BYTES / 4) : 0));
    u32 *const qE = (u32 *)(qQ + MDX_PK_QCAP);
    // (their LDS addresses: the dynamic LDS starts at address 0)
    typedef __attribute__((address_space(3))) u32x4 lds_u4;
    typedef __attribute__((address_space(3))) u32 lds_u1;
    const u32 qQ_a = (u32)(size_t)(lds_u4 *)qQ, qE_a = qQ_a + 16u * MDX_PK_QCAP;
    // the wavefront's prefetch area (MdxTabArgs::pfl_off): [column][lane] dwords
    const u32 pfl_a = PFLC ? (u32)__builtin_amdgcn_readfirstlane((int)((u32)(size_t)(lds_u1 *)(lds + a.pfl_off) + (u32)wave * (u32)MDX_PFL_WAVE_BYTES)) : 0u;
    static_assert(MDX_PK_QCAP >= 64 && MDX_PK_QCAP % 4 == 0, "a step's events fit an empty queue");
    const u32x2 *const emtab = (const u32x2 *)(ltab + 17);
    const u16 *const pktab = (const u16 *)(ltab + 17 + 64);
    // PK: the bit-sliced counters of this lane's sixteen window nibbles (bit 4 j + k of plane i, low / high dword = nibbles
    // 0-7 / 8-15: bit i of the count of base k at nibble j; the lane's slot fixes the strand) and the steps added since they
    // were last folded into TC (at most 255: eight planes)
    u32 bsL[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, bsH[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    // PK with --min-basequal: a second set — the read bases of the masked columns (align.py:65-71 turns both symbols of such a
    // column into N: it counts its read base in the composition table and nothing else), folded into CMP by position
    u32 b2L[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, b2H[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    int bs_steps = 0;
    // Paired groups (the kernels with the registers for them): a group of four steps leaves one word of weight 4 per dword;
    // rippled through planes 2..7 that is twelve instructions.  Two groups one after the other hand their two words to one
    // adder into plane 2, and its carry ripples from plane 3: fourteen for both.
    constexpr bool HS = PK && MDX_PK_DEFER && !MASK && !RS && (!ML || MDX_PK_DEFER_ML);
    // fold the planes into the block's TC table, PK layout: word [base k][64 j + lane]; per bit position s of the bytes of
    // a plane, the four counters of bits s, s + 8, s + 16, s + 24 are gathered as the bytes of one word
    // (inlined at every call site: a call would take the planes through memory)
    auto MDX_PH_IMPL(bs_flush) = [&]() __attribute__((always_inline)) {
#ifdef MDX_ABL_NOFLUSH       // (ablation builds — wrong tables, the instruction counts of what is left: tools/ablate.sh)
        bs_steps = 0;
        return;
#endif
        if (PK) {
            u32 *const tcp = lds + d.off_tc() + lane;
#pragma unroll
            for (int half = 0; half < 2; half++) {
#pragma unroll
                for (int sft = 0; sft < 8; sft++) {
                    u32 acc = 0u;
#pragma unroll
                    for (int i = 0; i < 8; i++) acc |= (((half ? bsH[i] : bsL[i]) >> sft) & 0x01010101u) << i;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int b = sft + 8 * q, j = 8 * half + (b >> 2), k = b & 3;
                        atomicAdd(&tcp[k * 1024 + 64 * j], (acc >> (8 * q)) & 0xFFu);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 8; i++) { bsL[i] = 0u; bsH[i] = 0u; }
            if (MASK) {
                // nibble j of this lane is read column p = (right side: m16 + 15 - j, left: m16 + j) - A of its side; only
                // read columns have counted
                u32 *const cmp = lds + d.off_cmp() + (p_strand ? 2 * L * 4 : 0) + (c_side ? L * 4 : 0);
#pragma unroll
                for (int half = 0; half < 2; half++) {
#pragma unroll
                    for (int sft = 0; sft < 8; sft++) {
                        u32 acc = 0u;
#pragma unroll
                        for (int i = 0; i < 8; i++) acc |= (((half ? b2H[i] : b2L[i]) >> sft) & 0x01010101u) << i;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int b = sft + 8 * q, j = 8 * half + (b >> 2), k = b & 3;
                            const u32 cnt = (acc >> (8 * q)) & 0xFFu;
                            const int pcol = (c_side ? c_m8 + 15 - j : c_m8 + j) - A;
                            if (cnt) atomicAdd(&cmp[pcol * 4 + k], cnt);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; i++) { b2L[i] = 0u; b2H[i] = 0u; }
            }
            bs_steps = 0;
        }
    };
#ifdef MDX_PHASE_CLK
    auto bs_flush = [&]() __attribute__((always_inline)) { MDX_PH_IN(5); bs_flush_impl(); MDX_PH_OUT(); };
#endif

    // Undo the optimistic increment of each queued byte that was not a plain match and, for read
    // columns, count what the byte really is (rare_column) — lane-parallel over the queued events.
    // Event word: [7:0] masked-quality flags (MASK) | [17:8] TC base of the record (bytes >> 8) | [23:18] lane |
    // [29:24] library | [30] the entry is a single deletion (then [10:8] = deleted bases g, [14:11] = first byte of
    // the lane that lies behind the deletion — on the right side: the bytes below it — and the TC base is recomputed)
    // | [31] reverse strand.
    // RS: a transition of a fused record, its old quality q known: the summary word by old quality (rescale.py:108-143:
    // "before" words of T>C / A>G, occurrences of (substitution, key, old quality) for C>T / G>A), and for a column with
    // a position key the new quality from the lookup table (_rescale_qual_read, rescale.py:228-246)
    auto rs_apply = [&](const u32x2 t, const u32 q) {
        if (q > 93u) return;
        const MdxTabArgs *kp = (const MdxTabArgs *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        const u32 idx = t.y & 0x7FFFu;
#ifndef MDX_RSABL_NOATOM
        atomicAdd(&kp->rs.subs_part[(size_t)blockIdx.x * rs_ncnt + idx + q], 1u);
#endif
        if (t.y & 0x8000u) {
            const u32 newq = l_lut[idx - 752u + q];
            if (kp->rs.patch) patch_put(kp->rs.patch, kp->rs.n_patch, kp->rs.patch_cap, kp->rs.patch_parts, newq != q, t.x, newq);
            else if (newq != q) kp->rs.qual_out[t.x] = (u8)newq;
        }
    };
    // the listed transitions, three per lane and round trip
    auto MDX_PH_IMPL(rsq_flush) = [&]() {
        u32 n = *rsq_cnt;
        if (n == 0u) return;
        n = n < (u32)MDX_FUSE_RSQ ? n : (u32)MDX_FUSE_RSQ;
        const MdxTabArgs *kp = (const MdxTabArgs *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        const u8 *__restrict__ qin = kp->qual;
        u32x2 t[MDX_FUSE_RSQ / 64];
        u32 q[MDX_FUSE_RSQ / 64];
#pragma unroll
        for (int k = 0; k < MDX_FUSE_RSQ / 64; k++) {
            const u32 i = (u32)lane + 64u * k;
            q[k] = 0xFFu;
#ifdef MDX_RSABL_NOQLD
            if (i < n) { t[k] = rsq[i]; q[k] = 30u + (t[k].x & 7u); }
#else
            if (i < n) { t[k] = rsq[i]; q[k] = qin[t[k].x]; }
#endif
        }
        // (every vector-memory operation of the wavefront so far is complete — the qualities just requested, and the stores
        // of the tile's quality copy, which the new qualities must not overtake)
        __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
#pragma unroll
        for (int k = 0; k < MDX_FUSE_RSQ / 64; k++)
            if ((u32)lane + 64u * k < n) rs_apply(t[k], q[k]);
        if (lane == 0) *rsq_cnt = 0u;
    };
#ifdef MDX_PHASE_CLK
    auto rsq_flush = [&]() { MDX_PH_IN(11); rsq_flush_impl(); MDX_PH_OUT(); };
#endif
    // RS: an event byte of a fused record (its staging entry `e`, number `ix`, is still in place: the runs of the fused
    // kernel drain the queue before they return).  The byte is read column p of its side.  A reference base in a left
    // column is counted here (the plain matches: the second TC table).  A transition — each column once: the left
    // side's [0, min(nq, L)), the right side's columns beyond — is listed for rs_apply (rsq), and the MR term of a
    // C>T / G>A one with a position key is noted in mrm.
    // (rs_transition: the column is read column p of its side, `kind` its transition on the read's own strand — 0 C>T,
    // 1 G>A (rescaled), 2 T>C, 3 A>G —, sq the index of the record's first aligned base in the SEQ / quality columns)
    auto rs_transition = [&](const uint4 &e, const int ix, const int rev, const int side, const int p, const int kind, const u32 sq) {
        const int nq = (int)(e.z & 0x7FFFu);
        const MdxTabArgs *kp = (const MdxTabArgs *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        const int qi = side ? nq - 1 - p : p;
        const int len5p = kp->rs.len5p, len3p = kp->rs.len3p;
        int pp = (rev ? nq - 1 - qi : qi) + 1;                         // _corr_this_base, rescale.py:49-79
        const int back = pp - nq - 1;
        pp = (!((e.w >> 20) & 1u) && pp >= -back) ? back : pp;
        const int k5 = pp <= len5p ? pp : 0, k3 = -pp <= len3p ? len5p - pp : 0;
        const int key = pp > 0 ? k5 : k3;
        const bool resc = kind < 2 && key;
        if (resc) atomicOr(&mrm[ix], 1ull << (kind * rs_npos + key));
        // "before" words of T>C / A>G, or the occurrences of (substitution, key, old quality)
        const int idx = kind >= 2 ? (kind == 2 ? 2 : 6) * 94 : 752 + (kind * rs_npos + key) * 94;
        const u32 slot = atomicAdd(rsq_cnt, 1u);
        u32x2 ent2;
        ent2.x = sq + (u32)qi; ent2.y = (u32)idx | (resc ? 0x8000u : 0u);
        if (slot < (u32)MDX_FUSE_RSQ) rsq[slot] = ent2;
        else {
            // (a tile with more transitions than the list holds: at once — behind the stores of the tile's quality copy)
            const u32 q = (u32)kp->qual[ent2.x];
            __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
            rs_apply(ent2, q);
        }
    };
    auto rs_event = [&](const uint4 &e, const int ix, const int rev, const int side, const int p, const u32 sb, const u32 rb) {
        const int nq = (int)(e.z & 0x7FFFu);
        if (!side && rb < 0x80u) {
            const u32 k = (rb >> 1) & 3u;           // A,C,T,G
            u32 b = k ^ (k >> 1);                   // A,C,G,T
            if (rev) b = 3u - b;
            bcA += b == 0u; bcC += b == 1u; bcG += b == 2u; bcT += b == 3u;
        }
        if (side && p >= nq - L) return;
        const u32 pr = sb | (rb << 8);
        // stored pair -> transition of the read's own strand: 0 C>T, 1 G>A (rescaled), 2 T>C, 3 A>G
        const int kind = (int)(pr == ('T' | 'C' << 8)) * (1 + rev) + (int)(pr == ('A' | 'G' << 8)) * (2 - rev) +
                         (int)(pr == ('C' | 'T' << 8)) * (3 + rev) + (int)(pr == ('G' | 'A' << 8)) * (4 - rev) - 1;
        if (kind < 0) return;
        rs_transition(e, ix, rev, side, p, kind, e.y);
    };
    // PK: the same for a pair of 4-bit codes (A, C, T, G = 1, 2, 4, 8) — the reference bases of a fused record's columns are
    // counted by the steps themselves (count16), so only the transitions are looked at here; an entry's second word is its
    // SEQ window offset less its reference window offset (see the staging entries)
    // (p: the column of its side, pc: its query index from that end — less than p behind a deletion of g bases)
    auto rs_event4 = [&](const uint4 &e, const int ix, const int rev, const int side, const int p, const int pc, const int g, const u32 sn, const u32 rn) {
        const int nq = (int)(e.z & 0x7FFFu);
        if (side && p >= nq + g - L) return;
        const u32 pr = sn | (rn << 4);
        const int kind = (int)(pr == 0x24u) * (1 + rev) + (int)(pr == 0x81u) * (2 - rev) +
                         (int)(pr == 0x42u) * (3 + rev) + (int)(pr == 0x18u) * (4 - rev) - 1;
        if (kind < 0) return;
        rs_transition(e, ix, rev, side, pc, kind, e.y + e.x - pk_dso);
    };
    // the MR sum of a record from its word of mrm: the terms in column order — 5' keys upwards, then 3' keys downwards
    auto mr_of = [&](const u64 m) -> double {
        double mr = 0.0;
        const u32 m0 = (u32)(m & ((1ull << rs_npos) - 1ull)), m1 = (u32)(m >> rs_npos);
        const u32 c = m0 | m1;
        // (the terms one after the other, each read behind the addition before it: four reads in flight together —
        // more instructions — measured 0.7 % slower, round 5)
        u32 c5 = c & (u32)((2ull << a.rs.len5p) - 2ull);
        while (c5) {
            const int k = __ffs((int)c5) - 1;
            c5 &= c5 - 1;
            mr += l_term[((m0 >> k) & 1u) ? k : rs_npos + k];
        }
        u32 c3 = (u32)((u64)c >> (a.rs.len5p + 1));
        while (c3) {
            const int j = 31 - __clz((int)c3);
            c3 &= ~(1u << j);
            const int k = a.rs.len5p + 1 + j;
            mr += l_term[((m0 >> k) & 1u) ? k : rs_npos + k];
        }
        return mr;
    };
    // (PK: the events from `lo` on — behind a run lo = qcount % 64: full passes only, the rest waits for company)
    auto MDX_PH_IMPL(drain_all) = [&](const int lo = 0) {
        if (PK) {
#ifdef MDX_ABL_NODRAIN
            qcount = lo;
            return;
#endif
            // the packed kernel's events, 64 lanes at a time; only a nibble that holds a base was counted
            const int n = qcount;
#pragma unroll 1
            for (int base = lo; base < n; base += 64) {
                const int i = base + lane;
                if (i < n) {
                    const uint4 q = qQ[i];          // read nibbles (x, y), reference nibbles (z, w)
                    const u32 w = qE[i];
                    const int ln = (int)w & 63, side = (int)(w >> 6) & 1, m16 = (int)((w >> 7) & 15u) << 4, rev = (int)(w >> 11) & 1;
                    const bool del = (w >> 12) & 1u;
                    const int g = del ? (int)(w >> 13) & 7 : 0, bnd = (int)(w >> 16) & 31;
                    const u32x2 em = emtab[ln];
                    // (RS: an event of a fused record carries the place of its staging entry, [27:21], and bit 28)
                    const bool rsev = RS && ((w >> 28) & 1u);
                    uint4 rent = make_uint4(0u, 0u, 0u, 0u);
                    if (RS && rsev) rent = stg[(w >> 21) & 0x7Fu];
                    const u64 s64 = (u64)q.x | ((u64)q.y << 32), r64 = (u64)q.z | ((u64)q.w << 32);
                    u64 x = (s64 ^ r64) & ((u64)em.x | ((u64)em.y << 32));      // the read columns that differ
                    const int b_mis = d.off_mis() + (rev ? 2 * L * 25 : 0), b_cmp = d.off_cmp() + (rev ? 2 * L * 4 : 0);
                    // usually exactly one nibble differs: the lowest one (all lanes busy), again while some lane has another
                    // (two copies of the loop: a pass without an event of a deletion step — most of them — carries none of the
                    // by-position arithmetic)
                    auto nibbles = [&](auto del_tag) {
                    constexpr bool DEL = decltype(del_tag)::value;
                    while (x) {
                        const int sh = (__ffsll((long long)x) - 1) & ~3, jb = sh >> 2;
                        x &= ~(15ull << sh);
                        const u32 t = pktab[((u32)(r64 >> sh) & 15u) << 4 | ((u32)(s64 >> sh) & 15u)];
                        const int rc = (int)(t & 7u), sc = (int)((t >> 3) & 7u), col = (int)(t >> 6);
                        const int p = (side ? m16 + 15 - jb : m16 + jb) - A;
                        // a column behind the deletion of its record was counted, optimistically, as a match in MIS[p] and
                        // CMP[p - g] (its query index) instead of the lane's counters
                        const bool direct = DEL && del && (side ? jb < bnd : jb >= bnd);
                        const int pc = direct ? p - g : p;
                        const int sp = (side ? L : 0) + p, spc = (side ? L : 0) + pc;
                        if (rc < 4) {
                            if (direct) {
                                atomicAdd(&lds[b_mis + __mul24(sp, 25) + rc], 0xFFFFFFFFu);
                                atomicAdd(&lds[b_cmp + spc * 4 + rc], 0xFFFFFFFFu);
                            } else {
                                atomicAdd(&lds[d.off_tc() + (rc << 10) + 64 * jb + ln], 0xFFFFFFFFu);  // -1
                            }
                        }
                        // what the column really is (rare_column): the read base, and a substitution / indel
                        if (sc < 4) atomicAdd(&lds[b_cmp + spc * 4 + sc], 1u);
                        if (col != 31) atomicAdd(&lds[b_mis + __mul24(sp, 25) + col], 1u);
#ifndef MDX_RSABL_NOEV
                        if (RS && rsev) rs_event4(rent, (int)((w >> 21) & 0x7Fu), rev, side, p, pc, g, (u32)(s64 >> sh) & 15u, (u32)(r64 >> sh) & 15u);
#endif
                    }
                    };
                    if (__ballot(del)) nibbles(std::true_type{});
                    else nibbles(std::false_type{});
                }
            }
            qcount = lo;
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
            return;
        }
#pragma unroll 1
        for (int qb = 0; qb < qcount; qb += 64)
        if (qb + lane < qcount) {
            const u32x2 es = qS[qb + lane], er = qR[qb + lane];
            const u32 w = qW[qb + lane];
            // (RS: an event of a fused record is known by its TC table — the second one)
            const bool rsev = RS && !((w >> 30) & 1u) && ((w & 0x3FF00u) >> 2) >= (u32)a.rs.tcb_off;
            uint4 rent = make_uint4(0u, 0u, 0u, 0u);
            if (RS && rsev) rent = stg[w & 0x7Fu];
            const int ln = (int)(w >> 18) & 63;
            const int rev = (int)(w >> 31);
            const int lb = __mul24((int)((w >> 24) & 0x3Fu), d.w_lib);
            const bool del = (w >> 30) & 1u;
            const int g = del ? (int)(w >> 8) & 7 : 0, bnd = (int)(w >> 11) & 15;
            const int tcw = del ? lb + d.off_tc() + rev * 4 * 512 : (int)((w & 0x3FF00u) >> 2);   // first word of TC[library][strand]
            int ll = ln;  // lane within its slot (R <= 4)
            if (ll >= d.G) ll -= d.G;
            if (ll >= d.G) ll -= d.G;
            if (ll >= d.G) ll -= d.G;
            const int side = ll >= d.nl8;
            const int m8 = 8 * (ll - side * d.nl8);
            u64 vm, em;
            lane_masks(d, side, m8, vm, em);
            const u64 s64 = (u64)es.x | ((u64)es.y << 32), r64 = (u64)er.x | ((u64)er.y << 32);
            u64 x = (((s64 ^ r64) & em) | (r64 & 0x8080808080808080ull)) & vm;
            if (MASK) x |= spread_bits(w & 0xFFu);
            const int b_mis = lb + d.off_mis() + (rev ? 2 * L * 25 : 0), b_cmp = lb + d.off_cmp() + (rev ? 2 * L * 4 : 0);
            // usually exactly one byte of the lane differs: handle the lowest such byte (all lanes busy),
            // repeat only while some lane has another
            while (x) {
                const int jb = (__ffsll((long long)x) - 1) >> 3;
                const int sh = 8 * jb;
                x &= ~(0xFFull << sh);
                const u32 rb = (u32)(r64 >> sh) & 0xFFu, sb = (u32)(s64 >> sh) & 0xFFu;
                const int p = (side ? m8 + 7 - jb : m8 + jb) - A;
                // a column behind the deletion of its record was counted, optimistically, as a match in MIS[p] and
                // CMP[p - g] (its query index) instead of TC
                const bool direct = del && (side ? jb < bnd : jb >= bnd);
                const int pc = direct ? p - g : p;
                if (direct) {
                    bump_n<USE_LDS>(lds, raw, b_mis + __mul24((side ? L : 0) + p, 25) + (int)((rb >> 1) & 3u), 0xFFFFFFFFu);
                    bump_n<USE_LDS>(lds, raw, b_cmp + ((side ? L : 0) + pc) * 4 + (int)((rb >> 1) & 3u), 0xFFFFFFFFu);
                } else {
                    bump_n<USE_LDS>(lds, raw, tcw + (int)(((rb >> 1) & 3u) << 9) + 64 * jb + ln, 0xFFFFFFFFu);  // -1
                }
                if ((em >> sh) & 1ull) {
                    rare_column<USE_LDS>(lds, raw, b_mis, b_cmp, L, side, p, pc, sb, (int)(i8)rb, MASK && ((w >> jb) & 1u));
#ifndef MDX_RSABL_NOEV
                    if (RS && rsev) rs_event(rent, (int)(w & 0x7Fu), rev, side, p, sb, rb);
#endif
                }
            }
        }
        qcount = 0;
        // nothing LDS-returning may be pending when control rejoins the hot loop
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    };
#ifdef MDX_PHASE_CLK
    auto drain_all = [&](const int lo = 0) { MDX_PH_IN(4); drain_all_impl(lo); MDX_PH_OUT(); };
#endif

    // One step of the fast path: R records, one per slot of G lanes (see MdxDims).  Stage = the loaded bytes
    // of a step and its record words.
    struct Stage { u32x3 s12, r12, q12; u32 ro, so, pk, aux; int lim, k; bool valid; };
    // (declared here, in front of the runs: the prefetch of the next tile's second round trip is issued from inside a run)
    // (ML: the records of the pool's library, places [rec_lo, rec_lo + n_rec) of the bucketed columns; ml_lib = that library,
    // counted from the launch's first one — set where the pool's plan is read)
    u32 n_rec = (u32)a.n_reads, rec_lo = 0u;
    int ml_lib = 0;
    const u32 T = FAST ? (u32)mdx_tile_records(d, pfl) : 64u;
    const karg_p ka = (karg_p)__builtin_amdgcn_kernarg_segment_ptr();
    // pfl — the phase-1 columns of the wavefront's NEXT tile are requested into its prefetch area by LDS-DMA loads
    // (pfl_dma_cols) at the head of a tile, as soon as the current tile's have been read out of it; the next tile's second
    // round trip (pfl_dma_rt2: it needs those columns) goes out from inside the current tile's first run, behind that run's
    // first loads — by then the columns have landed — and lands under the run.  A tile's phase 1 then starts with everything
    // it needs in the LDS, and the one round trip a wavefront waits for per tile is that of its run's first loads.
    // Area: [flag | library | tid | pos | tlen | cigar_off | seq_off | op 0 | op 1 | op 2 | contig start | contig end][lane]; a
    // record's end offsets are its right neighbour's start offsets (a tile has 63 records at most: lane 63 brings the last end).
    auto pfl_cols = [&](const u32 tile) {
        karg_p kp = ka;
        asm volatile("" : "+s"(kp));
        const u32 i_ = tile * T + (u32)lane;
        // (lanes past the batch's end: its last record, and the offsets' last entry — their records are not valid)
        const u32 i1 = (i_ < n_rec - 1u ? i_ : n_rec - 1u) + (ML ? rec_lo : 0u), i2 = (i_ < n_rec ? i_ : n_rec) + (ML ? rec_lo : 0u);
        pfl_dma_cols<!ML>(pfl_a, i1 * 2u, i1 * 4u, i2 * 4u, kp->flag, kp->lib, kp->tid, kp->pos, kp->tlen, kp->cigar_off, kp->seq_off);
    };
    auto pfl_rt2 = [&](const u32 tile) {
        karg_p kp = ka;
        asm volatile("" : "+s"(kp));
        const u32 tb = tile * T, rh = tb + T < n_rec ? tb + T : n_rec;
        const lds_u1 *const C = (const lds_u1 *)(size_t)(pfl_a + 4u * (u32)lane);
        const u32 fl_ = tb + (u32)lane < rh ? C[0] : 0x4u;
        const int lib_ = ML ? a.lib_lo + ml_lib : (int)C[64], tid_ = (int)C[128];
        const u32 co0_ = C[320], co1_ = C[321];
        bool kept_ = (fl_ & 0xF04u) == 0;
        if (!ML && lib_ < a.nlib_total && (lib_ < a.lib_lo || lib_ >= a.lib_lo + d.nlib)) kept_ = false;
        const u32 cn_ = co1_ - co0_;
        const bool cand_ = kept_ && cn_ - 1u < 3u && tid_ >= 0 && tid_ < a.n_contig && lib_ < a.nlib_total;
        if (cand_) pfl_dma_rt2(pfl_a + 7u * 256u, co0_ * 4u, (co0_ + (cn_ >= 2u ? 1u : 0u)) * 4u, (co0_ + (cn_ >= 3u ? 2u : 0u)) * 4u,
                               (u32)tid_ * 8u, (u32)tid_ * 8u + 8u, kp->cigar, kp->contig_off);
    };
    // (pfl_due: the tile whose second round trip the next run is to request behind its first loads — 0xFFFFFFFE: nothing due)
    u32 pfl_due = 0xFFFFFFFEu;
    // Three kinds of steps (one instantiation each):
    //   STEP_C  complete records: every task present, static byte masks;
    //   STEP_P  a column range per side: short records and contig edges ([-flank, min(nq, L))), gapped records whose
    //           first / last match run is counted here and the rest by the CIGAR walk ([-A, run)), records gapped by
    //           N / P operations only ([-A, min(nq, L)): both windows are an ungapped record's);
    //   STEP_G  records with one short insertion or deletion ([H][S] M {I|D} M [S][H], D_ONE), the whole record in one
    //           entry.  Lanes work in *column* space.  On the left side column c of the string that carries the gap
    //           (reference for an insertion, read for a deletion) is byte c of its window in front of the gap
    //           (c < u, the first match run), the gap symbol for u <= c < u + g, and byte c - g behind it: a lane
    //           whose first column lies at or behind the gap loads its window g bytes lower (fill()), the lane that
    //           straddles the gap shifts its own bytes up by g.  The right side mirrors this from aend.
    //           An insertion keeps one position per column for both tables: the ordinary optimistic step.  Behind a
    //           deletion the composition position (query index) is g less than the misincorporation position
    //           (column): those bytes are counted, still optimistically, straight into MIS[c][base] and CMP[c - g][base]
    //           (direct8) instead of TC, and their events say so (drain_all).
    // (full_tag: every slot of the step holds a record — all steps of a run but its last one)
    // (qm_tag: the step looks at the qualities — QM kernels only; the run of the records that cannot be masked does not)
    auto count = [&](const Stage &st, auto kind_tag, auto full_tag, auto qm_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr bool QM = MASK && decltype(qm_tag)::value;
        constexpr int KIND = decltype(kind_tag)::value;
        // slots past the last record of the tile: no increments, no events
        const bool act = FULL || lane < st.lim;
        u32 s_lo = __builtin_amdgcn_alignbyte(st.s12.y, st.s12.x, st.so), s_hi = __builtin_amdgcn_alignbyte(st.s12.z, st.s12.y, st.so);
        u32 r_lo = __builtin_amdgcn_alignbyte(st.r12.y, st.r12.x, st.ro), r_hi = __builtin_amdgcn_alignbyte(st.r12.z, st.r12.y, st.ro);
        // (RS: not unless the queue takes what this step raises — see rs_ovf; a complete step's events are known before it
        // has touched a table, the other kinds make room for a whole step's worth)
        if (RS && KIND == STEP_C) {
            const u32 x0 = ((s_lo ^ r_lo) & c_em_lo) | (r_lo & c_hivm_lo) | ((s_hi ^ r_hi) & c_em_hi) | (r_hi & c_hivm_hi);
            if (qcount + __popcll(__ballot(act && x0 != 0u)) > QCAP) { rs_ovf = true; rs_kredo = st.k; return; }
        } else if (RS && qcount > QCAP - 64) { rs_ovf = true; rs_kredo = st.k; return; }
        u32 emvm_lo = c_em_lo, emvm_hi = c_em_hi;
        const u32 base_b = tc_base(st.pk, c_lane4);
        u32 q_lo = 0, q_hi = 0, qo_ = 0;
        u32x3 q12_ = {0u, 0u, 0u};
        bool q_ready = false;
        auto qwin = [&]() {
            if (!q_ready) {
                q_lo = __builtin_amdgcn_alignbyte(q12_.y, q12_.x, qo_);
                q_hi = __builtin_amdgcn_alignbyte(q12_.z, q12_.y, qo_);
                q_ready = true;
            }
        };
        if (QM) {
#if MDX_QPREFETCH
            // (requested by fill(), with the other two windows)
            q12_ = st.q12; qo_ = st.so - c_so + c_qo;
#else
            // the quality window is requested here, at the start of the step that uses it, not a step or two ahead like
            // the other two: three more registers per step in flight are more than the kernel has (the hot loop
            // spilled), and what a step does before it needs the qualities covers part of the latency.  Records that
            // cannot be masked — no qualities, or the caller's hint — read one fixed line instead of their window.
            const u32 qo = st.so - c_so + c_qo;
            const u32x3 q12 = *(const u32x3 *)(qualW + ((st.pk & 0x40000000u) ? (qo & ~3u) : 0u));
            q12_ = q12; qo_ = qo;       // (funnelled out where the qualities are first needed: the wait sits there)
#endif
        }
        // event word of this lane (without lane and quality bits; RS: with the staging index)
        u32 evw = st.pk & (RS ? 0xBF03FF7Fu : 0xBF03FF00u);
        if (KIND == STEP_C) {
            // optimistic: count every byte as a plain match (the base class of the reference
            // byte selects the plane of TC) ...
            tc_bump8_all(r_lo, r_hi, base_b, act ? 1u : 0u);
        } else {
            // byte thresholds of this lane, computed by fill() (times eight: offsets into the byte-mask table)
            const u32 aux = st.aux;
            u32 dyn_lo, dyn_hi, tcd_lo, tcd_hi;
            if (KIND == STEP_P) {
                // tasks = the bytes below the boundary (left side) / from it on (right side)
                const u64 M = *(const u64 *)((const u8 *)ltab + aux);
                const u32 sm = 0u - (u32)c_side;                         // all ones on the right side
                dyn_lo = ((u32)M ^ sm) & c_vm_lo; dyn_hi = ((u32)(M >> 32) ^ sm) & c_vm_hi;
                tcd_lo = tcd_hi = 0u;
            } else {
                // [ta, tb) = the gap, tasks end at byte tl (left side; on the right side they start there)
                const u64 X = *(const u64 *)((const u8 *)ltab + (aux & 0x7Fu)), Y = *(const u64 *)((const u8 *)ltab + ((aux >> 7) & 0x7Fu)),
                          T = *(const u64 *)((const u8 *)ltab + ((aux >> 14) & 0x7Fu));
                const u32 sm = 0u - (u32)c_side;                         // all ones on the right side
                dyn_lo = ((u32)T ^ sm) & c_vm_lo; dyn_hi = ((u32)(T >> 32) ^ sm) & c_vm_hi;
                // the string that carries the gap (reference for an insertion, read for a deletion): its bytes in front
                // of the gap as loaded (a lane at or behind the gap holds none), the gap symbol, its bytes behind the
                // gap moved by g within the lane that straddles it (shb = 8 g there, 0 elsewhere)
                const u32 shb = ((aux >> 21) & 7u) << 3, shr = shb & sm, shl = shb & ~sm;
                const u64 star = KIND == STEP_GD ? ((u64)s_lo | ((u64)s_hi << 32)) : ((u64)r_lo | ((u64)r_hi << 32));
                const u64 low = star >> shr, high = star << shl;
                const u32 gapc = KIND == STEP_GD ? 0x2D2D2D2Du : 0x84848484u;
                const u32 X_lo = (u32)X, X_hi = (u32)(X >> 32), Y_lo = (u32)Y, Y_hi = (u32)(Y >> 32);
                u32 m_lo = (X_lo & (u32)low) | (~X_lo & gapc), m_hi = (X_hi & (u32)(low >> 32)) | (~X_hi & gapc);
                m_lo = (Y_lo & m_lo) | (~Y_lo & (u32)high); m_hi = (Y_hi & m_hi) | (~Y_hi & (u32)(high >> 32));
                tcd_lo = tcd_hi = 0u;
                if (KIND == STEP_GD) {
                    s_lo = m_lo; s_hi = m_hi;
                    if (QM) {
                        // the qualities travel with the read; a deleted column has none (never masked, align.py:67)
                        qwin();
                        const u64 q64 = (u64)q_lo | ((u64)q_hi << 32), ql = q64 >> shr, qh = q64 << shl;
                        q_lo = (X_lo & (u32)ql) | ~X_lo; q_hi = (X_hi & (u32)(ql >> 32)) | ~X_hi;
                        q_lo = (Y_lo & q_lo) | (~Y_lo & (u32)qh); q_hi = (Y_hi & q_hi) | (~Y_hi & (u32)(qh >> 32));
                    }
                    // bytes behind the deletion (left: from tb on, right: below ta): MIS[column][base] and
                    // CMP[column - g][base], one increment each like tc_bump8 (bytes in position order: reversed on the
                    // right side), instead of TC
                    const u32 beh_lo = (X_lo & sm) | (~Y_lo & ~sm), beh_hi = (X_hi & sm) | (~Y_hi & ~sm);
                    const u32 dm_lo = dyn_lo & beh_lo, dm_hi = dyn_hi & beh_hi;
                    tcd_lo = ~beh_lo; tcd_hi = ~beh_hi;      // the bytes TC counts (the others: by position, below)
                    evw = (st.pk & 0xBF000000u) | 0x40000000u | (((aux >> 24) & 0x7Fu) << 8);
                    if (__ballot((dm_lo | dm_hi) != 0u)) {
                        const int g = (int)((aux >> 24) & 7u);
                        const int lbw = __mul24((int)((st.pk >> 24) & 0x3Fu), d.w_lib), rev = (int)(st.pk >> 31);
                        const int row = __mul24(rev * 2 + c_side, L) + c_m8 - A;        // row of the lane's lowest position
                        const u32 p_lo = c_side ? 0x04050607u : 0x03020100u, p_hi = c_side ? 0x00010203u : 0x07060504u;
                        const u32 rr_lo = __builtin_amdgcn_perm(r_hi, r_lo, p_lo), rr_hi = __builtin_amdgcn_perm(r_hi, r_lo, p_hi);
                        const u32 mm_lo = __builtin_amdgcn_perm(dm_hi, dm_lo, p_lo), mm_hi = __builtin_amdgcn_perm(dm_hi, dm_lo, p_hi);
                        direct8<100>(lds, rr_lo, rr_hi, (u32)(4 * (lbw + d.off_mis() + __mul24(row, 25))), mm_lo, mm_hi);
                        direct8<16>(lds, rr_lo, rr_hi, (u32)(4 * (lbw + d.off_cmp() + (row - g) * 4)), mm_lo, mm_hi);
                    }
                } else {
                    r_lo = m_lo; r_hi = m_hi;
                }
            }
            // bytes that are not tasks of this record: zero in both strings (a pair that raises no event and that
            // drain_all passes over)
            s_lo &= dyn_lo; s_hi &= dyn_hi; r_lo &= dyn_lo; r_hi &= dyn_hi;
            if (QM) { emvm_lo &= dyn_lo; emvm_hi &= dyn_hi; }   // (the quality bytes are not zeroed)
            // all eight bytes with one increment, like a complete record's: a zeroed byte lands in plane A, and DMP knows
            // how many of those there are per window byte (phase 1)
            if (KIND == STEP_GD) tc_bump8_all(r_lo & tcd_lo, r_hi & tcd_hi, base_b, act ? 1u : 0u);
            else tc_bump8_all(r_lo, r_hi, base_b, act ? 1u : 0u);
        }
        // MASK: bit 7 of the bytes whose quality is below --min-basequal (align.py:65-71)
        u32 lowq_lo = 0, lowq_hi = 0;
        if (QM) {
            qwin();
            const u32 minq4 = (st.pk & 0x40000000u) ? (u32)a.minqual * 0x01010101u : 0u;
            lowq_lo = ~((q_lo | 0x80808080u) - minq4) & 0x80808080u;
            lowq_hi = ~((q_hi | 0x80808080u) - minq4) & 0x80808080u;
        }
        // x: per byte, zero iff the byte is a plain match (read == reference, reference is
        // A/C/G/T); flank bytes only test the reference byte; bytes that are not tasks are zero
        u32 x_lo = ((s_lo ^ r_lo) & emvm_lo) | (r_lo & c_hivm_lo);
        u32 x_hi = ((s_hi ^ r_hi) & emvm_hi) | (r_hi & c_hivm_hi);
        u32 mq_lo = 0, mq_hi = 0;
        if (QM) {
            // the masked columns of this lane: bit 7 of the byte
            mq_lo = lowq_lo & emvm_lo;
            mq_hi = lowq_hi & emvm_hi;
            if (USE_LDS && KIND != STEP_GI && KIND != STEP_GD) {
                // a masked column counts its read base (CMP) and nothing else (align.py:65-71 turns both symbols into
                // N): corrected here — the optimistic TC increment undone, CMP bumped — one byte per iteration, as
                // many iterations as the fullest lane has masked bytes; such bytes raise no event
                // (worth it when many lanes hold masked bytes; a few of them are cheaper as events)
                u64 m = act ? ((u64)mq_lo | ((u64)mq_hi << 32)) : 0ull;
                const bool many = __popcll(__ballot(m != 0)) > 16;
                if (!many) m = 0;
                if (m) {
                    const u32 t_lo = mq_lo >> 7, t_hi = mq_hi >> 7;
                    x_lo &= ~((t_lo << 8) - t_lo); x_hi &= ~((t_hi << 8) - t_hi);
                    // (all in 32 bits, and the lane's first position derived here from the SEQ offset it holds anyway: as a
                    // loop invariant of its own it was a 64-bit register pair spilled across the hot loop)
                    u32 so_l = c_so;
                    asm volatile("" : "+v"(so_l));
                    const u32 right = c_cm != 0u ? 1u : 0u;
                    // position of byte 0 (left side: c_m8 - A, rising) or of byte 7 (right side: c_m8 - A, byte 7 - j rising)
                    const u32 p0 = right ? ph_seq - 8u - so_l : so_l - ph_seq;        // c_m8 - A on either side
                    const u32 lbw = __umul24((st.pk >> 24) & 0x3Fu, (u32)d.w_lib), rev = st.pk >> 31;
                    const u32 b_cmp = lbw + (u32)d.off_cmp() + __umul24(rev * 2u + right, (u32)(L * 4));
                    const u64 s64 = (u64)s_lo | ((u64)s_hi << 32), r64 = (u64)r_lo | ((u64)r_hi << 32);
#pragma unroll 1
                    while (m) {
                        const u32 sh = (u32)(__ffsll((long long)m) - 1) & ~7u, jb = sh >> 3;
                        m &= m - 1;
                        const u32 sb = (u32)(s64 >> sh) & 0xFFu, rb = (u32)(r64 >> sh) & 0xFFu;
                        atomicAdd(&lds[(base_b >> 2) + (((rb >> 1) & 3u) << 9) + 64u * jb], 0xFFFFFFFFu);   // -1
                        const int sc = classify_read(sb);
                        const u32 pos = right ? p0 + 7u - jb : p0 + jb;      // = (c_side ? c_m8 + 7 - jb : c_m8 + jb) - A
                        if (sc < 4) atomicAdd(&lds[b_cmp + pos * 4u + (u32)sc], 1u);
                    }
                    // (a matching pair in the event copy: drain_all must not look at these bytes again)
                    const u32 mb_lo = (t_lo << 8) - t_lo, mb_hi = (t_hi << 8) - t_hi;
                    s_lo &= ~mb_lo; s_hi &= ~mb_hi; r_lo &= ~mb_lo; r_hi &= ~mb_hi;
                }
                if (many) { mq_lo = 0; mq_hi = 0; }
            }
            x_lo |= mq_lo; x_hi |= mq_hi;
        }
        // ... and queue the lanes holding a byte that is not one (drain_all corrects them)
        u32 xx = x_lo | x_hi;
        if (KIND == STEP_C && !FULL) xx = act ? xx : 0u;   // (the other kinds mask by dyn already)
        const bool ev = xx != 0;
        const u64 mm = __ballot(ev);
        if (mm) {
            const int n = __popcll(mm);
            if (!RS && qcount + n > QCAP) drain_all();
            if (ev) {
                const int slot = mbcnt64(mm, qcount);
                u32x2 es, er;
                es.x = s_lo; es.y = s_hi; er.x = r_lo; er.y = r_hi;
                qS[slot] = es;
                qR[slot] = er;
                u32 w = evw | (c_lane4 << 16);
                if (QM) w |= gather_bits(mq_lo) | (gather_bits(mq_hi) << 4);
                qW[slot] = w;
            }
            qcount += n;
        }
    };

    // ---- the fast path's runs (used by the tile loop for the complete records and, behind it, for the lists)
        const int R = d.R, G = d.G;
        // A run = the steps of the nrec staged records from entry e0 on, all of one kind (count())
        // (n_plus >= 0 — PK, complete runs of the tile loop: the entries are sorted by strand, the first n_plus of them forward)
        auto run = [&](const int e0, const int nrec, auto kind_tag, auto qm_tag, const int n_plus = -1) {
            // (kind | 8: the partial records of a tile — ungapped: no n0 - nq in their entries)
            constexpr int KIND = decltype(kind_tag)::value & 7;
            constexpr bool TILE_P = (decltype(kind_tag)::value & 8) != 0;
            constexpr bool QM = MASK && decltype(qm_tag)::value;
            const int nsteps = (nrec + R - 1) / R;
            int kf = 0;
            if constexpr (PK) {
                // ---- the packed kernel's run: the entries [e0, e0 + nrec) are sorted by strand, the first n_plus of them
                // forward.  Step k takes the forward entries H k .. H k + H - 1 into its slots [0, H) and the reverse entries
                // with the same numbers into [H, 2 H); a slot past its strand's last entry shadows entry e0 and is masked out.
                const int H = d.H4;
                const int nP_ = n_plus, nM_ = nrec - n_plus;
                const int nsteps4 = ((nP_ > nM_ ? nP_ : nM_) + H - 1) / H;
                const int nfull = (nP_ < nM_ ? nP_ : nM_) / H;       // steps in which every slot holds a record
                if (bs_steps + nsteps4 > 255) bs_flush();
                bs_steps += nsteps4;
                const int base_l = e0 + (p_strand ? nP_ : 0) + c_slot, lim_l = (p_strand ? nM_ : nP_) - c_slot;
                // (FASTIDX — complete steps outside the fused kernels, which want nothing of an entry but its first three words: the
                // entry's LDS address is min(first + 16 H k, last entry of the lane's strand) — the dynamic LDS starts at address 0.
                // A slot past its strand's last entry reads that entry, or the run's first when the strand has none, and is
                // masked out by the step (actm))
                constexpr bool FIDX = MDX_PK_FASTIDX && KIND == STEP_C && !RS;
                // (FIDP — the partial steps of the kernels with the registers for it: the same addressing, the slot's being
                // past its strand's last entry from one compare of the address; and the lane's nibble mask, a table lookup by a
                // value of the entry, requested by the fill — four steps ahead of the step that wants it — instead of by the step,
                // which waited for it)
                constexpr bool FIDP = MDX_PK_FASTIDX && MDX_PK_FASTP && KIND == STEP_P && !RS && !MASK;
                const u32 stg_a = (u32)(size_t)(lds_u4 *)stg;
                const int last_l = e0 + (p_strand ? nP_ : 0) + (p_strand ? nM_ : nP_) - 1;
                const u32 ent_a0 = stg_a + 16u * (u32)base_l, ent_cap = stg_a + 16u * (u32)(last_l > e0 ? last_l : e0);
                const int H16 = 16 * H;
                const u32 ent_capr = stg_a + 16u * (u32)last_l;        // (unclamped: below the lane's first address when its strand has no entry)
                // (one <3 x i32> load per operand: a struct of three words is taken apart and put together again as the
                // vectorizer likes — two overlapping dwordx2 loads at times)
                struct St16 { u32v3 s, r; u32 sa, ra, pk, aux, aux2; int k; bool valid; u64 mk; };
                // (MDX_PK_ENT_AHEAD: the staging entry of a step is read from the LDS one fill ahead — LDS operations return in
                // order, so a fill that reads its own entry waits, in front of its window loads, for that read and for the
                // event writes of the step just counted)
                auto ent_index = [&](const int kk, bool &act) -> int {
                    const int k = kk < nsteps4 ? kk : nsteps4 - 1;
                    act = true;
                    int idx = base_l + H * k;
                    if (k >= nfull) {               // (wave-uniform: the last steps of a run only)
                        act = H * k < lim_l;
                        idx = act ? idx : e0;
                    }
                    return idx;
                };
#if MDX_PK_ENT_AHEAD
                uint4 ent_next = make_uint4(0u, 0u, 0u, 0u);
#endif
                auto fill16 = [&](St16 &st) {
                    st.valid = kf < nsteps4;
                    const int k = st.valid ? kf : nsteps4 - 1;
                    // (the slot holds a record iff H k + slot < the entries of its strand)
                    bool act = true;
                    int idx = 0;
                    u32 ref_copy = 0u;
                    if (!(FIDX || FIDP) || MDX_PK_ENT_AHEAD) idx = ent_index(kf, act);
                    kf++;
                    st.k = k;
#if MDX_PK_ENT_AHEAD
                    const uint4 ent = ent_next;
#else
                    uint4 ent;
                    if constexpr (FIDP) {
                        const u32 ad = ent_a0 + (u32)(H16 * k);
                        act = ad <= ent_capr;
                        const u32x4 e_ = *(const lds_u4 *)(size_t)(ad < ent_cap ? ad : ent_cap);
                        ent = make_uint4(e_.x, e_.y, e_.z, e_.w);
                        // (a tile's own partial entry: bit 31 of w = the copy of the reference, bits 1 and 0 are zero — the
                        // lane's byte offset takes all three through the bit-field insert that was an and)
                        if (TILE_P) ref_copy = e_.w & 0x80000003u;
                    } else
                    if constexpr (FIDX) {
                        const u32 ad = ent_a0 + (u32)(H16 * k);
                        const u32x4 e_ = *(const lds_u4 *)(size_t)(ad < ent_cap ? ad : ent_cap);
                        ent = make_uint4(e_.x, e_.y, e_.z, e_.w);
                        // (w of a complete record's entry: the byte offset of the copy of the reference its window is read from —
                        // 0, or 2 GiB: MdxTabArgs::ref2)
                        ref_copy = e_.w;
                    } else ent = stg[idx];
#endif
                    const u32 t = ent.z & c_cm;
                    u32 ro = ent.x + c_ro + t;
                    u32 so = ro + ent.y;
                    st.aux = 0u; st.aux2 = 0u;
                    if (RS && KIND == STEP_C) {
                        // a fused record (bit 18 of its entry): its events carry the entry's place, and the reference bases of
                        // its columns — each column once: the left side's, and the right side's beyond them, p < nq - L, the
                        // nibbles from t on — are counted by the step (subs[nt_ref], rescale.py:142-143); aux = the offset of
                        // the nibble-mask table's entry t (a record that is not fused: no nibble on either side)
                        const bool fz = act && ((ent.w >> 18) & 1u);
                        int t = c_side ? c_m8 + 16 - A + L - (int)(ent.z & 0x7FFFu) : 16;
                        t = t < 0 ? 0 : (t > 16 ? 16 : t);
                        st.aux = fz ? (((u32)t << 3) | ((u32)idx << 21) | (1u << 28)) : (c_side ? 128u : 0u);
                    }
                    if (KIND != STEP_C) {
                        if (!TILE_P) {
                        const int dd11 = (int)(((ent.w >> 13) & 0x700u) | (ent.w & 0xFFu));
                        ro += c_cm != 0u ? (u32)((dd11 << 21) >> 21) : 0u;
                        }
                        const u32 z = ent.z;
                        const int nq_ = (int)(z & 0x7FFFu);
                        const int t8 = (int)((z >> (16 + 8 * c_side)) & 0xFFu);      // task nibbles / run length of this side
                        // nibble of column k: jo + k on the left side, jo - 1 - k on the right side
                        const int jo = c_side ? c_m8 + 16 - A : A - c_m8, sgn = 1 - 2 * c_side;
                        auto c16 = [](int v) -> u32 { return (u32)(v < 0 ? 0 : (v > 16 ? 16 : v)) << 3; };   // offset into the nibble-mask table
                        if (KIND == STEP_P) {
                            // the lane's tasks: its nibbles [0, t8 - m16) on the left side, [16 - (t8 - m16), 16) on the right
                            const int dm = t8 - c_m8;
                            st.aux = act ? c16(c_side ? 16 - dm : dm) : (c_side ? 128u : 0u);
                            if constexpr (FIDP) st.mk = *(const u64 *)((const u8 *)ltab + (st.aux & 0xFFu));
                            if (RS) {
                                // (a fused partial record, see the complete ones above: a record shorter than --length has all its
                                // columns in its left window; [15:8] the offset of the nibble-mask table's entry, [27:21], bit 28)
                                const bool fz = act && ((ent.w >> 18) & 1u);
                                int t = c_side ? c_m8 + 16 - A + L - nq_ : 16;
                                t = t < 0 ? 0 : (t > 16 ? 16 : t);
                                st.aux |= fz ? (((u32)t << 11) | ((u32)idx << 21) | (1u << 28)) : (c_side ? 128u << 8 : 0u);
                            }
                        } else {
                            // one indel of g bases behind the first (left) / last (right) match run of t8 columns: the gap is
                            // nibbles [ta, tb), the tasks end (left) / start (right) at nibble tl (see the ASCII kernel's fill)
                            const int dd = (int)(i8)(ent.w & 0xFFu);
                            const int g = KIND == STEP_GD ? dd : -dd;
                            const int ncol = KIND == STEP_GD ? nq_ + g : nq_;
                            const int Lm = ncol < L ? ncol : L;
                            const int ta = jo + sgn * t8 - (c_side ? g : 0), tb = ta + g;
                            const int tl = act ? jo + sgn * Lm : (c_side ? 16 : 0);
                            const bool blane = c_side ? tb >= 16 : ta <= 0;
                            const u32 off = blane ? (u32)(-sgn * g) : 0u;
                            if (KIND == STEP_GD) so += off; else ro += off;
                            const int bnd = c_side ? ta : tb;
                            st.aux = c16(ta) | (c16(tb) << 8) | (c16(tl) << 16) | ((blane ? 0u : (u32)g) << 24);
                            st.aux2 = (u32)g | ((c16(bnd) >> 3) << 3);
                            if (RS) {
                                // (a fused single-indel record, see the complete ones above: ncol columns, each once; [15:8] the
                                // offset of the nibble-mask table's entry, [27:21] the entry's place, bit 28)
                                const bool fz = act && ((ent.w >> 18) & 1u);
                                int t = c_side ? c_m8 + 16 - A + L - ncol : 16;
                                t = t < 0 ? 0 : (t > 16 ? 16 : t);
                                st.aux2 |= fz ? (((u32)t << 11) | ((u32)idx << 21) | (1u << 28)) : (c_side ? 128u << 8 : 0u);
                            }
                        }
                    }
                    // sixteen nibbles from bit 4 (offset & 7) of the aligned dword triple
                    st.ra = ro << 2; st.sa = so << 2;
                    st.r = *(const u32v3_u *)(refW + (((ro >> 1) & 0x7FFFFFFCu) | ref_copy));
                    st.s = *(const u32v3_u *)(seqW + ((so >> 1) & ~3u));
                    st.pk = KIND == STEP_C ? 0u : ent.w;
#if MDX_PK_ENT_AHEAD
                    { bool a_; ent_next = stg[ent_index(kf, a_)]; }
#endif
                };
                // one step: X = the nibbles this step counts (one-hot codes: the increments themselves).  A step whose events do
                // not fit the queue does nothing and reports itself (ovf, kredo): the run stops behind the group, drains and
                // starts again from that step
                bool ovf = false;
                int kredo = 0;
                bool grp_y = false;     // MASK: some step of the group holds masked columns
                auto count16 = [&](const St16 &st, auto full_tag, u32 &Xlo, u32 &Xhi, u32 &Ylo, u32 &Yhi) {
                    constexpr bool FULL = decltype(full_tag)::value;
                    u32 s_lo = __builtin_amdgcn_alignbit(st.s.y, st.s.x, st.sa), s_hi = __builtin_amdgcn_alignbit(st.s.z, st.s.y, st.sa);
                    u32 r_lo = __builtin_amdgcn_alignbit(st.r.y, st.r.x, st.ra), r_hi = __builtin_amdgcn_alignbit(st.r.z, st.r.y, st.ra);
                    u32 evw = c_evw;
                    if (RS && (KIND == STEP_C || KIND == STEP_P)) evw |= st.aux & 0x1FE00000u;
                    if (RS && (KIND == STEP_GI || KIND == STEP_GD)) evw |= st.aux2 & 0x1FE00000u;
                    u32 by_lo = 0u, by_hi = 0u;     // RS, single-indel steps: the reference bases of the step's columns
                    u64 dmk = 0ull;         // STEP_GD: the nibbles behind the deletion, counted by position
                    // --min-basequal: the mask is in the column itself (MDX_SEQ_4BITQ) — a symbol whose quality is below the threshold
                    // is the complement of its code, three or four bits set, and every three of a nibble's four bits hold an
                    // adjacent pair.  mk64 = the lane's nibbles that hold such a base, which are turned back into codes here; a step none
                    // of whose lanes holds one — clean data — is the unmasked step
                    u64 mk64 = 0ull, behm = 0ull, ymk = 0ull;
                    bool has_m = false;
                    if (MASK) {
                        const u32 a_lo = s_lo & (s_lo >> 1) & 0x77777777u, a_hi = s_hi & (s_hi >> 1) & 0x77777777u;
                        has_m = __ballot((a_lo | a_hi) != 0u) != 0ull;
                        if (has_m) {
                            const u32 m_lo = (a_lo | (a_lo >> 1) | (a_lo >> 2)) & 0x11111111u, m_hi = (a_hi | (a_hi >> 1) | (a_hi >> 2)) & 0x11111111u;
                            const u32 k_lo = (m_lo << 4) - m_lo, k_hi = (m_hi << 4) - m_hi;
                            s_lo ^= k_lo; s_hi ^= k_hi;
                            mk64 = (u64)k_lo | ((u64)k_hi << 32);
                        }
                    }
                    if (KIND == STEP_C) {
                        // (a nibble that is not a task has counters of its own, which nothing reads)
                        if (!FULL) {
                            // (a slot past its strand's last entry: nothing counted, no event)
                            const u32 actm = H * st.k < lim_l ? ~0u : 0u;
                            r_lo &= actm; r_hi &= actm; s_lo &= actm; s_hi &= actm;
                        }
                        Xlo = r_lo; Xhi = r_hi;
                    } else {
                        const u32 aux = st.aux;
                        const u64 sm = c_side ? ~0ull : 0ull;                      // all ones on the right side
                        const u64 vm64 = (u64)c_vm_lo | ((u64)c_vm_hi << 32);
                        u64 s64 = (u64)s_lo | ((u64)s_hi << 32), r64 = (u64)r_lo | ((u64)r_hi << 32), X64;
                        if (KIND == STEP_P) {
                            u64 Mk;
                            if constexpr (FIDP) Mk = st.mk; else Mk = *(const u64 *)((const u8 *)ltab + (aux & 0xFFu));
                            const u64 dyn = (Mk ^ sm) & vm64;
                            s64 &= dyn; r64 &= dyn;
                            X64 = r64;
                        } else {
                            const u64 Xk = *(const u64 *)((const u8 *)ltab + (aux & 0xFFu)), Yk = *(const u64 *)((const u8 *)ltab + ((aux >> 8) & 0xFFu)),
                                      Tk = *(const u64 *)((const u8 *)ltab + ((aux >> 16) & 0xFFu));
                            const u64 dyn = (Tk ^ sm) & vm64;
                            // the string that carries the gap: its nibbles in front of the gap as loaded, the gap symbol,
                            // its nibbles behind the gap moved by g within the lane that straddles it
                            const u32 shb = ((aux >> 24) & 7u) << 2, shr = c_side ? shb : 0u, shl = c_side ? 0u : shb;
                            const u64 star = KIND == STEP_GD ? s64 : r64;
                            const u64 low = star >> shr, high = star << shl;
                            u64 m = (Xk & low) | ~Xk;                            // (the gap symbol: 15)
                            m = (Yk & m) | (~Yk & high);
                            if (KIND == STEP_GD) {
                                s64 = m;
                                // (the mask travels with the read; a deleted column has no quality: never masked, align.py:67)
                                if (MASK && has_m) mk64 = (Yk & Xk & (mk64 >> shr)) | (~Yk & (mk64 << shl));
                                // nibbles behind the deletion (left: from tb on, right: below ta): MIS[column][base] and
                                // CMP[column - g][base] by position instead of the counters (below)
                                const u64 beh = (Xk & sm) | (~Yk & ~sm);
                                behm = beh;
                                dmk = dyn & beh;
                                evw |= 0x1000u | ((st.aux2 & 7u) << 13) | (((st.aux2 >> 3) & 31u) << 16);
                                s64 &= dyn; r64 &= dyn;
                                X64 = r64 & ~beh;
                                if (RS) { by_lo = (u32)r64; by_hi = (u32)(r64 >> 32); }
                            } else {
                                r64 = m;
                                s64 &= dyn; r64 &= dyn;
                                X64 = r64 & (Xk | ~Yk);                          // (not the gap symbols)
                                if (RS) { by_lo = (u32)X64; by_hi = (u32)(X64 >> 32); }
                            }
                        }
                        s_lo = (u32)s64; s_hi = (u32)(s64 >> 32); r_lo = (u32)r64; r_hi = (u32)(r64 >> 32);
                        Xlo = (u32)X64; Xhi = (u32)(X64 >> 32);
                    }
                    if (MASK && has_m) {
                        // A masked read column counts its read base — the second set of planes — and nothing else: it leaves both
                        // strings (no count of its reference base, no event)
                        const u32 mk_lo = (u32)mk64 & c_em_lo, mk_hi = (u32)(mk64 >> 32) & c_em_hi;
                        Ylo = s_lo & mk_lo; Yhi = s_hi & mk_hi;
                        s_lo &= ~mk_lo; s_hi &= ~mk_hi; r_lo &= ~mk_lo; r_hi &= ~mk_hi;
                        Xlo &= ~mk_lo; Xhi &= ~mk_hi;
                        if (KIND == STEP_GD) {
                            // (behind the deletion of its record a column's composition position is g less than the lane's:
                            // those read bases by position, below — not through the planes)
                            dmk &= ~((u64)mk_lo | ((u64)mk_hi << 32));
                            ymk = ((u64)Ylo | ((u64)Yhi << 32)) & behm;
                            Ylo &= ~(u32)behm; Yhi &= ~(u32)(behm >> 32);
                        }
                        grp_y = true;
                    }
                    // the lanes holding a read column that is not a plain match queue their four dwords (see qQ)
                    const u32 x_lo = (s_lo ^ r_lo) & c_em_lo, x_hi = (s_hi ^ r_hi) & c_em_hi;
                    const bool ev = (x_lo | x_hi) != 0u;
#ifdef MDX_ABL_NOEVQ        // (instruction counts only: with no event the SEQ windows of a complete step are dead and their loads go
                           // too — a third of the kernel's time, which is not the queue's)
                    const u64 mm = 0ull;
#else
                    const u64 mm = __ballot(ev);
#endif
                    if (mm) {
                        const int n = __popcll(mm);
                        if (qcount + n > MDX_PK_QCAP) {
                            ovf = true; kredo = st.k;
                            Xlo = 0u; Xhi = 0u; Ylo = 0u; Yhi = 0u;
                            return;
                        }
                        if (ev) {
#if MDX_PK_FASTIDX
                            // (the event's place among this step's events, the queue's fill in the scalar part of both addresses)
                            const u32 slot = (u32)mbcnt64(mm, 0);
                            // (kept apart: the sum of slot and fill would be a vector add)
                            const u32 qa_s = (u32)__builtin_amdgcn_readfirstlane((int)(qQ_a + 16u * (u32)qcount)),
                                      ea_s = (u32)__builtin_amdgcn_readfirstlane((int)(qE_a + 4u * (u32)qcount));
                            *(lds_u4 *)(size_t)((slot << 4) + qa_s) = u32x4{s_lo, s_hi, r_lo, r_hi};
                            *(lds_u1 *)(size_t)((slot << 2) + ea_s) = evw;
#else
                            const int slot = mbcnt64(mm, qcount);
                            qQ[slot] = make_uint4(s_lo, s_hi, r_lo, r_hi);
                            qE[slot] = evw;
#endif
                        }
                        qcount += n;
                    }
#ifndef MDX_RSABL_NOBC
                    if (RS) {
                        // the reference bases of a fused record's columns, by class (a code is one-hot: four population counts
                        // per dword), in this lane's own counters — its slot fixes the strand
                        constexpr bool GK = KIND == STEP_GI || KIND == STEP_GD;
                        const u64 Mk = *(const u64 *)((const u8 *)ltab + (KIND == STEP_C ? st.aux & 0xFFu : (KIND == STEP_P ? (st.aux >> 8) & 0xFFu : (st.aux2 >> 8) & 0xFFu)));
                        const u32 sm = c_side ? ~0u : 0u;
                        const u32 y_lo = (GK ? by_lo : r_lo) & ((u32)Mk ^ sm) & c_em_lo, y_hi = (GK ? by_hi : r_hi) & ((u32)(Mk >> 32) ^ sm) & c_em_hi;
                        bcA += __builtin_popcount(y_lo & 0x11111111u) + __builtin_popcount(y_hi & 0x11111111u);
                        bcC += __builtin_popcount(y_lo & 0x22222222u) + __builtin_popcount(y_hi & 0x22222222u);
                        bcT += __builtin_popcount(y_lo & 0x44444444u) + __builtin_popcount(y_hi & 0x44444444u);
                        bcG += __builtin_popcount(y_lo & 0x88888888u) + __builtin_popcount(y_hi & 0x88888888u);
                    }
#endif
                    if (MASK && KIND == STEP_GD && __ballot(ymk != 0ull)) {
                        // (the masked read bases behind the deletion: CMP[column - g][base], like the pass below)
                        const int g = (int)(st.aux2 & 7u);
                        const int rev = (int)(st.pk >> 31);
                        const int row = __mul24(rev * 2 + c_side, L) + c_m8 - A;
                        u32 *const pc = lds + d.off_cmp() + (row - g) * 4;
                        const u64 rr = c_side ? __builtin_bitreverse64(ymk) : ymk;
                        const u32 kx = c_side ? 3u : 0u;
#pragma unroll 1
                        for (int j = 0; j < 16; j++) {
                            const u32 nib = (u32)(rr >> (4 * j)) & 15u;
                            const u32 k = (u32)(__ffs((int)nib) - 1) ^ kx;
                            if (nib) atomicAdd(pc + 4 * j + k, 1u);
                        }
                    }
                    if (KIND == STEP_GD && __ballot(dmk != 0ull)) {
                        // one nibble at a time in position order — nibble j on the left side, 15 - j on the right (whose bits
                        // are reversed too: class k is bit 3 - k) —, unrolled: the class of the (one-hot or zero) nibble picks
                        // the word of the row
                        const u64 r64 = ((u64)r_lo | ((u64)r_hi << 32)) & dmk;
                        const int g = (int)(st.aux2 & 7u);
                        const int rev = (int)(st.pk >> 31);
                        const int row = __mul24(rev * 2 + c_side, L) + c_m8 - A;        // row of the lane's lowest position
                        u32 *const pm = lds + d.off_mis() + __mul24(row, 25), *const pc = lds + d.off_cmp() + (row - g) * 4;
                        const u64 rr = c_side ? __builtin_bitreverse64(r64) : r64;
                        const u32 kx = c_side ? 3u : 0u;
#pragma unroll
                        for (int j = 0; j < 16; j++) {
                            const u32 nib = __builtin_amdgcn_ubfe(j < 8 ? (u32)rr : (u32)(rr >> 32), 4 * (j & 7), 4);
                            const u32 k = (u32)(__ffs((int)nib) - 1) ^ kx;
                            // (only the lanes that count take part: an add of 0 costs the LDS what an add of 1 costs)
                            if (nib) { atomicAdd(pm + 25 * j + k, 1u); atomicAdd(pc + 4 * j + k, 1u); }
                        }
                    }
                };
                // (--min-basequal: three — the second set of planes and the bitmap words want the registers of the fourth)
                // (ML without MASK: the depths of the one-library kernel since round 6 — its arguments come by scalar loads and
                // it has that kernel's registers; ML with MASK — rounds and a second set of planes — keeps depths of its own)
                constexpr bool MLM = ML && MASK;
                constexpr int PD4 = (KIND == STEP_GI || KIND == STEP_GD) ? (MLM ? MDX_PD_G_ML : ((MASK || RS) ? MDX_PD_G_M : MDX_PD_G)) : (KIND == STEP_P ? (HS ? MDX_PK_PD_P : (MLM ? MDX_PD_P_ML : MDX_PD_P)) : (MASK ? (ML ? MDX_PKM_PD_ML : MDX_PKM_PD) : (ML ? MDX_PK_PD_ML : MDX_PK_PD)));
                static_assert(PD4 >= 1 && PD4 <= 8, "steps in flight");
                St16 st[PD4];
                // a group of PD4 steps: their words through carry-save adders into the planes (bs_add_group: four at a time),
                // low and high dwords
                u32 pcL = 0u, pcH = 0u;     // (a pair of groups: the first one's words of weight 4)
                auto group = [&](auto full_tag, const bool refill, auto pair_tag) {
                    constexpr int PAIR = decltype(pair_tag)::value;     // 0: a group on its own; 1, 2: the first, second of a pair
                    u32 xl[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, xh[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
                    u32 yl[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, yh[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
                    grp_y = false;
#pragma unroll
                    for (int dd = 0; dd < PD4; dd++) {
                        if (!ovf && (refill || dd == 0 || st[dd].valid)) count16(st[dd], full_tag, xl[dd], xh[dd], yl[dd], yh[dd]);
                        if (refill) fill16(st[dd]);
#ifdef MDX_ABL_NOCSA
                        bsL[0] |= xl[dd]; bsH[0] |= xh[dd];
                        if (false) {
#else
                        if ((dd & 3) == 3 || dd == PD4 - 1) {
#endif
                            const u32 al[4] = {xl[dd & ~3], xl[(dd & ~3) + 1], xl[(dd & ~3) + 2], xl[(dd & ~3) + 3]};
                            const u32 ah[4] = {xh[dd & ~3], xh[(dd & ~3) + 1], xh[(dd & ~3) + 2], xh[(dd & ~3) + 3]};
                            if (PAIR == 1 && (dd & 3) == 3) { bs_group4(bsL, al, pcL); bs_group4(bsH, ah, pcH); }
                            else if (PAIR == 2 && (dd & 3) == 3) {
                                u32 cl_, ch_, dl_, dh_;
                                bs_group4(bsL, al, cl_); bs_group4(bsH, ah, ch_);
                                bs_csa(bsL[2], pcL, cl_, dl_); bs_csa(bsH[2], pcH, ch_, dh_);
                                bs_ripple<3>(bsL, dl_); bs_ripple<3>(bsH, dh_);
                            }
                            else if ((dd & 3) == 3) { bs_add_group<4>(bsL, al); bs_add_group<4>(bsH, ah); }
                            else if ((dd & 3) == 2) { bs_add_group<3>(bsL, al); bs_add_group<3>(bsH, ah); }
                            else if ((dd & 3) == 1) { bs_add_group<2>(bsL, al); bs_add_group<2>(bsH, ah); }
                            else { bs_add_group<1>(bsL, al); bs_add_group<1>(bsH, ah); }
                            if (MASK && grp_y) {
                                const u32 cl[4] = {yl[dd & ~3], yl[(dd & ~3) + 1], yl[(dd & ~3) + 2], yl[(dd & ~3) + 3]};
                                const u32 ch[4] = {yh[dd & ~3], yh[(dd & ~3) + 1], yh[(dd & ~3) + 2], yh[(dd & ~3) + 3]};
                                if ((dd & 3) == 3) { bs_add_group<4>(b2L, cl); bs_add_group<4>(b2H, ch); }
                                else if ((dd & 3) == 2) { bs_add_group<3>(b2L, cl); bs_add_group<3>(b2H, ch); }
                                else if ((dd & 3) == 1) { bs_add_group<2>(b2L, cl); bs_add_group<2>(b2H, ch); }
                                else { bs_add_group<1>(b2L, cl); bs_add_group<1>(b2H, ch); }
                            }
                        }
                    }
                };
                // (two loops over the same pipeline: the groups all of whose slots hold records, then the others; never the last
                // step of the run.  As a rule once through: again from step kredo, the queue drained, when that step's events
                // did not fit)
                // (nothing but loads in flight in the loops below — the caller has waited for the stores of the tile's phase 1 —:
                // with a store possibly pending the head of the loop would wait for vmcnt(0))
#pragma unroll 1
                for (int kstart = 0;;) {
                    kf = kstart;
                    ovf = false;
#if MDX_PK_ENT_AHEAD
                    { bool a_; ent_next = stg[ent_index(kf, a_)]; }
#endif
#pragma unroll
                    for (int dd = 0; dd < PD4; dd++) fill16(st[dd]);
                    if (pfl && pfl_due != 0xFFFFFFFEu) {
                        // the next tile's columns — older than the 2 PD4 loads just issued, and loads return in order — have landed:
                        // its second round trip out, behind this run's first loads (stores still on their way only make
                        // the count wait longer)
                        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(2 * PD4) : "memory");
                        if (pfl_due != 0xFFFFFFFFu) pfl_rt2(pfl_due);
                        pfl_due = 0xFFFFFFFEu;
                    }
                    int k = kstart + PD4;
                    using P0 = std::integral_constant<int, 0>;
                    if (HS && PD4 == 4) {
                        // (pairs while two whole groups of complete steps are left; the second one folds the first one's words
                        // whatever happened to its own steps)
                        for (; k + PD4 < nsteps4 && k + PD4 <= nfull && !ovf; k += 2 * PD4) {
                            group(std::true_type{}, true, std::integral_constant<int, 1>{});
                            group(std::true_type{}, true, std::integral_constant<int, 2>{});
                        }
                    }
                    for (; k < nsteps4 && k <= nfull && !ovf; k += PD4) group(std::true_type{}, true, P0{});
                    for (; k < nsteps4 && !ovf; k += PD4) group(std::false_type{}, true, P0{});
                    if (!ovf) group(std::false_type{}, false, P0{});
                    if (!ovf) break;
                    drain_all();
                    if (RS) {
                        rsq_flush();
                        __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): nothing but the run's loads in flight in its loop
                    }
                    kstart = kredo;
                }
                // (where the packed kernel drains as a rule: behind a run, once a pass's worth of events waits — whole passes;
                // RS: all of them — the events of fused records look their staging entries up, and the MR sums of the tile's
                // records are formed behind its run)
                if (RS) { if (qcount > 0) drain_all(); }
                else if (qcount >= 64) drain_all(qcount & 63);
                return;
            }
#if MDX_ENT_AHEAD
            // (the staging entry of a step is read from the LDS one fill ahead: its latency — behind the eight table
            // updates of the step just counted — is off the path to the window loads)
            uint4 ent_next = stg[e0 + c_slot];
#endif
            // fill() always issues its loads (past the last step it re-reads it), so the number of
            // loads in flight is static and the waits before count() are counted ones
            auto fill = [&](Stage &st) {
                st.valid = kf < nsteps;
                const int k = st.valid ? kf : nsteps - 1;
                kf++;
                int nv = nrec - k * R;
                nv = nv > R ? R : nv;
                st.lim = nv * G;
                st.k = k;
#if MDX_ENT_AHEAD
                const uint4 ent = ent_next;
                {
                    const int kn = kf < nsteps ? kf : nsteps - 1;
                    ent_next = stg[e0 + kn * R + c_slot];
                }
#else
                const uint4 ent = stg[e0 + k * R + c_slot];
#endif
                const u32 t = ent.z & c_cm;
                u32 ro = ent.x + c_ro + t;
                u32 so = ent.y + c_so + t;
                // gapped record, n0 - nq in the entry (11 bits signed): the right windows hang off aend = pos + n0,
                // not pos + nq
                if (KIND != STEP_C) {
                    const int dd11 = (int)(((ent.w >> 13) & 0x700u) | (ent.w & 0xFFu));
                    ro += c_cm != 0u ? (u32)((dd11 << 21) >> 21) : 0u;
                    // byte thresholds of the lane for count(), clamped to [0, 8] and times eight.  Byte of column k:
                    // jo + k on the left side, jo - 1 - k on the right side (sgn = +1 / -1)
                    const u32 z = ent.z;
                    const int nq_ = (int)(z & 0x7FFFu);
                    const int t8 = (int)((z >> (16 + 8 * c_side)) & 0xFFu);      // flank length / run length of this side
                    const int jo = c_side ? c_m8 + 8 - A : A - c_m8, sgn = 1 - 2 * c_side;
                    const bool act = lane < st.lim;
                    auto c8 = [](int v) -> u32 { return (u32)(v < 0 ? 0 : (v > 8 ? 8 : v)) << 3; };
                    if (KIND == STEP_P) {
                        // t8 = the task bytes of this side's window, counted from its outer end (the flank is complete:
                        // records at a contig edge walk): the lane's tasks are its bytes [0, t8 - c_m8) on the left
                        // side, [8 - (t8 - c_m8), 8) on the right side
                        // (aux = eight times the number of bytes below the boundary; a slot without a record: no tasks)
                        const int dm = t8 - c_m8;
                        st.aux = act ? c8(c_side ? 8 - dm : dm) : (c_side ? 64u : 0u);
                    } else {
                        // one indel of g bases behind the first (left) / last (right) match run of t8 columns: the gap
                        // is bytes [ta, tb), the tasks end (left) / start (right) at byte tl.  A lane whose first
                        // column, seen from its end of the record, lies at or behind the gap reads the string that
                        // carries the gap g bytes nearer to that end; the lane that straddles the gap moves its own
                        // bytes (count()).
                        const int dd = (int)(i8)(ent.w & 0xFFu);
                        const int g = KIND == STEP_GD ? dd : -dd;
                        const int ncol = KIND == STEP_GD ? nq_ + g : nq_;
                        const int Lm = ncol < L ? ncol : L;
                        const int ta = jo + sgn * t8 - (c_side ? g : 0), tb = ta + g;
                        const int tl = act ? jo + sgn * Lm : (c_side ? 8 : 0);
                        const bool blane = c_side ? tb >= 8 : ta <= 0;
                        const u32 off = blane ? (u32)(-sgn * g) : 0u;
                        if (KIND == STEP_GD) so += off; else ro += off;
                        const int bnd = c_side ? ta : tb;
                        st.aux = c8(ta) | (c8(tb) << 7) | (c8(tl) << 14) | ((blane ? 0u : (u32)g) << 21) | ((u32)g << 24) |
                                 ((c8(bnd) >> 3) << 27);
                    }
                }
                st.ro = ro; st.so = so;
                st.r12 = *(const u32x3 *)(refW + (ro & ~3u));
                st.s12 = ld12_stream(seqW + (so & ~3u));
                // (RS: the low byte — n0 - nq of a gapped entry, read above — makes room for the entry's place in the
                // staging area, which an event of a fused record hands to drain_all)
                st.pk = RS ? ((ent.w & 0xFFFFFF00u) | (u32)(e0 + k * R + c_slot)) : ent.w;
#if MDX_QPREFETCH
                if (QM) {
                    // records that cannot be masked — no qualities, or the caller's hint — read one fixed line instead
                    const u32 qo = so - c_so + c_qo;
                    st.q12 = ld12_stream(qualW + ((ent.w & 0x40000000u) ? (qo & ~3u) : 0u));
                }
#endif
            };
            // software pipeline: PIPE_DEPTH steps in flight, each in its own register set (no register
            // rotation: a copy of an in-flight destination would wait for its load).  Every point of
            // the loop has the same number of loads in flight (counted s_waitcnt vmcnt), and at most
            // PIPE_DEPTH - 1 fills per run go past the last step.
            // (the runs of gapped records are a step or two long: two register sets)
            // (complete runs: four steps in flight in the unmasked kernel, which has the registers since round 3 — plain
            // records 2 %, config 3 1.8 % faster than with three; the masked kernel, with its third window, would spill)
            constexpr int PD = (KIND == STEP_GI || KIND == STEP_GD) ? MDX_PD_G : ((KIND == STEP_P || MASK) ? MDX_PD_P : (RS ? MDX_FUSE_PD : PIPE_DEPTH));
            Stage st[PD];
            // (as a rule once through; RS: again from step rs_kredo, the queue drained — outside the pipelined loop, whose
            // waits stay counted —, when a step found it too full)
#pragma unroll 1
            for (int kstart = 0;;) {
                kf = kstart;
                rs_ovf = false;
    #pragma unroll
                for (int dd = 0; dd < PD; dd++) fill(st[dd]);
                for (int k = kstart + PD; k < nsteps && !(RS && rs_ovf); k += PD) {
    #pragma unroll
                    for (int dd = 0; dd < PD; dd++) {
                        if (!(RS && rs_ovf)) count(st[dd], kind_tag, std::true_type{}, qm_tag);      // (never the last step of the run)
                        fill(st[dd]);
                    }
                }
    #pragma unroll
                for (int dd = 0; dd < PD; dd++)
                    if (!(RS && rs_ovf) && (dd == 0 || st[dd].valid)) count(st[dd], kind_tag, std::false_type{}, qm_tag);
                if (!RS || !rs_ovf) break;
                drain_all();
                rsq_flush();
                __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): nothing but the run's loads in flight in its loop
                kstart = rs_kredo;
            }
            // (RS: the events of fused records look their staging entries up: drained before those are overwritten)
#ifndef MDX_RSABL_NOFD
            if (RS && qcount > 0) drain_all();
#endif
        };

    // Without the fast path, tiles of 64 records are dealt round-robin to the wavefronts (a run of expensive records —
    // reads over an assembly gap in a coordinate-sorted batch — is spread over many wavefronts instead of one) and the
    // records left after the last complete round are split evenly; the fast kernels hand their tiles out on demand
    // (see the tile loop).
    // (record indices fit 32 bits: mdx_tabulate_device rejects batches of 2^30 records and more)
    // A tile holds a multiple of R records (63 at three per step): the fast run of a tile of complete
    // records ends on a full step.
    // (ML: the records of the pool's library, places [rec_lo, rec_lo + n_rec) of the bucketed columns; ml_lib = that library,
    // counted from the launch's first one)
    u32 ml_k = 0u, ml_m = 1u, ml_first = 0u;     // ML: the pool's place among the ml_m pools of its library, the first of which is pool ml_first
    const u32 rounds = (n_rec / T) / nwaves;
    const u32 rem_lo = rounds * nwaves * T, rem = n_rec - rem_lo;
    const u32 t_lo = rem_lo + (u32)((u64)rem * gwave / nwaves), t_hi = rem_lo + (u32)((u64)rem * (gwave + 1) / nwaves);
    const u32 n_it = rounds + (t_hi - t_lo + T - 1) / T;
    // the wavefront's lists of staged entries that are not complete records (MdxTabArgs::lists)
    // The wavefront's part of MdxTabArgs::lists (MDX_WAVE_SCRATCH(ring_size) 16-byte entries): rings.
    //   ringP / ringI / ringD / ringC   MDX_LIST_RING entries each: partial records, single insertions, single deletions, the
    //                                   complete records the general pass finds.  A list is emptied at the end of a round
    //                                   of the tile loop (MdxTabArgs::round_tiles tiles), whole passes of 63 entries; a round
    //                                   appends at most its records, 63 round_tiles, and fewer than 63 are left over.
    //   dcols / dlist                   MDX_DRING records the tile loop leaves to the general pass (their columns, two entries
    //                                   each, and their indices): taken 64 at a time as soon as 64 wait — fewer than 127 do
    //   lri / lriI / lriD               RS: the record index of every entry of ringP (and, PK, of ringI / ringD)
    // l* = entries appended so far, h* = entries taken so far (entry k of a ring sits at k & (size - 1)).  Written with plain
    // stores and read back by the same wavefront past the vector L1 (ring_at: a line of a ring may sit there from the
    // turn before).
    // (the rings' size is the launch's: MDX_LIST_RING for the kernels that work in rounds; the fused kernels and the packed
    // masked kernel of a one-library launch — ROUNDS false: their registers do not take the loop of rounds around the tile
    // loop, measured: config 5 -6 %, --min-basequal -6 % (10 registers spilled) — keep one round and rings that hold what a
    // wavefront's quota of tiles can append.  A launch over several libraries works in rounds whatever else it is: its pools
    // are as full as their libraries are large, and a quota would have to know the fullest)
    constexpr bool ROUNDS = !(RS || (MASK && PK && !ML));
    constexpr u32 DM = MDX_DRING - 1;
    const u32 RM = (u32)a.ring_size - 1u;
    uint4 *const lists = a.lists + (i64)gwave * MDX_WAVE_SCRATCH(a.ring_size);
    uint4 *const ringP = lists, *const ringI = lists + a.ring_size, *const ringD = lists + 2 * a.ring_size, *const ringC = lists + 3 * a.ring_size;
    uint4 *const dcols = lists + 4 * a.ring_size;
    u32 *const dlist = (u32 *)(lists + 4 * a.ring_size + 2 * MDX_DRING);
    u32 *const lri = dlist + MDX_DRING, *const lriI = lri + a.ring_size, *const lriD = lriI + a.ring_size;
    u32 lP = 0, lI = 0, lD = 0, lC = 0, hP = 0, hI = 0, hD = 0, hC = 0;
    auto ring_u32 = [](const u32 *p) -> u32 { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto ring_at = [&](const uint4 *p) -> uint4 {
        const u32 *q = (const u32 *)p;
        return make_uint4(ring_u32(q), ring_u32(q + 1), ring_u32(q + 2), ring_u32(q + 3));
    };
    u32 n_rs = 0;           // RS: records left to the rescale kernels behind this one

    // class of the read symbol at index i of the SEQ column / of the reference symbol at (concatenated) genome coordinate i,
    // which may lie in the guard bands — in either form of the two columns
    // (PK with --min-basequal: the column holds masked bases as the complements of their codes — seq_cls gives the base,
    // seq_msk says whether its quality was below the threshold)
    auto seq_cls = [&](const u32 i) -> int {
        if (PK) {
            u32 nib = ((u32)a.seq[i >> 1] >> (4u * (i & 1u))) & 15u;
            if (MASK) { bool m; nib = unmask4(nib, m); }
            return cls4(nib);
        }
        return classify_read(a.seq[i]);
    };
    auto seq_msk = [&](const u32 i) -> bool {
        bool m;
        unmask4(((u32)a.seq[i >> 1] >> (4u * (i & 1u))) & 15u, m);
        return m;
    };
    auto ref_cls = [&](const i64 i) -> int {
        if (PK) { const i64 n = i + 256; return cls4(((u32)a.ref4[n >> 1] >> (4 * (int)(n & 1))) & 15u); }
        return classify_ref(((const i8 *)a.ref)[i]);
    };
    // ------------------------------------------------------------ the general pass: any record, lane per record
    // Everything the reference's loop body does to a record up to the columns the steps count: flag filter, CIGAR scan,
    // fragment length, soft clips, error checks, classification.  FAST: fed 64 at a time with the records the tile loop
    // cannot take (anything but a single match operation) — their entries go to the wavefront's lists; otherwise it is
    // the whole tile loop.
    // (the record's columns come with it: the tile loop has read them once already and hands them over through the list)
    // (rs_code — the packed fused kernel: 0 the record is not to be rescaled, or the tile loop has seen to it; 1 / 2 it is, from
    // both ends / from its 5' end only: a single-indel record of [L, 2 L] columns is then rescaled by its own steps — its entry
    // marked like a fused record of the tile loop —, any other one is listed for the rescale kernels here)
    auto general = [&](const u32 ri, const bool valid, const u32 fl, const int c_lib, const int c_tid, const int c_pos,
                       const int c_tlen, const u32 c_co0, const u32 c_co1, const u32 c_so0, const u32 c_so1, const u32 rs_code = 0u) {
        // the arguments phase 1 needs are read from the kernel-argument segment when they are used (scalar loads through
        // the constant cache) instead of living in SGPRs across the whole kernel: the kernel wants far more scalar
        // registers than there are, and every spilled one costs a v_readlane per use
        karg_p kp = ka;
        asm volatile("" : "+s"(kp));
        const __attribute__((address_space(4))) MdxTabArgs &p = *kp;
        // ------------------------------------------------------------ phase 1: lane per record
        bool kept = valid && (fl & 0xF04u) == 0;  // reader.py:121-132
        // a launch counts the libraries [lib_lo, lib_lo + d.nlib) (mdx_capi.cpp: as many as fit the LDS); records of
        // the others are left to their own launch (a library id beyond the last one is an error in every launch)
        if (!ML && c_lib < a.nlib_total && (c_lib < a.lib_lo || c_lib >= a.lib_lo + d.nlib)) kept = false;
        // (ML: the image holds the pool's library alone — table offsets are library 0's, the fragment lengths beyond the
        // LDS histogram go by lg_lib)
        const int lg_lib = ML ? ml_lib : c_lib - a.lib_lo;
        int w1 = 0, nq = 0, libid = 0, n0 = 0, ncols = 0, nI = 0, cig_n = 0;
        int vlr = 0;   // gapped records: columns of the first / last match run, capped at L (vl | vr << 8)
        bool one = false;   // [H][S] M {I|D} M [S][H]: one indel between two match runs
        bool skips = false; // an N or P operation (or four and more indels)
        bool nonly = false; // gapped by N / P operations only
        u32 sq = 0, cig_o = 0;
        i64 rbase = 0;
        int lkey = -1;  // fragment-length key for the LDS histogram
        if (kept) {
            const int rev = (fl >> 4) & 1;
            libid = ML ? 0 : c_lib - a.lib_lo;
            const int tid = c_tid;
            const int pos = c_pos;
            cig_o = c_co0;
            cig_n = (int)(c_co1 - cig_o);
            const u32 so = c_so0;
            const i64 lseq = (i64)c_so1 - (i64)so;
            bool bad = tid < 0 || tid >= a.n_contig || c_lib >= a.nlib_total || lseq <= 0 || pos < 0;
            const int lbase = bad ? 0 : libid * d.w_lib;
            // second round trip of the tile: the first five CIGAR operations and the contig bounds together (a record with one
            // indel between clips has five: the scan below then waits once, not once per operation)
            const u32 cg0 = cig_n > 0 ? a.cigar[cig_o] : 0u, cg1 = cig_n > 1 ? a.cigar[cig_o + 1] : 0u, cg2 = cig_n > 2 ? a.cigar[cig_o + 2] : 0u,
                      cg3 = cig_n > 3 ? a.cigar[cig_o + 3] : 0u, cg4 = cig_n > 4 ? a.cigar[cig_o + 4] : 0u;
            i64 c0 = 0, clen = 0;
            if (!bad) {
                c0 = a.contig_off[tid];
                clen = a.contig_off[tid + 1] - c0;
            }
            // statistics.py:37-51: positions [0, min(len, L)) of a soft clip, left side iff no alignment column precedes
            // it — as a difference: +1 at 0, -1 at m (finalize_kernel sums the prefix; the partial tables are folded as
            // signed words) instead of m increments
            auto clip = [&](const int side, const i64 len) {
                const int m = len < (i64)L ? (int)len : L;
                const int base = lbase + d.off_mis() + ((rev * 2 + side) * L) * 25 + COL_S;
                if (m > 0 && !bad) {
                    bump<USE_LDS>(lds, raw, base);
                    if (m < L) {
                        if (USE_LDS) atomicAdd(&lds[base + m * 25], 0xFFFFFFFFu);
                        else atomicAdd(&raw[base + m * 25], ~0ull);
                    }
                }
            };

            // CIGAR scan: pysam query_alignment_start/_end, htslib bam_endpos, parse_cigar.  One forward pass in 32-bit
            // arithmetic, branch-free but for soft clips: sums of the match (M = X), query-consuming, reference-
            // consuming and column operations (the I / D / N totals follow from them), first and last match run, leading
            // and trailing soft clips.  Exact for CIGARs of at most 16 operations shorter than 2^24 with at most one soft
            // clip per side; anything else is redone by the 64-bit pass below (ordinary data never takes it).
            u32 sM = 0, sQ = 0, sR = 0, sC = 0, qs = 0, trail = 0;
            int lead_m = 0, cur_run = 0;   // first and current (at the end: last) run of M/=/X columns
            int nID = 0, nE = 0;           // I / D operations; operations that end a run (I, D, N, P, unknown ones)
            bool redo = cig_n > 16;
            if (!redo) {
                u32 big = 0, cl_l = 0, cl_r = 0, lopen = ~0u, leading = ~0u;
                for (int k = 0; k < cig_n; k++) {
                    const u32 c = k == 0 ? cg0 : (k == 1 ? cg1 : (k == 2 ? cg2 : (k == 3 ? cg3 : (k == 4 ? cg4 : a.cigar[cig_o + k]))));
                    const u32 op = c & 0xFu, len = c >> 4;
                    big |= len >> 24;
                    const u32 m = (0x181u >> op) & 1u, e = (0xFE4Eu >> op) & 1u, keep = e - 1u;
                    sM = __umul24(len, m) + sM;
                    sQ = __umul24(len, (0x183u >> op) & 1u) + sQ;
                    sR = __umul24(len, (0x18Du >> op) & 1u) + sR;
                    const u32 run = __umul24(len, m) + (u32)cur_run;
                    lead_m = (int)((run & lopen) | ((u32)lead_m & ~lopen));
                    cur_run = (int)(run & keep);
                    lopen &= keep;
                    nID += (int)((6u >> op) & 1u);
                    nE += (int)e;
                    const u32 clipop = 0u - ((0x30u >> op) & 1u);     // S or H: neither ends the leading clips nor the trailing ones
                    trail &= clipop; leading &= clipop;
                    if (op == 4u) {
                        qs += len & leading;
                        if (k >= 1) trail += len;
                        if (sC == 0) { redo = redo || cl_l != 0; cl_l = len; }
                        else { redo = redo || cl_r != 0; cl_r = len; }
                    }
                    sC = __umul24(len, (0x187u >> op) & 1u) + sC;
                }
                redo = redo || big != 0;
                if (!redo) {
                    if (cl_l) clip(0, (i64)cl_l);
                    if (cl_r) clip(1, (i64)cl_r);
                }
            }
            bool over = false;             // sums beyond what a batch can hold (64-bit pass only)
            if (__ballot(redo)) {
                if (redo) {
                    i64 w_qs = 0, w_M = 0, w_Q = 0, w_R = 0, w_C = 0, w_trail = 0;
                    bool leading = true, lead_open = true;
                    lead_m = 0; cur_run = 0; nID = 0; nE = 0;
                    for (int k = 0; k < cig_n; k++) {
                        const u32 c = a.cigar[cig_o + k];
                        const int op = c & 0xF;
                        const i64 len = c >> 4;
                        if (op == 0 || op == 7 || op == 8) {
                            w_C += len; w_R += len; w_Q += len; w_M += len;
                            const int l15 = len < 0x7FFF ? (int)len : 0x7FFF;
                            cur_run = cur_run + l15 < 0x7FFF ? cur_run + l15 : 0x7FFF;
                            if (lead_open) lead_m = cur_run;
                            leading = false; w_trail = 0;
                        } else if (op == 4) {
                            if (leading) w_qs += len;
                            if (k >= 1) w_trail += len;
                            clip(w_C == 0 ? 0 : 1, len);
                        } else if (op != 5) {   // I, D, N, P end a run
                            lead_open = false; cur_run = 0; leading = false; w_trail = 0;
                            nE++;
                            if (op == 1) { w_C += len; w_Q += len; nID++; }
                            else if (op == 2) { w_C += len; w_R += len; nID++; }
                            else if (op == 3) w_R += len;
                        }
                    }
                    over = w_C > 0x3FFFFFFF || w_R + (w_Q - w_M) > 0x3FFFFFFF;
                    sM = (u32)w_M; sQ = (u32)w_Q; sR = (u32)w_R; sC = (u32)w_C;
                    // (clips beyond any SEQ length saturate: the aligned query is empty either way)
                    qs = w_qs < 0xFFFFFFFFll ? (u32)w_qs : 0xFFFFFFFFu; trail = w_trail < 0xFFFFFFFFll ? (u32)w_trail : 0xFFFFFFFFu;
                }
            }
            // totals of the I, D and N operations
            const u32 sI = sQ - sM, sD = sC - sQ, sN = sR - sM - sD;
            const i64 qcl = (i64)qs + (i64)trail;
            const i64 nq64 = lseq > qcl ? lseq - qcl : 0;
            const u32 n0u = sR ? sR : 1u;
            const i64 aend = (i64)pos + (i64)n0u;
            rbase = c0 + pos;
            // align.py:33 / main.py:180: fetch(start > end) raises once aend > contig length;
            // a CIGAR that disagrees with SEQ cannot come out of htslib
            bad = bad || over || cig_n == 0 || aend > clen || nq64 != (i64)sQ || sC > 0x3FFFFFFFu || n0u + sI > 0x3FFFFFFFu;
            if (bad) {
                flag_error(p.err, (i64)(ML ? p.perm[ri] : ri) + p.record_base, ERR_BAD_READ);
                kept = false;
            } else {
                const int n_gap = nID + 4 * (nE - nID);
                const u32 sDN = sD + sN, rlen = sR;
                nq = (int)nq64; n0 = (int)n0u; ncols = (int)sC; nI = (int)sI;
                sq = so + qs;
                const int nbefore = pos < A ? (int)pos : A;
                const int nafter = clen - aend < A ? (int)(clen - aend) : A;
                const bool simple = sI == 0 && sDN == 0 && rlen > 0 && nq < 32768;
                if (!simple) {
                    vlr = (lead_m < L ? lead_m : L) | ((cur_run < L ? cur_run : L) << 8);
                    one = n_gap == 1 && lead_m > 0 && cur_run > 0;
                    skips = n_gap >= 4;
                    // N (and P) operations only: align.py:38-50 inserts nothing for them, so both strings are indexed
                    // like an ungapped record's from either end — the left window hangs off pos, the right one off
                    // aend, and the whole record is covered by its staging entry (no CIGAR walk)
                    nonly = sI == 0 && sD == 0;
                    if (nonly) vlr = (nq < L ? nq : L) * 0x101;
                }
                w1 = rev | (simple ? D_SIMPLE : 0) | ((nbefore & 0xFF) << D_NB_SHIFT) | ((nafter & 0xFF) << D_NA_SHIFT);
                if (simple && nq >= L && nbefore == A && nafter == A) w1 |= D_FULL;
                // (flag bit 0x8000, include/mdx.h: the caller vouches that no quality of the record is below the threshold)
                // (PK: the masks are in the SEQ column — no quality is read)
                if (MASK && PK) w1 |= D_HASQ;
                else if (MASK && !(fl & 0x8000u) && a.qual != nullptr && a.qual[so] != 0xFF) w1 |= D_HASQ;
                // statistics.py:117-126
                int kind = -1;
                i64 flen = 0;
                if (fl & 0x1) {
                    if ((fl & 0x40) && (fl & 0x2)) {
                        kind = 0;
                        const i64 t = c_tlen;
                        flen = t < 0 ? -t : t;
                    }
                } else {
                    kind = 1;
                    flen = (i64)n0u;
                }
                if (kind >= 0) {
                    if (flen < d.lgd_lds) {
                        lkey = lbase + d.off_lgd() + (kind * 2 + rev) * d.lgd_lds + (int)flen;
                    } else if (flen < d.lgd_max) {
                        atomicAdd(&p.lgd_dense[(i64)(blockIdx.x & (MDX_LGD_COPIES - 1)) * ((i64)a.nlib_total * 4 * d.lgd_max) +
                                               (((i64)lg_lib * 2 + kind) * 2 + rev) * d.lgd_max + flen], 1ull);
                    } else {
                        const u64 slot = atomicAdd(p.n_lgd_over, 1ull);
                        if ((i64)slot < p.lgd_over_cap) {
                            p.lgd_over[4 * slot + 0] = lg_lib + a.lib_lo;
                            p.lgd_over[4 * slot + 1] = kind;
                            p.lgd_over[4 * slot + 2] = rev;
                            p.lgd_over[4 * slot + 3] = flen;
                        }
                    }
                }
            }
        }
        // fragment lengths: two rounds of wave-level aggregation (uniform read lengths give one
        // or two distinct keys per tile), the remainder as individual adds
        {
            u64 pend = __ballot(lkey >= 0);
#pragma unroll 1
            for (int round = 0; round < 2 && pend; round++) {
                const int leader = __ffsll((long long)pend) - 1;
                const int key = rl(lkey, leader);
                const u64 same = __ballot(lkey == key);
                if (lane == leader) bump_n<USE_LDS>(lds, raw, key, (u32)__popcll(same));
                if (lkey == key) lkey = -1;
                pend &= ~same;
            }
            if (lkey >= 0) bump<USE_LDS>(lds, raw, lkey);
        }
        const u64 todo_all = __ballot(kept);
        if (lane == 0 && todo_all) bump_n<USE_LDS>(lds, raw, (int)(d.w_total - 1), (u32)__popcll(todo_all));

        // ------------------------------------------------------------ classification of the kept records
        // plain (no I/D/N) records whose speculative 8-byte loads stay inside the SEQ buffer take the
        // fast path: complete ones (every task present) first, then short / contig-edge ones; their
        // scalars go to the staging area.  Everything else walks its CIGAR (phase 2b, done first so that
        // the per-record registers of phase 1 are dead during the fast loops).
        const int rb_lo = (int)(rbase & 0xFFFFFFFFll), rb_hi = (int)(rbase >> 32);
        u64 todo_g = todo_all;
        int nF = 0, nP = 0, nS = 0, nSI = 0;
        bool rsS = false;       // RS, PK: a single-indel record rescaled by its own steps
        if (FAST) {
            // (a record at a contig edge — a flank cut short — walks: the entries of the partial list have complete flanks,
            // so that a partial step's tasks are a prefix of each window)
            const bool plain = kept && (w1 & D_SIMPLE) && sq >= (u32)(8 * d.nl8) &&
                               (i64)sq + nq + 8 * d.nl8 <= a.n_bases &&
                               ((w1 >> D_NB_SHIFT) & 0xFF) == A && ((w1 >> D_NA_SHIFT) & 0xFF) == A;
            const bool isF = plain && (w1 & D_FULL);
            const u64 mF = __ballot(isF), mPp = __ballot(plain && !isF);
            u64 mP = mPp, mS = 0, mSI = 0;
            bool isD = false;
            const int rev = w1 & D_REV;
            uint4 ent;
            ent.x = (u32)(rbase - A + 256);
            // (PK: the SEQ window offset of a lane is its reference window offset plus this — one lane constant less)
            ent.y = PK ? sq - ent.x + pk_dso : sq;
            // (partial entries: the task bytes of each window, A + min(nq, L); complete ones do not look at them)
            ent.z = (u32)nq | ((u32)((A + (nq < L ? nq : L)) * 0x101) << 16);
            ent.w = ((u32)(libid * d.w_lib + d.off_tc() + rev * 4 * 512) << 2) | ((u32)libid << 24) |
                    ((w1 & D_HASQ) ? 0x40000000u : 0u) | ((u32)rev << 31);
            bool gpre = false, isS = false, covered = false;
            if (MDX_PREFIX && __ballot(kept && !(w1 & D_SIMPLE))) {   // (wave-uniform: tiles of plain records skip this)
                // gapped records with complete flanks: the columns of their first / last match run ride the
                // partial-plain list (and the CIGAR walk starts behind them)
                const int dnq = n0 - nq;
                // (with --min-basequal a reference symbol is masked by the read column of the same *left* index,
                // align.py:65-71: behind an N operation that is another base than the one it pairs with from the right
                // end — such records keep the full CIGAR walk, which follows the quirk)
                gpre = kept && !(w1 & D_SIMPLE) && !(MASK && skips) && ((w1 >> D_NB_SHIFT) & 0xFF) == A &&
                       ((w1 >> D_NA_SHIFT) & 0xFF) == A && nq < 32768 && dnq >= -1023 && dnq <= 1023 &&
                       sq >= (u32)(8 * d.nl8 + 16) && (i64)sq + nq + 8 * d.nl8 + 16 <= a.n_bases;
                // ... and those with a single short indel between two match runs are counted by the fast path entirely,
                // one entry each (STEP_G in count())
                // (up to seven bases: three bits of the event word; longer ones keep the CIGAR walk)
                isS = gpre && one && dnq >= -7 && dnq <= 7;
                if (gpre) {
                    w1 |= isS ? (D_PRE | D_ONE) : D_PRE;
                    // (single-indel entries carry the two run lengths, the others the task bytes of each window)
                    ent.z = (u32)nq | 0x8000u | ((u32)(isS ? vlr : vlr + A * 0x101) << 16);
                    // n0 - nq: 11 bits, the low eight in the low byte (all a D_ONE entry needs), the rest in [23:21]
                    ent.w |= ((u32)dnq & 0xFFu) | ((((u32)dnq >> 8) & 7u) << 21) | (isS ? PK_ONE : 0u);
                    if (RS && PK) {
                        // (every column a task of one of the two windows, and each window complete)
                        rsS = isS && rs_code != 0u && ncols >= L && ncols <= 2 * L;
                        if (rsS) ent.w |= (1u << 18) | ((rs_code == 2u ? 1u : 0u) << 20);
                    }
                    // both runs reach --length: the entry covers every task of the record, nothing is left to walk
                    // (an N between two long match runs: a spliced read)
                    covered = !isS && (((vlr & 0xFF) == L && (vlr >> 8) == L) || nonly);
                }
                mP |= __ballot(gpre && !isS);
                mS = __ballot(isS);
                mSI = __ballot(isS && dnq < 0);      // insertions first, then deletions (one kind of step each)
                isD = isS && dnq > 0;
                if (!PK && __ballot(isS && dnq > 0)) {
                    MDX_PH_IN(13);
                    // A single deletion of g bases: the lanes of its entry work in column space and reach query index
                    // min(n0, L) - g - 1 at most; the read bases of the (up to g) composition positions above that,
                    // per side, are counted here (statistics.py:75-83).
                    // (PK, round 6: by the pass of the list that counts the entry — above_deletion in list_pass —, whoever made the entry)
                    if (!PK && isS && dnq > 0) {
                        const int g = dnq, u = vlr & 0xFF, v = vlr >> 8, Lq = nq < L ? nq : L;
                        const int b_cmp = libid * d.w_lib + d.off_cmp() + rev * 2 * L * 4;
                        const int qa = u > L - g ? u : L - g, ia = v > L - g ? v : L - g;
                        if (PK) {
                            // (at most seven bases per side, adjacent in the read: one unaligned load of sixteen nibbles each, both
                            // in flight together — a load per base was fourteen round trips one after the other)
                            const u32 fl_ = sq + (u32)qa, fr_ = sq + (u32)(nq - Lq);
                            const u64 wl = *(const u64_u *)(a.seq + (fl_ >> 1)), wr = *(const u64_u *)(a.seq + (fr_ >> 1));
                            auto nib_cls = [&](const u64 w, const u32 k) -> int {
                                u32 nib = (u32)(w >> (4u * k)) & 15u;
                                if (MASK) { bool m; nib = unmask4(nib, m); }
                                return cls4(nib);
                            };
                            for (int q = qa; q < Lq; q++) {
                                const int sc = nib_cls(wl, (fl_ & 1u) + (u32)(q - qa));
                                if (sc < 4) bump<USE_LDS>(lds, raw, b_cmp + q * 4 + sc);
                            }
                            for (int i = ia; i < Lq; i++) {
                                const int sc = nib_cls(wr, (fr_ & 1u) + (u32)(Lq - 1 - i));
                                if (sc < 4) bump<USE_LDS>(lds, raw, b_cmp + (L + i) * 4 + sc);
                            }
                        } else {
                        for (int q = qa; q < Lq; q++) {
                            const int sc = seq_cls(sq + (u32)q);
                            if (sc < 4) bump<USE_LDS>(lds, raw, b_cmp + q * 4 + sc);
                        }
                        for (int i = ia; i < Lq; i++) {
                            const int sc = seq_cls(sq + (u32)(nq - 1 - i));
                            if (sc < 4) bump<USE_LDS>(lds, raw, b_cmp + (L + i) * 4 + sc);
                        }
                        }
                    }
                    MDX_PH_OUT();
                }
            }
            nF = __popcll(mF); nP = __popcll(mP); nS = __popcll(mS); nSI = __popcll(mSI);
            todo_g = todo_all & ~(mF | mPp | mS | __ballot(covered));
            if (mF) {
                // complete records (soft-clipped ones, mostly): counted behind the tile loop like the other lists
                if (isF) ringC[(lC + (u32)mbcnt64(mF, 0)) & RM] = ent;
                lC += (u32)nF;
            }
            if (mP | mS) {
                // The steps of these records zero the reference bytes that are not their tasks (or, behind a deletion,
                // are counted by position) and count all eight bytes of a lane into TC with one increment: those bytes
                // land in plane A.  DMP says, as differences over the window bytes of a side, how many such bytes
                // there are (finalize_kernel takes the prefix sums off the A counts): +1 where a stretch begins, -1
                // where it ends before the window does.
                // (PK: a nibble that is not a task is zero and counts nothing — no correction)
                if (USE_LDS && !PK) {
                    const int dl = libid * d.w_lib + d.off_dmp() + rev * 2 * (A + L), dr = dl + (A + L);
                    if (plain && !isF) {
                        // tasks [-A, min(nq, L)) per side
                        const int k1 = nq < L ? nq : L;
                        if (k1 < L) { atomicAdd(&lds[dl + A + k1], 1u); atomicAdd(&lds[dr + A + k1], 1u); }
                    } else if (gpre && !isS) {
                        // tasks [-A, first / last match run)
                        const int vl = vlr & 0xFF, vr = vlr >> 8;
                        if (vl < L) atomicAdd(&lds[dl + A + vl], 1u);
                        if (vr < L) atomicAdd(&lds[dr + A + vr], 1u);
                    } else if (isS) {
                        // an insertion counts its columns up to min(ncols, L) through TC; a deletion those in front
                        // of and in the gap — the columns behind it go to MIS / CMP by position
                        const int g = n0 - nq, u = vlr & 0xFF, v = vlr >> 8;
                        const int Lm = ncols < L ? ncols : L;
                        const int el = g > 0 ? (u + g < Lm ? u + g : Lm) : Lm, er = g > 0 ? (v + g < Lm ? v + g : Lm) : Lm;
                        if (el < L) atomicAdd(&lds[dl + A + el], 1u);
                        if (er < L) atomicAdd(&lds[dr + A + er], 1u);
                    }
                }
                if ((plain && !isF) || (gpre && !isS)) ringP[(lP + (u32)mbcnt64(mP, 0)) & RM] = ent;
                else if (isS) {
                    const u32 at = isD ? (lD + (u32)mbcnt64(mS & ~mSI, 0)) & RM : (lI + (u32)mbcnt64(mSI, 0)) & RM;
                    (isD ? ringD : ringI)[at] = ent;
                    if (RS && PK) (isD ? lriD : lriI)[at] = ri;      // (the record of a single-indel entry: where its MR goes)
                }
                lP += (u32)nP; lI += (u32)nSI; lD += (u32)(nS - nSI);
            }
        }

        if (RS && PK) {
            if (rsS) p.rs.status[ri] = (u8)(rs_code + 1u);     // (rescale.py:300-342: 2 = unpaired, 3 = an inward pair's mate)
            const bool lst = valid && rs_code != 0u && !rsS;
            const u64 mW = __ballot(lst);
            if (mW) {
                if (lst) (p.rs.gen_list + (size_t)gwave * (size_t)p.list_cap)[n_rs + (u32)mbcnt64(mW, 0)] = ri;
                n_rs += (u32)__popcll(mW);
            }
        }
        // ------------------------------------------------------------ phase 2b: gapped records
        MDX_PH_IN(12);
        while (todo_g) {
            const int j = __ffsll((long long)todo_g) - 1;
            todo_g &= todo_g - 1;
            const int s_w1 = rl(w1, j);
            const int s_nq = rl(nq, j);
            const int s_n0 = rl(n0, j);
            const int s_ncols = rl(ncols, j);
            const int s_nrg = s_n0 + rl(nI, j);
            const u32 s_co = (u32)rl((int)cig_o, j);
            const int s_cn = rl(cig_n, j);
            // the record's CIGAR, one op per lane (a single coalesced load); the walks below read it with
            // v_readlane instead of a dependent memory load per op (records with > 64 ops re-read memory)
            const u32 op_lane = lane < s_cn ? a.cigar[s_co + lane] : 0u;
            auto op_at = [&](int k) -> u32 { return s_cn <= 64 ? (u32)rl((int)op_lane, k) : a.cigar[s_co + k]; };
            const i64 s_rbase = ((i64)rl(rb_hi, j) << 32) | (u32)rl(rb_lo, j);
            const u32 s_sq = (u32)rl((int)sq, j);
            const int rev = s_w1 & D_REV;
            const bool hasq = MASK && (s_w1 & D_HASQ);
            const int lb = rl(libid, j) * d.w_lib;
            const int b_mis = lb + d.off_mis() + rev * 2 * L * 25;
            const int b_cmp = lb + d.off_cmp() + rev * 2 * L * 4;
            const int b_tc = lb + d.off_tc() + rev * 4 * d.t_pad;
            const u8 *__restrict__ qp = (MASK && !PK) ? a.qual + s_sq : nullptr;
            // flank lengths: from the packed descriptor (A < 248 with the fast path), else recomputed
            int s_nb = (s_w1 >> D_NB_SHIFT) & 0xFF, s_na = (s_w1 >> D_NA_SHIFT) & 0xFF;
            if (!FAST) {
                const u32 rk = (u32)rl((int)ri, j);
                const i64 pos = a.pos[rk];
                const int tid = a.tid[rk];
                const i64 clen = a.contig_off[tid + 1] - a.contig_off[tid];
                s_nb = pos < A ? (int)pos : A;
                s_na = clen - (pos + s_n0) < A ? (int)(clen - (pos + s_n0)) : A;
            }

            // The columns of the first match run (from the left) and of the last one (from the right) are
            // plain: same pairing, and the same position in the misincorporation and in the composition
            // table, as in an ungapped record.  Such a record (D_PRE) also has an entry in the partial-plain
            // list, which counts those columns and the flanks in the fast path; the walks below start
            // behind them.
            int s_vl = 0, s_vr = 0;
            const bool pre_done = FAST && (s_w1 & D_PRE);
            if (pre_done) {
                const int s_vlr = rl(vlr, j);
                s_vl = s_vlr & 0xFF; s_vr = s_vlr >> 8;
            }

            // misincorporation pairs, each string indexed from its own end (main.py:210-212)
            int Lm = s_ncols < s_nrg ? s_ncols : s_nrg;
            if (Lm > L) Lm = L;
            const int nml = Lm - s_vl > 0 ? Lm - s_vl : 0, nmr = Lm - s_vr > 0 ? Lm - s_vr : 0;
            for (int t = lane; t < nml + nmr; t += 64) {
                const int side = t >= nml;
                const int i = side ? s_vr + (t - nml) : s_vl + t;
                const int js = side ? s_ncols - 1 - i : i;
                const int jr = side ? s_nrg - 1 - i : i;
                // walk the CIGAR: query index under gapped-read column js (-1 = deletion gap),
                // reference index under gapped-reference column jr (-1 = insertion gap)
                int col = 0, qoff = 0, shift = 0, qi = -2, rix = -2;
                for (int k = 0; k < s_cn; k++) {
                    const u32 cg = op_at(k);
                    const int op = cg & 0xF;
                    const int len = (int)(cg >> 4);
                    if (op == 0 || op == 7 || op == 8) {
                        if (qi == -2 && js < col + len) qi = qoff + (js - col);
                        col += len; qoff += len;
                    } else if (op == 1) {
                        if (qi == -2 && js < col + len) qi = qoff + (js - col);
                        if (rix == -2) {
                            if (jr < col) rix = jr - shift;
                            else if (jr < col + len) rix = -1;
                        }
                        shift += len; col += len; qoff += len;
                    } else if (op == 2) {
                        if (qi == -2 && js < col + len) qi = -1;
                        col += len;
                    }
                }
                if (rix == -2) rix = jr - shift;
                int s = qi < 0 ? SYM_GAP : seq_cls(s_sq + (u32)qi);
                int r = rix < 0 ? SYM_GAP : ref_cls(s_rbase + rix);
                if (hasq) {
                    const bool ms = qi >= 0 && (PK ? seq_msk(s_sq + (u32)qi) : (int)qp[qi] < a.minqual);
                    bool mr = ms;
                    if (jr != js) {
                        // mask of the *reference* column jr follows the read column jr
                        mr = false;
                        if (jr < s_ncols) {
                            int c2 = 0, q2 = 0, qj = -2;
                            for (int k = 0; k < s_cn && qj == -2; k++) {
                                const u32 cg = op_at(k);
                                const int op = cg & 0xF;
                                const int len = (int)(cg >> 4);
                                if (op == 0 || op == 7 || op == 8 || op == 1) {
                                    if (jr < c2 + len) qj = q2 + (jr - c2);
                                    c2 += len; q2 += len;
                                } else if (op == 2) {
                                    if (jr < c2 + len) qj = -1;
                                    c2 += len;
                                }
                            }
                            mr = qj >= 0 && (PK ? seq_msk(s_sq + (u32)qj) : (int)qp[qj] < a.minqual);
                        }
                    }
                    if (ms) s = SYM_OTHER;
                    if (mr) r = SYM_OTHER;
                }
                if (s <= SYM_GAP && r <= SYM_GAP) {
                    // the reference-base count of a mismatching column is derived at finalisation
                    const int row = b_mis + (side * L + i) * 25;
                    if (r != s) bump<USE_LDS>(lds, raw, row + mis_col(r, s));
                    else if (r != SYM_GAP) bump<USE_LDS>(lds, raw, row + r);
                }
            }
            // read composition on the ungapped, unmasked query (statistics.py:75-83)
            const int Lq = s_nq < L ? s_nq : L;
            const int nql = Lq - s_vl > 0 ? Lq - s_vl : 0, nqr = Lq - s_vr > 0 ? Lq - s_vr : 0;
            for (int t = lane; t < nql + nqr; t += 64) {
                const int side = t >= nql;
                const int k0 = side ? s_vr + (t - nql) : s_vl + t;
                const int s = seq_cls(s_sq + (u32)(side ? s_nq - 1 - k0 : k0));
                if (s < 4) bump<USE_LDS>(lds, raw, b_cmp + (side * L + k0) * 4 + s);
            }
            // flanks (statistics.py:85-93) go to their task slots
            for (int t = lane; !pre_done && t < 2 * A; t += 64) {
                const int side = t >= A;
                const int dist = (side ? t - A : t) + 1;
                if (dist <= (side ? s_na : s_nb)) {
                    const int r = ref_cls(s_rbase + (side ? s_n0 - 1 + dist : -dist));
                    if (PK) {
                        // the packed kernel's TC table: [base][64 j + lane], the lane of the strand's first slot that owns window
                        // nibble A - dist of the side (right side: counted from the window's outer end, nibble 15 - j)
                        const int wn = A - dist, ln4 = (rev ? d.H4 : 0) * d.G4 + side * d.nl16 + (wn >> 4);
                        if (r < 4) atomicAdd(&lds[d.off_tc() + (r << 10) + 64 * (side ? 15 - (wn & 15) : (wn & 15)) + ln4], 1u);
                    } else
                    if (r < 4) bump<USE_LDS>(lds, raw, b_tc + r * d.t_pad + (side ? d.tau_rflank(dist) : d.tau_lflank(dist)));
                }
            }
        }
        MDX_PH_OUT();

    };   // general

    // (ML: a pool counts ONE library of the launch — MdxTabArgs::ml_plan, made on the device from the libraries' sizes: pool p
    // is the ml_k-th of the ml_m pools of library ml_lib, which share that library's tiles among themselves as the pools of a
    // one-library launch share the batch's; see the template's comment)
    {
    if (ML) {
        karg_p kp = ka;
        asm volatile("" : "+s"(kp));
        // (a kept record whose library the context does not know was given no place by the sort: reported here, like the
        // record the one-library kernel finds)
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const u64 v = *kp->sort_bad;
            if (v != ~0ull) flag_error(kp->err, (i64)(v >> 8) + kp->record_base, (int)(v & 0xFFu));
        }
        const uint4 pl = kp->ml_plan[blockIdx.x % mdx_n_pools(gridDim.x)];
        ml_lib = (int)__builtin_amdgcn_readfirstlane((int)pl.x);
        ml_k = (u32)__builtin_amdgcn_readfirstlane((int)pl.y);
        ml_m = (u32)__builtin_amdgcn_readfirstlane((int)pl.z);
        ml_first = (u32)__builtin_amdgcn_readfirstlane((int)pl.w);
        rec_lo = kp->lib_start[a.lib_lo + ml_lib];
        n_rec = kp->lib_start[a.lib_lo + ml_lib + 1] - rec_lo;
    }
    if (!FAST) {
        for (u32 it = 0; it < n_it; it++) {
            const u32 tbase = it < rounds ? (it * nwaves + gwave) * T : t_lo + (it - rounds) * T;
            const u32 r_hi = it < rounds ? tbase + T : (tbase + T < t_hi ? tbase + T : t_hi);
            // the per-record columns are requested together, before the flag is known (one memory round
            // trip for the tile instead of two)
            const u32 ri = tbase + lane;
            const bool valid = ri < r_hi;
            const u32 rj = valid ? ri : r_hi - 1;
            general(ri, valid, (u32)ld32(a.flag, rj), ld32(a.lib, rj), ld32(a.tid, rj), ld32(a.pos, rj), ld32(a.tlen, rj),
                    ld32(a.cigar_off, rj), ld32(a.cigar_off, rj + 1), ld32(a.seq_off, rj), ld32(a.seq_off, rj + 1));
        }
    } else {
        // The tile loop proper takes the records whose CIGAR is a single match operation (aDNA: most of them) with
        // a short phase 1 of its own — no CIGAR scan, no clips, no gaps — counts the complete ones of the tile at
        // once, and leaves every other kept record to the general pass: its index goes to a list, and as soon as 64
        // of them wait they are classified together, every lane busy (a tile of 63 records holds a dozen such records
        // at most, which used to drag the whole wavefront through the general code once per tile).  Wavefronts reach
        // their general passes at different moments, under the counting of the others.
        int nDef = 0, dDone = 0;
        u32 n_kept_lite = 0;
        u32 nb0 = 0, nb1 = 0;   // RS: byte range of the next tile's qualities (requested a tile ahead)
        // Tiles are handed out on demand within a *pool*: the two blocks that share a CU (blocks p and p + gridDim / 2: the
        // dispatcher places the first gridDim / 2 blocks one per CU, then the second half) own a contiguous stretch of the
        // tiles and one counter (MdxTabArgs::tile_ctr).  A CU's arbiters favour its older wavefronts: with the tiles dealt
        // out in advance the block dispatched first ran 30 % ahead of the other, and within a block the older wavefronts
        // ahead of the younger — the launch ended with a third of its time spent on half-empty CUs
        // (tools/experiments/wave_clk.py).  A wavefront asks for a tile two tiles before it starts it (the answer comes
        // back under a tile's work) and — RS, whose list of records left to the rescale kernels is sized for that many — takes
        // at most tile_quota of them (the quotas of a pool's wavefronts add up to twice its tiles).
        const u32 n_tiles = (n_rec + T - 1) / T;
        const u32 n_pools = mdx_n_pools(gridDim.x);
        const u32 pool = blockIdx.x % n_pools;
        // (a pool's tiles: chunks of MDX_POOL_CHUNK consecutive tiles, the pools' chunks interleaved — the whole chip works
        // on one neighbourhood of a coordinate-sorted batch at a time and shares its reference lines in the L2s, as it did
        // when the tiles were dealt round-robin; a stretch of its own per pool cost such a batch 5 %)
        u32 grabs = 0;
        // (STEAL — the packed kernels: a wavefront whose pool has run dry goes on with the tiles of other pools — ML: of its
        // library's —, see tile_of; pool_cur = the pool it asks at present.  A stolen tile takes the place of the answer
        // that found the pool empty: the quota counts it once)
        constexpr bool STEAL = MDX_PK_STEAL && PK;
        u32 pool_cur = pool;
        // (the pools that share these tiles: all of them, or — ML — the ml_m pools of the library, from pool ml_first on)
        const u32 pm = ML ? ml_m : n_pools, pf = ML ? ml_first : 0u;
        auto grab = [&]() -> u32 {
            // (lane 0 asks; the value is read — readfirstlane — where it is first needed)
            u32 v = 0xFFFFFFFFu;
            if (grabs < (u32)a.tile_quota) { if (lane == 0) v = atomicAdd(a.tile_ctr + (STEAL ? pool_cur : pool) * MDX_CTR_PAD, 1u); grabs++; }
            return v;
        };
        const u32 ch_first = pool - pf;
        auto tile_of = [&](const u32 raw) -> u32 {
            u32 v = (u32)__builtin_amdgcn_readfirstlane((int)raw);
            if (v == 0xFFFFFFFFu) return v;
            if (STEAL) {
                // The pools do not finish together — a CU's blocks run up to 5 % slower or faster than the chip's mean
                // (tools/experiments/wave_clk.py), on some boxes of the pool far more — and a launch lasts as long as its
                // slowest pool: a wavefront that finds its pool empty asks the counters of two others, 37 pools apart (a
                // pool's answer says whether it has a tile left), and stays with the first that has.  One answer is
                // outstanding at a time, asked of pool_cur and read here: a wavefront never reads an answer under another
                // pool's name.  Two boxes, 25 M config-3 records: 1.125 -> 1.122 ms on one, 1.61 -> 1.25 ms on the other (six
                // pools asked: 1.135 on the first — every wavefront's last act is then six round trips for nothing).
                u32 tile = 0xFFFFFFFFu;
#pragma unroll 1
                for (int tries = 0;; tries++) {
                    const u32 ch = v / MDX_POOL_CHUNK, t = (ch * pm + (pool_cur - pf)) * MDX_POOL_CHUNK + (v - ch * MDX_POOL_CHUNK);
                    if (t < n_tiles) { tile = t; break; }
                    if (tries == MDX_PK_STEAL) break;
                    pool_cur = pf + (pool_cur - pf + 37u) % pm;
                    u32 r = 0u;
                    if (lane == 0) r = atomicAdd(a.tile_ctr + pool_cur * MDX_CTR_PAD, 1u);
                    v = (u32)__builtin_amdgcn_readfirstlane((int)r);
                }
                return tile;
            }
            const u32 ch = v / MDX_POOL_CHUNK, tile = (ch * pm + ch_first) * MDX_POOL_CHUNK + (v - ch * MDX_POOL_CHUNK);
            return tile < n_tiles ? tile : 0xFFFFFFFFu;
        };
        // (the fused kernel knows its next tile while it works on one — the bounds of that tile's quality copy are requested
        // a tile ahead — and asks for the one after; the others ask for the next one: a wavefront that finds its pool
        // empty has one tile less left to do)
        u32 cur = tile_of(grab());
        u32 nxt = (RS || pfl) && cur != 0xFFFFFFFFu ? tile_of(grab()) : 0xFFFFFFFFu;
        // (pfl: the first tile's columns and second round trip, one after the other — every later tile's come under the tile
        // in front of it.  Rounds 4-6 had the same prefetch into registers behind MDX_PK_PREFETCH: 26 of them, spilled — DESIGN 4)
        if (pfl && cur != 0xFFFFFFFFu) {
            pfl_cols(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pfl_rt2(cur);
        }
        if (RS && cur != 0xFFFFFFFFu) {
            const u32 tb0 = cur * T, rh0 = tb0 + T < n_rec ? tb0 + T : n_rec;
            nb0 = ld32(a.seq_off, tb0); nb1 = ld32(a.seq_off, rh0);
        }
        // One pass over a list: up to TL entries from `head` on (whole steps: all but the last pass of a list), staged and
        // counted by a run of their kind.  Returns the entries taken.
        const int TL = 64 - 64 % d.R;
        auto list_pass = [&](const uint4 *ring, const u32 *ridx, const u32 head, const u32 avail, auto kind_tag) -> u32 {
            const int m = avail < (u32)TL ? (int)avail : TL;
            const u32 at = (head + (u32)(lane < m ? lane : 0)) & RM;
            const uint4 ent = ring_at(ring + at);
            if (PK && decltype(kind_tag)::value == STEP_GD) {
                // A single deletion of g bases: the lanes of its entry work in column space and reach query index min(n0, L) - g - 1
                // at most; the read bases of the (up to g) composition positions above that, per side, are counted here
                // (statistics.py:75-83), from the entry alone — whoever made it, the general pass or a tile's phase 1.  At most
                // seven bases per side, adjacent in the read: one unaligned load of sixteen nibbles each, both in flight together.
                MDX_PH_IN(13);
                const int g = (int)(i8)(ent.w & 0xFFu), nq_ = (int)(ent.z & 0x7FFFu), u = (int)((ent.z >> 16) & 0xFFu), v = (int)(ent.z >> 24);
                const int Lq = nq_ < L ? nq_ : L;
                const int qa = u > L - g ? u : L - g, ia = v > L - g ? v : L - g;
                const bool any = lane < m && g > 0 && (qa < Lq || ia < Lq);
                if (__ballot(any)) {
                    if (any) {
                        const int b_cmp = __mul24((int)((ent.w >> 24) & 0x3Fu), d.w_lib) + d.off_cmp() + (int)(ent.w >> 31) * 2 * L * 4;
                        const u32 sq_ = ent.y + ent.x - pk_dso;
                        const u32 fl_ = sq_ + (u32)qa, fr_ = sq_ + (u32)(nq_ - Lq);
                        const u64 wl = *(const u64_u *)(a.seq + (fl_ >> 1)), wr = *(const u64_u *)(a.seq + (fr_ >> 1));
                        auto nib_cls = [&](const u64 w, const u32 k) -> int {
                            u32 nib = (u32)(w >> (4u * k)) & 15u;
                            if (MASK) { bool mk; nib = unmask4(nib, mk); }
                            return cls4(nib);
                        };
                        for (int q = qa; q < Lq; q++) {
                            const int sc = nib_cls(wl, (fl_ & 1u) + (u32)(q - qa));
                            if (sc < 4) bump<USE_LDS>(lds, raw, b_cmp + q * 4 + sc);
                        }
                        for (int i = ia; i < Lq; i++) {
                            const int sc = nib_cls(wr, (fr_ & 1u) + (u32)(Lq - 1 - i));
                            if (sc < 4) bump<USE_LDS>(lds, raw, b_cmp + (L + i) * 4 + sc);
                        }
                    }
                }
                MDX_PH_OUT();
            }
            int n_fwd = -1;
            int sidx = 0;       // (PK: where this lane's entry is staged)
            if (PK) {
                // (sorted by strand, the forward entries first)
                const bool mine = lane < m, rv_ = (ent.w >> 31) != 0u;
                const u64 mR = __ballot(mine && rv_), mW = __ballot(mine && !rv_);
                n_fwd = __popcll(mW);
                sidx = rv_ ? n_fwd + mbcnt64(mR, 0) : mbcnt64(mW, 0);
                // (a complete record's w is the copy of the reference its window is read from, see phase 1: the first one for
                // the few such records the general pass finds)
                uint4 e2 = ent;
                if (!RS && decltype(kind_tag)::value == STEP_C) e2.w = 0u;
                if (mine) stg[sidx] = e2;
            } else {
                if (lane < m) stg[lane] = ent;
                if (lane < d.R - 1) stg[m + lane] = ent;    // (lanes 0 .. R-2 hold real entries: m > 0)
            }
            constexpr bool RSP = RS && decltype(kind_tag)::value == STEP_P;
            // (PK: the fused single-indel entries — bit 18 —, their records in lriI / lriD by the entry's place in its ring)
            constexpr bool RSG = RS && PK && (decltype(kind_tag)::value == STEP_GI || decltype(kind_tag)::value == STEP_GD);
            u32 ri_l = 0;
            if (RSG || RSP) {
                ri_l = ring_u32(ridx + at);
                mrm[lane] = 0ull;
                if (lane < MDX_FUSE_MRM - 64) mrm[64 + lane] = 0ull;
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);    // (see the tile loop's run)
            MDX_PH_IN(14);
            run(0, m, kind_tag, std::true_type{}, n_fwd);
            MDX_PH_OUT();
            if (RSP || RSG) rsq_flush();
            if ((RSG || (RSP && PK)) && lane < m && ((ent.w >> 18) & 1u)) a.rs.mr_raw[ri_l] = mr_of(mrm[sidx]);
            // RS: the MR sums of the fused records among them (known by their TC table)
            if (RSP && !PK && lane < m && ((ent.w & 0x3FF00u) >> 2) >= (u32)a.rs.tcb_off) a.rs.mr_raw[ri_l] = mr_of(mrm[lane]);
            return (u32)m;
        };
        // A wavefront works in ROUNDS of at most round_tiles tiles: the tile loop, then its lists (the rings hold a round's
        // entries: MDX_LIST_RING >= 63 round_tiles + 63), then the next round while its pool has tiles.
        for (;;) {
        u32 tiles_left = ROUNDS ? (u32)a.round_tiles : 0xFFFFFFFFu;
        for (;;) {
            // (over: the pool is empty — the records still waiting for the general pass are seen to, then the lists; past: no tile
            // in this turn — that, or the round is over: its lists, and the records that wait go on waiting)
            const bool over = cur == 0xFFFFFFFFu;
            const bool past = over || tiles_left == 0u;
            int nF = 0, nF0 = 0, nFp = 0;
            u32 nxt2_raw = 0xFFFFFFFFu;
            if (!past) {
                MDX_PH(1);
                tiles_left--;
                // pfl: the tile's columns and second round trip out of the prefetch area (requested a tile ago: landed), then the
                // next tile's columns into it
                u32 P_fl = 0u, P_lib = 0u, P_tid = 0u, P_pos = 0u, P_tlen = 0u, P_co0 = 0u, P_co1 = 0u, P_so0 = 0u, P_so1 = 0u;
                u32 P_g0 = 0u, P_g1 = 0u, P_g2 = 0u, P_c0 = 0u, P_c1 = 0u;
                if (pfl) {
                    MDX_PH(11);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    // (... said again for the compiler, which does not read asm statements: nothing is on its way from here, and
                    // the runs of this tile count their waits)
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                    const lds_u1 *const C = (const lds_u1 *)(size_t)(pfl_a + 4u * (u32)lane);
                    P_fl = C[0]; if (!ML) P_lib = C[64];
                    P_tid = C[128]; P_pos = C[192]; P_tlen = C[256]; P_co0 = C[320]; P_co1 = C[321]; P_so0 = C[384]; P_so1 = C[385];
                    P_g0 = C[448]; P_g1 = C[512]; P_g2 = C[576]; P_c0 = C[640]; P_c1 = C[704];
                    // (read before the next tile's columns may land on them)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    MDX_PH(1);
                }
                if (!(RS || pfl) || nxt != 0xFFFFFFFFu) nxt2_raw = grab();
                if (pfl && nxt != 0xFFFFFFFFu) pfl_cols(nxt);
                karg_p kp = ka;
                asm volatile("" : "+s"(kp));
                const __attribute__((address_space(4))) MdxTabArgs &p = *kp;
                const u32 tbase = cur * T;
                const u32 r_lo = tbase;
                const u32 r_hi = tbase + T < n_rec ? tbase + T : n_rec;
                // (the packed fused kernel: one unit — its registers are the tighter: config 5 2.79 -> 2.71 ms on one box)
                constexpr int FCPU = PK ? MDX_PKF_CPU : MDX_FUSE_CPU;
                u32x4 cpv[FCPU];
                u32 cp_a0 = 0, cp_nu = 0, cp_b0 = 0, cp_b1 = 0;
                if (RS) {
                    // qual_out starts as a copy of the quality column: the tile's own stretch (its bounds were requested a
                    // tile ahead), 16 bytes per lane, before any lane of this wavefront stores a rescaled byte into it (the
                    // up to 15 bytes in front of and behind the 16-byte units are moved byte by byte: the stretch belongs
                    // to this tile alone).  The loads of the first pass go out in front of the tile's column loads, its
                    // stores behind them: one round trip for both.
                    MDX_PH(15);
                    cp_b0 = nb0; cp_b1 = nb1;
                    // (patch mode: there is no second column — nothing to copy)
                    if (p.rs.patch) cp_b0 = cp_b1;
                    // (units at the same 16-byte phase as the source column: both columns are 16-byte aligned in any
                    // allocation this library is handed)
                    const u32 head = (0u - (cp_b0 + (u32)((size_t)p.qual & 15))) & 15u;
                    cp_a0 = cp_b0 + head < cp_b1 ? cp_b0 + head : cp_b1;
#ifndef MDX_RSABL_NOCOPY
                    cp_nu = (cp_b1 - cp_a0) >> 4;
#endif
                    const u8 *__restrict__ qin = p.qual;
#pragma unroll
                    for (int k = 0; k < FCPU; k++) {
                        const u32 u = (u32)lane + 64u * k;
                        if (u < cp_nu) cpv[k] = cp_load(qin + (cp_a0 + 16u * u));
                    }
                    if (nxt != 0xFFFFFFFFu) {
                        const u32 tb2 = nxt * T, rh2 = tb2 + T < n_rec ? tb2 + T : n_rec;
                        nb0 = ld32(a.seq_off, tb2); nb1 = ld32(a.seq_off, rh2);
                    }
                }
                // ---------------------------------------------------- phase 1 of the single-match records
                // (ML: place rec_lo + ... of the batch ordered by library; the record's library is the pool's)
                const u32 ri = r_lo + lane + (ML ? rec_lo : 0u);
                const bool valid = r_lo + lane < r_hi;
                const u32 rj = valid ? ri : tbase + (ML ? rec_lo : 0u);
                u32 fl;
                int c_lib, c_tid, c_pos, c_tlen;
                u32 c_co0, c_co1, c_so0, c_so1;
                if (pfl) {
                    fl = valid ? P_fl : 0x4u;
                    c_lib = ML ? a.lib_lo + ml_lib : (int)P_lib; c_tid = (int)P_tid; c_pos = (int)P_pos; c_tlen = (int)P_tlen;
                    c_co0 = P_co0; c_co1 = P_co1; c_so0 = P_so0; c_so1 = P_so1;
                } else {
                    fl = valid ? (u32)ld32(a.flag, rj) : 0x4u;
                    c_lib = ML ? a.lib_lo + ml_lib : ld32(a.lib, rj); c_tid = ld32(a.tid, rj); c_pos = ld32(a.pos, rj); c_tlen = ld32(a.tlen, rj);
                    c_co0 = ld32(a.cigar_off, rj); c_co1 = ld32(a.cigar_off, rj + 1); c_so0 = ld32(a.seq_off, rj); c_so1 = ld32(a.seq_off, rj + 1);
                }
                int c_mtid = 0, c_mpos = 0;
                bool rs_anyhi = false;       // RS: some quality byte of the tile has bit 7 set (0xFF: a record without qualities)
                if (RS) {
                    c_mtid = ld32(p.rs.mtid, rj); c_mpos = ld32(p.rs.mpos, rj);
                    // (the stores of the copy's first pass; then the passes a tile of long records needs beyond it)
                    const u8 *__restrict__ qin = p.qual;
                    u8 *__restrict__ qout = p.rs.qual_out;
                    u32 hi = 0;
#pragma unroll
                    for (int k = 0; k < FCPU; k++) {
                        const u32 u = (u32)lane + 64u * k;
                        if (u < cp_nu) {
                            cp_store(qout + (cp_a0 + 16u * u), cpv[k]);
                            hi |= cpv[k].x | cpv[k].y | cpv[k].z | cpv[k].w;
                        }
                    }
#if MDX_FUSE_CP2
                    // (two units per lane and round trip)
                    for (u32 u = 64u * FCPU + (u32)lane; u < cp_nu; u += 128u) {
                        const u32x4 v = cp_load(qin + (cp_a0 + 16u * u));
                        u32x4 v2 = u32x4{0u, 0u, 0u, 0u};
                        const bool two = u + 64u < cp_nu;
                        if (two) v2 = cp_load(qin + (cp_a0 + 16u * (u + 64u)));
                        cp_store(qout + (cp_a0 + 16u * u), v);
                        if (two) cp_store(qout + (cp_a0 + 16u * (u + 64u)), v2);
                        hi |= v.x | v.y | v.z | v.w | v2.x | v2.y | v2.z | v2.w;
                    }
#else
                    for (u32 u = 64u * FCPU + (u32)lane; u < cp_nu; u += 64u) {
                        const u32x4 v = cp_load(qin + (cp_a0 + 16u * u));
                        cp_store(qout + (cp_a0 + 16u * u), v);
                        hi |= v.x | v.y | v.z | v.w;
                    }
#endif
                    if (cp_b0 + (u32)lane < cp_a0) { const u8 b = qin[cp_b0 + (u32)lane]; qout[cp_b0 + (u32)lane] = b; hi |= b; }
                    const u32 t0 = cp_a0 + 16u * cp_nu + (u32)lane;
                    if (t0 < cp_b1) { const u8 b = qin[t0]; qout[t0] = b; hi |= b; }
                    // (patch mode: no pass over the tile's qualities has been made — every record looks at its first one)
#ifdef MDX_RSABL_NOQF
                    rs_anyhi = __ballot((hi & 0x80808080u) != 0u) != 0ull;
#else
                    rs_anyhi = p.rs.patch != nullptr || __ballot((hi & 0x80808080u) != 0u) != 0ull;
#endif
                    MDX_PH(1);
                }
                bool kept = (fl & 0xF04u) == 0;  // reader.py:121-132
                if (!ML && c_lib < a.nlib_total && (c_lib < a.lib_lo || c_lib >= a.lib_lo + d.nlib)) kept = false;
                // second round trip of the tile: the (up to three) operations, the contig bounds and (MASK) the first quality
                // together.  The tile loop's own records: one match operation (M, = or X), alone or between soft clips —
                // [S] M [S] — over the whole of SEQ.
                const u32 cn = c_co1 - c_co0;
                const bool cand = kept && cn - 1u < 3u && c_tid >= 0 && c_tid < a.n_contig && c_lib < a.nlib_total;
                u32 g0 = 0xFu, g1 = 0xFu, g2 = 0xFu;                  // (an operation code that is neither)
                // (32-bit reference coordinates: the fast path runs on references shorter than 4 GiB, MdxTabArgs::ref32)
                u32 c0 = 0, clen = 0;
                u32 q0 = 0xFFu;
                if (pfl) {
                    // (the lanes that asked: pfl_rt2 made the same test on the same columns)
                    if (cand) { g0 = P_g0; c0 = P_c0; clen = P_c1 - P_c0; }
                    if (cand && cn >= 2u) g1 = P_g1;
                    if (cand && cn >= 3u) g2 = P_g2;
                }
                else if (cand) {
                    g0 = a.cigar[c_co0];
                    if (cn >= 2u) g1 = a.cigar[c_co0 + 1];
                    if (cn >= 3u) g2 = a.cigar[c_co0 + 2];
                    c0 = (u32)a.contig_off[c_tid];
                    clen = (u32)a.contig_off[c_tid + 1] - c0;
                    if (MASK && !PK && a.qual != nullptr) q0 = a.qual[c_so0];
                }
                u32 rs_qf = 0xFFu;          // RS: the record's first quality (0xFF: none, rescale.py:306)
                if (RS) {
                    // (the copy has seen every quality byte of the tile: without a byte above 127 among them every record
                    // has qualities, and the 63 scattered loads are not needed)
                    // (... nor for a record that brings MDX_FLAG_HAS_QUAL)
                    rs_qf = 0u;
                    if (rs_anyhi && valid && c_so1 != c_so0 && !(fl & 0x4000u)) rs_qf = p.qual[c_so0];
                }
                const bool m0 = ((0x181u >> (g0 & 0xFu)) & 1u) != 0, m1 = ((0x181u >> (g1 & 0xFu)) & 1u) != 0;
                const bool s0 = (g0 & 0xFu) == 4u, s1 = (g1 & 0xFu) == 4u, s2 = (g2 & 0xFu) == 4u;
                // S M / S M S: the match is the second operation; M / M S: the first
                const bool lead = s0 && m1 && (cn == 2u || (cn == 3u && s2));
                const bool shape = lead || (m0 && (cn == 1u || (cn == 2u && s1)));
                const u32 qs = lead ? g0 >> 4 : 0u;                                  // leading clip
                const u32 tr = cn == 3u ? g2 >> 4 : ((cn == 2u && !lead) ? g1 >> 4 : 0u);   // trailing clip
                const u32 len = (lead ? g1 : g0) >> 4;                               // the match
                const u32 sq = c_so0 + qs;                                           // first aligned base (pysam's query)
                // ... inside the contig with both flanks complete (a record at a contig edge walks), and the speculative
                // window loads inside the SEQ buffer; anything else (and anything wrong) is the general pass's
                // (all in 32 bits: a clip of 2^28 bases and more is not this loop's, the match is shorter than 2^15, pos is
                // below 2^31, and n_bases is what a 32-bit seq_off column can address)
                const u32 lseq = c_so1 - c_so0, nb32 = (u32)a.n_bases, pad8 = (u32)(8 * d.nl8);
                const bool triv = cand && shape && ((qs | tr) >> 28) == 0u && qs + len + tr == lseq && len - 1u < 32767u && c_pos >= A &&
                                  (u32)c_pos + len + (u32)A <= clen && sq >= pad8 && len + pad8 <= nb32 && sq <= nb32 - len - pad8;
                // RS: record routing (rescale.py:300-342) — 0 unmapped, 1 no qualities, 2 unpaired, 3 the mate of an inward pair
                // (rescaled from its 5' end only), 4 any other pair
                int rs_st = 0, rs_fwd = 0;
                bool want = false;
                if (RS) {
                    const int rev_ = (fl >> 4) & 1, mate_rev = (fl >> 5) & 1;
                    if (fl & 0x4u) rs_st = 0;
                    else if (c_so1 == c_so0 || rs_qf == 0xFFu) rs_st = 1;
                    else if (fl & 0x1u) {
                        const bool same = c_tid == c_mtid;
                        if ((!rev_ && mate_rev && c_mpos > c_pos && same) || (rev_ && !mate_rev && c_mpos < c_pos && same)) { rs_st = 3; rs_fwd = 1; }
                        else rs_st = 4;
                    } else rs_st = 2;
                    want = valid && (rs_st == 2 || rs_st == 3);
                }
                // (PK: a record of the general pass that is to be rescaled says so there — see general())
                const u32 rs_def = (RS && PK && want) ? (u32)(1 + rs_fwd) : 0u;
                // SIP (the packed kernels but the fused one, round 6): a record with ONE indel of up to seven bases between two
                // match runs and no clip — M a, I | D g, M b: its three operations are here — is given its single-indel entry by
                // this phase instead of the general pass (which took 260 wave-cycles per record it saw, a plain record's whole
                // cost): the conditions are the general pass's for such an entry (complete flanks, the window loads inside the
                // SEQ column with the sixteen nibbles of slack of a gapped step, a CIGAR that agrees with SEQ), the entry is the
                // one it would have made, and anything it would have refused or reported is left to it.
                constexpr bool SIP = PK && !RS && MDX_PK_SIP;
                bool sone = false, s_del = false;
                u32 s_nq = 0u, s_n0 = 0u, s_vlr = 0u;
                // (a tile without such a record — any tile of ungapped data — skips all of it)
                if (SIP && __ballot(cand && cn == 3u && ((6u >> (g1 & 0xFu)) & 1u) != 0u)) {
                    const u32 op1 = g1 & 0xFu, s_a = g0 >> 4, s_g = g1 >> 4, s_b = g2 >> 4;
                    const bool m2 = ((0x181u >> (g2 & 0xFu)) & 1u) != 0;
                    s_del = op1 == 2u;
                    s_nq = s_a + s_b + (s_del ? 0u : s_g); s_n0 = s_a + s_b + (s_del ? s_g : 0u);
                    s_vlr = (s_a < (u32)L ? s_a : (u32)L) | ((s_b < (u32)L ? s_b : (u32)L) << 8);
                    sone = cand && cn == 3u && m0 && m2 && (op1 == 1u || op1 == 2u) && s_g - 1u < 7u && s_a - 1u < 32767u && s_b - 1u < 32767u &&
                           s_nq == lseq && s_nq < 32768u && c_pos >= A && (u32)c_pos + s_n0 + (u32)A <= clen &&
                           c_so0 >= pad8 + 16u && s_nq + pad8 + 16u <= nb32 && c_so0 <= nb32 - s_nq - pad8 - 16u;
                }
                const u64 mDef = __ballot(kept && !triv && !sone);
                if (mDef) {
                    if (kept && !triv && !sone) {
                        // the record's index and the columns just read (36 bytes): the general pass does not gather them again
                        const int at = nDef + mbcnt64(mDef, 0);
                        uint4 c0_, c1_;
                        c0_.x = fl | ((u32)c_lib << 16); c0_.y = (u32)c_tid; c0_.z = (u32)c_pos; c0_.w = (u32)c_tlen;
                        c1_.x = c_co0; c1_.y = c_co1; c1_.z = c_so0; c1_.w = c_so1;
                        if (pfl) {
                            // (stores the compiler does not know of: it would wait for them — vmcnt(0) — at the head of the runs'
                            // pipelined loops; they are waited for in front of the general pass that reads them back)
                            pfl_store_b32(dlist, ((u32)at & DM) * 4u, ri | (rs_def << 30));
                            pfl_store_b128(dcols, ((u32)at & DM) * 32u, u32x4{c0_.x, c0_.y, c0_.z, c0_.w});
                            pfl_store_b128(dcols, ((u32)at & DM) * 32u + 16u, u32x4{c1_.x, c1_.y, c1_.z, c1_.w});
                        } else {
                        dlist[(u32)at & DM] = ri | (rs_def << 30);
                        dcols[2 * ((u32)at & DM)] = c0_;
                        dcols[2 * ((u32)at & DM) + 1] = c1_;
                        }
                    }
                    nDef += __popcll(mDef);
                }
                const int rev = (fl >> 4) & 1, libid = ML ? 0 : c_lib - a.lib_lo, nq = (int)len;
                const int lg_lib = ML ? ml_lib : libid;       // (the library for the dense length histogram and the overflow list)
                const int lbase = __mul24(libid, d.w_lib);
                const bool isF = triv && nq >= L;
                // RS: record routing (rescale.py:300-342).  Rescaled here, while it is counted: a record of this loop with
                // at most 2 L aligned bases (every column is then a task of one of its windows) whose end windows can be
                // fetched eight bytes at a time; any other record that wants rescaling goes to the list of the kernels
                // behind this one.
                bool rs_fused = false;
                if (RS) {
                    rs_fused = want && triv && nq <= 2 * L && (u64)c_so0 + lseq + 16u <= (u64)a.n_bases;
                    // status of the records this kernel is done with (rescale.py:300-342): the fused ones and those written
                    // back unchanged (mr_raw is preset to NaN by the launch: such a record keeps it); the others go to the
                    // wavefront's list
#ifndef MDX_RSABL_NOST
                    if (valid && (rs_fused || !want)) p.rs.status[ri] = (u8)rs_st;
#endif
                    // (PK: the general pass lists its own records)
                    const bool lst = want && !rs_fused && !(PK && kept && !triv);
#ifdef MDX_RSABL_NOLIST
                    const u64 mW = 0;
#else
                    const u64 mW = __ballot(lst);
#endif
                    if (mW) {
                        if (lst) (p.rs.gen_list + (size_t)gwave * (size_t)p.list_cap)[n_rs + (u32)mbcnt64(mW, 0)] = ri;
                        n_rs += (u32)__popcll(mW);
                    }
                }
                // statistics.py:37-51: positions [0, min(len, L)) of a soft clip, the left side's table iff no alignment
                // column precedes it — as a difference (+1 at 0, -1 at the end), like the general pass
                if (__ballot(triv && (qs | tr) != 0u)) {
                    if (triv && qs) {
                        const int m = qs < (u32)L ? (int)qs : L, base = lbase + d.off_mis() + __mul24(__mul24(rev * 2, L), 25) + COL_S;
                        atomicAdd(&lds[base], 1u);
                        if (m < L) atomicAdd(&lds[base + __mul24(m, 25)], 0xFFFFFFFFu);
                    }
                    if (triv && tr) {
                        const int m = tr < (u32)L ? (int)tr : L, base = lbase + d.off_mis() + __mul24(__mul24(rev * 2 + 1, L), 25) + COL_S;
                        atomicAdd(&lds[base], 1u);
                        if (m < L) atomicAdd(&lds[base + __mul24(m, 25)], 0xFFFFFFFFu);
                    }
                }
                // statistics.py:117-126 — in 32 bits (|tlen| of INT_MIN is 2^31 as unsigned): a paired record counts as
                // read 1 of a proper pair, by |tlen|; an unpaired one by its reference length
                int lkey = -1;
                {
                    const u32 paired = fl & 1u;
                    const bool counts = (triv || sone) && (!paired || (fl & 0x42u) == 0x42u);
                    const u32 flen = paired ? (c_tlen < 0 ? 0u - (u32)c_tlen : (u32)c_tlen) : (sone ? s_n0 : len);
                    const int krow = (paired ? 0 : 2) + rev;                 // kind * 2 + strand
                    if (counts && flen < (u32)d.lgd_lds) lkey = lbase + d.off_lgd() + __mul24(krow, d.lgd_lds) + (int)flen;
                    if (__ballot(counts && flen >= (u32)d.lgd_lds)) {          // (a tile in five at the survey's insert sizes)
                        if (counts && flen >= (u32)d.lgd_lds) {
                            if (flen < (u32)d.lgd_max) {
                                atomicAdd(&p.lgd_dense[(i64)(blockIdx.x & (MDX_LGD_COPIES - 1)) * ((i64)a.nlib_total * 4 * d.lgd_max) +
                                                       ((i64)lg_lib * 4 + krow) * d.lgd_max + flen], 1ull);
                            } else {
                                const u64 slot = atomicAdd(p.n_lgd_over, 1ull);
                                if ((i64)slot < p.lgd_over_cap) {
                                    p.lgd_over[4 * slot + 0] = lg_lib + a.lib_lo;
                                    p.lgd_over[4 * slot + 1] = paired ? 0 : 1;
                                    p.lgd_over[4 * slot + 2] = rev;
                                    p.lgd_over[4 * slot + 3] = (i64)flen;
                                }
                            }
                        }
                    }
                }
                // fragment lengths: wave-level aggregation while it pays (uniform read lengths give one or two distinct keys
                // per tile; insert sizes give fifty, and a round that gathers fewer than four lanes is the last), the
                // remainder as individual adds
                {
                    u64 pend = __ballot(lkey >= 0);
#pragma unroll 1
                    for (int round = 0; round < 2 && pend; round++) {
                        const int leader = __ffsll((long long)pend) - 1;
                        const int key = rl(lkey, leader);
                        const u64 same = __ballot(lkey == key);
                        const int n_same = __popcll(same);
                        if (n_same < 4) break;
                        if (lane == leader) bump_n<USE_LDS>(lds, raw, key, (u32)n_same);
                        if (lkey == key) lkey = -1;
                        pend &= ~same;
                    }
                    if (lkey >= 0) bump<USE_LDS>(lds, raw, lkey);
                }
                const u64 mT = __ballot(triv);
                n_kept_lite += (u32)__popcll(mT);        // (added to the table once, behind the loop)
                if (SIP) {
                    const u64 mS1 = __ballot(sone);
                    if (mS1) {
                        n_kept_lite += (u32)__popcll(mS1);
                        // (the general pass's entry of such a record: see its gpre / isS)
                        uint4 es;
                        es.x = c0 + (u32)c_pos - (u32)A + 256u;
                        es.y = c_so0 - es.x + pk_dso;
                        es.z = s_nq | 0x8000u | (s_vlr << 16);
                        const u32 dnq = s_n0 - s_nq;
                        es.w = ((u32)(lbase + d.off_tc() + rev * 4 * 512) << 2) | ((u32)libid << 24) | (MASK ? 0x40000000u : 0u) | ((u32)rev << 31) |
                               (dnq & 0xFFu) | (((dnq >> 8) & 7u) << 21) | PK_ONE;
                        const u64 mD1 = __ballot(sone && s_del), mI1 = mS1 & ~mD1;
                        if (sone && !s_del) pfl_store_b128(ringI, ((lI + (u32)mbcnt64(mI1, 0)) & RM) * 16u, u32x4{es.x, es.y, es.z, es.w});
                        if (sone && s_del) pfl_store_b128(ringD, ((lD + (u32)mbcnt64(mD1, 0)) & RM) * 16u, u32x4{es.x, es.y, es.z, es.w});
                        lI += (u32)__popcll(mI1); lD += (u32)__popcll(mD1);
                    }
                }
                // staging entry (as in the general pass)
                uint4 ent;
                ent.x = c0 + (u32)c_pos - (u32)A + 256u;
                ent.y = PK ? sq - ent.x + pk_dso : sq;
                ent.z = (u32)nq | ((u32)((A + (nq < L ? nq : L)) * 0x101) << 16);
                ent.w = ((u32)(lbase + d.off_tc() + rev * 4 * 512) << 2) | ((u32)libid << 24) |
                        ((MASK && !(fl & 0x8000u) && q0 != 0xFFu) ? 0x40000000u : 0u) | ((u32)rev << 31);
                // (RS: a fused record counts into the second TC table — which is what marks its entry and events; bit 20 =
                // rescaled from the 5' end only)
                // (PK: no second table — bit 18 marks the entry)
                if (RS && rs_fused) {
                    if (PK) ent.w |= (1u << 18) | ((u32)rs_fwd << 20);
                    else
                    ent.w = ((u32)(p.rs.tcb_off + __mul24(libid, d.w_tc) + rev * 4 * 512) << 2) | ((u32)libid << 24) | ((u32)rev << 31) |
                            ((u32)rs_fwd << 20);
                }
                const u64 mF = __ballot(isF), mP = mT & ~mF;
                nF = __popcll(mF);
                // MASK: the complete records that cannot be masked (no qualities, or the caller's hint) are staged first and
                // counted by a run of their own that does not look at qualities — the unmasked step
                const u64 mF0 = MASK ? __ballot(isF && !(ent.w & 0x40000000u)) : 0ull;
                nF0 = MASK ? __popcll(mF0) : 0;
                // PK: the complete records are staged by strand, the forward ones first: all steps of the run but the one that
                // holds the border count into one set of planes
                const u64 mFm = PK ? __ballot(isF && rev) : 0ull;
                nFp = PK ? nF - __popcll(mFm) : 0;
                // MdxTabArgs::ref2 — a record's windows are read from the copy of the reference in which they lie in ONE 128-byte
                // line, where there is such a one: [lo, hi] = the bytes its lanes load (a dword-aligned triple each: the left
                // side's lanes from the window's first nibble on, the right side's up to the right flank's last).  The second
                // copy lies 2 GiB + 64 bytes behind the first: the 64 bytes — 128 nibbles — go into the window's offset (and out
                // of the SEQ column's, which is relative to it), the 2 GiB into bit 31 of the entry's word w, which the fill
                // of a complete step, and of a tile's own partial step, or's to the lane's byte offset (for both the bit is
                // free: the strand of such an entry is its place in the staging area).
                // (a tile of a coordinate-sorted batch — its first and last record within 64 KB of one another — reads the
                // first copy only: its records share their lines with their neighbours, and two copies are twice the lines;
                // 25 M sorted records over 3 Gb: 1.09 ms with one copy, 1.16 with both)
                constexpr bool REF2P = PK && !RS && !MASK && MDX_PK_FASTIDX && MDX_PK_FASTP;     // (... the partial steps that look at w: FIDP)
                bool copy_b = false;
                if (PK && !RS && a.ref2 && mT) {
                    const u32 xa = (u32)rl((int)ent.x, __ffsll((long long)mT) - 1), xb = (u32)rl((int)ent.x, 63 - __clzll((long long)mT));
                    if ((xa > xb ? xa - xb : xb - xa) >= (1u << 17)) {
                        const u32 Wn = (u32)(16 * d.nl16), span = (u32)nq + (u32)(2 * A);
                        const u32 first = span < Wn ? ent.x + span - Wn : ent.x, last = (span < Wn ? ent.x + Wn : ent.x + span) - 16u;
                        const u32 lo = (first >> 3) << 2, hi = ((last >> 3) << 2) + 11u;
                        const bool two_a = (lo >> 7) != (hi >> 7), two_b = ((lo + 64u) >> 7) != ((hi + 64u) >> 7);
                        copy_b = triv && (isF || REF2P) && two_a && !two_b;
                        if (copy_b) { ent.x += 128u; ent.y -= 128u; }
                    }
                }
                if (mF) {
                    if (PK) {
                        uint4 entC = ent;
                        if (!RS) entC.w = copy_b ? 0x80000000u : 0u;
                        if (isF) stg[rev ? nFp + mbcnt64(mFm, 0) : mbcnt64(mF & ~mFm, 0)] = entC;
                    }
                    else if (isF) stg[MASK ? ((ent.w & 0x40000000u) ? nF0 + mbcnt64(mF & ~mF0, 0) : mbcnt64(mF0, 0)) : mbcnt64(mF, 0)] = ent;
                    // the slots past the last record of a step shadow a real record (and are masked out); a run of the
                    // clean records in front (MASK) reads its own last slots from the maskable records' entries
                    // (PK: a slot past its strand's last entry reads the run's first entry, no padding)
                    if (!PK && (MASK || nF % R)) {
                        const int first = __ffsll((long long)mF) - 1;
                        uint4 pad;
                        pad.x = (u32)rl((int)ent.x, first); pad.y = (u32)rl((int)ent.y, first);
                        pad.z = (u32)rl((int)ent.z, first); pad.w = (u32)rl((int)ent.w, first);
                        if (lane < d.R - 1) stg[nF + lane] = pad;
                    }
                }
                // (PTILE — the packed fused kernel: the partial records of the tile are staged behind its complete ones, by strand,
                // and counted by a run of their own in the tile loop: their MR words are the tile's)
                constexpr bool PTILE = PK && (RS ? MDX_PKF_PTILE : MDX_PK_PTILE);
                int nPt = 0, nPtp = 0;
                const u64 mPm = PTILE ? __ballot(triv && !isF && rev) : 0ull;
                if (PTILE && mP) {
                    nPt = __popcll(mP); nPtp = nPt - __popcll(mPm);
                    uint4 entP = ent;
                    if (REF2P) entP.w = (ent.w & 0x7FFFFFFFu) | (copy_b ? 0x80000000u : 0u);
                    if (triv && !isF) stg[nF + (rev ? nPtp + mbcnt64(mPm, 0) : mbcnt64(mP & ~mPm, 0))] = entP;
                }
                if (!PTILE && mP) {
                    // short records and contig edges: tasks [-flank, min(nq, L)) per side (the list of partial records; DMP)
                    if (triv && !isF) {
                        const int dl = lbase + d.off_dmp() + rev * 2 * (A + L), dr = dl + (A + L);
                        const int k1 = nq < L ? nq : L;
                        if (!PK && k1 < L) { atomicAdd(&lds[dl + A + k1], 1u); atomicAdd(&lds[dr + A + k1], 1u); }
                        ringP[(lP + (u32)mbcnt64(mP, 0)) & RM] = ent;
                        if (RS) lri[(lP + (u32)mbcnt64(mP, 0)) & RM] = ri;      // (the record of the entry: where its MR goes)
                    }
                    lP += (u32)__popcll(mP);
                }
                if (RS) {
                    // (the MR words of the tile's staging entries)
                    mrm[lane] = 0ull;
                    if (lane < MDX_FUSE_MRM - 64) mrm[64 + lane] = 0ull;
                }
#ifndef MDX_ONLY_PHASE1   // probe build (tools/p1_probe.sh): phase 1 and the gapped walk only
                // ---------------------------------------------------- phase 2a: the complete records of the tile
                // One lane = eight consecutive bytes of a record's window; one wavefront step = R records.
                // Bytes that are not plain matches are not handled here: they are appended as events to a
                // wave-private LDS queue and counted later 64 at a time (drain_all), so the divergent
                // classification code runs once per 64 events instead of once per record.
                // (PK: the next tile's columns have had this tile's phase 1 to arrive: its second round trip goes out under the run)
                // (the stores of this phase 1 retired — they have had the phase —, so that the run's loops see nothing but loads
                // in flight and their waits are counted; PK: the next tile's second round trip goes out behind that wait and
                // lands under the run)
                // (pfl: no wait — this phase 1 has stored nothing the compiler knows of (pfl_store_*), and the next tile's second
                // round trip goes out from inside the tile's first run; a tile without a run: here)
                if (pfl) pfl_due = nxt;
                else
                __builtin_amdgcn_s_waitcnt(0x0F70);
                if (PK) {
                    MDX_PH(2);
                    if (nF) run(0, nF, std::integral_constant<int, STEP_C>{}, std::true_type{}, nFp);
                    MDX_PH(3);
                    if (PTILE && nPt) run(nF, nPt, std::integral_constant<int, STEP_P | (RS ? 0 : 8)>{}, std::true_type{}, nPtp);
                    if (pfl && pfl_due != 0xFFFFFFFEu) {
                        MDX_PH(15);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (pfl_due != 0xFFFFFFFFu) pfl_rt2(pfl_due);
                        pfl_due = 0xFFFFFFFEu;
                    }
                    MDX_PH(0);
                }
                else {
                if (nF0) run(0, nF0, std::integral_constant<int, STEP_C>{}, std::false_type{});
                if (nF - nF0) run(nF0, nF - nF0, std::integral_constant<int, STEP_C>{}, std::true_type{});
                }
                if (RS && PK) {
                    // The run has drained its events and counted the reference bases of the fused records' columns.  What is
                    // left: the qualities of the listed transitions, and the MR sums from the records' words.
                    rsq_flush();
                    // (a partial record that goes through the wavefront's list: behind the pass of the list that counts it)
                    if (rs_fused && (isF || PTILE)) p.rs.mr_raw[ri] = mr_of(mrm[isF ? (rev ? nFp + mbcnt64(mFm, 0) : mbcnt64(mF & ~mFm, 0))
                                                                  : nF + (rev ? nPtp + mbcnt64(mPm, 0) : mbcnt64(mP & ~mPm, 0))]);
                }
                if (RS && !PK) {
                    // The run has drained its events.  One round trip for what is left of the tile's fused records: the
                    // qualities of their listed transitions (rsq_flush) and the reference bytes of the columns the left
                    // windows do not hold, [L, nq) — subs[nt_ref], rescale.py:142-143 (the first 32 of them requested
                    // here; a record shorter than L leaves L - nq zeroed bytes in plane A of its left columns).
                    const u8 *__restrict__ g_ref = p.ref + (size_t)(c0 + (u32)c_pos);
#ifdef MDX_RSABL_NOE
                    const bool resting = false;
#else
                    const bool resting = rs_fused;
#endif
                    u32x4 rv2[2];
                    rv2[0] = rv2[1] = u32x4{0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
                    if (resting && nq > L) rv2[0] = *(const u32x4_u *)(g_ref + L);
                    if (resting && nq > L + 16) rv2[1] = *(const u32x4_u *)(g_ref + L + 16);
                    rsq_flush();
                    if (resting) {
                        int nA = 0, nC = 0, nG = 0, nT = 0;
                        auto count16 = [&](const u32x4 rv, const int qi) {
                            const u64 r64[2] = {(u64)rv.x | ((u64)rv.y << 32), (u64)rv.z | ((u64)rv.w << 32)};
#pragma unroll
                            for (int h = 0; h < 2; h++) {
                                const u64 ok7 = ~r64[h] & 0x8080808080808080ull & byte_range(0, nq - qi - 8 * h);   // bit 7 clear: a base
                                const u64 b1 = (r64[h] << 6) & ok7, b2 = (r64[h] << 5) & ok7;                      // bit 1, bit 2 of the byte
                                nA += __popcll(ok7 & ~b1 & ~b2); nC += __popcll(b1 & ~b2); nT += __popcll(~b1 & b2); nG += __popcll(b1 & b2);
                            }
                        };
                        count16(rv2[0], L);
                        count16(rv2[1], L + 16);
                        for (int qi = L + 32; qi < nq; qi += 16) count16(*(const u32x4_u *)(g_ref + qi), qi);
                        if (nq < L) nA -= L - nq;
                        if (rev) { bcA += nT; bcC += nG; bcG += nC; bcT += nA; }
                        else { bcA += nA; bcC += nC; bcG += nG; bcT += nT; }
                        // the MR sum of a complete record from its word (a record that is not complete: behind the run of
                        // the partial list)
                        if (isF) p.rs.mr_raw[ri] = mr_of(mrm[mbcnt64(mF, 0)]);
                    }
                }
#endif
            }
            // ---------------------------------------------------- the general pass over the records left to it
            const int pend = nDef - dDone;
            if (pend >= 64 || (over && pend > 0)) {
                const int m = pend < 64 ? pend : 64;
                MDX_PH(6);
                // (PK: the general pass is where the kernel wants the most registers: the bit-sliced counters are folded
                // into TC in front of it — every dozen tiles, about as often as their eight planes ask for anyway — and
                // are not live across it)
#ifdef MDX_PK_FLUSH_GENERAL
                if (PK) bs_flush();
#endif
                // (the wavefront's own stores: complete before they are read back)
                if (pfl) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const u32 at = (u32)(dDone + (lane < m ? lane : 0)) & DM;
                const u32 rk = ring_u32(dlist + at);       // (record indices are below 2^30; [31:30]: the packed fused kernel's rs_code)
                const uint4 c0_ = ring_at(dcols + 2 * at), c1_ = ring_at(dcols + 2 * at + 1);
                general(rk & 0x3FFFFFFFu, lane < m, c0_.x & 0xFFFFu, (int)(c0_.x >> 16), (int)c0_.y, (int)c0_.z, (int)c0_.w, c1_.x, c1_.y, c1_.z, c1_.w,
                        (RS && PK) ? rk >> 30 : 0u);
                dDone += m;
                MDX_PH(0);
            }
            if (over ? dDone >= nDef : past) break;
            if (!past) {
                if (RS || pfl) { cur = nxt; nxt = nxt != 0xFFFFFFFFu ? tile_of(nxt2_raw) : 0xFFFFFFFFu; }
                else cur = tile_of(nxt2_raw);
            }
        }
        // ---------------------------------------------------- the lists, at the end of a round: whole passes (63 entries:
        // whole steps) while a list holds one — what is left of a list waits for the next round's entries —, everything when
        // no tile is to come any more.  The rings stay small whatever the batch, and their entries come back from the L2.
#ifndef MDX_ONLY_PHASE1
        {
            const u32 need = (!ROUNDS || cur == 0xFFFFFFFFu) ? 1u : (u32)TL;
            if (lC - hC >= need || lP - hP >= need || lI - hI >= need || lD - hD >= need) {
                // the entries this wavefront appended (its own stores: complete before they are read back)
                if (pfl) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (phase 1's single-indel entries: stores the compiler does not count)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                MDX_PH(7);
                while (lC - hC >= need) hC += list_pass(ringC, lri, hC, lC - hC, std::integral_constant<int, STEP_C>{});
#ifndef MDX_ABL_NO_PRUN      // (ablation builds, tools/ablate.sh: wrong tables, instruction counts by part)
                while (lP - hP >= need) hP += list_pass(ringP, lri, hP, lP - hP, std::integral_constant<int, STEP_P>{});
#else
                hP = lP;
#endif
#ifndef MDX_ABL_NO_GRUN
                MDX_PH(8);
                while (lI - hI >= need) hI += list_pass(ringI, lriI, hI, lI - hI, std::integral_constant<int, STEP_GI>{});
                MDX_PH(9);
                while (lD - hD >= need) hD += list_pass(ringD, lriD, hD, lD - hD, std::integral_constant<int, STEP_GD>{});
#else
                hI = lI; hD = lD;
#endif
                MDX_PH(0);
            }
        }
#endif
        if (!ROUNDS || cur == 0xFFFFFFFFu) break;
        }   // rounds
        if (lane == 0 && n_kept_lite) bump_n<USE_LDS>(lds, raw, (int)(d.w_total - 1), n_kept_lite);
        if (RS && lane == 0) a.rs.gen_count[gwave] = n_rs;
    }

#ifdef MDX_WAVE_CLK
    if (a.dbg_clk && lane == 0) a.dbg_clk[3 * (size_t)gwave + 1] = wall_clock64();
#endif
    MDX_PH(10);
    if (FAST) {
        if (qcount > 0) drain_all();
        if (PK) bs_flush();
    }
#ifdef MDX_WAVE_CLK
    if (a.dbg_clk && lane == 0) a.dbg_clk[3 * (size_t)gwave + 2] = wall_clock64();
#endif
#ifdef MDX_PHASE_CLK
    MDX_PH(11);
    if (a.dbg_clk && lane == 0) {
#pragma unroll
        for (int q = 0; q < 16; q++) a.dbg_clk[3 * 20000 + 16 * (size_t)(gwave & 8191u) + q] += (unsigned long long)ph_acc[q];
    }
#endif
    if (USE_LDS) {
        __builtin_amdgcn_s_waitcnt(0xC07F);  // the hand-written ds_adds of this wavefront
        __syncthreads();
        if (RS) {
            // The reference bases of the fused records' columns (subs["A"] ... ["T"], read orientation): the plain matches
            // of the left columns from the second TC table (lane (slot, left side, m), byte j = window byte 8 m + j, a
            // column iff A <= 8 m + j < A + L), everything else from the lanes' own counts; words 0..3 of the block's
            // summary row.  Then the second table is added to the first.
            int tA = 0, tC = 0, tG = 0, tT = 0;
            // (PK: no second table; the lanes' own counts are by class — A, C, T, G — of the stored strand, and a lane's slot
            // fixes the strand: complemented here for the reverse slots)
            const int n_tcb = PK ? 0 : d.nlib * d.w_tc;
            if (PK) {
                const int cA = bcA, cC = bcC, cT = bcT, cG = bcG;
                bcA = p_strand ? cT : cA; bcC = p_strand ? cG : cC; bcG = p_strand ? cC : cG; bcT = p_strand ? cA : cT;
            }
            for (int i = threadIdx.x; i < n_tcb; i += BLOCK) {
                const int w = i & 511, k = (i >> 9) & 3, strand = (i >> 11) & 1;
                const int ln = w & 63, jb = w >> 6;
                const int g = ln / d.G, ll = ln - g * d.G, b = 8 * ll + jb;
                if (g < d.R && ll < d.nl8 && b >= A && b < A + L) {
                    const int v = (int)lds[a.rs.tcb_off + i];
                    int bb = k ^ (k >> 1);                  // A,C,T,G -> A,C,G,T
                    if (strand) bb = 3 - bb;
                    tA += bb == 0 ? v : 0; tC += bb == 1 ? v : 0; tG += bb == 2 ? v : 0; tT += bb == 3 ? v : 0;
                }
            }
            int v4[4] = {tA + bcA, tC + bcC, tG + bcG, tT + bcT};
            for (int b = 0; b < 4; b++) {
                int v = v4[b];
                for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
                if (lane == 0 && v) atomicAdd(&rs_cnt[b], (u32)v);
            }
            __syncthreads();
            for (int i = threadIdx.x; i < n_tcb; i += BLOCK) {
                const int lib = i / d.w_tc, j = i - lib * d.w_tc;
                lds[lib * d.w_lib + d.off_tc() + j] += lds[a.rs.tcb_off + i];
            }
            if (threadIdx.x < 4) a.rs.subs_part[(size_t)blockIdx.x * rs_ncnt + threadIdx.x] = rs_cnt[threadIdx.x];
            __syncthreads();
        }
        u32 *out = a.partials + (i64)blockIdx.x * d.w_total;
        if (PK) {
            // The TC words of the slot in the layout every consumer knows (MdxDims: [strand][base][64 byte + lane], the counts
            // in the words of slot 0): word (strand, k, lane (side, m), byte jb) = window byte b = 8 m + jb of the left side /
            // e = 8 m + 7 - jb of the right side, summed over the H4 slots of the strand in this kernel's table.
            for (int i = threadIdx.x; i < d.w_tc; i += BLOCK) {
                const int w = i & 511, k = (i >> 9) & 3, strand = (i >> 11) & 1;
                const int ln = w & 63, jb = w >> 6;
                u32 v = 0u;
                if (ln < d.G) {
                    const int side = ln >= d.nl8, m = ln - side * d.nl8;
                    const int wn = side ? 8 * m + 7 - jb : 8 * m + jb;          // window nibble (right side: from the outer end)
                    if (wn < A + L) {
                        const int j = side ? 15 - (wn & 15) : (wn & 15);
                        for (int gs = 0; gs < d.H4; gs++)
                            v += lds[d.off_tc() + (k << 10) + 64 * j + (strand * d.H4 + gs) * d.G4 + side * d.nl16 + (wn >> 4)];
                    }
                }
                out[d.off_tc() + i] = v;
            }
            for (i64 i = d.w_tc + threadIdx.x; i < d.w_total; i += BLOCK) out[i] = lds[i];
        } else
        for (i64 i = threadIdx.x; i < d.w_total; i += BLOCK) out[i] = lds[i];
    }
    }
}

template <bool MASK, bool FAST>
static hipError_t prep_one(size_t lds_bytes) {
    return hipFuncSetAttribute((const void *)tabulate_kernel<true, MASK, FAST>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

hipError_t mdx_k_fuse_prepare(size_t lds_bytes) {
    return hipFuncSetAttribute((const void *)tabulate_kernel<true, false, true, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

void mdx_k_tabulate_fused(const MdxTabArgs &a, int grid, size_t lds_bytes, hipStream_t s) {
    if (a.n_reads <= 0) return;
    hipLaunchKernelGGL((tabulate_kernel<true, false, true, true>), dim3(grid), dim3(MDX_FUSE_BLOCK), lds_bytes, s, a);
}

hipError_t mdx_k_pkf_prepare(size_t lds_bytes) {
    return hipFuncSetAttribute((const void *)tabulate_kernel<true, false, true, true, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

void mdx_k_tabulate_packed_fused(const MdxTabArgs &a, int grid, size_t lds_bytes, hipStream_t s) {
    if (a.n_reads <= 0) return;
    hipLaunchKernelGGL((tabulate_kernel<true, false, true, true, true>), dim3(grid), dim3(MDX_FUSE_BLOCK), lds_bytes, s, a);
}

// Behind the packed fused kernel: the records it has listed for the rescale kernels, which read ASCII — their stretches
// of the 4-bit SEQ column (soft clips included) written out as ASCII at the same offsets of a scratch column, eight bases
// per lane and step (whole aligned groups of eight: what lies outside a record belongs to its neighbours and is the same
// bytes whoever writes them), sixteen lanes per record, four wavefronts per list.
typedef u32 __attribute__((aligned(1))) u32_u1;
#define UNPK_SPLIT 4
__global__ __launch_bounds__(256) void unpack_listed_kernel(const u32 *__restrict__ in_count, const u32 *__restrict__ in_list, i64 in_cap, int n_in,
                                                            const u32 *__restrict__ seq_off, const u8 *__restrict__ seq4,
                                                            u8 *__restrict__ out, i64 n_bases) {
    const int lane = threadIdx.x & 63;
    const i64 gw = (i64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const i64 l = gw / UNPK_SPLIT;
    const int part = (int)(gw - l * UNPK_SPLIT);
    if (l >= n_in) return;
    const u32 n = in_count[l];
    const u32 *const list = in_list + l * in_cap;
    // 64 records per pass: a lane fetches the bounds of one (two round trips for the pass), then sixteen lanes take a record,
    // four records at a time — the loads of a pass do not depend on one another
    for (u32 base = 64u * (u32)part; base < n; base += 64u * UNPK_SPLIT) {
        u32 o0 = 0, o1 = 0;
        if (base + (u32)lane < n) {
            const u32 ri = list[base + (u32)lane];
            o0 = seq_off[ri] & ~7u; o1 = seq_off[ri + 1];
        }
        const u32 m = n - base < 64u ? n - base : 64u;
        for (u32 r0 = 0; r0 < m; r0 += 4u) {
            const int r = (int)r0 + (lane >> 4);
            const u32 a0 = (u32)__shfl((int)o0, r), a1 = (u32)__shfl((int)o1, r);
            for (u32 o = a0 + 8u * (u32)(lane & 15); o < a1; o += 128u) {
                u32 v;
                if ((i64)o + 8 <= n_bases) v = *(const u32_u1 *)(seq4 + (o >> 1));
                else {
                    v = 0u;
                    for (u32 k = 0; k < 4u && (i64)o + 2 * k < n_bases; k++) v |= (u32)seq4[(o >> 1) + k] << (8 * k);
                }
                u32 w[2] = {0u, 0u};
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const u32 nib = (v >> (4 * k)) & 15u;
                    const u32 ch = nib == 1u ? 'A' : (nib == 2u ? 'C' : (nib == 4u ? 'T' : (nib == 8u ? 'G' : 'N')));
                    w[k >> 2] |= ch << (8 * (k & 3));
                }
                u32x2 ww; ww.x = w[0]; ww.y = w[1];
                *(u32x2 *)(out + o) = ww;
            }
        }
    }
}
void mdx_k_unpack_listed(const uint32_t *in_count, const uint32_t *in_list, int64_t in_cap, int n_in, const uint32_t *seq_off,
                         const uint8_t *seq4, uint8_t *out, int64_t n_bases, hipStream_t s) {
    if (n_in <= 0) return;
    const i64 waves = (i64)n_in * UNPK_SPLIT;
    hipLaunchKernelGGL(unpack_listed_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, in_count, in_list, (i64)in_cap, n_in,
                       seq_off, seq4, out, (i64)n_bases);
}

hipError_t mdx_k_prepare_packed_masked(size_t lds_bytes) {
    hipError_t e = hipFuncSetAttribute((const void *)tabulate_kernel<true, true, true, false, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)tabulate_kernel<true, true, true, false, true, true>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    return e;
}
// (a.n_libs > 0: the libraries [lib_lo, lib_lo + n_libs) in one launch over the bucketed columns, a library per pool: see ML)
void mdx_k_tabulate_packed_masked(const MdxTabArgs &a, int grid, int threads, size_t lds_bytes, hipStream_t s) {
    if (a.n_reads <= 0) return;
    if (a.n_libs > 0) hipLaunchKernelGGL((tabulate_kernel<true, true, true, false, true, true>), dim3(grid), dim3(threads), lds_bytes, s, a);
    else
    hipLaunchKernelGGL((tabulate_kernel<true, true, true, false, true>), dim3(grid), dim3(threads), lds_bytes, s, a);
}
// --min-basequal for the packed kernels: the mask folded into the 4-bit SEQ column (MDX_SEQ_4BIT -> MDX_SEQ_4BITQ,
// include/mdx.h) — a symbol whose quality is below the threshold (align.py:65-71; 0xFF, no qualities, is not) becomes the
// complement of its code.  Eight bases per thread; the mask from the quality column or,
// if the caller brings one, from its bitmap (mdx_batch::lowq: bit i = quality i is below the threshold).  In place or
// into a copy.
__global__ void fold_mask_kernel(const u8 *__restrict__ seq_in, u8 *__restrict__ seq_out, const u8 *__restrict__ qual,
                                 const u8 *__restrict__ lowq, i64 n_bases, u32 minq) {
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 b0 = 8 * t;
    if (b0 >= n_bases) return;
    const i64 n_bytes = (n_bases + 1) / 2;
    u32 bits = 0u;
    if (lowq) bits = lowq[t];
    else if (b0 + 8 <= n_bases) {
        const u32x2 q = *(const u32x2_u *)(qual + b0);
        const u32 m4 = minq * 0x01010101u;
        const u32 l0 = ~((q.x | 0x80808080u) - m4) & ~q.x & 0x80808080u, l1 = ~((q.y | 0x80808080u) - m4) & ~q.y & 0x80808080u;
        bits = ((((l0 >> 7) * 0x00204081u) >> 21) & 0xFu) | (((((l1 >> 7) * 0x00204081u) >> 21) & 0xFu) << 4);
    } else {
        for (int k = 0; k < 8 && b0 + k < n_bases; k++) bits |= (u32)(qual[b0 + k] < minq) << k;
    }
    u32 v = 0u;
    const int nb = (int)(n_bytes - 4 * t < 4 ? n_bytes - 4 * t : 4);
    if (nb == 4) v = *(const u32_u *)(seq_in + 4 * t);
    else for (int k = 0; k < nb; k++) v |= (u32)seq_in[4 * t + k] << (8 * k);
    // the masked nibbles: all four bits flipped (beyond the column's last base: none — the bits end with the qualities)
    v ^= spread8(bits);
    if (nb == 4) *(u32_u *)(seq_out + 4 * t) = v;
    else for (int k = 0; k < nb; k++) seq_out[4 * t + k] = (u8)(v >> (8 * k));
}
void mdx_k_fold_mask(const uint8_t *seq4_in, uint8_t *seq4_out, const uint8_t *qual, const uint8_t *lowq, int64_t n_bases, int minqual,
                     hipStream_t s) {
    if (n_bases <= 0) return;
    const i64 n = (n_bases + 7) / 8;
    hipLaunchKernelGGL(fold_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, seq4_in, seq4_out, qual, lowq, (i64)n_bases, (u32)minqual);
}

hipError_t mdx_k_prepare_packed(size_t lds_bytes) {
    hipError_t e = hipFuncSetAttribute((const void *)tabulate_kernel<true, false, true, false, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)tabulate_kernel<true, false, true, false, true, true>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    return e;
}

// the packed kernel: 4-bit SEQ column and 4-bit reference; one library per launch, or (a.n_libs > 0) the libraries
// [lib_lo, lib_lo + n_libs) side by side over the bucketed columns, a library per pool
void mdx_k_tabulate_packed(const MdxTabArgs &a, int grid, int threads, size_t lds_bytes, hipStream_t s) {
    if (a.n_reads <= 0) return;
    if (a.n_libs > 0) hipLaunchKernelGGL((tabulate_kernel<true, false, true, false, true, true>), dim3(grid), dim3(threads), lds_bytes, s, a);
    else
    hipLaunchKernelGGL((tabulate_kernel<true, false, true, false, true>), dim3(grid), dim3(threads), lds_bytes, s, a);
}

hipError_t mdx_k_prepare(size_t lds_bytes) {
    hipError_t e = prep_one<false, false>(lds_bytes);
    if (e == hipSuccess) e = prep_one<true, false>(lds_bytes);
    if (e == hipSuccess) e = prep_one<false, true>(lds_bytes);
    if (e == hipSuccess) e = prep_one<true, true>(lds_bytes);
    return e;
}

template <bool USE_LDS, bool MASK, bool FAST>
static void launch_one(const MdxTabArgs &a, int grid, size_t lds_bytes, hipStream_t s) {
    hipLaunchKernelGGL((tabulate_kernel<USE_LDS, MASK, FAST>), dim3(grid), dim3(MDX_BLOCK),
                       USE_LDS ? lds_bytes : 0, s, a);
}

void mdx_k_tabulate(const MdxTabArgs &a, int mode, bool mask, int grid, size_t lds_bytes, hipStream_t s) {
    if (a.n_reads <= 0) return;
    if (mode != MDX_MODE_LDS) {
        // tables do not fit the LDS: global u64 atomics, generic code only
        if (mask) launch_one<false, true, false>(a, grid, 0, s);
        else launch_one<false, false, false>(a, grid, 0, s);
        return;
    }
    if (a.dims.fast_ok() && a.ref32) {
        if (mask) launch_one<true, true, true>(a, grid, lds_bytes, s);
        else launch_one<true, false, true>(a, grid, lds_bytes, s);
    } else {
        if (mask) launch_one<true, true, false>(a, grid, lds_bytes, s);
        else launch_one<true, false, false>(a, grid, lds_bytes, s);
    }
}

// raw[w] += sum over block slots; blocks are split into `parts` groups to expose parallelism
// (the last word, the number of kept reads, goes to *raw_tail: a launch may hold a group of the libraries only)
// (tile_ctr: the pools' tile counters of the launch just reduced — 4096 words, fewer than any table — zeroed here for the
// next launch: one memset node less in front of every launch)
// (blockIdx.z: the library of a launch over several — a block's slot holds the image of its pool's library, `plan`; every
// library is summed into its own stretch of raw, lib_stride words apart)
// Which library a pool counts in a launch over several (tabulate_kernel<.., ML>): the pools are dealt to the libraries in
// proportion to their tiles — every library that has a record gets one at least, and then a pool goes, one at a time, to the
// library whose pools have the most tiles each — and a library's pools lie side by side.  plan[p] = {library, the pool's
// place among the library's pools, their number, the first of them}.  One thread: a few thousand steps at most.
__global__ void ml_plan_kernel(const u32 *__restrict__ lib_start, int lib_lo, int nlib, u32 T, int n_pools, uint4 *__restrict__ plan) {
    // (one wavefront, lane l = library l: its tiles and the pools it has so far in registers)
    const int l = (int)threadIdx.x;
    if (blockIdx.x || l >= 64) return;
    const u32 tiles = l < nlib ? (lib_start[lib_lo + l + 1] - lib_start[lib_lo + l] + T - 1u) / T : 0u;
    // (its share of the pools rounded down, one at least; what is left over — fewer than there are libraries — one at a time to
    // the library whose pools have the most tiles each; should the ones have made it too many, back from the library
    // whose pools have the fewest)
    auto wsum = [&](u32 v) -> u32 {
#pragma unroll
        for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
        return v;
    };
    const u32 total = wsum(tiles);
    u32 m = tiles ? (u32)(((u64)tiles * (u32)n_pools) / (total ? total : 1u)) : 0u;
    if (tiles && !m) m = 1u;
    if (total == 0u && l == 0) m = (u32)n_pools;        // (no record at all: the pools are library 0's, which has no tile for them)
    int given = (int)wsum(m);
    while (given != n_pools) {
        const bool more = given < n_pools;
        // (more: the highest load, tiles per pool; fewer: the lowest load among the libraries that can spare a pool)
        float load = more ? (m ? (float)tiles / (float)m : -1.0f) : (m > 1u ? -(float)tiles / (float)(m - 1u) : -3.0e38f);
        int who = l;
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            const float lo_ = __shfl_xor(load, o);
            const int wo = __shfl_xor(who, o);
            if (lo_ > load || (lo_ == load && wo < who)) { load = lo_; who = wo; }
        }
        if (l == who) m += more ? 1u : 0xFFFFFFFFu;
        given += more ? 1 : -1;
    }
    // the first pool of every library: the sum of the pools of the libraries in front
    u32 first = m;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 v = __shfl_up(first, o);
        if (l >= o) first += v;
    }
    first -= m;
    for (u32 k = 0; k < m && (int)(first + k) < n_pools; k++) plan[first + k] = make_uint4((u32)l, k, m, first);
}
void mdx_k_ml_plan(const uint32_t *lib_start, int lib_lo, int nlib, int T, int grid, void *plan, hipStream_t s) {
    hipLaunchKernelGGL(ml_plan_kernel, dim3(1), dim3(64), 0, s, lib_start, lib_lo, nlib, (u32)T, (int)mdx_n_pools((unsigned)grid), (uint4 *)plan);
}

__global__ void reduce_partials_kernel(const u32 *__restrict__ partials, u64 *__restrict__ raw, u64 *__restrict__ raw_tail,
                                       i64 w_total, int grid, int parts, u32 *__restrict__ tile_ctr, i64 lib_stride,
                                       const uint4 *__restrict__ plan, int n_pools) {
    const i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= w_total) return;
    const int part = blockIdx.y;
    const i64 z = blockIdx.z;
    if (tile_ctr && part == 0 && z == 0 && w < 4096) tile_ctr[w * MDX_CTR_PAD] = 0u;       // (the counters of up to 4 096 pools, a line apart)
    const int b0 = (int)((i64)grid * part / parts), b1 = (int)((i64)grid * (part + 1) / parts);
    // (plan: a launch over several libraries — block b has counted the library of its pool, plan[b mod n_pools].x; grid z adds
    // up the blocks of library z)
    u64 acc = 0;
    for (int b = b0; b < b1; b++)
        if (!plan || (i64)plan[b % n_pools].x == z) acc += (u64)(i64)(int)partials[(i64)b * w_total + w];   // signed (soft-clip differences)
    if (acc) atomicAdd(w == w_total - 1 ? raw_tail : &raw[z * lib_stride + w], acc);
}

void mdx_k_reduce_partials(const uint32_t *partials, unsigned long long *raw, unsigned long long *raw_tail,
                           int64_t w_total, int grid, hipStream_t s, uint32_t *tile_ctr, int n_lib, int64_t lib_stride, const void *plan) {
    const int threads = 256;
    const int blocks = (int)((w_total + threads - 1) / threads);
    int parts = grid < 32 ? grid : 32;
    if (n_lib > 8 && parts > 8) parts = 8;
    if (parts < 1) parts = 1;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(blocks, parts, n_lib), dim3(threads), 0, s, partials, raw, raw_tail,
                       (i64)w_total, grid, parts, w_total >= 4096 ? tile_ctr : nullptr, (i64)lib_stride, (const uint4 *)plan, (int)mdx_n_pools((unsigned)grid));
}

// raw (reference orientation) -> canonical tables (mapdamage_amd/layout.py):
//   out = [ mis nlib*2*2*L*25 | comp nlib*2*2*(L+A)*4 | lgd nlib*2*2*lgd_max | n_kept | n_lgd_over ]
__global__ void finalize_kernel(const u64 *__restrict__ raw, const u64 *__restrict__ lgd_dense,
                                const u64 *__restrict__ n_lgd_over, MdxDims d, u64 *__restrict__ out) {
    const int L = d.L, A = d.A;
    const i64 n_mis = (i64)d.nlib * 2 * 2 * L * 25;
    const i64 n_comp = (i64)d.nlib * 2 * 2 * (L + A) * 4;
    const i64 n_lgd = (i64)d.nlib * 2 * 2 * d.lgd_max;
    const i64 total = n_mis + n_comp + n_lgd + 2;
    // bytes counted into plane A of TC at window byte w of (strand, side) without being A tasks (DMP: differences)
    auto dumped = [&](i64 lb, int strand, int side, int w) -> u64 {
        u64 sum = 0;
        if (d.w_dmp)
            for (int q = 0; q <= w; q++) sum += raw[lb + d.off_dmp() + (strand * 2 + side) * (A + L) + q];
        return sum;
    };
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
        u64 v;
        if (i < n_mis) {
            i64 x = i;
            const int col = x % 25; x /= 25;
            const int p = x % L; x /= L;
            const int strand = x % 2; x /= 2;
            const int end = x % 2; x /= 2;   // 0 = 3p, 1 = 5p
            const i64 lb = x * d.w_lib;
            // '+': 5p = left(0), 3p = right(1);  '-': 5p = right, 3p = left
            const int side = strand ? end : 1 - end;
            if (col < 4) {
                const int b = strand ? 3 - col : col;        // complement on the reverse strand
                const int k = b ^ (b >> 1);                  // A,C,G,T -> device class A,C,T,G
                const i64 row = lb + d.off_mis() + ((strand * 2 + side) * L + p) * 25;
                // matches (gapped records / plain records) + every column whose reference symbol is k
                v = raw[row + k];
                for (int g = 0; g < d.R; g++)  // plain records: one copy per slot of the wavefront step
                    v += raw[lb + d.off_tc() + (strand * 4 + k) * d.t_pad + (side ? d.tau_right(p) : d.tau_left(p)) + g * d.G];
                for (int x = 0; x < 4; x++) v += raw[row + c_refcols[k * 4 + x]];
                if (k == 0) v -= dumped(lb, strand, side, p + A);
            } else {
                const int rc = strand ? c_comp_col[col] : col;
                if (col == COL_S) {   // soft clips are stored as differences over the positions
                    v = 0;
                    for (int q = 0; q <= p; q++) v += raw[lb + d.off_mis() + ((strand * 2 + side) * L + q) * 25 + COL_S];
                } else {
                    v = raw[lb + d.off_mis() + ((strand * 2 + side) * L + p) * 25 + rc];
                }
            }
        } else if (i < n_mis + n_comp) {
            i64 x = i - n_mis;
            const int b0 = x % 4; x /= 4;
            const int row = x % (L + A); x /= (L + A);
            const int strand = x % 2; x /= 2;
            const int end = x % 2; x /= 2;
            const i64 lb = x * d.w_lib;
            const int side = strand ? end : 1 - end;
            const int b = strand ? 3 - b0 : b0;
            const int k = b ^ (b >> 1);
            // 5p rows: -A..-1 (flank, distance A-row) then 1..L (read slot row-A)
            // 3p rows: -L..-1 (read slot L-1-row) then 1..A (flank, distance row-L+1)
            int slot = -1, dist = 0;
            if (end == 1) { if (row < A) dist = A - row; else slot = row - A; }
            else { if (row < L) slot = L - 1 - row; else dist = row - L + 1; }
            const i64 tc = lb + d.off_tc() + (strand * 4 + k) * d.t_pad;
            if (slot >= 0) {
                v = raw[lb + d.off_cmp() + ((strand * 2 + side) * L + slot) * 4 + k];
                for (int g = 0; g < d.R; g++) v += raw[tc + (side ? d.tau_right(slot) : d.tau_left(slot)) + g * d.G];
                if (k == 0) v -= dumped(lb, strand, side, slot + A);
            } else {
                v = 0;
                const int t = side ? d.tau_rflank(dist) : d.tau_lflank(dist);
                for (int g = 0; g < (d.R > 0 ? d.R : 1); g++) v += raw[tc + t + g * d.G];
                if (k == 0) v -= dumped(lb, strand, side, A - dist);
            }
        } else if (i < n_mis + n_comp + n_lgd) {
            i64 x = i - n_mis - n_comp;
            const int len = x % d.lgd_max; x /= d.lgd_max;
            const int strand = x % 2; x /= 2;
            const int kind = x % 2; x /= 2;
            v = 0;
            for (int c = 0; c < MDX_LGD_COPIES; c++) v += lgd_dense[(i64)c * n_lgd + (i - n_mis - n_comp)];
            if (len < d.lgd_lds) v += raw[x * d.w_lib + d.off_lgd() + (kind * 2 + strand) * d.lgd_lds + len];
        } else if (i == total - 2) {
            v = raw[d.w_total - 1];
        } else {
            v = *n_lgd_over;
        }
        out[i] = v;
    }
}

void mdx_k_finalize(const unsigned long long *raw, const unsigned long long *lgd_dense,
                    const unsigned long long *n_lgd_over, MdxDims d, unsigned long long *out,
                    hipStream_t s) {
    hipLaunchKernelGGL(finalize_kernel, dim3(512), dim3(256), 0, s, raw, lgd_dense, n_lgd_over, d, out);
}

// Genome base composition (mapdamage/composition.py:6-25 over seqtk.c:79-104): per-contig counts of
// A, C, G, T with upper and lower case folded — here read from the resident, already case-folded
// reference.  HBM-streaming reduction: 16 bytes per lane, per-lane counts, wavefront reduction,
// one atomic per (wavefront, contig, base).
__global__ void genome_comp_kernel(const u8 *__restrict__ ref, const i64 *__restrict__ contig_off, int n_contig,
                                   u64 *__restrict__ out) {
    const i64 total = contig_off[n_contig];
    const i64 nchunk = (total + 15) >> 4;
    for (i64 chunk = (i64)blockIdx.x * blockDim.x + threadIdx.x; chunk < nchunk; chunk += (i64)gridDim.x * blockDim.x) {
        const i64 b0 = chunk << 4;
        // contig of the chunk's first base (binary search on the offsets)
        int lo = 0, hi = n_contig - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (contig_off[mid] <= b0) lo = mid; else hi = mid - 1;
        }
        int cur = lo;
        u32 cnt[4] = {0, 0, 0, 0};
        for (int k = 0; k < 16; k++) {
            const i64 b = b0 + k;
            if (b >= total) break;
            while (b >= contig_off[cur + 1]) {
                for (int c = 0; c < 4; c++) if (cnt[c]) { atomicAdd(&out[(i64)cur * 4 + c], (u64)cnt[c]); cnt[c] = 0; }
                cur++;
            }
            const int r = (i8)ref[b];
            if (r >= 0) {  // A,C,T,G classes 0,1,2,3 -> output order A,C,G,T
                const int k2 = (r >> 1) & 3;
                cnt[k2 ^ (k2 >> 1)]++;
            }
        }
        for (int c = 0; c < 4; c++) if (cnt[c]) atomicAdd(&out[(i64)cur * 4 + c], (u64)cnt[c]);
    }
}

void mdx_k_genome_comp(const uint8_t *ref, const int64_t *contig_off, int n_contig, unsigned long long *out,
                       hipStream_t s) {
    hipLaunchKernelGGL(genome_comp_kernel, dim3(2048), dim3(256), 0, s, ref, (const i64 *)contig_off, n_contig, out);
}

// ------------------------------------------------------------------------------------------------
// Quality rescaling (mapdamage/rescale.py:195-365; BASELINE config[4]).  The new quality is a byte lookup
// LUT[sub][position key][old quality] prepared on the host with the reference's floating-point expressions
// (mapdamage_amd/rescale.py); MR is the fp64 sum of term[sub][key] over the rescaled columns in the read's own
// 5'->3' order (bit-exact).  Only columns within len5p of the 5' end or len3p of the 3' end have a key other than 0,
// and key 0 leaves the quality as it is and adds 0.0 (checked on the host: MdxRescaleArgs::lds_tables), so a record
// whose CIGAR is [S] M [S] is rescaled by ONE lane walking its two end windows (phase E); the whole read is streamed
// only for the substitution summary of rescale.py:108-192 (phase S, eight bytes per lane).  Any other record is
// walked column by column by a whole wavefront (`generic`).
// 512-thread blocks, three per CU (their LDS tables: ~35 KB each), six wavefronts per SIMD (80 VGPRs): measured
// against 256 x 5 (93 VGPRs, LDS-limited) -8 %; eight per SIMD spill (43 VGPRs) and lose 30 %
#ifndef RS_BLOCK
#define RS_BLOCK 512
#endif
#ifndef RS_WPS
#define RS_WPS 6
#endif
#ifndef RS_BPC
#define RS_BPC 3
#endif
#ifndef RS_EG
#define RS_EG 4           // 8-byte groups of the end windows fetched per round trip of phase E (2: one window; 4: both)
#endif
#ifndef RS_WG
#define RS_WG 2          // 8-column groups the walk kernel fetches per round trip
#endif
#define RS_STG 192        // staging entries per wavefront: at most three per record of a tile
__global__ __launch_bounds__(RS_BLOCK, RS_WPS) void rescale_kernel(MdxRescaleArgs a) {
    const int lane = threadIdx.x & 63;
    const i64 gwave = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const i64 nwaves = ((i64)gridDim.x * blockDim.x) >> 6;
    const int npos = 1 + a.len5p + a.len3p;
    u32 bc[4] = {0, 0, 0, 0};   // summary (rescale.py:108-143): raw reference-base counts per lane, see phase S
    // In the LDS (the kernel is launched only when they fit, a.lds_tables): the lookup tables and the summary histograms
    // (global atomics on a few hot words serialise in the L2): [lut 2 npos 94 B, padded][term 2 npos f64]
    // [counters u32: 4 x 2 x 94 transitions | 2 x npos x 94 rescaled-column kinds, padded to 16 B], flushed at block
    // end, [staging: RS_STG entries of 16 B per wavefront].
    extern __shared__ __attribute__((aligned(16))) u8 rs_lds[];
    const int lut_bytes = (2 * npos * 94 + 15) & ~15, n_cnt = 752 + 2 * npos * 94;
    const u8 *const l_lut = rs_lds;
    const double *const l_term = (const double *)(rs_lds + lut_bytes);
    u32 *const l_cnt = (u32 *)(rs_lds + lut_bytes + 2 * npos * 8);
    uint4 *const stg = (uint4 *)(rs_lds + lut_bytes + 2 * npos * 8 + ((n_cnt * 4 + 15) & ~15)) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * RS_STG;
    {
        for (int i = threadIdx.x; i < 2 * npos * 94; i += blockDim.x) rs_lds[i] = a.lut[i];
        for (int i = threadIdx.x; i < 2 * npos; i += blockDim.x) ((double *)(rs_lds + lut_bytes))[i] = a.term[i];
        for (int i = threadIdx.x; i < n_cnt; i += blockDim.x) l_cnt[i] = 0;
        __syncthreads();
    }
    // ---- tiles of 64 records.  Phase 1, lane per record: routing (rescale.py:300-342) and the records the fast
    // path can take: unchanged ones (qual_out already holds their qualities) and rescaled ones whose CIGAR is
    // [S] M [S].  Phase E, still lane per record: the two end windows of a fast record — candidate columns by a
    // byte-parallel test, LUT, MR.  Phase S (summary only): the aligned part of four fast records per step, eight
    // bytes per lane.
    const i64 ntiles = (a.n_reads + 63) / 64;
    u32 *__restrict__ my_list = a.gen_list + gwave * a.gen_cap;
    u32 n_list = 0;
    const int slot = lane >> 3, sl = lane & 7;     // phase S: eight runs per step, sixteen bytes per lane
    auto load8 = [](const u8 *ptr) -> u64 {
        const u32x2 v = *(const u32x2_u *)ptr;
        return (u64)v.x | ((u64)v.y << 32);
    };
    auto do_tile = [&](const i64 tile, const i64 ri, const bool valid) __attribute__((always_inline)) {
        if (a.copy_qual) {
            // qual_out starts as a copy of qual: the tile's own stretch of the column, 16 bytes per lane, before any
            // lane of this wavefront stores a rescaled byte into it (the stretch belongs to this tile alone: its first
            // and last partial 16 bytes are moved byte by byte, never a neighbour's).  The lines it reads are the ones
            // phase 1 needs the first quality of every record from.
            const i64 r1 = tile * 64 + 64 < a.n_reads ? tile * 64 + 64 : a.n_reads;
            const u32 b0 = a.seq_off[tile * 64], b1 = a.seq_off[r1];
            const u32 nu = (b1 - b0) >> 4;                           // whole 16-byte units, then up to 15 single bytes
            for (u32 u = (u32)lane; u < nu; u += 128u) {
                const u32 o0 = b0 + 16u * u, o1 = o0 + 1024u;
                const bool two = u + 64u < nu;
                const u32x4 v0 = *(const u32x4_u *)(a.qual + o0);
                u32x4 v1 = v0;
                if (two) v1 = *(const u32x4_u *)(a.qual + o1);
                *(u32x4_u *)(a.qual_out + o0) = v0;
                if (two) *(u32x4_u *)(a.qual_out + o1) = v1;
            }
            const u32 t0 = b0 + 16u * nu + (u32)lane;
            if (t0 < b1) a.qual_out[t0] = a.qual[t0];
        }
        u32 so = 0;
        int lseq = 0, qs = 0, nq = 0, st = 0, fwd_only = 0, rev = 0;
        int m1 = 0, gi = 0, gd = 0;   // a fast record is M(m1) [I(gi) | D(gd)] M(nq - m1 - gi) between its soft clips
        i64 rbase = 0;
        bool fast = false, handled = false;
        if (valid) {
            // first round trip: the record's columns; second: what they point at
            const u32 fl = a.flag[ri];
            so = a.seq_off[ri];
            lseq = (int)(a.seq_off[ri + 1] - so);
            const u32 co = a.cigar_off[ri];
            const int cn = (int)(a.cigar_off[ri + 1] - co);
            const int c_tid = a.tid[ri], c_pos = a.pos[ri], c_mtid = a.mtid[ri], c_mpos = a.mpos[ri];
            const u32 q_first = lseq > 0 ? ((fl & 0x4000u) ? 0u : (u32)a.qual[so]) : 0xFFu;      // (MDX_FLAG_HAS_QUAL)
            const u32 c0 = cn > 0 ? a.cigar[co] : 0u, c1 = cn > 1 ? a.cigar[co + 1] : 0u, c2 = cn > 2 ? a.cigar[co + 2] : 0u;
            const u32 c3 = cn > 3 ? a.cigar[co + 3] : 0u, c4 = cn > 4 ? a.cigar[co + 4] : 0u;
            const bool tid_ok = c_tid >= 0 && c_tid < a.n_contig;
            const i64 c_off0 = tid_ok ? a.contig_off[c_tid] : 0, c_off1 = tid_ok ? a.contig_off[c_tid + 1] : 0;
            rev = (fl >> 4) & 1;
            const int mate_rev = (fl >> 5) & 1;
            if (fl & 0x4) st = 0;
            else if (lseq == 0 || q_first == 0xFF) st = 1;
            else if (fl & 0x1) {
                const int pos = c_pos, mp = c_mpos;
                const bool same = c_tid == c_mtid;
                if ((!rev && mate_rev && mp > pos && same) || (rev && !mate_rev && mp < pos && same)) { st = 3; fwd_only = 1; }
                else st = 4;
            } else st = 2;
            const bool room = (i64)so + lseq + 16 <= a.n_bases;  // the 8- and 16-byte loads stay inside the columns
            if (st < 2 || st == 4) {
                // written back unchanged: qual_out already holds the record's qualities (mdx_rescale_device copies the
                // column before the launch), only the status and the MR marker are left to set
                a.status[ri] = (u8)st;
                a.mr_raw[ri] = __builtin_nan("");
                handled = true;
            } else if (room && cn >= 1 && cn <= 5) {
                // [S] M [S] or [S] M (I | D) M [S], both runs of the second form at least as long as the end windows
                auto is_m = [](u32 c) { const u32 o = c & 0xF; return o == 0 || o == 7 || o == 8; };
                const int lead = (c0 & 0xF) == 4 ? 1 : 0;
                const u32 cl = cn == 1 ? c0 : (cn == 2 ? c1 : (cn == 3 ? c2 : (cn == 4 ? c3 : c4)));
                const int trail = (cn > 1 && (cl & 0xF) == 4) ? 1 : 0;
                const int core = cn - lead - trail;
                const u32 k0 = lead ? c1 : c0, k1 = lead ? c2 : c1, k2 = lead ? c3 : c2;
                qs = lead ? (int)(c0 >> 4) : 0;
                const int clipr = trail ? (int)(cl >> 4) : 0;
                bool ok = (core == 1 || core == 3) && is_m(k0);
                m1 = (int)(k0 >> 4);
                int m2 = 0;
                if (core == 3) {
                    const int ox = k1 & 0xF, g = (int)(k1 >> 4);
                    const int wreq = a.len5p > a.len3p ? a.len5p : a.len3p;
                    m2 = (int)(k2 >> 4);
                    ok = ok && is_m(k2) && (ox == 1 || ox == 2) && g >= 1 && m1 >= 1 && m2 >= 1 && m1 >= wreq && m2 >= wreq;
                    gi = ox == 1 ? g : 0;
                    gd = ox == 2 ? g : 0;
                }
                nq = m1 + gi + m2;
                const i64 pos = c_pos;
                // (so + qs >= 8: a reverse-strand window is loaded as the eight bytes that end at its last column;
                //  nq, and with it every run, fits 16 bits of a staging entry)
                ok = ok && nq >= 1 && nq <= 0xFFFF && gd <= 0xFFFF && qs + nq + clipr == lseq && tid_ok && pos >= 0 &&
                     pos + m1 + gd + m2 <= c_off1 - c_off0 && so + (u32)qs >= 8u;
                if (ok) rbase = c_off0 + pos;
                fast = ok;
                if (!ok) { qs = 0; nq = 0; m1 = 0; gi = 0; gd = 0; }
            }
        }
        // (the tile's quality copy has long been written back — two round trips ago — but nothing orders the stores of
        //  different lanes to one address, so the rescaled bytes wait for it explicitly)
        if (a.copy_qual) __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
        const u64 m_fast = __ballot(fast);
        const bool walk = valid && !fast && !handled;   // left to rescale_walk_kernel
        const u64 m_gen = __ballot(walk);

        // ---- phase E: lane per fast record.  Columns [0, n5) and [s3, nq) in read orientation are the only ones
        // that can carry a key (_corr_this_base, rescale.py:49-79); every other column keeps its quality and adds 0.
        if (fast) {
            const u32 sb = so + (u32)qs;          // 32-bit offsets into the read / quality columns (scalar base pointers)
            // reference byte under query base qi: at rbase + qi in the left run, rbase + gd - gi + qi in the right one
            // (the same when there is no gap); the 5' window lies in the left run of a forward read, the right run of a
            // reverse one
            const int rshift = gd - gi;
            const int n5 = a.len5p < nq ? a.len5p : nq;
            const int s3 = nq - a.len3p > n5 ? nq - a.len3p : n5;
            const int n3 = fwd_only ? 0 : nq - s3;
            // stored pair (read byte | reference byte << 8) of a C>T / G>A column of the read's own strand
            const u32 pair0 = rev ? ('A' | 'G' << 8) : ('T' | 'C' << 8), pair1 = rev ? ('T' | 'C' << 8) : ('A' | 'G' << 8);
            double mr = 0.0;
            // A round takes up to 16 columns of the 5' window and 16 of the 3' window as four groups of eight bytes,
            // all fetched (read, reference, quality) before any is looked at — one round trip.  Byte j of a group is
            // column oq0 + j (a reverse-strand group is loaded from its far end and byte-swapped), so the candidates
            // come out in the reference's order: 5' window first, then the 3' window.  Windows longer than 16 take
            // their own rounds, all of the 5' window before the 3' one.
            const int ra = (a.len5p + 15) >> 4, rb = (a.len3p + 15) >> 4;
            const bool one = RS_EG == 4 && ra <= 1 && rb <= 1;
            const int rounds = one ? 1 : ra + rb;
            for (int r = 0; r < rounds; r++) {
                int w[2] = {0, 0}, c[2] = {0, 0};
                if (one) { c[0] = n5; w[1] = s3; c[1] = n3; }
                else if (r < ra) { w[0] = 16 * r; c[0] = n5 - 16 * r; }
                else { w[1] = s3 + 16 * (r - ra); c[1] = n3 - 16 * (r - ra); }
                u64 sg[4], rg[4], qg[4];
                int qi0[4];
#if RS_EG == 2
                const int hw = r < ra ? 0 : 1;         // a round is one window: two groups
#define RS_H(h) (hw * 2 + (h))
#else
#define RS_H(h) (h)
#endif
#pragma unroll
                for (int h0 = 0; h0 < RS_EG; h0++) {
                    const int h = RS_H(h0);
                    const int oq0 = w[h >> 1] + 8 * (h & 1), cnt = c[h >> 1] - 8 * (h & 1);
                    qi0[h0] = rev ? nq - 8 - oq0 : oq0;
                    sg[h0] = 0; rg[h0] = 0; qg[h0] = 0;
                    if (cnt > 0) {
                        sg[h0] = load8(a.seq + (u32)(sb + qi0[h0]));
                        rg[h0] = load8(a.ref + (rbase + (((h >> 1) ^ rev) ? rshift : 0) + qi0[h0]));
                        qg[h0] = load8(a.qual + (u32)(sb + qi0[h0]));
                    }
                }
#pragma unroll
                for (int h0 = 0; h0 < RS_EG; h0++) {
                    const int h = RS_H(h0);
                    const int oq0 = w[h >> 1] + 8 * (h & 1), cnt = c[h >> 1] - 8 * (h & 1);
                    if (cnt <= 0) continue;
                    u64 s8 = sg[h0], r8 = rg[h0], q8 = qg[h0];
                    if (rev) { s8 = __builtin_bswap64(s8); r8 = __builtin_bswap64(r8); q8 = __builtin_bswap64(q8); }
                    // a transition differs in bits 1 and 2 of the byte ('A'^'G' = 0x06, 'C'^'T' = 0x17), no other
                    // pair of bases does: the exact test is left to the few candidates
                    const u64 x = s8 ^ r8;
                    u64 cd = x & (x >> 1) & 0x0202020202020202ull & byte_range(0, cnt);
                    while (cd) {
                        const int sh = (__ffsll((long long)cd) - 1) & ~7;
                        cd &= cd - 1;
                        const u32 pr = ((u32)(s8 >> sh) & 0xFFu) | (((u32)(r8 >> sh) & 0xFFu) << 8);
                        const int sub = pr == pair0 ? 0 : (pr == pair1 ? 1 : -1);
                        if (sub < 0) continue;
                        const int oq = oq0 + (sh >> 3);
                        int pp = oq + 1;                                 // _corr_this_base, rescale.py:49-79
                        const int back = pp - nq - 1;
                        if (!fwd_only && pp >= -back) pp = back;
                        const int key = pp > 0 ? (pp <= a.len5p ? pp : 0) : (-pp <= a.len3p ? a.len5p - pp : 0);
                        const int ti = sub * npos + key;
                        mr += l_term[ti];                                // (x + 0.0 == x: a zero term changes nothing)
                        const u32 q = (u32)(q8 >> sh) & 0xFFu;
                        if (q <= 93) {
                            const u32 newq = l_lut[ti * 94 + q];
                            if (a.patch) patch_put(a.patch, a.n_patch, a.patch_cap, a.patch_parts, newq != q, (u32)(sb + (rev ? nq - 1 - oq : oq)), newq);
                            else if (newq != q) a.qual_out[(u32)(sb + (rev ? nq - 1 - oq : oq))] = (u8)newq;
                        }
                    }
                }
            }
            a.status[ri] = (u8)st;
            a.mr_raw[ri] = mr;
        }

        // ---- phase S: the substitution summary of the fast records (rescale.py:108-143), four records per step; the
        // first 128 columns of the next four are fetched before the current ones are counted
        if (a.subs && m_fast) {
            // staging entries (16 B): a run of columns [seq/qual byte offset, reference offset lo, reference offset hi (8)
            // | rev << 8 | 5'-only << 9 | deletion << 10 | first query base of the run << 16, columns | nq << 16]; a
            // deleted stretch is an entry of its own whose "read" is the reference itself (base counts, no transition)
            const int ne = fast ? (m1 + gi == nq ? 1 : (gd ? 3 : 2)) : 0;
            const int e0 = mbcnt64(__ballot(ne & 1), 0) + 2 * mbcnt64(__ballot(ne & 2), 0);
            if (fast) {
                const u32 fl2 = ((u32)rev << 8) | ((u32)fwd_only << 9), nqh = (u32)nq << 16;
                const u32 sb = so + (u32)qs;
                auto entry = [&](const u32 soff, const i64 roff, const u32 flags, const int qoff, const int len) {
                    return make_uint4(soff, (u32)(roff & 0xFFFFFFFFll), (u32)(roff >> 32) | flags | ((u32)qoff << 16), (u32)len | nqh);
                };
                stg[e0] = entry(sb, rbase, fl2, 0, m1 + gi == nq ? nq : m1);
                if (ne >= 2) stg[e0 + ne - 1] = entry(sb + m1 + gi, rbase + m1 + gd, fl2, m1 + gi, nq - m1 - gi);
                if (ne == 3) stg[e0 + 1] = entry(sb, rbase + m1, fl2 | (1u << 10), 0, gd);
            }
            const int nfast = rl(e0 + ne, 63);     // entries of the tile
            for (int i0 = 0; i0 < nfast; i0 += 8) {
                const bool sact = i0 + slot < nfast;
                const uint4 e = stg[sact ? i0 + slot : 0];
                const int s_nq = sact ? (int)(e.w & 0xFFFFu) : 0;        // columns of the run
                const int s_rev = (e.z >> 8) & 1, s_fwd = (e.z >> 9) & 1, s_del = (e.z >> 10) & 1;
                const int s_qoff = (int)(e.z >> 16), s_tot = (int)(e.w >> 16);
                const i64 rb = ((i64)(e.z & 0xFFu) << 32) | e.y;
                const u32 fx = s_rev ? 0x04040404u : 0u;    // A <-> T, C <-> G in the two class bits: counts in read orientation
                // passes of 128 columns per run (one, unless a run is longer)
                for (int off = 16 * sl; __ballot(off < s_nq); off += 128) {
                    const int nb = s_nq - off;                           // columns from this lane's first byte on
                    if (nb <= 0) continue;
                    // both loads in one round trip (the entry of a deleted stretch points at its record's first base;
                    // the columns hold 16 readable bytes behind every record the fast path takes, see `room`)
                    const u32x4 rv = *(const u32x4_u *)(a.ref + rb + off);
                    const u32x4 sl16 = *(const u32x4_u *)(a.seq + (e.x + (s_del ? 0u : (u32)off)));
                    const u32x4 sv = s_del ? rv : sl16;
                    const int n_lo = nb < 8 ? nb : 8, n_hi = nb < 16 ? nb - 8 : 8;
                    const u64 am0 = ~0ull >> (64 - 8 * n_lo), am1 = n_hi > 0 ? ~0ull >> (64 - 8 * n_hi) : 0ull;
                    const u32 am[4] = {(u32)am0, (u32)(am0 >> 32), (u32)am1, (u32)(am1 >> 32)};
                    u32 cany = 0, cd[4];
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        // subs[nt_ref] += 1 for every column (rescale.py:142-143).  Raw per-lane counts: valid bytes
                        // (bit 7 clear: A,C,G,T), class bit 1 set (C,G), class bit 2 set (T,G), both (G) — in read
                        // orientation; A,C,G,T follow at the end of the kernel
                        const u32 ok = ~rv[w] & 0x80808080u & am[w];
                        const u32 b1 = (rv[w] << 6) & ok, b2 = ((rv[w] ^ fx) << 5) & ok;
                        bc[0] += __popc(ok); bc[1] += __popc(b1); bc[2] += __popc(b2); bc[3] += __popc(b1 & b2);
                        // transitions (and junk bytes that look like one): bits 1 and 2 of the byte differ
                        const u32 x = sv[w] ^ rv[w];
                        cd[w] = x & (x >> 1) & 0x02020202u & am[w];
                        cany |= cd[w];
                    }
                    if (!cany) continue;
                    const u32x4 qv = *(const u32x4_u *)(a.qual + (e.x + (u32)off));
                    // one bit per candidate byte
                    u32 m16 = (((cd[0] >> 1) * 0x00204081u >> 21) & 0xFu) | (((cd[1] >> 1) * 0x00204081u >> 17) & 0xF0u) |
                              (((cd[2] >> 1) * 0x00204081u >> 13) & 0xF00u) | (((cd[3] >> 1) * 0x00204081u >> 9) & 0xF000u);
                    // read-orientation position of byte 0, and the step to byte j
                    const int oq0 = s_rev ? s_tot - 1 - s_qoff - off : s_qoff + off, dq = s_rev ? -1 : 1;
                    while (m16) {
                        const int j = __ffs((int)m16) - 1;
                        m16 &= m16 - 1;
                        const u32 bo = (u32)(j & 3) * 8u;
                        const int w = j >> 2;
                        const u32 qw = w == 0 ? qv[0] : (w == 1 ? qv[1] : (w == 2 ? qv[2] : qv[3]));
                        const u32 sw = w == 0 ? sv[0] : (w == 1 ? sv[1] : (w == 2 ? sv[2] : sv[3]));
                        const u32 rw = w == 0 ? rv[0] : (w == 1 ? rv[1] : (w == 2 ? rv[2] : rv[3]));
                        const u32 q = __builtin_amdgcn_ubfe(qw, bo, 8u);
                        const u32 pr = __builtin_amdgcn_ubfe(sw, bo, 8u) | (__builtin_amdgcn_ubfe(rw, bo, 8u) << 8);
                        // stored pair -> transition of the read's own strand: 0 C>T, 1 G>A (rescaled), 2 T>C, 3 A>G;
                        // -1: not a transition of two bases (sums of 0/1 terms: no branches)
                        const int kind = (int)(pr == ('T' | 'C' << 8)) * (1 + s_rev) + (int)(pr == ('A' | 'G' << 8)) * (2 - s_rev) +
                                         (int)(pr == ('C' | 'T' << 8)) * (3 + s_rev) + (int)(pr == ('G' | 'A' << 8)) * (4 - s_rev) - 1;
                        int pp = oq0 + dq * j + 1;                           // _corr_this_base, rescale.py:49-79
                        const int back = pp - s_tot - 1;
                        pp = (!s_fwd && pp >= -back) ? back : pp;
                        const int k5 = pp <= a.len5p ? pp : 0, k3 = -pp <= a.len3p ? a.len5p - pp : 0;
                        const int key = pp > 0 ? k5 : k3;
                        // "before" words of T>C / A>G, or the occurrences of (substitution, key, old quality)
                        const int idx = kind >= 2 ? (kind == 2 ? 2 : 6) * 94 : 752 + (kind * npos + key) * 94;
                        if (kind >= 0 && q <= 93) atomicAdd(&l_cnt[idx + (int)q], 1u);
                    }
                }
            }
        }
        if (m_gen) {
            if (walk) my_list[n_list + (u32)mbcnt64(m_gen, 0)] = (u32)ri;
            n_list += (u32)__popcll(m_gen);
        }
    };
    if (a.in_list) {
        // behind the fused kernel: the records its wavefronts listed, 64 at a time (qual_out is complete: no copy).
        // (Measured against a scan of all tiles for records marked in their status: 0.51 against 0.71 ms per 25 M records
        // of config 5 — a tile costs its round trips however few of its lanes are busy.)
        for (i64 l = gwave; l < a.n_in; l += nwaves) {
            const u32 *__restrict__ in = a.in_list + l * a.in_cap;
            const u32 n = a.in_count[l];
            for (u32 k0 = 0; k0 < n; k0 += 64) do_tile(0, k0 + lane < n ? (i64)in[k0 + lane] : 0, k0 + lane < n);
        }
    } else {
        for (i64 tile = gwave; tile < ntiles; tile += nwaves) do_tile(tile, tile * 64 + lane, tile * 64 + lane < a.n_reads);
    }
    if (lane == 0) a.gen_count[gwave] = n_list;
    if (a.subs) {
        // The block's counters go to its own row of subs_part (plain stores; rescale_reduce_kernel adds the rows up):
        // atomics of every block on the same few thousand words cost 0.3 ms per launch whatever its size.
        // The four reference-base counts of the block are collected in the first counter words no transition uses
        // ("before" of C>T).
        __syncthreads();
        {
            u32 v[4];
            for (int b = 0; b < 4; b++) {
                v[b] = bc[b];
                for (int o = 32; o; o >>= 1) v[b] += __shfl_xor(v[b], o);
            }
            // valid, bit 1 (C,G), bit 2 (T,G), both (G) -> A, C, G, T
            if (lane == 0) {
                atomicAdd(&l_cnt[0], v[0] - v[1] - v[2] + v[3]);
                atomicAdd(&l_cnt[1], v[1] - v[3]);
                atomicAdd(&l_cnt[2], v[3]);
                atomicAdd(&l_cnt[3], v[2] - v[3]);
            }
        }
        __syncthreads();
        u32 *__restrict__ row = a.subs_part + (size_t)blockIdx.x * n_cnt;
        for (int i = threadIdx.x; i < n_cnt; i += blockDim.x) row[i] = l_cnt[i];
    }
}

// The records rescale_kernel leaves out — any CIGAR — one lane per record: the lane walks the record's CIGAR in the
// read's own 5'->3' order (operations and bytes backwards on the reverse strand), eight columns of a match run at a
// time, so that MR is summed in the reference's order (rescale.py:226-262).  Wavefront w takes the list
// rescale_kernel's wavefront w wrote (a.gen_list), 64 records at a time, or — without that kernel (tables too large
// for its LDS image, or key 0 not the identity) — every (number of wavefronts)-th tile of the batch.
__global__ __launch_bounds__(RS_BLOCK) void rescale_walk_kernel(MdxRescaleArgs a) {
    const int lane = threadIdx.x & 63;
    const i64 gwave = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const i64 nwaves = ((i64)gridDim.x * blockDim.x) >> 6;
    const int npos = 1 + a.len5p + a.len3p;
    u32 bc[4] = {0, 0, 0, 0};   // summary (rescale.py:108-143): reference bases A,C,G,T in read orientation, per lane
    // In the LDS when they fit (a.lds_tables): [lut 2 npos 94 B, padded][term 2 npos f64][counters u32: 4 x 2 x 94
    // transitions | 2 x npos x 94 rescaled-column kinds], the counters flushed at block end.
    extern __shared__ __attribute__((aligned(16))) u8 rs_lds[];
    const int lut_bytes = (2 * npos * 94 + 15) & ~15, n_cnt = 752 + 2 * npos * 94;
    u32 *const l_cnt = (u32 *)(rs_lds + lut_bytes + 2 * npos * 8);
    if (a.lds_tables) {
        for (int i = threadIdx.x; i < 2 * npos * 94; i += blockDim.x) rs_lds[i] = a.lut[i];
        for (int i = threadIdx.x; i < 2 * npos; i += blockDim.x) ((double *)(rs_lds + lut_bytes))[i] = a.term[i];
        for (int i = threadIdx.x; i < n_cnt; i += blockDim.x) l_cnt[i] = 0;
        __syncthreads();
    }
    const u8 *const t_lut = a.lds_tables ? (const u8 *)rs_lds : a.lut;
    const double *const t_term = a.lds_tables ? (const double *)(rs_lds + lut_bytes) : a.term;
    // summary word `idx` (>= 4) of include/mdx.h += 1
    auto sub_bump = [&](const int idx) {
        if (a.lds_tables) atomicAdd(&l_cnt[idx - 4], 1u);
        else atomicAdd(&a.subs[idx], 1ull);
    };

    // ---- one record by the whole wavefront, any CIGAR: one lane per query base, CIGAR walked per base, column by
    // column as the reference does it.  Only for what the lane walk below cannot follow: a reverse-strand read with a
    // reference skip (see there).
    auto generic = [&](const i64 ri) __attribute__((always_inline)) {
        const u32 fl = a.flag[ri];
        const u32 so = a.seq_off[ri];
        const int lseq = (int)(a.seq_off[ri + 1] - so);
        const u32 co = a.cigar_off[ri];
        const int cn = (int)(a.cigar_off[ri + 1] - co);
        const u8 *__restrict__ qin = a.qual + so;
        // (patch mode: no second column — what is written back unchanged is not written at all)
        const bool to_list = a.patch != nullptr;
        u8 *__restrict__ qout = to_list ? nullptr : a.qual_out + so;
        const int rev = (fl >> 4) & 1, mate_rev = (fl >> 5) & 1;
        // record routing, rescale.py:300-342
        int st, forward_only = 0;
        if (fl & 0x4) st = 0;
        else if (lseq == 0 || qin[0] == 0xFF) st = 1;
        else if (fl & 0x1) {
            const int pos = a.pos[ri], mp = a.mpos[ri];
            const bool same = a.tid[ri] == a.mtid[ri];
            if ((!rev && mate_rev && mp > pos && same) || (rev && !mate_rev && mp < pos && same)) { st = 3; forward_only = 1; }
            else st = 4;
        } else st = 2;
        if (lane == 0) { a.status[ri] = (u8)st; a.mr_raw[ri] = __builtin_nan(""); }
        if (st < 2 || st == 4) {
            if (!to_list) for (int b = lane; b < lseq; b += 64) qout[b] = qin[b];
            return;
        }
        // CIGAR: one op per lane; scan by lane 0's view via readlane
        const u32 op_lane = lane < cn ? a.cigar[co + lane] : 0u;
        auto op_at = [&](int k) -> u32 { return cn <= 64 ? (u32)rl((int)op_lane, k) : a.cigar[co + k]; };
        int qs = 0, clipr = 0, rlen = 0, ncols = 0, nI = 0, qcons = 0;
        bool leading = true;
        for (int k = 0; k < cn; k++) {
            const u32 c = op_at(k);
            const int op = c & 0xF, len = (int)(c >> 4);
            if (leading) { if (op == 4) qs += len; else if (op != 5) leading = false; }
            if (op == 0 || op == 7 || op == 8) { ncols += len; rlen += len; qcons += len; }
            else if (op == 1) { ncols += len; nI += len; qcons += len; }
            else if (op == 2) { ncols += len; rlen += len; }
            else if (op == 3) rlen += len;
        }
        for (int k = cn - 1; k >= 1; k--) {
            const u32 c = op_at(k);
            const int op = c & 0xF;
            if (op == 5) continue;
            if (op == 4) clipr += (int)(c >> 4); else break;
        }
        const int nq = lseq - qs - clipr > 0 ? lseq - qs - clipr : 0;
        const int n0 = rlen ? rlen : 1;
        const int nrg = n0 + nI;
        const int tid = a.tid[ri];
        const i64 pos = a.pos[ri];
        bool bad = cn == 0 || tid < 0 || tid >= a.n_contig || pos < 0 || nq != qcons;
        i64 rbase = 0;
        if (!bad) {
            const i64 c0 = a.contig_off[tid];
            bad = pos + n0 > a.contig_off[tid + 1] - c0;
            rbase = c0 + pos;
        }
        // rescale.py:266-271 re-attaches clips only when the first / last op is S: any other clip
        // layout (H before S) leaves a quality string of the wrong length, which pysam rejects
        if (!bad) {
            const u32 f = op_at(0), l = op_at(cn - 1);
            const int pre = (f & 0xF) == 4 ? (int)(f >> 4) : 0, suf = (l & 0xF) == 4 ? (int)(l >> 4) : 0;
            bad = pre != qs || suf != clipr || (cn == 1 && (f & 0xF) == 4);
        }
        if (bad) {
            if (lane == 0) flag_error(a.err, ri, ERR_BAD_READ);
            if (!to_list) for (int b = lane; b < lseq; b += 64) qout[b] = qin[b];
            return;
        }
        // soft-clipped qualities are kept
        if (!to_list) {
            for (int b = lane; b < qs; b += 64) qout[b] = qin[b];
            for (int b = qs + nq + lane; b < lseq; b += 64) qout[b] = qin[b];
        }

        const i8 *__restrict__ rp = (const i8 *)a.ref + rbase;
        const u8 *__restrict__ sp = a.seq + so + qs;
        // reference byte under gapped-reference column jr (-1: an insertion gap)
        auto ref_at = [&](const int jr) -> int {
            int c2 = 0, shift = 0, rix = -2;
            for (int k = 0; k < cn && rix == -2; k++) {
                const u32 c = op_at(k);
                const int op = c & 0xF, len = (int)(c >> 4);
                if (op == 1) {
                    if (jr < c2) rix = jr - shift;
                    else if (jr < c2 + len) rix = -1;
                    shift += len; c2 += len;
                } else if (op == 0 || op == 7 || op == 8 || op == 2) c2 += len;
            }
            if (rix == -2) rix = jr - shift;
            return rix < 0 ? -1 : (int)rp[rix];
        };
        // subs[nt_ref] += 1 (rescale.py:142-143): valid reference bytes are 'A','C','G','T'
        auto count_ref = [&](const int rch) {
            if (rch >= 0) {
                const int k = (rch >> 1) & 3;       // A,C,T,G
                int b = k ^ (k >> 1);               // A,C,G,T
                if (rev) b = 3 - b;                 // complemented on the reverse strand
                bc[0] += b == 0; bc[1] += b == 1; bc[2] += b == 2; bc[3] += b == 3;
            }
        };
        double mr = 0.0;
        for (int base = 0; base < nq; base += 64) {
            const int oq = base + lane;                 // query base in read orientation (0 = 5' end)
            double term = 0.0;
            if (oq < nq) {
                const int qi = rev ? nq - 1 - oq : oq;  // forward query index
                // gapped-read column of query base qi, then the gapped-reference column facing it
                // (each string is reversed from its own end on the reverse strand, rescale.py:221-224)
                int col = 0, qoff = 0, js = -1;
                for (int k = 0; k < cn && js < 0; k++) {
                    const u32 c = op_at(k);
                    const int op = c & 0xF, len = (int)(c >> 4);
                    if (op == 0 || op == 7 || op == 8 || op == 1) {
                        if (qi < qoff + len) js = col + (qi - qoff);
                        col += len; qoff += len;
                    } else if (op == 2) col += len;
                }
                const int rch = ref_at(rev ? nrg - ncols + js : js);
                const u32 ch = sp[qi];
                const u32 q = qin[qs + qi];
                // read-orientation pair (T,C) -> C>T ; (A,G) -> G>A; complemented on the reverse strand
                int sub = -1;
                if (!rev) { if (ch == 'T' && rch == 'C') sub = 0; else if (ch == 'A' && rch == 'G') sub = 1; }
                else { if (ch == 'A' && rch == 'G') sub = 0; else if (ch == 'T' && rch == 'C') sub = 1; }
                u32 newq = q;
                int skey = 0;
                if (sub >= 0) {
                    // _corr_this_base, rescale.py:49-79
                    int p = oq + 1;
                    const int back = p - nq - 1;
                    if (!forward_only && p >= -back) p = back;
                    const int key = p > 0 ? (p <= a.len5p ? p : 0) : (-p <= a.len3p ? a.len5p - p : 0);
                    skey = key;
                    term = a.term[sub * npos + key];
                    if (q <= 93) newq = a.lut[(sub * npos + key) * 94 + q];
                }
                if (to_list) patch_put(a.patch, a.n_patch, a.patch_cap, a.patch_parts, newq != q, so + (u32)(qs + qi), newq);
                else qout[qs + qi] = (u8)newq;
                if (a.subs) {
                    // _record_subs (rescale.py:108-143): transitions by old/new quality, reference bases
                    count_ref(rch);
                    int st = sub == 0 ? 0 : (sub == 1 ? 2 : -1);   // 0 CT, 1 TC, 2 GA, 3 AG
                    if (st < 0) {
                        const bool cg = rev ? (ch == 'G' && rch == 'A') : (ch == 'C' && rch == 'T');
                        const bool ga = rev ? (ch == 'C' && rch == 'T') : (ch == 'G' && rch == 'A');
                        st = cg ? 1 : (ga ? 3 : -1);
                    }
                    // (one counter per column, as in the lane walk)
                    if (st >= 0 && q <= 93) {
                        if (sub >= 0) sub_bump(756 + (sub * npos + skey) * 94 + q);
                        else sub_bump(4 + (st * 2 + 0) * 94 + q);
                    }
                }
            }
            // ordered fp64 accumulation of the non-zero terms (x + 0.0 == x exactly)
            u64 nz = __ballot(term != 0.0);
            while (nz) {
                const int l = __ffsll((long long)nz) - 1;
                nz &= nz - 1;
                const int lo = rl(__double2loint(term), l), hi = rl(__double2hiint(term), l);
                mr += __hiloint2double(hi, lo);
            }
        }
        if (lane == 0) a.mr_raw[ri] = mr;
        if (a.subs) {
            // deletion columns pair '-' with a reference base (counted while read bases remain in the
            // iteration order: `if pos_on_read < length_read`, rescale.py:252)
            int col = 0, qoff = 0;
            for (int k = 0; k < cn; k++) {
                const u32 c = op_at(k);
                const int op = c & 0xF, len = (int)(c >> 4);
                if (op == 0 || op == 7 || op == 8 || op == 1) { col += len; qoff += len; }
                else if (op == 2) {
                    if (rev ? qoff > 0 : qoff < nq)
                        for (int t = lane; t < len; t += 64) count_ref(ref_at(rev ? nrg - ncols + col + t : col + t));
                    col += len;
                }
            }
        }
    };

    // ---- one record by one lane; true: left to the whole wavefront
    auto walk = [&](const i64 ri) __attribute__((always_inline)) -> bool {
        const u32 fl = a.flag[ri];
        const u32 so = a.seq_off[ri];
        const int lseq = (int)(a.seq_off[ri + 1] - so);
        const u32 co = a.cigar_off[ri];
        const int cn = (int)(a.cigar_off[ri + 1] - co);
        const int rev = (fl >> 4) & 1, mate_rev = (fl >> 5) & 1;
        const int tid = a.tid[ri];
        const i64 pos = a.pos[ri];
        // record routing, rescale.py:300-342
        int st, fwd_only = 0;
        if (fl & 0x4) st = 0;
        else if (lseq == 0 || a.qual[so] == 0xFF) st = 1;
        else if (fl & 0x1) {
            const int mp = a.mpos[ri];
            const bool same = tid == a.mtid[ri];
            if ((!rev && mate_rev && mp > pos && same) || (rev && !mate_rev && mp < pos && same)) { st = 3; fwd_only = 1; }
            else st = 4;
        } else st = 2;
        a.status[ri] = (u8)st;
        a.mr_raw[ri] = __builtin_nan("");
        if (st < 2 || st == 4) return false;     // written back unchanged: qual_out starts as a copy of qual
        // CIGAR: clips, spans
        auto opk = [&](const int k) -> u32 { return a.cigar[co + k]; };
        int qs = 0, clipr = 0, rlen = 0, qcons = 0, n_skip = 0;
        u32 c_first = 0, c_last = 0;
        bool leading = true;
        for (int k = 0; k < cn; k++) {
            const u32 c = opk(k);
            const int op = c & 0xF, len = (int)(c >> 4);
            if (k == 0) c_first = c;
            c_last = c;
            if (leading) { if (op == 4) qs += len; else if (op != 5) leading = false; }
            if (op == 0 || op == 7 || op == 8) { rlen += len; qcons += len; }
            else if (op == 1) qcons += len;
            else if (op == 2) rlen += len;
            else if (op == 3) { rlen += len; n_skip += len; }
            // soft clips behind the last operation that is not a clip (the first operation never counts)
            if (k >= 1) { if (op == 4) clipr += len; else if (op != 5) clipr = 0; }
        }
        const int nq = lseq - qs - clipr > 0 ? lseq - qs - clipr : 0;
        const int n0 = rlen ? rlen : 1;
        bool bad = cn == 0 || tid < 0 || tid >= a.n_contig || pos < 0 || nq != qcons;
        i64 rbase = 0;
        if (!bad) {
            const i64 c0 = a.contig_off[tid];
            bad = pos + n0 > a.contig_off[tid + 1] - c0;
            rbase = c0 + pos;
        }
        // rescale.py:266-271 re-attaches clips only when the first / last op is S: any other clip
        // layout (H before S) leaves a quality string of the wrong length, which pysam rejects
        if (!bad) {
            const int pre = (c_first & 0xF) == 4 ? (int)(c_first >> 4) : 0, suf = (c_last & 0xF) == 4 ? (int)(c_last >> 4) : 0;
            bad = pre != qs || suf != clipr || (cn == 1 && (c_first & 0xF) == 4);
        }
        if (bad) { flag_error(a.err, ri, ERR_BAD_READ); return false; }

        const u32 sb = so + (u32)qs;
        const bool room = (i64)so + lseq + 8 <= a.n_bases;      // eight bytes can be loaded from any byte of the record
        auto col8 = [&](const u8 *__restrict__ colp, const u32 off, const int cnt) -> u64 {
            if (room) { const u32x2 v = *(const u32x2_u *)(colp + off); return (u64)v.x | ((u64)v.y << 32); }
            u64 v = 0;
            for (int j = 0; j < cnt; j++) v |= (u64)colp[off + j] << (8 * j);
            return v;
        };
        auto count_bases = [&](const u64 r64, const u64 am) {
            // subs[nt_ref] += 1 (rescale.py:142-143): A,C,G,T of the reference, complemented on the reverse strand
            const u64 ok7 = ~r64 & 0x8080808080808080ull & am;   // bit 7 clear: a base
            const u64 b1 = (r64 << 6) & ok7, b2 = (r64 << 5) & ok7;      // bit 1, bit 2 of the byte
            const int nA = __popcll(ok7 & ~b1 & ~b2), nC = __popcll(b1 & ~b2), nT = __popcll(~b1 & b2), nG = __popcll(b1 & b2);
            if (rev) { bc[0] += nT; bc[1] += nG; bc[2] += nC; bc[3] += nA; }
            else { bc[0] += nA; bc[1] += nC; bc[2] += nG; bc[3] += nT; }
        };
        double mr = 0.0;
        // A reverse-strand read with a reference skip: the reference's alignment strings hold gaps for insertions and
        // deletions only (align.py:53-73), the fetched reference still holds the skipped stretch, and both strings
        // are reversed from their own ends (rescale.py:221-224) — read column js then faces column js + (skipped
        // bases) of the gapped reference, whose insertion gaps stay where the forward walk put them.  No runs to
        // follow: left to `generic`.
        if (rev && n_skip > 0) return true;
        int q = rev ? nq : 0, r = rev ? rlen : 0;     // query bases / reference bases in front of the next operation
        for (int t = 0; t < cn; t++) {
            const u32 c = opk(rev ? cn - 1 - t : t);
            const int op = c & 0xF, len = (int)(c >> 4);
            const int step = rev ? -len : len;
            if (op == 0 || op == 7 || op == 8) {
                // RS_WG groups of eight columns at a time, all fetched before any is looked at: a lane waits for every
                // round trip to memory, and little else runs beside it in this kernel
                for (int done = 0; done < len; done += 8 * RS_WG) {
                    u64 sg[RS_WG], rg[RS_WG], cdg[RS_WG];
                    int q0g[RS_WG];
#pragma unroll
                    for (int g = 0; g < RS_WG; g++) {
                        const int d = done + 8 * g, cnt = len - d < 8 ? len - d : 8;
                        q0g[g] = rev ? q - d - cnt : q + d;
                        sg[g] = 0; rg[g] = 0;
                        if (cnt > 0) {
                            sg[g] = col8(a.seq, sb + (u32)q0g[g], cnt);
                            const u32x2 rv = *(const u32x2_u *)(a.ref + rbase + (rev ? r - d - cnt : r + d));   // (guard band)
                            rg[g] = (u64)rv.x | ((u64)rv.y << 32);
                        }
                    }
                    u64 any = 0;
#pragma unroll
                    for (int g = 0; g < RS_WG; g++) {
                        const int d = done + 8 * g, cnt = len - d < 8 ? len - d : 8;
                        const u64 am = cnt > 0 ? byte_range(0, cnt) : 0ull;
                        if (a.subs) count_bases(rg[g], am);
                        const u64 x = sg[g] ^ rg[g];
                        cdg[g] = x & (x >> 1) & 0x0202020202020202ull & am;   // transitions (and junk bytes that look like one)
                        any |= cdg[g];
                    }
                    if (!any) continue;
                    u64 qg[RS_WG];
#pragma unroll
                    for (int g = 0; g < RS_WG; g++) {
                        const int d = done + 8 * g, cnt = len - d < 8 ? len - d : 8;
                        qg[g] = cdg[g] ? col8(a.qual, sb + (u32)q0g[g], cnt) : 0ull;
                    }
#pragma unroll
                    for (int g = 0; g < RS_WG; g++) {
                        u64 cd = cdg[g];
                        const u64 s64 = sg[g], r64 = rg[g], q64 = qg[g];
                        const int q0 = q0g[g];
                        while (cd) {
                            // the next candidate in read order
                            const int sh = (rev ? 63 - __builtin_clzll(cd) : __ffsll((long long)cd) - 1) & ~7;
                            cd &= ~(0xFFull << sh);
                            const u32 pr = ((u32)(s64 >> sh) & 0xFFu) | (((u32)(r64 >> sh) & 0xFFu) << 8);
                            // stored pair -> transition of the read's own strand: 0 C>T, 1 G>A (rescaled), 2 T>C, 3 A>G
                            int kind = -1;
                            if (pr == ('T' | 'C' << 8)) kind = rev;
                            else if (pr == ('A' | 'G' << 8)) kind = 1 - rev;
                            else if (pr == ('C' | 'T' << 8)) kind = 2 + rev;
                            else if (pr == ('G' | 'A' << 8)) kind = 3 - rev;
                            if (kind < 0) continue;
                            const u32 qv = (u32)(q64 >> sh) & 0xFFu;
                            if (kind < 2) {
                                const int qi = q0 + (sh >> 3);
                                int pp = (rev ? nq - 1 - qi : qi) + 1;          // _corr_this_base, rescale.py:49-79
                                const int back = pp - nq - 1;
                                if (!fwd_only && pp >= -back) pp = back;
                                const int key = pp > 0 ? (pp <= a.len5p ? pp : 0) : (-pp <= a.len3p ? a.len5p - pp : 0);
                                const int ti = kind * npos + key;
                                mr += t_term[ti];                                // (x + 0.0 == x: a zero term changes nothing)
                                if (qv <= 93) {
                                    const u32 newq = t_lut[ti * 94 + qv];
                                    if (a.patch) patch_put(a.patch, a.n_patch, a.patch_cap, a.patch_parts, newq != qv, sb + (u32)qi, newq);
                                    else if (newq != qv) a.qual_out[sb + (u32)qi] = (u8)newq;
                                    if (a.subs) sub_bump(756 + ti * 94 + qv);
                                }
                            } else if (qv <= 93 && a.subs) {
                                sub_bump(4 + (kind == 2 ? 2 : 6) * 94 + qv);    // "before" words of T>C / A>G
                            }
                        }
                    }
                }
                q += step; r += step;
            } else if (op == 1) {
                q += step;
            } else if (op == 2) {
                // deletion columns pair '-' with a reference base, counted while read bases remain in the
                // iteration order (`if pos_on_read < length_read`, rescale.py:252)
                if (a.subs && (rev ? q > 0 : q < nq)) {
                    const int r0 = rev ? r - len : r;
                    for (int done = 0; done < len; done += 8) {
                        const u32x2 rv = *(const u32x2_u *)(a.ref + rbase + r0 + done);
                        count_bases((u64)rv.x | ((u64)rv.y << 32), byte_range(0, len - done < 8 ? len - done : 8));
                    }
                }
                r += step;
            }
            // (a reference skip, N, moves nothing: the reference's alignment strings know insertions and deletions
            //  only — align.py:53-73 — so the bases behind a skip face the skipped stretch itself)
        }
        a.mr_raw[ri] = mr;
        return false;
    };

    // 64 records at a time, then one by one those the lanes handed back
    auto pass = [&](const bool have, const i64 ri) __attribute__((always_inline)) {
        const bool hand = have && walk(ri);
        u64 m = __ballot(hand);
        while (m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1;
            generic(((i64)rl((int)(ri >> 32), j) << 32) | (u32)rl((int)(ri & 0xFFFFFFFFll), j));
        }
    };
    if (a.gen_list) {
        const u32 *__restrict__ mine = a.gen_list + gwave * a.gen_cap;
        const u32 n = a.gen_count[gwave];
        for (u32 k0 = 0; k0 < n; k0 += 64) pass(k0 + lane < n, k0 + lane < n ? (i64)mine[k0 + lane] : 0);
    } else {
        const i64 ntiles = (a.n_reads + 63) / 64;
        for (i64 tile = gwave; tile < ntiles; tile += nwaves) pass(tile * 64 + lane < a.n_reads, tile * 64 + lane);
    }
    if (a.subs && a.lds_tables) {
        // the block's own row of subs_part, as in rescale_kernel (rows a.row_base ..)
        __syncthreads();
        for (int b = 0; b < 4; b++) {
            u32 v = bc[b];
            for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0 && v) atomicAdd(&l_cnt[b], v);
        }
        __syncthreads();
        u32 *__restrict__ row = a.subs_part + (size_t)(a.row_base + blockIdx.x) * n_cnt;
        for (int i = threadIdx.x; i < n_cnt; i += blockDim.x) row[i] = l_cnt[i];
    } else if (a.subs) {
        for (int b = 0; b < 4; b++) {
            u32 v = bc[b];
            for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0 && v) atomicAdd(&a.subs[b], (u64)v);
        }
    }
}

// subs[4 + i] += sum over the blocks' rows of word i; words 0..3 of a row are the block's reference-base counts.
// blockIdx.y picks every RS_RED_Y-th row (one thread walking all rows of a word took 0.18 ms on its own).
#define RS_RED_Y 32
__global__ void rescale_reduce_kernel(const u32 *__restrict__ part, int rows, int n_cnt, u64 *__restrict__ subs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cnt) return;
    u64 v = 0;
#pragma unroll 4
    for (int r = blockIdx.y; r < rows; r += RS_RED_Y) v += part[(size_t)r * n_cnt + i];
    if (v) atomicAdd(&subs[i < 4 ? i : 4 + i], v);
}

static size_t rs_lds_bytes(int npos, bool staging) {
    return (size_t)((2 * npos * 94 + 15) & ~15) + (size_t)2 * npos * 8 + (((size_t)(752 + 2 * npos * 94) * 4 + 15) & ~(size_t)15) +
           (staging ? (size_t)(RS_BLOCK / 64) * RS_STG * 16 : 0);
}

void mdx_k_rescale(const MdxRescaleArgs &a0, int n_cu, hipStream_t s) {
    if (a0.n_reads <= 0) return;
    MdxRescaleArgs a = a0;
    const int npos = 1 + a.len5p + a.len3p;
    const size_t need = rs_lds_bytes(npos, true);
    a.lds_tables = (a.key0_plain && 2 * npos < 255 && need <= 60 * 1024 && a.gen_list && a.gen_count && a.subs_part) ? 1 : 0;
    // one launch-sized grid (RS_BPC blocks per CU); the tiles are dealt round-robin to the wavefronts
    const int64_t want = (a.n_reads + RS_BLOCK - 1) / RS_BLOCK;
    const int grid = (int)(want < (int64_t)n_cu * RS_BPC ? want : (int64_t)n_cu * RS_BPC);
    const int n_cnt = 752 + 2 * npos * 94;
    if (a.lds_tables) {
        a.copy_qual = (a.qual_out != a.qual && !a.patch) ? 1 : 0;      // the fast kernel copies the quality column as it goes
        if (need > 48 * 1024)
            (void)hipFuncSetAttribute((const void *)rescale_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
        hipLaunchKernelGGL(rescale_kernel, dim3(grid), dim3(RS_BLOCK), need, s, a);
        // what it left out (same grid: wavefront w reads the list wavefront w wrote), then the summary rows of both
        a.row_base = grid;
        hipLaunchKernelGGL(rescale_walk_kernel, dim3(grid), dim3(RS_BLOCK), rs_lds_bytes(npos, false), s, a);
        if (a.subs)
            hipLaunchKernelGGL(rescale_reduce_kernel, dim3((n_cnt + 255) / 256, RS_RED_Y), dim3(256), 0, s, a.subs_part, 2 * grid, n_cnt, a.subs);
    } else {
        // no fast path: every record by the walk; summary counters in the LDS when those alone fit
        const size_t walk_lds = rs_lds_bytes(npos, false);
        if (a.qual_out != a.qual && !a.patch)
            (void)hipMemcpyAsync(a.qual_out, a.qual, (size_t)a.n_bases, hipMemcpyDeviceToDevice, s);
        a.gen_list = nullptr;
        a.row_base = 0;
        a.lds_tables = (2 * npos < 255 && walk_lds <= 60 * 1024 && a.subs_part) ? 1 : 0;
        if (a.lds_tables && walk_lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void *)rescale_walk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_lds);
        hipLaunchKernelGGL(rescale_walk_kernel, dim3(grid), dim3(RS_BLOCK), a.lds_tables ? walk_lds : 0, s, a);
        if (a.subs && a.lds_tables)
            hipLaunchKernelGGL(rescale_reduce_kernel, dim3((n_cnt + 255) / 256, RS_RED_Y), dim3(256), 0, s, a.subs_part, grid, n_cnt, a.subs);
    }
}

void mdx_k_rescale_lists_pass(const MdxRescaleArgs &a0, int fused_rows, int n_cu, hipStream_t s) {
    MdxRescaleArgs a = a0;
    const int npos = 1 + a.len5p + a.len3p, n_cnt = 752 + 2 * npos * 94;
    const size_t need = rs_lds_bytes(npos, true);
    static_assert(RS_BPC * RS_BLOCK >= MDX_FUSE_BLOCK, "a wavefront of rescale_kernel per list of the fused kernel");
    a.lds_tables = 1;
    a.copy_qual = 0;
    // every wavefront of the fused kernel has a list: as many wavefronts here, at least (a wavefront takes the lists
    // l = its index, + the number of wavefronts, ...; its own list for the walk kernel holds what it leaves out)
    int64_t want = ((int64_t)a.n_in * 64 + RS_BLOCK - 1) / RS_BLOCK;
    if (want < 1) want = 1;
    const int grid = (int)(want < (int64_t)n_cu * RS_BPC ? want : (int64_t)n_cu * RS_BPC);
    if (need > 48 * 1024)
        (void)hipFuncSetAttribute((const void *)rescale_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
    a.subs_part = a0.subs_part + (size_t)fused_rows * n_cnt;
    hipLaunchKernelGGL(rescale_kernel, dim3(grid), dim3(RS_BLOCK), need, s, a);
    a.in_list = nullptr; a.in_count = nullptr; a.n_in = 0;
    a.row_base = grid;
    hipLaunchKernelGGL(rescale_walk_kernel, dim3(grid), dim3(RS_BLOCK), rs_lds_bytes(npos, false), s, a);
    if (a.subs)
        hipLaunchKernelGGL(rescale_reduce_kernel, dim3((n_cnt + 255) / 256, RS_RED_Y), dim3(256), 0, s, a0.subs_part,
                           fused_rows + 2 * grid, n_cnt, a.subs);
}

// wavefronts of a launch over n_reads records, and the list entries each may need (its tiles x 64)
void mdx_k_rescale_lists(int64_t n_reads, int n_cu, int64_t *n_waves, int64_t *cap) {
    const int64_t want = (n_reads + RS_BLOCK - 1) / RS_BLOCK;
    const int64_t grid = want < (int64_t)n_cu * RS_BPC ? want : (int64_t)n_cu * RS_BPC;
    const int64_t nw = grid * (RS_BLOCK / 64), ntiles = (n_reads + 63) / 64;
    *n_waves = nw > 0 ? nw : 1;
    *cap = ((ntiles + *n_waves - 1) / *n_waves) * 64;
}

// a patch list applied: qual_out (a copy of the quality column, or the column itself) takes the new Phred of every entry;
// blockIdx.y = the part of the list
__global__ void rescale_expand_kernel(u8 *__restrict__ qual_out, const u64 *__restrict__ patch, const u64 *__restrict__ n_patch, long long cap,
                                      i64 n_bases) {
    const u64 n = n_patch[blockIdx.y] < (u64)cap ? n_patch[blockIdx.y] : (u64)cap;
    const u64 *__restrict__ mine = patch + (size_t)blockIdx.y * (size_t)cap;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 e = mine[i];
        const u32 idx = (u32)e;
        if ((i64)idx < n_bases) qual_out[idx] = (u8)(e >> 32);
    }
}
void mdx_k_rescale_expand(const uint8_t *qual, uint8_t *qual_out, int64_t n_bases, const unsigned long long *patch,
                          const unsigned long long *n_patch, long long patch_cap, int patch_parts, hipStream_t s) {
    if (n_bases <= 0 || patch_parts <= 0) return;
    if (qual_out != qual) (void)hipMemcpyAsync(qual_out, qual, (size_t)n_bases, hipMemcpyDeviceToDevice, s);
    hipLaunchKernelGGL(rescale_expand_kernel, dim3(16, patch_parts), dim3(256), 0, s, qual_out, (const u64 *)patch, (const u64 *)n_patch, patch_cap, (i64)n_bases);
}

size_t mdx_k_rescale_part_bytes(int len5p, int len3p, int n_cu) {
    return (size_t)2 * n_cu * RS_BPC * (size_t)(752 + 2 * (1 + len5p + len3p) * 94) * 4;   // rows of both kernels
}
