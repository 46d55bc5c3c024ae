// gfx950 (MI355X / CDNA4) kernels of the damage-tabulation engine.
//
// Work decomposition (integer/byte histogram work — HBM/LDS bound, no MFMA):
//   * a wavefront (64 lanes) owns a tile of 64 consecutive records;
//   * phase 1, lane-per-record: coalesced SoA loads of the per-record columns, flag filter
//     (reader.py:121-132), CIGAR scan (clips, reference span, column count), fragment-length
//     update (statistics.py:117-126), soft-clip update (statistics.py:37-51), error checks;
//   * phase 2, wavefront-per-record: the record's scalars are broadcast with v_readlane and
//     the 64 lanes each own one *task* of the record (one left- or right-anchored alignment
//     column, or one flank base); bases and reference symbols are read with consecutive
//     addresses across lanes.  The loads of record j+1 are issued before record j is counted
//     (register double-buffering) so that the gather latency of the resident genome is hidden;
//   * the common outcome (read base == reference base, or an A/C/G/T flank base) is one
//     conflict-free ds_add_u32 into a task-indexed LDS table; everything else (substitutions,
//     indels, N, masked columns) takes a rare, divergent path into the MIS/CMP tables;
//   * at block end the LDS image is stored to a per-block slot and a second kernel sums the
//     slots into the u64 accumulators (no global atomics on the hot path).
// Counting is done in *reference orientation* (left-/right-anchored, no complementing); the
// strand step of main.py:200-205 (reverse-complement + flank swap) becomes a fixed permutation
// applied once by finalize_kernel.  The oracle (oracle/mdx_oracle.c) builds and reverses the
// strings literally instead, so the two share no derivation.
#include "mdx_internal.h"

#include <type_traits>

typedef uint8_t u8;
typedef int8_t i8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef u32 __attribute__((aligned(1))) u32_u;
typedef unsigned long long u64;
typedef long long i64;

#define MDX_BLOCK 768                   // 12 wavefronts; two blocks per CU share the 160 KiB LDS
#define EVQ_CAP 128                     // rare-event queue capacity per wavefront (16-byte events)
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
size_t mdx_k_lds_bytes(const MdxDims &d) { return (size_t)(d.w_total + 3) / 4 * 16 + (size_t)(MDX_BLOCK / 64) * EVQ_CAP * 16; }
int mdx_k_queue_off(const MdxDims &d) { return (int)((d.w_total + 3) / 4 * 4); }

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
// TC increment of the fast path: LDS byte address = base + ((r4 >> bit) & 3) * 1024 + imm.
// Hand-written (v_bfe_u32, v_lshl_add_u32, ds_add_u32): the compiler's own sequence is shift + and +
// add3 per byte.  The hidden ds_add only makes the compiler's lgkmcnt waits more conservative (LDS
// operations complete in order).
__device__ __forceinline__ void tc_bump(u32 r4, int bit, u32 base_bytes, int imm, u32 data) {
    u32 k, addr;
    asm("v_bfe_u32 %0, %1, %2, 2" : "=v"(k) : "v"(r4), "n"(bit));
    asm("v_lshl_add_u32 %0, %1, 10, %2" : "=v"(addr) : "v"(k), "v"(base_bytes));
    asm volatile("ds_add_u32 %0, %1 offset:%2" : : "v"(addr), "v"(data), "n"(imm) : "memory");
}
// The four bytes of a dword in one block (no filler s_nop between separate asm statements)
__device__ __forceinline__ void tc_bump4(u32 r4, u32 base_bytes, u32 d0, u32 d1, u32 d2, u32 d3) {
    u32 t0, t1, t2, t3;
    asm volatile(
        "v_bfe_u32 %0, %4, 1, 2\n\t"
        "v_bfe_u32 %1, %4, 9, 2\n\t"
        "v_bfe_u32 %2, %4, 17, 2\n\t"
        "v_bfe_u32 %3, %4, 25, 2\n\t"
        "v_lshl_add_u32 %0, %0, 10, %5\n\t"
        "v_lshl_add_u32 %1, %1, 10, %5\n\t"
        "v_lshl_add_u32 %2, %2, 10, %5\n\t"
        "v_lshl_add_u32 %3, %3, 10, %5\n\t"
        "ds_add_u32 %0, %6\n\t"
        "ds_add_u32 %1, %7 offset:256\n\t"
        "ds_add_u32 %2, %8 offset:512\n\t"
        "ds_add_u32 %3, %9 offset:768"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(r4), "v"(base_bytes), "v"(d0), "v"(d1), "v"(d2), "v"(d3)
        : "memory");
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

// record descriptor word w1 (one int per record, broadcast in phase 2)
#define D_REV 1
#define D_SIMPLE 2
#define D_HASQ 4
#define D_FULL 8        // plain-match record with every task present (nq >= L, both flanks complete)
#define D_NB_SHIFT 8    // nbefore, 8 bits
#define D_NA_SHIFT 16   // nafter, 8 bits

// (final column - 4) of substitution ref>read for classes A,C,T,G, 4 bits per [ref][read] entry
#define SUB_LUT 0x0ba0803961072540ull
//   A>: -,4,5,2   C>: 7,-,1,6   T>: 9,3,-,8   G>: 0,10,11,-   (entry index = ref * 4 + read)

// Rare path of a plain-match record column: (ch, rch) is not a plain match.  The read base goes to
// CMP (statistics.py:75-83); a substitution/indel goes to its MIS column only — the accompanying
// reference-base count of statistics.py:30 is derived by finalize_kernel (mis[r] = matches +
// sum of the r>x columns).
template <bool USE_LDS>
__device__ __forceinline__ void rare_column(u32 *lds, u64 *raw, int b_mis, int b_cmp, int L, int side,
                                            int p, u32 ch, int rch, bool masked) {
    const int s = classify_read(ch);
    const int sp = side * L + p;
    if (s < 4) bump<USE_LDS>(lds, raw, b_cmp + sp * 4 + s);
    if (!masked && s <= SYM_GAP) {
        const int r = classify_ref(rch);
        if (r <= SYM_GAP && r != s) {  // statistics.py:26-35
            int col;
            if (r < 4 && s < 4) col = 4 + (int)((SUB_LUT >> (4 * (r * 4 + s))) & 15ull);
            else col = c_col[r * 5 + s];
            bump<USE_LDS>(lds, raw, b_mis + sp * 25 + col);
        }
    }
}

// FAST: the dword-lane path for plain, complete records (MdxDims::fast_ok()); otherwise every
// record takes the generic CIGAR walk.
template <bool USE_LDS, bool MASK, bool FAST>
__global__ __launch_bounds__(MDX_BLOCK, 6) void tabulate_kernel(MdxTabArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const MdxDims d = a.dims;
    const int L = d.L, A = d.A;
    const int lane = threadIdx.x & 63;
    const int waves_per_block = MDX_BLOCK / 64;
    const i64 gwave = (i64)blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    const i64 nwaves = (i64)gridDim.x * waves_per_block;
    u64 *raw = a.raw;

    if (USE_LDS) {
        for (i64 i = threadIdx.x; i < d.w_total; i += MDX_BLOCK) lds[i] = 0;
        __syncthreads();
    }

    // per-lane constants of the dword fast path (see MdxDims): the lane's four bytes are
    //   reference: refB[rcoef * nq + r0]  with refB = ref + rbase - apad
    //   SEQ:       seq[scoef * nq + s0]   (flank lanes: a dummy, disabled by em = 0)
    // byte j is a task iff bit 8j of vm is set; (side, p) of byte j: p = pbase + pstep * j.
    int c_r0 = d.apad, c_rcoef = 0, c_s0 = 0, c_scoef = 0, c_side = 0, c_pbase = 0, c_pstep = 1;
    int c_kind = 4, c_m4 = 0;  // lane kind: 0 left columns, 1 right columns, 2 left flank, 3 right flank, 4 none
    u32 c_em = 0, c_vm = 0;
    bool c_read = false;
    if (FAST) {
        const int m0 = lane, m1 = lane - d.nl4, m2 = lane - 2 * d.nl4, m3 = lane - 2 * d.nl4 - d.nf4;
        if (m0 < d.nl4) {
            c_r0 = d.apad + 4 * m0; c_s0 = 4 * m0; c_em = ~0u; c_read = true; c_side = 0; c_pbase = 4 * m0; c_pstep = 1;
            c_kind = 0; c_m4 = 4 * m0;
            for (int j = 0; j < 4; j++) if (4 * m0 + j < L) c_vm |= 0xFFu << (8 * j);
        } else if (m1 < d.nl4) {
            c_rcoef = 1; c_r0 = d.apad - 4 - 4 * m1; c_scoef = 1; c_s0 = -4 - 4 * m1; c_em = ~0u; c_read = true;
            c_side = 1; c_pbase = 4 * m1 + 3; c_pstep = -1; c_kind = 1; c_m4 = 4 * m1;
            for (int j = 0; j < 4; j++) if (4 * m1 + 3 - j < L) c_vm |= 0xFFu << (8 * j);
        } else if (m2 < d.nf4) {
            c_r0 = d.apad - 4 * (m2 + 1); c_kind = 2; c_m4 = 4 * m2;
            for (int j = 0; j < 4; j++) if (4 * (m2 + 1) - j <= A) c_vm |= 0xFFu << (8 * j);
        } else if (m3 < d.nf4) {
            c_rcoef = 1; c_r0 = d.apad + 4 * m3; c_kind = 3; c_m4 = 4 * m3;
            for (int j = 0; j < 4; j++) if (4 * m3 + j + 1 <= A) c_vm |= 0xFFu << (8 * j);
        }
    }
    const u32 c_lane16 = (u32)lane << 4;                                           // lane field of an event word
    const u32 c_lane4 = ((u32)lane << 2) + __builtin_amdgcn_groupstaticsize();      // LDS byte address of word `lane`
    const u32 c_d0 = c_vm & 1u, c_d1 = (c_vm >> 8) & 1u, c_d2 = (c_vm >> 16) & 1u, c_d3 = (c_vm >> 24) & 1u;

    // Rare-event queue of the fast path (wave-private ring in the LDS, EVQ_CAP 16-byte events):
    // event = {read dword, reference dword, x (nonzero bytes = not a plain match),
    //          masked-quality flags [3:0] | lane [9:4] | reverse strand [10] | TC base of the record [26:11]}
    // Events are self-contained, so they survive tile changes and are drained in full passes of 64.
    uint4 *const queue = (uint4 *)(lds + a.queue_off) + (threadIdx.x >> 6) * EVQ_CAP;
    int qhead = 0, qcount = 0;

    // One pass: undo the optimistic TC increment of each byte that was not a plain match and, for read
    // bytes, count what the byte really is (rare_column) — lane-parallel over up to 64 events.
    auto drain_pass = [&]() {
        const bool ev_ok = lane < qcount;
        uint4 ev = make_uint4(0, 0, 0, 0);
        if (ev_ok) ev = queue[(qhead + lane) & (EVQ_CAP - 1)];
        const int n = qcount < 64 ? qcount : 64;
        qhead = (qhead + n) & (EVQ_CAP - 1);
        qcount -= n;
        if (ev_ok) {
            const int ln = (int)(ev.w >> 4) & 63;
            const int rev = (int)(ev.w >> 10) & 1;
            const int e_tcb = (int)(ev.w >> 11);                       // TC base incl. strand (word index)
            const int lb = e_tcb - d.off_tc() - rev * 1024;            // first word of the record's library
            const bool is_read = ln < 2 * d.nl4;
            const int side = ln >= d.nl4;
            const int m = ln - (side ? d.nl4 : 0);
            const int b_mis = lb + rev * 2 * L * 25, b_cmp = lb + d.off_cmp() + rev * 2 * L * 4;
            // usually exactly one byte of the dword differs: handle the lowest non-matching byte
            // with per-lane shifts (all lanes busy), repeat only while some lane has another
            u32 xr = ev.z;
            while (xr) {
                const int jb = (__ffs((int)xr) - 1) >> 3;
                const int sh = 8 * jb;
                xr &= ~(0xFFu << sh);
                const u32 rb = (ev.y >> sh) & 0xFFu;
                bump_n<USE_LDS>(lds, raw, e_tcb + 64 * jb + ln + (int)(((rb >> 1) & 3u) << 8), 0xFFFFFFFFu);  // -1
                if (is_read)
                    rare_column<USE_LDS>(lds, raw, b_mis, b_cmp, L, side, side ? 4 * m + 3 - jb : 4 * m + jb,
                                         (ev.x >> sh) & 0xFFu, (int)(i8)rb, MASK && ((ev.w >> jb) & 1u));
            }
        }
        // nothing LDS-returning may be pending when control rejoins the hot loop: otherwise the
        // compiler guards the loop's first instructions with s_waitcnt lgkmcnt(0), which also
        // waits for the previous record's ds_add_u32s on every iteration
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    };

    // each wavefront owns one contiguous range of records (balanced to +-1 record), walked in tiles of 64
    const i64 r_lo = a.n_reads * gwave / nwaves, r_hi = a.n_reads * (gwave + 1) / nwaves;
    for (i64 tbase = r_lo; tbase < r_hi; tbase += 64) {
        // ------------------------------------------------------------ phase 1: lane per record
        const i64 ri = tbase + lane;
        const bool valid = ri < r_hi;
        const u32 fl = valid ? (u32)a.flag[ri] : 0x4u;
        bool kept = (fl & 0xF04u) == 0;  // reader.py:121-132
        int w0 = 0, w1 = 0, nq = 0, libid = 0, n0 = 0, ncols = 0, nI = 0, cig_n = 0;
        u32 sq = 0, cig_o = 0;
        i64 rbase = 0;
        int lkey = -1;  // fragment-length key for the LDS histogram
        if (kept) {
            const int rev = (fl >> 4) & 1;
            libid = a.lib[ri];
            const int tid = a.tid[ri];
            const i64 pos = a.pos[ri];
            cig_o = a.cigar_off[ri];
            cig_n = (int)(a.cigar_off[ri + 1] - cig_o);
            const u32 so = a.seq_off[ri];
            const i64 lseq = (i64)a.seq_off[ri + 1] - (i64)so;
            bool bad = tid < 0 || tid >= a.n_contig || libid >= d.nlib || lseq <= 0 || pos < 0;
            const int lbase = bad ? 0 : libid * d.w_lib;

            // CIGAR scan: pysam query_alignment_start/_end, htslib bam_endpos, parse_cigar
            i64 qs = 0, rlen = 0, qcons = 0, tl = 0, sI = 0, sDN = 0;
            bool leading = true;
            for (int k = 0; k < cig_n; k++) {
                const u32 c = a.cigar[cig_o + k];
                const int op = c & 0xF;
                const i64 len = c >> 4;
                if (leading) {
                    if (op == 4) qs += len;
                    else if (op != 5) leading = false;
                }
                if (op == 0 || op == 7 || op == 8) { tl += len; rlen += len; qcons += len; }
                else if (op == 1) { tl += len; sI += len; qcons += len; }
                else if (op == 2) { tl += len; rlen += len; sDN += len; }
                else if (op == 3) { rlen += len; sDN += len; }
                else if (op == 4 && !bad) {
                    // statistics.py:37-51: left side iff no alignment column precedes the clip
                    const int side = tl == 0 ? 0 : 1;
                    const int m = len < L ? (int)len : L;
                    const int base = lbase + ((rev * 2 + side) * L) * 25 + COL_S;
                    for (int x = 0; x < m; x++) bump<USE_LDS>(lds, raw, base + x * 25);
                }
            }
            i64 qe = lseq;
            for (int k = cig_n - 1; k >= 1; k--) {
                const u32 c = a.cigar[cig_o + k];
                const int op = c & 0xF;
                if (op == 5) continue;
                if (op == 4) qe -= (i64)(c >> 4);
                else break;
            }
            const i64 nq64 = qe > qs ? qe - qs : 0;
            const i64 n064 = rlen ? rlen : 1;
            const i64 aend = pos + n064;
            i64 clen = 0;
            if (!bad) {
                const i64 c0 = a.contig_off[tid];
                clen = a.contig_off[tid + 1] - c0;
                rbase = c0 + pos;
            }
            // align.py:33 / main.py:180: fetch(start > end) raises once aend > contig length;
            // a CIGAR that disagrees with SEQ cannot come out of htslib
            bad = bad || cig_n == 0 || aend > clen || nq64 != qcons || tl > 0x3FFFFFFF ||
                  n064 + sI > 0x3FFFFFFF;
            if (bad) {
                flag_error(a.err, ri, ERR_BAD_READ);
                kept = false;
            } else {
                nq = (int)nq64; n0 = (int)n064; ncols = (int)tl; nI = (int)sI;
                sq = so + (u32)qs;
                const int nbefore = pos < A ? (int)pos : A;
                const int nafter = clen - aend < A ? (int)(clen - aend) : A;
                const bool simple = sI == 0 && sDN == 0 && rlen > 0 && nq < 32768;
                w1 = rev | (simple ? D_SIMPLE : 0) | ((nbefore & 0xFF) << D_NB_SHIFT) | ((nafter & 0xFF) << D_NA_SHIFT);
                if (simple && nq >= 4 * d.nl4 && nq >= L && nbefore == A && nafter == A) w1 |= D_FULL;
                if (MASK && a.qual != nullptr && a.qual[so] != 0xFF) w1 |= D_HASQ;
                w0 = (nq & 0xFFFF) | (libid << 16);
                // statistics.py:117-126
                int kind = -1;
                i64 flen = 0;
                if (fl & 0x1) {
                    if ((fl & 0x40) && (fl & 0x2)) {
                        kind = 0;
                        const i64 t = a.tlen[ri];
                        flen = t < 0 ? -t : t;
                    }
                } else {
                    kind = 1;
                    flen = n064;
                }
                if (kind >= 0) {
                    if (flen < d.lgd_lds) {
                        lkey = lbase + d.off_lgd() + (kind * 2 + rev) * d.lgd_lds + (int)flen;
                    } else if (flen < d.lgd_max) {
                        atomicAdd(&a.lgd_dense[(((i64)libid * 2 + kind) * 2 + rev) * d.lgd_max + flen], 1ull);
                    } else {
                        const u64 slot = atomicAdd(a.n_lgd_over, 1ull);
                        if ((i64)slot < a.lgd_over_cap) {
                            a.lgd_over[4 * slot + 0] = libid;
                            a.lgd_over[4 * slot + 1] = kind;
                            a.lgd_over[4 * slot + 2] = rev;
                            a.lgd_over[4 * slot + 3] = flen;
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

        // ------------------------------------------------------------ phase 2a: plain records
        // One lane = one dword (four consecutive bytes) of the record; one wavefront step = one record.
        // Bytes that are not plain matches are not handled here: they are appended as events to a
        // wave-private LDS queue and counted later 64 at a time (drain_pass), so the divergent
        // classification code runs once per 64 events instead of once per record.
        const int rb_lo = (int)(rbase & 0xFFFFFFFFll), rb_hi = (int)(rbase >> 32);
        u64 todo_g = todo_all;
        if (FAST) {
            u64 todo = __ballot(kept && (w1 & D_FULL));
            todo_g = todo_all & ~todo;
            // offset of the record's reference window and its TC base, per lane (phase-1 layout)
            // offset from the start of the guard band (>= 0; not an absolute address: keeps the loads global)
            const i64 refw = rbase - d.apad + 256;
            const int rf_lo = (int)(refw & 0xFFFFFFFFll), rf_hi = (int)(refw >> 32);
            const int tcb = libid * d.w_lib + d.off_tc() + (w1 & D_REV) * 1024;
            // nq (15 bits) | reverse strand | TC base incl. strand (tcb < 40960 words: 160 KiB of LDS)
            const int wq = (nq & 0x7FFF) | ((w1 & D_REV) << 15) | (tcb << 16);
            // software pipeline: four records in flight, each in its own register set (no register
            // rotation: a copy of an in-flight destination would wait for its load)
            struct Stage { u32 s4, r4, q4; int tcb, w1; u32 ev; bool valid; };
            u64 pending = todo;

            // fill() always issues its two loads (past the last record it re-reads the previous one), so
            // the number of loads in flight is static and the waits before count() are counted ones
            int last_j = todo ? __ffsll((long long)todo) - 1 : 0;
            auto fill = [&](Stage &st) {
                st.valid = pending != 0;
                const int j = st.valid ? __ffsll((long long)pending) - 1 : last_j;
                pending &= pending - 1;
                last_j = j;
                const int s_wq = rl(wq, j);
                const int s_nq = s_wq & 0x7FFF;
                st.tcb = (int)((u32)s_wq >> 16);
                st.ev = ((u32)s_wq >> 15) << 10;  // record part of an event's 4th word
                if (MASK) st.w1 = rl(w1, j);
                u64 roff = (u32)rl(rf_lo, j);
                if (!a.ref32) roff |= (u64)(u32)rl(rf_hi, j) << 32;  // genomes of 4 Gbases and more
                const u8 *__restrict__ refB = (a.ref - 256) + roff;
                const u32 s_sq = (u32)rl((int)sq, j);
                const u8 *__restrict__ seqP = a.seq + s_sq;
                const u32 ro = (u32)(c_rcoef * s_nq + c_r0);
                const u32 so = (u32)(c_scoef * s_nq + c_s0);
                st.r4 = *(const u32_u *)(refB + ro);
                st.s4 = *(const u32_u *)(seqP + so);
                if (MASK) st.q4 = *(const u32_u *)(a.qual + s_sq + so);
            };

            auto count = [&](const Stage &st) {
                const u32 s4_c = st.s4, r4_c = st.r4;
                // x: per byte, zero iff the byte is a plain match (read == reference, reference is A/C/G/T);
                // flank lanes only test the reference byte; bytes that are not tasks are forced to zero
                u32 x = ((s4_c ^ r4_c) & c_em) | (r4_c & 0x80808080u);
                u32 mq = 0;
                if (MASK) {
                    // bytes whose quality is below --min-basequal (align.py:65-71): bit 7 of the byte
                    const u32 minq4 = (st.w1 & D_HASQ) ? (u32)a.minqual * 0x01010101u : 0u;
                    mq = ~((st.q4 | 0x80808080u) - minq4) & 0x80808080u & c_em;
                    x |= mq;
                }
                x &= c_vm;
                // optimistic: count every task byte as a plain match (the base class of the reference
                // byte, (ascii >> 1) & 3, selects the 1 KiB plane of TC) ...
                const u32 base_b = ((u32)st.tcb << 2) + c_lane4;
                tc_bump4(r4_c, base_b, c_d0, c_d1, c_d2, c_d3);
                // ... and queue the lanes holding a byte that is not one (drain_pass corrects them)
                const u64 mm = __ballot(x != 0);
                if (mm) {
                    if (x != 0) {
                        const int slot = qhead + qcount + (int)__builtin_amdgcn_mbcnt_hi((u32)(mm >> 32), __builtin_amdgcn_mbcnt_lo((u32)mm, 0u));
                        u32 w = st.ev | c_lane16;
                        if (MASK) w |= ((mq >> 7) & 1u) | ((mq >> 14) & 2u) | ((mq >> 21) & 4u) | ((mq >> 28) & 8u);
                        queue[slot & (EVQ_CAP - 1)] = make_uint4(s4_c, r4_c, x, w);
                    }
                    qcount += __popcll(mm);
                    if (qcount >= 64) drain_pass();  // at most 63 + 64 events are ever queued
                }
            };

            Stage st0{}, st1{}, st2{}, st3{};
            if (todo) {
                fill(st0); fill(st1); fill(st2); fill(st3);
                // single-exit loop, eight loads in flight at every point of it (stages past the last record
                // re-read it and are skipped by their valid flag): lets the compiler count its vmcnt waits
                do {
                    if (st0.valid) count(st0);
                    fill(st0);
                    if (st1.valid) count(st1);
                    fill(st1);
                    if (st2.valid) count(st2);
                    fill(st2);
                    if (st3.valid) count(st3);
                    fill(st3);
                } while (st0.valid);
            }

            // -------- plain records with missing tasks (shorter than the window, or at a contig edge):
            // same dword layout, the byte-validity mask is computed per record instead of per lane
            u64 todo_p = __ballot(kept && (w1 & D_SIMPLE) && !(w1 & D_FULL));
            todo_g &= ~todo_p;
            if (todo_p) {
                struct PStage { u32 s4, r4, q4; int nq, tcb, w1; u32 ev; bool valid; };
                u64 pend_p = todo_p;
                int last_p = __ffsll((long long)todo_p) - 1;
                auto fill_p = [&](PStage &st) {
                    st.valid = pend_p != 0;
                    const int j = st.valid ? __ffsll((long long)pend_p) - 1 : last_p;
                    pend_p &= pend_p - 1;
                    last_p = j;
                    const int s_wq = rl(wq, j);
                    const int s_nq = s_wq & 0x7FFF;
                    st.nq = s_nq;
                    st.tcb = (int)((u32)s_wq >> 16);
                    st.ev = ((u32)s_wq >> 15) << 10;
                    st.w1 = rl(w1, j);
                    u64 roff = (u32)rl(rf_lo, j);
                    if (!a.ref32) roff |= (u64)(u32)rl(rf_hi, j) << 32;
                    // a right-column dword of a short record may start before the record: its window offset
                    // is negative (down to -(4 nl4 - 1), still inside the 256-byte guard band), so bias it
                    const u8 *__restrict__ refB = (a.ref - 512) + roff;
                    const u32 s_sq = (u32)rl((int)sq, j);
                    const u8 *__restrict__ seqP = a.seq + s_sq;
                    const u32 ro = (u32)(c_rcoef * s_nq + c_r0 + 256);
                    int so = c_scoef * s_nq + c_s0;                     // may leave the record: clamp to 0
                    if (so < 0 || so >= s_nq) so = 0;
                    st.r4 = *(const u32_u *)(refB + ro);
                    st.s4 = *(const u32_u *)(seqP + (u32)so);
                    if (MASK) st.q4 = *(const u32_u *)(a.qual + s_sq + (u32)so);
                };
                auto count_p = [&](const PStage &st) {
                    const int nb = (st.w1 >> D_NB_SHIFT) & 0xFF, na = (st.w1 >> D_NA_SHIFT) & 0xFF;
                    // valid bytes of this lane: nv of them, at the low end (left columns, right flank:
                    // increasing position) or at the high end (right columns, left flank)
                    const int avail = (c_kind < 2 ? st.nq : (c_kind == 2 ? nb : na)) - c_m4;
                    const int nv = avail < 0 ? 0 : (avail > 4 ? 4 : avail);
                    const bool top = c_kind == 1 || c_kind == 2;
                    const u32 ones = 0xFFFFFFFFu;
                    const u32 dyn = nv == 0 ? 0u : (top ? ones << (32 - 8 * nv) : ones >> (32 - 8 * nv));
                    const u32 vm = c_vm & dyn;
                    // a right-column dword that starts before the record was loaded at offset 0: shift it
                    u32 s4_c = st.s4, q4_c = st.q4;
                    const int before = c_m4 + 4 - st.nq;  // bytes of the dword in front of the record
                    if (c_kind == 1 && before > 0 && before < 4) { s4_c <<= 8 * before; q4_c <<= 8 * before; }
                    const u32 r4_c = st.r4;
                    u32 x = ((s4_c ^ r4_c) & c_em) | (r4_c & 0x80808080u);
                    u32 mq = 0;
                    if (MASK) {
                        const u32 minq4 = (st.w1 & D_HASQ) ? (u32)a.minqual * 0x01010101u : 0u;
                        mq = ~((q4_c | 0x80808080u) - minq4) & 0x80808080u & c_em;
                        x |= mq;
                    }
                    x &= vm;
                    const u32 base_b = ((u32)st.tcb << 2) + c_lane4;
                    tc_bump(r4_c, 1, base_b, 0, vm & 1u);
                    tc_bump(r4_c, 9, base_b, 256, (vm >> 8) & 1u);
                    tc_bump(r4_c, 17, base_b, 512, (vm >> 16) & 1u);
                    tc_bump(r4_c, 25, base_b, 768, (vm >> 24) & 1u);
                    const u64 mm = __ballot(x != 0);
                    if (mm) {
                        if (x != 0) {
                            const int slot = qhead + qcount + (int)__builtin_amdgcn_mbcnt_hi((u32)(mm >> 32), __builtin_amdgcn_mbcnt_lo((u32)mm, 0u));
                            u32 w = st.ev | c_lane16;
                            if (MASK) w |= ((mq >> 7) & 1u) | ((mq >> 14) & 2u) | ((mq >> 21) & 4u) | ((mq >> 28) & 8u);
                            queue[slot & (EVQ_CAP - 1)] = make_uint4(s4_c, r4_c, x, w);
                        }
                        qcount += __popcll(mm);
                        if (qcount >= 64) drain_pass();
                    }
                };
                PStage p0{}, p1{};
                fill_p(p0); fill_p(p1);
                do {
                    if (p0.valid) count_p(p0);
                    fill_p(p0);
                    if (p1.valid) count_p(p1);
                    fill_p(p1);
                } while (p0.valid);
            }
        }

        // ------------------------------------------------------------ phase 2b: gapped records
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
            const int b_mis = lb + rev * 2 * L * 25;
            const int b_cmp = lb + d.off_cmp() + rev * 2 * L * 4;
            const int b_tc = lb + d.off_tc() + rev * 4 * d.t_pad;
            const i8 *__restrict__ rp = (const i8 *)a.ref + s_rbase;
            const u8 *__restrict__ sp = a.seq + s_sq;
            const u8 *__restrict__ qp = MASK ? a.qual + s_sq : nullptr;
            // flank lengths (not from the packed descriptor: A may exceed 255 here)
            int s_nb, s_na;
            {
                const i64 pos = a.pos[tbase + j];
                const int tid = a.tid[tbase + j];
                const i64 clen = a.contig_off[tid + 1] - a.contig_off[tid];
                s_nb = pos < A ? (int)pos : A;
                s_na = clen - (pos + s_n0) < A ? (int)(clen - (pos + s_n0)) : A;
            }

            // misincorporation pairs, each string indexed from its own end (main.py:210-212)
            int Lm = s_ncols < s_nrg ? s_ncols : s_nrg;
            if (Lm > L) Lm = L;
            for (int t = lane; t < 2 * Lm; t += 64) {
                const int side = t >= Lm;
                const int i = side ? t - Lm : t;
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
                int s = qi < 0 ? SYM_GAP : classify_read(sp[qi]);
                int r = rix < 0 ? SYM_GAP : classify_ref(rp[rix]);
                if (hasq) {
                    const bool ms = qi >= 0 && (int)qp[qi] < a.minqual;
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
                            mr = qj >= 0 && (int)qp[qj] < a.minqual;
                        }
                    }
                    if (ms) s = SYM_OTHER;
                    if (mr) r = SYM_OTHER;
                }
                if (s <= SYM_GAP && r <= SYM_GAP) {
                    // the reference-base count of a mismatching column is derived at finalisation
                    const int row = b_mis + (side * L + i) * 25;
                    if (r != s) bump<USE_LDS>(lds, raw, row + c_col[r * 5 + s]);
                    else if (r != SYM_GAP) bump<USE_LDS>(lds, raw, row + r);
                }
            }
            // read composition on the ungapped, unmasked query (statistics.py:75-83)
            const int Lq = s_nq < L ? s_nq : L;
            for (int t = lane; t < 2 * Lq; t += 64) {
                const int side = t >= Lq;
                const int k0 = side ? t - Lq : t;
                const int s = classify_read(sp[side ? s_nq - 1 - k0 : k0]);
                if (s < 4) bump<USE_LDS>(lds, raw, b_cmp + (side * L + k0) * 4 + s);
            }
            // flanks (statistics.py:85-93) go to their task slots
            for (int t = lane; t < 2 * A; t += 64) {
                const int side = t >= A;
                const int dist = (side ? t - A : t) + 1;
                if (dist <= (side ? s_na : s_nb)) {
                    const int r = side ? rp[s_n0 - 1 + dist] : rp[-dist];
                    if (r >= 0) bump<USE_LDS>(lds, raw, b_tc + ((r >> 1) & 3) * d.t_pad + (side ? d.tau_rflank(dist) : d.tau_lflank(dist)));
                }
            }
        }
    }

    if (FAST) {
        while (qcount > 0) drain_pass();
    }
    if (USE_LDS) {
        __syncthreads();
        u32 *out = a.partials + (i64)blockIdx.x * d.w_total;
        for (i64 i = threadIdx.x; i < d.w_total; i += MDX_BLOCK) out[i] = lds[i];
    }
}

template <bool MASK, bool FAST>
static hipError_t prep_one(size_t lds_bytes) {
    return hipFuncSetAttribute((const void *)tabulate_kernel<true, MASK, FAST>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
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
    if (a.dims.fast_ok()) {
        if (mask) launch_one<true, true, true>(a, grid, lds_bytes, s);
        else launch_one<true, false, true>(a, grid, lds_bytes, s);
    } else {
        if (mask) launch_one<true, true, false>(a, grid, lds_bytes, s);
        else launch_one<true, false, false>(a, grid, lds_bytes, s);
    }
}

// raw[w] += sum over block slots; blocks are split into `parts` groups to expose parallelism
__global__ void reduce_partials_kernel(const u32 *__restrict__ partials, u64 *__restrict__ raw, i64 w_total,
                                       int grid, int parts) {
    const i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= w_total) return;
    const int part = blockIdx.y;
    const int b0 = (int)((i64)grid * part / parts), b1 = (int)((i64)grid * (part + 1) / parts);
    u64 acc = 0;
    for (int b = b0; b < b1; b++) acc += partials[(i64)b * w_total + w];
    if (acc) atomicAdd(&raw[w], acc);
}

void mdx_k_reduce_partials(const uint32_t *partials, unsigned long long *raw, int64_t w_total, int grid,
                           hipStream_t s) {
    const int threads = 256;
    const int blocks = (int)((w_total + threads - 1) / threads);
    int parts = grid < 32 ? grid : 32;
    if (parts < 1) parts = 1;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(blocks, parts), dim3(threads), 0, s, partials, raw,
                       (i64)w_total, grid, parts);
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
                const i64 row = lb + ((strand * 2 + side) * L + p) * 25;
                // matches (gapped records / plain records) + every column whose reference symbol is k
                v = raw[row + k];
                if (d.nl4 > 0) v += raw[lb + d.off_tc() + (strand * 4 + k) * d.t_pad + (side ? d.tau_right(p) : d.tau_left(p))];
                for (int x = 0; x < 4; x++) v += raw[row + c_refcols[k * 4 + x]];
            } else {
                const int rc = strand ? c_comp_col[col] : col;
                v = raw[lb + ((strand * 2 + side) * L + p) * 25 + rc];
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
                if (d.nl4 > 0) v += raw[tc + (side ? d.tau_right(slot) : d.tau_left(slot))];
            }
            else v = raw[tc + (side ? d.tau_rflank(dist) : d.tau_lflank(dist))];
        } else if (i < n_mis + n_comp + n_lgd) {
            i64 x = i - n_mis - n_comp;
            const int len = x % d.lgd_max; x /= d.lgd_max;
            const int strand = x % 2; x /= 2;
            const int kind = x % 2; x /= 2;
            v = lgd_dense[i - n_mis - n_comp];
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
// Quality rescaling (mapdamage/rescale.py:195-365; BASELINE config[4]).  One wavefront per record;
// one lane per query base, walked in the read's own 5'->3' order so that the MR sum is
// accumulated in the reference's column order (fp64, bit-exact).  The new quality is a byte lookup
// LUT[sub][position key][old quality] prepared on the host with the reference's floating-point
// expressions (mapdamage_amd/rescale.py).
__global__ __launch_bounds__(256) void rescale_kernel(MdxRescaleArgs a) {
    const int lane = threadIdx.x & 63;
    const i64 gwave = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const i64 nwaves = ((i64)gridDim.x * blockDim.x) >> 6;
    const int npos = 1 + a.len5p + a.len3p;
    for (i64 ri = gwave; ri < a.n_reads; ri += nwaves) {
        const u32 fl = a.flag[ri];
        const u32 so = a.seq_off[ri];
        const int lseq = (int)(a.seq_off[ri + 1] - so);
        const u32 co = a.cigar_off[ri];
        const int cn = (int)(a.cigar_off[ri + 1] - co);
        const u8 *__restrict__ qin = a.qual + so;
        u8 *__restrict__ qout = a.qual_out + so;
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
            for (int b = lane; b < lseq; b += 64) qout[b] = qin[b];
            continue;
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
            for (int b = lane; b < lseq; b += 64) qout[b] = qin[b];
            continue;
        }
        // soft-clipped qualities are kept
        for (int b = lane; b < qs; b += 64) qout[b] = qin[b];
        for (int b = qs + nq + lane; b < lseq; b += 64) qout[b] = qin[b];

        const i8 *__restrict__ rp = (const i8 *)a.ref + rbase;
        const u8 *__restrict__ sp = a.seq + so + qs;
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
                const int jr = rev ? nrg - ncols + js : js;
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
                const int rch = rix < 0 ? -1 : (int)rp[rix];
                const u32 ch = sp[qi];
                const u32 q = qin[qs + qi];
                // read-orientation pair (T,C) -> C>T ; (A,G) -> G>A; complemented on the reverse strand
                int sub = -1;
                if (!rev) { if (ch == 'T' && rch == 'C') sub = 0; else if (ch == 'A' && rch == 'G') sub = 1; }
                else { if (ch == 'A' && rch == 'G') sub = 0; else if (ch == 'T' && rch == 'C') sub = 1; }
                u32 newq = q;
                if (sub >= 0) {
                    // _corr_this_base, rescale.py:49-79
                    int p = oq + 1;
                    const int back = p - nq - 1;
                    if (!forward_only && p >= -back) p = back;
                    const int key = p > 0 ? (p <= a.len5p ? p : 0) : (-p <= a.len3p ? a.len5p - p : 0);
                    term = a.term[sub * npos + key];
                    if (q <= 93) newq = a.lut[(sub * npos + key) * 94 + q];
                }
                qout[qs + qi] = (u8)newq;
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
    }
}

void mdx_k_rescale(const MdxRescaleArgs &a, int grid, hipStream_t s) {
    if (a.n_reads <= 0) return;
    hipLaunchKernelGGL(rescale_kernel, dim3(grid), dim3(256), 0, s, a);
}
