// GPU-side BAM decode: BGZF blocks inflated and BAM records unpacked into the batch columns without
// leaving HBM — the device counterpart of mdx_bamio.cpp, which is the counterpart of pysam.AlignmentFile behind
// mapdamage/reader.py:20-46.  Host orchestration: mdx_gbam_* in mdx_bamio.cpp.
//
//   gbam_inflate_kernel   one wavefront per BGZF block (mdx_inflate.h): the last 4 KiB of the output and the tables in
//                         the LDS (~10 KB: sixteen blocks per CU), output written to HBM 2 KiB at a time, 16 bytes per
//                         lane; matches that reach further back read the output in HBM
//   gbam_crc_kernel       one wavefront per BGZF block: CRC32 of the inflated bytes against the gzip trailer
//   gbam_scan_kernel      one lane per BGZF block: the chain of block_size fields of the records that start in it, from a
//                         guessed first record (any layout: records may straddle blocks; the host accepts a guess only
//                         if the chain before it lands on it), counts records, CIGAR operations, bases
//   gbam_offsets_kernel   one lane per BGZF block again: record offsets and the cigar_off / seq_off columns
//   gbam_unpack_kernel    eight lanes per record: fixed fields, CIGAR, 4-bit bases -> ASCII, qualities, RG:Z -> library
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mdx_crc32.h"
#include "mdx_deflate.h"
#include "mdx_inflate.h"
#include "mdx_internal.h"

namespace {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

__device__ __forceinline__ u32 g16(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8); }
__device__ __forceinline__ u32 g32(const u8 *p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }

#ifndef MDX_INFL_WPS
#define MDX_INFL_WPS 4
#endif
__global__ __launch_bounds__(64, MDX_INFL_WPS) void gbam_inflate_kernel(const u8 *__restrict__ comp, const uint4 *__restrict__ blk,
                                                           u8 *__restrict__ unc, int *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    u8 *const win = lds;                                                       // the window (mdx_inflate::RING bytes)
    mdx_inflate::Tables &t = *(mdx_inflate::Tables *)(lds + mdx_inflate::RING);
    const uint4 b = blk[blockIdx.x];                                           // in_off, in_size, out_off, out_size
    int r = 0;
    if (b.w > 0) {
        // (the output leaves the window for HBM half a ring at a time; b.w <= 65536 was checked on the host)
        r = mdx_inflate::inflate_block(comp + b.x, b.y, win, unc + b.z, b.w, t);
        if (r >= 0 && (u32)r != b.w) r = -4;                                   // ISIZE disagrees
    }
    if (threadIdx.x == 0) status[blockIdx.x] = r;
}

// The CRC32 of every block's inflated bytes against the one in its gzip trailer (htslib, behind the reference's pysam,
// refuses a block that fails it): one wavefront per block, 1 KiB per lane, four bytes per step (tables in the LDS), the 64
// partial values joined by lane 0 with the "append n zero bytes" matrices.  bad = the lowest failing block.
__global__ __launch_bounds__(64) void gbam_crc_kernel(const u8 *__restrict__ unc, const uint4 *__restrict__ blk,
                                                       const u32 *__restrict__ want, const mdx_crc32::Tables *__restrict__ tb,
                                                       int *__restrict__ bad) {
    __shared__ u32 tab[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) tab[i] = (&tb->tab[0][0])[i];
    __syncthreads();
    const uint4 e = blk[blockIdx.x];
    const u32 n = e.w;
    const u32 lo = 1024u * (u32)lane;
    const u32 m = lo < n ? (n - lo < 1024u ? n - lo : 1024u) : 0u;
    const u32 mine = mdx_crc32::crc_bytes(tab, unc + e.z + lo, m);
    // (lane 0 alone: 63 matrix products of 32 steps each)
    u32 total = (u32)__builtin_amdgcn_readlane((int)mine, 0);
    for (int i = 1; i < 64; i++) {
        const u32 li = 1024u * (u32)i;
        if (li >= n) break;
        const u32 mi = n - li < 1024u ? n - li : 1024u;
        total = mdx_crc32::shift(tb->mat, total, mi) ^ (u32)__builtin_amdgcn_readlane((int)mine, i);
    }
    if (lane == 0 && total != want[blockIdx.x]) atomicMin(bad, (int)blockIdx.x);
}

// Does data[o, total) look like the start of a BAM record?  The fields a writer cannot choose freely: block_size against
// the sizes it must hold, reference ids against the header's dictionary, a read name that ends in NUL where l_read_name
// says.  Returns the record's block_size (>= 32), 0 when the bytes are not a record, and 1 when the record's fixed part
// does not lie inside the data (nothing can be said).
__device__ __forceinline__ u32 gbam_plausible(const u8 *__restrict__ unc, u32 o, u32 total, int n_ref) {
    if (o + 36u > total) return 1u;
    const u8 *r = unc + o + 4;
    const u32 bs = g32(unc + o);
    if (bs < 32u || bs > 0x10000000u) return 0u;
    const int tid = (int)g32(r), mtid = (int)g32(r + 20);
    const u32 l_name = r[8], n_c = g16(r + 12), l_seq = g32(r + 16);
    if (tid < -1 || tid >= n_ref || mtid < -1 || mtid >= n_ref || l_name == 0u || l_seq > 0x7FFFFFFFu) return 0u;
    if ((int)g32(r + 4) < -1 || (int)g32(r + 24) < -1) return 0u;
    if ((unsigned long long)32u + l_name + 4ull * n_c + ((unsigned long long)l_seq + 1u) / 2u + l_seq > bs) return 0u;
    const u32 nul = o + 4u + 32u + l_name - 1u;
    if (nul < total && unc[nul] != 0u) return 0u;
    return bs;
}

// One lane per BGZF block (segment) of the slab: the chain of records that starts in it.  The inflated blocks lie back
// to back in `unc`, so a record may straddle any number of them (htsjdk, sambamba and biobambam fill their blocks
// whatever the record boundaries; htslib starts every block at a record: bgzf_flush_try in bam_write1).  Where the
// first record of a segment starts is known for the slab's first segment only (`start0`; forced[b] >= 0 the same for
// segment b: a repair pass of the host).  Every other lane guesses: the first offset of its segment at which three
// plausible records follow one another (for an htslib file the segment's first byte).  The chain is followed from there
// to the first record that starts at or behind the segment's end.  info[b] = (first record, landing offset, status, 0)
// with status 0 = fine, 1 = no record starts in the segment, 2 = the chain ends in a record that is not complete in
// `unc` (landing = its start), 3 = a record whose CIGAR field is the placeholder of a CG tag, -1 = a record that cannot be
// one, -2 = the block did not inflate;
// cnt[b] = (records, CIGAR operations, bases, 0) of the chain.  The host accepts a guess only if the chain before it lands
// on it (mdx_gbam_next) — the result is exact whatever the guesses were.
__global__ void gbam_scan_kernel(const u8 *__restrict__ unc, const uint4 *__restrict__ blk, const int *__restrict__ inflated,
                                 int n_blocks, int first, const int *__restrict__ forced, u32 start0, u32 total, int n_ref,
                                 uint4 *__restrict__ info, uint4 *__restrict__ cnt) {
    const int b = first + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (b >= n_blocks) return;
    const uint4 e = blk[b];
    const u32 lo = e.z, hi = e.z + e.w;
    if (inflated[b] < 0) { info[b] = make_uint4(lo, lo, (u32)-2, 0); cnt[b] = make_uint4(0, 0, 0, 0); return; }
    u32 f = 0xFFFFFFFFu;
    if (forced && forced[b] >= 0) f = (u32)forced[b];
    else if (b == 0 && start0 != 0xFFFFFFFFu) f = start0;       // (0xFFFFFFFF: unknown — the slab behind a skipped one)
    else {
        for (u32 o = lo; o < hi && f == 0xFFFFFFFFu; o++) {
            u32 at = o;
            bool ok = true;
            for (int k = 0; k < 3 && ok; k++) {
                const u32 bs = gbam_plausible(unc, at, total, n_ref);
                if (bs == 0u) ok = false;
                else if (bs == 1u) break;              // the data ends: as far as can be seen, records
                else at += 4u + bs;
                if (at > total) break;
            }
            if (ok) f = o;
        }
    }
    u32 n_rec = 0, n_cig = 0, n_seq = 0;
    if (f == 0xFFFFFFFFu) { info[b] = make_uint4(hi, hi, 1u, 0); cnt[b] = make_uint4(0, 0, 0, 0); return; }
    u32 off = f, status = 0;
    while (off < hi) {
        if (off + 36u > total) { status = 2u; break; }
        const u32 bs = g32(unc + off);
        const u8 *r = unc + off + 4;
        const u32 l_name = r[8], n_c = g16(r + 12), l_seq = g32(r + 16);
        if (bs < 32u || bs > 0x7FFFFFF0u || l_seq > 0x7FFFFFFFu ||
            (unsigned long long)32u + l_name + 4ull * n_c + ((unsigned long long)l_seq + 1u) / 2u + l_seq > bs) { status = (u32)-1; break; }
        if ((unsigned long long)off + 4u + bs > total) { status = 2u; break; }
        // (a CIGAR of more than 65 535 operations lives in the record's CG tag, the field holds the placeholder <l_seq>S<n>N:
        // the device path does not take such a file — status 3, MDX_ERR_UNSUPPORTED, the host decoder puts the operations back)
        if (n_c == 2u && g32(r + 32 + l_name) == ((l_seq << 4) | 4u) && (g32(r + 36 + l_name) & 15u) == 3u) { status = 3u; break; }
        n_rec++; n_cig += n_c; n_seq += l_seq;
        off += 4u + bs;
    }
    info[b] = make_uint4(f, off, status, 0);
    cnt[b] = make_uint4(n_rec, n_cig, n_seq, 0);
}

// rec_off[r] = offset of record r's first field (behind block_size) in `unc`; cigar_off / seq_off: the columns.
// pre[b] = (records, CIGAR operations, bases in front of segment b's chain, first record of the chain), cnt[b].x = the
// records of the chain (0: the segment is skipped) — both from the host, which has checked the chains against each other
__global__ void gbam_offsets_kernel(const u8 *__restrict__ unc, const uint4 *__restrict__ pre,
                                    const uint4 *__restrict__ cnt, int n_blocks, u32 *__restrict__ rec_off,
                                    u32 *__restrict__ cigar_off, u32 *__restrict__ seq_off, u32 n_rec, u32 n_cig, u32 n_seq) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const uint4 s = pre[b], c = cnt[b];
    u32 off = s.w, r = s.x, co = s.y, so = s.z;
    for (u32 k = 0; k < c.x; k++) {
        const u32 bs = g32(unc + off);
        const u8 *q = unc + off + 4;
        rec_off[r] = off + 4u;
        cigar_off[r] = co; seq_off[r] = so;
        co += g16(q + 12); so += g32(q + 16);
        r++;
        off += 4u + bs;
    }
    if (b == 0) { cigar_off[n_rec] = n_cig; seq_off[n_rec] = n_seq; }     // the columns' closing entries
}

// eight lanes per record: the fixed fields and the read group by the first of them, the CIGAR, the bases (four per
// step: two bytes in, one unaligned dword out) and the qualities (four per step) spread over all eight
__global__ void gbam_unpack_kernel(const u8 *__restrict__ unc, const u32 *__restrict__ rec_off, u32 n_rec, MdxGbamCols c) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 r = t >> 3, j = t & 7u;
    if (r >= n_rec) return;
    typedef u32 u32u __attribute__((aligned(1)));
    const u8 *__restrict__ p = unc + rec_off[r];
    const u32 bs = g32(p - 4);
    const u32 l_name = p[8], n_cig = g16(p + 12), l_seq = g32(p + 16);
    if (j == 0) {
        c.flag[r] = (uint16_t)(g16(p + 14) & 0x3FFFu);     // bits 14 and 15 are the hints MDX_FLAG_HAS_QUAL / _QUAL_ABOVE_MIN, never the file's
        c.tid[r] = (int32_t)g32(p); c.pos[r] = (int32_t)g32(p + 4);
        c.tlen[r] = (int32_t)g32(p + 28);
        if (c.mtid) { c.mtid[r] = (int32_t)g32(p + 20); c.mpos[r] = (int32_t)g32(p + 24); }
    }
    const u8 *q = p + 32 + l_name;
    u32 *__restrict__ cg = c.cigar + c.cigar_off[r];
    for (u32 k = j; k < n_cig; k += 8u) cg[k] = g32(q + 4 * k);
    q += 4u * n_cig;
    // bases: two per byte, "=ACMGRSVTWYHKDBN" — the 16 letters as two 64-bit constants (registers, no table in memory)
    const u32 so = c.seq_off[r];
    u8 *__restrict__ s = c.seq + so;
    const u64 lo8 = ((u64)'=') | ((u64)'A' << 8) | ((u64)'C' << 16) | ((u64)'M' << 24) | ((u64)'G' << 32) | ((u64)'R' << 40) | ((u64)'S' << 48) | ((u64)'V' << 56);
    const u64 hi8 = ((u64)'T') | ((u64)'W' << 8) | ((u64)'Y' << 16) | ((u64)'H' << 24) | ((u64)'K' << 32) | ((u64)'D' << 40) | ((u64)'B' << 48) | ((u64)'N' << 56);
    auto letter = [&](const u32 nib) -> u32 { return (u32)(((nib < 8u ? lo8 : hi8) >> (8u * (nib & 7u))) & 0xFFu); };
    const u32 n4 = l_seq >> 2;
    if (c.seq_packed) {
        // MDX_SEQ_4BIT (include/mdx.h): the nibbles stay nibbles — recoded (A C G T = 1 2 4 8 in BAM become 1 2 8 4: bit k =
        // symbol class k of the tabulation kernel; every other code, '=' and the ambiguity letters, becomes 0: the
        // reference counts a read symbol only when it is one of "ACGT", statistics.py:27, 101), low nibble first, the
        // record's first base at nibble seq_off[r] of the column whatever its parity.  Eight bases per step: four bytes
        // in, one dword out, OR-ed into the (zeroed) column across its dword boundary.
        u32 *__restrict__ d32 = (u32 *)c.seq;
        const u64 lut = 0x0000000400080210ull;      // code of BAM nibble n at bits [4 n, 4 n + 4)
        auto code = [&](const u32 nib) -> u32 { return (u32)(lut >> (4u * nib)) & 15u; };
        const u32 n8 = (l_seq + 7u) >> 3;
        // (--min-basequal, MDX_SEQ_4BITQ: a base whose quality is below the threshold goes into the column as the complement
        // of its code — align.py:65-71 masks exactly those columns; 0xFF, no qualities, is not below any threshold)
        const u8 *qq = q + (l_seq + 1u) / 2u;
        const u32 m4 = (u32)c.minqual * 0x01010101u;
        for (u32 k = j; k < n8; k += 8u) {
            const u32 nb = l_seq - 8u * k < 8u ? l_seq - 8u * k : 8u;       // bases of this step
            // four bytes of BAM nibbles (the high nibble of a byte is the earlier base), one unaligned load where the record
            // holds them all
            u32 w;
            if (nb == 8u) w = *(const u32u *)(q + 4u * k);
            else { w = 0; for (u32 i = 0; i < (nb + 1u) >> 1; i++) w |= (u32)q[4u * k + i] << (8u * i); }
            u32 v = 0;
#pragma unroll
            for (u32 i = 0; i < 8u; i++) v |= code((w >> (8u * (i >> 1) + ((i & 1u) ? 0u : 4u))) & 15u) << (4u * i);
            if (c.fold) {
                u32 bits = 0;
                if (nb == 8u) {
                    const u32 q0 = *(const u32u *)(qq + 8u * k), q1 = *(const u32u *)(qq + 8u * k + 4u);
                    const u32 l0 = ~((q0 | 0x80808080u) - m4) & ~q0 & 0x80808080u, l1 = ~((q1 | 0x80808080u) - m4) & ~q1 & 0x80808080u;
                    bits = ((((l0 >> 7) * 0x00204081u) >> 21) & 0xFu) | (((((l1 >> 7) * 0x00204081u) >> 21) & 0xFu) << 4);
                } else for (u32 i = 0; i < nb; i++) bits |= (u32)(qq[8u * k + i] < (u32)c.minqual) << i;
                // (bit i -> all four bits of nibble i)
                u32 x = bits & 0xFFu;
                x = (x | (x << 12)) & 0x000F000Fu;
                x = (x | (x << 6)) & 0x03030303u;
                x = (x | (x << 3)) & 0x11111111u;
                v ^= (x << 4) - x;
            }
            if (nb < 8u) v &= (1u << (4u * nb)) - 1u;
            const u32 n0 = so + 8u * k, sh = 4u * (n0 & 7u);
            if (v << sh) atomicOr(&d32[n0 >> 3], v << sh);
            if (sh && (v >> (32u - sh))) atomicOr(&d32[(n0 >> 3) + 1u], v >> (32u - sh));
        }
    } else {
    for (u32 k = j; k < n4; k += 8u) {
        const u32 b0 = q[2 * k], b1 = q[2 * k + 1];
        *(u32u *)(s + 4 * k) = letter(b0 >> 4) | (letter(b0 & 15u) << 8) | (letter(b1 >> 4) << 16) | (letter(b1 & 15u) << 24);
    }
    if (j == 0)
        for (u32 k = 4 * n4; k < l_seq; k++) {
            const u32 byte = q[k >> 1];
            s[k] = (u8)letter((k & 1u) ? (byte & 15u) : (byte >> 4));
        }
    }
    q += (l_seq + 1u) / 2u;
    u32 qmin = 0xFFu;                                  // lowest quality of the record (0xFF: none)
    if (c.qual) {
        u8 *__restrict__ ql = c.qual + so;
        for (u32 k = j; k < n4; k += 8u) {
            const u32 w = *(const u32u *)(q + 4 * k);
            *(u32u *)(ql + 4 * k) = w;
            const u32 a = w & 0xFFu, b = (w >> 8) & 0xFFu, cc = (w >> 16) & 0xFFu, d = w >> 24;
            const u32 m = min(min(a, b), min(cc, d));
            qmin = m < qmin ? m : qmin;
        }
        if (j == 0)
            for (u32 k = 4 * n4; k < l_seq; k++) { ql[k] = q[k]; qmin = q[k] < qmin ? q[k] : qmin; }
        if (c.minqual > 0)
            for (int o = 1; o < 8; o <<= 1) { const u32 other = (u32)__shfl_xor((int)qmin, o); qmin = other < qmin ? other : qmin; }
    }
    if (j != 0) return;
    // (MDX_FLAG_HAS_QUAL: the rescaling kernels need not look at the record's first quality to route it, rescale.py:306)
    const u32 hasq = (c.qual && l_seq > 0u && q[0] != 0xFFu) ? 0x4000u : 0u;
    if (hasq) c.flag[r] = (uint16_t)((g16(p + 14) & 0x3FFFu) | hasq);
    if (c.qual && c.minqual > 0) {
        // --min-basequal: a record none of whose qualities is below the threshold cannot be masked (flag bit
        // MDX_FLAG_QUAL_ABOVE_MIN: the tabulation kernel skips its quality windows); a counted record without
        // qualities is what main.py:185-192 warns about
        const u32 fl = (g16(p + 14) & 0x3FFFu) | hasq;
        if (qmin >= (u32)c.minqual) c.flag[r] = (uint16_t)(fl | 0x8000u);
        // (only whether there is one matters: a store where the word is still clear — eight million atomics on one address
        // were 10 ms of the 11.6 ms this kernel took on a file with qualities)
        else if (__hip_atomic_load(c.counters + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicOr(c.counters + 1, 1u);
        if ((fl & 0xF04u) == 0 && (l_seq == 0 || q[0] == 0xFFu)) atomicOr(c.counters, 1u);
    }
    q += l_seq;
    // library: the RG:Z tag against the header's read groups
    int lib = c.lib_default;
    if (c.n_rg > 0) {
        const u8 *end = p + bs;
        bool found = false;
        while (q + 3 <= end && !found) {
            const u32 t0 = q[0], t1 = q[1], ty = q[2];
            q += 3;
            if (ty == 'Z' || ty == 'H') {
                const u8 *z = q;
                while (z < end && *z) z++;
                if (z >= end) break;
                if (t0 == 'R' && t1 == 'G' && ty == 'Z') {
                    const u32 len = (u32)(z - q);
                    lib = 0xFFFF;                        // a read group the header does not list
                    for (int g = 0; g < c.n_rg; g++) {
                        const u32 a0 = c.rg_off[g], a1 = c.rg_off[g + 1];
                        if (a1 - a0 != len) continue;
                        bool same = true;
                        for (u32 j = 0; j < len && same; j++) same = c.rg_names[a0 + j] == q[j];
                        if (same) { lib = c.lib_of_rg[g]; break; }
                    }
                    found = true;
                }
                q = z + 1;
            } else if (ty == 'A' || ty == 'c' || ty == 'C') q += 1;
            else if (ty == 's' || ty == 'S') q += 2;
            else if (ty == 'i' || ty == 'I' || ty == 'f') q += 4;
            else if (ty == 'B') {
                if (q + 5 > end) break;
                const u32 sub = q[0], cntb = g32(q + 1);
                const u32 w = (sub == 'c' || sub == 'C') ? 1u : ((sub == 's' || sub == 'S') ? 2u : 4u);
                if ((u64)cntb * w > (u64)(end - q)) break;
                q += 5 + cntb * w;
            } else break;
        }
    }
    c.lib[r] = (uint16_t)(lib < 0 ? 0xFFFF : lib);
}

}  // namespace

size_t mdx_k_gbam_inflate_lds() { return mdx_inflate::RING + sizeof(mdx_inflate::Tables); }

hipError_t mdx_k_gbam_prepare() {
    return hipFuncSetAttribute((const void *)gbam_inflate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mdx_k_gbam_inflate_lds());
}

void mdx_k_gbam_inflate(const uint8_t *comp, const uint4 *blk, int n_blocks, uint8_t *unc, int *status, hipStream_t s) {
    if (n_blocks <= 0) return;
    hipLaunchKernelGGL(gbam_inflate_kernel, dim3(n_blocks), dim3(64), mdx_k_gbam_inflate_lds(), s, comp, blk, unc, status);
}

void mdx_k_gbam_crc(const uint8_t *unc, const uint4 *blk, const uint32_t *want, const void *tables, int n_blocks, int *bad, hipStream_t s) {
    if (n_blocks <= 0) return;
    hipLaunchKernelGGL(gbam_crc_kernel, dim3(n_blocks), dim3(64), 0, s, unc, blk, want, (const mdx_crc32::Tables *)tables, bad);
}

void mdx_k_gbam_scan(const uint8_t *unc, const uint4 *blk, const int *status, int n_blocks, int first, const int *forced,
                     uint32_t start0, uint32_t total, int n_ref, uint4 *info, uint4 *cnt, hipStream_t s) {
    if (n_blocks - first <= 0) return;
    hipLaunchKernelGGL(gbam_scan_kernel, dim3((n_blocks - first + 63) / 64), dim3(64), 0, s, unc, blk, status, n_blocks, first, forced,
                       start0, total, n_ref, info, cnt);
}

void mdx_k_gbam_unpack(const uint8_t *unc, const uint4 *pre, const uint4 *cnt, int n_blocks, uint32_t n_rec, uint32_t n_cig,
                       uint32_t n_seq, uint32_t *rec_off, const MdxGbamCols &c, hipStream_t s) {
    if (n_blocks <= 0) return;
    hipLaunchKernelGGL(gbam_offsets_kernel, dim3((n_blocks + 63) / 64), dim3(64), 0, s, unc, pre, cnt, n_blocks, rec_off,
                       c.cigar_off, c.seq_off, n_rec, n_cig, n_seq);
    if (n_rec > 0)
        hipLaunchKernelGGL(gbam_unpack_kernel, dim3((n_rec + 31) / 32), dim3(256), 0, s, unc, rec_off, n_rec, c);
}

// ---- the way back: BGZF members out of a stream of encoded records (the rescaling pass writes every record back,
// rescale.py:290-291, :344).  A member = 0xFF00 bytes of the stream in PIECES pieces, a LANE per piece — a gigabyte of records is
// sixty thousand independent pieces, and a lane's encoder (mdx_deflate.h: LZ77 with one hash candidate, reaching back into the
// piece in front, a dynamic Huffman code of the piece's own counts) is serial from end to end —, its working memory (a
// PieceScratch, 75 KB) and its output slot in HBM.  sizes[t] = the piece's bytes (0 for a piece behind the member's end).
namespace {
enum { BGZF_PIECE_SLOT = mdx_deflate::PIECE + 64 };
__global__ void __launch_bounds__(64) bgzf_deflate_kernel(const uint8_t *__restrict__ in, long long n, int n_pieces, uint8_t *__restrict__ slots,
                                                           uint32_t *__restrict__ sizes, mdx_deflate::PieceScratch *__restrict__ scratch) {
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= n_pieces) return;
    const int b = t / mdx_deflate::PIECES, q = t % mdx_deflate::PIECES;
    const long long m0 = (long long)b * mdx_deflate::MAX_IN;
    const uint32_t mlen = (uint32_t)(n - m0 < (long long)mdx_deflate::MAX_IN ? n - m0 : (long long)mdx_deflate::MAX_IN);
    const uint32_t lo = (uint32_t)q * mdx_deflate::PIECE;
    if (lo >= mlen && !(lo == 0 && mlen == 0)) { sizes[t] = 0; return; }
    const uint32_t hi = lo + mdx_deflate::PIECE < mlen ? lo + mdx_deflate::PIECE : mlen;
    sizes[t] = mdx_deflate::deflate_piece(in + m0, lo, hi, lo >= (uint32_t)mdx_deflate::PIECE ? lo - mdx_deflate::PIECE : 0u, hi == mlen,
                                          slots + (size_t)t * BGZF_PIECE_SLOT, (uint32_t)BGZF_PIECE_SLOT, scratch[t]);
}
// A wavefront per member: the gzip header with the BC subfield, the member's pieces side by side, CRC32 (1 KiB per lane, the
// partial values joined by the "append n zero bytes" matrices, as gbam_crc_kernel does it) and ISIZE — at offsets[b] of out.
__global__ void __launch_bounds__(64) bgzf_gather_kernel(const uint8_t *__restrict__ in, long long n, const uint8_t *__restrict__ slots,
                                                          const uint32_t *__restrict__ sizes, const unsigned long long *__restrict__ offsets,
                                                          const mdx_crc32::Tables *__restrict__ tb, uint8_t *__restrict__ out) {
    __shared__ u32 tab[1024];
    const int lane = threadIdx.x, b = (int)blockIdx.x;
    for (int i = lane; i < 1024; i += 64) tab[i] = (&tb->tab[0][0])[i];
    __syncthreads();
    const long long m0 = (long long)b * mdx_deflate::MAX_IN;
    const u32 mlen = (u32)(n - m0 < (long long)mdx_deflate::MAX_IN ? n - m0 : (long long)mdx_deflate::MAX_IN);
    const u32 lo = 1024u * (u32)lane;
    const u32 m = lo < mlen ? (mlen - lo < 1024u ? mlen - lo : 1024u) : 0u;
    const u32 mine = mdx_crc32::crc_bytes(tab, in + m0 + lo, m);
    u32 crc = (u32)__builtin_amdgcn_readlane((int)mine, 0);
    for (int i = 1; i < 64; i++) {
        const u32 li = 1024u * (u32)i;
        if (li >= mlen) break;
        const u32 mi = mlen - li < 1024u ? mlen - li : 1024u;
        crc = mdx_crc32::shift(tb->mat, crc, mi) ^ (u32)__builtin_amdgcn_readlane((int)mine, i);
    }
    uint8_t *dst = out + offsets[b];
    u32 body = 0;
    for (int q = 0; q < mdx_deflate::PIECES; q++) {
        const u32 sz = sizes[b * mdx_deflate::PIECES + q];
        const uint8_t *src = slots + (size_t)(b * mdx_deflate::PIECES + q) * BGZF_PIECE_SLOT;
        for (u32 i = (u32)lane; i < sz; i += 64) dst[18 + body + i] = src[i];
        body += sz;
    }
    if (lane == 0) {
        const u32 total = body + 26;
        const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        for (int i = 0; i < 16; i++) dst[i] = head[i];
        dst[16] = (uint8_t)(total - 1); dst[17] = (uint8_t)((total - 1) >> 8);
        uint8_t *t = dst + 18 + body;
        t[0] = (uint8_t)crc; t[1] = (uint8_t)(crc >> 8); t[2] = (uint8_t)(crc >> 16); t[3] = (uint8_t)(crc >> 24);
        t[4] = (uint8_t)mlen; t[5] = (uint8_t)(mlen >> 8); t[6] = (uint8_t)(mlen >> 16); t[7] = (uint8_t)(mlen >> 24);
    }
}
// ---- the rescaled records of a decoded slab, written back on the device (rescale.py:266-273, :275-281, :344: new QUAL, an MR:f
// tag, every other byte of the record as it stood): the slab's inflated bytes are still in HBM (`unc`), its records at rec_off.
// (1) the patch list of the rescale kernels goes into the records' QUAL fields themselves
__global__ void gbam_patch_qual_kernel(u8 *__restrict__ unc, const u32 *__restrict__ rec_off, const u32 *__restrict__ seq_off, u32 n_rec,
                                       const u64 *__restrict__ patch, const u64 *__restrict__ n_patch, long long cap) {
    const u64 n = n_patch[blockIdx.y] < (u64)cap ? n_patch[blockIdx.y] : (u64)cap;
    const u64 *__restrict__ mine = patch + (size_t)blockIdx.y * (size_t)cap;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 e = mine[i];
        const u32 idx = (u32)e;
        // the record that holds byte idx of the quality column: the last one whose seq_off is at or below it
        u32 lo = 0, hi = n_rec;
        while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (seq_off[mid] <= idx) lo = mid; else hi = mid; }
        u8 *p = unc + rec_off[lo];
        const u32 l_name = p[8], n_cig = g16(p + 12), l_seq = g32(p + 16);
        const u32 at = idx - seq_off[lo];
        if (at < l_seq) p[32 + l_name + 4u * n_cig + (l_seq + 1) / 2 + at] = (u8)(e >> 32);
    }
}
// (2) the bytes every record takes in the output: its own, and seven more for the tag of a rescaled one
__global__ void gbam_out_sizes_kernel(const u8 *__restrict__ unc, const u32 *__restrict__ rec_off, const u8 *__restrict__ rescaled, u32 n_rec,
                                      u32 *__restrict__ sizes) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    sizes[r] = g32(unc + rec_off[r] - 4) + 4u + (rescaled[r] ? 7u : 0u);
}
// (3) eight lanes per record: the record to its place in the output stream; a rescaled one with block_size seven larger and
// "MR" 'f' <float> behind its last tag — unless it has an MR tag already (rescale.py:277-278: the reference stops there),
// whose lowest record index lands in *clash
__global__ void gbam_write_back_kernel(const u8 *__restrict__ unc, const u32 *__restrict__ rec_off, const unsigned long long *__restrict__ out_off,
                                       const u8 *__restrict__ rescaled, const float *__restrict__ mr, u32 n_rec, u8 *__restrict__ out,
                                       int *__restrict__ clash) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 r = t >> 3, j = t & 7u;
    if (r >= n_rec) return;
    const u8 *__restrict__ p = unc + rec_off[r];
    const u32 bs = g32(p - 4);
    u8 *__restrict__ o = out + out_off[r];
    const bool rs = rescaled[r] != 0;
    for (u32 k = j; k < bs; k += 8u) o[4 + k] = p[k];
    if (j != 0) return;
    const u32 nbs = bs + (rs ? 7u : 0u);
    o[0] = (u8)nbs; o[1] = (u8)(nbs >> 8); o[2] = (u8)(nbs >> 16); o[3] = (u8)(nbs >> 24);
    if (!rs) return;
    u8 *t7 = o + 4 + bs;
    const u32 fb = __float_as_uint(mr[r]);
    t7[0] = 'M'; t7[1] = 'R'; t7[2] = 'f'; t7[3] = (u8)fb; t7[4] = (u8)(fb >> 8); t7[5] = (u8)(fb >> 16); t7[6] = (u8)(fb >> 24);
    // an MR tag among the record's own?
    const u32 l_name = p[8], n_cig = g16(p + 12), l_seq = g32(p + 16);
    const u8 *q = p + 32 + l_name + 4u * n_cig + (l_seq + 1) / 2 + l_seq, *end = p + bs;
    while (q + 3 <= end) {
        const u32 t0 = q[0], t1 = q[1], ty = q[2];
        if (t0 == 'M' && t1 == 'R') { atomicMin(clash, (int)r); break; }
        q += 3;
        if (ty == 'Z' || ty == 'H') { while (q < end && *q) q++; q++; }
        else if (ty == 'A' || ty == 'c' || ty == 'C') q += 1;
        else if (ty == 's' || ty == 'S') q += 2;
        else if (ty == 'i' || ty == 'I' || ty == 'f') q += 4;
        else if (ty == 'B') {
            if (q + 5 > end) break;
            const u32 sub = q[0], cntb = g32(q + 1);
            const u32 w = (sub == 'c' || sub == 'C') ? 1u : ((sub == 's' || sub == 'S') ? 2u : 4u);
            if ((u64)cntb * w > (u64)(end - q)) break;
            q += 5 + cntb * w;
        } else break;
    }
}
}  // namespace
void mdx_k_gbam_patch_qual(uint8_t *unc, const uint32_t *rec_off, const uint32_t *seq_off, uint32_t n_rec, const unsigned long long *patch,
                           const unsigned long long *n_patch, long long cap, int parts, hipStream_t s) {
    if (n_rec == 0 || parts <= 0) return;
    hipLaunchKernelGGL(gbam_patch_qual_kernel, dim3(16, parts), dim3(256), 0, s, unc, rec_off, seq_off, n_rec, (const u64 *)patch, (const u64 *)n_patch, cap);
}
void mdx_k_gbam_out_sizes(const uint8_t *unc, const uint32_t *rec_off, const uint8_t *rescaled, uint32_t n_rec, uint32_t *sizes, hipStream_t s) {
    if (n_rec == 0) return;
    hipLaunchKernelGGL(gbam_out_sizes_kernel, dim3((n_rec + 255) / 256), dim3(256), 0, s, unc, rec_off, rescaled, n_rec, sizes);
}
void mdx_k_gbam_write_back(const uint8_t *unc, const uint32_t *rec_off, const unsigned long long *out_off, const uint8_t *rescaled, const float *mr,
                           uint32_t n_rec, uint8_t *out, int *clash, hipStream_t s) {
    if (n_rec == 0) return;
    hipLaunchKernelGGL(gbam_write_back_kernel, dim3((n_rec + 31) / 32), dim3(256), 0, s, unc, rec_off, out_off, rescaled, mr, n_rec, out, clash);
}
int mdx_k_bgzf_pieces() { return mdx_deflate::PIECES; }
size_t mdx_k_bgzf_slot_bytes() { return BGZF_PIECE_SLOT; }
size_t mdx_k_bgzf_scratch_bytes(int n_members) { return (size_t)n_members * mdx_deflate::PIECES * sizeof(mdx_deflate::PieceScratch); }
void mdx_k_bgzf_deflate(const uint8_t *d_in, long long n, int n_members, uint8_t *d_slots, uint32_t *d_sizes, void *d_scratch, hipStream_t s) {
    if (n_members <= 0) return;
    const int n_pieces = n_members * mdx_deflate::PIECES;
    hipLaunchKernelGGL(bgzf_deflate_kernel, dim3((n_pieces + 63) / 64), dim3(64), 0, s, d_in, n, n_pieces, d_slots, d_sizes,
                       (mdx_deflate::PieceScratch *)d_scratch);
}
void mdx_k_bgzf_gather(const uint8_t *d_in, long long n, const uint8_t *d_slots, const uint32_t *d_sizes, const unsigned long long *d_offsets,
                       int n_members, const void *tables, uint8_t *d_out, hipStream_t s) {
    if (n_members <= 0) return;
    hipLaunchKernelGGL(bgzf_gather_kernel, dim3(n_members), dim3(64), 0, s, d_in, n, d_slots, d_sizes, d_offsets, (const mdx_crc32::Tables *)tables, d_out);
}
