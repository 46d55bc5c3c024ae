// Bucketing the records of a batch by library, on the device (gfx950).
//
// The reference keys its tables by library — the (sample, library) of a record's read group, reader.py:47-50,63-81,
// statistics.py:12-20,60-73 — and a BAM file interleaves the libraries in any order.  The packed tabulation kernel keeps
// ONE library's tables in the LDS and counts plain matches in registers, so it wants a library's records together: this
// file turns a batch into a copy of itself ordered by library, batch order kept within a library, the flag filter of
// reader.py:121-132 applied on the way (a record it drops has no place) — the per-record columns, and the CIGAR
// operations and the 4-bit SEQ codes they point at (with --min-basequal the mask is in those codes, MDX_SEQ_4BITQ).  The bytes a record
// points at move with it because a library's records, left where the file put them, are every n-th record of the
// columns: with 128-byte requests a SEQ line (two or three records) and a CIGAR line (twenty) would cross the fabric once
// per library that owns a record in it — measured with the columns alone bucketed: 8 libraries 1.63 x the time of one.
//
// A stable counting sort: per-block counts by library (transposed, [library][block]); one exclusive scan over that matrix
// read row by row, which is at once the first place of every library and of every block's share of it; the scatter of
// the fixed columns, one wavefront per block walking its records in order, the places of a step's records of one
// library handed out by ballot; an exclusive scan of the CIGAR and SEQ lengths in their new order (the new offset
// columns); and the copy — CIGAR words, and SEQ nibbles shifted from the phase they had to the phase they get.  HBM-bound streaming work, about 100 bytes in and 100 out per record.
#include "mdx_internal.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef unsigned long long u64;
typedef long long i64;

#define LS_COUNT_THREADS 256
#define LS_LDS_LIBS 4096          // libraries whose counts / places a block keeps in the LDS (beyond: global memory)
#define LS_ERR_BAD_READ 6
#define LS_SCAN_ITEMS 16          // entries per thread of the offset scans (256 threads: 4096 entries per block)

namespace {

struct Geometry { int nblk; i64 per; };
// blocks of `per` consecutive records (a multiple of 64); at most 2^20 (library, block) pairs
Geometry geometry(i64 n, int nlib) {
    i64 cap = ((i64)1 << 20) / (nlib > 0 ? nlib : 1);
    if (cap < 64) cap = 64;
    if (cap > 16384) cap = 16384;
    i64 nblk = (n + 4095) / 4096;
    if (nblk < 1) nblk = 1;
    if (nblk > cap) nblk = cap;
    i64 per = (n + nblk - 1) / nblk;
    per = (per + 63) / 64 * 64;
    if (per < 64) per = 64;
    nblk = (n + per - 1) / per;
    if (nblk < 1) nblk = 1;
    Geometry g;
    g.nblk = (int)nblk; g.per = per;
    return g;
}

size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
i64 scan_blocks(i64 n) { return (n + 1 + 256 * LS_SCAN_ITEMS - 1) / (256 * LS_SCAN_ITEMS); }

// scratch of one sort: [counts nblk x nlib][source CIGAR offset n][source SEQ offset n][block sums 2 x scan_blocks]
struct Scratch { u32 *cnt, *src_co, *src_so, *bs_c, *bs_s; };
Scratch scratch_layout(void *p, i64 n, int nlib) {
    const Geometry g = geometry(n, nlib);
    Scratch s;
    u8 *q = (u8 *)p;
    s.cnt = (u32 *)q; q += align16(((size_t)g.nblk * (size_t)nlib + 1) * 4);
    s.src_co = (u32 *)q; q += align16((size_t)(n + 1) * 4);
    s.src_so = (u32 *)q; q += align16((size_t)(n + 1) * 4);
    s.bs_c = (u32 *)q; q += align16((size_t)(scan_blocks(n) + 1) * 4);
    s.bs_s = (u32 *)q;
    return s;
}

}  // namespace

size_t mdx_k_libsort_bytes(int64_t n, int64_t n_cigar, int64_t n_bases, int nlib) {
    const size_t n1 = (size_t)(n > 0 ? n : 0) + 16;
    return 16 + align16((size_t)(nlib + 1) * 4) + align16(n1 * 4) * 6 + align16(n1 * 2) + align16((size_t)(n_cigar + 16) * 4) +
           align16(((size_t)n_bases + 1) / 2 + 64) + 256;
}

void mdx_k_libsort_layout(void *blob, int64_t n, int64_t n_cigar, int64_t n_bases, int nlib, MdxLibSort *out) {
    const size_t n1 = (size_t)(n > 0 ? n : 0) + 16;
    u8 *p = (u8 *)blob;
    out->bad = (unsigned long long *)p; p += 16;
    out->lib_start = (u32 *)p; p += align16((size_t)(nlib + 1) * 4);
    out->perm = (u32 *)p; p += align16(n1 * 4);
    out->tid = (int32_t *)p; p += align16(n1 * 4);
    out->pos = (int32_t *)p; p += align16(n1 * 4);
    out->tlen = (int32_t *)p; p += align16(n1 * 4);
    out->cigar_off = (u32 *)p; p += align16(n1 * 4);
    out->seq_off = (u32 *)p; p += align16(n1 * 4);
    out->flag = (u16 *)p; p += align16(n1 * 2);
    out->cigar = (u32 *)p; p += align16((size_t)(n_cigar + 16) * 4);
    out->seq = p;
    out->seq_bytes = ((size_t)n_bases + 1) / 2 + 64;
}

size_t mdx_k_libsort_scratch_bytes(int64_t n, int nlib) {
    const Geometry g = geometry(n, nlib);
    return align16(((size_t)g.nblk * (size_t)nlib + 1) * 4) + 2 * align16((size_t)(n + 1) * 4) + 2 * align16((size_t)(scan_blocks(n) + 1) * 4) + 64;
}

// cnt[l * nblk + b] = kept records of library l among block b's records
__global__ __launch_bounds__(LS_COUNT_THREADS) void libsort_count_kernel(i64 n, i64 per, const u16 *__restrict__ flag, const u16 *__restrict__ lib,
                                                                         int nlib, int nblk, u32 *__restrict__ cnt, u64 *__restrict__ bad) {
    __shared__ u32 h[LS_LDS_LIBS];
    const bool in_lds = nlib <= LS_LDS_LIBS;
    const int b = blockIdx.x;
    if (in_lds) {
        for (int l = threadIdx.x; l < nlib; l += LS_COUNT_THREADS) h[l] = 0u;
        __syncthreads();
    }
    const i64 lo = (i64)b * per, hi = lo + per < n ? lo + per : n;
    for (i64 i = lo + threadIdx.x; i < hi; i += LS_COUNT_THREADS) {
        const u32 fl = flag[i], lb = lib[i];
        if (fl & 0xF04u) continue;                      // reader.py:121-132
        if (lb >= (u32)nlib) {
            // (a library the header does not know: what the tabulation kernel reports for a record it would count)
            atomicMin(bad, ((u64)i << 8) | (u64)LS_ERR_BAD_READ);
            continue;
        }
        if (in_lds) atomicAdd(&h[lb], 1u);
        else atomicAdd(&cnt[(size_t)lb * nblk + b], 1u);      // (zeroed by the launch)
    }
    if (in_lds) {
        __syncthreads();
        for (int l = threadIdx.x; l < nlib; l += LS_COUNT_THREADS) cnt[(size_t)l * nblk + b] = h[l];
    }
}

// exclusive scan of v[0 .. m) in place (one block; every thread a contiguous stretch); v[m] = the total
__global__ __launch_bounds__(1024) void libsort_scan_kernel(u32 *__restrict__ v, i64 m) {
    __shared__ u32 part[1024];
    const int t = threadIdx.x;
    const i64 per = (m + 1023) / 1024;
    const i64 lo = (i64)t * per < m ? (i64)t * per : m, hi = lo + per < m ? lo + per : m;
    u32 sum = 0;
    for (i64 i = lo; i < hi; i++) sum += v[i];
    part[t] = sum;
    __syncthreads();
    // (Hillis-Steele over the 1024 partial sums)
    for (int o = 1; o < 1024; o <<= 1) {
        const u32 x = t >= o ? part[t - o] : 0u;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    u32 run = t ? part[t - 1] : 0u;
    for (i64 i = lo; i < hi; i++) {
        const u32 c = v[i];
        v[i] = run;
        run += c;
    }
    if (t == 1023) v[m] = part[1023];
}
// lib_start[l] = first place of library l (place[l * nblk]), lib_start[nlib] = the kept records
__global__ void libsort_starts_kernel(const u32 *__restrict__ place, int nlib, int nblk, u32 *__restrict__ lib_start) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l <= nlib) lib_start[l] = place[(size_t)l * nblk];
}

// one wavefront per block: its records in batch order, 64 at a time; the record's CIGAR and SEQ lengths go to the places of
// the new offset columns (scanned next), its old offsets to the scratch columns the copy reads
__global__ __launch_bounds__(64) void libsort_scatter_kernel(i64 n, i64 per, const u16 *__restrict__ flag, const u16 *__restrict__ lib,
                                                             const int32_t *__restrict__ tid, const int32_t *__restrict__ pos,
                                                             const int32_t *__restrict__ tlen, const u32 *__restrict__ cigar_off,
                                                             const u32 *__restrict__ seq_off, int nlib, int nblk, u32 *__restrict__ place,
                                                             MdxLibSort out, u32 *__restrict__ src_co, u32 *__restrict__ src_so) {
    __shared__ u32 pl[LS_LDS_LIBS];
    const bool in_lds = nlib <= LS_LDS_LIBS;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (in_lds) {
        for (int l = lane; l < nlib; l += 64) pl[l] = place[(size_t)l * nblk + b];
        __syncthreads();
    }
    const i64 lo = (i64)b * per, hi = lo + per < n ? lo + per : n;
    for (i64 base = lo; base < hi; base += 64) {
        const i64 i = base + lane;
        const bool valid = i < hi;
        const u32 fl = valid ? flag[i] : 0x4u, lb = valid ? lib[i] : 0u;
        const bool kept = valid && !(fl & 0xF04u) && lb < (u32)nlib;
        // the columns of the record, requested before its place is known
        int32_t c_tid = 0, c_pos = 0, c_tlen = 0;
        u32 c0 = 0, c1 = 0, s0 = 0, s1 = 0;
        if (kept) {
            c_tid = tid[i]; c_pos = pos[i]; c_tlen = tlen[i];
            c0 = cigar_off[i]; c1 = cigar_off[i + 1]; s0 = seq_off[i]; s1 = seq_off[i + 1];
        }
        u32 dst = 0;
        u64 todo = __ballot(kept);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const u32 key = (u32)__builtin_amdgcn_readlane((int)lb, leader);
            const u64 same = __ballot(kept && lb == key);
            u32 first;
            if (in_lds) first = pl[key];
            else first = place[(size_t)key * nblk + b];
            first = (u32)__builtin_amdgcn_readfirstlane((int)first);
            if (kept && lb == key) dst = first + (u32)__builtin_amdgcn_mbcnt_hi((u32)(same >> 32), __builtin_amdgcn_mbcnt_lo((u32)same, 0u));
            __syncthreads();        // (one wavefront: the read of the place in front of its update)
            if (lane == leader) {
                if (in_lds) pl[key] = first + (u32)__popcll(same);
                else place[(size_t)key * nblk + b] = first + (u32)__popcll(same);
            }
            __syncthreads();
            todo &= ~same;
        }
        if (kept) {
            out.perm[dst] = (u32)i;
            out.flag[dst] = (u16)fl;
            out.tid[dst] = c_tid; out.pos[dst] = c_pos; out.tlen[dst] = c_tlen;
            // (a record whose offsets run backwards — not a batch any decoder makes — keeps a length of zero and is the
            // tabulation kernel's to report)
            out.cigar_off[dst] = c1 >= c0 ? c1 - c0 : 0u; out.seq_off[dst] = s1 >= s0 ? s1 - s0 : 0u;
            src_co[dst] = c0; src_so[dst] = s0;
        }
    }
}

// The new offset columns: exclusive scans of the lengths the scatter left in them, over places [0, kept] (entry `kept`
// becomes the total), in three steps — sums per block of 4096, their scan (libsort_scan_kernel), the scan within the blocks
__global__ __launch_bounds__(256) void libsort_blocksum_kernel(const u32 *__restrict__ kept_p, const u32 *__restrict__ a, const u32 *__restrict__ b,
                                                               u32 *__restrict__ bs_a, u32 *__restrict__ bs_b) {
    __shared__ u32 ra[4], rb[4];
    const i64 kept = *kept_p;
    const i64 i0 = ((i64)blockIdx.x * 256 + threadIdx.x) * LS_SCAN_ITEMS;
    u32 sa = 0, sb = 0;
#pragma unroll
    for (int k = 0; k < LS_SCAN_ITEMS; k++)
        if (i0 + k < kept) { sa += a[i0 + k]; sb += b[i0 + k]; }
    for (int o = 32; o; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
    if ((threadIdx.x & 63) == 0) { ra[threadIdx.x >> 6] = sa; rb[threadIdx.x >> 6] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) { bs_a[blockIdx.x] = ra[0] + ra[1] + ra[2] + ra[3]; bs_b[blockIdx.x] = rb[0] + rb[1] + rb[2] + rb[3]; }
}
__global__ __launch_bounds__(256) void libsort_blockscan_kernel(const u32 *__restrict__ kept_p, u32 *__restrict__ a, u32 *__restrict__ b,
                                                                const u32 *__restrict__ bs_a, const u32 *__restrict__ bs_b) {
    __shared__ u32 ta[256], tb[256];
    const i64 kept = *kept_p;
    const int t = threadIdx.x;
    const i64 i0 = ((i64)blockIdx.x * 256 + t) * LS_SCAN_ITEMS;
    u32 va[LS_SCAN_ITEMS], vb[LS_SCAN_ITEMS];
    u32 sa = 0, sb = 0;
#pragma unroll
    for (int k = 0; k < LS_SCAN_ITEMS; k++) {
        va[k] = i0 + k < kept ? a[i0 + k] : 0u; vb[k] = i0 + k < kept ? b[i0 + k] : 0u;
        sa += va[k]; sb += vb[k];
    }
    ta[t] = sa; tb[t] = sb;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const u32 xa = t >= o ? ta[t - o] : 0u, xb = t >= o ? tb[t - o] : 0u;
        __syncthreads();
        ta[t] += xa; tb[t] += xb;
        __syncthreads();
    }
    u32 ra = bs_a[blockIdx.x] + (t ? ta[t - 1] : 0u), rb = bs_b[blockIdx.x] + (t ? tb[t - 1] : 0u);
#pragma unroll
    for (int k = 0; k < LS_SCAN_ITEMS; k++) {
        if (i0 + k <= kept) { a[i0 + k] = ra; b[i0 + k] = rb; }
        ra += va[k]; rb += vb[k];
    }
}

// The bytes the records point at.  A wavefront takes 64 consecutive places: a lane copies the CIGAR words of one record,
// then the wavefront fills the stretch of the new SEQ column its records occupy a dword at a time — a lane owns the dwords
// whose first nibble lies in the stretch, finds the record that nibble belongs to among the 64 offsets in the LDS, and
// funnels the nibbles of that record (and, at a border, of the next ones) out of the source column from nibble
// src + ph on (ph: the nibbles between the dword-aligned base and the column's first one).  Whole dwords, plain stores, the
// loads of a lane independent of one another.
#define LS_COPY_WAVES 4
__global__ __launch_bounds__(64 * LS_COPY_WAVES) void libsort_copy_kernel(const u32 *__restrict__ kept_p, const u32 *__restrict__ src_co,
                                                                          const u32 *__restrict__ src_so, const u32 *__restrict__ cigar,
                                                                          const u32 *__restrict__ seq32, u32 ph, u32 seq_words, MdxLibSort out) {
    __shared__ u32 s_off[LS_COPY_WAVES][72], s_src[LS_COPY_WAVES][72];
    const i64 kept = *kept_p;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 *const off = s_off[wave], *const src = s_src[wave];
    u32 *const oseq = (u32 *)out.seq;
    const i64 nw = (i64)gridDim.x * LS_COPY_WAVES;
    for (i64 r0 = ((i64)blockIdx.x * LS_COPY_WAVES + wave) * 64; r0 < kept; r0 += nw * 64) {
        const i64 r = r0 + lane;
        const bool valid = r < kept;
        // the places' new offsets (entry `kept` = the total) and old ones; four more places behind the 64 for the dword that
        // straddles the stretch's end
        const i64 rc = r < kept ? r : kept;
        off[lane] = out.seq_off[rc];
        src[lane] = valid ? src_so[r] : 0u;
        if (lane < 8) {
            const i64 rx = r0 + 64 + lane < kept ? r0 + 64 + lane : kept;
            off[64 + lane] = out.seq_off[rx];
            src[64 + lane] = rx < kept ? src_so[rx] : 0u;
        }
        if (valid) {
            const u32 c_src = src_co[r], c_dst = out.cigar_off[r], clen = out.cigar_off[r + 1] - c_dst;
            for (u32 k = 0; k < clen; k++) out.cigar[c_dst + k] = cigar[c_src + k];
        }
        // (the wavefront's own part of the LDS: its writes land in order, in front of its reads — no barrier of the block, whose
        // wavefronts take different numbers of turns)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const u64 D0 = off[0], D1 = off[64];
        // place i's offsets: from the LDS up to 71, beyond (records of a nibble or two at a border) from the column
        auto off_at = [&](const int i) -> u64 {
            if (i <= 71) return off[i];
            const i64 rx = r0 + i < kept ? r0 + i : kept;
            return out.seq_off[rx];
        };
        auto src_at = [&](const int i) -> u64 {
            if (i <= 71) return src[i];
            const i64 rx = r0 + i;
            return rx < kept ? src_so[rx] : 0u;
        };
        // the last place that begins at or in front of nibble x of the stretch
        auto find = [&](const u64 x) -> int {
            int lo = 0, hi = 64;            // off[lo] <= x < off[hi] (x < D1)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if ((u64)off[mid] <= x) lo = mid; else hi = mid;
            }
            return lo;
        };
        for (u64 j = ((D0 + 7) >> 3) + (u32)lane; j * 8 < D1; j += 64) {
            const u64 w_lo = j * 8, w_hi = w_lo + 8;
            int i = find(w_lo);
            u32 v = 0u;
            u64 x = w_lo;
            while (x < w_hi) {
                u64 e = off_at(i + 1);
                while (e <= x && r0 + i + 1 < kept) { i++; e = off_at(i + 1); }
                if (e <= x) break;                          // behind the last kept base
                const u32 take = (u32)((e < w_hi ? e : w_hi) - x);
                const u64 sn = src_at(i) + ph + (x - off_at(i));
                // (the second word only where the column still has one: a caller's own column ends with its last base)
                const u32 wi = (u32)(sn >> 3);
                const u32 w0 = seq32[wi], w1 = wi + 1u < seq_words ? seq32[wi + 1u] : 0u;
                u32 bits = __builtin_amdgcn_alignbit(w1, w0, 4u * (u32)(sn & 7));
                if (take < 8u) bits &= (1u << (4u * take)) - 1u;
                v |= bits << (4u * (u32)(x - w_lo));
                x += take;
            }
            oseq[j] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

void mdx_k_libsort(int64_t n, int64_t n_cigar, int64_t n_bases, const uint16_t *flag, const uint16_t *lib, const int32_t *tid,
                   const int32_t *pos, const int32_t *tlen, const uint32_t *cigar_off, const uint32_t *cigar, const uint32_t *seq_off,
                   const uint8_t *seq4, int nlib, void *scratch, const MdxLibSort &out, hipStream_t s) {
    const Geometry g = geometry(n, nlib);
    const Scratch sc = scratch_layout(scratch, n, nlib);
    const i64 m = (i64)g.nblk * nlib;
    if (nlib > LS_LDS_LIBS) (void)hipMemsetAsync(sc.cnt, 0, (size_t)m * 4, s);
    (void)hipMemsetAsync(out.bad, 0xFF, 8, s);
    // (what lies behind the last kept base stays zero)
    (void)hipMemsetAsync(out.seq, 0, out.seq_bytes, s);
    hipLaunchKernelGGL(libsort_count_kernel, dim3(g.nblk), dim3(LS_COUNT_THREADS), 0, s, (i64)n, g.per, flag, lib, nlib, g.nblk, sc.cnt,
                       (u64 *)out.bad);
    hipLaunchKernelGGL(libsort_scan_kernel, dim3(1), dim3(1024), 0, s, sc.cnt, m);
    hipLaunchKernelGGL(libsort_starts_kernel, dim3((nlib + 1 + 255) / 256), dim3(256), 0, s, sc.cnt, nlib, g.nblk, out.lib_start);
    hipLaunchKernelGGL(libsort_scatter_kernel, dim3(g.nblk), dim3(64), 0, s, (i64)n, g.per, flag, lib, tid, pos, tlen, cigar_off, seq_off,
                       nlib, g.nblk, sc.cnt, out, sc.src_co, sc.src_so);
    const i64 nb = scan_blocks(n);
    const u32 *const kept_p = out.lib_start + nlib;
    hipLaunchKernelGGL(libsort_blocksum_kernel, dim3((unsigned)nb), dim3(256), 0, s, kept_p, out.cigar_off, out.seq_off, sc.bs_c, sc.bs_s);
    hipLaunchKernelGGL(libsort_scan_kernel, dim3(1), dim3(1024), 0, s, sc.bs_c, nb);
    hipLaunchKernelGGL(libsort_scan_kernel, dim3(1), dim3(1024), 0, s, sc.bs_s, nb);
    hipLaunchKernelGGL(libsort_blockscan_kernel, dim3((unsigned)nb), dim3(256), 0, s, kept_p, out.cigar_off, out.seq_off, sc.bs_c, sc.bs_s);
    // (the SEQ column from its dword-aligned base: the phase counts nibbles)
    const u32 ph = 2u * (u32)((size_t)seq4 & 3);
    const u32 *const seq32 = (const u32 *)(seq4 - ((size_t)seq4 & 3));
    const i64 tiles = (n + 64 * LS_COPY_WAVES - 1) / (64 * LS_COPY_WAVES);
    const unsigned cgrid = (unsigned)(tiles < 8192 ? (tiles > 0 ? tiles : 1) : 8192);
    // (dwords of the column counted from that base: no byte behind its last one is read)
    const u32 seq_words = (u32)((((size_t)seq4 & 3) + ((size_t)n_bases + 1) / 2 + 3) / 4);
    hipLaunchKernelGGL(libsort_copy_kernel, dim3(cgrid), dim3(64 * LS_COPY_WAVES), 0, s, kept_p, sc.src_co, sc.src_so, cigar, seq32, ph, seq_words, out);
}
