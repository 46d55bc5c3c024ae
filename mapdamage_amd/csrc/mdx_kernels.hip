// gfx950 (MI355X / CDNA4) kernels of the damage-tabulation engine.
//
// Work decomposition (integer/byte histogram work — HBM/LDS bound, no MFMA):
//   * a wavefront (64 lanes) owns a tile of 64 consecutive records;
//   * phase 1, lane-per-record: coalesced SoA loads of the per-record columns, flag filter
//     (reader.py:121-132), CIGAR scan (clips, reference span, column count), fragment-length
//     update (statistics.py:117-126), soft-clip update (statistics.py:37-51), error checks;
//   * phase 2, wavefront-per-record: the record's scalars are broadcast with v_readlane and
//     the 64 lanes walk its alignment columns (bases and reference classes are read with
//     consecutive addresses across lanes) and bump counters;
//   * counters live in a block-private LDS image of the raw tables (ds_add_u32); at block end
//     the image is stored to a per-block slot and a second kernel column-sums the slots into
//     the u64 accumulators (no global atomics on the hot path).
// Counting is done in *reference orientation* (left-/right-anchored, no complementing); the
// strand step of main.py:200-205 (reverse-complement + flank swap) becomes a fixed permutation
// applied once by finalize_kernel.  The oracle (oracle/mdx_oracle.c) builds and reverses the
// strings literally instead, so the two share no derivation.
#include "mdx_internal.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef unsigned long long u64;
typedef long long i64;

#define MDX_BLOCK 512
#define SYM_GAP 4
#define SYM_OTHER 5
#define COL_S 24
#define ERR_BAD_READ 6

// column of substitution ref>read (mapdamage/seq.py:6-30 order, "Total" removed); 31 = none
__constant__ u8 c_col[25] = {
    31, 8, 6, 9, 16,
    11, 31, 10, 5, 18,
    4, 14, 31, 15, 19,
    13, 7, 12, 31, 17,
    20, 22, 23, 21, 31};
// column seen from the reverse strand (complement both symbols)
__constant__ u8 c_comp_col[25] = {3, 2, 1, 0, 5, 4, 7, 6, 12, 13, 14, 15, 8,
                                  9, 10, 11, 17, 16, 19, 18, 21, 20, 23, 22, 24};

int mdx_k_block_threads() { return MDX_BLOCK; }
size_t mdx_k_lds_bytes(const MdxDims &d) { return (size_t)d.w_total * 4; }

// ASCII -> symbol class.  (ch >> 1) & 3 maps A,C,T,G to 0,1,2,3; the byte is accepted only if
// it is exactly that upper-case letter ("nt in 'ACGT-'", statistics.py:27).
__device__ __forceinline__ int classify_ascii(u32 ch) {
    u32 k = (ch >> 1) & 3u;
    u32 recon = (0x47544341u >> (k * 8)) & 0xFFu;  // 'A','C','T','G'
    int code = (int)(k ^ (k >> 1));                // A0 C1 G2 T3
    return ch == recon ? code : (ch == (u32)'-' ? SYM_GAP : SYM_OTHER);
}

__global__ void encode_ref_kernel(const u8 *__restrict__ in, u8 *__restrict__ out, i64 n) {
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    i64 stride = (i64)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        u32 ch = in[i];
        if (ch >= 'a' && ch <= 'z') ch -= 32;  // .upper() of main.py:180 / align.py:32-33
        out[i] = (u8)classify_ascii(ch);
    }
}

void mdx_k_encode_ref(const u8 *d_ascii, u8 *d_codes, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(encode_ref_kernel, dim3(grid), dim3(256), 0, s, d_ascii, d_codes, (i64)n);
}

template <bool USE_LDS>
__device__ __forceinline__ void bump(u32 *lds, u64 *raw, int idx) {
    if (USE_LDS) atomicAdd(&lds[idx], 1u);
    else atomicAdd(&raw[idx], 1ull);
}

__device__ __forceinline__ void flag_error(u64 *err, i64 read, int code) {
    atomicMin(err, ((u64)read << 8) | (u64)code);
}

__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ i64 rl64(i64 v, int lane) {
    int lo = __builtin_amdgcn_readlane((int)(v & 0xFFFFFFFFll), lane);
    int hi = __builtin_amdgcn_readlane((int)(v >> 32), lane);
    return ((i64)hi << 32) | (u32)lo;
}

// record descriptor bits
#define D_REV 1
#define D_SIMPLE 2
#define D_HASQ 4

template <bool USE_LDS, bool MASK>
__global__ __launch_bounds__(MDX_BLOCK) void tabulate_kernel(MdxTabArgs a) {
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

    const i64 ntiles = (a.n_reads + 63) >> 6;
    for (i64 tile = gwave; tile < ntiles; tile += nwaves) {
        // ------------------------------------------------------------ phase 1: lane per record
        const i64 ri = tile * 64 + lane;
        const bool valid = ri < a.n_reads;
        const u32 fl = valid ? (u32)a.flag[ri] : 0x4u;
        bool kept = (fl & 0xF04u) == 0;  // reader.py:121-132
        int desc = 0, libid = 0, nq = 0, n0 = 0, ncols = 0, nI = 0, nbefore = 0, nafter = 0;
        int cig_n = 0;
        i64 rbase = 0, sq = 0, cig_o = 0;
        if (kept) {
            const int rev = (fl >> 4) & 1;
            libid = a.lib[ri];
            const int tid = a.tid[ri];
            const i64 pos = a.pos[ri];
            cig_o = a.cigar_off[ri];
            cig_n = (int)(a.cigar_off[ri + 1] - (u32)cig_o);
            const i64 so = a.seq_off[ri];
            const i64 lseq = (i64)a.seq_off[ri + 1] - so;
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
                sq = so + qs;
                nbefore = pos < A ? (int)pos : A;
                nafter = clen - aend < A ? (int)(clen - aend) : A;
                desc = rev | ((sI == 0 && sDN == 0 && rlen > 0) ? D_SIMPLE : 0);
                if (MASK && a.qual != nullptr && a.qual[so] != 0xFF) desc |= D_HASQ;
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
                        bump<USE_LDS>(lds, raw, lbase + d.off_lgd() + (kind * 2 + rev) * d.lgd_lds + (int)flen);
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
        u64 todo = __ballot(kept);
        if (lane == 0 && todo) {
            const int cnt = __popcll(todo);
            if (USE_LDS) atomicAdd(&lds[d.w_total - 1], (u32)cnt);
            else atomicAdd(&raw[d.w_total - 1], (u64)cnt);
        }

        // ------------------------------------------------------------ phase 2: wave per record
        while (todo) {
            const int j = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int s_desc = rl(desc, j);
            const int s_nq = rl(nq, j);
            const int s_n0 = rl(n0, j);
            const int s_nbefore = rl(nbefore, j);
            const int s_nafter = rl(nafter, j);
            const i64 s_rbase = rl64(rbase, j);
            const i64 s_sq = rl64(sq, j);
            const int rev = s_desc & D_REV;
            const bool hasq = MASK && (s_desc & D_HASQ);
            const int lb = rl(libid, j) * d.w_lib;
            const int b_mis = lb + (rev * 2) * L * 25;
            const int b_comp = lb + d.off_comp() + (rev * 2) * (L + A) * 4;
            const int b_m = lb + d.off_m() + (rev * 2) * L * 4;
            const u8 *__restrict__ rp = a.ref + s_rbase;
            const u8 *__restrict__ sp = a.seq + s_sq;
            const u8 *__restrict__ qp = MASK ? a.qual + s_sq : nullptr;

            if (s_desc & D_SIMPLE) {
                // gapped read == query, gapped reference == reference slice (align.py:38-50 is
                // the identity): column c pairs sp[c] with rp[c]; flanks are rp[-d], rp[nq-1+d]
                const int Lp = s_nq < L ? s_nq : L;
                const int span = Lp + A;
                for (int t = lane; t < 2 * span; t += 64) {
                    const int side = t >= span;
                    const int p = (side ? t - span : t) - A;  // [-A, Lp)
                    const int c = side ? s_nq - 1 - p : p;
                    if (p < 0) {
                        // statistics.py:85-93 (flank bases, already clamped to the contig)
                        const int dist = -p;
                        if (dist <= (side ? s_nafter : s_nbefore)) {
                            const int r = rp[c];
                            if (r < 4) bump<USE_LDS>(lds, raw, b_comp + (side * (L + A) + L + dist - 1) * 4 + r);
                        }
                    } else {
                        const int r = rp[c];
                        const int s = classify_ascii(sp[c]);
                        const bool m = hasq && (int)qp[c] < a.minqual;  // align.py:65-71
                        if (s < 4) {
                            if (!m && r == s) bump<USE_LDS>(lds, raw, b_m + (side * L + p) * 4 + s);
                            else bump<USE_LDS>(lds, raw, b_comp + (side * (L + A) + p) * 4 + s);  // statistics.py:75-83
                        }
                        if (!m && s <= SYM_GAP && r <= SYM_GAP && r != s) {  // statistics.py:26-35
                            const int row = b_mis + (side * L + p) * 25;
                            if (r != SYM_GAP) bump<USE_LDS>(lds, raw, row + r);
                            bump<USE_LDS>(lds, raw, row + c_col[r * 5 + s]);
                        }
                    }
                }
            } else {
                const int s_ncols = rl(ncols, j);
                const int s_nrg = s_n0 + rl(nI, j);
                const i64 s_co = rl64(cig_o, j);
                const int s_cn = rl(cig_n, j);
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
                        const u32 cg = a.cigar[s_co + k];
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
                    int s = qi < 0 ? SYM_GAP : classify_ascii(sp[qi]);
                    int r = rix < 0 ? SYM_GAP : (int)rp[rix];
                    if (hasq) {
                        const bool ms = qi >= 0 && (int)qp[qi] < a.minqual;
                        bool mr = ms;
                        if (jr != js) {
                            // mask of the *reference* column jr follows the read column jr
                            mr = false;
                            if (jr < s_ncols) {
                                int c2 = 0, q2 = 0, qj = -2;
                                for (int k = 0; k < s_cn && qj == -2; k++) {
                                    const u32 cg = a.cigar[s_co + k];
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
                        const int row = b_mis + (side * L + i) * 25;
                        if (r != SYM_GAP) bump<USE_LDS>(lds, raw, row + r);
                        if (r != s) bump<USE_LDS>(lds, raw, row + c_col[r * 5 + s]);
                    }
                }
                // read composition on the ungapped, unmasked query (statistics.py:75-83)
                const int Lq = s_nq < L ? s_nq : L;
                for (int t = lane; t < 2 * Lq; t += 64) {
                    const int side = t >= Lq;
                    const int k0 = side ? t - Lq : t;
                    const int s = classify_ascii(sp[side ? s_nq - 1 - k0 : k0]);
                    if (s < 4) bump<USE_LDS>(lds, raw, b_comp + (side * (L + A) + k0) * 4 + s);
                }
                // flanks (statistics.py:85-93)
                for (int t = lane; t < 2 * A; t += 64) {
                    const int side = t >= A;
                    const int dist = (side ? t - A : t) + 1;
                    if (dist <= (side ? s_nafter : s_nbefore)) {
                        const int r = side ? rp[s_n0 - 1 + dist] : rp[-dist];
                        if (r < 4) bump<USE_LDS>(lds, raw, b_comp + (side * (L + A) + L + dist - 1) * 4 + r);
                    }
                }
            }
        }
    }

    if (USE_LDS) {
        __syncthreads();
        u32 *out = a.partials + (i64)blockIdx.x * d.w_total;
        for (i64 i = threadIdx.x; i < d.w_total; i += MDX_BLOCK) out[i] = lds[i];
    }
}

hipError_t mdx_k_prepare(size_t lds_bytes) {
    hipError_t e;
    e = hipFuncSetAttribute((const void *)tabulate_kernel<true, false>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void *)tabulate_kernel<true, true>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

void mdx_k_tabulate(const MdxTabArgs &a, int mode, bool mask, int grid, size_t lds_bytes, hipStream_t s) {
    if (a.n_reads <= 0) return;
    if (mode == MDX_MODE_LDS) {
        if (mask) hipLaunchKernelGGL((tabulate_kernel<true, true>), dim3(grid), dim3(MDX_BLOCK), lds_bytes, s, a);
        else hipLaunchKernelGGL((tabulate_kernel<true, false>), dim3(grid), dim3(MDX_BLOCK), lds_bytes, s, a);
    } else {
        if (mask) hipLaunchKernelGGL((tabulate_kernel<false, true>), dim3(grid), dim3(MDX_BLOCK), 0, s, a);
        else hipLaunchKernelGGL((tabulate_kernel<false, false>), dim3(grid), dim3(MDX_BLOCK), 0, s, a);
    }
}

// raw[w] += sum over block slots (coalesced across w)
__global__ void reduce_partials_kernel(const u32 *__restrict__ partials, u64 *__restrict__ raw, i64 w_total,
                                       int grid) {
    const i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= w_total) return;
    u64 acc = 0;
    for (int b = 0; b < grid; b++) acc += partials[(i64)b * w_total + w];
    if (acc) raw[w] += acc;
}

void mdx_k_reduce_partials(const uint32_t *partials, unsigned long long *raw, int64_t w_total, int grid,
                           hipStream_t s) {
    const int threads = 256;
    const int blocks = (int)((w_total + threads - 1) / threads);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(blocks), dim3(threads), 0, s, partials, raw, (i64)w_total, grid);
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
            const int rc = strand ? c_comp_col[col] : col;
            v = raw[lb + ((strand * 2 + side) * L + p) * 25 + rc];
            if (col < 4) v += raw[lb + d.off_m() + ((strand * 2 + side) * L + p) * 4 + rc];
        } else if (i < n_mis + n_comp) {
            i64 x = i - n_mis;
            const int b = x % 4; x /= 4;
            const int row = x % (L + A); x /= (L + A);
            const int strand = x % 2; x /= 2;
            const int end = x % 2; x /= 2;
            const i64 lb = x * d.w_lib;
            const int side = strand ? end : 1 - end;
            const int rb = strand ? 3 - b : b;
            // 5p rows: -A..-1 (flank, distance A-row) then 1..L (read slot row-A)
            // 3p rows: -L..-1 (read slot L-1-row) then 1..A (flank, distance row-L+1)
            int slot;
            if (end == 1) slot = row < A ? L + (A - row) - 1 : row - A;
            else slot = row < L ? L - 1 - row : L + (row - L + 1) - 1;
            v = raw[lb + d.off_comp() + ((strand * 2 + side) * (L + A) + slot) * 4 + rb];
            if (slot < L) v += raw[lb + d.off_m() + ((strand * 2 + side) * L + slot) * 4 + rb];
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
