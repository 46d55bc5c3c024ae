// Internal interface between the C-ABI layer (mdx_capi.cpp) and the gfx950 kernels
// (mdx_kernels.hip).  Not installed; the public boundary is include/mdx.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// Raw (reference-orientation) accumulator layout, in words, per library:
//   MIS [strand 2][side 2][L][25]   rare events: substitutions, indels, soft clips; base columns
//                                    0..3 (A,C,T,G order = (ascii >> 1) & 3) count the matching
//                                    columns of gapped records
//   CMP [strand 2][side 2][L][4]    read-base counts of columns that are not plain matches
//   TC  [strand 2][base 4][256]     the common case of plain (ungapped, complete) records: read base ==
//                                    reference base, or an A/C/G/T flank base.  One lane of the
//                                    wavefront owns four consecutive bytes (one dword) of a record:
//                                      lanes [0, nl4)            left-anchored columns 4m .. 4m+3
//                                      lanes [nl4, 2 nl4)        right-anchored columns 4m+3 .. 4m (byte order)
//                                      lanes [2 nl4, +nf4)       left flank, distances 4(m+1) .. 4m+1
//                                      lanes [2 nl4 + nf4, +nf4) right flank, distances 4m+1 .. 4m+4
//                                    and the table index of (lane, byte j) is tau = 64 j + lane, so the
//                                    64 increments of one ds_add_u32 fall in 64 consecutive words.
//   LGD [kind 2][strand 2][lgd_lds] short fragment lengths
// followed, after the last library, by one word: number of kept reads.
// side 0 = left-anchored (columns counted from the leftmost reference coordinate),
// side 1 = right-anchored.  The canonical 5p/3p tables are a fixed permutation of these
// (strand '+': 5p = left, 3p = right; strand '-': swapped and complemented), applied once by
// finalize_kernel.
struct MdxDims {
    int L, A, nlib, lgd_max, lgd_lds;
    int nl4, nf4, apad;   // dword lanes per side, per flank; 4 * nf4
    int t_pad;            // 256
    int w_mis, w_cmp, w_tc, w_lgd, w_lib;
    int64_t w_total;      // nlib * w_lib + 1
    __host__ __device__ int off_cmp() const { return w_mis; }
    __host__ __device__ int off_tc() const { return w_mis + w_cmp; }
    __host__ __device__ int off_lgd() const { return w_mis + w_cmp + w_tc; }
    __host__ __device__ int tau_left(int p) const { return 64 * (p & 3) + (p >> 2); }
    __host__ __device__ int tau_right(int p) const { return 64 * (3 - (p & 3)) + nl4 + (p >> 2); }
    __host__ __device__ int tau_lflank(int dist) const {
        if (nl4 == 0) return dist - 1;  // no fast path: flank tasks numbered densely
        const int m = (dist - 1) >> 2;
        return 64 * (4 * (m + 1) - dist) + 2 * nl4 + m;
    }
    __host__ __device__ int tau_rflank(int dist) const {
        if (nl4 == 0) return A + dist - 1;
        const int m = (dist - 1) >> 2;
        return 64 * (dist - 1 - 4 * m) + 2 * nl4 + nf4 + m;
    }
    // the dword fast path needs all its lanes in one wavefront and its flank window in the guard band
    __host__ __device__ bool fast_ok() const { return nl4 > 0 && apad <= 248 && L + A <= 248; }
};

static inline MdxDims mdx_make_dims(int L, int A, int nlib, int lgd_max, int lgd_lds) {
    MdxDims d;
    d.L = L; d.A = A; d.nlib = nlib; d.lgd_max = lgd_max; d.lgd_lds = lgd_lds;
    d.nl4 = (L + 3) / 4;
    d.nf4 = (A + 3) / 4;
    d.apad = 4 * d.nf4;
    d.t_pad = 256;
    if (2 * d.nl4 + 2 * d.nf4 > 64 || d.apad > 248 || L + A > 248) {  // no fast path
        d.nl4 = 0; d.nf4 = 0;
        d.t_pad = ((2 * A + 63) / 64) * 64;
        if (d.t_pad == 0) d.t_pad = 64;
    }
    d.w_mis = 2 * 2 * L * 25;
    d.w_cmp = 2 * 2 * L * 4;
    d.w_tc = 2 * 4 * d.t_pad;
    d.w_lgd = 2 * 2 * lgd_lds;
    d.w_lib = d.w_mis + d.w_cmp + d.w_tc + d.w_lgd;
    d.w_total = (int64_t)nlib * d.w_lib + 1;
    return d;
}

struct MdxTabArgs {
    // batch (device pointers)
    int64_t n_reads;
    const uint16_t *flag;
    const uint16_t *lib;
    const int32_t *tid;
    const int32_t *pos;
    const int32_t *tlen;
    const uint32_t *cigar_off;
    const uint32_t *cigar;
    const uint32_t *seq_off;
    const uint8_t *seq;
    const uint8_t *qual;
    // resident reference: one symbol class per base (0..3 ACGT, 4 '-', 5 other)
    const uint8_t *ref;
    const int64_t *contig_off;
    int n_contig;
    int minqual;
    MdxDims dims;
    // accumulators
    uint32_t *partials;              // [grid][w_total] (LDS mode)
    unsigned long long *raw;         // [w_total] u64 (global-atomic mode writes here directly)
    unsigned long long *lgd_dense;   // [nlib][2][2][lgd_max]
    long long *lgd_over;             // [cap][4]
    long long lgd_over_cap;
    unsigned long long *n_lgd_over;
    unsigned long long *err;         // min over (read_index << 8 | -code); ~0 = no error
    int queue_off;                   // word offset of the per-wave rare-event queues in the LDS
    int ref32;                       // reference (with guard bands) shorter than 4 GiB: 32-bit window offsets
};

enum { MDX_MODE_LDS = 0, MDX_MODE_GLOBAL = 1 };

int mdx_k_block_threads();
size_t mdx_k_lds_bytes(const MdxDims &d);
int mdx_k_queue_off(const MdxDims &d);
hipError_t mdx_k_prepare(size_t lds_bytes);
void mdx_k_encode_ref(const uint8_t *d_ascii, uint8_t *d_codes, int64_t n, hipStream_t s);
void mdx_k_tabulate(const MdxTabArgs &a, int mode, bool mask, int grid, size_t lds_bytes, hipStream_t s);
void mdx_k_reduce_partials(const uint32_t *partials, unsigned long long *raw, int64_t w_total, int grid,
                           hipStream_t s);
void mdx_k_finalize(const unsigned long long *raw, const unsigned long long *lgd_dense,
                    const unsigned long long *n_lgd_over, MdxDims d, unsigned long long *out,
                    hipStream_t s);
void mdx_k_genome_comp(const uint8_t *ref, const int64_t *contig_off, int n_contig, unsigned long long *out,
                       hipStream_t s);

struct MdxRescaleArgs {
    int64_t n_reads;
    const uint16_t *flag;
    const int32_t *tid, *pos, *mtid, *mpos;
    const uint32_t *cigar_off, *cigar, *seq_off;
    const uint8_t *seq, *qual;
    const uint8_t *ref;
    const int64_t *contig_off;
    int n_contig;
    const uint8_t *lut;      // [2][npos][94]
    const double *term;      // [2][npos]
    int len5p, len3p;
    uint8_t *qual_out;
    double *mr_raw;
    uint8_t *status;
    unsigned long long *err;
};
void mdx_k_rescale(const MdxRescaleArgs &a, int grid, hipStream_t s);
