// Internal interface between the C-ABI layer (mdx_capi.cpp) and the gfx950 kernels
// (mdx_kernels.hip).  Not installed; the public boundary is include/mdx.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

// Raw (reference-orientation) accumulator layout, in words, per library (order: TC, MIS, CMP, LGD):
//   MIS [strand 2][side 2][L][25]   rare events: substitutions, indels, soft clips; base columns
//                                    0..3 (A,C,T,G order = (ascii >> 1) & 3) count the matching
//                                    columns of gapped records
//   CMP [strand 2][side 2][L][4]    read-base counts of columns that are not plain matches
//   TC  [strand 2][base 4][512]     the common case of plain (ungapped) records: read base == reference
//                                    base, or an A/C/G/T flank base.  One lane of the wavefront owns
//                                    eight consecutive bytes of a record's window, flank and columns
//                                    merged (the left flank is contiguous with the left columns in
//                                    the reference, the right flank with the right columns):
//                                      lanes [0, nl8)      left side:  window bytes b = 8m + j, b = p + A
//                                                          (p < 0: left flank at distance -p, else column p)
//                                      lanes [nl8, 2 nl8)  right side: e = 8m + 7 - j counted back from
//                                                          aend + A (e < A: right flank at distance A - e,
//                                                          else right-anchored column e - A)
//                                    G = 2 nl8 lanes per record, so one wavefront step counts R = 64 / G
//                                    records (slot g = lane / G); the table index of (lane, byte j) is
//                                    tau = 64 j + lane: the 64 increments of one ds_add_u32 fall in 64
//                                    consecutive words, and every slot has its own words.
//   DMP [strand 2][side 2][A + L]   fast path only: difference-encoded count, by window byte (left side: b = p + A;
//                                    right side: e, as above), of the bytes that steps of partial and single-indel
//                                    records put into plane A of TC although they are not tasks (or are counted
//                                    elsewhere): such a step zeroes the reference bytes outside its tasks and counts
//                                    all eight bytes of a lane with one increment, like a complete record's step;
//                                    a record adds +1 where such a stretch of its window begins and -1 where it ends
//                                    (phase 1), finalize_kernel subtracts the prefix sums from the A counts.
//   LGD [kind 2][strand 2][lgd_lds] short fragment lengths
// followed, after the last library, by one word: number of kept reads.
// side 0 = left-anchored (columns counted from the leftmost reference coordinate),
// side 1 = right-anchored.  The canonical 5p/3p tables are a fixed permutation of these
// (strand '+': 5p = left, 3p = right; strand '-': swapped and complemented), applied once by
// finalize_kernel.
struct MdxDims {
    int L, A, nlib, lgd_max, lgd_lds;
    int nl8;              // 8-byte lanes per side (0: no fast path)
    int G, R;             // lanes per record, records per wavefront step
    // The packed kernel's own geometry (tabulate_kernel<.., PK>): a lane owns sixteen consecutive window bases — eight bytes of
    // 4-bit codes —, a record takes G4 = 2 nl16 lanes, and the slots of a step are tied to a strand: slots [0, H4) take
    // forward-strand records, [H4, 2 H4) reverse-strand ones, so that a lane's counters belong to one strand.  Its TC
    // table in the LDS is [base 4][16 j + ... : 64 j + lane], the same 4096 words; the block's partial slot receives it in
    // the layout above (slot 0), so nothing downstream knows.
    int nl16, G4, H4;
    int t_pad;            // words per TC plane: 512 with the fast path
    int w_mis, w_cmp, w_mc, w_tc, w_dmp, w_lgd, w_lib;
    int64_t w_total;      // nlib * w_lib + 1
    // word offsets within a library: TC first (256-byte aligned planes; the optimistic increments of a gapped record
    // behind a deletion address MIS / CMP rows by position — for bytes that are not tasks the position can be a few
    // rows below row 0: such adds of 0 then fall into the TC words instead of outside the table)
    __host__ __device__ int off_tc() const { return 0; }
    __host__ __device__ int off_mis() const { return w_tc; }
    __host__ __device__ int off_cmp() const { return w_tc + w_mis; }
    __host__ __device__ int off_dmp() const { return w_tc + w_mc; }
    __host__ __device__ int off_lgd() const { return w_tc + w_mc + w_dmp; }
    // task -> TC index of slot 0 (slot g adds g * G)
    __host__ __device__ int tau_left(int p) const { const int b = p + A; return 64 * (b & 7) + (b >> 3); }
    __host__ __device__ int tau_right(int p) const { const int e = p + A; return 64 * (7 - (e & 7)) + nl8 + (e >> 3); }
    __host__ __device__ int tau_lflank(int dist) const {
        if (nl8 == 0) return dist - 1;  // no fast path: flank tasks numbered densely
        const int b = A - dist;
        return 64 * (b & 7) + (b >> 3);
    }
    __host__ __device__ int tau_rflank(int dist) const {
        if (nl8 == 0) return A + dist - 1;
        const int e = A - dist;
        return 64 * (7 - (e & 7)) + nl8 + (e >> 3);
    }
    // the 8-byte-lane fast path needs at least one record per wavefront and its window in the guard band
    __host__ __device__ bool fast_ok() const { return nl8 > 0; }
};

#define MDX_MAX_R 4       // records per wavefront step (staging pad = R - 1 entries)
// blocks of a pool (the fast kernels' hand-out of tiles): the grid in pools of that many blocks each where it divides
#ifndef MDX_POOL_BLOCKS
#define MDX_POOL_BLOCKS 2
#endif
static inline __host__ __device__ unsigned mdx_n_pools(unsigned grid) {
    return (grid >= MDX_POOL_BLOCKS && grid % MDX_POOL_BLOCKS == 0u) ? grid / MDX_POOL_BLOCKS : ((grid >= 2u && !(grid & 1u)) ? grid / 2u : grid);
}
// words between two tile counters: a counter per 128-byte line — atomics on one line queue at the L2, whichever word they add to
#ifndef MDX_CTR_PAD
#define MDX_CTR_PAD 32
#endif
#define MDX_CTR_WORDS 262144   // the counters' buffer (MdxTabArgs::tile_ctr)
#define MDX_ML_MAX_LIBS 64     // libraries of a context the packed kernels count in one launch (six bits of a staging entry)
#ifndef MDX_POOL_CHUNK
#define MDX_POOL_CHUNK 24  // consecutive tiles a pool of two blocks takes at a time (the fast kernels' hand-out of tiles)
#endif
// staging entries (16 B) per wavefront: a tile of 64 - 64 % R records and the R - 1 entries that pad its last step
static inline __host__ __device__ int mdx_stage_entries(const MdxDims &d) {
    return d.R > 0 ? 64 - 64 % d.R + d.R - 1 : 64;
}
// copies of the dense fragment-length histogram (lengths >= lgd_lds take global atomics: a block adds to copy
// blockIdx & (copies - 1), finalize_kernel sums them) — a paired-end library with 350 bp inserts put every
// second record on a few hundred words of a single copy: 0.73 ms instead of 0.13 ms per 2 M records
#define MDX_LGD_COPIES 32

static inline MdxDims mdx_make_dims(int L, int A, int nlib, int lgd_max, int lgd_lds) {
    MdxDims d;
    d.L = L; d.A = A; d.nlib = nlib; d.lgd_max = lgd_max; d.lgd_lds = lgd_lds;
    d.nl8 = (L + A + 7) / 8;
    d.G = 2 * d.nl8;
    d.t_pad = 512;
    d.nl16 = (L + A + 15) / 16;
    d.G4 = 2 * d.nl16;
    d.H4 = d.G4 > 0 ? 32 / d.G4 : 0;
    if (d.H4 > 3) d.H4 = 3;
    if (d.G > 64 || L + A > 248) {  // no fast path
        d.nl8 = 0; d.G = 0; d.R = 0;
        d.t_pad = ((2 * A + 63) / 64) * 64;
        if (d.t_pad == 0) d.t_pad = 64;
    } else {
        d.R = 64 / d.G < MDX_MAX_R ? 64 / d.G : MDX_MAX_R;
    }
    d.w_mis = 2 * 2 * L * 25;
    d.w_cmp = 2 * 2 * L * 4;
    d.w_mc = d.w_mis + d.w_cmp;
    d.w_tc = 2 * 4 * d.t_pad;
    d.w_dmp = d.nl8 > 0 ? 2 * 2 * (A + L) : 0;
    d.w_lgd = 2 * 2 * lgd_lds;
    d.w_lib = (d.w_tc + d.w_mc + d.w_dmp + d.w_lgd + 63) / 64 * 64;
    d.w_total = (int64_t)nlib * d.w_lib + 1;
    return d;
}

// The fused tabulate + rescale launch (mdx_tabulate_rescale_device, BASELINE configs[4]): what the tabulation kernel
// needs to rescale the records of its own tile loop — [S] M [S] records of at most 2 L aligned bases that
// mapdamage/rescale.py:300-342 routes to _rescale_qual_read — while it counts them.  Every other record that wants
// rescaling is appended to the wavefront's list (gen_list / gen_count, list_cap entries per wavefront) for
// rescale_kernel and rescale_walk_kernel behind it.
struct MdxFuse {
    const int32_t *mtid, *mpos;
    uint8_t *qual_out;          // becomes a copy of the batch's quality column, tile by tile, then takes the rescaled bytes
    double *mr_raw;
    uint8_t *status;
    const uint8_t *lut;         // [2][npos][94]
    const double *term;         // [2][npos]
    int len5p, len3p;
    uint32_t *subs_part;        // [grid][752 + 2 npos 94]: the block's own summary row (zeroed by the block itself)
    uint32_t *gen_list, *gen_count;
    // patch mode (patch not null; qual_out is then null): the quality bytes that change are appended to a list instead of
    // being stored into a copy of the column — entry = index of the byte in the quality column | new Phred << 32 — and
    // *n_patch counts them (it may pass patch_cap: the entries beyond are dropped, the caller sees the count)
    // The list comes in patch_parts parts (a power of two), patch_cap entries and a counter each — a thousand wavefronts
    // appending to ONE list queue at one address of the L2 (2.7 M appends of a 25 M-record launch: 3 ms): a block appends
    // to part blockIdx & (patch_parts - 1).
    unsigned long long *patch, *n_patch;
    long long patch_cap;
    int patch_parts;
    int qcap;                   // events of a wavefront's queue (mdx_k_fuse_qcap)
    int tcb_off;                // word offset in the LDS of the second TC table (the fused records' own), then 4 words of
                                // reference-base counts, the lookup table and the terms (mdx_k_fuse_lds_bytes)
};

// A wavefront's part of MdxTabArgs::lists, in 16-byte entries — rings, the same size whatever the batch (tabulate_kernel):
// four lists of MDX_LIST_RING staging entries (partial records, single insertions, single deletions, the complete records the
// general pass finds), MDX_DRING records waiting for the general pass (two entries of columns and an index each), and three
// rings of record indices for the fused kernels
#define MDX_LIST_RING 1024
#define MDX_ROUND_TILES 14      // a round appends its records (63 x 14) and the < 64 that waited for the general pass, < 63 entries are left over: <= MDX_LIST_RING
#define MDX_DRING 128
#define MDX_WAVE_SCRATCH(ring) (4 * (ring) + 2 * MDX_DRING + MDX_DRING / 4 + 3 * ((ring) / 4))
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
    // ... and its 4-bit form (MDX_SEQ_4BIT codes, two bases per byte, the 256-base guard bands included: nibble i is
    // genome coordinate i - 256), read by the packed kernel (tabulate_kernel<.., PK>) together with a 4-bit SEQ column
    const uint8_t *ref4;
    // ref2 (a large genome): a SECOND copy of the 4-bit reference at ref4 + 2 GiB + 64 bytes — half a 128-byte line out of
    // phase.  What a random access to the reference costs the memory system is the lines it touches, not the bytes (55 G lines/s
    // chip-wide whatever a window's width, tools/experiments/randwin.hip); the window of a 100-base record — 68 bytes — straddles
    // two lines 47 % of the time in one copy and 3 % of the time in the better of the two.  Phase 1 of the packed kernels picks
    // the copy per record (bit 31 of a complete record's staging word w: a byte offset of 2 GiB, or'ed to the lane's).
    int ref2;
    int seq_packed;                  // the batch's seq column holds MDX_SEQ_4BIT codes (include/mdx.h)
    const int64_t *contig_off;
    int n_contig;
    int minqual;
    MdxDims dims;
    // accumulators
    uint32_t *partials;              // [grid][w_total] (LDS mode)
    unsigned long long *raw;         // [w_total] u64 (global-atomic mode writes here directly)
    unsigned long long *lgd_dense;   // [MDX_LGD_COPIES][nlib_total][2][2][lgd_max] (+ the launch's first library)
    long long *lgd_over;             // [cap][4]
    long long lgd_over_cap;
    unsigned long long *n_lgd_over;
    unsigned long long *err;         // min over ((record_base + read_index) << 8 | -code); ~0 = no error
    long long record_base;           // index of the batch's first record in the caller's numbering (mdx_set_record_base)
    int stage_off;                   // word offset of the per-wave record staging areas in the LDS
    int queue_off;                   // word offset of the per-wave rare-event queues in the LDS
    int pfl_off;                     // the packed kernels: word offset of the per-wave areas the next tile's phase-1 columns are
                                     // prefetched into (MDX_PFL_WAVE_BYTES each; 0: none — the columns come by plain loads)
    int ref32;                       // reference (with guard bands) shorter than 4 GiB: 32-bit window offsets
    int64_t n_bases;                 // bytes in seq (and qual): bounds the speculative 8-byte loads
    int lib_lo, nlib_total;          // this launch counts libraries [lib_lo, lib_lo + dims.nlib) of nlib_total
    // Per-wavefront lists: wavefront w owns MDX_WAVE_SCRATCH(ring_size) 16-byte entries (rings: see above and the kernel).
    // Written in the tile loop, read back by the same wavefront at the end of its round.  ring_size (a power of two): MDX_LIST_RING —
    // 81 KB per wavefront whatever the batch — for the kernels that work in rounds; for the fused kernels and the masked
    // launches, which do not, what a wavefront's tile_quota of tiles can append.
    uint4 *lists;
    int ring_size;
    // (the fused kernels: entries of a wavefront's list of records left to the rescale kernels, MdxFuse::gen_list — the
    // records of tile_quota tiles)
    int64_t list_cap;
    MdxFuse rs;                      // used by the fused kernel only
    // Fast kernels: tiles are handed out on demand within pools of two blocks (the two that share a CU): one counter per
    // pool, zeroed before the launch; a wavefront takes at most tile_quota tiles (the fused kernels and the masked kernel: their rings, and the fused kernels' list_cap, hold the records of that many; the others: no limit)
    uint32_t *tile_ctr;
    const uint4 *ml_plan;            // launches over several libraries: per pool {library, place among its pools, their number, the first}
    int tile_quota;
    int round_tiles;                 // tiles of a round (MDX_ROUND_TILES; the list rings hold a round's entries)
    // Several libraries in one launch of the packed kernel (tabulate_kernel<.., PK, ML>; n_libs > 0): the batch above is the
    // copy of mdx_libsort.hip, ordered by library — place i holding one kept record (the flag filter of reader.py:121-132
    // applied on the way), the records of library l at places [lib_start[l], lib_start[l + 1]) in batch order, CIGAR, SEQ and
    // low-quality bitmap in the same order; perm = a record's index within the caller's batch (for the error word).  The
    // launch counts the libraries [lib_lo, lib_lo + n_libs), every pool of blocks one of them (ml_plan); dims are one
    // library's; partials holds a slot per block, tile_ctr a counter per pool, as in a one-library launch.
    const uint32_t *perm, *lib_start;
    const unsigned long long *sort_bad;     // MdxLibSort::bad
    int n_libs;
#ifdef MDX_WAVE_CLK
    // instrumented builds (-DMDX_WAVE_CLK, tools/experiments/wave_clk.py): three clock readings per wavefront — start,
    // end of the tile loop, end — read back with mdx_dbg_clk_read
    unsigned long long *dbg_clk;
#endif
};
// events (20 bytes: a lane's sixteen read and reference nibbles and a word) of the packed kernel a wavefront's LDS queue
// holds (MDX_PK_EVQ_BYTES of mdx_kernels.hip); at least 64: the events of a step fit an empty queue
#ifndef MDX_PK_QCAP
#define MDX_PK_QCAP 112
#endif

enum { MDX_MODE_LDS = 0, MDX_MODE_GLOBAL = 1 };

int mdx_k_block_threads();
size_t mdx_k_lds_bytes(const MdxDims &d);
int mdx_k_stage_off(const MdxDims &d);
int mdx_k_queue_off(const MdxDims &d);
hipError_t mdx_k_prepare(size_t lds_bytes);
// the fused tabulate + rescale kernel: one 1024-thread block per CU (its LDS image holds a second TC table and the
// rescale model), queue offset and size of its image, the word offset of the second TC table in it
int mdx_k_fuse_block_threads();
int mdx_k_fuse_queue_off(const MdxDims &d);
int mdx_k_fuse_tcb_off(const MdxDims &d, int qcap);
size_t mdx_k_fuse_lds_bytes(const MdxDims &d, int npos, int qcap);
// events of the fused kernel's queue per wavefront: what the LDS has room for, between 64 and 160 (even)
static inline int mdx_k_fuse_qcap(const MdxDims &d, int npos, size_t lds_limit) {
    int q = 160;
    while (q > 64 && mdx_k_fuse_lds_bytes(d, npos, q) > lds_limit) q -= 2;
    return q;
}
hipError_t mdx_k_fuse_prepare(size_t lds_bytes);
// the packed fused kernel (4-bit SEQ column and reference, one library): one 1024-thread block per CU too, no second TC
// table; MdxFuse::tcb_off = mdx_k_pkf_tcb_off, the queue offset is mdx_k_fuse_queue_off, MdxFuse::qcap unused
int mdx_k_pkf_tcb_off(const MdxDims &d);
size_t mdx_k_pkf_lds_bytes(const MdxDims &d, int npos);
hipError_t mdx_k_pkf_prepare(size_t lds_bytes);
void mdx_k_tabulate_packed_fused(const MdxTabArgs &a, int grid, size_t lds_bytes, hipStream_t s);
// the SEQ stretches of the records of n_in lists (in_list[l * in_cap ..], in_count[l]) from the 4-bit column to ASCII at the
// same offsets of `out` (n_bases + 8 bytes at least)
void mdx_k_unpack_listed(const uint32_t *in_count, const uint32_t *in_list, int64_t in_cap, int n_in, const uint32_t *seq_off,
                         const uint8_t *seq4, uint8_t *out, int64_t n_bases, hipStream_t s);
void mdx_k_encode_ref(const uint8_t *d_ascii, uint8_t *d_codes, int64_t n, hipStream_t s);
// resident reference bytes (guard bands included, n even) -> 4-bit codes, n / 2 bytes
void mdx_k_encode_ref4(const uint8_t *d_codes, uint8_t *d_ref4, int64_t n, hipStream_t s);
// SEQ columns between the two forms of mdx_batch::seq (n bases; the packed column holds (n + 1) / 2 bytes)
void mdx_k_pack_seq(const uint8_t *d_ascii, uint8_t *d_packed, int64_t n, hipStream_t s);
// MDX_FLAG_HAS_QUAL set on the records of a resident batch whose first quality byte is not 0xFF
void mdx_k_mark_has_qual(uint16_t *d_flag, const uint32_t *d_seq_off, const uint8_t *d_qual, int64_t n, hipStream_t s);
void mdx_k_unpack_seq(const uint8_t *d_packed, uint8_t *d_ascii, int64_t n, hipStream_t s);
// The packed kernels' block and LDS image (the staging offset is mdx_k_stage_off).  Their registers allow four wavefronts per
// SIMD, and the launch puts them into ONE block of 1024 threads per CU (rounds 4-5: two of 512): one image of the tables
// instead of two, and what that frees of the LDS gives every wavefront an area (MDX_PFL_WAVE_BYTES) into which the phase-1
// columns of its NEXT tile are prefetched with LDS-DMA loads (global_load_lds_*: no registers) under the current tile's work.
// Where the tables of a long --length leave less room the block is 512 or 256 threads (MdxPkConfig, mdx_k_pk_config); the
// fused kernel (tabulate + rescale) has an image of its own and no such areas (MdxTabArgs::pfl_off = 0).
#define MDX_PFL_COLS 12                          // flag, library, tid, pos, tlen, cigar_off, seq_off; three operations, two contig bounds
#define MDX_PFL_WAVE_BYTES (MDX_PFL_COLS * 256)  // a dword per lane and column
struct MdxPkConfig { int threads; int pfl; };
MdxPkConfig mdx_k_pk_config(const MdxDims &d, size_t lds_limit);
int mdx_k_pk_blocks_per_cu(int threads);     // by its registers
int mdx_k_pk_queue_off(const MdxDims &d, int threads);
int mdx_k_pk_pfl_off(const MdxDims &d, int threads);
size_t mdx_k_pk_lds_bytes(const MdxDims &d, const MdxPkConfig &k);
// records of a tile of the fast kernels: a multiple of R (the ASCII kernels' steps); at most 63 where the columns are
// prefetched (a record's end offsets are its neighbour's start offsets: lane 63 brings the last one)
static inline __host__ __device__ int mdx_tile_records(const MdxDims &d, bool pfl) {
    const int t = d.R > 0 ? 64 - 64 % d.R : 64;
    return pfl && t > 63 ? 63 : t;
}
hipError_t mdx_k_prepare_packed(size_t lds_bytes);
// ... with --min-basequal (a MDX_SEQ_4BITQ column: the mask is in the nibbles)
hipError_t mdx_k_prepare_packed_masked(size_t lds_bytes);
void mdx_k_tabulate_packed_masked(const MdxTabArgs &a, int grid, int threads, size_t lds_bytes, hipStream_t s);
// MDX_SEQ_4BIT -> MDX_SEQ_4BITQ: the bases whose quality is below minqual (qual, or the bitmap lowq if not null) complemented;
// seq4_out may be seq4_in
void mdx_k_fold_mask(const uint8_t *seq4_in, uint8_t *seq4_out, const uint8_t *qual, const uint8_t *lowq, int64_t n_bases, int minqual,
                     hipStream_t s);
void mdx_k_tabulate_packed(const MdxTabArgs &a, int grid, int threads, size_t lds_bytes, hipStream_t s);
void mdx_k_tabulate(const MdxTabArgs &a, int mode, bool mask, int grid, size_t lds_bytes, hipStream_t s);
void mdx_k_tabulate_fused(const MdxTabArgs &a, int grid, size_t lds_bytes, hipStream_t s);
// (tile_ctr, if not null and w_total >= 4096: 4096 words zeroed on the way — the next launch's tile counters)
// (n_lib > 1: a launch over several libraries — `plan` says which library a block's slot holds; each library is summed into its
// own stretch of raw, lib_stride words apart)
void mdx_k_reduce_partials(const uint32_t *partials, unsigned long long *raw, unsigned long long *raw_tail,
                           int64_t w_total, int grid, hipStream_t s, uint32_t *tile_ctr = nullptr, int n_lib = 1, int64_t lib_stride = 0,
                           const void *plan = nullptr);
// the pools of a launch over several libraries dealt to the libraries (MdxTabArgs::ml_plan: n_pools x uint4)
void mdx_k_ml_plan(const uint32_t *lib_start, int lib_lo, int nlib, int T, int grid, void *plan, hipStream_t s);

// ---- bucketing a batch's records by library (mdx_libsort.hip), for the packed kernel's launches over several libraries
struct MdxLibSort {
    unsigned long long *bad; // min over index << 8 | code of the kept records whose library is not below nlib (~0: none): they have
                             // no place, and every launch over the bucketed columns reports the first of them (MdxTabArgs::sort_bad)
    uint32_t *lib_start;     // [nlib + 1]; lib_start[nlib] = the kept records
    uint32_t *perm;          // place -> index within the batch
    uint16_t *flag;
    int32_t *tid, *pos, *tlen;
    uint32_t *cigar_off, *seq_off;   // [kept + 1]
    uint32_t *cigar;
    uint8_t *seq;            // 4-bit codes, (n_bases + 1) / 2 + 64 bytes, what lies behind the last kept base zero
    size_t seq_bytes;
};
// bytes of the blob that holds all of it for a batch of n records (n_cigar operations, n_bases bases), and the pointers into one
size_t mdx_k_libsort_bytes(int64_t n, int64_t n_cigar, int64_t n_bases, int nlib);
void mdx_k_libsort_layout(void *blob, int64_t n, int64_t n_cigar, int64_t n_bases, int nlib, MdxLibSort *out);
// scratch of one sort
size_t mdx_k_libsort_scratch_bytes(int64_t n, int nlib);
// a batch (device columns, 4-bit SEQ) -> out
void mdx_k_libsort(int64_t n, int64_t n_cigar, int64_t n_bases, const uint16_t *flag, const uint16_t *lib, const int32_t *tid,
                   const int32_t *pos, const int32_t *tlen, const uint32_t *cigar_off, const uint32_t *cigar, const uint32_t *seq_off,
                   const uint8_t *seq4, int nlib, void *scratch, const MdxLibSort &out, hipStream_t s);
void mdx_k_finalize(const unsigned long long *raw, const unsigned long long *lgd_dense,
                    const unsigned long long *n_lgd_over, MdxDims d, unsigned long long *out,
                    hipStream_t s);
void mdx_k_genome_comp(const uint8_t *ref, const int64_t *contig_off, int n_contig, unsigned long long *out,
                       hipStream_t s);

struct MdxRescaleArgs {
    int64_t n_reads;
    int64_t n_bases;         // bytes in seq / qual / qual_out
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
    int key0_plain;          // position key 0 (columns outside both end windows) is the identity in lut and 0.0 in term
    int lds_tables;          // set by mdx_k_rescale: lut, term and the summary counters live in the LDS
    uint8_t *qual_out;       // becomes a copy of qual (by rescale_kernel, tile by tile, or by a device copy in front of the walk
                             // kernel when that runs alone); the kernels then store the rescaled bytes only
    unsigned long long *patch, *n_patch;   // patch mode, as in MdxFuse (qual_out null: nothing is copied, nothing stored)
    long long patch_cap;
    int patch_parts;
    double *mr_raw;
    uint8_t *status;
    unsigned long long *err;
    unsigned long long *subs;   // summary counters (rescale.py:108-192), may be null:
                                // [4 reference bases | 4 transitions x before/after x 94 | 2 x npos x 94]; of the middle
                                // part the kernel fills only the "before" words of T>C and A>G — the rest follows from the
                                // last part and the LUT (mdx_rescale_summary)
    uint32_t *subs_part;        // [blocks][752 + 2 npos 94]: the blocks' own counters (mdx_k_rescale_part_bytes), summed
                                // into subs by a second small kernel
    // records rescale_kernel leaves to rescale_walk_kernel: wavefront w appends record indices to
    // gen_list[w * gen_cap ..] and stores their number in gen_count[w] (mdx_k_rescale_lists sizes both)
    uint32_t *gen_list;
    uint32_t *gen_count;
    int64_t gen_cap;
    // list mode (behind the fused kernel): the records to take are those of n_in lists of up to in_cap indices each
    // (in_list[l * in_cap ..], in_count[l]) instead of every record of the batch; qual_out is complete already
    const uint32_t *in_list, *in_count;
    int64_t in_cap;
    int n_in;
    int row_base;               // first row of subs_part the walk kernel's blocks write (set by mdx_k_rescale)
    int copy_qual;              // set by mdx_k_rescale: rescale_kernel copies qual to qual_out tile by tile
};
void mdx_k_rescale(const MdxRescaleArgs &a, int n_cu, hipStream_t s);
// qual_out = qual with the n_patch entries of a patch list applied (qual_out may be qual: in place)
void mdx_k_rescale_expand(const uint8_t *qual, uint8_t *qual_out, int64_t n_bases, const unsigned long long *patch,
                          const unsigned long long *n_patch, long long patch_cap, int patch_parts, hipStream_t s);
// behind the fused kernel: rescale_kernel over a.in_list, the walk kernel over what that leaves, and the reduction of the
// summary rows of all three kernels (the fused kernel's `fused_rows` rows come first in subs_part)
void mdx_k_rescale_lists_pass(const MdxRescaleArgs &a, int fused_rows, int n_cu, hipStream_t s);
size_t mdx_k_rescale_part_bytes(int len5p, int len3p, int n_cu);
void mdx_k_rescale_lists(int64_t n_reads, int n_cu, int64_t *n_waves, int64_t *cap);

// (mdx_capi.cpp) the device decode's arena and CRC tables, kept by the context between files
struct mdx_ctx;
extern "C" int mdx_ctx_minqual(const mdx_ctx *c);        // the context's --min-basequal (-1: no context)
extern "C" void mdx_ctx_scratch_give(mdx_ctx *c, void *arena, size_t cap, void *tables, void *stream, void *event);
extern "C" void mdx_ctx_scratch_take(mdx_ctx *c, void **arena, size_t *cap, void **tables, void **stream, void **event);

// ---- GPU-side BAM decode (mdx_gbam.hip; host side: mdx_gbam_* in mdx_bamio.cpp)
struct MdxGbamCols {
    uint16_t *flag, *lib;
    int32_t *tid, *pos, *tlen, *mtid, *mpos;     // mtid / mpos may be null
    uint32_t *cigar_off, *cigar, *seq_off;
    uint8_t *seq, *qual;                         // qual may be null (not wanted)
    int seq_packed;                              // seq in its 4-bit form (MDX_SEQ_4BIT; the column zeroed beforehand)
    // read groups of the header: names concatenated, rg_off[n_rg + 1], library of each; lib_default: library of a
    // record without RG tag (-1: none -> 0xFFFF, which the tabulation kernel reports if the record is counted)
    const uint8_t *rg_names;
    const uint32_t *rg_off;
    const int32_t *lib_of_rg;
    int n_rg, lib_default;
    // --min-basequal (0: off; needs qual): counters[0] |= 1 when a counted record has no qualities, counters[1] != 0 when a
    // record holds a quality below the threshold
    int minqual;
    uint32_t *counters;
    // ... and a 4-bit SEQ column takes the mask into its nibbles (MDX_SEQ_4BITQ: a base whose quality is below the threshold
    // is stored as the complement of its code)
    int fold;
};
size_t mdx_k_gbam_inflate_lds();
hipError_t mdx_k_gbam_prepare();
// blk[b] = (payload offset in comp, payload bytes, offset in unc, bytes out)
void mdx_k_gbam_inflate(const uint8_t *comp, const uint4 *blk, int n_blocks, uint8_t *unc, int *status, hipStream_t s);
// want[b] = CRC32 of block b's inflated bytes (gzip trailer); tables: a device copy of mdx_crc32::Tables; *bad = min failing block
void mdx_k_gbam_crc(const uint8_t *unc, const uint4 *blk, const uint32_t *want, const void *tables, int n_blocks, int *bad, hipStream_t s);
// info[b] = (first record of segment b's chain, landing offset, status, 0), cnt[b] = (records, CIGAR operations, bases) —
// see gbam_scan_kernel; segments [first, n_blocks); forced: per segment a known first record (>= 0) or null
void mdx_k_gbam_scan(const uint8_t *unc, const uint4 *blk, const int *status, int n_blocks, int first, const int *forced,
                     uint32_t start0, uint32_t total, int n_ref, uint4 *info, uint4 *cnt, hipStream_t s);
// pre[b] = (records, CIGAR operations, bases in front of segment b, first record of its chain); cnt[b].x = its records
void mdx_k_gbam_unpack(const uint8_t *unc, const uint4 *pre, const uint4 *cnt, int n_blocks, uint32_t n_rec, uint32_t n_cig,
                       uint32_t n_seq, uint32_t *rec_off, const MdxGbamCols &c, hipStream_t s);

// ---- FASTA file -> resident reference (mdx_fasta.hip; the context's side: mdx_set_reference_fasta in mdx_capi.cpp).  The
// wanted sequences' lengths and offsets (contig_off: n_contig + 1 sums), then alloc_out(alloc_arg, total bases) = the device
// buffer base 0 of sequence 0 goes to, then the file's pieces through two staging buffers and the strip kernel on `stream`.
int mdx_fasta_to_device(const char *fasta_path, int32_t n_contig, const char *const *names, int missing_ok, int64_t *lengths,
                        std::vector<int64_t> &contig_off, std::string &err, hipStream_t stream,
                        uint8_t *(*alloc_out)(void *, int64_t), void *alloc_arg);

// ---- BGZF members on the device (mdx_gbam.hip; host side: mdx_bgzf_deflate in mdx_bamio.cpp): member b of the input = bytes
// [b * 0xFF00, ...) in mdx_k_bgzf_pieces() pieces, a lane and a slot (mdx_k_bgzf_slot_bytes()) each, sizes[piece] its bytes;
// then header, pieces, CRC32 and ISIZE of every member at offsets[b] of the output
int mdx_k_bgzf_pieces();
size_t mdx_k_bgzf_slot_bytes();
size_t mdx_k_bgzf_scratch_bytes(int n_members);
void mdx_k_bgzf_deflate(const uint8_t *d_in, long long n, int n_members, uint8_t *d_slots, uint32_t *d_sizes, void *d_scratch, hipStream_t s);
void mdx_k_bgzf_gather(const uint8_t *d_in, long long n, const uint8_t *d_slots, const uint32_t *d_sizes, const unsigned long long *d_offsets,
                       int n_members, const void *tables, uint8_t *d_out, hipStream_t s);

// ---- the rescaled records of a decoded slab written back on the device (mdx_gbam.hip; host side: mdx_gbam_write_rescaled):
// the patch list into the QUAL fields of the inflated records, the records' sizes in the output (+ 7 for an MR:f tag), the
// records with their tags to their places in the output stream; *clash = lowest rescaled record that has an MR tag already
void mdx_k_gbam_patch_qual(uint8_t *unc, const uint32_t *rec_off, const uint32_t *seq_off, uint32_t n_rec, const unsigned long long *patch,
                           const unsigned long long *n_patch, long long cap, int parts, hipStream_t s);
void mdx_k_gbam_out_sizes(const uint8_t *unc, const uint32_t *rec_off, const uint8_t *rescaled, uint32_t n_rec, uint32_t *sizes, hipStream_t s);
void mdx_k_gbam_write_back(const uint8_t *unc, const uint32_t *rec_off, const unsigned long long *out_off, const uint8_t *rescaled, const float *mr,
                           uint32_t n_rec, uint8_t *out, int *clash, hipStream_t s);
