/*
 * mdx.h — C ABI of libmdx.so, the MI355X-native damage-tabulation engine.
 *
 * The reference (ginolhac/mapDamage) has no FFI/plugin interface for this path; its seam is
 * the set of accumulator objects created in mapdamage/main.py:147-155 and the loop body
 * main.py:165-217 that feeds them.  Each entry point below names the reference interface it
 * replaces.  Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * Threading: one host thread drives one context; contexts are independent (one per GPU).
 * All work is enqueued on the context's HIP stream (mdx_set_stream lets the caller share a
 * stream with e.g. PyTorch); calls return after enqueueing unless stated otherwise.
 * Ownership: the caller owns every buffer it passes in; device memory allocated by the
 * library is released by mdx_destroy / mdx_batch_free.
 * Errors: functions return 0 (MDX_OK) or a negative code; mdx_last_error() gives text.
 */
#ifndef MDX_H
#define MDX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDX_ABI_VERSION 6   /* 6: mdx_fasta_index, mdx_set_reference_fasta, mdx_reference_fetch, mdx_host_threads, mdx_warm, mdx_*_patches_device, mdx_rescale_expand_device, mdx_mr_round, mdx_batch_fold, mdx_bgzf_deflate, mdx_gbam_rescale_slab / _write_rescaled / _record_name; 2: mdx_batch::seq_format, mdx_pack_seq, mdx_gbam_set_seq_format; 3: mdx_gbam_tell / _fixups, mdx_bam_seek; 4: mdx_batch::lowq (the struct grew by one pointer); 5: mdx_batch::libsort (another one), mdx_libsorts, mdx_gbam_view_flags / _set_flags */

#define MDX_OK 0
#define MDX_ERR_ARG (-1)          /* bad argument / unsupported configuration */
#define MDX_ERR_HIP (-2)          /* HIP runtime failure */
#define MDX_ERR_STATE (-3)        /* call order violated (e.g. tabulate before set_reference) */
#define MDX_ERR_MASK_INDEX (-4)   /* reserved, never returned: the IndexError of align.py:69-71 (a masked column beyond
                                     the gapped reference) needs a reference slice cut short by its contig end, and
                                     such a record is MDX_ERR_BAD_READ already (the reference raises at align.py:33) */
#define MDX_ERR_LGD_OVERFLOW (-5) /* more out-of-range fragment lengths than lgd_over_cap */
#define MDX_ERR_BAD_READ (-6)     /* record the reference cannot process: alignment past the
                                     contig end (pysam ValueError from align.py:33 / main.py:180),
                                     tid/library out of range, CIGAR/SEQ length mismatch */

#define MDX_ERR_UNSUPPORTED (-8)  /* mdx_gbam_*: something the GPU decode path does not take (use mdx_bam_*, from mdx_gbam_tell on) */
#define MDX_ERR_COMM (-7)         /* RCCL failure, or another rank of the communicator reported an error */

/* Optional hint in the `flag` column (a bit SAM does not define): every base quality of the record is at least
 * --min-basequal, so nothing of it can be masked (align.py:65-71 masks only qualities below the threshold).  The
 * kernel then skips the record's quality window loads.  The caller vouches for it (mdx_bam_qmin gives the lowest
 * quality of each record; mapdamage_amd.batch.mark_unmaskable sets the bit); without the bit nothing changes.
 * SAM defines FLAG bits 0..11 only, but a file may carry anything in its 16 bits: the decoders of this library
 * (mdx_bam_*, mdx_gbam_*, the SAM text parser) clear bits 14 and 15 (see MDX_FLAG_HAS_QUAL) of what they read, and a caller that packs its own flag
 * column must do the same unless it vouches for the record. */
#define MDX_FLAG_QUAL_ABOVE_MIN 0x8000
/* A second hint, bit 14 (round 6): the record HAS base qualities — it holds at least one base and the first byte of its
 * quality string is not 0xFF (what `not read.qual` tests at main.py:185 and rescale.py:306).  The rescaling kernels then do
 * not fetch that byte to route the record (a scattered load per record: a tenth of the fused launch's time when no second
 * quality column is written, mdx_tabulate_rescale_patches_device); without the bit they look.  Set by the library where it
 * sees the qualities anyway: mdx_batch_upload (a batch with a quality column) and the device decoder; the decoders clear
 * bits 14 and 15 of what a file carries, and a caller that packs its own flag column does the same unless it vouches. */
#define MDX_FLAG_HAS_QUAL 0x4000

#define MDX_N_MIS_COLS 25         /* mapdamage/seq.py:6-30 without the derived "Total" */

typedef struct mdx_ctx mdx_ctx;

/* Per-run options: --length/--around/--min-basequal (mapdamage/config.py:143-166) and the
 * number of libraries (mapdamage/reader.py:47-50; 1 with --merge-libraries, reader.py:44-46). */
typedef struct {
    int32_t length;        /* L >= 1 */
    int32_t around;        /* A >= 0 */
    int32_t minqual;       /* 0..93; 0 disables masking */
    int32_t nlib;          /* >= 1 */
    int32_t lgd_max;       /* dense fragment-length histogram covers [0, lgd_max) */
    int32_t device;        /* HIP device ordinal */
    int64_t lgd_over_cap;  /* capacity (records) of the out-of-range length list */
} mdx_config;

/* The two forms of the `seq` column ("seq:u8[] (ASCII or 4-bit)").
 *   MDX_SEQ_ASCII  one byte per base, the characters pysam's read.query holds.
 *   MDX_SEQ_4BIT   two bases per byte, base i of the column in bits [4 (i & 1), 4 (i & 1) + 4) of byte i / 2 (the
 *                  low nibble first — BAM stores the high nibble first), code 1 = 'A', 2 = 'C', 4 = 'T', 8 = 'G' and 0 for
 *                  every other symbol.  Nothing is lost: the reference counts a read symbol only when it is exactly one
 *                  of "ACGT" (statistics.py:27, 101; rescale.py:228-246), every other one behaves alike.  seq_off and
 *                  n_bases keep counting bases; the column holds (n_bases + 1) / 2 bytes.  mdx_pack_seq converts;
 *                  mdx_gbam_set_seq_format makes the device decode path write this form straight from BAM's nibbles.
 *                  Such a batch runs through the packed kernels (half the SEQ and reference bytes, no LDS update per
 *                  plain match): the plain tabulation, --min-basequal (mdx_batch::lowq) and, with one library,
 *                  mdx_tabulate_rescale_device — tables in the LDS, reference below 4 Gbases; for anything else the
 *                  library unpacks it into a scratch column first and the results are the same bytes. */
#define MDX_SEQ_ASCII 0
#define MDX_SEQ_4BIT 1
/*   MDX_SEQ_4BITQ  MDX_SEQ_4BIT with --min-basequal folded in (align.py:53-73: a read column whose quality is below the
 *                  threshold turns into N / N — it counts its read base in the composition table, statistics.py:100-103, and
 *                  nothing else): a symbol whose quality is below the CONTEXT's threshold is stored as the complement of its
 *                  code (14, 13, 11, 7 for A, C, T, G; 15 for a symbol that is no base), every other nibble as in MDX_SEQ_4BIT.  The packed masked kernel reads
 *                  nothing but this column — no quality byte, no bitmap.  Made by the library: mdx_batch_upload and the device
 *                  decoder (mdx_gbam_set_min_basequal) hand such a column over when the context has a --min-basequal,
 *                  mdx_tabulate_host folds on the copy stream; a MDX_SEQ_4BIT batch with qualities (or a bitmap, `lowq`) is
 *                  folded into a scratch column in front of every launch.  The column belongs to the context whose threshold
 *                  made it (a context without one refuses it). */
#define MDX_SEQ_4BITQ 2

/* One batch of alignment records as SoA columns: exactly what main.py:165-217 reads from each
 * pysam.AlignedSegment.  `seq` is the full SEQ (soft clips included); `qual`
 * raw Phred (BAM convention, first byte 0xFF = absent) or NULL; `cigar` BAM-encoded len<<4|op.
 * Limits of one batch: fewer than 2^30 records, at most 4 GiB of bases (32-bit seq_off); larger inputs are
 * fed as several batches (the context accumulates).  n_bases must be exact: it bounds the 8/12-byte window
 * loads of the kernels (no byte outside [seq, seq + n_bases) is read). */
typedef struct {
    int64_t n_reads;
    int64_t n_cigar;       /* == cigar_off[n_reads] */
    int64_t n_bases;       /* == seq_off[n_reads]   */
    const uint16_t *flag;
    const uint16_t *lib;
    const int32_t *tid;
    const int32_t *pos;
    const int32_t *tlen;
    const uint32_t *cigar_off; /* n_reads + 1 */
    const uint32_t *cigar;
    const uint32_t *seq_off;   /* n_reads + 1 */
    const uint8_t *seq;
    const uint8_t *qual;       /* may be NULL */
    int32_t seq_format;        /* MDX_SEQ_ASCII (0), MDX_SEQ_4BIT or MDX_SEQ_4BITQ */
    int32_t reserved;          /* 0 */
    /* Optional, device batches only, used with --min-basequal and a MDX_SEQ_4BIT column: bit i & 31 of 32-bit word i / 32 =
     * qual[i] is below the context's --min-basequal (align.py:65-71; 0xFF — no qualities — is not); (n_bases + 31) / 32 words.
     * A caller that has the bits spares the library the pass over the quality column when it folds the mask into a scratch
     * copy of the column in front of the launch (MDX_SEQ_4BITQ).  NULL: folded from `qual`.  The library's own batches
     * (mdx_batch_upload, the device decoder) are MDX_SEQ_4BITQ already and leave this NULL. */
    const uint8_t *lowq;
    /* Optional, device batches only, used when the context has several libraries and the seq column is 4-bit: the
     * per-record columns bucketed by library (reader.py:47-50, statistics.py:12-20 — the tables are keyed by library, a file
     * interleaves them): an opaque device blob.  The packed kernel then counts all libraries in ONE launch, one after the
     * other, each over its own records.  NULL: the library sorts in front of every launch (mdx_libsorts counts those).
     * mdx_batch_upload fills it in, mdx_batch_free releases it; a caller that builds mdx_batch itself leaves it NULL. */
    const uint8_t *libsort;
} mdx_batch;

/* ASCII SEQ bytes -> the MDX_SEQ_4BIT column (host buffers; `packed` holds (n_bases + 1) / 2 bytes; `threads` host
 * threads, 0 = all).  What main.py:180-205 and statistics.py:22-35 do with a read symbol depends only on which of
 * "ACGT" it is, if any. */
int mdx_pack_seq(const uint8_t *ascii, int64_t n_bases, uint8_t *packed, int32_t threads);

int mdx_abi_version(void);
const char *mdx_strerror(int code);

/* Replaces the construction of MisincorporationRates / DNAComposition / FragmentLengths
 * (mapdamage/main.py:147-155, statistics.py:10-20,59-73,107-115): zeroed device accumulators. */
int mdx_create(const mdx_config *cfg, mdx_ctx **out);
void mdx_destroy(mdx_ctx *ctx);
const char *mdx_last_error(const mdx_ctx *ctx);

/* Use `hip_stream` (a hipStream_t) for all subsequent work; NULL = the context's own stream. */
int mdx_set_stream(mdx_ctx *ctx, void *hip_stream);

/* Replaces pysam.FastaFile(options.ref) + the per-read ref.fetch(...).upper() calls
 * (main.py:115,180; align.py:32-33): uploads the contigs (original FASTA bytes, BAM tid order,
 * contig i = bases[contig_off[i] .. contig_off[i+1])) once and keeps them resident in HBM,
 * case-folded and symbol-classified by a device kernel.  Host pointers. */
int mdx_set_reference(mdx_ctx *ctx, const uint8_t *bases, const int64_t *contig_off, int32_t n_contig);

/* The same from the FASTA file itself: replaces pysam.FastaFile(options.ref) (main.py:115 — htslib's faidx, which reads
 * `<fasta>.fai` and builds it when the file has none) and the fetches behind it (main.py:180, align.py:32-33).
 * mdx_fasta_index: makes sure `<fasta_path>.fai` exists (name, length, offset, linebases, linewidth per sequence, htslib's
 * format; lines of unequal length within a sequence are an error, as they are to faidx); err (may be NULL) receives the text.
 * mdx_set_reference_fasta: the sequences `names[0 .. n_contig)` (the BAM header's, in tid order: chrom lookup is by name,
 * main.py:175-180) become the resident reference.  The file's bytes go to HBM as they lie on disk, a piece at a time, and a
 * kernel takes the line ends out by the index's arithmetic — no pass over the bases on the host; pieces of the file that
 * hold none of the wanted sequences are not read.  lengths (may be NULL) receives the length of each; a name the index
 * lacks is MDX_ERR_ARG, or with missing_ok an empty contig (a record mapped to it is MDX_ERR_BAD_READ when it is met: the
 * reference fails in fetch at that read, not before).  Uncompressed FASTA only (a bgzip-ed one goes through
 * mdx_set_reference).  Synchronous. */
int mdx_fasta_index(const char *fasta_path, char *err, int32_t err_cap);
int mdx_set_reference_fasta(mdx_ctx *ctx, const char *fasta_path, int32_t n_contig, const char *const *names, int32_t missing_ok,
                            int64_t *lengths);
/* Introspection for tests: bases [start, end) of contig tid of the resident reference as the kernels see them — the letter
 * where ref.fetch(chrom, start, end).upper() (main.py:180) holds one of "ACGT", '-' for '-', 'N' for anything else.  Host
 * buffer of end - start bytes; synchronous. */
int mdx_reference_fetch(mdx_ctx *ctx, int32_t tid, int64_t start, int64_t end, uint8_t *out);

/* Copies a host batch into device memory owned by the library (for resident/benchmark use).
 * `dev` receives device pointers; release with mdx_batch_free. */
int mdx_batch_upload(mdx_ctx *ctx, const mdx_batch *host, mdx_batch *dev);
int mdx_batch_free(mdx_ctx *ctx, mdx_batch *dev);
/* A caller's own device batch under --min-basequal (align.py:53-73): a MDX_SEQ_4BIT column with qualities (or the bitmap
 * `lowq`) is folded into a scratch column in front of EVERY launch that counts it — a pass over column and qualities each
 * time (2.2 x the unmasked launch against 1.3 x).  mdx_batch_fold does it ONCE, in place: the batch's seq column takes the
 * mask into its nibbles and *dev_batch becomes MDX_SEQ_4BITQ — from then on it belongs to this context's threshold (another
 * context with another threshold must not be handed it: check_batch cannot tell).  Enqueued on the context's stream.  The
 * library's own batches (mdx_batch_upload, the device decoder) are folded when they are made. */
int mdx_batch_fold(mdx_ctx *ctx, mdx_batch *dev_batch);

/* Replaces the loop body main.py:165-217 for every record of the batch: flag filter
 * (reader.py:121-132), coordinates/flanks (align.py:14-35), CIGAR gapping with optional
 * quality masking (align.py:38-88), strand step (main.py:200-205), and the accumulator
 * updates (statistics.py:22-51,75-93,117-126).  Accumulates; may be called repeatedly.
 * _host: columns in (pageable) host memory, copied through two pinned bounce buffers of the context, the call
 * returns when the last column has left the caller's buffers; _device: columns already in HBM.
 * Several libraries: a 4-bit seq column is counted by one launch over the records bucketed by library (mdx_batch::libsort);
 * an ASCII one by one launch per group of libraries that fits the LDS, each over all records.  Transparently. */
int mdx_tabulate_host(mdx_ctx *ctx, const mdx_batch *batch);
int mdx_tabulate_device(mdx_ctx *ctx, const mdx_batch *batch);

/* The index mdx_sync reports for a bad record is its index within its batch plus this base (default 0): a caller
 * that enqueues several batches before it synchronises (mdx_tabulate_host returns as soon as the host columns are
 * staged; copies and kernels of consecutive batches overlap) sets the base to the number of records in front of each
 * batch and gets back an index in its own numbering.  The lowest such index of all batches since the last
 * mdx_sync / mdx_reset is reported. */
int mdx_set_record_base(mdx_ctx *ctx, int64_t base);

/* Waits for the stream and reports deferred per-read errors (MDX_ERR_BAD_READ ...);
 * *bad_read (may be NULL) receives the index of the first offending record of its batch. */
int mdx_sync(mdx_ctx *ctx, int64_t *bad_read);

/* Number of uint64 words of the packed canonical table block written by mdx_finish_device:
 * [ mis nlib*2*2*L*25 | comp nlib*2*2*(L+A)*4 | lgd nlib*2*2*lgd_max | n_kept | n_lgd_over ]. */
int64_t mdx_table_words(const mdx_ctx *ctx);

/* Writes the canonical tables (layout: mapdamage_amd/layout.py) into a *device* buffer of
 * mdx_table_words() uint64 — this context's own counts; mdx_finish_allreduce sums them across GPUs
 * (or the caller all-reduces the buffer itself, e.g. with torch.distributed). */
int mdx_finish_device(mdx_ctx *ctx, uint64_t *d_tables);

/* Replaces reading the `.data` dicts before `.write()` (main.py:229-231): synchronises and
 * copies the canonical tables to host memory.  lgd_over receives (lib, kind, strand, length)
 * quadruples for lengths >= lgd_max (at most lgd_over_cap); any pointer may be NULL.
 * With a communicator attached (mdx_comm_init / mdx_comm_adopt) the call is collective and returns the
 * totals over all ranks on every rank; lgd_over then receives the lists of all ranks in rank order (every rank is
 * bounded by the context's lgd_over_cap, so mdx_comm_size() x that many entries always suffice) and a buffer too
 * small for them is MDX_ERR_LGD_OVERFLOW, never a shortened list. */
int mdx_finish(mdx_ctx *ctx, uint64_t *mis, uint64_t *comp, uint64_t *lgd, int64_t *lgd_over,
               int64_t lgd_over_cap, int64_t *n_lgd_over, int64_t *n_kept);

/* Cross-device reduction (SURVEY.md §8b "finish() performs cross-device reduction", §8e).  The reference has no
 * counterpart (it is a single process); what makes the reduction legal is that every accumulator update is a
 * `+= 1` (statistics.py:30,35,40,103,124,126).  One context per GPU / per process; records are sharded over
 * the contexts with no data-path exchange, and the tables are summed with one ncclAllReduce(ncclUint64, ncclSum)
 * over xGMI on the context's stream.  librccl.so.1 is resolved at run time (dlopen) by the first of these calls;
 * a process that never calls them does not need RCCL.
 *   mdx_comm_unique_id  rank 0 obtains the 128-byte rendezvous id (ncclGetUniqueId) and hands it to the other
 *                       ranks by any means (MPI, a file, torch.distributed's store);
 *   mdx_comm_init       every rank: ncclCommInitRank on the context's device (collective, blocks until all joined);
 *   mdx_comm_adopt      alternatively use an ncclComm_t the caller already owns (not destroyed by mdx_destroy);
 *   mdx_finish_allreduce  mdx_finish_device + in-place all-reduce of the packed block: every rank ends up with
 *                       the totals in its device buffer.  Enqueued on the context's stream; collective.
 * With a communicator attached, mdx_finish itself performs the reduction: an error flag is agreed on first (a rank
 * whose mdx_sync fails makes every rank return — MDX_ERR_COMM on the healthy ones — instead of leaving them in
 * the collective), then the tables are all-reduced and the out-of-range length lists all-gathered in rank order. */
#define MDX_COMM_ID_BYTES 128
int mdx_comm_unique_id(uint8_t *id /* MDX_COMM_ID_BYTES */);
int mdx_comm_init(mdx_ctx *ctx, const uint8_t *id, int32_t nranks, int32_t rank);
int mdx_comm_adopt(mdx_ctx *ctx, void *rccl_comm, int32_t nranks, int32_t rank);
int mdx_comm_size(const mdx_ctx *ctx);   /* 0 = no communicator attached */
/* The ranks RCCL itself counts in the attached communicator (ncclCommCount; 0 = none attached, < 0 = error): what a
 * benchmark line quotes as proof that the collective ran over that many ranks. */
int mdx_comm_count(mdx_ctx *ctx);
int mdx_finish_allreduce(mdx_ctx *ctx, uint64_t *d_tables);

/* Zero all accumulators (new run with the same options and reference). */
int mdx_reset(mdx_ctx *ctx);

/* Kernel timing with HIP events recorded on the launch stream around the tabulation kernel.
 * mdx_timing_read synchronises, returns the number of timed launches and their summed
 * duration since the last call, and clears the record. */
int mdx_timing_enable(mdx_ctx *ctx, int enable);
int mdx_timing_read(mdx_ctx *ctx, int64_t *n_launches, double *total_ms);

/* Replaces seqtk.comp() of the reference's native extension (mapdamage/seqtk/seqtk.c:55-143) as
 * used by composition.write_base_comp (mapdamage/composition.py:6-25): per-contig counts of
 * A, C, G, T (upper and lower case folded) of the resident reference.
 * counts: host buffer of n_contig * 4 uint64, order A, C, G, T.  Synchronous. */
int mdx_genome_composition(mdx_ctx *ctx, uint64_t *counts);

/* Quality rescaling, mapdamage/rescale.py (BASELINE configs[4]).
 * mdx_rescale_set_model replaces _get_corr_prob + the per-column floating-point work of
 * _rescale_qual_read (rescale.py:23-46, 228-246): `lut` [2][1+len5p+len3p][94] maps (substitution
 * 0 = C>T / 1 = G>A, position key, old Phred) to the new Phred; `term` [2][1+len5p+len3p] is the
 * per-column contribution to the MR tag.  Position key 0 = no correction, 1..len5p = position from
 * the 5' end, len5p+k = position -k from the 3' end (mapdamage_amd/rescale.py builds both with the
 * reference's own expressions).
 * mdx_rescale_host replaces _rescale_qual_core's loop (rescale.py:300-344) for a batch: record
 * routing, _rescale_qual_read, soft-clip re-attachment.  Host pointers; synchronous.  qual_out has
 * the layout of batch->qual; mr_raw[i] is the MR sum before its "%.5f" truncation (NaN when record i
 * is written unchanged); status[i]: 0 unmapped, 1 no qualities, 2 rescaled from both ends,
 * 3 rescaled from the 5' end only (inward-facing pair), 4 improperly paired (unchanged). */
int mdx_rescale_set_model(mdx_ctx *ctx, const uint8_t *lut, const double *term, int32_t len5p, int32_t len3p);
int mdx_rescale_host(mdx_ctx *ctx, const mdx_batch *batch, const int32_t *mtid, const int32_t *mpos,
                     uint8_t *qual_out, double *mr_raw, uint8_t *status);
/* The same for a batch resident in HBM (device pointers throughout, enqueued on the context's stream; errors surface
 * at mdx_sync), and — BASELINE configs[4], "rescaling fused into the same pass" — both results of one resident
 * batch in one call: the count tables (main.py:165-217) and the rescaled qualities (rescale.py:300-344) from the
 * same columns, uploaded once.  d_qual_out is a column of its own (n_bases bytes; it must not be the batch's quality
 * column: MDX_ERR_ARG), filled completely — the qualities of records written back unchanged included.
 * mdx_tabulate_rescale_device is one fused launch when the tabulation runs as the plain fast kernel (tables in the LDS,
 * all libraries in one launch, no --min-basequal, reference below 4 GiB) and the model has at most 31 window positions:
 * the tabulation kernel rescales the [S] M [S] records of at most 2 x --length aligned bases while it counts them and
 * lists the others for the rescale kernels behind it; otherwise — and with MDX_NO_FUSE=1 in the environment — it is
 * mdx_tabulate_device followed by mdx_rescale_device.  Either way the results are the same bytes.  A NaN in mr_raw
 * (record written back unchanged) is any NaN.
 * mdx_rescale_timing_read: time of the rescale launches (the rescale kernels with the copy of the quality column
 * they contain — after a fused launch: of the kernels that take the records it has listed; HIP events, like
 * mdx_timing_read for the tabulation kernel, which then times the fused kernel; enabled by mdx_timing_enable). */
int mdx_rescale_device(mdx_ctx *ctx, const mdx_batch *dev_batch, const int32_t *d_mtid, const int32_t *d_mpos,
                       uint8_t *d_qual_out, double *d_mr_raw, uint8_t *d_status);
int mdx_tabulate_rescale_device(mdx_ctx *ctx, const mdx_batch *dev_batch, const int32_t *d_mtid, const int32_t *d_mpos,
                                uint8_t *d_qual_out, double *d_mr_raw, uint8_t *d_status);
/* The same two calls with the rescaled qualities as a LIST instead of a second column (round 6): _rescale_qual_read changes a
 * few quality bytes of a read — the C>T / G>A columns near its ends (rescale.py:228-246) — and a caller that writes the
 * records back (rescale.py:266-273, 344) needs those bytes, not a copy of the hundred it leaves alone.  One entry per quality
 * byte whose value changes: index of the byte in the batch's quality column | new Phred << 32, in no particular order.  The
 * list comes in n_parts parts (a power of two; 256 and more for a launch of millions of records: thousands of wavefronts
 * appending to one list queue at one address): part p = d_patch[p * patch_cap ..], d_n_patch[p] (device; zeroed by the call)
 * its number of entries — which may pass patch_cap: the entries beyond are lost and the caller repeats the call with longer
 * parts (n_reads x (len5p + len3p) entries in all always suffice; the parts fill evenly).  Nothing else is written: of a
 * record that is written back unchanged no quality byte is read but its first (rescale.py:306).  mr_raw, status, the
 * summary, errors: as above.  mdx_rescale_expand_device: d_qual_out = the batch's quality column with a list applied (for
 * callers that want the column after all, and the tests; d_qual_out may be the column itself). */
int mdx_rescale_patches_device(mdx_ctx *ctx, const mdx_batch *dev_batch, const int32_t *d_mtid, const int32_t *d_mpos,
                               uint64_t *d_patch, int64_t patch_cap, int32_t n_parts, uint64_t *d_n_patch, double *d_mr_raw,
                               uint8_t *d_status);
int mdx_tabulate_rescale_patches_device(mdx_ctx *ctx, const mdx_batch *dev_batch, const int32_t *d_mtid, const int32_t *d_mpos,
                                        uint64_t *d_patch, int64_t patch_cap, int32_t n_parts, uint64_t *d_n_patch, double *d_mr_raw,
                                        uint8_t *d_status);
int mdx_rescale_expand_device(mdx_ctx *ctx, const mdx_batch *dev_batch, const uint64_t *d_patch, int64_t patch_cap, int32_t n_parts,
                              const uint64_t *d_n_patch, uint8_t *d_qual_out);
int mdx_rescale_timing_read(mdx_ctx *ctx, int64_t *n_launches, double *total_ms);
/* Calls of mdx_tabulate_rescale_device so far that ran as the fused launch (the others: two kernels). */
int64_t mdx_fused_launches(const mdx_ctx *ctx);
/* Kernel launches so far that ran as the packed kernel (a MDX_SEQ_4BIT batch in a plain tabulation or with --min-basequal:
 * one per call whatever the number of libraries — up to 64 libraries per launch, MDX_ML_MAX_LIBS, and as many as the
 * launch has pools of two blocks). */
int64_t mdx_packed_launches(const mdx_ctx *ctx);
/* Calls so far that bucketed their batch by library themselves (several libraries, a batch without mdx_batch::libsort). */
int64_t mdx_libsorts(const mdx_ctx *ctx);
/* The integer content of the `subs` dictionary that _rescale_qual_read fills through _record_subs
 * (rescale.py:82-143) and _print_subs logs (:159-192), accumulated over every mdx_rescale_host call
 * since mdx_rescale_set_model.  words (uint64), npos = 1 + len5p + len3p:
 *   [0, 4)                  subs["A"], ["C"], ["G"], ["T"]: reference bases of the rescaled reads' columns
 *   [4, 4 + 4*2*94)         [transition 0=CT,1=TC,2=GA,3=AG][0=before,1=after][Phred]: subs["CT-before"] ...
 *   [756, 756 + 2*npos*94)  [0=C>T,1=G>A][position key][old Phred]: occurrences of each rescaled column kind,
 *                           from which the host sums the "-pvals" terms (each is a function of that triple)
 * mdx_rescale_summary_words returns the number of words (0 before a model is set). */
int64_t mdx_rescale_summary_words(const mdx_ctx *ctx);
int mdx_rescale_summary(mdx_ctx *ctx, uint64_t *words);

/* Native BAM decoding (host side, no GPU involved).  Replaces opening and iterating a
 * pysam.AlignmentFile (mapdamage/reader.py:38, 83-96; pysam is not needed): the BGZF blocks are
 * inflated on `threads` host threads and every record is unpacked into the SoA columns of mdx_batch.
 * The handle owns all memory; mdx_bam_batch returns host views valid until mdx_bam_free.
 * `lib` is zero-filled (the caller maps read groups to libraries, reader.py:63-81, from rg_index:
 * per-record index into mdx_bam_rg_name(), -1 = no RG tag).  mtid/mpos are the mate fields used by
 * the rescale routing; has_mr flags records that already carry an MR tag (rescale.py:277). */
typedef struct mdx_bam mdx_bam;
int mdx_bam_read(const char *path, int threads, mdx_bam **out);
void mdx_bam_free(mdx_bam *bam);
const char *mdx_bam_error(const mdx_bam *bam);
const char *mdx_bam_header_text(const mdx_bam *bam);
int32_t mdx_bam_n_ref(const mdx_bam *bam);
const char *mdx_bam_ref_name(const mdx_bam *bam, int32_t i);
int64_t mdx_bam_ref_length(const mdx_bam *bam, int32_t i);
int mdx_bam_batch(const mdx_bam *bam, mdx_batch *view, const int32_t **mtid, const int32_t **mpos,
                  const int32_t **rg_index, const uint8_t **has_mr);
const uint8_t *mdx_bam_qmin(const mdx_bam *bam);     /* lowest quality per record (0xFF: none) */
int32_t mdx_bam_n_rg(const mdx_bam *bam);
const char *mdx_bam_rg_name(const mdx_bam *bam, int32_t i);
const char *mdx_bam_qnames(const mdx_bam *bam, const uint32_t **offsets);

/* Streaming form of the same decoder, for files larger than host memory and to overlap decoding with
 * tabulation (the reference iterates its AlignmentFile record by record, mapdamage/main.py:165; here the
 * unit is a chunk of records).  mdx_bam_open parses the header (mdx_bam_stream_header: a record-less
 * mdx_bam for the accessors above, owned by the stream, whose mdx_bam_error also carries stream errors).
 * mdx_bam_next returns the complete records within the next chunk_bytes of uncompressed BAM data as a new
 * mdx_bam (caller frees with mdx_bam_free; independent of the stream), *out = NULL at end of file.  Records
 * come in file order; rg_index of a chunk indexes that chunk's own mdx_bam_rg_name().  One thread per
 * stream; chunks may be consumed on other threads. */
typedef struct mdx_bam_stream mdx_bam_stream;
int mdx_bam_open(const char *path, int threads, mdx_bam_stream **out);
const mdx_bam *mdx_bam_stream_header(const mdx_bam_stream *stream);
int mdx_bam_next(mdx_bam_stream *stream, int64_t chunk_bytes, mdx_bam **out);
/* Points the stream at the BGZF block at compressed offset comp_off; the next chunk starts with the record `phase`
 * inflated bytes into that block (the pair mdx_gbam_tell hands out; 0, 0 of a file's first record block rewinds). */
int mdx_bam_seek(mdx_bam_stream *stream, int64_t comp_off, int64_t phase);
void mdx_bam_close(mdx_bam_stream *stream);
/* Rewriting a BAM (the `--rescale-only` output of mapdamage/rescale.py:285-365, which writes every record back
 * through pysam): with mdx_bam_stream_keep_raw(stream, 1) each chunk keeps its encoded records as they stood in
 * the file; mdx_bam_raw returns them (rec_off[i] = offset of record i's block_size field, rec_off[n] = end);
 * mdx_bam_patch_rescaled writes the chunk's records to `out` (capacity out_cap bytes: raw size + 7 per rescaled
 * record suffices), those with rescaled[i] != 0 with their QUAL replaced by qual_out[seq_off[i] ...] and an `MR:f`
 * tag (mr[i]) appended, every other byte unchanged. */
/* The BGZF writer on the device (round 6): replaces zlib's deflate behind pysam's AlignmentFile(..., "wb") for the output of the
 * rescaling pass (rescale.py:290-291, :344 — every record is written back; on sixteen host threads the output's blocks were
 * 2.1 of the pass's 3.5 seconds).  `data` (host, n bytes of encoded BAM: header or records, any cut) is cut into blocks of
 * 0xFF00 bytes, every block becomes one BGZF member — in four pieces, a lane per piece: LZ77 with one hash candidate that may
 * reach back into the piece in front, a dynamic Huffman code of the piece's own counts, a stored block where that is not
 * smaller, the pieces joined the way zlib's Z_SYNC_FLUSH joins blocks (csrc/mdx_deflate.h, held against zlib's inflate by
 * tests/test_inflate_core.py) —, and the members are written side by side to `out` (host; n + (n / 0xFF00 + 1) * 64 bytes
 * always suffice), *out_len their bytes.  No end-of-file marker.  Any inflater reads the result; it is not zlib's output bit
 * for bit (3 % larger than its level 6 on BAM records with qualities, smaller than its level 1).  2.9 GB/s of records in, host
 * buffer to host buffer (sixteen host threads at level 6: 0.63 GB/s).  Synchronous, on the context's stream. */
int mdx_bgzf_deflate(mdx_ctx *ctx, const uint8_t *data, int64_t n, uint8_t *out, int64_t out_cap, int64_t *out_len);
/* --rescale-only with the records never on the host (round 6; rescale.py:285-365).  The handle is configured with qualities
 * and mate columns and hands out ASCII seq columns (mdx_gbam_configure(.., want_qual = 1, want_mate = 1)); after every
 * mdx_gbam_next:
 *   mdx_gbam_rescale_slab   the slab's records through the context's rescale kernels (mdx_rescale_set_model first; the list
 *                           form, mdx_rescale_patches_device), the new quality bytes into the QUAL fields of the inflated
 *                           records in HBM, MR rounded as rescale.py:275-276 does, every record — rescaled ones with an MR:f tag
 *                           appended — laid out as the output stream in HBM and compressed there (mdx_bgzf_deflate's kernels): `out`
 *                           (host; the slab's inflated size + 7 bytes per record + 64 per member always suffice) receives BGZF
 *                           members only.  counts[0..5) += records by routing status (rescale.py:300-342).  A rescaled record
 *                           that has an MR tag already: MDX_ERR_BAD_READ, *mr_clash = its index in the slab (rescale.py:277-278;
 *                           mdx_gbam_record_name gives its name); a record the kernels cannot process: *mr_clash = -2 - index.
 *   mdx_gbam_write_rescaled the second half on its own, for a caller that ran the rescale kernels itself: patch list (device),
 *                           mr / rescaled per record (host).
 * The file's header is the caller's to write (mdx_bgzf_deflate takes any bytes), and the end-of-file marker. */
int mdx_gbam_rescale_slab(struct mdx_gbam *g, uint8_t *out, int64_t out_cap, int64_t *out_len, int64_t *counts, int64_t *mr_clash);
int mdx_gbam_write_rescaled(struct mdx_gbam *g, const uint64_t *d_patch, int64_t patch_cap, int32_t n_parts, const uint64_t *d_n_patch,
                            const float *mr, const uint8_t *rescaled, uint8_t *out, int64_t out_cap, int64_t *out_len, int64_t *mr_clash);
int mdx_gbam_record_name(struct mdx_gbam *g, int64_t index, char *buf, int32_t cap);
/* mdx_mr_round: float("%.5f" % x) of rescale.py:275-276 for n MR sums (mr_raw of the rescale calls) on `threads` host threads —
 * the value printed with five decimals, read back, and narrowed to the 32 bits of an MR:f tag; NaN (a record written back
 * unchanged) gives 0. */
int mdx_mr_round(const double *mr_raw, int64_t n, float *out, int32_t threads);
int mdx_bam_stream_keep_raw(mdx_bam_stream *stream, int on);
int mdx_bam_raw(const mdx_bam *bam, const uint8_t **data, const uint64_t **rec_off);
int mdx_bam_patch_rescaled(const mdx_bam *bam, const uint8_t *qual_out, const float *mr, const uint8_t *rescaled,
                           uint8_t *out, int64_t out_cap, int64_t *out_len);

/* Introspection for tests/benchmarks: 0 = LDS-privatised path, 1 = global-atomic fallback. */
int mdx_table_mode(const mdx_ctx *ctx);

/* ---- GPU-side BAM decode.  The device counterpart of mdx_bam_open / mdx_bam_next — and with them of
 * pysam.AlignmentFile behind mapdamage/reader.py:20-46: the compressed file goes to HBM a slab of BGZF blocks at a
 * time, is inflated there (one wavefront per block; RFC 1951, mapdamage_amd/csrc/mdx_inflate.h) and unpacked there
 * into the columns of an mdx_batch that never exist on the host; the view mdx_gbam_next fills holds DEVICE pointers,
 * ready for mdx_tabulate_device / mdx_rescale_device, valid until the next mdx_gbam_next / mdx_gbam_close (which
 * wait for the context's stream first).  The header is parsed on the host (mdx_gbam_header: for the mdx_bam_*
 * accessors).  mdx_gbam_configure: the header's read-group ids with the library of each, the library of a record
 * without RG tag (-1: none — such a record gets library 0xFFFF, MDX_ERR_BAD_READ at mdx_sync if it is one the kernel
 * counts, as is a read group the header does not list), and whether the quality and mate columns are wanted.
 * Any BGZF layout is taken: htslib starts every block at a record and flushes the header into blocks of its own
 * (bgzf_flush_try in bam_write1); htsjdk / Picard, sambamba and biobambam fill their blocks whatever the record
 * boundaries.  The inflated blocks of a slab lie back to back in HBM, every block guesses where its first record
 * starts, and a guess counts only if the chain of records in front of it ends there (a block whose guess was wrong is
 * scanned again: mdx_gbam_fixups); a slab's batch holds the records that START in it — the blocks behind it are inflated
 * as far as its last record reaches.  MDX_ERR_UNSUPPORTED is what is left: a record longer than a gigabyte, and (see
 * mdx_gbam_skip) a sharded run over a file whose record boundaries cannot be told without the slab in front.  The
 * CRC32 of every block is checked on the device, like its ISIZE (MDX_ERR_ARG, as in the host decoder).
 * chunk_bytes: compressed bytes per slab.  mdx_gbam_next at the end of the file: MDX_OK, n_reads 0,
 * mdx_gbam_at_end 1.  mdx_ctx_stream: the HIP stream (hipStream_t) and device the context works on. */
typedef struct mdx_gbam mdx_gbam;
int mdx_ctx_stream(mdx_ctx *ctx, void **stream, int *device);
int mdx_gbam_open(mdx_ctx *ctx, const char *path, mdx_gbam **out);
const mdx_bam *mdx_gbam_header(const mdx_gbam *g);
const char *mdx_gbam_error(const mdx_gbam *g);
int mdx_gbam_configure(mdx_gbam *g, int32_t n_rg, const char *const *rg_ids, const int32_t *lib_of_rg, int32_t lib_default,
                       int want_qual, int want_mate);
int mdx_gbam_next(mdx_gbam *g, int64_t chunk_bytes, mdx_batch *dev_view, const int32_t **d_mtid, const int32_t **d_mpos);
/* --min-basequal on the device path (needs want_qual): records none of whose qualities is below the threshold get
 * MDX_FLAG_QUAL_ABOVE_MIN in the flag column, a 4-bit SEQ column takes the mask into its nibbles (the views are
 * MDX_SEQ_4BITQ), a slab without a single maskable record is handed over without its quality column (the unmasked
 * kernel), and mdx_gbam_missing_qualities says whether a record the kernel counts has come by without qualities so far
 * (what main.py:185-192 warns about).  minqual must be the threshold of the context the file was opened on (or 0): the
 * views carry it in their nibbles (MDX_ERR_ARG otherwise). */
int mdx_gbam_set_min_basequal(mdx_gbam *g, int32_t minqual);
/* MDX_SEQ_4BIT: the unpack kernel keeps BAM's nibbles (recoded, low nibble first) instead of expanding them to ASCII;
 * the views of mdx_gbam_next then carry seq_format = MDX_SEQ_4BIT.  Default MDX_SEQ_ASCII. */
int mdx_gbam_set_seq_format(mdx_gbam *g, int32_t seq_format);
int mdx_gbam_missing_qualities(const mdx_gbam *g);
int mdx_gbam_at_end(const mdx_gbam *g);
/* Steps over the slab mdx_gbam_next would decode next (same chunk_bytes, same borders) without touching the device: in a
 * run over several GPUs (one process and one mdx_gbam per GPU) rank r decodes the slabs r, r + N, ... and skips
 * the others — the loop of mapdamage/main.py:165-217 sharded by record with no exchange until the tables are summed. */
int mdx_gbam_skip(mdx_gbam *g, int64_t chunk_bytes);
/* Where the next slab begins: compressed offset of its first BGZF block and the inflated bytes in front of its first
 * record (a record may straddle BGZF blocks and slabs; a slab holds the records that start in it).  mdx_bam_seek takes the
 * pair: when mdx_gbam_next fails, the tables hold the slabs in front and the host decoder can go on from here.
 * MDX_ERR_STATE behind mdx_gbam_skip (the offset is the device scan's guess then). */
int mdx_gbam_tell(const mdx_gbam *g, int64_t *comp_off, int64_t *phase);
/* BGZF blocks whose guessed first record was not where the chain of records in front of it ended; such a block is scanned
 * again from the right offset (the batch is exact either way; 0 for a file laid out the way htslib does). */
int mdx_gbam_fixups(const mdx_gbam *g);
/* --downsample on the device path (mapdamage/reader.py:134-146: a record the flag filter keeps stays with probability p,
 * one draw of Python's random.Random per kept record in file order — the generator stays with the caller).
 * mdx_gbam_view_flags copies the flag column of the view mdx_gbam_next handed out last to the host (n = its n_reads;
 * 2 bytes per record), mdx_gbam_view_set_flags writes it back: the caller draws, marks the records that leave with a bit
 * the flag filter drops (0x200) and tabulates the view.  Both synchronous. */
int mdx_gbam_view_flags(mdx_gbam *g, uint16_t *flags, int64_t n);
int mdx_gbam_view_set_flags(mdx_gbam *g, const uint16_t *flags, int64_t n);
void mdx_gbam_close(mdx_gbam *g);
/* The host beside the device.  mdx_host_threads: the threads this process inflates BGZF blocks on (the host's share of a slab,
 * mdx_gbam_next; the pool starts with the first such slab and keeps its size) — half of the hardware threads, at most 128 and
 * at most what the control group's cpu.max grants less two, divided by LOCAL_WORLD_SIZE (the ranks of this node, one per GPU,
 * SURVEY 8e, inflate at the same time); MDX_GBAM_HOST_THREADS overrides, MDX_CPU_MAX_FILE names a stand-in for
 * /sys/fs/cgroup/cpu.max.  mdx_host_pool_threads: starts the pool if need be and returns its size.
 * mdx_warm: what a process pays once whichever file comes first — the device's context, the decode kernels' code object, the
 * pool, a pinned buffer of pinned_bytes for the host's share — for a helper thread at the start of a run (the command line
 * warms up beside its header and index reads).  The reference has no counterpart (pysam opens a file in microseconds). */
int mdx_host_threads(void);
int mdx_host_pool_threads(void);
int mdx_warm(int32_t device, int64_t pinned_bytes);
/* Introspection for tests: the device inflate and CRC32 stages of the decode path alone, on BGZF payloads the caller
 * supplies (host buffers).  blk holds four words per block — payload offset in comp, payload bytes, offset in out,
 * bytes out (ISIZE, at most 65536) — and want_crc the CRC32 of each block's inflated bytes (may be NULL: no check).
 * status[b] = bytes produced, or a negative inflate code (-4: ISIZE disagrees); crc_ok[b] = 1 when the check passed.
 * tests/test_gpu_decode.py feeds it every deflate block type, window distances across the LDS ring and damaged
 * streams, against zlib. */
int mdx_gbam_inflate_blocks(mdx_ctx *ctx, const uint8_t *comp, int64_t comp_bytes, const uint32_t *blk, int32_t n_blocks,
                            uint8_t *out, int64_t out_bytes, int32_t *status, const uint32_t *want_crc, uint8_t *crc_ok);

#ifdef __cplusplus
}
#endif
#endif /* MDX_H */
