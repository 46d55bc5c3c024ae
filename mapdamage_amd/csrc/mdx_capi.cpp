// C-ABI layer of libmdx.so (declared in include/mdx.h): context, device memory, launches.
// No torch, no C++ types across the boundary, no exceptions escape.
#include "../../include/mdx.h"
#include "mdx_internal.h"

#include <dlfcn.h>

#include <chrono>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace {

constexpr size_t kLdsLimit = 160 * 1024;  // MI355X: 160 KiB LDS per CU
#ifdef MDX_WAVE_CLK
static unsigned long long *g_dbg_clk = nullptr;
extern "C" int mdx_dbg_clk_read(unsigned long long *out, int n) {
    if (!g_dbg_clk) return -1;
    (void)hipDeviceSynchronize();
    return (int)hipMemcpy(out, g_dbg_clk, (size_t)n * 24, hipMemcpyDeviceToHost);
}
// (-DMDX_PHASE_CLK: twelve sums of shader-clock ticks over all wavefronts since the last reset, tools/experiments/phase_clk.py)
extern "C" int mdx_dbg_phase_read(unsigned long long *out, int reset) {
    if (!g_dbg_clk) return -1;
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)16 * 8192);
    int rc = (int)hipMemcpy(h.data(), g_dbg_clk + 3 * 20000, h.size() * 8, hipMemcpyDeviceToHost);
    for (int q = 0; q < 16; q++) out[q] = 0;
    for (size_t w = 0; w < 8192; w++) for (int q = 0; q < 16; q++) out[q] += h[16 * w + q];
    if (reset) rc |= (int)hipMemset(g_dbg_clk + 3 * 20000, 0, h.size() * 8);
    return rc;
}
#endif
constexpr int kLgdLds = 256;           // fragment lengths below this are counted in the LDS

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

// RCCL entry points, resolved with dlopen on first use (a process that never reduces across GPUs does not need
// the library; when torch has already loaded its own librccl.so the same image is reused)
namespace {
struct RcclId { char internal[MDX_COMM_ID_BYTES]; };
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(RcclId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
};
constexpr int kNcclInt64 = 4, kNcclUint64 = 5, kNcclSum = 0;
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl *rccl() {
    std::call_once(g_rccl_once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.handle) break;
        }
        if (!g_rccl.handle) { g_rccl.err = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : ""); return; }
        auto sym = [&](const char *n) { void *p = dlsym(g_rccl.handle, n); if (!p) g_rccl.err = std::string("librccl: missing ") + n; return p; };
        g_rccl.GetUniqueId = (int (*)(RcclId *))sym("ncclGetUniqueId");
        g_rccl.CommInitRank = (int (*)(void **, int, RcclId, int))sym("ncclCommInitRank");
        g_rccl.CommDestroy = (int (*)(void *))sym("ncclCommDestroy");
        g_rccl.CommCount = (int (*)(void *, int *))sym("ncclCommCount");
        g_rccl.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))sym("ncclAllReduce");
        g_rccl.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))sym("ncclAllGather");
        g_rccl.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
    });
    return g_rccl.err.empty() ? &g_rccl : nullptr;
}
}  // namespace

struct mdx_ctx {
    void *comm = nullptr;      // ncclComm_t
    bool comm_owned = false;
    int comm_size = 0, comm_rank = 0;
    mdx_config cfg{};
    MdxDims dims{};
    int mode = MDX_MODE_LDS;
    int n_cu = 256;
    int max_grid = 0;
    int lib_group = 0;     // LDS mode: libraries per launch (as many as fit the LDS; all of them in one launch if they do)
    size_t lds_bytes = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // reference
    uint8_t *d_ref = nullptr;
    uint8_t *d_ref4 = nullptr;     // the same bases as 4-bit codes, guard bands included (the packed kernel's)
    bool ref2 = false;             // ... twice, the second copy 2 GiB + 64 bytes behind the first (MdxTabArgs::ref2)
    int64_t *d_contig_off = nullptr;
    int n_contig = 0;
    int64_t ref_len = 0;
    // accumulators
    unsigned long long *d_raw = nullptr;
    unsigned long long *d_lgd_dense = nullptr;
    long long *d_lgd_over = nullptr;
    unsigned long long *d_n_lgd_over = nullptr;
    unsigned long long *d_err = nullptr;
    uint32_t *d_partials = nullptr;
    // staging for mdx_tabulate_host: device columns, and two pinned bounce buffers the host columns go through
    // (the CPU fills one while the DMA engine drains the other)
    DevBuf st[2][10];      // two sets: the columns of batch k+1 are copied while the kernel of batch k reads its own
    hipStream_t copy_stream = nullptr;
    hipEvent_t st_copied[2] = {nullptr, nullptr}, st_done[2] = {nullptr, nullptr};
    bool st_busy[2] = {false, false};
    int st_turn = 0;
    void *gbam_arena = nullptr;      // the device decode's arena and CRC tables, kept between files (mdx_ctx_scratch_*)
    size_t gbam_arena_cap = 0;
    void *gbam_tables = nullptr;
    void *gbam_stream = nullptr, *gbam_event = nullptr;
    int64_t record_base = 0;   // added to the batch index of a record in the error word (mdx_set_record_base)
    DevBuf unpacked;       // ASCII copy of a 4-bit SEQ column, for the launches the packed kernel does not take
    DevBuf lists;          // per-wavefront entry lists of the tabulation kernel (MdxTabArgs::lists)
    DevBuf rs_part;        // per-block summary counters of the rescale kernel (MdxRescaleArgs::subs_part)
    DevBuf rs_lists;       // per-wavefront lists of the records left to rescale_walk_kernel (MdxRescaleArgs::gen_list)
    DevBuf rs_in;          // fused launch: per-wavefront lists of the records left to the rescale kernels (MdxFuse::gen_list)
    uint32_t *d_tile_ctr = nullptr; // tile counters of the fast kernels' pools (MdxTabArgs::tile_ctr)
    size_t fuse_prepared = 0;      // LDS bytes the fused kernel has been prepared for
    size_t pkf_prepared = 0;       // ... and the packed fused kernel
    MdxPkConfig pk = {512, 0};     // the packed kernels' block and whether they prefetch a tile's columns into the LDS (mdx_k_pk_config)
    bool tile_ctr_clean = false;   // d_tile_ctr is all zero (the reduction behind a launch leaves it so)
    bool pkm_prepared = false;     // the packed kernel's masked form
    DevBuf lowq;           // --min-basequal, packed kernel: the scratch column a MDX_SEQ_4BIT batch's mask is folded into (MDX_SEQ_4BITQ)
    DevBuf libsort;        // several libraries, packed kernel: the batch's columns bucketed by library (a batch that does not bring them)
    DevBuf libsort_scratch;
    const void *libsort_checked = nullptr;   // the batch's own blob (mdx_batch::libsort) whose signature was read back last
    uint64_t libsort_checked_sig = 0;        // ... and that signature
    DevBuf ml_partials;    // ... and the plan of a launch over several libraries: which library a pool of blocks counts (MdxTabArgs::ml_plan)
    int64_t n_libsorts = 0;        // sorts done inside a launch so far (a resident batch brings its own: mdx_batch::libsort)
    int64_t n_fused = 0;           // fused launches so far (mdx_fused_launches)
    int64_t n_packed = 0;          // launches of the packed kernel so far (mdx_packed_launches)
    int64_t fuse_list_cap = 0;     // entries per list of rs_in (the last fused launch)
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t pin_done[2] = {nullptr, nullptr};
    bool pin_busy[2] = {false, false};
    // rescale model (mdx_rescale_set_model)
    uint8_t *d_lut = nullptr;
    double *d_term = nullptr;
    unsigned long long *d_subs = nullptr;   // rescale summary counters
    std::vector<uint8_t> h_lut;             // host copy of the LUT (the summary's "after" histograms follow from it)
    bool key0_plain = false;                // key 0 keeps every quality and adds 0.0 to MR
    int len5p = 0, len3p = 0;
    // timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;      // tabulation kernel
    std::vector<std::pair<hipEvent_t, hipEvent_t>> rs_events;   // rescale kernel
    std::string err;
};

namespace {

int fail(mdx_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return code;
}

#define RCCL_TRY(ctx, r, call)                                                                  \
    do {                                                                                        \
        int e_ = (call);                                                                        \
        if (e_ != 0)                                                                            \
            return fail((ctx), MDX_ERR_COMM, std::string(#call) + ": " + (r)->GetErrorString(e_)); \
    } while (0)

#define HIP_TRY(ctx, call)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail((ctx), MDX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

int64_t lgd_words(const mdx_ctx *c) { return (int64_t)c->cfg.nlib * 4 * c->cfg.lgd_max; }
int64_t mis_words(const mdx_ctx *c) { return (int64_t)c->cfg.nlib * 4 * c->cfg.length * MDX_N_MIS_COLS; }
int64_t comp_words(const mdx_ctx *c) { return (int64_t)c->cfg.nlib * 4 * (c->cfg.length + c->cfg.around) * 4; }

int zero_accumulators(mdx_ctx *c) {
    HIP_TRY(c, hipMemsetAsync(c->d_raw, 0, (size_t)c->dims.w_total * 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_lgd_dense, 0, (size_t)lgd_words(c) * 8 * MDX_LGD_COPIES, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_n_lgd_over, 0, 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_err, 0xFF, 8, c->stream));
    return MDX_OK;
}

}  // namespace

extern "C" {

int mdx_abi_version(void) { return MDX_ABI_VERSION; }

int mdx_pack_seq(const uint8_t *ascii, int64_t n_bases, uint8_t *packed, int32_t threads) {
    if (n_bases < 0 || (n_bases > 0 && (!ascii || !packed))) return MDX_ERR_ARG;
    // exactly 'A', 'C', 'T', 'G' -> 1, 2, 4, 8 (bit k = symbol class k, the order of (ascii >> 1) & 3); anything else 0
    uint8_t lut[256];
    std::memset(lut, 0, sizeof lut);
    lut['A'] = 1; lut['C'] = 2; lut['T'] = 4; lut['G'] = 8;
    const int64_t nb = (n_bases + 1) / 2;
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nb < (int64_t)1 << 20) nt = 1;
    auto work = [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; i++) {
            const uint8_t a = lut[ascii[2 * i]], b = 2 * i + 1 < n_bases ? lut[ascii[2 * i + 1]] : 0;
            packed[i] = (uint8_t)(a | (b << 4));
        }
    };
    if (nt == 1) { work(0, nb); return MDX_OK; }
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++) pool.emplace_back(work, nb * t / nt, nb * (t + 1) / nt);
    for (auto &th : pool) th.join();
    return MDX_OK;
}

const char *mdx_strerror(int code) {
    switch (code) {
        case MDX_OK: return "ok";
        case MDX_ERR_ARG: return "invalid argument";
        case MDX_ERR_HIP: return "HIP runtime error";
        case MDX_ERR_STATE: return "invalid call order";
        case MDX_ERR_MASK_INDEX: return "masked column beyond gapped reference";
        case MDX_ERR_LGD_OVERFLOW: return "fragment-length overflow list full";
        case MDX_ERR_BAD_READ: return "record cannot be processed (alignment past contig end, bad tid/library, CIGAR/SEQ mismatch)";
        case MDX_ERR_COMM: return "RCCL failure, or another rank reported an error";
    }
    return "unknown error";
}

int mdx_create(const mdx_config *cfg, mdx_ctx **out) {
    if (!cfg || !out) return MDX_ERR_ARG;
    *out = nullptr;
    if (cfg->length < 1 || cfg->around < 0 || cfg->minqual < 0 || cfg->minqual > 93 || cfg->nlib < 1 ||
        cfg->nlib > 65535 || cfg->lgd_max < 1 || cfg->lgd_over_cap < 0)
        return MDX_ERR_ARG;
    mdx_ctx *c = new (std::nothrow) mdx_ctx();
    if (!c) return MDX_ERR_ARG;
    c->cfg = *cfg;
    // (MDX_INIT_TRACE=1: what each step took, on stderr)
    const bool trace = getenv("MDX_INIT_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "mdx_create %-14s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
        delete c;
        return MDX_ERR_HIP;
    }
    *out = c;  // from here on the caller destroys it, also on error (mdx_last_error stays readable)
    HIP_TRY(c, hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIP_TRY(c, hipGetDeviceProperties(&prop, cfg->device));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    lap("context, stream");

    int lgd_lds = cfg->lgd_max < kLgdLds ? cfg->lgd_max : kLgdLds;
    // two blocks per CU need half of the LDS each: the short-fragment histogram gives way first (lengths beyond it
    // take the dense histogram's global atomics) — at --length 70 --around 10 it ends up at 226 entries instead of 256
    while (lgd_lds > 128 && mdx_k_lds_bytes(mdx_make_dims(cfg->length, cfg->around, 1, cfg->lgd_max, lgd_lds)) > kLdsLimit / 2 &&
           mdx_k_lds_bytes(mdx_make_dims(cfg->length, cfg->around, 1, cfg->lgd_max, 128)) <= kLdsLimit / 2)
        lgd_lds -= 2;
    if ((int64_t)cfg->nlib * (4LL * cfg->length * 29 + 8LL * (2LL * cfg->around + 512) +
                              4LL * lgd_lds + 128) > 0x7FFFFFF0LL || cfg->length > (1 << 24) || cfg->around > (1 << 24))
        return fail(c, MDX_ERR_ARG, "table too large (nlib * length)");
    c->dims = mdx_make_dims(cfg->length, cfg->around, cfg->nlib, cfg->lgd_max, lgd_lds);
    // the tables of as many libraries as fit the LDS are counted per launch (all of them in one launch if they
    // fit); not even one fits: global-atomic fallback
    int group = cfg->nlib;
    while (group > 1 && mdx_k_lds_bytes(mdx_make_dims(cfg->length, cfg->around, group, cfg->lgd_max, lgd_lds)) > kLdsLimit)
        group--;
    const MdxDims gdims = mdx_make_dims(cfg->length, cfg->around, group, cfg->lgd_max, lgd_lds);
    c->lds_bytes = mdx_k_lds_bytes(gdims);
    if (c->lds_bytes <= kLdsLimit) {
        c->mode = MDX_MODE_LDS;
        c->lib_group = group;
        c->max_grid = c->n_cu * (2048 / mdx_k_block_threads());
        HIP_TRY(c, mdx_k_prepare(c->lds_bytes));
        // (the packed kernel counts one library per launch)
        {
            const MdxDims d1 = mdx_make_dims(cfg->length, cfg->around, 1, cfg->lgd_max, lgd_lds);
            c->pk = mdx_k_pk_config(d1, kLdsLimit);
            HIP_TRY(c, mdx_k_prepare_packed(mdx_k_pk_lds_bytes(d1, c->pk)));
            if (const char *e = getenv("MDX_DEBUG_PK")) if (*e && *e != '0')
                fprintf(stderr, "[mdx] packed kernels: blocks of %d threads, %zu bytes of LDS, columns %s\n", c->pk.threads, mdx_k_pk_lds_bytes(d1, c->pk),
                        c->pk.pfl ? "prefetched into the LDS" : "by plain loads");
        }
        lap("kernels");
        HIP_TRY(c, hipMalloc((void **)&c->d_partials, (size_t)c->max_grid * gdims.w_total * 4));
    } else {
        c->mode = MDX_MODE_GLOBAL;  // tables do not fit the LDS: global-atomic fallback
        c->max_grid = c->n_cu * (2048 / mdx_k_block_threads());
    }
    HIP_TRY(c, hipMalloc((void **)&c->d_raw, (size_t)c->dims.w_total * 8));
    HIP_TRY(c, hipMalloc((void **)&c->d_lgd_dense, (size_t)lgd_words(c) * 8 * MDX_LGD_COPIES));
    HIP_TRY(c, hipMalloc((void **)&c->d_lgd_over, (size_t)(cfg->lgd_over_cap > 0 ? cfg->lgd_over_cap : 1) * 32));
    HIP_TRY(c, hipMalloc((void **)&c->d_n_lgd_over, 8));
    HIP_TRY(c, hipMalloc((void **)&c->d_err, 8));
    int rc = zero_accumulators(c);
    if (rc != MDX_OK) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    lap("accumulators");
    return MDX_OK;
}

void mdx_destroy(mdx_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && c->comm_owned && rccl()) (void)rccl()->CommDestroy(c->comm);
    for (auto &ev : c->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto &ev : c->rs_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto &set : c->st) for (auto &b : set) b.release();
    for (int i = 0; i < 2; i++) {
        if (c->st_copied[i]) (void)hipEventDestroy(c->st_copied[i]);
        if (c->st_done[i]) (void)hipEventDestroy(c->st_done[i]);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->gbam_arena) (void)hipFree(c->gbam_arena);
    if (c->gbam_tables) (void)hipFree(c->gbam_tables);
    if (c->gbam_stream) (void)hipStreamDestroy((hipStream_t)c->gbam_stream);
    if (c->gbam_event) (void)hipEventDestroy((hipEvent_t)c->gbam_event);
    c->lists.release();
    c->unpacked.release();
    c->lowq.release();
    c->libsort.release();
    c->libsort_scratch.release();
    c->ml_partials.release();
    c->rs_part.release();
    c->rs_lists.release();
    c->rs_in.release();
    for (int i = 0; i < 2; i++) {
        if (c->pin[i]) (void)hipHostFree(c->pin[i]);
        if (c->pin_done[i]) (void)hipEventDestroy(c->pin_done[i]);
    }
    void *ptrs[] = {c->d_ref, c->d_ref4, c->d_contig_off, c->d_raw, c->d_lgd_dense, c->d_lgd_over,
                    c->d_n_lgd_over, c->d_err, c->d_partials, c->d_lut, c->d_term, c->d_subs, c->d_tile_ctr};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char *mdx_last_error(const mdx_ctx *c) { return c ? c->err.c_str() : "null context"; }

int mdx_set_stream(mdx_ctx *c, void *hip_stream) {
    if (!c) return MDX_ERR_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return MDX_OK;
}

// the resident reference of n bases: the old one released, the new one allocated with its guard bands (a guard band on both
// sides keeps speculative flank addresses inside the allocation) and filled with the "other symbol" code
static const size_t kRefPad = 256;
static int ref_begin(mdx_ctx *c, int64_t n, int32_t n_contig) {
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_ref) { (void)hipFree(c->d_ref); c->d_ref = nullptr; }
    if (c->d_ref4) { (void)hipFree(c->d_ref4); c->d_ref4 = nullptr; }
    if (c->d_contig_off) { (void)hipFree(c->d_contig_off); c->d_contig_off = nullptr; }
    c->n_contig = 0; c->ref_len = 0;
    HIP_TRY(c, hipMalloc((void **)&c->d_ref, (size_t)n + 2 * kRefPad + 1));
    HIP_TRY(c, hipMalloc((void **)&c->d_contig_off, (size_t)(n_contig + 1) * 8));
    HIP_TRY(c, hipMemsetAsync(c->d_ref, 0x85, (size_t)n + 2 * kRefPad + 1, c->stream));
    return MDX_OK;
}
// ... and, the bases in place: the contig offsets and the 4-bit form of the same bytes, guard bands included (an even number
// of them: one more guard byte behind an odd genome)
static int ref_finish(mdx_ctx *c, const int64_t *contig_off, int32_t n_contig, int64_t n) {
    const size_t n4 = ((size_t)n + 2 * kRefPad + 1) / 2;
    // (MdxTabArgs::ref2 — a genome whose 4-bit form is past what the caches hold, 64 MB and more: a second copy 2 GiB + 64 bytes
    // behind the first, for the packed kernels to pick from per record; both within the 4 GiB a 32-bit byte offset reaches.
    // MDX_NO_REF2=1: one copy, for A/B runs)
    static const bool no_ref2 = [] { const char *e = getenv("MDX_NO_REF2"); return e && *e && *e != '0'; }();
    const size_t kCopyB = ((size_t)1 << 31) + 64;
    // (MDX_REF2_MIN=bytes: the size of the 4-bit form from which there are two — the tests' way to the second copy on a small genome)
    static const size_t ref2_min = [] { const char *e = getenv("MDX_REF2_MIN"); return e && *e ? (size_t)strtoull(e, nullptr, 10) : (size_t)64 << 20; }();
    c->ref2 = n4 >= ref2_min && n4 + 128 <= ((size_t)1 << 31) - 64 && !no_ref2;
    hipError_t e = hipMemcpyAsync(c->d_contig_off, contig_off, (size_t)(n_contig + 1) * 8, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        e = hipMalloc((void **)&c->d_ref4, (c->ref2 ? kCopyB : 0) + n4 + 64);
        // (no room for the gap: one copy)
        if (e != hipSuccess && c->ref2) { (void)hipGetLastError(); c->ref2 = false; e = hipMalloc((void **)&c->d_ref4, n4 + 64); }
    }
    if (e == hipSuccess) e = hipMemsetAsync(c->d_ref4, 0, n4 + 64, c->stream);
    if (e == hipSuccess) { mdx_k_encode_ref4(c->d_ref, c->d_ref4, (int64_t)(2 * n4), c->stream); e = hipGetLastError(); }
    if (e == hipSuccess && c->ref2) e = hipMemcpyAsync(c->d_ref4 + kCopyB, c->d_ref4, n4 + 64, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        // (a reference half in place is none: the launches ask for d_ref)
        (void)hipFree(c->d_ref); c->d_ref = nullptr;
        return fail(c, MDX_ERR_HIP, std::string("set_reference: ") + hipGetErrorString(e));
    }
    c->n_contig = n_contig;
    c->ref_len = n;
    return MDX_OK;
}

int mdx_set_reference(mdx_ctx *c, const uint8_t *bases, const int64_t *contig_off, int32_t n_contig) {
    if (!c || !contig_off || n_contig < 1) return fail(c, MDX_ERR_ARG, "set_reference: bad arguments");
    for (int i = 0; i < n_contig; i++)
        if (contig_off[i + 1] < contig_off[i]) return fail(c, MDX_ERR_ARG, "contig_off not monotone");
    const int64_t n = contig_off[n_contig];
    if (contig_off[0] != 0 || (n > 0 && !bases)) return fail(c, MDX_ERR_ARG, "set_reference: bad arguments");
    int rc = ref_begin(c, n, n_contig);
    if (rc != MDX_OK) return rc;
    uint8_t *tmp = nullptr;
    HIP_TRY(c, hipMalloc((void **)&tmp, (size_t)n + 1));
    hipError_t e = n > 0 ? hipMemcpyAsync(tmp, bases, (size_t)n, hipMemcpyHostToDevice, c->stream) : hipSuccess;
    if (e == hipSuccess) { mdx_k_encode_ref(tmp, c->d_ref + kRefPad, n, c->stream); e = hipGetLastError(); }
    rc = e == hipSuccess ? ref_finish(c, contig_off, n_contig, n) : fail(c, MDX_ERR_HIP, std::string("set_reference: ") + hipGetErrorString(e));
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    return rc;
}

int mdx_set_reference_fasta(mdx_ctx *c, const char *fasta_path, int32_t n_contig, const char *const *names, int32_t missing_ok,
                            int64_t *lengths) {
    if (!c || !fasta_path || !names || n_contig < 1) return fail(c, MDX_ERR_ARG, "set_reference_fasta: bad arguments");
    try {
        HIP_TRY(c, hipSetDevice(c->cfg.device));
        struct Arg { mdx_ctx *c; int32_t n_contig; int rc; } arg{c, n_contig, MDX_OK};
        std::vector<int64_t> contig_off;
        std::string err;
        const int rc = mdx_fasta_to_device(fasta_path, n_contig, names, missing_ok, lengths, contig_off, err, c->stream,
            [](void *a_, int64_t n) -> uint8_t * {
                Arg *a = (Arg *)a_;
                a->rc = ref_begin(a->c, n, a->n_contig);
                return a->rc == MDX_OK ? a->c->d_ref + kRefPad : nullptr;
            }, &arg);
        if (arg.rc != MDX_OK) return arg.rc;
        if (rc != MDX_OK) {
            // (a reference half in place is none)
            if (c->d_ref) { (void)hipFree(c->d_ref); c->d_ref = nullptr; }
            return fail(c, rc, "set_reference_fasta: " + err);
        }
        return ref_finish(c, contig_off.data(), n_contig, contig_off[(size_t)n_contig]);
    } catch (const std::exception &e) {
        return fail(c, MDX_ERR_ARG, std::string("set_reference_fasta: ") + e.what());
    }
}

// Introspection for tests: bases [start, end) of contig `tid` of the resident reference as the kernels see them — 'A' 'C' 'G'
// 'T' where ref.fetch(chrom, start, end).upper() (main.py:180) holds one of the four, '-' for '-', 'N' for anything else.
int mdx_reference_fetch(mdx_ctx *c, int32_t tid, int64_t start, int64_t end, uint8_t *out) {
    if (!c || !out) return MDX_ERR_ARG;
    if (!c->d_ref) return fail(c, MDX_ERR_STATE, "mdx_set_reference has not been called");
    if (tid < 0 || tid >= c->n_contig || start < 0 || end < start) return fail(c, MDX_ERR_ARG, "reference_fetch: bad range");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    int64_t off[2];
    HIP_TRY(c, hipMemcpy(off, c->d_contig_off + tid, 16, hipMemcpyDeviceToHost));
    if (start + off[0] > off[1] || end + off[0] > off[1]) return fail(c, MDX_ERR_ARG, "reference_fetch: beyond the contig's end");
    if (end > start) HIP_TRY(c, hipMemcpy(out, c->d_ref + kRefPad + off[0] + start, (size_t)(end - start), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < end - start; i++) if (out[i] & 0x80) out[i] = out[i] == 0x84 ? '-' : 'N';
    return MDX_OK;
}

static int check_batch(mdx_ctx *c, const mdx_batch *b) {
    if (!c || !b) return MDX_ERR_ARG;
    if (b->n_reads < 0 || b->n_cigar < 0 || b->n_bases < 0) return fail(c, MDX_ERR_ARG, "negative batch size");
    if (b->n_bases > 0xFFFFFFFFLL || b->n_cigar > 0xFFFFFFFFLL)
        return fail(c, MDX_ERR_ARG, "batch exceeds 32-bit offsets; split it");
    if (b->n_reads > 0 && (!b->flag || !b->lib || !b->tid || !b->pos || !b->tlen || !b->cigar_off || !b->seq_off))
        return fail(c, MDX_ERR_ARG, "null column");
    if ((b->n_cigar > 0 && !b->cigar) || (b->n_bases > 0 && !b->seq)) return fail(c, MDX_ERR_ARG, "null column");
    if (b->seq_format != MDX_SEQ_ASCII && b->seq_format != MDX_SEQ_4BIT && b->seq_format != MDX_SEQ_4BITQ) return fail(c, MDX_ERR_ARG, "unknown seq_format");
    if (b->seq_format == MDX_SEQ_4BITQ && c->cfg.minqual == 0)
        return fail(c, MDX_ERR_ARG, "a MDX_SEQ_4BITQ column carries the --min-basequal of the context that made it; this context has none");
    return MDX_OK;
}

// bytes of the seq column in its form
static size_t seq_bytes(const mdx_batch *b) {
    return b->seq_format != MDX_SEQ_ASCII ? ((size_t)b->n_bases + 1) / 2 : (size_t)b->n_bases;
}

// For the launches that read ASCII: a batch whose SEQ column is 4-bit gets an ASCII copy in the context's scratch
// column (enqueued on the stream; valid until the next such call).  *out = the batch to launch with.
static int ascii_view(mdx_ctx *c, const mdx_batch *b, mdx_batch *out) {
    *out = *b;
    if (b->seq_format == MDX_SEQ_ASCII || b->n_bases == 0) { out->seq_format = MDX_SEQ_ASCII; return MDX_OK; }
    // (launches are in stream order: the kernels of the previous call are done with the scratch column when this one writes it)
    HIP_TRY(c, c->unpacked.reserve((size_t)b->n_bases + 64));
    mdx_k_unpack_seq(b->seq, (uint8_t *)c->unpacked.p, b->n_bases, c->stream);
    HIP_TRY(c, hipGetLastError());
    out->seq = (const uint8_t *)c->unpacked.p;
    out->seq_format = MDX_SEQ_ASCII;
    return MDX_OK;
}

// What a blob of mdx_k_libsort was laid out for, in the spare word of its header: the layout is a function of the batch's
// sizes and of the libraries of the context that built it — a resident batch handed to a context of another nlib, or a
// sub-view of it, would be read through the wrong offsets.
static uint64_t libsort_signature(int64_t n, int64_t n_cigar, int64_t n_bases, int nlib) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (uint64_t v : {(uint64_t)n, (uint64_t)n_cigar, (uint64_t)n_bases, (uint64_t)nlib}) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); h *= 0xBF58476D1CE4E5B9ull; }
    return h | 1ull;
}
// A device batch with a 4-bit SEQ column (either form), ordered by library, into `blob` (mdx_k_libsort_bytes; enqueued on
// the stream)
static int build_libsort(mdx_ctx *c, const mdx_batch *b, void *blob) {
    MdxLibSort ls;
    mdx_k_libsort_layout(blob, b->n_reads, b->n_cigar, b->n_bases, c->cfg.nlib, &ls);
    const uint64_t sig = libsort_signature(b->n_reads, b->n_cigar, b->n_bases, c->cfg.nlib);
    HIP_TRY(c, hipMemcpyAsync((char *)blob + 8, &sig, 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, c->libsort_scratch.reserve(mdx_k_libsort_scratch_bytes(b->n_reads, c->cfg.nlib)));
    mdx_k_libsort(b->n_reads, b->n_cigar, b->n_bases, b->flag, b->lib, b->tid, b->pos, b->tlen, b->cigar_off, b->cigar, b->seq_off, b->seq,
                  c->cfg.nlib, c->libsort_scratch.p, ls, c->stream);
    HIP_TRY(c, hipGetLastError());
    return MDX_OK;
}

int mdx_batch_upload(mdx_ctx *c, const mdx_batch *h, mdx_batch *dv) {
    int rc = check_batch(c, h);
    if (rc != MDX_OK) return rc;
    if (!dv) return MDX_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    std::memset(dv, 0, sizeof(*dv));
    dv->n_reads = h->n_reads; dv->n_cigar = h->n_cigar; dv->n_bases = h->n_bases;
    dv->seq_format = h->seq_format;
    const int64_t n = h->n_reads;
    struct Col { const void *src; const void **dst; size_t bytes; } cols[] = {
        {h->flag, (const void **)&dv->flag, (size_t)n * 2},
        {h->lib, (const void **)&dv->lib, (size_t)n * 2},
        {h->tid, (const void **)&dv->tid, (size_t)n * 4},
        {h->pos, (const void **)&dv->pos, (size_t)n * 4},
        {h->tlen, (const void **)&dv->tlen, (size_t)n * 4},
        {h->cigar_off, (const void **)&dv->cigar_off, (size_t)(n + 1) * 4},
        {h->cigar, (const void **)&dv->cigar, (size_t)h->n_cigar * 4},
        {h->seq_off, (const void **)&dv->seq_off, (size_t)(n + 1) * 4},
        {h->seq, (const void **)&dv->seq, seq_bytes(h)},
        {h->qual, (const void **)&dv->qual, h->qual ? (size_t)h->n_bases : 0},
    };
    for (auto &col : cols) {
        if (!col.src || n == 0) continue;
        void *p = nullptr;
        HIP_TRY(c, hipMalloc(&p, col.bytes + 64));
        *col.dst = p;
        HIP_TRY(c, hipMemcpyAsync(p, col.src, col.bytes, hipMemcpyHostToDevice, c->stream));
    }
    // a batch with qualities: the records that have them say so in their flags (MDX_FLAG_HAS_QUAL: the rescaling kernels then
    // route a record without a look at its first quality byte)
    if (dv->qual && n > 0) {
        mdx_k_mark_has_qual(const_cast<uint16_t *>(dv->flag), dv->seq_off, dv->qual, n, c->stream);
        HIP_TRY(c, hipGetLastError());
    }
    // --min-basequal and a 4-bit SEQ column: the mask goes into the resident column itself (MDX_SEQ_4BITQ, include/mdx.h) —
    // the packed masked kernel then reads no quality, and nothing is folded in front of the launches
    // (MDX_NO_BATCH_LOWQ=1 in the environment: not — every launch folds into a scratch column, for A/B runs)
    static const bool no_batch_lowq = [] { const char *e = getenv("MDX_NO_BATCH_LOWQ"); return e && *e && *e != '0'; }();
    if (c->cfg.minqual > 0 && dv->qual && h->seq_format == MDX_SEQ_4BIT && h->n_bases > 0 && c->mode == MDX_MODE_LDS && !no_batch_lowq) {
        mdx_k_fold_mask(dv->seq, const_cast<uint8_t *>(dv->seq), dv->qual, nullptr, h->n_bases, c->cfg.minqual, c->stream);
        HIP_TRY(c, hipGetLastError());
        dv->seq_format = MDX_SEQ_4BITQ;
    }
    // several libraries and a 4-bit SEQ column: the batch ordered by library travels with the resident batch
    // (mdx_batch::libsort) instead of being sorted in front of every launch
    // (MDX_NO_BATCH_LIBSORT=1 in the environment: not, for A/B runs)
    static const bool no_batch_sort = [] { const char *e = getenv("MDX_NO_BATCH_LIBSORT"); return e && *e && *e != '0'; }();
    if (c->cfg.nlib > 1 && h->seq_format != MDX_SEQ_ASCII && n > 0 && c->mode == MDX_MODE_LDS && !no_batch_sort) {
        void *p = nullptr;
        HIP_TRY(c, hipMalloc(&p, mdx_k_libsort_bytes(n, h->n_cigar, h->n_bases, c->cfg.nlib)));
        dv->libsort = (const uint8_t *)p;
        // (a record with a library the context does not know has no place: the blob remembers the first, the launches report it)
        rc = build_libsort(c, dv, p);
        if (rc != MDX_OK) return rc;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MDX_OK;
}

int mdx_batch_fold(mdx_ctx *c, mdx_batch *b) {
    int rc = check_batch(c, b);
    if (rc != MDX_OK) return rc;
    if (c->cfg.minqual <= 0) return fail(c, MDX_ERR_STATE, "mdx_batch_fold: the context has no --min-basequal");
    if (b->seq_format == MDX_SEQ_4BITQ) return MDX_OK;
    if (b->seq_format != MDX_SEQ_4BIT) return fail(c, MDX_ERR_ARG, "mdx_batch_fold: the seq column is not MDX_SEQ_4BIT");
    if (!b->qual && !b->lowq) return fail(c, MDX_ERR_ARG, "mdx_batch_fold: neither qualities nor a bitmap of the low ones");
    // (a copy of the batch ordered by library was made from the unmasked column)
    if (b->libsort) return fail(c, MDX_ERR_ARG, "mdx_batch_fold: the batch brings mdx_batch::libsort, built from the column as it was");
    if (b->n_bases == 0) { b->seq_format = MDX_SEQ_4BITQ; return MDX_OK; }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    mdx_k_fold_mask(b->seq, const_cast<uint8_t *>(b->seq), b->qual, b->lowq, b->n_bases, c->cfg.minqual, c->stream);
    HIP_TRY(c, hipGetLastError());
    b->seq_format = MDX_SEQ_4BITQ;
    return MDX_OK;
}

int mdx_batch_free(mdx_ctx *c, mdx_batch *dv) {
    if (!c || !dv) return MDX_ERR_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (dv->libsort && dv->libsort == c->libsort_checked) c->libsort_checked = nullptr;
    const void *ptrs[] = {dv->flag, dv->lib, dv->tid, dv->pos, dv->tlen, dv->cigar_off,
                          dv->cigar, dv->seq_off, dv->seq, dv->qual, dv->lowq, dv->libsort};
    for (const void *p : ptrs) if (p) (void)hipFree(const_cast<void *>(p));
    std::memset(dv, 0, sizeof(*dv));
    return MDX_OK;
}

// fuse: the rescale side of the fused launch (mdx_tabulate_rescale_device), or null
static int tabulate_impl(mdx_ctx *c, const mdx_batch *b_in, const MdxFuse *fuse, int *fused_grid) {
    int rc = check_batch(c, b_in);
    if (rc != MDX_OK) return rc;
    if (!c->d_ref) return fail(c, MDX_ERR_STATE, "mdx_set_reference has not been called");
    if (b_in->n_reads == 0) return MDX_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    // A 4-bit SEQ column runs through the packed kernel when the launch is the plain fast tabulation: tables in the
    // LDS, 8-base lanes, 32-bit reference offsets, no quality masking, no fused rescaling.  Anything else reads an ASCII
    // copy (MDX_NO_PACKED=1 in the environment: always, for A/B runs).
    // (MDX_FORCE_REF64=1 in the environment, tests only: every launch takes the path of a reference of 4 Gbases and more —
    // 64-bit window offsets, the generic CIGAR walk for every record — at whatever geometry)
    const bool force_ref64 = [] { const char *e = getenv("MDX_FORCE_REF64"); return e && *e && *e != '0'; }();
    const bool ref32 = c->ref_len + 1024 < (int64_t)0xFFFFFFFFLL && !force_ref64;
    static const bool no_packed = [] { const char *e = getenv("MDX_NO_PACKED"); return e && *e && *e != '0'; }();
    // (a fused launch: the packed fused kernel, one library — its caller has checked that it applies, packed_fuse_applies)
    // (--min-basequal: the packed kernel's masked form reads a bitmap of the qualities below the threshold, built in front of
    // the launch — MDX_NO_PACKED_MASK=1: the ASCII kernel instead, for A/B runs)
    static const bool no_pkm = [] { const char *e = getenv("MDX_NO_PACKED_MASK"); return e && *e && *e != '0'; }();
    // (a MDX_SEQ_4BITQ column has the mask in its nibbles: the packed masked kernel's input; a MDX_SEQ_4BIT one with qualities
    // is folded into a scratch column in front of the launch)
    const bool folded = b_in->seq_format == MDX_SEQ_4BITQ;
    const bool want_mask = c->cfg.minqual > 0 && (b_in->qual != nullptr || folded);
    const bool packed = b_in->seq_format != MDX_SEQ_ASCII && c->mode == MDX_MODE_LDS && c->dims.fast_ok() && ref32 &&
                        (fuse ? (c->cfg.nlib == 1 && !folded) : !(want_mask && no_pkm)) && !no_packed;
    const bool pmask = packed && !fuse && want_mask;
    // Several libraries through the packed kernel: ONE launch that counts the libraries side by side (a library per pool of blocks)
    // over the columns bucketed by library — the batch's own (mdx_batch::libsort, a resident batch) or sorted here, inside the
    // launch's timed region.  (MDX_NO_ML=1 in the environment: one launch per library, each over all records, for A/B runs)
    static const bool no_ml = [] { const char *e = getenv("MDX_NO_ML"); return e && *e && *e != '0'; }();
    const bool ml = packed && !fuse && c->cfg.nlib > 1 && !no_ml;
    mdx_batch b_ascii;
    const mdx_batch *b = b_in;
    if (!packed) {
        rc = ascii_view(c, b_in, &b_ascii);
        if (rc != MDX_OK) return rc;
        b = &b_ascii;
    }
    MdxTabArgs a{};
    a.ref4 = c->d_ref4;
    a.ref2 = c->ref2 ? 1 : 0;
    a.seq_packed = packed ? 1 : 0;
    a.n_reads = b->n_reads;
    a.flag = b->flag; a.lib = b->lib; a.tid = b->tid; a.pos = b->pos; a.tlen = b->tlen;
    a.cigar_off = b->cigar_off; a.cigar = b->cigar; a.seq_off = b->seq_off; a.seq = b->seq; a.qual = b->qual;
    a.ref = c->d_ref + 256;   // (kRefPad)
    a.contig_off = c->d_contig_off;
    a.n_contig = c->n_contig;
    a.minqual = c->cfg.minqual;
    a.dims = c->dims;
    a.partials = c->d_partials;
    a.raw = c->d_raw;
    a.lgd_dense = c->d_lgd_dense;
    a.lgd_over = c->d_lgd_over;
    a.lgd_over_cap = c->cfg.lgd_over_cap;
    a.n_lgd_over = c->d_n_lgd_over;
    a.err = c->d_err;
    a.record_base = c->record_base;
#ifdef MDX_WAVE_CLK
    if (!g_dbg_clk) {
        (void)hipMalloc((void **)&g_dbg_clk, 3 * 8 * 20000 + 16 * 8 * 8192);
        (void)hipMemset(g_dbg_clk, 0, 3 * 8 * 20000 + 16 * 8 * 8192);
    }
    a.dbg_clk = g_dbg_clk;
#endif
    a.stage_off = mdx_k_stage_off(c->dims);
    a.queue_off = mdx_k_queue_off(c->dims);
    a.n_bases = b->n_bases;
    a.ref32 = ref32 ? 1 : 0;
    const bool mask = c->cfg.minqual > 0 && b->qual != nullptr;
    const int wpb = (packed ? c->pk.threads : mdx_k_block_threads()) / 64;
    const int64_t ntiles = (b->n_reads + 63) / 64;
    const int64_t want = (ntiles + wpb - 1) / wpb;
    if (b->n_reads >= (int64_t)1 << 30) return fail(c, MDX_ERR_ARG, "batch of 2^30 records or more; split it");
    a.nlib_total = c->cfg.nlib;
    // LDS mode: one launch per group of libraries (usually a single one); every launch scans all records and
    // counts those of its group
    // (the packed kernel: one library per launch — its bit-sliced counters are one table's)
    int group = packed ? 1 : (c->mode == MDX_MODE_LDS ? c->lib_group : c->cfg.nlib);
    const MdxDims dims1 = mdx_make_dims(c->cfg.length, c->cfg.around, 1, c->cfg.lgd_max, c->dims.lgd_lds);
    if (ml) {
        // (a pool of blocks counts one library: as many libraries per launch as the largest launch has pools of two blocks —
        // every library of a launch must get one, ml_plan_kernel — and as the plan's arrays hold)
        group = c->cfg.nlib < MDX_ML_MAX_LIBS ? c->cfg.nlib : MDX_ML_MAX_LIBS;
        int per_cu = (int)(kLdsLimit / mdx_k_pk_lds_bytes(dims1, c->pk));
        if (per_cu > mdx_k_pk_blocks_per_cu(c->pk.threads)) per_cu = mdx_k_pk_blocks_per_cu(c->pk.threads);
        const int pools = (c->n_cu * per_cu) / 2;
        if (group > pools) group = pools > 0 ? pools : 1;
    }
    for (int lo = 0; lo < c->cfg.nlib; lo += group) {
        const int gn = c->cfg.nlib - lo < group ? c->cfg.nlib - lo : group;
        a.lib_lo = lo;
        a.n_libs = ml ? gn : 0;
        size_t lds = 0;
        int max_grid = c->max_grid;
        if (c->mode == MDX_MODE_LDS) {
            a.dims = ml ? dims1 : (gn == c->cfg.nlib ? c->dims : mdx_make_dims(c->cfg.length, c->cfg.around, gn, c->cfg.lgd_max, c->dims.lgd_lds));
            a.raw = c->d_raw + (size_t)lo * c->dims.w_lib;
            a.lgd_dense = c->d_lgd_dense + (size_t)lo * 4 * c->cfg.lgd_max;
            lds = packed ? mdx_k_pk_lds_bytes(a.dims, c->pk) : mdx_k_lds_bytes(a.dims);
            int per_cu = (int)(kLdsLimit / lds);
            // (the packed kernel: two blocks per CU at most — its registers)
            const int by_threads = packed ? mdx_k_pk_blocks_per_cu(c->pk.threads) : 2048 / mdx_k_block_threads();
            if (per_cu > by_threads) per_cu = by_threads;
            max_grid = c->n_cu * per_cu;
        }
        a.stage_off = mdx_k_stage_off(a.dims);
        a.queue_off = packed ? mdx_k_pk_queue_off(a.dims, c->pk.threads) : mdx_k_queue_off(a.dims);
        // (the fused kernels have an image of their own: no prefetch areas)
        a.pfl_off = packed && !fuse && c->pk.pfl ? mdx_k_pk_pfl_off(a.dims, c->pk.threads) : 0;
        int grid = (int)(want < max_grid ? want : max_grid);
        // (several libraries: a pool — two blocks — for every library at least)
        if (ml && grid < 2 * gn) grid = 2 * gn < max_grid ? 2 * gn : max_grid;
        // (an even number of blocks — pools of two —, never more than the launch's partial slots were sized for)
        if (ml && (grid & 1)) grid += grid + 1 <= max_grid ? 1 : -1;
        int wpb_l = wpb;
        if (fuse) {
            // one 1024-thread block per CU (mdx_k_fuse_*): the whole batch in one launch, all libraries
            const int npos = 1 + fuse->len5p + fuse->len3p;
            a.rs = *fuse;
            a.queue_off = mdx_k_fuse_queue_off(a.dims);
            if (packed) {
                a.rs.qcap = MDX_PK_QCAP;
                a.rs.tcb_off = mdx_k_pkf_tcb_off(a.dims);
                lds = mdx_k_pkf_lds_bytes(a.dims, npos);
            } else {
                a.rs.qcap = mdx_k_fuse_qcap(a.dims, npos, kLdsLimit);
                a.rs.tcb_off = mdx_k_fuse_tcb_off(a.dims, a.rs.qcap);
                lds = mdx_k_fuse_lds_bytes(a.dims, npos, a.rs.qcap);
            }
            wpb_l = mdx_k_fuse_block_threads() / 64;
            const int64_t want_f = (ntiles + wpb_l - 1) / wpb_l;
            grid = (int)(want_f < c->n_cu ? want_f : c->n_cu);
            size_t &prepared = packed ? c->pkf_prepared : c->fuse_prepared;
            if (!prepared || prepared < lds) {
                HIP_TRY(c, packed ? mdx_k_pkf_prepare(lds) : mdx_k_fuse_prepare(lds));
                prepared = lds;
            }
        }
        {
            // Tiles are handed out on demand within pools of two blocks (mdx_kernels.hip): a wavefront takes at most twice
            // its even share of its pool's tiles, and its lists hold the records of that many
            const int64_t nwaves = (int64_t)grid * wpb_l;
            const int64_t T = mdx_tile_records(a.dims, a.pfl_off != 0);
            const int64_t n_tiles = (b->n_reads + T - 1) / T;
            const int64_t n_pools = mdx_n_pools((unsigned)grid);
            // (a pool takes chunks of MDX_POOL_CHUNK tiles, the pools' chunks interleaved: at most one chunk more than its share)
            const int64_t chunk = MDX_POOL_CHUNK;
            const int64_t pool_tiles = ((n_tiles + n_pools * chunk - 1) / (n_pools * chunk)) * chunk, pool_waves = (grid / n_pools) * wpb_l;
            // (the fused kernels and the masked one-library kernel are held to a quota — their rings, and the fused kernels' lists of records
            // left to the rescale kernels, are sized by it; the others work in rounds with rings of a fixed size)
            // (a launch over several libraries works in rounds whatever else it is: its pools are as full as their libraries are
            // large, and a quota would have to know the fullest)
            const bool quota = fuse || (pmask && !ml);
            a.tile_quota = quota ? (int)(2 * ((pool_tiles + pool_waves - 1) / pool_waves) + 2) : 0x7FFFFFFF;
            a.list_cap = fuse ? (int64_t)a.tile_quota * T + 128 : 0;
            a.round_tiles = MDX_ROUND_TILES;
            a.ring_size = MDX_LIST_RING;
            if (quota) while ((int64_t)a.ring_size < (int64_t)a.tile_quota * T + 128) a.ring_size *= 2;
            if (!c->d_tile_ctr) {
                HIP_TRY(c, hipMalloc((void **)&c->d_tile_ctr, (size_t)MDX_CTR_WORDS * 4));
                HIP_TRY(c, hipMemsetAsync(c->d_tile_ctr, 0, (size_t)MDX_CTR_WORDS * 4, c->stream));
            }
            if (n_pools * MDX_CTR_PAD > MDX_CTR_WORDS / 2 || n_pools * MDX_CTR_PAD > MDX_CTR_WORDS) return fail(c, MDX_ERR_STATE, "more blocks than tile counters");
            a.tile_ctr = c->d_tile_ctr;
            {
                // (scratch of the launch: rings of 81 KB per wavefront — 330 MB for the packed kernel's 4096 wavefronts —, whatever the
                // batch; the fused kernels and the masked one-library kernel: 150 bytes per record of the batch)
                const size_t list_bytes = (size_t)nwaves * (size_t)MDX_WAVE_SCRATCH(a.ring_size) * 16;
                if (c->lists.reserve(list_bytes) != hipSuccess)
                    return fail(c, MDX_ERR_HIP, "the per-wavefront lists of this launch (" + std::to_string(list_bytes >> 20) + " MiB) could not be allocated" +
                                (quota ? ": tabulate the batch in smaller pieces" : ""));
            }
            a.lists = (uint4 *)c->lists.p;
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (c->timing) {
            HIP_TRY(c, hipEventCreate(&e0));
            HIP_TRY(c, hipEventCreate(&e1));
            HIP_TRY(c, hipEventRecord(e0, c->stream));
        }
        mdx_batch b_fold;
        const mdx_batch *bs = b;        // (the batch the sort below reads)
        if (pmask && !folded) {
            // a MDX_SEQ_4BIT column with qualities: the mask folded into a scratch copy of the column (one pass over column and
            // qualities — or the caller's bitmap of them —, inside the timed region), once for all launches of the call
            if (lo == 0) {
                HIP_TRY(c, c->lowq.reserve(((size_t)b->n_bases + 1) / 2 + 64));
                const bool bits = b->lowq != nullptr;
                if (b->libsort) return fail(c, MDX_ERR_ARG, "mdx_batch::libsort of a MDX_SEQ_4BIT batch under --min-basequal: upload the batch with the context it is tabulated with");
                mdx_k_fold_mask(b->seq, (uint8_t *)c->lowq.p, b->qual, bits ? b->lowq : nullptr, b->n_bases, c->cfg.minqual, c->stream);
                HIP_TRY(c, hipGetLastError());
            }
            a.seq = (const uint8_t *)c->lowq.p;
            b_fold = *b;
            b_fold.seq = a.seq;
            bs = &b_fold;
        }
        if (pmask && !c->pkm_prepared) {
            HIP_TRY(c, mdx_k_prepare_packed_masked(mdx_k_pk_lds_bytes(dims1, c->pk)));
            c->pkm_prepared = true;
        }
        if (ml) {
            // the batch ordered by library: its own copy (a resident batch's), or sorted now (once for all launches of the call)
            const void *blob = b->libsort;
            if (blob) {
                // (a batch's own copy: laid out for these sizes and this many libraries?  Its signature is read back once per blob.)
                if (blob != c->libsort_checked) {
                    uint64_t sig = 0;
                    HIP_TRY(c, hipMemcpyAsync(&sig, (const char *)blob + 8, 8, hipMemcpyDeviceToHost, c->stream));
                    HIP_TRY(c, hipStreamSynchronize(c->stream));
                    c->libsort_checked = blob;
                    c->libsort_checked_sig = sig;
                }
                if (c->libsort_checked_sig != libsort_signature(b->n_reads, b->n_cigar, b->n_bases, c->cfg.nlib))
                    return fail(c, MDX_ERR_ARG, "mdx_batch::libsort was built for another batch or a context with another number of libraries: "
                                                "upload the batch with the context that tabulates it, whole");
            }
            if (!blob) {
                if (lo == 0) {
                    HIP_TRY(c, c->libsort.reserve(mdx_k_libsort_bytes(b->n_reads, b->n_cigar, b->n_bases, c->cfg.nlib)));
                    rc = build_libsort(c, bs, c->libsort.p);
                    if (rc != MDX_OK) return rc;
                    c->n_libsorts++;
                }
                blob = c->libsort.p;
            }
            MdxLibSort ls;
            mdx_k_libsort_layout(const_cast<void *>(blob), b->n_reads, b->n_cigar, b->n_bases, c->cfg.nlib, &ls);
            a.flag = ls.flag; a.tid = ls.tid; a.pos = ls.pos; a.tlen = ls.tlen;
            a.cigar_off = ls.cigar_off; a.cigar = ls.cigar; a.seq_off = ls.seq_off; a.seq = ls.seq;
            a.perm = ls.perm; a.lib_start = ls.lib_start; a.sort_bad = ls.bad;
            // the pools dealt to the libraries by their sizes (on the device: nothing of the sort comes back to the host)
            HIP_TRY(c, c->ml_partials.reserve((size_t)mdx_n_pools((unsigned)grid) * 16));
            a.ml_plan = (const uint4 *)c->ml_partials.p;
            mdx_k_ml_plan(ls.lib_start, lo, gn, mdx_tile_records(a.dims, a.pfl_off != 0), grid, c->ml_partials.p, c->stream);
        }
        // (the pools' tile counters: zeroed by the reduction behind the previous launch, as a rule; else here, inside the timed region)
        // (a counter per 128-byte line; the reduction behind a launch zeroes those of the first 4 096 pools)
        const size_t n_ctr = (size_t)mdx_n_pools((unsigned)grid);
        if (!c->tile_ctr_clean || n_ctr > 4096) HIP_TRY(c, hipMemsetAsync(c->d_tile_ctr, 0, n_ctr * MDX_CTR_PAD * 4, c->stream));
        c->tile_ctr_clean = false;
        if (fuse) {
            // (a record written back unchanged keeps this NaN — all ones; inside the timed region)
            HIP_TRY(c, hipMemsetAsync(fuse->mr_raw, 0xFF, (size_t)b->n_reads * 8, c->stream));
            // the wavefronts' lists of records left to the rescale kernels: list_cap indices each, in front the counts
            const int64_t nwaves = (int64_t)grid * wpb_l;
            HIP_TRY(c, c->rs_in.reserve((size_t)nwaves * (size_t)(a.list_cap + 1) * 4));
            a.rs.gen_count = (uint32_t *)c->rs_in.p;
            a.rs.gen_list = a.rs.gen_count + nwaves;
            if (packed) { mdx_k_tabulate_packed_fused(a, grid, lds, c->stream); c->n_packed++; }
            else mdx_k_tabulate_fused(a, grid, lds, c->stream);
            c->fuse_list_cap = a.list_cap;
            if (fused_grid) *fused_grid = grid;
        } else if (packed) {
            if (pmask) mdx_k_tabulate_packed_masked(a, grid, c->pk.threads, lds, c->stream);
            else mdx_k_tabulate_packed(a, grid, c->pk.threads, lds, c->stream);
            c->n_packed++;
        } else {
            mdx_k_tabulate(a, c->mode, mask, grid, lds, c->stream);
        }
        if (c->timing) {
            HIP_TRY(c, hipEventRecord(e1, c->stream));
            c->events.emplace_back(e0, e1);
        }
        HIP_TRY(c, hipGetLastError());
        if (c->mode == MDX_MODE_LDS) {
            if (ml) mdx_k_reduce_partials(c->d_partials, a.raw, c->d_raw + c->dims.w_total - 1, a.dims.w_total, grid, c->stream, c->d_tile_ctr, gn, c->dims.w_lib, a.ml_plan);
            else
            mdx_k_reduce_partials(c->d_partials, a.raw, c->d_raw + c->dims.w_total - 1, a.dims.w_total, grid, c->stream, c->d_tile_ctr);
            c->tile_ctr_clean = a.dims.w_total >= 4096 && n_ctr <= 4096;
            HIP_TRY(c, hipGetLastError());
        }
    }
    return MDX_OK;
}

int mdx_tabulate_device(mdx_ctx *c, const mdx_batch *b) { return tabulate_impl(c, b, nullptr, nullptr); }

int mdx_tabulate_host(mdx_ctx *c, const mdx_batch *h) {
    int rc = check_batch(c, h);
    if (rc != MDX_OK) return rc;
    if (h->n_reads == 0) return MDX_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int64_t n = h->n_reads;
    const void *src[10] = {h->flag, h->lib, h->tid, h->pos, h->tlen, h->cigar_off, h->cigar, h->seq_off, h->seq, h->qual};
    const size_t bytes[10] = {(size_t)n * 2, (size_t)n * 2, (size_t)n * 4, (size_t)n * 4, (size_t)n * 4,
                              (size_t)(n + 1) * 4, (size_t)h->n_cigar * 4, (size_t)(n + 1) * 4,
                              seq_bytes(h), h->qual ? (size_t)h->n_bases : 0};
    // Two sets of staging buffers and a copy stream of the context's own: the columns of this batch are copied (set s)
    // while the kernel of the previous one still reads the other set; the kernel waits for its copies (an event), and
    // the copies into a set wait for the kernel that read it last (another one).  The call returns once the host
    // columns sit in the pinned bounce buffers — the caller may release them — without waiting for the device.
    constexpr size_t PIN_BYTES = (size_t)16 << 20;
    if (!c->copy_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
        if (!c->pin[i]) HIP_TRY(c, hipHostMalloc(&c->pin[i], PIN_BYTES, hipHostMallocDefault));
        if (!c->pin_done[i]) HIP_TRY(c, hipEventCreateWithFlags(&c->pin_done[i], hipEventDisableTiming));
        if (!c->st_copied[i]) HIP_TRY(c, hipEventCreateWithFlags(&c->st_copied[i], hipEventDisableTiming));
        if (!c->st_done[i]) HIP_TRY(c, hipEventCreateWithFlags(&c->st_done[i], hipEventDisableTiming));
    }
    const int s = c->st_turn;
    DevBuf *const st = c->st[s];
    // (a buffer that has to grow is freed first: the kernel that read it must be done — rare, the sets settle at the
    // size of the largest batch)
    // (--min-basequal and a 4-bit SEQ column: the mask is folded into the staged column — MDX_SEQ_4BITQ, the packed masked
    // kernel's input — on the copy stream behind the copies, under the kernel of the batch before)
    const bool stage_fold = c->cfg.minqual > 0 && h->qual && h->seq_format == MDX_SEQ_4BIT && h->n_bases > 0 && c->mode == MDX_MODE_LDS;
    bool grow = false;
    for (int i = 0; i < 10; i++) grow = grow || (src[i] && bytes[i] + 64 > st[i].cap);
    if (c->st_busy[s]) {
        if (grow) HIP_TRY(c, hipEventSynchronize(c->st_done[s]));
        else HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->st_done[s], 0));
    }
    int turn = 0;
    for (int i = 0; i < 10; i++) {
        if (!src[i] || bytes[i] == 0) continue;
        HIP_TRY(c, st[i].reserve(bytes[i] + 64));
        // pageable host memory -> pinned bounce buffer (CPU) -> device (DMA), 16 MiB at a time: a pageable
        // hipMemcpy moves ~3.5 GB/s on this platform, this pipeline what one core copies
        for (size_t off = 0; off < bytes[i]; off += PIN_BYTES) {
            const size_t len = bytes[i] - off < PIN_BYTES ? bytes[i] - off : PIN_BYTES;
            if (c->pin_busy[turn]) HIP_TRY(c, hipEventSynchronize(c->pin_done[turn]));
            std::memcpy(c->pin[turn], (const uint8_t *)src[i] + off, len);
            HIP_TRY(c, hipMemcpyAsync((uint8_t *)st[i].p + off, c->pin[turn], len, hipMemcpyHostToDevice, c->copy_stream));
            HIP_TRY(c, hipEventRecord(c->pin_done[turn], c->copy_stream));
            c->pin_busy[turn] = true;
            turn ^= 1;
        }
    }
    if (stage_fold) {
        mdx_k_fold_mask((const uint8_t *)st[8].p, (uint8_t *)st[8].p, (const uint8_t *)st[9].p, nullptr, h->n_bases, c->cfg.minqual, c->copy_stream);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipEventRecord(c->st_copied[s], c->copy_stream));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->st_copied[s], 0));
    mdx_batch dv = *h;
    dv.lowq = nullptr;
    dv.libsort = nullptr;
    if (stage_fold) dv.seq_format = MDX_SEQ_4BITQ;
    dv.flag = (const uint16_t *)st[0].p; dv.lib = (const uint16_t *)st[1].p;
    dv.tid = (const int32_t *)st[2].p; dv.pos = (const int32_t *)st[3].p;
    dv.tlen = (const int32_t *)st[4].p; dv.cigar_off = (const uint32_t *)st[5].p;
    dv.cigar = (const uint32_t *)st[6].p; dv.seq_off = (const uint32_t *)st[7].p;
    dv.seq = (const uint8_t *)st[8].p; dv.qual = h->qual ? (const uint8_t *)st[9].p : nullptr;
    rc = mdx_tabulate_device(c, &dv);
    // (recorded whatever the outcome: the set is in use until what was enqueued on the stream has run)
    HIP_TRY(c, hipEventRecord(c->st_done[s], c->stream));
    c->st_busy[s] = true;
    c->st_turn = s ^ 1;
    return rc;
}

int mdx_set_record_base(mdx_ctx *c, int64_t base) {
    if (!c || base < 0) return MDX_ERR_ARG;
    c->record_base = base;
    return MDX_OK;
}

int mdx_sync(mdx_ctx *c, int64_t *bad_read) {
    if (!c) return MDX_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned long long err = 0, nover = 0;
    HIP_TRY(c, hipMemcpy(&err, c->d_err, 8, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(&nover, c->d_n_lgd_over, 8, hipMemcpyDeviceToHost));
    if (err != ~0ull) {
        if (bad_read) *bad_read = (int64_t)(err >> 8);
        const int code = -(int)(err & 0xFF);
        char msg[160];
        std::snprintf(msg, sizeof msg, "record %lld (index within its batch + the record base): %s", (long long)(err >> 8), mdx_strerror(code));
        return fail(c, code, msg);
    }
    if ((int64_t)nover > c->cfg.lgd_over_cap) return fail(c, MDX_ERR_LGD_OVERFLOW, mdx_strerror(MDX_ERR_LGD_OVERFLOW));
    return MDX_OK;
}

int64_t mdx_table_words(const mdx_ctx *c) {
    if (!c) return 0;
    return mis_words(c) + comp_words(c) + lgd_words(c) + 2;
}

int mdx_finish_device(mdx_ctx *c, uint64_t *d_tables) {
    if (!c || !d_tables) return MDX_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    mdx_k_finalize(c->d_raw, c->d_lgd_dense, c->d_n_lgd_over, c->dims, (unsigned long long *)d_tables, c->stream);
    HIP_TRY(c, hipGetLastError());
    return MDX_OK;
}

int mdx_comm_unique_id(uint8_t *id) {
    if (!id) return MDX_ERR_ARG;
    const Rccl *r = rccl();
    if (!r) return MDX_ERR_COMM;
    RcclId u;
    if (r->GetUniqueId(&u) != 0) return MDX_ERR_COMM;
    std::memcpy(id, u.internal, MDX_COMM_ID_BYTES);
    return MDX_OK;
}

int mdx_comm_init(mdx_ctx *c, const uint8_t *id, int32_t nranks, int32_t rank) {
    if (!c || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, MDX_ERR_ARG, "comm_init: bad arguments");
    if (c->comm) return fail(c, MDX_ERR_STATE, "a communicator is already attached");
    const Rccl *r = rccl();
    if (!r) return fail(c, MDX_ERR_COMM, g_rccl.err);
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    RcclId u;
    std::memcpy(u.internal, id, MDX_COMM_ID_BYTES);
    void *comm = nullptr;
    RCCL_TRY(c, r, r->CommInitRank(&comm, nranks, u, rank));
    c->comm = comm; c->comm_owned = true; c->comm_size = nranks; c->comm_rank = rank;
    return MDX_OK;
}

int mdx_comm_adopt(mdx_ctx *c, void *rccl_comm, int32_t nranks, int32_t rank) {
    if (!c || !rccl_comm || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, MDX_ERR_ARG, "comm_adopt: bad arguments");
    if (c->comm) return fail(c, MDX_ERR_STATE, "a communicator is already attached");
    if (!rccl()) return fail(c, MDX_ERR_COMM, g_rccl.err);
    c->comm = rccl_comm; c->comm_owned = false; c->comm_size = nranks; c->comm_rank = rank;
    return MDX_OK;
}

int mdx_comm_size(const mdx_ctx *c) { return c && c->comm ? c->comm_size : 0; }

int mdx_comm_count(mdx_ctx *c) {
    if (!c) return MDX_ERR_ARG;
    if (!c->comm) return 0;
    const Rccl *r = rccl();
    if (!r) return fail(c, MDX_ERR_COMM, g_rccl.err);
    int n = 0;
    RCCL_TRY(c, r, r->CommCount(c->comm, &n));
    return n;
}

int mdx_finish_allreduce(mdx_ctx *c, uint64_t *d_tables) {
    if (!c || !d_tables) return MDX_ERR_ARG;
    if (!c->comm) return fail(c, MDX_ERR_STATE, "mdx_comm_init / mdx_comm_adopt first");
    const Rccl *r = rccl();
    int rc = mdx_finish_device(c, d_tables);
    if (rc != MDX_OK) return rc;
    RCCL_TRY(c, r, r->AllReduce(d_tables, d_tables, (size_t)mdx_table_words(c), kNcclUint64, kNcclSum, c->comm, c->stream));
    return MDX_OK;
}

int mdx_finish(mdx_ctx *c, uint64_t *mis, uint64_t *comp, uint64_t *lgd, int64_t *lgd_over,
               int64_t lgd_over_cap, int64_t *n_lgd_over, int64_t *n_kept) {
    if (!c) return MDX_ERR_ARG;
    int rc = mdx_sync(c, nullptr);
    const Rccl *r = c->comm ? rccl() : nullptr;
    const int64_t words = mdx_table_words(c);
    uint64_t *d = nullptr;      // [tables | error flag | this rank's list length]
    if (r) {
        // agree on an error flag first: a rank that failed must not leave the others in the collective
        const std::string own = c->err;
        HIP_TRY(c, hipMalloc((void **)&d, (size_t)(words + 2) * 8));
        const uint64_t flag = rc != MDX_OK ? 1 : 0;
        hipError_t e = hipMemcpyAsync(d + words, &flag, 8, hipMemcpyHostToDevice, c->stream);
        int ne = e == hipSuccess ? r->AllReduce(d + words, d + words, 1, kNcclUint64, kNcclSum, c->comm, c->stream) : 0;
        uint64_t any = 0;
        if (e == hipSuccess && ne == 0) e = hipMemcpyAsync(&any, d + words, 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && ne == 0) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess || ne != 0) {
            (void)hipFree(d);
            return fail(c, e != hipSuccess ? MDX_ERR_HIP : MDX_ERR_COMM, e != hipSuccess ? hipGetErrorString(e) : r->GetErrorString(ne));
        }
        if (rc != MDX_OK) { (void)hipFree(d); c->err = own; return rc; }
        if (any) { (void)hipFree(d); return fail(c, MDX_ERR_COMM, "another rank of the communicator reported an error"); }
    } else {
        if (rc != MDX_OK) return rc;
        HIP_TRY(c, hipMalloc((void **)&d, (size_t)words * 8));
    }
    rc = r ? mdx_finish_allreduce(c, d) : mdx_finish_device(c, d);
    std::vector<int64_t> gathered;   // the out-of-range lists of all ranks, rank order
    if (rc == MDX_OK && r) {
        // list lengths of all ranks, then the lists padded to the longest
        unsigned long long own_n = 0;
        hipError_t e = hipMemcpy(&own_n, c->d_n_lgd_over, 8, hipMemcpyDeviceToHost);
        if ((int64_t)own_n > c->cfg.lgd_over_cap) own_n = (unsigned long long)c->cfg.lgd_over_cap;
        uint64_t *d_cnt = nullptr;
        std::vector<uint64_t> cnt((size_t)c->comm_size, 0);
        if (e == hipSuccess) e = hipMalloc((void **)&d_cnt, (size_t)(c->comm_size + 1) * 8);
        if (e == hipSuccess) e = hipMemcpyAsync(d_cnt + c->comm_size, &own_n, 8, hipMemcpyHostToDevice, c->stream);
        int ne = 0;
        if (e == hipSuccess) ne = r->AllGather(d_cnt + c->comm_size, d_cnt, 1, kNcclUint64, c->comm, c->stream);
        if (e == hipSuccess && ne == 0) e = hipMemcpyAsync(cnt.data(), d_cnt, (size_t)c->comm_size * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess && ne == 0) e = hipStreamSynchronize(c->stream);
        uint64_t longest = 0;
        for (uint64_t v : cnt) longest = v > longest ? v : longest;
        if (e == hipSuccess && ne == 0 && longest > 0) {
            int64_t *d_all = nullptr, *d_own = nullptr;
            e = hipMalloc((void **)&d_all, (size_t)c->comm_size * longest * 32);
            if (e == hipSuccess) e = hipMalloc((void **)&d_own, (size_t)longest * 32);
            if (e == hipSuccess) e = hipMemsetAsync(d_own, 0, (size_t)longest * 32, c->stream);
            if (e == hipSuccess && own_n) e = hipMemcpyAsync(d_own, c->d_lgd_over, (size_t)own_n * 32, hipMemcpyDeviceToDevice, c->stream);
            if (e == hipSuccess) ne = r->AllGather(d_own, d_all, (size_t)longest * 4, kNcclInt64, c->comm, c->stream);
            std::vector<int64_t> all((size_t)c->comm_size * longest * 4);
            if (e == hipSuccess && ne == 0) e = hipMemcpyAsync(all.data(), d_all, all.size() * 8, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess && ne == 0) e = hipStreamSynchronize(c->stream);
            if (e == hipSuccess && ne == 0)
                for (int k = 0; k < c->comm_size; k++)
                    gathered.insert(gathered.end(), all.begin() + (size_t)k * longest * 4, all.begin() + ((size_t)k * longest + cnt[k]) * 4);
            if (d_all) (void)hipFree(d_all);
            if (d_own) (void)hipFree(d_own);
        }
        if (d_cnt) (void)hipFree(d_cnt);
        if (e != hipSuccess) rc = fail(c, MDX_ERR_HIP, hipGetErrorString(e));
        else if (ne != 0) rc = fail(c, MDX_ERR_COMM, r->GetErrorString(ne));
    }
    if (rc == MDX_OK) {
        hipError_t e = hipStreamSynchronize(c->stream);
        const int64_t nm = mis_words(c), nc = comp_words(c), nl = lgd_words(c);
        if (e == hipSuccess && mis) e = hipMemcpy(mis, d, (size_t)nm * 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess && comp) e = hipMemcpy(comp, d + nm, (size_t)nc * 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess && lgd) e = hipMemcpy(lgd, d + nm + nc, (size_t)nl * 8, hipMemcpyDeviceToHost);
        uint64_t tail[2] = {0, 0};
        if (e == hipSuccess) e = hipMemcpy(tail, d + nm + nc + nl, 16, hipMemcpyDeviceToHost);
        if (e == hipSuccess) {
            if (n_kept) *n_kept = (int64_t)tail[0];
            int64_t nov = (int64_t)tail[1];
            if (n_lgd_over) *n_lgd_over = nov;
            if (lgd_over && nov > 0) {
                if (r) {
                    // the gathered list of all ranks: every rank stays within lgd_over_cap (mdx_sync), their sum need not
                    // stay within the caller's buffer — an error, not a silently shortened list
                    nov = (int64_t)gathered.size() / 4;
                    if (nov > lgd_over_cap) {
                        // (the tables and n_kept are out already; *n_lgd_over says how many entries a retry needs)
                        if (n_lgd_over) *n_lgd_over = nov;
                        (void)hipFree(d);
                        return fail(c, MDX_ERR_LGD_OVERFLOW, "the out-of-range fragment lengths of all ranks exceed the "
                                    "caller's lgd_over buffer (give it comm_size x lgd_over_cap entries)");
                    }
                    if (n_lgd_over) *n_lgd_over = nov;
                    std::memcpy(lgd_over, gathered.data(), (size_t)nov * 32);
                } else {
                    if (nov > lgd_over_cap) nov = lgd_over_cap;
                    e = hipMemcpy(lgd_over, c->d_lgd_over, (size_t)nov * 32, hipMemcpyDeviceToHost);
                }
            }
        }
        if (e != hipSuccess) rc = fail(c, MDX_ERR_HIP, hipGetErrorString(e));
    }
    (void)hipFree(d);
    return rc;
}

int mdx_reset(mdx_ctx *c) {
    if (!c) return MDX_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    int rc = zero_accumulators(c);
    if (rc != MDX_OK) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->err.clear();
    return MDX_OK;
}

int mdx_timing_enable(mdx_ctx *c, int enable) {
    if (!c) return MDX_ERR_ARG;
    c->timing = enable != 0;
    return MDX_OK;
}

int mdx_timing_read(mdx_ctx *c, int64_t *n_launches, double *total_ms) {
    if (!c) return MDX_ERR_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double tot = 0;
    int64_t n = 0;
    for (auto &ev : c->events) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) { tot += ms; n++; }
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    c->events.clear();
    if (n_launches) *n_launches = n;
    if (total_ms) *total_ms = tot;
    return MDX_OK;
}

int mdx_table_mode(const mdx_ctx *c) { return c ? c->mode : -1; }

// The device decode (mdx_gbam_*) leaves its arena and its CRC tables with the context when a file is closed and takes them
// back when the next one is opened: freeing and allocating a few hundred megabytes costs milliseconds per file.
void mdx_ctx_scratch_give(mdx_ctx *c, void *arena, size_t cap, void *tables, void *stream, void *event) {
    if (!c) {
        if (arena) (void)hipFree(arena);
        if (tables) (void)hipFree(tables);
        if (stream) (void)hipStreamDestroy((hipStream_t)stream);
        if (event) (void)hipEventDestroy((hipEvent_t)event);
        return;
    }
    if (arena) {
        if (c->gbam_arena && c->gbam_arena_cap >= cap) (void)hipFree(arena);
        else { if (c->gbam_arena) (void)hipFree(c->gbam_arena); c->gbam_arena = arena; c->gbam_arena_cap = cap; }
    }
    if (tables) { if (c->gbam_tables) (void)hipFree(tables); else c->gbam_tables = tables; }
    if (stream) { if (c->gbam_stream) (void)hipStreamDestroy((hipStream_t)stream); else c->gbam_stream = stream; }
    if (event) { if (c->gbam_event) (void)hipEventDestroy((hipEvent_t)event); else c->gbam_event = event; }
}
void mdx_ctx_scratch_take(mdx_ctx *c, void **arena, size_t *cap, void **tables, void **stream, void **event) {
    *arena = nullptr; *cap = 0; *tables = nullptr; *stream = nullptr; *event = nullptr;
    if (!c) return;
    *arena = c->gbam_arena; *cap = c->gbam_arena_cap; *tables = c->gbam_tables; *stream = c->gbam_stream; *event = c->gbam_event;
    c->gbam_arena = nullptr; c->gbam_arena_cap = 0; c->gbam_tables = nullptr; c->gbam_stream = nullptr; c->gbam_event = nullptr;
}

int mdx_ctx_minqual(const mdx_ctx *c) { return c ? c->cfg.minqual : -1; }

int mdx_ctx_stream(mdx_ctx *c, void **stream, int *device) {
    if (!c) return MDX_ERR_ARG;
    if (stream) *stream = (void *)c->stream;
    if (device) *device = c->cfg.device;
    return MDX_OK;
}

int mdx_rescale_set_model(mdx_ctx *c, const uint8_t *lut, const double *term, int32_t len5p, int32_t len3p) {
    if (!c || !lut || !term || len5p < 0 || len3p < 0 || len5p + len3p > 100000) return fail(c, MDX_ERR_ARG, "rescale model");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_lut) { (void)hipFree(c->d_lut); c->d_lut = nullptr; }
    if (c->d_term) { (void)hipFree(c->d_term); c->d_term = nullptr; }
    if (c->d_subs) { (void)hipFree(c->d_subs); c->d_subs = nullptr; }
    const size_t npos = (size_t)1 + len5p + len3p;
    HIP_TRY(c, hipMalloc((void **)&c->d_lut, 2 * npos * 94));
    HIP_TRY(c, hipMalloc((void **)&c->d_term, 2 * npos * 8));
    HIP_TRY(c, hipMemcpy(c->d_lut, lut, 2 * npos * 94, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_term, term, 2 * npos * 8, hipMemcpyHostToDevice));
    c->len5p = len5p; c->len3p = len3p;
    c->h_lut.assign(lut, lut + 2 * npos * 94);
    // the fast path of the kernel walks only the end windows of a record: that needs the columns outside them
    // (position key 0) to stay as they are — true of every model get_corr_prob builds (probability 0 there)
    c->key0_plain = true;
    for (int sub = 0; sub < 2; sub++) {
        if (term[(size_t)sub * npos] != 0.0) c->key0_plain = false;
        for (int q = 0; q < 94; q++)
            if (lut[((size_t)sub * npos) * 94 + q] != (uint8_t)q) c->key0_plain = false;
    }
    const size_t sub_bytes = (size_t)(756 + 2 * npos * 94) * 8;
    HIP_TRY(c, hipMalloc((void **)&c->d_subs, sub_bytes));
    HIP_TRY(c, hipMemset(c->d_subs, 0, sub_bytes));
    return MDX_OK;
}

int64_t mdx_rescale_summary_words(const mdx_ctx *c) {
    return (c && c->d_subs) ? 756 + 2 * (int64_t)(1 + c->len5p + c->len3p) * 94 : 0;
}

int mdx_rescale_summary(mdx_ctx *c, uint64_t *words) {
    if (!c || !words) return MDX_ERR_ARG;
    if (!c->d_subs) return fail(c, MDX_ERR_STATE, "rescale_set_model first");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int64_t nw = mdx_rescale_summary_words(c);
    HIP_TRY(c, hipMemcpy(words, c->d_subs, (size_t)nw * 8, hipMemcpyDeviceToHost));
    // The kernel counts one word per column: occurrences per (substitution, key, old quality) for C>T / G>A, one
    // histogram for T>C / A>G (in their "before" words).  The before / after histograms of _record_subs
    // (rescale.py:108-141) follow: a rescaled column of old quality q has the new quality lut[sub][key][q].
    const int64_t npos = 1 + (int64_t)c->len5p + c->len3p;
    for (int t4 = 0; t4 < 4; t4++) {
        uint64_t *before = words + 4 + (t4 * 2 + 0) * 94, *after = words + 4 + (t4 * 2 + 1) * 94;
        if (t4 & 1) {
            for (int q = 0; q < 94; q++) after[q] = before[q];
            continue;
        }
        for (int q = 0; q < 94; q++) before[q] = after[q] = 0;
        const int sub = t4 >> 1;
        for (int64_t key = 0; key < npos; key++)
            for (int q = 0; q < 94; q++) {
                const uint64_t n = words[756 + (sub * npos + key) * 94 + q];
                if (!n) continue;
                before[q] += n;
                const uint8_t nq = c->h_lut[(size_t)((sub * npos + key) * 94 + q)];
                after[nq < 94 ? nq : 93] += n;
            }
    }
    return MDX_OK;
}

// where the rescaled bytes go: a second quality column (qual_out), or — patch mode — a list of the bytes that change
struct RsOut { uint8_t *qual_out; uint64_t *patch; int64_t patch_cap; int32_t parts; uint64_t *n_patch; };
static bool parts_ok(int32_t n) { return n >= 1 && n <= 65535 && (n & (n - 1)) == 0; }

static int rescale_device_impl(mdx_ctx *c, const mdx_batch *b_in, const int32_t *d_mtid, const int32_t *d_mpos, const RsOut &o,
                               double *d_mr_raw, uint8_t *d_status, bool zero_count) {
    uint8_t *const d_qual_out = o.qual_out;
    int rc = check_batch(c, b_in);
    if (rc != MDX_OK) return rc;
    // (the rescale kernels read ASCII)
    mdx_batch b_ascii;
    rc = ascii_view(c, b_in, &b_ascii);
    if (rc != MDX_OK) return rc;
    const mdx_batch *b = &b_ascii;
    if (!c->d_ref || !c->d_lut) return fail(c, MDX_ERR_STATE, "set_reference and rescale_set_model first");
    if (!b->qual || !d_mtid || !d_mpos || !d_mr_raw || !d_status) return fail(c, MDX_ERR_ARG, "null column");
    if (o.patch ? (!o.n_patch || o.patch_cap < 0 || !parts_ok(o.parts) || d_qual_out) : !d_qual_out) return fail(c, MDX_ERR_ARG, "null column");
    // (the kernels read the old qualities of a record after they have stored its new ones; offsets up to a few hundred
    //  bytes past the column are formed in 32 bits)
    if (d_qual_out == b->qual) return fail(c, MDX_ERR_ARG, "qual_out must not be the batch's own quality column");
    if (b->n_bases > 0xFFFF0000LL) return fail(c, MDX_ERR_ARG, "batch exceeds 32-bit offsets; split it");
    if (b->n_reads == 0) return MDX_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    MdxRescaleArgs a{};
    a.n_reads = b->n_reads; a.n_bases = b->n_bases; a.flag = b->flag; a.tid = b->tid; a.pos = b->pos; a.mtid = d_mtid; a.mpos = d_mpos;
    a.cigar_off = b->cigar_off; a.cigar = b->cigar; a.seq_off = b->seq_off; a.seq = b->seq; a.qual = b->qual;
    a.ref = c->d_ref + 256; a.contig_off = c->d_contig_off; a.n_contig = c->n_contig;
    a.lut = c->d_lut; a.term = c->d_term; a.len5p = c->len5p; a.len3p = c->len3p; a.key0_plain = c->key0_plain ? 1 : 0;
    a.qual_out = d_qual_out; a.mr_raw = d_mr_raw; a.status = d_status; a.err = c->d_err; a.subs = c->d_subs;
    a.patch = (unsigned long long *)o.patch; a.n_patch = (unsigned long long *)o.n_patch; a.patch_cap = o.patch_cap; a.patch_parts = o.parts;
    if (o.patch && zero_count) HIP_TRY(c, hipMemsetAsync(o.n_patch, 0, (size_t)o.parts * 8, c->stream));
    HIP_TRY(c, c->rs_part.reserve(mdx_k_rescale_part_bytes(c->len5p, c->len3p, c->n_cu)));
    a.subs_part = (uint32_t *)c->rs_part.p;
    int64_t list_waves = 0, list_cap = 0;
    mdx_k_rescale_lists(b->n_reads, c->n_cu, &list_waves, &list_cap);
    HIP_TRY(c, c->rs_lists.reserve((size_t)list_waves * (size_t)(list_cap + 1) * 4));
    a.gen_count = (uint32_t *)c->rs_lists.p;
    a.gen_list = a.gen_count + list_waves;
    a.gen_cap = list_cap;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess)
        (void)hipEventRecord(e0, c->stream);
    // qual_out becomes a copy of qual inside the launch (mdx_k_rescale): the kernels then only store the bytes they
    // rescale (a few per read) — the timed region holds both
    mdx_k_rescale(a, c->n_cu, c->stream);
    if (e0 && e1) {
        (void)hipEventRecord(e1, c->stream);
        c->rs_events.emplace_back(e0, e1);
    }
    HIP_TRY(c, hipGetLastError());
    return MDX_OK;
}

int mdx_rescale_device(mdx_ctx *c, const mdx_batch *b_in, const int32_t *d_mtid, const int32_t *d_mpos, uint8_t *d_qual_out,
                       double *d_mr_raw, uint8_t *d_status) {
    if (!d_qual_out) return fail(c, MDX_ERR_ARG, "null column");
    return rescale_device_impl(c, b_in, d_mtid, d_mpos, RsOut{d_qual_out, nullptr, 0, 0, nullptr}, d_mr_raw, d_status, true);
}

int mdx_rescale_patches_device(mdx_ctx *c, const mdx_batch *b_in, const int32_t *d_mtid, const int32_t *d_mpos, uint64_t *d_patch,
                               int64_t patch_cap, int32_t n_parts, uint64_t *d_n_patch, double *d_mr_raw, uint8_t *d_status) {
    if (!d_patch || !d_n_patch || patch_cap < 0 || !parts_ok(n_parts)) return fail(c, MDX_ERR_ARG, "patch list: null, or n_parts not a power of two");
    return rescale_device_impl(c, b_in, d_mtid, d_mpos, RsOut{nullptr, d_patch, patch_cap, n_parts, d_n_patch}, d_mr_raw, d_status, true);
}

int mdx_rescale_expand_device(mdx_ctx *c, const mdx_batch *b, const uint64_t *d_patch, int64_t patch_cap, int32_t n_parts,
                              const uint64_t *d_n_patch, uint8_t *d_qual_out) {
    int rc = check_batch(c, b);
    if (rc != MDX_OK) return rc;
    if (!b->qual || !d_patch || !d_n_patch || !d_qual_out || patch_cap < 0 || !parts_ok(n_parts)) return fail(c, MDX_ERR_ARG, "null column");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    mdx_k_rescale_expand(b->qual, d_qual_out, b->n_bases, (const unsigned long long *)d_patch, (const unsigned long long *)d_n_patch, patch_cap, n_parts, c->stream);
    HIP_TRY(c, hipGetLastError());
    return MDX_OK;
}

// The fused launch applies when the tabulation is the plain fast kernel in one launch (tables in the LDS, all libraries
// at once, no --min-basequal, 32-bit reference offsets) and the model is one the end-window walk can take (key 0 the
// identity) whose tables fit the image next to a second TC table.  MDX_NO_FUSE=1 in the environment: never (A/B).
static bool fuse_applies(const mdx_ctx *c, const mdx_batch *b, bool packed_form = false) {
    const char *env = getenv("MDX_NO_FUSE");
    const bool off = env && *env && *env != '0';
    if (off || c->mode != MDX_MODE_LDS || c->lib_group != c->cfg.nlib || !c->dims.fast_ok()) return false;
    { const char *e = getenv("MDX_FORCE_REF64"); if (e && *e && *e != '0') return false; }
    if (c->cfg.minqual > 0 || !(c->ref_len + 1024 < (int64_t)0xFFFFFFFFLL)) return false;
    const int npos = 1 + c->len5p + c->len3p;
    // (the MR terms of a record are noted as bits sub * npos + key of one 64-bit word)
    if (!c->key0_plain || npos > 32 || !c->d_subs) return false;
    if (packed_form) {
        // (the packed fused kernel: one library, its own image — no second TC table)
        static const bool no_packed = [] { const char *e = getenv("MDX_NO_PACKED"); return e && *e && *e != '0'; }();
        static const bool no_pkf = [] { const char *e = getenv("MDX_NO_PACKED_FUSE"); return e && *e && *e != '0'; }();
        if (no_packed || no_pkf || c->cfg.nlib != 1 || mdx_k_pkf_lds_bytes(c->dims, npos) > kLdsLimit) return false;
        return b->n_reads > 0 && b->n_bases <= 0xFFFF0000LL;
    }
    if (mdx_k_fuse_lds_bytes(c->dims, npos, 64) > kLdsLimit) return false;
    // (the second TC table's byte offset travels in 10 bits of a staging entry, in units of 256 bytes)
    if ((size_t)mdx_k_fuse_tcb_off(c->dims, 160) * 4 + (size_t)c->dims.nlib * c->dims.w_tc * 4 > ((size_t)1 << 18)) return false;
    return b->n_reads > 0 && b->n_bases <= 0xFFFF0000LL;
}

static int tabulate_rescale_impl(mdx_ctx *c, const mdx_batch *b_in, const int32_t *d_mtid, const int32_t *d_mpos,
                                 const RsOut &o, double *d_mr_raw, uint8_t *d_status) {
    uint8_t *const d_qual_out = o.qual_out;
    // one pass over one resident batch: the tables and, from the same columns in HBM, the rescaled qualities
    int rc = check_batch(c, b_in);
    if (rc != MDX_OK) return rc;
    // (the fused kernel copies the quality column in 16-byte units: both columns at the same 16-byte phase — true of any two
    // device allocations; patch mode: no second column)
    const bool args_ok = c->d_ref && c->d_lut && b_in->qual && d_mtid && d_mpos && d_mr_raw && d_status &&
                         (o.patch ? (o.n_patch != nullptr && !d_qual_out && parts_ok(o.parts))
                                  : (d_qual_out && d_qual_out != b_in->qual && (((uintptr_t)d_qual_out ^ (uintptr_t)b_in->qual) & 15) == 0));
    if (o.patch && o.n_patch && parts_ok(o.parts)) HIP_TRY(c, hipMemsetAsync(o.n_patch, 0, (size_t)o.parts * 8, c->stream));
    // A 4-bit SEQ column goes through the packed fused kernel as it is (one library); the records that kernel lists for the
    // rescale kernels, which read ASCII, get their stretches of an ASCII scratch column written behind it (mdx_k_unpack_listed)
    const bool pkf = args_ok && b_in->seq_format == MDX_SEQ_4BIT && b_in->n_bases > 0 && fuse_applies(c, b_in, true);
    mdx_batch b_ascii;
    if (pkf) {
        b_ascii = *b_in;
        HIP_TRY(c, c->unpacked.reserve((size_t)b_in->n_bases + 64));
        b_ascii.seq = (const uint8_t *)c->unpacked.p;
        b_ascii.seq_format = MDX_SEQ_ASCII;
    } else {
        rc = ascii_view(c, b_in, &b_ascii);
        if (rc != MDX_OK) return rc;
    }
    const mdx_batch *b = &b_ascii;
    if (!pkf && (!args_ok || !fuse_applies(c, b))) {
        rc = mdx_tabulate_device(c, b_in);
        if (rc != MDX_OK) return rc;
        return rescale_device_impl(c, b, d_mtid, d_mpos, o, d_mr_raw, d_status, false);
    }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    // The tabulation kernel rescales the records of its own tile loop as it counts them ([S] M [S] of at most 2 L
    // aligned bases: qualities, MR, status, their part of the summary) and copies the quality column on the way; the
    // records it lists — gapped ones, longer ones, records the tabulation does not count — go through rescale_kernel and
    // rescale_walk_kernel as in mdx_rescale_device.
    const int npos = 1 + c->len5p + c->len3p, n_cnt = 752 + 2 * npos * 94;
    MdxFuse f{};
    f.mtid = d_mtid; f.mpos = d_mpos; f.qual_out = d_qual_out; f.mr_raw = d_mr_raw; f.status = d_status;
    f.patch = (unsigned long long *)o.patch; f.n_patch = (unsigned long long *)o.n_patch; f.patch_cap = o.patch_cap; f.patch_parts = o.parts;
    f.lut = c->d_lut; f.term = c->d_term; f.len5p = c->len5p; f.len3p = c->len3p;
    HIP_TRY(c, c->rs_part.reserve(mdx_k_rescale_part_bytes(c->len5p, c->len3p, c->n_cu) + (size_t)c->n_cu * n_cnt * 4));
    f.subs_part = (uint32_t *)c->rs_part.p;
    int fgrid = 0;
    rc = tabulate_impl(c, pkf ? b_in : b, &f, &fgrid);
    if (rc != MDX_OK) return rc;
    MdxRescaleArgs a{};
    a.n_reads = b->n_reads; a.n_bases = b->n_bases; a.flag = b->flag; a.tid = b->tid; a.pos = b->pos; a.mtid = d_mtid; a.mpos = d_mpos;
    a.cigar_off = b->cigar_off; a.cigar = b->cigar; a.seq_off = b->seq_off; a.seq = b->seq; a.qual = b->qual;
    a.ref = c->d_ref + 256; a.contig_off = c->d_contig_off; a.n_contig = c->n_contig;
    a.lut = c->d_lut; a.term = c->d_term; a.len5p = c->len5p; a.len3p = c->len3p; a.key0_plain = 1;
    a.qual_out = d_qual_out; a.mr_raw = d_mr_raw; a.status = d_status; a.err = c->d_err; a.subs = c->d_subs;
    a.patch = (unsigned long long *)o.patch; a.n_patch = (unsigned long long *)o.n_patch; a.patch_cap = o.patch_cap; a.patch_parts = o.parts;
    a.subs_part = (uint32_t *)c->rs_part.p;
    const int64_t n_in = (int64_t)fgrid * (mdx_k_fuse_block_threads() / 64);
    a.in_count = (const uint32_t *)c->rs_in.p;
    a.in_list = a.in_count + n_in;
    a.in_cap = c->fuse_list_cap;
    a.n_in = (int)n_in;
    // (a wavefront of rescale_kernel takes one list — there are at least as many wavefronts as lists — and leaves at
    // most that many records to the walk kernel)
    int64_t cons_waves = 0, cons_cap = 0;
    mdx_k_rescale_lists((int64_t)1 << 40, c->n_cu, &cons_waves, &cons_cap);     // (the wavefronts of a full grid)
    HIP_TRY(c, c->rs_lists.reserve((size_t)cons_waves * (size_t)(a.in_cap + 1) * 4));
    a.gen_count = (uint32_t *)c->rs_lists.p;
    a.gen_list = a.gen_count + cons_waves;
    a.gen_cap = a.in_cap;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess)
        (void)hipEventRecord(e0, c->stream);
    if (pkf) mdx_k_unpack_listed(a.in_count, a.in_list, a.in_cap, a.n_in, b->seq_off, b_in->seq, (uint8_t *)c->unpacked.p, b->n_bases, c->stream);
    mdx_k_rescale_lists_pass(a, fgrid, c->n_cu, c->stream);
    c->n_fused++;
    if (e0 && e1) {
        (void)hipEventRecord(e1, c->stream);
        c->rs_events.emplace_back(e0, e1);
    }
    HIP_TRY(c, hipGetLastError());
    return MDX_OK;
}

int mdx_tabulate_rescale_device(mdx_ctx *c, const mdx_batch *b_in, const int32_t *d_mtid, const int32_t *d_mpos,
                                uint8_t *d_qual_out, double *d_mr_raw, uint8_t *d_status) {
    return tabulate_rescale_impl(c, b_in, d_mtid, d_mpos, RsOut{d_qual_out, nullptr, 0, 0, nullptr}, d_mr_raw, d_status);
}

int mdx_tabulate_rescale_patches_device(mdx_ctx *c, const mdx_batch *b_in, const int32_t *d_mtid, const int32_t *d_mpos, uint64_t *d_patch,
                                        int64_t patch_cap, int32_t n_parts, uint64_t *d_n_patch, double *d_mr_raw, uint8_t *d_status) {
    if (!d_patch || !d_n_patch || patch_cap < 0 || !parts_ok(n_parts)) return fail(c, MDX_ERR_ARG, "patch list: null, or n_parts not a power of two");
    return tabulate_rescale_impl(c, b_in, d_mtid, d_mpos, RsOut{nullptr, d_patch, patch_cap, n_parts, d_n_patch}, d_mr_raw, d_status);
}

int mdx_rescale_host(mdx_ctx *c, const mdx_batch *h, const int32_t *mtid, const int32_t *mpos, uint8_t *qual_out,
                     double *mr_raw, uint8_t *status) {
    int rc = check_batch(c, h);
    if (rc != MDX_OK) return rc;
    if (!c->d_ref || !c->d_lut) return fail(c, MDX_ERR_STATE, "set_reference and rescale_set_model first");
    if (!h->qual || !mtid || !mpos || !qual_out || !mr_raw || !status) return fail(c, MDX_ERR_ARG, "null column");
    if (h->n_reads == 0) return MDX_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    mdx_batch dv;
    rc = mdx_batch_upload(c, h, &dv);
    if (rc != MDX_OK) return rc;
    const int64_t n = h->n_reads;
    int32_t *d_mtid = nullptr, *d_mpos = nullptr;
    uint8_t *d_qout = nullptr, *d_status = nullptr;
    double *d_mr = nullptr;
    hipError_t e = hipMalloc((void **)&d_mtid, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_mpos, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_qout, (size_t)h->n_bases + 64);
    if (e == hipSuccess) e = hipMalloc((void **)&d_status, (size_t)n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_mr, (size_t)n * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_mtid, mtid, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_mpos, mpos, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        rc = mdx_rescale_device(c, &dv, d_mtid, d_mpos, d_qout, d_mr, d_status);
        if (rc != MDX_OK) e = hipErrorUnknown;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(qual_out, d_qout, (size_t)h->n_bases, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(mr_raw, d_mr, (size_t)n * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(status, d_status, (size_t)n, hipMemcpyDeviceToHost);
    void *tmp[] = {d_mtid, d_mpos, d_qout, d_status, d_mr};
    for (void *p : tmp) if (p) (void)hipFree(p);
    (void)mdx_batch_free(c, &dv);
    if (rc != MDX_OK) return rc;
    if (e != hipSuccess) return fail(c, MDX_ERR_HIP, hipGetErrorString(e));
    return mdx_sync(c, nullptr);
}

int64_t mdx_fused_launches(const mdx_ctx *c) { return c ? c->n_fused : -1; }
int64_t mdx_packed_launches(const mdx_ctx *c) { return c ? c->n_packed : -1; }
int64_t mdx_libsorts(const mdx_ctx *c) { return c ? c->n_libsorts : -1; }

int mdx_rescale_timing_read(mdx_ctx *c, int64_t *n_launches, double *total_ms) {
    if (!c) return MDX_ERR_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double tot = 0;
    int64_t n = 0;
    for (auto &ev : c->rs_events) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) { tot += ms; n++; }
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    c->rs_events.clear();
    if (n_launches) *n_launches = n;
    if (total_ms) *total_ms = tot;
    return MDX_OK;
}

int mdx_genome_composition(mdx_ctx *c, uint64_t *counts) {
    if (!c || !counts) return MDX_ERR_ARG;
    if (!c->d_ref) return fail(c, MDX_ERR_STATE, "mdx_set_reference has not been called");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    unsigned long long *d_out = nullptr;
    const size_t bytes = (size_t)c->n_contig * 4 * 8;
    HIP_TRY(c, hipMalloc((void **)&d_out, bytes));
    hipError_t e = hipMemsetAsync(d_out, 0, bytes, c->stream);
    if (e == hipSuccess) {
        mdx_k_genome_comp(c->d_ref + 256, c->d_contig_off, c->n_contig, d_out, c->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(counts, d_out, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(c, MDX_ERR_HIP, hipGetErrorString(e));
    return MDX_OK;
}

}  // extern "C"
