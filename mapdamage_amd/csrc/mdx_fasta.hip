// FASTA / .fai on the way to the resident reference (include/mdx.h mdx_fasta_index, mdx_set_reference_fasta): the counterpart
// of pysam.FastaFile(options.ref) at mapdamage/main.py:115 — htslib's faidx: the index is read, or built when the file has
// none — and of the ref.fetch(...) calls behind it (main.py:180, align.py:32-33).  The FILE's bytes go to HBM as they lie on
// disk, a piece at a time; a kernel takes the line ends out by the arithmetic of the index (base i of a sequence lies at
// offset + i / linebases * linewidth + i % linebases) and writes the case-folded, classified bases straight into the resident
// reference.  No pass over the bases on the host.
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "mdx_fasta.hip is written for gfx950 (MI355X) only"
#endif
#include "../../include/mdx.h"
#include "mdx_internal.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct FaiEntry { std::string name; int64_t len, off, lb, lw; };

struct Mapped {
    const uint8_t *p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool open(const char *path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) return true;
        void *m = mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) return false;
        p = (const uint8_t *)m;
        return true;
    }
    ~Mapped() {
        if (p) munmap((void *)p, n);
        if (fd >= 0) ::close(fd);
    }
};

// htslib's fai_build_core for FASTA: name = the header line up to the first white space; offset = first byte behind the header
// line; linebases / linewidth from the sequence's first line; every line but the last of a sequence must be that long
bool build_index(const Mapped &f, std::vector<FaiEntry> &out, std::string &err) {
    const uint8_t *p = f.p;
    const size_t n = f.n;
    size_t i = 0;
    // (what stands in front of the first header is skipped if it is white space only, as htslib does)
    while (i < n && (p[i] == '\n' || p[i] == '\r' || p[i] == ' ' || p[i] == '\t')) i++;
    if (i < n && p[i] != '>') { err = "not a FASTA file: the first line does not begin with '>'"; return false; }
    while (i < n) {
        // header line
        size_t e = i + 1;
        while (e < n && p[e] != '\n') e++;
        size_t ne = i + 1;
        while (ne < e && p[ne] != ' ' && p[ne] != '\t' && p[ne] != '\r' && p[ne] != '\v' && p[ne] != '\f') ne++;
        FaiEntry fe;
        fe.name.assign((const char *)p + i + 1, ne - (i + 1));
        fe.len = 0; fe.lb = 0; fe.lw = 0;
        i = e < n ? e + 1 : n;
        fe.off = (int64_t)i;
        bool short_seen = false;
        while (i < n && p[i] != '>') {
            const uint8_t *nl = (const uint8_t *)memchr(p + i, '\n', n - i);
            const size_t le = nl ? (size_t)(nl - p) : n;            // line is [i, le)
            size_t be = le;
            while (be > i && (p[be - 1] == '\r')) be--;
            const int64_t bases = (int64_t)(be - i), width = (int64_t)((nl ? le + 1 : le) - i);
            if (bases > 0 || width > 0) {
                if (short_seen && bases > 0) { err = "different line length in sequence '" + fe.name + "'"; return false; }
                if (fe.lb == 0 && fe.len == 0 && !short_seen) { fe.lb = bases; fe.lw = width; }
                if (bases != fe.lb || width != fe.lw) {
                    if (bases > fe.lb) { err = "different line length in sequence '" + fe.name + "'"; return false; }
                    short_seen = true;
                }
                fe.len += bases;
            }
            i = nl ? le + 1 : n;
        }
        out.push_back(std::move(fe));
    }
    if (out.empty()) { err = "no sequence in the FASTA file"; return false; }
    return true;
}

bool read_index(const std::string &path, std::vector<FaiEntry> &out, std::string &err) {
    FILE *fh = std::fopen(path.c_str(), "r");
    if (!fh) { err = "cannot open '" + path + "'"; return false; }
    char *line = nullptr;
    size_t cap = 0;
    ssize_t got;
    int lineno = 0;
    bool ok = true;
    while ((got = getline(&line, &cap, fh)) >= 0) {
        lineno++;
        std::string s(line, (size_t)got);
        while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
        if (s.empty()) continue;
        std::vector<std::string> col;
        size_t a = 0;
        for (;;) {
            const size_t b = s.find('\t', a);
            col.push_back(s.substr(a, b == std::string::npos ? std::string::npos : b - a));
            if (b == std::string::npos) break;
            a = b + 1;
        }
        if (col.size() < 5) { err = "line " + std::to_string(lineno) + " of '" + path + "' holds " + std::to_string(col.size()) + " fields, 5 expected"; ok = false; break; }
        FaiEntry fe;
        fe.name = col[0];
        char *end = nullptr;
        fe.len = std::strtoll(col[1].c_str(), &end, 10); if (*end) ok = false;
        fe.off = std::strtoll(col[2].c_str(), &end, 10); if (*end) ok = false;
        fe.lb = std::strtoll(col[3].c_str(), &end, 10); if (*end) ok = false;
        fe.lw = std::strtoll(col[4].c_str(), &end, 10); if (*end) ok = false;
        if (!ok || fe.len < 0 || fe.off < 0 || fe.lb < 0 || fe.lw < fe.lb || (fe.len > 0 && fe.lb == 0)) {
            err = "line " + std::to_string(lineno) + " of '" + path + "' is not a faidx record"; ok = false; break;
        }
        out.push_back(std::move(fe));
    }
    std::free(line);
    std::fclose(fh);
    return ok;
}

// bytes of the file a sequence's bases and line ends take
int64_t raw_span(const FaiEntry &e) {
    if (e.len == 0) return 0;
    return (e.len - 1) / e.lb * e.lw + (e.len - 1) % e.lb + 1;
}

}  // namespace

// one sequence wanted of the file: its bytes [raw_off, raw_end), line geometry, and where its base 0 goes in the output
struct MdxFastaSeq { long long raw_off, raw_end, len, lb, lw, out_off; };

// .upper() of main.py:180 / align.py:32-33, then the resident reference's classes (mdx_kernels.hip encode_ref_kernel): the four
// bases as their letters, '-' and everything else as the two codes above 0x80
__device__ __forceinline__ uint8_t fasta_encode(uint32_t ch) {
    if (ch >= 'a' && ch <= 'z') ch -= 32;
    if (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') return (uint8_t)ch;
    return ch == '-' ? (uint8_t)0x84 : (uint8_t)0x85;
}

// A thread takes sixteen consecutive bytes of the piece [f0, f0 + n) of the file: which sequence they lie in (the sequences by
// file offset; a binary search for the first byte, a step forward where a sequence ends), line and column once by division,
// then byte by byte.
__global__ void __launch_bounds__(256) fasta_strip_kernel(const uint8_t *__restrict__ raw, long long f0, long long n,
                                                           const MdxFastaSeq *__restrict__ seqs, int n_seq, uint8_t *__restrict__ out) {
    const long long units = (n + 15) / 16;
    for (long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (long long)gridDim.x * blockDim.x) {
        const long long r0 = u * 16;
        const int m = (int)(n - r0 < 16 ? n - r0 : 16);
        uint4 w = make_uint4(0, 0, 0, 0);
        if (m == 16) w = *(const uint4 *)(raw + r0);
        else { uint8_t *wb = (uint8_t *)&w; for (int j = 0; j < m; j++) wb[j] = raw[r0 + j]; }
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
        long long p = f0 + r0;
        // the last sequence whose bytes begin at or in front of p
        int lo = 0, hi = n_seq;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seqs[mid].raw_off <= p) lo = mid; else hi = mid; }
        int k = lo;
        MdxFastaSeq s = seqs[k];
        long long line = 0, col = 0;
        bool have = false;
        for (int j = 0; j < m; j++, p++) {
            while (k + 1 < n_seq && seqs[k + 1].raw_off <= p) { k++; s = seqs[k]; have = false; }
            if (p < s.raw_off || p >= s.raw_end) { have = false; continue; }
            if (!have) { const long long rel = p - s.raw_off; line = rel / s.lw; col = rel - line * s.lw; have = true; }
            if (col < s.lb) out[s.out_off + line * s.lb + col] = fasta_encode((ww[j >> 2] >> (8 * (j & 3))) & 0xFFu);
            if (++col == s.lw) { col = 0; line++; }
        }
    }
}

extern "C" int mdx_fasta_index(const char *fasta_path, char *err_out, int32_t err_cap) {
    auto say = [&](const std::string &m, int code) {
        if (err_out && err_cap > 0) { std::snprintf(err_out, (size_t)err_cap, "%s", m.c_str()); }
        return code;
    };
    try {
        if (!fasta_path) return say("null path", MDX_ERR_ARG);
        const std::string fai = std::string(fasta_path) + ".fai";
        if (access(fai.c_str(), R_OK) == 0) return say("", MDX_OK);
        Mapped f;
        if (!f.open(fasta_path)) return say(std::string("cannot open '") + fasta_path + "'", MDX_ERR_ARG);
        std::vector<FaiEntry> idx;
        std::string err;
        if (!build_index(f, idx, err)) return say(err, MDX_ERR_ARG);
        // (written under another name and moved into place: a reader never finds half an index)
        const std::string tmp = fai + "." + std::to_string((long)getpid()) + ".tmp";
        FILE *fh = std::fopen(tmp.c_str(), "w");
        if (!fh) return say("cannot write '" + fai + "'", MDX_ERR_ARG);
        for (const FaiEntry &e : idx)
            std::fprintf(fh, "%s\t%lld\t%lld\t%lld\t%lld\n", e.name.c_str(), (long long)e.len, (long long)e.off, (long long)e.lb, (long long)e.lw);
        if (std::fclose(fh) != 0 || std::rename(tmp.c_str(), fai.c_str()) != 0) { std::remove(tmp.c_str()); return say("cannot write '" + fai + "'", MDX_ERR_ARG); }
        return say("", MDX_OK);
    } catch (...) {
        return say("out of memory", MDX_ERR_ARG);
    }
}

// The host side of mdx_set_reference_fasta (mdx_capi.cpp owns the context): index, the wanted sequences, the file's pieces to
// `stage` buffers and the kernel over each.  d_out: base 0 of sequence i goes to d_out[contig_off[i]] (contig_off: n + 1 sums of
// the lengths, filled here together with `lengths`).
int mdx_fasta_to_device(const char *fasta_path, int32_t n_contig, const char *const *names, int missing_ok, int64_t *lengths,
                        std::vector<int64_t> &contig_off, std::string &err, hipStream_t stream,
                        uint8_t *(*alloc_out)(void *, int64_t), void *alloc_arg) {
    std::string ierr(256, '\0');
    if (mdx_fasta_index(fasta_path, &ierr[0], (int32_t)ierr.size()) != MDX_OK) { err = ierr.c_str(); return MDX_ERR_ARG; }
    std::vector<FaiEntry> idx;
    if (!read_index(std::string(fasta_path) + ".fai", idx, err)) return MDX_ERR_ARG;
    std::unordered_map<std::string, size_t> by_name;
    for (size_t i = 0; i < idx.size(); i++) by_name.emplace(idx[i].name, i);      // (the first of two sequences of one name, as faidx)
    Mapped f;
    if (!f.open(fasta_path)) { err = std::string("cannot open '") + fasta_path + "'"; return MDX_ERR_ARG; }
    contig_off.assign((size_t)n_contig + 1, 0);
    std::vector<MdxFastaSeq> seqs;
    for (int i = 0; i < n_contig; i++) {
        const auto it = by_name.find(names[i] ? names[i] : "");
        int64_t len = 0;
        if (it == by_name.end()) {
            if (!missing_ok) { err = std::string("sequence '") + (names[i] ? names[i] : "") + "' not found in the FASTA file"; return MDX_ERR_ARG; }
        } else {
            const FaiEntry &e = idx[it->second];
            len = e.len;
            if (len > 0) {
                if (e.off + raw_span(e) > (int64_t)f.n) { err = "the index does not fit the file (sequence '" + e.name + "'): re-index it with 'samtools faidx'"; return MDX_ERR_ARG; }
                seqs.push_back(MdxFastaSeq{e.off, e.off + raw_span(e), len, e.lb, e.lw, contig_off[(size_t)i]});
            }
        }
        if (lengths) lengths[i] = len;
        contig_off[(size_t)i + 1] = contig_off[(size_t)i] + len;
    }
    uint8_t *d_out = alloc_out(alloc_arg, contig_off[(size_t)n_contig]);
    if (!d_out) { err = "out of device memory"; return MDX_ERR_HIP; }
    if (seqs.empty()) return MDX_OK;
    std::sort(seqs.begin(), seqs.end(), [](const MdxFastaSeq &a, const MdxFastaSeq &b) { return a.raw_off < b.raw_off; });
    // (the SAM specification wants the names of a header unique: a sequence wanted twice has one place too few)
    for (size_t j = 1; j < seqs.size(); j++)
        if (seqs[j].raw_off == seqs[j - 1].raw_off) { err = "a sequence of the FASTA file is named twice"; return MDX_ERR_ARG; }
    MdxFastaSeq *d_seqs = nullptr;
    uint8_t *stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    const size_t piece = []() -> size_t { const char *e = std::getenv("MDX_FASTA_PIECE_BYTES"); return e ? (size_t)std::max(4096, std::atoi(e)) & ~(size_t)15 : (size_t)256 << 20; }();
    int rc = MDX_OK;
    auto bad = [&](const char *what) { err = what; rc = MDX_ERR_HIP; };
    if (hipMalloc((void **)&d_seqs, seqs.size() * sizeof(MdxFastaSeq)) != hipSuccess ||
        hipMemcpyAsync(d_seqs, seqs.data(), seqs.size() * sizeof(MdxFastaSeq), hipMemcpyHostToDevice, stream) != hipSuccess) bad("upload of the sequence table failed");
    const long long lo = seqs.front().raw_off, hi = [&] { long long h = 0; for (const MdxFastaSeq &s : seqs) h = std::max(h, s.raw_end); return h; }();
    const size_t stage_bytes = std::min<size_t>(piece, (size_t)(hi - lo) + 16);
    for (int k = 0; k < 2 && rc == MDX_OK; k++)
        if (hipMalloc((void **)&stage[k], stage_bytes + 16) != hipSuccess || hipEventCreateWithFlags(&done[k], hipEventDisableTiming) != hipSuccess) bad("out of device memory");
    int turn = 0;
    bool used[2] = {false, false};
    size_t si = 0;      // first sequence that may reach into the piece
    for (long long f0 = lo; f0 < hi && rc == MDX_OK; f0 += (long long)piece) {
        const long long n = std::min<long long>((long long)piece, hi - f0);
        // (a piece none of the wanted sequences reaches into — a BAM file that names a few sequences of a large FASTA — stays on disk)
        while (si < seqs.size() && seqs[si].raw_end <= f0) si++;
        bool any = false;
        for (size_t j = si; j < seqs.size() && seqs[j].raw_off < f0 + n; j++) if (seqs[j].raw_end > f0) { any = true; break; }
        if (!any) continue;
        if (used[turn] && hipEventSynchronize(done[turn]) != hipSuccess) { bad("FASTA upload failed"); break; }
        if (hipMemcpyAsync(stage[turn], f.p + f0, (size_t)n, hipMemcpyHostToDevice, stream) != hipSuccess) { bad("FASTA upload failed"); break; }
        const long long units = (n + 15) / 16;
        const int grid = (int)std::min<long long>((units + 255) / 256, 16384);
        hipLaunchKernelGGL(fasta_strip_kernel, dim3(grid), dim3(256), 0, stream, stage[turn], f0, n, d_seqs, (int)seqs.size(), d_out);
        if (hipGetLastError() != hipSuccess || hipEventRecord(done[turn], stream) != hipSuccess) { bad("FASTA strip launch failed"); break; }
        used[turn] = true;
        turn ^= 1;
    }
    if (hipStreamSynchronize(stream) != hipSuccess && rc == MDX_OK) bad("FASTA upload failed");
    for (int k = 0; k < 2; k++) { if (stage[k]) (void)hipFree(stage[k]); if (done[k]) (void)hipEventDestroy(done[k]); }
    if (d_seqs) (void)hipFree(d_seqs);
    return rc;
}
