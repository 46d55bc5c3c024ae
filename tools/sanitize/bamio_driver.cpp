// Sanitizer driver for the native BAM decoder (host code only): decodes a file in one piece and in chunks of
// several sizes and checks that the chunks add up to the one-piece result.  Built and run by
// tools/sanitize/run_bamio.sh with -fsanitize=address,undefined (CPU build only; no GPU involved).
#include "../../include/mdx.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static int fail(const char *what) { std::fprintf(stderr, "FAIL: %s\n", what); return 1; }

int main(int argc, char **argv) {
    if (argc < 2) return fail("usage: bamio_driver file.bam [expect-error]");
    const bool expect_error = argc > 2;
    mdx_bam *whole = nullptr;
    int rc = mdx_bam_read(argv[1], 5, &whole);
    if (expect_error) {
        std::printf("one piece: rc %d (%s)\n", rc, mdx_bam_error(whole));
        mdx_bam_free(whole);
        mdx_bam_stream *st = nullptr;
        rc = mdx_bam_open(argv[1], 3, &st);
        int rc2 = rc;
        while (rc2 == 0) {
            mdx_bam *chunk = nullptr;
            rc2 = mdx_bam_next(st, 4096, &chunk);
            if (!chunk) break;
            mdx_bam_free(chunk);
        }
        std::printf("chunked: open rc %d, last rc %d (%s)\n", rc, rc2, mdx_bam_error(mdx_bam_stream_header(st)));
        mdx_bam_close(st);
        return (rc2 != 0) ? 0 : fail("expected an error");
    }
    if (rc != 0) return fail(mdx_bam_error(whole));
    mdx_batch w;
    const int32_t *wrg = nullptr;
    mdx_bam_batch(whole, &w, nullptr, nullptr, &wrg, nullptr);
    std::printf("one piece: %lld records, %lld bases, %lld cigar ops\n", (long long)w.n_reads, (long long)w.n_bases, (long long)w.n_cigar);
    const long long sizes[] = {64, 1000, 70000, 3 << 20, 1LL << 40};
    for (long long chunk_bytes : sizes) {
        mdx_bam_stream *st = nullptr;
        if (mdx_bam_open(argv[1], 3, &st) != 0) return fail(mdx_bam_error(mdx_bam_stream_header(st)));
        if (std::strcmp(mdx_bam_header_text(mdx_bam_stream_header(st)), mdx_bam_header_text(whole)) != 0) return fail("header text");
        long long n = 0, nb = 0, nc = 0, chunks = 0;
        for (;;) {
            mdx_bam *chunk = nullptr;
            if (mdx_bam_next(st, chunk_bytes, &chunk) != 0) return fail(mdx_bam_error(mdx_bam_stream_header(st)));
            if (!chunk) break;
            mdx_batch c;
            mdx_bam_batch(chunk, &c, nullptr, nullptr, nullptr, nullptr);
            if (n + c.n_reads > w.n_reads) return fail("too many records");
            if (std::memcmp(c.flag, (const uint16_t *)w.flag + n, (size_t)c.n_reads * 2) != 0) return fail("flag");
            if (std::memcmp(c.pos, (const int32_t *)w.pos + n, (size_t)c.n_reads * 4) != 0) return fail("pos");
            if (std::memcmp(c.seq, (const uint8_t *)w.seq + nb, (size_t)c.n_bases) != 0) return fail("seq");
            if (std::memcmp(c.qual, (const uint8_t *)w.qual + nb, (size_t)c.n_bases) != 0) return fail("qual");
            if (std::memcmp(c.cigar, (const uint32_t *)w.cigar + nc, (size_t)c.n_cigar * 4) != 0) return fail("cigar");
            n += c.n_reads; nb += c.n_bases; nc += c.n_cigar; chunks++;
            mdx_bam_free(chunk);
        }
        mdx_bam_close(st);
        if (n != w.n_reads || nb != w.n_bases || nc != w.n_cigar) return fail("chunks do not add up");
        std::printf("chunk_bytes %lld: %lld chunks, equal\n", chunk_bytes, chunks);
    }
    mdx_bam_free(whole);
    return 0;
}
