// Host check of mapdamage_amd/csrc/mdx_inflate.h against zlib: every raw DEFLATE stream given on stdin as
// [u32 compressed length][u32 expected length][compressed bytes][expected bytes] records.
// Build: g++ -O2 -I mapdamage_amd/csrc tests/native/inflate_check.cpp -lz -o /tmp/inflate_check
#include <cstdio>
#include <cstring>
#include <vector>
#include <zlib.h>
#include "mdx_inflate.h"

int main() {
    std::vector<uint8_t> in, want, got(65536), ring(mdx_inflate::RING);
    static mdx_inflate::Tables t;
    long n = 0, bad = 0;
    for (;;) {
        uint32_t hdr[2];
        if (fread(hdr, 4, 2, stdin) != 2) break;
        const bool expect_error = hdr[1] == 0xFFFFFFFFu;     // garbage: no expected bytes follow
        in.resize(hdr[0]); want.resize(expect_error ? 0 : hdr[1]);
        if (hdr[0] && fread(in.data(), 1, hdr[0], stdin) != hdr[0]) return 2;
        if (!want.empty() && fread(want.data(), 1, want.size(), stdin) != want.size()) return 2;
        const int r = mdx_inflate::inflate_block(in.data(), hdr[0], ring.data(), got.data(), 65536, t);
        n++;
        if (expect_error) {
            // random bytes: whatever zlib makes of them (a few are valid streams)
            std::vector<uint8_t> z(65536 + 1);
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            inflateInit2(&zs, -15);
            zs.next_in = in.data(); zs.avail_in = hdr[0]; zs.next_out = z.data(); zs.avail_out = (uInt)z.size();
            const int zr = inflate(&zs, Z_FINISH);
            const long zn = (long)zs.total_out;
            inflateEnd(&zs);
            const bool z_ok = zr == Z_STREAM_END && zn <= 65536;
            if (z_ok != (r >= 0) || (z_ok && (r != zn || memcmp(got.data(), z.data(), (size_t)zn) != 0))) {
                bad++; printf("stream %ld: garbage: zlib %d (%ld bytes), here %d\n", n, zr, zn, r);
            }
            continue;
        }
        if (r != (int)hdr[1] || memcmp(got.data(), want.data(), hdr[1]) != 0) { bad++; printf("stream %ld: got %d want %u\n", n, r, hdr[1]); }
    }
    printf("%ld streams, %ld bad\n", n, bad);
    return bad ? 1 : 0;
}
