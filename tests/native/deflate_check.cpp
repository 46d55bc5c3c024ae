// Host build of the DEFLATE encoder of the GPU BGZF writer (mapdamage_amd/csrc/mdx_deflate.h) held against zlib's inflate:
// reads records {u32 n, n bytes} from a file, encodes each as a BGZF member, inflates it with zlib (gzip wrapper: the header,
// the CRC32 and ISIZE are checked by zlib itself) and compares; prints the bytes in and out.  tests/test_inflate_core.py.
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "mdx_crc32.h"
#include "mdx_deflate.h"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *fh = std::fopen(argv[1], "rb");
    if (!fh) return 2;
    static mdx_crc32::Tables tables;
    mdx_crc32::make_tables(tables);
    std::unique_ptr<mdx_deflate::Scratch> scr(new mdx_deflate::Scratch());
    std::unique_ptr<mdx_deflate::PieceScratch> pscr(new mdx_deflate::PieceScratch());
    std::vector<uint8_t> in, out(70000), back(70000);
    unsigned long total_in = 0, total_out = 0, total_one = 0;
    int cases = 0;
    for (;;) {
        uint32_t n = 0;
        if (std::fread(&n, 4, 1, fh) != 1) break;
        in.resize(n);
        if (n && std::fread(in.data(), 1, n, fh) != n) return 2;
        // (both forms: the member as one block, and in pieces as the device writes it)
        for (int form = 0; form < 2; form++) {
        const uint32_t sz = form ? mdx_deflate::bgzf_member_in_pieces(in.data(), n, out.data(), (uint32_t)out.size(), tables.tab[0], *pscr)
                                 : mdx_deflate::bgzf_member(in.data(), n, out.data(), (uint32_t)out.size(), tables.tab[0], *scr);
        if (!sz || sz > 65536) { std::printf("case %d: member of %u bytes for %u in\n", cases, sz, n); return 1; }
        if (out[16] + 256u * out[17] + 1u != sz) { std::printf("case %d: BSIZE\n", cases); return 1; }
        z_stream z;
        std::memset(&z, 0, sizeof z);
        if (inflateInit2(&z, 15 + 16) != Z_OK) return 2;       // gzip wrapper
        z.next_in = out.data(); z.avail_in = sz; z.next_out = back.data(); z.avail_out = (uInt)back.size();
        const int rc = inflate(&z, Z_FINISH);
        if (rc != Z_STREAM_END || z.total_out != n || z.avail_in != 0 || (n && std::memcmp(back.data(), in.data(), n) != 0)) {
            std::printf("case %d: zlib says %d (%s), %lu bytes out of %u, %u left\n", cases, rc, z.msg ? z.msg : "", z.total_out, n, z.avail_in);
            return 1;
        }
        inflateEnd(&z);
        if (form) { total_in += n; total_out += sz; } else total_one += sz;
        }
        cases++;
    }
    std::printf("ok %d cases, %lu bytes in, %lu out in pieces, %lu as one block each\n", cases, total_in, total_out, total_one);
    return 0;
}
