// Host check of mapdamage_amd/csrc/mdx_crc32.h against zlib: buffers of every length class cut into 1 KiB pieces,
// piece CRCs joined with the shift operator, compared with crc32() of the whole.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <zlib.h>
#include "mdx_crc32.h"

int main() {
    static mdx_crc32::Tables t;
    mdx_crc32::make_tables(t);
    srand(5);
    long bad = 0, n = 0;
    const unsigned lens[] = {0, 1, 2, 7, 1023, 1024, 1025, 2048, 4097, 30000, 65535, 65536};
    for (unsigned len : lens)
        for (int rep = 0; rep < 6; rep++) {
            std::vector<uint8_t> d(len);
            for (auto &x : d) x = (uint8_t)(rep == 0 ? 0 : rand());
            uint32_t total = 0;
            bool first = true;
            for (unsigned o = 0; o < len || first; o += 1024) {
                const unsigned m = len - o < 1024 ? len - o : 1024;
                const uint32_t c = mdx_crc32::crc_bytes(&t.tab[0][0], d.data() + o, m);
                total = first ? c : (mdx_crc32::shift(t.mat, total, m) ^ c);
                first = false;
                if (len == 0) break;
            }
            const uint32_t want = (uint32_t)crc32(0L, d.data(), len);
            n++;
            if (total != want) { bad++; printf("len %u: got %08x want %08x\n", len, total, want); }
        }
    printf("%ld buffers, %ld bad\n", n, bad);
    return bad ? 1 : 0;
}
