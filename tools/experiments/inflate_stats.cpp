// What the DEFLATE decoder of the device path meets in a BAM file (host build of csrc/mdx_inflate.h with its counters on):
// literals, matches and their lengths, matches by kind of copy, codes behind the fast tables and the bits walked for them.
// Build: g++ -O2 -I mapdamage_amd/csrc tools/experiments/inflate_stats.cpp -o /tmp/inflate_stats ; /tmp/inflate_stats file.bam [blocks]
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
static long g_stat[16];
#define MDX_INFLATE_STAT(what, n) (g_stat[(what)] += (long)(n))
#include "mdx_inflate.h"

int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    const long want = argc > 2 ? atol(argv[2]) : 2000;
    std::vector<uint8_t> in(65536 + 64), got(65536), ring(mdx_inflate::RING);
    static mdx_inflate::Tables t;
    long blocks = 0, out = 0, comp = 0;
    uint8_t h[18];
    while (blocks < want && fread(h, 1, 18, f) == 18) {
        const uint32_t bsize = (uint32_t)(h[16] | (h[17] << 8)) + 1u;      // (BGZF: one BC subfield, SAM specification 4.1)
        const uint32_t payload = bsize - 18u - 8u;
        if (fread(in.data(), 1, payload + 8u, f) != payload + 8u) break;
        const int r = mdx_inflate::inflate_block(in.data(), payload, ring.data(), got.data(), 65536, t);
        if (r < 0) { printf("block %ld: %d\n", blocks, r); return 2; }
        blocks++; out += r; comp += payload;
    }
    const char *names[] = {"literals", "matches", "match bytes", "matches beyond the ring", "matches of <= 64 bytes in front of their source", "matches with a period below 64",
                           "distance codes behind the fast table", "literal / length codes behind the fast table", "bits walked"};
    printf("%ld blocks, %ld bytes in, %ld out\n", blocks, comp, out);
    for (int i = 0; i < 9; i++) printf("  %-55s %12ld   %.4f per output byte\n", names[i], g_stat[i], (double)g_stat[i] / (double)out);
    printf("  bytes per match %.2f, literals per match %.2f\n", (double)g_stat[2] / g_stat[1], (double)g_stat[0] / g_stat[1]);
    return 0;
}
