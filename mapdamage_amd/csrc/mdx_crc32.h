// CRC-32 (gzip, RFC 1952) of a BGZF block's inflated bytes on the GPU: 64 lanes take 1 KiB each, the partial
// values are joined with the "append n zero bytes" operator (GF(2) matrices for 2^k bytes, as in zlib's
// crc32_combine).  Host + device; tests/native/crc_check.cpp holds it against zlib.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MDX_CRC_HD __host__ __device__ __forceinline__
#else
#define MDX_CRC_HD inline
#endif

namespace mdx_crc32 {

enum { N_MATS = 17 };   // operators for 2^0 .. 2^16 zero bytes (a BGZF block holds at most 2^16)

struct Tables {
    uint32_t tab[4][256];        // slicing-by-4: tab[0] is the byte table, tab[k][i] = tab[0] applied k more times
    uint32_t mat[N_MATS][32];
};

inline uint32_t gf2_times(const uint32_t *mat, uint32_t vec) {
    uint32_t sum = 0;
    for (int b = 0; vec; b++, vec >>= 1) if (vec & 1u) sum ^= mat[b];
    return sum;
}
inline void gf2_square(uint32_t *sq, const uint32_t *mat) { for (int n = 0; n < 32; n++) sq[n] = gf2_times(mat, mat[n]); }

inline void make_tables(Tables &t) {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        t.tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
        for (int k = 1; k < 4; k++) t.tab[k][i] = t.tab[0][t.tab[k - 1][i] & 0xFFu] ^ (t.tab[k - 1][i] >> 8);
    uint32_t a[32], b[32];
    a[0] = 0xEDB88320u;                                   // one zero bit
    for (int n = 1; n < 32; n++) a[n] = 1u << (n - 1);
    gf2_square(b, a);                                     // two
    gf2_square(a, b);                                     // four
    gf2_square(t.mat[0], a);                              // eight bits = one byte
    for (int k = 1; k < N_MATS; k++) gf2_square(t.mat[k], t.mat[k - 1]);
}

// tab: the four 256-entry tables, one after the other
MDX_CRC_HD uint32_t crc_bytes(const uint32_t *tab, const uint8_t *p, uint32_t n) {
    uint32_t c = 0xFFFFFFFFu;
    uint32_t i = 0;
    for (; i + 4u <= n; i += 4u) {
        typedef uint32_t u32u __attribute__((aligned(1)));
        c ^= *(const u32u *)(p + i);
        c = tab[768 + (c & 0xFFu)] ^ tab[512 + ((c >> 8) & 0xFFu)] ^ tab[256 + ((c >> 16) & 0xFFu)] ^ tab[c >> 24];
    }
    for (; i < n; i++) c = tab[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

// the CRC of A followed by n zero bytes ... as zlib's crc32_combine uses it: crc(A || B) = shift(crc(A), |B|) ^ crc(B)
MDX_CRC_HD uint32_t shift(const uint32_t (*mat)[32], uint32_t crc, uint32_t nbytes) {
    for (int k = 0; nbytes && k < N_MATS; k++, nbytes >>= 1) {
        if (!(nbytes & 1u)) continue;
        uint32_t sum = 0;
        for (int b = 0; b < 32; b++) sum ^= ((crc >> b) & 1u) ? mat[k][b] : 0u;
        crc = sum;
    }
    return crc;
}

}  // namespace mdx_crc32
