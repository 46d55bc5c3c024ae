// Raw DEFLATE (RFC 1951) ENCODER for one BGZF block — at most 0xFF00 bytes in, at most 64 KiB out (SAM specification 4.1) —
// written, like the decoder of mdx_inflate.h, so that the same code runs on the host (tests against zlib's inflate) and on the
// GPU (mdx_gbam.hip: bgzf_deflate_kernel, one block per lane — a BAM file of a gigabyte is fifteen thousand independent blocks).
//
// Replaces, for the rescaling pass, zlib's deflate() behind pysam's AlignmentFile(..., "wb") (rescale.py:290-291, :344 of the
// reference: every record is written back): the output's BGZF blocks were 2.1 of the pass's 3.5 seconds on sixteen host
// threads.  One final block per member: LZ77 with a hash of four bytes and one candidate per position (the
// greedy parse of zlib's fast levels), then a dynamic Huffman code built from the block's own symbol counts (lengths limited
// to 15 / 7 bits), or a stored block when that is not smaller.  Any inflater reads it; it is not bit-identical to zlib's
// output, and a few percent larger than zlib's level 6.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MDX_DF_HD __host__ __device__
#else
#define MDX_DF_HD
#endif

#ifndef MDX_DEFLATE_PIECES
#define MDX_DEFLATE_PIECES 4
#endif

namespace mdx_deflate {

enum { MAX_IN = 0xFF00, HASH_BITS = 12, HASH_SIZE = 1 << HASH_BITS, MIN_MATCH = 4, MAX_MATCH = 258, WINDOW = 32768,
       N_LL = 286, N_D = 30, N_CL = 19,
       // A member in PIECES pieces, a lane each on the device: every piece one deflate block of its own code whose matches
       // may reach back into the piece in front (the lane hashes that one's positions first), every piece but the last closed
       // with an empty stored block — zlib's Z_SYNC_FLUSH — so that the next one starts on a byte.  Four lanes per member
       // instead of one: 0.6 % more bytes.
       PIECES = MDX_DEFLATE_PIECES, PIECE = MAX_IN / PIECES };
static_assert(PIECE * PIECES == MAX_IN, "the pieces tile a member");

// Per-block working memory (the caller's: on the device a slot per lane in HBM).  A token: a literal byte, or
// 0x80000000 | (length - 3) << 16 | (distance - 1).
template <int TOKENS>
struct ScratchT {
    uint16_t head[HASH_SIZE];          // last position + 1 with this hash (0: none)
    uint32_t tokens[TOKENS];
    uint32_t freq_ll[N_LL], freq_d[N_D], freq_cl[N_CL];
    uint8_t len_ll[N_LL], len_d[N_D], len_cl[N_CL];
    uint16_t code_ll[N_LL], code_d[N_D], code_cl[N_CL];     // bit-reversed: ready for the LSB-first stream
    uint16_t order[N_LL];              // symbols by frequency (scratch of the code construction)
    uint32_t work[3 * N_LL];
    uint8_t cl_syms[N_LL + N_D];       // the two length arrays as code-length symbols (16 / 17 / 18: repeats), and their extra bits
    uint8_t cl_extra[N_LL + N_D];
};
typedef ScratchT<MAX_IN> Scratch;           // a whole member by one encoder
typedef ScratchT<PIECE> PieceScratch;       // a piece of one

struct BitWriter {
    uint8_t *out;
    uint32_t cap, pos;
    uint64_t acc;
    int nbits;
    bool overflow;
    MDX_DF_HD void put(uint32_t value, int n) {
        acc |= (uint64_t)value << nbits;
        nbits += n;
        while (nbits >= 8) {
            if (pos < cap) out[pos] = (uint8_t)acc; else overflow = true;
            pos++;
            acc >>= 8; nbits -= 8;
        }
    }
    MDX_DF_HD void finish() { if (nbits > 0) put(0, 8 - nbits); }
};

MDX_DF_HD inline uint32_t reverse_bits(uint32_t code, int len) {
#if defined(__clang__)
    return len ? __builtin_bitreverse32(code) >> (32 - len) : 0u;
#else
    uint32_t r = 0;
    for (int i = 0; i < len; i++) { r = (r << 1) | (code & 1u); code >>= 1; }
    return r;
#endif
}

MDX_DF_HD inline int floor_log2(uint32_t v) {       // (v > 0)
    return 31 - __builtin_clz(v);
}

// length 3..258 -> (code 257..285, number of extra bits, their value); RFC 1951 section 3.2.5
MDX_DF_HD inline void length_code(uint32_t len, uint32_t &code, int &ebits, uint32_t &extra) {
    const uint32_t l = len - 3;
    if (l < 8) { code = 257 + l; ebits = 0; extra = 0; return; }
    if (len == 258) { code = 285; ebits = 0; extra = 0; return; }
    const int e = floor_log2(l) - 2;
    code = 257 + 4 * (uint32_t)(e + 1) + ((l >> e) & 3u);
    ebits = e;
    extra = l & ((1u << e) - 1u);
}
// distance 1..32768 -> (code 0..29, extra bits, value)
MDX_DF_HD inline void dist_code(uint32_t dist, uint32_t &code, int &ebits, uint32_t &extra) {
    const uint32_t d = dist - 1;
    if (d < 4) { code = d; ebits = 0; extra = 0; return; }
    const int e = floor_log2(d) - 1;
    code = 2 * (uint32_t)(e + 1) + ((d >> e) & 1u);
    ebits = e;
    extra = d & ((1u << e) - 1u);
}

// Code lengths of a prefix code over n symbols with the given frequencies, none longer than max_len; symbols that do not
// occur get length 0.  (The lengths of a minimum-redundancy code — the two-queue construction over the symbols sorted by
// frequency — and, where the tree is deeper than max_len, the counts per length folded back under the Kraft sum, the way
// every deflate encoder does it; the lengths go to the symbols in order of frequency, the rarest the longest.)
MDX_DF_HD inline void build_lengths(const uint32_t *freq, int n, int max_len, uint8_t *len, uint16_t *order, uint32_t *work) {
    int m = 0;
    for (int i = 0; i < n; i++) { len[i] = 0; if (freq[i]) order[m++] = (uint16_t)i; }
    if (m == 0) return;
    if (m == 1) { len[order[0]] = 1; return; }
    // symbols by ascending frequency (insertion sort: a few hundred symbols; ties by symbol, so that the result is one)
    for (int i = 1; i < m; i++) {
        const uint16_t s = order[i];
        int j = i - 1;
        while (j >= 0 && (freq[order[j]] > freq[s] || (freq[order[j]] == freq[s] && order[j] > s))) { order[j + 1] = order[j]; j--; }
        order[j + 1] = s;
    }
    // two queues: the leaves in order of frequency, and the internal nodes in the order they are made (their weights never
    // decrease).  Node k < m is leaf order[k], node m + j the j-th internal node; work[k] = parent of node k, wint[j] = weight
    // of internal node j.
    uint32_t *parent = work, *wint = work + 2 * n;
    int leaf = 0, node = 0;
    for (int made = 0; made < m - 1; made++) {
        uint32_t sum = 0;
        for (int c = 0; c < 2; c++) {
            const bool take_leaf = leaf < m && (node >= made || freq[order[leaf]] <= wint[node]);
            if (take_leaf) { sum += freq[order[leaf]]; parent[leaf] = (uint32_t)(m + made); leaf++; }
            else { sum += wint[node]; parent[m + node] = (uint32_t)(m + made); node++; }
        }
        wint[made] = sum;
    }
    // depths, from the root (the last node made) down: a parent has a higher number than its children
    const int root = 2 * m - 2;
    parent[root] = 0;
    for (int k = root - 1; k >= 0; k--) parent[k] = parent[parent[k]] + 1;
    uint32_t count[64];
    for (int d = 0; d < 64; d++) count[d] = 0;
    for (int k = 0; k < m; k++) count[parent[k] < 63 ? parent[k] : 63]++;
    // fold the lengths beyond max_len back (Kraft: sum of count[d] * 2^(max_len - d) must be 2^max_len)
    for (int d = max_len + 1; d < 64; d++) { count[max_len] += count[d]; count[d] = 0; }
    uint32_t total = 0;
    for (int d = max_len; d >= 1; d--) total += count[d] << (max_len - d);
    while (total > (1u << max_len)) {
        count[max_len]--;
        for (int d = max_len - 1; d >= 1; d--)
            if (count[d]) { count[d]--; count[d + 1] += 2; break; }
        total--;
    }
    // the rarest symbols take the longest codes
    int at = 0;
    for (int d = max_len; d >= 1; d--)
        for (uint32_t k = 0; k < count[d]; k++) len[order[at++]] = (uint8_t)d;
}

// canonical codes of the lengths (RFC 1951 section 3.2.2), bit-reversed for the LSB-first stream
MDX_DF_HD inline void assign_codes(const uint8_t *len, int n, uint16_t *code) {
    uint32_t count[16], next[16];
    for (int d = 0; d < 16; d++) count[d] = 0;
    for (int i = 0; i < n; i++) count[len[i]]++;
    count[0] = 0;
    uint32_t c = 0;
    next[0] = 0;
    for (int d = 1; d < 16; d++) { c = (c + count[d - 1]) << 1; next[d] = c; }
    for (int i = 0; i < n; i++) code[i] = len[i] ? (uint16_t)reverse_bits(next[len[i]]++, len[i]) : 0;
}

MDX_DF_HD inline uint32_t load32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// in[lo .. hi) of a member that starts at in[0] -> one deflate block in out[0 .. cap) — the final one of its stream or not;
// a block that is not final is followed by an empty stored block, so that whatever comes next starts on a byte (the stream
// must stand on a byte where this block begins).  The matches may reach back to in[hist_lo]: those positions are hashed
// first.  Returns the bytes written, or 0 when cap does not hold them (hi - lo + 16 bytes always suffice).
template <class S>
MDX_DF_HD inline uint32_t deflate_piece(const uint8_t *in, uint32_t lo, uint32_t hi, uint32_t hist_lo, bool final, uint8_t *out, uint32_t cap, S &s) {
    const uint32_t n = hi;
    // ---- LZ77: greedy, one candidate per position (the last one with the same hash of four bytes)
    for (int i = 0; i < HASH_SIZE; i++) s.head[i] = 0;
    for (int i = 0; i < N_LL; i++) s.freq_ll[i] = 0;
    for (int i = 0; i < N_D; i++) s.freq_d[i] = 0;
    for (uint32_t q = hist_lo; q + MIN_MATCH <= n && q < lo; q += 2) {
        const uint32_t hv = (load32(in + q) * 2654435761u) >> (32 - HASH_BITS);
        s.head[hv] = (uint16_t)(q + 1);
    }
    uint32_t n_tok = 0, p = lo;
    while (p < n) {
        uint32_t best = 0, dist = 0;
        if (p + MIN_MATCH <= n) {
            const uint32_t v = load32(in + p);
            const uint32_t h = (v * 2654435761u) >> (32 - HASH_BITS);
            const uint32_t c = s.head[h];
            s.head[h] = (uint16_t)(p + 1);
            if (c && p + 1 - c <= WINDOW && load32(in + c - 1) == v) {
                const uint32_t q = c - 1, lim = n - p < MAX_MATCH ? n - p : MAX_MATCH;
                uint32_t l = 4;
                while (l < lim && in[q + l] == in[p + l]) l++;
                best = l; dist = p - q;
            }
        }
        if (best >= MIN_MATCH) {
            uint32_t code, extra; int eb;
            length_code(best, code, eb, extra);
            s.freq_ll[code]++;
            dist_code(dist, code, eb, extra);
            s.freq_d[code]++;
            s.tokens[n_tok++] = 0x80000000u | ((best - 3) << 16) | (dist - 1);
            // (the positions inside the match enter the hash too, every other one: the candidates of what follows)
            for (uint32_t k = 2; k < best && p + k + MIN_MATCH <= n; k += 2) {
                const uint32_t hv = (load32(in + p + k) * 2654435761u) >> (32 - HASH_BITS);
                s.head[hv] = (uint16_t)(p + k + 1);
            }
            p += best;
        } else {
            s.freq_ll[in[p]]++;
            s.tokens[n_tok++] = in[p];
            p++;
        }
    }
    s.freq_ll[256] = 1;     // end of block
    // ---- the two codes
    build_lengths(s.freq_ll, N_LL, 15, s.len_ll, s.order, s.work);
    build_lengths(s.freq_d, N_D, 15, s.len_d, s.order, s.work);
    int n_ll = N_LL, n_d = N_D;
    while (n_ll > 257 && s.len_ll[n_ll - 1] == 0) n_ll--;
    while (n_d > 1 && s.len_d[n_d - 1] == 0) n_d--;
    assign_codes(s.len_ll, N_LL, s.code_ll);
    assign_codes(s.len_d, N_D, s.code_d);
    // ---- their lengths, run-length coded into code-length symbols (section 3.2.7)
    uint8_t all[N_LL + N_D];
    for (int i = 0; i < n_ll; i++) all[i] = s.len_ll[i];
    for (int i = 0; i < n_d; i++) all[n_ll + i] = s.len_d[i];
    const int n_all = n_ll + n_d;
    for (int i = 0; i < N_CL; i++) s.freq_cl[i] = 0;
    int n_cl = 0;
    for (int i = 0; i < n_all;) {
        const uint8_t v = all[i];
        int run = 1;
        while (i + run < n_all && all[i + run] == v) run++;
        if (v == 0 && run >= 3) {
            const int r = run > 138 ? 138 : run;
            if (r <= 10) { s.cl_syms[n_cl] = 17; s.cl_extra[n_cl] = (uint8_t)(r - 3); }
            else { s.cl_syms[n_cl] = 18; s.cl_extra[n_cl] = (uint8_t)(r - 11); }
            s.freq_cl[s.cl_syms[n_cl]]++; n_cl++;
            i += r;
        } else if (v != 0 && run >= 4) {
            // the length itself, then repeats of it (3 to 6 at a time)
            s.cl_syms[n_cl] = v; s.cl_extra[n_cl] = 0; s.freq_cl[v]++; n_cl++;
            int left = run - 1;
            i += 1;
            while (left >= 3) {
                const int r = left > 6 ? 6 : left;
                s.cl_syms[n_cl] = 16; s.cl_extra[n_cl] = (uint8_t)(r - 3); s.freq_cl[16]++; n_cl++;
                left -= r; i += r;
            }
            // (what is left, fewer than three, goes round the loop again as plain lengths)
        } else {
            s.cl_syms[n_cl] = v; s.cl_extra[n_cl] = 0; s.freq_cl[v]++; n_cl++;
            i += 1;
        }
    }
    build_lengths(s.freq_cl, N_CL, 7, s.len_cl, s.order, s.work);
    assign_codes(s.len_cl, N_CL, s.code_cl);
    const uint8_t cl_order[N_CL] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int n_clen = N_CL;
    while (n_clen > 4 && s.len_cl[cl_order[n_clen - 1]] == 0) n_clen--;
    // ---- the stream
    BitWriter bw{out, cap, 0, 0, 0, false};
    bw.put(final ? 1u : 0u, 1);     // BFINAL
    bw.put(2, 2);           // BTYPE = dynamic Huffman
    bw.put((uint32_t)(n_ll - 257), 5);
    bw.put((uint32_t)(n_d - 1), 5);
    bw.put((uint32_t)(n_clen - 4), 4);
    for (int i = 0; i < n_clen; i++) bw.put(s.len_cl[cl_order[i]], 3);
    for (int i = 0; i < n_cl; i++) {
        const uint8_t sym = s.cl_syms[i];
        bw.put(s.code_cl[sym], s.len_cl[sym]);
        if (sym == 16) bw.put(s.cl_extra[i], 2);
        else if (sym == 17) bw.put(s.cl_extra[i], 3);
        else if (sym == 18) bw.put(s.cl_extra[i], 7);
    }
    for (uint32_t t = 0; t < n_tok; t++) {
        const uint32_t tok = s.tokens[t];
        if (tok & 0x80000000u) {
            uint32_t code, extra; int eb;
            length_code(((tok >> 16) & 0xFFu) + 3, code, eb, extra);
            bw.put(s.code_ll[code], s.len_ll[code]);
            if (eb) bw.put(extra, eb);
            dist_code((tok & 0xFFFFu) + 1, code, eb, extra);
            bw.put(s.code_d[code], s.len_d[code]);
            if (eb) bw.put(extra, eb);
        } else {
            bw.put(s.code_ll[tok], s.len_ll[tok]);
        }
    }
    bw.put(s.code_ll[256], s.len_ll[256]);
    const uint32_t len = hi - lo;
    if (!final) {
        // an empty stored block: three header bits, the padding to the next byte, LEN = 0, NLEN = 0xFFFF
        bw.put(0, 3);
        bw.finish();
        bw.put(0xFFFF0000u, 32);
    } else bw.finish();
    if (!bw.overflow && bw.pos < len + 5) return bw.pos;
    // ---- not smaller than the bytes themselves: a stored block (it stands on a byte, and what follows it does)
    if (cap < len + 5) return 0;
    out[0] = final ? 1 : 0; // BFINAL, BTYPE = 00, the rest of the byte padding
    out[1] = (uint8_t)len; out[2] = (uint8_t)(len >> 8);
    out[3] = (uint8_t)~len; out[4] = (uint8_t)(~len >> 8);
    for (uint32_t i = 0; i < len; i++) out[5 + i] = in[lo + i];
    return len + 5;
}

// in[0 .. n) -> one raw DEFLATE stream (a single final block); 0 when cap does not hold it (n + 5 bytes always suffice)
MDX_DF_HD inline uint32_t deflate_block(const uint8_t *in, uint32_t n, uint8_t *out, uint32_t cap, Scratch &s) {
    if (n > MAX_IN) return 0;
    return deflate_piece(in, 0, n, 0, true, out, cap, s);
}
// ... and the same bytes as the device writes them: PIECES blocks (mdx_gbam.hip gives every piece a lane of its own; here one
// after the other), each hashing the piece in front of it first
MDX_DF_HD inline uint32_t deflate_block_in_pieces(const uint8_t *in, uint32_t n, uint8_t *out, uint32_t cap, PieceScratch &s) {
    if (n > MAX_IN) return 0;
    uint32_t at = 0;
    for (uint32_t lo = 0; lo < n || lo == 0; lo += PIECE) {
        const uint32_t hi = lo + PIECE < n ? lo + PIECE : n;
        const uint32_t got = deflate_piece(in, lo, hi, lo >= PIECE ? lo - PIECE : 0, hi == n, out + at, cap - at, s);
        if (!got) return 0;
        at += got;
        if (hi == n) break;
    }
    return at;
}

// CRC-32 (gzip) of n bytes with the byte table `tab` (mdx_crc32::Tables::tab[0])
MDX_DF_HD inline uint32_t crc32_bytes(const uint32_t *tab, const uint8_t *p, uint32_t n) {
    uint32_t c = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n; i++) c = tab[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

// gzip header with the BC subfield in front of `body` bytes of deflate stream at out + 18, CRC32 and ISIZE behind them
MDX_DF_HD inline uint32_t bgzf_wrap(const uint8_t *in, uint32_t n, uint8_t *out, uint32_t body, const uint32_t *crc_tab) {
    const uint32_t total = body + 26;
    const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    for (int i = 0; i < 16; i++) out[i] = head[i];
    out[16] = (uint8_t)(total - 1); out[17] = (uint8_t)((total - 1) >> 8);
    const uint32_t crc = crc32_bytes(crc_tab, in, n);
    uint8_t *t = out + 18 + body;
    t[0] = (uint8_t)crc; t[1] = (uint8_t)(crc >> 8); t[2] = (uint8_t)(crc >> 16); t[3] = (uint8_t)(crc >> 24);
    t[4] = (uint8_t)n; t[5] = (uint8_t)(n >> 8); t[6] = (uint8_t)(n >> 16); t[7] = (uint8_t)(n >> 24);
    return total;
}

// One BGZF member (SAM specification 4.1): the 18-byte gzip header with the BC subfield, the deflate stream, CRC32 and ISIZE.
// out must hold n + 5 + 26 bytes; returns the member's size.
MDX_DF_HD inline uint32_t bgzf_member(const uint8_t *in, uint32_t n, uint8_t *out, uint32_t cap, const uint32_t *crc_tab, Scratch &s) {
    if (cap < 26) return 0;
    const uint32_t room = cap - 26 < 65535u - 25u ? cap - 26 : 65535u - 26u + 1u;   // BSIZE is sixteen bits: the member at most 65536 bytes
    const uint32_t body = deflate_block(in, n, out + 18, room, s);
    if (!body) return 0;
    return bgzf_wrap(in, n, out, body, crc_tab);
}
// ... in PIECES pieces (what the device writes, one lane per piece)
MDX_DF_HD inline uint32_t bgzf_member_in_pieces(const uint8_t *in, uint32_t n, uint8_t *out, uint32_t cap, const uint32_t *crc_tab, PieceScratch &s) {
    if (cap < 26) return 0;
    const uint32_t room = cap - 26 < 65535u - 25u ? cap - 26 : 65535u - 26u + 1u;
    const uint32_t body = deflate_block_in_pieces(in, n, out + 18, room, s);
    if (!body) return 0;
    return bgzf_wrap(in, n, out, body, crc_tab);
}

}  // namespace mdx_deflate
