// Raw DEFLATE (RFC 1951) decoder for one BGZF block — at most 64 KiB out (SAM specification 4.1) — written so
// that the same code runs on the host (tests against zlib) and, one block per wavefront, on the GPU
// (mdx_gbam.hip).  On the device every lane of the wavefront executes the decoder with the same values (the
// Huffman state machine is serial by nature; it costs the same as one lane executing it), which lets the lanes
// share the work wherever there is any: a match is copied 64 bytes at a time, the tables are filled 64 entries at a
// time, and the compressed bytes are fetched 512 at a time (each lane holds eight of them).
//
// Replaces, for the GPU decode path, zlib's inflate() behind pysam.AlignmentFile (reader.py:20-46 of the
// reference).  Not a general inflater: one block, all of its input present, output window = the block itself.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MDX_HD __host__ __device__ __forceinline__
#else
#define MDX_HD inline
#endif

#if defined(__HIPCC__)
#define MDX_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define MDX_NOINLINE __attribute__((noinline))
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define MDX_ON_DEVICE 1
#if !defined(__gfx950__)
#error "the s_waitcnt immediates of this file are gfx9 encodings (vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8]); the tree targets gfx950 only"
#endif
#else
#define MDX_ON_DEVICE 0
#endif

namespace mdx_inflate {

// A value every lane of the wavefront holds alike, told to the compiler: what follows from it is computed once, on
// the scalar unit, instead of 64 times over on the vector unit (the decoder's state is such a value throughout).
MDX_HD uint32_t uni(uint32_t v) {
#if MDX_ON_DEVICE
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#else
    return v;
#endif
}

#ifndef MDX_FAST_LL
#define MDX_FAST_LL 11
#endif
#ifndef MDX_FAST_D
#define MDX_FAST_D 8
#endif
enum { FAST_LL = MDX_FAST_LL, FAST_D = MDX_FAST_D };   // bits resolved by one table lookup (longer codes: canonical walk)

// Decoder tables of one block; on the device they live in the LDS (one set per wavefront).
struct Tables {
    uint16_t fast_ll[1 << FAST_LL];  // (symbol << 4) | code length, 0 = longer than FAST_LL bits
    uint16_t fast_d[1 << FAST_D];
    uint16_t count_ll[16], count_d[16];
    uint16_t sym_ll[288], sym_d[32];
    uint8_t lens[352];               // code lengths while a dynamic header is read: [code-length code 19][..32][literal/length + distance <= 316]
};

// lane id and width of the group that executes the decoder in lock step (1 on the host)
MDX_HD int lane_id() {
#if MDX_ON_DEVICE
    return (int)(threadIdx.x & 63);
#else
    return 0;
#endif
}
MDX_HD int lane_count() { return MDX_ON_DEVICE ? 64 : 1; }


// eight bytes at byte offset `at` of the n input bytes, zero beyond the end
MDX_HD uint64_t load8(const uint8_t *p, uint32_t n, uint32_t at) {
    if (__builtin_expect(at + 8u <= n, 1)) {
        // (a block's payload starts at an arbitrary byte of the file: an unaligned 8-byte load is one instruction)
        typedef uint64_t u64u __attribute__((aligned(1)));
        return *(const u64u *)(p + at);
    }
    uint64_t w = 0;
    for (uint32_t j = 0; j < 8u; j++) if (at + j < n) w |= (uint64_t)p[at + j] << (8 * j);
    return w;
}
struct Held { uint32_t lo, hi, nlo, nhi; };
#if MDX_ON_DEVICE
MDX_HD Held fetch(const uint8_t *p, uint32_t n, uint32_t stretch) {
    const uint64_t w = load8(p, n, stretch + 8u * (uint32_t)lane_id()), x = load8(p, n, stretch + 512u);
    Held h;
    h.lo = (uint32_t)w; h.hi = (uint32_t)(w >> 32); h.nlo = (uint32_t)x; h.nhi = (uint32_t)(x >> 32);
    return h;
}
#endif

// Compressed input, consumed as a bit stream (LSB first).  Device: 512 bytes at a time are held by the lanes
// (8 each, plus the 8 bytes behind them in every lane) and handed out by readlane; host: straight from memory.
struct BitIn {
    const uint8_t *p;
    uint32_t n;          // bytes of input
    uint32_t pos;        // next byte to move into the bit buffer
    uint64_t bits;
    int nbits;
#if MDX_ON_DEVICE
    uint32_t held_lo, held_hi;   // this lane's eight bytes of the 512-byte stretch starting at `held_at`
    uint32_t next_lo, next_hi;   // the eight bytes behind the stretch (the same in every lane)
    uint32_t held_at;
#endif
    MDX_HD void init(const uint8_t *src, uint32_t len) {
        p = src; n = len; pos = 0; bits = 0; nbits = 0;
#if MDX_ON_DEVICE
        held_at = 0xFFFFFFFFu; held_lo = held_hi = next_lo = next_hi = 0;
#endif
    }
#if MDX_ON_DEVICE
    // (one call every 512 bytes of input: kept out of line — by value, so that the reader's state stays in
    //  registers — and the decoder's loops stay small)
    MDX_HD void reload(uint32_t at) {
        held_at = at & ~511u;
        const Held h = fetch(p, n, held_at);
        held_lo = h.lo; held_hi = h.hi; next_lo = h.nlo; next_hi = h.nhi;
    }
#endif
    // the eight bytes at byte offset `at` (at % 8 == 0) of the stretch held, or right behind it
    MDX_HD uint64_t word_at(uint32_t at) {
#if MDX_ON_DEVICE
        const uint32_t k = (at - held_at) >> 3;                  // 0 .. 64
        const int src = (int)(k & 63u);
        uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)held_lo, src);
        uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)held_hi, src);
        if (k == 64u) { lo = uni(next_lo); hi = uni(next_hi); }
        return (uint64_t)lo | ((uint64_t)hi << 32);
#else
        return load8(p, n, at);
#endif
    }
    // 48 .. 63 valid bits in `bits` afterwards (zeros beyond the end of the input): enough for a literal/length
    // code, its extra bits, a distance code and its extra bits (48) without another look
    MDX_HD void refill() {
        if (nbits >= 48) return;                  // (still enough for a whole symbol pair)
#if MDX_ON_DEVICE
        if (__builtin_expect((pos & ~511u) != held_at, 0)) reload(pos);
#endif
        const uint32_t at = pos & ~7u, sh = (pos & 7u) * 8u;
        uint64_t w = word_at(at) >> sh;
        if (sh) w |= word_at(at + 8u) << (64u - sh);
        bits |= w << nbits;                       // (bits beyond nbits + 8 k are the input's next ones: or-ed in again later)
        pos += (uint32_t)(63 - nbits) >> 3;
        nbits |= 56;
    }
    MDX_HD uint32_t peek(int k) const { return (uint32_t)(bits & ((1ull << k) - 1ull)); }
    MDX_HD void drop(int k) { bits >>= k; nbits -= k; }
    MDX_HD uint32_t take(int k) { const uint32_t v = peek(k); drop(k); return v; }
    // bytes consumed so far (whole bytes still in the buffer are given back)
    MDX_HD uint32_t consumed() const { return pos - (uint32_t)(nbits >> 3); }
    MDX_HD bool overrun() const { return consumed() > n; }
};

MDX_HD uint32_t bitrev(uint32_t v, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; i++) r |= ((v >> i) & 1u) << (len - 1 - i);
    return r;
}

// Canonical Huffman tables from code lengths (RFC 1951 3.2.2).  (Inlined at its five call sites like everything
// else here: out of line the kernel is a third of the size and a fifth slower — a call pins the decoder's registers.)  false: over-subscribed or incomplete set
// (a single code of length 1 is accepted, as zlib does for distance codes).
MDX_HD bool build(const uint8_t *lens, int n, uint16_t *count, uint16_t *sym, uint16_t *fast, int fast_bits, bool may_be_empty = false) {
    const int lane = lane_id(), nl = lane_count();
    for (int i = lane; i < 16; i += nl) count[i] = 0;
    for (int i = lane; i < (1 << fast_bits); i += nl) fast[i] = 0;
#if MDX_ON_DEVICE
    __builtin_amdgcn_wave_barrier();
#endif
    // (serial parts: every lane computes the same values; only lane 0 stores)
    uint16_t cnt[16];
    for (int i = 0; i < 16; i++) cnt[i] = 0;
    for (int i = 0; i < n; i++) cnt[uni(lens[i])]++;
    if (cnt[0] == n) {                            // no codes at all: fine for distances (a block of literals only)
        if (lane == 0) for (int i = 0; i < 16; i++) count[i] = 0;
        return may_be_empty;
    }
    int left = 1;
    for (int len = 1; len < 16; len++) {
        left <<= 1;
        left -= cnt[len];
        if (left < 0) return false;              // over-subscribed
    }
    if (left > 0 && !(n - cnt[0] == 1 && cnt[1] == 1)) return false;   // incomplete
    uint16_t offs[16];
    offs[1] = 0;
    for (int len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + cnt[len]);
    if (lane == 0) {
        for (int i = 0; i < 16; i++) count[i] = cnt[i];
        for (int s = 0; s < n; s++) { const uint32_t l = uni(lens[s]); if (l) sym[offs[l]++] = (uint16_t)s; }
    }
#if MDX_ON_DEVICE
    __builtin_amdgcn_wave_barrier();
#endif
    // fast table: every index whose low `len` bits are the (bit-reversed) code
    uint32_t code = 0;
    int first_index = 0;
    for (int len = 1; len <= fast_bits; len++) {
        for (int k = 0; k < cnt[len]; k++) {
            const uint32_t rev = bitrev(code + (uint32_t)k, len);
            const uint16_t entry = (uint16_t)((uni(sym[first_index + k]) << 4) | (uint32_t)len);
            for (uint32_t idx = rev + ((uint32_t)lane << len); idx < (1u << fast_bits); idx += (uint32_t)nl << len) fast[idx] = entry;
        }
        code = (code + cnt[len]) << 1;
        first_index += cnt[len];
    }
#if MDX_ON_DEVICE
    __builtin_amdgcn_wave_barrier();
#endif
    return true;
}

// one symbol; -1: invalid code
// The numbers of codes of each length, read once per block into registers (on the device: scalar registers —
// the walk below then costs a few scalar instructions per bit instead of an LDS round trip per bit, which is what
// a distance code longer than the fast table's eight bits used to cost: two thirds of the whole kernel's time)
// (first / index: the state of the canonical walk behind the fast table's `fast_bits` lengths — a code the table does not
// hold is longer than that, so the walk starts there instead of at length 1: eight of its steps saved for every long
// distance code, eleven for a literal / length one)
struct Counts { uint32_t c[16]; int first, index; };
MDX_HD Counts counts_of(const uint16_t *count, int fast_bits) {
    Counts k;
#pragma unroll
    for (int i = 0; i < 16; i++) k.c[i] = uni(count[i]);
    int first = 0, index = 0;
#pragma unroll
    for (int len = 1; len <= 15; len++)
        if (len <= fast_bits) { const int c = (int)k.c[len]; index += c; first += c; first <<= 1; }
    k.first = first; k.index = index;
    return k;
}

// (host builds of tools/experiments/inflate_stats.cpp count what the decoder meets)
#ifndef MDX_INFLATE_STAT
#define MDX_INFLATE_STAT(what, n) do { } while (0)
#endif
// one symbol; -1: invalid code
MDX_HD int decode(BitIn &in, const Counts &k, const uint16_t *sym, const uint16_t *fast, int fast_bits) {
    const uint32_t e = uni(fast[in.peek(fast_bits)]);
    if (__builtin_expect(e != 0u, 1)) { in.drop((int)(e & 15u)); return (int)(e >> 4); }
    MDX_INFLATE_STAT(fast_bits == FAST_D ? 6 : 7, 1);
    // canonical walk, one bit at a time, from the first length the fast table does not hold (the code so far: the
    // stream's first fast_bits bits, which arrive LSB first, reversed)
#if MDX_ON_DEVICE
    int code = (int)(__builtin_bitreverse32(in.peek(fast_bits)) >> (32 - fast_bits)) << 1, first = k.first, index = k.index;
#else
    int code = (int)bitrev(in.peek(fast_bits), fast_bits) << 1, first = k.first, index = k.index;
#endif
    uint64_t b = in.bits >> fast_bits;
#pragma unroll
    for (int len = 1; len <= 15; len++) {
        if (len <= fast_bits) continue;
        code |= (int)(b & 1); b >>= 1;
        const int c = (int)k.c[len];
        MDX_INFLATE_STAT(8, 1);
        if (code - c < first) { in.drop(len); return (int)uni(sym[index + (code - first)]); }
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
}

// The window the LDS holds and the stretch written out at a time.  RFC 1951 allows distances up to 32 KiB, but a
// 32 KiB window per block is what kept the kernel at one wavefront per SIMD, and the decoder is bound by
// instruction issue: the LDS holds the last MDX_RING bytes only, and a match that reaches further back reads its
// source from the output already in memory (flushed at least half a ring ago).  Measured per 4 M BAM records:
// 32 KiB 56 ms (4 blocks per CU), 16 KiB 39, 8 KiB 30, 4 KiB 24.4 (16 per CU: the register file's limit), 2 KiB 24.7.
#ifndef MDX_RING
#define MDX_RING 4096
#endif
enum { RING = MDX_RING, SEG = MDX_RING / 2 };
static_assert(RING >= 2048 && RING <= 32768 && (RING & (RING - 1)) == 0, "the ring holds a stretch being flushed plus one step of the decoder (258 + 8 bytes)");

// bytes [from, to) of the output, which the ring still holds, to `dst`
MDX_HD void flush(const uint8_t *win, uint8_t *dst, uint32_t from, uint32_t to) {
#if MDX_ON_DEVICE
    __builtin_amdgcn_wave_barrier();
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4 __attribute__((aligned(1))) u32x4_u;
    for (uint32_t o = from + 16u * (uint32_t)lane_id(); o < to; o += 1024u) {
        if (o + 16u <= to) {
            // (from is a multiple of SEG: the 16 bytes do not wrap around the ring; LDS reads of 4 aligned bytes)
            const uint32_t *w = (const uint32_t *)(win + (o & (RING - 1)));
            const u32x4 v = {w[0], w[1], w[2], w[3]};
            *(u32x4_u *)(dst + o) = v;
        } else {
            for (uint32_t j = o; j < to; j++) dst[j] = win[j & (RING - 1)];
        }
    }
    __builtin_amdgcn_wave_barrier();
#else
    for (uint32_t j = from; j < to; j++) dst[j] = win[j & (RING - 1)];
#endif
}

// Inflate one raw DEFLATE stream of `in_len` bytes to `dst` (capacity `cap` <= 65536 bytes).  `win`: RING bytes —
// on the device in the LDS, so that a match reads what the lanes have just written; the output leaves it SEG bytes
// at a time.  Returns the number of bytes produced, or a negative code: -1 corrupt stream, -2 output beyond `cap`,
// -3 input exhausted (dst then holds a part of the output).
MDX_HD int inflate_block(const uint8_t *src, uint32_t in_len, uint8_t *win, uint8_t *dst, uint32_t cap, Tables &t) {
    static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const int lane = lane_id(), nl = lane_count();
    BitIn in;
    in.init(src, in_len);
    uint32_t out = 0, flushed = 0;           // bytes produced; bytes already written to dst (a multiple of SEG)
    for (;;) {
        in.refill();
        const uint32_t last = in.take(1), type = in.take(2);
        if (type == 0) {
            in.drop(in.nbits & 7);                       // to the next byte boundary
            in.refill();
            const uint32_t len = in.take(16);
            in.refill();
            const uint32_t nlen = in.take(16);
            if ((len ^ nlen) != 0xFFFFu) return -1;
            const uint32_t at = in.consumed();
            if (at + len > in_len) return -3;
            if (out + len > cap) return -2;
            // (a stored stretch can be longer than the ring: out to dst directly, and into the ring for later matches;
            //  what the ring alone holds so far goes out first)
            if (out > flushed) flush(win, dst, flushed, out);
            for (uint32_t i = (uint32_t)lane; i < len; i += (uint32_t)nl) { const uint8_t v = src[at + i]; win[(out + i) & (RING - 1)] = v; dst[out + i] = v; }
            flushed = (out + len) & ~(uint32_t)(SEG - 1);
            out += len;
            in.pos = at + len; in.bits = 0; in.nbits = 0;
        } else if (type == 1 || type == 2) {
            if (type == 1) {
                for (int i = lane; i < 288; i += nl) t.lens[i] = (uint8_t)(i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8)));
                for (int i = lane; i < 32; i += nl) t.lens[288 + i] = 5;     // (30 and 31 never occur in a valid stream)
#if MDX_ON_DEVICE
                __builtin_amdgcn_wave_barrier();
#endif
                if (!build(t.lens, 288, t.count_ll, t.sym_ll, t.fast_ll, FAST_LL)) return -1;
                if (!build(t.lens + 288, 32, t.count_d, t.sym_d, t.fast_d, FAST_D)) return -1;
            } else {
                in.refill();
                const int nlen = (int)in.take(5) + 257, ndist = (int)in.take(5) + 1, ncode = (int)in.take(4) + 4;
                if (nlen > 286 || ndist > 30) return -1;
                for (int i = lane; i < 19; i += nl) t.lens[i] = 0;
#if MDX_ON_DEVICE
                __builtin_amdgcn_wave_barrier();
#endif
                for (int i = 0; i < ncode; i++) {
                    in.refill();
                    const uint32_t v = in.take(3);
                    if (lane == 0) t.lens[ORDER[i]] = (uint8_t)v;
                }
#if MDX_ON_DEVICE
                __builtin_amdgcn_wave_barrier();
#endif
                // the code-length code shares the distance tables' storage until those are built
                if (!build(t.lens, 19, t.count_d, t.sym_d, t.fast_d, 7)) return -1;
#if MDX_ON_DEVICE
                __builtin_amdgcn_wave_barrier();
#endif
                const Counts kcl = counts_of(t.count_d, 7);
                int i = 0;
                while (i < nlen + ndist) {
                    in.refill();
                    int s = decode(in, kcl, t.sym_d, t.fast_d, 7);
                    if (s < 0) return -1;
                    if (s < 16) { if (lane == 0) t.lens[32 + i] = (uint8_t)s; i++; continue; }
                    int prev = 0, rep;
                    if (s == 16) {
                        if (i == 0) return -1;
#if MDX_ON_DEVICE
                        __builtin_amdgcn_wave_barrier();
#endif
                        prev = (int)uni(t.lens[32 + i - 1]);
                        rep = 3 + (int)in.take(2);
                    } else if (s == 17) rep = 3 + (int)in.take(3);
                    else rep = 11 + (int)in.take(7);
                    if (i + rep > nlen + ndist) return -1;
                    for (int k = lane; k < rep; k += nl) t.lens[32 + i + k] = (uint8_t)prev;
                    i += rep;
                }
#if MDX_ON_DEVICE
                __builtin_amdgcn_wave_barrier();
#endif
                if (t.lens[32 + 256] == 0) return -1;    // no end-of-block code
                if (!build(t.lens + 32, nlen, t.count_ll, t.sym_ll, t.fast_ll, FAST_LL)) return -1;
                if (!build(t.lens + 32 + nlen, ndist, t.count_d, t.sym_d, t.fast_d, FAST_D, true)) return -1;
            }
#if MDX_ON_DEVICE
            __builtin_amdgcn_wave_barrier();
#endif
            const Counts kll = counts_of(t.count_ll, FAST_LL), kd = counts_of(t.count_d, FAST_D);
            for (;;) {
                // the one place a finished stretch leaves the ring (a step of the loop adds at most 258 bytes: the
                // stretch is still whole in the ring, and still older than anything a far match may ask for)
                if (__builtin_expect(out - flushed >= SEG, 0)) { flush(win, dst, flushed, flushed + SEG); flushed += SEG; }
                in.refill();               // 56 bits: the whole symbol pair below needs at most 48
                int s;
#if MDX_ON_DEVICE
                {
                    // Lane L looks up the symbol that would begin at bit L of the buffer — one LDS access for all 64
                    // candidates.  The symbols that really follow one another are then found on the VECTOR unit (round 5:
                    // the decoder's state is wavefront-uniform and lives on the scalar unit, which four wavefronts share and
                    // which the counters show busy all the time — 15 scalar instructions per output byte against 2 vector
                    // ones; the loop that picked the literals out of the lanes with readlane was a sixth of them):
                    // g1[L] = where the symbol behind a short-coded literal at bit L begins (L itself where there is none:
                    // the chain stops), g2 = g1 o g1, g4 = g2 o g2 by cross-lane permutes, and lane i walks to the bit its
                    // i-th symbol begins at by the binary digits of i.  A run of up to seven literals leaves in one step.
                    const uint32_t e = t.fast_ll[(uint32_t)(in.bits >> lane) & ((1u << FAST_LL) - 1u)];
                    auto perm = [](const uint32_t from, const uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(from << 2), (int)v); };
                    const uint32_t lim = (uint32_t)in.nbits;
                    const bool lit_here = e != 0u && (e >> 4) < 256u && (uint32_t)lane + FAST_LL <= lim;
                    const uint32_t g1 = lit_here ? (uint32_t)lane + (e & 15u) : (uint32_t)lane;
                    const uint32_t g2 = perm(g1, g1), g4 = perm(g2, g2);
                    uint32_t at = 0u;
                    { const uint32_t q = perm(at, g1); at = (lane & 1) ? q : at; }
                    { const uint32_t q = perm(at, g2); at = (lane & 2) ? q : at; }
                    { const uint32_t q = perm(at, g4); at = (lane & 4) ? q : at; }
                    const uint32_t ei = perm(at, e);
                    const bool budget = at + FAST_LL <= lim;
                    const bool lit_i = ei != 0u && (ei >> 4) < 256u && budget;
                    // (the literals of the run are the lanes in front of the first one that holds none)
                    const uint32_t nlit = (uint32_t)__builtin_popcountll(__ballot(lit_i && lane < 7));
                    const int pbit = __builtin_amdgcn_readlane((int)at, (int)nlit);
                    const uint32_t ep = (uint32_t)__builtin_amdgcn_readlane((int)(budget ? ei : 0u), (int)nlit);   // what follows the run
                    if (nlit) {
                        if (out + nlit > cap) return -2;
                        if ((uint32_t)lane < nlit) win[(out + (uint32_t)lane) & (RING - 1)] = (uint8_t)(ei >> 4);
                        out += nlit;
                        in.drop(pbit);
                        // the match behind the run in the same step when the buffer still holds all of it (its length
                        // code is known already: what follows needs up to 5 + 15 + 13 bits)
                        if (!(ep && (ep >> 4) > 256u && in.nbits >= (int)(ep & 15u) + 33)) continue;
                    }
                    // the symbol at bit 0 is no literal with a short code
                    if (ep) { s = (int)(ep >> 4); in.drop((int)(ep & 15u)); }
                    else s = decode(in, kll, t.sym_ll, t.fast_ll, FAST_LL);
                }
#else
                s = decode(in, kll, t.sym_ll, t.fast_ll, FAST_LL);
#endif
                if (s < 0) return -1;
                if (s < 256) {
                    MDX_INFLATE_STAT(0, 1);
                    if (out >= cap) return -2;
                    win[out & (RING - 1)] = (uint8_t)s;  // (every lane stores the same byte: cheaper than masking 63 off)
                    out++;
                    continue;
                }
                if (s == 256) break;
                s -= 257;
                if (s >= 29) return -1;
                // length and distance codes -> base + extra bits (RFC 1951 3.2.5), computed: a table in memory costs a
                // round trip per look-up on the device, four of them in the middle of every match
                // (round 5: the same arithmetic on the vector unit, handed back through readfirstlane — thirty scalar
                // instructions fewer per match, the scalar unit being what the kernel is bound by — changed nothing:
                // 18.9 ms per 4 M records either way; what it saves it pays in the hand-over)
                const int lx = s < 8 ? 0 : (s == 28 ? 0 : (s >> 2) - 1);
                const uint32_t lb = s < 8 ? 3u + (uint32_t)s : (s == 28 ? 258u : ((4u + ((uint32_t)s & 3u)) << lx) + 3u);
                const uint32_t len = lb + in.take(lx);
                const int ds = decode(in, kd, t.sym_d, t.fast_d, FAST_D);
                if (ds < 0 || ds >= 30) return -1;
                const int dx = ds < 4 ? 0 : (ds >> 1) - 1;
                const uint32_t db = ds < 4 ? (uint32_t)ds + 1u : ((2u + ((uint32_t)ds & 1u)) << dx) + 1u;
                const uint32_t dist = db + in.take(dx);
                if (dist > out) return -1;
                if (out + len > cap) return -2;
                MDX_INFLATE_STAT(1, 1); MDX_INFLATE_STAT(2, len);
                if (dist > (uint32_t)RING) MDX_INFLATE_STAT(3, 1);
                else if (len <= 64u && dist >= len) MDX_INFLATE_STAT(4, 1);
                else if (dist < 64u) MDX_INFLATE_STAT(5, 1);
#if MDX_ON_DEVICE
                __builtin_amdgcn_wave_barrier();
                if (RING < 32768 && __builtin_expect(dist > (uint32_t)RING, 0)) {
                    // the source has left the LDS: it is in `dst` (flushed at least SEG bytes ago; loaded past the
                    // vector L1, which may hold an older copy of a line this wavefront has written since)
                    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): the flushes have arrived
                    for (uint32_t i = (uint32_t)lane; i < len; i += 64u)
                        win[(out + i) & (RING - 1)] = __hip_atomic_load(dst + (out - dist + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (len <= 64u && dist >= len) {
                    // the common case (a match is 15 bytes on average): one byte per lane, sources in front of the stretch
                    if ((uint32_t)lane < len) win[(out + (uint32_t)lane) & (RING - 1)] = win[(out - dist + (uint32_t)lane) & (RING - 1)];
                } else if (dist >= 64u) {
                    // 64 bytes at a time: their sources lie in front of the stretch being written
                    for (uint32_t i = (uint32_t)lane; i < len; i += 64u) {
                        win[(out + i) & (RING - 1)] = win[(out - dist + i) & (RING - 1)];
                        __builtin_amdgcn_wave_barrier();
                    }
                } else {
                    // a short period: byte i of the match is byte (i mod dist) of the `dist` bytes in front of it
                    for (uint32_t i = (uint32_t)lane; i < len; i += 64u) win[(out + i) & (RING - 1)] = win[(out - dist + (i % dist)) & (RING - 1)];
                }
                __builtin_amdgcn_wave_barrier();
#else
                if (RING < 32768 && dist > (uint32_t)RING) for (uint32_t i = 0; i < len; i++) win[(out + i) & (RING - 1)] = dst[out - dist + i];
                else for (uint32_t i = 0; i < len; i++) win[(out + i) & (RING - 1)] = win[(out - dist + i) & (RING - 1)];
#endif
                out += len;
                // (input that ends early decodes as zeros: caught at the end of the block, the output is bounded by cap)
            }
        } else {
            return -1;
        }
        if (in.overrun()) return -3;
        if (last) break;
    }
    if (out > flushed) flush(win, dst, flushed, out);
    return (int)out;
}

}  // namespace mdx_inflate
