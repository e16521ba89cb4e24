// fast_inflate.h -- a one-shot zlib-stream decoder for the PNG reader of the batched runner (System.cpp).
//
// Why: `run_kitti_stereo` on PNG files -- the reference's input format (src/System.cpp:77-85, cv::imread) -- is
// inflate-bound: on 4 541 KITTI-size frames the GPU waits for the decoder threads, and zlib
// 1.2.11's inflate() was 60 % of a frame's decode.  This decoder does what a streaming inflate cannot: the whole
// compressed stream and the whole output are in memory, so it keeps a 64-bit bit buffer refilled with one unaligned
// load, decodes up to three literals per refill from an 11-bit table, and copies matches eight bytes at a time into
// an output buffer with slack.  RFC 1950 / 1951 from the text; no code from zlib or any other inflate.
//
// Contract: finf::inflate_zlib() returns true ONLY when the stream decoded to exactly `out_size` bytes, ended where
// the input ended and its Adler-32 trailer matches the output (computed here).  On anything else -- including streams
// it merely does not like -- it returns false and the caller falls back to zlib, which stays the judge of what is a
// corrupt file.  The output buffer must have kSlack writable bytes behind out_size.
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace lzb_vio {
namespace finf {

constexpr int kSlack = 320;                         // three literals, or a match's last 16-byte store, may land past a full buffer before it is refused
constexpr int kLitRoot = 11, kDistRoot = 8;
constexpr int kLitCap = 2048 + 2048, kDistCap = 256 + 1024;
constexpr uint32_t OP_LIT = 0, OP_BASE = 16, OP_EOB = 64, OP_PTR = 96, OP_BAD = 128;

struct Tables {
    uint32_t lit[kLitCap];
    uint32_t dist[kDistCap];
};

static inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline void store64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }
static inline uint32_t entry(uint32_t value, uint32_t op, uint32_t nbits) { return (value << 16) | (op << 8) | nbits; }

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline uint32_t bit_reverse(uint32_t code, int len)
{
    uint32_t r = 0;
    for (int i = 0; i < len; i++) { r = (r << 1) | (code & 1); code >>= 1; }
    return r;
}

// what symbol `sym` of a literal / length (dist = false) or distance alphabet decodes to, with `nbits` still to consume
static inline uint32_t symbol_entry(int sym, bool dist, uint32_t nbits)
{
    if (dist) return sym < 30 ? entry(kDistBase[sym], OP_BASE + kDistExtra[sym], nbits) : entry(0, OP_BAD, nbits);
    if (sym < 256) return entry((uint32_t)sym, OP_LIT, nbits);
    if (sym == 256) return entry(0, OP_EOB, nbits);
    if (sym < 286) return entry(kLenBase[sym - 257], OP_BASE + kLenExtra[sym - 257], nbits);
    return entry(0, OP_BAD, nbits);
}

// Canonical Huffman code of `n` symbols with lengths lens[] (0 = unused) -> decode table: `root` index bits, longer codes
// through one level of subtables.  false: over-subscribed code or a table that would not fit `cap`.
static inline bool build_table(const uint8_t *lens, int n, uint32_t *table, int root, int cap, bool dist)
{
    int count[16] = {0};
    for (int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    int left = 1;
    for (int len = 1; len <= 15; len++) {
        left = (left << 1) - count[len];
        if (left < 0) return false;
    }
    uint32_t first[16];
    {
        uint32_t code = 0;
        for (int len = 1; len <= 15; len++) { code = (code + (uint32_t)count[len - 1]) << 1; first[len] = code; }
    }
    const int nroot = 1 << root;
    for (int i = 0; i < nroot; i++) table[i] = entry(0, OP_BAD, 1);
    uint8_t submax[1 << kLitRoot];
    memset(submax, 0, (size_t)nroot);
    uint32_t next[16];
    memcpy(next, first, sizeof(next));
    bool any_long = false;
    for (int sym = 0; sym < n; sym++) {
        const int len = lens[sym];
        if (!len) continue;
        const uint32_t rev = bit_reverse(next[len]++, len);
        if (len <= root) {
            const uint32_t e = symbol_entry(sym, dist, (uint32_t)len);
            for (uint32_t i = rev; i < (uint32_t)nroot; i += 1u << len) table[i] = e;
        } else {
            uint8_t &m = submax[rev & (uint32_t)(nroot - 1)];
            if (len - root > m) m = (uint8_t)(len - root);
            any_long = true;
        }
    }
    if (!any_long) return true;
    int off = nroot;
    for (int p = 0; p < nroot; p++) {
        if (!submax[p]) continue;
        const int size = 1 << submax[p];
        if (off + size > cap) return false;
        table[p] = entry((uint32_t)off, OP_PTR, submax[p]);
        for (int i = 0; i < size; i++) table[off + i] = entry(0, OP_BAD, 1);
        off += size;
    }
    memcpy(next, first, sizeof(next));
    for (int sym = 0; sym < n; sym++) {
        const int len = lens[sym];
        if (!len) continue;
        const uint32_t rev = bit_reverse(next[len]++, len);
        if (len <= root) continue;
        const uint32_t p = rev & (uint32_t)(nroot - 1);
        const uint32_t base = table[p] >> 16, sub_bits = submax[p];
        const uint32_t e = symbol_entry(sym, dist, (uint32_t)(len - root));
        for (uint32_t i = rev >> root; i < (1u << sub_bits); i += 1u << (len - root)) table[base + i] = e;
    }
    return true;
}

// Adler-32 (RFC 1950) of `n` bytes.  SSE2: sixteen bytes a step -- byte sums by psadbw, position-weighted sums by pmaddwd on
// the bytes widened to 16 bits --, the modulo deferred over blocks of at most 5 552 bytes (the sums stay inside 32 bits).
static inline uint32_t adler32(const uint8_t *p, size_t n)
{
    uint32_t a = 1, b = 0;
#if defined(__SSE2__)
    const __m128i zero = _mm_setzero_si128();
    const __m128i w_lo = _mm_set_epi16(9, 10, 11, 12, 13, 14, 15, 16), w_hi = _mm_set_epi16(1, 2, 3, 4, 5, 6, 7, 8);
    while (n >= 16) {
        size_t blocks = (n < 5552 ? n : 5552) / 16;
        n -= blocks * 16;
        __m128i v_ps = _mm_set_epi32(0, 0, 0, (int)(a * (uint32_t)blocks));      // sum over the blocks of `a` before the block
        __m128i v_b = _mm_set_epi32(0, 0, 0, (int)b), v_a = zero;
        do {
            const __m128i x = _mm_loadu_si128((const __m128i *)p);
            p += 16;
            v_ps = _mm_add_epi32(v_ps, v_a);
            v_a = _mm_add_epi32(v_a, _mm_sad_epu8(x, zero));
            v_b = _mm_add_epi32(v_b, _mm_add_epi32(_mm_madd_epi16(_mm_unpacklo_epi8(x, zero), w_lo), _mm_madd_epi16(_mm_unpackhi_epi8(x, zero), w_hi)));
        } while (--blocks);
        v_b = _mm_add_epi32(v_b, _mm_slli_epi32(v_ps, 4));
        // horizontal sums
        v_a = _mm_add_epi32(v_a, _mm_shuffle_epi32(v_a, _MM_SHUFFLE(1, 0, 3, 2)));
        a += (uint32_t)_mm_cvtsi128_si32(v_a);
        v_b = _mm_add_epi32(v_b, _mm_shuffle_epi32(v_b, _MM_SHUFFLE(1, 0, 3, 2)));
        v_b = _mm_add_epi32(v_b, _mm_shuffle_epi32(v_b, _MM_SHUFFLE(2, 3, 0, 1)));
        b = (uint32_t)_mm_cvtsi128_si32(v_b);
        a %= 65521; b %= 65521;
    }
#endif
    while (n) {
        const size_t m = n < 5552 ? n : 5552;
        for (size_t k = 0; k < m; k++) { a += p[k]; b += a; }
        a %= 65521; b %= 65521;
        p += m; n -= m;
    }
    return (b << 16) | a;
}

// `in`: the zlib stream, readable for in_size + 16 bytes (the caller pads); `out`: out_size + kSlack writable bytes.
static inline bool inflate_zlib(const uint8_t *in, size_t in_size, uint8_t *out, size_t out_size, Tables &T)
{
    if (in_size < 2 + 4) return false;
    if ((in[0] & 0x0F) != 8 || (in[0] >> 4) > 7 || ((in[0] << 8) | in[1]) % 31 != 0 || (in[1] & 0x20)) return false;
    const uint8_t *const in_begin = in, *const in_end = in + in_size - 4;        // the Adler-32 trailer is not deflate data
    const uint8_t *ip = in + 2;
    uint8_t *op = out, *const out_end = out + out_size;
    uint64_t bitbuf = 0;
    unsigned bitcnt = 0;
    // a load may run up to 16 bytes past in_end: the trailer + the caller's padding; bits from there are refused at the end
#define FINF_REFILL() do { bitbuf |= load64(ip) << bitcnt; ip += (63 - bitcnt) >> 3; bitcnt |= 56; } while (0)
#define FINF_BITS(n) ((uint32_t)(bitbuf & ((1ull << (n)) - 1)))
#define FINF_DROP(n) do { bitbuf >>= (n); bitcnt -= (unsigned)(n); } while (0)
    bool last = false;
    while (!last) {
        if (ip > in_end + 8) return false;
        FINF_REFILL();
        last = FINF_BITS(1); FINF_DROP(1);
        const uint32_t type = FINF_BITS(2); FINF_DROP(2);
        if (type == 0) {
            FINF_DROP(bitcnt & 7);
            ip -= bitcnt >> 3;                       // the whole bytes still in the buffer go back
            bitbuf = 0; bitcnt = 0;
            if (ip + 4 > in_end) return false;
            const uint32_t len = ip[0] | (ip[1] << 8), nlen = ip[2] | (ip[3] << 8);
            ip += 4;
            if ((len ^ 0xFFFFu) != nlen || ip + len > in_end || op + len > out_end) return false;
            memcpy(op, ip, len);
            ip += len; op += len;
            continue;
        }
        if (type == 3) return false;
        if (type == 1) {
            uint8_t lens[288 + 32];
            for (int i = 0; i < 144; i++) lens[i] = 8;
            for (int i = 144; i < 256; i++) lens[i] = 9;
            for (int i = 256; i < 280; i++) lens[i] = 7;
            for (int i = 280; i < 288; i++) lens[i] = 8;
            for (int i = 0; i < 32; i++) lens[288 + i] = 5;
            if (!build_table(lens, 288, T.lit, kLitRoot, kLitCap, false) || !build_table(lens + 288, 32, T.dist, kDistRoot, kDistCap, true)) return false;
        } else {
            const int hlit = (int)FINF_BITS(5) + 257; FINF_DROP(5);
            const int hdist = (int)FINF_BITS(5) + 1; FINF_DROP(5);
            const int hclen = (int)FINF_BITS(4) + 4; FINF_DROP(4);
            if (hlit > 286 || hdist > 30) return false;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t pre_lens[19] = {0};
            for (int i = 0; i < hclen; i++) {
                if (bitcnt < 3) FINF_REFILL();
                pre_lens[order[i]] = (uint8_t)FINF_BITS(3); FINF_DROP(3);
            }
            uint32_t pre[128];
            {
                // the code-length code: at most 7 bits, a direct table of (symbol << 16 | length)
                int count[8] = {0};
                for (int i = 0; i < 19; i++) count[pre_lens[i]]++;
                count[0] = 0;
                int left = 1;
                for (int len = 1; len <= 7; len++) { left = (left << 1) - count[len]; if (left < 0) return false; }
                uint32_t next[8], code = 0;
                for (int len = 1; len <= 7; len++) { code = (code + (uint32_t)count[len - 1]) << 1; next[len] = code; }
                for (int i = 0; i < 128; i++) pre[i] = 0;
                for (int sym = 0; sym < 19; sym++) {
                    const int len = pre_lens[sym];
                    if (!len) continue;
                    const uint32_t rev = bit_reverse(next[len]++, len);
                    for (uint32_t i = rev; i < 128; i += 1u << len) pre[i] = ((uint32_t)sym << 16) | (uint32_t)len;
                }
            }
            uint8_t lens[286 + 30 + 140];
            int i = 0;
            const int total = hlit + hdist;
            while (i < total) {
                if (ip > in_end + 8) return false;
                FINF_REFILL();
                const uint32_t e = pre[FINF_BITS(7)];
                if (!e) return false;
                FINF_DROP(e & 0xFF);
                const int sym = (int)(e >> 16);
                if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                int rep;
                uint8_t v = 0;
                if (sym == 16) {
                    if (i == 0) return false;
                    v = lens[i - 1]; rep = 3 + (int)FINF_BITS(2); FINF_DROP(2);
                } else if (sym == 17) { rep = 3 + (int)FINF_BITS(3); FINF_DROP(3); }
                else { rep = 11 + (int)FINF_BITS(7); FINF_DROP(7); }
                if (i + rep > total) return false;
                memset(lens + i, v, (size_t)rep);
                i += rep;
            }
            if (lens[256] == 0) return false;
            if (!build_table(lens, hlit, T.lit, kLitRoot, kLitCap, false) || !build_table(lens + hlit, hdist, T.dist, kDistRoot, kDistCap, true)) return false;
        }
        // ---- the symbols of the block.  `e` is always the table entry for the bits at the head of the buffer, looked up as
        // early as possible: a refill only adds bits ABOVE the ones it holds, so the entry stays valid across it, and the
        // lookup for the next symbol is issued before a literal is stored or a match is copied (its latency hides there).
        const uint32_t *const lit = T.lit, *const dtab = T.dist;
        FINF_REFILL();
        uint32_t e = lit[FINF_BITS(kLitRoot)];
        for (;;) {
            if (ip > in_end + 8 || op > out_end) return false;
            uint32_t opc = (e >> 8) & 0xFF;
            if (opc == OP_LIT) {                     // up to three literals per refill: 3 x 11 of at least 56 bits
                FINF_DROP(e & 0xFF);
                uint8_t v = (uint8_t)(e >> 16);
                e = lit[FINF_BITS(kLitRoot)];
                *op++ = v;
                if (((e >> 8) & 0xFF) == OP_LIT) {
                    FINF_DROP(e & 0xFF);
                    v = (uint8_t)(e >> 16);
                    e = lit[FINF_BITS(kLitRoot)];
                    *op++ = v;
                    if (((e >> 8) & 0xFF) == OP_LIT) {
                        FINF_DROP(e & 0xFF);
                        v = (uint8_t)(e >> 16);
                        e = lit[FINF_BITS(kLitRoot)];
                        *op++ = v;
                    }
                }
                FINF_REFILL();
                continue;
            }
            // at least 56 bits here (every path to this point ends with a refill): pointer 11 + subtable 4 + length extra 5 +
            // distance 8 + 7 + 13 = 48
            if (opc == OP_PTR) {
                FINF_DROP(kLitRoot);
                e = lit[(e >> 16) + FINF_BITS(e & 0xFF)];
                opc = (e >> 8) & 0xFF;
                if (opc == OP_LIT) {
                    FINF_DROP(e & 0xFF);
                    *op++ = (uint8_t)(e >> 16);
                    FINF_REFILL();
                    e = lit[FINF_BITS(kLitRoot)];
                    continue;
                }
            }
            if (opc == OP_EOB) { FINF_DROP(e & 0xFF); break; }
            if (opc >= OP_EOB || opc < OP_BASE) return false;      // OP_BAD (or a pointer inside a subtable: never built)
            FINF_DROP(e & 0xFF);
            const uint32_t lx = opc - OP_BASE;
            const uint32_t len = (e >> 16) + FINF_BITS(lx);
            FINF_DROP(lx);
            uint32_t d = dtab[FINF_BITS(kDistRoot)];
            uint32_t dop = (d >> 8) & 0xFF;
            if (dop == OP_PTR) {
                FINF_DROP(kDistRoot);
                d = dtab[(d >> 16) + FINF_BITS(d & 0xFF)];
                dop = (d >> 8) & 0xFF;
            }
            if (dop < OP_BASE || dop >= OP_EOB) return false;
            FINF_DROP(d & 0xFF);
            const uint32_t dx = dop - OP_BASE;
            const uint32_t dist = (d >> 16) + FINF_BITS(dx);
            FINF_DROP(dx);
            FINF_REFILL();
            e = lit[FINF_BITS(kLitRoot)];            // the next symbol's entry, before the copy
            if (dist > (size_t)(op - out) || op + len > out_end) return false;
            const uint8_t *s = op - dist;
            uint8_t *t = op, *const te = op + len;
            if (dist >= 16) {
#if defined(__SSE2__)
                do { _mm_storeu_si128((__m128i *)t, _mm_loadu_si128((const __m128i *)s)); t += 16; s += 16; } while (t < te);
#else
                do { store64(t, load64(s)); store64(t + 8, load64(s + 8)); t += 16; s += 16; } while (t < te);
#endif
            } else if (dist >= 8) {
                do { store64(t, load64(s)); t += 8; s += 8; } while (t < te);
            } else if (dist == 1) {
                const uint64_t v = 0x0101010101010101ull * s[0];
                do { store64(t, v); t += 8; } while (t < te);
            } else {
                do { *t++ = *s++; } while (t < te);
            }
            op = te;
        }
    }
#undef FINF_REFILL
#undef FINF_BITS
#undef FINF_DROP
    if (op != out_end) return false;
    // whole bytes left in the bit buffer were never consumed: the deflate data must end exactly where the trailer starts
    const uint8_t *used = ip - (bitcnt >> 3);
    if (used != in_end) return false;
    const uint32_t want = ((uint32_t)in_end[0] << 24) | ((uint32_t)in_end[1] << 16) | ((uint32_t)in_end[2] << 8) | in_end[3];
    (void)in_begin;
    return adler32(out, out_size) == want;
}

}  // namespace finf
}  // namespace lzb_vio
