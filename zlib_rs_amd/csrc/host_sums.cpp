// host_sums.cpp -- see host_sums.h.
#include "host_sums.h"
#include <string.h>
#include <mutex>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {
const uint32_t kBase = 65521u, kNmax = 5552u, kPoly = 0xEDB88320u;
uint32_t g_tab[8][256];   // slicing-by-8: table k advances a byte that is followed by k more bytes
std::once_flag g_once;
void tab_init() {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? kPoly : 0u);
        g_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int k = 1; k < 8; ++k) g_tab[k][i] = g_tab[0][g_tab[k - 1][i] & 0xFFu] ^ (g_tab[k - 1][i] >> 8);
}
// state in, state out (no pre / post inversion)
uint32_t crc_table(uint32_t crc, const uint8_t* buf, size_t len) {
    std::call_once(g_once, tab_init);
    while (len >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, buf, 4);
        memcpy(&hi, buf + 4, 4);
        lo ^= crc;
        crc = g_tab[7][lo & 0xFFu] ^ g_tab[6][(lo >> 8) & 0xFFu] ^ g_tab[5][(lo >> 16) & 0xFFu] ^ g_tab[4][lo >> 24] ^
              g_tab[3][hi & 0xFFu] ^ g_tab[2][(hi >> 8) & 0xFFu] ^ g_tab[1][(hi >> 16) & 0xFFu] ^ g_tab[0][hi >> 24];
        buf += 8;
        len -= 8;
    }
    while (len--) crc = g_tab[0][(crc ^ *buf++) & 0xFFu] ^ (crc >> 8);
    return crc;
}

#if defined(__x86_64__)
// Attribution: the folding scheme, its constants and the register naming (x0..x8, y5..y8) follow the public PCLMULQDQ CRC-32
// routine of Chromium's zlib (contrib/optimizations crc32_simd.c, BSD-style licence, (c) The Chromium Authors), itself an
// implementation of Intel's white paper below -- NOT the reference's crc32/pclmulqdq.rs, which is structured differently.
// CRC-32 by carry-less multiplication ("Fast CRC Computation for Generic Polynomials Using PCLMULQDQ", Gopal et al.): four
// 128-bit lanes folded 64 bytes at a time, then to one lane, to 64 bits, Barrett reduction.  len >= 64, a multiple of 16.
// The constants are x^n mod P for the bit-reflected polynomial 0x1DB710641: n = 4*128+32, 4*128-32, 128+32, 128-32, 64.
__attribute__((target("pclmul,sse4.1"))) uint32_t crc_clmul(uint32_t crc, const uint8_t* buf, size_t len) {
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5k0 = _mm_set_epi64x(0x0000000000ll, 0x0163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = k1k2;
    buf += 64;
    len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
        x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
        x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
        y6 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
        y8 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64;
        len -= 64;
    }
    // four lanes -> one
    x0 = k3k4;
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128((const __m128i*)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16;
        len -= 16;
    }
    // 128 -> 64 bits
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = k5k0;
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    // Barrett reduction to 32 bits
    x0 = poly;
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

// Adler-32, 32 bytes a step: s1 by byte sums (psadbw), s2 = 32 * s1_before + sum of (32 - i) * byte_i (pmaddubsw) -- the
// arrangement of the reference's adler32/avx2.rs on 128-bit registers.  len a multiple of 32, <= 5536 (below NMAX: no overflow).
__attribute__((target("ssse3"))) void adler_ssse3(uint32_t& a, uint32_t& b, const uint8_t* buf, size_t len) {
    const __m128i tap1 = _mm_setr_epi8(32, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17);
    const __m128i tap2 = _mm_setr_epi8(16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1);
    const __m128i zero = _mm_setzero_si128(), ones = _mm_set1_epi16(1);
    __m128i v_s1 = _mm_setzero_si128();   // byte sums of this call (two 64-bit halves)
    __m128i v_ps = _mm_setzero_si128();   // sum over the steps of s1 in front of the step (without `a`), 32-bit lanes
    __m128i v_s2 = _mm_setzero_si128();   // weighted sums, 32-bit lanes
    const size_t steps = len / 32;
    for (size_t i = 0; i < steps; ++i) {
        const __m128i d1 = _mm_loadu_si128((const __m128i*)(buf + 32 * i)), d2 = _mm_loadu_si128((const __m128i*)(buf + 32 * i + 16));
        v_ps = _mm_add_epi32(v_ps, v_s1);
        v_s1 = _mm_add_epi32(v_s1, _mm_add_epi32(_mm_sad_epu8(d1, zero), _mm_sad_epu8(d2, zero)));
        v_s2 = _mm_add_epi32(v_s2, _mm_madd_epi16(_mm_maddubs_epi16(d1, tap1), ones));
        v_s2 = _mm_add_epi32(v_s2, _mm_madd_epi16(_mm_maddubs_epi16(d2, tap2), ones));
    }
    uint32_t s1[4], ps[4], s2[4];
    _mm_storeu_si128((__m128i*)s1, v_s1);
    _mm_storeu_si128((__m128i*)ps, v_ps);
    _mm_storeu_si128((__m128i*)s2, v_s2);
    const uint32_t sum1 = s1[0] + s1[2];                       // all bytes
    const uint32_t before = ps[0] + ps[2];                     // sum over steps of the bytes in front of the step
    const uint32_t weighted = s2[0] + s2[1] + s2[2] + s2[3];
    // b' = b + len * a + 32 * before + weighted;  a' = a + sum1      (len * a <= 5536 * 65520, 32 * before <= 32 * 173 * 5536 * 255 / 2: fits)
    b = (uint32_t)(((uint64_t)b + (uint64_t)len * a + 32ull * before + weighted) % kBase);
    a = (a + sum1) % kBase;
}
#endif
}   // namespace

uint32_t zmi_host_crc32(uint32_t crc, const uint8_t* buf, size_t len) {
    crc = ~crc;
#if defined(__x86_64__)
    static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (have && len >= 64) {
        const size_t n = len & ~(size_t)15;
        crc = crc_clmul(crc, buf, n);
        buf += n;
        len -= n;
    }
#endif
    return ~crc_table(crc, buf, len);
}

uint32_t zmi_host_adler32(uint32_t adler, const uint8_t* buf, size_t len) {
    uint32_t a = adler & 0xFFFFu, b = (adler >> 16) & 0xFFFFu;
#if defined(__x86_64__)
    static const bool have = __builtin_cpu_supports("ssse3");
    while (have && len >= 32) {
        size_t n = len < 5536u ? len & ~(size_t)31 : 5536u;
        adler_ssse3(a, b, buf, n);
        buf += n;
        len -= n;
    }
#endif
    while (len) {
        size_t k = len < kNmax ? len : kNmax;
        len -= k;
        while (k--) { a += *buf++; b += a; }
        a %= kBase; b %= kBase;
    }
    return (b << 16) | a;
}
