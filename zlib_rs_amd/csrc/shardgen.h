/* shardgen.h -- definition of the synthetic "Silesia-like" benchmark shards.
 *
 * A shard is a pure function of (seed, shard index): 64-byte lines, each line a pure function of
 * (seed, shard, line).  One HIP thread (or one C loop iteration) produces one line, so the GPU
 * generator (gen.hip) and the CPU twin (oracle/zoracle.c: zo_gen_shard) regenerate any shard
 * independently and bit-identically.  Integer arithmetic only.  Plain C so that both hipcc and
 * gcc compile it.
 *
 * Content class = shard index mod 8, approximating the Silesia mix:
 *   0,1,2  English-like text (Zipf word/phrase draws from a fixed 4096-word vocabulary)
 *   3      XML-like records with repeated tag names and decimal fields
 *   4      32-byte binary database records (monotone ids, small ints, padded ASCII)
 *   5      16-bit little-endian random-walk samples
 *   6      executable-like: skewed opcode bytes, addresses sharing high bytes, zero runs
 *   7      first half incompressible uniform bytes, second half class-0 text
 * The reference has no generator of its own for this workload (its bench uses silesia-small.tar,
 * absent from the mount: /root/reference/.MISSING_LARGE_BLOBS); the low-entropy LCG used by its
 * inflate tests (test-libz-rs-sys/src/inflate.rs:1981-1993) is restated in oracle/zoracle.c.
 */
#ifndef ZMI_SHARDGEN_H
#define ZMI_SHARDGEN_H
#include <stdint.h>

#ifndef ZMI_HD
#if defined(__HIPCC__)
#define ZMI_HD __host__ __device__ static inline
#else
#define ZMI_HD static inline
#endif
#endif

#define ZMI_GEN_SEED 0x5A4C4942ull /* "ZLIB" */
#define ZMI_GEN_LINE 64

ZMI_HD uint64_t zmi_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* Zipf(s~1) rank in [0, 2^levels - 1): octave uniform, rank uniform inside the octave */
ZMI_HD uint32_t zmi_zipf(uint64_t r, uint32_t levels) {
    uint32_t lvl = (uint32_t)(r % levels);
    uint32_t m = (1u << lvl) - 1u;
    return m + ((uint32_t)(r >> 16) & m);
}

/* spelling of vocabulary word w: writes up to 12 lowercase letters, returns length (2..12) */
ZMI_HD uint32_t zmi_word(uint32_t w, uint8_t* dst) {
    /* 32-entry skewed letter table (approximate English letter frequencies) */
    const char* tab = "eeeetttaaaooiinnsshhrrdlcumwfgyp";
    uint64_t h = zmi_mix64(0x574F5244ull * 0x10001ull + w);
    uint64_t h2 = zmi_mix64(h);
    uint32_t len = 2u + (uint32_t)(h2 % 7u) + (w > 32u ? 1u : 0u) + (w > 512u ? (uint32_t)((h2 >> 8) % 3u) : 0u);
    uint32_t i;
    for (i = 0; i < len; ++i) {
        dst[i] = (uint8_t)tab[(h >> (5u * i)) & 31u];
    }
    return len;
}

ZMI_HD void zmi_line_text(uint64_t key, uint8_t* out) {
    uint32_t pos = 0;
    uint64_t st = key;
    uint8_t wbuf[12];
    while (pos < 63u) {
        uint64_t r;
        uint32_t nwords, j, pid = 0;
        st += 0x9E3779B97F4A7C15ull;
        r = zmi_mix64(st);
        if ((r & 3u) == 0u) { /* 25 %: a 3-word phrase from a Zipf-ranked phrase book */
            pid = zmi_zipf(r >> 2, 11);
            nwords = 3;
        } else {
            nwords = 1;
        }
        for (j = 0; j < nwords && pos < 63u; ++j) {
            uint32_t w, len, k;
            if (nwords == 3) {
                w = zmi_zipf(zmi_mix64(0x50485241ull + pid * 4u + j), 12);
            } else {
                w = zmi_zipf(r >> 2, 12);
            }
            len = zmi_word(w, wbuf);
            for (k = 0; k < len && pos < 63u; ++k) out[pos++] = wbuf[k];
            if (pos < 63u) {
                uint32_t p = (uint32_t)(r >> (40 + 4 * j)) & 15u;
                if (p == 0u) { out[pos++] = ','; }
                else if (p == 1u && j == nwords - 1) { out[pos++] = '.'; }
                if (pos < 63u) out[pos++] = ' ';
            }
        }
    }
    out[63] = '\n';
}

ZMI_HD void zmi_put_dec(uint8_t* dst, uint32_t v, uint32_t digits) {
    uint32_t i;
    for (i = 0; i < digits; ++i) {
        dst[digits - 1 - i] = (uint8_t)('0' + v % 10u);
        v /= 10u;
    }
}

ZMI_HD void zmi_line_xml(uint64_t key, uint32_t shard, uint32_t line, uint8_t* out) {
    /* <row id="0000000" k="ab" v="0000.00" s="x"/> padded with spaces, newline terminated */
    const char* tmpl = "<row id=\"0000000\" cat=\"aaaa\" val=\"0000.00\" st=\"ok\" n=\"00\"/>     \n";
    const char* cats = "itemuserpagenodelinkfiletasknote";
    uint64_t r = zmi_mix64(key);
    uint32_t i, c;
    for (i = 0; i < 64; ++i) out[i] = (uint8_t)tmpl[i];
    zmi_put_dec(out + 9, (shard & 0xFFu) * 16384u + line, 7);
    c = zmi_zipf(r, 4) & 7u;
    for (i = 0; i < 4; ++i) out[23 + i] = (uint8_t)cats[c * 4u + i];
    zmi_put_dec(out + 34, (uint32_t)(r >> 20) % 10000u, 4);
    zmi_put_dec(out + 39, (uint32_t)(r >> 36) % 100u, 2);
    if (((r >> 44) & 7u) == 0u) { out[47] = 'n'; out[48] = 'o'; }
    zmi_put_dec(out + 54, zmi_zipf(r >> 48, 6), 2);
}

ZMI_HD void zmi_line_db(uint64_t key, uint32_t shard, uint32_t line, uint8_t* out) {
    /* two 32-byte records: u32 id, u32 timestamp (small delta), u16 qty, u16 flags, 12-byte name, 8 zero pad */
    const char* names = "alpha       bravo       charlie     delta       echo        foxtrot     golf        hotel       ";
    uint32_t k, i;
    for (k = 0; k < 2; ++k) {
        uint8_t* rec = out + 32u * k;
        uint64_t r = zmi_mix64(key + k);
        uint32_t id = (shard & 0xFFFu) * 32768u + line * 2u + k;
        uint32_t ts = 1700000000u + (line * 2u + k) * 3u + (uint32_t)(r & 3u);
        uint32_t qty = zmi_zipf(r >> 8, 8);
        uint32_t flags = (uint32_t)(r >> 24) & 0x0101u;
        uint32_t n = zmi_zipf(r >> 32, 3) & 7u;
        rec[0] = (uint8_t)id; rec[1] = (uint8_t)(id >> 8); rec[2] = (uint8_t)(id >> 16); rec[3] = (uint8_t)(id >> 24);
        rec[4] = (uint8_t)ts; rec[5] = (uint8_t)(ts >> 8); rec[6] = (uint8_t)(ts >> 16); rec[7] = (uint8_t)(ts >> 24);
        rec[8] = (uint8_t)qty; rec[9] = (uint8_t)(qty >> 8);
        rec[10] = (uint8_t)flags; rec[11] = (uint8_t)(flags >> 8);
        for (i = 0; i < 12; ++i) rec[12 + i] = (uint8_t)names[n * 12u + i];
        for (i = 24; i < 32; ++i) rec[i] = 0;
    }
}

ZMI_HD void zmi_line_walk(uint64_t key, uint32_t line, uint8_t* out) {
    /* 32 signed 16-bit samples: slow triangle carrier + per-line random walk with |delta| <= 64 */
    uint32_t tri = (line * 37u) & 0x3FFFu;
    int32_t v = (int32_t)(tri < 0x2000u ? tri : 0x3FFFu - tri) * 2 - 8192;
    uint64_t st = key;
    uint32_t i;
    for (i = 0; i < 32; ++i) {
        uint64_t r;
        int32_t d;
        if ((i & 7u) == 0u) { st += 0x9E3779B97F4A7C15ull; }
        r = zmi_mix64(st) >> (8u * (i & 7u));
        d = (int32_t)(r & 0x7Fu) - 64;
        v += d;
        out[2 * i] = (uint8_t)(v & 0xFF);
        out[2 * i + 1] = (uint8_t)((v >> 8) & 0xFF);
    }
}

ZMI_HD void zmi_line_exe(uint64_t key, uint32_t line, uint8_t* out) {
    /* opcode-like bytes from a skewed 64-symbol alphabet, 4-byte addresses with shared high bytes, zero runs */
    uint64_t st = key;
    uint32_t pos = 0;
    while (pos < 64u) {
        uint64_t r;
        uint32_t kind;
        st += 0x9E3779B97F4A7C15ull;
        r = zmi_mix64(st);
        kind = (uint32_t)(r & 15u);
        if (kind < 2u) { /* zero run of 4..11 bytes */
            uint32_t n = 4u + ((uint32_t)(r >> 4) & 7u), i;
            for (i = 0; i < n && pos < 64u; ++i) out[pos++] = 0;
        } else if (kind < 6u) { /* call/jump: opcode + 4-byte LE address, high 2 bytes shared */
            uint32_t addr = 0x00400000u + ((line >> 6) << 12) + ((uint32_t)(r >> 8) & 0xFFCu);
            uint32_t i;
            if (pos < 64u) out[pos++] = (uint8_t)(((r >> 4) & 1u) ? 0xE8u : 0xE9u);
            for (i = 0; i < 4 && pos < 64u; ++i) out[pos++] = (uint8_t)(addr >> (8u * i));
        } else { /* 2..4 opcode bytes */
            uint32_t n = 2u + ((uint32_t)(r >> 4) & 3u), i;
            if (n > 4u) n = 4u;
            for (i = 0; i < n && pos < 64u; ++i) {
                uint32_t z = zmi_zipf(r >> (12u + 12u * i), 6);
                out[pos++] = (uint8_t)((z * 0x4Du + 0x0Fu) & 0xFFu);
            }
        }
    }
}

ZMI_HD void zmi_line_rand(uint64_t key, uint8_t* out) {
    uint32_t i, b;
    for (i = 0; i < 8; ++i) {
        uint64_t r = zmi_mix64(key + i * 0x632BE59BD9B4E019ull);
        for (b = 0; b < 8; ++b) out[8 * i + b] = (uint8_t)(r >> (8u * b));
    }
}

/* one 64-byte line of shard `shard` (lines_per_shard lines in total) */
ZMI_HD void zmi_gen_line(uint64_t seed, uint32_t shard, uint32_t line, uint32_t lines_per_shard, uint8_t* out) {
    uint64_t key = zmi_mix64(seed ^ zmi_mix64(((uint64_t)shard << 32) | line));
    uint32_t cls = shard & 7u;
    if (cls == 7u) cls = (line < lines_per_shard / 2u) ? 8u : 0u;
    switch (cls) {
        case 0: case 1: case 2: zmi_line_text(key, out); break;
        case 3: zmi_line_xml(key, shard, line, out); break;
        case 4: zmi_line_db(key, shard, line, out); break;
        case 5: zmi_line_walk(key, line, out); break;
        case 6: zmi_line_exe(key, line, out); break;
        default: zmi_line_rand(key, out); break;
    }
}
#endif
