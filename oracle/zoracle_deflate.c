/* zoracle_deflate.c -- CPU restatement of the reference's deflate path (one-shot: whole input,
 * Z_FINISH, output buffer >= compress_bound).  TEST INFRASTRUCTURE, see the header of zoracle.c.
 *
 * Restated from (paths under /root/reference/zlib-rs/src):
 *   level table                deflate/algorithm/mod.rs:69-82
 *   window / hash              deflate.rs:1776-1861 (fill_window), :1687-1726 (read_buf_window),
 *                              deflate/hash_calc.rs:25-137, deflate/slide_hash.rs:11-47
 *   match search               deflate/longest_match.rs:15-346
 *   strategies                 deflate/algorithm/{stored,quick,fast,medium,slow,huff,rle}.rs
 *   symbol buffer / tally      deflate/sym_buf.rs, deflate.rs:1464-1521
 *   Huffman trees              deflate.rs:1945-2160 (build_tree, gen_bitlen, gen_codes), :2997-3135 (Heap)
 *   tree transmission          deflate.rs:2171-2314,1177-1260
 *   block flush                deflate.rs:2316-2434, stored block :1734-1763
 *   bit writer                 deflate.rs:907-1175
 *   framing                    deflate.rs:1572-1601 (zlib header), :2574-2627 (gzip header), :2772-2789 (trailers)
 * This is the zlib algorithm family (the reference follows zlib-ng); the C below is written from
 * the format and the behaviour of the cited code, not transliterated from it.
 *
 * Parity: the byte-exact golden vectors of the reference that this file reproduces are listed in
 * tests/test_oracle.py (hello-world huffman-only / quick / gzip level 9 / stored, `Ferris`,
 * `deflate_medium_bypass`, flush-free vectors); Z_HUFFMAN_ONLY, Z_RLE and level 0 output is
 * additionally byte-identical to system zlib 1.2.11 on arbitrary input, which pins the tree
 * builder and bit writer independently of any match finder.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

uint32_t zo_adler32(uint32_t adler, const uint8_t* buf, size_t len);
uint32_t zo_crc32(uint32_t crc, const uint8_t* buf, size_t len);

#define MIN_MATCH 3
#define MAX_MATCH 258
#define WANT_MIN_MATCH 4
#define MIN_LOOKAHEAD (MAX_MATCH + MIN_MATCH + 1)
#define L_CODES 286
#define D_CODES 30
#define BL_CODES 19
#define HEAP_SIZE (2 * L_CODES + 1)
#define MAX_BITS 15
#define MAX_BL_BITS 7
#define END_BLOCK 256
#define HASH_SIZE 65536u
#define MAX_STORED 65535u

typedef struct { uint16_t fc; uint16_t dl; } ct_data; /* freq|code , dad|len */

static const uint8_t extra_lbits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint8_t extra_dbits[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t extra_blbits[19] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 3, 7};
static const uint8_t bl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

/* static tables, generated once (RFC 1951 3.2.5 / 3.2.6) */
static ct_data static_ltree[L_CODES + 2];
static ct_data static_dtree[D_CODES];
static uint8_t length_code[256];
static uint8_t dist_code[512];
static uint16_t base_length[29];
static uint16_t base_dist[30];
static int tables_ready = 0;

static unsigned bi_reverse(unsigned code, int len) {
    unsigned res = 0;
    do { res |= code & 1u; code >>= 1; res <<= 1; } while (--len > 0);
    return res >> 1;
}

typedef struct {
    ct_data* dyn;
    const ct_data* stat;
    const uint8_t* extra;
    int extra_base, elems, max_length, max_code;
} tree_desc;

static void gen_codes(ct_data* tree, int max_code, const uint16_t* bl_count) {
    uint16_t next_code[MAX_BITS + 1];
    unsigned code = 0;
    for (int bits = 1; bits <= MAX_BITS; ++bits) {
        code = (code + bl_count[bits - 1]) << 1;
        next_code[bits] = (uint16_t)code;
    }
    for (int n = 0; n <= max_code; ++n) {
        int len = tree[n].dl;
        if (len == 0) continue;
        tree[n].fc = (uint16_t)bi_reverse(next_code[len]++, len);
    }
}

static void tables_init(void) {
    int length = 0, code, n, dist = 0;
    for (code = 0; code < 28; ++code) {
        base_length[code] = (uint16_t)length;
        for (n = 0; n < (1 << extra_lbits[code]); ++n) length_code[length++] = (uint8_t)code;
    }
    length_code[length - 1] = (uint8_t)code; /* length 258 -> code 28 */
    base_length[28] = 255;
    for (code = 0; code < 16; ++code) {
        base_dist[code] = (uint16_t)dist;
        for (n = 0; n < (1 << extra_dbits[code]); ++n) dist_code[dist++] = (uint8_t)code;
    }
    dist >>= 7;
    for (; code < D_CODES; ++code) {
        base_dist[code] = (uint16_t)(dist << 7);
        for (n = 0; n < (1 << (extra_dbits[code] - 7)); ++n) dist_code[256 + dist++] = (uint8_t)code;
    }
    uint16_t bl_count[MAX_BITS + 1];
    memset(bl_count, 0, sizeof bl_count);
    n = 0;
    while (n <= 143) { static_ltree[n++].dl = 8; bl_count[8]++; }
    while (n <= 255) { static_ltree[n++].dl = 9; bl_count[9]++; }
    while (n <= 279) { static_ltree[n++].dl = 7; bl_count[7]++; }
    while (n <= 287) { static_ltree[n++].dl = 8; bl_count[8]++; }
    gen_codes(static_ltree, L_CODES + 1, bl_count);
    for (n = 0; n < D_CODES; ++n) { static_dtree[n].dl = 5; static_dtree[n].fc = (uint16_t)bi_reverse((unsigned)n, 5); }
    tables_ready = 1;
}
#define D_CODE(d) ((d) < 256 ? dist_code[d] : dist_code[256 + ((d) >> 7)])

typedef struct {
    /* input / output */
    const uint8_t* in; size_t in_len, in_pos;
    uint8_t* out; size_t out_cap, out_pos; int overflow;
    int level, strategy, wrap;
    /* window */
    uint32_t w_size, w_mask, window_size;
    uint8_t* window; uint16_t* prev; uint16_t* head;
    uint32_t strstart, lookahead, match_start, prev_match, prev_length, insert, ins_h;
    int match_available, roll_hash;
    long block_start;
    uint32_t max_chain, good_match, nice_match, max_lazy;
    uint32_t check;
    /* symbols */
    uint8_t* sym_buf; uint32_t sym_next, sym_end, lit_bufsize;
    ct_data dyn_ltree[HEAP_SIZE], dyn_dtree[2 * D_CODES + 1], bl_tree[2 * BL_CODES + 1];
    tree_desc l_desc, d_desc, bl_desc;
    int heap[HEAP_SIZE]; int heap_len, heap_max; uint8_t depth[HEAP_SIZE];
    uint64_t opt_len, static_len;
    /* bits */
    uint64_t bi_buf; uint32_t bi_valid;
    int block_open;
} zst;

/* ---------------- bit writer ---------------- */
static void put_byte(zst* s, uint8_t b) { if (s->out_pos < s->out_cap) s->out[s->out_pos++] = b; else s->overflow = 1; }
static void send_bits(zst* s, uint64_t value, uint32_t len) {
    s->bi_buf |= value << s->bi_valid;
    s->bi_valid += len;
    while (s->bi_valid >= 8) { put_byte(s, (uint8_t)s->bi_buf); s->bi_buf >>= 8; s->bi_valid -= 8; }
}
static void bi_align(zst* s) {
    if (s->bi_valid > 0) put_byte(s, (uint8_t)s->bi_buf);
    s->bi_buf = 0;
    s->bi_valid = 0;
}
#define send_code(s, c, tree) send_bits(s, (tree)[c].fc, (tree)[c].dl)

/* ---------------- trees ---------------- */
static void init_block(zst* s) {
    for (int n = 0; n < L_CODES; ++n) s->dyn_ltree[n].fc = 0;
    for (int n = 0; n < D_CODES; ++n) s->dyn_dtree[n].fc = 0;
    for (int n = 0; n < BL_CODES; ++n) s->bl_tree[n].fc = 0;
    s->dyn_ltree[END_BLOCK].fc = 1;
    s->opt_len = s->static_len = 0;
    s->sym_next = 0;
}
#define KEY(s, tree, i) (((uint32_t)(tree)[i].fc << 8) | (s)->depth[i])
static void pqdownheap(zst* s, const ct_data* tree, int k) {
    int v = s->heap[k];
    uint32_t vk = KEY(s, tree, v);
    int j = k << 1;
    while (j <= s->heap_len) {
        uint32_t jk = KEY(s, tree, s->heap[j]);
        if (j < s->heap_len) {
            uint32_t j1 = KEY(s, tree, s->heap[j + 1]);
            if (j1 <= jk) { ++j; jk = j1; }
        }
        if (vk <= jk) break;
        s->heap[k] = s->heap[j];
        k = j;
        j <<= 1;
    }
    s->heap[k] = v;
}
static void gen_bitlen(zst* s, tree_desc* d, uint16_t* bl_count) {
    ct_data* tree = d->dyn;
    int overflow = 0, h;
    memset(bl_count, 0, sizeof(uint16_t) * (MAX_BITS + 1));
    tree[s->heap[s->heap_max]].dl = 0;
    for (h = s->heap_max + 1; h < HEAP_SIZE; ++h) {
        int n = s->heap[h];
        int bits = tree[tree[n].dl].dl + 1;
        if (bits > d->max_length) { bits = d->max_length; ++overflow; }
        tree[n].dl = (uint16_t)bits;
        if (n > d->max_code) continue;
        bl_count[bits]++;
        int xbits = (n >= d->extra_base) ? d->extra[n - d->extra_base] : 0;
        uint64_t f = tree[n].fc;
        s->opt_len += f * (uint64_t)(bits + xbits);
        if (d->stat) s->static_len += f * (uint64_t)(d->stat[n].dl + xbits);
    }
    if (overflow == 0) return;
    do {
        int bits = d->max_length - 1;
        while (bl_count[bits] == 0) --bits;
        bl_count[bits]--;
        bl_count[bits + 1] += 2;
        bl_count[d->max_length]--;
        overflow -= 2;
    } while (overflow > 0);
    h = HEAP_SIZE;
    for (int bits = d->max_length; bits != 0; --bits) {
        int n = bl_count[bits];
        while (n != 0) {
            int m = s->heap[--h];
            if (m > d->max_code) continue;
            if (tree[m].dl != (unsigned)bits) {
                s->opt_len += ((uint64_t)bits - tree[m].dl) * tree[m].fc;
                tree[m].dl = (uint16_t)bits;
            }
            --n;
        }
    }
}
static void build_tree(zst* s, tree_desc* d) {
    ct_data* tree = d->dyn;
    int n, m, max_code = -1, node;
    s->heap_len = 0;
    s->heap_max = HEAP_SIZE;
    for (n = 0; n < d->elems; ++n) {
        if (tree[n].fc != 0) { s->heap[++s->heap_len] = max_code = n; s->depth[n] = 0; }
        else tree[n].dl = 0;
    }
    while (s->heap_len < 2) {
        node = s->heap[++s->heap_len] = (max_code < 2 ? ++max_code : 0);
        tree[node].fc = 1;
        s->depth[node] = 0;
        s->opt_len--;
        if (d->stat) s->static_len -= d->stat[node].dl;
    }
    d->max_code = max_code;
    for (n = s->heap_len / 2; n >= 1; --n) pqdownheap(s, tree, n);
    node = d->elems;
    do {
        n = s->heap[1];
        s->heap[1] = s->heap[s->heap_len--];
        pqdownheap(s, tree, 1);
        m = s->heap[1];
        s->heap[--s->heap_max] = n;
        s->heap[--s->heap_max] = m;
        tree[node].fc = (uint16_t)(tree[n].fc + tree[m].fc);
        s->depth[node] = (uint8_t)((s->depth[n] >= s->depth[m] ? s->depth[n] : s->depth[m]) + 1);
        tree[n].dl = tree[m].dl = (uint16_t)node;
        s->heap[1] = node++;
        pqdownheap(s, tree, 1);
    } while (s->heap_len >= 2);
    s->heap[--s->heap_max] = s->heap[1];
    uint16_t bl_count[MAX_BITS + 1];
    gen_bitlen(s, d, bl_count);
    gen_codes(tree, max_code, bl_count);
}
static void scan_tree(zst* s, ct_data* tree, int max_code) {
    int prevlen = -1, curlen, nextlen = tree[0].dl, count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) { max_count = 138; min_count = 3; }
    tree[max_code + 1].dl = 0xFFFF;
    for (int n = 0; n <= max_code; ++n) {
        curlen = nextlen;
        nextlen = tree[n + 1].dl;
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) s->bl_tree[curlen].fc = (uint16_t)(s->bl_tree[curlen].fc + count);
        else if (curlen != 0) { if (curlen != prevlen) s->bl_tree[curlen].fc++; s->bl_tree[16].fc++; }
        else if (count <= 10) s->bl_tree[17].fc++;
        else s->bl_tree[18].fc++;
        count = 0;
        prevlen = curlen;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        else if (curlen == nextlen) { max_count = 6; min_count = 3; }
        else { max_count = 7; min_count = 4; }
    }
}
static void send_tree(zst* s, ct_data* tree, int max_code) {
    int prevlen = -1, curlen, nextlen = tree[0].dl, count = 0, max_count = 7, min_count = 4;
    if (nextlen == 0) { max_count = 138; min_count = 3; }
    for (int n = 0; n <= max_code; ++n) {
        curlen = nextlen;
        nextlen = tree[n + 1].dl;
        if (++count < max_count && curlen == nextlen) continue;
        else if (count < min_count) { do { send_code(s, curlen, s->bl_tree); } while (--count != 0); }
        else if (curlen != 0) {
            if (curlen != prevlen) { send_code(s, curlen, s->bl_tree); --count; }
            send_code(s, 16, s->bl_tree);
            send_bits(s, (uint64_t)(count - 3), 2);
        } else if (count <= 10) { send_code(s, 17, s->bl_tree); send_bits(s, (uint64_t)(count - 3), 3); }
        else { send_code(s, 18, s->bl_tree); send_bits(s, (uint64_t)(count - 11), 7); }
        count = 0;
        prevlen = curlen;
        if (nextlen == 0) { max_count = 138; min_count = 3; }
        else if (curlen == nextlen) { max_count = 6; min_count = 3; }
        else { max_count = 7; min_count = 4; }
    }
}
static int build_bl_tree(zst* s) {
    scan_tree(s, s->dyn_ltree, s->l_desc.max_code);
    scan_tree(s, s->dyn_dtree, s->d_desc.max_code);
    build_tree(s, &s->bl_desc);
    int max_blindex;
    for (max_blindex = BL_CODES - 1; max_blindex >= 3; --max_blindex)
        if (s->bl_tree[bl_order[max_blindex]].dl != 0) break;
    s->opt_len += 3ull * ((uint64_t)max_blindex + 1) + 5 + 5 + 4;
    return max_blindex;
}
static void compress_block(zst* s, const ct_data* ltree, const ct_data* dtree) {
    for (uint32_t sx = 0; sx < s->sym_next; sx += 3) {
        unsigned dist = s->sym_buf[sx] | ((unsigned)s->sym_buf[sx + 1] << 8);
        unsigned lc = s->sym_buf[sx + 2];
        if (dist == 0) { send_code(s, lc, ltree); continue; }
        unsigned code = length_code[lc];
        send_code(s, code + 256 + 1, ltree);
        if (extra_lbits[code]) send_bits(s, lc - base_length[code], extra_lbits[code]);
        --dist;
        code = D_CODE(dist);
        send_code(s, code, dtree);
        if (extra_dbits[code]) send_bits(s, dist - base_dist[code], extra_dbits[code]);
    }
    send_code(s, END_BLOCK, ltree);
}
static void stored_block(zst* s, const uint8_t* buf, uint32_t len, int last) {
    send_bits(s, (uint64_t)last, 3);
    bi_align(s);
    put_byte(s, (uint8_t)len); put_byte(s, (uint8_t)(len >> 8));
    put_byte(s, (uint8_t)~len); put_byte(s, (uint8_t)(~len >> 8));
    for (uint32_t i = 0; i < len; ++i) put_byte(s, buf[i]);
}
/* deflate.rs:2316-2434 */
static void flush_block(zst* s, int last) {
    const uint8_t* buf = s->block_start >= 0 ? s->window + s->block_start : NULL;
    uint32_t stored_len = (uint32_t)((long)s->strstart - s->block_start);
    uint64_t opt_lenb, static_lenb;
    int max_blindex = 0;
    if (s->sym_next == 0) {
        opt_lenb = 0; static_lenb = 0; s->static_len = 7;
    } else if (s->level > 0) {
        build_tree(s, &s->l_desc);
        build_tree(s, &s->d_desc);
        max_blindex = build_bl_tree(s);
        opt_lenb = (s->opt_len + 3 + 7) >> 3;
        static_lenb = (s->static_len + 3 + 7) >> 3;
        if (static_lenb <= opt_lenb || s->strategy == 4) opt_lenb = static_lenb;
    } else {
        opt_lenb = static_lenb = (uint64_t)stored_len + 5;
    }
    if ((uint64_t)stored_len + 4 <= opt_lenb && buf != NULL) {
        stored_block(s, buf, stored_len, last);
    } else if (static_lenb == opt_lenb) {
        send_bits(s, (uint64_t)((1 << 1) + last), 3);
        compress_block(s, static_ltree, static_dtree);
    } else {
        send_bits(s, (uint64_t)((2 << 1) + last), 3);
        send_bits(s, (uint64_t)(s->l_desc.max_code + 1 - 257), 5);
        send_bits(s, (uint64_t)(s->d_desc.max_code + 1 - 1), 5);
        send_bits(s, (uint64_t)(max_blindex + 1 - 4), 4);
        for (int rank = 0; rank <= max_blindex; ++rank) send_bits(s, s->bl_tree[bl_order[rank]].dl, 3);
        send_tree(s, s->dyn_ltree, s->l_desc.max_code);
        send_tree(s, s->dyn_dtree, s->d_desc.max_code);
        compress_block(s, s->dyn_ltree, s->dyn_dtree);
    }
    init_block(s);
    if (last) bi_align(s);
    s->block_start = (long)s->strstart;
}
static int tally_lit(zst* s, uint8_t c) {
    s->sym_buf[s->sym_next++] = 0; s->sym_buf[s->sym_next++] = 0; s->sym_buf[s->sym_next++] = c;
    s->dyn_ltree[c].fc++;
    return s->sym_next == s->sym_end;
}
static int tally_dist(zst* s, uint32_t dist, uint32_t len) {
    s->sym_buf[s->sym_next++] = (uint8_t)dist; s->sym_buf[s->sym_next++] = (uint8_t)(dist >> 8); s->sym_buf[s->sym_next++] = (uint8_t)len;
    --dist;
    s->dyn_ltree[length_code[len] + 256 + 1].fc++;
    s->dyn_dtree[D_CODE(dist)].fc++;
    return s->sym_next == s->sym_end;
}

/* ---------------- window / hash ---------------- */
static uint32_t rd32(const uint8_t* p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }
#define MAX_DIST(s) ((s)->w_size - MIN_LOOKAHEAD)

static uint16_t std_insert_value(zst* s, uint32_t str, uint32_t val) {
    uint32_t hm = (val * 2654435761u) >> 16;
    uint16_t head = s->head[hm];
    if (head != (uint16_t)str) { s->prev[str & s->w_mask] = head; s->head[hm] = (uint16_t)str; }
    return head;
}
static uint16_t quick_insert_string(zst* s, uint32_t str) {
    if (!s->roll_hash) return std_insert_value(s, str, rd32(s->window + str));
    s->ins_h = ((s->ins_h << 5) ^ s->window[str + 2]) & 0x7FFFu;
    uint16_t head = s->head[s->ins_h];
    if (head != (uint16_t)str) { s->prev[str & s->w_mask] = head; s->head[s->ins_h] = (uint16_t)str; }
    return head;
}
static void insert_string(zst* s, uint32_t str, uint32_t count) {
    if (!s->roll_hash) {
        /* bounded by the window allocation: hash_calc.rs:61-83 */
        uint32_t avail = s->window_size - str;
        uint32_t span = count + 3 < avail ? count + 3 : avail;
        for (uint32_t i = 0; i + 4 <= span; ++i) std_insert_value(s, (uint16_t)(str + i), rd32(s->window + str + i));
    } else {
        for (uint32_t i = 0; i < count; ++i) {
            uint32_t idx = (uint16_t)(str + i);
            s->ins_h = ((s->ins_h << 5) ^ s->window[str + i + 2]) & 0x7FFFu;
            uint16_t head = s->head[s->ins_h];
            if (head != (uint16_t)idx) { s->prev[idx & s->w_mask] = head; s->head[s->ins_h] = (uint16_t)idx; }
        }
    }
}
static uint32_t read_buf(zst* s, uint32_t off, uint32_t size) {
    size_t avail = s->in_len - s->in_pos;
    uint32_t len = avail < size ? (uint32_t)avail : size;
    if (len == 0) return 0;
    memcpy(s->window + off, s->in + s->in_pos, len);
    if (s->wrap == 2) s->check = zo_crc32(s->check, s->in + s->in_pos, len);
    else if (s->wrap == 1) s->check = zo_adler32(s->check, s->in + s->in_pos, len);
    s->in_pos += len;
    return len;
}
static void fill_window(zst* s) {
    uint32_t wsize = s->w_size;
    for (;;) {
        uint32_t more = s->window_size - s->lookahead - s->strstart;
        if (s->strstart >= wsize + MAX_DIST(s)) {
            memcpy(s->window, s->window + wsize, wsize);
            if (s->match_start >= wsize) s->match_start -= wsize; else { s->match_start = 0; s->prev_length = 0; }
            s->strstart -= wsize;
            s->block_start -= (long)wsize;
            if (s->insert > s->strstart) s->insert = s->strstart;
            for (uint32_t i = 0; i < HASH_SIZE; ++i) s->head[i] = (uint16_t)(s->head[i] >= wsize ? s->head[i] - wsize : 0);
            for (uint32_t i = 0; i < wsize; ++i) s->prev[i] = (uint16_t)(s->prev[i] >= wsize ? s->prev[i] - wsize : 0);
            more += wsize;
        }
        if (s->in_pos >= s->in_len) break;
        uint32_t n = read_buf(s, s->strstart + s->lookahead, more);
        s->lookahead += n;
        if (s->lookahead + s->insert >= MIN_MATCH) {
            uint32_t str = s->strstart - s->insert;
            if (s->roll_hash) {
                s->ins_h = (((uint32_t)s->window[str] << 5) ^ s->window[str + 1]) & 0x7FFFu;
            } else if (str >= 1) {
                quick_insert_string(s, str + 2 - MIN_MATCH);
            }
            uint32_t count = s->insert;
            if (s->lookahead == 1) count -= 1;
            if (count > 0) { insert_string(s, str, count); s->insert -= count; }
        }
        if (!(s->lookahead < MIN_LOOKAHEAD && s->in_pos < s->in_len)) break;
    }
}

static uint32_t compare256(const uint8_t* a, const uint8_t* b) {
    uint32_t n = 0;
    while (n < 256 && a[n] == b[n]) ++n;
    return n;
}

/* deflate/longest_match.rs:15-346; returns length, writes the match position to *mstart */
static uint32_t longest_match(zst* s, uint32_t cur_match, int slow, uint32_t* mstart) {
    uint32_t match_start = s->match_start;
    const uint32_t strstart = s->strstart, wmask = s->w_mask;
    const uint8_t* window = s->window;
    const uint8_t* scan = window + strstart;
    uint32_t best_len = s->prev_length > 0 ? s->prev_length : MIN_MATCH - 1;
    uint32_t offset = best_len - 1;
    if (best_len >= 4) { offset -= 2; if (best_len >= 8) offset -= 4; }
    uint32_t chain_length = s->max_chain;
    if (best_len >= s->good_match) chain_length >>= 2;
    const uint32_t nice_match = s->nice_match, lookahead = s->lookahead;
    uint32_t limit = strstart > MAX_DIST(s) ? strstart - MAX_DIST(s) : 0;
    uint32_t limit_base = limit, match_offset = 0;
    int early_exit = 0;
    long mbase = 0; /* candidate strings are read at window[cur_match - match_offset + ...] */
#define BREAK_MATCHING() do { *mstart = match_start; return best_len < lookahead ? best_len : lookahead; } while (0)
    if (slow) {
        if (best_len >= MIN_MATCH) {
            uint32_t hash = 0;
            hash = ((hash << 5) ^ scan[1]) & 0x7FFFu;
            hash = ((hash << 5) ^ scan[2]) & 0x7FFFu;
            for (uint32_t i = 3; i <= best_len; ++i) {
                hash = ((hash << 5) ^ scan[i]) & 0x7FFFu;
                uint32_t pos = s->head[hash];
                if (pos < cur_match) { match_offset = i - 2; cur_match = pos; }
            }
            limit = limit_base + match_offset;
            if (cur_match <= limit) BREAK_MATCHING();
            mbase = -(long)match_offset;
        }
    } else {
        early_exit = s->level < 5;
    }
    uint32_t scan_end_off = offset;
    for (;;) {
        if (cur_match >= strstart) break;
        uint32_t len = 0;
        int advance = 0;
        const uint8_t* cand = window + (long)cur_match + mbase;
        if (best_len < 8) {
            uint64_t sv = rd64(scan);
            for (;;) {
                cand = window + (long)cur_match + mbase;
                uint64_t cmp = sv ^ rd64(cand);
                if (cmp == 0) break;
                uint32_t cmp_len = (uint32_t)(__builtin_ctzll(cmp) / 8);
                if (cmp_len > best_len) { len = cmp_len; break; }
                /* next in chain */
                if (--chain_length > 0) {
                    cur_match = s->prev[cur_match & wmask];
                    if (cur_match > limit) continue;
                }
                *mstart = match_start;
                return best_len;
            }
        } else {
            for (;;) {
                cand = window + (long)cur_match + mbase;
                if (rd64(cand + scan_end_off) == rd64(scan + scan_end_off) && rd64(cand) == rd64(scan)) break;
                if (--chain_length > 0) {
                    cur_match = s->prev[cur_match & wmask];
                    if (cur_match > limit) continue;
                }
                *mstart = match_start;
                return best_len;
            }
        }
        if (len == 0) len = compare256(scan + 2, cand + 2) + 2;
        if (len > best_len) {
            match_start = cur_match - match_offset;
            if (len >= lookahead) { *mstart = match_start; return lookahead; }
            best_len = len;
            if (best_len >= nice_match) { *mstart = match_start; return best_len; }
            offset = best_len - 1;
            if (best_len >= 4) { offset -= 2; if (best_len >= 8) offset -= 4; }
            scan_end_off = offset;
            if (slow && len > MIN_MATCH && match_start + len < strstart) {
                uint32_t pos, next_pos;
                cur_match -= match_offset;
                match_offset = 0;
                next_pos = cur_match;
                for (uint32_t i = 0; i <= len - MIN_MATCH; ++i) {
                    pos = s->prev[(cur_match + i) & wmask];
                    if (pos < next_pos) {
                        if (pos <= limit_base + i) BREAK_MATCHING();
                        next_pos = pos;
                        match_offset = i;
                    }
                }
                cur_match = next_pos;
                const uint8_t* t = scan + len - (MIN_MATCH + 1);
                uint32_t hash = 0;
                hash = ((hash << 5) ^ t[0]) & 0x7FFFu;
                hash = ((hash << 5) ^ t[1]) & 0x7FFFu;
                hash = ((hash << 5) ^ t[2]) & 0x7FFFu;
                pos = s->head[hash];
                if (pos < cur_match) {
                    match_offset = len - (MIN_MATCH + 1);
                    if (pos <= limit_base + match_offset) BREAK_MATCHING();
                    cur_match = pos;
                }
                limit = limit_base + match_offset;
                mbase = -(long)match_offset;
                continue;
            }
            advance = 1;
        } else if (!slow && early_exit) {
            break;
        }
        (void)advance;
        if (--chain_length > 0) {
            cur_match = s->prev[cur_match & wmask];
            if (cur_match > limit) continue;
        }
        break;
    }
    *mstart = match_start;
    return best_len;
#undef BREAK_MATCHING
}

/* ---------------- strategies ---------------- */
static void deflate_stored(zst* s) { /* algorithm/stored.rs, one-shot with ample output */
    size_t left = s->in_len;
    const uint8_t* p = s->in;
    if (s->wrap == 2) s->check = zo_crc32(s->check, p, left); else if (s->wrap == 1) s->check = zo_adler32(s->check, p, left);
    do {
        uint32_t len = left > MAX_STORED ? MAX_STORED : (uint32_t)left;
        int last = len == left;
        stored_block(s, p, len, last);
        p += len; left -= len;
    } while (left > 0);
    s->in_pos = s->in_len;
}
static void deflate_huff(zst* s) {
    for (;;) {
        if (s->lookahead == 0) { fill_window(s); if (s->lookahead == 0) break; }
        int bflush = tally_lit(s, s->window[s->strstart]);
        s->lookahead--; s->strstart++;
        if (bflush) flush_block(s, 0);
    }
    flush_block(s, 1);
}
static void deflate_rle(zst* s) {
    uint32_t match_len = 0;
    for (;;) {
        if (s->lookahead < MIN_LOOKAHEAD) { fill_window(s); if (s->lookahead == 0) break; }
        if (s->lookahead >= MIN_MATCH && s->strstart > 0) {
            const uint8_t* scan = s->window + s->strstart - 1;
            if (scan[0] == scan[1] && scan[1] == scan[2]) {
                uint32_t n = 0;
                while (n < 256 && scan[3 + n] == scan[0]) ++n;
                match_len = n + 2;
                if (match_len > s->lookahead) match_len = s->lookahead;
                if (match_len > MAX_MATCH) match_len = MAX_MATCH;
            }
        }
        int bflush;
        if (match_len >= MIN_MATCH) {
            bflush = tally_dist(s, 1, match_len - MIN_MATCH);
            s->lookahead -= match_len; s->strstart += match_len; match_len = 0;
        } else {
            bflush = tally_lit(s, s->window[s->strstart]);
            s->lookahead--; s->strstart++;
        }
        if (bflush) flush_block(s, 0);
    }
    flush_block(s, 1);
}
static void deflate_quick(zst* s) { /* algorithm/quick.rs: one static block, codes emitted directly */
    int started = 0;
    for (;;) {
        if (s->lookahead < MIN_LOOKAHEAD) {
            fill_window(s);
            if (s->lookahead == 0) break;
        }
        if (!started) { send_bits(s, (1 << 1) + 1, 3); started = 1; }
        uint8_t lc;
        if (s->lookahead >= WANT_MIN_MATCH) {
            uint32_t val = rd32(s->window + s->strstart);
            uint16_t hh = std_insert_value(s, s->strstart, val);
            long dist = (long)s->strstart - hh;
            if (dist <= (long)MAX_DIST(s) && dist > 0) {
                const uint8_t* m = s->window + hh;
                if (val == rd32(m)) {
                    uint32_t ml = compare256(s->window + s->strstart + 2, m + 2) + 2;
                    if (ml >= WANT_MIN_MATCH) {
                        if (ml > s->lookahead) ml = s->lookahead;
                        if (ml > MAX_MATCH) ml = MAX_MATCH;
                        uint32_t lcv = ml - MIN_MATCH, code = length_code[lcv];
                        send_code(s, code + 257, static_ltree);
                        if (extra_lbits[code]) send_bits(s, lcv - base_length[code], extra_lbits[code]);
                        uint32_t d = (uint32_t)dist - 1;
                        code = D_CODE(d);
                        send_code(s, code, static_dtree);
                        if (extra_dbits[code]) send_bits(s, d - base_dist[code], extra_dbits[code]);
                        s->lookahead -= ml; s->strstart += ml;
                        continue;
                    }
                }
            }
            lc = (uint8_t)val;
        } else {
            lc = s->window[s->strstart];
        }
        send_code(s, lc, static_ltree);
        s->strstart++; s->lookahead--;
    }
    if (!started) send_bits(s, (1 << 1) + 1, 3);  /* empty input: an empty final static block */
    send_code(s, END_BLOCK, static_ltree);
    bi_align(s);
}
static void deflate_fast(zst* s) {
    for (;;) {
        if (s->lookahead < MIN_LOOKAHEAD) { fill_window(s); if (s->lookahead == 0) break; }
        uint8_t lc;
        if (s->lookahead >= WANT_MIN_MATCH) {
            uint32_t val = rd32(s->window + s->strstart);
            uint16_t hh = std_insert_value(s, s->strstart, val);
            long dist = (long)s->strstart - hh;
            if (dist <= (long)MAX_DIST(s) && dist > 0 && hh != 0) {
                uint32_t ms;
                uint32_t ml = longest_match(s, hh, 0, &ms);
                s->match_start = ms;
                if (ml >= WANT_MIN_MATCH) {
                    int bflush = tally_dist(s, s->strstart - s->match_start, ml - MIN_MATCH);
                    s->lookahead -= ml;
                    if (ml <= s->max_lazy && s->lookahead >= WANT_MIN_MATCH) {
                        --ml; s->strstart++;
                        insert_string(s, s->strstart, ml);
                        s->strstart += ml;
                    } else {
                        s->strstart += ml;
                        std_insert_value(s, s->strstart + 2 - MIN_MATCH, rd32(s->window + s->strstart + 2 - MIN_MATCH));
                    }
                    if (bflush) flush_block(s, 0);
                    continue;
                }
            }
            lc = (uint8_t)val;
        } else {
            lc = s->window[s->strstart];
        }
        int bflush = tally_lit(s, lc);
        s->lookahead--; s->strstart++;
        if (bflush) flush_block(s, 0);
    }
    flush_block(s, 1);
}

/* algorithm/medium.rs */
typedef struct { uint16_t match_start, match_length, strstart, orgstart; } zmatch;
static int medium_emit(zst* s, zmatch m) {
    int bflush = 0;
    if (m.match_length < WANT_MIN_MATCH) {
        for (uint32_t i = 0; i < m.match_length; ++i) bflush |= tally_lit(s, s->window[s->strstart + i]);
    } else {
        bflush |= tally_dist(s, (uint32_t)(m.strstart - m.match_start), (uint32_t)m.match_length - MIN_MATCH);
    }
    s->lookahead -= m.match_length;
    return bflush;
}
static void medium_insert(zst* s, zmatch m) {
    if (s->lookahead <= (uint32_t)m.match_length + WANT_MIN_MATCH) return;
    if (m.match_length < WANT_MIN_MATCH) {
        m.strstart++; m.match_length--;
        if (m.match_length > 0 && m.strstart >= m.orgstart) {
            if ((uint32_t)m.strstart + m.match_length > m.orgstart) insert_string(s, m.strstart, m.match_length);
            else insert_string(s, m.strstart, (uint32_t)(m.orgstart - m.strstart + 1));
        }
        return;
    }
    if ((uint32_t)m.match_length <= 16 * s->max_lazy && s->lookahead >= WANT_MIN_MATCH) {
        m.match_length--; m.strstart++;
        if (m.strstart >= m.orgstart) {
            if ((uint32_t)m.strstart + m.match_length > m.orgstart) insert_string(s, m.strstart, m.match_length);
            else insert_string(s, m.strstart, (uint32_t)(m.orgstart - m.strstart + 1));
        } else if ((uint32_t)m.orgstart < (uint32_t)m.strstart + m.match_length) {
            insert_string(s, m.orgstart, (uint32_t)m.strstart + m.match_length - m.orgstart);
        }
    } else {
        m.strstart = (uint16_t)(m.strstart + m.match_length);
        m.match_length = 0;
        if (m.strstart >= MIN_MATCH - 2) quick_insert_string(s, (uint32_t)m.strstart + 2 - MIN_MATCH);
    }
}
static void medium_fizzle(zst* s, zmatch* current, zmatch* next) {
    const uint8_t* window = s->window;
    if (current->match_length <= 1) return;
    if ((uint32_t)current->match_length > 1u + next->match_start) return;
    if ((uint32_t)current->match_length > 1u + next->strstart) return;
    const uint8_t* m = window + (1 + (long)next->match_start - current->match_length);
    const uint8_t* orig = window + (1 + (long)next->strstart - current->match_length);
    if (m[0] != orig[0]) return;
    uint32_t md = MAX_DIST(s);
    uint16_t limit = next->strstart > (uint16_t)md ? (uint16_t)(next->strstart - (uint16_t)md) : 0;
    zmatch c = *current, n = *next;
    long mi = (long)n.match_start - 1, oi = (long)n.strstart - 1;
    int changed = 0;
    while (mi >= 0 && oi >= 0 && window[mi] == window[oi]) {
        if (c.match_length < 1) break;
        if (n.strstart <= limit) break;
        if (n.match_length >= 256) break;
        if (n.match_start <= 1) break;
        n.strstart--; n.match_start--; n.match_length++; c.match_length--;
        --mi; --oi;
        ++changed;
    }
    if (!changed) return;
    if (c.match_length <= 1 && n.match_length != 2) { n.orgstart++; *current = c; *next = n; }
}
static void deflate_medium(zst* s) {
    const int early_exit = s->level < 5;
    zmatch cur = {0, 0, 0, 0}, nxt = {0, 0, 0, 0};
    for (;;) {
        uint16_t hash_head;
        if (s->lookahead < MIN_LOOKAHEAD) {
            fill_window(s);
            if (s->lookahead == 0) break;
            nxt.match_length = 0;
        }
        if (!early_exit && nxt.match_length > 0) {
            cur = nxt;
            nxt.match_length = 0;
        } else {
            hash_head = 0;
            if (s->lookahead >= WANT_MIN_MATCH) hash_head = std_insert_value(s, s->strstart, rd32(s->window + s->strstart));
            cur.strstart = (uint16_t)s->strstart;
            cur.orgstart = cur.strstart;
            long dist = (long)s->strstart - hash_head;
            if (dist <= (long)MAX_DIST(s) && dist > 0 && hash_head != 0) {
                uint32_t ms;
                uint32_t ml = longest_match(s, hash_head, 0, &ms);
                s->match_start = ms;
                cur.match_length = (uint16_t)ml;
                cur.match_start = (uint16_t)ms;
                if (cur.match_length < WANT_MIN_MATCH) cur.match_length = 1;
                if (cur.match_start >= cur.strstart) cur.match_length = 1;
            } else {
                cur.match_start = 0;
                cur.match_length = 1;
            }
        }
        medium_insert(s, cur);
        if (!early_exit && s->lookahead > MIN_LOOKAHEAD &&
            (uint32_t)(uint16_t)(cur.strstart + cur.match_length) < s->window_size - MIN_LOOKAHEAD) {
            s->strstart = (uint32_t)(uint16_t)(cur.strstart + cur.match_length);
            hash_head = std_insert_value(s, s->strstart, rd32(s->window + s->strstart));
            nxt.strstart = (uint16_t)s->strstart;
            nxt.orgstart = nxt.strstart;
            long dist = (long)s->strstart - hash_head;
            if (dist <= (long)MAX_DIST(s) && dist > 0 && hash_head != 0) {
                uint32_t ms;
                uint32_t ml = longest_match(s, hash_head, 0, &ms);
                s->match_start = ms;
                nxt.match_length = (uint16_t)ml;
                nxt.match_start = (uint16_t)ms;
                if (nxt.match_start >= nxt.strstart) nxt.match_length = 1;
                if (nxt.match_length < WANT_MIN_MATCH) nxt.match_length = 1;
                else medium_fizzle(s, &cur, &nxt);
            } else {
                nxt.match_start = 0;
                nxt.match_length = 1;
            }
            s->strstart = cur.strstart;
        } else {
            nxt.match_length = 0;
        }
        int bflush = medium_emit(s, cur);
        s->strstart += cur.match_length;
        if (bflush) flush_block(s, 0);
    }
    flush_block(s, 1);
}
static void deflate_slow(zst* s) { /* algorithm/slow.rs */
    const int use_slow = s->max_chain > 1024;
    int match_available = 0;
    for (;;) {
        if (s->lookahead < MIN_LOOKAHEAD) { fill_window(s); if (s->lookahead == 0) break; }
        uint16_t hash_head = s->lookahead >= WANT_MIN_MATCH ? quick_insert_string(s, s->strstart) : 0;
        s->prev_match = s->match_start;
        uint32_t match_len = MIN_MATCH - 1;
        long dist = (long)s->strstart - hash_head;
        if (dist >= 1 && dist <= (long)MAX_DIST(s) && s->prev_length < s->max_lazy && hash_head != 0) {
            uint32_t ms;
            match_len = longest_match(s, hash_head, use_slow, &ms);
            s->match_start = ms;
            if (match_len <= 5 && s->strategy == 1) match_len = MIN_MATCH - 1;
        }
        if (s->prev_length >= MIN_MATCH && match_len <= s->prev_length) {
            uint32_t max_insert = s->strstart + s->lookahead - MIN_MATCH;
            int bflush = tally_dist(s, s->strstart - 1 - s->prev_match, s->prev_length - MIN_MATCH);
            s->prev_length -= 1;
            s->lookahead -= s->prev_length;
            uint32_t mov_fwd = s->prev_length - 1;
            if (max_insert > s->strstart) {
                uint32_t cnt = mov_fwd < max_insert - s->strstart ? mov_fwd : max_insert - s->strstart;
                insert_string(s, s->strstart + 1, cnt);
            }
            s->prev_length = 0;
            match_available = 0;
            s->strstart += mov_fwd + 1;
            if (bflush) flush_block(s, 0);
        } else if (match_available) {
            int bflush = tally_lit(s, s->window[s->strstart - 1]);
            if (bflush) flush_block(s, 0);
            s->prev_length = match_len;
            s->strstart++; s->lookahead--;
        } else {
            s->prev_length = match_len;
            match_available = 1;
            s->strstart++; s->lookahead--;
        }
    }
    if (match_available) tally_lit(s, s->window[s->strstart - 1]);
    flush_block(s, 1);
}

static const struct { uint16_t good, lazy, nice, chain; int func; } config_table[10] = {
    {0, 0, 0, 0, 0},        /* stored */
    {0, 0, 0, 0, 1},        /* quick */
    {4, 4, 8, 4, 2},        /* fast */
    {4, 6, 16, 6, 3},       /* medium */
    {4, 12, 32, 24, 3},     {8, 16, 32, 32, 3},   {8, 16, 128, 128, 3},
    {8, 32, 128, 256, 4},   /* slow */
    {32, 128, 258, 1024, 4}, {32, 258, 258, 4096, 4},
};

/* one-shot deflate; wrap 0 raw / 1 zlib / 2 gzip; strategy 0 default 1 filtered 2 huffman 3 rle 4 fixed;
 * returns 0 (Z_OK) or -5 (Z_BUF_ERROR) when out_cap is too small */
int zo_deflate2(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap, int level, int wrap, int strategy,
                int mem_level, int wbits, size_t* out_len) {
    if (!tables_ready) tables_init();
    if (level == -1) level = 6;
    if (level < 0 || level > 9 || mem_level < 1 || mem_level > 9 || wbits < 8 || wbits > 15) return -2;
    if (wbits == 8) wbits = 9; /* deflate.rs:308-312 */
    zst* s = (zst*)calloc(1, sizeof(zst));
    if (!s) return -4;
    s->in = in; s->in_len = in_len; s->out = out; s->out_cap = out_cap;
    s->level = level; s->strategy = strategy; s->wrap = wrap;
    s->w_size = 1u << wbits; s->w_mask = s->w_size - 1; s->window_size = 2 * s->w_size;
    s->window = (uint8_t*)calloc(1, s->window_size + 8 + 512);
    s->prev = (uint16_t*)calloc(s->w_size, 2);
    s->head = (uint16_t*)calloc(HASH_SIZE, 2);
    s->lit_bufsize = 1u << (mem_level + 6);
    s->sym_buf = (uint8_t*)calloc(s->lit_bufsize, 3);
    s->sym_end = (s->lit_bufsize - 1) * 3;
    s->l_desc = (tree_desc){s->dyn_ltree, static_ltree, extra_lbits, 257, L_CODES, MAX_BITS, 0};
    s->d_desc = (tree_desc){s->dyn_dtree, static_dtree, extra_dbits, 0, D_CODES, MAX_BITS, 0};
    s->bl_desc = (tree_desc){s->bl_tree, NULL, extra_blbits, 0, BL_CODES, MAX_BL_BITS, 0};
    s->good_match = config_table[level].good; s->max_lazy = config_table[level].lazy;
    s->nice_match = config_table[level].nice; s->max_chain = config_table[level].chain;
    s->roll_hash = s->max_chain > 1024;
    s->check = wrap == 1 ? 1u : 0u;
    init_block(s);
    /* header: deflate.rs:1572-1601 / :2574-2627 */
    if (wrap == 1) {
        unsigned lf = (strategy >= 2 || level < 2) ? 0 : (level < 6 ? 1 : (level == 6 ? 2 : 3));
        unsigned h = ((8u + ((unsigned)(wbits - 8) << 4)) << 8) | (lf << 6);
        h += 31 - (h % 31);
        put_byte(s, (uint8_t)(h >> 8)); put_byte(s, (uint8_t)h);
    } else if (wrap == 2) {
        static const uint8_t g[8] = {0x1F, 0x8B, 8, 0, 0, 0, 0, 0};
        for (int i = 0; i < 8; ++i) put_byte(s, g[i]);
        put_byte(s, (uint8_t)(level == 9 ? 2 : ((strategy >= 2 || level < 2) ? 4 : 0)));
        put_byte(s, 3);
    }
    if (level == 0) deflate_stored(s);
    else if (strategy == 2) deflate_huff(s);
    else if (strategy == 3) deflate_rle(s);
    else switch (config_table[level].func) {
        case 1: deflate_quick(s); break;
        case 2: deflate_fast(s); break;
        case 3: deflate_medium(s); break;
        default: deflate_slow(s); break;
    }
    if (wrap == 1) {
        put_byte(s, (uint8_t)(s->check >> 24)); put_byte(s, (uint8_t)(s->check >> 16));
        put_byte(s, (uint8_t)(s->check >> 8)); put_byte(s, (uint8_t)s->check);
    } else if (wrap == 2) {
        for (int i = 0; i < 4; ++i) put_byte(s, (uint8_t)(s->check >> (8 * i)));
        for (int i = 0; i < 4; ++i) put_byte(s, (uint8_t)((uint32_t)in_len >> (8 * i)));
    }
    int rc = s->overflow ? -5 : 0;
    if (out_len) *out_len = s->out_pos;
    free(s->window); free(s->prev); free(s->head); free(s->sym_buf); free(s);
    return rc;
}

int zo_deflate(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap, int level, int wrap, int strategy,
               int mem_level, size_t* out_len) {
    return zo_deflate2(in, in_len, out, out_cap, level, wrap, strategy, mem_level, 15, out_len);
}

/* ------------------------------------------------------------------------------------------
 * Multi-threaded timing harness for bench.py's cpu_baseline leg: generates `n_shards` synthetic
 * shards (csrc/shardgen.h), then compresses them with zo_deflate on `nthreads` POSIX threads
 * (static partition, one deflate state per call exactly as blogpost-compress.rs:43-122 drives the
 * reference).  Only the compression is timed.  Returns seconds; *total_out = compressed bytes.
 * ---------------------------------------------------------------------------------------- */
#include <pthread.h>
#include <time.h>
void zo_gen_shard(uint64_t seed, uint32_t shard, uint32_t nbytes, uint8_t* out);

typedef struct {
    uint8_t* data; uint32_t shard_bytes, first, count; int level; uint64_t out_bytes; uint8_t* scratch; size_t cap;
    pthread_barrier_t* bar;
} zo_job;
static void* zo_bench_worker(void* arg) {
    zo_job* j = (zo_job*)arg;
    pthread_barrier_wait(j->bar);
    for (uint32_t i = 0; i < j->count; ++i) {
        size_t olen = 0;
        zo_deflate(j->data + (size_t)(j->first + i) * j->shard_bytes, j->shard_bytes, j->scratch, j->cap, j->level, 1, 0, 8, &olen);
        j->out_bytes += olen;
    }
    pthread_barrier_wait(j->bar);
    return NULL;
}
static double zo_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
double zo_bench_deflate(uint64_t seed, uint32_t first_shard, uint32_t n_shards, uint32_t shard_bytes, int level, int nthreads,
                        uint64_t* total_out) {
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > n_shards) nthreads = (int)n_shards;
    uint8_t* data = (uint8_t*)malloc((size_t)n_shards * shard_bytes);
    if (!data) return -1.0;
    for (uint32_t i = 0; i < n_shards; ++i) zo_gen_shard(seed, first_shard + i, shard_bytes, data + (size_t)i * shard_bytes);
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    zo_job* jobs = (zo_job*)calloc((size_t)nthreads, sizeof(zo_job));
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)nthreads + 1);
    size_t cap = shard_bytes + shard_bytes / 8 + 4096;
    uint32_t per = n_shards / (uint32_t)nthreads, extra = n_shards % (uint32_t)nthreads, at = 0;
    for (int t = 0; t < nthreads; ++t) {
        jobs[t].data = data; jobs[t].shard_bytes = shard_bytes; jobs[t].first = at;
        jobs[t].count = per + ((uint32_t)t < extra ? 1u : 0u); at += jobs[t].count;
        jobs[t].level = level; jobs[t].scratch = (uint8_t*)malloc(cap); jobs[t].cap = cap; jobs[t].bar = &bar;
        pthread_create(&th[t], NULL, zo_bench_worker, &jobs[t]);
    }
    pthread_barrier_wait(&bar);
    double t0 = zo_now();
    pthread_barrier_wait(&bar);
    double dt = zo_now() - t0;
    uint64_t tot = 0;
    for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], NULL); tot += jobs[t].out_bytes; free(jobs[t].scratch); }
    pthread_barrier_destroy(&bar);
    free(th); free(jobs); free(data);
    if (total_out) *total_out = tot;
    return dt;
}

/* ------------------------------------------------------------------------------------------
 * Producer of the inflate benchmark's input (BASELINE.json configs[2]: "precompressed 1 MiB gzip blocks", produced by
 * the CPU reference path and not by the GPU deflater): the synthetic shards first_shard .. first_shard + n_shards - 1
 * compressed by this restatement of the reference at `level` with wrapper `wrap` (2 = gzip), `nthreads` POSIX threads,
 * member i at out + i * out_stride, its size in out_len[i].  Returns the seconds the compression took (a second
 * cpu_baseline sample), < 0 on failure.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t seed; uint32_t shard_bytes, first_shard, first, count; int level, wrap; uint8_t* out; size_t out_stride;
    uint32_t* out_len; int failed; pthread_barrier_t* bar;
} zo_member_job;
static void* zo_member_worker(void* arg) {
    zo_member_job* j = (zo_member_job*)arg;
    uint8_t* data = (uint8_t*)malloc(j->shard_bytes ? j->shard_bytes : 1);
    pthread_barrier_wait(j->bar);
    for (uint32_t i = 0; i < j->count && data; ++i) {
        const uint32_t k = j->first + i;
        size_t olen = 0;
        zo_gen_shard(j->seed, j->first_shard + k, j->shard_bytes, data);
        const int rc = zo_deflate(data, j->shard_bytes, j->out + (size_t)k * j->out_stride, j->out_stride, j->level, j->wrap, 0, 8, &olen);
        if (rc != 0 && rc != 1) j->failed = 1;
        j->out_len[k] = (uint32_t)olen;
    }
    if (!data) j->failed = 1;
    pthread_barrier_wait(j->bar);
    free(data);
    return NULL;
}
double zo_deflate_shards(uint64_t seed, uint32_t first_shard, uint32_t n_shards, uint32_t shard_bytes, int level, int wrap,
                         int nthreads, uint8_t* out, size_t out_stride, uint32_t* out_len) {
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > n_shards) nthreads = (int)n_shards;
    if (n_shards == 0) return 0.0;
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    zo_member_job* jobs = (zo_member_job*)calloc((size_t)nthreads, sizeof(zo_member_job));
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)nthreads + 1);
    uint32_t per = n_shards / (uint32_t)nthreads, extra = n_shards % (uint32_t)nthreads, at = 0;
    for (int t = 0; t < nthreads; ++t) {
        zo_member_job* j = &jobs[t];
        j->seed = seed; j->shard_bytes = shard_bytes; j->first_shard = first_shard; j->first = at;
        j->count = per + ((uint32_t)t < extra ? 1u : 0u); at += j->count;
        j->level = level; j->wrap = wrap; j->out = out; j->out_stride = out_stride; j->out_len = out_len; j->bar = &bar;
        pthread_create(&th[t], NULL, zo_member_worker, j);
    }
    pthread_barrier_wait(&bar);
    double t0 = zo_now();
    pthread_barrier_wait(&bar);
    double dt = zo_now() - t0;
    int failed = 0;
    for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], NULL); failed |= jobs[t].failed; }
    pthread_barrier_destroy(&bar);
    free(th); free(jobs);
    return failed ? -1.0 : dt;
}
