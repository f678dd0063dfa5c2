/* parse_lab.c -- ANALYSIS TOOL (not on any product path): given the per-position match words of lz77.hip
 * (lit | len << 8 | (dist-1) << 17) compare parse rules by the zeroth-order cost of the token stream they produce.
 * Built and driven by tools/parse_lab.py. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int len_idx(uint32_t len) {
    uint32_t l = len - 3;
    if (l == 255) return 28;
    if (l < 8) return (int)l;
    int k = 31 - __builtin_clz(l);
    return 4 * (k - 1) + ((l >> (k - 2)) & 3);
}
static int len_eb(int idx) { return (idx < 8 || idx == 28) ? 0 : (idx >> 2) - 1; }
static int dist_idx(uint32_t dist) {
    uint32_t d = dist - 1;
    if (d < 4) return (int)d;
    int k = 31 - __builtin_clz(d);
    return 2 * k + ((d >> (k - 1)) & 1);
}
static int dist_eb(int idx) { return idx < 4 ? 0 : (idx >> 1) - 1; }

/* tokens out: tok[i] = lit (len 0) or len << 8 | (dist-1) << 17 | lit; returns count.  The GPU's three-deep lazy rule. */
int parse_lazy3(const uint32_t* m, uint32_t n, uint32_t max_lazy, uint32_t lazy2, uint32_t lazy3, uint32_t* tok, uint32_t* tpos) {
    uint32_t p = 0; int nt = 0;
    while (p < n) {
        uint32_t l0 = (m[p] >> 8) & 0x1FF;
        uint32_t l1 = p + 1 < n ? (m[p + 1] >> 8) & 0x1FF : 0, l2 = p + 2 < n ? (m[p + 2] >> 8) & 0x1FF : 0, l3 = p + 3 < n ? (m[p + 3] >> 8) & 0x1FF : 0;
        int defer = (l0 < max_lazy) && ((l1 > l0) || (l2 > l0 + lazy2) || (l3 > l0 + lazy3));
        uint32_t step = 1;
        if (l0 >= 4 && !defer) step = l0;
        if (p + step > n) { step = n - p; if (step < 3) step = 1; }
        tpos[nt] = p;
        tok[nt++] = step > 1 ? ((m[p] & ~(0x1FFu << 8)) | (step << 8)) : (m[p] & 0xFF);
        p += step;
    }
    return nt;
}

/* zeroth-order cost (bits) of tokens [0, nt), in blocks of `blk` tokens: entropy of the two alphabets + extra bits + hdr per block */
double cost_tokens(const uint32_t* tok, int nt, int blk, double hdr_bits) {
    double total = 0;
    for (int b0 = 0; b0 < nt; b0 += blk) {
        int b1 = b0 + blk < nt ? b0 + blk : nt;
        double lf[288] = {0}, df[32] = {0}, extra = 0, nl = 0, nd = 0;
        for (int i = b0; i < b1; ++i) {
            uint32_t len = (tok[i] >> 8) & 0x1FF;
            if (!len) { lf[tok[i] & 0xFF]++; nl++; }
            else {
                int li = len_idx(len), di = dist_idx((tok[i] >> 17) + 1);
                lf[257 + li]++; nl++; df[di]++; nd++;
                extra += len_eb(li) + dist_eb(di);
            }
        }
        lf[256]++; nl++;
        double bits = extra + hdr_bits;
        for (int s = 0; s < 288; ++s) if (lf[s] > 0) bits += lf[s] * log2(nl / lf[s]);
        for (int s = 0; s < 32; ++s) if (df[s] > 0) bits += df[s] * log2(nd / df[s]);
        total += bits;
    }
    return total;
}

/* prices (in 1/16 bit) from the statistics of tokens [t0, t1) */
void prices_from(const uint32_t* tok, int t0, int t1, uint32_t* lprice, uint32_t* dprice) {
    double lf[288], df[32], nl = 0, nd = 0;
    for (int s = 0; s < 288; ++s) lf[s] = 0.25;   /* smoothing: unseen symbols are expensive, not impossible */
    for (int s = 0; s < 32; ++s) df[s] = 0.25;
    for (int i = t0; i < t1; ++i) {
        uint32_t len = (tok[i] >> 8) & 0x1FF;
        if (!len) lf[tok[i] & 0xFF]++;
        else { lf[257 + len_idx(len)]++; df[dist_idx((tok[i] >> 17) + 1)]++; }
    }
    for (int s = 0; s < 288; ++s) nl += lf[s];
    for (int s = 0; s < 32; ++s) nd += df[s];
    for (int s = 0; s < 288; ++s) { double b = log2(nl / lf[s]); if (b > 15) b = 15; if (b < 1) b = 1; lprice[s] = (uint32_t)(b * 16 + 0.5); }
    for (int s = 0; s < 32; ++s) { double b = log2(nd / df[s]); if (b > 15) b = 15; if (b < 1) b = 1; dprice[s] = (uint32_t)(b * 16 + 0.5); }
}

/* backward cost parse of positions [p0, p1) given prices; mode 0: literal or the full match; mode 1: also every shorter
 * length >= minlen at the same distance; mode 2: full match, or truncated to land where a later match starts (lengths l with
 * p + l in [p+minlen, p+len], cheap form: try len, len-1, len-2, len-3 only).  Writes the chosen step per position in step[]
 * and returns tokens appended to tok.  cost[] must hold p1 - p0 + 1 entries. */
int parse_dp(const uint32_t* m, uint32_t p0, uint32_t p1, const uint32_t* lprice, const uint32_t* dprice, int mode, uint32_t minlen,
             uint32_t* cost, uint16_t* step, uint32_t* tok, uint32_t* tpos, int nt) {
    const uint32_t n = p1 - p0;
    cost[n] = 0;
    for (int64_t i = (int64_t)n - 1; i >= 0; --i) {
        const uint32_t w = m[p0 + i];
        uint32_t best = lprice[w & 0xFF] + cost[i + 1], bs = 1;
        uint32_t len = (w >> 8) & 0x1FF;
        if (len >= 3) {
            if (i + len > n) len = n - (uint32_t)i;
            const uint32_t di = (uint32_t)dist_idx((w >> 17) + 1);
            const uint32_t dc = dprice[di] + 16u * (uint32_t)dist_eb((int)di);
            uint32_t lo = len;
            if (mode == 1) lo = minlen;
            else if (mode == 2) lo = len > minlen + 3 ? len - 3 : minlen;
            for (uint32_t l = len; l >= lo && l >= 3; --l) {
                const int li = len_idx(l);
                const uint32_t c = lprice[257 + li] + 16u * (uint32_t)len_eb(li) + dc + cost[i + l];
                if (c < best) { best = c; bs = l; }
            }
        }
        cost[i] = best;
        step[i] = (uint16_t)bs;
    }
    uint32_t i = 0;
    while (i < n) {
        const uint32_t w = m[p0 + i], s = step[i];
        tpos[nt] = p0 + i;
        tok[nt++] = s > 1 ? ((w & ~(0x1FFu << 8)) | (s << 8)) : (w & 0xFF);
        i += s;
    }
    return nt;
}

/* ---- the GPU's algorithm (golden model of encode.hip's strip parse): chunks of 64 strips x S positions; every strip is a
 * backward cost parse on its own (costs relative to its end; a target behind the end is priced by extrapolation: -avgq per
 * byte); candidates len, len-1 .. len-(ncand-1) (>= 3); prices in 1/4 bit from the tokens emitted so far in the piece
 * (first chunk: `first` = 0 static prices, 1 = statistics over all positions' words, 2 = parse it twice).  The forward walk
 * follows the real chain across strips and chunks. */
static const uint32_t* g_ltok = 0; static const uint32_t* g_lpos = 0; static int g_lnt = 0; static uint32_t g_lwin = 65536; static int g_quant = 4; static double g_smooth = 0.25; static int g_pmax = 15; static int g_half = 64; static int g_every = 1; static int g_stat0 = 0; static int g_minl = 3; static int g_lmax = 9999; static int g_force = 999; static int g_flat = 99; static int g_pen = 0; static int g_pen_short = 0;
static void static_prices(int32_t* lp, int32_t* dp) {
    for (int s = 0; s < 288; ++s) lp[s] = 4 * (s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
    for (int s = 0; s < 32; ++s) dp[s] = 4 * 5;
}
static void prices_q(const double* lf, const double* df, int32_t* lp, int32_t* dp) {
    double nl = 0, nd = 0;
    for (int s = 0; s < 288; ++s) nl += lf[s] + g_smooth;
    for (int s = 0; s < 32; ++s) nd += df[s] + g_smooth;
    double lf2[288]; for (int s = 0; s < 288; ++s) lf2[s] = lf[s];
    if (g_flat < 29) { double sum = 0; int cnt = 0; for (int s = 257 + g_flat; s < 286; ++s) { sum += lf[s]; cnt++; } for (int s = 257 + g_flat; s < 286; ++s) lf2[s] = sum / cnt; }
    for (int s = 0; s < 288; ++s) { double b = log2(nl / (lf2[s] + g_smooth)); if (b > g_pmax) b = g_pmax; if (b < 1) b = 1; lp[s] = (int32_t)(b * 4 + 0.5); }
    for (int s = 0; s < 32; ++s) { double b = log2(nd / (df[s] + g_smooth)); if (b > g_pmax) b = g_pmax; if (b < 1) b = 1; dp[s] = (int32_t)(b * 4 + 0.5); }
}
static void strip_dp(const uint32_t* m, uint32_t q0, uint32_t q1, const int32_t* lp, const int32_t* dp, int ncand, int32_t avgq, uint8_t* dec) {
    /* positions [q0, q1), q1 - q0 <= 1024 */
    int32_t cost[1025];
    const uint32_t S = q1 - q0;
    cost[S] = 0;
    for (int k = (int)S - 1; k >= 0; --k) {
        const uint32_t w = m[q0 + k];
        int32_t best = lp[w & 0xFF] + cost[k + 1];
        int d = 0;
        const uint32_t len = (w >> 8) & 0x1FF;
        if (len >= (uint32_t)g_force) best = 1 << 28;   /* a long match is never deferred */
        if (len >= 3) {
            const int di = dist_idx((w >> 17) + 1);
            const int32_t dc = dp[di] + 4 * dist_eb(di);
            for (int t = 0; t < ncand; ++t) {
                if (len < (uint32_t)g_minl + (uint32_t)t) break;
                const uint32_t l = len - (uint32_t)t;
                const int li = len_idx(l);
                const uint32_t tg = (uint32_t)k + l;
                int32_t ct;
                if ((int)l <= g_lmax) ct = tg <= S ? cost[tg] : -avgq * (int32_t)(tg - S);
                else {   /* outside the register window: extrapolate from the last cost inside it (or from the strip's end) */
                    const uint32_t ref = (uint32_t)k + (uint32_t)g_lmax;
                    const int32_t base = ref <= S ? cost[ref] : -avgq * (int32_t)(ref - S);
                    ct = base - avgq * (int32_t)(l - (uint32_t)g_lmax);
                }
                const int32_t c = lp[257 + li] + 4 * len_eb(li) + dc + ct + g_pen + (l <= 6 ? g_pen_short : 0);
                if (c < best) { best = c; d = 1 + t; }
            }
        }
        cost[k] = best;
        dec[q0 + k] = (uint8_t)d;
    }
}
void set_ext(const uint32_t* ltok, const uint32_t* lpos, int lnt, uint32_t win) { g_ltok = ltok; g_lpos = lpos; g_lnt = lnt; g_lwin = win; }
void set_price_model(double smooth, int pmax) { g_smooth = smooth; g_pmax = pmax; }
void set_flat(int k) { g_flat = k; }
void set_force(int k) { g_force = k; }
void set_lmax(int k) { g_lmax = k; }
void set_stat0(int k, int minl) { g_stat0 = k; g_minl = minl; }
void set_sample(int half, int every) { g_half = half; g_every = every; }
void set_pen(int pen, int pen_short) { g_pen = pen; g_pen_short = pen_short; }
int parse_strips(const uint32_t* m, uint32_t p0, uint32_t p1, uint32_t S, int ncand, int first, int decay, uint8_t* dec, uint32_t* tok, uint32_t* tpos, int nt) {
    double lf[288] = {0}, df[32] = {0};
    int32_t lp[288], dp[32];
    uint32_t e = p0;
    int32_t avgq = 12;
    const uint32_t CH = 64 * S;
    for (uint32_t c0 = p0; c0 < p1; c0 += CH) {
        const uint32_t c1 = c0 + CH < p1 ? c0 + CH : p1;
        int passes = 1;
        if (c0 == p0) {
            if (first == 0) static_prices(lp, dp);
            else if (first == 1 || (first >= 3 && first != 6)) {
                double l2[288] = {0}, d2[32] = {0};
                for (uint32_t i = c0; i < c1; ++i) {
                    const uint32_t len = (m[i] >> 8) & 0x1FF;
                    if (len < 3) l2[m[i] & 0xFF] += 1;
                    else { l2[257 + len_idx(len)] += 1.0; d2[dist_idx((m[i] >> 17) + 1)] += 1.0; }
                }
                prices_q(l2, d2, lp, dp);
            } else if (first == 6) static_prices(lp, dp); else { static_prices(lp, dp); passes = 2; }
        } else if (first == 6) { static_prices(lp, dp);
        } else if (first >= 3) {
            /* position statistics: every (sampled) position's literal or best match; first = 3: this chunk, all positions; 4: every
             * 4th segment of 64; 5: cumulative over the piece so far + this chunk, every 4th segment */
            static double l3[288], d3[32];
            if (first != 5 || c0 == p0) { memset(l3, 0, sizeof l3); memset(d3, 0, sizeof d3); }
            for (uint32_t i = c0; i < c1; ++i) {
                if (first >= 4 && ((i >> 6) & 3)) continue;
                const uint32_t len = (m[i] >> 8) & 0x1FF;
                if (len < 3) l3[m[i] & 0xFF] += 1;
                else { l3[257 + len_idx(len)] += 1.0; d3[dist_idx((m[i] >> 17) + 1)] += 1.0; }
            }
            prices_q(l3, d3, lp, dp);
        } else prices_q(lf, df, lp, dp);
        if (g_ltok && c0 > 0) {
            double l2[288] = {0}, d2[32] = {0};
            const uint32_t lo = c0 > g_lwin ? c0 - g_lwin : 0;
            for (int i = 0; i < g_lnt; ++i) {
                if (g_lpos[i] < lo || g_lpos[i] >= c0) continue;
                const uint32_t len = (g_ltok[i] >> 8) & 0x1FF;
                if (!len) l2[g_ltok[i] & 0xFF]++; else { l2[257 + len_idx(len)]++; d2[dist_idx((g_ltok[i] >> 17) + 1)]++; }
            }
            prices_q(l2, d2, lp, dp);
        }
        for (int pass = 0; pass < passes; ++pass) {
            for (uint32_t q0 = c0; q0 < c1; q0 += S) strip_dp(m, q0, q0 + S < c1 ? q0 + S : c1, lp, dp, ncand, avgq, dec);
            if (pass + 1 < passes) {   /* statistics of the trial parse price the real one */
                double l2[288] = {0}, d2[32] = {0};
                uint32_t i = e;
                while (i < c1) {
                    const uint32_t w = m[i];
                    if (dec[i] == 0) { l2[w & 0xFF]++; i += 1; }
                    else { uint32_t l = ((w >> 8) & 0x1FF) - (dec[i] - 1u); l2[257 + len_idx(l)]++; d2[dist_idx((w >> 17) + 1)]++; i += l; }
                }
                prices_q(l2, d2, lp, dp);
            }
        }
        if (decay) { for (int s = 0; s < 288; ++s) lf[s] *= 0.5; for (int s = 0; s < 32; ++s) df[s] *= 0.5; }
        uint32_t i = e;
        double bits = 0;
        while (i < c1) {
            const uint32_t w = m[i];
            uint32_t l = 1;
            if (dec[i]) {
                l = ((w >> 8) & 0x1FF) - (dec[i] - 1u);
                if (i + l > p1) { l = p1 - i; if (l < 3) l = 1; }
            }
            tpos[nt] = i;
            if (l > 1) {
                tok[nt++] = (w & ~(0x1FFu << 8)) | (l << 8);
                const int li = len_idx(l), di = dist_idx((w >> 17) + 1);
                if (!g_stat0) { lf[257 + li]++; df[di]++; }
                bits += (lp[257 + li] + dp[di]) / 4.0 + len_eb(li) + dist_eb(di);
            } else { tok[nt++] = w & 0xFF; if (!g_stat0) lf[w & 0xFF]++; bits += lp[w & 0xFF] / 4.0; }
            i += l;
        }
        if (g_stat0 && ((c0 - p0) / CH) % (uint32_t)g_every == 0) {   /* the separate parse kernel's statistics: every strip walked from its own first position */
            for (uint32_t q0 = c0; q0 < c1; q0 += S) {
                uint32_t q = q0; uint32_t q1 = q0 + S < c1 ? q0 + S : c1;
                if (q1 > q0 + (uint32_t)g_half) q1 = q0 + (uint32_t)g_half;
                while (q < q1) {
                    const uint32_t w = m[q];
                    uint32_t l = 1;
                    if (dec[q]) { l = ((w >> 8) & 0x1FF) - (dec[q] - 1u); if (q + l > p1) { l = p1 - q; if (l < 3) l = 1; } }
                    if (l > 1) { lf[257 + len_idx(l)]++; df[dist_idx((w >> 17) + 1)]++; } else lf[w & 0xFF]++;
                    q += l;
                }
            }
        }
        avgq = (int32_t)(4.0 * bits / (double)(c1 - c0) + 0.5);
        if (avgq < 1) avgq = 1;
        e = i;
    }
    return nt;
}

/* the three-deep lazy rule + a local truncation rule: a match gives up its last byte (its last two) when the match that starts
 * there is that much better than the one behind it */
int parse_lazy3_trunc(const uint32_t* m, uint32_t n, uint32_t max_lazy, uint32_t lazy2, uint32_t lazy3, int margin, int depth, uint32_t* tok, uint32_t* tpos) {
    uint32_t p = 0; int nt = 0;
#define LEN_AT(q) ((q) < n ? (m[(q)] >> 8) & 0x1FF : 0)
    while (p < n) {
        uint32_t l0 = LEN_AT(p), l1 = LEN_AT(p + 1), l2 = LEN_AT(p + 2), l3 = LEN_AT(p + 3);
        int defer = (l0 < max_lazy) && ((l1 > l0) || (l2 > l0 + lazy2) || (l3 > l0 + lazy3));
        uint32_t step = 1;
        if (l0 >= 4 && !defer) {
            step = l0;
            if (l0 < 258) {
                const uint32_t a = LEN_AT(p + l0);
                uint32_t a_eff = a >= 4 ? a : 1;
                for (int t = 1; t <= depth; ++t) {
                    if (l0 < 4u + (uint32_t)t) break;
                    const uint32_t b = LEN_AT(p + l0 - t);
                    if (b >= 4 && b >= a_eff + (uint32_t)t + (uint32_t)margin) { step = l0 - t; a_eff = b - t; }
                }
            }
        }
        if (p + step > n) { step = n - p; if (step < 3) step = 1; }
        tpos[nt] = p;
        tok[nt++] = step > 1 ? ((m[p] & ~(0x1FFu << 8)) | (step << 8)) : (m[p] & 0xFF);
        p += step;
    }
    return nt;
}
