// checksum.hip -- Adler-32 and CRC-32 of every shard of a batch, one workgroup per shard.
//
// These are the RFC 1950 / RFC 1952 trailer checksums; the reference computes them in
// zlib-rs/src/adler32.rs:19-47 (BASE 65521, NMAX 5552) + adler32/generic.rs and
// zlib-rs/src/crc32.rs:19-29 + crc32/braid.rs (reflected polynomial 0xEDB88320), fused into
// fill_window / Window::extend.  On the GPU both are restructured around linearity instead of a
// serial fold:
//   Adler-32:  s1 = 1 + sum(b_i),  s2 = n + sum((n - i) * b_i)   (mod 65521)
//              -> every thread takes 16-byte stripes (coalesced), no sequential dependency.
//   CRC-32:    thread t owns one contiguous segment, computes the zero-init raw CRC with a
//              slice-by-16 table in LDS, multiplies it by x^(8 * bytes_after_segment) mod P
//              (the crc32_combine algebra of crc32/combine.rs:26-61 applied per thread) and the
//              workgroup XOR-reduces.  Pre/post conditioning is folded in at the end.
// HBM-bound by construction: 1 byte read per input byte, 4 bytes written per shard.
#include "zmi_device.h"

#define ZMI_ADLER_BASE 65521u

// sum of the four bytes of w, plus acc / sum of byte_i(w) * byte_i(k), plus acc
#ifdef ZMI_EMU
static inline uint32_t zmi_sum4(uint32_t w, uint32_t acc) { return acc + (w & 0xFFu) + ((w >> 8) & 0xFFu) + ((w >> 16) & 0xFFu) + (w >> 24); }
static inline uint32_t zmi_dot4(uint32_t w, uint32_t k, uint32_t acc) {
    for (int i = 0; i < 4; ++i) acc += ((w >> (8 * i)) & 0xFFu) * ((k >> (8 * i)) & 0xFFu);
    return acc;
}
#else
static __device__ __forceinline__ uint32_t zmi_sum4(uint32_t w, uint32_t acc) { return __builtin_amdgcn_sad_u8(w, 0u, acc); }
static __device__ __forceinline__ uint32_t zmi_dot4(uint32_t w, uint32_t k, uint32_t acc) { return __builtin_amdgcn_udot4(w, k, acc, false); }
#endif
#define ZMI_CRC_POLY 0xEDB88320u

// a(x) * b(x) mod P, reflected bit order (bit 31 = x^0)
static __device__ __forceinline__ uint32_t zmi_gf2_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 31; i >= 0; --i) {
        if ((a >> i) & 1u) p ^= b;
        b = (b >> 1) ^ ((b & 1u) ? ZMI_CRC_POLY : 0u);
    }
    return p;
}
// x^(8*nbytes) mod P
static __device__ __forceinline__ uint32_t zmi_gf2_xpow8(uint64_t nbytes) {
    uint32_t r = 0x80000000u;   // x^0
    uint32_t pw = 0x00800000u;  // x^8
    while (nbytes) {
        if (nbytes & 1u) r = zmi_gf2_mulmod(r, pw);
        pw = zmi_gf2_mulmod(pw, pw);
        nbytes >>= 1;
    }
    return r;
}

// the same algebra at compile time, for the two constants of the block scheme below
constexpr uint32_t zmi_cx_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 31; i >= 0; --i) {
        if ((a >> i) & 1u) p ^= b;
        b = (b >> 1) ^ ((b & 1u) ? ZMI_CRC_POLY : 0u);
    }
    return p;
}
constexpr uint32_t zmi_cx_xpow8(uint64_t nbytes) {
    uint32_t r = 0x80000000u, pw = 0x00800000u;
    while (nbytes) {
        if (nbytes & 1u) r = zmi_cx_mulmod(r, pw);
        pw = zmi_cx_mulmod(pw, pw);
        nbytes >>= 1;
    }
    return r;
}
#define CRC_CHUNK 64u                       // bytes a thread takes from a block
#define CRC_BLOCK (256u * CRC_CHUNK)        // bytes the workgroup takes per step
constexpr uint32_t kCrcSkipBlock = zmi_cx_xpow8(CRC_BLOCK - CRC_CHUNK);   // x^(8 * (block - chunk)) mod P
constexpr uint32_t kCrcSkipChunk = zmi_cx_xpow8(CRC_CHUNK);               // x^(8 * chunk) mod P

// v * K mod P through four byte tables of K (multiplication by a constant is linear in v)
static __device__ __forceinline__ uint32_t zmi_crc_mulk(const uint32_t (*kt)[256], uint32_t v) {
    return kt[0][v & 0xFFu] ^ kt[1][(v >> 8) & 0xFFu] ^ kt[2][(v >> 16) & 0xFFu] ^ kt[3][v >> 24];
}

// kind: bit0 = adler32 wanted, bit1 = crc32 wanted.  out_adler/out_crc indexed by shard.
__global__ void __launch_bounds__(256) zmi_checksum_kernel(const uint8_t* __restrict__ data,
                                                           const uint64_t* __restrict__ off,
                                                           const uint32_t* __restrict__ len, uint32_t kind,
                                                           uint32_t* __restrict__ out_adler,
                                                           uint32_t* __restrict__ out_crc) {
    __shared__ uint32_t tab[16][256];
    __shared__ uint32_t kblk[4][256], kchk[4][256], fold[256];
    __shared__ uint32_t red[3][4];
    const uint32_t s = blockIdx.x;
    const uint32_t t = threadIdx.x;
    const uint8_t* src = data + off[s];
    const uint32_t n = len[s];
    const bool aligned = (((uintptr_t)src) & 15u) == 0;

    if (kind & 1u) {
        uint32_t A = 0;
        uint64_t B = 0;   // the weight of byte i is n - i: (n - i0) * (sum of the stripe) - sum(j * b_j) per 16-byte stripe;
                          // a thread's share stays below n * n / 256 * 255 < 2^64, so nothing is reduced inside the loop
        uint32_t i0 = t * 16u;
        // four stripes (four independent 16-byte loads) per trip; byte sums are v_sad_u8 against zero, the weighted sums
        // v_dot4_u32_u8 against the byte offsets: 2 instructions per 4 bytes
        for (; aligned && (uint64_t)i0 + 3u * 4096u + 16u <= n; i0 += 4u * 4096u) {
            uint4 q[4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) q[k] = *(const uint4*)(src + i0 + k * 4096u);
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                const uint32_t sb = zmi_sum4(q[k].x, zmi_sum4(q[k].y, zmi_sum4(q[k].z, zmi_sum4(q[k].w, 0u))));
                const uint32_t sjb = zmi_dot4(q[k].x, 0x03020100u, zmi_dot4(q[k].y, 0x07060504u,
                                     zmi_dot4(q[k].z, 0x0B0A0908u, zmi_dot4(q[k].w, 0x0F0E0D0Cu, 0u))));
                A += sb;
                B += (uint64_t)(n - i0 - k * 4096u) * sb - sjb;
            }
        }
        for (; i0 < n; i0 += 256u * 16u) {
            uint32_t nv = n - i0;
            zmi_b16 v = zmi_ld16(src + i0, nv, aligned);
            uint32_t sb = 0, sjb = 0;
#pragma unroll
            for (uint32_t j = 0; j < 16u; ++j) {
                uint32_t b = (v.w[j >> 2] >> (8u * (j & 3u))) & 0xFFu;
                sb += b;
                sjb += j * b;
            }
            A += sb;
            // bytes past n were read as zero so they add nothing
            B += (uint64_t)(n - i0) * sb - sjb;
        }
        uint32_t a = A % ZMI_ADLER_BASE;
        uint32_t b = (uint32_t)(B % ZMI_ADLER_BASE);
        a = zmi_wave_sum(a);
        b = zmi_wave_sum(b);
        if (zmi_lane() == 0) { red[0][zmi_wave()] = a % ZMI_ADLER_BASE; red[1][zmi_wave()] = b % ZMI_ADLER_BASE; }
    }
    if (kind & 2u) {
        // slice-by-16 tables: tab[k][b] = CRC of byte b followed by k zero bytes.  One 16-byte step is 16 independent
        // lookups (only the four of the first word wait for the running CRC): one LDS round trip per 16 bytes instead of
        // the four of slice-by-4
        uint32_t c = t;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? ZMI_CRC_POLY : 0u);
        tab[0][t] = c;
        __syncthreads();
        for (uint32_t k = 1; k < 16u; ++k) {
            c = (c >> 8) ^ tab[0][c & 0xFFu];
            tab[k][t] = c;
        }
        __syncthreads();
        // Full blocks of 16 KiB: thread t takes the 64 bytes at t * 64 of every block (neighbouring lanes read
        // neighbouring memory; one contiguous 4 KiB segment per thread made every load of a wave touch 64 different
        // lines).  Between two blocks a thread's running value skips the 16 KiB - 64 bytes of the others: a multiplication
        // by the constant x^(8 * 16320), four table reads.  At the end thread 0 folds the 256 values in order
        // (value * x^(8 * 64) + next).
        const uint32_t nfull = aligned ? n / CRC_BLOCK : 0u;
        uint32_t full_raw = 0;
        if (nfull) {
#pragma unroll
            for (uint32_t i = 0; i < 4u; ++i) {
                kblk[i][t] = zmi_gf2_mulmod(t << (8u * i), kCrcSkipBlock);
                kchk[i][t] = zmi_gf2_mulmod(t << (8u * i), kCrcSkipChunk);
            }
            __syncthreads();
            uint32_t acc = 0;
            const uint8_t* pch = src + t * CRC_CHUNK;
            for (uint32_t blk = 0; blk < nfull; ++blk, pch += CRC_BLOCK) {
                uint4 q[4];
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) q[k] = *(const uint4*)(pch + 16u * k);
                if (blk) acc = zmi_crc_mulk(kblk, acc);
#pragma unroll
                for (uint32_t k = 0; k < 4u; ++k) {
                    const uint32_t x = acc ^ q[k].x;
                    acc = tab[15][x & 0xFFu] ^ tab[14][(x >> 8) & 0xFFu] ^ tab[13][(x >> 16) & 0xFFu] ^ tab[12][x >> 24] ^
                          tab[11][q[k].y & 0xFFu] ^ tab[10][(q[k].y >> 8) & 0xFFu] ^ tab[9][(q[k].y >> 16) & 0xFFu] ^ tab[8][q[k].y >> 24] ^
                          tab[7][q[k].z & 0xFFu] ^ tab[6][(q[k].z >> 8) & 0xFFu] ^ tab[5][(q[k].z >> 16) & 0xFFu] ^ tab[4][q[k].z >> 24] ^
                          tab[3][q[k].w & 0xFFu] ^ tab[2][(q[k].w >> 8) & 0xFFu] ^ tab[1][(q[k].w >> 16) & 0xFFu] ^ tab[0][q[k].w >> 24];
                }
            }
            fold[t] = acc;
            __syncthreads();
            if (t == 0) {
                uint32_t tot = 0;
                for (uint32_t i = 0; i < 256u; ++i) tot = zmi_crc_mulk(kchk, tot) ^ fold[i];
                full_raw = tot;
            }
        }
        // the rest (everything, for an unaligned shard): one contiguous segment per thread, length multiple of 16
        const uint32_t n0 = nfull * CRC_BLOCK, nt = n - n0;
        const uint8_t* tsrc = src + n0;
        uint32_t seg = ((nt + 255u) / 256u + 15u) & ~15u;
        uint64_t beg = (uint64_t)t * seg;
        uint32_t crc = 0;
        uint64_t after = 0;
        if (beg < nt) {
            uint32_t end = (beg + seg < nt) ? (uint32_t)(beg + seg) : nt;
            after = nt - end;
            for (uint32_t i0 = (uint32_t)beg; i0 < end; i0 += 16u) {
                uint32_t nv = end - i0;
                zmi_b16 v = zmi_ld16(tsrc + i0, nv, aligned);
                if (nv >= 16u) {
                    const uint32_t x = crc ^ v.w[0];
                    uint32_t r = tab[15][x & 0xFFu] ^ tab[14][(x >> 8) & 0xFFu] ^ tab[13][(x >> 16) & 0xFFu] ^ tab[12][x >> 24];
#pragma unroll
                    for (int q = 1; q < 4; ++q) {
                        const uint32_t y = v.w[q];
                        r ^= tab[15 - 4 * q][y & 0xFFu] ^ tab[14 - 4 * q][(y >> 8) & 0xFFu] ^ tab[13 - 4 * q][(y >> 16) & 0xFFu] ^
                             tab[12 - 4 * q][y >> 24];
                    }
                    crc = r;
                } else {
                    for (uint32_t j = 0; j < nv; ++j) {
                        uint32_t b = (v.w[j >> 2] >> (8u * (j & 3u))) & 0xFFu;
                        crc = tab[0][(crc ^ b) & 0xFFu] ^ (crc >> 8);
                    }
                }
            }
            crc = zmi_gf2_mulmod(crc, zmi_gf2_xpow8(after));
        }
        if (t == 0 && nfull) crc ^= zmi_gf2_mulmod(full_raw, zmi_gf2_xpow8(nt));   // the full blocks, moved in front of the rest
        // xor-reduce
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) crc ^= __shfl_xor(crc, d);
        if (zmi_lane() == 0) red[2][zmi_wave()] = crc;
    }
    __syncthreads();
    if (t == 0) {
        if (kind & 1u) {
            uint32_t a = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) % ZMI_ADLER_BASE;
            uint32_t b = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) % ZMI_ADLER_BASE;
            uint32_t s1 = (1u + a) % ZMI_ADLER_BASE;
            uint32_t s2 = (n % ZMI_ADLER_BASE + b) % ZMI_ADLER_BASE;
            out_adler[s] = (s2 << 16) | s1;
        }
        if (kind & 2u) {
            uint32_t raw = red[2][0] ^ red[2][1] ^ red[2][2] ^ red[2][3];
            // init 0xFFFFFFFF travels through n bytes, then final inversion
            uint32_t init = zmi_gf2_mulmod(0xFFFFFFFFu, zmi_gf2_xpow8(n));
            out_crc[s] = raw ^ init ^ 0xFFFFFFFFu;
        }
    }
}

extern "C" int zmi_launch_checksum(const uint8_t* d_data, const uint64_t* d_off, const uint32_t* d_len,
                                   uint32_t n_shards, uint32_t kind, uint32_t* d_adler, uint32_t* d_crc,
                                   hipStream_t stream) {
    if (n_shards == 0 || (kind & 3u) == 0) return 0;
    ZMI_LAUNCH(zmi_checksum_kernel, dim3(n_shards), dim3(256), 0, stream, d_data, d_off, d_len, kind, d_adler, d_crc);
    return 0;
}
