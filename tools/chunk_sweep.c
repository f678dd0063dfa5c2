/* chunk_sweep.c -- the reference's inflate benchmark (test-libz-rs-sys/examples/blogpost-uncompress.rs:6-44, the sweep of
 * zlib_benchmarks.json: input chunks of 2^4 ... 2^24 bytes, the whole output buffer available, Z_NO_FLUSH) as a C driver that
 * binds ANY zlib-ABI library at run time: bench.py runs it once with libz_mi355.so and once with the system's libz, so both go
 * through the same loop with no interpreter in it.
 * usage: chunk_sweep LIB.so FILE.gz|.zz OUTPUT_BYTES WBITS CHUNK [CHUNK ...]   -> one line per chunk: "chunk seconds total_out rc polls fnv1a(output)"
 * The loop is the reference's: one inflate() per chunk, stop at Z_STREAM_END.  zlib allows an inflate() to take input without
 * producing its output yet; a caller that has handed over everything and has not seen Z_STREAM_END asks again with avail_in = 0
 * (`polls` counts those calls: 0 for an eager library). */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    const unsigned char* next_in; unsigned avail_in; unsigned long total_in;
    unsigned char* next_out; unsigned avail_out; unsigned long total_out;
    const char* msg; void* state; void* zalloc; void* zfree; void* opaque;
    int data_type; unsigned long adler; unsigned long reserved;
} zs;

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s LIB FILE OUT_BYTES WBITS CHUNK...\n", argv[0]); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    int (*init2)(zs*, int, const char*, int) = (int (*)(zs*, int, const char*, int))dlsym(h, "inflateInit2_");
    int (*inf)(zs*, int) = (int (*)(zs*, int))dlsym(h, "inflate");
    int (*end)(zs*) = (int (*)(zs*))dlsym(h, "inflateEnd");
    const char* (*ver)(void) = (const char* (*)(void))dlsym(h, "zlibVersion");
    if (!init2 || !inf || !end || !ver) { fprintf(stderr, "missing symbol\n"); return 2; }
    FILE* f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 2; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char* src = (unsigned char*)malloc((size_t)n + 1);
    if (fread(src, 1, (size_t)n, f) != (size_t)n) return 2;
    fclose(f);
    const size_t cap = (size_t)atoll(argv[3]) + 64;
    const int wbits = atoi(argv[4]);
    unsigned char* dst = (unsigned char*)malloc(cap);
    memset(dst, 0, cap);   /* touched before the clock runs */
    for (int a = 5; a < argc; ++a) {
        const size_t chunk = (size_t)atoll(argv[a]);
        zs s;
        memset(&s, 0, sizeof s);
        if (init2(&s, wbits, ver(), (int)sizeof s) != 0) { fprintf(stderr, "inflateInit2_ failed\n"); return 3; }
        s.next_out = dst;
        s.avail_out = (unsigned)cap;
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        int rc = 0, polls = 0;
        for (size_t pos = 0; pos < (size_t)n && rc == 0; pos += chunk) {
            s.next_in = src + pos;
            s.avail_in = (unsigned)((size_t)n - pos < chunk ? (size_t)n - pos : chunk);
            rc = inf(&s, 0);
        }
        while (rc == 0 && polls < 1000) { s.avail_in = 0; rc = inf(&s, 0); ++polls; }   /* everything is handed over: ask for the rest */
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const unsigned long out = s.total_out;
        end(&s);
        unsigned long long hsh = 1469598103934665603ull;   /* FNV-1a of what was produced: bench.py compares the two libraries' outputs */
        for (unsigned long i = 0; i < out && i < cap; ++i) { hsh ^= dst[i]; hsh *= 1099511628211ull; }
        printf("%zu %.6f %lu %d %d %016llx\n", chunk, (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec), out, rc, polls, hsh);
        fflush(stdout);
    }
    return 0;
}
