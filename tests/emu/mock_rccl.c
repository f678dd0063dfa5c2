/* mock_rccl.c -- TEST INFRASTRUCTURE ONLY: the dozen RCCL entry points csrc/exchange.hip binds, implemented for several
 * PROCESSES on one host that pass the bytes through files under $ZMI_MOCK_RCCL_DIR.  It lets the CPU suite run the real
 * zmi_exchange_* code (emulator build: "device" pointers are host pointers) with world_size 2 and 3 -- the group semantics
 * are the ones that matter: every operation between ncclGroupStart and ncclGroupEnd is posted before any of them is waited
 * for (sends first, then the receives), so the exchange pattern of exchange.hip (one send + one receive per peer per
 * round) cannot deadlock here unless it would on RCCL too.  Never part of the product library. */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef struct mock_comm {
    char dir[512];
    int world, rank;
    uint64_t sent[64], rcvd[64];   /* messages exchanged with every peer, in order */
} mock_comm;
typedef mock_comm* ncclComm_t;

enum { OP_SEND, OP_RECV };
typedef struct { int kind; void* buf; size_t bytes; int peer; mock_comm* c; } mock_op;
static __thread mock_op g_ops[4096];
static __thread int g_nops = 0, g_depth = 0;

static size_t dt_size(int dt) {
    switch (dt) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: case 9: return 2; default: return 1; }
}

static int do_send(mock_op* o) {
    char tmp[700], fin[700];
    snprintf(tmp, sizeof tmp, "%s/t_%d_%d_%llu", o->c->dir, o->c->rank, o->peer, (unsigned long long)o->c->sent[o->peer]);
    snprintf(fin, sizeof fin, "%s/m_%d_%d_%llu", o->c->dir, o->c->rank, o->peer, (unsigned long long)o->c->sent[o->peer]);
    o->c->sent[o->peer]++;
    FILE* f = fopen(tmp, "wb");
    if (!f) return 2;
    if (o->bytes && fwrite(o->buf, 1, o->bytes, f) != o->bytes) { fclose(f); return 2; }
    fclose(f);
    return rename(tmp, fin) == 0 ? 0 : 2;
}

static int do_recv(mock_op* o) {
    char fin[700];
    snprintf(fin, sizeof fin, "%s/m_%d_%d_%llu", o->c->dir, o->peer, o->c->rank, (unsigned long long)o->c->rcvd[o->peer]);
    o->c->rcvd[o->peer]++;
    struct timespec ts = {0, 2000000};
    for (int spin = 0; spin < 30000; ++spin) {   /* 60 s */
        FILE* f = fopen(fin, "rb");
        if (f) {
            size_t got = o->bytes ? fread(o->buf, 1, o->bytes, f) : 0;
            int extra = fgetc(f) != EOF;
            fclose(f);
            unlink(fin);
            return (got == o->bytes && !extra) ? 0 : 5;   /* size mismatch between a send and its receive */
        }
        nanosleep(&ts, NULL);
    }
    return 6;
}

/* a send is complete when its receiver has taken it (the file is gone): a send nobody receives blocks its group, as on RCCL,
 * and fails the test after the time-out instead of passing silently */
static int wait_taken(mock_op* o, unsigned long long seq) {
    char fin[700];
    snprintf(fin, sizeof fin, "%s/m_%d_%d_%llu", o->c->dir, o->c->rank, o->peer, seq);
    struct timespec ts = {0, 2000000};
    for (int spin = 0; spin < 30000; ++spin) {   /* 60 s */
        if (access(fin, F_OK) != 0) return 0;
        nanosleep(&ts, NULL);
    }
    return 7;
}

static int run_ops(void) {
    int rc = 0;
    unsigned long long seq[4096];
    for (int i = 0; i < g_nops && !rc; ++i) if (g_ops[i].kind == OP_SEND) { seq[i] = g_ops[i].c->sent[g_ops[i].peer]; rc = do_send(&g_ops[i]); }
    for (int i = 0; i < g_nops && !rc; ++i) if (g_ops[i].kind == OP_RECV) rc = do_recv(&g_ops[i]);
    for (int i = 0; i < g_nops && !rc; ++i) if (g_ops[i].kind == OP_SEND) rc = wait_taken(&g_ops[i], seq[i]);
    g_nops = 0;
    return rc;
}

static int post(int kind, void* buf, size_t bytes, int peer, mock_comm* c) {
    if (!c || peer < 0 || peer >= c->world || peer == c->rank || g_nops >= 4096) return 4;
    mock_op o = {kind, buf, bytes, peer, c};
    g_ops[g_nops++] = o;
    return g_depth ? 0 : run_ops();
}

int ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "mock-%ld-%ld", (long)getpid(), (long)time(NULL));
    return 0;
}
int ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    const char* base = getenv("ZMI_MOCK_RCCL_DIR");
    if (!base || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return 4;
    mock_comm* c = (mock_comm*)calloc(1, sizeof *c);
    id.internal[127] = 0;
    snprintf(c->dir, sizeof c->dir, "%s/%s", base, id.internal);
    mkdir(c->dir, 0700);
    c->world = nranks; c->rank = rank;
    *comm = c;
    return 0;
}
int ncclCommDestroy(ncclComm_t c) { free(c); return 0; }
int ncclCommAbort(ncclComm_t c) { free(c); return 0; }
int ncclCommCount(ncclComm_t c, int* n) { *n = c->world; return 0; }
int ncclCommUserRank(ncclComm_t c, int* r) { *r = c->rank; return 0; }
const char* ncclGetErrorString(int r) {
    switch (r) { case 0: return "ok"; case 2: return "mock: file error"; case 4: return "mock: invalid argument";
                 case 5: return "mock: send / receive size mismatch"; case 6: return "mock: receive timed out"; case 7: return "mock: a send was never received"; default: return "mock: error"; }
}
int ncclGroupStart(void) { g_depth++; return 0; }
int ncclGroupEnd(void) {
    if (g_depth <= 0) return 4;
    return --g_depth == 0 ? run_ops() : 0;
}
int ncclSend(const void* buf, size_t count, int dt, int peer, ncclComm_t c, void* stream) {
    (void)stream;
    return post(OP_SEND, (void*)buf, count * dt_size(dt), peer, c);
}
int ncclRecv(void* buf, size_t count, int dt, int peer, ncclComm_t c, void* stream) {
    (void)stream;
    return post(OP_RECV, buf, count * dt_size(dt), peer, c);
}
int ncclAllGather(const void* send, void* recv, size_t count, int dt, ncclComm_t c, void* stream) {
    (void)stream;
    const size_t bytes = count * dt_size(dt);
    memmove((char*)recv + (size_t)c->rank * bytes, send, bytes);
    int rc = ncclGroupStart();
    for (int p = 0; p < c->world && !rc; ++p) {
        if (p == c->rank) continue;
        rc = post(OP_SEND, (char*)recv + (size_t)c->rank * bytes, bytes, p, c);
        if (!rc) rc = post(OP_RECV, (char*)recv + (size_t)p * bytes, bytes, p, c);
    }
    int re = ncclGroupEnd();
    return rc ? rc : re;
}
