/* comm_host.c -- the multi-GPU exchange step from a compiled host, no Python and no torch: what the Zig host of
 * INTEGRATION.md does with zh_comm_*.  `comm_host N` forks N rank processes (rank r on GPU r; the parent never
 * touches a GPU), rank 0 makes the RCCL id and the parent relays its 128 bytes to the other ranks over pipes (the
 * "any host channel" of include/zang_hip.h), every rank fills a [48][2][1024] block with values that depend on its
 * rank, runs zh_allreduce_mix and zh_reduce_mix on its context's stream and checks the sums (small integers: exact
 * in any order).  Exit code 0 and "PASS" when every rank agrees. */
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>
#include "zang_hip.h"

enum { BUFFERS = 48, CHANNELS = 2, FRAMES = 1024, N = BUFFERS * CHANNELS * FRAMES };

#define CHECK(expr)                                                                                     \
    do {                                                                                                \
        int _rc = (expr);                                                                               \
        if (_rc != ZH_OK) {                                                                             \
            fprintf(stderr, "rank %d: %s -> %d (%s) %s\n", rank, #expr, _rc, zh_error_string(_rc), zh_comm_last_error()); \
            return 1;                                                                                   \
        }                                                                                               \
    } while (0)

static int read_all(int fd, void *p, size_t n) {
    uint8_t *b = (uint8_t *)p;
    while (n) { ssize_t k = read(fd, b, n); if (k <= 0) return -1; b += k; n -= (size_t)k; }
    return 0;
}
static int write_all(int fd, const void *p, size_t n) {
    const uint8_t *b = (const uint8_t *)p;
    while (n) { ssize_t k = write(fd, b, n); if (k <= 0) return -1; b += k; n -= (size_t)k; }
    return 0;
}

static int run_rank(int rank, int world, int id_in, int id_out) {
    uint8_t id[ZH_COMM_ID_BYTES];
    zh_ctx *ctx = NULL;
    /* COMM_HOST_ONE_DEVICE=1: every rank on device 0 (an experiment: RCCL refuses communicators with a duplicate GPU) */
    CHECK(zh_create(&ctx, getenv("COMM_HOST_ONE_DEVICE") ? 0 : rank));
    if (!zh_comm_available()) { fprintf(stderr, "rank %d: librccl: %s\n", rank, zh_comm_last_error()); return 1; }
    if (rank == 0) {
        CHECK(zh_comm_unique_id(id));
        if (write_all(id_out, id, sizeof id)) return 1;
    } else if (read_all(id_in, id, sizeof id)) return 1;
    zh_comm *comm = NULL;
    CHECK(zh_comm_create(ctx, (uint32_t)world, (uint32_t)rank, id, &comm));
    float *host = (float *)malloc(N * sizeof(float)), *dev = NULL;
    CHECK(zh_malloc(ctx, (void **)&dev, N * sizeof(float)));
    int bad = 0;
    for (int pass = 0; pass < 2; pass++) {            /* 0: all-reduce, 1: reduce to the last rank */
        for (int i = 0; i < N; i++) host[i] = (float)((i % 251) * (rank + 1));
        CHECK(zh_upload(ctx, dev, host, N * sizeof(float)));
        const uint32_t root = (uint32_t)(world - 1);
        if (pass == 0) CHECK(zh_allreduce_mix(comm, dev, N));
        else CHECK(zh_reduce_mix(comm, dev, N, root));
        CHECK(zh_sync(ctx));
        CHECK(zh_download(ctx, host, dev, N * sizeof(float)));
        if (pass == 0 || (uint32_t)rank == root)
            for (int i = 0; i < N; i++) bad += host[i] != (float)((i % 251) * (world * (world + 1) / 2));
    }
    if (rank == 0) printf("rccl %d from %s, world %d\n", zh_comm_version(), zh_comm_library(), world);
    fflush(stdout);                                   /* the rank leaves through _exit */
    CHECK(zh_comm_destroy(comm));
    CHECK(zh_free(ctx, dev));
    CHECK(zh_destroy(ctx));
    free(host);
    if (bad) fprintf(stderr, "rank %d: %d wrong sums\n", rank, bad);
    return bad ? 1 : 0;
}

int main(int argc, char **argv) {
    const int world = argc > 1 ? atoi(argv[1]) : 1;
    if (world < 1 || world > 64) { fprintf(stderr, "usage: comm_host [ranks]\n"); return 2; }
    signal(SIGPIPE, SIG_IGN);                         /* a dead rank's pipe is an error return, not the parent's death */
    int up[2], down[64][2];
    pid_t pid[64];
    if (pipe(up)) return 2;
    for (int r = 1; r < world; r++) if (pipe(down[r])) return 2;
    for (int r = 0; r < world; r++) {
        pid[r] = fork();
        if (pid[r] < 0) return 2;
        if (pid[r] == 0) {
            /* a rank keeps only its own ends: the id pipe's write end lives in rank 0 alone, so that a rank 0 that dies before
             * writing the id is an EOF for the parent, and a parent that gives up is an EOF for the waiting ranks */
            close(up[0]);
            if (r != 0) close(up[1]);
            for (int q = 1; q < world; q++) { close(down[q][1]); if (q != r) close(down[q][0]); }
            _exit(run_rank(r, world, r ? down[r][0] : -1, r ? -1 : up[1]));
        }
    }
    close(up[1]);                                     /*  the parent holds no write end of `up`, no read end of `down` */
    for (int r = 1; r < world; r++) close(down[r][0]);
    uint8_t id[ZH_COMM_ID_BYTES];
    int rc = read_all(up[0], id, sizeof id) ? 1 : 0;
    for (int r = 1; r < world && !rc; r++) rc |= write_all(down[r][1], id, sizeof id) ? 1 : 0;
    for (int r = 1; r < world; r++) close(down[r][1]);            /* no id to hand on (rc != 0): the ranks read EOF and leave */
    for (int r = 0; r < world; r++) {
        int st = 0;
        waitpid(pid[r], &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st)) rc = 1;
    }
    puts(rc ? "FAIL" : "PASS");
    return rc;
}
