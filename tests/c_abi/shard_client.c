/* Plain-C MULTI-RANK client of include/brie_amd.h (SURVEY 8b threading row, 8e): one process per GPU, no Python, no torch.
 *
 *   shard_client <rank> <world> <id_file> <n_devices>
 *
 * Every rank is a fresh process that touches no GPU before it is told its device (rank % n_devices).  Rank 0 makes the RCCL
 * unique id (brie_comm_unique_id) and hands it over through <id_file> (written under a temporary name, then renamed); the
 * others wait for the file.  Then, per rank: brie_comm_init -> brie_create for the rank's contiguous gene block (gene_offset =
 * its first gene, boundaries on multiples of 4 = the noise stream's quads) -> upload of the block's COLUMNS of the caller's
 * (Nc, Ng) matrices (ld = Ng, no repacking on the host) -> the reference's six learning-rate stages with a fresh Adam each
 * (model_TFProb.py:234-241) -> brie_loss_gene -> brie_comm_allgather of [Wc_loc rows, intercept, sigma, loss_gene], the
 * "RCCL weight all-gather" of BASELINE configs[3], which replaces the reference's concatenation over its sequential gene
 * batches (model_wrap.py:241-260: same order, gene blocks in rank order).  Rank 0 then fits ALL genes on one handle and
 * demands the gathered vectors bit for bit (a gene's trajectory does not depend on the shard it is fitted in).
 *
 * Build:  gcc -std=c99 tests/c_abi/shard_client.c -Iinclude -Lbrie_amd/lib -lbrie_amd -lm -Wl,-rpath,$PWD/brie_amd/lib -o shard_client
 * Run:    bash tests/c_abi/run_shards.sh <world>          (tests/test_c_abi.py: world 1 on any GPU box, the node's world beyond)
 */
#define _DEFAULT_SOURCE          /* usleep */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "brie_amd.h"

#define CHECK(call)                                                                         \
    do {                                                                                    \
        int rc_ = (call);                                                                   \
        if (rc_ != BRIE_OK) {                                                               \
            fprintf(stderr, "rank %d: %s -> %d: %s\n", rank, #call, rc_, brie_last_error()); \
            return 1;                                                                       \
        }                                                                                   \
    } while (0)

enum { NC = 160, NG = 1000, KC = 2, STAGE_STEPS = 10, LOSS_REPEATS = 5, ROWS = KC + 3 };
static const float LRS[6] = {0.001f, 0.005f, 0.01f, 0.02f, 0.01f, 0.005f};      /* model_TFProb.py:234 */

static float c1[NC * NG], c2[NC * NG], xc[NC * KC];

/* first gene of rank r's block: whole quads, the remainder spread over the first ranks */
static int block_start(int r, int world) {
    const int quads = (NG + 3) / 4, q = quads / world, rem = quads % world;
    const int g = 4 * (r * q + (r < rem ? r : rem));
    return g < NG ? g : NG;
}

/* fit genes [g0, g1) on `device`; out[ROWS][ng]: Wc_loc rows, intercept, sigma, loss_gene */
static int fit_block(int rank, int device, int g0, int g1, float *out) {
    const int ng = g1 - g0;
    brie_problem p;
    memset(&p, 0, sizeof p);
    p.abi_version = BRIE_AMD_ABI_VERSION;
    p.device = device;
    p.Nc = NC; p.Ng = ng; p.gene_offset = g0; p.Kc = KC; p.n_layers = 2;
    p.train_intercept = 1; p.train_sigma = 1; p.seed = 20240617;
    brie_handle *h = NULL;
    CHECK(brie_create(&p, &h));
    CHECK(brie_upload(h, BRIE_COUNT1, c1 + g0, NC, ng, NG));       /* the block's columns of the (Nc, Ng) matrix */
    CHECK(brie_upload(h, BRIE_COUNT2, c2 + g0, NC, ng, NG));
    CHECK(brie_add_pseudo_count(h, 0.01f));                         /* model_wrap.py:113-117 */
    CHECK(brie_upload(h, BRIE_XC, xc, NC, KC, KC));
    CHECK(brie_init_state(h, NAN, NAN));
    for (int s = 0; s < 6; ++s) {                                   /* model_TFProb.py:235-241 */
        CHECK(brie_reset_optimizer(h));
        CHECK(brie_step(h, STAGE_STEPS, LRS[s], 1, NULL));
    }
    CHECK(brie_loss_gene(h, LOSS_REPEATS, out + (size_t)(KC + 2) * ng));
    CHECK(brie_read(h, BRIE_WC_LOC, out, KC, ng, ng));
    CHECK(brie_read(h, BRIE_INTERCEPT, out + (size_t)KC * ng, 1, ng, ng));
    CHECK(brie_read(h, BRIE_SIGMA, out + (size_t)(KC + 1) * ng, 1, ng, ng));
    CHECK(brie_destroy(h));
    return 0;
}

int main(int argc, char **argv) {
    if (argc != 5) { fprintf(stderr, "usage: %s <rank> <world> <id_file> <n_devices>\n", argv[0]); return 2; }
    const int rank = atoi(argv[1]), world = atoi(argv[2]), n_dev = atoi(argv[4]);
    const char *id_file = argv[3];
    if (world < 1 || rank < 0 || rank >= world || n_dev < 1) { fprintf(stderr, "bad rank / world / n_devices\n"); return 2; }
    const int device = rank % n_dev;
    if (brie_abi_version() != BRIE_AMD_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

    /* the caller's data: every rank holds the same (Nc, Ng) matrices (a real caller would read its block from disk) */
    unsigned long long s = 88172645463325252ULL;
    for (int i = 0; i < NC * NG; ++i) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        c1[i] = (float)((s >> 33) % 6);
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        c2[i] = (float)((s >> 33) % 4);
    }
    for (int i = 0; i < NC; ++i) { xc[i * KC] = (float)(i & 1); xc[i * KC + 1] = (float)cos(0.11 * i); }

    /* 1. the communicator: every rank checks RCCL alone, rank 0 publishes the id, everybody joins */
    CHECK(brie_comm_available(device));
    uint8_t id[BRIE_COMM_ID_BYTES];
    if (rank == 0) {
        char tmp[4096];
        CHECK(brie_comm_unique_id(id));
        snprintf(tmp, sizeof tmp, "%s.tmp", id_file);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || fclose(f) != 0 || rename(tmp, id_file) != 0) {
            fprintf(stderr, "rank 0: cannot write %s\n", id_file);
            return 1;
        }
    } else {
        FILE *f = NULL;
        for (int tries = 0; tries < 600 && !f; ++tries) {           /* up to 60 s */
            f = fopen(id_file, "rb");
            if (!f) usleep(100000);
        }
        if (!f || fread(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "rank %d: no unique id in %s\n", rank, id_file); return 1; }
        fclose(f);
    }
    brie_comm *comm = NULL;
    CHECK(brie_comm_init(device, rank, world, id, &comm));
    if (brie_comm_rank(comm) != rank || brie_comm_world(comm) != world) { fprintf(stderr, "rank %d: communicator disagrees\n", rank); return 1; }

    /* 2. the rank's gene block */
    const int g0 = block_start(rank, world), g1 = block_start(rank + 1, world), ng = g1 - g0;
    int ng_max = 0;
    for (int r = 0; r < world; ++r) {
        const int n = block_start(r + 1, world) - block_start(r, world);
        if (n > ng_max) ng_max = n;
    }
    if (ng <= 0) { fprintf(stderr, "rank %d: empty gene block (world %d > %d quads)\n", rank, world, (NG + 3) / 4); return 1; }
    float *mine = (float *)calloc((size_t)ROWS * ng, sizeof(float));
    float *send = (float *)calloc((size_t)ROWS * ng_max, sizeof(float));           /* equal counts per rank: padded */
    float *recv = (float *)calloc((size_t)world * ROWS * ng_max, sizeof(float));
    if (!mine || !send || !recv) return 1;
    if (fit_block(rank, device, g0, g1, mine)) return 1;
    for (int k = 0; k < ROWS; ++k) memcpy(send + (size_t)k * ng_max, mine + (size_t)k * ng, sizeof(float) * (size_t)ng);

    /* 3. the end-of-fit gather (replaces model_wrap.py:260's concatenation) */
    CHECK(brie_comm_allgather(comm, send, (int64_t)ROWS * ng_max, recv));
    double n_fit = (double)ng;
    CHECK(brie_comm_allreduce(comm, &n_fit, 1, BRIE_F64, BRIE_SUM));
    if ((int)n_fit != NG) { fprintf(stderr, "rank %d: the blocks hold %g genes, not %d\n", rank, n_fit, NG); return 1; }

    /* 4. rank 0: the same genes on ONE handle, bit for bit */
    int bad = 0;
    if (rank == 0) {
        float *full = (float *)calloc((size_t)ROWS * NG, sizeof(float)), *ref = (float *)calloc((size_t)ROWS * NG, sizeof(float));
        if (!full || !ref) return 1;
        for (int r = 0; r < world; ++r) {
            const int a = block_start(r, world), n = block_start(r + 1, world) - a;
            for (int k = 0; k < ROWS; ++k)
                memcpy(full + (size_t)k * NG + a, recv + ((size_t)r * ROWS + k) * ng_max, sizeof(float) * (size_t)n);
        }
        if (fit_block(rank, device, 0, NG, ref)) return 1;
        for (size_t i = 0; i < (size_t)ROWS * NG; ++i)
            if (memcmp(&full[i], &ref[i], sizeof(float)) != 0) {
                if (bad < 5) fprintf(stderr, "row %zu gene %zu: gathered %.9g, one handle %.9g\n", i / NG, i % NG, full[i], ref[i]);
                ++bad;
            }
        printf("%s world=%d genes=%d rows=%d mismatches=%d  Wc_loc[0][0]=%.6f sigma[%d]=%.6f loss_gene[%d]=%.4f\n",
               bad ? "MISMATCH" : "OK", world, NG, ROWS, bad, full[0], NG - 1, full[(size_t)(KC + 1) * NG + NG - 1], NG - 1,
               full[(size_t)(KC + 2) * NG + NG - 1]);
        free(full); free(ref);
    }
    CHECK(brie_comm_destroy(comm));
    free(mine); free(send); free(recv);
    return bad ? 1 : 0;
}
