/* Plain-C client of include/brie_amd.h: proves the boundary needs nothing but a C compiler.
 * Fits a tiny problem (counts drawn with a fixed LCG), prints the loss trace and a checksum of Psi.
 * Build:  gcc tests/c_abi/smoke.c -Iinclude -Lbrie_amd/lib -lbrie_amd -Wl,-rpath,$PWD/brie_amd/lib -lm -o smoke
 * (tests/test_c_abi.py compiles it on CPU and runs it on the GPU box) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "brie_amd.h"

#define CHECK(call)                                                            \
    do {                                                                       \
        int rc_ = (call);                                                      \
        if (rc_ != BRIE_OK) {                                                  \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, brie_last_error()); \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main(void) {
    enum { NC = 96, NG = 300, KC = 2, STEPS = 40 };
    static float c1[NC * NG], c2[NC * NG], xc[NC * KC], psi[NC * NG], trace[STEPS], lg[NG];
    unsigned long long s = 88172645463325252ULL;
    for (int i = 0; i < NC * NG; ++i) {
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        c1[i] = (float)((s >> 33) % 7);
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        c2[i] = (float)((s >> 33) % 5);
    }
    for (int i = 0; i < NC; ++i) { xc[i * KC] = (float)(i & 1); xc[i * KC + 1] = (float)sin(0.37 * i); }

    brie_problem p;
    memset(&p, 0, sizeof p);
    p.abi_version = BRIE_AMD_ABI_VERSION;
    p.Nc = NC; p.Ng = NG; p.Kc = KC; p.n_layers = 2;
    p.train_intercept = 1; p.train_sigma = 1; p.seed = 2024;
    brie_handle *h = NULL;
    if (brie_abi_version() != BRIE_AMD_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    CHECK(brie_create(&p, &h));
    CHECK(brie_upload(h, BRIE_COUNT1, c1, NC, NG, NG));
    CHECK(brie_upload(h, BRIE_COUNT2, c2, NC, NG, NG));
    CHECK(brie_add_pseudo_count(h, 0.01f));
    CHECK(brie_upload(h, BRIE_XC, xc, NC, KC, KC));
    CHECK(brie_init_state(h, NAN, NAN));
    CHECK(brie_reset_optimizer(h));
    CHECK(brie_step(h, STEPS, 0.02f, 1, trace));
    CHECK(brie_loss_gene(h, 5, lg));
    CHECK(brie_read(h, BRIE_PSI, psi, NC, NG, NG));
    double sum = 0.0;
    for (int i = 0; i < NC * NG; ++i) {
        if (!(psi[i] > 0.0f && psi[i] < 1.0f)) { fprintf(stderr, "psi[%d] = %g\n", i, psi[i]); return 1; }
        sum += psi[i];
    }
    /* round-2 entry points from plain C: one-pass asynchronous export, next LRT model on the same counts, RCCL */
    {
        static float psi2[NC * NG], zstd[NC * NG], ci[NC * NG], lg2[NG];
        CHECK(brie_read_results_async(h, psi2, zstd, ci, NULL, NG));
        CHECK(brie_loss_gene(h, 3, lg2));                        /* overlaps the export */
        CHECK(brie_read_wait(h));
        if (memcmp(psi2, psi, sizeof psi) != 0) { fprintf(stderr, "async Psi differs from brie_read\n"); return 1; }
        for (int i = 0; i < NC * NG; ++i)
            if (!(zstd[i] > 0.0f && ci[i] >= 0.0f && ci[i] <= 1.0f)) { fprintf(stderr, "bad Z_std / CI at %d\n", i); return 1; }
        CHECK(brie_reconfigure(h, 1, 77, 1, 1));                 /* reduced model: feature 0 dropped, counts stay */
        static float x1[NC];
        for (int i = 0; i < NC; ++i) x1[i] = xc[i * KC + 1];
        CHECK(brie_upload(h, BRIE_XC, x1, NC, 1, 1));
        CHECK(brie_init_state(h, NAN, NAN));
        float tr2[8];
        CHECK(brie_step(h, 8, 0.02f, 1, tr2));
        if (!(tr2[7] < tr2[0])) { fprintf(stderr, "reconfigured model does not descend\n"); return 1; }
        uint8_t id[BRIE_COMM_ID_BYTES];
        brie_comm *comm = NULL;
        float v[4] = {1.0f, 2.0f, 3.0f, 4.0f}, g[4];
        double d[2] = {0.5, -1.0};
        CHECK(brie_comm_available(0));                           /* what every rank checks alone before an id exists */
        if (brie_comm_available(9999) == BRIE_OK) { fprintf(stderr, "device 9999 accepted\n"); return 1; }
        CHECK(brie_comm_unique_id(id));
        CHECK(brie_comm_init(0, 0, 1, id, &comm));
        CHECK(brie_comm_allgather(comm, v, 4, g));
        CHECK(brie_comm_allreduce(comm, d, 2, BRIE_F64, BRIE_SUM));
        if (memcmp(v, g, sizeof v) != 0 || d[0] != 0.5 || brie_comm_world(comm) != 1) { fprintf(stderr, "comm\n"); return 1; }
        CHECK(brie_comm_destroy(comm));
    }
    /* round 5: the placement search on demand from plain C -- two sets, unreachable stop rate: best effort, says how it ended */
    {
        int32_t tries = 0, kept = 0, status = -1;
        int64_t peak = 0;
        double gbs[BRIE_PLACEMENT_MAX_SETS], secs = 0.0, rate = 0.0;
        char why[192];
        CHECK(brie_placement_probe(h, 2, &rate));
        CHECK(brie_placement_tune(h, 2, 1e30));
        CHECK(brie_placement_info(h, &tries, &kept, gbs, BRIE_PLACEMENT_MAX_SETS, &secs));
        CHECK(brie_placement_status(h, &status, &peak, why, (int32_t)sizeof why));
        if (!(rate > 0.0) || tries != 2 || kept < 0 || kept > 1 || status != BRIE_PLACEMENT_BEST_OF_ALL || peak <= 0 || !why[0]) {
            fprintf(stderr, "placement: rate %g tries %d kept %d status %d peak %lld note \"%s\"\n", rate, (int)tries, (int)kept,
                    (int)status, (long long)peak, why);
            return 1;
        }
        float tr3[2];
        CHECK(brie_step(h, 2, 0.02f, 1, tr3));                   /* the moved arrays step on */
    }
    if (brie_step(h, 1, 0.02f, 0, NULL) == BRIE_OK) { fprintf(stderr, "mc_size 0 accepted\n"); return 1; }
    printf("loss %.3f -> %.3f  mean_psi %.6f  loss_gene0 %.3f  last_error \"%s\"\n", trace[0], trace[STEPS - 1],
           sum / (NC * NG), lg[0], brie_last_error());
    CHECK(brie_destroy(h));
    return trace[STEPS - 1] < trace[0] ? 0 : 1;
}
