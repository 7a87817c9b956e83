/* The C restatement of the step (oracle/brie_oracle.c, TEST INFRASTRUCTURE) under AddressSanitizer +
 * UndefinedBehaviorSanitizer on the CPU -- GPU sanitizers are not available on this pool, so the checker at
 * least is checked.  Built and run by tests/test_oracle_c.py; ragged shapes on purpose (Ng % 4 != 0). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#ifdef BRIE_ORACLE_F64
typedef double real;
#else
typedef float real;
#endif
typedef struct {
    int32_t Nc, Ng, Kc, n_layers, has_efflen, mc, train_b, train_lam;
    int64_t gene_offset;
    uint64_t seed;
} brie_oracle_problem;
int brie_oracle_steps(const brie_oracle_problem *p, int32_t n_steps, double lr, int32_t t0, uint32_t draw0, const real *c1,
                      const real *c2, const real *c3, const real *Xc, const real *effLen, real *Z_loc, real *Z_std_log,
                      real *m_mu, real *v_mu, real *m_rho, real *v_rho, real *W, real *m_W, real *v_W, real *b, real *m_b,
                      real *v_b, real *lam, real *m_lam, real *v_lam, real *trace);

static real *arr(size_t n, real v) {
    real *p = (real *)malloc(n * sizeof(real));
    for (size_t i = 0; i < n; ++i) p[i] = v;
    return p;
}

int main(void) {
    for (int L = 2; L <= 3; ++L) {
        const int Nc = 37, Ng = 23, Kc = 2, steps = 6;
        brie_oracle_problem p = {Nc, Ng, Kc, L, L == 3, 3, 1, 1, 8, 1234};
        const size_t n = (size_t)Nc * Ng;
        real *c1 = arr(n, 0), *c2 = arr(n, 0), *c3 = arr(n, 0), *Xc = arr((size_t)Nc * Kc, 0), *eff = arr((size_t)Ng * 6, 100);
        for (size_t i = 0; i < n; ++i) { c1[i] = (real)((i * 7) % 5); c2[i] = (real)((i * 3) % 4); c3[i] = (real)(i % 3); }
        for (int i = 0; i < Nc * Kc; ++i) Xc[i] = (real)sin(0.3 * i);
        real *Z = arr(n, 0.1f), *R = arr(n, -0.2f), *mm = arr(n, 0), *vm = arr(n, 0), *mr = arr(n, 0), *vr = arr(n, 0);
        real *W = arr((size_t)Kc * Ng, 0.05f), *mW = arr((size_t)Kc * Ng, 0), *vW = arr((size_t)Kc * Ng, 0);
        real *b = arr(Ng, 0), *mb = arr(Ng, 0), *vb = arr(Ng, 0), *lam = arr(Ng, 0), *ml = arr(Ng, 0), *vl = arr(Ng, 0);
        real *tr = arr(steps, 0);
        if (brie_oracle_steps(&p, steps, 0.01, 0, 0, c1, c2, L == 3 ? c3 : NULL, Xc, L == 3 ? eff : NULL, Z, R, mm, vm, mr, vr, W,
                              mW, vW, b, mb, vb, lam, ml, vl, tr) != 0)
            return 2;
        if (!(tr[steps - 1] < tr[0]) || !isfinite((double)tr[0])) { fprintf(stderr, "trace %g -> %g\n", (double)tr[0], (double)tr[steps - 1]); return 1; }
        real *all[] = {c1, c2, c3, Xc, eff, Z, R, mm, vm, mr, vr, W, mW, vW, b, mb, vb, lam, ml, vl, tr};
        for (size_t i = 0; i < sizeof all / sizeof all[0]; ++i) free(all[i]);
    }
    puts("oracle clean under ASan + UBSan");
    return 0;
}
