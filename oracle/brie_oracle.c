/*
 * brie_oracle.c -- fused C / OpenMP restatement of the ELBO + Adam step.  TEST INFRASTRUCTURE ONLY.
 *
 * Same arithmetic as oracle/brie_oracle.py (OracleBRIE2.loss_and_grads + adam_step, fp32, gene intercept
 * mode, Kg = 0), i.e. of /root/reference/brie/models/model_TFProb.py:118-127 (Z_prior), 130-191
 * (logLik_MC), 194-211 (get_loss) and of Keras Adam + the clip constraints (:69,81,237-241), written as
 * one pass over the cells with per-thread per-gene accumulators.  Two uses:
 *   1. a second, independent implementation the NumPy oracle is checked against (tests/test_oracle_c.py);
 *   2. bench.py's "cpu_baseline_fused": what a tuned multi-core CPU kernel of the same algorithm reaches
 *      (reported NEXT TO the reference-shaped eager baseline, never instead of it -- BASELINE.md section 3).
 * Build: gcc -O3 -fopenmp -shared -fPIC oracle/brie_oracle.c -o oracle/_build/libbrie_oracle.so -lm
 * Nothing in brie_amd/ may link or load this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- noise stream of oracle/philox.py: Philox4x32-10 + Box-Muller (double, rounded once to float) ---- */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    const uint64_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = M0 * c[0], p1 = M1 * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
static double u01(uint32_t x) { return ((double)(x >> 9) + 0.5) * (1.0 / 8388608.0); }
static void normal4(uint32_t quad, uint32_t cell, uint32_t draw, uint32_t k, uint64_t seed, float e[4]) {
    uint32_t c[4] = {quad, cell, draw, k};
    philox4x32_10(c, (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32));
    for (int p = 0; p < 2; ++p) {
        const double r = sqrt(-2.0 * log(u01(c[2 * p]))), th = 6.283185307179586476925 * u01(c[2 * p + 1]);
        e[2 * p] = (float)(r * cos(th));
        e[2 * p + 1] = (float)(r * sin(th));
    }
}

static inline float sigmoidf_(float x) {
    const float e = expf(-fabsf(x));
    return x >= 0.0f ? 1.0f / (1.0f + e) : e / (1.0f + e);
}
static inline float log_sigmoidf_(float x) { return fminf(x, 0.0f) - log1pf(expf(-fabsf(x))); }

typedef struct {
    int32_t Nc, Ng, Kc, n_layers, has_efflen, mc, train_b, train_lam;
    int64_t gene_offset;
    uint64_t seed;
} brie_oracle_problem;

static void adam(float *x, float *m, float *v, float g, float alpha, int clip) {
    *m += (g - *m) * (1.0f - 0.9f);
    *v += (g * g - *v) * (1.0f - 0.999f);
    *x -= (*m * alpha) / (sqrtf(*v) + 1e-7f);
    if (clip) *x = fminf(fmaxf(*x, -9.0f), 9.0f);
}

/* n_steps optimisation steps in place; trace[i] = loss BEFORE update i (sum KL - sum ll, double sums).
 * t0 = Adam iterations already taken by this optimiser, draw0 = first noise draw id.  Returns 0. */
int brie_oracle_steps(const brie_oracle_problem *p, int32_t n_steps, float lr, int32_t t0, uint32_t draw0,
                      const float *c1, const float *c2, const float *c3, const float *Xc, const float *effLen,
                      float *Z_loc, float *Z_std_log, float *m_mu, float *v_mu, float *m_rho, float *v_rho,
                      float *W, float *m_W, float *v_W, float *b, float *m_b, float *v_b, float *lam, float *m_lam,
                      float *v_lam, float *trace) {
    const int Nc = p->Nc, Ng = p->Ng, Kc = p->Kc, S = Kc + 4;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    double *acc = (double *)malloc(sizeof(double) * (size_t)nthreads * S * Ng);
    float *lL = (float *)malloc(sizeof(float) * 3 * (size_t)Ng);
    if (!acc || !lL) return -1;
    if (p->has_efflen)
        for (int j = 0; j < Ng; ++j) {
            lL[j] = logf(effLen[6 * j + 0]); lL[Ng + j] = logf(effLen[6 * j + 4]); lL[2 * Ng + j] = logf(effLen[6 * j + 5]);
        }
    for (int step = 0; step < n_steps; ++step) {
        const int t = t0 + step + 1;
        const float alpha = lr * sqrtf(1.0f - powf(0.999f, (float)t)) / (1.0f - powf(0.9f, (float)t));
        const uint32_t draw = draw0 + (uint32_t)step;
        memset(acc, 0, sizeof(double) * (size_t)nthreads * S * Ng);
#pragma omp parallel
        {
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            double *a = acc + (size_t)tid * S * Ng;
#pragma omp for schedule(static)
            for (int i = 0; i < Nc; ++i) {
                const float *x = Xc + (size_t)i * Kc;
                for (int j0 = 0; j0 < Ng; j0 += 4) {
                    float eps[8][4];                       /* up to 8 MC samples */
                    const uint32_t quad = (uint32_t)((p->gene_offset + j0) / 4);
                    for (int k = 0; k < p->mc; ++k) normal4(quad, (uint32_t)i, draw, (uint32_t)k, p->seed, eps[k]);
                    for (int v = 0; v < 4 && j0 + v < Ng; ++v) {
                        const int j = j0 + v;
                        const size_t o = (size_t)i * Ng + j;
                        const float mu = Z_loc[o], rho = Z_std_log[o], s = expf(rho);
                        float m = b[j];
                        for (int k = 0; k < Kc; ++k) m += x[k] * W[(size_t)k * Ng + j];
                        const float isig2 = expf(-2.0f * lam[j]), d = mu - m, rr = d * isig2, s2r = s * s * isig2;
                        const float kl = 0.5f * d * d * isig2 + 0.5f * expm1f(2.0f * (rho - lam[j])) - (rho - lam[j]);
                        float gbar = 0.0f, gse = 0.0f, ll = 0.0f;
                        for (int k = 0; k < p->mc; ++k) {
                            const float z = mu + s * eps[k][v];
                            const float ls1 = log_sigmoidf_(z), ls2 = log_sigmoidf_(-z), sp = sigmoidf_(z);
                            float l, g;
                            if (!p->has_efflen) {
                                l = c1[o] * ls1 + c2[o] * ls2;
                                g = c1[o] - (c1[o] + c2[o]) * sp;
                            } else {
                                const float a1 = ls1 + lL[j], a2 = ls2 + lL[Ng + j], a3 = lL[2 * Ng + j];
                                const float mx = fmaxf(a1, fmaxf(a2, a3));
                                const float lse = mx + logf(expf(a1 - mx) + expf(a2 - mx) + expf(a3 - mx));
                                const float cc3 = p->n_layers > 2 ? c3[o] : 0.0f, N = c1[o] + c2[o] + cc3;
                                l = c1[o] * (a1 - lse) + c2[o] * (a2 - lse) + cc3 * (a3 - lse);
                                g = c1[o] * (1.0f - sp) - c2[o] * sp - N * (expf(a1 - lse) * (1.0f - sp) - expf(a2 - lse) * sp);
                            }
                            ll += l; gbar += g; gse += g * s * eps[k][v];
                        }
                        ll /= (float)p->mc; gbar /= (float)p->mc; gse /= (float)p->mc;
                        for (int k = 0; k < Kc; ++k) a[(size_t)k * Ng + j] += (double)(x[k] * rr);
                        a[(size_t)(Kc + 0) * Ng + j] += rr;
                        a[(size_t)(Kc + 1) * Ng + j] += 1.0f - d * d * isig2 - s2r;
                        a[(size_t)(Kc + 2) * Ng + j] += kl;
                        a[(size_t)(Kc + 3) * Ng + j] += ll;
                        adam(&Z_loc[o], &m_mu[o], &v_mu[o], rr - gbar, alpha, 1);
                        adam(&Z_std_log[o], &m_rho[o], &v_rho[o], s2r - 1.0f - gse, alpha, 0);
                    }
                }
            }
        }
        double loss_kl = 0.0, loss_ll = 0.0;
        for (int j = 0; j < Ng; ++j) {
            double tot[64];
            for (int s = 0; s < S; ++s) {
                tot[s] = 0.0;
                for (int th = 0; th < nthreads; ++th) tot[s] += acc[((size_t)th * S + s) * Ng + j];
            }
            for (int k = 0; k < Kc; ++k) adam(&W[(size_t)k * Ng + j], &m_W[(size_t)k * Ng + j], &v_W[(size_t)k * Ng + j], (float)(-tot[k]), alpha, 0);
            if (p->train_b) adam(&b[j], &m_b[j], &v_b[j], (float)(-tot[Kc]), alpha, 1);
            if (p->train_lam) adam(&lam[j], &m_lam[j], &v_lam[j], (float)tot[Kc + 1], alpha, 0);
            loss_kl += tot[Kc + 2]; loss_ll += tot[Kc + 3];
        }
        if (trace) trace[step] = (float)(loss_kl - loss_ll);
    }
    free(acc); free(lL);
    return 0;
}

int brie_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
