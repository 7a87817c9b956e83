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
 *        (+ -DBRIE_ORACLE_F64 -o .../libbrie_oracle_f64.so: the same code with every fp32 quantity held in
 *         double -- the precision-independent answer the fp32 results are measured against)
 *        (+ -DBRIE_ORACLE_B -ffp-contract=fast [-mfma] -o .../libbrie_oracle_b.so: "o32b", a SECOND fp32 evaluation of the
 *         same algorithm that differs from the first the way any other fp32 implementation legitimately may -- the
 *         noise stream's Box-Muller evaluated in float (logf / sqrtf / cosf / sinf of the float-rounded angle: within
 *         2e-6 of the fp64-rounded value, the tolerance tests/test_gpu_parity.py grants the device), the cells of a
 *         thread walked in REVERSE order with the per-gene sums formed in fp32 over 64 cells at a time before they
 *         enter the fp64 totals (the HIP kernels form fp32 partial sums per lane and chunk), and the compiler free to
 *         contract a * b + c into fused multiply-adds.  o32b-vs-o32 is the null distribution the HIP-vs-o32
 *         differences are held against: tests/util.py::psi_null_rule, profiles/psi_null.py)
 *         The o32b build takes three run-time knobs (brie_oracle_b_config: float or exact Box-Muller, reversed or forward
 *         cell order, cells per fp32 partial sum) which, with the OpenMP thread count (= where the per-thread sums are cut),
 *         define the MEMBERS of the pre-registered null ensemble (tests/util.py::psi_ensemble_rule,
 *         tests/golden/psi_ensemble_manifest.json); the defaults (1, 1, 64) are the single draw of round 4.
 *        (+ -DBRIE_ORACLE_MUTANTS [with -DBRIE_ORACLE_B ...] -o .../libbrie_oracle_mut.so: NEGATIVE CONTROLS.  The build takes
 *         brie_oracle_set_mutant(id): deliberately WRONG variants of the algorithm (enum below) that the parity rules of
 *         tests/util.py must reject -- tests/test_rule_power.py, tests/tools/rule_power.py.  Without the flag every MUT(x)
 *         is the constant 0 and the code is what it was.)
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
#ifdef BRIE_ORACLE_F64
typedef double real;
#define R_EXP exp
#define R_LOG log
#define R_LOG1P log1p
#define R_EXPM1 expm1
#define R_SQRT sqrt
#define R_POW pow
#define R_FABS fabs
#define R_FMIN fmin
#define R_FMAX fmax
#else
typedef float real;
#define R_EXP expf
#define R_LOG logf
#define R_LOG1P log1pf
#define R_EXPM1 expm1f
#define R_SQRT sqrtf
#define R_POW powf
#define R_FABS fabsf
#define R_FMIN fminf
#define R_FMAX fmaxf
#endif
#define RC(x) ((real)(x))

/* ---- negative controls (tests/tools/rule_power.py): one deliberate error at a time, selected at run time ---- */
enum { MUT_NONE = 0,
       MUT_ADAM_EPS_TORCH = 1,   /* eps inside the bias correction: lr/(1-b1^t) m / (sqrt(v/(1-b2^t)) + eps)  (torch.optim.Adam) */
       MUT_ADAM_EPS_1E8 = 2,     /* eps = 1e-8 (torch's default) instead of Keras' 1e-7 */
       MUT_NO_CLIP = 3,          /* no clip of Z_loc / intercept to [-9, 9] (model_TFProb.py:69,81) */
       MUT_BETA2_DOUBLE = 4,     /* 1 - beta_2 formed in double and then rounded (0.001f) instead of 1.0f - 0.999f */
       MUT_KL_NO_EXPM1 = 6,      /* KL without 0.5 expm1(2 (rho - lambda)): the s^2 / sigma^2 term and its gradients dropped */
       MUT_MC_SAME_NOISE = 9,    /* every MC sample of a step uses the noise of sample 0 */
       MUT_NO_BIAS_CORR = 10,    /* Adam without bias correction: alpha = lr */
       MUT_LIK_GRAD_1PCT = 11,   /* d loglik / dz scaled by 1.01 */
       MUT_LIK_GRAD_01PCT = 12,  /* d loglik / dz scaled by 1.001 */
       MUT_KL_GRAD_1PCT = 13,    /* prior pull (mu - m) / sigma^2 scaled by 1.01 in the Z_loc gradient */
       MUT_SIGMA_GRAD_SIGN = 14  /* gradient of sigma_log with the sign of its s^2 / sigma^2 term flipped */
};
#ifdef BRIE_ORACLE_MUTANTS
static int g_mutant = 0;
#define MUT(x) (g_mutant == (x))
#else
#define MUT(x) 0
#endif

/* The cells are cut into PARTS with their own per-gene accumulators, summed in part order: by default one part per OpenMP
 * thread with the ranges of `omp for schedule(static)` (so the sums depend on the thread count); brie_oracle_set_parts(n)
 * fixes the number of parts whatever the thread count, which makes a run with "n threads" reproducible on any host. */
static int g_parts = 0;
#ifdef BRIE_ORACLE_B
static int g_b_float_noise = 1, g_b_reverse = 1, g_b_chunk = 64;      /* brie_oracle_b_config */
#endif
static double u01(uint32_t x) { return ((double)(x >> 9) + 0.5) * (1.0 / 8388608.0); }
/* eps is the fp32 value of oracle/philox.py in both precisions (the noise stream is DEFINED in fp32) */
static void normal4(uint32_t quad, uint32_t cell, uint32_t draw, uint32_t k, uint64_t seed, real e[4]) {
    uint32_t c[4] = {quad, cell, draw, k};
    philox4x32_10(c, (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32));
    for (int p = 0; p < 2; ++p) {
#ifdef BRIE_ORACLE_B     /* Box-Muller in float: the uniforms are exact in fp32, everything after them rounds in fp32 */
        if (!g_b_float_noise) {
            const double r = sqrt(-2.0 * log(u01(c[2 * p]))), th = 6.283185307179586476925 * u01(c[2 * p + 1]);
            e[2 * p] = (real)(float)(r * cos(th));
            e[2 * p + 1] = (real)(float)(r * sin(th));
            continue;
        }
        const float ua = ((float)(c[2 * p] >> 9) + 0.5f) * 0x1p-23f, ub = ((float)(c[2 * p + 1] >> 9) + 0.5f) * 0x1p-23f;
        const float r = sqrtf(-2.0f * logf(ua)), th = 6.283185307179586f * ub;
        e[2 * p] = (real)(r * cosf(th));
        e[2 * p + 1] = (real)(r * sinf(th));
#else
        const double r = sqrt(-2.0 * log(u01(c[2 * p]))), th = 6.283185307179586476925 * u01(c[2 * p + 1]);
        e[2 * p] = (real)(float)(r * cos(th));
        e[2 * p + 1] = (real)(float)(r * sin(th));
#endif
    }
}

static inline real sigmoidf_(real x) {
    const real e = R_EXP(-R_FABS(x));
    return x >= RC(0) ? RC(1) / (RC(1) + e) : e / (RC(1) + e);
}
static inline real log_sigmoidf_(real x) { return R_FMIN(x, RC(0)) - R_LOG1P(R_EXP(-R_FABS(x))); }

typedef struct {
    int32_t Nc, Ng, Kc, n_layers, has_efflen, mc, train_b, train_lam;
    int64_t gene_offset;
    uint64_t seed;
} brie_oracle_problem;

static real g_lr_over_bc1 = 0, g_sqrt_bc2 = 1;      /* the two halves of alpha, for MUT_ADAM_EPS_TORCH (set once per step) */
static void adam(real *x, real *m, real *v, real g, real alpha, int clip) {
    *m += (g - *m) * (RC(1) - RC(0.9));
    *v += (g * g - *v) * (MUT(MUT_BETA2_DOUBLE) ? (real)(1.0 - 0.999) : RC(1) - RC(0.999));
    if (MUT(MUT_ADAM_EPS_TORCH)) *x -= (*m * g_lr_over_bc1) / (R_SQRT(*v) / g_sqrt_bc2 + RC(1e-7));
    else *x -= (*m * alpha) / (R_SQRT(*v) + (MUT(MUT_ADAM_EPS_1E8) ? RC(1e-8) : RC(1e-7)));
    if (clip && !MUT(MUT_NO_CLIP)) *x = R_FMIN(R_FMAX(*x, RC(-9)), RC(9));
}

/* n_steps optimisation steps in place; trace[i] = loss BEFORE update i (sum KL - sum ll, double sums).
 * t0 = Adam iterations already taken by this optimiser, draw0 = first noise draw id.  Returns 0. */
int brie_oracle_steps(const brie_oracle_problem *p, int32_t n_steps, double lr_, int32_t t0, uint32_t draw0,
                      const real *c1, const real *c2, const real *c3, const real *Xc, const real *effLen,
                      real *Z_loc, real *Z_std_log, real *m_mu, real *v_mu, real *m_rho, real *v_rho,
                      real *W, real *m_W, real *v_W, real *b, real *m_b, real *v_b, real *lam, real *m_lam,
                      real *v_lam, real *trace) {
    const real lr = (real)lr_;
    const int Nc = p->Nc, Ng = p->Ng, Kc = p->Kc, S = Kc + 4;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    const int parts = g_parts > 0 ? g_parts : nthreads;
    double *acc = (double *)malloc(sizeof(double) * (size_t)parts * S * Ng);
    real *lL = (real *)malloc(sizeof(real) * 3 * (size_t)Ng);
#ifdef BRIE_ORACLE_B
    float *accf = (float *)malloc(sizeof(float) * (size_t)parts * S * Ng);     /* fp32 sums of <= 64 cells */
    if (!accf) return -1;
#endif
    if (!acc || !lL) return -1;
    if (p->has_efflen)
        for (int j = 0; j < Ng; ++j) {
            lL[j] = R_LOG(effLen[6 * j + 0]); lL[Ng + j] = R_LOG(effLen[6 * j + 4]); lL[2 * Ng + j] = R_LOG(effLen[6 * j + 5]);
        }
    for (int step = 0; step < n_steps; ++step) {
        const int t = t0 + step + 1;
        const real alpha = MUT(MUT_NO_BIAS_CORR) ? lr : lr * R_SQRT(RC(1) - R_POW(RC(0.999), (real)t)) / (RC(1) - R_POW(RC(0.9), (real)t));
        g_lr_over_bc1 = lr / (RC(1) - R_POW(RC(0.9), (real)t));
        g_sqrt_bc2 = R_SQRT(RC(1) - R_POW(RC(0.999), (real)t));
        const uint32_t draw = draw0 + (uint32_t)step;
        memset(acc, 0, sizeof(double) * (size_t)parts * S * Ng);
#pragma omp parallel for schedule(static, 1)
        for (int part = 0; part < parts; ++part) {
            /* the range `omp for schedule(static)` gives thread `part` of `parts` (libgomp: the first Nc % parts get one more) */
            const int q_ = Nc / parts, r_ = Nc % parts;
            const int ii0 = part < r_ ? (q_ + 1) * part : q_ * part + r_, ii1 = ii0 + q_ + (part < r_ ? 1 : 0);
#ifdef BRIE_ORACLE_B
            double *a64 = acc + (size_t)part * S * Ng;
            float *a = accf + (size_t)part * S * Ng;
            memset(a, 0, sizeof(float) * (size_t)S * Ng);
            int in_chunk = 0;
#else
            double *a = acc + (size_t)part * S * Ng;
#endif
            for (int ii = ii0; ii < ii1; ++ii) {
#ifdef BRIE_ORACLE_B
                const int i = g_b_reverse ? Nc - 1 - ii : ii;      /* the thread's cells in reverse order */
#else
                const int i = ii;
#endif
                const real *x = Xc + (size_t)i * Kc;
                for (int j0 = 0; j0 < Ng; j0 += 4) {
                    real eps[8][4];                        /* up to 8 MC samples */
                    const uint32_t quad = (uint32_t)((p->gene_offset + j0) / 4);
                    for (int k = 0; k < p->mc; ++k)
                        normal4(quad, (uint32_t)i, draw, MUT(MUT_MC_SAME_NOISE) ? 0u : (uint32_t)k, p->seed, eps[k]);
                    for (int v = 0; v < 4 && j0 + v < Ng; ++v) {
                        const int j = j0 + v;
                        const size_t o = (size_t)i * Ng + j;
                        const real mu = Z_loc[o], rho = Z_std_log[o], s = R_EXP(rho);
                        real m = b[j];
                        for (int k = 0; k < Kc; ++k) m += x[k] * W[(size_t)k * Ng + j];
                        const real isig2 = R_EXP(RC(-2) * lam[j]), d = mu - m, rr = d * isig2, s2r = s * s * isig2;
                        const real kl = RC(0.5) * d * d * isig2 + (MUT(MUT_KL_NO_EXPM1) ? RC(0) : RC(0.5) * R_EXPM1(RC(2) * (rho - lam[j]))) - (rho - lam[j]);
                        real gbar = RC(0), gse = RC(0), ll = RC(0);
                        for (int k = 0; k < p->mc; ++k) {
                            const real z = mu + s * eps[k][v];
                            const real ls1 = log_sigmoidf_(z), ls2 = log_sigmoidf_(-z), sp = sigmoidf_(z);
                            real l, g;
                            if (!p->has_efflen) {
                                l = c1[o] * ls1 + c2[o] * ls2;
                                g = c1[o] - (c1[o] + c2[o]) * sp;
                            } else {
                                const real a1 = ls1 + lL[j], a2 = ls2 + lL[Ng + j], a3 = lL[2 * Ng + j];
                                const real mx = R_FMAX(a1, R_FMAX(a2, a3));
                                const real lse = mx + R_LOG(R_EXP(a1 - mx) + R_EXP(a2 - mx) + R_EXP(a3 - mx));
                                const real cc3 = p->n_layers > 2 ? c3[o] : RC(0), N = c1[o] + c2[o] + cc3;
                                l = c1[o] * (a1 - lse) + c2[o] * (a2 - lse) + cc3 * (a3 - lse);
                                g = c1[o] * (RC(1) - sp) - c2[o] * sp - N * (R_EXP(a1 - lse) * (RC(1) - sp) - R_EXP(a2 - lse) * sp);
                            }
                            if (MUT(MUT_LIK_GRAD_1PCT)) g *= RC(1.01);
                            if (MUT(MUT_LIK_GRAD_01PCT)) g *= RC(1.001);
                            ll += l; gbar += g; gse += g * s * eps[k][v];
                        }
                        ll /= (real)p->mc; gbar /= (real)p->mc; gse /= (real)p->mc;
                        for (int k = 0; k < Kc; ++k) a[(size_t)k * Ng + j] += x[k] * rr;
                        a[(size_t)(Kc + 0) * Ng + j] += rr;
                        a[(size_t)(Kc + 1) * Ng + j] += RC(1) - d * d * isig2 - (MUT(MUT_KL_NO_EXPM1) ? RC(0) : MUT(MUT_SIGMA_GRAD_SIGN) ? -s2r : s2r);
                        a[(size_t)(Kc + 2) * Ng + j] += kl;
                        a[(size_t)(Kc + 3) * Ng + j] += ll;
                        adam(&Z_loc[o], &m_mu[o], &v_mu[o], (MUT(MUT_KL_GRAD_1PCT) ? RC(1.01) * rr : rr) - gbar, alpha, 1);
                        adam(&Z_std_log[o], &m_rho[o], &v_rho[o], (MUT(MUT_KL_NO_EXPM1) ? RC(0) : s2r) - RC(1) - gse, alpha, 0);
                    }
                }
#ifdef BRIE_ORACLE_B
                if (++in_chunk >= g_b_chunk) {
                    for (size_t q = 0; q < (size_t)S * Ng; ++q) { a64[q] += (double)a[q]; a[q] = 0.0f; }
                    in_chunk = 0;
                }
#endif
            }
#ifdef BRIE_ORACLE_B
            for (size_t q = 0; q < (size_t)S * Ng; ++q) a64[q] += (double)a[q];
#endif
        }
        double loss_kl = 0.0, loss_ll = 0.0;
        for (int j = 0; j < Ng; ++j) {
            double tot[64];
            for (int s = 0; s < S; ++s) {
                tot[s] = 0.0;
                for (int th = 0; th < parts; ++th) tot[s] += acc[((size_t)th * S + s) * Ng + j];
            }
            for (int k = 0; k < Kc; ++k) adam(&W[(size_t)k * Ng + j], &m_W[(size_t)k * Ng + j], &v_W[(size_t)k * Ng + j], (real)(-tot[k]), alpha, 0);
            if (p->train_b) adam(&b[j], &m_b[j], &v_b[j], (real)(-tot[Kc]), alpha, 1);
            if (p->train_lam) adam(&lam[j], &m_lam[j], &v_lam[j], (real)tot[Kc + 1], alpha, 0);
            loss_kl += tot[Kc + 2]; loss_ll += tot[Kc + 3];
        }
        if (trace) trace[step] = (real)(loss_kl - loss_ll);
    }
    free(acc); free(lL);
#ifdef BRIE_ORACLE_B
    free(accf);
#endif
    return 0;
}

/* the noise stream of this build, for tests: out[4] = eps of gene quad `quad`, cell `cell` at (draw, k) as float */
void brie_oracle_normal4(uint32_t quad, uint32_t cell, uint32_t draw, uint32_t k, uint64_t seed, float *out) {
    real e[4];
    normal4(quad, cell, draw, k, seed, e);
    for (int v = 0; v < 4; ++v) out[v] = (float)e[v];
}

/* members of the null ensemble: only the o32b build has these knobs (returns 0 there, -1 elsewhere) */
int brie_oracle_b_config(int float_noise, int reverse, int chunk) {
#ifdef BRIE_ORACLE_B
    g_b_float_noise = float_noise != 0; g_b_reverse = reverse != 0; g_b_chunk = chunk > 0 ? chunk : 64;
    return 0;
#else
    (void)float_noise; (void)reverse; (void)chunk;
    return -1;
#endif
}

/* negative controls: only the -DBRIE_ORACLE_MUTANTS build has the switch (returns 0 there, -1 elsewhere) */
int brie_oracle_set_mutant(int id) {
#ifdef BRIE_ORACLE_MUTANTS
    g_mutant = id;
    return 0;
#else
    (void)id;
    return -1;
#endif
}

void brie_oracle_set_parts(int n) { g_parts = n > 0 ? n : 0; }

int brie_oracle_real_bytes(void) { return (int)sizeof(real); }

int brie_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* torch.distributed.run exports OMP_NUM_THREADS=1 to its ranks: the cpu_baseline leg of bench.py sets the count itself */
void brie_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
