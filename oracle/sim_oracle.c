/*
 * sim_oracle.c -- CPU restatement of the count simulator.  TEST INFRASTRUCTURE ONLY.
 *
 * What it restates: /root/reference/brie/models/simulator.py:45-69 -- reads of every (cell, gene) are
 * Multinomial(total, phi) with phi proportional to [Psi, 1-Psi, 1] * effLen[:, [0,4,5]] -- and :31-41,
 * Psi = expit(clip(mean + sigma_j * N(0,1), -9, 9)).  The reference samples with
 * tfd.Multinomial(...).sample() / np.random.normal, both unseeded, so only the DISTRIBUTION is defined by the
 * reference; tests/test_oracle.py pins this file's distribution against scipy.stats.binom / multinomial
 * moments, and tests/test_gpu_sim.py demands bit-identical counts from the HIP kernel, which draws from the
 * same Philox stream with the same fp64 arithmetic:
 *   multinomial = two conditional binomials (c1 ~ B(n, p1); c2 ~ B(n - c1, p2 / (p2 + p3)));
 *   binomial    = sequential inversion when n * min(p, 1-p) < 10, else Hoermann's BTRS (transformed rejection
 *                 with squeeze, 1993) with the usual Stirling-tail table;
 *   uniforms    = 53 bits from two words of Philox4x32-10(counter = (global gene, cell, draw id, k), key = seed).
 * Build (no FMA contraction, so gcc and hipcc evaluate the same IEEE expression tree):
 *   gcc -O2 -ffp-contract=off -shared -fPIC oracle/sim_oracle.c -o oracle/_build/libsim_oracle.so -lm
 * Nothing in brie_amd/ may link or load this file.
 */
#include <math.h>
#include <stdint.h>

static void philox(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

typedef struct { uint32_t gene, cell, draw, k; uint64_t seed; uint32_t w[4]; int left; } rng_t;

static double next_uniform(rng_t *g) {
    if (g->left == 0) {
        g->w[0] = g->gene; g->w[1] = g->cell; g->w[2] = g->draw; g->w[3] = g->k++;
        philox(g->w, (uint32_t)g->seed, (uint32_t)(g->seed >> 32));
        g->left = 2;
    }
    const int o = g->left == 2 ? 0 : 2;
    g->left--;
    return ((double)(g->w[o] >> 5) * 67108864.0 + (double)(g->w[o + 1] >> 6) + 0.5) / 9007199254740992.0;
}

/* log(k!) minus Stirling's leading terms; exact table below 10, asymptotic series above */
static double tail(double k) {
    static const double t[10] = {0.08106146679532726, 0.04134069595540929, 0.02767792568499834,
                                 0.02079067210376509, 0.01664469118982119, 0.01387612882307075,
                                 0.01189670994589177, 0.01041126526197209, 0.009255462182712733,
                                 0.008330563433362871};
    if (k <= 9.0) return t[(int)k];
    const double a = k + 1.0, a2 = a * a;
    return (1.0 / 12.0 - (1.0 / 360.0 - 1.0 / 1260.0 / a2) / a2) / a;
}

static double binomial(double n, double p, rng_t *g) {
    if (n <= 0.0 || p <= 0.0) return 0.0;
    if (p >= 1.0) return n;
    const int flip = p > 0.5;
    if (flip) p = 1.0 - p;
    const double q = 1.0 - p;
    double x = 0.0;
    if (n * p < 10.0) {
        const double f0 = exp(n * log(q));
        const double limit = fmin(n, n * p + 10.0 * sqrt(n * p * q + 1.0));
        double f = f0, u = next_uniform(g);
        while (u > f) {
            x += 1.0;
            if (x > limit) { x = 0.0; f = f0; u = next_uniform(g); continue; }
            u -= f;
            f = ((n - x + 1.0) * p * f) / (x * q);
        }
    } else {
        const double spq = sqrt(n * p * q), b = 1.15 + 2.53 * spq, a = -0.0873 + 0.0248 * b + 0.01 * p;
        const double c = n * p + 0.5, vr = 0.92 - 4.2 / b, alpha = (2.83 + 5.1 / b) * spq;
        const double m = floor((n + 1.0) * p), r = p / q;
        for (;;) {
            const double u = next_uniform(g) - 0.5;
            double v = next_uniform(g);
            const double us = 0.5 - fabs(u);
            const double k = floor((2.0 * a / us + b) * u + c);
            if (us >= 0.07 && v <= vr) { x = k; break; }
            if (k < 0.0 || k > n) continue;
            v = log(v * alpha / (a / (us * us) + b));
            const double bound = (m + 0.5) * log((m + 1.0) / (r * (n - m + 1.0))) +
                                 (n + 1.0) * log((n - m + 1.0) / (n - k + 1.0)) +
                                 (k + 0.5) * log(r * (n - k + 1.0) / (k + 1.0)) +
                                 tail(m) + tail(n - m) - tail(k) - tail(n - k);
            if (v <= bound) { x = k; break; }
        }
    }
    return flip ? n - x : x;
}

#define DRAW_B1 0xFFFFFFFDu
#define DRAW_B2 0xFFFFFFFCu

/* one binomial from the stream of (gene, cell): exposed for the distribution tests */
double sim_oracle_binomial(double n, double p, uint32_t gene, uint32_t cell, uint64_t seed) {
    rng_t g = {gene, cell, DRAW_B1, 0, seed, {0, 0, 0, 0}, 0};
    return binomial(n, p, &g);
}

void sim_oracle_counts(int64_t Nc, int64_t Ng, int64_t gene_offset, uint64_t seed, const float *psi,
                       const float *total, const float *effLen, float *o1, float *o2, float *o3) {
    for (int64_t i = 0; i < Nc; ++i)
        for (int64_t j = 0; j < Ng; ++j) {
            const int64_t e = i * Ng + j;
            const double n = floor((double)total[e]), ps = (double)psi[e];
            double p1 = ps, pc = 1.0;
            if (effLen) {
                const double w1 = ps * (double)effLen[j * 6 + 0], w2 = (1.0 - ps) * (double)effLen[j * 6 + 4],
                             w3 = (double)effLen[j * 6 + 5];
                p1 = w1 / (w1 + w2 + w3);
                pc = (w2 + w3) > 0.0 ? w2 / (w2 + w3) : 0.0;
            }
            rng_t g1 = {(uint32_t)(gene_offset + j), (uint32_t)i, DRAW_B1, 0, seed, {0, 0, 0, 0}, 0};
            const double c1 = binomial(n, p1, &g1);
            double c2 = n - c1;
            if (effLen) {
                rng_t g2 = {(uint32_t)(gene_offset + j), (uint32_t)i, DRAW_B2, 0, seed, {0, 0, 0, 0}, 0};
                c2 = binomial(n - c1, pc, &g2);
                o3[e] = (float)(n - c1 - c2);
            }
            o1[e] = (float)c1;
            o2[e] = (float)c2;
        }
}
