/*
 * TEST INFRASTRUCTURE — CPU restatement of the reference's belief-propagation decoders.
 *
 * Scalar float64, one frame at a time, written to follow the reference line by line (same
 * visiting order, same summation order, same tie rules).  It is the checker for the HIP path
 * and the CPU baseline timed beside it; nothing in the product links or calls it.
 * Pinned against golden vectors produced by running the reference's own decode methods
 * (tests/golden, generator oracle/gen_golden.py): bit-exact for R0/R1; for the sum-product
 * rules the transcendental functions come from libm instead of numpy, so those agree to
 * rounding only.
 *
 *   R0  /root/reference/ldpc.py:137-227     non-extrinsic min-sum, 0.75 on the whole posterior
 *   R1  /root/reference/bpsk.py:154-260     extrinsic normalised min-sum (channel.py:151-257)
 *   R2  /root/reference/bpsk.py:263-387     tanh-rule sum-product
 *   R3  /root/reference/channel.py:402-513  0.85-scaled tanh-rule sum-product
 *   R4  /root/reference/channel.py:260-398  phi-function sum-product
 *   syndrome: /root/reference/ldpc.py:76-80
 *   LNMS row-layered normalised min-sum: a build extension with no counterpart in the reference
 *   OMS  (rule 6) offset min-sum, flooding, R1's conventions: likewise a build extension
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int32_t rule;              /* 0..4 */
    int32_t max_iters;         /* 10 / 30 / 50 / 50 / 40 in the reference */
    int32_t early_exit;        /* 1: reference behaviour (check at the top of every iteration) */
    int32_t float_syndrome0;   /* R0: iteration-0 check is made on the raw LLR floats (ldpc.py:158,165) */
    int32_t num_active_checks; /* R0: check-node loop covers rows [0, this) (ldpc.py:30,175); <0 = all */
    int32_t num_refresh_vars;  /* R1-R4: v2c refresh covers variables [0, this) (bpsk.py:244); <0 = all */
    double scale;              /* 0.75 (R0, R1), 0.85 (R3) */
    double clip_tanh_arg;      /* R2: 20 (bpsk.py:333); <=0 = none */
    double clip_prod;          /* R2: 0.999999999 (bpsk.py:337); R3: 0.99999999 (channel.py:473) */
    double clip_post;          /* R2: 30, R3: 50, R4: 500 */
    double offset;             /* rule 6 (offset min-sum, build extension): beta */
} oracle_cfg;

typedef struct {
    int32_t m, n, E;
    const int32_t *row_ptr, *col_idx; /* CSR, ascending column */
    const int32_t *col_ptr, *csc_edge; /* CSC, ascending check; csc_edge = CSR position of the edge */
} oracle_code;

static int syndrome_ok(const oracle_code *c, const uint8_t *hard)
{
    for (int j = 0; j < c->m; j++) {
        int p = 0;
        for (int e = c->row_ptr[j]; e < c->row_ptr[j + 1]; e++) p ^= hard[c->col_idx[e]];
        if (p) return 0;
    }
    return 1;
}

/* numpy.dot(H, x) % 2 == 0 on float x: true iff every row sum is an even integer */
static int float_syndrome_ok(const oracle_code *c, const double *x)
{
    for (int j = 0; j < c->m; j++) {
        double s = 0.0;
        for (int e = c->row_ptr[j]; e < c->row_ptr[j + 1]; e++) s += x[c->col_idx[e]];
        double r = fmod(s, 2.0);
        if (!(r == 0.0)) return 0;
    }
    return 1;
}

static double np_sign(double x) { return x > 0 ? 1.0 : (x < 0 ? -1.0 : 0.0); }
static double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* phi(x) = log((e^x + 1) / (e^x - 1)), x clipped to [1e-12, 100]  (channel.py:392-398) */
static double phi(double x)
{
    x = clipd(x, 1e-12, 100.0);
    double ex = exp(x);
    return log((ex + 1.0) / (ex - 1.0));
}

/* numpy's add.reduce over a contiguous float64 vector (pairwise with 8 accumulators) for n < 128 */
static double np_sum(const double *a, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r[8];
    int i;
    for (i = 0; i < 8; i++) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; k++) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

/* ---------------------------------------------------------------- R0 */
static void decode_r0(const oracle_code *c, const oracle_cfg *g, const double *L, double *b,
                      double *S, double *msg, uint8_t *hard, int32_t *iters, uint8_t *conv)
{
    const int n = c->n;
    int mact = g->num_active_checks < 0 || g->num_active_checks > c->m ? c->m : g->num_active_checks;
    memcpy(b, L, sizeof(double) * n);
    *conv = 0;
    int t;
    for (t = 0; t < g->max_iters; t++) {
        if (g->early_exit) {
            if (t == 0 && g->float_syndrome0) {
                if (float_syndrome_ok(c, b)) {
                    /* the reference returns the LLR floats as "hard decisions" */
                    for (int i = 0; i < n; i++) hard[i] = b[i] == 0.0 ? 0 : (b[i] == 1.0 ? 1 : 2);
                    *iters = 0; *conv = 1; return;
                }
            } else {
                for (int i = 0; i < n; i++) hard[i] = b[i] < 0 ? 0 : 1;
                if (syndrome_ok(c, hard)) { *iters = t; *conv = 1; return; }
            }
        }
        for (int i = 0; i < n; i++) S[i] = 0.0;
        for (int j = 0; j < mact; j++) {
            int e0 = c->row_ptr[j], e1 = c->row_ptr[j + 1];
            if (e1 - e0 < 2) continue;
            for (int e = e0; e < e1; e++) {
                double mn = INFINITY, sg = 1.0;
                for (int f = e0; f < e1; f++) {
                    if (f == e) continue;
                    double v = b[c->col_idx[f]];
                    double a = fabs(v);
                    if (a < mn) mn = a;
                    sg *= np_sign(v);
                }
                msg[e] = mn * sg;
            }
            /* messagesReceivedByBits[target] += message, checks visited in ascending order */
            for (int e = e0; e < e1; e++) S[c->col_idx[e]] += msg[e];
        }
        for (int i = 0; i < n; i++) b[i] = g->scale * (S[i] + L[i]);
    }
    for (int i = 0; i < n; i++) hard[i] = b[i] < 0 ? 0 : 1;
    if (g->early_exit) { *iters = g->max_iters - 1; *conv = 0; }
    else { *iters = g->max_iters; *conv = (uint8_t)syndrome_ok(c, hard); }
}

/* ---------------------------------------------------------------- R1..R4 (extrinsic) */
static void decode_ext(const oracle_code *c, const oracle_cfg *g, const double *L, double *b,
                       double *q /*v2c, CSR order*/, double *r /*c2v, CSR order*/, uint8_t *hard,
                       int32_t *iters, uint8_t *conv)
{
    const int n = c->n, m = c->m, rule = g->rule;
    int nref = g->num_refresh_vars < 0 || g->num_refresh_vars > n ? n : g->num_refresh_vars;
    double tmp[64];
    for (int j = 0; j < m; j++)
        for (int e = c->row_ptr[j]; e < c->row_ptr[j + 1]; e++) { q[e] = L[c->col_idx[e]]; r[e] = 0.0; }
    for (int i = 0; i < n; i++) { b[i] = L[i]; hard[i] = b[i] > 0 ? 0 : 1; }
    *conv = 0;
    int t;
    for (t = 0; t < g->max_iters; t++) {
        if (g->early_exit && syndrome_ok(c, hard)) { *iters = t; *conv = 1; return; }
        /* check nodes */
        for (int j = 0; j < m; j++) {
            int e0 = c->row_ptr[j], e1 = c->row_ptr[j + 1];
            if ((rule == 1 || rule == 6) && e1 - e0 < 2) continue;
            for (int e = e0; e < e1; e++) {
                if (rule == 1 || rule == 6) {
                    double mn = INFINITY, sg = 1.0;
                    for (int f = e0; f < e1; f++) {
                        if (f == e) continue;
                        double a = fabs(q[f]);
                        if (a < mn) mn = a;
                        double s = np_sign(q[f]);
                        if (s == 0.0) s = 1.0;
                        sg *= s;
                    }
                    if (rule == 6) { /* offset min-sum: max(scale * min - offset, 0) * sign */
                        double mag = g->scale * mn - g->offset;
                        if (!(mag > 0)) mag = 0.0;
                        r[e] = mag * sg;
                    } else {
                        r[e] = g->scale * mn * sg;
                    }
                } else if (rule == 2 || rule == 3) {
                    double prod = 1.0;
                    for (int f = e0; f < e1; f++) {
                        if (f == e) continue;
                        double a = q[f] / 2;
                        if (g->clip_tanh_arg > 0) a = clipd(a, -g->clip_tanh_arg, g->clip_tanh_arg);
                        prod *= tanh(a);
                    }
                    prod = clipd(prod, -g->clip_prod, g->clip_prod);
                    r[e] = rule == 2 ? 2 * atanh(prod) : (g->scale * 2) * atanh(prod);
                } else { /* rule 4 */
                    double sg = 1.0;
                    int cnt = 0;
                    for (int f = e0; f < e1; f++) {
                        if (f == e) continue;
                        double v = q[f];
                        tmp[cnt++] = phi(fabs(v));
                        sg *= (v == 0 || v == 1e-9) ? 1.0 : np_sign(v);
                    }
                    double ps = clipd(np_sum(tmp, cnt), 1e-12, 20.0);
                    r[e] = sg * phi(ps);
                }
            }
        }
        /* variable nodes: totals (ascending check order, Python sum() starts from integer 0) */
        for (int i = 0; i < n; i++) {
            double s = 0.0;
            for (int k = c->col_ptr[i]; k < c->col_ptr[i + 1]; k++) s += r[c->csc_edge[k]];
            double tot = L[i] + s;
            if (g->clip_post > 0) tot = clipd(tot, -g->clip_post, g->clip_post);
            b[i] = tot;
            hard[i] = tot > 0 ? 0 : 1;
        }
        /* extrinsic refresh */
        for (int i = 0; i < nref; i++) {
            int k0 = c->col_ptr[i], k1 = c->col_ptr[i + 1];
            for (int k = k0; k < k1; k++) {
                double s = 0.0;
                for (int kk = k0; kk < k1; kk++)
                    if (kk != k) s += r[c->csc_edge[kk]];
                q[c->csc_edge[k]] = L[i] + s;
            }
        }
    }
    if (g->early_exit) { *iters = g->max_iters - 1; *conv = 0; }
    else { *iters = g->max_iters; *conv = (uint8_t)syndrome_ok(c, hard); }
}

/*
 * Decode `frames` frames of LLRs (row-major [frames, n]).  posterior may be NULL.
 * Returns 0, or -1 on bad arguments.  nthreads <= 0: all OpenMP threads.
 */
int bp_oracle_decode(const oracle_code *code, const oracle_cfg *cfg, const double *llr,
                     int64_t frames, uint8_t *hard, int32_t *iters, uint8_t *converged,
                     double *posterior, int nthreads)
{
    if (!code || !cfg || cfg->rule < 0 || cfg->rule == 5 || cfg->rule > 6 || cfg->max_iters < 1) return -1; /* layered: bp_oracle_decode_lnms */
    for (int j = 0; j < code->m; j++)
        if (code->row_ptr[j + 1] - code->row_ptr[j] > 64) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    const int n = code->n, E = code->E;
#pragma omp parallel
    {
        double *b = malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
        double *S = malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
        double *q = malloc(sizeof(double) * (size_t)(E > 0 ? E : 1));
        double *r = malloc(sizeof(double) * (size_t)(E > 0 ? E : 1));
#pragma omp for schedule(dynamic, 4)
        for (int64_t f = 0; f < frames; f++) {
            const double *L = llr + f * n;
            if (cfg->rule == 0)
                decode_r0(code, cfg, L, b, S, r, hard + f * n, iters + f, converged + f);
            else
                decode_ext(code, cfg, L, b, q, r, hard + f * n, iters + f, converged + f);
            if (posterior) memcpy(posterior + f * n, b, sizeof(double) * n);
        }
        free(b); free(S); free(q); free(r);
    }
    return 0;
}

/* ---------------------------------------------------------------- row-layered normalised min-sum
 * Build extension (nothing in the reference corresponds to it): checks in ascending order, each
 * working on the current posteriors.  Generated for double and for float so that the float GPU
 * kernel has a bit-exact CPU counterpart.  The channel LLR is stored as L + 0 (no -0.0 anywhere,
 * so "negative" and "sign bit set" coincide).  A variable that takes part in one check only
 * sends that check its channel LLR itself: its extrinsic information is exactly that, and
 * (L + r) - r would only add rounding noise.
 */
#define DEFINE_LNMS(NAME, REAL, FABS)                                                              \
    static void NAME(const oracle_code *c, const oracle_cfg *g, const double *Lin, REAL *P, REAL *R, \
                     uint8_t *hard, int32_t *iters, uint8_t *conv)                                 \
    {                                                                                              \
        const int n = c->n, m = c->m;                                                              \
        const REAL scale = (REAL)g->scale;                                                         \
        const int *cp = c->col_ptr;                                                                \
        REAL q[64];                                                                                \
        for (int i = 0; i < n; i++) { P[i] = (REAL)Lin[i] + (REAL)0; hard[i] = P[i] > 0 ? 0 : 1; } \
        for (int e = 0; e < c->E; e++) R[e] = (REAL)0;                                             \
        *conv = 0;                                                                                 \
        for (int t = 0; t < g->max_iters; t++) {                                                   \
            if (g->early_exit && syndrome_ok(c, hard)) { *iters = t; *conv = 1; return; }          \
            for (int j = 0; j < m; j++) {                                                          \
                const int e0 = c->row_ptr[j], d = c->row_ptr[j + 1] - e0;                          \
                if (d < 2) continue;                                                               \
                REAL m1 = INFINITY, m2 = INFINITY;                                                 \
                int neg = 0;                                                                       \
                for (int p = 0; p < d; p++) {                                                      \
                    const int i = c->col_idx[e0 + p];                                              \
                    q[p] = cp[i + 1] - cp[i] == 1 ? (REAL)Lin[i] + (REAL)0 : P[i] - R[e0 + p];     \
                    REAL a = FABS(q[p]);                                                           \
                    if (a < m1) { m2 = m1; m1 = a; } else if (a < m2) { m2 = a; }                  \
                    neg ^= q[p] < 0;                                                               \
                }                                                                                  \
                const REAL A = scale * m1, B = scale * m2;                                         \
                for (int p = 0; p < d; p++) {                                                      \
                    REAL mag = FABS(q[p]) == m1 ? B : A;                                           \
                    REAL r = (neg ^ (q[p] < 0)) ? -mag : mag;                                      \
                    R[e0 + p] = r;                                                                 \
                    P[c->col_idx[e0 + p]] = q[p] + r;                                              \
                }                                                                                  \
            }                                                                                      \
            for (int i = 0; i < n; i++) hard[i] = P[i] > 0 ? 0 : 1;                                \
        }                                                                                          \
        if (g->early_exit) { *iters = g->max_iters - 1; *conv = 0; }                               \
        else { *iters = g->max_iters; *conv = (uint8_t)syndrome_ok(c, hard); }                     \
    }
DEFINE_LNMS(decode_lnms_f64, double, fabs)
DEFINE_LNMS(decode_lnms_f32, float, fabsf)

/* Layered decode; use_float selects float arithmetic (posteriors are returned as doubles). */
int bp_oracle_decode_lnms(const oracle_code *code, const oracle_cfg *cfg, const double *llr,
                          int64_t frames, uint8_t *hard, int32_t *iters, uint8_t *converged,
                          double *posterior, int nthreads, int use_float)
{
    if (!code || !cfg || cfg->max_iters < 1) return -1;
    for (int j = 0; j < code->m; j++)
        if (code->row_ptr[j + 1] - code->row_ptr[j] > 64) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    const int n = code->n, E = code->E;
#pragma omp parallel
    {
        void *P = malloc(sizeof(double) * (size_t)n);
        void *R = malloc(sizeof(double) * (size_t)E);
#pragma omp for schedule(dynamic, 2)
        for (int64_t f = 0; f < frames; f++) {
            if (use_float) {
                decode_lnms_f32(code, cfg, llr + f * n, (float *)P, (float *)R, hard + f * n, iters + f, converged + f);
                if (posterior) for (int i = 0; i < n; i++) posterior[f * n + i] = ((float *)P)[i];
            } else {
                decode_lnms_f64(code, cfg, llr + f * n, (double *)P, (double *)R, hard + f * n, iters + f, converged + f);
                if (posterior) memcpy(posterior + f * n, P, sizeof(double) * n);
            }
        }
        free(P); free(R);
    }
    return 0;
}

/* ---------------------------------------------------------------- bit flipping, ldpc.py:82-135
 * bits = 0 if sample > 0 else 1 (ldpc.py:86-90); at most max_iters passes, each preceded by the
 * syndrome test on every row (ldpc.py:103); in a pass the first num_active_checks rows (int(self.m),
 * ldpc.py:113) vote +1 / -1 on each of their bits according to the parity of the check's other bits
 * (ldpc.py:123-128) and a bit flips when its total exceeds len(Ej) / 2 of the last row visited
 * (ldpc.py:130-132).  iters = passes made, converged = a syndrome test passed.
 */
int bp_oracle_bitflip(const oracle_code *code, const oracle_cfg *cfg, const double *samples, int64_t frames,
                      uint8_t *hard, int32_t *iters, uint8_t *converged, int nthreads)
{
    if (!code || !cfg || cfg->max_iters < 1) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    const int n = code->n, m = code->m;
    const int mact = cfg->num_active_checks < 0 || cfg->num_active_checks > m ? m : cfg->num_active_checks;
#pragma omp parallel
    {
        int *votes = malloc(sizeof(int) * (size_t)n);
        uint8_t *msg = malloc((size_t)n);
#pragma omp for schedule(dynamic, 4)
        for (int64_t f = 0; f < frames; f++) {
            uint8_t *b = hard + f * n;
            for (int i = 0; i < n; i++) b[i] = samples[f * n + i] > 0 ? 0 : 1;
            int t = 0, ok = 0, last_deg = 0;
            for (; t < cfg->max_iters; t++) {
                if (syndrome_ok(code, b)) { ok = 1; break; }
                memcpy(msg, b, (size_t)n);
                memset(votes, 0, sizeof(int) * (size_t)n);
                for (int j = 0; j < mact; j++) {
                    int e0 = code->row_ptr[j], e1 = code->row_ptr[j + 1], par = 0;
                    for (int e = e0; e < e1; e++) par ^= msg[code->col_idx[e]];
                    for (int e = e0; e < e1; e++) votes[code->col_idx[e]] += (par ^ msg[code->col_idx[e]]) ? 1 : -1;
                    last_deg = e1 - e0;
                }
                for (int i = 0; i < n; i++)
                    if (votes[i] > last_deg / 2.0) b[i] ^= 1;
            }
            iters[f] = t;
            converged[f] = (uint8_t)ok;
        }
        free(votes); free(msg);
    }
    return 0;
}

int bp_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
