/*
 * CPU oracle, plain C: one-frame-at-a-time restatement of the reference's flooding
 * BP hot path (thadikari/ldpc_decoders), OpenMP over frames.
 *
 * TEST INFRASTRUCTURE ONLY -- used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker / CPU timing baseline.  It is never
 * linked into, loaded by, or called from the product library (libldpc_hip.so).
 *
 * Parity status: PINNED through oracle/bp_oracle.py (tests/test_oracle_golden.py
 * checks both against the reference's known-answer tests and the golden vectors
 * captured from the reference; the MSA and BEC paths must agree bit-for-bit, the
 * SPA path agrees up to libm-vs-numpy rounding of tanh/log/exp/atanh).
 *
 * Algorithm statements followed (reference file:line, relative to the upstream repo):
 *   flooding loop, exits, variable update, decision ..... src/bpa.py:17-63
 *   min-sum check rule (first-argmin gets min2) ......... src/bpa.py:86-102, src/math_utils.py:10,38-43,78-94
 *   tanh-product check rule (exp-sum-log, divide, atanh)  src/bpa.py:71-75,  src/math_utils.py:47-60
 *   ternary erasure decoder with stopping-set exit ...... src/bec.py:83-122
 *
 * Edge k = (chk[k], var[k]) in row-major order of H (np.where order, src/bpa.py:12).
 * The variable-side sum follows scipy's COO accumulation: from 0.0 in ascending
 * edge order, the prior is added last (src/bpa.py:35).
 *
 * The file is compiled twice through the REAL macro (double, float); the float
 * instance is the bit-exact checker of the GPU's fp32 min-sum mode (min-sum only
 * adds, subtracts and compares, so identical operation order => identical bits).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef ORACLE_TEMPLATE_PASS

typedef struct {
    int m, n;
    int64_t E;
    const int32_t *chk, *var;
    int32_t *row_ptr; /* [m+1]  edges of check c are row_ptr[c] .. row_ptr[c+1]-1      */
    int32_t *col_ptr; /* [n+1]                                                          */
    int32_t *col_edge; /* [E]   edges of variable v in ascending edge order              */
} graph_t;

static int graph_build(graph_t *g, int m, int n, int64_t E, const int32_t *chk, const int32_t *var) {
    g->m = m; g->n = n; g->E = E; g->chk = chk; g->var = var;
    g->row_ptr = (int32_t *)calloc((size_t)m + 1, sizeof(int32_t));
    g->col_ptr = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
    g->col_edge = (int32_t *)malloc((size_t)(E > 0 ? E : 1) * sizeof(int32_t));
    if (!g->row_ptr || !g->col_ptr || !g->col_edge) return -1;
    for (int64_t k = 0; k < E; ++k) {
        if (chk[k] < 0 || chk[k] >= m || var[k] < 0 || var[k] >= n) return -2;
        if (k && (chk[k] < chk[k - 1] || (chk[k] == chk[k - 1] && var[k] <= var[k - 1]))) return -3;
        g->row_ptr[chk[k] + 1]++;
        g->col_ptr[var[k] + 1]++;
    }
    for (int c = 0; c < m; ++c) g->row_ptr[c + 1] += g->row_ptr[c];
    for (int v = 0; v < n; ++v) g->col_ptr[v + 1] += g->col_ptr[v];
    int32_t *fill = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
    if (!fill) return -1;
    for (int64_t k = 0; k < E; ++k) g->col_edge[g->col_ptr[var[k]] + fill[var[k]]++] = (int32_t)k;
    free(fill);
    return 0;
}

static void graph_free(graph_t *g) {
    free(g->row_ptr); free(g->col_ptr); free(g->col_edge);
}

#define ORACLE_TEMPLATE_PASS 1
#define REAL double
#define SUF(x) x##_f64
#define R_TANH tanh
#define R_LOG log
#define R_EXP exp
#define R_ATANH atanh
#define R_FABS fabs
#define R_FMOD fmod
#include "bp_oracle.c"
#undef REAL
#undef SUF
#undef R_TANH
#undef R_LOG
#undef R_EXP
#undef R_ATANH
#undef R_FABS
#undef R_FMOD
#define REAL float
#define SUF(x) x##_f32
#define R_TANH tanhf
#define R_LOG logf
#define R_EXP expf
#define R_ATANH atanhf
#define R_FABS fabsf
#define R_FMOD fmodf
#include "bp_oracle.c"
#undef ORACLE_TEMPLATE_PASS

/* ---- ternary erasure decoder (integer only), src/bec.py:83-122 ------------------- */
static void bec_frame(const graph_t *g, const uint8_t *y, int max_iter, uint8_t *xhat, int32_t *iters,
                      int8_t *v2c, int8_t *c2v, int32_t *marg) {
    const int n = g->n, m = g->m;
    static const int8_t msg_of[3] = {-1, 1, 0}; /* bec.py:76 */
    for (int64_t k = 0; k < g->E; ++k) { v2c[k] = msg_of[y[g->var[k]]]; c2v[k] = 0; }
    memcpy(xhat, y, (size_t)n);
    int it = 0, sweeps = 0;
    for (;;) {
        if (max_iter > 0 && it >= max_iter) break;
        int erased_any = 0;
        for (int v = 0; v < n; ++v) erased_any |= (xhat[v] == 2);
        if (!erased_any) break;
        for (int c = 0; c < m; ++c) {
            int ne = 0, ones = 0;
            for (int k = g->row_ptr[c]; k < g->row_ptr[c + 1]; ++k) { ne += 1 - abs(v2c[k]); ones += v2c[k] > 0; }
            for (int k = g->row_ptr[c]; k < g->row_ptr[c + 1]; ++k) {
                if (ne == 0) c2v[k] = v2c[k];
                else if (ne > 1) c2v[k] = 0;
                else c2v[k] = (int8_t)((1 - abs(v2c[k])) * (2 * (ones % 2) - 1));
            }
        }
        int same = 1;
        for (int v = 0; v < n; ++v) {
            int s = 0;
            for (int j = g->col_ptr[v]; j < g->col_ptr[v + 1]; ++j) s += c2v[g->col_edge[j]];
            marg[v] = msg_of[y[v]] + s;
        }
        for (int64_t k = 0; k < g->E; ++k) { int d = marg[g->var[k]] - c2v[k]; v2c[k] = (int8_t)((d > 0) - (d < 0)); }
        for (int v = 0; v < n; ++v) {
            uint8_t s = marg[v] == 0 ? 2 : (marg[v] > 0 ? 1 : 0); /* bec.py:75,119 */
            marg[v] = s; same &= (s == xhat[v]);
        }
        ++sweeps;
        if (same) break; /* stopping set: the OLD x_hat is returned (bec.py:120) */
        for (int v = 0; v < n; ++v) xhat[v] = (uint8_t)marg[v];
        ++it;
    }
    *iters = sweeps;
}

int oracle_bec_decode(int m, int n, int64_t E, const int32_t *chk, const int32_t *var, const uint8_t *y, int64_t B,
                      int max_iter, uint8_t *xhat, int32_t *iters, int nthreads) {
    graph_t g; int rc = graph_build(&g, m, n, E, chk, var);
    if (rc) return rc;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        int8_t *v2c = (int8_t *)malloc((size_t)E + 1), *c2v = (int8_t *)malloc((size_t)E + 1);
        int32_t *marg = (int32_t *)malloc(((size_t)n + 1) * sizeof(int32_t));
#pragma omp for schedule(dynamic, 4)
        for (int64_t b = 0; b < B; ++b)
            bec_frame(&g, y + b * n, max_iter, xhat + b * n, iters + b, v2c, c2v, marg);
        free(v2c); free(c2v); free(marg);
    }
    graph_free(&g);
    return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

#else /* ================= ORACLE_TEMPLATE_PASS: instantiated for REAL ================= */

/* min-sum check rule, one row.  src/bpa.py:86-102 */
static void SUF(cn_msa)(const REAL *v2c, REAL *c2v, int deg) {
    int neg = 0, arg1 = 0;
    REAL min1 = INFINITY, min2 = INFINITY;
    for (int j = 0; j < deg; ++j) neg += v2c[j] < 0;
    for (int j = 0; j < deg; ++j) { REAL a = R_FABS(v2c[j]); if (a < min1) { min1 = a; arg1 = j; } }
    /* NaN magnitudes never compare below; arg1 stays at the first strict minimum = first arg-min */
    for (int j = 0; j < deg; ++j) { if (j == arg1) continue; REAL a = R_FABS(v2c[j]); if (a < min2) min2 = a; }
    const REAL rowsign = (neg & 1) ? (REAL)-1 : (REAL)1;
    for (int j = 0; j < deg; ++j) {
        const REAL own = v2c[j] >= 0 ? (REAL)1 : (REAL)-1; /* sgn(0)=+1, src/math_utils.py:10 */
        c2v[j] = (rowsign / own) * (j == arg1 ? min2 : min1);
    }
}

/* tanh-product check rule, one row.  src/bpa.py:71-75 */
static void SUF(cn_spa)(const REAL *v2c, REAL *c2v, int deg, REAL *t) {
    int neg = 0;
    REAL slog = 0;
    for (int j = 0; j < deg; ++j) { t[j] = R_TANH(v2c[j] / (REAL)2); neg += t[j] < 0; slog += R_LOG(R_FABS(t[j])); }
    const REAL prod = ((neg & 1) ? (REAL)-1 : (REAL)1) * R_EXP(slog);
    for (int j = 0; j < deg; ++j) {
        const REAL q = prod / t[j];
        c2v[j] = (REAL)2 * (R_FABS(q) == (REAL)1 ? (REAL)INFINITY * q : R_ATANH(q));
    }
}

static void SUF(bp_frame)(const graph_t *g, int alg, const REAL *y0, const REAL *prior, int max_iter, uint8_t *xhat,
                          int32_t *iters, REAL *v2c, REAL *c2v, REAL *marg, REAL *scratch) {
    const int n = g->n, m = g->m;
    for (int64_t k = 0; k < g->E; ++k) v2c[k] = prior[g->var[k]];
    int it = 0;
    for (;;) {
        if (max_iter > 0 && it >= max_iter) break;
        /* syndrome of the current word: raw y0 at iteration 0 (src/bpa.py:20,29), hard bits afterwards */
        int ok = 1;
        if (it == 0 && !y0) ok = 0;
        for (int c = 0; c < m && ok; ++c) {
            if (it == 0) {
                REAL s = 0;
                for (int k = g->row_ptr[c]; k < g->row_ptr[c + 1]; ++k) s += y0[g->var[k]];
                REAL r = R_FMOD(s, (REAL)2);
                ok = (r == 0);
            } else {
                int s = 0;
                for (int k = g->row_ptr[c]; k < g->row_ptr[c + 1]; ++k) s ^= xhat[g->var[k]];
                ok = !s;
            }
        }
        if (ok) break;
        for (int c = 0; c < m; ++c) {
            const int a = g->row_ptr[c], d = g->row_ptr[c + 1] - a;
            if (alg == 0) SUF(cn_msa)(v2c + a, c2v + a, d); else SUF(cn_spa)(v2c + a, c2v + a, d, scratch);
        }
        for (int v = 0; v < n; ++v) {
            REAL s = 0;
            for (int j = g->col_ptr[v]; j < g->col_ptr[v + 1]; ++j) s += c2v[g->col_edge[j]];
            marg[v] = prior[v] + s;
        }
        for (int64_t k = 0; k < g->E; ++k) v2c[k] = marg[g->var[k]] - c2v[k];
        for (int v = 0; v < n; ++v) { REAL mv = marg[v]; if (mv != mv) mv = 0; xhat[v] = mv < 0; }
        ++it;
    }
    *iters = it;
}

/* alg: 0 = MSA, 1 = SPA.  y0 may be NULL (no iteration-0 syndrome check: BI-AWGN).
 * xhat [B,n] bytes; when a frame leaves at iteration 0 (iters==0) xhat holds (y0 != 0). */
int SUF(oracle_bp_decode)(int m, int n, int64_t E, const int32_t *chk, const int32_t *var, int alg, const REAL *y0,
                          const REAL *priors, int64_t B, int max_iter, uint8_t *xhat, int32_t *iters, int nthreads) {
    graph_t g; int rc = graph_build(&g, m, n, E, chk, var);
    if (rc) return rc;
    int maxdeg = 1;
    for (int c = 0; c < m; ++c) if (g.row_ptr[c + 1] - g.row_ptr[c] > maxdeg) maxdeg = g.row_ptr[c + 1] - g.row_ptr[c];
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        REAL *v2c = (REAL *)malloc(((size_t)E + 1) * sizeof(REAL)), *c2v = (REAL *)malloc(((size_t)E + 1) * sizeof(REAL));
        REAL *marg = (REAL *)malloc(((size_t)n + 1) * sizeof(REAL)), *scr = (REAL *)malloc((size_t)maxdeg * sizeof(REAL));
#pragma omp for schedule(dynamic, 4)
        for (int64_t b = 0; b < B; ++b) {
            const REAL *yb = y0 ? y0 + b * n : NULL;
            if (yb) for (int v = 0; v < n; ++v) xhat[b * n + v] = yb[v] != 0;
            SUF(bp_frame)(&g, alg, yb, priors + b * n, max_iter, xhat + b * n, iters + b, v2c, c2v, marg, scr);
        }
        free(v2c); free(c2v); free(marg); free(scr);
    }
    graph_free(&g);
    return 0;
}

#endif
