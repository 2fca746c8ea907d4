/*
 * admm_oracle.c -- CPU restatement of the reference's ADMM LP decoder.  TEST INFRASTRUCTURE ONLY: used by tests/ and
 * oracle/ scripts as the checker, never by the product package.
 *
 *   ADMM iteration ............ reference src/admm.py:42-69   (x / z / lambda updates, stopping rule, iteration counter)
 *   stopping test ............. reference src/admm.py:17-23   (two squared-distance sums against eps^2 * nnz(H))
 *   parity-polytope projection  reference src/parity_polytope/projection.cpp:30-249 (sort, clip, even floor, water-filling)
 *
 * Pinned: (1) pp_project against the reference's own projection.cpp compiled as oracle/_ref/libppolytope.so (bit-exact on
 * random vectors, tests/test_oracle_admm.py); (2) the whole loop against golden vectors captured from the reference
 * (oracle/make_goldens_admm.py -> tests/golden/admm_*.npz).
 *
 * Arithmetic notes (what makes the restatement bit-exact):
 *   - column sums are scipy COO sums: from +0.0, ascending edge order (as in bp_oracle.c);
 *   - the two stopping sums are numpy .sum() of a contiguous float64 vector: pairwise summation (np_sum below);
 *   - every other expression is elementwise fp64 in the order the reference writes it (divisions stay divisions).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PP_MAX_LEN 16 /* std::sort is a stable insertion sort up to 16 elements; beyond that its tie order is unspecified */

/* numpy's pairwise summation of n contiguous doubles (numpy/core/src/umath/loops_utils.h.src), + the reduction's 0.0 start */
static double np_pairwise(const double* a, int64_t n) {
    if (n < 8) {
        double res = -0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}
double oracle_np_sum(const double* a, int64_t n) { return 0.0 + np_pairwise(a, n); }

static double clamp01(double x) {
    const double lo = (x < 0.0) ? 0.0 : x; /* std::max(x, 0.0) */
    return (1.0 < lo) ? 1.0 : lo;          /* std::min(lo, 1.0) */
}

/* Euclidean projection of v[0..len) onto the parity polytope (even-weight vertices of the unit cube). */
int oracle_pp_project(int len, const double* v, double* out) {
    if (len < 1 || len > PP_MAX_LEN) return -1;
    int none_positive = 1, all_above_one = 1;
    for (int i = 0; i < len; ++i) {
        if (v[i] > 0) none_positive = 0;
        if (v[i] <= 1) all_above_one = 0;
    }
    if (none_positive) {
        for (int i = 0; i < len; ++i) out[i] = 0;
        return 0;
    }
    if (all_above_one && len % 2 == 0) {
        for (int i = 0; i < len; ++i) out[i] = 1;
        return 0;
    }
    /* stable sort, decreasing */
    double s[PP_MAX_LEN];
    int who[PP_MAX_LEN];
    for (int i = 0; i < len; ++i) {
        const double val = v[i];
        int j = i;
        while (j > 0 && val > s[j - 1]) {
            s[j] = s[j - 1];
            who[j] = who[j - 1];
            --j;
        }
        s[j] = val;
        who[j] = i;
    }
    /* cube projection and the even number r of leading ones of the candidate facet */
    double c[PP_MAX_LEN], mass = 0;
    for (int i = 0; i < len; ++i) {
        c[i] = clamp01(s[i]);
        mass += c[i];
    }
    int r = (int)floor(mass);
    if (r & 1) --r;
    double facet = 0;
    for (int i = 0; i < r + 1 && i < len; ++i) facet += c[i]; /* r == len: upstream reads one element past its arrays (projection.cpp:79-80); taken as 0 */
    for (int i = r + 1; i < len; ++i) facet -= c[i];
    if (facet <= r) { /* the cube projection already satisfies the facet inequality */
        for (int i = 0; i < len; ++i) out[who[i]] = c[i];
        return 0;
    }
    const double beta_cap = (r + 2 <= len) ? (s[r] - s[r + 1]) / 2 : s[r];
    /* break points of the piecewise-linear constraint in beta, ascending: s[i]-1 for i = r..0 merged with -s[i] for i = r+1.. */
    double bp[PP_MAX_LEN];
    int bp_who[PP_MAX_LEN];
    {
        int L = r, R = r + 1, k = 0;
        while (k < len) {
            if (L < 0) {
                for (; k < len; ++k, ++R) { bp_who[k] = R; bp[k] = -s[R]; }
                break;
            }
            if (R >= len) {
                for (; k < len; ++k, --L) { bp_who[k] = L; bp[k] = s[L] - 1; }
                break;
            }
            const double a = s[L] - 1, b = -s[R];
            if (a > b) { bp_who[k] = R; bp[k] = b; ++R; } else { bp_who[k] = L; bp[k] = a; --L; }
            ++k;
        }
    }
    const double tol = 1e-10;
    int n_clipped = 0, n_nonneg = 0; /* entries above 1 / entries not below 0 at beta = 0 */
    for (int i = 0; i < len; ++i) {
        if (s[i] > 1) ++n_clipped;
        if (s[i] >= 0 - tol) ++n_nonneg;
    }
    int clip = n_clipped - 1, zero = n_nonneg;
    int first = 0, last = 0;
    for (int i = 0; i < len; ++i) {
        if (bp[i] < 0 + tol) ++first;
        if (bp[i] < beta_cap) ++last;
    }
    --last;
    double active = 0;
    for (int i = 0; i < len; ++i) {
        if (i > clip && i <= r) active += s[i];
        if (i > r && i < zero) active -= s[i];
    }
    double total = active + clip + 1;
    int prev_clip = clip, prev_zero = zero, fresh = 1;
    double prev_active = active, beta = 0;
    for (int i = first; i <= last; ++i) {
        if (fresh) {
            prev_clip = clip;
            prev_zero = zero;
            prev_active = active;
        }
        fresh = 0;
        beta = bp[i];
        if (bp_who[i] <= r) {
            --clip;
            active += s[bp_who[i]];
        } else {
            ++zero;
            active -= s[bp_who[i]];
        }
        if (i < len - 1) {
            if (beta != bp[i + 1]) {
                total = (clip + 1) + active - beta * (zero - clip - 1);
                fresh = 1;
                if (total < r) break;
            }
        } else if (i == len - 1) {
            total = (clip + 1) + active - beta * (zero - clip - 1);
            fresh = 1;
        }
    }
    if (total > r)
        beta = -(r - clip - 1 - active) / (zero - clip - 1);
    else
        beta = -(r - prev_clip - 1 - prev_active) / (prev_zero - prev_clip - 1);
    for (int i = 0; i < len; ++i) out[who[i]] = clamp01(i <= r ? s[i] - beta : s[i] + beta);
    return 0;
}

/* ADMM_Base.decode for B frames.  gamma [B,n] LLRs; x_out [B,n] = x_hat at return (before pseudo_to_cw);
 * iters [B] = iter_count at return; converged [B] = 1 if left through the stopping test.  max_iter <= 0: no cap (bounded by hard_cap). */
int oracle_admm_decode(int m, int n, int64_t E, const int32_t* chk, const int32_t* var, const double* gamma, int64_t B, double mu,
                       double eps, int max_iter, int hard_cap, double* x_out, int32_t* iters, uint8_t* converged) {
    int32_t* row_ptr = (int32_t*)calloc((size_t)m + 1, sizeof(int32_t));
    int32_t* col_ptr = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
    int32_t* col_edge = (int32_t*)malloc((size_t)E * sizeof(int32_t));
    int64_t nnz = E;
    for (int64_t k = 0; k < E; ++k) {
        row_ptr[chk[k] + 1]++;
        col_ptr[var[k] + 1]++;
    }
    for (int i = 0; i < m; ++i) row_ptr[i + 1] += row_ptr[i];
    for (int i = 0; i < n; ++i) col_ptr[i + 1] += col_ptr[i];
    {
        int32_t* fill = (int32_t*)malloc((size_t)n * sizeof(int32_t));
        memcpy(fill, col_ptr, (size_t)n * sizeof(int32_t));
        for (int64_t k = 0; k < E; ++k) col_edge[fill[var[k]]++] = (int32_t)k;
        free(fill);
    }
    for (int i = 0; i < m; ++i)
        if (row_ptr[i + 1] - row_ptr[i] > PP_MAX_LEN) return -2;
    const double thresh = (eps * eps) * (double)nnz; /* (eps ** 2) * parity_mtx.sum() */
    int rc = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t f = 0; f < B; ++f) {
        double* z = (double*)malloc((size_t)E * sizeof(double));
        double* znew = (double*)malloc((size_t)E * sizeof(double));
        double* lam = (double*)calloc((size_t)E, sizeof(double));
        double* d1 = (double*)malloc((size_t)E * sizeof(double));
        double* d2 = (double*)malloc((size_t)E * sizeof(double));
        double* x = x_out + f * n;
        const double* g = gamma + f * n;
        for (int64_t k = 0; k < E; ++k) z[k] = 0.5;
        for (int v = 0; v < n; ++v) x[v] = 0.0; /* overwritten in the first iteration */
        int it = 0, conv = 0;
        for (;;) {
            if (max_iter > 0 && it >= max_iter) break;
            if (it >= hard_cap) break;
            for (int v = 0; v < n; ++v) { /* x update */
                double s = 0.0;
                for (int p = col_ptr[v]; p < col_ptr[v + 1]; ++p) {
                    const int32_t k = col_edge[p];
                    s += z[k] - lam[k] / mu;
                }
                x[v] = clamp01((s - g[v] / mu) / (double)(col_ptr[v + 1] - col_ptr[v]));
            }
            for (int c = 0; c < m; ++c) { /* z update: one projection per check */
                const int k0 = row_ptr[c], len = row_ptr[c + 1] - row_ptr[c];
                double vv[PP_MAX_LEN];
                for (int j = 0; j < len; ++j) vv[j] = x[var[k0 + j]] + lam[k0 + j] / mu;
                oracle_pp_project(len, vv, znew + k0);
            }
            for (int64_t k = 0; k < E; ++k) { /* lambda update + the two distance vectors */
                const double xk = x[var[k]];
                lam[k] = lam[k] + mu * (xk - znew[k]);
                const double a = xk - znew[k], b = z[k] - znew[k];
                d1[k] = a * a;
                d2[k] = b * b;
            }
            const double aa1 = oracle_np_sum(d1, E), aa2 = oracle_np_sum(d2, E);
            if (aa1 < thresh && aa2 < thresh) {
                conv = 1;
                break;
            }
            double* t = z;
            z = znew;
            znew = t;
            ++it;
        }
        iters[f] = it;
        if (converged) converged[f] = (uint8_t)conv;
        free(z); free(znew); free(lam); free(d1); free(d2);
    }
    free(row_ptr); free(col_ptr); free(col_edge);
    return rc;
}
