/* Sparse closed-form CPU restatement of one training epoch in C with OpenMP.
 *
 * TEST INFRASTRUCTURE (see oracle/__init__.py) - parity unpinned except gather_matrix_indices:
 * TensorFlow cannot be installed here, so nothing under oracle/ has been run against the reference
 * itself.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (teamoflow_amd/) never does.
 *
 * Same algebra as oracle/sparse_ref.py, which is checked against the
 * dense autograd restatement oracle/dense_ref.py.  What each block follows in the reference:
 *   MSE   /root/reference/src/teamoflow/mf/loss_graphs.py:47-52   l_k = (a_k - p_k)^2 on every stored entry
 *   WMRB  loss_graphs.py:74-88, matrix_factorization.py:153-154   l_k = log(1 + c * sum_s max(1 - p_k + sp[u,s], 0)),
 *         c = n_items / n_samples, positives (a_k > 0) only, tf.maximum subgradient 1 at x == 0
 *   step  matrix_factorization.py:170-176   gradient of the SUM of l_k, fresh Keras Adam (t = 1) every epoch,
 *         both tables updated from the PRE-update U and V
 * This file is the multi-threaded CPU baseline SURVEY.md 8(d) asks for at C4/C5 ("sparse CPU restatement,
 * not the reference formulation") and a third, independently written oracle for the tests.
 *
 * Arithmetic: scores, hinge terms and the Adam step in fp32 (the reference's dtype); the sum of a positive's hinge
 * terms and the row sums of the gradients accumulate in fp64 and are rounded once (a CPU can afford it, and it puts this oracle
 * nearer the exact closed form than any fp32 summation order).
 * Build: gcc -O3 -fopenmp -ffp-contract=off -mavx2 -shared -fPIC (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

typedef struct {
  float alpha;      /* lr * sqrt(1 - b2) / (1 - b1), evaluated in fp32 by the caller */
  float one_m_b1;   /* 1 - 0.9   in fp32 */
  float one_m_b2;   /* 1 - 0.999 in fp32 */
  float eps;        /* 1e-7 */
} oracle_adam;

static inline float adam_fresh(float w, float g, const oracle_adam* a) {
  /* m = (g - 0)(1 - b1); v = (g*g - 0)(1 - b2); w -= m*alpha / (sqrt(v) + eps)    */
  float m = g * a->one_m_b1;
  float v = (g * g) * a->one_m_b2;
  return w - (m * a->alpha) / (sqrtf(v) + a->eps);
}

static inline float dotf(const float* x, const float* y, int r) {
  float acc = 0.f;
#pragma omp simd reduction(+ : acc)
  for (int c = 0; c < r; ++c) acc += x[c] * y[c];
  return acc;
}

static inline float absdotf(const float* x, const float* y, int r) {
  float acc = 0.f;
#pragma omp simd reduction(+ : acc)
  for (int c = 0; c < r; ++c) acc += fabsf(x[c] * y[c]);
  return acc;
}

static inline void axpy_d(double* acc, float a, const float* x, int r) {
#pragma omp simd
  for (int c = 0; c < r; ++c) acc[c] += (double)a * (double)x[c];
}

int oracle_threads(void) { return omp_get_max_threads(); }
void oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

/* Stable counting sort of `count` keys in [0, nkeys): ptr[nkeys + 1] and the entry order grouped by key
 * (ascending entry index inside a key).  Used for the item-side lists (CSC of the interactions, inverse
 * index of the negative table); built once per fit, outside the timed epoch - like the engine's plans. */
int oracle_group_by_key(int64_t count, int64_t nkeys, const int32_t* key, int64_t* ptr, int64_t* order) {
  memset(ptr, 0, (size_t)(nkeys + 1) * sizeof(int64_t));
  for (int64_t e = 0; e < count; ++e) {
    if (key[e] < 0 || key[e] >= nkeys) return -1;
    ptr[key[e] + 1]++;
  }
  for (int64_t k = 0; k < nkeys; ++k) ptr[k + 1] += ptr[k];
  int64_t* cur = (int64_t*)malloc((size_t)(nkeys > 0 ? nkeys : 1) * sizeof(int64_t));
  if (!cur) return -2;
  memcpy(cur, ptr, (size_t)nkeys * sizeof(int64_t));
  for (int64_t e = 0; e < count; ++e) order[cur[key[e]]++] = e;
  free(cur);
  return 0;
}

/* One MSE epoch.  Interactions are CSR by user (rowptr[m+1], col[nnz], val[nnz]); colptr/centry is their
 * grouping by item from oracle_group_by_key(nnz, n, col, ...); ent_user[nnz] = user of each entry.
 * Outputs: U_new [m,r], V_new [n,r], loss_sum (sum of l_k in fp64), optional raw gradients gU/gV (fp32). */
int oracle_mse_epoch(int64_t m, int64_t n, int r, const int64_t* rowptr, const int32_t* col, const float* val,
                     const int32_t* ent_user, const int64_t* colptr, const int64_t* centry,
                     const float* U, const float* V, const oracle_adam* adam,
                     float* U_new, float* V_new, double* loss_sum, float* delta, float* gU, float* gV) {
  double total = 0.0;
#pragma omp parallel
  {
    double* acc = (double*)malloc((size_t)r * sizeof(double));
#pragma omp for schedule(dynamic, 64) reduction(+ : total)
    for (int64_t u = 0; u < m; ++u) {
      const float* Uu = U + u * r;
      for (int c = 0; c < r; ++c) acc[c] = 0.0;
      for (int64_t k = rowptr[u]; k < rowptr[u + 1]; ++k) {
        const float* Vj = V + (int64_t)col[k] * r;
        float e = val[k] - dotf(Uu, Vj, r);
        float d = -2.0f * e;
        total += (double)(e * e);
        delta[k] = d;
        axpy_d(acc, d, Vj, r);
      }
      for (int c = 0; c < r; ++c) {
        float g = (float)acc[c];
        if (gU) gU[u * r + c] = g;
        U_new[u * r + c] = adam_fresh(Uu[c], g, adam);
      }
    }
#pragma omp for schedule(dynamic, 16)
    for (int64_t j = 0; j < n; ++j) {
      for (int c = 0; c < r; ++c) acc[c] = 0.0;
      for (int64_t q = colptr[j]; q < colptr[j + 1]; ++q) {
        int64_t k = centry[q];
        axpy_d(acc, delta[k], U + (int64_t)ent_user[k] * r, r);
      }
      for (int c = 0; c < r; ++c) {
        float g = (float)acc[c];
        if (gV) gV[j * r + c] = g;
        V_new[j * r + c] = adam_fresh(V[j * r + c], g, adam);
      }
    }
    free(acc);
  }
  *loss_sum = total;
  return 0;
}

/* One WMRB epoch.  R [m,S] int32 is the static negative table; negptr/negentry its grouping by item from
 * oracle_group_by_key(m*S, n, R, ...) (entry e = u*S + s).  Work buffers: delta [nnz], D [m,S].
 * loss_sum = sum over positives of l_k; n_pos = number of positives. */
int oracle_wmrb_epoch(int64_t m, int64_t n, int r, int64_t S, float c_ratio,
                      const int64_t* rowptr, const int32_t* col, const float* val, const int32_t* ent_user,
                      const int64_t* colptr, const int64_t* centry,
                      const int32_t* R, const int64_t* negptr, const int64_t* negentry,
                      const float* U, const float* V, const oracle_adam* adam,
                      float* U_new, float* V_new, double* loss_sum, int64_t* n_pos,
                      float* delta, float* D, float* gU, float* gV) {
  double total = 0.0;
  int64_t positives = 0;
#pragma omp parallel
  {
    double* acc = (double*)malloc((size_t)r * sizeof(double));
    float* sp = (float*)malloc((size_t)(S > 0 ? S : 1) * sizeof(float));
#pragma omp for schedule(dynamic, 16) reduction(+ : total, positives)
    for (int64_t u = 0; u < m; ++u) {
      const float* Uu = U + u * r;
      const int32_t* Ru = R + u * S;
      float* Du = D + u * S;
      for (int64_t s = 0; s < S; ++s) {
        sp[s] = dotf(Uu, V + (int64_t)Ru[s] * r, r);
        Du[s] = 0.f;
      }
      for (int c = 0; c < r; ++c) acc[c] = 0.0;
      for (int64_t k = rowptr[u]; k < rowptr[u + 1]; ++k) {
        delta[k] = 0.f;
        if (!(val[k] > 0.f)) continue;                 /* loss_graphs.py:76-77: positives only */
        const float* Vj = V + (int64_t)col[k] * r;
        float base = 1.0f - dotf(Uu, Vj, r);
        double hinge = 0.0;                            /* fp32 terms as the reference forms them, summed in fp64 */
        int cnt = 0;
#pragma omp simd reduction(+ : hinge, cnt)
        for (int64_t s = 0; s < S; ++s) {
          float x = base + sp[s];
          hinge += x > 0.f ? (double)x : 0.0;
          cnt += x >= 0.f;                             /* tf.maximum: gradient to x when x >= 0 */
        }
        float M = c_ratio * (float)hinge;
        float w = c_ratio / (1.0f + M);
        total += (double)logf(1.0f + M);
        positives++;
        float d = -w * (float)cnt;
        delta[k] = d;
#pragma omp simd
        for (int64_t s = 0; s < S; ++s) Du[s] += (base + sp[s] >= 0.f) ? w : 0.f;
        axpy_d(acc, d, Vj, r);
      }
      for (int64_t s = 0; s < S; ++s)
        if (Du[s] != 0.f) axpy_d(acc, Du[s], V + (int64_t)Ru[s] * r, r);
      for (int c = 0; c < r; ++c) {
        float g = (float)acc[c];
        if (gU) gU[u * r + c] = g;
        U_new[u * r + c] = adam_fresh(Uu[c], g, adam);
      }
    }
#pragma omp for schedule(dynamic, 16)
    for (int64_t j = 0; j < n; ++j) {
      for (int c = 0; c < r; ++c) acc[c] = 0.0;
      for (int64_t q = colptr[j]; q < colptr[j + 1]; ++q) {
        int64_t k = centry[q];
        if (delta[k] != 0.f) axpy_d(acc, delta[k], U + (int64_t)ent_user[k] * r, r);
      }
      for (int64_t q = negptr[j]; q < negptr[j + 1]; ++q) {
        int64_t e = negentry[q];
        if (D[e] != 0.f) axpy_d(acc, D[e], U + (e / S) * r, r);
      }
      for (int c = 0; c < r; ++c) {
        float g = (float)acc[c];
        if (gV) gV[j * r + c] = g;
        V_new[j * r + c] = adam_fresh(V[j * r + c], g, adam);
      }
    }
    free(acc);
    free(sp);
  }
  *loss_sum = total;
  *n_pos = positives;
  return 0;
}

/* How far the WMRB gradient may move when the predictions move by the parity tolerance.
 *
 * The hinge max(1 - p_k + sp[u,s], 0) has a kink at 0 (loss_graphs.py:83-84): a term whose argument x lies within the
 * tolerance the predictions themselves are compared at is active in one valid fp32 evaluation of the reference and
 * inactive in another (different summation order of the r-wide dot products):
 *     |x| <= tol_rel * (1 + sum_c |U[u,c] V[j_k,c]| + sum_c |U[u,c] V[R[u,s],c]|)
 * (an fp32 dot product of length r is only determined up to r * 2^-24 times the sum of the magnitudes of its terms, so
 * callers pass tol_rel = max(1e-5, 2 r 2^-24)).  Switching such a term moves D[u,s] by w_k, delta_k by -w_k, gU[u] by w_k (V[R[u,s]] - V[j_k]) and the two
 * item rows by +-w_k U[u].  This routine adds up those possible moves ("slack") per element, so that a test can state
 * "equal to the closed form up to the activity of the boundary terms" instead of an unconditional 1e-5, which no
 * fp32 implementation (the reference included) can meet once training has pushed hinge arguments onto the kink.
 * Outputs (all optional but D_slack): D_slack [m,S], delta_slack [nnz], gU_slack [m,r], gV_slack [n,r]; returns the
 * number of boundary (k, s) pairs, or -1. */
int64_t oracle_wmrb_boundary_slack(int64_t m, int64_t n, int r, int64_t S, float c_ratio, float tol_rel,
                                   const int64_t* rowptr, const int32_t* col, const float* val, const int32_t* R,
                                   const float* U, const float* V,
                                   float* D_slack, float* delta_slack, float* gU_slack, float* gV_slack) {
  (void)n;
  int64_t pairs = 0;
#pragma omp parallel
  {
    float* sp = (float*)malloc((size_t)(S > 0 ? S : 1) * sizeof(float));
    float* asp = (float*)malloc((size_t)(S > 0 ? S : 1) * sizeof(float));
#pragma omp for schedule(dynamic, 16) reduction(+ : pairs)
    for (int64_t u = 0; u < m; ++u) {
      const float* Uu = U + u * r;
      const int32_t* Ru = R + u * S;
      for (int64_t s = 0; s < S; ++s) {
        sp[s] = dotf(Uu, V + (int64_t)Ru[s] * r, r);
        asp[s] = absdotf(Uu, V + (int64_t)Ru[s] * r, r);
      }
      for (int64_t k = rowptr[u]; k < rowptr[u + 1]; ++k) {
        if (!(val[k] > 0.f)) continue;
        const float* Vj = V + (int64_t)col[k] * r;
        float p = dotf(Uu, Vj, r);
        float ap = 1.0f + absdotf(Uu, Vj, r);
        float base = 1.0f - p;
        float hinge = 0.f;
        int near = 0;
        for (int64_t s = 0; s < S; ++s) {
          float x = base + sp[s];
          hinge += x > 0.f ? x : 0.f;
          near += fabsf(x) <= tol_rel * (ap + asp[s]);
        }
        if (!near) continue;
        float w = c_ratio / (1.0f + c_ratio * hinge);
        for (int64_t s = 0; s < S; ++s) {
          float x = base + sp[s];
          if (!(fabsf(x) <= tol_rel * (ap + asp[s]))) continue;
          pairs++;
          const float* Vs = V + (int64_t)Ru[s] * r;
          D_slack[u * S + s] += w;               /* row u belongs to this thread */
          if (delta_slack) delta_slack[k] += w;
          for (int c = 0; c < r; ++c) {
            if (gU_slack) gU_slack[u * r + c] += w * fabsf(Vs[c] - Vj[c]);
            if (gV_slack) {
              float a = w * fabsf(Uu[c]);
#pragma omp atomic
              gV_slack[(int64_t)Ru[s] * r + c] += a;
#pragma omp atomic
              gV_slack[(int64_t)col[k] * r + c] += a;
            }
          }
        }
      }
    }
    free(sp);
    free(asp);
  }
  return pairs;
}

