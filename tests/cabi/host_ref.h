// Plain fp64 restatement of one training epoch and of the top-k contract, as the compiled ABI caller (cabi_fit.cpp) judges
// the GPU by.  TEST INFRASTRUCTURE: header-only, no HIP, no library - so that it also builds stand-alone with g++ under
// AddressSanitizer / UBSan (host_ref_selftest.cpp, tests/test_oracle_sanitized.py).
// Follows /root/reference/src/teamoflow/mf/matrix_factorization.py:130-176 (epoch), loss_graphs.py:47-52 (MSE),
// loss_graphs.py:74-88 (WMRB), SURVEY.md A.1-A.3 (closed forms).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <set>
#include <utility>
#include <vector>

namespace host_ref {

struct Lcg {
    uint64_t s;
    uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); }
    float unit() { return (next() & 0xffffff) / 16777216.0f; }
};

struct AdamT1 { float alpha, one_minus_b1, one_minus_b2, eps; };   // same fields as tmf_adam (include/tmf.h)

// fresh Keras-Adam step in fp32, as SURVEY.md A.1 spells it out
inline float adam_fresh(float w, float g, const AdamT1& a) {
    const float m = g * a.one_minus_b1, v = g * g * a.one_minus_b2;
    return w - (m * a.alpha) / (sqrtf(v) + a.eps);
}

// Interactions in CSR order (user ascending, item ascending inside a user); tables row-major with leading dimension ld.
struct Problem {
    int m, n, r, ld;
    std::vector<int64_t> rowptr_u;          // [m + 1]
    std::vector<int32_t> col_u, user_of;    // [nnz]
    std::vector<float> val_u;               // [nnz]
    int64_t nnz() const { return (int64_t)val_u.size(); }
};

inline double dot(const std::vector<double>& U, const std::vector<double>& V, int u, int j, int r, int ld) {
    double p = 0;
    for (int c = 0; c < r; ++c) p += U[(size_t)u * ld + c] * V[(size_t)j * ld + c];
    return p;
}

// MSE: loss = sum_k (a_k - p_k)^2; delta_k = -2 (a_k - p_k); gU[u] += delta_k V[j]; gV[j] += delta_k U[u] (pre-update tables).
// gU / gV are [rows, r] (no padding) and may be null (loss only).  Returns the SUM of the losses.
inline double mse_epoch(const Problem& P, const std::vector<double>& U, const std::vector<double>& V, std::vector<double>* gU,
                        std::vector<double>* gV) {
    if (gU) gU->assign((size_t)P.m * P.r, 0.0);
    if (gV) gV->assign((size_t)P.n * P.r, 0.0);
    double loss = 0;
    for (int64_t q = 0; q < P.nnz(); ++q) {
        const int u = P.user_of[q], j = P.col_u[q];
        const double e = P.val_u[q] - dot(U, V, u, j, P.r, P.ld), d = -2 * e;
        loss += e * e;
        if (gU)
            for (int c = 0; c < P.r; ++c) (*gU)[(size_t)u * P.r + c] += d * V[(size_t)j * P.ld + c];
        if (gV)
            for (int c = 0; c < P.r; ++c) (*gV)[(size_t)j * P.r + c] += d * U[(size_t)u * P.ld + c];
    }
    return loss;
}

// WMRB with the static negative table R [m, S] (any order inside a row), c = n_items / n_samples; positives = stored value > 0.
// x_ks = 1 - p_k + sp[u, s]; M_k = c sum_s max(x_ks, 0); loss_k = log(1 + M_k); term (k, s) active iff x_ks >= 0 (tf.maximum).
inline double wmrb_epoch(const Problem& P, const std::vector<int32_t>& R, int S, double cc, const std::vector<double>& U,
                         const std::vector<double>& V, std::vector<double>* gU, std::vector<double>* gV) {
    if (gU) gU->assign((size_t)P.m * P.r, 0.0);
    if (gV) gV->assign((size_t)P.n * P.r, 0.0);
    double loss = 0;
    std::vector<double> sps(S), Dd(S);
    for (int u = 0; u < P.m; ++u) {
        std::fill(Dd.begin(), Dd.end(), 0.0);
        for (int t = 0; t < S; ++t) sps[t] = dot(U, V, u, R[(size_t)u * S + t], P.r, P.ld);
        for (int64_t q = P.rowptr_u[u]; q < P.rowptr_u[u + 1]; ++q) {
            if (!(P.val_u[q] > 0)) continue;
            const int j = P.col_u[q];
            const double p = dot(U, V, u, j, P.r, P.ld);
            double M = 0;
            int cnt = 0;
            for (int t = 0; t < S; ++t) { const double x = 1.0 - p + sps[t]; if (x >= 0) { M += x; ++cnt; } }
            M *= cc;
            loss += log1p(M);
            const double w = cc / (1.0 + M), dk = -w * cnt;
            for (int t = 0; t < S; ++t) if (1.0 - p + sps[t] >= 0) Dd[t] += w;
            if (gU)
                for (int c = 0; c < P.r; ++c) (*gU)[(size_t)u * P.r + c] += dk * V[(size_t)j * P.ld + c];
            if (gV)
                for (int c = 0; c < P.r; ++c) (*gV)[(size_t)j * P.r + c] += dk * U[(size_t)u * P.ld + c];
        }
        for (int t = 0; t < S; ++t) {
            const int j = R[(size_t)u * S + t];
            if (gU)
                for (int c = 0; c < P.r; ++c) (*gU)[(size_t)u * P.r + c] += Dd[t] * V[(size_t)j * P.ld + c];
            if (gV)
                for (int c = 0; c < P.r; ++c) (*gV)[(size_t)j * P.r + c] += Dd[t] * U[(size_t)u * P.ld + c];
        }
    }
    return loss;
}

// Tables after a step against the restatement's gradient: the step is a near-sign function of g, so elements are compared
// only where |g| > 1e-3 max|g|; the restatement then continues from the engine's values (both trajectories stay on the same
// tables).  Returns the worst absolute difference.
inline double check_step(const std::vector<float>& Wn, std::vector<double>& Wr, const std::vector<double>& g, int rows, int r, int ld,
                         const AdamT1& adam) {
    double gmax = 0, worst = 0;
    for (double x : g) gmax = std::max(gmax, fabs(x));
    for (int i = 0; i < rows; ++i)
        for (int c = 0; c < r; ++c) {
            const double gi = g[(size_t)i * r + c];
            const float want = adam_fresh((float)Wr[(size_t)i * ld + c], (float)gi, adam);
            if (fabs(gi) > 1e-3 * gmax) worst = std::max(worst, (double)fabsf(Wn[(size_t)i * ld + c] - want));
            Wr[(size_t)i * ld + c] = Wn[(size_t)i * ld + c];
        }
    return worst;
}

// A top-k answer (ids, values) of every user against the fp64 scores: ids in range and distinct, values the scores of their ids
// (1e-5), descending, and nothing better left out.  Returns the number of violations.
inline int topk_violations(const std::vector<int32_t>& top, const std::vector<float>& topv, const std::vector<double>& U,
                           const std::vector<double>& V, int m, int n, int r, int ld, int k) {
    int wrong = 0;
    std::vector<double> sc(n);
    std::vector<char> taken(n);
    for (int u = 0; u < m; ++u) {
        for (int j = 0; j < n; ++j) sc[j] = dot(U, V, u, j, r, ld);
        std::fill(taken.begin(), taken.end(), 0);
        for (int t = 0; t < k; ++t) {
            const int j = top[(size_t)u * k + t];
            const bool in_range = j >= 0 && j < n;
            wrong += !in_range || taken[in_range ? j : 0] || fabs(topv[(size_t)u * k + t] - sc[in_range ? j : 0]) > 1e-5;
            if (in_range) taken[j] = 1;
            wrong += t > 0 && topv[(size_t)u * k + t] > topv[(size_t)u * k + t - 1];       // descending
        }
        const double kth = topv[(size_t)u * k + k - 1];
        for (int j = 0; j < n; ++j) wrong += !taken[j] && sc[j] > kth + 1e-5;              // nothing better was left out
    }
    return wrong;
}

// nnz distinct (user, item) pairs with values 1..5 in INSERTION order (arbitrary: the engine sorts), as [nnz, 2] int64 + values.
inline void draw_interactions(Lcg& rng, int m, int n, int nnz, std::vector<int64_t>& idx, std::vector<float>& val,
                              std::set<std::pair<int, int>>& seen) {
    while ((int)seen.size() < nnz) {
        const int u = rng.next() % m, j = rng.next() % n;
        if (seen.insert({u, j}).second) { idx.push_back(u); idx.push_back(j); val.push_back(1.f + rng.next() % 5); }
    }
}

// S distinct items per user (utils.py:20 draws without replacement), in random order.
inline std::vector<int32_t> draw_negatives(Lcg& rng, int m, int n, int S) {
    std::vector<int32_t> R((size_t)m * S);
    for (int u = 0; u < m; ++u) {
        std::set<int> pick;
        while ((int)pick.size() < S) pick.insert(rng.next() % n);
        std::vector<int> v(pick.begin(), pick.end());
        for (int q = S - 1; q > 0; --q) std::swap(v[q], v[rng.next() % (q + 1)]);
        for (int t = 0; t < S; ++t) R[(size_t)u * S + t] = v[t];
    }
    return R;
}

}  // namespace host_ref
