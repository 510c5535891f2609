// host_ref.h checked on its own (g++, AddressSanitizer + UBSan; tests/test_oracle_sanitized.py): the closed-form gradients
// against central differences of the loss written as a plain sum, the fresh-Adam step against its near-sign form, and the
// top-k checker against planted violations.  No GPU, no library.  Exit code 0 + "PASS" = every check passed.
#include "host_ref.h"

using namespace host_ref;

int main() {
    const int m = 60, n = 45, r = 7, ld = 8, S = 9;
    Lcg rng{777};
    std::set<std::pair<int, int>> seen;
    std::vector<int64_t> idx;
    std::vector<float> val;
    draw_interactions(rng, m, n, 400, idx, val, seen);
    // CSR order (what tmf_csr_build produces on the device)
    std::vector<int> order(val.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return std::make_pair(idx[2 * a], idx[2 * a + 1]) < std::make_pair(idx[2 * b], idx[2 * b + 1]); });
    Problem P{m, n, r, ld, std::vector<int64_t>(m + 1, 0), {}, {}, {}};
    for (int q : order) {
        P.user_of.push_back((int32_t)idx[2 * q]);
        P.col_u.push_back((int32_t)idx[2 * q + 1]);
        P.val_u.push_back(q % 11 == 0 ? -1.f : val[q]);   // a few non-positive values: WMRB must ignore them, MSE must not
        P.rowptr_u[idx[2 * q] + 1]++;
    }
    for (int u = 0; u < m; ++u) P.rowptr_u[u + 1] += P.rowptr_u[u];
    if (P.rowptr_u[m] != P.nnz()) { fprintf(stderr, "FAIL csr\n"); return 1; }
    std::vector<double> U((size_t)m * ld, 0.0), V((size_t)n * ld, 0.0);
    for (int i = 0; i < m; ++i) for (int c = 0; c < r; ++c) U[(size_t)i * ld + c] = 0.8 * (rng.unit() - 0.5);
    for (int i = 0; i < n; ++i) for (int c = 0; c < r; ++c) V[(size_t)i * ld + c] = 0.8 * (rng.unit() - 0.5);
    const std::vector<int32_t> R = draw_negatives(rng, m, n, S);
    for (int u = 0; u < m; ++u) {   // distinct and in range
        std::set<int> s(R.begin() + (size_t)u * S, R.begin() + (size_t)(u + 1) * S);
        if ((int)s.size() != S || *s.begin() < 0 || *s.rbegin() >= n) { fprintf(stderr, "FAIL negatives\n"); return 2; }
    }
    const double cc = (double)n / S;

    // ---- gradients against central differences ----
    std::vector<double> gU, gV;
    for (int which = 0; which < 2; ++which) {
        auto loss = [&](const std::vector<double>& U_, const std::vector<double>& V_) {
            return which == 0 ? mse_epoch(P, U_, V_, nullptr, nullptr) : wmrb_epoch(P, R, S, cc, U_, V_, nullptr, nullptr);
        };
        const double base = which == 0 ? mse_epoch(P, U, V, &gU, &gV) : wmrb_epoch(P, R, S, cc, U, V, &gU, &gV);
        if (!(fabs(base - loss(U, V)) <= 1e-12 * fabs(base))) { fprintf(stderr, "FAIL loss with / without gradients\n"); return 3; }
        double worst = 0, gmax = 0;
        for (double x : gU) gmax = std::max(gmax, fabs(x));
        for (double x : gV) gmax = std::max(gmax, fabs(x));
        const double h = 1e-6;
        for (int t = 0; t < 200; ++t) {
            const bool user_side = t % 2 == 0;
            const int row = rng.next() % (user_side ? m : n), c = rng.next() % r;
            std::vector<double>& W = user_side ? U : V;
            const double keep = W[(size_t)row * ld + c];
            W[(size_t)row * ld + c] = keep + h;
            const double lp = loss(U, V);
            W[(size_t)row * ld + c] = keep - h;
            const double lm = loss(U, V);
            W[(size_t)row * ld + c] = keep;
            const double fd = (lp - lm) / (2 * h), an = (user_side ? gU : gV)[(size_t)row * r + c];
            // (a hinge term crossing its kink inside +-h would show up here as an O(1) error: none does with these seeds)
            worst = std::max(worst, fabs(fd - an));
        }
        printf("%s: loss %.6f, worst |central difference - closed form| %.3g (max |g| %.3g)\n", which ? "WMRB" : "MSE", base, worst, gmax);
        if (!(worst < 2e-5 * gmax)) { fprintf(stderr, "FAIL gradient check\n"); return 4; }
    }

    // ---- fresh Adam: w - lr g / (|g| + 3.1623e-6) ----
    const float lr = 1e-2f;
    const AdamT1 a{lr * sqrtf(1.f - 0.999f) / (1.f - 0.9f), 1.f - 0.9f, 1.f - 0.999f, 1e-7f};
    for (float g : {1.f, -1.f, 1e-3f, -7.5f, 1e-6f, 0.f}) {
        const double want = 0.25 - (double)lr * g / (fabs((double)g) + 1e-7 / sqrt(0.001));
        if (fabs(adam_fresh(0.25f, g, a) - want) > 2e-7) { fprintf(stderr, "FAIL adam at g=%g\n", g); return 5; }
    }
    // check_step: the engine's value replaces the restatement's, differences are only counted where |g| is large
    {
        std::vector<double> Wr = {0.5, 0.5, 0.0, 0.0}, g = {1.0, 1e-9};
        std::vector<float> Wn = {adam_fresh(0.5f, 1.f, a), 123.f, 0.f, 0.f};
        const double w = check_step(Wn, Wr, g, 1, 2, 4, a);
        if (w != 0.0 || Wr[1] != 123.0) { fprintf(stderr, "FAIL check_step\n"); return 6; }
    }

    // ---- the top-k checker accepts the true answer and counts planted violations ----
    const int k = 5;
    std::vector<int32_t> top((size_t)m * k);
    std::vector<float> topv((size_t)m * k);
    for (int u = 0; u < m; ++u) {
        std::vector<std::pair<double, int>> sc(n);
        for (int j = 0; j < n; ++j) sc[j] = {-dot(U, V, u, j, r, ld), j};
        std::sort(sc.begin(), sc.end());
        for (int t = 0; t < k; ++t) { top[(size_t)u * k + t] = sc[t].second; topv[(size_t)u * k + t] = (float)-sc[t].first; }
    }
    if (topk_violations(top, topv, U, V, m, n, r, ld, k) != 0) { fprintf(stderr, "FAIL top-k checker rejects the truth\n"); return 7; }
    auto t2 = top;
    auto v2 = topv;
    t2[3] = t2[2];                                  // a repeated id
    if (topk_violations(t2, v2, U, V, m, n, r, ld, k) == 0) { fprintf(stderr, "FAIL repeated id not seen\n"); return 8; }
    t2 = top; t2[k] = n + 5;                        // out of range: must be counted, not dereferenced
    if (topk_violations(t2, v2, U, V, m, n, r, ld, k) == 0) { fprintf(stderr, "FAIL out-of-range id not seen\n"); return 9; }
    t2 = top; t2[2 * k] = -1;
    if (topk_violations(t2, v2, U, V, m, n, r, ld, k) == 0) { fprintf(stderr, "FAIL negative id not seen\n"); return 10; }
    v2 = topv; std::swap(v2[0], v2[k - 1]);         // not descending / wrong values
    if (topk_violations(top, v2, U, V, m, n, r, ld, k) == 0) { fprintf(stderr, "FAIL order not checked\n"); return 11; }
    printf("PASS\n");
    return 0;
}
