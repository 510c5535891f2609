// A compiled caller of the C ABI (include/tmf.h) with no Python and no torch in the process: builds the index
// structures, runs MSE epochs (matrix_factorization.py:130-176 with MSELoss), a fused predict + top-k and WMRB epochs
// (the sliced scores / hinge / gradU / finish kernels and the item-side gather-sum) on the GPU
// through libtmf.so, and checks the results against the plain fp64 restatement of tests/cabi/host_ref.h (test infrastructure;
// the same header is built stand-alone under AddressSanitizer / UBSan by tests/test_oracle_sanitized.py).
// Built by tests/cabi/Makefile (hipcc; only the HIP runtime API is used on the host side), run by
// tests/test_gpu_cabi.py.  Exit code 0 = every check passed.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <set>
#include <utility>
#include <vector>

#include "host_ref.h"
#include "tmf.h"

#define HIP_OK(x)                                                                         \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
    } while (0)
#define TMF_OK_(x)                                                                        \
    do {                                                                                  \
        int rc_ = (x);                                                                    \
        if (rc_ != TMF_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, tmf_last_error()); return 3; } \
    } while (0)

template <typename T>
static T* dev_copy(const std::vector<T>& h) {
    T* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)) != hipSuccess) return nullptr;
    if (!h.empty() && hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}
template <typename T>
static std::vector<T> host_copy(const T* d, size_t n) {
    std::vector<T> h(n);
    if (n) (void)hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost);
    return h;
}

using host_ref::Lcg;

static host_ref::AdamT1 ref_adam(const tmf_adam& a) { return host_ref::AdamT1{a.alpha, a.one_minus_b1, a.one_minus_b2, a.eps}; }

int main() {
    const int m = 300, n = 200, r = 24, epochs = 5, k = 5;
    const float lr = 1e-3f;
    const int ld = tmf_padded_ld(r);
    if (tmf_version() < 203 || ld < r) { fprintf(stderr, "library version %d, ld %d\n", tmf_version(), ld); return 1; }
    Lcg rng{12345};
    std::set<std::pair<int, int>> seen;
    std::vector<int64_t> idx;
    std::vector<float> val;
    host_ref::draw_interactions(rng, m, n, 3000, idx, val, seen);
    const int64_t nnz = (int64_t)val.size();   // in insertion (i.e. arbitrary) order: tmf_csr_build sorts
    std::vector<float> U((size_t)m * ld, 0.f), V((size_t)n * ld, 0.f);
    for (int i = 0; i < m; ++i) for (int c = 0; c < r; ++c) U[(size_t)i * ld + c] = 0.3f * (rng.unit() - 0.5f);
    for (int i = 0; i < n; ++i) for (int c = 0; c < r; ++c) V[(size_t)i * ld + c] = 0.3f * (rng.unit() - 0.5f);

    // ---- index structures on the device ----
    int64_t *d_idx = dev_copy(idx), *d_rowptr_u, *d_rowptr_i, *d_perm;
    float *d_val = dev_copy(val), *d_val_u;
    int32_t *d_col_u, *d_user_of;
    HIP_OK(hipMalloc(&d_rowptr_u, (m + 1) * 8)); HIP_OK(hipMalloc(&d_rowptr_i, (n + 1) * 8)); HIP_OK(hipMalloc(&d_perm, nnz * 8));
    HIP_OK(hipMalloc(&d_val_u, nnz * 4)); HIP_OK(hipMalloc(&d_col_u, nnz * 4)); HIP_OK(hipMalloc(&d_user_of, nnz * 4));
    size_t ws_bytes = std::max(tmf_csr_build_workspace_bytes(nnz), tmf_stable_order_workspace_bytes(nnz));
    void* d_ws;
    HIP_OK(hipMalloc(&d_ws, ws_bytes));
    TMF_OK_(tmf_csr_build(d_idx, d_val, nnz, m, n, d_rowptr_u, d_col_u, d_val_u, d_user_of, d_ws, ws_bytes, nullptr));
    TMF_OK_(tmf_csc_perm(d_col_u, nnz, n, d_rowptr_i, d_perm, d_ws, ws_bytes, nullptr));
    HIP_OK(hipDeviceSynchronize());
    const auto rowptr_u = host_copy(d_rowptr_u, m + 1), rowptr_i = host_copy(d_rowptr_i, n + 1), perm = host_copy(d_perm, nnz);
    const auto col_u = host_copy(d_col_u, nnz), user_of = host_copy(d_user_of, nnz);
    const auto val_u = host_copy(d_val_u, nnz);
    const host_ref::Problem P{m, n, r, ld, rowptr_u, col_u, user_of, val_u};   // the restatement works on the CSR the device built (checked below)
    int bad = 0;
    for (int u = 0; u < m; ++u)     // CSR: users ascending, items ascending inside a user, every pair present
        for (int64_t q = rowptr_u[u]; q < rowptr_u[u + 1]; ++q) {
            bad += user_of[q] != u || !seen.count({u, col_u[q]});
            bad += q > rowptr_u[u] && col_u[q - 1] >= col_u[q];
        }
    bad += rowptr_u[m] != nnz || rowptr_i[n] != nnz;
    std::vector<int32_t> row_i(nnz);
    std::vector<float> val_i(nnz);
    for (int64_t q = 0; q < nnz; ++q) { row_i[q] = user_of[perm[q]]; val_i[q] = val_u[perm[q]]; }
    for (int j = 0; j < n; ++j)
        for (int64_t q = rowptr_i[j]; q < rowptr_i[j + 1]; ++q) bad += col_u[perm[q]] != j;
    if (bad) { fprintf(stderr, "FAIL index structures: %d violations\n", bad); return 10; }
    int32_t *d_row_i = dev_copy(row_i);
    float* d_val_i = dev_copy(val_i);

    // one segment per row (every row is shorter than the chunk): the epilogue runs inside the pass
    auto segments = [&](int rows, const int64_t* d_rowptr, tmf_segments& s) -> int {
        std::vector<int32_t> iota(rows), zero(rows, 0), none(rows, -1);
        std::iota(iota.begin(), iota.end(), 0);
        s = tmf_segments{d_rowptr, dev_copy(iota), dev_copy(zero), dev_copy(none), rows, 1024, 0};
        return s.seg_row && s.seg_chunk && s.seg_slab ? 0 : 1;
    };
    tmf_segments seg_u, seg_i;
    if (segments(m, d_rowptr_u, seg_u) || segments(n, d_rowptr_i, seg_i)) return 2;

    float *d_U = dev_copy(U), *d_V = dev_copy(V), *d_Un, *d_Vn, *d_slab, *d_lp;
    double* d_loss;
    HIP_OK(hipMalloc(&d_Un, U.size() * 4)); HIP_OK(hipMalloc(&d_Vn, V.size() * 4)); HIP_OK(hipMalloc(&d_slab, (size_t)ld * 4));
    HIP_OK(hipMalloc(&d_lp, m * 4)); HIP_OK(hipMalloc(&d_loss, 8));
    HIP_OK(hipMemset(d_Un, 0, U.size() * 4)); HIP_OK(hipMemset(d_Vn, 0, V.size() * 4));
    const tmf_adam adam = tmf_adam_fresh(lr);

    // ---- epochs: GPU through the ABI, fp64 restatement next to it ----
    std::vector<double> Ur(U.begin(), U.end()), Vr(V.begin(), V.end());
    double worst_loss = 0, worst_step = 0;
    for (int ep = 0; ep < epochs; ++ep) {
        TMF_OK_(tmf_mse_pass_f32(&seg_u, d_col_u, d_val_u, d_U, d_V, d_Un, d_slab, d_lp, r, TMF_EPI_ADAM, adam, nullptr));
        TMF_OK_(tmf_mse_pass_f32(&seg_i, d_row_i, d_val_i, d_V, d_U, d_Vn, d_slab, nullptr, r, TMF_EPI_ADAM, adam, nullptr));
        TMF_OK_(tmf_sum_f32(d_lp, m, d_loss, nullptr));
        HIP_OK(hipDeviceSynchronize());
        double loss_gpu;
        HIP_OK(hipMemcpy(&loss_gpu, d_loss, 8, hipMemcpyDeviceToHost));
        // reference (host_ref.h): delta_k = -2 (a_k - p_k), gU[u] += delta_k V[j], gV[j] += delta_k U[u], all from the pre-update tables
        std::vector<double> gU, gV;
        const double loss_ref = host_ref::mse_epoch(P, Ur, Vr, &gU, &gV);
        worst_loss = std::max(worst_loss, fabs(loss_gpu - loss_ref) / loss_ref);
        const auto Un = host_copy(d_Un, U.size()), Vn = host_copy(d_Vn, V.size());
        // the step is a near-sign function of g: compared only where g is well away from 0; the restatement then continues from the
        // GPU's values so that the two trajectories stay on the same tables (host_ref::check_step)
        worst_step = std::max(worst_step, host_ref::check_step(Un, Ur, gU, m, r, ld, ref_adam(adam)));
        worst_step = std::max(worst_step, host_ref::check_step(Vn, Vr, gV, n, r, ld, ref_adam(adam)));
        std::swap(d_U, d_Un);
        std::swap(d_V, d_Vn);
    }
    printf("epochs %d: worst relative loss error %.3g, worst table error after a step %.3g\n", epochs, worst_loss, worst_step);
    if (!(worst_loss < 1e-5) || !(worst_step < 2e-6)) { fprintf(stderr, "FAIL training parity\n"); return 11; }

    // ---- fused predict + top-k over the catalog ----
    int32_t* d_top;
    float* d_topv;
    HIP_OK(hipMalloc(&d_top, (size_t)m * k * 4)); HIP_OK(hipMalloc(&d_topv, (size_t)m * k * 4));
    TMF_OK_(tmf_predict_topk_f32(d_U, d_V, m, n, r, ld, ld, k, 0, d_top, d_topv, nullptr));
    HIP_OK(hipDeviceSynchronize());
    const auto top = host_copy(d_top, (size_t)m * k);
    const auto topv = host_copy(d_topv, (size_t)m * k);
    const int wrong = host_ref::topk_violations(top, topv, Ur, Vr, m, n, r, ld, k);
    printf("top-%d of %d users over %d items: %d violations\n", k, m, n, wrong);
    if (wrong) { fprintf(stderr, "FAIL top-k\n"); return 12; }

    // ---- WMRB epochs through the sliced kernels (loss_graphs.py:74-88), static negative table R ----
    {
        const int S = 16, ns = 3, C = 2, wepochs = 3;
        const float wlr = 1e-3f, cc = (float)n / (float)S;
        const std::vector<int32_t> R = host_ref::draw_negatives(rng, m, n, S);   // any order inside a row: the engine sorts its copy
        int32_t *d_R = dev_copy(R), *d_Rs, *d_off, *d_poff, *d_ent_row, *d_ent_id;
        int64_t* d_rowptr_e;
        const int64_t E = nnz + (int64_t)m * S;
        HIP_OK(hipMalloc(&d_Rs, R.size() * 4)); HIP_OK(hipMalloc(&d_off, (size_t)m * (ns + 1) * 4)); HIP_OK(hipMalloc(&d_poff, (size_t)m * (ns + 1) * 4));
        HIP_OK(hipMalloc(&d_ent_row, E * 4)); HIP_OK(hipMalloc(&d_ent_id, E * 4)); HIP_OK(hipMalloc(&d_rowptr_e, ((size_t)C * n + 2) * 8));
        const size_t wsb = std::max(tmf_sort_samples_workspace_bytes(m, S), tmf_wmrb_entry_lists_workspace_bytes(nnz, m, S));
        void* d_ws2;
        HIP_OK(hipMalloc(&d_ws2, wsb));
        TMF_OK_(tmf_sort_samples(d_R, m, S, n, d_Rs, d_ws2, wsb, nullptr));
        TMF_OK_(tmf_slice_offsets(d_Rs, nullptr, S, m, n, ns, d_off, nullptr));
        TMF_OK_(tmf_slice_offsets(d_col_u, d_rowptr_u, 0, m, n, ns, d_poff, nullptr));
        TMF_OK_(tmf_wmrb_entry_lists(d_user_of, d_col_u, d_val_u, nnz, d_Rs, m, S, n, C, d_ent_row, d_ent_id, d_rowptr_e, d_ws2, wsb, nullptr));
        HIP_OK(hipDeviceSynchronize());
        const auto Rs = host_copy(d_Rs, R.size());
        // lists: one segment per (user block, item) list, every one with a slab slot, slots item-major (tmf_combine_rows sums them)
        std::vector<int32_t> seg_row(C * n), seg_chunk(C * n, 0), seg_slab(C * n), items(n);
        std::vector<int64_t> slab_beg(n + 1);
        for (int c = 0; c < C; ++c) for (int j = 0; j < n; ++j) { seg_row[c * n + j] = c * n + j; seg_slab[c * n + j] = j * C + c; }
        for (int j = 0; j <= n; ++j) slab_beg[j] = (int64_t)j * C;
        std::iota(items.begin(), items.end(), 0);
        tmf_segments seg_e{d_rowptr_e, dev_copy(seg_row), dev_copy(seg_chunk), dev_copy(seg_slab), (int64_t)C * n, 1024, 0};
        int32_t* d_items = dev_copy(items);
        int64_t* d_slab_beg = dev_copy(slab_beg);
        tmf_slice_lists lists{d_Rs, d_off, d_rowptr_u, d_col_u, d_poff, m, S, ns, 0, 0, 0, TMF_SLICE_N_ITEMS_STATED, n};
        float *d_sp, *d_pk, *d_wbuf, *d_part, *d_slab2, *d_lp2;
        HIP_OK(hipMalloc(&d_sp, (size_t)m * S * 4)); HIP_OK(hipMalloc(&d_pk, nnz * 4)); HIP_OK(hipMalloc(&d_wbuf, E * 4));
        HIP_OK(hipMalloc(&d_part, (size_t)ns * m * ld * 4)); HIP_OK(hipMalloc(&d_slab2, (size_t)C * n * ld * 4)); HIP_OK(hipMalloc(&d_lp2, m * 4));
        float *d_delta = d_wbuf, *d_D = d_wbuf + nnz;
        HIP_OK(hipMemcpy(d_U, U.data(), U.size() * 4, hipMemcpyHostToDevice));     // start again from the initial tables
        HIP_OK(hipMemcpy(d_V, V.data(), V.size() * 4, hipMemcpyHostToDevice));
        std::vector<double> Uw(U.begin(), U.end()), Vw(V.begin(), V.end());
        const tmf_adam wadam = tmf_adam_fresh(wlr);
        double wl = 0, wst = 0;
        for (int ep = 0; ep < wepochs; ++ep) {
            TMF_OK_(tmf_wmrb_scores3_f32(&lists, d_U, d_V, d_sp, d_pk, r, nullptr));
            TMF_OK_(tmf_wmrb_hinge2(d_rowptr_u, d_val_u, d_pk, d_sp, m, S, cc, d_delta, d_D, d_lp2, nullptr));
            TMF_OK_(tmf_wmrb_gradu3_f32(&lists, d_D, d_delta, d_V, d_part, 0, r, nullptr));
            TMF_OK_(tmf_wmrb_finish_f32(d_part, ns, m, d_U, d_Un, r, TMF_EPI_ADAM, wadam, nullptr));
            TMF_OK_(tmf_sum_f32(d_lp2, m, d_loss, nullptr));
            TMF_OK_(tmf_wsum_pass_f32(&seg_e, d_ent_row, d_ent_id, d_wbuf, d_U, d_V, d_Vn, d_slab2, r, TMF_EPI_ADAM, wadam, nullptr));
            TMF_OK_(tmf_combine_rows_f32(d_items, d_slab_beg, n, d_slab2, d_V, d_Vn, r, TMF_EPI_ADAM, wadam, nullptr));
            HIP_OK(hipDeviceSynchronize());
            double loss_gpu;
            HIP_OK(hipMemcpy(&loss_gpu, d_loss, 8, hipMemcpyDeviceToHost));
            // reference (host_ref.h; every stored value is > 0 here, so every interaction is a positive)
            std::vector<double> gU, gV;
            const double loss_ref = host_ref::wmrb_epoch(P, Rs, S, cc, Uw, Vw, &gU, &gV);
            wl = std::max(wl, fabs(loss_gpu - loss_ref) / loss_ref);
            const auto Un = host_copy(d_Un, U.size()), Vn = host_copy(d_Vn, V.size());
            wst = std::max(wst, host_ref::check_step(Un, Uw, gU, m, r, ld, ref_adam(wadam)));
            wst = std::max(wst, host_ref::check_step(Vn, Vw, gV, n, r, ld, ref_adam(wadam)));
            std::swap(d_U, d_Un);
            std::swap(d_V, d_Vn);
        }
        printf("WMRB (S=%d, %d slices, %d user blocks) %d epochs: worst relative loss error %.3g, worst table error after a step %.3g\n", S,
               ns, C, wepochs, wl, wst);
        if (!(wl < 1e-5) || !(wst < 2e-6)) { fprintf(stderr, "FAIL WMRB parity\n"); return 14; }
    }

    // ---- error path: a null table is an argument error with a message, not a crash ----
    const int rc = tmf_mse_pass_f32(&seg_u, d_col_u, d_val_u, nullptr, d_V, d_Un, d_slab, d_lp, r, TMF_EPI_ADAM, adam, nullptr);
    if (rc == TMF_OK || !tmf_last_error()[0]) { fprintf(stderr, "FAIL error reporting\n"); return 13; }
    printf("PASS (argument error reported as %d: %s)\n", rc, tmf_last_error());
    return 0;
}
