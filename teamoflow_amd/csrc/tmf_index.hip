// Index preparation done once per fit(): CSR-by-user view of the COO interactions and the stable
// CSC (or any key) ordering of its entries.  The reference has no counterpart - it gathers from the dense
// [m, n] score matrix with the COO indices (loss_graphs.py:47-50) - this is what makes the sparse passes
// stream.  Sorting is rocPRIM's stable radix sort; everything else is hand-written.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "tmf_common.h"

namespace tmf {

__global__ __launch_bounds__(256) void k_extract_keys(const int64_t* __restrict__ indices, int64_t nnz, int column,
                                                      int32_t* __restrict__ keys, int64_t* __restrict__ iota) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        keys[k] = (int32_t)indices[2 * k + column];
        iota[k] = k;
    }
}

__global__ __launch_bounds__(256) void k_iota(int64_t n, int64_t* __restrict__ iota) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) iota[k] = k;
}

// rowptr[r] = first position whose sorted key is >= r  (r = 0 .. n_rows; rowptr[n_rows] = n)
__global__ __launch_bounds__(256) void k_rowptr(const int32_t* __restrict__ sorted_keys, int64_t n, int64_t n_rows,
                                                int64_t* __restrict__ rowptr) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_rows; r += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if ((int64_t)sorted_keys[mid] < r) lo = mid + 1;
            else hi = mid;
        }
        rowptr[r] = lo;
    }
}

__global__ __launch_bounds__(256) void k_gather_csr(const int64_t* __restrict__ indices, const float* __restrict__ values,
                                                    const int64_t* __restrict__ perm, int64_t nnz,
                                                    int32_t* __restrict__ col, float* __restrict__ val) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t src = perm[k];
        col[k] = (int32_t)indices[2 * src + 1];
        val[k] = values[src];
    }
}

static unsigned grid_for(int64_t n) {
    const int64_t want = (n + 255) / 256;
    return (unsigned)(want < 4096 ? (want > 0 ? want : 1) : 4096);
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t sort_temp_bytes(int64_t n) {
    size_t bytes = 0;
    rocprim::radix_sort_pairs(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (const int64_t*)nullptr,
                              (int64_t*)nullptr, (size_t)n, 0, 32, (hipStream_t)0);
    return bytes;
}

// workspace layout: [keys_in n*4][keys_out n*4][iota n*8][rocprim temp]
static size_t stable_order_workspace(int64_t n) {
    return align256((size_t)n * 4) * 2 + align256((size_t)n * 8) + align256(sort_temp_bytes(n));
}

}  // namespace tmf

using namespace tmf;

extern "C" size_t tmf_csr_build_workspace_bytes(int64_t nnz) { return stable_order_workspace(nnz > 0 ? nnz : 1) + align256((size_t)(nnz > 0 ? nnz : 1) * 8); }

extern "C" size_t tmf_stable_order_workspace_bytes(int64_t n) { return stable_order_workspace(n > 0 ? n : 1); }

extern "C" int tmf_stable_order_i32(const int32_t* keys, int64_t n, int64_t n_rows, int64_t* perm, int32_t* sorted_keys,
                                    int64_t* rowptr, void* workspace, size_t workspace_bytes, void* stream) {
    TMF_REQUIRE(n >= 0 && n_rows >= 0 && rowptr && (n == 0 || (keys && perm)), "stable_order: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        hipLaunchKernelGGL(k_rowptr, dim3(grid_for(n_rows + 1)), dim3(256), 0, s, (const int32_t*)nullptr, (int64_t)0, n_rows, rowptr);
        return check_launch("tmf_stable_order_i32");
    }
    TMF_REQUIRE(n < ((int64_t)1 << 31), "stable_order: more than 2^31 keys");
    TMF_REQUIRE(workspace && workspace_bytes >= stable_order_workspace(n), "stable_order: workspace too small (%zu < %zu)",
                workspace_bytes, stable_order_workspace(n));
    char* w = static_cast<char*>(workspace);
    int32_t* keys_out = sorted_keys ? sorted_keys : reinterpret_cast<int32_t*>(w + align256((size_t)n * 4));
    int64_t* iota = reinterpret_cast<int64_t*>(w + 2 * align256((size_t)n * 4));
    void* temp = w + 2 * align256((size_t)n * 4) + align256((size_t)n * 8);
    size_t temp_bytes = sort_temp_bytes(n);
    hipLaunchKernelGGL(k_iota, dim3(grid_for(n)), dim3(256), 0, s, n, iota);
    int bits = 1;
    while (bits < 32 && ((int64_t)1 << bits) < (n_rows > 1 ? n_rows : 2)) ++bits;  // keys are in [0, n_rows)
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_out, iota, perm, (size_t)n, 0, bits, s);
    if (e != hipSuccess) { set_error("radix_sort_pairs: %s", hipGetErrorString(e)); return TMF_E_LAUNCH; }
    hipLaunchKernelGGL(k_rowptr, dim3(grid_for(n_rows + 1)), dim3(256), 0, s, (const int32_t*)keys_out, n, n_rows, rowptr);
    return check_launch("tmf_stable_order_i32");
}

extern "C" int tmf_csr_build(const int64_t* indices, const float* values, int64_t nnz, int32_t n_users,
                             int64_t* rowptr_u, int32_t* col_u, float* val_u, int32_t* user_of, void* workspace,
                             size_t workspace_bytes, void* stream) {
    TMF_REQUIRE(nnz >= 0 && n_users >= 0 && rowptr_u && (nnz == 0 || (indices && values && col_u && val_u && user_of)),
                "csr_build: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (nnz == 0) {
        hipLaunchKernelGGL(k_rowptr, dim3(grid_for((int64_t)n_users + 1)), dim3(256), 0, s, (const int32_t*)nullptr, (int64_t)0,
                           (int64_t)n_users, rowptr_u);
        return check_launch("tmf_csr_build");
    }
    TMF_REQUIRE(workspace && workspace_bytes >= tmf_csr_build_workspace_bytes(nnz), "csr_build: workspace too small");
    char* w = static_cast<char*>(workspace);
    int32_t* keys_in = reinterpret_cast<int32_t*>(w);
    int64_t* iota = reinterpret_cast<int64_t*>(w + 2 * align256((size_t)nnz * 4));
    int64_t* perm = reinterpret_cast<int64_t*>(w + stable_order_workspace(nnz));
    hipLaunchKernelGGL(k_extract_keys, dim3(grid_for(nnz)), dim3(256), 0, s, indices, nnz, 0, keys_in, iota);
    // sorted user ids land in user_of (= the CSR row of every entry); perm = stable order by user
    if (int rc = tmf_stable_order_i32(keys_in, nnz, n_users, perm, user_of, rowptr_u, workspace, stable_order_workspace(nnz), stream))
        return rc;
    hipLaunchKernelGGL(k_gather_csr, dim3(grid_for(nnz)), dim3(256), 0, s, indices, values, (const int64_t*)perm, nnz, col_u, val_u);
    return check_launch("tmf_csr_build");
}

extern "C" int tmf_csc_perm(const int32_t* col_u, int64_t nnz, int32_t n_items, int64_t* rowptr_i, int64_t* perm,
                            void* workspace, size_t workspace_bytes, void* stream) {
    return tmf_stable_order_i32(col_u, nnz, n_items, perm, nullptr, rowptr_i, workspace, workspace_bytes, stream);
}
