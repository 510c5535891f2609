// Index preparation done once per fit(): CSR-by-user view of the COO interactions and the stable
// CSC (or any key) ordering of its entries.  The reference has no counterpart - it gathers from the dense
// [m, n] score matrix with the COO indices (loss_graphs.py:47-50) - this is what makes the sparse passes
// stream.  Sorting is rocPRIM's stable radix sort; everything else is hand-written.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "tmf_common.h"

namespace tmf {

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

// key = user * n_items + item: one stable sort gives the row-major order tf.sparse.SparseTensor is specified in
__global__ __launch_bounds__(256) void k_pair_keys(const int64_t* __restrict__ indices, int64_t nnz, int64_t n_items,
                                                   int64_t* __restrict__ keys, int64_t* __restrict__ iota) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        keys[k] = indices[2 * k] * n_items + indices[2 * k + 1];
        iota[k] = k;
    }
}

__global__ __launch_bounds__(256) void k_decode_csr(const int64_t* __restrict__ sorted_keys, const float* __restrict__ values,
                                                    const int64_t* __restrict__ perm, int64_t nnz, int64_t n_items,
                                                    int32_t* __restrict__ user_of, int32_t* __restrict__ col,
                                                    float* __restrict__ val) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t key = sorted_keys[k];
        user_of[k] = (int32_t)(key / n_items);
        col[k] = (int32_t)(key % n_items);
        val[k] = values[perm[k]];
    }
}

// off[row][b] = first position p of the row's ascending ids with ids[p] >= b * width (b = 0 .. n_slices; the last one is
// the row length).  Rows are [row * stride, +len) with a fixed length (rowptr == nullptr) or [rowptr[row], rowptr[row + 1]).
__global__ __launch_bounds__(256) void k_slice_offsets(const int32_t* __restrict__ ids, const int64_t* __restrict__ rowptr,
                                                       int64_t stride, int64_t n_rows, int32_t width, int32_t n_slices,
                                                       int32_t* __restrict__ off) {
    const int64_t total = n_rows * (n_slices + 1);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / (n_slices + 1);
        const int b = (int)(t % (n_slices + 1));
        const int64_t beg = rowptr ? rowptr[row] : row * stride;
        const int64_t len = rowptr ? rowptr[row + 1] - beg : stride;
        int64_t lo = 0, hi = len;
        if (b == n_slices) lo = len;
        else {
            const int64_t bound = (int64_t)b * width;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if ((int64_t)ids[beg + mid] < bound) lo = mid + 1;
                else hi = mid;
            }
        }
        off[t] = (int32_t)lo;
    }
}

// Keys of the WMRB item-side entry lists.  Entry id i < nnz is interaction i of the CSR (list row = user block * n_items
// + item for a positive, the dummy row n_rows for a stored value <= 0: those take part in nothing), id nnz + u * S + s
// is negative s of user u.
__global__ __launch_bounds__(256) void k_entry_keys(const int32_t* __restrict__ user_of, const int32_t* __restrict__ col,
                                                    const float* __restrict__ val, int64_t nnz,
                                                    const int32_t* __restrict__ R, int64_t n_users, int64_t S, int64_t n_items,
                                                    int64_t users_per_block, int32_t n_rows, int32_t* __restrict__ keys,
                                                    int32_t* __restrict__ ids) {
    const int64_t total = nnz + n_users * S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t key;
        if (i < nnz) key = (val[i] > 0.f) ? (user_of[i] / users_per_block) * n_items + col[i] : n_rows;
        else key = ((i - nnz) / S / users_per_block) * n_items + R[i - nnz];
        keys[i] = (int32_t)key;
        ids[i] = (int32_t)i;
    }
}

// After the sort: the user of every list entry.
__global__ __launch_bounds__(256) void k_entry_rows(const int32_t* __restrict__ order, const int32_t* __restrict__ user_of,
                                                    int64_t nnz, int64_t S, int64_t total, int32_t* __restrict__ ent_row) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t id = order[e];
        ent_row[e] = (id < nnz) ? user_of[id] : (int32_t)((id - nnz) / S);
    }
}

struct RowStart {
    unsigned int stride;
    __host__ __device__ unsigned int operator()(unsigned int row) const { return row * stride; }
};

static unsigned grid_for(int64_t n) {
    const int64_t want = (n + 255) / 256;
    return (unsigned)(want < 4096 ? (want > 0 ? want : 1) : 4096);
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t sort_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (const int64_t*)nullptr,
                              (int64_t*)nullptr, (size_t)n, 0, 32, (hipStream_t)0);
    return bytes;
}

static size_t sort_temp_bytes_k64(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int64_t*)nullptr, (int64_t*)nullptr, (const int64_t*)nullptr,
                              (int64_t*)nullptr, (size_t)n, 0, 64, (hipStream_t)0);
    return bytes;
}

static size_t sort_temp_bytes_v32(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (const int32_t*)nullptr,
                              (int32_t*)nullptr, (size_t)n, 0, 32, (hipStream_t)0);
    return bytes;
}

static int bits_for(int64_t n_values) {  // bits needed for keys in [0, n_values)
    int bits = 1;
    while (bits < 63 && ((int64_t)1 << bits) < (n_values > 1 ? n_values : 2)) ++bits;
    return bits;
}

// workspace layout: [keys_in n*4][keys_out n*4][iota n*8][rocprim temp]
static size_t stable_order_workspace(int64_t n) {
    return align256((size_t)n * 4) * 2 + align256((size_t)n * 8) + align256(sort_temp_bytes(n));
}

}  // namespace tmf

using namespace tmf;

// workspace of tmf_csr_build: [keys_in n*8][keys_out n*8][iota n*8][perm n*8][rocprim temp]
extern "C" size_t tmf_csr_build_workspace_bytes(int64_t nnz) {
    const int64_t n = nnz > 0 ? nnz : 1;
    return 4 * align256((size_t)n * 8) + align256(sort_temp_bytes_k64(n));
}

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
    const int bits = bits_for(n_rows);  // keys are in [0, n_rows)
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_out, iota, perm, (size_t)n, 0, bits, s);
    if (e != hipSuccess) { set_error("radix_sort_pairs: %s", hipGetErrorString(e)); return TMF_E_LAUNCH; }
    hipLaunchKernelGGL(k_rowptr, dim3(grid_for(n_rows + 1)), dim3(256), 0, s, (const int32_t*)keys_out, n, n_rows, rowptr);
    return check_launch("tmf_stable_order_i32");
}

extern "C" int tmf_csr_build(const int64_t* indices, const float* values, int64_t nnz, int32_t n_users, int32_t n_items,
                             int64_t* rowptr_u, int32_t* col_u, float* val_u, int32_t* user_of, void* workspace,
                             size_t workspace_bytes, void* stream) {
    TMF_REQUIRE(nnz >= 0 && n_users >= 0 && n_items >= 0 && rowptr_u &&
                    (nnz == 0 || (indices && values && col_u && val_u && user_of)), "csr_build: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (nnz == 0) {
        hipLaunchKernelGGL(k_rowptr, dim3(grid_for((int64_t)n_users + 1)), dim3(256), 0, s, (const int32_t*)nullptr, (int64_t)0,
                           (int64_t)n_users, rowptr_u);
        return check_launch("tmf_csr_build");
    }
    TMF_REQUIRE(nnz < ((int64_t)1 << 31), "csr_build: more than 2^31 interactions");
    TMF_REQUIRE(workspace && workspace_bytes >= tmf_csr_build_workspace_bytes(nnz), "csr_build: workspace too small");
    char* w = static_cast<char*>(workspace);
    const size_t a8 = align256((size_t)nnz * 8);
    int64_t* keys_in = reinterpret_cast<int64_t*>(w);
    int64_t* keys_out = reinterpret_cast<int64_t*>(w + a8);
    int64_t* iota = reinterpret_cast<int64_t*>(w + 2 * a8);
    int64_t* perm = reinterpret_cast<int64_t*>(w + 3 * a8);
    void* temp = w + 4 * a8;
    size_t temp_bytes = sort_temp_bytes_k64(nnz);
    hipLaunchKernelGGL(k_pair_keys, dim3(grid_for(nnz)), dim3(256), 0, s, indices, nnz, (int64_t)n_items, keys_in, iota);
    const int bits = bits_for((int64_t)n_users * (int64_t)n_items);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, iota, perm, (size_t)nnz, 0, bits, s);
    if (e != hipSuccess) { set_error("radix_sort_pairs: %s", hipGetErrorString(e)); return TMF_E_LAUNCH; }
    hipLaunchKernelGGL(k_decode_csr, dim3(grid_for(nnz)), dim3(256), 0, s, (const int64_t*)keys_out, values, (const int64_t*)perm,
                       nnz, (int64_t)n_items, user_of, col_u, val_u);
    hipLaunchKernelGGL(k_rowptr, dim3(grid_for((int64_t)n_users + 1)), dim3(256), 0, s, (const int32_t*)user_of, nnz,
                       (int64_t)n_users, rowptr_u);
    return check_launch("tmf_csr_build");
}

extern "C" int tmf_csc_perm(const int32_t* col_u, int64_t nnz, int32_t n_items, int64_t* rowptr_i, int64_t* perm,
                            void* workspace, size_t workspace_bytes, void* stream) {
    return tmf_stable_order_i32(col_u, nnz, n_items, perm, nullptr, rowptr_i, workspace, workspace_bytes, stream);
}


// ---- index structures of the sliced WMRB pass ----
extern "C" size_t tmf_sort_samples_workspace_bytes(int32_t n_users, int32_t S) {
    if (n_users <= 0 || S <= 0) return 256;
    size_t bytes = 0;
    auto offs = rocprim::make_transform_iterator(rocprim::make_counting_iterator(0u), RowStart{(unsigned)S});
    (void)rocprim::segmented_radix_sort_keys(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr,
                                       (unsigned)((int64_t)n_users * S), (unsigned)n_users, offs, offs + 1, 0, 32, (hipStream_t)0);
    return align256(bytes) + 256;
}

extern "C" int tmf_sort_samples(const int32_t* R, int32_t n_users, int32_t S, int32_t n_items, int32_t* R_sorted,
                                void* workspace, size_t workspace_bytes, void* stream) {
    if (n_users == 0) return TMF_OK;
    TMF_REQUIRE(R && R_sorted && n_users > 0 && S > 0 && n_items > 0, "sort_samples: bad arguments");
    TMF_REQUIRE((int64_t)n_users * S < ((int64_t)1 << 32), "sort_samples: n_users * n_samples >= 2^32");
    TMF_REQUIRE(workspace && workspace_bytes >= tmf_sort_samples_workspace_bytes(n_users, S), "sort_samples: workspace too small");
    size_t bytes = workspace_bytes;
    auto offs = rocprim::make_transform_iterator(rocprim::make_counting_iterator(0u), RowStart{(unsigned)S});
    hipError_t e = rocprim::segmented_radix_sort_keys(workspace, bytes, R, R_sorted, (unsigned)((int64_t)n_users * S),
                                                      (unsigned)n_users, offs, offs + 1, 0, (unsigned)bits_for(n_items),
                                                      (hipStream_t)stream);
    if (e != hipSuccess) { set_error("segmented_radix_sort_keys: %s", hipGetErrorString(e)); return TMF_E_LAUNCH; }
    return check_launch("tmf_sort_samples");
}

extern "C" int tmf_slice_offsets(const int32_t* ids, const int64_t* rowptr, int64_t stride, int32_t n_rows, int32_t n_items,
                                 int32_t n_slices, int32_t* off, void* stream) {
    if (n_rows == 0) return TMF_OK;
    TMF_REQUIRE(off && n_rows > 0 && n_items > 0 && n_slices > 0 && (rowptr || stride >= 0), "slice_offsets: bad arguments");
    const int32_t width = (int32_t)(((int64_t)n_items + n_slices - 1) / n_slices);
    hipLaunchKernelGGL(k_slice_offsets, dim3(grid_for((int64_t)n_rows * (n_slices + 1))), dim3(256), 0, (hipStream_t)stream, ids,
                       rowptr, stride, (int64_t)n_rows, width, n_slices, off);
    return check_launch("tmf_slice_offsets");
}

// workspace: [keys_in E*4][keys_out E*4][ids E*4][rocprim temp],  E = nnz + n_users * S (the sorted ids land in ent_id)
extern "C" size_t tmf_wmrb_entry_lists_workspace_bytes(int64_t nnz, int32_t n_users, int32_t S) {
    const int64_t E = nnz + (int64_t)n_users * S;
    const int64_t n = E > 0 ? E : 1;
    return 3 * align256((size_t)n * 4) + align256(sort_temp_bytes_v32(n));
}

extern "C" int tmf_wmrb_entry_lists(const int32_t* user_of, const int32_t* col_u, const float* val_u, int64_t nnz,
                                    const int32_t* R_sorted, int32_t n_users, int32_t S, int32_t n_items, int32_t user_chunks,
                                    int32_t* ent_row, int32_t* ent_id, int64_t* rowptr_e, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    const int64_t E = nnz + (int64_t)n_users * S;
    TMF_REQUIRE(nnz >= 0 && n_users >= 0 && S >= 0 && n_items > 0 && user_chunks > 0 && rowptr_e, "entry_lists: bad arguments");
    const int64_t n_rows = (int64_t)user_chunks * n_items;  // + 1 dummy row for the non-positive interactions
    TMF_REQUIRE(n_rows + 1 < ((int64_t)1 << 31), "entry_lists: user_chunks * n_items >= 2^31");
    TMF_REQUIRE(E < ((int64_t)1 << 31), "entry_lists: interactions + n_users * n_samples = %lld >= 2^31 per GPU: split the users "
                "over more GPUs", (long long)E);
    hipStream_t s = (hipStream_t)stream;
    if (E == 0) {
        hipLaunchKernelGGL(k_rowptr, dim3(grid_for(n_rows + 2)), dim3(256), 0, s, (const int32_t*)nullptr, (int64_t)0, n_rows + 1,
                           rowptr_e);
        return check_launch("tmf_wmrb_entry_lists");
    }
    TMF_REQUIRE(ent_row && ent_id && (nnz == 0 || (user_of && col_u && val_u)) && ((int64_t)n_users * S == 0 || R_sorted),
                "entry_lists: null pointer");
    TMF_REQUIRE(workspace && workspace_bytes >= tmf_wmrb_entry_lists_workspace_bytes(nnz, n_users, S), "entry_lists: workspace too small");
    char* w = static_cast<char*>(workspace);
    const size_t a4 = align256((size_t)E * 4);
    int32_t* keys_in = reinterpret_cast<int32_t*>(w);
    int32_t* keys_out = reinterpret_cast<int32_t*>(w + a4);
    int32_t* ids = reinterpret_cast<int32_t*>(w + 2 * a4);
    int32_t* order = ent_id;
    void* temp = w + 3 * a4;
    size_t temp_bytes = sort_temp_bytes_v32(E);
    const int64_t upb = ((int64_t)n_users + user_chunks - 1) / user_chunks;
    hipLaunchKernelGGL(k_entry_keys, dim3(grid_for(E)), dim3(256), 0, s, user_of, col_u, val_u, nnz, R_sorted, (int64_t)n_users,
                       (int64_t)S, (int64_t)n_items, upb > 0 ? upb : 1, (int32_t)n_rows, keys_in, ids);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, ids, order, (size_t)E, 0, bits_for(n_rows + 1), s);
    if (e != hipSuccess) { set_error("radix_sort_pairs: %s", hipGetErrorString(e)); return TMF_E_LAUNCH; }
    hipLaunchKernelGGL(k_rowptr, dim3(grid_for(n_rows + 2)), dim3(256), 0, s, (const int32_t*)keys_out, E, n_rows + 1, rowptr_e);
    hipLaunchKernelGGL(k_entry_rows, dim3(grid_for(E)), dim3(256), 0, s, (const int32_t*)order, user_of, nnz, (int64_t)S, E, ent_row);
    return check_launch("tmf_wmrb_entry_lists");
}
