// K7 predict GEMM on the exact-fp32 MFMA, K8 tie-stable row top-k, and gather_matrix_indices.
#include <math.h>

#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "tmf_common.h"

namespace tmf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// C[m, n] = A[m, :K] . B[n, :K]^T   (both operands K-contiguous: U and V as stored).
// 256 threads = 4 waves in a 2x2 grid; each wave owns a 64x64 block as 2x2 v_mfma_f32_32x32x2_f32
// tiles (64 accumulator registers).  A and B panels are staged k-major in LDS ([k][row], row
// stride BM+1) so the MFMA operand read (lane l -> row l&31, k l>>5) is conflict-free.
// ---------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 16, LDT = BM + 1;

__global__ __launch_bounds__(256) void k_predict_gemm(const float* __restrict__ A, const float* __restrict__ B,
                                                      float* __restrict__ C, int64_t m, int64_t n, int K,
                                                      int64_t lda, int64_t ldb, int64_t ldc, int tiles_n) {
    __shared__ float As[BK * LDT];
    __shared__ float Bs[BK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int64_t row0 = tile_m * BM, col0 = tile_n * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // staging map: thread -> (row = tid / 4 + 64 * h, k-quad = tid % 4), float4 along k
    const int srow = tid >> 2, skq = tid & 3;
    for (int k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = srow + 64 * h;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            const int kk = k0 + 4 * skq;
            if (row0 + r < m) {
                const float* p = A + (row0 + r) * lda + kk;
                if (kk + 3 < K) a = *reinterpret_cast<const float4*>(p);
                else { if (kk < K) a.x = p[0]; if (kk + 1 < K) a.y = p[1]; if (kk + 2 < K) a.z = p[2]; }
            }
            if (col0 + r < n) {
                const float* p = B + (col0 + r) * ldb + kk;
                if (kk + 3 < K) b = *reinterpret_cast<const float4*>(p);
                else { if (kk < K) b.x = p[0]; if (kk + 1 < K) b.y = p[1]; if (kk + 2 < K) b.z = p[2]; }
            }
            As[(4 * skq + 0) * LDT + r] = a.x; As[(4 * skq + 1) * LDT + r] = a.y;
            As[(4 * skq + 2) * LDT + r] = a.z; As[(4 * skq + 3) * LDT + r] = a.w;
            Bs[(4 * skq + 0) * LDT + r] = b.x; Bs[(4 * skq + 1) * LDT + r] = b.y;
            Bs[(4 * skq + 2) * LDT + r] = b.z; Bs[(4 * skq + 3) * LDT + r] = b.w;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < BK; ks += 2) {
            const int kl = ks + (lane >> 5), rl = lane & 31;
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[kl * LDT + wr * 64 + i * 32 + rl];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Bs[kl * LDT + wc * 64 + j * 32 + rl];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D map of the 32x32 tile: col = lane & 31, row = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t c = col0 + wc * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int64_t r = row0 + wr * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                if (r < m && c < n) C[r * ldc + c] = acc[i][j][q];
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Row top-k, ordered (value desc, index asc).  One wave per row.  Each lane keeps its own k best
// (sorted, in a lane-private LDS column) over the columns it scans in increasing index order - a
// later equal value never displaces an earlier one - then the wave extracts the global k best by
// k rounds of arg-max over the lanes' heads under the same total order.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool before(float va, int ia, float vb, int ib) {
    return (va > vb) || (va == vb && ia < ib);
}

template <bool VEC4>
__global__ __launch_bounds__(64) void k_topk_small(const float* __restrict__ X, int64_t cols, int64_t ldx, int k,
                                                   int clamp, int32_t* __restrict__ out_idx,
                                                   float* __restrict__ out_val) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* lv = reinterpret_cast<float*>(smem_raw);        // [k][64]
    int* li = reinterpret_cast<int*>(lv + (size_t)k * 64); // [k][64]
    const int lane = threadIdx.x;
    const int64_t row = blockIdx.x;
    const float* x = X + row * ldx;
    for (int j = 0; j < k; ++j) {
        lv[j * 64 + lane] = -INFINITY;
        li[j * 64 + lane] = 0x7fffffff;
    }
    float thr = -INFINITY;  // lane's current k-th best
    auto offer = [&](float v, int idx) {
        if (clamp) v = (v > 0.f) ? v : 0.f;
        if (v > thr) {
            int j = k - 1;
            while (j > 0 && lv[(j - 1) * 64 + lane] < v) {
                lv[j * 64 + lane] = lv[(j - 1) * 64 + lane];
                li[j * 64 + lane] = li[(j - 1) * 64 + lane];
                --j;
            }
            lv[j * 64 + lane] = v;
            li[j * 64 + lane] = idx;
            thr = lv[(k - 1) * 64 + lane];
        }
    };
    if (VEC4) {
        const int64_t n4 = cols / 4;
        for (int64_t t = lane; t < n4; t += 64) {
            const float4 q = reinterpret_cast<const float4*>(x)[t];
            const int b = (int)(4 * t);
            offer(q.x, b); offer(q.y, b + 1); offer(q.z, b + 2); offer(q.w, b + 3);
        }
        // the tail (cols % 4) comes after every vector element in index order
        for (int64_t t = 4 * n4 + lane; t < cols; t += 64) offer(x[t], (int)t);
    } else {
        for (int64_t t = lane; t < cols; t += 64) offer(x[t], (int)t);
    }
    // extraction
    int head = 0;
    float hv = lv[lane];
    int hi = li[lane];
    for (int r = 0; r < k; ++r) {
        float bv = hv;
        int bi = hi;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (before(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) {
            out_idx[row * k + r] = bi;
            if (out_val) out_val[row * k + r] = bv;
        }
        if (hi == bi && head < k) {  // the winner pops its head (indices are unique)
            ++head;
            if (head < k) { hv = lv[head * 64 + lane]; hi = li[head * 64 + lane]; }
            else { hv = -INFINITY; hi = 0x7fffffff; }
        }
    }
}

// Full ranking (k > 64): one block per row, bitonic sort of (value, index) in LDS under the same
// total order.  cols padded to a power of two with (-inf, INT_MAX).
__global__ __launch_bounds__(1024) void k_sort_rows(const float* __restrict__ X, int64_t cols, int64_t ldx, int k,
                                                    int clamp, int npow2, int32_t* __restrict__ out_idx,
                                                    float* __restrict__ out_val) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sv = reinterpret_cast<float*>(smem_raw);
    int* si = reinterpret_cast<int*>(sv + npow2);
    const int64_t row = blockIdx.x;
    const float* x = X + row * ldx;
    for (int t = threadIdx.x; t < npow2; t += blockDim.x) {
        float v = -INFINITY;
        int i = 0x7fffffff;
        if (t < cols) {
            v = x[t];
            if (clamp) v = (v > 0.f) ? v : 0.f;
            i = t;
        }
        sv[t] = v;
        si[t] = i;
    }
    __syncthreads();
    for (int size = 2; size <= npow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < npow2 / 2; t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);  // "up" blocks end sorted in final order
                const float va = sv[lo], vb = sv[hi];
                const int ia = si[lo], ib = si[hi];
                const bool wrong = up ? before(vb, ib, va, ia) : before(va, ia, vb, ib);
                if (wrong) { sv[lo] = vb; sv[hi] = va; si[lo] = ib; si[hi] = ia; }
            }
            __syncthreads();
        }
    }
    for (int t = threadIdx.x; t < k; t += blockDim.x) {
        out_idx[row * k + t] = si[t];
        if (out_val) out_val[row * k + t] = sv[t];
    }
}

// Rows wider than the LDS sort holds (k > 64 and cols > 16384: full rankings and large k on real catalogs): keys and
// column ids go through rocPRIM's segmented radix sort in DESCENDING key order.  Radix sort is stable, so equal values
// keep their ascending column order - tf.math.top_k's rule; -0.0 is stored as +0.0 so that the two compare equal.
__global__ __launch_bounds__(256) void k_rank_prepare(const float* __restrict__ X, int64_t rows, int64_t cols, int64_t ldx,
                                                      int clamp, float* __restrict__ keys, int32_t* __restrict__ ids) {
    const int64_t total = rows * cols;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / cols, c = t - r * cols;
        float v = X[r * ldx + c];
        if (clamp) v = (v > 0.f) ? v : 0.f;
        keys[t] = (v == 0.f) ? 0.f : v;
        ids[t] = (int32_t)c;
    }
}

__global__ __launch_bounds__(256) void k_rank_emit(const float* __restrict__ keys, const int32_t* __restrict__ ids, int64_t rows,
                                                   int64_t cols, int k, int32_t* __restrict__ out_idx,
                                                   float* __restrict__ out_val) {
    const int64_t total = rows * k;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / k, j = t - r * k;
        out_idx[t] = ids[r * cols + j];
        if (out_val) out_val[t] = keys[r * cols + j];
    }
}

struct RankRowStart {
    unsigned int stride;
    __host__ __device__ unsigned int operator()(unsigned int row) const { return row * stride; }
};

static size_t rank_align(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t rank_sort_temp(int64_t rows, int64_t cols) {
    size_t bytes = 0;
    auto offs = rocprim::make_transform_iterator(rocprim::make_counting_iterator(0u), RankRowStart{(unsigned)cols});
    (void)rocprim::segmented_radix_sort_pairs_desc(nullptr, bytes, (const float*)nullptr, (float*)nullptr, (const int32_t*)nullptr,
                                                   (int32_t*)nullptr, (unsigned)(rows * cols), (unsigned)rows, offs, offs + 1, 0, 32,
                                                   (hipStream_t)0);
    return bytes;
}

static bool rank_needs_global_sort(int64_t cols, int k) {
    int64_t npow2 = 1;
    while (npow2 < cols) npow2 <<= 1;
    return k > 64 && npow2 * 8 > 160 * 1024;
}

__global__ __launch_bounds__(256) void k_gather_rows_cols(const float* __restrict__ X, const int64_t* __restrict__ idx,
                                                          float* __restrict__ out, int64_t rows, int64_t cols, int64_t k) {
    const int64_t total = rows * k;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / k;
        out[t] = X[r * cols + idx[t]];
    }
}

}  // namespace tmf

using namespace tmf;

extern "C" int tmf_predict_gemm_f32(const float* A, const float* B, float* C, int64_t m, int64_t n, int r,
                                    int64_t lda, int64_t ldb, int64_t ldc, void* stream) {
    if (m == 0 || n == 0) return TMF_OK;
    TMF_REQUIRE(A && B && C && m > 0 && n > 0 && r > 0, "predict_gemm: bad arguments");
    TMF_REQUIRE(lda >= r && ldb >= r && ldc >= n, "predict_gemm: leading dimension too small");
    TMF_REQUIRE((lda % 4 == 0) && (ldb % 4 == 0) && ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0),
                "predict_gemm: operands must be 16-byte aligned with ld %% 4 == 0");
    const int64_t tm = (m + BM - 1) / BM, tn = (n + BN - 1) / BN;
    TMF_REQUIRE(tm * tn < ((int64_t)1 << 31), "predict_gemm: too many tiles, chunk the users");
    hipLaunchKernelGGL(k_predict_gemm, dim3((unsigned)(tm * tn)), dim3(256), 0, (hipStream_t)stream, A, B, C, m, n, r,
                       lda, ldb, ldc, (int)tn);
    return check_launch("tmf_predict_gemm_f32");
}

extern "C" size_t tmf_topk_workspace_bytes(int64_t rows, int64_t cols, int k) {
    if (rows <= 0 || cols <= 0 || !rank_needs_global_sort(cols, k) || rows * cols >= ((int64_t)1 << 32)) return 0;
    return 4 * rank_align((size_t)(rows * cols) * 4) + rank_align(rank_sort_temp(rows, cols));
}

extern "C" int tmf_topk_stable_f32(const float* X, int64_t rows, int64_t cols, int64_t ldx, int k,
                                   int clamp_negatives, int32_t* out_idx, float* out_val, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    if (rows == 0) return TMF_OK;
    TMF_REQUIRE(X && out_idx && rows > 0 && cols > 0 && ldx >= cols, "topk: bad arguments");
    TMF_REQUIRE(k >= 1 && k <= cols, "topk: k=%d must be in [1, cols=%lld]", k, (long long)cols);
    TMF_REQUIRE(cols < ((int64_t)1 << 31) && rows < ((int64_t)1 << 31), "topk: too large");
    if (k <= 64) {
        const size_t lds = (size_t)k * 64 * 8;
        const bool vec = (ldx % 4 == 0) && ((uintptr_t)X % 16 == 0);
        if (vec)
            hipLaunchKernelGGL((k_topk_small<true>), dim3((unsigned)rows), dim3(64), lds, (hipStream_t)stream, X, cols,
                               ldx, k, clamp_negatives, out_idx, out_val);
        else
            hipLaunchKernelGGL((k_topk_small<false>), dim3((unsigned)rows), dim3(64), lds, (hipStream_t)stream, X, cols,
                               ldx, k, clamp_negatives, out_idx, out_val);
        return check_launch("tmf_topk_stable_f32");
    }
    if (rank_needs_global_sort(cols, k)) {
        TMF_REQUIRE(rows * cols < ((int64_t)1 << 32), "topk: rows * cols >= 2^32 in one call: pass fewer rows");
        const size_t need = tmf_topk_workspace_bytes(rows, cols, k);
        TMF_REQUIRE(workspace && workspace_bytes >= need, "topk: k=%d over %lld columns needs a workspace of tmf_topk_workspace_bytes() "
                    "= %zu bytes", k, (long long)cols, need);
        char* w = static_cast<char*>(workspace);
        const size_t a4 = rank_align((size_t)(rows * cols) * 4);
        float* keys_in = reinterpret_cast<float*>(w);
        float* keys_out = reinterpret_cast<float*>(w + a4);
        int32_t* ids_in = reinterpret_cast<int32_t*>(w + 2 * a4);
        int32_t* ids_out = reinterpret_cast<int32_t*>(w + 3 * a4);
        size_t temp_bytes = workspace_bytes - 4 * a4;
        const int64_t want = (rows * cols + 255) / 256;
        const unsigned grid = (unsigned)(want < 8192 ? want : 8192);
        hipStream_t s = (hipStream_t)stream;
        hipLaunchKernelGGL(k_rank_prepare, dim3(grid), dim3(256), 0, s, X, rows, cols, ldx, clamp_negatives, keys_in, ids_in);
        auto offs = rocprim::make_transform_iterator(rocprim::make_counting_iterator(0u), RankRowStart{(unsigned)cols});
        hipError_t e = rocprim::segmented_radix_sort_pairs_desc(w + 4 * a4, temp_bytes, keys_in, keys_out, ids_in, ids_out,
                                                                (unsigned)(rows * cols), (unsigned)rows, offs, offs + 1, 0, 32, s);
        if (e != hipSuccess) { set_error("segmented_radix_sort_pairs_desc: %s", hipGetErrorString(e)); return TMF_E_LAUNCH; }
        const int64_t want2 = (rows * k + 255) / 256;
        hipLaunchKernelGGL(k_rank_emit, dim3((unsigned)(want2 < 8192 ? want2 : 8192)), dim3(256), 0, s, (const float*)keys_out,
                           (const int32_t*)ids_out, rows, cols, k, out_idx, out_val);
        return check_launch("tmf_topk_stable_f32");
    }
    int npow2 = 1;
    while (npow2 < cols) npow2 <<= 1;
    const size_t lds = (size_t)npow2 * 8;
    static LdsGrant grant;
    if (int rc = grant_dynamic_lds(reinterpret_cast<const void*>(&k_sort_rows), lds, grant)) return rc;
    const int threads = npow2 / 2 < 1024 ? (npow2 / 2 < 64 ? 64 : npow2 / 2) : 1024;
    hipLaunchKernelGGL(k_sort_rows, dim3((unsigned)rows), dim3(threads), lds, (hipStream_t)stream, X, cols, ldx, k,
                       clamp_negatives, npow2, out_idx, out_val);
    return check_launch("tmf_topk_stable_f32");
}

extern "C" int tmf_gather_rows_cols_f32(const float* X, const int64_t* idx, float* out, int64_t rows,
                                        int64_t cols, int64_t k, void* stream) {
    if (rows * k == 0) return TMF_OK;
    TMF_REQUIRE(X && idx && out && rows > 0 && cols > 0 && k > 0, "gather_rows_cols: bad arguments");
    const int64_t want = (rows * k + 255) / 256;
    hipLaunchKernelGGL(k_gather_rows_cols, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0,
                       (hipStream_t)stream, X, idx, out, rows, cols, k);
    return check_launch("tmf_gather_rows_cols_f32");
}

// =============================================================================================
// K7+K8 fused: top-k of U.V^T per user without materialising the [m, n] score matrix
// (recall_at_k / retrieve_user_recs at catalog sizes where 4*m*n bytes do not fit).
//
// One 256-thread workgroup owns 128 users, 32 per wave.  Their rows live in REGISTERS as MFMA A fragments
// for the whole kernel (K/2 floats per lane); item tiles of 128 stream through a 3-slot LDS ring in
// k-chunks of 32 (global loads for chunk g+2 are issued before the MFMAs of chunk g and written to
// LDS after them, one barrier per chunk).  After each 128x128 tile every lane tests its 64
// accumulator values against the per-row threshold (the row's current k-th best, in registers); the few that pass reach
// the row's sorted list under the total order (value desc, index asc) in one of two ways (template CAND):
//   FCAND_PEND (k <= 12)  append to a per-row pending buffer (one returning LDS atomic per candidate), one lane per row merges
//                         the buffer into the sorted list when some row of the wave has filled half of it - the serial,
//                         LDS-latency-bound insertion then runs ~20 times per row block instead of on most tiles;
//   FCAND_INS  (k <= 64)  insert at once with all 64 lanes: lane j holds list entry j, the candidate's position is a ballot +
//                         popcount, everything behind it moves down one lane (DPP wave_shr:1) - no atomics, no pending
//                         buffer, no overflow path, and k = 64 costs per insertion what k = 13 costs; thresholds are refreshed
//                         in registers after every insertion.  One copy of the code walks the rows by a wave-uniform
//                         register index (sixteen unrolled copies overflowed the instruction cache).
//   FCAND_MERGE (round 4, large k)  append like FCAND_PEND (32 pending entries per row); a row whose buffer is half full is merged by
//                         the WHOLE wave: the pending entries are bitonic-sorted worst-first in the upper half-wave (15
//                         compare-exchange stages), one element-wise "keep the better" against the best-first list leaves the 64
//                         best of both as a bitonic sequence, six more stages sort it - 22 stages for up to 32 candidates instead
//                         of one ~160-instruction insertion for each.  The comparator is the total order (value desc, index asc),
//                         so the merged list is THE sorted list whatever the network does with ties.
// Items arrive in ascending index order, so "strictly greater than the k-th value" at the start of a tile is exactly
// tf.math.top_k's tie rule; inside a tile the full comparator decides.  Measured (r = 128, 262144 x 100000, same box,
// TF pending / insertion): k = 10: 118.5 / 115.3, 16: 89.9 / 110.0, 24: 77.1 / 103.6, 32: 65.9 / 75.7, 64: 36.4 / 61.1.
// An insertion costs ~1000 cycles of wave time whatever holds the list (LDS or registers - both were built): it is ~160
// dependent instructions of one wave, not memory; the pending path wins at small k because it batches them.
// =============================================================================================
namespace tmf {

constexpr int FBM = 128, FBN = 128, FBK = 32, FLD = FBN + 1, FCAP = 16, FMAXK = 64;
constexpr int FCAND_PEND = 0, FCAND_INS = 1, FCAND_MERGE = 2;   // candidate handling of k_predict_topk: pending buffers + rare one-lane merges,
                                                                // immediate 64-lane insertion, or pending buffers + wave-wide bitonic merges
constexpr int FPC = 32;   // FCAND_MERGE: pending entries per row
#ifndef TMF_FPT
#define TMF_FPT 24
#endif
constexpr int FPT = TMF_FPT;   // ... and the fill from which a row is merged after a tile (a 16-column group of the overflow path: FPC / 2)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 2: K % 4 == 0 and V < 4 GB (buffer loads, constant per-thread offsets), 1: K % 4 == 0 (branch-free), 0: any K
// OR of a per-lane word over the 64 lanes of a wave (every lane ends with the result): DPP quad permutes and row mirrors, then the
// gfx950 half-row / half-wave swaps - six VALU steps instead of one ballot per bit.
__device__ __forceinline__ unsigned wave_or(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);  // row_half_mirror
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false);  // row_mirror
    auto r = __builtin_amdgcn_permlane16_swap((int)v, (int)v, false, false);
    v = (unsigned)r[0] | (unsigned)r[1];
    r = __builtin_amdgcn_permlane32_swap((int)v, (int)v, false, false);
    return (unsigned)r[0] | (unsigned)r[1];
}

// x of lane (lane ^ STRIDE), without the LDS crossbar: DPP quad permutes (1, 2), two bank-masked row shifts (4), row_ror:8 (8),
// the gfx950 half-row / half-wave swaps + a select (16, 32)
template <int STRIDE>
__device__ __forceinline__ unsigned lane_xor(unsigned x, int lane) {
    const int v = (int)x;
    if constexpr (STRIDE == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
    else if constexpr (STRIDE == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
    else if constexpr (STRIDE == 4) {
        const int t = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xf, 0x5, false);    // row_shl:4 into the lanes with bit 2 clear
        return (unsigned)__builtin_amdgcn_update_dpp(t, v, 0x114, 0xf, 0xA, false);   // row_shr:4 into the others
    } else if constexpr (STRIDE == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);
    else if constexpr (STRIDE == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);   // r[0] = even rows twice, r[1] = odd rows twice
        return (unsigned)((lane & 16) ? r[0] : r[1]);
    } else {
        static_assert(STRIDE == 32, "lane distances 1 .. 32");
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // r[0] = lower half twice, r[1] = upper half twice
        return (unsigned)((lane & 32) ? r[0] : r[1]);
    }
}

// (value, index) as ONE orderable 64-bit key: a ranks before b under (value desc, index asc)  <=>  key(a) > key(b).
// (NaN scores never become candidates; -0 and +0 map to different keys, like the comparator's v == v test they are told apart
// nowhere else: both are only ever produced by the clamp, as +0.)
__device__ __forceinline__ unsigned long long rank_key(float v, int ix) {
    const unsigned u = __float_as_uint(v);
    const unsigned o = u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
    return ((unsigned long long)o << 32) | (unsigned)(0x7fffffff - ix);
}
__device__ __forceinline__ void rank_unkey(unsigned long long key, float& v, int& ix) {
    const unsigned o = (unsigned)(key >> 32);
    v = __uint_as_float(o ^ ((o >> 31) ? 0x80000000u : 0xffffffffu));
    ix = 0x7fffffff - (int)(unsigned)key;
}

// One compare-exchange stage of a sorting network over lane distance STRIDE on 64-bit keys: the lane with the STRIDE bit clear
// keeps the larger key when `big_first`, else the smaller.
template <int STRIDE>
__device__ __forceinline__ void key_cmpx(unsigned long long& key, bool big_first, int lane) {
    const unsigned olo = lane_xor<STRIDE>((unsigned)key, lane), ohi = lane_xor<STRIDE>((unsigned)(key >> 32), lane);
    const unsigned long long other = ((unsigned long long)ohi << 32) | olo;
    const bool keep_big = ((lane & STRIDE) == 0) == big_first;
    key = (keep_big == (other > key)) ? other : key;
}

template <int NCH, int MODE, int CAND>  // K_PAD = 32 * NCH; CAND: how candidates reach the rows' sorted lists (FCAND_*)
__global__ __launch_bounds__(256, NCH >= 8 ? 1 : 2) void k_predict_topk(const float* __restrict__ A, const float* __restrict__ B,
                                                         int64_t m, int64_t n, int K, int64_t lda, int64_t ldb, int k,
                                                         int clamp, int32_t* __restrict__ out_idx,
                                                         float* __restrict__ out_val) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr bool PEND = CAND == FCAND_PEND, MERGE = CAND == FCAND_MERGE;
    float* Bs = reinterpret_cast<float*>(smem_raw);          // [3][FBK][FLD]
    // FCAND_PEND (k <= 12): thresholds, per-row pending buffers and [k][FBM] sorted lists, merged by one lane per row
    float* tau = Bs + 3 * FBK * FLD;                         // [FBM]
    int* cnt = reinterpret_cast<int*>(tau + FBM);            // [FBM]
    int* ovf = cnt + FBM;                                    // [4] (one word used)
    float* pend_v = reinterpret_cast<float*>(ovf + 4);       // [FCAP][FBM]
    int* pend_i = reinterpret_cast<int*>(pend_v + FCAP * FBM);
    float* plist_v = reinterpret_cast<float*>(pend_i + FCAP * FBM);  // [k][FBM]
    int* plist_i = reinterpret_cast<int*>(plist_v + (size_t)k * FBM);
    float* list_v = Bs + 3 * FBK * FLD;                      // FCAND_INS: [FBM][k], every row's k best so far, sorted (value desc, index asc)
    int* list_i = reinterpret_cast<int*>(list_v + (size_t)FBM * k);
    // FCAND_MERGE: tau / cnt / ovf as FCAND_PEND, then row-major pending buffers [FBM][FPC] and row-major lists [FBM][k]
    float* mpend_v = reinterpret_cast<float*>(ovf + 4);
    int* mpend_i = reinterpret_cast<int*>(mpend_v + FPC * FBM);
    float* mlist_v = reinterpret_cast<float*>(mpend_i + FPC * FBM);
    int* mlist_i = reinterpret_cast<int*>(mlist_v + (size_t)FBM * k);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int64_t row0 = (int64_t)blockIdx.x * FBM;

    // ---- A fragments of this wave's 32 users: a[kk] = U[row0 + 32 wave + l31][2 kk + h] ----
    float a[16 * NCH];
    {
        const int64_t r = row0 + 32 * wave + l31;
        const float* p = A + (r < m ? r : 0) * lda;
#pragma unroll
        for (int q = 0; q < 8 * NCH; ++q) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < m) {
                if (4 * q + 3 < K) v = *reinterpret_cast<const float4*>(p + 4 * q);
                else { if (4 * q < K) v.x = p[4 * q]; if (4 * q + 1 < K) v.y = p[4 * q + 1]; if (4 * q + 2 < K) v.z = p[4 * q + 2]; }
            }
            a[2 * q] = h ? v.y : v.x;
            a[2 * q + 1] = h ? v.w : v.z;
        }
    }
    if constexpr (PEND) {
        for (int t = tid; t < FBM; t += 256) {
            tau[t] = (row0 + t < m) ? -INFINITY : INFINITY;  // rows past m never accept anything
            cnt[t] = 0;
            for (int j = 0; j < k; ++j) { plist_v[j * FBM + t] = -INFINITY; plist_i[j * FBM + t] = 0x7fffffff; }
        }
        if (tid == 0) ovf[0] = 0;
    } else if constexpr (MERGE) {
        for (int t = tid; t < FBM; t += 256) {
            tau[t] = (row0 + t < m) ? -INFINITY : INFINITY;
            cnt[t] = 0;
        }
        for (int t = tid; t < FBM * k; t += 256) { mlist_v[t] = -INFINITY; mlist_i[t] = 0x7fffffff; }
    } else {
        for (int t = tid; t < FBM * k; t += 256) { list_v[t] = -INFINITY; list_i[t] = 0x7fffffff; }
    }

    // ---- staging: thread -> (item = tid/8 + 32 q, float4 #tid%8 of the 32-wide k-chunk) ----
    const int s_item = tid >> 3, s_k4 = tid & 7;
    const int64_t ntiles = (n + FBN - 1) / FBN;
    const int64_t nchunks = ntiles * NCH;
    constexpr bool ALIGNED = MODE >= 1;
    uint32_t voff[4];  // MODE 2: byte offset of this thread's four 16-byte pieces inside a (tile, k-chunk) block of V
    const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(B), 0, MODE == 2 ? (int)(uint32_t)(n * ldb * 4) : 0, 0x00020000);
#pragma unroll
    for (int q = 0; q < 4; ++q) voff[q] = (uint32_t)(((int64_t)(s_item + 32 * q) * ldb + 4 * s_k4) * 4);
    float4 stg[2][4];  // two staging register sets, alternating by chunk parity (see the prologue below)
    auto g_load = [&](int64_t g, float4* stage) {
        if constexpr (MODE == 2) {
            // V is read through a buffer descriptor: every thread keeps the same four 32-bit byte offsets for the whole
            // kernel and adds the wave-uniform offset of the (tile, k-chunk) block - a chunk costs a few scalar
            // instructions, four v_add_u32 and four loads.  The hardware range check returns zeros for rows >= n (the
            // ragged last tile and the two prefetches past the end); k-slots >= K are zeroed at the LDS write.
            const uint32_t block_off = (uint32_t)(((g / NCH) * FBN * ldb + 32 * (g % NCH)) * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(vrsrc, (int)(voff[q] + block_off), 0, 0);
                stage[q] = make_float4(__uint_as_float(raw[0]), __uint_as_float(raw[1]), __uint_as_float(raw[2]),
                                       __uint_as_float(raw[3]));
            }
        } else if constexpr (MODE == 1) {
            // Branch-free: chunk and item indices are clamped into range (columns >= n are ignored by offer(), the two
            // prefetches past the last chunk re-read it) and a k-slot past K is zeroed by a select, so a chunk costs
            // four address computations and four 16-byte loads - no exec-mask juggling next to the MFMAs (the select runs when
            // the registers are written to LDS, not here: touching a load's target waits for the load).
            const int64_t gg = g < nchunks ? g : nchunks - 1;
            const int64_t tile = gg / NCH;
            const int c = (int)(gg % NCH);
            const int kk = 32 * c + 4 * s_k4;
            const int kc = kk + 4 <= K ? kk : K - 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int64_t item = tile * FBN + s_item + 32 * q;
                item = item < n ? item : n - 1;
                stage[q] = *reinterpret_cast<const float4*>(B + item * ldb + kc);  // zeroed in s_write_from if kk >= K
            }
        } else {
            const int64_t tile = g / NCH;
            const int c = (int)(g % NCH);
            const int kk = 32 * c + 4 * s_k4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t item = tile * FBN + s_item + 32 * q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (g < nchunks && item < n) {
                    const float* p = B + item * ldb + kk;
                    if (kk + 3 < K) v = *reinterpret_cast<const float4*>(p);
                    else { if (kk < K) v.x = p[0]; if (kk + 1 < K) v.y = p[1]; if (kk + 2 < K) v.z = p[2]; }
                }
                stage[q] = v;
            }
        }
    };
    auto s_write_from = [&](int slot, const float4* src, int c) {  // c = k-chunk of the data in `src`
        float* dst = Bs + slot * FBK * FLD;
        const bool live = !ALIGNED || (32 * c + 4 * s_k4 < K);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int item = s_item + 32 * q;
            dst[(4 * s_k4 + 0) * FLD + item] = live ? src[q].x : 0.f;
            dst[(4 * s_k4 + 1) * FLD + item] = live ? src[q].y : 0.f;
            dst[(4 * s_k4 + 2) * FLD + item] = live ? src[q].z : 0.f;
            dst[(4 * s_k4 + 3) * FLD + item] = live ? src[q].w : 0.f;
        }
    };
    // Global loads run THREE chunks ahead of the MFMAs that consume them: chunk g + 3 is requested at the top of chunk
    // g into one register set, while the other set (requested a chunk earlier, for chunk g + 2) is written to LDS at
    // the end of chunk g - two chunks (>= 3 us) of latency budget instead of one, which the L2-miss / Infinity-Cache
    // latency under load overran (measured: 17 % of the kernel).  The sets swap roles every chunk; with an even NCH
    // the role is the compile-time parity of the chunk index inside the tile, so no register is ever copied (a copy
    // of an in-flight load target would wait for it).  NCH = 1 (r <= 32, far from MFMA-bound) copies instead.
    g_load(0, stg[0]); s_write_from(0, stg[0], 0);
    g_load(1, stg[0]); s_write_from(1, stg[0], 1 % NCH);
    g_load(2, stg[1]);
    __syncthreads();

    f32x16 acc[4];  // the wave's 32 users x the tile's 4 x 32 items
    // Thresholds of the 16 rows this lane holds accumulator elements of - the row's current k-th best (value, index) - live in
    // REGISTERS only: row (q, h) belongs to accumulator element q of the 32 lanes with this h, and an insertion into that row
    // refreshes exactly that register in those lanes.  The steady state of a tile is 16 x (max of the row's four values, one
    // compare) and ONE wave-uniform branch.
    float tq[16];
    int tqi[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const bool live_row = row0 + 32 * wave + (q & 3) + 8 * (q >> 2) + 4 * h < m;
        tq[q] = live_row ? -INFINITY : INFINITY;   // rows past m never accept anything
        tqi[q] = live_row ? 0x7fffffff : -1;
    }
    auto prefilter = [&]() -> unsigned {
        // strict: an equal value can only beat the k-th entry through a lower index, and every entry of an earlier tile has one
        unsigned pass = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float mx = fmaxf(fmaxf(acc[0][q], acc[1][q]), fmaxf(acc[2][q], acc[3][q]));
            if (clamp) mx = fmaxf(mx, 0.f);
            pass |= (mx > tq[q]) ? (1u << q) : 0u;
        }
        return pass;
    };
    // Insert (v, ix) into the sorted list of `row` (wave-uniform arguments) with all 64 lanes: lane j holds entry j, the
    // candidate's position is the number of entries that rank before it (a prefix, by a ballot), everything behind moves
    // down by one lane.  Two LDS reads and two writes, all conflict-free - no atomics, no pending buffer, no serial shifting
    // (k = 64 costs what k = 10 costs).  Returns the row's new k-th entry.
    auto insert = [&](int row, float v, int ix, float& kv, int& ki) {
        float* Lv = list_v + row * k;
        int* Li = list_i + row * k;
        const bool in = lane < k;
        const float mv = in ? Lv[lane] : INFINITY;       // lanes past k: "before everything", so they never count as movers
        const int mi = in ? Li[lane] : -1;
        const unsigned long long ahead = __builtin_amdgcn_ballot_w64(in && before(mv, mi, v, ix));
        const int pos = __builtin_popcountll(ahead);     // entries that stay in front of the candidate
        // entry of the lane below (DPP wave_shr:1 - no LDS round trip; lane 0 keeps its own value, which it never uses)
        const float pv = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(mv), __float_as_int(mv), 0x138, 0xf, 0xf, false));
        const int pi = __builtin_amdgcn_update_dpp(mi, mi, 0x138, 0xf, 0xf, false);
        const float nv = lane < pos ? mv : (lane == pos ? v : pv);
        const int ni = lane < pos ? mi : (lane == pos ? ix : pi);
        if (in && lane >= pos) { Lv[lane] = nv; Li[lane] = ni; }
        kv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nv), k - 1));
        ki = __builtin_amdgcn_readlane(ni, k - 1);
        wave_lds_sync();
    };
    auto take_candidates = [&](int64_t col0, unsigned pass) {
        // ONE copy of the candidate code: the rows that have a candidate anywhere in the wave are walked by a wave-uniform q
        // (register-indexed reads of the accumulators, thresholds and lists) - sixteen unrolled copies of the insertion pushed
        // the kernel past the instruction cache.
        unsigned rows_hit = __builtin_amdgcn_readfirstlane(wave_or(pass));
        while (rows_hit) {
            const int q = __builtin_ctz(rows_hit);
            rows_hit &= rows_hit - 1;
            const bool mine = (pass >> q) & 1u;
            float tv = tq[q];
            int ti = tqi[q];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[j][q];
                if (clamp) v = (v > 0.f) ? v : 0.f;
                const int ix = (int)(col0 + 32 * j + l31);
                bool c = mine && (col0 + 32 * j + l31 < n) && before(v, ix, tv, ti);
                unsigned long long mask = __builtin_amdgcn_ballot_w64(c);
                while (mask) {   // wave-uniform: one candidate per turn, lowest lane first
                    const int src = __builtin_ctzll(mask);
                    const float cv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
                    const int row = 32 * wave + (q & 3) + 8 * (q >> 2) + 4 * (src >> 5);
                    float kv;
                    int ki;
                    insert(row, cv, (int)(col0 + 32 * j + (src & 31)), kv, ki);
                    if (h == (src >> 5)) { tv = kv; ti = ki; }            // the lanes that hold this row
                    c = c && lane != src && before(v, ix, tv, ti);         // the others re-check against the new threshold
                    mask = __builtin_amdgcn_ballot_w64(c);
                }
            }
            tqi[q] = ti;
            tq[q] = tv;
        }
    };

    // ---- FCAND_PEND: append to pending buffers, merge rarely (round-1 / round-2 path; fastest for k <= 12) ----
    auto load_tau = [&]() {
#pragma unroll
        for (int q = 0; q < 16; ++q) tq[q] = tau[32 * wave + (q & 3) + 8 * (q >> 2) + 4 * h];
    };
    if constexpr (PEND || MERGE) load_tau();
    auto offer = [&](int64_t col0, int group, unsigned pass) {
        // group < 0: every column; otherwise only local columns [16 group, 16 group + 16)
        // (a ballot + popcount slot assignment instead of the LDS atomic was measured slower: it makes all 64 lanes walk
        // every row that has a candidate anywhere in the wave)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (!((pass >> q) & 1u)) continue;
            const int row = 32 * wave + (q & 3) + 8 * (q >> 2) + 4 * h;
            const float t = tau[row];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int lc = 32 * j + l31;
                float v = acc[j][q];
                if (clamp) v = (v > 0.f) ? v : 0.f;
                const bool in_group = (group < 0) || ((lc >> 4) == group);
                if (in_group && (col0 + lc < n) && v > t) {
                    const int pos = atomicAdd(&cnt[row], 1);
                    if (pos < FCAP) { pend_v[pos * FBM + row] = v; pend_i[pos * FBM + row] = (int)(col0 + lc); }
                }
            }
        }
    };
    auto merge_wave = [&]() {  // lanes 0..31 of a wave merge the pending candidates of the wave's own rows
        if (h == 0) {
            const int row = 32 * wave + l31;
            const int c = cnt[row] < FCAP ? cnt[row] : FCAP;
            for (int p = 0; p < c; ++p) {
                const float v = pend_v[p * FBM + row];
                const int ix = pend_i[p * FBM + row];
                int j = k - 1;
                if (before(v, ix, plist_v[j * FBM + row], plist_i[j * FBM + row])) {
                    while (j > 0 && before(v, ix, plist_v[(j - 1) * FBM + row], plist_i[(j - 1) * FBM + row])) {
                        plist_v[j * FBM + row] = plist_v[(j - 1) * FBM + row];
                        plist_i[j * FBM + row] = plist_i[(j - 1) * FBM + row];
                        --j;
                    }
                    plist_v[j * FBM + row] = v;
                    plist_i[j * FBM + row] = ix;
                }
            }
            cnt[row] = 0;
            if (row0 + row < m) tau[row] = plist_v[(k - 1) * FBM + row];
        }
    };

    // ---- FCAND_MERGE: append like FCAND_PEND (row-major buffers), merge a row with the whole wave ----
    auto offer_m = [&](int64_t col0, int group, unsigned pass) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (!((pass >> q) & 1u)) continue;
            const int row = 32 * wave + (q & 3) + 8 * (q >> 2) + 4 * h;
            const float t = tau[row];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int lc = 32 * j + l31;
                float v = acc[j][q];
                if (clamp) v = (v > 0.f) ? v : 0.f;
                const bool in_group = (group < 0) || ((lc >> 4) == group);
                if (in_group && (col0 + lc < n) && v > t) {
                    const int pos = atomicAdd(&cnt[row], 1);
                    if (pos < FPC) { mpend_v[row * FPC + pos] = v; mpend_i[row * FPC + pos] = (int)(col0 + lc); }
                }
            }
        }
    };
    auto merge_row = [&](int row) {   // wave-uniform row; all 64 lanes
        const int c = cnt[row] < FPC ? cnt[row] : FPC;
        const unsigned long long worst = rank_key(-INFINITY, 0x7fffffff);
        // the list, best first (largest key first); lanes past k: worst
        unsigned long long a = lane < k ? rank_key(mlist_v[row * k + lane], mlist_i[row * k + lane]) : worst;
        // the pending entries in the TOP lanes of the wave - 8, 16 or 32 of them, whichever holds them all - bitonic-sorted worst
        // (smallest key) first; the lanes below hold equal "worst" keys: already in place
        const int W = c <= 8 ? 8 : (c <= 16 ? 16 : 32);   // wave-uniform
        const int pl = lane - (64 - W);
        unsigned long long pk = (pl >= 0 && pl < c) ? rank_key(mpend_v[row * FPC + pl], mpend_i[row * FPC + pl]) : worst;
        key_cmpx<1>(pk, (lane & 2) != 0, lane);
        key_cmpx<2>(pk, (lane & 4) != 0, lane);
        key_cmpx<1>(pk, (lane & 4) != 0, lane);
        key_cmpx<4>(pk, W > 8 && (lane & 8) != 0, lane);
        key_cmpx<2>(pk, W > 8 && (lane & 8) != 0, lane);
        key_cmpx<1>(pk, W > 8 && (lane & 8) != 0, lane);
        if (W > 8) {
            key_cmpx<8>(pk, W > 16 && (lane & 16) != 0, lane);
            key_cmpx<4>(pk, W > 16 && (lane & 16) != 0, lane);
            key_cmpx<2>(pk, W > 16 && (lane & 16) != 0, lane);
            key_cmpx<1>(pk, W > 16 && (lane & 16) != 0, lane);
        }
        if (W > 16) {
            key_cmpx<16>(pk, false, lane);
            key_cmpx<8>(pk, false, lane);
            key_cmpx<4>(pk, false, lane);
            key_cmpx<2>(pk, false, lane);
            key_cmpx<1>(pk, false, lane);
        }
        // list best-first against pending worst-first: the better of each pair = the 64 best of both, as a bitonic sequence
        a = pk > a ? pk : a;
        key_cmpx<32>(a, true, lane);
        key_cmpx<16>(a, true, lane);
        key_cmpx<8>(a, true, lane);
        key_cmpx<4>(a, true, lane);
        key_cmpx<2>(a, true, lane);
        key_cmpx<1>(a, true, lane);
        float av;
        int ai;
        rank_unkey(a, av, ai);
        if (lane < k) { mlist_v[row * k + lane] = av; mlist_i[row * k + lane] = ai; }
        const float kth = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(av), k - 1));
        if (lane == 0) {
            cnt[row] = 0;
            tau[row] = kth;   // rows past m never get here (their threshold is +inf: nothing is ever appended)
        }
    };
    auto merge_rows = [&](unsigned rows) {   // wave-uniform mask over the wave's 32 rows
        wave_lds_sync();
        while (rows) {
            const int l = __builtin_ctz(rows);
            rows &= rows - 1;
            merge_row(32 * wave + l);
        }
        wave_lds_sync();
    };

    int64_t g = 0;
    int c_prev = 0;  // FCAND_PEND / FCAND_MERGE, lanes h == 0: pending entries of the lane's row that predate the current tile
    for (int64_t tile = 0; tile < ntiles; ++tile) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c, ++g) {
            g_load(g + 3, stg[c & 1]);
            const float* bs = Bs + (int)(g % 3) * FBK * FLD + h * FLD + l31;
            // B operands one k-step ahead of the MFMAs that consume them (two register sets): the LDS latency of
            // step ks + 1 hides behind the four MFMAs of step ks instead of stalling every second MFMA
            float bq[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bq[0][j] = bs[32 * j];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                if (ks + 1 < 16) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bq[(ks + 1) & 1][j] = bs[2 * (ks + 1) * FLD + 32 * j];
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the reads of step ks + 1 ahead of the MFMAs of step ks
                const float av = a[16 * c + ks];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bq[ks & 1][j], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            s_write_from((int)((g + 2) % 3), stg[(c & 1) ^ 1], (c + 2) % NCH);
            if constexpr (NCH == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) stg[1][q] = stg[0][q];
            }
            __syncthreads();
        }
        // top-k update: the 32 rows of a wave are touched by that wave only (sorted lists in LDS, thresholds in registers), so
        // this part needs no workgroup barrier - LDS operations of one wave complete in order.
        if constexpr (PEND) {
            const int64_t col0 = tile * FBN;
            const unsigned pass = prefilter();
            if (__any(pass != 0u)) {  // wave-uniform
                // Candidates are only APPENDED to the row's pending buffer here; the sorted lists (and with them the
                // thresholds) are brought up to date when some row of the wave has filled half of its buffer.  A stale
                // threshold is a lower bound of the true one, so nothing is lost - a few more candidates are appended and
                // rejected by the merge - while the serial, LDS-latency-bound insertion runs ~20 times per row block
                // instead of on most tiles.
                offer(col0, -1, pass);
                const int my_row = 32 * wave + l31;
                const int c_now = (h == 0) ? cnt[my_row] : 0;
                if (__any(c_now > FCAP)) {
                    // overflow: drop this tile's partial appends (keep the older ones), merge, then re-offer the tile in
                    // 8 groups of 16 columns with a merge after each (thresholds only rise: `pass` stays a superset)
                    if (h == 0) cnt[my_row] = c_prev;
                    merge_wave();
                    for (int grp = 0; grp < FBN / 16; ++grp) {
                        offer(col0, grp, pass);
                        merge_wave();
                    }
                    c_prev = 0;
                    load_tau();
                } else if (__any(c_now > FCAP / 2)) {
                    merge_wave();
                    c_prev = 0;
                    load_tau();
                } else {
                    c_prev = c_now;
                }
            }

        } else if constexpr (MERGE) {
            const int64_t col0 = tile * FBN;
            const unsigned pass = prefilter();
            if (__any(pass != 0u)) {  // wave-uniform
                offer_m(col0, -1, pass);
                wave_lds_sync();
                const int my_row = 32 * wave + l31;
                const int c_now = (h == 0) ? cnt[my_row] : 0;
                if (__any(c_now > FPC)) {
                    // overflow: drop this tile's partial appends (keep the older ones), merge, then re-offer the tile in 8 groups
                    // of 16 columns with a merge after each (thresholds only rise: `pass` stays a superset)
                    // Only rows whose buffer is half full are merged: a group adds at most 16 entries to a row, so a row below
                    // FPC / 2 cannot overflow in the next group (merging every row with anything pending after every group cost
                    // 256 merges per tile in the first tiles of a scan).
                    if (h == 0) cnt[my_row] = c_prev;
                    merge_rows((unsigned)__builtin_amdgcn_ballot_w64(h == 0 && c_prev >= FPC / 2));
                    for (int grp = 0; grp < FBN / 16; ++grp) {
                        offer_m(col0, grp, pass);
                        wave_lds_sync();
                        merge_rows((unsigned)__builtin_amdgcn_ballot_w64(h == 0 && cnt[my_row] >= FPC / 2));
                    }
                    c_prev = (h == 0) ? cnt[my_row] : 0;
                    load_tau();
                } else {
                    const unsigned due = (unsigned)__builtin_amdgcn_ballot_w64(h == 0 && c_now >= FPT);
                    if (due) {
                        merge_rows(due);
                        load_tau();
                    }
                    c_prev = (h == 0) ? cnt[my_row] : 0;
                }
            }
        } else {
            const unsigned pass = prefilter();
            if (__builtin_amdgcn_ballot_w64(pass != 0u) != 0) take_candidates(tile * FBN, pass);   // wave-uniform
        }
    }
    if constexpr (PEND) merge_wave();  // whatever is still pending
    if constexpr (MERGE) merge_rows((unsigned)__builtin_amdgcn_ballot_w64(h == 0 && cnt[32 * wave + l31] > 0));
    __syncthreads();
    if constexpr (PEND) {
        if (tid < FBM && row0 + tid < m) {
            for (int j = 0; j < k; ++j) {
                out_idx[(row0 + tid) * k + j] = plist_i[j * FBM + tid];
                if (out_val) out_val[(row0 + tid) * k + j] = plist_v[j * FBM + tid];
            }
        }
    } else {
        // the lists are [row][k] like the output: one coalesced copy
        const float* Lv = MERGE ? mlist_v : list_v;
        const int* Li = MERGE ? mlist_i : list_i;
        const int64_t live = (m - row0 < FBM) ? m - row0 : FBM;
        for (int64_t t = tid; t < live * k; t += 256) {
            out_idx[row0 * k + t] = Li[t];
            if (out_val) out_val[row0 * k + t] = Lv[t];
        }
    }
}

template <int NCH, int MODE, int CAND>
static int launch_predict_topk_mode(const float* A, const float* B, int64_t m, int64_t n, int K, int64_t lda, int64_t ldb,
                               int k, int clamp, int32_t* out_idx, float* out_val, hipStream_t stream) {
    const size_t lds = CAND == FCAND_PEND ? sizeof(float) * (3 * FBK * FLD + FBM) + sizeof(int) * (FBM + 4) + 8 * (size_t)FCAP * FBM + 8 * (size_t)k * FBM
                       : CAND == FCAND_MERGE ? sizeof(float) * (3 * FBK * FLD + FBM) + sizeof(int) * (FBM + 4) + 8 * (size_t)FPC * FBM + 8 * (size_t)k * FBM
                       : sizeof(float) * (3 * FBK * FLD) + 8 * (size_t)k * FBM;
    static LdsGrant grant;  // per template instance
    if (int rc = grant_dynamic_lds(reinterpret_cast<const void*>(&k_predict_topk<NCH, MODE, CAND>), lds, grant)) return rc;
    const int64_t blocks = (m + FBM - 1) / FBM;
    hipLaunchKernelGGL((k_predict_topk<NCH, MODE, CAND>), dim3((unsigned)blocks), dim3(256), lds, stream, A, B, m, n, K, lda, ldb, k,
                       clamp, out_idx, out_val);
    return check_launch("tmf_predict_topk_f32");
}

template <int NCH, int MODE>
static int launch_predict_topk_impl(const float* A, const float* B, int64_t m, int64_t n, int K, int64_t lda, int64_t ldb,
                               int k, int clamp, int32_t* out_idx, float* out_val, hipStream_t stream) {
    // same box, r = 128, TF by k (pending + merge / insertion): 10: 118.5 / 115.3   16: 89.9 / 110.0   24: 77.1 / 103.6
    // 32: 65.9 / 75.7   48: 48.3 / 67.3   64: 36.4 / 61.1 (profiles/r03_predict_candidates.txt)
    // round 4, same box, r = 128, TF (insertion / wave-wide merges): 16: 109.4 / 94.8   24: 103.4 / 91.4   32: 76.2 / 88.6   48: 67.4 / 83.8
    // 64: 61.1 / 80.1 - the insertion path keeps two workgroups per CU up to k = 27, beyond that the merges win
    int cand = k <= 12 ? FCAND_PEND : (k <= 27 ? FCAND_INS : FCAND_MERGE);
    if (const char* env = getenv("TMF_PREDICT_CAND")) cand = atoi(env);   // A/B runs: 0 pending, 1 insertion, 2 wave-wide merges
    if (cand == FCAND_PEND) return launch_predict_topk_mode<NCH, MODE, FCAND_PEND>(A, B, m, n, K, lda, ldb, k, clamp, out_idx, out_val, stream);
    if (cand == FCAND_MERGE) return launch_predict_topk_mode<NCH, MODE, FCAND_MERGE>(A, B, m, n, K, lda, ldb, k, clamp, out_idx, out_val, stream);
    return launch_predict_topk_mode<NCH, MODE, FCAND_INS>(A, B, m, n, K, lda, ldb, k, clamp, out_idx, out_val, stream);
}

template <int NCH>
static int launch_predict_topk(const float* A, const float* B, int64_t m, int64_t n, int K, int64_t lda, int64_t ldb,
                               int k, int clamp, int32_t* out_idx, float* out_val, hipStream_t stream) {
    if (K % 4 == 0 && (n + 4 * FBN) * ldb * 4 < ((int64_t)1 << 32))  // 32-bit byte offsets into V
        return launch_predict_topk_impl<NCH, 2>(A, B, m, n, K, lda, ldb, k, clamp, out_idx, out_val, stream);
    if (K % 4 == 0) return launch_predict_topk_impl<NCH, 1>(A, B, m, n, K, lda, ldb, k, clamp, out_idx, out_val, stream);
    return launch_predict_topk_impl<NCH, 0>(A, B, m, n, K, lda, ldb, k, clamp, out_idx, out_val, stream);
}

}  // namespace tmf

extern "C" int tmf_predict_topk_f32(const float* A, const float* B, int64_t m, int64_t n, int r, int64_t lda,
                                    int64_t ldb, int k, int clamp_negatives, int32_t* out_idx, float* out_val,
                                    void* stream) {
    if (m == 0) return TMF_OK;
    TMF_REQUIRE(A && B && out_idx && m > 0 && n > 0 && r > 0, "predict_topk: bad arguments");
    TMF_REQUIRE(lda >= r && ldb >= r && (lda % 4 == 0) && (ldb % 4 == 0) && ((uintptr_t)A % 16 == 0) &&
                    ((uintptr_t)B % 16 == 0), "predict_topk: operands must be 16-byte aligned with ld %% 4 == 0");
    TMF_REQUIRE(k >= 1 && k <= n, "predict_topk: k=%d must be in [1, n=%lld]", k, (long long)n);
    TMF_REQUIRE(n < ((int64_t)1 << 31), "predict_topk: too many items");
    if (k > tmf::FMAXK || r > 256) {
        tmf::set_error("predict_topk: fused kernel supports k <= %d and n_components <= 256 (got k=%d, r=%d)", tmf::FMAXK, k, r);
        return TMF_E_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    if (r <= 32) return tmf::launch_predict_topk<1>(A, B, m, n, r, lda, ldb, k, clamp_negatives, out_idx, out_val, s);
    if (r <= 64) return tmf::launch_predict_topk<2>(A, B, m, n, r, lda, ldb, k, clamp_negatives, out_idx, out_val, s);
    if (r <= 128) return tmf::launch_predict_topk<4>(A, B, m, n, r, lda, ldb, k, clamp_negatives, out_idx, out_val, s);
    return tmf::launch_predict_topk<8>(A, B, m, n, r, lda, ldb, k, clamp_negatives, out_idx, out_val, s);  // 128 A registers per lane
}

// =============================================================================================
// Fused top-k for bf16-stored factors (BASELINE config 5): the same structure as k_predict_topk on the bf16
// MFMA (v_mfma_f32_32x32x16_bf16: exact bf16 products, fp32 accumulation - i.e. "bf16 factors / fp32 accum").
// 512 threads = 8 waves own 256 users (32 per wave, rows in registers as A fragments: 8 bf16 per lane and
// k-step); item tiles of 128 stream through a 3-slot LDS ring in k-chunks of 64 (128-byte rows padded to 144
// bytes: 16 consecutive rows then hit 16 different 16-byte bank slots, so the ds_read_b128 operand reads are
// conflict-free).  256 users per workgroup halve the V bytes streamed per flop, which is what bounds this kernel.
// =============================================================================================
namespace tmf {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
constexpr int HBM_ = 256, HBN = 128, HBK = 64, HROW = 144 /* bytes */, HCAP = 16, HMAXK = 32;

// MODE 2: K % 8 == 0 and V < 4 GB (buffer loads with constant per-thread offsets, 3 chunks of read-ahead); 0: any K
template <int NCH, int MODE>  // K_PAD = 64 * NCH
__global__ __launch_bounds__(512, 2) void k_predict_topk_bf16(const __bf16* __restrict__ A, const __bf16* __restrict__ B,
                                                              int64_t m, int64_t n, int K, int64_t lda, int64_t ldb, int k,
                                                              int clamp, int32_t* __restrict__ out_idx,
                                                              float* __restrict__ out_val) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* Bs = smem_raw;                                               // [3][HBN][HROW] bytes
    float* tau = reinterpret_cast<float*>(Bs + 3 * HBN * HROW);        // [HBM_]
    int* cnt = reinterpret_cast<int*>(tau + HBM_);                     // [HBM_]
    float* pend_v = reinterpret_cast<float*>(cnt + HBM_);              // [HCAP][HBM_]
    int* pend_i = reinterpret_cast<int*>(pend_v + HCAP * HBM_);
    float* list_v = reinterpret_cast<float*>(pend_i + HCAP * HBM_);    // [k][HBM_]
    int* list_i = reinterpret_cast<int*>(list_v + (size_t)k * HBM_);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int64_t row0 = (int64_t)blockIdx.x * HBM_;

    // A fragments: a[kk] = U[row0 + 32 wave + l31][16 kk + 8 h .. + 8)
    bf16x8_t a[4 * NCH];
    {
        const int64_t r = row0 + 32 * wave + l31;
        const __bf16* p = A + (r < m ? r : 0) * lda;
#pragma unroll
        for (int kk = 0; kk < 4 * NCH; ++kk) {
            bf16x8_t v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.0f;
            const int k0 = 16 * kk + 8 * h;
            if (r < m && k0 < K) {
                if (k0 + 7 < K) v = *reinterpret_cast<const bf16x8_t*>(p + k0);
                else
                    for (int e = 0; e < 8; ++e) if (k0 + e < K) v[e] = p[k0 + e];
            }
            a[kk] = v;
        }
    }
    for (int t = tid; t < HBM_; t += 512) {
        tau[t] = (row0 + t < m) ? -INFINITY : INFINITY;
        cnt[t] = 0;
        for (int j = 0; j < k; ++j) { list_v[j * HBM_ + t] = -INFINITY; list_i[j * HBM_ + t] = 0x7fffffff; }
    }

    // staging: thread -> (item = tid/8 + 64 q, 16-byte slot tid%8 of the 128-byte k-chunk), q = 0, 1
    const int s_item = tid >> 3, s_slot = tid & 7;
    const int64_t ntiles = (n + HBN - 1) / HBN;
    const int64_t nchunks = ntiles * NCH;
    uint32_t voff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) voff[q] = (uint32_t)(((int64_t)(s_item + 64 * q) * ldb + 8 * s_slot) * 2);
    const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(B), 0, MODE == 2 ? (int)(uint32_t)(n * ldb * 2) : 0, 0x00020000);
    bf16x8_t stg[2][2];  // two staging register sets, alternating by chunk parity (as in k_predict_topk)
    auto g_load = [&](int64_t g, bf16x8_t* stage) {
        if constexpr (MODE == 2) {
            // rows >= n come back as zeros from the buffer range check; k-slots >= K are zeroed at the LDS write
            const uint32_t block_off = (uint32_t)(((g / NCH) * HBN * ldb + 64 * (g % NCH)) * 2);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(vrsrc, (int)(voff[q] + block_off), 0, 0);
                stage[q] = __builtin_bit_cast(bf16x8_t, raw);
            }
        } else {
            const int64_t tile = g / NCH;
            const int c = (int)(g % NCH);
            const int kk = 64 * c + 8 * s_slot;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int64_t item = tile * HBN + s_item + 64 * q;
                bf16x8_t v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.0f;
                if (g < nchunks && item < n && kk < K) {
                    const __bf16* p = B + item * ldb + kk;
                    if (kk + 7 < K) v = *reinterpret_cast<const bf16x8_t*>(p);
                    else
                        for (int e = 0; e < 8; ++e) if (kk + e < K) v[e] = p[e];
                }
                stage[q] = v;
            }
        }
    };
    auto s_write_from = [&](int slot, const bf16x8_t* src, int c) {  // c = k-chunk of the data in `src`
        char* dst = Bs + slot * HBN * HROW;
        const bool live = MODE != 2 || (64 * c + 8 * s_slot < K);
        bf16x8_t zero;
#pragma unroll
        for (int e = 0; e < 8; ++e) zero[e] = (__bf16)0.0f;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            *reinterpret_cast<bf16x8_t*>(dst + (s_item + 64 * q) * HROW + s_slot * 16) = live ? src[q] : zero;
    };
    g_load(0, stg[0]); s_write_from(0, stg[0], 0);
    g_load(1, stg[0]); s_write_from(1, stg[0], 1 % NCH);
    g_load(2, stg[1]);
    __syncthreads();

    f32x16 acc[4];
    float tq[16];  // register copy of this lane's 16 row thresholds (see k_predict_topk)
    auto load_tau = [&]() {
#pragma unroll
        for (int q = 0; q < 16; ++q) tq[q] = tau[32 * wave + (q & 3) + 8 * (q >> 2) + 4 * h];
    };
    load_tau();
    auto prefilter = [&]() -> unsigned {
        unsigned pass = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float mx = fmaxf(fmaxf(acc[0][q], acc[1][q]), fmaxf(acc[2][q], acc[3][q]));
            if (clamp) mx = fmaxf(mx, 0.f);
            pass |= (mx > tq[q]) ? (1u << q) : 0u;
        }
        return pass;
    };
    auto offer = [&](int64_t col0, int group, unsigned pass) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (!((pass >> q) & 1u)) continue;
            const int row = 32 * wave + (q & 3) + 8 * (q >> 2) + 4 * h;
            const float t = tau[row];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int lc = 32 * j + l31;
                float v = acc[j][q];
                if (clamp) v = (v > 0.f) ? v : 0.f;
                const bool in_group = (group < 0) || ((lc >> 3) == group);
                if (in_group && (col0 + lc < n) && v > t) {
                    const int pos = atomicAdd(&cnt[row], 1);
                    if (pos < HCAP) { pend_v[pos * HBM_ + row] = v; pend_i[pos * HBM_ + row] = (int)(col0 + lc); }
                }
            }
        }
    };
    auto merge_wave = [&]() {
        if (h == 0) {
            const int row = 32 * wave + l31;
            const int c = cnt[row] < HCAP ? cnt[row] : HCAP;
            for (int p = 0; p < c; ++p) {
                const float v = pend_v[p * HBM_ + row];
                const int ix = pend_i[p * HBM_ + row];
                int j = k - 1;
                if (before(v, ix, list_v[j * HBM_ + row], list_i[j * HBM_ + row])) {
                    while (j > 0 && before(v, ix, list_v[(j - 1) * HBM_ + row], list_i[(j - 1) * HBM_ + row])) {
                        list_v[j * HBM_ + row] = list_v[(j - 1) * HBM_ + row];
                        list_i[j * HBM_ + row] = list_i[(j - 1) * HBM_ + row];
                        --j;
                    }
                    list_v[j * HBM_ + row] = v;
                    list_i[j * HBM_ + row] = ix;
                }
            }
            cnt[row] = 0;
            if (row0 + row < m) tau[row] = list_v[(k - 1) * HBM_ + row];
        }
    };

    int64_t g = 0;
    int c_prev = 0;
    for (int64_t tile = 0; tile < ntiles; ++tile) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c, ++g) {
            g_load(g + 3, stg[c & 1]);
            const char* bs = Bs + (int)(g % 3) * HBN * HROW + l31 * HROW + h * 16;
            bf16x8_t bq[2][4];  // B operands one k-step ahead of their MFMAs (see k_predict_topk)
#pragma unroll
            for (int j = 0; j < 4; ++j) bq[0][j] = *reinterpret_cast<const bf16x8_t*>(bs + 32 * j * HROW);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        bq[(ks + 1) & 1][j] = *reinterpret_cast<const bf16x8_t*>(bs + 32 * j * HROW + (ks + 1) * 32);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[4 * c + ks], bq[ks & 1][j], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            s_write_from((int)((g + 2) % 3), stg[(c & 1) ^ 1], (c + 2) % NCH);
            if constexpr (NCH == 1) {
#pragma unroll
                for (int q = 0; q < 2; ++q) stg[1][q] = stg[0][q];
            }
            __syncthreads();
        }
        const int64_t col0 = tile * HBN;
        const unsigned pass = prefilter();
        if (__any(pass != 0u)) {  // candidates are appended; lists and thresholds catch up when a buffer is half full
            offer(col0, -1, pass);
            const int my_row = 32 * wave + l31;
            const int c_now = (h == 0) ? cnt[my_row] : 0;
            if (__any(c_now > HCAP)) {  // overflow: keep the older appends, re-offer this tile in 16 groups of 8 columns
                if (h == 0) cnt[my_row] = c_prev;
                merge_wave();
                for (int grp = 0; grp < HBN / 8; ++grp) {
                    offer(col0, grp, pass);
                    merge_wave();
                }
                c_prev = 0;
                load_tau();
            } else if (__any(c_now > HCAP / 2)) {
                merge_wave();
                c_prev = 0;
                load_tau();
            } else {
                c_prev = c_now;
            }
        }
    }
    merge_wave();  // whatever is still pending
    __syncthreads();
    if (tid < HBM_ && row0 + tid < m) {
        for (int j = 0; j < k; ++j) {
            out_idx[(row0 + tid) * k + j] = list_i[j * HBM_ + tid];
            if (out_val) out_val[(row0 + tid) * k + j] = list_v[j * HBM_ + tid];
        }
    }
}

template <int NCH, int MODE>
static int launch_predict_topk_bf16_impl(const void* A, const void* B, int64_t m, int64_t n, int K, int64_t lda, int64_t ldb,
                                    int k, int clamp, int32_t* out_idx, float* out_val, hipStream_t stream) {
    const size_t lds = (size_t)3 * HBN * HROW + sizeof(float) * HBM_ + sizeof(int) * HBM_ + 8 * (size_t)HCAP * HBM_ +
                       8 * (size_t)k * HBM_;
    static LdsGrant grant;  // per template instance
    if (int rc = grant_dynamic_lds(reinterpret_cast<const void*>(&k_predict_topk_bf16<NCH, MODE>), lds, grant)) return rc;
    const int64_t blocks = (m + HBM_ - 1) / HBM_;
    hipLaunchKernelGGL((k_predict_topk_bf16<NCH, MODE>), dim3((unsigned)blocks), dim3(512), lds, stream, (const __bf16*)A,
                       (const __bf16*)B, m, n, K, lda, ldb, k, clamp, out_idx, out_val);
    return check_launch("tmf_predict_topk_bf16");
}

template <int NCH>
static int launch_predict_topk_bf16(const void* A, const void* B, int64_t m, int64_t n, int K, int64_t lda, int64_t ldb,
                                    int k, int clamp, int32_t* out_idx, float* out_val, hipStream_t stream) {
    if (K % 8 == 0 && (n + 4 * HBN) * ldb * 2 < ((int64_t)1 << 32))
        return launch_predict_topk_bf16_impl<NCH, 2>(A, B, m, n, K, lda, ldb, k, clamp, out_idx, out_val, stream);
    return launch_predict_topk_bf16_impl<NCH, 0>(A, B, m, n, K, lda, ldb, k, clamp, out_idx, out_val, stream);
}

}  // namespace tmf

extern "C" int tmf_predict_topk_bf16(const void* A, const void* B, int64_t m, int64_t n, int r, int64_t lda,
                                     int64_t ldb, int k, int clamp_negatives, int32_t* out_idx, float* out_val,
                                     void* stream) {
    if (m == 0) return TMF_OK;
    TMF_REQUIRE(A && B && out_idx && m > 0 && n > 0 && r > 0, "predict_topk_bf16: bad arguments");
    TMF_REQUIRE(lda >= r && ldb >= r && (lda % 8 == 0) && (ldb % 8 == 0) && ((uintptr_t)A % 16 == 0) &&
                    ((uintptr_t)B % 16 == 0), "predict_topk_bf16: operands must be 16-byte aligned with ld %% 8 == 0");
    TMF_REQUIRE(k >= 1 && k <= n, "predict_topk_bf16: k=%d must be in [1, n=%lld]", k, (long long)n);
    TMF_REQUIRE(n < ((int64_t)1 << 31), "predict_topk_bf16: too many items");
    if (k > tmf::HMAXK || r > 256) {
        tmf::set_error("predict_topk_bf16: supports k <= %d and n_components <= 256 (got k=%d, r=%d)", tmf::HMAXK, k, r);
        return TMF_E_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    if (r <= 64) return tmf::launch_predict_topk_bf16<1>(A, B, m, n, r, lda, ldb, k, clamp_negatives, out_idx, out_val, s);
    if (r <= 128) return tmf::launch_predict_topk_bf16<2>(A, B, m, n, r, lda, ldb, k, clamp_negatives, out_idx, out_val, s);
    return tmf::launch_predict_topk_bf16<4>(A, B, m, n, r, lda, ldb, k, clamp_negatives, out_idx, out_val, s);
}
