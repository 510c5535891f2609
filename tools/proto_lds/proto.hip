// Prototype: sampled scores with the V tile staged in LDS and ONE USER PER LANE (its U row in registers).
// sp[u, s] = <U[u], V[Rs[u, s]]>, Rs[u, :] ascending.  fp32, rank 128 (512-byte rows).
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int RK = 128;      // rank
constexpr int STRIDE = 132;  // floats per LDS row (528 B): breaks the 512-byte bank alignment of random rows

template <int THREADS, int CH>
__global__ __launch_bounds__(THREADS) void k_scores_lds(const float* __restrict__ U, const float* __restrict__ V,
                                                        const int32_t* __restrict__ Rs, float* __restrict__ sp, int m, int n,
                                                        int S, int T, int mode) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int tid = threadIdx.x;
    for (int64_t ub = (int64_t)blockIdx.x * THREADS; ub < m; ub += (int64_t)gridDim.x * THREADS) {
        const int64_t u = ub + tid;
        const bool live = u < m;
        float x[RK];
        {
            const float4* up = reinterpret_cast<const float4*>(U + (live ? u : 0) * RK);
#pragma unroll
            for (int c = 0; c < RK / 4; ++c) {
                const float4 v = up[c];
                x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
            }
        }
        const int32_t* ids = Rs + (live ? u : 0) * (int64_t)S;
        float* out = sp + (live ? u : 0) * (int64_t)S;
        int cur = live ? 0 : S;
        int4 q = *reinterpret_cast<const int4*>(ids);        // ids[0..3]
        int4 qn = *reinterpret_cast<const int4*>(ids + 4);   // ids[4..7]  (S >= 8)
        int nid = q.x;
        float chk = 0.f;
        const int fake_step = 60 + (tid * 7) % 77;
        for (int lo = 0; lo < n; lo += T) {
            const int hi = (lo + T < n) ? lo + T : n;
            __syncthreads();   // the previous tile has been consumed
            if (mode != 1) {
                // the tile is (hi - lo) * 32 float4, contiguous in V: LD loads in flight per thread, then the LDS writes
                constexpr int LD = 4;
                const float4* src = reinterpret_cast<const float4*>(V + (int64_t)lo * RK);
                const int total = (hi - lo) * (RK / 4);
                for (int i0 = tid; i0 < total; i0 += THREADS * LD) {
                    float4 v[LD];
#pragma unroll
                    for (int t = 0; t < LD; ++t) {
                        const int i = i0 + t * THREADS;
                        v[t] = src[i < total ? i : total - 1];
                    }
#pragma unroll
                    for (int t = 0; t < LD; ++t) {
                        const int i = i0 + t * THREADS;
                        if (i < total) *reinterpret_cast<float4*>(tile + (i >> 5) * STRIDE + 4 * (i & 31)) = v[t];
                    }
                }
            }
            __syncthreads();
            while (mode != 2 && cur < S && nid < hi) {
                const float* row = tile + (nid - lo) * STRIDE;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int c0 = 0; c0 < RK / 4; c0 += CH) {
                    float4 v[CH];
#pragma unroll
                    for (int c = 0; c < CH; ++c) v[c] = *reinterpret_cast<const float4*>(row + 4 * (c0 + c));
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        a0 = fmaf(x[4 * (c0 + c)], v[c].x, a0);
                        a1 = fmaf(x[4 * (c0 + c) + 1], v[c].y, a1);
                        a2 = fmaf(x[4 * (c0 + c) + 2], v[c].z, a2);
                        a3 = fmaf(x[4 * (c0 + c) + 3], v[c].w, a3);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep the reads of the next batch behind this batch's FMAs
                }
                if (mode < 3) out[cur] = (a0 + a1) + (a2 + a3); else chk += (a0 + a1) + (a2 + a3);
                ++cur;
                if (mode == 4) { nid += fake_step; continue; }
                const int k = cur & 3;
                if (k == 0) {
                    q = qn;
                    if (cur + 4 < S) qn = *reinterpret_cast<const int4*>(ids + cur + 4);
                }
                nid = (k == 0) ? q.x : (k == 1) ? q.y : (k == 2) ? q.z : q.w;
            }
        }
        if (mode >= 3 && live) out[0] = chk;
    }
}

extern "C" int proto_scores_lds(const float* U, const float* V, const int32_t* Rs, float* sp, int m, int n, int S, int T,
                                int threads, int blocks, int mode, void* stream) {
    const size_t lds = (size_t)T * STRIDE * sizeof(float);
    hipError_t e;
#define GO(TH)                                                                                                            \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scores_lds<TH, CHV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)lds);                                                                                    \
    if (e != hipSuccess) return -1;                                                                                       \
    hipLaunchKernelGGL((k_scores_lds<TH, CHV>), dim3(blocks), dim3(TH), lds, (hipStream_t)stream, U, V, Rs, sp, m, n, S, T, mode)
    if (threads == 512) { constexpr int CHV = 8; GO(512); }
    else if (threads == 768) { constexpr int CHV = 4; GO(768); }
    else return -2;
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// ---------------------------------------------------------------------------------------------------------------
// v2: no global load or store inside the per-entry loop.  Per tile every lane has its next NQ ids in registers (loaded
// at the end of the previous tile, so the latency hides behind the tile load), runs NQ predicated slots, then stores its
// results and fetches the next ids.  More than NQ entries of one lane in one tile: extra rounds of the same wave.
// ---------------------------------------------------------------------------------------------------------------
constexpr int NQ = 8;

template <int CH>
__device__ __forceinline__ float dot_row(const float (&x)[RK], const float* row) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int c0 = 0; c0 < RK / 4; c0 += CH) {
        float4 v[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) v[c] = *reinterpret_cast<const float4*>(row + 4 * (c0 + c));
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            a0 = fmaf(x[4 * (c0 + c)], v[c].x, a0);
            a1 = fmaf(x[4 * (c0 + c) + 1], v[c].y, a1);
            a2 = fmaf(x[4 * (c0 + c) + 2], v[c].z, a2);
            a3 = fmaf(x[4 * (c0 + c) + 3], v[c].w, a3);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return (a0 + a1) + (a2 + a3);
}

__device__ __forceinline__ void load_ids(int (&idq)[NQ], const int32_t* ids, int cur, int len) {
    // ids + cur is only 4-byte aligned; the buffer is padded by NQ entries so the reads stay inside it
    typedef int v4i __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
    for (int j = 0; j < NQ; j += 4) {
        const v4i t = *reinterpret_cast<const v4i*>(ids + cur + j);
        idq[j] = t[0]; idq[j + 1] = t[1]; idq[j + 2] = t[2]; idq[j + 3] = t[3];
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) idq[j] = (cur + j < len) ? idq[j] : 0x7fffffff;
}

template <int THREADS, int CH, int LD>
__global__ __launch_bounds__(THREADS) void k_scores_lds2(const float* __restrict__ U, const float* __restrict__ V,
                                                         const int32_t* __restrict__ Rs, float* __restrict__ sp, int m, int n,
                                                         int S, int T, int mode) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int tid = threadIdx.x;
    for (int64_t ub = (int64_t)blockIdx.x * THREADS; ub < m; ub += (int64_t)gridDim.x * THREADS) {
        const int64_t u = ub + tid;
        const bool live = u < m;
        float x[RK];
        {
            const float4* up = reinterpret_cast<const float4*>(U + (live ? u : 0) * RK);
#pragma unroll
            for (int c = 0; c < RK / 4; ++c) {
                const float4 v = up[c];
                x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
            }
        }
        const int32_t* ids = Rs + (live ? u : 0) * (int64_t)S;
        float* out = sp + (live ? u : 0) * (int64_t)S;
        const int len = live ? S : 0;
        float chk = 0.f;
        int cur = 0;
        int idq[NQ];
        load_ids(idq, ids, cur, len);
        for (int lo = 0; lo < n; lo += T) {
            const int hi = (lo + T < n) ? lo + T : n;
            __syncthreads();   // the previous tile has been consumed
            {
                const float4* src = reinterpret_cast<const float4*>(V + (int64_t)lo * RK);
                const int total = (hi - lo) * (RK / 4);
                for (int i0 = tid; i0 < total; i0 += THREADS * LD) {
                    float4 v[LD];
#pragma unroll
                    for (int t = 0; t < LD; ++t) {
                        const int i = i0 + t * THREADS;
                        v[t] = src[i < total ? i : total - 1];
                    }
#pragma unroll
                    for (int t = 0; t < LD; ++t) {
                        const int i = i0 + t * THREADS;
                        if (i < total) *reinterpret_cast<float4*>(tile + (i >> 5) * STRIDE + 4 * (i & 31)) = v[t];
                    }
                }
            }
            __syncthreads();
            bool again;
            do {
                float res[NQ];
                int c = 0;
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const bool act = idq[j] < hi;
                    if (__builtin_amdgcn_ballot_w64(act) == 0) break;   // ids ascend: no later slot is active either
                    if (act) {
                        res[j] = dot_row<CH>(x, tile + (idq[j] - lo) * STRIDE);
                        c = j + 1;
                    }
                }
#pragma unroll
                for (int j = 0; j < NQ; ++j)
                    if (j < c) { if (mode & 1) chk += res[j]; else out[cur + j] = res[j]; }
                cur += c;
                again = __builtin_amdgcn_ballot_w64(c == NQ) != 0;
                if (mode & 2) {   // synthetic ids: no id loads
#pragma unroll
                    for (int j = 0; j < NQ; ++j) idq[j] = (cur + j < len) ? (cur + j) * 97 + (tid & 63) : 0x7fffffff;
                } else {
                    load_ids(idq, ids, cur, len);
                }
            } while (again);
        }
        if ((mode & 1) && live) out[0] = chk;
    }
}

extern "C" int proto_scores_lds2(const float* U, const float* V, const int32_t* Rs, float* sp, int m, int n, int S, int T,
                                 int threads, int blocks, int mode, void* stream) {
    const size_t lds = (size_t)T * STRIDE * sizeof(float);
    hipError_t e;
#define GO2(TH, CHV, LDV)                                                                                                       \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scores_lds2<TH, CHV, LDV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)lds);                                                                                          \
    if (e != hipSuccess) return -1;                                                                                             \
    hipLaunchKernelGGL((k_scores_lds2<TH, CHV, LDV>), dim3(blocks), dim3(TH), lds, (hipStream_t)stream, U, V, Rs, sp, m, n, S, T, mode)
    if (threads == 512) { GO2(512, 8, 8); }
    else if (threads == 256) { GO2(256, 8, 8); }
    else return -2;
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
