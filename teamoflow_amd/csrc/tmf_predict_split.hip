// Fused predict + stable top-k for fp32 factors on the bf16 matrix cores: every fp32 value is split EXACTLY into three bf16
// planes (x = x1 + x2 + x3: 8 + 8 + 8 significant bits, round-to-nearest residuals), and <u, v> is the sum of the six
// bf16 x bf16 products whose weight is not below 2^-24 of the leading one,
//     u3 v1 + u1 v3 + u2 v2 + u2 v1 + u1 v2 + u1 v1          (small terms first inside every k-step),
// each of them exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.  The dropped products (u2 v3, u3 v2, u3 v3) are
// <= 2^-24 relative per term - one fp32 rounding - so the result is an fp32-accurate dot product: measured against an fp64
// reference the values are as close as (big terms first) or closer than (small terms first) those of the fp32 MFMA kernel
// (tools/split_accuracy.py: max |err| / max |ref| 0.8e-7 ... 3.0e-7 against 2.4e-7 ... 3.1e-7).  Six bf16 MFMAs cost 6 / 16
// of one fp32 MFMA of the same shape on gfx950 (2.5 PFLOP/s against 157 TFLOP/s dense), so the fp32-equivalent ceiling of
// this kernel is 2.7 x the fp32 MFMA peak.
//
// Structure: that of k_predict_topk_bf16 (tmf_predict.hip).  A workgroup owns 32 users per wave; the users' rows are split
// in registers at load time and stay there as A fragments of the three planes (3 x K/16 x 4 VGPRs per lane: 96 at r = 128).
// The item table is split once per call into a caller-provided workspace ([3][n_pad][K_PAD] bf16, zero padded), and item
// tiles of 128 (64 at r > 64) stream through an LDS ring in k-chunks of 32 (64) by LDS-DMA - no staging registers, a counted
// vmcnt and a raw barrier per chunk; the unpadded image is swizzled on the source side so that the ds_read_b128 operand
// reads are conflict-free.  Per (k-step, 32-column block) a wave reads three B fragments and issues six MFMAs: half an LDS
// read per MFMA, where the one-plane bf16 kernel needs one.  Candidates reach the rows' sorted lists as in the bf16 kernel
// (per-row threshold in registers, pending buffer, one lane per row merges) but without LDS atomics, and a warm-up pass over
// the first 1/64 of the catalog gives every row a threshold to start from.
//
// The HALF2 form of the same kernel (tmf_predict_topk_half2_f32) takes two fp16 planes per factor and three products
// (h2 v1 + h1 v2 + h1 v1 on v_mfma_f32_32x32x16_f16): fp16 has 11 significant bits, so two planes carry 22 bits of a factor once
// a power-of-two scale has put them into fp16's normal range - one scale per user row (found while the row is loaded), one for
// the item table (a max reduction before the split).  A row's scores all carry the same factor, which the ranking ignores
// and the epilogue takes out of the values.  Half the matrix-core work, 64 instead of 96 A registers at r = 128 (r <= 256
// fits), values against fp64 at the fp32 MFMA kernel's error: 19.9 ms = 338 TF fp32-equivalent at r = 128 (item 7 of the notes).
//
// Round 5: r <= 256 on the three bf16 planes too (192 A registers per lane, 64-item tiles, some spills outside the chunk loop, eight
// waves per workgroup at two per SIMD): 262144 x 100000, k = 10: 78.5 ms = 170.9 TF fp32-equivalent against 121.8 ms = 110.2 TF for the
// fp32 MFMA kernel, same error against fp64 (7.0e-7), lists identical (tools/time_predict_r256.py).
// Measured (262144 x 100000, k = 10, random factors; tools/split_time.py, profiles/r03_predict_split.txt): r = 128: 29.9 ms =
// 224.7 TF fp32-equivalent = 1.35 PFLOP/s of bf16 MFMA work (fp32 MFMA kernel: 57 ms, 118 TF); r = 64: 195 TF; r = 32: 149 TF.
// Timing-only variants of the r = 128 run: without the candidate handling 27.3 ms, also without the chunk barrier 28.4 (no
// gain: the barrier is not what the loop waits for), without the item loads 24.3 ms - the MFMA + operand-read loop itself runs
// at ~75 % of the matrix pipe, which is where the guide's best bf16 GEMM templates sit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/tmf.h"
#include "tmf_common.h"

namespace tmf {

typedef float f32x16_s __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_s __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_s __attribute__((ext_vector_type(8)));
typedef unsigned int raw16_s __attribute__((ext_vector_type(4)));   // one 16-byte MFMA operand of either element type
typedef float f32x4_s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_s;
typedef __attribute__((address_space(1))) const void gbl_void_s;

constexpr int SMAXK = 40 /* bf16 planes (round 5: two 4-wave workgroups' lists fit the LDS beside 8 pending entries per row up to 40) */, SMAXK_HALF2 = 32, SMAXR = 256, SMAXR_HALF2 = 256;
// Two shapes of workgroup.  WAVES = 4 (128 users, two workgroups per CU): the workgroups of a CU drift
// apart, so one multiplies while the other stands at its chunk barrier or files candidates - the two waves of a SIMD no longer
// stall together.  WAVES = 8 (256 users, 3-slot ring, one workgroup per CU): half the item-table bytes per flop; it was the k > 22
// shape until round 5 (two sets of lists did not fit the LDS beside rings of 64-wide k-chunks) and still is for the fp16 planes;
// the bf16 planes now run k > 22 on 4-wave workgroups with 32-wide k-chunks (launch_predict_topk_split: +5 ... +26 %).
// Pending entries per row: what the LDS leaves next to the lists.
__host__ __device__ constexpr int split_cap(int waves, int k) { return waves == 4 ? (k <= 12 ? 16 : 8) : (k <= 16 ? 16 : 8); }   // never below 8: an overflowing tile is re-offered in groups of 8 columns
// LDS ring slots: the two-plane chunks of the fp16 form are small enough for three of them beside two workgroups' lists
__host__ __device__ constexpr int split_ring(int waves, bool half2) { return (waves == 4 && !half2) ? 2 : 3; }
constexpr int kSplitRowsPad = 128;   // the item planes are padded to a multiple of this many rows (a multiple of every tile width)

// max of two accumulator values without the canonicalising v_max(x, x) hipcc puts in front of fmaxf (NaNs do not matter here:
// a NaN score never compares above a threshold)
__device__ __forceinline__ float max_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ bool before_s(float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia < ib); }

// x -> (x1, x2, x3) with x1 + x2 + x3 == x exactly for every finite x whose residuals stay normal.  A value that rounds to
// +-inf in bf16 is truncated instead; non-finite inputs keep their class in x1 and contribute nothing through x2, x3.
__device__ __forceinline__ void split3(float x, __bf16& x1, __bf16& x2, __bf16& x3) {
    __bf16 h = (__bf16)x;
    if (__builtin_isinf((float)h) && !__builtin_isinf(x)) h = __builtin_bit_cast(__bf16, (uint16_t)(__float_as_uint(x) >> 16));
    x1 = h;
    if (!__builtin_isfinite(x)) { x2 = (__bf16)0.0f; x3 = (__bf16)0.0f; return; }
    const float r1 = x - (float)h;
    x2 = (__bf16)r1;
    x3 = (__bf16)(r1 - (float)x2);
}

// Two-plane fp16 split of x * scale (scale a power of two that puts the largest magnitude of the row / table into
// [2^14, 2^15)): x = (h1 + h2) / scale to 22 bits, and to 2^-25 / scale absolutely where h2 is subnormal.  Non-finite values
// keep their class in h1.
__device__ __forceinline__ void split2h(float x, float scale, _Float16& h1, _Float16& h2) {
    const float xs = x * scale;
    h1 = (_Float16)xs;
    h2 = __builtin_isfinite(xs) && __builtin_isfinite((float)h1) ? (_Float16)(xs - (float)h1) : (_Float16)0.0f;
}

// The power of two that scales a largest magnitude mx into [2^14, 2^15) (1 for mx == 0 or a non-finite mx)
__device__ __forceinline__ float half_scale_for(float mx) {
    if (!(mx > 0.f) || !__builtin_isfinite(mx)) return 1.f;
    int e;
    (void)frexpf(mx, &e);            // mx = f 2^e, f in [0.5, 1)
    e = 15 - e;
    e = e > 126 ? 126 : (e < -126 ? -126 : e);
    return ldexpf(1.f, e);
}

template <bool HALF2>
__device__ __forceinline__ f32x16_s mfma_planes(const raw16_s a, const raw16_s b, const f32x16_s c) {
    if constexpr (HALF2)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_s, a), __builtin_bit_cast(f16x8_s, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_s, a), __builtin_bit_cast(bf16x8_s, b), c, 0, 0, 0);
}

// The k-th largest of the 32 values a half-wave holds (one per lane), returned to every lane of the half.
__device__ __forceinline__ float half_kth_largest(float x, int k, int l31, int h) {
    float res = -INFINITY;
    for (int i = 0; i < k; ++i) {
        float mx = x;
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        res = mx;
        const uint64_t b = __ballot(x == mx);
        const uint32_t mine = h ? (uint32_t)(b >> 32) : (uint32_t)b;
        if (mine != 0u && l31 == __ffs(mine) - 1) x = -INFINITY;   // one holder of the maximum leaves
    }
    return res;
}

// Rows [0, rows_pad) x columns [0, ldp) of the three planes; everything outside [0, rows) x [0, r) is zero.
__global__ __launch_bounds__(256) void k_split3_rows(const float* __restrict__ X, int64_t rows, int r, int64_t ld,
                                                     __bf16* __restrict__ out, int64_t rows_pad, int ldp) {
    const int groups = ldp / 8;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows_pad * groups) return;
    const int64_t row = t / groups;
    const int c0 = (int)(t % groups) * 8;
    bf16x8_s p1, p2, p3;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = (row < rows && c0 + e < r) ? X[row * ld + c0 + e] : 0.f;
        __bf16 a, b, c;
        split3(x, a, b, c);
        p1[e] = a; p2[e] = b; p3[e] = c;
    }
    const int64_t plane = rows_pad * ldp, o = row * ldp + c0;
    *reinterpret_cast<bf16x8_s*>(out + o) = p1;
    *reinterpret_cast<bf16x8_s*>(out + plane + o) = p2;
    *reinterpret_cast<bf16x8_s*>(out + 2 * plane + o) = p3;
}

// max |x| over rows [0, rows) x columns [0, r) -> *out (as the bits of a non-negative float: atomicMax on the integer)
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ X, int64_t rows, int r, int64_t ld, unsigned* __restrict__ out) {
    float mx = 0.f;
    const int64_t total = rows * r;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const float v = fabsf(X[(t / r) * ld + t % r]);
        if (v > mx && __builtin_isfinite(v)) mx = v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(out, __float_as_uint(mx));
}

// The two fp16 planes of the item table under the table-wide scale (from *absmax): [2][rows_pad][ldp], zero padded; the
// scale itself goes to *scale_out for the epilogue of the top-k kernel.
__global__ __launch_bounds__(256) void k_split2_rows(const float* __restrict__ X, int64_t rows, int r, int64_t ld,
                                                     _Float16* __restrict__ out, int64_t rows_pad, int ldp,
                                                     const unsigned* __restrict__ absmax, float* __restrict__ scale_out) {
    const float scale = half_scale_for(__uint_as_float(*absmax));
    if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
    const int groups = ldp / 8;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows_pad * groups) return;
    const int64_t row = t / groups;
    const int c0 = (int)(t % groups) * 8;
    f16x8_s p1, p2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = (row < rows && c0 + e < r) ? X[row * ld + c0 + e] : 0.f;
        _Float16 a, b;
        split2h(x, scale, a, b);
        p1[e] = a; p2[e] = b;
    }
    const int64_t plane = rows_pad * ldp, o = row * ldp + c0;
    *reinterpret_cast<f16x8_s*>(out + o) = p1;
    *reinterpret_cast<f16x8_s*>(out + plane + o) = p2;
}

// NJ column blocks of 32 items per tile, KS k-steps of 16 per chunk, NCH chunks per tile: K_PAD = 16 KS NCH.
// (NJ, KS) = (4, 2): 128-item tiles, 64 accumulator registers - narrow tables, where the A fragments are few;
//            (2, 4):  64-item tiles, 32 accumulator registers - leaves room for the 96 A registers of r = 128 without spills.
// Either way a chunk is eight 1-KB LDS-DMA pieces per plane and 48 MFMAs per wave between two barriers.
// HALF2: two fp16 planes per factor and three products (h2 v1 + h1 v2 + h1 v1) instead of three bf16 planes and six: every
// user row is scaled by its own power of two and the item table by one (so that the planes sit in fp16's normal range); a
// row's scores all carry the same factor, which the ranking ignores and the epilogue takes out of the values.
template <int NJ, int KS, int NCH, int WAVES, bool HALF2>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 2 : 1) void k_predict_topk_split(const float* __restrict__ A, const uint16_t* __restrict__ Bp,
                                                               int64_t m, int64_t n, int64_t n_pad, int K, int64_t lda, int k,
                                                               int clamp, int32_t* __restrict__ out_idx,
                                                               float* __restrict__ out_val, const float* __restrict__ item_scale) {
    constexpr int NP = HALF2 ? 2 : 3;   // planes
    constexpr int LDP = 16 * KS * NCH, SBN = 32 * NJ, SROW = 32 * KS /* bytes, unpadded: the image is written by LDS-DMA */, SPLANE = SBN * SROW, SSLOT = NP * SPLANE;
    constexpr int NK = KS * NCH;   // k-steps per plane
    constexpr int SBM = 32 * WAVES, THREADS = 64 * WAVES, SRING = split_ring(WAVES, HALF2);
    constexpr int LPW = NJ * KS / WAVES;   // LDS-DMA loads per plane, chunk and wave (a plane of a chunk is NJ KS pieces of 1 KB)
    static_assert(LPW * WAVES == NJ * KS, "the waves share the pieces of a plane evenly");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* Bs = smem_raw;                                               // [SRING slots][3 planes][SBN][SROW] bytes
    float* tau = reinterpret_cast<float*>(Bs + SRING * SSLOT);         // [SBM]
    const int SCAP = split_cap(WAVES, k);
    int* cnt = reinterpret_cast<int*>(tau + SBM);                      // [SBM]
    float* inv_scale = reinterpret_cast<float*>(cnt + SBM);            // [SBM] what the epilogue multiplies a row's values by
    float* pend_v = inv_scale + SBM;                                   // [SCAP][SBM]
    int* pend_i = reinterpret_cast<int*>(pend_v + SCAP * SBM);
    float* list_v = reinterpret_cast<float*>(pend_i + SCAP * SBM);     // [k][SBM]
    int* list_i = reinterpret_cast<int*>(list_v + (size_t)k * SBM);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int64_t row0 = (int64_t)blockIdx.x * SBM;

    // A fragments of the planes: a[p][kk] = plane p of U[row0 + 32 wave + l31][16 kk + 8 h .. + 8)
    raw16_s a[NP][NK];
    {
        const int64_t r = row0 + 32 * wave + l31;
        const float* p = A + (r < m ? r : 0) * lda;
        auto load8 = [&](int kk, float (&x)[8]) {   // this lane's eight factors of k-step kk (zeros past K and past the last row)
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = 0.f;
            const int k0 = 16 * kk + 8 * h;
            if (r < m && k0 < K) {
                if (k0 + 7 < K) {
                    const f32x4_s lo = *reinterpret_cast<const f32x4_s*>(p + k0), hi = *reinterpret_cast<const f32x4_s*>(p + k0 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { x[e] = lo[e]; x[4 + e] = hi[e]; }
                } else {
                    for (int e = 0; e < 8; ++e) if (k0 + e < K) x[e] = p[k0 + e];
                }
            }
        };
        float sc = 1.f;
        if constexpr (HALF2) {
            float mx = 0.f;   // the row's largest finite magnitude (the row is read twice: once for this, once for the planes)
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
                float x[8];
                load8(kk, x);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = fabsf(x[e]);
                    mx = (v > mx && __builtin_isfinite(v)) ? v : mx;
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));   // the other half of the k-slices
            sc = half_scale_for(mx);
            if (h == 0) inv_scale[32 * wave + l31] = (1.f / sc) * (1.f / *item_scale);   // two exact powers of two; their PRODUCT sc * item_scale can overflow (tiny tables), the reciprocals' only when the scores themselves do
        } else {
            if (h == 0) inv_scale[32 * wave + l31] = 1.f;
        }
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            float x[8];
            load8(kk, x);
            if constexpr (HALF2) {
                f16x8_s p1, p2;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    _Float16 u, v;
                    split2h(x[e], sc, u, v);
                    p1[e] = u; p2[e] = v;
                }
                a[0][kk] = __builtin_bit_cast(raw16_s, p1);
                a[1][kk] = __builtin_bit_cast(raw16_s, p2);
            } else {
                bf16x8_s p1, p2, p3;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    __bf16 u, v, w;
                    split3(x[e], u, v, w);
                    p1[e] = u; p2[e] = v; p3[e] = w;
                }
                a[0][kk] = __builtin_bit_cast(raw16_s, p1);
                a[1][kk] = __builtin_bit_cast(raw16_s, p2);
                a[2][kk] = __builtin_bit_cast(raw16_s, p3);
            }
        }
    }
    for (int t = tid; t < SBM; t += THREADS) {
        tau[t] = (row0 + t < m) ? -INFINITY : INFINITY;
        cnt[t] = 0;
        for (int j = 0; j < k; ++j) { list_v[j * SBM + t] = -INFINITY; list_i[j * SBM + t] = 0x7fffffff; }
    }

    // Staging by LDS-DMA (global_load_lds_dwordx4: no staging registers, two chunks in flight): one wave-instruction writes
    // 1 KB of LDS lane by lane, so wave w fills rows [RW w, RW (w + 1)) of each plane - RW = 64 / SR rows of SR = 2 KS
    // 16-byte pieces, unpadded.  Bank conflicts are avoided by a swizzle on the SOURCE side: LDS piece q of row i holds the
    // row's piece q ^ swz(i), swz(i) = (i / (16 / SR)) % SR, which spreads the pieces that 16 consecutive rows read together
    // over all 16 bank slots; the operand reads apply the same XOR.
    constexpr int SR = 2 * KS, RW = 64 / SR;
    static_assert(RW * NJ * KS == SBN, "NJ KS wave-instructions fill one plane of a tile");
    int s_off[LPW];   // element offset of this lane's source piece in row group i of the wave (rows RW LPW wave + RW i + lane / SR)
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int row = RW * (LPW * wave + i) + lane / SR;
        s_off[i] = row * LDP + 8 * ((lane % SR) ^ ((row / (16 / SR)) % SR));
    }
    const int ntiles = (int)((n + SBN - 1) / SBN);
    const int nchunks = ntiles * NCH;
    // Warm-up: the first `warm` tiles are multiplied twice.  The first time only a running maximum per (row, lane) is kept -
    // 32 maxima of disjoint item groups per row, so their k-th largest is the score of a k-th distinct item and a valid lower
    // bound of the row's final k-th value.  The scan then restarts at tile 0 with every threshold just below that bound:
    // the rows skip the phase in which nearly every score is a candidate (half of all k (1 + ln(n / k)) insertions of a row
    // fall into its first ~5 tiles) for warm / ntiles (<= 1/64) more MFMA work.
#ifndef TMF_SPLIT_WARM_MAXK
#define TMF_SPLIT_WARM_MAXK 32   /* round 4: 24 -> 32 (k = 25: 130.7 -> 134.6 TF, k = 28: 119.6 -> 123.6, k = 32 unchanged; same box) */
#endif
    const int warm = (k <= TMF_SPLIT_WARM_MAXK && ntiles >= 256) ? (ntiles / 64 < 128 ? ntiles / 64 : 128) : 0;
    const int warm_chunks = warm * NCH;
    const int64_t plane = n_pad * LDP;
    auto g_issue = [&](int g, int slot) {   // chunk g -> ring slot `slot` (three LDS-DMA loads per wave)
        int gg = g < warm_chunks ? g : g - warm_chunks;
        gg = gg < nchunks ? gg : nchunks - 1;   // the read-ahead past the last chunk re-reads it
        const uint16_t* s = Bp + (int64_t)(gg / NCH) * (SBN * LDP) + 16 * KS * (gg % NCH);
        char* dst = Bs + slot * SSLOT + wave * (1024 * LPW);
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < LPW; ++i)
                __builtin_amdgcn_global_load_lds((gbl_void_s*)(s + p * plane + s_off[i]), (lds_void_s*)(dst + p * SPLANE + i * 1024), 16, 0, 0);
    };
    // Chunk g + 2 is issued at the top of iteration g into the slot read in iteration g - 1 (every wave passed the barrier
    // that ended it).  At the bottom a counted wait leaves only those three loads in flight - chunk g + 1 has landed - and
    // the raw barrier publishes it to the readers of iteration g + 1 (a __syncthreads() would drain the DMA queue).
    auto ring_step = [&]() {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SRING - 2) * NP * LPW) : "memory");
        __builtin_amdgcn_s_barrier();
    };
    __syncthreads();   // lists and thresholds initialised; no DMA in flight yet
    g_issue(0, 0);
    if (SRING == 3) g_issue(1, 1);
    ring_step();
    const int rd_swz = (l31 / (16 / SR)) % SR;   // swz(32 j + l31) for every column block j

    f32x16_s acc[NJ];
    float tq[16];  // register copy of this lane's 16 row thresholds
    // accumulator register q of this lane belongs to row rbase + qoff(q) of the workgroup: per-lane base pointers + constants,
    // so that every per-row LDS access is one base register and an immediate offset (16 address registers per array otherwise)
    const int rbase = 32 * wave + 4 * h;
    auto qoff = [](int q) { return (q & 3) + 8 * (q >> 2); };
    float* const tau_l = tau + rbase;
    int* const cnt_l = cnt + rbase;
    float* const pend_v_l = pend_v + rbase;
    int* const pend_i_l = pend_i + rbase;
    auto load_tau = [&]() {
#pragma unroll
        for (int q = 0; q < 16; ++q) tq[q] = tau_l[qoff(q)];
    };
    if (warm > 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) tq[q] = -INFINITY;   // running maxima during the warm-up
    } else {
        load_tau();
    }
    // Which of this wave's sixteen accumulator registers (row pairs) hold a score above its row's threshold in some lane: two
    // three-input maxima and one compare per register, the verdicts collected in a scalar mask (the lanes that pass are found
    // again by offer(), which compares every score of such a register anyway).
    const float floor_v = clamp ? 0.f : -INFINITY;
    auto prefilter = [&]() -> unsigned {
        unsigned qmask = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float mx;
            if constexpr (NJ == 4) mx = max3_raw(max3_raw(acc[0][q], acc[1][q], acc[2][q]), acc[3][q], floor_v);
            else mx = max3_raw(acc[0][q], acc[1][q], floor_v);
            qmask |= (__ballot(mx > tq[q]) != 0) ? (1u << q) : 0u;
        }
        return qmask;
    };
    static_assert(NJ == 2 || NJ == 4, "prefilter is written for two or four column blocks");
    const int n32 = (int)n;
    // Appending candidates takes no LDS atomic (hipcc drains the LDS-DMA queue, vmcnt(0), before every LDS atomic): the lanes
    // of a half-wave that hold candidates of one row take consecutive slots by ballot + popcount behind the row's count, which
    // they read once per row and write back (all the same value) at the end.  Rows are private to a wave.
    const uint32_t lt_mask = (1u << l31) - 1u;
    auto offer = [&](int col0, int group, unsigned pass) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (!((pass >> q) & 1u)) continue;
            const float t = tau_l[qoff(q)];
            int c0 = cnt_l[qoff(q)];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int lc = 32 * j + l31;
                float v = acc[j][q];
                if (clamp) v = (v > 0.f) ? v : 0.f;
                const bool in_group = (group < 0) || ((lc >> 3) == group);
                const bool cand = in_group && (col0 + lc < n32) && v > t;
                const uint64_t b = __ballot(cand);
                const uint32_t mh = h ? (uint32_t)(b >> 32) : (uint32_t)b;
                const int pos = c0 + __popc(mh & lt_mask);
                if (cand && pos < SCAP) { pend_v_l[pos * SBM + qoff(q)] = v; pend_i_l[pos * SBM + qoff(q)] = col0 + lc; }
                c0 += __popc(mh);
            }
            cnt_l[qoff(q)] = c0;
        }
    };
    auto merge_wave = [&]() {
        if (h == 0) {
            const int row = 32 * wave + l31;
            const int c = cnt[row] < SCAP ? cnt[row] : SCAP;
            for (int p = 0; p < c; ++p) {
                const float v = pend_v[p * SBM + row];
                const int ix = pend_i[p * SBM + row];
                int j = k - 1;
                if (before_s(v, ix, list_v[j * SBM + row], list_i[j * SBM + row])) {
                    while (j > 0 && before_s(v, ix, list_v[(j - 1) * SBM + row], list_i[(j - 1) * SBM + row])) {
                        list_v[j * SBM + row] = list_v[(j - 1) * SBM + row];
                        list_i[j * SBM + row] = list_i[(j - 1) * SBM + row];
                        --j;
                    }
                    list_v[j * SBM + row] = v;
                    list_i[j * SBM + row] = ix;
                }
            }
            cnt[row] = 0;
            if (row0 + row < m) tau[row] = fmaxf(tau[row], list_v[(k - 1) * SBM + row]);   // never below the warm-up bound
        }
    };

    int g = 0, slot = 0;   // chunk counter and its ring slot (g % SRING)
    int c_prev = 0;
    for (int vt = 0; vt < warm + ntiles; ++vt) {
        const int tile = vt < warm ? vt : vt - warm;
        if (warm > 0 && vt == warm) {   // thresholds from the warm-up maxima; the scan restarts at tile 0
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float t0 = half_kth_largest(tq[q], k, l31, h);
                // strictly below the bound; a row with k or more +inf scores in the warm-up tiles has t0 = +inf (inf - inf = NaN would
                // reject every candidate of the scan and leave the row's list empty): its bound is the largest finite value
                if (l31 == 0 && row0 + rbase + qoff(q) < m)
                    tau_l[qoff(q)] = __builtin_isfinite(t0) ? t0 - fabsf(t0) * 1e-6f - 1e-30f : (t0 > 0.f ? 3.0e38f : -INFINITY);
            }
            wave_lds_sync();
            load_tau();
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c, ++g) {
            g_issue(g + SRING - 1, slot == 0 ? SRING - 1 : slot - 1);
            const char* bs = Bs + slot * SSLOT + l31 * SROW;
            int po[KS];   // byte offset of this lane's piece of k-step ks inside its row
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) po[ks] = 16 * ((2 * ks + h) ^ rd_swz);
            raw16_s bq[2][NP];  // the B planes of one (k-step, column block), one step ahead of their MFMAs
#pragma unroll
            for (int p = 0; p < NP; ++p) bq[0][p] = *reinterpret_cast<const raw16_s*>(bs + p * SPLANE + po[0]);
#pragma unroll
            for (int st = 0; st < KS * NJ; ++st) {   // st = NJ ks + j
                const int ks = st / NJ, j = st % NJ;
                const raw16_s* b = bq[st & 1];
                const int kk = KS * c + ks;
                // The operands of step st + 1 are requested BEHIND the first MFMA of step st: the only wait hipcc places is a
                // full lgkmcnt(0) in front of a step's first MFMA, which then covers nothing newer than the operands it needs
                // (requested five MFMAs = 160 cycles earlier); requested in front of step st they sat behind that wait.
                acc[j] = mfma_planes<HALF2>(a[NP - 1][kk], b[0], acc[j]);   // the smallest product first
                __builtin_amdgcn_sched_barrier(0);
                if (st + 1 < KS * NJ) {
                    const int ks1 = (st + 1) / NJ, j1 = (st + 1) % NJ;
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        bq[(st + 1) & 1][p] = *reinterpret_cast<const raw16_s*>(bs + p * SPLANE + 32 * j1 * SROW + po[ks1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (HALF2) {
                    acc[j] = mfma_planes<HALF2>(a[0][kk], b[1], acc[j]);
                    acc[j] = mfma_planes<HALF2>(a[0][kk], b[0], acc[j]);
                } else {
                    acc[j] = mfma_planes<HALF2>(a[0][kk], b[2], acc[j]);
                    acc[j] = mfma_planes<HALF2>(a[1][kk], b[1], acc[j]);
                    acc[j] = mfma_planes<HALF2>(a[1][kk], b[0], acc[j]);
                    acc[j] = mfma_planes<HALF2>(a[0][kk], b[1], acc[j]);
                    acc[j] = mfma_planes<HALF2>(a[0][kk], b[0], acc[j]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            ring_step();
            slot = (slot == SRING - 1) ? 0 : slot + 1;
        }
        if (vt < warm) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float mx = acc[0][q];
#pragma unroll
                for (int j = 1; j < NJ; ++j) mx = max_raw(mx, acc[j][q]);
                if (clamp) mx = max_raw(mx, 0.f);
                tq[q] = max_raw(tq[q], mx);
            }
            continue;
        }
        const int col0 = tile * SBN;
        const unsigned pass = prefilter();
        if (pass != 0u) {  // (wave-uniform) candidates are appended; lists and thresholds catch up when a buffer is half full
            offer(col0, -1, pass);
            const int my_row = 32 * wave + l31;
            const int c_now = (h == 0) ? cnt[my_row] : 0;
            if (__any(c_now > SCAP)) {  // overflow: keep the older appends, re-offer this tile in 16 groups of 8 columns
                if (h == 0) cnt[my_row] = c_prev;
                merge_wave();
                for (int grp = 0; grp < SBN / 8; ++grp) {
                    offer(col0, grp, pass);
                    merge_wave();
                }
                c_prev = 0;
                load_tau();
            } else if (__any(c_now > SCAP / 2)) {
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
    if (tid < SBM && row0 + tid < m) {
        for (int j = 0; j < k; ++j) {
            out_idx[(row0 + tid) * k + j] = list_i[j * SBM + tid];
            if (out_val) out_val[(row0 + tid) * k + j] = list_v[j * SBM + tid] * inv_scale[tid];
        }
    }
}

static int64_t split_rows_pad(int64_t n) { return (n + kSplitRowsPad - 1) / kSplitRowsPad * kSplitRowsPad; }
static int split_ldp(int r) { return r <= 32 ? 32 : r <= 64 ? 64 : r <= 128 ? 128 : 256; }

#ifndef TMF_SPLIT_LDS_PAD
#define TMF_SPLIT_LDS_PAD 0   /* timing-only: LDS bytes asked for beyond what the kernel uses (45000: one 4-wave workgroup per CU) */
#endif
template <int NJ, int KS, int NCH, int WAVES, bool HALF2>
static int launch_predict_topk_split_w(const float* A, const uint16_t* Bp, int64_t m, int64_t n, int64_t n_pad, int K, int64_t lda,
                                       int k, int clamp, int32_t* out_idx, float* out_val, const float* item_scale, hipStream_t stream) {
    constexpr int SBM = 32 * WAVES, SRING = split_ring(WAVES, HALF2), NP = HALF2 ? 2 : 3;
    const size_t lds = (size_t)SRING * NP * (32 * NJ) * (32 * KS) + 3 * sizeof(float) * SBM + 8 * (size_t)split_cap(WAVES, k) * SBM +
                       8 * (size_t)k * SBM + TMF_SPLIT_LDS_PAD;
    static LdsGrant grant;  // per template instance
    if (int rc = grant_dynamic_lds(reinterpret_cast<const void*>(&k_predict_topk_split<NJ, KS, NCH, WAVES, HALF2>), lds, grant)) return rc;
    const int64_t blocks = (m + SBM - 1) / SBM;
    TMF_REQUIRE_LAUNCH(blocks, 64 * WAVES, "predict_topk_split");
    hipLaunchKernelGGL((k_predict_topk_split<NJ, KS, NCH, WAVES, HALF2>), dim3((unsigned)blocks), dim3(64 * WAVES), lds, stream, A, Bp, m,
                       n, n_pad, K, lda, k, clamp, out_idx, out_val, item_scale);
    return check_launch(HALF2 ? "tmf_predict_topk_half2_f32" : "tmf_predict_topk_split_f32");
}

static bool split_force8() {   // TMF_SPLIT_WAVES=8: the 8-wave instances for every k (A/B runs)
    static const bool forced = [] { const char* e = getenv("TMF_SPLIT_WAVES"); return e && atoi(e) == 8; }();
    return forced;
}
static int split_waves(int k) {
    if (split_force8() || k > 22) return 8;   // k <= 22: two workgroups' lists still fit beside their rings of 64-wide k-chunks (80 KB each)
    return 4;
}

template <bool HALF2>
static int launch_predict_topk_split(int ldp, int k, const float* A, const uint16_t* Bp, int64_t m, int64_t n, int64_t n_pad, int K,
                                     int64_t lda, int clamp, int32_t* out_idx, float* out_val, const float* item_scale, hipStream_t s) {
    const bool w4 = split_waves(k) == 4;
#define TMF_SPLIT_GO(NJ, KS, NCH)                                                                                                         \
    return w4 ? launch_predict_topk_split_w<NJ, KS, NCH, 4, HALF2>(A, Bp, m, n, n_pad, K, lda, k, clamp, out_idx, out_val, item_scale, s)  \
              : launch_predict_topk_split_w<NJ, KS, NCH, 8, HALF2>(A, Bp, m, n, n_pad, K, lda, k, clamp, out_idx, out_val, item_scale, s)
    if constexpr (HALF2) {   // 64 A registers at r = 128: 128-item tiles throughout, k-chunks of 32
        if (!w4 && !split_force8()) {   // k > 22: 4-wave workgroups on 64-item tiles, like the bf16 planes below
            if (ldp == 32) return launch_predict_topk_split_w<2, 2, 1, 4, HALF2>(A, Bp, m, n, n_pad, K, lda, k, clamp, out_idx, out_val, item_scale, s);
            if (ldp == 64) return launch_predict_topk_split_w<2, 2, 2, 4, HALF2>(A, Bp, m, n, n_pad, K, lda, k, clamp, out_idx, out_val, item_scale, s);
            if (ldp == 128) return launch_predict_topk_split_w<2, 2, 4, 4, HALF2>(A, Bp, m, n, n_pad, K, lda, k, clamp, out_idx, out_val, item_scale, s);
        }
        if (ldp == 32) { TMF_SPLIT_GO(4, 2, 1); }
        if (ldp == 64) { TMF_SPLIT_GO(4, 2, 2); }
        if (ldp == 128) { TMF_SPLIT_GO(4, 2, 4); }
        // r <= 256: 128 A registers, 64-item tiles - on the 4-wave workgroups for every k.  The 8-wave instance of this shape
        // (k > 22) returned wrong lists for every row (found in round 5 when the wide-table test got a k = 30 case:
        // tools/half2_diag.py; the bf16 form of the same shape and the fp16 form at r <= 128 are right); 128 users' lists fit the
        // LDS up to k = 32 beside the ring, so the shape that is tested is the one that runs.
        return launch_predict_topk_split_w<2, 4, 4, 4, HALF2>(A, Bp, m, n, n_pad, K, lda, k, clamp, out_idx, out_val, item_scale, s);
    } else {
        // k > 22 (round 5): 4-wave workgroups as well, on k-chunks of 32 (64-item tiles, 12 KB ring slots) - two workgroups' lists fit
        // the LDS up to k = 32, and the two workgroups of a CU drift apart again: in ONE 8-wave workgroup every wave that files or merges
        // candidates holds the other seven at the chunk barrier.  262144 x 100000, 8-wave -> 4-wave instances, same box:
        //   r = 128: k = 23 144 -> 164 TF, 25: 136 -> 156, 28: 124 -> 148, 32: 108 -> 133;  r = 256: k = 25 145 -> 152, 32: 128 -> 140;  r = 96, k = 32: 82 -> 103
        // TMF_SPLIT_WAVES=8 brings the 8-wave instances back (A/B runs).
        if (!w4 && !split_force8()) {
            if (ldp == 32) return launch_predict_topk_split_w<2, 2, 1, 4, HALF2>(A, Bp, m, n, n_pad, K, lda, k, clamp, out_idx, out_val, item_scale, s);
            if (ldp == 64) return launch_predict_topk_split_w<2, 2, 2, 4, HALF2>(A, Bp, m, n, n_pad, K, lda, k, clamp, out_idx, out_val, item_scale, s);
            if (ldp == 128) return launch_predict_topk_split_w<2, 2, 4, 4, HALF2>(A, Bp, m, n, n_pad, K, lda, k, clamp, out_idx, out_val, item_scale, s);
            return launch_predict_topk_split_w<2, 2, 8, 4, HALF2>(A, Bp, m, n, n_pad, K, lda, k, clamp, out_idx, out_val, item_scale, s);
        }
        if (ldp == 32) { TMF_SPLIT_GO(4, 2, 1); }
        if (ldp == 64) { TMF_SPLIT_GO(4, 2, 2); }
        if (ldp == 128) { TMF_SPLIT_GO(2, 4, 2); }
        TMF_SPLIT_GO(2, 4, 4);   // r <= 256 (round 5): 192 A registers, two waves per SIMD, 256 VGPRs each
    }
#undef TMF_SPLIT_GO
}

static int check_split_args(const char* what, const float* A, const float* B, int32_t* out_idx, int64_t m, int64_t n, int r,
                            int64_t lda, int64_t ldb, int k, void* workspace, size_t workspace_bytes, size_t need, int max_r, int max_k) {
    TMF_REQUIRE(A && B && out_idx && m > 0 && n > 0 && r > 0, "%s: bad arguments", what);
    TMF_REQUIRE(lda >= r && ldb >= r && (lda % 4 == 0) && ((uintptr_t)A % 16 == 0),
                "%s: the user table must be 16-byte aligned with ld %% 4 == 0", what);
    TMF_REQUIRE(k >= 1 && k <= n, "%s: k=%d must be in [1, n=%lld]", what, k, (long long)n);
    TMF_REQUIRE(n < ((int64_t)1 << 31), "%s: too many items", what);
    if (r > max_r || k > max_k) {
        set_error("%s: supports k <= %d and n_components <= %d (got k=%d, r=%d)", what, max_k, max_r, k, r);
        return TMF_E_UNSUPPORTED;
    }
    TMF_REQUIRE(workspace && workspace_bytes >= need && ((uintptr_t)workspace % 16 == 0),
                "%s: workspace of %zu bytes (16-byte aligned) needed, got %zu", what, need, workspace_bytes);
    return TMF_OK;
}

}  // namespace tmf

extern "C" int tmf_predict_topk_split_supported(int r, int k) {
    return r >= 1 && r <= tmf::SMAXR && k >= 1 && k <= tmf::SMAXK;
}

extern "C" size_t tmf_predict_topk_split_workspace_bytes(int64_t n, int r) {
    if (n <= 0 || r < 1 || r > tmf::SMAXR) return 0;
    return (size_t)3 * (size_t)tmf::split_rows_pad(n) * (size_t)tmf::split_ldp(r) * sizeof(uint16_t);
}

extern "C" int tmf_predict_topk_half2_supported(int r, int k) {
    return r >= 1 && r <= tmf::SMAXR_HALF2 && k >= 1 && k <= tmf::SMAXK_HALF2;
}

extern "C" size_t tmf_predict_topk_half2_workspace_bytes(int64_t n, int r) {
    if (n <= 0 || r < 1 || r > tmf::SMAXR_HALF2) return 0;
    return (size_t)2 * (size_t)tmf::split_rows_pad(n) * (size_t)tmf::split_ldp(r) * sizeof(uint16_t) + 16;   // + max |V| and the scale
}

extern "C" int tmf_predict_topk_split_f32(const float* A, const float* B, int64_t m, int64_t n, int r, int64_t lda,
                                          int64_t ldb, int k, int clamp_negatives, int32_t* out_idx, float* out_val,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    if (m == 0) return TMF_OK;
    if (int rc = tmf::check_split_args("predict_topk_split", A, B, out_idx, m, n, r, lda, ldb, k, workspace, workspace_bytes,
                                       tmf_predict_topk_split_workspace_bytes(n, r), tmf::SMAXR, tmf::SMAXK))
        return rc;
    hipStream_t s = (hipStream_t)stream;
    const int ldp = tmf::split_ldp(r);
    const int64_t n_pad = tmf::split_rows_pad(n);
    __bf16* Bp = reinterpret_cast<__bf16*>(workspace);
    {
        const int64_t threads = n_pad * (ldp / 8), blocks = (threads + 255) / 256;
        TMF_REQUIRE_LAUNCH(blocks, 256, "predict_topk_split (item planes)");
        hipLaunchKernelGGL(tmf::k_split3_rows, dim3((unsigned)blocks), dim3(256), 0, s, B, n, r, ldb, Bp, n_pad, ldp);
        if (int rc = tmf::check_launch("tmf_predict_topk_split_f32 (item planes)")) return rc;
    }
    return tmf::launch_predict_topk_split<false>(ldp, k, A, reinterpret_cast<const uint16_t*>(Bp), m, n, n_pad, r, lda, clamp_negatives,
                                                 out_idx, out_val, nullptr, s);
}

extern "C" int tmf_predict_topk_half2_f32(const float* A, const float* B, int64_t m, int64_t n, int r, int64_t lda,
                                          int64_t ldb, int k, int clamp_negatives, int32_t* out_idx, float* out_val,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    if (m == 0) return TMF_OK;
    if (int rc = tmf::check_split_args("predict_topk_half2", A, B, out_idx, m, n, r, lda, ldb, k, workspace, workspace_bytes,
                                       tmf_predict_topk_half2_workspace_bytes(n, r), tmf::SMAXR_HALF2, tmf::SMAXK_HALF2))
        return rc;
    hipStream_t s = (hipStream_t)stream;
    const int ldp = tmf::split_ldp(r);
    const int64_t n_pad = tmf::split_rows_pad(n);
    _Float16* Bp = reinterpret_cast<_Float16*>(workspace);
    char* tail = reinterpret_cast<char*>(workspace) + (size_t)2 * n_pad * ldp * sizeof(uint16_t);
    unsigned* absmax = reinterpret_cast<unsigned*>(tail);
    float* scale = reinterpret_cast<float*>(tail + 4);
    if (hipMemsetAsync(absmax, 0, 8, s) != hipSuccess) { tmf::set_error("predict_topk_half2: hipMemsetAsync failed"); return TMF_E_LAUNCH; }
    {
        const int64_t total = n * (int64_t)r;
        const unsigned blocks = (unsigned)(total / 4096 + 1 < 2048 ? total / 4096 + 1 : 2048);
        hipLaunchKernelGGL(tmf::k_absmax, dim3(blocks), dim3(256), 0, s, B, n, r, ldb, absmax);
        if (int rc = tmf::check_launch("tmf_predict_topk_half2_f32 (max)")) return rc;
    }
    {
        const int64_t threads = n_pad * (ldp / 8), blocks = (threads + 255) / 256;
        TMF_REQUIRE_LAUNCH(blocks, 256, "predict_topk_half2 (item planes)");
        hipLaunchKernelGGL(tmf::k_split2_rows, dim3((unsigned)blocks), dim3(256), 0, s, B, n, r, ldb, Bp, n_pad, ldp, absmax, scale);
        if (int rc = tmf::check_launch("tmf_predict_topk_half2_f32 (item planes)")) return rc;
    }
    return tmf::launch_predict_topk_split<true>(ldp, k, A, reinterpret_cast<const uint16_t*>(Bp), m, n, n_pad, r, lda, clamp_negatives,
                                                out_idx, out_val, scale, s);
}
