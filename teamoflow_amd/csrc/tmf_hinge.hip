// WMRB hinge step of the sliced user pass (loss_graphs.py:74-88) in O((S + P) log P) per user instead of O(P S).
//
// Inputs per user u: the sampled scores sp[u, 0..S) and, for its interactions k, the scores p_k and values a_k.
// With c1_k = fl(1 - p_k) and the threshold t_k = -c1_k the hinge term of (k, s) is active iff
//     fl(c1_k + sp_s) >= 0   <=>   sp_s >= t_k
// exactly (the sum of two floats has the sign of the exact sum; gradual underflow keeps tiny sums non-zero), so the
// active set of a positive is { s : sp_s >= t_k } and
//     cnt_k = #{ s : sp_s >= t_k }                       M_k = c (cnt_k c1_k + sum_{sp_s >= t_k} sp_s)
//     w_k   = c / (1 + M_k), delta_k = -w_k cnt_k        D[s] = sum_{k : t_k <= sp_s} w_k.
// One wave per user, positives in chunks of <= 255:
//   1. sort the chunk's thresholds (rank sort on (t, index) keys in LDS - ties keep index order, so the result is a
//      fixed function of the input);
//   2. every sample finds rho_s = #{ t <= sp_s } by an 8-step binary search over the sorted (padded) array;
//   3. samples are bucketed by rho: count and sum of sp per bucket.  rho = 0 (inactive for the whole chunk) is skipped,
//      rho = "all" is accumulated per lane in fp64 and reduced by a fixed butterfly, the buckets in between go through
//      LDS integer atomics on fixed-point values (62 - ceil(log2 S) bits) - integer addition commutes, so the sums do not depend on the
//      order the atomics land in (no float atomics anywhere: runs stay bit-reproducible);
//   4. a suffix scan over the buckets gives cnt_k and the sum of the active scores of every sorted positive (fp64, so
//      the cancellation in cnt c1 + sum sp costs nothing), then M_k, the loss, w_k and delta_k in fp32 as before;
//   5. an exclusive prefix scan of w over the sorted positives gives D[s] += prefix[rho_s].
// No factor table is touched: p comes from tmf_wmrb_scores3, sum_k delta_k V[j_k] is done by tmf_wmrb_gradu3.
// Round 5, two attempts to take the sampled scores' load off the critical path (asked for BEFORE the chain rowptr -> val / p -> sort,
// so that its latency overlaps it), both measured slower and removed (profiles/r05_all_ab_runs_raw.txt; C4 / config-5 shard ms):
//   in registers (16 more: 144 VGPRs, three waves per SIMD)     4.68 -> 5.33      5.7 -> 6.4
//   by LDS-DMA into 4 KB more LDS per wave, registers capped at 128 (52 bytes of scratch)   4.67 -> 5.17      5.6 -> 6.3
// The kernel sits exactly on the 128-register border of four waves per SIMD; anything added costs more than the latency it hides.
#include <math.h>

#include "tmf_common.h"

namespace tmf {

constexpr int HK = 256;             // slots of a chunk's sorted threshold array
constexpr int HCHUNK = HK - 1;      // entries per chunk: at least one +inf slot stays, so rho <= HK - 1
constexpr int HPAD = HK + HK / 32;  // slot i lives at i + (i >> 5): the search's reads spread over the banks
constexpr int HSPL = 8;             // samples per lane and tile (a tile = 512 samples)
constexpr int HWAVES = 1;             // one user = one wave = one workgroup: 5.0 ms at C4 against 5.85 with four users per workgroup (8: 7.1)

struct HingeLds {                   // per wave, 5440 bytes
    unsigned long long keys[HK];    // sort keys (orderable(t) << 32 | index); afterwards the fixed-point bucket sums
    float ts[HPAD];                 // sorted thresholds, padded index, +inf behind the chunk's entries
    float pw[HPAD];                 // exclusive prefix of w over the sorted entries, padded index
    unsigned int cnt[HK];           // bucket counts
    unsigned char ord[HK];          // chunk-local index of the entry at a sorted position
};

__device__ __forceinline__ unsigned int orderable(float t) {
    const unsigned int u = __float_as_uint(t);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}

__device__ __forceinline__ int pad_slot(int i) { return i + (i >> 5); }

// rho = #{ i : ts[i] <= x } for the sorted, +inf-terminated array of HK slots: 8 steps, the padded position is tracked
// directly (step' = step + step / 32; read offset step' - 2 for steps >= 32, step - 1 below)
template <int STEP>
__device__ __forceinline__ void search_step(const float* __restrict__ ts, const float (&x)[HSPL], int (&pp)[HSPL]) {
    constexpr int SP = STEP + (STEP >> 5);
    constexpr int OFF = (STEP >= 32) ? SP - 2 : STEP - 1;
    float tv[HSPL];
#pragma unroll
    for (int j = 0; j < HSPL; ++j) tv[j] = ts[pp[j] + OFF];
#pragma unroll
    for (int j = 0; j < HSPL; ++j) pp[j] += (tv[j] <= x[j]) ? SP : 0;
}

// `nfin` = finite thresholds of the chunk (they sort to the front; everything behind them is +inf).  A step whose probe position
// lies at or beyond nfin reads +inf and never moves: the search starts at the first step that can (a chunk of <= 63 thresholds - most
// users - takes 6 steps instead of 8, <= 31 five).  Wave-uniform choice.  A sample of +inf (an overflowed score) also passes the
// +inf padding, so its position depends on the steps taken: rho is clamped to nfin, i.e. such a sample is active for EVERY positive -
// what the reference computes (1 - p + inf = inf: M = inf, loss = inf, w = c / (1 + inf) = 0, so delta and D are 0).
__device__ __forceinline__ void search_tile(const float* __restrict__ ts, const float (&x)[HSPL], int (&rho)[HSPL], int nfin) {
    int pp[HSPL];
#pragma unroll
    for (int j = 0; j < HSPL; ++j) pp[j] = 0;
    if (nfin >= 128) search_step<128>(ts, x, pp);
    if (nfin >= 64) search_step<64>(ts, x, pp);
    if (nfin >= 32) search_step<32>(ts, x, pp);
    if (nfin >= 16) search_step<16>(ts, x, pp);
    search_step<8>(ts, x, pp);
    search_step<4>(ts, x, pp);
    search_step<2>(ts, x, pp);
    search_step<1>(ts, x, pp);
#pragma unroll
    for (int j = 0; j < HSPL; ++j) {
        const int r = pp[j] - pp[j] / 33;  // padded -> plain position
        rho[j] = r < nfin ? r : nfin;
    }
}

__device__ __forceinline__ void load_tile(const float* __restrict__ spu, int s0, int S, int lane, float (&x)[HSPL]) {
#pragma unroll
    for (int j = 0; j < HSPL; ++j) {
        const int s = s0 + lane + 64 * j;
        x[j] = (s < S) ? spu[s] : __builtin_nanf("");  // NaN: every comparison is false -> rho = 0, never counted
    }
}

__global__ __launch_bounds__(64 * HWAVES) void k_wmrb_hinge2(
    const int64_t* __restrict__ rowptr, const float* __restrict__ val, const float* __restrict__ p,
    const float* __restrict__ sp, int S, int fixbits, float c, int64_t n_users, float* __restrict__ delta,
    float* __restrict__ D, float* __restrict__ loss_part, const int32_t* __restrict__ order) {
    __shared__ HingeLds lds_all[HWAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t slot = (int64_t)blockIdx.x * HWAVES + wave;
    if (slot >= n_users) return;
    // `order` (optional): the users in the order the waves should take them - the heavy ones first, so that the one user with
    // twenty chunks of interactions does not start last and finish alone.  Every user is computed by itself; the order
    // changes no result.
    const int64_t u = order ? (int64_t)order[slot] : slot;
    HingeLds& L = lds_all[wave];
    const int64_t rb = rowptr[u], re = rowptr[u + 1];
    const float* spu = sp + u * (int64_t)S;
    float* Du = D + u * (int64_t)S;
    const int ntiles = (S + 64 * HSPL - 1) / (64 * HSPL);

    float lsum = 0.f;
    bool wrote = false;  // D[u, :] holds this user's values already (later chunks add)
    for (int64_t cb = rb; cb < re; cb += HCHUNK) {
        const int len = (int)((re - cb < HCHUNK) ? (re - cb) : HCHUNK);
        // ---- 1. thresholds and sort keys of the chunk (entry i = lane + 64 q) ----
        float tq[4];
        unsigned long long kq[4];
        int nfin = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = lane + 64 * q;
            float t = INFINITY;  // non-positive entries and the slots behind the chunk: never active
            if (i < len) {
                const float a = val[cb + i];
                if (a > 0.f) t = -(1.0f - p[cb + i]);
            }
            tq[q] = t;
            kq[q] = ((unsigned long long)orderable(t) << 32) | (unsigned int)i;
            nfin += __popcll(__ballot(t < INFINITY));
            if (i < len) L.keys[i] = kq[q];
        }
        if (nfin == 0) {  // no positive in this chunk
            for (int i = lane; i < len; i += 64) delta[cb + i] = 0.f;
            continue;
        }
        wave_lds_sync();
        // rank sort: position of entry i = #{ j : key_j < key_i } (keys are distinct)
        int rank[4] = {0, 0, 0, 0};
        // only the key slots the chunk populates are ranked (entry i = lane + 64 q < len): a chunk of <= 64 / <= 128 entries -
        // most users - compares one / two keys per step instead of four (wave-uniform choice of the loop)
        if (len <= 64) {
            for (int j = 0; j < len; ++j) rank[0] += (L.keys[j] < kq[0]) ? 1 : 0;   // same address in every lane: broadcast
        } else if (len <= 128) {
            for (int j = 0; j < len; ++j) {
                const unsigned long long kj = L.keys[j];
                rank[0] += (kj < kq[0]) ? 1 : 0;
                rank[1] += (kj < kq[1]) ? 1 : 0;
            }
        } else {
            for (int j = 0; j < len; ++j) {
                const unsigned long long kj = L.keys[j];
#pragma unroll
                for (int q = 0; q < 4; ++q) rank[q] += (kj < kq[q]) ? 1 : 0;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = lane + 64 * q;
            if (i < len) {
                L.ts[pad_slot(rank[q])] = tq[q];
                L.ord[rank[q]] = (unsigned char)i;
            } else if (i < HK) {
                L.ts[pad_slot(i)] = INFINITY;  // slots len .. HK - 1 (ranks cover 0 .. len - 1)
            }
        }
        wave_lds_sync();
        // ---- the sorted positions this lane owns: 4 lane .. 4 lane + 3; buckets 4 lane + 1 .. 4 lane + 4 ----
        float town[4];
        int oown[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pos = 4 * lane + q;
            town[q] = L.ts[pad_slot(pos)];
            oown[q] = (pos < len) ? (int)L.ord[pos] : -1;
            L.cnt[pos] = 0u;
            L.keys[pos] = 0ull;  // from here on: fixed-point bucket sums (every lane is past the rank loop)
        }
        // scale of the fixed-point sums: the samples that land in a middle bucket lie between the smallest and the
        // largest finite threshold, so |sp| <= B there
        const float B = fmaxf(fabsf(L.ts[0]), fabsf(L.ts[pad_slot(nfin - 1)]));
        int eB = 0;
        (void)frexpf(B, &eB);  // B < 2^eB (eB = 0 for B = 0)
        if (!(B < INFINITY)) eB = 128;
        const double to_fix = ldexp(1.0, fixbits - eB), from_fix = ldexp(1.0, eB - fixbits);  // |sp| <= B < 2^eB: |v| < 2^fixbits
        wave_lds_sync();
        // ---- 2 + 3. ranks of the samples, buckets ----
        int ctop = 0;
        double stop = 0.0;
        unsigned int rpk[4];  // ranks of the first two tiles (1024 samples), one byte each: kept for step 5
        for (int tp = 0; tp < ntiles; tp += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tile = tp + h;
                if (tile >= ntiles) break;
                float x[HSPL];
                int rho[HSPL];
                load_tile(spu, tile * 64 * HSPL, S, lane, x);
                search_tile(L.ts, x, rho, nfin);
#pragma unroll
                for (int j = 0; j < HSPL; ++j) {
                    if (rho[j] == nfin) {
                        ctop += 1;
                        stop += (double)x[j];
                    } else if (rho[j] != 0 && fabsf(x[j]) <= B) {
                        atomicAdd(&L.cnt[rho[j]], 1u);
                        atomicAdd(&L.keys[rho[j]], (unsigned long long)__double2ll_rn((double)x[j] * to_fix));
                    }
                }
                if (tp == 0) {
                    rpk[2 * h] = (unsigned)rho[0] | ((unsigned)rho[1] << 8) | ((unsigned)rho[2] << 16) | ((unsigned)rho[3] << 24);
                    rpk[2 * h + 1] = (unsigned)rho[4] | ((unsigned)rho[5] << 8) | ((unsigned)rho[6] << 16) | ((unsigned)rho[7] << 24);
                }
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {  // fixed butterfly: every lane ends with the same bits
            ctop += __shfl_xor(ctop, off, 64);
            stop += __shfl_xor(stop, off, 64);
        }
        wave_lds_sync();
        // ---- 4. suffix scan over the buckets: position pos sees buckets pos + 1 .. (the "all" bucket nfin is ctop / stop) ----
        int csuf[4];
        long long fsuf[4];
        {
            int crun = 0;
            long long frun = 0;
#pragma unroll
            for (int q = 3; q >= 0; --q) {
                const int b = 4 * lane + q + 1;
                if (b < HK) {
                    crun += (int)L.cnt[b];
                    frun += (long long)L.keys[b];
                }
                csuf[q] = crun;
                fsuf[q] = frun;
            }
            int cinc = crun;  // inclusive suffix over the lanes >= this one
            long long finc = frun;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int co = __shfl_down(cinc, off, 64);
                const long long fo = __shfl_down(finc, off, 64);
                if (lane + off < 64) {
                    cinc += co;
                    finc += fo;
                }
            }
            const int cex = cinc - crun;  // lanes above this one
            const long long fex = finc - frun;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                csuf[q] += cex;
                fsuf[q] += fex;
            }
        }
        float wown[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pos = 4 * lane + q;
            float w = 0.f, d = 0.f;
            if (pos < nfin) {
                const int cn = csuf[q] + ctop;
                const float c1 = -town[q];
                double hs = (double)cn * (double)c1 + ((double)fsuf[q] * from_fix + stop);
                hs = hs > 0.0 ? hs : 0.0;  // a sum of non-negative terms; rounding may leave -1e-17
                const float M = c * (float)hs;
                lsum += logf(1.0f + M);
                w = c * __frcp_rn(1.0f + M);
                d = -(w * (float)cn);
            }
            wown[q] = w;
            if (oown[q] >= 0) delta[cb + oown[q]] = d;
        }
        // ---- 5. exclusive prefix of w over the sorted positions ----
        {
            const float l0 = wown[0], l1 = l0 + wown[1], l2 = l1 + wown[2], l3 = l2 + wown[3];
            float inc = l3;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float o = __shfl_up(inc, off, 64);
                if (lane >= off) inc += o;
            }
            const float ex = __shfl_up(inc, 1, 64);  // total of the lanes below
            const float base = (lane == 0) ? 0.f : ex;
            L.pw[pad_slot(4 * lane)] = base;
            L.pw[pad_slot(4 * lane + 1)] = base + l0;
            L.pw[pad_slot(4 * lane + 2)] = base + l1;
            L.pw[pad_slot(4 * lane + 3)] = base + l2;
        }
        wave_lds_sync();
        for (int tp = 0; tp < ntiles; tp += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tile = tp + h;
                if (tile >= ntiles) break;
                int rho[HSPL];
                if (ntiles <= 2) {
#pragma unroll
                    for (int j = 0; j < HSPL; ++j) rho[j] = (int)((rpk[2 * h + (j >> 2)] >> (8 * (j & 3))) & 0xffu);
                } else {  // only the ranks of the first two tiles stay in registers: search again
                    float x[HSPL];
                    load_tile(spu, tile * 64 * HSPL, S, lane, x);
                    search_tile(L.ts, x, rho, nfin);
                }
#pragma unroll
                for (int j = 0; j < HSPL; ++j) {
                    const int s = tile * 64 * HSPL + lane + 64 * j;
                    if (s < S) {
                        const float dv = L.pw[pad_slot(rho[j])];
                        if (wrote) Du[s] += dv;
                        else Du[s] = dv;
                    }
                }
            }
        }
        wrote = true;
        wave_lds_sync();  // the next chunk rewrites the arrays
    }
    if (!wrote)
        for (int s = lane; s < S; s += 64) Du[s] = 0.f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lsum += __shfl_xor(lsum, off, 64);
    if (lane == 0 && loss_part) loss_part[u] = lsum;
}

}  // namespace tmf

using namespace tmf;

extern "C" int tmf_wmrb_hinge2(const int64_t* rowptr, const float* val, const float* p, const float* sp, int32_t n_users,
                               int32_t S, float c, float* delta, float* D, float* loss_part, void* stream) {
    return tmf_wmrb_hinge2_ordered(rowptr, val, p, sp, n_users, S, c, delta, D, loss_part, nullptr, stream);
}

extern "C" int tmf_wmrb_hinge2_ordered(const int64_t* rowptr, const float* val, const float* p, const float* sp, int32_t n_users,
                                       int32_t S, float c, float* delta, float* D, float* loss_part, const int32_t* user_order,
                                       void* stream) {
    if (n_users == 0) return TMF_OK;
    TMF_REQUIRE(rowptr && sp && D && n_users > 0 && S > 0, "wmrb_hinge2: bad arguments");
    int lg = 0;  // ceil(log2 S)
    while (((int64_t)1 << lg) < (int64_t)S) ++lg;
    const int fixbits = 62 - lg;  // S values below 2^fixbits each: their sum fits a signed 64-bit integer
    const unsigned blocks = (unsigned)(((int64_t)n_users + HWAVES - 1) / HWAVES);
    hipLaunchKernelGGL(k_wmrb_hinge2, dim3(blocks), dim3(64 * HWAVES), 0, (hipStream_t)stream, rowptr, val, p, sp, (int)S, fixbits,
                       c, (int64_t)n_users, delta, D, loss_part, user_order);
    return check_launch("tmf_wmrb_hinge2");
}
