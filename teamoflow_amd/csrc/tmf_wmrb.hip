// K4+K5: user side of one WMRB epoch.  One 256-thread workgroup per user:
//   phase 1  sp[s] = <U[u], V[R[u,s]]> for the S static negatives      -> LDS
//   phase 2  per chunk of the user's interactions (<= kPosChunk):
//              2a p_k for positives, c1_k = 1 - p_k                      -> LDS
//              2b M_k = c*sum_s max(c1_k + sp[s], 0), cnt_k, w_k, delta_k, log(1+M_k)
//              2c D[s] += sum_k [c1_k + sp[s] >= 0] w_k                  (thread per sample)
//              2d gU += sum_k delta_k V[j_k]
//   phase 3  gU += sum_s D[s] V[R[u,s]], D -> global, gU -> epilogue (fresh Adam or raw gradient)
// The [P, S] hinge tensor of loss_graphs.py:80-84 is never materialised.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "tmf_common.h"

namespace tmf {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kPosChunk = 512;
constexpr int kUnrollW = 4;
#ifndef TMF_G4_UNROLL
#define TMF_G4_UNROLL 4
#endif
constexpr int kG4Unroll = TMF_G4_UNROLL;   // rows in flight per lane group of k_wmrb_gradu4 (4 or 8; A/B builds)

__host__ __device__ inline int round4(int x) { return (x + 3) & ~3; }

// Length of the scores row as the hinge loops see it: the eight half-waves of a block sweep an eighth each, two float4
// per step, so the row is padded with -inf (terms that add 0 and never count) to 8 equal parts of an even number of
// float4 - the loops then have one wave-uniform trip count and no tail handling.
__host__ __device__ inline int sp_padded(int S) {
    const int nf4 = (S + 3) / 4;
    const int per = ((nf4 + 2 * kWaves - 1) / (2 * kWaves) + 1) & ~1;
    return 4 * 2 * kWaves * per;
}

// c0 += [x0 >= 0] + [x2 >= 0], c1 += [x1 >= 0] + [x3 >= 0]: four compares into four SGPR pairs, then four adds that take
// them as carry-in (2 VALU per term instead of cmp + cndmask + add).  Every reader is three instructions behind its
// writer: gfx940+ needs two wait states between a VALU that writes an SGPR and a VALU that reads it.
__device__ __forceinline__ void count_ge0_x4(int& c0, int& c1, float x0, float x1, float x2, float x3) {
    unsigned long long m0, m1, m2, m3;
    asm("v_cmp_le_f32_e64 %2, 0, %6\n\t"
        "v_cmp_le_f32_e64 %3, 0, %7\n\t"
        "v_cmp_le_f32_e64 %4, 0, %8\n\t"
        "v_cmp_le_f32_e64 %5, 0, %9\n\t"
        "v_addc_co_u32_e64 %0, %2, 0, %0, %2\n\t"
        "v_addc_co_u32_e64 %1, %3, 0, %1, %3\n\t"
        "v_addc_co_u32_e64 %0, %4, 0, %0, %4\n\t"
        "v_addc_co_u32_e64 %1, %5, 0, %1, %5"
        : "+v"(c0), "+v"(c1), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
}

// r_i = (sps >= -c_i) ? w_i : 0 for four (c, w) pairs, compares first, selects after (same SGPR hazard rule; the
// compiler's own sequence pads every compare / select pair with s_nop).
__device__ __forceinline__ float4 select_ge_x4(float sps, const float4 c, const float4 w) {
    unsigned long long m0, m1, m2, m3;
    float4 r;
    asm("v_cmp_ge_f32_e64 %4, %8, -%9\n\t"
        "v_cmp_ge_f32_e64 %5, %8, -%10\n\t"
        "v_cmp_ge_f32_e64 %6, %8, -%11\n\t"
        "v_cmp_ge_f32_e64 %7, %8, -%12\n\t"
        "v_cndmask_b32_e64 %0, 0, %13, %4\n\t"
        "v_cndmask_b32_e64 %1, 0, %14, %5\n\t"
        "v_cndmask_b32_e64 %2, 0, %15, %6\n\t"
        "v_cndmask_b32_e64 %3, 0, %16, %7"
        : "=&v"(r.x), "=&v"(r.y), "=&v"(r.z), "=&v"(r.w), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(sps), "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w), "v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w));
    return r;
}

// (Catalogs whose V table is larger than the L2s, or sample counts whose scores do not fit LDS, go through the sliced
// pass below instead, whose hinge step is O((S + P) log P) per user - csrc/tmf_hinge.hip.)
template <int G, int NV, typename T>
__device__ __forceinline__ void wmrb_user_body(
    const int64_t u, char* smem_raw,
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const int32_t* __restrict__ R, int S, float c,
    const T* __restrict__ U_old,
    const T* __restrict__ V_old, void* __restrict__ U_out, float* __restrict__ delta,
    float* __restrict__ Dg, float* __restrict__ loss_part, float* __restrict__ pos_part, int epi,
    tmf_adam adam) {
    constexpr int NG = 64 / G;          // groups per wave
    constexpr int NGB = NG * kWaves;    // groups per block
    constexpr int LD = 4 * G * NV;
    const int S4 = round4(S), SP = sp_padded(S);
    float* sp = reinterpret_cast<float*>(smem_raw);   // [SP] scores, tail -inf
    float* Dl = sp + SP;                              // [S]  D[u, :]
    float* c1 = Dl + S4;                              // [kPosChunk] 1 - p_k, -inf for non-positives
    float* wl = c1 + kPosChunk;                       // [kPosChunk] w_k
    float* dl = wl + kPosChunk;                       // [kPosChunk] delta_k
    int* ci = reinterpret_cast<int*>(dl + kPosChunk); // [kPosChunk] item of a positive entry, -1 otherwise
    int* Rl = ci + kPosChunk;                         // [S4] the user's negatives: the row gathers then depend on an
                                                      //      LDS read, not on a second global load
    float* pm = reinterpret_cast<float*>(Rl + S4);    // [2 kWaves][32] partials of M_k
    int* pc = reinterpret_cast<int*>(pm + kWaves * 64);                     // [2 kWaves][32] partials of cnt_k
    float* red = reinterpret_cast<float*>(pc + kWaves * 64);                // [kWaves][LD] + 2*kWaves

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane & (G - 1), grp = lane / G, gid = wave * NG + grp;
    const int64_t rb = rowptr[u], re = rowptr[u + 1];
    const int32_t* Ru = R + u * (int64_t)S;

    // phase 0: any positive at all?
    int mine = 0;
    for (int64_t k = rb + tid; k < re; k += kThreads) mine |= (val[k] > 0.f);
    const int anypos = __syncthreads_or(mine);

    Frag<NV> acc;
    zero<NV>(acc);
    float lsum = 0.f, npos = 0.f;

    if (!anypos) {
        for (int64_t k = rb + tid; k < re; k += kThreads) delta[k] = 0.f;
        for (int s = tid; s < S; s += kThreads) Dg[u * (int64_t)S + s] = 0.f;
    } else {
        Frag<NV> x;
        load_row<G, NV>(x, U_old, u, g);
        // ---- phase 1 ----
        for (int s = tid; s < S; s += kThreads) Rl[s] = Ru[s];
        __syncthreads();
        const int32_t* Ri = Rl;
        for (int s0 = gid; s0 < S; s0 += NGB * kUnrollW) {
            Raw<NV, T> raw[kUnrollW];
#pragma unroll
            for (int t = 0; t < kUnrollW; ++t) {
                const int s = s0 + t * NGB;
                load_raw<G, NV>(raw[t], V_old, Ri[s < S ? s : 0], g);
            }
#pragma unroll
            for (int t = 0; t < kUnrollW; ++t) {
                const int s = s0 + t * NGB;
                Frag<NV> y;
                to_frag<NV>(y, raw[t]);
                const float d = group_allsum<G>(dot_partial<NV>(x, y));
                if (g == 0 && s < S) sp[s] = d;
            }
        }
        for (int s = tid; s < SP; s += kThreads) {
            if (s < S) Dl[s] = 0.f;
            else sp[s] = -INFINITY;
        }
        __syncthreads();

        // ---- phase 2 ----
        for (int64_t cb = rb; cb < re; cb += kPosChunk) {
            const int len = (int)((re - cb < kPosChunk) ? (re - cb) : kPosChunk);
            const int len4 = round4(len);
            // 2a
            for (int kk = tid; kk < len; kk += kThreads) ci[kk] = (val[cb + kk] > 0.f) ? col[cb + kk] : -1;
            __syncthreads();
            for (int k0 = gid; k0 < len; k0 += NGB * kUnrollW) {
                Raw<NV, T> raw[kUnrollW];
                bool pos[kUnrollW];
#pragma unroll
                for (int t = 0; t < kUnrollW; ++t) {
                    const int kk = k0 + t * NGB;
                    const int item = (kk < len) ? ci[kk] : -1;
                    pos[t] = item >= 0;
                    load_raw<G, NV>(raw[t], V_old, pos[t] ? item : 0, g);
                }
#pragma unroll
                for (int t = 0; t < kUnrollW; ++t) {
                    const int kk = k0 + t * NGB;
                    Frag<NV> y;
                    to_frag<NV>(y, raw[t]);
                    const float p = group_allsum<G>(dot_partial<NV>(x, y));
                    if (g == 0 && kk < len) c1[kk] = pos[t] ? (1.0f - p) : -INFINITY;
                }
            }
            for (int kk = len + tid; kk < len4; kk += kThreads) {
                c1[kk] = -INFINITY;
                wl[kk] = 0.f;
            }
            __syncthreads();
            // 2b: hinge sums.  Tiles of 32 interactions on the lanes of a half-wave; the 8 half-waves sweep one
            // eighth of the samples each with broadcast ds_read_b128 of sp (two float4 in flight); the eight
            // partials are combined in fixed order.
            {
                constexpr int kSG = 2 * kWaves;  // sample groups = half-waves of the block
                const int per_sg = SP / (4 * kSG);  // float4 per sample group: even, the same for every group
                const int sg = wave * 2 + (lane >> 5), l32 = lane & 31;
                const float4* sp4 = reinterpret_cast<const float4*>(sp) + sg * per_sg;
                for (int kb = 0; kb < len; kb += 32) {
                    const int kk = kb + l32;
                    const float c1v = (kk < len) ? c1[kk] : -INFINITY;  // -inf: every term is 0 and never counts
                    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
                    int cnt = 0, cnt2 = 0;
#pragma unroll 2
                    for (int f = 0; f < per_sg; f += 2) {
                        const float4 q = sp4[f];
                        const float4 q2 = sp4[f + 1];
                        const float x0 = c1v + q.x, x1 = c1v + q.y, x2 = c1v + q.z, x3 = c1v + q.w;
                        const float x4 = c1v + q2.x, x5 = c1v + q2.y, x6 = c1v + q2.z, x7 = c1v + q2.w;
                        m0 += fmaxf(x0, 0.f);
                        m1 += fmaxf(x1, 0.f);
                        m2 += fmaxf(x2, 0.f);
                        m3 += fmaxf(x3, 0.f);
                        m0 += fmaxf(x4, 0.f);
                        m1 += fmaxf(x5, 0.f);
                        m2 += fmaxf(x6, 0.f);
                        m3 += fmaxf(x7, 0.f);
                        count_ge0_x4(cnt, cnt2, x0, x1, x2, x3);
                        count_ge0_x4(cnt, cnt2, x4, x5, x6, x7);
                    }
                    cnt += cnt2;
                    pm[sg * 32 + l32] = (m0 + m1) + (m2 + m3);
                    pc[sg * 32 + l32] = cnt;
                    __syncthreads();
                    if (tid < 32 && kk < len) {
                        float w = 0.f, d = 0.f;
                        if (c1v != -INFINITY) {
                            float Ms = 0.f;
                            int cn = 0;
#pragma unroll
                            for (int h = 0; h < kSG; ++h) {
                                Ms += pm[h * 32 + tid];
                                cn += pc[h * 32 + tid];
                            }
                            const float M = c * Ms;
                            lsum += logf(1.0f + M);
                            npos += 1.f;
                            w = c * __frcp_rn(1.0f + M);
                            d = -(w * (float)cn);
                        }
                        wl[kk] = w;
                        dl[kk] = d;
                        delta[cb + kk] = d;
                    }
                    __syncthreads();
                }
            }
            // 2c: one thread per sample; [c1_k + sp_s >= 0] == [sp_s >= -c1_k] exactly (rounding keeps the sign)
            for (int s = tid; s < S; s += kThreads) {
                const float sps = sp[s];
                float d0 = 0.f, d1 = 0.f;
                const float4* c4 = reinterpret_cast<const float4*>(c1);
                const float4* w4 = reinterpret_cast<const float4*>(wl);
#pragma unroll 4
                for (int q = 0; q < len4 / 4; ++q) {
                    const float4 r = select_ge_x4(sps, c4[q], w4[q]);
                    d0 += r.x;
                    d1 += r.y;
                    d0 += r.z;
                    d1 += r.w;
                }
                Dl[s] += d0 + d1;
            }
            // 2d
            for (int k0 = gid; k0 < len; k0 += NGB * kUnrollW) {
                Raw<NV, T> raw[kUnrollW];
                float d[kUnrollW];
#pragma unroll
                for (int t = 0; t < kUnrollW; ++t) {
                    const int kk = k0 + t * NGB;
                    d[t] = (kk < len) ? dl[kk] : 0.f;
                    load_raw<G, NV>(raw[t], V_old, d[t] != 0.f ? ci[kk] : 0, g);
                }
#pragma unroll
                for (int t = 0; t < kUnrollW; ++t) {
                    Frag<NV> y;
                    to_frag<NV>(y, raw[t]);
                    axpy<NV>(acc, d[t], y);
                }
            }
            __syncthreads();
        }
        // ---- phase 3 ----
        for (int s0 = gid; s0 < S; s0 += NGB * kUnrollW) {
            Raw<NV, T> raw[kUnrollW];
            float d[kUnrollW];
#pragma unroll
            for (int t = 0; t < kUnrollW; ++t) {
                const int s = s0 + t * NGB;
                d[t] = (s < S) ? Dl[s] : 0.f;
                load_raw<G, NV>(raw[t], V_old, d[t] != 0.f ? Ri[s] : 0, g);
            }
#pragma unroll
            for (int t = 0; t < kUnrollW; ++t) {
                Frag<NV> y;
                to_frag<NV>(y, raw[t]);
                axpy<NV>(acc, d[t], y);
            }
        }
        for (int s = tid; s < S; s += kThreads) __builtin_nontemporal_store(Dl[s], Dg + u * (int64_t)S + s);
    }

    // ---- block reduction of gU (groups of a wave, then the four waves in order) and of the loss ----
    across_groups_sum<G, NV>(acc);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lsum += __shfl_xor(lsum, off, 64);
        npos += __shfl_xor(npos, off, 64);
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) reinterpret_cast<float4*>(red + wave * LD)[g + G * v] = acc.v[v];
    }
    if (lane == 0) {
        red[kWaves * LD + wave] = lsum;
        red[kWaves * LD + kWaves + wave] = npos;
    }
    __syncthreads();
    if (wave == 0 && grp == 0) {
        Frag<NV> tot;
#pragma unroll
        for (int v = 0; v < NV; ++v) tot.v[v] = reinterpret_cast<const float4*>(red)[g + G * v];
        for (int w = 1; w < kWaves; ++w) {
            Frag<NV> part;
#pragma unroll
            for (int v = 0; v < NV; ++v) part.v[v] = reinterpret_cast<const float4*>(red + w * LD)[g + G * v];
            add<NV>(tot, part);
        }
        row_epilogue<G, NV, T>(tot, U_old, U_out, u, g, epi, adam);
    }
    if (tid == 0) {
        const float* t = red + kWaves * LD;
        if (loss_part) loss_part[u] = (t[0] + t[1]) + (t[2] + t[3]);
        if (pos_part) pos_part[u] = (t[4] + t[5]) + (t[6] + t[7]);
    }
}

template <int G, int NV, typename T>
__global__ __launch_bounds__(kThreads) void k_wmrb_user(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const int32_t* __restrict__ R, int S, float c,
    const T* __restrict__ U_old, const T* __restrict__ V_old, void* __restrict__ U_out, float* __restrict__ delta,
    float* __restrict__ Dg, float* __restrict__ loss_part, float* __restrict__ pos_part, int epi, tmf_adam adam) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    wmrb_user_body<G, NV, T>(blockIdx.x, smem_raw, rowptr, col, val, R, S, c, U_old, V_old, U_out, delta, Dg, loss_part,
                             pos_part, epi, adam);
}

static size_t wmrb_user_lds(int S, int ld) {
    // sp + D, c1 / w / delta / item per chunk entry, the user's negatives, the reduction
    return sizeof(float) * ((size_t)sp_padded(S) + round4(S) + 4 * kPosChunk + (size_t)round4(S) + 2 * kWaves * 64 +
                            (size_t)kWaves * ld + 2 * kWaves);
}

template <int G, int NV, typename T>
static int launch_wmrb_user(const int64_t* rowptr, const int32_t* col, const float* val, const int32_t* R,
                            int32_t n_users, int32_t S, float c, const T* U_old, const T* V_old,
                            void* U_out, float* delta, float* D, float* loss_part, float* pos_part, int epi,
                            tmf_adam adam, hipStream_t stream) {
    const size_t lds = wmrb_user_lds(S, 4 * G * NV);
    if (lds > 160 * 1024) {
        set_error("wmrb_user_pass: n_samples=%d does not fit LDS (%zu bytes); use the sliced pass (tmf_wmrb_scores3 ...)", S, lds);
        return TMF_E_UNSUPPORTED;
    }
    static LdsGrant grant;  // per template instance
    if (int rc = grant_dynamic_lds(reinterpret_cast<const void*>(&k_wmrb_user<G, NV, T>), lds, grant)) return rc;
    TMF_REQUIRE_LAUNCH(n_users, kThreads, "wmrb_user_pass");
    hipLaunchKernelGGL((k_wmrb_user<G, NV, T>), dim3((unsigned)n_users), dim3(kThreads), lds, stream, rowptr, col,
                       val, R, (int)S, c, U_old, V_old, U_out, delta, D, loss_part, pos_part, epi, adam);
    return check_launch("tmf_wmrb_user_pass");
}

// ---------------------------------------------------------------------------------------------
// Sliced user pass.  The fused kernel above gathers V rows over the whole item table (51 MB at
// 100K x 128: served by the Infinity Cache at ~8 TB/s).  Here the S negatives of every user are kept
// sorted by item id (the interactions of a user are sorted by item anyway) and the catalog is cut into
// slices of ~4 MB of V rows; blocks are numbered slice-major, so the workgroups resident at any moment
// gather from ONE slice, which the XCD L2s hold (20 TB/s measured for these gathers).  Kernels:
//   k_wmrb_scores3   sp[u, s] = <U[u], V[R[u, s]]> for the negatives in the slice, p[k] = <U[u], V[j_k]> for the
//                    user's interactions in the slice                                             -> global
//   k_wmrb_hinge2    (tmf_hinge.hip) sp, p -> delta, D, loss; touches no table
//   k_wmrb_gradu3    part[slice][u] = sum_{s in slice} D[u, s] V[R[u, s]] + sum_{k in slice} delta_k V[j_k]
//                    (storing every weight at its position in the item-side entry lists on the way, so that the item
//                    pass could stream them, was measured: the 4-byte scatter cost this kernel +36 ms and saved the
//                    item pass 8 ms - writes from different XCDs do not merge; profiles/r02_hinge_rewrite.txt)
//   k_wmrb_finish    gU[u] = sum_slice part[slice][u] (fixed order) -> epilogue
// Block placement is used for speed only; any dispatch order gives the same bits.
// Both slice roles stage the ids (and weights) of a (user, slice) range through LDS so that a row gather
// depends on an LDS read only.
// (An in-launch variant - scores + hinge in one launch behind an agent-scope ticket - was measured slower and
// removed; see profiles/r01_sliced_user_pass.txt.)
// ---------------------------------------------------------------------------------------------
#ifndef TMF_STREAM_NT
#define TMF_STREAM_NT 1
#endif
// Read-once streams (ids, weights of a (user, slice) range) are loaded with the non-temporal hint, so that they do not displace
// the slice's V rows in the L2s: same box, C4 fp32, gradU 26.6 -> 25.0 ms, scores 27.9 -> 27.75 (profiles/r03_c5_experiments.txt item 10)
constexpr bool kStreamNT = TMF_STREAM_NT;
constexpr int kSliceUsers = 16;   // users per workgroup of the slice kernels (SliceLists::upg)
// Waves per workgroup of the slice kernels: 4, or 8 when a (user, slice) range is long (C4 fp32: 86 rows).  With 512-byte rows
// eight waves are 16 lane groups = ONE user per lane group, and the workgroup retires when its users are done.  Same box,
// C4 fp32 epoch (scores / gradU ms): 4 waves x 16 users 95.7 (28.15 / 27.6), 4 x 8 95.0, 8 x 16 94.1 (27.4 / 26.8), 16 x 32 94.0,
// 2 x 4 95.5; config-5 shard (18 rows per range) 4 waves 241.9, 8 waves 249.7; C4 bf16 (16-lane rows) 62.6 / 63.0 - no gain.
constexpr int kLongRange = 48;   // mean rows per (user, slice) range from which the 8-wave form is used

template <int G>
struct Stage {
    static constexpr int tile = 8 * G;  // entries staged per lane group and step: 16 KB of LDS per workgroup for every G
};

template <int G>
__device__ __forceinline__ int* slice_stage(char* smem_raw, int gid) {
    return reinterpret_cast<int*>(smem_raw) + gid * 2 * Stage<G>::tile;  // [ids | weights] per lane group
}

// One list (the negatives or the interactions of a user that fall into the slice): entries [beg, end) of `list`.
//   GRADU = false: out[t] = <x, V[list[t]]>
//   GRADU = true : acc += wts[t] V[list[t]] (rows with weight 0 are not loaded)
template <int G, int NV, typename T, bool GRADU>
__device__ __forceinline__ void slice_list(int* ids, float* dst, const int32_t* __restrict__ list, int beg, int end,
                                           const T* __restrict__ V, const Frag<NV>& x, Frag<NV>& acc,
                                           float* __restrict__ out, const float* __restrict__ wts, int g, int safe) {
    constexpr int kStageTile = Stage<G>::tile;
    for (int t0 = beg; t0 < end; t0 += kStageTile) {
        const int cnt = (end - t0 < kStageTile) ? end - t0 : kStageTile;
        for (int e = g; e < cnt; e += G) {
            ids[e] = kStreamNT ? __builtin_nontemporal_load(list + t0 + e) : list[t0 + e];
            if (GRADU) dst[e] = kStreamNT ? __builtin_nontemporal_load(wts + t0 + e) : wts[t0 + e];
        }
        wave_lds_sync();
        float keep = 0.f;  // scores: lane g keeps the score of entry (e & (G-1)) == g until G of them are complete
        for (int e0 = 0; e0 < cnt; e0 += kUnrollW) {
            Raw<NV, T> raw[kUnrollW];
            float d[kUnrollW];
            // the four ids (and weights) of this step in ONE LDS read each (e0 % 4 == 0, 16-byte aligned tile buffers;
            // slots past cnt hold stale values and are never used): one LDS round trip before the four row loads
            static_assert(kUnrollW == 4, "vector LDS reads assume four entries per step");
            const int4 id4 = *reinterpret_cast<const int4*>(ids + e0);
            const float4 w4 = GRADU ? *reinterpret_cast<const float4*>(dst + e0) : make_float4(0.f, 0.f, 0.f, 0.f);
            const int idv[4] = {id4.x, id4.y, id4.z, id4.w};
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int t = 0; t < kUnrollW; ++t) {
                const int e = e0 + t;
                bool want = e < cnt;
                d[t] = 0.f;
                if (GRADU && want) { d[t] = wv[t]; want = d[t] != 0.f; }
                load_raw<G, NV>(raw[t], V, want ? idv[t] : safe, g);  // `safe`: a row that is resident (first of the window)
            }
#pragma unroll
            for (int t = 0; t < kUnrollW; ++t) {
                Frag<NV> y;
                to_frag<NV>(y, raw[t]);
                if (GRADU) {
                    axpy<NV>(acc, d[t], y);
                } else {
                    const int e = e0 + t;
                    const float pr = group_allsum<G>(dot_partial<NV>(x, y));
                    if (g == (e & (G - 1))) keep = pr;
                    // a full run of G scores (or the tail of the tile): one contiguous 4*G-byte store per group
                    if (e < cnt && ((e & (G - 1)) == G - 1 || e == cnt - 1)) {
                        if (g <= (e & (G - 1))) __builtin_nontemporal_store(keep, out + t0 + (e & ~(G - 1)) + g);
                    }
                }
            }
        }
    }
}

// The scores walk of slice_list in a leaner form for rows of 32 lanes whose table is addressable with 32-bit byte offsets
// (SliceLists::lean): four rows per step as before, but ONE instruction per row address (load_raw32), the dot product on packed
// pairs (dot_raw), the four scores of a step reduced together (Reduce4x32: 10 instructions instead of 4 x 7) and stored by four
// lanes, no compare / select per entry - the tail of a tile points at the resident row `safe` from the moment it is staged.
// Scores agree with the general form to rounding (another summation tree), exactly on dyadic data.
template <int NV, typename T>
__device__ __forceinline__ void slice_scores_lean(int* ids, const int32_t* __restrict__ list, int beg, int end, const T* __restrict__ V,
                                                  const Raw<NV, T>& x, float* __restrict__ out, int g, int lane, int safe) {
    constexpr int G = 32, kStageTile = Stage<G>::tile;
    const uint32_t loff = 16u * (uint32_t)g;
    const int my_entry = Reduce4x32::entry_of_lane(lane);
    // every entry of a step ends up on 8 lanes: lane l keeps the score of step (l & 7) of a block of eight steps, so that after
    // eight steps the 32 lanes hold 32 consecutive scores and leave in ONE 128-byte store (16-byte stores per step cost the kernel
    // 4 ms at C4: partial lines)
    const int my_step = lane & 7, my_slot = 4 * my_step + my_entry;
    for (int t0 = beg; t0 < end; t0 += kStageTile) {
        const int cnt = (end - t0 < kStageTile) ? end - t0 : kStageTile;
        const int cnt4 = (cnt + 3) & ~3;
        for (int e = g; e < cnt4; e += G) ids[e] = e < cnt ? (kStreamNT ? __builtin_nontemporal_load(list + t0 + e) : list[t0 + e]) : safe;
        wave_lds_sync();
        float keep = 0.f;
        for (int e0 = 0; e0 < cnt4; e0 += 4) {
            const int4 id4 = *reinterpret_cast<const int4*>(ids + e0);
            Raw<NV, T> y[4];
            load_raw32<G, NV>(y[0], V, (uint32_t)id4.x, loff);
            load_raw32<G, NV>(y[1], V, (uint32_t)id4.y, loff);
            load_raw32<G, NV>(y[2], V, (uint32_t)id4.z, loff);
            load_raw32<G, NV>(y[3], V, (uint32_t)id4.w, loff);
            float pr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) pr[t] = dot_raw<NV>(x, y[t]);
            const float sc = Reduce4x32::run(pr, lane);
            const int step = (e0 >> 2) & 7;
            keep = (step == my_step) ? sc : keep;
            if (step == 7 || e0 + 4 >= cnt4) {   // a full block of 32 scores, or the tail of the tile
                const int at = (e0 & ~31) + my_slot;
                if (at < cnt) __builtin_nontemporal_store(keep, out + t0 + at);
            }
        }
    }
}

// Arguments shared by the two slice kernels: the per-user sorted negatives with their slice offsets, and the CSR of the
// interactions (sorted by item inside a user) with theirs.
struct SliceLists {
    const int32_t* R;        // [n_users, S] negatives, ascending item id per user
    const int32_t* off;      // [n_users, n_slices + 1] first negative of every slice
    const int64_t* rowptr;   // [n_users + 1]
    const int32_t* col;      // [nnz]
    const int32_t* poff;     // [n_users, n_slices + 1] first interaction of every slice, relative to rowptr[u]
    int n_slices, S;
    int64_t n_users, n_groups;
    int upg;                 // users per workgroup
    int sl0, nsl;            // slices [sl0, sl0 + nsl) are covered by this launch ...
    int item_base;           // ... and V points at item row item_base (windowed V: only these rows are resident)
    int xcd;                 // 1: XCD-major block order (slice_of_block)
    int lean;                // 1: V is addressable with 32-bit byte offsets (tmf_slice_lists.n_items says so): slice_scores_lean
};

// Which (slice, user group) a block works on.  Plain order is slice-major: every resident workgroup walks the SAME slice, so
// each of the 8 XCD L2s holds a copy of it.  XCD-major order (a.xcd) relies on the observed round-robin placement - blocks b
// and b + 8 share an XCD - and gives XCD x the slices sl0 + 8 i + x: eight DIFFERENT slices are resident at a time, one per
// L2, and the eight workgroups that visit one user group together share its U rows / offsets through the Infinity Cache.
// Speed only: both orders enumerate every (slice, group) pair exactly once, whatever the hardware does with the blocks.
__device__ __forceinline__ bool slice_of_block(const SliceLists& a, int64_t b, int64_t& sl, int64_t& grp) {
    if (!a.xcd) {
        sl = a.sl0 + b / a.n_groups;
        grp = b % a.n_groups;
        return true;
    }
    const int64_t x = b & 7, j = b >> 3;
    sl = a.sl0 + 8 * (j / a.n_groups) + x;
    grp = j % a.n_groups;
    return sl < a.sl0 + a.nsl;
}

// V rebased so that global item ids index it (rows outside the window are never touched: every id of the launched slices
// lies inside, idle lanes read row item_base)
template <int G, int NV, typename T>
__device__ __forceinline__ const T* window_base(const T* V, int item_base) {
    return V - (int64_t)item_base * (4 * G * NV);
}

template <int G, int NV, typename T, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_wmrb_scores3(SliceLists a, const T* __restrict__ U, const T* __restrict__ V,
                                                           float* __restrict__ sp, float* __restrict__ p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NG = 64 / G, NGB = NG * WAVES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane & (G - 1), gid = wave * NG + lane / G;
    int* ids = slice_stage<G>(smem_raw, gid);
    float* dst = reinterpret_cast<float*>(ids + Stage<G>::tile);
    int64_t sl, grp;
    if (!slice_of_block(a, blockIdx.x, sl, grp)) return;
    const int64_t ubeg = grp * a.upg;
    const int64_t uend = (ubeg + a.upg < a.n_users) ? ubeg + a.upg : a.n_users;
    V = window_base<G, NV, T>(V, a.item_base);
    if constexpr (G == 32 && sizeof(Raw<NV, T>) <= 16) {
        if (a.lean) {
            for (int64_t u = ubeg + gid; u < uend; u += NGB) {
                const int64_t o = u * (a.n_slices + 1) + sl;
                const int nb = a.off[o], ne = a.off[o + 1], pb = a.poff[o], pe = a.poff[o + 1];
                if (nb == ne && pb == pe) continue;
                Raw<NV, T> x;   // the user's own row as stored: read once per (user, slice) visit
                {
                    const char* xp = reinterpret_cast<const char*>(U) + u * (int64_t)RowBytes<G, NV, T>::value + 16 * g;
                    if constexpr (std::is_same<T, float>::value) {
#pragma unroll
                        for (int v = 0; v < NV; ++v) x.v[v] = load_f4_nt(reinterpret_cast<const float*>(xp + 16 * G * v));
                    } else {
#pragma unroll
                        for (int pv = 0; pv < NV / 2; ++pv) x.v[pv] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(xp + 16 * G * pv));
                    }
                }
                slice_scores_lean<NV, T>(ids, a.R + u * (int64_t)a.S, nb, ne, V, x, sp + u * (int64_t)a.S, g, lane, a.item_base);
                const int64_t rb = a.rowptr[u];
                slice_scores_lean<NV, T>(ids, a.col + rb, pb, pe, V, x, p + rb, g, lane, a.item_base);
            }
            return;
        }
    }
    for (int64_t u = ubeg + gid; u < uend; u += NGB) {
        const int64_t o = u * (a.n_slices + 1) + sl;
        const int nb = a.off[o], ne = a.off[o + 1], pb = a.poff[o], pe = a.poff[o + 1];
        if (nb == ne && pb == pe) continue;
        Frag<NV> x, none;
        load_row_nt<G, NV>(x, U, u, g);   // read once per (user, slice) visit
        slice_list<G, NV, T, false>(ids, dst, a.R + u * (int64_t)a.S, nb, ne, V, x, none, sp + u * (int64_t)a.S, nullptr, g,
                                    a.item_base);
        const int64_t rb = a.rowptr[u];
        slice_list<G, NV, T, false>(ids, dst, a.col + rb, pb, pe, V, x, none, p + rb, nullptr, g, a.item_base);
    }
}

// slice_first >= 0, accumulate 1 | 2: this launch covers ONE slice (slice_first) and writes (accumulate == 1) or adds (2) into
// the single-layer `part` (plain read-modify-write; launches of consecutive slices are ordered by the stream).
// slice_first >= 0, accumulate 3 | 4: this launch covers the ROUND of up to eight slices slice_first + x, x = blockIdx.x & 7
// (XCD-major order), each into its own layer part[x]: 3 writes the layer, 4 adds to it.
template <int G, int NV, typename T, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_wmrb_gradu3(SliceLists a, const T* __restrict__ V, const float* __restrict__ D,
                                                          const float* __restrict__ delta, float* __restrict__ part,
                                                          int slice_first, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NG = 64 / G, NGB = NG * WAVES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane & (G - 1), gid = wave * NG + lane / G;
    int* ids = slice_stage<G>(smem_raw, gid);
    float* dst = reinterpret_cast<float*>(ids + Stage<G>::tile);
    int64_t sl, grp, layer = 0;
    if (slice_first < 0) {
        if (!slice_of_block(a, blockIdx.x, sl, grp)) return;
    } else if (accumulate >= 3) {
        layer = blockIdx.x & 7;
        sl = slice_first + layer;
        grp = blockIdx.x >> 3;
        if (sl >= a.sl0 + a.nsl) return;
        accumulate -= 2;
    } else {
        sl = slice_first;
        grp = blockIdx.x;
    }
    V = window_base<G, NV, T>(V, a.item_base);
    const int64_t ubeg = grp * a.upg;
    const int64_t uend = (ubeg + a.upg < a.n_users) ? ubeg + a.upg : a.n_users;
    for (int64_t u = ubeg + gid; u < uend; u += NGB) {
        const int64_t o = u * (a.n_slices + 1) + sl;
        const int nb = a.off[o], ne = a.off[o + 1], pb = a.poff[o], pe = a.poff[o + 1];
        Frag<NV> acc, none;
        zero<NV>(acc);
        const int64_t us = u * (int64_t)a.S, rb = a.rowptr[u];
        slice_list<G, NV, T, true>(ids, dst, a.R + us, nb, ne, V, none, acc, nullptr, D + us, g, a.item_base);
        slice_list<G, NV, T, true>(ids, dst, a.col + rb, pb, pe, V, none, acc, nullptr, delta + rb, g, a.item_base);
        if (accumulate == 0) {
            store_row_f32<G, NV, T>(acc, part, sl * a.n_users + u, g);
        } else {  // one launch per slice: part is a single [users, ld] layer summed in slice order
            const int64_t prow = layer * a.n_users + u;
            if (accumulate == 2) {
                Frag<NV> prev;
                load_row_f32<G, NV, T>(prev, part, prow, g);
                add<NV>(prev, acc);
                acc = prev;
            }
            store_row_f32<G, NV, T>(acc, part, prow, g);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Row-stationary user gradient ("gradu4"): a lane group OWNS K users, keeps their gradient rows in registers and walks
// ALL slices itself - no per-(user, slice) partial row, no finish kernel, no re-read of anything per visit.  Built for
// catalogs far beyond the L2s, where a slice small enough for one L2 leaves a (user, slice) range a handful of rows and
// the slice-major kernels above spend their time on the offsets -> ids -> rows round trips of every visit:
//   * the slice offsets of a user are cached 32 slices at a time in one register per list (lane l holds slice base + l);
//   * the ids and weights of slice sl + 1 are loaded (one entry per lane, both lists, K users) while slice sl is being
//     gathered, and handed over through a double-buffered LDS stage - a row gather only ever waits for LDS;
//   * a launch covers a block of users small enough for all its workgroups to be resident together, so they walk the
//     slices in loose lockstep (one workgroup barrier per slice keeps the lane groups of a workgroup together) and the
//     slice being gathered from stays in the L2s; the host launches block after block.
// Sum order of a user's row: slices ascending, negatives then interactions inside a slice, entries ascending - ONE running
// fp32 sum (gradu3 + finish add per-slice partial sums), so the two forms agree to rounding, not to the bit.
// ---------------------------------------------------------------------------------------------
template <int G>
struct Stage4 {
    static constexpr int cap = G < 16 ? G : 16;   // entries of one (user, slice) visit staged per step; longer visits finish inline
};

template <int G, int NV, typename T, int K, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 2) void k_wmrb_gradu4(   // two workgroups per CU: 4 waves per SIMD, <= 128 VGPRs
    SliceLists a, const T* __restrict__ V, const float* __restrict__ D,
                                                          const float* __restrict__ delta, const T* __restrict__ U_old,
                                                          void* __restrict__ U_out, int epi, tmf_adam adam, int64_t user_begin,
                                                          int64_t user_end, int* __restrict__ sync, int lag) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NG = 64 / G, NGB = NG * WAVES, CAP = Stage4<G>::cap, WIN = G - 1, UG = (CAP >= 8) ? kG4Unroll : kUnrollW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane & (G - 1), gid = wave * NG + lane / G;
    int* stage = reinterpret_cast<int*>(smem_raw) + gid * (2 * K * 2 * CAP);   // [buffer][user][ids CAP | weights CAP]
    V = window_base<G, NV, T>(V, a.item_base);
    const int64_t u0 = user_begin + ((int64_t)blockIdx.x * NGB + gid) * K;
    const int sl_end = a.sl0 + a.nsl;
    Frag<NV> acc[K];
    int us[K], rb[K];          // first negative / first interaction of the user (interactions + n_users * n_samples < 2^31)
    bool live[K];
    int ocn[K], ocp[K];        // offsets cache: lane l = offset of slice base + l (negatives / interactions)
    int c_tot[K];              // entries of the visit being gathered
#pragma unroll
    for (int k = 0; k < K; ++k) {
        zero<NV>(acc[k]);
        live[k] = u0 + k < user_end;
        const int64_t u = live[k] ? u0 + k : user_begin;
        us[k] = (int)(u * (int64_t)a.S);
        rb[k] = (int)a.rowptr[u];
    }
    int base = a.sl0;
    auto refill = [&](int b) {
        base = b;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int64_t u = live[k] ? u0 + k : user_begin;
            const int64_t o = u * (a.n_slices + 1) + ((b + g <= a.n_slices) ? b + g : a.n_slices);
            ocn[k] = a.off[o];
            ocp[k] = a.poff[o];
        }
    };
    // entry e (= this lane) of the visit (user k, slice sl): its id and weight, or the resident row with weight 0
    int n_tot[K], n_id[K];
    float n_w[K];
    auto fetch = [&](int sl) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int nb = group_read<G>(ocn[k], sl - base), ne = group_read<G>(ocn[k], sl + 1 - base);
            const int pb = group_read<G>(ocp[k], sl - base), pe = group_read<G>(ocp[k], sl + 1 - base);
            const int cn = live[k] ? ne - nb : 0;
            n_tot[k] = live[k] ? cn + (pe - pb) : 0;
            const bool neg = g < cn, any = g < n_tot[k] && g < CAP;
            const int at = neg ? us[k] + nb + g : rb[k] + pb + (g - cn);
            n_id[k] = any ? (neg ? a.R : a.col)[at] : a.item_base;
            n_w[k] = any ? (neg ? D : delta)[at] : 0.f;
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            int* ids = stage + (buf * K + k) * 2 * CAP;
            if (g < CAP) {
                ids[g] = n_id[k];
                reinterpret_cast<float*>(ids + CAP)[g] = n_w[k];
            }
            c_tot[k] = n_tot[k];
        }
        wave_lds_sync();
    };
    refill(a.sl0);
    fetch(a.sl0);
    commit(0);
    for (int sl = a.sl0; sl < sl_end; ++sl) {
        const int buf = (sl - a.sl0) & 1;
        const bool more = sl + 1 < sl_end;
        if (more) {
            if (sl + 2 - base > WIN) refill(sl + 1);
            fetch(sl + 1);   // in flight while this slice's rows are gathered
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int* ids = stage + (buf * K + k) * 2 * CAP;
            const float* ws = reinterpret_cast<const float*>(ids + CAP);
            const int cnt = c_tot[k] < CAP ? c_tot[k] : CAP;
            for (int e0 = 0; e0 < cnt; e0 += UG) {
                static_assert(((UG == 4 || UG == 8) && CAP % UG == 0) || CAP < 4, "vector LDS reads assume four or eight entries per step");
                Raw<NV, T> raw[UG];
                float d[UG];
                int idv[UG];
                if constexpr (CAP >= 4) {
#pragma unroll
                    for (int q = 0; q < UG / 4; ++q) {
                        const int4 id4 = *reinterpret_cast<const int4*>(ids + e0 + 4 * q);
                        const float4 w4 = *reinterpret_cast<const float4*>(ws + e0 + 4 * q);
                        idv[4 * q] = id4.x; idv[4 * q + 1] = id4.y; idv[4 * q + 2] = id4.z; idv[4 * q + 3] = id4.w;
                        d[4 * q] = w4.x; d[4 * q + 1] = w4.y; d[4 * q + 2] = w4.z; d[4 * q + 3] = w4.w;
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < UG; ++t) {
                        idv[t] = (e0 + t < CAP) ? ids[e0 + t] : a.item_base;
                        d[t] = (e0 + t < CAP) ? ws[e0 + t] : 0.f;
                    }
                }
#pragma unroll
                for (int t = 0; t < UG; ++t) {
                    const bool want = e0 + t < cnt && d[t] != 0.f;   // slots past cnt hold the resident row with weight 0
                    if (!want) d[t] = 0.f;
                    load_raw<G, NV>(raw[t], V, want ? idv[t] : a.item_base, g);
                }
#pragma unroll
                for (int t = 0; t < UG; ++t) {
                    Frag<NV> y;
                    to_frag<NV>(y, raw[t]);
                    axpy<NV>(acc[k], d[t], y);
                }
            }
            // a visit longer than the stage (more than CAP entries of ONE user in ONE slice): the rest straight from memory
            if (c_tot[k] > CAP) {
                const int64_t o = (u0 + k) * (a.n_slices + 1) + sl;
                const int nb = a.off[o], cn = a.off[o + 1] - nb, pb = a.poff[o];
                for (int e = CAP; e < c_tot[k]; ++e) {
                    const bool neg = e < cn;
                    const int at = neg ? us[k] + nb + e : rb[k] + pb + (e - cn);
                    const float w = (neg ? D : delta)[at];
                    if (w != 0.f) {
                        Frag<NV> y;
                        load_row<G, NV>(y, V, (neg ? a.R : a.col)[at], g);
                        axpy<NV>(acc[k], w, y);
                    }
                }
            }
        }
        if (more) commit(buf ^ 1);
        if (sync != nullptr && tid == 0) step_rendezvous(sync, sl - a.sl0, lag, (int)gridDim.x);
        __syncthreads();   // lockstep of the workgroup's lane groups (the other buffer is private to this wave: no hazard)
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (live[k]) row_epilogue<G, NV, T>(acc[k], U_old, U_out, u0 + k, g, epi, adam);
}

// `part` and `U_out` are NOT restrict-qualified: with TMF_EPI_GRAD the caller may sum the layers in place (U_out == part,
// include/tmf.h) - every lane loads all layers of its row before it stores the row.
template <int G, int NV, typename T>
__global__ __launch_bounds__(kThreads) void k_wmrb_finish(const float* part, int n_slices, int64_t n_users,
                                                          const T* __restrict__ U_old, void* U_out, int epi,
                                                          tmf_adam adam) {
    constexpr int NG = 64 / G, NGB = NG * kWaves;
    const int lane = threadIdx.x & 63, g = lane & (G - 1);
    const int64_t u = (int64_t)blockIdx.x * NGB + (threadIdx.x >> 6) * NG + lane / G;
    if (u >= n_users) return;
    Frag<NV> acc;
    load_row_f32<G, NV, T, TMF_NT_FIN>(acc, part, u, g);
    for (int sl = 1; sl < n_slices; ++sl) {
        Frag<NV> y;
        load_row_f32<G, NV, T, TMF_NT_FIN>(y, part, sl * n_users + u, g);
        add<NV>(acc, y);
    }
    row_epilogue<G, NV, T>(acc, U_old, U_out, u, g, epi, adam);
}

}  // namespace tmf

using namespace tmf;

template <typename T>
static int wmrb_user_pass_impl(const int64_t* rowptr, const int32_t* col, const float* val, const int32_t* R,
                               int32_t n_users, int32_t S, float c, const void* U_old, const void* V_old, void* U_out,
                               float* delta, float* D, float* loss_part, float* pos_part, int n_components, int epi,
                               tmf_adam adam, void* stream) {
    if (n_users == 0) return TMF_OK;
    TMF_REQUIRE(n_users > 0 && S > 0, "wmrb_user_pass: n_users=%d S=%d", n_users, S);
    TMF_REQUIRE(rowptr && R && U_old && V_old && U_out && D, "wmrb_user_pass: null pointer");
    TMF_REQUIRE(epi == TMF_EPI_ADAM || epi == TMF_EPI_GRAD, "wmrb_user_pass: bad epilogue %d", epi);
    const RowGeom geom = row_geom_of<T>(n_components);
#define CALL(G_, NV_)                                                                                                  \
    return launch_wmrb_user<G_, NV_, T>(rowptr, col, val, R, n_users, S, c, (const T*)U_old, (const T*)V_old, U_out, \
                                        delta, D, loss_part, pos_part, epi, adam, (hipStream_t)stream)
    TMF_DISPATCH(T, geom, CALL);
#undef CALL
    return TMF_OK;
}

extern "C" int tmf_wmrb_gradu4_supported(int n_components, int bf16) {
    const RowGeom geom = bf16 ? row_geom_bf16(n_components) : row_geom(n_components);
    return geom.ld != 0 && geom.G >= 8;
}

extern "C" int tmf_wmrb_user_pass_fits(int32_t S, int n_components) {
    const RowGeom geom = row_geom(n_components);
    return geom.ld != 0 && S > 0 && wmrb_user_lds(S, geom.ld) <= 160 * 1024;
}

extern "C" int tmf_wmrb_user_pass_f32(const int64_t* rowptr, const int32_t* col, const float* val,
                                      const int32_t* R, int32_t n_users, int32_t S, float c,
                                      const float* U_old, const float* V_old, float* U_out, float* delta,
                                      float* D, float* loss_part, float* pos_part,
                                      int n_components, int epi, tmf_adam adam, void* stream) {
    return wmrb_user_pass_impl<float>(rowptr, col, val, R, n_users, S, c, U_old, V_old, U_out, delta, D, loss_part,
                                      pos_part, n_components, epi, adam, stream);
}
extern "C" int tmf_wmrb_user_pass_bf16(const int64_t* rowptr, const int32_t* col, const float* val,
                                       const int32_t* R, int32_t n_users, int32_t S, float c,
                                       const void* U_old, const void* V_old, void* U_out, float* delta,
                                       float* D, float* loss_part, float* pos_part,
                                       int n_components, int epi, tmf_adam adam, void* stream) {
    return wmrb_user_pass_impl<__bf16>(rowptr, col, val, R, n_users, S, c, U_old, V_old, U_out, delta, D, loss_part,
                                       pos_part, n_components, epi, adam, stream);
}

// ---- sliced pass: storage-type generic implementations + the _f32 / _bf16 entry points ----
static int check_lists(const tmf_slice_lists* l, SliceLists& a, const char* what, int lanes_per_row, int waves, size_t row_bytes = 0) {
    TMF_REQUIRE(l != nullptr, "%s: lists is null", what);
    TMF_REQUIRE(l->n_users >= 0 && l->n_slices > 0 && l->n_samples > 0, "%s: n_users=%d n_slices=%d n_samples=%d", what,
                l->n_users, l->n_slices, l->n_samples);
    TMF_REQUIRE(l->n_users == 0 || (l->R_sorted && l->slice_off && l->rowptr && l->pos_off), "%s: null list array", what);
    // users per workgroup: 16 (one or two per lane group).  Small groups balance skewed users and leave no tail: against 128
    // per workgroup, same box - C4 99.2 vs 100.4 ms, config-5 shard 249.6 vs 254.7, 200K x 50K r=64 6.99 vs 7.40,
    // 20K x 200K r=128 2.61 vs 3.59, MovieLens-1M shape user pass 0.37 vs 1.07 ms.  TMF_SLICE_USERS overrides (A/B runs).
    const int lane_groups = (64 / (lanes_per_row > 0 ? lanes_per_row : 64)) * waves;   // narrow rows: many lane groups per workgroup
    int upg = kSliceUsers > lane_groups ? kSliceUsers : lane_groups;                    // at least one user for each of them
    if (const char* env = getenv("TMF_SLICE_USERS")) {
        const int v = atoi(env);
        if (v >= 16 && v <= 1024 && (v & (v - 1)) == 0) upg = v;
    }
    const int64_t groups = ((int64_t)l->n_users + upg - 1) / upg;
    const int sl0 = l->slice_begin, nsl = l->slice_count > 0 ? l->slice_count : l->n_slices - sl0;
    TMF_REQUIRE(sl0 >= 0 && nsl > 0 && sl0 + nsl <= l->n_slices && l->item_base >= 0, "%s: window [%d, +%d) of %d slices, item_base=%d",
                what, sl0, l->slice_count, l->n_slices, l->item_base);
    int xcd = (l->flags & TMF_SLICE_XCD_MAJOR) != 0;
    if (const char* env = getenv("TMF_SLICE_XCD")) xcd = env[0] == '1';   // A/B runs
    // the lean scores walk addresses V with 32-bit byte offsets: only when the caller says how many items there are (n_items = 0:
    // not stated) and they fit; TMF_LEAN=0 forces the general form (A/B runs, tests)
    int lean = (l->flags & TMF_SLICE_N_ITEMS_STATED) && l->n_items > 0 && row_bytes > 0 && (int64_t)l->n_items * (int64_t)row_bytes < ((int64_t)1 << 32);
    if (const char* env = getenv("TMF_LEAN")) lean = lean && env[0] != '0';
    a = SliceLists{l->R_sorted, l->slice_off, l->rowptr, l->col, l->pos_off, l->n_slices, l->n_samples, l->n_users, groups,
                   upg, sl0, nsl, l->item_base, xcd, lean};
    return TMF_OK;
}

// TMF_CHECK_IDS=1 (debug): every id of the lists really is below the n_items the caller stated - the lean walk addresses V with
// 32-bit offsets on the strength of that number.  Synchronous, allocates a flag: never on by default.
namespace tmf {
__global__ __launch_bounds__(256) void k_check_ids(const int32_t* __restrict__ R, int64_t n_neg, const int32_t* __restrict__ col,
                                                   const int64_t* __restrict__ rowptr, int64_t n_users, int n_items, int* __restrict__ bad) {
    const int64_t nnz = col ? rowptr[n_users] : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_neg + nnz; i += (int64_t)gridDim.x * blockDim.x) {
        const int id = i < n_neg ? R[i] : col[i - n_neg];
        if (id < 0 || id >= n_items) atomicAdd(bad, 1);
    }
}
}  // namespace tmf
static int check_ids_debug(const tmf_slice_lists* l, hipStream_t stream) {
    const char* env = getenv("TMF_CHECK_IDS");
    if (!(env && env[0] == '1') || !(l->flags & TMF_SLICE_N_ITEMS_STATED) || l->n_users == 0) return TMF_OK;
    int* bad = nullptr;
    int host = 0;
    if (hipMalloc(&bad, sizeof(int)) != hipSuccess || hipMemsetAsync(bad, 0, sizeof(int), stream) != hipSuccess) {
        set_error("TMF_CHECK_IDS: could not allocate the flag");
        return TMF_E_LAUNCH;
    }
    hipLaunchKernelGGL(tmf::k_check_ids, dim3(1024), dim3(256), 0, stream, l->R_sorted, (int64_t)l->n_users * l->n_samples, l->col,
                       l->rowptr, (int64_t)l->n_users, (int)l->n_items, bad);
    const bool ok = hipMemcpyAsync(&host, bad, sizeof(int), hipMemcpyDeviceToHost, stream) == hipSuccess &&
                    hipStreamSynchronize(stream) == hipSuccess;
    (void)hipFree(bad);
    if (!ok) { set_error("TMF_CHECK_IDS: the check itself failed"); return TMF_E_LAUNCH; }
    TMF_REQUIRE(host == 0, "tmf_slice_lists: %d ids of R_sorted / col are outside [0, n_items = %d)", host, l->n_items);
    return TMF_OK;
}

// blocks of a launch that covers all slices of `a` (plain order: slice-major; XCD-major: rounds of eight slices)
static unsigned slice_grid(const SliceLists& a) {
    return (unsigned)(a.xcd ? a.n_groups * 8 * ((a.nsl + 7) / 8) : a.n_groups * a.nsl);
}

static size_t slice_lds(const RowGeom& geom, int waves) {
    return (size_t)(64 / (geom.G ? geom.G : 1)) * waves * 2 * 8 * geom.G * sizeof(int);
}

// 8 waves per workgroup for long ranges, 4 otherwise (kLongRange); TMF_SLICE_WAVES = 4 | 8 forces one form (A/B runs, tests)
static int slice_waves(const tmf_slice_lists* l, int lanes_per_row) {
    if (const char* env = getenv("TMF_SLICE_WAVES")) {
        const int v = atoi(env);
        if (v == 4 || v == 8) return v;
    }
    // rows of 32 lanes and more: four waves are only 8 lane groups (narrower rows already have one group per user)
    return (l && l->n_slices > 0 && l->n_samples / l->n_slices >= kLongRange && lanes_per_row >= 32) ? 8 : 4;
}

template <typename T>
static int wmrb_scores3_impl(const tmf_slice_lists* lists, const void* U, const void* V, float* sp, float* p,
                             int n_components, void* stream) {
    SliceLists a;
    const RowGeom geom = row_geom_of<T>(n_components);
    const int waves = slice_waves(lists, geom.G);
    if (int rc = check_lists(lists, a, "wmrb_scores3", geom.G, waves, (size_t)geom.ld * sizeof(T))) return rc;
    if (a.n_users == 0) return TMF_OK;
    if (int rc = check_ids_debug(lists, (hipStream_t)stream)) return rc;
    TMF_REQUIRE(U && V && sp && (p || lists->col == nullptr), "wmrb_scores3: null pointer");
    const size_t lds = slice_lds(geom, waves);
    // one launch carries < 2^32 work-items: many slices x many user groups go out in several launches of whole slices
    // (XCD-major: whole rounds of eight)
    const int unit = a.xcd ? 8 : 1;
    const int64_t per_unit = a.n_groups * unit * 64 * waves;
    const int max_units = (int)((((int64_t)1 << 32) - 1) / per_unit);
    TMF_REQUIRE(max_units >= 1, "wmrb_scores3: %lld user groups exceed one launch", (long long)a.n_groups);
    const int first = a.sl0, last = a.sl0 + a.nsl;
    for (int s0 = first; s0 < last; s0 += max_units * unit) {
        a.sl0 = s0;
        a.nsl = (last - s0 < max_units * unit) ? last - s0 : max_units * unit;
#define CALLW(G_, NV_, W_)                                                                                             \
    hipLaunchKernelGGL((k_wmrb_scores3<G_, NV_, T, W_>), dim3(slice_grid(a)), dim3(64 * W_), lds, \
                       (hipStream_t)stream, a, (const T*)U, (const T*)V, sp, p)
#define CALL4(G_, NV_) CALLW(G_, NV_, 4)
#define CALL8(G_, NV_) CALLW(G_, NV_, 8)
        if (waves == 8) { TMF_DISPATCH(T, geom, CALL8); } else { TMF_DISPATCH(T, geom, CALL4); }
#undef CALL4
#undef CALL8
#undef CALLW
    }
    return check_launch("tmf_wmrb_scores3");
}

template <typename T>
static int wmrb_gradu3_impl(const tmf_slice_lists* lists, const float* D, const float* delta, const void* V, float* part,
                            int per_slice_launches, int n_components, void* stream) {
    SliceLists a;
    const RowGeom geom = row_geom_of<T>(n_components);
    const int waves = slice_waves(lists, geom.G);
    if (int rc = check_lists(lists, a, "wmrb_gradu3", geom.G, waves)) return rc;
    if (a.n_users == 0) return TMF_OK;
    TMF_REQUIRE(D && V && part && (delta || lists->col == nullptr), "wmrb_gradu3: null pointer");   // no interactions at all: no delta either
    const size_t lds = slice_lds(geom, waves);
    TMF_REQUIRE(per_slice_launches >= 0 && per_slice_launches <= 3, "wmrb_gradu3: per_slice_launches=%d", per_slice_launches);
    if (per_slice_launches == 3) {   // rounds of eight slices, one layer per XCD lane (see k_wmrb_gradu3)
        TMF_REQUIRE_LAUNCH(a.n_groups * 8, 64 * waves, "wmrb_gradu3 (rounds of eight slices)");
        for (int sl = a.sl0; sl < a.sl0 + a.nsl; sl += 8) {
            const int accumulate = (sl == a.sl0) ? 3 : 4;
#define CALLW(G_, NV_, W_)                                                                                                       \
    hipLaunchKernelGGL((k_wmrb_gradu3<G_, NV_, T, W_>), dim3((unsigned)(a.n_groups * 8)), dim3(64 * W_), lds, (hipStream_t)stream, a, \
                       (const T*)V, D, delta, part, sl, accumulate)
#define CALL4(G_, NV_) CALLW(G_, NV_, 4)
#define CALL8(G_, NV_) CALLW(G_, NV_, 8)
            if (waves == 8) { TMF_DISPATCH(T, geom, CALL8); } else { TMF_DISPATCH(T, geom, CALL4); }
#undef CALL4
#undef CALL8
#undef CALLW
        }
        return check_launch("tmf_wmrb_gradu3");
    }
    if (per_slice_launches) {
        TMF_REQUIRE_LAUNCH(a.n_groups, 64 * waves, "wmrb_gradu3 (one launch per slice)");
        for (int sl = a.sl0; sl < a.sl0 + a.nsl; ++sl) {
            const int accumulate = (sl == a.sl0 && per_slice_launches == 1) ? 1 : 2;
#define CALLW(G_, NV_, W_)                                                                                                   \
    hipLaunchKernelGGL((k_wmrb_gradu3<G_, NV_, T, W_>), dim3((unsigned)a.n_groups), dim3(64 * W_), lds, (hipStream_t)stream, a, \
                       (const T*)V, D, delta, part, sl, accumulate)
#define CALL4(G_, NV_) CALLW(G_, NV_, 4)
#define CALL8(G_, NV_) CALLW(G_, NV_, 8)
            if (waves == 8) { TMF_DISPATCH(T, geom, CALL8); } else { TMF_DISPATCH(T, geom, CALL4); }
#undef CALL4
#undef CALL8
#undef CALLW
        }
        return check_launch("tmf_wmrb_gradu3");
    }
    // several launches of whole slices when one would exceed 2^32 work-items (see wmrb_scores3_impl)
    const int unit = a.xcd ? 8 : 1;
    const int64_t per_unit = a.n_groups * unit * 64 * waves;
    const int max_units = (int)((((int64_t)1 << 32) - 1) / per_unit);
    TMF_REQUIRE(max_units >= 1, "wmrb_gradu3: %lld user groups exceed one launch", (long long)a.n_groups);
    const int first = a.sl0, last = a.sl0 + a.nsl;
    for (int s0 = first; s0 < last; s0 += max_units * unit) {
        a.sl0 = s0;
        a.nsl = (last - s0 < max_units * unit) ? last - s0 : max_units * unit;
#define CALLW(G_, NV_, W_)                                                                                   \
    hipLaunchKernelGGL((k_wmrb_gradu3<G_, NV_, T, W_>), dim3(slice_grid(a)), dim3(64 * W_), lds, \
                       (hipStream_t)stream, a, (const T*)V, D, delta, part, -1, 0)
#define CALL4(G_, NV_) CALLW(G_, NV_, 4)
#define CALL8(G_, NV_) CALLW(G_, NV_, 8)
        if (waves == 8) { TMF_DISPATCH(T, geom, CALL8); } else { TMF_DISPATCH(T, geom, CALL4); }
#undef CALL4
#undef CALL8
#undef CALLW
    }
    return check_launch("tmf_wmrb_gradu3");
}

static size_t gradu4_workspace(int64_t n_users, int n_slices, int64_t users_per_launch) {
    const int64_t launches = users_per_launch > 0 ? (n_users + users_per_launch - 1) / users_per_launch : 0;
    return rendezvous_bytes(launches, n_slices);
}

extern "C" size_t tmf_wmrb_gradu4_workspace_bytes(int32_t n_users, int32_t n_slices, int32_t users_per_launch) {
    return gradu4_workspace(n_users, n_slices, users_per_launch);
}

template <typename T>
static int wmrb_gradu4_impl(const tmf_slice_lists* lists, const float* D, const float* delta, const void* V, const void* U_old,
                            void* U_out, int n_components, int epi, tmf_adam adam, int32_t users_per_launch, void* workspace,
                            size_t workspace_bytes, void* stream) {
    SliceLists a;
    const RowGeom geom = row_geom_of<T>(n_components);
    constexpr int W4 = 8;
    if (int rc = check_lists(lists, a, "wmrb_gradu4", geom.G, W4)) return rc;
    if (a.n_users == 0) return TMF_OK;
    TMF_REQUIRE(D && V && U_out && (delta || lists->col == nullptr) && (epi == TMF_EPI_GRAD || U_old), "wmrb_gradu4: null pointer");
    TMF_REQUIRE(epi == TMF_EPI_ADAM || epi == TMF_EPI_GRAD, "wmrb_gradu4: bad epilogue %d", epi);
    TMF_REQUIRE(users_per_launch > 0, "wmrb_gradu4: users_per_launch=%d", users_per_launch);
    if (geom.G < 8) {   // the offsets cache holds G slices per refill
        set_error("wmrb_gradu4: rows of %d lanes are too narrow for the row-stationary form; use tmf_wmrb_gradu3 + tmf_wmrb_finish", geom.G);
        return TMF_E_UNSUPPORTED;
    }
#ifndef TMF_G4_K
#define TMF_G4_K 4
#endif
    constexpr int K4 = TMF_G4_K;   // users per lane group (A/B builds)
    const int cap = geom.G < 16 ? geom.G : 16;
    const size_t lds = (size_t)(64 / geom.G) * W4 * 2 * K4 * 2 * cap * sizeof(int);
    const int64_t per_block = (int64_t)(64 / geom.G) * W4 * K4;
    // per-launch rendezvous counters (optional): zeroed here, one int per (launch, slice)
    int* sync = nullptr;
    if (workspace != nullptr) {
        const size_t need = gradu4_workspace(a.n_users, a.nsl, users_per_launch);
        TMF_REQUIRE(workspace_bytes >= need, "wmrb_gradu4: workspace of %zu bytes, %zu needed", workspace_bytes, need);
        if (hipMemsetAsync(workspace, 0, need, (hipStream_t)stream) != hipSuccess) { set_error("wmrb_gradu4: hipMemsetAsync failed"); return TMF_E_LAUNCH; }
        sync = static_cast<int*>(workspace);
    }
    int lag = 1;
    if (const char* env = getenv("TMF_G4_LAG")) lag = atoi(env);
    if (getenv("TMF_DEBUG")) {
        int nb = -1, dev = 0, cus = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
#define CALL(G_, NV_) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_wmrb_gradu4<G_, NV_, T, K4, W4>, 64 * W4, lds)
        TMF_DISPATCH(T, geom, CALL);
#undef CALL
        fprintf(stderr, "[tmf] gradu4: %d workgroups per CU x %d CUs resident, %lld users per workgroup, lds %zu, lag %d\n", nb, cus,
                (long long)per_block, lds, lag);
    }
    for (int64_t b = 0, launch = 0; b < a.n_users; b += users_per_launch, ++launch) {
        const int64_t e = (b + users_per_launch < a.n_users) ? b + users_per_launch : a.n_users;
        const unsigned blocks = (unsigned)((e - b + per_block - 1) / per_block);
        int* sy = (sync && lag >= 0) ? sync + launch * a.nsl * 8 * kSyncStride : nullptr;
#define CALL(G_, NV_)                                                                                                         \
    hipLaunchKernelGGL((k_wmrb_gradu4<G_, NV_, T, K4, W4>), dim3(blocks), dim3(64 * W4), lds, (hipStream_t)stream, a, (const T*)V, \
                       D, delta, (const T*)U_old, U_out, epi, adam, b, e, sy, lag)
        TMF_DISPATCH(T, geom, CALL);
#undef CALL
    }
    return check_launch("tmf_wmrb_gradu4");
}


// ---------------------------------------------------------------------------------------------
// Row-stationary scores ("scores5", round 4).  Built for catalogs far beyond the L2s (config 5: V = 512 MB), where a slice
// small enough for an L2 leaves a (user, slice) visit a handful of rows and k_wmrb_scores3 pays for every visit with the
// user's own row (512 bytes from HBM), two offsets and a 17-byte piece of the id list - it ends up bound by the fabric at
// 34 x its compulsory traffic (profiles/r03_c5_pmc.json).  Here a workgroup OWNS `UB` users, keeps their rows in LDS for the
// whole launch (UB x row bytes: 128 KB of the CU's 160) and walks ONE flat stream of entries prepared once per fit
// (_engine.Scores5Plan): all (user, item) pairs of its users - negatives and interactions alike - ordered by item slice,
// eight per step, each a packed (local user << 24 | item) id and the place its score goes to.  No offsets, no per-visit
// reads, no user row from memory after the first: what is left is the row gathers themselves - and since every workgroup of
// a launch walks the catalog front to back at the same pace, the rows they gather at any moment come from a window of a
// few MB that the L2s hold.  A launch covers as many workgroups as are resident together (one per CU); the host launches
// generation after generation.
//   step = 8 consecutive entries of the workgroup's stream; lane group gid takes steps gid, gid + NGB, ...
//   ids   : the 8 ids of a step in two 16-byte loads that every lane of the group issues to the same address (one request),
//           fetched one step ahead of their use
//   x     : the entry's user row from LDS (ds_read_b128, conflict-free: 32 lanes x 16 bytes = the row)
//   score : dot_raw + Reduce8x32 (tmf_common.h); the four lanes that end up with entry e hold its score, one of them stores
//           it where out[] says: >= 0 -> sp[out], < 0 -> p[~out], INT_MIN -> a padding entry (the stream of a workgroup is
//           padded to whole steps with a valid row id)
// Scores agree with k_wmrb_scores3 to rounding (another summation tree), exactly on dyadic data.
// ---------------------------------------------------------------------------------------------
namespace tmf {
constexpr int kS5Users = 256;        // users per workgroup (8-bit local user in the packed id)
constexpr int kS5Waves = 15;         // worker waves (+ 1 pacer wave = 1024 threads, the largest workgroup): one workgroup per CU
constexpr int kS5Pad = INT32_MIN;    // out[] of a padding entry
constexpr int kS5MaxWindows = 4096;  // pace windows of a stream (their first steps sit in LDS)
constexpr int kS5Stride = 16;        // ints between two progress counters (a 64-byte line each)

// Pacing (speed only - no data is handed over, so no fences; every wait is bounded and the kernel computes the same bits with or
// without it).  The stream of a workgroup is cut into `nw` windows of the catalog (wst[w] = its first step of window w); the
// workgroups of a launch that share an XCD (blocks b, b + 8, ... under the observed round-robin placement) walk them together:
// a workgroup may work on window w only when all of them have COMPLETED window w - lag - 1, so the rows being gathered on that XCD
// at any time span lag + 1 windows, which its L2 holds.  Worker waves note the window they are in in LDS (s_win) and wait on an
// LDS gate; the 16th wave of the workgroup is the pacer: it publishes the workgroup's completed windows (one relaxed agent-scope
// add per window to cnt[x][w]) and moves the gate when the counter it waits for reaches the number of peers.  If nothing moves
// for ~1 ms (a workgroup of the launch is not resident) the pacer opens the gate for good.
struct S5Pace {
    const int32_t* wst;   // [n_wg, nw + 1] first step of every window of every workgroup (wst[nw] = steps of the workgroup)
    int* cnt;             // [8, nw] x kS5Stride progress counters of this launch, zero at launch
    int nw, lag;
};

template <int NV, typename T>
__global__ __launch_bounds__(64 * (kS5Waves + 1)) void k_wmrb_scores5(const int32_t* __restrict__ ids, const int32_t* __restrict__ outs,
                                                                     const int64_t* __restrict__ wg_ptr, int64_t wg0, int64_t n_users,
                                                                     const T* __restrict__ U, const T* __restrict__ V,
                                                                     float* __restrict__ sp, float* __restrict__ p, S5Pace pace) {
    constexpr int G = 32, NGB = 2 * kS5Waves;
    constexpr uint32_t RB = RowBytes<G, NV, T>::value;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];   // [kS5Users] rows as stored | [nw + 1] window starts
    __shared__ int s_win[kS5Waves];
    __shared__ int s_gate;
    int* wst = reinterpret_cast<int*>(smem_raw + (size_t)kS5Users * RB);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane & (G - 1), gid = tid / G;
    const int64_t wg = wg0 + blockIdx.x;
    const int64_t ubeg = wg * kS5Users;
    const int nu = (int)((n_users - ubeg < kS5Users) ? n_users - ubeg : kS5Users);
    const bool paced = pace.cnt != nullptr;
    // the workgroup's rows -> LDS, 16 bytes per lane (rows of U are contiguous: one flat copy)
    {
        const tmf_f4* src = reinterpret_cast<const tmf_f4*>(reinterpret_cast<const char*>(U) + ubeg * (int64_t)RB);
        tmf_f4* dst = reinterpret_cast<tmf_f4*>(smem_raw);
        for (int i = tid; i < nu * (int)(RB / 16); i += 64 * (kS5Waves + 1)) dst[i] = __builtin_nontemporal_load(src + i);
        if (paced)
            for (int i = tid; i <= pace.nw; i += 64 * (kS5Waves + 1)) wst[i] = pace.wst[wg * (pace.nw + 1) + i];
        if (tid < kS5Waves) s_win[tid] = 0;
        if (tid == 0) s_gate = paced ? pace.lag : INT32_MAX;
    }
    __syncthreads();
    if (wave == kS5Waves) {   // ---- the pacer ----
        if (!paced) return;
        const int x = blockIdx.x & 7, peers = ((int)gridDim.x - x + 7) / 8;
        int* cnt = pace.cnt + (size_t)x * pace.nw * kS5Stride;
        int published = 0, gate = pace.lag, idle = 0;
        bool open = false;
        for (;;) {
            int mn = (lane < kS5Waves) ? __hip_atomic_load(&s_win[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : INT32_MAX;
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) mn = min(mn, __shfl_xor(mn, o, 64));   // lanes 0..15 hold the workers' windows (lane 15: none)
            mn = __builtin_amdgcn_readfirstlane(mn);
            const bool done = mn == INT32_MAX;
            const int target = done ? pace.nw : mn;       // every wave has left the windows below
            if (lane == 0)
                for (int w = published; w < target; ++w) __hip_atomic_fetch_add(cnt + (size_t)w * kS5Stride, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (target > published) published = target;
            if (done) return;
            if (!open && gate < pace.nw - 1) {
                int c = (lane == 0) ? __hip_atomic_load(cnt + (size_t)(gate - pace.lag) * kS5Stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                c = __builtin_amdgcn_readfirstlane(c);
                if (c >= peers) {
                    ++gate;
                    idle = 0;
                    if (lane == 0) __hip_atomic_store(&s_gate, gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    continue;
                }
                if (++idle > 4000) {   // ~1 ms without progress: stop pacing rather than wait for a workgroup that is not there
                    open = true;
                    if (lane == 0) __hip_atomic_store(&s_gate, INT32_MAX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    const int64_t beg = wg_ptr[wg], end = wg_ptr[wg + 1];   // multiples of 8 (padded)
    const int steps = (int)((end - beg) >> 3);
    const uint32_t loff = 16u * (uint32_t)g;
    const int my_entry = Reduce8x32::entry_of_lane(lane);
    const bool owner = Reduce8x32::owner(lane);
    const int4* id4 = reinterpret_cast<const int4*>(ids + beg);   // 32-byte aligned steps
    const int32_t* out8 = outs + beg + my_entry;
    // Step i of this lane group is gid + NGB i; the wave runs as many rounds as its first group has steps (the second has as
    // many or one fewer: beyond its last step a group repeats it and stores nothing), so every load below is unconditional -
    // behind a divergent branch the compiler waits for a load at the end of the branch.
    const int mine = steps > gid ? (steps - gid + NGB - 1) / NGB : 0;
    const int rounds = __builtin_amdgcn_readfirstlane(mine);
    if (rounds > 0) {
        const int last = mine > 0 ? gid + NGB * (mine - 1) : 0;
        struct Ids {
            int4 a, b;
            int o;
        };
        auto fetch = [&](int i) {   // ids and place of round i (8 ids: every lane of the group reads the same 32 bytes)
            const int st = gid + NGB * i;
            const int sc = st <= last ? st : last;
            Ids r{id4[2 * sc], id4[2 * sc + 1], out8[8 * sc]};
            if (i >= mine) r.o = kS5Pad;
            return r;
        };
        int cur_w = 0, next_b = paced ? wst[1] : INT32_MAX;   // the wave's window and the first step of the next one
        auto pace_to = [&](int i) {   // before the rows of round i are asked for: is the wave allowed into their window?
            // Rounds past the end (the prefetch of the loop below asks for round `rounds`) repeat the last step and are never
            // paced, and a wave in the last window has nothing left to wait for.  (Round 4 paced them: st_w >= wst[nw] moved
            // cur_w to nw, read wst[nw + 1] - one int past the array - and spun the full 20000 x s_sleep on a gate that stops at
            // nw - 1: ~1.7 ms at the end of EVERY launch, 20 launches per epoch.  ADVICE r04.)
            if (!paced || i >= rounds || cur_w >= pace.nw - 1) return;
            const int st_w = __builtin_amdgcn_readfirstlane(gid + NGB * i);   // the wave's first group decides for both
            if (st_w < next_b) return;
            do {
                ++cur_w;
                next_b = wst[cur_w + 1];
            } while (st_w >= next_b && cur_w < pace.nw - 1);
            if (lane == 0) __hip_atomic_store(&s_win[wave], cur_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (int spin = 0; spin < 20000; ++spin) {   // bounded: ~2 ms
                if (__hip_atomic_load(&s_gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= cur_w) break;
                __builtin_amdgcn_s_sleep(2);
            }
        };
        auto gather = [&](Raw<NV, T> (&y)[8], const Ids& d) {
            const int idv[8] = {d.a.x, d.a.y, d.a.z, d.a.w, d.b.x, d.b.y, d.b.z, d.b.w};
#pragma unroll
            for (int t = 0; t < 8; ++t) load_raw32<G, NV>(y[t], V, (uint32_t)idv[t] & 0xffffffu, loff);
        };
        auto finish = [&](const Raw<NV, T> (&y)[8], const Ids& d) {
            const int idv[8] = {d.a.x, d.a.y, d.a.z, d.a.w, d.b.x, d.b.y, d.b.z, d.b.w};
            float pr[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                Raw<NV, T> x;   // the entry's user row, from LDS
                const char* xr = smem_raw + ((uint32_t)idv[t] >> 24) * RB + loff;
                if constexpr (std::is_same<T, float>::value) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) x.v[v] = *reinterpret_cast<const float4*>(xr + 16 * G * v);
                } else {
#pragma unroll
                    for (int pv = 0; pv < NV / 2; ++pv) x.v[pv] = *reinterpret_cast<const bf16x8*>(xr + 16 * G * pv);
                }
                pr[t] = dot_raw<NV>(x, y[t]);
            }
            const float sc = Reduce8x32::run(pr, lane);
            if (owner && d.o != kS5Pad) __builtin_nontemporal_store(sc, d.o >= 0 ? sp + d.o : p + ~d.o);
        };
        // Two rounds in flight per lane group: while the rows of round i are reduced, those of round i + 1 are on their way.
        // Loads return in order (vmcnt), so a wait is for everything OLDER as well: the ids of round i + 2 are therefore asked for
        // BEFORE the rows of round i + 1 (waiting for ids then never waits for rows issued after them, and the stream read - it
        // comes from HBM - has a whole round to arrive before anything behind it is waited for).
        Raw<NV, T> yA[8], yB[8];
        Ids A = fetch(0);
        Ids B = fetch(1);
        pace_to(0);
        gather(yA, A);
        for (int i = 0; i < rounds; i += 2) {
            Ids C = fetch(i + 2);
            __builtin_amdgcn_sched_barrier(0);
            pace_to(i + 1);
            gather(yB, B);
            __builtin_amdgcn_sched_barrier(0);
            finish(yA, A);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 >= rounds) break;
            Ids Dn = fetch(i + 3);
            __builtin_amdgcn_sched_barrier(0);
            pace_to(i + 2);
            gather(yA, C);
            __builtin_amdgcn_sched_barrier(0);
            finish(yB, B);
            __builtin_amdgcn_sched_barrier(0);
            A = C;
            B = Dn;
        }
    }
    if (lane == 0) __hip_atomic_store(&s_win[wave], INT32_MAX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace tmf

// ---------------------------------------------------------------------------------------------
// Flat streams on the slice-major grid ("scores6", round 5).  k_wmrb_scores3 pays for every (user, slice)
// visit with three dependent round trips - offsets -> ids -> rows - and at the config-5 shard a visit is 9 rows at 4 MB slices:
// with 128 slices it runs 69.5 ms against 59.5 with 64 although its fabric traffic falls from 437 to 256 GB
// (profiles/r05_c5_shard.txt).  Here a workgroup takes one (slice, group of kS6Users users) CHUNK: the group's rows go to LDS
// once (as in scores5), and the chunk's entries - interactions, then negatives, each by user and item - are ONE contiguous piece
// of a stream prepared once per fit (_engine.Scores6Plan): packed (local user << 24 | item) ids and the place every score goes to.
// One read of the chunk's bounds, then ids and rows: no per-user offsets, no per-visit row of U.  Inside a chunk the entries of a
// user are consecutive, so the eight scores of a step mostly leave as one 32-byte piece of sp / p.
// The grid is scores3's: slice-major, every resident workgroup gathers from the same ~4 MB slice of V - no pacing, no rendezvous.
// Scores agree with k_wmrb_scores3 to rounding (the summation tree of scores5), exactly on dyadic data.
// ---------------------------------------------------------------------------------------------
namespace tmf {
// Config-5 shard, scores ms (profiles/r05_c5_shard.txt item 2): TWO steps of 8 gathers in flight per lane group (110 VGPRs, 4 waves per
// SIMD): 32 users / 2 waves 55.4 (4 MB slices), 52.7 (6 MB); ONE step in flight (68 VGPRs, 7 waves per SIMD): 32 / 2: 51.0, 32 / 4: 50.0,
// 32 / 8: 50.1, 64 / 4: 50.2, 64 / 8: 49.6, 16 / 4: 50.6, 128 / 8: 49.4, 128 / 16: 50.5 - more waves beat more rows per wave.
#ifndef TMF_S6_USERS
#define TMF_S6_USERS 64
#endif
#ifndef TMF_S6_WAVES
#define TMF_S6_WAVES 8
#endif
#ifndef TMF_S6_ROUNDS
#define TMF_S6_ROUNDS 1
#endif
constexpr int kS6Users = TMF_S6_USERS;    // users per chunk (their rows: 32 KB of LDS at 512-byte rows); at most 256 (8-bit local user)
constexpr int kS6Waves = TMF_S6_WAVES;    // 16 lane groups: a chunk of ~580 entries is ~4.5 steps of 8 for each

template <int NV, typename T>
__global__ __launch_bounds__(64 * kS6Waves) void k_wmrb_scores6(const int32_t* __restrict__ ids, const int32_t* __restrict__ outs,
                                                                const int64_t* __restrict__ chunk_ptr, int64_t chunk0, int64_t n_groups,
                                                                int64_t n_users, const T* __restrict__ U, const T* __restrict__ V,
                                                                float* __restrict__ sp, float* __restrict__ p) {
    constexpr int G = 32, NGB = 2 * kS6Waves;
    constexpr uint32_t RB = RowBytes<G, NV, T>::value;
    __shared__ __attribute__((aligned(16))) char smem_raw[kS6Users * RB];
    const int tid = threadIdx.x, lane = tid & 63, g = lane & (G - 1), gid = tid / G;
    const int64_t chunk = chunk0 + blockIdx.x;            // = slice * n_groups + group
    const int64_t beg = chunk_ptr[chunk], end = chunk_ptr[chunk + 1];   // multiples of 8 (padded)
    if (beg == end) return;                                // no entry of this group in this slice (whole workgroup)
    const int64_t ubeg = (chunk % n_groups) * kS6Users;
    const int nu = (int)((n_users - ubeg < kS6Users) ? n_users - ubeg : kS6Users);
    {
        const tmf_f4* src = reinterpret_cast<const tmf_f4*>(reinterpret_cast<const char*>(U) + ubeg * (int64_t)RB);
        tmf_f4* dst = reinterpret_cast<tmf_f4*>(smem_raw);
        for (int i = tid; i < nu * (int)(RB / 16); i += 64 * kS6Waves) dst[i] = __builtin_nontemporal_load(src + i);
    }
    __syncthreads();
    const int steps = (int)((end - beg) >> 3);
    const uint32_t loff = 16u * (uint32_t)g;
    const int my_entry = Reduce8x32::entry_of_lane(lane);
    const bool owner = Reduce8x32::owner(lane);
    const int4* id4 = reinterpret_cast<const int4*>(ids + beg);   // 32-byte aligned steps
    const int32_t* out8 = outs + beg + my_entry;
    // step i of this lane group is gid + NGB i; the wave runs as many rounds as its first group has steps (the second has as many
    // or one fewer: beyond its last step a group repeats it and stores nothing), so every load below is unconditional
    const int mine = steps > gid ? (steps - gid + NGB - 1) / NGB : 0;
    const int rounds = __builtin_amdgcn_readfirstlane(mine);
    if (rounds == 0) return;
    const int last = mine > 0 ? gid + NGB * (mine - 1) : 0;
    struct Ids {
        int4 a, b;
        int o;
    };
    auto fetch = [&](int i) {
        const int st = gid + NGB * i;
        const int sc = st <= last ? st : last;
        Ids r{id4[2 * sc], id4[2 * sc + 1], out8[8 * sc]};
        if (i >= mine) r.o = kS5Pad;
        return r;
    };
    auto gather = [&](Raw<NV, T> (&y)[8], const Ids& d) {
        const int idv[8] = {d.a.x, d.a.y, d.a.z, d.a.w, d.b.x, d.b.y, d.b.z, d.b.w};
#pragma unroll
        for (int t = 0; t < 8; ++t) load_raw32<G, NV>(y[t], V, (uint32_t)idv[t] & 0xffffffu, loff);
    };
    auto finish = [&](const Raw<NV, T> (&y)[8], const Ids& d) {
        const int idv[8] = {d.a.x, d.a.y, d.a.z, d.a.w, d.b.x, d.b.y, d.b.z, d.b.w};
        float pr[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            Raw<NV, T> x;   // the entry's user row, from LDS
            const char* xr = smem_raw + ((uint32_t)idv[t] >> 24) * RB + loff;
            if constexpr (std::is_same<T, float>::value) {
#pragma unroll
                for (int v = 0; v < NV; ++v) x.v[v] = *reinterpret_cast<const float4*>(xr + 16 * G * v);
            } else {
#pragma unroll
                for (int pv = 0; pv < NV / 2; ++pv) x.v[pv] = *reinterpret_cast<const bf16x8*>(xr + 16 * G * pv);
            }
            pr[t] = dot_raw<NV>(x, y[t]);
        }
        const float sc = Reduce8x32::run(pr, lane);
        if (owner && d.o != kS5Pad) __builtin_nontemporal_store(sc, d.o >= 0 ? sp + d.o : p + ~d.o);
    };
#if TMF_S6_ROUNDS == 1
    // one step of 8 gathers in flight per lane group: 68 VGPRs, seven waves per SIMD
    {
        Raw<NV, T> yA[8];
        Ids A = fetch(0);
        for (int i = 0; i < rounds; ++i) {
            Ids B = fetch(i + 1);
            __builtin_amdgcn_sched_barrier(0);
            gather(yA, A);
            __builtin_amdgcn_sched_barrier(0);
            finish(yA, A);
            A = B;
        }
        return;
    }
#endif
    // two rounds in flight per lane group, the ids of round i + 2 asked for BEFORE the rows of round i + 1 (as in scores5)
    Raw<NV, T> yA[8], yB[8];
    Ids A = fetch(0);
    Ids B = fetch(1);
    gather(yA, A);
    for (int i = 0; i < rounds; i += 2) {
        Ids C = fetch(i + 2);
        __builtin_amdgcn_sched_barrier(0);
        gather(yB, B);
        __builtin_amdgcn_sched_barrier(0);
        finish(yA, A);
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 >= rounds) break;
        Ids Dn = fetch(i + 3);
        __builtin_amdgcn_sched_barrier(0);
        gather(yA, C);
        __builtin_amdgcn_sched_barrier(0);
        finish(yB, B);
        __builtin_amdgcn_sched_barrier(0);
        A = C;
        B = Dn;
    }
}

}  // namespace tmf

static bool s6_supported(int n_components, int bf16, int64_t n_items) {
    const tmf::RowGeom geom = bf16 ? tmf::row_geom_bf16(n_components) : tmf::row_geom(n_components);
    const int64_t row_bytes = (int64_t)geom.ld * (bf16 ? 2 : 4);
    return geom.G == 32 && n_items > 0 && n_items < (1 << 24) && n_items * row_bytes < ((int64_t)1 << 32);
}

template <typename T>
static int wmrb_scores6_impl(const int32_t* ids, const int32_t* outs, const int64_t* chunk_ptr, int64_t n_groups, int32_t n_slices,
                             int64_t n_users, int64_t n_items, const void* U, const void* V, float* sp, float* p, int n_components,
                             void* stream) {
    if (n_users == 0 || n_slices == 0) return TMF_OK;
    const RowGeom geom = row_geom_of<T>(n_components);
    TMF_REQUIRE(ids && outs && chunk_ptr && U && V && sp && p, "wmrb_scores6: null pointer");
    TMF_REQUIRE(n_groups == (n_users + kS6Users - 1) / kS6Users && n_slices > 0, "wmrb_scores6: %lld groups for %lld users (%d per group), %d slices",
                (long long)n_groups, (long long)n_users, kS6Users, n_slices);
    if (!s6_supported(n_components, std::is_same<T, __bf16>::value, n_items)) {
        set_error("wmrb_scores6: needs rows of 32 lanes (fp32 65..128 / bf16 129..256 components) and fewer than 2^24 items in a table "
                  "below 4 GB (got %d components, %lld items)", n_components, (long long)n_items);
        return TMF_E_UNSUPPORTED;
    }
    // whole slices per launch, fewer than 2^32 work-items each
    const int64_t per_slice = n_groups * 64 * kS6Waves;
    const int max_slices = (int)((((int64_t)1 << 32) - 1) / per_slice);
    TMF_REQUIRE(max_slices >= 1, "wmrb_scores6: %lld user groups exceed one launch", (long long)n_groups);
    for (int s0 = 0; s0 < n_slices; s0 += max_slices) {
        const int nsl = (n_slices - s0 < max_slices) ? n_slices - s0 : max_slices;
        const unsigned blocks = (unsigned)(n_groups * nsl);
#define CALL(G_, NV_)                                                                                                       \
    {                                                                                                                       \
        if constexpr (G_ == 32) {                                                                                           \
            hipLaunchKernelGGL((k_wmrb_scores6<NV_, T>), dim3(blocks), dim3(64 * kS6Waves), 0, (hipStream_t)stream, ids, outs, \
                               chunk_ptr, (int64_t)s0 * n_groups, n_groups, n_users, (const T*)U, (const T*)V, sp, p);      \
        }                                                                                                                   \
    }
        TMF_DISPATCH(T, geom, CALL);
#undef CALL
    }
    return check_launch("tmf_wmrb_scores6");
}

static int s5_launches(int64_t n_wg, int wgs_per_launch) { return (int)((n_wg + wgs_per_launch - 1) / wgs_per_launch); }
static int s5_wgs_per_launch(int wgs_per_launch) {
    if (wgs_per_launch > 0) return wgs_per_launch;
    int dev = 0, cus = 256;   // one workgroup per CU: every workgroup of a launch resident, all walking the catalog together
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus > 0 ? cus : 256;
}

template <typename T>
static int wmrb_scores5_impl(const int32_t* ids, const int32_t* outs, const int64_t* wg_ptr, int64_t n_wg, int64_t n_users,
                             int64_t n_items, const void* U, const void* V, float* sp, float* p, int n_components,
                             int wgs_per_launch, const int32_t* wstart, int32_t n_windows, int lag, void* workspace,
                             size_t workspace_bytes, void* stream) {
    if (n_wg == 0 || n_users == 0) return TMF_OK;
    const RowGeom geom = row_geom_of<T>(n_components);
    TMF_REQUIRE(ids && outs && wg_ptr && U && V && sp && p, "wmrb_scores5: null pointer");
    TMF_REQUIRE(n_wg == (n_users + kS5Users - 1) / kS5Users, "wmrb_scores5: %lld workgroups for %lld users (%d per workgroup)",
                (long long)n_wg, (long long)n_users, kS5Users);
    const size_t row_bytes = (size_t)geom.ld * sizeof(T);
    if (geom.G != 32 || row_bytes * kS5Users > 128 * 1024 || n_items <= 0 || n_items >= (1 << 24) ||
        (int64_t)n_items * (int64_t)row_bytes >= ((int64_t)1 << 32)) {
        set_error("wmrb_scores5: needs rows of 32 lanes (fp32 65..128 / bf16 129..256 components) and fewer than 2^24 items "
                  "in a table below 4 GB (got %d components, %lld items)", n_components, (long long)n_items);
        return TMF_E_UNSUPPORTED;
    }
    wgs_per_launch = s5_wgs_per_launch(wgs_per_launch);
    const int launches = s5_launches(n_wg, wgs_per_launch);
    const bool paced = wstart != nullptr && workspace != nullptr && n_windows > 1;
    const size_t per_launch = (size_t)8 * (paced ? n_windows : 0) * kS5Stride * sizeof(int);
    if (paced) {
        TMF_REQUIRE(n_windows <= kS5MaxWindows && lag >= 0, "wmrb_scores5: %d windows (at most %d), lag %d", n_windows, kS5MaxWindows, lag);
        TMF_REQUIRE(workspace_bytes >= per_launch * launches, "wmrb_scores5: workspace of %zu bytes needed, got %zu", per_launch * launches,
                    workspace_bytes);
        if (hipMemsetAsync(workspace, 0, per_launch * launches, (hipStream_t)stream) != hipSuccess) {
            set_error("wmrb_scores5: hipMemsetAsync failed");
            return TMF_E_LAUNCH;
        }
    }
    const size_t lds = row_bytes * kS5Users + (paced ? (size_t)(n_windows + 1) * sizeof(int) : 0);
#define CALL(G_, NV_)                                                                                                       \
    {                                                                                                                       \
        if constexpr (G_ == 32) {                                                                                           \
            static LdsGrant grant;                                                                                          \
            if (int rc = grant_dynamic_lds(reinterpret_cast<const void*>(&k_wmrb_scores5<NV_, T>), lds, grant)) return rc;  \
            for (int l = 0; l < launches; ++l) {                                                                            \
                const int64_t w0 = (int64_t)l * wgs_per_launch;                                                             \
                const unsigned blocks = (unsigned)((n_wg - w0 < wgs_per_launch) ? n_wg - w0 : wgs_per_launch);              \
                const S5Pace pace{wstart, paced ? reinterpret_cast<int*>(static_cast<char*>(workspace) + per_launch * l) : nullptr, \
                                  n_windows, lag};                                                                          \
                hipLaunchKernelGGL((k_wmrb_scores5<NV_, T>), dim3(blocks), dim3(64 * (kS5Waves + 1)), lds, (hipStream_t)stream, \
                                   ids, outs, wg_ptr, w0, n_users, (const T*)U, (const T*)V, sp, p, pace);                  \
            }                                                                                                               \
        }                                                                                                                   \
    }
    TMF_DISPATCH(T, geom, CALL);
#undef CALL
    return check_launch("tmf_wmrb_scores5");
}

template <typename T>
static int wmrb_finish_impl(const float* part, int32_t n_slices, int32_t n_users, const void* U_old, void* U_out,
                            int n_components, int epi, tmf_adam adam, void* stream) {
    if (n_users == 0) return TMF_OK;
    TMF_REQUIRE(part && U_out && n_slices > 0 && (epi == TMF_EPI_GRAD || U_old), "wmrb_finish: bad arguments");
    TMF_REQUIRE(epi == TMF_EPI_ADAM || epi == TMF_EPI_GRAD, "wmrb_finish: bad epilogue %d", epi);
    const RowGeom geom = row_geom_of<T>(n_components);
#define CALL(G_, NV_)                                                                                                  \
    {                                                                                                                  \
        constexpr int per_block = (64 / G_) * kWaves;                                                                  \
        hipLaunchKernelGGL((k_wmrb_finish<G_, NV_, T>), dim3((unsigned)(((int64_t)n_users + per_block - 1) / per_block)), \
                           dim3(kThreads), 0, (hipStream_t)stream, part, (int)n_slices, (int64_t)n_users,                \
                           (const T*)U_old, U_out, epi, adam);                                                         \
    }
    TMF_DISPATCH(T, geom, CALL);
#undef CALL
    return check_launch("tmf_wmrb_finish");
}

#define TMF_SLICED_ENTRY_POINTS(SFX, T_)                                                                                  \
    extern "C" int tmf_wmrb_scores3_##SFX(const tmf_slice_lists* lists, const void* U, const void* V, float* sp, float* p, \
                                          int n_components, void* stream) {                                               \
        return wmrb_scores3_impl<T_>(lists, U, V, sp, p, n_components, stream);                                           \
    }                                                                                                                     \
    extern "C" int tmf_wmrb_scores5_##SFX(const int32_t* ids, const int32_t* outs, const int64_t* wg_ptr, int64_t n_wg,    \
                                          int64_t n_users, int64_t n_items, const void* U, const void* V, float* sp,      \
                                          float* p, int n_components, int wgs_per_launch, const int32_t* wstart,          \
                                          int32_t n_windows, int lag, void* workspace, size_t workspace_bytes,            \
                                          void* stream) {                                                                 \
        return wmrb_scores5_impl<T_>(ids, outs, wg_ptr, n_wg, n_users, n_items, U, V, sp, p, n_components, wgs_per_launch, \
                                     wstart, n_windows, lag, workspace, workspace_bytes, stream);                         \
    }                                                                                                                     \
    extern "C" int tmf_wmrb_scores6_##SFX(const int32_t* ids, const int32_t* outs, const int64_t* chunk_ptr, int64_t n_groups, \
                                          int32_t n_slices, int64_t n_users, int64_t n_items, const void* U, const void* V, \
                                          float* sp, float* p, int n_components, void* stream) {                          \
        return wmrb_scores6_impl<T_>(ids, outs, chunk_ptr, n_groups, n_slices, n_users, n_items, U, V, sp, p, n_components, stream); \
    }                                                                                                                     \
    extern "C" int tmf_wmrb_gradu3_##SFX(const tmf_slice_lists* lists, const float* D, const float* delta,                \
                                         const void* V, float* part, int per_slice_launches, int n_components,            \
                                         void* stream) {                                                                  \
        return wmrb_gradu3_impl<T_>(lists, D, delta, V, part, per_slice_launches, n_components, stream);                  \
    }                                                                                                                     \
    extern "C" int tmf_wmrb_gradu4_##SFX(const tmf_slice_lists* lists, const float* D, const float* delta, const void* V,  \
                                         const void* U_old, void* U_out, int n_components, int epi, tmf_adam adam,       \
                                         int32_t users_per_launch, void* workspace, size_t workspace_bytes, void* stream) { \
        return wmrb_gradu4_impl<T_>(lists, D, delta, V, U_old, U_out, n_components, epi, adam, users_per_launch, workspace, \
                                    workspace_bytes, stream);                                                            \
    }                                                                                                                     \
    extern "C" int tmf_wmrb_finish_##SFX(const float* part, int32_t n_slices, int32_t n_users, const void* U_old,         \
                                         void* U_out, int n_components, int epi, tmf_adam adam, void* stream) {           \
        return wmrb_finish_impl<T_>(part, n_slices, n_users, U_old, U_out, n_components, epi, adam, stream);               \
    }

TMF_SLICED_ENTRY_POINTS(f32, float)
TMF_SLICED_ENTRY_POINTS(bf16, __bf16)

extern "C" int tmf_wmrb_scores6_users_per_group(void) { return tmf::kS6Users; }
extern "C" int tmf_wmrb_scores6_supported(int n_components, int bf16, int64_t n_items) { return s6_supported(n_components, bf16, n_items); }
extern "C" int tmf_wmrb_scores5_users_per_workgroup(void) { return tmf::kS5Users; }
extern "C" size_t tmf_wmrb_scores5_workspace_bytes(int64_t n_wg, int32_t n_windows, int wgs_per_launch) {
    if (n_wg <= 0 || n_windows <= 1) return 0;
    return (size_t)s5_launches(n_wg, s5_wgs_per_launch(wgs_per_launch)) * 8 * (size_t)n_windows * tmf::kS5Stride * sizeof(int);
}
extern "C" int tmf_wmrb_scores5_supported(int n_components, int bf16, int64_t n_items) {
    const tmf::RowGeom geom = bf16 ? tmf::row_geom_bf16(n_components) : tmf::row_geom(n_components);
    const int64_t row_bytes = (int64_t)geom.ld * (bf16 ? 2 : 4);
    return geom.G == 32 && row_bytes * tmf::kS5Users <= 128 * 1024 && n_items > 0 && n_items < (1 << 24) &&
           n_items * row_bytes < ((int64_t)1 << 32);
}
