// Row passes of one training epoch: MSE gather->dot->loss->gradient->Adam, the weighted
// gather-sum used by the WMRB item side, the combine of multi-segment rows, the standalone
// fresh-Adam row update and the deterministic loss sum.  See include/tmf.h for the contracts
// and DESIGN.md for the bytes each kernel moves.
#include <stdlib.h>
#include <type_traits>

#include "tmf_common.h"

namespace tmf {

constexpr int kWavesPerBlock = 2;   // independent waves, no barrier; 2 per workgroup measured best (C4 item pass 32.1 ms; 4: 32.5, 8: 36.6; MSE epoch 11.1 vs 11.45)
constexpr int kUnroll = 4;          // list entries a group keeps in flight
#ifndef TMF_MSE_UNROLL
#define TMF_MSE_UNROLL 4
#endif
constexpr int kMseUnroll = TMF_MSE_UNROLL;   // the same for k_mse_pass (A/B builds)
constexpr int64_t kMaxBlocks = ((int64_t)1 << 32) / (64 * kWavesPerBlock) - 1;   // workgroups of one launch: < 2^32 work-items

struct SegView {
    const int64_t* rowptr;
    const int32_t* seg_row;
    const int32_t* seg_chunk;
    const int32_t* seg_slab;
    int64_t nseg;
    int32_t chunk;
    int32_t row_mod;  // > 0: list rows are (block * row_mod + table row); 0: list row == table row
    int64_t seg0;     // first segment of this launch (a launch carries < 2^32 work-items: long segment lists go out in pieces)
    int xcd_run;      // k_wsum_pass_pg: consecutive workgroups per XCD run (0 = plain block order)
};

// ---------------------------------------------------------------------------------------------
// MSE pass.  One wave per segment; 64/G groups of G lanes each take every (64/G)-th entry of the
// segment, kUnroll entries in flight per group.  p_k is reduced over the group with xor shuffles,
// the gradient row is accumulated in registers and reduced over the groups once at the end.
// ---------------------------------------------------------------------------------------------
template <int G, int NV, typename T>
__global__ __launch_bounds__(64 * kWavesPerBlock) void k_mse_pass(
    SegView sv, const int32_t* __restrict__ other, const float* __restrict__ val,
    const T* __restrict__ X_old, const T* __restrict__ Y_old, void* __restrict__ X_out,
    float* __restrict__ slab, float* __restrict__ loss_part, int epi, tmf_adam adam) {
    constexpr int NG = 64 / G;
    const int lane = threadIdx.x & 63;
    const int64_t seg = sv.seg0 + (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (seg >= sv.nseg) return;
    const int g = lane & (G - 1), grp = lane / G;
    const int lrow = sv.seg_row[seg];
    const int row = sv.row_mod > 0 ? lrow % sv.row_mod : lrow;  // table row this list belongs to
    const int64_t rbeg = sv.rowptr[lrow], rend = sv.rowptr[lrow + 1];
    const int64_t beg = rbeg + (int64_t)sv.seg_chunk[seg] * sv.chunk;
    const int64_t end = (beg + sv.chunk < rend) ? beg + sv.chunk : rend;

    Frag<NV> x, acc;
    load_row<G, NV>(x, X_old, row, g);
    zero<NV>(acc);
    float lsum = 0.f;

    for (int64_t k0 = beg + grp; k0 < end; k0 += (int64_t)NG * kMseUnroll) {
        Raw<NV, T> raw[kMseUnroll];
        float a[kMseUnroll];
        int j[kMseUnroll];
        bool ok[kMseUnroll];
        // ids and values first, unconditionally (index clamped into the segment), so that the four id loads
        // are in flight together instead of one id -> row round trip after the other
#pragma unroll
        for (int t = 0; t < kMseUnroll; ++t) {
            const int64_t k = k0 + (int64_t)t * NG;
            ok[t] = k < end;
            const int64_t kc = ok[t] ? k : end - 1;
            j[t] = other[kc];
            a[t] = val[kc];
        }
#pragma unroll
        for (int t = 0; t < kMseUnroll; ++t) {
            load_raw<G, NV>(raw[t], Y_old, j[t], g);  // padded slots re-read the segment's last entry; masked below
        }
#pragma unroll
        for (int t = 0; t < kMseUnroll; ++t) {
            Frag<NV> y;
            to_frag<NV>(y, raw[t]);
            const float p = group_allsum<G>(dot_partial<NV>(x, y));
            const float e = ok[t] ? a[t] - p : 0.f;  // padded slots contribute nothing
            lsum += e * e;
            axpy<NV>(acc, -2.0f * e, y);
        }
    }
    across_groups_sum<G, NV>(acc);
    if (loss_part != nullptr) {
        // every lane of a group carries the same e*e: take lane 0 of each group
        float l = (g == 0) ? lsum : 0.f;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) l += __shfl_xor(l, off, 64);
        if (lane == 0) loss_part[seg] = l;
    }
    if (grp == 0) {
        const int slot = sv.seg_slab[seg];
        if (slot < 0) row_epilogue<G, NV, T>(acc, X_old, X_out, row, g, epi, adam);
        else store_row_f32<G, NV, T>(acc, slab, slot, g);
    }
}

// ---------------------------------------------------------------------------------------------
// Weighted gather-sum pass: g[row] = sum_e wbuf[ent_w[e]] * T[ent_row[e]].  Each wave stages a tile of
// its segment's entries (row id + gathered weight) in LDS first, so that the row gathers depend on an LDS
// read only - the chain entry -> weight -> row would otherwise be three dependent global reads.
// (Giving every XCD a contiguous eighth of the segments - a user block of its own, so that the 4-byte weight gathers of
// neighbouring lists are served by ONE L2 - changed nothing: 32.8 vs 32.6 ms at C4, profiles/r02_hinge_rewrite.txt.)
// ---------------------------------------------------------------------------------------------
constexpr int kWsumTile = 512;  // entries a wave stages in LDS per step (ids + weights: 4 KB per wave)
#ifndef TMF_WSUM_VARIANT
#define TMF_WSUM_VARIANT 0   // 1 / 2: timing-only builds that split the item pass's time (tools/build_variant.sh; profiles/r05_item_pass_split.txt)
#endif
#ifndef TMF_WSUM_OCC
#define TMF_WSUM_OCC   // e.g. -DTMF_WSUM_OCC='__attribute__((amdgpu_waves_per_eu(8,8)))' for an occupancy experiment
#endif
#ifndef TMF_WSUM_INNER
// 0 (default): scalar LDS reads and a branch around the load of a zero-weight row; 1: gradu3's form - the four ids / weights of a
// step in one ds_read_b128 each, a select to a resident row instead of the branch.  Same box, C4 fp32 item pass (two runs each,
// profiles/r05_item_pass_split.txt): 0 -> 30.40 / 30.40 ms, 1 -> 31.52 / 31.52 ms.  The form with fewer instructions is slower,
// as the leaner walks of round 4 were: the pass is bound by the rows' way through the texture addresser / L1 (20.4 ms with
// every row an L1 hit and no weight gather) and by the fabric traffic of the 4-byte weight gathers (+7 ms), not by issue.
#define TMF_WSUM_INNER 0
#endif
#ifndef TMF_WSUM_UNROLL
#define TMF_WSUM_UNROLL 4    // rows a lane group keeps in flight in k_wsum_pass_pg (A/B builds)
#endif

template <int G, int NV, typename T>
__global__ __launch_bounds__(64 * kWavesPerBlock) void k_wsum_pass(
    SegView sv, const int32_t* __restrict__ ent_row, const int32_t* __restrict__ ent_w,
    const float* __restrict__ wbuf, const T* __restrict__ Tab, const T* __restrict__ X_old,
    void* __restrict__ X_out, float* __restrict__ slab, int epi, tmf_adam adam) {
    constexpr int NG = 64 / G;
    __shared__ int s_ids[kWavesPerBlock][kWsumTile];
    __shared__ float s_w[kWavesPerBlock][kWsumTile];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t seg = sv.seg0 + (int64_t)blockIdx.x * kWavesPerBlock + wave;
    if (seg >= sv.nseg) return;
    const int g = lane & (G - 1), grp = lane / G;
    const int row = sv.seg_row[seg];
    const int64_t rbeg = sv.rowptr[row], rend = sv.rowptr[row + 1];
    const int64_t beg = rbeg + (int64_t)sv.seg_chunk[seg] * sv.chunk;
    const int64_t end = (beg + sv.chunk < rend) ? beg + sv.chunk : rend;
    int* ids = s_ids[wave];
    float* ws = s_w[wave];

    Frag<NV> acc;
    zero<NV>(acc);
    for (int64_t t0 = beg; t0 < end; t0 += kWsumTile) {
        const int cnt = (int)((end - t0 < kWsumTile) ? end - t0 : kWsumTile);
        // the wave stages the tile: coalesced entry reads, all weight gathers in flight together
        for (int e = lane; e < cnt; e += 64) {
            ids[e] = ent_row[t0 + e];
            ws[e] = wbuf[ent_w[t0 + e]];
        }
        wave_lds_sync();
        // lane group `grp` takes entries grp, grp + NG, ...; a row is only loaded when its weight is not 0
        for (int e0 = grp; e0 < cnt; e0 += NG * kUnroll) {
            Raw<NV, T> raw[kUnroll];
            float wc[kUnroll];
#pragma unroll
            for (int t = 0; t < kUnroll; ++t) {
                const int e = e0 + t * NG;
                wc[t] = (e < cnt) ? ws[e] : 0.f;
                if constexpr (std::is_same<T, float>::value) {
                    // fp32: skipping the load of a zero-weight row measured 0.7 ms faster at C4 than the branch-free form
                    if (wc[t] != 0.f) load_raw<G, NV>(raw[t], Tab, ids[e], g);
                    else zero_raw<NV>(raw[t]);
                } else {
                    // bf16: no branch around the load (a zero weight reads the tile's first row, an L1 hit, and multiplies it
                    // by 0) - with a branch the compiler waits for every load before issuing the next
                    load_raw<G, NV>(raw[t], Tab, ids[wc[t] != 0.f ? e : 0], g);
                }
            }
#pragma unroll
            for (int t = 0; t < kUnroll; ++t) {
                Frag<NV> y;
                to_frag<NV>(y, raw[t]);
                axpy<NV>(acc, wc[t], y);
            }
        }
    }
    across_groups_sum<G, NV>(acc);
    if (grp == 0) {
        const int slot = sv.seg_slab[seg];
        if (slot < 0) row_epilogue<G, NV, T>(acc, X_old, X_out, row, g, epi, adam);
        else store_row_f32<G, NV, T>(acc, slab, slot, g);
    }
}

// One lane GROUP per segment instead of one wave: the 64/G groups of a wave walk 64/G different segments (two 69-entry lists
// per wave at C4), there is no sum across groups at the end and half as many waves.  Same box, item pass: C4 fp32 34.0 ->
// 32.6 ms, C4 bf16 27.8 -> 25.0 ms.  Used for rows of 16 lanes or more (narrower rows would leave each group a tile of
// a few dozen staged entries); TMF_WSUM_PER_GROUP=0 selects k_wsum_pass for A/B runs.
template <int G, int NV, typename T>
__global__ __launch_bounds__(64 * kWavesPerBlock) TMF_WSUM_OCC void k_wsum_pass_pg(
    SegView sv, const int32_t* __restrict__ ent_row, const int32_t* __restrict__ ent_w,
    const float* __restrict__ wbuf, const T* __restrict__ Tab, const T* __restrict__ X_old,
    void* __restrict__ X_out, float* __restrict__ slab, int epi, tmf_adam adam) {
    constexpr int NG = 64 / G, TILE = kWsumTile / NG, kWsumUnroll = TMF_WSUM_UNROLL;
    __shared__ __attribute__((aligned(16))) int s_ids[kWavesPerBlock][kWsumTile];
    __shared__ __attribute__((aligned(16))) float s_w[kWavesPerBlock][kWsumTile];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane & (G - 1), grp = lane / G;
    // XCD-contiguous block order (sv.xcd_run > 0; speed only): blocks b and b + 8 share an XCD under the observed round-robin
    // placement, so XCD x is given RUNS of xcd_run consecutive workgroups (= consecutive lists of one user block) instead of every
    // eighth one - neighbouring lists gather their 4-byte weights from the same lines of D (a 64-byte line of a user's row holds
    // the weights of 16 items that lie ~1,500 item ids apart at C4), which then meet in ONE L2 instead of eight.
    int64_t blk = blockIdx.x;
    if (sv.xcd_run > 0) {
        const int64_t super = (int64_t)8 * sv.xcd_run, full = (int64_t)gridDim.x / super * super;
        if (blk < full) {
            const int64_t sc = blk / super, r = blk % super;
            blk = sc * super + (r & 7) * sv.xcd_run + (r >> 3);
        }
    }
    const int64_t seg = sv.seg0 + (blk * kWavesPerBlock + wave) * NG + grp;
    const bool live = seg < sv.nseg;
    const int row = live ? sv.seg_row[seg] : 0;
    int64_t beg = 0, end = 0;
    if (live) {
        const int64_t rbeg = sv.rowptr[row], rend = sv.rowptr[row + 1];
        beg = rbeg + (int64_t)sv.seg_chunk[seg] * sv.chunk;
        end = (beg + sv.chunk < rend) ? beg + sv.chunk : rend;
    }
    int* ids = s_ids[wave] + grp * TILE;
    float* ws = s_w[wave] + grp * TILE;
    Frag<NV> acc;
    zero<NV>(acc);
    for (int64_t t0 = beg; t0 < end; t0 += TILE) {
        const int cnt = (int)((end - t0 < TILE) ? end - t0 : TILE);
        for (int e = g; e < cnt; e += G) {   // (non-temporal loads of the entry lists were measured: 30.65 -> 32.7 ms; not used)
#if TMF_WSUM_VARIANT == 1   // timing only: weight derived from the streamed entry, no 4-byte gather
            ids[e] = ent_row[t0 + e];
            ws[e] = __int_as_float(0x3f800000 | (ent_w[t0 + e] & 0xffff));
#elif TMF_WSUM_VARIANT == 2   // timing only: sixteen L1-resident rows, no row gather from the L2s
            ids[e] = ent_row[t0 + e] & 15;
            ws[e] = wbuf[ent_w[t0 + e]];
#elif TMF_WSUM_VARIANT == 3   // timing only: neither gather - streams, LDS, arithmetic and the partial-row stores remain
            ids[e] = ent_row[t0 + e] & 15;
            ws[e] = __int_as_float(0x3f800000 | (ent_w[t0 + e] & 0xffff));
#else
            ids[e] = ent_row[t0 + e];
            ws[e] = wbuf[ent_w[t0 + e]];
#endif
        }
        wave_lds_sync();
#if TMF_WSUM_INNER == 0
        for (int e0 = 0; e0 < cnt; e0 += kWsumUnroll) {
            Raw<NV, T> raw[kWsumUnroll];
            float wc[kWsumUnroll];
#pragma unroll
            for (int t = 0; t < kWsumUnroll; ++t) {
                const int e = e0 + t;
                wc[t] = (e < cnt) ? ws[e] : 0.f;
                if constexpr (std::is_same<T, float>::value) {
                    if (wc[t] != 0.f) load_raw<G, NV>(raw[t], Tab, ids[e], g);
                    else zero_raw<NV>(raw[t]);
                } else {
                    load_raw<G, NV>(raw[t], Tab, ids[wc[t] != 0.f ? e : 0], g);
                }
            }
#pragma unroll
            for (int t = 0; t < kWsumUnroll; ++t) {
                Frag<NV> y;
                to_frag<NV>(y, raw[t]);
                axpy<NV>(acc, wc[t], y);
            }
        }
#else
        // the four ids and the four weights of a step in ONE LDS read each (16-byte aligned tile, e0 % 4 == 0; slots past cnt hold
        // stale values and are never used), the four row loads issued back to back, a zero weight (or a slot past the list) as a
        // select to the tile's first row - an L1 hit multiplied by 0 - instead of a branch around the load
        static_assert(TILE % 4 == 0, "vector LDS reads assume four entries per step");
        const int safe = ids[0];
        for (int e0 = 0; e0 < cnt; e0 += 4) {
            const int4 id4 = *reinterpret_cast<const int4*>(ids + e0);
            const float4 w4 = *reinterpret_cast<const float4*>(ws + e0);
            const int idv[4] = {id4.x, id4.y, id4.z, id4.w};
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
            Raw<NV, T> raw[4];
            float wc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool want = e0 + t < cnt && wv[t] != 0.f;
                wc[t] = want ? wv[t] : 0.f;
                load_raw<G, NV>(raw[t], Tab, want ? idv[t] : safe, g);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                Frag<NV> y;
                to_frag<NV>(y, raw[t]);
                axpy<NV>(acc, wc[t], y);
            }
        }
#endif
    }
    if (live) {
        const int slot = sv.seg_slab[seg];
        if (slot < 0) row_epilogue<G, NV, T>(acc, X_old, X_out, row, g, epi, adam);
        else store_row_f32<G, NV, T>(acc, slab, slot, g);
    }
}

// ---------------------------------------------------------------------------------------------
// Row-stationary weighted gather-sum ("rows4"): a lane group OWNS up to K consecutive output rows (items), keeps their
// sums in registers and walks the user blocks itself: the entry lists are the same per-(user block, item) lists the pass
// above consumes (list row = block * n_rows + item), but no list produces a partial row - no slab, no combine.  In block t
// the lists of the lane group's items are ONE contiguous range of entries, staged flat (row id + gathered weight) in tiles of
// kRows4Tile; item k takes its part of every tile.  A launch covers as many rows as are resident together, so the
// workgroups walk the blocks in loose lockstep (one workgroup barrier + one bounded rendezvous per block) and the block of U
// rows being gathered from stays in the L2s.
// Sum order of a row: blocks ascending, list order inside a block - ONE running fp32 sum (the slab form adds per-block
// partial sums), so the two forms agree to rounding, not to the bit.
// ---------------------------------------------------------------------------------------------
constexpr int kRows4Tile = 64;
// Rows per lane group (K) and rows in flight (U) of the row-stationary pass for rows of 8 accumulator registers (bf16 r = 256, the
// config-5 shard), item pass ms at 128 VGPRs / 4 waves per SIMD (profiles/r05_c5_shard.txt): K=8 U=4 81.1 (rounds 3-4)  K=8 U=6 108.6
// K=8 U=8 129.3 (spills)  K=7 U=5 75.7  K=6 U=4 74.9  K=6 U=5 70.6  K=6 U=6 70.5  K=6 U=7 79.4  K=5 U=6 70.4  K=5 U=8 83.0;
// 4-wave workgroups at 3 per CU (168 VGPRs): K=8 U=6 82.0, K=10 U=6 81.4.  A (user block, row) run is ~9 entries: 6 + 3 instead of
// 4 + 4 + 1 dependent trips.  Rows of 4 accumulator registers (fp32 r <= 256) keep K = 15, U = 4 (not re-measured).
#ifndef TMF_ROWS4_UNROLL
#define TMF_ROWS4_UNROLL 6
#endif
constexpr int kRows4Unroll2 = TMF_ROWS4_UNROLL, kRows4Unroll1 = 4;   // rows in flight per lane group: NV >= 2 / NV = 1
#ifndef TMF_ROWS4_K2
#define TMF_ROWS4_K2 6
#endif
#ifndef TMF_ROWS4_WAVES
#define TMF_ROWS4_WAVES 8
#endif
#ifndef TMF_ROWS4_MINW
#define TMF_ROWS4_MINW (TMF_ROWS4_WAVES / 2)   // waves per SIMD the register allocation must allow (4: two 8-wave workgroups per CU)
#endif
constexpr int kRows4Waves = TMF_ROWS4_WAVES;
constexpr int kRows4K2 = TMF_ROWS4_K2;           // output rows per lane group when a row takes 8 accumulator registers (NV = 2)

// BALANCED form ("rows5", round 5): the rows a lane group owns are VIRTUAL rows - (output row, part p of P) - listed in (row, part)
// order, so that a popular item (C4 / config 5: the top item has one list entry per user, the average 1,400) is cut into P parts
// of about the average size and every lane group walks about the same number of entries: in block t part p of row i takes the
// entries [b + p L / P, b + (p + 1) L / P) of that block's list [b, b + L) of the row.  Consecutive virtual rows are consecutive
// parts of one row or consecutive rows, so the K virtual rows of a lane group are still ONE contiguous range of entries in every
// block.  A row with P = 1 is finished by its lane group (epilogue); the parts of a cut row go to slab slots and
// tmf_combine_rows adds them in part order - a fixed order, so the result is bit-reproducible.  vr.item == nullptr: the plain
// form above (virtual row v = output row v).
struct VRows {
    const int32_t* item;     // [n_vrows + 1] output row of every virtual row; item[n_vrows] = n_rows (the end of a block's lists)
    const int32_t* part;     // [n_vrows + 1] part index, 0 for the sentinel
    const int32_t* nparts;   // [n_vrows + 1] parts of that row, 1 for the sentinel
    const int32_t* slot;     // [n_vrows] slab slot of a part of a cut row, -1 = the whole row: epilogue
    float* slab;
};

template <int G, int NV, typename T, int K, int WAVES>
__global__ __launch_bounds__(64 * WAVES, TMF_ROWS4_MINW) void k_wsum_rows4(
    const int64_t* __restrict__ rowptr, int64_t n_rows, int n_blocks, const int32_t* __restrict__ ent_row,
    const int32_t* __restrict__ ent_w, const float* __restrict__ wbuf, const T* __restrict__ Tab, const T* __restrict__ X_old,
    void* __restrict__ X_out, int epi, tmf_adam adam, int64_t row_begin, int64_t row_end, int* __restrict__ sync, int lag,
    VRows vr) {
    static_assert(K + 1 <= G, "the list boundaries of a block live in one register of the lane group");
    constexpr int NG = 64 / G, NGB = NG * WAVES, U = NV >= 2 ? kRows4Unroll2 : kRows4Unroll1;
    __shared__ int s_ids[NGB][kRows4Tile];
    __shared__ float s_w[NGB][kRows4Tile];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane & (G - 1), gid = wave * NG + lane / G;
    int* ids = s_ids[gid];
    float* ws = s_w[gid];
    const int64_t j0 = row_begin + ((int64_t)blockIdx.x * NGB + gid) * K;
    const int kv = (int)((row_end - j0 < K) ? (row_end > j0 ? row_end - j0 : 0) : K);   // rows this lane group really owns
    Frag<NV> acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) zero<NV>(acc[k]);
    // lane l <= kv describes the boundary "first entry of (virtual) row j0 + l"; lanes beyond kv repeat the last one
    const bool balanced = vr.item != nullptr;
    const int64_t vq = j0 + (g < kv ? g : kv);
    const int64_t it = (balanced && kv > 0) ? (int64_t)vr.item[vq] : vq;
    const int pt = (balanced && kv > 0) ? vr.part[vq] : 0, np = (balanced && kv > 0) ? vr.nparts[vq] : 1;
    for (int t = 0; t < n_blocks; ++t) {
        // entries fit 31 bits (include/tmf.h); a cut row: its part's share of this block's list
        int bnd = 0;
        if (kv > 0) {
            const int64_t lb = rowptr[(int64_t)t * n_rows + it];
            bnd = (int)lb;
            if (np > 1) bnd += (int)(((int64_t)pt * (rowptr[(int64_t)t * n_rows + it + 1] - lb)) / np);
        }
        const int r_beg = group_read<G>(bnd, 0), r_end = group_read<G>(bnd, kv);
        int b[K + 1];
#pragma unroll
        for (int k = 0; k <= K; ++k) b[k] = group_read<G>(bnd, k < kv ? k : kv);
        for (int c0 = r_beg; c0 < r_end; c0 += kRows4Tile) {
            const int cnt = (r_end - c0 < kRows4Tile) ? r_end - c0 : kRows4Tile;
            for (int e = g; e < cnt; e += G) {
#if TMF_WSUM_VARIANT == 1   // timing only (see k_wsum_pass_pg): no 4-byte weight gather
                ids[e] = ent_row[c0 + e];
                ws[e] = __int_as_float(0x3f800000 | (ent_w[c0 + e] & 0xffff));
#elif TMF_WSUM_VARIANT == 2   // timing only: sixteen L1-resident rows
                ids[e] = ent_row[c0 + e] & 15;
                ws[e] = wbuf[ent_w[c0 + e]];
#elif TMF_WSUM_VARIANT == 3
                ids[e] = ent_row[c0 + e] & 15;
                ws[e] = __int_as_float(0x3f800000 | (ent_w[c0 + e] & 0xffff));
#else
                ids[e] = ent_row[c0 + e];
                ws[e] = wbuf[ent_w[c0 + e]];
#endif
            }
            wave_lds_sync();
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int lo = (b[k] > c0 ? b[k] : c0) - c0;
                const int hi = ((b[k + 1] < c0 + cnt) ? b[k + 1] : c0 + cnt) - c0;
                for (int e0 = lo; e0 < hi; e0 += U) {
                    Raw<NV, T> raw[U];
                    float wc[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int e = e0 + u;
                        wc[u] = (e < hi) ? ws[e] : 0.f;
                        load_raw<G, NV>(raw[u], Tab, ids[(wc[u] != 0.f) ? e : lo], g);   // weight 0: an L1-hot row times 0
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        Frag<NV> y;
                        to_frag<NV>(y, raw[u]);
                        axpy<NV>(acc[k], wc[u], y);
                    }
                }
            }
            wave_lds_sync();   // the next tile rewrites the stage
        }
        if (sync != nullptr && tid == 0) step_rendezvous(sync, t, lag, (int)gridDim.x);
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int64_t row = group_read<G>((int)it, k < kv ? k : 0);   // every lane of the wave takes part in the exchange
        if (k >= kv) continue;
        const int slot = balanced ? vr.slot[j0 + k] : -1;
        if (slot < 0) row_epilogue<G, NV, T>(acc[k], X_old, X_out, row, g, epi, adam);
        else store_row_f32<G, NV, T>(acc[k], vr.slab, slot, g);
    }
}

// ---------------------------------------------------------------------------------------------
// Rows cut into several segments: sum the slab slots in slot order (group t takes slots t, t+NG..
// then the fixed butterfly over groups), then the epilogue.
// ---------------------------------------------------------------------------------------------
template <int G, int NV, typename T>
__global__ __launch_bounds__(64 * kWavesPerBlock) void k_combine_rows(
    const int32_t* __restrict__ long_rows, const int64_t* __restrict__ slab_beg, int64_t n_long,
    const float* __restrict__ slab, const T* __restrict__ X_old, void* __restrict__ X_out, int epi,
    tmf_adam adam) {
    constexpr int NG = 64 / G;
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (i >= n_long) return;
    const int g = lane & (G - 1), grp = lane / G;
    const int row = long_rows[i];
    const int64_t beg = slab_beg[i], end = slab_beg[i + 1];
    Frag<NV> acc;
    zero<NV>(acc);
    for (int64_t s = beg + grp; s < end; s += NG) {
        Frag<NV> y;
        load_row_f32<G, NV, T, TMF_NT_FIN>(y, slab, s, g);
        add<NV>(acc, y);
    }
    across_groups_sum<G, NV>(acc);
    if (grp == 0) row_epilogue<G, NV, T>(acc, X_old, X_out, row, g, epi, adam);
}

// ---------------------------------------------------------------------------------------------
// K6 standalone: elementwise fresh-Adam over a [n_rows, ld] table (float4 per lane, grid-stride).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adam_rows(float4* __restrict__ W, const float4* __restrict__ Gr,
                                                   int64_t n4, tmf_adam adam) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 w = W[i];
        const float4 g = Gr[i];
        w.x = adam_fresh(w.x, g.x, adam);
        w.y = adam_fresh(w.y, g.y, adam);
        w.z = adam_fresh(w.z, g.z, adam);
        w.w = adam_fresh(w.w, g.w, adam);
        W[i] = w;
    }
}

__global__ __launch_bounds__(256) void k_adam_rows_bf16(bf16x8* __restrict__ W, const float4* __restrict__ Gr,
                                                        int64_t n8, tmf_adam adam) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        f32x8 w = __builtin_convertvector(W[i], f32x8);
        const float4 a = Gr[2 * i], b = Gr[2 * i + 1];
        w[0] = adam_fresh(w[0], a.x, adam); w[1] = adam_fresh(w[1], a.y, adam);
        w[2] = adam_fresh(w[2], a.z, adam); w[3] = adam_fresh(w[3], a.w, adam);
        w[4] = adam_fresh(w[4], b.x, adam); w[5] = adam_fresh(w[5], b.y, adam);
        w[6] = adam_fresh(w[6], b.z, adam); w[7] = adam_fresh(w[7], b.w, adam);
        W[i] = __builtin_convertvector(w, bf16x8);
    }
}

// Opt-in extension (not the reference's behaviour): Keras Adam with PERSISTENT moments, one step over a whole table.
//   m += (g - m)(1 - b1);  v += (g g - v)(1 - b2);  w -= (m alpha_t) / (sqrt(v) + eps)
// - the same fp32 op sequence as adam_fresh, which it equals bit for bit at step 1 with zero moments.
__global__ __launch_bounds__(256) void k_adam_state_rows(float4* __restrict__ W, const float4* __restrict__ Gr,
                                                         float4* __restrict__ M, float4* __restrict__ V2, int64_t n4,
                                                         tmf_adam a) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 w = W[i], m = M[i], v = V2[i];
        const float4 g = Gr[i];
        float* wp = &w.x;
        float* mp = &m.x;
        float* vp = &v.x;
        const float* gp = &g.x;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            mp[c] = mp[c] + (gp[c] - mp[c]) * a.one_minus_b1;
            vp[c] = vp[c] + (gp[c] * gp[c] - vp[c]) * a.one_minus_b2;
            wp[c] = wp[c] - __fdiv_rn(mp[c] * a.alpha, __fsqrt_rn(vp[c]) + a.eps);
        }
        W[i] = w;
        M[i] = m;
        V2[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Deterministic sum: one 1024-thread block, fp64 partials per thread in a fixed strided order, fixed LDS tree.
// The block reads 16 bytes per lane with four loads in flight (~1M partials = 4 MB at C4: 0.42 ms with scalar
// loads one at a time; this form is bound by what one CU pulls from L2 / HBM).  A second stage over several
// blocks would need a workspace the ABI does not have; one CU is enough for a few MB.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_sum_f32(const float* __restrict__ x, int64_t n, double* __restrict__ out) {
    __shared__ double sh[1024];
    constexpr int kInFlight = 4;
    double s = 0.0;
    const int64_t head = (16 - (reinterpret_cast<uintptr_t>(x) & 15)) & 15;   // bytes to the first 16-byte boundary
    const int64_t pre = (head / 4 < n) ? head / 4 : n;                          // scalar prefix
    if ((int64_t)threadIdx.x < pre) s += (double)x[threadIdx.x];
    const tmf_f4* x4 = reinterpret_cast<const tmf_f4*>(x + pre);
    const int64_t n4 = (n - pre) / 4;
    for (int64_t i = threadIdx.x; i < n4; i += 1024 * kInFlight) {
        tmf_f4 v[kInFlight];
#pragma unroll
        for (int t = 0; t < kInFlight; ++t) {
            const int64_t k = i + (int64_t)t * 1024;
            const tmf_f4 zero4 = {0.f, 0.f, 0.f, 0.f};
            v[t] = (k < n4) ? __builtin_nontemporal_load(x4 + k) : zero4;
        }
#pragma unroll
        for (int t = 0; t < kInFlight; ++t) s += ((double)v[t][0] + (double)v[t][1]) + ((double)v[t][2] + (double)v[t][3]);
    }
    const int64_t tail = pre + 4 * n4;
    if (tail + threadIdx.x < n) s += (double)x[tail + threadIdx.x];   // fewer than four elements
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

static inline SegView view(const tmf_segments* s) {
    return SegView{s->rowptr, s->seg_row, s->seg_chunk, s->seg_slab, s->nseg, s->chunk, s->row_mod, 0, 0};
}

static int check_segments(const tmf_segments* s) {
    TMF_REQUIRE(s != nullptr, "segments is null");
    TMF_REQUIRE(s->nseg >= 0 && s->chunk > 0, "segments: nseg=%lld chunk=%d", (long long)s->nseg, s->chunk);
    TMF_REQUIRE(s->nseg == 0 || (s->rowptr && s->seg_row && s->seg_chunk && s->seg_slab), "segments: null array");
    return TMF_OK;
}

}  // namespace tmf

using namespace tmf;

template <typename T>
static int mse_pass_impl(const tmf_segments* seg, const int32_t* other, const float* val, const void* X_old,
                         const void* Y_old, void* X_out, float* slab, float* loss_part, int n_components, int epi,
                         tmf_adam adam, void* stream) {
    if (int rc = check_segments(seg)) return rc;
    if (seg->nseg == 0) return TMF_OK;
    TMF_REQUIRE(X_old && Y_old && X_out, "mse_pass: null table");
    TMF_REQUIRE(epi == TMF_EPI_ADAM || epi == TMF_EPI_GRAD, "mse_pass: bad epilogue %d", epi);
    const RowGeom geom = row_geom_of<T>(n_components);
    SegView sv = view(seg);
    // a launch carries < 2^32 work-items (tmf::launch_fits): the segment list goes out in pieces of kMaxBlocks workgroups
    for (sv.seg0 = 0; sv.seg0 < seg->nseg; sv.seg0 += kMaxBlocks * kWavesPerBlock) {
        const int64_t left = seg->nseg - sv.seg0;
        const int64_t want = (left + kWavesPerBlock - 1) / kWavesPerBlock;
        const unsigned blocks = (unsigned)(want < kMaxBlocks ? want : kMaxBlocks);
#define CALL(G_, NV_)                                                                                          \
    hipLaunchKernelGGL((k_mse_pass<G_, NV_, T>), dim3(blocks), dim3(64 * kWavesPerBlock), 0, (hipStream_t)stream, \
                       sv, other, val, (const T*)X_old, (const T*)Y_old, X_out, slab, loss_part, epi, adam)
        TMF_DISPATCH(T, geom, CALL);
#undef CALL
    }
    return check_launch("tmf_mse_pass");
}

template <typename T>
static int wsum_pass_impl(const tmf_segments* seg, const int32_t* ent_row, const int32_t* ent_w, const float* wbuf,
                          const void* Tab, const void* X_old, void* X_out, float* slab, int n_components, int epi,
                          tmf_adam adam, void* stream) {
    if (int rc = check_segments(seg)) return rc;
    if (seg->nseg == 0) return TMF_OK;
    TMF_REQUIRE(Tab && X_out && ent_row && ent_w && wbuf && (epi == TMF_EPI_GRAD || X_old), "wsum_pass: null pointer");
    TMF_REQUIRE(epi == TMF_EPI_ADAM || epi == TMF_EPI_GRAD, "wsum_pass: bad epilogue %d", epi);
    const RowGeom geom = row_geom_of<T>(n_components);
    SegView sv = view(seg);
    {
        const char* env = getenv("TMF_WSUM_PER_GROUP");
        // ... and only when the segments still fill the chip at 64/G of them per wave (MovieLens-1M shape: 15K segments of
        // 1024 entries - 245 us per group against 206 us per wave)
        const bool forced = env && env[0] == '1';
        if (!(env && env[0] == '0') && geom.G >= 16 && (forced || seg->nseg / (64 / geom.G) >= 16384)) {
            const int64_t per_block = (int64_t)kWavesPerBlock * (64 / geom.G);
            // XCD runs of 512 workgroups (2048 lists) when the pass has at least a few super-chunks of them.  Same box, C4 fp32 item
            // pass ms by run length: 0 (plain order) 31.8 - 32.0, 64: 31.6, 128: 31.1, 256: 30.7 - 30.9, 384: 30.7, 512: 30.7,
            // 1024: 30.9, 2048: 31.4, 8192: 32.0 (profiles/r03_c5_experiments.txt item 8).  TMF_WSUM_XCD_RUN overrides (A/B runs).
            sv.xcd_run = (seg->nseg / per_block >= 4 * 8 * 512) ? 512 : 0;
            if (const char* xr = getenv("TMF_WSUM_XCD_RUN")) sv.xcd_run = atoi(xr);
            for (sv.seg0 = 0; sv.seg0 < seg->nseg; sv.seg0 += kMaxBlocks * per_block) {   // pieces of < 2^32 work-items
                const int64_t want = (seg->nseg - sv.seg0 + per_block - 1) / per_block;
                const unsigned pblocks = (unsigned)(want < kMaxBlocks ? want : kMaxBlocks);
#define CALLPG(G_, NV_)                                                                                               \
    hipLaunchKernelGGL((k_wsum_pass_pg<G_, NV_, T>), dim3(pblocks), dim3(64 * kWavesPerBlock), 0, (hipStream_t)stream, \
                       sv, ent_row, ent_w, wbuf, (const T*)Tab, (const T*)X_old, X_out, slab, epi, adam)
                TMF_DISPATCH(T, geom, CALLPG);
#undef CALLPG
            }
            return check_launch("tmf_wsum_pass (per group)");
        }
    }
    for (sv.seg0 = 0; sv.seg0 < seg->nseg; sv.seg0 += kMaxBlocks * kWavesPerBlock) {   // pieces of < 2^32 work-items
        const int64_t want = (seg->nseg - sv.seg0 + kWavesPerBlock - 1) / kWavesPerBlock;
        const unsigned blocks = (unsigned)(want < kMaxBlocks ? want : kMaxBlocks);
#define CALL(G_, NV_)                                                                                           \
    hipLaunchKernelGGL((k_wsum_pass<G_, NV_, T>), dim3(blocks), dim3(64 * kWavesPerBlock), 0, (hipStream_t)stream, \
                       sv, ent_row, ent_w, wbuf, (const T*)Tab, (const T*)X_old, X_out, slab, epi, adam)
        TMF_DISPATCH(T, geom, CALL);
#undef CALL
    }
    return check_launch("tmf_wsum_pass");
}

template <typename T>
static int combine_rows_impl(const int32_t* long_rows, const int64_t* slab_beg, int64_t n_long, const float* slab,
                             const void* X_old, void* X_out, int n_components, int epi, tmf_adam adam, void* stream) {
    if (n_long == 0) return TMF_OK;
    TMF_REQUIRE(n_long > 0 && long_rows && slab_beg && slab && X_out, "combine_rows: bad arguments");
    TMF_REQUIRE(epi == TMF_EPI_GRAD || X_old, "combine_rows: X_old is null");
    const RowGeom geom = row_geom_of<T>(n_components);
    const unsigned blocks = (unsigned)((n_long + kWavesPerBlock - 1) / kWavesPerBlock);
    TMF_REQUIRE_LAUNCH((n_long + kWavesPerBlock - 1) / kWavesPerBlock, 64 * kWavesPerBlock, "combine_rows");
#define CALL(G_, NV_)                                                                                              \
    hipLaunchKernelGGL((k_combine_rows<G_, NV_, T>), dim3(blocks), dim3(64 * kWavesPerBlock), 0, (hipStream_t)stream, \
                       long_rows, slab_beg, n_long, slab, (const T*)X_old, X_out, epi, adam)
    TMF_DISPATCH(T, geom, CALL);
#undef CALL
    return check_launch("tmf_combine_rows");
}

extern "C" int tmf_mse_pass_f32(const tmf_segments* seg, const int32_t* other, const float* val,
                                const float* X_old, const float* Y_old, float* X_out, float* slab,
                                float* loss_part, int n_components, int epi, tmf_adam adam, void* stream) {
    return mse_pass_impl<float>(seg, other, val, X_old, Y_old, X_out, slab, loss_part, n_components, epi, adam, stream);
}
extern "C" int tmf_mse_pass_bf16(const tmf_segments* seg, const int32_t* other, const float* val,
                                 const void* X_old, const void* Y_old, void* X_out, float* slab,
                                 float* loss_part, int n_components, int epi, tmf_adam adam, void* stream) {
    return mse_pass_impl<__bf16>(seg, other, val, X_old, Y_old, X_out, slab, loss_part, n_components, epi, adam, stream);
}

extern "C" int tmf_wsum_pass_f32(const tmf_segments* seg, const int32_t* ent_row, const int32_t* ent_w,
                                 const float* wbuf, const float* T, const float* X_old, float* X_out,
                                 float* slab, int n_components, int epi, tmf_adam adam, void* stream) {
    return wsum_pass_impl<float>(seg, ent_row, ent_w, wbuf, T, X_old, X_out, slab, n_components, epi, adam, stream);
}
extern "C" int tmf_wsum_pass_bf16(const tmf_segments* seg, const int32_t* ent_row, const int32_t* ent_w,
                                  const float* wbuf, const void* T, const void* X_old, void* X_out,
                                  float* slab, int n_components, int epi, tmf_adam adam, void* stream) {
    return wsum_pass_impl<__bf16>(seg, ent_row, ent_w, wbuf, T, X_old, X_out, slab, n_components, epi, adam, stream);
}

template <typename T>
static int wsum_rows4_impl(const int64_t* rowptr, int32_t n_rows, int32_t n_blocks, const int32_t* ent_row, const int32_t* ent_w,
                           const float* wbuf, const void* Tab, const void* X_old, void* X_out, int n_components, int epi,
                           tmf_adam adam, int32_t rows_per_launch, void* workspace, size_t workspace_bytes, void* stream,
                           VRows vr = VRows{nullptr, nullptr, nullptr, nullptr, nullptr}, int32_t n_vrows = 0) {
    if (n_rows == 0) return TMF_OK;
    if (vr.item != nullptr) {
        TMF_REQUIRE(vr.part && vr.nparts && vr.slot && n_vrows >= n_rows, "wsum_rows5: virtual rows: null array or %d virtual rows for %d rows",
                    n_vrows, n_rows);
    }
    const int64_t n_work = vr.item != nullptr ? n_vrows : n_rows;   // what the lane groups own: virtual rows, or the rows themselves
    TMF_REQUIRE(n_rows > 0 && n_blocks > 0 && rows_per_launch > 0, "wsum_rows4: n_rows=%d n_blocks=%d rows_per_launch=%d", n_rows,
                n_blocks, rows_per_launch);
    TMF_REQUIRE(rowptr && ent_row && ent_w && wbuf && Tab && X_out && (epi == TMF_EPI_GRAD || X_old), "wsum_rows4: null pointer");
    TMF_REQUIRE(epi == TMF_EPI_ADAM || epi == TMF_EPI_GRAD, "wsum_rows4: bad epilogue %d", epi);
    const RowGeom geom = row_geom_of<T>(n_components);
    if (geom.G < 16) {
        set_error("wsum_rows4: rows of %d lanes are too narrow for the row-stationary form; use tmf_wsum_pass + tmf_combine_rows", geom.G);
        return TMF_E_UNSUPPORTED;
    }
    constexpr int W4 = kRows4Waves;
    const int K4 = geom.NV >= 2 ? kRows4K2 : 15;   // rows per lane group: 6 x 8 or 15 x 4 accumulator registers
    const int64_t per_block = (int64_t)(64 / geom.G) * W4 * K4;
    const int64_t launches = (n_work + rows_per_launch - 1) / rows_per_launch;
    int* sync = nullptr;
    int lag = 1;
    if (const char* env = getenv("TMF_G4_LAG")) lag = atoi(env);
    if (workspace != nullptr && lag >= 0) {
        const size_t need = rendezvous_bytes(launches, n_blocks);
        TMF_REQUIRE(workspace_bytes >= need, "wsum_rows4: workspace of %zu bytes, %zu needed", workspace_bytes, need);
        if (hipMemsetAsync(workspace, 0, need, (hipStream_t)stream) != hipSuccess) { set_error("wsum_rows4: hipMemsetAsync failed"); return TMF_E_LAUNCH; }
        sync = static_cast<int*>(workspace);
    }
    for (int64_t b = 0, launch = 0; b < n_work; b += rows_per_launch, ++launch) {
        const int64_t e = (b + rows_per_launch < n_work) ? b + rows_per_launch : n_work;
        const unsigned blocks = (unsigned)((e - b + per_block - 1) / per_block);
        int* sy = sync ? sync + launch * n_blocks * 8 * kSyncStride : nullptr;
#define CALLK(G_, NV_, K_)                                                                                                     \
    hipLaunchKernelGGL((k_wsum_rows4<G_, NV_, T, K_, W4>), dim3(blocks), dim3(64 * W4), 0, (hipStream_t)stream, rowptr,          \
                       (int64_t)n_rows, (int)n_blocks, ent_row, ent_w, wbuf, (const T*)Tab, (const T*)X_old, X_out, epi, adam, b, \
                       e, sy, lag, vr)
        if constexpr (std::is_same<T, float>::value) {
            if (geom.NV == 1 && geom.G == 16) { CALLK(16, 1, 15); }
            else if (geom.NV == 1 && geom.G == 32) { CALLK(32, 1, 15); }
            else if (geom.NV == 1 && geom.G == 64) { CALLK(64, 1, 15); }
            else if (geom.NV == 2 && geom.G == 64) { CALLK(64, 2, kRows4K2); }
            else { set_error("wsum_rows4: unsupported n_components"); return TMF_E_UNSUPPORTED; }
        } else {
            if (geom.NV == 2 && geom.G == 16) { CALLK(16, 2, kRows4K2); }
            else if (geom.NV == 2 && geom.G == 32) { CALLK(32, 2, kRows4K2); }
            else if (geom.NV == 2 && geom.G == 64) { CALLK(64, 2, kRows4K2); }
            else { set_error("wsum_rows4: unsupported n_components"); return TMF_E_UNSUPPORTED; }
        }
#undef CALLK
    }
    return check_launch("tmf_wsum_rows4");
}

extern "C" int tmf_wsum_rows4_rows_per_group(int n_components, int bf16) {
    const RowGeom geom = bf16 ? row_geom_bf16(n_components) : row_geom(n_components);
    if (geom.ld == 0 || geom.G < 16 || (geom.NV != 1 && geom.NV != 2)) return 0;
    return (64 / geom.G) * kRows4Waves * (geom.NV >= 2 ? kRows4K2 : 15);   // rows one workgroup owns
}
extern "C" size_t tmf_wsum_rows4_workspace_bytes(int32_t n_rows, int32_t n_blocks, int32_t rows_per_launch) {
    return rows_per_launch > 0 ? rendezvous_bytes(((int64_t)n_rows + rows_per_launch - 1) / rows_per_launch, n_blocks) : 0;
}
extern "C" int tmf_wsum_rows4_f32(const int64_t* rowptr, int32_t n_rows, int32_t n_blocks, const int32_t* ent_row,
                                  const int32_t* ent_w, const float* wbuf, const float* T, const float* X_old, float* X_out,
                                  int n_components, int epi, tmf_adam adam, int32_t rows_per_launch, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    return wsum_rows4_impl<float>(rowptr, n_rows, n_blocks, ent_row, ent_w, wbuf, T, X_old, X_out, n_components, epi, adam,
                                  rows_per_launch, workspace, workspace_bytes, stream);
}
extern "C" int tmf_wsum_rows4_bf16(const int64_t* rowptr, int32_t n_rows, int32_t n_blocks, const int32_t* ent_row,
                                   const int32_t* ent_w, const float* wbuf, const void* T, const void* X_old, void* X_out,
                                   int n_components, int epi, tmf_adam adam, int32_t rows_per_launch, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    return wsum_rows4_impl<__bf16>(rowptr, n_rows, n_blocks, ent_row, ent_w, wbuf, T, X_old, X_out, n_components, epi, adam,
                                   rows_per_launch, workspace, workspace_bytes, stream);
}

extern "C" size_t tmf_wsum_rows5_workspace_bytes(int32_t n_vrows, int32_t n_blocks, int32_t rows_per_launch) {
    return rows_per_launch > 0 ? rendezvous_bytes(((int64_t)n_vrows + rows_per_launch - 1) / rows_per_launch, n_blocks) : 0;
}
extern "C" int tmf_wsum_rows5_f32(const int64_t* rowptr, int32_t n_rows, int32_t n_blocks, const int32_t* ent_row,
                                  const int32_t* ent_w, const float* wbuf, const float* T, const float* X_old, float* X_out,
                                  float* slab, const int32_t* vr_item, const int32_t* vr_part, const int32_t* vr_nparts,
                                  const int32_t* vr_slot, int32_t n_vrows, int n_components, int epi, tmf_adam adam,
                                  int32_t rows_per_launch, void* workspace, size_t workspace_bytes, void* stream) {
    TMF_REQUIRE(vr_item != nullptr, "wsum_rows5: vr_item is null (tmf_wsum_rows4 is the form without virtual rows)");
    return wsum_rows4_impl<float>(rowptr, n_rows, n_blocks, ent_row, ent_w, wbuf, T, X_old, X_out, n_components, epi, adam,
                                  rows_per_launch, workspace, workspace_bytes, stream, VRows{vr_item, vr_part, vr_nparts, vr_slot, slab},
                                  n_vrows);
}
extern "C" int tmf_wsum_rows5_bf16(const int64_t* rowptr, int32_t n_rows, int32_t n_blocks, const int32_t* ent_row,
                                   const int32_t* ent_w, const float* wbuf, const void* T, const void* X_old, void* X_out,
                                   float* slab, const int32_t* vr_item, const int32_t* vr_part, const int32_t* vr_nparts,
                                   const int32_t* vr_slot, int32_t n_vrows, int n_components, int epi, tmf_adam adam,
                                   int32_t rows_per_launch, void* workspace, size_t workspace_bytes, void* stream) {
    TMF_REQUIRE(vr_item != nullptr, "wsum_rows5: vr_item is null (tmf_wsum_rows4 is the form without virtual rows)");
    return wsum_rows4_impl<__bf16>(rowptr, n_rows, n_blocks, ent_row, ent_w, wbuf, T, X_old, X_out, n_components, epi, adam,
                                   rows_per_launch, workspace, workspace_bytes, stream, VRows{vr_item, vr_part, vr_nparts, vr_slot, slab},
                                   n_vrows);
}

extern "C" int tmf_combine_rows_f32(const int32_t* long_rows, const int64_t* slab_beg, int64_t n_long,
                                    const float* slab, const float* X_old, float* X_out, int n_components,
                                    int epi, tmf_adam adam, void* stream) {
    return combine_rows_impl<float>(long_rows, slab_beg, n_long, slab, X_old, X_out, n_components, epi, adam, stream);
}
extern "C" int tmf_combine_rows_bf16(const int32_t* long_rows, const int64_t* slab_beg, int64_t n_long,
                                     const float* slab, const void* X_old, void* X_out, int n_components,
                                     int epi, tmf_adam adam, void* stream) {
    return combine_rows_impl<__bf16>(long_rows, slab_beg, n_long, slab, X_old, X_out, n_components, epi, adam, stream);
}

extern "C" int tmf_adam_fresh_rows_f32(float* W, const float* G, int64_t n_rows, int n_components,
                                       tmf_adam adam, void* stream) {
    if (n_rows == 0) return TMF_OK;
    const RowGeom geom = row_geom(n_components);
    TMF_REQUIRE(geom.ld > 0, "adam_fresh_rows: unsupported n_components %d", n_components);
    TMF_REQUIRE(W && G && n_rows > 0, "adam_fresh_rows: bad arguments");
    const int64_t n4 = n_rows * geom.ld / 4;
    const int64_t want = (n4 + 255) / 256;
    const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);
    hipLaunchKernelGGL(k_adam_rows, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4*>(W), reinterpret_cast<const float4*>(G), n4, adam);
    return check_launch("tmf_adam_fresh_rows_f32");
}

extern "C" int tmf_adam_fresh_rows_bf16(void* W, const float* G, int64_t n_rows, int n_components,
                                        tmf_adam adam, void* stream) {
    if (n_rows == 0) return TMF_OK;
    const RowGeom geom = row_geom_bf16(n_components);
    TMF_REQUIRE(geom.ld > 0, "adam_fresh_rows: unsupported n_components %d", n_components);
    TMF_REQUIRE(W && G && n_rows > 0, "adam_fresh_rows: bad arguments");
    const int64_t n8 = n_rows * geom.ld / 8;
    const int64_t want = (n8 + 255) / 256;
    const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);
    hipLaunchKernelGGL(k_adam_rows_bf16, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<bf16x8*>(W), reinterpret_cast<const float4*>(G), n8, adam);
    return check_launch("tmf_adam_fresh_rows_bf16");
}

extern "C" int tmf_adam_state_rows_f32(float* W, const float* G, float* M, float* V, int64_t n_rows, int n_components,
                                       tmf_adam adam, void* stream) {
    if (n_rows == 0) return TMF_OK;
    const RowGeom geom = row_geom(n_components);
    TMF_REQUIRE(geom.ld > 0, "adam_state_rows: unsupported n_components %d", n_components);
    TMF_REQUIRE(W && G && M && V && n_rows > 0, "adam_state_rows: bad arguments");
    const int64_t n4 = n_rows * geom.ld / 4;
    const int64_t want = (n4 + 255) / 256;
    hipLaunchKernelGGL(k_adam_state_rows, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4*>(W), reinterpret_cast<const float4*>(G), reinterpret_cast<float4*>(M),
                       reinterpret_cast<float4*>(V), n4, adam);
    return check_launch("tmf_adam_state_rows_f32");
}

extern "C" int tmf_sum_f32(const float* x, int64_t n, double* out, void* stream) {
    TMF_REQUIRE(out && n >= 0 && (n == 0 || x), "sum: bad arguments");
    hipLaunchKernelGGL(k_sum_f32, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, out);
    return check_launch("tmf_sum_f32");
}
