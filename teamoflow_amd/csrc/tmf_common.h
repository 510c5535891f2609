// Shared device helpers for libtmf (gfx950 only: 64-lane waves, 16-byte lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <type_traits>

#include "../../include/tmf.h"

namespace tmf {

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define TMF_REQUIRE(cond, ...)             \
    do {                                   \
        if (!(cond)) {                     \
            tmf::set_error(__VA_ARGS__);   \
            return TMF_E_INVALID;          \
        }                                  \
    } while (0)

// A HIP launch carries its global size (blocks x threads) in 32 bits: beyond 2^32 work-items the grid is silently truncated
// (found in round 3: 78125 user groups x 256 slices x 256 threads "ran" in a third of the time and skipped most of the work).
inline bool launch_fits(uint64_t blocks, unsigned threads) {
    return blocks > 0 && blocks < (1ull << 31) && blocks * threads < (1ull << 32);
}
#define TMF_REQUIRE_LAUNCH(blocks, threads, what)                                                                      \
    TMF_REQUIRE(tmf::launch_fits((uint64_t)(blocks), (unsigned)(threads)), "%s: %llu blocks of %u threads exceed one launch", \
                what, (unsigned long long)(blocks), (unsigned)(threads))

// Kernels that ask for more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize, which applies to
// the CURRENT device only: remember what was granted per device (one LdsGrant per kernel instance; a process may
// drive several devices, and the call is cheap enough to repeat when two host threads race on a first use).
struct LdsGrant {
    size_t allowed[32];
    LdsGrant() { for (size_t& a : allowed) a = 64 * 1024; }
};
inline int grant_dynamic_lds(const void* func, size_t lds, LdsGrant& g) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = -1;
    if (dev >= 0 && lds <= g.allowed[dev]) return TMF_OK;
    const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
        return TMF_E_LAUNCH;
    }
    if (dev >= 0) g.allowed[dev] = lds;
    return TMF_OK;
}

// ---------------------------------------------------------------------------------------------
// Row geometry.  A factor row of `ld` floats is read by a GROUP of G lanes, NV float4 per lane:
// lane g of the group holds floats [4*(g + G*v), 4*(g + G*v) + 4) for v < NV, so every load
// instruction of a group is one contiguous 16*G-byte piece of the row.  64/G groups per wave work
// on different list entries at once.
// ---------------------------------------------------------------------------------------------
struct RowGeom {
    int G, NV, ld;
};
inline RowGeom row_geom(int r) {
    RowGeom g{0, 0, 0};
    if (r < 1 || r > 1024) return g;
    if (r <= 256) {
        int lanes = (r + 3) / 4, G = 1;
        while (G < lanes) G <<= 1;
        g = {G, 1, 4 * G};
    } else {
        int NV = (r + 255) / 256;
        if (NV == 3) NV = 4;
        g = {64, NV, 256 * NV};
    }
    return g;
}

// bf16 storage: 8 elements (16 bytes) per lane and fragment pair.
inline RowGeom row_geom_bf16(int r) {
    RowGeom g{0, 0, 0};
    if (r < 1 || r > 1024) return g;
    if (r <= 512) {
        int lanes = (r + 7) / 8, G = 1;
        while (G < lanes) G <<= 1;
        g = {G, 2, 8 * G};
    } else {
        g = {64, 4, 1024};
    }
    return g;
}

template <typename T>
inline RowGeom row_geom_of(int r);
template <>
inline RowGeom row_geom_of<float>(int r) { return row_geom(r); }
template <>
inline RowGeom row_geom_of<__bf16>(int r) { return row_geom_bf16(r); }

// Dispatch a callable templated on <G, NV> for the geometry of rank r.
#define TMF_DISPATCH_GEOM(geom, CALL)                       \
    switch ((geom).G * 8 + (geom).NV) {                     \
        case 1 * 8 + 1: { CALL(1, 1); } break;              \
        case 2 * 8 + 1: { CALL(2, 1); } break;              \
        case 4 * 8 + 1: { CALL(4, 1); } break;              \
        case 8 * 8 + 1: { CALL(8, 1); } break;              \
        case 16 * 8 + 1: { CALL(16, 1); } break;            \
        case 32 * 8 + 1: { CALL(32, 1); } break;            \
        case 64 * 8 + 1: { CALL(64, 1); } break;            \
        case 64 * 8 + 2: { CALL(64, 2); } break;            \
        case 64 * 8 + 4: { CALL(64, 4); } break;            \
        default: tmf::set_error("unsupported n_components"); \
            return TMF_E_UNSUPPORTED;                       \
    }

#define TMF_DISPATCH_GEOM_BF16(geom, CALL)                  \
    switch ((geom).G * 8 + (geom).NV) {                     \
        case 1 * 8 + 2: { CALL(1, 2); } break;              \
        case 2 * 8 + 2: { CALL(2, 2); } break;              \
        case 4 * 8 + 2: { CALL(4, 2); } break;              \
        case 8 * 8 + 2: { CALL(8, 2); } break;              \
        case 16 * 8 + 2: { CALL(16, 2); } break;            \
        case 32 * 8 + 2: { CALL(32, 2); } break;            \
        case 64 * 8 + 2: { CALL(64, 2); } break;            \
        case 64 * 8 + 4: { CALL(64, 4); } break;            \
        default: tmf::set_error("unsupported n_components"); \
            return TMF_E_UNSUPPORTED;                       \
    }

// selection of the dispatch table by storage type (use inside a function templated on T_)
#define TMF_DISPATCH(T_, geom, CALL)                                                 \
    if constexpr (std::is_same<T_, float>::value) { TMF_DISPATCH_GEOM(geom, CALL); } \
    else { TMF_DISPATCH_GEOM_BF16(geom, CALL); }

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
template <int NV>
struct Frag {
    float4 v[NV];
};

// Storage types of a factor table: float, or __bf16 (bf16 storage / fp32 arithmetic, BASELINE config 5).
// A lane's registers always hold NV float4; what differs is which elements of the row they are:
//   float : fragment v = elements [4 (g + G v), +4)                     (one 16-byte load per fragment)
//   __bf16: fragments 2p, 2p+1 = elements [8 (g + G p), +8)             (one 16-byte load per PAIR)
// so that every load instruction of a group is one contiguous 16*G-byte piece of the row either way.
// fp32 side arrays that belong to a table of storage type T (slab partials, raw gradients) are
// written in natural element order through the same element map (load/store_row_f32).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

template <int G, int NV, typename T>
struct RowMap;
template <int G, int NV>
struct RowMap<G, NV, float> {
    static __device__ __forceinline__ int off(int v, int g) { return 4 * (g + G * v); }
};
template <int G, int NV>
struct RowMap<G, NV, __bf16> {
    static_assert(NV % 2 == 0, "bf16 rows are loaded in pairs of fragments");
    static __device__ __forceinline__ int off(int v, int g) { return 8 * (g + G * (v >> 1)) + 4 * (v & 1); }
};

typedef float tmf_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_f4_nt(const float* p) {
    const tmf_f4 t = __builtin_nontemporal_load(reinterpret_cast<const tmf_f4*>(p));
    return make_float4(t[0], t[1], t[2], t[3]);
}

template <int G, int NV>
__device__ __forceinline__ void load_row(Frag<NV>& f, const float* __restrict__ T, int64_t row, int g) {
    const float4* p = reinterpret_cast<const float4*>(T + row * (int64_t)(4 * G * NV)) + g;
#pragma unroll
    for (int v = 0; v < NV; ++v) f.v[v] = p[G * v];
}

template <int G, int NV>
__device__ __forceinline__ void load_row(Frag<NV>& f, const __bf16* __restrict__ T, int64_t row, int g) {
    const bf16x8* p = reinterpret_cast<const bf16x8*>(T + row * (int64_t)(4 * G * NV)) + g;
#pragma unroll
    for (int pv = 0; pv < NV / 2; ++pv) {
        const f32x8 w = __builtin_convertvector(p[G * pv], f32x8);
        f.v[2 * pv] = make_float4(w[0], w[1], w[2], w[3]);
        f.v[2 * pv + 1] = make_float4(w[4], w[5], w[6], w[7]);
    }
}

// A row that is read ONCE by the kernel (the user's own row of a (user, slice) visit): non-temporal, so that it does not displace
// the rows the kernel gathers again and again in the L2s (scores 27.5 -> 26.9 ms at C4).
template <int G, int NV>
__device__ __forceinline__ void load_row_nt(Frag<NV>& f, const float* __restrict__ T, int64_t row, int g) {
    const float* p = T + row * (int64_t)(4 * G * NV) + 4 * g;
#pragma unroll
    for (int v = 0; v < NV; ++v) f.v[v] = load_f4_nt(p + 4 * G * v);
}
template <int G, int NV>
__device__ __forceinline__ void load_row_nt(Frag<NV>& f, const __bf16* __restrict__ T, int64_t row, int g) {
    const bf16x8* p = reinterpret_cast<const bf16x8*>(T + row * (int64_t)(4 * G * NV)) + g;
#pragma unroll
    for (int pv = 0; pv < NV / 2; ++pv) {
        const f32x8 w = __builtin_convertvector(__builtin_nontemporal_load(p + G * pv), f32x8);
        f.v[2 * pv] = make_float4(w[0], w[1], w[2], w[3]);
        f.v[2 * pv + 1] = make_float4(w[4], w[5], w[6], w[7]);
    }
}

// What the load instructions of a row return, before any arithmetic touches it.  The gather loops fill kUnroll Raw
// registers (row or zeros) first and convert afterwards: a bf16 -> fp32 conversion placed right after a load makes the
// compiler wait for that load before issuing the next one (one row in flight per lane group instead of kUnroll).
template <int NV, typename T>
struct Raw;
template <int NV>
struct Raw<NV, float> {
    float4 v[NV];
};
template <int NV>
struct Raw<NV, __bf16> {
    bf16x8 v[NV / 2];
};
template <int G, int NV>
__device__ __forceinline__ void load_raw(Raw<NV, float>& r, const float* __restrict__ T, int64_t row, int g) {
    const float4* p = reinterpret_cast<const float4*>(T + row * (int64_t)(4 * G * NV)) + g;
#pragma unroll
    for (int v = 0; v < NV; ++v) r.v[v] = p[G * v];
}
template <int G, int NV>
__device__ __forceinline__ void load_raw(Raw<NV, __bf16>& r, const __bf16* __restrict__ T, int64_t row, int g) {
    const bf16x8* p = reinterpret_cast<const bf16x8*>(T + row * (int64_t)(4 * G * NV)) + g;
#pragma unroll
    for (int pv = 0; pv < NV / 2; ++pv) r.v[pv] = p[G * pv];
}
template <int NV>
__device__ __forceinline__ void zero_raw(Raw<NV, float>& r) {
#pragma unroll
    for (int v = 0; v < NV; ++v) r.v[v] = make_float4(0.f, 0.f, 0.f, 0.f);
}
template <int NV>
__device__ __forceinline__ void zero_raw(Raw<NV, __bf16>& r) {
#pragma unroll
    for (int pv = 0; pv < NV / 2; ++pv)
#pragma unroll
        for (int e = 0; e < 8; ++e) r.v[pv][e] = (__bf16)0.0f;
}
template <int NV>
__device__ __forceinline__ void to_frag(Frag<NV>& f, const Raw<NV, float>& r) {
#pragma unroll
    for (int v = 0; v < NV; ++v) f.v[v] = r.v[v];
}
template <int NV>
__device__ __forceinline__ void to_frag(Frag<NV>& f, const Raw<NV, __bf16>& r) {
#pragma unroll
    for (int pv = 0; pv < NV / 2; ++pv) {
        const f32x8 w = __builtin_convertvector(r.v[pv], f32x8);
        f.v[2 * pv] = make_float4(w[0], w[1], w[2], w[3]);
        f.v[2 * pv + 1] = make_float4(w[4], w[5], w[6], w[7]);
    }
}

// ---------------------------------------------------------------------------------------------
// Helpers of the row-stationary scores kernel (k_wmrb_scores5, tmf_wmrb.hip): rows addressed with 32-bit byte offsets from a
// wave-uniform base, dot products on packed pairs, eight scores reduced at once.
// ---------------------------------------------------------------------------------------------
typedef float tmf_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int G, int NV, typename T>
struct RowBytes {
    static constexpr uint32_t value = 4u * G * NV * (uint32_t)sizeof(T);
};

// row * row_bytes + lane_off in ONE instruction (row_bytes is a power of two and lane_off < row_bytes, so the sum is an OR; the
// compiler's own choice for the same expression is v_lshlrev_b32 + v_or_b32)
template <uint32_t ROW_BYTES>
__device__ __forceinline__ uint32_t row_byte_off(uint32_t row, uint32_t lane_off) {
    static_assert((ROW_BYTES & (ROW_BYTES - 1)) == 0, "rows are a power of two bytes long");
    uint32_t off;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(off) : "v"(row), "n"(__builtin_ctz(ROW_BYTES)), "v"(lane_off));
    return off;
}

// Row `row` of a table of fewer than 2^32 bytes whose base is wave-uniform: global_load ... v_off, s[base:base+1]
template <int G, int NV>
__device__ __forceinline__ void load_raw32(Raw<NV, float>& r, const float* __restrict__ T, uint32_t row, uint32_t lane_off) {
    const char* p = reinterpret_cast<const char*>(T) + row_byte_off<RowBytes<G, NV, float>::value>(row, lane_off);
#pragma unroll
    for (int v = 0; v < NV; ++v) r.v[v] = *reinterpret_cast<const float4*>(p + 16 * G * v);
}
template <int G, int NV>
__device__ __forceinline__ void load_raw32(Raw<NV, __bf16>& r, const __bf16* __restrict__ T, uint32_t row, uint32_t lane_off) {
    const char* p = reinterpret_cast<const char*>(T) + row_byte_off<RowBytes<G, NV, __bf16>::value>(row, lane_off);
#pragma unroll
    for (int pv = 0; pv < NV / 2; ++pv) r.v[pv] = *reinterpret_cast<const bf16x8*>(p + 16 * G * pv);
}

// This lane's share of <x, y>, both rows as loaded (Raw).  fp32: two independent chains on the (x, y) and (z, w) pairs, added at
// the end (v_pk_mul / v_pk_fma / v_pk_add); bf16: v_dot2c_f32_bf16 - exact products, fp32 accumulation, no conversions.
// Another summation order than dot_partial: the two agree to rounding, not to the bit (and exactly on dyadic data).
template <int NV>
__device__ __forceinline__ float dot_raw(const Raw<NV, float>& x, const Raw<NV, float>& y) {
    tmf_f2 a = tmf_f2{x.v[0].x, x.v[0].y} * tmf_f2{y.v[0].x, y.v[0].y};
    a = __builtin_elementwise_fma(tmf_f2{x.v[0].z, x.v[0].w}, tmf_f2{y.v[0].z, y.v[0].w}, a);
#pragma unroll
    for (int v = 1; v < NV; ++v) {
        a = __builtin_elementwise_fma(tmf_f2{x.v[v].x, x.v[v].y}, tmf_f2{y.v[v].x, y.v[v].y}, a);
        a = __builtin_elementwise_fma(tmf_f2{x.v[v].z, x.v[v].w}, tmf_f2{y.v[v].z, y.v[v].w}, a);
    }
    return a.x + a.y;
}
template <int NV>
__device__ __forceinline__ float dot_raw(const Raw<NV, __bf16>& x, const Raw<NV, __bf16>& y) {
    float s = 0.f;
#pragma unroll
    for (int pv = 0; pv < NV / 2; ++pv)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            s = __builtin_amdgcn_fdot2_f32_bf16(bf16x2{x.v[pv][2 * i], x.v[pv][2 * i + 1]}, bf16x2{y.v[pv][2 * i], y.v[pv][2 * i + 1]}, s, false);
    return s;
}

#define TMF_DPP_MOV(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xf, 0xf, false))

// Eight per-lane partial sums p[0..7] (eight gathered rows) -> their eight sums over a 32-lane group, TRANSPOSED: afterwards lane l
// holds the complete sum of entry entry_of_lane(l) (every entry on 4 lanes).  v_permlane16_swap exchanges halves of two registers
// (one add then finishes the xor-16 level for BOTH entries), row_ror:8 is a true xor-8 butterfly followed by a select; only the
// quad levels run on a register that already holds four entries: 25 instructions per 8 sums instead of 8 x 7.
// A fixed summation tree, so results are bit-reproducible; every lane that holds entry e holds the same bits.
struct Reduce8x32 {
    static __device__ __forceinline__ int entry_of_lane(int lane) { return ((lane >> 4) & 1) | ((lane >> 2) & 2) | (lane & 4); }
    static __device__ __forceinline__ bool owner(int lane) { return (lane & 3) == 0; }
    static __device__ __forceinline__ float run(const float (&p)[8], int lane) {
        float q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // xor 16: lanes with bit 4 clear keep p[2i], the others p[2i + 1]
            const auto s = __builtin_amdgcn_permlane16_swap(__float_as_int(p[2 * i]), __float_as_int(p[2 * i + 1]), false, false);
            q[i] = __int_as_float(s[0]) + __int_as_float(s[1]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] += TMF_DPP_MOV(q[i], 0x128);   // row_ror:8 = xor 8
        const bool b3 = lane & 8, b2 = lane & 4;
        float r0 = b3 ? q[1] : q[0], r1 = b3 ? q[3] : q[2];
        r0 += TMF_DPP_MOV(r0, 0xB1);    // quad_perm [1,0,3,2]
        r1 += TMF_DPP_MOV(r1, 0xB1);
        r0 += TMF_DPP_MOV(r0, 0x4E);    // quad_perm [2,3,0,1]
        r1 += TMF_DPP_MOV(r1, 0x4E);
        r0 += TMF_DPP_MOV(r0, 0x141);   // row_half_mirror (quads are uniform by now)
        r1 += TMF_DPP_MOV(r1, 0x141);
        return b2 ? r1 : r0;
    }
};

// The same for FOUR partial sums (10 instructions): afterwards lane l holds the sum of entry ((l >> 4) & 1) | ((l >> 2) & 2), every
// entry on 8 lanes.  Four rows in flight keep the scores kernel at 8 waves per SIMD (eight cost it two of them: profiles/r04_*).
struct Reduce4x32 {
    static __device__ __forceinline__ int entry_of_lane(int lane) { return ((lane >> 4) & 1) | ((lane >> 2) & 2); }
    static __device__ __forceinline__ bool owner(int lane) { return (lane & 7) == 0; }
    static __device__ __forceinline__ float run(const float (&p)[4], int lane) {
        float q[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const auto s = __builtin_amdgcn_permlane16_swap(__float_as_int(p[2 * i]), __float_as_int(p[2 * i + 1]), false, false);
            q[i] = __int_as_float(s[0]) + __int_as_float(s[1]);
        }
        q[0] += TMF_DPP_MOV(q[0], 0x128);
        q[1] += TMF_DPP_MOV(q[1], 0x128);
        float r = (lane & 8) ? q[1] : q[0];
        r += TMF_DPP_MOV(r, 0xB1);
        r += TMF_DPP_MOV(r, 0x4E);
        r += TMF_DPP_MOV(r, 0x141);
        return r;
    }
};

// Row stores.  Every row this engine writes (new table rows, slab partials, raw gradients, per-slice partials) is written
// once and read by a LATER kernel, so the stores carry the non-temporal hint.  Measured on one box, same run: the hint
// itself changes nothing (item pass at C4 36.8 ms without, 36.6 ms with); what took that kernel from 44.0 to 36.8 ms was
// the code the compiler emits around this helper - with the previous `*(float4*)p = v` form it waited for each of the
// four unrolled row loads of the gather loop before issuing the next (one row in flight per lane group).  Check the
// ISA of k_wsum_pass for `global_load_dwordx4` directly followed by `s_waitcnt vmcnt(0)` after touching this file
// (profiles/r01_sliced_user_pass.txt has the same-box numbers, including the variant that loads into raw registers and
// converts / selects afterwards in every gather loop: slower on C4, 3 % faster on the config-5 shard - not kept).
__device__ __forceinline__ void store_f4_nt(float* p, const float4& v) {
    const tmf_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<tmf_f4*>(p));
}

template <int G, int NV>
__device__ __forceinline__ void store_row(const Frag<NV>& f, float* __restrict__ T, int64_t row, int g) {
    float* p = T + row * (int64_t)(4 * G * NV) + 4 * g;
#pragma unroll
    for (int v = 0; v < NV; ++v) store_f4_nt(p + 4 * G * v, f.v[v]);
}

template <int G, int NV>
__device__ __forceinline__ void store_row(const Frag<NV>& f, __bf16* __restrict__ T, int64_t row, int g) {
    bf16x8* p = reinterpret_cast<bf16x8*>(T + row * (int64_t)(4 * G * NV)) + g;
#pragma unroll
    for (int pv = 0; pv < NV / 2; ++pv) {
        f32x8 w;
        w[0] = f.v[2 * pv].x; w[1] = f.v[2 * pv].y; w[2] = f.v[2 * pv].z; w[3] = f.v[2 * pv].w;
        w[4] = f.v[2 * pv + 1].x; w[5] = f.v[2 * pv + 1].y; w[6] = f.v[2 * pv + 1].z; w[7] = f.v[2 * pv + 1].w;
        __builtin_nontemporal_store(__builtin_convertvector(w, bf16x8), p + G * pv);  // round-to-nearest-even (v_cvt_pk_bf16_f32)
    }
}

// Partial rows (slab slots, per-slice layers) are written once and read once by the kernel that sums them: that read carries the
// non-temporal hint too (combine 1.42 -> 1.30 ms, finish 1.30 -> 1.28 at C4; the same hint on slice offsets or on the hinge
// kernel's score loads changed nothing, on the item pass's entry lists it cost 2 ms - profiles/r03_c5_experiments.txt item 10).
#ifndef TMF_NT_FIN
#define TMF_NT_FIN 1
#endif
template <int G, int NV, typename T, bool NT = false>
__device__ __forceinline__ void load_row_f32(Frag<NV>& f, const float* __restrict__ B, int64_t row, int g) {
    const float* p = B + row * (int64_t)(4 * G * NV);
#pragma unroll
    for (int v = 0; v < NV; ++v)
        f.v[v] = NT ? load_f4_nt(p + RowMap<G, NV, T>::off(v, g)) : *reinterpret_cast<const float4*>(p + RowMap<G, NV, T>::off(v, g));
}

template <int G, int NV, typename T>
__device__ __forceinline__ void store_row_f32(const Frag<NV>& f, float* __restrict__ B, int64_t row, int g) {
    float* p = B + row * (int64_t)(4 * G * NV);
#pragma unroll
    for (int v = 0; v < NV; ++v) store_f4_nt(p + RowMap<G, NV, T>::off(v, g), f.v[v]);
}

template <int NV>
__device__ __forceinline__ void zero(Frag<NV>& f) {
#pragma unroll
    for (int v = 0; v < NV; ++v) f.v[v] = make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int NV>
__device__ __forceinline__ float dot_partial(const Frag<NV>& a, const Frag<NV>& b) {
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        s = fmaf(a.v[v].x, b.v[v].x, s);
        s = fmaf(a.v[v].y, b.v[v].y, s);
        s = fmaf(a.v[v].z, b.v[v].z, s);
        s = fmaf(a.v[v].w, b.v[v].w, s);
    }
    return s;
}

template <int NV>
__device__ __forceinline__ void axpy(Frag<NV>& acc, float w, const Frag<NV>& x) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        acc.v[v].x = fmaf(w, x.v[v].x, acc.v[v].x);
        acc.v[v].y = fmaf(w, x.v[v].y, acc.v[v].y);
        acc.v[v].z = fmaf(w, x.v[v].z, acc.v[v].z);
        acc.v[v].w = fmaf(w, x.v[v].w, acc.v[v].w);
    }
}

template <int NV>
__device__ __forceinline__ void add(Frag<NV>& acc, const Frag<NV>& x) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        acc.v[v].x += x.v[v].x;
        acc.v[v].y += x.v[v].y;
        acc.v[v].z += x.v[v].z;
        acc.v[v].w += x.v[v].w;
    }
}

// LDS written by some lanes of a wave and read by others of the SAME wave: the hardware completes a wave's LDS
// operations in issue order, so no s_barrier is needed - this only keeps the compiler from moving the accesses
// across the hand-over (it emits no instruction beyond the s_waitcnt the reads need anyway).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// v of lane `idx` of this lane group (ds_bpermute: every lane of the wave has to execute it)
template <int G>
__device__ __forceinline__ int group_read(int v, int idx) {
    return __shfl(v, (threadIdx.x & 63 & ~(G - 1)) + idx, 64);
}

// Soft rendezvous of the workgroups of a launch, once per step i (speed only - no data is handed over, so no fences):
// "I have finished step i" is one relaxed agent-scope add; before going on, wait until every workgroup of this block's XCD
// lane (blocks b and b + 8 share an XCD under the observed round-robin placement, and a window has to stay resident per XCD
// L2 only) has finished step i - lag, so those workgroups are spread over at most lag + 1 windows.  One counter per
// (step, lane), each on a 64-byte line of its own.  The wait is BOUNDED: if a workgroup of the launch is not resident (or
// the counters are stale) the others give up after ~50 us and run on, unsynchronised but correct - a launch never hangs.
constexpr int kSyncStride = 16;   // ints between counters
__device__ __forceinline__ void step_rendezvous(int* sync, int i, int lag, int n_groups) {
    const int x = blockIdx.x & 7;
    const int mine = (n_groups - x + 7) / 8;   // workgroups of this lane
    __hip_atomic_fetch_add(sync + (i * 8 + x) * kSyncStride, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int need = i - lag;
    if (need < 0) return;
    for (int spin = 0; spin < 256; ++spin) {
        if (__hip_atomic_load(sync + (need * 8 + x) * kSyncStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= mine) return;
        __builtin_amdgcn_s_sleep(8);
    }
}
inline size_t rendezvous_bytes(int64_t launches, int64_t steps) {
    return (size_t)launches * (size_t)steps * 8 * kSyncStride * sizeof(int);
}

// Sum over the G lanes of a group (lanes [G*k, G*k+G)); every lane of the group ends with the same bits
// (x+y == y+x at every level), so the value can be used as a per-entry weight by all of them.
// Pure VALU: DPP quad permutes (lane^1, lane^2), row_half_mirror / row_mirror (the other quad / the other
// half-row already hold uniform sums, so a mirror is as good as an xor) and the gfx950 half-row / half-wave
// swaps for 16 and 32 - no LDS traffic (ds_bpermute), which the gather kernels need for their staging.
#define TMF_DPP_ADD(v, ctrl) \
    ((v) + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xf, 0xf, false)))

template <int G>
__device__ __forceinline__ float group_allsum(float v) {
    if (G >= 2) v = TMF_DPP_ADD(v, 0xB1);   // quad_perm [1,0,3,2]
    if (G >= 4) v = TMF_DPP_ADD(v, 0x4E);   // quad_perm [2,3,0,1]
    if (G >= 8) v = TMF_DPP_ADD(v, 0x141);  // row_half_mirror
    if (G >= 16) v = TMF_DPP_ADD(v, 0x140); // row_mirror
    if (G >= 32) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
        v = __int_as_float(r[0]) + __int_as_float(r[1]);
    }
    if (G >= 64) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
        v = __int_as_float(r[0]) + __int_as_float(r[1]);
    }
    return v;
}

// Sum of a per-lane value over the 64/G groups of a wave (same lane-in-group across groups).
template <int G>
__device__ __forceinline__ float across_groups_sum(float v) {
#pragma unroll
    for (int off = 32; off >= G; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int G, int NV>
__device__ __forceinline__ void across_groups_sum(Frag<NV>& f) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        f.v[v].x = across_groups_sum<G>(f.v[v].x);
        f.v[v].y = across_groups_sum<G>(f.v[v].y);
        f.v[v].z = across_groups_sum<G>(f.v[v].z);
        f.v[v].w = across_groups_sum<G>(f.v[v].w);
    }
}

// Fresh Keras-Adam step (t = 1, zero moments), unsimplified fp32 op sequence of SURVEY.md A.1:
//   m = (g - 0)(1-b1); v = (g*g - 0)(1-b2); w -= (m*alpha)/(sqrt(v)+eps)
// __fsqrt_rn / __fdiv_rn keep IEEE rounding whatever -ffast-math-like flags the build uses.
__device__ __forceinline__ float adam_fresh(float w, float g, const tmf_adam a) {
    const float m = g * a.one_minus_b1;
    const float v = (g * g) * a.one_minus_b2;
    return w - __fdiv_rn(m * a.alpha, __fsqrt_rn(v) + a.eps);
}

template <int NV>
__device__ __forceinline__ void adam_fresh(Frag<NV>& w, const Frag<NV>& g, const tmf_adam a) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        w.v[v].x = adam_fresh(w.v[v].x, g.v[v].x, a);
        w.v[v].y = adam_fresh(w.v[v].y, g.v[v].y, a);
        w.v[v].z = adam_fresh(w.v[v].z, g.v[v].z, a);
        w.v[v].w = adam_fresh(w.v[v].w, g.v[v].w, a);
    }
}

// Epilogue shared by every row pass: lanes of group 0 write the row.  TMF_EPI_GRAD writes the raw fp32
// gradient (natural element order) whatever the table's storage type.
template <int G, int NV, typename T>
__device__ __forceinline__ void row_epilogue(const Frag<NV>& g, const T* __restrict__ X_old, void* __restrict__ X_out,
                                             int64_t row, int lane_in_group, int epi, const tmf_adam adam) {
    if (epi == TMF_EPI_GRAD) {
        store_row_f32<G, NV, T>(g, static_cast<float*>(X_out), row, lane_in_group);
    } else {
        Frag<NV> w;
        load_row<G, NV>(w, X_old, row, lane_in_group);
        adam_fresh<NV>(w, g, adam);
        store_row<G, NV>(w, static_cast<T*>(X_out), row, lane_in_group);
    }
}

}  // namespace tmf
