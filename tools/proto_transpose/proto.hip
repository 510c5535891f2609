// Timing prototype (round 5): what would it cost to put the WMRB weights D[u, pos] into the ENTRY ORDER of the
// item pass ((user block, item, user)) once per epoch?  One workgroup per tile = (user block, range of J items whose entries fit LDS):
//   phase 1: every thread takes users of the block: the run D[u, lo .. hi) of the user's negatives that fall into the tile's item
//            range (a handful of floats; lo | source base packed per (tile column, user), read coalesced) goes to LDS in SOURCE order;
//   phase 2: the tile's entries in DESTINATION order read their weight from LDS through a 16-bit local index (streamed) and leave as
//            one contiguous piece of w_ent.
// Not part of libtmf: hipcc -O3 --offload-arch=gfx950 -shared -fPIC proto.hip -o libproto_t.so
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int FORM> __global__ __launch_bounds__(1024) void k_transpose_tiles(
    const float* __restrict__ D, int S, const uint32_t* __restrict__ lo_sb /* [n_jt + 1][m]: lo (low 16 bits) | source base (high 16) */,
    const uint16_t* __restrict__ src_local /* [E] */, const int64_t* __restrict__ tile_ptr /* [n_blocks * n_jt + 1] */,
    float* __restrict__ w_ent, int64_t m, int users_per_block, int n_jt) {
    extern __shared__ float tile[];
    const int b = blockIdx.x / n_jt, jt = blockIdx.x % n_jt;
    const int64_t u0 = (int64_t)b * users_per_block;
    const int nu = (int)((m - u0 < users_per_block) ? m - u0 : users_per_block);
    const uint32_t* c0 = lo_sb + (int64_t)jt * m + u0;
    const uint32_t* c1 = lo_sb + (int64_t)(jt + 1) * m + u0;
    for (int ul = threadIdx.x; ul < nu; ul += blockDim.x) {
        const uint32_t a = c0[ul], nx = c1[ul];
        const int lo = a & 0xffff, hi = nx & 0xffff, sb = a >> 16;
        const float* row = D + (u0 + ul) * (int64_t)S;
        if (FORM == 0) {
            for (int i = lo; i < hi; ++i) tile[sb + (i - lo)] = row[i];
        } else {                                   // FORM 1: the first 8 loads of a run are issued together
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (lo + i < hi) ? row[lo + i] : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) if (lo + i < hi) tile[sb + i] = v[i];
            for (int i = lo + 8; i < hi; ++i) tile[sb + (i - lo)] = row[i];
        }
    }
    __syncthreads();
    const int64_t t0 = tile_ptr[blockIdx.x], t1 = tile_ptr[blockIdx.x + 1];
    for (int64_t e = t0 + threadIdx.x; e < t1; e += blockDim.x) w_ent[e] = tile[src_local[e]];
}

extern "C" int proto_transpose(const float* D, int S, const uint32_t* lo_sb, const uint16_t* src_local, const int64_t* tile_ptr,
                               float* w_ent, int64_t m, int users_per_block, int n_jt, int n_blocks, int lds_bytes, int form, int threads,
                               void* stream) {
    auto k = form ? k_transpose_tiles<1> : k_transpose_tiles<0>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(k, dim3((unsigned)(n_blocks * n_jt)), dim3(threads), lds_bytes, (hipStream_t)stream, D, S, lo_sb, src_local,
                       tile_ptr, w_ent, m, users_per_block, n_jt);
    return (int)hipGetLastError();
}
