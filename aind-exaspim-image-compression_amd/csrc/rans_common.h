// rans_common.h -- helpers shared by the chunk coders (rans_kernels.hip: EXAC v1 byte planes;
// rans2_kernels.hip: EXAC v2 predictive context model): chunk geometry, row cursor, wave primitives.
#pragma once
#include "exabm4d_kernels.h"

namespace exabm4d {
namespace {

constexpr uint32_t RANS_L = 1u << 15;
constexpr int RANS_BITS = 12;
constexpr uint32_t RANS_M = 1u << RANS_BITS;
constexpr int HDR_TABLE = 32 + 512;   // bitmap + up to 256 frequencies, per plane, in a slot

struct ChunkBox {
    size_t base;        // element index of the chunk's first element in the volume
    int ey, ex;         // extent of this chunk along y, x
    uint32_t n;         // elements in this chunk
};

__device__ __forceinline__ ChunkBox chunk_box(const CodecGeom& g, int c) {
    const int bx = c % g.gx, by = (c / g.gx) % g.gy, bz = c / (g.gx * g.gy);
    const int z0 = bz * g.cz, y0 = by * g.cy, x0 = bx * g.cx;
    const int ez = min(g.cz, g.nz - z0), ey = min(g.cy, g.ny - y0), ex = min(g.cx, g.nx - x0);
    ChunkBox b;
    b.base = ((size_t)z0 * g.ny + y0) * g.nx + x0;
    b.ey = ey;
    b.ex = ex;
    b.n = (uint32_t)ez * (uint32_t)ey * (uint32_t)ex;
    return b;
}

// element offset (in the volume, relative to the chunk's first element) of chunk element i
__device__ __forceinline__ size_t elem_offset(const CodecGeom& g, const ChunkBox& b, uint32_t i) {
    const uint32_t x = i % (uint32_t)b.ex, t = i / (uint32_t)b.ex;
    const uint32_t y = t % (uint32_t)b.ey, z = t / (uint32_t)b.ey;
    return ((size_t)z * g.ny + y) * g.nx + x;
}

// Rows of 64 consecutive chunk elements when the chunk's x extent is a multiple of 64: a row is
// then 64 consecutive elements of one x-row of the volume, and walking the rows forwards or
// backwards only needs three wave-uniform counters (no per-lane division).
struct RowCursor {
    uint32_t xr, y, z;      // 64-element segment inside the x-row, y, z of the current row
    uint32_t rpx, ey;       // segments per x-row, chunk extent along y
    __device__ __forceinline__ void seek(uint32_t r) {
        xr = r % rpx;
        const uint32_t t = r / rpx;
        y = t % ey;
        z = t / ey;
    }
    __device__ __forceinline__ void next() {
        if (++xr == rpx) {
            xr = 0;
            if (++y == ey) {
                y = 0;
                z++;
            }
        }
    }
    __device__ __forceinline__ void prev() {
        if (xr-- == 0) {
            xr = rpx - 1;
            if (y-- == 0) {
                y = ey - 1;
                z--;
            }
        }
    }
    __device__ __forceinline__ size_t offset(const CodecGeom& g) const {
        return ((size_t)z * g.ny + y) * g.nx + xr * 64u;
    }
};

template <int TS>
__device__ __forceinline__ uint32_t load_bits(const void* vol, size_t e) {
    if (TS == 2) return static_cast<const uint16_t*>(vol)[e];
    const int32_t v = static_cast<const int32_t*>(vol)[e];
    return ((uint32_t)v << 1) ^ (uint32_t)(v >> 31);
}

__device__ __forceinline__ uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ uint32_t rank_below(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = max(v, (uint32_t)__shfl_xor(v, o, 64));
    return v;
}
// exclusive prefix sum over the lanes of a wave
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t lane) {
    uint32_t s = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(s, o, 64);
        if (lane >= (uint32_t)o) s += t;
    }
    return s - v;
}

}  // namespace
}  // namespace exabm4d
