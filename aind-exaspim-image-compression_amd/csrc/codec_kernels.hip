// codec_kernels.hip -- block-DCT transform quantiser (the "3-D
// wavelet/DCT quantise" step of BASELINE.json config 5 has no reference implementation; its
// specification is DESIGN.md 3.10 and its checker oracle/exabm4d_oracle.c orc_dctq_*).
//
//   forward: uint16 volume -> non-overlapping 8^3 blocks (edge voxels replicated) -> 3-D DCT
//            (the chains of DESIGN.md 3.5) -> idx = (int32) rintf(c / q) -> [block][512]
//   inverse: idx * q -> inverse DCT -> clamp [0, 65535] -> rintf -> uint16
//
// One wave transforms two x-adjacent blocks at a time as the two streams of the packed-fp32 DCT
// (dct_pairs.h); HBM traffic is the algorithmic 2 B of volume + 4 B of indices per voxel.  Indices are
// integer results: bit-exact against the oracle.
#include "dct_pairs.h"
#include "exabm4d_kernels.h"

namespace exabm4d {

constexpr int CQ_WAVES = 4;
constexpr float DCTQ_MAX = 1073741824.0f;   // 2^30: indices are clamped to +-2^30 (never reached for q >= 0.002)

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

__global__ __launch_bounds__(CQ_WAVES * 64) void dctq_forward_kernel(
    const uint16_t* __restrict__ vol, int nz, int ny, int nx, int nbz, int nby, int nbx, DctTable T,
    float q, int32_t* __restrict__ idx) {
    __shared__ __align__(16) float lds[CQ_WAVES * 2 * TBUF];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hi = lane >> 3, lo = lane & 7;
    f2* tb = reinterpret_cast<f2*>(lds + wave * 2 * TBUF);
    const int pairs_x = (nbx + 1) / 2;
    const long long npairs = (long long)nbz * nby * pairs_x;
    for (long long p = (long long)blockIdx.x * CQ_WAVES + wave; p < npairs;
         p += (long long)gridDim.x * CQ_WAVES) {
        const int px = (int)(p % pairs_x), by = (int)((p / pairs_x) % nby);
        const int bz = (int)(p / ((long long)pairs_x * nby));
        const int bx0 = 2 * px, bx1 = min(2 * px + 1, nbx - 1);
        // layout L1: lane = (z, x), registers = y
        const size_t zrow = (size_t)clampi(8 * bz + hi, 0, nz - 1) * ny;
        const int xa = clampi(8 * bx0 + lo, 0, nx - 1), xb = clampi(8 * bx1 + lo, 0, nx - 1);
        f2 v[8];
#pragma unroll
        for (int y = 0; y < 8; y++) {
            const size_t row = (zrow + clampi(8 * by + y, 0, ny - 1)) * nx;
            v[y] = mk2((float)vol[row + xa], (float)vol[row + xb]);
        }
        pair_fwd(T, tb, hi, lo, v);
        // layout L3: lane = (ux, uy) = (hi, lo), registers = uz; coefficient index (uz, uy, ux)
        int32_t* oa = idx + ((size_t)((size_t)bz * nby + by) * nbx + bx0) * BVOX + lo * 8 + hi;
        int32_t* ob = idx + ((size_t)((size_t)bz * nby + by) * nbx + bx1) * BVOX + lo * 8 + hi;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float ca = rintf(v[u].x / q), cb = rintf(v[u].y / q);
            oa[u * 64] = (int32_t)fminf(fmaxf(ca, -DCTQ_MAX), DCTQ_MAX);
            if (bx1 != bx0) ob[u * 64] = (int32_t)fminf(fmaxf(cb, -DCTQ_MAX), DCTQ_MAX);
        }
    }
}

__global__ __launch_bounds__(CQ_WAVES * 64) void dctq_inverse_kernel(
    const int32_t* __restrict__ idx, int nz, int ny, int nx, int nbz, int nby, int nbx, DctTable T,
    float q, uint16_t* __restrict__ vol) {
    __shared__ __align__(16) float lds[CQ_WAVES * 2 * TBUF];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hi = lane >> 3, lo = lane & 7;
    f2* tb = reinterpret_cast<f2*>(lds + wave * 2 * TBUF);
    const int pairs_x = (nbx + 1) / 2;
    const long long npairs = (long long)nbz * nby * pairs_x;
    for (long long p = (long long)blockIdx.x * CQ_WAVES + wave; p < npairs;
         p += (long long)gridDim.x * CQ_WAVES) {
        const int px = (int)(p % pairs_x), by = (int)((p / pairs_x) % nby);
        const int bz = (int)(p / ((long long)pairs_x * nby));
        const int bx0 = 2 * px, bx1 = min(2 * px + 1, nbx - 1);
        const int32_t* ia = idx + ((size_t)((size_t)bz * nby + by) * nbx + bx0) * BVOX + lo * 8 + hi;
        const int32_t* ib = idx + ((size_t)((size_t)bz * nby + by) * nbx + bx1) * BVOX + lo * 8 + hi;
        f2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = mk2((float)ia[u * 64] * q, (float)ib[u * 64] * q);
        pair_inv(T, tb, hi, lo, v);
        // layout L1 up to tr_x (dct_pairs.h): lane = (z, x = tr_x(hi, lo)), registers = y; only voxels inside
        // the volume are written
        const int z = 8 * bz + hi, xa = 8 * bx0 + tr_x<false>(hi, lo), xb = 8 * bx1 + tr_x<false>(hi, lo);
#pragma unroll
        for (int y = 0; y < 8; y++) {
            const int yy = 8 * by + y;
            if (z < nz && yy < ny) {
                const size_t row = ((size_t)z * ny + yy) * nx;
                if (xa < nx)
                    vol[row + xa] = (uint16_t)(int)rintf(fminf(fmaxf(v[y].x, 0.0f), 65535.0f));
                if (bx1 != bx0 && xb < nx)
                    vol[row + xb] = (uint16_t)(int)rintf(fminf(fmaxf(v[y].y, 0.0f), 65535.0f));
            }
        }
    }
}

static inline unsigned dctq_blocks(int nbz, int nby, int nbx) {
    const long long npairs = (long long)nbz * nby * ((nbx + 1) / 2);
    long long b = (npairs + CQ_WAVES - 1) / CQ_WAVES;
    if (b > 16384) b = 16384;
    return (unsigned)(b ? b : 1);
}

hipError_t launch_dctq_forward(const uint16_t* vol, int nz, int ny, int nx, const float* dct64, float q,
                               int32_t* idx, hipStream_t s) {
    DctTable T;
    for (int i = 0; i < 64; i++) T.d[i] = dct64[i];
    const int nbz = (nz + 7) / 8, nby = (ny + 7) / 8, nbx = (nx + 7) / 8;
    hipLaunchKernelGGL(dctq_forward_kernel, dim3(dctq_blocks(nbz, nby, nbx)), dim3(CQ_WAVES * 64), 0, s,
                       vol, nz, ny, nx, nbz, nby, nbx, T, q, idx);
    return hipGetLastError();
}
hipError_t launch_dctq_inverse(const int32_t* idx, int nz, int ny, int nx, const float* dct64, float q,
                               uint16_t* vol, hipStream_t s) {
    DctTable T;
    for (int i = 0; i < 64; i++) T.d[i] = dct64[i];
    const int nbz = (nz + 7) / 8, nby = (ny + 7) / 8, nbx = (nx + 7) / 8;
    hipLaunchKernelGGL(dctq_inverse_kernel, dim3(dctq_blocks(nbz, nby, nbx)), dim3(CQ_WAVES * 64), 0, s,
                       idx, nz, ny, nx, nbz, nby, nbx, T, q, vol);
    return hipGetLastError();
}

}  // namespace exabm4d
