// bm_kernels.hip -- 3-D block matching (SURVEY.md section 8 row a-B1).
//
// No reference source exists for this arithmetic (the reference calls the closed bm4d wheel,
// machine_learning/data_handling.py:332); the spec is DESIGN.md section 3 and the checker is
// oracle/exabm4d_oracle.c:orc_blockmatch.
//
// Distance: a block is 2x2x2 cells of 4^3 voxels.  cell SSD = 64-term fmaf chain
// in (z,y,x) raster order from +0; block SSD = pairwise tree over the 8 cells.  Cell SSDs of
// grid-aligned cells are shared by the 8 reference blocks that contain the cell, which is what
// bm_tile_kernel exploits: one lane owns one cell, cell sums go through LDS, one lane owns one
// reference block's running top-16 list.
#include "exabm4d_kernels.h"

namespace exabm4d {

// ------------------------------------------------------------------------------------------------
// Tile kernel: 512 lanes = 8x8x8 cells; tiles overlap by one cell so a tile yields 7x7x7
// grid-aligned reference blocks.
// ------------------------------------------------------------------------------------------------
// 16-byte load from a 4-byte aligned address (gfx950 global loads need only dword alignment;
// hipcc emits global_load_dwordx4 for this type).
struct __attribute__((packed, aligned(4))) float4u {
    float x, y, z, w;
};

constexpr int TC = 8;        // cells per tile edge
constexpr int TR = TC - 1;   // reference blocks per tile edge

template <bool WIDE>
__device__ __forceinline__ void bm_tile_loop(const float* __restrict__ vol, const VolGeom& g,
                                             uint32_t keymax, const float (&A)[64],
                                             const int (&xo)[14], int qz, int qy, int qx, int rz,
                                             int ry, int rx, bool ref_ok, int tid, size_t sy,
                                             size_t sz, float (*cs)[TC * TC * TC],
                                             uint32_t (&list)[MAXG]) {
    for (int dzi = 0; dzi < SWIN; dzi++) {
        const int dz = dzi - RAD;
        const bool vz = (rz + dz >= 0) && (rz + dz <= g.nz - BLK);
        for (int dyi = 0; dyi < SWIN; dyi++) {
            const int dy = dyi - RAD;
            const bool vzy = vz && (ry + dy >= 0) && (ry + dy <= g.ny - BLK);

            float acc[SWIN];
#pragma unroll
            for (int d = 0; d < SWIN; d++) acc[d] = 0.0f;

#pragma unroll
            for (int z = 0; z < 4; z++) {
                const int wz = min(max(qz + z + dz, 0), g.nz - 1);
#pragma unroll
                for (int y = 0; y < 4; y++) {
                    const int wy = min(max(qy + y + dy, 0), g.ny - 1);
                    const float* __restrict__ rowp = vol + (size_t)wz * sz + (size_t)wy * sy;
                    float w[16];
                    if (WIDE) {
                        const float4u* q = reinterpret_cast<const float4u*>(rowp + (qx - RAD));
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const float4u t4 = q[j];
                            w[4 * j] = t4.x;
                            w[4 * j + 1] = t4.y;
                            w[4 * j + 2] = t4.z;
                            w[4 * j + 3] = t4.w;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 14; j++) w[j] = rowp[xo[j]];
                    }
#pragma unroll
                    for (int x = 0; x < 4; x++) {
                        const float a = A[(z * 4 + y) * 4 + x];
#pragma unroll
                        for (int d = 0; d < SWIN; d++) {
                            const float t = a - w[d + x];
                            acc[d] = fmaf(t, t, acc[d]);
                        }
                    }
                }
            }

#pragma unroll
            for (int d = 0; d < SWIN; d++) cs[d][tid] = acc[d];
            __syncthreads();

            if (ref_ok) {
#pragma unroll
                for (int d = 0; d < SWIN; d++) {
                    const float* c = &cs[d][tid];
                    const float lo = (c[0] + c[1]) + (c[8] + c[9]);
                    const float hi = (c[64] + c[65]) + (c[72] + c[73]);
                    const float S = lo + hi;
                    const int dx = d - RAD;
                    const bool valid = vzy && (rx + dx >= 0) && (rx + dx <= g.nx - BLK);
                    uint32_t key = (__float_as_uint(S) & KEY_DMASK) | disp_code(dz, dy, dx);
                    key = (valid && key < keymax) ? key : KEY_EMPTY;
                    if (__any(key < list[MAXG - 1])) list_insert(list, key);
                }
            }
            __syncthreads();
        }
    }

}

__global__ __launch_bounds__(512) void bm_tile_kernel(const float* __restrict__ vol_all, VolGeom g,
                                                      uint32_t keymax,
                                                      uint32_t* __restrict__ keys_all, int tiles_y,
                                                      int tiles_x) {
    __shared__ float cs[SWIN][TC * TC * TC];

    const float* __restrict__ vol = vol_all + (size_t)blockIdx.y * (size_t)g.nvox;
    uint32_t* __restrict__ keys = keys_all + (size_t)blockIdx.y * (size_t)g.nref * MAXG;

    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, tz = tile / (tiles_x * tiles_y);
    const int tid = threadIdx.x;
    const int cx = tid & 7, cy = (tid >> 3) & 7, cz = tid >> 6;
    const int ix = TR * tx + cx, iy = TR * ty + cy, iz = TR * tz + cz;  // cell == ref index

    const size_t sy = (size_t)g.nx, sz = (size_t)g.nx * (size_t)g.ny;

    // Cell origin; cells beyond the volume are clamped inside it (their sums are never used).
    const int qz = min(STEP * iz, g.nz - STEP), qy = min(STEP * iy, g.ny - STEP),
              qx = min(STEP * ix, g.nx - STEP);

    // Own cell in registers.
    float A[64];
#pragma unroll
    for (int z = 0; z < 4; z++)
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const float* p = vol + (size_t)(qz + z) * sz + (size_t)(qy + y) * sy + qx;
#pragma unroll
            for (int x = 0; x < 4; x++) A[(z * 4 + y) * 4 + x] = p[x];
        }

    // x offsets of the 14-wide candidate window, clamped into the row.  Clamping only ever
    // affects candidates that lie outside the volume, which are masked below.
    int xo[14];
#pragma unroll
    for (int j = 0; j < 14; j++) xo[j] = min(max(qx - RAD + j, 0), g.nx - 1);
    // Interior cells read the window with four (unaligned) 16-byte loads instead.
    const bool wide = (qx - RAD >= 0) && (qx - RAD + 15 <= g.nx - 1);

    const bool ref_ok = cx < TR && cy < TR && cz < TR && iz < g.az && iy < g.ay && ix < g.ax;
    const int rz = STEP * iz, ry = STEP * iy, rx = STEP * ix;

    uint32_t list[MAXG];
#pragma unroll
    for (int k = 0; k < MAXG; k++) list[k] = KEY_EMPTY;

    // Wave-uniform choice: waves whose every cell lies in the x-interior use the wide-load body.
    if (__all(wide))
        bm_tile_loop<true>(vol, g, keymax, A, xo, qz, qy, qx, rz, ry, rx, ref_ok, tid, sy, sz, cs,
                           list);
    else
        bm_tile_loop<false>(vol, g, keymax, A, xo, qz, qy, qx, rz, ry, rx, ref_ok, tid, sy, sz, cs,
                            list);

    if (ref_ok) {
        uint32_t* out = keys + ((size_t)((size_t)iz * g.gy + iy) * g.gx + ix) * MAXG;
#pragma unroll
        for (int k = 0; k < MAXG; k += 4) {
            uint4 v = make_uint4(list[k], list[k + 1], list[k + 2], list[k + 3]);
            *reinterpret_cast<uint4*>(out + k) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Generic kernel: one wave per reference block, lanes share the 1331 candidates.  Used for the
// clamped last grid position of an axis whose extent is not 8 (mod 4) (e.g. the 54^3 crops of
// evaluate.py:201), and as an independent cross-check of the tile kernel in the parity tests.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void bm_generic_kernel(const float* __restrict__ vol_all,
                                                        VolGeom g, uint32_t keymax,
                                                        uint32_t* __restrict__ keys_all,
                                                        int only_unaligned) {
    __shared__ float rb[BVOX];
    const float* __restrict__ vol = vol_all + (size_t)blockIdx.y * (size_t)g.nvox;
    uint32_t* __restrict__ keys = keys_all + (size_t)blockIdx.y * (size_t)g.nref * MAXG;

    const long long r = blockIdx.x;
    const int ix = (int)(r % g.gx), iy = (int)((r / g.gx) % g.gy),
              iz = (int)(r / ((long long)g.gx * g.gy));
    if (only_unaligned && iz < g.az && iy < g.ay && ix < g.ax) return;
    const int rz = grid_pos(iz, g.az, g.nz), ry = grid_pos(iy, g.ay, g.ny),
              rx = grid_pos(ix, g.ax, g.nx);
    const size_t sy = (size_t)g.nx, sz = (size_t)g.nx * (size_t)g.ny;
    const int lane = threadIdx.x;

    for (int i = lane; i < BVOX; i += 64) {
        const int bx = i & 7, by = (i >> 3) & 7, bz = i >> 6;
        rb[i] = vol[(size_t)(rz + bz) * sz + (size_t)(ry + by) * sy + (rx + bx)];
    }
    __syncthreads();

    uint32_t list[MAXG];
#pragma unroll
    for (int k = 0; k < MAXG; k++) list[k] = KEY_EMPTY;

    for (int c = lane; c < NCAND; c += 64) {
        const int dx = c % SWIN - RAD, dy = (c / SWIN) % SWIN - RAD, dz = c / (SWIN * SWIN) - RAD;
        const int pz = rz + dz, py = ry + dy, px = rx + dx;
        const bool valid = pz >= 0 && pz <= g.nz - BLK && py >= 0 && py <= g.ny - BLK && px >= 0 &&
                           px <= g.nx - BLK;
        uint32_t key = KEY_EMPTY;
        if (valid) {
            const float* __restrict__ b = vol + (size_t)pz * sz + (size_t)py * sy + px;
            float cell[8];
#pragma unroll 1
            for (int kc = 0; kc < 8; kc++) {
                const int kz = kc >> 2, ky = (kc >> 1) & 1, kx = kc & 1;
                float acc = 0.0f;
#pragma unroll 1
                for (int z = 0; z < 4; z++)
#pragma unroll
                    for (int y = 0; y < 4; y++) {
                        const int bz = 4 * kz + z, by = 4 * ky + y;
                        const float* bp = b + (size_t)bz * sz + (size_t)by * sy + 4 * kx;
                        const float* ap = rb + (bz * 8 + by) * 8 + 4 * kx;
#pragma unroll
                        for (int x = 0; x < 4; x++) {
                            const float t = ap[x] - bp[x];
                            acc = fmaf(t, t, acc);
                        }
                    }
                cell[kc] = acc;
            }
            const float lo = (cell[0] + cell[1]) + (cell[2] + cell[3]);
            const float hi = (cell[4] + cell[5]) + (cell[6] + cell[7]);
            const float S = lo + hi;
            key = (__float_as_uint(S) & KEY_DMASK) | disp_code(dz, dy, dx);
            if (key >= keymax) key = KEY_EMPTY;
        }
        list_insert(list, key);
    }

    // 64 sorted lists -> global top 16: repeatedly take the wave-wide minimum head.
    uint32_t* out = keys + (size_t)r * MAXG;
    for (int k = 0; k < MAXG; k++) {
        uint32_t m = list[0];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off));
        if (list[0] == m && m != KEY_EMPTY) {
#pragma unroll
            for (int i = 0; i < MAXG - 1; i++) list[i] = list[i + 1];
            list[MAXG - 1] = KEY_EMPTY;
        }
        if (lane == 0) out[k] = m;
    }
}

// ------------------------------------------------------------------------------------------------
// host launchers (called from exabm4d_api.cpp)
// ------------------------------------------------------------------------------------------------
hipError_t launch_blockmatch(const float* vol, const VolGeom& g, int batch, uint32_t keymax,
                             uint32_t* keys, hipStream_t stream, int force_generic) {
    if (!force_generic && g.az > 0 && g.ay > 0 && g.ax > 0) {
        const int tz = (g.az + TR - 1) / TR, ty = (g.ay + TR - 1) / TR, tx = (g.ax + TR - 1) / TR;
        dim3 grid((unsigned)(tz * ty * tx), (unsigned)batch);
        hipLaunchKernelGGL(bm_tile_kernel, grid, dim3(512), 0, stream, vol, g, keymax, keys, ty, tx);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const bool unaligned = g.gz != g.az || g.gy != g.ay || g.gx != g.ax;
    if (force_generic || unaligned) {
        dim3 grid((unsigned)g.nref, (unsigned)batch);
        hipLaunchKernelGGL(bm_generic_kernel, grid, dim3(64), 0, stream, vol, g, keymax, keys,
                           force_generic ? 0 : 1);
        return hipGetLastError();
    }
    return hipSuccess;
}

}  // namespace exabm4d
