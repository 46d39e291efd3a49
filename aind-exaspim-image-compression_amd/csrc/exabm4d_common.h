// exabm4d_common.h -- constants and small device helpers shared by the gfx950 kernels.
// The algorithm these kernels implement is frozen in DESIGN.md section 3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace exabm4d {

constexpr int BLK = 8;          // cubic block edge
constexpr int BVOX = 512;       // voxels per block
constexpr int STEP = 4;         // reference grid step == cell edge
constexpr int RAD = 5;          // search radius
constexpr int SWIN = 11;        // search window edge
constexpr int NCAND = 1331;     // SWIN^3
constexpr int MAXG = 16;        // blocks per group
constexpr uint32_t KEY_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t KEY_DMASK = 0xFFFFF800u;  // distance bits of a match key
constexpr uint32_t KEY_CMASK = 0x000007FFu;  // displacement code bits

// Geometry of one volume and its reference grid (host fills it; passed by value to kernels).
struct VolGeom {
    int nz, ny, nx;          // voxels
    int gz, gy, gx;          // reference-grid points per axis (aligned + optional clamped last)
    int az, ay, ax;          // aligned reference positions per axis: 4*i, i < a*
    long long nvox;          // nz*ny*nx
    long long nref;          // gz*gy*gx
};

__host__ __device__ inline int grid_count(int n) {
    if (n < BLK) return 0;
    int c = (n - BLK) / STEP + 1;
    if ((n - BLK) % STEP) c++;
    return c;
}
__host__ __device__ inline int aligned_count(int n) { return n < BLK ? 0 : (n - BLK) / STEP + 1; }
// voxel position of grid point i along an axis of n voxels with a aligned points
__host__ __device__ inline int grid_pos(int i, int a, int n) { return i < a ? STEP * i : n - BLK; }

__host__ __device__ inline uint32_t disp_code(int dz, int dy, int dx) {
    if (dz == 0 && dy == 0 && dx == 0) return 0u;
    return 1u + (uint32_t)(((dz + RAD) * SWIN + (dy + RAD)) * SWIN + (dx + RAD));
}
__host__ __device__ inline void code_to_disp(uint32_t code, int& dz, int& dy, int& dx) {
    if (code == 0) {
        dz = dy = dx = 0;
        return;
    }
    uint32_t l = code - 1;
    dx = (int)(l % SWIN) - RAD;
    dy = (int)((l / SWIN) % SWIN) - RAD;
    dz = (int)(l / (SWIN * SWIN)) - RAD;
}

// Sorted (ascending) insertion of `key` into a 16-entry register list, dropping the largest.
// For a sorted list L the updated entry is L'[k] = median(L[k-1], key, L[k]) (L[-1] = 0): one
// v_med3_u32 per entry, all reading the OLD neighbours, so the list is updated in place from
// the top down.  (Keys are unique; a key >= L[15] leaves the list unchanged.)
__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) {
    return max(min(a, b), min(max(a, b), c));   // hipcc folds this into v_med3_u32
}
__device__ __forceinline__ void list_insert(uint32_t (&list)[MAXG], uint32_t key) {
#pragma unroll
    for (int k = MAXG - 1; k >= 1; k--) list[k] = med3_u32(list[k - 1], key, list[k]);
    list[0] = min(list[0], key);
}

// The same update strictly in place, top entry first (every v_med3 reads the OLD lower neighbour,
// which has not been rewritten yet).  Written as instructions because hipcc, given the C form
// behind a wave-uniform branch, builds the new list in sixteen other registers and copies it back
// with eight 64-bit moves.
__device__ __forceinline__ void list_insert_inplace(uint32_t (&list)[MAXG], uint32_t key) {
#pragma unroll
    for (int k = MAXG - 1; k >= 1; k--)
        asm volatile("v_med3_u32 %0, %1, %2, %0" : "+v"(list[k]) : "v"(list[k - 1]), "v"(key));
    asm volatile("v_min_u32 %0, %0, %1" : "+v"(list[0]) : "v"(key));
}

// Workgroups of a launch are dealt round-robin to the 8 XCDs, each with its own L2.  Remapping
// the linear workgroup id so that every XCD walks one contiguous range of tiles keeps the halo
// data neighbouring tiles share in the same L2 instead of fetching it once per XCD.
__device__ __forceinline__ int xcd_contiguous(int bid, int nb) {
    const int c = bid & 7, i = bid >> 3, q = nb >> 3, r = nb & 7;
    return c * q + min(c, r) + i;
}

// The same idea with all eight XCDs inside ONE slab of `per` tiles at a time (tile index = slab * per +
// position): XCD c walks positions [c q, (c + 1) q) of every slab, q = ceil(per / 8); positions >= per
// are padding (returns -1).  Launch 8 q workgroups per slab.  The XCDs' working sets then lie in the
// same planes, which the memory-side cache (256 MB) can hold once instead of eight times.
__device__ __forceinline__ int xcd_slab_sync(int bid, int per, int q) {
    const int c = bid & 7, j = bid >> 3;
    const int slab = j / q, w = j - slab * q;
    const int t = c * q + w;
    return t < per ? slab * per + t : -1;
}

}  // namespace exabm4d
