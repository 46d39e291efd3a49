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
#include <type_traits>

namespace exabm4d {

// ------------------------------------------------------------------------------------------------
// Tile kernel: 512 lanes = 8x8x8 cells; tiles overlap by one cell in y and x so a tile yields 7x7
// grid-aligned reference blocks per layer, and 7 layers -- or 8 with the carry between tiles (CARRY below).
// ------------------------------------------------------------------------------------------------
// 16-byte load from a 4-byte aligned address (gfx950 global loads need only dword alignment;
// hipcc emits global_load_dwordx4 for this type).
struct __attribute__((packed, aligned(4))) float4u {
    float x, y, z, w;
};

constexpr int TCZ = 8;       // cell layers per tile = waves per workgroup
constexpr int NE = 6;        // dy values per pass (two passes: dy = -5..0 and 1..6, 6 is masked)
constexpr int NSTEP = SWIN * 2 * 4;             // (dz, pass, z) steps
// A wave is one z-layer of TCY x TCX cells (64 lanes).  8 x 8 is the shape for volumes; 4 x 16
// tiles a 64^3 patch (15 reference positions per axis) exactly in x (15) and y (5 x 3), where the
// cube needs 3 x 3 tiles of 7 x 7 positions for the same 15 x 15.
template <int TCY_, int TCX_>
struct TileShape {
    static constexpr int TCY = TCY_, TCX = TCX_;
    static constexpr int TRY = TCY - 1, TRX = TCX - 1;      // reference blocks per tile edge
    static constexpr int PROWS = 4 * (TCY - 1) + 3 + NE;    // staged rows: 4*cy + y + e (37 / 21)
    static constexpr int PCOLS = 4 * (TCX - 1) + 16;        // staged columns: 4*cx + 0..15 (44 / 76)
    // row stride in floats.  8 x 8: 56 = 8 (mod 16) makes the b128 window reads of the two cell
    // rows that share 16 lanes conflict-free; 4 x 16: 16 lanes are one cell row, any stride does.
    static constexpr int PSTR = TCX == 8 ? 56 : PCOLS;
    static constexpr int PCH = PSTR / 4;                    // 16-byte chunks per staged row
    static constexpr int NDMA = (PROWS * PCH + 63) / 64;    // LDS-DMA instructions per plane
    // floats per plane buffer: the staged rows (the tail of the last LDS-DMA instruction is masked off)
    // or the cell-sum exchange, 33 sums x 64 cells + the y-neighbour overhang of the last row
    static constexpr int PLANE = (PROWS * PSTR + 3) / 4 * 4;
    static constexpr int PBUF = PLANE > 2144 ? PLANE : 2144;
    static_assert(TCY * TCX == 64 && PCOLS % 4 == 0 && PSTR % 4 == 0, "one wave per cell layer");
};

// CARRY (round 3).  A reference block is two cell layers, so a tile of eight cell layers gives seven
// reference layers and every eighth layer is computed twice (as the top layer of a tile and as the
// bottom layer of the next): 37 tiles for the 255 reference layers of a 1024^3 volume.  With the carry
// a tile advances by EIGHT layers: wave w computes cell layer L = 8 tz + w and owns the reference layer
// L - 1 -- its lower cells are wave w - 1's sums (LDS), and for wave 0 they are the sums wave 7 of the
// tile BELOW formed, which travel through global memory: [2][columns][22 (dz, pass)][2 rounds][33 x 64],
// a slot per column and tile parity.  `done[column]` (tiles finished) orders producer and consumer: waves 0
// and 7 wait for done >= tz before they read / overwrite a slot.
// ORDER (round 4): a workgroup does not take its tile from blockIdx -- HIP does not promise that workgroups
// start in id order -- but from a TICKET: thread 0 draws `atomicAdd(ticket, 1)` when the workgroup starts,
// and the ticket goes through the slab-order map (xcd_slab_sync; one ticket sequence per XCD, launch_position).
// The producer of a tile's carry is the same slab position one slab earlier: q tickets back in the same
// sequence.  A smaller ticket of a sequence has been drawn earlier, so its
// workgroup is resident or finished, and it in turn only waits for a smaller ticket still: by induction
// every wait ends, whatever order the hardware dispatches in (the look-back argument of rocPRIM's scans).
// The poll is bounded all the same (CARRY_POLL_LIMIT, seconds): on expiry the wave raises bit 0 of the
// context's status word (host-visible), goes on WITHOUT a valid carried layer -- that launch's tables are
// void -- and the host reports EXABM4D_ERR_HIP at its next synchronisation and switches the carry off for
// the context (exabm4d_api.hip: check_async_status); never a hang.  All carry traffic is system-scope (stores written through,
// loads and the LDS-DMA prefetch with sc0 sc1): no assumption about which XCD's L2 a tile runs on.  Wave 0
// fetches a (dz, pass)'s two rounds by LDS-DMA one plane ahead of their use: no registers, no wait in
// front of the exchange barriers.  The same sums enter the same adds: tables are unchanged.
constexpr int CARRY_ROUND = (NE / 2) * SWIN * 64;                 // elements of one exchange round (33 x 64)
constexpr int CARRY_TILE = (NSTEP / 4) * 2 * CARRY_ROUND;         // one tile's top layer: 22 (dz, pass) x 2 rounds
struct Carry {
    int on;                  // 1: tiles advance by TCZ cell layers and carry their top layer
    int strip;               // tile order inside a slab: strips of this many tile rows, column-major (0 = raster)
    uint32_t* buf;           // [2][columns][CARRY_TILE]
    int* done;               // [columns]: tiles of the column that have finished (zeroed per launch)
    int* ticket;             // the launch's eight ticket counters (zeroed per launch), see ORDER above and launch_position
    unsigned* status;        // host-visible status word of the context: bit 0 = a carry wait ran out
    int fault;               // debug option "bm_carry_fault": every wait counts as run out (tests of the error path)
};
constexpr int CARRY_POLL_LIMIT = 1 << 23;       // x s_sleep(32) = 2048 cycles: ~7 s at 2.4 GHz
// Both rounds of one (dz, pass) into `dst` (2 * CARRY_ROUND elements, linear): 16 full 1 KB transfers
// and one of 512 bytes.
template <class T>
__device__ __forceinline__ void carry_prefetch(const T* src, T* dst, int lane) {
    static_assert(2 * CARRY_ROUND == 16 * 256 + 128, "two rounds = 16.5 KB");
#pragma unroll
    for (int i = 0; i < 16; i++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 256 * i + 4 * lane),
                                         (__attribute__((address_space(3))) void*)(dst + 256 * i), 16, 0, 17);
    if (lane < 32)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4096 + 4 * lane),
                                         (__attribute__((address_space(3))) void*)(dst + 4096), 16, 0, 17);
}
__device__ __forceinline__ void carry_wait_for(const int* done, int tiles, int lane, const Carry& carry) {
    if (lane == 0) {
        int polls = carry.fault ? CARRY_POLL_LIMIT : 0;
        while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < tiles) {
            if (++polls > CARRY_POLL_LIMIT) {
                __hip_atomic_fetch_or(carry.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(32);
        }
        // (the forced fault must not depend on a race with the producer)
        if (carry.fault) __hip_atomic_fetch_or(carry.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// The workgroup's position in the launch order: its ticket when tiles carry (see ORDER), else its id.
// EIGHT ticket counters, one per XCD sequence of the slab order (position = 8 j + c: XCD c's j-th tile): a
// workgroup draws from the counter of the XCD it runs on (HW_REG_XCC_ID), so that the tiles an XCD works on
// stay neighbours in ITS L2 -- one global counter handed consecutive tickets to whichever XCD asked first
// and cost block matching its cache sharing (PMC: 43 -> 151 GB and 152 -> 280 GB per 1024^3 launch,
// 116 -> 120-128 ms for the fp32 kernel).  A launch has exactly `quota` positions per sequence; should the
// hardware give an XCD more workgroups than that, the surplus ones draw from the other sequences in turn
// (8 * quota workgroups, 8 * quota positions: everybody finds one, every position is taken).  The order
// argument holds per counter: a tile's producer is ticket j - q of the SAME counter, drawn earlier.
__device__ __forceinline__ int launch_position(const Carry& carry, int quota) {
    if (!carry.on) return (int)blockIdx.x;
    __shared__ int s_pos;
    if (threadIdx.x == 0) {
        const int xcc = (int)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7);     // HW_REG_XCC_ID
        int pos = -1;
        for (int k = 0; k < 8 && pos < 0; k++) {
            const int c = (xcc + k) & 7;
            const int j = __hip_atomic_fetch_add(carry.ticket + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (j < quota) pos = 8 * j + c;
        }
        s_pos = pos;      // (-1 cannot happen: as many workgroups as positions; treated as padding)
    }
    __syncthreads();
    return s_pos;
}

// Each WAVE (= one z-layer of 8x8 cells) streams the candidate planes it needs through its own
// pair of LDS buffers: for a fixed dz and a pass of NE dy values, plane z+dz of the volume
// (37 rows x 44 columns around the tile) is staged once and serves all 4 x NE x 11 (row, dy, dx)
// combinations of every cell of the wave.  Interior tiles stage with LDS-DMA
// (global_load_lds_dwordx4: no registers, one step ahead of the arithmetic); tiles at an x edge
// of the volume stage synchronously with per-element clamping.  The search window is therefore
// read from L2/HBM 2 x 11 times per plane instead of 121 x 4 times per row (measured before:
// 1.8 TB of fabric traffic per 1024^3 launch at 7 TB/s -- bandwidth-bound on re-reads).
// The cell-sum exchange buffer aliases the waves' current plane buffers (each wave publishes the
// sums of its own 64 cells in its own buffer once its reads of the plane are done).
template <class TS>
__global__ __launch_bounds__(512) void bm_tile_kernel(const float* __restrict__ vol_all, VolGeom g,
                                                      uint32_t keymax,
                                                      uint32_t* __restrict__ keys_all, int tiles_y,
                                                      int tiles_x, int guarded, int xcd_q, int nbatch, Carry carry) {
    constexpr int TCX = TS::TCX, TRX = TS::TRX, TRY = TS::TRY, PROWS = TS::PROWS,
                  PCOLS = TS::PCOLS, PSTR = TS::PSTR, PCH = TS::PCH, NDMA = TS::NDMA, PBUF = TS::PBUF;
    __shared__ __align__(16) float pbuf_all[TCZ][2][PBUF];
    __shared__ __align__(16) float lower0[2 * CARRY_ROUND + 64];   // wave 0's lower cell layer (from the carry), both rounds

    // Tile order.  Slab order (xcd_q != 0): one slab = the tiles of ONE tz of all batch elements, all XCDs
    // inside it (xcd_slab_sync); a column is (batch element, ty, tx).  Otherwise every XCD walks its own
    // contiguous range of a batch element's tiles (blockIdx.y = batch element).
    const int per = tiles_y * tiles_x, cols = per * nbatch;
    int tz, col;
    if (xcd_q) {
        const int lp = launch_position(carry, (int)(gridDim.x >> 3));
        const int t = lp < 0 ? -1 : xcd_slab_sync(lp, cols, xcd_q);
        if (t < 0) return;                         // padding of the slab order
        tz = t / cols;
        col = t - tz * cols;
    } else {
        const int t = xcd_contiguous(blockIdx.x, gridDim.x);
        tz = t / per;
        col = (int)blockIdx.y * per + (t - tz * per);
    }
    const int bi = col / per, pos = col - bi * per;
    int ty, tx;
    if (carry.strip) {      // strips of `strip` tile rows, column-major inside a strip: concurrent tiles form a compact patch
        const int sidx = pos / (carry.strip * tiles_x), r = pos - sidx * (carry.strip * tiles_x);
        const int h = min(carry.strip, tiles_y - sidx * carry.strip);
        tx = r / h;
        ty = sidx * carry.strip + (r - tx * h);
    } else {
        ty = pos / tiles_x;
        tx = pos - ty * tiles_x;
    }
    const float* __restrict__ vol = vol_all + (size_t)bi * (size_t)g.nvox;
    uint32_t* __restrict__ keys = keys_all + (size_t)bi * (size_t)g.nref * MAXG;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int cx = lane % TCX, cy = lane / TCX, cz = __builtin_amdgcn_readfirstlane(tid >> 6);   // cz == wave index
    const int ix = TRX * tx + cx, iy = TRY * ty + cy;                 // cell == ref index in y, x

    const size_t sy = (size_t)g.nx, sz = (size_t)g.nx * (size_t)g.ny;

    // cell layers a tile advances by (CARRY above); this tile's column and its two carry slots
    const int Ls = (carry.on ? TCZ : TCZ - 1) * tz;
    const float* carry_rd = reinterpret_cast<const float*>(carry.buf) + ((size_t)((tz + 1) & 1) * cols + col) * CARRY_TILE;
    float* carry_wr = reinterpret_cast<float*>(carry.buf) + ((size_t)(tz & 1) * cols + col) * CARRY_TILE;
    int* done_col = carry.done + col;

    // Cell origin; cells beyond the volume are clamped inside it (their sums are never used).
    const int qy = min(STEP * iy, g.ny - STEP), qx = min(STEP * ix, g.nx - STEP);
    // Tile origin (voxels) of the staged window; wave-uniform.
    const int Y0 = STEP * TRY * ty, X0 = STEP * TRX * tx - RAD;
    // Whole staged column range inside the volume?  (wave-uniform; edge tiles clamp per element.)
    // `guarded`: the volume is one of the library's own buffers, with >= 256 bytes of mapped memory
    // on either side.  Columns outside the volume then need no clamping at all: they only ever
    // enter the distances of candidates that lie partly outside the volume, which `valid` masks
    // (a block inside the volume has all its columns inside), so x-edge tiles can take the
    // LDS-DMA path too and read whatever lies beyond the row ends.
    const bool xin = guarded || ((X0 >= 0) && (X0 + PCOLS - 1 <= g.nx - 1));

    const bool ref_yx = cx < TRX && cy < TRY && iy < g.ay && ix < g.ax;
    const int ry = STEP * iy, rx = STEP * ix;

    // bit d set iff candidate displacement dx = d - 5 keeps the block inside the volume
    uint32_t xmask = 0;
#pragma unroll
    for (int d = 0; d < SWIN; d++)
        xmask |= ((rx + d - RAD >= 0) && (rx + d - RAD <= g.nx - BLK)) ? (1u << d) : 0u;

    const bool yin = (Y0 - RAD >= 0) && (Y0 + 1 + PROWS - 1 <= g.ny - 1);
    do {   // once; `break` = an idle wave skips the tile's work, not the barriers and the tail below
    const int L = Ls + cz;                             // this wave's cell layer
    const bool active = L <= g.az;
    const int iz = L - 1;                              // the reference layer it owns (lower cells: wave cz - 1)
    const int qz = min(STEP * L, g.nz - STEP);
    const int Z0 = STEP * L;
    const bool lower_carried = carry.on && cz == 0 && tz > 0;          // the lower layer is the tile below's top layer
    const bool carries = carry.on && cz == TCZ - 1 && L + 1 <= g.az;   // ... and this wave's sums are the next tile's
    const bool ref_ok = ref_yx && active && (cz > 0 || lower_carried);
    if ((lower_carried || carries) && tz > 0) carry_wait_for(done_col, tz, lane, carry);
    const int rz = STEP * iz;

    uint32_t list[MAXG];
#pragma unroll
    for (int k = 0; k < MAXG; k++) list[k] = KEY_EMPTY;
    uint32_t thr = keymax;       // min(list[15], keymax): a key below it enters the list

    auto step_plane = [&](int step, int& dylo) -> const float* {
        const int z = step & 3, pass = (step >> 2) & 1, dz = (step >> 3) - RAD;
        dylo = pass == 0 ? -RAD : 1;
        const int pz = min(max(Z0 + z + dz, 0), g.nz - 1);
        return vol + (size_t)pz * sz;
    };
    // LDS-DMA of one plane into buffer `dst`: instruction i writes floats [256 i, 256 i + 256)
    // linearly, lane l the 16 bytes at chunk p = 64 i + l = (row p / 14, column chunk p % 14);
    // chunks beyond column 10 or row 36 are padding and re-read a valid address.
    // Tiles whose staged rows all lie inside the volume in y (all but the first and last tile row):
    // the row of a chunk needs no clamp, so its source is a wave-uniform base (plane, first row,
    // first column) plus a lane constant r * sy + 4 q -- eight instructions per DMA instead of the
    // thirty of the clamped form (a fifth of a step's instructions went into these addresses).
    auto issue_dma = [&](int step, float* dst) {
        int dylo;
        const float* plane = step_plane(step, dylo);
        int l = lane;
        asm volatile("" : "+v"(l));          // keeps the 9 (row, column) pairs out of registers
        if (yin) {
            const float* base = plane + (ptrdiff_t)(Y0 + dylo) * (ptrdiff_t)sy + X0;
#pragma unroll
            for (int i = 0; i < NDMA; i++) {
                const unsigned p = 64u * i + (unsigned)l;
                const unsigned r0 = (p * (65536u / PCH + 1u)) >> 16;          // p / PCH for p < 2^12
                const unsigned r = min(r0, (unsigned)(PROWS - 1));
                const unsigned q = min(p - r0 * PCH, (unsigned)(PCOLS / 4 - 1));
                const float* src = base + (r * (unsigned)sy + 4u * q);
                if (64 * i + 64 <= PROWS * PCH || p < (unsigned)(PROWS * PCH))       // chunks past the last row stay unwritten
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)src,
                        (__attribute__((address_space(3))) void*)(dst + 256 * i), 16, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NDMA; i++) {
            const int p = 64 * i + l;
            const int r = min(p / PCH, PROWS - 1), q = min(p % PCH, PCOLS / 4 - 1);
            const int yy = min(max(Y0 + dylo + r, 0), g.ny - 1);
            const float* src = plane + (size_t)yy * sy + (X0 + 4 * q);
            if (64 * i + 64 <= PROWS * PCH || p < PROWS * PCH)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)src,
                    (__attribute__((address_space(3))) void*)(dst + 256 * i), 16, 0, 0);
        }
    };
    auto stage_edge = [&](int step, float* dst) {               // edge tiles: clamp per element
        int dylo;
        const float* plane = step_plane(step, dylo);
        for (int c = lane; c < PROWS * PCOLS; c += 64) {
            const int r = c / PCOLS, q = c - r * PCOLS;
            const int yy = min(max(Y0 + dylo + r, 0), g.ny - 1);
            const int xx = min(max(X0 + q, 0), g.nx - 1);
            dst[r * PSTR + q] = plane[(size_t)yy * sy + xx];
        }
    };

    // A wave whose cell layer lies beyond the last one any reference block uses (layers 0 .. az)
    // has nothing to contribute -- the last z tile of a 64^3 patch needs 2 of its 8 layers -- and
    // only keeps the workgroup's barrier count: four per (dz, pass).
    if (!active) {
        for (int i = 0; i < (NSTEP / 4) * 4; i++) __syncthreads();
        break;
    }

    if (xin)
        issue_dma(0, pbuf_all[cz][0]);
    else
        stage_edge(0, pbuf_all[cz][0]);

    float acc[NE][SWIN];
#pragma unroll
    for (int e = 0; e < NE; e++)
#pragma unroll
        for (int d = 0; d < SWIN; d++) acc[e][d] = 0.0f;

#pragma unroll 1
    for (int step = 0; step < NSTEP; step++) {
        const int z = step & 3, pass = (step >> 2) & 1, dz = (step >> 3) - RAD;
        const int dylo = pass == 0 ? -RAD : 1;
        float* cur = pbuf_all[cz][step & 1];
        float* nxt = pbuf_all[cz][(step + 1) & 1];

        // this step's plane has landed (DMA counts in vmcnt) ...
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // own cell plane z (16 values; L1-resident after the first pass)
        float A[16];
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const float4u t4 = *reinterpret_cast<const float4u*>(
                vol + (size_t)(qz + z) * sz + (size_t)(qy + y) * sy + qx);
            A[4 * y] = t4.x;
            A[4 * y + 1] = t4.y;
            A[4 * y + 2] = t4.z;
            A[4 * y + 3] = t4.w;
        }
        // ... and the next plane starts flying while this one is consumed.  The A loads are
        // drained first: vmcnt retires in order and hipcc waits vmcnt(0) for A after the
        // (conditional) DMA block, which would expose the whole DMA latency every step.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (xin && step + 1 < NSTEP) issue_dma(step + 1, nxt);
        if (lower_carried && z == 2) carry_prefetch(carry_rd + (size_t)(step >> 2) * 2 * CARRY_ROUND, lower0, lane);

        // ---- accumulate: row rp = y + e of the cell's window -----------------------------------
        {
            const float* wrow = cur + (4 * cy) * PSTR + 4 * cx;
            float4 wq[4], wn[4];
#pragma unroll
            for (int j = 0; j < 4; j++) wq[j] = *reinterpret_cast<const float4*>(wrow + 4 * j);
#pragma unroll
            for (int rp = 0; rp < 3 + NE; rp++) {
                // one-row lookahead; the fence keeps hipcc from hoisting all nine rows' loads
                // (and their 144 registers) to the top of the plane
                if (rp + 1 < 3 + NE) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        wn[j] = *reinterpret_cast<const float4*>(wrow + (rp + 1) * PSTR + 4 * j);
                }
                asm volatile("" ::: "memory");
                const float w[16] = {wq[0].x, wq[0].y, wq[0].z, wq[0].w, wq[1].x, wq[1].y,
                                     wq[1].z, wq[1].w, wq[2].x, wq[2].y, wq[2].z, wq[2].w,
                                     wq[3].x, wq[3].y, wq[3].z, wq[3].w};
#pragma unroll
                for (int y = 0; y < 4; y++) {
                    const int e = rp - y;
                    if (e >= 0 && e < NE) {
#pragma unroll
                        for (int x = 0; x < 4; x++) {
                            const float a = A[4 * y + x];
#pragma unroll
                            for (int d = 0; d < SWIN; d++) {
                                const float t = a - w[d + x];
                                acc[e][d] = fmaf(t, t, acc[e][d]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) wq[j] = wn[j];
            }
        }
        asm volatile("" ::: "memory");

        if (z == 3) {
            // ---- cell sums -> LDS -> reference lanes: combine 8 cells, build keys, top-16 ------
            // exchange slot of cell (wave w, local l), displacement d: pbuf_all[w][cur][64 d + l]
            const bool vz = (rz + dz >= 0) && (rz + dz <= g.nz - BLK);
            float* mine = cur + lane;
            const float* up_w = cur + lane;                                  // own cells: the upper layer
            // the lower layer: wave cz - 1's sums; for wave 0 the previous block's top layer
            const float* lo_w = (cz > 0 ? pbuf_all[max(cz - 1, 0)][step & 1] : lower0) + lane;
            float* cslot = carry_wr + (size_t)((step >> 2) * 2) * CARRY_ROUND + lane;
            // NE / 2 dy values per round (33 sums per cell fit the 36 slots of a plane buffer):
            // two barrier pairs per pass instead of six.  Each cell lane first adds its
            // x-neighbour's sum (DPP row_shl:1, no LDS), so a reference lane reads 4 values per
            // candidate instead of 8: S = ((c000+c001)+(c010+c011)) + ((c100+c101)+(c110+c111)).
#pragma unroll
            for (int e0 = 0; e0 < NE; e0 += NE / 2) {
#pragma unroll
                for (int e = e0; e < e0 + NE / 2; e++)
#pragma unroll
                    for (int d = 0; d < SWIN; d++) {
                        const float right = __int_as_float(__builtin_amdgcn_update_dpp(
                            0, __float_as_int(acc[e][d]), 0x101 /* row_shl:1 */, 0xF, 0xF, true));
                        mine[64 * ((e - e0) * SWIN + d)] = acc[e][d] + right;
                    }
                __syncthreads();
                if (ref_ok) {
#pragma unroll
                    for (int e = e0; e < e0 + NE / 2; e++) {
                        const int dy = dylo + e;
                        if (dy <= RAD) {                   // dy = 6 of the second pass is a dummy
                            const bool vzy = vz && (ry + dy >= 0) && (ry + dy <= g.ny - BLK);
                            const uint32_t cbase =
                                1u + (uint32_t)(((dz + RAD) * SWIN + (dy + RAD)) * SWIN);
                            const bool self_row = (dz == 0) && (dy == 0);
#pragma unroll
                            for (int d = 0; d < SWIN; d++) {
                                const float* c0 = lo_w + (cz == 0 && e0 ? CARRY_ROUND : 0) + 64 * ((e - e0) * SWIN + d);
                                const float* c1 = up_w + 64 * ((e - e0) * SWIN + d);
                                // S = (c0[0] + c0[TCX]) + (c1[0] + c1[TCX]) as three plain adds: left
                                // to the SLP vectoriser this becomes two packed adds, three moves and
                                // two wait states
                                float s0 = c0[0] + c0[TCX];
                                asm("" : "+v"(s0));        // (an opaque use keeps the two sums apart)
                                const float s1 = c1[0] + c1[TCX];
                                const float S = s0 + s1;
                                const uint32_t code = (d == RAD && self_row) ? 0u : cbase + d;
                                const bool valid = vzy && ((xmask >> d) & 1u);
                                uint32_t key = (__float_as_uint(S) & KEY_DMASK) | code;
                                key = valid ? key : KEY_EMPTY;
                                // thr = min(16th best so far, admission bound): one compare decides
                                if (__any(key < thr)) {
                                    const uint32_t kins = key < thr ? key : KEY_EMPTY;
                                    list_insert_inplace(list, kins);
                                    thr = min(list[MAXG - 1], keymax);
                                }
                            }
                        }
                    }
                }
                __syncthreads();
                if (carries) {
                    // the top wave's sums are the next block's lowest layer (read there by wave 0 in
                    // this same round, i.e. before this slot is written again)
#pragma unroll
                    for (int e = e0; e < e0 + NE / 2; e++)
#pragma unroll
                        for (int d = 0; d < SWIN; d++) {
                            const float right = __int_as_float(__builtin_amdgcn_update_dpp(
                                0, __float_as_int(acc[e][d]), 0x101 /* row_shl:1 */, 0xF, 0xF, true));
                            __hip_atomic_store(cslot + (e0 ? CARRY_ROUND : 0) + 64 * ((e - e0) * SWIN + d), acc[e][d] + right,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                }
            }
#pragma unroll
            for (int e = 0; e < NE; e++)
#pragma unroll
                for (int d = 0; d < SWIN; d++) acc[e][d] = 0.0f;
        }

        // edge tiles stage the next plane synchronously (all reads of `nxt` are long done)
        if (!xin && step + 1 < NSTEP) stage_edge(step + 1, nxt);
    }

    if (ref_ok) {
        uint32_t* out = keys + ((size_t)((size_t)iz * g.gy + iy) * g.gx + ix) * MAXG;
        // never empty: a block with an infinity or a NaN in it has no admissible distance, not
        // even to itself; it forms the one-block group of key 0, so that every key downstream names a block
        list[0] = list[0] == KEY_EMPTY ? 0u : list[0];
#pragma unroll
        for (int k = 0; k < MAXG; k += 4) {
            uint4 v = make_uint4(list[k], list[k + 1], list[k + 2], list[k + 3]);
            *reinterpret_cast<uint4*>(out + k) = v;
        }
    }
    } while (false);

    if (carry.on) {
        // the tile is finished when wave 7's stores have been acknowledged and wave 0's last prefetch has landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(done_col, tz + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------------
// Integer tile kernel for uint16 volumes (stage-1 matching of the uint16 pipelines).
//
// The noisy volume of the uint16 entry points is `(float)v - offset`: differences of two voxels are
// exact integers, so every partial sum of the specification's fmaf chains is an integer, and as
// long as a block distance stays below 2^24 it is EXACT in fp32 -- the chain order no longer
// matters and the same number can be formed in integer arithmetic.  A candidate is admitted only
// below keymax <= 2^24 (checked by the launcher: c_match * sigma^2 * 512 < 2^24), so for every
// candidate that can enter a match table the integer sum converted to float has exactly the bits
// the float kernel produces, and everything else is rejected by both.  v_pk_sub_i16 (saturating)
// forms two differences, v_dot2_i32_i16 (saturating) squares and accumulates both: the same two
// instructions per two (voxel, displacement) pairs as the packed-fp32 form, and -- measured,
// tools/dbg/valu_rate_bench.hip -- at the same ~4.3 cycles each; what the integer form saves is
// LDS: a staged plane is half the bytes and a window row two ds_read_b64 instead of four b128.
//
// Input: the volume with every voxel XOR 0x8000 (= v - 32768 as int16), in the library's scratch
// with mapped memory around it (written next to the fp32 counts by the uint16 pipelines).  The
// bias makes the saturating int16 subtraction exact for |a - w| < 32768; a saturated difference
// squares to > 2^29 and can never be admitted.  nx must be even (4-byte aligned LDS-DMA rows).
// ------------------------------------------------------------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));
template <int TCY_, int TCX_>
struct TileShape16 {
    static constexpr int TCY = TCY_, TCX = TCX_;
    static constexpr int TRY = TCY - 1, TRX = TCX - 1;
    static constexpr int PROWS = 4 * (TCY - 1) + 3 + NE;          // staged rows (37 / 21)
    static constexpr int PCOLS = ((4 * TCX + 10 + 7) / 8) * 8;    // staged uint16 columns (48 / 80)
    // row stride in uint16: two cell rows of a 32-lane ds_read_b64 group must fall into different
    // 16-bank windows (8 x 8: 56 = 24 mod 32), one 16-lane cell row is 32 banks (4 x 16: 80 = 16 mod 32)
    static constexpr int PSTR = TCX == 8 ? 56 : 80;
    static constexpr int PCH = PSTR / 8;                          // 16-byte chunks per staged row
    static constexpr int NDMA = (PROWS * PCH + 63) / 64;
    static constexpr int PLANE_DW = NDMA * 256;                   // dwords of one plane buffer
    static constexpr int XCH_DW = (NE / 2) * SWIN * 64;           // cell-sum exchange: 33 sums x 64 cells
    static constexpr int PBUF = PLANE_DW > XCH_DW ? PLANE_DW : XCH_DW;
    static_assert(TCY * TCX == 64 && PCOLS <= PSTR, "one wave per cell layer");
};

template <class TS>
__global__ __launch_bounds__(512) void bm_tile16_kernel(const uint16_t* __restrict__ vol_all, VolGeom g,
                                                        uint32_t keymax,
                                                        uint32_t* __restrict__ keys_all, int tiles_y,
                                                        int tiles_x, int xcd_q, int nbatch, Carry carry) {
    constexpr int TCX = TS::TCX, TRX = TS::TRX, TRY = TS::TRY, PROWS = TS::PROWS, PCOLS = TS::PCOLS,
                  PSTR = TS::PSTR, PCH = TS::PCH, NDMA = TS::NDMA, PBUF = TS::PBUF;
    __shared__ __align__(16) uint32_t pbuf_all[TCZ][2][PBUF];
    __shared__ __align__(16) uint32_t lower0[2 * CARRY_ROUND + 64];   // wave 0's lower cell layer (from the carry), both rounds

    // tile order: see bm_tile_kernel
    const int per = tiles_y * tiles_x, cols = per * nbatch;
    int tz, col;
    if (xcd_q) {
        const int lp = launch_position(carry, (int)(gridDim.x >> 3));
        const int t = lp < 0 ? -1 : xcd_slab_sync(lp, cols, xcd_q);
        if (t < 0) return;                         // padding of the slab order
        tz = t / cols;
        col = t - tz * cols;
    } else {
        const int t = xcd_contiguous(blockIdx.x, gridDim.x);
        tz = t / per;
        col = (int)blockIdx.y * per + (t - tz * per);
    }
    const int bi = col / per, pos = col - bi * per;
    int ty, tx;
    if (carry.strip) {      // strips of `strip` tile rows, column-major inside a strip: concurrent tiles form a compact patch
        const int sidx = pos / (carry.strip * tiles_x), r = pos - sidx * (carry.strip * tiles_x);
        const int h = min(carry.strip, tiles_y - sidx * carry.strip);
        tx = r / h;
        ty = sidx * carry.strip + (r - tx * h);
    } else {
        ty = pos / tiles_x;
        tx = pos - ty * tiles_x;
    }
    const uint16_t* __restrict__ vol = vol_all + (size_t)bi * (size_t)g.nvox;
    uint32_t* __restrict__ keys = keys_all + (size_t)bi * (size_t)g.nref * MAXG;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int cx = lane % TCX, cy = lane / TCX, cz = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ix = TRX * tx + cx, iy = TRY * ty + cy;
    const size_t sy = (size_t)g.nx, sz = (size_t)g.nx * (size_t)g.ny;

    // cell layers a tile advances by (CARRY above); this tile's column and its two carry slots
    const int Ls = (carry.on ? TCZ : TCZ - 1) * tz;
    const uint32_t* carry_rd = carry.buf + ((size_t)((tz + 1) & 1) * cols + col) * CARRY_TILE;
    uint32_t* carry_wr = carry.buf + ((size_t)(tz & 1) * cols + col) * CARRY_TILE;
    int* done_col = carry.done + col;

    const int qy = min(STEP * iy, g.ny - STEP), qx = min(STEP * ix, g.nx - STEP);
    // staged window: rows Y0 + dylo ..., columns X0 ... with X0 even (one column more to the left
    // than the search needs): cell cx finds candidate column x + d of its voxel x at staged
    // column 4 cx + 1 + x + d
    const int Y0 = STEP * TRY * ty, X0 = STEP * TRX * tx - RAD - 1;
    const int ry = STEP * iy, rx = STEP * ix;
    const bool ref_yx = cx < TRX && cy < TRY && iy < g.ay && ix < g.ax;

    uint32_t xmask = 0;
#pragma unroll
    for (int d = 0; d < SWIN; d++)
        xmask |= ((rx + d - RAD >= 0) && (rx + d - RAD <= g.nx - BLK)) ? (1u << d) : 0u;
    const bool yin = (Y0 - RAD >= 0) && (Y0 + 1 + PROWS - 1 <= g.ny - 1);

    do {   // once; `break` = an idle wave skips the tile's work, not the barriers and the tail below
    const int L = Ls + cz;                             // this wave's cell layer
    const bool active = L <= g.az;                     // cell layers 0 .. az exist
    const int iz = L - 1;                              // ... and the reference layer it owns
    const int qz = min(STEP * L, g.nz - STEP);
    const int Z0 = STEP * L;
    const bool lower_carried = carry.on && cz == 0 && tz > 0;          // the lower layer is the tile below's top layer
    const bool carries = carry.on && cz == TCZ - 1 && L + 1 <= g.az;   // ... and this wave's sums are the next tile's
    const bool ref_ok = ref_yx && active && (cz > 0 || lower_carried);
    if ((lower_carried || carries) && tz > 0) carry_wait_for(done_col, tz, lane, carry);
    const int rz = STEP * iz;

    uint32_t list[MAXG];
#pragma unroll
    for (int k = 0; k < MAXG; k++) list[k] = KEY_EMPTY;
    uint32_t thr = keymax;

    auto issue_dma = [&](int step, uint32_t* dst) {
        const int z = step & 3, pass = (step >> 2) & 1, dz = (step >> 3) - RAD;
        const int dylo = pass == 0 ? -RAD : 1;
        const int pz = min(max(Z0 + z + dz, 0), g.nz - 1);
        const uint16_t* plane = vol + (size_t)pz * sz;
        int l = lane;
        asm volatile("" : "+v"(l));
#pragma unroll
        for (int i = 0; i < NDMA; i++) {
            const unsigned p = 64u * i + (unsigned)l;
            const unsigned r0 = (p * (65536u / PCH + 1u)) >> 16;          // p / PCH for p < 2^12
            const unsigned r = min(r0, (unsigned)(PROWS - 1));
            const unsigned q = min(p - r0 * PCH, (unsigned)(PCOLS / 8 - 1));
            const int yy = yin ? Y0 + dylo + (int)r : min(max(Y0 + dylo + (int)r, 0), g.ny - 1);
            const uint16_t* src = plane + (ptrdiff_t)yy * (ptrdiff_t)sy + (X0 + 8 * (int)q);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)src,
                (__attribute__((address_space(3))) void*)(dst + 256 * i), 16, 0, 0);
        }
    };

    if (!active) {
        // a wave whose cell layer lies beyond the last one any reference block uses only keeps the
        // workgroup's barrier count: four per (dz, pass)
        for (int i = 0; i < (NSTEP / 4) * 4; i++) __syncthreads();
        break;
    }
    issue_dma(0, pbuf_all[cz][0]);

    int acc[NE][SWIN];
#pragma unroll
    for (int e = 0; e < NE; e++)
#pragma unroll
        for (int d = 0; d < SWIN; d++) acc[e][d] = 0;

    // One step = one staged plane.  The first pass of a dz covers dy = -5..0 (six rows of sums), the
    // second dy = 1..5 (five): the body is instantiated for both counts, so no cycles go into a
    // twelfth, masked dy (1024^3: 117.7 -> 114.6 ms).  The float kernel keeps the single body with
    // a masked row: it already sits at 252 registers and the second instantiation spills (131 ->
    // 158 ms measured).
    auto step_body = [&](auto NEc, const int step) {
        constexpr int NEP = decltype(NEc)::value;                 // dy values of this pass
        constexpr int NR0 = (NEP + 1) / 2;                        // ... of its first exchange round
        const int z = step & 3, dz = (step >> 3) - RAD;
        const int dylo = NEP == NE ? -RAD : 1;
        uint32_t* cur = pbuf_all[cz][step & 1];
        uint32_t* nxt = pbuf_all[cz][(step + 1) & 1];

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // own cell plane z: four rows of four biased uint16 (two dwords each)
        uint2 A[4];
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const uint32_t* ap = reinterpret_cast<const uint32_t*>(
                vol + (size_t)(qz + z) * sz + (size_t)(qy + y) * sy + qx);
            A[y] = make_uint2(ap[0], ap[1]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (step + 1 < NSTEP) issue_dma(step + 1, nxt);
        if (lower_carried && z == 2) carry_prefetch(carry_rd + (size_t)(step >> 2) * 2 * CARRY_ROUND, lower0, lane);

        {
            const uint32_t* wrow = cur + ((4 * cy) * PSTR + 4 * cx) / 2;       // dword index
            uint2 wq[4], wn[4];
#pragma unroll
            for (int j = 0; j < 4; j++) wq[j] = *reinterpret_cast<const uint2*>(wrow + 2 * j);
#pragma unroll
            for (int rp = 0; rp < 3 + NEP; rp++) {
                if (rp + 1 < 3 + NEP) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        wn[j] = *reinterpret_cast<const uint2*>(wrow + (rp + 1) * (PSTR / 2) + 2 * j);
                }
                asm volatile("" ::: "memory");
                // W[i] = staged columns (2 i, 2 i + 1); Wo[i] = columns (2 i + 1, 2 i + 2)
                const uint32_t W[8] = {wq[0].x, wq[0].y, wq[1].x, wq[1].y, wq[2].x, wq[2].y, wq[3].x, wq[3].y};
                uint32_t Wo[7];
#pragma unroll
                for (int i = 0; i < 7; i++) Wo[i] = __builtin_amdgcn_alignbit(W[i + 1], W[i], 16);
#pragma unroll
                for (int y = 0; y < 4; y++) {
                    const int e = rp - y;
                    if (e >= 0 && e < NEP) {
#pragma unroll
                        for (int xp = 0; xp < 2; xp++) {
                            const s16x2 a = __builtin_bit_cast(s16x2, xp ? A[y].y : A[y].x);
#pragma unroll
                            for (int d = 0; d < SWIN; d++) {
                                const int j = 1 + 2 * xp + d;            // staged column of voxel x = 2 xp
                                const uint32_t wp = (j & 1) ? Wo[(j - 1) / 2] : W[j / 2];
                                const s16x2 t = __builtin_elementwise_sub_sat(a, __builtin_bit_cast(s16x2, wp));
                                acc[e][d] = __builtin_amdgcn_sdot2(t, t, acc[e][d], true);
                            }
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) wq[j] = wn[j];
            }
        }
        asm volatile("" ::: "memory");

        if (z == 3) {
            const bool vz = (rz + dz >= 0) && (rz + dz <= g.nz - BLK);
            uint32_t* mine = cur + lane;
            const uint32_t* up_w = cur + lane;                                      // own cells: the upper layer
            // the lower layer: wave cz - 1's sums; for wave 0 the previous block's top layer
            const uint32_t* lo_w = (cz > 0 ? pbuf_all[max(cz - 1, 0)][step & 1] : lower0) + lane;
            uint32_t* cslot = carry_wr + (size_t)((step >> 2) * 2) * CARRY_ROUND + lane;
            // sum of this cell and its x-neighbour; cell sums are capped at 2^27 so that eight of
            // them cannot wrap (a capped sum is far beyond any admissible distance)
            auto pair_sum = [&](int e, int d) {
                const uint32_t c = min((uint32_t)acc[e][d], 1u << 27);
                const uint32_t right = (uint32_t)__builtin_amdgcn_update_dpp(
                    0, (int)c, 0x101 /* row_shl:1 */, 0xF, 0xF, true);
                return c + right;
            };
#pragma unroll
            for (int e0 = 0; e0 < NEP; e0 += NR0) {
#pragma unroll
                for (int e = e0; e < (e0 + NR0 < NEP ? e0 + NR0 : NEP); e++)
#pragma unroll
                    for (int d = 0; d < SWIN; d++) mine[64 * ((e - e0) * SWIN + d)] = pair_sum(e, d);
                __syncthreads();
                if (ref_ok) {
                    // a dy row's block sums first (its 22 LDS reads in flight together), then its
                    // candidates one by one: read-add-compare-branch per candidate exposed an LDS round
                    // trip each, with every wave of the workgroup in this phase at the same time
                    // (114.7 -> 111.7 ms; the same change to the fp32 kernel measured no gain)
#pragma unroll
                    for (int e = e0; e < (e0 + NR0 < NEP ? e0 + NR0 : NEP); e++) {
                        uint32_t Sv[1][SWIN];          // one dy row of sums at a time (registers)
#pragma unroll
                        for (int d = 0; d < SWIN; d++) {
                            const uint32_t* c0 = lo_w + (cz == 0 && e0 ? CARRY_ROUND : 0) + 64 * ((e - e0) * SWIN + d);
                            const uint32_t* c1 = up_w + 64 * ((e - e0) * SWIN + d);
                            Sv[0][d] = (c0[0] + c0[TCX]) + (c1[0] + c1[TCX]);
                        }
                        asm volatile("" ::: "memory");
                        const int dy = dylo + e;
                        {
                            const bool vzy = vz && (ry + dy >= 0) && (ry + dy <= g.ny - BLK);
                            const uint32_t cbase =
                                1u + (uint32_t)(((dz + RAD) * SWIN + (dy + RAD)) * SWIN);
                            const bool self_row = (dz == 0) && (dy == 0);
#pragma unroll
                            for (int d = 0; d < SWIN; d++) {
                                const uint32_t S = Sv[0][d];
                                const uint32_t code = (d == RAD && self_row) ? 0u : cbase + d;
                                const bool valid = vzy && ((xmask >> d) & 1u);
                                uint32_t key = (__float_as_uint((float)S) & KEY_DMASK) | code;
                                key = valid ? key : KEY_EMPTY;
                                if (__any(key < thr)) {
                                    const uint32_t kins = key < thr ? key : KEY_EMPTY;
                                    list_insert_inplace(list, kins);
                                    thr = min(list[MAXG - 1], keymax);
                                }
                            }
                        }
                    }
                }
                __syncthreads();
                if (carries) {
                    // the top wave's sums are the next block's lowest layer (read there by wave 0 in
                    // this same round, i.e. before this slot is written again)
                    uint32_t* cw = cslot + (size_t)(e0 ? CARRY_ROUND : 0);
#pragma unroll
                    for (int e = e0; e < (e0 + NR0 < NEP ? e0 + NR0 : NEP); e++)
#pragma unroll
                        for (int d = 0; d < SWIN; d++)
                            __hip_atomic_store(cw + 64 * ((e - e0) * SWIN + d), pair_sum(e, d), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
#pragma unroll
            for (int e = 0; e < NE; e++)
#pragma unroll
                for (int d = 0; d < SWIN; d++) acc[e][d] = 0;
        }
    };
#pragma unroll 1
    for (int step = 0; step < NSTEP; step++) {
        if (((step >> 2) & 1) == 0)
            step_body(std::integral_constant<int, NE>{}, step);
        else
            step_body(std::integral_constant<int, NE - 1>{}, step);
    }

    if (ref_ok) {
        uint32_t* out = keys + ((size_t)((size_t)iz * g.gy + iy) * g.gx + ix) * MAXG;
        // never empty: a block with an infinity or a NaN in it has no admissible distance, not
        // even to itself; it forms the one-block group of key 0, so that every key downstream names a block
        list[0] = list[0] == KEY_EMPTY ? 0u : list[0];
#pragma unroll
        for (int k = 0; k < MAXG; k += 4) {
            uint4 v = make_uint4(list[k], list[k + 1], list[k + 2], list[k + 3]);
            *reinterpret_cast<uint4*>(out + k) = v;
        }
    }
    } while (false);

    if (carry.on) {
        // the tile is finished when wave 7's stores have been acknowledged and wave 0's last prefetch has landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(done_col, tz + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
        if (lane == 0) out[k] = (k == 0 && m == KEY_EMPTY) ? 0u : m;      // never empty
    }
}

// ------------------------------------------------------------------------------------------------
// host launchers (called from exabm4d_api.cpp)
// ------------------------------------------------------------------------------------------------
// Workgroup order: 0 = every XCD walks its own contiguous range of tiles (so the XCDs sit in different z
// slabs of a large volume); 1 = all XCDs inside one slab of ty x tx tiles at a time (worth it once a slab
// has a few tiles per CU), raster order inside the slab; n >= 2 = the same, the slab walked in strips of n
// tile rows, column-major inside a strip: the 32 tiles an XCD works on together then form a 2-D patch
// instead of one long row of tiles, a smaller set of cache lines (measured at 1024^3 on a noisy volume: fp32
// kernel 109.4 -> 107.5 ms with strips of 2, 108.7 with 3, 110.0 with 4; the integer kernel and the fp32 kernel
// on the pipeline's smooth basic estimate are indifferent -- DESIGN.md 5.1d).
// (BmOpts::xcd_mode, default 2; BmOpts::carry: 0 = off (tiles advance by seven cell layers), 1 = on wherever it
// saves a tile per column, 2 = on whenever a column has two tiles (tests).  Per context since round 4.)

// Tile shape: fewer (y, x) tiles = fewer idle cell lanes (64^3 patches: 5 flat tiles against 9 cubes)
template <class Cube, class Flat>
static bool flat_tiles(const VolGeom& g) {
    auto tiles = [&](int try_, int trx) {
        return (long long)((g.ay + try_ - 1) / try_) * ((g.ax + trx - 1) / trx);
    };
    return tiles(Flat::TRY, Flat::TRX) < tiles(Cube::TRY, Cube::TRX);
}
// The launch block matching chooses for a geometry (host logic only; the float and the integer kernel share
// tile shapes).  Evaluated ONCE per launch by the API layer, which also provides plan.carry_bytes of device
// memory when plan.carry is set.
BmPlan bm_plan(const VolGeom& g, int batch, const BmOpts& opt) {
    using Cube = TileShape<8, 8>;
    using Flat = TileShape<4, 16>;
    BmPlan p = {};
    if (g.az <= 0 || g.ay <= 0 || g.ax <= 0) return p;
    p.flat = flat_tiles<Cube, Flat>(g) ? 1 : 0;
    const int try_ = p.flat ? Flat::TRY : Cube::TRY, trx = p.flat ? Flat::TRX : Cube::TRX;
    p.ty = (g.ay + try_ - 1) / try_;
    p.tx = (g.ax + trx - 1) / trx;
    const long long cols = (long long)p.ty * p.tx * batch;          // columns of tiles: (batch element, ty, tx)
    const int tz7 = (g.az + TCZ - 2) / (TCZ - 1), tz8 = g.az / TCZ + 1;
    const bool slab = opt.xcd_mode != 0 && cols >= 512;
    // The carry pays where it saves a tile per column (a 64^3 patch: 15 reference layers = 3 tiles without,
    // 2 with) AND a slab is a couple of rounds of the 256 CUs: a tile waits for the tile below it, and when
    // both are resident together the upper one only spins on a CU (a single small volume gains nothing).
    const bool worth = tz8 >= 2 && tz8 < tz7 && slab;
    // 744 KB per column (+ done[columns] + the ticket counter): beyond 16 GB the launch goes without
    const size_t need = 2 * (size_t)cols * CARRY_TILE * sizeof(uint32_t) + ((size_t)cols + 64) * sizeof(int);
    p.carry = (opt.carry != 0 && (worth || (opt.carry == 2 && tz8 >= 2)) && need <= ((size_t)16 << 30)) ? 1 : 0;
    p.carry_bytes = p.carry ? need : 0;
    p.tz = p.carry ? tz8 : tz7;
    p.xq = (p.carry || (slab && p.tz >= 2)) ? (int)((cols + 7) / 8) : 0;
    p.strip = (opt.xcd_mode >= 2 && p.xq) ? opt.xcd_mode : 0;
    p.fault = opt.carry_fault;
    return p;
}

hipError_t launch_blockmatch(const float* vol, const VolGeom& g, int batch, uint32_t keymax,
                             uint32_t* keys, hipStream_t stream, int force_generic, int guarded,
                             const uint16_t* vol16, const BmPlan& p, void* carry_mem, unsigned* status) {
    // vol16 != nullptr: the volume's uint16 counts XOR 0x8000 in guarded scratch; the caller has
    // checked that the integer kernel gives the float kernel's tables (keymax <= 2^24, nx even)
    // carry_mem: p.carry_bytes of device memory when p.carry; status: the context's host-visible status word
    if (p.carry && (!carry_mem || !status)) return hipErrorInvalidValue;
    if (!force_generic && g.az > 0 && g.ay > 0 && g.ax > 0) {
        using Cube16 = TileShape16<8, 8>;
        using Flat16 = TileShape16<4, 16>;
        using Cube = TileShape<8, 8>;
        using Flat = TileShape<4, 16>;
        static_assert(Cube::TRY == Cube16::TRY && Cube::TRX == Cube16::TRX && Flat::TRY == Flat16::TRY &&
                      Flat::TRX == Flat16::TRX, "one tile plan for both kernels");
        const bool flat = p.flat != 0;
        Carry carry;
        carry.strip = p.strip;
        carry.on = p.carry;
        carry.buf = static_cast<uint32_t*>(carry_mem);
        const size_t cols = (size_t)p.ty * p.tx * (size_t)batch;
        carry.done = p.carry ? reinterpret_cast<int*>(carry.buf + 2 * cols * CARRY_TILE) : nullptr;
        carry.ticket = p.carry ? carry.done + cols : nullptr;
        carry.status = status;
        carry.fault = p.fault;
        if (p.carry) {      // done[columns] and the eight ticket counters behind it
            hipError_t e = hipMemsetAsync(carry.done, 0, (cols + 8) * sizeof(int), stream);
            if (e != hipSuccess) return e;
        }
        dim3 grid((unsigned)(p.xq ? 8 * p.xq * p.tz : p.tz * p.ty * p.tx), (unsigned)(p.xq ? 1 : batch));
        if (vol16) {
            if (flat)
                hipLaunchKernelGGL(bm_tile16_kernel<Flat16>, grid, dim3(512), 0, stream, vol16, g, keymax,
                                   keys, p.ty, p.tx, p.xq, batch, carry);
            else
                hipLaunchKernelGGL(bm_tile16_kernel<Cube16>, grid, dim3(512), 0, stream, vol16, g, keymax,
                                   keys, p.ty, p.tx, p.xq, batch, carry);
        } else {
            if (flat)
                hipLaunchKernelGGL(bm_tile_kernel<Flat>, grid, dim3(512), 0, stream, vol, g, keymax, keys,
                                   p.ty, p.tx, guarded, p.xq, batch, carry);
            else
                hipLaunchKernelGGL(bm_tile_kernel<Cube>, grid, dim3(512), 0, stream, vol, g, keymax, keys,
                                   p.ty, p.tx, guarded, p.xq, batch, carry);
        }
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
