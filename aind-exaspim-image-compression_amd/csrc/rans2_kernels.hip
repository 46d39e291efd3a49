// rans2_kernels.hip -- EXAC v2 chunk coder (oracle/exac_codec.c states the format and
// the kernels' bytes are bit-identical to it): every element is predicted from the voxel above and
// the voxel in the plane before, the zigzag residual is split into a 64-symbol alphabet (32 direct
// values + one symbol per octave with the mantissa as raw bits), and the symbol is coded with one of
// 16 static tables chosen by the neighbours' residual magnitudes -- all of it through the 64
// interleaved rANS states of v1, lane = state, row of 64 elements = one coalesced wave access.
// Replaces the arithmetic behind `len(codec.encode(chunk))` of the reference's compute_cratio
// (utils/img_util.py:401-441; its codec is third-party Blosc-zstd, evaluate.py:40).
//
// Integer work only.  One workgroup = one chunk = one wave: pass 1 counts (context, symbol) pairs in
// LDS, the wave turns the 16 histograms into tables (lane = symbol), pass 2 walks the rows from the
// last to the first and appends renormalisation words with a ballot + mbcnt rank.
#include <algorithm>

#include "exabm4d_kernels.h"
#include "rans_common.h"

namespace exabm4d {

namespace {

constexpr int NCTX = 16, NSYM = 64;
constexpr uint32_t TAP_LIMIT = 8000u;     // largest tap distance (elements) the format uses
constexpr int HDR2 = 276;                 // magic, n, ey, ex, nwords, present[16], wide[16]
constexpr int TAB2_MAX = NCTX * NSYM * 2;
constexpr int SLOT2_TABLEN = HDR2 + TAB2_MAX;          // u32: table bytes of the chunk (scratch only)
constexpr uint32_t RING_MAX = 8192u;      // decoder history (elements) at most: >= TAP_LIMIT + 2 * 64

// activity -> context: number of edges {1,2,3,4,5,6,8,10,13,17,22,30,45,70,120} that are <= a
__device__ __forceinline__ uint32_t ctx_of_activity(uint32_t a) {
    uint32_t c = a < 7u ? a : 6u;
    c += (a >= 8u) + (a >= 10u) + (a >= 13u) + (a >= 17u) + (a >= 22u) + (a >= 30u) + (a >= 45u) + (a >= 70u) +
         (a >= 120u);
    return c;
}

__device__ __forceinline__ uint32_t mag_of(uint32_t u) {
    return min((u >> 1) + (u & 1u), 127u);
}

// chunk-local coordinates of the elements of a row
struct RowPos {
    uint32_t z, y, x;
};

template <int TS>
struct Taps {
    // geometry of one chunk (wave-uniform)
    uint32_t ex, ey, plane, n;
    size_t sy, sz;              // volume strides (elements) of y and z
    bool wide_x, wide_p;        // ex >= 64 / plane >= 64: tap multiples are 1 for every lane

    // which taps element (i; z, y) uses, and their multiples
    __device__ __forceinline__ void flags(uint32_t i, uint32_t z, uint32_t y, bool act, bool& U, bool& B,
                                          uint32_t& ku, uint32_t& kb) const {
        const uint32_t lane = i & 63u;
        ku = wide_x ? 1u : lane / ex + 1u;
        kb = wide_p ? 1u : lane / plane + 1u;
        U = act && y >= ku && ku * ex <= TAP_LIMIT;
        B = act && z >= kb && (uint64_t)kb * plane <= TAP_LIMIT;
    }

    // zigzag residual of element i at volume offset `off` (relative to the chunk's first element)
    __device__ __forceinline__ uint32_t resid(const void* __restrict__ vol, size_t base, uint32_t i, uint32_t z,
                                              uint32_t y, size_t off, bool act) const {
        if (TS == 4) {
            const int32_t v = static_cast<const int32_t*>(vol)[base + (act ? off : 0)];
            return act ? ((uint32_t)v << 1) ^ (uint32_t)(v >> 31) : 0u;
        }
        bool U, B;
        uint32_t ku, kb;
        flags(i, z, y, act, U, B, ku, kb);
        const uint16_t* v16 = static_cast<const uint16_t*>(vol) + base;
        const uint32_t v = v16[act ? off : 0];
        const uint32_t vu = v16[U ? off - (size_t)ku * sy : 0];
        const uint32_t vb = v16[B ? off - (size_t)kb * sz : 0];
        const uint32_t pred = U && B ? (vu + vb + 1u) >> 1 : (U ? vu : (B ? vb : 0u));
        const int32_t r = (int32_t)(int16_t)(uint16_t)(v - pred);
        return act ? (uint32_t)(((r << 1) ^ (r >> 15)) & 0xFFFF) : 0u;
    }
};

struct Model {
    uint32_t s, nb, e, ctx;
};

// symbol, raw-bit count and raw value of a zigzag residual
__device__ __forceinline__ void symbol_of(uint32_t u, uint32_t& s, uint32_t& nb, uint32_t& e) {
    const uint32_t w = u - 32u, t = (w >> 2) + 1u;
    const uint32_t c = 31u - (uint32_t)__clz((int)t);
    const bool direct = u < 32u;
    s = direct ? u : 32u + c;
    nb = direct ? 0u : 2u + c;
    e = direct ? 0u : w - (((1u << c) - 1u) << 2);
}

// (symbol, raw bits, context) of element i = 64 r + lane of the chunk; coordinates from the caller
template <int TS>
__device__ __forceinline__ Model model_of(const void* __restrict__ vol, size_t base, const Taps<TS>& t, uint32_t i,
                                          const RowPos& p, bool act) {
    const size_t off = (size_t)p.z * t.sz + (size_t)p.y * t.sy + p.x;
    bool U, B;
    uint32_t ku, kb;
    t.flags(i, p.z, p.y, act, U, B, ku, kb);
    const uint32_t u = t.resid(vol, base, i, p.z, p.y, off, act);
    // the taps' own residuals (their lanes, hence their tap multiples, are their own)
    const uint32_t ju = i - ku * t.ex, jb = i - kb * t.plane;
    const uint32_t mu = mag_of(t.resid(vol, base, ju, p.z, p.y - ku, off - (size_t)ku * t.sy, U));
    const uint32_t mb = mag_of(t.resid(vol, base, jb, p.z - kb, p.y, off - (size_t)kb * t.sz, B));
    const uint32_t a = U && B ? mu + mb : (U ? 2u * mu : (B ? 2u * mb : 0u));
    Model m;
    symbol_of(u, m.s, m.nb, m.e);
    m.ctx = ctx_of_activity(a);
    return m;
}

// coordinates of element i of a chunk whose rows do not line up with the volume's x-rows
__device__ __forceinline__ RowPos pos_of(uint32_t i, uint32_t ex, uint32_t ey) {
    RowPos p;
    p.x = i % ex;
    const uint32_t t = i / ex;
    p.y = t % ey;
    p.z = t / ey;
    return p;
}

// one rANS step: renormalise (append the low word of the lanes that must, in lane order), then
// x = (x / f) * 4096 + x mod f + c with the division as a multiply-high by `rcp` >> `shift`
__device__ __forceinline__ void renorm_put(uint32_t& x, uint32_t f, bool act, uint16_t* __restrict__ out,
                                           uint32_t& nwords) {
    const bool emit = act && x >= (f << 19);
    const uint64_t em = __ballot(emit);
    if (emit) {
        out[nwords + rank_below(em)] = (uint16_t)(x & 0xFFFFu);
        x >>= 16;
    }
    nwords += (uint32_t)__popcll(em);
}

#ifndef EXABM4D_ENC2_RB
#define EXABM4D_ENC2_RB 4       // rows whose loads are in flight together
#endif
#ifndef EXABM4D_ENC2_NC
#define EXABM4D_ENC2_NC 2       // copies of every histogram counter (copy = lane mod NC)
#endif

}  // namespace

// ---- encode ---------------------------------------------------------------------------------------------------
// Two kernels.  (1) rans2_model_kernel: every element's (context, symbol, raw bits) -- embarrassingly
// parallel, many workgroups per chunk -- written as one packed code per element into scratch, and
// counted into the chunk's 16 x 64 histogram (LDS per workgroup, then global atomics on the non-zero
// bins).  (2) rans2_code_kernel: one wave per chunk turns the histograms into tables (lane = symbol),
// writes header + tables into the chunk's slot and walks the codes from the last row to the first;
// that loop is the serial rANS chain and touches nothing but the codes and an 8 KB table in LDS.
// Codes: TS = 2: u32 = ctx | s << 4 | e << 10 (e < 2^15).  TS = 4: u16 = ctx | s << 4; the coder
// takes the raw bits from the element itself (zigzag of the value, no taps needed).
// Slot (scratch, per chunk): [0, 276) header, [276, 276 + 2048) table bytes, u32 table length at
// SLOT2_TABLEN, 16-bit words from g.slot_hdr.
constexpr int MODEL_WAVES = 4;            // waves per workgroup of the model kernel
constexpr int MODEL_ROWS = 512;           // rows of 64 elements per workgroup (8 planes of a 64^3 chunk)

template <int TS>
__device__ __forceinline__ void put_code(void* __restrict__ codes, size_t at, const Model& m) {
    if (TS == 2)
        static_cast<uint32_t*>(codes)[at] = m.ctx | (m.s << 4) | (m.e << 10);
    else
        static_cast<uint16_t*>(codes)[at] = (uint16_t)(m.ctx | (m.s << 4));
}

// Generic form: any chunk shape, every element models itself (three residuals for uint16; for int32
// the residual is the zigzag value, so this is also the fast form of that kind).
template <int TS>
__global__ __launch_bounds__(64 * MODEL_WAVES) void rans2_model_kernel(const void* __restrict__ vol, CodecGeom g,
                                                                        int blocks_per_chunk,
                                                                        void* __restrict__ codes,
                                                                        uint32_t* __restrict__ ghist) {
    constexpr int NC = EXABM4D_ENC2_NC, RB = EXABM4D_ENC2_RB;
    __shared__ uint32_t hist[NCTX * NSYM * NC];
    const int c = blockIdx.x / blocks_per_chunk, blk = blockIdx.x % blocks_per_chunk;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const ChunkBox b = chunk_box(g, c);
    const uint32_t n = b.n;
    const uint32_t rows = (n + 63u) >> 6;
    const uint32_t r_lo = (uint32_t)blk * MODEL_ROWS, r_hi = min(rows, r_lo + (uint32_t)MODEL_ROWS);
    for (uint32_t j = threadIdx.x; j < (uint32_t)(NCTX * NSYM * NC); j += 64 * MODEL_WAVES) hist[j] = 0u;
    __syncthreads();
    if (r_lo < r_hi) {
        const bool fast = (b.ex & 63) == 0;
        Taps<TS> t;
        t.ex = (uint32_t)b.ex;
        t.ey = (uint32_t)b.ey;
        t.plane = t.ex * t.ey;
        t.n = n;
        t.sy = (size_t)g.nx;
        t.sz = (size_t)g.nx * g.ny;
        t.wide_x = t.ex >= 64u;
        t.wide_p = t.plane >= 64u;
        RowCursor rc;
        rc.rpx = (uint32_t)b.ex >> 6;
        rc.ey = (uint32_t)b.ey;
        const size_t cbase = (size_t)c * g.chunk_elems;
        for (uint32_t r0 = r_lo + wave * RB; r0 < r_hi; r0 += MODEL_WAVES * RB) {
            if (fast) rc.seek(r0);
            Model m[RB];
            bool act[RB];
#pragma unroll
            for (int k = 0; k < RB; k++) {
                const uint32_t r = r0 + k, i = r * 64u + lane;
                act[k] = r < r_hi && i < n;
                RowPos p;
                if (fast) {
                    p.x = rc.xr * 64u + lane;
                    p.y = rc.y;
                    p.z = rc.z;
                    if (r + 1 < r_hi) rc.next();
                } else {
                    p = pos_of(min(i, n - 1u), t.ex, t.ey);
                }
                m[k] = model_of<TS>(vol, b.base, t, i, p, act[k]);
            }
#pragma unroll
            for (int k = 0; k < RB; k++)
                if (act[k]) {
                    atomicAdd(&hist[(m[k].ctx * NSYM + m[k].s) * NC + (lane & (NC - 1))], 1u);
                    put_code<TS>(codes, cbase + (size_t)(r0 + k) * 64u + lane, m[k]);
                }
        }
    }
    __syncthreads();
    uint32_t* gh = ghist + (size_t)c * (NCTX * NSYM);
    for (uint32_t j = threadIdx.x; j < (uint32_t)(NCTX * NSYM); j += 64 * MODEL_WAVES) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < NC; k++) v += hist[j * NC + k];
        if (v) atomicAdd(&gh[j], v);
    }
}

// uint16 chunks of 64-element rows and at most 64 rows per plane -- the reference's 64^3 chunks
// (utils/img_util.py:401): no workgroup barrier and one residual per element.  A wave owns a strip of
// 16 consecutive rows of every plane of its z-block and the lane is x: the tap above a row is the row
// the wave has just loaded, the tap behind it is what the wave held for the plane before, and the
// same holds for the two magnitudes of the context -- all in registers.  Only the strip's halo row
// (the row above its first) is loaded and modelled a second time.
constexpr int STRIP = 16;
__global__ __launch_bounds__(64 * MODEL_WAVES) void rans2_model_strips_kernel(const uint16_t* __restrict__ vol,
                                                                               CodecGeom g, int blocks_per_chunk,
                                                                               int planes_per_block,
                                                                               uint32_t* __restrict__ codes,
                                                                               uint32_t* __restrict__ ghist) {
    constexpr int NC = EXABM4D_ENC2_NC;
    __shared__ uint32_t hist[NCTX * NSYM * NC];
    __shared__ uint8_t lut[256];
    const int c = blockIdx.x / blocks_per_chunk, blk = blockIdx.x % blocks_per_chunk;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const ChunkBox b = chunk_box(g, c);
    const uint32_t ey = (uint32_t)b.ey, ez = b.n / (64u * ey);
    const uint32_t z_lo = (uint32_t)blk * (uint32_t)planes_per_block, z_hi = min(ez, z_lo + (uint32_t)planes_per_block);
    for (uint32_t j = threadIdx.x; j < (uint32_t)(NCTX * NSYM * NC); j += 64 * MODEL_WAVES) hist[j] = 0u;
    lut[threadIdx.x] = (uint8_t)ctx_of_activity(threadIdx.x);
    __syncthreads();
    const uint32_t q0 = wave * STRIP;                       // first row of the strip
    if (z_lo < z_hi && q0 < ey) {
        const size_t sy = (size_t)g.nx, sz = (size_t)g.nx * g.ny;
        const uint16_t* v16 = vol + b.base + lane;
        uint32_t* crow = codes + (size_t)c * g.chunk_elems + lane;
        auto zig = [](uint32_t v, uint32_t pred) -> uint32_t {
            const int32_t r = (int32_t)(int16_t)(uint16_t)(v - pred);
            return (uint32_t)(((r << 1) ^ (r >> 15)) & 0xFFFF);
        };
        // prediction of a voxel whose taps are given; U / B say which exist
        auto pred_of = [](uint32_t vu, uint32_t vb, bool U, bool B) -> uint32_t {
            return U && B ? (vu + vb + 1u) >> 1 : (U ? vu : (B ? vb : 0u));
        };
        uint32_t vprev[STRIP], mprev[STRIP];
#pragma unroll
        for (int k = 0; k < STRIP; k++) vprev[k] = mprev[k] = 0u;
        if (z_lo > 0u) {                                     // plane before the block: values and magnitudes
            const size_t zo = (size_t)(z_lo - 1u) * sz;
            const bool B = z_lo > 1u;
            uint32_t vb2[STRIP];
            uint32_t vh = 0u;
            if (q0 > 0u) vh = v16[zo + (size_t)(q0 - 1u) * sy];
#pragma unroll
            for (int k = 0; k < STRIP; k++) {
                const uint32_t q = q0 + k;
                vprev[k] = q < ey ? v16[zo + (size_t)q * sy] : 0u;
                vb2[k] = (q < ey && B) ? v16[zo - sz + (size_t)q * sy] : 0u;
            }
#pragma unroll
            for (int k = 0; k < STRIP; k++) {
                const uint32_t q = q0 + k;
                const uint32_t up = k ? vprev[k ? k - 1 : 0] : vh;
                mprev[k] = mag_of(zig(vprev[k], pred_of(up, vb2[k], q > 0u, B)));
            }
        }
        for (uint32_t z = z_lo; z < z_hi; z++) {
            const size_t zo = (size_t)z * sz;
            const bool B = z > 0u;
            uint32_t v[STRIP], m[STRIP];
#pragma unroll
            for (int k = 0; k < STRIP; k++) {
                const uint32_t q = q0 + k;
                v[k] = q < ey ? v16[zo + (size_t)q * sy] : 0u;
            }
            // halo row q0 - 1: its value (tap of the strip's first row) and its magnitude
            uint32_t vh = 0u, mh = 0u;
            if (q0 > 0u) {
                const size_t ho = zo + (size_t)(q0 - 1u) * sy;
                vh = v16[ho];
                const uint32_t vhu = q0 > 1u ? v16[ho - sy] : 0u;
                const uint32_t vhb = B ? v16[ho - sz] : 0u;
                mh = mag_of(zig(vh, pred_of(vhu, vhb, q0 > 1u, B)));
            }
#pragma unroll
            for (int k = 0; k < STRIP; k++) {
                const uint32_t q = q0 + k;
                if (q >= ey) break;                          // wave-uniform
                const bool U = q > 0u;
                const uint32_t up = k ? v[k ? k - 1 : 0] : vh;
                const uint32_t u = zig(v[k], pred_of(up, vprev[k], U, B));
                m[k] = mag_of(u);
                const uint32_t mu = k ? m[k ? k - 1 : 0] : mh, mb = mprev[k];
                const uint32_t a = U && B ? mu + mb : (U ? 2u * mu : (B ? 2u * mb : 0u));
                const uint32_t ctx = lut[a];
                uint32_t sy_, nb_, e_;
                symbol_of(u, sy_, nb_, e_);
                atomicAdd(&hist[(ctx * NSYM + sy_) * NC + (lane & (NC - 1))], 1u);
                crow[((size_t)z * ey + q) * 64u] = ctx | (sy_ << 4) | (e_ << 10);
            }
#pragma unroll
            for (int k = 0; k < STRIP; k++) {
                vprev[k] = v[k];
                mprev[k] = (q0 + k < ey) ? m[k] : 0u;
            }
        }
    }
    __syncthreads();
    uint32_t* gh = ghist + (size_t)c * (NCTX * NSYM);
    for (uint32_t j = threadIdx.x; j < (uint32_t)(NCTX * NSYM); j += 64 * MODEL_WAVES) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < NC; k++) v += hist[j * NC + k];
        if (v) atomicAdd(&gh[j], v);
    }
}

// int32 chunks whose rows are x-rows of the volume (ex a multiple of 64): the residual of an element is
// the zigzag of its value, so a row needs three coalesced loads -- itself, the row above, the row a
// plane before, all from wave-uniform bases -- and no per-lane index arithmetic.  This is the form of
// the DCT-index leg (chunks of 512 blocks x 8 x 64 coefficients).
__global__ __launch_bounds__(64 * MODEL_WAVES) void rans2_model_rows32_kernel(const int32_t* __restrict__ vol,
                                                                               CodecGeom g, int blocks_per_chunk,
                                                                               uint16_t* __restrict__ codes,
                                                                               uint32_t* __restrict__ ghist) {
    constexpr int NC = EXABM4D_ENC2_NC, RB = 8;
    __shared__ uint32_t hist[NCTX * NSYM * NC];
    __shared__ uint8_t lut[256];
    const int c = blockIdx.x / blocks_per_chunk, blk = blockIdx.x % blocks_per_chunk;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const ChunkBox b = chunk_box(g, c);
    const uint32_t rows = b.n >> 6;                          // whole rows only (ex is a multiple of 64)
    const uint32_t r_lo = (uint32_t)blk * MODEL_ROWS, r_hi = min(rows, r_lo + (uint32_t)MODEL_ROWS);
    for (uint32_t j = threadIdx.x; j < (uint32_t)(NCTX * NSYM * NC); j += 64 * MODEL_WAVES) hist[j] = 0u;
    lut[threadIdx.x] = (uint8_t)ctx_of_activity(threadIdx.x);
    __syncthreads();
    if (r_lo < r_hi) {
        const uint32_t ex = (uint32_t)b.ex, ey = (uint32_t)b.ey, rpx = ex >> 6;
        const bool use_u = ex <= TAP_LIMIT, use_b = (uint64_t)ex * ey <= TAP_LIMIT;
        const size_t sy = (size_t)g.nx, sz = (size_t)g.nx * g.ny;
        const int32_t* v32 = vol + b.base + lane;
        uint16_t* crow = codes + (size_t)c * g.chunk_elems + lane;
        auto zz = [](int32_t v) -> uint32_t { return ((uint32_t)v << 1) ^ (uint32_t)(v >> 31); };
        RowCursor rc;
        rc.rpx = rpx;
        rc.ey = ey;
        for (uint32_t r0 = r_lo + wave * RB; r0 < r_hi; r0 += MODEL_WAVES * RB) {
            rc.seek(r0);
            int32_t v[RB], vu[RB], vb[RB];
            bool U[RB], B[RB];
#pragma unroll
            for (int k = 0; k < RB; k++) {
                const bool act = r0 + k < r_hi;              // wave-uniform
                const size_t off = act ? ((size_t)rc.z * sz + (size_t)rc.y * sy + rc.xr * 64u) : 0;
                U[k] = act && use_u && rc.y > 0u;
                B[k] = act && use_b && rc.z > 0u;
                v[k] = v32[off];
                vu[k] = v32[U[k] ? off - sy : off];
                vb[k] = v32[B[k] ? off - sz : off];
                if (r0 + k + 1 < r_hi) rc.next();
            }
#pragma unroll
            for (int k = 0; k < RB; k++) {
                if (r0 + k >= r_hi) break;
                const uint32_t mu = mag_of(zz(vu[k])), mb = mag_of(zz(vb[k]));
                const uint32_t a = U[k] && B[k] ? mu + mb : (U[k] ? 2u * mu : (B[k] ? 2u * mb : 0u));
                const uint32_t ctx = lut[a];
                uint32_t s_, nb_, e_;
                symbol_of(zz(v[k]), s_, nb_, e_);
                atomicAdd(&hist[(ctx * NSYM + s_) * NC + (lane & (NC - 1))], 1u);
                crow[(size_t)(r0 + k) * 64u] = (uint16_t)(ctx | (s_ << 4));
            }
        }
    }
    __syncthreads();
    uint32_t* gh = ghist + (size_t)c * (NCTX * NSYM);
    for (uint32_t j = threadIdx.x; j < (uint32_t)(NCTX * NSYM); j += 64 * MODEL_WAVES) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < NC; k++) v += hist[j * NC + k];
        if (v) atomicAdd(&gh[j], v);
    }
}

template <int TS>
__global__ __launch_bounds__(64) void rans2_code_kernel(const void* __restrict__ vol, CodecGeom g,
                                                        const uint2* __restrict__ rcp_tab,
                                                        const void* __restrict__ codes,
                                                        const uint32_t* __restrict__ ghist,
                                                        uint8_t* __restrict__ slots, uint32_t* __restrict__ sizes) {
    constexpr int RB = 8;
    __shared__ uint2 etab[NCTX * NSYM];
    const int c = blockIdx.x;
    const uint32_t lane = lane_id();
    const ChunkBox b = chunk_box(g, c);
    const uint32_t n = b.n;
    const uint32_t rows = (n + 63u) >> 6;
    uint8_t* slot = slots + (size_t)c * g.slot_bytes;

    // -- tables: lane = symbol ----------------------------------------------------------------------------------
    uint32_t toff = 0;
    bool coded = false;
    uint8_t* tab = slot + HDR2;
    const uint32_t* gh = ghist + (size_t)c * (NCTX * NSYM);
    for (int q = 0; q < NCTX; q++) {
        const uint32_t cnt = gh[q * NSYM + lane];
        const uint32_t tot = wave_sum(cnt);
        uint32_t F = 0;
        if (cnt) {
            const uint32_t f = (uint32_t)(((uint64_t)cnt << RANS_BITS) / tot);
            F = f < 1u ? 1u : f;
        }
        uint32_t sum = wave_sum(F);
        if (tot) {
            uint32_t best = wave_max((F << 8) | (63u - lane));
            const int32_t diff = (int32_t)RANS_M - (int32_t)sum;
            if ((int32_t)(best >> 8) + diff >= 1) {
                if (lane == 63u - (best & 255u)) F = (uint32_t)((int32_t)F + diff);
            } else {
                while (sum > RANS_M) {
                    best = wave_max((F << 8) | (63u - lane));
                    if (lane == 63u - (best & 255u)) F -= 1u;
                    sum--;
                }
            }
        }
        const uint64_t pm = __ballot(F != 0u), wm = __ballot(F != 0u && F - 1u >= 256u);
        const uint32_t np = (uint32_t)__popcll(pm), nw = (uint32_t)__popcll(wm);
        if (lane == 0) {                                          // (offsets 20 / 148: 4-byte aligned)
            uint32_t* hp = reinterpret_cast<uint32_t*>(slot + 20) + 2 * q;
            uint32_t* hw = reinterpret_cast<uint32_t*>(slot + 148) + 2 * q;
            hp[0] = (uint32_t)pm;
            hp[1] = (uint32_t)(pm >> 32);
            hw[0] = (uint32_t)wm;
            hw[1] = (uint32_t)(wm >> 32);
        }
        if (F) tab[toff + rank_below(pm)] = (uint8_t)((F - 1u) & 255u);
        if (F && F - 1u >= 256u) tab[toff + np + rank_below(wm)] = (uint8_t)((F - 1u) >> 8);
        toff += np + nw;
        coded = coded || np > 1u || (pm >> 32) != 0ull;
        const uint32_t C = wave_excl_scan(F, lane);
        uint2 e = make_uint2(0u, 0u);
        if (F) {
            const uint2 rs = rcp_tab[F];
            const uint32_t bias = F == 1u ? C + RANS_M - 1u : C;
            e.x = F | (bias << 13) | (rs.y << 26);
            e.y = rs.x;
        }
        etab[lane * NCTX + q] = e;              // indexed by the low ten bits of a code: ctx | s << 4
    }
    if (toff & 1u) {
        if (lane == 0) tab[toff] = 0;
        toff++;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // -- rows from the last to the first ---------------------------------------------------------------------
    // The last rows -- a partial row, and whatever keeps the rest from being whole batches -- go one
    // at a time with a per-lane predicate; all other rows are whole and run as batches of RB with
    // nothing predicated: RB code loads and table look-ups in flight, then the serial state chain.
    uint32_t nwords = 0, x = RANS_L;
    uint16_t* out = reinterpret_cast<uint16_t*>(slot + g.slot_hdr);
    if (coded) {
        const bool fast = (b.ex & 63) == 0;
        RowCursor rc;
        rc.rpx = (uint32_t)b.ex >> 6;
        rc.ey = (uint32_t)b.ey;
        rc.xr = rc.y = rc.z = 0;
        const size_t cbase = (size_t)c * g.chunk_elems;
        const uint32_t* c32 = static_cast<const uint32_t*>(codes) + (TS == 2 ? cbase : 0);
        const uint16_t* c16 = static_cast<const uint16_t*>(codes) + (TS == 4 ? cbase : 0);

        // code word and raw value of element i of row r
        auto fetch = [&](uint32_t r, uint32_t i, bool act, uint32_t& code, uint32_t& ev) {
            if (TS == 2) {
                code = act ? c32[i] : 0u;
                ev = code >> 10;
            } else {
                int32_t v = 0;
                code = 0u;
                if (act) {
                    code = c16[i];
                    const size_t off = fast ? rc.offset(g) + lane : elem_offset(g, b, i);
                    v = static_cast<const int32_t*>(vol)[b.base + off];
                }
                const uint32_t u = ((uint32_t)v << 1) ^ (uint32_t)(v >> 31);
                uint32_t s_, nb_;
                symbol_of(u, s_, nb_, ev);
            }
        };
        // one row through the coder: raw bits (last step first: the decoder reads them low bits
        // first), then the symbol
        auto code_row = [&](uint32_t code, uint32_t ev, uint2 e, bool act) {
            if (__ballot(act && (code & 0x200u)) != 0ull) {          // a symbol >= 32 somewhere in the row
                const uint32_t s = (code >> 4) & 63u;
                const uint32_t nb = s < 32u ? 0u : s - 30u;
#pragma unroll
                for (int j = (TS == 2 ? 1 : 2); j >= 0; j--) {
                    const bool has = act && nb > 12u * (uint32_t)j;
                    if (__ballot(has) == 0ull) continue;
                    const uint32_t kk = min(nb - 12u * (uint32_t)j, 12u);      // garbage where !has
                    const uint32_t f = RANS_M >> (has ? kk : 0u);
                    renorm_put(x, f, has, out, nwords);
                    if (has) {
                        const uint32_t val = (ev >> (12 * j)) & ((1u << kk) - 1u);
                        x = ((x >> (12u - kk)) << RANS_BITS) | (x & (f - 1u)) | (val << (12u - kk));
                    }
                }
            }
            const uint32_t f = e.x & 0x1FFFu;
            const bool emit = act && x >= (e.x << 19);               // F << 19: the upper fields shift out
            const uint64_t em = __ballot(emit);
            if (emit) {
                out[nwords + rank_below(em)] = (uint16_t)(x & 0xFFFFu);
                x >>= 16;
            }
            nwords += (uint32_t)__popcll(em);
            const uint32_t qd = __umulhi(x, e.y) >> (e.x >> 26);
            const uint32_t nx = x + ((e.x >> 13) & 0x1FFFu) + qd * (RANS_M - f);
            x = act ? nx : x;
        };

        const uint32_t whole = n >> 6;                               // rows without an inactive lane
        const uint32_t batched = (whole / RB) * RB;                  // rows [0, batched) run as batches
        if (TS == 4 && fast && rows) rc.seek(rows - 1);
        for (uint32_t r = rows; r > batched; r--) {
            const uint32_t i = (r - 1u) * 64u + lane;
            const bool act = i < n;
            uint32_t code, ev;
            fetch(r - 1u, i, act, code, ev);
            if (TS == 4 && fast) rc.prev();
            code_row(code, ev, etab[code & 0x3FFu], act);
        }
        for (uint32_t rb = batched; rb > 0; rb -= RB) {
            uint32_t code[RB], ev[RB];
            uint2 e[RB];
#pragma unroll
            for (int k = 0; k < RB; k++) {
                const uint32_t r = rb - 1 - k;
                fetch(r, r * 64u + lane, true, code[k], ev[k]);
                if (TS == 4 && fast) rc.prev();
            }
#pragma unroll
            for (int k = 0; k < RB; k++) e[k] = etab[code[k] & 0x3FFu];
#pragma unroll
            for (int k = 0; k < RB; k++) code_row(code[k], ev[k], e[k], true);
        }
        out[nwords + 2 * lane] = (uint16_t)(x & 0xFFFFu);
        out[nwords + 2 * lane + 1] = (uint16_t)(x >> 16);
        nwords += 128;
    }
    if (lane == 0) {
        slot[0] = 'E';
        slot[1] = 'X';
        slot[2] = 2;
        slot[3] = (uint8_t)TS;
        uint32_t* h = reinterpret_cast<uint32_t*>(slot);
        h[1] = n;
        h[2] = (uint32_t)b.ey;
        h[3] = (uint32_t)b.ex;
        h[4] = nwords;
        *reinterpret_cast<uint32_t*>(slot + SLOT2_TABLEN) = toff;
        sizes[c] = (uint32_t)HDR2 + toff + 2u * nwords;
    }
}

// slot -> packed stream at out + offsets[c]: header + tables (an even number of bytes), then the words
__global__ __launch_bounds__(256) void rans2_pack_kernel(const uint8_t* __restrict__ slots, CodecGeom g,
                                                         const unsigned long long* __restrict__ offsets,
                                                         const uint32_t* __restrict__ sizes,
                                                         uint8_t* __restrict__ out) {
    const int c = blockIdx.x;
    const uint8_t* slot = slots + (size_t)c * g.slot_bytes;
    uint16_t* dst = reinterpret_cast<uint16_t*>(out + offsets[c]);
    const uint32_t head = ((uint32_t)HDR2 + *reinterpret_cast<const uint32_t*>(slot + SLOT2_TABLEN)) / 2u;
    const uint32_t nw = reinterpret_cast<const uint32_t*>(slot)[4];
    const uint16_t* src = reinterpret_cast<const uint16_t*>(slot);
    for (uint32_t i = threadIdx.x; i < head; i += 256) dst[i] = src[i];
    src = reinterpret_cast<const uint16_t*>(slot + g.slot_hdr);
    for (uint32_t i = threadIdx.x; i < nw; i += 256) dst[head + i] = src[i];
    const uint32_t sz = sizes[c], padded = (sz + 15u) & ~15u;
    for (uint32_t i = sz / 2 + threadIdx.x; i < padded / 2; i += 256) dst[i] = 0;
}

// ---- decode: one wave per chunk ------------------------------------------------------------------------------
// status bits: 1 header / sizes, 2 tables, 4 word stream exhausted, 8 offsets, 16 symbol out of range
template <int TS>
__global__ __launch_bounds__(64) void rans2_decode_kernel(const uint8_t* __restrict__ in, size_t in_bytes,
                                                          const unsigned long long* __restrict__ offsets,
                                                          CodecGeom g, void* __restrict__ vol,
                                                          uint32_t* __restrict__ status, uint32_t ring) {
    // History ring of `ring` elements (a multiple of 64 that covers the farthest tap of any chunk of
    // this volume plus the row being written: decode_ring_elems): magnitudes, and for uint16 the
    // values.  Sized per launch -- 4224 elements for 64^3 chunks instead of a fixed 8192 -- because
    // LDS is what limits the resident waves of this one-wave-per-chunk kernel.
    extern __shared__ __align__(16) uint8_t dyn_lds[];
    uint16_t* cum = reinterpret_cast<uint16_t*>(dyn_lds);
    uint16_t* vring = cum + ((NCTX * (NSYM + 1) + 7) & ~7);
    uint8_t* mring = reinterpret_cast<uint8_t*>(vring + (TS == 2 ? ring : 0));
    const int c = blockIdx.x;
    const uint32_t lane = lane_id();
    const ChunkBox b = chunk_box(g, c);
    const uint32_t n = b.n;
    const uint32_t rows = (n + 63u) >> 6;
    const bool fast = (b.ex & 63) == 0;
    const unsigned long long o0 = offsets[c], o1 = offsets[c + 1];
    if (o0 > o1 || o1 > in_bytes || (o0 & 1ull)) {
        if (lane == 0) atomicOr(status, 8u);
        return;
    }
    const uint8_t* s0 = in + o0;
    const size_t avail = (size_t)(o1 - o0);
    const uint32_t* h = reinterpret_cast<const uint32_t*>(s0);
    bool ok = avail >= (size_t)HDR2 && s0[0] == 'E' && s0[1] == 'X' && s0[2] == 2 && s0[3] == TS && h[1] == n &&
              h[2] == (uint32_t)b.ey && h[3] == (uint32_t)b.ex;
    if (!ok) {
        if (lane == 0) atomicOr(status, 1u);
        return;
    }
    const uint32_t nwords = h[4];
    // table sizes: lane q < 16 owns context q
    uint64_t pm = 0, wm = 0;
    if (lane < NCTX) {
        const uint32_t* p32 = reinterpret_cast<const uint32_t*>(s0 + 20) + 2 * lane;
        const uint32_t* w32 = reinterpret_cast<const uint32_t*>(s0 + 148) + 2 * lane;
        pm = (uint64_t)p32[0] | ((uint64_t)p32[1] << 32);
        wm = (uint64_t)w32[0] | ((uint64_t)w32[1] << 32);
    }
    const uint32_t tsz = (uint32_t)__popcll(pm) + (uint32_t)__popcll(wm);
    const uint32_t tstart = wave_excl_scan(tsz, lane);
    const uint32_t ttot = wave_sum(tsz), tpad = (ttot + 1u) & ~1u;
    ok = __ballot((wm & ~pm) != 0ull) == 0ull && (size_t)HDR2 + tpad + 2 * (size_t)nwords <= avail &&
         (nwords == 0u || nwords >= 128u);
    if (!ok) {
        if (lane == 0) atomicOr(status, 1u);
        return;
    }
    if (n == 0) return;
    const uint8_t* tab = s0 + HDR2;
    bool tables_ok = true;
    for (int q = 0; q < NCTX; q++) {
        const uint64_t pq = __shfl(pm, q, 64), wq = __shfl(wm, q, 64);
        const uint32_t tq = __shfl(tstart, q, 64), np = (uint32_t)__popcll(pq);
        uint32_t F = 0;
        if ((pq >> lane) & 1ull) {
            F = tab[tq + rank_below(pq)];
            if ((wq >> lane) & 1ull) F |= (uint32_t)tab[tq + np + rank_below(wq)] << 8;
            F += 1u;
        }
        const uint32_t C = wave_excl_scan(F, lane);
        const uint32_t sum = wave_sum(F);
        cum[q * (NSYM + 1) + lane] = (uint16_t)C;
        if (lane == 63) cum[q * (NSYM + 1) + NSYM] = (uint16_t)min(sum, 0xFFFFu);
        tables_ok = tables_ok && (np == 0u || sum == RANS_M);
    }
    if (!tables_ok) {
        if (lane == 0) atomicOr(status, 2u);
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    Taps<TS> t;
    t.ex = (uint32_t)b.ex;
    t.ey = (uint32_t)b.ey;
    t.plane = t.ex * t.ey;
    t.n = n;
    t.sy = (size_t)g.nx;
    t.sz = (size_t)g.nx * g.ny;
    t.wide_x = t.ex >= 64u;
    t.wide_p = t.plane >= 64u;
    RowCursor rc;
    rc.rpx = (uint32_t)b.ex >> 6;
    rc.ey = (uint32_t)b.ey;
    rc.xr = rc.y = rc.z = 0;

    const uint16_t* words = reinterpret_cast<const uint16_t*>(s0 + HDR2 + tpad);
    uint32_t cursor = nwords ? nwords - 128u : 0u;
    uint32_t x = RANS_L;
    if (nwords) x = (uint32_t)words[cursor + 2 * lane] | ((uint32_t)words[cursor + 2 * lane + 1] << 16);
    uint32_t bad = 0;
    // renormalise the lanes that need it: take their words in lane order from below the cursor
    auto refill = [&](bool need) -> bool {
        const uint64_t nm = __ballot(need);
        const uint32_t k = (uint32_t)__popcll(nm);
        if (k > cursor) return false;
        cursor -= k;
        if (need) x = (x << 16) | words[cursor + rank_below(nm)];
        return true;
    };
    uint32_t rbase = 0;
    for (uint32_t r = 0; r < rows; r++, rbase = rbase + 64u >= ring ? 0u : rbase + 64u) {
        const uint32_t i = r * 64u + lane;
        const bool act = i < n;
        RowPos p;
        if (fast) {
            p.x = rc.xr * 64u + lane;
            p.y = rc.y;
            p.z = rc.z;
            rc.next();
        } else {
            p = pos_of(min(i, n - 1u), t.ex, t.ey);
        }
        bool U, B;
        uint32_t ku, kb;
        t.flags(i, p.z, p.y, act, U, B, ku, kb);
        // ring slots: this row starts at `rbase` (= 64 r mod ring, kept incrementally), a tap lies
        // its distance back (distances of used taps are < ring - 64)
        const uint32_t slot_i = rbase + lane;
        const uint32_t du = ku * t.ex, db = kb * t.plane;
        const uint32_t ju = U ? (slot_i >= du ? slot_i - du : slot_i + ring - du) : 0u;
        const uint32_t jb = B ? (slot_i >= db ? slot_i - db : slot_i + ring - db) : 0u;
        const uint32_t mu = U ? mring[ju] : 0u, mb = B ? mring[jb] : 0u;
        const uint32_t a = U && B ? mu + mb : (U ? 2u * mu : (B ? 2u * mb : 0u));
        const uint32_t q = ctx_of_activity(a);
        uint32_t pred = 0;
        if (TS == 2) {
            const uint32_t vu = U ? vring[ju] : 0u, vb = B ? vring[jb] : 0u;
            pred = U && B ? (vu + vb + 1u) >> 1 : (U ? vu : vb);
        }
        // (1) symbol: binary search of the slot in the context's cumulative table
        uint32_t s = 0, nb = 0, e = 0;
        bool need = false;
        if (act) {
            const uint16_t* cq = cum + q * (NSYM + 1);
            const uint32_t slot = x & (RANS_M - 1u);
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1)
                if (cq[s + step] <= slot) s += step;
            const uint32_t C = cq[s], F = (uint32_t)cq[s + 1] - C;
            if (F == 0u) bad |= 2u;            // a context the encoder never used
            x = F * (x >> RANS_BITS) + slot - C;
            nb = s < 32u ? 0u : s - 30u;
            need = x < RANS_L;
        }
        if (!refill(need)) {
            bad |= 4u;
            break;
        }
        // (2) raw bits, low step first
        bool under = false;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const bool has = act && nb > 12u * (uint32_t)j;
            if (__ballot(has) == 0ull) break;
            need = false;
            if (has) {
                const uint32_t kk = min(nb - 12u * (uint32_t)j, 12u), f = RANS_M >> kk;
                const uint32_t slot = x & (RANS_M - 1u);
                e |= (slot >> (12u - kk)) << (12 * j);
                x = f * (x >> RANS_BITS) + (slot & (f - 1u));
                need = x < RANS_L;
            }
            if (!refill(need)) {
                under = true;
                break;
            }
        }
        if (under) {
            bad |= 4u;
            break;
        }
        // (3) value
        if (act) {
            uint64_t u = s;
            if (s >= 32u) {
                const uint32_t cc = s - 32u;
                u = 32ull + ((((uint64_t)1 << cc) - 1ull) << 2) + e;
                if (cc > 29u || (TS == 2 && cc > 13u) || u > (TS == 2 ? 0xFFFFull : 0xFFFFFFFFull)) {
                    bad |= 16u;
                    u = 0;
                }
            }
            const uint32_t u32 = (uint32_t)u;
            mring[slot_i] = (uint8_t)mag_of(u32);
            const size_t off = (size_t)p.z * t.sz + (size_t)p.y * t.sy + p.x;
            if (TS == 2) {
                const uint32_t r16 = (u32 >> 1) ^ (0u - (u32 & 1u));
                const uint16_t v = (uint16_t)(pred + r16);
                vring[slot_i] = v;
                static_cast<uint16_t*>(vol)[b.base + off] = v;
            } else {
                static_cast<int32_t*>(vol)[b.base + off] = (int32_t)((u32 >> 1) ^ (0u - (u32 & 1u)));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const uint64_t bm = __ballot(bad != 0u);
    if (bm) {
        uint32_t all = bad;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) all |= (uint32_t)__shfl_xor(all, o, 64);
        if (lane == 0) atomicOr(status, all);
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------
size_t codec2_chunk_bound(size_t n, int ts) {
    // header, worst-case tables, one word per coding step (symbol + 2 / 3 raw steps) + final states
    return (size_t)HDR2 + TAB2_MAX + 2 * ((size_t)(ts == 2 ? 3 : 4) * n + 128);
}

void codec2_slot_layout(size_t chunk_elems, int ts, size_t& slot_hdr, size_t& slot_bytes) {
    slot_hdr = ((size_t)SLOT2_TABLEN + 4 + 15) & ~(size_t)15;
    slot_bytes = slot_hdr + ((2 * ((size_t)(ts == 2 ? 3 : 4) * chunk_elems + 128) + 15) & ~(size_t)15);
}

size_t codec2_work_bytes(const CodecGeom& g) {
    // packed codes of every element (u32 / u16) + the 16 x 64 histogram of every chunk
    return (((size_t)g.nchunks * g.chunk_elems * (g.ts == 2 ? 4 : 2) + 255) & ~(size_t)255) +
           (size_t)g.nchunks * NCTX * NSYM * sizeof(uint32_t);
}

hipError_t launch_rans2_encode(const void* vol, const CodecGeom& g, const uint32_t* rcp_tab, uint8_t* slots,
                               uint8_t* work, uint32_t* sizes, uint8_t* out, const unsigned long long* offsets,
                               int stage, hipStream_t s) {
    const uint2* rt = reinterpret_cast<const uint2*>(rcp_tab);
    if (stage == 1) {
        hipLaunchKernelGGL(rans2_pack_kernel, dim3((unsigned)g.nchunks), dim3(256), 0, s, slots, g, offsets, sizes,
                           out);
        return hipGetLastError();
    }
    void* codes = work;
    const size_t cbytes = ((size_t)g.nchunks * g.chunk_elems * (g.ts == 2 ? 4 : 2) + 255) & ~(size_t)255;
    uint32_t* ghist = reinterpret_cast<uint32_t*>(work + cbytes);
    hipError_t e = hipMemsetAsync(ghist, 0, (size_t)g.nchunks * NCTX * NSYM * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    // every chunk of the volume has ex = min(cx, rest of the row): the strips form needs 64 everywhere
    const bool strips = g.ts == 2 && g.cx == 64 && (g.nx % 64) == 0 && g.cy <= STRIP * MODEL_WAVES;
    if (strips) {
        const int ppb = 16;                                                // planes per workgroup
        const int bpc = (g.cz + ppb - 1) / ppb;
        hipLaunchKernelGGL(rans2_model_strips_kernel, dim3((unsigned)g.nchunks * bpc), dim3(64 * MODEL_WAVES), 0, s,
                           static_cast<const uint16_t*>(vol), g, bpc, ppb, static_cast<uint32_t*>(codes), ghist);
    } else if (g.ts == 4 && (g.cx % 64) == 0 && (g.nx % g.cx) == 0) {
        const size_t rows = g.chunk_elems / 64;
        const int bpc = (int)((rows + MODEL_ROWS - 1) / MODEL_ROWS);
        hipLaunchKernelGGL(rans2_model_rows32_kernel, dim3((unsigned)g.nchunks * bpc), dim3(64 * MODEL_WAVES), 0, s,
                           static_cast<const int32_t*>(vol), g, bpc, static_cast<uint16_t*>(codes), ghist);
    } else {
        const size_t rows = (g.chunk_elems + 63) / 64;
        const int bpc = (int)((rows + MODEL_ROWS - 1) / MODEL_ROWS);
        if (g.ts == 2)
            hipLaunchKernelGGL(rans2_model_kernel<2>, dim3((unsigned)g.nchunks * bpc), dim3(64 * MODEL_WAVES), 0, s,
                               vol, g, bpc, codes, ghist);
        else
            hipLaunchKernelGGL(rans2_model_kernel<4>, dim3((unsigned)g.nchunks * bpc), dim3(64 * MODEL_WAVES), 0, s,
                               vol, g, bpc, codes, ghist);
    }
    if (g.ts == 2)
        hipLaunchKernelGGL(rans2_code_kernel<2>, dim3((unsigned)g.nchunks), dim3(64), 0, s, vol, g, rt, codes, ghist,
                           slots, sizes);
    else
        hipLaunchKernelGGL(rans2_code_kernel<4>, dim3((unsigned)g.nchunks), dim3(64), 0, s, vol, g, rt, codes, ghist,
                           slots, sizes);
    return hipGetLastError();
}

// elements of decoder history a volume's chunks need: the farthest tap any lane of any chunk uses
// (multiples of ex / ey * ex up to TAP_LIMIT; for rows or planes narrower than a wave up to 64 more)
// plus the row being written, as a multiple of 64
static uint32_t decode_ring_elems(const CodecGeom& g) {
    auto reach = [](size_t stride) -> size_t {
        if (stride > TAP_LIMIT) return 0;
        return stride >= 64 ? stride : ((63 / stride) + 1) * stride;      // k = lane / stride + 1, lane <= 63
    };
    // edge chunks are smaller than (cy, cx): take the largest reach over the extents that occur
    size_t far = 0;
    const int exs[2] = {g.cx, g.nx % g.cx ? g.nx % g.cx : g.cx};
    const int eys[2] = {g.cy, g.ny % g.cy ? g.ny % g.cy : g.cy};
    for (int a = 0; a < 2; a++) {
        far = std::max(far, reach((size_t)exs[a]));
        for (int b = 0; b < 2; b++) far = std::max(far, reach((size_t)exs[a] * eys[b]));
    }
    size_t ring = ((far + 64 + 63) / 64) * 64;
    if (ring > RING_MAX) ring = RING_MAX;
    return (uint32_t)ring;
}

hipError_t launch_rans2_decode(const uint8_t* in, size_t in_bytes, const unsigned long long* offsets,
                               const CodecGeom& g, void* vol, uint32_t* status, hipStream_t s) {
    const uint32_t ring = decode_ring_elems(g);
    const size_t lds = 2 * (size_t)((NCTX * (NSYM + 1) + 7) & ~7) + (g.ts == 2 ? 2 * (size_t)ring : 0) + ring;
    if (g.ts == 2)
        hipLaunchKernelGGL(rans2_decode_kernel<2>, dim3((unsigned)g.nchunks), dim3(64), lds, s, in, in_bytes, offsets,
                           g, vol, status, ring);
    else
        hipLaunchKernelGGL(rans2_decode_kernel<4>, dim3((unsigned)g.nchunks), dim3(64), lds, s, in, in_bytes, offsets,
                           g, vol, status, ring);
    return hipGetLastError();
}

}  // namespace exabm4d
