// stage_kernels.hip -- grouped 4-D collaborative filtering + overlap-add aggregation
// (SURVEY.md section 8 rows a-B2 .. a-B5).  Checker: oracle orc_stage_q.
//
// Structure (MI355X-first; there is no reference kernel to follow):
//   * A workgroup owns a tile of reference-grid points in (y,x) and MARCHES along z.  The numerator
//     sums of everything its groups can touch live in an LDS ring of 64-bit INTEGERS (fixed point,
//     DESIGN.md 3.8); a plane leaves the ring exactly once, through 64-bit global integer atomics.
//     Integer sums are associative, so the result does not depend on the order in which waves,
//     workgroups or launches add: the GPU equals the oracle bit for bit and itself run for run.
//   * Two waves process one group (half groups, below).  A half group's spectrum lives in registers;
//     each block goes through gather -> DCT(y) -> LDS transpose -> DCT(x) -> LDS transpose -> DCT(z);
//     the Haar transform along the group and the shrinkage then run in registers.
//   * Lane layouts of one 8^3 block (8 values per lane):
//       L1: lane = (z,x), regs = y    gather / scatter (LDS adds conflict-free: bank = 8z + x)
//       L2: lane = (z,y), regs = x
//       L3: lane = (x,y), regs = z    spectrum layout
// All DCT arithmetic is the even/odd-folded fmaf chain of DESIGN.md 3.5, bit-identical to the oracle.
// (Rounds 1-3 also carried a one-wave-per-group kernel with an fp32 (num, den) ring under a lock and a
// four-waves-per-group Wiener kernel; both measured slower -- DESIGN.md 5.2, 5.2i -- and were removed in
// round 4 when the aggregation became integer.)
#include <algorithm>
#include <type_traits>

#include "exabm4d_kernels.h"
#include "dct_pairs.h"

namespace exabm4d {

typedef float f16v __attribute__((ext_vector_type(16)));

#ifdef EXABM4D_STAMPS
// Diagnostic build only: per-phase cycle sums (s_memtime) accumulated over all waves.
__device__ unsigned long long g_stamps[16];
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define STAMP(var) const unsigned long long var = stamp()
#define STAMP_ADD(i, a, b) st[i] += (b) - (a)
#else
#define STAMP(var)
#define STAMP_ADD(i, a, b)
#endif

constexpr float HAAR_C = 0.70710678118654752440f;

#ifndef EXABM4D_GATHER_AUX
#define EXABM4D_GATHER_AUX 0     // cache-policy bits of the gathers' buffer loads (A/B builds: 1 = sc0, 2 = nt, 3 = both)
#endif
// Gather of one block in layout L1 (hi = z, lo = x, regs y), the lane's eight row offsets
// (hi * sz + y * sy + lo, constant for the whole kernel) precomputed in registers: one uniform base
// per block (the corner) in an SGPR pair and no scalar address arithmetic per row.
// Buffer loads: the volume window of a group (planes rz - 5 ...) behind one 128-bit descriptor,
// the block's corner as the scalar offset, the lane's row as the 32-bit vector offset -- no 64-bit
// address arithmetic at all (hipcc does not form the global_load saddr + voffset variant here and
// spends a v_lshl_add_u64 per row otherwise).  Out-of-range offsets would read 0, not fault.
__device__ __forceinline__ void gather8v(__amdgpu_buffer_rsrc_t rsrc, int corner, const unsigned (&voff)[8],
                                         float (&v)[8]) {
#pragma unroll
    for (int y = 0; y < 8; y++)
        v[y] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff[y], corner, EXABM4D_GATHER_AUX));
}

// Packed Haar along the group axis: v[k] holds two independent sequences in .x / .y.
template <int K>
__device__ __forceinline__ void haar_fwd2(f2 (&v)[MAXG]) {
#pragma unroll
    for (int len = K; len > 1; len >>= 1) {
        const int half = len >> 1;
        f2 t[MAXG];
#pragma unroll
        for (int i = 0; i < half; i++) {
            t[i] = (v[2 * i] + v[2 * i + 1]) * HAAR_C;
            t[half + i] = (v[2 * i] - v[2 * i + 1]) * HAAR_C;
        }
#pragma unroll
        for (int i = 0; i < len; i++) v[i] = t[i];
    }
}
template <int K>
__device__ __forceinline__ void haar_inv2(f2 (&v)[MAXG]) {
#pragma unroll
    for (int len = 1; len < K; len <<= 1) {
        f2 t[MAXG];
#pragma unroll
        for (int i = 0; i < len; i++) {
            t[2 * i] = (v[i] + v[len + i]) * HAAR_C;
            t[2 * i + 1] = (v[i] - v[len + i]) * HAAR_C;
        }
#pragma unroll
        for (int i = 0; i < 2 * len; i++) v[i] = t[i];
    }
}

// Wiener stage: both volumes of a block with ONE load per row from the interleaved (noisy, basic)
// float2 volume -- half the cache lines of two separate gathers (the Wiener kernel spends half of
// its time stalled on L1 misses of its gathers).
typedef unsigned u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gather8v2(__amdgpu_buffer_rsrc_t rsrc, int corner2, const unsigned (&voff)[8],
                                          float (&a)[8], float (&b)[8]) {
#pragma unroll
    for (int y = 0; y < 8; y++) {
        const u2v t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(2u * voff[y]), corner2, EXABM4D_GATHER_AUX);
        a[y] = __uint_as_float(t.x);
        b[y] = __uint_as_float(t.y);
    }
}

struct TileGeom {
    int y0, x0;       // voxel coordinates of ring region element (0,0): first ref position - 5
    int nry, nrx;     // grid points of this tile in y and x (1..4)
};

// =================================================================================================
// Two waves per group ("half groups").
//
// One wave alone issues a VALU instruction every 4 cycles, two waves sharing a SIMD every 2
// (MI355X_MICROARCH.md, per-instruction constants), and a whole group's spectrum is 128 registers:
// one wave per group cannot have two waves per SIMD.  A group is split between the two waves of a
// pair: wave h (0/1) transforms blocks [h K/2, (h+1) K/2), so its spectrum is 64 registers and eight
// waves (two per SIMD; twelve in the hard-threshold kernel) fit.  The Haar transform along the group
// splits exactly: its first log2(K)-1 levels act inside each half, and the last level combines only
// the two halves' approximation coefficients -- 8 values per lane -- which the waves swap through
// their transpose buffers.  Every operation is the one the oracle performs, in the same order.
//
// The ring holds the numerator only.  The denominator of the aggregation is a convolution:
// den(v) = sum over blocks of u_b * win(v - corner_b) = (C (*) win)(v) with C(c) = sum of the
// weights of the blocks whose corner is c, and win separable.  So a block costs ONE global atomic
// (its weight, as a 2^-40 fixed-point integer, onto its corner in C) instead of 512 ring updates, and
// the launcher turns C into den with three 8-tap passes (launch_den_from_corners).
//
// The ring is 64-bit INTEGER (fixed point, unit 2^(E - 43)) and updated by LDS atomics
// without any lock: a term is fl64(est) * fl64(u * win) * 2^(43 - E) + 1.5 * 2^52 in ONE v_fma_f64 --
// the exact product, rounded once, half to even -- whose low bits are the integer; ds_add_u64 returns
// nothing, so a wave fires the 16 atomics of a block pair and moves on.  (Rounds 2-3 had an fp64 ring
// under ds_add_f64: as fast, but its sums, the fp32 global atomics of the flush and the fp32 corner
// weights depended on arrival order, and stage 2's match tables are discontinuous in the last bits of
// the basic estimate: uint16 results moved by up to 4 counts between launches.)
// =================================================================================================
#ifndef EXABM4D_PRIO
#define EXABM4D_PRIO 1                         // 0: no wave priorities (A/B builds)
#endif
#ifndef EXABM4D_X2
#define EXABM4D_X2 1                           // 0: one block pair per transform everywhere (A/B builds)
#endif
#ifndef EXABM4D_X2INV
#define EXABM4D_X2INV 1                        // inverse transforms of four blocks as two interleaved pairs
#endif
// Per-kernel geometry of the two-waves-per-group kernels.  The hard-threshold kernel needs 133
// registers and runs THREE waves per SIMD (12 per workgroup, six pairs); the Wiener kernel holds
// two spectra (245 registers) and stays at two (8 per workgroup).  LDS = ring (NPL planes of
// ROWS x COLS doubles) + one transpose buffer per wave, at most 160 KB.
#ifndef EXABM4D_HT_NW
#define EXABM4D_HT_NW 12
#endif
#ifndef EXABM4D_HT_TY
#define EXABM4D_HT_TY 2
#endif
#ifndef EXABM4D_HT_TX
#define EXABM4D_HT_TX 4
#endif
#ifndef EXABM4D_HT_NPL
#define EXABM4D_HT_NPL 18
#endif
#ifndef EXABM4D_WIE_NW
#define EXABM4D_WIE_NW 8
#endif
#ifndef EXABM4D_WIE_TY
#define EXABM4D_WIE_TY 2
#endif
#ifndef EXABM4D_WIE_TX
#define EXABM4D_WIE_TX 4
#endif
#ifndef EXABM4D_WIE_NPL
#define EXABM4D_WIE_NPL 22
#endif
#ifndef EXABM4D_WIE_PAIR_SAME_SIMD
#define EXABM4D_WIE_PAIR_SAME_SIMD 0           // 1 (A/B builds): the two waves of a Wiener pair share a SIMD (waves w, w + 4)
#endif
// Which waves form a pair.  Default: neighbours (w, w ^ 1) -- consecutive wave ids go to different SIMDs.
// SAME_SIMD (8-wave workgroups only): w and w + 4 land on the same SIMD.
template <bool SAME>
struct PairMap {
    static __device__ __forceinline__ int half(int w) { return SAME ? (w >> 2) & 1 : w & 1; }
    static __device__ __forceinline__ int partner(int w) { return SAME ? w ^ 4 : w ^ 1; }
    static __device__ __forceinline__ int pair(int w) { return SAME ? (w & 3) : w >> 1; }
};
template <int NW_, int TY_, int TX_, int NPL_>
struct HalfGeom {
    static constexpr int TBW = 2 * TBUF;                // floats of LDS per wave buffer
    static constexpr int NW = NW_;                      // waves: pair p = wave >> 1, half h = wave & 1 (the
                                                        // waves of a pair sit on different SIMDs)
    static constexpr int TY = TY_, TX = TX_;            // grid points per tile in y and x
    static constexpr int ROWS = (TY - 1) * STEP + 18;   // region rows / columns: 4 per further grid point
    static constexpr int COLS = (TX - 1) * STEP + 18;   //   + 8 (block) + 2 * 5 (search)
    static constexpr int PS = ((ROWS * COLS + 23) / 32) * 32 + 8;  // plane stride in elements, 8 (mod 32)
    static constexpr int NPL = NPL_;                    // ring planes: 18 live ones + slack for running
                                                        // ahead of the flush (per-block gate below)
    static constexpr size_t LDS_FLOATS = (size_t)2 * NPL * PS + (size_t)NW * TBW + 4 + 2 * NW + 8;
    static_assert(NW % 2 == 0 && NW <= 16 && NPL >= 18 && NPL <= 26, "pairs of waves; at most 3 layers in flight");
    static_assert(LDS_FLOATS * sizeof(float) <= 160 * 1024, "ring + transpose buffers exceed the CU's LDS");
};
template <bool WIENER>
struct HalfCfg : HalfGeom<EXABM4D_HT_NW, EXABM4D_HT_TY, EXABM4D_HT_TX, EXABM4D_HT_NPL> {};
template <>
struct HalfCfg<true> : HalfGeom<EXABM4D_WIE_NW, EXABM4D_WIE_TY, EXABM4D_WIE_TX, EXABM4D_WIE_NPL> {};
constexpr int HNCNT = 8;                      // per-layer report counters
typedef long long ring_t;                     // 64-bit fixed point: ds_add_u64, lock-free ring
constexpr double RINT_MAGIC = 6755399441055744.0;      // 1.5 * 2^52: fl64(x + MAGIC) holds rint(x) in its low bits
__device__ __forceinline__ long long magic_bits(double m) {
    // bits(1.5 * 2^52) = 0x43380000'00000000: only the high word is touched (one v_sub_u32)
    const unsigned long long b = (unsigned long long)__double_as_longlong(m);
    const unsigned hi = (unsigned)(b >> 32) - 0x43380000u;
    return (long long)(((unsigned long long)hi << 32) | (b & 0xFFFFFFFFull));
}
// R(d) of DESIGN.md 3.7 for two values at once: integer-subtraction seed (5 % off), three Newton steps as fused
// multiply-adds (v_pk_fma_f32) -- IEEE operations only, so the oracle produces the same bits (v_rcp_f32 is a
// table the CPU does not have), and about what two quarter-rate v_rcp_f32 cost.  Measured at 1024^3, Wiener
// stage, A/B on one box each: v_rcp_f32 233.1 ms, two Newton steps 232.1, three 239.6; with the chains of two
// coefficient pairs interleaved (EXABM4D_WIE_ILV) 239.6 -> 235.4.  A five-operation form -- one cubic step
// r (1 + t + t^2) and one Newton step, 0.8 ulp -- was built into specification, oracle and kernel and measured
// SLOWER than the six operations of three Newton steps (235.9 against 233.8, A/B/A/B): reverted.
#ifndef EXABM4D_WIE_ILV
#define EXABM4D_WIE_ILV 2                      // Newton chains interleaved per scheduling region (0: no limit)
#endif
#ifndef EXABM4D_WIE_RCP
#define EXABM4D_WIE_RCP 0                      // 1 (timing probe, wrong bits): v_rcp_f32 as in rounds 1-3
#endif
__device__ __forceinline__ f2 rcp_nr2(f2 d) {
#if EXABM4D_WIE_RCP == 1
    return mk2(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y));
#endif
    f2 r = mk2(__uint_as_float(0x7EF311C7u - __float_as_uint(d.x)), __uint_as_float(0x7EF311C7u - __float_as_uint(d.y)));
#pragma unroll
    for (int i = 0; i < 3; i++) r = __builtin_elementwise_fma(__builtin_elementwise_fma(-d, r, (f2)(1.0f)), r, r);
    return r;
}
// W = e * R(e + sigma^2) and this lane's share of sum W^2: bits(fl(1 + W^2)) accumulated as integers --
// each is 0x3F800000 + W^2 in units of 2^-23; the caller takes the 0x3F800000s off again (WSQ_BIAS per call)
__device__ __forceinline__ f2 wiener_w2(f2 e, float sigma2, unsigned& acc) {
    const f2 W = e * rcp_nr2(e + sigma2);
    const f2 q = __builtin_elementwise_fma(W, W, (f2)(1.0f));
    acc += __float_as_uint(q.x);
    acc += __float_as_uint(q.y);
    return W;
}
constexpr unsigned WSQ_BIAS = 2u * 0x3F800000u;

// Wait until *flag >= want (lane 0 spins; LDS serves the CU's instructions in arrival order, so
// data the partner wrote before raising the flag is visible once the flag is).
__device__ __forceinline__ void wait_flag(const int* flag, int want, int lane) {
    cbar();
    if (lane == 0) {
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want)
            __builtin_amdgcn_s_sleep(4);
    }
    cbar();
}
__device__ __forceinline__ void raise_flag(int* flag, int value, int lane) {
    cbar();
    if (lane == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    cbar();
}

// Local part of the hard-threshold shrinkage on a half group of KH blocks: forward Haar over the
// half, threshold and count the detail coefficients, hand back the approximation pairs.
// Spectrum layout as in shrink_ht with the local block index: S[jp][2 kl + c] = coefficient plane
// 2 jp + c of local block kl, so that two planes run through the Haar transform as one packed
// stream.  (Whole-vector loads and stores only: element-wise stores into S at constant offsets
// get merged with neighbouring accesses into wider ones, after which the array can no longer be
// split into registers and ends up in scratch.)
template <int KH>
__device__ __forceinline__ void half_shrink_local(f16v (&S)[4], float thr, int& nnz, f2 (&approx)[4]) {
#pragma unroll
    for (int jp = 0; jp < 4; jp++) {
        f16v A = S[jp];
        f2 x[MAXG];
#pragma unroll
        for (int k = 0; k < KH; k++) x[k] = mk2(A[2 * k], A[2 * k + 1]);
        haar_fwd2<KH>(x);
        approx[jp] = x[0];
#pragma unroll
        for (int k = 1; k < KH; k++) {
            const bool k0 = fabsf(x[k].x) >= thr, k1 = fabsf(x[k].y) >= thr;
            nnz += (k0 ? 1 : 0) + (k1 ? 1 : 0);
            float f0 = k0 ? x[k].x : 0.0f, f1 = k1 ? x[k].y : 0.0f;
            // select now: left to itself hipcc keeps the 64-bit compare masks of a whole half group
            // alive in scalar registers until the inverse transforms and spills 64 of them per group
            asm volatile("" : "+v"(f0), "+v"(f1));
            A[2 * k] = f0;
            A[2 * k + 1] = f1;
        }
        S[jp] = A;
    }
}
// Inverse Haar over the half once the filtered approximation coefficients are known.
template <int KH>
__device__ __forceinline__ void half_unshrink_local(f16v (&S)[4], const f2 (&approx)[4]) {
#pragma unroll
    for (int jp = 0; jp < 4; jp++) {
        f16v A = S[jp];
        f2 x[MAXG];
        x[0] = approx[jp];
#pragma unroll
        for (int k = 1; k < KH; k++) x[k] = mk2(A[2 * k], A[2 * k + 1]);
        haar_inv2<KH>(x);
#pragma unroll
        for (int k = 0; k < KH; k++) {
            A[2 * k] = x[k].x;
            A[2 * k + 1] = x[k].y;
        }
        S[jp] = A;
    }
}

// Numerator ring: move plane z to global memory (64-bit integer atomics: neighbouring tiles and z chunks
// overlap; the sums are exact, so their order is immaterial) and zero it.  One wave.
template <class C>
__device__ __forceinline__ void flush_num_plane(ring_t* ring, long long* __restrict__ num, int z,
                                                const TileGeom& tg, const VolGeom& g, int lane) {
    constexpr int REG = C::COLS;
    ring_t* plane = ring + ((z + 5) % C::NPL) * C::PS;
    // LDS reads in flight per lane: six take a 22 x 30 plane in two rounds (four: three rounds, 171.6 /
    // 231.5 ms against 169.8 / 230.1 -- the waves at the ring gate wait for this flush; eleven, one
    // round, costs the Wiener kernel its registers: 275 ms)
    constexpr int N = C::ROWS * REG, U = 6;
    for (int rem0 = lane; rem0 < N; rem0 += 64 * U) {
        ring_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = (rem0 + 64 * u < N) ? plane[rem0 + 64 * u] : 0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int rem = rem0 + 64 * u;
            if (v[u] != 0) {
                const int ryy = rem / REG, rxx = rem - ryy * REG;
                // a non-zero sum implies a block covered this voxel, so it lies inside the volume
                // (64-bit integer atomic, no return value: global_atomic_add_x2)
                __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(num) +
                                           (((size_t)z * g.ny + (tg.y0 + ryy)) * g.nx + (tg.x0 + rxx)),
                                       (unsigned long long)v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                plane[rem] = 0;
            }
        }
    }
}

// Cooperative flush of a closed layer's planes (hard-threshold kernel).  The closing wave opens the
// flush -- lock[3] = planes done = 0, then lock[2] = (layer + 1) << 16: generation and next plane in
// ONE word -- and takes planes like everybody else; waves waiting at the ring gate take planes too.
// A taker draws with one fetch-add, so the index it gets provably belongs to the generation that
// came with it (a closed word has generation 0: the draw is void; a taker looks before it draws, so
// at most one void draw per wave lands on a closed word -- far from its 16 bits).  The closer waits
// for lock[3] to reach the plane count, closes the word and retires the layer.  Returns whether a
// plane was flushed.
#ifndef EXABM4D_HELP_FLUSH
#define EXABM4D_HELP_FLUSH 1
#endif
template <class C>
__device__ __forceinline__ bool flush_take(int* lock, ring_t* ring, long long* __restrict__ num, const TileGeom& tg,
                                           const VolGeom& g, int izb, int lane) {
    int w = lane == 0 ? __hip_atomic_load(lock + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
    w = __builtin_amdgcn_readfirstlane(w);
    if ((w >> 16) == 0) return false;
    {
        // every plane of the open flush already drawn: do not draw again.  A wave polling here while
        // the last takers are still flushing would otherwise keep incrementing the 16-bit index
        // field (one void draw per poll) and, given a long enough stall, carry into the generation
        // bits; with this check a flush sees at most one void draw per wave.
        const int izp = izb + (w >> 16) - 1;
        if ((w & 0xFFFF) >= grid_pos(izp + 1, g.az, g.nz) - grid_pos(izp, g.az, g.nz)) return false;
    }
    w = lane == 0 ? __hip_atomic_fetch_add(lock + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
    w = __builtin_amdgcn_readfirstlane(w);
    const int gen = w >> 16, i = w & 0xFFFF;
    if (gen == 0) return false;
    const int iz = izb + gen - 1;
    const int z0 = grid_pos(iz, g.az, g.nz), zn = grid_pos(iz + 1, g.az, g.nz);
    if (i >= zn - z0) return false;
    cbar();
    flush_num_plane<C>(ring, num, z0 - RAD + i, tg, g, lane);
    cbar();
    if (lane == 0) __hip_atomic_fetch_add(lock + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    cbar();
    return true;
}

// Wiener counterparts.  The noisy and the basic-estimate groups keep separate spectra, both in the
// hard-threshold layout (S[jp][2 kl + c] = coefficient plane 2 jp + c of local block kl), so that
// (i) blocks (kl, kl + 1) of one volume are the two packed streams of a transform, exactly as in
// the hard-threshold kernel, (ii) the basic spectrum is dead once the weights are known -- a wave
// holds 128 spectrum registers only between the forward transforms and the local shrinkage, 64
// afterwards -- and (iii) the inverse half is the hard-threshold code.  (The first version
// interleaved noisy and basic in one 128-register array that stayed live to the end and spilled
// 54 registers per lane: 280 GB of scratch traffic per 1024^3 launch.)
// Local part: Haar over the half of both spectra, Wiener-filter the detail coefficients
// (W = e R(e + sigma^2) from the basic estimate), hand back the approximation pairs.  The
// filtered details stay in S; `sw` collects the lane's share of sum W^2 as biased integers (wiener_w2).
template <int KH>
__device__ __forceinline__ void wiener_half_local(f16v (&S)[4], const f16v (&SB)[4], float sigma2,
                                                  unsigned& sw, f2 (&approx)[8]) {
#pragma unroll
    for (int jp = 0; jp < 4; jp++) {
        f16v A = S[jp];
        const f16v B = SB[jp];
        f2 x[MAXG], y[MAXG];
#pragma unroll
        for (int k = 0; k < KH; k++) {
            x[k] = mk2(A[2 * k], A[2 * k + 1]);
            y[k] = mk2(B[2 * k], B[2 * k + 1]);
        }
        haar_fwd2<KH>(x);
        haar_fwd2<KH>(y);
        approx[jp] = x[0];
        approx[4 + jp] = y[0];
#pragma unroll
        for (int k = 1; k < KH; k++) {
            const f2 W = wiener_w2(y[k] * y[k], sigma2, sw);
            const f2 f = W * x[k];
            A[2 * k] = f.x;
            A[2 * k + 1] = f.y;
            // EXABM4D_WIE_ILV coefficient pairs at a time: interleaving the Newton chains of ALL k for latency
            // costs the registers the two spectra leave (30 spilled), one at a time exposes the chain
            if (EXABM4D_WIE_ILV && (k % EXABM4D_WIE_ILV) == 0) __builtin_amdgcn_sched_barrier(0);
        }
        S[jp] = A;
    }
}

// One wave, one half of a group.  `sync` = int[2 * HNW]: ready[w], ack[w]; `seq` = exchanges this
// pair has done so far (both waves of a pair count alike).  Returns true for the wave that
// completes the layer.
template <bool WIENER, typename TableT>
__device__ __forceinline__ bool process_half_group(
    const float* __restrict__ noisy, const float* __restrict__ basic, const uint32_t* __restrict__ kk,
    int rz, int ry, int rx, const TileGeom& tg, size_t sy, size_t sz, const TableT& T,
    const float (&win)[8], const float* __restrict__ win_g, float thr, float sigma2, double up, ring_t* ring,
    unsigned long long* __restrict__ cvol, f2* tb,
    f2* partner_tb, int* lock, int* sync, int* cnt, int wave, int& seq, int& seen, int layer, int target,
    int lane, long long g_nvox, long long* __restrict__ num, const VolGeom& g, int izb,
    const f2* __restrict__ pair
#ifdef EXABM4D_STAMPS
    , unsigned long long (&st)[16]
#endif
    ) {
    using C = HalfCfg<WIENER>;
    constexpr int HNW = C::NW, HNPL = C::NPL, HPS = C::PS, REG = C::COLS;
    constexpr bool HELP_FLUSH = !WIENER && EXABM4D_HELP_FLUSH;   // (the Wiener kernel has no registers for it
                                                                 //  and waits 1.5 % of its time at the gate)
    constexpr int NP = WIENER ? 8 : 4;         // f2 values per lane swapped with the partner
    const int hi = lane >> 3, lo = lane & 7;
    const int xl = tr_x<WIENER>(hi, lo);               // the block column this lane holds after the inverse transforms (dct_pairs.h)
    STAMP(t0);
    using PM = PairMap<WIENER && EXABM4D_WIE_PAIR_SAME_SIMD && HalfCfg<WIENER>::NW == 8>;
    const int half = PM::half(wave), partner = PM::partner(wave);
    const uint32_t mykey = lane < MAXG ? kk[lane] : KEY_EMPTY;
    const int count = __popcll(__ballot(mykey != KEY_EMPTY));
    int K = 1;
    while (K * 2 <= count) K *= 2;
    const int KH = K > 1 ? K / 2 : 1;          // blocks of this half
    const int kb = K > 1 ? half * KH : 0;      // first block of this half
    const bool active = K > 1 || half == 0;    // a one-block group is the first wave's alone

    int closer = 0;
    // The body of an active half, as a generic lambda over the half's block count: with KH a
    // compile-time constant every loop over blocks unrolls and every access to the spectrum has a
    // constant register index.  (With a run-time KH each element went through s_set_gpr_idx_on /
    // v_mov / s_set_gpr_idx_off: three instructions per moved float, a fifth of the kernel.)
    auto body = [&](auto KHc) {
        constexpr int KH = decltype(KHc)::value;
        f16v S[4];                             // (noisy) spectrum of my blocks
        f16v SB[WIENER ? 4 : 1];               // Wiener: basic-estimate spectrum, dead after the local step
#pragma unroll
        for (int j = 0; j < 4; j++) S[j] = (f16v)(0.0f);
        if constexpr (WIENER) {
#pragma unroll
            for (int j = 0; j < 4; j++) SB[j] = (f16v)(0.0f);
        }

        int my_dz, my_dy, my_dx;
        code_to_disp(mykey & KEY_CMASK, my_dz, my_dy, my_dx);
        const unsigned long long my_corner =
            lane < MAXG ? (unsigned long long)(rz + my_dz) * sz +
                              (unsigned long long)(ry + my_dy) * sy + (unsigned long long)(rx + my_dx)
                        : 0ull;
        // buffer descriptors of the group's window: planes zb .. (blocks start at rz - 5 at the
        // lowest), byte offsets inside it fit 32 bits (make_geom: 24 planes do)
        const int zb = max(rz - RAD, 0);
        const size_t win_off = (size_t)zb * sz;
        const size_t win_left = (size_t)g_nvox - win_off;
        const int win_bytes = (int)min(win_left * sizeof(float), (size_t)0x7FFFFFFFu * 2u);
        const __amdgpu_buffer_rsrc_t noisy_r = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(noisy + win_off), 0, win_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t basic_r = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>((WIENER ? basic : noisy) + win_off), 0, win_bytes, 0x00020000);
        const int my_rel = lane < MAXG ? (int)(4u * (unsigned)(my_corner - win_off)) : 0;
        auto corner_of = [&](int k) -> int { return __builtin_amdgcn_readlane(my_rel, k); };
        // Wiener with the interleaved volume: 8-byte elements, the same window
        const __amdgpu_buffer_rsrc_t pair_r = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<f2*>((WIENER && pair ? pair : reinterpret_cast<const f2*>(noisy)) + (WIENER && pair ? win_off : 0)), 0,
            (int)min(win_left * sizeof(f2), (size_t)0x7FFFFFFFu * 2u), 0x00020000);
        // Forward transforms of my blocks, two streams per transform: blocks (kl, kl + 1) of one
        // volume (a half of one block pairs the noisy block with the basic one in the Wiener
        // stage, and runs the block twice in the hard-threshold stage).  The next gather is
        // issued as soon as the current values have left a / b (/ c / d).
        float a[8], b[8];
        f2 v2[8];
        unsigned voff[8];
#pragma unroll
        for (int y = 0; y < 8; y++)
            voff[y] = 4u * ((unsigned)hi * (unsigned)sz + (unsigned)y * (unsigned)sy + (unsigned)lo);
        if constexpr (!WIENER && KH >= 4 && EXABM4D_X2) {
            float c[8], d[8];
            f2 w2[8];
            gather8v(noisy_r, corner_of(kb), voff, a);
            gather8v(noisy_r, corner_of(kb + 1), voff, b);
            gather8v(noisy_r, corner_of(kb + 2), voff, c);
            gather8v(noisy_r, corner_of(kb + 3), voff, d);
#pragma unroll
            for (int kl = 0; kl < KH; kl += 4) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    v2[j] = mk2(a[j], b[j]);
                    w2[j] = mk2(c[j], d[j]);
                }
                if (kl + 4 < KH) {
                    gather8v(noisy_r, corner_of(kb + kl + 4), voff, a);
                    gather8v(noisy_r, corner_of(kb + kl + 5), voff, b);
                    gather8v(noisy_r, corner_of(kb + kl + 6), voff, c);
                    gather8v(noisy_r, corner_of(kb + kl + 7), voff, d);
                }
                pair_fwd_x2<WIENER>(T, tb, hi, lo, v2, w2);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    S[j >> 1][2 * kl + (j & 1)] = v2[j].x;
                    S[j >> 1][2 * kl + 2 + (j & 1)] = v2[j].y;
                    S[j >> 1][2 * kl + 4 + (j & 1)] = w2[j].x;
                    S[j >> 1][2 * kl + 6 + (j & 1)] = w2[j].y;
                }
            }
        } else if constexpr (!WIENER) {
            gather8v(noisy_r, corner_of(kb), voff, a);
            gather8v(noisy_r, corner_of(kb + (KH > 1 ? 1 : 0)), voff, b);
#pragma unroll
            for (int kl = 0; kl < KH; kl += 2) {
#pragma unroll
                for (int j = 0; j < 8; j++) v2[j] = mk2(a[j], b[j]);
                if (kl + 2 < KH) {
                    gather8v(noisy_r, corner_of(kb + kl + 2), voff, a);
                    gather8v(noisy_r, corner_of(kb + kl + 3), voff, b);
                }
                pair_fwd<WIENER>(T, tb, hi, lo, v2);
#pragma unroll
                for (int j = 0; j < 8; j++) S[j >> 1][2 * kl + (j & 1)] = v2[j].x;
                if constexpr (KH > 1) {
#pragma unroll
                    for (int j = 0; j < 8; j++) S[j >> 1][2 * kl + 2 + (j & 1)] = v2[j].y;
                }
            }
        } else if constexpr (KH == 1) {
            const size_t c0 = corner_of(kb);
            if (pair) {
                gather8v2(pair_r, 2 * (int)c0, voff, a, b);
            } else {
                gather8v(noisy_r, c0, voff, a);
                gather8v(basic_r, c0, voff, b);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) v2[j] = mk2(a[j], b[j]);
            pair_fwd<WIENER>(T, tb, hi, lo, v2);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                S[j >> 1][j & 1] = v2[j].x;
                SB[j >> 1][j & 1] = v2[j].y;
            }
        } else if (pair) {
            // interleaved volume: block k's noisy and basic values arrive together and are the two
            // packed streams of its transform (same arithmetic per stream as the other pairing)
            gather8v2(pair_r, 2 * corner_of(kb), voff, a, b);
#pragma unroll
            for (int kl = 0; kl < KH; kl++) {
#pragma unroll
                for (int j = 0; j < 8; j++) v2[j] = mk2(a[j], b[j]);
                if (kl + 1 < KH) gather8v2(pair_r, 2 * corner_of(kb + kl + 1), voff, a, b);
                pair_fwd<WIENER>(T, tb, hi, lo, v2);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    S[j >> 1][2 * kl + (j & 1)] = v2[j].x;
                    SB[j >> 1][2 * kl + (j & 1)] = v2[j].y;
                }
            }
        } else {
            gather8v(noisy_r, corner_of(kb), voff, a);
            gather8v(noisy_r, corner_of(kb + 1), voff, b);
#pragma unroll
            for (int kl = 0; kl < KH; kl += 2) {
#pragma unroll
                for (int j = 0; j < 8; j++) v2[j] = mk2(a[j], b[j]);
                gather8v(basic_r, corner_of(kb + kl), voff, a);
                gather8v(basic_r, corner_of(kb + kl + 1), voff, b);
                pair_fwd<WIENER>(T, tb, hi, lo, v2);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    S[j >> 1][2 * kl + (j & 1)] = v2[j].x;
                    S[j >> 1][2 * kl + 2 + (j & 1)] = v2[j].y;
                }
#pragma unroll
                for (int j = 0; j < 8; j++) v2[j] = mk2(a[j], b[j]);
                if (kl + 2 < KH) {
                    gather8v(noisy_r, corner_of(kb + kl + 2), voff, a);
                    gather8v(noisy_r, corner_of(kb + kl + 3), voff, b);
                }
                pair_fwd<WIENER>(T, tb, hi, lo, v2);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    SB[j >> 1][2 * kl + (j & 1)] = v2[j].x;
                    SB[j >> 1][2 * kl + 2 + (j & 1)] = v2[j].y;
                }
            }
        }

        STAMP(t1);
        STAMP_ADD(0, t0, t1);
        // Shrinkage: local levels, swap approximations with the partner, top level, local inverse.
        // `tally` is this lane's share of the aggregation-weight statistic: the count of kept
        // coefficients (hard threshold, as int bits) or the sum of squared Wiener weights.
        int nnz = 0;
        unsigned sw = 0;               // Wiener: sum of bits(fl(1 + W^2)), see wiener_w2
        f2 approx[NP];                 // Wiener: [0, 4) noisy plane pairs, [4, 8) basic plane pairs
        if constexpr (WIENER)
            wiener_half_local<KH>(S, SB, sigma2, sw, approx);
        else
            half_shrink_local<KH>(S, thr, nnz, approx);
        f2 top[4];                     // filtered approximation pairs of my half, per plane pair
        if (K > 1) {
            seq++;
            // my transpose buffer is free (forward transforms done; the partner acknowledged the
            // previous exchange before my last inverse transforms started)
#pragma unroll
            for (int jp = 0; jp < NP; jp++) tb[jp * 64 + lane] = approx[jp];
            tb[NP * 64 + lane] = mk2(__int_as_float(WIENER ? (int)sw : nnz), 0.0f);
            raise_flag(sync + wave, seq, lane);
            STAMP(te0);
            wait_flag(sync + partner, seq, lane);
            STAMP(te1);
            STAMP_ADD(2, te0, te1);
            f2 other[NP];
#pragma unroll
            for (int jp = 0; jp < NP; jp++) other[jp] = partner_tb[jp * 64 + lane];
            const float theirs = partner_tb[NP * 64 + lane].x;
            cbar();
            raise_flag(sync + HNW + wave, seq, lane);          // partner may reuse its buffer
            if constexpr (WIENER)
                sw += __float_as_uint(theirs);
            else
                nnz += __float_as_int(theirs);
#pragma unroll
            for (int jp = 0; jp < 4; jp++) {
                const f2 a0 = half ? other[jp] : approx[jp], a1 = half ? approx[jp] : other[jp];
                f2 t0 = (a0 + a1) * HAAR_C, t1 = (a0 - a1) * HAAR_C;
                if constexpr (WIENER) {
                    const f2 b0 = half ? other[4 + jp] : approx[4 + jp];
                    const f2 b1 = half ? approx[4 + jp] : other[4 + jp];
                    const f2 u0 = (b0 + b1) * HAAR_C, u1 = (b0 - b1) * HAAR_C;
                    const f2 W0 = wiener_w2(u0 * u0, sigma2, sw);
                    const f2 W1 = wiener_w2(u1 * u1, sigma2, sw);
                    t0 = W0 * t0;
                    t1 = W1 * t1;
                } else {
                    const bool p0 = fabsf(t0.x) >= thr, p1 = fabsf(t0.y) >= thr;
                    const bool q0 = fabsf(t1.x) >= thr, q1 = fabsf(t1.y) >= thr;
                    nnz += (p0 ? 1 : 0) + (p1 ? 1 : 0) + (q0 ? 1 : 0) + (q1 ? 1 : 0);
                    t0 = mk2(p0 ? t0.x : 0.0f, p1 ? t0.y : 0.0f);
                    t1 = mk2(q0 ? t1.x : 0.0f, q1 ? t1.y : 0.0f);
                }
                top[jp] = half ? (t0 - t1) * HAAR_C : (t0 + t1) * HAAR_C;
                if constexpr (WIENER) __builtin_amdgcn_sched_barrier(0);      // as in wiener_half_local
            }
        } else {
#pragma unroll
            for (int jp = 0; jp < 4; jp++) {
                if constexpr (WIENER) {
                    const f2 W = wiener_w2(approx[4 + jp] * approx[4 + jp], sigma2, sw);
                    top[jp] = W * approx[jp];
                } else {
                    const bool p0 = fabsf(approx[jp].x) >= thr, p1 = fabsf(approx[jp].y) >= thr;
                    nnz += (p0 ? 1 : 0) + (p1 ? 1 : 0);
                    top[jp] = mk2(p0 ? approx[jp].x : 0.0f, p1 ? approx[jp].y : 0.0f);
                }
            }
        }
        half_unshrink_local<KH>(S, top);
        // group weight u = 1 / max(statistic, 1) (IEEE division).  Both statistics are
        // integers, so neither the split into halves nor the order of the lane reduction matters.
        float w;
        if constexpr (WIENER) {
            // wiener_w2 calls behind `sw`: 4 (KH - 1) per half + 8 at the top level (4 for a one-block group)
            sw -= (K > 1 ? 8u * (unsigned)(KH - 1) + 8u : 4u) * WSQ_BIAS;
            // a lane's share is < 2^31 (128 coefficients of at most 2^23 + 1), the group's < 2^37
            unsigned lo16 = sw & 0xFFFFu, hi16 = sw >> 16;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                lo16 += __shfl_xor(lo16, off);
                hi16 += __shfl_xor(hi16, off);
            }
            const double q = (double)hi16 * 65536.0 + (double)lo16;           // exact
            const float stat = (float)(q * (1.0 / 8388608.0));
            w = 1.0f / (stat > 1.0f ? stat : 1.0f);
        } else {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) nnz += __shfl_xor(nnz, off);
            w = 1.0f / (float)(nnz > 1 ? nnz : 1);
        }
        // fl64(fl32(u * win)) * 2^(43 - E): the second factor of a term's single fused multiply-add
        double ww[8];
        if constexpr (WIENER) {
            // the Wiener kernel has no registers to spare for the window between groups: its 8
            // values per lane are re-read here (2 KB table, L1-resident)
#pragma unroll
            for (int y = 0; y < 8; y++) ww[y] = (double)(w * win_g[(hi * 8 + y) * 8 + xl]) * up;
        } else {
#pragma unroll
            for (int y = 0; y < 8; y++) ww[y] = (double)(w * win[y]) * up;
        }
        // denominator: this half's blocks put their weight, rint(u 2^40), onto their corners (see above)
        if (lane >= kb && lane < kb + KH)
            __hip_atomic_fetch_add(cvol + (size_t)my_corner,
                                   (unsigned long long)magic_bits(__builtin_fma((double)w, 1099511627776.0, RINT_MAGIC)),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        const int my_slot0 = (rz + my_dz + 5 + HNPL) % HNPL;
        const int my_yx = (ry + my_dy - tg.y0) * REG + (rx + my_dx - tg.x0);
        // the partner must have read my approximations before the inverse transposes overwrite them
        STAMP(t2);
        STAMP_ADD(1, t1, t2);
        if (K > 1) wait_flag(sync + HNW + partner, seq, lane);
        // Ring slots are re-used every HNPL planes.  A block whose top plane is z0 + t (t = dz + 7)
        // lands on the slot of plane z0 + t - HNPL, which left the ring when the layer
        // ceil((HNPL - 5 - t) / 4) layers back was closed: the block may be added once lock[1]
        // (layers retired so far, in order) has reached layer + 1 - that number.  The gate is per
        // block, right in front of its adds -- only the blocks that reach highest wait for the
        // latest flush -- and the last value seen of the monotonic counter is kept, so most
        // blocks cost no LDS read at all.
        auto gate = [&](int k) {
            const int t = __builtin_amdgcn_readlane(my_dz, k) + 7;
            const int need = layer + 1 - (HNPL - 5 - t + 3) / 4;
            if (seen < need) {
                STAMP(tg0);
                cbar();
                if constexpr (HELP_FLUSH) {
                    // The layer this block waits for retires when its planes have left the ring: a
                    // waiting wave takes planes of the open flush (protocol: flush_take below).
                    for (;;) {
                        int v = lane == 0 ? __hip_atomic_load(lock + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
                        v = __builtin_amdgcn_readfirstlane(v);
                        if (v >= need) {
                            seen = v;
                            break;
                        }
                        if (!flush_take<C>(lock, ring, num, tg, g, izb, lane)) __builtin_amdgcn_s_sleep(4);
                    }
                } else {
                    int v = 0;
                    if (lane == 0) {
                        while ((v = __hip_atomic_load(lock + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < need)
                            __builtin_amdgcn_s_sleep(8);
                    }
                    seen = __builtin_amdgcn_readfirstlane(v);
                }
                cbar();
                STAMP(tg1);
                STAMP_ADD(3, tg0, tg1);
            }
        };
        STAMP(t3);
        STAMP_ADD(5, t2, t3);
        // ring offsets of a block: first plane slot of the lane's z, (y, x) offset of the lane's x
        auto ring_off = [&](int k) -> int {
            int slot = __builtin_amdgcn_readlane(my_slot0, k) + hi;
            slot -= slot >= HNPL ? HNPL : 0;
            return slot * HPS + __builtin_amdgcn_readlane(my_yx, k) + xl;
        };
        auto ring_add = [&](int off, const f2 (&v)[8], int comp) {
            // lock-free: 64-bit integer LDS atomics, no return value.  term = rint(est * ww), the exact
            // product rounded once (v_cvt_f64_f32, v_fma_f64, v_sub_u32, ds_add_u64 per voxel)
#pragma unroll
            for (int y = 0; y < 8; y++)
                __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(ring + off + y * REG),
                                       (unsigned long long)magic_bits(__builtin_fma((double)(comp ? v[y].y : v[y].x), ww[y], RINT_MAGIC)),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        if constexpr (KH >= 4 && EXABM4D_X2INV) {
            f2 w2[8];
#pragma unroll
            for (int kl = 0; kl < KH; kl += 4) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    v2[j] = mk2(S[j >> 1][2 * kl + (j & 1)], S[j >> 1][2 * kl + 2 + (j & 1)]);
                    w2[j] = mk2(S[j >> 1][2 * kl + 4 + (j & 1)], S[j >> 1][2 * kl + 6 + (j & 1)]);
                }
                pair_inv_x2<WIENER>(T, tb, hi, lo, v2, w2);
                STAMP(tl1);
                gate(kb + kl);
                ring_add(ring_off(kb + kl), v2, 0);
                gate(kb + kl + 1);
                ring_add(ring_off(kb + kl + 1), v2, 1);
                gate(kb + kl + 2);
                ring_add(ring_off(kb + kl + 2), w2, 0);
                gate(kb + kl + 3);
                ring_add(ring_off(kb + kl + 3), w2, 1);
                STAMP(tl2);
                STAMP_ADD(4, tl1, tl2);
            }
        } else {
#pragma unroll
            for (int kl = 0; kl < KH; kl += 2) {
                constexpr bool two = KH > 1;
                const int kl2 = two ? kl + 1 : kl;
#pragma unroll
                for (int j = 0; j < 8; j++)
                    v2[j] = mk2(S[j >> 1][2 * kl + (j & 1)], S[j >> 1][2 * kl2 + (j & 1)]);
                pair_inv<WIENER>(T, tb, hi, lo, v2);
                STAMP(tl1);
                gate(kb + kl);
                ring_add(ring_off(kb + kl), v2, 0);
                if constexpr (two) {
                    gate(kb + kl2);
                    ring_add(ring_off(kb + kl2), v2, 1);
                }
                STAMP(tl2);
                STAMP_ADD(4, tl1, tl2);
            }
        }
    };
    if (active) {
        switch (KH) {
            case 8: body(std::integral_constant<int, 8>{}); break;
            case 4: body(std::integral_constant<int, 4>{}); break;
            case 2: body(std::integral_constant<int, 2>{}); break;
            default: body(std::integral_constant<int, 1>{}); break;
        }
    }
    if (!active) {
        // An idle half (one-block group) still reports, and like everybody else only once the layers
        // whose planes this layer re-uses have been retired: at most 2 + HGATE layers are in flight,
        // the report counters have eight slots.
        if (lane == 0) {
            while (__hip_atomic_load(lock + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < layer - 2)
                __builtin_amdgcn_s_sleep(8);
        }
        cbar();
    }
    // Every wave of the pair reports once per group, into the counter of the group's layer (waves
    // run up to two layers ahead, whose reports must not be mistaken for this layer's).  All lanes
    // take part in the atomic (a lane-0-only fetch in front of wave-wide code has been miscompiled
    // before).
    closer = (__hip_atomic_fetch_add(cnt + (layer & (HNCNT - 1)), lane == 0 ? 1 : 0, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_WORKGROUP) + 1 == target) ? 1 : 0;
    return __builtin_amdgcn_readfirstlane(closer) != 0;
}

typedef Dct7 HalfTable;       // seven scalars instead of a 64-entry table in SGPRs (dct_pairs.h)
template <bool WIENER>
__global__ __launch_bounds__(HalfCfg<WIENER>::NW * 64) void stage_half_kernel(
    const float* __restrict__ noisy_all, const float* __restrict__ basic_all,
    const uint32_t* __restrict__ keys_all, VolGeom g, HalfTable T, const float* __restrict__ win_g,
    float thr, float sigma2, const double* __restrict__ qscale, long long* __restrict__ num_all,
    unsigned long long* __restrict__ cvol_all, int tiles_x, int layers_per_chunk, const f2* __restrict__ pair_all,
    int strip) {
    using C = HalfCfg<WIENER>;
    constexpr int HNW = C::NW, HNPL = C::NPL, HPS = C::PS, HTY = C::TY, HTX = C::TX;
    extern __shared__ __align__(16) float lds[];
    ring_t* ring = reinterpret_cast<ring_t*>(lds);         // [HNPL][HPS] numerator sums (int64 fixed point)
    // readfirstlane: the wave index steers register indexing below and must be provably uniform
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    f2* tb = reinterpret_cast<f2*>(lds + 2 * HNPL * HPS + wave * C::TBW);
    using PM = PairMap<WIENER && EXABM4D_WIE_PAIR_SAME_SIMD && C::NW == 8>;
    f2* partner_tb = reinterpret_cast<f2*>(lds + 2 * HNPL * HPS + PM::partner(wave) * C::TBW);
    int* lock = reinterpret_cast<int*>(lds + 2 * HNPL * HPS + HNW * C::TBW);
    int* sync = lock + 4;                                  // ready[HNW], ack[HNW]
    int* cnt = sync + 2 * HNW;                             // reports per layer (slot = layer & 7)

    const size_t voff = (size_t)blockIdx.z * (size_t)g.nvox;
    const float* __restrict__ noisy = noisy_all + voff;
    const float* __restrict__ basic = WIENER ? basic_all + voff : nullptr;
    const f2* __restrict__ pair = (WIENER && pair_all) ? pair_all + voff : nullptr;
    long long* __restrict__ num = num_all + voff;
    unsigned long long* __restrict__ cvol = cvol_all + voff;
    const double up = qscale[2 * blockIdx.z];              // 2^(43 - E) of this volume
    const uint32_t* __restrict__ keys = keys_all + (size_t)blockIdx.z * (size_t)g.nref * MAXG;
    const size_t sy = (size_t)g.nx, sz = (size_t)g.nx * (size_t)g.ny;

    const int tile = xcd_contiguous(blockIdx.x, gridDim.x);
    int ty, tx;
    if (strip) {        // stage_strip: columns walked in strips of `strip` tile rows, column-major
        const int tiles_y = (int)gridDim.x / tiles_x;
        const int sidx = tile / (strip * tiles_x), r = tile - sidx * (strip * tiles_x);
        const int h = min(strip, tiles_y - sidx * strip);
        tx = r / h;
        ty = sidx * strip + (r - tx * h);
    } else {
        ty = tile / tiles_x;
        tx = tile - ty * tiles_x;
    }
    const int iy0 = HTY * ty, ix0 = HTX * tx;
    TileGeom tg;
    tg.nry = min(HTY, g.gy - iy0);
    tg.nrx = min(HTX, g.gx - ix0);
    tg.y0 = grid_pos(iy0, g.ay, g.ny) - RAD;
    tg.x0 = grid_pos(ix0, g.ax, g.nx) - RAD;
    const int nrefs = tg.nry * tg.nrx;

    const int izb = blockIdx.y * layers_per_chunk;
    const int ize = min(g.gz, izb + layers_per_chunk);

    for (int i = threadIdx.x; i < 2 * HNPL * HPS; i += HNW * 64) lds[i] = 0.0f;      // (all-zero bits: the int64 ring)
    // lock[1] = layers retired (in order); the ring itself needs no lock (fp64 LDS atomics)
    if (threadIdx.x < 4 + 2 * HNW + HNCNT) lock[threadIdx.x] = 0;

    float win[8] = {};
    if constexpr (!WIENER) {
        const int hi = lane >> 3, lo = lane & 7;
#pragma unroll
        for (int y = 0; y < 8; y++) win[y] = win_g[(hi * 8 + y) * 8 + tr_x<WIENER>(hi, lo)];    // the x this lane adds (dct_pairs.h)
    }
    __syncthreads();

    const int pairid = PM::pair(wave);
    int seq = 0;
    int seen = 0;           // last value read of lock[1] (layers retired), see the per-block gate
    const Dct7& tab = T;
#ifdef EXABM4D_STAMPS
    unsigned long long st[16] = {};
    const unsigned long long tk0 = stamp();
#endif
    for (int iz = izb; iz < ize; iz++) {
        const int layer = iz - izb;
        const int z0 = grid_pos(iz, g.az, g.nz);
        // The tile's groups go round the pairs, starting one pair further every layer, so that the
        // extra groups rotate when they do not divide by the pairs -- but only while every pair has
        // a group in every layer: a pair without one would skip ahead, a layer could then be
        // complete before the one below it, and the ring protocol counts on layers completing in
        // order (edge tiles with fewer groups than pairs keep the fixed assignment).
        const int rot = nrefs >= HNW / 2 ? layer % (HNW / 2) : 0;
        for (int r = (pairid + rot) % (HNW / 2); r < nrefs; r += HNW / 2) {
            const int jy = r / tg.nrx, jx = r - jy * tg.nrx;
            const int iy = iy0 + jy, ix = ix0 + jx;
            const int ry = grid_pos(iy, g.ay, g.ny), rx = grid_pos(ix, g.ax, g.nx);
            const uint32_t* kk = keys + ((size_t)((size_t)iz * g.gy + iy) * g.gx + ix) * MAXG;
#if EXABM4D_PRIO
            {
                // A wave that starts a group of the oldest open layer holds up everybody at the
                // gate: it gets issue priority (s_setprio) over a SIMD neighbour that started its
                // group ahead of the retired layers.  Measured -6 % on both stage kernels; raising
                // the priority again once a wave is past the gate gives half of that back, and a
                // purely phase-based rule (forward transforms over inverse + ring adds) is on par.
                int f = __hip_atomic_load(lock + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                f = __builtin_amdgcn_readfirstlane(f);
                if (layer <= f)
                    __builtin_amdgcn_s_setprio(1);
                else
                    __builtin_amdgcn_s_setprio(0);
            }
#endif
            const bool closer = process_half_group<WIENER>(
                noisy, basic, kk, z0, ry, rx, tg, sy, sz, tab, win, win_g, thr, sigma2, up, ring, cvol, tb,
                partner_tb, lock, sync, cnt, wave, seq, seen, layer, 2 * nrefs, lane, g.nvox, num, g, izb, pair
#ifdef EXABM4D_STAMPS
                , st
#endif
            );
            if (closer) {
                // this layer's counter slot is next used eight layers on
                if (lane == 0)
                    __hip_atomic_store(cnt + (layer & (HNCNT - 1)), 0, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                // layers retire in order: the layers below are complete (and flushed) before this
                // one's planes leave the ring
                if (lane == 0) {
                    while (__hip_atomic_load(lock + 1, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_WORKGROUP) != layer)
                        __builtin_amdgcn_s_sleep(2);
                }
                cbar();
                if (iz + 1 < ize) {
                    const int zn = grid_pos(iz + 1, g.az, g.nz);
                    if constexpr (!WIENER && EXABM4D_HELP_FLUSH) {
                        if (lane == 0) {
                            __hip_atomic_store(lock + 3, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_store(lock + 2, (layer + 1) << 16, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        cbar();
                        while (flush_take<C>(lock, ring, num, tg, g, izb, lane)) {
                        }
                        if (lane == 0) {
                            while (__hip_atomic_load(lock + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < zn - z0)
                                __builtin_amdgcn_s_sleep(1);
                            __hip_atomic_store(lock + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    } else {
                        for (int z = z0 - RAD; z < zn - RAD; z++) flush_num_plane<C>(ring, num, z, tg, g, lane);
                    }
                }
                cbar();
                if (lane == 0)
                    __hip_atomic_store(lock + 1, layer + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                cbar();
            }
        }
    }
    __syncthreads();
    if (ize > izb) {
        const int base = grid_pos(ize - 1, g.az, g.nz) - RAD;
        for (int z = base + wave; z < base + HNPL; z += HNW) flush_num_plane<C>(ring, num, z, tg, g, lane);
    }
#ifdef EXABM4D_STAMPS
    st[7] = stamp() - tk0;
    if (lane == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&g_stamps[i + (WIENER ? 8 : 0)], st[i]);
#endif
}

// (noisy, basic) -> float2 volume for the Wiener kernel's gathers
__global__ __launch_bounds__(256) void interleave_pair_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              f2* __restrict__ out, size_t n) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
        if (i + 4 <= n) {
            const f4 x = *reinterpret_cast<const f4*>(a + i), y = *reinterpret_cast<const f4*>(b + i);
            f4 lo, hi;
            lo.x = x.x; lo.y = y.x; lo.z = x.y; lo.w = y.y;
            hi.x = x.z; hi.y = y.z; hi.z = x.w; hi.w = y.w;
            *reinterpret_cast<f4*>(out + i) = lo;
            *reinterpret_cast<f4*>(out + i + 2) = hi;
        } else {
            for (size_t j = i; j < n; j++) out[j] = mk2(a[j], b[j]);
        }
    }
}
// One collaborative-filtering stage (basic == NULL: hard threshold; else Wiener).  Adds the numerator
// terms to `num` (int64, units of 2^(43 - E) as qscale[2 b] says for volume b) and every block's weight
// rint(u 2^40) to its corner voxel in `cw`; the caller zeroes both and turns cw into the denominator
// (launch_den_*).  `pair`: 2 n floats of scratch for the Wiener stage's interleaved (noisy, basic)
// volume, or NULL; pair_ready: the caller has filled it.
hipError_t launch_stage(const float* noisy, const float* basic, const uint32_t* keys,
                        const VolGeom& g, int batch, const float* dct64, const float* win_dev,
                        float thr, float sigma2, const double* qscale, long long* num,
                        unsigned long long* cw, hipStream_t stream, const StageOpts& opt, float* pair,
                        int pair_ready) {
    DctTable T;
    for (int i = 0; i < 64; i++) T.d[i] = dct64[i];
    const size_t n = (size_t)g.nvox * (size_t)batch;
    Dct7 HT;
    if (!make_dct7(T, HT)) return hipErrorInvalidValue;    // the table lost its symmetry
    auto launch = [&](auto wiener_c) -> hipError_t {
        constexpr bool W = decltype(wiener_c)::value;
        using C = HalfCfg<W>;
        const int hty = (g.gy + C::TY - 1) / C::TY, htx = (g.gx + C::TX - 1) / C::TX;
        const long long htiles = (long long)hty * htx * batch;
        // z chunks: at least ~4 work items per CU, and -- while a chunk keeps >= 16 layers, so
        // that its ring start-up and final flush stay small -- about 128 per CU (a 1024^3
        // volume has few tile columns per CU and the last ones leave most of the chip idle;
        // measured flat between 2 and 8 chunks at 1024^3).
        int hchunks = (int)((1024 + htiles - 1) / htiles);
        const int fine = (int)std::min<long long>((32768 + htiles - 1) / htiles, g.gz / 16);
        if (hchunks < fine) hchunks = fine;
        if (opt.chunks > 0) hchunks = opt.chunks;
        if (hchunks < 1) hchunks = 1;
        if (hchunks > g.gz) hchunks = g.gz;
        const int hlpc = (g.gz + hchunks - 1) / hchunks;
        hchunks = (g.gz + hlpc - 1) / hlpc;
        const dim3 hgrid((unsigned)(hty * htx), (unsigned)hchunks, (unsigned)batch);
        const size_t lds = sizeof(float) * C::LDS_FLOATS;
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_half_kernel<W>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return err;
        const f2* pairvol = nullptr;
        if (W && pair && opt.pairvol && (pair_ready || (n % 4) == 0)) {
            f2* pv = reinterpret_cast<f2*>(pair);
            if (!pair_ready)
                hipLaunchKernelGGL(interleave_pair_kernel, dim3(65536), dim3(256), 0, stream, noisy, basic, pv, n);
            pairvol = pv;
        }
        hipLaunchKernelGGL(stage_half_kernel<W>, hgrid, dim3(C::NW * 64), lds, stream, noisy, basic, keys,
                           g, HT, win_dev, thr, sigma2, qscale, num, cw, htx, hlpc, pairvol, opt.strip);
        return hipGetLastError();
    };
    return basic ? launch(std::true_type{}) : launch(std::false_type{});
}

#ifdef EXABM4D_STAMPS
extern "C" void exabm4d_debug_stamps(unsigned long long* out, int reset) {
    unsigned long long z[16] = {};
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(z));
    if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z));
}
#endif

}  // namespace exabm4d
