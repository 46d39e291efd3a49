// dct_pairs.h -- the 8x8x8 separable DCT of TWO blocks at once as packed-fp32 streams, shared by
// the collaborative-filtering kernels (stage_kernels.hip) and the transform quantiser
// (codec_kernels.hip).  Arithmetic: DESIGN.md 3.5 (even/odd fold, the even half folded once more, fmaf chains, axis order
// y, x, z forward and z, x, y inverse), bit-identical to the oracle.
#pragma once
#include "exabm4d_common.h"

namespace exabm4d {

struct DctTable {
    float d[64];  // [u][n], orthonormal DCT-II, rounded once from double (exabm4d_tables)
};


constexpr int TBUF = 640;                 // per-wave transpose buffer: [8][8][8] float2, strides below
constexpr int TSI = 80, TSJ = 10;         // (found by enumeration) make the b64 writes and b128 reads
                                          // of all four transposes bank-conflict-free but one 2-way write

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
// (In C++ `(f2)(a, b)` is a cast of a comma expression, not a vector literal.)
__device__ __forceinline__ f2 mk2(float a, float b) {
    f2 r;
    r.x = a;
    r.y = b;
    return r;
}

#ifndef EXABM4D_TR_SWAP
#define EXABM4D_TR_SWAP 1                 // 0 (A/B builds): the transposes of rounds 1-3, whose [hi][r][lo] writes conflict 2-way
#endif
// Bank conflicts of the transposes (round 4).  ds_write_b64 is served in groups of 16
// consecutive lanes with bank = dword address mod 32 (MI355X_MICROARCH.md, LDS): a [hi][r][lo] write puts the
// lanes of hi = 2g and 2g + 1 of a group 160 dwords = 0 (mod 32) apart -- every such instruction 2-way
// conflicted (6.5e9 / 9.3e9 conflict cycles per 1024^3 launch of the two stage kernels; rounds 1-3 had
// enumerated the strides for mod-64 banking, which holds for the b128 READS only).  Fix without a single extra
// instruction: odd-hi lanes store their row r at position r ^ 4 -- 4 rows = 80 dwords = 16 (mod 32), the other
// half of the banks -- which is a lane-dependent BASE (two bases: rows 0-3 go 4 rows up, rows 4-7 four down),
// the row offsets stay immediates.  The lane that reads position `lo` back then holds row lo ^ 4 (odd hi): a
// permutation of which lane holds what, undone for free at the next write (whose address is per lane anyway)
// in the forward direction and handed to the caller in the inverse one (tr_x).
// SWAP is a template parameter of the pair transforms: measured on one box at 1024^3 (A/B/A/B), the Wiener kernel
// gains 1.6 ms (235.0 -> 233.4), the hard-threshold kernel LOSES 2.2 (163.0 -> 165.2: its waves wait at the ring
// gate, not for the LDS, and the two more address registers per lane cost it more than the conflicts did) and
// the transform quantiser is HBM-bound -- so only the Wiener kernel instantiates it.
template <bool SWAP>
__device__ __forceinline__ int tr_swap(int hi) { return (SWAP && EXABM4D_TR_SWAP) ? 4 * (hi & 1) : 0; }
// x coordinate held by lane (hi, lo) after pair_inv / pair_inv_x2 (layout L1 up to that permutation)
template <bool SWAP>
__device__ __forceinline__ int tr_x(int hi, int lo) { return lo ^ tr_swap<SWAP>(hi); }
// the [hi][r][lo] store: v[r] -> position r ^ tr_swap(hi) of the lane's (hi, lo) column
template <bool SWAP>
__device__ __forceinline__ void tr_store_a(f2* tb, int hi, int lo, const f2 (&v)[8]) {
    const int s4 = tr_swap<SWAP>(hi);
    f2* up = tb + hi * TSI + lo + s4 * TSJ;      // rows 0..3 land s4 rows further up,
    f2* dn = tb + hi * TSI + lo - s4 * TSJ;      // rows 4..7 as many down
#pragma unroll
    for (int r = 0; r < 4; r++) up[r * TSJ] = v[r];
#pragma unroll
    for (int r = 4; r < 8; r++) dn[r * TSJ] = v[r];
}
// Two independent streams (.x, .y) through one packed-fp32 instruction stream: v_pk_fma_f32 etc.
// do both components with one issue slot, which is what matters at one wave per SIMD.  Each
// component is an ordinary IEEE fp32 operation, so results stay bit-identical to the oracle.
__device__ __forceinline__ f2 chain4p(float c0, f2 v0, float c1, f2 v1, float c2, f2 v2, float c3,
                                      f2 v3) {
    f2 t = v0 * c0;
    t = __builtin_elementwise_fma((f2)(c1), v1, t);
    t = __builtin_elementwise_fma((f2)(c2), v2, t);
    t = __builtin_elementwise_fma((f2)(c3), v3, t);
    return t;
}
// (the even half is folded a second time -- two-term outputs instead of four-term chains)
__device__ __forceinline__ void dct8_fwd2(const DctTable& T, f2 (&v)[8]) {
    f2 s[4], d[4], o[8];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        s[n] = v[n] + v[7 - n];
        d[n] = v[n] - v[7 - n];
    }
    const float c = T.d[0], a = T.d[2 * 8 + 0], b = T.d[2 * 8 + 1];
    const f2 ss0 = s[0] + s[3], ss1 = s[1] + s[2], sd0 = s[0] - s[3], sd1 = s[1] - s[2];
    o[0] = (ss0 + ss1) * c;
    o[4] = (ss0 - ss1) * c;
    o[2] = __builtin_elementwise_fma((f2)(b), sd1, sd0 * a);
    o[6] = __builtin_elementwise_fma((f2)(a), -sd1, sd0 * b);
#pragma unroll
    for (int u = 1; u < 8; u += 2) {
        const float* k = T.d + u * 8;
        o[u] = chain4p(k[0], d[0], k[1], d[1], k[2], d[2], k[3], d[3]);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = o[u];
}
__device__ __forceinline__ void dct8_inv2(const DctTable& T, f2 (&v)[8]) {
    f2 x[8];
    const float c = T.d[0], a = T.d[2 * 8 + 0], b = T.d[2 * 8 + 1];
    const f2 p0 = (v[0] + v[4]) * c, p1 = (v[0] - v[4]) * c;
    const f2 q0 = __builtin_elementwise_fma((f2)(b), v[6], v[2] * a);
    const f2 q1 = __builtin_elementwise_fma((f2)(a), -v[6], v[2] * b);
    const f2 e[4] = {p0 + q0, p1 + q1, p1 - q1, p0 - q0};
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const f2 o = chain4p(T.d[1 * 8 + n], v[1], T.d[3 * 8 + n], v[3], T.d[5 * 8 + n], v[5],
                             T.d[7 * 8 + n], v[7]);
        x[n] = e[n] + o;
        x[7 - n] = e[n] - o;
    }
#pragma unroll
    for (int n = 0; n < 8; n++) v[n] = x[n];
}

// ---- the same chains from the table's 7 distinct magnitudes -----------------------------------------
// D[u][n] = s_u cos((2n + 1) u pi / 16) takes only seven absolute values (bit-identical after the one
// rounding to fp32: checked by make_dct7): c = D[0][*] = |D[4][*]|, a, b = D[2][0], D[2][1],
// e, f, g, h = D[1][0..3].  A kernel that receives the 64-entry table by value keeps it in 64
// scalar registers and spills scalars by the dozen (v_writelane / v_readlane in the inner loops);
// seven fit easily.  Signs move onto the vector operand -- fma(-c, v, t) and fma(c, -v, t) are the
// same IEEE operation, and the negation is a free source modifier -- so every result is
// bit-identical to the table form and to the oracle.
struct Dct7 {
    float c, a, b, e, f, g, h;
};
inline bool make_dct7(const DctTable& T, Dct7& q) {
    q.c = T.d[0];
    q.a = T.d[2 * 8 + 0];
    q.b = T.d[2 * 8 + 1];
    q.e = T.d[1 * 8 + 0];
    q.f = T.d[1 * 8 + 1];
    q.g = T.d[1 * 8 + 2];
    q.h = T.d[1 * 8 + 3];
    const float want[8][4] = {{q.c, q.c, q.c, q.c},     {q.e, q.f, q.g, q.h},   {q.a, q.b, -q.b, -q.a},
                              {q.f, -q.h, -q.e, -q.g},  {q.c, -q.c, -q.c, q.c}, {q.g, -q.e, q.h, q.f},
                              {q.b, -q.a, q.a, -q.b},   {q.h, -q.g, q.f, -q.e}};
    for (int u = 0; u < 8; u++)
        for (int n = 0; n < 4; n++)
            if (T.d[u * 8 + n] != want[u][n]) return false;
    return true;
}
__device__ __forceinline__ void dct8_fwd2(const Dct7& q, f2 (&v)[8]) {
    f2 s[4], d[4], o[8];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        s[n] = v[n] + v[7 - n];
        d[n] = v[n] - v[7 - n];
    }
    const f2 ss0 = s[0] + s[3], ss1 = s[1] + s[2], sd0 = s[0] - s[3], sd1 = s[1] - s[2];
    o[0] = (ss0 + ss1) * q.c;
    o[4] = (ss0 - ss1) * q.c;
    o[2] = __builtin_elementwise_fma((f2)(q.b), sd1, sd0 * q.a);
    o[6] = __builtin_elementwise_fma((f2)(q.a), -sd1, sd0 * q.b);
    o[1] = chain4p(q.e, d[0], q.f, d[1], q.g, d[2], q.h, d[3]);
    o[3] = chain4p(q.f, d[0], q.h, -d[1], q.e, -d[2], q.g, -d[3]);
    o[5] = chain4p(q.g, d[0], q.e, -d[1], q.h, d[2], q.f, d[3]);
    o[7] = chain4p(q.h, d[0], q.g, -d[1], q.f, d[2], q.e, -d[3]);
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = o[u];
}
__device__ __forceinline__ void dct8_inv2(const Dct7& q, f2 (&v)[8]) {
    f2 od[4];
    const f2 p0 = (v[0] + v[4]) * q.c, p1 = (v[0] - v[4]) * q.c;
    const f2 q0 = __builtin_elementwise_fma((f2)(q.b), v[6], v[2] * q.a);
    const f2 q1 = __builtin_elementwise_fma((f2)(q.a), -v[6], v[2] * q.b);
    const f2 ev[4] = {p0 + q0, p1 + q1, p1 - q1, p0 - q0};
    od[0] = chain4p(q.e, v[1], q.f, v[3], q.g, v[5], q.h, v[7]);
    od[1] = chain4p(q.f, v[1], q.h, -v[3], q.e, -v[5], q.g, -v[7]);
    od[2] = chain4p(q.g, v[1], q.e, -v[3], q.h, v[5], q.f, v[7]);
    od[3] = chain4p(q.h, v[1], q.g, -v[3], q.f, v[5], q.e, -v[7]);
#pragma unroll
    for (int n = 0; n < 4; n++) {
        v[n] = ev[n] + od[n];
        v[7 - n] = ev[n] - od[n];
    }
}

// (Round 1 also carried the transforms on the matrix pipe -- v_mfma_f32_4x4x1_16b_f32, whose chained
// accumulation is bit-for-bit the four-term fmaf chain; 3 % / 11 % slower than the packed VALU form.
// That option ended with the four-term even outputs it reproduced: tools/dbg/mfma4x4_probe.hip keeps
// the operand-layout probe.)

// The transpose buffer is private to one wave and LDS executes a wave's instructions in issue
// order, so between its writes and its (cross-lane) reads only the COMPILER must be kept from
// reordering; no s_waitcnt or barrier is needed.
__device__ __forceinline__ void cbar() { asm volatile("" ::: "memory"); }

__device__ __forceinline__ void load8p(const f2* p, f2 (&v)[8]) {
    const f4* q = reinterpret_cast<const f4*>(p);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const f4 t = q[i];
        v[2 * i] = mk2(t.x, t.y);
        v[2 * i + 1] = mk2(t.z, t.w);
    }
}


// (Rounds 1-3 also carried two alternatives that measured slower and were removed in round 4, when the lane
// permutation of tr_store_a made them inconsistent with the callers: half-size transpose buffers -- one pair
// component at a time through a float buffer, more ring planes for more write instructions, DESIGN.md 5.2f --
// and DPP register transposes of the lo index, 80 VALU instructions per block pair)
// In: layout L1, out: L3.
template <bool SWAP = false, typename TableT>
__device__ __forceinline__ void pair_fwd(const TableT& T, f2* tb, int hi, int lo, f2 (&v)[8]) {
    dct8_fwd2(T, v);                                             // along y
    tr_store_a<SWAP>(tb, hi, lo, v);                                   // buffer [z][y][x] (odd z: rows swapped by 4)
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, v);                         // L2: hi = z, lo = y ^ tr_swap(z), regs x
    cbar();
    dct8_fwd2(T, v);                                             // along x
    const int ly = lo ^ tr_swap<SWAP>(hi);                             // the y this lane holds
#pragma unroll
    for (int x = 0; x < 8; x++) tb[x * TSI + ly * TSJ + hi] = v[x];  // buffer [x][y][z], canonical again
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, v);                         // L3: hi = x, lo = y, regs z
    cbar();
    dct8_fwd2(T, v);                                             // along z
}

// Two block pairs through ONE transpose buffer, interleaved: while pair A's transposed data is on
// its way back from LDS, pair B's arithmetic runs, and vice versa.  LDS executes a wave's
// instructions in issue order, so B's writes (issued after A's reads) cannot overtake them and
// one buffer serves both; the compiler fences only pin the order of the LDS accesses.
template <bool SWAP = false, typename TableT>
__device__ __forceinline__ void pair_fwd_x2(const TableT& T, f2* tb, int hi, int lo, f2 (&a)[8], f2 (&b)[8]) {
    const int ly = lo ^ tr_swap<SWAP>(hi);                             // the y this lane holds in L2 (see tr_store_a)
    dct8_fwd2(T, a);                                             // A along y
    tr_store_a<SWAP>(tb, hi, lo, a);
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, a);                         // A -> L2 (in flight)
    cbar();
    dct8_fwd2(T, b);                                             // B along y
    tr_store_a<SWAP>(tb, hi, lo, b);
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, b);                         // B -> L2 (in flight)
    cbar();
    dct8_fwd2(T, a);                                             // A along x
#pragma unroll
    for (int x = 0; x < 8; x++) tb[x * TSI + ly * TSJ + hi] = a[x];
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, a);                         // A -> L3
    cbar();
    dct8_fwd2(T, b);                                             // B along x
#pragma unroll
    for (int x = 0; x < 8; x++) tb[x * TSI + ly * TSJ + hi] = b[x];
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, b);                         // B -> L3
    cbar();
    dct8_fwd2(T, a);                                             // A along z
    dct8_fwd2(T, b);                                             // B along z
}
template <bool SWAP = false, typename TableT>
__device__ __forceinline__ void pair_inv_x2(const TableT& T, f2* tb, int hi, int lo, f2 (&a)[8], f2 (&b)[8]) {
    dct8_inv2(T, a);                                             // A along z
#pragma unroll
    for (int z = 0; z < 8; z++) tb[z * TSI + lo * TSJ + hi] = a[z];
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, a);
    cbar();
    dct8_inv2(T, b);                                             // B along z
#pragma unroll
    for (int z = 0; z < 8; z++) tb[z * TSI + lo * TSJ + hi] = b[z];
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, b);
    cbar();
    dct8_inv2(T, a);                                             // A along x
    tr_store_a<SWAP>(tb, hi, lo, a);                                   // [z][x][y], odd z: x positions swapped by 4
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, a);                         // -> lane (z, lo) holds x = tr_x(z, lo)
    cbar();
    dct8_inv2(T, b);                                             // B along x
    tr_store_a<SWAP>(tb, hi, lo, b);
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, b);
    cbar();
    dct8_inv2(T, a);                                             // A along y
    dct8_inv2(T, b);                                             // B along y
}

// Inverse of pair_fwd: L3 spectra in, spatial blocks out in layout L1 UP TO tr_x: lane (hi = z, lo) holds the
// voxels of column x = tr_x(hi, lo) (callers address their output with that x).
template <bool SWAP = false, typename TableT>
__device__ __forceinline__ void pair_inv(const TableT& T, f2* tb, int hi, int lo, f2 (&v)[8]) {
    dct8_inv2(T, v);                                             // along z (L3: hi = x, lo = y)
#pragma unroll
    for (int z = 0; z < 8; z++) tb[z * TSI + lo * TSJ + hi] = v[z];  // buffer [z][y][x]
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, v);                         // L2: hi = z, lo = y, regs x
    cbar();
    dct8_inv2(T, v);                                             // along x
    tr_store_a<SWAP>(tb, hi, lo, v);                                   // buffer [z][x][y] (odd z: x positions swapped by 4)
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, v);                         // L1 up to tr_x: hi = z, lo -> x = tr_x(z, lo), regs y
    cbar();
    dct8_inv2(T, v);                                             // along y
}


}  // namespace exabm4d
