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

#ifndef EXABM4D_TR_PROBE
#define EXABM4D_TR_PROBE 0                // 1: TIMING PROBE, wrong results -- the [hi][r][lo] writes land 8 elements lower
#endif                                    // for odd hi: conflict-free under ds_write_b64's mod-32 banking
#define TRA_PROBE(hi) (EXABM4D_TR_PROBE ? 8 * ((hi) & 1) : 0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
// (In C++ `(f2)(a, b)` is a cast of a comma expression, not a vector literal.)
__device__ __forceinline__ f2 mk2(float a, float b) {
    f2 r;
    r.x = a;
    r.y = b;
    return r;
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


// ---- half-size transpose buffer ------------------------------------------------------------------------
// One COMPONENT of the pair at a time through an [8][8][8] FLOAT buffer (strides below): the .x
// values are written and read back transposed, then the .y values through the same words -- LDS
// executes a wave's instructions in order, so the second component's writes cannot overtake the first
// one's reads and no wait separates them.  Half the LDS per wave (the hard-threshold kernel trades it
// for ring planes, stage_kernels.hip) for eight more write instructions per transposition.
// Strides in floats: TJ1 = 8 keeps the b128 reads 16-byte aligned and the [r][lo][hi] writes
// conflict-free; TI1 = 76 (4 mod 8) makes the reads of the two hi values of a 16-lane pass fall into
// different banks (the [hi][r][lo] writes are then two-way conflicted: 4 hi + lo).
constexpr int TI1 = 76, TJ1 = 8;
constexpr int TBUF1 = 640;                // floats per wave: 7 * 76 + 7 * 8 + 8 = 596 for the transposes,
                                          // 5 * 64 float2 for the half groups' exchange
// WB: write index r * TI1 + lo * TJ1 + hi (else hi * TI1 + r * TJ1 + lo); read 8 floats at hi * TI1 + lo * TJ1
template <bool WB>
__device__ __forceinline__ void transpose_half(float* tf, int hi, int lo, f2 (&v)[8]) {
    f4 q[2][2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
            tf[WB ? r * TI1 + lo * TJ1 + hi : hi * TI1 + r * TJ1 + lo] = c ? v[r].y : v[r].x;
        cbar();
        const f4* src = reinterpret_cast<const f4*>(tf + hi * TI1 + lo * TJ1);
        q[c][0] = src[0];
        q[c][1] = src[1];
        cbar();
    }
    v[0] = mk2(q[0][0].x, q[1][0].x);
    v[1] = mk2(q[0][0].y, q[1][0].y);
    v[2] = mk2(q[0][0].z, q[1][0].z);
    v[3] = mk2(q[0][0].w, q[1][0].w);
    v[4] = mk2(q[0][1].x, q[1][1].x);
    v[5] = mk2(q[0][1].y, q[1][1].y);
    v[6] = mk2(q[0][1].z, q[1][1].z);
    v[7] = mk2(q[0][1].w, q[1][1].w);
}

// ---- lo <-> register transposition without LDS ----------------------------------------------------
// Within every group of 8 lanes (same hi), lane a / register b holds X[a][b] before and X[b][a]
// after: three butterfly stages over the index bits 4, 2, 1, each swapping X[l][r] with
// X[l ^ bit][r ^ bit] where the lane's and the register's bit differ.  Bit 4 is a DPP row shift
// written under a bank mask (one instruction per register), bits 2 and 1 are quad permutes plus a
// select.  Data movement only: bit-identical to the LDS round trip it replaces.
#ifndef EXABM4D_DPP_TR
#define EXABM4D_DPP_TR 0
#endif
template <int CTRL>
__device__ __forceinline__ float dpp_qp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ void tr_bit4(float& A, float& B) {
    const int a = __float_as_int(A), b = __float_as_int(B);
    // lanes 4-7 of a group (banks 1, 3): A <- B of lane - 4;  lanes 0-3 (banks 0, 2): B <- A of lane + 4
    A = __int_as_float(__builtin_amdgcn_update_dpp(a, b, 0x114 /* row_shr:4 */, 0xF, 0xA, false));
    B = __int_as_float(__builtin_amdgcn_update_dpp(b, a, 0x104 /* row_shl:4 */, 0xF, 0x5, false));
}
template <int CTRL>
__device__ __forceinline__ void tr_quad(float& A, float& B, bool bitset) {
    const float pa = dpp_qp<CTRL>(A), pb = dpp_qp<CTRL>(B);
    A = bitset ? pb : A;
    B = bitset ? B : pa;
}
__device__ __forceinline__ void transpose_lo(f2 (&v)[8], int lo) {
    const bool b2 = (lo & 2) != 0, b1 = (lo & 1) != 0;
    float x[8], y[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        x[r] = v[r].x;
        y[r] = v[r].y;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {              // bit 4: registers r, r + 4
        tr_bit4(x[r], x[r + 4]);
        tr_bit4(y[r], y[r + 4]);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {              // bit 2: registers r, r + 2 with (r & 2) == 0
        const int r = (q & 1) | ((q & 2) << 1);
        tr_quad<0x4E>(x[r], x[r + 2], b2);     // quad_perm [2,3,0,1]
        tr_quad<0x4E>(y[r], y[r + 2], b2);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {              // bit 1: registers 2 q, 2 q + 1
        tr_quad<0xB1>(x[2 * q], x[2 * q + 1], b1);   // quad_perm [1,0,3,2]
        tr_quad<0xB1>(y[2 * q], y[2 * q + 1], b1);
    }
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = mk2(x[r], y[r]);
}

// 3-D DCT of TWO blocks at once (streams .x / .y; the transpose buffer holds float2 elements).
// In: layout L1, out: L3.
template <bool HALF = false, typename TableT>
__device__ __forceinline__ void pair_fwd(const TableT& T, f2* tb, int hi, int lo, f2 (&v)[8]) {
    dct8_fwd2(T, v);                                             // along y
    if constexpr (HALF) {
        float* tf = reinterpret_cast<float*>(tb);
        transpose_half<false>(tf, hi, lo, v);                    // L2: hi = z, lo = y, regs x
        dct8_fwd2(T, v);                                         // along x
        transpose_half<true>(tf, hi, lo, v);                     // L3: hi = x, lo = y, regs z
        dct8_fwd2(T, v);                                         // along z
        return;
    }
#if EXABM4D_DPP_TR
    transpose_lo(v, lo);                                         // L2: hi = z, lo = y, regs x
#else
#pragma unroll
    for (int y = 0; y < 8; y++) tb[hi * TSI + y * TSJ + lo - TRA_PROBE(hi)] = v[y];  // buffer [z][y][x]
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, v);                         // L2: hi = z, lo = y, regs x
    cbar();
#endif
    dct8_fwd2(T, v);                                             // along x
#pragma unroll
    for (int x = 0; x < 8; x++) tb[x * TSI + lo * TSJ + hi] = v[x];  // buffer [x][y][z]
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, v);                         // L3: hi = x, lo = y, regs z
    cbar();
    dct8_fwd2(T, v);                                             // along z
}

// Two block pairs through ONE transpose buffer, interleaved: while pair A's transposed data is on
// its way back from LDS, pair B's arithmetic runs, and vice versa.  LDS executes a wave's
// instructions in issue order, so B's writes (issued after A's reads) cannot overtake them and
// one buffer serves both; the compiler fences only pin the order of the LDS accesses.
template <bool HALF = false, typename TableT>
__device__ __forceinline__ void pair_fwd_x2(const TableT& T, f2* tb, int hi, int lo, f2 (&a)[8], f2 (&b)[8]) {
    if constexpr (HALF) {
        float* tf = reinterpret_cast<float*>(tb);
        dct8_fwd2(T, a);                                         // A along y
        transpose_half<false>(tf, hi, lo, a);                    // (in flight)
        dct8_fwd2(T, b);                                         // B along y
        transpose_half<false>(tf, hi, lo, b);
        dct8_fwd2(T, a);                                         // A along x
        transpose_half<true>(tf, hi, lo, a);
        dct8_fwd2(T, b);                                         // B along x
        transpose_half<true>(tf, hi, lo, b);
        dct8_fwd2(T, a);                                         // A along z
        dct8_fwd2(T, b);                                         // B along z
        return;
    }
    dct8_fwd2(T, a);                                             // A along y
#pragma unroll
    for (int y = 0; y < 8; y++) tb[hi * TSI + y * TSJ + lo - TRA_PROBE(hi)] = a[y];
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, a);                         // A -> L2 (in flight)
    cbar();
    dct8_fwd2(T, b);                                             // B along y
#pragma unroll
    for (int y = 0; y < 8; y++) tb[hi * TSI + y * TSJ + lo - TRA_PROBE(hi)] = b[y];
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, b);                         // B -> L2 (in flight)
    cbar();
    dct8_fwd2(T, a);                                             // A along x
#pragma unroll
    for (int x = 0; x < 8; x++) tb[x * TSI + lo * TSJ + hi] = a[x];
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, a);                         // A -> L3
    cbar();
    dct8_fwd2(T, b);                                             // B along x
#pragma unroll
    for (int x = 0; x < 8; x++) tb[x * TSI + lo * TSJ + hi] = b[x];
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, b);                         // B -> L3
    cbar();
    dct8_fwd2(T, a);                                             // A along z
    dct8_fwd2(T, b);                                             // B along z
}
template <bool HALF = false, typename TableT>
__device__ __forceinline__ void pair_inv_x2(const TableT& T, f2* tb, int hi, int lo, f2 (&a)[8], f2 (&b)[8]) {
    if constexpr (HALF) {
        float* tf = reinterpret_cast<float*>(tb);
        dct8_inv2(T, a);                                         // A along z
        transpose_half<true>(tf, hi, lo, a);
        dct8_inv2(T, b);                                         // B along z
        transpose_half<true>(tf, hi, lo, b);
        dct8_inv2(T, a);                                         // A along x
        transpose_half<false>(tf, hi, lo, a);
        dct8_inv2(T, b);                                         // B along x
        transpose_half<false>(tf, hi, lo, b);
        dct8_inv2(T, a);                                         // A along y
        dct8_inv2(T, b);                                         // B along y
        return;
    }
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
#pragma unroll
    for (int x = 0; x < 8; x++) tb[hi * TSI + x * TSJ + lo - TRA_PROBE(hi)] = a[x];
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, a);
    cbar();
    dct8_inv2(T, b);                                             // B along x
#pragma unroll
    for (int x = 0; x < 8; x++) tb[hi * TSI + x * TSJ + lo - TRA_PROBE(hi)] = b[x];
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, b);
    cbar();
    dct8_inv2(T, a);                                             // A along y
    dct8_inv2(T, b);                                             // B along y
}

// Inverse of pair_fwd: L3 spectra in, spatial blocks in layout L1 out.
template <bool HALF = false, typename TableT>
__device__ __forceinline__ void pair_inv(const TableT& T, f2* tb, int hi, int lo, f2 (&v)[8]) {
    dct8_inv2(T, v);                                             // along z (L3: hi = x, lo = y)
    if constexpr (HALF) {
        float* tf = reinterpret_cast<float*>(tb);
        transpose_half<true>(tf, hi, lo, v);                     // L2: hi = z, lo = y, regs x
        dct8_inv2(T, v);                                         // along x
        transpose_half<false>(tf, hi, lo, v);                    // L1: hi = z, lo = x, regs y
        dct8_inv2(T, v);                                         // along y
        return;
    }
#pragma unroll
    for (int z = 0; z < 8; z++) tb[z * TSI + lo * TSJ + hi] = v[z];  // buffer [z][y][x]
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, v);                         // L2: hi = z, lo = y, regs x
    cbar();
    dct8_inv2(T, v);                                             // along x
#if EXABM4D_DPP_TR
    transpose_lo(v, lo);                                         // L1: hi = z, lo = x, regs y
#else
#pragma unroll
    for (int x = 0; x < 8; x++) tb[hi * TSI + x * TSJ + lo - TRA_PROBE(hi)] = v[x];  // buffer [z][x][y]
    cbar();
    load8p(tb + hi * TSI + lo * TSJ, v);                         // L1: hi = z, lo = x, regs y
    cbar();
#endif
    dct8_inv2(T, v);                                             // along y
}


}  // namespace exabm4d
