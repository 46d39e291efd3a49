// rans_kernels.hip -- chunk entropy coder of the encode half (BASELINE
// config 5): byte shuffle + static order-0 rANS per byte plane, one 64^3 chunk per workgroup, one
// wave per byte plane, the 64 lanes of the wave being the 64 interleaved rANS states of the EXAC v1
// stream.  Replaces the arithmetic behind the reference's
// `codec.encode(chunk)` loop in compute_cratio (utils/img_util.py:401-441; the reference's codec
// is third-party Blosc-zstd with the SHUFFLE filter, evaluate.py:40).
//
// Integer work only: HBM/LDS-bound, no MFMA.  A row of a chunk (64 elements) is one coalesced
// wave load; the emitted 16-bit words of a row are compacted with a ballot + mbcnt rank.
#include "exabm4d_kernels.h"
#include "rans_common.h"

namespace exabm4d {

namespace {

#ifndef EXABM4D_ENC_NC
#define EXABM4D_ENC_NC 2        // copies of the encoder's histogram counters (power of two; 1: 6.3 + 8.4 ms, 2: 5.0 + 8.4, 4: 5.4 + 11.0 -- LDS per workgroup)
#endif
#ifndef EXABM4D_ENC_PP
#define EXABM4D_ENC_PP 2        // byte planes per wave of the encoder (1: one wave per plane)
#endif
// Normalised frequencies of one plane from its 256 counts (oracle
// orc_exac_normalize).  Lane l owns symbols 64 j + l, j = 0..3.
__device__ __forceinline__ void normalize_plane(const uint32_t (&cnt)[4], uint32_t n, uint32_t lane,
                                                uint32_t (&F)[4]) {
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        F[j] = 0;
        if (cnt[j]) {
            const uint64_t f = ((uint64_t)cnt[j] * RANS_M + n / 2) / n;
            F[j] = f < 1 ? 1u : (uint32_t)f;
        }
        mine += F[j];
    }
    uint32_t sum = wave_sum(mine);
    for (;;) {
        // largest F, lowest symbol on ties: key = F << 8 | (255 - s)
        uint32_t key = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) key = max(key, (F[j] << 8) | (255u - (64u * j + lane)));
        const uint32_t best = wave_max(key);
        const uint32_t bs = 255u - (best & 255u);
        const bool owner = (bs & 63u) == lane;
        if (sum < RANS_M) {
            if (owner) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if ((bs >> 6) == (uint32_t)j) F[j] += RANS_M - sum;
            }
            sum = RANS_M;
        }
        if (sum == RANS_M) break;
        if (owner) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((bs >> 6) == (uint32_t)j) F[j] -= 1;
        }
        sum--;
    }
}

}  // namespace

// ---- encode: one workgroup per chunk, wave p codes byte plane p into the chunk's slot ---------------
// Slot layout (scratch, worst case per chunk): [0, 8 + 4 TS) header, then TS table regions of
// HDR_TABLE bytes, then at g.slot_hdr TS stream regions of g.slot_plane bytes.
// One wave codes PP byte planes of the chunk side by side: the element loads, the row cursor and the
// loop are shared by the planes, and a lane carries PP independent rANS states (two chains in
// flight instead of one).  The streams are those of the one-plane-per-wave form, byte for byte.
template <int TS, int PP>
__global__ __launch_bounds__(64 * TS / PP) void rans_encode_kernel(const void* __restrict__ vol, CodecGeom g,
                                                                  const uint2* __restrict__ rcp_tab,
                                                                  uint8_t* __restrict__ slots,
                                                                  uint32_t* __restrict__ sizes) {
    static_assert(TS % PP == 0, "whole planes per wave");
    // NC interleaved copies of every counter (copy = lane mod NC): the lanes of a wave that meet on
    // one symbol of a skewed plane spread over NC addresses in NC different banks
    constexpr int NC = EXABM4D_ENC_NC;
    __shared__ uint32_t hist[TS][256 * NC];
    __shared__ uint2 etab[TS][256];
    __shared__ uint32_t plane_bytes[TS];
    const int c = blockIdx.x;
    const uint32_t lane = lane_id();
    const int p0 = (threadIdx.x >> 6) * PP;           // first plane of this wave
    const ChunkBox b = chunk_box(g, c);
    const uint32_t n = b.n;
    const uint32_t rows = (n + 63u) >> 6;
    const bool fast = (b.ex & 63) == 0;   // a row of 64 elements lies inside one x-row of the volume
    RowCursor rc;
    rc.rpx = (uint32_t)b.ex >> 6;
    rc.ey = (uint32_t)b.ey;
    uint8_t* slot = slots + (size_t)c * g.slot_bytes;

#pragma unroll
    for (int q = 0; q < PP; q++)
#pragma unroll
        for (int j = 0; j < 4 * NC; j++) hist[p0 + q][64 * j + lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // -- pass 1: byte histograms of this wave's planes ---------------------------------------------
    // (RB1 rows' loads in flight together: one 128-byte row per load is far too little to keep the
    // memory system busy with a load-use round trip per row)
    rc.xr = rc.y = rc.z = 0;
    constexpr int RB1 = 8;
    // A histogram does not care which lane sees which element: when the rows are whole and 4-byte
    // aligned, a lane loads 16 bytes (8 uint16 / 4 int32) and the wave takes RPI rows per load.
    const bool wide = fast && (TS == 4 || (g.nx % 2) == 0) && (((uintptr_t)vol + b.base * TS) % 4) == 0;
    if (wide) {
        constexpr int EPL = 16 / TS, LPR = 64 / EPL, RPI = 64 / LPR;
        const uint32_t rpl = (uint32_t)b.ey * rc.rpx;                    // rows per chunk plane
        for (uint32_t r0 = 0; r0 < rows; r0 += RPI) {
            const uint32_t row = r0 + lane / LPR;
            const bool act = row < rows;
            uint4 w = make_uint4(0u, 0u, 0u, 0u);
            if (act) {
                const uint32_t z = row / rpl, rem = row - z * rpl, y = rem / rc.rpx, xr = rem - y * rc.rpx;
                const size_t off = ((size_t)z * g.ny + y) * g.nx + xr * 64u + (lane % LPR) * EPL;
                w = *reinterpret_cast<const uint4*>(static_cast<const char*>(vol) + (b.base + off) * TS);
            }
            uint32_t bits[EPL];
            if (TS == 2) {
                const uint32_t d[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int k = 0; k < EPL; k++) bits[k] = (k & 1) ? d[k / 2] >> 16 : d[k / 2] & 0xFFFFu;
            } else {
                const int32_t d[4] = {(int32_t)w.x, (int32_t)w.y, (int32_t)w.z, (int32_t)w.w};
#pragma unroll
                for (int k = 0; k < EPL; k++) bits[k % 4] = ((uint32_t)d[k % 4] << 1) ^ (uint32_t)(d[k % 4] >> 31);
            }
            const uint64_t am = __ballot(act);
#pragma unroll
            for (int q = 0; q < PP; q++) {
                const int sh = 8 * (p0 + q);
                const uint32_t s0 = __builtin_amdgcn_readfirstlane((bits[0] >> sh) & 255u);   // lane 0 is active
                bool all = true;
#pragma unroll
                for (int k = 0; k < EPL; k++) all = all && ((bits[k] >> sh) & 255u) == s0;
                if (__ballot(act && all) == am) {      // one symbol in all these rows: a single add
                    if (lane == 0) atomicAdd(&hist[p0 + q][s0 * NC], (uint32_t)__popcll(am) * EPL);
                } else if (act) {
#pragma unroll
                    for (int k = 0; k < EPL; k++)
                        atomicAdd(&hist[p0 + q][((bits[k] >> sh) & 255u) * NC + (lane & (NC - 1))], 1u);
                }
            }
        }
    } else
    for (uint32_t r0 = 0; r0 < rows; r0 += RB1) {
        uint32_t bits[RB1];
        uint64_t am[RB1];
#pragma unroll
        for (int k = 0; k < RB1; k++) {
            const uint32_t i = (r0 + k) * 64u + lane;
            const bool act = r0 + k < rows && i < n;
            bits[k] = 0;
            if (act) {
                const size_t off = fast ? rc.offset(g) + lane : elem_offset(g, b, i);
                bits[k] = load_bits<TS>(vol, b.base + off);
            }
            if (fast && r0 + k < rows) rc.next();
            am[k] = __ballot(act);
        }
#pragma unroll
        for (int k = 0; k < RB1; k++) {
            if (am[k] == 0) continue;              // rows past the end (wave-uniform)
            const bool act = (am[k] >> lane) & 1ull;
#pragma unroll
            for (int q = 0; q < PP; q++) {
                const uint32_t s = (bits[k] >> (8 * (p0 + q))) & 255u;
                const uint32_t s0 = __builtin_amdgcn_readlane(s, (int)__builtin_ctzll(am[k]));
                const uint64_t same = __ballot(act && s == s0);
                if (same == am[k]) {               // one symbol in the whole row: a single add
                    if (lane == (uint32_t)__builtin_ctzll(am[k]))
                        atomicAdd(&hist[p0 + q][s0 * NC], (uint32_t)__popcll(am[k]));
                } else if (act) {
                    atomicAdd(&hist[p0 + q][s * NC + (lane & (NC - 1))], 1u);
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // -- tables ------------------------------------------------------------------------------------
    uint32_t nsym[PP];
#pragma unroll
    for (int q = 0; q < PP; q++) {
        const int p = p0 + q;
        uint32_t cnt[4], F[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            cnt[j] = 0;
#pragma unroll
            for (int c = 0; c < NC; c++) cnt[j] += hist[p][(64 * j + lane) * NC + c];
        }
        normalize_plane(cnt, n, lane, F);

        uint8_t* tab = slot + 8 + 4 * TS + p * HDR_TABLE;
        uint32_t ns = 0, cum = 0;
        uint32_t C[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t pm = __ballot(F[j] != 0u);
            if (lane == 0) reinterpret_cast<uint64_t*>(tab)[j] = pm;
            if (F[j]) reinterpret_cast<uint16_t*>(tab + 32)[ns + rank_below(pm)] = (uint16_t)F[j];
            ns += (uint32_t)__popcll(pm);
            C[j] = cum + wave_excl_scan(F[j], lane);
            cum += wave_sum(F[j]);
        }
        nsym[q] = ns;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint2 e = make_uint2(0u, 0u);
            if (F[j]) {
                const uint2 rs = rcp_tab[F[j]];    // {reciprocal, shift}
                const uint32_t bias = F[j] == 1u ? C[j] + RANS_M - 1u : C[j];
                e.x = F[j] | (bias << 13) | (rs.y << 26);
                e.y = rs.x;
            }
            etab[p][64 * j + lane] = e;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // -- pass 2: rows from the last to the first ----------------------------------------------------------
    uint32_t nwords[PP], x[PP];
    uint16_t* out[PP];
    bool any_coded = false;
#pragma unroll
    for (int q = 0; q < PP; q++) {
        nwords[q] = 0;
        x[q] = RANS_L;
        out[q] = reinterpret_cast<uint16_t*>(slot + g.slot_hdr + (size_t)(p0 + q) * g.slot_plane);
        any_coded = any_coded || nsym[q] > 1;
    }
#ifdef EXABM4D_ENC_SKIP2
    any_coded = false;      // timing probe: histogram + tables only
#endif
    if (any_coded) {
        constexpr int RB = 8;
        if (fast && rows) rc.seek(rows - 1);
        for (uint32_t rb = ((rows + RB - 1) / RB) * RB; rb > 0; rb -= RB) {
            uint32_t bits[RB];
            bool live[RB];
#pragma unroll
            for (int k = 0; k < RB; k++) {
                const uint32_t r = rb - 1 - k;
                const uint32_t i = r * 64u + lane;
                bits[k] = 0;
                live[k] = false;
                if (r < rows) {
                    if (i < n) {
                        const size_t off = fast ? rc.offset(g) + lane : elem_offset(g, b, i);
                        bits[k] = load_bits<TS>(vol, b.base + off);
                        live[k] = true;
                    }
                    if (fast) rc.prev();
                }
            }
#pragma unroll
            for (int q = 0; q < PP; q++) {
                if (nsym[q] <= 1) continue;        // a constant plane has no stream (wave-uniform)
                uint2 e[RB];
#pragma unroll
                for (int k = 0; k < RB; k++) e[k] = etab[p0 + q][(bits[k] >> (8 * (p0 + q))) & 255u];
#pragma unroll
                for (int k = 0; k < RB; k++) {
                    const bool act = live[k];
                    const uint32_t f = e[k].x & 0x1FFFu;
                    const bool emit = act && x[q] >= (f << 19);
                    const uint64_t em = __ballot(emit);
                    if (emit) {
                        out[q][nwords[q] + rank_below(em)] = (uint16_t)(x[q] & 0xFFFFu);
                        x[q] >>= 16;
                    }
                    nwords[q] += (uint32_t)__popcll(em);
                    if (act) {
                        const uint32_t qd = __umulhi(x[q], e[k].y) >> (e[k].x >> 26);
                        x[q] = x[q] + ((e[k].x >> 13) & 0x1FFFu) + qd * (RANS_M - f);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < PP; q++)
            if (nsym[q] > 1) {
                out[q][nwords[q] + 2 * lane] = (uint16_t)(x[q] & 0xFFFFu);       // low word, high word per lane
                out[q][nwords[q] + 2 * lane + 1] = (uint16_t)(x[q] >> 16);
                nwords[q] += 128;
            }
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < PP; q++) {
            reinterpret_cast<uint32_t*>(slot + 8)[p0 + q] = nwords[q];
            plane_bytes[p0 + q] = 32u + 2u * nsym[q] + 2u * nwords[q];
        }
        if (p0 == 0) {
            slot[0] = 'E';
            slot[1] = 'X';
            slot[2] = 1;
            slot[3] = (uint8_t)TS;
            reinterpret_cast<uint32_t*>(slot)[1] = n;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 8u + 4u * TS;
#pragma unroll
        for (int q = 0; q < TS; q++) total += plane_bytes[q];
        sizes[c] = total;
    }
}

// offsets[c] = sum of the 16-byte-aligned sizes of the chunks before c; offsets[nchunks] = container
// bytes; totals = { sum of the exact sizes, container bytes }.  One workgroup.
__global__ __launch_bounds__(1024) void rans_scan_kernel(const uint32_t* __restrict__ sizes, int nchunks,
                                                         unsigned long long* __restrict__ offsets,
                                                         unsigned long long* __restrict__ totals) {
    __shared__ unsigned long long part[1024], exact[1024];
    const int t = threadIdx.x;
    const int per = (nchunks + 1023) / 1024;
    const int c0 = min(t * per, nchunks), c1 = min(c0 + per, nchunks);
    unsigned long long a = 0, e = 0;
    for (int c = c0; c < c1; c++) {
        a += ((unsigned long long)sizes[c] + 15ull) & ~15ull;
        e += sizes[c];
    }
    part[t] = a;
    exact[t] = e;
    __syncthreads();
    if (t == 0) {
        unsigned long long run = 0, ex = 0;
        for (int i = 0; i < 1024; i++) {
            const unsigned long long v = part[i];
            part[i] = run;
            run += v;
            ex += exact[i];
        }
        offsets[nchunks] = run;
        totals[0] = ex;
        totals[1] = run;
    }
    __syncthreads();
    unsigned long long run = part[t];
    for (int c = c0; c < c1; c++) {
        offsets[c] = run;
        run += ((unsigned long long)sizes[c] + 15ull) & ~15ull;
    }
}

// slot -> packed stream at out + offsets[c]; all pieces are whole 16-bit words.
template <int TS>
__global__ __launch_bounds__(256) void rans_pack_kernel(const uint8_t* __restrict__ slots, CodecGeom g,
                                                        const unsigned long long* __restrict__ offsets,
                                                        const uint32_t* __restrict__ sizes,
                                                        uint8_t* __restrict__ out) {
    const int c = blockIdx.x;
    const uint8_t* slot = slots + (size_t)c * g.slot_bytes;
    uint16_t* dst = reinterpret_cast<uint16_t*>(out + offsets[c]);
    const uint32_t* nwords = reinterpret_cast<const uint32_t*>(slot + 8);
    uint32_t pos = 0;
    {   // header
        const uint16_t* src = reinterpret_cast<const uint16_t*>(slot);
        for (uint32_t i = threadIdx.x; i < 4u + 2u * TS; i += 256) dst[i] = src[i];
        pos = 4u + 2u * TS;
    }
    for (int p = 0; p < TS; p++) {   // tables
        const uint8_t* tab = slot + 8 + 4 * TS + p * HDR_TABLE;
        const uint64_t* bm = reinterpret_cast<const uint64_t*>(tab);
        const uint32_t nsym = (uint32_t)(__popcll(bm[0]) + __popcll(bm[1]) + __popcll(bm[2]) + __popcll(bm[3]));
        const uint16_t* src = reinterpret_cast<const uint16_t*>(tab);
        for (uint32_t i = threadIdx.x; i < 16u + nsym; i += 256) dst[pos + i] = src[i];
        pos += 16u + nsym;
    }
    for (int p = 0; p < TS; p++) {   // streams
        const uint16_t* src = reinterpret_cast<const uint16_t*>(slot + g.slot_hdr + (size_t)p * g.slot_plane);
        const uint32_t nw = nwords[p];
        for (uint32_t i = threadIdx.x; i < nw; i += 256) dst[pos + i] = src[i];
        pos += nw;
    }
    // zero the alignment padding so that the container is deterministic
    const uint32_t sz = sizes[c], padded = (sz + 15u) & ~15u;
    for (uint32_t i = sz / 2 + threadIdx.x; i < padded / 2; i += 256) dst[i] = 0;
}

// ---- decode: one workgroup per chunk, wave p decodes byte plane p -----------------------------------------
// status[0] is set to a non-zero code by any chunk whose stream is malformed (that chunk is skipped).
template <int TS>
__global__ __launch_bounds__(64 * TS) void rans_decode_kernel(const uint8_t* __restrict__ in,
                                                             size_t in_bytes,
                                                             const unsigned long long* __restrict__ offsets,
                                                             CodecGeom g, void* __restrict__ vol,
                                                             uint32_t* __restrict__ status) {
    __shared__ uint8_t slot2sym[TS][RANS_M];
    __shared__ uint32_t dtab[TS][256];     // F | C << 16
    const int c = blockIdx.x;
    const uint32_t lane = lane_id();
    const int p = threadIdx.x >> 6;
    const ChunkBox b = chunk_box(g, c);
    const uint32_t n = b.n;
    const uint32_t rows = (n + 63u) >> 6;
    const bool fast = (b.ex & 63) == 0;
    RowCursor rc;
    rc.rpx = (uint32_t)b.ex >> 6;
    rc.ey = (uint32_t)b.ey;
    rc.xr = rc.y = rc.z = 0;
    // a corrupt container must not turn into out-of-bounds reads: offsets ascending, inside the
    // buffer, 2-byte aligned (status bit 8)
    if (offsets[c] > offsets[c + 1] || offsets[c + 1] > in_bytes || (offsets[c] & 1ull)) {
        if (lane == 0) atomicOr(status, 8u);
        return;
    }
    const uint8_t* s0 = in + offsets[c];
    const size_t avail = (size_t)(offsets[c + 1] - offsets[c]);

    // header (uniform)
    bool ok = avail >= 8u + 4u * TS && s0[0] == 'E' && s0[1] == 'X' && s0[2] == 1 && s0[3] == TS &&
              reinterpret_cast<const uint32_t*>(s0)[1] == n;
    uint32_t my_nwords = 0, my_nsym = 0;
    size_t tab_off = 8 + 4 * TS, my_tab = 0, words_before = 0, total_words = 0;
    if (ok) {
#pragma unroll
        for (int q = 0; q < TS; q++) {
            const uint32_t nw = reinterpret_cast<const uint32_t*>(s0 + 8)[q];
            if (tab_off + 32 > avail) {
                ok = false;
                break;
            }
            uint32_t ns = 0;
            for (int k = 0; k < 16; k++) ns += (uint32_t)__popc(reinterpret_cast<const uint16_t*>(s0 + tab_off)[k]);
            if (q == p) {
                my_nwords = nw;
                my_nsym = ns;
                my_tab = tab_off;
            }
            if (q < p) words_before += nw;
            total_words += nw;
            tab_off += 32 + 2 * (size_t)ns;
        }
    }
    if (ok && tab_off + 2 * total_words > avail) ok = false;
    if (ok && my_nsym > 1 && my_nwords < 128) ok = false;
    if (ok && n > 0 && my_nsym == 0) ok = false;
    // every wave of the chunk reaches the same verdict on the shared fields; a wave whose own plane
    // is malformed only flags the chunk
    if (!ok) {
        if (lane == 0) atomicOr(status, 1u);
        return;
    }
    if (n == 0) return;

    // frequencies of this plane: lane l owns symbols 64 j + l
    const uint16_t* bm16 = reinterpret_cast<const uint16_t*>(s0 + my_tab);
    const uint16_t* fl = reinterpret_cast<const uint16_t*>(s0 + my_tab + 32);
    uint32_t F[4], C[4], seen = 0, cum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint64_t pm = (uint64_t)bm16[4 * j] | ((uint64_t)bm16[4 * j + 1] << 16) |
                            ((uint64_t)bm16[4 * j + 2] << 32) | ((uint64_t)bm16[4 * j + 3] << 48);
        F[j] = ((pm >> lane) & 1ull) ? fl[seen + rank_below(pm)] : 0u;
        seen += (uint32_t)__popcll(pm);
        C[j] = cum + wave_excl_scan(F[j], lane);
        cum += wave_sum(F[j]);
    }
    if (cum != RANS_M) {
        if (lane == 0) atomicOr(status, 2u);
        return;
    }

    uint8_t* dst = static_cast<uint8_t*>(vol);
    if (my_nsym == 1) {          // constant plane
        uint32_t only = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) only = max(only, F[j] ? 64u * j + lane : 0u);
        only = wave_max(only);
        for (uint32_t r = 0; r < rows; r++) {
            const uint32_t i = r * 64u + lane;
            if (i < n) {
                const size_t off = fast ? rc.offset(g) + lane : elem_offset(g, b, i);
                dst[(b.base + off) * TS + p] = (uint8_t)only;
            }
            if (fast) rc.next();
        }
        return;
    }

#pragma unroll
    for (int j = 0; j < 4; j++) {
        dtab[p][64 * j + lane] = F[j] | (C[j] << 16);
        // slots [C, C + F) of symbol 64 j + l: filled cooperatively below
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int s = 0; s < 256; s++) {
        const uint32_t e = dtab[p][s];
        const uint32_t f = e & 0xFFFFu, cs = e >> 16;
        for (uint32_t t = lane; t < f; t += 64) slot2sym[p][cs + t] = (uint8_t)s;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const uint16_t* words = reinterpret_cast<const uint16_t*>(s0 + tab_off) + words_before;
    uint32_t cursor = my_nwords - 128;
    uint32_t x = (uint32_t)words[cursor + 2 * lane] | ((uint32_t)words[cursor + 2 * lane + 1] << 16);
    bool bad = false;
    for (uint32_t r = 0; r < rows; r++) {
        const uint32_t i = r * 64u + lane;
        const bool act = i < n;
        bool need = false;
        if (act) {
            const uint32_t slot = x & (RANS_M - 1u);
            const uint32_t s = slot2sym[p][slot];
            const uint32_t e = dtab[p][s];
            x = (e & 0xFFFFu) * (x >> RANS_BITS) + slot - (e >> 16);
            need = x < RANS_L;
            const size_t off = fast ? rc.offset(g) + lane : elem_offset(g, b, i);
            dst[(b.base + off) * TS + p] = (uint8_t)s;
        }
        if (fast) rc.next();
        const uint64_t nm = __ballot(need);
        const uint32_t k = (uint32_t)__popcll(nm);
        if (k > cursor) {
            bad = true;
            break;
        }
        cursor -= k;
        if (need) x = (x << 16) | words[cursor + rank_below(nm)];
    }
    if (bad && lane == 0) atomicOr(status, 4u);
}

// in-place inverse of the int32 -> unsigned map of the 4-byte element kind
__global__ __launch_bounds__(256) void unzigzag_kernel(uint32_t* __restrict__ v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t u = v[i];
        v[i] = (u >> 1) ^ (0u - (u & 1u));
    }
}

// ---- host side ----------------------------------------------------------------------------------------------
size_t codec_chunk_bound(size_t n, int ts) {
    // either format: callers size their buffers before they choose one
    const size_t v1 = 8 + 4 * (size_t)ts + (size_t)ts * HDR_TABLE + (size_t)ts * 2 * (n + 128);
    const size_t v2 = codec2_chunk_bound(n, ts);
    return v1 > v2 ? v1 : v2;
}

int make_codec_geom(int ts, int nz, int ny, int nx, int cz, int cy, int cx, CodecGeom& g, int version) {
    if (ts != 2 && ts != 4) return -1;
    if (version != 1 && version != 2) return -1;
    if (nz < 1 || ny < 1 || nx < 1 || cz < 1 || cy < 1 || cx < 1) return -1;
    cz = cz < nz ? cz : nz;
    cy = cy < ny ? cy : ny;
    cx = cx < nx ? cx : nx;
    const unsigned long long cn = (unsigned long long)cz * cy * cx;
    if (cn > (1ull << 28)) return -1;            // rows * 64 must stay inside 32 bits
    g.ts = ts;
    g.nz = nz; g.ny = ny; g.nx = nx;
    g.cz = cz; g.cy = cy; g.cx = cx;
    g.gz = (nz + cz - 1) / cz;
    g.gy = (ny + cy - 1) / cy;
    g.gx = (nx + cx - 1) / cx;
    const unsigned long long nchunks = (unsigned long long)g.gz * g.gy * g.gx;
    if (nchunks > 0x7FFFFFFFull) return -1;
    g.nchunks = (int)nchunks;
    g.slot_hdr = ((size_t)(8 + 4 * ts + ts * HDR_TABLE) + 15) & ~(size_t)15;
    g.slot_plane = ((size_t)2 * ((size_t)cn + 128) + 15) & ~(size_t)15;
    g.slot_bytes = g.slot_hdr + (size_t)ts * g.slot_plane;
    g.version = version;
    g.chunk_elems = (size_t)cn;
    if (version == 2) {
        g.slot_plane = 0;
        codec2_slot_layout((size_t)cn, ts, g.slot_hdr, g.slot_bytes);
    }
    return 0;
}

size_t codec_volume_bound(const CodecGeom& g) {
    // every chunk at its own worst case, 16-byte aligned
    size_t total = 0;
    for (int bz = 0; bz < g.gz; bz++) {
        const size_t ez = (size_t)(g.cz < g.nz - bz * g.cz ? g.cz : g.nz - bz * g.cz);
        for (int by = 0; by < g.gy; by++) {
            const size_t ey = (size_t)(g.cy < g.ny - by * g.cy ? g.cy : g.ny - by * g.cy);
            const int full = g.nx / g.cx, rem = g.nx - full * g.cx;
            total += (size_t)full * ((codec_chunk_bound(ez * ey * (size_t)g.cx, g.ts) + 15) & ~(size_t)15);
            if (rem) total += (codec_chunk_bound(ez * ey * (size_t)rem, g.ts) + 15) & ~(size_t)15;
        }
    }
    return total;
}

void codec_fill_rcp_table(uint32_t* tab /* [4097][2] */) {
    tab[0] = tab[1] = 0;
    for (uint32_t f = 1; f <= RANS_M; f++) {
        if (f == 1) {
            tab[2] = 0xFFFFFFFFu;    // q = x - 1, bias carries the + M - 1 (ryg's freq = 1 form)
            tab[3] = 0;
            continue;
        }
        uint32_t shift = 0;
        while (f > (1u << shift)) shift++;
        tab[2 * f] = (uint32_t)((((unsigned long long)1 << (shift + 31)) + f - 1) / f);
        tab[2 * f + 1] = shift - 1;
    }
}

hipError_t launch_rans_encode(const void* vol, const CodecGeom& g, const uint32_t* rcp_tab, uint8_t* slots,
                              uint32_t* sizes, unsigned long long* offsets, unsigned long long* totals,
                              uint8_t* out, hipStream_t s) {
    const uint2* rt = reinterpret_cast<const uint2*>(rcp_tab);
    if (g.version == 2) {
        uint8_t* work = slots + (((size_t)g.nchunks * g.slot_bytes + 255) & ~(size_t)255);
        const hipError_t e = launch_rans2_encode(vol, g, rcp_tab, slots, work, sizes, nullptr, nullptr, 0, s);
        if (e != hipSuccess) return e;
    } else if (g.ts == 2)
        hipLaunchKernelGGL((rans_encode_kernel<2, EXABM4D_ENC_PP>), dim3((unsigned)g.nchunks), dim3(128 / EXABM4D_ENC_PP), 0, s, vol, g, rt,
                           slots, sizes);
    else
        hipLaunchKernelGGL((rans_encode_kernel<4, EXABM4D_ENC_PP>), dim3((unsigned)g.nchunks), dim3(256 / EXABM4D_ENC_PP), 0, s, vol, g, rt,
                           slots, sizes);
    hipLaunchKernelGGL(rans_scan_kernel, dim3(1), dim3(1024), 0, s, sizes, g.nchunks, offsets, totals);
    if (out) {
        if (g.version == 2) return launch_rans2_encode(vol, g, rcp_tab, slots, nullptr, sizes, out, offsets, 1, s);
        if (g.ts == 2)
            hipLaunchKernelGGL(rans_pack_kernel<2>, dim3((unsigned)g.nchunks), dim3(256), 0, s, slots, g,
                               offsets, sizes, out);
        else
            hipLaunchKernelGGL(rans_pack_kernel<4>, dim3((unsigned)g.nchunks), dim3(256), 0, s, slots, g,
                               offsets, sizes, out);
    }
    return hipGetLastError();
}

hipError_t launch_rans_decode(const uint8_t* in, size_t in_bytes, const unsigned long long* offsets,
                              const CodecGeom& g, void* vol, uint32_t* status, hipStream_t s) {
    if (g.version == 2) return launch_rans2_decode(in, in_bytes, offsets, g, vol, status, s);
    if (g.ts == 2) {
        hipLaunchKernelGGL(rans_decode_kernel<2>, dim3((unsigned)g.nchunks), dim3(128), 0, s, in, in_bytes,
                           offsets, g, vol, status);
    } else {
        hipLaunchKernelGGL(rans_decode_kernel<4>, dim3((unsigned)g.nchunks), dim3(256), 0, s, in, in_bytes,
                           offsets, g, vol, status);
        const size_t n = (size_t)g.nz * g.ny * g.nx;
        const unsigned blocks = (unsigned)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
        hipLaunchKernelGGL(unzigzag_kernel, dim3(blocks), dim3(256), 0, s, static_cast<uint32_t*>(vol), n);
    }
    return hipGetLastError();
}

}  // namespace exabm4d
