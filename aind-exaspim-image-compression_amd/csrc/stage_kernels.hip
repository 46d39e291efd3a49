// stage_kernels.hip -- grouped 4-D collaborative filtering + overlap-add aggregation
// (SURVEY.md section 8 rows a-B2 .. a-B5).  Checker: oracle orc_stage.
//
// One 256-lane workgroup per reference block (group).  The K <= 16 matched 8^3 blocks are
// gathered into an LDS tile, transformed by a separable 8-point DCT-II along x, y, z (even/odd
// folded, 4-term fmaf chains -- bit-identical to the oracle) and a Haar transform along the group
// axis, shrunk (hard threshold, or empirical Wiener against the basic estimate's spectrum),
// transformed back and scattered into the num/den accumulators with fp32 atomics.
#include "exabm4d_kernels.h"

namespace exabm4d {

struct DctTable {
    float d[64];  // [u][n], orthonormal DCT-II, rounded once from double (exabm4d_tables)
};

constexpr int ZS = 72;            // LDS stride of a z-plane (64 + 8 pad: conflict-free y pass)
constexpr int PB = 8 * ZS;        // LDS stride of a block
constexpr float HAAR_C = 0.70710678118654752440f;
constexpr int META_FLOATS = 64;  // block corners, K, nnz, partial sums at the end of the LDS region

__device__ __forceinline__ float chain4(float c0, float v0, float c1, float v1, float c2, float v2,
                                        float c3, float v3) {
    float t = c0 * v0;
    t = fmaf(c1, v1, t);
    t = fmaf(c2, v2, t);
    t = fmaf(c3, v3, t);
    return t;
}
__device__ __forceinline__ void dct8_fwd(const DctTable& T, float (&v)[8]) {
    float s[4], d[4], o[8];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        s[n] = v[n] + v[7 - n];
        d[n] = v[n] - v[7 - n];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const float* c = T.d + u * 8;
        o[u] = (u & 1) ? chain4(c[0], d[0], c[1], d[1], c[2], d[2], c[3], d[3])
                       : chain4(c[0], s[0], c[1], s[1], c[2], s[2], c[3], s[3]);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = o[u];
}
__device__ __forceinline__ void dct8_inv(const DctTable& T, float (&v)[8]) {
    float x[8];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        const float e = chain4(T.d[0 * 8 + n], v[0], T.d[2 * 8 + n], v[2], T.d[4 * 8 + n], v[4],
                               T.d[6 * 8 + n], v[6]);
        const float o = chain4(T.d[1 * 8 + n], v[1], T.d[3 * 8 + n], v[3], T.d[5 * 8 + n], v[5],
                               T.d[7 * 8 + n], v[7]);
        x[n] = e + o;
        x[7 - n] = e - o;
    }
#pragma unroll
    for (int n = 0; n < 8; n++) v[n] = x[n];
}

template <int K>
__device__ __forceinline__ void haar_fwd(float (&v)[MAXG]) {
    float t[MAXG];
#pragma unroll
    for (int len = K; len > 1; len >>= 1) {
        const int half = len >> 1;
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
__device__ __forceinline__ void haar_inv(float (&v)[MAXG]) {
    float t[MAXG];
#pragma unroll
    for (int len = 1; len < K; len <<= 1) {
#pragma unroll
        for (int i = 0; i < len; i++) {
            t[2 * i] = (v[i] + v[len + i]) * HAAR_C;
            t[2 * i + 1] = (v[i] - v[len + i]) * HAAR_C;
        }
#pragma unroll
        for (int i = 0; i < 2 * len; i++) v[i] = t[i];
    }
}

// 1-D DCT along LDS stride `stride` for `ncol` columns whose base offsets come from colbase(c).
template <bool INV, typename F>
__device__ __forceinline__ void lds_dct_pass(float* g, const DctTable& T, int ncol, int stride,
                                             F colbase) {
    for (int c = threadIdx.x; c < ncol; c += 256) {
        float* p = g + colbase(c);
        float v[8];
#pragma unroll
        for (int n = 0; n < 8; n++) v[n] = p[n * stride];
        if (INV)
            dct8_inv(T, v);
        else
            dct8_fwd(T, v);
#pragma unroll
        for (int n = 0; n < 8; n++) p[n * stride] = v[n];
    }
}

// Gather the K blocks of `vol` (x-DCT applied on the fly) into the LDS tile.
__device__ __forceinline__ void gather_fwd_x(float* g, const float* __restrict__ vol,
                                             const int* bpos, int K, size_t sy, size_t sz,
                                             const DctTable& T) {
    for (int r = threadIdx.x; r < K * 64; r += 256) {
        const int k = r >> 6, z = (r >> 3) & 7, y = r & 7;
        const float* p = vol + (size_t)(bpos[3 * k] + z) * sz + (size_t)(bpos[3 * k + 1] + y) * sy +
                         bpos[3 * k + 2];
        float v[8];
#pragma unroll
        for (int n = 0; n < 8; n++) v[n] = p[n];
        dct8_fwd(T, v);
        float4* q = reinterpret_cast<float4*>(g + k * PB + z * ZS + y * 8);
        q[0] = make_float4(v[0], v[1], v[2], v[3]);
        q[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}

template <bool WIENER, int K>
__device__ __forceinline__ void shrink_pass(float* g, float* gb, float thr, float sigma2,
                                            int& nnz, float& sw) {
    for (int p = threadIdx.x; p < BVOX; p += 256) {
        const int o = (p >> 6) * ZS + (p & 63);
        float v[MAXG];
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = g[k * PB + o];
        haar_fwd<K>(v);
        if (!WIENER) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                const bool keep = fabsf(v[k]) >= thr;
                nnz += keep ? 1 : 0;
                v[k] = keep ? v[k] : 0.0f;
            }
        } else {
            float b[MAXG];
#pragma unroll
            for (int k = 0; k < K; k++) b[k] = gb[k * PB + o];
            haar_fwd<K>(b);
#pragma unroll
            for (int k = 0; k < K; k++) {
                const float e = b[k] * b[k];
                const float W = e / (e + sigma2);
                v[k] = W * v[k];
                sw += W * W;
            }
        }
        haar_inv<K>(v);
#pragma unroll
        for (int k = 0; k < K; k++) g[k * PB + o] = v[k];
    }
}

template <bool WIENER>
__global__ __launch_bounds__(256) void stage_kernel(const float* __restrict__ noisy_all,
                                                    const float* __restrict__ basic_all,
                                                    const uint32_t* __restrict__ keys_all,
                                                    VolGeom g, DctTable T,
                                                    const float* __restrict__ win, float thr,
                                                    float sigma2, float* __restrict__ num_all,
                                                    float* __restrict__ den_all) {
    // One dynamic LDS region (no static __shared__ in front of it: keeps the base 16-B aligned).
    extern __shared__ __align__(16) float lds[];
    float* gn = lds;                             // noisy group   [16][PB]
    float* gb = lds + (WIENER ? MAXG * PB : 0);  // basic group   [16][PB] (Wiener only)
    float* meta = lds + (WIENER ? 2 : 1) * MAXG * PB;
    int* bpos = reinterpret_cast<int*>(meta);            // [3*16] block corners (z,y,x)
    int& s_K = *reinterpret_cast<int*>(meta + 48);
    int& s_nnz = *reinterpret_cast<int*>(meta + 49);
    float* s_sw = meta + 52;                             // [4] per-wave partial sums

    const size_t voff = (size_t)blockIdx.y * (size_t)g.nvox;
    const float* __restrict__ noisy = noisy_all + voff;
    const float* __restrict__ basic = WIENER ? basic_all + voff : nullptr;
    float* __restrict__ num = num_all + voff;
    float* __restrict__ den = den_all + voff;
    const long long r = blockIdx.x;
    const uint32_t* __restrict__ kk = keys_all + ((size_t)blockIdx.y * (size_t)g.nref + (size_t)r) * MAXG;
    const size_t sy = (size_t)g.nx, sz = (size_t)g.nx * (size_t)g.ny;
    const int tid = threadIdx.x;

    if (tid < MAXG) {
        const int ix = (int)(r % g.gx), iy = (int)((r / g.gx) % g.gy),
                  iz = (int)(r / ((long long)g.gx * g.gy));
        const uint32_t key = kk[tid];
        int dz, dy, dx;
        code_to_disp(key & KEY_CMASK, dz, dy, dx);
        bpos[3 * tid + 0] = grid_pos(iz, g.az, g.nz) + dz;
        bpos[3 * tid + 1] = grid_pos(iy, g.ay, g.ny) + dy;
        bpos[3 * tid + 2] = grid_pos(ix, g.ax, g.nx) + dx;
        const unsigned long long m = __ballot(key != KEY_EMPTY) & 0xFFFFull;
        if (tid == 0) {
            const int count = __popcll(m);
            int K = 1;
            while (K * 2 <= count) K *= 2;
            s_K = K;
            s_nnz = 0;
        }
    }
    __syncthreads();
    const int K = s_K;

    // forward: x (on the fly), y, z
    gather_fwd_x(gn, noisy, bpos, K, sy, sz, T);
    if (WIENER) gather_fwd_x(gb, basic, bpos, K, sy, sz, T);
    __syncthreads();
    auto col_y = [](int c) { return (c >> 6) * PB + ((c >> 3) & 7) * ZS + (c & 7); };
    auto col_z = [](int c) { return (c >> 6) * PB + (c & 63); };
    lds_dct_pass<false>(gn, T, K * 64, 8, col_y);
    if (WIENER) lds_dct_pass<false>(gb, T, K * 64, 8, col_y);
    __syncthreads();
    lds_dct_pass<false>(gn, T, K * 64, ZS, col_z);
    if (WIENER) lds_dct_pass<false>(gb, T, K * 64, ZS, col_z);
    __syncthreads();

    // Haar along the group + shrinkage + inverse Haar
    int nnz = 0;
    float sw = 0.0f;
    switch (K) {
        case 16: shrink_pass<WIENER, 16>(gn, gb, thr, sigma2, nnz, sw); break;
        case 8: shrink_pass<WIENER, 8>(gn, gb, thr, sigma2, nnz, sw); break;
        case 4: shrink_pass<WIENER, 4>(gn, gb, thr, sigma2, nnz, sw); break;
        case 2: shrink_pass<WIENER, 2>(gn, gb, thr, sigma2, nnz, sw); break;
        default: shrink_pass<WIENER, 1>(gn, gb, thr, sigma2, nnz, sw); break;
    }
    if (!WIENER) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) nnz += __shfl_xor(nnz, off);
        if ((tid & 63) == 0) atomicAdd(&s_nnz, nnz);
    } else {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sw += __shfl_xor(sw, off);
        if ((tid & 63) == 0) s_sw[tid >> 6] = sw;
    }
    __syncthreads();
    float w;
    if (!WIENER) {
        const int n = s_nnz;
        w = 1.0f / (sigma2 * (float)(n > 1 ? n : 1));
    } else {
        const float s = (s_sw[0] + s_sw[1]) + (s_sw[2] + s_sw[3]);
        w = 1.0f / (sigma2 * (s > 1.0f ? s : 1.0f));
    }

    // inverse: z, y, then x fused with the scatter
    lds_dct_pass<true>(gn, T, K * 64, ZS, col_z);
    __syncthreads();
    lds_dct_pass<true>(gn, T, K * 64, 8, col_y);
    __syncthreads();
    for (int rr = tid; rr < K * 64; rr += 256) {
        const int k = rr >> 6, z = (rr >> 3) & 7, y = rr & 7;
        const float4* q = reinterpret_cast<const float4*>(gn + k * PB + z * ZS + y * 8);
        const float4 a = q[0], b = q[1];
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        dct8_inv(T, v);
        const size_t o = (size_t)(bpos[3 * k] + z) * sz + (size_t)(bpos[3 * k + 1] + y) * sy +
                        (size_t)bpos[3 * k + 2];
        const float* wr = win + (z * 8 + y) * 8;
#pragma unroll
        for (int x = 0; x < 8; x++) {
            const float ww = w * wr[x];
            atomicAdd(num + o + x, ww * v[x]);
            atomicAdd(den + o + x, ww);
        }
    }
}

hipError_t launch_stage(const float* noisy, const float* basic, const uint32_t* keys,
                        const VolGeom& g, int batch, const float* dct64, const float* win_dev,
                        float thr, float sigma2, float* num, float* den, hipStream_t stream) {
    DctTable T;
    for (int i = 0; i < 64; i++) T.d[i] = dct64[i];
    dim3 grid((unsigned)g.nref, (unsigned)batch);
    if (basic) {
        const size_t lds = sizeof(float) * (2 * MAXG * PB + META_FLOATS);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(stage_kernel<true>, grid, dim3(256), lds, stream, noisy, basic, keys, g,
                           T, win_dev, thr, sigma2, num, den);
    } else {
        const size_t lds = sizeof(float) * (MAXG * PB + META_FLOATS);
        hipLaunchKernelGGL(stage_kernel<false>, grid, dim3(256), lds, stream, noisy, basic, keys, g,
                           T, win_dev, thr, sigma2, num, den);
    }
    return hipGetLastError();
}

}  // namespace exabm4d
