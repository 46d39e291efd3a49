// elementwise_kernels.hip -- HBM-bound streams of the hot path:
//   normalise (a-B6) + clip (a-C)                  data_handling.py:333
//   u16 -> f32 - offset (a-A)                      data_handling.py:353-354
//   intensity transforms forward / inverse (a-D/E) transforms.py:113-152, :223-285, :332-371, :398-411
//   overlap-tile gather / accumulate / finalise    inference.py:81-116, :153-199
// Checker: oracle/host_oracle.py (pinned by tests/golden/*.npz generated from the reference).
//
// fp32 rounding points follow numpy: every Python-float constant of the reference object is
// rounded to fp32 where numpy rounds it (host side, TfDev below) and every array operation is one
// fp32 operation here (-ffp-contract=off; no fused multiply-add).  arcsinh / sinh are evaluated in
// fp64 and rounded once.
#include <algorithm>

#include "exabm4d_kernels.h"

namespace exabm4d {


__device__ __forceinline__ float tf_forward(const TfDev& t, float x) {
    if (t.wrapped) x = x - t.woff;
    if (t.kind == 0) {
        const float u = (x - t.off) / t.scale;
        const float a = (float)asinh((double)u);
        return a / t.norm;
    } else if (t.kind == 1) {
        float arg = t.gain * (x - t.off);
        arg = arg + t.c38g2;
        arg = arg + t.rn2;
        const float g = t.two_over_gain * sqrtf(fmaxf(arg, 0.0f));
        return g / t.norm;
    } else {
        const float y = (x - t.mn) / t.fden;
        return fminf(fmaxf(y, 0.0f), t.clip);
    }
}
__device__ __forceinline__ float tf_inverse_float(const TfDev& t, float y) {
    float c;
    if (t.kind == 0) {
        const float s = (float)sinh((double)(y * t.norm));
        c = t.off + t.scale * s;
    } else if (t.kind == 1) {
        const float d = fmaxf(y, 0.0f) * t.norm;
        const float h = d * t.gain / 2.0f;
        const float arg = h * h;
        const float u = (arg - t.cinvg2) - t.rn2;
        c = t.off + u / t.gain;
    } else {
        c = y * t.range + t.mn;
    }
    if (t.wrapped) c = c + t.woff;
    return c;
}
__device__ __forceinline__ uint16_t quantise_u16(float c, float maxc) {
    c = fminf(fmaxf(c, 0.0f), maxc);     // np.clip(counts, 0, max_count)
    return (uint16_t)rintf(c);           // np.rint (half-to-even) then astype(uint16)
}

constexpr int EW_THREADS = 256;
constexpr int EW_MAX_BLOCKS = 256 * 8;

// ---- 8 voxels per lane: 16-byte uint16 accesses, 2 x 16-byte fp32 accesses ----------------------
__device__ __forceinline__ void ld8(const uint16_t* p, float (&v)[8]) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        v[2 * i] = (float)(w[i] & 0xFFFFu);
        v[2 * i + 1] = (float)(w[i] >> 16);
    }
}
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void st8(uint16_t* p, const uint16_t (&q)[8]) {
    uint4 r;
    r.x = (unsigned)q[0] | ((unsigned)q[1] << 16);
    r.y = (unsigned)q[2] | ((unsigned)q[3] << 16);
    r.z = (unsigned)q[4] | ((unsigned)q[5] << 16);
    r.w = (unsigned)q[6] | ((unsigned)q[7] << 16);
    *reinterpret_cast<uint4*>(p) = r;
}

// A stream operator provides one(i) for a single voxel and eight(i) for voxels [i, i+8).
template <class Op>
__global__ __launch_bounds__(EW_THREADS) void stream8_kernel(Op op, size_t n8) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < n8; g += stride)
        op.eight(8 * g);
}
template <class Op>
__global__ __launch_bounds__(EW_THREADS) void stream1_kernel(Op op, size_t begin, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = begin + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        op.one(i);
}

struct OpCountsFromU16 {
    const uint16_t* in;
    float* out;
    float offset;
    __device__ void one(size_t i) const { out[i] = (float)in[i] - offset; }
    __device__ void eight(size_t i) const {
        float v[8];
        ld8(in + i, v);
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = v[k] - offset;
        st8(out + i, v);
    }
};
// the same, plus the counts XOR 0x8000 (= v - 32768 as int16) for the integer block matching
struct OpCountsFromU16Both {
    const uint16_t* in;
    float* out;
    uint16_t* out16;
    float offset;
    __device__ void one(size_t i) const {
        out[i] = (float)in[i] - offset;
        out16[i] = (uint16_t)(in[i] ^ 0x8000u);
    }
    __device__ void eight(size_t i) const {
        const uint4 r = *reinterpret_cast<const uint4*>(in + i);
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v[2 * k] = (float)(w[k] & 0xFFFFu) - offset;
            v[2 * k + 1] = (float)(w[k] >> 16) - offset;
        }
        st8(out + i, v);
        *reinterpret_cast<uint4*>(out16 + i) =
            make_uint4(r.x ^ 0x80008000u, r.y ^ 0x80008000u, r.z ^ 0x80008000u, r.w ^ 0x80008000u);
    }
};
struct OpNormalizeU16 {
    const float* num;
    const float* den;
    uint16_t* out;
    float offset;
    __device__ void one(size_t i) const { out[i] = quantise_u16(num[i] / den[i] + offset, 65535.0f); }
    __device__ void eight(size_t i) const {
        float a[8], b[8];
        uint16_t q[8];
        ld8(num + i, a);
        ld8(den + i, b);
#pragma unroll
        for (int k = 0; k < 8; k++) q[k] = quantise_u16(a[k] / b[k] + offset, 65535.0f);
        st8(out + i, q);
    }
};
// what stage 2 of the uint16 pipelines matches on -- the basic estimate as the counts a uint16
// caller would see.  As fp32 (counts - offset: the float kernel's input) or as counts XOR 0x8000 (the integer
// kernel's); the Wiener filter itself keeps the unrounded estimate.
struct OpRoundCountsF32 {
    const float* in;
    float* out;
    float offset;
    __device__ void one(size_t i) const { out[i] = (float)quantise_u16(in[i] + offset, 65535.0f) - offset; }
    __device__ void eight(size_t i) const {
        float v[8];
        ld8(in + i, v);
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = (float)quantise_u16(v[k] + offset, 65535.0f) - offset;
        st8(out + i, v);
    }
};
struct OpRoundCountsU16 {
    const float* in;
    uint16_t* out16;
    float offset;
    __device__ void one(size_t i) const {
        out16[i] = (uint16_t)(quantise_u16(in[i] + offset, 65535.0f) ^ 0x8000u);
    }
    __device__ void eight(size_t i) const {
        float v[8];
        uint16_t q[8];
        ld8(in + i, v);
#pragma unroll
        for (int k = 0; k < 8; k++) q[k] = (uint16_t)(quantise_u16(v[k] + offset, 65535.0f) ^ 0x8000u);
        st8(out16 + i, q);
    }
};
template <typename TIn>
struct OpTfForward {
    TfDev t;
    const TIn* in;
    float* out;
    __device__ void one(size_t i) const { out[i] = tf_forward(t, (float)in[i]); }
    __device__ void eight(size_t i) const {
        float v[8];
        ld8(in + i, v);
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = tf_forward(t, v[k]);
        st8(out + i, v);
    }
};
struct OpTfInverseU16 {
    TfDev t;
    const float* in;
    uint16_t* out;
    __device__ void one(size_t i) const { out[i] = quantise_u16(tf_inverse_float(t, in[i]), t.maxc); }
    __device__ void eight(size_t i) const {
        float v[8];
        uint16_t q[8];
        ld8(in + i, v);
#pragma unroll
        for (int k = 0; k < 8; k++) q[k] = quantise_u16(tf_inverse_float(t, v[k]), t.maxc);
        st8(out + i, q);
    }
};
struct OpTileFinalize {
    TfDev t;
    const float* acc;
    const float* wgt;
    uint16_t* out;
    __device__ uint16_t f(float a, float w) const {
        const float y = a / (w + 1e-8f);     // accum_wgt += 1e-8 ; accum_pred /= accum_wgt
        return quantise_u16(tf_inverse_float(t, y), t.maxc);
    }
    __device__ void one(size_t i) const { out[i] = f(acc[i], wgt[i]); }
    __device__ void eight(size_t i) const {
        float a[8], w[8];
        uint16_t q[8];
        ld8(acc + i, a);
        ld8(wgt + i, w);
#pragma unroll
        for (int k = 0; k < 8; k++) q[k] = f(a[k], w[k]);
        st8(out + i, q);
    }
};

// uint16 input has only 65536 distinct values: for the fp64-math-bound asinh transform the
// forward pass becomes a table lookup.  The table is filled by the same device function, so the
// results are bit-identical to direct evaluation.
__global__ __launch_bounds__(EW_THREADS) void tf_lut_build_kernel(TfDev t, float* __restrict__ lut) {
    const unsigned v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < 65536u) lut[v] = tf_forward(t, (float)v);
}
struct OpLutForward {
    const float* lut;
    const uint16_t* in;
    float* out;
    __device__ void one(size_t i) const { out[i] = lut[in[i]]; }
    __device__ void eight(size_t i) const {
        const uint4 r = *reinterpret_cast<const uint4*>(in + i);
        const unsigned w[4] = {r.x, r.y, r.z, r.w};
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v[2 * k] = lut[w[k] & 0xFFFFu];
            v[2 * k + 1] = lut[w[k] >> 16];
        }
        st8(out + i, v);
    }
};

static inline unsigned ew_blocks(size_t n);
static inline bool aligned16(const void* a, const void* b, const void* c = nullptr) {
    return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15u) == 0;
}
// Launch `op` over n voxels: 8 per lane where every pointer is 16-byte aligned, one per lane for
// the tail (or for everything when a caller hands in an unaligned view).
template <class Op>
static hipError_t launch_stream(const Op& op, size_t n, bool vec_ok, hipStream_t s) {
    const size_t n8 = vec_ok ? n / 8 : 0;
    if (n8) hipLaunchKernelGGL(stream8_kernel<Op>, dim3(ew_blocks(n8)), dim3(EW_THREADS), 0, s, op, n8);
    if (8 * n8 < n)
        hipLaunchKernelGGL(stream1_kernel<Op>, dim3(ew_blocks(n - 8 * n8)), dim3(EW_THREADS), 0, s, op,
                           8 * n8, n);
    return hipGetLastError();
}

__global__ __launch_bounds__(EW_THREADS) void normalize_kernel(const float* __restrict__ num,
                                                               const float* __restrict__ den,
                                                               float* __restrict__ out, size_t n,
                                                               float lo, float hi, int do_clip) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = num[i] / den[i];
        if (do_clip) v = fminf(fmaxf(v, lo), hi);
        out[i] = v;
    }
}




__global__ __launch_bounds__(EW_THREADS) void tf_inverse_float_kernel(TfDev t,
                                                                      const float* __restrict__ in,
                                                                      float* __restrict__ out,
                                                                      size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = tf_inverse_float(t, in[i]);
}

// ---- overlap tiling ------------------------------------------------------------------------------
struct PatchStarts {
    int v[64 * 3];  // (z,y,x) of up to 64 patches, by value in the kernarg segment
};

__global__ __launch_bounds__(EW_THREADS) void tile_gather_kernel(const float* __restrict__ vol,
                                                                 int nz, int ny, int nx,
                                                                 PatchStarts st, int patch,
                                                                 float* __restrict__ out) {
    const int b = blockIdx.y;
    const int sz0 = st.v[3 * b], sy0 = st.v[3 * b + 1], sx0 = st.v[3 * b + 2];
    const size_t pv = (size_t)patch * patch * patch;
    float* __restrict__ o = out + (size_t)b * pv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pv;
         i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % patch), y = (int)((i / patch) % patch),
                  z = (int)(i / ((size_t)patch * patch));
        const int gz = sz0 + z, gy = sy0 + y, gx = sx0 + x;
        float v = 0.0f;  // add_padding: zeros beyond the volume
        if (gz < nz && gy < ny && gx < nx) v = vol[((size_t)gz * ny + gy) * nx + gx];
        o[i] = v;
    }
}

// One patch per launch: patches of one batch overlap, and the reference adds them one after the
// other in patch order (inference.py:89-103).  Stream order reproduces exactly that order, so the
// sums are bit-identical to the reference's and no atomics are needed.
__global__ __launch_bounds__(EW_THREADS) void tile_accumulate_kernel(const float* __restrict__ pred,
                                                                     int sz0, int sy0, int sx0,
                                                                     int patch, int trim,
                                                                     float* __restrict__ acc,
                                                                     float* __restrict__ wgt,
                                                                     int nz, int ny, int nx) {
    const int core = patch - 2 * trim;
    const size_t cv = (size_t)core * core * core;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cv;
         i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % core), y = (int)((i / core) % core),
                  z = (int)(i / ((size_t)core * core));
        const int gz = sz0 + trim + z, gy = sy0 + trim + y, gx = sx0 + trim + x;
        if (gz < nz && gy < ny && gx < nx) {
            const size_t o = ((size_t)gz * ny + gy) * nx + gx;
            const float p = pred[((size_t)(z + trim) * patch + (y + trim)) * patch + (x + trim)];
            acc[o] = acc[o] + p;
            wgt[o] = wgt[o] + 1.0f;
        }
    }
}


// ---- chunked byte-plane histograms (row f-1) -----------------------------------------------------
// One workgroup per chunk; LDS integer atomics (full rate, unlike ds_add_f32) on 2 x 256 bins.
__global__ __launch_bounds__(EW_THREADS) void chunk_hist_kernel(const uint16_t* __restrict__ vol,
                                                                int nz, int ny, int nx, int cz,
                                                                int cy, int cx, int gcy, int gcx,
                                                                uint32_t* __restrict__ hist) {
    __shared__ unsigned int h[512];
    for (int i = threadIdx.x; i < 512; i += EW_THREADS) h[i] = 0u;
    __syncthreads();
    const int c = blockIdx.x;
    const int bx = c % gcx, by = (c / gcx) % gcy, bz = c / (gcx * gcy);
    const int z0 = bz * cz, y0 = by * cy, x0 = bx * cx;
    const int ez = min(cz, nz - z0), ey = min(cy, ny - y0), ex = min(cx, nx - x0);
    const size_t total = (size_t)ez * ey * ex;
    for (size_t i = threadIdx.x; i < total; i += EW_THREADS) {
        const int x = (int)(i % ex), y = (int)((i / ex) % ey), z = (int)(i / ((size_t)ex * ey));
        const unsigned v = vol[((size_t)(z0 + z) * ny + (y0 + y)) * nx + (x0 + x)];
        atomicAdd(&h[v & 255u], 1u);
        atomicAdd(&h[256u + (v >> 8)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += EW_THREADS) hist[(size_t)c * 512 + i] = h[i];
}

hipError_t launch_chunk_hist(const uint16_t* vol, int nz, int ny, int nx, int cz, int cy, int cx,
                             uint32_t* hist, hipStream_t s) {
    const int gz = (nz + cz - 1) / cz, gy = (ny + cy - 1) / cy, gx = (nx + cx - 1) / cx;
    hipLaunchKernelGGL(chunk_hist_kernel, dim3((unsigned)(gz * gy * gx)), dim3(EW_THREADS), 0, s,
                       vol, nz, ny, nx, cz, cy, cx, gy, gx, hist);
    return hipGetLastError();
}

// ---- chunk-local mode (BASELINE config 4) ------------------------------
// A batch of padded chunks: chunk (bz, by, bx) of a sub-grid has its core at (z0 + bz*cz, ...),
// extent (ez, ey, ex), and is read with (lz, ly, lx) voxels in front and (pz - ez - lz, ...) behind
// it -- the halo, cut off where the buffer ends.  gather: u16 -> (float) - offset into [batch][pz][py][px];
// scatter: the denoised padded chunks' cores -> + offset -> clip -> rint -> u16 into the output.
__global__ __launch_bounds__(EW_THREADS) void chunk_gather_kernel(const uint16_t* __restrict__ in,
                                                                  ChunkBatch cb, float offset,
                                                                  float* __restrict__ out,
                                                                  uint16_t* __restrict__ out16) {
    const size_t pvox = (size_t)cb.pz * cb.py * cb.px;
    const size_t total = pvox * (size_t)cb.count;
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total;
         i += (size_t)gridDim.x * EW_THREADS) {
        const int b = (int)(i / pvox);
        const size_t r = i - (size_t)b * pvox;
        const int x = (int)(r % cb.px), y = (int)((r / cb.px) % cb.py), z = (int)(r / ((size_t)cb.px * cb.py));
        const int c = cb.first + b;
        const int bx = c % cb.sgx, by = (c / cb.sgx) % cb.sgy, bz = c / (cb.sgx * cb.sgy);
        const int gz = min(max(cb.z0 + bz * cb.cz - cb.lz + z, 0), cb.nz - 1);
        const int gy = min(max(cb.y0 + by * cb.cy - cb.ly + y, 0), cb.ny - 1);
        const int gx = min(max(cb.x0 + bx * cb.cx - cb.lx + x, 0), cb.nx - 1);
        const unsigned v = in[((size_t)gz * cb.ny + gy) * cb.nx + gx];
        out[i] = (float)v - offset;
        if (out16) out16[i] = (uint16_t)(v ^ 0x8000u);
    }
}
__global__ __launch_bounds__(EW_THREADS) void chunk_scatter_kernel(const float* __restrict__ est,
                                                                   ChunkBatch cb, float offset,
                                                                   uint16_t* __restrict__ out) {
    const size_t cvox = (size_t)cb.ez * cb.ey * cb.ex;
    const size_t total = cvox * (size_t)cb.count;
    const size_t pvox = (size_t)cb.pz * cb.py * cb.px;
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total;
         i += (size_t)gridDim.x * EW_THREADS) {
        const int b = (int)(i / cvox);
        const size_t r = i - (size_t)b * cvox;
        const int x = (int)(r % cb.ex), y = (int)((r / cb.ex) % cb.ey), z = (int)(r / ((size_t)cb.ex * cb.ey));
        const int c = cb.first + b;
        const int bx = c % cb.sgx, by = (c / cb.sgx) % cb.sgy, bz = c / (cb.sgx * cb.sgy);
        const float v = est[(size_t)b * pvox + ((size_t)(z + cb.lz) * cb.py + (y + cb.ly)) * cb.px + (x + cb.lx)];
        const int oz = cb.z0 + bz * cb.cz + z - cb.out_z0, oy = cb.y0 + by * cb.cy + y, ox = cb.x0 + bx * cb.cx + x;
        out[((size_t)oz * cb.ny + oy) * cb.nx + ox] = quantise_u16(v + offset, 65535.0f);
    }
}
hipError_t launch_chunk_gather(const uint16_t* in, const ChunkBatch& cb, float offset, float* out,
                               hipStream_t s, uint16_t* out16) {
    const size_t total = (size_t)cb.pz * cb.py * cb.px * (size_t)cb.count;
    const unsigned blocks = (unsigned)std::min<size_t>((total + EW_THREADS - 1) / EW_THREADS, 1u << 20);
    hipLaunchKernelGGL(chunk_gather_kernel, dim3(blocks), dim3(EW_THREADS), 0, s, in, cb, offset, out, out16);
    return hipGetLastError();
}
hipError_t launch_chunk_scatter(const float* est, const ChunkBatch& cb, float offset, uint16_t* out,
                                hipStream_t s) {
    const size_t total = (size_t)cb.ez * cb.ey * cb.ex * (size_t)cb.count;
    const unsigned blocks = (unsigned)std::min<size_t>((total + EW_THREADS - 1) / EW_THREADS, 1u << 20);
    hipLaunchKernelGGL(chunk_scatter_kernel, dim3(blocks), dim3(EW_THREADS), 0, s, est, cb, offset, out);
    return hipGetLastError();
}

// ---- denominator of the aggregation as a convolution (stage_kernels.hip) -------------------------
// The stage kernels leave sum(rint(u 2^40)) of the blocks on their corner voxels (64-bit integers, exact
// whatever the order of the atomics).  C = fl32(cw 2^-40); out(i) = sum_{t=0..7} k[t] * in(i - t) along
// one axis: a block corner c with weight u spreads u * k[t] over voxels c .. c + 7.  Fixed summation
// order (fmaf chain, t ascending, from +0): DESIGN.md 3.8, oracle orc_den_from_corners.
struct Win1D {
    float k[8];
};
__device__ __forceinline__ float cw_to_float(unsigned long long c) {
    // c < 2^53 (at most 2^12 blocks of at most 2^40 per corner): both conversions and the scaling are
    // exact, the cast to float rounds once
    return (float)(((double)(unsigned)(c >> 32) * 4294967296.0 + (double)(unsigned)c) * 9.094947017729282e-13);
}
// int64 fixed point -> fp32: fl64(num) (one rounding beyond 2^53), exact scaling, one rounding to fp32
__device__ __forceinline__ float num_to_float(long long v, double down) {
    return (float)(__ll2double_rn(v) * down);
}
// along y or z: one thread per line (lanes along x: coalesced), marching with the last 8 inputs
// in registers, so every input is read once.  Line (o, i): base = o * extent * inner + i,
// element stride `inner`.
// With XFIRST the input is the corner-weight volume and the value fed into the line is the 8-tap
// convolution along x of its row (8 neighbouring loads, L1 hits), which fuses the x and y passes.
template <bool XFIRST>
__global__ __launch_bounds__(EW_THREADS) void conv8_line_kernel(const void* __restrict__ in_,
                                                                float* __restrict__ out, size_t nlines,
                                                                size_t inner, int extent, Win1D w) {
    for (size_t l = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; l < nlines;
         l += (size_t)gridDim.x * EW_THREADS) {
        const size_t o = l / inner, i = l - o * inner;
        const size_t base = o * (size_t)extent * inner + i;
        float h[8];
#pragma unroll
        for (int t = 0; t < 8; t++) h[t] = 0.0f;
        for (int e = 0; e < extent; e++) {
#pragma unroll
            for (int t = 7; t > 0; t--) h[t] = h[t - 1];
            if (XFIRST) {
                // lines run along y, `inner` is the row length and i the x coordinate
                const unsigned long long* row = static_cast<const unsigned long long*>(in_) + base + (size_t)e * inner;
                float ax = 0.0f;
#pragma unroll
                for (int t = 0; t < 8; t++)
                    if ((size_t)t <= i) ax = fmaf(w.k[t], cw_to_float(*(row - t)), ax);
                h[0] = ax;
            } else {
                h[0] = static_cast<const float*>(in_)[base + (size_t)e * inner];
            }
            float acc = 0.0f;
#pragma unroll
            for (int t = 0; t < 8; t++) acc = fmaf(w.k[t], h[t], acc);
            out[base + (size_t)e * inner] = acc;
        }
    }
}

// Fused x / y pass for row lengths that are multiples of 4: one thread owns four consecutive x
// outputs and marches along y.  Per row it loads its own four corner weights and the eight to its left
// (six 16-byte loads for four outputs), forms the four x-convolutions in registers and pushes them
// through four 8-deep y shift registers.
__global__ __launch_bounds__(EW_THREADS) void conv8_xy4_kernel(const unsigned long long* __restrict__ in,
                                                               float* __restrict__ out, size_t nlines,
                                                               int nx4, int ny, Win1D w) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    for (size_t l = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; l < nlines;
         l += (size_t)gridDim.x * EW_THREADS) {
        const size_t o = l / (size_t)nx4;              // (volume, z) index
        const int x0 = 4 * (int)(l - o * (size_t)nx4);
        const size_t nx = 4 * (size_t)nx4;
        const size_t base = o * (size_t)ny * nx + (size_t)x0;
        float h[8][4];
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) h[t][j] = 0.0f;
        for (int y = 0; y < ny; y++) {
            const unsigned long long* row = in + base + (size_t)y * nx;
            float win_[12];
#pragma unroll
            for (int q = 0; q < 6; q++) {
                // elements row[2 q - 8], row[2 q - 7]: left of the row start they count as zero
                u64x2 v = {0ull, 0ull};
                if (x0 + 2 * q - 8 >= 0) v = *reinterpret_cast<const u64x2*>(row + 2 * q - 8);
                win_[2 * q] = cw_to_float(v.x);
                win_[2 * q + 1] = cw_to_float(v.y);
            }
#pragma unroll
            for (int t = 7; t > 0; t--)
#pragma unroll
                for (int j = 0; j < 4; j++) h[t][j] = h[t - 1][j];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float ax = 0.0f;
#pragma unroll
                for (int t = 0; t < 8; t++) ax = fmaf(w.k[t], win_[8 + j - t], ax);
                h[0][j] = ax;
            }
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float acc = 0.0f;
#pragma unroll
                for (int t = 0; t < 8; t++) acc = fmaf(w.k[t], h[t][j], acc);
                r[j] = acc;
            }
            *reinterpret_cast<float4*>(out + base + (size_t)y * nx) = make_float4(r[0], r[1], r[2], r[3]);
        }
    }
}

// The z pass of the denominator convolution fused into the normalisation (round 3): a thread owns W
// consecutive x of one (volume, y) line and marches along z with the 8-deep shift register of
// conv8_line_kernel -- den = the same fmaf chain, in the same order -- then
// out = fl32(fl64(num) 2^(E - 43)) / den (+ clip, or + offset, clamp, rint, uint16).  Saves the pass that
// would write den and the one that would read it back; bit-identical to the separate passes of the staged
// entry points.
template <int W, bool U16>
__global__ __launch_bounds__(EW_THREADS) void normalize_zconv_kernel(const long long* __restrict__ num,
                                                                     const double* __restrict__ qscale,
                                                                     const float* __restrict__ txy,
                                                                     void* __restrict__ out, size_t nlines,
                                                                     size_t plane, int nz, Win1D w, float lo,
                                                                     float hi, int do_clip, float offset,
                                                                     const float* __restrict__ pair_src,
                                                                     float* __restrict__ pair_out,
                                                                     uint16_t* __restrict__ match16,
                                                                     float match_offset) {
    // pair_out (fp32 output, W = 4 only): additionally the interleaved volume (pair_src, out) the Wiener
    // kernel gathers from, so that it does not cost a pass of its own
    // match16 (fp32 output): additionally the estimate rounded to counts, XOR 0x8000 -- what stage 2 of the
    // uint16 pipelines matches on in the integer kernel (OpRoundCountsU16 as a pass: 2.5 ms)
    typedef long long i64x2 __attribute__((ext_vector_type(2)));
    const size_t lines_per_vol = plane / W;
    for (size_t l = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; l < nlines;
         l += (size_t)gridDim.x * EW_THREADS) {
        const size_t o = l / lines_per_vol, i = l - o * lines_per_vol;
        const size_t base = o * (size_t)nz * plane + (size_t)W * i;
        const double down = qscale[2 * o + 1];
        float h[8][W];
#pragma unroll
        for (int t = 0; t < 8; t++)
#pragma unroll
            for (int j = 0; j < W; j++) h[t][j] = 0.0f;
        for (int z = 0; z < nz; z++) {
            const size_t at = base + (size_t)z * plane;
#pragma unroll
            for (int t = 7; t > 0; t--)
#pragma unroll
                for (int j = 0; j < W; j++) h[t][j] = h[t - 1][j];
            float a[W];
            if (W == 4) {
                const float4 c = *reinterpret_cast<const float4*>(txy + at);
                const i64x2 m0 = *reinterpret_cast<const i64x2*>(num + at);
                const i64x2 m1 = *reinterpret_cast<const i64x2*>(num + at + 2);
                h[0][0] = c.x; h[0][1 % W] = c.y; h[0][2 % W] = c.z; h[0][3 % W] = c.w;
                a[0] = num_to_float(m0.x, down); a[1 % W] = num_to_float(m0.y, down);
                a[2 % W] = num_to_float(m1.x, down); a[3 % W] = num_to_float(m1.y, down);
            } else {
                h[0][0] = txy[at];
                a[0] = num_to_float(num[at], down);
            }
            float r[W];
#pragma unroll
            for (int j = 0; j < W; j++) {
                float den = 0.0f;
#pragma unroll
                for (int t = 0; t < 8; t++) den = fmaf(w.k[t], h[t][j], den);
                r[j] = a[j] / den;
            }
            if (U16) {
                uint16_t* o16 = static_cast<uint16_t*>(out) + at;
                if (W == 4) {
                    const uint32_t q0 = quantise_u16(r[0] + offset, 65535.0f), q1 = quantise_u16(r[1 % W] + offset, 65535.0f);
                    const uint32_t q2 = quantise_u16(r[2 % W] + offset, 65535.0f), q3 = quantise_u16(r[3 % W] + offset, 65535.0f);
                    *reinterpret_cast<uint2*>(o16) = make_uint2(q0 | (q1 << 16), q2 | (q3 << 16));
                } else {
                    o16[0] = quantise_u16(r[0] + offset, 65535.0f);
                }
            } else {
                float* o32 = static_cast<float*>(out) + at;
#pragma unroll
                for (int j = 0; j < W; j++)
                    if (do_clip) r[j] = fminf(fmaxf(r[j], lo), hi);
                if (match16) {
                    uint16_t* m16 = match16 + at;
                    if (W == 4) {
                        const uint32_t q0 = quantise_u16(r[0] + match_offset, 65535.0f) ^ 0x8000u;
                        const uint32_t q1 = quantise_u16(r[1 % W] + match_offset, 65535.0f) ^ 0x8000u;
                        const uint32_t q2 = quantise_u16(r[2 % W] + match_offset, 65535.0f) ^ 0x8000u;
                        const uint32_t q3 = quantise_u16(r[3 % W] + match_offset, 65535.0f) ^ 0x8000u;
                        *reinterpret_cast<uint2*>(m16) = make_uint2(q0 | (q1 << 16), q2 | (q3 << 16));
                    } else {
                        m16[0] = (uint16_t)(quantise_u16(r[0] + match_offset, 65535.0f) ^ 0x8000u);
                    }
                }
                if (W == 4) {
                    *reinterpret_cast<float4*>(o32) = make_float4(r[0], r[1 % W], r[2 % W], r[3 % W]);
                    if (pair_out) {
                        const float4 s4 = *reinterpret_cast<const float4*>(pair_src + at);
                        float4* po = reinterpret_cast<float4*>(pair_out + 2 * at);
                        po[0] = make_float4(s4.x, r[0], s4.y, r[1 % W]);
                        po[1] = make_float4(s4.z, r[2 % W], s4.w, r[3 % W]);
                    }
                } else {
                    o32[0] = r[0];
                }
            }
        }
    }
}

static dim3 conv_blocks(size_t items) {
    size_t b = (items + EW_THREADS - 1) / EW_THREADS;
    if (b > 65536) b = 65536;
    return dim3((unsigned)(b ? b : 1));
}
// den's x / y passes only (cw -> tmp); the z pass then rides with the normalisation (launch_normalize_zconv)
hipError_t launch_den_xy_from_corners(const unsigned long long* cw, float* tmp, int nz, int ny, int nx, int batch,
                                      const float* win1d, hipStream_t s) {
    Win1D w;
    for (int t = 0; t < 8; t++) w.k[t] = win1d[t];
    if (nx % 4 == 0 && ((uintptr_t)cw & 15u) == 0 && ((uintptr_t)tmp & 15u) == 0) {
        const size_t lines4 = (size_t)batch * nz * (nx / 4);
        hipLaunchKernelGGL(conv8_xy4_kernel, conv_blocks(lines4), dim3(EW_THREADS), 0, s, cw, tmp, lines4, nx / 4, ny, w);
    } else {
        const size_t ylines = (size_t)batch * nz * nx;
        hipLaunchKernelGGL((conv8_line_kernel<true>), conv_blocks(ylines), dim3(EW_THREADS), 0, s,
                           static_cast<const void*>(cw), tmp, ylines, (size_t)nx, ny, w);
    }
    return hipGetLastError();
}

// out = fl32(fl64(num) 2^(E - 43)) / (txy (*)_z win): exactly one of out_f32 / out_u16
hipError_t launch_normalize_zconv(const long long* num, const double* qscale, const float* txy, float* out_f32,
                                  uint16_t* out_u16, int nz, int ny, int nx, int batch, const float* win1d,
                                  float lo, float hi, float offset, hipStream_t s, const float* pair_src,
                                  float* pair_out, int* pair_written, uint16_t* match16, float match_offset,
                                  int* match_written) {
    if (pair_written) *pair_written = 0;
    if (match_written) *match_written = 0;
    // (the rounded copy is of the UNCLIPPED fp32 estimate; 8-byte stores in the wide form)
    if (out_u16 || lo <= hi || ((uintptr_t)match16 & 7u) != 0) match16 = nullptr;
    if (match16 && match_written) *match_written = 1;
    Win1D w;
    for (int t = 0; t < 8; t++) w.k[t] = win1d[t];
    const size_t plane = (size_t)ny * nx;
    void* out = out_u16 ? static_cast<void*>(out_u16) : static_cast<void*>(out_f32);
    const bool wide = plane % 4 == 0 && ((uintptr_t)num & 15u) == 0 && ((uintptr_t)txy & 15u) == 0 &&
                      ((uintptr_t)out & 15u) == 0 && ((size_t)nz * plane) % 4 == 0;
    const size_t nlines = (size_t)batch * (wide ? plane / 4 : plane);
    const dim3 grid = conv_blocks(nlines);
    const int clip = lo <= hi ? 1 : 0;
    const float* nul = nullptr;
    float* nulw = nullptr;
    if (wide && out_u16) {
        hipLaunchKernelGGL((normalize_zconv_kernel<4, true>), grid, dim3(EW_THREADS), 0, s, num, qscale, txy, out,
                           nlines, plane, nz, w, lo, hi, clip, offset, nul, nulw, nullptr, 0.0f);
    } else if (wide) {
        const bool pw = pair_src && pair_out && ((uintptr_t)pair_src & 15u) == 0 && ((uintptr_t)pair_out & 15u) == 0;
        hipLaunchKernelGGL((normalize_zconv_kernel<4, false>), grid, dim3(EW_THREADS), 0, s, num, qscale, txy, out,
                           nlines, plane, nz, w, lo, hi, clip, offset, pw ? pair_src : nul, pw ? pair_out : nulw,
                           match16, match_offset);
        if (pair_written && pw) *pair_written = 1;
    } else if (out_u16) {
        hipLaunchKernelGGL((normalize_zconv_kernel<1, true>), grid, dim3(EW_THREADS), 0, s, num, qscale, txy, out,
                           nlines, plane, nz, w, lo, hi, clip, offset, nul, nulw, nullptr, 0.0f);
    } else {
        hipLaunchKernelGGL((normalize_zconv_kernel<1, false>), grid, dim3(EW_THREADS), 0, s, num, qscale, txy, out,
                           nlines, plane, nz, w, lo, hi, clip, offset, nul, nulw, match16, match_offset);
    }
    return hipGetLastError();
}

// the staged entry point's form: all three passes, den written
hipError_t launch_den_from_corners(const unsigned long long* cw, float* tmp, float* den, int nz, int ny, int nx,
                                   int batch, const float* win1d, hipStream_t s) {
    hipError_t e = launch_den_xy_from_corners(cw, tmp, nz, ny, nx, batch, win1d, s);
    if (e != hipSuccess) return e;
    Win1D w;
    for (int t = 0; t < 8; t++) w.k[t] = win1d[t];
    const size_t zlines = (size_t)batch * ny * nx;
    hipLaunchKernelGGL((conv8_line_kernel<false>), conv_blocks(zlines), dim3(EW_THREADS), 0, s,
                       static_cast<const void*>(tmp), den, zlines, (size_t)ny * nx, nz, w);
    return hipGetLastError();
}

__global__ __launch_bounds__(EW_THREADS) void num_to_float_kernel(const long long* __restrict__ num,
                                                                  const double* __restrict__ qscale,
                                                                  float* __restrict__ out, size_t nvox) {
    const double down = qscale[2 * blockIdx.y + 1];
    const size_t off = (size_t)blockIdx.y * nvox;
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < nvox; i += (size_t)gridDim.x * EW_THREADS)
        out[off + i] = num_to_float(num[off + i], down);
}
hipError_t launch_num_to_float(const long long* num, const double* qscale, float* out, size_t nvox, int batch,
                               hipStream_t s) {
    size_t b = (nvox + EW_THREADS - 1) / EW_THREADS;
    if (b > 16384) b = 16384;
    hipLaunchKernelGGL(num_to_float_kernel, dim3((unsigned)(b ? b : 1), (unsigned)batch), dim3(EW_THREADS), 0, s, num,
                       qscale, out, nvox);
    return hipGetLastError();
}

// ---- the numerator's unit -----------------------------------------------------------
// E with max |v| < 2^E from the largest |v| bit pattern of each volume (an integer maximum: exact and
// order-independent), then qscale[2 b] = 2^(43 - E), qscale[2 b + 1] = 2^(E - 43).
__global__ __launch_bounds__(EW_THREADS) void absmax_bits_kernel(const float* __restrict__ vol, size_t nvox,
                                                                 unsigned* __restrict__ maxbits) {
    const float* v = vol + (size_t)blockIdx.y * nvox;
    unsigned m = 0;
    for (size_t i = (size_t)blockIdx.x * EW_THREADS + threadIdx.x; i < nvox; i += (size_t)gridDim.x * EW_THREADS)
        m = max(m, __float_as_uint(v[i]) & 0x7FFFFFFFu);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(maxbits + blockIdx.y, m);
}
__global__ void qscale_kernel(const unsigned* __restrict__ maxbits, int batch, int fixed_exp,
                              double* __restrict__ qscale, unsigned* __restrict__ status) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int E = fixed_exp != INT32_MIN ? fixed_exp : (int)(maxbits[b] >> 23) - 126;
    // DESIGN.md 3.8, domain: beyond |v| < 2^56 the squares of transform coefficients (|Y| <= sqrt(8192) max|v|)
    // and of voxel differences leave fp32; infinities and NaNs land here too.  The pipeline still runs (no
    // fault, garbage out); the host reports EXABM4D_ERR_INVALID at its next synchronisation (status bit 1).
    if (E > 56 && status) __hip_atomic_fetch_or(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    qscale[2 * b] = ldexp(1.0, 43 - E);
    qscale[2 * b + 1] = ldexp(1.0, E - 43);
}
hipError_t launch_qscale(const float* vol, size_t nvox, int batch, int fixed_exp, unsigned* maxbits,
                         double* qscale, hipStream_t s, unsigned* status) {
    if (fixed_exp == INT32_MIN) {
        hipError_t e = hipMemsetAsync(maxbits, 0, sizeof(unsigned) * (size_t)batch, s);
        if (e != hipSuccess) return e;
        size_t b = (nvox + (size_t)EW_THREADS * 8 - 1) / ((size_t)EW_THREADS * 8);
        if (b > 4096) b = 4096;
        hipLaunchKernelGGL(absmax_bits_kernel, dim3((unsigned)(b ? b : 1), (unsigned)batch), dim3(EW_THREADS), 0, s, vol,
                           nvox, maxbits);
    }
    hipLaunchKernelGGL(qscale_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, s, maxbits, batch, fixed_exp,
                       qscale, status);
    return hipGetLastError();
}

static inline unsigned ew_blocks(size_t n) {
    size_t b = (n + EW_THREADS - 1) / EW_THREADS;
    if (b > (size_t)EW_MAX_BLOCKS) b = EW_MAX_BLOCKS;
    if (b == 0) b = 1;
    return (unsigned)b;
}

hipError_t launch_normalize(const float* num, const float* den, float* out, size_t n, float lo,
                            float hi, hipStream_t s) {
    hipLaunchKernelGGL(normalize_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, s, num, den, out,
                       n, lo, hi, lo <= hi ? 1 : 0);
    return hipGetLastError();
}
hipError_t launch_counts_from_u16(const uint16_t* in, float* out, size_t n, float offset,
                                  hipStream_t s, uint16_t* out16) {
    if (out16)
        return launch_stream(OpCountsFromU16Both{in, out, out16, offset}, n,
                             aligned16(in, out) && ((uintptr_t)out16 & 15) == 0, s);
    return launch_stream(OpCountsFromU16{in, out, offset}, n, aligned16(in, out), s);
}
hipError_t launch_round_counts(const float* in, float* out_f32, uint16_t* out_u16x, size_t n, float offset,
                               hipStream_t s) {
    if (out_u16x)
        return launch_stream(OpRoundCountsU16{in, out_u16x, offset}, n,
                             aligned16(in, in) && ((uintptr_t)out_u16x & 15) == 0, s);
    return launch_stream(OpRoundCountsF32{in, out_f32, offset}, n, aligned16(in, out_f32), s);
}
hipError_t launch_normalize_u16(const float* num, const float* den, uint16_t* out, size_t n,
                                float offset, hipStream_t s) {
    return launch_stream(OpNormalizeU16{num, den, out, offset}, n, aligned16(num, den, out), s);
}
hipError_t launch_tf_forward_u16(const TfDev& t, const uint16_t* in, float* out, size_t n,
                                 hipStream_t s) {
    return launch_stream(OpTfForward<uint16_t>{t, in, out}, n, aligned16(in, out), s);
}
hipError_t launch_tf_forward_u16_lut(const TfDev& t, float* lut, const uint16_t* in, float* out,
                                     size_t n, hipStream_t s) {
    hipLaunchKernelGGL(tf_lut_build_kernel, dim3(65536 / EW_THREADS), dim3(EW_THREADS), 0, s, t, lut);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_stream(OpLutForward{lut, in, out}, n, aligned16(in, out), s);
}
hipError_t launch_tf_forward_f32(const TfDev& t, const float* in, float* out, size_t n,
                                 hipStream_t s) {
    return launch_stream(OpTfForward<float>{t, in, out}, n, aligned16(in, out), s);
}
hipError_t launch_tf_inverse(const TfDev& t, const float* in, void* out, size_t n, int quant,
                             hipStream_t s) {
    if (quant)
        return launch_stream(OpTfInverseU16{t, in, reinterpret_cast<uint16_t*>(out)}, n,
                             aligned16(in, out), s);
    hipLaunchKernelGGL(tf_inverse_float_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, s, t, in,
                       reinterpret_cast<float*>(out), n);
    return hipGetLastError();
}
hipError_t launch_tile_gather(const float* vol, int nz, int ny, int nx, const int* starts, int nb,
                              int patch, float* out, hipStream_t s) {
    const size_t pv = (size_t)patch * patch * patch;
    for (int b0 = 0; b0 < nb; b0 += 64) {
        const int cnt = nb - b0 < 64 ? nb - b0 : 64;
        PatchStarts st;
        for (int i = 0; i < 3 * cnt; i++) st.v[i] = starts[3 * b0 + i];
        dim3 grid(ew_blocks(pv) > 64 ? 64 : ew_blocks(pv), (unsigned)cnt);
        hipLaunchKernelGGL(tile_gather_kernel, grid, dim3(EW_THREADS), 0, s, vol, nz, ny, nx, st,
                           patch, out + (size_t)b0 * pv);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
hipError_t launch_tile_accumulate(const float* preds, const int* starts, int nb, int patch, int trim,
                                  float* acc, float* wgt, int nz, int ny, int nx, hipStream_t s) {
    const size_t pv = (size_t)patch * patch * patch;
    const int core = patch - 2 * trim;
    const size_t cv = (size_t)core * core * core;
    for (int b = 0; b < nb; b++) {
        hipLaunchKernelGGL(tile_accumulate_kernel, dim3(ew_blocks(cv)), dim3(EW_THREADS), 0, s,
                           preds + (size_t)b * pv, starts[3 * b], starts[3 * b + 1],
                           starts[3 * b + 2], patch, trim, acc, wgt, nz, ny, nx);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
hipError_t launch_tile_finalize(const TfDev& t, const float* acc, const float* wgt, uint16_t* out,
                                size_t n, hipStream_t s) {
    return launch_stream(OpTileFinalize{t, acc, wgt, out}, n, aligned16(acc, wgt, out), s);
}

}  // namespace exabm4d
