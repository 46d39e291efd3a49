// GroupNorm + LeakyReLU on NDHWC tensors, for the BM4DNet stage's forward passes (inference.predict).
//
// The U-Net (reference machine_learning/unet3d.py:137-208: Conv3d -> GroupNorm(gcd(8, C)) -> LeakyReLU(0.01),
// eighteen times per forward) runs its convolutions through MIOpen's NDHWC implicit-GEMM solvers.  PyTorch's
// GroupNorm kernels want NCDHW: every norm layer cost a layout copy in, two statistics / apply kernels, the
// activation as a pass of its own and a layout copy back for the next convolution -- nine passes over the
// tensor, 45 % of the forward's kernel time (rocprofv3, tools/dbg/unet_forward_trace.py).  Here: one pass for
// the statistics, one tiny kernel for the per-(sample, channel) scale and shift, one read-modify-write pass
// for normalisation and activation, all on the layout the convolutions produce and consume.
//
// x[b][s][c], s = (d, h, w) flattened, c fastest; groups of C / G consecutive channels.  C % 4 == 0 and
// (C / G) % 4 == 0: a float4 of channels never straddles a group.  fp32 data, fp64 statistics; the partial sums
// are combined in a fixed order, so the result is a deterministic function of the input.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "exabm4d_kernels.h"

namespace exabm4d {

constexpr int GN_THREADS = 256;

// Partial sums of one (sample, chunk of rows): part[((b * nchunk + chunk) * G + g) * 2 + {0: sum, 1: sum of squares}]
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const float* __restrict__ x, size_t spatial, int C,
                                                             int G, int nchunk, size_t rows_per_chunk,
                                                             double* __restrict__ part,
                                                             const float* __restrict__ cbias) {
    // cbias (optional): the preceding convolution's bias, added here instead of in a pass of its own --
    // the statistics are those of x + cbias[c]
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int lanes = C / 4;                         // float4 lanes per row
    const int rows_per_iter = GN_THREADS / lanes;    // lanes divides GN_THREADS (C in {16 ... 1024}, power of two)
    const int lane = threadIdx.x % lanes, rsub = threadIdx.x / lanes;
    const size_t r0 = (size_t)chunk * rows_per_chunk;
    const size_t r1 = r0 + rows_per_chunk < spatial ? r0 + rows_per_chunk : spatial;
    const float4* base = reinterpret_cast<const float4*>(x + (size_t)b * spatial * C) + lane;
    float s0 = 0.0f, s1 = 0.0f, q0 = 0.0f, q1 = 0.0f;     // two accumulators each: shorter dependency chains
    const float4 cb = cbias ? reinterpret_cast<const float4*>(cbias)[lane] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto ld = [&](size_t row) {
        float4 v = base[row * lanes];
        v.x += cb.x; v.y += cb.y; v.z += cb.z; v.w += cb.w;
        return v;
    };
    size_t r = r0 + rsub;
    for (; r + rows_per_iter < r1; r += 2 * (size_t)rows_per_iter) {
        const float4 a = ld(r);
        const float4 c = ld(r + rows_per_iter);
        s0 += (a.x + a.y) + (a.z + a.w);
        q0 += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
        s1 += (c.x + c.y) + (c.z + c.w);
        q1 += (c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w);
    }
    if (r < r1) {
        const float4 a = ld(r);
        s0 += (a.x + a.y) + (a.z + a.w);
        q0 += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
    }
    __shared__ double sh[GN_THREADS][2];
    sh[threadIdx.x][0] = (double)s0 + (double)s1;
    sh[threadIdx.x][1] = (double)q0 + (double)q1;
    __syncthreads();
    // thread g sums its group's entries in a fixed order: lanes [g * lanes / G, (g + 1) * lanes / G) of every row slot
    if ((int)threadIdx.x < G) {
        const int g = threadIdx.x, lpg = lanes / G;
        double s = 0.0, q = 0.0;
        for (int rs = 0; rs < rows_per_iter; rs++)
            for (int l = g * lpg; l < (g + 1) * lpg; l++) {
                s += sh[rs * lanes + l][0];
                q += sh[rs * lanes + l][1];
            }
        double* p = part + (((size_t)b * nchunk + chunk) * G + g) * 2;
        p[0] = s;
        p[1] = q;
    }
}

// a[b][c] = rstd * gamma[c], sh[b][c] = beta[c] - mean * a[b][c]   (y = a x + sh, as PyTorch's fused parameters)
__global__ void gn_params_kernel(const double* __restrict__ part, int batch, int C, int G, int nchunk,
                                 double count, const float* __restrict__ gamma, const float* __restrict__ beta,
                                 float eps, float* __restrict__ a, float* __restrict__ shift,
                                 const float* __restrict__ cbias) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * C) return;
    const int b = i / C, c = i - b * C, g = c / (C / G);
    double s = 0.0, q = 0.0;
    for (int k = 0; k < nchunk; k++) {
        const double* p = part + (((size_t)b * nchunk + k) * G + g) * 2;
        s += p[0];
        q += p[1];
    }
    const double mean = s / count;
    double var = q / count - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
    const float av = rstd * ga;
    a[i] = av;
    // y = av (x + cbias - mean) + beta = av x + (beta + av (cbias - mean))
    shift[i] = be + av * ((cbias ? cbias[c] : 0.0f) - (float)mean);
}

__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             size_t spatial, int C, const float* __restrict__ a,
                                                             const float* __restrict__ shift, float slope) {
    const int b = blockIdx.y;
    const int lanes = C / 4;
    const size_t n4 = spatial * (size_t)lanes;
    const float4* xi = reinterpret_cast<const float4*>(x + (size_t)b * spatial * C);
    float4* yo = reinterpret_cast<float4*>(y + (size_t)b * spatial * C);
    const float4* a4 = reinterpret_cast<const float4*>(a + (size_t)b * C);
    const float4* s4 = reinterpret_cast<const float4*>(shift + (size_t)b * C);
    // a thread keeps its float4 lane over the rows it visits when the stride is a multiple of `lanes`
    const size_t stride = (size_t)gridDim.x * GN_THREADS;
    size_t i = (size_t)blockIdx.x * GN_THREADS + threadIdx.x;
    const bool fixed_lane = stride % lanes == 0;
    float4 av = a4[i % lanes], sv = s4[i % lanes];
    for (; i < n4; i += stride) {
        if (!fixed_lane) {
            av = a4[i % lanes];
            sv = s4[i % lanes];
        }
        float4 v = xi[i];
        v.x = fmaf(v.x, av.x, sv.x);
        v.y = fmaf(v.y, av.y, sv.y);
        v.z = fmaf(v.z, av.z, sv.z);
        v.w = fmaf(v.w, av.w, sv.w);
        v.x = v.x > 0.0f ? v.x : v.x * slope;
        v.y = v.y > 0.0f ? v.y : v.y * slope;
        v.z = v.z > 0.0f ? v.z : v.z * slope;
        v.w = v.w > 0.0f ? v.w : v.w * slope;
        yo[i] = v;
    }
}

// ---- MaxPool3d(2) and trilinear x2 up-sampling (align_corners = True) on NDHWC ---------------------------
// The U-Net's four down- and four up-samplings (reference unet3d.py:211-342).  PyTorch's kernels for these walk
// an NDHWC tensor through generic strides (3.8 ms per call here) or want an NCDHW copy; a float4 of channels per
// thread makes both trivially coalesced on the layout the convolutions use.
__device__ __forceinline__ float max_nan(float a, float b) { return (b > a || b != b) ? b : a; }   // NaN wins, as in torch
__global__ __launch_bounds__(GN_THREADS) void maxpool2_ndhwc_kernel(const float4* __restrict__ x,
                                                                   float4* __restrict__ y, size_t total, int OD,
                                                                   int OH, int OW, int H, int W, int D, int lanes) {
    for (size_t o = (size_t)blockIdx.x * GN_THREADS + threadIdx.x; o < total; o += (size_t)gridDim.x * GN_THREADS) {
        const int l = (int)(o % lanes);
        size_t t = o / lanes;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH); t /= OH;
        const int od = (int)(t % OD);
        const size_t b = t / OD;
        const float4* p = x + ((((b * D + 2 * od) * H + 2 * oh) * (size_t)W + 2 * ow) * lanes + l);
        float4 m = p[0];
#pragma unroll
        for (int k = 1; k < 8; k++) {
            const float4 v = p[(((size_t)(k >> 2) * H + ((k >> 1) & 1)) * W + (k & 1)) * lanes];
            m.x = max_nan(m.x, v.x); m.y = max_nan(m.y, v.y); m.z = max_nan(m.z, v.z); m.w = max_nan(m.w, v.w);
        }
        y[o] = m;
    }
}
hipError_t launch_maxpool2_ndhwc(const float* x, float* y, int batch, int D, int H, int W, int C, hipStream_t s) {
    const int OD = D / 2, OH = H / 2, OW = W / 2, lanes = C / 4;
    const size_t total = (size_t)batch * OD * OH * OW * lanes;
    if (total == 0) return hipSuccess;
    size_t blocks = (total + GN_THREADS - 1) / GN_THREADS;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(maxpool2_ndhwc_kernel, dim3((unsigned)blocks), dim3(GN_THREADS), 0, s,
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), total, OD, OH, OW, H, W, D,
                       lanes);
    return hipGetLastError();
}

// out extent = 2 * in; source coordinate r * o with r = (in - 1) / (out - 1) in fp32, i0 = (int)(r o),
// i1 = i0 + (i0 < in - 1), weights (1 - lambda, lambda): PyTorch's upsample_trilinear3d with align_corners
struct UpAxis {
    int i0, i1;
    float w0, w1;
};
__device__ __forceinline__ UpAxis up_axis(int o, int in, float r) {
    const float src = r * (float)o;
    UpAxis a;
    a.i0 = (int)src;
    a.i1 = a.i0 + (a.i0 < in - 1 ? 1 : 0);
    a.w1 = src - (float)a.i0;
    a.w0 = 1.0f - a.w1;
    return a;
}
__global__ __launch_bounds__(GN_THREADS) void upsample2_ndhwc_kernel(const float4* __restrict__ x,
                                                                    float4* __restrict__ y, size_t total, int D,
                                                                    int H, int W, int lanes, float rd, float rh,
                                                                    float rw) {
    const int OD = 2 * D, OH = 2 * H, OW = 2 * W;
    for (size_t o = (size_t)blockIdx.x * GN_THREADS + threadIdx.x; o < total; o += (size_t)gridDim.x * GN_THREADS) {
        const int l = (int)(o % lanes);
        size_t t = o / lanes;
        const int ow = (int)(t % OW); t /= OW;
        const int oh = (int)(t % OH); t /= OH;
        const int od = (int)(t % OD);
        const size_t b = t / OD;
        const UpAxis ad = up_axis(od, D, rd), ah = up_axis(oh, H, rh), aw = up_axis(ow, W, rw);
        const float4* base = x + (b * D * H * (size_t)W) * lanes + l;
        auto at = [&](int d, int h, int w) { return base[(((size_t)d * H + h) * W + w) * lanes]; };
        auto lerp_w = [&](int d, int h) {
            const float4 a = at(d, h, aw.i0), c = at(d, h, aw.i1);
            return make_float4(aw.w0 * a.x + aw.w1 * c.x, aw.w0 * a.y + aw.w1 * c.y, aw.w0 * a.z + aw.w1 * c.z,
                               aw.w0 * a.w + aw.w1 * c.w);
        };
        auto lerp_h = [&](int d) {
            const float4 a = lerp_w(d, ah.i0), c = lerp_w(d, ah.i1);
            return make_float4(ah.w0 * a.x + ah.w1 * c.x, ah.w0 * a.y + ah.w1 * c.y, ah.w0 * a.z + ah.w1 * c.z,
                               ah.w0 * a.w + ah.w1 * c.w);
        };
        const float4 a = lerp_h(ad.i0), c = lerp_h(ad.i1);
        y[o] = make_float4(ad.w0 * a.x + ad.w1 * c.x, ad.w0 * a.y + ad.w1 * c.y, ad.w0 * a.z + ad.w1 * c.z,
                           ad.w0 * a.w + ad.w1 * c.w);
    }
}
hipError_t launch_upsample2_trilinear_ndhwc(const float* x, float* y, int batch, int D, int H, int W, int C,
                                            hipStream_t s) {
    const int lanes = C / 4;
    const size_t total = (size_t)batch * (2 * (size_t)D) * (2 * (size_t)H) * (2 * (size_t)W) * lanes;
    if (total == 0) return hipSuccess;
    auto ratio = [](int in) { return 2 * in > 1 ? (float)(in - 1) / (float)(2 * in - 1) : 0.0f; };
    size_t blocks = (total + GN_THREADS - 1) / GN_THREADS;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(upsample2_ndhwc_kernel, dim3((unsigned)blocks), dim3(GN_THREADS), 0, s,
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), total, D, H, W, lanes,
                       ratio(D), ratio(H), ratio(W));
    return hipGetLastError();
}

size_t groupnorm_workspace_bytes(int batch, size_t spatial, int C, int G) {
    // partial sums (at most 64 chunks per sample) + the two parameter arrays
    return (size_t)batch * 64 * (size_t)G * 2 * sizeof(double) + 2 * (size_t)batch * C * sizeof(float);
}

hipError_t launch_groupnorm_lrelu_ndhwc(const float* x, float* y, int batch, size_t spatial, int C, int G,
                                        const float* gamma, const float* beta, float eps, float slope,
                                        void* workspace, hipStream_t s, const float* cbias) {
    const int lanes = C / 4;
    const int rows_per_iter = GN_THREADS / lanes;
    // chunks: enough workgroups for the chip (>= ~2048 in all), at least 2 * rows_per_iter rows each, at most 64
    size_t nchunk = (2048 + (size_t)batch - 1) / (size_t)batch;
    const size_t max_by_rows = spatial / (2 * (size_t)rows_per_iter);
    if (nchunk > max_by_rows) nchunk = max_by_rows;
    if (nchunk > 64) nchunk = 64;
    if (nchunk < 1) nchunk = 1;
    size_t rows_per_chunk = (spatial + nchunk - 1) / nchunk;
    rows_per_chunk = (rows_per_chunk + rows_per_iter - 1) / rows_per_iter * rows_per_iter;
    nchunk = (spatial + rows_per_chunk - 1) / rows_per_chunk;
    double* part = static_cast<double*>(workspace);
    float* a = reinterpret_cast<float*>(part + (size_t)batch * 64 * G * 2);
    float* shift = a + (size_t)batch * C;
    hipLaunchKernelGGL(gn_stats_kernel, dim3((unsigned)nchunk, (unsigned)batch), dim3(GN_THREADS), 0, s, x, spatial,
                       C, G, (int)nchunk, rows_per_chunk, part, cbias);
    const int total = batch * C;
    hipLaunchKernelGGL(gn_params_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, part, batch, C, G,
                       (int)nchunk, (double)spatial * (double)(C / G), gamma, beta, eps, a, shift, cbias);
    const size_t n4 = spatial * (size_t)lanes;
    size_t blocks = (n4 + GN_THREADS - 1) / GN_THREADS;
    const size_t cap = (8192 + (size_t)batch - 1) / (size_t)batch;
    if (blocks > cap) blocks = cap;
    // (GN_THREADS is a multiple of the row's lanes -- checked by the caller -- so is the grid stride: a thread
    // keeps its (scale, shift) in registers)
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)blocks, (unsigned)batch), dim3(GN_THREADS), 0, s, x, y,
                       spatial, C, a, shift, slope);
    return hipGetLastError();
}

}  // namespace exabm4d
