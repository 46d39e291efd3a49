// metrics_kernels.hip -- SURVEY.md section 8 row f-4: background-offset statistics and quality
// metrics on volumes that already live in HBM (gfx950).
//
//   hist16_kernel        exact 65536-bin histogram of 16-bit keys (uint16 counts, or one 16-bit
//                        digit of an order-preserving fp64 key): every percentile / median / MAD the
//                        reference takes with np.percentile (machine_learning/transforms.py:414-438,
//                        scripts/estimate_background_offsets.py:31-67, machine_learning/metrics.py:352-424)
//                        follows from it on the host without touching the volume again.
//   masked_stats_kernel  sums of |pred - ref| split by a foreground mask, maxima, and the count of
//                        background voxels above a threshold (metrics.py:306-381, img_util compute_mae).
//   ssim3d_kernel        SSIM with a cubic uniform window and scipy's "reflect" boundary
//                        (utils/img_util.py:953-1003), fp64, box sums separable and marched along z.
//   minmax_kernel        data range for SSIM.
//
// All reductions go through per-workgroup partials and one fixed-order final pass, so results
// are run-to-run deterministic; integer-valued inputs give exact sums (every partial < 2^53).
#include "exabm4d_kernels.h"

namespace exabm4d {

// ---- 16-bit-key histogram ------------------------------------------------------------------------
// 128 KB of LDS hold all 65536 bins as packed 16-bit counters; a workgroup consumes at most
// HIST_CHUNK <= 65535 keys between flushes, so no counter can carry into its neighbour.  LDS
// integer atomics run at full rate (unlike ds_add_f32); only non-zero bins are flushed to HBM.
constexpr int HIST_T = 1024;
constexpr int HIST_V = 7;   // 16-byte vectors per lane per chunk: 1024 * 7 * 8 = 57344 keys

struct KeyU16 {
    const uint16_t* p;
    static constexpr int PER = 8;
    __device__ void load(size_t vec, uint32_t (&k)[8]) const {
        const uint4 v = reinterpret_cast<const uint4*>(p)[vec];
        k[0] = v.x & 0xFFFFu; k[1] = v.x >> 16; k[2] = v.y & 0xFFFFu; k[3] = v.y >> 16;
        k[4] = v.z & 0xFFFFu; k[5] = v.z >> 16; k[6] = v.w & 0xFFFFu; k[7] = v.w >> 16;
    }
    __device__ uint32_t one(size_t i) const { return p[i]; }
};
// int32 quantisation indices as the symbols of an escape code: values in [-32767, 32767] keep their
// own bin (value + 32768), everything beyond shares bin 0, the escape symbol.
struct KeyI32Symbol {
    const int32_t* p;
    static constexpr int PER = 4;
    __device__ static uint32_t sym(int v) { return (v < -32767 || v > 32767) ? 0u : (uint32_t)(v + 32768); }
    __device__ void load(size_t vec, uint32_t (&k)[8]) const {
        const int4 v = reinterpret_cast<const int4*>(p)[vec];
        k[0] = sym(v.x); k[1] = sym(v.y); k[2] = sym(v.z); k[3] = sym(v.w);
        k[4] = k[5] = k[6] = k[7] = 0xFFFFFFFFu;
    }
    __device__ uint32_t one(size_t i) const { return sym(p[i]); }
};
// Order-preserving 64-bit key of a value widened to fp64 (optionally of its absolute deviation
// from a centre): ascending unsigned keys <=> ascending values.  A radix selection over its four
// 16-bit digits gives exact order statistics of any element type: pass d counts digit d of the
// elements whose higher digits equal `prefix`.
template <class T>
struct KeyF64 {
    const T* p;
    double center;
    int absdev;
    int d;
    unsigned long long prefix;
    static constexpr int PER = 8;
    __device__ uint32_t digit(T x) const {
        double v = (double)x;
        if (absdev) v = fabs(v - center);
        const long long b = __double_as_longlong(v);
        const unsigned long long k = b < 0 ? ~(unsigned long long)b
                                           : ((unsigned long long)b | 0x8000000000000000ull);
        const int shift = 48 - 16 * d;
        if (d > 0 && (k >> (shift + 16)) != prefix) return 0xFFFFFFFFu;
        return (uint32_t)(k >> shift) & 0xFFFFu;
    }
    __device__ void load(size_t vec, uint32_t (&k)[8]) const {
        const T* q = p + vec * 8;
#pragma unroll
        for (int e = 0; e < 8; e++) k[e] = digit(q[e]);
    }
    __device__ uint32_t one(size_t i) const { return digit(p[i]); }
};

template <class Key>
__global__ __launch_bounds__(HIST_T) void hist16_kernel(Key key, size_t n, int vector_ok,
                                                        unsigned long long* __restrict__ hist) {
    extern __shared__ uint32_t bins[];   // 32768 dwords, two 16-bit counters each
    const int tid = threadIdx.x;
    for (int i = tid; i < 32768; i += HIST_T) bins[i] = 0u;
    __syncthreads();
    const size_t nvec = vector_ok ? n / Key::PER : 0;
    const size_t per_chunk = (size_t)HIST_T * HIST_V;
    const size_t nchunks = (nvec + per_chunk - 1) / per_chunk;
    for (size_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        uint32_t k[HIST_V][8];
#pragma unroll
        for (int j = 0; j < HIST_V; j++) {
            const size_t vec = chunk * per_chunk + (size_t)j * HIST_T + tid;
            if (vec < nvec) {
                key.load(vec, k[j]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++) k[j][e] = 0xFFFFFFFFu;
            }
        }
#pragma unroll
        for (int j = 0; j < HIST_V; j++)
#pragma unroll
            for (int e = 0; e < Key::PER; e++)
                if (k[j][e] != 0xFFFFFFFFu) atomicAdd(&bins[k[j][e] >> 1], 1u << ((k[j][e] & 1u) * 16));
        __syncthreads();
        for (int i = tid; i < 32768; i += HIST_T) {
            const uint32_t c = bins[i];
            if (c) {
                bins[i] = 0u;
                if (c & 0xFFFFu) atomicAdd(&hist[2 * i], (unsigned long long)(c & 0xFFFFu));
                if (c >> 16) atomicAdd(&hist[2 * i + 1], (unsigned long long)(c >> 16));
            }
        }
        __syncthreads();
    }
    // elements the vector path does not cover (ragged tail, or everything when the base pointer
    // is not 16-byte aligned): straight to HBM atomics, spread over the grid
    const size_t done = nvec * Key::PER;
    for (size_t i = done + (size_t)blockIdx.x * HIST_T + tid; i < n; i += (size_t)gridDim.x * HIST_T) {
        const uint32_t kk = key.one(i);
        if (kk != 0xFFFFFFFFu) atomicAdd(&hist[kk], 1ull);
    }
}

template <class Key>
static hipError_t launch_hist(Key key, size_t n, bool aligned, unsigned long long* hist, hipStream_t s) {
    hipError_t e = hipMemsetAsync(hist, 0, 65536 * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    if (n == 0) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hist16_kernel<Key>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    if (e != hipSuccess) return e;
    const size_t per_chunk = (size_t)HIST_T * HIST_V * Key::PER;
    size_t blocks = (n + per_chunk - 1) / per_chunk;
    if (blocks > 1024) blocks = 1024;   // 4 rounds of one workgroup per CU
    hipLaunchKernelGGL(hist16_kernel<Key>, dim3((unsigned)blocks), dim3(HIST_T), 131072, s, key, n,
                       aligned ? 1 : 0, hist);
    return hipGetLastError();
}

hipError_t launch_hist_u16(const uint16_t* vol, size_t n, unsigned long long* hist, hipStream_t s) {
    return launch_hist(KeyU16{vol}, n, ((uintptr_t)vol & 15u) == 0, hist, s);
}
hipError_t launch_hist_i32_clamped(const int32_t* idx, size_t n, unsigned long long* hist, hipStream_t s) {
    return launch_hist(KeyI32Symbol{idx}, n, ((uintptr_t)idx & 15u) == 0, hist, s);
}
hipError_t launch_hist_key(const void* vol, int dtype, size_t n, int absdev, double center, int digit,
                           unsigned long long prefix, unsigned long long* hist, hipStream_t s) {
    switch (dtype) {
        case 0: return launch_hist(KeyF64<uint16_t>{(const uint16_t*)vol, center, absdev, digit, prefix}, n, true, hist, s);
        case 1: return launch_hist(KeyF64<float>{(const float*)vol, center, absdev, digit, prefix}, n, true, hist, s);
        default: return launch_hist(KeyF64<double>{(const double*)vol, center, absdev, digit, prefix}, n, true, hist, s);
    }
}

// ---- fixed-order final reduction of per-workgroup partials ------------------------------------------
// partials[w * K + k]; column k is summed, or max/min-reduced when its bit is set in the masks.
constexpr int RED_T = 256;
__global__ __launch_bounds__(RED_T) void reduce_partials_kernel(const double* __restrict__ partials,
                                                                int nwg, int K, unsigned max_mask,
                                                                unsigned min_mask,
                                                                double* __restrict__ out) {
    __shared__ double sh[RED_T];
    for (int k = 0; k < K; k++) {
        const bool is_max = (max_mask >> k) & 1u, is_min = (min_mask >> k) & 1u;
        double a = is_max ? -INFINITY : (is_min ? INFINITY : 0.0);
        for (int w = threadIdx.x; w < nwg; w += RED_T) {
            const double v = partials[(size_t)w * K + k];
            a = is_max ? fmax(a, v) : (is_min ? fmin(a, v) : a + v);
        }
        sh[threadIdx.x] = a;
        __syncthreads();
        for (int s = RED_T / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                const double x = sh[threadIdx.x], y = sh[threadIdx.x + s];
                sh[threadIdx.x] = is_max ? fmax(x, y) : (is_min ? fmin(x, y) : x + y);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) out[k] = sh[0];
        __syncthreads();
    }
}

hipError_t launch_reduce_partials(const double* partials, int nwg, int K, unsigned max_mask,
                                  unsigned min_mask, double* out, hipStream_t s) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(RED_T), 0, s, partials, nwg, K, max_mask,
                       min_mask, out);
    return hipGetLastError();
}

// workgroup reduction of K per-lane doubles into partials[blockIdx.x * K + k] (fixed order)
template <int K, int T>
__device__ __forceinline__ void block_reduce_store(double (&v)[K], unsigned max_mask, unsigned min_mask,
                                                   double* __restrict__ partials) {
    __shared__ double sh[T];
    for (int k = 0; k < K; k++) {
        const bool is_max = (max_mask >> k) & 1u, is_min = (min_mask >> k) & 1u;
        sh[threadIdx.x] = v[k];
        __syncthreads();
        for (int s = T / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                const double x = sh[threadIdx.x], y = sh[threadIdx.x + s];
                sh[threadIdx.x] = is_max ? fmax(x, y) : (is_min ? fmin(x, y) : x + y);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) partials[(size_t)blockIdx.x * K + k] = sh[0];
        __syncthreads();
    }
}

// ---- masked absolute-error statistics --------------------------------------------------------------
// columns: 0 sum|p-r| over foreground, 1 sum|p-r| over background, 2 foreground voxels,
//          3 background voxels with p > thr, 4 max p, 5 max r, 6 max |p-r|
constexpr int MS_T = 256;
constexpr int MS_K = 7;
constexpr unsigned MS_MAXMASK = 0x70u;
template <class TP, class TR>
__global__ __launch_bounds__(MS_T) void masked_stats_kernel(const TP* __restrict__ pred,
                                                            const TR* __restrict__ ref,
                                                            const uint8_t* __restrict__ mask, size_t n,
                                                            double thr, double* __restrict__ partials) {
    double v[MS_K] = {0.0, 0.0, 0.0, 0.0, -INFINITY, -INFINITY, -INFINITY};
    for (size_t i = (size_t)blockIdx.x * MS_T + threadIdx.x; i < n; i += (size_t)gridDim.x * MS_T) {
        const double p = (double)pred[i], r = (double)ref[i];
        const bool fg = mask ? mask[i] != 0 : false;
        const double e = fabs(p - r);
        if (fg) {
            v[0] += e;
            v[2] += 1.0;
        } else {
            v[1] += e;
            if (p > thr) v[3] += 1.0;
        }
        v[4] = fmax(v[4], p);
        v[5] = fmax(v[5], r);
        v[6] = fmax(v[6], e);
    }
    block_reduce_store<MS_K, MS_T>(v, MS_MAXMASK, 0u, partials);
}

static inline int stats_blocks(size_t n) {
    size_t b = (n + MS_T - 1) / MS_T;
    if (b > 8192) b = 8192;
    return b ? (int)b : 1;
}

template <class TP>
static hipError_t masked_stats_ref(const TP* pred, const void* ref, int ref_dtype, const uint8_t* mask,
                                   size_t n, double thr, double* partials, int blocks, hipStream_t s) {
    switch (ref_dtype) {
        case 0:
            hipLaunchKernelGGL((masked_stats_kernel<TP, uint16_t>), dim3(blocks), dim3(MS_T), 0, s, pred,
                               (const uint16_t*)ref, mask, n, thr, partials);
            break;
        case 1:
            hipLaunchKernelGGL((masked_stats_kernel<TP, float>), dim3(blocks), dim3(MS_T), 0, s, pred,
                               (const float*)ref, mask, n, thr, partials);
            break;
        default:
            hipLaunchKernelGGL((masked_stats_kernel<TP, double>), dim3(blocks), dim3(MS_T), 0, s, pred,
                               (const double*)ref, mask, n, thr, partials);
    }
    return hipGetLastError();
}

int masked_stats_partials(size_t n) { return stats_blocks(n); }

hipError_t launch_masked_stats(const void* pred, int pred_dtype, const void* ref, int ref_dtype,
                               const uint8_t* mask, size_t n, double thr, double* partials,
                               double* out7, hipStream_t s) {
    const int blocks = stats_blocks(n);
    hipError_t e;
    switch (pred_dtype) {
        case 0: e = masked_stats_ref((const uint16_t*)pred, ref, ref_dtype, mask, n, thr, partials, blocks, s); break;
        case 1: e = masked_stats_ref((const float*)pred, ref, ref_dtype, mask, n, thr, partials, blocks, s); break;
        default: e = masked_stats_ref((const double*)pred, ref, ref_dtype, mask, n, thr, partials, blocks, s);
    }
    if (e != hipSuccess) return e;
    return launch_reduce_partials(partials, blocks, MS_K, MS_MAXMASK, 0u, out7, s);
}

// ---- min / max --------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(MS_T) void minmax_kernel(const T* __restrict__ a, size_t n,
                                                      double* __restrict__ partials) {
    double v[2] = {INFINITY, -INFINITY};
    for (size_t i = (size_t)blockIdx.x * MS_T + threadIdx.x; i < n; i += (size_t)gridDim.x * MS_T) {
        const double x = (double)a[i];
        v[0] = fmin(v[0], x);
        v[1] = fmax(v[1], x);
    }
    block_reduce_store<2, MS_T>(v, 0x2u, 0x1u, partials);
}

hipError_t launch_minmax(const void* a, int dtype, size_t n, double* partials, double* out2,
                         hipStream_t s) {
    const int blocks = stats_blocks(n);
    switch (dtype) {
        case 0: hipLaunchKernelGGL(minmax_kernel<uint16_t>, dim3(blocks), dim3(MS_T), 0, s, (const uint16_t*)a, n, partials); break;
        case 1: hipLaunchKernelGGL(minmax_kernel<float>, dim3(blocks), dim3(MS_T), 0, s, (const float*)a, n, partials); break;
        default: hipLaunchKernelGGL(minmax_kernel<double>, dim3(blocks), dim3(MS_T), 0, s, (const double*)a, n, partials);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_reduce_partials(partials, blocks, 2, 0x2u, 0x1u, out2, s);
}

// ---- SSIM, cubic uniform window, reflect boundary ---------------------------------------------------
// One workgroup owns a 16 (y) x 64 (x) column of outputs over a z-chunk and marches along z with
// running 3-D box sums of a, b, a^2, b^2, ab held in registers: the plane entering the window is
// added, the plane leaving it subtracted (the same running-sum scheme scipy's uniform_filter1d
// uses).  A plane's 2-D box sums are formed in LDS: y-pass into ys[5][16][RW], x-pass into
// registers.  For integer-valued input every sum is an exact integer below 2^53.
constexpr int SS_T = 256;
constexpr int SS_TY = 16;
constexpr int SS_TX = 64;
constexpr int SS_MAXW = 32;

__device__ __forceinline__ int reflect_idx(int i, int n) {
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

template <class T>
__global__ __launch_bounds__(SS_T) void ssim3d_kernel(const T* __restrict__ A, const T* __restrict__ B,
                                                      int nz, int ny, int nx, int w, int zc, double C1,
                                                      double C2, double* __restrict__ partials) {
    extern __shared__ double ss_lds[];
    const int RW = SS_TX + w - 1, RH = SS_TY + w - 1;
    double* ys = ss_lds;                                   // [5][SS_TY][RW]
    T* ra = reinterpret_cast<T*>(ys + 5 * SS_TY * RW);     // [RH][RW]
    T* rb = ra + RH * RW;
    const int left = w / 2;
    const int tiles_x = (nx + SS_TX - 1) / SS_TX, tiles_y = (ny + SS_TY - 1) / SS_TY;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y;
    const int tz = blockIdx.x / (tiles_x * tiles_y);
    const int x0 = tx * SS_TX, y0 = ty * SS_TY, zs = tz * zc, ze = min(nz, zs + zc);
    const int lane = threadIdx.x;
    const int ox = lane % SS_TX, oy = lane / SS_TX;   // outputs (oy + 4 j, ox), j < 4
    double acc[4][5];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int q = 0; q < 5; q++) acc[j][q] = 0.0;
    const double w3 = (double)w * (double)w * (double)w;
    double total = 0.0;

    auto plane = [&](int p, bool add) {
        const size_t src = (size_t)reflect_idx(p, nz) * ny;
        for (int it = lane; it < RH * RW; it += SS_T) {
            const int r = it / RW, c = it - r * RW;
            const size_t g = (src + reflect_idx(y0 - left + r, ny)) * nx + reflect_idx(x0 - left + c, nx);
            ra[it] = A[g];
            rb[it] = B[g];
        }
        __syncthreads();
        for (int it = lane; it < SS_TY * RW; it += SS_T) {
            const int y = it / RW, c = it - y * RW;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0;
            for (int t = 0; t < w; t++) {
                const double a = (double)ra[(y + t) * RW + c], b = (double)rb[(y + t) * RW + c];
                s0 += a;
                s1 += b;
                s2 += a * a;
                s3 += b * b;
                s4 += a * b;
            }
            ys[0 * SS_TY * RW + it] = s0;
            ys[1 * SS_TY * RW + it] = s1;
            ys[2 * SS_TY * RW + it] = s2;
            ys[3 * SS_TY * RW + it] = s3;
            ys[4 * SS_TY * RW + it] = s4;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int y = oy + 4 * j;
#pragma unroll
            for (int q = 0; q < 5; q++) {
                const double* row = ys + (q * SS_TY + y) * RW + ox;
                double s = 0.0;
                for (int t = 0; t < w; t++) s += row[t];
                acc[j][q] = add ? acc[j][q] + s : acc[j][q] - s;
            }
        }
        // the next call's first barrier orders these reads before ys is rewritten; ra/rb are
        // only rewritten by lanes that have passed the second barrier above
    };

    for (int p = zs - left; p < zs - left + w; p++) plane(p, true);
    for (int z = zs; z < ze; z++) {
        if (z > zs) {
            plane(z - left + w - 1, true);
            plane(z - 1 - left, false);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int y = y0 + oy + 4 * j, x = x0 + ox;
            if (y < ny && x < nx) {
                const double mu1 = acc[j][0] / w3, mu2 = acc[j][1] / w3;
                const double s1 = acc[j][2] / w3 - mu1 * mu1;
                const double s2 = acc[j][3] / w3 - mu2 * mu2;
                const double s12 = acc[j][4] / w3 - mu1 * mu2;
                const double num = (2.0 * mu1 * mu2 + C1) * (2.0 * s12 + C2);
                const double den = (mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2);
                total += num / (fmax(den, 1e-8) + 1e-6);
            }
        }
    }
    __syncthreads();
    double v[1] = {total};
    block_reduce_store<1, SS_T>(v, 0u, 0u, partials);
}

static inline int ssim_zchunk(int nz, int ny, int nx) {
    const long long tiles = (long long)((nx + SS_TX - 1) / SS_TX) * ((ny + SS_TY - 1) / SS_TY);
    int zc = 64;
    while (zc > 8 && tiles * ((nz + zc - 1) / zc) < 2048) zc >>= 1;
    return zc;
}
int ssim3d_partials(int nz, int ny, int nx) {
    const int zc = ssim_zchunk(nz, ny, nx);
    return ((nx + SS_TX - 1) / SS_TX) * ((ny + SS_TY - 1) / SS_TY) * ((nz + zc - 1) / zc);
}

template <class T>
static hipError_t launch_ssim_t(const T* a, const T* b, int nz, int ny, int nx, int w, double C1,
                                double C2, double* partials, int blocks, int zc, hipStream_t s) {
    const int RW = SS_TX + w - 1, RH = SS_TY + w - 1;
    const size_t lds = (size_t)5 * SS_TY * RW * sizeof(double) + (size_t)2 * RH * RW * sizeof(T);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ssim3d_kernel<T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ssim3d_kernel<T>, dim3(blocks), dim3(SS_T), lds, s, a, b, nz, ny, nx, w, zc, C1,
                       C2, partials);
    return hipGetLastError();
}

hipError_t launch_ssim3d(const void* a, const void* b, int dtype, int nz, int ny, int nx, int w,
                         double C1, double C2, double* partials, double* out1, hipStream_t s) {
    const int zc = ssim_zchunk(nz, ny, nx);
    const int blocks = ssim3d_partials(nz, ny, nx);
    hipError_t e;
    switch (dtype) {
        case 0: e = launch_ssim_t((const uint16_t*)a, (const uint16_t*)b, nz, ny, nx, w, C1, C2, partials, blocks, zc, s); break;
        case 1: e = launch_ssim_t((const float*)a, (const float*)b, nz, ny, nx, w, C1, C2, partials, blocks, zc, s); break;
        default: e = launch_ssim_t((const double*)a, (const double*)b, nz, ny, nx, w, C1, C2, partials, blocks, zc, s);
    }
    if (e != hipSuccess) return e;
    return launch_reduce_partials(partials, blocks, 1, 0u, 0u, out1, s);
}

int ssim3d_max_window() { return SS_MAXW; }

}  // namespace exabm4d
