// exabm4d_api.hip -- the C-ABI of libexabm4d.so (include/exabm4d.h): context, scratch, tables,
// argument checking and the launch sequences.  Host code only; kernels live in *_kernels.hip.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/exabm4d.h"
#include "exabm4d_kernels.h"


using namespace exabm4d;

struct exabm4d_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    float dct[64];
    float win[512];
    float win1d[8];            // the window's 1-D factor (fp32), for den = C (*) win
    float* win_dev = nullptr;
    float* tf_lut = nullptr;   // 65536-entry forward table for uint16 input (asinh)
    double win_beta = -1.0;
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    uint32_t* rcp_dev = nullptr;   // chunk coder: reciprocal table, [4097][2]
    void* codec_aux = nullptr;     // chunk coder: sizes / offsets / totals / status
    size_t codec_aux_bytes = 0;
    void* red = nullptr;       // metric entry points: histogram / partials / results
    size_t red_bytes = 0;
    int force_generic_bm = 0;  // exabm4d_set_option("force_generic_bm")
    int bm_guarded_copy = 0;   // exabm4d_set_option("bm_guarded_copy"): staged block matching on a guarded copy
    StageOpts stage;           // exabm4d_set_option("stage_pairvol" / "stage_strip" / "stage_chunks")
    BmOpts bm;                 // exabm4d_set_option("bm_xcd_mode" / "bm_carry" / "bm_carry_fault")
    // The 8-byte sums are zeroed on a second stream, under the block matching that precedes every stage
    // kernel (compute-bound, and it touches neither array): exabm4d_set_option("zero_overlap", 0) puts the
    // memsets back on the context's stream.
    int zero_overlap = 1;
    bool zero_on_side = false;  // the last zero_begin() went to the second stream
    // exabm4d_denoise_f32_host: large batches in double-buffered sub-batches ("host_pipeline" = 0: one piece)
    int host_pipeline = 1;
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_ev[3] = {nullptr, nullptr, nullptr};
    hipStream_t side = nullptr;
    hipEvent_t side_ev[2] = {nullptr, nullptr};     // [0] main -> side: the sums' last reader is done; [1] side -> main: zeroed
    unsigned* status_host = nullptr;   // one pinned, device-visible word: bit 0 = a carry wait of block matching ran out
    unsigned* status_dev = nullptr;
    int profile = 0;           // exabm4d_set_option("profile")
    int bm_int = 1;            // exabm4d_set_option("bm_int"): integer block matching on uint16 input
    int codec_version = 2;     // exabm4d_set_option("codec_version"): stream format the encoder writes
    int chunk_budget_mb = 32768;   // exabm4d_set_option("chunk_budget_mb"): scratch per batch of chunks
    hipEvent_t ev[2 * EXABM4D_PHASE_COUNT] = {};
    bool ev_used[EXABM4D_PHASE_COUNT] = {};
    std::string err;
};

static thread_local std::string g_err;

static int fail(exabm4d_ctx* ctx, int code, const std::string& msg) {
    g_err = msg;
    if (ctx) ctx->err = msg;
    return code;
}
static int fail_hip(exabm4d_ctx* ctx, hipError_t e, const char* what) {
    return fail(ctx, EXABM4D_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(ctx, expr)                                         \
    do {                                                           \
        hipError_t _e = (expr);                                    \
        if (_e != hipSuccess) return fail_hip((ctx), _e, #expr);   \
    } while (0)

// ---- tables -----------------------------------------------------------------
static double bessel_i0(double x) {
    double sum = 1.0, term = 1.0, q = x * x / 4.0;
    for (int k = 1; k < 200; k++) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}
static void make_tables(double beta, float* dct64, float* win512, float* win1d = nullptr) {
    const double pi = 3.14159265358979323846;
    for (int u = 0; u < 8; u++)
        for (int n = 0; n < 8; n++) {
            const double c = (u == 0) ? std::sqrt(1.0 / 8.0) : std::sqrt(2.0 / 8.0);
            dct64[u * 8 + n] = (float)(c * std::cos(pi * (2.0 * n + 1.0) * u / 16.0));
        }
    double k[8];
    for (int n = 0; n < 8; n++) {
        if (beta == 0.0) {
            k[n] = 1.0;
        } else {
            const double r = 2.0 * n / 7.0 - 1.0;
            k[n] = bessel_i0(beta * std::sqrt(1.0 - r * r)) / bessel_i0(beta);
        }
    }
    for (int z = 0; z < 8; z++)
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) win512[(z * 8 + y) * 8 + x] = (float)(k[z] * k[y] * k[x]);
    if (win1d)
        for (int n = 0; n < 8; n++) win1d[n] = (float)k[n];
}

static int check_params(exabm4d_ctx* ctx, const exabm4d_params* p) {
    if (!p) return fail(ctx, EXABM4D_ERR_INVALID, "params is NULL");
    if (p->size != sizeof(exabm4d_params))
        return fail(ctx, EXABM4D_ERR_INVALID, "params.size does not match this library");
    if (p->block != 8 || p->step != 4 || p->search != 11 || p->max_group != 16)
        return fail(ctx, EXABM4D_ERR_UNSUPPORTED,
                    "only block=8, step=4, search=11, max_group=16 are implemented");
    if (!(p->lambda_ht >= 0.0f) || !(p->c_match_ht > 0.0f) || !(p->c_match_wie > 0.0f) ||
        !(p->kaiser_beta >= 0.0f))
        return fail(ctx, EXABM4D_ERR_INVALID, "params: thresholds must be positive, beta >= 0");
    return EXABM4D_OK;
}
static int make_geom(exabm4d_ctx* ctx, int nz, int ny, int nx, int batch, VolGeom& g) {
    if (nz < 8 || ny < 8 || nx < 8) return fail(ctx, EXABM4D_ERR_INVALID, "every volume axis must be >= 8");
    if (batch < 1 || batch > 65535) return fail(ctx, EXABM4D_ERR_INVALID, "batch must be in [1, 65535]");
    g.nz = nz; g.ny = ny; g.nx = nx;
    g.gz = grid_count(nz); g.gy = grid_count(ny); g.gx = grid_count(nx);
    g.az = aligned_count(nz); g.ay = aligned_count(ny); g.ax = aligned_count(nx);
    g.nvox = (long long)nz * ny * nx;
    g.nref = (long long)g.gz * g.gy * g.gx;
    if (g.nref > 0x7FFFFFFFLL) return fail(ctx, EXABM4D_ERR_INVALID, "volume too large for one launch");
    if ((long long)ny * nx * 24 * 4 > 0xFFFFFFFFLL)
        return fail(ctx, EXABM4D_ERR_INVALID, "z-plane too large (24 planes must fit 32-bit byte offsets)");
    return EXABM4D_OK;
}
static uint32_t keymax_of(float sigma, float c_match) {
    const float tau512 = (float)((double)c_match * (double)sigma * (double)sigma * 512.0);
    uint32_t u;
    std::memcpy(&u, &tau512, 4);
    return (u & KEY_DMASK) + 0x800u;
}
static int ensure_window(exabm4d_ctx* ctx, double beta) {
    if (ctx->win_dev && ctx->win_beta == beta) return EXABM4D_OK;
    make_tables(beta, ctx->dct, ctx->win, ctx->win1d);
    if (!ctx->win_dev) HIP_TRY(ctx, hipMalloc((void**)&ctx->win_dev, sizeof(float) * 512));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->win_dev, ctx->win, sizeof(float) * 512, hipMemcpyHostToDevice));
    ctx->win_beta = beta;
    return EXABM4D_OK;
}
// Block matching's `guarded` variant streams whole plane rows by LDS-DMA and reads up to 124 bytes
// in front of the first and past the last row of the volume (bm_tile_kernel): such a volume must
// lie inside the scratch allocation with 256 mapped bytes on either side.
constexpr size_t GUARD_BYTES = 256;
static bool guarded_region_ok(const exabm4d_ctx* ctx, const void* ptr, size_t bytes) {
    const char* lo = static_cast<const char*>(ctx->scratch);
    const char* hi = lo + ctx->scratch_bytes + GUARD_BYTES;      // ensure_scratch allocates + GUARD_BYTES
    const char* p = static_cast<const char*>(ptr);
    return ctx->scratch && p >= lo + GUARD_BYTES && p + bytes + GUARD_BYTES <= hi;
}
static int ensure_scratch(exabm4d_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->scratch_bytes) return EXABM4D_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->side) HIP_TRY(ctx, hipStreamSynchronize(ctx->side));   // (memsets of a call that failed half way)
    if (ctx->copy_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    // + GUARD_BYTES: block matching reads up to 124 bytes past the last row of the library's own
    // volumes (bm_tile_kernel, `guarded`); the region in front of each of them is another scratch
    // region (checked per launch: guarded_region_ok)
    hipError_t e = hipMalloc(&ctx->scratch, bytes + GUARD_BYTES);
    if (e != hipSuccess) {
        char msg[160];
        std::snprintf(msg, sizeof msg, "device scratch allocation of %zu bytes failed: %s", bytes,
                      hipGetErrorString(e));
        return fail(ctx, EXABM4D_ERR_NOMEM, msg);
    }
    ctx->scratch_bytes = bytes;
    return EXABM4D_OK;
}
static inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
// The kernels raise bits of the context's status word instead of hanging (block matching's carry: bm_kernels.hip
// ORDER).  Looked at wherever the host has just synchronised with the context's stream.
static int check_async_status(exabm4d_ctx* ctx) {
    if (!ctx->status_host || *ctx->status_host == 0) return EXABM4D_OK;
    const unsigned bits = *ctx->status_host;
    *ctx->status_host = 0;
    if (bits & 1u) {
        ctx->bm.carry = 0;       // the assumption behind the carry failed on this device: do without it from now on
        return fail(ctx, EXABM4D_ERR_HIP,
                    "block matching: a tile waited for the tile below it longer than the poll limit (carry between "
                    "tiles, DESIGN.md 5.1c); the match tables of the calls since the last synchronisation are void. "
                    "The carry is now off for this context (option bm_carry = 0): repeat the call.");
    }
    if (bits & 2u)
        return fail(ctx, EXABM4D_ERR_INVALID,
                    "fp32 input outside the working range of the specification: a volume holds |v| >= 2^56, an "
                    "infinity or a NaN (the squares of its transform coefficients leave fp32, DESIGN.md 3.8); the "
                    "results of the calls since the last synchronisation are void");
    return fail(ctx, EXABM4D_ERR_HIP, "a kernel reported an unknown status bit");
}

static int make_tfdev(exabm4d_ctx* ctx, const exabm4d_transform* t, TfDev& d) {
    if (!t) return fail(ctx, EXABM4D_ERR_INVALID, "transform is NULL");
    if (t->size != sizeof(exabm4d_transform))
        return fail(ctx, EXABM4D_ERR_INVALID, "transform.size does not match this library");
    if (t->kind < 0 || t->kind > 2) return fail(ctx, EXABM4D_ERR_INVALID, "unknown transform kind");
    std::memset(&d, 0, sizeof d);
    d.kind = t->kind;
    d.wrapped = t->wrapped ? 1 : 0;
    d.woff = (float)t->wrap_offset;
    d.maxc = (float)t->max_count;
    d.off = (float)t->offset;
    d.scale = (float)t->scale;
    d.norm = (float)t->norm;
    d.gain = (float)t->gain;
    d.c38g2 = (float)((3.0 / 8.0) * t->gain * t->gain);
    d.rn2 = (float)(t->read_noise * t->read_noise);
    d.two_over_gain = (float)(2.0 / t->gain);
    d.cinvg2 = (float)(t->c_inv * t->gain * t->gain);
    d.mn = (float)t->mn;
    d.fden = (float)(t->mx - t->mn + 1e-8);
    d.clip = (float)t->clip;
    d.range = (float)(t->mx - t->mn);
    return EXABM4D_OK;
}

namespace {
struct ChunkRun {
    int i0, count;       // chunks [i0, i0 + count) of the axis ...
    int e, lo, hi;       // ... share the core extent and the halo in front / behind
};
// chunks of `chunk` voxels tile [c0, c1) inside a buffer axis of n voxels
std::vector<ChunkRun> chunk_runs(int n, int c0, int c1, int chunk, int halo) {
    std::vector<ChunkRun> runs;
    int i = 0;
    for (int start = c0; start < c1; start += chunk, i++) {
        const int e = std::min(chunk, c1 - start);
        const int lo = std::min(halo, start), hi = std::min(halo, n - (start + e));
        if (!runs.empty() && runs.back().e == e && runs.back().lo == lo && runs.back().hi == hi)
            runs.back().count++;
        else
            runs.push_back({i, 1, e, lo, hi});
    }
    return runs;
}
}  // namespace

extern "C" {

// for the other translation units of the library (comm_rccl.hip); not part of the ABI
int exabm4d_internal_fail(exabm4d_ctx* ctx, int code, const char* msg) { return fail(ctx, code, msg ? msg : ""); }
hipStream_t exabm4d_internal_stream(exabm4d_ctx* ctx) { return ctx->stream; }
int exabm4d_internal_device(exabm4d_ctx* ctx) { return ctx->device; }

int exabm4d_version(void) { return EXABM4D_VERSION; }

const char* exabm4d_last_error(const exabm4d_ctx* ctx) {
    if (ctx && !ctx->err.empty()) return ctx->err.c_str();
    return g_err.c_str();
}

int exabm4d_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int exabm4d_create(int device, exabm4d_ctx** out) {
    if (!out) return fail(nullptr, EXABM4D_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, EXABM4D_ERR_NODEVICE, "no HIP device visible (libexabm4d needs a gfx950 GPU)");
    if (device < 0 || device >= n) return fail(nullptr, EXABM4D_ERR_INVALID, "device index out of range");
    HIP_TRY(nullptr, hipSetDevice(device));
    exabm4d_ctx* ctx = new (std::nothrow) exabm4d_ctx();
    if (!ctx) return fail(nullptr, EXABM4D_ERR_NOMEM, "out of host memory");
    ctx->device = device;
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return fail_hip(nullptr, e, "hipStreamCreate");
    }
    ctx->own_stream = true;
    e = hipHostMalloc((void**)&ctx->status_host, sizeof(unsigned), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&ctx->status_dev, ctx->status_host, 0);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        if (ctx->status_host) (void)hipHostFree(ctx->status_host);
        delete ctx;
        return fail_hip(nullptr, e, "status word (hipHostMalloc)");
    }
    *ctx->status_host = 0;
    e = hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipEventCreateWithFlags(&ctx->side_ev[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)exabm4d_destroy(ctx);
        return fail_hip(nullptr, e, "second stream (hipStreamCreate / hipEventCreate)");
    }
    *out = ctx;
    return EXABM4D_OK;
}

int exabm4d_destroy(exabm4d_ctx* ctx) {
    if (!ctx) return EXABM4D_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->side) {
        (void)hipStreamSynchronize(ctx->side);
        (void)hipStreamDestroy(ctx->side);
    }
    for (int i = 0; i < 2; i++)
        if (ctx->side_ev[i]) (void)hipEventDestroy(ctx->side_ev[i]);
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    for (int i = 0; i < 3; i++)
        if (ctx->copy_ev[i]) (void)hipEventDestroy(ctx->copy_ev[i]);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->status_host) (void)hipHostFree(ctx->status_host);
    if (ctx->red) (void)hipFree(ctx->red);
    if (ctx->rcp_dev) (void)hipFree(ctx->rcp_dev);
    if (ctx->codec_aux) (void)hipFree(ctx->codec_aux);
    if (ctx->win_dev) (void)hipFree(ctx->win_dev);
    if (ctx->tf_lut) (void)hipFree(ctx->tf_lut);
    for (int i = 0; i < 2 * EXABM4D_PHASE_COUNT; i++)
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return EXABM4D_OK;
}

int exabm4d_set_stream(exabm4d_ctx* ctx, void* hip_stream) {
    if (!ctx) return fail(nullptr, EXABM4D_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;   // NULL = the HIP null stream (PyTorch's default)
    ctx->own_stream = false;
    return EXABM4D_OK;
}

int exabm4d_reset_stream(exabm4d_ctx* ctx) {
    if (!ctx) return fail(nullptr, EXABM4D_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream) return check_async_status(ctx);
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
    return check_async_status(ctx);
}

int exabm4d_sync(exabm4d_ctx* ctx) {
    if (!ctx) return fail(nullptr, EXABM4D_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return check_async_status(ctx);
}

int exabm4d_default_params(exabm4d_params* p) {
    if (!p) return fail(nullptr, EXABM4D_ERR_INVALID, "params is NULL");
    p->size = sizeof(exabm4d_params);
    p->block = 8;
    p->step = 4;
    p->search = 11;
    p->max_group = 16;
    p->lambda_ht = 2.7f;
    p->c_match_ht = 3.0f;
    p->c_match_wie = 0.6f;
    p->kaiser_beta = 2.0f;
    return EXABM4D_OK;
}

int exabm4d_set_option(exabm4d_ctx* ctx, const char* name, int value) {
    if (!ctx || !name) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    if (std::strcmp(name, "force_generic_bm") == 0) {
        ctx->force_generic_bm = value ? 1 : 0;
        return EXABM4D_OK;
    }
    if (std::strcmp(name, "bm_guarded_copy") == 0) {
        ctx->bm_guarded_copy = value ? 1 : 0;
        return EXABM4D_OK;
    }
    if (std::strcmp(name, "codec_version") == 0) {
        if (value != 1 && value != 2) return fail(ctx, EXABM4D_ERR_INVALID, "codec_version must be 1 or 2");
        ctx->codec_version = value;
        return EXABM4D_OK;
    }
    if (std::strcmp(name, "bm_int") == 0) {
        ctx->bm_int = value ? 1 : 0;
        return EXABM4D_OK;
    }
    if (std::strcmp(name, "stage_pairvol") == 0) {      // Wiener gathers from an interleaved (noisy, basic) volume
        ctx->stage.pairvol = value ? 1 : 0;
        return EXABM4D_OK;
    }
    if (std::strcmp(name, "bm_carry") == 0) {           // block matching: carry between the tiles of a column (0 off, 1 automatic, 2 forced)
        if (value < 0 || value > 2) return fail(ctx, EXABM4D_ERR_INVALID, "bm_carry must be 0, 1 or 2");
        ctx->bm.carry = value;
        return EXABM4D_OK;
    }
    if (std::strcmp(name, "host_pipeline") == 0) {      // exabm4d_denoise_f32_host: large batches in overlapped sub-batches
        ctx->host_pipeline = value ? 1 : 0;
        return EXABM4D_OK;
    }
    if (std::strcmp(name, "zero_overlap") == 0) {       // the sums' memsets under block matching (second stream) or in line
        ctx->zero_overlap = value ? 1 : 0;
        return EXABM4D_OK;
    }
    if (std::strcmp(name, "bm_carry_fault") == 0) {     // debug: every carry wait counts as run out (error-path test)
        ctx->bm.carry_fault = value ? 1 : 0;
        return EXABM4D_OK;
    }
    if (std::strcmp(name, "bm_xcd_mode") == 0) {        // block matching's workgroup order (bm_kernels.hip)
        ctx->bm.xcd_mode = value < 0 ? 0 : (value > 16 ? 16 : value);   // >= 2: strips of that many tile rows
        return EXABM4D_OK;
    }
    if (std::strcmp(name, "stage_strip") == 0) {        // tile-column order of the two-waves-per-group stage kernels (0 = raster, n = strips of n tile rows)
        ctx->stage.strip = value > 0 ? value : 0;
        return EXABM4D_OK;
    }
    if (std::strcmp(name, "stage_chunks") == 0) {       // diagnostic: z chunks of the stage kernels
        ctx->stage.chunks = value > 0 ? value : 0;
        return EXABM4D_OK;
    }
    if (std::strcmp(name, "chunk_budget_mb") == 0) {
        if (value < 1) return fail(ctx, EXABM4D_ERR_INVALID, "chunk_budget_mb must be >= 1");
        ctx->chunk_budget_mb = value;
        return EXABM4D_OK;
    }
    if (std::strcmp(name, "profile") == 0) {
        if (value && !ctx->ev[0])
            for (int i = 0; i < 2 * EXABM4D_PHASE_COUNT; i++) HIP_TRY(ctx, hipEventCreate(&ctx->ev[i]));
        ctx->profile = value ? 1 : 0;
        return EXABM4D_OK;
    }
    return fail(ctx, EXABM4D_ERR_INVALID, std::string("unknown option: ") + name);
}

// ---- memory helpers --------------------------------------------------------------------------------
int exabm4d_malloc(exabm4d_ctx* ctx, size_t bytes, void** dptr) {
    if (!ctx || !dptr) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    *dptr = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) return fail(ctx, EXABM4D_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return EXABM4D_OK;
}
int exabm4d_free(exabm4d_ctx* ctx, void* dptr) {
    if (!ctx) return fail(nullptr, EXABM4D_ERR_INVALID, "ctx is NULL");
    if (dptr) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(dptr));
    }
    return EXABM4D_OK;
}
int exabm4d_memcpy_h2d(exabm4d_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx || (!dst && bytes) || (!src && bytes)) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return EXABM4D_OK;
}
int exabm4d_memcpy_d2h(exabm4d_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx || (!dst && bytes) || (!src && bytes)) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return check_async_status(ctx);
}
int exabm4d_memset(exabm4d_ctx* ctx, void* dst, int value, size_t bytes) {
    if (!ctx || (!dst && bytes)) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    HIP_TRY(ctx, hipMemsetAsync(dst, value, bytes, ctx->stream));
    return EXABM4D_OK;
}
int exabm4d_event_create(exabm4d_ctx* ctx, void** ev) {
    if (!ctx || !ev) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    hipEvent_t e;
    HIP_TRY(ctx, hipEventCreate(&e));
    *ev = (void*)e;
    return EXABM4D_OK;
}
int exabm4d_event_destroy(exabm4d_ctx* ctx, void* ev) {
    if (!ctx) return fail(nullptr, EXABM4D_ERR_INVALID, "ctx is NULL");
    if (ev) HIP_TRY(ctx, hipEventDestroy((hipEvent_t)ev));
    return EXABM4D_OK;
}
int exabm4d_event_record(exabm4d_ctx* ctx, void* ev) {
    if (!ctx || !ev) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    HIP_TRY(ctx, hipEventRecord((hipEvent_t)ev, ctx->stream));
    return EXABM4D_OK;
}
int exabm4d_event_elapsed_ms(exabm4d_ctx* ctx, void* a, void* b, float* ms) {
    if (!ctx || !a || !b || !ms) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    HIP_TRY(ctx, hipEventSynchronize((hipEvent_t)b));
    HIP_TRY(ctx, hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return EXABM4D_OK;
}

// ---- geometry + tables --------------------------------------------------------------------------------
int exabm4d_grid_count(int n) { return grid_count(n); }
int exabm4d_grid_positions(int n, int32_t* pos) {
    if (!pos) return fail(nullptr, EXABM4D_ERR_INVALID, "pos is NULL");
    const int c = grid_count(n), a = aligned_count(n);
    for (int i = 0; i < c; i++) pos[i] = grid_pos(i, a, n);
    return EXABM4D_OK;
}
int exabm4d_tables(const exabm4d_params* p, float* dct64, float* win512) {
    int rc = check_params(nullptr, p);
    if (rc) return rc;
    if (!dct64 || !win512) return fail(nullptr, EXABM4D_ERR_INVALID, "NULL argument");
    make_tables((double)p->kaiser_beta, dct64, win512);
    return EXABM4D_OK;
}
int exabm4d_blockmatch_plan(const exabm4d_ctx* ctx, int nz, int ny, int nx, int batch, int32_t plan[6],
                            uint64_t* carry_bytes) {
    if (!plan || !carry_bytes) return EXABM4D_ERR_INVALID;
    VolGeom g;
    const int rc = make_geom(nullptr, nz, ny, nx, batch, g);
    if (rc) return rc;
    const BmPlan p = bm_plan(g, batch, ctx ? ctx->bm : BmOpts());
    plan[0] = p.tz; plan[1] = p.ty; plan[2] = p.tx; plan[3] = p.xq; plan[4] = p.carry; plan[5] = p.flat;
    *carry_bytes = (uint64_t)p.carry_bytes;
    return EXABM4D_OK;
}

// Scratch layout of one pipeline run (run_pipeline below walks it in this order)
namespace {
struct PipeLayout {
    size_t keys, num, basic, cw, tmp, pair, qscale, maxbits, carry, total;
};
PipeLayout pipe_layout(size_t n, size_t nref, int batch, int stages, size_t carry_bytes) {
    PipeLayout L;
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t o = at; at += align256(bytes); return o; };
    L.keys = take(nref * MAXG * sizeof(uint32_t));
    L.num = take(n * sizeof(long long));                        // numerator, int64 fixed point
    L.basic = take(stages >= 2 ? n * sizeof(float) : 0);
    L.cw = take(n * sizeof(unsigned long long));                // corner weights, int64 fixed point
    L.tmp = take(n * sizeof(float));                            // x / y passes of the denominator convolution
    L.pair = take(stages >= 2 ? 2 * n * sizeof(float) : 0);     // interleaved (noisy, basic) volume of the Wiener gathers
    L.qscale = take((size_t)batch * 2 * sizeof(double));
    L.maxbits = take((size_t)batch * sizeof(unsigned));
    L.carry = take(carry_bytes);                                // block matching's carry between tiles (BmPlan)
    L.total = at;
    return L;
}
}  // namespace

// scratch of one pipeline run under the given block-matching options (the carry's memory depends on them)
static size_t pipe_bytes(const BmOpts& bm, int nz, int ny, int nx, int batch, int stages) {
    VolGeom g;
    if (make_geom(nullptr, nz, ny, nx, batch, g) != EXABM4D_OK) return 0;
    return pipe_layout((size_t)g.nvox * (size_t)batch, (size_t)g.nref * (size_t)batch, batch, stages,
                       bm_plan(g, batch, bm).carry_bytes).total;
}
size_t exabm4d_scratch_bytes(int nz, int ny, int nx, int batch, int stages) {
    return pipe_bytes(BmOpts(), nz, ny, nx, batch, stages);     // default options; includes the carry (round 4)
}

// ---- staged entry points ---------------------------------------------------------------------------------
int exabm4d_blockmatch_dev(exabm4d_ctx* ctx, const float* vol, int nz, int ny, int nx, int batch,
                           float sigma, float c_match, const exabm4d_params* p, uint32_t* keys) {
    if (!ctx || !vol || !keys) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (!(sigma > 0.0f) || !(c_match > 0.0f)) return fail(ctx, EXABM4D_ERR_INVALID, "sigma and c_match must be > 0");
    VolGeom g;
    rc = make_geom(ctx, nz, ny, nx, batch, g);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const BmPlan plan = bm_plan(g, batch, ctx->bm);
    if (ctx->bm_guarded_copy) {
        // parity hook for the pipeline's path: match on a copy inside the scratch allocation, with
        // 256 bytes of poison on either side, through the kernel's `guarded` variant
        const size_t bytes = (size_t)g.nvox * (size_t)batch * sizeof(float);
        rc = ensure_scratch(ctx, align256(bytes + 512) + plan.carry_bytes);
        if (rc) return rc;
        char* base = static_cast<char*>(ctx->scratch);
        if (!guarded_region_ok(ctx, base + 256, bytes))
            return fail(ctx, EXABM4D_ERR_INVALID, "internal: guarded volume without mapped slack around it");
        HIP_TRY(ctx, hipMemsetAsync(base, 0xFF, bytes + 512, ctx->stream));      // NaN bit patterns
        HIP_TRY(ctx, hipMemcpyAsync(base + 256, vol, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(ctx, launch_blockmatch(reinterpret_cast<const float*>(base + 256), g, batch,
                                       keymax_of(sigma, c_match), keys, ctx->stream,
                                       ctx->force_generic_bm, 1, nullptr, plan, base + align256(bytes + 512),
                                       ctx->status_dev));
        return EXABM4D_OK;
    }
    rc = ensure_scratch(ctx, plan.carry_bytes);
    if (rc) return rc;
    HIP_TRY(ctx, launch_blockmatch(vol, g, batch, keymax_of(sigma, c_match), keys, ctx->stream,
                                   ctx->force_generic_bm, 0, nullptr, plan, ctx->scratch, ctx->status_dev));
    return EXABM4D_OK;
}

// Block matching on a uint16 volume the way the uint16 pipelines do it: fp32 counts and the biased
// uint16 copy side by side in guarded scratch, integer tile kernel where its tables are the float
// kernel's (else the float kernel), one-wave kernel for clamped last grid positions.
int exabm4d_blockmatch_u16_dev(exabm4d_ctx* ctx, const uint16_t* vol, int nz, int ny, int nx, int batch,
                               float sigma, float c_match, const exabm4d_params* p, uint32_t* keys) {
    if (!ctx || !vol || !keys) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (!(sigma > 0.0f) || !(c_match > 0.0f)) return fail(ctx, EXABM4D_ERR_INVALID, "sigma and c_match must be > 0");
    VolGeom g;
    rc = make_geom(ctx, nz, ny, nx, batch, g);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)g.nvox * (size_t)batch;
    const size_t fbytes = align256(n * sizeof(float));
    const BmPlan plan = bm_plan(g, batch, ctx->bm);
    const size_t vols = align256(2 * GUARD_BYTES + fbytes + align256(n * sizeof(uint16_t)) + GUARD_BYTES);
    rc = ensure_scratch(ctx, vols + plan.carry_bytes);
    if (rc) return rc;
    char* base = static_cast<char*>(ctx->scratch);
    float* f32 = reinterpret_cast<float*>(base + GUARD_BYTES);
    uint16_t* u16 = reinterpret_cast<uint16_t*>(base + 2 * GUARD_BYTES + fbytes);
    HIP_TRY(ctx, launch_counts_from_u16(vol, f32, n, 0.0f, ctx->stream, u16));
    const double tau512 = (double)c_match * (double)sigma * (double)sigma * 512.0;
    const bool use16 = ctx->bm_int && tau512 < 16777216.0 && (nx % 2) == 0 &&
                       guarded_region_ok(ctx, u16, n * sizeof(uint16_t));
    if (!guarded_region_ok(ctx, f32, n * sizeof(float)))
        return fail(ctx, EXABM4D_ERR_INVALID, "internal: guarded volume without mapped slack around it");
    HIP_TRY(ctx, launch_blockmatch(f32, g, batch, keymax_of(sigma, c_match), keys, ctx->stream,
                                   ctx->force_generic_bm, 1, use16 ? u16 : nullptr, plan, base + vols, ctx->status_dev));
    return EXABM4D_OK;
}

int exabm4d_match_decode(const uint32_t* keys16, int rz, int ry, int rx, int ny, int nx,
                         int64_t* idx, float* dist, int* count) {
    if (!keys16 || !idx || !dist || !count) return fail(nullptr, EXABM4D_ERR_INVALID, "NULL argument");
    int c = 0;
    for (int k = 0; k < MAXG; k++) {
        const uint32_t key = keys16[k];
        if (key == KEY_EMPTY) {
            idx[k] = -1;
            dist[k] = INFINITY;
            continue;
        }
        int dz, dy, dx;
        code_to_disp(key & KEY_CMASK, dz, dy, dx);
        idx[k] = ((int64_t)(rz + dz) * ny + (ry + dy)) * nx + (rx + dx);
        const uint32_t sb = key & KEY_DMASK;
        float s;
        std::memcpy(&s, &sb, 4);
        dist[k] = s / 512.0f;
        c++;
    }
    *count = c;
    return EXABM4D_OK;
}

int exabm4d_stage_dev(exabm4d_ctx* ctx, const float* noisy, const float* basic,
                      const uint32_t* keys, int nz, int ny, int nx, int batch, float sigma,
                      const exabm4d_params* p, int data_exp, float* num, float* den) {
    if (!ctx || !noisy || !keys || !num || !den) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (!(sigma > 0.0f)) return fail(ctx, EXABM4D_ERR_INVALID, "sigma must be > 0");
    if (data_exp != EXABM4D_DATA_EXP_AUTO && (data_exp < -200 || data_exp > 200))
        return fail(ctx, EXABM4D_ERR_INVALID, "data_exp must be EXABM4D_DATA_EXP_AUTO or within [-200, 200]");
    VolGeom g;
    rc = make_geom(ctx, nz, ny, nx, batch, g);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    rc = ensure_window(ctx, (double)p->kaiser_beta);
    if (rc) return rc;
    const float thr = (float)((double)p->lambda_ht * (double)sigma);
    const float sigma2 = (float)((double)sigma * (double)sigma);
    const size_t n = (size_t)g.nvox * (size_t)batch;
    // int64 numerator, int64 corner weights, fp32 ping-pong, [pair volume], the units
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t o = at; at += align256(bytes); return o; };
    const size_t o_num = take(n * sizeof(long long)), o_cw = take(n * sizeof(unsigned long long));
    const size_t o_tmp = take(n * sizeof(float)), o_pair = take(basic ? 2 * n * sizeof(float) : 0);
    const size_t o_qs = take((size_t)batch * 2 * sizeof(double)), o_mb = take((size_t)batch * sizeof(unsigned));
    rc = ensure_scratch(ctx, at);
    if (rc) return rc;
    char* sc = static_cast<char*>(ctx->scratch);
    long long* numq = reinterpret_cast<long long*>(sc + o_num);
    unsigned long long* cw = reinterpret_cast<unsigned long long*>(sc + o_cw);
    double* qs = reinterpret_cast<double*>(sc + o_qs);
    HIP_TRY(ctx, hipMemsetAsync(sc + o_num, 0, o_tmp, ctx->stream));       // numerator and corner weights
    HIP_TRY(ctx, launch_qscale(noisy, (size_t)g.nvox, batch, data_exp, reinterpret_cast<unsigned*>(sc + o_mb), qs,
                               ctx->stream, ctx->status_dev));
    HIP_TRY(ctx, launch_stage(noisy, basic, keys, g, batch, ctx->dct, ctx->win_dev, thr, sigma2, qs, numq, cw,
                              ctx->stream, ctx->stage, basic ? reinterpret_cast<float*>(sc + o_pair) : nullptr, 0));
    HIP_TRY(ctx, launch_num_to_float(numq, qs, num, (size_t)g.nvox, batch, ctx->stream));
    HIP_TRY(ctx, launch_den_from_corners(cw, reinterpret_cast<float*>(sc + o_tmp), den, g.nz, g.ny, g.nx, batch,
                                         ctx->win1d, ctx->stream));
    return EXABM4D_OK;
}

int exabm4d_normalize_dev(exabm4d_ctx* ctx, const float* num, const float* den, float* out,
                          size_t n, float clip_lo, float clip_hi) {
    if (!ctx || !num || !den || !out) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_normalize(num, den, out, n, clip_lo, clip_hi, ctx->stream));
    return EXABM4D_OK;
}

int exabm4d_counts_from_u16_dev(exabm4d_ctx* ctx, const uint16_t* in, float* out, size_t n, float offset) {
    if (!ctx || !in || !out) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_counts_from_u16(in, out, n, offset, ctx->stream));
    return EXABM4D_OK;
}

int exabm4d_round_counts_f32_dev(exabm4d_ctx* ctx, const float* in, float* out, size_t n, float offset) {
    if (!ctx || !in || !out) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_round_counts(in, out, nullptr, n, offset, ctx->stream));
    return EXABM4D_OK;
}

int exabm4d_normalize_u16_dev(exabm4d_ctx* ctx, const float* num, const float* den, uint16_t* out,
                              size_t n, float offset) {
    if (!ctx || !num || !den || !out) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_normalize_u16(num, den, out, n, offset, ctx->stream));
    return EXABM4D_OK;
}

// (float)v - offset is exact in fp32 for every uint16 v iff the offset has at most 7 fractional bits
// (17 integer bits of |v - offset| + 7 = 24) -- 0, 37, 100.5 ...; only then do two voxels of the
// fp32 counts differ by an exact integer and the integer matching kernel reproduce the float
// kernel's (and the oracle's) tables.  A percentile such as 36.73 takes the float kernel.
static bool offset_exact_in_fp32(float offset) {
    const float s = offset * 128.0f;
    return std::fabs(offset) <= 65536.0f && s == std::rint(s);
}

// Bracket one phase of a pipeline call with events when profiling is on.
struct PhaseTimer {
    exabm4d_ctx* ctx;
    int phase;
    PhaseTimer(exabm4d_ctx* c, int p) : ctx(c), phase(p) {
        if (ctx->profile) (void)hipEventRecord(ctx->ev[2 * phase], ctx->stream);
    }
    ~PhaseTimer() {
        if (ctx->profile) {
            (void)hipEventRecord(ctx->ev[2 * phase + 1], ctx->stream);
            ctx->ev_used[phase] = true;
        }
    }
};

// ---- zeroing of the 8-byte sums -------------------------------------------------------------------------------
// NUM and CW (16 bytes per voxel) are zeroed before every stage kernel: 2.7 ms per stage at 1024^3 when the
// memsets sit on the context's stream.  Block matching runs between the sums' last reader (the previous
// normalisation) and their next writer (the stage kernel), is bound by instruction issue and touches neither
// array: the memsets go to a second stream there -- zero_begin() after the last reader, zero_join() before the
// stage kernel -- and cost the step nothing.  Only where there is something to hide: below 2^25 voxels (0.1 ms
// of memsets) the two cross-stream dependencies cost more than they save (+15 us on a 64^3 patch's 1.8 ms).
static int zero_begin(exabm4d_ctx* ctx, long long* num, unsigned long long* cw, size_t n, hipStream_t s) {
    hipStream_t z = s;
    ctx->zero_on_side = ctx->zero_overlap && n >= ((size_t)1 << 25);
    if (ctx->zero_on_side) {
        HIP_TRY(ctx, hipEventRecord(ctx->side_ev[0], s));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->side, ctx->side_ev[0], 0));
        z = ctx->side;
    }
    HIP_TRY(ctx, hipMemsetAsync(num, 0, n * sizeof(long long), z));
    HIP_TRY(ctx, hipMemsetAsync(cw, 0, n * sizeof(unsigned long long), z));
    if (ctx->zero_on_side) HIP_TRY(ctx, hipEventRecord(ctx->side_ev[1], ctx->side));
    return EXABM4D_OK;
}
static int zero_join(exabm4d_ctx* ctx, hipStream_t s) {
    if (ctx->zero_on_side) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->side_ev[1], 0));
    ctx->zero_on_side = false;
    return EXABM4D_OK;
}

// ---- whole pipeline -----------------------------------------------------------------------------------------
// noisy: fp32 counts on device.  Exactly one of out_f32 / out_u16 is written.
static int run_pipeline(exabm4d_ctx* ctx, const float* noisy, float* out_f32, uint16_t* out_u16,
                        const VolGeom& g, int batch, float sigma, const exabm4d_params* p,
                        int stages, float clip_lo, float clip_hi, float u16_offset, char* scratch,
                        int noisy_guarded, int data_exp, const uint16_t* noisy16 = nullptr,
                        int match_counts = 0, float match_offset = 0.0f) {
    // noisy16: the same volume as uint16 counts XOR 0x8000, guarded like `noisy`, when the caller
    // is a uint16 pipeline: stage-1 matching then runs in integer arithmetic (bm_tile16_kernel),
    // provided its tables equal the float kernel's -- admission bound below 2^24, even row length
    // noisy_guarded: `noisy` lies inside the scratch allocation (mapped memory on both sides, see
    // ensure_scratch and bm_tile_kernel); a caller's own device buffer is not assumed to.
    // data_exp: E of the numerator's unit: 17 from the uint16 entry points,
    // EXABM4D_DATA_EXP_AUTO (from every volume's largest |v|) from the fp32 ones.
    // match_counts (the uint16 entry points): stage 2 matches on the basic estimate ROUNDED TO
    // COUNTS -- rint(clamp(basic + match_offset, 0, 65535)) -- so that it can run in integer arithmetic like
    // stage 1 (noisy16's memory is free by then and takes the rounded volume); where the integer kernel does
    // not apply, the float kernel runs on the same counts as fp32 (in `tmp`, dead between the stages).
    const size_t n = (size_t)g.nvox * (size_t)batch;
    const BmPlan plan = bm_plan(g, batch, ctx->bm);           // one plan for both matching launches
    const PipeLayout L = pipe_layout(n, (size_t)g.nref * (size_t)batch, batch, stages, plan.carry_bytes);
    uint32_t* keys = reinterpret_cast<uint32_t*>(scratch + L.keys);
    long long* num = reinterpret_cast<long long*>(scratch + L.num);
    float* basic = reinterpret_cast<float*>(scratch + L.basic);  // only touched when stages >= 2
    unsigned long long* cw = reinterpret_cast<unsigned long long*>(scratch + L.cw);
    float* tmp = reinterpret_cast<float*>(scratch + L.tmp);
    float* pairvol = reinterpret_cast<float*>(scratch + L.pair);
    double* qs = reinterpret_cast<double*>(scratch + L.qscale);

    const float thr = (float)((double)p->lambda_ht * (double)sigma);
    const float sigma2 = (float)((double)sigma * (double)sigma);
    hipStream_t s = ctx->stream;
    int pair_ready = 0;      // the first normalisation wrote the Wiener stage's (noisy, basic) volume
    // stage 2 of a uint16 pipeline in the integer kernel?  Decided here because the first
    // normalisation then also writes the rounded estimate (into noisy16's memory: stage 1 is done with it)
    const bool match_use16 = match_counts && stages >= 2 && noisy16 && ctx->bm_int &&
                             (double)p->c_match_wie * (double)sigma * (double)sigma * 512.0 < 16777216.0 &&
                             (g.nx % 2) == 0 && offset_exact_in_fp32(match_offset) &&
                             guarded_region_ok(ctx, noisy16, n * sizeof(uint16_t));
    int match16_ready = 0;
    if (ctx->profile)
        for (int i = 1; i < EXABM4D_PHASE_COUNT; i++) ctx->ev_used[i] = false;
    if ((noisy_guarded && !guarded_region_ok(ctx, noisy, n * sizeof(float))) ||
        (stages >= 2 && !guarded_region_ok(ctx, basic, n * sizeof(float))))
        return fail(ctx, EXABM4D_ERR_INVALID, "internal: guarded volume without mapped slack around it");

    {
        PhaseTimer t(ctx, EXABM4D_PHASE_ZERO_ACC_1);
        int rc = zero_begin(ctx, num, cw, n, s);
        if (rc) return rc;
        HIP_TRY(ctx, launch_qscale(noisy, (size_t)g.nvox, batch, data_exp,
                                   reinterpret_cast<unsigned*>(scratch + L.maxbits), qs, s, ctx->status_dev));
    }
    {
        PhaseTimer t(ctx, EXABM4D_PHASE_BLOCKMATCH_HT);
        const double tau512 = (double)p->c_match_ht * (double)sigma * (double)sigma * 512.0;
        const bool use16 = noisy16 && ctx->bm_int && tau512 < 16777216.0 && (g.nx % 2) == 0 &&
                           offset_exact_in_fp32(u16_offset) &&
                           guarded_region_ok(ctx, noisy16, n * sizeof(uint16_t));
        HIP_TRY(ctx, launch_blockmatch(noisy, g, batch, keymax_of(sigma, p->c_match_ht), keys, s,
                                       ctx->force_generic_bm, noisy_guarded, use16 ? noisy16 : nullptr, plan,
                                       scratch + L.carry, ctx->status_dev));
    }
    {
        PhaseTimer t(ctx, EXABM4D_PHASE_STAGE_HT);
        int rc = zero_join(ctx, s);
        if (rc) return rc;
        HIP_TRY(ctx, launch_stage(noisy, nullptr, keys, g, batch, ctx->dct, ctx->win_dev, thr, sigma2, qs, num, cw,
                                  s, ctx->stage));
        HIP_TRY(ctx, launch_den_xy_from_corners(cw, tmp, g.nz, g.ny, g.nx, batch, ctx->win1d, s));
    }
    if (stages >= 2) {
        {
            PhaseTimer t(ctx, EXABM4D_PHASE_NORMALIZE_BASIC);
            // ... and, where it can, the Wiener stage's interleaved (noisy, basic) volume
            HIP_TRY(ctx, launch_normalize_zconv(num, qs, tmp, basic, nullptr, g.nz, g.ny, g.nx, batch, ctx->win1d,
                                                1.0f, 0.0f, 0.0f, s, ctx->stage.pairvol ? noisy : nullptr, pairvol,
                                                &pair_ready, match_use16 ? const_cast<uint16_t*>(noisy16) : nullptr,
                                                match_offset, &match16_ready));
        }
        {
            PhaseTimer t(ctx, EXABM4D_PHASE_ZERO_ACC_2);
            int rc = zero_begin(ctx, num, cw, n, s);
            if (rc) return rc;
        }
        {
            PhaseTimer t(ctx, EXABM4D_PHASE_BLOCKMATCH_WIE);
            const float* match_on = basic;
            const uint16_t* match16 = nullptr;
            int match_guarded = 1;
            if (match_counts) {
                const bool use16 = match_use16;
                if (use16) {
                    uint16_t* m16 = const_cast<uint16_t*>(noisy16);      // our own scratch; stage 1 is done with it
                    if (!match16_ready)                                  // (normally written by the normalisation)
                        HIP_TRY(ctx, launch_round_counts(basic, nullptr, m16, n, match_offset, s));
                    match16 = m16;
                }
                // reference blocks at clamped grid positions (an extent - 8 that is no multiple of 4) go through
                // the one-wave kernel, which reads fp32: it needs the same counts as fp32
                const bool generic_too = ctx->force_generic_bm || g.gz != g.az || g.gy != g.ay || g.gx != g.ax;
                if (!use16 || generic_too) {
                    HIP_TRY(ctx, launch_round_counts(basic, tmp, nullptr, n, match_offset, s));
                    match_on = tmp;
                    match_guarded = guarded_region_ok(ctx, tmp, n * sizeof(float)) ? 1 : 0;
                }
            }
            HIP_TRY(ctx, launch_blockmatch(match_on, g, batch, keymax_of(sigma, p->c_match_wie), keys,
                                           s, ctx->force_generic_bm, match_guarded, match16, plan, scratch + L.carry,
                                           ctx->status_dev));
        }
        {
            PhaseTimer t(ctx, EXABM4D_PHASE_STAGE_WIE);
            int rc = zero_join(ctx, s);
            if (rc) return rc;
            HIP_TRY(ctx, launch_stage(noisy, basic, keys, g, batch, ctx->dct, ctx->win_dev, thr, sigma2, qs, num, cw,
                                      s, ctx->stage, pairvol, pair_ready));
            HIP_TRY(ctx, launch_den_xy_from_corners(cw, tmp, g.nz, g.ny, g.nx, batch, ctx->win1d, s));
        }
    }
    {
        PhaseTimer t(ctx, EXABM4D_PHASE_NORMALIZE_OUT);
        HIP_TRY(ctx, launch_normalize_zconv(num, qs, tmp, out_f32, out_u16, g.nz, g.ny, g.nx, batch, ctx->win1d,
                                            clip_lo, clip_hi, u16_offset, s));
    }
    return EXABM4D_OK;
}

static int pipeline_checks(exabm4d_ctx* ctx, const void* in, const void* out, int nz, int ny, int nx,
                           int batch, float sigma, const exabm4d_params* p, int stages, VolGeom& g) {
    if (!ctx || !in || !out) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (!(sigma > 0.0f)) return fail(ctx, EXABM4D_ERR_INVALID, "sigma must be > 0");
    if (stages != 1 && stages != 2) return fail(ctx, EXABM4D_ERR_INVALID, "stages must be 1 or 2");
    rc = make_geom(ctx, nz, ny, nx, batch, g);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return ensure_window(ctx, (double)p->kaiser_beta);
}

int exabm4d_denoise_f32_dev(exabm4d_ctx* ctx, const float* in, float* out, int nz, int ny, int nx,
                            int batch, float sigma, const exabm4d_params* p, int stages,
                            float clip_lo, float clip_hi) {
    VolGeom g;
    int rc = pipeline_checks(ctx, in, out, nz, ny, nx, batch, sigma, p, stages, g);
    if (rc) return rc;
    rc = ensure_scratch(ctx, pipe_bytes(ctx->bm, nz, ny, nx, batch, stages));
    if (rc) return rc;
    return run_pipeline(ctx, in, out, nullptr, g, batch, sigma, p, stages, clip_lo, clip_hi, 0.0f,
                        static_cast<char*>(ctx->scratch), 0, EXABM4D_DATA_EXP_AUTO);
}

int exabm4d_denoise_u16_dev(exabm4d_ctx* ctx, const uint16_t* in, uint16_t* out, int nz, int ny,
                            int nx, int batch, float sigma, float offset, const exabm4d_params* p,
                            int stages) {
    VolGeom g;
    int rc = pipeline_checks(ctx, in, out, nz, ny, nx, batch, sigma, p, stages, g);
    if (rc) return rc;
    const size_t n = (size_t)g.nvox * (size_t)batch;
    if (!(std::fabs(offset) <= 65536.0f))     // |v - offset| < 2^17: the fixed unit of the uint16 pipelines
        return fail(ctx, EXABM4D_ERR_INVALID, "offset must lie within [-65536, 65536]");
    const size_t base = pipe_bytes(ctx->bm, nz, ny, nx, batch, stages);
    const size_t fbytes = align256(n * sizeof(float));
    rc = ensure_scratch(ctx, base + fbytes + GUARD_BYTES + align256(n * sizeof(uint16_t)));
    if (rc) return rc;
    char* scratch = static_cast<char*>(ctx->scratch);
    float* noisy = reinterpret_cast<float*>(scratch + base);
    uint16_t* noisy16 = reinterpret_cast<uint16_t*>(scratch + base + fbytes + GUARD_BYTES);
    ctx->ev_used[EXABM4D_PHASE_COUNTS_FROM_U16] = false;
    {
        PhaseTimer t(ctx, EXABM4D_PHASE_COUNTS_FROM_U16);
        HIP_TRY(ctx, launch_counts_from_u16(in, noisy, n, offset, ctx->stream, noisy16));
    }
    return run_pipeline(ctx, noisy, nullptr, out, g, batch, sigma, p, stages, 0.0f, 0.0f, offset,
                        scratch, 1, EXABM4D_DATA_EXP_U16, noisy16, 1, offset);
}

// Chunk-local mode: every chunk (core + halo, the halo cut off where the buffer ends) is denoised
// in isolation, batches of equally shaped chunks per pipeline run; only the cores are written.

int exabm4d_denoise_chunked_u16_dev(exabm4d_ctx* ctx, const uint16_t* in, uint16_t* out, int nz, int ny,
                                    int nx, int zc0, int zc1, int chunk, int halo, float sigma,
                                    float offset, const exabm4d_params* p, int stages) {
    if (!ctx || !in || !out) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (!(sigma > 0.0f)) return fail(ctx, EXABM4D_ERR_INVALID, "sigma must be > 0");
    if (stages != 1 && stages != 2) return fail(ctx, EXABM4D_ERR_INVALID, "stages must be 1 or 2");
    if (nz < 1 || ny < 1 || nx < 1 || chunk < 1 || halo < 0 || halo > 64)
        return fail(ctx, EXABM4D_ERR_INVALID, "chunked: sizes >= 1, chunk >= 1, 0 <= halo <= 64");
    if (zc0 < 0 || zc1 > nz || zc0 >= zc1) return fail(ctx, EXABM4D_ERR_INVALID, "chunked: bad core plane range");
    if (!(std::fabs(offset) <= 65536.0f)) return fail(ctx, EXABM4D_ERR_INVALID, "offset must lie within [-65536, 65536]");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    rc = ensure_window(ctx, (double)p->kaiser_beta);
    if (rc) return rc;
    const std::vector<ChunkRun> runs[3] = {chunk_runs(nz, zc0, zc1, chunk, halo),
                                           chunk_runs(ny, 0, ny, chunk, halo),
                                           chunk_runs(nx, 0, nx, chunk, halo)};
    for (int a = 0; a < 3; a++)
        for (const ChunkRun& r : runs[a])
            if (r.e + r.lo + r.hi < 8)
                return fail(ctx, EXABM4D_ERR_INVALID, "chunked: a padded chunk would be thinner than one block (8)");
    ctx->ev_used[EXABM4D_PHASE_COUNTS_FROM_U16] = false;
    for (const ChunkRun& rz : runs[0])
        for (const ChunkRun& ry : runs[1])
            for (const ChunkRun& rx : runs[2]) {
                ChunkBatch cb;
                cb.nz = nz; cb.ny = ny; cb.nx = nx;
                cb.z0 = zc0 + rz.i0 * chunk; cb.y0 = ry.i0 * chunk; cb.x0 = rx.i0 * chunk;
                cb.cz = cb.cy = cb.cx = chunk;
                cb.ez = rz.e; cb.ey = ry.e; cb.ex = rx.e;
                cb.lz = rz.lo; cb.ly = ry.lo; cb.lx = rx.lo;
                cb.pz = rz.e + rz.lo + rz.hi; cb.py = ry.e + ry.lo + ry.hi; cb.px = rx.e + rx.lo + rx.hi;
                cb.sgy = ry.count; cb.sgx = rx.count;
                cb.out_z0 = zc0;
                const long long nchunks = (long long)rz.count * ry.count * rx.count;
                const size_t per = pipe_bytes(ctx->bm, cb.pz, cb.py, cb.px, 1, stages) +
                                   align256((size_t)cb.pz * cb.py * cb.px * (sizeof(float) + sizeof(uint16_t)));
                long long bmax = (long long)(((size_t)ctx->chunk_budget_mb << 20) / per);
                if (bmax < 1) bmax = 1;
                if (bmax > 65535) bmax = 65535;
                for (long long first = 0; first < nchunks; first += bmax) {
                    const int count = (int)std::min<long long>(bmax, nchunks - first);
                    cb.first = (int)first;
                    cb.count = count;
                    VolGeom g;
                    rc = make_geom(ctx, cb.pz, cb.py, cb.px, count, g);
                    if (rc) return rc;
                    const size_t n = (size_t)g.nvox * (size_t)count;
                    const size_t base = pipe_bytes(ctx->bm, cb.pz, cb.py, cb.px, count, stages);
                    const size_t fbytes = align256(n * sizeof(float));
                    rc = ensure_scratch(ctx, base + fbytes + GUARD_BYTES + align256(n * sizeof(uint16_t)));
                    if (rc) return rc;
                    char* scratch = static_cast<char*>(ctx->scratch);
                    float* vol = reinterpret_cast<float*>(scratch + base);
                    uint16_t* vol16 = reinterpret_cast<uint16_t*>(scratch + base + fbytes + GUARD_BYTES);
                    HIP_TRY(ctx, launch_chunk_gather(in, cb, offset, vol, ctx->stream, vol16));
                    rc = run_pipeline(ctx, vol, vol, nullptr, g, count, sigma, p, stages, 1.0f, 0.0f, 0.0f,
                                      scratch, 1, EXABM4D_DATA_EXP_U16, offset_exact_in_fp32(offset) ? vol16 : nullptr,
                                      1, offset);
                    if (rc) return rc;
                    HIP_TRY(ctx, launch_chunk_scatter(vol, cb, offset, out, ctx->stream));
                }
            }
    return EXABM4D_OK;
}

// ---- chunk-local mode, host volume streamed through the device ------------------------------------
// A host volume of any size (BASELINE config 4's tile is 64 GiB of uint16) goes through the device one
// LAYER of chunks at a time: planes [k chunk - halo, (k + 1) chunk + halo) up, cores down.  Two
// device windows and two result buffers; an uploader and a downloader thread (plain copies on their
// own streams: the host side of a pageable copy blocks, so each direction gets a thread) run one layer
// ahead of / behind exabm4d_denoise_chunked_u16_dev on the context's stream.  Chunks are independent
// units, so the result is the one-call result of exabm4d_denoise_chunked_u16_dev on the whole volume.
namespace {
struct StreamedLayers {
    std::mutex m;
    std::condition_variable cv;
    int uploaded = 0;       // layers whose window is on the device
    int enqueued = 0;       // layers whose compute has been enqueued (comp_ev[k & 1] recorded)
    int downloaded = 0;     // layers whose cores are back in the caller's array
    bool failed = false;
    std::string err;

    void advance(int StreamedLayers::*field) {
        { std::lock_guard<std::mutex> l(m); (this->*field)++; }
        cv.notify_all();
    }
    bool wait_for(int StreamedLayers::*field, int value) {      // false: somebody failed
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return failed || this->*field >= value; });
        return !failed;
    }
    void fail_with(const std::string& what) {
        { std::lock_guard<std::mutex> l(m); if (!failed) { failed = true; err = what; } }
        cv.notify_all();
    }
};
}  // namespace

int exabm4d_denoise_chunked_u16_host(exabm4d_ctx* ctx, const uint16_t* in, uint16_t* out, int nz, int ny,
                                     int nx, int chunk, int halo, float sigma, float offset,
                                     const exabm4d_params* p, int stages) {
    if (!ctx || !in || !out) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if (!(sigma > 0.0f)) return fail(ctx, EXABM4D_ERR_INVALID, "sigma must be > 0");
    if (stages != 1 && stages != 2) return fail(ctx, EXABM4D_ERR_INVALID, "stages must be 1 or 2");
    if (nz < 1 || ny < 1 || nx < 1 || chunk < 1 || halo < 0 || halo > 64)
        return fail(ctx, EXABM4D_ERR_INVALID, "chunked: sizes >= 1, chunk >= 1, 0 <= halo <= 64");
    {   // the downloads of early layers would overwrite planes that later layers still have to upload
        const size_t bytes = (size_t)nz * (size_t)ny * (size_t)nx * sizeof(uint16_t);
        const char *a = reinterpret_cast<const char*>(in), *b = reinterpret_cast<const char*>(out);
        if (a < b + bytes && b < a + bytes)
            return fail(ctx, EXABM4D_ERR_INVALID, "streamed chunk mode: input and output arrays overlap");
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int layers = (nz + chunk - 1) / chunk;
    const size_t plane = (size_t)ny * (size_t)nx;
    const int wmax = std::min(nz, chunk + 2 * halo), cmax = std::min(nz, chunk);
    const int device = ctx->device;

    uint16_t* win[2] = {nullptr, nullptr};
    uint16_t* res[2] = {nullptr, nullptr};
    hipStream_t s_up = nullptr, s_down = nullptr;
    hipEvent_t comp_ev[2] = {nullptr, nullptr};
    const int nbuf = layers > 1 ? 2 : 1;
    auto release = [&]() {
        for (int i = 0; i < 2; i++) {
            if (win[i]) (void)hipFree(win[i]);
            if (res[i]) (void)hipFree(res[i]);
            if (comp_ev[i]) (void)hipEventDestroy(comp_ev[i]);
        }
        if (s_up) (void)hipStreamDestroy(s_up);
        if (s_down) (void)hipStreamDestroy(s_down);
    };
    hipError_t e = hipSuccess;
    for (int i = 0; i < nbuf && e == hipSuccess; i++) {
        e = hipMalloc((void**)&win[i], (size_t)wmax * plane * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMalloc((void**)&res[i], (size_t)cmax * plane * sizeof(uint16_t));
        if (e == hipSuccess) e = hipEventCreateWithFlags(&comp_ev[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking);
    if (e != hipSuccess) {
        release();
        return fail_hip(ctx, e, "streamed chunk mode: device windows / streams");
    }

    StreamedLayers st;
    auto window_of = [&](int k, int& w0, int& w1, int& c0, int& c1) {
        c0 = k * chunk;
        c1 = std::min(nz, c0 + chunk);
        w0 = std::max(0, c0 - halo);
        w1 = std::min(nz, c1 + halo);
    };
    auto upload_layers = [&]() {
        if (hipSetDevice(device) != hipSuccess) return st.fail_with("uploader: hipSetDevice");
        for (int k = 0; k < layers; k++) {
            if (k >= 2) {       // window k & 1 was read by layer k - 2
                if (!st.wait_for(&StreamedLayers::enqueued, k - 1)) return;
                if (hipEventSynchronize(comp_ev[k & 1]) != hipSuccess) return st.fail_with("uploader: hipEventSynchronize");
            }
            int w0, w1, c0, c1;
            window_of(k, w0, w1, c0, c1);
            hipError_t r = hipMemcpyAsync(win[k & 1], in + (size_t)w0 * plane, (size_t)(w1 - w0) * plane * sizeof(uint16_t),
                                          hipMemcpyHostToDevice, s_up);
            if (r == hipSuccess) r = hipStreamSynchronize(s_up);
            if (r != hipSuccess) return st.fail_with(std::string("upload of a chunk layer: ") + hipGetErrorString(r));
            st.advance(&StreamedLayers::uploaded);
        }
    };
    auto download_layers = [&]() {
        if (hipSetDevice(device) != hipSuccess) return st.fail_with("downloader: hipSetDevice");
        for (int k = 0; k < layers; k++) {
            if (!st.wait_for(&StreamedLayers::enqueued, k + 1)) return;
            int w0, w1, c0, c1;
            window_of(k, w0, w1, c0, c1);
            hipError_t r = hipEventSynchronize(comp_ev[k & 1]);
            if (r == hipSuccess)
                r = hipMemcpyAsync(out + (size_t)c0 * plane, res[k & 1], (size_t)(c1 - c0) * plane * sizeof(uint16_t),
                                   hipMemcpyDeviceToHost, s_down);
            if (r == hipSuccess) r = hipStreamSynchronize(s_down);
            if (r != hipSuccess) return st.fail_with(std::string("download of a chunk layer: ") + hipGetErrorString(r));
            st.advance(&StreamedLayers::downloaded);
        }
    };
    // no C++ exception may cross the C boundary: a thread that cannot be started is an error code
    std::thread uploader, downloader;
    try {
        uploader = std::thread(upload_layers);
        downloader = std::thread(download_layers);
    } catch (const std::exception& ex) {
        st.fail_with(std::string("cannot start a copy thread: ") + ex.what());
        if (uploader.joinable()) uploader.join();
        release();
        return fail(ctx, EXABM4D_ERR_NOMEM, "streamed chunk mode: " + st.err);
    }

    std::string compute_err;
    for (int k = 0; k < layers; k++) {
        // the window is up; the result buffer k & 1 (layer k - 2's) has been fetched
        if (!st.wait_for(&StreamedLayers::uploaded, k + 1) || !st.wait_for(&StreamedLayers::downloaded, k - 1)) break;
        int w0, w1, c0, c1;
        window_of(k, w0, w1, c0, c1);
        rc = exabm4d_denoise_chunked_u16_dev(ctx, win[k & 1], res[k & 1], w1 - w0, ny, nx, c0 - w0, c1 - w0, chunk,
                                             halo, sigma, offset, p, stages);
        hipError_t r = rc ? hipSuccess : hipEventRecord(comp_ev[k & 1], ctx->stream);
        if (rc || r != hipSuccess) {
            compute_err = rc ? ctx->err : std::string("hipEventRecord: ") + hipGetErrorString(r);
            if (!rc) rc = EXABM4D_ERR_HIP;
            st.fail_with(compute_err);
            break;
        }
        st.advance(&StreamedLayers::enqueued);
    }
    uploader.join();
    downloader.join();
    (void)hipStreamSynchronize(ctx->stream);
    release();
    if (st.failed) return fail(ctx, rc ? rc : EXABM4D_ERR_HIP, "streamed chunk mode: " + st.err);
    return check_async_status(ctx);
}

// A large batch goes through the device in sub-batches of about 2^26 voxels, double-buffered: while sub-batch k
// is computed, the results of k - 1 come down and the input of k + 1 goes up on a copy stream of the context's
// (the host side of a pageable copy blocks, which is all the ordering this thread needs; the device side is
// ordered by events).  Every volume carries its own fixed-point unit, so the cut changes no bit.  1000 patches
// of 64^3, host to host: 245 -> see DESIGN.md 8a; the scratch is that of one sub-batch, not of the batch.
static constexpr size_t HOST_SUB_VOXELS = (size_t)1 << 26;
static int denoise_f32_host_pipelined(exabm4d_ctx* ctx, const float* in, float* out, int nz, int ny, int nx,
                                      int batch, int sub, float sigma, const exabm4d_params* p, int stages,
                                      float clip_lo, float clip_hi) {
    VolGeom g;
    int rc = make_geom(ctx, nz, ny, nx, sub, g);
    if (rc) return rc;
    const size_t nv = (size_t)g.nvox, nsubvox = nv * (size_t)sub;
    const size_t base = pipe_bytes(ctx->bm, nz, ny, nx, sub, stages);
    const size_t bufbytes = align256(nsubvox * sizeof(float)) + GUARD_BYTES;
    rc = ensure_scratch(ctx, base + 2 * bufbytes);
    if (rc) return rc;
    if (!ctx->copy_stream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 3; i++) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->copy_ev[i], hipEventDisableTiming));
    }
    char* scratch = static_cast<char*>(ctx->scratch);
    float* buf[2] = {reinterpret_cast<float*>(scratch + base), reinterpret_cast<float*>(scratch + base + bufbytes)};
    hipStream_t cs = ctx->copy_stream, s = ctx->stream;
    const int nsub = (batch + sub - 1) / sub;
    auto count_of = [&](int k) { return std::min(sub, batch - k * sub); };
    // the buffers' last users were earlier calls on the compute stream
    HIP_TRY(ctx, hipEventRecord(ctx->copy_ev[2], s));
    HIP_TRY(ctx, hipStreamWaitEvent(cs, ctx->copy_ev[2], 0));
    HIP_TRY(ctx, hipMemcpyAsync(buf[0], in, nv * count_of(0) * sizeof(float), hipMemcpyHostToDevice, cs));
    for (int k = 0; k < nsub; k++) {
        const int cnt = count_of(k);
        VolGeom gk = g;
        if (cnt != sub) {
            rc = make_geom(ctx, nz, ny, nx, cnt, gk);
            if (rc) return rc;
        }
        HIP_TRY(ctx, hipEventRecord(ctx->copy_ev[2], cs));                     // input k is up
        HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->copy_ev[2], 0));
        rc = run_pipeline(ctx, buf[k & 1], buf[k & 1], nullptr, gk, cnt, sigma, p, stages, clip_lo, clip_hi, 0.0f,
                          scratch, 1, EXABM4D_DATA_EXP_AUTO);
        if (rc) return rc;
        HIP_TRY(ctx, hipEventRecord(ctx->copy_ev[k & 1], s));                  // result k is ready
        if (k >= 1) {                                                          // result k - 1 down, under compute k
            HIP_TRY(ctx, hipStreamWaitEvent(cs, ctx->copy_ev[(k - 1) & 1], 0));
            HIP_TRY(ctx, hipMemcpyAsync(out + (size_t)(k - 1) * nsubvox, buf[(k - 1) & 1],
                                        nv * count_of(k - 1) * sizeof(float), hipMemcpyDeviceToHost, cs));
        }
        if (k + 1 < nsub)                                                      // input k + 1 up, into the buffer just emptied
            HIP_TRY(ctx, hipMemcpyAsync(buf[(k + 1) & 1], in + (size_t)(k + 1) * nsubvox,
                                        nv * count_of(k + 1) * sizeof(float), hipMemcpyHostToDevice, cs));
    }
    HIP_TRY(ctx, hipStreamWaitEvent(cs, ctx->copy_ev[(nsub - 1) & 1], 0));
    HIP_TRY(ctx, hipMemcpyAsync(out + (size_t)(nsub - 1) * nsubvox, buf[(nsub - 1) & 1],
                                nv * count_of(nsub - 1) * sizeof(float), hipMemcpyDeviceToHost, cs));
    HIP_TRY(ctx, hipStreamSynchronize(cs));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return check_async_status(ctx);
}

int exabm4d_denoise_f32_host(exabm4d_ctx* ctx, const float* in, float* out, int nz, int ny, int nx,
                             int batch, float sigma, const exabm4d_params* p, int stages,
                             float clip_lo, float clip_hi) {
    VolGeom g;
    int rc = pipeline_checks(ctx, in, out, nz, ny, nx, batch, sigma, p, stages, g);
    if (rc) return rc;
    const size_t n = (size_t)g.nvox * (size_t)batch;
    // Large batches of volumes: sub-batches with the copies under the kernels.  Not in place (a repeated run
    // must find its input), not while a debug option wants to see one launch.
    const size_t per = std::max<size_t>(1, HOST_SUB_VOXELS / (size_t)g.nvox);
    if (ctx->host_pipeline && in != out && batch >= 2 && (size_t)batch >= 2 * per && per <= 65535) {
        rc = denoise_f32_host_pipelined(ctx, in, out, nz, ny, nx, batch, (int)per, sigma, p, stages, clip_lo,
                                        clip_hi);
        if (rc != EXABM4D_OK && ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);   // nothing of it left in flight
        if (rc == EXABM4D_OK || ctx->bm.carry != 0) return rc;
        // the carry's wait ran out somewhere (it is off now): once more below, in one piece
    }
    // This call owns its input and synchronises, so a run whose carry wait ran out (check_async_status) is
    // simply repeated without the carry: the caller of bm4d(raw, sigma) sees a result, never a hang or a retry.
    for (int attempt = 0;; attempt++) {
        const size_t base = pipe_bytes(ctx->bm, nz, ny, nx, batch, stages);
        rc = ensure_scratch(ctx, base + align256(n * sizeof(float)));
        if (rc) return rc;
        char* scratch = static_cast<char*>(ctx->scratch);
        float* vol = reinterpret_cast<float*>(scratch + base);
        HIP_TRY(ctx, hipMemcpyAsync(vol, in, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        rc = run_pipeline(ctx, vol, vol, nullptr, g, batch, sigma, p, stages, clip_lo, clip_hi, 0.0f,
                          scratch, 1, EXABM4D_DATA_EXP_AUTO);
        if (rc) return rc;
        // (out may be in: the result goes to the host only once the run is known to be good)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        rc = check_async_status(ctx);
        if (rc != EXABM4D_OK) {
            if (attempt == 1 || ctx->bm.carry != 0) return rc;
            continue;
        }
        HIP_TRY(ctx, hipMemcpyAsync(out, vol, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return EXABM4D_OK;
    }
}

// The same for a batch whose volumes lie anywhere in host memory (the broker's shape: every caller's patch in
// its own shared-memory segment): in[i] / out[i] per volume, out[i] may be in[i].
int exabm4d_denoise_f32_host_v(exabm4d_ctx* ctx, const float* const* in, float* const* out, int nz, int ny,
                               int nx, int batch, float sigma, const exabm4d_params* p, int stages,
                               float clip_lo, float clip_hi) {
    VolGeom g;
    int rc = pipeline_checks(ctx, in, out, nz, ny, nx, batch, sigma, p, stages, g);
    if (rc) return rc;
    for (int b = 0; b < batch; b++)
        if (!in[b] || !out[b]) return fail(ctx, EXABM4D_ERR_INVALID, "NULL volume pointer");
    const size_t nv = (size_t)g.nvox, n = nv * (size_t)batch;
    for (int attempt = 0;; attempt++) {
        const size_t base = pipe_bytes(ctx->bm, nz, ny, nx, batch, stages);
        rc = ensure_scratch(ctx, base + align256(n * sizeof(float)));
        if (rc) return rc;
        char* scratch = static_cast<char*>(ctx->scratch);
        float* vol = reinterpret_cast<float*>(scratch + base);
        for (int b = 0; b < batch; b++)
            HIP_TRY(ctx, hipMemcpyAsync(vol + (size_t)b * nv, in[b], nv * sizeof(float), hipMemcpyHostToDevice,
                                        ctx->stream));
        rc = run_pipeline(ctx, vol, vol, nullptr, g, batch, sigma, p, stages, clip_lo, clip_hi, 0.0f,
                          scratch, 1, EXABM4D_DATA_EXP_AUTO);
        if (rc) return rc;
        // (a repeated run reads in[] again: the results go to the host only once the run is known to be good)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        rc = check_async_status(ctx);
        if (rc != EXABM4D_OK) {
            if (attempt == 1 || ctx->bm.carry != 0) return rc;
            continue;
        }
        for (int b = 0; b < batch; b++)
            HIP_TRY(ctx, hipMemcpyAsync(out[b], vol + (size_t)b * nv, nv * sizeof(float), hipMemcpyDeviceToHost,
                                        ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return EXABM4D_OK;
    }
}

// ---- BM4DNet stage: fused GroupNorm + LeakyReLU on NDHWC tensors (nn_kernels.hip) ------------------------
size_t exabm4d_groupnorm_workspace_bytes(int batch, size_t spatial, int channels, int groups) {
    if (batch < 1 || channels < 1 || groups < 1) return 0;
    return groupnorm_workspace_bytes(batch, spatial, channels, groups);
}
int exabm4d_groupnorm_lrelu_ndhwc_dev(exabm4d_ctx* ctx, void* hip_stream, const float* x, float* y, int batch,
                                      size_t spatial, int channels, int groups, const float* gamma,
                                      const float* beta, float eps, float slope, void* workspace,
                                      size_t workspace_bytes, const float* conv_bias) {
    if (!ctx || !x || !y || !workspace) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    if (conv_bias && ((uintptr_t)conv_bias & 15u) != 0)
        return fail(ctx, EXABM4D_ERR_INVALID, "groupnorm_lrelu_ndhwc: conv_bias must be 16-byte aligned");
    if (batch < 1 || batch > 65535 || spatial < 1 || channels < 4 || groups < 1 || groups > 32 ||
        channels % groups != 0 || channels % 4 != 0 || (channels / groups) % 4 != 0 || 256 % (channels / 4) != 0)
        return fail(ctx, EXABM4D_ERR_UNSUPPORTED,
                    "groupnorm_lrelu_ndhwc: needs channels % 4 == 0, (channels / groups) % 4 == 0, "
                    "256 % (channels / 4) == 0 and groups <= 32 (use the framework's GroupNorm otherwise)");
    if (workspace_bytes < groupnorm_workspace_bytes(batch, spatial, channels, groups))
        return fail(ctx, EXABM4D_ERR_INVALID, "groupnorm_lrelu_ndhwc: workspace too small");
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)workspace) & 15u) != 0)
        return fail(ctx, EXABM4D_ERR_INVALID, "groupnorm_lrelu_ndhwc: 16-byte aligned tensors expected");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_groupnorm_lrelu_ndhwc(x, y, batch, spatial, channels, groups, gamma, beta, eps, slope,
                                              workspace, (hipStream_t)hip_stream, conv_bias));
    return EXABM4D_OK;
}

static int nn_resample_checks(exabm4d_ctx* ctx, const void* x, const void* y, int batch, int d, int h, int w,
                              int channels) {
    if (!ctx || !x || !y) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    if (batch < 1 || d < 1 || h < 1 || w < 1 || channels < 4 || channels % 4 != 0)
        return fail(ctx, EXABM4D_ERR_UNSUPPORTED, "NDHWC resampling: sizes >= 1 and channels % 4 == 0");
    if ((((uintptr_t)x | (uintptr_t)y) & 15u) != 0)
        return fail(ctx, EXABM4D_ERR_INVALID, "NDHWC resampling: 16-byte aligned tensors expected");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return EXABM4D_OK;
}
int exabm4d_maxpool2_ndhwc_dev(exabm4d_ctx* ctx, void* hip_stream, const float* x, float* y, int batch, int d,
                               int h, int w, int channels) {
    int rc = nn_resample_checks(ctx, x, y, batch, d, h, w, channels);
    if (rc) return rc;
    HIP_TRY(ctx, launch_maxpool2_ndhwc(x, y, batch, d, h, w, channels, (hipStream_t)hip_stream));
    return EXABM4D_OK;
}
int exabm4d_upsample2_trilinear_ndhwc_dev(exabm4d_ctx* ctx, void* hip_stream, const float* x, float* y, int batch,
                                          int d, int h, int w, int channels) {
    int rc = nn_resample_checks(ctx, x, y, batch, d, h, w, channels);
    if (rc) return rc;
    HIP_TRY(ctx, launch_upsample2_trilinear_ndhwc(x, y, batch, d, h, w, channels, (hipStream_t)hip_stream));
    return EXABM4D_OK;
}

// Page-lock caller memory that host entry points will copy from / to many times (the broker: every worker's
// shared-memory segment, for the life of the connection): the copies then are DMA transfers instead of staged
// ones.  hipHostRegisterDefault; the mapping is per process, the registration per (pointer, size).
int exabm4d_host_register(exabm4d_ctx* ctx, void* ptr, size_t bytes) {
    if (!ctx || !ptr || !bytes) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument or empty range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return EXABM4D_OK;
}
int exabm4d_host_unregister(exabm4d_ctx* ctx, void* ptr) {
    if (!ctx || !ptr) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipHostUnregister(ptr));
    return EXABM4D_OK;
}

int exabm4d_profile_read(exabm4d_ctx* ctx, float* ms, int max_phases) {
    if (!ctx || !ms) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    if (!ctx->ev[0]) return fail(ctx, EXABM4D_ERR_INVALID, "profiling was never enabled");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int n = max_phases < EXABM4D_PHASE_COUNT ? max_phases : EXABM4D_PHASE_COUNT;
    for (int i = 0; i < n; i++) {
        ms[i] = 0.0f;
        if (ctx->ev_used[i]) HIP_TRY(ctx, hipEventElapsedTime(&ms[i], ctx->ev[2 * i], ctx->ev[2 * i + 1]));
    }
    return n;
}

// ---- intensity transforms ---------------------------------------------------------------------------------
int exabm4d_transform_forward_u16_dev(exabm4d_ctx* ctx, const exabm4d_transform* t,
                                      const uint16_t* in, float* out, size_t n) {
    if (!ctx || !in || !out) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    TfDev d;
    int rc = make_tfdev(ctx, t, d);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (d.kind == EXABM4D_TF_ASINH && n >= ((size_t)1 << 20)) {
        // asinh is evaluated in fp64: large uint16 volumes go through a 65536-entry table
        if (!ctx->tf_lut) HIP_TRY(ctx, hipMalloc((void**)&ctx->tf_lut, 65536 * sizeof(float)));
        HIP_TRY(ctx, launch_tf_forward_u16_lut(d, ctx->tf_lut, in, out, n, ctx->stream));
        return EXABM4D_OK;
    }
    HIP_TRY(ctx, launch_tf_forward_u16(d, in, out, n, ctx->stream));
    return EXABM4D_OK;
}
int exabm4d_transform_forward_f32_dev(exabm4d_ctx* ctx, const exabm4d_transform* t,
                                      const float* in, float* out, size_t n) {
    if (!ctx || !in || !out) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    TfDev d;
    int rc = make_tfdev(ctx, t, d);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_tf_forward_f32(d, in, out, n, ctx->stream));
    return EXABM4D_OK;
}
int exabm4d_transform_inverse_u16_dev(exabm4d_ctx* ctx, const exabm4d_transform* t,
                                      const float* in, uint16_t* out, size_t n) {
    if (!ctx || !in || !out) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    TfDev d;
    int rc = make_tfdev(ctx, t, d);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_tf_inverse(d, in, out, n, 1, ctx->stream));
    return EXABM4D_OK;
}
int exabm4d_transform_inverse_f32_dev(exabm4d_ctx* ctx, const exabm4d_transform* t,
                                      const float* in, float* out, size_t n) {
    if (!ctx || !in || !out) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    TfDev d;
    int rc = make_tfdev(ctx, t, d);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_tf_inverse(d, in, out, n, 0, ctx->stream));
    return EXABM4D_OK;
}

// ---- overlap tiling ---------------------------------------------------------------------------------------------
int exabm4d_tile_gather_dev(exabm4d_ctx* ctx, const float* vol, int nz, int ny, int nx,
                            const int32_t* starts, int nb, int patch, float* out) {
    if (!ctx || !vol || !starts || !out) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    if (nb < 0 || patch < 1 || nz < 1 || ny < 1 || nx < 1) return fail(ctx, EXABM4D_ERR_INVALID, "bad sizes");
    for (int i = 0; i < 3 * nb; i++)
        if (starts[i] < 0) return fail(ctx, EXABM4D_ERR_INVALID, "negative patch start");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_tile_gather(vol, nz, ny, nx, starts, nb, patch, out, ctx->stream));
    return EXABM4D_OK;
}
int exabm4d_tile_accumulate_dev(exabm4d_ctx* ctx, const float* preds, const int32_t* starts, int nb,
                                int patch, int trim, float* accum_pred, float* accum_wgt, int nz,
                                int ny, int nx) {
    if (!ctx || !preds || !starts || !accum_pred || !accum_wgt)
        return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    if (nb < 0 || patch < 1 || trim < 0 || 2 * trim >= patch) return fail(ctx, EXABM4D_ERR_INVALID, "bad sizes");
    for (int i = 0; i < 3 * nb; i++)
        if (starts[i] < 0) return fail(ctx, EXABM4D_ERR_INVALID, "negative patch start");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_tile_accumulate(preds, starts, nb, patch, trim, accum_pred, accum_wgt, nz, ny,
                                        nx, ctx->stream));
    return EXABM4D_OK;
}
int exabm4d_tile_finalize_u16_dev(exabm4d_ctx* ctx, const exabm4d_transform* t,
                                  const float* accum_pred, const float* accum_wgt, uint16_t* out,
                                  size_t n) {
    if (!ctx || !accum_pred || !accum_wgt || !out) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    TfDev d;
    int rc = make_tfdev(ctx, t, d);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_tile_finalize(d, accum_pred, accum_wgt, out, n, ctx->stream));
    return EXABM4D_OK;
}

// ---- encode front end (row f-1) ---------------------------------------------------------------------
int exabm4d_chunk_byte_histograms_dev(exabm4d_ctx* ctx, const uint16_t* vol, int nz, int ny, int nx,
                                      int cz, int cy, int cx, uint32_t* hist) {
    if (!ctx || !vol || !hist) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    if (nz < 1 || ny < 1 || nx < 1 || cz < 1 || cy < 1 || cx < 1)
        return fail(ctx, EXABM4D_ERR_INVALID, "bad sizes");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_chunk_hist(vol, nz, ny, nx, cz, cy, cx, hist, ctx->stream));
    return EXABM4D_OK;
}

// ---- transform quantiser (row f-1) ------------------------------------------------------
int exabm4d_dctq_forward_dev(exabm4d_ctx* ctx, const uint16_t* vol, int nz, int ny, int nx, float q,
                             int32_t* idx) {
    if (!ctx || !vol || !idx) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    if (nz < 1 || ny < 1 || nx < 1 || !(q > 0.0f)) return fail(ctx, EXABM4D_ERR_INVALID, "bad sizes / step");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float dct[64], win[512];
    make_tables(0.0, dct, win);            // the DCT table does not depend on the window's beta
    HIP_TRY(ctx, launch_dctq_forward(vol, nz, ny, nx, dct, q, idx, ctx->stream));
    return EXABM4D_OK;
}
int exabm4d_dctq_inverse_dev(exabm4d_ctx* ctx, const int32_t* idx, int nz, int ny, int nx, float q,
                             uint16_t* vol) {
    if (!ctx || !vol || !idx) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    if (nz < 1 || ny < 1 || nx < 1 || !(q > 0.0f)) return fail(ctx, EXABM4D_ERR_INVALID, "bad sizes / step");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float dct[64], win[512];
    make_tables(0.0, dct, win);
    HIP_TRY(ctx, launch_dctq_inverse(idx, nz, ny, nx, dct, q, vol, ctx->stream));
    return EXABM4D_OK;
}

// ---- chunk entropy coder (row f-1) --------------------------------------------------------
size_t exabm4d_codec_chunk_bound(size_t n_elems, int typesize) {
    if (typesize != 2 && typesize != 4) return 0;
    return codec_chunk_bound(n_elems, typesize);
}
size_t exabm4d_codec_volume_bound(int typesize, int nz, int ny, int nx, int cz, int cy, int cx) {
    CodecGeom g;
    if (make_codec_geom(typesize, nz, ny, nx, cz, cy, cx, g)) return 0;
    return codec_volume_bound(g);
}
// aux layout: sizes u32[nchunks] | offsets u64[nchunks + 1] | totals u64[2] | status u32[4]
static int codec_aux(exabm4d_ctx* ctx, int nchunks, uint32_t*& sizes, unsigned long long*& offsets,
                     unsigned long long*& totals, uint32_t*& status) {
    const size_t a = align256((size_t)nchunks * 4), b = align256(((size_t)nchunks + 1) * 8);
    const size_t need = a + b + 256 + 256;
    if (ctx->codec_aux_bytes < need) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->codec_aux) (void)hipFree(ctx->codec_aux);
        ctx->codec_aux = nullptr;
        ctx->codec_aux_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->codec_aux, need));
        ctx->codec_aux_bytes = need;
    }
    char* base = static_cast<char*>(ctx->codec_aux);
    sizes = reinterpret_cast<uint32_t*>(base);
    offsets = reinterpret_cast<unsigned long long*>(base + a);
    totals = reinterpret_cast<unsigned long long*>(base + a + b);
    status = reinterpret_cast<uint32_t*>(base + a + b + 256);
    return EXABM4D_OK;
}
int exabm4d_codec_encode_dev(exabm4d_ctx* ctx, const void* vol, int typesize, int version, int nz, int ny, int nx,
                             int cz, int cy, int cx, uint8_t* out, size_t out_capacity,
                             uint64_t* offsets_dev, uint32_t* sizes_dev, uint64_t* totals_host) {
    if (!ctx || !vol) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    if (version < 0 || version > 2) return fail(ctx, EXABM4D_ERR_INVALID, "codec: version must be 0 (context default), 1 or 2");
    CodecGeom g;
    if (make_codec_geom(typesize, nz, ny, nx, cz, cy, cx, g, version ? version : ctx->codec_version))
        return fail(ctx, EXABM4D_ERR_INVALID, "codec: typesize must be 2 or 4, sizes >= 1, chunk <= 2^28 elements");
    if (out && !offsets_dev) return fail(ctx, EXABM4D_ERR_INVALID, "codec: offsets_dev is required with out");
    if (out && out_capacity < codec_volume_bound(g))
        return fail(ctx, EXABM4D_ERR_INVALID, "codec: out_capacity is below exabm4d_codec_volume_bound()");
    if (out && ((uintptr_t)out & 15)) return fail(ctx, EXABM4D_ERR_INVALID, "codec: out must be 16-byte aligned");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->rcp_dev) {
        static uint32_t tab[4097 * 2];
        codec_fill_rcp_table(tab);
        HIP_TRY(ctx, hipMalloc((void**)&ctx->rcp_dev, sizeof tab));
        HIP_TRY(ctx, hipMemcpy(ctx->rcp_dev, tab, sizeof tab, hipMemcpyHostToDevice));
    }
    uint32_t *sizes, *status;
    unsigned long long *offsets, *totals;
    int rc = codec_aux(ctx, g.nchunks, sizes, offsets, totals, status);
    if (rc) return rc;
    rc = ensure_scratch(ctx, (((size_t)g.nchunks * g.slot_bytes + 255) & ~(size_t)255) +
                                 (g.version == 2 ? codec2_work_bytes(g) : 0));
    if (rc) return rc;
    if (sizes_dev) sizes = sizes_dev;
    if (offsets_dev) offsets = reinterpret_cast<unsigned long long*>(offsets_dev);
    HIP_TRY(ctx, launch_rans_encode(vol, g, ctx->rcp_dev, static_cast<uint8_t*>(ctx->scratch), sizes,
                                    offsets, totals, out, ctx->stream));
    if (totals_host) {
        HIP_TRY(ctx, hipMemcpyAsync(totals_host, totals, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost,
                                    ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return EXABM4D_OK;
}
int exabm4d_codec_decode_dev(exabm4d_ctx* ctx, const uint8_t* in, size_t in_bytes, const uint64_t* offsets_dev,
                             int typesize, int nz, int ny, int nx, int cz, int cy, int cx, void* vol) {
    if (!ctx || !in || !offsets_dev || !vol) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    if ((uintptr_t)in & 15) return fail(ctx, EXABM4D_ERR_INVALID, "codec: in must be 16-byte aligned");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the format version is the third byte of every chunk stream: look at the first one
    uint64_t first[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(first, offsets_dev, sizeof first, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (first[0] > first[1] || first[1] > in_bytes || first[1] - first[0] < 4)
        return fail(ctx, EXABM4D_ERR_INVALID, "codec: malformed chunk stream (offsets outside the buffer)");
    uint8_t magic[4] = {0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(magic, in + first[0], 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (magic[0] != 'E' || magic[1] != 'X' || (magic[2] != 1 && magic[2] != 2))
        return fail(ctx, EXABM4D_ERR_INVALID, "codec: malformed chunk stream (not an EXAC v1 / v2 stream)");
    CodecGeom g;
    if (make_codec_geom(typesize, nz, ny, nx, cz, cy, cx, g, magic[2]))
        return fail(ctx, EXABM4D_ERR_INVALID, "codec: typesize must be 2 or 4, sizes >= 1, chunk <= 2^28 elements");
    uint32_t *sizes, *status;
    unsigned long long *offsets, *totals;
    int rc = codec_aux(ctx, g.nchunks, sizes, offsets, totals, status);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemsetAsync(status, 0, 16, ctx->stream));
    HIP_TRY(ctx, launch_rans_decode(in, in_bytes, reinterpret_cast<const unsigned long long*>(offsets_dev), g,
                                    vol, status, ctx->stream));
    uint32_t st = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&st, status, sizeof st, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (st) {
        char msg[96];
        std::snprintf(msg, sizeof msg, "codec: malformed chunk stream (status 0x%x)", st);
        return fail(ctx, EXABM4D_ERR_INVALID, msg);
    }
    return EXABM4D_OK;
}

// ---- background offset + quality metrics (row f-4) ---------------------------------------------------
static int metric_scratch(exabm4d_ctx* ctx, size_t bytes) {
    if (ctx->red_bytes >= bytes) return EXABM4D_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->red) (void)hipFree(ctx->red);
    ctx->red = nullptr;
    ctx->red_bytes = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->red, bytes));
    ctx->red_bytes = bytes;
    return EXABM4D_OK;
}
static int metric_fetch(exabm4d_ctx* ctx, void* host, const void* dev, size_t bytes) {
    HIP_TRY(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return EXABM4D_OK;
}
static bool bad_dtype(int d) { return d < EXABM4D_DT_U16 || d > EXABM4D_DT_F64; }

int exabm4d_u16_histogram_dev(exabm4d_ctx* ctx, const uint16_t* vol, size_t n, uint64_t* hist_host) {
    if (!ctx || !hist_host || (!vol && n)) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = metric_scratch(ctx, 65536 * sizeof(uint64_t))) return rc;
    HIP_TRY(ctx, launch_hist_u16(vol, n, (unsigned long long*)ctx->red, ctx->stream));
    return metric_fetch(ctx, hist_host, ctx->red, 65536 * sizeof(uint64_t));
}

int exabm4d_i32_symbol_histogram_dev(exabm4d_ctx* ctx, const int32_t* idx, size_t n, uint64_t* hist_host) {
    if (!ctx || !hist_host || (!idx && n)) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = metric_scratch(ctx, 65536 * sizeof(uint64_t))) return rc;
    HIP_TRY(ctx, launch_hist_i32_clamped(idx, n, (unsigned long long*)ctx->red, ctx->stream));
    return metric_fetch(ctx, hist_host, ctx->red, 65536 * sizeof(uint64_t));
}

int exabm4d_key_histogram_dev(exabm4d_ctx* ctx, const void* vol, int dtype, size_t n, int absdev,
                              double center, int digit, uint64_t prefix, uint64_t* hist_host) {
    if (!ctx || !hist_host || (!vol && n)) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    if (bad_dtype(dtype) || digit < 0 || digit > 3) return fail(ctx, EXABM4D_ERR_INVALID, "bad dtype / digit");
    if (digit > 0 && digit < 4 && (prefix >> (16 * digit)) != 0)
        return fail(ctx, EXABM4D_ERR_INVALID, "prefix wider than the digits above");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = metric_scratch(ctx, 65536 * sizeof(uint64_t))) return rc;
    HIP_TRY(ctx, launch_hist_key(vol, dtype, n, absdev ? 1 : 0, center, digit,
                                 (unsigned long long)prefix, (unsigned long long*)ctx->red,
                                 ctx->stream));
    return metric_fetch(ctx, hist_host, ctx->red, 65536 * sizeof(uint64_t));
}

int exabm4d_minmax_dev(exabm4d_ctx* ctx, const void* vol, int dtype, size_t n, double* out_host) {
    if (!ctx || !vol || !out_host) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    if (bad_dtype(dtype) || n == 0) return fail(ctx, EXABM4D_ERR_INVALID, "bad dtype / empty input");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t np = (size_t)masked_stats_partials(n) * 2;
    if (int rc = metric_scratch(ctx, (np + 2) * sizeof(double))) return rc;
    double* d = (double*)ctx->red;
    HIP_TRY(ctx, launch_minmax(vol, dtype, n, d + 2, d, ctx->stream));
    return metric_fetch(ctx, out_host, d, 2 * sizeof(double));
}

int exabm4d_masked_error_stats_dev(exabm4d_ctx* ctx, const void* pred, int pred_dtype,
                                   const void* ref, int ref_dtype, const uint8_t* mask, size_t n,
                                   double thr, double* out_host) {
    if (!ctx || !pred || !ref || !out_host) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    if (bad_dtype(pred_dtype) || bad_dtype(ref_dtype) || n == 0)
        return fail(ctx, EXABM4D_ERR_INVALID, "bad dtype / empty input");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t np = (size_t)masked_stats_partials(n) * 7;
    if (int rc = metric_scratch(ctx, (np + 8) * sizeof(double))) return rc;
    double* d = (double*)ctx->red;
    HIP_TRY(ctx, launch_masked_stats(pred, pred_dtype, ref, ref_dtype, mask, n, thr, d + 8, d,
                                     ctx->stream));
    return metric_fetch(ctx, out_host, d, 7 * sizeof(double));
}

int exabm4d_ssim3d_dev(exabm4d_ctx* ctx, const void* a, const void* b, int dtype, int nz, int ny,
                       int nx, int window, double c1, double c2, double* sum_host) {
    if (!ctx || !a || !b || !sum_host) return fail(ctx, EXABM4D_ERR_INVALID, "NULL argument");
    if (bad_dtype(dtype) || nz < 1 || ny < 1 || nx < 1) return fail(ctx, EXABM4D_ERR_INVALID, "bad dtype / sizes");
    if (window < 1 || window > ssim3d_max_window())
        return fail(ctx, EXABM4D_ERR_UNSUPPORTED, "ssim window must be 1..32");
    if ((long long)nz * ny * nx > (1ll << 40) || nz > (1 << 20) || ny > (1 << 20) || nx > (1 << 20))
        return fail(ctx, EXABM4D_ERR_INVALID, "volume too large");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t np = (size_t)ssim3d_partials(nz, ny, nx);
    if (int rc = metric_scratch(ctx, (np + 1) * sizeof(double))) return rc;
    double* d = (double*)ctx->red;
    HIP_TRY(ctx, launch_ssim3d(a, b, dtype, nz, ny, nx, window, c1, c2, d + 1, d, ctx->stream));
    return metric_fetch(ctx, sum_host, d, sizeof(double));
}

}  // extern "C"
