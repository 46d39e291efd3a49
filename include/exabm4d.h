/*
 * exabm4d.h -- C-ABI of libexabm4d.so (MI355X / gfx950 native BM4D denoise + intensity
 * transform + overlap-tile stitching hot path for ExaSPIM uint16 volumes).
 *
 * This is the drop-in boundary for the hot path of
 * AllenNeuralDynamics/aind-exaspim-image-compression.  The reference is pure Python; the
 * arithmetic on its hot path is reached through these interfaces, each of which one entry
 * point below replaces (paths relative to the reference checkout, src/aind_exaspim_image_compression/):
 *
 *   bm4d(raw, sigma)                      machine_learning/data_handling.py:332, :926 ; evaluate.py:202
 *   np.clip(teacher, 0, max_count)        machine_learning/data_handling.py:333, :927
 *   read_counts (u16 -> f32 - offset)     machine_learning/data_handling.py:337-354
 *   IntensityTransform.forward            machine_learning/transforms.py:113-129, :244-259, :332-348, :398-401
 *   IntensityTransform.inverse[_float]    machine_learning/transforms.py:131-152, :261-285, :350-371, :403-411
 *   predict(): patch gather / pad         inference.py:153-174, :178-199, :202-226
 *   predict(): accumulate / normalise     inference.py:81-116
 *   estimate_offset (np.percentile)       machine_learning/transforms.py:414-438
 *   ssim3D, compute_mae, compute_lmax     utils/img_util.py:953-1050
 *   evaluate_example and its parts        machine_learning/metrics.py:306-424
 *   compute_cratio's codec.encode loop    utils/img_util.py:401-441 (codec: evaluate.py:40)
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, POD structs whose first field is their own sizeof
 *     (ABI evolution).  No allocation is returned across the ABI except through
 *     exabm4d_malloc/exabm4d_free (thin hipMalloc wrappers for hosts without a device
 *     allocator of their own).
 *   - Every function returns 0 on success or a negative exabm4d_status; a message is available
 *     from exabm4d_last_error().  No C++ exception crosses the boundary.
 *   - "_dev" functions take DEVICE pointers and enqueue on the context's stream without
 *     synchronising; "_host" functions take HOST pointers, copy in/out and synchronise.
 *   - Volumes are C-contiguous [z][y][x]; a batch is `batch` such volumes back to back.
 *   - A context belongs to one (process, device); create it after fork(), never before
 *     (the reference calls bm4d from forked ProcessPoolExecutor workers, scripts/precompute.py:215).
 *   - The BM4D algorithm is the specification frozen in DESIGN.md section 3 (the reference's
 *     own BM4D is the closed third-party wheel bm4d==4.2.5, uv.lock:387-400; parity with it is
 *     unpinned -- see DESIGN.md).
 */
#ifndef EXABM4D_H
#define EXABM4D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EXABM4D_VERSION 400 /* 0.4.0 (round 4): order-independent aggregation -- exabm4d_stage_dev takes data_exp and WRITES num / den, options "stage_pairs" / "stage_quads" / "fuse_den_z" are gone, the stage / block-matching options are per context; 0.3.2: + exabm4d_blockmatch_plan, option "stage_strip"; 0.3.1: + exabm4d_denoise_chunked_u16_host, options "bm_carry" / "bm_xcd_mode"; 0.3.0: EXAC v2 coder, exabm4d_codec_decode_dev takes in_bytes (round 3) */

typedef enum exabm4d_status {
    EXABM4D_OK = 0,
    EXABM4D_ERR_INVALID = -1,     /* bad argument (NULL pointer, dims < 8, bad struct size ...) */
    EXABM4D_ERR_UNSUPPORTED = -2, /* profile value outside what the kernels implement          */
    EXABM4D_ERR_HIP = -3,         /* a HIP runtime call failed (message has hipGetErrorString)  */
    EXABM4D_ERR_NOMEM = -4,       /* device scratch allocation failed                            */
    EXABM4D_ERR_NODEVICE = -5     /* no usable gfx950 device                                     */
} exabm4d_status;

/* BM4D profile.  Defaults (exabm4d_default_params) are BASELINE.json's parameter contract:
 * block 8^3, step 4, search window 11^3, groups of <= 16, 3-D DCT + Haar, lambda 2.7. */
typedef struct exabm4d_params {
    uint32_t size;        /* = sizeof(exabm4d_params)                                    */
    int32_t block;        /* cubic block edge; only 8 is implemented                      */
    int32_t step;         /* reference-block grid step; only 4                            */
    int32_t search;       /* search window edge (displacements -5..5); only 11            */
    int32_t max_group;    /* blocks per group, power of two <= 16; only 16               */
    float lambda_ht;      /* hard threshold = lambda_ht * sigma                           */
    float c_match_ht;     /* stage-1 match threshold: mean sq. diff <= c_match_ht*sigma^2 */
    float c_match_wie;    /* stage-2 match threshold (on the basic estimate)              */
    float kaiser_beta;    /* aggregation window, separable Kaiser(8, beta); 0 => all ones */
} exabm4d_params;

/* Intensity transform descriptor (machine_learning/transforms.py).  `kind` selects the base
 * transform; `wrapped` != 0 composes OffsetTransform(base, wrap_offset) around it
 * (transforms.py:374-411).  All parameters are the Python floats of the reference object;
 * the kernels round them to fp32 exactly where numpy does. */
enum { EXABM4D_TF_ASINH = 0, EXABM4D_TF_ANSCOMBE = 1, EXABM4D_TF_LINEAR = 2 };
typedef struct exabm4d_transform {
    uint32_t size;        /* = sizeof(exabm4d_transform) */
    int32_t kind;         /* EXABM4D_TF_*                */
    int32_t wrapped;      /* 0/1: OffsetTransform wrapper */
    int32_t reserved;
    double wrap_offset;   /* OffsetTransform.offset       */
    double max_count;     /* clamp of inverse(); for a wrapper: base.max_count */
    double offset;        /* asinh.offset / anscombe.offset                    */
    double scale;         /* asinh.scale                                        */
    double norm;          /* asinh._norm / anscombe._norm (python float)        */
    double gain;          /* anscombe.gain                                      */
    double read_noise;    /* anscombe.read_noise                                */
    double c_inv;         /* anscombe._c_inv (1/8 or 3/8)                       */
    double mn, mx, clip;  /* linear                                             */
} exabm4d_transform;

typedef struct exabm4d_ctx exabm4d_ctx;

/* ---- library / context ------------------------------------------------------------------ */
int exabm4d_version(void);
/* Message of the last failing call on this thread (ctx may be NULL). Never NULL. */
const char* exabm4d_last_error(const exabm4d_ctx* ctx);
/* Number of HIP devices visible (no context needed; does not initialise a device). */
int exabm4d_device_count(void);
int exabm4d_create(int device, exabm4d_ctx** out);
int exabm4d_destroy(exabm4d_ctx* ctx);
/* Enqueue on an existing hipStream_t, e.g. torch.cuda.current_stream().cuda_stream.  The handle
 * is used as given: NULL is the HIP null (legacy default) stream, which is what PyTorch's default
 * stream is.  exabm4d_reset_stream returns to the context's private non-blocking stream. */
int exabm4d_set_stream(exabm4d_ctx* ctx, void* hip_stream);
int exabm4d_reset_stream(exabm4d_ctx* ctx);
int exabm4d_sync(exabm4d_ctx* ctx);
int exabm4d_default_params(exabm4d_params* p);
/* Diagnostic switches. "force_generic_bm" = 1 routes every reference block through the
 * one-wave-per-block matching kernel (normally used only for grid points that are not a
 * multiple of 4); the parity tests use it to check the two kernels against each other.
 * "bm_guarded_copy" = 1 makes exabm4d_blockmatch_dev match on a copy of the volume inside the
 * library's scratch allocation, the way the exabm4d_denoise_* pipelines do (x-edge tiles then
 * stream their planes by LDS-DMA without clamping); for the parity tests.
 * Round 3: "codec_version" = 1 | 2 (default 2): the EXAC format exabm4d_codec_encode_dev writes (the
 * option is per context, not per call: do not switch it from two threads of one context);
 * "stage_pairvol" = 0 makes the Wiener kernel gather its two volumes separately (default 1: one
 * interleaved volume);
 * "bm_int" = 0 keeps the uint16 pipelines' stage-1 matching on the float kernel; "stage_chunks",
 * "chunk_budget_mb", "profile": z chunks of the stage kernels (0 = automatic), scratch budget of the
 * chunk-local mode, per-phase HIP events for exabm4d_profile_read.  "bm_carry" = 0 | 1 | 2 (default 1):
 * block matching's tiles hand their top cell layer to the tile above through device memory, eight
 * reference layers per tile instead of seven (1: where it saves a tile per column and the launch is large
 * enough, 2: wherever a column has two tiles; needs 744 KB of the context's scratch per tile column; tables
 * are identical).  A tile waits for the tile below it; workgroups take their tiles by ticket,
 * so the wait always ends, and it is bounded besides: should it ever run out, the kernel raises a status word
 * instead of hanging, the next synchronising call (exabm4d_sync, exabm4d_memcpy_d2h, *_host, ...) returns
 * EXABM4D_ERR_HIP, and the carry is off for that context from then on (exabm4d_denoise_f32_host repeats its
 * run without the carry by itself).  "bm_carry_fault" = 1 (debug) makes every such wait count as run out; "bm_xcd_mode" = 0 | 1 | n (default 2): workgroup order of
 * block matching; "host_pipeline" = 0 | 1 (default 1): exabm4d_denoise_f32_host cuts batches of >= 2^27 voxels
 * into sub-batches and copies under the kernels; "zero_overlap" = 0 | 1 (default 1): the 8-byte sums are zeroed on a second
 * stream of the context's, under the block matching that precedes each stage kernel, instead of in line
 * (DESIGN.md 5.3); "stage_strip" = 0 | n (default 3): tile-column order of the stage
 * kernels (0 = raster, n = strips of n tile rows; the same results, bit for bit: the sums are integers).
 * Every option belongs to the context it is set on (round 4; rounds 1-3 kept some in process globals). */
int exabm4d_set_option(exabm4d_ctx* ctx, const char* name, int value);
/* With option "profile" = 1 every exabm4d_denoise_* call brackets each of its kernel launches
 * with HIP events on the context's stream.  exabm4d_profile_read waits for the last call and
 * returns its per-phase durations in milliseconds, in EXABM4D_PHASE_* order (0 for a phase the
 * call did not run); returns the number of phases written (<= max_phases) or a negative status. */
enum {
    EXABM4D_PHASE_COUNTS_FROM_U16 = 0,
    EXABM4D_PHASE_ZERO_ACC_1 = 1,
    EXABM4D_PHASE_BLOCKMATCH_HT = 2,
    EXABM4D_PHASE_STAGE_HT = 3,
    EXABM4D_PHASE_NORMALIZE_BASIC = 4,
    EXABM4D_PHASE_ZERO_ACC_2 = 5,
    EXABM4D_PHASE_BLOCKMATCH_WIE = 6,
    EXABM4D_PHASE_STAGE_WIE = 7,
    EXABM4D_PHASE_NORMALIZE_OUT = 8,
    EXABM4D_PHASE_COUNT = 9
};
int exabm4d_profile_read(exabm4d_ctx* ctx, float* ms, int max_phases);

/* ---- device memory helpers (for ctypes hosts without torch) ------------------------------ */
int exabm4d_malloc(exabm4d_ctx* ctx, size_t bytes, void** dptr);
int exabm4d_free(exabm4d_ctx* ctx, void* dptr);
int exabm4d_memcpy_h2d(exabm4d_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int exabm4d_memcpy_d2h(exabm4d_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int exabm4d_memset(exabm4d_ctx* ctx, void* dst_dev, int value, size_t bytes);
/* HIP-event timing on the context's stream (bench.py measures kernels with these, not with
 * torch events, so the timing is on the stream the kernels are launched on). */
int exabm4d_event_create(exabm4d_ctx* ctx, void** ev);
int exabm4d_event_destroy(exabm4d_ctx* ctx, void* ev);
int exabm4d_event_record(exabm4d_ctx* ctx, void* ev);
int exabm4d_event_elapsed_ms(exabm4d_ctx* ctx, void* ev_start, void* ev_stop, float* ms);

/* ---- geometry + tables (host, no device) -------------------------------------------------- */
/* Reference-block grid along an axis of n voxels: 0,4,8,... plus n-8 when (n-8)%4 != 0. */
int exabm4d_grid_count(int n);
int exabm4d_grid_positions(int n, int32_t* pos);
/* The constant tables the kernels use: orthonormal DCT-II 8x8 (row u, col n) and the 8^3
 * aggregation window (z,y,x raster). Computed in double, rounded once to fp32. */
int exabm4d_tables(const exabm4d_params* p, float* dct64, float* win512);
/* Bytes of device scratch an exabm4d_denoise_*_dev call of this shape takes from the context (match table,
 * 64-bit numerator and corner-weight sums, basic estimate, work volumes, and -- since round 4 -- the 744 KB per
 * tile column of block matching's carry where the default options use it: 1.0 GB of 36.4 GB at 1024^3), under
 * the default options; the uint16 entry points add the fp32 + uint16 counts (6 bytes per voxel). */
size_t exabm4d_scratch_bytes(int nz, int ny, int nx, int batch, int stages);

/* The launch block matching chooses for a geometry (host logic only, no device call): plan = {tile slabs in z,
 * tile rows, tile columns, slab-order parameter (0: every XCD walks its own contiguous range), carry between
 * tiles on, flat 4 x 16 tile shape}; carry_bytes = the part of the context's scratch the
 * carry takes (0 without).  Follows ctx's options "bm_carry" / "bm_xcd_mode"; ctx == NULL: the defaults. */
int exabm4d_blockmatch_plan(const exabm4d_ctx* ctx, int nz, int ny, int nx, int batch, int32_t plan[6],
                            uint64_t* carry_bytes);

/* ---- BM4D staged entry points (parity hooks; a-B1 .. a-B6 of SURVEY.md section 8) ---------- */
/* Block matching.  keys: [batch][nref][16] uint32, nref = gz*gy*gx in (z,y,x) raster of the
 * reference grid.  key = (bits(S) & 0xFFFFF800) | code, S = block SSD (fp32),
 * code = 0 for the reference block itself else 1 + ((dz+5)*11 + (dy+5))*11 + (dx+5).
 * Sorted ascending; unused slots 0xFFFFFFFF.  c_match: threshold factor (c_match*sigma^2). */
int exabm4d_blockmatch_dev(exabm4d_ctx* ctx, const float* vol, int nz, int ny, int nx, int batch,
                           float sigma, float c_match, const exabm4d_params* p, uint32_t* keys);
/* The same on uint16 counts, the way exabm4d_denoise_u16_dev / _chunked_u16_dev match in stage 1.
 * The counts of a uint16 volume differ by exact integers whatever the offset, so block distances
 * below 2^24 are exact in fp32 and equal their integer form: where c_match * sigma^2 * 512 < 2^24
 * and nx is even the tiles run in integer arithmetic (v_pk_sub_i16 + v_dot2_i32_i16, half the
 * vector-ALU cycles); the tables are bit-identical to exabm4d_blockmatch_dev on (float)vol.
 * Option "bm_int" = 0 forces the float kernels (the parity tests compare both). */
int exabm4d_blockmatch_u16_dev(exabm4d_ctx* ctx, const uint16_t* vol, int nz, int ny, int nx, int batch,
                               float sigma, float c_match, const exabm4d_params* p, uint32_t* keys);
/* Host decode of one reference's 16 keys at grid position (rz,ry,rx) [voxels]:
 * idx[k] = linear voxel offset of the matched block corner, dist[k] = quantised S/512,
 * *count = number of valid entries (group size is the largest power of two <= count). */
int exabm4d_match_decode(const uint32_t* keys16, int rz, int ry, int rx, int ny, int nx,
                         int64_t* idx, float* dist, int* count);
/* Collaborative filtering + aggregation.  basic == NULL: hard-threshold stage on `noisy`;
 * basic != NULL: Wiener stage (groups of `noisy` and `basic` at the same positions).
 * The sums over blocks are 64-bit integers, so the result does not depend on the order
 * the GPU adds in: num[batch][nz][ny][nx] = fl32(NUM 2^(E - 43)) with NUM = sum of
 * rint(est * fl32(u win) * 2^(43 - E)), den = fl32(CW 2^-40) convolved with the separable Kaiser window,
 * CW = sum of rint(u 2^40) on block corners, u = 1 / max(N_kept, 1) resp. 1 / max(sum W^2, 1).  Both are
 * WRITTEN (round 4; rounds 1-3 added into them).  data_exp = E: EXABM4D_DATA_EXP_AUTO takes, per volume of
 * the batch, the exponent with max |noisy| < 2^E (what exabm4d_denoise_f32_* do); a caller that shards one
 * volume passes one value for all shards (the uint16 pipelines use EXABM4D_DATA_EXP_U16 = 17). */
#define EXABM4D_DATA_EXP_AUTO INT32_MIN
#define EXABM4D_DATA_EXP_U16 17
int exabm4d_stage_dev(exabm4d_ctx* ctx, const float* noisy, const float* basic,
                      const uint32_t* keys, int nz, int ny, int nx, int batch, float sigma,
                      const exabm4d_params* p, int data_exp, float* num, float* den);
/* out = num/den, then clamp to [clip_lo, clip_hi] when clip_lo <= clip_hi (np.clip,
 * data_handling.py:333); pass clip_lo > clip_hi for no clamp. */
int exabm4d_normalize_dev(exabm4d_ctx* ctx, const float* num, const float* den, float* out,
                          size_t n, float clip_lo, float clip_hi);

/* The two ends of exabm4d_denoise_u16_dev as staged calls, for callers that run the stages
 * themselves (z-slab sharding, distributed.py): out = (float)in - offset (read_counts,
 * machine_learning/data_handling.py:337-354) ... */
int exabm4d_counts_from_u16_dev(exabm4d_ctx* ctx, const uint16_t* in, float* out, size_t n, float offset);
/* ... and out = uint16(rint(clamp(num/den + offset, 0, 65535))) (np.clip of data_handling.py:333 and
 * the cast of IntensityTransform.inverse, transforms.py:168-171). */
int exabm4d_normalize_u16_dev(exabm4d_ctx* ctx, const float* num, const float* den, uint16_t* out,
                              size_t n, float offset);
/* What stage 2 of the uint16 pipelines matches on: the basic estimate as the counts a uint16
 * caller would see, out = (float)rint(clamp(in + offset, 0, 65535)) - offset; in / out may alias.  For callers
 * that run the stages themselves: exabm4d_blockmatch_dev on `out`, exabm4d_stage_dev on the unrounded `in`. */
int exabm4d_round_counts_f32_dev(exabm4d_ctx* ctx, const float* in, float* out, size_t n, float offset);

/* ---- BM4D whole pipeline ------------------------------------------------------------------- */
/* stages: 1 = hard-threshold only, 2 = hard-threshold + Wiener.  in/out may alias.
 * Working range of the fp32 forms: finite data with max |v| < 2^56 per volume.  Outside it
 * (the squares of transform coefficients leave fp32; infinities; NaNs) the kernels still run and stay inside
 * their buffers, the device raises a status bit, and the next synchronising call on the context
 * (exabm4d_denoise_f32_host itself, exabm4d_sync, exabm4d_memcpy_d2h, ...) returns EXABM4D_ERR_INVALID:
 * the results since the last synchronisation are void.  The uint16 forms are always inside. */
int exabm4d_denoise_f32_dev(exabm4d_ctx* ctx, const float* in, float* out, int nz, int ny, int nx,
                            int batch, float sigma, const exabm4d_params* p, int stages,
                            float clip_lo, float clip_hi);
/* uint16 in -> (float)in - offset -> BM4D -> + offset -> clamp [0,65535] -> rint -> uint16.
 * (read_counts + bm4d + clip + the rint/uint16 cast of IntensityTransform.inverse.)  |offset| <= 65536.
 * The result is a deterministic function of the input: two calls, the chunked / streamed forms on
 * identical padded chunks and the oracle give the same uint16 values.
 * The uint16 forms (this, the chunk-local ones) match stage 2 on the basic estimate ROUNDED TO COUNTS
 * (both matching passes run in 16-bit integer arithmetic); the fp32 forms match on the estimate
 * itself.  rint(clamp(exabm4d_denoise_f32_dev(in - offset) + offset)) is therefore a different -- equally
 * good -- result than this call's on a few per cent of the voxels. */
int exabm4d_denoise_u16_dev(exabm4d_ctx* ctx, const uint16_t* in, uint16_t* out, int nz, int ny,
                            int nx, int batch, float sigma, float offset, const exabm4d_params* p,
                            int stages);
/* Chunk-local mode (BASELINE.json config 4: "256^3 chunks with 8-voxel halo"): independent units,
 * like the reference's one-bm4d()-call-per-patch pool (scripts/precompute.py:215-228).  The core
 * planes [zc0, zc1) of the input buffer `in` [nz][ny][nx] are tiled by cubic chunks of `chunk`
 * voxels (ragged last chunks allowed), chunk grid origin (zc0, 0, 0).  Every chunk is read with
 * `halo` voxels on each side -- the read window is clamped to the buffer: no padding is invented
 * at the volume's faces; planes of `in` outside [zc0, zc1) are real neighbour data, e.g. a slab's
 * exchanged halo -- and denoised IN ISOLATION by the uint16 pipeline of exabm4d_denoise_u16_dev
 * (reference grid and search clamped to the padded chunk); only its core is written, to
 * out[(zc1 - zc0)][ny][nx].  Chunks of one shape are batched per pipeline run (scratch per batch
 * bounded by option "chunk_budget_mb", default 32768).  The oracle processes the identical padded
 * arrays (oracle/bm4d_oracle.py: bm4d_u16_chunked). */
int exabm4d_denoise_chunked_u16_dev(exabm4d_ctx* ctx, const uint16_t* in, uint16_t* out, int nz, int ny,
                                    int nx, int zc0, int zc1, int chunk, int halo, float sigma,
                                    float offset, const exabm4d_params* p, int stages);
/* Chunk-local mode on a HOST volume of any size, streamed through the device (the harness shape of
 * scripts/evaluate_bm4dnet.py:51-181 -- a whole image in, a whole image out -- at BASELINE config 4's
 * tile size, 64 GiB of uint16, which no device call above takes in one piece).  `in` and `out` are
 * host arrays [nz][ny][nx] (pageable or pinned; they may not overlap).  One LAYER of chunks at a time:
 * planes [k chunk - halo, (k + 1) chunk + halo) go up, exabm4d_denoise_chunked_u16_dev runs on them
 * on the context's stream, the cores come down; an uploader and a downloader thread inside the call
 * keep the copies of layer k + 1 and k - 1 under the kernels of layer k.  Device memory: two windows
 * of (chunk + 2 halo) planes, two results of chunk planes, plus the scratch of the device call.  The
 * result is that of exabm4d_denoise_chunked_u16_dev on the whole volume (chunks are independent).
 * Blocks until `out` is complete. */
int exabm4d_denoise_chunked_u16_host(exabm4d_ctx* ctx, const uint16_t* in, uint16_t* out, int nz, int ny,
                                     int nx, int chunk, int halo, float sigma, float offset,
                                     const exabm4d_params* p, int stages);
/* Host-pointer form of exabm4d_denoise_f32_dev (the bm4d(z, sigma) shim calls this). */
int exabm4d_denoise_f32_host(exabm4d_ctx* ctx, const float* in, float* out, int nz, int ny, int nx,
                             int batch, float sigma, const exabm4d_params* p, int stages,
                             float clip_lo, float clip_hi);
/* The same for a batch of `batch` volumes that lie anywhere in host memory: in[i] / out[i] point to volume i
 * ([nz][ny][nx] fp32; out[i] may be in[i]).  What the per-device broker calls with every worker's patch in that
 * worker's own shared-memory segment (broker.py; reference scripts/precompute.py:215-228: one bm4d(raw, sigma)
 * per 64^3 patch and worker).  Per volume the result of exabm4d_denoise_f32_host. */
int exabm4d_denoise_f32_host_v(exabm4d_ctx* ctx, const float* const* in, float* const* out, int nz, int ny,
                               int nx, int batch, float sigma, const exabm4d_params* p, int stages,
                               float clip_lo, float clip_hi);
/* ---- BM4DNet stage (reference machine_learning/unet3d.py:137-208: Conv3d -> GroupNorm(gcd(8, C)) ->
 * LeakyReLU(0.01)): GroupNorm + LeakyReLU fused, on the NDHWC layout MIOpen's fast convolutions produce and
 * consume.  x, y: fp32 [batch][spatial][channels] (a torch channels_last_3d tensor's memory; y may be x);
 * groups of channels / groups consecutive channels; gamma, beta: [channels] or NULL; statistics in fp64,
 * combined in a fixed order.  conv_bias ([channels] or NULL): the preceding convolution's bias, added on the
 * fly -- y = lrelu(GN(x + conv_bias)) -- so that it does not cost a pass of its own.
 * Runs on `hip_stream` (the framework's current stream), not on the context's.
 * Implemented for channels % 4 == 0, (channels / groups) % 4 == 0, 256 % (channels / 4) == 0 and
 * groups <= 32 (every layer of the reference's U-Net at width_multiplier 1, 2, 4); anything else returns
 * EXABM4D_ERR_UNSUPPORTED and the caller keeps the framework's own GroupNorm.  workspace: exabm4d_groupnorm_workspace_bytes() of device memory. */
size_t exabm4d_groupnorm_workspace_bytes(int batch, size_t spatial, int channels, int groups);
int exabm4d_groupnorm_lrelu_ndhwc_dev(exabm4d_ctx* ctx, void* hip_stream, const float* x, float* y, int batch,
                                      size_t spatial, int channels, int groups, const float* gamma,
                                      const float* beta, float eps, float slope, void* workspace,
                                      size_t workspace_bytes, const float* conv_bias);

/* The U-Net's resampling layers on NDHWC fp32 tensors (reference unet3d.py:211-255 MaxPool3d(2), :258-342
 * Upsample(scale_factor=2, mode="trilinear", align_corners=True)): x[batch][d][h][w][channels] ->
 * y[batch][d/2][h/2][w/2][channels] (floor) resp. y[batch][2d][2h][2w][channels]; channels % 4 == 0. */
int exabm4d_maxpool2_ndhwc_dev(exabm4d_ctx* ctx, void* hip_stream, const float* x, float* y, int batch, int d,
                               int h, int w, int channels);
int exabm4d_upsample2_trilinear_ndhwc_dev(exabm4d_ctx* ctx, void* hip_stream, const float* x, float* y, int batch,
                                          int d, int h, int w, int channels);

/* Page-lock `bytes` of caller memory at `ptr` that the host entry points will copy from / to repeatedly (the
 * broker registers every worker's shared-memory segment once): copies become DMA transfers instead of staged
 * ones.  Unregister before the memory is unmapped. */
int exabm4d_host_register(exabm4d_ctx* ctx, void* ptr, size_t bytes);
int exabm4d_host_unregister(exabm4d_ctx* ctx, void* ptr);

/* ---- multi-GPU: halo exchange over RCCL (north_star "RCCL over xGMI only for halo
 * exchange at chunk borders") ------------------------------------------------------------------------------
 * One process per GPU.  librccl.so is dlopen()ed on first use (EXABM4D_RCCL_LIB names another copy, e.g.
 * the one a PyTorch in the same process ships); a host without RCCL gets EXABM4D_ERR_UNSUPPORTED.
 * Rank 0 draws the id (ncclGetUniqueId) and the host layer hands it to the other ranks (distributed.py does it
 * over MASTER_ADDR / MASTER_PORT without torch); exabm4d_comm_create is collective over the ranks
 * (ncclCommInitRank on the context's device). */
#define EXABM4D_COMM_ID_BYTES 128
typedef struct exabm4d_comm exabm4d_comm;
int exabm4d_comm_unique_id(uint8_t id[EXABM4D_COMM_ID_BYTES]);
int exabm4d_comm_create(exabm4d_ctx* ctx, int nranks, int rank, const uint8_t id[EXABM4D_COMM_ID_BYTES],
                        exabm4d_comm** out);
int exabm4d_comm_destroy(exabm4d_comm* comm);
/* The exchange of SURVEY.md 8e on the context's stream, no host synchronisation: ncclGroupStart; send
 * `send_lo` to rank lo_peer and receive `recv_lo` from it (bytes_lo each way); the same with hi_peer;
 * ncclGroupEnd.  A peer of -1 (or 0 bytes) skips that side (the first and the last slab).  All pointers are
 * device pointers; the planes of a z-slab are contiguous, so callers pass addresses inside their slab
 * buffers.  Replaces the torch.distributed isend / irecv pairs of distributed.HaloExchange. */
int exabm4d_halo_exchange_dev(exabm4d_ctx* ctx, exabm4d_comm* comm, int lo_peer, const void* send_lo, void* recv_lo,
                              size_t bytes_lo, int hi_peer, const void* send_hi, void* recv_hi, size_t bytes_hi);

/* *value = max over the ranks of *value (ncclAllReduce of one double on the context's stream, then a
 * synchronisation): the barrier + MAX that brackets a timed region (bench.py) without torch.distributed. */
int exabm4d_comm_max_f64_host(exabm4d_ctx* ctx, exabm4d_comm* comm, double* value);

/* ---- intensity transforms (a-D, a-E) ------------------------------------------------------- */
int exabm4d_transform_forward_u16_dev(exabm4d_ctx* ctx, const exabm4d_transform* t,
                                      const uint16_t* in, float* out, size_t n);
int exabm4d_transform_forward_f32_dev(exabm4d_ctx* ctx, const exabm4d_transform* t,
                                      const float* in, float* out, size_t n);
int exabm4d_transform_inverse_u16_dev(exabm4d_ctx* ctx, const exabm4d_transform* t,
                                      const float* in, uint16_t* out, size_t n);
int exabm4d_transform_inverse_f32_dev(exabm4d_ctx* ctx, const exabm4d_transform* t,
                                      const float* in, float* out, size_t n);

/* ---- overlap-tile stitching around the BM4DNet stage (a-F, a-G) ----------------------------- */
/* `starts` is a HOST array [nb][3] of patch corners (z,y,x); all other pointers are device
 * pointers.  Gather nb cubic patches of edge `patch` from vol [nz][ny][nx] into
 * out[nb][patch^3]; voxels beyond the volume are zero (add_padding, inference.py:178-199). */
int exabm4d_tile_gather_dev(exabm4d_ctx* ctx, const float* vol, int nz, int ny, int nx,
                            const int32_t* starts, int nb, int patch, float* out);
/* accum_pred[s:e] += pred[trim:-trim][: e-s]; accum_wgt[s:e] += 1 with s = start+trim,
 * e = min(s + patch-2*trim, dim) per axis, patch after patch in array order so that the fp32
 * sums are bit-identical to the reference's (inference.py:89-103). */
int exabm4d_tile_accumulate_dev(exabm4d_ctx* ctx, const float* preds, const int32_t* starts,
                                int nb, int patch, int trim, float* accum_pred, float* accum_wgt,
                                int nz, int ny, int nx);
/* out = transform.inverse(accum_pred / (accum_wgt + 1e-8f)) (inference.py:113-116). */
int exabm4d_tile_finalize_u16_dev(exabm4d_ctx* ctx, const exabm4d_transform* t,
                                  const float* accum_pred, const float* accum_wgt, uint16_t* out,
                                  size_t n);

/* ---- encode half of the metric (SURVEY.md section 8 "next" row f-1) ---------------------------- */
/* Per-chunk byte-plane histograms of a uint16 volume: the front end of the reference's
 * chunked compression ratio, compute_cratio(img, Blosc(zstd, SHUFFLE), patch_shape=(64,64,64))
 * (utils/img_util.py:401-441; codec evaluate.py:40).  Blosc's SHUFFLE filter regroups a chunk's
 * bytes into a low-byte plane and a high-byte plane before the entropy coder; hist receives
 * [nchunks][2][256] uint32 counts (plane 0 = low bytes), chunks in (z,y,x) raster order of
 * ceil(n/c) chunks per axis, edge chunks truncated like numpy slicing.  From these the host
 * derives a zeroth-order entropy bound of the shuffled stream (a rate proxy; the exact Blosc/zstd
 * byte counts need the third-party codec). */
int exabm4d_chunk_byte_histograms_dev(exabm4d_ctx* ctx, const uint16_t* vol, int nz, int ny, int nx,
                                      int cz, int cy, int cx, uint32_t* hist);

/* Transform quantiser of the encode half (BASELINE.json config 5 "3D wavelet/DCT quantise"; the
 * reference has no such step -- it hands the denoised uint16 volume to Blosc / JPEG-XL,
 * evaluate.py:40 -- so the specification is this repo's): non-overlapping 8^3
 * blocks (edge voxels replicated), orthonormal 3-D DCT with the arithmetic of the BM4D
 * transforms, idx = (int32) rintf(c / q) (clamped to +-2^30).  idx receives ceil(nz/8) *
 * ceil(ny/8) * ceil(nx/8) blocks of 512 coefficients (block raster, then (uz, uy, ux) raster).
 * The inverse dequantises, inverts, clamps to [0, 65535] and rounds half to even.  Indices are
 * bit-exact against the oracle (orc_dctq_forward). */
int exabm4d_dctq_forward_dev(exabm4d_ctx* ctx, const uint16_t* vol, int nz, int ny, int nx, float q,
                             int32_t* idx);
int exabm4d_dctq_inverse_dev(exabm4d_ctx* ctx, const int32_t* idx, int nz, int ny, int nx, float q,
                             uint16_t* vol);
/* Symbol histogram of n quantisation indices for an escape-coded rate estimate: bin v + 32768
 * for -32767 <= v <= 32767, bin 0 (the escape symbol) for everything else.  hist_host[65536]. */
int exabm4d_i32_symbol_histogram_dev(exabm4d_ctx* ctx, const int32_t* idx, size_t n,
                                     uint64_t* hist_host);

/* Chunk entropy coder (BASELINE.json config 5 "entropy encode").  Replaces the
 * arithmetic behind `len(codec.encode(chunk))` in compute_cratio(img, codec, patch_shape=(64,64,64))
 * (utils/img_util.py:401-441: C-order chunks, edge chunks truncated) with the codec the reference
 * builds at evaluate.py:40 / scripts/evaluate_bm4dnet.py:140 -- Blosc(zstd, SHUFFLE).  zstd is
 * third-party and absent, so the coder is this repo's; oracle/exac_codec.c states both formats and the
 * kernels' bytes are bit-identical to it:
 *   EXAC v2 (default): every element predicted from the voxel above and the voxel in the plane
 *     before, zigzag residual -> 64-symbol alphabet + raw mantissa bits, 16 static tables per chunk
 *     selected by the neighbours' residual magnitudes, 64 interleaved rANS states (the lanes of one
 *     wave).  5.0 : 1 on the denoised bench volume where byte shuffle + zstd-5 gives 3.9 : 1.
 *   EXAC v1 (exabm4d_set_option("codec_version", 1)): byte shuffle + static order-0 rANS per byte
 *     plane (round 2's format; still decoded).
 * typesize 2 = uint16 volumes, 4 = int32 quantisation indices (exabm4d_dctq_forward_dev; coded
 * without prediction), mapped to unsigned by (v << 1) ^ (v >> 31).  version = 1 | 2 selects the format per
 * CALL (round 4: two codecs of different versions may share a context from two threads); 0 = the context's
 * "codec_version" option (default 2).
 *
 * One call codes every chunk of a volume.  out (device, may be NULL: sizes only) receives the chunk
 * streams back to back, each starting at a multiple of 16 bytes (padding zeroed); out_capacity must
 * be >= exabm4d_codec_volume_bound() (which covers either format).  offsets_dev[nchunks + 1] (device;
 * may be NULL when out is NULL) receives the start of every chunk's stream and the container length;
 * sizes_dev[nchunks] (device, may be NULL) the exact stream lengths, i.e. len(codec.encode(chunk));
 * totals_host[2] (host, may be NULL; non-NULL makes the call synchronise) = { sum of the exact
 * lengths, container bytes }.  Chunks are numbered in (z, y, x) raster order. */
size_t exabm4d_codec_chunk_bound(size_t n_elems, int typesize);
size_t exabm4d_codec_volume_bound(int typesize, int nz, int ny, int nx, int cz, int cy, int cx);
int exabm4d_codec_encode_dev(exabm4d_ctx* ctx, const void* vol, int typesize, int version, int nz, int ny, int nx,
                             int cz, int cy, int cx, uint8_t* out, size_t out_capacity,
                             uint64_t* offsets_dev, uint32_t* sizes_dev, uint64_t* totals_host);
/* Inverse: in (in_bytes bytes on the device) + offsets_dev as produced above (or assembled by a host
 * from stored streams) -> vol[nz][ny][nx] of `typesize`-byte elements; the format version is read
 * from the first chunk's header.  Synchronises.  A malformed container -- offsets that are not
 * ascending, not 2-byte aligned or beyond in_bytes, bad magic, wrong element count or chunk shape,
 * truncated tables or words, symbols outside the alphabet -- gives EXABM4D_ERR_INVALID and never
 * reads outside [in, in + in_bytes). */
int exabm4d_codec_decode_dev(exabm4d_ctx* ctx, const uint8_t* in, size_t in_bytes, const uint64_t* offsets_dev,
                             int typesize, int nz, int ny, int nx, int cz, int cy, int cx, void* vol);

/* ---- background offset + quality metrics on device (SURVEY.md section 8 "next" row f-4) --------- */
/* Element types of the metric entry points. */
enum { EXABM4D_DT_U16 = 0, EXABM4D_DT_F32 = 1, EXABM4D_DT_F64 = 2 };

/* Exact histogram of a uint16 volume resident in HBM into hist_host[65536] (host memory; the
 * call synchronises the context's stream).  It carries every order statistic the reference takes
 * with numpy on the host: estimate_offset's np.percentile of the non-zero counts
 * (machine_learning/transforms.py:414-438), the offset / all-voxel offset / median / zero
 * fraction of scripts/estimate_background_offsets.py:31-67, the median, MAD and top percentile
 * of machine_learning/metrics.py:352-424. */
int exabm4d_u16_histogram_dev(exabm4d_ctx* ctx, const uint16_t* vol, size_t n, uint64_t* hist_host);

/* One 16-bit digit of the order-preserving 64-bit key of the values widened to fp64
 * (key = bits | 2^63 for non-negative values, ~bits for negative ones), or of their absolute
 * deviation |v - center| when absdev != 0: the building block of an exact radix selection for
 * data a 65536-bin histogram cannot hold -- np.percentile of float predictions and
 * np.median(|raw - median|) in machine_learning/metrics.py:375-377,413-415.  Pass `digit`
 * (0 = most significant .. 3) counts that digit of the elements whose higher digits equal
 * `prefix`.  hist_host[65536]. */
int exabm4d_key_histogram_dev(exabm4d_ctx* ctx, const void* vol, int dtype, size_t n, int absdev,
                              double center, int digit, uint64_t prefix, uint64_t* hist_host);

/* min and max of n elements -> out_host[2] (data range of ssim3D, utils/img_util.py:985-988). */
int exabm4d_minmax_dev(exabm4d_ctx* ctx, const void* vol, int dtype, size_t n, double* out_host);

/* Absolute-error statistics split by a foreground mask (uint8, non-zero = foreground; NULL = all
 * background): out_host[7] = { sum |pred-ref| over foreground, the same over background,
 * foreground voxels, background voxels with pred > thr, max pred, max ref, max |pred-ref| }.  Replaces the
 * numpy passes of foreground_background_mae / false_bright_rate / mip_max_error
 * (machine_learning/metrics.py:306-381) and compute_mae / compute_lmax's reductions
 * (utils/img_util.py).  Sums are fp64; integer-valued inputs give exact results. */
int exabm4d_masked_error_stats_dev(exabm4d_ctx* ctx, const void* pred, int pred_dtype,
                                   const void* ref, int ref_dtype, const uint8_t* mask, size_t n,
                                   double thr, double* out_host);

/* Sum over all voxels of the SSIM map of two volumes of the same element type, cubic uniform
 * window of `window` voxels (1..32; window i - window/2 .. i + window - window/2 - 1 per axis,
 * scipy.ndimage.uniform_filter's "reflect" boundary), constants c1 = (0.01 L)^2, c2 = (0.03 L)^2
 * supplied by the caller: ssim3D of utils/img_util.py:953-1003 is *sum_host / (nz ny nx). */
int exabm4d_ssim3d_dev(exabm4d_ctx* ctx, const void* a, const void* b, int dtype, int nz, int ny,
                       int nx, int window, double c1, double c2, double* sum_host);

#ifdef __cplusplus
}
#endif
#endif /* EXABM4D_H */
