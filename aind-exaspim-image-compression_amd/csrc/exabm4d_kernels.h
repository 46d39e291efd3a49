// exabm4d_kernels.h -- host-side launcher interface between exabm4d_api.hip and *_kernels.hip.
#pragma once
#include "exabm4d_common.h"

namespace exabm4d {

// Intensity-transform constants, already rounded to fp32 exactly where numpy rounds the
// reference object's Python floats (machine_learning/transforms.py).
struct TfDev {
    int kind;     // 0 asinh, 1 anscombe, 2 linear
    int wrapped;  // OffsetTransform wrapper
    float woff;   // wrapper offset
    float maxc;   // max_count
    float off, scale, norm;                           // asinh (off/norm shared with anscombe)
    float gain, c38g2, rn2, two_over_gain, cinvg2;    // anscombe
    float mn, fden, clip, range;                      // linear: mn, mx-mn+1e-8, clip, mx-mn
};

hipError_t launch_normalize(const float* num, const float* den, float* out, size_t n, float lo,
                            float hi, hipStream_t s);
hipError_t launch_counts_from_u16(const uint16_t* in, float* out, size_t n, float offset,
                                  hipStream_t s, uint16_t* out16 = nullptr);
// out_u16x != NULL: counts XOR 0x8000 (integer block matching); else out_f32 = counts - offset.  DESIGN.md 3.9.
hipError_t launch_round_counts(const float* in, float* out_f32, uint16_t* out_u16x, size_t n, float offset,
                               hipStream_t s);
hipError_t launch_normalize_u16(const float* num, const float* den, uint16_t* out, size_t n,
                                float offset, hipStream_t s);
hipError_t launch_tf_forward_u16(const TfDev& t, const uint16_t* in, float* out, size_t n,
                                 hipStream_t s);
hipError_t launch_tf_forward_u16_lut(const TfDev& t, float* lut, const uint16_t* in, float* out,
                                     size_t n, hipStream_t s);
hipError_t launch_tf_forward_f32(const TfDev& t, const float* in, float* out, size_t n,
                                 hipStream_t s);
hipError_t launch_tf_inverse(const TfDev& t, const float* in, void* out, size_t n, int quant,
                             hipStream_t s);
hipError_t launch_tile_gather(const float* vol, int nz, int ny, int nx, const int* starts, int nb,
                              int patch, float* out, hipStream_t s);
hipError_t launch_tile_accumulate(const float* preds, const int* starts, int nb, int patch, int trim,
                                  float* acc, float* wgt, int nz, int ny, int nx, hipStream_t s);
hipError_t launch_tile_finalize(const TfDev& t, const float* acc, const float* wgt, uint16_t* out,
                                size_t n, hipStream_t s);
hipError_t launch_chunk_hist(const uint16_t* vol, int nz, int ny, int nx, int cz, int cy, int cx,
                             uint32_t* hist, hipStream_t s);
hipError_t launch_dctq_forward(const uint16_t* vol, int nz, int ny, int nx, const float* dct64, float q,
                               int32_t* idx, hipStream_t s);
hipError_t launch_dctq_inverse(const int32_t* idx, int nz, int ny, int nx, const float* dct64, float q,
                               uint16_t* vol, hipStream_t s);
hipError_t launch_hist_i32_clamped(const int32_t* idx, size_t n, unsigned long long* hist, hipStream_t s);
hipError_t launch_hist_u16(const uint16_t* vol, size_t n, unsigned long long* hist, hipStream_t s);
hipError_t launch_hist_key(const void* vol, int dtype, size_t n, int absdev, double center, int digit,
                           unsigned long long prefix, unsigned long long* hist, hipStream_t s);
int masked_stats_partials(size_t n);
hipError_t launch_masked_stats(const void* pred, int pred_dtype, const void* ref, int ref_dtype,
                               const uint8_t* mask, size_t n, double thr, double* partials,
                               double* out7, hipStream_t s);
hipError_t launch_minmax(const void* a, int dtype, size_t n, double* partials, double* out2,
                         hipStream_t s);
int ssim3d_partials(int nz, int ny, int nx);
int ssim3d_max_window();
hipError_t launch_ssim3d(const void* a, const void* b, int dtype, int nz, int ny, int nx, int w,
                         double C1, double C2, double* partials, double* out1, hipStream_t s);
// Options of block matching (per context since round 4; exabm4d_set_option "bm_xcd_mode" / "bm_carry" / "bm_carry_fault")
struct BmOpts {
    int xcd_mode = 2;       // workgroup order: 0 = contiguous per XCD, 1 = all XCDs in one z slab of tiles (raster), n >= 2 = in strips of n tile rows
    int carry = 1;          // tiles of a column hand their top cell layer upwards (0 off, 1 automatic, 2 forced)
    int carry_fault = 0;    // debug: every carry wait counts as run out (the error path's test)
};
// The launch block matching chooses for a geometry (bm_kernels.hip: bm_plan), evaluated once per launch
struct BmPlan {
    int tz, ty, tx;         // tile slabs, tile rows, tile columns
    int xq;                 // slab-order parameter (0: every XCD walks its own contiguous range)
    int carry;              // carry between the tiles of a column on
    int flat;               // 4 x 16 tile shape instead of 8 x 8
    int strip, fault;
    size_t carry_bytes;     // device memory the launch needs for the carry (slots + done[] + ticket), 0 without
};
BmPlan bm_plan(const VolGeom& g, int batch, const BmOpts& opt);
hipError_t launch_blockmatch(const float* vol, const VolGeom& g, int batch, uint32_t keymax,
                             uint32_t* keys, hipStream_t stream, int force_generic, int guarded,
                             const uint16_t* vol16, const BmPlan& plan, void* carry_mem, unsigned* status);
// Options of the stage kernels (per context since round 4; exabm4d_set_option "stage_pairvol" / "stage_strip" /
// "stage_chunks"): Wiener gathers from an interleaved (noisy, basic) volume; tile columns walked in strips of n
// tile rows (0 = raster); diagnostic override of the z chunk count (0 = automatic).
struct StageOpts {
    int pairvol = 1;
    int strip = 3;
    int chunks = 0;
};
// One stage: adds to num (int64 fixed point) and to the corner weights cw; see stage_kernels.hip.
hipError_t launch_stage(const float* noisy, const float* basic, const uint32_t* keys,
                        const VolGeom& g, int batch, const float* dct64, const float* win_dev,
                        float thr, float sigma2, const double* qscale, long long* num,
                        unsigned long long* cw, hipStream_t stream, const StageOpts& opt,
                        float* pair = nullptr, int pair_ready = 0);
// The numerator's unit per volume: qscale[2 b] = 2^(43 - E), qscale[2 b + 1] = 2^(E - 43).
// fixed_exp != INT32_MIN: E = fixed_exp for every volume (the uint16 entry points: 17); else E from the
// largest |v| bit pattern of volume b of `vol` (maxbits: `batch` words of scratch).  No host synchronisation.
hipError_t launch_qscale(const float* vol, size_t nvox, int batch, int fixed_exp, unsigned* maxbits,
                         double* qscale, hipStream_t s, unsigned* status = nullptr);
// den = fl32(cw 2^-40) (*) win for the separable window win = k (x) k (x) k: fused x / y pass cw -> tmp,
// z pass tmp -> den (written).
hipError_t launch_den_from_corners(const unsigned long long* cw, float* tmp, float* den, int nz, int ny, int nx,
                                   int batch, const float* win1d, hipStream_t s);
// The pipelines' form: only the x / y passes (cw -> tmp); the z pass rides with the normalisation:
// out = fl32(fl64(num) 2^(E - 43)) / (tmp (*)_z win), then clip (f32) or + offset, clamp, rint (uint16).
hipError_t launch_den_xy_from_corners(const unsigned long long* cw, float* tmp, int nz, int ny, int nx, int batch,
                                      const float* win1d, hipStream_t s);
// pair_src / pair_out (optional, fp32 output only): also write the interleaved (pair_src, out) volume of the
// Wiener stage's gathers; *pair_written says whether this launch could do it (16-byte aligned float4 form)
hipError_t launch_normalize_zconv(const long long* num, const double* qscale, const float* txy, float* out_f32,
                                  uint16_t* out_u16, int nz, int ny, int nx, int batch, const float* win1d,
                                  float lo, float hi, float offset, hipStream_t s, const float* pair_src = nullptr,
                                  float* pair_out = nullptr, int* pair_written = nullptr,
                                  // match16 (optional, unclipped fp32 output only): also the estimate rounded to
                                  // counts XOR 0x8000, rint(clamp(out + match_offset, 0, 65535))
                                  uint16_t* match16 = nullptr, float match_offset = 0.0f,
                                  int* match_written = nullptr);
// BM4DNet stage: GroupNorm + LeakyReLU on an NDHWC tensor x[batch][spatial][C] (nn_kernels.hip); y may be x.
// Requires C % 4 == 0, (C / G) % 4 == 0, 256 % (C / 4) == 0, G <= 32.
size_t groupnorm_workspace_bytes(int batch, size_t spatial, int C, int G);
hipError_t launch_groupnorm_lrelu_ndhwc(const float* x, float* y, int batch, size_t spatial, int C, int G,
                                        const float* gamma, const float* beta, float eps, float slope,
                                        void* workspace, hipStream_t s, const float* cbias = nullptr);
// MaxPool3d(2) (floor) and trilinear x2 up-sampling (align_corners) on NDHWC fp32 tensors, C % 4 == 0
hipError_t launch_maxpool2_ndhwc(const float* x, float* y, int batch, int D, int H, int W, int C, hipStream_t s);
hipError_t launch_upsample2_trilinear_ndhwc(const float* x, float* y, int batch, int D, int H, int W, int C,
                                            hipStream_t s);
// staged entry point: num_f = fl32(fl64(num) 2^(E - 43))
hipError_t launch_num_to_float(const long long* num, const double* qscale, float* out, size_t nvox, int batch,
                               hipStream_t s);

// ---- chunk-local mode (elementwise_kernels.hip) ------------------------------------------------------
// One batch of equally shaped padded chunks out of a sub-grid of chunks (sgz x sgy x sgx chunks
// whose first core starts at (z0, y0, x0)); chunk number first + b of the sub-grid is batch entry b.
struct ChunkBatch {
    int nz, ny, nx;          // input buffer
    int z0, y0, x0;          // core origin of sub-grid chunk (0, 0, 0)
    int cz, cy, cx;          // core pitch of the chunk grid
    int ez, ey, ex;          // core extent of the chunks of this sub-grid (<= pitch)
    int pz, py, px;          // padded extent = halo in front (cut at the buffer) + core + halo behind
    int lz, ly, lx;          // voxels in front of the core inside the padded chunk
    int sgy, sgx;            // sub-grid chunks along y, x
    int first, count;        // batch = sub-grid chunks [first, first + count)
    int out_z0;              // output plane 0 is input plane out_z0
};
hipError_t launch_chunk_gather(const uint16_t* in, const ChunkBatch& cb, float offset, float* out,
                               hipStream_t s, uint16_t* out16 = nullptr);
hipError_t launch_chunk_scatter(const float* est, const ChunkBatch& cb, float offset, uint16_t* out,
                                hipStream_t s);

// ---- chunk entropy coder (rans_kernels.hip) ----------------------------------------
struct CodecGeom {
    int ts;                  // element bytes: 2 (uint16) or 4 (int32, zigzag mapped)
    int nz, ny, nx;          // volume, elements
    int cz, cy, cx;          // chunk shape (clamped to the volume)
    int gz, gy, gx;          // chunks per axis
    int nchunks;
    size_t slot_hdr;         // scratch slot of one chunk: header + tables ...
    size_t slot_plane;       // ... then `ts` stream regions of this many bytes
    size_t slot_bytes;
    size_t chunk_elems;      // cz * cy * cx: stride of a chunk in the v2 encoder's code scratch
    int version;             // stream format: 1 = byte planes, 2 = predictive context model (3.11b)
};
int make_codec_geom(int ts, int nz, int ny, int nx, int cz, int cy, int cx, CodecGeom& g, int version = 2);
size_t codec_chunk_bound(size_t n, int ts);
size_t codec_volume_bound(const CodecGeom& g);
void codec_fill_rcp_table(uint32_t* tab /* [4097][2]: reciprocal, shift */);
// sizes[nchunks], offsets[nchunks + 1], totals[2] are device arrays; out == nullptr skips the packing
// slots: nchunks * g.slot_bytes of scratch, followed (v2) by codec2_work_bytes(g) more
hipError_t launch_rans_encode(const void* vol, const CodecGeom& g, const uint32_t* rcp_tab, uint8_t* slots,
                              uint32_t* sizes, unsigned long long* offsets, unsigned long long* totals,
                              uint8_t* out, hipStream_t s);
hipError_t launch_rans_decode(const uint8_t* in, size_t in_bytes, const unsigned long long* offsets,
                              const CodecGeom& g, void* vol, uint32_t* status, hipStream_t s);
// EXAC v2 (rans2_kernels.hip); stage 0: code every chunk into its slot, stage 1: pack the slots
size_t codec2_chunk_bound(size_t n, int ts);
void codec2_slot_layout(size_t chunk_elems, int ts, size_t& slot_hdr, size_t& slot_bytes);
size_t codec2_work_bytes(const CodecGeom& g);      // encoder scratch behind the slots: codes + histograms
hipError_t launch_rans2_encode(const void* vol, const CodecGeom& g, const uint32_t* rcp_tab, uint8_t* slots,
                               uint8_t* work, uint32_t* sizes, uint8_t* out, const unsigned long long* offsets,
                               int stage, hipStream_t s);
hipError_t launch_rans2_decode(const uint8_t* in, size_t in_bytes, const unsigned long long* offsets,
                               const CodecGeom& g, void* vol, uint32_t* status, hipStream_t s);

}  // namespace exabm4d
