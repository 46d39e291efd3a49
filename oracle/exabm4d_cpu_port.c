/*
 * exabm4d_cpu_port.c -- the CPU baseline that bench.py times beside the MI355X path
 * (cpu_baseline.kind = "port").  TEST / MEASUREMENT INFRASTRUCTURE ONLY: only tests/ and bench.py's
 * cpu_baseline leg may build, link, load or call this file; the product never does.
 *
 * Why a port and not the reference: the reference reaches BM4D only through the closed third-party
 * wheel bm4d==4.2.5 (machine_learning/data_handling.py:332, :926; evaluate.py:202), which cannot
 * travel to the GPU box (PARITY UNPINNED, see exabm4d_oracle.c).  The checker in exabm4d_oracle.c
 * is written to be obviously correct -- direct 1331 x 512 block distances, qsort, serial scatter --
 * and is a strawman as a speed baseline.  This file restates the SAME specification
 * the way one would write it for a many-core host:
 *   - block matching shares the 4^3 cell sums between the 8 reference blocks that contain a cell
 *     (what the HIP kernel does), with the 11 dx candidates of a cell as one SIMD vector, a rolling
 *     window of two cell layers, and insertion into a sorted 16-entry list instead of qsort;
 *   - the collaborative filtering scatters in parallel: reference positions are coloured by
 *     (iz mod 5, iy mod 5) -- blocks of groups five grid steps apart cannot overlap -- and the
 *     25 colours run one after the other, every colour fully parallel.
 * Match tables are bit-identical to the oracle's (same fmaf chains, same tree, same keys), and since
 * round 4 -- the aggregation sums are integers, DESIGN.md 3.8 -- so is everything else
 * (tests/test_oracle_bm4d.py compares both bit for bit).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BLK 8
#define BVOX 512
#define STEP 4
#define RAD 5
#define SWIN 11
#define NCAND 1331
#define MAXG 16
#define KEY_EMPTY 0xFFFFFFFFu

int orc_grid_count(int n);
void orc_grid_positions(int n, int32_t* pos);
void orc_tables(double beta, float* dct64, float* win512);
uint32_t orc_keymax(float sigma, float c_match);
void orc_code_to_disp(uint32_t code, int* dz, int* dy, int* dx);
float orc_block_ssd(const float* a, const float* b, size_t sy, size_t sz);
void orc_group_fwd(const float* D, float* g, int K);
void orc_group_inv(const float* D, float* g, int K);
void orc_gather_block(const float* vol, size_t sy, size_t sz, int z, int y, int x, float* dst);
void orc_normalize(const float* num, const float* den, float* out, size_t n, float clip_lo, float clip_hi);
float orc_rcp_nr(float d);
float orc_weight_ht(int nnz);
float orc_weight_wiener(uint64_t q);
int orc_data_exp(const float* vol, size_t n);
void orc_den_from_corners(const int64_t* CW, int nz, int ny, int nx, double beta, float* den);
void orc_num_to_float(const int64_t* NUM, size_t n, int data_exp, float* num);
#define NUM_FRAC 43
#define CW_FRAC 40
#define RINT_MAGIC 6755399441055744.0
static inline int64_t magic_bits(double m) {
    int64_t b;
    memcpy(&b, &m, 8);
    return b - 0x4338000000000000LL;
}

static inline uint32_t f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline void list_insert(uint32_t* list, uint32_t key) {
    if (key >= list[MAXG - 1]) return;
    int k = MAXG - 1;
    while (k > 0 && list[k - 1] > key) {
        list[k] = list[k - 1];
        k--;
    }
    list[k] = key;
}

/* cell sums of cell layer c: T[(cy * ncx + cx) * NCAND + cand]; candidates whose cell leaves the
 * volume are left untouched (a block inside the volume never uses them) */
static void cell_layer(const float* vol, int nz, int ny, int nx, int c, int ncy, int ncx, float* T) {
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
#pragma omp parallel for collapse(2) schedule(static)
    for (int cy = 0; cy < ncy; cy++)
        for (int cx = 0; cx < ncx; cx++) {
            float* out = T + ((size_t)cy * ncx + cx) * NCAND;
            const int z0 = 4 * c, y0 = 4 * cy, x0 = 4 * cx;
            const float* a0 = vol + (size_t)z0 * sz + (size_t)y0 * sy + x0;
            const int dlo = x0 - RAD >= 0 ? 0 : RAD - x0;                       /* first valid dx index */
            const int dhi = x0 + RAD + 3 <= nx - 1 ? SWIN : nx - 4 - x0 + RAD + 1; /* one past the last */
            for (int dz = -RAD; dz <= RAD; dz++) {
                if (z0 + dz < 0 || z0 + dz + 3 > nz - 1) continue;
                for (int dy = -RAD; dy <= RAD; dy++) {
                    if (y0 + dy < 0 || y0 + dy + 3 > ny - 1) continue;
                    float acc[SWIN];
                    for (int d = 0; d < SWIN; d++) acc[d] = 0.0f;
                    const float* w0 = a0 + (ptrdiff_t)dz * (ptrdiff_t)sz + (ptrdiff_t)dy * (ptrdiff_t)sy - RAD;
                    if (dlo == 0 && dhi == SWIN) {
                        for (int z = 0; z < 4; z++)
                            for (int y = 0; y < 4; y++)
                                for (int x = 0; x < 4; x++) {
                                    const float a = a0[z * sz + y * sy + x];
                                    const float* w = w0 + z * sz + y * sy + x;
#pragma omp simd
                                    for (int d = 0; d < SWIN; d++) {
                                        const float t = a - w[d];
                                        acc[d] = fmaf(t, t, acc[d]);
                                    }
                                }
                    } else {
                        for (int z = 0; z < 4; z++)
                            for (int y = 0; y < 4; y++)
                                for (int x = 0; x < 4; x++) {
                                    const float a = a0[z * sz + y * sy + x];
                                    const float* w = w0 + z * sz + y * sy + x;
                                    for (int d = dlo; d < dhi; d++) {
                                        const float t = a - w[d];
                                        acc[d] = fmaf(t, t, acc[d]);
                                    }
                                }
                    }
                    float* o = out + ((dz + RAD) * SWIN + (dy + RAD)) * SWIN;
                    for (int d = dlo; d < dhi; d++) o[d] = acc[d];
                }
            }
        }
}

void cpu_blockmatch(const float* vol, int nz, int ny, int nx, float sigma, float c_match, uint32_t* keys) {
    const int gz = orc_grid_count(nz), gy = orc_grid_count(ny), gx = orc_grid_count(nx);
    const int az = (nz - BLK) / STEP + 1, ay = (ny - BLK) / STEP + 1, ax = (nx - BLK) / STEP + 1;
    const int ncy = ay + 1, ncx = ax + 1;
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
    const uint32_t keymax = orc_keymax(sigma, c_match);
    const size_t layer_floats = (size_t)ncy * ncx * NCAND;
    float* T[2] = {malloc(layer_floats * sizeof(float)), malloc(layer_floats * sizeof(float))};
    cell_layer(vol, nz, ny, nx, 0, ncy, ncx, T[0]);
    for (int iz = 0; iz < az; iz++) {
        const float* lo = T[iz & 1];
        float* hi = T[(iz + 1) & 1];
        cell_layer(vol, nz, ny, nx, iz + 1, ncy, ncx, hi);
        const int rz = STEP * iz;
#pragma omp parallel for collapse(2) schedule(static)
        for (int iy = 0; iy < ay; iy++)
            for (int ix = 0; ix < ax; ix++) {
                const int ry = STEP * iy, rx = STEP * ix;
                const float* c00 = lo + ((size_t)iy * ncx + ix) * NCAND;
                const float* c01 = c00 + NCAND;
                const float* c10 = c00 + (size_t)ncx * NCAND;
                const float* c11 = c10 + NCAND;
                const float* d00 = hi + ((size_t)iy * ncx + ix) * NCAND;
                const float* d01 = d00 + NCAND;
                const float* d10 = d00 + (size_t)ncx * NCAND;
                const float* d11 = d10 + NCAND;
                uint32_t list[MAXG];
                for (int k = 0; k < MAXG; k++) list[k] = KEY_EMPTY;
                for (int dz = -RAD; dz <= RAD; dz++) {
                    if (rz + dz < 0 || rz + dz > nz - BLK) continue;
                    for (int dy = -RAD; dy <= RAD; dy++) {
                        if (ry + dy < 0 || ry + dy > ny - BLK) continue;
                        const int base = ((dz + RAD) * SWIN + (dy + RAD)) * SWIN;
                        for (int dx = -RAD; dx <= RAD; dx++) {
                            if (rx + dx < 0 || rx + dx > nx - BLK) continue;
                            const int i = base + dx + RAD;
                            const float s = ((c00[i] + c01[i]) + (c10[i] + c11[i])) +
                                            ((d00[i] + d01[i]) + (d10[i] + d11[i]));
                            const uint32_t code = (dz | dy | dx) ? 1u + (uint32_t)i : 0u;
                            const uint32_t key = (f2u(s) & 0xFFFFF800u) | code;
                            if (key < keymax) list_insert(list, key);
                        }
                    }
                }
                if (list[0] == KEY_EMPTY) list[0] = 0u;      /* never empty: DESIGN.md 3.4 */
                memcpy(keys + ((size_t)((size_t)iz * gy + iy) * gx + ix) * MAXG, list, sizeof list);
            }
    }
    free(T[0]);
    free(T[1]);
    if (gz == az && gy == ay && gx == ax) return;
    /* clamped last grid positions (extent - 8 not a multiple of 4): direct distances */
    int32_t* pz = malloc(sizeof(int32_t) * (size_t)gz);
    int32_t* py = malloc(sizeof(int32_t) * (size_t)gy);
    int32_t* px = malloc(sizeof(int32_t) * (size_t)gx);
    orc_grid_positions(nz, pz);
    orc_grid_positions(ny, py);
    orc_grid_positions(nx, px);
    const long nref = (long)gz * gy * gx;
#pragma omp parallel for schedule(dynamic, 8)
    for (long r = 0; r < nref; r++) {
        const int ix = (int)(r % gx), iy = (int)((r / gx) % gy), iz = (int)(r / ((long)gx * gy));
        if (iz < az && iy < ay && ix < ax) continue;
        const int rz = pz[iz], ry = py[iy], rx = px[ix];
        const float* a = vol + rz * sz + ry * sy + rx;
        uint32_t list[MAXG];
        for (int k = 0; k < MAXG; k++) list[k] = KEY_EMPTY;
        for (int dz = -RAD; dz <= RAD; dz++) {
            if (rz + dz < 0 || rz + dz > nz - BLK) continue;
            for (int dy = -RAD; dy <= RAD; dy++) {
                if (ry + dy < 0 || ry + dy > ny - BLK) continue;
                for (int dx = -RAD; dx <= RAD; dx++) {
                    if (rx + dx < 0 || rx + dx > nx - BLK) continue;
                    const float s = orc_block_ssd(a, vol + (rz + dz) * sz + (ry + dy) * sy + (rx + dx), sy, sz);
                    const uint32_t code =
                        (dz | dy | dx) ? 1u + (uint32_t)(((dz + RAD) * SWIN + (dy + RAD)) * SWIN + (dx + RAD)) : 0u;
                    const uint32_t key = (f2u(s) & 0xFFFFF800u) | code;
                    if (key < keymax) list_insert(list, key);
                }
            }
        }
        if (list[0] == KEY_EMPTY) list[0] = 0u;
        memcpy(keys + (size_t)r * MAXG, list, sizeof list);
    }
    free(pz);
    free(py);
    free(px);
}

/* ---- 4-D transforms, SIMD across lines (same chains per element as the oracle's dct8_fwd / dct8_inv /
 * haar_fwd / haar_inv: bit-identical results, tests/test_oracle_bm4d.py) -------------------------- */
#define HAAR_C 0.70710678118654752440f
/* 8-point DCT along an axis of element stride `es` for `nl` lines that are contiguous in memory */
static inline void dct_lines_fwd(const float* D, float* v, size_t es, int nl) {
    const float c = D[0], a = D[2 * 8 + 0], b = D[2 * 8 + 1];
#pragma omp simd
    for (int l = 0; l < nl; l++) {
        float s[4], d[4], o[8];
        for (int n = 0; n < 4; n++) {
            s[n] = v[n * es + l] + v[(7 - n) * es + l];
            d[n] = v[n * es + l] - v[(7 - n) * es + l];
        }
        const float ss0 = s[0] + s[3], ss1 = s[1] + s[2], sd0 = s[0] - s[3], sd1 = s[1] - s[2];
        o[0] = c * (ss0 + ss1);
        o[4] = c * (ss0 - ss1);
        o[2] = fmaf(b, sd1, a * sd0);
        o[6] = fmaf(-a, sd1, b * sd0);
        for (int u = 1; u < 8; u += 2) {
            const float* k = D + u * 8;
            float t = k[0] * d[0];
            t = fmaf(k[1], d[1], t);
            t = fmaf(k[2], d[2], t);
            t = fmaf(k[3], d[3], t);
            o[u] = t;
        }
        for (int u = 0; u < 8; u++) v[u * es + l] = o[u];
    }
}
static inline void dct_lines_inv(const float* D, float* v, size_t es, int nl) {
    const float c = D[0], a = D[2 * 8 + 0], b = D[2 * 8 + 1];
#pragma omp simd
    for (int l = 0; l < nl; l++) {
        float k[8], x[8];
        for (int u = 0; u < 8; u++) k[u] = v[u * es + l];
        const float p0 = c * (k[0] + k[4]), p1 = c * (k[0] - k[4]);
        const float q0 = fmaf(b, k[6], a * k[2]), q1 = fmaf(-a, k[6], b * k[2]);
        const float e[4] = {p0 + q0, p1 + q1, p1 - q1, p0 - q0};
        for (int n = 0; n < 4; n++) {
            float o = D[1 * 8 + n] * k[1];
            o = fmaf(D[3 * 8 + n], k[3], o);
            o = fmaf(D[5 * 8 + n], k[5], o);
            o = fmaf(D[7 * 8 + n], k[7], o);
            x[n] = e[n] + o;
            x[7 - n] = e[n] - o;
        }
        for (int n = 0; n < 8; n++) v[n * es + l] = x[n];
    }
}
static inline void transpose_yx(float* b) {           /* every z-slice [y][x] -> [x][y] */
    for (int z = 0; z < 8; z++) {
        float* p = b + z * 64;
        for (int y = 0; y < 8; y++)
            for (int x = y + 1; x < 8; x++) {
                const float t = p[y * 8 + x];
                p[y * 8 + x] = p[x * 8 + y];
                p[x * 8 + y] = t;
            }
    }
}
/* axis order: forward y, x, z then Haar along the group; inverse the other way */
static void port_group_fwd(const float* D, float* g, int K) {
    for (int k = 0; k < K; k++) {
        float* b = g + (size_t)k * BVOX;
        for (int z = 0; z < 8; z++) dct_lines_fwd(D, b + z * 64, 8, 8);     /* y: lines (z, x) */
        transpose_yx(b);
        for (int z = 0; z < 8; z++) dct_lines_fwd(D, b + z * 64, 8, 8);     /* x */
        transpose_yx(b);
        dct_lines_fwd(D, b, 64, 64);                                         /* z: lines (y, x) */
    }
    for (int len = K; len > 1; len >>= 1) {
        const int half = len >> 1;
        float t[MAXG / 2 * 2][8];
        for (int i0 = 0; i0 < BVOX; i0 += 8) {
            for (int j = 0; j < half; j++)
#pragma omp simd
                for (int i = 0; i < 8; i++) {
                    const float a = g[(size_t)(2 * j) * BVOX + i0 + i], b = g[(size_t)(2 * j + 1) * BVOX + i0 + i];
                    t[j][i] = (a + b) * HAAR_C;
                    t[half + j][i] = (a - b) * HAAR_C;
                }
            for (int j = 0; j < len; j++)
                for (int i = 0; i < 8; i++) g[(size_t)j * BVOX + i0 + i] = t[j][i];
        }
    }
}
static void port_group_inv(const float* D, float* g, int K) {
    for (int len = 1; len < K; len <<= 1) {
        float t[MAXG][8];
        for (int i0 = 0; i0 < BVOX; i0 += 8) {
            for (int j = 0; j < len; j++)
#pragma omp simd
                for (int i = 0; i < 8; i++) {
                    const float a = g[(size_t)j * BVOX + i0 + i], d = g[(size_t)(len + j) * BVOX + i0 + i];
                    t[2 * j][i] = (a + d) * HAAR_C;
                    t[2 * j + 1][i] = (a - d) * HAAR_C;
                }
            for (int j = 0; j < 2 * len; j++)
                for (int i = 0; i < 8; i++) g[(size_t)j * BVOX + i0 + i] = t[j][i];
        }
    }
    for (int k = 0; k < K; k++) {
        float* b = g + (size_t)k * BVOX;
        dct_lines_inv(D, b, 64, 64);                                         /* z */
        transpose_yx(b);
        for (int z = 0; z < 8; z++) dct_lines_inv(D, b + z * 64, 8, 8);     /* x */
        transpose_yx(b);
        for (int z = 0; z < 8; z++) dct_lines_inv(D, b + z * 64, 8, 8);     /* y */
    }
}
/* parity hook against orc_group_transform */
void cpu_group_transform(float* g, int K, int inverse) {
    float D[64], win[BVOX];
    orc_tables(0.0, D, win);
    if (inverse)
        port_group_inv(D, g, K);
    else
        port_group_fwd(D, g, K);
}

/* one group: transform, shrink, inverse, scatter (same arithmetic as orc_stage_q: DESIGN.md 3.6-3.8) */
static void one_group(const float* noisy, const float* basic, const uint32_t* kk, int rz, int ry, int rx,
                      size_t sy, size_t sz, const float* D, const float* win, float thr, float sigma2,
                      double up, int64_t* NUM, int64_t* CW) {
    int count = 0;
    while (count < MAXG && kk[count] != KEY_EMPTY) count++;
    int K = 1;
    while (K * 2 <= count) K *= 2;
    float g[MAXG * BVOX], gb[MAXG * BVOX];
    int dz[MAXG], dy[MAXG], dx[MAXG];
    for (int k = 0; k < K; k++) {
        orc_code_to_disp(kk[k] & 0x7FFu, &dz[k], &dy[k], &dx[k]);
        orc_gather_block(noisy, sy, sz, rz + dz[k], ry + dy[k], rx + dx[k], g + (size_t)k * BVOX);
        if (basic) orc_gather_block(basic, sy, sz, rz + dz[k], ry + dy[k], rx + dx[k], gb + (size_t)k * BVOX);
    }
    port_group_fwd(D, g, K);
    float w;
    if (!basic) {
        int nnz = 0;
        for (int i = 0; i < K * BVOX; i++) {
            if (fabsf(g[i]) >= thr)
                nnz++;
            else
                g[i] = 0.0f;
        }
        w = orc_weight_ht(nnz);
    } else {
        port_group_fwd(D, gb, K);
        uint64_t q = 0;
        for (int i = 0; i < K * BVOX; i++) {
            const float e = gb[i] * gb[i];
            const float d = e + sigma2;
            /* R(d) of DESIGN.md 3.7, inline so that the loop vectorises */
            uint32_t rb;
            memcpy(&rb, &d, 4);
            rb = 0x7EF311C7u - rb;
            float r;
            memcpy(&r, &rb, 4);
            for (int it = 0; it < 3; it++) {
                const float t = fmaf(-d, r, 1.0f);
                r = fmaf(t, r, r);
            }
            const float W = e * r;
            g[i] = W * g[i];
            const float t1 = fmaf(W, W, 1.0f);
            uint32_t qb;
            memcpy(&qb, &t1, 4);
            q += qb - 0x3F800000u;
        }
        w = orc_weight_wiener(q);
    }
    port_group_inv(D, g, K);
    const int64_t U = magic_bits(fma((double)w, ldexp(1.0, CW_FRAC), RINT_MAGIC));
    for (int k = 0; k < K; k++) {
        const size_t base = (size_t)(rz + dz[k]) * sz + (size_t)(ry + dy[k]) * sy + (size_t)(rx + dx[k]);
        CW[base] += U;
        for (int bz = 0; bz < 8; bz++)
            for (int by = 0; by < 8; by++) {
                int64_t* pn = NUM + base + bz * sz + by * sy;
                const float* e = g + (size_t)k * BVOX + (bz * 8 + by) * 8;
                const float* wn = win + (bz * 8 + by) * 8;
                for (int bx = 0; bx < 8; bx++) {
                    const float ww = w * wn[bx];
                    pn[bx] += magic_bits(fma((double)e[bx], (double)ww * up, RINT_MAGIC));
                }
            }
    }
}

void cpu_stage_q(const float* noisy, const float* basic, const uint32_t* keys, int nz, int ny, int nx,
                 float sigma, float lambda_ht, double beta, int data_exp, int64_t* NUM, int64_t* CW) {
    const int gz = orc_grid_count(nz), gy = orc_grid_count(ny), gx = orc_grid_count(nx);
    int32_t* pz = malloc(sizeof(int32_t) * (size_t)gz);
    int32_t* py = malloc(sizeof(int32_t) * (size_t)gy);
    int32_t* px = malloc(sizeof(int32_t) * (size_t)gx);
    orc_grid_positions(nz, pz);
    orc_grid_positions(ny, py);
    orc_grid_positions(nx, px);
    float D[64], win[BVOX];
    orc_tables(beta, D, win);
    const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
    const float thr = (float)((double)lambda_ht * (double)sigma);
    const float sigma2 = (float)((double)sigma * (double)sigma);
    const double up = ldexp(1.0, NUM_FRAC - data_exp);
    /* blocks of a group reach 5 voxels in front of and 12 behind the reference corner: references
     * five grid steps (>= 17 voxels; the clamped last position is closer, so it gets a colour of
     * its own) apart in z or in y never touch the same voxel.  (The sums are integers, so the
     * colouring only keeps the threads off each other's voxels; it no longer fixes an order.) */
    const int CZ = 6, CY = 6;
    for (int cz = 0; cz < CZ; cz++)
        for (int cy = 0; cy < CY; cy++) {
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
            for (int iz = 0; iz < gz; iz++)
                for (int iy = 0; iy < gy; iy++) {
                    const int colz = (iz == gz - 1 && pz[iz] % STEP) ? 5 : iz % 5;
                    const int coly = (iy == gy - 1 && py[iy] % STEP) ? 5 : iy % 5;
                    if (colz != cz || coly != cy) continue;
                    for (int ix = 0; ix < gx; ix++)
                        one_group(noisy, basic, keys + ((size_t)((size_t)iz * gy + iy) * gx + ix) * MAXG, pz[iz],
                                  py[iy], px[ix], sy, sz, D, win, thr, sigma2, up, NUM, CW);
                }
        }
    free(pz);
    free(py);
    free(px);
}

/* the staged form: num = fl32(NUM 2^(E - 43)), den = C (*) window, both written (like orc_stage) */
void cpu_stage(const float* noisy, const float* basic, const uint32_t* keys, int nz, int ny, int nx,
               float sigma, float lambda_ht, double beta, int data_exp, float* num, float* den) {
    const size_t n = (size_t)nz * ny * nx;
    if (data_exp == INT32_MIN) data_exp = orc_data_exp(noisy, n);
    int64_t* NUM = calloc(n, sizeof(int64_t));
    int64_t* CW = calloc(n, sizeof(int64_t));
    cpu_stage_q(noisy, basic, keys, nz, ny, nx, sigma, lambda_ht, beta, data_exp, NUM, CW);
    orc_num_to_float(NUM, n, data_exp, num);
    orc_den_from_corners(CW, nz, ny, nx, beta, den);
    free(NUM);
    free(CW);
}

static void cpu_bm4d_e(const float* in, float* out, int nz, int ny, int nx, float sigma, float lambda_ht,
                       float c_match_ht, float c_match_wie, double beta, int stages, float clip_lo,
                       float clip_hi, int data_exp, int match_counts, float offset) {
    const size_t n = (size_t)nz * ny * nx;
    const long nref = (long)orc_grid_count(nz) * orc_grid_count(ny) * orc_grid_count(nx);
    uint32_t* keys = malloc(sizeof(uint32_t) * (size_t)nref * MAXG);
    float* num = malloc(n * sizeof(float));
    float* den = malloc(n * sizeof(float));
    if (data_exp == INT32_MIN) data_exp = orc_data_exp(in, n);
    cpu_blockmatch(in, nz, ny, nx, sigma, c_match_ht, keys);
    cpu_stage(in, NULL, keys, nz, ny, nx, sigma, lambda_ht, beta, data_exp, num, den);
    if (stages < 2) {
        orc_normalize(num, den, out, n, clip_lo, clip_hi);
    } else {
        float* basic = malloc(sizeof(float) * n);
        orc_normalize(num, den, basic, n, 1.0f, 0.0f);
        if (match_counts) {                    /* the uint16 form: stage 2 matches on counts */
            float* m = malloc(sizeof(float) * n);
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < n; i++) {
                float v = basic[i] + offset;
                v = v < 0.0f ? 0.0f : v;
                v = v > 65535.0f ? 65535.0f : v;
                m[i] = rintf(v) - offset;
            }
            cpu_blockmatch(m, nz, ny, nx, sigma, c_match_wie, keys);
            free(m);
        } else
            cpu_blockmatch(basic, nz, ny, nx, sigma, c_match_wie, keys);
        cpu_stage(in, basic, keys, nz, ny, nx, sigma, lambda_ht, beta, data_exp, num, den);
        orc_normalize(num, den, out, n, clip_lo, clip_hi);
        free(basic);
    }
    free(keys);
    free(num);
    free(den);
}
void cpu_bm4d(const float* in, float* out, int nz, int ny, int nx, float sigma, float lambda_ht,
              float c_match_ht, float c_match_wie, double beta, int stages, float clip_lo, float clip_hi) {
    cpu_bm4d_e(in, out, nz, ny, nx, sigma, lambda_ht, c_match_ht, c_match_wie, beta, stages, clip_lo, clip_hi,
               INT32_MIN, 0, 0.0f);
}

void cpu_bm4d_u16(const uint16_t* in, uint16_t* out, int nz, int ny, int nx, float sigma, float offset,
                  float lambda_ht, float c_match_ht, float c_match_wie, double beta, int stages) {
    const size_t n = (size_t)nz * ny * nx;
    float* f = malloc(sizeof(float) * n);
    float* g = malloc(sizeof(float) * n);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) f[i] = (float)in[i] - offset;
    cpu_bm4d_e(f, g, nz, ny, nx, sigma, lambda_ht, c_match_ht, c_match_wie, beta, stages, 1.0f, 0.0f, 17, 1, offset);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        float v = g[i] + offset;
        v = v < 0.0f ? 0.0f : v;
        v = v > 65535.0f ? 65535.0f : v;
        out[i] = (uint16_t)rintf(v);
    }
    free(f);
    free(g);
}
