// Micro-benchmark: LDS cycles per wave-instruction of the stage kernels' REAL access patterns (gfx950) --
// the ring's ds_add_u64 with its plane / row strides (with and without the ring's wrap inside a block), and
// the transposes' ds_write_b64 in both index orders, plain and with the round-4 row swap (dct_pairs.h).
// hipcc -O3 --offload-arch=gfx950 tools/dbg/lds_pattern_bench.hip -o tools/dbg/lds_pattern_bench
// Run it alone for the timings, or under `rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS`
// for the conflict cycles per pattern (one kernel per pattern: read the counters per kernel name).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int PS = 680, REG = 30, TSI = 80, TSJ = 10;     // stage_kernels.hip / dct_pairs.h
typedef float f2 __attribute__((ext_vector_type(2)));
// MODE 0: ring add, no wrap   1: ring add, the block straddles the ring's wrap (18 planes; a wrap of 22 planes
//      shifts the banks by the same 17 x 16 = 21 x 16 = 48 (mod 64) dwords)   3: [hi][r][lo] b64 stores (rounds 1-3)
//      4: [hi][r][lo] b64 stores with the row swap   5: [r][lo][hi] b64 stores   6: the b128 reads
namespace exabm4d {       // (tools/pmc_summary.py keeps kernels of this namespace)
template <int MODE>
__global__ __launch_bounds__(256) void pat(unsigned long long* out, int iters) {
    __shared__ __align__(16) unsigned long long lds[18 * PS + 1024];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, hi = lane >> 3, lo = lane & 7;
    for (int i = t; i < 18 * PS + 1024; i += 256) lds[i] = 0;
    __syncthreads();
    f2* tb = reinterpret_cast<f2*>(lds) + wave * 648;
    const int slot = ((MODE == 0 ? 2 : 18 - 4 + wave) + hi) % 18;       // MODE 1: planes 14+w .. 17, 0 .. : the wrap inside the block
    const int base = slot * PS + 3 * REG + 5 + lo;
    const int s4 = 4 * (hi & 1);
    f2* up = tb + hi * TSI + lo + s4 * TSJ;                 // the two bases of tr_store_a (dct_pairs.h)
    f2* dn = tb + hi * TSI + lo - s4 * TSJ;
    unsigned long long acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int y = 0; y < 8; y++) {
            if (MODE <= 1) __hip_atomic_fetch_add(lds + base + y * REG, 3ull + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 3) tb[hi * TSI + y * TSJ + lo] = (f2)((float)it);
            if (MODE == 4) (y < 4 ? up : dn)[y * TSJ] = (f2)((float)it);
            if (MODE == 5) tb[y * TSI + lo * TSJ + hi] = (f2)((float)it);
        }
        if (MODE == 6) {
            const float4* q = reinterpret_cast<const float4*>(tb + hi * TSI + lo * TSJ);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float4 v = q[i];
                acc += (unsigned long long)__float_as_uint(v.x + v.y + v.z + v.w);
            }
        }
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    unsigned long long t1 = __builtin_readcyclecounter();
    if (t == 0) out[blockIdx.x] = t1 - t0;
    if (t == 1) out[gridDim.x + blockIdx.x] = lds[5] + acc;
}
}  // namespace exabm4d
using exabm4d::pat;
template <int MODE>
void run(const char* name, unsigned long long* d, int per_iter) {
    const int iters = 4000, blocks = 256;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(pat<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
        (void)hipDeviceSynchronize();
    }
    unsigned long long h[256];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < 256; i++) avg += (double)h[i];
    avg /= 256;
    printf("%-58s %7.2f counter ticks per wave-instruction (4 waves on the CU)\n", name, avg / ((double)per_iter * iters * 4));
}
int main() {
    unsigned long long* d;
    (void)hipMalloc(&d, 4096 * 8);
    run<0>("ring ds_add_u64, block inside the ring", d, 8);
    run<1>("ring ds_add_u64, block straddles the wrap of 18 planes", d, 8);
    run<3>("transpose store [hi][r][lo] (b64), rounds 1-3", d, 8);
    run<4>("transpose store [hi][r^4s][lo] (b64), round 4 swap", d, 8);
    run<5>("transpose store [r][lo][hi] (b64)", d, 8);
    run<6>("transpose read b128 x 4", d, 4);
    return 0;
}
