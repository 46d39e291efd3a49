// VALU issue / pipeline rates on gfx950: cycles per wave-instruction for a few opcodes at 1, 2, 4
// waves per SIMD (one workgroup per CU, all CUs).  hipcc -O3 --offload-arch=gfx950 valu_rate_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096, UN = 16;

template <int OP>
__global__ void k(unsigned* out, unsigned seed) {
    unsigned a[UN], b = seed + threadIdx.x;
    f2 fa[UN];
    for (int i = 0; i < UN; i++) { a[i] = seed * (i + 1) + threadIdx.x; fa[i] = (f2)((float)a[i]); }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < UN; i++) {
            if (OP == 0) { asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(fa[i].x) : "v"(b)); }
            if (OP == 1) { asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(fa[i]) : "v"(fa[(i + 1) % UN])); }
            if (OP == 2) { asm volatile("v_pk_sub_i16 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(b)); }
            if (OP == 3) { asm volatile("v_dot2_i32_i16 %0, %1, %1, %0 clamp" : "+v"(a[i]) : "v"(b)); }
            if (OP == 4) { asm volatile("v_dot2_f32_f16 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b)); }
            if (OP == 5) { asm volatile("v_mad_i32_i16 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b)); }
            if (OP == 6) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(fa[i]) : "v"(fa[(i + 1) % UN])); }
            if (OP == 7) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }
            if (OP == 8) { asm volatile("v_dot4_i32_i8 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b)); }
            if (OP == 9) { asm volatile("v_pk_mad_u16 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b)); }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
    for (int i = 0; i < UN; i++) s += a[i] + (unsigned)fa[i].x + (unsigned)fa[i].y;
    // span of the whole workgroup: first start to last end (the oldest wave alone runs at full speed)
    __shared__ unsigned long long tmin, tmax;
    if (threadIdx.x == 0) { tmin = ~0ull; tmax = 0; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { atomicMin(&tmin, (unsigned long long)t0); atomicMax(&tmax, (unsigned long long)t1); }
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (unsigned)(tmax - tmin); }
    if (s == 0x12345678u) out[1] = s;
}
template <int OP>
void run(const char* name, unsigned* d) {
    for (int waves : {1, 2, 4}) {
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * 4 * waves), 0, 0, d, 7u);
        hipDeviceSynchronize();
        unsigned h[2];
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        // memtime ticks at 100 MHz on gfx950? report ticks per instruction and let the ratios speak
        printf("%-16s waves/SIMD %d: %7.3f cycles per instruction and SIMD\n", name, waves,
               h[0] / (double)(ITER * UN) / waves);
    }
}
int main() {
    unsigned* d;
    hipMalloc(&d, 64);
    run<0>("v_fma_f32", d);
    run<1>("v_pk_fma_f32", d);
    run<6>("v_pk_add_f32", d);
    run<2>("v_pk_sub_i16", d);
    run<3>("v_dot2_i32_i16", d);
    run<4>("v_dot2_f32_f16", d);
    run<5>("v_mad_i32_i16", d);
    run<8>("v_dot4_i32_i8", d);
    run<9>("v_pk_mad_u16", d);
    run<7>("v_add_u32", d);
    return 0;
}
