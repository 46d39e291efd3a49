// Micro-benchmark: LDS atomic add throughput per wave-instruction (gfx950), conflict-free addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters) {
    __shared__ unsigned long long lds[8192];
    const int t = threadIdx.x;
    for (int i = t; i < 8192; i += 256) lds[i] = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int a = (t + 256 * u + 2048 * (it & 3)) & 8191;
            if (MODE == 0) atomicAdd(reinterpret_cast<unsigned int*>(lds) + a, 3u);
            if (MODE == 1) atomicAdd(lds + a, 3ull);
            if (MODE == 2) atomicAdd(reinterpret_cast<float*>(lds) + a, 1.5f);
            if (MODE == 3) { unsigned long long v = lds[a]; lds[a] = v + 3ull; }
            if (MODE == 4) atomicAdd(reinterpret_cast<double*>(lds) + a, 1.5);
        }
    }
    __syncthreads();
    unsigned long long t1 = __builtin_readcyclecounter();
    if (t == 0) out[blockIdx.x] = t1 - t0;
    if (t == 1) out[gridDim.x + blockIdx.x] = lds[5];
}
int main() {
    unsigned long long* d;
    hipMalloc(&d, 4096 * 8);
    const char* names[] = {"ds_add_u32", "ds_add_u64", "ds_add_f32", "read+write b64 (no lock)", "ds_add_f64"};
    for (int mode = 0; mode < 5; mode++) {
        const int iters = 2000, blocks = 256;
        for (int rep = 0; rep < 2; rep++) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, iters);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, iters);
            hipDeviceSynchronize();
        }
        unsigned long long h[256];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0;
        for (int i = 0; i < 256; i++) avg += (double)h[i];
        avg /= 256;
        // 4 waves per block, 8 * iters wave-instructions each -> LDS cycles per wave-instruction
        printf("%-28s %8.1f cycles per wave-instruction (4 waves sharing the LDS; counter ticks)\n", names[mode],
               avg / (8.0 * iters * 4));
    }
    return 0;
}
