// Microbenchmark: LDS atomic add rate on gfx950 for f32 / u32 / u64 and plain stores,
// random addresses over a 512-entry array (the tile kernel's W/S2 pattern).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(1024) void k(int iters, float *out)
{
    __shared__ float f[8192];
    __shared__ unsigned long long u64[4096];
    unsigned *u = reinterpret_cast<unsigned *>(f);
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) f[i] = 0;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) u64[i] = 0;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const unsigned a = (s >> 10) & 511;
        if (MODE == 0) atomicAdd(&f[a], 1.0f);
        if (MODE == 1) atomicAdd(&u[a], 1u);
        if (MODE == 2) atomicAdd(&u64[a], 1ull);
        if (MODE == 3) f[a] = (float)it;
        if (MODE == 4) { atomicAdd(&f[a], 1.0f); atomicAdd(&f[512 + a], 2.0f); }
        if (MODE == 5) { float r = atomicAdd(&f[a], 1.0f); asm volatile("" ::"v"(r)); }
        if (MODE == 6) atomicAdd(&f[(s >> 10) & 8191], 1.0f);
        if (MODE == 7) {      // float add by compare-and-swap on the bit pattern
            unsigned *p = &u[a];
            unsigned old = *p, assumed;
            do {
                assumed = old;
                old = atomicCAS(p, assumed, __float_as_uint(__uint_as_float(assumed) + 1.0f));
            } while (old != assumed);
        }
        if (MODE == 8) {      // same, 27648-entry table (the D array of a tile)
            unsigned *p = &u[(s >> 10) % 8000];
            unsigned old = *p, assumed;
            do {
                assumed = old;
                old = atomicCAS(p, assumed, __float_as_uint(__uint_as_float(assumed) + 1.0f));
            } while (old != assumed);
        }
        if (MODE == 9) atomicAdd(&u[(s >> 10) % 8000], 3u);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = f[0] + (float)u64[0];
}

int main()
{
    float *out; hipMalloc(&out, 4096);
    const int iters = 20000, blocks = 256, threads = 1024;
    const char *names[] = {"ds_add_f32 (512 addr)", "ds_add_u32", "ds_add_u64", "plain ds_write_b32", "2x ds_add_f32", "ds_add_rtn_f32", "ds_add_f32 (8192 addr)", "CAS-loop f32 (512 addr)", "CAS-loop f32 (8000 addr)", "ds_add_u32 (8000 addr)"};
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int m = 0; m < 10; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            switch (m) {
            case 0: hipLaunchKernelGGL(k<0>, blocks, threads, 0, 0, iters, out); break;
            case 1: hipLaunchKernelGGL(k<1>, blocks, threads, 0, 0, iters, out); break;
            case 2: hipLaunchKernelGGL(k<2>, blocks, threads, 0, 0, iters, out); break;
            case 3: hipLaunchKernelGGL(k<3>, blocks, threads, 0, 0, iters, out); break;
            case 4: hipLaunchKernelGGL(k<4>, blocks, threads, 0, 0, iters, out); break;
            case 5: hipLaunchKernelGGL(k<5>, blocks, threads, 0, 0, iters, out); break;
            case 6: hipLaunchKernelGGL(k<6>, blocks, threads, 0, 0, iters, out); break;
            case 7: hipLaunchKernelGGL(k<7>, blocks, threads, 0, 0, iters, out); break;
            case 8: hipLaunchKernelGGL(k<8>, blocks, threads, 0, 0, iters, out); break;
            case 9: hipLaunchKernelGGL(k<9>, blocks, threads, 0, 0, iters, out); break;
            }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 1) {
                const double ops = (double)iters * threads * (m == 4 ? 2 : 1);   // lane-ops per CU (1 block per CU)
                printf("%-26s %8.3f ms  %6.2f ns per wave-instr/CU  %.3f lane-ops/clk/CU @2.4GHz\n", names[m], ms,
                       ms * 1e6 / (ops / 64), ops / (ms * 1e-3 * 2.4e9));
            }
        }
    }
    return 0;
}
