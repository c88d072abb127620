// Issue-rate microbenchmark for gfx950: wave-instructions per cycle per CU for VALU-only, SALU-only and mixed streams
// at 8 waves per SIMD (the occupancy of the pair kernel).  Build: hipcc --offload-arch=gfx950 -O3 -o issue_rates issue_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256, 8) k_rate(unsigned* out, int iters)
{
    unsigned a = threadIdx.x, b = a * 3, c = a * 5, d = a * 7;
    unsigned s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 32 VALU
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_xor_b32 %0, %0, %1\n\tv_xor_b32 %1, %1, %2\n\tv_xor_b32 %2, %2, %3\n\tv_xor_b32 %3, %3, %0"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (MODE == 1) {   // 32 SALU
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("s_xor_b32 %0, %0, %1\n\ts_xor_b32 %1, %1, %2\n\ts_xor_b32 %2, %2, %3\n\ts_xor_b32 %3, %3, %0"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
        } else if (MODE == 2) {   // 16 VALU + 16 SALU interleaved
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_xor_b32 %0, %0, %1\n\ts_xor_b32 %4, %4, %5\n\tv_xor_b32 %1, %1, %2\n\ts_xor_b32 %5, %5, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1) :: "scc");
        } else if (MODE == 3) {   // 32 VALU + 32 SALU interleaved (twice the work of mode 2)
#pragma unroll
            for (int k = 0; k < 16; ++k)
                asm volatile("v_xor_b32 %0, %0, %1\n\ts_xor_b32 %4, %4, %5\n\tv_xor_b32 %1, %1, %2\n\ts_xor_b32 %5, %5, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1) :: "scc");
        } else if (MODE == 4) {   // 32 v_readlane (VALU-encoded, scalar destination)
#pragma unroll
            for (int k = 0; k < 32; ++k)
                asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s0) : "v"(a));
        } else if (MODE == 5) {   // 16 s_cmp + 16 s_cbranch (never taken)
#pragma unroll
            for (int k = 0; k < 16; ++k)
                asm volatile("s_cmp_eq_u32 %0, 0x12345\n\ts_cbranch_scc1 L_never_%=\n\tL_never_%=:" :: "s"(s0) : "scc");
        } else if (MODE == 6) {   // 32 v_cmp writing an SGPR pair
            unsigned long long m;
#pragma unroll
            for (int k = 0; k < 32; ++k)
                asm volatile("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
            s0 ^= (unsigned)m;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ s0 ^ s1 ^ s2 ^ s3;
}

template <int MODE>
double run(const char* name, int per_iter, unsigned* d_out, int blocks)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // waves per CU = blocks * 4 / 256 CUs (all resident: blocks = 256 CUs * 8)
    const double insts_per_cu = (double)(blocks / 256) * 4.0 * iters * per_iter;
    printf("%-44s %8.2f ms  %.3f wave-instructions per ns per CU (x/2.4 = per cycle at 2.4 GHz: %.3f)\n", name, ms,
           insts_per_cu / (ms * 1e6), insts_per_cu / (ms * 1e6) / 2.4);
    return ms;
}

int main()
{
    const int blocks = 256 * 8;
    unsigned* d_out;
    hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    run<0>("VALU only (32 v_xor per iteration)", 32, d_out, blocks);
    run<1>("SALU only (32 s_xor)", 32, d_out, blocks);
    run<2>("16 VALU + 16 SALU interleaved", 32, d_out, blocks);
    run<3>("32 VALU + 32 SALU interleaved", 64, d_out, blocks);
    run<4>("32 v_readlane", 32, d_out, blocks);
    run<5>("16 s_cmp + 16 s_cbranch", 32, d_out, blocks);
    run<6>("32 v_cmp_e64 (SGPR-pair result)", 32, d_out, blocks);
    return 0;
}
