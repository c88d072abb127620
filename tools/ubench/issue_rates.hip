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
        } else if (MODE == 7) {   // 24 VALU + 8 SALU
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_xor_b32 %0, %0, %1\n\ts_xor_b32 %4, %4, %5\n\tv_xor_b32 %1, %1, %2\n\tv_xor_b32 %2, %2, %3"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1) :: "scc");
        } else if (MODE == 8) {   // 24 VALU + 12 SALU (2:1)
#pragma unroll
            for (int k = 0; k < 12; ++k)
                asm volatile("v_xor_b32 %0, %0, %1\n\ts_xor_b32 %4, %4, %5\n\tv_xor_b32 %1, %1, %2"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1) :: "scc");
        } else if (MODE == 9) {   // 12 VALU + 24 SALU (1:2)
#pragma unroll
            for (int k = 0; k < 12; ++k)
                asm volatile("s_xor_b32 %5, %5, %4\n\ts_xor_b32 %4, %4, %5\n\tv_xor_b32 %1, %1, %2"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1) :: "scc");
        } else if (MODE == 10) {  // 16 ds_bpermute + 16 VALU
#pragma unroll
            for (int k = 0; k < 16; ++k)
                asm volatile("ds_bpermute_b32 %0, %1, %2\n\tv_xor_b32 %3, %3, %1\n\ts_waitcnt lgkmcnt(0)"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (MODE == 11) {  // 32 VALU with one SGPR operand
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_xor_b32 %0, %4, %1\n\tv_xor_b32 %1, %4, %2\n\tv_xor_b32 %2, %4, %3\n\tv_xor_b32 %3, %4, %0"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s0));
        } else if (MODE == 12) {  // 32 v_cmp_e32 (vcc)
#pragma unroll
            for (int k = 0; k < 32; ++k)
                asm volatile("v_cmp_eq_u32_e32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");
        } else if (MODE == 13) {  // 16 (v_cmp vcc + s_cbranch_vccnz never taken)
#pragma unroll
            for (int k = 0; k < 16; ++k)
                asm volatile("v_cmp_eq_u32_e32 vcc, 0x12345, %0\n\ts_cbranch_vccnz L_nv_%=\n\tL_nv_%=:" :: "v"(a) : "vcc");
        } else if (MODE == 14) {  // 32 v_cndmask with vcc
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %1, %1, %2, vcc\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_cndmask_b32 %3, %3, %0, vcc"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");
        } else if (MODE == 15) {  // 32 v_cndmask (vcc), independent destinations
            unsigned t0, t1, t2, t3;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n\tv_cndmask_b32 %1, %5, %6, vcc\n\tv_cndmask_b32 %2, %6, %7, vcc\n\tv_cndmask_b32 %3, %7, %4, vcc"
                             : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");
            a ^= t0 ^ t1 ^ t2 ^ t3;
        } else if (MODE == 16) {  // 32 v_cndmask e64 with an SGPR-pair mask
            unsigned long long m = ((unsigned long long)s1 << 32) | s0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_cndmask_b32_e64 %0, %0, %1, %4\n\tv_cndmask_b32_e64 %1, %1, %2, %4\n\tv_cndmask_b32_e64 %2, %2, %3, %4\n\tv_cndmask_b32_e64 %3, %3, %0, %4"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(m));
        } else if (MODE == 17) {  // 32 v_add_u32 chain (same structure as the xor loop)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %3, %3, %0"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (MODE == 18) {  // 32 v_readfirstlane
#pragma unroll
            for (int k = 0; k < 32; ++k)
                asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s0) : "v"(a));
        } else if (MODE == 19) {  // 32 DPP moves (row_shr:1)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (MODE == 20) {  // 16 ds_bpermute, one wait per 16
#pragma unroll
            for (int k = 0; k < 4; ++k)
                asm volatile("ds_bpermute_b32 %0, %1, %2\n\tds_bpermute_b32 %3, %1, %2\n\tds_bpermute_b32 %0, %1, %2\n\tds_bpermute_b32 %3, %1, %2"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (MODE == 21) {  // 32 v_cmp_e64 + 32 v_cndmask_e64 pairs on an SGPR pair
            unsigned long long m;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                asm volatile("v_cmp_lt_u32_e64 %0, %1, %2\n\tv_cndmask_b32_e64 %1, %1, %2, %0" : "=&s"(m), "+v"(a) : "v"(b));
        } else if (MODE == 22) {  // 32 v_min_u32 / v_max / v_sub chain (branch-free selects)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_min_u32 %0, %0, %1\n\tv_max_u32 %1, %1, %2\n\tv_sub_u32 %2, %2, %3\n\tv_min_i32 %3, %3, %0"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (MODE == 23) {  // 32 s_load_dword (scalar cache), one wait per 8
#pragma unroll
            for (int k = 0; k < 4; ++k)
                asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %2, 0x4\n\ts_load_dword %0, %2, 0x8\n\ts_load_dword %1, %2, 0xc\n\t"
                             "s_load_dword %0, %2, 0x10\n\ts_load_dword %1, %2, 0x14\n\ts_load_dword %0, %2, 0x18\n\ts_load_dword %1, %2, 0x1c\n\ts_waitcnt lgkmcnt(0)"
                             : "=&s"(s2), "=&s"(s3) : "s"(out) : "memory");
        } else if (MODE == 24) {  // 32 v_bfe / v_lshl_add / v_and_or (VOP3, three VGPR operands)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_bfe_u32 %0, %0, %1, %2\n\tv_lshl_add_u32 %1, %1, 3, %2\n\tv_and_or_b32 %2, %2, %3, %0\n\tv_add3_u32 %3, %3, %0, %1"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        } else if (MODE == 25) {  // 16 v_cmp_e32 (vcc) + 16 v_cndmask_e32 (vcc)
#pragma unroll
            for (int k = 0; k < 16; ++k)
                asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : "vcc");
        } else if (MODE == 26) {  // 4 v_cmp_e32 + 28 v_cndmask_e32 reading the same vcc
#pragma unroll
            for (int k = 0; k < 4; ++k)
                asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_cndmask_b32_e32 %1, %1, %2, vcc\n\tv_cndmask_b32_e32 %2, %2, %3, vcc\n\t"
                             "v_cndmask_b32_e32 %3, %3, %0, vcc\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_cndmask_b32_e32 %1, %1, %2, vcc\n\tv_cndmask_b32_e32 %2, %2, %3, vcc"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");
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
    run<7>("24 VALU + 8 SALU", 32, d_out, blocks);
    run<8>("24 VALU + 12 SALU", 36, d_out, blocks);
    run<9>("12 VALU + 24 SALU", 36, d_out, blocks);
    run<10>("16 ds_bpermute + 16 VALU (wait each)", 32, d_out, blocks);
    run<11>("32 VALU with an SGPR operand", 32, d_out, blocks);
    run<12>("32 v_cmp_e32 (vcc)", 32, d_out, blocks);
    run<13>("16 v_cmp vcc + 16 s_cbranch_vccnz", 32, d_out, blocks);
    run<14>("32 v_cndmask (vcc)", 32, d_out, blocks);
    run<15>("32 v_cndmask (vcc), independent dst", 32, d_out, blocks);
    run<25>("16 v_cmp_e32 + 16 v_cndmask_e32 (vcc)", 32, d_out, blocks);
    run<26>("4 v_cmp_e32 + 28 v_cndmask_e32 (vcc)", 32, d_out, blocks);
    run<16>("32 v_cndmask_e64 (SGPR-pair mask)", 32, d_out, blocks);
    run<17>("32 v_add_u32", 32, d_out, blocks);
    run<18>("32 v_readfirstlane", 32, d_out, blocks);
    run<19>("32 v_mov_dpp row_shr:1", 32, d_out, blocks);
    run<20>("16 ds_bpermute, one wait", 16, d_out, blocks);
    run<21>("16 v_cmp_e64 + 16 v_cndmask_e64", 32, d_out, blocks);
    run<22>("32 v_min/max/sub", 32, d_out, blocks);
    run<23>("32 s_load_dword, wait per 8", 32, d_out, blocks);
    run<24>("32 VOP3 (bfe, lshl_add, and_or, add3)", 32, d_out, blocks);
    return 0;
}
