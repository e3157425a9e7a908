// Issue-rate probe for the integer VALU instructions the byte kernels lean on (gfx950): cycles per wave64 instruction and SIMD.
// Four independent dependency chains per lane, 8 workgroups of 256 threads per CU.
// hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAIN4(INS) \
    asm volatile(INS " %0, %0, %4\n\t" INS " %1, %1, %4\n\t" INS " %2, %2, %4\n\t" INS " %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
#define CHAIN4_3(INS) \
    asm volatile(INS " %0, %0, %4, %5\n\t" INS " %1, %1, %4, %5\n\t" INS " %2, %2, %4, %5\n\t" INS " %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));
template <int OP> __global__ void k(unsigned *out, unsigned seed, int iters) {
    unsigned a = seed + threadIdx.x, b = seed * 3 + threadIdx.x, c = seed * 7, d = seed * 11, e = seed * 13 + threadIdx.x, f = seed * 17;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (OP == 0) CHAIN4("v_min_i32")
            if (OP == 1) CHAIN4("v_pk_min_i16")
            if (OP == 2) CHAIN4_3("v_dot4_u32_u8")
            if (OP == 3) CHAIN4_3("v_min3_i32")
            if (OP == 4) CHAIN4("v_bcnt_u32_b32")
            if (OP == 5) CHAIN4_3("v_perm_b32")
            if (OP == 6) CHAIN4_3("v_alignbyte_b32")
            if (OP == 7) CHAIN4("v_pk_sub_i16")
            if (OP == 8) CHAIN4_3("v_mad_u32_u24")
            if (OP == 9) CHAIN4("v_mul_lo_u32")
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
template <int OP> void run(const char *name) {
    unsigned *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4096, blocks = 256 * 8;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u, 16);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double waveInstr = (double)blocks * 4 * iters * 16 * 4;
    std::printf("%-18s %7.1f G wave-instr/s   %.2f cycles per instruction and SIMD (2.4 GHz, 1024 SIMDs)\n", name, waveInstr / ms / 1e6, 1024 * 2.4e9 / (waveInstr / (ms * 1e-3)));
    (void)hipFree(d);
}
int main() {
    run<0>("v_min_i32"); run<3>("v_min3_i32"); run<1>("v_pk_min_i16"); run<7>("v_pk_sub_i16"); run<2>("v_dot4_u32_u8"); run<4>("v_bcnt_u32_b32");
    run<5>("v_perm_b32"); run<6>("v_alignbyte_b32"); run<8>("v_mad_u32_u24"); run<9>("v_mul_lo_u32");
    return 0;
}
