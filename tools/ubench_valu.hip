// Developer microbenchmark: issue rate of the integer VALU ops the bit-vector sweep uses (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters) {
    uint32_t x0 = threadIdx.x, x1 = x0 * 3 + 1, x2 = x0 ^ 0x55, x3 = x0 + 7, x4 = x0 * 5, x5 = x0 + 11, x6 = x0 ^ 0x33, x7 = x0 + 13;
    uint32_t b = blockIdx.x + 17, c = threadIdx.x * 7 + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (OP == 0) {
#define X(n) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x##n) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 1) {
#define X(n) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x##n) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 2) {
#define X(n) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x##n) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 3) {
#define X(n) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x##n) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 4) {
#define X(n) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(x##n) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 5) {
#define X(n) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x##n));
                REP8(X)
#undef X
            } else if (OP == 6) {
#define X(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x##n) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 7) {
#define X(n) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x##n) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 8) {
#define X(n) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(x##n) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 9) {
#define X(n) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(x##n) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if (OP == 10) {  // bitop3 with one SGPR-free constant operand
#define X(n) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(x##n) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 11) {
#define X(n) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x##n) : "v"(b));
                REP8(X)
#undef X
            } else if (OP == 12) {
#define X(n) asm volatile("v_add_co_u32 %0, vcc, %0, %0\n\tv_addc_co_u32 %1, vcc, %1, 0, vcc" : "+v"(x##n), "+v"(c) : : "vcc");
                REP8(X)
#undef X
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ c;
}

template <int OP>
void run(const char *name, int per_iter_mult = 1) {
    uint32_t *d;
    hipMalloc(&d, 256 * 2048 * 4);
    const int iters = 4000, blocks = 2048;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 /*waves*/ * iters * 32.0 * per_iter_mult;
    double per_cu_per_clk = winstr / (ms * 1e-3) / 256.0 / 2.4e9;
    printf("%-28s %8.3f ms  %.3f wave-instr/clk/CU (@2.4GHz)  -> %.2f clk/instr/SIMD\n", name, ms, per_cu_per_clk, 4.0 / per_cu_per_clk);
    hipFree(d);
}

int main() {
    run<0>("v_or_b32 (2 src)");
    run<11>("v_xor_b32 (2 src)");
    run<6>("v_add_u32 (2 src)");
    run<5>("v_lshlrev_b32");
    run<1>("v_bitop3_b32 (3 vgpr)");
    run<10>("v_bitop3_b32 (2 distinct)");
    run<2>("v_add3_u32");
    run<3>("v_lshl_add_u32");
    run<8>("v_lshl_or_b32");
    run<7>("v_and_or_b32");
    run<9>("v_bfi_b32");
    run<4>("v_min3_i32");
    run<12>("v_add_co + v_addc_co pair", 2);
    return 0;
}
