// Developer microbenchmark (gfx950): issue rate of the integer VALU instructions the filter kernels are built from.
// Eight independent dependency chains per lane, 2048 workgroups x 256 lanes (16 waves per CU resident).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_ops tools/ubench_ops.hip && /tmp/ubench_ops
// The table is relative: the clock under load is below 2.4 GHz, so read the column as "x times v_or_b32".
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define OPS(F)                                                                                            \
    F(0, "v_or_b32 x,x,b", "v_or_b32 %0, %0, %1", 1)                                                      \
    F(1, "v_and_b32 x,x,b", "v_and_b32 %0, %0, %1", 1)                                                    \
    F(2, "v_and_b32 x,literal,x", "v_and_b32 %0, 0x0f0f0f0f, %0", 1)                                      \
    F(3, "v_add_u32 x,x,b", "v_add_u32 %0, %0, %1", 1)                                                    \
    F(4, "v_sub_u32 x,x,b", "v_sub_u32 %0, %0, %1", 1)                                                    \
    F(5, "v_min_u32 x,x,b", "v_min_u32 %0, %0, %1", 1)                                                    \
    F(6, "v_mov_b32 x,b", "v_mov_b32 %0, %1", 1)                                                          \
    F(7, "v_lshlrev_b32 x,1,x", "v_lshlrev_b32 %0, 1, %0", 1)                                             \
    F(8, "v_lshrrev_b32 x,3,x", "v_lshrrev_b32 %0, 3, %0", 1)                                             \
    F(9, "v_lshrrev_b32 x,b,x (vgpr amount)", "v_lshrrev_b32 %0, %1, %0", 1)                              \
    F(10, "v_bfe_u32 x,x,3,13", "v_bfe_u32 %0, %0, 3, 13", 1)                                             \
    F(11, "v_bfe_u32 x,x,b,1 (vgpr offset)", "v_bfe_u32 %0, %0, %1, 1", 1)                                \
    F(12, "v_alignbit_b32 x,x,b,7", "v_alignbit_b32 %0, %0, %1, 7", 1)                                    \
    F(13, "v_alignbit_b32 x,x,b,c (vgpr)", "v_alignbit_b32 %0, %0, %1, %2", 1)                            \
    F(14, "v_perm_b32 x,x,b,c", "v_perm_b32 %0, %0, %1, %2", 1)                                           \
    F(15, "v_mul_u32_u24 x,x,b", "v_mul_u32_u24 %0, %0, %1", 1)                                           \
    F(16, "v_mad_u32_u24 x,x,b,c", "v_mad_u32_u24 %0, %0, %1, %2", 1)                                     \
    F(17, "v_mul_lo_u32 x,x,b", "v_mul_lo_u32 %0, %0, %1", 1)                                             \
    F(18, "v_cndmask_b32 x,x,b,vcc", "v_cndmask_b32 %0, %0, %1, vcc", 1)                                  \
    F(19, "v_sad_u8 x,x,b,c", "v_sad_u8 %0, %0, %1, %2", 1)                                               \
    F(20, "v_bcnt_u32_b32 x,x,b", "v_bcnt_u32_b32 %0, %0, %1", 1)                                         \
    F(21, "v_ffbl_b32 x,x", "v_ffbl_b32 %0, %0", 1)                                                       \
    F(22, "v_bitop3_b32 x,x,b,c", "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96", 1)                           \
    F(23, "v_or3_b32 x,x,b,c", "v_or3_b32 %0, %0, %1, %2", 1)                                             \
    F(24, "v_and_or_b32 x,x,b,c", "v_and_or_b32 %0, %0, %1, %2", 1)                                       \
    F(25, "v_lshl_or_b32 x,x,1,b", "v_lshl_or_b32 %0, %0, 1, %1", 1)                                      \
    F(26, "v_lshl_add_u32 x,x,2,b", "v_lshl_add_u32 %0, %0, 2, %1", 1)                                    \
    F(27, "v_xad_u32 x,x,b,c", "v_xad_u32 %0, %0, %1, %2", 1)                                             \
    F(28, "v_add_co_u32 x,vcc,x,b", "v_add_co_u32 %0, vcc, %0, %1", 1)                                    \
    F(29, "v_add_co + v_addc_co pair", "v_add_co_u32 %0, vcc, %0, %0\n\tv_addc_co_u32 %2, vcc, %2, 0, vcc", 2) \
    F(30, "v_cmp_lt_u32 vcc,x,b + v_cndmask", "v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc", 2) \
    F(31, "v_bfi_b32 x,x,b,c", "v_bfi_b32 %0, %0, %1, %2", 1)                                             \
    F(32, "v_add3_u32 x,x,b,c", "v_add3_u32 %0, %0, %1, %2", 1)                                           \
    F(33, "v_min3_u32 x,x,b,c", "v_min3_u32 %0, %0, %1, %2", 1)                                           \
    F(34, "v_lshlrev_b32 x,b,x (vgpr amount)", "v_lshlrev_b32 %0, %1, %0", 1)                             \
    F(35, "v_mbcnt_lo_u32_b32 x,b,x", "v_mbcnt_lo_u32_b32 %0, %1, %0", 1)                                 \
    F(36, "v_bfm_b32 x,x,b", "v_bfm_b32 %0, %0, %1", 1)                                                   \
    F(37, "v_xor_b32 x,x,b", "v_xor_b32 %0, %0, %1", 1)                                                   \
    F(38, "v_lshrrev_b64 x2,3,x2", "v_lshrrev_b64 %3, 3, %3", 1)                                          \
    F(39, "v_and_b32 x,x,s (sgpr)", "v_and_b32 %0, %4, %0", 1)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t sarg) {
    uint32_t x0 = threadIdx.x, x1 = x0 * 3 + 1, x2 = x0 ^ 0x55, x3 = x0 + 7, x4 = x0 * 5, x5 = x0 + 11, x6 = x0 ^ 0x33, x7 = x0 + 13;
    unsigned long long y0 = x0, y1 = x1, y2 = x2, y3 = x3, y4 = x4, y5 = x5, y6 = x6, y7 = x7;
    uint32_t b = blockIdx.x + 17, c = threadIdx.x * 7 + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#define CASE(id, name, text, n)                                                                           \
    if (OP == id) {                                                                                       \
        asm volatile(text : "+v"(x0), "+v"(b), "+v"(c), "+v"(y0) : "s"(sarg) : "vcc");                      \
        asm volatile(text : "+v"(x1), "+v"(b), "+v"(c), "+v"(y1) : "s"(sarg) : "vcc");                      \
        asm volatile(text : "+v"(x2), "+v"(b), "+v"(c), "+v"(y2) : "s"(sarg) : "vcc");                      \
        asm volatile(text : "+v"(x3), "+v"(b), "+v"(c), "+v"(y3) : "s"(sarg) : "vcc");                      \
        asm volatile(text : "+v"(x4), "+v"(b), "+v"(c), "+v"(y4) : "s"(sarg) : "vcc");                      \
        asm volatile(text : "+v"(x5), "+v"(b), "+v"(c), "+v"(y5) : "s"(sarg) : "vcc");                      \
        asm volatile(text : "+v"(x6), "+v"(b), "+v"(c), "+v"(y6) : "s"(sarg) : "vcc");                      \
        asm volatile(text : "+v"(x7), "+v"(b), "+v"(c), "+v"(y7) : "s"(sarg) : "vcc");                      \
    }
            OPS(CASE)
#undef CASE
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ c ^ b ^ (uint32_t)(y0 ^ y1 ^ y2 ^ y3 ^ y4 ^ y5 ^ y6 ^ y7);
}

static double base_ms = 0;

template <int OP>
void run(const char *name, int mult) {
    uint32_t *d;
    hipMalloc(&d, 256 * 2048 * 4);
    const int iters = 2000, blocks = 2048;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 100, 0x0f0f0f0fu);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 0x0f0f0f0fu);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_instr = ms / mult;
    if (OP == 0) base_ms = per_instr;
    const double winstr = (double)blocks * 4 * iters * 32.0 * mult;
    const double per_cu_per_clk = winstr / (ms * 1e-3) / 256.0 / 2.4e9;
    printf("%-40s %8.3f ms  %5.2f clk/instr/SIMD @2.4GHz   %.2f x v_or_b32\n", name, ms, 4.0 / per_cu_per_clk, per_instr / base_ms);
    hipFree(d);
}

int main() {
#define RUN(id, name, text, n) run<id>(name, n);
    OPS(RUN)
#undef RUN
    return 0;
}
