// Micro-benchmark: issue cost of VALU instruction flavours on gfx950 (clk per wave64 instruction per SIMD), with W
// waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_rate.hip -o scripts/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define QP "quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf"
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND>
__global__ void k(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0001f, c = 0.5f;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, ib = 3;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile(REP64("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if (KIND == 1) asm volatile(REP64("v_fmac_f32_dpp %0, %4, %5 " QP "\n v_fmac_f32_dpp %1, %4, %5 " QP "\n v_fmac_f32_dpp %2, %4, %5 " QP "\n v_fmac_f32_dpp %3, %4, %5 " QP "\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if (KIND == 2) asm volatile(REP64("v_add_u32_dpp %0, %4, %0 " QP "\n v_add_u32_dpp %1, %4, %1 " QP "\n v_add_u32_dpp %2, %4, %2 " QP "\n v_add_u32_dpp %3, %4, %3 " QP "\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(ib));
        if (KIND == 3) asm volatile(REP64("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %4, %2\n v_add_u32 %3, %4, %3\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(ib));
        if (KIND == 4) asm volatile(REP64("v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n") : "+v"(*(double *)&a0), "+v"(*(double *)&a2) : "v"(*(double *)&b), "v"(*(double *)&c));
        if (KIND == 5) asm volatile(REP64("v_mul_u32_u24 %0, %4, %0\n v_mul_u32_u24 %1, %4, %1\n v_mul_u32_u24 %2, %4, %2\n v_mul_u32_u24 %3, %4, %3\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(ib));
        if (KIND == 6) asm volatile(REP64("v_mul_lo_u32 %0, %4, %0\n v_mul_lo_u32 %1, %4, %1\n v_mul_lo_u32 %2, %4, %2\n v_mul_lo_u32 %3, %4, %3\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(ib));
        if (KIND == 7) asm volatile(REP64("v_mov_b32_dpp %0, %4 " QP "\n v_mov_b32_dpp %1, %4 " QP "\n v_mov_b32_dpp %2, %4 " QP "\n v_mov_b32_dpp %3, %4 " QP "\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(ib));
        if (KIND == 8) asm volatile(REP64("v_cvt_f64_f32 %0, %2\n v_cvt_f64_f32 %1, %3\n") : "+v"(*(double *)&a0), "+v"(*(double *)&a2) : "v"(b), "v"(c));
        if (KIND == 9) asm volatile(REP64("v_cndmask_b32 %0, %4, %0, vcc\n v_cndmask_b32 %1, %4, %1, vcc\n v_cndmask_b32 %2, %4, %2, vcc\n v_cndmask_b32 %3, %4, %3, vcc\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(ib) : "vcc");
        if (KIND == 10) asm volatile(REP64("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if (KIND == 11) asm volatile(REP64("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n") ::: "s20", "s21", "s22", "s23", "scc");
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + i0 + i1 + i2 + i3;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long *)out)[1 << 20] = t1 - t0;
}
template <int KIND> void run(const char *name, int waves_per_simd)
{
    float *out; hipMalloc(&out, (1 << 22) + 64);
    const int iters = 200, threads = 64 * 4 * waves_per_simd;      // one block per CU, waves_per_simd per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256, threads>>>(out, iters);
    hipEventRecord(e0); k<KIND><<<256, threads>>>(out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int per = (KIND == 4 || KIND == 8) ? 2 : 4;
    const double n = (double)iters * 64 * per;                       // instructions per wave
    long long clk; hipMemcpy(&clk, (char *)out + (1 << 22), 8, hipMemcpyDeviceToHost);
    printf("%-18s waves/SIMD %d: %.2f clk/instr/SIMD by wall (2.4 GHz), %.2f by s_memtime/clock64 of wave 0 per own instr\n", name,
           waves_per_simd, ms * 1e-3 * 2.4e9 / (n * waves_per_simd), (double)clk / n);
    hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_fmac_f32", w); run<10>("v_fma_f32", w); run<1>("v_fmac_f32_dpp", w); run<3>("v_add_u32", w); run<2>("v_add_u32_dpp", w);
        run<7>("v_mov_b32_dpp", w); run<4>("v_pk_fma_f32", w); run<5>("v_mul_u32_u24", w); run<6>("v_mul_lo_u32", w);
        run<8>("v_cvt_f64_f32", w); run<9>("v_cndmask_b32", w); run<11>("s_add_u32", w);
    }
    return 0;
}
