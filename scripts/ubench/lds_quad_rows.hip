// Micro-benchmark: ds_read_b128 where the 4 lanes of a quad read 64 contiguous bytes of a random 128-byte row (the
// owner-computes scatter's walk / the resident-slab gathers).  Which quads should take the first / second half of
// their row so that a wave instruction spreads over all banks?  half = (Q >> SH) & 1 for SH = 0..3, all quads on
// the first half (worst case), and a row-parity swizzle.  16 waves per CU; reports clk per wave instruction per CU.
// Build: hipcc -w --offload-arch=gfx950 -O3 scripts/ubench/lds_quad_rows.hip -o scripts/ubench/lds_quad_rows
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(1024) k(float *out, int iters, int mode, int live_mod)
{
    extern __shared__ __attribute__((aligned(128))) unsigned char lds[];
    for (int i = threadIdx.x; i < 16384; i += 1024) reinterpret_cast<unsigned *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    const int Q = threadIdx.x / 4, c = threadIdx.x & 3;
    float acc = 0.f;
    if (Q % live_mod == 0) {
        // 8 row addresses per lane, advanced by one add + one and per read (the loop must not be VALU-bound)
        unsigned ad[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned row = ((Q * 2654435761u) >> 9) + u * 7919u;
            unsigned half = 0;
            if (mode >= 0 && mode <= 3) half = (Q >> mode) & 1;
            if (mode == 5) half = (row ^ Q) & 1;
            if (mode == 6) half = row & 1;
            if (mode == 7) ad[u] = ((threadIdx.x * 2654435761u) >> 9) * 16u + u * 7919u * 16u;      // random 16 B per lane
            else if (mode == 8) ad[u] = threadIdx.x % 64 * 16u + (row & 0x3f) * 1024u;               // a wave reads 1 KiB contiguous
            else ad[u] = row * 128u + half * 64u + c * 16u;
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ad[u] = (ad[u] + (mode == 8 ? 1024u * 13u : 128u * 37u)) & 0xffffu;                // stays 128-B (row) aligned + offset
                const float4 v = *reinterpret_cast<const float4 *>(lds + ad[u]);
                acc += v.x;
            }
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}
int main()
{
    float *out; hipMalloc(&out, 1 << 22);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"half = Q & 1", "half = (Q >> 1) & 1", "half = (Q >> 2) & 1", "half = (Q >> 3) & 1", "all first half", "half = (row ^ Q) & 1", "half = row & 1", "random 16 B per lane", "1 KiB contiguous per wave"};
    for (int live : {1, 2, 3})
        for (int mode = 0; mode < 9; ++mode) {
            k<<<256, 1024, 65536>>>(out, iters, mode, live);
            hipEventRecord(e0); k<<<256, 1024, 65536>>>(out, iters, mode, live); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("live quads 1/%d  %-22s: %.2f clk per wave instruction per CU\n", live, names[mode], ms * 1e-3 * 2.4e9 / ((double)iters * 8 * 16));
        }
    return 0;
}
