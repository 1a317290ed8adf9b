// Micro-benchmark: does an exec-masked-off lane cost LDS bandwidth?  16 waves per CU stream ds_read_b128 / b64 / b32
// (random 16-byte-aligned addresses inside 64 KiB, 8 independent reads in flight per lane) with ALL lanes, every
// second quad, every fourth quad, one quad per wave active.  Reports clk per wave instruction per CU.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/lds_read_mask.hip -o scripts/ubench/lds_read_mask
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int BYTES>
__global__ void __launch_bounds__(1024) k(float *out, int iters, int quad_mod, int same_addr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (int i = threadIdx.x; i < 16384; i += 1024) reinterpret_cast<unsigned *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    const int quad = threadIdx.x / 4;
    float acc = 0.f;
    long long t0 = clock64();
    if (quad % quad_mod == 0) {
        unsigned a = (threadIdx.x * 2654435761u) >> 7;
        if (same_addr) a = (quad * 2654435761u) >> 7;                 // the 4 lanes of a quad read the same address
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned ad = ((a + u * 7919u + it * 104729u) * 16u) & 0xfff0u;
                if (BYTES == 16) { const float4 v = *reinterpret_cast<const float4 *>(lds + ad); acc += v.x + v.w; }
                if (BYTES == 8) { const float2 v = *reinterpret_cast<const float2 *>(lds + ad); acc += v.x + v.y; }
                if (BYTES == 4) { const float v = *reinterpret_cast<const float *>(lds + ad); acc += v; }
            }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long *)out)[1 << 19] = t1 - t0;
}
template <int BYTES> void run(int quad_mod, int same)
{
    float *out; hipMalloc(&out, (1 << 22) + 64);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<BYTES><<<256, 1024, 65536>>>(out, iters, quad_mod, same);
    hipEventRecord(e0); k<BYTES><<<256, 1024, 65536>>>(out, iters, quad_mod, same); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 8 * 16;                          // wave instructions per CU
    printf("ds_read_b%-3d active quads 1/%-2d %s: %.2f clk per wave instruction per CU (wall, 2.4 GHz)\n", BYTES * 8, quad_mod,
           same ? "quad-uniform address" : "per-lane address   ", ms * 1e-3 * 2.4e9 / n);
    hipFree(out);
}
int main()
{
    for (int same : {0, 1})
        for (int qm : {1, 2, 4, 16}) { run<16>(qm, same); run<8>(qm, same); run<4>(qm, same); }
    return 0;
}
