// Microbenchmark: LDS float-atomic / read / write-exchange throughput per CU on gfx950, with the access shapes the
// grad_value scatter could use (rows of 32 floats at random row indices).  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kRows = 1216;            // 1216 rows x 128 B = 152 KiB
constexpr int kIters = 2048;

__device__ inline unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// mode 0: ds_add_f32, half-wave per row (lane = channel): 2 rows per instruction
// mode 1: ds_read_b128, 8 lanes per row: 8 rows per instruction (what the list walk does)
// mode 2: ds_add_f32, all 64 lanes to different rows' channel (lane & 31) -- i.e. 64 rows per instr, bank = lane&31
// mode 3: ds_read_b32 half-wave per row
// mode 4: ds_add_rtn_f32 half-wave per row
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long* clocks, int rows_pow2_mask) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < kRows * 32; i += 1024) lds[i] = 0.f;
    __syncthreads();
    unsigned s = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 12345u;
    unsigned sw = ((blockIdx.x * 1024 + threadIdx.x) >> (MODE == 1 ? 3 : 5)) * 2654435761u + 777u;   // shared by the lanes of one row
    float acc = 0.f;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc4 = {0, 0, 0, 0};
    long long t0 = wall_clock64();
    long long c0 = clock64();
#pragma unroll 8
    for (int it = 0; it < kIters; ++it) {
        if (MODE == 0) {
            int row = lcg(sw) % kRows;
            __hip_atomic_fetch_add(&lds[row * 32 + (lane & 31)], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 1) {
            int row = lcg(sw) % kRows;
            acc4 += *reinterpret_cast<f4*>(&lds[row * 32 + (lane & 7) * 4]);
        } else if (MODE == 2) {
            int row = lcg(s) % kRows;
            __hip_atomic_fetch_add(&lds[row * 32 + (lane & 31)], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 3) {
            int row = lcg(sw) % kRows;
            acc += lds[row * 32 + (lane & 31)];
        } else if (MODE == 5) {
            int row = lcg(sw) % (kRows / 2);
            __hip_atomic_fetch_add(reinterpret_cast<double*>(lds) + row * 32 + (lane & 31), 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 6) {
            int row = lcg(sw) % kRows;
            __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(lds) + row * 32 + (lane & 31), 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 7) {     // plain (racy) read-modify-write through registers: ds_read_b32 + add + ds_write_b32
            int row = lcg(sw) % kRows;
            volatile float* q = &lds[row * 32 + (lane & 31)];
            *q = *q + 1.0f;
        } else if (MODE == 4) {
            int row = lcg(sw) % kRows;
            acc += __hip_atomic_fetch_add(&lds[row * 32 + (lane & 31)], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    long long c1 = clock64();
    long long t1 = wall_clock64();
    __syncthreads();
    float r = acc + acc4.x + acc4.y + acc4.z + acc4.w;
    for (int i = threadIdx.x; i < kRows * 32; i += 1024) r += lds[i];
    out[blockIdx.x * 1024 + threadIdx.x] = r;
    if (threadIdx.x == 0) { clocks[blockIdx.x * 2] = c1 - c0; clocks[blockIdx.x * 2 + 1] = t1 - t0; }
}

template <int MODE>
void run(const char* name, int bytes_per_lane) {
    const int grid = 256;
    float* out; long long* clk;
    hipMalloc(&out, grid * 1024 * 4); hipMalloc(&clk, grid * 16);
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, kRows * 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 1024, kRows * 128>>>(out, clk, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 1024, kRows * 128>>>(out, clk, 0);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid * 2);
    hipMemcpy(h.data(), clk, grid * 16, hipMemcpyDeviceToHost);
    double c = 0, w = 0; for (int i = 0; i < grid; ++i) { c += h[2 * i]; w += h[2 * i + 1]; }
    c /= grid; w /= grid;
    double bytes = 1024.0 * kIters * bytes_per_lane;       // per CU
    printf("%-44s %8.3f ms  shader clocks %9.0f  -> %6.1f B/clk/CU   (wall clock ticks %9.0f @100MHz = %.3f ms)\n", name, ms, c, bytes / c, w, w / 1e5);
    hipFree(out); hipFree(clk);
}

int main() {
    run<0>("ds_add_f32, 2 rows/instr (lane=channel)", 4);
    run<4>("ds_add_rtn_f32, 2 rows/instr", 4);
    run<2>("ds_add_f32, 64 random rows/instr", 4);
    run<5>("ds_add_f64, 2 rows/instr (lane=channel)", 8);
    run<6>("ds_add_u32, 2 rows/instr", 4);
    run<7>("read+add+write b32 (racy), 2 rows/instr", 4);
    run<3>("ds_read_b32, 2 rows/instr", 4);
    run<1>("ds_read_b128, 8 rows/instr", 16);
    return 0;
}
