// Microbenchmark + correctness probe for the REGISTER-ACCUMULATOR scatter (round 5): a wave keeps a tile of grad_value in NACC
// VGPRs (one register per pixel; lane = (channel c, half h): the lower half-wave holds the tile's rows for the TOP corners of a
// point, the upper half-wave the same register indices for the row below) and adds a sampling point with two wave-uniform
// VGPR-INDEXED FMAs (gfx9 s_set_gpr_idx_on: dst and src2 relative to M0).  Records are 32 bytes in LDS -- one 16-byte half per
// half-wave: {grad_out row address, register index, weight of the left corner, weight of the right corner} -- read with ONE
// ds_read_b128 at a per-half address, so a record costs: that read, one address add + one ds_read_b32 for the lane's channel of
// the grad_out row, one v_readfirstlane + one s_set_gpr_idx_idx, two FMAs.  Prints clocks per record and wave, and checks the
// accumulators against a CPU double sum.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -DWPS=2 -o racc_rate racc_rate.hip && ./racc_rate       (WPS = waves per SIMD: 2, 3, 4)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#ifndef WPS
#define WPS 2
#endif
typedef float f32x32 __attribute__((ext_vector_type(32)));
#if WPS == 2            // 256 VGPRs: scratch v[64:95], accumulators v[96:255]
constexpr int kAcc = 160;
#define REGS ".set S, 64\n\t.set A, 96\n\t"
#define ACC_DECL f32x32 a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f
#define ACC_OPERANDS "+{v[96:127]}"(a0), "+{v[128:159]}"(a1), "+{v[160:191]}"(a2), "+{v[192:223]}"(a3), "+{v[224:255]}"(a4)
#define ACC_PARAMS f32x32 &a0, f32x32 &a1, f32x32 &a2, f32x32 &a3, f32x32 &a4
#define ACC_ARGS a0, a1, a2, a3, a4
#define SCRATCH "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", \
                "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95"
#elif WPS == 3          // 168 VGPRs: scratch v[40:71], accumulators v[72:167]
constexpr int kAcc = 96;
#define REGS ".set S, 40\n\t.set A, 72\n\t"
#define ACC_DECL f32x32 a0 = 0.f, a1 = 0.f, a2 = 0.f
#define ACC_OPERANDS "+{v[72:103]}"(a0), "+{v[104:135]}"(a1), "+{v[136:167]}"(a2)
#define ACC_PARAMS f32x32 &a0, f32x32 &a1, f32x32 &a2
#define ACC_ARGS a0, a1, a2
#define SCRATCH "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
                "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71"
#else                   // 128 VGPRs: scratch v[32:63], accumulators v[64:127]
constexpr int kAcc = 64;
#define REGS ".set S, 32\n\t.set A, 64\n\t"
#define ACC_DECL f32x32 a0 = 0.f, a1 = 0.f
#define ACC_OPERANDS "+{v[64:95]}"(a0), "+{v[96:127]}"(a1)
#define ACC_PARAMS f32x32 &a0, f32x32 &a1
#define ACC_ARGS a0, a1
#define SCRATCH "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", \
                "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"
#endif
constexpr int kWaves = 4 * WPS;
constexpr int kRows = 256;           // grad_out rows of a chunk
constexpr int kRecPerWave = 256;     // records per wave list (multiple of 8)

// record u of a batch lives in v[S+4u : S+4u+3] = {row address, register index, w left, w right} (this half-wave's half)
#define RD(u, off) "ds_read_b128 v[S+4*" #u ":S+4*" #u "+3], %[rec] offset:" #off "\n\t"
#define GA(u) "v_add_u32 v[S+4*" #u "], v[S+4*" #u "], %[l4]\n\t"
#define GR(u) "ds_read_b32 v[S+4*" #u "], v[S+4*" #u "]\n\t"
#define RF(u, s) "v_readfirstlane_b32 " s ", v[S+4*" #u "+1]\n\t"
#define FM(u) "v_fma_f32 v[A], v[S+4*" #u "+2], v[S+4*" #u "], v[A]\n\t" \
              "v_fma_f32 v[A+1], v[S+4*" #u "+3], v[S+4*" #u "], v[A+1]\n\t"

// one pass over a wave's record list: n8 batches of 8 records at LDS byte address rec_h (per half-wave: list + 16 h),
// lane4 = 4 * (lane & 31)
__device__ __forceinline__ void accumulate(ACC_PARAMS, unsigned rec_h, int n8, unsigned lane4)
{
    int cnt = n8;
    asm volatile(
        REGS
        "s_cmp_lt_i32 %[cnt], 1\n\t"
        "s_cbranch_scc1 2f\n"
        "1:\n\t"
        RD(0, 0) RD(1, 32) RD(2, 64) RD(3, 96) RD(4, 128) RD(5, 160) RD(6, 192) RD(7, 224)
        "v_add_u32 %[rec], 0x100, %[rec]\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        GA(0) GA(1) GA(2) GA(3)
        GR(0) GR(1) GR(2) GR(3)
        RF(0, "s40") RF(1, "s41") RF(2, "s42") RF(3, "s43")
        "s_waitcnt lgkmcnt(4)\n\t"
        GA(4) GA(5) GA(6) GA(7)
        GR(4) GR(5) GR(6) GR(7)
        RF(4, "s44") RF(5, "s45") RF(6, "s46") RF(7, "s47")
        "s_waitcnt lgkmcnt(4)\n\t"
        "s_set_gpr_idx_on s40, 0xc\n\t"               // dst and src2 relative to M0
        FM(0)
        "s_set_gpr_idx_idx s41\n\t"
        FM(1)
        "s_set_gpr_idx_idx s42\n\t"
        FM(2)
        "s_set_gpr_idx_idx s43\n\t"
        FM(3)
        "s_set_gpr_idx_off\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_set_gpr_idx_on s44, 0xc\n\t"
        FM(4)
        "s_set_gpr_idx_idx s45\n\t"
        FM(5)
        "s_set_gpr_idx_idx s46\n\t"
        FM(6)
        "s_set_gpr_idx_idx s47\n\t"
        FM(7)
        "s_set_gpr_idx_off\n\t"
        "s_sub_u32 %[cnt], %[cnt], 1\n\t"
        "s_cmp_lg_u32 %[cnt], 0\n\t"
        "s_cbranch_scc1 1b\n"
        "2:\n\t"
        : ACC_OPERANDS, [rec] "+v"(rec_h), [cnt] "+s"(cnt)
        : [l4] "v"(lane4)
        : SCRATCH, "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "m0", "scc", "memory");
}

// acc[i] of this lane, i wave-uniform (src0 relative)
__device__ __forceinline__ float acc_read(ACC_PARAMS, int i)
{
    float v;
    asm volatile(
        REGS
        "s_set_gpr_idx_on %[i], 0x1\n\t"
        "v_mov_b32 %[v], v[A]\n\t"
        "s_set_gpr_idx_off\n\t"
        : ACC_OPERANDS, [v] "=v"(v) : [i] "s"(i) : "m0");
    return v;
}

__global__ void __launch_bounds__(kWaves * 64) k(const float *rows_g, const uint4 *recs_g, float *out, long long *clocks, int iters)
{
    extern __shared__ __attribute__((aligned(128))) unsigned char lds[];
    float *rows = reinterpret_cast<float *>(lds);                       // [kRows][32]
    uint4 *recs = reinterpret_cast<uint4 *>(lds + kRows * 128);         // [kWaves][kRecPerWave][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < kRows * 32; i += kWaves * 64) rows[i] = rows_g[i];
    for (int i = tid; i < kWaves * kRecPerWave * 2; i += kWaves * 64) {
        uint4 r = recs_g[(size_t)blockIdx.x * kWaves * kRecPerWave * 2 + i];
        r.x += (unsigned)(size_t)(__attribute__((address_space(3))) void *)rows;       // row offset -> LDS address
        recs[i] = r;
    }
    __syncthreads();
    ACC_DECL;
    const unsigned rec0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(recs + wave * kRecPerWave * 2) + (lane >> 5) * 16;
    const unsigned lane4 = 4u * (lane & 31);
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) accumulate(ACC_ARGS, rec0, kRecPerWave / 8, lane4);
    const long long c1 = clock64();
    float *o = out + ((size_t)blockIdx.x * kWaves + wave) * kAcc * 64;
    for (int i = 0; i < kAcc; ++i) o[i * 64 + lane] = acc_read(ACC_ARGS, i);
    if (lane == 0) clocks[blockIdx.x * kWaves + wave] = c1 - c0;
}

static unsigned fbits(float f) { return *reinterpret_cast<unsigned *>(&f); }
static float bitsf(unsigned u) { return *reinterpret_cast<float *>(&u); }

int main()
{
    const int grid = 256, iters = 64;
    std::vector<float> rows(kRows * 32);
    std::vector<uint4> recs((size_t)grid * kWaves * kRecPerWave * 2);
    srand(1234);
    for (auto &v : rows) v = (rand() % 2001 - 1000) / 1000.f;
    for (size_t i = 0; i < recs.size(); i += 2) {
        const unsigned reg = rand() % (kAcc - 1), row = rand() % kRows;
        for (int h = 0; h < 2; ++h) {
            recs[i + h].x = row * 128u;
            recs[i + h].y = reg;
            recs[i + h].z = fbits((rand() % 1000) / 1000.f);
            recs[i + h].w = fbits((rand() % 1000) / 1000.f);
        }
    }
    float *d_rows, *d_out; uint4 *d_recs; long long *d_clk;
    const size_t n_out = (size_t)grid * kWaves * kAcc * 64;
    (void)hipMalloc(&d_rows, rows.size() * 4); (void)hipMalloc(&d_recs, recs.size() * 16); (void)hipMalloc(&d_out, n_out * 4);
    (void)hipMalloc(&d_clk, grid * kWaves * 8);
    (void)hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_recs, recs.data(), recs.size() * 16, hipMemcpyHostToDevice);
    const size_t lds = kRows * 128 + (size_t)kWaves * kRecPerWave * 32;
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<<<grid, kWaves * 64, lds>>>(d_rows, d_recs, d_out, d_clk, 1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<float> out(n_out);
    (void)hipMemcpy(out.data(), d_out, n_out * 4, hipMemcpyDeviceToHost);
    double worst = 0.0;
    for (int b = 0; b < grid; b += grid - 1)
        for (int w = 0; w < kWaves; ++w) {
            std::vector<double> ref(kAcc * 64, 0.0);
            for (int i = 0; i < kRecPerWave; ++i)
                for (int l = 0; l < 64; ++l) {
                    const uint4 r = recs[(((size_t)b * kWaves + w) * kRecPerWave + i) * 2 + (l >> 5)];
                    const double g = rows[(r.x / 128u) * 32 + (l & 31)];
                    ref[r.y * 64 + l] += (double)bitsf(r.z) * g;
                    ref[(r.y + 1) * 64 + l] += (double)bitsf(r.w) * g;
                }
            for (int i = 0; i < kAcc * 64; ++i)
                worst = fmax(worst, fabs(ref[i] - out[((size_t)b * kWaves + w) * kAcc * 64 + i]));
        }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<<<grid, kWaves * 64, lds>>>(d_rows, d_recs, d_out, d_clk, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> clk(grid * kWaves);
    (void)hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : clk) avg += c; avg /= clk.size();
    const double recs_total = (double)grid * kWaves * kRecPerWave * iters;
    printf("%d waves per SIMD, %d accumulators: max |err| %.3g   %.1f shader clocks per record and wave = %.1f per record and SIMD   "
           "%.3f ms -> %.1f G records/s chip-wide\n", WPS, kAcc, worst, avg / ((double)kRecPerWave * iters),
           avg / ((double)kRecPerWave * iters) / WPS, ms, recs_total / ms * 1e-6);
    return 0;
}
