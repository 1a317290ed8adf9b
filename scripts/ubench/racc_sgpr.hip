// Microbenchmark + correctness probe, second form of the REGISTER-ACCUMULATOR scatter (round 5; first form: racc_rate.hip,
// records broadcast from LDS, LDS-bound at ~7 CU-clocks per record).  Here a wave's records come through the SCALAR path:
// 16 bytes per sampling point {row offset | register index, ly, w left, w right} in global memory, four records per
// s_load_dwordx16, and the lane's channel of the grad_out row comes from memory (L2) with the row offset as the buffer load's
// SGPR offset -- no LDS at all.  Lane = (channel c, half h): the lower half-wave accumulates the TOP corners of a point
// (weight 1 - ly), the upper half-wave, under the same register index, the row below (weight ly):
//      wy = k0 + k1 * ly;  t = wy * g;  acc[r] += wL * t;  acc[r + 1] += wR * t      (4 VALU, 2 of them VGPR-indexed)
//   hipcc --offload-arch=gfx950 -O3 -DWPS=2 -o racc_sgpr racc_sgpr.hip && ./racc_sgpr       (WPS = waves per SIMD: 2, 3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#ifndef WPS
#define WPS 2
#endif
typedef float f32x32 __attribute__((ext_vector_type(32)));
#if WPS == 2            // 256 VGPRs: g ring v[64:79], temporaries v[80:87], accumulators v[96:255]
constexpr int kAcc = 160;
#define REGS ".set G, 64\n\t.set T, 80\n\t.set A, 96\n\t"
#define ACC_DECL f32x32 a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f
#define ACC_OPERANDS "+{v[96:127]}"(a0), "+{v[128:159]}"(a1), "+{v[160:191]}"(a2), "+{v[192:223]}"(a3), "+{v[224:255]}"(a4)
#define ACC_PARAMS f32x32 &a0, f32x32 &a1, f32x32 &a2, f32x32 &a3, f32x32 &a4
#define ACC_ARGS a0, a1, a2, a3, a4
#define SCRATCH "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", \
                "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87"
#else                   // 168 VGPRs: g ring v[40:55], temporaries v[56:63], accumulators v[72:167]
constexpr int kAcc = 96;
#define REGS ".set G, 40\n\t.set T, 56\n\t.set A, 72\n\t"
#define ACC_DECL f32x32 a0 = 0.f, a1 = 0.f, a2 = 0.f
#define ACC_OPERANDS "+{v[72:103]}"(a0), "+{v[104:135]}"(a1), "+{v[136:167]}"(a2)
#define ACC_PARAMS f32x32 &a0, f32x32 &a1, f32x32 &a2
#define ACC_ARGS a0, a1, a2
#define SCRATCH "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
                "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"
#endif
constexpr int kWaves = 4 * WPS;
constexpr int kRowsG = 1800;         // grad_out rows of a clip (1 KiB apart: 8 heads x 32 channels)
constexpr int kRecPerWave = 1024;    // records per wave list (multiple of 16)

// SGPR ring: set j (0..3) = s[36 + 16 j : 51 + 16 j] = four records {p, ly, wL, wR}; record u of set j starts at s[36 + 16 j + 4 u]
#define SLOAD(j) "s_load_dwordx16 s[36+16*" #j ":51+16*" #j "], s[32:33], 0x0\n\t" \
                 "s_add_u32 s32, s32, 64\n\t" "s_addc_u32 s33, s33, 0\n\t"
// the four row loads of set j into g ring slot j: v[G + 4 j + u]
#define GL1(j, u) "s_andn2_b32 s35, s[36+16*" #j "+4*" #u "], 0xff\n\t" \
                  "buffer_load_dword v[G+4*" #j "+" #u "], %[l4], %[rsrc], s35 offen\n\t"
#define GLOAD(j) GL1(j, 0) GL1(j, 1) GL1(j, 2) GL1(j, 3)
// the FMAs of set j
#define WY(j, u) "v_fma_f32 v[T+" #u "], %[k1], s[36+16*" #j "+4*" #u "+1], %[k0]\n\t"
#define TT(j, u) "v_mul_f32 v[T+" #u "], v[T+" #u "], v[G+4*" #j "+" #u "]\n\t"
#define IX(j, u, op) op " s[36+16*" #j "+4*" #u "]"
#define FM1(j, u) "v_fma_f32 v[A], s[36+16*" #j "+4*" #u "+2], v[T+" #u "], v[A]\n\t" \
                  "v_fma_f32 v[A+1], s[36+16*" #j "+4*" #u "+3], v[T+" #u "], v[A+1]\n\t"
#define FMAS(j) WY(j, 0) WY(j, 1) WY(j, 2) WY(j, 3) TT(j, 0) TT(j, 1) TT(j, 2) TT(j, 3) \
                IX(j, 0, "s_set_gpr_idx_on") ", 0xc\n\t" FM1(j, 0) IX(j, 1, "s_set_gpr_idx_idx") "\n\t" FM1(j, 1) \
                IX(j, 2, "s_set_gpr_idx_idx") "\n\t" FM1(j, 2) IX(j, 3, "s_set_gpr_idx_idx") "\n\t" FM1(j, 3) "s_set_gpr_idx_off\n\t"
// stage k of the steady state: records of batch k+2 have arrived -> their rows requested; batch k+3's records requested; batch k's
// rows have arrived (two younger batches = 8 loads stay in flight) -> its FMAs
#define STAGE(k, k2, k3) "s_waitcnt lgkmcnt(0)\n\t" GLOAD(k2) SLOAD(k3) "s_waitcnt vmcnt(8)\n\t" FMAS(k)

// one pass over a wave's record list of n16 * 16 records (+ 3 batches of padding that are loaded but never used)
__device__ __forceinline__ void accumulate(ACC_PARAMS, const void *recs, int n16, __amdgpu_buffer_rsrc_t rsrc, unsigned lane4, float k0, float k1)
{
    int cnt = n16;
    unsigned long long rp = (unsigned long long)recs;
    asm volatile(
        REGS
        "s_cmp_lt_i32 %[cnt], 1\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_mov_b64 s[32:33], %[rec]\n\t"
        // prologue: batches 0, 1 loaded and their rows requested; batch 2 requested
        SLOAD(0) SLOAD(1)
        "s_waitcnt lgkmcnt(0)\n\t"
        GLOAD(0) GLOAD(1)
        SLOAD(2)
        "1:\n\t"
        STAGE(0, 2, 3)
        STAGE(1, 3, 0)
        STAGE(2, 0, 1)
        STAGE(3, 1, 2)
        "s_sub_u32 %[cnt], %[cnt], 1\n\t"
        "s_cmp_lg_u32 %[cnt], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
        "2:\n\t"
        : ACC_OPERANDS, [cnt] "+s"(cnt)
        : [rec] "s"(rp), [rsrc] "s"(rsrc), [l4] "v"(lane4), [k0] "v"(k0), [k1] "v"(k1)
        : SCRATCH, "s32", "s33", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
          "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",
          "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83",
          "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99",
          "m0", "scc", "memory");
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
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    ACC_DECL;
    const uint4 *mine = recs_g + ((size_t)blockIdx.x * kWaves + wave) * (kRecPerWave + 16);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(rows_g), 0, kRowsG * 1024, 0x00020000);
    const unsigned lane4 = 4u * (lane & 31);
    const float k0 = lane < 32 ? 1.f : 0.f, k1 = lane < 32 ? -1.f : 1.f;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) accumulate(ACC_ARGS, mine, kRecPerWave / 16, rsrc, lane4, k0, k1);
    const long long c1 = clock64();
    float *o = out + ((size_t)blockIdx.x * kWaves + wave) * kAcc * 64;
    for (int i = 0; i < kAcc; ++i) o[i * 64 + lane] = acc_read(ACC_ARGS, i);
    if (lane == 0) clocks[blockIdx.x * kWaves + wave] = c1 - c0;
}

static unsigned fbits(float f) { return *reinterpret_cast<unsigned *>(&f); }
static float bitsf(unsigned u) { return *reinterpret_cast<float *>(&u); }

int main()
{
    const int grid = 256, iters = 16;
    std::vector<float> rows((size_t)kRowsG * 256);
    const size_t per_wave = kRecPerWave + 16;
    std::vector<uint4> recs((size_t)grid * kWaves * per_wave);
    srand(1234);
    for (auto &v : rows) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto &r : recs) {
        const unsigned reg = rand() % (kAcc - 1), row = rand() % kRowsG;
        r.x = (row * 1024u) | reg;
        r.y = fbits((rand() % 1000) / 1000.f);
        r.z = fbits((rand() % 1000) / 1000.f);
        r.w = fbits((rand() % 1000) / 1000.f);
    }
    float *d_rows, *d_out; uint4 *d_recs; long long *d_clk;
    const size_t n_out = (size_t)grid * kWaves * kAcc * 64;
    (void)hipMalloc(&d_rows, rows.size() * 4); (void)hipMalloc(&d_recs, recs.size() * 16); (void)hipMalloc(&d_out, n_out * 4);
    (void)hipMalloc(&d_clk, grid * kWaves * 8);
    (void)hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_recs, recs.data(), recs.size() * 16, hipMemcpyHostToDevice);
    k<<<grid, kWaves * 64>>>(d_rows, d_recs, d_out, d_clk, 1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<float> out(n_out);
    (void)hipMemcpy(out.data(), d_out, n_out * 4, hipMemcpyDeviceToHost);
    double worst = 0.0;
    for (int b = 0; b < grid; b += grid - 1)
        for (int w = 0; w < kWaves; ++w) {
            std::vector<double> ref(kAcc * 64, 0.0);
            for (int i = 0; i < kRecPerWave; ++i) {
                const uint4 r = recs[((size_t)b * kWaves + w) * per_wave + i];
                const unsigned reg = r.x & 0xffu, row = r.x >> 10;
                for (int l = 0; l < 64; ++l) {
                    const double wy = l < 32 ? 1.0 - bitsf(r.y) : bitsf(r.y), g = rows[(size_t)row * 256 + (l & 31)];
                    ref[reg * 64 + l] += (double)bitsf(r.z) * wy * g;
                    ref[(reg + 1) * 64 + l] += (double)bitsf(r.w) * wy * g;
                }
            }
            for (int i = 0; i < kAcc * 64; ++i)
                worst = fmax(worst, fabs(ref[i] - out[((size_t)b * kWaves + w) * kAcc * 64 + i]));
        }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    k<<<grid, kWaves * 64>>>(d_rows, d_recs, d_out, d_clk, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> clk(grid * kWaves);
    (void)hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : clk) avg += c; avg /= clk.size();
    const double recs_total = (double)grid * kWaves * kRecPerWave * iters;
    printf("scalar-path records, %d waves per SIMD, %d accumulators: max |err| %.3g   %.1f shader clocks per record and wave = %.1f per "
           "record and SIMD   %.3f ms -> %.1f G records/s chip-wide\n", WPS, kAcc, worst, avg / ((double)kRecPerWave * iters),
           avg / ((double)kRecPerWave * iters) / WPS, ms, recs_total / ms * 1e-6);
    return 0;
}
