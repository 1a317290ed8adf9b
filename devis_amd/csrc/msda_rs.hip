// msda_rs.hip -- "resident-slab" kernels: forward and backward gather pass of the DeVIS shapes (D = 32).
#include "msda_common.h"

namespace msda {
namespace {

// ------------------------------------------------------------------------------------------------
// "resident-slab" kernels: levels l0..L-1 of one source frame live in LDS, everything per point in registers
// ------------------------------------------------------------------------------------------------
// ~156 KiB of a CU's 160 KiB of LDS are slab (levels 1-3 of the DeVIS pyramids: 75 % of the taps), staged by LDS-DMA:
//   * a 1024-thread workgroup owns (clip, head, a run of up to NT*16 row tiles); its waves keep the accumulators
//     of NT tiles in registers while the workgroup walks the clip's SOURCE FRAMES; per frame the slab
//     value[frame, levels >= l0, head, :] is staged once, then every wave runs, for each of its tiles, the slots of
//     that tile that read the frame (slot masks [T, T] built once in LDS: no frame-table chasing);
//   * a row (query, head) is served by ONE QUAD: 16 rows per wave, lane c of the quad holding channels
//     [4c, 4c+4) of both halves of the row (D = 32; 2-byte types: channels [8c, 8c+8)).  Lane c also owns point
//     (g0 + c) of the row's current group of four points and turns it into four corner records (rs_geometry); in step R
//     the quad reads lane R's records through quad_perm DPP operands (no LDS crossbar) and loads the four corners with
//     16-byte loads.  (Measured, scripts/ubench/valu_rate.hip: a DPP operand makes a VALU instruction half rate, so a
//     weight is moved once per corner with v_mov_b32_dpp and then feeds 8 plain FMAs: that is why a row is a quad with
//     8 channels per lane and not 8 lanes with 4.)
//   * quads alternate which 64-byte half of a 128-byte row they read first, which halves the LDS bank conflicts
//     of the 16-lane ds_read_b128 groups (4 quads = 4 half rows on 4 different 16-bank quarters when row
//     parities differ);
//   * a corner outside the map reads a zero row kept in LDS (slab levels) or an out-of-range buffer offset
//     (other levels: buffer loads return 0 without touching memory), so a non-finite value at an unrelated
//     pixel can never leak into a row that does not sample it.
#ifndef MSDA_RS_PAIR
#define MSDA_RS_PAIR 1       // forward: the two LDS corners of a pair are requested together, then consumed (0: one after the other).
                             // Same speed on every shape (round 4, same box: 0.375 / 0.376 ms fp32, 0.282 / 0.282 bf16), but the
                             // production 16-bit kernel (4 tiles per wave) no longer spills VGPRs (6 -> 0; profiles/r04_resource_usage.txt)
#endif
#ifndef MSDA_RS_PREFETCH
#define MSDA_RS_PREFETCH 0   // forward, one tile per wave: the first slot's points of a frame loaded into registers early -- 1: behind the
                             // slab's LDS-DMA (round 4: slower, loads return in order and the slab wait inherits their HBM latency),
                             // 2: BEFORE the barrier that frees the previous slab, so that they fly while the wave waits for the slowest
                             // wave of the frame before and land ahead of the DMA pieces (round 5)
#endif
#ifndef MSDA_RS_PIPE
#define MSDA_RS_PIPE 1       // software-pipelined level-0 corners (0: the plain group loop only; A/B builds)
#endif
#include "msda_rs_common.inc"

// Timeline probe (-DMSDA_RS_TRACE, experimental builds only; scripts/rs_trace.py): lane 0 of one wave of the first 8 workgroups
// stamps the shader clock at phase boundaries of the forward.
#ifdef MSDA_RS_TRACE
constexpr int kRsTraceLen = 4096;
__device__ unsigned long long g_rs_trace[8][kRsTraceLen];
__device__ int g_rs_trace_n[8];
#define MSDA_RTR(id) do { if (tr_on && tr_n < kRsTraceLen) g_rs_trace[blockIdx.x][tr_n++] = ((unsigned long long)__builtin_readcyclecounter() << 8) | (unsigned)(id); } while (0)
#else
#define MSDA_RTR(id) do { } while (0)
#endif

// Static issue priority by wave index: the waves of a workgroup share their SIMD's issue slots by priority, then AGE, so the
// last-dispatched waves of a 16-wave workgroup lose every arbitration and set the time of each frame's barrier (timeline probe,
// profiles/r04_logs/rs_trace_*.txt: the last wave's slots took twice as long as wave 1's).  MSDA_RS_PRIO = 1: priority 0..3 by
// quarter of the workgroup, youngest highest.
#ifndef MSDA_RS_PRIO
#define MSDA_RS_PRIO 0
#endif
__device__ __forceinline__ void rs_wave_priority(int wave)
{
#if MSDA_RS_PRIO
    const int q = wave * 4 / kRsWaves;
    if (q == 1) __builtin_amdgcn_s_setprio(1);
    else if (q == 2) __builtin_amdgcn_s_setprio(2);
    else if (q == 3) __builtin_amdgcn_s_setprio(3);
#else
    (void)wave;
#endif
}

// levels >= l0 of source frame f (head m) -> LDS slab, 16 bytes per lane by LDS-DMA (8 lanes per pixel)
template <typename T>
__device__ __forceinline__ void rs_stage_slab(const Params &p, T *slab, int clip, int m, int f, int px0, int npx,
                                              int wave, int lane, bool wait = true)
{
    constexpr int GL = rs_row_bytes<T>() / 16, D = 32;
    constexpr int PXW = kWave / GL;                 // pixels per LDS-DMA wave instruction
    const T *src = static_cast<const T *>(p.value) + clip * p.v_clip + m * p.v_head + ((int64_t)f * p.S + px0) * p.v_pix;
    for (int pb = wave * PXW; pb < npx; pb += kRsWaves * PXW) {
        const int px = min(pb + lane / GL, npx - 1);
        const T *gp = src + (int64_t)px * p.v_pix + (lane % GL) * (16 / (int)sizeof(T));
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_global_load_lds(gp, (__attribute__((address_space(3))) void *)(slab + (size_t)pb * D), 16, 0, 0);
#else
        (void)gp;
#endif
    }
    if (wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The shared front of the resident-slab kernels: LDS carve, level tables, slot masks, tile geometry.
struct RsShared {
    int *H, *W, *lsi, *sst;         // level tables (LDS)
    unsigned *mask;                 // [frames, frames] slot masks (LDS): bit 0 = current-frame points, bit 1 + w = slot w
    int zero_off;                   // byte offset of the zero row
    int l0, px0, npx;               // slab = levels [l0, L) = pixels [px0, px0 + npx) of a frame
};

__device__ __forceinline__ RsShared rs_setup(const Params &p, unsigned char *lds_raw, int slab_bytes, int elem_bytes)
{
    RsShared sh;
    sh.zero_off = slab_bytes;
    sh.mask = reinterpret_cast<unsigned *>(lds_raw + slab_bytes + kRsRowB);
    sh.H = reinterpret_cast<int *>(sh.mask + kRsMaxFrames * kRsMaxFrames);
    sh.W = sh.H + kSlabMaxLevels; sh.lsi = sh.W + kSlabMaxLevels; sh.sst = sh.lsi + kSlabMaxLevels;
    int *geo = sh.sst + kSlabMaxLevels;
    const int tid = threadIdx.x, L = p.L;
    // mask[t * frames + f]: which slots of frame t read frame f -- built once, so that the frame loop does not
    // chase the frame table through memory (frames <= kRsMaxFrames, window <= 31: host-checked)
    for (int i = tid; i < p.frames * p.frames; i += kRsThreads) {
        const int t = i / p.frames, f = i - t * p.frames;
        unsigned mk = (t == f) ? 1u : 0u;
        for (int w = 0; w < p.window; ++w) mk |= (p.ftab[t * p.window + w] == f) ? (2u << w) : 0u;
        sh.mask[i] = mk;
    }
    if (tid == 0) {
        const int l0 = first_slab_level(p, (slab_bytes - kRsSlack) / elem_bytes);
        const int px0 = l0 < L ? (int)p.lsi[l0] : 0;
        int npx = 0;
        for (int l = 0; l < L; ++l) {
            sh.H[l] = (int)p.shapes[2 * l]; sh.W[l] = (int)p.shapes[2 * l + 1]; sh.lsi[l] = (int)p.lsi[l];
            sh.sst[l] = (int)p.lsi[l] - px0;
            if (l >= l0) npx += sh.H[l] * sh.W[l];
        }
        geo[0] = l0; geo[1] = px0; geo[2] = npx;
    }
    if (tid < kRsRowB / 4) reinterpret_cast<float *>(lds_raw + sh.zero_off)[tid] = 0.f;
    __syncthreads();
    sh.l0 = geo[0]; sh.px0 = geo[1]; sh.npx = geo[2];
    return sh;
}

// ---- forward ------------------------------------------------------------------------------------------------------
// One quad = one row, 16 rows per wave tile, slab = levels >= l0 of one source frame (geometry: see the head of this
// file).  Round 3 changes against the round-2 kernel, each from a measurement:
//   * PER-LANE CORNER RECORDS: lane c of a quad turns ITS point into the four corner records (LDS / buffer byte address
//     with the out-of-map substitution, weight x attention) with plain VALU instructions; a step then only broadcasts
//     lane R's eight values to the quad (v_add_u32_dpp with the lane's slice offset, v_mov_b32_dpp).  Round 2 derived
//     each corner in the lane of the same number with six DPP-operand instructions per step and then broadcast all four:
//     16.5 VALU instructions per corner, now 7 (forward 0.430 -> 0.369 ms on the bench workload, same box).
//   * ABLATIONS (timing-only builds, MSDA_RS_EXP, profiles/r03_logs): without the level-0 (memory) corners the kernel takes
//     0.154 ms, without the slab (LDS) corners 0.377 ms = the time of the full kernel then (0.374): the LDS / VALU side is
//     completely hidden behind the level-0 gathers, i.e. behind the rate at which a CU's vector-memory path returns
//     scattered 128-byte lines that miss the L1 (86 k lines per CU and launch; 31 M L2 read requests per launch, 34 % of
//     the L2's peak request rate).  What helps: (a) a smaller L2 working set -- fewer (clip, head) pairs in flight per XCD
//     (1 tile per wave 0.345 ms against 0.374 with 2), which the host's choice of `parts` optimises; (b) PAIRS of memory
//     corners software-pipelined under the LDS pairs of the same slot (0.374 -> 0.337 ms; PL0 below).  What does not:
//     4-8 level-0 loads in flight per lane across whole LDS steps (0.380), non-temporal point loads, 512-thread
//     workgroups with a 256-VGPR budget (-DMSDA_RS_THREADS=512: 0.342 at 4 tiles per wave = the 1024-thread kernel;
//     gather pass 0.46 against 0.435).
struct RsRec { int a[4]; float w[4]; };      // this lane's point: byte address (without the lane's slice offset) and weight of corners 0..3

// The lane's own point at level `lvl` of source frame f: cuh:285-288 (pixel coords, range test), cuh:38-53 (floor,
// fractions), cuh:56-80 (per-corner validity).  Corners outside the map (or of a point outside the range) get the zero
// row of the slab / an out-of-range buffer offset: their loads return zeros.
struct RsGeom {
    float lh, lw, a;        // fractions and attention weight (all 0 for a point outside the range)
    int H, W, yl, bits;     // level shape, top tap row, validity bits of corners 0..3 (0: point skipped)
    int adr[4];             // byte address of each corner WITHOUT the lane's slice offset
};

template <int ROWSH>
__device__ __forceinline__ RsGeom rs_geometry(float x, float y, float a, int lvl, int l0, int fS, const RsShared &sh, int pixB)
{
    RsGeom g;
    g.H = sh.H[lvl]; g.W = sh.W[lvl];
    const int H = g.H, W = g.W;
    const bool slab = lvl >= l0;
    const int base = slab ? sh.sst[lvl] : fS + sh.lsi[lvl];
    const float h_im = __fsub_rn(__fmul_rn(y, (float)H), 0.5f);
    const float w_im = __fsub_rn(__fmul_rn(x, (float)W), 0.5f);
    const bool rng = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;     // false for NaN
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int yl = rng ? (int)hf : 0, xl = rng ? (int)wf : 0;
    g.yl = yl;
    g.lh = rng ? h_im - hf : 0.f; g.lw = rng ? w_im - wf : 0.f; g.a = rng ? a : 0.f;
    const bool vy0 = rng && yl >= 0, vy1 = rng && yl + 1 <= H - 1, vx0 = xl >= 0, vx1 = xl + 1 <= W - 1;
    const bool ok[4] = {vy0 && vx0, vy0 && vx1, vy1 && vx0, vy1 && vx1};
    g.bits = (ok[0] ? 1 : 0) | (ok[1] ? 2 : 0) | (ok[2] ? 4 : 0) | (ok[3] ? 8 : 0);
    const int p00 = base + yl * W + xl;
    const int pix[4] = {p00, p00 + 1, p00 + W, p00 + W + 1};
    const int none = slab ? sh.zero_off : (int)0x80000000u;
#pragma unroll
    for (int s = 0; s < 4; ++s)
        g.adr[s] = ok[s] ? (slab ? (pix[s] << ROWSH) : (int)((unsigned)pix[s] * (unsigned)pixB)) : none;
    return g;
}

// forward: corner addresses + weights (bilinear weight x attention weight)
template <int ROWSH>
__device__ __forceinline__ RsRec rs_records(float x, float y, float a, int lvl, int l0, int fS, const RsShared &sh, int pixB)
{
    const RsGeom g = rs_geometry<ROWSH>(x, y, a, lvl, l0, fS, sh, pixB);
    const float hh = 1.f - g.lh, hw = 1.f - g.lw;
    RsRec r;
#pragma unroll
    for (int s = 0; s < 4; ++s) r.a[s] = g.adr[s];
    r.w[0] = g.a * (hh * hw); r.w[1] = g.a * (hh * g.lw); r.w[2] = g.a * (g.lh * hw); r.w[3] = g.a * (g.lh * g.lw);
    return r;
}

// T = storage type of value / out, TL = of sampling_loc / attn_weight (T, or float with a 16-bit T)
template <typename T, typename TL, int NT, int PL0>       // PL0: the first slab level the software-pipelined slot body is compiled for (1 or 2)
__global__ void __launch_bounds__(kRsThreads, MSDA_RS_MIN_WAVES)
msda_fwd_rs_kernel(const Params p, int slab_bytes, int parts)
{
    constexpr int RPW = kRsRows, D = 32, ROWB = rs_row_bytes<T>(), ROWSH = ROWB == 128 ? 7 : 6;
    constexpr bool kHalf = sizeof(T) == 2;
    constexpr bool kPrefetch = MSDA_RS_PREFETCH != 0 && NT == 1;      // (more accumulator sets: the carried points spill)
    extern __shared__ __attribute__((aligned(128))) unsigned char lds_raw[];       // (no static LDS: the slab starts at 0)
    const int tid = threadIdx.x, lane = tid % kWave;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int L = p.L;
    T *slab = reinterpret_cast<T *>(lds_raw);
    const RsShared sh = rs_setup(p, lds_raw, slab_bytes, (int)sizeof(T));
    const int l0 = sh.l0;

    // workgroup -> (clip, head, part of the clip's tiles); wave -> up to NT tiles, 16 apart.  Blocks are dealt
    // round-robin to the 8 XCDs; each XCD takes a CONTIGUOUS run of (clip, head, part) triples, i.e. whole clips:
    // the parts of one (clip, head) share their slab and gathers in one L2, and -- unlike a head-per-XCD
    // mapping -- every XCD touches all heads, so the 1 KiB head pitch of the dense layout does not pin address
    // bits 7..9 and starve the L2 channels (speed only; results do not depend on placement)
    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u;
    const unsigned lin = xcd * (nwg / 8u) + min(xcd, nwg % 8u) + blockIdx.x / 8u;
    const int part = (int)(lin % (unsigned)parts), m = (int)((lin / (unsigned)parts) % (unsigned)p.M);
    const int clip = (int)(lin / ((unsigned)parts * (unsigned)p.M));
    const int tiles_per_group = (p.Lq + RPW - 1) / RPW, tiles_per_clip = p.frames * tiles_per_group;
    const int tpw = (tiles_per_clip + parts - 1) / parts;
    const int tile_lo = part * tpw + wave, tile_hi = min((part + 1) * tpw, tiles_per_clip);
    const int my_tiles = tile_lo < tile_hi ? (tile_hi - tile_lo + kRsWaves - 1) / kRsWaves : 0;      // <= NT (host)
    // tile k of this wave -> (frame t, first query q0); the tile loop is a RUNTIME loop (one copy of the body):
    // the NT accumulator sets are swapped in and out of a working set through uniform branches
    auto tile_of = [&](int k, int &t, int &q0) {
        const int ct = tile_lo + k * kRsWaves;
        t = ct / tiles_per_group;
        q0 = (ct - t * tiles_per_group) * RPW;
    };

    const int j = lane / 4, cor = lane & 3, hsw = j & 1;
    // byte offset of the lane's first 16-byte slice inside a pixel row; 4-byte types: the second slice at ^64 (LDS) / + delta2
    const int off1 = kHalf ? cor * 16 : cor * 16 + hsw * 64, delta2 = hsw ? -64 : 64;
    const int pixB = p.v_pix * (int)sizeof(T);
    const char *vbase = reinterpret_cast<const char *>(static_cast<const T *>(p.value) + clip * p.v_clip + m * p.v_head);
    const unsigned vbytes = (unsigned)(((int64_t)p.frames * p.S - 1) * pixB + ROWB);
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(vbase), 0, (int)vbytes, 0x00020000);
#endif

    float acc[NT][8];
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[k][c] = 0.f;
#ifdef MSDA_RS_TRACE
    const bool tr_on = blockIdx.x < 8 && tid == (blockIdx.x < 4 ? 64 : kRsThreads - 64);
    int tr_n = 0;
#endif
    rs_wave_priority(wave);
    MSDA_RTR(9);                    // prologue done

    for (int f = 0; f < p.frames; ++f) {
#if MSDA_RS_PREFETCH != 2
        __syncthreads();                                   // every wave is done with the previous slab
        MSDA_RTR(1);                // barrier: previous slab free
#endif
#if MSDA_RS_PREFETCH
        // The points of this wave's FIRST (tile, slot) of the frame are requested behind the slab's LDS-DMA and land while it does
        // (the timeline of round 4: 2.6 k of a frame's 24.8 k clocks were these loads, exposed in every wave at once after the
        // barrier).  Everything that reads LDS -- the slot mask -- comes before the DMA is issued: hipcc puts an s_waitcnt vmcnt(0)
        // in front of any LDS access it sees while an LDS-DMA is in flight.
        RawPoints<TL> pre;
        int pre_sl = -2;                                   // the slot the prefetched points belong to (-2: none)
        int64_t pre_idx0 = 0;
        if (kPrefetch && my_tiles > 0 && p.wide_loads) {
            int t0, q00;
            tile_of(0, t0, q00);
            const unsigned todo0 = __builtin_amdgcn_readfirstlane(sh.mask[t0 * p.frames + f]);
            if (todo0) {
                const int sl0 = (int)__builtin_ctz(todo0) - 1, P0 = sl0 < 0 ? p.PA : p.PB;
                if (P0 == 4 && (sl0 < 0 ? p.LA : L) * P0 == 16) {
                    const int64_t row0 = (((int64_t)clip * p.frames + t0) * p.Lq + q00 + j) * p.M + m;
                    pre_idx0 = row0 * ((sl0 < 0 ? p.LA : p.LB) * P0) + (sl0 < 0 ? 0 : sl0 * L * P0);
                    pre_sl = sl0;
                }
            }
        }
#endif
#if MSDA_RS_PREFETCH == 2
        if (pre_sl != -2) {
            int t0, q00;
            tile_of(0, t0, q00);
            pre = issue_slot_points<TL>(static_cast<const TL *>(pre_sl < 0 ? p.locA : p.locB), static_cast<const TL *>(pre_sl < 0 ? p.awA : p.awB),
                                        pre_idx0, cor, j < min(RPW, p.Lq - q00));
        }
        __syncthreads();                                   // every wave is done with the previous slab (the points fly meanwhile)
        MSDA_RTR(1);                // barrier: previous slab free
#endif
#if !defined(MSDA_RS_EXP) || MSDA_RS_EXP != 3
        if (l0 < L) rs_stage_slab<T>(p, slab, clip, m, f, sh.px0, sh.npx, wave, lane, !kPrefetch);
#endif
#if MSDA_RS_PREFETCH == 1
        if (pre_sl != -2) {
            int t0, q00;
            tile_of(0, t0, q00);
            pre = issue_slot_points<TL>(static_cast<const TL *>(pre_sl < 0 ? p.locA : p.locB), static_cast<const TL *>(pre_sl < 0 ? p.awA : p.awB),
                                        pre_idx0, cor, j < min(RPW, p.Lq - q00));
        }
#endif
#if MSDA_RS_PREFETCH
        if (kPrefetch) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        MSDA_RTR(2);                // slab pieces issued and landed
        __syncthreads();
        MSDA_RTR(3);                // barrier: slab complete
#if MSDA_RS_PREFETCH
        // (xs / ys / as are carried into the tile loop: the first (tile, slot) it runs is the one the points were requested for --
        // tile 0 has work in this frame, or nothing was requested -- and every later slot loads its own at its head)
        float xs[4], ys[4], as[4];
        bool have_points = false;
        if (pre_sl != -2) {
            int t0, q00;
            tile_of(0, t0, q00);
            finish_slot_points<TL>(pre, cor, j < min(RPW, p.Lq - q00), xs, ys, as);
            have_points = true;
        }
#endif
        const int fS = f * p.S;
#pragma unroll 1
        for (int k = 0; k < my_tiles; ++k) {
            int t, q0;
            tile_of(k, t, q0);
            unsigned todo = __builtin_amdgcn_readfirstlane(sh.mask[t * p.frames + f]);
            if (!todo) continue;
            const bool live = j < min(RPW, p.Lq - q0);
            const int64_t row = (((int64_t)clip * p.frames + t) * p.Lq + q0 + j) * p.M + m;
            float wacc[8];                                     // working accumulators = set k
            static_for<NT>([&](auto Kc) {
                constexpr int K = decltype(Kc)::value;
                if (k == K) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) wacc[c] = acc[K][c];
                }
            });
            // one corner: the lane's 8 channels of the row at byte address A, times W
            auto corner = [&](auto Sc, int A, float W) {      // (A by value: the experiments below may change it)
                constexpr bool SLAB = decltype(Sc)::value;
#if defined(MSDA_RS_EXP)          // timing experiments (wrong results; 1 / 2 / 6 live in rs_issue_row: no memory / LDS / any corner reads),
                                  // 4 = memory corners read one 16-byte slice instead of two, 5 = ... from one 32 KiB window
#if defined(__HIP_DEVICE_COMPILE__)
                if constexpr (!SLAB && MSDA_RS_EXP == 4) {
                    const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, A, 0, 0);
                    wacc[0] = fmaf(W, __uint_as_float(q.x), wacc[0]); wacc[1] = fmaf(W, __uint_as_float(q.y), wacc[1]);
                    wacc[2] = fmaf(W, __uint_as_float(q.z), wacc[2]); wacc[3] = fmaf(W, __uint_as_float(q.w), wacc[3]);
                    return;
                }
#endif
                if constexpr (!SLAB && MSDA_RS_EXP == 5) A &= 0x7fff;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
                const RsRaw<T> raw = rs_issue_row<T, SLAB>(rsrc, A, delta2);
                rs_fma_row<T>(raw, W, wacc);
#endif
                // corner by corner: the next corner's loads are not hoisted above these FMAs (measured in round 2:
                // eight loads in flight per wave only queue up in the LDS / TA pipes)
                asm volatile("" ::: "memory");
            };
            // one step = point R of the group, for the 16 rows of the wave: lane R's records go to the whole quad
            auto step = [&](auto Rc, auto Sc, const RsRec &rec) {
                constexpr int R = decltype(Rc)::value;
#pragma unroll
                for (int s = 0; s < 4; ++s) corner(Sc, quad_bcast<R>(rec.a[s]) + off1, quad_bcast<R>(rec.w[s]));
            };
#pragma unroll 1
            while (todo) {                                     // sl = -1: the tile's current-frame points
                const int sl = (int)__builtin_ctz(todo) - 1;
                todo &= todo - 1;
                const TL *loc = static_cast<const TL *>(sl < 0 ? p.locA : p.locB);
                const TL *aw = static_cast<const TL *>(sl < 0 ? p.awA : p.awB);
                const int P = sl < 0 ? p.PA : p.PB;
                const int LP = (sl < 0 ? p.LA : p.LB) * P;
                const int npts = (sl < 0 ? p.LA : L) * P;
                const int64_t idx0 = row * LP + (sl < 0 ? 0 : sl * L * P);
                const unsigned invP = (65536u + (unsigned)P - 1u) / (unsigned)P;      // kk / P for kk * P < 2^16
                const int first_slab_pt = l0 * P;              // points of levels >= l0 read the slab
                const bool wide = p.wide_loads && P == 4 && npts == 16;           // (uniform) see load_slot_points
#if MSDA_RS_PREFETCH
                if (wide && !have_points) load_slot_points<TL>(loc, aw, idx0, cor, live, xs, ys, as);
                have_points = false;
#else
                float xs[4], ys[4], as[4];
                if (wide) load_slot_points<TL>(loc, aw, idx0, cor, live, xs, ys, as);
#endif
                MSDA_RTR(4);        // slot: points loaded (first use waits)
#if MSDA_RS_PIPE
                if (wide && l0 == PL0) {
                    // 4 levels x 4 points, levels < l0 outside the slab.  Work units are corner PAIRS: 8 * l0 memory pairs (4
                    // buffer loads each) ride along the 8 * (4 - l0) LDS pairs -- one memory pair is consumed, and the next
                    // issued, after every third (l0 = 1) or every (l0 = 2) LDS pair
                    auto pipelined = [&](auto L0c) {
                        constexpr int L0 = decltype(L0c)::value, NM = 8 * L0, NL = 8 * (4 - L0), PERIOD = NL / NM;
                        RsRec rm[L0];
                        RsRaw<T> mv[2];                        // the memory pair in flight
                        // (the "+v" pins below are empty asm statements: they tie a value to a point of the instruction stream --
                        // after the FMAs that produced wacc, before the address arithmetic / loads that use it -- so that the
                        // compiler neither computes all records of a slot up front nor lets eight loads pile up in registers)
                        auto issue = [&](auto Jc) {
                            constexpr int J = decltype(Jc)::value, G = J / 8, R = (J % 8) / 2, S0 = 2 * (J % 2);
                            if constexpr (J % 8 == 0) {
                                asm volatile("" : "+v"(xs[G]), "+v"(ys[G]), "+v"(as[G]) : "v"(wacc[0]));
                                rm[G] = rs_records<ROWSH>(xs[G], ys[G], as[G], G, l0, fS, sh, pixB);
                            }
                            int a0 = quad_bcast<R>(rm[G].a[S0]) + off1, a1 = quad_bcast<R>(rm[G].a[S0 + 1]) + off1;
                            asm volatile("" : "+v"(a0), "+v"(a1) : "v"(wacc[7]));
#if defined(__HIP_DEVICE_COMPILE__)
                            mv[0] = rs_issue_row<T, false>(rsrc, a0, delta2);
                            mv[1] = rs_issue_row<T, false>(rsrc, a1, delta2);
#endif
                        };
                        auto consume = [&](auto Jc) {
                            constexpr int J = decltype(Jc)::value, G = J / 8, R = (J % 8) / 2, S0 = 2 * (J % 2);
                            const float w0 = quad_bcast<R>(rm[G].w[S0]), w1 = quad_bcast<R>(rm[G].w[S0 + 1]);
                            rs_fma_row<T>(mv[0], w0, wacc);
                            rs_fma_row<T>(mv[1], w1, wacc);
                        };
                        issue(std::integral_constant<int, 0>{});
                        static_for<4 - L0>([&](auto Gc) {
                            constexpr int GL = decltype(Gc)::value + L0;       // a level of the slab
                            asm volatile("" : "+v"(xs[GL]), "+v"(ys[GL]), "+v"(as[GL]) : "v"(wacc[0]));
                            const RsRec rl = rs_records<ROWSH>(xs[GL], ys[GL], as[GL], GL, l0, fS, sh, pixB);
                            static_for<8>([&](auto Hc) {
                                constexpr int R = decltype(Hc)::value / 2, S0 = 2 * (decltype(Hc)::value % 2);
                                constexpr int HS = (GL - L0) * 8 + decltype(Hc)::value;      // LDS pair index
#if MSDA_RS_PAIR
                                {   // both corners of the pair requested before either is consumed
                                    const int A0 = quad_bcast<R>(rl.a[S0]) + off1, A1 = quad_bcast<R>(rl.a[S0 + 1]) + off1;
                                    const float W0 = quad_bcast<R>(rl.w[S0]), W1 = quad_bcast<R>(rl.w[S0 + 1]);
#if defined(__HIP_DEVICE_COMPILE__)
                                    const RsRaw<T> r0 = rs_issue_row<T, true>(rsrc, A0, delta2);
                                    const RsRaw<T> r1 = rs_issue_row<T, true>(rsrc, A1, delta2);
                                    rs_fma_row<T>(r0, W0, wacc);
                                    rs_fma_row<T>(r1, W1, wacc);
#endif
                                    asm volatile("" ::: "memory");
                                }
#else
                                corner(std::true_type{}, quad_bcast<R>(rl.a[S0]) + off1, quad_bcast<R>(rl.w[S0]));
                                corner(std::true_type{}, quad_bcast<R>(rl.a[S0 + 1]) + off1, quad_bcast<R>(rl.w[S0 + 1]));
#endif
                                if constexpr ((HS + 1) % PERIOD == 0) {
                                    constexpr int J = (HS + 1) / PERIOD - 1;
                                    consume(std::integral_constant<int, J>{});
                                    if constexpr (J + 1 < NM) issue(std::integral_constant<int, J + 1>{});
                                }
                            });
                        });
                    };
                    pipelined(std::integral_constant<int, PL0>{});
                } else
#endif
                {
#pragma unroll 1
                    for (int g0 = 0; g0 < npts; g0 += 4) {
                        const int kk = g0 + cor;
                        float x = -10.f, y = -10.f, a = 0.f;       // far outside every map
                        if (wide) {
                            x = get4(xs, g0 >> 2); y = get4(ys, g0 >> 2); a = get4(as, g0 >> 2);
                        } else if (live && kk < npts) {
                            load_xy(loc + 2 * (idx0 + kk), x, y);
                            a = Store<TL>::get(aw + idx0 + kk);
                        }
                        const int lvl = min((int)(((unsigned)kk * invP) >> 16), L - 1);
                        const RsRec rec = rs_records<ROWSH>(x, y, a, lvl, l0, fS, sh, pixB);
                        if (g0 >= first_slab_pt) {                 // the whole group reads the slab (uniform)
                            static_for<4>([&](auto Rc) { if (g0 + decltype(Rc)::value < npts) step(Rc, std::true_type{}, rec); });
                        } else if (g0 + 3 < first_slab_pt) {       // the whole group reads memory
                            static_for<4>([&](auto Rc) { if (g0 + decltype(Rc)::value < npts) step(Rc, std::false_type{}, rec); });
                        } else {
                            static_for<4>([&](auto Rc) {
                                constexpr int R = decltype(Rc)::value;
                                if (g0 + R >= npts) return;
                                if (g0 + R >= first_slab_pt) step(Rc, std::true_type{}, rec); else step(Rc, std::false_type{}, rec);
                            });
                        }
                    }
                }
            }
            MSDA_RTR(5);            // tile done for this frame
            static_for<NT>([&](auto Kc) {
                constexpr int K = decltype(Kc)::value;
                if (k == K) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[K][c] = wacc[c];
                }
            });
        }
    }
    MSDA_RTR(6);
    static_for<NT>([&](auto Kc) {
        constexpr int K = decltype(Kc)::value;
        if (K >= my_tiles) return;
        int t, q0;
        tile_of(K, t, q0);
        if (j < min(RPW, p.Lq - q0)) {
            const int64_t row = (((int64_t)clip * p.frames + t) * p.Lq + q0 + j) * p.M + m;
            T *o = static_cast<T *>(p.out) + row * D;
            if constexpr (kHalf) {
                Store<T>::store(o + cor * 8, acc[K]);       // channels [8c, 8c+8): one 16-byte store
            } else {
                const float a1[4] = {acc[K][0], acc[K][1], acc[K][2], acc[K][3]}, a2[4] = {acc[K][4], acc[K][5], acc[K][6], acc[K][7]};
                Store<T>::store(o + off1 / 4, a1);
                Store<T>::store(o + (off1 + delta2) / 4, a2);
            }
        }
    });
#ifdef MSDA_RS_TRACE
    if (tr_on) g_rs_trace_n[blockIdx.x] = tr_n;
#endif
}

// Backward gather pass (grad_loc / grad_attn) on the resident slab: same workgroup / tile / quad geometry as
// msda_fwd_rs_kernel, but nothing is carried across source frames -- every (tile, slot) writes its own gradients --
// so there are no accumulator sets and a wave may take any number of tiles.  Per point the four dots
// <grad_out row, corner k> (cuh:123-158) are 8 FMAs per corner and lane, reduced over the quad with two DPP adds;
// lane R of the quad keeps the dots of point R, and after the group's four points every lane finishes ITS point and
// stores its (grad_x, grad_y, grad_attn) directly: the 4 points of a group are 32 + 16 contiguous bytes per row.
// Also leaves the per-point culling records (top tap row as int16) the scatter pass reads.
template <typename T, typename TL, int PL0>      // T: value / grad_out, TL: sampling_loc / attn_weight and their gradients
__global__ void __launch_bounds__(kRsThreads, MSDA_RS_MIN_WAVES)
msda_bwd_rs_kernel(const Params p, int slab_bytes, int parts, int frame_split)
{
    constexpr int RPW = kRsRows, D = 32, ROWB = rs_row_bytes<T>(), ROWSH = ROWB == 128 ? 7 : 6;
    constexpr bool kHalf = sizeof(T) == 2;
    extern __shared__ __attribute__((aligned(128))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid % kWave;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int L = p.L, VL = p.LA + p.LB;
    // the scatter pass that follows draws its work tickets from the head of the workspace (see msda_bwd_slab_kernel)
    if (blockIdx.x == 0 && tid < MSDA_BWD_WORKSPACE_BYTES / 4 && p.workspace) p.workspace[tid] = 0u;
    T *slab = reinterpret_cast<T *>(lds_raw);
    const RsShared sh = rs_setup(p, lds_raw, slab_bytes, (int)sizeof(T));
    const int l0 = sh.l0;

    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u;       // clip-major XCD mapping, as in the forward
    const unsigned lin = xcd * (nwg / 8u) + min(xcd, nwg % 8u) + blockIdx.x / 8u;
    // frame_split (round 4): a workgroup = (clip, head, SOURCE FRAME, part of the clip's tiles) instead of (clip, head, part)
    // walking the frames: nothing is carried from one source frame to the next in this pass, so the frame loop can be a grid
    // dimension -- one slab per workgroup, no barrier or staging between frames
    const int part = (int)(lin % (unsigned)parts);
    const unsigned rest = lin / (unsigned)parts;
    const int f_only = frame_split ? (int)(rest % (unsigned)p.frames) : -1;
    const unsigned rest2 = frame_split ? rest / (unsigned)p.frames : rest;
    const int m = (int)(rest2 % (unsigned)p.M);
    const int clip = (int)(rest2 / (unsigned)p.M);
    const int tiles_per_group = (p.Lq + RPW - 1) / RPW, tiles_per_clip = p.frames * tiles_per_group;
    const int tpw = (tiles_per_clip + parts - 1) / parts;
    const int tile_lo = part * tpw + wave, tile_hi = min((part + 1) * tpw, tiles_per_clip);

    const int j = lane / 4, cor = lane & 3, hsw = j & 1;
    // byte offset of the lane's first 16-byte slice inside a pixel row; 4-byte types: the second slice at + delta2
    const int off1 = kHalf ? cor * 16 : cor * 16 + hsw * 64, delta2 = hsw ? -64 : 64;
    const int pixB = p.v_pix * (int)sizeof(T);
    const char *vbase = reinterpret_cast<const char *>(static_cast<const T *>(p.value) + clip * p.v_clip + m * p.v_head);
    const unsigned vbytes = (unsigned)(((int64_t)p.frames * p.S - 1) * pixB + ROWB);
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(vbase), 0, (int)vbytes, 0x00020000);
#endif
    const bool records = p.bbox != nullptr;        // per-point culling records (host: only with cull_points)
    rs_wave_priority(wave);

    for (int f = frame_split ? f_only : 0; f < (frame_split ? f_only + 1 : p.frames); ++f) {
        __syncthreads();                                   // every wave is done with the previous slab
        if (l0 < L) rs_stage_slab<T>(p, slab, clip, m, f, sh.px0, sh.npx, wave, lane);
        __syncthreads();
        const int fS = f * p.S;
#pragma unroll 1
        for (int ct = tile_lo; ct < tile_hi; ct += kRsWaves) {
            const int t = ct / tiles_per_group, q0 = (ct - t * tiles_per_group) * RPW;
            unsigned todo = __builtin_amdgcn_readfirstlane(sh.mask[t * p.frames + f]);
            if (!todo) continue;
            const bool live = j < min(RPW, p.Lq - q0);
            const int64_t group = (int64_t)clip * p.frames + t;
            const int64_t row = ((group * p.Lq) + q0 + j) * p.M + m;
            // this row's grad_out: channels [4c, 4c+4) of both halves, as the value slices are read
            float g[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) g[c] = 0.f;
            if (live) {
                const T *go = static_cast<const T *>(p.grad_out) + row * D;
                if constexpr (kHalf) {
                    Store<T>::load(go + cor * 8, g);
                } else {
                    const float4 g1 = *reinterpret_cast<const float4 *>(go + off1 / 4);
                    const float4 g2 = *reinterpret_cast<const float4 *>(go + (off1 + delta2) / 4);
                    g[0] = g1.x; g[1] = g1.y; g[2] = g1.z; g[3] = g1.w; g[4] = g2.x; g[5] = g2.y; g[6] = g2.z; g[7] = g2.w;
                }
            }
#pragma unroll 1
            while (todo) {                                     // sl = -1: the tile's current-frame points
                const int sl = (int)__builtin_ctz(todo) - 1;
                todo &= todo - 1;
                const TL *loc = static_cast<const TL *>(sl < 0 ? p.locA : p.locB);
                const TL *aw = static_cast<const TL *>(sl < 0 ? p.awA : p.awB);
                TL *gloc = static_cast<TL *>(sl < 0 ? p.glocA : p.glocB);
                TL *gaw = static_cast<TL *>(sl < 0 ? p.gawA : p.gawB);
                const int P = sl < 0 ? p.PA : p.PB;
                const int LP = (sl < 0 ? p.LA : p.LB) * P;
                const int npts = (sl < 0 ? p.LA : L) * P;
                const int vl0 = sl < 0 ? 0 : p.LA + sl * L;   // virtual level of the slot's level 0
                const int64_t idx0 = row * LP + (sl < 0 ? 0 : sl * L * P);
                const unsigned invP = (65536u + (unsigned)P - 1u) / (unsigned)P;      // kk / P for kk * P < 2^16
                const int first_slab_pt = l0 * P;              // points of levels >= l0 read the slab
                // 4 levels x 4 points (every DeVIS call): the slot's results are kept in registers and leave as whole rows --
                // per quad 128 contiguous bytes of grad_loc and 64 of grad_attn in three 16-byte stores per lane, and the
                // culling records of a level as one 8-byte store per row -- instead of 4- and 2-byte stores group by group
                // (the 16-byte grad_attn pieces and 2-byte records were written back as partial lines: WRITE_SIZE 572 MB
                // for 309 MB of results)
                const bool wide = p.wide_stores && P == 4 && npts == 16;
                float wx[4] = {0.f, 0.f, 0.f, 0.f}, wy[4] = {0.f, 0.f, 0.f, 0.f}, wa[4] = {0.f, 0.f, 0.f, 0.f};
                int wr[4] = {0, 0, 0, 0};
                const bool wide_ld = p.wide_loads && P == 4 && npts == 16;
                float xs[4], ys[4], as[4];
                if (wide_ld) load_slot_points<TL>(loc, aw, idx0, cor, live, xs, ys, as);
                // every lane finishes its own point (cuh:123-158 on the reduced dots k[]; dots of corners outside the map are 0:
                // their loads returned zeros)
                auto finish = [&](const RsGeom &pt, const float (&k)[4], float &gx, float &gy, float &g_aw) {
                    const float lh = pt.lh, lw = pt.lw, hh = 1.f - lh, hw = 1.f - lw;
                    g_aw = (hh * hw) * k[0] + (hh * lw) * k[1] + (lh * hw) * k[2] + (lh * lw) * k[3];
                    const float g_w = hh * (k[1] - k[0]) + lh * (k[3] - k[2]);
                    const float g_h = hw * (k[2] - k[0]) + lw * (k[3] - k[1]);
                    gx = (float)pt.W * g_w * pt.a; gy = (float)pt.H * g_h * pt.a;
                };
                bool done = false;
#if MSDA_RS_PIPE
                if (wide && wide_ld && l0 == PL0) {
                    // 4 levels x 4 points, levels < l0 outside the slab: their 8 * l0 corner PAIRS (4 buffer loads each) ride along
                    // the 8 * (4 - l0) LDS pairs, one consumed -- and the next issued -- after every third (l0 = 1) or every
                    // (l0 = 2) LDS pair, so that a wave overlaps its own L2 round trips with its own slab work (see the forward)
                    constexpr int L0 = PL0, NM = 8 * L0, NL = 8 * (4 - L0), PERIOD = NL / NM;
                    RsGeom gm[L0];
                    float km[L0][4], dm[4];          // dots of this lane's points of the memory levels; of the point in flight
#pragma unroll
                    for (int G = 0; G < L0; ++G)
#pragma unroll
                        for (int u = 0; u < 4; ++u) km[G][u] = 0.f;
                    RsRaw<T> mv[2];
                    auto issue = [&](auto Jc) {
                        constexpr int J = decltype(Jc)::value, G = J / 8, R = (J % 8) / 2, S0 = 2 * (J % 2);
                        if constexpr (J % 8 == 0) {
                            const float pin = wa[0];
                            asm volatile("" : "+v"(xs[G]), "+v"(ys[G]), "+v"(as[G]) : "v"(pin));
                            gm[G] = rs_geometry<ROWSH>(xs[G], ys[G], as[G], G, l0, fS, sh, pixB);
                            wr[G] = gm[G].bits ? min(gm[G].yl, 32767) : kNoRow16;
                        }
                        int a0 = quad_bcast<R>(gm[G].adr[S0]) + off1, a1 = quad_bcast<R>(gm[G].adr[S0 + 1]) + off1;
                        const float pin = wa[0];
                        asm volatile("" : "+v"(a0), "+v"(a1) : "v"(pin));        // (ties the loads to this point of the stream)
#if defined(__HIP_DEVICE_COMPILE__)
                        mv[0] = rs_issue_row<T, false>(rsrc, a0, delta2);
                        mv[1] = rs_issue_row<T, false>(rsrc, a1, delta2);
#endif
                    };
                    auto consume = [&](auto Jc) {
                        constexpr int J = decltype(Jc)::value, G = J / 8, R = (J % 8) / 2, S0 = 2 * (J % 2);
                        dm[S0] = rs_dot_row<T>(mv[0], g); dm[S0 + 1] = rs_dot_row<T>(mv[1], g);
                        if constexpr (S0 == 2) {
                            quad_sum4(dm);
                            const bool me = cor == R;
#pragma unroll
                            for (int u = 0; u < 4; ++u) km[G][u] = me ? dm[u] : km[G][u];
                        }
                    };
                    issue(std::integral_constant<int, 0>{});
                    static_for<4 - L0>([&](auto Gc) {
                        constexpr int GL = decltype(Gc)::value + L0;           // a level of the slab
                        const float pin = wa[0];
                        asm volatile("" : "+v"(xs[GL]), "+v"(ys[GL]), "+v"(as[GL]) : "v"(pin));
                        const RsGeom gl = rs_geometry<ROWSH>(xs[GL], ys[GL], as[GL], GL, l0, fS, sh, pixB);
                        wr[GL] = gl.bits ? min(gl.yl, 32767) : kNoRow16;
                        float kl[4] = {0.f, 0.f, 0.f, 0.f}, d[4];
                        static_for<8>([&](auto Hc) {
                            constexpr int R = decltype(Hc)::value / 2, S0 = 2 * (decltype(Hc)::value % 2);
                            constexpr int HS = (GL - L0) * 8 + decltype(Hc)::value;      // LDS pair index
#pragma unroll
                            for (int s = S0; s < S0 + 2; ++s) {
#if defined(__HIP_DEVICE_COMPILE__)
                                const RsRaw<T> raw = rs_issue_row<T, true>(rsrc, quad_bcast<R>(gl.adr[s]) + off1, delta2);
                                d[s] = rs_dot_row<T>(raw, g);
#endif
                                asm volatile("" ::: "memory");
                            }
                            if constexpr (S0 == 2) {
                                quad_sum4(d);
                                const bool me = cor == R;
#pragma unroll
                                for (int u = 0; u < 4; ++u) kl[u] = me ? d[u] : kl[u];
                            }
                            if constexpr ((HS + 1) % PERIOD == 0) {
                                constexpr int J = (HS + 1) / PERIOD - 1;
                                consume(std::integral_constant<int, J>{});
                                if constexpr (J + 1 < NM) issue(std::integral_constant<int, J + 1>{});
                            }
                        });
                        finish(gl, kl, wx[GL], wy[GL], wa[GL]);
                    });
#pragma unroll
                    for (int G = 0; G < L0; ++G) finish(gm[G], km[G], wx[G], wy[G], wa[G]);
                    done = true;
                }
#endif
#pragma unroll 1
                for (int g0 = 0; g0 < (done ? 0 : npts); g0 += 4) {
                    const int kk = g0 + cor;
                    const bool mine = live && kk < npts;
                    float x = -10.f, y = -10.f, a = 0.f;       // far outside every map
                    if (wide_ld) {
                        x = get4(xs, g0 >> 2); y = get4(ys, g0 >> 2); a = get4(as, g0 >> 2);
                    } else if (mine) {
                        load_xy(loc + 2 * (idx0 + kk), x, y);
                        a = Store<TL>::get(aw + idx0 + kk);
                    }
                    const int lvl = min((int)(((unsigned)kk * invP) >> 16), L - 1);
                    // own point: corner addresses, fractions, validity (rs_geometry) -- the quad reads lane R's addresses in step R
                    const RsGeom pt = rs_geometry<ROWSH>(x, y, a, lvl, l0, fS, sh, pixB);
                    const int yl = pt.yl, bits = pt.bits;
                    if (wide) set4(wr, g0 >> 2, bits ? min(yl, 32767) : kNoRow16);
                    if (records && mine && !wide) {      // the point's top tap row, for the scatter's band test
                        const int pin = kk - lvl * P;
                        short *rec = reinterpret_cast<short *>(p.bbox + (((group * p.M + m) * VL + vl0 + lvl) * p.Lq + q0 + j) * 2);
                        rec[pin] = bits ? (short)min(yl, 32767) : (short)kNoRow16;
                        if (pin == 0)
                            for (int u = P; u < 4; ++u) rec[u] = (short)kNoRow16;
                    }
                    float k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f;      // the dots of THIS lane's point
                    auto step = [&](auto Rc, auto Sc) {
                        constexpr int R = decltype(Rc)::value;
                        constexpr bool SLAB = decltype(Sc)::value;
                        float d[4];
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
#if defined(__HIP_DEVICE_COMPILE__)
                            const RsRaw<T> raw = rs_issue_row<T, SLAB>(rsrc, quad_bcast<R>(pt.adr[s]) + off1, delta2);
                            d[s] = rs_dot_row<T>(raw, g);
#endif
                            // (corner by corner; measured in round 2: all eight loads of a point ahead of the dots is slower)
                            asm volatile("" ::: "memory");
                        }
                        quad_sum4(d);
                        const bool me = cor == R;
                        k0 = me ? d[0] : k0; k1 = me ? d[1] : k1; k2 = me ? d[2] : k2; k3 = me ? d[3] : k3;
                    };
                    if (g0 >= first_slab_pt) {                 // the whole group reads the slab (uniform)
                        static_for<4>([&](auto Rc) { if (g0 + decltype(Rc)::value < npts) step(Rc, std::true_type{}); });
                    } else if (g0 + 3 < first_slab_pt) {       // the whole group reads memory
                        static_for<4>([&](auto Rc) { if (g0 + decltype(Rc)::value < npts) step(Rc, std::false_type{}); });
                    } else {
                        static_for<4>([&](auto Rc) {
                            constexpr int R = decltype(Rc)::value;
                            if (g0 + R >= npts) return;
                            if (g0 + R >= first_slab_pt) step(Rc, std::true_type{}); else step(Rc, std::false_type{});
                        });
                    }
                    {
                        const float kq[4] = {k0, k1, k2, k3};
                        float gx, gy, g_aw;
                        finish(pt, kq, gx, gy, g_aw);
                        if (wide) {
                            set4(wx, g0 >> 2, gx); set4(wy, g0 >> 2, gy); set4(wa, g0 >> 2, g_aw);
                        } else if (mine) {
                            Store<TL>::put(gloc + 2 * (idx0 + kk), gx);
                            Store<TL>::put(gloc + 2 * (idx0 + kk) + 1, gy);
                            Store<TL>::put(gaw + idx0 + kk, g_aw);
                        }
                    }
                }
                if (wide) {
                    // lane c held point c of every level; after the transposes it holds the four points of level c
                    quad_transpose4(wx, cor); quad_transpose4(wy, cor); quad_transpose4(wa, cor); quad_transpose4(wr, cor);
                    if (live) {
                        const float xy[8] = {wx[0], wy[0], wx[1], wy[1], wx[2], wy[2], wx[3], wy[3]};
                        TL *gl = gloc + 2 * (idx0 + 4 * cor);
                        if constexpr (sizeof(TL) == 2) {
                            Store<TL>::store(gl, xy);
                        } else {
                            // (non-temporal: the 309 MB of results must not evict the level-0 lines the gathers live on)
                            typedef float f32x4 __attribute__((ext_vector_type(4)));
                            __builtin_nontemporal_store((f32x4){xy[0], xy[1], xy[2], xy[3]}, reinterpret_cast<f32x4 *>(gl));
                            __builtin_nontemporal_store((f32x4){xy[4], xy[5], xy[6], xy[7]}, reinterpret_cast<f32x4 *>(gl + 4));
                        }
                        if constexpr (sizeof(TL) == 2) {
                            SlabStore<TL>::store(gaw + idx0 + 4 * cor, wa);
                        } else {
                            typedef float f32x4 __attribute__((ext_vector_type(4)));
                            __builtin_nontemporal_store((f32x4){wa[0], wa[1], wa[2], wa[3]}, reinterpret_cast<f32x4 *>(gaw + idx0 + 4 * cor));
                        }
                        if (records && ((p.rec_mask >> cor) & 1u))     // (not for levels the owner-computes scatter does not walk or walks whole)
                            *reinterpret_cast<int2 *>(p.bbox + (((group * p.M + m) * VL + vl0 + cor) * p.Lq + q0 + j) * 2) =
                                make_int2((wr[0] & 0xffff) | (wr[1] << 16), (wr[2] & 0xffff) | (wr[3] << 16));
                    }
                }
            }
        }
    }
}

template <typename T, typename TL, int NT, int PL0>
int fwd_rs(const Params &p, int parts, unsigned grid, hipStream_t stream, const char *what)
{
    static LdsGrant granted;
    const size_t total = (size_t)kRsSlabBytes + kRsTailBytes;
    const auto kern = &msda_fwd_rs_kernel<T, TL, NT, PL0>;
    if (const int rc = grant_lds(reinterpret_cast<const void *>(kern), total, granted, "the resident-slab forward kernel")) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kRsThreads), total, stream, p, kRsSlabBytes, parts);
    return check_launch(what);
}

template <typename T, typename TL, int PL0>
int fwd_rs_nt(int nt, const Params &p, int parts, unsigned grid, hipStream_t stream)
{
    switch (nt) {
        case 4: return fwd_rs<T, TL, 4, PL0>(p, parts, grid, stream, "msda forward (resident-slab kernel, 4 tiles per wave)");
        case 2: return fwd_rs<T, TL, 2, PL0>(p, parts, grid, stream, "msda forward (resident-slab kernel, 2 tiles per wave)");
        default: return fwd_rs<T, TL, 1, PL0>(p, parts, grid, stream, "msda forward (resident-slab kernel, 1 tiles per wave)");
    }
}

template <typename T, typename TL>
int fwd_rs_l0(int nt, int pl0, const Params &p, int parts, unsigned grid, hipStream_t stream)
{
    return pl0 == 2 ? fwd_rs_nt<T, TL, 2>(nt, p, parts, grid, stream) : fwd_rs_nt<T, TL, 1>(nt, p, parts, grid, stream);
}

template <typename T, typename TL, int PL0>
int bwd_rs(const Params &p, int parts, unsigned grid, hipStream_t stream, int frame_split)
{
    static LdsGrant granted;
    const size_t total = (size_t)kRsSlabBytes + kRsTailBytes;
    const auto kern = &msda_bwd_rs_kernel<T, TL, PL0>;
    if (const int rc = grant_lds(reinterpret_cast<const void *>(kern), total, granted, "the resident-slab gather-pass kernel")) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kRsThreads), total, stream, p, kRsSlabBytes, parts, frame_split);
    return check_launch(frame_split ? "msda backward (resident-slab kernel, grad_loc/grad_attn, one source frame per workgroup)"
                                    : "msda backward (resident-slab kernel, grad_loc/grad_attn)");
}

template <typename T, typename TL>
int bwd_rs_l0(int pl0, const Params &p, int parts, unsigned grid, hipStream_t stream, int frame_split)
{
    return pl0 == 2 ? bwd_rs<T, TL, 2>(p, parts, grid, stream, frame_split) : bwd_rs<T, TL, 1>(p, parts, grid, stream, frame_split);
}

}  // namespace

int launch_fwd_rs(int dtype, int nt, int first_slab_level, const Params &p, int parts, unsigned grid, hipStream_t stream)
{
    return dispatch_types(dtype, [&](auto t, auto tl) {
        return fwd_rs_l0<typename decltype(t)::type, typename decltype(tl)::type>(nt, first_slab_level, p, parts, grid, stream);
    });
}

int launch_bwd_rs(int dtype, int first_slab_level, const Params &p, int parts, unsigned grid, hipStream_t stream, int frame_split)
{
    return dispatch_types(dtype, [&](auto t, auto tl) {
        return bwd_rs_l0<typename decltype(t)::type, typename decltype(tl)::type>(first_slab_level, p, parts, grid, stream, frame_split);
    });
}

}  // namespace msda

#ifdef MSDA_RS_TRACE
extern "C" int msda_debug_trace_rs(unsigned long long *dst, int *counts)
{
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(msda::g_rs_trace), sizeof(unsigned long long) * 8 * msda::kRsTraceLen) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(counts, HIP_SYMBOL(msda::g_rs_trace_n), sizeof(int) * 8) != hipSuccess) return -2;
    return msda::kRsTraceLen;
}
#endif
