// msda_mfma.hip -- grad_value of the COARSE pyramid levels on the matrix pipe (round 6).
//
// Replaces, for the trailing levels of the pyramid whose pixels together fit ~300 (levels 2-3 of the 360x640 DeVIS pyramid:
// 12x20 + 6x10), the atomicAdd scatter of the reference (ms_deform_im2col_cuda.cuh:87-159, 301-403) and this library's
// owner-computes list walk (msda_scatter.hip), which spends 40 % of its time on those two levels although they hold 6 % of the
// pixels: every query puts 4 points on each of them, so their per-pixel lists are long and their items are the heaviest.
//
//   grad_value[P pixels x 32 channels] of one (clip, source frame, head)  =  A[P x K] . G[K x 32]
//     K  = the (source, query) groups that sample the frame: its own current-frame points + every temporal slot (t, w) with
//          frame_table[t, w] == frame;  G[k] = the group's grad_out row (this head's 32 channels);
//     A[pix, k] = sum over the group's <= 4 points on the level of  attention x bilinear weight of the point at pixel pix:
//          <= 16 non-zeros per column, built on the fly, never stored outside the LDS.
//   fp32 storage: A and G are split a = a_hi + a_lo (bf16 each, round to nearest) and three products run on
//   v_mfma_f32_32x32x16_bf16 -- a_hi.g_hi + a_lo.g_hi + a_hi.g_lo, fp32 accumulation; the dropped a_lo.g_lo and the split
//   residues are <= 2^-17 of a term (measured: 5e-6 of the gradient's scale, scripts/ubench/mfma_scatter.hip; the fp32 tests allow
//   2e-5).  16-bit storage: G is used as stored (bf16 -> the bf16 instruction, f16 -> v_mfma_f32_32x32x16_f16), A split in two.
//
// Workgroup = NW waves working on one item (clip, frame, head) at a time, each wave on every NW-th STEP of 16 groups with
// accumulators for ALL pixel tiles of its own, no barrier inside an item's loop:
//   (1) thread (g, pt) = point pt of group g: loads (one step ahead, unconditional so that the compiler can count them), geometry
//       in the reference's arithmetic (cuh:285-288);
//   (2) the quad's four points are merged in registers: each thread evaluates all four points' bilinear "tents"
//       max(0, 1 - |h_j - row|) . max(0, 1 - |w_j - col|) at its own four corner pixels (quad_perm broadcasts, point order), so
//       two threads whose corners coincide hold BIT-IDENTICAL totals and may both write the cell;
//   (3) totals -> hi + lo 16-bit cells of the wave's private A tiles ([2 k-chunks][pixel rows][8 k]: the MFMA operand of a 32-pixel
//       tile is one conflict-free ds_read_b128 per lane, a point's four cells are cell00 + {0, 16, 16 W, 16 W + 16});
//   (4) G rows straight from memory into the B-operand layout (8 coalesced row segments per wave), split in registers;
//   (5) per pixel tile the 2-3 products; (6) the cells are written back to zero.
// At the end of an item the waves' accumulators are added through the LDS and stored as 128-byte rows; grad_value of these
// levels is OVERWRITTEN (include/msda.h), in fp32 or in the storage type.  The owner-computes scatter runs with
// Params::own_levels = first level handled here and never sees these levels.
#include "msda_common.h"
#include <algorithm>

#ifndef MSDA_MFMA_TB
#define MSDA_MFMA_TB 2      // pixel tiles whose operands are read together (experiment builds: see profiles/r06_logs)
#endif
// Timing-only build (-DMSDA_MFMA_TRACE beside -DMSDA_TIMING_ONLY_BUILD; scripts/mfma_trace.py): cycle stamps at the phase boundaries
// of an item; every wave leaves its per-phase sums in the first pixels of its frame's LEVEL-0 grad_value (run with MSDA_SCATTER_PART=2).
#if defined(MSDA_MFMA_TRACE)
#define MSDA_TR(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long tr_now = __builtin_readcyclecounter(); \
                     tr[k] += (unsigned)(tr_now - tr_t); tr_t = tr_now; __builtin_amdgcn_sched_barrier(0); }
#else
#define MSDA_TR(k)
#endif
#ifndef MSDA_MFMA_RED
#define MSDA_MFMA_RED 2     // accumulator rows a wave sums at a time in the reduction
#endif
#ifndef MSDA_MFMA_SB
#define MSDA_MFMA_SB 1      // scheduling barrier behind every batch of tiles (bounds the live operand registers)
#endif

namespace msda {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int CTRL> __device__ __forceinline__ float quad_bcast(float v)       // lane CTRL's value of every quad
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL * 0x55, 0xf, 0xf, true));
}
__device__ __forceinline__ float tent(float d)      // max(0, 1 - |d|): one v_sub with |src| and clamp
{
    return __builtin_amdgcn_fmed3f(1.f - __builtin_fabsf(d), 0.f, 1.f);
}

// element type of the matrix instruction for a storage type
template <typename T> struct Mx { using E = __bf16; using V = bf16x8_t; static constexpr int kProducts = 3; };
template <> struct Mx<bf16_t> { using E = __bf16; using V = bf16x8_t; static constexpr int kProducts = 2; };
template <> struct Mx<f16_t> { using E = _Float16; using V = f16x8_t; static constexpr int kProducts = 2; };

__device__ __forceinline__ f32x16_t mma(bf16x8_t a, bf16x8_t b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16_t mma(f16x8_t a, f16x8_t b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// raw G element -> float (for the split) or straight into the operand
__device__ __forceinline__ float g_value(const float *p) { return *p; }
__device__ __forceinline__ unsigned short g_bits(const bf16_t *p) { return *reinterpret_cast<const unsigned short *>(p); }
__device__ __forceinline__ unsigned short g_bits(const f16_t *p) { return *reinterpret_cast<const unsigned short *>(p); }

constexpr int mfma_rows(int MT) { return MT * 32 - 16; }                 // pixel rows per k-chunk of a tile: pixels + 1 trash row <= this
constexpr int mfma_tile_bytes(int MT) { return 2 * mfma_rows(MT) * 16; }  // one A tile (hi or lo)
constexpr int mfma_phase_tiles(int MT) { return MT < 4 ? MT : 4; }      // tiles per reduction phase: NW x tiles x 4 KiB of LDS
constexpr int mfma_lds_bytes(int MT, int NW)
{
    return std::max(NW * 2 * mfma_tile_bytes(MT) + 256, NW * mfma_phase_tiles(MT) * 4096);
}

template <typename T, typename TL, typename GV, int MT, int NL, int NW>
__global__ void __launch_bounds__(NW * 64, 1)
msda_bwd_value_mfma_kernel(const Params p, int l0)
{
    using X = Mx<T>;
    using E = typename X::E;
    using V = typename X::V;
    constexpr int D = 32, NP = mfma_rows(MT), kTile = mfma_tile_bytes(MT), kPhase = mfma_phase_tiles(MT), kLds = mfma_lds_bytes(MT, NW);
    constexpr bool kSplitG = sizeof(T) == 4;
    constexpr int kRedRows = MSDA_MFMA_RED;
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *Ahi = lds + wave * (2 * kTile), *Alo = Ahi + kTile;

    // the levels of this launch, from the DEVICE shapes (the host's copy only chose the kernel)
    int H[NL], Wd[NL], poff[NL], npix = 0;
#pragma unroll
    for (int li = 0; li < NL; ++li) {
        H[li] = (int)p.shapes[2 * (l0 + li)]; Wd[li] = (int)p.shapes[2 * (l0 + li) + 1];
        poff[li] = npix; npix += H[li] * Wd[li];
    }
    const int lsi0 = (int)p.lsi[l0];
    const int MD = p.M * D;
    const int clips = p.groups / p.frames;
    const int n_items = clips * p.frames * p.M;
    // a stale host copy (include/msda.h: it MUST be a true copy for backward calls) that hid levels too large for the tiles:
    // no silently wrong sums -- these levels' pixels are filled with NaN
    const bool fits = npix + 1 <= NP;
    if (!fits) {
        for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
            const int m = item % p.M, gf = item / p.M;
            GV *gmap = static_cast<GV *>(p.grad_value) + ((long long)gf * p.S + lsi0) * MD + m * D;
            for (int i = tid; i < npix * D; i += NW * 64) Store<GV>::put(gmap + (long long)(i / D) * MD + (i % D), __builtin_nanf(""));
        }
        return;
    }

    const int g = lane >> 2, pt = lane & 3;               // builder role: point pt of group g of the step
    const int n = lane & 31, kh = lane >> 5;              // MFMA role: channel n / pixel n of a tile, k-chunk kh
    const unsigned wr_hi = lds_addr(Ahi) + (unsigned)((g >> 3) * NP) * 16u + (unsigned)(g & 7) * 2u;      // + pixel * 16
    const unsigned wr_trash = wr_hi + (unsigned)npix * 16u;       // row npix of the tile: written, never stored
    const unsigned char *rd_hi = Ahi + (kh * NP + n) * 16, *rd_lo = rd_hi + kTile;
    // lane parts of a point's index: current-frame points [.., M, LA, PA], temporal points [.., M, LB, PB]
    const unsigned lane_c = (unsigned)(g * p.M * p.LA * p.PA + pt), lane_t = (unsigned)(g * p.M * p.LB * p.PB + pt);
    const unsigned lane_g = (unsigned)(8 * kh * MD + n);  // lane part of a G element: row 8 kh (+ j), channel n
    const int nst = (p.Lq + 15) / 16;
    const int n_tw = p.frames * p.window, win = max(p.window, 1);
    float *red = reinterpret_cast<float *>(lds);

    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int m = item % p.M, gf = item / p.M, f = gf % p.frames, clip = gf / p.frames;
#if defined(MSDA_MFMA_TRACE)
        unsigned tr[16] = {};
        unsigned long long tr_t = __builtin_readcyclecounter();
#endif
        // the tiles are zero before any wave writes a cell (first item, and after every reduction, which borrows them)
        for (int i = tid * 16; i < kLds; i += NW * 64 * 16) *reinterpret_cast<u32x4 *>(lds + i) = u32x4{0u, 0u, 0u, 0u};
        // sources reading frame f: lane s of every wave holds source s (-1 = the frame's own current-frame points, else t * window
        // + w with frame_table[t, w] == f); a step reads its source with a readlane
        int my_src = -1, nsrc = 1;
        if (n_tw > 0) {
            const bool hit = lane < n_tw && p.ftab[lane] == f;
            u64 bal = __ballot(hit);
            nsrc = 1 + (int)__popcll(bal);
            int cnt = 0;
            while (bal) {                                   // lane s >= 1 takes the s-th set bit
                const int b = __builtin_ctzll(bal);
                bal &= bal - 1;
                ++cnt;
                if (cnt == lane) my_src = b;
            }
        }
        const int nsteps = nsrc * nst;
        __syncthreads();

        f32x16_t acc[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        // Every load of the loop is a BUFFER load: a resource per array over this clip (SGPRs, built once per item), the lane's
        // part of the address as a 32-bit byte offset that never changes (one VGPR per array), the step's part as a scalar offset --
        // no address arithmetic on the vector unit, no address registers.
        const long long clip_rows = (long long)clip * p.frames * p.Lq;                   // first query row of the clip
        const __amdgpu_buffer_rsrc_t r_go = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T *>(static_cast<const T *>(p.grad_out) + clip_rows * MD), 0, (int)((long long)p.frames * p.Lq * MD * (long long)sizeof(T)), 0x00020000);
        const long long eA = (long long)p.M * p.LA * p.PA, eB = (long long)p.M * p.LB * p.PB;     // point elements per query row
        const __amdgpu_buffer_rsrc_t r_locA = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<TL *>(static_cast<const TL *>(p.locA) + 2 * clip_rows * eA), 0, (int)((long long)p.frames * p.Lq * eA * 2 * (long long)sizeof(TL)), 0x00020000);
        const __amdgpu_buffer_rsrc_t r_awA = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<TL *>(static_cast<const TL *>(p.awA) + clip_rows * eA), 0, (int)((long long)p.frames * p.Lq * eA * (long long)sizeof(TL)), 0x00020000);
        const __amdgpu_buffer_rsrc_t r_locB = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<TL *>(static_cast<const TL *>(n_tw > 0 ? p.locB : p.locA) + (n_tw > 0 ? 2 * clip_rows * eB : 0)), 0,
            (int)(n_tw > 0 ? (long long)p.frames * p.Lq * eB * 2 * (long long)sizeof(TL) : 0), 0x00020000);
        const __amdgpu_buffer_rsrc_t r_awB = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<TL *>(static_cast<const TL *>(n_tw > 0 ? p.awB : p.awA) + (n_tw > 0 ? clip_rows * eB : 0)), 0,
            (int)(n_tw > 0 ? (long long)p.frames * p.Lq * eB * (long long)sizeof(TL) : 0), 0x00020000);
        // Scalar bookkeeping of a wave's steps, advanced incrementally (no division in the loop): source s, step j inside it; per
        // source the element offset of its points and the byte offset of its rows at query 0; per step the first query q0.  The
        // last step of a source starts at Lq - 16 and masks the groups the step before it has done: every step reads 16 valid
        // rows.  Steps beyond the item's last repeat it (their loads are issued and never used).
        struct Cursor { int s, j, temporal, points; unsigned pts0, go0, pts_row; };
        // (a source's constants are worked out once per item by the lane that holds it -- a division and five kernel arguments --
        // and a wave entering a source reads them with four readlanes: it does so every second or third of its steps)
        const int src_l = lane < nsrc ? my_src : -1;
        const int t_l = src_l < 0 ? f : src_l / win, w_l = src_l < 0 ? 0 : src_l - t_l * win;
        const int points_l = src_l < 0 ? p.PA : p.PB, levels_l = src_l < 0 ? p.LA : p.LB;
        const unsigned pts_row_l = (unsigned)(p.M * levels_l * points_l);                          // point elements per query row
        const unsigned pts0_l = (unsigned)(t_l * p.Lq) * pts_row_l + (unsigned)((m * levels_l + (src_l < 0 ? l0 : w_l * p.L + l0)) * points_l);
        const unsigned go0_l = ((unsigned)(t_l * p.Lq) * (unsigned)MD + (unsigned)(m * D)) * (unsigned)sizeof(T);
        auto enter_source = [&](Cursor &c) {
            const int sl = min(c.s, nsrc - 1);
            c.temporal = sl > 0 ? 1 : 0;                    // (lane 0 holds the frame's own current-frame points)
            c.points = __builtin_amdgcn_readlane(points_l, sl);
            c.pts_row = (unsigned)__builtin_amdgcn_readlane((int)pts_row_l, sl);
            c.pts0 = (unsigned)__builtin_amdgcn_readlane((int)pts0_l, sl);
            c.go0 = (unsigned)__builtin_amdgcn_readlane((int)go0_l, sl);
        };
        auto advance = [&](Cursor &c) {
            c.j += NW;
            if (c.j >= nst) {
                do { c.j -= nst; ++c.s; } while (c.j >= nst);
                if (c.s < nsrc) enter_source(c);
            }
        };
        // the loads of a step, issued one step AHEAD, all unconditional (s_waitcnt vmcnt(N) needs a count the compiler knows)
        struct Raw { float x[NL], y[NL], a[NL]; float gf32[kSplitG ? 8 : 1]; unsigned short g16[kSplitG ? 1 : 8]; };
        auto issue = [&](const Cursor &c, Raw &r) {
            const unsigned q0 = (unsigned)min(16 * c.j, p.Lq - 16);
            const unsigned pts_off = c.pts0 + q0 * c.pts_row, go_off = c.go0 + q0 * (unsigned)MD * (unsigned)sizeof(T);
            const unsigned lo = c.temporal ? lane_t : lane_c;
            const unsigned ptc = (unsigned)min(pt, c.points - 1) - (unsigned)pt;            // (a point beyond the source's count re-reads the last one)
            const __amdgpu_buffer_rsrc_t r_loc = c.temporal ? r_locB : r_locA, r_aw = c.temporal ? r_awB : r_awA;
#pragma unroll
            for (int li = 0; li < NL; ++li) {
                const unsigned e = lo + ptc + (unsigned)(li * c.points);
                if constexpr (sizeof(TL) == 4) {
                    const u32x2 xy = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r_loc, e * 8u, pts_off * 8u, 0));
                    r.x[li] = __uint_as_float(xy.x); r.y[li] = __uint_as_float(xy.y);
                    r.a[li] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_aw, e * 4u, pts_off * 4u, 0));
                } else {
                    const unsigned xy = __builtin_amdgcn_raw_buffer_load_b32(r_loc, e * 4u, pts_off * 4u, 0);
                    const unsigned short ab = __builtin_amdgcn_raw_buffer_load_b16(r_aw, e * 2u, pts_off * 2u, 0);
                    load_xy(reinterpret_cast<const TL *>(&xy), r.x[li], r.y[li]);
                    r.a[li] = Store<TL>::get(reinterpret_cast<const TL *>(&ab));
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned so = go_off + (unsigned)(j * MD) * (unsigned)sizeof(T);
                if constexpr (kSplitG) r.gf32[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_go, lane_g * 4u, so, 0));
                else r.g16[j] = __builtin_amdgcn_raw_buffer_load_b16(r_go, lane_g * 2u, so, 0);
            }
        };
        Cursor cur;
        cur.s = 0; cur.j = wave;
        while (cur.j >= nst) { cur.j -= nst; ++cur.s; }          // (fewer steps per source than waves)
        enter_source(cur);
        Raw raw;
        issue(cur, raw);
        MSDA_TR(0)
        for (int st = wave; st < nsteps; st += NW) {
#if defined(MSDA_MFMA_TRACE)
            asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
            MSDA_TR(1)
            ++tr[15];
#endif
            const bool act = g >= 16 * cur.j - min(16 * cur.j, p.Lq - 16) && pt < cur.points;
            // ---- this step's values out of the load registers: the B operand G[k = 8 kh + j][n], the points
            V bhi, blo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (kSplitG) {
                    const E hi = (E)raw.gf32[j];
                    bhi[j] = hi;
                    blo[j] = (E)(raw.gf32[j] - (float)hi);
                } else {
                    bhi[j] = __builtin_bit_cast(E, raw.g16[j]);
                }
            }
            float xs[NL], ys[NL], as[NL];
#pragma unroll
            for (int li = 0; li < NL; ++li) { xs[li] = raw.x[li]; ys[li] = raw.y[li]; as[li] = raw.a[li]; }
            advance(cur);
#if defined(MSDA_MFMA_EXP) && (MSDA_MFMA_EXP & 4)        // timing only: no loads inside the loop (every step reuses the first one's)
            asm volatile("" : "+v"(raw.x[0]), "+v"(raw.y[0]), "+v"(raw.a[0]));
#else
            issue(cur, raw);                                  // the next step's loads fly under this step's work
#endif
            MSDA_TR(2)
            unsigned cells[NL][4];
#pragma unroll
            for (int li = 0; li < NL; ++li) {
                const float Hf = (float)H[li], Wf = (float)Wd[li];
                float a = as[li];
                // ---- geometry: rounded product, then the subtraction (cuh:285-286); the range test of cuh:288
                float h_im = __fsub_rn(__fmul_rn(ys[li], Hf), 0.5f), w_im = __fsub_rn(__fmul_rn(xs[li], Wf), 0.5f);
                const bool inr = act && h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
                if (!inr) { h_im = -100.f; w_im = -100.f; a = 0.f; }
                const float r0f = floorf(h_im), c0f = floorf(w_im), r1f = r0f + 1.f, c1f = c0f + 1.f;
                const int hl = (int)r0f, wl = (int)c0f;
                // ---- merge: totals of all four points of the quad at my four corner pixels, in point order
                float t00 = 0.f, t01 = 0.f, t10 = 0.f, t11 = 0.f;
#define MSDA_MERGE(J) { \
                    const float th0 = tent(quad_bcast<J>(h_im) - r0f) * quad_bcast<J>(a), th1 = tent(quad_bcast<J>(h_im) - r1f) * quad_bcast<J>(a); \
                    const float tw0 = tent(quad_bcast<J>(w_im) - c0f), tw1 = tent(quad_bcast<J>(w_im) - c1f); \
                    t00 = fmaf(th0, tw0, t00); t01 = fmaf(th0, tw1, t01); t10 = fmaf(th1, tw0, t10); t11 = fmaf(th1, tw1, t11); }
#if defined(MSDA_MFMA_EXP) && (MSDA_MFMA_EXP & 16)       // timing only: one point of the quad instead of four
                MSDA_MERGE(0)
#else
                MSDA_MERGE(0) MSDA_MERGE(1) MSDA_MERGE(2) MSDA_MERGE(3)
#endif
#undef MSDA_MERGE
                // ---- cells; corners outside the map (cuh:56-78) and skipped points go to the trash row
                const bool rv0 = inr && (unsigned)hl < (unsigned)H[li], rv1 = inr && (unsigned)(hl + 1) < (unsigned)H[li];
                const bool cv0 = (unsigned)wl < (unsigned)Wd[li], cv1 = (unsigned)(wl + 1) < (unsigned)Wd[li];
                // (wr_hi folded into the two candidates first: one select per cell, no add behind it)
                const unsigned c00 = wr_hi + (unsigned)(poff[li] + hl * Wd[li] + wl) * 16u, wrow = (unsigned)Wd[li] * 16u;
                cells[li][0] = (rv0 && cv0) ? c00 : wr_trash; cells[li][1] = (rv0 && cv1) ? c00 + 16u : wr_trash;
                cells[li][2] = (rv1 && cv0) ? c00 + wrow : wr_trash; cells[li][3] = (rv1 && cv1) ? c00 + wrow + 16u : wr_trash;
                // hi = E(t) (round to nearest), lo = E(t - hi), two cells per packed conversion; the low half of a pair leaves with
                // ds_write_b16, the high half with ds_write_b16_d16_hi: no shifts or masks between the conversion and the stores
                const float tt[4] = {t00, t01, t10, t11};
#pragma unroll
                for (int c = 0; c < 4; c += 2) {
                    typedef __attribute__((ext_vector_type(2))) E E2;
                    const E2 hi = {(E)tt[c], (E)tt[c + 1]};
                    const E2 lo = {(E)(tt[c] - (float)hi[0]), (E)(tt[c + 1] - (float)hi[1])};
                    const unsigned hb = __builtin_bit_cast(unsigned, hi), lb = __builtin_bit_cast(unsigned, lo);
#if defined(MSDA_MFMA_EXP) && (MSDA_MFMA_EXP & 8)        // timing only: the cells are computed and not written
                    asm volatile("" : : "v"(cells[li][c]), "v"(cells[li][c + 1]), "v"(hb), "v"(lb));
#else
                    asm volatile("ds_write_b16 %0, %1" : : "v"(cells[li][c]), "v"(hb) : "memory");
                    asm volatile("ds_write_b16_d16_hi %0, %1" : : "v"(cells[li][c + 1]), "v"(hb) : "memory");
                    asm volatile("ds_write_b16 %0, %1 offset:%2" : : "v"(cells[li][c]), "v"(lb), "n"(kTile) : "memory");
                    asm volatile("ds_write_b16_d16_hi %0, %1 offset:%2" : : "v"(cells[li][c + 1]), "v"(lb), "n"(kTile) : "memory");
#endif
                }
            }
            // ---- the products, two pixel tiles at a time, the next pair's operands read under this pair's instructions (the LDS
            // operations of one wave complete in order: these reads see the cells written above)
            MSDA_TR(3)
            constexpr int TB = MSDA_MFMA_TB < MT ? MSDA_MFMA_TB : MT, NB = (MT + TB - 1) / TB;
            V fh[2][TB], fl[2][TB];
            auto read_pair = [&](int b, V (&h)[TB], V (&l)[TB]) {
#pragma unroll
                for (int u = 0; u < TB; ++u)
                    if (b * TB + u < MT) {
                        h[u] = *reinterpret_cast<const V *>(rd_hi + (b * TB + u) * 512);
                        l[u] = *reinterpret_cast<const V *>(rd_lo + (b * TB + u) * 512);
                    }
            };
            read_pair(0, fh[0], fl[0]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b + 1 < NB) read_pair(b + 1, fh[(b + 1) & 1], fl[(b + 1) & 1]);
#if defined(MSDA_MFMA_EXP) && (MSDA_MFMA_EXP & 1)        // timing only: one product instead of the 20-30 (the operands still read)
                if (b == 0) acc[0] = mma(fh[0][0], bhi, acc[0]);
#pragma unroll
                for (int u = 0; u < TB; ++u) if (b * TB + u < MT) { acc[b * TB + u][0] += (float)fh[b & 1][u][0] + (float)fl[b & 1][u][0] + (float)blo[0]; }
#else
#pragma unroll
                for (int u = 0; u < TB; ++u) if (b * TB + u < MT) acc[b * TB + u] = mma(fh[b & 1][u], bhi, acc[b * TB + u]);
#pragma unroll
                for (int u = 0; u < TB; ++u) if (b * TB + u < MT) acc[b * TB + u] = mma(fl[b & 1][u], bhi, acc[b * TB + u]);
                if constexpr (kSplitG) {
#pragma unroll
                    for (int u = 0; u < TB; ++u) if (b * TB + u < MT) acc[b * TB + u] = mma(fh[b & 1][u], blo, acc[b * TB + u]);
                }
#endif
#if MSDA_MFMA_SB
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            MSDA_TR(4)
            // ---- cells back to zero
            const unsigned zero = 0u;
#pragma unroll
            for (int li = 0; li < NL; ++li)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#if defined(MSDA_MFMA_EXP) && (MSDA_MFMA_EXP & 8)
                    asm volatile("" : : "v"(cells[li][c]), "v"(zero));
#else
                    asm volatile("ds_write_b16 %0, %1" : : "v"(cells[li][c]), "v"(zero) : "memory");
                    asm volatile("ds_write_b16 %0, %1 offset:%2" : : "v"(cells[li][c]), "v"(zero), "n"(kTile) : "memory");
#endif
                }
            MSDA_TR(5)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
#if defined(MSDA_MFMA_TRACE)
        __syncthreads();
        MSDA_TR(6)
#endif

        // ---- the waves' accumulators -> one, through the LDS, kPhase tiles at a time
        GV *gmap = static_cast<GV *>(p.grad_value) + ((long long)gf * p.S + lsi0) * MD + m * D;
#pragma unroll
        for (int ph = 0; ph * kPhase < MT; ++ph) {
            __syncthreads();
            MSDA_TR(7)
#pragma unroll
            for (int t = 0; t < kPhase; ++t)
                if (ph * kPhase + t < MT)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((wave * kPhase + t) * 16 + r) * 64 + lane] = acc[ph * kPhase + t][r];
            MSDA_TR(8)
            __syncthreads();
            MSDA_TR(9)
            // the phase's nt x 16 accumulator rows (64 lanes each) dealt over ALL the waves -- 2 nt consecutive rows of one tile per
            // wave -- and summed kRedRows rows at a time (8 LDS loads in flight per row), the partial sums added in wave order
            static_assert(NW == 8 && MT % 2 == 0 && kPhase % 2 == 0, "a wave's rows: a multiple of four inside one tile");
            const int nt = MT - ph * kPhase < kPhase ? MT - ph * kPhase : kPhase, per = 2 * nt;      // (constants once ph is unrolled)
            int row0 = wave * per;
            asm volatile("" : "+s"(row0));                      // (opaque, like the lane's part below: nothing of the stores' addresses hoisted)
            const int tw = row0 >> 4, r0 = row0 & 15, t = ph * kPhase + tw;
            const float *src = red + (tw * 16 + r0) * 64 + lane;
            int kh4 = 4 * kh, lane_off = 4 * kh * MD + n;
            asm volatile("" : "+v"(kh4), "+v"(lane_off));       // (opaque: or the item loop keeps every store's 64-bit offset in registers)
#pragma unroll
            for (int i = 0; i < per; i += kRedRows) {
                float part[NW][kRedRows];
#pragma unroll
                for (int w = 0; w < NW; ++w)
#pragma unroll
                    for (int k = 0; k < kRedRows; ++k) part[w][k] = src[(w * kPhase * 16 + i + k) * 64];
#pragma unroll
                for (int k = 0; k < kRedRows; ++k) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) v += part[w][k];
                    const int r = r0 + i + k, pixs = 32 * t + (r & 3) + 8 * (r >> 2);       // (uniform; the lane's pixel: + 4 kh)
                    if (pixs + kh4 < npix) Store<GV>::put(gmap + (long long)pixs * MD + lane_off, v);
                }
                __builtin_amdgcn_sched_barrier(0);          // (without it every batch's loads are hoisted to the top: 64 registers more)
            }
            MSDA_TR(10)
        }
        __syncthreads();                                    // the reduction buffer is the tiles: zeroed again at the top
#if defined(MSDA_MFMA_TRACE)
        MSDA_TR(11)
        if (lane < 16) {
            unsigned v = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) v = lane == k ? tr[k] : v;
            float *t0 = reinterpret_cast<float *>(p.grad_value) + ((long long)gf * p.S + wave) * MD + m * D + lane;
            *t0 = __uint_as_float(v);
        }
#endif
    }
}

template <typename T, typename TL, typename GV, int MT, int NL>
int scatter_mfma_launch(const Params &p, int l0, unsigned grid, hipStream_t stream)
{
    constexpr int NW = 8;
    static LdsGrant granted;
    const auto kern = &msda_bwd_value_mfma_kernel<T, TL, GV, MT, NL, NW>;
    constexpr size_t lds = mfma_lds_bytes(MT, NW);
    if (const int rc = grant_lds(reinterpret_cast<const void *>(kern), lds, granted, "the matrix-pipe scatter kernel")) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, stream, p, l0);
    return check_launch(std::is_same<GV, float>::value ? "msda backward (matrix-pipe scatter kernel, coarse levels)"
                                                       : "msda backward (matrix-pipe scatter kernel, coarse levels, grad_value in the storage type)");
}

template <typename T, typename TL, typename GV>
int scatter_mfma_sizes(const Params &p, int l0, int tiles, unsigned grid, hipStream_t stream)
{
    const bool two = p.L - l0 == 2;
    switch (tiles) {
        case 2: return two ? scatter_mfma_launch<T, TL, GV, 2, 2>(p, l0, grid, stream) : scatter_mfma_launch<T, TL, GV, 2, 1>(p, l0, grid, stream);
        case 4: return two ? scatter_mfma_launch<T, TL, GV, 4, 2>(p, l0, grid, stream) : scatter_mfma_launch<T, TL, GV, 4, 1>(p, l0, grid, stream);
        case 6: return two ? scatter_mfma_launch<T, TL, GV, 6, 2>(p, l0, grid, stream) : scatter_mfma_launch<T, TL, GV, 6, 1>(p, l0, grid, stream);
        case 8: return two ? scatter_mfma_launch<T, TL, GV, 8, 2>(p, l0, grid, stream) : scatter_mfma_launch<T, TL, GV, 8, 1>(p, l0, grid, stream);
        case 10: return two ? scatter_mfma_launch<T, TL, GV, 10, 2>(p, l0, grid, stream) : scatter_mfma_launch<T, TL, GV, 10, 1>(p, l0, grid, stream);
        default: return fail(MSDA_ERR_ARG, "msda: no matrix-pipe scatter kernel for this tile count%s");
    }
}

}  // namespace

// Pixel tiles (of 32) the matrix-pipe scatter needs for `pixels` coarse pixels, or 0 when they do not fit its largest kernel.
int mfma_scatter_tiles(long long pixels)
{
    for (int mt : {2, 4, 6, 8, 10})
        if (pixels + 1 <= mfma_rows(mt)) return mt;
    return 0;
}

int launch_scatter_mfma(int dtype, bool storage_typed, const Params &p, int l0, int tiles, hipStream_t stream)
{
    const long long items = (long long)p.groups * p.M;
    const unsigned grid = (unsigned)std::min<long long>(items, 1LL << 20);       // one item per workgroup (beyond 2^20: strided)
    return dispatch_types(dtype, [&](auto t, auto tl) {
        using T = typename decltype(t)::type;
        using TL = typename decltype(tl)::type;
        if constexpr (sizeof(T) == 2) {
            if (storage_typed) return scatter_mfma_sizes<T, TL, T>(p, l0, tiles, grid, stream);
        }
        return scatter_mfma_sizes<T, TL, float>(p, l0, tiles, grid, stream);
    });
}

}  // namespace msda
