// msda_win.hip -- "resident-window" kernels: forward and backward gather pass of ENCODER-shaped calls (D = 32), where query
// i is pixel i of the pyramid and samples round its own position in every frame (ms_deform_attn.py:435-460; the plain
// MSDeformAttn of the single-frame encoder, deformable_transformer.py:212-226).
//
// The resident-slab kernels (msda_rs.hip) keep WHOLE levels of a source frame in LDS; at 800x1333 only the last level (fp32)
// or the last two (16-bit) fit and 75 % / 50 % of the taps are scattered gathers through L1 / L2.  Here a workgroup owns the
// queries of one spatial TILE -- B x B pixels of level 0 and the pixels of the other levels whose centres fall into it, for
// every query frame -- and stages per source frame a WINDOW of each level: the tile's footprint on that level plus a halo
// (include/..: WinPlan, win_axis in msda_common.h).  With local sampling every tap is an LDS read; a corner outside its
// window is read from memory in a second, normally skipped, pass over the group (buffer loads; corners that were served from
// LDS get an out-of-range offset there, corners served from memory the zero row here), so ANY input is computed exactly --
// only the speed depends on locality.  Rows, quads, records, accumulators: as in msda_rs.hip (one quad = one row, lane c owns
// point c of the current group, NT accumulator sets per wave carried across the source frames).
#include "msda_common.h"
#include <cstdio>

namespace msda {
namespace {

#include "msda_rs_common.inc"

struct WinShared {
    int *H, *W, *lsi, *wb, *wy0, *wx0, *wh, *ww, *qy0, *qx0, *qw, *qb, *qcum;   // level tables (LDS); qcum has L + 1 entries
    unsigned *mask;             // [frames, frames] slot masks, as in msda_rs.hip
    int zero_off;               // byte offset of the zero row
    int nq;                     // queries of this tile (per frame)
};

// LDS carve + the tile's tables.  Thread l computes level l; thread 0 the running query counts.
__device__ __forceinline__ WinShared win_setup(const Params &p, const WinPlan &wp, unsigned char *lds_raw, int slab_bytes,
                                               int ty, int tx)
{
    WinShared sh;
    sh.zero_off = slab_bytes;
    sh.mask = reinterpret_cast<unsigned *>(lds_raw + slab_bytes + 128);
    int *tab = reinterpret_cast<int *>(sh.mask + kRsMaxFrames * kRsMaxFrames);
    sh.H = tab; sh.W = tab + kWinMaxLevels; sh.lsi = tab + 2 * kWinMaxLevels; sh.wb = tab + 3 * kWinMaxLevels;
    sh.wy0 = tab + 4 * kWinMaxLevels; sh.wx0 = tab + 5 * kWinMaxLevels; sh.wh = tab + 6 * kWinMaxLevels;
    sh.ww = tab + 7 * kWinMaxLevels; sh.qy0 = tab + 8 * kWinMaxLevels; sh.qx0 = tab + 9 * kWinMaxLevels;
    sh.qw = tab + 10 * kWinMaxLevels; sh.qb = tab + 11 * kWinMaxLevels; sh.qcum = tab + 12 * kWinMaxLevels;    // (L + 1 <= 9 entries: 2 slots)
    const int tid = threadIdx.x, L = p.L;
    for (int i = tid; i < p.frames * p.frames; i += kRsThreads) {
        const int t = i / p.frames, f = i - t * p.frames;
        unsigned mk = (t == f) ? 1u : 0u;
        for (int w = 0; w < p.window; ++w) mk |= (p.ftab[t * p.window + w] == f) ? (2u << w) : 0u;
        sh.mask[i] = mk;
    }
    if (tid < 32) reinterpret_cast<float *>(lds_raw + sh.zero_off)[tid] = 0.f;
    int qn = 0;
    if (tid < L) {
        const int l = tid, H0 = (int)p.shapes[0], W0 = (int)p.shapes[1];
        const int H = (int)p.shapes[2 * l], W = (int)p.shapes[2 * l + 1];
        const int halo = wp.halo[(wp.split > 0 && l >= wp.split) ? 1 : 0];
        int qy0, qy1, wy0, wy1, qx0, qx1, wx0, wx1;
        win_axis(H, H0, ty, wp.By, halo, qy0, qy1, wy0, wy1);
        win_axis(W, W0, tx, wp.Bx, halo, qx0, qx1, wx0, wx1);
        sh.H[l] = H; sh.W[l] = W; sh.lsi[l] = (int)p.lsi[l]; sh.wb[l] = wp.wbase[l];
        sh.wy0[l] = wy0; sh.wx0[l] = wx0; sh.wh[l] = wy1 - wy0; sh.ww[l] = wx1 - wx0;
        sh.qy0[l] = qy0; sh.qx0[l] = qx0; sh.qw[l] = qx1 - qx0;
        qn = (qy1 - qy0) * (qx1 - qx0);
        sh.qcum[l + 1] = qn;                    // (count for now; made cumulative below)
    }
    __syncthreads();
    if (tid == 0) {
        // qb[l]: the query index of pixel (0, 0) of level l -- queries are the pixels of the pyramid in level order (the host
        // checked that Lq is their number); `lsi`, which addresses `value`, is not assumed to follow that order
        int acc = 0, pix = 0;
        sh.qcum[0] = 0;
        for (int l = 0; l < L; ++l) {
            acc += sh.qcum[l + 1]; sh.qcum[l + 1] = acc;
            sh.qb[l] = pix; pix += sh.H[l] * sh.W[l];
        }
    }
    __syncthreads();
    sh.nq = sh.qcum[L];
    return sh;
}

// windows of levels [la, lb) of source frame f (head m) -> LDS, 16 bytes per lane by LDS-DMA.  A wave takes whole window rows
// (no per-lane division); an instruction moves up to PXW pixels of one row, the lanes past the row's end are switched off.
template <typename T>
__device__ __forceinline__ void win_stage(const Params &p, const WinShared &sh, T *slab, int clip, int m, int f, int la, int lb,
                                          int wave, int lane)
{
    constexpr int GL = rs_row_bytes<T>() / 16, D = 32;
    constexpr int PXW = kWave / GL;                 // pixels per LDS-DMA wave instruction
    const int lpx = lane / GL;
    const T *src = static_cast<const T *>(p.value) + clip * p.v_clip + m * p.v_head + ((int64_t)f * p.S) * p.v_pix +
                   (lane % GL) * (16 / (int)sizeof(T));
    for (int l = la; l < lb; ++l) {
        const int ww = __builtin_amdgcn_readfirstlane(sh.ww[l]), wh = __builtin_amdgcn_readfirstlane(sh.wh[l]);
        const int W = __builtin_amdgcn_readfirstlane(sh.W[l]);
        const int first = __builtin_amdgcn_readfirstlane(sh.lsi[l] + sh.wy0[l] * W + sh.wx0[l]);
        T *dst = slab + (size_t)__builtin_amdgcn_readfirstlane(sh.wb[l]) * D;
        for (int wy = wave; wy < wh; wy += kRsWaves) {
            const T *rowp = src + (int64_t)(first + wy * W + lpx) * p.v_pix;
            for (int c = 0; c < ww; c += PXW) {
                if (c + lpx < ww) {
#if defined(__HIP_DEVICE_COMPILE__)
                    __builtin_amdgcn_global_load_lds(rowp + (int64_t)c * p.v_pix,
                                                     (__attribute__((address_space(3))) void *)(dst + (size_t)(wy * ww + c) * D), 16, 0, 0);
#endif
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// One level as the geometry needs it: the map, where its pixels start in a frame of `value`, and the tile's window on it.
struct WinLevel { int H, W, lsi, wb, wy0, wx0, wh, ww; };

__device__ __forceinline__ WinLevel win_level(const WinShared &sh, int l)
{
    return WinLevel{sh.H[l], sh.W[l], sh.lsi[l], sh.wb[l], sh.wy0[l], sh.wx0[l], sh.wh[l], sh.ww[l]};
}

// This lane's point: per corner the LDS byte address (zero row for a corner outside the map OR outside the window); `far` =
// the corners inside the map but outside the window, left to the second pass, which reads them from memory at pixel p00 (+1,
// +W, +W+1) of the clip.
struct WinGeom {
    float lh, lw, a;        // fractions and attention weight (all 0 for a point outside the range)
    int H, W, yl, bits;     // level shape, top tap row, validity bits of corners 0..3 (0: point skipped)
    int adr[4];             // LDS byte address of each corner WITHOUT the lane's slice offset
    int far, p00;
};

template <int ROWSH>
__device__ __forceinline__ WinGeom win_geometry(float x, float y, float a, const WinLevel &lv, bool in_phase, int fS, int zero_off)
{
    WinGeom g;
    g.H = lv.H; g.W = lv.W;
    const int H = g.H, W = g.W;
    const float h_im = __fsub_rn(__fmul_rn(y, (float)H), 0.5f);
    const float w_im = __fsub_rn(__fmul_rn(x, (float)W), 0.5f);
    const bool rng = in_phase && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;     // false for NaN
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int yl = rng ? (int)hf : 0, xl = rng ? (int)wf : 0;
    g.yl = yl;
    g.lh = rng ? h_im - hf : 0.f; g.lw = rng ? w_im - wf : 0.f; g.a = rng ? a : 0.f;
    const bool vy0 = rng && yl >= 0, vy1 = rng && yl + 1 <= H - 1, vx0 = xl >= 0, vx1 = xl + 1 <= W - 1;
    const bool ok[4] = {vy0 && vx0, vy0 && vx1, vy1 && vx0, vy1 && vx1};
    g.bits = (ok[0] ? 1 : 0) | (ok[1] ? 2 : 0) | (ok[2] ? 4 : 0) | (ok[3] ? 8 : 0);
    const int ry = yl - lv.wy0, rx = xl - lv.wx0;
    const bool iy0 = (unsigned)ry < (unsigned)lv.wh, iy1 = (unsigned)(ry + 1) < (unsigned)lv.wh;
    const bool ix0 = (unsigned)rx < (unsigned)lv.ww, ix1 = (unsigned)(rx + 1) < (unsigned)lv.ww;
    const bool in[4] = {iy0 && ix0, iy0 && ix1, iy1 && ix0, iy1 && ix1};
    const int l00 = lv.wb + ry * lv.ww + rx;
    const int lpix[4] = {l00, l00 + 1, l00 + lv.ww, l00 + lv.ww + 1};
    g.p00 = fS + lv.lsi + yl * W + xl;
    g.far = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        g.adr[s] = (ok[s] && in[s]) ? (lpix[s] << ROWSH) : zero_off;
        g.far |= (ok[s] && !in[s]) ? (1 << s) : 0;
    }
    return g;
}

// query i of the tile -> query index of the call (pixel of the pyramid); i < nq
__device__ __forceinline__ int win_query(const WinShared &sh, int L, int i)
{
    int l = 0;
    for (int u = 1; u < L; ++u) l += (i >= sh.qcum[u]) ? 1 : 0;
    const int r = i - sh.qcum[l], qw = max(sh.qw[l], 1);
    const int yy = r / qw, xx = r - yy * qw;
    return sh.qb[l] + (sh.qy0[l] + yy) * sh.W[l] + sh.qx0[l] + xx;
}

// ---- forward ------------------------------------------------------------------------------------------------------
template <typename T, typename TL, int NT>
__global__ void __launch_bounds__(kRsThreads)
msda_fwd_win_kernel(const Params p, const WinPlan wp, int slab_bytes)
{
    constexpr int RPW = kRsRows, D = 32, ROWB = rs_row_bytes<T>(), ROWSH = ROWB == 128 ? 7 : 6;
    constexpr bool kHalf = sizeof(T) == 2;
    extern __shared__ __attribute__((aligned(128))) unsigned char lds_raw[];       // (no static LDS: the windows start at 0)
    const int tid = threadIdx.x, lane = tid % kWave;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int L = p.L;
    T *slab = reinterpret_cast<T *>(lds_raw);

    // workgroup -> (clip, head, tile); an XCD takes a contiguous run of them (neighbouring tiles share window pixels in its L2)
    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u;
    const unsigned lin = xcd * (nwg / 8u) + min(xcd, nwg % 8u) + blockIdx.x / 8u;
    const unsigned ntile = (unsigned)(wp.tiles_y * wp.tiles_x);
    const int tile = (int)(lin % ntile), m = (int)((lin / ntile) % (unsigned)p.M), clip = (int)(lin / (ntile * (unsigned)p.M));
    const int ty = tile / wp.tiles_x, tx = tile - ty * wp.tiles_x;
    const WinShared sh = win_setup(p, wp, lds_raw, slab_bytes, ty, tx);
    const int nq = sh.nq, tpg = wp.tpg, ntiles = p.frames * tpg;
    const int my_tiles = wave < ntiles ? (ntiles - wave + kRsWaves - 1) / kRsWaves : 0;       // <= NT (host)

    const int j = lane / 4, cor = lane & 3, hsw = j & 1;
    const int off1 = kHalf ? cor * 16 : cor * 16 + hsw * 64, delta2 = hsw ? -64 : 64;
    const int pixB = p.v_pix * (int)sizeof(T);
    const char *vbase = reinterpret_cast<const char *>(static_cast<const T *>(p.value) + clip * p.v_clip + m * p.v_head);
    const unsigned vbytes = (unsigned)(((int64_t)p.frames * p.S - 1) * pixB + ROWB);
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(vbase), 0, (int)vbytes, 0x00020000);
#endif

    // wave tile k of this wave = (query frame tk, 16 queries of the tile); this quad's query of it
    int qk[NT];
    float acc[NT][8];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const int ct = wave + k * kRsWaves, t = ct / tpg, i = (ct - t * tpg) * RPW + j;
        qk[k] = (ct < ntiles && i < nq) ? win_query(sh, L, i) : -1;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[k][c] = 0.f;
    }
    const int nph = wp.split > 0 ? 2 : 1;

    for (int f = 0; f < p.frames; ++f) {
        const int fS = f * p.S;
        for (int ph = 0; ph < nph; ++ph) {
            const int la = ph == 0 ? 0 : wp.split, lb = (ph == 0 && nph == 2) ? wp.split : L;     // levels of this phase
            __syncthreads();                                   // every wave is done with the previous windows
#if !defined(MSDA_WIN_EXP) || MSDA_WIN_EXP != 1       // (timing experiments, wrong results: 1 = no staging, 2 = no corner work)
            win_stage<T>(p, sh, slab, clip, m, f, la, lb, wave, lane);
#endif
            __syncthreads();
#pragma unroll 1
            for (int k = 0; k < my_tiles; ++k) {
                const int ct = wave + k * kRsWaves, t = ct / tpg;
                unsigned todo = __builtin_amdgcn_readfirstlane(sh.mask[t * p.frames + f]);
                if (!todo || (ct - t * tpg) * RPW >= nq) continue;         // (a wave tile past the tile's last query)
                float wacc[8];                                     // working accumulators = set k
                int q = -1;
                static_for<NT>([&](auto Kc) {
                    constexpr int K = decltype(Kc)::value;
                    if (k == K) {
                        q = qk[K];
#pragma unroll
                        for (int c = 0; c < 8; ++c) wacc[c] = acc[K][c];
                    }
                });
                const bool live = q >= 0;
                const int64_t row = (((int64_t)clip * p.frames + t) * p.Lq + max(q, 0)) * p.M + m;
                auto corner = [&](auto Sc, int A, float Wt) {
                    constexpr bool SLAB = decltype(Sc)::value;
#if defined(MSDA_WIN_EXP) && MSDA_WIN_EXP == 2
                    wacc[0] += Wt * (float)A; return;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
                    const RsRaw<T> raw = rs_issue_row<T, SLAB>(rsrc, A, delta2);
                    rs_fma_row<T>(raw, Wt, wacc);
#endif
                    asm volatile("" ::: "memory");
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
                    const bool wide = p.wide_loads && P == 4 && npts == 16;           // (uniform) see load_slot_points
                    float xs[4], ys[4], as[4];
#if defined(MSDA_WIN_EXP) && MSDA_WIN_EXP == 3           // (timing: no point loads)
                    if (wide) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) { xs[i] = 0.3f + 0.001f * (float)(lane + i); ys[i] = 0.4f + 0.001f * (float)(j + i); as[i] = 0.1f; }
                    }
#else
                    if (wide) load_slot_points<TL>(loc, aw, idx0, cor, live, xs, ys, as);
#endif
                    // one group: this lane's point (x, y, a) on level lv; step R serves point R of the 16 rows
                    auto group = [&](int g0, float x, float y, float a, const WinLevel &lv, bool in_phase) __attribute__((always_inline)) {
                        const WinGeom g = win_geometry<ROWSH>(x, y, a, lv, in_phase, fS, sh.zero_off);
                        const float hh = 1.f - g.lh, hw = 1.f - g.lw;
                        float w4[4] = {g.a * (hh * hw), g.a * (hh * g.lw), g.a * (g.lh * hw), g.a * (g.lh * g.lw)};
                        int a4[4] = {g.adr[0], g.adr[1], g.adr[2], g.adr[3]};
                        static_for<4>([&](auto Rc) {
                            constexpr int R = decltype(Rc)::value;
                            // (ties step R's broadcasts to this point of the instruction stream: after the previous step's FMAs --
                            // otherwise the 32 broadcast values of a group are all formed up front and spill)
                            asm volatile("" : "+v"(a4[0]), "+v"(a4[1]), "+v"(a4[2]), "+v"(a4[3]), "+v"(w4[0]), "+v"(w4[1]), "+v"(w4[2]), "+v"(w4[3]) : "v"(wacc[0]));
#pragma unroll
                            for (int s = 0; s < 4; ++s) corner(std::true_type{}, quad_bcast<R>(a4[s]) + off1, quad_bcast<R>(w4[s]));
                        });
                        // second pass (normally skipped): corners outside their window, from memory
                        if (__builtin_amdgcn_ballot_w64(g.far != 0)) {
                            static_for<4>([&](auto Rc) {
                                constexpr int R = decltype(Rc)::value;
                                const int far = quad_bcast<R>(g.far);
                                if (!__builtin_amdgcn_ballot_w64(far != 0)) return;
                                const int p00 = quad_bcast<R>(g.p00), Wm = quad_bcast<R>(g.W);
                                const int mpix[4] = {p00, p00 + 1, p00 + Wm, p00 + Wm + 1};
#pragma unroll
                                for (int s = 0; s < 4; ++s)
                                    corner(std::false_type{}, ((far >> s) & 1) ? (int)((unsigned)mpix[s] * (unsigned)pixB) + off1 : (int)0x80000000u,
                                           quad_bcast<R>(w4[s]));
                            });
                        }
                    };
#pragma unroll 1
                    for (int g0 = 0; g0 < npts; g0 += 4) {
                        // (uniform) the group's levels: skip it unless one of them is staged in this phase
                        const int gl0 = (int)(((unsigned)g0 * invP) >> 16), gl1 = min((int)(((unsigned)(g0 + 3) * invP) >> 16), L - 1);
                        if (gl1 < la || gl0 >= lb) continue;
                        const int kk = g0 + cor;
                        float x = -10.f, y = -10.f, a = 0.f;       // far outside every map
                        if (wide) {
                            x = get4(xs, g0 >> 2); y = get4(ys, g0 >> 2); a = get4(as, g0 >> 2);
                        } else if (live && kk < npts) {
                            load_xy(loc + 2 * (idx0 + kk), x, y);
                            a = Store<TL>::get(aw + idx0 + kk);
                        }
                        // (the level's tables come from LDS: keeping them in registers -- scalar selects, or one level per lane read
                        // with v_readlane -- cost ~24 branches per group or the registers the fourth accumulator set needs)
                        const int lvl = min((int)(((unsigned)kk * invP) >> 16), L - 1);
                        const WinLevel lv = win_level(sh, lvl);
                        const bool in_phase = lvl >= la && lvl < lb;
#if defined(MSDA_WIN_EXP) && MSDA_WIN_EXP == 4           // (timing: points loaded, nothing done with them)
                        wacc[0] += x + y + a + (float)lv.H;
#else
                        group(g0, x, y, a, lv, in_phase);
#endif
                    }
                }
                static_for<NT>([&](auto Kc) {
                    constexpr int K = decltype(Kc)::value;
                    if (k == K) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) acc[K][c] = wacc[c];
                    }
                });
            }
        }
    }
    static_for<NT>([&](auto Kc) {
        constexpr int K = decltype(Kc)::value;
        const int ct = wave + K * kRsWaves, t = ct / tpg;
        if (qk[K] >= 0) {
            const int64_t row = (((int64_t)clip * p.frames + t) * p.Lq + qk[K]) * p.M + m;
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
}

// "... (resident-window kernel, tiles 8x12, halo 12/7, 3 tiles per wave)" for msda_last_route()
const char *win_label(const char *head, const WinPlan &w)
{
    static thread_local char buf[192];
    if (w.split)
        snprintf(buf, sizeof buf, "%s, tiles %dx%d, halo %d (levels < %d) / %d, %d tiles per wave)", head, w.By, w.Bx, w.halo[0], w.split, w.halo[1], w.nt);
    else
        snprintf(buf, sizeof buf, "%s, tiles %dx%d, halo %d, %d tiles per wave)", head, w.By, w.Bx, w.halo[0], w.nt);
    return buf;
}

template <typename T, typename TL, int NT>
int fwd_win(const Params &p, const WinPlan &w, unsigned grid, hipStream_t stream)
{
    static LdsGrant granted;
    const size_t total = (size_t)kWinSlabBytes + kWinTailBytes;
    const auto kern = &msda_fwd_win_kernel<T, TL, NT>;
    if (const int rc = grant_lds(reinterpret_cast<const void *>(kern), total, granted, "the resident-window forward kernel")) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kRsThreads), total, stream, p, w, kWinSlabBytes);
    return check_launch(win_label("msda forward (resident-window kernel", w));
}

// ---- backward gather pass (grad_loc / grad_attn) ---------------------------------------------------------------------------
// Same tiles, windows and second pass as the forward; per point the four dots <grad_out row, corner> as in msda_bwd_rs_kernel
// (cuh:123-158), every (row, slot) writes its own gradients, so a wave takes any number of wave tiles.  With one staging
// phase a slot's results leave as whole rows (three 16-byte stores per lane); with two phases each group is stored when it
// is done (level 0 and the other levels of a slot are finished in different phases).  Also leaves the per-point culling
// records the scatter pass reads.
template <typename T, typename TL>
__global__ void __launch_bounds__(kRsThreads)
msda_bwd_win_kernel(const Params p, const WinPlan wp, int slab_bytes)
{
    constexpr int RPW = kRsRows, D = 32, ROWB = rs_row_bytes<T>(), ROWSH = ROWB == 128 ? 7 : 6;
    constexpr bool kHalf = sizeof(T) == 2;
    extern __shared__ __attribute__((aligned(128))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid % kWave;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int L = p.L, VL = p.LA + p.LB;
    // the scatter pass that follows draws its work tickets from the head of the workspace (see msda_bwd_rs_kernel)
    if (blockIdx.x == 0 && tid < MSDA_BWD_WORKSPACE_BYTES / 4 && p.workspace) p.workspace[tid] = 0u;
    T *slab = reinterpret_cast<T *>(lds_raw);

    const unsigned nwg = gridDim.x, xcd = blockIdx.x % 8u;
    const unsigned lin = xcd * (nwg / 8u) + min(xcd, nwg % 8u) + blockIdx.x / 8u;
    const unsigned ntile = (unsigned)(wp.tiles_y * wp.tiles_x);
    const int tile = (int)(lin % ntile), m = (int)((lin / ntile) % (unsigned)p.M), clip = (int)(lin / (ntile * (unsigned)p.M));
    const int ty = tile / wp.tiles_x, tx = tile - ty * wp.tiles_x;
    const WinShared sh = win_setup(p, wp, lds_raw, slab_bytes, ty, tx);
    const int nq = sh.nq, tpg = wp.tpg, ntiles = p.frames * tpg;

    const int j = lane / 4, cor = lane & 3, hsw = j & 1;
    const int off1 = kHalf ? cor * 16 : cor * 16 + hsw * 64, delta2 = hsw ? -64 : 64;
    const int pixB = p.v_pix * (int)sizeof(T);
    const char *vbase = reinterpret_cast<const char *>(static_cast<const T *>(p.value) + clip * p.v_clip + m * p.v_head);
    const unsigned vbytes = (unsigned)(((int64_t)p.frames * p.S - 1) * pixB + ROWB);
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(vbase), 0, (int)vbytes, 0x00020000);
#endif
    const bool records = p.bbox != nullptr;        // per-point culling records (host: only with cull_points)
    const int nph = wp.split > 0 ? 2 : 1;

    for (int f = 0; f < p.frames; ++f) {
        const int fS = f * p.S;
        for (int ph = 0; ph < nph; ++ph) {
            const int la = ph == 0 ? 0 : wp.split, lb = (ph == 0 && nph == 2) ? wp.split : L;     // levels of this phase
            __syncthreads();                                   // every wave is done with the previous windows
            win_stage<T>(p, sh, slab, clip, m, f, la, lb, wave, lane);
            __syncthreads();
#pragma unroll 1
            for (int ct = wave; ct < ntiles; ct += kRsWaves) {
                const int t = ct / tpg, i0 = (ct - t * tpg) * RPW;
                unsigned todo = __builtin_amdgcn_readfirstlane(sh.mask[t * p.frames + f]);
                if (!todo || i0 >= nq) continue;
                const bool live = i0 + j < nq;
                const int q = live ? win_query(sh, L, i0 + j) : 0;
                const int64_t group = (int64_t)clip * p.frames + t;
                const int64_t row = ((group * p.Lq) + q) * p.M + m;
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
                    const bool wide_ld = p.wide_loads && P == 4 && npts == 16;
                    const bool wide = p.wide_stores && P == 4 && npts == 16 && nph == 1;      // whole-row stores (one phase only)
                    float wx[4] = {0.f, 0.f, 0.f, 0.f}, wy[4] = {0.f, 0.f, 0.f, 0.f}, wa[4] = {0.f, 0.f, 0.f, 0.f};
                    int wr[4] = {0, 0, 0, 0};
                    float xs[4], ys[4], as[4];
                    if (wide_ld) load_slot_points<TL>(loc, aw, idx0, cor, live, xs, ys, as);
#pragma unroll 1
                    for (int g0 = 0; g0 < npts; g0 += 4) {
                        const int gl0 = (int)(((unsigned)g0 * invP) >> 16), gl1 = min((int)(((unsigned)(g0 + 3) * invP) >> 16), L - 1);
                        if (gl1 < la || gl0 >= lb) continue;               // (uniform) none of the group's levels is staged in this phase
                        const int kk = g0 + cor;
                        const int lvl = min((int)(((unsigned)kk * invP) >> 16), L - 1);
                        const bool in_phase = lvl >= la && lvl < lb;
                        const bool mine = live && kk < npts && in_phase;
                        float x = -10.f, y = -10.f, a = 0.f;       // far outside every map
                        if (wide_ld) {
                            x = get4(xs, g0 >> 2); y = get4(ys, g0 >> 2); a = get4(as, g0 >> 2);
                        } else if (live && kk < npts) {
                            load_xy(loc + 2 * (idx0 + kk), x, y);
                            a = Store<TL>::get(aw + idx0 + kk);
                        }
                        const WinLevel lv = win_level(sh, lvl);
                        const WinGeom pt = win_geometry<ROWSH>(x, y, a, lv, in_phase, fS, sh.zero_off);
                        const int rowrec = pt.bits ? min(pt.yl, 32767) : kNoRow16;
                        if (wide) set4(wr, g0 >> 2, rowrec);
                        if (records && mine && !wide) {      // the point's top tap row, for the scatter's band test
                            const int pin = kk - lvl * P;
                            short *rec = reinterpret_cast<short *>(p.bbox + (((group * p.M + m) * VL + vl0 + lvl) * p.Lq + q) * 2);
                            rec[pin] = (short)rowrec;
                            if (pin == 0)
                                for (int u = P; u < 4; ++u) rec[u] = (short)kNoRow16;
                        }
                        float k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f;      // the dots of THIS lane's point
                        static_for<4>([&](auto Rc) {
                            constexpr int R = decltype(Rc)::value;
                            float d[4];
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
#if defined(__HIP_DEVICE_COMPILE__)
                                const RsRaw<T> raw = rs_issue_row<T, true>(rsrc, quad_bcast<R>(pt.adr[s]) + off1, delta2);
                                d[s] = rs_dot_row<T>(raw, g);
#endif
                                asm volatile("" ::: "memory");
                            }
                            quad_sum4(d);
                            const bool me = cor == R;
                            k0 = me ? d[0] : k0; k1 = me ? d[1] : k1; k2 = me ? d[2] : k2; k3 = me ? d[3] : k3;
                        });
                        // second pass (normally skipped): corners outside their window, from memory
                        if (__builtin_amdgcn_ballot_w64(pt.far != 0)) {
                            static_for<4>([&](auto Rc) {
                                constexpr int R = decltype(Rc)::value;
                                    const int far = quad_bcast<R>(pt.far);
                                if (!__builtin_amdgcn_ballot_w64(far != 0)) return;
                                const int p00 = quad_bcast<R>(pt.p00), Wm = quad_bcast<R>(pt.W);
                                const int mpix[4] = {p00, p00 + 1, p00 + Wm, p00 + Wm + 1};
                                float d[4];
#pragma unroll
                                for (int s = 0; s < 4; ++s) {
#if defined(__HIP_DEVICE_COMPILE__)
                                    const int A = ((far >> s) & 1) ? (int)((unsigned)mpix[s] * (unsigned)pixB) + off1 : (int)0x80000000u;
                                    const RsRaw<T> raw = rs_issue_row<T, false>(rsrc, A, delta2);
                                    d[s] = rs_dot_row<T>(raw, g);
#endif
                                }
                                quad_sum4(d);
                                const bool me = cor == R;
                                k0 += me ? d[0] : 0.f; k1 += me ? d[1] : 0.f; k2 += me ? d[2] : 0.f; k3 += me ? d[3] : 0.f;
                            });
                        }
                        {   // every lane finishes its own point (cuh:123-158 on the reduced dots; dots of corners outside the map are 0)
                            const float lh = pt.lh, lw = pt.lw, hh = 1.f - lh, hw = 1.f - lw;
                            const float g_aw = (hh * hw) * k0 + (hh * lw) * k1 + (lh * hw) * k2 + (lh * lw) * k3;
                            const float g_w = hh * (k1 - k0) + lh * (k3 - k2);
                            const float g_h = hw * (k2 - k0) + lw * (k3 - k1);
                            const float gx = (float)pt.W * g_w * pt.a, gy = (float)pt.H * g_h * pt.a;
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
                                SlabStore<TL>::store(gaw + idx0 + 4 * cor, wa);
                            } else {
                                typedef float f32x4 __attribute__((ext_vector_type(4)));
                                __builtin_nontemporal_store((f32x4){xy[0], xy[1], xy[2], xy[3]}, reinterpret_cast<f32x4 *>(gl));
                                __builtin_nontemporal_store((f32x4){xy[4], xy[5], xy[6], xy[7]}, reinterpret_cast<f32x4 *>(gl + 4));
                                __builtin_nontemporal_store((f32x4){wa[0], wa[1], wa[2], wa[3]}, reinterpret_cast<f32x4 *>(gaw + idx0 + 4 * cor));
                            }
                            if (records)
                                *reinterpret_cast<int2 *>(p.bbox + (((group * p.M + m) * VL + vl0 + cor) * p.Lq + q) * 2) =
                                    make_int2((wr[0] & 0xffff) | (wr[1] << 16), (wr[2] & 0xffff) | (wr[3] << 16));
                        }
                    }
                }
            }
        }
    }
}

template <typename T, typename TL>
int bwd_win(const Params &p, const WinPlan &w, unsigned grid, hipStream_t stream)
{
    static LdsGrant granted;
    const size_t total = (size_t)kWinSlabBytes + kWinTailBytes;
    const auto kern = &msda_bwd_win_kernel<T, TL>;
    if (const int rc = grant_lds(reinterpret_cast<const void *>(kern), total, granted, "the resident-window gather-pass kernel")) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kRsThreads), total, stream, p, w, kWinSlabBytes);
    return check_launch(win_label("msda backward (resident-window kernel, grad_loc/grad_attn", w));
}

}  // namespace

int launch_fwd_win(int dtype, const Params &p, const WinPlan &w, hipStream_t stream)
{
    const int64_t clips = p.groups / p.frames;
    const unsigned grid = (unsigned)(clips * p.M * w.tiles_y * w.tiles_x);
    return dispatch_types(dtype, [&](auto t, auto tl) {
        typedef typename decltype(t)::type T;
        typedef typename decltype(tl)::type TL;
        // (plans have at most 3 wave tiles per wave: a fourth accumulator set does not fit the 128 registers without spills)
        return w.nt <= 2 ? fwd_win<T, TL, 2>(p, w, grid, stream) : fwd_win<T, TL, 3>(p, w, grid, stream);
    });
}

int launch_bwd_win(int dtype, const Params &p, const WinPlan &w, hipStream_t stream)
{
    const int64_t clips = p.groups / p.frames;
    const unsigned grid = (unsigned)(clips * p.M * w.tiles_y * w.tiles_x);
    return dispatch_types(dtype, [&](auto t, auto tl) {
        return bwd_win<typename decltype(t)::type, typename decltype(tl)::type>(p, w, grid, stream);
    });
}

}  // namespace msda
