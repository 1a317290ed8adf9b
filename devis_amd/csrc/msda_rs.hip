// msda_rs.hip -- "resident-slab" kernels: forward and backward gather pass of the DeVIS shapes (D = 32).
#include "msda_common.h"

namespace msda {
namespace {

// ------------------------------------------------------------------------------------------------
// "resident-slab" kernels (round 2): levels 1..L-1 of one source frame live in LDS, tap records in registers
// ------------------------------------------------------------------------------------------------
// The slab kernels above keep 70 KiB of per-wave tap records in LDS, which leaves room for levels 2-3 only
// (50 % of the taps); everything else crosses the L1 path at ~25 B/clk/CU.  Here the records never touch
// LDS, so ~156 KiB of the 160 are slab (levels 1-3 of the DeVIS pyramids: 75 % of the taps):
//   * a 1024-thread workgroup owns (clip, head, a run of up to NT*16 row tiles); its waves keep the accumulators
//     of NT tiles in registers while the workgroup walks the clip's SOURCE FRAMES; per frame the slab
//     value[frame, levels >= l0, head, :] is staged by LDS-DMA (once per NT*16 tiles instead of once per 16),
//     then every wave runs, for each of its tiles, the slots of that tile that read the frame;
//   * a row (query, head) is served by ONE QUAD: 16 rows per wave, lane c of the quad holding channels
//     [4c, 4c+4) of both halves of the row (D = 32).  Lane c also fetches point (g0 + c) of the row and turns it
//     into "point data" (fractions, attention weight, top-left pixel, validity bits).  In step R the quad's
//     lanes read lane R's data through quad_perm DPP operands folded into the consuming VALU instruction
//     (v_and/v_add/v_fmac/v_mul ..._dpp: no LDS crossbar), each lane deriving the address and weight of ITS
//     corner (lane & 3); the four corners of the point are then read with 16-byte loads whose addresses and
//     weights come from lanes 0..3 of the quad, again by DPP.  (Measured, scripts/ubench/valu_rate.hip: a DPP
//     operand makes a VALU instruction half rate -- 4.3 vs 2.3 clk per wave64 instruction -- so a weight is
//     moved once per corner with v_mov_b32_dpp and then feeds 8 plain v_fmac_f32: that is why a row is a quad
//     with 8 channels per lane and not 8 lanes with 4.)
//   * quads alternate which 64-byte half of a 128-byte row they read first, which halves the LDS bank conflicts
//     of the 16-lane ds_read_b128 groups (4 quads = 4 half rows on 4 different 16-bank quarters when row
//     parities differ);
//   * a corner outside the map reads a zero row kept in LDS (slab levels) or an out-of-range buffer offset
//     (other levels: buffer loads return 0 without touching memory), so a non-finite value at an unrelated
//     pixel can never leak into a row that does not sample it.
template <typename T> constexpr int rs_row_bytes() { return 32 * (int)sizeof(T); }

// The 8 channels a lane holds of one pixel row whose (this lane's) slice starts at LDS byte address `a` / buffer byte
// offset `a`: 4-byte types -- [4c, 4c+4) of both 64-byte halves, the second half at a ^ 64 (LDS) or a + delta2
// (memory); 2-byte types -- the 8 contiguous channels [8c, 8c+8) = ONE 16-byte load.
template <typename T, bool SLAB>
__device__ __forceinline__ void rs_load_row8(const unsigned char *lds_raw, __amdgpu_buffer_rsrc_t rsrc, int a, int delta2,
                                             float (&v)[8])
{
    if constexpr (sizeof(T) == 4) {
        if constexpr (SLAB) {
            const float4 q1 = *reinterpret_cast<const float4 *>(lds_raw + a);
            const float4 q2 = *reinterpret_cast<const float4 *>(lds_raw + (a ^ 64));
            v[0] = q1.x; v[1] = q1.y; v[2] = q1.z; v[3] = q1.w; v[4] = q2.x; v[5] = q2.y; v[6] = q2.z; v[7] = q2.w;
        } else {
            const u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, a, 0, 0);
            const u32x4 q2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, a + delta2, 0, 0);
            v[0] = __uint_as_float(q1.x); v[1] = __uint_as_float(q1.y); v[2] = __uint_as_float(q1.z); v[3] = __uint_as_float(q1.w);
            v[4] = __uint_as_float(q2.x); v[5] = __uint_as_float(q2.y); v[6] = __uint_as_float(q2.z); v[7] = __uint_as_float(q2.w);
        }
    } else {
        u32x4 q;
        if constexpr (SLAB) q = *reinterpret_cast<const u32x4 *>(lds_raw + a);
        else q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, a, 0, 0);
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (sizeof(T) == 2 && std::is_same<T, bf16_t>::value) {
                v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
            } else {
                const float2 f = __half22float2(*reinterpret_cast<const __half2 *>(&w[i]));
                v[2 * i] = f.x; v[2 * i + 1] = f.y;
            }
        }
    }
}

#define MSDA_QP(s) "quad_perm:[" #s "," #s "," #s "," #s "] row_mask:0xf bank_mask:0xf"

// dst = quad_lane_R(src) <op> other, R a compile-time constant
#define MSDA_DEF_QUAD_OP(name, ctype, mnem)                                                                \
    template <int R> __device__ __forceinline__ ctype name(ctype src, ctype other)                         \
    {                                                                                                      \
        ctype r;                                                                                           \
        if constexpr (R == 0) asm(mnem " %0, %1, %2 " MSDA_QP(0) : "=v"(r) : "v"(src), "v"(other));       \
        else if constexpr (R == 1) asm(mnem " %0, %1, %2 " MSDA_QP(1) : "=v"(r) : "v"(src), "v"(other));  \
        else if constexpr (R == 2) asm(mnem " %0, %1, %2 " MSDA_QP(2) : "=v"(r) : "v"(src), "v"(other));  \
        else asm(mnem " %0, %1, %2 " MSDA_QP(3) : "=v"(r) : "v"(src), "v"(other));                        \
        return r;                                                                                          \
    }
MSDA_DEF_QUAD_OP(quad_and, int, "v_and_b32_dpp")
MSDA_DEF_QUAD_OP(quad_add, int, "v_add_u32_dpp")
MSDA_DEF_QUAD_OP(quad_mul, float, "v_mul_f32_dpp")
#undef MSDA_DEF_QUAD_OP

// acc += quad_lane_R(src) * other
template <int R> __device__ __forceinline__ void quad_fmac(float &acc, float src, float other)
{
    if constexpr (R == 0) asm("v_fmac_f32_dpp %0, %1, %2 " MSDA_QP(0) : "+v"(acc) : "v"(src), "v"(other));
    else if constexpr (R == 1) asm("v_fmac_f32_dpp %0, %1, %2 " MSDA_QP(1) : "+v"(acc) : "v"(src), "v"(other));
    else if constexpr (R == 2) asm("v_fmac_f32_dpp %0, %1, %2 " MSDA_QP(2) : "+v"(acc) : "v"(src), "v"(other));
    else asm("v_fmac_f32_dpp %0, %1, %2 " MSDA_QP(3) : "+v"(acc) : "v"(src), "v"(other));
}

// A VGPR written by a VALU instruction may be read through DPP only two wait states later; inline asm is
// invisible to the compiler's hazard recogniser, so values about to be read that way pass through a fence.
__device__ __forceinline__ void dpp_fence(float &a, float &b, float &c, int &d, int &e)
{
    asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
}

// lanes 0..3 of the quad hold the records of corners 0..3: their addresses (+ this lane's offset) and weights
__device__ __forceinline__ void quad_corner_records(int addr, float w, int lane_off, int (&A)[4], float (&W)[4])
{
    asm volatile("s_nop 1\n"
                 "v_add_u32_dpp %0, %8, %9 " MSDA_QP(0) "\n v_add_u32_dpp %1, %8, %9 " MSDA_QP(1) "\n"
                 "v_add_u32_dpp %2, %8, %9 " MSDA_QP(2) "\n v_add_u32_dpp %3, %8, %9 " MSDA_QP(3) "\n"
                 "v_mov_b32_dpp %4, %10 " MSDA_QP(0) "\n v_mov_b32_dpp %5, %10 " MSDA_QP(1) "\n"
                 "v_mov_b32_dpp %6, %10 " MSDA_QP(2) "\n v_mov_b32_dpp %7, %10 " MSDA_QP(3)
                 : "=&v"(A[0]), "=&v"(A[1]), "=&v"(A[2]), "=&v"(A[3]), "=&v"(W[0]), "=&v"(W[1]), "=&v"(W[2]), "=&v"(W[3])
                 : "v"(addr), "v"(lane_off), "v"(w));
}


// Per-lane constants of the corner a lane serves inside its quad (corner = lane & 3: bit 0 = x+1, bit 1 = y+1).
struct RsLane {
    int vmask, dymask, dx;                  // validity bit of the corner in Wb; (y+1 ? 0xffffff : 0); x+1
    int off1, delta2;                       // byte offset of the lane's first 16-byte slice inside a pixel row; second = first + delta2
    float fy0, fys, fx0, fxs;               // corner weight = (fy0 + fys * lh) * (fx0 + fxs * lw)
};

// What the lane that fetched a point shows to its quad.
struct RsPoint { float lh, lw, a; int pbase, Wb; };     // Wb = W | validity bits << 24

// levels >= l0 of source frame f (head m) -> LDS slab, 16 bytes per lane by LDS-DMA (8 lanes per pixel)
template <typename T>
__device__ __forceinline__ void rs_stage_slab(const Params &p, T *slab, int clip, int m, int f, int px0, int npx,
                                              int wave, int lane)
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Point data of THIS lane's own point (x, y, a) at level `lvl` of source frame f: cuh:285-288 (pixel coords,
// range test), cuh:38-53 (floor, fractions), cuh:56-80 (per-corner validity).  Levels of the slab are addressed
// by their pixel index inside the slab, the others by their pixel index inside the clip.
__device__ __forceinline__ RsPoint rs_point(float x, float y, float a, int lvl, int l0, int fS,
                                            const int *s_H, const int *s_W, const int *s_lsi, const int *s_sst)
{
    const int H = s_H[lvl], W = s_W[lvl];
    const int base = lvl >= l0 ? s_sst[lvl] : fS + s_lsi[lvl];
    const float h_im = __fsub_rn(__fmul_rn(y, (float)H), 0.5f);
    const float w_im = __fsub_rn(__fmul_rn(x, (float)W), 0.5f);
    const bool rng = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;     // false for NaN
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int yl = rng ? (int)hf : 0, xl = rng ? (int)wf : 0;
    RsPoint r;
    r.lh = rng ? h_im - hf : 0.f;
    r.lw = rng ? w_im - wf : 0.f;
    r.a = rng ? a : 0.f;
    const int vy0 = yl >= 0, vy1 = yl + 1 <= H - 1, vx0 = xl >= 0, vx1 = xl + 1 <= W - 1;
    const int bits = rng ? ((vy0 & vx0) | ((vy0 & vx1) << 1) | ((vy1 & vx0) << 2) | ((vy1 & vx1) << 3)) : 0;
    r.pbase = base + yl * W + xl;
    r.Wb = W | (bits << 24);
    return r;
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

// 4 x 4 transpose inside a quad: lane c, element e  <-  lane e, element c  (two butterfly stages of one DPP move and
// three selects per pair of elements)
template <typename V>
__device__ __forceinline__ void quad_transpose4(V (&a)[4], int c)
{
    static_assert(sizeof(V) == 4, "32-bit elements");
    auto xchg = [&](int lo, int hi, bool up, int ctrl) {
        const V send = up ? a[lo] : a[hi];
        int bits;
        __builtin_memcpy(&bits, &send, 4);
        const int got = ctrl == 1 ? __builtin_amdgcn_mov_dpp(bits, 0xB1, 0xf, 0xf, true)      // quad_perm [1,0,3,2]
                                  : __builtin_amdgcn_mov_dpp(bits, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
        V r;
        __builtin_memcpy(&r, &got, 4);
        a[lo] = up ? r : a[lo];
        a[hi] = up ? a[hi] : r;
    };
    xchg(0, 1, (c & 1) != 0, 1); xchg(2, 3, (c & 1) != 0, 1);
    xchg(0, 2, (c & 2) != 0, 2); xchg(1, 3, (c & 2) != 0, 2);
}
// a[i] = v for the (wave-uniform) index i: four selects instead of a dynamically indexed register array
template <typename V> __device__ __forceinline__ void set4(V (&a)[4], int i, V v)
{
    a[0] = i == 0 ? v : a[0]; a[1] = i == 1 ? v : a[1]; a[2] = i == 2 ? v : a[2]; a[3] = i == 3 ? v : a[3];
}

// a[i] for the (wave-uniform) index i
template <typename V> __device__ __forceinline__ V get4(const V (&a)[4], int i)
{
    return i == 0 ? a[0] : i == 1 ? a[1] : i == 2 ? a[2] : a[3];
}
// The 16 sampling points (4 levels x 4 points) of one (row, slot), loaded as whole rows: lane c of the row's quad reads
// points 4c..4c+3 (32 + 16 contiguous bytes for 4-byte types) and the quad transposes, so that element g of lane c is
// point c of level g -- three 16-byte loads per slot instead of eight 8- / 4-byte loads, one memory latency instead of four.
template <typename T>
__device__ __forceinline__ void load_slot_points(const T *loc, const T *aw, int64_t idx0, int cor, bool live,
                                                 float (&xs)[4], float (&ys)[4], float (&as)[4])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) { xs[i] = ys[i] = -10.f; as[i] = 0.f; }       // far outside every map
    if (live) {
        float xy[8];
        if constexpr (sizeof(T) == 2) {
            Store<T>::load(loc + 2 * (idx0 + 4 * cor), xy);
        } else {
            float lo[4], hi[4];
            Store<T>::load(loc + 2 * (idx0 + 4 * cor), lo); Store<T>::load(loc + 2 * (idx0 + 4 * cor) + 4, hi);
#pragma unroll
            for (int i = 0; i < 4; ++i) { xy[i] = lo[i]; xy[4 + i] = hi[i]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { xs[i] = xy[2 * i]; ys[i] = xy[2 * i + 1]; }
        SlabStore<T>::load(aw + idx0 + 4 * cor, as);
    }
    quad_transpose4(xs, cor); quad_transpose4(ys, cor); quad_transpose4(as, cor);
}

template <typename T, int NT>
__global__ void __launch_bounds__(kRsThreads)
msda_fwd_rs_kernel(const Params p, int slab_bytes, int parts)
{
    constexpr int RPW = kRsRows, D = 32, ROWB = rs_row_bytes<T>(), ROWSH = ROWB == 128 ? 7 : 6;
    constexpr bool kHalf = sizeof(T) == 2;
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
    RsLane ln;
    ln.vmask = 1 << (24 + cor); ln.dymask = (cor & 2) ? 0xffffff : 0; ln.dx = cor & 1;
    ln.off1 = kHalf ? cor * 16 : cor * 16 + hsw * 64; ln.delta2 = hsw ? -64 : 64;
    ln.fy0 = (cor & 2) ? 0.f : 1.f; ln.fys = (cor & 2) ? 1.f : -1.f;
    ln.fx0 = (cor & 1) ? 0.f : 1.f; ln.fxs = (cor & 1) ? 1.f : -1.f;
    const int pixB = p.v_pix * (int)sizeof(T);
    // buffer resource over value[clip, :, m, :] (stride 0 = raw, num_records in bytes): out-of-range -> 0
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

    for (int f = 0; f < p.frames; ++f) {
        __syncthreads();                                   // every wave is done with the previous slab
        if (l0 < L) rs_stage_slab<T>(p, slab, clip, m, f, sh.px0, sh.npx, wave, lane);
        __syncthreads();
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
#pragma unroll 1
            while (todo) {                                     // sl = -1: the tile's current-frame points
                const int sl = (int)__builtin_ctz(todo) - 1;
                todo &= todo - 1;
                const T *loc = static_cast<const T *>(sl < 0 ? p.locA : p.locB);
                const T *aw = static_cast<const T *>(sl < 0 ? p.awA : p.awB);
                const int P = sl < 0 ? p.PA : p.PB;
                const int LP = (sl < 0 ? p.LA : p.LB) * P;
                const int npts = (sl < 0 ? p.LA : L) * P;
                const int64_t idx0 = row * LP + (sl < 0 ? 0 : sl * L * P);
                const unsigned invP = (65536u + (unsigned)P - 1u) / (unsigned)P;      // kk / P for kk * P < 2^16
                const int first_slab_pt = l0 * P;              // points of levels >= l0 read the slab
                const bool wide = p.wide_loads && P == 4 && npts == 16;           // (uniform) see load_slot_points
                float xs[4], ys[4], as[4];
                if (wide) load_slot_points<T>(loc, aw, idx0, cor, live, xs, ys, as);
#pragma unroll 1
                for (int g0 = 0; g0 < npts; g0 += 4) {
                    const int kk = g0 + cor;
                    float x = -10.f, y = -10.f, a = 0.f;       // far outside every map
                    if (wide) {
                        x = get4(xs, g0 >> 2); y = get4(ys, g0 >> 2); a = get4(as, g0 >> 2);
                    } else if (live && kk < npts) {
                        load_xy(loc + 2 * (idx0 + kk), x, y);
                        a = Store<T>::get(aw + idx0 + kk);
                    }
                    const int lvl = min((int)(((unsigned)kk * invP) >> 16), L - 1);
                    RsPoint pt = rs_point(x, y, a, lvl, l0, fS, sh.H, sh.W, sh.lsi, sh.sst);
                    dpp_fence(pt.lh, pt.lw, pt.a, pt.pbase, pt.Wb);
                    // one step = one point of the 16 rows: this lane's corner record, then the four corners
                    auto step = [&](auto Rc, auto Sc) {
                        constexpr int R = decltype(Rc)::value;
                        constexpr bool SLAB = decltype(Sc)::value;
                        const int vb = quad_and<R>(pt.Wb, ln.vmask);
                        const int tw = quad_and<R>(pt.Wb, ln.dymask);
                        const int pix = quad_add<R>(pt.pbase, ln.dx) + tw;
                        int addr = SLAB ? (pix << ROWSH) : (int)((unsigned)pix * (unsigned)pixB);
                        addr = vb ? addr : (SLAB ? sh.zero_off : (int)0x80000000u);
                        float wy = ln.fy0, wx = ln.fx0;
                        quad_fmac<R>(wy, pt.lh, ln.fys);
                        quad_fmac<R>(wx, pt.lw, ln.fxs);
                        const float w = quad_mul<R>(pt.a, wy * wx);
                        int A[4];
                        float W[4];
                        quad_corner_records(addr, w, ln.off1, A, W);
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            float v[8];
#if defined(__HIP_DEVICE_COMPILE__)
                            rs_load_row8<T, SLAB>(lds_raw, rsrc, A[s], ln.delta2, v);
#endif
#pragma unroll
                            for (int c = 0; c < 8; ++c) wacc[c] = fmaf(W[s], v[c], wacc[c]);
                            // corner by corner: the next corner's loads are not hoisted above these FMAs (measured: 0.466 ->
                            // 0.430 ms; eight loads in flight per wave only queue up in the LDS / TA pipes)
                            asm volatile("" ::: "memory");
                        }
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
                Store<T>::store(o + ln.off1 / 4, a1);
                Store<T>::store(o + (ln.off1 + ln.delta2) / 4, a2);
            }
        }
    });
}

// Backward gather pass (grad_loc / grad_attn) on the resident slab: same workgroup / tile / quad geometry as
// msda_fwd_rs_kernel, but nothing is carried across source frames -- every (tile, slot) writes its own gradients --
// so there are no accumulator sets and a wave may take any number of tiles.  Per point the four dots
// <grad_out row, corner k> (cuh:123-158) are 8 FMAs per corner and lane, reduced over the quad with two DPP adds;
// lane R of the quad keeps the dots of point R, and after the group's four points every lane finishes ITS point and
// stores its (grad_x, grad_y, grad_attn) directly: the 4 points of a group are 32 + 16 contiguous bytes per row.
// Also leaves the per-point culling records (top tap row as int16) the scatter pass reads.
__device__ __forceinline__ void quad_corner_addrs(int addr, int lane_off, int (&A)[4])
{
    asm volatile("s_nop 1\n"
                 "v_add_u32_dpp %0, %4, %5 " MSDA_QP(0) "\n v_add_u32_dpp %1, %4, %5 " MSDA_QP(1) "\n"
                 "v_add_u32_dpp %2, %4, %5 " MSDA_QP(2) "\n v_add_u32_dpp %3, %4, %5 " MSDA_QP(3)
                 : "=&v"(A[0]), "=&v"(A[1]), "=&v"(A[2]), "=&v"(A[3]) : "v"(addr), "v"(lane_off));
}

// d[k] <- sum of d[k] over the four lanes of the quad (all lanes get the total)
__device__ __forceinline__ void quad_sum4(float (&d)[4])
{
    asm volatile("s_nop 1\n"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "s_nop 0\n"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                 "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                 "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                 "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                 : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
}

template <typename T>
__global__ void __launch_bounds__(kRsThreads)
msda_bwd_rs_kernel(const Params p, int slab_bytes, int parts)
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
    const int part = (int)(lin % (unsigned)parts), m = (int)((lin / (unsigned)parts) % (unsigned)p.M);
    const int clip = (int)(lin / ((unsigned)parts * (unsigned)p.M));
    const int tiles_per_group = (p.Lq + RPW - 1) / RPW, tiles_per_clip = p.frames * tiles_per_group;
    const int tpw = (tiles_per_clip + parts - 1) / parts;
    const int tile_lo = part * tpw + wave, tile_hi = min((part + 1) * tpw, tiles_per_clip);

    const int j = lane / 4, cor = lane & 3, hsw = j & 1;
    RsLane ln;
    ln.vmask = 1 << (24 + cor); ln.dymask = (cor & 2) ? 0xffffff : 0; ln.dx = cor & 1;
    ln.off1 = kHalf ? cor * 16 : cor * 16 + hsw * 64; ln.delta2 = hsw ? -64 : 64;
    ln.fy0 = ln.fys = ln.fx0 = ln.fxs = 0.f;      // (corner weights are not needed for the dots)
    const int pixB = p.v_pix * (int)sizeof(T);
    const char *vbase = reinterpret_cast<const char *>(static_cast<const T *>(p.value) + clip * p.v_clip + m * p.v_head);
    const unsigned vbytes = (unsigned)(((int64_t)p.frames * p.S - 1) * pixB + ROWB);
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(vbase), 0, (int)vbytes, 0x00020000);
#endif
    const bool records = p.bbox != nullptr;        // per-point culling records (host: only with cull_points)

    for (int f = 0; f < p.frames; ++f) {
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
                    const float4 g1 = *reinterpret_cast<const float4 *>(go + ln.off1 / 4);
                    const float4 g2 = *reinterpret_cast<const float4 *>(go + (ln.off1 + ln.delta2) / 4);
                    g[0] = g1.x; g[1] = g1.y; g[2] = g1.z; g[3] = g1.w; g[4] = g2.x; g[5] = g2.y; g[6] = g2.z; g[7] = g2.w;
                }
            }
#pragma unroll 1
            while (todo) {                                     // sl = -1: the tile's current-frame points
                const int sl = (int)__builtin_ctz(todo) - 1;
                todo &= todo - 1;
                const T *loc = static_cast<const T *>(sl < 0 ? p.locA : p.locB);
                const T *aw = static_cast<const T *>(sl < 0 ? p.awA : p.awB);
                T *gloc = static_cast<T *>(sl < 0 ? p.glocA : p.glocB);
                T *gaw = static_cast<T *>(sl < 0 ? p.gawA : p.gawB);
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
                if (wide_ld) load_slot_points<T>(loc, aw, idx0, cor, live, xs, ys, as);
#pragma unroll 1
                for (int g0 = 0; g0 < npts; g0 += 4) {
                    const int kk = g0 + cor;
                    const bool mine = live && kk < npts;
                    float x = -10.f, y = -10.f, a = 0.f;       // far outside every map
                    if (wide_ld) {
                        x = get4(xs, g0 >> 2); y = get4(ys, g0 >> 2); a = get4(as, g0 >> 2);
                    } else if (mine) {
                        load_xy(loc + 2 * (idx0 + kk), x, y);
                        a = Store<T>::get(aw + idx0 + kk);
                    }
                    const int lvl = min((int)(((unsigned)kk * invP) >> 16), L - 1);
                    // own point: fractions, validity, top-left pixel (as rs_point) + what the gradients need
                    const int H = sh.H[lvl], W = sh.W[lvl];
                    const int base = lvl >= l0 ? sh.sst[lvl] : fS + sh.lsi[lvl];
                    const float h_im = __fsub_rn(__fmul_rn(y, (float)H), 0.5f);
                    const float w_im = __fsub_rn(__fmul_rn(x, (float)W), 0.5f);
                    const bool rng = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
                    const float hf = floorf(h_im), wf = floorf(w_im);
                    const int yl = rng ? (int)hf : 0, xl = rng ? (int)wf : 0;
                    RsPoint pt;
                    pt.lh = rng ? h_im - hf : 0.f;
                    pt.lw = rng ? w_im - wf : 0.f;
                    pt.a = rng ? a : 0.f;
                    const int vy0 = yl >= 0, vy1 = yl + 1 <= H - 1, vx0 = xl >= 0, vx1 = xl + 1 <= W - 1;
                    const int bits = rng ? ((vy0 & vx0) | ((vy0 & vx1) << 1) | ((vy1 & vx0) << 2) | ((vy1 & vx1) << 3)) : 0;
                    pt.pbase = base + yl * W + xl;
                    pt.Wb = W | (bits << 24);
                    if (wide) set4(wr, g0 >> 2, bits ? min(yl, 32767) : kNoRow16);
                    if (records && mine && !wide) {      // the point's top tap row, for the scatter's band test
                        const int pin = kk - lvl * P;
                        short *rec = reinterpret_cast<short *>(p.bbox + (((group * p.M + m) * VL + vl0 + lvl) * p.Lq + q0 + j) * 2);
                        rec[pin] = bits ? (short)min(yl, 32767) : (short)kNoRow16;
                        if (pin == 0)
                            for (int u = P; u < 4; ++u) rec[u] = (short)kNoRow16;
                    }
                    dpp_fence(pt.lh, pt.lw, pt.a, pt.pbase, pt.Wb);
                    float k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f;      // the dots of THIS lane's point
                    auto step = [&](auto Rc, auto Sc) {
                        constexpr int R = decltype(Rc)::value;
                        constexpr bool SLAB = decltype(Sc)::value;
                        const int vb = quad_and<R>(pt.Wb, ln.vmask);
                        const int tw = quad_and<R>(pt.Wb, ln.dymask);
                        const int pix = quad_add<R>(pt.pbase, ln.dx) + tw;
                        int addr = SLAB ? (pix << ROWSH) : (int)((unsigned)pix * (unsigned)pixB);
                        addr = vb ? addr : (SLAB ? sh.zero_off : (int)0x80000000u);
                        int A[4];
                        quad_corner_addrs(addr, ln.off1, A);
                        float d[4];
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            float v[8];
#if defined(__HIP_DEVICE_COMPILE__)
                            rs_load_row8<T, SLAB>(lds_raw, rsrc, A[s], ln.delta2, v);
#endif
                            // (measured: issuing all eight loads of the point ahead of the dots is SLOWER, 0.63 -> 0.67 ms)
                            float acc = g[0] * v[0];
#pragma unroll
                            for (int c = 1; c < 8; ++c) acc = fmaf(g[c], v[c], acc);
                            d[s] = acc;
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
                    // every lane finishes its own point (cuh:123-158 on the reduced dots; dots of corners outside
                    // the map are 0: their loads returned zeros)
                    {
                        const float lh = pt.lh, lw = pt.lw, hh = 1.f - lh, hw = 1.f - lw;
                        const float g_aw = (hh * hw) * k0 + (hh * lw) * k1 + (lh * hw) * k2 + (lh * lw) * k3;
                        const float g_w = hh * (k1 - k0) + lh * (k3 - k2);
                        const float g_h = hw * (k2 - k0) + lw * (k3 - k1);
                        const float gx = (float)W * g_w * pt.a, gy = (float)H * g_h * pt.a;
                        if (wide) {
                            set4(wx, g0 >> 2, gx); set4(wy, g0 >> 2, gy); set4(wa, g0 >> 2, g_aw);
                        } else if (mine) {
                            Store<T>::put(gloc + 2 * (idx0 + kk), gx);
                            Store<T>::put(gloc + 2 * (idx0 + kk) + 1, gy);
                            Store<T>::put(gaw + idx0 + kk, g_aw);
                        }
                    }
                }
                if (wide) {
                    // lane c held point c of every level; after the transposes it holds the four points of level c
                    quad_transpose4(wx, cor); quad_transpose4(wy, cor); quad_transpose4(wa, cor); quad_transpose4(wr, cor);
                    if (live) {
                        const float xy[8] = {wx[0], wy[0], wx[1], wy[1], wx[2], wy[2], wx[3], wy[3]};
                        T *gl = gloc + 2 * (idx0 + 4 * cor);
                        if constexpr (kHalf) {
                            Store<T>::store(gl, xy);
                        } else {
                            // (non-temporal: the 309 MB of results must not evict the level-0 lines the gathers live on)
                            typedef float f32x4 __attribute__((ext_vector_type(4)));
                            __builtin_nontemporal_store((f32x4){xy[0], xy[1], xy[2], xy[3]}, reinterpret_cast<f32x4 *>(gl));
                            __builtin_nontemporal_store((f32x4){xy[4], xy[5], xy[6], xy[7]}, reinterpret_cast<f32x4 *>(gl + 4));
                        }
                        if constexpr (kHalf) {
                            SlabStore<T>::store(gaw + idx0 + 4 * cor, wa);
                        } else {
                            typedef float f32x4 __attribute__((ext_vector_type(4)));
                            __builtin_nontemporal_store((f32x4){wa[0], wa[1], wa[2], wa[3]}, reinterpret_cast<f32x4 *>(gaw + idx0 + 4 * cor));
                        }
                        if (records)
                            *reinterpret_cast<int2 *>(p.bbox + (((group * p.M + m) * VL + vl0 + cor) * p.Lq + q0 + j) * 2) =
                                make_int2((wr[0] & 0xffff) | (wr[1] << 16), (wr[2] & 0xffff) | (wr[3] << 16));
                    }
                }
            }
        }
    }
}

template <typename T, int NT>
int fwd_rs(const Params &p, int parts, unsigned grid, hipStream_t stream, const char *what)
{
    static LdsGrant granted;
    const size_t total = (size_t)kRsSlabBytes + kRsTailBytes;
    if (const int rc = grant_lds(reinterpret_cast<const void *>(&msda_fwd_rs_kernel<T, NT>), total, granted,
                                 "the resident-slab forward kernel")) return rc;
    hipLaunchKernelGGL((msda_fwd_rs_kernel<T, NT>), dim3(grid), dim3(kRsThreads), total, stream, p, kRsSlabBytes, parts);
    return check_launch(what);
}

template <typename T>
int fwd_rs_nt(int nt, const Params &p, int parts, unsigned grid, hipStream_t stream)
{
    switch (nt) {
        case 4: return fwd_rs<T, 4>(p, parts, grid, stream, "msda forward (resident-slab kernel, 4 tiles per wave)");
        case 2: return fwd_rs<T, 2>(p, parts, grid, stream, "msda forward (resident-slab kernel, 2 tiles per wave)");
        default: return fwd_rs<T, 1>(p, parts, grid, stream, "msda forward (resident-slab kernel, 1 tiles per wave)");
    }
}

template <typename T>
int bwd_rs(const Params &p, int parts, unsigned grid, hipStream_t stream)
{
    static LdsGrant granted;
    const size_t total = (size_t)kRsSlabBytes + kRsTailBytes;
    if (const int rc = grant_lds(reinterpret_cast<const void *>(&msda_bwd_rs_kernel<T>), total, granted,
                                 "the resident-slab gather-pass kernel")) return rc;
    hipLaunchKernelGGL((msda_bwd_rs_kernel<T>), dim3(grid), dim3(kRsThreads), total, stream, p, kRsSlabBytes, parts);
    return check_launch("msda backward (resident-slab kernel, grad_loc/grad_attn)");
}

}  // namespace

int launch_fwd_rs(int dtype, int nt, const Params &p, int parts, unsigned grid, hipStream_t stream)
{
    switch (dtype) {
        case MSDA_F32: return fwd_rs_nt<float>(nt, p, parts, grid, stream);
        case MSDA_BF16: return fwd_rs_nt<bf16_t>(nt, p, parts, grid, stream);
        case MSDA_F16: return fwd_rs_nt<f16_t>(nt, p, parts, grid, stream);
        default: return fail(MSDA_ERR_DTYPE, "msda: unknown dtype code%s");
    }
}

int launch_bwd_rs(int dtype, const Params &p, int parts, unsigned grid, hipStream_t stream)
{
    switch (dtype) {
        case MSDA_F32: return bwd_rs<float>(p, parts, grid, stream);
        case MSDA_BF16: return bwd_rs<bf16_t>(p, parts, grid, stream);
        case MSDA_F16: return bwd_rs<f16_t>(p, parts, grid, stream);
        default: return fail(MSDA_ERR_DTYPE, "msda: unknown dtype code%s");
    }
}

}  // namespace msda
