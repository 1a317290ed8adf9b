// msda_hip.hip -- hand-written gfx950 (CDNA4, wave64) kernels for multi-scale deformable attention
// and the extern "C" ABI declared in include/msda.h.
//
// What is computed (semantics of the reference kernels being replaced,
// /root/reference/src/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:33-299 forward, :87-920 backward):
//   out[n,q,m,:] = sum_{l,p} a[n,q,m,l,p] * bilinear(value_l[:, :, m, :], x*W_l-0.5, y*H_l-0.5)
// with zero padding, a point contributing only if -1 < h < H_l and -1 < w < W_l (cuh:288), and the
// matching gradients wrt value (scatter-add), sampling locations and attention weights.
//
// Design (MI355X-first, not a translation of the reference's one-thread-per-channel CUDA kernels):
//   * "tile" kernels: ONE wave64 owns RPW = 64/G (query, head) rows of the same head, G lanes per row,
//     each lane holding VEC contiguous channels (16 B: float4 or 8 x bf16/f16), so every bilinear
//     corner is one coalesced D*sizeof(T) segment per row and one 16-B load per lane.
//   * the wave first turns its rows' (x, y, weight) triples into "tap records" in LDS -- 4 element
//     offsets + 4 premultiplied weights per sampling point, computed ONCE per point instead of once
//     per channel lane -- then the gather loop is LDS-broadcast read + 4 global loads + FMAs.
//   * out-of-range corners become (offset 0, weight 0): the gather loop is branch free.
//   * a per-wave "virtual level" table in LDS holds (H, W, first pixel) for every level of every
//     source frame, so the plain op (levels of one map) and the fused temporal op (current frame +
//     `window` other frames of the clip, ms_deform_attn.py:325-364) are the SAME kernel.
//   * blockIdx -> (query tile, head) with head = blockIdx % M: workgroups are dealt round-robin to the
//     8 XCDs, so with M = 8 each XCD's private 4 MiB L2 only ever sees ONE head's 1/8 slice of the
//     value maps (XCD-aware mapping; affects speed only, never results).
//   * "slab" variants of the forward and of the backward gather pass: 16 waves share a workgroup and an
//     LDS slab of the small pyramid levels of one source frame (staged with LDS-DMA), taking half of the
//     taps off the L1/TA path.
//   * backward = two passes.  Gather pass (grad_loc / grad_attn): per-point partial dot products
//     <grad_out, corner_k> are reduced across the G lanes with DPP butterflies (no LDS round trip, no
//     serial thread-0 sum as in cuh:376-394), lane pp % G keeps the sums of point pp and G points are
//     finished at once; it also leaves per-point culling records.  Scatter pass (grad_value): privatised
//     in LDS per (clip, frame, head, band of pixel rows), accumulated in fp64 with ds_add_f64, software-
//     pipelined over the items, hit records handed to the lane teams by DPP; grad_value is OVERWRITTEN.
//     A one-kernel backward with hardware float global atomics remains as the fallback.
//   * generic kernels (any D, any dtype incl. fp64) back the shapes the tile kernels do not take.
//   * msda_prep_kernel: the modules' joint softmax + sampling-location arithmetic as one pass each way.
//   * no implicit FMA contraction in this file (see the pragma below).
//
// No CUDA compatibility layer, no hipify output: this file targets gfx950 only.

#include <hip/hip_runtime.h>
#include <utility>
#include <type_traits>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "msda.h"

// No implicit FMA contraction anywhere in this file: HIP's __fmul_rn / __fsub_rn are plain `*` / `-` unless
// OCML_BASIC_ROUNDED_OPERATIONS is defined, and hipcc contracts by default, so `x * W - 0.5` could become one
// FMA in some inlining contexts and not in others -- forward, gather pass and scatter would then disagree with
// each other (and with the oracle) about the pixel cell of a point that sits within one ulp of a border.
// Every FMA the kernels want is spelled fmaf().  (devis_amd/build.py also passes -ffp-contract=off.)
#pragma clang fp contract(off)

#ifndef MSDA_GRP_F32
#define MSDA_GRP_F32 512
#endif
#ifndef MSDA_GRP_16
#define MSDA_GRP_16 512
#endif

namespace {

constexpr int kWave = 64;   // gfx950 wavefront
constexpr int kPch = 16;    // sampling points per LDS chunk (= L*P of the DeVIS configs)

// ------------------------------------------------------------------------------------------------
// storage-type helpers: everything is computed in fp32 (fp64 for double)
// ------------------------------------------------------------------------------------------------
typedef __hip_bfloat16 bf16_t;
typedef __half f16_t;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// raw lane vector (as a buffer load returns it) -> fp32 channels
__device__ __forceinline__ void unpack_raw(const float *, u32x4 q, float (&v)[4])
{
    v[0] = __uint_as_float(q.x); v[1] = __uint_as_float(q.y); v[2] = __uint_as_float(q.z); v[3] = __uint_as_float(q.w);
}
__device__ __forceinline__ void unpack_pair(const __hip_bfloat16 *, uint32_t w, float &lo, float &hi)
{
    lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u);      // bf16 -> f32 is a 16-bit shift
}
__device__ __forceinline__ void unpack_pair(const __half *, uint32_t w, float &lo, float &hi)
{
    const float2 f = __half22float2(*reinterpret_cast<const __half2 *>(&w));
    lo = f.x; hi = f.y;
}
template <typename T> __device__ __forceinline__ void unpack_raw(const T *t, u32x4 q, float (&v)[8])
{
    unpack_pair(t, q.x, v[0], v[1]); unpack_pair(t, q.y, v[2], v[3]);
    unpack_pair(t, q.z, v[4], v[5]); unpack_pair(t, q.w, v[6], v[7]);
}
template <typename T> __device__ __forceinline__ void unpack_raw(const T *t, u32x2 q, float (&v)[4])
{
    unpack_pair(t, q.x, v[0], v[1]); unpack_pair(t, q.y, v[2], v[3]);
}

template <typename T> struct Store;   // VEC = elements per 16-byte lane vector
template <> struct Store<float> {
    static constexpr int VEC = 4;
    __device__ static float get(const float *p) { return *p; }
    __device__ static void put(float *p, float v) { *p = v; }
    __device__ static void load(const float *p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ static void store(float *p, const float (&v)[4]) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct Store<bf16_t> {
    static constexpr int VEC = 8;
    __device__ static float get(const bf16_t *p) { return __bfloat162float(*p); }
    __device__ static void put(bf16_t *p, float v) { *p = __float2bfloat16(v); }
    __device__ static void load(const bf16_t *p, float (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4 *>(p);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {       // bf16 -> f32 is a 16-bit shift
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ static void store(bf16_t *p, const float (&v)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t lo = __float2bfloat16(v[2 * i]), hi = __float2bfloat16(v[2 * i + 1]);
            w[i] = (uint32_t)(*reinterpret_cast<const uint16_t *>(&lo)) |
                   ((uint32_t)(*reinterpret_cast<const uint16_t *>(&hi)) << 16);
        }
        *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <> struct Store<f16_t> {
    static constexpr int VEC = 8;
    __device__ static float get(const f16_t *p) { return __half2float(*p); }
    __device__ static void put(f16_t *p, float v) { *p = __float2half(v); }
    __device__ static void load(const f16_t *p, float (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4 *>(p);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __half2 h = *reinterpret_cast<const __half2 *>(&w[i]);
            const float2 f = __half22float2(h);
            v[2 * i] = f.x;
            v[2 * i + 1] = f.y;
        }
    }
    __device__ static void store(f16_t *p, const float (&v)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __half2 h = __floats2half2_rn(v[2 * i], v[2 * i + 1]);
            w[i] = *reinterpret_cast<const uint32_t *>(&h);
        }
        *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <> struct Store<double> {
    static constexpr int VEC = 2;
    __device__ static double get(const double *p) { return *p; }
    __device__ static void put(double *p, double v) { *p = v; }
};

// Storage policy of the slab kernels: 4 channels per lane for every dtype (8-byte lanes for the 16-bit
// types), so that a wave is 8 rows x 8 lanes at D = 32 whatever the dtype: the per-wave record area stays
// 4.4 KiB, 16 waves fit beside the slab, and a 16-bit gather instruction touches 8 granules, not 16.
template <typename T> struct SlabStore : Store<T> {};
template <> struct SlabStore<bf16_t> {
    static constexpr int VEC = 4;
    __device__ static float get(const bf16_t *p) { return __bfloat162float(*p); }
    __device__ static void put(bf16_t *p, float v) { *p = __float2bfloat16(v); }
    __device__ static void load(const bf16_t *p, float (&v)[4]) {
        const uint2 t = *reinterpret_cast<const uint2 *>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    }
    __device__ static void store(bf16_t *p, const float (&v)[4]) {
        uint32_t w[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16_t lo = __float2bfloat16(v[2 * i]), hi = __float2bfloat16(v[2 * i + 1]);
            w[i] = (uint32_t)(*reinterpret_cast<const uint16_t *>(&lo)) |
                   ((uint32_t)(*reinterpret_cast<const uint16_t *>(&hi)) << 16);
        }
        *reinterpret_cast<uint2 *>(p) = make_uint2(w[0], w[1]);
    }
};
template <> struct SlabStore<f16_t> {
    static constexpr int VEC = 4;
    __device__ static float get(const f16_t *p) { return __half2float(*p); }
    __device__ static void put(f16_t *p, float v) { *p = __float2half(v); }
    __device__ static void load(const f16_t *p, float (&v)[4]) {
        const uint2 t = *reinterpret_cast<const uint2 *>(p);
        const float2 a = __half22float2(*reinterpret_cast<const __half2 *>(&t.x));
        const float2 b = __half22float2(*reinterpret_cast<const __half2 *>(&t.y));
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    }
    __device__ static void store(f16_t *p, const float (&v)[4]) {
        const __half2 a = __floats2half2_rn(v[0], v[1]), b = __floats2half2_rn(v[2], v[3]);
        *reinterpret_cast<uint2 *>(p) = make_uint2(*reinterpret_cast<const uint32_t *>(&a), *reinterpret_cast<const uint32_t *>(&b));
    }
};

// Gather load of one 16-/8-byte lane vector at  base + (element offset from the tap record) + (this lane's
// byte offset):  `base` is wave-uniform (it lives in SGPRs), so the access is the saddr form with ONE 32-bit
// VGPR offset -- one v_lshl_add_u32 per load instead of a sign extension and a 64-bit add per load.
template <typename S, typename T, int N>
__device__ __forceinline__ void gather_load(const T *base, int elem_off, unsigned lane_bytes, float (&v)[N])
{
    const unsigned off = ((unsigned)elem_off * (unsigned)sizeof(T)) + lane_bytes;
    S::load(reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + off), v);
}

// The forward flavour: the same access as a BUFFER load whose resource ends with the clip (`num_records` = the bytes
// one (clip, head) base can reach).  Corners outside the map carry the offset kOobBytes, beyond every resource, and
// read as zeros without touching memory -- not as pixel 0 times weight 0, which would turn a non-finite value at
// an unrelated pixel into NaN (the reference does not read such corners at all, cuh:56-80).  The backward kernels
// mask the dots of such corners by their validity bits instead and keep the plain loads.
constexpr unsigned kOobBytes = 0xF0000000u;
template <typename T> constexpr int oob_elems() { return (int)(kOobBytes / sizeof(T)); }
template <typename S, typename T, int N>
__device__ __forceinline__ void gather_load_z(__amdgpu_buffer_rsrc_t rsrc, int elem_off, unsigned lane_bytes, float (&v)[N])
{
    const unsigned off = ((unsigned)elem_off * (unsigned)sizeof(T)) + lane_bytes;
    if constexpr (N * sizeof(T) == 16) unpack_raw(static_cast<const T *>(nullptr), __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0), v);
    else unpack_raw(static_cast<const T *>(nullptr), __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0), v);
}
// resource over everything `vbase` (a (clip, head) base) can reach inside its clip
template <typename T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t clip_resource(const T *vbase, int64_t pixels, int v_pix, int D)
{
    const int64_t bytes = ((pixels - 1) * v_pix + D) * (int64_t)sizeof(T);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(vbase), 0, (int)bytes, 0x00020000);
}

// LDS byte address of a pointer into shared memory
__device__ __forceinline__ unsigned lds_addr(const void *q)
{
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)q;
}

// hardware float atomics (global_atomic_add_f32 / _f64, no return value, no CAS loop)
__device__ __forceinline__ void atomic_accumulate(float *p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_accumulate(double *p, double v) { unsafeAtomicAdd(p, v); }

// ------------------------------------------------------------------------------------------------
// kernel parameters: one struct serves the plain op (frames = 1, window = 0, LB = 0) and the fused
// temporal op (array A = current-frame points, array B = temporal points)
// ------------------------------------------------------------------------------------------------
struct Params {
    const void *value;          // [groups, S, M, D]; groups = clips * frames
    const int64_t *shapes;      // [L, 2] (H, W)
    const int64_t *lsi;         // [L]
    const int32_t *ftab;        // [frames, window] or null
    const void *locA, *awA;     // [groups, Lq, M, LA, PA, 2], [groups, Lq, M, LA, PA]
    const void *locB, *awB;     // [groups, Lq, M, LB, PB, 2], ...      (LB = window * L)
    void *out;                  // fwd: [groups, Lq, M*D]
    const void *grad_out;       // bwd
    void *grad_value;           // bwd: acc type, pre-zeroed
    void *glocA, *gawA, *glocB, *gawB;
    unsigned *workspace;        // bwd: 8 work-ticket counters of the scatter pass (zeroed by the caller) or null
    int *bbox;                  // bwd: [groups, M, LA+LB, Lq, 2] (min, max) top tap row per (row, virtual level) -- or, with
                                // cull_points, the top tap row of each of its <= 4 points as int16 (same 8 bytes) --
                                // written by the gather pass, read by the scatter pass to cull rows; or null
    int groups, frames, window;
    int S, M, D, L, Lq;
    int LA, PA, LB, PB;
    int64_t v_clip, v_head;     // element strides of `value`: between clips (= frames * S pixels), between heads
    int v_pix;                  // ... and between consecutive pixels (standard [S, M, D]: frames*S*M*D, D, M*D)
    int *bsum;                  // bwd: [groups, M, LA+LB, ceil(Lq/64), 2] (min, max) top tap row over blocks of 64 queries
                                // (built from the per-point records; lets long candidate ranges skip dead blocks) or null
    const int64_t *shapes_host; // HOST copy of `shapes` or null: kernel selection only (never dereferenced on the device)
    int cull_points;            // bbox entries are 4 x int16 top tap rows, one per POINT (PA, PB <= 4), not (min, max)
    int wide_stores;            // bwd: the four gradient arrays are 16-byte aligned (resident-slab gather pass: whole-row stores)
    int wide_loads;             // loc / attn arrays are 16-byte aligned (resident-slab kernels: whole-row loads)
    int dbg;                    // measurement hooks (MSDA_DBG env), 0 in production
};

struct Level { int H, W, start, pad; };   // start = first pixel of the level inside the CLIP slab

// virtual level j of the wave's (clip, frame t):  j < LA -> level j of frame t (plain op: of the only
// map); j >= LA -> temporal slot w = (j-LA)/L, level (j-LA)%L of frame ftab[t, w].
__device__ __forceinline__ Level make_level(const Params &p, int t, int j)
{
    int l = j, f = t;
    if (j >= p.LA) {
        const int w = (j - p.LA) / p.L;
        l = (j - p.LA) - w * p.L;
        f = p.ftab[t * p.window + w];
        f = min(max(f, 0), p.frames - 1);   // memory safety only; valid tables never clamp
    }
    Level lv;
    lv.H = (int)p.shapes[2 * l];
    lv.W = (int)p.shapes[2 * l + 1];
    lv.start = f * p.S + (int)p.lsi[l];
    lv.pad = 0;
    return lv;
}

// One sampling point -> tap record.  Follows cuh:285-288 (pixel coords, range test), cuh:38-53
// (floor, fractions, strides) and cuh:56-80 (per-corner validity, weights).
struct Taps {
    int off[4];        // element offsets (inside the clip slab, without m*D + c) of the 4 corners
    float w[4];        // hh*hw, hh*lw, lh*hw, lh*lw  -- zero for corners outside the map
    float lh, lw;
    int valid;         // bit k set = corner k inside the map; 0 = point skipped
    int hl;            // floor(h_im): top tap row (-1 .. H-1), meaningful when valid != 0
};

__device__ __forceinline__ Taps make_taps(float x, float y, const Level lv, int MD, int oob = 0)
{
    Taps t;
    t.off[0] = t.off[1] = t.off[2] = t.off[3] = oob;      // corners outside the map (forward: reads as zeros)
    t.w[0] = t.w[1] = t.w[2] = t.w[3] = 0.f;
    t.lh = t.lw = 0.f;
    t.valid = 0;
    t.hl = 0;
    // rounded multiply THEN subtract (no FMA contraction): the reference evaluates
    // `loc * spatial - 0.5` with a float product (cuh:285-286), and which pixel cell a point falls
    // in must not depend on the compiler's contraction choices.
    const float h_im = __fsub_rn(__fmul_rn(y, (float)lv.H), 0.5f);
    const float w_im = __fsub_rn(__fmul_rn(x, (float)lv.W), 0.5f);
    if (h_im > -1.f && w_im > -1.f && h_im < (float)lv.H && w_im < (float)lv.W) {
        const float hf = floorf(h_im), wf = floorf(w_im);
        const int h_low = (int)hf, w_low = (int)wf;
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - hf, lw = w_im - wf;
        const float hh = 1.f - lh, hw = 1.f - lw;
        const bool y0 = h_low >= 0, y1 = h_high <= lv.H - 1;
        const bool x0 = w_low >= 0, x1 = w_high <= lv.W - 1;
        const int r0 = (lv.start + h_low * lv.W) * MD, r1 = r0 + lv.W * MD;
        const int c0 = w_low * MD, c1 = c0 + MD;
        t.lh = lh; t.lw = lw; t.hl = h_low;
        if (y0 && x0) { t.off[0] = r0 + c0; t.w[0] = hh * hw; t.valid |= 1; }
        if (y0 && x1) { t.off[1] = r0 + c1; t.w[1] = hh * lw; t.valid |= 2; }
        if (y1 && x0) { t.off[2] = r1 + c0; t.w[2] = lh * hw; t.valid |= 4; }
        if (y1 && x1) { t.off[3] = r1 + c1; t.w[3] = lh * lw; t.valid |= 8; }
    }
    return t;
}

// ------------------------------------------------------------------------------------------------
// tile kernels
// ------------------------------------------------------------------------------------------------
// LDS carve (dynamic, 16-byte aligned): [s_off RPW*(kPch+1) int4][s_w RPW*(kPch+1) float4]
//                                       [s_e RPW*(kPch+1) float4 (bwd only)][levels nvl * Level]
// Row stride kPch+1 (odd number of 16-B slots) keeps the RPW rows of a wave on different LDS slots
// for the broadcast ds_read_b128 of the gather loop.
constexpr int kRowSlots = kPch + 1;

template <int RPW>
__device__ __forceinline__ void tile_coords(const Params &p, int &m, int &group, int &q0)
{
    // head = blockIdx % M -> XCD affinity (see file header); tiles of one group are consecutive
    m = blockIdx.x % p.M;
    const int tile = blockIdx.x / p.M;
    if (p.dbg & 32) m = (m + tile) % p.M;      // measurement: break the head <-> XCD affinity
    const int tiles_per_group = (p.Lq + RPW - 1) / RPW;
    group = tile / tiles_per_group;
    q0 = (tile - group * tiles_per_group) * RPW;
}

// The wave walks its rows' sampling points in chunks of kPch: first the chunks of array A (current
// frame / plain op), then those of array B (temporal points).
template <typename T> struct ChunkRef {
    const T *loc, *aw;
    int LP, P, vl_base, p0, arr;
};

template <typename T>
__device__ __forceinline__ ChunkRef<T> get_chunk(const Params &p, int c, int nA)
{
    ChunkRef<T> r;
    r.arr = (c >= nA);
    r.loc = static_cast<const T *>(r.arr ? p.locB : p.locA);
    r.aw = static_cast<const T *>(r.arr ? p.awB : p.awA);
    r.P = r.arr ? p.PB : p.PA;
    r.LP = (r.arr ? p.LB : p.LA) * r.P;
    r.vl_base = r.arr ? p.LA : 0;
    r.p0 = (r.arr ? c - nA : c) * kPch;
    return r;
}

__device__ __forceinline__ int n_chunks(int levels, int points) { return (levels * points + kPch - 1) / kPch; }

// (x, y, weight) of the points this lane stages for one chunk: RPW*kPch/64 points per lane, all loads
// issued before any tap arithmetic.
template <int NPL> struct Staged { float x[NPL], y[NPL], a[NPL]; };
// points a lane stages per chunk (wide rows, G >= 32, leave some lanes without a point)
template <int RPW> constexpr int staged_per_lane() { return (RPW * kPch + kWave - 1) / kWave; }

template <typename T, int RPW>
__device__ __forceinline__ void load_chunk(const Params &p, const ChunkRef<T> &c, int64_t row0,
                                           int rows_valid, int lane, Staged<staged_per_lane<RPW>()> &st)
{
#pragma unroll
    for (int k = 0; k < staged_per_lane<RPW>(); ++k) {
        const int i = lane + k * kWave;
        const int rr = i / kPch, pt = c.p0 + i % kPch;
        st.x[k] = st.y[k] = -10.f;      // far outside every map: yields an all-zero tap record
        st.a[k] = 0.f;
        if (rr < rows_valid && pt < c.LP) {
            const int64_t idx = (row0 + (int64_t)rr * p.M) * c.LP + pt;
            st.x[k] = Store<T>::get(c.loc + 2 * idx);
            st.y[k] = Store<T>::get(c.loc + 2 * idx + 1);
            st.a[k] = Store<T>::get(c.aw + idx);
        }
    }
}

// Culling record of one sampling point inside its (row, level) entry of the interval table: either widens
// the (min, max) interval (ds_min/ds_max_i32) or, in point mode, stores the point's own top tap row as
// int16 (rows beyond 32767 saturate: the scatter's test saturates the same way, so it stays conservative).
constexpr int kNoRow16 = -32768;
__device__ __forceinline__ void note_tap_row(int *entry, bool points, int pt, int valid, int hl)
{
    if (points) reinterpret_cast<short *>(entry)[pt] = valid ? (short)min(hl, 32767) : (short)kNoRow16;
    else if (valid) { atomicMin(entry, hl); atomicMax(entry + 1, hl); }
}
__device__ __forceinline__ void init_tap_rows(int *entry, bool points)
{
    entry[0] = points ? (int)0x80008000u : 0x7fffffff;
    entry[1] = points ? (int)0x80008000u : -0x7fffffff - 1;
}

// Builds the tap records of one chunk (<= kPch points of every row of the wave) in LDS.
template <typename T, int RPW, bool BWD>
__device__ __forceinline__ void build_chunk(const Params &p, const ChunkRef<T> &c,
                                            const Staged<staged_per_lane<RPW>()> &st, const Level *s_lvl,
                                            int4 *s_off, float4 *s_w, float4 *s_e, int lane,
                                            int *s_bb = nullptr, int nvl = 0)
{
    const int MD = p.v_pix;      // pixel stride of `value`
#pragma unroll
    for (int k = 0; k < staged_per_lane<RPW>(); ++k) {
        const int i = lane + k * kWave;
        if (i >= RPW * kPch) break;
        const int rr = i / kPch, pp = i % kPch;
        const int vl = c.vl_base + min(c.p0 + pp, c.LP - 1) / c.P;
        const float a = st.a[k];
        const Taps t = make_taps(st.x[k], st.y[k], s_lvl[vl], MD, BWD ? 0 : oob_elems<T>());
        s_off[rr * kRowSlots + pp] = make_int4(t.off[0], t.off[1], t.off[2], t.off[3]);
        if (BWD) {
            s_w[rr * kRowSlots + pp] = make_float4(t.w[0], t.w[1], t.w[2], t.w[3]);
            // a, fractions, and (valid bits | level index << 4) for the final gradient lane
            s_e[rr * kRowSlots + pp] = make_float4(a, t.lh, t.lw, __int_as_float(t.valid | (vl << 4)));
            if (s_bb) {
                const int kk = min(c.p0 + pp, c.LP - 1);
                if (c.p0 + pp < c.LP)
                    note_tap_row(s_bb + (rr * nvl + vl) * 2, p.cull_points != 0, kk - (kk / c.P) * c.P, t.valid, t.hl);
            }
        } else {
            s_w[rr * kRowSlots + pp] = make_float4(t.w[0] * a, t.w[1] * a, t.w[2] * a, t.w[3] * a);
        }
    }
}

// NB = sampling points whose 4*NB corner loads are issued back to back before any FMA consumes them
// (memory-level parallelism per wave); more points in flight cost VGPRs, i.e. waves per SIMD.
template <typename T, int G, int NB>
__global__ void __launch_bounds__(kWave, (NB <= 2 && kPch / G <= 2) ? 8 : 4)
msda_fwd_tile_kernel(const Params p)
{
    constexpr int VEC = Store<T>::VEC;
    constexpr int RPW = kWave / G;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    int4 *s_off = reinterpret_cast<int4 *>(lds_raw);
    float4 *s_w = reinterpret_cast<float4 *>(s_off + RPW * kRowSlots);
    Level *s_lvl = reinterpret_cast<Level *>(s_w + RPW * kRowSlots);

    const int lane = threadIdx.x;
    int m, group, q0;
    tile_coords<RPW>(p, m, group, q0);
    const int clip = group / p.frames, t = group - clip * p.frames;
    const int nvl = p.LA + p.LB;
    for (int j = lane; j < nvl; j += kWave) s_lvl[j] = make_level(p, t, j);
    __syncthreads();

    const int r = lane / G, sub = lane % G;
    const int rows_valid = min(RPW, p.Lq - q0);
    const int MD = p.M * p.D;
    const T *__restrict__ vbase = static_cast<const T *>(p.value) + clip * p.v_clip + m * p.v_head;   // wave-uniform
    const __amdgpu_buffer_rsrc_t rsrc = clip_resource(vbase, (int64_t)p.frames * p.S, p.v_pix, p.D);
    const unsigned lane_bytes = (unsigned)(sub * VEC * (int)sizeof(T));
    const int64_t row0 = ((int64_t)group * p.Lq + q0) * p.M + m;   // row of rr = 0; next row: + M

    float acc[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = 0.f;

    const int nA = n_chunks(p.LA, p.PA), n_all = nA + n_chunks(p.LB, p.PB);
    // No cross-chunk prefetch on purpose: vector-memory loads return in order, so an HBM-latency load of
    // the next chunk's (x, y, weight) issued ahead of the gathers only makes every gather wait for it
    // (measured: 0.73 -> 0.82 ms); the other waves of the SIMD cover the stage phase instead.
    Staged<staged_per_lane<RPW>()> st;
#pragma unroll 1
    for (int ci = 0; ci < n_all; ++ci) {
        const ChunkRef<T> c = get_chunk<T>(p, ci, nA);
        load_chunk<T, RPW>(p, c, row0, rows_valid, lane, st);
        build_chunk<T, RPW, false>(p, c, st, s_lvl, s_off, s_w, nullptr, lane);
        __syncthreads();
        const int np = min(kPch, c.LP - c.p0);
        const int4 *ro = s_off + r * kRowSlots;
        const float4 *rw = s_w + r * kRowSlots;
        // slots np..kPch-1 hold zero-weight records (offset 0), so a batch may run past np
#pragma unroll 1
        for (int pp = 0; pp < np; pp += NB) {
            int4 o[NB];
            float4 w[NB];
            float v[NB][4][VEC];
#pragma unroll
            for (int b = 0; b < NB; ++b) { o[b] = ro[pp + b]; w[b] = rw[pp + b]; }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                gather_load_z<Store<T>, T>(rsrc, o[b].x, lane_bytes, v[b][0]);
                gather_load_z<Store<T>, T>(rsrc, o[b].y, lane_bytes, v[b][1]);
                gather_load_z<Store<T>, T>(rsrc, o[b].z, lane_bytes, v[b][2]);
                gather_load_z<Store<T>, T>(rsrc, o[b].w, lane_bytes, v[b][3]);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int ch = 0; ch < VEC; ++ch) {
                    acc[ch] = fmaf(w[b].x, v[b][0][ch], acc[ch]);
                    acc[ch] = fmaf(w[b].y, v[b][1][ch], acc[ch]);
                    acc[ch] = fmaf(w[b].z, v[b][2][ch], acc[ch]);
                    acc[ch] = fmaf(w[b].w, v[b][3][ch], acc[ch]);
                }
            }
        }
        __syncthreads();
    }
    if (r < rows_valid) {
        T *out = static_cast<T *>(p.out) + (row0 + (int64_t)r * p.M) * p.D + sub * VEC;
        Store<T>::store(out, acc);
    }
}

// ------------------------------------------------------------------------------------------------
// forward with the small pyramid levels served from a workgroup-shared LDS slab
// ------------------------------------------------------------------------------------------------
// The tile forward is bound by the rate at which a CU's L1 serves distinct 64-byte granules (DESIGN.md
// section 5): ~25-29 B/clk/CU for scattered 128-byte rows, whatever the locality.  Random 16-byte-per-lane
// LDS reads run at ~94 B/clk/CU.  So 16 waves share one workgroup: each wave still owns a tile of RPW rows
// and builds its tap records once per point in its own LDS region exactly as the tile kernel does, but
// the workgroup walks the clip's SOURCE FRAMES together and, for each, stages the slab
// value[frame, levels >= l0, head, :] (as many of the last levels as fit; levels 2-3 of the DeVIS pyramid =
// 50 % of all taps, 38 KiB) into LDS with LDS-DMA.  A tap whose level is in the slab reads LDS, the
// others go through L1 as before; the level of chunk position pp is the same for every row, so the choice
// is wave-uniform.  Accumulators stay in registers across the frames.
constexpr int kSlabWaves = 16;
constexpr int kSlabThreads = kSlabWaves * kWave;
constexpr int kSlabMaxLevels = 32;

// single-wave producer/consumer hand-off through LDS: DS operations of one wave execute in order, only the
// compiler must be kept from moving them across
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// levels [result, L) form the slab: the last levels whose pixels are one contiguous tail of the map and
// whose [pixels, D] slab fits `cap` elements
__device__ __forceinline__ int first_slab_level(const Params &p, int cap)
{
    int l0 = p.L;
    long long acc = 0;
    for (int l = p.L - 1; l >= 0; --l) {
        const long long hw = (long long)p.shapes[2 * l] * p.shapes[2 * l + 1];
        if (l + 1 < p.L && p.lsi[l] + hw != p.lsi[l + 1]) break;
        acc += hw * p.D;
        if (acc > cap) break;
        l0 = l;
    }
    return l0;
}

template <typename T, int G, int NB>
__global__ void __launch_bounds__(kSlabThreads)
msda_fwd_slab_kernel(const Params p, int slab_elems)
{
    constexpr int VEC = SlabStore<T>::VEC;
    constexpr int RPW = kWave / G;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int s_sH[kSlabMaxLevels], s_sW[kSlabMaxLevels], s_sStart[kSlabMaxLevels];
    __shared__ int s_l0, s_px0, s_npx;

    const int tid = threadIdx.x, wave = tid / kWave, lane = tid % kWave;
    const int nvl = p.LA + p.LB, L = p.L;
    T *slab = reinterpret_cast<T *>(lds_raw);
    unsigned char *mine = lds_raw + (size_t)slab_elems * sizeof(T) +
                          (size_t)wave * (RPW * kRowSlots * 32 + nvl * sizeof(Level));
    int4 *s_off = reinterpret_cast<int4 *>(mine);
    float4 *s_w = reinterpret_cast<float4 *>(s_off + RPW * kRowSlots);
    Level *s_lvl = reinterpret_cast<Level *>(s_w + RPW * kRowSlots);

    if (tid == 0) {
        // slack: last LDS-DMA piece; one more row: slab row 0 is the ZERO ROW corners outside the map read
        const int l0 = first_slab_level(p, slab_elems - 2048 / (int)sizeof(T) - p.D);
        const int px0 = l0 < L ? (int)p.lsi[l0] : 0;
        int npx = 0;
        for (int l = l0; l < L; ++l) {
            s_sH[l] = (int)p.shapes[2 * l]; s_sW[l] = (int)p.shapes[2 * l + 1];
            s_sStart[l] = (int)p.lsi[l] - px0 + 1;                 // + the zero row
            npx += s_sH[l] * s_sW[l];
        }
        s_l0 = l0; s_px0 = px0; s_npx = npx;
    }

    // workgroup -> (clip, head, 16 consecutive tiles of the clip); wave -> tile
    const int m = blockIdx.x % p.M;
    const int tiles_per_group = (p.Lq + RPW - 1) / RPW;
    const int tiles_per_clip = p.frames * tiles_per_group;
    const int blocks_per_clip = (tiles_per_clip + kSlabWaves - 1) / kSlabWaves;
    const int rest = blockIdx.x / p.M;
    const int clip = rest / blocks_per_clip;
    const int ct = (rest - clip * blocks_per_clip) * kSlabWaves + wave;
    const bool have_tile = ct < tiles_per_clip;
    const int t = have_tile ? ct / tiles_per_group : 0;
    const int q0 = have_tile ? (ct - t * tiles_per_group) * RPW : 0;
    const int group = clip * p.frames + t;
    if (have_tile)
        for (int j = lane; j < nvl; j += kWave) s_lvl[j] = make_level(p, t, j);
    __syncthreads();
    const int l0 = s_l0, px0 = s_px0, npx = s_npx;

    const int r = lane / G, sub = lane % G;
    const int rows_valid = have_tile ? min(RPW, p.Lq - q0) : 0;
    const int D = p.D, MD = p.M * p.D;
    const T *__restrict__ vbase = static_cast<const T *>(p.value) + clip * p.v_clip + m * p.v_head;   // wave-uniform
    const unsigned lane_bytes = (unsigned)(sub * VEC * (int)sizeof(T));
    const T *slab_lane = slab + sub * VEC;
    const __amdgpu_buffer_rsrc_t rsrc = clip_resource(vbase, (int64_t)p.frames * p.S, p.v_pix, p.D);
    // the zero row (never overwritten by the staging): a corner outside the map carries slab offset 0 / a buffer
    // offset beyond the clip and reads zeros either way (see gather_load_z)
    for (int i = tid; i < D; i += kSlabThreads) SlabStore<T>::put(slab + i, 0.f);
    const int64_t row0 = ((int64_t)group * p.Lq + q0) * p.M + m;

    float acc[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = 0.f;

    for (int f = 0; f < p.frames; ++f) {
        __syncthreads();                                   // every wave is done with the previous slab
        if (l0 < L) {
            constexpr int GS = G * VEC * (int)sizeof(T) >= 16 ? G * VEC * (int)sizeof(T) / 16 : 1;      // 16-byte DMA lanes per pixel
            constexpr int PXW = kWave / GS;                 // pixels per LDS-DMA wave instruction
            const T *src = static_cast<const T *>(p.value) + clip * p.v_clip + m * p.v_head +
                           ((int64_t)f * p.S + px0) * p.v_pix;
            for (int pb = wave * PXW; pb < npx; pb += kSlabWaves * PXW) {
                const int px = min(pb + lane / GS, npx - 1);
                const T *gp = src + (int64_t)px * p.v_pix + (lane % GS) * (16 / (int)sizeof(T));
#if defined(__HIP_DEVICE_COMPILE__)      // device-only builtin: keep the host pass (kernel stub) clean
                __builtin_amdgcn_global_load_lds(
                    gp, (__attribute__((address_space(3))) void *)(slab + (size_t)(pb + 1) * D), 16, 0, 0);
#else
                (void)gp;
#endif
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (!have_tile) continue;
        for (int sl = -1; sl < p.window; ++sl) {           // sl = -1: the tile's current-frame points
            if (sl < 0) { if (t != f) continue; }
            else if (p.ftab[t * p.window + sl] != f) continue;
            const T *loc = static_cast<const T *>(sl < 0 ? p.locA : p.locB);
            const T *aw = static_cast<const T *>(sl < 0 ? p.awA : p.awB);
            const int P = sl < 0 ? p.PA : p.PB;
            const int LP = (sl < 0 ? p.LA : p.LB) * P;
            const int nlev = sl < 0 ? p.LA : L;
            const int pt0 = sl < 0 ? 0 : sl * L * P;
            const int vl0 = sl < 0 ? 0 : p.LA + sl * L;    // virtual level of the slot's level 0
            const int npts = nlev * P;
            const int first_slab_pt = l0 < nlev ? l0 * P : 0x7fffffff;     // points of levels >= l0 read the slab
#pragma unroll 1
            for (int c0 = 0; c0 < npts; c0 += kPch) {
                // ---- stage: tap records of this chunk (one point per lane and step), LDS or global flavour
#pragma unroll
                for (int k = 0; k < staged_per_lane<RPW>(); ++k) {
                    const int i = lane + k * kWave;
                    if (i >= RPW * kPch) break;
                    const int rr = i / kPch, pp = i % kPch, kk = c0 + pp;
                    float x = -10.f, y = -10.f, a = 0.f;
                    if (rr < rows_valid && kk < npts) {
                        const int64_t idx = (row0 + (int64_t)rr * p.M) * LP + pt0 + kk;
                        x = SlabStore<T>::get(loc + 2 * idx);
                        y = SlabStore<T>::get(loc + 2 * idx + 1);
                        a = SlabStore<T>::get(aw + idx);
                    }
                    const int l = min(kk, npts - 1) / P;
                    Taps tp;
                    if (l >= l0) {
                        Level lv; lv.H = s_sH[l]; lv.W = s_sW[l]; lv.start = s_sStart[l]; lv.pad = 0;
                        tp = make_taps(x, y, lv, D);               // offsets inside the slab (stride D)
                    } else {
                        tp = make_taps(x, y, s_lvl[vl0 + l], p.v_pix, oob_elems<T>());
                    }
                    s_off[rr * kRowSlots + pp] = make_int4(tp.off[0], tp.off[1], tp.off[2], tp.off[3]);
                    s_w[rr * kRowSlots + pp] = make_float4(tp.w[0] * a, tp.w[1] * a, tp.w[2] * a, tp.w[3] * a);
                }
                wave_sync();
                // ---- gather
                const int np = min(kPch, npts - c0);
                const int4 *ro = s_off + r * kRowSlots;
                const float4 *rw = s_w + r * kRowSlots;
#pragma unroll 1
                for (int pp = 0; pp < np; pp += NB) {
                    int4 o[NB];
                    float4 w[NB];
                    float v[NB][4][VEC];
#pragma unroll
                    for (int b = 0; b < NB; ++b) { o[b] = ro[pp + b]; w[b] = rw[pp + b]; }
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const bool in_slab = c0 + pp + b >= first_slab_pt;             // wave-uniform
                        if (in_slab) {
                            SlabStore<T>::load(slab_lane + o[b].x, v[b][0]);
                            SlabStore<T>::load(slab_lane + o[b].y, v[b][1]);
                            SlabStore<T>::load(slab_lane + o[b].z, v[b][2]);
                            SlabStore<T>::load(slab_lane + o[b].w, v[b][3]);
                        } else {
                            gather_load_z<SlabStore<T>, T>(rsrc, o[b].x, lane_bytes, v[b][0]);
                            gather_load_z<SlabStore<T>, T>(rsrc, o[b].y, lane_bytes, v[b][1]);
                            gather_load_z<SlabStore<T>, T>(rsrc, o[b].z, lane_bytes, v[b][2]);
                            gather_load_z<SlabStore<T>, T>(rsrc, o[b].w, lane_bytes, v[b][3]);
                        }
                    }
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
#pragma unroll
                        for (int ch = 0; ch < VEC; ++ch) {
                            acc[ch] = fmaf(w[b].x, v[b][0][ch], acc[ch]);
                            acc[ch] = fmaf(w[b].y, v[b][1][ch], acc[ch]);
                            acc[ch] = fmaf(w[b].z, v[b][2][ch], acc[ch]);
                            acc[ch] = fmaf(w[b].w, v[b][3][ch], acc[ch]);
                        }
                    }
                }
                wave_sync();
            }
        }
    }
    if (r < rows_valid) {
        T *out = static_cast<T *>(p.out) + (row0 + (int64_t)r * p.M) * D + sub * VEC;
        SlabStore<T>::store(out, acc);
    }
}

// sum over the G lanes of a row (G a power of two <= 64; rows are G-aligned lane groups).  Up to 16
// lanes the butterfly is pure DPP (no LDS crossbar, no waits): quad_perm xor1 / xor2, row_half_mirror
// (lane i <-> 7-i inside each 8), row_mirror (i <-> 15-i inside each 16); wider rows finish with
// shuffles.  Every lane of the row ends up with the total.
// <g, v> over a lane's VEC channels with separate even / odd partial sums: the pairs (g[2i], g[2i+1]) and
// (v[2i], v[2i+1]) sit in adjacent registers, so the compiler emits v_pk_fma_f32 without operand shuffles
// (the straightforward four-dots-at-once loop costs one v_mov per packed FMA).
typedef float float2v __attribute__((ext_vector_type(2)));
template <int N>
__device__ __forceinline__ float dot_eo(const float (&g)[N], const float (&v)[N])
{
    float2v acc = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c + 1 < N; c += 2) {
        const float2v gp = {g[c], g[c + 1]}, vp = {v[c], v[c + 1]};
        acc = __builtin_elementwise_fma(gp, vp, acc);          // v_pk_fma_f32 on adjacent registers
    }
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(acc.x), "v"(acc.y));     // (kept scalar: no re-packing with v_movs)
    if (N & 1) r = fmaf(g[N - 1], v[N - 1], r);
    return r;
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)
{
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

template <int G>
__device__ __forceinline__ float row_sum(float v)
{
    if (G >= 2) v = dpp_add<0xB1>(v);     // quad_perm [1,0,3,2]
    if (G >= 4) v = dpp_add<0x4E>(v);     // quad_perm [2,3,0,1]
    if (G >= 8) v = dpp_add<0x141>(v);    // row_half_mirror
    if (G >= 16) v = dpp_add<0x140>(v);   // row_mirror
    if (G >= 32) v += __shfl_xor(v, 16, kWave);
    if (G >= 64) v += __shfl_xor(v, 32, kWave);
    return v;
}

// The four dots of a point reduced over the G lanes of the row at once: four independent DPP butterflies
// interleaved, each step ONE v_add_f32 with a DPP operand (the compiler's own lowering of row_sum is a
// v_mov_b32_dpp per value plus a packed add: 1.5 instructions per value and step, and s_nops between
// dependent steps; interleaving the four chains needs none).  G = 2, 4, 8, 16 only.
template <int G>
__device__ __forceinline__ void row_sum4(float &d0, float &d1, float &d2, float &d3)
{
    if constexpr (G == 2 || G == 4 || G == 8 || G == 16) {
#define MSDA_DPP4(ctrl)                                                                                   \
        "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                     \
        "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                     \
        "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"                     \
        "v_add_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        if constexpr (G == 2)
            asm volatile("s_nop 1\n" MSDA_DPP4("quad_perm:[1,0,3,2]") : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        else if constexpr (G == 4)
            asm volatile("s_nop 1\n" MSDA_DPP4("quad_perm:[1,0,3,2]") MSDA_DPP4("quad_perm:[2,3,0,1]")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        else if constexpr (G == 8)
            asm volatile("s_nop 1\n" MSDA_DPP4("quad_perm:[1,0,3,2]") MSDA_DPP4("quad_perm:[2,3,0,1]") MSDA_DPP4("row_half_mirror")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        else
            asm volatile("s_nop 1\n" MSDA_DPP4("quad_perm:[1,0,3,2]") MSDA_DPP4("quad_perm:[2,3,0,1]") MSDA_DPP4("row_half_mirror")
                         MSDA_DPP4("row_mirror") : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
#undef MSDA_DPP4
    } else {
        d0 = row_sum<G>(d0); d1 = row_sum<G>(d1); d2 = row_sum<G>(d2); d3 = row_sum<G>(d3);
    }
}

// Backward gather pass (grad_loc / grad_attn) with the same workgroup-shared slab of the small levels.
// To keep 16 waves per workgroup its tap-row interval table is per SLOT ([RPW, L], flushed after each
// slot) instead of per virtual level as in the tile kernel.
template <typename T, int G>
__global__ void __launch_bounds__(kSlabThreads)
msda_bwd_slab_kernel(const Params p, int slab_elems, int per_wave_bytes)
{
    constexpr int VEC = SlabStore<T>::VEC;
    constexpr int RPW = kWave / G;
    constexpr int NW = kSlabWaves;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int s_sH[kSlabMaxLevels], s_sW[kSlabMaxLevels], s_sStart[kSlabMaxLevels];
    __shared__ int s_l0, s_px0, s_npx;

    const int tid = threadIdx.x, wave = tid / kWave, lane = tid % kWave;
    const int nvl = p.LA + p.LB, L = p.L;
    // the scatter pass that follows on the stream draws its work tickets from the head of the workspace: zeroed here
    // (ABI v8), so that the caller does not have to launch a memset for 64 bytes
    if (blockIdx.x == 0 && tid < MSDA_BWD_WORKSPACE_BYTES / 4 && p.workspace) p.workspace[tid] = 0u;
    T *slab = reinterpret_cast<T *>(lds_raw);
    unsigned char *mine = lds_raw + (size_t)slab_elems * sizeof(T) + (size_t)wave * per_wave_bytes;
    int4 *s_off = reinterpret_cast<int4 *>(mine);
    float4 *s_w = reinterpret_cast<float4 *>(s_off + RPW * kRowSlots);
    float4 *s_e = s_w + RPW * kRowSlots;
    Level *s_lvl = reinterpret_cast<Level *>(s_e + RPW * kRowSlots);
    int *s_bb = p.bbox ? reinterpret_cast<int *>(s_lvl + nvl) : nullptr;      // [RPW, L, 2] of the current slot

    if (tid == 0) {
        const int l0 = first_slab_level(p, slab_elems - 2048 / (int)sizeof(T));
        const int px0 = l0 < L ? (int)p.lsi[l0] : 0;
        int npx = 0;
        for (int l = l0; l < L; ++l) {
            s_sH[l] = (int)p.shapes[2 * l]; s_sW[l] = (int)p.shapes[2 * l + 1];
            s_sStart[l] = (int)p.lsi[l] - px0;
            npx += s_sH[l] * s_sW[l];
        }
        s_l0 = l0; s_px0 = px0; s_npx = npx;
    }
    const int m = blockIdx.x % p.M;
    const int tiles_per_group = (p.Lq + RPW - 1) / RPW;
    const int tiles_per_clip = p.frames * tiles_per_group;
    const int blocks_per_clip = (tiles_per_clip + NW - 1) / NW;
    const int rest = blockIdx.x / p.M;
    const int clip = rest / blocks_per_clip;
    const int ct = (rest - clip * blocks_per_clip) * NW + wave;
    const bool have_tile = ct < tiles_per_clip;
    const int t = have_tile ? ct / tiles_per_group : 0;
    const int q0 = have_tile ? (ct - t * tiles_per_group) * RPW : 0;
    const int group = clip * p.frames + t;
    if (have_tile)
        for (int j = lane; j < nvl; j += kWave) s_lvl[j] = make_level(p, t, j);
    __syncthreads();
    const int l0 = s_l0, px0 = s_px0, npx = s_npx;

    const int r = lane / G, sub = lane % G;
    const int rows_valid = have_tile ? min(RPW, p.Lq - q0) : 0;
    const int D = p.D, MD = p.M * p.D;
    const T *__restrict__ vbase = static_cast<const T *>(p.value) + clip * p.v_clip + m * p.v_head;   // wave-uniform
    const unsigned lane_bytes = (unsigned)(sub * VEC * (int)sizeof(T));
    const T *slab_lane = slab + sub * VEC;
    const int64_t row0 = ((int64_t)group * p.Lq + q0) * p.M + m;
    const int64_t row = row0 + (int64_t)r * p.M;

    float g[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) g[c] = 0.f;
    if (r < rows_valid) SlabStore<T>::load(static_cast<const T *>(p.grad_out) + row * D + sub * VEC, g);

    for (int f = 0; f < p.frames; ++f) {
        __syncthreads();
        if (l0 < L) {
            constexpr int GS = G * VEC * (int)sizeof(T) >= 16 ? G * VEC * (int)sizeof(T) / 16 : 1;      // 16-byte DMA lanes per pixel
            constexpr int PXW = kWave / GS;
            const T *src = static_cast<const T *>(p.value) + clip * p.v_clip + m * p.v_head +
                           ((int64_t)f * p.S + px0) * p.v_pix;
            for (int pb = wave * PXW; pb < npx; pb += NW * PXW) {
                const int px = min(pb + lane / GS, npx - 1);
                const T *gp = src + (int64_t)px * p.v_pix + (lane % GS) * (16 / (int)sizeof(T));
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_global_load_lds(
                    gp, (__attribute__((address_space(3))) void *)(slab + (size_t)pb * D), 16, 0, 0);
#else
                (void)gp;
#endif
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (!have_tile) continue;
        for (int sl = -1; sl < p.window; ++sl) {
            if (sl < 0) { if (t != f) continue; }
            else if (p.ftab[t * p.window + sl] != f) continue;
            const T *loc = static_cast<const T *>(sl < 0 ? p.locA : p.locB);
            const T *aw = static_cast<const T *>(sl < 0 ? p.awA : p.awB);
            T *gloc = static_cast<T *>(sl < 0 ? p.glocA : p.glocB);
            T *gaw = static_cast<T *>(sl < 0 ? p.gawA : p.gawB);
            const int P = sl < 0 ? p.PA : p.PB;
            const int LP = (sl < 0 ? p.LA : p.LB) * P;
            const int nlev = sl < 0 ? p.LA : L;
            const int pt0 = sl < 0 ? 0 : sl * L * P;
            const int vl0 = sl < 0 ? 0 : p.LA + sl * L;
            const int npts = nlev * P;
            const int first_slab_pt = l0 < nlev ? l0 * P : 0x7fffffff;     // points of levels >= l0 read the slab
            if (s_bb) {
                for (int j = lane; j < RPW * nlev; j += kWave) init_tap_rows(s_bb + 2 * j, p.cull_points != 0);
                wave_sync();
            }
#pragma unroll 1
            for (int c0 = 0; c0 < npts; c0 += kPch) {
#pragma unroll
                for (int k = 0; k < staged_per_lane<RPW>(); ++k) {
                    const int i = lane + k * kWave;
                    if (i >= RPW * kPch) break;
                    const int rr = i / kPch, pp = i % kPch, kk = c0 + pp;
                    float x = -10.f, y = -10.f, a = 0.f;
                    if (rr < rows_valid && kk < npts) {
                        const int64_t idx = (row0 + (int64_t)rr * p.M) * LP + pt0 + kk;
                        x = SlabStore<T>::get(loc + 2 * idx);
                        y = SlabStore<T>::get(loc + 2 * idx + 1);
                        a = SlabStore<T>::get(aw + idx);
                    }
                    const int l = min(kk, npts - 1) / P, vl = vl0 + l;
                    Taps tp;
                    if (l >= l0) {
                        Level lv; lv.H = s_sH[l]; lv.W = s_sW[l]; lv.start = s_sStart[l]; lv.pad = 0;
                        tp = make_taps(x, y, lv, D);
                    } else {
                        tp = make_taps(x, y, s_lvl[vl], p.v_pix);
                    }
                    s_off[rr * kRowSlots + pp] = make_int4(tp.off[0], tp.off[1], tp.off[2], tp.off[3]);
                    s_w[rr * kRowSlots + pp] = make_float4(tp.w[0], tp.w[1], tp.w[2], tp.w[3]);
                    s_e[rr * kRowSlots + pp] = make_float4(a, tp.lh, tp.lw, __int_as_float(tp.valid | (vl << 4)));
                    if (s_bb && kk < npts)
                        note_tap_row(s_bb + (rr * nlev + l) * 2, p.cull_points != 0, kk - l * P, tp.valid, tp.hl);
                }
                wave_sync();
                const int np = min(kPch, npts - c0);
                const int4 *ro = s_off + r * kRowSlots;
                const float4 *rw = s_w + r * kRowSlots;
                const float4 *re = s_e + r * kRowSlots;
                // The four reduced dots of point pp are kept by lane pp % G of the row; after G points (or
                // at the chunk's end) every lane finishes ITS point at once -- cuh:123-158 rewritten on the
                // reduced dots -- instead of one lane in G finishing each point under an exec mask.  The
                // main loop then reads only the offsets record.
                float k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f;
#pragma unroll 2
                for (int pp = 0; pp < np; ++pp) {
                    const int4 o = ro[pp];
                    const bool in_slab = c0 + pp >= first_slab_pt;     // wave-uniform
                    float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
                    if (in_slab) {
                        SlabStore<T>::load(slab_lane + o.x, v0);
                        SlabStore<T>::load(slab_lane + o.y, v1);
                        SlabStore<T>::load(slab_lane + o.z, v2);
                        SlabStore<T>::load(slab_lane + o.w, v3);
                    } else {
                        gather_load<SlabStore<T>>(vbase, o.x, lane_bytes, v0);
                        gather_load<SlabStore<T>>(vbase, o.y, lane_bytes, v1);
                        gather_load<SlabStore<T>>(vbase, o.z, lane_bytes, v2);
                        gather_load<SlabStore<T>>(vbase, o.w, lane_bytes, v3);
                    }
                    float d0 = dot_eo(g, v0), d1 = dot_eo(g, v1), d2 = dot_eo(g, v2), d3 = dot_eo(g, v3);
                    row_sum4<G>(d0, d1, d2, d3);
                    const bool mine = sub == (pp & (G - 1));
                    k0 = mine ? d0 : k0; k1 = mine ? d1 : k1; k2 = mine ? d2 : k2; k3 = mine ? d3 : k3;
                    if ((pp & (G - 1)) == G - 1 || pp == np - 1) {      // wave-uniform
                        const int mp = (pp & ~(G - 1)) + sub;
                        if (mp <= pp) {
                            const float4 w = rw[mp];
                            const float4 e = re[mp];
                            const int bits = __float_as_int(e.w);
                            const float q0d = (bits & 1) ? k0 : 0.f, q1d = (bits & 2) ? k1 : 0.f;
                            const float q2d = (bits & 4) ? k2 : 0.f, q3d = (bits & 8) ? k3 : 0.f;
                            const float a = e.x, lh = e.y, lw = e.z, hh = 1.f - lh, hw = 1.f - lw;
                            const Level lv = s_lvl[bits >> 4];
                            const float g_aw = w.x * q0d + w.y * q1d + w.z * q2d + w.w * q3d;
                            const float g_w = hh * (q1d - q0d) + lh * (q3d - q2d);
                            const float g_h = hw * (q2d - q0d) + lw * (q3d - q1d);
                            s_e[r * kRowSlots + mp] = make_float4((float)lv.W * g_w * a, (float)lv.H * g_h * a, g_aw, 0.f);
                        }
                    }
                }
                wave_sync();
                if (r < rows_valid) {        // coalesced write-out, as in the tile kernel
                    const int64_t idx0 = row * LP + pt0 + c0;
                    const float *res = reinterpret_cast<const float *>(s_e + r * kRowSlots);
                    for (int el = sub; el < 2 * np; el += G)
                        SlabStore<T>::put(gloc + 2 * idx0 + el, res[(el >> 1) * 4 + (el & 1)]);
                    for (int el = sub; el < np; el += G)
                        SlabStore<T>::put(gaw + idx0 + el, res[el * 4 + 2]);
                }
                wave_sync();
            }
            if (s_bb) {      // this slot's tap-row intervals -> [group, head, virtual level, query]
                const int64_t gm = ((int64_t)group * p.M + m) * nvl + vl0;
                for (int i = lane; i < rows_valid * nlev; i += kWave) {
                    const int l = i / rows_valid, rr = i - l * rows_valid;
                    *reinterpret_cast<int2 *>(p.bbox + ((gm + l) * p.Lq + q0 + rr) * 2) =
                        make_int2(s_bb[(rr * nlev + l) * 2], s_bb[(rr * nlev + l) * 2 + 1]);
                }
                wave_sync();
            }
        }
    }
}

// ATOMICS = true : also scatters grad_value with global float atomics (one-kernel backward; used when
//                  the LDS scatter kernel below cannot take the shape).
// ATOMICS = false: computes grad_sampling_loc / grad_attn_weight only; grad_value comes from
//                  msda_bwd_value_lds_kernel.
template <typename T, int G, bool ATOMICS>
__global__ void __launch_bounds__(kWave)
msda_bwd_tile_kernel(const Params p)
{
    constexpr int VEC = Store<T>::VEC;
    constexpr int RPW = kWave / G;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    int4 *s_off = reinterpret_cast<int4 *>(lds_raw);
    float4 *s_w = reinterpret_cast<float4 *>(s_off + RPW * kRowSlots);
    float4 *s_e = s_w + RPW * kRowSlots;
    Level *s_lvl = reinterpret_cast<Level *>(s_e + RPW * kRowSlots);

    const int lane = threadIdx.x;
    if (blockIdx.x == 0 && lane < MSDA_BWD_WORKSPACE_BYTES / 4 && p.workspace) p.workspace[lane] = 0u;     // (see msda_bwd_slab_kernel)
    int m, group, q0;
    tile_coords<RPW>(p, m, group, q0);
    const int clip = group / p.frames, t = group - clip * p.frames;
    const int nvl = p.LA + p.LB;
    int *s_bb = p.bbox ? reinterpret_cast<int *>(s_lvl + nvl) : nullptr;      // [RPW, nvl, 2]
    for (int j = lane; j < nvl; j += kWave) s_lvl[j] = make_level(p, t, j);
    if (s_bb)
        for (int j = lane; j < RPW * nvl; j += kWave) init_tap_rows(s_bb + 2 * j, p.cull_points != 0);
    __syncthreads();

    const int r = lane / G, sub = lane % G;
    const int rows_valid = min(RPW, p.Lq - q0);
    const int MD = p.M * p.D;
    // (the ATOMICS variant scatters grad_value at value's offsets: the host only takes it for the standard layout)
    const int64_t lane_off = clip * p.v_clip + m * p.v_head + sub * VEC;
    const T *__restrict__ vbase = static_cast<const T *>(p.value) + clip * p.v_clip + m * p.v_head;   // wave-uniform
    const unsigned lane_bytes = (unsigned)(sub * VEC * (int)sizeof(T));
    float *__restrict__ gvalue = static_cast<float *>(p.grad_value) + lane_off;
    const int64_t row0 = ((int64_t)group * p.Lq + q0) * p.M + m;
    const int64_t row = row0 + (int64_t)r * p.M;

    float g[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) g[c] = 0.f;
    if (r < rows_valid) Store<T>::load(static_cast<const T *>(p.grad_out) + row * p.D + sub * VEC, g);
    const int nA = n_chunks(p.LA, p.PA), n_all = nA + n_chunks(p.LB, p.PB);
    Staged<staged_per_lane<RPW>()> st;
#pragma unroll 1
    for (int ci = 0; ci < n_all; ++ci) {
        {
            const ChunkRef<T> c = get_chunk<T>(p, ci, nA);
            load_chunk<T, RPW>(p, c, row0, rows_valid, lane, st);
            T *gloc = static_cast<T *>(c.arr ? p.glocB : p.glocA);
            T *gaw = static_cast<T *>(c.arr ? p.gawB : p.gawA);
            const int LP = c.LP, p0 = c.p0;
            build_chunk<T, RPW, true>(p, c, st, s_lvl, s_off, s_w, s_e, lane, s_bb, nvl);
            __syncthreads();
            const int np = min(kPch, LP - p0);
            const int4 *ro = s_off + r * kRowSlots;
            const float4 *rw = s_w + r * kRowSlots;
            const float4 *re = s_e + r * kRowSlots;
            // The four reduced dots of point pp are kept by lane pp % G of the row; after G points (or at
            // the chunk's end) every lane finishes ITS point at once (cuh:123-158 rewritten on the reduced
            // dots) instead of one lane in G finishing each point under an exec mask.
            float k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f;
#pragma unroll 2
            for (int pp = 0; pp < np; ++pp) {
                const int4 o = ro[pp];
                float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
                gather_load<Store<T>>(vbase, o.x, lane_bytes, v0);
                gather_load<Store<T>>(vbase, o.y, lane_bytes, v1);
                gather_load<Store<T>>(vbase, o.z, lane_bytes, v2);
                gather_load<Store<T>>(vbase, o.w, lane_bytes, v3);
                // d_k = <grad_out row, corner k> over this lane's channels
                float d0 = dot_eo(g, v0), d1 = dot_eo(g, v1), d2 = dot_eo(g, v2), d3 = dot_eo(g, v3);
                if (ATOMICS) {
                    // grad_value[corner k] += w_k * a * grad_out   (cuh:125,134,143,152)
                    const float4 w = rw[pp];
                    const float4 e = re[pp];
                    const int bits = __float_as_int(e.w);
                    const float a = e.x;
                    const float wa0 = w.x * a, wa1 = w.y * a, wa2 = w.z * a, wa3 = w.w * a;
                    if (bits & 1) {
#pragma unroll
                        for (int c = 0; c < VEC; ++c) atomic_accumulate(gvalue + o.x + c, wa0 * g[c]);
                    }
                    if (bits & 2) {
#pragma unroll
                        for (int c = 0; c < VEC; ++c) atomic_accumulate(gvalue + o.y + c, wa1 * g[c]);
                    }
                    if (bits & 4) {
#pragma unroll
                        for (int c = 0; c < VEC; ++c) atomic_accumulate(gvalue + o.z + c, wa2 * g[c]);
                    }
                    if (bits & 8) {
#pragma unroll
                        for (int c = 0; c < VEC; ++c) atomic_accumulate(gvalue + o.w + c, wa3 * g[c]);
                    }
                }
                if (!(p.dbg & 16)) row_sum4<G>(d0, d1, d2, d3);
                const bool mine = sub == (pp & (G - 1));
                k0 = mine ? d0 : k0; k1 = mine ? d1 : k1; k2 = mine ? d2 : k2; k3 = mine ? d3 : k3;
                if ((pp & (G - 1)) == G - 1 || pp == np - 1) {      // wave-uniform
                    const int mp = (pp & ~(G - 1)) + sub;
                    if (mp <= pp) {
                        const float4 w = rw[mp];
                        const float4 e = re[mp];
                        const int bits = __float_as_int(e.w);
                        // invalid corners count as zeros in every formula (their weight is not 0 in the
                        // fraction terms, so mask the dots)
                        const float q0d = (bits & 1) ? k0 : 0.f, q1d = (bits & 2) ? k1 : 0.f;
                        const float q2d = (bits & 4) ? k2 : 0.f, q3d = (bits & 8) ? k3 : 0.f;
                        const float a = e.x, lh = e.y, lw = e.z, hh = 1.f - lh, hw = 1.f - lw;
                        const Level lv = s_lvl[bits >> 4];
                        const float g_aw = w.x * q0d + w.y * q1d + w.z * q2d + w.w * q3d;
                        const float g_w = hh * (q1d - q0d) + lh * (q3d - q2d);
                        const float g_h = hw * (q2d - q0d) + lw * (q3d - q1d);
                        // park the point's three gradients in its (now consumed) LDS slot; they leave for
                        // HBM below as whole rows -- one 4-byte store per point and component cost as much
                        // as the entire gather (measured: 91 -> 52 us per clip without them)
                        s_e[r * kRowSlots + mp] = make_float4((float)lv.W * g_w * a, (float)lv.H * g_h * a, g_aw, 0.f);
                    }
                }
            }
            __syncthreads();
            // coalesced write-out: the chunk's 2*np grad_loc and np grad_attn elements of a row are
            // contiguous in memory; the row's G lanes write them G elements per instruction
            if (r < rows_valid && !(p.dbg & 8)) {
                const int64_t idx0 = row * LP + p0;
                const float *res = reinterpret_cast<const float *>(s_e + r * kRowSlots);
                for (int el = sub; el < 2 * np; el += G)
                    Store<T>::put(gloc + 2 * idx0 + el, res[(el >> 1) * 4 + (el & 1)]);
                for (int el = sub; el < np; el += G)
                    Store<T>::put(gaw + idx0 + el, res[el * 4 + 2]);
            }
            __syncthreads();
        }
    }
    if (s_bb) {     // the rows' tap-row intervals; layout [group, head, level, query] (query fastest, so
                    // that the scatter pass reads them coalesced while it walks the queries)
        const int64_t gm = ((int64_t)group * p.M + m) * nvl;
        for (int i = lane; i < rows_valid * nvl; i += kWave) {
            const int vl = i / rows_valid, rr = i - vl * rows_valid;
            *reinterpret_cast<int2 *>(p.bbox + ((gm + vl) * p.Lq + q0 + rr) * 2) =
                make_int2(s_bb[(rr * nvl + vl) * 2], s_bb[(rr * nvl + vl) * 2 + 1]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// grad_value by LDS-privatised scatter, accumulated in fp64
// ------------------------------------------------------------------------------------------------
// Measured on MI355X (scripts/ubench/lds_atomics.hip), clk per wave instruction per CU:
//   global fp32 atomics            ~80 G lane-op/s chip-wide (34 ms for 16 clips of the DeVIS decoder layer)
//   LDS ds_add_f32 / ds_add_rtn_f32 / ds_pk_add_f16      193      (a slow path: 3 clk per LANE)
//   LDS ds_add_u32 / ds_max_i32    5.0      ds_add_u64   7.5      ds_add_f64   9.1
// So grad_value is accumulated with ds_add_f64 in LDS.  (A 64-bit fixed-point variant with ds_add_u64
// was built first -- exact and order-independent -- but its float->fixed conversion costs ~13 VALU
// instructions per term and made the kernel VALU-bound: 1.25 G VALU wave-instructions per launch.)
//
//   work item = (clip, source frame f, head m, band); a band is a run of pixel ROWS of one level map
//   whose [rows, W, D] accumulator (8 bytes per channel) fits the workgroup's LDS budget.
//   The workgroup zeroes the band, scans every sampling point that reads (f, m, level) -- the
//   current-frame points of frame f and the temporal points of every (t, w) with
//   frame_table[t, w] == f --, and adds each bilinear corner that falls on a row it OWNS:
//       term  = fp32 product  w_corner * attn * grad_out[c]   (exactly the reference's atomicAdd
//               operand, cuh:125-152), widened to fp64 and added with ds_add_f64;
//   then streams the band to grad_value as float(sum) with plain coalesced 16-byte stores.  The fp64
//   sum of fp32 terms carries 29 more bits than the reference's fp32 running sum, so the result is the
//   correctly rounded sum for all practical purposes and independent of summation order up to 2^-53
//   relative (the reference's float atomicAdd result depends on the order at the 2^-24 level).
//   Every (pixel, head) of grad_value belongs to exactly one band, so the kernel OVERWRITES
//   grad_value -- no global atomics; a point whose two rows straddle two bands is visited by both
//   owners, each adding only its own row.  A level whose single row does not fit the budget takes the
//   float global-atomic branch of the same loop ("direct"), so any shape is handled.
//
//   lane mapping: stage 1 -- one lane per candidate point (tap arithmetic once per point, band test,
//   __ballot); stage 2 -- the hits are dealt to teams of G lanes (one team per point, RPW points per
//   wave pass) which fetch the tap record from the finder lane by ds_bpermute.  Lane i of team k adds
//   channel ((c + k) % VEC) * G + i at step c, so the teams of one half-wave hit disjoint LDS banks.
//
//   The grid is persistent (one 1024-thread workgroup per CU striding over the items) because the
//   number of bands depends on spatial_shapes, which lives in device memory (no host sync allowed);
//   item % M = head keeps the head -> XCD affinity of the gather kernels.
constexpr int kScatterThreads = 1024;
constexpr int kScatterMaxLevels = 32;
constexpr int kScatterMaxSources = 64;     // 1 + frames * window must fit
constexpr int kPointsCapBytes = 142 * 1024;   // band budget of msda_bwd_value_points_kernel (dynamic LDS)
constexpr int kScatterList = 3072;         // capacity of the survivor list (12 KiB of the 16 KiB LDS left by the band)
typedef unsigned long long u64;

template <typename T, int G>
__global__ void __launch_bounds__(kScatterThreads)
msda_bwd_value_lds_kernel(const Params p, int cap_slots, int dbg)
{
    constexpr int VEC = Store<T>::VEC;
    constexpr int RPW = kWave / G;
    constexpr int kWaves = kScatterThreads / kWave;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double *band = reinterpret_cast<double *>(lds_raw);
    __shared__ int s_H[kScatterMaxLevels], s_W[kScatterMaxLevels], s_R[kScatterMaxLevels],
        s_first[kScatterMaxLevels + 1], s_lsi[kScatterMaxLevels];
    __shared__ int s_src_t[kScatterMaxSources], s_src_vl[kScatterMaxSources], s_nsrc;    // sources of frame f
    __shared__ int s_list[kScatterList], s_count;      // surviving (source, query) groups awaiting their scan

    const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
    const int D = p.D, MD = p.M * p.D;
    const int L = p.L;      // levels of ONE source map (temporal virtual levels share them)
    if (tid == 0) {
        int first = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)p.shapes[2 * l], W = (int)p.shapes[2 * l + 1];
            const int R = min(H, cap_slots / max(1, W * D));     // rows per band; 0 = "direct" level
            s_H[l] = H; s_W[l] = W; s_R[l] = R; s_lsi[l] = (int)p.lsi[l];
            s_first[l] = first;
            first += (R > 0) ? (H + R - 1) / R : 1;
        }
        s_first[L] = first;
    }
    __syncthreads();
    const int NB = s_first[L];
    const int clips = p.groups / p.frames;
    const int64_t n_items = (int64_t)clips * p.frames * p.M * NB;
    const int team = lane / G, sub = lane % G;

    // Item order: heaviest first.  A band of a small level catches a larger share of its level's points
    // (a 1-band level catches all of them), so parts are walked from the last level down; with few
    // clips the items are also handed out DYNAMICALLY -- one atomic ticket counter per XCD residue
    // (blockIdx % 8) in the caller-zeroed workspace -- because their costs differ by ~7x and a static
    // stride leaves most CUs idle behind the unlucky ones (encoder shape, 1 clip: 2.85 -> see DESIGN).
    __shared__ long long s_item;
    // plenty of items per workgroup: a static stride balances well enough and skips the ticket traffic
    const bool dynamic = p.workspace != nullptr && (dbg & 16) == 0 && n_items < (int64_t)16 * gridDim.x;
    const int lane8 = blockIdx.x % 8;
    for (int64_t it = blockIdx.x;; it += gridDim.x) {
        int64_t item = it;
        if (dynamic) {
            if (tid == 0) s_item = (long long)atomicAdd(p.workspace + lane8, 1u) * 8 + lane8;
            __syncthreads();
            item = s_item;
        }
        if (item >= n_items) break;
        int l, part, m, f, clip;
        if (dynamic) {
            // heaviest first: levels from the last to the first; inside a level the bands of one
            // (clip, frame) stay adjacent
            const int64_t ctm = (int64_t)clips * p.frames * p.M;
            l = L - 1;
            int64_t local = item;
            while (l > 0 && local >= ctm * (s_first[l + 1] - s_first[l])) {
                local -= ctm * (s_first[l + 1] - s_first[l]);
                --l;
            }
            const int nb_l = s_first[l + 1] - s_first[l];
            m = (int)(local % p.M);
            int64_t rest = local / p.M;
            part = s_first[l] + (int)(rest % nb_l); rest /= nb_l;
            f = (int)(rest % p.frames);
            clip = (int)(rest / p.frames);
        } else {
            // static stride: all parts of one (clip, frame) adjacent -- one 128-byte loc line holds the
            // points of all levels, so concurrently running workgroups share their scan traffic in L2
            m = (int)(item % p.M);
            int64_t rest = item / p.M;
            part = (int)(rest % NB); rest /= NB;
            f = (int)(rest % p.frames);
            clip = (int)(rest / p.frames);
            l = 0;
            while (l + 1 < L && s_first[l + 1] <= part) ++l;
        }
        const int H = s_H[l], W = s_W[l], R = s_R[l];
        const bool whole_level = (R == 0);
        const bool direct = whole_level;
        const int r0 = whole_level ? 0 : (part - s_first[l]) * R;
        const int r1 = whole_level ? H - 1 : min(H, r0 + R) - 1;
        const int band_slots = direct ? 0 : (r1 - r0 + 1) * W * D;
        for (int i = tid * 2; i < band_slots; i += kScatterThreads * 2)
            *reinterpret_cast<uint4 *>(band + i) = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();

        // pixel (0, 0) of the level inside grad_value, for head m
        float *gmap = static_cast<float *>(p.grad_value) +
                      (((int64_t)clip * p.frames + f) * p.S + s_lsi[l]) * MD + m * D;

        // sources that read frame f: the current-frame points of frame f, then every temporal slot
        // (t, w) with frame_table[t, w] == f (list built once per item; repeats allowed)
        if (wave == 0) {        // frames * window <= 63: one wave tests every (t, w) at once and compacts
            const int tw = lane, n_tw = p.frames * p.window;
            const bool hit = tw < n_tw && p.ftab[tw] == f;
            const u64 bal = __ballot(hit);
            if (lane == 0) { s_src_t[0] = f; s_src_vl[0] = l; s_nsrc = (dbg & 4) ? 0 : 1 + (int)__popcll(bal); }
            if (hit) {
                const int n = 1 + (int)__popcll(bal & ((1ull << lane) - 1ull)), t = tw / p.window;
                s_src_t[n] = t; s_src_vl[n] = (tw - t * p.window) * L + l;
            }
        }
        __syncthreads();
        const int n_srcs = s_nsrc;
        // Candidate GROUPS are (source k, query q) pairs, each with P points at this level.  They are
        // culled in batches of 1024 against the band before they are scanned -- the
        // gather pass left, per (row, level), the interval of top tap rows in p.bbox; a group whose
        // interval misses rows [r0-1, r1] cannot touch the band -- and the survivors are compacted
        // into s_list; only they are scanned.  With local (encoder) or clustered (decoder) sampling
        // most groups die here; without p.bbox every group survives.
        const int n_groups = n_srcs * p.Lq;
        const int Pmax = max(p.PA, p.window > 0 ? p.PB : p.PA);
        const int VL = p.LA + p.LB;
        int n_cand = 0;
        const int pshift = (Pmax & (Pmax - 1)) == 0 ? __builtin_ctz(Pmax) : -1;     // i / Pmax as a shift
        auto source_of = [&](int k, int &t, int &vl, int &vlg, int &P, int &LP, const T *&loc, const T *&aw) {
            t = s_src_t[k]; vl = s_src_vl[k];
            const bool cur = (k == 0);
            vlg = cur ? vl : p.LA + vl;
            P = cur ? p.PA : p.PB;
            LP = cur ? p.LA * p.PA : p.LB * p.PB;
            loc = static_cast<const T *>(cur ? p.locA : p.locB);
            aw = static_cast<const T *>(cur ? p.awA : p.awB);
        };
        // one candidate per lane per pass; the NEXT pass's (x, y, attn) are loaded before this pass's
        // hits are processed, so the scan's memory latency hides behind stage 2
        auto fetch = [&](int i, float &x, float &y, float &a, int &qrow) {
            x = y = -10.f; a = 0.f; qrow = 0;
            if (i < n_cand) {
                const int ei = pshift >= 0 ? (i >> pshift) : i / Pmax, pt = i - ei * Pmax;
                const int e = s_list[ei];
                int t, vl, vlg, P, LP;
                const T *loc, *aw;
                source_of(e >> 24, t, vl, vlg, P, LP, loc, aw);
                if (pt < P) {
                    const int64_t gq = ((int64_t)clip * p.frames + t) * p.Lq + (e & 0xffffff);
                    const int64_t idx = (gq * p.M + m) * LP + vl * P + pt;
                    x = Store<T>::get(loc + 2 * idx);
                    y = Store<T>::get(loc + 2 * idx + 1);
                    a = Store<T>::get(aw + idx);
                    qrow = (int)gq;
                }
            }
        };
        // NC candidates per lane per pass: a band catches only ~1/7 of its level's points, so the hits
        // of NC candidates per lane are merged into dense rounds before they are dealt to teams.
        constexpr int NC = 2;
        constexpr int kPass = kScatterThreads * NC;
        float cx[NC], cy[NC], ca[NC];
        int cq[NC];

        // Stage 2 is software-pipelined over "hit groups" (RPW hits, one team of G lanes each): prep()
        // finds the team's hit, fetches its tap record from the finder lane (ds_bpermute) and ISSUES the
        // grad_out loads; the conversions + LDS adds of a group run only after the NEXT group's prep, so
        // the load latency (one 1024-thread workgroup per CU = only 4 waves per SIMD to hide it) overlaps
        // useful work.  Measured per clip: LDS adds 12 us, conversion VALU 44 us, exposed latency 58 us.
        struct Hit { int pix, bits; float w0, w1, w2, w3; float g[VEC]; };
        Hit pend;
        pend.bits = 0;
        auto prep = [&](u64 mk, int pix00, int bits, int qrow, float wa0, float wa1, float wa2, float wa3) {
            Hit h;
            // the j-th set bit of the (wave-uniform) mask goes to team j: found with scalar
            // s_ff1/s_bitset0, one v_cndmask per team instead of a per-lane 64-bit loop
            int from = 0;
            bool has = false;
            u64 mm = mk;
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const int sj = mm ? (int)__builtin_ctzll(mm) : -1;
                if (team == j) { from = sj < 0 ? 0 : sj; has = sj >= 0; }
                mm &= mm - 1;
            }
            h.pix = __shfl(pix00, from, kWave);
            const int b_all = __shfl(bits, from, kWave);
            h.bits = has ? b_all : 0;
            const int h_q = __shfl(qrow, from, kWave);
            h.w0 = __shfl(wa0, from, kWave); h.w1 = __shfl(wa1, from, kWave);
            h.w2 = __shfl(wa2, from, kWave); h.w3 = __shfl(wa3, from, kWave);
#pragma unroll
            for (int c = 0; c < VEC; ++c) h.g[c] = 0.f;
            if (h.bits) {
                const T *go = static_cast<const T *>(p.grad_out) + (int64_t)h_q * MD + m * D;
#pragma unroll
                for (int c = 0; c < VEC; ++c) h.g[c] = Store<T>::get(go + ((c + team) % VEC) * G + sub);
            }
            return h;
        };
        auto consume = [&](const Hit &h) {
            if (!h.bits) return;
            if (direct) {
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    float *dst = gmap + (int64_t)h.pix * MD + ((c + team) % VEC) * G + sub;
                    if (h.bits & 1) atomic_accumulate(dst, h.w0 * h.g[c]);
                    if (h.bits & 2) atomic_accumulate(dst + MD, h.w1 * h.g[c]);
                    if (h.bits & 4) atomic_accumulate(dst + (int64_t)W * MD, h.w2 * h.g[c]);
                    if (h.bits & 8) atomic_accumulate(dst + (int64_t)(W + 1) * MD, h.w3 * h.g[c]);
                }
                return;
            }
            // Branch-free: a corner this band does not own (or outside the map) adds 0 at the address
            // of a corner it does own -- 4*VEC independent ds_add_f64 per lane, no exec-mask juggling.
            // Terms are the fp32 products the reference hands to atomicAdd (cuh:125-152), widened to
            // fp64 (one v_cvt_f64_f32) and summed in fp64.
            if (dbg & 8) {          // measurement: everything but the LDS adds
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < VEC; ++c) acc += h.g[c];
                if (acc * h.w0 == 123.456f) band[0] = 1.0;
                return;
            }
            const int o1 = D, o2 = W * D, o3 = (W + 1) * D;
            const int safe = (h.bits & 1) ? 0 : (h.bits & 2) ? o1 : (h.bits & 4) ? o2 : o3;
            const int a0 = (h.bits & 1) ? 0 : safe, a1 = (h.bits & 2) ? o1 : safe;
            const int a2 = (h.bits & 4) ? o2 : safe, a3 = (h.bits & 8) ? o3 : safe;
            const float m0 = h.w0, m1 = h.w1, m2 = h.w2, m3 = h.w3;      // already 0 for unowned corners
            double *pixel = band + h.pix * D;
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                double *dst = pixel + ((c + team) % VEC) * G + sub;
                unsafeAtomicAdd(dst + a0, (double)(m0 * h.g[c]));
                unsafeAtomicAdd(dst + a1, (double)(m1 * h.g[c]));
                unsafeAtomicAdd(dst + a2, (double)(m2 * h.g[c]));
                unsafeAtomicAdd(dst + a3, (double)(m3 * h.g[c]));
            }
        };

        __syncthreads();
        if (tid == 0) s_count = 0;
        __syncthreads();
        // Cull in batches of one group per thread, appending survivors to s_list; the list is scanned
        // when another batch might not fit (or the groups are exhausted), so that sparse survivors
        // (local / clustered sampling) still fill whole scan passes.
        for (int gi0 = 0; gi0 < n_groups || gi0 == 0; gi0 += kScatterThreads) {
        {
            const int gi = gi0 + tid;
            bool keep = gi < n_groups;
            int k = 0, q = 0;
            if (keep) {
                k = gi / p.Lq; q = gi - k * p.Lq;
                if (p.bbox) {
                    int t, vl, vlg, P, LP;
                    const T *loc, *aw;
                    source_of(k, t, vl, vlg, P, LP, loc, aw);
                    const int64_t gm = (((int64_t)clip * p.frames + t) * p.M + m) * VL + vlg;
                    const int2 iv = *reinterpret_cast<const int2 *>(p.bbox + (gm * p.Lq + q) * 2);
                    keep = iv.y >= r0 - 1 && iv.x <= r1;      // empty interval (no valid point): false
                }
            }
            const u64 bal = __ballot(keep);
            int wbase = 0;
            if (lane == 0 && bal) wbase = atomicAdd(&s_count, (int)__popcll(bal));
            wbase = __shfl(wbase, 0, kWave);
            if (keep) s_list[wbase + (int)__popcll(bal & ((1ull << lane) - 1ull))] = (k << 24) | q;
        }
        __syncthreads();
        const int listed = s_count;
        const bool last = gi0 + kScatterThreads >= n_groups;
        if (!last && listed <= kScatterList - kScatterThreads) continue;     // room for another batch
        n_cand = listed * Pmax;
#pragma unroll
        for (int c = 0; c < NC; ++c) fetch(c * kScatterThreads + tid, cx[c], cy[c], ca[c], cq[c]);
        for (int base = 0; base < n_cand; base += kPass) {
            // ---- stage 1: tap arithmetic + band test for this lane's NC candidates
            int pixs[NC], bitss[NC], qrows[NC];
            float was[NC][4];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float x = cx[c], y = cy[c], a = ca[c];
                qrows[c] = cq[c];
                pixs[c] = 0; bitss[c] = 0;
                was[c][0] = was[c][1] = was[c][2] = was[c][3] = 0.f;
                const float h_im = __fsub_rn(__fmul_rn(y, (float)H), 0.5f);
                const float w_im = __fsub_rn(__fmul_rn(x, (float)W), 0.5f);
                if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                    const float hf = floorf(h_im), wf = floorf(w_im);
                    const int h_low = (int)hf, w_low = (int)wf;
                    // rows this band owns among the point's two rows
                    const bool top = h_low >= max(r0, 0) && h_low <= r1;
                    const bool bot = h_low + 1 >= r0 && h_low + 1 <= min(r1, H - 1);
                    const bool x0 = w_low >= 0, x1 = w_low + 1 <= W - 1;
                    bitss[c] = (top && x0 ? 1 : 0) | (top && x1 ? 2 : 0) | (bot && x0 ? 4 : 0) | (bot && x1 ? 8 : 0);
                    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
                    // weights of corners this band does not own are zeroed here, once per point
                    was[c][0] = (top && x0) ? hh * hw * a : 0.f; was[c][1] = (top && x1) ? hh * lw * a : 0.f;
                    was[c][2] = (bot && x0) ? lh * hw * a : 0.f; was[c][3] = (bot && x1) ? lh * lw * a : 0.f;
                    pixs[c] = (h_low - r0) * W + w_low;       // may be "virtual" for unowned corners
                }
            }
            // next pass's (x, y, attn) fly while this pass's hits are processed
#pragma unroll
            for (int c = 0; c < NC; ++c) fetch(base + kPass + c * kScatterThreads + tid, cx[c], cy[c], ca[c], cq[c]);
            if (dbg & 2) continue;
            // ---- merge: every round each lane offers its first unprocessed hit
#pragma unroll 1
            for (int round = 0; round < NC; ++round) {
                int pix00 = 0, bits = 0, qrow = 0;
                float wa0 = 0.f, wa1 = 0.f, wa2 = 0.f, wa3 = 0.f;
                bool taken = false;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const bool pick = !taken && bitss[c] != 0;
                    if (pick) {
                        pix00 = pixs[c]; bits = bitss[c]; qrow = qrows[c];
                        wa0 = was[c][0]; wa1 = was[c][1]; wa2 = was[c][2]; wa3 = was[c][3];
                        bitss[c] = 0;
                        taken = true;
                    }
                }
                u64 mask = __ballot(bits != 0);
                if (!mask) break;
                while (mask) {
                    const Hit h = prep(mask, pix00, bits, qrow, wa0, wa1, wa2, wa3);
#pragma unroll
                    for (int j = 0; j < RPW; ++j) mask &= mask - 1;
                    consume(pend);
                    pend = h;
                }
            }
        }
        __syncthreads();                                    // everyone is done with s_list
        if (tid == 0) s_count = 0;
        __syncthreads();
        }   // cull batches
        consume(pend);
        __syncthreads();
        // ---- flush the band: fixed point -> fp32, plain coalesced stores (D floats per pixel at stride M*D)
        const int vec_per_pix = D / 4;
        const int n_vec = band_slots / 4;
        float *gband = gmap + (int64_t)r0 * W * MD;
        for (int i = tid; i < n_vec; i += kScatterThreads) {
            const int pix = i / vec_per_pix, c4 = i - pix * vec_per_pix;
            const double *src = band + i * 4;
            float4 v;
            v.x = (float)src[0];
            v.y = (float)src[1];
            v.z = (float)src[2];
            v.w = (float)src[3];
            *reinterpret_cast<float4 *>(gband + (int64_t)pix * MD + c4 * 4) = v;
        }
        __syncthreads();
    }
}

// Broadcast of lane R of every team of G lanes to the whole team, R a compile-time constant: pure DPP for
// G = 4 (quad_perm) and G = 8 (quad_perm, then row_half_mirror brings the other quad's copy), so the
// scatter's hit hand-off does not go through the LDS crossbar its atomics are saturating.
template <int G, int R>
__device__ __forceinline__ int team_bcast(int v, int lane)
{
    if constexpr (G == 4) {
        return __builtin_amdgcn_update_dpp(0, v, R | (R << 2) | (R << 4) | (R << 6), 0xf, 0xf, true);
    } else if constexpr (G == 8) {
        constexpr int q = R & 3;
        const int t = __builtin_amdgcn_update_dpp(0, v, q | (q << 2) | (q << 4) | (q << 6), 0xf, 0xf, true);
        // row_half_mirror written only into the quads that do NOT hold lane R (bank_mask), the others keep t
        return __builtin_amdgcn_update_dpp(t, t, 0x141, 0xf, (R & 4) ? 0x5 : 0xA, false);
    } else {
        return __shfl(v, (lane / G) * G + R, kWave);
    }
}
template <int G, int R>
__device__ __forceinline__ float team_bcast(float v, int lane)
{
    return __int_as_float(team_bcast<G, R>(__float_as_int(v), lane));
}
template <int G, int R> constexpr u64 team_lane_mask()       // lane R of every team
{
    u64 m = 0;
    for (int j = 0; j < kWave / G; ++j) m |= 1ull << (j * G + R);
    return m;
}
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f)
{
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// Coarse culling summary for long candidate ranges (encoder shapes, Lq = S): (min, max) top tap row over blocks
// of kCullBlock consecutive queries of every (group, head, virtual level), reduced from the per-point records
// the gather pass left.  One wave per block.
constexpr int kCullBlock = 64;
constexpr int kLiveWords = 64;          // up to 2048 cull batches per item take the block-summary pre-pass
__global__ void __launch_bounds__(256)
msda_cull_summary_kernel(const Params p)
{
    const int VL = p.LA + p.LB, nblk = (p.Lq + kCullBlock - 1) / kCullBlock;
    const int64_t total = (int64_t)p.groups * p.M * VL * nblk;
    const int lane = threadIdx.x % kWave;
    for (int64_t e = (int64_t)blockIdx.x * 4 + threadIdx.x / kWave; e < total; e += (int64_t)gridDim.x * 4) {
        const int64_t gmv = e / nblk;
        const int q = (int)(e - gmv * nblk) * kCullBlock + lane;
        int mn = 0x7fffffff, mx = -0x7fffffff - 1;
        if (q < p.Lq) {
            const int2 iv = *reinterpret_cast<const int2 *>(p.bbox + (gmv * p.Lq + q) * 2);
            const int hr[4] = {(int)(short)(iv.x & 0xffff), iv.x >> 16, (int)(short)(iv.y & 0xffff), iv.y >> 16};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (hr[j] != kNoRow16) { mn = min(mn, hr[j]); mx = max(mx, hr[j]); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mn = min(mn, __shfl_xor(mn, o, kWave)); mx = max(mx, __shfl_xor(mx, o, kWave)); }
        if (lane == 0) *reinterpret_cast<int2 *>(p.bsum + e * 2) = make_int2(mn, mx);
    }
}

// msda_bwd_value_points_kernel -- the scatter for the common case PA, PB <= 4, where the gather pass
// leaves the top tap row of every POINT (4 x int16 per (row, level)) in the workspace.  Same work items,
// same band accumulators, same arithmetic and flush as msda_bwd_value_lds_kernel; what differs is
//   * the cull keeps exactly the POINTS that have a row in the band and compacts them into the list as
//     (source k : 6 | point : 2 | query : 24), so in the scan (nearly) every lane is a hit and stage 2
//     deals lanes [R*RPW, R*RPW + RPW) to the teams in G fixed sub-rounds: no merge rounds, no scalar
//     bit-select chain;
//   * the items are SOFTWARE-PIPELINED: an item is ~800 hits, i.e. only ~6 hit groups per wave, so its
//     fixed latencies (ticket, frame table, the cull's table reads, the scan's loc/attn reads) used to
//     be exposed one after the other with all 16 waves waiting in lock-step.  Now wave 0 stages item
//     i+2 (ticket, decode, source list) and every thread issues the cull loads of item i+1 BEFORE the
//     scan of item i, compacts them after it, and issues the first scan pass of item i+1 before
//     flushing the band of item i; the flush re-zeroes the band as it reads it.
template <typename T, int G>
__global__ void __launch_bounds__(kScatterThreads)
msda_bwd_value_points_kernel(const Params p, int cap_slots, int dbg)
{
    constexpr int VEC = Store<T>::VEC;
    constexpr int RPW = kWave / G;
    constexpr int U = 2;                  // groups per thread per cull batch
    constexpr int kBatch = U * kScatterThreads;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double *band = reinterpret_cast<double *>(lds_raw) + G * VEC;       // one guard pixel before (and after) the band
    cap_slots -= 2 * G * VEC;
    __shared__ int s_H[kScatterMaxLevels], s_W[kScatterMaxLevels], s_R[kScatterMaxLevels],
        s_first[kScatterMaxLevels + 1], s_lsi[kScatterMaxLevels];
    __shared__ int s_nsrc[3];
    __shared__ int s_dec[3][8];
    // per source, precomputed when the item is staged: first culling-table entry, first loc/attn element, first query row
    __shared__ long long s_src_tab[3][kScatterMaxSources], s_src_loc[3][kScatterMaxSources];
    __shared__ int s_src_q0[3][kScatterMaxSources];
    __shared__ int s_src_gmv[3][kScatterMaxSources];   // (group, head, virtual level) index of the source: row of p.bsum
    __shared__ unsigned s_live[kLiveWords];            // bitmap of the cull batches that hold a live block           // staged item: valid, l, m, f, clip, r0, r1, direct
    __shared__ int s_list[kScatterList], s_cnt[3], s_valid;
    __shared__ int s_ftab[kScatterMaxSources];      // the frame table, read once

    const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
    const int D = p.D, MD = p.M * p.D, L = p.L, VL = p.LA + p.LB;
    if (tid == 0) {
        int first = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)p.shapes[2 * l], W = (int)p.shapes[2 * l + 1];
            const int R = min(H, cap_slots / max(1, W * D));     // rows per band; 0 = "direct" level
            s_H[l] = H; s_W[l] = W; s_R[l] = R; s_lsi[l] = (int)p.lsi[l];
            s_first[l] = first;
            first += (R > 0) ? (H + R - 1) / R : 1;
        }
        s_first[L] = first;
        s_cnt[0] = s_cnt[1] = s_cnt[2] = 0;
        s_valid = 0;
    }
    if (tid < p.frames * p.window) s_ftab[tid] = p.ftab[tid];
    __syncthreads();
    const int NB = s_first[L];
    const int clips = p.groups / p.frames;
    const int64_t n_items = (int64_t)clips * p.frames * p.M * NB;
    const int team = lane / G, sub = lane % G;
    const bool dynamic = p.workspace != nullptr && (dbg & 16) == 0 && n_items < (int64_t)16 * gridDim.x;
    const int lane8 = blockIdx.x % 8;

    struct Item { int valid, l, m, f, clip, r0, r1, direct, H, W; };

    // wave 0: draw the seq-th item of this workgroup, decode it, list the sources that read its frame
    auto stage_item = [&](int64_t seq, int buf) {
        unsigned lo32 = 0, hi32 = 0;
        if (lane == 0) {
            const int64_t id = dynamic ? (int64_t)atomicAdd(p.workspace + lane8, 1u) * 8 + lane8
                                       : (int64_t)blockIdx.x + seq * gridDim.x;
            lo32 = (unsigned)id; hi32 = (unsigned)((u64)id >> 32);
        }
        lo32 = __builtin_amdgcn_readfirstlane(lo32); hi32 = __builtin_amdgcn_readfirstlane(hi32);
        const int64_t item = (int64_t)(((u64)hi32 << 32) | lo32);
        if (item >= n_items) { if (lane == 0) s_dec[buf][0] = 0; return; }
        int l, part, m, f, clip;
        if (dynamic) {      // heaviest first: levels from the last to the first (see msda_bwd_value_lds_kernel)
            const int64_t ctm = (int64_t)clips * p.frames * p.M;
            l = L - 1;
            int64_t local = item;
            while (l > 0 && local >= ctm * (s_first[l + 1] - s_first[l])) {
                local -= ctm * (s_first[l + 1] - s_first[l]);
                --l;
            }
            const unsigned nb_l = (unsigned)(s_first[l + 1] - s_first[l]);
            unsigned rest = (unsigned)local;                 // dynamic => n_items < 16 * grid: fits 32 bits
            m = (int)(rest % (unsigned)p.M); rest /= (unsigned)p.M;
            part = s_first[l] + (int)(rest % nb_l); rest /= nb_l;
            f = (int)(rest % (unsigned)p.frames);
            clip = (int)(rest / (unsigned)p.frames);
        } else if (n_items < 0x7fffffffLL) {
            unsigned rest = (unsigned)item;
            m = (int)(rest % (unsigned)p.M); rest /= (unsigned)p.M;
            part = (int)(rest % (unsigned)NB); rest /= (unsigned)NB;
            f = (int)(rest % (unsigned)p.frames);
            clip = (int)(rest / (unsigned)p.frames);
            l = 0;
            while (l + 1 < L && s_first[l + 1] <= part) ++l;
        } else {
            m = (int)(item % p.M);
            int64_t rest = item / p.M;
            part = (int)(rest % NB); rest /= NB;
            f = (int)(rest % p.frames);
            clip = (int)(rest / p.frames);
            l = 0;
            while (l + 1 < L && s_first[l + 1] <= part) ++l;
        }
        const int H = s_H[l], R = s_R[l];
        const bool direct = (R == 0);
        const int r0 = direct ? 0 : (part - s_first[l]) * R;
        const int r1 = direct ? H - 1 : min(H, r0 + R) - 1;
        const int n_tw = p.frames * p.window;
        const bool hit = lane < n_tw && s_ftab[lane] == f;
        const u64 bal = __ballot(hit);
        if (lane == 0) {
            s_dec[buf][0] = 1; s_dec[buf][1] = l; s_dec[buf][2] = m; s_dec[buf][3] = f; s_dec[buf][4] = clip;
            s_dec[buf][5] = r0; s_dec[buf][6] = r1; s_dec[buf][7] = direct ? 1 : 0;
            const int64_t g = (int64_t)clip * p.frames + f;
            s_src_tab[buf][0] = ((g * p.M + m) * VL + l) * p.Lq;
            s_src_loc[buf][0] = (g * p.Lq * p.M + m) * ((int64_t)p.LA * p.PA) + l * p.PA;
            s_src_q0[buf][0] = (int)(g * p.Lq);
            s_src_gmv[buf][0] = (int)((g * p.M + m) * VL + l);
            s_nsrc[buf] = ((dbg & 4) || ((dbg >> 8) & (1 << l))) ? 0 : 1 + (int)__popcll(bal);    // dbg: skip all / a level's sources
        }
        if (hit) {
            const int n = 1 + (int)__popcll(bal & ((1ull << lane) - 1ull)), t = lane / p.window;
            const int vl = (lane - t * p.window) * L + l;
            const int64_t g = (int64_t)clip * p.frames + t;
            s_src_tab[buf][n] = ((g * p.M + m) * VL + p.LA + vl) * p.Lq;
            s_src_loc[buf][n] = (g * p.Lq * p.M + m) * ((int64_t)p.LB * p.PB) + vl * p.PB;
            s_src_q0[buf][n] = (int)(g * p.Lq);
            s_src_gmv[buf][n] = (int)((g * p.M + m) * VL + p.LA + vl);
        }
    };
    auto load_item = [&](int buf) {
        Item it;
        it.valid = s_dec[buf][0];
        it.l = s_dec[buf][1]; it.m = s_dec[buf][2]; it.f = s_dec[buf][3]; it.clip = s_dec[buf][4];
        it.r0 = s_dec[buf][5]; it.r1 = s_dec[buf][6]; it.direct = s_dec[buf][7];
        it.H = it.valid ? s_H[it.l] : 1; it.W = it.valid ? s_W[it.l] : 1;
        return it;
    };
    // the per-point rows of U groups per thread, all loads in flight together
    const float inv_Lq = 1.0f / (float)p.Lq;
    const int strideA = p.M * p.LA * p.PA, strideB = p.M * p.LB * p.PB;      // loc/attn elements per query
    auto cull_load = [&](const Item &it, int buf, int gi0, int2 (&iv)[U], unsigned (&ent)[U]) {
        const int ng = s_nsrc[buf] * p.Lq;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int gi = gi0 + u * kScatterThreads + tid;
            iv[u] = make_int2((int)0x80008000u, (int)0x80008000u);
            ent[u] = 0u;
            if (gi < ng) {
                int k, q;
                if (ng < (1 << 22)) {       // gi / Lq by reciprocal: exact to +-1 below 2^22, then fixed up
                    k = (int)((float)gi * inv_Lq);
                    q = gi - k * p.Lq;
                    if (q < 0) { --k; q += p.Lq; }
                    if (q >= p.Lq) { ++k; q -= p.Lq; }
                } else {
                    k = gi / p.Lq; q = gi - k * p.Lq;
                }
                iv[u] = *reinterpret_cast<const int2 *>(p.bbox + (s_src_tab[buf][k] + q) * 2);
                ent[u] = ((unsigned)k << 26) | (unsigned)q;
            }
        }
    };
    // Surviving points of this lane's U groups (bit 4u + j = point j of group u), their exclusive rank in the wave
    // and the wave's total.  Entries are listed LANE-major: the points of one (query, head, level) stay adjacent in
    // the list, so when several of them survive (local / clustered sampling) neighbouring lanes of the scan read
    // the same 32-byte loc / 16-byte attn segment instead of one 64-byte granule each.
    struct Marks { unsigned pm; int excl, total; };
    auto mark = [&](const Item &it, const int2 (&iv)[U]) {
        Marks mk;
        mk.pm = 0u;
        const int lo = min(it.r0 - 1, 32767), hi = min(it.r1, 32767);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int hr[4] = {(int)(short)(iv[u].x & 0xffff), iv[u].x >> 16, (int)(short)(iv[u].y & 0xffff), iv[u].y >> 16};
#pragma unroll
            for (int j = 0; j < 4; ++j) mk.pm |= (hr[j] >= lo && hr[j] <= hi) ? (1u << (4 * u + j)) : 0u;
        }
        // wave-wide inclusive scan of the per-lane counts with DPP (row_shr 1, 2, 4, 8, then row_bcast 15 / 31)
        const int cnt = __popc(mk.pm);
        int v = cnt;
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
        mk.excl = v - cnt;
        mk.total = __builtin_amdgcn_readlane(v, kWave - 1);
        return mk;
    };
    // One attempt at appending a batch's surviving points behind the `listed` entries of s_list, with ONE
    // workgroup barrier: every wave that still has entries (`pendw`) reserves a range with one atomic on
    // the batch's relative counter and writes it if it ends inside the list.  Ranges are handed out in
    // order, so when the batch does not fit the written entries are a prefix that ends where the first
    // failing wave starts (s_valid).  Counters rotate over three slots: slot j is reset two barriers
    // before it is used again, so no wave can still be reading it.  Returns the batch's entry count.
    int ci = 0;
    auto try_add = [&](const Marks &mk, const unsigned (&ent)[U], bool pendw, int listed, bool &fit) {
        int wbase = 0;
        if (lane == 0 && pendw) wbase = atomicAdd(&s_cnt[ci], mk.total);
        wbase = __shfl(wbase, 0, kWave);
        fit = pendw && listed + wbase + mk.total <= kScatterList;
        if (fit) {
            int pos = listed + wbase + mk.excl;
#pragma unroll
            for (int b = 0; b < 4 * U; ++b) {
                if ((mk.pm >> b) & 1u) {
                    s_list[pos] = (int)(ent[b / 4] | ((unsigned)(b & 3) << 24));
                    ++pos;
                }
            }
        } else if (pendw && listed + wbase <= kScatterList && lane == 0) {
            s_valid = listed + wbase;
        }
        __syncthreads();
        const int rel = s_cnt[ci];
        if (tid == 0) s_cnt[(ci + 2) % 3] = 0;
        ci = (ci + 1) % 3;
        return rel;
    };

    struct Hit { int pix, bits; float w0, w1, w2, w3; float g[VEC]; };
    Hit pend;
    pend.bits = 0;
    int pend_q = -1;            // query row of `pend` (its grad_out row may serve the next hit of the team)
    auto consume = [&](const Item &it, const Hit &h) {
        if (!h.bits) return;
        const int W = it.W;
        if (it.direct) {
            float *gmap = static_cast<float *>(p.grad_value) +
                          (((int64_t)it.clip * p.frames + it.f) * p.S + s_lsi[it.l]) * MD + it.m * D;
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                float *dst = gmap + (int64_t)h.pix * MD + ((c + team) % VEC) * G + sub;
                if (h.bits & 1) atomic_accumulate(dst, h.w0 * h.g[c]);
                if (h.bits & 2) atomic_accumulate(dst + MD, h.w1 * h.g[c]);
                if (h.bits & 4) atomic_accumulate(dst + (int64_t)W * MD, h.w2 * h.g[c]);
                if (h.bits & 8) atomic_accumulate(dst + (int64_t)(W + 1) * MD, h.w3 * h.g[c]);
            }
            return;
        }
        if (dbg & 8) {          // measurement: everything but the LDS adds
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < VEC; ++c) acc += h.g[c];
            if (acc * h.w0 == 123.456f) band[0] = 1.0;
            return;
        }
        // Branch-free ds_add_f64 (terms as in msda_bwd_value_lds_kernel).  The weights of corners this band
        // does not own are already 0; such a corner adds 0.0 at a harmless address: a row the band does
        // not own is replaced by the other row of the point, a column outside the map falls on the
        // neighbouring pixel (one guard pixel sits before and after the band).  With D = G * VEC a
        // compile-time constant the x+1 corner is an immediate offset of the same address register.
        constexpr int kD = G * VEC;
        const int top_pix = (h.bits & 3) ? h.pix : h.pix + W;
        const int bot_pix = (h.bits & 12) ? h.pix + W : h.pix;
        double *top = band + top_pix * kD + sub, *bot = band + bot_pix * kD + sub;
        const float2v w01 = {h.w0, h.w1}, w23 = {h.w2, h.w3};
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const int ch = ((c + team) % VEC) * G;
            const float2v gg = {h.g[c], h.g[c]};
            const float2v t01 = w01 * gg, t23 = w23 * gg;               // v_pk_mul_f32: the fp32 products of cuh:125-152
            unsafeAtomicAdd(top + ch, (double)t01.x);
            unsafeAtomicAdd(top + ch + kD, (double)t01.y);
            unsafeAtomicAdd(bot + ch, (double)t23.x);
            unsafeAtomicAdd(bot + ch + kD, (double)t23.y);
        }
    };
    auto fetchp = [&](const Item &it, int buf, int i, int listed, float &x, float &y, float &a, int &qrow) {
        x = y = -10.f; a = 0.f; qrow = 0;
        if (i < listed && !(dbg & 64)) {      // dbg 64: no loc/attn reads (measurement)
            const unsigned e = (unsigned)s_list[i];
            const int k = (int)(e >> 26), q = (int)(e & 0xffffffu);
            const bool curf = (k == 0);
            const int64_t idx = s_src_loc[buf][k] + (int64_t)q * (curf ? strideA : strideB) + (int)((e >> 24) & 3u);
            const T *loc = static_cast<const T *>(curf ? p.locA : p.locB);
            const T *aw = static_cast<const T *>(curf ? p.awA : p.awB);
            x = Store<T>::get(loc + 2 * idx);
            y = Store<T>::get(loc + 2 * idx + 1);
            a = Store<T>::get(aw + idx);
            qrow = s_src_q0[buf][k] + q;
        }
    };
    // Scans the `listed` points of s_list (one per lane per pass); `primed`: the first pass's (x, y, attn)
    // are already in (x, y, a, qrow).  No barrier: the caller separates it from the next list write.
    auto scan_points = [&](const Item &it, int buf, int listed, bool primed, float &x, float &y, float &a, int &qrow) {
        const int H = it.H, W = it.W, r0 = it.r0, r1 = it.r1;
        if (!primed) fetchp(it, buf, tid, listed, x, y, a, qrow);
        for (int base = 0; base < listed; base += kScatterThreads) {
            int pix = 0, bits = 0;
            const int qr = qrow;
            float wa0 = 0.f, wa1 = 0.f, wa2 = 0.f, wa3 = 0.f;
            const float h_im = __fsub_rn(__fmul_rn(y, (float)H), 0.5f);
            const float w_im = __fsub_rn(__fmul_rn(x, (float)W), 0.5f);
            if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                const float hf = floorf(h_im), wf = floorf(w_im);
                const int h_low = (int)hf, w_low = (int)wf;
                const bool top = h_low >= max(r0, 0) && h_low <= r1;          // rows this band owns
                const bool bot = h_low + 1 >= r0 && h_low + 1 <= min(r1, H - 1);
                const bool x0 = w_low >= 0, x1 = w_low + 1 <= W - 1;
                bits = (top && x0 ? 1 : 0) | (top && x1 ? 2 : 0) | (bot && x0 ? 4 : 0) | (bot && x1 ? 8 : 0);
                const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
                wa0 = (top && x0) ? hh * hw * a : 0.f; wa1 = (top && x1) ? hh * lw * a : 0.f;
                wa2 = (bot && x0) ? lh * hw * a : 0.f; wa3 = (bot && x1) ? lh * lw * a : 0.f;
                pix = (h_low - r0) * W + w_low;
            }
            fetchp(it, buf, base + kScatterThreads + tid, listed, x, y, a, qrow);      // next pass's loads fly
            if (dbg & 2) continue;
            const u64 mask = __ballot(bits != 0);
            // sub-round R: every team takes the hit of ITS OWN lane R (so the record moves by DPP, not
            // through the LDS crossbar, when G is 4 or 8), loads the hit's grad_out and retires the
            // previous hit's adds
            auto sub_round = [&](auto Rc) {
                constexpr int R = decltype(Rc)::value;
                if (!(mask & team_lane_mask<G, R>())) return;
                Hit h;
                h.pix = team_bcast<G, R>(pix, lane);
                h.bits = team_bcast<G, R>(bits, lane);
                const int h_q = team_bcast<G, R>(qr, lane);
                h.w0 = team_bcast<G, R>(wa0, lane); h.w1 = team_bcast<G, R>(wa1, lane);
                h.w2 = team_bcast<G, R>(wa2, lane); h.w3 = team_bcast<G, R>(wa3, lane);
                // The list is lane-major, so with local / clustered sampling a team's consecutive hits are often
                // points of the SAME query: its grad_out row is then already in the previous record.
                const bool same_row = VEC <= 4 && pend.bits != 0 && h_q == pend_q;
#pragma unroll
                for (int c = 0; c < VEC; ++c) h.g[c] = same_row ? pend.g[c] : 0.f;
                if (h.bits && !same_row && !(dbg & 32)) {
                    const T *go = static_cast<const T *>(p.grad_out) + (int64_t)h_q * MD + it.m * D;
#pragma unroll
                    for (int c = 0; c < VEC; ++c) h.g[c] = Store<T>::get(go + ((c + team) % VEC) * G + sub);
                }
                // (8-channel lanes: no one-group-ahead pipelining, the second record does not fit the registers)
                if constexpr (VEC > 4) { consume(it, h); } else { consume(it, pend); pend = h; pend_q = h_q; }
            };
            if constexpr (G == 4 || G == 8) {
                static_for<G>(sub_round);
            } else {
#pragma unroll 1
                for (int R = 0; R < G; ++R) {
                    if (!((mask >> R) & team_lane_mask<G, 0>())) continue;
                    const int from = team * G + R;
                    Hit h;
                    h.pix = __shfl(pix, from, kWave);
                    h.bits = __shfl(bits, from, kWave);
                    const int h_q = __shfl(qr, from, kWave);
                    h.w0 = __shfl(wa0, from, kWave); h.w1 = __shfl(wa1, from, kWave);
                    h.w2 = __shfl(wa2, from, kWave); h.w3 = __shfl(wa3, from, kWave);
#pragma unroll
                    for (int c = 0; c < VEC; ++c) h.g[c] = 0.f;
                    if (h.bits && !(dbg & 32)) {
                        const T *go = static_cast<const T *>(p.grad_out) + (int64_t)h_q * MD + it.m * D;
#pragma unroll
                        for (int c = 0; c < VEC; ++c) h.g[c] = Store<T>::get(go + ((c + team) % VEC) * G + sub);
                    }
                    if constexpr (VEC > 4) { consume(it, h); } else { consume(it, pend); pend = h; }
                }
            }
        }
    };
    // Lists one batch; when it does not fit, scans the written prefix and retries the waves that failed.
    auto add_batch = [&](const Item &it, int buf, const Marks &mk, const unsigned (&ent)[U], int listed,
                         float &x, float &y, float &a, int &qrow) {
        bool pendw = mk.total > 0;
        for (;;) {
            bool fit;
            const int rel = try_add(mk, ent, pendw, listed, fit);
            if (listed + rel <= kScatterList) return listed + rel;
            scan_points(it, buf, s_valid, false, x, y, a, qrow);
            __syncthreads();
            listed = 0;
            pendw = pendw && !fit;
        }
    };

    // Flushes 16-byte vectors [v0, v1) of the item's band as float(sum) with plain coalesced stores
    // (D floats per pixel at stride M*D), re-zeroing the band on the way.
    auto flush = [&](const Item &it, int v0, int v1) {
        const int vec_per_pix = D / 4;
        float *gband = static_cast<float *>(p.grad_value) +
                       (((int64_t)it.clip * p.frames + it.f) * p.S + s_lsi[it.l] + (int64_t)it.r0 * it.W) * MD + it.m * D;
        for (int i = v0 + tid; i < v1; i += kScatterThreads) {
            const int pix = i / vec_per_pix, c4 = i - pix * vec_per_pix;
            double *src = band + i * 4;
            const double2 lo2 = *reinterpret_cast<const double2 *>(src), hi2 = *reinterpret_cast<const double2 *>(src + 2);
            *reinterpret_cast<uint4 *>(src) = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4 *>(src + 2) = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<float4 *>(gband + (int64_t)pix * MD + c4 * 4) =
                make_float4((float)lo2.x, (float)lo2.y, (float)hi2.x, (float)hi2.y);
        }
    };

    // ---- prologue: stage item 0 (and 1), zero the band, cull + prime item 0.
    // With a STATIC item order the pipeline runs two items ahead (item i+1's table reads fly during the
    // scan of item i).  With dynamic tickets an item is claimed only when the previous one has been
    // scanned -- a workgroup sitting on a heavy item must not hoard work the others could take -- and
    // its table reads overlap the first half of the flush instead.
    int bc = 0, bn = 1, bnn = 2;
    if (wave == 0) { stage_item(0, bc); if (!dynamic) stage_item(1, bn); else if (lane == 0) s_dec[bn][0] = 0; }
    for (int i = tid * 2; i < cap_slots + 2 * G * VEC; i += kScatterThreads * 2)
        *reinterpret_cast<uint4 *>(band - G * VEC + i) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    Item cur = load_item(bc), nxt = load_item(bn);
    int64_t seq = dynamic ? 1 : 2;
    int2 iv[U];
    unsigned ent[U];
    float x = -10.f, y = -10.f, a = 0.f;
    int qrow = 0;
    int cur_listed = 0;
    bool cur_overflow = false;
    if (cur.valid) {
        cull_load(cur, bc, 0, iv, ent);
        const Marks mk = mark(cur, iv);
        bool fit;
        const int rel = try_add(mk, ent, mk.total > 0, 0, fit);
        if (rel > kScatterList) cur_overflow = true;
        else { cur_listed = rel; fetchp(cur, bc, tid, cur_listed, x, y, a, qrow); }
    }
    while (cur.valid) {
        // (a) wave 0 stages the item after next; the next item's table reads start now and land while
        //     this item is scanned
        if (!dynamic) {
            if (wave == 0) stage_item(seq, bnn);
            ++seq;
            if (nxt.valid) cull_load(nxt, bn, 0, iv, ent);
        }
        // (b) scan this item
        {
            const int ng = s_nsrc[bc] * p.Lq;
            const int nbat = (ng + kBatch - 1) / kBatch;
            // Long candidate ranges (encoder shapes): a pre-pass over the 64-query block summaries marks the cull
            // batches that hold a live block; with local sampling all but a few are skipped outright.
            const bool skipping = p.bsum != nullptr && nbat > 4 && nbat <= 32 * kLiveWords;
            if (skipping) {
                if (tid < kLiveWords) s_live[tid] = 0u;
                __syncthreads();
                const int nblk = (p.Lq + kCullBlock - 1) / kCullBlock, nb_tot = s_nsrc[bc] * nblk;
                const int lo = min(cur.r0 - 1, 32767), hi = min(cur.r1, 32767);
                for (int b = tid; b < nb_tot; b += kScatterThreads) {
                    const int ks = b / nblk, blk = b - ks * nblk;
                    const int2 mm = *reinterpret_cast<const int2 *>(p.bsum + ((int64_t)s_src_gmv[bc][ks] * nblk + blk) * 2);
                    if (mm.y >= lo && mm.x <= hi) {
                        const int g0 = ks * p.Lq + blk * kCullBlock, g1 = min(g0 + kCullBlock, ks * p.Lq + p.Lq) - 1;
                        atomicOr(&s_live[(g0 / kBatch) >> 5], 1u << ((g0 / kBatch) & 31));
                        atomicOr(&s_live[(g1 / kBatch) >> 5], 1u << ((g1 / kBatch) & 31));
                    }
                }
                __syncthreads();
            }
            auto next_live = [&](int b) {       // first batch >= b worth culling (nbat if none); workgroup-uniform
                if (!skipping) return min(b, nbat);
                while (b < nbat) {
                    const unsigned w = s_live[b >> 5] >> (b & 31);
                    if (w) return min(b + (int)__builtin_ctz(w), nbat);
                    b = (b | 31) + 1;
                }
                return nbat;
            };
            int listed = cur_listed;
            int2 iv2[U];
            unsigned ent2[U], ent3[U];
            int b = next_live(1);
            if (cur_overflow) {         // the first batch did not fit when it was compacted early: redo it here
                cull_load(cur, bc, 0, iv2, ent2);
                const Marks mk = mark(cur, iv2);
                __syncthreads();
                listed = add_batch(cur, bc, mk, ent2, 0, x, y, a, qrow);
                if (b < nbat) cull_load(cur, bc, b * kBatch, iv2, ent2);
                if (b >= nbat || listed >= kScatterThreads) {
                    scan_points(cur, bc, listed, false, x, y, a, qrow);
                    __syncthreads();
                    listed = 0;
                }
            } else {
                if (b < nbat) cull_load(cur, bc, b * kBatch, iv2, ent2);
                if (b >= nbat || listed >= kScatterThreads) {
                    scan_points(cur, bc, listed, true, x, y, a, qrow);
                    if (b < nbat) __syncthreads();
                    listed = 0;
                }   // else: a sparse first batch of many -- keep accumulating (the primed pass is dropped)
            }
            while (b < nbat) {
                const Marks mk = mark(cur, iv2);
#pragma unroll
                for (int u = 0; u < U; ++u) ent3[u] = ent2[u];
                const int bnext = next_live(b + 1);
                if (bnext < nbat) cull_load(cur, bc, bnext * kBatch, iv2, ent2);      // next live batch's reads fly
                listed = add_batch(cur, bc, mk, ent3, listed, x, y, a, qrow);
                b = bnext;
                if (b < nbat && listed < kScatterThreads) continue;      // not yet a full pass
                scan_points(cur, bc, listed, false, x, y, a, qrow);
                if (b < nbat) __syncthreads();
                listed = 0;
            }
        }
        // (c) last hit group of this item
        consume(cur, pend);
        pend.bits = 0;
        if (dynamic) {                  // claim the next item now that this one is scanned
            if (wave == 0) stage_item(seq, bn);
            ++seq;
        }
        __syncthreads();                // (d) every add of this item is in the band; s_list is free
        if (dynamic) {
            nxt = load_item(bn);
            if (nxt.valid) cull_load(nxt, bn, 0, iv, ent);
        }
        const int flush_vecs = cur.direct ? 0 : (cur.r1 - cur.r0 + 1) * cur.W * D / 4;
        // dynamic order: the first half of the flush hides the table reads just issued; static order: they
        // landed long ago, so the whole flush can hide the first scan pass's reads instead
        const int flush_cut = dynamic ? flush_vecs / 2 : 0;
        flush(cur, 0, flush_cut);
        // (e) compact the next item's first batch and start its first scan pass
        int nxt_listed = 0;
        bool nxt_overflow = false;
        if (nxt.valid) {
            const Marks mk = mark(nxt, iv);
            bool fit;
            const int rel = try_add(mk, ent, mk.total > 0, 0, fit);
            if (rel > kScatterList) nxt_overflow = true;
            else { nxt_listed = rel; fetchp(nxt, bn, tid, nxt_listed, x, y, a, qrow); }
        }
        // (f) the rest of the flush hides the first scan pass's reads
        flush(cur, flush_cut, flush_vecs);
        __syncthreads();
        cur = nxt; cur_listed = nxt_listed; cur_overflow = nxt_overflow;
        if (dynamic) {
            const int tmp = bc; bc = bn; bn = tmp;
            nxt.valid = 0;
        } else {
            nxt = load_item(bnn);
            const int tmp = bc; bc = bn; bn = bnn; bnn = tmp;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// grad_value by OWNER-COMPUTES scatter (round 2): no floating-point atomics at all
// ------------------------------------------------------------------------------------------------
// The LDS scatter above is bound by the fp64 LDS atomic unit: 2 KiB of read-modify-write per hit, 9.1 clk per
// ds_add_f64 wave instruction, a floor of 0.65 ms on the DeVIS decoder workload.  Here every pixel of a band
// has an OWNER -- one quad of the workgroup, lane c holding channels [4c, 4c+4) of both halves of the pixel in
// registers -- and a hit only (1) has its grad_out row staged in LDS once (LDS-DMA, 128 B) and (2) links one
// 8-byte entry {weight, next} per owned corner into that pixel's list (ds_wrxchg_rtn_b32 on the list head: an
// integer exchange, not a float atomic).  The owners then walk their lists with plain LDS reads
// (8 B entry + 2 x 16 B of the row per lane) and accumulate in fp32 registers: ~650 B of plain LDS traffic
// per hit instead of 2 KiB of atomics.  Same items (clip, source frame, head, band of pixel rows of one
// level), same per-point culling records from the gather pass, same survivor list as
// msda_bwd_value_points_kernel; bands are sized by the owners' registers (kOwnPix pixels) instead of by LDS.
// The sum of a pixel's terms is an fp32 sum in list order (the reference's atomicAdd order is arbitrary too,
// cuh:125-152); terms are the products (w_corner * attn) * grad_out[c].
constexpr int kOwnThreads = 1024;
constexpr int kOwnQuads = kOwnThreads / 4;
constexpr int kOwnSlots = 4;                        // pixels per owner quad
constexpr int kOwnPix = kOwnQuads * kOwnSlots;      // pixels per band
// hits per chunk: their grad_out rows live in LDS (128 B per fp32 row: 768 hits = 96 KiB; 64 B per 16-bit row: one hit
// per thread = 64 KiB)
template <typename T> constexpr int own_chunk() { return sizeof(T) == 4 ? 768 : kOwnThreads; }
template <typename T> constexpr int own_list() { return own_chunk<T>() + 4 * kOwnThreads; }   // a chunk's worth + one cull batch, worst case
constexpr unsigned kOwnNil = 0xffffffffu;
template <typename T> constexpr int own_lds_bytes()      // (+ 32: the entries start on a 32-byte boundary)
{
    return own_chunk<T>() * 32 * (int)sizeof(T) + 32 + 4 * own_chunk<T>() * 8 + kOwnPix * 4 + own_list<T>() * 4;
}

// (x, y) of one sampling point: one 8- / 4-byte load
__device__ __forceinline__ void load_xy(const float *loc2, float &x, float &y)
{
    const float2 v = *reinterpret_cast<const float2 *>(loc2);
    x = v.x; y = v.y;
}
__device__ __forceinline__ void load_xy(const bf16_t *loc2, float &x, float &y)
{
    const unsigned v = *reinterpret_cast<const unsigned *>(loc2);
    x = __uint_as_float(v << 16); y = __uint_as_float(v & 0xffff0000u);
}
__device__ __forceinline__ void load_xy(const f16_t *loc2, float &x, float &y)
{
    const float2 v = __half22float2(*reinterpret_cast<const __half2 *>(loc2));
    x = v.x; y = v.y;
}

template <typename T, int VARIANT>       // VARIANT bit 1: the next chunk's hits are fetched before the walk (bit 0: unused)
__global__ void __launch_bounds__(kOwnThreads)
msda_bwd_value_own_kernel(const Params p, int dbg)
{
    constexpr int D = 32, kRowB = D * (int)sizeof(T);          // bytes of one staged grad_out row
    constexpr int kOwnChunk = own_chunk<T>();
    constexpr bool kHalf = sizeof(T) == 2;
    extern __shared__ __attribute__((aligned(128))) unsigned char lds_raw[];
    unsigned char *rows = lds_raw;                                              // [kOwnChunk][kRowB]
    // [4 * kOwnChunk] {weight bits, next reference}, on a 32-byte boundary; a reference = the absolute LDS address of an entry
    uint2 *ents = reinterpret_cast<uint2 *>(lds_raw + kOwnChunk * kRowB +
                                            ((32u - (lds_addr(lds_raw) & 31u)) & 31u));
    unsigned *head = reinterpret_cast<unsigned *>(ents + 4 * kOwnChunk);        // [kOwnPix]
    unsigned *list = head + kOwnPix;                                            // [own_list<T>()] (k:6 | pt:2 | q:24)
    __shared__ int s_H[kScatterMaxLevels], s_W[kScatterMaxLevels], s_R[kScatterMaxLevels],
        s_first[kScatterMaxLevels + 1], s_lsi[kScatterMaxLevels];
    __shared__ int s_nsrc, s_cnt[3];     // survivor counters rotate: slot j is reset two barriers before it is used again
    __shared__ long long s_src_tab[kScatterMaxSources], s_src_loc[kScatterMaxSources];
    __shared__ int s_src_q0[kScatterMaxSources], s_src_gmv[kScatterMaxSources];
    __shared__ unsigned s_live[kLiveWords];            // bitmap of the cull batches that hold a live 64-query block
    __shared__ long long s_item;

    const int tid = threadIdx.x, lane = tid % kWave;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int MD = p.M * D, L = p.L, VL = p.LA + p.LB;
    if (tid == 0) {
        int first = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)p.shapes[2 * l], W = (int)p.shapes[2 * l + 1];
            const int R = min(H, kOwnPix / max(1, W));          // rows per band; 0 = "direct" level (row wider than a band)
            s_H[l] = H; s_W[l] = W; s_R[l] = R; s_lsi[l] = (int)p.lsi[l];
            s_first[l] = first;
            first += (R > 0) ? (H + R - 1) / R : 1;
        }
        s_first[L] = first;
        s_cnt[0] = s_cnt[1] = s_cnt[2] = 0;
    }
    int ci = 0;                             // counter of the current cull batch
    for (int i = tid; i < kOwnPix; i += kOwnThreads) head[i] = kOwnNil;
    __syncthreads();
    const int NB = s_first[L];
    const int clips = p.groups / p.frames;
    const int64_t n_items = (int64_t)clips * p.frames * p.M * NB;
    const bool dynamic = p.workspace != nullptr && (dbg & 16) == 0 && n_items < (int64_t)16 * gridDim.x;
    const int lane8 = blockIdx.x % 8;
    const int strideA = p.M * p.LA * p.PA, strideB = p.M * p.LB * p.PB;      // loc/attn elements per query
    // owner side: quad Q owns pixels s * kOwnQuads + Q of the band.  4-byte types: lane c of the quad holds the
    // channels [4c, 4c+4) of both 64-byte halves of the row (odd quads read the second half first: LDS banks, as
    // in the forward); 2-byte types: the 8 channels [8c, 8c+8) = one 16-byte slice of the 64-byte row.
    const int Q = tid / 4, cq = tid & 3, hsw = Q & 1;
    const int off1 = kHalf ? cq * 16 : cq * 16 + hsw * 64;
    const int ch1 = kHalf ? cq * 8 : off1 / 4, ch2 = kHalf ? cq * 8 + 4 : (off1 ^ 64) / 4;     // channels of acc[0..3] / acc[4..7]
    const unsigned ents_lds = lds_addr(ents);

    for (int64_t it = blockIdx.x;; it += gridDim.x) {
        int64_t item = it;
        if (dynamic) {
            if (tid == 0) s_item = (long long)atomicAdd(p.workspace + lane8, 1u) * 8 + lane8;
            __syncthreads();
            item = s_item;
        }
        if (item >= n_items) break;
        int l, part, m, f, clip;
        if (dynamic) {      // heaviest first: levels from the last to the first (see msda_bwd_value_lds_kernel)
            const int64_t ctm = (int64_t)clips * p.frames * p.M;
            l = L - 1;
            int64_t local = item;
            while (l > 0 && local >= ctm * (s_first[l + 1] - s_first[l])) {
                local -= ctm * (s_first[l + 1] - s_first[l]);
                --l;
            }
            const int nb_l = s_first[l + 1] - s_first[l];
            m = (int)(local % p.M);
            int64_t rest = local / p.M;
            part = s_first[l] + (int)(rest % nb_l); rest /= nb_l;
            f = (int)(rest % p.frames);
            clip = (int)(rest / p.frames);
        } else {
            m = (int)(item % p.M);
            int64_t rest = item / p.M;
            part = (int)(rest % NB); rest /= NB;
            f = (int)(rest % p.frames);
            clip = (int)(rest / p.frames);
            l = 0;
            while (l + 1 < L && s_first[l + 1] <= part) ++l;
        }
        const int H = s_H[l], W = s_W[l], R = s_R[l];
        const bool direct = (R == 0);
        const int r0 = direct ? 0 : (part - s_first[l]) * R;
        const int r1 = direct ? H - 1 : min(H, r0 + R) - 1;
        const int npix = direct ? 0 : (r1 - r0 + 1) * W;
        // Small bands (the last pyramid levels: 60 pixels at 360x640) would keep only npix of the 256 owner quads busy
        // while every pixel's list is long; their hits are dealt round-robin to SF sub-lists per pixel ("virtual
        // pixels" pix * SF + hit % SF), each with an owner quad of its own, and the SF partial sums of a pixel are
        // added up through LDS when the item is finished.  SF = largest power of two with npix * SF <= 256 quads.
        int sfs = 0;
        while (npix > 0 && (npix << (sfs + 1)) <= kOwnQuads && sfs < 4) ++sfs;
        const int SF = 1 << sfs, nvpix = npix << sfs;
        float *gmap = static_cast<float *>(p.grad_value) +
                      (((int64_t)clip * p.frames + f) * p.S + s_lsi[l]) * MD + m * D;     // pixel (0, 0) of the level, head m

        // sources that read frame f: the current-frame points of frame f, then every temporal slot (t, w) with
        // frame_table[t, w] == f; per source the first culling-table entry, first loc/attn element, first query row
        if (wave == 0) {
            const int n_tw = p.frames * p.window;
            const bool hit = lane < n_tw && p.ftab[lane] == f;
            const u64 bal = __ballot(hit);
            if (lane == 0) {
                const int64_t g = (int64_t)clip * p.frames + f;
                s_src_tab[0] = ((g * p.M + m) * VL + l) * p.Lq;
                s_src_loc[0] = (g * p.Lq * p.M + m) * ((int64_t)p.LA * p.PA) + l * p.PA;
                s_src_q0[0] = (int)(g * p.Lq);
                s_src_gmv[0] = (int)((g * p.M + m) * VL + l);
                s_nsrc = 1 + (int)__popcll(bal);
            }
            if (hit) {
                const int n = 1 + (int)__popcll(bal & ((1ull << lane) - 1ull)), t = lane / p.window;
                const int vl = (lane - t * p.window) * L + l;
                const int64_t g = (int64_t)clip * p.frames + t;
                s_src_tab[n] = ((g * p.M + m) * VL + p.LA + vl) * p.Lq;
                s_src_loc[n] = (g * p.Lq * p.M + m) * ((int64_t)p.LB * p.PB) + vl * p.PB;
                s_src_q0[n] = (int)(g * p.Lq);
                s_src_gmv[n] = (int)((g * p.M + m) * VL + p.LA + vl);
            }
        }
        __syncthreads();
        const int ng = s_nsrc * p.Lq;              // candidate groups: (source, query) pairs, <= 4 points each

        float acc[kOwnSlots][8];
#pragma unroll
        for (int s = 0; s < kOwnSlots; ++s)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[s][c] = 0.f;

        // Stage A of a chunk: the hit of this thread -- survivor entry -> (x, y, attention weight) loads in flight.
        auto fetch_hit = [&](int base, int n, float &x, float &y, float &a, int &qrow) {
            x = y = -10.f; a = 0.f; qrow = 0;
            if (tid < n) {
                const unsigned e = list[base + tid];
                const int k = (int)(e >> 26), q = (int)(e & 0xffffffu);
                const bool curf = (k == 0);
                const int64_t idx = s_src_loc[k] + (int64_t)q * (curf ? strideA : strideB) + (int)((e >> 24) & 3u);
                const T *loc = static_cast<const T *>(curf ? p.locA : p.locB);
                const T *aw = static_cast<const T *>(curf ? p.awA : p.awB);
                load_xy(loc + 2 * idx, x, y);
                a = Store<T>::get(aw + idx);
                qrow = s_src_q0[k] + q;
            }
        };
        // One chunk of n hits (already fetched into x, y, a, qrow): rows staged, entries linked, lists walked by
        // the owners; the NEXT chunk's hits (list[nbase, nbase + nn)) are fetched before the walk, so their memory
        // latency hides behind it.
        auto process_chunk = [&](int n, float &x, float &y, float &a, int &qrow, int nbase, int nn) {
            // ---- grad_out rows -> LDS: kRowB / 16 lanes x 16 B per hit, 8 (16) hits per LDS-DMA wave instruction
            if (!direct && !(dbg & 4)) {
                constexpr int LPR = kRowB / 16, HPI = kWave / LPR;      // lanes per row, hits per instruction
                const T *go = static_cast<const T *>(p.grad_out) + m * D + (lane % LPR) * (16 / (int)sizeof(T));
#pragma unroll
                for (int i = 0; i < kWave / HPI; ++i) {
                    const int src_lane = HPI * i + lane / LPR;
                    const int qr = __shfl(qrow, src_lane, kWave);
                    const int h = wave * kWave + src_lane;
                    if (wave * kWave + HPI * i < n) {                   // uniform: this instruction has at least one live hit
                        const T *gp = go + (int64_t)(h < n ? qr : 0) * MD;
#if defined(__HIP_DEVICE_COMPILE__)
                        __builtin_amdgcn_global_load_lds(gp, (__attribute__((address_space(3))) void *)(rows + (wave * kWave + HPI * i) * kRowB), 16, 0, 0);
#else
                        (void)gp;
#endif
                    }
                }
            }
            // ---- taps (cuh:285-288, 38-80) and the entries of the corners this band owns
            {
                const float h_im = __fsub_rn(__fmul_rn(y, (float)H), 0.5f);
                const float w_im = __fsub_rn(__fmul_rn(x, (float)W), 0.5f);
                if (tid < n && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                    const float hf = floorf(h_im), wf = floorf(w_im);
                    const int h_low = (int)hf, w_low = (int)wf;
                    const bool top = h_low >= max(r0, 0) && h_low <= r1;          // rows this band owns
                    const bool bot = h_low + 1 >= r0 && h_low + 1 <= min(r1, H - 1);
                    const bool x0 = w_low >= 0, x1 = w_low + 1 <= W - 1;
                    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
                    const float wgt[4] = {hh * hw * a, hh * lw * a, lh * hw * a, lh * lw * a};
                    const bool own[4] = {top && x0, top && x1, bot && x0, bot && x1};
                    const int pix00 = (h_low - r0) * W + w_low;
                    const int dpix[4] = {0, 1, W, W + 1};
                    if (direct) {
                        // a level whose single row does not fit a band: float atomics straight to memory
                        const T *gr = static_cast<const T *>(p.grad_out) + (int64_t)qrow * MD + m * D;
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (own[c]) {
                                float *dst = gmap + (int64_t)(pix00 + dpix[c]) * MD;
                                for (int ch = 0; ch < D; ++ch) atomic_accumulate(dst + ch, wgt[c] * Store<T>::get(gr + ch));
                            }
                    } else if (!(dbg & 2)) {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (own[c]) {
                                const unsigned ei = 4u * (unsigned)tid + (unsigned)c;
                                const unsigned prev = atomicExch(&head[((pix00 + dpix[c]) << sfs) + (tid & (SF - 1))], ents_lds + 8u * ei);
                                ents[ei] = make_uint2(__float_as_uint(wgt[c]), prev);
                            }
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's rows have landed
            __syncthreads();
            if constexpr (VARIANT & 2) fetch_hit(nbase, nn, x, y, a, qrow);      // (nn = 0: nothing)
            // ---- owners walk their pixels' lists
            if (!direct && !(dbg & 1)) {
#pragma unroll
                for (int s = 0; s < kOwnSlots; ++s) {
                    const int pix = s * kOwnQuads + Q;
                    unsigned e = kOwnNil;
                    if (pix < nvpix) { e = head[pix]; if (e != kOwnNil) head[pix] = kOwnNil; }
                    // (list references are absolute LDS addresses of the entries: the entry read needs no address
                    // arithmetic, the row is (A - entries) >> 5 since entries are 8 bytes, 4 per hit)
                    while (e != kOwnNil) {
                        const uint2 en = *reinterpret_cast<const uint2 *>(lds_raw + (e - lds_addr(lds_raw)));
                        const float w = __uint_as_float(en.x);
                        const unsigned char *r = rows + ((e - ents_lds) >> 5) * kRowB;
                        float v[8];
                        if constexpr (kHalf) {
                            Store<T>::load(reinterpret_cast<const T *>(r + off1), v);
                        } else {
                            const float4 v1 = *reinterpret_cast<const float4 *>(r + off1);
                            const float4 v2 = *reinterpret_cast<const float4 *>(r + (off1 ^ 64));
                            v[0] = v1.x; v[1] = v1.y; v[2] = v1.z; v[3] = v1.w; v[4] = v2.x; v[5] = v2.y; v[6] = v2.z; v[7] = v2.w;
                        }
#pragma unroll
                        for (int c = 0; c < 8; ++c) acc[s][c] = fmaf(w, v[c], acc[s][c]);
                        e = en.y;
                    }
                }
            }
            __syncthreads();
        };

        // ---- cull the candidate groups in batches of one per thread against the band; chunks are cut from the END
        // of the survivor list, so nothing has to move
        int listed = 0;
        const int lo = min(r0 - 1, 32767), hi = min(r1, 32767);
        auto load_records = [&](int gi0, int2 &iv, unsigned &ent, bool &live) {
            const int gi = gi0 + tid;
            live = gi < ng;
            iv = make_int2((int)0x80008000u, (int)0x80008000u);
            ent = 0u;
            if (live) {
                const int k = gi / p.Lq, q = gi - k * p.Lq;
                ent = ((unsigned)k << 26) | (unsigned)q;
                if (p.bbox) iv = *reinterpret_cast<const int2 *>(p.bbox + (s_src_tab[k] + q) * 2);
            }
        };
        // Long candidate ranges (encoder shapes, Lq = S): a pre-pass over the 64-query block summaries marks the cull
        // batches that hold a block whose tap rows can reach the band; with local sampling all but a few are skipped.
        const int nbat = (ng + kOwnThreads - 1) / kOwnThreads;
        const bool skipping = p.bsum != nullptr && nbat > 4 && nbat <= 32 * kLiveWords;
        if (skipping) {
            if (tid < kLiveWords) s_live[tid] = 0u;
            __syncthreads();
            const int nblk = (p.Lq + kCullBlock - 1) / kCullBlock, nb_tot = s_nsrc * nblk;
            for (int bk = tid; bk < nb_tot; bk += kOwnThreads) {
                const int ks = bk / nblk, blk = bk - ks * nblk;
                const int2 mm = *reinterpret_cast<const int2 *>(p.bsum + ((int64_t)s_src_gmv[ks] * nblk + blk) * 2);
                if (mm.y >= lo && mm.x <= hi) {
                    const int g0 = ks * p.Lq + blk * kCullBlock, g1 = min(g0 + kCullBlock, ks * p.Lq + p.Lq) - 1;
                    atomicOr(&s_live[(g0 / kOwnThreads) >> 5], 1u << ((g0 / kOwnThreads) & 31));
                    atomicOr(&s_live[(g1 / kOwnThreads) >> 5], 1u << ((g1 / kOwnThreads) & 31));
                }
            }
            __syncthreads();
        }
        auto next_live = [&](int bq) {       // first batch >= bq worth culling (nbat if none); workgroup-uniform
            if (!skipping) return min(bq, nbat);
            while (bq < nbat) {
                const unsigned wv = s_live[bq >> 5] >> (bq & 31);
                if (wv) return min(bq + (int)__builtin_ctz(wv), nbat);
                bq = (bq | 31) + 1;
            }
            return nbat;
        };
        int2 iv;
        unsigned ent;
        bool live;
        int bcur = next_live(0);
        if (bcur < nbat) load_records(bcur * kOwnThreads, iv, ent, live);
        while (bcur < nbat) {
            const int bnext = next_live(bcur + 1);
            unsigned pm = 0u;
            if (live) {
                if (p.bbox) {
                    const int hr[4] = {(int)(short)(iv.x & 0xffff), iv.x >> 16, (int)(short)(iv.y & 0xffff), iv.y >> 16};
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp) pm |= (hr[jp] >= lo && hr[jp] <= hi) ? (1u << jp) : 0u;
                } else {
                    pm = (1u << ((ent >> 26) == 0u ? p.PA : p.PB)) - 1u;       // no culling table: every point is a candidate
                }
            }
            const unsigned ent_now = ent;
            const bool last = bnext >= nbat;
            if (!last) load_records(bnext * kOwnThreads, iv, ent, live);      // the next live batch's records fly meanwhile
            // wave-wide exclusive scan of the per-lane survivor counts (DPP), one LDS atomic per wave
            const int cnt = __popc(pm);
            int v = cnt;
            v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
            v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
            v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
            v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
            v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
            v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
            const int total = __builtin_amdgcn_readlane(v, kWave - 1);
            int wbase = 0;
            if (tid == 0) s_cnt[(ci + 1) % 3] = 0;      // last read before the previous barrier, next used after the next one
            if (lane == 0 && total) wbase = atomicAdd(&s_cnt[ci], total);
            wbase = __shfl(wbase, 0, kWave);
            int pos = listed + wbase + v - cnt;
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if ((pm >> b) & 1u) { list[pos] = ent_now | ((unsigned)b << 24); ++pos; }
            __syncthreads();
            listed += s_cnt[ci];
            ci = (ci + 1) % 3;
            float x, y, a;
            int qrow;
            bool primed = false;
            if (dbg & 8) listed = 0;                    // measurement: cull only
            while (listed >= kOwnChunk || (last && listed > 0)) {
                const int n = min(kOwnChunk, listed);
                if (!primed || !(VARIANT & 2)) fetch_hit(listed - n, n, x, y, a, qrow);
                listed -= n;
                const bool more = listed >= kOwnChunk || (last && listed > 0);
                const int nn = more ? min(kOwnChunk, listed) : 0;
                process_chunk(n, x, y, a, qrow, listed - nn, nn);
                primed = more;
            }
            bcur = bnext;
        }
        // ---- owners store their pixels: grad_value is overwritten, every pixel of the band exactly once
        if (!direct && SF == 1) {
            float *gband = gmap + (int64_t)r0 * W * MD;
#pragma unroll
            for (int s = 0; s < kOwnSlots; ++s) {
                const int pix = s * kOwnQuads + Q;
                if (pix < npix) {
                    float *o = gband + (int64_t)pix * MD;
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store((f32x4){acc[s][0], acc[s][1], acc[s][2], acc[s][3]}, reinterpret_cast<f32x4 *>(o + ch1));
                    __builtin_nontemporal_store((f32x4){acc[s][4], acc[s][5], acc[s][6], acc[s][7]}, reinterpret_cast<f32x4 *>(o + ch2));
                }
            }
        } else if (!direct) {
            // split lists: the partial sums of virtual pixel v = pix * SF + sub (slot 0 of quad v) go through the (now
            // free) row area as [v][32 channels] floats; one thread per (pixel, 4 channels) adds the SF partials
            float *part = reinterpret_cast<float *>(rows);
            if (Q < nvpix) {
                *reinterpret_cast<float4 *>(part + Q * D + ch1) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
                *reinterpret_cast<float4 *>(part + Q * D + ch2) = make_float4(acc[0][4], acc[0][5], acc[0][6], acc[0][7]);
            }
            __syncthreads();
            float *gband = gmap + (int64_t)r0 * W * MD;
            for (int i = tid; i < npix * (D / 4); i += kOwnThreads) {
                const int pix = i / (D / 4), c4 = (i - pix * (D / 4)) * 4;
                float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int u = 0; u < SF; ++u) {
                    const float4 t4 = *reinterpret_cast<const float4 *>(part + ((pix << sfs) + u) * D + c4);
                    sum.x += t4.x; sum.y += t4.y; sum.z += t4.z; sum.w += t4.w;
                }
                *reinterpret_cast<float4 *>(gband + (int64_t)pix * MD + c4) = sum;
            }
        }
        __syncthreads();
    }
}

// ---- group-granular variant -------------------------------------------------------------------------------------
// A chunk is kGrpChunk (row, level) GROUPS -- the <= 4 sampling points one query puts on one level of one source frame --
// instead of 768 single points: the points of a group share their grad_out row, so the row is staged ONCE per group
// (the owner kernel above stages it once per point: 22 M 128-byte LDS-DMA requests per launch, 3/4 of them duplicates
// on the levels that are a single band), a chunk holds up to 4 x 512 = 2048 hits (fewer barriers and exposed latencies
// per hit, longer lists = better lock-step efficiency of the walk), and the survivor list holds groups (1 entry per
// cull thread, 6 KiB instead of 19).  Thread t of pass j handles point (t & 3) of group 256 j + t / 4; a group's 16
// entries are one 128-byte block, so the row of an entry at LDS address A is (A - entries) >> 7.
constexpr int kGrpList = 3 * kOwnThreads;       // survivor list entries (groups)
template <typename T> constexpr int grp_chunk() { return sizeof(T) == 4 ? MSDA_GRP_F32 : MSDA_GRP_16; }       // groups per chunk
template <typename T> constexpr int grp_lds_bytes()
{
    return grp_chunk<T>() * 32 * (int)sizeof(T) + 32 + 16 * grp_chunk<T>() * 8 + kOwnPix * 4 + kGrpList * 4;
}

template <typename T>
__global__ void __launch_bounds__(kOwnThreads)
msda_bwd_value_grp_kernel(const Params p, int dbg)
{
    constexpr int D = 32, kRowB = D * (int)sizeof(T);          // bytes of one staged grad_out row
    constexpr int kOwnChunk = grp_chunk<T>();                   // groups per chunk
    constexpr int kPasses = (4 * kOwnChunk + kOwnThreads - 1) / kOwnThreads;
    constexpr bool kHalf = sizeof(T) == 2;
    extern __shared__ __attribute__((aligned(128))) unsigned char lds_raw[];
    unsigned char *rows = lds_raw;                                              // [kOwnChunk][kRowB]  one row per group
    // [16 * kOwnChunk] {weight bits, next reference}, on a 32-byte boundary; a reference = the absolute LDS address of an entry
    uint2 *ents = reinterpret_cast<uint2 *>(lds_raw + kOwnChunk * kRowB +
                                            ((32u - (lds_addr(lds_raw) & 31u)) & 31u));
    unsigned *head = reinterpret_cast<unsigned *>(ents + 16 * kOwnChunk);       // [kOwnPix]
    unsigned *list = head + kOwnPix;                                            // [kGrpList] (k:6 | points:4 | q:22)
    __shared__ int s_H[kScatterMaxLevels], s_W[kScatterMaxLevels], s_R[kScatterMaxLevels],
        s_first[kScatterMaxLevels + 1], s_lsi[kScatterMaxLevels];
    __shared__ int s_nsrc, s_cnt[3];     // survivor counters rotate: slot j is reset two barriers before it is used again
    __shared__ long long s_src_tab[kScatterMaxSources], s_src_loc[kScatterMaxSources];
    __shared__ int s_src_q0[kScatterMaxSources], s_src_gmv[kScatterMaxSources];
    __shared__ unsigned s_live[kLiveWords];            // bitmap of the cull batches that hold a live 64-query block
    __shared__ long long s_item;

    const int tid = threadIdx.x, lane = tid % kWave;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int MD = p.M * D, L = p.L, VL = p.LA + p.LB;
    if (tid == 0) {
        int first = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)p.shapes[2 * l], W = (int)p.shapes[2 * l + 1];
            const int R = min(H, kOwnPix / max(1, W));          // rows per band; 0 = "direct" level (row wider than a band)
            s_H[l] = H; s_W[l] = W; s_R[l] = R; s_lsi[l] = (int)p.lsi[l];
            s_first[l] = first;
            first += (R > 0) ? (H + R - 1) / R : 1;
        }
        s_first[L] = first;
        s_cnt[0] = s_cnt[1] = s_cnt[2] = 0;
    }
    int ci = 0;                             // counter of the current cull batch
    for (int i = tid; i < kOwnPix; i += kOwnThreads) head[i] = kOwnNil;
    __syncthreads();
    const int NB = s_first[L];
    const int clips = p.groups / p.frames;
    const int64_t n_items = (int64_t)clips * p.frames * p.M * NB;
    const bool dynamic = p.workspace != nullptr && (dbg & 16) == 0 && n_items < (int64_t)16 * gridDim.x;
    const int lane8 = blockIdx.x % 8;
    const int strideA = p.M * p.LA * p.PA, strideB = p.M * p.LB * p.PB;      // loc/attn elements per query
    // owner side: quad Q owns pixels s * kOwnQuads + Q of the band.  4-byte types: lane c of the quad holds the
    // channels [4c, 4c+4) of both 64-byte halves of the row (odd quads read the second half first: LDS banks, as
    // in the forward); 2-byte types: the 8 channels [8c, 8c+8) = one 16-byte slice of the 64-byte row.
    const int Q = tid / 4, cq = tid & 3, hsw = Q & 1;
    const int off1 = kHalf ? cq * 16 : cq * 16 + hsw * 64;
    const int ch1 = kHalf ? cq * 8 : off1 / 4, ch2 = kHalf ? cq * 8 + 4 : (off1 ^ 64) / 4;     // channels of acc[0..3] / acc[4..7]
    const unsigned ents_lds = lds_addr(ents);

    for (int64_t it = blockIdx.x;; it += gridDim.x) {
        int64_t item = it;
        if (dynamic) {
            if (tid == 0) s_item = (long long)atomicAdd(p.workspace + lane8, 1u) * 8 + lane8;
            __syncthreads();
            item = s_item;
        }
        if (item >= n_items) break;
        int l, part, m, f, clip;
        if (dynamic) {      // heaviest first: levels from the last to the first (see msda_bwd_value_lds_kernel)
            const int64_t ctm = (int64_t)clips * p.frames * p.M;
            l = L - 1;
            int64_t local = item;
            while (l > 0 && local >= ctm * (s_first[l + 1] - s_first[l])) {
                local -= ctm * (s_first[l + 1] - s_first[l]);
                --l;
            }
            const int nb_l = s_first[l + 1] - s_first[l];
            m = (int)(local % p.M);
            int64_t rest = local / p.M;
            part = s_first[l] + (int)(rest % nb_l); rest /= nb_l;
            f = (int)(rest % p.frames);
            clip = (int)(rest / p.frames);
        } else {
            m = (int)(item % p.M);
            int64_t rest = item / p.M;
            part = (int)(rest % NB); rest /= NB;
            f = (int)(rest % p.frames);
            clip = (int)(rest / p.frames);
            l = 0;
            while (l + 1 < L && s_first[l + 1] <= part) ++l;
        }
        const int H = s_H[l], W = s_W[l], R = s_R[l];
        const bool direct = (R == 0);
        const int r0 = direct ? 0 : (part - s_first[l]) * R;
        const int r1 = direct ? H - 1 : min(H, r0 + R) - 1;
        const int npix = direct ? 0 : (r1 - r0 + 1) * W;
        // Small bands (the last pyramid levels: 60 pixels at 360x640) would keep only npix of the 256 owner quads busy
        // while every pixel's list is long; their hits are dealt round-robin to SF sub-lists per pixel ("virtual
        // pixels" pix * SF + hit % SF), each with an owner quad of its own, and the SF partial sums of a pixel are
        // added up through LDS when the item is finished.  SF = largest power of two with npix * SF <= 256 quads.
        int sfs = 0;
        while (npix > 0 && (npix << (sfs + 1)) <= kOwnQuads && sfs < 4) ++sfs;
        const int SF = 1 << sfs, nvpix = npix << sfs;
        float *gmap = static_cast<float *>(p.grad_value) +
                      (((int64_t)clip * p.frames + f) * p.S + s_lsi[l]) * MD + m * D;     // pixel (0, 0) of the level, head m

        // sources that read frame f: the current-frame points of frame f, then every temporal slot (t, w) with
        // frame_table[t, w] == f; per source the first culling-table entry, first loc/attn element, first query row
        if (wave == 0) {
            const int n_tw = p.frames * p.window;
            const bool hit = lane < n_tw && p.ftab[lane] == f;
            const u64 bal = __ballot(hit);
            if (lane == 0) {
                const int64_t g = (int64_t)clip * p.frames + f;
                s_src_tab[0] = ((g * p.M + m) * VL + l) * p.Lq;
                s_src_loc[0] = (g * p.Lq * p.M + m) * ((int64_t)p.LA * p.PA) + l * p.PA;
                s_src_q0[0] = (int)(g * p.Lq);
                s_src_gmv[0] = (int)((g * p.M + m) * VL + l);
                s_nsrc = 1 + (int)__popcll(bal);
            }
            if (hit) {
                const int n = 1 + (int)__popcll(bal & ((1ull << lane) - 1ull)), t = lane / p.window;
                const int vl = (lane - t * p.window) * L + l;
                const int64_t g = (int64_t)clip * p.frames + t;
                s_src_tab[n] = ((g * p.M + m) * VL + p.LA + vl) * p.Lq;
                s_src_loc[n] = (g * p.Lq * p.M + m) * ((int64_t)p.LB * p.PB) + vl * p.PB;
                s_src_q0[n] = (int)(g * p.Lq);
                s_src_gmv[n] = (int)((g * p.M + m) * VL + p.LA + vl);
            }
        }
        __syncthreads();
        const int ng = s_nsrc * p.Lq;              // candidate groups: (source, query) pairs, <= 4 points each

        float acc[kOwnSlots][8];
#pragma unroll
        for (int s = 0; s < kOwnSlots; ++s)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[s][c] = 0.f;

        // The hit of this thread in pass j of a chunk: point (tid & 3) of group 256 j + tid / 4 if it survived the cull;
        // its (x, y, attention weight) loads are issued here (the 4 threads of a group read 32 + 16 contiguous bytes).
        auto fetch_hit = [&](int base, int n, int pass, float &x, float &y, float &a, int &qrow, bool &act) {
            x = y = -10.f; a = 0.f; qrow = 0; act = false;
            const int g = pass * (kOwnThreads / 4) + tid / 4, pt = tid & 3;
            if (g < n) {
                const unsigned e = list[base + g];
                const int k = (int)(e >> 26), q = (int)(e & 0x3fffffu);
                if ((e >> (22 + pt)) & 1u) {
                    const bool curf = (k == 0);
                    const int64_t idx = s_src_loc[k] + (int64_t)q * (curf ? strideA : strideB) + pt;
                    const T *loc = static_cast<const T *>(curf ? p.locA : p.locB);
                    const T *aw = static_cast<const T *>(curf ? p.awA : p.awB);
                    load_xy(loc + 2 * idx, x, y);
                    a = Store<T>::get(aw + idx);
                    qrow = s_src_q0[k] + q;
                    act = true;
                }
            }
        };
        // ---- grad_out rows -> LDS, one per GROUP: wave w stages rows [RPWV w, RPWV (w + 1)) of the chunk
        auto stage_rows = [&](int base, int n) {
            if (direct || (dbg & 4)) return;
            constexpr int LPR = kRowB / 16, HPI = kWave / LPR;      // lanes per row, rows per instruction
            const T *go = static_cast<const T *>(p.grad_out) + m * D + (lane % LPR) * (16 / (int)sizeof(T));
            constexpr int RPWV = kOwnChunk / (kOwnThreads / kWave);   // rows per wave
            static_assert(kOwnChunk % (kOwnThreads / kWave) == 0 && RPWV % HPI == 0, "rows per wave");
#pragma unroll
            for (int i = 0; i < RPWV / HPI; ++i) {
                const int r0w = wave * RPWV + HPI * i;
                if (r0w < n) {                                      // uniform: this instruction has at least one live row
                    const unsigned e = list[base + min(r0w + lane / LPR, n - 1)];
                    const int qr = s_src_q0[e >> 26] + (int)(e & 0x3fffffu);
                    const T *gp = go + (int64_t)qr * MD;
#if defined(__HIP_DEVICE_COMPILE__)
                    __builtin_amdgcn_global_load_lds(gp, (__attribute__((address_space(3))) void *)(rows + r0w * kRowB), 16, 0, 0);
#else
                    (void)gp;
#endif
                }
            }
        };
        // ---- taps (cuh:285-288, 38-80) and the entries of the corners this band owns
        auto taps_link = [&](int pass, bool act, float x, float y, float a, int qrow) {
            const float h_im = __fsub_rn(__fmul_rn(y, (float)H), 0.5f);
            const float w_im = __fsub_rn(__fmul_rn(x, (float)W), 0.5f);
            if (act && h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
                const float hf = floorf(h_im), wf = floorf(w_im);
                const int h_low = (int)hf, w_low = (int)wf;
                const bool top = h_low >= max(r0, 0) && h_low <= r1;          // rows this band owns
                const bool bot = h_low + 1 >= r0 && h_low + 1 <= min(r1, H - 1);
                const bool x0 = w_low >= 0, x1 = w_low + 1 <= W - 1;
                const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
                const float wgt[4] = {hh * hw * a, hh * lw * a, lh * hw * a, lh * lw * a};
                const bool own[4] = {top && x0, top && x1, bot && x0, bot && x1};
                const int pix00 = (h_low - r0) * W + w_low;
                const int dpix[4] = {0, 1, W, W + 1};
                if (direct) {
                    // a level whose single row does not fit a band: float atomics straight to memory
                    const T *gr = static_cast<const T *>(p.grad_out) + (int64_t)qrow * MD + m * D;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (own[c]) {
                            float *dst = gmap + (int64_t)(pix00 + dpix[c]) * MD;
                            for (int ch = 0; ch < D; ++ch) atomic_accumulate(dst + ch, wgt[c] * Store<T>::get(gr + ch));
                        }
                } else if (!(dbg & 2)) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (own[c]) {
                            const unsigned ei = 4u * (unsigned)(pass * kOwnThreads + tid) + (unsigned)c;
                            const unsigned prev = atomicExch(&head[((pix00 + dpix[c]) << sfs) + (tid & (SF - 1))], ents_lds + 8u * ei);
                            ents[ei] = make_uint2(__float_as_uint(wgt[c]), prev);
                        }
                }
            }
        };
        // ---- owners walk their pixels' lists (references are absolute LDS addresses; the row is (A - entries) >> 7)
        auto walk = [&]() {
            if (direct || (dbg & 1)) return;
#pragma unroll
            for (int s = 0; s < kOwnSlots; ++s) {
                const int pix = s * kOwnQuads + Q;
                unsigned e = kOwnNil;
                if (pix < nvpix) { e = head[pix]; if (e != kOwnNil) head[pix] = kOwnNil; }
                while (e != kOwnNil) {
                    const uint2 en = *reinterpret_cast<const uint2 *>(lds_raw + (e - lds_addr(lds_raw)));
                    const float w = __uint_as_float(en.x);
                    const unsigned char *r = rows + ((e - ents_lds) >> 7) * kRowB;
                    float v[8];
                    if constexpr (kHalf) {
                        Store<T>::load(reinterpret_cast<const T *>(r + off1), v);
                    } else {
                        const float4 v1 = *reinterpret_cast<const float4 *>(r + off1);
                        const float4 v2 = *reinterpret_cast<const float4 *>(r + (off1 ^ 64));
                        v[0] = v1.x; v[1] = v1.y; v[2] = v1.z; v[3] = v1.w; v[4] = v2.x; v[5] = v2.y; v[6] = v2.z; v[7] = v2.w;
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[s][c] = fmaf(w, v[c], acc[s][c]);
                    e = en.y;
                }
            }
        };
        // One chunk of n groups: rows on their way, all passes' point loads issued (unless the previous chunk already did:
        // `primed`), entries linked, barrier, the NEXT chunk's point loads issued so that their memory latency hides behind
        // the walk, lists walked.
        float hx[kPasses], hy[kPasses], ha[kPasses];
        int hq[kPasses];
        bool hact[kPasses];
        auto fetch_chunk = [&](int base, int n) {
#pragma unroll
            for (int j = 0; j < kPasses; ++j) {
                hact[j] = false;
                if (j == 0 || n > j * (kOwnThreads / 4)) fetch_hit(base, n, j, hx[j], hy[j], ha[j], hq[j], hact[j]);
            }
        };
        auto process_chunk = [&](int base, int n, bool primed, int nbase, int nn) {
            stage_rows(base, n);
            if (!primed) fetch_chunk(base, n);
#pragma unroll
            for (int j = 0; j < kPasses; ++j)
                if (j == 0 || n > j * (kOwnThreads / 4)) taps_link(j, hact[j], hx[j], hy[j], ha[j], hq[j]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's rows have landed
            __syncthreads();
            if (nn > 0) fetch_chunk(nbase, nn);
            walk();
            __syncthreads();
        };

        // ---- cull the candidate groups in batches of one per thread against the band; chunks are cut from the END
        // of the survivor list, so nothing has to move
        int listed = 0;
        const int lo = min(r0 - 1, 32767), hi = min(r1, 32767);
        auto load_records = [&](int gi0, int2 &iv, unsigned &ent, bool &live) {
            const int gi = gi0 + tid;
            live = gi < ng;
            iv = make_int2((int)0x80008000u, (int)0x80008000u);
            ent = 0u;
            if (live) {
                const int k = gi / p.Lq, q = gi - k * p.Lq;
                ent = ((unsigned)k << 26) | (unsigned)q;              // (q < 2^22: host)
                if (p.bbox) iv = *reinterpret_cast<const int2 *>(p.bbox + (s_src_tab[k] + q) * 2);
            }
        };
        // Long candidate ranges (encoder shapes, Lq = S): a pre-pass over the 64-query block summaries marks the cull
        // batches that hold a block whose tap rows can reach the band; with local sampling all but a few are skipped.
        const int nbat = (ng + kOwnThreads - 1) / kOwnThreads;
        const bool skipping = p.bsum != nullptr && nbat > 4 && nbat <= 32 * kLiveWords;
        if (skipping) {
            if (tid < kLiveWords) s_live[tid] = 0u;
            __syncthreads();
            const int nblk = (p.Lq + kCullBlock - 1) / kCullBlock, nb_tot = s_nsrc * nblk;
            for (int bk = tid; bk < nb_tot; bk += kOwnThreads) {
                const int ks = bk / nblk, blk = bk - ks * nblk;
                const int2 mm = *reinterpret_cast<const int2 *>(p.bsum + ((int64_t)s_src_gmv[ks] * nblk + blk) * 2);
                if (mm.y >= lo && mm.x <= hi) {
                    const int g0 = ks * p.Lq + blk * kCullBlock, g1 = min(g0 + kCullBlock, ks * p.Lq + p.Lq) - 1;
                    atomicOr(&s_live[(g0 / kOwnThreads) >> 5], 1u << ((g0 / kOwnThreads) & 31));
                    atomicOr(&s_live[(g1 / kOwnThreads) >> 5], 1u << ((g1 / kOwnThreads) & 31));
                }
            }
            __syncthreads();
        }
        auto next_live = [&](int bq) {       // first batch >= bq worth culling (nbat if none); workgroup-uniform
            if (!skipping) return min(bq, nbat);
            while (bq < nbat) {
                const unsigned wv = s_live[bq >> 5] >> (bq & 31);
                if (wv) return min(bq + (int)__builtin_ctz(wv), nbat);
                bq = (bq | 31) + 1;
            }
            return nbat;
        };
        int2 iv;
        unsigned ent;
        bool live;
        int bcur = next_live(0);
        if (bcur < nbat) load_records(bcur * kOwnThreads, iv, ent, live);
        while (bcur < nbat) {
            const int bnext = next_live(bcur + 1);
            unsigned pm = 0u;
            if (live) {
                if (p.bbox) {
                    const int hr[4] = {(int)(short)(iv.x & 0xffff), iv.x >> 16, (int)(short)(iv.y & 0xffff), iv.y >> 16};
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp) pm |= (hr[jp] >= lo && hr[jp] <= hi) ? (1u << jp) : 0u;
                } else {
                    pm = (1u << ((ent >> 26) == 0u ? p.PA : p.PB)) - 1u;       // no culling table: every point is a candidate
                }
            }
            const unsigned ent_now = ent;
            const bool last = bnext >= nbat;
            if (!last) load_records(bnext * kOwnThreads, iv, ent, live);      // the next live batch's records fly meanwhile
            // wave-wide exclusive scan of the per-lane survivor flags (DPP), one LDS atomic per wave
            const int cnt = pm != 0u;
            int v = cnt;
            v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
            v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
            v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
            v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
            v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
            v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
            const int total = __builtin_amdgcn_readlane(v, kWave - 1);
            int wbase = 0;
            if (tid == 0) s_cnt[(ci + 1) % 3] = 0;      // last read before the previous barrier, next used after the next one
            if (lane == 0 && total) wbase = atomicAdd(&s_cnt[ci], total);
            wbase = __shfl(wbase, 0, kWave);
            if (cnt) list[listed + wbase + v - cnt] = ent_now | (pm << 22);
            __syncthreads();
            listed += s_cnt[ci];
            ci = (ci + 1) % 3;
            if (dbg & 8) listed = 0;                    // measurement: cull only
            // chunks are processed when the list could not take another cull batch, or at the end: an item of <= 2 batches
            // (every decoder call) is culled completely first and its chunks then run back to back, each one's point loads
            // issued under the previous one's walk
            bool primed = false;
            while (listed > kGrpList - kOwnThreads || (last && listed > 0)) {
                const int n = min(kOwnChunk, listed);
                listed -= n;
                const bool more = listed > kGrpList - kOwnThreads || (last && listed > 0);
                const int nn = more ? min(kOwnChunk, listed) : 0;
                process_chunk(listed, n, primed, listed - nn, nn);
                primed = more;
            }
            bcur = bnext;
        }
        // ---- owners store their pixels: grad_value is overwritten, every pixel of the band exactly once
        if (!direct && SF == 1) {
            float *gband = gmap + (int64_t)r0 * W * MD;
#pragma unroll
            for (int s = 0; s < kOwnSlots; ++s) {
                const int pix = s * kOwnQuads + Q;
                if (pix < npix) {
                    float *o = gband + (int64_t)pix * MD;
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store((f32x4){acc[s][0], acc[s][1], acc[s][2], acc[s][3]}, reinterpret_cast<f32x4 *>(o + ch1));
                    __builtin_nontemporal_store((f32x4){acc[s][4], acc[s][5], acc[s][6], acc[s][7]}, reinterpret_cast<f32x4 *>(o + ch2));
                }
            }
        } else if (!direct) {
            // split lists: the partial sums of virtual pixel v = pix * SF + sub (slot 0 of quad v) go through the (now
            // free) row area as [v][32 channels] floats; one thread per (pixel, 4 channels) adds the SF partials
            float *part = reinterpret_cast<float *>(rows);
            if (Q < nvpix) {
                *reinterpret_cast<float4 *>(part + Q * D + ch1) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
                *reinterpret_cast<float4 *>(part + Q * D + ch2) = make_float4(acc[0][4], acc[0][5], acc[0][6], acc[0][7]);
            }
            __syncthreads();
            float *gband = gmap + (int64_t)r0 * W * MD;
            for (int i = tid; i < npix * (D / 4); i += kOwnThreads) {
                const int pix = i / (D / 4), c4 = (i - pix * (D / 4)) * 4;
                float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int u = 0; u < SF; ++u) {
                    const float4 t4 = *reinterpret_cast<const float4 *>(part + ((pix << sfs) + u) * D + c4);
                    sum.x += t4.x; sum.y += t4.y; sum.z += t4.z; sum.w += t4.w;
                }
                *reinterpret_cast<float4 *>(gband + (int64_t)pix * MD + c4) = sum;
            }
        }
        __syncthreads();
    }
}

// The LDS scatter kernels OVERWRITE every pixel of a level whose row fits the band budget.  Pixels they
// do not own -- levels that take the float-atomic branch, or rows of `value` outside every level when
// spatial_shapes does not tile [0, S) -- are zero-filled here, so that callers need not memset grad_value.
__global__ void __launch_bounds__(256)
msda_zero_unowned_kernel(const Params p, int cap_slots)
{
    const int MD = p.M * p.D;
    const int64_t total = (int64_t)p.groups * p.S;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = idx % p.S;
        bool owned = false;
        for (int l = 0; l < p.L; ++l) {
            const int64_t H = p.shapes[2 * l], W = p.shapes[2 * l + 1], start = p.lsi[l];
            if (s >= start && s < start + H * W) { owned = cap_slots / max((int64_t)1, W * p.D) > 0; break; }
        }
        if (owned) continue;
        float *dst = static_cast<float *>(p.grad_value) + idx * MD;
        for (int c = 0; c < MD; ++c) dst[c] = 0.f;
    }
}

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
constexpr int kRsThreads = 1024, kRsWaves = kRsThreads / kWave;
constexpr int kRsRows = kWave / 4;       // rows per wave tile: one quad per row
constexpr int kRsSlack = 1024;          // bytes: the last LDS-DMA piece may overrun the slab's pixels
constexpr int kRsMaxFrames = 32;        // frames x frames slot masks live in LDS
constexpr int kRsRowB = 128;            // bytes of one pixel of one head in a 4-byte type (D = 32); 64 in a 2-byte type
constexpr int kRsTailBytes = kRsRowB + kRsMaxFrames * kRsMaxFrames * 4 + 4 * kSlabMaxLevels * 4 + 16;   // after the slab
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

// ------------------------------------------------------------------------------------------------
// generic kernels: any D / M / L / P, any dtype (fp64 included).  Correctness path for shapes the
// tile kernels do not take (D not a power-of-two multiple of the 16-B lane vector, fp64 gradcheck).
// ------------------------------------------------------------------------------------------------
template <typename A> struct GTaps { int64_t off[4]; A w[4]; A lh, lw; int valid; };

template <typename A>
__device__ __forceinline__ GTaps<A> make_gtaps(A x, A y, const Level lv, int MD)
{
    GTaps<A> t;
    for (int k = 0; k < 4; ++k) { t.off[k] = 0; t.w[k] = 0; }
    t.lh = t.lw = 0; t.valid = 0;
    const A h_im = y * (A)lv.H - (A)0.5, w_im = x * (A)lv.W - (A)0.5;
    if (h_im > -1 && w_im > -1 && h_im < lv.H && w_im < lv.W) {
        const A hf = floor(h_im), wf = floor(w_im);
        const int h_low = (int)hf, w_low = (int)wf, h_high = h_low + 1, w_high = w_low + 1;
        const A lh = h_im - hf, lw = w_im - wf, hh = 1 - lh, hw = 1 - lw;
        const bool y0 = h_low >= 0, y1 = h_high <= lv.H - 1, x0 = w_low >= 0, x1 = w_high <= lv.W - 1;
        const int64_t r0 = ((int64_t)lv.start + (int64_t)h_low * lv.W) * MD, r1 = r0 + (int64_t)lv.W * MD;
        const int64_t c0 = (int64_t)w_low * MD, c1 = c0 + MD;
        t.lh = lh; t.lw = lw;
        if (y0 && x0) { t.off[0] = r0 + c0; t.w[0] = hh * hw; t.valid |= 1; }
        if (y0 && x1) { t.off[1] = r0 + c1; t.w[1] = hh * lw; t.valid |= 2; }
        if (y1 && x0) { t.off[2] = r1 + c0; t.w[2] = lh * hw; t.valid |= 4; }
        if (y1 && x1) { t.off[3] = r1 + c1; t.w[3] = lh * lw; t.valid |= 8; }
    }
    return t;
}

template <typename T, typename A>
__global__ void __launch_bounds__(256)
msda_fwd_generic_kernel(const Params p, int64_t total)
{
    const int MD = p.M * p.D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % p.D);
        const int64_t row = i / p.D;                      // (group, q, m)
        const int m = (int)(row % p.M);
        const int group = (int)(row / ((int64_t)p.M * p.Lq));
        const int clip = group / p.frames, t = group - clip * p.frames;
        const T *value = static_cast<const T *>(p.value) + clip * p.v_clip + m * p.v_head + c;
        A acc = 0;
        for (int arr = 0; arr < 2; ++arr) {
            const T *loc = static_cast<const T *>(arr ? p.locB : p.locA);
            const T *aw = static_cast<const T *>(arr ? p.awB : p.awA);
            const int P = arr ? p.PB : p.PA, nl = arr ? p.LB : p.LA, LP = nl * P;
            for (int pt = 0; pt < LP; ++pt) {
                const Level lv = make_level(p, t, (arr ? p.LA : 0) + pt / P);
                const int64_t idx = row * LP + pt;
                const GTaps<A> tp = make_gtaps<A>((A)Store<T>::get(loc + 2 * idx),
                                                  (A)Store<T>::get(loc + 2 * idx + 1), lv, p.v_pix);
                if (!tp.valid) continue;
                A val = 0;
                for (int k = 0; k < 4; ++k)
                    if (tp.valid & (1 << k)) val += tp.w[k] * (A)Store<T>::get(value + tp.off[k]);
                acc += val * (A)Store<T>::get(aw + idx);
            }
        }
        Store<T>::put(static_cast<T *>(p.out) + i, acc);
    }
}

template <typename A>
__device__ __forceinline__ A wave_sum(A v)
{
#pragma unroll
    for (int s = 1; s < kWave; s <<= 1) v += __shfl_xor(v, s, kWave);
    return v;
}

// one wave per (group, q, m) row; lanes stride over the D channels
template <typename T, typename A>
__global__ void __launch_bounds__(kWave)
msda_bwd_generic_kernel(const Params p, int64_t rows)
{
    const int MD = p.M * p.D;
    const int lane = threadIdx.x;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const int m = (int)(row % p.M);
        const int group = (int)(row / ((int64_t)p.M * p.Lq));
        const int clip = group / p.frames, t = group - clip * p.frames;
        const T *value = static_cast<const T *>(p.value) + clip * p.v_clip + m * p.v_head;
        A *gvalue = static_cast<A *>(p.grad_value) + (int64_t)clip * p.frames * p.S * MD + m * p.D;
        const T *go = static_cast<const T *>(p.grad_out) + row * p.D;
        for (int arr = 0; arr < 2; ++arr) {
            const T *loc = static_cast<const T *>(arr ? p.locB : p.locA);
            const T *aw = static_cast<const T *>(arr ? p.awB : p.awA);
            T *gloc = static_cast<T *>(arr ? p.glocB : p.glocA);
            T *gaw = static_cast<T *>(arr ? p.gawB : p.gawA);
            const int P = arr ? p.PB : p.PA, nl = arr ? p.LB : p.LA, LP = nl * P;
            for (int pt = 0; pt < LP; ++pt) {
                const Level lv = make_level(p, t, (arr ? p.LA : 0) + pt / P);
                const int64_t idx = row * LP + pt;
                const A a = (A)Store<T>::get(aw + idx);
                // offsets in PIXELS: value and grad_value (always the standard layout) have different strides
                const GTaps<A> tp = make_gtaps<A>((A)Store<T>::get(loc + 2 * idx),
                                                  (A)Store<T>::get(loc + 2 * idx + 1), lv, 1);
                A d[4] = {0, 0, 0, 0};
                if (tp.valid) {
                    for (int c = lane; c < p.D; c += kWave) {
                        const A gc = (A)Store<T>::get(go + c);
                        for (int k = 0; k < 4; ++k) {
                            if (tp.valid & (1 << k)) {
                                d[k] += gc * (A)Store<T>::get(value + tp.off[k] * p.v_pix + c);
                                atomic_accumulate(gvalue + tp.off[k] * MD + c, tp.w[k] * a * gc);
                            }
                        }
                    }
                }
                for (int k = 0; k < 4; ++k) d[k] = wave_sum<A>(d[k]);
                if (lane == 0) {
                    const A hh = 1 - tp.lh, hw = 1 - tp.lw;
                    const A g_aw = tp.w[0] * d[0] + tp.w[1] * d[1] + tp.w[2] * d[2] + tp.w[3] * d[3];
                    const A g_w = hh * (d[1] - d[0]) + tp.lh * (d[3] - d[2]);
                    const A g_h = hw * (d[2] - d[0]) + tp.lw * (d[3] - d[1]);
                    Store<T>::put(gaw + idx, g_aw);
                    Store<T>::put(gloc + 2 * idx, (A)lv.W * g_w * a);
                    Store<T>::put(gloc + 2 * idx + 1, (A)lv.H * g_h * a);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pre-op fusion: the modules' chain between their Linears and the operator --
// cat(logits) -> softmax -> split -> reshape, and reference + offsets / normalizer (or the box form) for
// the current-frame and the temporal points (ref ms_deform_attn.py:105-121, 225-266, 327-352) -- as ONE
// pass over the Linear outputs that writes sampling_loc / attn_weight in the operator's layouts, and ONE
// pass back.  loc / attn are still materialised (the decoder returns them); what disappears are the ~10
// elementwise passes and copies in between.
// Mapping: one 32-lane half-wave per (row, head) walks that head's n = L*Pc + W*L*Pt points (lane j takes
// points j, j+32, ...); the joint softmax is two half-wave butterflies.
// ------------------------------------------------------------------------------------------------
struct PrepParams {
    const void *off_c, *off_t;        // [rows, M, L, Pc, 2], [rows, M, W*L, Pt, 2]   raw sampling offsets
    const void *logit_c, *logit_t;    // [rows, M, L*Pc], [rows, M, W*L*Pt]           raw attention logits
    const void *ref_c, *ref_t;        // [rows, L, d], [rows, W*L, d]                 reference points (d = 2 | 4)
    const int64_t *shapes;            // [L, 2] (H, W)
    void *loc_c, *loc_t, *aw_c, *aw_t;            // forward outputs (backward: aw_* are inputs)
    const void *gloc_c, *gloc_t, *gaw_c, *gaw_t;  // backward inputs
    void *goff_c, *goff_t, *glogit_c, *glogit_t;  // backward outputs
    int64_t rows;
    int64_t ld;                       // row stride of the Linear-side tensors (offsets / logits forward, their grads
                                      // backward) when they are column slices of one fused matrix; 0 = each dense
    int M, L, W, Pc, Pt, d;
};

template <typename A> __device__ __forceinline__ A half_wave_max(A v)
{
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { const A u = __shfl_xor(v, o, 32); v = u > v ? u : v; }
    return v;
}
template <typename A> __device__ __forceinline__ A half_wave_sum(A v)
{
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
    return v;
}
__device__ __forceinline__ float prep_exp(float x) { return expf(x); }
__device__ __forceinline__ double prep_exp(double x) { return exp(x); }

template <typename T, typename A, bool BWD>
__global__ void __launch_bounds__(256)
msda_prep_kernel(const PrepParams p)
{
    const int lane = threadIdx.x % 32;
    const int nc = p.L * p.Pc, nt = p.W * p.L * p.Pt, n = nc + nt;
    const int64_t pairs = p.rows * p.M;
    for (int64_t pair = (int64_t)blockIdx.x * 8 + threadIdx.x / 32; pair < pairs; pair += (int64_t)gridDim.x * 8) {
        const int64_t row = pair / p.M;
        const int m = (int)(pair - row * p.M);
        // first element of this (row, head) in a Linear-side tensor with n_ (x2 for offsets) elements per head
        auto raw = [&](int n_) { return p.ld ? row * p.ld + (int64_t)m * n_ : pair * n_; };
        const T *lc = static_cast<const T *>(BWD ? p.aw_c : p.logit_c) + (BWD ? pair * nc : raw(nc));
        const T *lt = static_cast<const T *>(BWD ? p.aw_t : p.logit_t) + (BWD ? pair * nt : raw(nt));
        constexpr int NE = 8;                 // register-resident fast path: n <= 32 * NE logits per (row, head)
        if (!BWD) {
            // ---- joint softmax over the n logits of this (row, head)   (ref :252-258 / F.softmax)
            T *ac = static_cast<T *>(p.aw_c) + pair * nc, *at = static_cast<T *>(p.aw_t) + pair * nt;
            if (n <= 32 * NE) {               // each logit is read once and exponentiated once
                A v[NE];
                A mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const int e = lane + 32 * i;
                    v[i] = e < n ? (A)Store<T>::get(e < nc ? lc + e : lt + (e - nc)) : (A)-INFINITY;
                    mx = v[i] > mx ? v[i] : mx;
                }
                mx = half_wave_max<A>(mx);
                A sum = 0;
#pragma unroll
                for (int i = 0; i < NE; ++i) { v[i] = lane + 32 * i < n ? prep_exp(v[i] - mx) : (A)0; sum += v[i]; }
                sum = half_wave_sum<A>(sum);
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const int e = lane + 32 * i;
                    if (e < n) Store<T>::put(e < nc ? ac + e : at + (e - nc), v[i] / sum);
                }
            } else {
                A mx = -INFINITY;
                for (int e = lane; e < n; e += 32) {
                    const A v = (A)Store<T>::get(e < nc ? lc + e : lt + (e - nc));
                    mx = v > mx ? v : mx;
                }
                mx = half_wave_max<A>(mx);
                A sum = 0;
                for (int e = lane; e < n; e += 32) sum += prep_exp((A)Store<T>::get(e < nc ? lc + e : lt + (e - nc)) - mx);
                sum = half_wave_sum<A>(sum);
                for (int e = lane; e < n; e += 32) {
                    const A v = prep_exp((A)Store<T>::get(e < nc ? lc + e : lt + (e - nc)) - mx) / sum;
                    Store<T>::put(e < nc ? ac + e : at + (e - nc), v);
                }
            }
        } else {
            // ---- softmax backward: g_logit = p * (g - sum_j p_j g_j)
            const T *gc = static_cast<const T *>(p.gaw_c) + pair * nc, *gt = static_cast<const T *>(p.gaw_t) + pair * nt;
            T *oc = static_cast<T *>(p.glogit_c) + raw(nc), *ot = static_cast<T *>(p.glogit_t) + raw(nt);
            if (n <= 32 * NE) {
                A pe[NE], ge[NE];
                A dot = 0;
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const int e = lane + 32 * i;
                    pe[i] = e < n ? (A)Store<T>::get(e < nc ? lc + e : lt + (e - nc)) : (A)0;
                    ge[i] = e < n ? (A)Store<T>::get(e < nc ? gc + e : gt + (e - nc)) : (A)0;
                    dot += pe[i] * ge[i];
                }
                dot = half_wave_sum<A>(dot);
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const int e = lane + 32 * i;
                    if (e < n) Store<T>::put(e < nc ? oc + e : ot + (e - nc), pe[i] * (ge[i] - dot));
                }
            } else {
                A dot = 0;
                for (int e = lane; e < n; e += 32)
                    dot += (A)Store<T>::get(e < nc ? lc + e : lt + (e - nc)) * (A)Store<T>::get(e < nc ? gc + e : gt + (e - nc));
                dot = half_wave_sum<A>(dot);
                for (int e = lane; e < n; e += 32) {
                    const A pe = (A)Store<T>::get(e < nc ? lc + e : lt + (e - nc));
                    const A ge = (A)Store<T>::get(e < nc ? gc + e : gt + (e - nc));
                    Store<T>::put(e < nc ? oc + e : ot + (e - nc), pe * (ge - dot));
                }
            }
        }
        // ---- sampling locations (ref :112-121): 2-d refs add offsets in pixels of the level, boxes add them
        // as a fraction of half the box; backward: the same factors on grad_loc
        for (int e = lane; e < n; e += 32) {
            const bool cur = e < nc;
            const int ee = cur ? e : e - nc, P = cur ? p.Pc : p.Pt;
            const int vl = ee / P;                            // level (current) or slot*L + level (temporal)
            const int l = cur ? vl : vl % p.L;
            const int64_t idx = (pair * (cur ? nc : nt) + ee) * 2;                 // dense tensors (loc, grad_loc)
            const int64_t ridx = raw(2 * (cur ? nc : nt)) + 2 * ee;                // Linear-side tensors
            const T *ref = static_cast<const T *>(cur ? p.ref_c : p.ref_t) + (row * (cur ? p.L : p.W * p.L) + vl) * p.d;
            const T *in = static_cast<const T *>(BWD ? (cur ? p.gloc_c : p.gloc_t) : (cur ? p.off_c : p.off_t)) + (BWD ? idx : ridx);
            T *out = static_cast<T *>(BWD ? (cur ? p.goff_c : p.goff_t) : (cur ? p.loc_c : p.loc_t)) + (BWD ? ridx : idx);
            const A x = (A)Store<T>::get(in), y = (A)Store<T>::get(in + 1);
            if (p.d == 2) {
                const A nx = (A)p.shapes[2 * l + 1], ny = (A)p.shapes[2 * l];      // (W_l, H_l)
                if (!BWD) {
                    Store<T>::put(out, (A)Store<T>::get(ref) + x / nx);
                    Store<T>::put(out + 1, (A)Store<T>::get(ref + 1) + y / ny);
                } else {
                    Store<T>::put(out, x / nx);
                    Store<T>::put(out + 1, y / ny);
                }
            } else {
                const A bw = (A)Store<T>::get(ref + 2), bh = (A)Store<T>::get(ref + 3);
                if (!BWD) {
                    Store<T>::put(out, (A)Store<T>::get(ref) + x / (A)P * bw * (A)0.5);
                    Store<T>::put(out + 1, (A)Store<T>::get(ref + 1) + y / (A)P * bh * (A)0.5);
                } else {
                    Store<T>::put(out, x * (A)0.5 * bw / (A)P);
                    Store<T>::put(out + 1, y * (A)0.5 * bh / (A)P);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *detail = "")
{
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

thread_local char g_route[512] = "";     // kernels launched by the last entry-point call of this thread (msda_last_route)

int check_launch(const char *what)
{
    const size_t used = strlen(g_route);
    if (used + 3 < sizeof(g_route)) snprintf(g_route + used, sizeof(g_route) - used, "%s%s", used ? "; " : "", what);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return MSDA_ERR_HIP;
    }
    return MSDA_OK;
}

bool aligned16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

size_t tile_lds_bytes(int rpw, int nvl, bool bwd, bool intervals = false)
{
    return (size_t)rpw * kRowSlots * 16 * (bwd ? 3 : 2) + (size_t)nvl * sizeof(Level) +
           (intervals ? (size_t)rpw * nvl * 8 : 0);     // + the per-(row, level) tap-row intervals
}

// ---- test / measurement knobs -------------------------------------------------------------------------------
// All of them are environment variables that are read ONCE (first call into the library, or msda_reload_knobs())
// and only when MSDA_ENABLE_HOOKS=1: a production process cannot have its results or speed changed by a stray
// variable, and the launch path does not call getenv.  tests/ and bench.py set MSDA_ENABLE_HOOKS=1 and call
// msda_reload_knobs() after changing a knob.
struct Knobs {
    int fwd_rs = -1, fwd_rs_nt = 0;     // resident-slab forward: -1 auto, 0 off, 1 force; tiles per wave (0 = auto)
    int bwd_rs = -1;                    // resident-slab gather pass: -1 auto, 0 off, 1 force
    int fwd_slab = -1, bwd_slab = -1;   // slab kernels: -1 auto, 0 off, 1 force
    int fwd_nb = 4;                     // tile forward: points in flight
    int bwd_atomic = 0;                 // MSDA_BWD_MODE=atomic: one-kernel backward with global atomics
    int bwd_phases = 3;                 // 1 = gather pass only, 2 = scatter pass only, 3 = both
    int bwd_cull = 1;                   // 0: no culling structure, 2: (min, max) intervals instead of per-point records
    int bwd_summary = 1;                // 64-query block summaries for long candidate ranges
    int scatter_lds_kb = 144, scatter_wg_per_cu = 1, scatter_dbg = 0;
    int scatter_own = -1;               // owner-computes scatter: -1 auto, 0 off, 1 force (where it applies), 2 the point-granular kernel
    int force_generic = 0;
    int dbg = 0;
};
Knobs g_knobs;
int g_knobs_loaded = 0;

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return (e && e[0]) ? atoi(e) : dflt;
}

void load_knobs()
{
    Knobs k;
    if (env_int("MSDA_ENABLE_HOOKS", 0) == 1) {
        k.fwd_rs = env_int("MSDA_FWD_RS", k.fwd_rs); k.fwd_rs_nt = env_int("MSDA_FWD_RS_NT", k.fwd_rs_nt);
        k.fwd_slab = env_int("MSDA_FWD_SLAB", k.fwd_slab); k.bwd_slab = env_int("MSDA_BWD_SLAB", k.bwd_slab);
        k.bwd_rs = env_int("MSDA_BWD_RS", k.bwd_rs);
        k.fwd_nb = env_int("MSDA_FWD_NB", k.fwd_nb);
        const char *mode = getenv("MSDA_BWD_MODE");
        k.bwd_atomic = (mode && !strcmp(mode, "atomic")) ? 1 : 0;
        k.bwd_phases = env_int("MSDA_BWD_PHASES", k.bwd_phases);
        k.bwd_cull = env_int("MSDA_BWD_CULL", k.bwd_cull);
        k.bwd_summary = env_int("MSDA_BWD_SUMMARY", k.bwd_summary);
        k.scatter_lds_kb = env_int("MSDA_SCATTER_LDS_KB", k.scatter_lds_kb);
        k.scatter_wg_per_cu = env_int("MSDA_SCATTER_WG_PER_CU", k.scatter_wg_per_cu);
        k.scatter_dbg = env_int("MSDA_SCATTER_DBG", k.scatter_dbg);
        k.scatter_own = env_int("MSDA_SCATTER_OWN", k.scatter_own);
        k.force_generic = env_int("MSDA_FORCE_GENERIC", 0) == 1;
        k.dbg = env_int("MSDA_DBG", 0);
    }
    g_knobs = k;
    __atomic_store_n(&g_knobs_loaded, 1, __ATOMIC_RELEASE);
}

inline const Knobs &knobs()
{
    if (!__atomic_load_n(&g_knobs_loaded, __ATOMIC_ACQUIRE)) load_knobs();      // benign race: every thread reads the same environment
    return g_knobs;
}

// ---- per-device caches --------------------------------------------------------------------------------------
constexpr int kMaxDevices = 64;

int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    return dev;
}

int device_cus()
{
    static int cus[kMaxDevices];        // 0 = not asked yet; benign race: every thread computes the same value
    const int dev = current_device();
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

// Dynamic LDS above 64 KiB must be opted into per kernel function AND per device; `granted` is the caller's
// per-instantiation table of what each device has been given so far.
struct LdsGrant { size_t bytes[kMaxDevices]; };
int grant_lds(const void *kernel, size_t bytes, LdsGrant &granted, const char *what)
{
    const int dev = current_device();
    if (bytes <= granted.bytes[dev]) return MSDA_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        return fail(MSDA_ERR_HIP, "msda: cannot reserve the LDS budget of %s", what);
    granted.bytes[dev] = bytes;
    return MSDA_OK;
}

// How many of the LAST pyramid levels fit `cap_pixels` pixels of LDS slab (the device-side rule of first_slab_level,
// evaluated on the host copy of spatial_shapes when the caller passed one; otherwise guessed from the pixel count:
// with the usual stride-2 pyramids level 0 holds ~3/4 of the S pixels).  Returns the first slab level l0.
int host_first_slab_level(const Params &p, long long cap_pixels)
{
    if (p.shapes_host) {
        int l0 = p.L;
        long long acc = 0;
        for (int l = p.L - 1; l >= 0; --l) {
            acc += (long long)p.shapes_host[2 * l] * p.shapes_host[2 * l + 1];
            if (acc > cap_pixels) break;
            l0 = l;
        }
        return l0;
    }
    if (p.L == 1) return (long long)p.S <= cap_pixels ? 0 : 1;
    return (long long)p.S <= cap_pixels ? 0 : ((double)p.S * 0.2551 <= (double)cap_pixels ? 1 : 2);
}

// Pixels of the levels below l0 (the ones the resident-slab kernels gather through the L2), from the host copy of the
// shapes or, without one, from the usual stride-2 pyramid proportions.
long long host_pixels_below(const Params &p, int l0)
{
    if (l0 <= 0) return 0;
    if (p.shapes_host) {
        long long acc = 0;
        for (int l = 0; l < l0 && l < p.L; ++l) acc += (long long)p.shapes_host[2 * l] * p.shapes_host[2 * l + 1];
        return acc;
    }
    return l0 >= p.L ? p.S : (long long)((double)p.S * (l0 == 1 ? 0.75 : 0.94));
}

// Tiles per wave of the resident-slab kernels = how many workgroups share one (clip, head).  Every workgroup of a
// pair gathers the non-resident levels from the same maps, and what an XCD's 4 MiB L2 keeps of them decides the
// kernels' speed: take the LARGEST workgroups (least slab staging) whose pairs in flight per
// XCD still fit `l2_budget`, else the smallest.  Measured on the final kernels, 16 / 32 clips of the DeVIS decoder shape,
// 4 / 2 / 1 tiles per wave: forward fp32 (460 KiB per map) 0.440 / 0.441 / 0.473 and 0.754 / 0.800 / 0.915 ms, bf16
// (230 KiB) 0.343 / 0.372 / 0.395 ms; gather pass fp32 4 vs 2 tiles: 0.547 vs 0.524 ms.  Hence 8 MiB for the forward
// (4 tiles per wave for fp32 too: equal at 16 clips, -6 % at 32) and 4 MiB for the gather pass, whose extra streams
// (grad_out rows, 309 MB of results) compete for the same L2.
int rs_tiles_per_wave(const Params &p, int tiles_per_clip, long long outside_bytes, bool force, long long l2_budget)
{
    const int64_t clips = p.groups / p.frames;
    const int cus_per_xcd = device_cus() / 8 > 0 ? device_cus() / 8 : 1;
    int pick = 0;
    for (int cand : {4, 2, 1}) {
        const int parts = (tiles_per_clip + kRsWaves * cand - 1) / (kRsWaves * cand);
        if (!force && clips * p.M * parts < device_cus()) continue;          // must fill the chip
        pick = cand;
        const long long pairs = (cus_per_xcd + parts - 1) / parts;
        // (beyond 4 MiB only with at least two workgroups per CU: measured equal-or-worse with exactly one)
        if (pairs * outside_bytes <= (4ll << 20) || (pairs * outside_bytes <= l2_budget && clips * p.M * parts >= 2 * device_cus())) break;
    }
    return pick;
}

// Can grad_value go through the LDS scatter kernel?  MSDA_BWD_MODE=atomic forces the one-kernel
// backward with global atomics (kept for A/B measurements and as the any-shape path).
bool standard_value_layout(const Params &p)
{
    return p.v_clip == (int64_t)p.frames * p.S * p.M * p.D && p.v_head == p.D && p.v_pix == p.M * p.D;
}

bool scatter_applicable(const Params &p)
{
    if (knobs().bwd_atomic) return false;
    if (p.L > kScatterMaxLevels || (p.D % 4) != 0) return false;
    if (1 + p.frames * p.window > kScatterMaxSources || p.Lq >= (1 << 24)) return false;   // survivor-list entry fields
    if (p.window == 0 && p.LA != p.L) return false;
    if ((int64_t)p.groups * p.Lq >= 0x7fffffffLL) return false;       // query rows are 32-bit in the hit records
    return true;
}

template <typename T, int G>
int launch_tile(const Params &p, bool bwd, hipStream_t stream)
{
    constexpr int RPW = kWave / G;
    const int64_t tiles = (int64_t)p.groups * ((p.Lq + RPW - 1) / RPW);
    const int64_t blocks = tiles * p.M;
    if (blocks > 0x7fffffffLL) return fail(MSDA_ERR_ARG, "msda: problem too large for one launch%s");
    const size_t lds = tile_lds_bytes(RPW, p.LA + p.LB, bwd, bwd && p.bbox != nullptr);
    // the slab kernels run 4 channels per lane for every dtype (SlabStore): twice the lanes per row for 16-bit types
    constexpr int GSL = SlabStore<T>::VEC == Store<T>::VEC ? G : 2 * G;
    constexpr int RPWS = kWave / (GSL <= kWave ? GSL : kWave);
    if constexpr (G * Store<T>::VEC == 32) if (!bwd && p.LA == p.L && p.L <= kSlabMaxLevels) {
        // resident-slab forward (D = 32, 4-byte types): up to NT * 16 tiles of 16 rows per workgroup, so that the
        // per-frame slab staging is amortised
        const int mode = knobs().fwd_rs;                               // -1 auto, 0 off, 1 force
        const int tiles_per_clip = p.frames * ((p.Lq + kRsRows - 1) / kRsRows);
        const int64_t clips = p.groups / p.frames;
        const int64_t pixB = (int64_t)p.v_pix * (int64_t)sizeof(T);
        const bool fits = (int64_t)p.frames * p.S < (1 << 24) && pixB < (1 << 24) && p.D == 32 &&
                          (int64_t)p.frames * p.S * pixB < 0x7fffffffLL && p.frames <= kRsMaxFrames && p.window <= 31;
        // tiles per wave (NT) and workgroups per (clip, head) (parts): see rs_tiles_per_wave
        const int slab_bytes = ((160 * 1024 - 256 - kRsTailBytes) / 128) * 128;
        const int l0_host = host_first_slab_level(p, (slab_bytes - kRsSlack) / rs_row_bytes<T>());
        int nt = rs_tiles_per_wave(p, tiles_per_clip, host_pixels_below(p, l0_host) * rs_row_bytes<T>(), mode == 1, 8ll << 20);
        int parts = nt ? (tiles_per_clip + kRsWaves * nt - 1) / (kRsWaves * nt) : 0;
        // the slab must hold at least the last level.  (Since the whole-row loads / stores of the points and gradients the
        // kernel wins for every dtype as soon as ANY level fits -- 800x1333, levels 2-3 resident: bf16 forward 0.44 -> 0.39
        // ms, gather pass 0.79 -> 0.54; fp32 0.61 -> 0.55 and 0.80 -> 0.63 against the round-1 slab kernels.)
        const int l0_max = p.L - 1;
        if (mode != 1 && l0_host > l0_max) nt = 0;
        const int force_nt = knobs().fwd_rs_nt;
        if (force_nt == 1 || force_nt == 2 || force_nt == 4) { nt = force_nt; parts = (tiles_per_clip + kRsWaves * nt - 1) / (kRsWaves * nt); }
        if (mode != 0 && fits && nt && clips * p.M * parts <= 0x7fffffffLL) {
            const size_t total = (size_t)slab_bytes + kRsTailBytes;
            const unsigned grid = (unsigned)(clips * p.M * parts);
            auto launch = [&](auto kern, LdsGrant &granted) {
                const int rc = grant_lds(reinterpret_cast<const void *>(kern), total, granted, "the resident-slab forward kernel");
                if (rc) return rc;
                hipLaunchKernelGGL(kern, dim3(grid), dim3(kRsThreads), total, stream, p, slab_bytes, parts);
                return check_launch("msda forward (resident-slab kernel)");
            };
            static LdsGrant g4, g2, g1;
            return nt == 4 ? launch(&msda_fwd_rs_kernel<T, 4>, g4) : nt == 2 ? launch(&msda_fwd_rs_kernel<T, 2>, g2)
                                                                              : launch(&msda_fwd_rs_kernel<T, 1>, g1);
        }
    }
    if constexpr (GSL >= 4 && GSL <= kWave) if (!bwd && p.LA == p.L && p.L <= kSlabMaxLevels) {   // (1- and 2-lane rows spill)
        // slab forward: 16 waves per workgroup share the small levels in LDS; needs enough workgroups.
        // Picked automatically for 4-byte types only: for the 16-bit types the 16-byte-lane tile kernel (half
        // the instructions per point) is as fast or faster (cfg2 bf16: 0.48 vs 0.57 ms); the gather pass
        // below does profit (0.85 -> 0.75 ms) and takes the slab for every dtype.
        const int tiles_per_clip = p.frames * ((p.Lq + RPWS - 1) / RPWS);
        const int blocks_per_clip = (tiles_per_clip + kSlabWaves - 1) / kSlabWaves;
        const int64_t slab_blocks = (int64_t)(p.groups / p.frames) * blocks_per_clip * p.M;
        const int mode = knobs().fwd_slab;                             // -1 auto, 0 off, 1 force
        const size_t per_wave = (size_t)RPWS * kRowSlots * 32 + (size_t)(p.LA + p.LB) * sizeof(Level);
        const long long slab_bytes = ((160 * 1024 - 1024 - (long long)kSlabWaves * (long long)per_wave) / 1024) * 1024;
        if (mode != 0 && slab_bytes >= 16 * 1024 && (mode == 1 || (slab_blocks >= 2 * device_cus() && sizeof(T) == 4)) &&
            slab_blocks <= 0x7fffffffLL) {
            const size_t total = (size_t)slab_bytes + kSlabWaves * per_wave;
            static LdsGrant granted;
            if (const int rc = grant_lds(reinterpret_cast<const void *>(&msda_fwd_slab_kernel<T, GSL, 4>), total, granted,
                                         "the slab forward kernel")) return rc;
            hipLaunchKernelGGL((msda_fwd_slab_kernel<T, GSL, 4>), dim3((unsigned)slab_blocks), dim3(kSlabThreads),
                               total, stream, p, (int)(slab_bytes / (long long)sizeof(T)));
            return check_launch("msda forward (slab kernel)");
        }
    }
    if (!bwd) {
        const int nb = knobs().fwd_nb;
        if (nb == 1)
            hipLaunchKernelGGL((msda_fwd_tile_kernel<T, G, 1>), dim3((unsigned)blocks), dim3(kWave), lds, stream, p);
        else if (nb == 2)
            hipLaunchKernelGGL((msda_fwd_tile_kernel<T, G, 2>), dim3((unsigned)blocks), dim3(kWave), lds, stream, p);
        else
            hipLaunchKernelGGL((msda_fwd_tile_kernel<T, G, 4>), dim3((unsigned)blocks), dim3(kWave), lds, stream, p);
        return check_launch("msda forward (tile kernel)");
    }
    if (!scatter_applicable(p)) {
        if (hipMemsetAsync(p.grad_value, 0, (size_t)p.groups * p.S * p.M * p.D * sizeof(float), stream) != hipSuccess)
            return fail(MSDA_ERR_HIP, "msda backward: hipMemsetAsync(grad_value) failed%s");
        hipLaunchKernelGGL((msda_bwd_tile_kernel<T, G, true>), dim3((unsigned)blocks), dim3(kWave), lds, stream, p);
        return check_launch("msda backward (tile kernel, global atomics)");
    }
    // MSDA_BWD_PHASES (measurement hook for bench.py): 1 = gather pass only, 2 = scatter pass only
    // (needs the workspace a previous gather pass filled), 3 = both (default)
    const int phases = knobs().bwd_phases;
    int rc = MSDA_OK;
    if (phases & 1) {
        bool done = false;
        if constexpr (G * Store<T>::VEC == 32) if (p.LA == p.L && p.L <= kSlabMaxLevels && (p.cull_points || !p.bbox)) {
            // resident-slab gather pass (D = 32, 4-byte types): same applicability rule as the forward
            const int mode = knobs().bwd_rs;
            const int tiles_per_clip = p.frames * ((p.Lq + kRsRows - 1) / kRsRows);
            const int64_t clips = p.groups / p.frames;
            const int64_t pixB = (int64_t)p.v_pix * (int64_t)sizeof(T);
            const bool fits = (int64_t)p.frames * p.S < (1 << 24) && pixB < (1 << 24) && p.D == 32 &&
                              (int64_t)p.frames * p.S * pixB < 0x7fffffffLL && p.frames <= kRsMaxFrames && p.window <= 31 &&
                              (int64_t)(p.PA > p.PB ? p.PA : p.PB) * (p.PA > p.PB ? p.PA : p.PB) * p.L < 65536;   // kk / P by reciprocal
            const int slab_bytes = ((160 * 1024 - 256 - kRsTailBytes) / 128) * 128;
            const int l0_host = host_first_slab_level(p, (slab_bytes - kRsSlack) / rs_row_bytes<T>());
            const int tpw = rs_tiles_per_wave(p, tiles_per_clip, host_pixels_below(p, l0_host) * rs_row_bytes<T>(), mode == 1, 4ll << 20);
            const int parts = tpw ? (tiles_per_clip + tpw * kRsWaves - 1) / (tpw * kRsWaves) : 1;       // (L2: see the forward)
            bool want = mode == 1 || (mode == -1 && tpw && l0_host <= p.L - 1);
            if (want && fits && clips * p.M * parts <= 0x7fffffffLL) {
                const size_t total = (size_t)slab_bytes + kRsTailBytes;
                static LdsGrant granted;
                if (const int grc = grant_lds(reinterpret_cast<const void *>(&msda_bwd_rs_kernel<T>), total, granted,
                                              "the resident-slab gather-pass kernel")) return grc;
                hipLaunchKernelGGL((msda_bwd_rs_kernel<T>), dim3((unsigned)(clips * p.M * parts)), dim3(kRsThreads), total, stream, p,
                                   slab_bytes, parts);
                rc = check_launch("msda backward (resident-slab kernel, grad_loc/grad_attn)");
                if (rc) return rc;
                done = true;
            }
        }
        if constexpr (GSL >= 4 && GSL <= kWave) if (!done && p.LA == p.L && p.L <= kSlabMaxLevels) {      // slab variant of the gather pass
            const int tiles_per_clip = p.frames * ((p.Lq + RPWS - 1) / RPWS);
            const int blocks_per_clip = (tiles_per_clip + kSlabWaves - 1) / kSlabWaves;
            const int64_t slab_blocks = (int64_t)(p.groups / p.frames) * blocks_per_clip * p.M;
            const int mode = knobs().bwd_slab;                          // -1 auto, 0 off, 1 force
            const size_t per_wave = (size_t)RPWS * kRowSlots * 48 + (size_t)(p.LA + p.LB) * sizeof(Level) +
                                    (p.bbox ? (size_t)RPWS * p.L * 8 : 0);
            const long long slab_bytes =
                ((160 * 1024 - 1024 - (long long)kSlabWaves * (long long)per_wave) / 1024) * 1024;
            if (mode != 0 && slab_bytes >= 16 * 1024 && (mode == 1 || slab_blocks >= 2 * device_cus()) &&
                slab_blocks <= 0x7fffffffLL) {
                const size_t total = (size_t)slab_bytes + kSlabWaves * per_wave;
                static LdsGrant granted;
                if (const int grc = grant_lds(reinterpret_cast<const void *>(&msda_bwd_slab_kernel<T, GSL>), total, granted,
                                              "the slab gather-pass kernel")) return grc;
                hipLaunchKernelGGL((msda_bwd_slab_kernel<T, GSL>), dim3((unsigned)slab_blocks), dim3(kSlabThreads),
                                   total, stream, p, (int)(slab_bytes / (long long)sizeof(T)), (int)per_wave);
                rc = check_launch("msda backward (slab kernel, grad_loc/grad_attn)");
                if (rc) return rc;
                done = true;
            }
        }
        if (!done) {
            hipLaunchKernelGGL((msda_bwd_tile_kernel<T, G, false>), dim3((unsigned)blocks), dim3(kWave), lds, stream, p);
            rc = check_launch("msda backward (tile kernel, grad_loc/grad_attn)");
            if (rc) return rc;
        }
    }
    if ((phases & 1) && p.cull_points && p.bsum) {       // block summaries of the per-point records just written
        const int64_t entries = (int64_t)p.groups * p.M * (p.LA + p.LB) * ((p.Lq + kCullBlock - 1) / kCullBlock);
        const unsigned sb = (unsigned)((entries + 3) / 4 < 65536 ? (entries + 3) / 4 : 65536);
        hipLaunchKernelGGL(msda_cull_summary_kernel, dim3(sb), dim3(256), 0, stream, p);
        rc = check_launch("msda backward (culling block summaries)");
        if (rc) return rc;
    }
    if (!(phases & 2)) return rc;
    if constexpr (G * Store<T>::VEC == 32) {
        // owner-computes scatter (D = 32, <= 4 points per level): no float atomics
        const int own = knobs().scatter_own;
        if (own != 0 && p.PA <= 4 && p.PB <= 4 && (p.cull_points || !p.bbox) && knobs().scatter_lds_kb == 144 &&
            knobs().scatter_wg_per_cu == 1) {
            unsigned grid = (unsigned)device_cus();
            grid -= grid % 8;
            static LdsGrant granted_own[4];
            const int variant = (knobs().scatter_dbg & 256) ? 0 : 2;       // (measurement: 256 = no cross-chunk prefetch)
            auto own_launch = [&](auto kern) {
                if (const int grc = grant_lds(reinterpret_cast<const void *>(kern), (size_t)own_lds_bytes<T>(), granted_own[variant],
                                              "the owner-computes scatter kernel")) return grc;
                hipLaunchKernelGGL(kern, dim3(grid), dim3(kOwnThreads), (size_t)own_lds_bytes<T>(), stream, p, knobs().scatter_dbg);
                return check_launch("msda backward (owner-computes scatter kernel)");
            };
            const int64_t rows = (int64_t)p.groups * p.S;
            const unsigned zb = (unsigned)((rows + 255) / 256 < 16384 ? (rows + 255) / 256 : 16384);
            hipLaunchKernelGGL(msda_zero_unowned_kernel, dim3(zb), dim3(256), 0, stream, p, kOwnPix * p.D);
            rc = check_launch("msda backward (zero-fill of pixels outside the bands)");
            if (rc) return rc;
            if (own != 2 && p.Lq < (1 << 22)) {      // group-granular variant (MSDA_SCATTER_OWN=2: the point-granular kernel)
                static LdsGrant granted_grp;
                if (const int grc = grant_lds(reinterpret_cast<const void *>(&msda_bwd_value_grp_kernel<T>), (size_t)grp_lds_bytes<T>(), granted_grp,
                                              "the group-granular owner-computes scatter kernel")) return grc;
                hipLaunchKernelGGL((msda_bwd_value_grp_kernel<T>), dim3(grid), dim3(kOwnThreads), (size_t)grp_lds_bytes<T>(), stream, p,
                                   knobs().scatter_dbg & 255);
                return check_launch("msda backward (owner-computes scatter kernel, group-granular)");
            }
            return variant == 0 ? own_launch(&msda_bwd_value_own_kernel<T, 0>) : own_launch(&msda_bwd_value_own_kernel<T, 2>);
        }
    }
    // LDS budget: one 1024-thread workgroup per CU with 144 KiB of 8-byte accumulators
    const int cap_bytes = knobs().scatter_lds_kb * 1024;
    const int per_cu = knobs().scatter_wg_per_cu;
    unsigned grid = (unsigned)(device_cus() * per_cu);
    grid -= grid % 8;                                   // multiple of the XCD count: item % M stays put
    {
        static LdsGrant granted;       // per instantiation and device
        if (const int grc = grant_lds(reinterpret_cast<const void *>(&msda_bwd_value_lds_kernel<T, G>), (size_t)cap_bytes,
                                      granted, "the LDS scatter kernel")) return grc;
    }
    {
        // pixels the scatter will not overwrite (normally none) are zero-filled first
        const int slots = (p.cull_points ? (cap_bytes < kPointsCapBytes ? cap_bytes : kPointsCapBytes) / 8 - 2 * p.D : cap_bytes / 8);
        const int64_t rows = (int64_t)p.groups * p.S;
        const unsigned zb = (unsigned)((rows + 255) / 256 < 16384 ? (rows + 255) / 256 : 16384);
        hipLaunchKernelGGL(msda_zero_unowned_kernel, dim3(zb), dim3(256), 0, stream, p, slots);
        rc = check_launch("msda backward (zero-fill of pixels outside the LDS bands)");
        if (rc) return rc;
    }
    if (p.cull_points) {
        static LdsGrant granted_points;
        const int cap_pts = cap_bytes < kPointsCapBytes ? cap_bytes : kPointsCapBytes;   // its static tables need ~17 KiB
        if (const int grc = grant_lds(reinterpret_cast<const void *>(&msda_bwd_value_points_kernel<T, G>), (size_t)cap_pts,
                                      granted_points, "the per-point LDS scatter kernel")) return grc;
        hipLaunchKernelGGL((msda_bwd_value_points_kernel<T, G>), dim3(grid), dim3(kScatterThreads),
                           (size_t)cap_pts, stream, p, cap_pts / 8, knobs().scatter_dbg);
        return check_launch("msda backward (LDS scatter kernel, per-point culling)");
    }
    hipLaunchKernelGGL((msda_bwd_value_lds_kernel<T, G>), dim3(grid), dim3(kScatterThreads),
                       (size_t)cap_bytes, stream, p, cap_bytes / 8, knobs().scatter_dbg);
    return check_launch("msda backward (LDS scatter kernel)");
}

template <typename T>
int dispatch_tile(const Params &p, bool bwd, hipStream_t stream, bool &taken)
{
    constexpr int VEC = Store<T>::VEC;
    taken = false;
    if (p.D % VEC) return MSDA_OK;
    const int G = p.D / VEC;
    // every 16-B lane vector must be aligned: bases 16-B aligned and D a multiple of VEC
    if (!aligned16(p.value) || (!bwd && !aligned16(p.out)) ||
        (bwd && (!aligned16(p.grad_out) || !aligned16(p.grad_value))))
        return MSDA_OK;
    // element offsets inside one clip slab are 32-bit in the tap records
    if ((int64_t)p.frames * p.S * p.M * p.D >= 0x7fffffffLL || (int64_t)p.frames * p.S * p.v_pix >= 0x7fffffffLL ||
        (int64_t)p.frames * p.S * p.v_pix * (int64_t)sizeof(T) >= (int64_t)kOobBytes)  // gather_load: 32-bit byte offsets < kOobBytes
        return MSDA_OK;
    if (p.v_clip % VEC || p.v_head % VEC || p.v_pix % VEC) return MSDA_OK;
    // the one-kernel backward scatters grad_value (always dense) at value's offsets
    if (bwd && !scatter_applicable(p) && !standard_value_layout(p)) return MSDA_OK;
    if (tile_lds_bytes(kWave / (G > 0 ? G : 1), p.LA + p.LB, bwd) > 60 * 1024) return MSDA_OK;
    taken = true;
    switch (G) {
        case 1: return launch_tile<T, 1>(p, bwd, stream);
        case 2: return launch_tile<T, 2>(p, bwd, stream);
        case 4: return launch_tile<T, 4>(p, bwd, stream);
        case 8: return launch_tile<T, 8>(p, bwd, stream);
        case 16: return launch_tile<T, 16>(p, bwd, stream);
        case 32: return launch_tile<T, 32>(p, bwd, stream);
        case 64: return launch_tile<T, 64>(p, bwd, stream);
        default: taken = false; return MSDA_OK;
    }
}

template <typename T, typename A>
int launch_generic(const Params &p, bool bwd, hipStream_t stream)
{
    const int64_t rows = (int64_t)p.groups * p.Lq * p.M;
    if (bwd) {
        if (hipMemsetAsync(p.grad_value, 0, (size_t)p.groups * p.S * p.M * p.D * sizeof(A), stream) != hipSuccess)
            return fail(MSDA_ERR_HIP, "msda backward: hipMemsetAsync(grad_value) failed%s");
        const unsigned blocks = (unsigned)(rows < 65536 * 16 ? rows : 65536 * 16);
        hipLaunchKernelGGL((msda_bwd_generic_kernel<T, A>), dim3(blocks), dim3(kWave), 0, stream, p, rows);
        return check_launch("msda backward (generic kernel)");
    }
    const int64_t total = rows * p.D;
    const int64_t want = (total + 255) / 256;
    const unsigned blocks = (unsigned)(want < 65536 * 8 ? want : 65536 * 8);
    hipLaunchKernelGGL((msda_fwd_generic_kernel<T, A>), dim3(blocks), dim3(256), 0, stream, p, total);
    return check_launch("msda forward (generic kernel)");
}

int run(int dtype, const Params &p_in, bool bwd, hipStream_t stream)
{
    Params p = p_in;
    p.dbg = knobs().dbg;
    // culling records per point when a level has <= 4 points (MSDA_BWD_CULL=2: force (min, max) intervals)
    p.cull_points = bwd && p.bbox && p.PA <= 4 && p.PB <= 4 && knobs().bwd_cull != 2;
    if (!p.cull_points) p.bsum = nullptr;
    p.wide_stores = bwd && aligned16(p.glocA) && aligned16(p.gawA) && (p.LB == 0 || (aligned16(p.glocB) && aligned16(p.gawB))) &&
                    (knobs().dbg & 64) == 0;                  // (measurement: MSDA_DBG=64 keeps the narrow stores)
    p.wide_loads = aligned16(p.locA) && aligned16(p.awA) && (p.LB == 0 || (aligned16(p.locB) && aligned16(p.awB))) &&
                   (knobs().dbg & 128) == 0;                  // (measurement: MSDA_DBG=128 keeps the narrow loads)
    if (p.groups == 0 || p.Lq == 0) return MSDA_OK;
    bool taken = false;
    int rc = MSDA_OK;
    const bool force_generic = knobs().force_generic != 0;
    switch (dtype) {
        case MSDA_F32:
            if (!force_generic) rc = dispatch_tile<float>(p, bwd, stream, taken);
            if (!taken) rc = launch_generic<float, float>(p, bwd, stream);
            return rc;
        case MSDA_BF16:
            if (!force_generic) rc = dispatch_tile<bf16_t>(p, bwd, stream, taken);
            if (!taken) rc = launch_generic<bf16_t, float>(p, bwd, stream);
            return rc;
        case MSDA_F16:
            if (!force_generic) rc = dispatch_tile<f16_t>(p, bwd, stream, taken);
            if (!taken) rc = launch_generic<f16_t, float>(p, bwd, stream);
            return rc;
        case MSDA_F64:
            return launch_generic<double, double>(p, bwd, stream);
        default:
            return fail(MSDA_ERR_DTYPE, "msda: unknown dtype code%s");
    }
}

int check_common(const void *value, const int64_t *shapes, const int64_t *lsi, int groups, int S,
                 int M, int D, int L, int Lq)
{
    if (!value || !shapes || !lsi) return fail(MSDA_ERR_ARG, "msda: null pointer argument%s");
    if (groups < 0 || Lq < 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0)
        return fail(MSDA_ERR_ARG, "msda: sizes must be positive%s");
    return MSDA_OK;
}

// `value_strides` (host pointer, may be null): element strides {between clips, between heads, between pixels}
// of `value`; null = the standard dense [groups, S, M, D].
int set_value_strides(Params &p, const int64_t *vs)
{
    p.v_clip = (int64_t)p.frames * p.S * p.M * p.D; p.v_head = p.D; p.v_pix = p.M * p.D;
    if (!vs) return MSDA_OK;
    if (vs[0] < 0 || vs[1] < 0 || vs[2] <= 0 || vs[2] > 0x7fffffffLL)
        return fail(MSDA_ERR_ARG, "msda: bad value_strides%s");
    p.v_clip = vs[0]; p.v_head = vs[1]; p.v_pix = (int)vs[2];
    return MSDA_OK;
}

int zero_grad_value(int dtype, void *grad_value, int groups, int S, int M, int D, void *stream)
{
    if (dtype < MSDA_F32 || dtype > MSDA_F16) return fail(MSDA_ERR_DTYPE, "msda: unknown dtype code%s");
    if (!grad_value) return fail(MSDA_ERR_ARG, "msda backward: null grad_value%s");
    const size_t bytes = (size_t)groups * S * M * D * (dtype == MSDA_F64 ? sizeof(double) : sizeof(float));
    if (hipMemsetAsync(grad_value, 0, bytes, static_cast<hipStream_t>(stream)) != hipSuccess)
        return fail(MSDA_ERR_HIP, "msda backward: hipMemsetAsync(grad_value) failed%s");
    return MSDA_OK;
}

long long workspace_table_bytes(int batch, int num_query, int num_heads, int virtual_levels)
{
    return (long long)batch * num_query * num_heads * virtual_levels * 8;
}

// ticket counters + per-point culling records + their 64-query block summaries
long long workspace_need(int batch, int num_query, int num_heads, int virtual_levels)
{
    const long long nblk = (num_query + kCullBlock - 1) / kCullBlock;
    return MSDA_BWD_WORKSPACE_BYTES + workspace_table_bytes(batch, num_query, num_heads, virtual_levels) +
           (long long)batch * num_heads * virtual_levels * nblk * 8;
}

void attach_workspace(Params &p, void *workspace, long long bytes, int batch, int num_query, int num_heads, int vl)
{
    p.workspace = (workspace && bytes >= MSDA_BWD_WORKSPACE_BYTES) ? static_cast<unsigned *>(workspace) : nullptr;
    p.bbox = nullptr;
    p.bsum = nullptr;
    if (p.workspace && bytes >= workspace_need(batch, num_query, num_heads, vl) && knobs().bwd_cull != 0) {
        p.bbox = reinterpret_cast<int *>(p.workspace) + MSDA_BWD_WORKSPACE_BYTES / 4;
        // block summaries only pay for long candidate ranges (and index (group, head, level) rows with 32 bits)
        if (num_query >= 2048 && (long long)batch * num_heads * vl < 0x7fffffffLL && knobs().bwd_summary != 0)
            p.bsum = p.bbox + workspace_table_bytes(batch, num_query, num_heads, vl) / 4;
    }
}

template <typename T, typename A>
int launch_prep(const PrepParams &p, bool bwd, hipStream_t stream)
{
    const int64_t pairs = p.rows * p.M;
    const int64_t want = (pairs + 7) / 8;
    const unsigned blocks = (unsigned)(want < 65536 * 4 ? want : 65536 * 4);
    if (bwd) hipLaunchKernelGGL((msda_prep_kernel<T, A, true>), dim3(blocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((msda_prep_kernel<T, A, false>), dim3(blocks), dim3(256), 0, stream, p);
    return check_launch(bwd ? "msda prep backward" : "msda prep forward");
}

int run_prep(int dtype, const PrepParams &p, bool bwd, void *stream)
{
    if (p.rows < 0 || p.M <= 0 || p.L <= 0 || p.W < 0 || p.Pc <= 0 || (p.W > 0 && p.Pt <= 0) || (p.d != 2 && p.d != 4))
        return fail(MSDA_ERR_ARG, "msda prep: bad sizes (rows, heads, levels, window, points, reference dim)%s");
    if (!p.shapes || !p.ref_c || (p.W > 0 && !p.ref_t)) return fail(MSDA_ERR_ARG, "msda prep: null pointer argument%s");
    if (p.rows == 0) return MSDA_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case MSDA_F32: return launch_prep<float, float>(p, bwd, st);
        case MSDA_F64: return launch_prep<double, double>(p, bwd, st);
        case MSDA_BF16: return launch_prep<bf16_t, float>(p, bwd, st);
        case MSDA_F16: return launch_prep<f16_t, float>(p, bwd, st);
        default: return fail(MSDA_ERR_DTYPE, "msda: unknown dtype code%s");
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Padding mask (ref ms_deform_attn.py:102-103 `value.masked_fill(mask[..., None], 0)`).
// The reference's masked_fill is a full read + write of `value`; only the masked rows change, so this pass reads the
// [pixels] byte mask and WRITES the masked rows only (G bytes per thread): cost ~ pixels bytes + the masked rows.
template <int G>
__global__ __launch_bounds__(256) void msda_mask_rows_kernel(char *__restrict__ rows, const uint8_t *__restrict__ mask,
                                                             long long pixels, int chunks, long long stride_bytes)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long pix = idx / chunks;
    if (pix >= pixels || !mask[pix]) return;
    char *dst = rows + pix * stride_bytes + (idx - pix * chunks) * G;
    if constexpr (G == 16) *reinterpret_cast<uint4 *>(dst) = make_uint4(0, 0, 0, 0);
    else if constexpr (G == 8) *reinterpret_cast<uint2 *>(dst) = make_uint2(0, 0);
    else if constexpr (G == 4) *reinterpret_cast<uint32_t *>(dst) = 0;
    else *reinterpret_cast<uint16_t *>(dst) = 0;
}

}  // namespace

extern "C" {

int msda_version(void) { return MSDA_ABI_VERSION; }

void msda_reload_knobs(void) { load_knobs(); }

const char *msda_last_route(void) { return g_route; }

long long msda_backward_workspace_bytes(int batch, int num_query, int num_heads, int virtual_levels)
{
    return workspace_need(batch, num_query, num_heads, virtual_levels);
}

const char *msda_last_error(void) { return g_err; }

int msda_forward(int dtype, const void *value, const int64_t *spatial_shapes,
                 const int64_t *level_start_index, const void *sampling_loc, const void *attn_weight,
                 int batch, int spatial_size, int num_heads, int channels, int num_levels,
                 int num_query, int num_point, void *out, const int64_t *value_strides,
                 const int64_t *spatial_shapes_host, void *stream)
{
    g_err[0] = 0; g_route[0] = 0;
    int rc = check_common(value, spatial_shapes, level_start_index, batch, spatial_size, num_heads,
                          channels, num_levels, num_query);
    if (rc) return rc;
    if (batch == 0 || num_query == 0) return MSDA_OK;
    if (!sampling_loc || !attn_weight || !out || num_point <= 0)
        return fail(MSDA_ERR_ARG, "msda_forward: null pointer or non-positive num_point%s");
    Params p;
    memset(&p, 0, sizeof(p));
    p.value = value; p.shapes = spatial_shapes; p.lsi = level_start_index;
    p.locA = sampling_loc; p.awA = attn_weight; p.out = out;
    p.groups = batch; p.frames = 1; p.window = 0;
    p.S = spatial_size; p.M = num_heads; p.D = channels; p.L = num_levels; p.Lq = num_query;
    p.LA = num_levels; p.PA = num_point; p.LB = 0; p.PB = 1;
    p.shapes_host = spatial_shapes_host;
    rc = set_value_strides(p, value_strides);
    if (rc) return rc;
    return run(dtype, p, false, static_cast<hipStream_t>(stream));
}

int msda_backward(int dtype, const void *value, const int64_t *spatial_shapes,
                  const int64_t *level_start_index, const void *sampling_loc,
                  const void *attn_weight, const void *grad_out,
                  int batch, int spatial_size, int num_heads, int channels, int num_levels,
                  int num_query, int num_point,
                  void *grad_value, void *grad_sampling_loc, void *grad_attn_weight,
                  void *workspace, long long workspace_bytes, const int64_t *value_strides,
                  const int64_t *spatial_shapes_host, void *stream)
{
    g_err[0] = 0; g_route[0] = 0;
    int rc = check_common(value, spatial_shapes, level_start_index, batch, spatial_size, num_heads,
                          channels, num_levels, num_query);
    if (rc) return rc;
    if (batch == 0) return MSDA_OK;
    if (num_query == 0) return zero_grad_value(dtype, grad_value, batch, spatial_size, num_heads, channels, stream);
    if (!sampling_loc || !attn_weight || !grad_out || !grad_value || !grad_sampling_loc ||
        !grad_attn_weight || num_point <= 0)
        return fail(MSDA_ERR_ARG, "msda_backward: null pointer or non-positive num_point%s");
    Params p;
    memset(&p, 0, sizeof(p));
    p.value = value; p.shapes = spatial_shapes; p.lsi = level_start_index;
    p.locA = sampling_loc; p.awA = attn_weight; p.grad_out = grad_out;
    p.grad_value = grad_value; p.glocA = grad_sampling_loc; p.gawA = grad_attn_weight;
    attach_workspace(p, workspace, workspace_bytes, batch, num_query, num_heads, num_levels);
    p.groups = batch; p.frames = 1; p.window = 0;
    p.S = spatial_size; p.M = num_heads; p.D = channels; p.L = num_levels; p.Lq = num_query;
    p.LA = num_levels; p.PA = num_point; p.LB = 0; p.PB = 1;
    p.shapes_host = spatial_shapes_host;
    rc = set_value_strides(p, value_strides);
    if (rc) return rc;
    return run(dtype, p, true, static_cast<hipStream_t>(stream));
}

int msda_temporal_forward(int dtype, const void *value, const int64_t *spatial_shapes,
                          const int64_t *level_start_index, const int32_t *frame_table,
                          const void *loc_curr, const void *aw_curr,
                          const void *loc_temp, const void *aw_temp,
                          int clips, int frames, int window, int spatial_size, int num_heads,
                          int channels, int num_levels, int num_query,
                          int num_curr_point, int num_temp_point, void *out, const int64_t *value_strides,
                          const int64_t *spatial_shapes_host, void *stream)
{
    g_err[0] = 0; g_route[0] = 0;
    int rc = check_common(value, spatial_shapes, level_start_index, clips, spatial_size, num_heads,
                          channels, num_levels, num_query);
    if (rc) return rc;
    if (frames <= 0 || window < 0 || num_curr_point <= 0 || (window > 0 && num_temp_point <= 0))
        return fail(MSDA_ERR_ARG, "msda_temporal_forward: bad frames/window/points%s");
    if (clips == 0 || num_query == 0) return MSDA_OK;
    if (!loc_curr || !aw_curr || !out || (window > 0 && (!frame_table || !loc_temp || !aw_temp)))
        return fail(MSDA_ERR_ARG, "msda_temporal_forward: null pointer argument%s");
    Params p;
    memset(&p, 0, sizeof(p));
    p.value = value; p.shapes = spatial_shapes; p.lsi = level_start_index; p.ftab = frame_table;
    p.locA = loc_curr; p.awA = aw_curr; p.locB = loc_temp; p.awB = aw_temp; p.out = out;
    p.groups = clips * frames; p.frames = frames; p.window = window;
    p.S = spatial_size; p.M = num_heads; p.D = channels; p.L = num_levels; p.Lq = num_query;
    p.LA = num_levels; p.PA = num_curr_point;
    p.LB = window * num_levels; p.PB = window > 0 ? num_temp_point : 1;
    p.shapes_host = spatial_shapes_host;
    rc = set_value_strides(p, value_strides);
    if (rc) return rc;
    return run(dtype, p, false, static_cast<hipStream_t>(stream));
}

int msda_temporal_backward(int dtype, const void *value, const int64_t *spatial_shapes,
                           const int64_t *level_start_index, const int32_t *frame_table,
                           const void *loc_curr, const void *aw_curr,
                           const void *loc_temp, const void *aw_temp, const void *grad_out,
                           int clips, int frames, int window, int spatial_size, int num_heads,
                           int channels, int num_levels, int num_query,
                           int num_curr_point, int num_temp_point,
                           void *grad_value, void *grad_loc_curr, void *grad_aw_curr,
                           void *grad_loc_temp, void *grad_aw_temp, void *workspace, long long workspace_bytes,
                           const int64_t *value_strides, const int64_t *spatial_shapes_host, void *stream)
{
    g_err[0] = 0; g_route[0] = 0;
    int rc = check_common(value, spatial_shapes, level_start_index, clips, spatial_size, num_heads,
                          channels, num_levels, num_query);
    if (rc) return rc;
    if (frames <= 0 || window < 0 || num_curr_point <= 0 || (window > 0 && num_temp_point <= 0))
        return fail(MSDA_ERR_ARG, "msda_temporal_backward: bad frames/window/points%s");
    if (clips == 0) return MSDA_OK;
    if (num_query == 0)
        return zero_grad_value(dtype, grad_value, clips * frames, spatial_size, num_heads, channels, stream);
    if (!loc_curr || !aw_curr || !grad_out || !grad_value || !grad_loc_curr || !grad_aw_curr ||
        (window > 0 && (!frame_table || !loc_temp || !aw_temp || !grad_loc_temp || !grad_aw_temp)))
        return fail(MSDA_ERR_ARG, "msda_temporal_backward: null pointer argument%s");
    Params p;
    memset(&p, 0, sizeof(p));
    p.value = value; p.shapes = spatial_shapes; p.lsi = level_start_index; p.ftab = frame_table;
    p.locA = loc_curr; p.awA = aw_curr; p.locB = loc_temp; p.awB = aw_temp; p.grad_out = grad_out;
    p.grad_value = grad_value; p.glocA = grad_loc_curr; p.gawA = grad_aw_curr;
    p.glocB = grad_loc_temp; p.gawB = grad_aw_temp;
    attach_workspace(p, workspace, workspace_bytes, clips * frames, num_query, num_heads, num_levels * (1 + window));
    p.groups = clips * frames; p.frames = frames; p.window = window;
    p.S = spatial_size; p.M = num_heads; p.D = channels; p.L = num_levels; p.Lq = num_query;
    p.LA = num_levels; p.PA = num_curr_point;
    p.LB = window * num_levels; p.PB = window > 0 ? num_temp_point : 1;
    p.shapes_host = spatial_shapes_host;
    rc = set_value_strides(p, value_strides);
    if (rc) return rc;
    return run(dtype, p, true, static_cast<hipStream_t>(stream));
}

int msda_prep_forward(int dtype, const void *offsets_curr, const void *offsets_temp, const void *logits_curr,
                      const void *logits_temp, const void *ref_curr, const void *ref_temp,
                      const int64_t *spatial_shapes, long long rows, int num_heads, int num_levels, int window,
                      int num_curr_point, int num_temp_point, int ref_dim, long long raw_row_stride,
                      void *loc_curr, void *loc_temp, void *aw_curr, void *aw_temp, void *stream)
{
    g_err[0] = 0; g_route[0] = 0;
    PrepParams p;
    memset(&p, 0, sizeof(p));
    p.off_c = offsets_curr; p.off_t = offsets_temp; p.logit_c = logits_curr; p.logit_t = logits_temp;
    p.ref_c = ref_curr; p.ref_t = ref_temp; p.shapes = spatial_shapes;
    p.loc_c = loc_curr; p.loc_t = loc_temp; p.aw_c = aw_curr; p.aw_t = aw_temp;
    p.rows = rows; p.M = num_heads; p.L = num_levels; p.W = window; p.Pc = num_curr_point;
    p.Pt = window > 0 ? num_temp_point : 1; p.d = ref_dim; p.ld = raw_row_stride;
    if (rows > 0 && (!offsets_curr || !logits_curr || !loc_curr || !aw_curr ||
                     (window > 0 && (!offsets_temp || !logits_temp || !loc_temp || !aw_temp))))
        return fail(MSDA_ERR_ARG, "msda_prep_forward: null pointer argument%s");
    return run_prep(dtype, p, false, stream);
}

int msda_prep_backward(int dtype, const void *grad_loc_curr, const void *grad_loc_temp, const void *grad_aw_curr,
                       const void *grad_aw_temp, const void *aw_curr, const void *aw_temp, const void *ref_curr,
                       const void *ref_temp, const int64_t *spatial_shapes, long long rows, int num_heads,
                       int num_levels, int window, int num_curr_point, int num_temp_point, int ref_dim,
                       long long raw_row_stride, void *grad_offsets_curr, void *grad_offsets_temp,
                       void *grad_logits_curr, void *grad_logits_temp, void *stream)
{
    g_err[0] = 0; g_route[0] = 0;
    PrepParams p;
    memset(&p, 0, sizeof(p));
    p.gloc_c = grad_loc_curr; p.gloc_t = grad_loc_temp; p.gaw_c = grad_aw_curr; p.gaw_t = grad_aw_temp;
    p.aw_c = const_cast<void *>(aw_curr); p.aw_t = const_cast<void *>(aw_temp);
    p.ref_c = ref_curr; p.ref_t = ref_temp; p.shapes = spatial_shapes;
    p.goff_c = grad_offsets_curr; p.goff_t = grad_offsets_temp; p.glogit_c = grad_logits_curr; p.glogit_t = grad_logits_temp;
    p.rows = rows; p.M = num_heads; p.L = num_levels; p.W = window; p.Pc = num_curr_point;
    p.Pt = window > 0 ? num_temp_point : 1; p.d = ref_dim; p.ld = raw_row_stride;
    if (rows > 0 && (!grad_loc_curr || !grad_aw_curr || !aw_curr || !grad_offsets_curr || !grad_logits_curr ||
                     (window > 0 && (!grad_loc_temp || !grad_aw_temp || !aw_temp || !grad_offsets_temp || !grad_logits_temp))))
        return fail(MSDA_ERR_ARG, "msda_prep_backward: null pointer argument%s");
    return run_prep(dtype, p, true, stream);
}

int msda_mask_rows(int dtype, void *rows, const void *padding_mask, long long pixels, long long row_elems,
                   long long row_stride, void *stream)
{
    g_err[0] = 0; g_route[0] = 0;
    const int e = dtype == MSDA_F32 ? 4 : dtype == MSDA_F64 ? 8 : (dtype == MSDA_BF16 || dtype == MSDA_F16) ? 2 : 0;
    if (!e) return fail(MSDA_ERR_DTYPE, "msda: unknown dtype code%s");
    if (pixels < 0 || row_elems <= 0 || row_stride < row_elems)
        return fail(MSDA_ERR_ARG, "msda_mask_rows: bad sizes (pixels, row elements, row stride)%s");
    if (pixels == 0) return MSDA_OK;
    if (!rows || !padding_mask) return fail(MSDA_ERR_ARG, "msda_mask_rows: null pointer argument%s");
    const long long rb = row_elems * e, sb = row_stride * e;
    const int g = (rb % 16 == 0 && sb % 16 == 0 && (reinterpret_cast<uintptr_t>(rows) & 15) == 0) ? 16 : e;
    const long long chunks = rb / g, threads = pixels * chunks;
    if (chunks > 0x7fffffffLL || (threads + 255) / 256 > 0x7fffffffLL)
        return fail(MSDA_ERR_ARG, "msda_mask_rows: tensor too large for one launch%s");
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *r = static_cast<char *>(rows);
    const uint8_t *m = static_cast<const uint8_t *>(padding_mask);
    switch (g) {
        case 16: hipLaunchKernelGGL(msda_mask_rows_kernel<16>, grid, block, 0, st, r, m, pixels, (int)chunks, sb); break;
        case 8: hipLaunchKernelGGL(msda_mask_rows_kernel<8>, grid, block, 0, st, r, m, pixels, (int)chunks, sb); break;
        case 4: hipLaunchKernelGGL(msda_mask_rows_kernel<4>, grid, block, 0, st, r, m, pixels, (int)chunks, sb); break;
        default: hipLaunchKernelGGL(msda_mask_rows_kernel<2>, grid, block, 0, st, r, m, pixels, (int)chunks, sb); break;
    }
    return check_launch("msda mask rows");
}

}  // extern "C"
