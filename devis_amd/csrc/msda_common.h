// msda_common.h -- shared by the translation units of libmsda_hip.so (devis_amd/csrc/*.hip, one per kernel family):
// storage-type helpers, the kernel parameter block, tap arithmetic, launch geometry constants, and the host-side
// services (error / route strings, test knobs, per-device caches) and per-family launchers the dispatcher in
// msda_api.hip calls.  gfx950 only; no CUDA compatibility layer.
//
// What the kernels compute (semantics of the reference kernels being replaced,
// /root/reference/src/models/ops/src/cuda/ms_deform_im2col_cuda.cuh:33-299 forward, :87-920 backward):
//   out[n,q,m,:] = sum_{l,p} a[n,q,m,l,p] * bilinear(value_l[:, :, m, :], x*W_l-0.5, y*H_l-0.5)
// with zero padding, a point contributing only if -1 < h < H_l and -1 < w < W_l (cuh:288), and the matching
// gradients wrt value (scatter-add), sampling locations and attention weights.
#pragma once

#include <hip/hip_runtime.h>
#include <utility>
#include <type_traits>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

// Timing-only experiment macros (MSDA_RS_EXP, MSDA_WIN_EXP, MSDA_MFMA_EXP: kernels that skip part of their work and return WRONG RESULTS, used to
// measure floors -- profiles/NEGATIVE_RESULTS.md) may only be compiled into a library that says what it is: the build must also
// define MSDA_TIMING_ONLY_BUILD, which makes msda_build_info() / msda_last_route() announce it and every entry point refuse to
// run unless MSDA_ENABLE_HOOKS=1 (msda_api.hip).  A stray -D alone does not compile.
#if defined(MSDA_RS_EXP) || defined(MSDA_WIN_EXP) || defined(MSDA_MFMA_EXP) || defined(MSDA_MFMA_TRACE)
#  if !defined(MSDA_TIMING_ONLY_BUILD)
#    error "MSDA_RS_EXP / MSDA_WIN_EXP / MSDA_MFMA_EXP produce wrong results by construction: also pass -DMSDA_TIMING_ONLY_BUILD (the library then refuses to run without MSDA_ENABLE_HOOKS=1 and labels its routes)"
#  endif
#endif
#if defined(MSDA_TIMING_ONLY_BUILD)
#  define MSDA_IS_TIMING_ONLY 1
#else
#  define MSDA_IS_TIMING_ONLY 0
#endif


#include "msda.h"

// No implicit FMA contraction anywhere in the library: HIP's __fmul_rn / __fsub_rn are plain `*` / `-` unless
// OCML_BASIC_ROUNDED_OPERATIONS is defined, and hipcc contracts by default, so `x * W - 0.5` could become one
// FMA in some inlining contexts and not in others -- forward, gather pass and scatter would then disagree with
// each other (and with the oracle) about the pixel cell of a point that sits within one ulp of a border.
// Every FMA the kernels want is spelled fmaf().  (devis_amd/build.py also passes -ffp-contract=off.)
#pragma clang fp contract(off)

namespace msda {


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
    __device__ static bf16_t cvt(float v) { return __float2bfloat16(v); }
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
    __device__ static f16_t cvt(float v) { return __float2half(v); }
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
    int gv_storage;             // bwd: grad_value is in the STORAGE type (16-bit), not the arithmetic type (owner-computes scatter only)
    int own_levels;             // bwd: the owner-computes scatter handles levels [0, own_levels); the trailing (coarse) levels are
                                // the matrix-pipe scatter's (msda_mfma.hip).  = L when that kernel does not run
    unsigned rec_mask;          // bwd: bit l = the gather pass leaves culling records for level l and the owner-computes scatter reads
                                // them; clear for the matrix-pipe levels and for levels that are ONE band (every group a candidate)
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
// small shared helpers
// ------------------------------------------------------------------------------------------------
typedef unsigned long long u64;
constexpr int kNoRow16 = -32768;            // culling record of a point that touches no row (see the gather passes)
constexpr int kSlabMaxLevels = 32;          // levels the resident-slab kernels keep tables for
constexpr int kScatterMaxLevels = 32;
constexpr int kOwnMaxSorted = 256;            // (level, band) pairs the owner-computes scatter sorts by position (more: level order)
constexpr int kScatterMaxSources = 64;      // 1 + frames * window must fit
constexpr int kCullBlock = 64;              // queries per block summary of the culling records
constexpr int kLiveWords = 64;              // up to 2048 cull batches per item take the block-summary pre-pass

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

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F &&f)
{
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
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

// ------------------------------------------------------------------------------------------------
// launch geometry shared by kernels and dispatcher
// ------------------------------------------------------------------------------------------------
constexpr int kRowSlots = kPch + 1;         // tile kernels: 16-byte record slots per row (odd: LDS banks)
constexpr int kTileMaxWaves = 8;            // forward tile kernel: waves of one workgroup that share a tile (small calls)
// owner-computes scatter
#ifndef MSDA_OWN_THREADS
#define MSDA_OWN_THREADS 1024
#endif
#ifndef MSDA_OWN_SLOTS
#define MSDA_OWN_SLOTS 4
#endif
constexpr int kOwnThreads = MSDA_OWN_THREADS;       // (512: two workgroups per CU out of phase with each other)
constexpr int kOwnQuads = kOwnThreads / 4;
constexpr int kOwnSlots = MSDA_OWN_SLOTS;           // pixels per owner quad
constexpr int kOwnPix = kOwnQuads * kOwnSlots;      // pixels per band
// resident-slab kernels
#ifndef MSDA_RS_THREADS
#define MSDA_RS_THREADS 1024        // (512: 8 waves with a 256-VGPR budget each -- measured slower)
#endif
constexpr int kRsThreads = MSDA_RS_THREADS, kRsWaves = kRsThreads / kWave;
constexpr int kRsRows = kWave / 4;       // rows per wave tile: one quad per row
constexpr int kRsSlack = 1024;          // bytes: the last LDS-DMA piece may overrun the slab's pixels
#ifndef MSDA_RS_MAXFRAMES
#define MSDA_RS_MAXFRAMES 32
#endif
#ifndef MSDA_RS_LDS_BYTES
#define MSDA_RS_LDS_BYTES (160 * 1024)  // (experiments: 80 KiB with MSDA_RS_THREADS=512, MSDA_RS_MIN_WAVES=4, MSDA_RS_MAXFRAMES=8 = two workgroups per CU on 16-bit slabs)
#endif
#ifndef MSDA_RS_MIN_WAVES
#define MSDA_RS_MIN_WAVES 1
#endif
constexpr int kRsMaxFrames = MSDA_RS_MAXFRAMES;        // frames x frames slot masks live in LDS
constexpr int kRsRowB = 128;            // bytes of one pixel of one head in a 4-byte type (D = 32); 64 in a 2-byte type
constexpr int kRsTailBytes = kRsRowB + kRsMaxFrames * kRsMaxFrames * 4 + 4 * kSlabMaxLevels * 4 + 16;   // after the slab
constexpr int kRsSlabBytes = ((MSDA_RS_LDS_BYTES - 256 - kRsTailBytes) / 128) * 128;

// ---- resident-window kernels (msda_win.hip): encoder-shaped calls, where query i IS pixel i of the pyramid and samples round
// its own position.  A workgroup owns the queries of one spatial TILE (By x Bx level-0 pixels and the pixels of the other
// levels whose centres fall into it) and stages, per source frame, a WINDOW of every level -- the tile's footprint on that
// level plus `halo` pixels on every side -- in LDS; taps outside the window are read from memory (any input is computed
// exactly; only speed depends on locality).  When all windows do not fit at a useful halo the levels are staged in two
// phases per source frame: levels < split, then levels >= split (split = 0: one phase).
constexpr int kWinMaxLevels = 8;
struct WinPlan {
    int By, Bx, tiles_y, tiles_x;       // tile size in level-0 pixels; tiles of one map
    int split;                          // first level of the second staging phase (0 = one phase)
    int halo[2];                        // pixels round the footprint, per phase
    int wbase[kWinMaxLevels];           // first LDS pixel of level l's window (inside its phase's layout)
    int tpg;                            // wave tiles (16 rows) per query frame: ceil(most queries of a tile / 16)
    int nt;                             // wave tiles per wave: ceil(frames * tpg / 16 waves) <= 3
};

// One axis of a tile's geometry on a level with n_l pixels (n_0 on level 0): the level's own pixels the tile OWNS
// [q0, q1) -- pixel y belongs to the tile its centre falls into, i.e. floor((2y + 1) n_0 / (2 n_l) / B) -- and the window
// [w0, w1) that holds every tap of a point within `halo` pixels of the tile's extent (floor(y n_l - 0.5) and the row below).
// Host (planning: maxima over the tiles) and device (the tile's own tables) evaluate the same integers.
__host__ __device__ inline int win_fdiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
__host__ __device__ inline void win_axis(int n_l, int n_0, int t, int B, int halo, int &q0, int &q1, int &w0, int &w1)
{
    const int a = 2 * n_l * t * B - n_0, b = 2 * n_l * (t + 1) * B - n_0, d = 2 * n_0;
    auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    q0 = clampi(-win_fdiv(-a, d), 0, n_l);          // ceil
    q1 = clampi(-win_fdiv(-b, d), 0, n_l);
    w0 = clampi(win_fdiv(a, d) - halo, 0, n_l);
    w1 = clampi(win_fdiv(b, d) + 2 + halo, 0, n_l);
}

constexpr int kWinTailBytes = 128 + kRsMaxFrames * kRsMaxFrames * 4 + 14 * kWinMaxLevels * 4 + 64;    // zero row, slot masks, level tables
constexpr int kWinSlabBytes = ((160 * 1024 - 256 - kWinTailBytes) / 128) * 128;

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

// ------------------------------------------------------------------------------------------------
// host side (defined in msda_api.hip)
// ------------------------------------------------------------------------------------------------
int fail(int code, const char *fmt, const char *detail = "");
int check_launch(const char *what);                 // appends `what` to the route string, reports launch errors
int device_cus();
constexpr int kMaxDevices = 64;
// Dynamic LDS above 64 KiB must be opted into per kernel function AND per device; `granted` is the caller's
// per-instantiation table of what each device has been given so far.
struct LdsGrant { size_t bytes[kMaxDevices]; };
int grant_lds(const void *kernel, size_t bytes, LdsGrant &granted, const char *what);

// Storage-type combinations of a call (include/msda.h msda_dtype): T = value / out / grad_out, TL = sampling_loc /
// attn_weight and their gradients (TL = float with a 16-bit T for MSDA_BF16_LOC32 / MSDA_F16_LOC32).  f(type_tag<T>, type_tag<TL>).
template <typename X> struct type_tag { using type = X; };
template <class F>
int dispatch_types(int dtype, F &&f)
{
    switch (dtype) {
        case MSDA_F32: return f(type_tag<float>{}, type_tag<float>{});
        case MSDA_BF16: return f(type_tag<bf16_t>{}, type_tag<bf16_t>{});
        case MSDA_F16: return f(type_tag<f16_t>{}, type_tag<f16_t>{});
        case MSDA_BF16_LOC32: return f(type_tag<bf16_t>{}, type_tag<float>{});
        case MSDA_F16_LOC32: return f(type_tag<f16_t>{}, type_tag<float>{});
        default: return fail(MSDA_ERR_DTYPE, "msda: unknown dtype code%s");
    }
}

// ---- per-family launchers (each translation unit owns its kernels; MSDA_OK or a negative msda_status) ----
// msda_tile.hip: G = D / (16 / sizeof(T)) lanes per row, one wave per workgroup
int launch_fwd_tile(int dtype, int G, const Params &p, unsigned blocks, size_t lds, hipStream_t stream, int waves = 1);
int launch_bwd_tile(int dtype, int G, bool atomics, const Params &p, unsigned blocks, size_t lds, hipStream_t stream);
// msda_rs.hip: resident-slab kernels (D = 32); nt = tiles per wave of the forward (1, 2, 4); first_slab_level = the host's
// guess of the first pyramid level inside the slab (1 or 2: picks the kernel whose software-pipelined slot body is compiled
// for it -- speed only: a kernel whose device-side level differs falls back to its plain loop)
int launch_fwd_rs(int dtype, int nt, int first_slab_level, const Params &p, int parts, unsigned grid, hipStream_t stream);
int launch_bwd_rs(int dtype, int first_slab_level, const Params &p, int parts, unsigned grid, hipStream_t stream, int frame_split = 0);
int launch_fwd_win(int dtype, const Params &p, const WinPlan &w, hipStream_t stream);
int launch_bwd_win(int dtype, const Params &p, const WinPlan &w, hipStream_t stream);
// msda_scatter.hip: grad_value
int launch_zero_unowned(const Params &p, int cap_slots, int grad_value_elem_bytes, hipStream_t stream);
int launch_cull_summary(const Params &p, hipStream_t stream);
int launch_scatter_lds(int dtype, int G, const Params &p, unsigned grid, int cap_bytes, int dbg, hipStream_t stream);
int launch_scatter_grp(int dtype, bool storage_typed_grad_value, const Params &p, unsigned grid, int dbg, hipStream_t stream);
// msda_mfma.hip: grad_value of the trailing levels [l0, L) (at most 2, `tiles` = mfma_scatter_tiles(their pixels)) on the matrix pipe
int mfma_scatter_tiles(long long pixels);
int launch_scatter_mfma(int dtype, bool storage_typed_grad_value, const Params &p, int l0, int tiles, hipStream_t stream);
// msda_generic.hip: any-shape kernels, the modules' fused pre-op pass, the padding-mask pass
int launch_generic(int dtype, const Params &p, bool bwd, hipStream_t stream);
int launch_prep(int dtype, const PrepParams &p, bool bwd, hipStream_t stream);
int launch_mask_rows(int bytes_per_thread, char *rows, const uint8_t *mask, long long pixels, int chunks,
                     long long stride_bytes, unsigned blocks, hipStream_t stream);

}  // namespace msda
