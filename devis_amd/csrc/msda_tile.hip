// msda_tile.hip -- "tile" kernels: forward and backward gather pass for every shape the resident-slab kernels
// (msda_rs.hip) do not take, and the one-kernel backward with global float atomics.
//
//   * ONE wave64 owns RPW = 64/G (query, head) rows of the same head, G lanes per row, each lane holding VEC
//     contiguous channels (16 B: float4 or 8 x bf16/f16), so every bilinear corner is one coalesced D*sizeof(T)
//     segment per row and one 16-B load per lane.
//   * the wave first turns its rows' (x, y, weight) triples into "tap records" in LDS -- 4 element offsets + 4
//     premultiplied weights per sampling point, computed ONCE per point instead of once per channel lane -- then
//     the gather loop is LDS-broadcast read + 4 global loads + FMAs; out-of-range corners are (offset 0 / out of
//     the buffer, weight 0): the loop is branch free.
//   * a per-wave "virtual level" table in LDS holds (H, W, first pixel) for every level of every source frame, so
//     the plain op and the fused temporal op (current frame + `window` other frames of the clip,
//     ms_deform_attn.py:325-364) are the SAME kernel.
//   * blockIdx -> (query tile, head) with head = blockIdx % M: workgroups are dealt round-robin to the 8 XCDs, so
//     with M = 8 each XCD's private 4 MiB L2 only ever sees ONE head's 1/8 slice of the value maps (speed only).
//   * backward gather pass: per-point partial dot products <grad_out, corner_k> are reduced across the G lanes
//     with DPP butterflies (no LDS round trip, no serial thread-0 sum as in cuh:376-394), lane pp % G keeps the
//     sums of point pp and G points are finished at once; it also leaves the culling records of the scatter.
#include "msda_common.h"

namespace msda {
namespace {

// ------------------------------------------------------------------------------------------------
// tile kernels
// ------------------------------------------------------------------------------------------------
// LDS carve (dynamic, 16-byte aligned): [s_off RPW*(kPch+1) int4][s_w RPW*(kPch+1) float4]
//                                       [s_e RPW*(kPch+1) float4 (bwd only)][levels nvl * Level]
// Row stride kPch+1 (odd number of 16-B slots) keeps the RPW rows of a wave on different LDS slots
// for the broadcast ds_read_b128 of the gather loop.

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
template <typename T, int RPW, bool BWD, typename TL>      // T: value type (out-of-map offset), TL: type of the chunk's loc / attn
__device__ __forceinline__ void build_chunk(const Params &p, const ChunkRef<TL> &c,
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
// MW (round 4): SMALL calls -- one clip at the query counts of DeVIS's shipped configs (60 / 180 per frame) is a few hundred
// single-wave workgroups on 1024 SIMDs, each walking its rows' 96 points as ~24 dependent batches of gathers: latency-bound.  With
// MW a workgroup is blockDim.x / 64 waves on the SAME tile: wave w takes the chunks w, w + nw, ... (its own record slots in LDS,
// no workgroup barrier inside the loop) and the partial rows are added up through LDS in wave order at the end -- the same sums
// on every run.  The host uses it (3 waves) while the workgroups are fewer than the SIMDs (msda_api.hip): 60 queries 0.020 ->
// 0.013 ms (fp16 0.034 -> 0.015); at 300 queries in fp32 (1824 workgroups) it changes nothing.
template <typename T, typename TL, int G, int NB, bool MW>       // T: value / out, TL: sampling_loc / attn_weight
__global__ void __launch_bounds__(MW ? kWave * kTileMaxWaves : kWave, 4)
msda_fwd_tile_kernel(const Params p)
{
    constexpr int VEC = Store<T>::VEC;
    constexpr int RPW = kWave / G;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x % kWave;
    const int wave = MW ? __builtin_amdgcn_readfirstlane(threadIdx.x / kWave) : 0, nw = MW ? (int)blockDim.x / kWave : 1;
    int4 *s_off = reinterpret_cast<int4 *>(lds_raw) + wave * (RPW * kRowSlots);
    float4 *s_w = reinterpret_cast<float4 *>(reinterpret_cast<int4 *>(lds_raw) + nw * (RPW * kRowSlots)) + wave * (RPW * kRowSlots);
    Level *s_lvl = reinterpret_cast<Level *>(reinterpret_cast<int4 *>(lds_raw) + 2 * nw * (RPW * kRowSlots));

    int m, group, q0;
    tile_coords<RPW>(p, m, group, q0);
    const int clip = group / p.frames, t = group - clip * p.frames;
    const int nvl = p.LA + p.LB;
    for (int j = threadIdx.x; j < nvl; j += (int)blockDim.x) s_lvl[j] = make_level(p, t, j);
    __syncthreads();
    // the records of a chunk are written and read by ONE wave: its LDS operations complete in order, so with MW a compiler
    // fence + s_waitcnt replaces the workgroup barrier (the waves of a workgroup run different numbers of chunks)
    auto chunk_sync = [&]() {
        if constexpr (MW) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else __syncthreads();
    };

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
    for (int ci = wave; ci < n_all; ci += nw) {
        const ChunkRef<TL> c = get_chunk<TL>(p, ci, nA);
        load_chunk<TL, RPW>(p, c, row0, rows_valid, lane, st);
        build_chunk<T, RPW, false>(p, c, st, s_lvl, s_off, s_w, nullptr, lane);
        chunk_sync();
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
        chunk_sync();
    }
    if constexpr (MW) {
        float *s_red = reinterpret_cast<float *>(s_lvl + nvl);          // [nw][64][VEC]
#pragma unroll
        for (int c = 0; c < VEC; ++c) s_red[(wave * kWave + lane) * VEC + c] = acc[c];
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[c] = s_red[lane * VEC + c];
        for (int w = 1; w < nw; ++w)
#pragma unroll
            for (int c = 0; c < VEC; ++c) acc[c] += s_red[(w * kWave + lane) * VEC + c];
    }
    if (r < rows_valid) {
        T *out = static_cast<T *>(p.out) + (row0 + (int64_t)r * p.M) * p.D + sub * VEC;
        Store<T>::store(out, acc);
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

// ATOMICS = true : also scatters grad_value with global float atomics (one-kernel backward; used when
//                  the LDS scatter kernel below cannot take the shape).
// ATOMICS = false: computes grad_sampling_loc / grad_attn_weight only; grad_value comes from
//                  msda_bwd_value_lds_kernel.
template <typename T, typename TL, int G, bool ATOMICS>      // T: value / grad_out, TL: sampling_loc / attn_weight + gradients
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
            const ChunkRef<TL> c = get_chunk<TL>(p, ci, nA);
            load_chunk<TL, RPW>(p, c, row0, rows_valid, lane, st);
            TL *gloc = static_cast<TL *>(c.arr ? p.glocB : p.glocA);
            TL *gaw = static_cast<TL *>(c.arr ? p.gawB : p.gawA);
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
                    Store<TL>::put(gloc + 2 * idx0 + el, res[(el >> 1) * 4 + (el & 1)]);
                for (int el = sub; el < np; el += G)
                    Store<TL>::put(gaw + idx0 + el, res[el * 4 + 2]);
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


template <typename T, typename TL, int G>
int fwd_tile(const Params &p, unsigned blocks, size_t lds, hipStream_t stream, int waves)
{
    // points whose 4 * NB corner loads are in flight: 4 for 4-byte types; 2 for 2-byte types, whose lanes hold 8 channels
    // (4 points x 4 corners x 8 fp32 channels would be the whole register budget)
    constexpr int NB = sizeof(T) == 4 ? 4 : 2;
    if constexpr (G == 4 || G == 8) {          // (D = 32: the only rows small calls were measured on)
        if (waves > 1) {
            hipLaunchKernelGGL((msda_fwd_tile_kernel<T, TL, G, NB, true>), dim3(blocks), dim3(kWave * waves), lds, stream, p);
            return check_launch("msda forward (tile kernel, several waves per tile)");
        }
    }
    hipLaunchKernelGGL((msda_fwd_tile_kernel<T, TL, G, NB, false>), dim3(blocks), dim3(kWave), lds, stream, p);
    return check_launch("msda forward (tile kernel)");
}

template <typename T, typename TL, int G>
int bwd_tile(bool atomics, const Params &p, unsigned blocks, size_t lds, hipStream_t stream)
{
    if (atomics) {
        hipLaunchKernelGGL((msda_bwd_tile_kernel<T, TL, G, true>), dim3(blocks), dim3(kWave), lds, stream, p);
        return check_launch("msda backward (tile kernel, global atomics)");
    }
    hipLaunchKernelGGL((msda_bwd_tile_kernel<T, TL, G, false>), dim3(blocks), dim3(kWave), lds, stream, p);
    return check_launch("msda backward (tile kernel, grad_loc/grad_attn)");
}

// G (lanes per row) -> instantiation
template <class F>
int by_lanes(int G, F &&f)
{
    switch (G) {
        case 1: return f(std::integral_constant<int, 1>{});
        case 2: return f(std::integral_constant<int, 2>{});
        case 4: return f(std::integral_constant<int, 4>{});
        case 8: return f(std::integral_constant<int, 8>{});
        case 16: return f(std::integral_constant<int, 16>{});
        case 32: return f(std::integral_constant<int, 32>{});
        case 64: return f(std::integral_constant<int, 64>{});
        default: return fail(MSDA_ERR_ARG, "msda: unsupported lanes per row%s");
    }
}

}  // namespace

int launch_fwd_tile(int dtype, int G, const Params &p, unsigned blocks, size_t lds, hipStream_t stream, int waves)
{
    return dispatch_types(dtype, [&](auto t, auto tl) {
        return by_lanes(G, [&](auto g) {
            return fwd_tile<typename decltype(t)::type, typename decltype(tl)::type, decltype(g)::value>(p, blocks, lds, stream, waves);
        });
    });
}

int launch_bwd_tile(int dtype, int G, bool atomics, const Params &p, unsigned blocks, size_t lds, hipStream_t stream)
{
    return dispatch_types(dtype, [&](auto t, auto tl) {
        return by_lanes(G, [&](auto g) {
            return bwd_tile<typename decltype(t)::type, typename decltype(tl)::type, decltype(g)::value>(atomics, p, blocks, lds, stream);
        });
    });
}

}  // namespace msda
