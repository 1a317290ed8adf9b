// msda_scatter.hip -- grad_value: the scatter half of the backward.
#include "msda_common.h"
#include <algorithm>

#ifndef MSDA_GRP_F32
#define MSDA_GRP_F32 512
#endif
#ifndef MSDA_GRP_16
#define MSDA_GRP_16 512
#endif

namespace msda {
namespace {

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
constexpr int kScatterList = 3072;         // capacity of the survivor list (12 KiB of the 16 KiB LDS left by the band)

template <typename T, typename TL, int G>        // T: grad_out, TL: sampling_loc / attn_weight
__global__ void __launch_bounds__(kScatterThreads)
msda_bwd_value_lds_kernel(const Params p, int cap_slots, int dbg)
{
    constexpr int VEC = 4;          // channels per lane, whatever the storage type: G = D / 4 lanes per hit (a lane with the 8
                                    // channels of a 16-byte 2-byte-type vector carried 32 LDS adds per hit and spilled)
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
        auto source_of = [&](int k, int &t, int &vl, int &vlg, int &P, int &LP, const TL *&loc, const TL *&aw) {
            t = s_src_t[k]; vl = s_src_vl[k];
            const bool cur = (k == 0);
            vlg = cur ? vl : p.LA + vl;
            P = cur ? p.PA : p.PB;
            LP = cur ? p.LA * p.PA : p.LB * p.PB;
            loc = static_cast<const TL *>(cur ? p.locA : p.locB);
            aw = static_cast<const TL *>(cur ? p.awA : p.awB);
        };
        // one candidate per lane per pass; the NEXT pass's (x, y, attn) are loaded before this pass's
        // hits are processed, so the scan's memory latency hides behind stage 2
        auto fetch = [&](int i, float &x, float &y, float &a, int &qrow) {
            x = y = -10.f; a = 0.f; qrow = 0;
            if (i < n_cand) {
                const int ei = pshift >= 0 ? (i >> pshift) : i / Pmax, pt = i - ei * Pmax;
                const int e = s_list[ei];
                int t, vl, vlg, P, LP;
                const TL *loc, *aw;
                source_of(e >> 24, t, vl, vlg, P, LP, loc, aw);
                if (pt < P) {
                    const int64_t gq = ((int64_t)clip * p.frames + t) * p.Lq + (e & 0xffffff);
                    const int64_t idx = (gq * p.M + m) * LP + vl * P + pt;
                    x = Store<TL>::get(loc + 2 * idx);
                    y = Store<TL>::get(loc + 2 * idx + 1);
                    a = Store<TL>::get(aw + idx);
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
                    const TL *loc, *aw;
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

// Coarse culling summary for long candidate ranges (encoder shapes, Lq = S): (min, max) top tap row over blocks
// of kCullBlock consecutive queries of every (group, head, virtual level), reduced from the per-point records
// the gather pass left.  One wave per block.
__global__ void __launch_bounds__(256)
msda_cull_summary_kernel(const Params p)
{
    const int VL = p.LA + p.LB, nblk = (p.Lq + kCullBlock - 1) / kCullBlock;
    const int64_t total = (int64_t)p.groups * p.M * VL * nblk;
    const int lane = threadIdx.x % kWave;
    for (int64_t e = (int64_t)blockIdx.x * 4 + threadIdx.x / kWave; e < total; e += (int64_t)gridDim.x * 4) {
        const int64_t gmv = e / nblk;
        const int vl = (int)(gmv % VL), lvl = vl < p.LA ? vl : (vl - p.LA) % p.L;
        if (!((p.rec_mask >> lvl) & 1u)) continue;  // (no records were left for this level: Params::rec_mask)
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
constexpr unsigned kOwnNil = 0xffffffffu;

// Timeline probe (-DMSDA_SCATTER_TRACE, experimental builds only): lane 0 of wave 1 of the first 8 workgroups stamps the shader
// clock at phase boundaries; scripts/scatter_trace.py reads the stamps back through msda_debug_trace and sums the phases.
#ifdef MSDA_SCATTER_TRACE
constexpr int kTraceLen = 8192;
__device__ unsigned long long g_trace[8][kTraceLen];
__device__ int g_trace_n[8];
#define MSDA_TR(id) do { if (tr_on && tr_n < kTraceLen) g_trace[blockIdx.x][tr_n++] = ((unsigned long long)__builtin_readcyclecounter() << 8) | (unsigned)(id); } while (0)
#else
#define MSDA_TR(id) do { } while (0)
#endif

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
// (rows are staged as fp32 for every storage type: 128 bytes per group)
template <typename T> constexpr int grp_lds_bytes()
{
    return grp_chunk<T>() * 128 + 128 + 16 * grp_chunk<T>() * 8 + kOwnPix * 4 + kGrpList * 4;
}

// T = storage type of loc / attn / grad_out; GV = type of grad_value as written (float, or T: include/msda.h grad_value_dtype).
// The grad_out rows are staged in LDS as FP32 whatever T is: a 16-bit row would have to be unpacked once per list entry in
// the walk (16 entries read each row: measured 0.29 ms of walk for bf16 against 0.185 for fp32 on the bench workload),
// now it is converted once, on its way in (4-byte types keep the LDS-DMA; 2-byte types go through registers).
// A value the compiler must recompute where it is used (keeps loop-invariant per-thread addresses out of long-lived registers).
__device__ __forceinline__ int per_item(int x)
{
    asm volatile("" : "+v"(x));
    return x;
}

template <typename T, typename TL, typename GV, bool SORTED>        // SORTED: items in image order (long candidate ranges); TL: storage type of sampling_loc / attn_weight (T, or float with a 16-bit T)
__global__ void __launch_bounds__(kOwnThreads, 4)
msda_bwd_value_grp_kernel(const Params p, int dbg)
{
    constexpr int D = 32, kRowB = D * 4;                        // bytes of one staged grad_out row (fp32)
    constexpr int kOwnChunk = grp_chunk<T>();                   // groups per chunk
    constexpr int kPasses = (4 * kOwnChunk + kOwnThreads - 1) / kOwnThreads;
    constexpr bool kHalf = sizeof(T) == 2;
    extern __shared__ __attribute__((aligned(128))) unsigned char lds_raw[];
    unsigned char *rows = lds_raw;                                              // [kOwnChunk][kRowB]  one row per group
    // [16 * kOwnChunk] {weight bits, next reference}, on a 128-byte boundary of the LDS address space (the walk derives an
    // entry's row from its address); a reference = the absolute LDS address of an entry
    uint2 *ents = reinterpret_cast<uint2 *>(lds_raw + kOwnChunk * kRowB +
                                            ((128u - (lds_addr(lds_raw) & 127u)) & 127u));
    unsigned *head = reinterpret_cast<unsigned *>(ents + 16 * kOwnChunk);       // [kOwnPix]
    unsigned *list = head + kOwnPix;                                            // [kGrpList] (k:6 | points:4 | q:22)
    __shared__ int s_H[kScatterMaxLevels], s_W[kScatterMaxLevels], s_R[kScatterMaxLevels],
        s_first[kScatterMaxLevels + 1], s_lsi[kScatterMaxLevels];
    __shared__ int s_cnt[3];             // survivor counters rotate: slot j is reset two barriers before it is used again
    // Item descriptors and source tables are double-buffered: wave 0 prepares item n + 1 (ticket, decode, tables) in the shadow
    // of item n's first culling records, so an item starts without a barrier or a division of its own.
    __shared__ int s_desc2[2][8];        // valid, level, band, head, frame, clip
    __shared__ int s_nsrc2[2];
    __shared__ long long s_src_tab2[2][kScatterMaxSources], s_src_loc2[2][kScatterMaxSources];
    __shared__ int s_src_q02[2][kScatterMaxSources], s_src_gmv2[2][kScatterMaxSources];
    __shared__ unsigned s_live[kLiveWords];            // bitmap of the cull batches that hold a live 64-query block
    __shared__ int s_ftab[kWave];                      // the frame table (<= 64 slots: scatter_applicable), read once
    __shared__ unsigned short s_order[SORTED ? kOwnMaxSorted : 1];  // bands of all levels sorted by where they start in the image (see below)
    __shared__ float s_key[SORTED ? kOwnMaxSorted : 1];

    const int tid = threadIdx.x, lane = tid % kWave;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int MD = p.M * D, L = p.L, VL = p.LA + p.LB;
#ifdef MSDA_SCATTER_TRACE
    const bool tr_on = blockIdx.x < 8 && tid == 64;
    int tr_n = 0;
#endif
    if (tid == 0) {
        int first = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)p.shapes[2 * l], W = (int)p.shapes[2 * l + 1];
            const int R = min(H, kOwnPix / max(1, W));          // rows per band; 0 = "direct" level (row wider than a band)
            s_H[l] = H; s_W[l] = W; s_R[l] = R; s_lsi[l] = (int)p.lsi[l];
            s_first[l] = first;
            first += l >= p.own_levels ? 0 : (R > 0) ? (H + R - 1) / R : 1;       // (levels from own_levels on: no items here, msda_mfma.hip)
        }
        s_first[L] = first;
        s_cnt[0] = s_cnt[1] = s_cnt[2] = 0;
    }
    if (tid < kWave) s_ftab[tid] = tid < p.frames * p.window ? p.ftab[tid] : -1;
    int ci = 0;                             // counter of the current cull batch
    for (int i = tid; i < kOwnPix; i += kOwnThreads) head[i] = kOwnNil;
    __syncthreads();
    const int NB = s_first[L];
    if (dbg & 512) {
        // the zero-fill of msda_zero_unowned_kernel, dealt over this kernel's threads: pixels of `value` rows outside every level
        // (spatial_shapes that do not tile [0, S)); the host only sets the bit when no level takes the float-atomic branch, so
        // nothing else in this launch writes these pixels
        const int64_t total = (int64_t)p.groups * p.S;
        for (int64_t idx = (int64_t)blockIdx.x * kOwnThreads + tid; idx < total; idx += (int64_t)gridDim.x * kOwnThreads) {
            const int sp = (int)(idx % p.S);
            bool inside = false, wide = false;
            for (int l = 0; l < L; ++l)
                if (sp >= s_lsi[l] && sp < s_lsi[l] + s_H[l] * s_W[l]) { inside = true; wide = s_R[l] == 0; }
            if (inside && !wide) continue;
            // `wide`: the host's copy of the shapes (a selection HINT, include/msda.h) said every level fits a band, so nothing
            // zero-filled grad_value for the float-atomic branch -- but the DEVICE shapes have a level wider than a band: the hint
            // was stale.  Sums added into the unzeroed buffer would be silently wrong; the level's pixels are poisoned with NaN
            // instead (the atomic adds of its items, before or after this store, leave NaN), so that the caller's mistake shows in
            // the first value it reads.  (In the prologue, not in the item loop: the kernel sits at its register budget.)
            GV *dst = static_cast<GV *>(p.grad_value) + idx * (p.M * D);
            const GV fill = wide ? GV(__builtin_nanf("")) : GV(0.f);
            for (int c = 0; c < p.M * D; ++c) dst[c] = fill;
        }
    }
    // Long candidate ranges (encoder shapes): the dynamic schedule deals the bands of ALL levels in the order of their
    // position in the image, frames innermost, so that the items an XCD runs at one time read the grad_out rows and the
    // points of the SAME queries (those near that part of the image) from its L2 -- with the bands of one level and frame
    // after another every item fetched them from memory again: FETCH_SIZE 4.5 GB for 1.6 GB of inputs on the 800x1333
    // encoder call (profiles/r03_logs/pmc_enc_hbm.txt).  Band b's key = start row / level height, its place = the number of
    // bands with a smaller (key, index): one thread per band, keys computed once (round 3: an insertion sort on one lane).
    const bool image_order = SORTED && NB <= kOwnMaxSorted;      // (the host counts the bands the same way; never trusted)
    if constexpr (SORTED) {
        if (image_order) {
            if (tid < NB) {
                int l = 0;
                while (l + 1 < L && s_first[l + 1] <= tid) ++l;
                s_key[tid] = s_R[l] > 0 ? (float)((tid - s_first[l]) * s_R[l]) / (float)s_H[l] : 0.f;
            }
            __syncthreads();
            if (tid < NB) {
                const float key = s_key[tid];
                int place = 0;
                for (int o = 0; o < NB; ++o) place += (s_key[o] < key || (s_key[o] == key && o < tid)) ? 1 : 0;
                s_order[place] = (unsigned short)tid;
            }
            __syncthreads();
        }
    }
    const int clips = p.groups / p.frames;
    const unsigned n_items = (unsigned)clips * (unsigned)p.frames * (unsigned)p.M * (unsigned)NB;      // (< 2^31: host)
    const bool dynamic = p.workspace != nullptr && (dbg & 16) == 0 && n_items < 16u * gridDim.x;
    // static stride only: does the walk of one workgroup -- (item / M) in steps of gridDim / M -- share a factor with the band count?
    // (then it meets the same few bands of every frame and prepare() rotates the bands by the (clip, frame) index)
    unsigned stride_gcd = 1u;
    {
        unsigned a = max(gridDim.x / (unsigned)max(p.M, 1), 1u), b = (unsigned)max(NB, 1);
        while (b) { const unsigned t = a % b; a = b; b = t; }
        stride_gcd = a;
    }
    const bool rotate_bands = stride_gcd > 1u && (dbg & 4096) == 0;
    const bool clip_major = (long long)p.Lq * (1 + p.window) <= 4096;       // see prepare()
    const int lane8 = blockIdx.x % 8;
    const int strideA = p.M * p.LA * p.PA, strideB = p.M * p.LB * p.PB;      // loc/attn elements per query
    // owner side: quad Q owns pixels s * kOwnQuads + Q of the band.  4-byte types: lane c of the quad holds the
    // channels [4c, 4c+4) of both 64-byte halves of the row (odd quads read the second half first: LDS banks, as
    // in the forward); 2-byte types: the 8 channels [8c, 8c+8) = one 16-byte slice of the 64-byte row.
    const int Q = tid / 4, cq = tid & 3, hsw = Q & 1;
    const int off1 = cq * 16 + hsw * 64;
    // (channels of acc[0..3] / acc[4..7]: off1 / 4 and (off1 ^ 64) / 4, formed where they are used)
    const unsigned ents_lds = lds_addr(ents);

    // Wave 0: the workgroup's n-th item -> descriptor + source tables in buffer `buf`.  Static stride or one ticket per item
    // (dynamic); 32-bit arithmetic throughout.
    auto prepare = [&](unsigned n, int buf) {
        unsigned item;
        if (dynamic) {
            unsigned tk = 0u;
            if (lane == 0) tk = atomicAdd(p.workspace + lane8, 1u);
            item = (unsigned)__builtin_amdgcn_readfirstlane((int)tk) * 8u + (unsigned)lane8;
        } else {
            item = blockIdx.x + n * gridDim.x;
        }
        const bool valid = item < n_items;
        int l = 0, part = 0, m = 0, f = 0, clip = 0;
        if (valid) {
            const unsigned M = (unsigned)p.M, F = (unsigned)p.frames;
            if (image_order && dynamic) {      // bands by position in the image, frames innermost (see s_order)
                m = (int)(item % M);
                unsigned rest = item / M;
                f = (int)(rest % F); rest /= F;
                part = s_order[rest % (unsigned)NB];
                clip = (int)(rest / (unsigned)NB);
                while (l + 1 < L && s_first[l + 1] <= part) ++l;
            } else if (dynamic) {
                // clip by clip, and inside a clip heaviest first: levels from the last to the first (see msda_bwd_value_lds_kernel).
                // The items in flight at one time then belong to one or two clips -- they share the clips' grad_out rows and
                // culling records in the L2, as the static stride's order does (round 4: with the levels outermost over ALL clips
                // the static stride was 5-9 % faster at 6-12 clips) -- and the item list still ends on light items.
                // (light items only -- a decoder's few hundred queries per source: an encoder-shaped call's one-band levels are
                // items of 20-40 chunks each, which must ALL start first; clip by clip BASELINE configs[1] went 0.63 -> 0.89 ms)
                const unsigned ctm = clip_major ? F * M : (unsigned)clips * F * M, per_clip = F * M * (unsigned)NB;
                clip = clip_major ? (int)(item / per_clip) : 0;
                l = L - 1;
                unsigned local = item - (unsigned)clip * per_clip;
                while (l > 0 && local >= ctm * (unsigned)(s_first[l + 1] - s_first[l])) {
                    local -= ctm * (unsigned)(s_first[l + 1] - s_first[l]);
                    --l;
                }
                const unsigned nb_l = (unsigned)(s_first[l + 1] - s_first[l]);
                m = (int)(local % M);
                unsigned rest = local / M;
                part = s_first[l] + (int)(rest % nb_l); rest /= nb_l;
                f = (int)(rest % F);
                if (!clip_major) clip = (int)(rest / F);
            } else {
                // static stride: the bands of one (clip, frame) adjacent.  A workgroup walks (item / M) in steps of gridDim / M (64):
                // with a band count that shares a factor with it (24 bands: every third band only) it would meet the same few bands
                // of every frame, all heavy or all light (round 6: the 800x1333 pyramid without its last level 0.51 -> 0.67 ms), so
                // THEN the bands are rotated by the (clip, frame) index (0.445 ms; MSDA_SCATTER_DBG = 4096: never).  Without a common
                // factor the plain order stays: rotating it cost the same pyramid WITH its last level (25 bands) 0.516 -> 0.553 ms.
                m = (int)(item % M);
                unsigned rest = item / M;
                const unsigned fc = rest / (unsigned)NB;
                part = (int)((rest + (rotate_bands ? fc : 0u)) % (unsigned)NB);
                f = (int)(fc % F);
                clip = (int)(fc / F);
                while (l + 1 < L && s_first[l + 1] <= part) ++l;
            }
        }
        if (lane == 0) {
            int *d = s_desc2[buf];
            d[0] = valid ? 1 : 0; d[1] = l; d[2] = part; d[3] = m; d[4] = f; d[5] = clip;
        }
        if (!valid) return;
        // sources that read frame f: the current-frame points of frame f, then every temporal slot (t, w) with
        // frame_table[t, w] == f; per source the first culling-table entry, first loc/attn element, first query row
        const int n_tw = p.frames * p.window;
        const bool hit = lane < n_tw && s_ftab[lane] == f;
        const u64 bal = __ballot(hit);
        if (lane == 0) {
            const int64_t g = (int64_t)clip * p.frames + f;
            s_src_tab2[buf][0] = ((g * p.M + m) * VL + l) * p.Lq;
            s_src_loc2[buf][0] = (g * p.Lq * p.M + m) * ((int64_t)p.LA * p.PA) + l * p.PA;
            s_src_q02[buf][0] = (int)(g * p.Lq);
            s_src_gmv2[buf][0] = (int)((g * p.M + m) * VL + l);
            s_nsrc2[buf] = 1 + (int)__popcll(bal);
        }
        if (hit) {
            const int k = 1 + (int)__popcll(bal & ((1ull << lane) - 1ull)), t = lane / p.window;
            const int vl = (lane - t * p.window) * L + l;
            const int64_t g = (int64_t)clip * p.frames + t;
            s_src_tab2[buf][k] = ((g * p.M + m) * VL + p.LA + vl) * p.Lq;
            s_src_loc2[buf][k] = (g * p.Lq * p.M + m) * ((int64_t)p.LB * p.PB) + vl * p.PB;
            s_src_q02[buf][k] = (int)(g * p.Lq);
            s_src_gmv2[buf][k] = (int)((g * p.M + m) * VL + p.LA + vl);
        }
    };
    if (wave == 0) prepare(0u, 0);
    __syncthreads();

    for (unsigned it = 0;; ++it) {
        const int cur = (int)(it & 1u);
        if (!s_desc2[cur][0]) break;
        MSDA_TR(1);                 // item start
        const int l = s_desc2[cur][1], part = s_desc2[cur][2], m = s_desc2[cur][3], f = s_desc2[cur][4], clip = s_desc2[cur][5];
        const long long *s_src_tab = s_src_tab2[cur], *s_src_loc = s_src_loc2[cur];
        const int *s_src_q0 = s_src_q02[cur], *s_src_gmv = s_src_gmv2[cur];
        const int s_nsrc = s_nsrc2[cur];
        if (((dbg >> 5) & 7) != 0 && l != ((dbg >> 5) & 7) - 1) {       // (measurement: MSDA_SCATTER_DBG = 32 * (level + 1): that level only)
            if (wave == 0) prepare(it + 1u, cur ^ 1);
            __syncthreads();
            continue;
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
        GV *gmap = static_cast<GV *>(p.grad_value) +
                   (((int64_t)clip * p.frames + f) * p.S + s_lsi[l]) * MD + m * D;        // pixel (0, 0) of the level, head m
        MSDA_TR(2);                 // item decoded
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
                    const TL *loc = static_cast<const TL *>(curf ? p.locA : p.locB);
                    const TL *aw = static_cast<const TL *>(curf ? p.awA : p.awB);
                    load_xy(loc + 2 * idx, x, y);
                    a = Store<TL>::get(aw + idx);
                    qrow = s_src_q0[k] + q;
                    act = true;
                }
            }
        };
        // ---- grad_out rows -> LDS (fp32), one per GROUP: wave w stages rows [RPWV w, RPWV (w + 1)) of the chunk.  4-byte types:
        // LDS-DMA, 8 lanes x 16 bytes per row.  2-byte types: 4 lanes x 16 bytes per row into registers (stage_rows), converted
        // and written as 2 x 16 bytes per lane once they have landed (stage_rows_finish, before the chunk's barrier).
        constexpr int RPWV = kOwnChunk / (kOwnThreads / kWave);   // rows per wave
        constexpr int LPR = kHalf ? 4 : 8, HPI = kWave / LPR;     // lanes per row, rows per instruction
        static_assert(kOwnChunk % (kOwnThreads / kWave) == 0 && RPWV % HPI == 0, "rows per wave");
        u32x4 raw_rows[kHalf ? RPWV / HPI : 1];
        auto stage_rows = [&](int base, int n) {
            if (direct || (dbg & 4)) return;
            const T *go = static_cast<const T *>(p.grad_out) + m * D + per_item(lane % LPR) * (16 / (int)sizeof(T));
            // (all list entries, then all first-query rows, then the loads: read one by one inside the per-instruction branches
            // below, the two dependent LDS round trips of every instruction ran back to back)
            unsigned es[RPWV / HPI];
            int qrs[RPWV / HPI];
#pragma unroll
            for (int i = 0; i < RPWV / HPI; ++i) es[i] = list[base + min(wave * RPWV + HPI * i + lane / LPR, n - 1)];
#pragma unroll
            for (int i = 0; i < RPWV / HPI; ++i) qrs[i] = s_src_q0[es[i] >> 26] + (int)(es[i] & 0x3fffffu);
#pragma unroll
            for (int i = 0; i < RPWV / HPI; ++i) {
                const int r0w = wave * RPWV + HPI * i;
                if (r0w < n) {                                      // uniform: this instruction has at least one live row
                    const int qr = qrs[i];
                    const T *gp = go + (int64_t)qr * MD;
                    if constexpr (kHalf) {
                        raw_rows[i] = *reinterpret_cast<const u32x4 *>(gp);
                    } else {
#if defined(__HIP_DEVICE_COMPILE__)
                        __builtin_amdgcn_global_load_lds(gp, (__attribute__((address_space(3))) void *)(rows + r0w * kRowB), 16, 0, 0);
#else
                        (void)gp;
#endif
                    }
                }
            }
        };
        auto stage_rows_finish = [&](int n) {
            if constexpr (kHalf) {
                if (direct || (dbg & 4)) return;
#pragma unroll
                for (int i = 0; i < RPWV / HPI; ++i) {
                    const int r0w = wave * RPWV + HPI * i;
                    if (r0w < n) {
                        float v[8];
                        unpack_raw(static_cast<const T *>(nullptr), raw_rows[i], v);
                        float4 *dst = reinterpret_cast<float4 *>(rows + (r0w + per_item(lane) / LPR) * kRowB + (lane % LPR) * 32);
                        dst[0] = make_float4(v[0], v[1], v[2], v[3]);
                        dst[1] = make_float4(v[4], v[5], v[6], v[7]);
                    }
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
                            if constexpr (std::is_same<GV, float>::value) {      // (host: levels wider than a band only with fp32 grad_value)
                                float *dst = gmap + (int64_t)(pix00 + dpix[c]) * MD;
                                for (int ch = 0; ch < D; ++ch) atomic_accumulate(dst + ch, wgt[c] * Store<T>::get(gr + ch));
                            }
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
            // this lane's two 16-byte pieces of a row, relative to the 128-byte group of the entry's own address
            const unsigned row_k1 = lds_addr(rows) - ents_lds + (unsigned)off1, row_k2 = lds_addr(rows) - ents_lds + (unsigned)(off1 ^ 64);
#pragma unroll
            for (int s = 0; s < kOwnSlots; ++s) {
                const int pix = s * kOwnQuads + Q;
                unsigned e = kOwnNil;
                if (pix < nvpix) { e = head[pix]; if (e != kOwnNil) head[pix] = kOwnNil; }
                while (e != kOwnNil) {
                    const uint2 en = *reinterpret_cast<const uint2 *>(lds_raw + (e - lds_addr(lds_raw)));
                    const float w = __uint_as_float(en.x);
                    // the row of entry e: a group has 16 x 8 = 128 = kRowB bytes of entries and the entries start on a
                    // 128-byte boundary, so (e & ~127) + (rows - ents) is the row's address
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    typedef const __attribute__((address_space(3))) f32x4 *lds_f32x4;
                    const unsigned rg = e & ~127u;
                    const f32x4 v1 = *(lds_f32x4)(size_t)(rg + row_k1);
                    const f32x4 v2 = *(lds_f32x4)(size_t)(rg + row_k2);
                    const float v[8] = {v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
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
            MSDA_TR(10);            // chunk start
            stage_rows(base, n);
            if (!primed) fetch_chunk(base, n);
            MSDA_TR(11);            // rows / points issued
#pragma unroll
            for (int j = 0; j < kPasses; ++j)
                if (j == 0 || n > j * (kOwnThreads / 4)) taps_link(j, hact[j], hx[j], hy[j], ha[j], hq[j]);
            MSDA_TR(12);            // taps + links done
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's rows have landed
            stage_rows_finish(n);
            MSDA_TR(13);            // rows landed
            __syncthreads();
            MSDA_TR(14);            // barrier 1
            if (nn > 0) fetch_chunk(nbase, nn);
            walk();
            MSDA_TR(15);            // walk done
            __syncthreads();
            MSDA_TR(16);            // barrier 2
        };

        // ---- cull the candidate groups in batches of one per thread against the band; chunks are cut from the END
        // of the survivor list, so nothing has to move
        int listed = 0;
        const int lo = min(r0 - 1, 32767), hi = min(r1, 32767);
        // (a level without records -- one band, Params::rec_mask -- has every point of every group as a candidate, like a call without a table)
        const bool has_rec = p.bbox != nullptr && ((p.rec_mask >> l) & 1u);
        auto load_records = [&](int gi0, int2 &iv, unsigned &ent, bool &live) {
            const int gi = gi0 + tid;
            live = gi < ng;
            iv = make_int2((int)0x80008000u, (int)0x80008000u);
            ent = 0u;
            if (live) {
                const int k = gi / p.Lq, q = gi - k * p.Lq;
                ent = ((unsigned)k << 26) | (unsigned)q;              // (q < 2^22: host)
                if (has_rec) iv = *reinterpret_cast<const int2 *>(p.bbox + (s_src_tab[k] + q) * 2);
            }
        };
        // Long candidate ranges (encoder shapes, Lq = S): a pre-pass over the 64-query block summaries marks the cull
        // batches that hold a block whose tap rows can reach the band; with local sampling all but a few are skipped.
        const int nbat = (ng + kOwnThreads - 1) / kOwnThreads;
        const bool skipping = has_rec && p.bsum != nullptr && nbat > 4 && nbat <= 32 * kLiveWords;
        if (skipping) {
            if (tid < kLiveWords) s_live[per_item(tid)] = 0u;
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
        // the next item's descriptor and tables, while the records fly -- with a static stride; a dynamic schedule draws its
        // ticket as late as it can (below): a workgroup that commits itself to a heavy item one item early is missing at the tail
        // (measured: 800x1333 encoder call 2.76 -> 3.58 ms with early tickets)
        if (wave == 0 && !dynamic) prepare(it + 1u, cur ^ 1);
        while (bcur < nbat) {
            const int bnext = next_live(bcur + 1);
            unsigned pm = 0u;
            if (live) {
                if (has_rec) {
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
            MSDA_TR(3);             // cull batch: records consumed, survivors listed
            __syncthreads();
            MSDA_TR(4);             // cull batch barrier
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
        MSDA_TR(5);                 // all chunks done
        if (wave == 0 && dynamic) prepare(it + 1u, cur ^ 1);
        // ---- owners store their pixels: grad_value is overwritten, every pixel of the band exactly once
        auto put4 = [&](GV *dst, float a, float b, float c, float d) {      // 4 consecutive channels, non-temporal
            if constexpr (std::is_same<GV, float>::value) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store((f32x4){a, b, c, d}, reinterpret_cast<f32x4 *>(dst));
            } else {
                const float v[4] = {a, b, c, d};
                SlabStore<GV>::store(dst, v);
            }
        };
        if (!direct && SF == 1) {
            GV *gband = gmap + (int64_t)r0 * W * MD;
            GV *gquad = gband + (int64_t)per_item(Q) * MD;          // pixel Q of the band; slot s is kOwnQuads pixels further
#pragma unroll
            for (int s = 0; s < kOwnSlots; ++s) {
                const int pix = s * kOwnQuads + Q;
                if constexpr (std::is_same<GV, float>::value) {
                    if (pix < npix) {
                        GV *o = gquad + (int64_t)s * kOwnQuads * MD;
                        put4(o + off1 / 4, acc[s][0], acc[s][1], acc[s][2], acc[s][3]);
                        put4(o + (off1 ^ 64) / 4, acc[s][4], acc[s][5], acc[s][6], acc[s][7]);
                    }
                } else {
                    // 16-bit grad_value: lane c of the quad gathers channels [8c, 8c+8) -- the first four from lane 2c mod 4, the
                    // next four from lane 2c+1 mod 4, out of their low (c < 2) or high channel block -- and writes ONE 16-byte
                    // piece (8-byte pieces from the fp32 register layout made partial-line writes: +0.05 ms on the bench workload)
                    float lo[4], hi[4], o8[8];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {       // lo / hi = this lane's channels [4cq, 4cq+4) / [16+4cq, 16+4cq+4)
                        lo[c] = hsw ? acc[s][4 + c] : acc[s][c];
                        hi[c] = hsw ? acc[s][c] : acc[s][4 + c];
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int la = __builtin_amdgcn_mov_dpp(__float_as_int(lo[c]), 0x88, 0xf, 0xf, true);     // quad_perm [0,2,0,2]
                        const int ha = __builtin_amdgcn_mov_dpp(__float_as_int(hi[c]), 0x88, 0xf, 0xf, true);
                        const int lb = __builtin_amdgcn_mov_dpp(__float_as_int(lo[c]), 0xDD, 0xf, 0xf, true);     // quad_perm [1,3,1,3]
                        const int hb = __builtin_amdgcn_mov_dpp(__float_as_int(hi[c]), 0xDD, 0xf, 0xf, true);
                        o8[c] = __int_as_float(cq < 2 ? la : ha);
                        o8[4 + c] = __int_as_float(cq < 2 ? lb : hb);
                    }
                    if (pix < npix) {
                        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                        u32x4_t w;
                        GV tmp[8];
#pragma unroll
                        for (int c = 0; c < 8; ++c) tmp[c] = SlabStore<GV>::cvt(o8[c]);
                        __builtin_memcpy(&w, tmp, 16);
                        __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t *>(gquad + (int64_t)s * kOwnQuads * MD + cq * 8));
                    }
                }
            }
        } else if (!direct) {
            // split lists: the partial sums of virtual pixel v = pix * SF + sub (slot 0 of quad v) go through the (now
            // free) row area as [v][32 channels] floats; one thread per (pixel, 4 channels) adds the SF partials
            float *part = reinterpret_cast<float *>(rows);
            if (Q < nvpix) {
                float *mine = part + per_item(Q) * D;
                *reinterpret_cast<float4 *>(mine + per_item(off1) / 4) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
                *reinterpret_cast<float4 *>(mine + (per_item(off1) ^ 64) / 4) = make_float4(acc[0][4], acc[0][5], acc[0][6], acc[0][7]);
            }
            __syncthreads();
            GV *gband = gmap + (int64_t)r0 * W * MD;
            for (int i = tid; i < npix * (D / 4); i += kOwnThreads) {
                const int pix = i / (D / 4), c4 = (i - pix * (D / 4)) * 4;
                float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int u = 0; u < SF; ++u) {
                    const float4 t4 = *reinterpret_cast<const float4 *>(part + ((pix << sfs) + u) * D + c4);
                    sum.x += t4.x; sum.y += t4.y; sum.z += t4.z; sum.w += t4.w;
                }
                put4(gband + (int64_t)pix * MD + c4, sum.x, sum.y, sum.z, sum.w);
            }
        }
        MSDA_TR(6);                 // stores issued
        __syncthreads();
        MSDA_TR(7);                 // item end
    }
#ifdef MSDA_SCATTER_TRACE
    if (tr_on) g_trace_n[blockIdx.x] = tr_n;
#endif
}

// The LDS scatter kernels OVERWRITE every pixel of a level whose row fits the band budget.  Pixels they
// do not own -- levels that take the float-atomic branch, or rows of `value` outside every level when
// spatial_shapes does not tile [0, S) -- are zero-filled here, so that callers need not memset grad_value.
__global__ void __launch_bounds__(256)
msda_zero_unowned_kernel(const Params p, int cap_slots, int gv_bytes)
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
        // (grad_value elements are 4 or 2 bytes: include/msda.h grad_value_dtype; MD * gv_bytes is a multiple of 4)
        unsigned *dst = reinterpret_cast<unsigned *>(static_cast<char *>(p.grad_value) + idx * MD * gv_bytes);
        for (int c = 0; c < MD * gv_bytes / 4; ++c) dst[c] = 0u;
    }
}

template <typename T, typename TL, int G>
int scatter_lds(const Params &p, unsigned grid, int cap_bytes, int dbg, hipStream_t stream)
{
    static LdsGrant granted;       // per instantiation and device
    const auto kern = &msda_bwd_value_lds_kernel<T, TL, G>;
    if (const int rc = grant_lds(reinterpret_cast<const void *>(kern), (size_t)cap_bytes, granted, "the LDS scatter kernel")) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kScatterThreads), (size_t)cap_bytes, stream, p, cap_bytes / 8, dbg);
    return check_launch("msda backward (LDS scatter kernel)");
}

template <typename T, typename TL>
int scatter_lds_g(int G, const Params &p, unsigned grid, int cap_bytes, int dbg, hipStream_t stream)
{
    switch (G) {
        case 1: return scatter_lds<T, TL, 1>(p, grid, cap_bytes, dbg, stream);
        case 2: return scatter_lds<T, TL, 2>(p, grid, cap_bytes, dbg, stream);
        case 4: return scatter_lds<T, TL, 4>(p, grid, cap_bytes, dbg, stream);
        case 8: return scatter_lds<T, TL, 8>(p, grid, cap_bytes, dbg, stream);
        case 16: return scatter_lds<T, TL, 16>(p, grid, cap_bytes, dbg, stream);
        case 32: return scatter_lds<T, TL, 32>(p, grid, cap_bytes, dbg, stream);
        case 64: return scatter_lds<T, TL, 64>(p, grid, cap_bytes, dbg, stream);
        default: return fail(MSDA_ERR_ARG, "msda: unsupported lanes per row%s");
    }
}

template <typename T, typename TL, typename GV, bool SORTED>
int scatter_grp(const Params &p, unsigned grid, int dbg, hipStream_t stream)
{
    static LdsGrant granted;
    const auto kern = &msda_bwd_value_grp_kernel<T, TL, GV, SORTED>;
    if (const int rc = grant_lds(reinterpret_cast<const void *>(kern), (size_t)grp_lds_bytes<T>(), granted,
                                 "the group-granular owner-computes scatter kernel")) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kOwnThreads), (size_t)grp_lds_bytes<T>(), stream, p, dbg);
    if (SORTED)
        return check_launch(std::is_same<GV, float>::value ? "msda backward (owner-computes scatter kernel, group-granular, items in image order)"
                                                           : "msda backward (owner-computes scatter kernel, group-granular, items in image order, grad_value in the storage type)");
    return check_launch(std::is_same<GV, float>::value ? "msda backward (owner-computes scatter kernel, group-granular)"
                                                       : "msda backward (owner-computes scatter kernel, group-granular, grad_value in the storage type)");
}

}  // namespace

int launch_zero_unowned(const Params &p, int cap_slots, int grad_value_elem_bytes, hipStream_t stream)
{
    const int64_t rows = (int64_t)p.groups * p.S;
    const unsigned zb = (unsigned)((rows + 255) / 256 < 16384 ? (rows + 255) / 256 : 16384);
    hipLaunchKernelGGL(msda_zero_unowned_kernel, dim3(zb), dim3(256), 0, stream, p, cap_slots, grad_value_elem_bytes);
    return check_launch("msda backward (zero-fill of pixels outside the bands)");
}

int launch_cull_summary(const Params &p, hipStream_t stream)
{
    const int64_t entries = (int64_t)p.groups * p.M * (p.LA + p.LB) * ((p.Lq + kCullBlock - 1) / kCullBlock);
    const unsigned sb = (unsigned)((entries + 3) / 4 < 65536 ? (entries + 3) / 4 : 65536);
    hipLaunchKernelGGL(msda_cull_summary_kernel, dim3(sb), dim3(256), 0, stream, p);
    return check_launch("msda backward (culling block summaries)");
}

int launch_scatter_lds(int dtype, int G, const Params &p, unsigned grid, int cap_bytes, int dbg, hipStream_t stream)
{
    return dispatch_types(dtype, [&](auto t, auto tl) {
        return scatter_lds_g<typename decltype(t)::type, typename decltype(tl)::type>(G, p, grid, cap_bytes, dbg, stream);
    });
}

int launch_scatter_grp(int dtype, bool storage_typed, const Params &p, unsigned grid, int dbg, hipStream_t stream)
{
    return dispatch_types(dtype, [&](auto t, auto tl) {
        using T = typename decltype(t)::type;
        using TL = typename decltype(tl)::type;
        // items in image order for long candidate ranges of a TEMPORAL call (the encoder's fused call: the frames of one band
        // run side by side and share the rows and points of the queries near it), when the host knows the band count
        // (MSDA_SCATTER_DBG = 256: the level-by-level order).  Plain calls keep the heaviest-first order: measured on one box,
        // image order / level order: 800x1333 T = 6 one clip 2.76 / 2.86 ms, 360x640 T = 6 0.54 / 0.55, but the single-frame
        // encoder call of BASELINE configs[1] (N = 8, bf16) 0.92 / 0.63 and the SwinL one (N = 6, fp16) 0.23 / 0.17 -- with
        // `clip` outermost the batch is 8 serial tails.
        // (round 4, after the per-item fixed costs shrank: at 360x640, Lq = 4820, image order is now the slower one, 0.555 / 0.529)
        // (a pinned route, msda_pin_route scatter_order: 1 = level order (bit 256), 2 = image order wherever the bands can be sorted (bit 2048))
        bool sorted = ((p.Lq >= 8192 && p.frames > 1) || (dbg & 2048)) && p.shapes_host != nullptr && (dbg & 256) == 0;
        int bands = 0;
        for (int l = 0; sorted && l < p.own_levels; ++l) {
            const long long H = p.shapes_host[2 * l], W = p.shapes_host[2 * l + 1];
            if (H <= 0 || W <= 0) { sorted = false; break; }         // (degenerate level: the device counts its bands differently)
            const long long R = W > 0 ? std::min<long long>(H, kOwnPix / W) : 0;
            bands += R > 0 ? (int)((H + R - 1) / R) : 1;
        }
        sorted = sorted && bands <= kOwnMaxSorted;
        if constexpr (sizeof(T) == 2) {
            if (storage_typed) return sorted ? scatter_grp<T, TL, T, true>(p, grid, dbg, stream) : scatter_grp<T, TL, T, false>(p, grid, dbg, stream);
        }
        return sorted ? scatter_grp<T, TL, float, true>(p, grid, dbg, stream) : scatter_grp<T, TL, float, false>(p, grid, dbg, stream);
    });
}

}  // namespace msda

#ifdef MSDA_SCATTER_TRACE
extern "C" int msda_debug_trace(unsigned long long *dst, int *counts)
{
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(msda::g_trace), sizeof(unsigned long long) * 8 * msda::kTraceLen) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(counts, HIP_SYMBOL(msda::g_trace_n), sizeof(int) * 8) != hipSuccess) return -2;
    return msda::kTraceLen;
}
#endif
