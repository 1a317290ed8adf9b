// msda_generic.hip -- any-shape kernels (correctness path), the modules' fused pre-op pass, the padding-mask pass.
#include "msda_common.h"

namespace msda {
namespace {

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

template <typename T, typename TL, typename A>        // T: value / out, TL: sampling_loc / attn_weight, A: arithmetic type
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
            const TL *loc = static_cast<const TL *>(arr ? p.locB : p.locA);
            const TL *aw = static_cast<const TL *>(arr ? p.awB : p.awA);
            const int P = arr ? p.PB : p.PA, nl = arr ? p.LB : p.LA, LP = nl * P;
            for (int pt = 0; pt < LP; ++pt) {
                const Level lv = make_level(p, t, (arr ? p.LA : 0) + pt / P);
                const int64_t idx = row * LP + pt;
                const GTaps<A> tp = make_gtaps<A>((A)Store<TL>::get(loc + 2 * idx),
                                                  (A)Store<TL>::get(loc + 2 * idx + 1), lv, p.v_pix);
                if (!tp.valid) continue;
                A val = 0;
                for (int k = 0; k < 4; ++k)
                    if (tp.valid & (1 << k)) val += tp.w[k] * (A)Store<T>::get(value + tp.off[k]);
                acc += val * (A)Store<TL>::get(aw + idx);
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
template <typename T, typename TL, typename A>
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
            const TL *loc = static_cast<const TL *>(arr ? p.locB : p.locA);
            const TL *aw = static_cast<const TL *>(arr ? p.awB : p.awA);
            TL *gloc = static_cast<TL *>(arr ? p.glocB : p.glocA);
            TL *gaw = static_cast<TL *>(arr ? p.gawB : p.gawA);
            const int P = arr ? p.PB : p.PA, nl = arr ? p.LB : p.LA, LP = nl * P;
            for (int pt = 0; pt < LP; ++pt) {
                const Level lv = make_level(p, t, (arr ? p.LA : 0) + pt / P);
                const int64_t idx = row * LP + pt;
                const A a = (A)Store<TL>::get(aw + idx);
                // offsets in PIXELS: value and grad_value (always the standard layout) have different strides
                const GTaps<A> tp = make_gtaps<A>((A)Store<TL>::get(loc + 2 * idx),
                                                  (A)Store<TL>::get(loc + 2 * idx + 1), lv, 1);
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
                    Store<TL>::put(gaw + idx, g_aw);
                    Store<TL>::put(gloc + 2 * idx, (A)lv.W * g_w * a);
                    Store<TL>::put(gloc + 2 * idx + 1, (A)lv.H * g_h * a);
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

// T: type of the Linear-side tensors (raw offsets / logits, their gradients); TL: type of the operator-side tensors
// (sampling_loc, attn_weight, their gradients) AND of the reference points -- T, or float with a 16-bit T (MSDA_*_LOC32)
template <typename T, typename TL, typename A, bool BWD>
__global__ void __launch_bounds__(256)
msda_prep_kernel(const PrepParams p)
{
    typedef typename std::conditional<BWD, TL, T>::type TI;      // what the pass reads per point: weights / grad_loc, or logits / offsets
    typedef typename std::conditional<BWD, T, TL>::type TO;      // ... and writes: grad of logits / offsets, or weights / locations
    const int lane = threadIdx.x % 32;
    const int nc = p.L * p.Pc, nt = p.W * p.L * p.Pt, n = nc + nt;
    const int64_t pairs = p.rows * p.M;
    for (int64_t pair = (int64_t)blockIdx.x * 8 + threadIdx.x / 32; pair < pairs; pair += (int64_t)gridDim.x * 8) {
        const int64_t row = pair / p.M;
        const int m = (int)(pair - row * p.M);
        // first element of this (row, head) in a Linear-side tensor with n_ (x2 for offsets) elements per head
        auto raw = [&](int n_) { return p.ld ? row * p.ld + (int64_t)m * n_ : pair * n_; };
        const TI *lc = static_cast<const TI *>(BWD ? p.aw_c : p.logit_c) + (BWD ? pair * nc : raw(nc));
        const TI *lt = static_cast<const TI *>(BWD ? p.aw_t : p.logit_t) + (BWD ? pair * nt : raw(nt));
        constexpr int NE = 8;                 // register-resident fast path: n <= 32 * NE logits per (row, head)
        if (!BWD) {
            // ---- joint softmax over the n logits of this (row, head)   (ref :252-258 / F.softmax)
            TL *ac = static_cast<TL *>(p.aw_c) + pair * nc, *at = static_cast<TL *>(p.aw_t) + pair * nt;
            if (n <= 32 * NE) {               // each logit is read once and exponentiated once
                A v[NE];
                A mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const int e = lane + 32 * i;
                    v[i] = e < n ? (A)Store<TI>::get(e < nc ? lc + e : lt + (e - nc)) : (A)-INFINITY;
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
                    if (e < n) Store<TL>::put(e < nc ? ac + e : at + (e - nc), v[i] / sum);
                }
            } else {
                A mx = -INFINITY;
                for (int e = lane; e < n; e += 32) {
                    const A v = (A)Store<TI>::get(e < nc ? lc + e : lt + (e - nc));
                    mx = v > mx ? v : mx;
                }
                mx = half_wave_max<A>(mx);
                A sum = 0;
                for (int e = lane; e < n; e += 32) sum += prep_exp((A)Store<TI>::get(e < nc ? lc + e : lt + (e - nc)) - mx);
                sum = half_wave_sum<A>(sum);
                for (int e = lane; e < n; e += 32) {
                    const A v = prep_exp((A)Store<TI>::get(e < nc ? lc + e : lt + (e - nc)) - mx) / sum;
                    Store<TL>::put(e < nc ? ac + e : at + (e - nc), v);
                }
            }
        } else {
            // ---- softmax backward: g_logit = p * (g - sum_j p_j g_j)
            const TL *gc = static_cast<const TL *>(p.gaw_c) + pair * nc, *gt = static_cast<const TL *>(p.gaw_t) + pair * nt;
            T *oc = static_cast<T *>(p.glogit_c) + raw(nc), *ot = static_cast<T *>(p.glogit_t) + raw(nt);
            if (n <= 32 * NE) {
                A pe[NE], ge[NE];
                A dot = 0;
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const int e = lane + 32 * i;
                    pe[i] = e < n ? (A)Store<TI>::get(e < nc ? lc + e : lt + (e - nc)) : (A)0;
                    ge[i] = e < n ? (A)Store<TL>::get(e < nc ? gc + e : gt + (e - nc)) : (A)0;
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
                    dot += (A)Store<TI>::get(e < nc ? lc + e : lt + (e - nc)) * (A)Store<TL>::get(e < nc ? gc + e : gt + (e - nc));
                dot = half_wave_sum<A>(dot);
                for (int e = lane; e < n; e += 32) {
                    const A pe = (A)Store<TI>::get(e < nc ? lc + e : lt + (e - nc));
                    const A ge = (A)Store<TL>::get(e < nc ? gc + e : gt + (e - nc));
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
            const TL *ref = static_cast<const TL *>(cur ? p.ref_c : p.ref_t) + (row * (cur ? p.L : p.W * p.L) + vl) * p.d;
            const TI *in = static_cast<const TI *>(BWD ? (cur ? p.gloc_c : p.gloc_t) : (cur ? p.off_c : p.off_t)) + (BWD ? idx : ridx);
            TO *out = static_cast<TO *>(BWD ? (cur ? p.goff_c : p.goff_t) : (cur ? p.loc_c : p.loc_t)) + (BWD ? ridx : idx);
            const A x = (A)Store<TI>::get(in), y = (A)Store<TI>::get(in + 1);
            if (p.d == 2) {
                const A nx = (A)p.shapes[2 * l + 1], ny = (A)p.shapes[2 * l];      // (W_l, H_l)
                if (!BWD) {
                    Store<TO>::put(out, (A)Store<TL>::get(ref) + x / nx);
                    Store<TO>::put(out + 1, (A)Store<TL>::get(ref + 1) + y / ny);
                } else {
                    Store<TO>::put(out, x / nx);
                    Store<TO>::put(out + 1, y / ny);
                }
            } else {
                const A bw = (A)Store<TL>::get(ref + 2), bh = (A)Store<TL>::get(ref + 3);
                if (!BWD) {
                    Store<TO>::put(out, (A)Store<TL>::get(ref) + x / (A)P * bw * (A)0.5);
                    Store<TO>::put(out + 1, (A)Store<TL>::get(ref + 1) + y / (A)P * bh * (A)0.5);
                } else {
                    Store<TO>::put(out, x * (A)0.5 * bw / (A)P);
                    Store<TO>::put(out + 1, y * (A)0.5 * bh / (A)P);
                }
            }
        }
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

template <typename T, typename TL, typename A>
int generic(const Params &p, bool bwd, hipStream_t stream)
{
    const int64_t rows = (int64_t)p.groups * p.Lq * p.M;
    if (bwd) {
        if (hipMemsetAsync(p.grad_value, 0, (size_t)p.groups * p.S * p.M * p.D * sizeof(A), stream) != hipSuccess)
            return fail(MSDA_ERR_HIP, "msda backward: hipMemsetAsync(grad_value) failed%s");
        const unsigned blocks = (unsigned)(rows < 65536 * 16 ? rows : 65536 * 16);
        hipLaunchKernelGGL((msda_bwd_generic_kernel<T, TL, A>), dim3(blocks), dim3(kWave), 0, stream, p, rows);
        return check_launch("msda backward (generic kernel)");
    }
    const int64_t total = rows * p.D;
    const int64_t want = (total + 255) / 256;
    const unsigned blocks = (unsigned)(want < 65536 * 8 ? want : 65536 * 8);
    hipLaunchKernelGGL((msda_fwd_generic_kernel<T, TL, A>), dim3(blocks), dim3(256), 0, stream, p, total);
    return check_launch("msda forward (generic kernel)");
}

template <typename T, typename TL, typename A>
int prep(const PrepParams &p, bool bwd, hipStream_t stream)
{
    const int64_t pairs = p.rows * p.M;
    const int64_t want = (pairs + 7) / 8;
    const unsigned blocks = (unsigned)(want < 65536 * 4 ? want : 65536 * 4);
    if (bwd) hipLaunchKernelGGL((msda_prep_kernel<T, TL, A, true>), dim3(blocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((msda_prep_kernel<T, TL, A, false>), dim3(blocks), dim3(256), 0, stream, p);
    return check_launch(bwd ? "msda prep backward" : "msda prep forward");
}

}  // namespace

int launch_generic(int dtype, const Params &p, bool bwd, hipStream_t stream)
{
    if (dtype == MSDA_F64) return generic<double, double, double>(p, bwd, stream);
    return dispatch_types(dtype, [&](auto t, auto tl) {
        return generic<typename decltype(t)::type, typename decltype(tl)::type, float>(p, bwd, stream);
    });
}

int launch_prep(int dtype, const PrepParams &p, bool bwd, hipStream_t stream)
{
    if (dtype == MSDA_F64) return prep<double, double, double>(p, bwd, stream);
    return dispatch_types(dtype, [&](auto t, auto tl) {
        return prep<typename decltype(t)::type, typename decltype(tl)::type, float>(p, bwd, stream);
    });
}

int launch_mask_rows(int bytes_per_thread, char *rows, const uint8_t *mask, long long pixels, int chunks,
                     long long stride_bytes, unsigned blocks, hipStream_t stream)
{
    const dim3 grid(blocks), block(256);
    switch (bytes_per_thread) {
        case 16: hipLaunchKernelGGL(msda_mask_rows_kernel<16>, grid, block, 0, stream, rows, mask, pixels, chunks, stride_bytes); break;
        case 8: hipLaunchKernelGGL(msda_mask_rows_kernel<8>, grid, block, 0, stream, rows, mask, pixels, chunks, stride_bytes); break;
        case 4: hipLaunchKernelGGL(msda_mask_rows_kernel<4>, grid, block, 0, stream, rows, mask, pixels, chunks, stride_bytes); break;
        default: hipLaunchKernelGGL(msda_mask_rows_kernel<2>, grid, block, 0, stream, rows, mask, pixels, chunks, stride_bytes); break;
    }
    return check_launch("msda mask rows");
}

}  // namespace msda
