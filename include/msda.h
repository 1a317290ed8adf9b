/*
 * msda.h -- C ABI of the MI355X-native multi-scale deformable attention library (libmsda_hip.so).
 *
 * This is the drop-in boundary for DeVIS's only native component.  It replaces the two functions the
 * reference's pybind module `MultiScaleDeformableAttention` exports
 *     ms_deform_attn_forward / ms_deform_attn_backward
 *         (/root/reference/src/models/ops/src/vision.cpp:13-16, src/ms_deform_attn.h:20-60,
 *          src/cuda/ms_deform_attn_cuda.cu:20-153)
 * and adds one fused entry point pair for the per-frame loop of the temporal modules
 *         (src/models/ops/modules/ms_deform_attn.py:325-364, 366-404, 435-460).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer (HBM) unless stated;
 *   - tensors are dense row-major ("contiguous") in the layouts named below;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only enqueue work,
 *     they never allocate and never synchronise, and are re-entrant (forward on the Python thread,
 *     backward on an autograd worker -- as in the reference).  The only process-wide state is caches
 *     keyed by HIP device id (compute-unit count, per-kernel LDS opt-in) and the test knobs below;
 *   - test / measurement knobs are environment variables (MSDA_FWD_SLAB, MSDA_BWD_MODE, ...; listed in
 *     devis_amd/csrc/msda_api.hip at `struct Knobs`).  They are IGNORED unless MSDA_ENABLE_HOOKS=1, and
 *     are read once -- at the first call or when msda_reload_knobs() is called -- never on the launch path;
 *   - return value: MSDA_OK (0) or a negative msda_status; on failure msda_last_error() returns a
 *     thread-local message.  Unlike the reference (errors only printf'd,
 *     ms_deform_im2col_cuda.cuh:948-952,1321-1325) launch failures ARE reported;
 *   - `dtype` names the storage type of value / sampling_loc / attn_weight / out / grad_out /
 *     grad_sampling_loc / grad_attn_weight.  Arithmetic is fp32 for f32/bf16/f16 and fp64 for f64.
 *     `grad_value` is in the arithmetic type (float for f32/bf16/f16, double for f64) -- or, since ABI v10 and only
 *     where msda_grad_value_dtype() says so, directly in the 16-bit storage type (`grad_value_dtype` of the backward
 *     entry points: no fp32 buffer, no conversion pass over it afterwards) -- and is
 *     FULLY OVERWRITTEN (ABI v4): it need not be zeroed by the caller.  The reference zero-fills it
 *     (at::zeros_like, ms_deform_attn_cuda.cu:121) because every one of its kernels accumulates with
 *     atomics; here the LDS scatter owns and overwrites whole level-row bands, and the library zero-fills
 *     on the stream only what a fallback kernel accumulates into -- saving a full pass over grad_value.
 *
 *   - `value_strides` (HOST pointer to three int64, or NULL): element strides of `value` between
 *     {clips (= batch entries of msda_forward/backward), heads, pixels}.  NULL = the reference's dense
 *     [N, S, M, D], i.e. {frames*S*M*D, D, M*D}.  The head-major alternative [M, N, S, D] = {frames*S*D,
 *     N*S*D, D} -- what a per-head batched GEMM for value_proj produces -- is ~25 % faster to gather from:
 *     with dense rows 1 KiB apart and one head per XCD (for L2 locality) address bits 7..9 are constant
 *     on an XCD and only a fraction of its L2 channels is used.  grad_value is
 *     always dense [N, S, M, D].
 *
 *   - `spatial_shapes_host` (HOST pointer to [L, 2] int64, or NULL; ABI v8): a host copy of `spatial_shapes`,
 *     used ONLY to choose between kernels (which pyramid levels fit the LDS slab); NULL makes the library guess
 *     from `spatial_size`.  The device tensor stays the one the kernels read (no host synchronisation inside
 *     the library, as in the reference).  FORWARD results never depend on the hint.  For the BACKWARD entry
 *     points the hint, when given, MUST be a true copy of the device tensor (as for msda_grad_value_dtype):
 *     whether grad_value is zero-filled for a level wider than a scatter band (more than 1024 pixels per row;
 *     never in DeVIS) is decided from it.  A stale hint that hides such a level does not return silently wrong
 *     sums: the kernel sees the device shapes and fills that level's pixels of grad_value with NaN.
 *
 * Symbols:  N batch, S = sum_l H_l*W_l, M heads, D channels per head, Lq queries, L levels,
 *           P points;  spatial_shapes[l] = (H_l, W_l);  sampling_loc[..., 0] = x (width), 1 = y.
 */
#ifndef MSDA_H_
#define MSDA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSDA_ABI_VERSION 13
#define MSDA_BWD_WORKSPACE_BYTES 64   /* minimum device scratch of the backward entry points (ticket counters) */

enum msda_dtype {
    MSDA_F32 = 0, MSDA_F64 = 1, MSDA_BF16 = 2, MSDA_F16 = 3,
    /* ABI v11, operator entry points and msda_prep_*: value / out / grad_out in bf16 / f16, but sampling_loc, attn_weight and
     * their gradients in FLOAT.  A normalised coordinate stored in bf16 resolves 2^-9 = 0.16 px on an 80-pixel-wide level
     * (f16: 0.02 px), which caps the accuracy of 16-bit modules; the fused pre-op pass computes locations in fp32 anyway and
     * hands them over unrounded with these codes.  msda_prep_*: offsets / logits in the 16-bit type, reference points in float. */
    MSDA_BF16_LOC32 = 4, MSDA_F16_LOC32 = 5
};

enum msda_status {
    MSDA_OK = 0,
    MSDA_ERR_ARG = -1,     /* null pointer, non-positive size, unsupported combination */
    MSDA_ERR_DTYPE = -2,   /* unknown dtype code */
    MSDA_ERR_HIP = -3      /* HIP runtime / launch error (message carries hipGetErrorString) */
};

/* ABI version of the loaded library (== MSDA_ABI_VERSION it was built with). */
int msda_version(void);

/* "abi=13 arch=gfx950 timing_only=0" (ABI v13).  timing_only=1: the library was compiled with timing-only experiment macros
 * (kernels that skip part of their work to measure floors; wrong results by construction -- such a build needs
 * -DMSDA_TIMING_ONLY_BUILD to compile at all).  It prefixes every msda_last_route() with "TIMING-ONLY BUILD", its operator entry
 * points fail with MSDA_ERR_ARG unless MSDA_ENABLE_HOOKS=1, and the bindings refuse to load it without that variable. */
const char *msda_build_info(void);

/* Thread-local description of the last failure on this thread ("" if none). */
const char *msda_last_error(void);

/* Re-read the test / measurement knobs from the environment (see Conventions).  Not for production use. */
void msda_reload_knobs(void);

/* Thread-local, human-readable list of the kernels the last entry-point call of this thread launched
 * ("msda forward (resident-slab kernel); ...").  For tests, benchmarks and bug reports. */
const char *msda_last_route(void);

/*
 * Measured route table (ABI v12).  Which kernel family serves a call (tile / resident-slab / resident-window), with how many
 * tiles per wave, on which grid, and in which order the scatter deals its items is chosen from the call's sizes by rules
 * calibrated on a few pyramids (devis_amd/csrc/msda_api.hip, launch_fast).  A caller that has TIMED the
 * alternatives for a call shape pins the winner: from then on every call of that shape (any thread) takes it.  The reference
 * has nothing like it (one kernel per direction, ms_deform_attn_cuda.cu:61-75, 121-153); results never depend on a pin.
 *
 *   msda_route_key    writes the NUL-terminated key of a call shape into buf (returns its length, or a negative msda_status):
 *                     direction, msda_dtype code, clips, frames, window, S, M, D, L, Lq, points, and the HOST copy of the
 *                     shapes -- calls made without spatial_shapes_host are never looked up.  Plain calls: clips = N, frames = 1,
 *                     window = 0.
 *   msda_pin_route    settings = "name=value name=value ..." with names fwd_rs, fwd_rs_nt, fwd_win, fwd_tile_waves, bwd_rs,
 *                     bwd_rs_tpw, bwd_rs_fsplit, bwd_win (the MSDA_* knobs of the same names, see Conventions) and
 *                     scatter_order (1 = level order, 2 = image order).  An empty string removes the pin.  A knob SET in
 *                     the environment (MSDA_ENABLE_HOOKS=1) wins over a pin, whatever its value (also its default).
 *   msda_clear_routes removes every pin;  msda_route_count: pins held.
 * devis_amd.tune() measures and pins; devis_amd/routes.json is the table audited on MI355X, loaded with the library.
 */
int msda_route_key(int backward, int dtype, int clips, int frames, int window, int spatial_size, int num_heads, int channels,
                   int num_levels, int num_query, int num_curr_point, int num_temp_point, const int64_t *spatial_shapes_host,
                   char *buf, int buf_len);
int msda_pin_route(const char *key, const char *settings);
void msda_clear_routes(void);
int msda_route_count(void);

/*
 * Forward of one MSDeformAttnFunction call.
 * Replaces ms_deform_attn_forward (vision.cpp:14 -> ms_deform_attn_cuda.cu:20-80 ->
 * ms_deformable_im2col_gpu_kernel, ms_deform_im2col_cuda.cuh:237-299).
 *
 *   value             [N, S, M, D]         dtype
 *   spatial_shapes    [L, 2]  int64 (H,W)  device (dereferenced in-kernel, as cuh:274-277)
 *   level_start_index [L]     int64        device
 *   sampling_loc      [N, Lq, M, L, P, 2]  dtype
 *   attn_weight       [N, Lq, M, L, P]     dtype
 *   out               [N, Lq, M*D]         dtype, fully overwritten (need not be zeroed)
 *
 * The whole batch is one launch: the reference's im2col_step only chunks the batch into
 * pointer-offset launches (ms_deform_attn_cuda.cu:50-75) and is result-neutral; the host side
 * validates it (same divisibility error) and may pass a sub-batch here to reproduce the chunking.
 */
int msda_forward(int dtype, const void *value, const int64_t *spatial_shapes,
                 const int64_t *level_start_index, const void *sampling_loc, const void *attn_weight,
                 int batch, int spatial_size, int num_heads, int channels, int num_levels,
                 int num_query, int num_point, void *out, const int64_t *value_strides,
                 const int64_t *spatial_shapes_host, void *stream);

/*
 * Backward of one MSDeformAttnFunction call.
 * Replaces ms_deform_attn_backward (vision.cpp:15 -> ms_deform_attn_cuda.cu:83-153 -> every
 * ms_deformable_col2im_gpu_kernel_* variant, ms_deform_im2col_cuda.cuh:301-920,956-1326).
 *
 *   grad_out          [N, Lq, M*D]         dtype
 *   grad_value        [N, S, M, D]         grad_value_dtype: float (double for MSDA_F64), or `dtype` itself where
 *                                          msda_grad_value_dtype() returns it; fully overwritten (need not be zeroed)
 *   grad_sampling_loc [N, Lq, M, L, P, 2]  dtype, fully overwritten (skipped points get 0)
 *   grad_attn_weight  [N, Lq, M, L, P]     dtype, fully overwritten
 *   workspace         device scratch private to this call until it completes, `workspace_bytes` long.  Its
 *                     first MSDA_BWD_WORKSPACE_BYTES hold the work-ticket counters of the scatter pass (dynamic
 *                     scheduling); since ABI v8 the library zeroes them itself (in the gather pass that
 *                     precedes the scatter on the stream), the caller passes uninitialised memory.
 *                     With at least msda_backward_workspace_bytes()
 *                     bytes the gather pass also leaves, per (row, level), the interval of pixel rows its
 *                     taps touch, and the scatter pass culls the rows that cannot reach its band -- a large
 *                     win whenever sampling is local (encoder) or clustered (decoder).  NULL / 0 is allowed
 *                     (static scheduling, no culling).
 */
int msda_backward(int dtype, const void *value, const int64_t *spatial_shapes,
                  const int64_t *level_start_index, const void *sampling_loc,
                  const void *attn_weight, const void *grad_out,
                  int batch, int spatial_size, int num_heads, int channels, int num_levels,
                  int num_query, int num_point,
                  void *grad_value, int grad_value_dtype, void *grad_sampling_loc, void *grad_attn_weight,
                  void *workspace, long long workspace_bytes, const int64_t *value_strides,
                  const int64_t *spatial_shapes_host, void *stream);

/* The msda_dtype the `grad_value` buffer of a backward call of this shape may have besides the arithmetic type (ABI v10):
 * `dtype` itself for MSDA_BF16 / MSDA_F16 when the owner-computes scatter will write the gradient (D = 32, at most 4
 * points per level, ...: it overwrites every pixel once from fp32 registers, so it can round on the way out; every other
 * route accumulates into grad_value and needs float), else the arithmetic type.  Plain op: clips = batch, frames = 1,
 * window = 0.  `spatial_shapes_host` as passed to the backward call (NULL: always the arithmetic type; it must be a true
 * copy of the device tensor).  Passing the arithmetic type to the backward call is always valid. */
int msda_grad_value_dtype(int dtype, int clips, int frames, int window, int spatial_size, int num_heads, int channels,
                          int num_levels, int num_query, int num_curr_point, int num_temp_point,
                          const int64_t *spatial_shapes_host);

/* Bytes of `workspace` that enable every feature of msda_backward / msda_temporal_backward:
 * 64 (ticket counters) + rows * virtual_levels * 8 (per-point culling records) + their summaries over blocks of 64
 * queries (batch * num_heads * virtual_levels * ceil(num_query / 64) * 8), with rows = batch * num_query *
 * num_heads and virtual_levels = num_levels (msda_backward) or num_levels * (1 + window)
 * (msda_temporal_backward; batch = clips * frames). */
long long msda_backward_workspace_bytes(int batch, int num_query, int num_heads, int virtual_levels);

/*
 * Fused temporal forward: for every frame t of every clip, current-frame attention on value[t] PLUS
 * temporal attention on the `window` frames frame_table[t, :], summed -- one launch instead of the
 * reference's 2*T launches and T gather-copies `value[temporal_frames].flatten(0,1)` per layer
 * (ms_deform_attn.py:333,340,358 / 374,381,397 / 441,446,454).  Result equals
 *     out[c,t] = MSDA(value[c,t], shapes, lsi, loc_curr[c,t], aw_curr[c,t])
 *              + MSDA(cat_w value[c, frame_table[t,w]], shapes.repeat(window), ...,
 *                     loc_temp[c,t], aw_temp[c,t])
 *
 *   value        [clips*frames, S, M, D]                      dtype
 *   spatial_shapes / level_start_index   [L,2] / [L] int64    the CURRENT-frame pyramid
 *   frame_table  [frames, window] int32   device; absolute frame index inside the clip (repeats allowed,
 *                                         devis_transformer.py:103-113 mirrors frames at clip borders)
 *   loc_curr     [clips*frames, Lq, M, L, Pc, 2]              dtype
 *   aw_curr      [clips*frames, Lq, M, L, Pc]                 dtype
 *   loc_temp     [clips*frames, Lq, M, window*L, Pt, 2]       dtype (slot-major, level-minor:
 *                                                             ms_deform_attn.py:232-238 flatten(3,4))
 *   aw_temp      [clips*frames, Lq, M, window*L, Pt]          dtype
 *   out          [clips*frames, Lq, M*D]                      dtype, fully overwritten
 */
int msda_temporal_forward(int dtype, const void *value, const int64_t *spatial_shapes,
                          const int64_t *level_start_index, const int32_t *frame_table,
                          const void *loc_curr, const void *aw_curr,
                          const void *loc_temp, const void *aw_temp,
                          int clips, int frames, int window, int spatial_size, int num_heads,
                          int channels, int num_levels, int num_query,
                          int num_curr_point, int num_temp_point, void *out, const int64_t *value_strides,
                          const int64_t *spatial_shapes_host, void *stream);

/*
 * Fused temporal backward.  grad_value [clips*frames, S, M, D] (grad_value_dtype as for msda_backward; fully
 * overwritten, need not be zeroed): contributions of the current-frame and of every temporal slot land in it directly,
 * replacing the reference's index_put-add backward of value[temporal_frames].
 * The four grad_loc / grad_aw outputs have the shapes of their inputs and are fully overwritten.
 * `workspace`: as for msda_backward.
 */
int msda_temporal_backward(int dtype, const void *value, const int64_t *spatial_shapes,
                           const int64_t *level_start_index, const int32_t *frame_table,
                           const void *loc_curr, const void *aw_curr,
                           const void *loc_temp, const void *aw_temp, const void *grad_out,
                           int clips, int frames, int window, int spatial_size, int num_heads,
                           int channels, int num_levels, int num_query,
                           int num_curr_point, int num_temp_point,
                           void *grad_value, int grad_value_dtype, void *grad_loc_curr, void *grad_aw_curr,
                           void *grad_loc_temp, void *grad_aw_temp, void *workspace, long long workspace_bytes,
                           const int64_t *value_strides, const int64_t *spatial_shapes_host, void *stream);

/*
 * Pre-op fusion: everything the modules do between their Linears and the operator
 * (ms_deform_attn.py:105-121, 225-266, 327-352) in one pass -- the JOINT softmax over a (row, head)'s
 * L*Pc current-frame + window*L*Pt temporal logits, split into the two attention tensors, and the sampling
 * locations  ref + offsets / (W_l, H_l)  (ref_dim 2)  or  ref_xy + offsets / P * ref_wh * 0.5  (ref_dim 4,
 * boxes) -- written in the layouts msda_temporal_forward takes.  window = 0: the plain module (temporal
 * pointers unused).  rows = frames * queries; every tensor is dense:
 *
 *   offsets_curr [rows, M, L, Pc, 2]       offsets_temp [rows, M, window*L, Pt, 2]      (Linear outputs, viewed)
 *   logits_curr  [rows, M, L*Pc]           logits_temp  [rows, M, window*L*Pt]
 *   ref_curr     [rows, L, ref_dim]        ref_temp     [rows, window*L, ref_dim]
 *   spatial_shapes [L, 2] int64 (H, W), device
 *   loc_curr / aw_curr, loc_temp / aw_temp: shapes of the offsets / logits, dense, fully overwritten.
 *   raw_row_stride: 0, or the row stride (elements) of offsets_* / logits_* when the four are column slices of
 *   ONE matrix -- the output of the modules' four query-side Linears run as a single GEMM.
 */
int msda_prep_forward(int dtype, const void *offsets_curr, const void *offsets_temp, const void *logits_curr,
                      const void *logits_temp, const void *ref_curr, const void *ref_temp,
                      const int64_t *spatial_shapes, long long rows, int num_heads, int num_levels, int window,
                      int num_curr_point, int num_temp_point, int ref_dim, long long raw_row_stride,
                      void *loc_curr, void *loc_temp, void *aw_curr, void *aw_temp, void *stream);

/*
 * Its backward: grad_offsets = grad_loc scaled by the same per-level (or per-box) factors, grad_logits =
 * aw * (grad_aw - sum_j aw_j grad_aw_j) over the joint softmax.  Gradients of the reference points are sums
 * of grad_loc over heads and points and are left to the caller (they are only needed in the decoder).
 * raw_row_stride: as above, for grad_offsets_* / grad_logits_* (column slices of one gradient matrix).
 */
int msda_prep_backward(int dtype, const void *grad_loc_curr, const void *grad_loc_temp, const void *grad_aw_curr,
                       const void *grad_aw_temp, const void *aw_curr, const void *aw_temp, const void *ref_curr,
                       const void *ref_temp, const int64_t *spatial_shapes, long long rows, int num_heads,
                       int num_levels, int window, int num_curr_point, int num_temp_point, int ref_dim,
                       long long raw_row_stride, void *grad_offsets_curr, void *grad_offsets_temp,
                       void *grad_logits_curr, void *grad_logits_temp, void *stream);

/*
 * Padding mask.  Replaces `value.masked_fill(input_padding_mask[..., None], 0)`
 * (ms_deform_attn.py:102-103) and the matching mask on grad_value: zeroes, IN PLACE, row i of `rows` for every i with
 * padding_mask[i] != 0 and touches nothing else -- the cost is the `pixels` mask bytes plus the masked rows, not a
 * read + write of the whole tensor.
 *   rows          [pixels, row_elems] of `dtype`, row i at element offset i * row_stride (row_stride >= row_elems:
 *                 the dense value tensor has row_stride = M*D, the padded one (M + pad)*D)
 *   padding_mask  [pixels] bytes (a torch.bool tensor), device
 */
int msda_mask_rows(int dtype, void *rows, const void *padding_mask, long long pixels, long long row_elems,
                   long long row_stride, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MSDA_H_ */
