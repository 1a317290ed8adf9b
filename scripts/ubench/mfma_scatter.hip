// mfma_scatter.hip -- round-6 gate ("next" #2): grad_value of the COARSE pyramid levels as a split-precision bf16
// matrix product on the MFMA pipe, timed on the headline call's shapes and checked against a CPU double sum BEFORE anything is
// built into the library.
//
//   grad_value[P pixels x 32 channels] (one (clip, source frame, head, level)) = A[P x K] . G[K x 32]
//   K = the (source, query) groups that sample the frame (6 sources x 300 queries at T = 6), G[k] = the group's grad_out row,
//   A[pix, k] = sum over the group's <= 4 points of  attention x bilinear weight of the point at pixel pix  (<= 16 non-zeros).
//
// One 256-thread workgroup per item, its four waves independent until the end: a wave takes every fourth STEP of 16 groups;
// thread (g, pt) = point pt of group g.  A step: (1) geometry per point; (2) the quad's four points are merged IN REGISTERS: each
// thread evaluates all four points' bilinear "tents" at its own four corner pixels (quad_perm broadcasts, canonical point order,
// so that two threads whose corners coincide hold bit-identical totals and may both write); (3) totals split into bf16 hi + lo and
// written as 2-byte cells into the wave's private A tiles in LDS ([pixel][16 k], 32-byte rows, XOR-swizzled for ds_read_b128);
// (4) G rows straight from memory into the B-operand layout (8 coalesced dword loads per lane), split hi + lo in registers;
// (5) per 32-pixel tile: a_hi.g_hi + a_lo.g_hi + a_hi.g_lo on v_mfma_f32_32x32x16_bf16 into fp32 accumulators; (6) the cells are
// written back to zero.  No barriers, no lists, no atomics inside the loop; the four waves' accumulators are added through LDS
// at the end and stored with 128-byte rows.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/ubench/mfma_scatter.hip -o scripts/ubench/mfma_scatter
//   scripts/ubench/mfma_scatter [clips]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct Args {
    const float *loc_c, *aw_c, *loc_t, *aw_t, *go;
    float *gv;
    const int *ftab;                 // [T][W]
    int clips, T, M, Lq, L, P, W, S;
    int l0, nl;                      // the trailing levels [l0, l0 + nl) this launch handles (nl <= 2), fused in one pixel space
    int H[2], Wd[2], poff[2], npix, lsi;
};

template <int CTRL> __device__ __forceinline__ float quad(float v)       // quad_perm broadcast of one lane of each quad
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float tent(float d)      // max(0, 1 - |d|): one v_sub with |src| and clamp
{
    return __builtin_amdgcn_fmed3f(1.f - __builtin_fabsf(d), 0.f, 1.f);
}
__device__ __forceinline__ unsigned lds_off(const void *p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p; }

// LDS image of one A tile (hi or lo): [2 k-chunks][NP pixel rows][8 k] bf16 -- cell (pix, k) at ((k >> 3) * NP + pix) * 16 +
// (k & 7) * 2.  The MFMA A operand of 32-pixel tile t (lane = pixel 32 t + (lane & 31), k-chunk lane >> 5) is one ds_read_b128 at
// ((lane >> 5) * NP + 32 t + (lane & 31)) * 16: the 16 lanes of a ds_read_b128 group read 16 consecutive 16-byte slots (no
// swizzle needed), and a point's four cells are cell00 + {0, 16, W * 16, W * 16 + 16}.
template <int MT, int NL, int NPROD>
__global__ void __launch_bounds__(256, MT >= 8 ? 2 : 4)
mfma_scatter_kernel(const Args p)
{
    constexpr int D = 32;
    constexpr int NP = MT == 10 ? 304 : MT * 32 + 16;           // pixel rows per k-chunk: npix + 1 (trash row) <= NP
    constexpr int kTile = 2 * NP * 16;                          // bytes of one A tile (hi or lo)
    constexpr int kPhase = MT >= 8 ? 4 : MT;                    // tiles per reduction phase: 4 waves x kPhase x 4 KiB
    extern __shared__ __attribute__((aligned(256))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *Ahi = lds + wave * (2 * kTile), *Alo = Ahi + kTile;
    for (int i = tid * 16; i < 4 * 2 * kTile + 256; i += 256 * 16) *reinterpret_cast<u32x4 *>(lds + i) = u32x4{0u, 0u, 0u, 0u};

    const int m = blockIdx.x % p.M, f = (blockIdx.x / p.M) % p.T, clip = blockIdx.x / (p.M * p.T);
    const int MD = p.M * D, npix = p.npix;
    // sources reading frame f: lane s of every wave holds source s (-1 = the frame's own current-frame points, else t * W + w
    // with ftab[t][w] == f); a step reads its source with a readlane
    int my_src = -1, nsrc = 1;
    {
        const bool hit = lane < p.T * p.W && p.ftab[lane] == f;
        const unsigned long long bal = __ballot(hit);
        nsrc = 1 + __popcll(bal);
        // lane s >= 1 wants the position of the s-th set bit of bal
        int pos = -1, cnt = 0;
        for (int b = 0; b < 64; ++b)
            if ((bal >> b) & 1ull) { ++cnt; if (cnt == lane) pos = b; }
        if (lane >= 1) my_src = pos;
    }
    const int nst = (p.Lq + 15) / 16, nsteps = nsrc * nst;
    __syncthreads();                                      // the tiles are zero before any wave writes a cell

    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int g = lane >> 2, pt = lane & 3;               // builder role: point pt of group g of the step
    const int n = lane & 31, kh = lane >> 5;              // MFMA role: channel n / pixel n of a tile, k-chunk kh
    const unsigned wr_hi = lds_off(Ahi) + (unsigned)((g >> 3) * NP) * 16u + (unsigned)(g & 7) * 2u;      // + pix * 16
    const unsigned trash = (unsigned)npix * 16u;          // row npix of the tile: written, never stored
    // lane parts of the point index: current-frame points [.., M, L, P], temporal points [.., M, W * L, P]
    const int lane_c = (g * p.M * p.L) * p.P + pt, lane_t = (g * p.M * p.W * p.L) * p.P + pt;
    const int lane_g = (8 * kh * MD + n);                 // lane part of a G element: row 8 kh (+ j), channel n

    struct Step { const float *loc, *aw, *go; int gmin, temporal; };
    // scalar description of step st (clamped): the source's point and row bases at the step's first query
    auto describe = [&](int st_) -> Step {
        const int st = min(st_, nsteps - 1);
        const int s = st / nst, j = st - s * nst;
        const int q0 = min(16 * j, p.Lq - 16);                                      // the last step of a source overlaps the one before
        const int src = __builtin_amdgcn_readlane(my_src, s);
        const int t_src = src < 0 ? f : src / p.W, w_src = src < 0 ? 0 : src - (src / p.W) * p.W;
        const long long row = ((long long)clip * p.T + t_src) * p.Lq + q0;          // first query row of the step
        Step d;
        const long long lv = src < 0 ? (long long)p.L : (long long)p.W * p.L;
        const long long sbase = ((row * p.M + m) * lv + (src < 0 ? p.l0 : w_src * p.L + p.l0)) * p.P;
        d.loc = (src < 0 ? p.loc_c : p.loc_t) + 2 * sbase;
        d.aw = (src < 0 ? p.aw_c : p.aw_t) + sbase;
        d.go = p.go + row * MD + m * D;
        d.gmin = 16 * j - q0;
        d.temporal = src < 0 ? 0 : 1;
        return d;
    };
    // the loads of a step, issued one step AHEAD, all unconditional (the compiler must be able to count them: s_waitcnt vmcnt(N))
    auto issue = [&](const Step &d, float (&x)[NL], float (&y)[NL], float (&a)[NL], float (&gr)[8]) {
        const int lo = d.temporal ? lane_t : lane_c;
#pragma unroll
        for (int li = 0; li < NL; ++li) {
            const float2 xy = *reinterpret_cast<const float2 *>(d.loc + 2 * (lo + li * p.P));
            x[li] = xy.x; y[li] = xy.y; a[li] = d.aw[lo + li * p.P];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) gr[j] = (d.go + (long long)j * MD)[lane_g];
    };
    float nx[NL], ny[NL], na[NL], ngr[8];
    Step cur = describe(wave);
    issue(cur, nx, ny, na, ngr);
    for (int st = wave; st < nsteps; st += 4) {
        float xs[NL], ys[NL], as[NL], gr[8];
        const bool act = g >= cur.gmin;
#pragma unroll
        for (int li = 0; li < NL; ++li) { xs[li] = nx[li]; ys[li] = ny[li]; as[li] = na[li]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) gr[j] = ngr[j];
        cur = describe(st + 4);
        issue(cur, nx, ny, na, ngr);
        unsigned cells[NL][4];
#pragma unroll
        for (int li = 0; li < NL; ++li) {
            const int H = p.H[li], Wd = p.Wd[li];
            const float Hf = (float)H, Wf = (float)Wd;
            float a = as[li];
            // ---- geometry (reference arithmetic: rounded product, then the subtraction; cuh:288 range test)
            float h_im = ys[li] * Hf - 0.5f, w_im = xs[li] * Wf - 0.5f;
            const bool inr = act && h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
            if (!inr) { h_im = -100.f; w_im = -100.f; a = 0.f; }
            const float r0f = floorf(h_im), c0f = floorf(w_im), r1f = r0f + 1.f, c1f = c0f + 1.f;
            const int hl = (int)r0f, wl = (int)c0f;
            // ---- merge: totals of all four points of the quad at my four corner pixels, in point order
            float t00 = 0.f, t01 = 0.f, t10 = 0.f, t11 = 0.f;
#define MERGE(CTRL) { \
                const float th0 = tent(quad<CTRL>(h_im) - r0f) * quad<CTRL>(a), th1 = tent(quad<CTRL>(h_im) - r1f) * quad<CTRL>(a); \
                const float tw0 = tent(quad<CTRL>(w_im) - c0f), tw1 = tent(quad<CTRL>(w_im) - c1f); \
                t00 = fmaf(th0, tw0, t00); t01 = fmaf(th0, tw1, t01); t10 = fmaf(th1, tw0, t10); t11 = fmaf(th1, tw1, t11); }
            MERGE(0x00) MERGE(0x55) MERGE(0xaa) MERGE(0xff)
#undef MERGE
            // ---- cells; masked-out corners go to the trash row
            const bool rv0 = inr && (unsigned)hl < (unsigned)H, rv1 = inr && (unsigned)(hl + 1) < (unsigned)H;
            const bool cv0 = (unsigned)wl < (unsigned)Wd, cv1 = (unsigned)(wl + 1) < (unsigned)Wd;
            const unsigned c00 = (unsigned)(p.poff[li] + hl * Wd + wl) * 16u, wrow = (unsigned)Wd * 16u;
            cells[li][0] = wr_hi + ((rv0 && cv0) ? c00 : trash); cells[li][1] = wr_hi + ((rv0 && cv1) ? c00 + 16u : trash);
            cells[li][2] = wr_hi + ((rv1 && cv0) ? c00 + wrow : trash); cells[li][3] = wr_hi + ((rv1 && cv1) ? c00 + wrow + 16u : trash);
            // hi = bf16(t) (round to nearest even), lo = bf16(t - hi); two cells per conversion
            const float tt[4] = {t00, t01, t10, t11};
#pragma unroll
            for (int c = 0; c < 4; c += 2) {
                const bf16x2 hi = {(__bf16)tt[c], (__bf16)tt[c + 1]};
                const unsigned hb = __builtin_bit_cast(unsigned, hi);
                const bf16x2 lo = {(__bf16)(tt[c] - __uint_as_float(hb << 16)), (__bf16)(tt[c + 1] - __uint_as_float(hb & 0xffff0000u))};
                const unsigned lb = __builtin_bit_cast(unsigned, lo);
                asm volatile("ds_write_b16 %0, %1" : : "v"(cells[li][c]), "v"(hb) : "memory");
                asm volatile("ds_write_b16_d16_hi %0, %1" : : "v"(cells[li][c + 1]), "v"(hb) : "memory");
                asm volatile("ds_write_b16 %0, %1 offset:%2" : : "v"(cells[li][c]), "v"(lb), "n"(kTile) : "memory");
                asm volatile("ds_write_b16_d16_hi %0, %1 offset:%2" : : "v"(cells[li][c + 1]), "v"(lb), "n"(kTile) : "memory");
            }
        }
        // ---- B operand: G[k = 8 kh + j][n], hi + lo
        bf16x8 bhi, blo;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const bf16x2 hi = {(__bf16)gr[j], (__bf16)gr[j + 1]};
            const unsigned hb = __builtin_bit_cast(unsigned, hi);
            bhi[j] = hi[0]; bhi[j + 1] = hi[1];
            blo[j] = (__bf16)(gr[j] - __uint_as_float(hb << 16));
            blo[j + 1] = (__bf16)(gr[j + 1] - __uint_as_float(hb & 0xffff0000u));
        }
        // ---- the products, tile by tile (LDS operations of one wave complete in order: the reads see the writes above)
        bf16x8 ahi[MT], alo[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            ahi[t] = *reinterpret_cast<const bf16x8 *>(Ahi + (kh * NP + n) * 16 + t * 512);
            alo[t] = *reinterpret_cast<const bf16x8 *>(Alo + (kh * NP + n) * 16 + t * 512);
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[t], bhi, acc[t], 0, 0, 0);
        if (NPROD >= 2) {
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[t], bhi, acc[t], 0, 0, 0);
        }
        if (NPROD >= 3) {
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[t], blo, acc[t], 0, 0, 0);
        }
        // ---- cells back to zero
        const unsigned zero = 0u;
#pragma unroll
        for (int li = 0; li < NL; ++li)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                asm volatile("ds_write_b16 %0, %1" : : "v"(cells[li][c]), "v"(zero) : "memory");
                asm volatile("ds_write_b16 %0, %1 offset:%2" : : "v"(cells[li][c]), "v"(zero), "n"(kTile) : "memory");
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");

    // ---- the four waves' accumulators -> one, through LDS, kPhase tiles at a time; wave w adds up tile (phase * kPhase + w)
    float *red = reinterpret_cast<float *>(lds);
    float *gmap = p.gv + (((long long)clip * p.T + f) * p.S + p.lsi) * MD + m * D;
#pragma unroll
    for (int ph = 0; ph * kPhase < MT; ++ph) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < kPhase; ++t)
            if (ph * kPhase + t < MT)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((wave * kPhase + t) * 16 + r) * 64 + lane] = acc[ph * kPhase + t][r];
        __syncthreads();
        for (int tw = wave; tw < kPhase && ph * kPhase + tw < MT; tw += 4) {
            const int t = ph * kPhase + tw;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) v += red[((w * kPhase + tw) * 16 + r) * 64 + lane];
                const int pix = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (pix < npix) gmap[(long long)pix * MD + n] = v;
            }
        }
    }
}

template <int MT> static size_t lds_bytes()
{
    const size_t tile = 2 * (MT == 10 ? 304 : MT * 32 + 16) * 16, phase = MT >= 8 ? 4 : MT, red = 4 * phase * 4096;
    return std::max<size_t>(4 * 2 * tile + 256, red);
}

static double frand(unsigned long long &s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; }

int main(int argc, char **argv)
{
    const int clips = argc > 1 ? atoi(argv[1]) : 16;
    const int T = 6, M = 8, D = 32, Lq = 300, L = 4, P = 4, W = T - 1;
    const int Hs[4] = {45, 23, 12, 6}, Ws[4] = {80, 40, 20, 10};
    int lsi[4], S = 0;
    for (int l = 0; l < L; ++l) { lsi[l] = S; S += Hs[l] * Ws[l]; }
    const int G = clips * T;
    const size_t n_lc = (size_t)G * Lq * M * L * P, n_lt = (size_t)G * Lq * M * W * L * P, n_go = (size_t)G * Lq * M * D;
    const size_t n_gv = (size_t)G * S * M * D;
    std::vector<float> loc_c(2 * n_lc), aw_c(n_lc), loc_t(2 * n_lt), aw_t(n_lt), go(n_go);
    unsigned long long seed = 12345;
    const char *mode = getenv("LOCS");      // "clustered": the four points of a group within ~1.5 pixels of each other (duplicates galore)
    const bool clustered = mode && !strcmp(mode, "clustered");
    auto fill_loc = [&](std::vector<float> &loc, std::vector<float> &aw, size_t npts) {
        for (size_t i = 0; i < npts; i += P) {
            const double cx = frand(seed), cy = frand(seed);
            for (int k = 0; k < P; ++k) {
                double x = frand(seed) * 1.2 - 0.1, y = frand(seed) * 1.2 - 0.1;          // a few points outside the map
                if (clustered) { x = cx + (frand(seed) - 0.5) * 0.15; y = cy + (frand(seed) - 0.5) * 0.25; }
                loc[2 * (i + k)] = (float)x; loc[2 * (i + k) + 1] = (float)y;
                aw[i + k] = (float)(frand(seed) * 0.05);
            }
        }
    };
    fill_loc(loc_c, aw_c, n_lc);
    fill_loc(loc_t, aw_t, n_lt);
    for (size_t i = 0; i < n_go; ++i) go[i] = (float)((frand(seed) - 0.5) * 4.0 * std::exp((frand(seed) - 0.5) * 6.0));
    std::vector<int> ftab(T * W);
    for (int t = 0; t < T; ++t) { int w = 0; for (int f = 0; f < T; ++f) if (f != t) ftab[t * W + w++] = f; }

    Args a;
    float *d_lc, *d_ac, *d_lt, *d_at, *d_go, *d_gv; int *d_ft;
    CHECK(hipMalloc(&d_lc, loc_c.size() * 4)); CHECK(hipMalloc(&d_ac, aw_c.size() * 4));
    CHECK(hipMalloc(&d_lt, loc_t.size() * 4)); CHECK(hipMalloc(&d_at, aw_t.size() * 4));
    CHECK(hipMalloc(&d_go, go.size() * 4)); CHECK(hipMalloc(&d_gv, n_gv * 4)); CHECK(hipMalloc(&d_ft, ftab.size() * 4));
    CHECK(hipMemcpy(d_lc, loc_c.data(), loc_c.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_ac, aw_c.data(), aw_c.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_lt, loc_t.data(), loc_t.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_at, aw_t.data(), aw_t.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_go, go.data(), go.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_ft, ftab.data(), ftab.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_gv, 0, n_gv * 4));
    a.loc_c = d_lc; a.aw_c = d_ac; a.loc_t = d_lt; a.aw_t = d_at; a.go = d_go; a.gv = d_gv; a.ftab = d_ft;
    a.clips = clips; a.T = T; a.M = M; a.Lq = Lq; a.L = L; a.P = P; a.W = W; a.S = S;

    const int items = clips * T * M;
    auto launch = [&](int l0, int nl, int nprod) {
        a.l0 = l0; a.nl = nl; a.lsi = lsi[l0]; a.npix = 0;
        for (int li = 0; li < nl; ++li) { a.H[li] = Hs[l0 + li]; a.Wd[li] = Ws[l0 + li]; a.poff[li] = a.npix; a.npix += Hs[l0 + li] * Ws[l0 + li]; }
#define GO(MT, NL, NP) { static bool once = false; if (!once) { once = true; CHECK(hipFuncSetAttribute((const void *)mfma_scatter_kernel<MT, NL, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes<MT>())); } \
            mfma_scatter_kernel<MT, NL, NP><<<items, 256, lds_bytes<MT>(), 0>>>(a); }
        if (nl == 2) { if (a.npix > 320) { printf("too large\n"); exit(1); } if (nprod == 3) GO(10, 2, 3) else GO(10, 2, 1) }
        else if (a.npix <= 64) { if (nprod == 3) GO(2, 1, 3) else GO(2, 1, 1) }
        else if (a.npix <= 256) { if (nprod == 3) GO(8, 1, 3) else GO(8, 1, 1) }
        else { printf("level %d too large\n", l0); exit(1); }
#undef GO
        CHECK(hipGetLastError());
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto time_ms = [&](auto fn) {
        for (int i = 0; i < 3; ++i) fn();
        CHECK(hipDeviceSynchronize());
        std::vector<float> ts;
        for (int i = 0; i < 20; ++i) {
            CHECK(hipEventRecord(e0)); fn(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        return ts[ts.size() / 2];
    };
    printf("mfma_scatter: %d clips, T=%d, %d queries/frame, M=%d, %s locations; items per level %d\n", clips, T, Lq, M, clustered ? "clustered" : "uniform", items);
    for (int nprod = 3; nprod >= 1; nprod -= 2) {
        const float t2 = time_ms([&] { launch(2, 1, nprod); }), t3 = time_ms([&] { launch(3, 1, nprod); });
        const float both = time_ms([&] { launch(2, 1, nprod); launch(3, 1, nprod); });
        const float fused = time_ms([&] { launch(2, 2, nprod); });
        printf("  %d products: level 2 (12x20) %.4f ms   level 3 (6x10) %.4f ms   both, back to back %.4f ms   FUSED (one launch, 10 tiles) %.4f ms\n", nprod, t2, t3, both, fused);
    }
    // ---- check (3 products) against a CPU double sum on a sample of items
    CHECK(hipMemset(d_gv, 0, n_gv * 4));
    if (getenv("UNFUSED")) { launch(2, 1, 3); launch(3, 1, 3); } else launch(2, 2, 3);
    CHECK(hipDeviceSynchronize());
    std::vector<float> gv(n_gv);
    CHECK(hipMemcpy(gv.data(), d_gv, n_gv * 4, hipMemcpyDeviceToHost));
    double worst = 0.0, scale = 0.0, sumsq = 0.0, refsq = 0.0;
    for (int lvl = 2; lvl <= 3; ++lvl) {
        const int H = Hs[lvl], Wd = Ws[lvl];
        for (int it = 0; it < items; it += std::max(1, items / 24)) {
            const int m = it % M, f = (it / M) % T, clip = it / (M * T);
            std::vector<double> ref((size_t)H * Wd * D, 0.0);
            auto add_points = [&](const float *loc, const float *aw, long long row, long long lv_total, int lv) {
                for (int pt = 0; pt < P; ++pt) {
                    const long long idx = ((row * M + m) * lv_total + lv) * P + pt;
                    const float x = loc[2 * idx], y = loc[2 * idx + 1], at = aw[idx];
                    const float h_im = y * (float)H - 0.5f, w_im = x * (float)Wd - 0.5f;
                    if (!(h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)Wd)) continue;
                    const int hl = (int)std::floor(h_im), wl = (int)std::floor(w_im);
                    const double lh = (double)h_im - hl, lw = (double)w_im - wl;
                    const double wts[4] = {(1 - lh) * (1 - lw), (1 - lh) * lw, lh * (1 - lw), lh * lw};
                    for (int c = 0; c < 4; ++c) {
                        const int r = hl + (c >> 1), cc = wl + (c & 1);
                        if (r < 0 || r >= H || cc < 0 || cc >= Wd) continue;
                        for (int d = 0; d < D; ++d) ref[((size_t)r * Wd + cc) * D + d] += wts[c] * at * (double)go[row * M * D + m * D + d];
                    }
                }
            };
            for (int q = 0; q < Lq; ++q) add_points(loc_c.data(), aw_c.data(), ((long long)clip * T + f) * Lq + q, L, lvl);
            for (int t = 0; t < T; ++t)
                for (int w = 0; w < W; ++w)
                    if (ftab[t * W + w] == f)
                        for (int q = 0; q < Lq; ++q) add_points(loc_t.data(), aw_t.data(), ((long long)clip * T + t) * Lq + q, (long long)W * L, w * L + lvl);
            for (int pix = 0; pix < H * Wd; ++pix)
                for (int d = 0; d < D; ++d) {
                    const double got = gv[(((size_t)clip * T + f) * S + lsi[lvl] + pix) * M * D + m * D + d], want = ref[(size_t)pix * D + d];
                    worst = std::max(worst, std::fabs(got - want)); scale = std::max(scale, std::fabs(want));
                    sumsq += (got - want) * (got - want); refsq += want * want;
                }
        }
    }
    printf("  check (3 products): max |err| %.3e, max |ref| %.3e -> %.2e of scale (fp32 tests allow 2e-5); rms err / rms ref %.2e\n",
           worst, scale, worst / scale, std::sqrt(sumsq / refsq));
    return worst <= 2e-5 * scale ? 0 : 1;
}
