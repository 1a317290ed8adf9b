// msda_api.hip -- host side of libmsda_hip.so: the extern "C" ABI of include/msda.h, argument checks, kernel selection
// (which family takes a shape: resident-slab / tile / generic kernels; owner-computes / LDS / atomic scatter), the test
// knobs and the per-device caches.  The kernels live in the other translation units of this directory.
#include "msda_common.h"
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

namespace msda {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *detail)
{
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

thread_local char g_route[512] = "";     // kernels launched by the last entry-point call of this thread (msda_last_route)

int check_launch(const char *what)
{
    size_t used = strlen(g_route);
    if (MSDA_IS_TIMING_ONLY && used == 0) {       // a library with timing-only kernels compiled in says so in every route
        snprintf(g_route, sizeof(g_route), "TIMING-ONLY BUILD (results are wrong by construction)");
        used = strlen(g_route);
    }
    if (used + 3 < sizeof(g_route)) snprintf(g_route + used, sizeof(g_route) - used, "%s%s", used ? "; " : "", what);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return MSDA_ERR_HIP;
    }
    return MSDA_OK;
}

namespace {

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
    int bwd_rs = -1, bwd_rs_tpw = 0;    // resident-slab gather pass: -1 auto, 0 off, 1 force; tiles per wave (0 = auto)
    int fwd_tile_waves = -1;            // forward tile kernel: waves per tile (-1 auto)
    int fwd_rs_body = -1;               // resident-slab forward: slot body compiled for a slab from level 1 / 2 (-1: the rule in launch_fast)
    int bwd_rs_fsplit = -1;             // gather pass with one source frame per workgroup: parts per (clip, head, frame); -1 auto, 0 off
    int fwd_win = -1, bwd_win = -1;     // resident-window kernels (encoder-shaped calls): -1 auto, 0 off, 1 force
    int win_min_halo = 5;               // narrowest halo a window plan may have; one staging phase is preferred from here on (5 holds
                                        // the reference's initial offsets, <= 4 pixels of every level: ms_deform_attn.py:64-76)
    int bwd_atomic = 0;                 // MSDA_BWD_MODE=atomic: one-kernel backward with global atomics
    int bwd_phases = 3;                 // 1 = gather pass only, 2 = scatter pass only, 3 = both
    int bwd_cull = 1;                   // 0: no culling structure, 2: (min, max) intervals instead of per-point records
    int bwd_all_records = 0;            // measurement: the gather pass leaves records for every level (a later scatter-only call may walk them)
    int bwd_summary = 1;                // 64-query block summaries for long candidate ranges
    int scatter_lds_kb = 144, scatter_dbg = 0;
    int scatter_own = -1;               // owner-computes scatter: -1 auto, 0 off (the LDS-atomic scatter instead)
    int scatter_mfma = -1;              // matrix-pipe scatter of the coarse levels (msda_mfma.hip): -1 auto, 0 off, 1 wherever it applies
    int scatter_part = 0;               // measurement: 1 = only the owner-computes kernel of a scatter that runs both, 2 = only the matrix-pipe kernel
    int scatter_own_levels = -1;        // measurement: the owner-computes scatter handles only the first n levels (grad_value of the others is NOT computed)
    int force_generic = 0;
    int gv_storage = 1;                 // 0: msda_grad_value_dtype always answers the arithmetic type (A/B measurements)
    int dbg = 0;
    int hooks = 0;                      // MSDA_ENABLE_HOOKS=1 was set when the knobs were read
    unsigned forced = 0;                // route knobs that were SET in the environment (kForce* bits), whatever their value: a knob
                                        // forced to its default (MSDA_FWD_RS=-1 for a rules-only A/B run) still wins over a pin
};
enum : unsigned { kForceFwdRs = 1, kForceFwdRsNt = 2, kForceFwdWin = 4, kForceFwdTileWaves = 8, kForceBwdRs = 16, kForceBwdRsTpw = 32,
                  kForceBwdRsFsplit = 64, kForceBwdWin = 128, kForceScatterDbg = 256, kForceScatterMfma = 512 };
Knobs g_knobs;
int g_knobs_loaded = 0;

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return (e && e[0]) ? atoi(e) : dflt;
}

bool env_set(const char *name)
{
    const char *e = getenv(name);
    return e && e[0];
}

void load_knobs()
{
    Knobs k;
    if (env_int("MSDA_ENABLE_HOOKS", 0) == 1) {
        k.hooks = 1;
        k.forced = (env_set("MSDA_FWD_RS") ? kForceFwdRs : 0u) | (env_set("MSDA_FWD_RS_NT") ? kForceFwdRsNt : 0u) |
                   (env_set("MSDA_FWD_WIN") ? kForceFwdWin : 0u) | (env_set("MSDA_FWD_TILE_WAVES") ? kForceFwdTileWaves : 0u) |
                   (env_set("MSDA_BWD_RS") ? kForceBwdRs : 0u) | (env_set("MSDA_BWD_RS_TPW") ? kForceBwdRsTpw : 0u) |
                   (env_set("MSDA_BWD_RS_FSPLIT") ? kForceBwdRsFsplit : 0u) | (env_set("MSDA_BWD_WIN") ? kForceBwdWin : 0u) |
                   (env_set("MSDA_SCATTER_DBG") ? kForceScatterDbg : 0u) | (env_set("MSDA_SCATTER_MFMA") ? kForceScatterMfma : 0u);
        k.fwd_rs = env_int("MSDA_FWD_RS", k.fwd_rs); k.fwd_rs_nt = env_int("MSDA_FWD_RS_NT", k.fwd_rs_nt);
        k.bwd_rs = env_int("MSDA_BWD_RS", k.bwd_rs); k.bwd_rs_tpw = env_int("MSDA_BWD_RS_TPW", k.bwd_rs_tpw);
        k.bwd_rs_fsplit = env_int("MSDA_BWD_RS_FSPLIT", k.bwd_rs_fsplit);
        k.fwd_tile_waves = env_int("MSDA_FWD_TILE_WAVES", k.fwd_tile_waves);
        k.fwd_rs_body = env_int("MSDA_FWD_RS_BODY", k.fwd_rs_body);
        k.fwd_win = env_int("MSDA_FWD_WIN", k.fwd_win); k.bwd_win = env_int("MSDA_BWD_WIN", k.bwd_win);
        k.win_min_halo = env_int("MSDA_WIN_MIN_HALO", k.win_min_halo);
        const char *mode = getenv("MSDA_BWD_MODE");
        k.bwd_atomic = (mode && !strcmp(mode, "atomic")) ? 1 : 0;
        k.bwd_phases = env_int("MSDA_BWD_PHASES", k.bwd_phases);
        k.bwd_cull = env_int("MSDA_BWD_CULL", k.bwd_cull);
        k.bwd_all_records = env_int("MSDA_BWD_ALL_RECORDS", k.bwd_all_records);
        k.bwd_summary = env_int("MSDA_BWD_SUMMARY", k.bwd_summary);
        k.scatter_lds_kb = env_int("MSDA_SCATTER_LDS_KB", k.scatter_lds_kb);
        k.scatter_dbg = env_int("MSDA_SCATTER_DBG", k.scatter_dbg);
        k.scatter_own = env_int("MSDA_SCATTER_OWN", k.scatter_own);
        k.scatter_own_levels = env_int("MSDA_SCATTER_OWN_LEVELS", k.scatter_own_levels);
        k.scatter_mfma = env_int("MSDA_SCATTER_MFMA", k.scatter_mfma);
        k.scatter_part = env_int("MSDA_SCATTER_PART", k.scatter_part);
        k.force_generic = env_int("MSDA_FORCE_GENERIC", 0) == 1;
        k.gv_storage = env_int("MSDA_GV_STORAGE", k.gv_storage);
        k.dbg = env_int("MSDA_DBG", 0);
    }
    g_knobs = k;
    __atomic_store_n(&g_knobs_loaded, 1, __ATOMIC_RELEASE);
}

// While a call runs with a pinned route (msda_pin_route), this thread's knobs() answers the pinned settings laid over the
// environment's: see RouteScope below.
thread_local const Knobs *tl_route_knobs = nullptr;

inline const Knobs &knobs()
{
    if (tl_route_knobs) return *tl_route_knobs;
    if (!__atomic_load_n(&g_knobs_loaded, __ATOMIC_ACQUIRE)) load_knobs();      // benign race: every thread reads the same environment
    return g_knobs;
}

// ---- measured route table (ABI v12) ---------------------------------------------------------------------------------
// The rules in launch_fast choose a kernel family, tiles per wave, the gather pass's grid and the scatter's item order from
// sizes alone; they were calibrated on three pyramids and a few batch sizes and are the FALLBACK.  A
// caller that has TIMED the alternatives for a call shape (devis_amd.tune, or the audited table shipped as
// devis_amd/routes.json) pins the winner here: key = everything the rules look at (direction, dtype code, clips, frames,
// window, S, M, D, L, Lq, points, the host copy of the shapes), settings = the route knobs.  A knob forced through the
// environment (tests, A/B runs) wins over a pin.  Results never depend on a pin: every route computes the same function.
struct RoutePin {
    std::string key;
    int fwd_rs = -2, fwd_rs_nt = -2, fwd_win = -2, fwd_tile_waves = -2;        // -2 = not pinned
    int bwd_rs = -2, bwd_rs_tpw = -2, bwd_rs_fsplit = -2, bwd_win = -2, scatter_order = -2, scatter_mfma = -2;
};
std::mutex g_routes_mutex;
std::vector<RoutePin> g_routes;
int g_routes_n = 0;                     // (read without the lock on the launch path: 0 = nothing pinned, skip the key)

int route_key(char *buf, int len, bool bwd, int dtype, const Params &p)
{
    if (!p.shapes_host || p.L > 16) return -1;
    int n = snprintf(buf, len, "%c|%d|%d|%d|%d|%d|%d|%d|%d|%d|%d|%d|", bwd ? 'b' : 'f', dtype, p.groups / (p.frames > 0 ? p.frames : 1),
                     p.frames, p.window, p.S, p.M, p.D, p.L, p.Lq, p.PA, p.PB);
    for (int l = 0; l < p.L && n > 0 && n < len; ++l)
        n += snprintf(buf + n, len - n, "%s%lldx%lld", l ? "," : "", (long long)p.shapes_host[2 * l], (long long)p.shapes_host[2 * l + 1]);
    return (n > 0 && n < len) ? n : -1;
}

bool parse_route_settings(const char *text, RoutePin &pin)
{
    std::string t(text ? text : "");
    size_t i = 0;
    while (i < t.size()) {
        while (i < t.size() && (t[i] == ' ' || t[i] == ',')) ++i;
        if (i >= t.size()) break;
        const size_t eq = t.find('=', i);
        if (eq == std::string::npos) return false;
        size_t end = t.find_first_of(" ,", eq);
        if (end == std::string::npos) end = t.size();
        const std::string name = t.substr(i, eq - i);
        const int v = atoi(t.substr(eq + 1, end - eq - 1).c_str());
        if (name == "fwd_rs") pin.fwd_rs = v; else if (name == "fwd_rs_nt") pin.fwd_rs_nt = v;
        else if (name == "fwd_win") pin.fwd_win = v; else if (name == "fwd_tile_waves") pin.fwd_tile_waves = v;
        else if (name == "bwd_rs") pin.bwd_rs = v; else if (name == "bwd_rs_tpw") pin.bwd_rs_tpw = v;
        else if (name == "bwd_rs_fsplit") pin.bwd_rs_fsplit = v; else if (name == "bwd_win") pin.bwd_win = v;
        else if (name == "scatter_order") pin.scatter_order = v; else if (name == "scatter_mfma") pin.scatter_mfma = v;
        else return false;
        i = end;
    }
    return true;
}

// For the duration of one entry-point call: the pinned settings of this call's shape (if any) laid over the knobs.
struct RouteScope {
    Knobs merged;
    bool active = false;
    RouteScope(bool bwd, int dtype, const Params &p)
    {
        if (__atomic_load_n(&g_routes_n, __ATOMIC_ACQUIRE) == 0 || tl_route_knobs) return;
        char key[512];
        if (route_key(key, (int)sizeof key, bwd, dtype, p) < 0) return;
        RoutePin pin;
        {
            std::lock_guard<std::mutex> lock(g_routes_mutex);
            bool found = false;
            for (const RoutePin &r : g_routes)
                if (r.key == key) { pin = r; found = true; break; }
            if (!found) return;
        }
        merged = knobs();
        const unsigned forced = merged.forced;
        auto lay = [forced](int &dst, unsigned bit, int pinned) { if (pinned != -2 && !(forced & bit)) dst = pinned; };
        lay(merged.fwd_rs, kForceFwdRs, pin.fwd_rs); lay(merged.fwd_rs_nt, kForceFwdRsNt, pin.fwd_rs_nt);
        lay(merged.fwd_win, kForceFwdWin, pin.fwd_win); lay(merged.fwd_tile_waves, kForceFwdTileWaves, pin.fwd_tile_waves);
        lay(merged.bwd_rs, kForceBwdRs, pin.bwd_rs); lay(merged.bwd_rs_tpw, kForceBwdRsTpw, pin.bwd_rs_tpw);
        lay(merged.bwd_rs_fsplit, kForceBwdRsFsplit, pin.bwd_rs_fsplit); lay(merged.bwd_win, kForceBwdWin, pin.bwd_win);
        lay(merged.scatter_mfma, kForceScatterMfma, pin.scatter_mfma);
        if (pin.scatter_order == 1 && !(forced & kForceScatterDbg)) merged.scatter_dbg |= 256;
        if (pin.scatter_order == 2 && !(forced & kForceScatterDbg)) merged.scatter_dbg |= 2048;
        tl_route_knobs = &merged;
        active = true;
    }
    ~RouteScope() { if (active) tl_route_knobs = nullptr; }
    RouteScope(const RouteScope &) = delete;
    RouteScope &operator=(const RouteScope &) = delete;
};

int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    return dev;
}

}  // namespace

// ---- per-device caches --------------------------------------------------------------------------------------
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

int grant_lds(const void *kernel, size_t bytes, LdsGrant &granted, const char *what)
{
    const int dev = current_device();
    if (bytes <= granted.bytes[dev]) return MSDA_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        return fail(MSDA_ERR_HIP, "msda: cannot reserve the LDS budget of %s", what);
    granted.bytes[dev] = bytes;
    return MSDA_OK;
}

namespace {

int elem_bytes(int dtype) { return dtype == MSDA_F32 ? 4 : dtype == MSDA_F64 ? 8 : 2; }        // of value / out / grad_out
// the storage type of value / out / grad_out behind a dtype code (MSDA_*_LOC32: the 16-bit type)
int storage_dtype(int dtype) { return dtype == MSDA_BF16_LOC32 ? MSDA_BF16 : dtype == MSDA_F16_LOC32 ? MSDA_F16 : dtype; }

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
// pair gathers the non-resident levels from the same maps, and both kernels are bound by how fast a CU's vector-memory
// path returns those scattered lines, i.e. by what an XCD's 4 MiB L2 keeps of the maps: take the
// LARGEST workgroups (least slab staging) whose pairs in flight per XCD keep the non-resident levels within
// `l2_budget`, else the smallest.  Measured on the round-3 kernels, 16 clips of the DeVIS decoder shape (same box),
// 1 / 2 / 4 tiles per wave: fp32 (460 KiB per level-0 map) forward 0.337 / 0.364 / -- ms, gather pass 0.423 / 0.471 /
// 0.560; bf16 (230 KiB, two heads per 128-byte line) forward -- / 0.246 / 0.238, gather pass 0.363 / 0.326 / 0.312.
// Hence a budget of 2 MiB for 4-byte types and 4 MiB for 2-byte types.
int rs_tiles_per_wave(const Params &p, int tiles_per_clip, long long outside_bytes, bool force, long long l2_budget, int max_tiles = 4)
{
    // Among the candidates that (nearly) fill the chip and keep the non-resident levels of the (clip, head) pairs in flight per XCD
    // within `l2_budget` -- or the smallest one if none does -- the one with the least (rounds of workgroups over the CUs) x (tiles
    // per wave), larger workgroups on a tie: 360 workgroups of 4 tiles are two rounds of 4, 720 of 2 are three rounds of 2 (round
    // 4 audit, 1-clip encoder call on the SwinL pyramid in bf16: forward 0.455 -> 0.373 ms, gather pass 0.634 -> 0.534).
    const int64_t clips = p.groups / p.frames;
    const int cus = device_cus(), cus_per_xcd = cus / 8 > 0 ? cus / 8 : 1;
    int pick = 0, fallback = 0;
    long long best = 0;
    for (int cand : {4, 2, 1}) {
        if (cand > max_tiles) continue;
        const int parts = (tiles_per_clip + kRsWaves * cand - 1) / (kRsWaves * cand);
        const long long wgs = clips * p.M * parts;
        // must fill the chip -- nearly, for workgroups of several tiles per wave: 232 workgroups of 4 tiles (1-clip encoder call,
        // bf16 forward 0.263 -> 0.227 ms) do; 240 of 1 tile (1-image SwinL encoder call) are 7 % behind the tile kernels
        if (!force && (cand > 1 ? 4 * wgs < 3LL * cus : wgs < cus)) continue;
        fallback = cand;                                              // (ends as the smallest admissible candidate)
        const long long pairs = (cus_per_xcd + parts - 1) / parts;
        // (encoder-shaped calls -- one query per pixel, sampling round its own position -- find their non-resident lines in the L2
        // whatever the pairs in flight: single-frame fp32 encoder call on the SwinL pyramid, 6 images, 0.134 -> 0.089 ms)
        if (p.Lq != p.S && pairs * outside_bytes > l2_budget) continue;
        // (x2, + 1: half a tile's worth of fixed cost per workgroup -- slab staging, set-up; 8-image encoder call 0.072 -> 0.064 ms)
        const long long cost = ((wgs + cus - 1) / cus) * (2 * cand + 1);
        if (!pick || cost < best) { pick = cand; best = cost; }
    }
    return pick ? pick : fallback;
}

bool standard_value_layout(const Params &p)
{
    return p.v_clip == (int64_t)p.frames * p.S * p.M * p.D && p.v_head == p.D && p.v_pix == p.M * p.D;
}

// Can grad_value go through a scatter kernel (owner-computes or LDS atomics)?  MSDA_BWD_MODE=atomic forces the
// one-kernel backward with global atomics (kept for A/B measurements and as the any-shape path).
bool scatter_applicable(const Params &p)
{
    if (knobs().bwd_atomic) return false;
    if (p.L > kScatterMaxLevels || (p.D % 4) != 0 || p.D / 4 > kWave || ((p.D / 4) & (p.D / 4 - 1)) != 0) return false;   // D / 4 lanes per hit
    if (1 + p.frames * p.window > kScatterMaxSources || p.Lq >= (1 << 24)) return false;   // survivor-list entry fields
    if (p.window == 0 && p.LA != p.L) return false;
    if ((int64_t)p.groups * p.Lq >= 0x7fffffffLL) return false;       // query rows are 32-bit in the hit records
    return true;
}

// The owner-computes scatter (msda_bwd_value_grp_kernel) takes D = 32 with <= 4 points per level and the per-point
// culling records (or no records at all).
bool owner_scatter_applicable(const Params &p, int esz)
{
    // (work items -- (clip, frame, head, band) with at most one band per pixel row -- are counted in 32 bits)
    return p.D == 32 && (esz == 4 || esz == 2) && p.PA <= 4 && p.PB <= 4 && p.Lq < (1 << 22) && knobs().scatter_own != 0 &&
           knobs().scatter_lds_kb == 144 && (int64_t)p.groups * p.M * ((int64_t)p.S + p.L) < 0x7fffffffLL && scatter_applicable(p);
}

// The resident-slab kernels take D = 32 in 2- / 4-byte types when the index arithmetic fits and at least the last
// pyramid level fits the slab.  One predicate for forward and gather pass: a forward / backward pair never splits
// between kernel families on a shape limit.
bool rs_fits(const Params &p, int esz)
{
    const int64_t pixB = (int64_t)p.v_pix * esz;
    const int64_t pmax = p.PA > p.PB ? p.PA : p.PB;
    return p.D == 32 && (esz == 4 || esz == 2) && p.LA == p.L && p.L <= kSlabMaxLevels &&
           (int64_t)p.frames * p.S < (1 << 24) && pixB < (1 << 24) && (int64_t)p.frames * p.S * pixB < 0x7fffffffLL &&
           p.frames <= kRsMaxFrames && p.window <= 31 &&
           pmax * pmax * p.L < 65536;         // the kernels take a point's level as (k * ceil(2^16 / P)) >> 16
}

// Plan of the resident-window kernels (msda_win.hip; WinPlan in msda_common.h) for an encoder-shaped call: the largest tile
// whose rows fit a workgroup (frames * ceil(queries / 16) wave tiles <= 3 per wave), then the widest halo whose windows fit
// the LDS -- all levels at once when that halo reaches 6 pixels, else level 0 and the other levels in two staging phases.
// `force`: the test knob; without it the call must LOOK like an encoder (one query per pixel).
bool win_plan(const Params &p, int esz, bool force, WinPlan &w)
{
    if (!rs_fits(p, esz) || !p.shapes_host || p.L > kWinMaxLevels || p.L < 1) return false;
    long long pixels = 0;
    for (int l = 0; l < p.L; ++l) {
        if (p.shapes_host[2 * l] <= 0 || p.shapes_host[2 * l + 1] <= 0 || p.shapes_host[2 * l] > 16000 || p.shapes_host[2 * l + 1] > 16000) return false;       // (win_axis: 2 * n_l * n_0 in 32 bits)
        pixels += p.shapes_host[2 * l] * p.shapes_host[2 * l + 1];
    }
    if (pixels != p.Lq) return false;                 // the tiles enumerate the queries as the pixels of the pyramid
    (void)force;
    const int cap_px = kWinSlabBytes / (32 * esz), H0 = (int)p.shapes_host[0], W0 = (int)p.shapes_host[1];
    auto H = [&](int l) { return (int)p.shapes_host[2 * l]; };
    auto W = [&](int l) { return (int)p.shapes_host[2 * l + 1]; };
    // most pixels of level l a tile owns / needs in its window, per axis (maxima over the tiles)
    auto extent = [&](int n_l, int n_0, int B, int halo, bool window) {
        int best = 0;
        for (int t = 0; t < (n_0 + B - 1) / B; ++t) {
            int q0, q1, w0, w1;
            win_axis(n_l, n_0, t, B, halo, q0, q1, w0, w1);
            best = std::max(best, window ? w1 - w0 : q1 - q0);
        }
        return best;
    };
    // Tile sizes are tried from large to small; a size is taken when its rows fill the workgroup's wave tiles (16 waves x nt x
    // 16 rows) best -- rows of tiles at the map's edge and the idle tail of the last wave tiles cost as much as full ones.
    double best_score = 0.0;
    bool found = false;
    static const int kEdges[] = {24, 20, 16, 12, 10, 8, 6, 4};
    for (int By : kEdges) for (int Bx : kEdges) {
        if (By > 2 * Bx || Bx > 2 * By) continue;
        const int tiles_y = (H0 + By - 1) / By, tiles_x = (W0 + Bx - 1) / Bx;
        int nq = 0;
        for (int l = 0; l < p.L; ++l) nq += extent(H(l), H0, By, 0, false) * extent(W(l), W0, Bx, 0, false);
        const int tpg = (nq + kRsRows - 1) / kRsRows, nt = (p.frames * tpg + kRsWaves - 1) / kRsWaves;
        if (nq <= 0 || nt > 3) continue;                // (the forward keeps nt accumulator sets in registers)
        auto need = [&](int la, int lb, int halo) {       // LDS pixels of the windows of levels [la, lb) (each rounded to a DMA piece)
            int acc = 0;
            for (int l = la; l < lb; ++l)
                acc += (extent(H(l), H0, By, halo, true) * extent(W(l), W0, Bx, halo, true) + 15) / 16 * 16;
            return acc;
        };
        auto widest = [&](int la, int lb) {
            int h = -1;
            while (h < 16 && need(la, lb, h + 1) <= cap_px) ++h;
            return h;
        };
        int split = 0, h0 = widest(0, p.L), h1 = h0;
        const int min_halo = knobs().win_min_halo;
        if (h0 < min_halo && p.L > 1) {
            int best = h0;
            for (int sp = 1; sp < p.L; ++sp) {
                const int a = widest(0, sp), b = widest(sp, p.L);
                if (std::min(a, b) > best) { best = std::min(a, b); split = sp; h0 = a; h1 = b; }
            }
        }
        if (std::min(h0, h1) < min_halo) continue;      // (a corner outside its window costs a memory round trip of the whole wave)
        // useful rows per row slot of a workgroup, less the share of a source frame's time spent staging (estimated as the
        // window pixels per row served, one pixel ~ the LDS time of a sixth of a row's slot)
        const double rows = (double)p.Lq / ((double)tiles_y * tiles_x) * p.frames;
        const double staged = (split ? need(0, split, h0) + need(split, p.L, h1) : need(0, p.L, h0));
        const double score = rows / (nt * kRsWaves * kRsRows) / (1.0 + staged / (6.0 * rows)) / (split ? 1.08 : 1.0);
        if (score <= best_score) continue;
        best_score = score; found = true;
        w.By = By; w.Bx = Bx; w.tiles_y = tiles_y; w.tiles_x = tiles_x; w.split = split; w.halo[0] = h0; w.halo[1] = h1;
        w.tpg = tpg; w.nt = nt;
        int acc = 0;
        for (int l = 0; l < kWinMaxLevels; ++l) {
            if (l == split && split > 0) acc = 0;
            w.wbase[l] = acc;
            if (l < p.L) {
                const int halo = (split > 0 && l >= split) ? h1 : h0;
                acc += (extent(H(l), H0, By, halo, true) * extent(W(l), W0, Bx, halo, true) + 15) / 16 * 16;
            }
        }
    }
    if (!found) return false;
    return (long long)(p.groups / p.frames) * p.M * w.tiles_y * w.tiles_x <= 0x7fffffffLL;
}

// win_plan searches tile sizes and halos (~10^5 integer operations): the last plan is kept, keyed by everything it depends on.
bool win_plan_cached(const Params &p, int esz, bool force, WinPlan &w)
{
    struct Key { int L, frames, esz, Lq, min_halo; int64_t shapes[2 * kWinMaxLevels]; };
    static thread_local Key last_key;
    static thread_local WinPlan last_plan;
    static thread_local int last_state = -1;           // -1 nothing cached, 0 no plan, 1 plan
    if (!p.shapes_host || p.L < 1 || p.L > kWinMaxLevels || !rs_fits(p, esz)) return false;
    Key k;
    memset(&k, 0, sizeof k);
    k.L = p.L; k.frames = p.frames; k.esz = esz; k.Lq = p.Lq; k.min_halo = knobs().win_min_halo;
    for (int i = 0; i < 2 * p.L; ++i) k.shapes[i] = p.shapes_host[i];
    if (last_state < 0 || memcmp(&k, &last_key, sizeof k) != 0) {
        // (the remaining inputs of win_plan -- D, M, strides, window -- only gate it through rs_fits, checked above)
        last_state = win_plan(p, esz, force, last_plan) ? 1 : 0;
        last_key = k;
    }
    if (last_state != 1) return false;
    w = last_plan;
    return (long long)(p.groups / p.frames) * p.M * w.tiles_y * w.tiles_x <= 0x7fffffffLL;
}

// grad_value may be written in the 16-bit STORAGE type (include/msda.h, msda_grad_value_dtype) when the owner-computes
// scatter will produce it: that kernel overwrites every pixel exactly once from fp32 registers.  Every other route
// accumulates into grad_value (LDS-atomic flush aside, float atomics) and needs the arithmetic type.  Levels wider than a
// band take that kernel's float-atomic branch, so the host copy of the shapes must be there and say they do not occur.
bool storage_typed_grad_value_ok(int dtype, const Params &p)
{
    if (storage_dtype(dtype) != MSDA_BF16 && storage_dtype(dtype) != MSDA_F16) return false;
    if (knobs().force_generic || knobs().bwd_cull == 2 || !knobs().gv_storage) return false;
    if (!owner_scatter_applicable(p, 2) || !p.shapes_host) return false;
    for (int l = 0; l < p.L; ++l)
        if (p.shapes_host[2 * l + 1] > kOwnPix || p.shapes_host[2 * l + 1] <= 0) return false;
    // the shape conditions of fast_path_takes (pointer alignment is checked at the call: a mismatch is an error there)
    if (p.D % 8 || (int64_t)p.frames * p.S * p.M * p.D >= 0x7fffffffLL) return false;
    if (tile_lds_bytes(kWave / (p.D / 8), p.LA + p.LB, true) > 60 * 1024) return false;
    return true;
}

// Launches the forward, or the backward's gather pass + scatter, on the tile / resident-slab / scatter kernels.
int launch_fast(int dtype, const Params &p, bool bwd, hipStream_t stream)
{
    const int esz = elem_bytes(dtype), VEC = 16 / esz, G = p.D / VEC, RPW = kWave / G;
    const int64_t tiles = (int64_t)p.groups * ((p.Lq + RPW - 1) / RPW);
    const int64_t blocks = tiles * p.M;
    if (blocks > 0x7fffffffLL) return fail(MSDA_ERR_ARG, "msda: problem too large for one launch%s");
    const size_t lds = tile_lds_bytes(RPW, p.LA + p.LB, bwd, bwd && p.bbox != nullptr);
    const int rs_tiles_per_clip = p.frames * ((p.Lq + kRsRows - 1) / kRsRows);
    const int64_t clips = p.groups / p.frames;
    const int rs_row = 32 * esz;                                  // bytes of one pixel of one head
    const bool rs_ok = rs_fits(p, esz);
    const int l0_host = rs_ok ? host_first_slab_level(p, (kRsSlabBytes - kRsSlack) / rs_row) : p.L;
    const long long l2_budget = esz == 4 ? (2ll << 20) : (4ll << 20);        // see rs_tiles_per_wave

    // Encoder-shaped calls (one query per pixel) take the resident-window kernels when the slab of the resident-slab kernels
    // would hold the last level at most (fp32 at 800x1333: 273 of 22223 pixels) -- measured forward 2.38 -> 1.07 ms, gather pass 2.85 -> 1.60 ms there; where
    // more levels fit the slab (16-bit types, the 360x640 pyramid) the two families are on a par and the slab kernels stay.
    auto window_route = [&](int mode, WinPlan &w) {
        // (round 4 audit: also when a 4-byte slab holds only the last TWO levels -- SwinL 480x768 in fp32: forward 0.49 -> 0.37 ms,
        // gather pass 0.66 -> 0.55; 2-byte slabs of that kind -- 800x1333 bf16 -- are on a par and stay)
        // (temporal calls only: single-frame encoder calls on that pyramid are 6-26 % faster on the slab kernels)
        const bool few_levels = l0_host >= p.L - 1 || (esz == 4 && p.frames > 1 && p.L > 2 && l0_host >= p.L - 2);
        if (mode == 0 || (mode != 1 && !(p.Lq == p.S && p.L > 1 && few_levels))) return false;     // (cheap tests first)
        return win_plan_cached(p, esz, mode == 1, w);
    };
    if (!bwd) {
        WinPlan w;
        if (window_route(knobs().fwd_win, w)) return launch_fwd_win(dtype, p, w, stream);
        if (rs_ok) {
            // resident-slab forward: up to NT * 16 tiles of 16 rows per workgroup, so that the per-frame slab staging is
            // amortised; tiles per wave (NT) and workgroups per (clip, head) (parts): see rs_tiles_per_wave
            const int mode = knobs().fwd_rs;                               // -1 auto, 0 off, 1 force
            // at most 2 tiles per wave for slabs that start at level 2 (large pyramids) and for fp32: the 4-tile instantiations of
            // those slot bodies spill 10-40 VGPRs (profiles/r04_resource_usage.txt); BASELINE configs[1] forward 0.306 -> 0.290 ms,
            // SwinL 0.088 -> 0.082, 2-clip fp32 encoder call 0.52 -> 0.46
            const int max_nt = (l0_host >= 2 || esz == 4) ? 2 : 4;
            int nt = rs_tiles_per_wave(p, rs_tiles_per_clip, host_pixels_below(p, l0_host) * rs_row, mode == 1, l2_budget, max_nt);
            // the slab must hold at least the last level.  (Since the whole-row loads / stores of the points and gradients
            // the kernel wins for every dtype as soon as ANY level fits -- 800x1333, levels 2-3 resident.)
            if (mode != 1 && l0_host > p.L - 1) nt = 0;
            const int force_nt = knobs().fwd_rs_nt;
            if (force_nt == 1 || force_nt == 2 || force_nt == 4) nt = force_nt;
            const int parts = nt ? (rs_tiles_per_clip + kRsWaves * nt - 1) / (kRsWaves * nt) : 0;
            // few tiles per (clip, head) leave waves of the workgroups without one: 19 tiles (300 queries of a single-frame call)
            // on 2 x 16 waves -- 36-image decoder-like call in fp32 on the SwinL pyramid 0.078 ms here, 0.055 on the tile kernels
            // (only where the slab starts at level 2 in a 4-byte type, i.e. saves the least: elsewhere, and in the gather pass, the slab
            // kernels stay ahead by 4-19 %)
            if (mode != 1 && nt && esz == 4 && l0_host >= 2 && 10LL * rs_tiles_per_clip < 7LL * parts * kRsWaves * nt) nt = 0;
            if (mode != 0 && nt && clips * p.M * parts <= 0x7fffffffLL) {
                // fp32, one tile per wave, slab from level 2 on: the software-pipelined slot body of that instantiation spills 31 VGPRs
                // and its plain loop (the kernel compiled for a level-1 slab falls back to it) is 16-27 % faster on the SwinL pyramid
                // (decoder call, 4 / 16 / 32 clips: 0.175 -> 0.137, 0.644 -> 0.540, 1.178 -> 0.973 ms; 800x1333: the same)
                int body_l0 = (esz == 4 && nt == 1 && l0_host >= 2) ? 1 : l0_host;
                if (knobs().fwd_rs_body == 1 || knobs().fwd_rs_body == 2) body_l0 = knobs().fwd_rs_body;     // (A/B measurements)
                return launch_fwd_rs(dtype, nt, body_l0, p, parts, (unsigned)(clips * p.M * parts), stream);
            }
        }
        // SMALL forwards -- one clip at the 60 / 180 queries per frame of DeVIS's shipped configs is 192-1104 single-wave workgroups on
        // 1024 SIMDs, each walking its rows' 96 points as a chain of dependent gather batches -- put THREE waves on a tile, each with
        // a share of the tile's 16-point chunks, partial rows added through LDS in wave order (msda_fwd_tile_kernel, MW): 60 queries
        // fp32 0.020 -> 0.013 ms, fp16 0.034 -> 0.015; 180 queries fp16 0.041 -> 0.027; 300 queries bf16 0.043 -> 0.031.  With more
        // workgroups than SIMDs (fp32 from 180 queries on) the chip is busy anyway and the split only adds the exchange
        // (profiles/r04_logs/small_batch_tile_waves.log: no gain at 1824 workgroups).
        const int chunks = (p.LA * p.PA + kPch - 1) / kPch + (p.LB * p.PB + kPch - 1) / kPch;
        int waves = knobs().fwd_tile_waves;
        // (fewer single-wave workgroups than SIMDs -- 4 per CU; 4-byte types: than three quarters of them)
        if (waves < 0) waves = blocks <= (long long)device_cus() * (esz == 4 ? 3 : 4) ? 3 : 1;
        waves = std::max(1, std::min(std::min(waves, chunks), kTileMaxWaves));
        if (!(G == 4 || G == 8)) waves = 1;
        auto lds_of = [&](int w) { return (size_t)w * RPW * kRowSlots * 32 + (size_t)(p.LA + p.LB) * sizeof(Level) + (size_t)w * kWave * VEC * 4; };
        while (waves > 1 && lds_of(waves) > 48 * 1024) --waves;
        return launch_fwd_tile(dtype, G, p, (unsigned)blocks, waves > 1 ? lds_of(waves) : lds, stream, waves);
    }
    if (!scatter_applicable(p)) {
        if (p.gv_storage) return fail(MSDA_ERR_ARG, "msda backward: this call needs grad_value in the arithmetic type (see msda_grad_value_dtype)%s");
        if (hipMemsetAsync(p.grad_value, 0, (size_t)p.groups * p.S * p.M * p.D * sizeof(float), stream) != hipSuccess)
            return fail(MSDA_ERR_HIP, "msda backward: hipMemsetAsync(grad_value) failed%s");
        return launch_bwd_tile(dtype, G, true, p, (unsigned)blocks, lds, stream);
    }
    // Which levels the owner-computes scatter walks and which go to the matrix pipe is settled BEFORE the gather pass: the gather
    // pass leaves culling records only for the levels the owner kernel will walk band by band (16 clips: 0.400 -> 0.394 ms, 22 MB of stores less).
    int l0 = p.L, mfma_tiles = 0;
    const bool owner_route = owner_scatter_applicable(p, esz) && (p.cull_points || !p.bbox);
    if (owner_route) {
        // The coarse levels -- the last one or two of the pyramid, together at most ~300 pixels -- on the matrix pipe (msda_mfma.hip):
        // the owner-computes kernel then runs on levels [0, l0).  Needs the host copy of the shapes (a true copy: include/msda.h)
        // and at least 16 queries (a step is 16 groups).  Automatic for decoder-shaped batches: an item walks (1 + sources) x Lq
        // groups in 8 waves, so a handful of items of encoder length would be the kernel's whole duration.
        // (its loads are buffer loads with 32-bit byte offsets inside one clip: grad_out and the point arrays of a clip below 2 GiB)
        const long long lesz = (dtype == MSDA_BF16_LOC32 || dtype == MSDA_F16_LOC32) ? 4 : esz;
        const long long clip_rows = (long long)p.frames * p.Lq;
        const bool mfma_fits = clip_rows * p.M * p.D * esz < 0x7fffffffLL &&
                               clip_rows * p.M * std::max((long long)p.LA * p.PA, (long long)p.LB * p.PB) * 2 * lesz < 0x7fffffffLL;
        if (knobs().scatter_mfma != 0 && mfma_fits && p.shapes_host && p.Lq >= 16 && p.L >= 2 && !(knobs().scatter_own_levels >= 0 && knobs().scatter_own_levels < p.L)) {
            long long px = 0;
            for (int l = p.L - 1; l >= 1 && l >= p.L - 2; --l) {
                const long long hw = p.shapes_host[2 * l] * p.shapes_host[2 * l + 1];
                if (p.shapes_host[2 * l] <= 0 || p.shapes_host[2 * l + 1] <= 0 || !mfma_scatter_tiles(px + hw)) break;
                px += hw; l0 = l; mfma_tiles = mfma_scatter_tiles(px);
            }
            const long long items = (long long)p.groups * p.M, per_item = (long long)p.Lq * (1 + p.window);
            // Automatic rule (profiles/r06_logs/mfma_check.log; scatter pass, owner kernel alone -> with this kernel, ms): the two coarse
            // levels of the 360x640 pyramid cost the owner kernel 0.19 ms at 16 clips of 300 queries and this one 0.12 (0.563 -> 0.502;
            // bf16 0.564 -> 0.473; 4 / 8 / 32 clips 0.159 -> 0.148 / 0.289 -> 0.265 / 1.096 -> 1.064); the single 273-pixel level of the
            // 800x1333 pyramid 0.553 -> 0.512 (with the owner kernel's bands rotated unconditionally; see below).
            // It needs items to fill the chip -- 2 clips (96 items) 0.089 -> 0.110, one clip 0.053 -> 0.088 -- and items long enough to
            // pay for their zero-fill and reduction: the plain op on 48 images x 300 queries (19 steps per item) 0.100 -> 0.109.  Encoder-
            // shaped calls (one query per pixel: tens of thousands of groups per item) win once there are enough items -- 4 clips at
            // 360x640: 1.924 -> 1.733 -- and lose with one clip's 48 (0.556 -> 0.987; BASELINE configs[1], 64 items: 0.644 -> 0.688).
            const bool enough = items >= 128 && per_item >= 512 && (per_item <= 8192 || items >= 192);
            // (after the owner kernel's band rotation became conditional -- msda_scatter.hip -- the 96-pixel last level of the SwinL
            // pyramid pays as well: 16 clips 0.679 -> 0.636; the 273-pixel one of 800x1333 is level: 0.514 -> 0.510)
            if (mfma_tiles && knobs().scatter_mfma < 0 && !(enough && (p.L - l0 == 2 || px >= 64))) { l0 = p.L; mfma_tiles = 0; }
        }
    }
    // Culling records: not for the matrix-pipe levels, and not for levels of ONE band (the host copy of the shapes says so: every
    // group is a candidate of the only band, the owner kernel takes all its points) -- the 23x40 level of the 360x640 pyramid.
    unsigned rec_mask = ~0u;
    if (owner_route && !knobs().bwd_all_records) {
        for (int l = l0; l < p.L && l < 32; ++l) rec_mask &= ~(1u << l);
        for (int l = 0; p.shapes_host && l < l0 && l < 32; ++l) {
            const long long H = p.shapes_host[2 * l], W = p.shapes_host[2 * l + 1];
            if (H > 0 && W > 0 && W <= kOwnPix && H <= kOwnPix / W) rec_mask &= ~(1u << l);
        }
    }
    Params pq = p;                                 // the gather pass's view
    pq.rec_mask = rec_mask;
    // MSDA_BWD_PHASES (measurement hook for bench.py): 1 = gather pass only, 2 = scatter pass only
    // (needs the workspace a previous gather pass filled), 3 = both (default)
    const int phases = knobs().bwd_phases;
    int rc = MSDA_OK;
    if (phases & 1) {
        bool done = false;
        WinPlan w;
        if ((p.cull_points || !p.bbox) && window_route(knobs().bwd_win, w)) {
            rc = launch_bwd_win(dtype, pq, w, stream);
            if (rc) return rc;
            done = true;
        }
        if (!done && rs_ok && (p.cull_points || !p.bbox)) {
            // resident-slab gather pass: same applicability rule as the forward
            const int mode = knobs().bwd_rs;
            int tpw = rs_tiles_per_wave(p, rs_tiles_per_clip, host_pixels_below(p, l0_host) * rs_row, mode == 1, l2_budget,
                                        l0_host >= 2 ? 2 : 4);      // (as in the forward: configs[1] gather pass 0.407 -> 0.395 ms)
            if (knobs().bwd_rs_tpw > 0) tpw = knobs().bwd_rs_tpw;
            const int parts = tpw ? (rs_tiles_per_clip + tpw * kRsWaves - 1) / (tpw * kRsWaves) : 1;       // (L2: see the forward)
            const bool want = mode == 1 || (mode == -1 && tpw && l0_host <= p.L - 1);
            // One source frame per workgroup (round 4): the gather pass carries nothing from frame to frame, so (clip, head, frame,
            // half of the clip's tiles) workgroups stage ONE slab each and meet at no barrier afterwards -- a quarter of the staging
            // traffic of (clip, head, part) workgroups walking the frames.  Pays from ~8 clips on (same box, fp32:
            // 8 / 16 / 32 clips 0.245 -> 0.229 / 0.48 -> 0.44 / 0.881 -> 0.874 ms; 4 clips with TWO workgroups per frame
            // 0.100 -> 0.122 -- with four it pays there too, see below).
            // SMALL batches -- the one clip per GPU DeVIS itself issues (main.py:85) -- cannot fill the chip with (clip, head, part)
            // workgroups at all (tpw = 0) and used to fall to the tile kernels: with the frames as a workgroup index 1 / 2 clips make
            // 192 / 384 workgroups of <= 2 tiles per wave (same box, gather pass of 1 clip fp32 0.058 -> 0.040 ms, bf16 0.074 -> 0.037;
            // 2 clips 0.088 -> 0.063, 0.094 -> 0.065; profiles/r04_logs/small_batch_sweep.log).
            int fparts = knobs().bwd_rs_fsplit;
            bool want_small = false;
            if (fparts < 0) {
                // (round 4, second sweep, profiles/r04_logs/gather_fsplit_sweep.log: 2-byte types gain 5-10 % at 8 / 16 / 32 / 64 clips;
                // fp32 gains 4-11 % up to 32 clips and loses 3 % at 64)
                // Only while the levels outside the slab are small: the frame-split grid keeps 16 frame maps per XCD in flight instead
                // of 4 -- fine for the 360x640 pyramid's level 0 (460 KB in fp32), 12-16 % SLOWER on the 800x1333 one (levels 0-1
                // outside: 2.7 MB per map; 16 clips bf16 0.531 -> 0.618 ms, fp32 0.906 -> 1.012).
                const long long wgs = clips * p.M * p.frames * 2;
                // (route audit, profiles/r04_logs/route_audit_*.log: SwinL pyramid in fp32, 737 KB outside, 8-32 clips 14-20 % slower)
                const bool small_outside = host_pixels_below(p, l0_host) * rs_row <= (512ll << 10);
                // (encoder-shaped batches: 8 clips at 360x640 in fp32 2.50 -> 2.80 ms on this grid, in bf16 2.49 -> 2.13)
                fparts = (p.frames > 1 && small_outside && wgs >= 3LL * device_cus() &&
                          (esz == 2 || (wgs < 24LL * device_cus() && p.Lq != p.S))) ? 2 : 0;
                // (2-byte encoder-shaped batches one size below that: four workgroups per frame at 4 clips, 1.23 -> 1.13 ms at 360x640,
                // 1.94 -> 1.80 on the SwinL pyramid)
                if (!fparts && esz == 2 && p.frames > 1 && p.Lq == p.S && small_outside && 2 * wgs >= 3LL * device_cus()) fparts = 4;
                // fp32 batches whose (clip, head, part) workgroups are a single round over the CUs (4 clips): four workgroups per (clip,
                // head, frame) balance better -- 0.103 -> 0.099 / 0.177 -> 0.149 / 0.259 -> 0.242 ms on the three audited pyramids;
                // 2-byte types lose 2-14 % there and stay, and so do encoder-shaped calls (one clip is 232 workgroups of 4 tiles: 0.36 vs
                // 0.47 ms on the frame-split grid)
                if (!fparts && tpw && esz == 4 && mode == -1 && p.frames > 1 && p.Lq != p.S && l0_host <= p.L - 1 && clips * p.M * parts <= device_cus()) fparts = 4;
                if (!fparts && !tpw && mode == -1 && p.frames > 1 && l0_host <= p.L - 1 && clips * p.M * p.frames * 4 >= device_cus() / 2) {
                    // (2-byte types with two clips: 2 workgroups per (clip, head, frame) -- 0.067 -> 0.056 ms; fp32 the other way round)
                    // Whatever the query count -- DeVIS's shipped configs run 60 queries per frame (YouTube-VIS) and 180 (OVIS), 24 / 72
                    // tiles per clip: one clip of 60 queries 0.050 -> 0.020 ms in fp32, 0.065 -> 0.020 in fp16, 10 queries 0.042 -> 0.018
                    // (the tile kernels walk a chain of 24 dependent gather batches per wave however few rows there are)
                    fparts = (esz == 2 && clips * p.M * p.frames * 2 >= 3LL * device_cus() / 4) ? 2 : 4;
                    want_small = true;
                }
            }
            if ((want || want_small) && fparts > 0 && p.frames > 1 && clips * p.M * p.frames * fparts <= 0x7fffffffLL) {
                rc = launch_bwd_rs(dtype, l0_host, pq, fparts, (unsigned)(clips * p.M * p.frames * fparts), stream, 1);
                if (rc) return rc;
                done = true;
            } else if (want && clips * p.M * parts <= 0x7fffffffLL) {
                rc = launch_bwd_rs(dtype, l0_host, pq, parts, (unsigned)(clips * p.M * parts), stream);
                if (rc) return rc;
                done = true;
            }
        }
        if (!done) {
            rc = launch_bwd_tile(dtype, G, false, pq, (unsigned)blocks, lds, stream);
            if (rc) return rc;
        }
        if (p.cull_points && p.bsum) {       // block summaries of the per-point records just written
            rc = launch_cull_summary(pq, stream);
            if (rc) return rc;
        }
    }
    if (!(phases & 2)) return rc;
    unsigned grid = (unsigned)device_cus();      // persistent: one 1024-thread workgroup per CU
    grid -= grid % 8;                            // multiple of the XCD count: item % M stays put
    if (owner_route) {
        // owner-computes scatter: no float atomics; pixels outside its bands are zero-filled first
        // When every level's row fits a band (the host copy of the shapes says so) no pixel takes the float-atomic branch, and the
        // zero-fill of the pixels outside the levels -- normally none -- rides in the scatter kernel's prologue (bit 512) instead of
        // a launch of its own in front of it: one dependent dispatch less per backward (one clip from a HIP graph 0.127 -> see r04 logs)
        bool fused_zero = p.shapes_host != nullptr && (knobs().scatter_dbg & 1024) == 0;
        for (int l = 0; fused_zero && l < p.L; ++l) fused_zero = p.shapes_host[2 * l + 1] > 0 && p.shapes_host[2 * l + 1] <= kOwnPix;
        if (!fused_zero) {
            rc = launch_zero_unowned(p, kOwnPix * p.D, p.gv_storage ? 2 : 4, stream);
            if (rc) return rc;
        }
        Params pg = p;
        if (knobs().scatter_own_levels >= 0 && knobs().scatter_own_levels < p.L) pg.own_levels = knobs().scatter_own_levels;     // (measurement only: wrong results)
        if (l0 < p.L) pg.own_levels = l0;
        pg.rec_mask = rec_mask;
        if (!(mfma_tiles && knobs().scatter_part == 2))
            rc = launch_scatter_grp(dtype, p.gv_storage != 0, pg, grid * (1024 / kOwnThreads), (knobs().scatter_dbg & (511 | 2048 | 4096)) | (fused_zero ? 512 : 0), stream);
        if (rc || !mfma_tiles || knobs().scatter_part == 1) return rc;
        return launch_scatter_mfma(dtype, p.gv_storage != 0, p, l0, mfma_tiles, stream);
    }
    if (p.gv_storage) return fail(MSDA_ERR_ARG, "msda backward: this call needs grad_value in the arithmetic type (see msda_grad_value_dtype)%s");
    // LDS-atomic scatter: 144 KiB of 8-byte accumulators per workgroup
    const int cap_bytes = knobs().scatter_lds_kb * 1024;
    rc = launch_zero_unowned(p, cap_bytes / 8, 4, stream);
    if (rc) return rc;
    return launch_scatter_lds(dtype, p.D / 4, p, grid, cap_bytes, knobs().scatter_dbg, stream);      // 4 channels per lane
}

// Shapes the 16-byte-lane kernels take: D a multiple of the lane vector with 64 / G rows per wave, aligned bases,
// 32-bit element offsets inside a clip.
bool fast_path_takes(int dtype, const Params &p, bool bwd)
{
    if (dtype == MSDA_F64) return false;
    const int esz = elem_bytes(dtype), VEC = 16 / esz;
    if (p.D % VEC) return false;
    const int G = p.D / VEC;
    if (G != 1 && G != 2 && G != 4 && G != 8 && G != 16 && G != 32 && G != 64) return false;
    // every 16-B lane vector must be aligned: bases 16-B aligned and D a multiple of VEC
    if (!aligned16(p.value) || (!bwd && !aligned16(p.out)) || (bwd && (!aligned16(p.grad_out) || !aligned16(p.grad_value))))
        return false;
    // element offsets inside one clip slab are 32-bit in the tap records
    if ((int64_t)p.frames * p.S * p.M * p.D >= 0x7fffffffLL || (int64_t)p.frames * p.S * p.v_pix >= 0x7fffffffLL ||
        (int64_t)p.frames * p.S * p.v_pix * (int64_t)esz >= (int64_t)kOobBytes)  // gather_load: 32-bit byte offsets < kOobBytes
        return false;
    if (p.v_clip % VEC || p.v_head % VEC || p.v_pix % VEC) return false;
    // the one-kernel backward scatters grad_value (always dense) at value's offsets
    if (bwd && !scatter_applicable(p) && !standard_value_layout(p)) return false;
    if (tile_lds_bytes(kWave / G, p.LA + p.LB, bwd) > 60 * 1024) return false;
    return true;
}

int run(int dtype, const Params &p_in, bool bwd, hipStream_t stream)
{
    if (dtype < MSDA_F32 || dtype > MSDA_F16_LOC32) return fail(MSDA_ERR_DTYPE, "msda: unknown dtype code%s");
    if (MSDA_IS_TIMING_ONLY && !knobs().hooks)
        return fail(MSDA_ERR_ARG, "msda: this library is a TIMING-ONLY build (kernels that skip work: wrong results); it runs only "
                                  "with MSDA_ENABLE_HOOKS=1%s");
    Params p = p_in;
    p.own_levels = p.L;
    p.rec_mask = ~0u;
    const RouteScope pinned(bwd, dtype, p);     // (the pinned settings of this call shape, if any, are what knobs() answers below)
    p.dbg = knobs().dbg;
    // culling records per point (4 x int16) when the owner-computes scatter will read them; (min, max) intervals for the
    // LDS-atomic scatter (MSDA_BWD_CULL=2 forces them)
    p.cull_points = bwd && p.bbox && knobs().bwd_cull != 2 && owner_scatter_applicable(p, elem_bytes(dtype));
    if (!p.cull_points) p.bsum = nullptr;
    p.wide_stores = bwd && aligned16(p.glocA) && aligned16(p.gawA) && (p.LB == 0 || (aligned16(p.glocB) && aligned16(p.gawB))) &&
                    (knobs().dbg & 64) == 0;                  // (measurement: MSDA_DBG=64 keeps the narrow stores)
    p.wide_loads = aligned16(p.locA) && aligned16(p.awA) && (p.LB == 0 || (aligned16(p.locB) && aligned16(p.awB))) &&
                   (knobs().dbg & 128) == 0;                  // (measurement: MSDA_DBG=128 keeps the narrow loads)
    if (p.groups == 0 || p.Lq == 0) return MSDA_OK;
    if (!knobs().force_generic && fast_path_takes(dtype, p, bwd)) return launch_fast(dtype, p, bwd, stream);
    if (p.gv_storage) return fail(MSDA_ERR_ARG, "msda backward: this call needs grad_value in the arithmetic type (see msda_grad_value_dtype)%s");
    return launch_generic(dtype, p, bwd, stream);
}

// `grad_value_dtype` of the backward entry points: the arithmetic type, or the 16-bit storage type where allowed.
int set_grad_value_dtype(int dtype, int grad_value_dtype, Params &p)
{
    const int arith = dtype == MSDA_F64 ? MSDA_F64 : MSDA_F32;
    p.gv_storage = 0;
    if (grad_value_dtype == arith) return MSDA_OK;
    if (grad_value_dtype != storage_dtype(dtype) || !storage_typed_grad_value_ok(dtype, p))
        return fail(MSDA_ERR_ARG, "msda backward: grad_value_dtype must be what msda_grad_value_dtype returns for this call%s");
    p.gv_storage = 1;
    return MSDA_OK;
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

int zero_grad_value(int grad_value_dtype, void *grad_value, int groups, int S, int M, int D, void *stream)
{
    if (grad_value_dtype < MSDA_F32 || grad_value_dtype > MSDA_F16) return fail(MSDA_ERR_DTYPE, "msda: unknown dtype code%s");
    if (!grad_value) return fail(MSDA_ERR_ARG, "msda backward: null grad_value%s");
    const size_t bytes = (size_t)groups * S * M * D * (size_t)elem_bytes(grad_value_dtype);
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

int run_prep(int dtype, const PrepParams &p, bool bwd, void *stream)
{
    if (p.rows < 0 || p.M <= 0 || p.L <= 0 || p.W < 0 || p.Pc <= 0 || (p.W > 0 && p.Pt <= 0) || (p.d != 2 && p.d != 4))
        return fail(MSDA_ERR_ARG, "msda prep: bad sizes (rows, heads, levels, window, points, reference dim)%s");
    if (!p.shapes || !p.ref_c || (p.W > 0 && !p.ref_t)) return fail(MSDA_ERR_ARG, "msda prep: null pointer argument%s");
    if (p.rows == 0) return MSDA_OK;
    return launch_prep(dtype, p, bwd, static_cast<hipStream_t>(stream));
}

}  // namespace
}  // namespace msda

using namespace msda;

extern "C" {

int msda_version(void) { return MSDA_ABI_VERSION; }

const char *msda_build_info(void)
{
    return MSDA_IS_TIMING_ONLY ? "abi=13 arch=gfx950 timing_only=1" : "abi=13 arch=gfx950 timing_only=0";
}

void msda_reload_knobs(void) { load_knobs(); }

int msda_route_key(int backward, int dtype, int clips, int frames, int window, int spatial_size, int num_heads, int channels,
                   int num_levels, int num_query, int num_curr_point, int num_temp_point, const int64_t *spatial_shapes_host,
                   char *buf, int buf_len)
{
    if (!buf || buf_len <= 0 || !spatial_shapes_host || clips <= 0 || frames <= 0) return fail(MSDA_ERR_ARG, "msda_route_key: bad arguments%s");
    Params p;
    memset(&p, 0, sizeof p);
    p.groups = clips * frames; p.frames = frames; p.window = window; p.S = spatial_size; p.M = num_heads; p.D = channels;
    p.L = num_levels; p.Lq = num_query; p.PA = num_curr_point; p.PB = window > 0 ? num_temp_point : 1;
    p.shapes_host = spatial_shapes_host;
    const int n = route_key(buf, buf_len, backward != 0, dtype, p);
    return n < 0 ? fail(MSDA_ERR_ARG, "msda_route_key: the key does not fit the buffer (or more than 16 levels)%s") : n;
}

int msda_pin_route(const char *key, const char *settings)
{
    if (!key || !key[0]) return fail(MSDA_ERR_ARG, "msda_pin_route: empty key%s");
    RoutePin pin;
    pin.key = key;
    if (!parse_route_settings(settings, pin)) return fail(MSDA_ERR_ARG, "msda_pin_route: cannot parse the settings (name=value ...)%s");
    const bool remove = !settings || !settings[0];
    std::lock_guard<std::mutex> lock(g_routes_mutex);
    for (size_t i = 0; i < g_routes.size(); ++i)
        if (g_routes[i].key == pin.key) {
            if (remove) g_routes.erase(g_routes.begin() + (long)i); else g_routes[i] = pin;
            __atomic_store_n(&g_routes_n, (int)g_routes.size(), __ATOMIC_RELEASE);
            return MSDA_OK;
        }
    if (!remove) g_routes.push_back(pin);
    __atomic_store_n(&g_routes_n, (int)g_routes.size(), __ATOMIC_RELEASE);
    return MSDA_OK;
}

void msda_clear_routes(void)
{
    std::lock_guard<std::mutex> lock(g_routes_mutex);
    g_routes.clear();
    __atomic_store_n(&g_routes_n, 0, __ATOMIC_RELEASE);
}

int msda_route_count(void) { return __atomic_load_n(&g_routes_n, __ATOMIC_ACQUIRE); }

const char *msda_last_route(void) { return g_route; }

long long msda_backward_workspace_bytes(int batch, int num_query, int num_heads, int virtual_levels)
{
    return workspace_need(batch, num_query, num_heads, virtual_levels);
}

const char *msda_last_error(void) { return g_err; }

int msda_grad_value_dtype(int dtype, int clips, int frames, int window, int spatial_size, int num_heads, int channels,
                          int num_levels, int num_query, int num_curr_point, int num_temp_point,
                          const int64_t *spatial_shapes_host)
{
    const int arith = dtype == MSDA_F64 ? MSDA_F64 : MSDA_F32;
    if (clips <= 0 || frames <= 0 || window < 0 || spatial_size <= 0 || num_heads <= 0 || channels <= 0 || num_levels <= 0 ||
        num_query <= 0 || num_curr_point <= 0)
        return arith;
    Params p;
    memset(&p, 0, sizeof(p));
    p.groups = clips * frames; p.frames = frames; p.window = window;
    p.S = spatial_size; p.M = num_heads; p.D = channels; p.L = num_levels; p.Lq = num_query;
    p.LA = num_levels; p.PA = num_curr_point; p.LB = window * num_levels; p.PB = window > 0 ? num_temp_point : 1;
    p.shapes_host = spatial_shapes_host;
    return storage_typed_grad_value_ok(dtype, p) ? storage_dtype(dtype) : arith;
}

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
                  void *grad_value, int grad_value_dtype, void *grad_sampling_loc, void *grad_attn_weight,
                  void *workspace, long long workspace_bytes, const int64_t *value_strides,
                  const int64_t *spatial_shapes_host, void *stream)
{
    g_err[0] = 0; g_route[0] = 0;
    int rc = check_common(value, spatial_shapes, level_start_index, batch, spatial_size, num_heads,
                          channels, num_levels, num_query);
    if (rc) return rc;
    if (batch == 0) return MSDA_OK;
    if (num_query == 0) return zero_grad_value(grad_value_dtype, grad_value, batch, spatial_size, num_heads, channels, stream);
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
    rc = set_grad_value_dtype(dtype, grad_value_dtype, p);
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
                           void *grad_value, int grad_value_dtype, void *grad_loc_curr, void *grad_aw_curr,
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
        return zero_grad_value(grad_value_dtype, grad_value, clips * frames, spatial_size, num_heads, channels, stream);
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
    rc = set_grad_value_dtype(dtype, grad_value_dtype, p);
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
    return launch_mask_rows(g, static_cast<char *>(rows), static_cast<const uint8_t *>(padding_mask), pixels, (int)chunks, sb,
                            (unsigned)((threads + 255) / 256), static_cast<hipStream_t>(stream));
}

}  // extern "C"
