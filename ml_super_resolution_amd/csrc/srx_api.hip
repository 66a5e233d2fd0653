// srx_api.hip -- the C ABI of libsrx.so (include/srx.h): argument checking, tile planning,
// kernel-instance dispatch.  No device synchronisation, no allocation.
#include <math.h>
#include <stdarg.h>
#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/srx.h"
#include "elementwise.h"
#include "launchers.h"

using namespace srx;

namespace {

thread_local char g_err[512] = "";

// Tuning / A-B knobs from the environment, read ONCE (C++11 function-local statics are initialised thread-safely):
// the ABI promises concurrent calls from several host threads on different streams.
struct Knobs {
    int pipe_default, min_full_th, small_rule, th, grid, narrow, dyn, stagger, wgrad_lin, wgrad_pipe, wgrad_pipe_strip, wgrad_rows_full, wgrad_1x1, wgrad_pack3, wgrad_nt, kwrows_min_pixels, big_route_min_pixels, strip_d2s, conv_1x1_min_pixels, pack3_dgrad;
    int subpixel_chunk_kb, subpixel_db, subpixel_grid, subpixel_depth, subpixel_throttle, subpixel_even;
    unsigned long long* trace;
    int dbg;
};
int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
Knobs read_knobs() {
    Knobs k;
    k.pipe_default = env_int("SRX_PIPE", 1);
    k.min_full_th = env_int("SRX_MIN_FULL_TH", 3);   // full-width tiles of 1-2 rows re-stage 3 input rows per output row: measured slower than column tiles
    k.small_rule = env_int("SRX_SMALL_RULE", 1);     // 0: tallest tile that leaves two per slot (A/B)
    k.th = env_int("SRX_TH", -1);                    // force the tile height (tuning experiments; clamped to what fits)
    k.grid = env_int("SRX_GRID", -1);
    k.narrow = env_int("SRX_NARROW", 1);
    k.dyn = env_int("SRX_DYN", 0);
    k.stagger = env_int("SRX_STAGGER", -1);
    k.wgrad_lin = env_int("SRX_WGRAD_LIN", 1);
    k.wgrad_pipe = env_int("SRX_WGRAD_PIPE", 1);
    k.wgrad_pipe_strip = env_int("SRX_WGRAD_PIPE_STRIP", 1);   // 0: column-strip filter gradients on the two-workgroup kernel (A/B)
    // SRCNN's forward layers on their own kernels (5x5 32->3: conv_kwrows_kernel; 9x9 3->64: conv_pack3_kernel) from this many
    // output pixels (negative: never).  Round 4 started them at 60,000; timed against conv_mfma_kernel they win from the
    // smallest images on (one 100 x 100 image: level; 170 x 170: 42 against 73 us for the three layers)
    k.kwrows_min_pixels = env_int("SRX_KWROWS_MIN_PIXELS", 4096);
    // ESPCN's 5x5 3->64 on conv_pack3_kernel, 3x3 32->27 on conv_rows3x3_kernel (from 150,000), and the 5x5 32->3 filter
    // gradient on wgrad_kwcols_kernel: from this many output pixels (below: the one-launch ESPCN kernel's window and the
    // training patches, measured on the older kernels only)
    k.big_route_min_pixels = env_int("SRX_BIG_ROUTE_MIN_PIXELS", 60000);
    // 1x1 forward / data gradient (32 / 64 channels) on the streaming conv_1x1_kernel from this many pixels (negative: never)
    k.conv_1x1_min_pixels = env_int("SRX_CONV_1X1_MIN_PIXELS", 100000);
    k.pack3_dgrad = env_int("SRX_PACK3_DGRAD", 1);             // 0: the 5x5 32->3 layer's data gradient stays on conv_mfma_kernel (A/B)
    k.strip_d2s = env_int("SRX_STRIP_D2S", 1);                 // 0: ESPCN's f3 on wide images stays off the pipelined strip kernel (A/B)
    k.wgrad_rows_full = env_int("SRX_WGRAD_ROWS_FULL", 1);     // 0: 41-pixel rows on the padded-position walk (wgrad_pipe_kernel) instead of wgrad_rows_full_kernel (A/B)
    k.wgrad_1x1 = env_int("SRX_WGRAD_1X1", 1);                 // 0: 1x1 filter gradients on wgrad_mfma_kernel instead of the streaming wgrad_1x1_kernel (A/B)
    k.wgrad_pack3 = env_int("SRX_WGRAD_PACK3", 1);             // 0: RGB-input 9x9 / 5x5 filter gradients on the cursor kernel's 4-channel rows (A/B)
    k.wgrad_nt = env_int("SRX_WGRAD_NT", 1);                   // strip filter gradient: dpre loads marked non-temporal (A/B)
    k.subpixel_chunk_kb = env_int("SRX_SUBPIXEL_CHUNK_KB", 24);   // sub-pixel map: chunk size bound, double buffering,
    k.subpixel_db = env_int("SRX_SUBPIXEL_DB", 1);                // persistent-grid cap (tuning experiments)
    k.subpixel_grid = env_int("SRX_SUBPIXEL_GRID", 0);
    k.subpixel_depth = env_int("SRX_SUBPIXEL_DEPTH", 0);
    k.subpixel_throttle = env_int("SRX_SUBPIXEL_THROTTLE", -1);  // requests per wave in flight (0: unbounded; -1: the launcher's choice)
    k.subpixel_even = env_int("SRX_SUBPIXEL_EVEN", 1);           // 0: chunks of 4 blocks (subpixel_pipe_kernel); 1 + t: t trips
    k.trace = nullptr;
    k.dbg = 0;
#ifdef SRX_TRACE
    { const char* tp = getenv("SRX_TRACE_PTR"); k.trace = tp ? (unsigned long long*)strtoull(tp, nullptr, 0) : nullptr; }
    k.dbg = env_int("SRX_DBG", 0);                   // diagnostic builds only (make EXTRA=-DSRX_TRACE)
#endif
    return k;
}
const Knobs& knobs() { static const Knobs k = read_knobs(); return k; }

// conv kernel family, see srx_set_conv_path: -1 = not set by the caller (the environment's default applies)
std::atomic<int> g_use_pipe{-1};
int use_pipe() { const int v = g_use_pipe.load(std::memory_order_relaxed); return v < 0 ? knobs().pipe_default : v; }
// filter-gradient kernel family, see srx_set_wgrad_path: -1 = not set by the caller (the environment's defaults apply)
std::atomic<int> g_wgrad_path{-1};
int wgrad_path_default() { return !knobs().wgrad_lin ? 0 : ((knobs().wgrad_pipe || knobs().wgrad_pipe_strip) ? 2 : 1); }
int wgrad_path() { const int v = g_wgrad_path.load(std::memory_order_relaxed); return v < 0 ? wgrad_path_default() : v; }

// Compute units of the current device (persistent-workgroup grids are sized from it): queried once per device.
// Without a device (host-only workspace queries on a build box) the MI355X's 256.
int cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        n = 256;
    }
    if (n > 256) n = 256;     // grids and partial counts are planned for at most 2 x 256 persistent workgroups
    cached[dev].store(n, std::memory_order_relaxed);
    return n;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace
namespace srx {
// the same for the other translation units with extern "C" entry points (enet_ops.hip)
int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace srx
namespace {

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr size_t kLdsBudget = 80 * 1024 - 336;  // two workgroups per CU (160 KiB); 320 B spare for wgrad's zero slot
constexpr int kMaxGridLimit = 512;        // upper bound of any persistent grid (2 x 256 CUs): sizes the partial-count check
inline int max_grid() { return 2 * cu_count(); }   // two persistent workgroups per CU
inline int pipe_grid() { return cu_count(); }      // pipelined kernels: one persistent workgroup per CU

struct Plan {
    int KH, KW, cinp, nch;
    int pad_t, pad_l, OH, OW;
    int TH, TW, NTX, RS;
    int units_total, grid;
    size_t lds_bytes;
};

int pad_channels(int c) { return c <= 4 ? 4 : (c <= 32 ? 32 : (c <= 64 ? 64 : -1)); }
int chunks_for(int c) { return c <= 16 ? 1 : (c <= 32 ? 2 : (c <= 64 ? 4 : -1)); }

// in_c / out_c: channels of the tensor staged through LDS / produced by the kernel.
// (H,W) staged tensor dims, (OH,OW) produced tensor dims, pads relative to the staged tensor.
// lean_epilogue: the launch carries one of the epilogues the one-workgroup-per-CU kernels implement (none / ReLU;
// every dgrad and wgrad) -- only used to tell which kernel family the plan is for.
int make_plan(int N, int H, int W, int OH, int OW, int in_c, int out_c, int KH, int KW, int pad_t, int pad_l,
              Plan* p, bool lean_epilogue = true) {
    p->KH = KH; p->KW = KW;
    p->cinp = pad_channels(in_c);
    p->nch = chunks_for(out_c);
    if (p->cinp < 0 || p->nch < 0)
        return fail(SRX_ERR_UNSUPPORTED, "channel counts %d->%d outside the kernel set (<=64)", in_c, out_c);
    p->pad_t = pad_t; p->pad_l = pad_l; p->OH = OH; p->OW = OW;
    const int ps = (p->cinp == 4) ? 4 : p->cinp + 4;
    const size_t slot_bytes = (size_t)ps * 4;
    const size_t max_slots = kLdsBudget / slot_bytes;
    // full-width tiles: the zero column(s) left of row r+1 double as the right padding of row r
    int RS = W + pad_l;
    if (RS < OW + KW - 1 - (KW - 1 - pad_l)) RS = OW + pad_l;
    // A tile row holds the OW + KW - 1 input columns from -pad_l on; what does not fit wraps into the next row's
    // pad_l leading zero slots.  Even filter sizes with SAME padding pad one more column AFTER than before (TF:
    // pad_before = (K-1)/2), so their rows need one slot more than W + pad_l (round 2: found by the extended shape
    // sweep -- 2x2 / 4x4 SAME read the next row's first pixel as right padding).
    if (RS < OW + KW - 1 - pad_l) RS = OW + KW - 1 - pad_l;
    int TW = OW, NTX = 1;
    long th_max = ((long)max_slots - (KW - 1)) / RS - (KH - 1);
    const int kMaxGrid = max_grid();
    if (th_max < knobs().min_full_th) {
        // column tiling: each tile carries its own halo columns; narrow the tile until it fits
        for (TW = 32; TW >= 8; TW >>= 1) {
            RS = TW + KW - 1;
            th_max = ((long)max_slots - (KW - 1)) / RS - (KH - 1);
            if (th_max >= 1) break;
        }
        if (th_max < 1) return fail(SRX_ERR_UNSUPPORTED, "filter %dx%d too large for the LDS tile", KH, KW);
        NTX = (OW + TW - 1) / TW;
    }
    if (th_max > OH) th_max = OH;
    if (th_max > 16) th_max = 16;
    const long th_fit = th_max;
    // Small problems (fewer than two tallest tiles per workgroup slot): the launch is latency-bound, so pick the tile
    // height that minimises the time of the busiest workgroup instead of the tallest one.  Per tile a workgroup pays
    // one staging round trip (load -> LDS -> barrier, epilogue: ~1.5 units) plus one unit (a 16-pixel sub-tile's MFMA
    // chain over all taps, ~2 us for 3x3x64) per sub-tile its busiest wave owns.  Measured at 32x17x17, 64->32: one-row
    // tiles (the previous rule) 35-39 us per launch.
    const long rows_total = (long)N * NTX * OH;
    bool small = false, rows_split = false;
    if (!knobs().small_rule) {
        while (th_max > 1 && rows_total / th_max < 2 * kMaxGrid && rows_total >= 64) th_max -= 1;
    } else if (rows_total / th_max < 2 * kMaxGrid && KH == 3 && KW == 3 && p->cinp >= 16 && in_c == p->cinp && lean_epilogue &&
               (out_c & 3) == 0 && RS >= 256 / (p->cinp / 4) && (NTX > 1 ? p->nch == 4 : 16 * (4 / p->nch) <= OW)) {
        // ... except for the layers the one-workgroup-per-CU kernels take (3x3, exact-fit channels): they stage the next
        // tile under the running one and pay per group, not per tile, so the tallest tile stays best (measured at
        // 64x41x41: 4.57 ms per train step with 5-row tiles, 4.85 with 3, 5.15 with 2); only the row split must
        // still reach every CU.
        rows_split = true;
    } else if (rows_total / th_max < 2 * kMaxGrid) {
        const int npart = 4 / p->nch;
        double best_cost = 1e30;
        int best_th = 1;
        for (int th = 1; th <= (int)th_max; ++th) {
            const long tiles = (long)N * NTX * ((OH + th - 1) / th);
            const long slots = tiles < kMaxGrid ? tiles : kMaxGrid;
            const long per_wg = (tiles + slots - 1) / slots;
            const long sub = ((long)th * TW + 15) / 16;
            const double cost = (double)per_wg * (1.5 + (double)((sub + npart - 1) / npart));
            if (cost <= best_cost + 1e-9) { best_cost = cost; best_th = th; }   // ties: the taller tile (less halo)
        }
        th_max = best_th;
        small = true;
    }
    // among the tallest candidates pick the one wasting the fewest lanes of the 16-pixel sub-tiles
    int best = (int)th_max;
    double best_eff = 0.0;
    for (int th = (int)th_max; !small && th >= (int)((th_max + 1) / 2) && th >= 1; --th) {
        const int px = th * TW;
        const double eff = (double)px / (16.0 * ((px + 15) / 16));
        if (eff > best_eff + 1e-9) { best_eff = eff; best = th; }
    }
    if (knobs().th > 0) best = knobs().th < th_fit ? knobs().th : (int)th_fit;
    p->TH = best; p->TW = TW; p->NTX = NTX; p->RS = RS;
    p->units_total = (int)rows_total;
    long g = rows_total / p->TH;
    // (small problems: one workgroup per tile -- with tiles-per-image workgroups per image the static row split falls
    // on image boundaries, so no workgroup straddles two images and pays for two tiles)
    if (small) g = (long)N * NTX * ((OH + p->TH - 1) / p->TH);
    if (rows_split) g = rows_total;
    if (g < 1) g = 1;
    p->grid = (int)(g < kMaxGrid ? g : kMaxGrid);
    if (knobs().grid > 0 && knobs().grid < p->grid) p->grid = knobs().grid;
    p->lds_bytes = ((size_t)(p->TH + KH - 1) * RS + (KW - 1)) * slot_bytes;
    return SRX_OK;
}

int check_desc(const srx_conv_desc* d) {
    if (!d) return fail(SRX_ERR_BAD_ARG, "null descriptor");
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->KH <= 0 || d->KW <= 0)
        return fail(SRX_ERR_BAD_ARG, "non-positive dimension in descriptor");
    if (d->stride != 1 && d->stride != 2) return fail(SRX_ERR_UNSUPPORTED, "stride %d: strides 1 and 2 are implemented", d->stride);
    if (d->stride == 2 && d->subpixel_r > 1) return fail(SRX_ERR_UNSUPPORTED, "stride 2 with the sub-pixel store is not implemented");
    if (d->pad_mode != SRX_PAD_SAME && d->pad_mode != SRX_PAD_VALID) return fail(SRX_ERR_BAD_ARG, "bad pad_mode");
    if (d->act < SRX_ACT_NONE || d->act > SRX_ACT_SIGMOID) return fail(SRX_ERR_BAD_ARG, "bad activation");
    if (d->post_add_relu != 0 && d->post_add_relu != SRX_ACT_RELU && d->post_add_relu != SRX_ACT_LRELU)
        return fail(SRX_ERR_BAD_ARG, "post_add_relu must be 0, SRX_ACT_RELU (1) or SRX_ACT_LRELU (3)");
    if (d->precision != 0) return fail(SRX_ERR_UNSUPPORTED, "precision mode %d not implemented", d->precision);
    if (d->subpixel_r < 0 || d->subpixel_r > 16) return fail(SRX_ERR_BAD_ARG, "bad subpixel_r %d", d->subpixel_r);
    if (d->subpixel_r > 1 && d->Cout % (d->subpixel_r * d->subpixel_r))
        return fail(SRX_ERR_BAD_ARG, "subpixel_r %d: Cout %d is not a multiple of r*r", d->subpixel_r, d->Cout);
    if (d->pad_mode == SRX_PAD_VALID && (d->H < d->KH || d->W < d->KW))
        return fail(SRX_ERR_BAD_ARG, "VALID convolution with input smaller than the filter");
    if ((long)d->H * d->W * (long)(d->Cin > d->Cout ? d->Cin : d->Cout) >= (1L << 31))
        return fail(SRX_ERR_UNSUPPORTED, "one image has >= 2^31 elements: beyond the kernels' 32-bit in-image offsets");
    if ((long)d->N * d->H * 64 >= (1L << 31))
        return fail(SRX_ERR_UNSUPPORTED, "too many rows for 32-bit unit indexing");
    return SRX_OK;
}

// TensorFlow's geometry: SAME: out = ceil(in / s), pad_total = max((out - 1) s + k - in, 0), pad_before = pad_total / 2
// (the odd one goes AFTER: a stride-2 3x3 layer on an even-sized image pads 0 before / 1 after,
// enet/enet/model_enet.py:136-146; on an odd-sized one 1 / 1).  VALID: out = (in - k) / s + 1.
void geometry(const srx_conv_desc* d, int* pad_t, int* pad_l, int* OH, int* OW) {
    const int s = d->stride;
    if (d->pad_mode == SRX_PAD_SAME) {
        *OH = (d->H + s - 1) / s; *OW = (d->W + s - 1) / s;
        const int ph = (*OH - 1) * s + d->KH - d->H, pw = (*OW - 1) * s + d->KW - d->W;
        *pad_t = (ph > 0 ? ph : 0) / 2; *pad_l = (pw > 0 ? pw : 0) / 2;
    } else {
        *pad_t = 0; *pad_l = 0; *OH = (d->H - d->KH) / s + 1; *OW = (d->W - d->KW) / s + 1;
    }
}

// Stride 2 (forward and filter gradient on the two-workgroup kernels): every tile carries its own halo -- RS = (TW - 1) 2 +
// KW slots per tile row, (TH - 1) 2 + KH rows -- and output pixel (r, c) of a tile reads its window at tile slot (2r, 2c).
// Picks the column width that puts the most output pixels into the 80-KiB tile.
int make_plan_s2(int N, int H, int W, int OH, int OW, int in_c, int out_c, int KH, int KW, int pad_t, int pad_l, Plan* p) {
    p->KH = KH; p->KW = KW;
    p->cinp = pad_channels(in_c);
    p->nch = chunks_for(out_c);
    if (p->cinp < 0 || p->nch < 0)
        return fail(SRX_ERR_UNSUPPORTED, "channel counts %d->%d outside the kernel set (<=64)", in_c, out_c);
    p->pad_t = pad_t; p->pad_l = pad_l; p->OH = OH; p->OW = OW;
    const size_t slot_bytes = (size_t)((p->cinp == 4) ? 4 : p->cinp + 4) * 4;
    const long max_slots = (long)(kLdsBudget / slot_bytes) - 8;     // (the cursor wgrad kernel keeps a zeroed slot behind the tile)
    int best_tw = 0, best_th = 0;
    long best_px = 0;
    for (int TW = 32; TW >= 4; TW >>= 1) {
        const int tw = TW < OW ? TW : OW;
        const long RS = (long)(tw - 1) * 2 + KW;
        long rows = (max_slots - (KW - 1)) / RS;                 // tile rows that fit
        long th = (rows - KH) / 2 + 1;
        if (rows < KH || th < 1) continue;
        if (th > OH) th = OH;
        if (th > 16) th = 16;
        const long px = th * tw;
        if (px > best_px) { best_px = px; best_tw = tw; best_th = (int)th; }
        if (TW >= OW && TW > 4) continue;
    }
    if (!best_px) return fail(SRX_ERR_UNSUPPORTED, "filter %dx%d at stride 2 too large for the LDS tile", KH, KW);
    p->TW = best_tw; p->TH = best_th;
    p->NTX = (OW + best_tw - 1) / best_tw;
    p->RS = (best_tw - 1) * 2 + KW;
    const long rows_total = (long)N * p->NTX * OH;
    if (rows_total >= (1L << 31)) return fail(SRX_ERR_UNSUPPORTED, "too many rows for 32-bit unit indexing");
    p->units_total = (int)rows_total;
    long g = rows_total / p->TH;
    if (g < 1) g = 1;
    const int kMaxGrid = max_grid();
    p->grid = (int)(g < kMaxGrid ? g : kMaxGrid);
    p->lds_bytes = ((size_t)((p->TH - 1) * 2 + KH) * p->RS + (KW - 1)) * slot_bytes;
    return SRX_OK;
}

size_t part_stride(const srx_conv_desc* d) {
    const size_t per = (size_t)d->KH * d->KW * d->Cin * d->Cout + (size_t)d->Cout;
    return (per + 3) / 4 * 4;
}

int dispatch_conv(const Plan& p_in, bool wt, const ConvArgs& a_in, hipStream_t s, void* ws, size_t ws_bytes) {
    Plan p = p_in;
    ConvKey k{p.KH, p.KW, p.cinp, p.nch, wt};
    hipError_t err = hipSuccess;
    ConvArgs a = a_in;
    // Main path for the 3x3 body layers: the pipelined one-wave-per-SIMD kernel, one workgroup per CU with
    // two LDS tile buffers (measured 7 % faster than the two-workgroups-per-CU kernels at 256x41x41x64).
    // SRX_PIPE=0 / srx_set_conv_path(0) selects the two-workgroup kernels for everything (A/B).
    const int g_use_pipe = use_pipe();
    const int kMaxGrid = max_grid(), kPipeGrid = pipe_grid();
    // Three output channels (the RGB output layer): 16 lanes per pixel, no MFMA -- see conv_narrow.hip.  (The
    // mirror case, 3 -> 64 channels, was tried the same way and lost to the MFMA kernel: 56 vs 48 us.)
    // SRX_NARROW=0 keeps them on the MFMA kernels (A/B).
    {
        if (knobs().narrow && !a.d2s_r && a.stride == 1 && launch_conv_narrow(k, a, s, &err)) {
            if (err != hipSuccess) return fail(SRX_ERR_LAUNCH, "conv launch failed: %s", hipGetErrorString(err));
            return SRX_OK;
        }
    }
    // SRCNN's 5x5 32 -> 3 reconstruction layer: (kw, co) pairs as the MFMA's rows -- see conv_kwrows.hip; its 9x9 3 -> 64 and
    // ESPCN's 5x5 3 -> 64: (kw, ci) along K -- conv_pack3.hip (bit-identical to the kernels below: same order of the products).
    // conv_kwrows_kernel's results agree with the kernels below to rounding (the kw partial sums are added in another order);
    // the one-launch SRCNN kernel is bit-identical to the kernels below.
    // (srx_set_conv_path(0) keeps every shape on the conv_mfma_kernel family: the reference point of the bit-identity tests.)
    if (g_use_pipe && knobs().kwrows_min_pixels >= 0 &&
        (launch_conv_kwrows(k, a, knobs().kwrows_min_pixels, s, &err) ||                                      // (RGB-input 9x9 / 5x5: conv_pack3.hip)
         ((!k.wt || knobs().pack3_dgrad) &&
          launch_conv_pack3(k, a, k.kh == 9 ? knobs().kwrows_min_pixels : knobs().big_route_min_pixels, s, &err)))) {
        if (err != hipSuccess) return fail(SRX_ERR_LAUNCH, "conv launch failed: %s", hipGetErrorString(err));
        return SRX_OK;
    }
    // 1x1 layers on large inputs: HBM-bound, no LDS -- conv_1x1.hip (bit-identical to the kernels below)
    if (g_use_pipe && knobs().conv_1x1_min_pixels >= 0 && launch_conv_1x1(k, a, knobs().conv_1x1_min_pixels, s, &err)) {
        if (err != hipSuccess) return fail(SRX_ERR_LAUNCH, "conv launch failed: %s", hipGetErrorString(err));
        return SRX_OK;
    }
    // Its lean staging cursor / sub-tile walk / buffer-store epilogue cover: exact-fit channels, full-width
    // tiles, a row stride of at least one staging pass, sub-tile steps of at most one row, 16-byte output
    // vectors and images below 2^31 bytes.  Every other shape stays on the two-workgroup kernels.
    const int ppp = 256 / (p.cinp / 4), npart = 4 / p.nch;
    // Epilogue forms it implements (integer-VALU only): none / ReLU, ReLU-gradient mask, residual add (+ ReLU).
    const bool epi_ok = (a.act == ACT_NONE || a.act == ACT_RELU) && (a.post_relu == 0 || a.post_relu == ACT_RELU) && !(a.mask && a.skip) && !(wt && a.skip) && !(!wt && a.mask) &&
                        (!a.mask || (a.mask_act == ACT_RELU && a.act == ACT_NONE && !a.post_relu));
    // (p.RS == W + pad_l: the only pad slots of a tile row are the pad_l leading ones, which the scalar staging
    // never writes; a dgrad of a VALID layer has trailing pad slots inside the row as well and stays on path 0)
    const bool pipe_ok = epi_ok && !a.d2s_r && a.stride == 1 && a.Cin == p.cinp && p.NTX == 1 && p.RS >= ppp && p.RS == a.W + a.pad_l &&
                         16 * npart <= a.OW && (a.Cout & 3) == 0 &&
                         (long)a.H * a.W * a.Cin * 4 < (1L << 31) - 64 && (long)a.OH * a.OW * a.Cout * 4 < (1L << 31) - 64;
    // Column-strip variant (images too wide for full-width tiles): strips of 16 or 32 columns no wider than the
    // image, one sub-tile sequence per workgroup (64 output channels); any padding.
    // (32 output channels: forward only, no aux operand, and the one shape with a tanh form of the deferred epilogue -- ESPCN's f2)
    const bool strip2 = npart == 2 && !wt && !a.skip && !a.mask && !a.post_relu && a.Cout == 32 && p.cinp == 64 &&
                        (a.act == ACT_NONE || a.act == ACT_RELU || a.act == ACT_TANH);
    // (ESPCN's f3 -- 3x3 32 -> 3 r^2 through the sub-pixel map, no activation -- on the same two-chunk strips: AUX = 4)
    const bool strip_d2s = knobs().strip_d2s && npart == 2 && !wt && !a.skip && !a.mask && !a.post_relu && a.d2s_r > 0 && a.act == ACT_NONE &&
                           p.cinp == 32 && a.Cout > 16 && a.Cout <= 32 && k.kh == 3 && k.kw == 3 &&
                           (long)a.OH * a.d2s_r * a.OW * a.d2s_r * 3 * 4 < (1L << 31) - 64;
    const bool strip_ok = (epi_ok || strip2 || strip_d2s) && (!a.d2s_r || strip_d2s) && a.stride == 1 && a.Cin == p.cinp && p.NTX > 1 &&
                          (p.TW == 16 || p.TW == 32) && a.OW >= p.TW &&
                          p.RS >= ppp && (npart == 1 || strip2 || strip_d2s) && ((a.Cout & 3) == 0 || strip_d2s) &&
                          a.y != a.skip && a.y != a.mask &&   // (the columns two strips share are computed twice: no in-place epilogue operand)
                          (long)a.H * a.W * a.Cin * 4 < (1L << 31) - 64 && (long)a.OH * a.OW * a.Cout * 4 < (1L << 31) - 64;
    if (g_use_pipe && p.cinp >= 16 && strip_ok) {
        const int pgrid = p.grid < kPipeGrid ? p.grid : kPipeGrid;
        a.buf_floats = (int)(p.lds_bytes / 4);
        if (launch_pipe_strip(k, a, pgrid, 2 * p.lds_bytes, s, &err)) {
            if (err != hipSuccess) return fail(SRX_ERR_LAUNCH, "conv launch failed: %s", hipGetErrorString(err));
            return SRX_OK;
        }
    }
    if (g_use_pipe && p.cinp >= 16 && pipe_ok) {
        const int pgrid = p.grid < kPipeGrid ? p.grid : kPipeGrid;
        a.buf_floats = (int)(p.lds_bytes / 4);
        if (launch_pipe_k3c64(k, a, pgrid, 2 * p.lds_bytes, s, &err) ||
            launch_pipe_other(k, a, pgrid, 2 * p.lds_bytes, s, &err)) {
            if (err != hipSuccess) return fail(SRX_ERR_LAUNCH, "conv launch failed: %s", hipGetErrorString(err));
            return SRX_OK;
        }
    }
    // 3x3 layers of 17..32 output channels that the pipelined family did not take, on large inputs (ESPCN's f2 / f3 on whole
    // images): one workgroup of 8 waves per CU, see conv_rows3x3.hip.  Bit-identical to the kernels below.
    if (g_use_pipe && knobs().big_route_min_pixels >= 0 && launch_conv_rows3x3(k, a, knobs().big_route_min_pixels, s, &err)) {
        if (err != hipSuccess) return fail(SRX_ERR_LAUNCH, "conv launch failed: %s", hipGetErrorString(err));
        return SRX_OK;
    }
    // Dynamic tile scheduling (one atomic counter) for the two-workgroup kernels, OFF by default: measured
    // 2-5 % slower than the static row split on MI355X (the atomic's round trip is exposed once per tile,
    // and the "tail" of the static split is not wasted: the workgroup left alone runs unstarved).
    // SRX_DYN=1 enables it when the caller lends a counter word.
    {
        const int use_dyn = knobs().dyn;
        const int tiles_per_col = (p.OH + p.TH - 1) / p.TH;
        const long tiles_total = (long)a.N * p.NTX * tiles_per_col;
        a.tile_counter = nullptr;
        a.lds_sched_slot = (int)(p.lds_bytes / 4);
        if (use_dyn && ws && ws_bytes >= 64 && p.grid == kMaxGrid && tiles_total >= 2L * kMaxGrid &&
            tiles_total < (1L << 30)) {
            a.tile_counter = (int*)ws;
            a.tiles_total = (int)tiles_total;
            a.tiles_per_col = tiles_per_col;
            err = hipMemsetD32Async((hipDeviceptr_t)ws, p.grid, 1, s);   // first free tile = gridDim.x
            if (err != hipSuccess) return fail(SRX_ERR_LAUNCH, "counter memset failed: %s", hipGetErrorString(err));
        }
        p.lds_bytes += 16;   // mailbox word after the tile
    }
    bool hit = launch_conv_k3c64(k, a, p.grid, p.lds_bytes, s, &err) ||
               launch_conv_k3c32(k, a, p.grid, p.lds_bytes, s, &err) ||
               launch_conv_c4(k, a, p.grid, p.lds_bytes, s, &err) ||
               launch_conv_misc(k, a, p.grid, p.lds_bytes, s, &err) ||
               launch_conv_generic(k, a, p.grid, p.lds_bytes, s, &err);
    if (!hit)
        return fail(SRX_ERR_UNSUPPORTED, "no kernel instance for %dx%d, Cin<=%d, Cout chunks %d, %s", p.KH, p.KW,
                    p.cinp, p.nch, wt ? "dgrad" : "fwd");
    if (err != hipSuccess) return fail(SRX_ERR_LAUNCH, "conv launch failed: %s", hipGetErrorString(err));
    return SRX_OK;
}

// s_sleep(127) iterations (~3.4 us each) by which the second workgroup of a CU is delayed; only
// worth it when every CU really hosts two long-running workgroups.  SRX_STAGGER overrides.
int stagger_sleeps(const Plan& p) {
    const int env = knobs().stagger;
    if (env >= 0) return env;
    if (p.grid < max_grid()) return 0;
    const long tiles_per_wg = (long)p.units_total / ((long)p.grid * p.TH);
    return tiles_per_wg >= 2 ? 4 : 0;
}

void fill_conv_args(ConvArgs* a, const Plan& p, int N, int H, int W, int in_c, int out_c) {
    a->N = N; a->H = H; a->W = W; a->OH = p.OH; a->OW = p.OW; a->Cin = in_c; a->Cout = out_c;
    a->pad_t = p.pad_t; a->pad_l = p.pad_l;
    a->TH = p.TH; a->TW = p.TW; a->NTX = p.NTX; a->RS = p.RS;
    a->units_total = p.units_total;
    a->inv_rs = 1.0f / (float)p.RS;
    a->stride = 1;
    a->stagger = stagger_sleeps(p);
    a->dbg = knobs().dbg;        // both zero / null unless built with -DSRX_TRACE
    a->trace = knobs().trace;
}

}  // namespace

extern "C" {

const char* srx_version(void) { return "srx 0.1 (gfx950, fp32 MFMA 16x16x4)"; }
const char* srx_last_error(void) { return g_err; }
int srx_set_conv_path(int pipelined) {
    const int old = g_use_pipe.exchange(pipelined ? 1 : 0, std::memory_order_relaxed);
    return old < 0 ? knobs().pipe_default : old;
}
int srx_set_wgrad_path(int path) {
    const int old = g_wgrad_path.exchange(path < 0 ? -1 : (path > 2 ? 2 : path), std::memory_order_relaxed);
    return old < 0 ? wgrad_path_default() : old;
}
size_t srx_reduce_scratch_bytes(void) { return (size_t)kReduceBlocks * sizeof(float); }

size_t srx_conv2d_workspace_bytes(const srx_conv_desc* d, int op) {
    if (check_desc(d) != SRX_OK) return 0;
    if (op != SRX_OP_BWD_FILTER) return 256;   // optional: one counter word for dynamic tile scheduling
    int pt, pl, OH, OW;
    geometry(d, &pt, &pl, &OH, &OW);
    Plan p;
    if ((d->stride == 2 ? make_plan_s2(d->N, d->H, d->W, OH, OW, d->Cin, d->Cout, d->KH, d->KW, pt, pl, &p)
                        : make_plan(d->N, d->H, d->W, OH, OW, d->Cin, d->Cout, d->KH, d->KW, pt, pl, &p)) != SRX_OK) return 0;
    return (size_t)p.grid * part_stride(d) * sizeof(float);
}

int srx_conv2d_fwd(const srx_conv_desc* d, const float* x, const float* w, const float* bias, const float* skip,
                   float* y, void* ws, size_t ws_bytes, srx_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !w || !y) return fail(SRX_ERR_BAD_ARG, "null tensor pointer");
    if (!aligned16(x) || !aligned16(w) || !aligned16(y) || (skip && !aligned16(skip)))
        return fail(SRX_ERR_ALIGN, "tensor base pointers must be 16-byte aligned");
    int pt, pl, OH, OW;
    geometry(d, &pt, &pl, &OH, &OW);
    Plan p;
    if (d->stride == 2) rc = make_plan_s2(d->N, d->H, d->W, OH, OW, d->Cin, d->Cout, d->KH, d->KW, pt, pl, &p);
    else rc = make_plan(d->N, d->H, d->W, OH, OW, d->Cin, d->Cout, d->KH, d->KW, pt, pl, &p,
                        d->act == SRX_ACT_NONE || d->act == SRX_ACT_RELU);
    if (rc) return rc;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = w; a.bias = bias; a.skip = skip; a.mask = nullptr; a.y = y;
    fill_conv_args(&a, p, d->N, d->H, d->W, d->Cin, d->Cout);
    a.act = d->act; a.post_relu = d->post_add_relu; a.mask_act = 0;
    a.stride = d->stride;
    if (d->subpixel_r > 1) {
        // the store goes through the depth-to-space map: y is [N, OH*r, OW*r, Cout/(r*r)]
        if (skip) return fail(SRX_ERR_UNSUPPORTED, "subpixel_r with a skip operand is not implemented");
        a.d2s_r = d->subpixel_r;
        a.d2s_rc = d->Cout / d->subpixel_r;
    }
    return dispatch_conv(p, false, a, (hipStream_t)stream, ws, ws_bytes);
}

static int bwd_data_impl(const srx_conv_desc* d, const float* dpre, const float* w, const float* x_in, int in_act,
                         const float* dx_acc, float* dx_out, void* ws, size_t ws_bytes, srx_stream_t stream);

int srx_conv2d_bwd_data(const srx_conv_desc* d, const float* dpre, const float* w, const float* x_in, int in_act,
                        float* dx_out, void* ws, size_t ws_bytes, srx_stream_t stream) {
    return bwd_data_impl(d, dpre, w, x_in, in_act, nullptr, dx_out, ws, ws_bytes, stream);
}

int srx_conv2d_bwd_data_acc(const srx_conv_desc* d, const float* dpre, const float* w, const float* dx_acc, float* dx_out,
                            void* ws, size_t ws_bytes, srx_stream_t stream) {
    if (!dx_acc) return fail(SRX_ERR_BAD_ARG, "null tensor pointer");
    return bwd_data_impl(d, dpre, w, nullptr, SRX_ACT_NONE, dx_acc, dx_out, ws, ws_bytes, stream);
}

static int bwd_data_impl(const srx_conv_desc* d, const float* dpre, const float* w, const float* x_in, int in_act,
                         const float* dx_acc, float* dx_out, void* ws, size_t ws_bytes, srx_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!dpre || !w || !dx_out) return fail(SRX_ERR_BAD_ARG, "null tensor pointer");
    if (!aligned16(dpre) || !aligned16(w) || !aligned16(dx_out) || (x_in && !aligned16(x_in)) || (dx_acc && !aligned16(dx_acc)))
        return fail(SRX_ERR_ALIGN, "tensor base pointers must be 16-byte aligned");
    if (in_act < SRX_ACT_NONE || in_act > SRX_ACT_SIGMOID) return fail(SRX_ERR_BAD_ARG, "bad in_act");
    if (d->stride != 1)
        return fail(SRX_ERR_UNSUPPORTED, "bwd_data at stride %d: the data gradient of a stride-2 layer is the stride-1 data gradient of the "
                                         "zero-stuffed upstream gradient (srx_subsample2_bwd, then this entry point with stride 1)", d->stride);
    int pt, pl, OH, OW;
    geometry(d, &pt, &pl, &OH, &OW);
    // the kernel stages dpre [N,OH,OW,Cout] and produces dx [N,H,W,Cin]; full-correlation padding
    Plan p;
    rc = make_plan(d->N, OH, OW, d->H, d->W, d->Cout, d->Cin, d->KH, d->KW, d->KH - 1 - pt, d->KW - 1 - pl, &p);
    if (rc) return rc;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = dpre; a.w = w; a.bias = nullptr; a.skip = dx_acc; a.mask = x_in; a.y = dx_out;
    fill_conv_args(&a, p, d->N, OH, OW, d->Cout, d->Cin);
    a.act = SRX_ACT_NONE; a.post_relu = 0; a.mask_act = in_act;
    return dispatch_conv(p, true, a, (hipStream_t)stream, ws, ws_bytes);
}

// The linear-walk filter-gradient kernels keep 4 zeroed slots behind the tile (their last step may read past it): a
// plan whose tile fills the 80-KiB budget to the last slot (32-wide rows, 64 channels: 7 + 2 rows of 33 slots) would send
// the layer to the cursor kernel (the EnhanceNet generator's residual blocks at 64 x 32 x 32; at that size either kernel
// takes 58 us -- the launch is latency-bound -- but larger batches of such rows are not).  One row less and it fits.  The grid (= number of partial filters = workspace size) does not depend on the tile height.
static void wgrad_tile_for_linear_walk(const srx_conv_desc* d, Plan* p) {
    if (p->NTX != 1 || wgrad_path() < 1) return;
    const size_t slot_bytes = (size_t)((p->cinp == 4) ? 4 : p->cinp + 4) * 4;
    while (p->TH > 1 && p->lds_bytes + 4 * slot_bytes > 80 * 1024) {
        p->TH -= 1;
        p->lds_bytes = ((size_t)(p->TH + d->KH - 1) * p->RS + (d->KW - 1)) * slot_bytes;
    }
}

int srx_conv2d_bwd_filter_partials(const srx_conv_desc* d, const float* x, const float* dpre, void* ws, size_t ws_bytes,
                                   int* n_partials, srx_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!x || !dpre || !n_partials) return fail(SRX_ERR_BAD_ARG, "null tensor pointer");
    if (!aligned16(x) || !aligned16(dpre)) return fail(SRX_ERR_ALIGN, "tensor base pointers must be 16-byte aligned");
    int pt, pl, OH, OW;
    geometry(d, &pt, &pl, &OH, &OW);
    Plan p;
    const bool s2 = d->stride == 2;
    rc = s2 ? make_plan_s2(d->N, d->H, d->W, OH, OW, d->Cin, d->Cout, d->KH, d->KW, pt, pl, &p)
            : make_plan(d->N, d->H, d->W, OH, OW, d->Cin, d->Cout, d->KH, d->KW, pt, pl, &p);
    if (rc) return rc;
    if (!s2) wgrad_tile_for_linear_walk(d, &p);
    const size_t need = (size_t)p.grid * part_stride(d) * sizeof(float);
    if (!ws || ws_bytes < need)
        return fail(SRX_ERR_WORKSPACE, "bwd_filter needs %zu workspace bytes, got %zu", need, ws_bytes);
    if (!aligned16(ws)) return fail(SRX_ERR_ALIGN, "workspace must be 16-byte aligned");
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.dpre = dpre;
    a.part = (float*)ws;
    a.part_stride = (int)part_stride(d);
    a.N = d->N; a.H = d->H; a.W = d->W; a.OH = OH; a.OW = OW; a.Cin = d->Cin; a.Cout = d->Cout;
    a.pad_t = pt; a.pad_l = pl; a.TH = p.TH; a.TW = p.TW; a.NTX = p.NTX; a.RS = p.RS;
    a.units_total = p.units_total; a.inv_rs = 1.0f / (float)p.RS;
    a.stagger = stagger_sleeps(p);
    a.stride = d->stride;
    a.zero_slot = ((p.TH - 1) * d->stride + d->KH) * p.RS + (d->KW - 1);      // first slot after the largest tile
    a.trace = knobs().trace;   // diagnostic builds only
    const size_t wg_lds = p.lds_bytes + (size_t)((p.cinp == 4) ? 4 : p.cinp + 4) * 4;
    ConvKey k{d->KH, d->KW, p.cinp, p.nch, false};
    hipError_t err = hipSuccess;
    hipStream_t s = (hipStream_t)stream;
    // Linear-walk kernel for full-width tiles (see wgrad_lin_kernel); SRX_WGRAD_LIN=0 selects the cursor kernel (A/B).
    const int wpath = wgrad_path();                     // 0 cursor kernel, 1 two-workgroup linear walk, 2 one-workgroup pipelined (default)
    const int use_lin = wpath >= 1 && !s2;              // (stride 2: the cursor kernel, whose pixel cursor simply steps two slots)
    const int kPipeGrid = pipe_grid();
    const bool lin_ok = use_lin && p.NTX == 1 && OW >= 4 && p.RS >= 8 && p.RS == d->W + pl && (long)p.TH * OW * d->Cout * 4 < (1L << 30) &&
                        (long)d->H * d->W * d->Cin * 4 < (1L << 31) - 4096;
    // (its last step may read up to 3 slots past the tile: they are allocated and zeroed, their dpre operand is 0)
    const size_t lin_lds = p.lds_bytes + 4 * (size_t)((p.cinp == 4) ? 4 : p.cinp + 4) * 4;
    // one workgroup per CU, two tile buffers (wgrad_pipe_kernel): exact-fit channels, a row stride of at least one pass;
    // SRX_WGRAD_PIPE=0 keeps the two-workgroup kernel (A/B)
    const int use_wpipe = wpath >= 2 && knobs().wgrad_pipe;
    const int wppp = (p.cinp >= 16) ? 256 / (p.cinp / 4) : 256;
    int wgrid = p.grid;
    bool wdone = false;
    // 1x1 layers (SRCNN 64 -> 32, EnhanceNet's residual blocks 64 -> 64): a streaming GEMM over the pixels, no LDS tile
    if (knobs().wgrad_1x1 && wpath >= 1 && !s2) {
        int n1 = 0;
        if (launch_wgrad_1x1(k, a, p.grid, &n1, s, &err)) { wdone = true; wgrid = n1; }
    }
    // SRCNN's 5x5 32 -> 3 layer on large inputs: (kw, co) pairs as the MFMA's columns (conv_kwrows.hip)
    if (!wdone && knobs().big_route_min_pixels >= 0 && wpath >= 1 && !s2) {
        int n1 = 0;
        if (launch_wgrad_kwcols(k, a, p.grid, knobs().big_route_min_pixels, &n1, s, &err)) { wdone = true; wgrid = n1; }
    }
    // 41-pixel rows (the VDSR patch of BASELINE's metric): exact rows, one 31-step window per 3-row unit (wgrad_rows_full_kernel)
    if (!wdone && use_wpipe && knobs().wgrad_rows_full && lin_ok && d->KH == 3 && d->KW == 3 && d->Cin == 64 && d->Cout == 64 && OW == 41 && d->W == 41 &&
        pl == 1 && pt == 1 && p.RS == 42) {
        WgradArgs ap = a;
        ap.TH = 3;                                  // units of 3 rows: one window of 31 steps
        ap.zero_slot = (3 + 3) * 42 + 2;            // (a tile row more than the unit needs: the column step's idle lane group reads it)
        wgrid = p.grid < kPipeGrid ? p.grid : kPipeGrid;
        wdone = launch_wgrad_rows_full(k, ap, wgrid, 2 * (size_t)(ap.zero_slot + 4) * (64 + 4) * 4, s, &err);
        if (!wdone) wgrid = p.grid;
    }
    if (!wdone && use_wpipe && lin_ok && d->Cin == p.cinp && p.RS >= wppp && 2 * lin_lds <= 160 * 1024) {
        // Its step loop runs whole windows of 14 steps (4 positions each), padding a unit's last window with
        // zero-operand steps: pick the tile height that wastes the fewest (41-wide rows: 4 rows = 168 positions =
        // 3 windows exactly), and make sure the padded walk stays inside the tile buffer.
        const int kWin = 14 * 4;
        int best_th = 0;
        double best_waste = 1e9;
        for (int th = p.TH; th >= 1; --th) {
            const long pos = (long)th * p.RS, padded = (pos + kWin - 1) / kWin * kWin;
            const double waste = (double)(padded - pos) / (double)pos;
            const bool fits = padded + 2L * p.RS + 6 <= (long)a.zero_slot + 4;
            if (fits && waste < best_waste - 1e-9) { best_waste = waste; best_th = th; }
        }
        if (best_th > 0 && best_waste <= 0.10) {
            WgradArgs ap = a;
            ap.TH = best_th;
            wgrid = p.grid < kPipeGrid ? p.grid : kPipeGrid;
            wdone = launch_wgrad_pipe(k, ap, wgrid, 2 * lin_lds, s, &err);
            if (!wdone) wgrid = p.grid;
        }
    }
    {
        // 64 <-> 3 channel layers: 16 lanes per position, no MFMA (conv_narrow.hip); SRX_NARROW=0 for A/B
        if (!wdone && knobs().narrow && !s2) wdone = launch_wgrad_narrow(k, a, p.grid, s, &err);
    }
    // column strips: exact-fit channels and a tile row of at least one staging pass (the strip stager is the scalar one)
    // (and every strip, the narrower last one included, at least one 4-position step wide: the lanes of a step that
    // fall into the next tile row are taken to be real positions)
    const bool lin_strip_ok = use_lin && p.NTX > 1 && d->Cin == p.cinp && p.RS >= wppp && p.TW >= 4 && lin_lds <= 80 * 1024 &&
                              (OW % p.TW == 0 || OW % p.TW >= 4) && p.RS <= 3 * wppp &&
                              (long)d->H * d->W * d->Cin * 4 < (1L << 31) - 4096 && (long)OH * OW * d->Cout * 4 < (1L << 30);
    // ... on one workgroup per CU with two tile buffers, exact rows (wgrad_rows_strip_kernel): a 32-column strip row is 8
    // whole steps, so the K loop walks real pixels only -- no fake positions, any last-strip width >= 1;
    // SRX_WGRAD_PIPE_STRIP=0 / srx_set_wgrad_path(1) keep the two-workgroup padded walk (A/B).
    const bool rows_ok = wpath >= 2 && knobs().wgrad_pipe_strip && !s2 && p.NTX > 1 && p.TW == 32 && p.RS == 32 + d->KW - 1 &&
                         d->Cin == p.cinp && 2 * lin_lds <= 160 * 1024 &&
                         (long)d->H * d->W * d->Cin * 4 < (1L << 31) - 4096 && (long)OH * OW * d->Cout * 4 < (1L << 30);
    if (!wdone && rows_ok) {
        wgrid = p.grid < kPipeGrid ? p.grid : kPipeGrid;
        wdone = launch_wgrad_rows_strip(k, a, wgrid, 2 * lin_lds, knobs().wgrad_nt != 0, s, &err);
        if (!wdone) wgrid = p.grid;
    }
    if (!wdone && knobs().wgrad_pack3 && lin_ok && lin_lds <= 80 * 1024 && d->Cin == 3) wdone = launch_wgrad_lin_pack3(k, a, p.grid, lin_lds, s, &err);
    if (wdone) {
    } else if (lin_strip_ok && launch_wgrad_lin_strip(k, a, p.grid, lin_lds, s, &err)) {
    } else if (lin_ok && lin_lds <= 80 * 1024 && launch_wgrad_lin(k, a, p.grid, lin_lds, s, &err)) {
    } else if (!launch_wgrad(k, a, p.grid, wg_lds, s, &err) && !(wg_lds <= 80 * 1024 && launch_wgrad_generic(k, a, p.grid, wg_lds, s, &err)))
        return fail(SRX_ERR_UNSUPPORTED, "no wgrad instance for %dx%d, Cin<=%d, Cout chunks %d%s", d->KH, d->KW, p.cinp,
                    p.nch, s2 ? " at stride 2" : "");
    if (err != hipSuccess) return fail(SRX_ERR_LAUNCH, "wgrad launch failed: %s", hipGetErrorString(err));
    *n_partials = wgrid;
    return SRX_OK;
}

int srx_conv2d_bwd_filter_reduce(const srx_conv_desc* d, const void* ws, int n_partials, float* dw, float* dbias,
                                 const float* w_for_decay, float wd_scale, srx_stream_t stream) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!ws || !dw) return fail(SRX_ERR_BAD_ARG, "null tensor pointer");
    if (n_partials <= 0 || n_partials > kMaxGridLimit) return fail(SRX_ERR_BAD_ARG, "bad partial count %d", n_partials);
    const size_t wn = (size_t)d->KH * d->KW * d->Cin * d->Cout;
    hipError_t err = launch_reduce_partials((const float*)ws, n_partials, (int)part_stride(d), (int)wn, d->Cout, dw, dbias,
                                            w_for_decay, wd_scale, (hipStream_t)stream);
    if (err != hipSuccess) return fail(SRX_ERR_LAUNCH, "reduce launch failed: %s", hipGetErrorString(err));
    return SRX_OK;
}

int srx_conv2d_bwd_filter(const srx_conv_desc* d, const float* x, const float* dpre, float* dw, float* dbias,
                          const float* w_for_decay, float wd_scale, void* ws, size_t ws_bytes, srx_stream_t stream) {
    if (!dw) return fail(SRX_ERR_BAD_ARG, "null tensor pointer");
    int n = 0;
    const int rc = srx_conv2d_bwd_filter_partials(d, x, dpre, ws, ws_bytes, &n, stream);
    if (rc) return rc;
    return srx_conv2d_bwd_filter_reduce(d, ws, n, dw, dbias, w_for_decay, wd_scale, stream);
}

// ---- layers wider than 64 channels: the filter gradients of all (input block, output block) pairs -------------------
namespace {
void blocked_desc(srx_conv_desc* d, int N, int H, int W) {
    memset(d, 0, sizeof(*d));
    d->N = N; d->H = H; d->W = W; d->Cin = 64; d->Cout = 64; d->KH = 3; d->KW = 3; d->stride = 1; d->pad_mode = SRX_PAD_SAME;
}
// One launch for all pairs needs the linear-walk kernel (two workgroups per CU; full-width tiles, or column strips under
// the conditions of the single-layer entry point); G partials per pair so that pairs x G fills the chip about once.
// Returns 0 (one launch per pair), 1 (full-width tiles) or 2 (column strips).
int pairs_route(const srx_conv_desc* d, const Plan& p, int pairs, int* G, size_t* lin_lds) {
    *lin_lds = p.lds_bytes + 4 * (size_t)(p.cinp + 4) * 4;
    const bool common = wgrad_path() >= 1 && p.cinp == 64 && p.nch == 4 && d->W >= 4 && p.RS >= 8 && *lin_lds <= 80 * 1024 &&
                        (long)d->H * d->W * 64 * 4 < (1L << 31) - 4096;
    const bool lin_ok = common && p.NTX == 1 && p.RS == d->W + p.pad_l && (long)p.TH * d->W * 64 * 4 < (1L << 30);
    const int wppp = 256 / (64 / 4);
    const bool strip_ok = common && p.NTX > 1 && p.RS >= wppp && p.TW >= 4 && (d->W % p.TW == 0 || d->W % p.TW >= 4) &&
                          p.RS <= 3 * wppp && (long)d->H * d->W * 64 * 4 < (1L << 30);
    int g = max_grid() / pairs;
    if (g < 1) g = 1;
    if (g > p.grid) g = p.grid;
    *G = g;
    if (pairs <= 1 || pairs > 65535) return 0;
    return lin_ok ? 1 : (strip_ok ? 2 : 0);
}
}  // namespace

size_t srx_conv3x3_blocked_bwd_filter_workspace_bytes(int N, int H, int W, int staged_blocks, int produced_blocks) {
    srx_conv_desc d;
    blocked_desc(&d, N, H, W);
    if (check_desc(&d) || staged_blocks <= 0 || produced_blocks <= 0) return 0;
    Plan p;
    if (make_plan(N, H, W, H, W, 64, 64, 3, 3, 1, 1, &p)) return 0;
    wgrad_tile_for_linear_walk(&d, &p);
    int G;
    size_t lin_lds;
    const int pairs = staged_blocks * produced_blocks;
    const size_t one = (size_t)p.grid * part_stride(&d) * sizeof(float);
    if (!pairs_route(&d, p, pairs, &G, &lin_lds)) return one;
    const size_t all = (size_t)pairs * G * part_stride(&d) * sizeof(float);
    return all > one ? all : one;
}

int srx_conv3x3_blocked_bwd_filter(const float* x, const float* dpre, float* dw, float* dbias, int N, int H, int W,
                                   int staged_blocks, int produced_blocks, void* ws, size_t ws_bytes, srx_stream_t stream) {
    if (!x || !dpre || !dw) return fail(SRX_ERR_BAD_ARG, "null tensor pointer");
    if (staged_blocks <= 0 || produced_blocks <= 0) return fail(SRX_ERR_BAD_ARG, "non-positive block count");
    srx_conv_desc d;
    blocked_desc(&d, N, H, W);
    int rc = check_desc(&d);
    if (rc) return rc;
    if (!aligned16(x) || !aligned16(dpre) || !aligned16(dw) || !aligned16(ws))
        return fail(SRX_ERR_ALIGN, "tensor base pointers must be 16-byte aligned");
    const size_t need = srx_conv3x3_blocked_bwd_filter_workspace_bytes(N, H, W, staged_blocks, produced_blocks);
    if (!ws || ws_bytes < need) return fail(SRX_ERR_WORKSPACE, "blocked bwd_filter needs %zu workspace bytes, got %zu", need, ws_bytes);
    Plan p;
    rc = make_plan(N, H, W, H, W, 64, 64, 3, 3, 1, 1, &p);
    if (rc) return rc;
    // (32-wide rows: the plan's 7 + 2 rows of 33 slots fill the 80-KiB budget to the last slot and the linear-walk kernel's
    // zeroed slots no longer fit -- every pair would become its own launch; one row less, as on the single-layer entry)
    wgrad_tile_for_linear_walk(&d, &p);
    const int pairs = staged_blocks * produced_blocks;
    const size_t blk = (size_t)N * H * W * 64, wn = (size_t)9 * 64 * 64;
    int G;
    size_t lin_lds;
    const int route = pairs_route(&d, p, pairs, &G, &lin_lds);
    if (!route) {
        // one pair at a time on the single-layer entry point (odd shapes, a single pair)
        for (int ib = 0; ib < staged_blocks; ++ib)
            for (int ob = 0; ob < produced_blocks; ++ob) {
                rc = srx_conv2d_bwd_filter(&d, x + ib * blk, dpre + ob * blk, dw + ((size_t)ib * produced_blocks + ob) * wn,
                                           (ib == 0 && dbias) ? dbias + ob * 64 : nullptr, nullptr, 0.f, ws, ws_bytes, stream);
                if (rc) return rc;
            }
        return SRX_OK;
    }
    WgradPairs q;
    memset(&q, 0, sizeof(q));
    WgradArgs& a = q.a;
    a.x = x; a.dpre = dpre;
    a.part = (float*)ws;
    a.part_stride = (int)part_stride(&d);
    a.N = N; a.H = H; a.W = W; a.OH = H; a.OW = W; a.Cin = 64; a.Cout = 64;
    a.pad_t = 1; a.pad_l = 1; a.TH = p.TH; a.TW = p.TW; a.NTX = p.NTX; a.RS = p.RS;
    a.units_total = p.units_total; a.inv_rs = 1.0f / (float)p.RS;
    a.stagger = 0;
    a.stride = 1;
    a.zero_slot = (p.TH + 2) * p.RS + 2;
    a.trace = nullptr;
    q.x_pair_stride = (long)blk; q.d_pair_stride = (long)blk;
    q.part_pair_stride = (long)G * a.part_stride;
    q.cob = produced_blocks;
    ConvKey k{3, 3, 64, 4, false};
    hipError_t err = hipSuccess;
    if (!launch_wgrad_lin_pairs(k, q, G, pairs, route == 2, lin_lds, (hipStream_t)stream, &err))
        return fail(SRX_ERR_UNSUPPORTED, "no pairs instance of the linear-walk wgrad kernel");
    if (err != hipSuccess) return fail(SRX_ERR_LAUNCH, "blocked wgrad launch failed: %s", hipGetErrorString(err));
    err = launch_reduce_partials_pairs((const float*)ws, G, a.part_stride, (int)wn, 64, dw, dbias, pairs, produced_blocks,
                                       (hipStream_t)stream);
    if (err != hipSuccess) return fail(SRX_ERR_LAUNCH, "blocked reduce launch failed: %s", hipGetErrorString(err));
    return SRX_OK;
}

#define SRX_CHECK_LAUNCH(expr, what)                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail(SRX_ERR_LAUNCH, what ": %s", hipGetErrorString(e_)); \
        return SRX_OK;                                                                       \
    } while (0)

int srx_act_bwd(const float* dy, const float* y, float* dpre, size_t numel, int act, srx_stream_t stream) {
    if (!dy || !y || !dpre) return fail(SRX_ERR_BAD_ARG, "null tensor pointer");
    if (act < SRX_ACT_NONE || act > SRX_ACT_SIGMOID) return fail(SRX_ERR_BAD_ARG, "bad activation");
    if (numel == 0) return SRX_OK;
    SRX_CHECK_LAUNCH(launch_act_bwd(dy, y, dpre, numel, act, (hipStream_t)stream), "act_bwd");
}

static int subpixel(const float* in, float* out, int N, int H, int W, int C, int r, bool inv, srx_stream_t stream) {
    if (!in || !out) return fail(SRX_ERR_BAD_ARG, "null tensor pointer");
    if (N < 0 || H < 0 || W < 0 || C <= 0 || r <= 0) return fail(SRX_ERR_BAD_ARG, "bad sub-pixel dims");
    if (!aligned16(in) || !aligned16(out)) return fail(SRX_ERR_ALIGN, "tensor base pointers must be 16-byte aligned");
    if (in == out) return fail(SRX_ERR_BAD_ARG, "sub-pixel map cannot run in place");
    if (N == 0 || H == 0 || W == 0) return SRX_OK;
    const SubpixelTune tune = {knobs().subpixel_chunk_kb, knobs().subpixel_db,
                               knobs().subpixel_grid > 0 ? knobs().subpixel_grid : 4 * cu_count(), knobs().subpixel_depth, knobs().subpixel_throttle, knobs().subpixel_even};
    SRX_CHECK_LAUNCH(launch_subpixel(in, out, N, H, W, C, r, inv, tune, (hipStream_t)stream), "sub-pixel map");
}

int srx_stream_copy(const void* in, void* out, size_t bytes, srx_stream_t stream) {
    if (!in || !out) return fail(SRX_ERR_BAD_ARG, "null pointer");
    if (!aligned16(in) || !aligned16(out) || (bytes & 15u)) return fail(SRX_ERR_ALIGN, "stream copy moves whole 16-byte vectors");
    if (bytes == 0) return SRX_OK;
    SRX_CHECK_LAUNCH(launch_stream_copy((const float*)in, (float*)out, bytes, (hipStream_t)stream), "stream copy");
}

int srx_depth_to_space(const float* in, float* out, int N, int H, int W, int C, int r, srx_stream_t stream) {
    return subpixel(in, out, N, H, W, C, r, false, stream);
}
int srx_space_to_depth(const float* in, float* out, int N, int H, int W, int C, int r, srx_stream_t stream) {
    return subpixel(in, out, N, H, W, C, r, true, stream);
}

int srx_mse_fwd_bwd(const float* pred, const float* target, size_t numel, float inv_numel, float* loss_out,
                    int accumulate, float* dpred, void* scratch, srx_stream_t stream) {
    if (!pred || !target || !scratch) return fail(SRX_ERR_BAD_ARG, "null pointer");
    if (!aligned16(pred) || !aligned16(target) || (dpred && !aligned16(dpred)))
        return fail(SRX_ERR_ALIGN, "tensor base pointers must be 16-byte aligned");
    if (numel == 0) return SRX_OK;
    SRX_CHECK_LAUNCH(launch_mse(pred, target, numel, inv_numel, loss_out, accumulate, dpred, (float*)scratch,
                                (hipStream_t)stream), "mse");
}

int srx_l2_loss(const float* w, const float* mask, size_t numel, float scale, float* loss_out, int accumulate,
                void* scratch, srx_stream_t stream) {
    if (!w || !loss_out || !scratch) return fail(SRX_ERR_BAD_ARG, "null pointer");
    if (!aligned16(w) || (mask && !aligned16(mask)))
        return fail(SRX_ERR_ALIGN, "tensor base pointers must be 16-byte aligned");
    if (numel == 0) return SRX_OK;
    SRX_CHECK_LAUNCH(launch_l2(w, mask, numel, scale, loss_out, accumulate, (float*)scratch, (hipStream_t)stream), "l2");
}

int srx_adam_tf_step(float* w, const float* g, float* m, float* v, size_t numel, float lr, float beta1, float beta2,
                     float eps, int64_t t, float grad_scale, srx_stream_t stream) {
    if (!w || !g || !m || !v) return fail(SRX_ERR_BAD_ARG, "null pointer");
    if (t < 1) return fail(SRX_ERR_BAD_ARG, "Adam step count t must be >= 1");
    if (!aligned16(w) || !aligned16(g) || !aligned16(m) || !aligned16(v))
        return fail(SRX_ERR_ALIGN, "flat buffers must be 16-byte aligned");
    if (numel == 0) return SRX_OK;
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)t)) / (1.0 - pow((double)beta1, (double)t));
    SRX_CHECK_LAUNCH(launch_adam(w, g, m, v, numel, (float)lr_t, beta1, beta2, eps, grad_scale, (hipStream_t)stream),
                     "adam");
}

int srx_adam_tf_step_dev(float* w, const float* g, float* m, float* v, size_t numel, void* state, float beta1, float beta2,
                         float eps, float grad_scale, srx_stream_t stream) {
    if (!w || !g || !m || !v || !state) return fail(SRX_ERR_BAD_ARG, "null pointer");
    if (!aligned16(w) || !aligned16(g) || !aligned16(m) || !aligned16(v) || !aligned16(state))
        return fail(SRX_ERR_ALIGN, "flat buffers and the state block must be 16-byte aligned");
    if (numel == 0) return fail(SRX_ERR_BAD_ARG, "adam (device state): empty parameter buffer (the step count would not advance)");
    SRX_CHECK_LAUNCH(launch_adam_dev(w, g, m, v, numel, state, beta1, beta2, eps, grad_scale, (hipStream_t)stream), "adam (device state)");
}

int srx_momentum_clip_step(float* w, const float* g, float* acc, size_t numel, float lr, float momentum, float cap,
                           float grad_scale, srx_stream_t stream) {
    if (!w || !g || !acc) return fail(SRX_ERR_BAD_ARG, "null pointer");
    if (numel == 0) return SRX_OK;
    SRX_CHECK_LAUNCH(launch_momentum(w, g, acc, numel, lr, momentum, cap, grad_scale, (hipStream_t)stream), "momentum");
}

int srx_rownorm_loss_fwd_bwd(const float* pred, const float* target, size_t rows, size_t row_len, float* loss_out,
                             float* dpred, float* row_norms, srx_stream_t stream) {
    if (!pred || !target || !row_norms) return fail(SRX_ERR_BAD_ARG, "null pointer");
    if (rows == 0 || row_len == 0 || rows > (1u << 30)) return fail(SRX_ERR_BAD_ARG, "bad row-norm dims");
    SRX_CHECK_LAUNCH(launch_rownorm_loss(pred, target, rows, row_len, loss_out, dpred, row_norms, (hipStream_t)stream),
                     "rownorm loss");
}

int srx_psnr(const float* a, const float* b, float* out, int N, size_t per_image, float max_val, srx_stream_t stream) {
    if (!a || !b || !out) return fail(SRX_ERR_BAD_ARG, "null pointer");
    if (N <= 0 || per_image == 0) return fail(SRX_ERR_BAD_ARG, "bad psnr dims");
    SRX_CHECK_LAUNCH(launch_psnr(a, b, out, N, per_image, max_val, (hipStream_t)stream), "psnr");
}

size_t srx_ssim_scratch_bytes(int N) { return ssim_scratch_bytes(N); }

int srx_ssim(const float* a, const float* b, float* out, int N, int H, int W, int C, float max_val, void* scratch,
             srx_stream_t stream) {
    if (!a || !b || !out || !scratch) return fail(SRX_ERR_BAD_ARG, "null pointer");
    if (N <= 0 || C <= 0 || H < 11 || W < 11) return fail(SRX_ERR_BAD_ARG, "ssim needs N, C > 0 and H, W >= 11 (11x11 windows)");
    if (N > 65535) return fail(SRX_ERR_UNSUPPORTED, "ssim: more than 65535 images per call");
    SRX_CHECK_LAUNCH(launch_ssim(a, b, out, N, H, W, C, max_val, (float*)scratch, (hipStream_t)stream), "ssim");
}

int srx_saturate_u8(const float* x, uint8_t* out, size_t numel, srx_stream_t stream) {
    if (!x || !out) return fail(SRX_ERR_BAD_ARG, "null pointer");
    if (numel == 0) return SRX_OK;
    SRX_CHECK_LAUNCH(launch_saturate_u8(x, out, numel, (hipStream_t)stream), "saturate_u8");
}

int srx_affine(const float* x, float* out, size_t numel, float a, float b, srx_stream_t stream) {
    if (!x || !out) return fail(SRX_ERR_BAD_ARG, "null pointer");
    if (numel == 0) return SRX_OK;
    SRX_CHECK_LAUNCH(launch_affine(x, out, numel, a, b, (hipStream_t)stream), "affine");
}

int srx_u8_to_unit_float(const uint8_t* in, float* out, size_t numel, srx_stream_t stream) {
    if (!in || !out) return fail(SRX_ERR_BAD_ARG, "null pointer");
    if (numel == 0) return SRX_OK;
    SRX_CHECK_LAUNCH(launch_u8_to_float(in, out, numel, (hipStream_t)stream), "u8_to_unit_float");
}

int srx_gaussian_blur(const float* in, float* out, float* tmp, int N, int H, int W, int C, float sigma,
                      srx_stream_t stream) {
    if (!in || !out || !tmp) return fail(SRX_ERR_BAD_ARG, "null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return fail(SRX_ERR_BAD_ARG, "bad blur dims");
    if (in == out || tmp == in || tmp == out) return fail(SRX_ERR_BAD_ARG, "gaussian blur buffers must be distinct");
    if (sigma > 15.0f) return fail(SRX_ERR_UNSUPPORTED, "sigma %g: radius above 63 not supported", sigma);
    SRX_CHECK_LAUNCH(launch_gaussian_blur(in, out, tmp, N, H, W, C, sigma, (hipStream_t)stream), "gaussian_blur");
}

int srx_resize_bilinear(const float* in, float* out, int N, int H, int W, int C, int OH, int OW, srx_stream_t stream) {
    if (!in || !out) return fail(SRX_ERR_BAD_ARG, "null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) return fail(SRX_ERR_BAD_ARG, "bad resize dims");
    SRX_CHECK_LAUNCH(launch_resize_bilinear(in, out, N, H, W, C, OH, OW, (hipStream_t)stream), "resize_bilinear");
}

int srx_upsample_nearest(const float* in, float* out, int N, int H, int W, int C, int f, srx_stream_t stream) {
    if (!in || !out) return fail(SRX_ERR_BAD_ARG, "null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || f <= 0) return fail(SRX_ERR_BAD_ARG, "bad upsample dims");
    SRX_CHECK_LAUNCH(launch_upsample_nearest(in, out, N, H, W, C, f, (hipStream_t)stream), "upsample");
}

int srx_upsample_nearest_bwd(const float* dout, float* din, int N, int H, int W, int C, int f, srx_stream_t stream) {
    if (!dout || !din) return fail(SRX_ERR_BAD_ARG, "null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || f <= 0) return fail(SRX_ERR_BAD_ARG, "bad upsample dims");
    SRX_CHECK_LAUNCH(launch_upsample_nearest_bwd(dout, din, N, H, W, C, f, (hipStream_t)stream), "upsample_bwd");
}

int srx_debug_poison_lds(srx_stream_t stream) {
    hipError_t err = launch_poison_lds((hipStream_t)stream);
    if (err != hipSuccess) return fail(SRX_ERR_LAUNCH, "poison_lds launch failed: %s", hipGetErrorString(err));
    return SRX_OK;
}

int srx_add_relu_grad(const float* a, const float* b, const float* y, float* out, size_t numel, srx_stream_t stream) {
    if (!a || !b || !y || !out) return fail(SRX_ERR_BAD_ARG, "null pointer");
    if (numel == 0) return SRX_OK;
    SRX_CHECK_LAUNCH(launch_add_relu_grad(a, b, y, out, numel, (hipStream_t)stream), "add_relu_grad");
}

}  // extern "C"
