// b4d_kernels.hip -- C ABI (include/b4d.h) of the FFT -> PSD -> autocorrelation hot path
// (SURVEY.md §8 rows a1-a5).  Kernels live in b4d_fft2d.hpp / b4d_fft.hpp.
#include "b4d_passes.hpp"
#include "b4d_wiener_mr.hpp"

#include <cstring>

namespace b4d {
std::string& last_error() {
    static thread_local std::string err;
    return err;
}
int ensure_dynamic_lds(const void* kernel, size_t bytes) {
    struct Done {
        const void* kernel;
        int dev;
        size_t bytes;
    };
    static std::mutex mu;
    static std::vector<Done> done;
    int dev = 0;
    B4D_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    for (Done& d : done)
        if (d.kernel == kernel && d.dev == dev) {
            if (d.bytes >= bytes) return B4D_OK;
            B4D_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            d.bytes = bytes;
            return B4D_OK;
        }
    B4D_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done.push_back(Done{kernel, dev, bytes});
    return B4D_OK;
}
int lane_stream(int idx, hipStream_t* out) {
    static std::mutex mu;
    struct Made {
        int dev, idx;
        hipStream_t st;
    };
    static std::vector<Made> made;
    if (idx < 0 || idx > 1) return fail(B4D_EINVAL, "lane_stream: idx");
    int dev = 0;
    B4D_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    for (const Made& m : made)
        if (m.dev == dev && m.idx == idx) {
            *out = m.st;
            return B4D_OK;
        }
    hipStream_t st = nullptr;
    B4D_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    made.push_back(Made{dev, idx, st});
    *out = st;
    return B4D_OK;
}
std::atomic<int> g_opt_track_predict{1};
std::atomic<int> g_opt_lanes{1};
std::atomic<int> g_opt_exp{0};
// b4d_spectrum.hip
int spectrum_rows_last(const b4d_plan* pl, float2* spec, const float* frames, int batch, float2* out, hipStream_t st);
}  // namespace b4d

int Lanes::open(b4d_plan* p, hipStream_t s, int batch, size_t inter_bytes, size_t frame_bytes, bool allow_two, size_t min_group_bytes) {
    pl = p;
    st = s;
    const int want = (int)std::max<size_t>(1, ((size_t)64 << 20) / std::max<size_t>(1, inter_bytes));
    // two lanes need two slots of the workspaces and enough work to pay for the fork / join (a few microseconds of host time)
    two = allow_two && b4d::g_opt_lanes.load() != 0 && batch >= 2 && p->chunk >= 2 && (size_t)batch * frame_bytes >= 2 * min_group_bytes &&
          (size_t)std::min(want, p->chunk / 2) * frame_bytes >= min_group_bytes;
    sub = two ? std::min(want, p->chunk / 2) : p->chunk;   // one lane: the plan's own groups
    if (two) {   // an even number of groups of equal size
        int groups = (batch + sub - 1) / sub;
        groups += groups & 1;
        sub = (batch + groups - 1) / groups;
    }
    return fork(p, s, two);
}

int Lanes::fork(b4d_plan* p, hipStream_t s, bool want_two) {
    pl = p;
    st = s;
    two = want_two && b4d::g_opt_lanes.load() != 0;
    if (!two) return B4D_OK;
    if (!p->aux) {
        const int rs = b4d::lane_stream(0, &p->aux);
        if (rs) return rs;
    }
    if (!p->ev_fork) B4D_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
    if (!p->ev_join) B4D_HIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
    B4D_HIP(hipEventRecord(p->ev_fork, st));
    B4D_HIP(hipStreamWaitEvent(p->aux, p->ev_fork, 0));
    return B4D_OK;
}

int Lanes::close() {
    if (!two) return B4D_OK;
    B4D_HIP(hipEventRecord(pl->ev_join, pl->aux));
    B4D_HIP(hipStreamWaitEvent(st, pl->ev_join, 0));
    return B4D_OK;
}

namespace b4d {
}  // namespace b4d

extern "C" {

const char* b4d_version(void) { return "b4d 0.3.0 (gfx950)" B4D_VERSION_SUFFIX; }
const char* b4d_last_error(void) { return last_error().c_str(); }
int b4d_set_option(const char* name, int value) {
    if (!name) return fail(B4D_EINVAL, "null option name");
    struct Opt {
        const char* name;
        std::atomic<int>* v;
        int lo, hi;
    };
    const Opt opts[] = {{"track_predict_bin", &g_opt_track_predict, 0, 2}, {"lanes", &g_opt_lanes, 0, 1}, {"exp", &g_opt_exp, 0, 255}};
    for (const Opt& o : opts)
        if (!strcmp(name, o.name)) {
            if (value < o.lo || value > o.hi)
                return fail(B4D_EINVAL, std::string(name) + " takes " + std::to_string(o.lo) + " .. " + std::to_string(o.hi));
            o.v->store(value);
            return B4D_OK;
        }
    return fail(B4D_EINVAL, std::string("unknown option: ") + name);
}
static bool large_ok(int ny, int nx) {
    return ny >= 2 && nx >= 2 && ny <= 8192 && nx <= 8192 && (size_t)ny * nx <= ((size_t)1 << 26) && pm_supported(ny) && pm_supported(nx);
}
int b4d_size_supported(int ny, int nx) {
    return (pow2_ok(ny) && pow2_ok(nx)) || (general_ok(ny) && general_ok(nx)) || large_ok(ny, nx);
}

static int plan_create_impl(int ny, int nx, int chunk, bool force_general, b4d_plan** out);
int b4d_plan_create(int ny, int nx, int chunk, b4d_plan** out) { return plan_create_impl(ny, nx, chunk, false, out); }
// complex-to-complex transforms (b4d_fft2d_c2c) run on the general-length engines for every size, powers of two included
int b4d_plan_create_general(int ny, int nx, int chunk, b4d_plan** out) {
    if (out) *out = nullptr;
    if (!((general_ok(ny) && general_ok(nx)) || large_ok(ny, nx)))
        return fail(B4D_ESIZE, "general plans need ny, nx <= 512 or sides up to 8192 that split as 2^k * A * B; got " +
                                   std::to_string(ny) + "x" + std::to_string(nx));
    return plan_create_impl(ny, nx, chunk, true, out);
}

static int plan_create_impl(int ny, int nx, int chunk, bool force_general, b4d_plan** out) {
    if (!out) return fail(B4D_EINVAL, "out is null");
    *out = nullptr;
    if (!b4d_size_supported(ny, nx))
        return fail(B4D_ESIZE, "plans need power-of-two ny, nx in [64, 4096] (FFT kernels), any ny, nx in [2, 512] "
                               "(DFT-matrix path), sides up to 8192 that split as 2^k * A * B with A + B <= 128 (fused "
                               "mixed-radix path) or any other side up to 4096 (Bluestein); got " + std::to_string(ny) + "x" + std::to_string(nx));
    if (chunk < 1) return fail(B4D_EINVAL, "chunk must be >= 1");
    b4d_plan* p = new b4d_plan();
    p->ny = ny;
    p->nx = nx;
    p->chunk = chunk;
    if (force_general || !(pow2_ok(ny) && pow2_ok(nx))) {  // general-length plan
        p->general = true;
        p->large = !(general_ok(ny) && general_ok(nx));
        p->wmr = wmr_supported(ny) && wmr_supported(nx);
        int rc = B4D_OK;
        if (p->wmr && !p->large) {   // small frames on the mixed-radix passes keep their DFT matrices for the complex entry points
            rc = make_twiddles(nx, &p->tw_x);
            if (rc == B4D_OK) rc = make_twiddles(ny, &p->tw_y);
        }
        if (rc != B4D_OK) {
            b4d_plan_destroy(p);
            return rc;
        }
        if (p->large) {
            rc = make_twiddles(nx, &p->tw_x);
            if (rc == B4D_OK) rc = make_twiddles(ny, &p->tw_y);
        } else {
            rc = make_dft_matrix(nx, &p->wx);
            if (rc == B4D_OK) rc = make_dft_matrix(ny, &p->wy);
        }
        if (rc != B4D_OK) {
            b4d_plan_destroy(p);
            return rc;
        }
        p->ws_bytes = 3 * sizeof(float2) * (size_t)chunk * ny * nx;
        hipError_t e = hipMalloc((void**)&p->gbuf1, p->ws_bytes / 3);
        if (e == hipSuccess) e = hipMalloc((void**)&p->gbuf2, p->ws_bytes / 3);
        if (e == hipSuccess) e = hipMalloc((void**)&p->gbuf3, p->ws_bytes / 3);
        if (e != hipSuccess) {
            b4d_plan_destroy(p);
            return fail(B4D_ENOMEM, std::string("workspace allocation failed: ") + hipGetErrorString(e));
        }
        *out = p;
        return B4D_OK;
    }
    p->ct_w = col_ct(ny);
    int rc = make_twiddles(nx, &p->tw_x);
    if (rc == B4D_OK) rc = make_twiddles(ny, &p->tw_y);
    if (rc != B4D_OK) {
        b4d_plan_destroy(p);
        return rc;
    }
    p->ws_bytes = sizeof(float2) * (size_t)chunk * ny * (nx / 2);
    hipError_t e = hipMalloc((void**)&p->spec, p->ws_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&p->nyq_rows, sizeof(float) * (size_t)chunk * ny);
    if (e == hipSuccess) e = hipMalloc((void**)&p->gnyq, sizeof(float) * (size_t)chunk * ny);
    if (e == hipSuccess) e = hipMalloc((void**)&p->peak, sizeof(float) * chunk);
    if (e != hipSuccess) {
        b4d_plan_destroy(p);
        return fail(B4D_ENOMEM, std::string("workspace allocation failed: ") + hipGetErrorString(e));
    }
    *out = p;
    return B4D_OK;
}

int b4d_plan_destroy(b4d_plan* p) {
    if (!p) return B4D_OK;
    if (p->tw_x) (void)hipFree(p->tw_x);
    if (p->tw_y) (void)hipFree(p->tw_y);
    if (p->spec) (void)hipFree(p->spec);
    if (p->peak) (void)hipFree(p->peak);
    if (p->nyq_rows) (void)hipFree(p->nyq_rows);
    if (p->gnyq) (void)hipFree(p->gnyq);
    if (p->track_ws) (void)hipFree(p->track_ws);
    if (p->aux) (void)hipStreamSynchronize(p->aux);   // the library's shared lane stream: not destroyed with the plan
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    for (float2* q : {p->wx, p->wy, p->gbuf1, p->gbuf2, p->gbuf3})
        if (q) (void)hipFree(q);
    delete p;
    return B4D_OK;
}

size_t b4d_plan_workspace_bytes(const b4d_plan* p) { return p ? p->ws_bytes : 0; }

#ifdef B4D_DIAG
static unsigned long long* g_diag = nullptr;
extern "C" void b4d_debug_set_diag(void* buf) { g_diag = static_cast<unsigned long long*>(buf); }
#endif

// Shared body.  With `kernel_ms` != null every kernel launch is bracketed by HIP events on `st`
// and the per-kernel elapsed times (ms; row R2C, column, peak, row C2R) are ADDED to kernel_ms[0..3]
// after a final hipEventSynchronize -- used by bench.py to price each kernel inside its timed region.
static int psd_autocorr_impl(b4d_plan* pl, const float* frames, int batch, float* psd, float psd_scale,
                             float* autocorr, unsigned flags, hipStream_t st, float* kernel_ms) {
    if (!pl) return fail(B4D_EINVAL, "plan is null");
    B4D_PLAN_LOCK(pl);
    if (!pl || !frames) return fail(B4D_EINVAL, "null plan or input");
    if (batch < 1) return fail(B4D_EINVAL, "batch must be >= 1");
    if (!psd && !autocorr) return fail(B4D_EINVAL, "both outputs are null");
    if (pl->general) return general_psd_autocorr(pl, frames, batch, psd, psd_scale, autocorr, flags, st);
    const size_t fpix = (size_t)pl->ny * pl->nx;
    std::vector<hipEvent_t> ev;
    auto mark = [&]() -> int {
        if (!kernel_ms) return B4D_OK;
        hipEvent_t e;
        B4D_HIP(hipEventCreate(&e));
        ev.push_back(e);
        B4D_HIP(hipEventRecord(e, st));
        return B4D_OK;
    };
    // 4096^2 frames (and larger pixel counts): the launch groups are dealt to two lanes (Lanes, b4d_fft2d.hpp), half the plan's chunk
    // each, every lane in its own slot of the workspaces: 9.2 -> 9.75 k frames/s (tools/dev_pipe_chunk.py).  Not at 2048^2 and
    // below (+1 % ... 0: the column kernel owns the CUs' register files and nothing runs under it), and never on the timed variant.
    int rc = B4D_OK;
    Lanes ln;
    if ((rc = ln.fork(pl, st, !kernel_ms && fpix >= (size_t)4096 * 4096 && pl->chunk >= 2 && batch >= 2))) return rc;
    ln.sub = ln.two ? std::max(1, std::min(pl->chunk / 2, (batch + 1) / 2)) : pl->chunk;
    const size_t half = fpix / 2;
    int grp = 0;
    for (int b0 = 0; b0 < batch && rc == B4D_OK; b0 += ln.sub, ++grp) {
        const int nb = std::min(ln.sub, batch - b0);
        const size_t so = (size_t)ln.slot(grp) * ln.sub;
        hipStream_t ls = ln.stream(grp);
        float2* spec = pl->spec + half * so;
        float* nyq_rows = pl->nyq_rows + (size_t)pl->ny * so;
        float* gnyq = pl->gnyq + (size_t)pl->ny * so;
        if ((rc = mark())) break;
        if ((rc = dispatch_r2c(pl, frames + b0 * fpix, nb, ls, spec, nyq_rows))) break;
        if ((rc = mark())) break;
        ColArgs ca{};
        ca.spec = spec;
        ca.psd = psd ? psd + b0 * fpix : nullptr;
        ca.tw = pl->tw_y;
        ca.tw_inv = pl->tw_y;
        ca.psd_scale = psd_scale;
        ca.nx = pl->nx;
        ca.flags = flags;
        ca.half_rows = autocorr ? 1 : 0;   // the autocorrelation is even: K3 transforms rows 0..ny/2 only
#ifdef B4D_DIAG
        ca.diag = g_diag;
#endif
        if ((rc = col_psd_ac_pass(pl, ca, nb, ls))) break;
        NyqArgs na{};
        na.rows = nyq_rows;
        na.g_out = gnyq;
        na.psd = ca.psd;
        na.psd_scale = psd_scale;
        if ((rc = dispatch_nyq<NYQ_PSD_AC>(pl, na, nb, ls))) break;
        if ((rc = mark())) break;
        if (autocorr) {
            RowOutArgs ra{};
            ra.g = spec;
            ra.gnyq = gnyq;
            ra.out = autocorr + b0 * fpix;
            ra.peak = pl->peak + so;
            ra.tw = pl->tw_x;
            ra.scale = 1.0f / ((float)pl->nx * (float)pl->ny);
            ra.ny = pl->ny;
            ra.ct_w = pl->ct_w;
            ra.flags = flags;
            ra.half = 1;
            if ((rc = row_out_pass(pl, ra, nb, ls, kernel_ms ? &ev : nullptr))) break;
        }
        if ((rc = mark())) break;
    }
    {
        const int rj = ln.close();
        if (rc == B4D_OK) rc = rj;
    }
    if (kernel_ms && rc == B4D_OK && !ev.empty()) {
        hipError_t e = hipEventSynchronize(ev.back());
        if (e != hipSuccess) rc = fail(B4D_EHIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e));
        // per chunk the marks are: t0 | r2c | t1 | col | t2 | [peak | t3] | c2r | t4   (t3 only with NORM_PEAK + autocorr)
        const bool has_peak = autocorr && (flags & B4D_NORM_PEAK);
        const size_t per = 4 + (has_peak ? 1 : 0);
        for (size_t i = 0; rc == B4D_OK && i + per <= ev.size(); i += per) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            kernel_ms[0] += ms;
            (void)hipEventElapsedTime(&ms, ev[i + 1], ev[i + 2]);
            kernel_ms[1] += ms;
            if (has_peak) {
                (void)hipEventElapsedTime(&ms, ev[i + 2], ev[i + 3]);
                kernel_ms[2] += ms;
                (void)hipEventElapsedTime(&ms, ev[i + 3], ev[i + 4]);
                kernel_ms[3] += ms;
            } else {
                (void)hipEventElapsedTime(&ms, ev[i + 2], ev[i + 3]);
                kernel_ms[3] += ms;
            }
        }
    }
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    return rc;
}

int b4d_psd_autocorr2d(b4d_plan* pl, const float* frames, int batch, float* psd, float psd_scale, float* autocorr,
                       unsigned flags, void* stream) {
    return psd_autocorr_impl(pl, frames, batch, psd, psd_scale, autocorr, flags, (hipStream_t)stream, nullptr);
}

int b4d_psd_autocorr2d_timed(b4d_plan* pl, const float* frames, int batch, float* psd, float psd_scale,
                             float* autocorr, unsigned flags, void* stream, float* kernel_ms) {
    if (!kernel_ms) return fail(B4D_EINVAL, "kernel_ms is null");
    return psd_autocorr_impl(pl, frames, batch, psd, psd_scale, autocorr, flags, (hipStream_t)stream, kernel_ms);
}

// Workspace placement (include/b4d.h).  The plan's half-spectrum workspace is exchanged for up to `candidates` - 1 fresh
// allocations, each timed on the caller's own call; the fastest stays.
int b4d_plan_tune(b4d_plan* pl, const float* frames, int batch, float* psd, float psd_scale, float* autocorr, unsigned flags,
                  int candidates, float* best_ms, float* worst_ms, void* stream) {
    if (!pl) return fail(B4D_EINVAL, "plan is null");
    B4D_PLAN_LOCK(pl);
    if (best_ms) *best_ms = 0.f;
    if (worst_ms) *worst_ms = 0.f;
    if (pl->general || !pl->spec) return B4D_OK;   // power-of-two plans only: the other engines size their buffers per call
    if (candidates < 2) return fail(B4D_EINVAL, "candidates must be >= 2");
    candidates = std::min(candidates, 8);
    hipStream_t st = (hipStream_t)stream;
    std::vector<float2*> cand{pl->spec};
    std::vector<float> ms;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    B4D_HIP(hipEventCreate(&e0));
    B4D_HIP(hipEventCreate(&e1));
    int rc = B4D_OK;
    for (int i = 0; i < candidates && rc == B4D_OK; ++i) {
        if (i > 0) {   // an allocation failure only ends the search
            float2* q = nullptr;
            if (hipMalloc((void**)&q, pl->ws_bytes) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            cand.push_back(q);
            pl->spec = q;
        }
        float t = 0.f;
        for (int rep = 0; rep < 3 && rc == B4D_OK; ++rep) {   // first pass untimed
            if (rep == 1 && hipEventRecord(e0, st) != hipSuccess) rc = fail(B4D_EHIP, "hipEventRecord");
            if (rc == B4D_OK) rc = psd_autocorr_impl(pl, frames, batch, psd, psd_scale, autocorr, flags, st, nullptr);
        }
        if (rc == B4D_OK && (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
                             hipEventElapsedTime(&t, e0, e1) != hipSuccess))
            rc = fail(B4D_EHIP, "timing a candidate workspace failed");
        ms.push_back(0.5f * t);
    }
    (void)hipStreamSynchronize(st);
    size_t best = 0, worst = 0;
    if (rc == B4D_OK)
        for (size_t i = 1; i < ms.size(); ++i) {
            if (ms[i] < ms[best]) best = i;
            if (ms[i] > ms[worst]) worst = i;
        }
    for (size_t i = 0; i < cand.size(); ++i)
        if (i != best) (void)hipFree(cand[i]);
    pl->spec = cand[best];
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != B4D_OK) return rc;
    if (best_ms) *best_ms = ms[best];
    if (worst_ms) *worst_ms = ms[worst];
    // the kept workspace holds whatever its own timing passes left: the outputs are those of a normal call only if it ran last
    if (best + 1 != cand.size()) rc = psd_autocorr_impl(pl, frames, batch, psd, psd_scale, autocorr, flags, st, nullptr);
    return rc;
}

int b4d_psd2d(b4d_plan* pl, const float* frames, int batch, float* psd, float scale, void* stream) {
    if (!psd) return fail(B4D_EINVAL, "psd is null");
    return b4d_psd_autocorr2d(pl, frames, batch, psd, scale, nullptr, 0u, stream);
}

int b4d_autocorr2d(b4d_plan* pl, const float* frames, int batch, float* autocorr, unsigned flags, void* stream) {
    if (!autocorr) return fail(B4D_EINVAL, "autocorr is null");
    return b4d_psd_autocorr2d(pl, frames, batch, nullptr, 1.0f, autocorr, flags, stream);
}

int b4d_fft2d(b4d_plan* pl, const float* frames, int batch, float* out_c64, void* stream) {
    if (!pl || !frames || !out_c64) return fail(B4D_EINVAL, "null argument");
    if (batch < 1) return fail(B4D_EINVAL, "batch must be >= 1");
    B4D_PLAN_LOCK(pl);
    hipStream_t st = (hipStream_t)stream;
    if (pl->general) return general_fft2d(pl, frames, batch, reinterpret_cast<float2*>(out_c64), st);
    // Columns first, rows last (b4d_spectrum.hip): every store of the full spectrum is a whole line.  The half spectrum between the
    // two passes (4 B / pixel, written and read once) stays in the memory-side cache: two-lane launch groups (Lanes); a 4096^2
    // frame is a group of its own and stays on one lane (12.0 -> 13.8 k frames/s; 13.1 k on two).
    const size_t fpix = (size_t)pl->ny * pl->nx, half = fpix / 2;
    Lanes ln;
    int rc = ln.open(pl, st, batch, half * sizeof(float2), fpix * sizeof(float), fpix < (size_t)4096 * 4096, (size_t)32 << 20);
    if (!ln.two && fpix == (size_t)4096 * 4096) ln.sub = 1;
    if (rc) return rc;
    int g = 0;
    for (int b0 = 0; b0 < batch && rc == B4D_OK; b0 += ln.sub, ++g)
        rc = spectrum_rows_last(pl, pl->spec + (size_t)ln.slot(g) * ln.sub * half, frames + b0 * fpix, std::min(ln.sub, batch - b0),
                                reinterpret_cast<float2*>(out_c64) + b0 * fpix, ln.stream(g));
    const int rj = ln.close();
    return rc ? rc : rj;
}

int b4d_fft2d_c2c(b4d_plan* pl, const float* in_c64, int batch, int inverse, float* out_c64, void* stream) {
    if (!pl || !in_c64 || !out_c64) return fail(B4D_EINVAL, "null argument");
    if (batch < 1) return fail(B4D_EINVAL, "batch must be >= 1");
    if (!pl->general) return fail(B4D_EINVAL, "b4d_fft2d_c2c needs a plan from b4d_plan_create_general");
    B4D_PLAN_LOCK(pl);
    return general_fft2d_c2c(pl, reinterpret_cast<const float2*>(in_c64), batch, inverse, reinterpret_cast<float2*>(out_c64),
                             (hipStream_t)stream);
}

}  // extern "C"
