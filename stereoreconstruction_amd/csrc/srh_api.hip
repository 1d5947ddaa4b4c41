// srh_api.hip -- the C-ABI of include/stereo_recon_hip.h: context, view upload,
// run entry points, measurement.  Host code only; every numeric result comes from
// the kernels in srh_kernels.hip / srh_dense.hip.  There is no CPU fallback: without
// a HIP device srh_create fails with SRH_E_NO_DEVICE.
#include "srh_internal.hpp"
#include "srh_geom.hpp"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <limits>
#include <vector>

using namespace srh;

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
	return code;
}

#define HIP_TRY(expr)                                                                   \
	do {                                                                                \
		hipError_t e_ = (expr);                                                         \
		if (e_ != hipSuccess)                                                           \
			return fail(SRH_E_DEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

// ------------------------------------------------------------------ context
struct ViewHost {
	int w = 0, h = 0;
	bool present = false;
	uint32_t *rgba = nullptr;
	uint8_t  *mask = nullptr;
	double   *gray = nullptr, *gray_tv = nullptr, *depth = nullptr;
	double   *edges = nullptr;     // 4 planes of neighbour colour distances (geodesic windows)
	uint8_t  *full = nullptr;      // 1 where the whole (2*full_r+1)^2 TwoView window is usable
	int       full_r = 0;
	// strip kernel (srh_strip.hip): NaN-bordered copy of gray_tv, zero-bordered "window fully usable" plane for radius fullp_r
	double   *tvp = nullptr;   bool tvp_valid = false;
	double   *geo5 = nullptr;  bool geo5_valid = false;   // edge + tap planes with their borders written out (geodesic_dma_kernel)
	bool      geo5_denied = false;                        // released by an out-of-memory retry: this upload keeps the register-staged windows kernel
	uint8_t  *fullp = nullptr; int fullp_r = 0;
	// how the candidate lists of this view against slot j are best evaluated, learnt from the last run:
	// 0 unknown, 1 row runs (srh_rows.hip), 2 list order (srh_list.hip: steep curves)
	uint8_t   list_mode[SRH_MAX_VIEWS] = {0};
	// MultiViewStereo list path: the masked-in pixels (y*w + x) row by row, even rows left to right, odd rows right to
	// left, so that consecutive entries are neighbours in the image also across a row change; act_row[y] = first entry of
	// row y (h + 1 values).  A wave of the walk / cost kernels takes 64 consecutive entries.
	uint32_t *act = nullptr; size_t act_cap = 0;
	std::vector<uint32_t> act_row, act_host;
	std::vector<uint8_t> hmask;    // host copy of the mask bytes (empty: no mask, every pixel counts): `act` is built from it
	bool act_valid = false;        //   when the MultiViewStereo list path first asks (ensure_act); TwoView callers never pay for it
	srh_camera cam;
	// MRF branch over several views (srh_mvs_initial_estimate_peaks / srh_mvs_mrf_estimate_views)
	double   *peaks = nullptr; size_t peaks_cap = 0; int peaks_k = 0;   // top-K peaks of the last initial estimate
	double   *mrf = nullptr;   size_t mrf_cap = 0;                      // this view's own TRW-S scratch
};

struct ProfEntry { double ms = 0; int64_t n = 0; };
struct PendingEvt { std::string name; hipEvent_t a, b; };

#ifndef SRH_MVS_SLOTS
#define SRH_MVS_SLOTS 2
#endif
// one MultiViewStereo estimate in flight (srh_mvs_initial_estimate, "two estimates in flight")
struct MvsSlot {
	hipStream_t stream = nullptr;
	hipEvent_t ev = nullptr, done = nullptr;
	int *h_maxc = nullptr;                              // pinned: longest list of the queued pass
	bool own_buffers = false;                           // slot 1: its own band buffers, swapped into the context for its launches
	double *wbuf = nullptr; size_t wbuf_cap = 0;
	double *cost = nullptr; size_t cost_cap = 0;
	double *tnum = nullptr; size_t tnum_cap = 0;
	int32_t *lcount = nullptr; size_t lcount_cap = 0;
	uint32_t *lcand = nullptr; size_t lcand_cap = 0;
	uint32_t *mvs_wdesc = nullptr; size_t mvs_wdesc_cap = 0;
	int32_t *mvs_nwin = nullptr; size_t mvs_nwin_cap = 0;
	Counters *d_cnt = nullptr; int *d_span = nullptr;
	bool pending = false;                               // kernels queued, capacity check outstanding
	int view = -1, nneigh = 0, y0 = 0, y1 = 0, cmax = 0;
	int32_t neigh[SRH_MAX_NEIGH] = {0};
	srh_params p;
};

struct srh_context {
	int device = 0;
	hipStream_t own_stream = nullptr;
	hipStream_t stream = nullptr;
	ViewHost views[SRH_MAX_VIEWS];
	ViewDev *d_views = nullptr;
	int32_t *d_slots = nullptr;
	int32_t slots_host[SRH_MAX_VIEWS] = {0}; int slots_n = 0;   // what d_slots holds (srh_mvs_cross_check: the same list for every view of a run)
	Counters *d_cnt = nullptr;
	int *d_span = nullptr;
	double *wbuf = nullptr;   size_t wbuf_cap = 0;      // doubles
	double *cost = nullptr;   size_t cost_cap = 0;      // doubles
	double *tnum = nullptr;   size_t tnum_cap = 0;      // per-label table of the pinhole walk
	hipStream_t mrf_stream[SRH_MAX_VIEWS] = { nullptr };            // one stream per view of srh_mvs_mrf_estimate_views
	void *mrf_host = nullptr;                                       // pinned: per view {energy, pad, status[4]}
	double *mrf_peaks = nullptr; size_t mrf_peaks_cap = 0;   // top-K peaks of srh_mvs_initial_estimate_mrf
	double *mrf = nullptr;    size_t mrf_cap = 0;       // MRF stage scratch (srh_mrf.hip); mrf_w/h/k: what the last run left in it
	int mrf_w = 0, mrf_h = 0, mrf_k = 0;
	double *pconst = nullptr; size_t pconst_cap = 0;    // per-pixel constants of the dense kernel's fast form (4 doubles per pixel of a band)
	PixRange *prange = nullptr; size_t prange_cap = 0;  // per-pixel candidate column range of a band (strip kernel, scan)
	uint32_t *cflag = nullptr; size_t cflag_cap = 0;    // certified arithmetic: [count | band pixels whose decisions the bound does not cover]
	uint8_t *stpl = nullptr; size_t stpl_cap = 0;       // template scan: the pass's candidate template (twoview_template_kernel)
	uint32_t *tileflag = nullptr; size_t tileflag_cap = 0;  //   and the tiles it leaves to twoview_scan_kernel: [count | tile indices]
	int geodma = 1;                                     // option "geodma": 1 = the dense path's r = 5 geodesic windows by the persistent LDS-DMA kernel (default), 0 = geodesic_reg_kernel
	int tscan = 1;                                      // option "tscan": 1 = template scan on the dense path (default), 0 = every tile through twoview_scan_kernel
	int strip = 1;                                      // option "strip": 1 = persistent strip cost kernel (default), 0 = one workgroup per tile, 4 / 8 = force the 4- / 8-wave form
	int num_cus = 256;
	int32_t *lcount = nullptr; size_t lcount_cap = 0;   // candidate-list path: candidates per pixel
	uint32_t *lcand = nullptr; size_t lcand_cap = 0;    //   candidate pixels (cx | cy<<16)
	bool force_walk = false;                            // option "force_generic" = 2: never use the list path either
	int list_count_pass = 0;                            // option "list_count_pass": size a pair's first lists by a counting pass (round 5) instead of the host's guess
	int list_cmax_hint = 0;                             // longest candidate list seen so far (list-path capacity)
	int list_smax_hint = 0;                             // most cost slots a pixel needed so far (run-blocked lists)
	int mvs_cmax_hint = 0;                              // longest MultiViewStereo candidate list seen so far
	// side stream of the row-run list path: the support windows are computed while the list kernel's last waves drain
	hipStream_t side_stream = nullptr;
	hipEvent_t side_go = nullptr, side_done = nullptr;
	int mvs_async = 1;                                  // option "mvs_async": srh_mvs_initial_estimate queues a view on one of two side streams and returns (default); 0 = waits for each view
	MvsSlot mvs_slot[SRH_MVS_SLOTS];
	int mvs_turn = 0, mvs_last = -1;
	bool in_settle = false;
	int mvs_staged = 1;                                 // option "mvs_staged": the list cost kernel takes its windows from LDS copies of the other view where they fit (default), 0 = gathers only
	uint32_t *mvs_wdesc = nullptr; size_t mvs_wdesc_cap = 0;   // window descriptors of the walk kernel's waves
	int32_t *mvs_nwin = nullptr; size_t mvs_nwin_cap = 0;
	bool list_rows = true;                              // option "list_rows": evaluate lists in row runs (srh_rows.hip)
	uint32_t *lrowinfo = nullptr; size_t lrowinfo_cap = 0;
	int32_t *lmeta = nullptr; size_t lmeta_cap = 0;
	void *comm = nullptr; int comm_ranks = 0, comm_rank = 0;   // RCCL communicator (srh_comm_init)
	// bytes per band of scratch (windows, cost rows, candidate lists) the caller ASKS for: every band ends in the tails of
	// its kernels, so the largest configuration should take one band where the device has the room (MI355X: 288 GB).
	// What a run uses is band_budget(): this, capped by a share of the memory that is free right now, halved after an
	// allocation failure (budget_cap).
	size_t wbuf_budget = (size_t)32768 << 20;
	size_t budget_cap = 0;                              // 0 = none; set by with_thinner_bands after an out-of-memory run
	size_t budget_used = 0;                             // what band_budget() returned last
	size_t mem_limit = 0;                               // option "mem_limit_mb": pretend the device has only this much free (tests)
	int debug_trace = 0;                                // option "debug_trace": the list path's capacity decisions on stderr (diagnostics)
	size_t alloc_limit = 0;                             // option "debug_alloc_limit_mb": band buffers above this size are refused (tests)
	const volatile int *cancel = nullptr;
	srh_progress_fn progress = nullptr;
	void *user = nullptr;
	bool profiling = false;
	bool force_generic = false;
 	bool use_fused = false;                             // option "fused": single fused kernel for row-aligned pairs
	// option "arith": 0 = the reference's arithmetic everywhere; 3 (default) = CERTIFIED: fused multiply-adds in the strip
	// kernel's cost loops, every decision checked against an error bound, uncovered pixels redone in the reference's
	// arithmetic -- the reference's bits at the fused speed; 1 = fused multiply-adds unchecked; 2 = packed single precision
	int arith = 3;
	// srh_twoview_compute queues both passes and the cross-check and verifies the passes' counters with ONE wait at the end
	// (the plans are refuted once in a blue moon; a wait per pass leaves the GPU idle while the host launches the next one)
	// (lists: the pass took the row-run candidate lists with the capacities learnt from earlier runs -- span = the kernels'
	// maxima: longest list, most cost slots, a curve over too many rows -- to be compared with cmax / smax)
	// (guessed: the capacities were the host's first guess, estimate_list_capacity -- a pass that stands then teaches the
	// context what it measured, for reference view `ref` against `oth`)
	struct TvDefer { Counters *host = nullptr; int *span = nullptr; bool queued = false, strip = false, cert = false, lists = false, guessed = false;
	                 int cmax = 0, smax = 0, ref = -1, oth = -1; };
	TvDefer tv_defer[2];
	TvDefer *defer = nullptr;
	// ... and runs the second pass on a stream of its own with its own band buffers (swapped into the context for its
	// launches, tv_slot_swap): the two passes share nothing but the views, and one pass alone leaves the device idle between
	// its kernels (~10 us each, a dozen per pass), in their tails and -- the weights kernels, one wave per SIMD -- inside them
	struct TvSlot {
		hipStream_t stream = nullptr;
		hipEvent_t go = nullptr, done = nullptr;
		Counters *d_cnt = nullptr; int *d_span = nullptr;
		double *wbuf = nullptr, *cost = nullptr, *tnum = nullptr, *pconst = nullptr;
		PixRange *prange = nullptr; uint32_t *cflag = nullptr, *lcand = nullptr, *lrowinfo = nullptr; int32_t *lcount = nullptr, *lmeta = nullptr;
		uint8_t *stpl = nullptr; uint32_t *tileflag = nullptr;
		size_t wbuf_cap = 0, cost_cap = 0, tnum_cap = 0, pconst_cap = 0, prange_cap = 0, cflag_cap = 0, lcand_cap = 0, lrowinfo_cap = 0,
		       lcount_cap = 0, lmeta_cap = 0, stpl_cap = 0, tileflag_cap = 0;
	} tv_slot;
	int tv_overlap = 1;                                 // option "tv_overlap": 0 = both passes on the context's stream, one after the other
	int f32_form = 0;                                   // option "f32_form" (f32 mode only, not a parity mode): 1 = the one-pass sums in single precision (priced in DESIGN.md 9.0'' (i))
	int rows_masked = 1;                                // option "rows_masked": the certified row-run cost kernel's masked blocks + single candidates: 1 = when at least 90 % of the other view's usable pixels have a fully usable window (decided on the device), 2 = always, 0 = never (a block is fast only when all 8 of its candidates are: round 5's rule)
	int side_weights = 1;                               // option "side_weights": 0 = the row-run path computes its support windows on the pass's own stream, behind the list kernel (profiling: every kernel's own duration)
	int cert_form = 1;                                  // option "cert_form": certified strip kernel in 1 = the one-pass form (default), 2 = two fused sweeps
	bool force_dense = false;                           // option "force_dense": propose the dense plan for any pinhole pair
	// srh_twoview_cost_rows (diagnostic): the dense plan stops after the cost kernel of its one band and hands the rows out
	struct Diag { int form = 0; bool raw = false; double *cost = nullptr; size_t cost_doubles = 0; int32_t *range = nullptr;
	              int cstride = 0, rows = 0; bool done = false, strip = false; } *diag = nullptr;
	std::map<std::string, ProfEntry> prof;
	std::vector<PendingEvt> pending;
	srh_stats stats;
	bool last_fused = false;                            // the last TwoView pass ran the fused kernel
};

static bool cancelled(srh_context *c) { return c->cancel && *c->cancel; }
static void progress(srh_context *c, int step, const char *stage) { if (c->progress) c->progress(step, stage, c->user); }

static int drain_profile(srh_context *c) {
	for (auto &p : c->pending) {
		float ms = 0;
		HIP_TRY(hipEventSynchronize(p.b));
		HIP_TRY(hipEventElapsedTime(&ms, p.a, p.b));
		ProfEntry &e = c->prof[p.name];
		e.ms += ms; e.n += 1;
		hipEventDestroy(p.a); hipEventDestroy(p.b);
	}
	c->pending.clear();
	return SRH_OK;
}

// bracket one kernel launch with events when profiling
struct Scope {
	srh_context *c; PendingEvt p; bool on;
	Scope(srh_context *c_, const char *name) : c(c_), on(c_->profiling) {
		if (on) {
			p.name = name;
			hipEventCreate(&p.a); hipEventCreate(&p.b);
			hipEventRecord(p.a, c->stream);
		}
	}
	~Scope() {
		if (on) {
			hipEventRecord(p.b, c->stream);
			c->pending.push_back(p);
			if (c->pending.size() > 4096) drain_profile(c);
		}
	}
};

// Band buffers grow on demand.  A request the device cannot serve is not the end of the call: g_oom tells the entry
// point (with_thinner_bands below) to release the band buffers, halve the band budget and run again.
// g_alloc_limit (option "debug_alloc_limit_mb") makes requests above a size fail the same way: the test of that path.
// (both are set from the context at the start of every retrying entry point -- with_thinner_bands -- so that a limit set
// through one thread applies to calls made on another)
static thread_local bool g_oom = false;
static thread_local size_t g_alloc_limit = 0;

// ---- the process's pool of band buffers ------------------------------------------------------------------------------
// A TwoViewStereo / MultiViewStereo object of the reference lives for one computation (twoviewstereo.cpp:150-227; the GUI
// makes a new one per run), so a drop-in creates and destroys contexts again and again in one process -- and a context's
// band buffers are gigabytes.  Handing them back to the driver at srh_destroy and asking for them again at the next
// object's first call is not free on this stack: measured (profiles/r06_first_call.txt, HIP API trace) ONE hipMalloc of 11 GB
// now and then takes 0.9 ... 1.5 s when tens of GB have just been freed by the same process (its siblings: 0.3 ms).  So band
// buffers released by srh_destroy (and by a growing buffer) go to a process-wide pool, per device, and the next context's
// requests are served from it (best fit, at most 1.5x the request); what does not fit the pool's cap is freed.  The pool
// is emptied before an allocation is declared out of memory, and counts as available memory for the band planner.
// Option "band_pool_mb" (any context; process-wide): the cap, default 65 536; 0 empties and disables the pool.
struct PoolBlock { void *p; size_t bytes; int device; };
static std::mutex g_pool_mu;
static std::vector<PoolBlock> g_pool;
static size_t g_pool_bytes = 0, g_pool_cap = (size_t)65536 << 20;
static const size_t POOL_MIN_BYTES = (size_t)4 << 20;                // smaller buffers are not worth a pool entry

static void pool_flush(int device) {                                 // (device < 0: every device)
	std::vector<PoolBlock> out;
	{
		std::lock_guard<std::mutex> g(g_pool_mu);
		for (size_t k = 0; k < g_pool.size();)
			if (device < 0 || g_pool[k].device == device) { out.push_back(g_pool[k]); g_pool_bytes -= g_pool[k].bytes; g_pool[k] = g_pool.back(); g_pool.pop_back(); }
			else ++k;
	}
	int cur = 0;
	const bool have = hipGetDevice(&cur) == hipSuccess;
	for (const PoolBlock &b : out) { (void)hipSetDevice(b.device); (void)hipFree(b.p); }
	if (have) (void)hipSetDevice(cur);
}
static size_t pool_bytes_on(int device) {
	std::lock_guard<std::mutex> g(g_pool_mu);
	size_t b = 0;
	for (const PoolBlock &k : g_pool) if (k.device == device) b += k.bytes;
	return b;
}
// a released band buffer of `bytes` on the current device: into the pool, or back to the driver
static void pool_give(void *p, size_t bytes) {
	if (!p) return;
	int dev = 0;
	if (bytes >= POOL_MIN_BYTES && hipGetDevice(&dev) == hipSuccess) {
		std::lock_guard<std::mutex> g(g_pool_mu);
		if (g_pool_bytes + bytes <= g_pool_cap) { g_pool.push_back({p, bytes, dev}); g_pool_bytes += bytes; return; }
	}
	(void)hipFree(p);
}
// a block of at least `bytes` (at most 1.5x) from the pool, or null; *got = its size
static void *pool_take(size_t bytes, size_t *got) {
	int dev = 0;
	if (bytes < POOL_MIN_BYTES || hipGetDevice(&dev) != hipSuccess) return nullptr;
	std::lock_guard<std::mutex> g(g_pool_mu);
	size_t best = g_pool.size();
	for (size_t k = 0; k < g_pool.size(); ++k)
		if (g_pool[k].device == dev && g_pool[k].bytes >= bytes && g_pool[k].bytes <= bytes + bytes/2 &&
		    (best == g_pool.size() || g_pool[k].bytes < g_pool[best].bytes)) best = k;
	if (best == g_pool.size()) return nullptr;
	void *p = g_pool[best].p;
	*got = g_pool[best].bytes;
	g_pool_bytes -= g_pool[best].bytes;
	g_pool[best] = g_pool.back(); g_pool.pop_back();
	return p;
}

template <class T>
static int ensure(T *&ptr, size_t &cap, size_t need) {
	if (cap >= need) return SRH_OK;
	// (a buffer that is being outgrown: hipFree used to wait for the device before it let go of it; the pool must not hand it to
	// another context while a kernel queued earlier still reads it -- growth is rare, the wait is kept)
	if (ptr) { (void)hipDeviceSynchronize(); pool_give(ptr, cap*sizeof(T)); ptr = nullptr; cap = 0; }
	if (g_alloc_limit && need*sizeof(T) > g_alloc_limit) {
		g_oom = true;
		return fail(SRH_E_DEVICE, "band buffer of %zu bytes refused (debug_alloc_limit_mb)", need*sizeof(T));
	}
	size_t got = 0;
	if (void *q = pool_take(need*sizeof(T), &got)) { ptr = (T *)q; cap = got/sizeof(T); return SRH_OK; }
	hipError_t e = hipMalloc((void **)&ptr, need*sizeof(T));
	if (e == hipErrorOutOfMemory) {
		// the pool's blocks are memory too: give them back and ask once more before the bands get thinner
		(void)hipGetLastError();
		int dev = 0;
		if (hipGetDevice(&dev) == hipSuccess && pool_bytes_on(dev)) { pool_flush(dev); e = hipMalloc((void **)&ptr, need*sizeof(T)); }
	}
	if (e != hipSuccess) {
		ptr = nullptr;
		(void)hipGetLastError();
		if (e == hipErrorOutOfMemory) g_oom = true;
		return fail(SRH_E_DEVICE, "hipMalloc of a %zu-byte band buffer: %s", need*sizeof(T), hipGetErrorString(e));
	}
	cap = need;
	return SRH_OK;
}

static int mvs_settle_all(srh_context *c);
// settle: finish the MultiViewStereo estimates in flight first (every entry point but srh_mvs_initial_estimate itself)
static int check_slot(srh_context *c, int slot, bool must_exist, bool settle = true) {
	if (!c) return fail(SRH_E_INVALID, "null context");
	if (settle) { const int rc = mvs_settle_all(c); if (rc) return rc; }
	if (slot < 0 || slot >= SRH_MAX_VIEWS) return fail(SRH_E_INVALID, "view slot %d out of range [0,%d)", slot, SRH_MAX_VIEWS);
	if (must_exist && !c->views[slot].present) return fail(SRH_E_INVALID, "view slot %d has no image", slot);
	return SRH_OK;
}

static int check_params(const srh_params *p) {
	if (!p) return fail(SRH_E_INVALID, "null params");
	if (p->num_depth_levels < 2) return fail(SRH_E_INVALID, "num_depth_levels %d < 2", p->num_depth_levels);
	if (p->window_radius < 1 || p->window_radius > 15) return fail(SRH_E_INVALID, "window_radius %d outside [1,15]", p->window_radius);
	if (!(p->image_scale > 0)) return fail(SRH_E_INVALID, "image_scale must be > 0");
	if (p->weight_kind != SRH_WEIGHT_ADAPTIVE && p->weight_kind != SRH_WEIGHT_GEODESIC)
		return fail(SRH_E_INVALID, "weight_kind %d unknown", p->weight_kind);
	return SRH_OK;
}

// ------------------------------------------------------------------ library
// srh_mvs_mrf_estimate_views keeps one stream per view busy; the HIP runtime multiplexes streams onto 4 hardware
// queues unless the process asks for more BEFORE HIP initialises (GPU_MAX_HW_QUEUES=16; measured: 8 views 277 ms
// with 4 queues, 212 with 8, 151 with 16).  The library never touches the environment: the host application sets
// the variable (bench.py and profiles/mrf_views.py do), srh_hw_queues_requested() reports what applies.
extern "C" int srh_hw_queues_requested(void) {
	const char *s = getenv("GPU_MAX_HW_QUEUES");
	const int n = s ? atoi(s) : 0;
	return n > 0 ? n : 4;                                       // the HIP runtime's default
}

extern "C" int srh_abi_version(void) { return SRH_ABI_VERSION; }
extern "C" const char *srh_build_id(void) {
	return
#include "build_id.inc"
	;
}
extern "C" const char *srh_last_error(void) { return g_err; }

extern "C" int srh_device_count(int *count) {
	if (!count) return fail(SRH_E_INVALID, "null count");
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
	*count = n;
	return SRH_OK;
}

// ------------------------------------------------------------------ params / cameras (host math)
extern "C" void srh_params_twoview_defaults(srh_params *p) {
	memset(p, 0, sizeof(*p));
	p->min_depth = 10; p->max_depth = 100; p->num_depth_levels = 100;   // gui/forms/stereowidget.ui:199-288
	p->window_radius = 5;                 // twoviewstereo.cpp:66
	p->image_scale = 1.0;
	p->weight_kind = SRH_WEIGHT_GEODESIC; // twoviewstereo.cpp:84
	p->geodesic_iters = 3; p->geodesic_sigma = 50.0; p->geodesic_init = 1000000.0;   // geodesicweight.cpp:33-36,68
	p->adaptive_color_sigma = 10.0;       // adaptiveweight.cpp:26
	p->weight_cutoff = 1e-10;
	p->bad_ret = 1000; p->max_color_diff = 120;            // twoviewstereo.cpp:65,74
	p->second_best_factor = 0.95; p->wta_margin = 1e-10; p->inconsistency_thresh = 1;   // :78-79,293
	p->peak_threshold = 0.95;             // multiviewstereo.cpp:589
	p->cross_check_threshold = 1.0;
	p->neighbour_min_dot = 0.2;           // multiviewstereo.cpp:343
	p->top_k = 9; p->num_neighbours = 3;  // multiviewstereo.cpp:94,97
}

extern "C" void srh_params_mvs_defaults(srh_params *p) {
	srh_params_twoview_defaults(p);
	p->window_radius = 2;                 // multiviewstereo.cpp:91
}

// The constants of the certified arithmetic's error bound for these parameters (srh_internal.hpp, CertBound; DESIGN.md
// 2b): host arithmetic only -- what tests/test_cert_bound.py checks against an exact replay of the three arithmetics.
extern "C" int srh_cert_bound(const srh_params *p, int mvs, srh_cert_info *out) {
	if (!p || !out) return fail(SRH_E_INVALID, "null argument");
	if (p->window_radius < 1 || p->window_radius > 15) return fail(SRH_E_INVALID, "window_radius %d outside [1,15]", p->window_radius);
	const CertBound c = cert_bound(*p, mvs != 0);
	out->e0 = c.e0; out->k1 = c.k1; out->k2 = c.k2; out->k3 = c.e0 - c.room; out->zmax2 = c.zmax2; out->m_hi = c.m_hi;
	out->ok = c.ok; out->taps = (2*p->window_radius + 1)*(2*p->window_radius + 1);
	return SRH_OK;
}

extern "C" double srh_cert_sigma3(const srh_params *p, int mvs, double sum2) {
	if (!p || p->window_radius < 1 || p->window_radius > 15) return __builtin_nan("");
	return cert_bound(*p, mvs != 0).sigma3(sum2);
}

static bool near_zero(double x) { return (x <= 1e-10 && x >= -1e-10); }   // camera.cpp:51-52

extern "C" int srh_camera_from_krt(const double K[9], const double Rin[9], const double t[3],
                                   const double dist[5], const double plane_normal[3],
                                   double plane_dist, double refr_index, srh_camera *out)
{
	if (!K || !Rin || !t || !out) return fail(SRH_E_INVALID, "null argument");
	srh_camera c;
	memset(&c, 0, sizeof(c));
	memcpy(c.K, K, sizeof(c.K));
	memcpy(c.R, Rin, sizeof(c.R));
	memcpy(c.t, t, sizeof(c.t));
	// orthonormalize(R_): Gram-Schmidt over columns, then flush |v| < 1e-10 to 0 (camera.cpp:140-160)
	for (int i = 0; i < 3; ++i) {
		Vec3 accum = v3(0, 0, 0);
		for (int j = 0; j < i; ++j) {
			const Vec3 vi = v3(c.R[i], c.R[3 + i], c.R[6 + i]);
			const Vec3 vj = v3(c.R[j], c.R[3 + j], c.R[6 + j]);
			const double scale = dot(vi, vj) / dot(vj, vj);
			accum = accum + vj*scale;
		}
		const Vec3 col = normalized(v3(c.R[i], c.R[3 + i], c.R[6 + i]) - accum);
		c.R[i] = col.x; c.R[3 + i] = col.y; c.R[6 + i] = col.z;
	}
	for (int k = 0; k < 9; ++k) if (-1e-10 < c.R[k] && c.R[k] < 1e-10) c.R[k] = 0.0;
	{   // Kinv_ = K_.inverse() (3x3 cofactor inverse)
		const double *m = c.K;
		const double c00 = m[4]*m[8] - m[5]*m[7];
		const double c01 = m[5]*m[6] - m[3]*m[8];
		const double c02 = m[3]*m[7] - m[4]*m[6];
		const double det = (m[0]*c00 + m[1]*c01) + m[2]*c02;
		const double invdet = 1.0 / det;
		c.Kinv[0] = c00*invdet;
		c.Kinv[1] = (m[2]*m[7] - m[1]*m[8])*invdet;
		c.Kinv[2] = (m[1]*m[5] - m[2]*m[4])*invdet;
		c.Kinv[3] = c01*invdet;
		c.Kinv[4] = (m[0]*m[8] - m[2]*m[6])*invdet;
		c.Kinv[5] = (m[2]*m[3] - m[0]*m[5])*invdet;
		c.Kinv[6] = c02*invdet;
		c.Kinv[7] = (m[1]*m[6] - m[0]*m[7])*invdet;
		c.Kinv[8] = (m[0]*m[4] - m[1]*m[3])*invdet;
	}
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c.Rinv[i*3 + j] = c.R[j*3 + i];
	{ const Vec3 C = matvec(c.Rinv, v3(-t[0], -t[1], -t[2])); c.C[0] = C.x; c.C[1] = C.y; c.C[2] = C.z; }
	{   // updatePrincipleRay (camera.cpp:292-298)
		const Vec3 tc = v3(c.K[2]/c.K[8], c.K[5]/c.K[8], c.K[8]/c.K[8]);
		const Vec3 dir = normalized(matvec(c.Kinv, tc));
		const Vec3 pd = normalized(matvec(c.Rinv, dir));
		c.pdir[0] = pd.x; c.pdir[1] = pd.y; c.pdir[2] = pd.z;
	}
	if (dist) {
		memcpy(c.dist, dist, sizeof(c.dist));
		c.is_distorted = !near_zero(dist[0]) || !near_zero(dist[1]) || !near_zero(dist[2])
		              || !near_zero(dist[3]) || !near_zero(dist[4]);
	}
	c.plane_normal[2] = 1.0; c.plane_dist = 0.0; c.refr_index = 1.0;
	if (plane_normal) {
		const Vec3 n = normalized(load3(plane_normal));       // Plane3d ctor, plane.hpp:32
		c.plane_normal[0] = n.x; c.plane_normal[1] = n.y; c.plane_normal[2] = n.z;
		c.plane_dist = plane_dist;
		c.refr_index = refr_index;
	}
	c.is_refractive = (!near_zero(c.refr_index - 1) && !near_zero(c.plane_dist));   // camera.cpp:326-344
	*out = c;
	return SRH_OK;
}

extern "C" int srh_camera_from_p(const double Pin[12], const double dist[5], const double plane_normal[3],
                                 double plane_dist, double refr_index, srh_camera *out)
{
	if (!Pin || !out) return fail(SRH_E_INVALID, "null argument");
	// Camera::updateOthers (camera.cpp:251-288)
	const double n2 = (Pin[8]*Pin[8] + Pin[9]*Pin[9]) + Pin[10]*Pin[10];
	if (!(n2 > 0)) return fail(SRH_E_INVALID, "projection matrix with a zero third row");
	double P[3][4];
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) P[i][j] = Pin[i*4 + j] / n2;
	// A = (reverseRows * M)^T, Householder QR (Eigen::HouseholderQR, unblocked): A = Q * Rt
	double A[3][3], tau[3];
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = P[2 - j][i];
	for (int k = 0; k < 3; ++k) {
		const double c0 = A[k][k];
		double tailSq = 0;
		for (int i = k + 1; i < 3; ++i) tailSq += A[i][k]*A[i][k];
		double beta;
		if (tailSq <= std::numeric_limits<double>::min()) {       // makeHouseholder: nothing to annihilate
			tau[k] = 0; beta = c0;
			for (int i = k + 1; i < 3; ++i) A[i][k] = 0;
		} else {
			beta = sqrt(c0*c0 + tailSq);
			if (c0 >= 0) beta = -beta;
			for (int i = k + 1; i < 3; ++i) A[i][k] /= (c0 - beta);   // essential part, stored below the diagonal
			tau[k] = (beta - c0)/beta;
		}
		A[k][k] = beta;
		for (int j = k + 1; j < 3; ++j) {                             // H_k applied to the remaining columns
			double tmp = 0;
			for (int i = k + 1; i < 3; ++i) tmp += A[i][k]*A[i][j];
			tmp += A[k][j];
			A[k][j] -= tau[k]*tmp;
			for (int i = k + 1; i < 3; ++i) A[i][j] -= tau[k]*A[i][k]*tmp;
		}
	}
	double Q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
	for (int k = 2; k >= 0; --k)                                     // householderQ(): H0 H1 H2
		for (int j = k; j < 3; ++j) {
			double tmp = 0;
			for (int i = k + 1; i < 3; ++i) tmp += A[i][k]*Q[i][j];
			tmp += Q[k][j];
			Q[k][j] -= tau[k]*tmp;
			for (int i = k + 1; i < 3; ++i) Q[i][j] -= tau[k]*A[i][k]*tmp;
		}
	double K[9], R[9];
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j) {
			R[i*3 + j] = Q[j][2 - i];                                  // reverseRows * Q^T
			K[i*3 + j] = (2 - j <= 2 - i) ? A[2 - j][2 - i] : 0.0;     // reverseRows * Rt^T * reverseRows
		}
	for (int axis = 2; axis >= 0; --axis) {                          // positive diagonal of K (camera.cpp:266-275)
		if (K[axis*3 + axis] < 0) {
			K[axis*3 + axis] = -K[axis*3 + axis];
			for (int j = 0; j < 3; ++j) R[axis*3 + j] = -R[axis*3 + j];
		}
		if (K[axis*3 + 2] < 0) K[axis*3 + 2] = -K[axis*3 + 2];
	}
	// t_ = Kinv_ * P_.col(3): Kinv is the cofactor inverse srh_camera_from_krt computes again below
	srh_camera tmpc;
	const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z3[3] = {0, 0, 0};
	int rc = srh_camera_from_krt(K, I3, z3, nullptr, nullptr, 0.0, 1.0, &tmpc);
	if (rc) return rc;
	const Vec3 t = matvec(tmpc.Kinv, v3(P[0][3], P[1][3], P[2][3]));
	const double tt[3] = { t.x, t.y, t.z };
	return srh_camera_from_krt(K, R, tt, dist, plane_normal, plane_dist, refr_index, out);
}

extern "C" int srh_mvs_neighbours(int nviews, const srh_camera *cams, const srh_params *p,
                                  int32_t *neigh, int32_t *count)
{
	if (nviews < 0 || !cams || !p || !neigh || !count) return fail(SRH_E_INVALID, "null argument");
	if (p->num_neighbours < 0) return fail(SRH_E_INVALID, "num_neighbours < 0");
	// multiviewstereo.cpp:335-360
	for (int v = 0; v < nviews; ++v) {
		std::vector<std::pair<double, int>> nearViews;
		for (int v2 = 0; v2 < nviews; ++v2) {
			if (v == v2) continue;
			if (fabs(dot(load3(cams[v].pdir), load3(cams[v2].pdir))) > p->neighbour_min_dot) {
				const Vec3 d = load3(cams[v].C) - load3(cams[v2].C);
				nearViews.push_back(std::make_pair(dot(d, d), v2));
			}
		}
		size_t end = nearViews.size();
		if ((size_t)p->num_neighbours < nearViews.size()) {
			std::sort(nearViews.begin(), nearViews.end());
			end = (size_t)p->num_neighbours;
		}
		for (size_t k = 0; k < end; ++k) neigh[(size_t)v*p->num_neighbours + k] = nearViews[k].second;
		count[v] = (int32_t)end;
	}
	return SRH_OK;
}

// ------------------------------------------------------------------ context
extern "C" int srh_create(int device, srh_context **out) {
	if (!out) return fail(SRH_E_INVALID, "null out");
	*out = nullptr;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) {
		(void)hipGetLastError();
		return fail(SRH_E_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
		            e != hipSuccess ? hipGetErrorString(e) : "device count 0");
	}
	if (device < 0 || device >= n) return fail(SRH_E_INVALID, "device ordinal %d outside [0,%d)", device, n);
	HIP_TRY(hipSetDevice(device));
	srh_context *c = new srh_context();
	c->device = device;
	memset(&c->stats, 0, sizeof(c->stats));
	// (no environment overrides: which arithmetic / path a host runs is decided by srh_set_option alone, and reported
	// by srh_get_stats; bench.py maps its own SRH_BENCH_* variables to options)
	{ int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->num_cus = cus; }
	hipError_t e2 = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
	if (e2 != hipSuccess) { delete c; return fail(SRH_E_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e2)); }
	c->stream = c->own_stream;
	if (hipMalloc((void **)&c->d_views, sizeof(ViewDev)*SRH_MAX_VIEWS) != hipSuccess ||
	    hipMalloc((void **)&c->d_slots, sizeof(int32_t)*SRH_MAX_VIEWS) != hipSuccess ||
	    hipMalloc((void **)&c->d_cnt, sizeof(Counters)) != hipSuccess ||
	    hipMalloc((void **)&c->d_span, 4*sizeof(int)) != hipSuccess) {
		srh_destroy(c);
		return fail(SRH_E_DEVICE, "hipMalloc of context tables failed");
	}
	// ordered on the context's own (non-blocking) stream, which the null stream does not synchronise with
	if (hipMemsetAsync(c->d_views, 0, sizeof(ViewDev)*SRH_MAX_VIEWS, c->stream) != hipSuccess ||
	    hipMemsetAsync(c->d_cnt, 0, sizeof(Counters), c->stream) != hipSuccess ||
	    hipStreamSynchronize(c->stream) != hipSuccess) {
		srh_destroy(c);
		return fail(SRH_E_DEVICE, "initialising the context tables failed");
	}
	*out = c;
	return SRH_OK;
}

// a view plane: from the process's pool when it holds a block of that size, else from the driver (the planes of a 1920 x 1080
// view are 230 MB: released and asked for again with every stereo object, like the band buffers)
static hipError_t plane_alloc(void **p, size_t bytes) {
	size_t got = 0;
	if (void *q = pool_take(bytes, &got)) { *p = q; return hipSuccess; }
	return hipMalloc(p, bytes);
}

// (hipFree waited for the device before it released a plane; so does this, before a plane can reach another context)
static void free_view(ViewHost &v) {
	const size_t n = (size_t)v.w*v.h;
	if (v.rgba || v.gray || v.edges || v.tvp || v.geo5) (void)hipDeviceSynchronize();
	pool_give(v.rgba, n*4);
	if (v.mask) hipFree(v.mask);
	pool_give(v.gray, n*sizeof(double));
	pool_give(v.gray_tv, n*sizeof(double));
	pool_give(v.depth, n*sizeof(double));
	pool_give(v.edges, 4*n*sizeof(double));
	if (v.full) hipFree(v.full);
	pool_give(v.tvp, v.tvp ? padded_size(v.w, v.h)*sizeof(double) : 0);
	pool_give(v.geo5, v.geo5 ? geo5_doubles(v.w, v.h)*sizeof(double) : 0);
	if (v.fullp) hipFree(v.fullp);
	if (v.peaks) hipFree(v.peaks);
	if (v.mrf) hipFree(v.mrf);
	if (v.act) hipFree(v.act);
	v = ViewHost();
}

extern "C" void srh_destroy(srh_context *c) {
	if (!c) return;
	hipSetDevice(c->device);
	if (c->own_stream) hipStreamSynchronize(c->own_stream);
	for (MvsSlot &S : c->mvs_slot) {
		if (S.stream) { hipStreamSynchronize(S.stream); }
		S.pending = false;
	}
	if (c->side_stream) { hipStreamSynchronize(c->side_stream); hipStreamDestroy(c->side_stream); }
	if (c->side_go) hipEventDestroy(c->side_go);
	if (c->side_done) hipEventDestroy(c->side_done);
	drain_profile(c);
	for (MvsSlot &S : c->mvs_slot) {
		if (S.stream) hipStreamDestroy(S.stream);
		if (S.ev) hipEventDestroy(S.ev);
		if (S.done) hipEventDestroy(S.done);
		if (S.h_maxc) hipHostFree(S.h_maxc);
		pool_give(S.wbuf, S.wbuf_cap*sizeof(double));
		pool_give(S.cost, S.cost_cap*sizeof(double));
		if (S.tnum) hipFree(S.tnum);
		pool_give(S.lcount, S.lcount_cap*sizeof(int32_t));
		pool_give(S.lcand, S.lcand_cap*sizeof(uint32_t));
		if (S.mvs_wdesc) hipFree(S.mvs_wdesc);
		if (S.mvs_nwin) hipFree(S.mvs_nwin);
		if (S.d_cnt) hipFree(S.d_cnt);
		if (S.d_span) hipFree(S.d_span);
	}
	for (auto &v : c->views) free_view(v);
	if (c->d_views) hipFree(c->d_views);
	if (c->d_slots) hipFree(c->d_slots);
	if (c->d_cnt) hipFree(c->d_cnt);
	if (c->d_span) hipFree(c->d_span);
	// (every stream of the context has been waited for above: its band buffers are idle and go to the process's pool)
	(void)hipDeviceSynchronize();
	(void)hipStreamSynchronize(c->stream);
	if (c->tv_slot.stream) (void)hipStreamSynchronize(c->tv_slot.stream);
	pool_give(c->wbuf, c->wbuf_cap*sizeof(double));
	pool_give(c->cost, c->cost_cap*sizeof(double));
	if (c->tnum) hipFree(c->tnum);
	pool_give(c->pconst, c->pconst_cap*sizeof(double));
	pool_give(c->prange, c->prange_cap*sizeof(PixRange));
	pool_give(c->cflag, c->cflag_cap*sizeof(uint32_t));
	if (c->stpl) hipFree(c->stpl);
	if (c->tileflag) hipFree(c->tileflag);
	for (auto &d : c->tv_defer) { if (d.host) hipHostFree(d.host); if (d.span) hipHostFree(d.span); }
	{
		srh_context::TvSlot &T = c->tv_slot;
		if (T.stream) { hipStreamSynchronize(T.stream); hipStreamDestroy(T.stream); }
		if (T.go) hipEventDestroy(T.go);
		if (T.done) hipEventDestroy(T.done);
		if (T.d_cnt) hipFree(T.d_cnt);
		if (T.d_span) hipFree(T.d_span);
		if (T.tnum) hipFree(T.tnum);
		pool_give(T.wbuf, T.wbuf_cap*sizeof(double));
		pool_give(T.cost, T.cost_cap*sizeof(double));
		pool_give(T.pconst, T.pconst_cap*sizeof(double));
		pool_give(T.prange, T.prange_cap*sizeof(PixRange));
		pool_give(T.cflag, T.cflag_cap*sizeof(uint32_t));
		pool_give(T.lcand, T.lcand_cap*sizeof(uint32_t));
		pool_give(T.lrowinfo, T.lrowinfo_cap*sizeof(uint32_t));
		pool_give(T.lcount, T.lcount_cap*sizeof(int32_t));
		pool_give(T.lmeta, T.lmeta_cap*sizeof(int32_t));
		if (T.stpl) hipFree(T.stpl);
		if (T.tileflag) hipFree(T.tileflag);
	}
	if (c->mrf) hipFree(c->mrf);
	if (c->mrf_peaks) hipFree(c->mrf_peaks);
	for (int i = 0; i < SRH_MAX_VIEWS; ++i) if (c->mrf_stream[i]) hipStreamDestroy(c->mrf_stream[i]);
	if (c->mrf_host) hipHostFree(c->mrf_host);
	pool_give(c->lcount, c->lcount_cap*sizeof(int32_t));
	pool_give(c->lcand, c->lcand_cap*sizeof(uint32_t));
	pool_give(c->lrowinfo, c->lrowinfo_cap*sizeof(uint32_t));
	pool_give(c->lmeta, c->lmeta_cap*sizeof(int32_t));
	if (c->mvs_wdesc) hipFree(c->mvs_wdesc);
	if (c->mvs_nwin) hipFree(c->mvs_nwin);
	if (c->comm) (void)rccl_comm_destroy(c->comm);
	if (c->own_stream) hipStreamDestroy(c->own_stream);
	delete c;
}

extern "C" int srh_set_stream(srh_context *c, void *hip_stream) {
	if (!c) return fail(SRH_E_INVALID, "null context");
	{ const int rc = mvs_settle_all(c); if (rc) return rc; }
	c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
	return SRH_OK;
}

extern "C" int srh_set_hooks(srh_context *c, const volatile int *cancel, srh_progress_fn progress_fn, void *user) {
	if (!c) return fail(SRH_E_INVALID, "null context");
	c->cancel = cancel; c->progress = progress_fn; c->user = user;
	return SRH_OK;
}

static void release_band_buffers(srh_context *c);

extern "C" int srh_set_option(srh_context *c, const char *name, long value) {
	if (!c || !name) return fail(SRH_E_INVALID, "null argument");
	{ const int rc = mvs_settle_all(c); if (rc) return rc; }       // options apply to work queued from here on
	if (!strcmp(name, "list_rows")) { c->list_rows = value != 0; return SRH_OK; }
	if (!strcmp(name, "list_count_pass")) { c->list_count_pass = value != 0; return SRH_OK; }
	if (!strcmp(name, "force_generic")) { c->force_generic = value != 0; c->force_walk = value == 2; return SRH_OK; }
	if (!strcmp(name, "fused")) { c->use_fused = value != 0; return SRH_OK; }
	if (!strcmp(name, "arith")) {
		if (value < 0 || value > 3) return fail(SRH_E_INVALID, "arith must be 0 (exact), 1 (fma), 2 (f32) or 3 (certified fma)");
		c->arith = (int)value; return SRH_OK;
	}
	if (!strcmp(name, "force_dense")) { c->force_dense = value != 0; return SRH_OK; }
	if (!strcmp(name, "cert_form")) {
		if (value != 1 && value != 2) return fail(SRH_E_INVALID, "cert_form must be 1 (one-pass) or 2 (two fused sweeps)");
		c->cert_form = (int)value; return SRH_OK;
	}
	if (!strcmp(name, "strip")) {
		if (value != 0 && value != 1 && value != 4 && value != 8) return fail(SRH_E_INVALID, "strip must be 0, 1, 4 or 8");
		c->strip = (int)value; return SRH_OK;
	}
	if (!strcmp(name, "mvs_staged")) { c->mvs_staged = value != 0; return SRH_OK; }
	if (!strcmp(name, "mvs_async")) { c->mvs_async = value != 0; return SRH_OK; }
	if (!strcmp(name, "band_budget_mb")) {
		if (value < 1) return fail(SRH_E_INVALID, "band_budget_mb must be >= 1");
		c->wbuf_budget = (size_t)value << 20;
		c->budget_cap = 0;                                         // a new request: earlier out-of-memory halvings are forgotten
		return SRH_OK;
	}
	// tests of the budget logic: pretend the device has only `value` MB to give / refuse band buffers above `value` MB
	if (!strcmp(name, "mem_limit_mb")) { c->mem_limit = value > 0 ? (size_t)value << 20 : 0; return SRH_OK; }
	if (!strcmp(name, "debug_trace")) { c->debug_trace = (int)value; return SRH_OK; }
	if (!strcmp(name, "band_pool_mb")) {
		// process-wide: how many bytes of released band buffers wait for the next context (0: none, and what waits now is freed)
		if (value < 0) return fail(SRH_E_INVALID, "band_pool_mb must be >= 0");
		{ std::lock_guard<std::mutex> g(g_pool_mu); g_pool_cap = (size_t)value << 20; }
		if (value == 0) pool_flush(-1);
		return SRH_OK;
	}
	if (!strcmp(name, "debug_alloc_limit_mb")) {
		c->alloc_limit = g_alloc_limit = value > 0 ? (size_t)value << 20 : 0;
		// the band buffers a bigger run left behind would serve every later request without an allocation: start afresh
		for (MvsSlot &S : c->mvs_slot) if (S.stream) HIP_TRY(hipStreamSynchronize(S.stream));
		if (c->side_stream) HIP_TRY(hipStreamSynchronize(c->side_stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		release_band_buffers(c);
		return SRH_OK;
	}
	if (!strcmp(name, "tv_overlap")) { c->tv_overlap = value != 0; return SRH_OK; }
	if (!strcmp(name, "tscan")) { c->tscan = value != 0; return SRH_OK; }
	if (!strcmp(name, "geodma")) { c->geodma = value != 0; return SRH_OK; }
	if (!strcmp(name, "f32_form")) { c->f32_form = value != 0; return SRH_OK; }
	if (!strcmp(name, "rows_masked")) { c->rows_masked = (int)value; return SRH_OK; }   // 0 off, 1 by the other view's share of fully usable windows (default), 2 always
	if (!strcmp(name, "side_weights")) { c->side_weights = value != 0; return SRH_OK; }
	// test of the cut-list redo: the capacity the next MultiViewStereo estimate is queued with (0 = forget what was learnt)
	if (!strcmp(name, "debug_mvs_cmax_hint")) { c->mvs_cmax_hint = value > 0 ? (int)((value + 7) & ~7L) : 0; return SRH_OK; }
#ifdef SRH_EXPERIMENT
	if (!strcmp(name, "exp_repeat")) { exp_set((int)value, -1); return SRH_OK; }
	if (!strcmp(name, "exp_lds_pad")) { exp_set(-1, (int)value); return SRH_OK; }
	if (!strcmp(name, "exp_scan_mode")) { exp_set_scan((int)value); return SRH_OK; }
	if (!strcmp(name, "exp_walk_mode")) { exp_set_walk((int)value); return SRH_OK; }
	if (!strcmp(name, "exp_rows_mode")) { exp_set_rows((int)value); return SRH_OK; }
#endif
	return fail(SRH_E_INVALID, "unknown option '%s'", name);
}

extern "C" int srh_synchronize(srh_context *c) {
	if (!c) return fail(SRH_E_INVALID, "null context");
	HIP_TRY(hipSetDevice(c->device));
	{ const int rc = mvs_settle_all(c); if (rc) return rc; }
	HIP_TRY(hipStreamSynchronize(c->stream));
	return SRH_OK;
}

// ------------------------------------------------------------------ views
extern "C" int srh_view_upload(srh_context *c, int slot, int w, int h,
                               const uint8_t *rgba, const uint8_t *mask, const srh_camera *cam)
{
	int rc = check_slot(c, slot, false); if (rc) return rc;
	if (w <= 0 || h <= 0 || (size_t)w*h > ((size_t)1 << 30)) return fail(SRH_E_INVALID, "bad image size %dx%d", w, h);
	if (!rgba || !cam) return fail(SRH_E_INVALID, "null rgba / camera");
	HIP_TRY(hipSetDevice(c->device));
	ViewHost &v = c->views[slot];
	const size_t n = (size_t)w*h;
	if (!v.present || v.w != w || v.h != h) {
		HIP_TRY(hipStreamSynchronize(c->stream));
		free_view(v);
		HIP_TRY(plane_alloc((void **)&v.rgba, n*4));
		HIP_TRY(hipMalloc((void **)&v.mask, n));
		HIP_TRY(plane_alloc((void **)&v.gray, n*sizeof(double)));
		HIP_TRY(plane_alloc((void **)&v.gray_tv, n*sizeof(double)));
		HIP_TRY(plane_alloc((void **)&v.depth, n*sizeof(double)));
		HIP_TRY(plane_alloc((void **)&v.edges, 4*n*sizeof(double)));
		HIP_TRY(hipMalloc((void **)&v.full, full_stat_offset(n) + 16));      // (+ the map's two counters, full_window_kernel)
		v.w = w; v.h = h; v.present = true;
	}
	v.cam = *cam;
	v.full_r = 0;                                               // recomputed on demand for the new pixels
	v.tvp_valid = false; v.fullp_r = 0; v.geo5_valid = false; v.geo5_denied = false;
	v.peaks_k = 0;                                              // the top-K peaks belonged to the previous image
	if (c->mrf_w == w && c->mrf_h == h) c->mrf_w = c->mrf_h = c->mrf_k = 0;
	for (int j = 0; j < SRH_MAX_VIEWS; ++j) { v.list_mode[j] = 0; c->views[j].list_mode[slot] = 0; }   // new geometry
	HIP_TRY(hipMemcpyAsync(v.rgba, rgba, n*4, hipMemcpyHostToDevice, c->stream));
	if (mask) HIP_TRY(hipMemcpyAsync(v.mask, mask, n, hipMemcpyHostToDevice, c->stream));
	else      HIP_TRY(hipMemsetAsync(v.mask, 1, n, c->stream));
	if (mask) v.hmask.assign(mask, mask + n); else v.hmask.clear();
	v.act_valid = false;
	{ Scope s(c, "prep_view_kernel"); launch_prep_view(c->stream, v.rgba, v.mask, w, h, v.gray, v.gray_tv); }
	{ Scope s(c, "edge_planes_kernel"); launch_edge_planes(c->stream, v.rgba, w, h, v.edges); }
	launch_fill(c->stream, v.depth, n, __builtin_nan(""));
	ViewDev d;
	d.w = w; d.h = h; d.rgba = v.rgba; d.mask = v.mask; d.gray = v.gray; d.gray_tv = v.gray_tv; d.depth = v.depth;
	d.cam = v.cam;
	HIP_TRY(hipMemcpyAsync(c->d_views + slot, &d, sizeof(d), hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));   // the host buffers and `d` may go away after return
	HIP_TRY(hipGetLastError());
	return SRH_OK;
}

extern "C" int srh_view_size(srh_context *c, int slot, int *w, int *h) {
	int rc = check_slot(c, slot, true); if (rc) return rc;
	if (w) *w = c->views[slot].w;
	if (h) *h = c->views[slot].h;
	return SRH_OK;
}

extern "C" int srh_view_depth_download(srh_context *c, int slot, double *host_out) {
	int rc = check_slot(c, slot, true); if (rc) return rc;
	if (!host_out) return fail(SRH_E_INVALID, "null output");
	HIP_TRY(hipSetDevice(c->device));
	const ViewHost &v = c->views[slot];
	HIP_TRY(hipMemcpyAsync(host_out, v.depth, (size_t)v.w*v.h*sizeof(double), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return SRH_OK;
}

extern "C" int srh_view_depth_upload(srh_context *c, int slot, const double *host_in) {
	int rc = check_slot(c, slot, true); if (rc) return rc;
	if (!host_in) return fail(SRH_E_INVALID, "null input");
	HIP_TRY(hipSetDevice(c->device));
	const ViewHost &v = c->views[slot];
	HIP_TRY(hipMemcpyAsync(v.depth, host_in, (size_t)v.w*v.h*sizeof(double), hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return SRH_OK;
}

extern "C" int srh_view_depth_device_ptr(srh_context *c, int slot, void **dev_ptr) {
	int rc = check_slot(c, slot, true); if (rc) return rc;
	if (!dev_ptr) return fail(SRH_E_INVALID, "null output");
	*dev_ptr = c->views[slot].depth;
	return SRH_OK;
}

extern "C" int srh_view_depth_copy_to_device(srh_context *c, int slot, void *dst_dev, size_t dst_bytes) {
	int rc = check_slot(c, slot, true); if (rc) return rc;
	if (!dst_dev) return fail(SRH_E_INVALID, "null destination");
	HIP_TRY(hipSetDevice(c->device));
	const ViewHost &v = c->views[slot];
	if (dst_bytes < (size_t)v.w*v.h*sizeof(double))
		return fail(SRH_E_INVALID, "destination of %zu bytes is smaller than the %dx%d depth map of slot %d", dst_bytes, v.w, v.h, slot);
	HIP_TRY(hipMemcpyAsync(dst_dev, v.depth, (size_t)v.w*v.h*sizeof(double), hipMemcpyDeviceToDevice, c->stream));
	return SRH_OK;
}

extern "C" int srh_view_depth_copy_from_device(srh_context *c, int slot, const void *src_dev, size_t src_bytes) {
	int rc = check_slot(c, slot, true); if (rc) return rc;
	if (!src_dev) return fail(SRH_E_INVALID, "null source");
	HIP_TRY(hipSetDevice(c->device));
	const ViewHost &v = c->views[slot];
	if (src_bytes < (size_t)v.w*v.h*sizeof(double))
		return fail(SRH_E_INVALID, "source of %zu bytes is smaller than the %dx%d depth map of slot %d", src_bytes, v.w, v.h, slot);
	HIP_TRY(hipMemcpyAsync(v.depth, src_dev, (size_t)v.w*v.h*sizeof(double), hipMemcpyDeviceToDevice, c->stream));
	return SRH_OK;
}

// ------------------------------------------------------------------ runs
// bytes of device memory the band buffers hold now (they are re-used or re-allocated by the next run)
static size_t held_band_bytes(const srh_context *c) {
	size_t b = c->wbuf_cap*8 + c->cost_cap*8 + c->pconst_cap*8 + c->prange_cap*sizeof(PixRange) + c->lcount_cap*4 + c->lcand_cap*4
	         + c->lrowinfo_cap*4 + c->lmeta_cap*4 + c->mvs_wdesc_cap*4 + c->mvs_nwin_cap*4 + c->cflag_cap*4;
	for (const MvsSlot &S : c->mvs_slot)
		b += S.wbuf_cap*8 + S.cost_cap*8 + S.lcount_cap*4 + S.lcand_cap*4 + S.mvs_wdesc_cap*4 + S.mvs_nwin_cap*4;
	const srh_context::TvSlot &T = c->tv_slot;
	b += T.wbuf_cap*8 + T.cost_cap*8 + T.pconst_cap*8 + T.prange_cap*sizeof(PixRange) + T.cflag_cap*4 + T.lcand_cap*4 + T.lrowinfo_cap*4
	   + T.lcount_cap*4 + T.lmeta_cap*4;
	for (const ViewHost &v : c->views) if (v.geo5) b += geo5_doubles(v.w, v.h)*sizeof(double);   // (released with the band buffers)
	return b;
}

// The band budget of a run: what the caller asked for, but no more than a quarter of what the device can give right now
// (free memory + what the band buffers already hold): the MultiViewStereo path treats the budget as a target (bands up
// to 1.25x) and keeps two views' band buffers, the TwoView path adds the per-pixel constant planes.  A GPU shared with
// other processes, several contexts on one GPU or a smaller device get thinner bands instead of an allocation failure.
static size_t band_budget(srh_context *c) {
	size_t b = c->wbuf_budget;
	if (c->budget_cap && b > c->budget_cap) b = c->budget_cap;
	size_t free_b = 0, total_b = 0;
	if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
		size_t avail = free_b + held_band_bytes(c) + pool_bytes_on(c->device);   // (the pool's blocks are this context's to take)
		if (c->mem_limit && avail > c->mem_limit) avail = c->mem_limit;
		if (b > avail/4) b = avail/4;
	} else (void)hipGetLastError();
	if (b < ((size_t)1 << 20)) b = (size_t)1 << 20;
	c->budget_used = b;
	return b;
}

static void release_band_buffers(srh_context *c) {
	pool_flush(c->device);                                        // (out of memory: the pool's blocks go back to the driver first)
	auto drop = [](auto *&p, size_t &cap) { if (p) (void)hipFree(p); p = nullptr; cap = 0; };
	drop(c->wbuf, c->wbuf_cap); drop(c->cost, c->cost_cap); drop(c->pconst, c->pconst_cap); drop(c->prange, c->prange_cap);
	drop(c->lcount, c->lcount_cap); drop(c->lcand, c->lcand_cap); drop(c->lrowinfo, c->lrowinfo_cap); drop(c->lmeta, c->lmeta_cap);
	drop(c->mvs_wdesc, c->mvs_wdesc_cap); drop(c->mvs_nwin, c->mvs_nwin_cap); drop(c->cflag, c->cflag_cap);
	drop(c->stpl, c->stpl_cap); drop(c->tileflag, c->tileflag_cap);
	for (MvsSlot &S : c->mvs_slot) {
		drop(S.wbuf, S.wbuf_cap); drop(S.cost, S.cost_cap); drop(S.lcount, S.lcount_cap); drop(S.lcand, S.lcand_cap);
		drop(S.mvs_wdesc, S.mvs_wdesc_cap); drop(S.mvs_nwin, S.mvs_nwin_cap);
	}
	// (called with everything drained; while the second TwoView pass is being queued the slot holds the FIRST pass's
	// buffers, which may be in use: with_thinner_bands waits for that stream as well)
	srh_context::TvSlot &T = c->tv_slot;
	drop(T.wbuf, T.wbuf_cap); drop(T.cost, T.cost_cap); drop(T.pconst, T.pconst_cap); drop(T.prange, T.prange_cap);
	drop(T.cflag, T.cflag_cap); drop(T.lcand, T.lcand_cap); drop(T.lrowinfo, T.lrowinfo_cap); drop(T.lcount, T.lcount_cap);
	drop(T.lmeta, T.lmeta_cap);
	drop(T.stpl, T.stpl_cap); drop(T.tileflag, T.tileflag_cap);
	// the geodesic_dma_kernel's second copy of the edge / tap planes (42 B per pixel and view, made on first use): memory an
	// out-of-memory retry must be able to reclaim -- the register-staged kernel needs no such copy and serves the view until
	// it is uploaded again
	for (ViewHost &v : c->views)
		if (v.geo5) { (void)hipFree(v.geo5); v.geo5 = nullptr; v.geo5_valid = false; v.geo5_denied = true; }
}

// Run `body`; when it fails because a band buffer could not be allocated, wait for everything in flight, release the
// band buffers, halve the band budget and run it again -- down to one-megabyte bands (a band is at least one image row).
template <class F>
static int with_thinner_bands(srh_context *c, F body) {
	for (;;) {
		g_oom = false;
		g_alloc_limit = c->alloc_limit;
		const int rc = body();
		if (rc != SRH_E_DEVICE || !g_oom) return rc;
		g_oom = false;
		const size_t used = c->budget_used;
		if (used <= ((size_t)1 << 20)) return rc;                   // already the thinnest bands there are
		// Everything in flight is drained before the band buffers go.  A MultiViewStereo view queued earlier on the other
		// slot stays PENDING: its kernels are complete now, but its list-capacity check (mvs_settle_slot: h_maxc is pinned
		// and stays valid) has not been made -- the retried call, or whichever entry point comes next, settles it and
		// redoes the view if a list was cut.
		for (MvsSlot &S : c->mvs_slot) if (S.stream) (void)hipStreamSynchronize(S.stream);
		if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);
		(void)hipStreamSynchronize(c->stream);
		if (c->tv_slot.stream) (void)hipStreamSynchronize(c->tv_slot.stream);   // (either of the two is the other pass's)
		release_band_buffers(c);
		c->budget_cap = used/2;
		c->stats.band_retries += 1;
	}
}

static int band_rows(srh_context *c, int W, int H, int T) {
	size_t rows = band_budget(c) / ((size_t)T*sizeof(double)*(size_t)W);
	if (rows < 1) rows = 1;
	if (rows > (size_t)H) rows = H;
	return (int)rows;
}

static int fetch_counters(srh_context *c, int used_dense) {
	Counters h;
	HIP_TRY(hipMemcpyAsync(&h, c->d_cnt, sizeof(h), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	c->stats.n_pixels = (int64_t)h.n_pixels;
	c->stats.n_eval = (int64_t)h.n_eval;
	c->stats.n_eval_device = (int64_t)h.n_eval_device;
	c->stats.scan_tiles_template = (int64_t)h.scan_tiles_template;
	c->stats.scan_tiles_walked = (int64_t)h.scan_tiles_walked;
	c->stats.mvs_waves_staged = (int64_t)h.mvs_waves_staged;
	c->stats.mvs_waves_listed = (int64_t)h.mvs_waves_listed;
	c->stats.used_dense_path = used_dense;
	c->stats.used_fused_kernel = c->last_fused ? 1 : 0;
#ifdef SRH_PROFILE_PHASES
	{
		unsigned long long g[8];
		geodesic_phases_fetch(g);
		if (g[7])
			fprintf(stderr, "[srh prof] geodesic kernel: %llu waves, cycles per wave: staging %.0f  sweeps %.0f  exp %.0f  rows out %.0f  pconst: first sweep %.0f  second sweep %.0f  rest %.0f\n",
			        g[7], (double)g[0]/g[7], (double)g[1]/g[7], (double)g[2]/g[7], (double)g[4]/g[7], (double)g[5]/g[7], (double)g[6]/g[7], (double)g[3]/g[7]);
	}
	if (h.dbg_waves && h.dbg_blocks && h.dbg_phase[3] && h.strip_ticket) {
		const double nw = (double)h.dbg_waves, nt = (double)h.dbg_blocks/nw;
		fprintf(stderr, "[srh dbg] strip: waves %llu, tiles/wave %.1f (%.1f with phase 2), cycles/wave %.0f | per tile: ticket %.0f  own requests %.0f  barriers %.0f  setup %.0f  "
		        "block loops %.0f  p2 lists %.0f  p2 blocks %.0f  p2 singles %.0f\n", h.dbg_waves, nt, (double)h.dbg_cycles/nw, (double)h.dbg_total_cycles/nw,
		        h.dbg_phase[0]/nw/nt, h.dbg_phase[1]/nw/nt, h.dbg_phase[2]/nw/nt, h.dbg_phase[3]/nw/nt, h.dbg_phase[4]/nw/nt,
		        h.dbg_phase[5]/nw/nt, h.dbg_phase[6]/nw/nt, h.dbg_phase[7]/nw/nt);
		static const char *names[8] = { "ticket", "own requests", "barriers", "setup", "block loops", "p2 lists", "p2 blocks", "p2 singles" };
		const int nwv = h.dbg_wave[8*4 + 4] ? 8 : 4;
		for (int k = 0; k < 8; ++k) {
			fprintf(stderr, "[srh dbg]   %-13s by wave:", names[k]);
			for (int w = 0; w < nwv; ++w) fprintf(stderr, " %8.0f", h.dbg_wave[8*k + w]/((double)h.dbg_blocks/nwv));
			fprintf(stderr, "\n");
		}
	} else
	if (h.dbg_waves && !c->last_fused)
		fprintf(stderr, "[srh dbg] dense: waves %llu, cycles/wave %.0f, fast blocks/wave %.1f, cycles/fast block %.0f\n",
		        h.dbg_waves, (double)h.dbg_total_cycles/h.dbg_waves, (double)h.dbg_blocks/h.dbg_waves,
		        h.dbg_blocks ? (double)h.dbg_cycles/h.dbg_blocks : 0.0);
	if (h.dbg_waves && !c->last_fused && !h.strip_ticket)
		fprintf(stderr, "[srh dbg] phases/wave: stage_w %.0f prologue %.0f sync %.0f stage_rt %.0f compute %.0f tail %.0f\n",
		        (double)h.dbg_phase[0]/h.dbg_waves, (double)h.dbg_phase[1]/h.dbg_waves, (double)h.dbg_phase[2]/h.dbg_waves,
		        (double)h.dbg_phase[3]/h.dbg_waves, (double)h.dbg_phase[4]/h.dbg_waves, (double)h.dbg_phase[5]/h.dbg_waves);
	if (h.dbg_waves && c->last_fused)
		fprintf(stderr, "[srh dbg] fused: %llu tiles, %llu with select-form work, %llu pixels with an unusable tap of their own\n",
		        h.dbg_waves, h.n_listed, h.n_slots);
	if (h.dbg_waves && c->last_fused)
		fprintf(stderr, "[srh dbg] fused, role wave cycles/tile: A geometry %.0f  B accept %.0f  C stage %.0f  D prologue %.0f  E1 blocks %.0f  E2 general %.0f  F wta %.0f  barriers %.0f | other waves: blocks %.0f barriers %.0f total %.0f\n",
		        (double)h.dbg_phase[0]/h.dbg_waves, (double)h.dbg_phase[1]/h.dbg_waves, (double)h.dbg_phase[2]/h.dbg_waves,
		        (double)h.dbg_phase[3]/h.dbg_waves, (double)h.dbg_phase[4]/h.dbg_waves, (double)h.dbg_phase[5]/h.dbg_waves,
		        (double)h.dbg_phase[6]/h.dbg_waves, (double)h.dbg_phase[7]/h.dbg_waves,
		        (double)h.dbg_cycles/(3.0*h.dbg_waves), (double)h.dbg_blocks/(3.0*h.dbg_waves), (double)h.dbg_total_cycles/(3.0*h.dbg_waves));
	if (h.dbg_phase[6] && !c->last_fused)
		fprintf(stderr, "[srh dbg] rows: tasks %llu fast %llu, rows/pixel %.2f, slots/px %.1f, wave iterations %llu all-fast %llu\n",
		        h.dbg_phase[6], h.dbg_phase[7], (double)h.dbg_cycles/(double)h.n_pixels, 8.0*h.dbg_phase[6]/(double)h.n_pixels,
		        h.dbg_blocks, h.dbg_total_cycles);
	if (h.dbg_phase[6] && !c->last_fused && h.dbg_wave[8])
		fprintf(stderr, "[srh dbg] rows cost kernel, wave tiles %llu: cycles per tile: staging %.0f  constants %.0f  phase 1 %.0f  p2 blocks %.0f  p2 singles %.0f | tiles with blocks %llu (%llu blocks: %.0f cycles per such tile), with singles %llu (%llu singles: %.0f cycles per such tile); rounds if the tile's tasks were dealt to its 64 lanes: %.2f per tile\n",
		        h.dbg_wave[8], (double)h.dbg_wave[0]/h.dbg_wave[8], (double)h.dbg_wave[1]/h.dbg_wave[8], (double)h.dbg_wave[2]/h.dbg_wave[8],
		        (double)h.dbg_wave[3]/h.dbg_wave[8], (double)h.dbg_wave[4]/h.dbg_wave[8], h.dbg_wave[9], h.dbg_wave[10],
		        h.dbg_wave[9] ? (double)h.dbg_wave[3]/h.dbg_wave[9] : 0.0, h.dbg_wave[11], h.dbg_wave[12], h.dbg_wave[11] ? (double)h.dbg_wave[4]/h.dbg_wave[11] : 0.0, (double)h.dbg_wave[13]/h.dbg_wave[8]);
#endif
	return SRH_OK;
}

// support windows of rows [by, by+nr) of view `ref` into c->wbuf
static void run_weights(srh_context *c, int ref, int W, const srh_params &p, int by, int nr, size_t wstride,
                        double *pconst = nullptr, bool wimg = false) {
	if (p.weight_kind == SRH_WEIGHT_GEODESIC && !c->force_generic && wimg && c->geodma && p.window_radius == 5 && !c->views[ref].geo5_denied) {
		// the dense path's windows: tiles by LDS-DMA from planes with their borders written out (made once per uploaded view)
		ViewHost &v = c->views[ref];
		bool ok = true;
		if (!v.geo5) {
			if (plane_alloc((void **)&v.geo5, geo5_doubles(v.w, v.h)*sizeof(double)) != hipSuccess) { (void)hipGetLastError(); v.geo5 = nullptr; ok = false; }   // (no room: the register-staged kernel needs no second copy)
			v.geo5_valid = false;
		}
		if (ok && !v.geo5_valid) {
			Scope s(c, "geo5_planes_kernel");
			launch_geo5_planes(c->stream, v.edges, v.gray_tv, v.mask, v.w, v.h, v.geo5);
			v.geo5_valid = true;
		}
		if (ok) {
			Scope s(c, "geodesic_dma_kernel");
			if (launch_geodesic_dma(c->stream, c->d_views, ref, W, v.geo5, p, by, nr, c->wbuf, pconst, c->num_cus)) return;
		}
	}
	if (p.weight_kind == SRH_WEIGHT_GEODESIC && !c->force_generic) {
		Scope s(c, "geodesic_reg_kernel");
		if (launch_geodesic_reg(c->stream, c->d_views, ref, W, c->views[ref].edges, p, by, nr, c->wbuf, wstride, pconst, wimg)) return;
	}
	if (p.weight_kind == SRH_WEIGHT_ADAPTIVE && !c->force_generic) {
		Scope s(c, "adaptive_reg_kernel");
		if (launch_adaptive_reg(c->stream, c->d_views, ref, W, p, by, nr, c->wbuf, wstride, pconst, wimg)) return;
	}
	Scope s(c, "weights_kernel");
	launch_weights(c->stream, c->d_views, ref, W, p, by, nr, c->wbuf, wstride, pconst, wimg);
}

// Can every epipolar curve of `a` in `b` stay on its own image row?  Undistorted, non-refractive
// cameras with the same orientation and the same second and third rows of K, displaced along the
// camera x axis.  This only *proposes* the dense path: the scan kernel verifies every candidate.
static bool rig_is_row_aligned(const srh_camera &a, const srh_camera &b, double *fx_bx) {
	if (a.is_distorted || b.is_distorted || a.is_refractive || b.is_refractive) return false;
	for (int k = 0; k < 9; ++k) if (fabs(a.R[k] - b.R[k]) > 1e-13) return false;
	for (int k = 3; k < 9; ++k) if (fabs(a.K[k] - b.K[k]) > 1e-9*(1.0 + fabs(a.K[k]))) return false;
	if (fabs(a.K[1]) > 1e-12 || fabs(b.K[1]) > 1e-12 || fabs(a.K[3]) > 1e-12 || fabs(a.K[6]) > 1e-12 || fabs(a.K[7]) > 1e-12)
		return false;
	const Vec3 d = load3(b.C) - load3(a.C);
	const Vec3 bc = matvec(a.R, d);                               // baseline in camera axes
	const double len = norm(bc);
	if (!(len > 0) || fabs(bc.y) > 1e-12*len || fabs(bc.z) > 1e-12*len) return false;
	*fx_bx = fabs(b.K[0]*bc.x);
	return true;
}

static int twoview_wta_run(srh_context *c, int ref, int oth, const srh_params *p, int y0, int y1);

// A first guess of a pair's list capacities, before any pass has measured them (the list path of srh_twoview_wta):
// cmax = candidates per pixel, smax = cost slots per pixel (row runs in 8-column blocks).  A curve's candidates are the
// raster points of its kept segments with the joints counted once, so at most the Chebyshev length of the polyline through
// the kept label points + 1, and its slots at most that + 8 per image row it touches.  The host walks a coarse polyline
// (nine labels) for a grid of reference pixels with the very camera model the kernels use (srh_geom.hpp), takes the longest,
// and adds a margin for what nine labels do not see.  Too small a guess repeats the pass with the measured maximum.
static void estimate_list_capacity(const srh_camera &rc, const srh_camera &oc, int W, int H, int OW, int OH,
                                   const srh_params &p, int y0, int y1, int &cmax, int &smax, bool mvs = false)
{
	(void)H;
	const int D = p.num_depth_levels, NL = D < 9 ? D : 9, GX = 13, GY = 9;   // (the grid includes the band's border rows and columns:
	                                                                          // distortion and refraction stretch the curves most there)
	const Vec3 normal = load3(rc.pdir);
	long best_len = 0, best_rows = 0;
	for (int gy = 0; gy < GY; ++gy)
		for (int gx = 0; gx < GX; ++gx) {
			const int x = (int)((gx*(long)(W - 1))/(GX - 1)), y = y0 + (int)((gy*(long)(y1 - y0 - 1))/(GY - 1));
			const Ray ray = cam_unproject(rc, (x + 0.5)/p.image_scale, (y + 0.5)/p.image_scale);
			long len = 0, rows = 0;
			bool have = false;
			int px = 0, py = 0;
			for (int k = 0; k < NL; ++k) {
				const int d = NL > 1 ? (int)(((long)k*(D - 1))/(NL - 1)) : 0;
				Vec3 point = load3(rc.C);                          // (in: the camera centre, out: the label's point on the ray)
				if (!point_from_depth(ray, normal, depth_from_label(p, mvs, d), point)) continue;
				if (!cam_project(oc, point)) continue;
				// (clamped a little outside the other image: what lies further out is no candidate)
				const double cxd = fmin(fmax(point.x*p.image_scale, -64.0), (double)OW + 64.0);
				const double cyd = fmin(fmax(point.y*p.image_scale, -64.0), (double)OH + 64.0);
				if (!(cxd == cxd) || !(cyd == cyd)) continue;
				const int ix = (int)cxd, iy = (int)cyd;
				if (have) {
					const long dx = labs((long)ix - px), dy = labs((long)iy - py);
					len += dx > dy ? dx : dy;
					rows += dy;
				}
				have = true; px = ix; py = iy;
			}
			if (len > best_len) best_len = len;
			if (rows > best_rows) best_rows = rows;
		}
	if (best_rows > SRH_ROWS_NR) best_rows = SRH_ROWS_NR;
	// + 12 % for what nine labels do not see, + one entry per kept segment: a list holds a segment's both end points, the
	// joint of two segments twice (the C5 rig's corner pixel: 379 distinct candidates, 157 joints, a list of 536)
	// (MultiViewStereo's lists drop consecutive duplicates, std::unique: no joint term there)
	long cm = best_len + best_len/8 + (mvs ? 0 : (best_len < D - 1 ? best_len : D - 1)) + 40;
	if (cm > 65520) cm = 65520;
	cmax = (int)((cm + 7) & ~7L);
	long sm = cmax + 8*(best_rows + 2) + 32;
	if (sm > 65528) sm = 65528;
	smax = (int)((sm + 7) & ~7L);
}

extern "C" int srh_twoview_wta(srh_context *c, int ref, int oth, const srh_params *p, int y0, int y1) {
	int rc;
	if ((rc = check_slot(c, ref, true)) || (rc = check_slot(c, oth, true)) || (rc = check_params(p))) return rc;
	return with_thinner_bands(c, [&] { return twoview_wta_run(c, ref, oth, p, y0, y1); });
}

static int twoview_wta_run(srh_context *c, int ref, int oth, const srh_params *p, int y0, int y1) {
	int rc;
	if (ref == oth) return fail(SRH_E_INVALID, "ref and other view are the same slot");
	const ViewHost &L = c->views[ref], &Rv = c->views[oth];
	if (L.w != Rv.w || L.h != Rv.h)
		return fail(SRH_E_INVALID, "TwoViewStereo needs equal-sized views (%dx%d vs %dx%d; twoviewstereo.cpp:116-119)",
		            L.w, L.h, Rv.w, Rv.h);
	HIP_TRY(hipSetDevice(c->device));
	const int W = L.w, H = L.h;
	if (y0 < 0) y0 = 0;
	if (y1 <= 0 || y1 > H) y1 = H;
	if (y1 <= y0) return SRH_OK;
	const int R = p->window_radius;
	const int T = (2*R + 1)*(2*R + 1);
	const size_t budget = band_budget(c);

	// ---- plan: dense row-aligned kernels, or the general curve-walk kernel
	bool dense = !c->force_generic && (R == 5 || R == 2);
	int cstride = 0;
	double fx_bx = 0;
	if (dense && !rig_is_row_aligned(L.cam, Rv.cam, &fx_bx)) {
		// test hook: any pinhole pair may be *proposed* (the scan kernel refutes the proposal, see the redo below)
		const bool pinhole = !L.cam.is_distorted && !Rv.cam.is_distorted && !L.cam.is_refractive && !Rv.cam.is_refractive;
		if (c->force_dense && pinhole) fx_bx = (double)(W + 8)/(p->image_scale*fabs(1.0/p->min_depth - 1.0/p->max_depth));
		else dense = false;
	}
	if (dense) {
		// widest candidate range a pixel can have: disparity(min_depth) - disparity(max_depth) + margins
		const double span = fx_bx*p->image_scale*fabs(1.0/p->min_depth - 1.0/p->max_depth);
		if (!(p->min_depth > 0) || !(p->max_depth > 0) || !(span < 4096.0)) dense = false;
		else cstride = (((int)ceil(span) + 3) + 7) & ~7;
		if (cstride > W + 8) cstride = (W + 8 + 7) & ~7;
	}

	c->last_fused = false;
	// ---- row-aligned rig whose candidate range fits an LDS cost row: one fused kernel per band (srh_fused.hip)
	if (dense && c->use_fused && (c->arith == 0 || c->arith == 3) && p->num_depth_levels <= SRH_FUSED_MAXC &&
	    fx_bx*p->image_scale*fabs(1.0/p->min_depth - 1.0/p->max_depth) + 1.0 <= (double)SRH_FUSED_MAXC) {
		HIP_TRY(hipMemsetAsync(c->d_cnt, 0, sizeof(Counters), c->stream));
		if ((rc = ensure(c->tnum, c->tnum_cap, (size_t)p->num_depth_levels))) return rc;
		{ Scope s(c, "pinhole_label_table_kernel");
		  launch_pinhole_label_table(c->stream, c->d_views, ref, *p, false, c->tnum); }
		size_t rows = budget / ((size_t)T*sizeof(double)*(size_t)W);
		if (rows < 1) rows = 1;
		if (rows > (size_t)(y1 - y0)) rows = (size_t)(y1 - y0);
		if ((rc = ensure(c->wbuf, c->wbuf_cap, wbuf_doubles(W, (int)rows, T)))) return rc;
		bool launched = true;
		for (int by = y0; by < y1 && launched; by += (int)rows) {
			if (cancelled(c)) return fail(SRH_E_CANCELLED, "cancelled");
			const int nr = std::min((int)rows, y1 - by);
			run_weights(c, ref, W, *p, by, nr, SRH_WTILE);
			Scope s(c, "twoview_fused_kernel");
			launched = launch_twoview_fused(c->stream, c->d_views, ref, oth, W, *p, by, nr, c->wbuf, c->tnum, c->d_cnt);
		}
		HIP_TRY(hipGetLastError());
		if (launched) {
			// the result stands only if every curve was monotone, on its row and inside the LDS tile
			Counters hc;
			HIP_TRY(hipMemcpyAsync(&hc, c->d_cnt, sizeof(hc), hipMemcpyDeviceToHost, c->stream));
			HIP_TRY(hipStreamSynchronize(c->stream));
			if (hc.not_row_aligned == 0) {
				c->stats.used_dense_path = 1;
				c->last_fused = true;
				return SRH_OK;
			}
		}
	}

	// the persistent strip form of the cost kernel (srh_strip.hip): exact / fma arithmetic, candidate ranges of a
	// 32-pixel tile inside one LDS chunk; anything else, or a range that turns out wider, takes the per-tile kernel
	// (and enough tiles to keep every persistent workgroup busy for dozens of tiles: a small image is better served by
	// one workgroup per tile; option strip = 4 / 8 forces the strip kernel for tests)
	bool strip = dense && c->strip != 0 && c->arith != 2 && cstride + SRH_WTILE <= strip_chunk_columns();
	// certified arithmetic (arith = 3, the default): fused cost loops in the strip kernel + the certified scan; where the
	// strip kernel does not run, or the parameters leave the bound no room, the reference's arithmetic
	bool cert_ok = c->arith == 3 && cert_bound(*p).ok != 0;
	c->stats.n_certified = c->stats.n_flagged = 0;
	// The exact redo of flagged pixels is launched for a CAPACITY -- the whole band: the list buffer holds every pixel of it,
	// the redo kernels share the list in grid-stride loops and read the count on the device -- so however many pixels flag
	// (adversarial images: exact ties everywhere) each is redone once, the host never waits for the count, and no pass is
	// ever repeated as a whole for it.  (Round 4 launched for 1/32 of the band, at most 16 384 pixels, and repeated the
	// whole pass in mode 0 beyond that: a cliff at 0.79 % of C3.)
	auto redo_capacity = [](size_t band_pixels) { return (int)std::min<size_t>(band_pixels, (size_t)1 << 30); };
	for (int attempt = 0; attempt < 4; ++attempt) {
		HIP_TRY(hipMemsetAsync(c->d_cnt, 0, sizeof(Counters), c->stream));
		// ---- arbitrary geometry: candidate lists (the one-thread-per-pixel walk kernel is the last resort)
		// (a negative wta_margin -- not the reference's: its margin is the constant +1e-10, twoviewstereo.cpp:293 -- makes a
		// revisited winner beat itself, so the candidate lists' dropped joint duplicates would matter: the walk kernel, which
		// visits every point, takes such a run)
		if (!dense && !c->force_walk && R <= 5 && W < 65536 && H < 65536 && p->wta_margin >= 0) {
			const size_t npix = (size_t)(y1 - y0)*W;
			if ((rc = ensure(c->lcount, c->lcount_cap, npix))) return rc;
			ViewHost &O = c->views[oth];
			if (O.full_r != R) {
				Scope s(c, "full_window_kernel");
				uint32_t *stat = (uint32_t *)(O.full + full_stat_offset((size_t)O.w*O.h));
				HIP_TRY(hipMemsetAsync(stat, 0, 2*sizeof(uint32_t), c->stream));
				launch_full_window(c->stream, O.gray_tv, O.w, O.h, R, O.full, stat);
				O.full_r = R;
			}
			// the label-only part of pointFromDepth, once per pass instead of once per pixel and label
			if ((rc = ensure(c->tnum, c->tnum_cap, (size_t)p->num_depth_levels))) return rc;
			{ Scope s(c, "label_plane_table_kernel");
			  launch_label_plane_table(c->stream, c->d_views, ref, *p, false, c->tnum); }
			// list capacity: the longest list seen so far on this context (hint); the FIRST time a guess from the geometry
			// (estimate_list_capacity: a few dozen curves' coarse polylines projected on the host -- round 5 ran a counting pass
			// there, twoview_count_kernel, 10 ms of a C5 pair's first call, and then sized the slots too tightly, so that the
			// pass was repeated: 104 ms for a pair whose steady state is 61).  A run that overflows its capacity is repeated
			// with the true maximum: the guess decides how long a pair's first call takes, never what it computes.
			int cmax = c->list_cmax_hint, smax_guess = 0;
			bool guessed = false;
			if (cmax <= 0 && c->list_count_pass) {
				HIP_TRY(hipMemsetAsync(c->d_span, 0, sizeof(int), c->stream));
				{ Scope s(c, "twoview_count_kernel");
				  launch_twoview_count(c->stream, c->d_views, ref, oth, W, *p, y0, y1 - y0, c->lcount, c->d_cnt, c->d_span, c->tnum); }
				int maxc = 0;
				HIP_TRY(hipMemcpyAsync(&maxc, c->d_span, sizeof(int), hipMemcpyDeviceToHost, c->stream));
				HIP_TRY(hipStreamSynchronize(c->stream));
				cmax = std::max(8, (maxc + 7) & ~7);
			} else if (cmax <= 0) {
				estimate_list_capacity(c->views[ref].cam, O.cam, W, H, O.w, O.h, *p, y0, y1, cmax, smax_guess);
				guessed = true;
			}
			bool rows_mode = c->list_rows && W < 32768 && H < 32768;       // spans and row origins are stored as 16-bit signed
			if (c->views[ref].list_mode[oth] == 2) rows_mode = false;      // learnt: steep curves, list order is cheaper
			int smax = c->list_smax_hint > 0 ? c->list_smax_hint : (guessed ? smax_guess : cmax + 64);
			for (int pass = 0; pass < 6; ++pass) {
				HIP_TRY(hipMemsetAsync(c->d_cnt, 0, sizeof(Counters), c->stream));
				HIP_TRY(hipMemsetAsync(c->d_span, 0, 4*sizeof(int), c->stream));
				const int ccap = rows_mode ? smax : cmax;             // cost values per pixel
				// (row runs: the windows in the LDS-image layout, window rows padded to an even tap count -- the cost kernel's
				// waves fetch them by LDS-DMA)
				const size_t per_px = (size_t)(rows_mode ? (2*p->window_radius + 1)*wimg_wp(p->window_radius) : T)*sizeof(double) + (size_t)ccap*sizeof(double) + (size_t)cmax*sizeof(uint32_t)
				                      + (rows_mode ? (SRH_ROWS_NR + 1)*sizeof(uint32_t) : 0);
				size_t lrows = budget / (per_px*(size_t)W);
				if (lrows < 1) lrows = 1;
				if (lrows > (size_t)(y1 - y0)) lrows = (size_t)(y1 - y0);
				if ((rc = ensure(c->wbuf, c->wbuf_cap, rows_mode ? wimg_doubles(W, (int)lrows, p->window_radius) : wbuf_doubles(W, (int)lrows, T)))) return rc;
				// row runs: lists and row tables are tiled per 64 pixels, cost slots per 32-pixel tile of a row
				const size_t px64 = (lrows*W + 63) & ~(size_t)63, px32 = lrows*(size_t)((W + 7)/8)*8;        /* (cost slots: tiles of 8 pixels) */
				if ((rc = ensure(c->cost, c->cost_cap, (rows_mode ? px32 : lrows*W)*(size_t)ccap))) return rc;
				if ((rc = ensure(c->lcand, c->lcand_cap, (rows_mode ? px64 : lrows*W)*(size_t)cmax))) return rc;
				const bool rows_cert = rows_mode && cert_ok;
				// the row-run cost kernel takes the pixels' constants (meanL, totalWeight, sum2, 1/totalWeight, SA) from the weights
				// kernel, which has the window in registers anyway, instead of making them on one lane in eight per tile
				const bool rows_pc = rows_mode;
				if (rows_pc && (rc = ensure(c->pconst, c->pconst_cap, (lrows*(size_t)W + SRH_WTILE)*SRH_PC))) return rc;
				if (rows_mode) {
					if ((rc = ensure(c->lrowinfo, c->lrowinfo_cap, px64*(size_t)SRH_ROWS_NR))) return rc;
					if ((rc = ensure(c->lmeta, c->lmeta_cap, lrows*W))) return rc;
					if (rows_cert && (rc = ensure(c->cflag, c->cflag_cap, lrows*W + 1))) return rc;
					if (rows_cert) {
						// NaN-bordered planes of both views: the general cost of the certified redo reads them without bound tests
						for (int k = 0; k < 2; ++k) {
							ViewHost &v = c->views[k == 0 ? ref : oth];
							if (!v.tvp) HIP_TRY(plane_alloc((void **)&v.tvp, padded_size(v.w, v.h)*sizeof(double)));
							if (!v.tvp_valid) {
								Scope s(c, "padded_plane_kernel");
								launch_padded_plane(c->stream, v.gray_tv, v.w, v.h, v.tvp);
								v.tvp_valid = true;
							}
						}
					}
				}
				for (int by = y0; by < y1; by += (int)lrows) {
					if (cancelled(c)) return fail(SRH_E_CANCELLED, "cancelled");
					const int nr = std::min((int)lrows, y1 - by);
					int32_t *cnt_band = c->lcount + (size_t)(by - y0)*W;
					if (rows_mode) {
						// The list kernel does not need the support windows, and its last waves drain for long (a wave walks
						// its 64 curves for ~2 ms): the windows are computed on a side stream queued behind it -- the geodesic
						// kernel's register-heavy waves only find room on a SIMD once the list kernel's have left it.
						if (!c->side_stream) {
							HIP_TRY(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
							HIP_TRY(hipEventCreateWithFlags(&c->side_go, hipEventDisableTiming));
							HIP_TRY(hipEventCreateWithFlags(&c->side_done, hipEventDisableTiming));
						}
						HIP_TRY(hipEventRecord(c->side_go, c->stream));             // the window buffer's last readers are done by then
						{ Scope s(c, "twoview_rows_list_kernel");
						  launch_twoview_rows_list(c->stream, c->d_views, ref, oth, W, *p, by, nr, c->lcand, cmax,
						                           cnt_band, c->lrowinfo, c->lmeta, smax, c->d_cnt, c->d_span, c->tnum); }
						if (c->side_weights) {
							HIP_TRY(hipStreamWaitEvent(c->side_stream, c->side_go, 0));
							std::swap(c->stream, c->side_stream);
							run_weights(c, ref, W, *p, by, nr, SRH_WTILE, rows_pc ? c->pconst : nullptr, true);
							std::swap(c->stream, c->side_stream);
							HIP_TRY(hipEventRecord(c->side_done, c->side_stream));
							HIP_TRY(hipStreamWaitEvent(c->stream, c->side_done, 0));
						} else run_weights(c, ref, W, *p, by, nr, SRH_WTILE, rows_pc ? c->pconst : nullptr, true);
						if (rows_cert) HIP_TRY(hipMemsetAsync(c->cflag, 0, sizeof(uint32_t), c->stream));
						HIP_TRY(hipMemsetAsync(&c->d_cnt->strip_ticket, 0, 2*sizeof(unsigned int), c->stream));   // the cost kernel's waves draw their tiles from it
						{ Scope s(c, "twoview_rows_cost_kernel");
						  launch_twoview_rows_cost(c->stream, c->d_views, ref, oth, W, *p, by, nr, c->wbuf, O.full,
						                           c->lrowinfo, c->lmeta, c->cost, smax, c->d_cnt, rows_cert ? (c->cert_form == 1 ? 5 : 3) : 0,
						                           rows_pc ? c->pconst : nullptr, rows_cert && c->rows_masked ? O.tvp : nullptr, c->num_cus,
						                           c->rows_masked == 1 ? (const uint32_t *)(O.full + full_stat_offset((size_t)O.w*O.h)) : nullptr); }
						{ Scope s(c, "twoview_rows_scan_kernel");
						  launch_twoview_rows_scan(c->stream, c->d_views, ref, oth, W, *p, by, nr, cnt_band, c->lcand, cmax,
						                           c->lrowinfo, c->lmeta, c->cost, smax, rows_cert ? c->cflag : nullptr, -1, c->d_cnt); }
						if (rows_cert) {
							// certified arithmetic: the flagged pixels once more in the reference's arithmetic, launched for a capacity
							// (a list cut by a too small capacity is harmless here: the pass is repeated anyway)
							const int cap = redo_capacity((size_t)nr*W);
							{ Scope s(c, "twoview_rows_refill_kernel");
							  launch_twoview_rows_refill(c->stream, W, O.w, *p, by, c->cflag, cap, c->wbuf, c->views[ref].tvp, O.tvp,
							                             c->lrowinfo, c->lmeta, c->cost, smax, c->d_cnt); }
							Scope s(c, "twoview_rows_rescan_kernel");
							launch_twoview_rows_scan(c->stream, c->d_views, ref, oth, W, *p, by, nr, cnt_band, c->lcand, cmax,
							                         c->lrowinfo, c->lmeta, c->cost, smax, c->cflag, cap, c->d_cnt);
						}
						continue;
					}
					run_weights(c, ref, W, *p, by, nr, SRH_WTILE);
					{ Scope s(c, "twoview_list_kernel");
					  launch_twoview_list(c->stream, c->d_views, ref, oth, W, *p, by, nr, c->lcand, cmax,
					                      cnt_band, c->d_cnt, c->d_span, c->tnum); }
					{ Scope s(c, "twoview_list_cost_kernel");
					  launch_twoview_list_cost(c->stream, c->d_views, ref, oth, W, *p, by, nr, c->wbuf, O.full,
					                           cnt_band, c->lcand, c->cost, cmax, c->d_cnt); }
					{ Scope s(c, "twoview_list_scan_kernel");
					  launch_twoview_list_scan(c->stream, c->d_views, ref, oth, W, *p, by, nr, cnt_band, c->lcand, c->cost, cmax); }
				}
				if (c->defer && attempt == 0 && pass == 0 && rows_mode &&
				    ((c->list_cmax_hint > 0 && c->list_smax_hint > 0 && c->views[ref].list_mode[oth] == 1) ||
				     (guessed && c->views[ref].list_mode[oth] == 0))) {
					// optimistic (srh_twoview_compute): capacities and path are those earlier runs of this pair learnt -- or, for a
					// pair's FIRST call, the host's guess and the row-run path --; the maxima
					// and counters travel to pinned memory behind the kernels, the caller verifies both passes with one wait (a
					// pass that does not stand -- a longer list after a re-upload, say -- is redone with the wait per pass)
					HIP_TRY(hipMemcpyAsync(c->defer->host, c->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
					HIP_TRY(hipMemcpyAsync(c->defer->span, c->d_span, 4*sizeof(int), hipMemcpyDeviceToHost, c->stream));
					c->defer->queued = true; c->defer->lists = true; c->defer->strip = false; c->defer->cert = rows_cert;
					c->defer->cmax = cmax; c->defer->smax = smax;
					c->defer->guessed = guessed; c->defer->ref = ref; c->defer->oth = oth;
					if (c->debug_trace) fprintf(stderr, "[srh trace] lists %d>%d queued unverified: cmax %d smax %d guessed %d\n", ref, oth, cmax, smax, (int)guessed);
					c->stats.used_strip_kernel = 0;
					c->stats.used_dense_path = 0;
					HIP_TRY(hipGetLastError());
					return SRH_OK;
				}
				int mx[4] = { 0, 0, 0, 0 };
				HIP_TRY(hipMemcpyAsync(mx, c->d_span, 4*sizeof(int), hipMemcpyDeviceToHost, c->stream));
				HIP_TRY(hipStreamSynchronize(c->stream));
				const int maxc = mx[0];
				if (c->debug_trace) fprintf(stderr, "[srh trace] lists %d>%d verified pass %d: cmax %d smax %d guessed %d -> longest list %d, slots %d, rows over %d\n",
				                            ref, oth, pass, cmax, smax, (int)guessed, mx[0], mx[1], mx[2]);
				if (rows_mode) {
					// a curve crossing more than SRH_ROWS_NR rows, or more slots than the 16-bit slot base
					// holds: this pair is evaluated in list order instead
					const int need = (mx[1] + 7) & ~7;
					if (mx[2] || need > 65528) {
						rows_mode = false; c->views[ref].list_mode[oth] = 2;
						if (maxc > cmax) cmax = (maxc + 7) & ~7;
						continue;
					}
					if (maxc <= cmax && need <= smax) {
						// what later runs of the pair are queued with: the capacities this pass stood on -- or, when they were a
						// guess, what the pass measured (the tight strides a counting pass would have given)
						const int cm = guessed ? std::max(8, (maxc + 7) & ~7) : cmax;
						const int sm = guessed ? std::max(cm + 64, need) : smax;
						if (cm > c->list_cmax_hint) c->list_cmax_hint = cm;
						if (sm > c->list_smax_hint) c->list_smax_hint = sm;
						// short spans (steep curves) fill their 8-column blocks badly: a slot costs ~0.4 of a
						// candidate evaluated in list order, so beyond 2.2 slots per candidate the other path wins
						Counters hc;
						HIP_TRY(hipMemcpyAsync(&hc, c->d_cnt, sizeof(hc), hipMemcpyDeviceToHost, c->stream));
						HIP_TRY(hipStreamSynchronize(c->stream));
						c->views[ref].list_mode[oth] = (hc.n_slots > 2.2*(double)hc.n_listed) ? 2 : 1;
						if (hc.cert_overflow != 0) { cert_ok = false; continue; }   // more flagged pixels than the redo covers: mode 0
						if (rows_cert) { c->stats.n_certified = (int64_t)hc.n_pixels; c->stats.n_flagged = (int64_t)hc.n_flagged; }
						break;
					}
					if (maxc > cmax) { cmax = (maxc + 7) & ~7; if (need <= smax) smax = std::max(smax, cmax + 64); }
					if (need > smax) smax = need;
					guessed = false;                                      // (the repeat runs on measured maxima)
					continue;
				}
				if (maxc <= cmax) {
					const int cm = guessed ? std::max(8, (maxc + 7) & ~7) : cmax;
					if (cm > c->list_cmax_hint) c->list_cmax_hint = cm;
					break;
				}
				cmax = (maxc + 7) & ~7;                               // hint too small: repeat with the true maximum
				guessed = false;
			}
			HIP_TRY(hipGetLastError());
			break;
		}
		const bool tscan = dense && c->tscan != 0;
		if (dense) {
			if ((rc = ensure(c->tnum, c->tnum_cap, (size_t)p->num_depth_levels + 8))) return rc;   // (+ 8: the template scan reads the table in whole eights)
			{ Scope s(c, "pinhole_label_table_kernel");
			  launch_pinhole_label_table(c->stream, c->d_views, ref, *p, false, c->tnum); }
			if (tscan) {
				// the pass's candidate template: the reference's walk at one pixel (srh_dense.hip, "template scan")
				if ((rc = ensure(c->stpl, c->stpl_cap, scan_template_bytes()))) return rc;
				Scope s(c, "twoview_template_kernel");
				launch_scan_template(c->stream, c->d_views, ref, oth, *p, y0, y1 - y0, c->tnum, c->stpl);
			}
		}
		size_t rows = 0;
		for (int pass = 0; pass < 2; ++pass) {
			const int wdoubles = strip ? (2*R + 1)*wimg_wp(R) : T;     // doubles per pixel window in the band buffer
			const size_t per_pixel = (size_t)wdoubles*sizeof(double) + (dense ? (size_t)cstride*sizeof(double) : 0);
			rows = budget / (per_pixel*(size_t)W);
			if (rows < 1) rows = 1;
			if (rows > (size_t)(y1 - y0)) rows = (size_t)(y1 - y0);
			// the strip kernel wants dozens of tiles per persistent workgroup and launch; thin bands take the per-tile kernel
			if (strip && c->strip == 1 && (size_t)((W + SRH_WTILE - 1)/SRH_WTILE)*rows < (size_t)48*2*c->num_cus) strip = false;
			else break;
		}
		const bool wimg = strip;                                       // the band's windows in the LDS image's layout
		const size_t wstride = SRH_WTILE;
		if ((rc = ensure(c->wbuf, c->wbuf_cap, wimg ? wimg_doubles(W, (int)rows, R) : wbuf_doubles(W, (int)rows, T)))) return rc;
		if (dense && (rc = ensure(c->cost, c->cost_cap, rows*(size_t)((W + 31)/32)*32*(size_t)cstride))) return rc;   // 32-pixel tiles
		// (+ one tile of slack: the strip kernel copies whole 32-pixel pieces of these rows)
		if (dense && (rc = ensure(c->pconst, c->pconst_cap, (rows*(size_t)W + SRH_WTILE)*SRH_PC))) return rc;
		int lanes = 8;
		if (dense && (rc = ensure(c->prange, c->prange_cap, rows*(size_t)W + SRH_WTILE))) return rc;
		if (tscan && (rc = ensure(c->tileflag, c->tileflag_cap, rows*(size_t)((W + 63)/64) + 1))) return rc;
		const bool cert = dense && cert_ok;
		// (the strip kernel's certified form: 5 = one sweep over the window, 3 = the reference's two sweeps fused; option "cert_form")
		const int cost_arith = c->arith == 3 ? (cert ? (c->cert_form == 1 ? 5 : 3) : 0) : c->arith;
		if (cert && (rc = ensure(c->cflag, c->cflag_cap, rows*(size_t)W + 1))) return rc;
		const bool planes = dense && (R == 5 || R == 2);
		if (strip) lanes = strip_block_lanes(cstride, c->strip == 1 ? 0 : c->strip);
		if (planes) {
			// NaN-bordered gray_tv planes of both views (strip kernel; the general cost of the left-out columns and of the
			// certified redo on either dense path)
			for (int k = 0; k < 2; ++k) {
				ViewHost &v = c->views[k == 0 ? ref : oth];
				if (!v.tvp) HIP_TRY(plane_alloc((void **)&v.tvp, padded_size(v.w, v.h)*sizeof(double)));
				if (!v.tvp_valid) {
					Scope s(c, "padded_plane_kernel");
					launch_padded_plane(c->stream, v.gray_tv, v.w, v.h, v.tvp);
					v.tvp_valid = true;
				}
			}
		}
		if (strip) {
			// zero-bordered "window fully usable" plane of the other view
			ViewHost &O = c->views[oth];
			if (!O.fullp) HIP_TRY(hipMalloc((void **)&O.fullp, padded_size(O.w, O.h)));
			if (O.fullp_r != R) {
				Scope s(c, "padded_full_kernel");
				launch_padded_full(c->stream, O.gray_tv, O.w, O.h, R, O.fullp);
				O.fullp_r = R;
			}
		}

		for (int by = y0; by < y1; by += (int)rows) {
			if (cancelled(c)) return fail(SRH_E_CANCELLED, "cancelled");
			const int nr = std::min((int)rows, y1 - by);
			run_weights(c, ref, W, *p, by, nr, wstride, dense ? c->pconst : nullptr, wimg);
			if (dense) {
				Scope s(c, "pixel_range_kernel");
				launch_pixel_range(c->stream, c->d_views, ref, oth, W, *p, by, nr, c->tnum, cstride, c->prange);
			}
			if (dense) {
				// the band's cost rows under one arithmetic: strip kernel or one workgroup per tile, then the left-out columns
				auto cost_pass = [&](int arith) -> int {
					if (strip) {
						HIP_TRY(hipMemsetAsync(&c->d_cnt->strip_ticket, 0, 2*sizeof(unsigned int), c->stream));
						{ Scope s(c, "twoview_strip_cost_kernel");
						  launch_twoview_strip_cost(c->stream, c->d_views, ref, oth, W, H, *p, by, nr, c->wbuf, c->pconst, c->prange,
						                            c->views[ref].tvp, c->views[oth].tvp, c->views[oth].fullp, c->cost, cstride,
						                            c->d_cnt, arith, c->num_cus, lanes, c->diag && c->diag->raw); }
						Scope s(c, "twoview_lazy_fill_kernel");
						launch_twoview_lazy_fill(c->stream, c->d_views, ref, oth, W, *p, by, nr, c->prange, c->wbuf, wstride,
						                         c->views[ref].tvp, c->views[oth].tvp, true, lanes, lanes == 8, c->cost, cstride, c->d_cnt);
					} else {
						{ Scope s(c, "twoview_dense_cost_kernel");
						  if (arith == 2)
							launch_twoview_dense_cost_f32(c->stream, c->d_views, ref, oth, W, *p, by, nr, c->wbuf, wstride,
							                              c->tnum, c->cost, cstride, c->d_cnt, c->pconst, c->f32_form);
						  else
							launch_twoview_dense_cost(c->stream, c->d_views, ref, oth, W, *p, by, nr, c->wbuf, wstride,
							                          c->tnum, c->cost, cstride, c->d_cnt, c->pconst, arith, c->prange); }
						Scope s(c, "twoview_lazy_fill_kernel");
						launch_twoview_lazy_fill(c->stream, c->d_views, ref, oth, W, *p, by, nr, c->prange, c->wbuf, wstride,
						                         planes ? c->views[ref].tvp : nullptr, planes ? c->views[oth].tvp : nullptr, false, 8,
						                         arith != 5, c->cost, cstride, c->d_cnt);   // (the one-pass form leaves no column out)
					}
					return SRH_OK;
				};
				if (c->diag) {
					// diagnostic: this band's cost rows under the arithmetic asked for, as the cost kernel leaves them (raw: the
					// certified forms without the in-kernel exact redo, NaN where uncertified), and the pixels' column ranges
					srh_context::Diag &dg = *c->diag;
					dg.cstride = cstride; dg.rows = nr; dg.strip = strip;
					const size_t nd = (size_t)nr*((W + 31)/32)*32*(size_t)cstride;
					if (by != y0 || nr != y1 - y0) return fail(SRH_E_UNSUPPORTED, "cost rows: the rows asked for do not fit one band (%d of %d)", nr, y1 - y0);
					if (dg.cost) {
						if (dg.cost_doubles < nd) return fail(SRH_E_INVALID, "cost rows: buffer of %zu doubles, %zu needed", dg.cost_doubles, nd);
						HIP_TRY(hipMemsetAsync(c->cost, 0xff, nd*sizeof(double), c->stream));   // (never-written entries read as a NaN with payload -1)
						if ((rc = cost_pass(dg.form))) return rc;
						HIP_TRY(hipMemcpyAsync(dg.cost, c->cost, nd*sizeof(double), hipMemcpyDeviceToHost, c->stream));
						if (dg.range) HIP_TRY(hipMemcpyAsync(dg.range, c->prange, (size_t)nr*W*sizeof(PixRange), hipMemcpyDeviceToHost, c->stream));
						HIP_TRY(hipStreamSynchronize(c->stream));
					}
					dg.done = true;
					return SRH_OK;
				}
				if (cert) HIP_TRY(hipMemsetAsync(c->cflag, 0, sizeof(uint32_t), c->stream));
				if ((rc = cost_pass(cost_arith))) return rc;
				{ Scope s(c, "twoview_scan_kernel");
				  launch_twoview_scan(c->stream, c->d_views, ref, oth, W, *p, by, nr, c->tnum, c->cost, cstride, c->d_cnt, c->prange,
				                      cert ? c->cflag : nullptr, -1, cert && strip ? c->pconst : nullptr, tscan ? c->stpl : nullptr, c->tileflag, c->num_cus); }
				if (cert) {
					// the pixels whose decisions the bound does not cover, in the reference's arithmetic: their cost rows are
					// refilled and they are scanned again -- launched for a capacity, the count stays on the device
					const int cap = redo_capacity((size_t)nr*W);
					{ Scope s(c, "twoview_refill_kernel");
					  launch_twoview_refill(c->stream, W, *p, by, c->prange, c->cflag, cap, c->wbuf, wimg, c->views[ref].tvp,
					                        c->views[oth].tvp, c->cost, cstride, c->d_cnt); }
					Scope s(c, "twoview_rescan_kernel");
					launch_twoview_scan(c->stream, c->d_views, ref, oth, W, *p, by, nr, c->tnum, c->cost, cstride, c->d_cnt, c->prange,
					                    c->cflag, cap);
				}
			} else {
				Scope s(c, "twoview_generic_kernel");
				launch_twoview_generic(c->stream, c->d_views, ref, oth, W, *p, by, nr, c->wbuf, wstride, c->d_cnt);
			}
		}
		HIP_TRY(hipGetLastError());
		if (!dense) break;
		if (c->defer && attempt == 0) {
			// optimistic (srh_twoview_compute): the counters travel to pinned memory behind the kernels, the caller looks at them
			HIP_TRY(hipMemcpyAsync(c->defer->host, c->d_cnt, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
			c->defer->queued = true; c->defer->strip = strip; c->defer->cert = cert_ok;
			c->stats.used_strip_kernel = strip ? 1 : 0;
			c->stats.used_dense_path = 1;
			return SRH_OK;
		}
		// the dense result stands only if no candidate left its row / column range
		Counters hc;
		HIP_TRY(hipMemcpyAsync(&hc, c->d_cnt, sizeof(hc), hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		if (strip && hc.strip_overflow != 0) { strip = false; continue; }   // a tile's ranges did not fit one chunk: per-tile kernel
		if (hc.cert_overflow != 0) { cert_ok = false; continue; }           // more flagged pixels than the redo covers: mode 0
		c->stats.n_certified = (int64_t)hc.n_certified; c->stats.n_flagged = (int64_t)hc.n_flagged;
		if (hc.not_row_aligned == 0) break;
		dense = false;                                              // redo with the general kernel
		strip = false;
	}
	c->stats.used_strip_kernel = strip ? 1 : 0;
	c->stats.used_dense_path = dense ? 1 : 0;
	return SRH_OK;
}

// Diagnostic (tests/test_gpu_cert_rows.py): the cost rows of rows [y0, y1) on the row-aligned dense plan, straight from the
// cost kernel -- what the scan would look up.  form: 0 the reference's arithmetic, 3 two fused sweeps, 5 one-pass (the
// certified forms); raw != 0: without the in-kernel exact redo (uncertified candidates are NaN).
extern "C" int srh_twoview_cost_rows(srh_context *c, int ref, int oth, const srh_params *p, int y0, int y1, int form, int raw,
                                     double *cost_out, size_t cost_doubles, int32_t *range_out, int *cstride_out, int *used_strip)
{
	int rc;
	if ((rc = check_slot(c, ref, true)) || (rc = check_slot(c, oth, true)) || (rc = check_params(p))) return rc;
	if (form != 0 && form != 3 && form != 5 && form != 1) return fail(SRH_E_INVALID, "form must be 0, 1, 3 or 5");
	if ((form == 3 || form == 5) && !cert_bound(*p).ok) return fail(SRH_E_UNSUPPORTED, "the parameters leave the error bound no room");
	srh_context::Diag dg;
	dg.form = form; dg.raw = raw != 0; dg.cost = cost_out; dg.cost_doubles = cost_doubles; dg.range = range_out;
	c->diag = &dg;
	const int keep_arith = c->arith; const bool keep_fused = c->use_fused;
	c->arith = form == 0 ? 0 : 3; c->use_fused = false;
	rc = twoview_wta_run(c, ref, oth, p, y0, y1);
	c->arith = keep_arith; c->use_fused = keep_fused;
	c->diag = nullptr;
	if (rc) return rc;
	if (!dg.done) return fail(SRH_E_UNSUPPORTED, "cost rows exist on the row-aligned dense plan only (radius 5 or 2, rectified pinhole pair)");
	if (cstride_out) *cstride_out = dg.cstride;
	if (used_strip) *used_strip = dg.strip ? 1 : 0;
	return SRH_OK;
}

extern "C" int srh_debug_exp(srh_context *c, const double *x, int n, double *kernel_out, double *library_out) {
	if (!c || !x || !kernel_out || !library_out || n < 1) return fail(SRH_E_INVALID, "null argument");
	HIP_TRY(hipSetDevice(c->device));
	double *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, (size_t)n*3*sizeof(double)));
	hipError_t e = hipMemcpyAsync(d, x, (size_t)n*sizeof(double), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) {
		launch_geo_exp_probe(c->stream, d, n, d + n, d + 2*(size_t)n);
		e = hipMemcpyAsync(kernel_out, d + n, (size_t)n*sizeof(double), hipMemcpyDeviceToHost, c->stream);
	}
	if (e == hipSuccess) e = hipMemcpyAsync(library_out, d + 2*(size_t)n, (size_t)n*sizeof(double), hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(SRH_E_DEVICE, "srh_debug_exp: %s", hipGetErrorString(e));
	return SRH_OK;
}

extern "C" int srh_twoview_cross_check(srh_context *c, int left, int right, const srh_params *p) {
	int rc;
	if ((rc = check_slot(c, left, true)) || (rc = check_slot(c, right, true)) || (rc = check_params(p))) return rc;
	const ViewHost &L = c->views[left], &Rv = c->views[right];
	if (L.w != Rv.w || L.h != Rv.h) return fail(SRH_E_INVALID, "TwoViewStereo needs equal-sized views");
	HIP_TRY(hipSetDevice(c->device));
	// left pass completes (stream order) before the right pass reads the filtered left map
	{ Scope s(c, "twoview_cross_check_kernel"); launch_twoview_cross_check(c->stream, c->d_views, left, right, L.w, L.h, *p); }
	{ Scope s(c, "twoview_cross_check_kernel"); launch_twoview_cross_check(c->stream, c->d_views, right, left, L.w, L.h, *p); }
	HIP_TRY(hipGetLastError());
	return SRH_OK;
}

// the context's stream and TwoView band buffers <-> the second pass's
static void tv_slot_swap(srh_context *c) {
	srh_context::TvSlot &T = c->tv_slot;
	std::swap(c->stream, T.stream);
	std::swap(c->d_cnt, T.d_cnt); std::swap(c->d_span, T.d_span);
	std::swap(c->wbuf, T.wbuf); std::swap(c->wbuf_cap, T.wbuf_cap);
	std::swap(c->cost, T.cost); std::swap(c->cost_cap, T.cost_cap);
	std::swap(c->tnum, T.tnum); std::swap(c->tnum_cap, T.tnum_cap);
	std::swap(c->pconst, T.pconst); std::swap(c->pconst_cap, T.pconst_cap);
	std::swap(c->prange, T.prange); std::swap(c->prange_cap, T.prange_cap);
	std::swap(c->cflag, T.cflag); std::swap(c->cflag_cap, T.cflag_cap);
	std::swap(c->lcand, T.lcand); std::swap(c->lcand_cap, T.lcand_cap);
	std::swap(c->lrowinfo, T.lrowinfo); std::swap(c->lrowinfo_cap, T.lrowinfo_cap);
	std::swap(c->lcount, T.lcount); std::swap(c->lcount_cap, T.lcount_cap);
	std::swap(c->lmeta, T.lmeta); std::swap(c->lmeta_cap, T.lmeta_cap);
	std::swap(c->stpl, T.stpl); std::swap(c->stpl_cap, T.stpl_cap);
	std::swap(c->tileflag, T.tileflag); std::swap(c->tileflag_cap, T.tileflag_cap);
}

static bool tv_pass_stands(const srh_context::TvDefer &d) {
	const Counters &h = *d.host;
	if (d.lists)                                                   // every list and every pixel's cost slots fitted, no curve over too many rows
		return d.span[0] <= d.cmax && ((d.span[1] + 7) & ~7) <= d.smax && d.span[2] == 0 && h.cert_overflow == 0;
	return !(d.strip && h.strip_overflow != 0) && h.cert_overflow == 0 && h.not_row_aligned == 0;
}

extern "C" int srh_twoview_compute(srh_context *c, int left, int right, const srh_params *p,
                                   double *left_out, double *right_out)
{
	int rc;
	if ((rc = check_slot(c, left, true)) || (rc = check_slot(c, right, true)) || (rc = check_params(p))) return rc;
	HIP_TRY(hipSetDevice(c->device));
	for (auto &d : c->tv_defer) {
		if (!d.host) HIP_TRY(hipHostMalloc((void **)&d.host, sizeof(Counters)));
		if (!d.span) HIP_TRY(hipHostMalloc((void **)&d.span, 4*sizeof(int)));
		d.queued = false; d.lists = false; d.guessed = false;
	}
	if (c->tv_overlap) {
		srh_context::TvSlot &T = c->tv_slot;
		if (!T.stream) {
			HIP_TRY(hipStreamCreateWithFlags(&T.stream, hipStreamNonBlocking));
			HIP_TRY(hipEventCreateWithFlags(&T.go, hipEventDisableTiming));
			HIP_TRY(hipEventCreateWithFlags(&T.done, hipEventDisableTiming));
			HIP_TRY(hipMalloc((void **)&T.d_cnt, sizeof(Counters)));
			HIP_TRY(hipMalloc((void **)&T.d_span, 4*sizeof(int)));
		}
		// the NaN-bordered planes both passes read (twoview_wta_run makes them on demand, on its own stream) exist at `go`
		if (p->window_radius == 5 || p->window_radius == 2)
			for (int k = 0; k < 2; ++k) {
				ViewHost &v = c->views[k == 0 ? left : right];
				if (!v.tvp) HIP_TRY(plane_alloc((void **)&v.tvp, padded_size(v.w, v.h)*sizeof(double)));
				if (!v.tvp_valid) {
					Scope s(c, "padded_plane_kernel");
					launch_padded_plane(c->stream, v.gray_tv, v.w, v.h, v.tvp);
					v.tvp_valid = true;
				}
			}
		HIP_TRY(hipEventRecord(T.go, c->stream));
	}
	// progress steps as TwoViewStereo emits them (twoviewstereo.cpp:234,405,597,225)
	progress(c, 1, "Computing cost volume for left image...");
	c->defer = &c->tv_defer[0];
	rc = srh_twoview_wta(c, left, right, p, 0, 0);
	c->defer = nullptr;
	if (rc) return rc;
	if (cancelled(c)) return fail(SRH_E_CANCELLED, "cancelled");
	progress(c, 3, "Computing cost volume for right image...");
	if (c->tv_overlap && c->tv_defer[0].queued) {
		// the first pass is queued and unverified (the dense plan): the second one beside it, from the point `go` of the
		// context's stream (before the first pass: the views and their padded planes are complete there)
		srh_context::TvSlot &T = c->tv_slot;
		HIP_TRY(hipStreamWaitEvent(T.stream, T.go, 0));
		tv_slot_swap(c);
		c->defer = &c->tv_defer[1];
		rc = srh_twoview_wta(c, right, left, p, 0, 0);
		c->defer = nullptr;
		const hipError_t e = rc ? hipStreamSynchronize(c->stream) : hipEventRecord(T.done, c->stream);   // (c->stream: the slot's, still)
		tv_slot_swap(c);
		if (rc) return rc;
		if (e != hipSuccess) return fail(SRH_E_DEVICE, "second TwoView pass: %s", hipGetErrorString(e));
		HIP_TRY(hipStreamWaitEvent(c->stream, T.done, 0));            // whatever follows on the context's stream sees both maps
		// (srh_get_stats reads the context's counters and reports the last pass's, as it does without the overlap)
		HIP_TRY(hipMemcpyAsync(c->d_cnt, T.d_cnt, sizeof(Counters), hipMemcpyDeviceToDevice, c->stream));
	} else {
		c->defer = &c->tv_defer[1];
		rc = srh_twoview_wta(c, right, left, p, 0, 0);
		c->defer = nullptr;
		if (rc) return rc;
	}
	if (cancelled(c)) return fail(SRH_E_CANCELLED, "cancelled");
	srh_context::TvDefer &d0 = c->tv_defer[0], &d1 = c->tv_defer[1];
	if (d0.queued && d1.queued) {
		// both passes took the dense plan and are still unverified: queue the cross-check and the hand-over behind them, wait once
		progress(c, 5, "Detecting inconsistencies...");
		if ((rc = srh_twoview_cross_check(c, left, right, p))) return rc;
		const ViewHost &L = c->views[left], &Rv = c->views[right];
		if (left_out) HIP_TRY(hipMemcpyAsync(left_out, L.depth, (size_t)L.w*L.h*sizeof(double), hipMemcpyDeviceToHost, c->stream));
		if (right_out) HIP_TRY(hipMemcpyAsync(right_out, Rv.depth, (size_t)Rv.w*Rv.h*sizeof(double), hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		if (c->debug_trace)
			for (srh_context::TvDefer *d : { &d0, &d1 })
				fprintf(stderr, "[srh trace] unverified pass %d>%d: lists %d cmax %d smax %d guessed %d -> longest %d slots %d rows over %d cert_overflow %llu stands %d\n",
				        d->ref, d->oth, (int)d->lists, d->cmax, d->smax, (int)d->guessed, d->span[0], d->span[1], d->span[2],
				        (unsigned long long)d->host->cert_overflow, (int)tv_pass_stands(*d));
		if (tv_pass_stands(d0) && tv_pass_stands(d1)) {
			// a first call on guessed capacities: what the passes measured is what later calls are queued with (the tight
			// strides a verified pass would have recorded), and how the pair's lists are best evaluated
			for (srh_context::TvDefer *d : { &d0, &d1 })
				if (d->lists && d->guessed) {
					const int cm = std::max(8, (d->span[0] + 7) & ~7), sm = std::max(cm + 64, (d->span[1] + 7) & ~7);
					if (cm > c->list_cmax_hint) c->list_cmax_hint = cm;
					if (sm > c->list_smax_hint) c->list_smax_hint = sm;
					c->views[d->ref].list_mode[d->oth] = (d->host->n_slots > 2.2*(double)d->host->n_listed) ? 2 : 1;
					d->guessed = false;
				}
			// (every counter of srh_stats is the LAST pass's -- right -> left -- on this path as on the verified ones and in
			// srh_get_stats, which reads the context's device counters: the second pass's were copied there above)
			c->stats.n_pixels = (int64_t)d1.host->n_pixels;
			c->stats.n_eval = (int64_t)d1.host->n_eval;
			c->stats.n_eval_device = (int64_t)d1.host->n_eval_device;
			// (the row-run certified scan counts the pixels it scanned in n_pixels: what the verified list path reports as well)
			c->stats.n_certified = d1.lists ? (d1.cert ? (int64_t)d1.host->n_pixels : 0) : (int64_t)d1.host->n_certified;
			c->stats.n_flagged = (int64_t)d1.host->n_flagged;
			c->stats.used_dense_path = d1.lists ? 0 : 1;
			c->stats.used_fused_kernel = c->last_fused ? 1 : 0;
			progress(c, 8, "Finished!");
			return SRH_OK;
		}
		// a plan was refuted on the device (a curve off its row, a range wider than a chunk, more flagged pixels than the
		// redo covers): everything once more, each pass verified before the next step -- the maps are rewritten whole
		d0.queued = d1.queued = false;
		if ((rc = srh_twoview_wta(c, left, right, p, 0, 0))) return rc;
		if ((rc = srh_twoview_wta(c, right, left, p, 0, 0))) return rc;
	} else {
		// one pass (or none) was left unverified: settle it now, before anything reads its map
		for (int k = 0; k < 2; ++k) {
			srh_context::TvDefer &d = c->tv_defer[k];
			if (!d.queued) continue;
			HIP_TRY(hipStreamSynchronize(c->stream));
			d.queued = false;
			if (!tv_pass_stands(d) && (rc = (k == 0 ? srh_twoview_wta(c, left, right, p, 0, 0) : srh_twoview_wta(c, right, left, p, 0, 0)))) return rc;
		}
	}
	// (counters of the two passes for the statistics: an untimed recount is not worth a pass; the last pass's are reported)
	if ((rc = fetch_counters(c, c->stats.used_dense_path))) return rc;
	progress(c, 5, "Detecting inconsistencies...");
	if ((rc = srh_twoview_cross_check(c, left, right, p))) return rc;
	if (left_out && (rc = srh_view_depth_download(c, left, left_out))) return rc;
	if (right_out && (rc = srh_view_depth_download(c, right, right_out))) return rc;
	if ((rc = srh_synchronize(c))) return rc;
	progress(c, 8, "Finished!");
	return SRH_OK;
}

extern "C" int srh_mvs_initial_estimate(srh_context *c, int view, const int32_t *neigh, int nneigh,
                                        const srh_params *p, int y0, int y1, void *peaks_dev);

// The serpentine list of a view's masked-in pixels (mask.pixel == WHITE <=> byte 1; no mask: every pixel), built when the
// MultiViewStereo list path first needs it after an upload.  The device buffer keeps its capacity across uploads.
static int ensure_act(srh_context *c, ViewHost &v) {
	if (v.act_valid) return SRH_OK;
	const int w = v.w, h = v.h;
	std::vector<uint32_t> &act = v.act_host;
	act.clear();
	act.reserve((size_t)w*h);
	v.act_row.assign((size_t)h + 1, 0u);
	for (int y = 0; y < h; ++y) {
		v.act_row[y] = (uint32_t)act.size();
		const uint8_t *mr = v.hmask.empty() ? nullptr : v.hmask.data() + (size_t)y*w;
		if (!(y & 1)) { for (int x = 0; x < w; ++x) if (!mr || mr[x] == 1) act.push_back((uint32_t)((size_t)y*w + x)); }
		else          { for (int x = w - 1; x >= 0; --x) if (!mr || mr[x] == 1) act.push_back((uint32_t)((size_t)y*w + x)); }
	}
	v.act_row[h] = (uint32_t)act.size();
	if (!act.empty()) {
		if (v.act_cap < act.size()) {
			// (other streams may still read the old list: estimates of this view queued before the re-upload were settled by it)
			if (v.act) { HIP_TRY(hipFree(v.act)); v.act = nullptr; v.act_cap = 0; }
			HIP_TRY(hipMalloc((void **)&v.act, (size_t)w*h*sizeof(uint32_t)));
			v.act_cap = (size_t)w*h;
		}
		HIP_TRY(hipMemcpyAsync(v.act, act.data(), act.size()*sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));                  // once per upload: either slot's stream may read it next
	}
	v.act_valid = true;
	return SRH_OK;
}

// One pass of the list path of srh_mvs_initial_estimate for list capacity cmax: everything is queued on c->stream with
// c's band buffers, nothing is waited for.  d_span[0] receives the longest list (slots), for the caller to compare with cmax.
static int mvs_list_launch(srh_context *c, int view, const int32_t *neigh, int nneigh, const srh_params *p, int y0, int y1,
                           void *peaks_dev, int cmax)
{
	int rc;
	if ((rc = ensure_act(c, c->views[view]))) return rc;
	const ViewHost &A = c->views[view];
	const int W = A.w;
	const int T = (2*p->window_radius + 1)*(2*p->window_radius + 1);
	const size_t wstride = SRH_WTILE;
	// without a refractive interface every ray of the view starts at the camera centre: the per-label
	// part of pointFromDepth is tabulated once (same operands and operations, see srh_walk.hpp)
	const bool table = !A.cam.is_refractive;
	bool neigh_pinhole = true;                                  // (certified label projections: plain pinhole neighbours only)
	for (int i = 0; i < nneigh; ++i) { const srh_camera &nc = c->views[neigh[i]].cam; if (nc.is_refractive || nc.is_distorted) neigh_pinhole = false; }
	if (table) {
		if ((rc = ensure(c->tnum, c->tnum_cap, (size_t)p->num_depth_levels))) return rc;
		Scope s(c, "pinhole_label_table_kernel");
		launch_pinhole_label_table(c->stream, c->d_views, view, *p, true, c->tnum);
	}
		HIP_TRY(hipMemsetAsync(c->d_cnt, 0, sizeof(Counters), c->stream));
		HIP_TRY(hipMemsetAsync(c->d_span, 0, 4*sizeof(int), c->stream));
		const size_t per_px = (size_t)T*sizeof(double) + (size_t)nneigh*((size_t)cmax*sizeof(uint32_t) + sizeof(int32_t) + 2*sizeof(double)
		                      + (peaks_dev ? (size_t)p->top_k*2*sizeof(double) : 0));
		size_t lrows = band_budget(c) / (per_px*(size_t)W);
		if (lrows < 1) lrows = 1;
		{
			// the budget is a target, not a limit: no sliver band for a few rows over it, and bands of equal height
			const size_t rows = (size_t)(y1 - y0);
			if (rows <= lrows + lrows/4) lrows = rows;
			else { const size_t nb = (rows + lrows - 1)/lrows; lrows = (rows + nb - 1)/nb; }
		}
		const size_t units = lrows*W*(size_t)nneigh;
		// the list kernels' units: the masked-in pixels of a band, padded to whole 128-pixel blocks per link
		size_t lunits = 0;
		for (int by = y0; by < y1; by += (int)lrows) {
			const int nr = std::min((int)lrows, y1 - by);
			const size_t na = (size_t)A.act_row[by + nr] - A.act_row[by];
			lunits = std::max(lunits, ((na + 127) & ~(size_t)127)*(size_t)nneigh);
		}
		if ((rc = ensure(c->wbuf, c->wbuf_cap, wbuf_doubles(W, (int)lrows, T)))) return rc;
		if ((rc = ensure(c->cost, c->cost_cap, units*2 + (peaks_dev ? units*(size_t)p->top_k*2 : 0)))) return rc;   // best pairs + per-unit top-K, by pixel
		if ((rc = ensure(c->lcand, c->lcand_cap, std::max<size_t>(lunits, 128)*(size_t)cmax))) return rc;   // wave-tiled lists
		if ((rc = ensure(c->lcount, c->lcount_cap, std::max<size_t>(lunits, 128)))) return rc;
		const bool staged = c->mvs_staged != 0;
		if (staged) {
			int maxw; size_t words;
			mvs_staging_shape(&maxw, &words);
			const size_t waves = std::max<size_t>(lunits, 128)/64;
			if ((rc = ensure(c->mvs_wdesc, c->mvs_wdesc_cap, waves*words))) return rc;
			if ((rc = ensure(c->mvs_nwin, c->mvs_nwin_cap, waves))) return rc;
		}
		for (int by = y0; by < y1; by += (int)lrows) {
			if (cancelled(c)) return fail(SRH_E_CANCELLED, "cancelled");
			const int nr = std::min((int)lrows, y1 - by);
			const uint32_t *act = A.act ? A.act + A.act_row[by] : nullptr;
			const int nact = (int)(A.act_row[by + nr] - A.act_row[by]);
			run_weights(c, view, W, *p, by, nr, wstride);
			{ Scope s(c, "mvs_walk_kernel");
			  launch_mvs_walk(c->stream, c->d_views, view, neigh, nneigh, W, *p, by, nr, table ? c->tnum : nullptr, c->lcand, cmax, c->lcount,
			                  c->d_cnt, c->d_span, staged ? c->mvs_wdesc : nullptr, staged ? c->mvs_nwin : nullptr, act, nact,
			                  peaks_dev != nullptr, neigh_pinhole && c->arith == 3); }
			double *const upk = peaks_dev ? c->cost + units*2 : nullptr;
			if (staged) {
				Scope s(c, "mvs_staged_cost_kernel");
				launch_mvs_staged_cost(c->stream, c->d_views, view, neigh, nneigh, W, *p, by, nr, c->wbuf, wstride,
				                       c->lcand, cmax, c->lcount, c->cost, c->mvs_wdesc, c->mvs_nwin, c->d_cnt, act, nact, upk,
				                       c->arith == 3 && !peaks_dev);
			}
			{ Scope s(c, "mvs_list_cost_kernel");
			  launch_mvs_list_cost(c->stream, c->d_views, view, neigh, nneigh, W, *p, by, nr, c->wbuf, wstride,
			                       c->lcand, cmax, c->lcount, c->cost, upk, peaks_dev != nullptr, staged ? c->mvs_nwin : nullptr, act, nact); }
			{ Scope s(c, "mvs_combine_kernel");
			  launch_mvs_combine(c->stream, c->d_views, view, nneigh, W, *p, by, nr, c->cost, upk, (double *)peaks_dev); }
		}
	HIP_TRY(hipGetLastError());
	return SRH_OK;
}

#ifdef SRH_PROFILE_PHASES
static void mvs_print_phases(srh_context *c) {
	Counters h;
	if (hipMemcpy(&h, c->d_cnt, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
	if (h.dbg_waves)
		fprintf(stderr, "[srh prof] staged MVS cost: %llu waves, %.1f windows and %.1f slots per wave; cycles per wave: life %.0f = "
		        "set-up %.0f + copies %.0f + slots %.0f (%.0f per slot) + end %.0f\n", h.dbg_waves, (double)h.dbg_cycles/h.dbg_waves,
		        (double)h.dbg_blocks/h.dbg_waves, (double)h.dbg_total_cycles/h.dbg_waves, (double)h.dbg_phase[0]/h.dbg_waves,
		        (double)h.dbg_phase[1]/h.dbg_waves, (double)h.dbg_phase[2]/h.dbg_waves,
		        h.dbg_blocks ? (double)h.dbg_phase[2]/h.dbg_blocks : 0.0, (double)h.dbg_phase[3]/h.dbg_waves);
}
#endif

// ---- two estimates in flight ------------------------------------------------------------------------------------
// The kernels of one view end in tails (the last, longest waves of the walk and cost kernels) and the call used to end
// in a host synchronisation for the list-capacity check.  The default list path now only QUEUES a view's kernels, on
// one of two side streams in turn (slot 1 with its own band buffers), so that the next view's kernels fill the
// tails of this one's; the capacity check is made when the slot is used again or when any other entry point touches
// the context (check_slot), and a view whose lists were cut is then redone.  Results are complete whenever a caller
// can observe them.
static void mvs_slot_swap(srh_context *c, MvsSlot &S) {
	std::swap(c->stream, S.stream);
	if (!S.own_buffers) return;
	std::swap(c->wbuf, S.wbuf); std::swap(c->wbuf_cap, S.wbuf_cap);
	std::swap(c->cost, S.cost); std::swap(c->cost_cap, S.cost_cap);
	std::swap(c->tnum, S.tnum); std::swap(c->tnum_cap, S.tnum_cap);
	std::swap(c->lcount, S.lcount); std::swap(c->lcount_cap, S.lcount_cap);
	std::swap(c->lcand, S.lcand); std::swap(c->lcand_cap, S.lcand_cap);
	std::swap(c->mvs_wdesc, S.mvs_wdesc); std::swap(c->mvs_wdesc_cap, S.mvs_wdesc_cap);
	std::swap(c->mvs_nwin, S.mvs_nwin); std::swap(c->mvs_nwin_cap, S.mvs_nwin_cap);
	std::swap(c->d_cnt, S.d_cnt); std::swap(c->d_span, S.d_span);
}

static int mvs_settle_slot(srh_context *c, int k) {
	MvsSlot &S = c->mvs_slot[k];
	if (!S.pending) return SRH_OK;
	HIP_TRY(hipStreamSynchronize(S.stream));
	S.pending = false;
	if (k == c->mvs_last && S.own_buffers)                       // srh_get_stats reads the context's counters, on its stream
		HIP_TRY(hipMemcpyAsync(c->d_cnt, S.d_cnt, sizeof(Counters), hipMemcpyDeviceToDevice, c->stream));
#ifdef SRH_PROFILE_PHASES
	mvs_slot_swap(c, S); mvs_print_phases(c); mvs_slot_swap(c, S);
#endif
	const int maxc = *S.h_maxc;
	if (c->debug_trace) fprintf(stderr, "[srh trace] mvs view %d settled: capacity %d, longest list %d%s\n", S.view, S.cmax, maxc, maxc <= S.cmax ? "" : " -> redone");
	if (maxc <= S.cmax) { if (S.cmax > c->mvs_cmax_hint) c->mvs_cmax_hint = S.cmax; return SRH_OK; }
	// a list was cut: redo the view, waiting for it, with the true maximum as the capacity (on the context's own
	// buffers, which slot 0 shares: nothing may be in flight)
	for (MvsSlot &O : c->mvs_slot) if (O.stream) HIP_TRY(hipStreamSynchronize(O.stream));
	c->mvs_cmax_hint = std::max(c->mvs_cmax_hint, (maxc + 7) & ~7);
	c->in_settle = true;
	const int rc = srh_mvs_initial_estimate(c, S.view, S.neigh, S.nneigh, &S.p, S.y0, S.y1, nullptr);
	c->in_settle = false;
	return rc;
}

static int mvs_settle_all(srh_context *c) {
	if (c->in_settle) return SRH_OK;
	int rc;
	for (int k = 0; k < SRH_MVS_SLOTS; ++k) if ((rc = mvs_settle_slot(c, k))) return rc;
	return SRH_OK;
}

static int mvs_initial_estimate_run(srh_context *c, int view, const int32_t *neigh, int nneigh,
                                    const srh_params *p, int y0, int y1, void *peaks_dev);

extern "C" int srh_mvs_initial_estimate(srh_context *c, int view, const int32_t *neigh, int nneigh,
                                        const srh_params *p, int y0, int y1, void *peaks_dev)
{
	if (!c) return fail(SRH_E_INVALID, "null context");
	if (c->in_settle) return mvs_initial_estimate_run(c, view, neigh, nneigh, p, y0, y1, peaks_dev);   // (the redo of a settle: its caller retries)
	return with_thinner_bands(c, [&] { return mvs_initial_estimate_run(c, view, neigh, nneigh, p, y0, y1, peaks_dev); });
}

static int mvs_initial_estimate_run(srh_context *c, int view, const int32_t *neigh, int nneigh,
                                    const srh_params *p, int y0, int y1, void *peaks_dev)
{
	int rc;
	if ((rc = check_slot(c, view, true, false)) || (rc = check_params(p))) return rc;
	if (nneigh < 0 || nneigh > SRH_MAX_NEIGH)
		return fail(SRH_E_UNSUPPORTED, "nneigh %d outside [0,%d] (the reference's NUM_NEIGHBOURING_VIEWS is 3)", nneigh, SRH_MAX_NEIGH);
	if (nneigh > 0 && !neigh) return fail(SRH_E_INVALID, "null neighbour list");
	if (peaks_dev && p->top_k < 1) return fail(SRH_E_INVALID, "top_k < 1");
	for (int i = 0; i < nneigh; ++i) {
		if ((rc = check_slot(c, neigh[i], true, false))) return rc;
		if (neigh[i] == view) return fail(SRH_E_INVALID, "view %d listed as its own neighbour", view);
	}
	HIP_TRY(hipSetDevice(c->device));
	const ViewHost &A = c->views[view];
	const int W = A.w, H = A.h;
	if (y0 < 0) y0 = 0;
	if (y1 <= 0 || y1 > H) y1 = H;
	const int T = (2*p->window_radius + 1)*(2*p->window_radius + 1);
	if (y1 <= y0) return SRH_OK;

	const size_t wstride = SRH_WTILE;
	if (y1 <= y0) return SRH_OK;

	// ---- default: walk kernel -> candidate lists -> cost kernel -> maximum (and merged top-K lists) over the
	// neighbours.  Other radii stay on the one-thread-per-pixel kernels.
	if (!c->force_generic && nneigh > 0 && p->window_radius == 2 && W < 65536 && H < 65536) {
		// (no hint yet: a guess from the geometry -- the longest coarse curve over the neighbours, estimate_list_capacity;
		// round 5 started from 2 D, which on the C4 rig cut every view's lists once: its first runTask took twice a later one)
		int cmax = c->mvs_cmax_hint;
		if (cmax <= 0) {
			cmax = 64;
			for (int i = 0; i < nneigh; ++i) {
				int cm = 0, sm = 0;
				const ViewHost &N = c->views[neigh[i]];
				estimate_list_capacity(c->views[view].cam, N.cam, W, H, N.w, N.h, *p, y0, y1, cm, sm, true);
				cmax = std::max(cmax, cm);
			}
		}
		if (c->mvs_async && !peaks_dev && !c->in_settle) {
			// queue the view on the next slot and return
			const int k = c->mvs_turn;
			c->mvs_turn = (c->mvs_turn + 1) % SRH_MVS_SLOTS;
			MvsSlot &S = c->mvs_slot[k];
			if ((rc = mvs_settle_slot(c, k))) return rc;
			cmax = std::max(cmax, c->mvs_cmax_hint);
			if (!S.stream) {
				HIP_TRY(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
				HIP_TRY(hipEventCreateWithFlags(&S.ev, hipEventDisableTiming));
				HIP_TRY(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
				HIP_TRY(hipHostMalloc((void **)&S.h_maxc, sizeof(int)));
				*S.h_maxc = 0;
				S.own_buffers = k >= 1;
				if (S.own_buffers) {
					HIP_TRY(hipMalloc((void **)&S.d_cnt, sizeof(Counters)));
					HIP_TRY(hipMalloc((void **)&S.d_span, 4*sizeof(int)));
				}
			}
			HIP_TRY(hipEventRecord(S.ev, c->stream));                 // after whatever the caller's stream holds
			HIP_TRY(hipStreamWaitEvent(S.stream, S.ev, 0));
			*S.h_maxc = 0;                                             // (a queue that fails half-way must not leave a stale length behind)
			mvs_slot_swap(c, S);
			rc = mvs_list_launch(c, view, neigh, nneigh, p, y0, y1, nullptr, cmax);
			if (!rc && hipMemcpyAsync(S.h_maxc, c->d_span, sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess)
				rc = fail(SRH_E_DEVICE, "queueing the list-length read-back failed");
			mvs_slot_swap(c, S);
			if (rc) {
				// nothing of this view counts: drain what was queued, the slot is free again
				(void)hipStreamSynchronize(S.stream);
				S.pending = false;
				return rc;
			}
			// A caller-owned stream (srh_set_stream) keeps its contract: work the caller enqueues on it after this call
			// is ordered after the estimate.  (The context's own stream needs no such edge: every entry point settles.)
			if (c->stream != c->own_stream) {
				HIP_TRY(hipEventRecord(S.done, S.stream));
				HIP_TRY(hipStreamWaitEvent(c->stream, S.done, 0));
			}
			S.pending = true;
			S.view = view; S.nneigh = nneigh; S.y0 = y0; S.y1 = y1; S.cmax = cmax; S.p = *p;
			for (int i = 0; i < nneigh; ++i) S.neigh[i] = neigh[i];
			c->mvs_last = k;
			c->stats.used_dense_path = 0;
			return SRH_OK;
		}
		if ((rc = mvs_settle_all(c))) return rc;
		for (int pass = 0; pass < 3; ++pass) {
			if ((rc = mvs_list_launch(c, view, neigh, nneigh, p, y0, y1, peaks_dev, cmax))) return rc;
			int maxc = 0;
			HIP_TRY(hipMemcpyAsync(&maxc, c->d_span, sizeof(int), hipMemcpyDeviceToHost, c->stream));
			HIP_TRY(hipStreamSynchronize(c->stream));
#ifdef SRH_PROFILE_PHASES
			mvs_print_phases(c);
#endif
			if (maxc <= cmax) { if (cmax > c->mvs_cmax_hint) c->mvs_cmax_hint = cmax; break; }
			cmax = (maxc + 7) & ~7;                                   // a list was cut: repeat with the true maximum
		}
		c->mvs_last = -1;
		HIP_TRY(hipGetLastError());
		c->stats.used_dense_path = 0;
		return SRH_OK;
	}
	if ((rc = mvs_settle_all(c))) return rc;

	const int rows = band_rows(c, W, H, T);
	if ((rc = ensure(c->wbuf, c->wbuf_cap, wbuf_doubles(W, rows, T)))) return rc;
	if ((rc = ensure(c->cost, c->cost_cap, (size_t)rows*W*2*(size_t)std::max(nneigh, 1)))) return rc;   // per-neighbour best (cost, depth)
	HIP_TRY(hipMemsetAsync(c->d_cnt, 0, sizeof(Counters), c->stream));
	for (int by = y0; by < y1; by += rows) {
		if (cancelled(c)) return fail(SRH_E_CANCELLED, "cancelled");
		const int nr = std::min(rows, y1 - by);
		run_weights(c, view, W, *p, by, nr, wstride);
		{ Scope s(c, "mvs_generic_kernel");
		  launch_mvs_generic(c->stream, c->d_views, view, neigh, nneigh, W, *p, by, nr, c->wbuf, wstride,
		                     (double *)peaks_dev, c->force_generic ? nullptr : c->cost, c->d_cnt); }
	}
	HIP_TRY(hipGetLastError());
	c->stats.used_dense_path = 0;
	return SRH_OK;
}

extern "C" int srh_mvs_cross_check(srh_context *c, const int32_t *slots, int nviews, int view_index,
                                   const srh_params *p)
{
	int rc;
	if (!c) return fail(SRH_E_INVALID, "null context");
	if ((rc = check_params(p))) return rc;
	if (!slots || nviews < 1 || nviews > SRH_MAX_VIEWS) return fail(SRH_E_INVALID, "bad view list");
	if (view_index < 0 || view_index >= nviews) return fail(SRH_E_INVALID, "view_index %d outside [0,%d)", view_index, nviews);
	for (int i = 0; i < nviews; ++i) if ((rc = check_slot(c, slots[i], true))) return rc;
	HIP_TRY(hipSetDevice(c->device));
	// the view list travels once per run: MultiViewStereo::crossCheck is called per view with the same list, and a copy per
	// call (from caller memory: with a host wait) put a copy, two dispatch gaps and a host round trip between the kernels
	if (c->slots_n != nviews || memcmp(c->slots_host, slots, sizeof(int32_t)*nviews) != 0) {
		c->slots_n = 0;                                             // (a failed copy must not leave a cache entry behind)
		memcpy(c->slots_host, slots, sizeof(int32_t)*nviews);
		HIP_TRY(hipMemcpyAsync(c->d_slots, c->slots_host, sizeof(int32_t)*nviews, hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));                   // (the context's copy may change with the next call)
		c->slots_n = nviews;
	}
	const ViewHost &A = c->views[slots[view_index]];
	{ Scope s(c, "mvs_cross_check_kernel");
	  launch_mvs_cross_check(c->stream, c->d_views, c->d_slots, nviews, view_index, A.w, A.h, *p); }
	HIP_TRY(hipGetLastError());
	return SRH_OK;
}

// ------------------------------------------------------------------ depth map -> point cloud
extern "C" int srh_view_point_cloud(srh_context *c, int slot, const srh_params *p, double *xyz_out, uint8_t *rgb_out,
                                    uint8_t *valid_out, int64_t *n_points, int64_t *n_masked, int64_t *n_finite)
{
	int rc;
	if ((rc = check_slot(c, slot, true)) || (rc = check_params(p))) return rc;
	HIP_TRY(hipSetDevice(c->device));
	const ViewHost &v = c->views[slot];
	const size_t n = (size_t)v.w*v.h;
	// scratch: xyz (3n doubles) | rgb (3n) | valid (n) | 3 counters, in the band buffer (free between runs)
	const size_t need = 3*n + (3*n + n + 7)/8 + 4;
	if ((rc = ensure(c->wbuf, c->wbuf_cap, need))) return rc;
	double *d_xyz = c->wbuf;
	uint8_t *d_rgb = reinterpret_cast<uint8_t *>(c->wbuf + 3*n);
	uint8_t *d_valid = d_rgb + 3*n;
	unsigned long long *d_counts = reinterpret_cast<unsigned long long *>(c->wbuf + need - 4);
	HIP_TRY(hipMemsetAsync(d_counts, 0, 3*sizeof(unsigned long long), c->stream));
	{ Scope s(c, "point_cloud_kernel");
	  launch_point_cloud(c->stream, c->d_views, slot, v.w, v.h, *p, d_xyz, d_rgb, d_valid, d_counts); }
	HIP_TRY(hipGetLastError());
	unsigned long long hc[3] = { 0, 0, 0 };
	HIP_TRY(hipMemcpyAsync(hc, d_counts, sizeof(hc), hipMemcpyDeviceToHost, c->stream));
	if (xyz_out) HIP_TRY(hipMemcpyAsync(xyz_out, d_xyz, 3*n*sizeof(double), hipMemcpyDeviceToHost, c->stream));
	if (rgb_out) HIP_TRY(hipMemcpyAsync(rgb_out, d_rgb, 3*n, hipMemcpyDeviceToHost, c->stream));
	if (valid_out) HIP_TRY(hipMemcpyAsync(valid_out, d_valid, n, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (n_points) *n_points = (int64_t)hc[0];
	if (n_masked) *n_masked = (int64_t)hc[1];
	if (n_finite) *n_finite = (int64_t)hc[2];
	return SRH_OK;
}

// ------------------------------------------------------------------ epipolar curves on request
extern "C" int srh_epipolar_curves(srh_context *c, int ref, int oth, const srh_params *p, int mvs,
                                   int nq, const int32_t *xy, int32_t *out_xy, int max_pts, int32_t *counts)
{
	int rc;
	if ((rc = check_slot(c, ref, true)) || (rc = check_slot(c, oth, true)) || (rc = check_params(p))) return rc;
	if (ref == oth) return fail(SRH_E_INVALID, "ref and other view are the same slot");
	if (nq < 0 || max_pts < 0 || (nq > 0 && (!xy || !counts)) || (nq > 0 && max_pts > 0 && !out_xy))
		return fail(SRH_E_INVALID, "bad query buffers");
	if (nq == 0) return SRH_OK;
	HIP_TRY(hipSetDevice(c->device));
	int32_t *d_xy = nullptr, *d_out = nullptr, *d_n = nullptr;
	const size_t out_bytes = (size_t)nq*2*(size_t)max_pts*sizeof(int32_t);
	hipError_t e = hipMalloc((void **)&d_xy, (size_t)nq*2*sizeof(int32_t));
	if (e == hipSuccess) e = hipMalloc((void **)&d_n, (size_t)nq*sizeof(int32_t));
	if (e == hipSuccess && out_bytes) e = hipMalloc((void **)&d_out, out_bytes);
	if (e == hipSuccess) e = hipMemcpyAsync(d_xy, xy, (size_t)nq*2*sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) {
		Scope s(c, "epipolar_curves_kernel");
		launch_epipolar_curves(c->stream, c->d_views, ref, oth, *p, mvs ? 1 : 0, nq, d_xy, d_out, max_pts, d_n);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(counts, d_n, (size_t)nq*sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess && out_bytes) e = hipMemcpyAsync(out_xy, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	else (void)hipStreamSynchronize(c->stream);
	if (d_xy) (void)hipFree(d_xy);
	if (d_n) (void)hipFree(d_n);
	if (d_out) (void)hipFree(d_out);
	if (e != hipSuccess) return fail(SRH_E_DEVICE, "epipolar curves: %s", hipGetErrorString(e));
	return SRH_OK;
}

// ------------------------------------------------------------------ MultiViewStereo, MRF branch
extern "C" void srh_mrf_params_defaults(srh_mrf_params *m)
{
	if (!m) return;
	m->beta = 1; m->lambda = 1; m->phi_u = 0.5; m->psi_u = 0.002;     // multiviewstereo.cpp:98-101
	m->max_iters = 50; m->min_energy_drop = 5;                       // :631, :641
}

extern "C" int srh_mvs_mrf_estimate(srh_context *c, int slot, int K, const void *peaks_dev, const srh_mrf_params *m, srh_mrf_info *info)
{
	int rc;
	if ((rc = check_slot(c, slot, true))) return rc;
	if (!peaks_dev || !m) return fail(SRH_E_INVALID, "null peaks / params");
	if (K < 1 || K > 15) return fail(SRH_E_UNSUPPORTED, "top_k %d outside [1,15] (one lane per label, 16 lanes per pixel)", K);
	HIP_TRY(hipSetDevice(c->device));
	const ViewHost &v = c->views[slot];
	const int w = v.w, h = v.h;
	if ((rc = ensure(c->mrf, c->mrf_cap, mrf_scratch_doubles(w, h)))) return rc;
	c->mrf_w = c->mrf_h = c->mrf_k = 0;
	MrfLayout lay;
	{ Scope s(c, "mrf_data_kernel");
	  HIP_TRY(launch_mrf_setup(c->stream, c->mrf, w, h, K, m->beta, m->lambda, m->phi_u, static_cast<const double *>(peaks_dev), lay)); }
	struct { double energy, pad; unsigned status[4]; } hs;
	auto energy = [&](double &e) -> int {
		{ Scope s(c, "mrf_energy_kernel");
		  HIP_TRY(launch_mrf_energy(c->stream, c->mrf, w, h, K, m->psi_u)); }
		HIP_TRY(hipMemcpyAsync(&hs, lay.energy, sizeof(hs), hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		if (hs.status[0])
			return fail(SRH_E_DEVICE, "MRF sweep %u: band %u waited too long for the band above (hand-off never arrived)", hs.status[2], hs.status[1] - 1);
		e = hs.energy;
		return SRH_OK;
	};
	// multiviewstereo.cpp:627-641
	double e = 0.0, prev = 0.0;
	if ((rc = energy(e))) return rc;
	const double e0 = e;
	int num_iters = m->max_iters, iters = 0;
	do {
		if (cancelled(c)) return fail(SRH_E_CANCELLED, "cancelled");
		prev = e;
		{ Scope s(c, "mrf_pass_kernel");
		  HIP_TRY(launch_mrf_sweep(c->stream, c->mrf, w, h, K, m->psi_u, iters)); }
		if ((rc = energy(e))) return rc;
		++iters;
	} while (prev - e > m->min_energy_drop && num_iters-- > 0);
	{ Scope s(c, "mrf_depth_kernel");
	  HIP_TRY(launch_mrf_depth(c->stream, c->d_views, slot, c->mrf, w, h, K)); }
	HIP_TRY(hipStreamSynchronize(c->stream));
	c->mrf_w = w; c->mrf_h = h; c->mrf_k = K;
	if (info) { info->iterations = iters; info->energy_initial = e0; info->energy_final = e; }
	return SRH_OK;
}

extern "C" int srh_mvs_initial_estimate_mrf(srh_context *c, int slot, const int32_t *neigh, int nneigh, const srh_params *p,
                                            const srh_mrf_params *m, srh_mrf_info *info)
{
	int rc;
	if ((rc = check_slot(c, slot, true)) || (rc = check_params(p))) return rc;
	if (!m) return fail(SRH_E_INVALID, "null MRF params");
	if (p->top_k < 1 || p->top_k > 15) return fail(SRH_E_UNSUPPORTED, "top_k %d outside [1,15]", p->top_k);
	HIP_TRY(hipSetDevice(c->device));
	const ViewHost &v = c->views[slot];
	if ((rc = ensure(c->mrf_peaks, c->mrf_peaks_cap, (size_t)v.w*v.h*p->top_k*2))) return rc;
	if ((rc = srh_mvs_initial_estimate(c, slot, neigh, nneigh, p, 0, 0, c->mrf_peaks))) return rc;
	return srh_mvs_mrf_estimate(c, slot, p->top_k, c->mrf_peaks, m, info);
}

extern "C" int srh_mvs_initial_estimate_peaks(srh_context *c, int slot, const int32_t *neigh, int nneigh, const srh_params *p)
{
	int rc;
	if ((rc = check_slot(c, slot, true)) || (rc = check_params(p))) return rc;
	if (p->top_k < 1 || p->top_k > 15) return fail(SRH_E_UNSUPPORTED, "top_k %d outside [1,15]", p->top_k);
	HIP_TRY(hipSetDevice(c->device));
	ViewHost &v = c->views[slot];
	v.peaks_k = 0;
	if ((rc = ensure(v.peaks, v.peaks_cap, (size_t)v.w*v.h*p->top_k*2))) return rc;
	if ((rc = srh_mvs_initial_estimate(c, slot, neigh, nneigh, p, 0, 0, v.peaks))) return rc;
	v.peaks_k = p->top_k;
	return SRH_OK;
}

// The MRF stage of several views at once.  A sweep occupies one workgroup per 16 image rows and is bound by its own
// dependency chain (60 of 256 CUs at 1280x960), so the views' sweeps run side by side, each on a stream of its own with
// its own scratch; every view keeps the reference's stopping rule for itself (multiviewstereo.cpp:627-641).
extern "C" int srh_mvs_mrf_estimate_views(srh_context *c, const int32_t *slots, int nviews, const srh_mrf_params *m, srh_mrf_info *infos)
{
	int rc;
	if (!c) return fail(SRH_E_INVALID, "null context");
	if (!slots || nviews < 1 || nviews > SRH_MAX_VIEWS || !m) return fail(SRH_E_INVALID, "bad view list / params");
	for (int i = 0; i < nviews; ++i) {
		if ((rc = check_slot(c, slots[i], true))) return rc;
		if (c->views[slots[i]].peaks_k < 1) return fail(SRH_E_INVALID, "view slot %d has no peaks (srh_mvs_initial_estimate_peaks first)", slots[i]);
		for (int j = 0; j < i; ++j) if (slots[j] == slots[i]) return fail(SRH_E_INVALID, "view slot %d listed twice", slots[i]);
	}
	HIP_TRY(hipSetDevice(c->device));
	c->mrf_w = c->mrf_h = c->mrf_k = 0;                       // the per-view runs leave no state srh_mvs_mrf_state could report
	struct HostState { double energy, pad; unsigned status[4]; };
	if (!c->mrf_host) HIP_TRY(hipHostMalloc(&c->mrf_host, sizeof(HostState)*SRH_MAX_VIEWS, hipHostMallocDefault));
	HostState *hs = static_cast<HostState *>(c->mrf_host);
	hipEvent_t ev = nullptr;
	HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
	HIP_TRY(hipEventRecord(ev, c->stream));                         // the peaks were written on the context's stream
	struct Run { int slot, w, h, K, iters, left; double e, prev, e0; bool active; MrfLayout lay; hipStream_t st; double *buf; };
	std::vector<Run> runs(nviews);
	auto cleanup = [&](int code) { for (int i = 0; i < nviews; ++i) if (runs[i].st) hipStreamSynchronize(runs[i].st); hipEventDestroy(ev); return code; };
	for (int i = 0; i < nviews; ++i) {
		Run &r = runs[i];
		ViewHost &v = c->views[slots[i]];
		r.slot = slots[i]; r.w = v.w; r.h = v.h; r.K = v.peaks_k; r.iters = 0; r.left = m->max_iters; r.active = true; r.st = nullptr;
		if (!c->mrf_stream[i]) { if (hipStreamCreateWithFlags(&c->mrf_stream[i], hipStreamNonBlocking) != hipSuccess) return cleanup(fail(SRH_E_DEVICE, "stream creation failed")); }
		r.st = c->mrf_stream[i];
		if ((rc = ensure(v.mrf, v.mrf_cap, mrf_scratch_doubles(r.w, r.h)))) return cleanup(rc);
		r.buf = v.mrf;
		if (hipStreamWaitEvent(r.st, ev, 0) != hipSuccess ||
		    launch_mrf_setup(r.st, r.buf, r.w, r.h, r.K, m->beta, m->lambda, m->phi_u, v.peaks, r.lay) != hipSuccess ||
		    launch_mrf_energy(r.st, r.buf, r.w, r.h, r.K, m->psi_u) != hipSuccess ||
		    hipMemcpyAsync(&hs[i], r.lay.energy, sizeof(HostState), hipMemcpyDeviceToHost, r.st) != hipSuccess)
			return cleanup(fail(SRH_E_DEVICE, "MRF setup of view slot %d failed: %s", r.slot, hipGetErrorString(hipGetLastError())));
	}
	for (int i = 0; i < nviews; ++i) {
		if (hipStreamSynchronize(runs[i].st) != hipSuccess) return cleanup(fail(SRH_E_DEVICE, "MRF setup of view slot %d failed", runs[i].slot));
		runs[i].e = runs[i].e0 = hs[i].energy;
	}
	for (int nactive = nviews; nactive > 0; ) {
		if (cancelled(c)) return cleanup(fail(SRH_E_CANCELLED, "cancelled"));
		for (int i = 0; i < nviews; ++i) {
			Run &r = runs[i];
			if (!r.active) continue;
			r.prev = r.e;
			if (launch_mrf_sweep(r.st, r.buf, r.w, r.h, r.K, m->psi_u, r.iters) != hipSuccess ||
			    launch_mrf_energy(r.st, r.buf, r.w, r.h, r.K, m->psi_u) != hipSuccess ||
			    hipMemcpyAsync(&hs[i], r.lay.energy, sizeof(HostState), hipMemcpyDeviceToHost, r.st) != hipSuccess)
				return cleanup(fail(SRH_E_DEVICE, "MRF sweep of view slot %d failed: %s", r.slot, hipGetErrorString(hipGetLastError())));
		}
		for (int i = 0; i < nviews; ++i) {
			Run &r = runs[i];
			if (!r.active) continue;
			if (hipStreamSynchronize(r.st) != hipSuccess) return cleanup(fail(SRH_E_DEVICE, "MRF sweep of view slot %d failed", r.slot));
			if (hs[i].status[0])
				return cleanup(fail(SRH_E_DEVICE, "MRF sweep %u of view slot %d: band %u waited too long for the band above", hs[i].status[2], r.slot, hs[i].status[1] - 1));
			r.e = hs[i].energy;
			++r.iters;
			if (!(r.prev - r.e > m->min_energy_drop && r.left-- > 0)) {     // do { ... } while (prev - energy > 5 && numIters-- > 0)
				r.active = false; --nactive;
				if (launch_mrf_depth(r.st, c->d_views, r.slot, r.buf, r.w, r.h, r.K) != hipSuccess)
					return cleanup(fail(SRH_E_DEVICE, "MRF depth of view slot %d failed", r.slot));
			}
		}
	}
	for (int i = 0; i < nviews; ++i) {
		if (hipStreamSynchronize(runs[i].st) != hipSuccess) return cleanup(fail(SRH_E_DEVICE, "MRF depth of view slot %d failed", runs[i].slot));
		if (infos) { infos[i].iterations = runs[i].iters; infos[i].energy_initial = runs[i].e0; infos[i].energy_final = runs[i].e; }
	}
	hipEventDestroy(ev);
	return SRH_OK;
}

extern "C" int srh_mvs_mrf_dims(srh_context *c, int *w, int *h, int *top_k)
{
	if (!c) return fail(SRH_E_INVALID, "null context");
	if (!c->mrf_w) return fail(SRH_E_INVALID, "no finished single-view MRF run on this context");
	if (w) *w = c->mrf_w;
	if (h) *h = c->mrf_h;
	if (top_k) *top_k = c->mrf_k;
	return SRH_OK;
}

extern "C" int srh_mvs_mrf_state(srh_context *c, int bw, int bh, int bk, int32_t *labels, double *data_costs, double *messages)
{
	if (!c) return fail(SRH_E_INVALID, "null context");
	if (!c->mrf_w) return fail(SRH_E_INVALID, "no finished single-view MRF run on this context");
	if (bw != c->mrf_w || bh != c->mrf_h || bk != c->mrf_k)
		return fail(SRH_E_INVALID, "buffers sized for %dx%d, K = %d, the last MRF run was %dx%d, K = %d", bw, bh, bk, c->mrf_w, c->mrf_h, c->mrf_k);
	HIP_TRY(hipSetDevice(c->device));
	const int w = c->mrf_w, h = c->mrf_h, K = c->mrf_k, L = K + 1;
	const size_t n = (size_t)w*h;
	MrfLayout lay;
	launch_mrf_layout(c->mrf, w, h, lay);
	if (labels) HIP_TRY(hipMemcpyAsync(labels, lay.ans, n*sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
	// device rows are 16 labels wide: copy the first L of each
	if (data_costs) HIP_TRY(hipMemcpy2DAsync(data_costs, L*sizeof(double), lay.D, 16*sizeof(double), L*sizeof(double), n, hipMemcpyDeviceToHost, c->stream));
	if (messages) {
		HIP_TRY(hipMemcpy2DAsync(messages, 2*L*sizeof(double), lay.Mh, 16*sizeof(double), L*sizeof(double), n, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipMemcpy2DAsync(messages + L, 2*L*sizeof(double), lay.Mv, 16*sizeof(double), L*sizeof(double), n, hipMemcpyDeviceToHost, c->stream));
	}
	HIP_TRY(hipStreamSynchronize(c->stream));
	return SRH_OK;
}

// ------------------------------------------------------------------ GUI-side users of the camera model
extern "C" int srh_epipolar_preview(srh_context *c, int ref, int oth, double zmin, double zmax, int nd,
                                    int nq, const double *xy, double *out_xy, int32_t *counts)
{
	int rc;
	if ((rc = check_slot(c, ref, true)) || (rc = check_slot(c, oth, true))) return rc;
	if (nd < 2 || nq < 0 || (nq > 0 && (!xy || !out_xy || !counts))) return fail(SRH_E_INVALID, "bad preview arguments");
	if (nq == 0) return SRH_OK;
	HIP_TRY(hipSetDevice(c->device));
	const size_t nout = (size_t)nq*nd*2;
	// scratch in the band buffer: [xy | out | counts]
	if ((rc = ensure(c->wbuf, c->wbuf_cap, (size_t)2*nq + nout + (size_t)(nq + 1)/2 + 2))) return rc;
	double *d_xy = c->wbuf, *d_out = c->wbuf + 2*(size_t)nq;
	int32_t *d_cnt = reinterpret_cast<int32_t *>(d_out + nout);
	HIP_TRY(hipMemcpyAsync(d_xy, xy, (size_t)2*nq*sizeof(double), hipMemcpyHostToDevice, c->stream));
	{ Scope s(c, "epipolar_preview_kernel");
	  launch_epipolar_preview(c->stream, c->d_views, ref, oth, zmin, zmax, nd, nq, d_xy, d_out, d_cnt); }
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipMemcpyAsync(out_xy, d_out, nout*sizeof(double), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipMemcpyAsync(counts, d_cnt, (size_t)nq*sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return SRH_OK;
}

extern "C" int srh_refraction_error(srh_context *c, int v1, int v2, int n, const double *p1, const double *p2,
                                    double *err_out, double *total, double *average)
{
	int rc;
	if ((rc = check_slot(c, v1, true)) || (rc = check_slot(c, v2, true))) return rc;
	if (n < 0 || (n > 0 && (!p1 || !p2))) return fail(SRH_E_INVALID, "bad correspondence arguments");
	HIP_TRY(hipSetDevice(c->device));
	std::vector<double> err((size_t)n);
	if (n > 0) {
		if ((rc = ensure(c->wbuf, c->wbuf_cap, (size_t)5*n))) return rc;
		double *d1 = c->wbuf, *d2 = c->wbuf + 2*(size_t)n, *de = c->wbuf + 4*(size_t)n;
		HIP_TRY(hipMemcpyAsync(d1, p1, (size_t)2*n*sizeof(double), hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync(d2, p2, (size_t)2*n*sizeof(double), hipMemcpyHostToDevice, c->stream));
		{ Scope s(c, "refraction_error_kernel");
		  launch_refraction_error(c->stream, c->d_views, v1, v2, n, d1, d2, de); }
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMemcpyAsync(err.data(), de, (size_t)n*sizeof(double), hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	double tot = 0.0;                                             // totalError: summed in correspondence order (:440)
	for (int i = 0; i < n; ++i) tot += err[i]*err[i];
	if (err_out) for (int i = 0; i < n; ++i) err_out[i] = err[i];
	if (total) *total = tot;
	if (average) *average = tot / n;
	return SRH_OK;
}

// ------------------------------------------------------------------ multi-GPU exchange
extern "C" int srh_comm_unique_id(void *id_out) {
	if (!id_out) return fail(SRH_E_INVALID, "null id buffer");
	if (const char *e = rccl_unique_id_get(id_out)) return fail(SRH_E_UNSUPPORTED, "RCCL: %s", e);
	return SRH_OK;
}

extern "C" int srh_comm_init(srh_context *c, int nranks, int rank, const void *id) {
	if (!c || !id) return fail(SRH_E_INVALID, "null argument");
	if (nranks < 1 || rank < 0 || rank >= nranks) return fail(SRH_E_INVALID, "rank %d of %d", rank, nranks);
	HIP_TRY(hipSetDevice(c->device));
	if (c->comm) { (void)rccl_comm_destroy(c->comm); c->comm = nullptr; }
	// librccl absent, or without the entry points this file needs: not a device failure (include/stereo_recon_hip.h)
	if (!rccl_available()) { c->comm_ranks = 0; return fail(SRH_E_UNSUPPORTED, "RCCL: librccl cannot be loaded"); }
	// (non-blocking communicator: a rank that never arrives is an error after the timeout, not a hang)
	if (const char *e = rccl_comm_init(&c->comm, nranks, rank, id)) { c->comm = nullptr; c->comm_ranks = 0; return fail(SRH_E_DEVICE, "RCCL: %s", e); }
	c->comm_ranks = nranks; c->comm_rank = rank;
	return SRH_OK;
}

// a collective that failed or timed out: the communicator is aborted (its peers' calls then fail too instead of
// waiting), the context has none until srh_comm_init is called again
static int comm_failed(srh_context *c, const char *what, const char *e) {
	const int rc = fail(SRH_E_DEVICE, "RCCL %s: %s", what, e);
	rccl_comm_abort(c->comm);
	c->comm = nullptr; c->comm_ranks = 0;
	return rc;
}

extern "C" int srh_comm_version(void) { return rccl_version(); }

extern "C" int srh_comm_set_timeout_ms(int ms) {
	if (ms <= 0) return fail(SRH_E_INVALID, "timeout must be > 0 ms");
	rccl_set_timeout_ms(ms);
	return SRH_OK;
}

extern "C" int srh_comm_info(srh_context *c, int *nranks, int *rank) {
	if (!c) return fail(SRH_E_INVALID, "null context");
	if (nranks) *nranks = c->comm ? c->comm_ranks : 0;
	if (rank) *rank = c->comm ? c->comm_rank : -1;
	return SRH_OK;
}

extern "C" int srh_comm_gather_depth(srh_context *c, int slot, int root, void *recv_dev) {
	int rc = check_slot(c, slot, true); if (rc) return rc;
	if (!c->comm) return fail(SRH_E_INVALID, "srh_comm_init has not been called");
	if (root < 0 || root >= c->comm_ranks) return fail(SRH_E_INVALID, "root %d of %d", root, c->comm_ranks);
	if (c->comm_rank == root && !recv_dev) return fail(SRH_E_INVALID, "null receive buffer on the root");
	HIP_TRY(hipSetDevice(c->device));
	const ViewHost &v = c->views[slot];
	if (const char *e = rccl_gather_f64(c->comm, c->comm_ranks, c->comm_rank, root, v.depth, (double *)recv_dev,
	                                    (size_t)v.w*v.h, c->stream))
		return comm_failed(c, "gather", e);
	// the collective is ON the stream now; its completion is waited for here, with the bound -- a peer that dies inside it
	// must not leave this rank in a later, unbounded hipStreamSynchronize
	if (const char *e = rccl_wait_stream(c->comm, c->stream, "the depth-map gather")) return comm_failed(c, "gather", e);
	return SRH_OK;
}

extern "C" int srh_comm_allgather_depth(srh_context *c, int slot, void *recv_dev) {
	int rc = check_slot(c, slot, true); if (rc) return rc;
	if (!c->comm) return fail(SRH_E_INVALID, "srh_comm_init has not been called");
	if (!recv_dev) return fail(SRH_E_INVALID, "null receive buffer");
	HIP_TRY(hipSetDevice(c->device));
	const ViewHost &v = c->views[slot];
	if (const char *e = rccl_allgather_f64(c->comm, v.depth, (double *)recv_dev, (size_t)v.w*v.h, c->stream))
		return comm_failed(c, "all-gather", e);
	if (const char *e = rccl_wait_stream(c->comm, c->stream, "the depth-map all-gather")) return comm_failed(c, "all-gather", e);
	return SRH_OK;
}

extern "C" int srh_comm_allgather_host(srh_context *c, const double *send_host, size_t count, double *recv_host) {
	if (!c || !send_host || !recv_host || count == 0) return fail(SRH_E_INVALID, "null argument");
	if (!c->comm) return fail(SRH_E_INVALID, "srh_comm_init has not been called");
	HIP_TRY(hipSetDevice(c->device));
	int rc;
	// the staging below is the band scratch, which a queued MultiViewStereo estimate (slot 0) still reads its support
	// windows from on its own stream: finish the estimates in flight first, like every other entry point
	if ((rc = mvs_settle_all(c))) return rc;
	// staging in the band scratch (free between runs): [send | recv]
	if ((rc = ensure(c->wbuf, c->wbuf_cap, count*(size_t)(c->comm_ranks + 1)))) return rc;
	HIP_TRY(hipMemcpyAsync(c->wbuf, send_host, count*sizeof(double), hipMemcpyHostToDevice, c->stream));
	if (const char *e = rccl_allgather_f64(c->comm, c->wbuf, c->wbuf + count, count, c->stream))
		return comm_failed(c, "all-gather", e);
	HIP_TRY(hipMemcpyAsync(recv_host, c->wbuf + count, count*(size_t)c->comm_ranks*sizeof(double), hipMemcpyDeviceToHost, c->stream));
	if (const char *e = rccl_wait_stream(c->comm, c->stream, "the all-gather")) return comm_failed(c, "all-gather", e);   // (bounded: never a bare hipStreamSynchronize behind a collective)
	return SRH_OK;
}

extern "C" int srh_comm_allgather_views(srh_context *c, const int32_t *slots, int nviews) {
	int rc;
	if (!c) return fail(SRH_E_INVALID, "null context");
	if (!c->comm) return fail(SRH_E_INVALID, "srh_comm_init has not been called");
	if (!slots || nviews < 1 || nviews > SRH_MAX_VIEWS) return fail(SRH_E_INVALID, "bad view list");
	size_t npix = 0;
	for (int v = 0; v < nviews; ++v) {
		if ((rc = check_slot(c, slots[v], true))) return rc;
		npix = std::max(npix, (size_t)c->views[slots[v]].w*c->views[slots[v]].h);
	}
	HIP_TRY(hipSetDevice(c->device));
	const int world = c->comm_ranks, rank = c->comm_rank;
	const int per = (nviews + world - 1)/world;                   // equal contributions: short ranks pad with NaN maps
	auto shard = [&](int r, int &lo, int &hi) { const int base = nviews/world, extra = nviews % world;
	                                            lo = r*base + std::min(r, extra); hi = lo + base + (r < extra ? 1 : 0); };
	// [send | recv] in the band scratch (free between runs); everything below is ordered on the context's stream
	if ((rc = ensure(c->wbuf, c->wbuf_cap, (size_t)(world + 1)*per*npix))) return rc;
	double *send = c->wbuf, *recv = c->wbuf + (size_t)per*npix;
	launch_fill(c->stream, send, (size_t)per*npix, __builtin_nan(""));
	int lo, hi;
	shard(rank, lo, hi);
	for (int v = lo; v < hi; ++v) {
		const ViewHost &vh = c->views[slots[v]];
		HIP_TRY(hipMemcpyAsync(send + (size_t)(v - lo)*npix, vh.depth, (size_t)vh.w*vh.h*sizeof(double), hipMemcpyDeviceToDevice, c->stream));
	}
	if (const char *e = rccl_allgather_f64(c->comm, send, recv, (size_t)per*npix, c->stream))
		return comm_failed(c, "all-gather", e);
	for (int r = 0; r < world; ++r) {
		if (r == rank) continue;
		int rlo, rhi;
		shard(r, rlo, rhi);
		for (int v = rlo; v < rhi; ++v) {
			const ViewHost &vh = c->views[slots[v]];
			HIP_TRY(hipMemcpyAsync(vh.depth, recv + ((size_t)r*per + (v - rlo))*npix, (size_t)vh.w*vh.h*sizeof(double), hipMemcpyDeviceToDevice, c->stream));
		}
	}
	if (const char *e = rccl_wait_stream(c->comm, c->stream, "the views' all-gather")) return comm_failed(c, "all-gather", e);
	return SRH_OK;
}

extern "C" int srh_comm_destroy(srh_context *c) {
	if (!c) return fail(SRH_E_INVALID, "null context");
	if (c->comm) {
		const char *e = rccl_comm_destroy(c->comm);
		c->comm = nullptr; c->comm_ranks = 0;
		if (e) return fail(SRH_E_DEVICE, "RCCL: the communicator did not finalize: %s", e);
	}
	return SRH_OK;
}

// ------------------------------------------------------------------ measurement
extern "C" int srh_get_stats(srh_context *c, srh_stats *out) {
	if (!c || !out) return fail(SRH_E_INVALID, "null argument");
	HIP_TRY(hipSetDevice(c->device));
	int rc = mvs_settle_all(c);
	if (rc) return rc;
	rc = fetch_counters(c, c->stats.used_dense_path);
	if (rc) return rc;
	c->stats.band_budget_bytes = (int64_t)c->budget_used;
	*out = c->stats;
	return SRH_OK;
}

extern "C" int srh_profile_enable(srh_context *c, int on) {
	if (!c) return fail(SRH_E_INVALID, "null context");
	c->profiling = on != 0;
	return SRH_OK;
}

extern "C" int srh_profile_reset(srh_context *c) {
	if (!c) return fail(SRH_E_INVALID, "null context");
	HIP_TRY(hipSetDevice(c->device));
	int rc = drain_profile(c); if (rc) return rc;
	c->prof.clear();
	return SRH_OK;
}

extern "C" int srh_profile_get(srh_context *c, const char *name, double *total_ms, int64_t *launches) {
	if (!c || !name) return fail(SRH_E_INVALID, "null argument");
	HIP_TRY(hipSetDevice(c->device));
	int rc = drain_profile(c); if (rc) return rc;
	auto it = c->prof.find(name);
	if (it == c->prof.end()) return fail(SRH_E_INVALID, "kernel '%s' was never launched under profiling", name);
	if (total_ms) *total_ms = it->second.ms;
	if (launches) *launches = it->second.n;
	return SRH_OK;
}

extern "C" int srh_profile_dump(srh_context *c, char *buf, size_t cap) {
	if (!c || !buf || cap == 0) return fail(SRH_E_INVALID, "null argument");
	HIP_TRY(hipSetDevice(c->device));
	int rc = drain_profile(c); if (rc) return rc;
	size_t off = 0;
	buf[0] = 0;
	for (auto &kv : c->prof) {
		int n = snprintf(buf + off, cap - off, "%s %.6f %lld\n", kv.first.c_str(), kv.second.ms, (long long)kv.second.n);
		if (n < 0 || (size_t)n >= cap - off) break;
		off += (size_t)n;
	}
	return SRH_OK;
}
