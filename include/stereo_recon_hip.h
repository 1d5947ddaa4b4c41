/*
 * stereo_recon_hip.h -- C-ABI of libstereo_recon_hip.so: the MI355X (gfx950) dense
 * matching-cost / support-weight aggregation / winner-take-all path of
 * StereoReconstruction, i.e. what the reference computes inside
 *   TwoViewStereo::computeCostVolumes / crossCheck   (stereo/twoviewstereo.cpp:233-502, 596-672)
 *   MultiViewStereo::computeInitialEstimate / crossCheck (stereo/multiviewstereo.cpp:524-662, 666-729)
 * and the helpers they call.
 *
 * Plain C: opaque context, POD structs, caller-owned buffers, integer status
 * codes, no exceptions, no torch / Qt / Eigen types.  A Qt adapter (or the
 * Qt-free C++ classes in stereoreconstruction_amd/host/) sits on top and keeps
 * the reference's TwoViewStereo / MultiViewStereo class API; INTEGRATION.md shows
 * the binding.  All arithmetic on the path is IEEE double, evaluated in the
 * reference's operation order with FMA contraction off.
 *
 * Every entry point returns SRH_OK (0) or a negative SRH_E_* code;
 * srh_last_error() returns a human-readable message for the calling thread.
 */
#ifndef STEREO_RECON_HIP_H
#define STEREO_RECON_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRH_ABI_VERSION 5

enum {
	SRH_OK = 0,
	SRH_E_INVALID = -1,    /* bad argument (NULL, size mismatch, slot out of range ...) */
	SRH_E_DEVICE = -2,     /* HIP runtime error (message has the hipError string) */
	SRH_E_NO_DEVICE = -3,  /* no usable GPU: the library has NO CPU fallback */
	SRH_E_CANCELLED = -4,  /* cancel flag observed between launches */
	SRH_E_UNSUPPORTED = -5
};

enum { SRH_WEIGHT_ADAPTIVE = 0, SRH_WEIGHT_GEODESIC = 1 };
enum { SRH_MAX_VIEWS = 64 };

/* Snapshot of a reference `Camera` (project/camera.hpp:168-185): the adapter
 * copies these at TwoViewStereo construction / MultiViewStereo::initialize so the
 * GUI may keep mutating its Camera objects (races).
 * 3x3 matrices are row-major. */
typedef struct srh_camera {
	double K[9], Kinv[9], R[9], Rinv[9];
	double t[3], C[3];
	double dist[5];          /* k1,k2,p1,p2,k3 -- LensDistortions order (project.cpp:143-147) */
	int32_t is_distorted;    /* Camera::isDistorted() */
	int32_t is_refractive;   /* Camera::isRefractive() */
	double plane_normal[3];  /* Camera::plane().normal(), camera-local, unit */
	double plane_dist;       /* Camera::plane().distance() */
	double refr_index;       /* Camera::refractiveIndex() */
	double pdir[3];          /* Camera::principleRay().direction() */
} srh_camera;

/* Every constant the reference hard-codes on this path, with its default
 * (twoviewstereo.cpp:64-80, multiviewstereo.cpp:90-102, adaptiveweight.cpp:26,
 * geodesicweight.cpp:33-41) plus the run-time arguments of
 * TwoViewStereo::TwoViewStereo / MultiViewStereo::initialize. */
typedef struct srh_params {
	double  min_depth, max_depth;
	int32_t num_depth_levels;
	int32_t window_radius;        /* TwoView 5, MVS 2 */
	double  image_scale;          /* images handed over are ALREADY scaled by this */
	int32_t weight_kind;          /* SRH_WEIGHT_* ; reference default geodesic */
	int32_t geodesic_iters;       /* 3 */
	double  geodesic_sigma;       /* 50 */
	double  geodesic_init;        /* 1e6 */
	double  adaptive_color_sigma; /* 10 */
	double  weight_cutoff;        /* 1e-10 */
	double  bad_ret;              /* 1000 */
	double  max_color_diff;       /* 120 */
	double  second_best_factor;   /* 0.95 */
	double  wta_margin;           /* 1e-10 (the reference's constant; a negative value is honoured by visiting every curve point: slow paths) */
	double  inconsistency_thresh; /* 1 */
	double  peak_threshold;       /* 0.95 */
	double  cross_check_threshold;
	double  neighbour_min_dot;    /* 0.2 */
	int32_t top_k;                /* 9 */
	int32_t num_neighbours;       /* 3 */
} srh_params;

/* Counters of the last run on a context (for N_eval reporting). */
typedef struct srh_stats {
	int64_t n_pixels;        /* reference pixels with mask == WHITE */
	int64_t n_eval;          /* cost evaluations the reference would have performed */
	int64_t n_eval_device;   /* cost evaluations actually performed on the device */
	int32_t used_dense_path; /* 1 if the row-aligned dense kernels ran */
	int32_t used_fused_kernel; /* 1 if that was the single fused kernel (cost rows never leave the CU) */
	int32_t used_strip_kernel; /* 1 if the cost kernel was the persistent strip form (srh_strip.hip) */
	int32_t band_retries;    /* runs repeated with thinner row bands after a band buffer did not fit (since srh_create) */
	/* certified arithmetic (option "arith" = 3), last TwoView pass: reference pixels scanned on fused costs, and those of
	 * them whose decisions the error bound did not cover and that were re-evaluated in the reference's arithmetic */
	int64_t n_certified;
	int64_t n_flagged;
	int64_t band_budget_bytes;  /* band budget the last run worked with (requested, capped by free device memory) */
	/* last MultiViewStereo estimate on the list path: 64-pixel waves (per neighbour) whose candidate windows were
	 * evaluated from LDS copies of the other view (mvs_staged_cost_kernel) / by gathers (mvs_list_cost_kernel: a window
	 * over an image border, too large, too many) */
	int64_t mvs_waves_staged, mvs_waves_listed;
	/* last TwoView pass on the dense path: 64-pixel tiles whose pixels all verified the pass's candidate template and were
	 * scanned over it (twoview_tscan_kernel) / tiles left to the per-pixel curve walk (twoview_scan_kernel) */
	int64_t scan_tiles_template, scan_tiles_walked;
} srh_stats;

typedef struct srh_context srh_context;

/* Progress / cancellation hooks = Task::progressUpdate / stageUpdate / isCancelled
 * (gui/task.hpp:57-105).  `cancel` is polled between kernel launches. */
typedef void (*srh_progress_fn)(int step, const char *stage, void *user);

/* ---- library ---- */
int         srh_abi_version(void);
/* 16 hex digits: hash of the library's sources at build time (which build produced a measurement) */
const char *srh_build_id(void);
const char *srh_last_error(void);
int         srh_device_count(int *count);
/* Hardware queues the HIP runtime was asked for (GPU_MAX_HW_QUEUES in the process environment, else the runtime's
 * default of 4).  srh_mvs_mrf_estimate_views keeps one stream per view busy: with 16 queues the MRF stage of 8 views
 * takes 150 ms, with the default 4 queues 277 ms (profiles/r02_mrf_views.txt).  The variable is read when HIP
 * initialises, so the HOST APPLICATION sets it before its first HIP call; this library never alters the environment. */
int         srh_hw_queues_requested(void);

/* ---- parameters and cameras (host-side math only) ---- */
void srh_params_twoview_defaults(srh_params *p);   /* twoviewstereo.cpp:64-80 */
void srh_params_mvs_defaults(srh_params *p);       /* multiviewstereo.cpp:90-102 */
/* Camera::set(K,R,t) + setLensDistortion + setPlane/setRefractiveIndex
 * (project/camera.cpp:225-240, 302-344).  dist / plane_normal may be NULL. */
int  srh_camera_from_krt(const double K[9], const double R[9], const double t[3],
                         const double dist[5],
                         const double plane_normal[3], double plane_dist, double refr_index,
                         srh_camera *out);
/* Camera::setP (project/camera.cpp:251-288, the path project files take: project.cpp:118-135): P is the
 * row-major 3x4 projection matrix; it is scaled by 1/|P[2,0:3]|^2, its left 3x3 block RQ-factorised
 * (Householder QR of the row-reversed transpose, as Eigen's HouseholderQR does it) into K (positive
 * diagonal) and R, t = K^-1 P[:,3]; then as srh_camera_from_krt. */
int  srh_camera_from_p(const double P[12], const double dist[5],
                       const double plane_normal[3], double plane_dist, double refr_index,
                       srh_camera *out);
/* The certified arithmetic's error bound (option "arith" = 3) for these parameters, host arithmetic only:
 * a fast-form candidate with A = sqrt(sum2), B = sqrt(sum3) has |cost_fused - cost_reference| <= k1/B + k2/A + k3, and is
 * certified when that is <= e0, i.e. sum3 >= srh_cert_sigma3(p, mvs, sum2); the one-pass form also needs
 * Q3 <= zmax2*sum3.  m_hi = max_color_diff + e0; ok = 0: the parameters leave the bound no room (the reference's
 * arithmetic runs).  mvs != 0: the free cost_ncc of MultiViewStereo (score itself, e0 = 2^-36). */
typedef struct srh_cert_info {
	double e0, k1, k2, k3, zmax2, m_hi;
	int32_t ok, taps;
} srh_cert_info;
int    srh_cert_bound(const srh_params *p, int mvs, srh_cert_info *out);
double srh_cert_sigma3(const srh_params *p, int mvs, double sum2);
/* MultiViewStereo::runTask neighbour selection (multiviewstereo.cpp:335-360):
 * neigh[v*p->num_neighbours + k], count[v]. */
int  srh_mvs_neighbours(int nviews, const srh_camera *cams, const srh_params *p,
                        int32_t *neigh, int32_t *count);

/* ---- context ---- */
int  srh_create(int device_ordinal, srh_context **out);
void srh_destroy(srh_context *ctx);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream); NULL = the context's own. */
int  srh_set_stream(srh_context *ctx, void *hip_stream);
int  srh_set_hooks(srh_context *ctx, const volatile int *cancel, srh_progress_fn progress, void *user);
int  srh_synchronize(srh_context *ctx);
/* Tuning / test switches (results never depend on them, "arith" = 1 / 2 excepted):
 *   "force_generic"   0 default paths; 1 never the dense row-aligned TwoView kernels nor the MVS list kernels;
 *                     2 additionally no candidate lists at all (one thread per pixel walks and costs its curve)
 *   "fused"           1: row-aligned pairs run the single fused kernel (geometry + cost + WTA per 16-pixel tile
 *                     in LDS: no cost rows or candidate lists in device memory); 0 (default): the three-kernel form
 *                     (cost rows staged in device memory, separate scan), which is the faster one on MI355X today
 *   "arith"           3 (default) CERTIFIED: the strip kernel's cost loops run with fused multiply-adds (half the FP64
 *                     instructions), the WTA scan checks every comparison it makes against a proven bound on the
 *                     difference between fused and reference costs, and the pixels with a comparison inside the bound
 *                     (srh_stats.n_flagged of n_certified) are re-evaluated in the reference's arithmetic: depth maps
 *                     are the same bits as mode 0 (tests/test_gpu_arith_modes.py).  Kernels without a
 *                     fused form run the reference's arithmetic.  0: the reference's arithmetic everywhere, operation
 *                     by operation.  1: opt-in "fma" -- fused cost loops, UNCHECKED; a winner can change between
 *                     near-tied candidates (rate measured by bench.py --arith fma).  2: opt-in "f32" -- the dense cost
 *                     loops in single precision, two candidates per packed instruction (srh_dense_f32.hip); costs agree
 *                     with the reference's to ~6 digits (rate measured by bench.py --arith f32).  MODES 1 AND 2 ARE THE
 *                     ONLY OPTION VALUES THAT CHANGE RESULTS; row-aligned TwoView path only.
 *   "force_dense"     1: the row-aligned dense plan is proposed for every undistorted, non-refractive pair,
 *                     not only for rigs the host check accepts (the device verifies every candidate and the
 *                     run is repeated on the general kernels when one leaves its row: a test hook for that path)
 *   "strip"           1 (default): the dense cost kernel runs in its persistent strip form (one workgroup walks a
 *                     tile column down the rows, every input enters LDS by LDS-DMA); 0: one workgroup per tile;
 *                     4 / 8: force the 4-wave / 8-wave form of the strip kernel.  Results are identical bits.
 *   "mvs_staged"      1 (default): the MultiViewStereo list cost kernel takes its 25-tap windows from LDS copies of the
 *                     other view (one box per stretch of list slots and wave) where they fit; 0: every window is gathered
 *                     from memory (round 2's kernel).  Results are identical bits.
 *   "mvs_async"       1 (default): srh_mvs_initial_estimate only queues a view's kernels, on one of two side streams in
 *                     turn (two views in flight); 0: one view at a time, the call returns when the view is done.
 *                     Either way the maps are complete whenever another entry point can observe them.
 *   "tv_overlap"      1 (default): srh_twoview_compute queues its second pass on a stream and band buffers of its own beside
 *                     the first (the passes share only the views; the cross-check waits for both); 0: one after the other
 *                     on the context's stream.  Identical bits; the second set of band buffers counts against the budget.
 *   "tscan"           1 (default): on the dense path the scan makes the candidate sequence once per pass and every pixel verifies
 *                     its own curve against it (template scan; a tile with a pixel that does not verify is walked per pixel);
 *                     0: every tile by the per-pixel curve walk.  Identical bits (srh_stats.scan_tiles_template / _walked).
 *   "geodma"          1 (default): on the dense path (GeodesicWeight, radius 5) the support windows come from the persistent
 *                     kernel that fetches its tiles by LDS-DMA from a second copy of the view's edge / tap / mask planes with
 *                     their borders written out (made on first use after an upload: 42 bytes per pixel of device memory; no
 *                     room for it: the other kernel runs); 0: the register-staged windows kernel.  Identical bits.
 *   "side_weights"    1 (default): the row-run path computes a band's support windows on a side stream beside its list kernel;
 *                     0: behind it on the pass's own stream (profiling: every kernel's own duration).  Identical bits.
 *   "list_rows"       1 (default) candidate lists are costed in row runs; 0 in list order
 *   "band_budget_mb"  device scratch per row band the caller asks for (default 32768: a 1920x1080x256 refractive pair
 *                     in one band).  A run never plans with more than a quarter of the device memory that is free at
 *                     that moment (hipMemGetInfo), and a run whose band buffers still do not fit is repeated with the
 *                     budget halved (srh_stats.band_retries) instead of failing: results do not depend on the split.
 *   "mem_limit_mb"    pretend the device has at most this much memory to give (0 = off; tests of the above)
 *   "debug_alloc_limit_mb"  refuse band buffers above this size as if the device were out of memory (0 = off; tests)
 *   "debug_mvs_cmax_hint"   the list capacity the next MultiViewStereo estimate is queued with (0 = forget; test of the redo of a
 *                     view whose lists were cut)
 * The library reads no environment variable: what a host runs is what it set here. */
int  srh_set_option(srh_context *ctx, const char *name, long value);

/* ---- views: what VectorImage::fromQImage + the mask test hold (util/vectorimage.cpp:48-64) ----
 * rgba: w*h*4 bytes R,G,B,A of the ALREADY SCALED image; mask: w*h bytes,
 * 1 <=> mask.pixel(x,y)==WHITE, NULL = all WHITE.  Host pointers; copied. */
int  srh_view_upload(srh_context *ctx, int slot, int w, int h,
                     const uint8_t *rgba, const uint8_t *mask, const srh_camera *cam);
int  srh_view_size(srh_context *ctx, int slot, int *w, int *h);
/* The context keeps one depth map (w*h doubles) per view slot in device memory. */
int  srh_view_depth_download(srh_context *ctx, int slot, double *host_out);
int  srh_view_depth_upload(srh_context *ctx, int slot, const double *host_in);
int  srh_view_depth_device_ptr(srh_context *ctx, int slot, void **dev_ptr);
/* Asynchronous device-to-device copy of the slot's depth map (w*h doubles) into caller-owned
 * DEVICE memory of `dst_bytes` bytes (e.g. a tensor that RCCL then gathers); ordered on the context
 * stream.  SRH_E_INVALID when the buffer is smaller than the map (views may differ in size). */
int  srh_view_depth_copy_to_device(srh_context *ctx, int slot, void *dst_dev, size_t dst_bytes);
/* The reverse: the slot's depth map is replaced by its first w*h doubles of `src_bytes` bytes of DEVICE
 * memory (a map another rank computed, received through RCCL); asynchronous, ordered on the context
 * stream.  SRH_E_INVALID when fewer than w*h doubles are offered. */
int  srh_view_depth_copy_from_device(srh_context *ctx, int slot, const void *src_dev, size_t src_bytes);

/* ---- TwoViewStereo ----
 * One pass of computeCostVolumes (twoviewstereo.cpp:260-333 with ref=left,
 * :431-501 with ref=right): depth map of `ref_slot` against `oth_slot`, rows
 * [y0,y1) (y1<=0: all).  The result stays in the slot's device depth map.  Kernels are
 * enqueued on the context stream; the call returns after the run's verification / sizing
 * read-backs (one or two stream synchronisations), results need no further synchronisation
 * before srh_view_depth_* / srh_twoview_cross_check on the same context.
 * Three implementations produce identical bits: dense row-aligned kernels (rectified
 * pinhole rigs, radius 5 or 2), candidate-list kernels (any geometry, radius <= 5), and a
 * one-thread-per-pixel curve-walk kernel (last resort). */
int  srh_twoview_wta(srh_context *ctx, int ref_slot, int oth_slot, const srh_params *p,
                     int y0, int y1);
/* crossCheck (twoviewstereo.cpp:596-672): left pass, then right pass reading the
 * filtered left map; in place on the two slots' depth maps. */
/* DIAGNOSTIC (evidence for the certified arithmetic, tests/test_gpu_cert_rows.py): the cost rows of reference rows
 * [y0, y1) on the row-aligned dense plan as the cost kernel leaves them -- what the scan looks up.  form: 0 = the
 * reference's arithmetic, 3 = two fused sweeps, 5 = one-pass (the certified forms), 1 = fused unchecked; raw != 0: the
 * certified forms WITHOUT their in-kernel exact redo (an uncertified candidate is NaN).  Layout of cost_out (doubles):
 * [row][tile of 32 pixels][k = column - range.lo, < *cstride_out][pixel of the tile]; an entry the kernels never wrote
 * reads as a NaN with all bits set.  range_out: (lo, hi) per pixel, hi < lo = no candidates.  cost_out == NULL: only
 * *cstride_out (to size the buffer: rows * ceil(w/32)*32 * cstride doubles).  The rows must fit one band;
 * SRH_E_UNSUPPORTED when the pair does not take the dense plan. */
int  srh_twoview_cost_rows(srh_context *ctx, int ref_slot, int other_slot, const srh_params *p, int y0, int y1, int form, int raw,
                           double *cost_out, size_t cost_doubles, int32_t *range_out, int *cstride_out, int *used_strip_kernel);
/* DIAGNOSTIC (tests/test_gpu_geodesic_exp.py): the GeodesicWeight kernels evaluate exp(-w/sigma) (geodesicweight.cpp:128-130)
 * by the device library's exp sequence written out in their source (srh_dense.hip, geo_exp_n: constants of THIS ROCm's
 * ocml) -- this entry evaluates n arguments by that sequence (kernel_out) and by the library's exp() itself (library_out),
 * so that a ROCm whose exp has moved shows up as a bit difference here, not as a last-bit drift of the windows. */
int  srh_debug_exp(srh_context *ctx, const double *x, int n, double *kernel_out, double *library_out);
int  srh_twoview_cross_check(srh_context *ctx, int left_slot, int right_slot, const srh_params *p);
/* computeDepthMaps minus colourisation (twoviewstereo.cpp:150-227): both passes +
 * cross-check, progress steps 1,3,5,8; synchronous; host outputs may be NULL.  The two passes
 * run side by side on the device (option "tv_overlap"); the progress steps are emitted as the
 * passes are QUEUED. */
int  srh_twoview_compute(srh_context *ctx, int left_slot, int right_slot, const srh_params *p,
                         double *left_depth_out, double *right_depth_out);

/* ---- MultiViewStereo ----
 * computeInitialEstimate(view) non-MRF result (multiviewstereo.cpp:524-604,654-660)
 * for `view_slot` against `nneigh` neighbour slots (0..8; the reference keeps 3), rows [y0,y1).
 * peaks_dev (optional, DEVICE pointer, w*h*top_k*2 doubles) receives the sorted
 * top-K (cost,depth) pairs the MRF branch would consume.
 * With option "mvs_async" (default) and no peaks_dev the call queues the view's kernels on a side stream and returns:
 * consecutive calls overlap two views; every other entry point of the context (downloads, copies, cross-checks,
 * srh_synchronize, srh_get_stats ...) first completes what is in flight, so a caller never sees an unfinished map.
 * An error of a queued view (cancellation, device error) is reported by the call that completes it. */
int  srh_mvs_initial_estimate(srh_context *ctx, int view_slot, const int32_t *neigh_slots, int nneigh,
                              const srh_params *p, int y0, int y1, void *peaks_dev);
/* crossCheck(view) (multiviewstereo.cpp:666-729) over `nviews` slots listed in view
 * order; call for view index 0..nviews-1 in order to reproduce the reference's
 * sequential dependence. */
int  srh_mvs_cross_check(srh_context *ctx, const int32_t *slots, int nviews, int view_index,
                         const srh_params *p);

/* ---- MultiViewStereo, MRF branch ----
 * What computeInitialEstimate does with the top-K peaks when the reference is built with CONFIG+=mrf
 * (multiviewstereo.cpp:481-516 cost functions, :610-652 optimisation and label -> depth): a W x H grid, K + 1 labels
 * (the K peaks of each pixel + "unknown"), sequential TRW-S sweeps until the energy drops by no more than
 * min_energy_drop or max_iters + 1 sweeps were made, then depth = the chosen peak's depth (INF for "unknown" or a
 * placeholder peak) where the mask is WHITE.  PARITY UNPINNED: the reference takes the optimiser from a third-party
 * library (-lMRF) that is not in its tree; DESIGN.md 5 says what was restated instead. */
typedef struct srh_mrf_params {
	double  beta, lambda;         /* BETA 1, LAMBDA 1 (multiviewstereo.cpp:98-99) */
	double  phi_u, psi_u;         /* PHIU 0.5, PSIU 0.002 (:100-101) */
	int32_t max_iters;            /* 50 (:631) */
	double  min_energy_drop;      /* 5 (:641) */
} srh_mrf_params;
typedef struct srh_mrf_info {
	int32_t iterations;           /* sweeps made */
	double  energy_initial;       /* totalEnergy() of the all-zero labelling */
	double  energy_final;
} srh_mrf_info;
void srh_mrf_params_defaults(srh_mrf_params *m);
/* peaks_dev: the DEVICE buffer srh_mvs_initial_estimate filled for this view (w*h*top_k (cost, depth) pairs),
 * 1 <= top_k <= 15.  Writes the slot's depth map.  info may be NULL. */
int  srh_mvs_mrf_estimate(srh_context *ctx, int view_slot, int top_k, const void *peaks_dev,
                          const srh_mrf_params *m, srh_mrf_info *info);
/* computeInitialEstimate as a CONFIG+=mrf build runs it: srh_mvs_initial_estimate with the peaks kept in the
 * context's own scratch, then srh_mvs_mrf_estimate on them. */
int  srh_mvs_initial_estimate_mrf(srh_context *ctx, int view_slot, const int32_t *neigh_slots, int nneigh,
                                  const srh_params *p, const srh_mrf_params *m, srh_mrf_info *info);
/* The same for several views at once: srh_mvs_initial_estimate_peaks keeps a view's peaks in the context, then
 * srh_mvs_mrf_estimate_views runs the MRF stage of all listed views side by side (a sweep fills a quarter of the
 * chip and is bound by its own dependency chain; each view stops by the reference's rule for itself) and writes
 * their depth maps.  infos: nviews entries, may be NULL. */
int  srh_mvs_initial_estimate_peaks(srh_context *ctx, int view_slot, const int32_t *neigh_slots, int nneigh, const srh_params *p);
int  srh_mvs_mrf_estimate_views(srh_context *ctx, const int32_t *view_slots, int nviews, const srh_mrf_params *m, srh_mrf_info *infos);
/* State of the last srh_mvs_mrf_estimate on this context, to HOST buffers (each may be NULL): labels (w*h, what
 * getLabel(p) returns), data_costs (w*h*(top_k+1)), messages (w*h*2*(top_k+1): [pixel][towards x+1, towards y+1][label],
 * the message currently stored on that edge).  w, h, top_k say what the caller sized its buffers for: the call fails
 * with SRH_E_INVALID unless they are the dimensions of that run (srh_mvs_mrf_dims reports them).  There is no such
 * state after srh_mvs_mrf_estimate_views (every view has scratch of its own) or after the view was uploaded again. */
int  srh_mvs_mrf_dims(srh_context *ctx, int *w, int *h, int *top_k);
int  srh_mvs_mrf_state(srh_context *ctx, int w, int h, int top_k, int32_t *labels, double *data_costs, double *messages);

/* ---- depth map -> point cloud ----
 * The output side of the path (the reference keeps only the PLY writer, multiviewstereo.cpp:291-315,
 * and the per-view coverage figure it logs, :402-421).  For every pixel of `slot` whose mask is WHITE and whose depth is
 * finite: the 3-D point unproject((x+0.5)/scale, (y+0.5)/scale) cut at that depth by pointFromDepth with the camera's
 * principal direction and centre -- the construction both cross-checks use (twoviewstereo.cpp:612-614,
 * multiviewstereo.cpp:688-692) -- and the pixel's colour.  HOST outputs, pixel order: xyz (w*h*3 doubles, NaN where
 * there is no point), rgb (w*h*3 bytes), valid (w*h bytes); each may be NULL.  *n_points = points produced,
 * *n_masked = pixels with a WHITE mask (coverage = finite depths / masked pixels is what the reference prints). */
int  srh_view_point_cloud(srh_context *ctx, int slot, const srh_params *p, double *xyz_out, uint8_t *rgb_out,
                          uint8_t *valid_out, int64_t *n_points, int64_t *n_masked, int64_t *n_finite);

/* ---- epipolar curves ----
 * TwoViewStereo::epipolarCurve (public member, twoviewstereo.hpp:66-70, twoviewstereo.cpp:999-1054;
 * the GUI's curve preview calls it, stereowidget.cpp:621-672) when mvs == 0, and
 * MultiViewStereo::epipolarCurve (multiviewstereo.cpp:754-810: uniform depths, clipped segments,
 * consecutive duplicates removed) when mvs != 0.  For each of the `nqueries` reference pixels
 * xy[2q], xy[2q+1] of slot `ref_slot`: the candidate pixels in slot `oth_slot`, in the order the
 * reference visits them, written as (x,y) int32 pairs to out_xy + q*2*max_pts; counts[q] receives
 * the curve length (which may exceed max_pts; only max_pts points are written).  The reference
 * pixel's own mask is not consulted (the callers do that).  All pointers are HOST memory. */
int  srh_epipolar_curves(srh_context *ctx, int ref_slot, int oth_slot, const srh_params *p, int mvs,
                         int nqueries, const int32_t *xy, int32_t *out_xy, int max_pts, int32_t *counts);

/* ---- GUI-side users of the camera model ----
 * StereoWidget::epipolarLineItem (gui/widgets/stereowidget.cpp:621-672): the curve preview drawn while the user moves
 * over the left image.  For each of the nqueries pixels xy[2q], xy[2q+1] (image coordinates as the GUI has them: no
 * +0.5, no scale): num_depths UNIFORM depths in [min_depth, max_depth], plane Plane3d(principal direction, depth),
 * projection into `oth_slot`; a point becomes a vertex of the path when it is more than one pixel from the last one.
 * out_xy: nqueries*num_depths*2 doubles (the vertices of query q start at q*num_depths*2), counts[q] = vertices
 * (0: nothing to draw).  HOST pointers. */
int  srh_epipolar_preview(srh_context *ctx, int ref_slot, int oth_slot, double min_depth, double max_depth, int num_depths,
                          int nqueries, const double *xy, double *out_xy, int32_t *counts);
/* RefractionCalibration::error / totalError (stereo/refractioncalibration.cpp:175-199, 408-469): for npairs
 * correspondences (p1 in slot1's image, p2 in slot2's): the ray-ray distance scaled to image space, per pair
 * (err_out, may be NULL: diff()'s single component -- RefractionCalibration::error is its absolute value), the sum
 * of its squares in pair order (*total) and total / npairs (*average, NaN for no pairs); the cameras are those
 * uploaded with the slots (re-upload a view to try other interface parameters: the LM loop above stays host code). */
int  srh_refraction_error(srh_context *ctx, int slot1, int slot2, int npairs, const double *p1_xy, const double *p2_xy,
                          double *err_out, double *total, double *average);

/* ---- multi-GPU exchange (RCCL over xGMI; one process and one context per GPU) ----
 * srh_comm_unique_id: rank 0 creates the 128-byte id and hands it to the other ranks by any
 * means (MPI, files, torch.distributed ...).  srh_comm_init is collective.  librccl is loaded on
 * first use; SRH_E_UNSUPPORTED if it is absent. */
#define SRH_COMM_ID_BYTES 128
int  srh_comm_unique_id(void *id_out);
/* The communicator is non-blocking (ncclConfig_t::blocking = 0): a rank that never arrives at the rendezvous, or a call
 * that stays "in progress" longer than the timeout (srh_comm_set_timeout_ms, default 120 000 ms), returns SRH_E_DEVICE
 * after the communicator has been aborted -- a missing rank ends the job with an error instead of hanging it. */
int  srh_comm_init(srh_context *ctx, int nranks, int rank, const void *id);
int  srh_comm_set_timeout_ms(int ms);
/* NCCL_VERSION_CODE of the loaded librccl (ncclGetVersion), 0 when there is none; what the context's communicator
 * spans (0 ranks, rank -1 without one) */
int  srh_comm_version(void);
int  srh_comm_info(srh_context *ctx, int *nranks, int *rank);
/* Gather the depth map of `slot` (w*h doubles, equal on all ranks) to `root`:
 * recv_dev (DEVICE, nranks*w*h doubles, rank order) is written on the root only. */
int  srh_comm_gather_depth(srh_context *ctx, int slot, int root, void *recv_dev);
/* Every rank receives every rank's map of `slot` (MVS cross-check input). */
int  srh_comm_allgather_depth(srh_context *ctx, int slot, void *recv_dev);
/* The same collective on HOST buffers: every rank contributes `count` doubles and receives nranks*count (rank
 * order); staged through device memory of the context, RCCL in between.  For callers above the C-ABI that hold
 * several maps per rank in host memory (host/sharded.hpp). */
int  srh_comm_allgather_host(srh_context *ctx, const double *send_host, size_t count, double *recv_host);
/* The one exchange of a sharded MultiViewStereo::runTask, device to device: the nviews views are dealt to the ranks in
 * contiguous balanced shards (view v belongs to the rank r with lo(r) <= v < hi(r), lo(r) = r*(n/world) + min(r, n%world));
 * every rank contributes the depth maps of ITS views, and on return (stream-ordered, nothing is staged through host
 * memory) the depth map of every slot on every rank is its owner's.  Views may differ in size (maps travel padded). */
int  srh_comm_allgather_views(srh_context *ctx, const int32_t *view_slots, int nviews);
int  srh_comm_destroy(srh_context *ctx);

/* ---- measurement ---- */
int  srh_get_stats(srh_context *ctx, srh_stats *out);
/* When enabled every kernel launch is bracketed by hipEvents on the context
 * stream; durations are accumulated per kernel name. */
int  srh_profile_enable(srh_context *ctx, int on);
int  srh_profile_reset(srh_context *ctx);
/* total_ms / launches of kernel `name`; returns SRH_E_INVALID if never launched. */
int  srh_profile_get(srh_context *ctx, const char *name, double *total_ms, int64_t *launches);
/* Writes up to cap bytes of "name total_ms launches\n" lines. */
int  srh_profile_dump(srh_context *ctx, char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* STEREO_RECON_HIP_H */
