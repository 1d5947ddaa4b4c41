/*
 * sr_oracle.h -- CPU restatement (plain C, IEEE double, single thread) of the
 * reference's dense matching-cost / support-weight / WTA path.
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may compile, link, load or call anything in
 * oracle/.  The product (stereoreconstruction_amd/, include/) never does.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic it follows, operation by operation and in the same order.
 * Build with -ffp-contract=off (the reference's x86-64 build has no FMA).
 *
 * Pinning status (details in DESIGN.md "Oracle"):
 *   - images / bilinear sample / gray, LineIterator + clipLine, AdaptiveWeight,
 *     GeodesicWeight: pinned against the reference's own unmodified sources
 *     (util/vectorimage.cpp, util/lineiter.cpp, stereo/adaptiveweight.cpp,
 *     stereo/geodesicweight.cpp) compiled into oracle/_ref, and against the
 *     known-answer vectors of SURVEY.md section 8(c).
 *   - camera / ray / plane geometry, cost_ncc, epipolarCurve, WTA, cross-check,
 *     MVS top-K: the reference sources need Eigen (absent from this image), so
 *     they cannot be compiled here without writing a stand-in library;
 *     PARITY UNPINNED for these rows beyond line-by-line restatement.
 *   - Camera::setP (project-file cameras): the reference factorises with Eigen's
 *     HouseholderQR (absent): PARITY UNPINNED at the last ulp; Eigen's unblocked
 *     Householder algorithm is restated and cross-checked against LAPACK.
 *   - refractive projection: reference calls GSL gsl_poly_complex_solve
 *     (gsl 1.14, absent): PARITY UNPINNED; restated as the physical root of the
 *     same quartic on [0, r].
 */
#ifndef SR_ORACLE_H
#define SR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Snapshot of project/camera.hpp:168-185.  3x3 matrices are row-major. */
typedef struct sro_camera {
	double K[9], Kinv[9], R[9], Rinv[9];
	double t[3], C[3];
	double dist[5];          /* k1,k2,p1,p2,k3 (OpenCV order, camera.cpp:406-415) */
	int32_t is_distorted;
	int32_t is_refractive;
	double plane_normal[3];  /* unit normal of the interface, camera-local */
	double plane_dist;
	double refr_index;
	double pdir[3];          /* principleRay().direction(), global (camera.cpp:292-298) */
} sro_camera;

/* An already-scaled image as VectorImage::fromQImage would hold it
 * (util/vectorimage.cpp:48-64) plus the "pixel == WHITE" predicate of its mask. */
typedef struct sro_image {
	int32_t w, h;
	const uint8_t *rgba;     /* w*h*4 bytes, order R,G,B,A */
	const uint8_t *mask;     /* w*h bytes, 1 <=> mask.pixel(x,y)==WHITE; NULL = all 1 */
} sro_image;

enum { SRO_WEIGHT_ADAPTIVE = 0, SRO_WEIGHT_GEODESIC = 1 };

/* Every hard-coded constant of the path, with the reference's defaults
 * (SURVEY.md 8(a) "Constants that must become runtime parameters"). */
typedef struct sro_params {
	double  min_depth, max_depth;
	int32_t num_depth_levels;
	int32_t window_radius;        /* TwoView 5 (twoviewstereo.cpp:66), MVS 2 (multiviewstereo.cpp:91) */
	double  image_scale;
	int32_t weight_kind;          /* typedef GeodesicWeight WeightFunc */
	int32_t geodesic_iters;       /* NUM_ITERS 3 (geodesicweight.cpp:36) */
	double  geodesic_sigma;       /* 50 (geodesicweight.cpp:33) */
	double  geodesic_init;        /* 1e6 (geodesicweight.cpp:68) */
	double  adaptive_color_sigma; /* 10 (adaptiveweight.cpp:26) */
	double  weight_cutoff;        /* 1e-10 */
	/* TwoView */
	double  bad_ret;              /* 1000 */
	double  max_color_diff;       /* 120 */
	double  second_best_factor;   /* 0.95 */
	double  wta_margin;           /* 1e-10 */
	double  inconsistency_thresh; /* 1 */
	/* MVS */
	double  peak_threshold;       /* 0.95 (multiviewstereo.cpp:589) */
	double  cross_check_threshold;
	double  neighbour_min_dot;    /* 0.2 (multiviewstereo.cpp:343) */
	int32_t top_k;                /* K 9 */
	int32_t num_neighbours;       /* NUM_NEIGHBOURING_VIEWS 3 */
} sro_params;

void sro_params_twoview_defaults(sro_params *p);
void sro_params_mvs_defaults(sro_params *p);

/* Camera::set(K,R,t) (camera.cpp:225-240) + setLensDistortion (:302-313) +
 * setPlane/setRefractiveIndex (:326-344).  dist may be NULL. */
/* Camera::setP -> updateOthers (project/camera.cpp:251-288): P /= |P[2,0:3]|^2, RQ factorisation of the left
 * 3x3 block through Eigen's HouseholderQR of (reverseRows*M)^T (restated: Eigen is a third-party dependency
 * absent from this image -- parity unpinned), positive-diagonal fix, then the Camera::set
 * tail (orthonormalize, inverses, C, principal ray).  P is row-major 3x4. */
void sro_camera_set_p(sro_camera *cam, const double P[12], const double dist[5],
                      const double plane_normal[3], double plane_dist, double refr_index);
void sro_camera_set(sro_camera *cam, const double K[9], const double R[9], const double t[3],
                    const double dist[5],
                    const double plane_normal[3], double plane_dist, double refr_index);

/* --- util/vectorimage --- */
/* VectorImage::sample (vectorimage.cpp:129-155). returns 0 if INVALID. out = r,g,b */
int    sro_image_sample(const sro_image *img, double x, double y, double out_rgb[3]);
/* RGBA::toGray (vectorimage.hpp:60-62) */
double sro_to_gray(double r, double g, double b);

/* --- util/lineiter --- */
/* LineIterator 4-arg (clip=0) or 6-arg (clip=1) ctor taking doubles truncated to
 * int, iterated to exhaustion (lineiter.hpp:32-118, lineiter.cpp:35-88).
 * clip=2: the 4-arg walk in the bounded form sro_epipolar_curve uses for TwoView curves:
 * by the closed form of the Bresenham state only the part whose major coordinate lies
 * in [0,w) x [0,h) is visited (the callers drop every other point anyway).
 * Writes up to max_pts (x,y) pairs; returns the number of points of the line. */
int sro_line_points(double x0, double y0, double x1, double y1, int clip, int w, int h,
                    int32_t *out_xy, int max_pts);

/* --- weights --- */
/* out[(row+r)*(2r+1) + (col+r)] == weightFunc(row, col) after init_weights(img,cx,cy)
 * (geodesicweight.cpp:59-131 / adaptiveweight.cpp:33-79). */
void sro_weights(const sro_image *img, int cx, int cy, const sro_params *p, double *out);

/* --- project/camera + util/ray --- */
/* Camera::unproject(x,y) (camera.cpp:423-459): ray = src[3], dir[3] in global space */
void sro_unproject(const sro_camera *cam, double x, double y, double src[3], double dir[3]);
/* Camera::project(Vector3d&) (camera.cpp:380-419): p in/out; returns 0 on failure */
int  sro_project(const sro_camera *cam, double p[3]);
/* Ray3d::closestPoints (util/ray.cpp:53-74) */
void sro_closest_points(const double s1[3], const double d1[3], const double s2[3], const double d2[3],
                        double p1[3], double p2[3]);

/* The 3-D point of pixel (x,y) at depth `depth`: unproject((x+0.5)/scale, (y+0.5)/scale), then pointFromDepth with
 * the camera's principal direction and centre -- the construction both cross-checks use (twoviewstereo.cpp:612-614,
 * multiviewstereo.cpp:688-692).  Returns 0 when pointFromDepth fails (e.g. depth -1, the "no peak" value). */
int  sro_back_project(const sro_camera *cam, const sro_params *p, int x, int y, double depth, double out[3]);

/* StereoWidget::epipolarLineItem (gui/widgets/stereowidget.cpp:621-672), the GUI's curve preview: `num_depths` UNIFORM
 * depths in [min_depth, max_depth]; the pixel (px,py) is unprojected as given (no +0.5, no scale); the plane of a
 * depth is Plane3d(principal direction, depth) -- normal . x = depth in global coordinates, not the plane through the
 * camera centre that pointFromDepth builds; a projected point becomes a vertex when it is MORE than one pixel (squared
 * distance > 1) from the last vertex; the first projected point is the path's start.  out_xy: up to max_pts (x,y)
 * doubles; returns the number of vertices (0: nothing drawn). */
int  sro_epipolar_preview(const sro_camera *left, const sro_camera *right, double px, double py,
                          double min_depth, double max_depth, int num_depths, double *out_xy, int max_pts);
/* RefractiveCalibrationFunction::diff (stereo/refractioncalibration.cpp:175-199): the error of one correspondence
 * (p1 in view 1, p2 in view 2): distance of the two unprojected rays, scaled to image space by 0.5*fx/z of the
 * mid-point in both cameras.  totalError (:408-447) sums the squares. */
double sro_refraction_pair_error(const sro_camera *v1, const sro_camera *v2, const double p1[2], const double p2[2]);

/* --- epipolar curves --- */
/* TwoViewStereo::epipolarCurve (twoviewstereo.cpp:999-1054) when mvs==0 (non-uniform
 * labels, no clipping, no de-duplication); MultiViewStereo::epipolarCurve
 * (multiviewstereo.cpp:754-810) when mvs==1.  Returns the curve length; writes up
 * to max_pts (x,y) pairs. */
int sro_epipolar_curve(const sro_camera *refcam, const sro_camera *othcam, const sro_image *oth,
                       const sro_params *p, int mvs, int x, int y, int32_t *out_xy, int max_pts);

/* --- costs --- */
/* TwoViewStereo::cost_ncc (twoviewstereo.cpp:909-977); weights from sro_weights */
double sro_twoview_cost_ncc(const sro_image *ref, const sro_image *oth, const double *weights,
                            const sro_params *p, int x1, int y1, int x2, int y2);
/* free cost_ncc of multiviewstereo.cpp:113-189 */
double sro_mvs_cost_ncc(const sro_image *ref, const sro_image *oth, const double *weights,
                        const sro_params *p, int x1, int y1, int x2, int y2);

/* Optional per-pixel diagnostics of the WTA scan */
typedef struct sro_diag {
	int32_t *win_xy;      /* w*h*2: winning candidate pixel, -1,-1 if none */
	double  *min_cost;    /* w*h */
	double  *second_cost; /* w*h */
	int64_t  n_eval;      /* number of cost evaluations performed */
} sro_diag;

/* One pass of TwoViewStereo::computeCostVolumes, non-MRF body
 * (twoviewstereo.cpp:260-333 with ref=left / :431-501 with ref=right), rows [y0,y1).
 * depth is the full w*h map; only rows [y0,y1) are written. */
void sro_twoview_wta(const sro_image *ref, const sro_image *oth,
                     const sro_camera *refcam, const sro_camera *othcam,
                     const sro_params *p, int y0, int y1, double *depth, sro_diag *diag);

/* TwoViewStereo::crossCheck (twoviewstereo.cpp:596-672): left pass then right pass, in place */
void sro_twoview_cross_check(int w, int h, const sro_camera *lcam, const sro_camera *rcam,
                             const sro_params *p, double *depth_left, double *depth_right);

/* MultiViewStereo::runTask neighbour selection (multiviewstereo.cpp:335-360).
 * out_neigh[v*num_neighbours + k], out_count[v]. */
void sro_mvs_neighbours(int nviews, const sro_camera *cams, const sro_params *p,
                        int32_t *out_neigh, int32_t *out_count);

/* MultiViewStereo::computeInitialEstimate, non-MRF result
 * (multiviewstereo.cpp:524-604,654-660), rows [y0,y1).  peaks (optional) receives
 * top_k (cost,depth) pairs per pixel: peaks[(pix*top_k + k)*2 + {0,1}], ascending. */
void sro_mvs_initial_estimate(int nviews, const sro_image *imgs, const sro_camera *cams,
                              int view, const int32_t *neigh, int nneigh,
                              const sro_params *p, int y0, int y1, double *depth, double *peaks,
                              int64_t *n_eval);

/* MultiViewStereo::crossCheck(view) (multiviewstereo.cpp:666-729), in place on
 * depths[view]; reads the other views' current maps. depths[v] is w[v]*h[v]. */
void sro_mvs_cross_check(int nviews, const sro_image *imgs, const sro_camera *cams, int view,
                         const sro_params *p, double *const *depths);

/* ---- MRF branch of computeInitialEstimate (multiviewstereo.cpp:481-516, 610-652; CONFIG+=mrf) ----
 * PARITY UNPINNED: the reference links a third-party library here (`LIBS *= -lMRF`, StereoReconstruction.pro:100-103,
 * <MRF/mrf.h>: the Middlebury "MRF energy minimization software" with V. Kolmogorov's TRW-S, no version pinned, not in
 * the reference tree, not in this image).  What follows restates the PUBLISHED algorithm (Kolmogorov, "Convergent
 * tree-reweighted message passing for energy minimization", PAMI 2006, sequential TRW-S on a 4-connected grid,
 * gamma = 1/2, forward then backward sweep, solution read off by a forward sweep) in the form that library's grid /
 * general-smoothness code takes, in double; the energies, the stopping rule and the label -> depth rule are the
 * reference's own lines.  No reference fixture or compiled reference can pin it. */
typedef struct sro_mrf_params {
	double  beta, lambda;         /* BETA 1, LAMBDA 1 (multiviewstereo.cpp:98-99) */
	double  phi_u, psi_u;         /* PHIU 0.5, PSIU 0.002 (:100-101) */
	int32_t max_iters;            /* numIters 50 (:631): at most max_iters + 1 sweeps */
	double  min_energy_drop;      /* 5 (:641) */
} sro_mrf_params;
typedef struct sro_mrf_info {
	int32_t iterations;           /* optimize(1) calls made */
	double  energy_initial;       /* totalEnergy() after clearAnswer() */
	double  energy_final;
	double  lower_bound;          /* TRW-S bound of the last sweep */
} sro_mrf_info;
void sro_mrf_params_defaults(sro_mrf_params *m);
/* dataCost / smoothnessCost of CostFunction (multiviewstereo.cpp:485-515); labels 0..K-1 = peaks, K = unknown */
double sro_mrf_data_cost(const sro_mrf_params *m, int K, const double *pixel_peaks, int label);
double sro_mrf_smooth_cost(const sro_mrf_params *m, int K, const double *peaks1, const double *peaks2, int l1, int l2);
/* peaks: w*h*K (cost, depth) pairs as sro_mvs_initial_estimate writes them; mask: w*h (NULL = all WHITE).
 * depth (w*h, in/out): written where mask is WHITE (:645-652).  labels (optional, w*h), data_costs (optional
 * OVERRIDE, w*h*(K+1): when given these are used instead of evaluating dataCost -- lets a test feed the engine
 * another exp()), messages (optional out, w*h*2*(K+1): [pixel][right, down][label]). */
void sro_mvs_mrf(int w, int h, int K, const double *peaks, const uint8_t *mask, const sro_mrf_params *m,
                 double *depth, int32_t *labels, const double *data_costs, double *messages, sro_mrf_info *info);

#ifdef __cplusplus
}
#endif
#endif
