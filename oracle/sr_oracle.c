/*
 * sr_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE
 * ONLY (see sr_oracle.h).  Plain C99, IEEE double, one thread, no FMA
 * contraction (build with -ffp-contract=off).
 *
 * All file:line citations are relative to /root/reference.
 */
#include "sr_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* small 3-vector helpers.  The reference uses Eigen::Vector3d; Eigen is a
 * third-party dependency that is absent from this image, so the association
 * order of its 3-term reductions is unpinned.  We use left-to-right. */

static double dot3(const double a[3], const double b[3]) {
	return (a[0]*b[0] + a[1]*b[1]) + a[2]*b[2];
}
static double norm3(const double a[3]) { return sqrt(dot3(a, a)); }
static void normalize3(double a[3]) {
	const double n = norm3(a);
	a[0] /= n; a[1] /= n; a[2] /= n;
}
static void matvec3(const double M[9], const double v[3], double out[3]) {
	double r0 = (M[0]*v[0] + M[1]*v[1]) + M[2]*v[2];
	double r1 = (M[3]*v[0] + M[4]*v[1]) + M[5]*v[2];
	double r2 = (M[6]*v[0] + M[7]*v[1]) + M[8]*v[2];
	out[0] = r0; out[1] = r1; out[2] = r2;
}

static int iszero_eps(double x) { return (x <= 1e-10 && x >= -1e-10); } /* camera.cpp:51-52 */

/* ------------------------------------------------------------------ */

void sro_params_twoview_defaults(sro_params *p) {
	memset(p, 0, sizeof(*p));
	p->min_depth = 10; p->max_depth = 100; p->num_depth_levels = 100;
	p->window_radius = 5;              /* twoviewstereo.cpp:66 */
	p->image_scale = 1.0;
	p->weight_kind = SRO_WEIGHT_GEODESIC; /* twoviewstereo.cpp:84 */
	p->geodesic_iters = 3; p->geodesic_sigma = 50.0; p->geodesic_init = 1000000.0;
	p->adaptive_color_sigma = 10.0;
	p->weight_cutoff = 1e-10;
	p->bad_ret = 1000; p->max_color_diff = 120; /* twoviewstereo.cpp:65,74 */
	p->second_best_factor = 0.95; p->wta_margin = 1e-10; p->inconsistency_thresh = 1; /* :78-79 */
	p->peak_threshold = 0.95; p->cross_check_threshold = 1.0; p->neighbour_min_dot = 0.2;
	p->top_k = 9; p->num_neighbours = 3;
}

void sro_params_mvs_defaults(sro_params *p) {
	sro_params_twoview_defaults(p);
	p->window_radius = 2;              /* multiviewstereo.cpp:91 */
}

/* ------------------------------------------------------------------ */
/* project/camera.cpp:140-160 orthonormalize (Gram-Schmidt on columns) */
static void orthonormalize(double M[9]) {
	for (int i = 0; i < 3; ++i) {
		double accum[3] = {0, 0, 0};
		for (int j = 0; j < i; ++j) {
			double vi[3] = {M[0*3+i], M[1*3+i], M[2*3+i]};
			double vj[3] = {M[0*3+j], M[1*3+j], M[2*3+j]};
			double scale = dot3(vi, vj) / dot3(vj, vj);
			accum[0] += vj[0]*scale; accum[1] += vj[1]*scale; accum[2] += vj[2]*scale;
		}
		double c[3] = {M[0*3+i] - accum[0], M[1*3+i] - accum[1], M[2*3+i] - accum[2]};
		normalize3(c);
		M[0*3+i] = c[0]; M[1*3+i] = c[1]; M[2*3+i] = c[2];
	}
	for (int k = 0; k < 9; ++k)
		if (-1e-10 < M[k] && M[k] < 1e-10) M[k] = 0.0;
}

/* general 3x3 inverse by cofactors (Eigen Matrix3d::inverse(), unpinned third party) */
static void inverse3(const double m[9], double out[9]) {
	const double c00 = m[4]*m[8] - m[5]*m[7];
	const double c01 = m[5]*m[6] - m[3]*m[8];
	const double c02 = m[3]*m[7] - m[4]*m[6];
	const double det = (m[0]*c00 + m[1]*c01) + m[2]*c02;
	const double invdet = 1.0 / det;
	out[0] = c00*invdet;
	out[1] = (m[2]*m[7] - m[1]*m[8])*invdet;
	out[2] = (m[1]*m[5] - m[2]*m[4])*invdet;
	out[3] = c01*invdet;
	out[4] = (m[0]*m[8] - m[2]*m[6])*invdet;
	out[5] = (m[2]*m[3] - m[0]*m[5])*invdet;
	out[6] = c02*invdet;
	out[7] = (m[1]*m[6] - m[0]*m[7])*invdet;
	out[8] = (m[0]*m[4] - m[1]*m[3])*invdet;
}

/* Eigen::HouseholderQR<Matrix3d> (unblocked path, Eigen/src/QR/HouseholderQR.h + Householder/Householder.h):
 * a is row-major 3x3, overwritten with R above/on the diagonal and the essential parts below; tau[k] out */
static void householder_qr3(double a[9], double tau[3])
{
	for (int k = 0; k < 3; ++k) {
		/* makeHouseholderInPlace on column k, rows k..2 */
		const double c0 = a[k*3+k];
		double tail = 0;
		for (int i = k + 1; i < 3; ++i) tail += a[i*3+k]*a[i*3+k];
		double beta;
		if (tail <= DBL_MIN) {
			tau[k] = 0; beta = c0;
			for (int i = k + 1; i < 3; ++i) a[i*3+k] = 0;
		} else {
			beta = sqrt(c0*c0 + tail);
			if (c0 >= 0) beta = -beta;
			for (int i = k + 1; i < 3; ++i) a[i*3+k] /= (c0 - beta);
			tau[k] = (beta - c0)/beta;
		}
		a[k*3+k] = beta;
		/* applyHouseholderOnTheLeft to the block rows k..2, cols k+1..2 */
		for (int j = k + 1; j < 3; ++j) {
			double tmp = 0;
			for (int i = k + 1; i < 3; ++i) tmp += a[i*3+k]*a[i*3+j];
			tmp += a[k*3+j];
			a[k*3+j] -= tau[k]*tmp;
			for (int i = k + 1; i < 3; ++i) a[i*3+j] -= tau[k]*a[i*3+k]*tmp;
		}
	}
}

/* householderQ(): Q = H0 H1 H2, H_k = I - tau_k v_k v_k^T, v_k = (0.., 1, essential_k) */
static void householder_q3(const double a[9], const double tau[3], double q[9])
{
	for (int i = 0; i < 9; ++i) q[i] = (i % 4 == 0) ? 1.0 : 0.0;
	for (int k = 2; k >= 0; --k) {
		for (int j = k; j < 3; ++j) {
			double tmp = 0;
			for (int i = k + 1; i < 3; ++i) tmp += a[i*3+k]*q[i*3+j];
			tmp += q[k*3+j];
			q[k*3+j] -= tau[k]*tmp;
			for (int i = k + 1; i < 3; ++i) q[i*3+j] -= tau[k]*a[i*3+k]*tmp;
		}
	}
}

void sro_camera_set_p(sro_camera *cam, const double Pin[12], const double dist[5],
                      const double plane_normal[3], double plane_dist, double refr_index)
{
	/* Camera::updateOthers, camera.cpp:251-288 */
	double P[12];
	const double n2 = (Pin[8]*Pin[8] + Pin[9]*Pin[9]) + Pin[10]*Pin[10];
	for (int i = 0; i < 12; ++i) P[i] = Pin[i] / n2;
	/* qrMatrix = (reverseRows * M)^T : element (i,j) = M(2-j, i) */
	double a[9], tau[3], q[9];
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j)
			a[i*3+j] = P[(2-j)*4 + i];
	householder_qr3(a, tau);
	householder_q3(a, tau, q);
	/* R_ = reverseRows * Q^T ; K_ = reverseRows * Rtri^T * reverseRows */
	double R[9], K[9];
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j) {
			R[i*3+j] = q[j*3 + (2-i)];
			const int ri = 2 - j, rj = 2 - i;             /* Rtri^T(2-i, 2-j) = Rtri(2-j, 2-i) */
			K[i*3+j] = (ri <= rj) ? a[ri*3+rj] : 0.0;
		}
	for (int axis = 2; axis >= 0; --axis) {
		if (K[axis*3+axis] < 0) {
			K[axis*3+axis] = -K[axis*3+axis];
			for (int j = 0; j < 3; ++j) R[axis*3+j] = -R[axis*3+j];
		}
		if (K[axis*3+2] < 0) K[axis*3+2] = -K[axis*3+2];
	}
	/* orthonormalize(R_), Kinv_, Rinv_, t_ = Kinv_ * P_.col(3), C_ = -Rinv_ * t_: the tail of sro_camera_set,
	 * which orthonormalises R itself; t needs the inverse of K first */
	double Kinv[9], p3[3] = { P[3], P[7], P[11] }, t[3];
	inverse3(K, Kinv);
	matvec3(Kinv, p3, t);
	sro_camera_set(cam, K, R, t, dist, plane_normal, plane_dist, refr_index);
}

void sro_camera_set(sro_camera *cam, const double K[9], const double R[9], const double t[3],
                    const double dist[5],
                    const double plane_normal[3], double plane_dist, double refr_index)
{
	memset(cam, 0, sizeof(*cam));
	/* Camera::set, camera.cpp:225-240 */
	memcpy(cam->K, K, sizeof(cam->K));
	memcpy(cam->R, R, sizeof(cam->R));
	memcpy(cam->t, t, sizeof(cam->t));
	orthonormalize(cam->R);
	inverse3(cam->K, cam->Kinv);
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j)
			cam->Rinv[i*3+j] = cam->R[j*3+i];
	{
		double nt[3] = {-t[0], -t[1], -t[2]};
		matvec3(cam->Rinv, nt, cam->C);
	}
	/* updatePrincipleRay, camera.cpp:292-298 */
	{
		double tcol[3] = {cam->K[2], cam->K[5], cam->K[8]};
		double tc[3] = {tcol[0]/tcol[2], tcol[1]/tcol[2], tcol[2]/tcol[2]};
		double dir[3];
		matvec3(cam->Kinv, tc, dir);
		normalize3(dir);
		matvec3(cam->Rinv, dir, cam->pdir);
		normalize3(cam->pdir);
	}
	/* setLensDistortion, camera.cpp:302-313 */
	if (dist) {
		memcpy(cam->dist, dist, sizeof(cam->dist));
		cam->is_distorted = !iszero_eps(dist[0]) || !iszero_eps(dist[1]) || !iszero_eps(dist[2])
		                 || !iszero_eps(dist[3]) || !iszero_eps(dist[4]);
	}
	/* setPlane / setRefractiveIndex, camera.cpp:326-344; Plane3d(normal,d) normalises (plane.hpp:32) */
	cam->plane_normal[0] = 0; cam->plane_normal[1] = 0; cam->plane_normal[2] = 1;
	cam->plane_dist = 0; cam->refr_index = 1.0;
	if (plane_normal) {
		cam->plane_normal[0] = plane_normal[0];
		cam->plane_normal[1] = plane_normal[1];
		cam->plane_normal[2] = plane_normal[2];
		normalize3(cam->plane_normal);
		cam->plane_dist = plane_dist;
		cam->refr_index = refr_index;
	}
	cam->is_refractive = (!iszero_eps(cam->refr_index - 1) && !iszero_eps(cam->plane_dist));
}

/* ------------------------------------------------------------------ */
/* util/vectorimage */

/* VectorImage::pixel (vectorimage.cpp:115-119): 0 => INVALID */
static int img_pixel(const sro_image *img, int x, int y, double rgb[3]) {
	if (x < 0 || y < 0 || x >= img->w || y >= img->h) return 0;
	const uint8_t *p = img->rgba + ((size_t)y*img->w + x)*4;
	rgb[0] = p[0]; rgb[1] = p[1]; rgb[2] = p[2];
	return 1;
}

/* mask.pixel(x,y) == WHITE with mask as a 0/1 plane; OOB => INVALID != WHITE */
static int mask_white(const sro_image *img, int x, int y) {
	if (x < 0 || y < 0 || x >= img->w || y >= img->h) return 0;
	return img->mask ? (img->mask[(size_t)y*img->w + x] == 1) : 1;
}

double sro_to_gray(double r, double g, double b) { /* vectorimage.hpp:60-62 */
	return (0.11*r + 0.59*g + 0.3*b);
}

int sro_image_sample(const sro_image *img, double x, double y, double out[3]) {
	/* vectorimage.cpp:129-155 */
	if (x >= 0 && y >= 0 && x + 1 < img->w && y + 1 < img->h) {
		int ix = (int)x, iy = (int)y;
		double dx = x - ix, dy = y - iy;
		double r[3] = {0.0, 0.0, 0.0}, t[3], s;
		const int w = img->w;
		const uint8_t *d = img->rgba;
		const int idx[4] = { ix + iy*w, ix + (iy + 1)*w, ix + 1 + iy*w, ix + 1 + (iy + 1)*w };
		const double sc[4] = { (1 - dx)*(1 - dy), (1 - dx)*dy, dx*(1 - dy), dx*dy };
		for (int k = 0; k < 4; ++k) {
			s = sc[k];
			t[0] = d[(size_t)idx[k]*4 + 0]; t[1] = d[(size_t)idx[k]*4 + 1]; t[2] = d[(size_t)idx[k]*4 + 2];
			t[0] *= s; t[1] *= s; t[2] *= s;
			r[0] += t[0]; r[1] += t[1]; r[2] += t[2];
		}
		out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
		return 1;
	}
	return 0;
}

/* forward: defined with the camera model below */
void sro_unproject(const sro_camera *cam, double px, double py, double src[3], double dir[3]);
static int point_from_depth(const double src[3], const double dir[3], const double normal[3], double depth, double p[3]);

int sro_back_project(const sro_camera *cam, const sro_params *p, int x, int y, double depth, double out[3]) {
	double src[3], dir[3];
	sro_unproject(cam, (x + 0.5) / p->image_scale, (y + 0.5) / p->image_scale, src, dir);
	double pt[3] = { cam->C[0], cam->C[1], cam->C[2] };
	if (!point_from_depth(src, dir, cam->pdir, depth, pt)) return 0;
	out[0] = pt[0]; out[1] = pt[1]; out[2] = pt[2];
	return 1;
}

/* ------------------------------------------------------------------ */
/* util/lineiter */

/* double -> int as the implicit conversion at the LineIterator call sites
 * (twoviewstereo.cpp:1028, multiviewstereo.cpp:783).  Out-of-range / NaN is
 * undefined behaviour in the reference; we saturate at +-2^29 (NaN -> 0) so that
 * the restatement and the GPU agree and integer deltas cannot overflow. */
static int trunc_sat(double v) {
	if (!(v == v)) return 0;
	if (v >= 536870912.0) return 536870912;
	if (v <= -536870912.0) return -536870912;
	return (int)v;
}

static int out_code(int x, int y, int w, int h) { /* lineiter.cpp:35-42 */
	int code = 0;
	if (x < 0) code |= 1; else if (x > w) code |= 2;
	if (y < 0) code |= 4; else if (y > h) code |= 8;
	return code;
}

/* lineiter.cpp:44-88 (Cohen-Sutherland, integer division).  Products are done in
 * 64 bits; the reference's 32-bit products overflow (UB) only for coordinates
 * beyond +-2^15 or so, which fixtures avoid. */
static int clip_line(int *x0, int *y0, int *x1, int *y1, int w, int h) {
	w--; h--;
	int oc0 = out_code(*x0, *y0, w, h);
	int oc1 = out_code(*x1, *y1, w, h);
	for (;;) {
		if (!(oc0 | oc1)) return 1;
		if (oc0 & oc1) return 0;
		int64_t x = 0, y = 0;
		const int oc = oc0 ? oc0 : oc1;
		const int64_t X0 = *x0, Y0 = *y0, X1 = *x1, Y1 = *y1;
		if (oc & 8)      { x = X0 + ((X1 - X0)*(h - Y0))/(Y1 - Y0); y = h; }
		else if (oc & 4) { x = X0 + ((X1 - X0)*(0 - Y0))/(Y1 - Y0); y = 0; }
		else if (oc & 2) { y = Y0 + ((Y1 - Y0)*(w - X0))/(X1 - X0); x = w; }
		else if (oc & 1) { y = Y0 + ((Y1 - Y0)*(0 - X0))/(X1 - X0); x = 0; }
		if (oc == oc0) { *x0 = (int)x; *y0 = (int)y; oc0 = out_code(*x0, *y0, w, h); }
		else           { *x1 = (int)x; *y1 = (int)y; oc1 = out_code(*x1, *y1, w, h); }
	}
}

typedef void (*line_cb)(int x, int y, void *user);

/* LineIterator (lineiter.hpp:32-118): visits every point of the line.  When
 * bound_w > 0 the walk is restricted, by the closed form of the Bresenham
 * state, to the part whose major coordinate lies in the image -- points outside
 * are never kept by the callers (mask.pixel is INVALID there), so the visible
 * result is unchanged while pathological segments stay bounded. */
static int line_walk(int x0, int y0, int x1, int y1, int bound_w, int bound_h, line_cb cb, void *user) {
	/* initialize(), lineiter.hpp:96-111 */
	const int steep = abs(y1 - y0) > abs(x1 - x0);
	int t;
	if (steep) { t = x0; x0 = y0; y0 = t; t = x1; x1 = y1; y1 = t; }
	if (x0 > x1) { t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
	const int deltax = x1 - x0;
	const int deltay = abs(y1 - y0);
	const int ystep = (y0 < y1 ? 1 : -1);
	const int total = deltax + 1;
	/* reset(): error = deltax/2 */
	int error = deltax / 2;
	int x = x0, y = y0;
	int xend = x1;
	if (bound_w > 0) {
		const int hi = (steep ? bound_h : bound_w) - 1;
		if (xend > hi) xend = hi;
		if (x < 0) {
			/* jump k steps ahead: after k steps 0 <= error_k < deltax and
			 * error_k = error_0 - k*deltay + s_k*deltax, s_k = #y-steps. */
			const int64_t k = -(int64_t)x0;
			if (k > deltax) return total;
			const int64_t s = (k*deltay - error + deltax - 1) / deltax;
			error = (int)(error - k*deltay + s*deltax);
			y = (int)(y0 + ystep*s);
			x = 0;
		}
	}
	for (; x <= xend; ) {           /* hasNext(): x <= x1 */
		if (steep) cb(y, x, user); else cb(x, y, user);   /* current() */
		++x;                        /* next() */
		error -= deltay;
		if (error < 0) { y += ystep; error += deltax; }
	}
	return total;
}

typedef struct { int32_t *out; int max; int n; } collect_ctx;
static void collect_cb(int x, int y, void *user) {
	collect_ctx *c = (collect_ctx *)user;
	if (c->n < c->max) { c->out[2*c->n] = x; c->out[2*c->n + 1] = y; }
	c->n++;
}

int sro_line_points(double fx0, double fy0, double fx1, double fy1, int clip, int w, int h,
                    int32_t *out_xy, int max_pts)
{
	int x0 = trunc_sat(fx0), y0 = trunc_sat(fy0), x1 = trunc_sat(fx1), y1 = trunc_sat(fy1);
	collect_ctx c = { out_xy, max_pts, 0 };
	if (clip == 2) {
		/* the bounded (jump-ahead) form the TwoView curve uses (sro_epipolar_curve, 4-arg
		 * LineIterator of twoviewstereo.cpp:1028 restricted to the image): no clipping */
		line_walk(x0, y0, x1, y1, w, h, collect_cb, &c);
		return c.n;
	}
	if (clip) {
		if (!clip_line(&x0, &y0, &x1, &y1, w, h)) return 0; /* lineiter.hpp:49-60 */
	}
	line_walk(x0, y0, x1, y1, 0, 0, collect_cb, &c);
	return c.n;
}

/* ------------------------------------------------------------------ */
/* weights */

static void geodesic_weights(const sro_image *img, int cx, int cy, const sro_params *p, double *wt) {
	/* geodesicweight.cpp:59-131 */
	static const int K1[8] = {-1, -1, 0, -1, 1, -1, -1, 0};
	static const int K2[8] = {-1,  1, 0,  1, 1,  1,  1, 0};
	const int radius = p->window_radius;
	const int WS = 2*radius + 1;
	for (int i = 0; i < WS*WS; ++i) wt[i] = p->geodesic_init;
	wt[radius*WS + radius] = 0.0;

	for (int iter = 0; iter < p->geodesic_iters; ++iter) {
		for (int pass = 0; pass < 2; ++pass) {
			const int *KK = pass == 0 ? K1 : K2;
			for (int yi = 0; yi < WS; ++yi) {
				const int y = pass == 0 ? (-radius + yi) : (radius - yi);
				for (int xi = 0; xi < WS; ++xi) {
					const int x = pass == 0 ? (-radius + xi) : (radius - xi);
					double rgb1[3];
					if (!img_pixel(img, cx + x, cy + y, rgb1)) continue;
					double *weight = &wt[(y + radius)*WS + (x + radius)];
					for (int ind = 0; ind < 8; ind += 2) {
						const int dx = KK[ind], dy = KK[ind + 1];
						if (x + dx > radius || y + dy > radius || x + dx < -radius || y + dy < -radius)
							continue;
						double rgb2[3];
						if (img_pixel(img, cx + x + dx, cy + y + dy, rgb2)) {
							rgb2[0] -= rgb1[0]; rgb2[1] -= rgb1[1]; rgb2[2] -= rgb1[2];
							const double diff = sqrt(rgb2[0]*rgb2[0] + rgb2[1]*rgb2[1] + rgb2[2]*rgb2[2]);
							const double cost = wt[(y + dy + radius)*WS + (x + dx + radius)];
							const double cand = cost + diff;
							if (cand < *weight) *weight = cand;   /* std::min(weight, cost+diff) */
						}
					}
				}
			}
		}
	}
	for (int i = 0; i < WS*WS; ++i) wt[i] = exp(-wt[i] / p->geodesic_sigma);
}

static void adaptive_weights(const sro_image *img, int cx, int cy, const sro_params *p, double *wt) {
	/* adaptiveweight.cpp:33-79 */
	const int radius = p->window_radius;
	const int WS = 2*radius + 1;
	double dw[64];
	for (int ind = 0; ind <= radius && ind < 64; ++ind)
		dw[ind] = exp(-ind / (1.0*radius));
	double crgb[3];
	const int cvalid = img_pixel(img, cx, cy, crgb);
	for (int row = -radius; row <= radius; ++row) {
		for (int col = -radius; col <= radius; ++col) {
			double weight = 0.0, rgb[3];
			if (img_pixel(img, cx + col, cy + row, rgb)) {
				if (cvalid) {
					rgb[0] -= crgb[0]; rgb[1] -= crgb[1]; rgb[2] -= crgb[2];
					const double diff = sqrt(rgb[0]*rgb[0] + rgb[1]*rgb[1] + rgb[2]*rgb[2]);
					const double w1 = dw[abs(row)]*dw[abs(col)];
					const double w2 = exp(-diff / p->adaptive_color_sigma);
					weight = w1*w2;
					if (isnan(weight)) weight = 0.0;
				} else {
					weight = 0.0; /* crgb INVALID => NaN => 0 (adaptiveweight.cpp:74-75) */
				}
			}
			wt[(row + radius)*WS + (col + radius)] = weight;
		}
	}
}

void sro_weights(const sro_image *img, int cx, int cy, const sro_params *p, double *out) {
	if (p->weight_kind == SRO_WEIGHT_GEODESIC) geodesic_weights(img, cx, cy, p, out);
	else adaptive_weights(img, cx, cy, p, out);
}

/* ------------------------------------------------------------------ */
/* util/ray, util/plane */

/* intersect(ray, Plane3d(normal, x0), p)  (plane.hpp:34, ray.cpp:78-88) */
static int intersect_plane(const double src[3], const double dir[3],
                           const double pn[3], double pdist, double p[3])
{
	const double nd = dot3(pn, dir);
	if (fabs(nd) < 1e-10) return 0;
	const double x0[3] = { pdist*pn[0], pdist*pn[1], pdist*pn[2] };   /* Plane3d::x0() */
	const double dlt[3] = { x0[0] - src[0], x0[1] - src[1], x0[2] - src[2] };
	const double t = dot3(pn, dlt) / nd;
	if (t < 1e-10) return 0;
	p[0] = src[0] + t*dir[0]; p[1] = src[1] + t*dir[1]; p[2] = src[2] + t*dir[2]; /* Ray3d::point */
	return 1;
}

/* pointFromDepth (twoviewstereo.cpp:987-995 == multiviewstereo.cpp:740-750):
 * Plane3d plane(normal, p + normal*depth); intersect(ray, plane, p) */
static int point_from_depth(const double src[3], const double dir[3], const double normal[3],
                            double depth, double p[3])
{
	double n[3] = { normal[0], normal[1], normal[2] };
	normalize3(n);                                            /* Plane3d ctor, plane.hpp:34 */
	const double x0[3] = { p[0] + normal[0]*depth, p[1] + normal[1]*depth, p[2] + normal[2]*depth };
	const double d = dot3(n, x0);
	return intersect_plane(src, dir, n, d, p);
}

void sro_closest_points(const double s1[3], const double d1[3], const double s2[3], const double d2[3],
                        double p1[3], double p2[3])
{
	/* ray.cpp:53-74 */
	const double w0[3] = { s1[0] - s2[0], s1[1] - s2[1], s1[2] - s2[2] };
	const double a = dot3(d1, d1);
	const double b = dot3(d1, d2);
	const double c = dot3(d2, d2);
	const double d = dot3(d1, w0);
	const double e = dot3(d2, w0);
	const double den = 1.0 / (a*c - b*b);
	const double tl = (b*e - c*d) * den;
	const double tr = (a*e - b*d) * den;
	p1[0] = s1[0]; p1[1] = s1[1]; p1[2] = s1[2];
	p2[0] = s2[0]; p2[1] = s2[1]; p2[2] = s2[2];
	if (tl > 0) { p1[0] += tl*d1[0]; p1[1] += tl*d1[1]; p1[2] += tl*d1[2]; }
	if (tr > 0) { p2[0] += tr*d2[0]; p2[1] += tr*d2[1]; p2[2] += tr*d2[2]; }
}

/* refract (ray.cpp:92-106); src/dir in and out */
static int refract_ray(double src[3], double dir[3], const double pn[3], double pdist, double n) {
	double p[3];
	if (intersect_plane(src, dir, pn, pdist, p)) {
		const double cosI = -(dot3(pn, dir));
		const double cosT2 = 1.0 - (1.0 - cosI*cosI) / (n*n);
		if (cosT2 > 0.0) {
			const double sign = (cosI > 0.0 ? -1.0 : 1.0);
			const double k = cosI + n*sign*sqrt(cosT2);
			src[0] = p[0]; src[1] = p[1]; src[2] = p[2];
			dir[0] = dir[0] + k*pn[0]; dir[1] = dir[1] + k*pn[1]; dir[2] = dir[2] + k*pn[2];
			normalize3(dir);
			return 1;
		}
	}
	return 0;
}

/* ------------------------------------------------------------------ */
/* project/camera */

/* Physical root on [0, r] of the quartic of camera.cpp:111-117.  The reference
 * obtains all four roots from gsl_poly_complex_solve (GSL 1.14, not available:
 * PARITY UNPINNED) and keeps the first real one in (about) [0, r]
 * (camera.cpp:119-135); q(0) = d^2 n^2 r^2 > 0 and q(r) = -(z-d)^2 r^2 < 0, so
 * that interval always brackets the Snell root.  Safeguarded Newton. */
static int quartic_root_0r(double a, double b, double c, double d, double e, double r, double guess, double *root) {
	double lo = 0.0, hi = r;
	const double f0 = e;
	const double fr = (((a*r + b)*r + c)*r + d)*r + e;
	if (!(r > 0.0)) return 0;
	if (!(f0 > 0.0)) { if (f0 == 0.0) { *root = 0.0; return 1; } return 0; }
	if (!(fr < 0.0)) { if (fr == 0.0) { *root = r; return 1; } return 0; }
	double x = guess;
	if (!(x > lo && x < hi)) x = 0.5*(lo + hi);
	for (int it = 0; it < 100; ++it) {
		const double f = (((a*x + b)*x + c)*x + d)*x + e;
		if (f == 0.0) break;
		if (f > 0.0) lo = x; else hi = x;
		const double df = ((4.0*a*x + 3.0*b)*x + 2.0*c)*x + d;
		double xn = x - f/df;
		/* (closed bracket: a step that rounds to nothing leaves xn ON the end point x has just become -- that is convergence,
		 * caught by the step test below; with the open test it sent the iteration to the bracket's mid-point, often r/2, to
		 * converge all over again: 6 % of the roots of the C5 configuration took that detour) */
		if (!(xn >= lo && xn <= hi)) xn = 0.5*(lo + hi);
		const double dx = fabs(xn - x);
		x = xn;
		if (dx <= 1e-15*(fabs(x) + r)) break;
	}
	*root = x;
	return 1;
}

/* projectRefraction (camera.cpp:95-138) */
static int project_refraction(double p[3], const double pn[3], double pdist, double n) {
	double bn[3] = { pn[0], pn[1], pn[2] };
	normalize3(bn);                                           /* linalg.hpp:34 */
	const double s = dot3(bn, p);
	const double proj[3] = { s*bn[0], s*bn[1], s*bn[2] };
	double dir[3] = { p[0] - proj[0], p[1] - proj[1], p[2] - proj[2] };
	const double y = dir[1];
	const double z = norm3(proj);
	const double r = norm3(dir);
	const double d = pdist;
	const double rr = r*r, nn = n*n, dd = d*d;
	normalize3(dir);
	if (dir[0] != dir[0] || dir[1] != dir[1] || dir[2] != dir[2]) return 0; /* r == 0: every test below is false */

	const double qa = nn - 1;
	const double qb = -2*r*(nn - 1);
	const double qc = rr*(nn - 1) + dd*nn - (z - d)*(z - d);
	const double qd = -2*dd*nn*r;
	const double qe = dd*nn*rr;
	double root;
	/* Newton's start: the paraxial Snell point r d / (d + (z - d)/n) -- the small-angle solution of the refraction at the
	 * interface (the air path d and the water path z - d shortened by the index).  r d / z, the straight-line crossing,
	 * is off by the factor n and costs Newton 1.4 more steps on average (3.7 instead of 5.1 over C5's geometry). */
	if (!quartic_root_0r(qa, qb, qc, qd, qe, r, r*d/(d + (z - d)/n), &root)) return 0;

	const double pp[3] = { root*dir[0], root*dir[1], root*dir[2] };
	const double py = pp[1];
	int ok = 0;
	if (py > -1e-3 && y > -1e-3) { if (py < y + 1e-3) ok = 1; }
	else if (py < 1e-3 && y < 1e-3) { if (y < py + 1e-3) ok = 1; }
	if (!ok) return 0;
	p[0] = pp[0] + pdist*pn[0]; p[1] = pp[1] + pdist*pn[1]; p[2] = pp[2] + pdist*pn[2];
	return 1;
}

int sro_project(const sro_camera *cam, double p[3]) {
	/* camera.cpp:380-419 */
	double point[3];
	matvec3(cam->R, p, point);
	point[0] += cam->t[0]; point[1] += cam->t[1]; point[2] += cam->t[2];   /* fromGlobalToLocal :346 */
	if (cam->is_refractive) {
		if (!project_refraction(point, cam->plane_normal, cam->plane_dist, cam->refr_index)) {
			p[0] = p[1] = p[2] = NAN;
			return 0;
		}
	}
	matvec3(cam->K, point, p);
	{ const double z = p[2]; p[0] /= z; p[1] /= z; p[2] /= z; }
	if (cam->is_distorted) {
		const double cx = cam->K[2], cy = cam->K[5], fx = cam->K[0], fy = cam->K[4];
		const double *k = cam->dist;
		double x = p[0], y = p[1];
		x = (x - cx) / fx;
		y = (y - cy) / fy;
		{
			const double r2 = x*x + y*y;
			const double cdist = 1 + ((k[4]*r2 + k[1])*r2 + k[0])*r2;
			x = x*cdist + 2*k[2]*x*y + k[3]*(r2 + 2*x*x);
			y = y*cdist + k[2]*(r2 + 2*y*y) + 2*k[3]*x*y;      /* uses the updated x, as the reference */
		}
		x = fx*x + cx;
		y = fy*y + cy;
		p[0] = x; p[1] = y;
	}
	return 1;
}

void sro_unproject(const sro_camera *cam, double px, double py, double src[3], double dir[3]) {
	/* camera.cpp:423-459 */
	double pp[3] = { px, py, 1.0 };
	if (cam->is_distorted) {
		const double cx = cam->K[2], cy = cam->K[5];
		const double ifx = 1.0 / cam->K[0], ify = 1.0 / cam->K[4];
		const double *k = cam->dist;
		double x = pp[0], y = pp[1];
		const double x0 = x = (x - cx)*ifx;
		const double y0 = y = (y - cy)*ify;
		for (int j = 0; j < 5; j++) {
			const double r2 = x*x + y*y;
			const double icdist = 1.0 / (1 + ((k[4]*r2 + k[1])*r2 + k[0])*r2);
			const double deltaX = 2*k[2]*x*y + k[3]*(r2 + 2*x*x);
			const double deltaY = k[2]*(r2 + 2*y*y) + 2*k[3]*x*y;
			x = (x0 - deltaX)*icdist;
			y = (y0 - deltaY)*icdist;
		}
		x /= ifx; y /= ify;
		x += cx;  y += cy;
		pp[0] = x; pp[1] = y;
	}
	double lsrc[3] = {0, 0, 0}, ldir[3];
	matvec3(cam->Kinv, pp, ldir);
	normalize3(ldir);                                         /* Ray3d ctor, ray.cpp:30-33 */
	if (cam->is_refractive)
		refract_ray(lsrc, ldir, cam->plane_normal, cam->plane_dist, cam->refr_index);
	/* fromLocalToGlobal(Ray3d), camera.cpp:372-376 */
	matvec3(cam->Rinv, ldir, dir);
	{
		const double q[3] = { lsrc[0] - cam->t[0], lsrc[1] - cam->t[1], lsrc[2] - cam->t[2] };
		matvec3(cam->Rinv, q, src);
	}
	normalize3(dir);
}

/* ------------------------------------------------------------------ */
/* epipolar curve */

static double depth_from_label(const sro_params *p, int mvs, int label) {
	double t = label / (p->num_depth_levels - 1.0);
	if (!mvs) t /= (5 - 4*t);                                 /* twoviewstereo.cpp:981-985 */
	return p->min_depth*(1 - t) + p->max_depth*t;             /* multiviewstereo.cpp:733-736 */
}

typedef struct { int32_t *pts; int n, cap; } ptvec;
static void ptvec_push(ptvec *v, int x, int y) {
	if (v->n == v->cap) {
		v->cap = v->cap ? 2*v->cap : 1024;
		v->pts = (int32_t *)realloc(v->pts, (size_t)v->cap*2*sizeof(int32_t));
	}
	v->pts[2*v->n] = x; v->pts[2*v->n + 1] = y; v->n++;
}

typedef struct { const sro_image *mask; ptvec *v; } curve_ctx;
static void curve_cb(int x, int y, void *user) {
	curve_ctx *c = (curve_ctx *)user;
	if (mask_white(c->mask, x, y)) ptvec_push(c->v, x, y);
}

/* twoviewstereo.cpp:999-1054 (mvs=0) / multiviewstereo.cpp:754-810 (mvs=1) */
static void epipolar_curve(const double rsrc[3], const double rdir[3], const double camC[3],
                           const double normal[3], const sro_camera *othcam, const sro_image *oth,
                           const sro_params *p, int mvs, ptvec *curve)
{
	curve->n = 0;
	double x1 = NAN, y1 = NAN;
	curve_ctx ctx = { oth, curve };
	for (int d = 0; d < p->num_depth_levels; ++d) {
		double point[3] = { camC[0], camC[1], camC[2] };
		const double depth = depth_from_label(p, mvs, d);
		if (point_from_depth(rsrc, rdir, normal, depth, point)) {
			if (sro_project(othcam, point)) {
				const double x2 = point[0]*p->image_scale;
				const double y2 = point[1]*p->image_scale;
				if (isnan(x1)) {
					x1 = x2; y1 = y2;
				} else {
					const double dx = x2 - x1, dy = y2 - y1;
					if (dx*dx + dy*dy >= 1) {
						int ix0 = trunc_sat(x1), iy0 = trunc_sat(y1), ix1 = trunc_sat(x2), iy1 = trunc_sat(y2);
						if (mvs) {
							if (clip_line(&ix0, &iy0, &ix1, &iy1, oth->w, oth->h))
								line_walk(ix0, iy0, ix1, iy1, 0, 0, curve_cb, &ctx);
						} else {
							line_walk(ix0, iy0, ix1, iy1, oth->w, oth->h, curve_cb, &ctx);
						}
						x1 = x2; y1 = y2;
					}
				}
			}
		}
	}
	if (mvs) {
		/* std::unique on consecutive equal points, multiviewstereo.cpp:801-807 */
		int m = 0;
		for (int i = 0; i < curve->n; ++i) {
			if (m > 0 && curve->pts[2*(m-1)] == curve->pts[2*i] && curve->pts[2*(m-1)+1] == curve->pts[2*i+1])
				continue;
			curve->pts[2*m] = curve->pts[2*i]; curve->pts[2*m+1] = curve->pts[2*i+1]; m++;
		}
		curve->n = m;
	}
}

int sro_epipolar_curve(const sro_camera *refcam, const sro_camera *othcam, const sro_image *oth,
                       const sro_params *p, int mvs, int x, int y, int32_t *out_xy, int max_pts)
{
	double src[3], dir[3];
	ptvec v = {0, 0, 0};
	sro_unproject(refcam, (x + 0.5) / p->image_scale, (y + 0.5) / p->image_scale, src, dir);
	epipolar_curve(src, dir, refcam->C, refcam->pdir, othcam, oth, p, mvs, &v);
	const int n = v.n;
	for (int i = 0; i < n && i < max_pts; ++i) { out_xy[2*i] = v.pts[2*i]; out_xy[2*i+1] = v.pts[2*i+1]; }
	free(v.pts);
	return n;
}

/* ------------------------------------------------------------------ */
/* costs */

static double sample_gray_int(const sro_image *img, int x, int y, int *valid) {
	double rgb[3];
	*valid = sro_image_sample(img, (double)x, (double)y, rgb);
	return *valid ? sro_to_gray(rgb[0], rgb[1], rgb[2]) : 0.0;
}

double sro_twoview_cost_ncc(const sro_image *left, const sro_image *right, const double *weights,
                            const sro_params *p, int x1, int y1, int x2, int y2)
{
	/* twoviewstereo.cpp:909-977 */
	const int R = p->window_radius, WS = 2*R + 1;
	double meanL = 0, meanR = 0, totalWeight = 0.0;
	for (int row = -R; row <= R; ++row) {
		for (int col = -R; col <= R; ++col) {
			if (!mask_white(left, x1 + col, y1 + row)) continue;
			if (!mask_white(right, x2 + col, y2 + row)) continue;
			int vl, vr;
			const double gl = sample_gray_int(left, x1 + col, y1 + row, &vl);
			if (!vl) continue;
			const double gr = sample_gray_int(right, x2 + col, y2 + row, &vr);
			if (!vr) continue;
			const double weight = weights[(row + R)*WS + (col + R)];
			if (weight > p->weight_cutoff) {
				meanL += weight*gl;
				meanR += weight*gr;
				totalWeight += weight;
			}
		}
	}
	if (totalWeight < 1e-10) return p->bad_ret;
	meanL /= totalWeight;
	meanR /= totalWeight;

	double sum1 = 0, sum2 = 0, sum3 = 0;
	for (int row = -R; row <= R; ++row) {
		for (int col = -R; col <= R; ++col) {
			if (!mask_white(left, x1 + col, y1 + row)) continue;
			if (!mask_white(right, x2 + col, y2 + row)) continue;
			int vl, vr;
			const double gl = sample_gray_int(left, x1 + col, y1 + row, &vl);
			const double gr = sample_gray_int(right, x2 + col, y2 + row, &vr);
			if (!vl) continue;
			if (!vr) continue;
			const double weight = weights[(row + R)*WS + (col + R)];
			if (weight > p->weight_cutoff) {
				const double pgl = weight*gl;
				const double pgr = weight*gr;
				sum1 += (pgl - meanL)*(pgr - meanR);
				sum2 += (pgl - meanL)*(pgl - meanL);
				sum3 += (pgr - meanR)*(pgr - meanR);
			}
		}
	}
	/* min(MAX_COLOR_DIFF, v): std::min(a,b) = (b < a) ? b : a, so NaN v => MAX_COLOR_DIFF */
	const double v = 255*(1.0 - fabs(sum1) / sqrt(sum2 * sum3));
	return (v < p->max_color_diff) ? v : p->max_color_diff;
}

static double pixel_gray(const sro_image *img, int x, int y, int *valid) {
	double rgb[3];
	*valid = img_pixel(img, x, y, rgb);
	return *valid ? sro_to_gray(rgb[0], rgb[1], rgb[2]) : 0.0;
}

double sro_mvs_cost_ncc(const sro_image *img1, const sro_image *img2, const double *weights,
                        const sro_params *p, int x1, int y1, int x2, int y2)
{
	/* multiviewstereo.cpp:113-189 */
	const int R = p->window_radius, WS = 2*R + 1;
	double meanL = 0, meanR = 0, totalWeight = 0.0;
	for (int row = -R; row <= R; ++row) {
		for (int col = -R; col <= R; ++col) {
			int vl, vr;
			const double gl = pixel_gray(img1, x1 + col, y1 + row, &vl);
			if (!vl) continue;
			const double gr = pixel_gray(img2, x2 + col, y2 + row, &vr);
			if (!vr) continue;
			const double weight = weights[(row + R)*WS + (col + R)];
			if (weight > p->weight_cutoff) {
				meanL += weight*gl;
				meanR += weight*gr;
				totalWeight += weight;
			}
		}
	}
	if (totalWeight < 1e-10) return 0;
	meanL /= totalWeight;
	meanR /= totalWeight;

	double sum1 = 0, sum2 = 0, sum3 = 0;
	for (int row = -R; row <= R; ++row) {
		for (int col = -R; col <= R; ++col) {
			int vl, vr;
			const double gl = pixel_gray(img1, x1 + col, y1 + row, &vl);
			if (!vl) continue;
			const double gr = pixel_gray(img2, x2 + col, y2 + row, &vr);
			if (!vr) continue;
			const double weight = weights[(row + R)*WS + (col + R)];
			if (weight > p->weight_cutoff) {
				const double pgl = weight*gl;
				const double pgr = weight*gr;
				sum1 += (pgl - meanL)*(pgr - meanR);
				sum2 += (pgl - meanL)*(pgl - meanL);
				sum3 += (pgr - meanR)*(pgr - meanR);
			}
		}
	}
	if (sum2 * sum3 < 1e-10) return 0;
	return sum1 / sqrt(sum2 * sum3);
}

/* ------------------------------------------------------------------ */
/* TwoView */

/* depth of the mid-point of closest approach in the reference camera frame
 * (twoviewstereo.cpp:287-300, multiviewstereo.cpp:584-593) */
static double candidate_depth(const sro_camera *refcam, const sro_camera *othcam, const sro_params *p,
                              const double rsrc[3], const double rdir[3], int cx, int cy)
{
	double s2[3], d2[3], p1[3], p2[3];
	sro_unproject(othcam, (cx + 0.5) / p->image_scale, (cy + 0.5) / p->image_scale, s2, d2);
	sro_closest_points(rsrc, rdir, s2, d2, p1, p2);
	p1[0] += p2[0]; p1[1] += p2[1]; p1[2] += p2[2];
	p1[0] *= 0.5;   p1[1] *= 0.5;   p1[2] *= 0.5;
	/* fromGlobalToLocal(p1).z() */
	return ((refcam->R[6]*p1[0] + refcam->R[7]*p1[1]) + refcam->R[8]*p1[2]) + refcam->t[2];
}

void sro_twoview_wta(const sro_image *ref, const sro_image *oth,
                     const sro_camera *refcam, const sro_camera *othcam,
                     const sro_params *p, int y0, int y1, double *depth, sro_diag *diag)
{
	/* twoviewstereo.cpp:260-333 (left) / :431-501 (right) */
	const int W = ref->w, H = ref->h;
	const int WS = 2*p->window_radius + 1;
	double *weights = (double *)malloc(sizeof(double)*WS*WS);
	ptvec curve = {0, 0, 0};
	if (y0 < 0) y0 = 0;
	if (y1 > H) y1 = H;
	if (diag) diag->n_eval = 0;
	for (int y = y0; y < y1; ++y) {
		for (int x = 0; x < W; ++x) {
			const size_t pv = (size_t)y*W + x;
			depth[pv] = NAN;
			if (diag) {
				if (diag->win_xy) { diag->win_xy[2*pv] = -1; diag->win_xy[2*pv+1] = -1; }
				if (diag->min_cost) diag->min_cost[pv] = INFINITY;
				if (diag->second_cost) diag->second_cost[pv] = INFINITY;
			}
			if (!mask_white(ref, x, y)) continue;

			sro_weights(ref, x, y, p, weights);
			double rsrc[3], rdir[3];
			sro_unproject(refcam, (x + 0.5) / p->image_scale, (y + 0.5) / p->image_scale, rsrc, rdir);

			double secondBestCost = INFINITY, minCost = INFINITY;
			epipolar_curve(rsrc, rdir, refcam->C, refcam->pdir, othcam, oth, p, 0, &curve);
			for (int i = 0; i < curve.n; ++i) {
				const int cx = curve.pts[2*i], cy = curve.pts[2*i+1];
				const double cost = sro_twoview_cost_ncc(ref, oth, weights, p, x, y, cx, cy);
				if (diag) diag->n_eval++;
				if (cost + p->wta_margin < minCost) {
					secondBestCost = minCost;
					minCost = cost;
					depth[pv] = candidate_depth(refcam, othcam, p, rsrc, rdir, cx, cy);
					if (diag && diag->win_xy) { diag->win_xy[2*pv] = cx; diag->win_xy[2*pv+1] = cy; }
				}
			}
			if (minCost > p->second_best_factor*secondBestCost)
				depth[pv] = INFINITY;
			if (diag) {
				if (diag->min_cost) diag->min_cost[pv] = minCost;
				if (diag->second_cost) diag->second_cost[pv] = secondBestCost;
			}
		}
	}
	free(curve.pts);
	free(weights);
}

/* one direction of twoviewstereo.cpp:604-636 / :638-670 */
static void twoview_cross_check_pass(int W, int H, const sro_camera *cam, const sro_camera *ocam,
                                     const sro_params *p, double *dthis, const double *dother)
{
	const double s = p->image_scale;
	for (int y = 0; y < H; ++y) {
		for (int x = 0; x < W; ++x) {
			double *depth = &dthis[(size_t)y*W + x];
			if (!isfinite(*depth)) continue;
			double rs[3], rd[3];
			sro_unproject(cam, (x + 0.5) / s, (y + 0.5) / s, rs, rd);
			double p1[3] = { cam->C[0], cam->C[1], cam->C[2] };
			if (point_from_depth(rs, rd, cam->pdir, *depth, p1)) {
				double q[3] = { p1[0], p1[1], p1[2] };
				if (sro_project(ocam, q)) {
					const double x2 = q[0]*s, y2 = q[1]*s;
					if (x2 >= 0 && y2 >= 0 && x2 < W && y2 < H) {
						const double odepth = dother[(size_t)((int)y2)*W + (int)x2];
						if (isfinite(odepth)) {
							double rs2[3], rd2[3];
							sro_unproject(ocam, (x2 + 0.5) / s, (y2 + 0.5) / s, rs2, rd2);
							double p2[3] = { ocam->C[0], ocam->C[1], ocam->C[2] };
							if (point_from_depth(rs2, rd2, ocam->pdir, odepth, p2)) {
								const double dv[3] = { p1[0]-p2[0], p1[1]-p2[1], p1[2]-p2[2] };
								const double norm = norm3(dv);
								if (!isfinite(norm) || norm > p->inconsistency_thresh)
									*depth = INFINITY;
							} else *depth = INFINITY;
						} else *depth = INFINITY;
					} else *depth = INFINITY;
				} else *depth = INFINITY;
			}
		}
	}
}

void sro_twoview_cross_check(int w, int h, const sro_camera *lcam, const sro_camera *rcam,
                             const sro_params *p, double *depth_left, double *depth_right)
{
	/* twoviewstereo.cpp:596-672: the right pass reads the already-filtered left map */
	twoview_cross_check_pass(w, h, lcam, rcam, p, depth_left, depth_right);
	twoview_cross_check_pass(w, h, rcam, lcam, p, depth_right, depth_left);
}

/* ------------------------------------------------------------------ */
/* MVS */

typedef struct { double d; int idx; } near_view;
static int near_cmp(const void *a, const void *b) {
	const near_view *x = (const near_view *)a, *y = (const near_view *)b;
	if (x->d < y->d) return -1;
	if (x->d > y->d) return 1;
	return (x->idx > y->idx) - (x->idx < y->idx);
}

void sro_mvs_neighbours(int nviews, const sro_camera *cams, const sro_params *p,
                        int32_t *out_neigh, int32_t *out_count)
{
	/* multiviewstereo.cpp:335-360 */
	near_view *nv = (near_view *)malloc(sizeof(near_view)*(nviews > 0 ? nviews : 1));
	for (int v = 0; v < nviews; ++v) {
		int n = 0;
		for (int v2 = 0; v2 < nviews; ++v2) {
			if (v == v2) continue;
			if (fabs(dot3(cams[v].pdir, cams[v2].pdir)) > p->neighbour_min_dot) {
				const double dv[3] = { cams[v].C[0]-cams[v2].C[0], cams[v].C[1]-cams[v2].C[1], cams[v].C[2]-cams[v2].C[2] };
				nv[n].d = dot3(dv, dv);
				nv[n].idx = v2;
				n++;
			}
		}
		int end = n;
		if (p->num_neighbours < n) {
			qsort(nv, n, sizeof(near_view), near_cmp);
			end = p->num_neighbours;
		}
		for (int k = 0; k < end && k < p->num_neighbours; ++k) out_neigh[v*p->num_neighbours + k] = nv[k].idx;
		out_count[v] = end < p->num_neighbours ? end : p->num_neighbours;
	}
	free(nv);
}

typedef struct { double c, z; } peak_pair;
static int peak_cmp(const void *a, const void *b) {
	const peak_pair *x = (const peak_pair *)a, *y = (const peak_pair *)b;
	if (x->c < y->c) return -1;
	if (x->c > y->c) return 1;
	if (x->z < y->z) return -1;
	if (x->z > y->z) return 1;
	return 0;
}

void sro_mvs_initial_estimate(int nviews, const sro_image *imgs, const sro_camera *cams,
                              int view, const int32_t *neigh, int nneigh,
                              const sro_params *p, int y0, int y1, double *depth, double *peaks_out,
                              int64_t *n_eval)
{
	/* multiviewstereo.cpp:524-604 and the non-MRF tail :654-660 */
	(void)nviews;
	const sro_image *image = &imgs[view];
	const sro_camera *cam = &cams[view];
	const int W = image->w, H = image->h;
	const int WS = 2*p->window_radius + 1;
	const int K = p->top_k;
	double *weights = (double *)malloc(sizeof(double)*WS*WS);
	ptvec curve = {0, 0, 0};
	peak_pair *peaks = NULL; int npeaks = 0, cappeaks = 0;
	int64_t evals = 0;
	if (y0 < 0) y0 = 0;
	if (y1 > H) y1 = H;
	for (int y = y0; y < y1; ++y) {
		for (int x = 0; x < W; ++x) {
			const size_t pv = (size_t)y*W + x;
			depth[pv] = INFINITY;
			npeaks = 0;
			if (cappeaks < K) { cappeaks = K + 1024; peaks = (peak_pair *)realloc(peaks, sizeof(peak_pair)*cappeaks); }
			for (int k = 0; k < K; ++k) { peaks[k].c = 0; peaks[k].z = -1; }
			npeaks = K;
			if (peaks_out)
				for (int k = 0; k < K; ++k) { peaks_out[(pv*K + k)*2] = 0; peaks_out[(pv*K + k)*2 + 1] = -1; }
			if (!mask_white(image, x, y)) continue;

			sro_weights(image, x, y, p, weights);
			double rsrc[3], rdir[3];
			sro_unproject(cam, (x + 0.5) / p->image_scale, (y + 0.5) / p->image_scale, rsrc, rdir);

			for (int ni = 0; ni < nneigh; ++ni) {
				const int v2 = neigh[ni];
				epipolar_curve(rsrc, rdir, cam->C, cam->pdir, &cams[v2], &imgs[v2], p, 1, &curve);
				for (int i = 0; i < curve.n; ++i) {
					const int cx = curve.pts[2*i], cy = curve.pts[2*i+1];
					const double cost = sro_mvs_cost_ncc(image, &imgs[v2], weights, p, x, y, cx, cy);
					evals++;
					if (cost > p->peak_threshold) {
						if (npeaks == cappeaks) { cappeaks *= 2; peaks = (peak_pair *)realloc(peaks, sizeof(peak_pair)*cappeaks); }
						peaks[npeaks].c = cost;
						peaks[npeaks].z = candidate_depth(cam, &cams[v2], p, rsrc, rdir, cx, cy);
						npeaks++;
					}
				}
			}
			qsort(peaks, npeaks, sizeof(peak_pair), peak_cmp);   /* std::sort of pairs, :600 */
			const peak_pair *last = peaks + (npeaks - K);           /* keep the last K, :602 */
			if (peaks_out)
				for (int k = 0; k < K; ++k) { peaks_out[(pv*K + k)*2] = last[k].c; peaks_out[(pv*K + k)*2 + 1] = last[k].z; }
			depth[pv] = last[K - 1].z;                              /* peakPairs[y][x].back().second, :658 */
		}
	}
	if (n_eval) *n_eval = evals;
	free(peaks);
	free(curve.pts);
	free(weights);
}

void sro_mvs_cross_check(int nviews, const sro_image *imgs, const sro_camera *cams, int view,
                         const sro_params *p, double *const *depths)
{
	/* multiviewstereo.cpp:666-729 */
	const sro_camera *cam = &cams[view];
	const int W = imgs[view].w, H = imgs[view].h;
	const double s = p->image_scale;
	for (int y = 0; y < H; ++y) {
		for (int x = 0; x < W; ++x) {
			double *depth = &depths[view][(size_t)y*W + x];
			if (!isfinite(*depth)) continue;
			double rs[3], rd[3];
			sro_unproject(cam, (x + 0.5) / s, (y + 0.5) / s, rs, rd);
			double p1[3] = { cam->C[0], cam->C[1], cam->C[2] };
			if (point_from_depth(rs, rd, cam->pdir, *depth, p1)) {
				int found = 0;
				for (int v2 = 0; v2 < nviews; ++v2) {
					if (v2 == view) continue;
					const sro_camera *oc = &cams[v2];
					const int W2 = imgs[v2].w, H2 = imgs[v2].h;
					double q[3] = { p1[0], p1[1], p1[2] };
					if (sro_project(oc, q)) {
						const double x2 = q[0]*s, y2 = q[1]*s;
						if (x2 >= 0 && y2 >= 0 && x2 < W2 && y2 < H2) {
							const double odepth = depths[v2][(size_t)((int)y2)*W2 + (int)x2];
							if (isfinite(odepth)) {
								double rs2[3], rd2[3];
								sro_unproject(oc, (x2 + 0.5) / s, (y2 + 0.5) / s, rs2, rd2);
								double p2[3] = { oc->C[0], oc->C[1], oc->C[2] };
								if (point_from_depth(rs2, rd2, oc->pdir, odepth, p2)) {
									const double dv[3] = { p1[0]-p2[0], p1[1]-p2[1], p1[2]-p2[2] };
									const double norm = norm3(dv);
									if (isfinite(norm) && norm < p->cross_check_threshold) { found = 1; break; }
								}
							}
						}
					}
				}
				if (!found) *depth = NAN;
			}
		}
	}
}

/* ------------------------------------------------------------------ */
/* GUI-side users of the camera model */

int sro_epipolar_preview(const sro_camera *left, const sro_camera *right, double px, double py,
                         double min_depth, double max_depth, int num_depths, double *out_xy, int max_pts)
{
	/* stereowidget.cpp:621-672 */
	double src[3], dir[3];
	sro_unproject(left, px, py, src, dir);
	double n[3] = { left->pdir[0], left->pdir[1], left->pdir[2] };
	normalize3(n);                                            /* Plane3d(normal, d): normal normalised */
	int nv = 0, first = 1;
	double p1x = NAN, p1y = 0;
	for (int k = 0; k < num_depths; ++k) {
		const double t = k / (num_depths - 1.0);
		const double depth = min_depth*(1 - t) + max_depth*t;
		double p2[3];
		if (!intersect_plane(src, dir, n, depth, p2)) continue;
		if (!sro_project(right, p2)) continue;
		if (isnan(p1x)) { p1x = p2[0]; p1y = p2[1]; }
		const double dx = p2[0] - p1x, dy = p2[1] - p1y;
		if ((dx*dx + dy*dy) > 1) {                            /* (p2 - p1).squaredNorm() > 1; z components are both 1 */
			if (first) {
				if (nv < max_pts) { out_xy[2*nv] = p1x; out_xy[2*nv + 1] = p1y; }
				++nv; first = 0;
			}
			if (nv < max_pts) { out_xy[2*nv] = p2[0]; out_xy[2*nv + 1] = p2[1]; }
			++nv;
			p1x = p2[0]; p1y = p2[1];
		}
	}
	return nv;
}

double sro_refraction_pair_error(const sro_camera *v1, const sro_camera *v2, const double p1[2], const double p2[2]) {
	/* refractioncalibration.cpp:175-199 */
	double s1[3], d1[3], s2[3], d2[3], q1[3], q2[3];
	sro_unproject(v1, p1[0], p1[1], s1, d1);
	sro_unproject(v2, p2[0], p2[1], s2, d2);
	sro_closest_points(s1, d1, s2, d2, q1, q2);
	const double df[3] = { q1[0] - q2[0], q1[1] - q2[1], q1[2] - q2[2] };
	const double out = sqrt(dot3(df, df));
	const double mid[3] = { (q1[0] + q2[0])*0.5, (q1[1] + q2[1])*0.5, (q1[2] + q2[2])*0.5 };
	const double z1 = ((v1->R[6]*mid[0] + v1->R[7]*mid[1]) + v1->R[8]*mid[2]) + v1->t[2];
	const double z2 = ((v2->R[6]*mid[0] + v2->R[7]*mid[1]) + v2->R[8]*mid[2]) + v2->t[2];
	const double e1 = (0.5 * v1->K[0] * out) / z1;
	const double e2 = (0.5 * v2->K[0] * out) / z2;
	return e1 + e2;
}

/* ------------------------------------------------------------------ */
/* MRF branch of MultiViewStereo::computeInitialEstimate -- PARITY UNPINNED, see sr_oracle.h */

void sro_mrf_params_defaults(sro_mrf_params *m) {
	m->beta = 1; m->lambda = 1; m->phi_u = 0.5; m->psi_u = 0.002;      /* multiviewstereo.cpp:98-101 */
	m->max_iters = 50; m->min_energy_drop = 5;                        /* :631, :641 */
}

double sro_mrf_data_cost(const sro_mrf_params *m, int K, const double *pk, int label) {
	/* multiviewstereo.cpp:485-497 */
	if (label == K) return m->phi_u;
	if (pk[2*label + 1] < 0) return m->lambda;
	return m->lambda*exp(-m->beta * pk[2*label]);
}

double sro_mrf_smooth_cost(const sro_mrf_params *m, int K, const double *pk1, const double *pk2, int l1, int l2) {
	/* multiviewstereo.cpp:499-514 */
	if (l1 == K && l2 == K) return 0.0;
	if (l1 == K || l2 == K) return m->psi_u;
	const double z1 = pk1[2*l1 + 1], z2 = pk2[2*l2 + 1];
	if (z1 < 0 || z2 < 0) return 2*m->psi_u;
	return 2.0 * fabs(z1 - z2) / (z1 + z2);
}

typedef struct {
	int w, h, K, L;
	const double *peaks;
	const sro_mrf_params *m;
	double *D, *M;               /* D: n*L; M: n*2*L, [pixel][0: edge to x+1, 1: edge to y+1][label] */
	int32_t *ans;
} trws_t;

static double trws_V(const trws_t *t, int p, int q, int lp, int lq) {
	return sro_mrf_smooth_cost(t->m, t->K, t->peaks + (size_t)p*t->K*2, t->peaks + (size_t)q*t->K*2, lp, lq);
}

/* new message over the edge p -> q, written over the stored (reverse) message; returns the constant taken out */
static double trws_update(const trws_t *t, double *M, const double *Di, int p, int q) {
	const int L = t->L;
	double buf[64], delta = 0;
	for (int ks = 0; ks < L; ks++) buf[ks] = 0.5*Di[ks] - M[ks];
	for (int kd = 0; kd < L; kd++) {
		double vmin = buf[0] + trws_V(t, p, q, 0, kd);
		for (int ks = 1; ks < L; ks++) {
			const double v = buf[ks] + trws_V(t, p, q, ks, kd);
			if (vmin > v) vmin = v;
		}
		M[kd] = vmin;
		if (kd == 0 || delta > vmin) delta = vmin;
	}
	for (int kd = 0; kd < L; kd++) M[kd] -= delta;
	return delta;
}

static void trws_gather(const trws_t *t, int x, int y, double *Di) {
	const int L = t->L, w = t->w, h = t->h, n = y*w + x;
	const double *M = t->M + (size_t)n*2*L;
	for (int k = 0; k < L; k++) Di[k] = t->D[(size_t)n*L + k];
	if (x > 0)     for (int k = 0; k < L; k++) Di[k] += (M - 2*L)[k];                 /* (x-1,y) -> (x,y) */
	if (y > 0)     for (int k = 0; k < L; k++) Di[k] += (M - (size_t)2*w*L + L)[k];   /* (x,y-1) -> (x,y) */
	if (x < w - 1) for (int k = 0; k < L; k++) Di[k] += M[k];                         /* (x+1,y) -> (x,y) */
	if (y < h - 1) for (int k = 0; k < L; k++) Di[k] += (M + L)[k];                   /* (x,y+1) -> (x,y) */
}

static double trws_sweep(trws_t *t) {
	const int L = t->L, w = t->w, h = t->h;
	double Di[64], lower = 0;
	for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {                         /* forward */
		const int n = y*w + x;
		double *M = t->M + (size_t)n*2*L;
		trws_gather(t, x, y, Di);
		if (x < w - 1) trws_update(t, M, Di, n, n + 1);
		if (y < h - 1) trws_update(t, M + L, Di, n, n + w);
	}
	for (int y = h - 1; y >= 0; y--) for (int x = w - 1; x >= 0; x--) {               /* backward */
		const int n = y*w + x;
		double *M = t->M + (size_t)n*2*L;
		trws_gather(t, x, y, Di);
		double vmin = Di[0];
		for (int k = 1; k < L; k++) if (vmin > Di[k]) vmin = Di[k];
		for (int k = 0; k < L; k++) Di[k] -= vmin;
		lower += vmin;
		if (x > 0) lower += trws_update(t, M - 2*L, Di, n, n - 1);
		if (y > 0) lower += trws_update(t, M - (size_t)2*w*L + L, Di, n, n - w);
	}
	for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {                         /* read the labels off */
		const int n = y*w + x;
		const double *M = t->M + (size_t)n*2*L;
		for (int k = 0; k < L; k++) Di[k] = t->D[(size_t)n*L + k];
		if (x > 0)     for (int k = 0; k < L; k++) Di[k] += trws_V(t, n - 1, n, t->ans[n - 1], k);
		if (y > 0)     for (int k = 0; k < L; k++) Di[k] += trws_V(t, n - w, n, t->ans[n - w], k);
		if (x < w - 1) for (int k = 0; k < L; k++) Di[k] += M[k];
		if (y < h - 1) for (int k = 0; k < L; k++) Di[k] += (M + L)[k];
		double best = Di[0];
		int a = 0;
		for (int k = 1; k < L; k++) if (best > Di[k]) { best = Di[k]; a = k; }
		t->ans[n] = a;
	}
	return lower;
}

static double trws_energy(const trws_t *t) {
	const int L = t->L, w = t->w, h = t->h;
	double data = 0, smooth = 0;
	for (int n = 0; n < w*h; n++) data += t->D[(size_t)n*L + t->ans[n]];
	for (int y = 0; y < h; y++) for (int x = 1; x < w; x++) { const int n = y*w + x; smooth += trws_V(t, n, n - 1, t->ans[n], t->ans[n - 1]); }
	for (int y = 1; y < h; y++) for (int x = 0; x < w; x++) { const int n = y*w + x; smooth += trws_V(t, n, n - w, t->ans[n], t->ans[n - w]); }
	return data + smooth;
}

void sro_mvs_mrf(int w, int h, int K, const double *peaks, const uint8_t *mask, const sro_mrf_params *m,
                 double *depth, int32_t *labels, const double *data_costs, double *messages, sro_mrf_info *info) {
	trws_t t;
	const int L = K + 1;
	const size_t n = (size_t)w*h;
	t.w = w; t.h = h; t.K = K; t.L = L; t.peaks = peaks; t.m = m;
	t.D = (double *)malloc(n*L*sizeof(double));
	t.M = (double *)calloc(n*2*L, sizeof(double));                     /* initialize(): messages zero */
	t.ans = (int32_t *)calloc(n, sizeof(int32_t));                     /* clearAnswer(): label 0 */
	if (data_costs) memcpy(t.D, data_costs, n*L*sizeof(double));
	else for (size_t p = 0; p < n; p++) for (int l = 0; l < L; l++) t.D[p*L + l] = sro_mrf_data_cost(m, K, peaks + p*K*2, l);

	/* multiviewstereo.cpp:627-641 */
	double energy = trws_energy(&t), prev = 0.0, lower = 0.0;
	const double e0 = energy;
	int num_iters = m->max_iters, iters = 0;
	do {
		prev = energy;
		lower = trws_sweep(&t);
		energy = trws_energy(&t);
		++iters;
	} while (prev - energy > m->min_energy_drop && num_iters-- > 0);

	/* :645-652 */
	for (size_t p = 0; p < n; p++) if (!mask || mask[p]) {
		const int label = t.ans[p];
		const double d = (label == K ? INFINITY : peaks[(p*K + label)*2 + 1]);
		depth[p] = d > 0 ? d : INFINITY;
	}
	if (labels) memcpy(labels, t.ans, n*sizeof(int32_t));
	if (messages) memcpy(messages, t.M, n*2*L*sizeof(double));
	if (info) { info->iterations = iters; info->energy_initial = e0; info->energy_final = energy; info->lower_bound = lower; }
	free(t.D); free(t.M); free(t.ans);
}
