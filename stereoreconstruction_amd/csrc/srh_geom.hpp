// srh_geom.hpp -- camera / ray / plane / line-rasteriser math of the stereo path,
// usable from host and gfx950 device code.  IEEE double throughout, evaluated in
// the reference's operation order; the translation unit is built with
// -ffp-contract=off so no multiply-add is fused.
//
// Reference rows: #4 Camera::unproject (project/camera.cpp:423-459),
// #5 Camera::project / projectRefraction (camera.cpp:95-138, 380-419),
// #6/#14 depthFromLabel / pointFromDepth (twoviewstereo.cpp:981-995,
// multiviewstereo.cpp:733-750), #7 LineIterator / clipLine (util/lineiter.hpp:32-118,
// util/lineiter.cpp:35-88), #10 Ray3d::closestPoints (util/ray.cpp:53-74),
// intersect / refract (util/ray.cpp:78-106).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "stereo_recon_hip.h"

#define SRH_HD __host__ __device__ __forceinline__

namespace srh {

struct Vec3 { double x, y, z; };

SRH_HD Vec3 v3(double x, double y, double z) { Vec3 r; r.x = x; r.y = y; r.z = z; return r; }
SRH_HD Vec3 operator+(Vec3 a, Vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
SRH_HD Vec3 operator-(Vec3 a, Vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
SRH_HD Vec3 operator*(double s, Vec3 a) { return v3(s*a.x, s*a.y, s*a.z); }
SRH_HD Vec3 operator*(Vec3 a, double s) { return v3(a.x*s, a.y*s, a.z*s); }
// 3-term reductions associate left to right (Eigen's order is unpinned third-party
// arithmetic)
SRH_HD double dot(Vec3 a, Vec3 b) { return (a.x*b.x + a.y*b.y) + a.z*b.z; }
SRH_HD double norm(Vec3 a) { return sqrt(dot(a, a)); }
SRH_HD Vec3 normalized(Vec3 a) { const double n = norm(a); return v3(a.x/n, a.y/n, a.z/n); }
SRH_HD Vec3 matvec(const double *M, Vec3 v) {
	return v3((M[0]*v.x + M[1]*v.y) + M[2]*v.z,
	          (M[3]*v.x + M[4]*v.y) + M[5]*v.z,
	          (M[6]*v.x + M[7]*v.y) + M[8]*v.z);
}
SRH_HD Vec3 load3(const double *p) { return v3(p[0], p[1], p[2]); }
SRH_HD bool isnan_d(double v) { return !(v == v); }
SRH_HD bool isfinite_d(double v) { return fabs(v) <= 1.7976931348623157e308; }

struct Ray { Vec3 src, dir; };

// ---- util/ray.cpp:78-88 with Plane3d::x0() = dist*normal (util/plane.hpp:42)
SRH_HD bool intersect_plane(const Ray &R, Vec3 pn, double pdist, Vec3 &p) {
	const double nd = dot(pn, R.dir);
	if (fabs(nd) < 1e-10) return false;
	const Vec3 x0 = pdist*pn;
	const double t = dot(pn, x0 - R.src) / nd;
	if (t < 1e-10) return false;
	p = R.src + t*R.dir;
	return true;
}

// ---- pointFromDepth: Plane3d plane(normal, p + normal*depth); intersect(ray, plane, p)
SRH_HD bool point_from_depth(const Ray &ray, Vec3 normal, double depth, Vec3 &p) {
	const Vec3 n = normalized(normal);
	const Vec3 x0 = p + normal*depth;
	const double d = dot(n, x0);
	return intersect_plane(ray, n, d, p);
}

// ---- util/ray.cpp:53-74
SRH_HD void closest_points(const Ray &a, const Ray &b, Vec3 &p1, Vec3 &p2) {
	const Vec3 w0 = a.src - b.src;
	const double A = dot(a.dir, a.dir);
	const double B = dot(a.dir, b.dir);
	const double C = dot(b.dir, b.dir);
	const double D = dot(a.dir, w0);
	const double E = dot(b.dir, w0);
	const double den = 1.0 / (A*C - B*B);
	const double tl = (B*E - C*D) * den;
	const double tr = (A*E - B*D) * den;
	p1 = a.src;
	p2 = b.src;
	if (tl > 0) p1 = p1 + tl*a.dir;
	if (tr > 0) p2 = p2 + tr*b.dir;
}

// ---- util/ray.cpp:92-106
SRH_HD bool refract(Ray &R, Vec3 pn, double pdist, double n) {
	Vec3 p;
	if (intersect_plane(R, pn, pdist, p)) {
		const double cosI = -(dot(pn, R.dir));
		const double cosT2 = 1.0 - (1.0 - cosI*cosI) / (n*n);
		if (cosT2 > 0.0) {
			const double sign = (cosI > 0.0 ? -1.0 : 1.0);
			const double k = cosI + n*sign*sqrt(cosT2);
			R.src = p;
			R.dir = normalized(R.dir + k*pn);
			return true;
		}
	}
	return false;
}

// ---- camera.cpp:95-138.  The reference takes all four roots of the quartic from
// GSL and keeps the first real one in (about) [0, r]; q(0) > 0 > q(r), so that
// interval brackets the physical (Snell) root, found here by safeguarded Newton.
SRH_HD bool quartic_root_0r(double a, double b, double c, double d, double e, double r, double guess, double &root) {
	double lo = 0.0, hi = r;
	const double f0 = e;
	const double fr = (((a*r + b)*r + c)*r + d)*r + e;
	if (!(r > 0.0)) return false;
	if (!(f0 > 0.0)) { if (f0 == 0.0) { root = 0.0; return true; } return false; }
	if (!(fr < 0.0)) { if (fr == 0.0) { root = r; return true; } return false; }
	double x = guess;
	if (!(x > lo && x < hi)) x = 0.5*(lo + hi);
	for (int it = 0; it < 100; ++it) {
		const double f = (((a*x + b)*x + c)*x + d)*x + e;
		if (f == 0.0) break;
		if (f > 0.0) lo = x; else hi = x;
		const double df = ((4.0*a*x + 3.0*b)*x + 2.0*c)*x + d;
		double xn = x - f/df;
		// (closed bracket: a step that rounds to nothing leaves xn ON the end point x has just become -- that is convergence,
		// caught by the step test below; with the open test it sent the iteration to the bracket's mid-point, often r/2, to
		// converge all over again: 6 % of the roots of the C5 configuration took that detour)
		if (!(xn >= lo && xn <= hi)) xn = 0.5*(lo + hi);
		const double dx = fabs(xn - x);
		x = xn;
		if (dx <= 1e-15*(fabs(x) + r)) break;
	}
	root = x;
	return true;
}

// bn_pre: normalized(pn) computed by the caller once (it depends on the camera only), or null
SRH_HD bool project_refraction(Vec3 &p, Vec3 pn, double pdist, double n, const Vec3 *bn_pre = nullptr) {
	const Vec3 bn = bn_pre ? *bn_pre : normalized(pn);
	const Vec3 proj = dot(bn, p)*bn;
	Vec3 dir = p - proj;
	const double y = dir.y;
	const double z = norm(proj);
	const double r = norm(dir);
	const double d = pdist;
	const double rr = r*r, nn = n*n, dd = d*d;
	dir = v3(dir.x/r, dir.y/r, dir.z/r);                      // normalized(dir): its norm is r, the same sqrt of the same sum
	if (isnan_d(dir.x) || isnan_d(dir.y) || isnan_d(dir.z)) return false;

	const double qa = nn - 1;
	const double qb = -2*r*(nn - 1);
	const double qc = rr*(nn - 1) + dd*nn - (z - d)*(z - d);
	const double qd = -2*dd*nn*r;
	const double qe = dd*nn*rr;
	double root;
	// (Newton's start: the paraxial Snell point, see the oracle's project_refraction)
	if (!quartic_root_0r(qa, qb, qc, qd, qe, r, r*d/(d + (z - d)/n), root)) return false;

	const Vec3 pp = root*dir;
	const double py = pp.y;
	bool ok = false;
	if (py > -1e-3 && y > -1e-3) { if (py < y + 1e-3) ok = true; }
	else if (py < 1e-3 && y < 1e-3) { if (y < py + 1e-3) ok = true; }
	if (!ok) return false;
	p = pp + pdist*pn;
	return true;
}

// ---- Camera::project (camera.cpp:380-419); p in/out (x, y, 1)
SRH_HD bool cam_project(const srh_camera &cam, Vec3 &p, const Vec3 *plane_bn = nullptr) {
	Vec3 point = matvec(cam.R, p) + load3(cam.t);
	if (cam.is_refractive) {
		if (!project_refraction(point, load3(cam.plane_normal), cam.plane_dist, cam.refr_index, plane_bn)) {
			p = v3(NAN, NAN, NAN);
			return false;
		}
	}
	p = matvec(cam.K, point);
	{ const double z = p.z; p = v3(p.x/z, p.y/z, p.z/z); }
	if (cam.is_distorted) {
		const double cx = cam.K[2], cy = cam.K[5], fx = cam.K[0], fy = cam.K[4];
		const double *k = cam.dist;
		double x = p.x, y = p.y;
		x = (x - cx) / fx;
		y = (y - cy) / fy;
		{
			const double r2 = x*x + y*y;
			const double cdist = 1 + ((k[4]*r2 + k[1])*r2 + k[0])*r2;
			x = x*cdist + 2*k[2]*x*y + k[3]*(r2 + 2*x*x);
			y = y*cdist + k[2]*(r2 + 2*y*y) + 2*k[3]*x*y;   // updated x, as the reference
		}
		x = fx*x + cx;
		y = fy*y + cy;
		p.x = x; p.y = y;
	}
	return true;
}

// ---- Camera::unproject(x, y) (camera.cpp:423-459) -> ray in global space
SRH_HD Ray cam_unproject(const srh_camera &cam, double px, double py) {
	double x = px, y = py;
	if (cam.is_distorted) {
		const double cx = cam.K[2], cy = cam.K[5];
		const double ifx = 1.0 / cam.K[0], ify = 1.0 / cam.K[4];
		const double *k = cam.dist;
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
	}
	Ray ray;
	ray.src = v3(0, 0, 0);
	ray.dir = normalized(matvec(cam.Kinv, v3(x, y, 1.0)));
	if (cam.is_refractive)
		refract(ray, load3(cam.plane_normal), cam.plane_dist, cam.refr_index);
	Ray out;
	out.dir = normalized(matvec(cam.Rinv, ray.dir));          // fromLocalToGlobal, camera.cpp:372-376
	out.src = matvec(cam.Rinv, ray.src - load3(cam.t));
	return out;
}

// fromGlobalToLocal(p).z()  (camera.cpp:346-348)
SRH_HD double cam_local_z(const srh_camera &cam, Vec3 p) {
	return ((cam.R[6]*p.x + cam.R[7]*p.y) + cam.R[8]*p.z) + cam.t[2];
}

// ---- depthFromLabel: twoviewstereo.cpp:981-985 (non-uniform) / multiviewstereo.cpp:733-736
SRH_HD double depth_from_label(const srh_params &P, bool mvs, int label) {
	double t = label / (P.num_depth_levels - 1.0);
	if (!mvs) t /= (5 - 4*t);
	return P.min_depth*(1 - t) + P.max_depth*t;
}

// ---- integer rasteriser ----------------------------------------------------
// double -> int as at the LineIterator call sites.  NaN / out of range is UB in
// the reference; saturate at +-2^29 (NaN -> 0) so deltas cannot overflow.
SRH_HD int trunc_sat(double v) {
#ifdef __HIP_DEVICE_COMPILE__
	// branch-free: clamp (a NaN comes out of the clamp as a bound), truncate, then NaN -> 0.  Four of these per kept label
	// were a quarter of the scan kernel's instructions as compare-and-branch chains.
	const double c = __builtin_fmin(__builtin_fmax(v, -536870912.0), 536870912.0);
	const int i = (int)c;
	return v == v ? i : 0;
#else
	if (isnan_d(v)) return 0;
	if (v >= 536870912.0) return 536870912;
	if (v <= -536870912.0) return -536870912;
	return (int)v;
#endif
}

SRH_HD int out_code(int x, int y, int w, int h) {             // lineiter.cpp:35-42
	int code = 0;
	if (x < 0) code |= 1; else if (x > w) code |= 2;
	if (y < 0) code |= 4; else if (y > h) code |= 8;
	return code;
}

// lineiter.cpp:44-88; 64-bit products (the reference's 32-bit ones overflow, UB,
// only for coordinates far outside any image)
SRH_HD bool clip_line(int &x0, int &y0, int &x1, int &y1, int w, int h) {
	w--; h--;
	int oc0 = out_code(x0, y0, w, h);
	int oc1 = out_code(x1, y1, w, h);
	for (int guard = 0; guard < 16; ++guard) {
		if (!(oc0 | oc1)) return true;
		if (oc0 & oc1) return false;
		const int oc = oc0 ? oc0 : oc1;
#ifdef __HIP_DEVICE_COMPILE__
		// One truncating division p0 + ((p1 - p0)*(lim - q0))/(q1 - q0) for the four cases (the GPU has no 64-bit integer
		// divide: four inlined software divisions were 600 instructions per round for the whole wave).  The factors are
		// 32-bit; when their product is below 2^52 it is exact in a double, the FP64 quotient truncates to the integer
		// quotient or one past it in magnitude, and the exact remainder (one fma) tells which.  Larger products (coordinates
		// beyond +-2^21, saturated projections) take the 64-bit division.
		const bool horiz = (oc & 12) != 0;                           // cut by y = h / y = 0: the result is an x
		const int p0 = horiz ? x0 : y0, p1 = horiz ? x1 : y1, q0 = horiz ? y0 : x0, q1 = horiz ? y1 : x1;
		const int lim = (oc & 8) ? h : (oc & 4) ? 0 : (oc & 2) ? w : 0;
		const double fa = (double)p1 - (double)p0, fb = (double)lim - (double)q0, fd = (double)q1 - (double)q0;
		const double num = fa*fb;
		long long cut;
		if (__builtin_fabs(num) < 0x1p52 && __builtin_fabs(fa) < 0x1p31) {
			// |quotient| <= |p1 - p0| < 2^31 (the cut lies on the segment; every caller's coordinates come out of trunc_sat,
			// +-2^29, and anything wider takes the 64-bit branch like the host)
			int q = (int)(num/fd);
			const double r = __builtin_fma(-(double)q, fd, num);        // exact
			if (r != 0.0 && ((r < 0.0) != (num < 0.0))) q -= ((num < 0.0) != (fd < 0.0)) ? -1 : 1;
			cut = (long long)p0 + q;
		} else cut = (long long)p0 + (((long long)p1 - p0)*((long long)lim - q0))/((long long)q1 - q0);
		const long long x = horiz ? cut : (long long)lim, y = horiz ? (long long)lim : cut;
#else
		long long x = 0, y = 0;
		const long long X0 = x0, Y0 = y0, X1 = x1, Y1 = y1;
		if (oc & 8)      { x = X0 + ((X1 - X0)*(h - Y0))/(Y1 - Y0); y = h; }
		else if (oc & 4) { x = X0 + ((X1 - X0)*(0 - Y0))/(Y1 - Y0); y = 0; }
		else if (oc & 2) { y = Y0 + ((Y1 - Y0)*(w - X0))/(X1 - X0); x = w; }
		else if (oc & 1) { y = Y0 + ((Y1 - Y0)*(0 - X0))/(X1 - X0); x = 0; }
#endif
		if (oc == oc0) { x0 = (int)x; y0 = (int)y; oc0 = out_code(x0, y0, w, h); }
		else           { x1 = (int)x; y1 = (int)y; oc1 = out_code(x1, y1, w, h); }
	}
	return false;   // unreachable for finite input: each round fixes one out-code bit
}

// LineIterator state (lineiter.hpp:32-118).  `begin` applies initialize()+reset();
// when bw > 0 the walk is limited, through the closed form of the Bresenham
// state, to points whose major coordinate is inside the image: points outside are
// dropped by every caller (mask.pixel() is INVALID there), so the visible sequence
// is unchanged while a segment is never longer than max(w,h) steps.
struct LineWalk {
	int x, y, xend, error, ystep, deltax, deltay;
	bool steep;

	SRH_HD void begin(int x0, int y0, int x1, int y1, int bw, int bh) {
		steep = abs(y1 - y0) > abs(x1 - x0);
		int t;
		if (steep) { t = x0; x0 = y0; y0 = t; t = x1; x1 = y1; y1 = t; }
		if (x0 > x1) { t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
		deltax = x1 - x0;
		deltay = abs(y1 - y0);
		ystep = (y0 < y1 ? 1 : -1);
		error = deltax / 2;
		x = x0; y = y0; xend = x1;
		if (bw > 0) {
			const int hi = (steep ? bh : bw) - 1;
			if (xend > hi) xend = hi;
			if (x < 0) {
				const long long k = -(long long)x0;
				if (k > deltax) { x = 1; xend = 0; return; }
				const long long s = (k*deltay - error + deltax - 1) / deltax;
				error = (int)(error - k*deltay + s*deltax);
				y = (int)(y0 + ystep*s);
				x = 0;
			}
		}
	}
	SRH_HD bool has_next() const { return x <= xend; }
	SRH_HD void current(int &px, int &py) const { if (steep) { px = y; py = x; } else { px = x; py = y; } }
	SRH_HD void next() {
		++x;
		error -= deltay;
		if (error < 0) { y += ystep; error += deltax; }
	}
};

} // namespace srh
