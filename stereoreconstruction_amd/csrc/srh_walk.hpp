// srh_walk.hpp -- device-side epipolar-curve walker and the two weighted-NCC costs.
//
//   walk_curve<MVS>   TwoViewStereo::epipolarCurve (stereo/twoviewstereo.cpp:999-1054) /
//                     MultiViewStereo::epipolarCurve (stereo/multiviewstereo.cpp:754-810):
//                     calls vis(cx, cy) for every candidate pixel, in the reference's order.
//   tv_cost           TwoViewStereo::cost_ncc (twoviewstereo.cpp:909-977), lower is better
//   mvs_cost          free cost_ncc (multiviewstereo.cpp:113-189), higher is better
//   candidate_depth   camera-space z of the two-ray mid-point (twoviewstereo.cpp:287-300)
#pragma once

#include "srh_internal.hpp"
#include "srh_geom.hpp"

namespace srh {

// Plane3d::dist_ of label d's depth plane: dot(n, C + normal*depth_d), n = normalized(normal) -- the part of
// pointFromDepth (point_from_depth above) that depends on the label only; walk_curve takes it from a table
// (label_plane_table_kernel) instead of recomputing two divisions, a square root and three more divisions
// per pixel and label.  The operations are point_from_depth's own, so nothing changes in the results.
__device__ __forceinline__ double label_plane_dist(const srh_camera &refcam, const srh_params &P, bool mvs, int label) {
	const Vec3 normal = load3(refcam.pdir);
	const Vec3 n = normalized(normal);
	const Vec3 x0 = load3(refcam.C) + normal*depth_from_label(P, mvs, label);
	return dot(n, x0);
}

// tdist: label_plane_dist per label, or null (computed in place)
template <bool MVS, class Visitor>
__device__ __forceinline__ void walk_curve(const Ray &ray, const srh_camera &refcam, const ViewDev &oth,
                                           const srh_params &P, Visitor &vis, const double *__restrict__ tdist = nullptr)
{
	const Vec3 camC = load3(refcam.C);
	const Vec3 normal = load3(refcam.pdir);
	const int OW = oth.w, OH = oth.h;
	double x1 = __builtin_nan(""), y1 = __builtin_nan("");
	int jx1 = 0, jy1 = 0;
	int lastx = -2147483647, lasty = -2147483647;               // MVS std::unique state
	// label-independent parts of intersect(ray, plane) and of the other camera's refraction
	const Vec3 pn = normalized(normal);
	const double nd = dot(pn, ray.dir);
	const Vec3 oth_bn = normalized(load3(oth.cam.plane_normal));
	for (int d = 0; d < P.num_depth_levels; ++d) {
		Vec3 point = camC;
		if (tdist) {
			// intersect_plane(ray, pn, tdist[d], point) with n . dir taken out of the loop
			if (fabs(nd) < 1e-10) continue;
			const Vec3 x0 = tdist[d]*pn;
			const double t = dot(pn, x0 - ray.src) / nd;
			if (t < 1e-10) continue;
			point = ray.src + t*ray.dir;
		} else {
			const double depth = depth_from_label(P, MVS, d);
			if (!point_from_depth(ray, normal, depth, point)) continue;
		}
		if (!cam_project(oth.cam, point, &oth_bn)) continue;
		const double x2 = point.x*P.image_scale;
		const double y2 = point.y*P.image_scale;
		if (isnan_d(x1)) { x1 = x2; y1 = y2; jx1 = trunc_sat(x2); jy1 = trunc_sat(y2); continue; }
		const double dx = x2 - x1, dy = y2 - y1;
		if (!(dx*dx + dy*dy >= 1)) continue;
		int ix0 = jx1, iy0 = jy1, ix1 = trunc_sat(x2), iy1 = trunc_sat(y2);   // (the kept point's truncations travel with it)
		jx1 = ix1; jy1 = iy1;
		LineWalk lw;
		bool ok = true;
		if (MVS) {
			ok = clip_line(ix0, iy0, ix1, iy1, OW, OH);          // 6-arg LineIterator, multiviewstereo.cpp:783
			if (ok) lw.begin(ix0, iy0, ix1, iy1, 0, 0);
		} else {
			lw.begin(ix0, iy0, ix1, iy1, OW, OH);                // 4-arg LineIterator, twoviewstereo.cpp:1028
		}
		if (ok) {
			while (lw.has_next()) {
				int tx, ty;
				lw.current(tx, ty);
				if ((unsigned)tx < (unsigned)OW && (unsigned)ty < (unsigned)OH && oth.mask[(size_t)ty*OW + tx] == 1) {
					if (MVS) {
						// std::unique over consecutive kept points, multiviewstereo.cpp:801-807
						if (!(tx == lastx && ty == lasty)) { lastx = tx; lasty = ty; vis(tx, ty); }
					} else {
						vis(tx, ty);
					}
				}
				lw.next();
			}
		}
		x1 = x2; y1 = y2;
	}
}

// ---- pinhole fast walk -----------------------------------------------------------------
// For an undistorted, non-refractive pair the per-label plane intersection of
// pointFromDepth/intersect (twoviewstereo.cpp:987-995, util/ray.cpp:78-88) splits into a part
// that depends only on the label -- t_num[d] = n . (dist_d*n - src), the ray source being the
// camera centre for every pixel -- and one division by  n . dir  per pixel and label.  The
// operands and operations are those of point_from_depth(), so the results are bit-identical;
// only the redundant recomputation (normalising the plane normal, the label depth, ...) goes.
__device__ __forceinline__ Vec3 pinhole_ray_source(const srh_camera &cam) {
	return matvec(cam.Rinv, v3(0, 0, 0) - load3(cam.t));        // cam_unproject: out.src
}

__device__ __forceinline__ double pinhole_label_tnum(const srh_camera &refcam, const srh_params &P, bool mvs, int label) {
	const Vec3 normal = load3(refcam.pdir);
	const Vec3 n = normalized(normal);                          // Plane3d ctor
	const double depth = depth_from_label(P, mvs, label);
	const Vec3 x0 = load3(refcam.C) + normal*depth;             // p + normal*depth, p = camera centre
	const double d = dot(n, x0);                                // Plane3d::dist_
	const Vec3 x0p = d*n;                                       // Plane3d::x0()
	return dot(n, x0p - pinhole_ray_source(refcam));
}

// projection of label `d` of the ray into the other (pinhole) view, in scaled pixels
__device__ __forceinline__ bool pinhole_project_label(const Ray &ray, double nd, double tnum, const srh_camera &oth,
                                                      double scale, double &x2, double &y2)
{
	const double t = tnum / nd;
	if (t < 1e-10) return false;
	const Vec3 point = ray.src + t*ray.dir;
	const Vec3 pl = matvec(oth.R, point) + load3(oth.t);
	const Vec3 pk = matvec(oth.K, pl);
	const double z = pk.z;
	x2 = (pk.x/z)*scale;
	y2 = (pk.y/z)*scale;
	return true;
}

// ---- one denominator, many numerators ------------------------------------------------------------------
// The compiler expands a double division into v_div_scale x2, v_rcp_f64, four Newton fmas on the reciprocal of
// the (scaled) denominator, a multiply, a remainder fma, v_div_fmas and v_div_fixup.  Six of those eleven
// instructions depend on the denominator alone.  When both operands are far from the exponent limits,
// v_div_scale scales nothing, v_div_fmas is a plain fma and v_div_fixup returns its first operand (CDNA4 ISA,
// V_DIV_SCALE_F64 / V_DIV_FIXUP_F64), so hoisting the denominator half changes no bit of the quotient; any
// other operand takes the ordinary division.  The scan kernel divides 256 label numerators by one n.dir per
// pixel and two image coordinates by one z per label.
struct SharedDivisor { double b, r; bool ok; };
__device__ __forceinline__ SharedDivisor shared_divisor(double b) {
	SharedDivisor q;
	q.b = b;
	const double ab = fabs(b);
	q.ok = ab > 0x1p-300 && ab < 0x1p300;
	double r = __builtin_amdgcn_rcp(b);
	double e = __builtin_fma(-b, r, 1.0);
	r = __builtin_fma(r, e, r);
	e = __builtin_fma(-b, r, 1.0);
	q.r = __builtin_fma(r, e, r);
	return q;
}
__device__ __forceinline__ double div_by(double a, const SharedDivisor &q) {
	const double aa = fabs(a);
	if (q.ok && aa > 0x1p-300 && aa < 0x1p300) {
		const double m = a*q.r;
		const double rem = __builtin_fma(-q.b, m, a);
		return __builtin_fma(rem, q.r, m);
	}
	return a / q.b;
}

// pinhole_project_label with the pixel's n.dir prepared as a shared divisor: the same quotients, fewer instructions
__device__ __forceinline__ bool pinhole_project_label_sd(const Ray &ray, const SharedDivisor &nd, double tnum,
                                                         const srh_camera &oth, double scale, double &x2, double &y2)
{
	const double t = div_by(tnum, nd);
	if (t < 1e-10) return false;
	const Vec3 point = ray.src + t*ray.dir;
	const Vec3 pl = matvec(oth.R, point) + load3(oth.t);
	const Vec3 pk = matvec(oth.K, pl);
	const SharedDivisor z = shared_divisor(pk.z);
	x2 = div_by(pk.x, z)*scale;
	y2 = div_by(pk.y, z)*scale;
	return true;
}

// ---- certified label projections (pinhole other camera) -------------------------------------------
// The reference projects label d of a pixel's ray as  point = src + t*dir;  pl = R*point + tvec;  pk = K*pl;
// (x, y) = (pk.x/pk.z, pk.y/pk.z)*scale  -- about 60 FP64 instructions per label -- and then only DECIDES with the
// result: is the step from the last kept point at least one pixel long (dx^2 + dy^2 >= 1), and which pixel do the
// coordinates truncate to.  k(t) = A + t*B with A = K*(R*src + tvec), B = K*R*dir (per pixel, once) gives the same point
// in 3 fused multiply-adds and a reciprocal.  Both evaluations are within gamma_10*Km of the real-number pk, Km_i =
// sum_j |K_ij| (sum_k |R_jk| (|src_k| + |t| |dir_k|) + |tvec_j|), hence within 2*gamma_10*Km of each other; divided by
// pk.z that is a bound (ex, ey) on the coordinates.  A decision that (ex, ey) cannot change -- a coordinate further
// than its bound from every integer, a squared step further from 1 than its propagated bound -- is the reference's; any
// other label is projected once more by the reference's own operations (exact_label_point), so the candidate lists
// are the reference's lists.
struct FastProj { Vec3 A, B; double ek, ekz; };              // ek: the larger of the x and y bounds (one register pair less)
__device__ __forceinline__ FastProj fast_proj_setup(const Ray &ray, const srh_camera &oth, double tmax) {
	FastProj f;
	f.A = matvec(oth.K, matvec(oth.R, ray.src) + load3(oth.t));
	f.B = matvec(oth.K, matvec(oth.R, ray.dir));
	const Vec3 pm = v3(fabs(ray.src.x) + tmax*fabs(ray.dir.x), fabs(ray.src.y) + tmax*fabs(ray.dir.y), fabs(ray.src.z) + tmax*fabs(ray.dir.z));
	const double *R = oth.R, *K = oth.K;
	const Vec3 lm = v3((fabs(R[0])*pm.x + fabs(R[1])*pm.y) + fabs(R[2])*pm.z + fabs(oth.t[0]),
	                   (fabs(R[3])*pm.x + fabs(R[4])*pm.y) + fabs(R[5])*pm.z + fabs(oth.t[1]),
	                   (fabs(R[6])*pm.x + fabs(R[7])*pm.y) + fabs(R[8])*pm.z + fabs(oth.t[2]));
	const double g = 2.02*(10*0x1p-53)/(1.0 - 10*0x1p-53);      // 2*gamma_10, + 1 % for the roundings of these sums
	const double ekx = g*((fabs(K[0])*lm.x + fabs(K[1])*lm.y) + fabs(K[2])*lm.z);
	const double eky = g*((fabs(K[3])*lm.x + fabs(K[4])*lm.y) + fabs(K[5])*lm.z);
	f.ek = ekx > eky ? ekx : eky;
	f.ekz = g*((fabs(K[6])*lm.x + fabs(K[7])*lm.y) + fabs(K[8])*lm.z);
	return f;
}
// the label's image point (scaled) with its bound; a degenerate pk.z gives NaN / huge bounds, which no decision accepts
// e: bound on BOTH coordinates
__device__ __forceinline__ void fast_project(const FastProj &f, double t, double scale, double &x2, double &y2, double &e) {
	const double kx = __builtin_fma(t, f.B.x, f.A.x), ky = __builtin_fma(t, f.B.y, f.A.y), kz = __builtin_fma(t, f.B.z, f.A.z);
	double r = __builtin_amdgcn_rcp(kz);
	r = __builtin_fma(r, __builtin_fma(-kz, r, 1.0), r);
	r = __builtin_fma(r, __builtin_fma(-kz, r, 1.0), r);           // 1/kz to 2 ulp
	const double rs = r*scale;
	x2 = kx*rs; y2 = ky*rs;
	// |pk.x/pk.z - kx/kz| <= (ekx + |x| ekz)/(|kz| - ekz): with ekz <= |kz|/1024 (else the bound is made useless) the
	// denominator is >= 0.999 |kz|; + 8u|x| for the quotient, reciprocal and product roundings of either evaluation
	const double ar = fabs(rs)*(f.ekz*fabs(r) <= 0x1p-10 ? 1.002 : __builtin_inf());
	const double am = fmax(fabs(kx), fabs(ky))*fabs(r);            // the larger |coordinate| / scale
	e = __builtin_fma(am*scale, 0x1p-49, (f.ek + am*f.ekz)*ar);
}
// the reference's own operations for one label (pinhole_project_label_sd without the t test, which the caller made)
// (returned by value: reference parameters would pin the caller's coordinates to scratch memory)
__device__ __noinline__ double2 exact_label_point(const srh_camera &refcam, const srh_camera &oth, int px, int py, double scale, double t)
{
	const Ray ray = cam_unproject(refcam, (px + 0.5) / scale, (py + 0.5) / scale);
	const Vec3 point = ray.src + t*ray.dir;
	const Vec3 pl = matvec(oth.R, point) + load3(oth.t);
	const Vec3 pk = matvec(oth.K, pl);
	const SharedDivisor z = shared_divisor(pk.z);
	double2 r;
	r.x = div_by(pk.x, z)*scale;
	r.y = div_by(pk.y, z)*scale;
	return r;
}
// is truncation of v certain under the bound e?  (every integer counts as a boundary; far outside any image: no)
__device__ __forceinline__ bool trunc_certain(double v, double e) { return fabs(v) < 0x1p28 && fabs(v - __builtin_rint(v)) > e; }

template <class Visitor>
__device__ __forceinline__ void walk_curve_pinhole(const Ray &ray, const srh_camera &refcam, const ViewDev &oth,
                                                   const srh_params &P, const double *__restrict__ tnum, Visitor &vis)
{
	const Vec3 n = normalized(load3(refcam.pdir));
	const double nd = dot(n, ray.dir);
	if (fabs(nd) < 1e-10) return;                               // intersect() fails for every label
	const int OW = oth.w, OH = oth.h;
	double x1 = __builtin_nan(""), y1 = __builtin_nan("");
	for (int d = 0; d < P.num_depth_levels; ++d) {
		double x2, y2;
		if (!pinhole_project_label(ray, nd, tnum[d], oth.cam, P.image_scale, x2, y2)) continue;
		if (isnan_d(x1)) { x1 = x2; y1 = y2; continue; }
		const double dx = x2 - x1, dy = y2 - y1;
		if (!(dx*dx + dy*dy >= 1)) continue;
		LineWalk lw;
		lw.begin(trunc_sat(x1), trunc_sat(y1), trunc_sat(x2), trunc_sat(y2), OW, OH);
		while (lw.has_next()) {
			int tx, ty;
			lw.current(tx, ty);
			if (tx >= 0 && ty >= 0 && tx < OW && ty < OH && oth.mask[(size_t)ty*OW + tx] == 1) vis(tx, ty);
			lw.next();
		}
		x1 = x2; y1 = y2;
	}
}

// Column range [lo, hi] that contains every candidate of the pixel when the curve stays on
// one image row: the truncated projections of the first and last label (Bresenham segments
// between successive projections cannot leave that interval while the projection is monotone
// in depth), clamped to the image.  The scan kernel verifies the claim for every candidate.
__device__ __forceinline__ void pinhole_column_range(const Ray &ray, const srh_camera &refcam, const ViewDev &oth,
                                                     const srh_params &P, const double *__restrict__ tnum,
                                                     int max_span, int &lo, int &hi)
{
	lo = 0; hi = -1;
	const Vec3 n = normalized(load3(refcam.pdir));
	const double nd = dot(n, ray.dir);
	if (fabs(nd) < 1e-10) return;
	double xa, ya, xb, yb;
	if (!pinhole_project_label(ray, nd, tnum[0], oth.cam, P.image_scale, xa, ya)) return;
	if (!pinhole_project_label(ray, nd, tnum[P.num_depth_levels - 1], oth.cam, P.image_scale, xb, yb)) return;
	const int ia = trunc_sat(xa), ib = trunc_sat(xb);
	lo = (ia < ib ? ia : ib);
	hi = (ia < ib ? ib : ia);
	if (lo < 0) lo = 0;
	if (hi > oth.w - 1) hi = oth.w - 1;
	if (hi - lo + 1 > max_span) hi = lo + max_span - 1;
}

// The dense kernel evaluates the columns [lo & ~1, cover_hi] of a pixel in blocks of `ncb`
// shared by `lanes` lanes.  When the last block would cost a whole extra round of the lanes for
// at most two columns (typically the never-visited column of the last label plus the alignment
// pad), it is left out: the scan evaluates such a column on demand with the general cost.
// (pad: blocks start on even columns -- the kernels that keep ONE copy of the other view's rows in LDS and read it 16
// bytes at a time; the 8-wave strip kernel keeps a second copy shifted by one column and starts its blocks at lo)
__device__ __forceinline__ int dense_cover_hi(int lo, int hi, int ncb, int lanes, bool pad = true) {
	const int lo_e = pad ? (lo & ~1) : lo;
	const int nblocks = (hi - lo_e + ncb)/ncb;
	const int last_cols = (hi - lo_e + 1) - (nblocks - 1)*ncb;
	if (nblocks > lanes && nblocks % lanes == 1 && last_cols <= 2) return lo_e + (nblocks - 1)*ncb - 1;
	return hi;
}

// gray value of a TwoView tap, NaN when the tap is skipped on that side
__device__ __forceinline__ double tv_tap(const ViewDev &V, int x, int y) {
	if (x < 0 || y < 0 || x >= V.w || y >= V.h) return __builtin_nan("");
	return V.gray_tv[(size_t)y*V.w + x];
}

// tap (row, col) of the window at wq[row*wrow + col*wstride]; wrow = 0 means the tile-major layout (row*WS + col)*wstride
__device__ __forceinline__ double tv_cost(const ViewDev &L, const ViewDev &Rv, const double *__restrict__ wq,
                                          size_t wstride, const srh_params &P, int x1, int y1, int x2, int y2,
                                          size_t wrow = 0)
{
	const int R = P.window_radius, WS = 2*R + 1;
	if (wrow == 0) wrow = (size_t)WS*wstride;
	double meanL = 0, meanR = 0, totalWeight = 0.0;
	for (int row = -R; row <= R; ++row) {
		for (int col = -R; col <= R; ++col) {
			const double gl = tv_tap(L, x1 + col, y1 + row);
			const double gr = tv_tap(Rv, x2 + col, y2 + row);
			const double weight = wq[(size_t)(row + R)*wrow + (size_t)(col + R)*wstride];
			if (gl == gl && gr == gr && weight > P.weight_cutoff) {
				meanL += weight*gl;
				meanR += weight*gr;
				totalWeight += weight;
			}
		}
	}
	if (totalWeight < 1e-10) return P.bad_ret;
	meanL /= totalWeight;
	meanR /= totalWeight;
	double sum1 = 0, sum2 = 0, sum3 = 0;
	for (int row = -R; row <= R; ++row) {
		for (int col = -R; col <= R; ++col) {
			const double gl = tv_tap(L, x1 + col, y1 + row);
			const double gr = tv_tap(Rv, x2 + col, y2 + row);
			const double weight = wq[(size_t)(row + R)*wrow + (size_t)(col + R)*wstride];
			if (gl == gl && gr == gr && weight > P.weight_cutoff) {
				const double pgl = weight*gl;
				const double pgr = weight*gr;
				sum1 += (pgl - meanL)*(pgr - meanR);
				sum2 += (pgl - meanL)*(pgl - meanL);
				sum3 += (pgr - meanR)*(pgr - meanR);
			}
		}
	}
	const double v = 255*(1.0 - fabs(sum1) / sqrt(sum2 * sum3));
	return (v < P.max_color_diff) ? v : P.max_color_diff;       // std::min(MAX_COLOR_DIFF, v): NaN -> 120
}

// cost_ncc in the reference's arithmetic for ANY validity pattern (a skipped tap adds +0.0: the same sums, same order as
// tv_cost, twoviewstereo.cpp:909-977), from a window in either band layout and the NaN-bordered planes (no bound tests,
// loops unrolled by window row so that a row's 33 loads travel together):
// tap (row, col) of the window at wq[row*wrow + col*wcol]; lp / rp = top-left tap of the two windows in the padded planes
// (strides SP / SPR)
template <int R>
__device__ __forceinline__ double window_exact_cost(const double *__restrict__ wq, int wrow, int wcol, const double *__restrict__ lp,
                                                    const double *__restrict__ rp, int SP, int SPR, const srh_params &P)
{
	constexpr int WS = 2*R + 1;
	double meanL = 0, meanR = 0, totalWeight = 0.0;
#pragma unroll 1
	for (int row = 0; row < WS; ++row) {
		double gl[WS], gr[WS], wt[WS];
#pragma unroll
		for (int col = 0; col < WS; ++col) { gl[col] = lp[(size_t)row*SP + col]; gr[col] = rp[(size_t)row*SPR + col]; wt[col] = wq[row*wrow + col*wcol]; }
#pragma unroll
		for (int col = 0; col < WS; ++col) {
			const bool ok = gl[col] == gl[col] && gr[col] == gr[col] && wt[col] > P.weight_cutoff;
			const double pl = wt[col]*gl[col], prr = wt[col]*gr[col];
			meanL += ok ? pl : 0.0;
			meanR += ok ? prr : 0.0;
			totalWeight += ok ? wt[col] : 0.0;
		}
	}
	if (totalWeight < 1e-10) return P.bad_ret;
	meanL /= totalWeight;
	meanR /= totalWeight;
	double sum1 = 0, sum2 = 0, sum3 = 0;
#pragma unroll 1
	for (int row = 0; row < WS; ++row) {
		double gl[WS], gr[WS], wt[WS];
#pragma unroll
		for (int col = 0; col < WS; ++col) { gl[col] = lp[(size_t)row*SP + col]; gr[col] = rp[(size_t)row*SPR + col]; wt[col] = wq[row*wrow + col*wcol]; }
#pragma unroll
		for (int col = 0; col < WS; ++col) {
			const bool ok = gl[col] == gl[col] && gr[col] == gr[col] && wt[col] > P.weight_cutoff;
			const double a = wt[col]*gl[col] - meanL, b = wt[col]*gr[col] - meanR;
			const double ab = a*b, aa = a*a, bb = b*b;
			sum1 += ok ? ab : 0.0;
			sum2 += ok ? aa : 0.0;
			sum3 += ok ? bb : 0.0;
		}
	}
	const double v = 255*(1.0 - fabs(sum1) / sqrt(sum2 * sum3));
	return (v < P.max_color_diff) ? v : P.max_color_diff;
}

// where a pixel's window lies in the band buffer: wimg != 0 the strip path's LDS-image layout, else tile-major
struct WindowAt { const double *wq; int wrow, wcol; };
template <int R>
__device__ __forceinline__ WindowAt window_at(const double *wbuf, int wimg, int W, int trow, int x) {
	constexpr int WS = 2*R + 1;
	WindowAt a;
	if (wimg) { a.wq = wbuf + wimg_offset(W, R, trow, x); a.wrow = wimg_row_stride(R); a.wcol = 1; }
	else      { a.wq = wbuf + wbuf_offset(W, WS*WS, trow, x); a.wrow = WS*SRH_WTILE; a.wcol = SRH_WTILE; }
	return a;
}

__device__ __forceinline__ double mvs_tap(const ViewDev &V, int x, int y) {
	if (x < 0 || y < 0 || x >= V.w || y >= V.h) return __builtin_nan("");
	return V.gray[(size_t)y*V.w + x];
}

__device__ __forceinline__ double mvs_cost(const ViewDev &A, const ViewDev &B, const double *__restrict__ wq,
                                           size_t wstride, const srh_params &P, int x1, int y1, int x2, int y2)
{
	const int R = P.window_radius, WS = 2*R + 1;
	double meanL = 0, meanR = 0, totalWeight = 0.0;
	for (int row = -R; row <= R; ++row) {
		for (int col = -R; col <= R; ++col) {
			const double gl = mvs_tap(A, x1 + col, y1 + row);
			const double gr = mvs_tap(B, x2 + col, y2 + row);
			const double weight = wq[(size_t)((row + R)*WS + (col + R))*wstride];
			if (gl == gl && gr == gr && weight > P.weight_cutoff) {
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
			const double gl = mvs_tap(A, x1 + col, y1 + row);
			const double gr = mvs_tap(B, x2 + col, y2 + row);
			const double weight = wq[(size_t)((row + R)*WS + (col + R))*wstride];
			if (gl == gl && gr == gr && weight > P.weight_cutoff) {
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

__device__ __forceinline__ double candidate_depth(const srh_camera &refcam, const srh_camera &othcam,
                                                  const srh_params &P, const Ray &ray, int cx, int cy)
{
	const Ray ray2 = cam_unproject(othcam, (cx + 0.5) / P.image_scale, (cy + 0.5) / P.image_scale);
	Vec3 p1, p2;
	closest_points(ray, ray2, p1, p2);
	p1 = p1 + p2;
	p1 = p1*0.5;
	return cam_local_z(refcam, p1);
}

// block-wide add of per-thread counts into one global counter
__device__ __forceinline__ void block_count_add(unsigned long long *dst, unsigned long long v) {
	__shared__ unsigned long long acc;
	if (threadIdx.x == 0) acc = 0;
	__syncthreads();
	if (v) atomicAdd(&acc, v);
	__syncthreads();
	if (threadIdx.x == 0 && acc) atomicAdd(dst, acc);
}

} // namespace srh
