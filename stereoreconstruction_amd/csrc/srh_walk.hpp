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

template <bool MVS, class Visitor>
__device__ __forceinline__ void walk_curve(const Ray &ray, const srh_camera &refcam, const ViewDev &oth,
                                           const srh_params &P, Visitor &vis)
{
	const Vec3 camC = load3(refcam.C);
	const Vec3 normal = load3(refcam.pdir);
	const int OW = oth.w, OH = oth.h;
	double x1 = __builtin_nan(""), y1 = __builtin_nan("");
	int lastx = -2147483647, lasty = -2147483647;               // MVS std::unique state
	for (int d = 0; d < P.num_depth_levels; ++d) {
		Vec3 point = camC;
		const double depth = depth_from_label(P, MVS, d);
		if (!point_from_depth(ray, normal, depth, point)) continue;
		if (!cam_project(oth.cam, point)) continue;
		const double x2 = point.x*P.image_scale;
		const double y2 = point.y*P.image_scale;
		if (isnan_d(x1)) { x1 = x2; y1 = y2; continue; }
		const double dx = x2 - x1, dy = y2 - y1;
		if (!(dx*dx + dy*dy >= 1)) continue;
		int ix0 = trunc_sat(x1), iy0 = trunc_sat(y1), ix1 = trunc_sat(x2), iy1 = trunc_sat(y2);
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
				if (tx >= 0 && ty >= 0 && tx < OW && ty < OH && oth.mask[(size_t)ty*OW + tx] == 1) {
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

// gray value of a TwoView tap, NaN when the tap is skipped on that side
__device__ __forceinline__ double tv_tap(const ViewDev &V, int x, int y) {
	if (x < 0 || y < 0 || x >= V.w || y >= V.h) return __builtin_nan("");
	return V.gray_tv[(size_t)y*V.w + x];
}

__device__ __forceinline__ double tv_cost(const ViewDev &L, const ViewDev &Rv, const double *__restrict__ wq,
                                          size_t wstride, const srh_params &P, int x1, int y1, int x2, int y2)
{
	const int R = P.window_radius, WS = 2*R + 1;
	double meanL = 0, meanR = 0, totalWeight = 0.0;
	for (int row = -R; row <= R; ++row) {
		for (int col = -R; col <= R; ++col) {
			const double gl = tv_tap(L, x1 + col, y1 + row);
			const double gr = tv_tap(Rv, x2 + col, y2 + row);
			const double weight = wq[(size_t)((row + R)*WS + (col + R))*wstride];
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
	const double v = 255*(1.0 - fabs(sum1) / sqrt(sum2 * sum3));
	return (v < P.max_color_diff) ? v : P.max_color_diff;       // std::min(MAX_COLOR_DIFF, v): NaN -> 120
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
