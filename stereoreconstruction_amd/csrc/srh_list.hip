// srh_list.hip -- TwoView kernels for ARBITRARY epipolar geometry (verged, distorted,
// refractive rigs: BASELINE configs C1 and C5) organised like the dense row-aligned path:
// the support windows of a 32-pixel tile live in LDS and 8 lanes share each pixel's work.
// Because the candidates of a pixel are now an arbitrary pixel chain instead of a row range,
// the curve is rasterised once into a candidate list and the other view's window values are
// gathered from its gray_tv plane (L1/L2 resident) instead of being staged as an LDS tile.
//
//   twoview_count_kernel      curve walk, candidates per pixel (sizes the lists)            (8(a) #6,#7)
//   full_window_kernel        per pixel of a view: is the whole (2R+1)^2 window usable?
//   twoview_list_kernel       curve walk again, writes the candidate list (cx | cy<<16)
//   twoview_list_cost_kernel  weighted NCC of every list entry                               (#8)
//   twoview_list_scan_kernel  running-min WTA over the list in order + depth of the winner   (#9,#10)
//
// Costs are bit-identical to srh_walk.hpp::tv_cost (same operations, same order; skipped taps
// add +0.0).  A list entry equal to its predecessor (segment joints) is neither evaluated nor
// scanned: an equal cost can never pass `cost + 1e-10 < minCost`.
#include "srh_internal.hpp"
#include "srh_geom.hpp"
#include "srh_walk.hpp"

namespace srh {

// ------------------------------------------------------------------ count / list
struct CountVisitor {
	unsigned n;
	__device__ __forceinline__ void operator()(int, int) { ++n; }
};

__global__ void twoview_count_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P,
                                     int y0, int nrows, int32_t *__restrict__ count, Counters *__restrict__ cnt,
                                     int *__restrict__ max_count, const double *__restrict__ tdist)
{
	const ViewDev &L = views[ref];
	const int W = L.w;
	const size_t q = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
	unsigned n_eval = 0, n_pix = 0;
	if (q < (size_t)nrows*W) {
		const int x = (int)(q % W), y = y0 + (int)(q / W);
		if (L.mask[(size_t)y*W + x] == 1) {
			n_pix = 1;
			const Ray ray = cam_unproject(L.cam, (x + 0.5) / P.image_scale, (y + 0.5) / P.image_scale);
			CountVisitor vis = { 0 };
			walk_curve<false>(ray, L.cam, views[oth], P, vis, tdist);
			n_eval = vis.n;
		}
		count[q] = (int32_t)n_eval;
	}
	__shared__ int s_max;
	if (threadIdx.x == 0) s_max = 0;
	__syncthreads();
	if (n_eval) atomicMax(&s_max, (int)n_eval);
	__syncthreads();
	if (threadIdx.x == 0 && s_max > 0) atomicMax(max_count, s_max);
	block_count_add(&cnt->n_eval, n_eval);
	block_count_add(&cnt->n_pixels, n_pix);
}

void launch_twoview_count(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                          int y0, int nrows, int32_t *count, Counters *cnt, int *max_count, const double *tdist)
{
	const size_t n = (size_t)nrows*width;
	hipLaunchKernelGGL(twoview_count_kernel, dim3((unsigned)((n + 127)/128)), dim3(128), 0, st,
	                   views, ref, oth, P, y0, nrows, count, cnt, max_count, tdist);
}

struct ListVisitor {
	uint32_t *out;
	int cap, n;
	__device__ __forceinline__ void operator()(int cx, int cy) {
		if (n < cap) out[n] = (uint32_t)cx | ((uint32_t)cy << 16);
		++n;
	}
};

// Writes the candidate list of every pixel (up to cmax entries), its full length count[q], the
// longest list of the launch and the work counters: when the caller's cmax (a hint kept from an
// earlier run on the same rig) turns out too small, it reruns with the reported maximum.
__global__ void twoview_list_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P,
                                    int y0, int nrows, uint32_t *__restrict__ cand, int cmax,
                                    int32_t *__restrict__ count, Counters *__restrict__ cnt, int *__restrict__ max_count,
                                    const double *__restrict__ tdist)
{
	const ViewDev &L = views[ref];
	const int W = L.w;
	const size_t q = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
	unsigned n_eval = 0, n_pix = 0;
	if (q < (size_t)nrows*W) {
		const int x = (int)(q % W), y = y0 + (int)(q / W);
		if (L.mask[(size_t)y*W + x] == 1) {
			n_pix = 1;
			const Ray ray = cam_unproject(L.cam, (x + 0.5) / P.image_scale, (y + 0.5) / P.image_scale);
			ListVisitor vis = { cand + q*(size_t)cmax, cmax, 0 };
			walk_curve<false>(ray, L.cam, views[oth], P, vis, tdist);
			n_eval = (unsigned)vis.n;
		}
		count[q] = (int32_t)n_eval;
	}
	__shared__ int s_max;
	if (threadIdx.x == 0) s_max = 0;
	__syncthreads();
	if (n_eval) atomicMax(&s_max, (int)n_eval);
	__syncthreads();
	if (threadIdx.x == 0 && s_max > 0) atomicMax(max_count, s_max);
	block_count_add(&cnt->n_eval, n_eval);
	block_count_add(&cnt->n_pixels, n_pix);
}

void launch_twoview_list(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                         int y0, int nrows, uint32_t *cand, int cmax, int32_t *count, Counters *cnt, int *max_count,
                         const double *tdist)
{
	const size_t n = (size_t)nrows*width;
	hipLaunchKernelGGL(twoview_list_kernel, dim3((unsigned)((n + 127)/128)), dim3(128), 0, st,
	                   views, ref, oth, P, y0, nrows, cand, cmax, count, cnt, max_count, tdist);
}

// ------------------------------------------------------------------ fully usable windows of a view
// stat[0] += pixels with a usable centre tap, stat[1] += those whose whole window is usable (zeroed by the caller): how much of
// a view's candidates the fast forms cover -- the row-run cost kernel picks its treatment of the others by it
__global__ void full_window_kernel(const double *__restrict__ gray_tv, int W, int H, int R, uint8_t *__restrict__ full,
                                   uint32_t *__restrict__ stat)
{
	const size_t n = (size_t)W*H;
	unsigned n_centre = 0, n_full = 0;
	for (size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) {
		const int x = (int)(i % (size_t)W), y = (int)(i / (size_t)W);
		bool ok = x - R >= 0 && y - R >= 0 && x + R < W && y + R < H;
		for (int row = -R; ok && row <= R; ++row)
			for (int col = -R; col <= R; ++col) {
				const double v = gray_tv[(size_t)(y + row)*W + (x + col)];
				ok = ok && (v == v);
			}
		full[i] = ok ? 1 : 0;
		const double c0 = gray_tv[i];
		n_centre += (c0 == c0) ? 1u : 0u; n_full += ok ? 1u : 0u;
	}
	__shared__ unsigned s_acc[2];
	if (threadIdx.x == 0) { s_acc[0] = 0; s_acc[1] = 0; }
	__syncthreads();
	if (n_centre) atomicAdd(&s_acc[0], n_centre);
	if (n_full) atomicAdd(&s_acc[1], n_full);
	__syncthreads();
	if (threadIdx.x == 0 && stat) { atomicAdd(&stat[0], s_acc[0]); atomicAdd(&stat[1], s_acc[1]); }
}

void launch_full_window(hipStream_t st, const double *gray_tv, int w, int h, int R, uint8_t *full, uint32_t *stat) {
	size_t n = (size_t)w*h;
	size_t b = (n + 255)/256; if (b > 4096) b = 4096; if (b < 1) b = 1;
	hipLaunchKernelGGL(full_window_kernel, dim3((unsigned)b), dim3(256), 0, st, gray_tv, w, h, R, full, stat);
}

// ------------------------------------------------------------------ list cost
#define LC_TP 32
#define LC_G 8
#define LC_THREADS (LC_TP*LC_G)

template <int R>
struct ListSmem {
	static constexpr int WS = 2*R + 1;
	static constexpr int T = WS*WS;
	static constexpr int WP = (WS + 1) & ~1;
	static constexpr int WPIX = WS*WP;
	static constexpr int LW = LC_TP + 2*R;
	double w[LC_TP][WPIX];                                  // w[pixel][row*WP + col]
	double lt[WS][LW];
	double meanL[LC_TP], totalW[LC_TP], sum2[LC_TP];
	int lall[LC_TP];
	int count[LC_TP];
};

template <int R>
__global__ __launch_bounds__(LC_THREADS, 2)
void twoview_list_cost_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P,
                              int y0, int nrows, const double *__restrict__ wbuf,
                              const uint8_t *__restrict__ full_oth,
                              const int32_t *__restrict__ count, const uint32_t *__restrict__ cand,
                              double *__restrict__ cost, int cmax, Counters *__restrict__ cnt)
{
	constexpr int WS = 2*R + 1;
	constexpr int T = WS*WS;
	typedef ListSmem<R> Smem;
	constexpr int WP = Smem::WP;
	extern __shared__ __align__(16) unsigned char smem_raw[];
	Smem &S = *reinterpret_cast<Smem *>(smem_raw);

	const ViewDev &L = views[ref];
	const ViewDev &Rv = views[oth];
	const int W = L.w, H = L.h, OW = Rv.w, OH = Rv.h;
	const int tiles_per_row = (W + LC_TP - 1)/LC_TP;
	const int trow = blockIdx.x / tiles_per_row;
	const int x0 = (blockIdx.x % tiles_per_row)*LC_TP;
	const int y = y0 + trow;
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int i = (tid >> 6)*8 + (lane & 7);   // pixel within the tile (pixel-fastest inside a wave)
	const int g = lane >> 3;                   // lane within the pixel
	const int x = x0 + i;
	const size_t qbase = (size_t)trow*W + x0;
	const double nan = __builtin_nan("");

	// ---- stage the windows and the reference rows (all loads first, then the LDS stores)
	{
		static_assert(LC_TP == SRH_WTILE, "tile = window-buffer tile");
		constexpr int NBW = (T*LC_TP + LC_THREADS - 1)/LC_THREADS;
		constexpr int NBL = (WS*Smem::LW + LC_THREADS - 1)/LC_THREADS;
		const double *wtile = wbuf + wbuf_offset(W, T, trow, x0);
		double tw_[NBW], tl_[NBL];
#pragma unroll
		for (int k = 0; k < NBW; ++k) {
			const int idx = tid + k*LC_THREADS;
			tw_[k] = (idx < T*LC_TP && x0 + (idx % LC_TP) < W) ? wtile[idx] : 0.0;
		}
#pragma unroll
		for (int k = 0; k < NBL; ++k) {
			const int idx = tid + k*LC_THREADS;
			const int ty = idx / Smem::LW, tx = idx % Smem::LW;
			const int gx = x0 - R + tx, gy = y - R + ty;
			tl_[k] = (idx < WS*Smem::LW && gx >= 0 && gy >= 0 && gx < W && gy < H) ? L.gray_tv[(size_t)gy*W + gx] : nan;
		}
#pragma unroll
		for (int k = 0; k < NBW; ++k) {
			const int idx = tid + k*LC_THREADS;
			const int t = idx / LC_TP, pi = idx % LC_TP;
			if (idx < T*LC_TP) S.w[pi][(t / WS)*WP + (t % WS)] = tw_[k];
		}
		if (WP != WS)
			for (int idx = tid; idx < WS*LC_TP; idx += LC_THREADS) S.w[idx % LC_TP][(idx / LC_TP)*WP + WS] = 0.0;
#pragma unroll
		for (int k = 0; k < NBL; ++k) {
			const int idx = tid + k*LC_THREADS;
			if (idx < WS*Smem::LW) S.lt[idx / Smem::LW][idx % Smem::LW] = tl_[k];
		}
		if (tid < LC_TP) S.count[tid] = (x0 + tid < W) ? count[qbase + tid] : 0;
	}
	__syncthreads();

	// ---- per-pixel constants of the all-taps-usable form (one lane per pixel)
	if (g == 0) {
		bool all = (x < W) && S.count[i] > 0;
		double mL = 0, tw = 0;
#pragma unroll 1
		for (int row = 0; row < WS; ++row)
#pragma unroll
			for (int col = 0; col < WS; ++col) {
				const double gl = S.lt[row][i + col];
				const double wt = S.w[i][row*WP + col];
				if (!(gl == gl && wt > P.weight_cutoff)) all = false;
				mL += wt*gl;
				tw += wt;
			}
		double s2 = 0;
		if (all && !(tw < 1e-10)) {
			mL /= tw;
#pragma unroll 1
			for (int row = 0; row < WS; ++row)
#pragma unroll
				for (int col = 0; col < WS; ++col) {
					const double a = S.w[i][row*WP + col]*S.lt[row][i + col] - mL;
					s2 += a*a;
				}
		} else all = false;
		S.meanL[i] = mL; S.totalW[i] = tw; S.sum2[i] = s2; S.lall[i] = all ? 1 : 0;
	}
	__syncthreads();

	unsigned n_dev = 0;
	const Smem &CS = S;
	if (x < W) {
		const int n = CS.count[i] < cmax ? CS.count[i] : cmax;
		const uint32_t *clist = cand + (qbase + i)*(size_t)cmax;
		double *crow = cost + (qbase + i)*(size_t)cmax;
		const bool lall = CS.lall[i] != 0;
		const double mL = CS.meanL[i], tw = CS.totalW[i], s2 = CS.sum2[i];
		for (int k = g; k < n; k += LC_G) {
			const uint32_t e = clist[k];
			if (k > 0 && clist[k - 1] == e) continue;           // joint duplicate: never evaluated, never scanned
			const int cx = (int)(e & 0xffffu), cy = (int)(e >> 16);
			++n_dev;
			double result;
			if (lall && full_oth[(size_t)cy*OW + cx] != 0) {
				// fast form: every tap usable on both sides (twoviewstereo.cpp:917-976 with constant meanL,
				// totalWeight, sum2 and a_t): p = w*gr; meanR += p;  b = p - meanR; sum1 += a*b; sum3 += b*b
				const double *rbase = Rv.gray_tv + (size_t)(cy - R)*OW + (cx - R);
				// the gathers of window row r+1 are issued before the arithmetic of row r
				double gn[WS];
#pragma unroll
				for (int col = 0; col < WS; ++col) gn[col] = rbase[col];
				double mR = 0;
#pragma unroll 1
				for (int row = 0; row < WS; ++row) {
					double gr[WS], wv[WS];
#pragma unroll
					for (int col = 0; col < WS; ++col) gr[col] = gn[col];
					const double *rn = rbase + (size_t)(row + 1 < WS ? row + 1 : 0)*OW;     // last: row 0 again, for pass 2
#pragma unroll
					for (int col = 0; col < WS; ++col) gn[col] = rn[col];
#pragma unroll
					for (int col = 0; col < WS; ++col) wv[col] = CS.w[i][row*WP + col];
					__builtin_amdgcn_sched_barrier(0);
#pragma unroll
					for (int col = 0; col < WS; ++col) mR += wv[col]*gr[col];
				}
				mR /= tw;
				double s1 = 0, s3 = 0;
#pragma unroll 1
				for (int row = 0; row < WS; ++row) {
					double gr[WS], wv[WS], av[WS];
#pragma unroll
					for (int col = 0; col < WS; ++col) gr[col] = gn[col];
					const double *rn = rbase + (size_t)(row + 1 < WS ? row + 1 : 0)*OW;
#pragma unroll
					for (int col = 0; col < WS; ++col) gn[col] = rn[col];
#pragma unroll
					for (int col = 0; col < WS; ++col) { wv[col] = CS.w[i][row*WP + col]; av[col] = CS.lt[row][i + col]; }
					__builtin_amdgcn_sched_barrier(0);
#pragma unroll
					for (int col = 0; col < WS; ++col) {
						const double a = wv[col]*av[col] - mL;
						const double b = wv[col]*gr[col] - mR;
						s1 += a*b;
						s3 += b*b;
					}
				}
				const double v = 255*(1.0 - fabs(s1) / sqrt(s2 * s3));
				result = (v < P.max_color_diff) ? v : P.max_color_diff;
			} else {
				// any validity pattern; a skipped tap adds +0.0
				double meanL = 0, meanR = 0, totalWeight = 0.0;
#pragma unroll 1
				for (int row = 0; row < WS; ++row) {
					double gr[WS];
#pragma unroll
					for (int col = 0; col < WS; ++col) {
						const int gx = cx - R + col, gy = cy - R + row;
						gr[col] = (gx >= 0 && gy >= 0 && gx < OW && gy < OH) ? Rv.gray_tv[(size_t)gy*OW + gx] : nan;
					}
#pragma unroll
					for (int col = 0; col < WS; ++col) {
						const double gl = CS.lt[row][i + col], wt = CS.w[i][row*WP + col];
						const bool ok = gl == gl && gr[col] == gr[col] && wt > P.weight_cutoff;
						const double pl = wt*gl, pr = wt*gr[col];
						meanL += ok ? pl : 0.0;
						meanR += ok ? pr : 0.0;
						totalWeight += ok ? wt : 0.0;
					}
				}
				if (totalWeight < 1e-10) result = P.bad_ret;
				else {
					meanL /= totalWeight;
					meanR /= totalWeight;
					double sum1 = 0, sum2 = 0, sum3 = 0;
#pragma unroll 1
					for (int row = 0; row < WS; ++row) {
						double gr[WS];
#pragma unroll
						for (int col = 0; col < WS; ++col) {
							const int gx = cx - R + col, gy = cy - R + row;
							gr[col] = (gx >= 0 && gy >= 0 && gx < OW && gy < OH) ? Rv.gray_tv[(size_t)gy*OW + gx] : nan;
						}
#pragma unroll
						for (int col = 0; col < WS; ++col) {
							const double gl = CS.lt[row][i + col], wt = CS.w[i][row*WP + col];
							const bool ok = gl == gl && gr[col] == gr[col] && wt > P.weight_cutoff;
							const double a = wt*gl - meanL, b = wt*gr[col] - meanR;
							const double ab = a*b, aa = a*a, bb = b*b;
							sum1 += ok ? ab : 0.0;
							sum2 += ok ? aa : 0.0;
							sum3 += ok ? bb : 0.0;
						}
					}
					const double v = 255*(1.0 - fabs(sum1) / sqrt(sum2 * sum3));
					result = (v < P.max_color_diff) ? v : P.max_color_diff;
				}
			}
			crow[k] = result;
		}
	}
	block_count_add(&cnt->n_eval_device, n_dev);
}

bool launch_twoview_list_cost(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                              int y0, int nrows, const double *wbuf, const uint8_t *full_oth,
                              const int32_t *count, const uint32_t *cand, double *cost, int cmax, Counters *cnt)
{
	const int tiles = (width + LC_TP - 1)/LC_TP;
	const dim3 grid((unsigned)(tiles*nrows));
#define SRH_LC_LAUNCH(RR)                                                                                   \
	{                                                                                                       \
		/* per device, hence on every launch */                                                             \
		(void)hipFuncSetAttribute((const void *)twoview_list_cost_kernel<RR>,                                                \
		                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ListSmem<RR>));                  \
		hipLaunchKernelGGL(twoview_list_cost_kernel<RR>, grid, dim3(LC_THREADS), sizeof(ListSmem<RR>), st,  \
		                   views, ref, oth, P, y0, nrows, wbuf, full_oth, count, cand, cost, cmax, cnt);    \
		return true;                                                                                        \
	}
	switch (P.window_radius) {
	case 1: SRH_LC_LAUNCH(1)
	case 2: SRH_LC_LAUNCH(2)
	case 3: SRH_LC_LAUNCH(3)
	case 4: SRH_LC_LAUNCH(4)
	case 5: SRH_LC_LAUNCH(5)
	default: return false;
	}
#undef SRH_LC_LAUNCH
}

// ------------------------------------------------------------------ list scan
#define LS_QN 16

__global__ void twoview_list_scan_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P,
                                         int y0, int nrows, const int32_t *__restrict__ count,
                                         const uint32_t *__restrict__ cand, const double *__restrict__ cost, int cmax)
{
	const ViewDev &L = views[ref];
	const ViewDev &Rv = views[oth];
	const int W = L.w;
	const size_t q = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
	if (q >= (size_t)nrows*W) return;
	const int x = (int)(q % W), y = y0 + (int)(q / W);
	const size_t pv = (size_t)y*W + x;
	double depth = __builtin_nan("");
	if (L.mask[pv] == 1) {
		const int n = count[q] < cmax ? count[q] : cmax;
		const uint32_t *clist = cand + q*(size_t)cmax;
		const double *crow = cost + q*(size_t)cmax;
		double minCost = __builtin_inf(), secondBest = __builtin_inf();
		uint32_t win = 0xffffffffu, prev = 0xffffffffu;
		for (int k0 = 0; k0 < n; k0 += LS_QN) {
			uint32_t e[LS_QN];
			double c[LS_QN];
#pragma unroll
			for (int j = 0; j < LS_QN; ++j) {
				const bool in = k0 + j < n;
				e[j] = in ? clist[k0 + j] : 0xffffffffu;
				c[j] = in ? crow[k0 + j] : __builtin_inf();       // (duplicates hold stale values: skipped below)
			}
#pragma unroll
			for (int j = 0; j < LS_QN; ++j) {
				if (k0 + j < n && e[j] != prev) {
					if (c[j] + P.wta_margin < minCost) {               // twoviewstereo.cpp:293-301
						secondBest = minCost;
						minCost = c[j];
						win = e[j];
					}
				}
				if (k0 + j < n) prev = e[j];
			}
		}
		if (win != 0xffffffffu) {
			const Ray ray = cam_unproject(L.cam, (x + 0.5) / P.image_scale, (y + 0.5) / P.image_scale);
			depth = candidate_depth(L.cam, Rv.cam, P, ray, (int)(win & 0xffffu), (int)(win >> 16));
		}
		if (minCost > P.second_best_factor*secondBest)                 // twoviewstereo.cpp:304-305
			depth = __builtin_inf();
	}
	L.depth[pv] = depth;
}

void launch_twoview_list_scan(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                              int y0, int nrows, const int32_t *count, const uint32_t *cand, const double *cost, int cmax)
{
	const size_t n = (size_t)nrows*width;
	hipLaunchKernelGGL(twoview_list_scan_kernel, dim3((unsigned)((n + 127)/128)), dim3(128), 0, st,
	                   views, ref, oth, P, y0, nrows, count, cand, cost, cmax);
}

} // namespace srh
