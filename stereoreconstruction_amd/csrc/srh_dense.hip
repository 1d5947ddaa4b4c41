// srh_dense.hip -- the tuned TwoView kernels for row-aligned epipolar geometry
// (every candidate of pixel (x,y) lies on row y of the other view: rectified rigs,
// BASELINE configs C2/C3).  Bit-identical costs to the general kernels: the same
// double operations in the same order, only organised so that the window data is
// shared through LDS and registers.
//
//   edge_planes_kernel        colour distances between 8-neighbours, once per view
//   geodesic_reg_kernel<R>    GeodesicWeight windows, window held in registers     (8(a) #2)
//   twoview_extent_kernel     per pixel: candidate column range, row-alignment test (#6,#7)
//   twoview_dense_cost_kernel weighted NCC for every candidate column, LDS-tiled    (#8)
//   twoview_scan_kernel       curve walk + cost look-up + running-min WTA + depth   (#9,#10)
#include "srh_internal.hpp"
#include "srh_geom.hpp"
#include "srh_walk.hpp"

namespace srh {

// ------------------------------------------------------------------ edge planes
// edges[k][y*W+x], k: 0 = E  (x,y)-(x+1,y)      1 = S  (x,y)-(x,y+1)
//                     2 = SE (x,y)-(x+1,y+1)    3 = SW (x,y)-(x-1,y+1)
// +inf when the second pixel is outside the image: such an edge can never relax a
// cell (std::min keeps the old value), which is exactly the reference's "skip
// INVALID pixels" (geodesicweight.cpp:76-77,87).
__device__ __forceinline__ double color_dist_u(uint32_t a, uint32_t b) {
	const double dr = (double)((int)(a & 255u) - (int)(b & 255u));
	const double dg = (double)((int)((a >> 8) & 255u) - (int)((b >> 8) & 255u));
	const double db = (double)((int)((a >> 16) & 255u) - (int)((b >> 16) & 255u));
	return sqrt(dr*dr + dg*dg + db*db);
}

__global__ void edge_planes_kernel(const uint32_t *__restrict__ rgba, int W, int H, double *__restrict__ edges)
{
	const size_t n = (size_t)W*H;
	const double inf = __builtin_inf();
	for (size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) {
		const int x = (int)(i % (size_t)W), y = (int)(i / (size_t)W);
		const uint32_t c = rgba[i];
		edges[0*n + i] = (x + 1 < W) ? color_dist_u(rgba[i + 1], c) : inf;
		edges[1*n + i] = (y + 1 < H) ? color_dist_u(rgba[i + W], c) : inf;
		edges[2*n + i] = (x + 1 < W && y + 1 < H) ? color_dist_u(rgba[i + W + 1], c) : inf;
		edges[3*n + i] = (x >= 1 && y + 1 < H) ? color_dist_u(rgba[i + W - 1], c) : inf;
	}
}

void launch_edge_planes(hipStream_t st, const uint32_t *rgba, int w, int h, double *edges) {
	size_t n = (size_t)w*h;
	size_t b = (n + 255)/256; if (b > 2048) b = 2048; if (b < 1) b = 1;
	hipLaunchKernelGGL(edge_planes_kernel, dim3((unsigned)b), dim3(256), 0, st, rgba, w, h, edges);
}

// ------------------------------------------------------------------ geodesic windows in registers
// One thread per reference pixel, GW_TW pixels of one row per workgroup.  The four
// edge planes of the (TW+2R) x (2R+1) neighbourhood are staged in LDS; the
// (2R+1)^2 window lives in registers (fully unrolled sweeps).  Cells outside the
// image keep geodesic_init because all their edges are +inf.
#define GW_TW 64

template <int R>
__global__ __launch_bounds__(GW_TW)
void geodesic_reg_kernel(const ViewDev *__restrict__ views, int ref, const double *__restrict__ edges,
                         srh_params P, int y0, int nrows, double *__restrict__ wbuf, size_t wstride)
{
	constexpr int WS = 2*R + 1;
	constexpr int TWD = GW_TW + 2*R;            // tile width
	const ViewDev &V = views[ref];
	const int W = V.w, H = V.h;
	const size_t n = (size_t)W*H;
	const int tiles_per_row = (W + GW_TW - 1)/GW_TW;
	const int trow = blockIdx.x / tiles_per_row;
	const int x0 = (blockIdx.x % tiles_per_row)*GW_TW;
	const int cy = y0 + trow;
	if (trow >= nrows) return;

	__shared__ double eE[WS][TWD], eS[WS][TWD], eSE[WS][TWD], eSW[WS][TWD];
	const double inf = __builtin_inf();
	for (int idx = threadIdx.x; idx < WS*TWD; idx += GW_TW) {
		const int ty = idx / TWD, tx = idx % TWD;
		const int gx = x0 - R + tx, gy = cy - R + ty;
		const bool in = gx >= 0 && gy >= 0 && gx < W && gy < H;
		const size_t gi = (size_t)gy*W + gx;
		eE[ty][tx]  = in ? edges[0*n + gi] : inf;
		eS[ty][tx]  = in ? edges[1*n + gi] : inf;
		eSE[ty][tx] = in ? edges[2*n + gi] : inf;
		eSW[ty][tx] = in ? edges[3*n + gi] : inf;
	}
	__syncthreads();

	const int i = threadIdx.x;
	const int cx = x0 + i;
	if (cx >= W) return;
	if (V.mask[(size_t)cy*W + cx] != 1) return;

	double w[WS][WS];
#pragma unroll
	for (int a = 0; a < WS; ++a)
#pragma unroll
		for (int b = 0; b < WS; ++b) w[a][b] = P.geodesic_init;
	w[R][R] = 0.0;

#pragma unroll 1
	for (int iter = 0; iter < P.geodesic_iters; ++iter) {
		// forward pass, K1 = (-1,-1) (0,-1) (1,-1) (-1,0)   (geodesicweight.cpp:73-97)
#pragma unroll
		for (int yy = 0; yy < WS; ++yy) {
			// keep the edge loads of each window row next to their use (no hoisting of
			// the loop-invariant LDS reads out of the sweep, which would spill)
			asm volatile("" ::: "memory");
#pragma unroll
			for (int xx = 0; xx < WS; ++xx) {
				const int tx = i + xx;                      // tile column of window cell xx
				double wt = w[yy][xx];
				// a cell outside the image is never relaxed: freeze it by testing its own
				// position through an edge that touches it (all its edges are +inf)
				if (yy > 0 && xx > 0)      { const double c = w[yy-1][xx-1] + eSE[yy-1][tx-1]; wt = (c < wt) ? c : wt; }
				if (yy > 0)                { const double c = w[yy-1][xx]   + eS[yy-1][tx];    wt = (c < wt) ? c : wt; }
				if (yy > 0 && xx < WS - 1) { const double c = w[yy-1][xx+1] + eSW[yy-1][tx+1]; wt = (c < wt) ? c : wt; }
				if (xx > 0)                { const double c = w[yy][xx-1]   + eE[yy][tx-1];    wt = (c < wt) ? c : wt; }
				w[yy][xx] = wt;
			}
		}
		// backward pass, K2 = (-1,1) (0,1) (1,1) (1,0)      (geodesicweight.cpp:99-125)
#pragma unroll
		for (int yy = WS - 1; yy >= 0; --yy) {
			asm volatile("" ::: "memory");
#pragma unroll
			for (int xx = WS - 1; xx >= 0; --xx) {
				const int tx = i + xx;
				double wt = w[yy][xx];
				if (yy < WS - 1 && xx > 0)      { const double c = w[yy+1][xx-1] + eSW[yy][tx]; wt = (c < wt) ? c : wt; }
				if (yy < WS - 1)                { const double c = w[yy+1][xx]   + eS[yy][tx];  wt = (c < wt) ? c : wt; }
				if (yy < WS - 1 && xx < WS - 1) { const double c = w[yy+1][xx+1] + eSE[yy][tx]; wt = (c < wt) ? c : wt; }
				if (xx < WS - 1)                { const double c = w[yy][xx+1]   + eE[yy][tx];  wt = (c < wt) ? c : wt; }
				w[yy][xx] = wt;
			}
		}
	}
	double *wb = wbuf + ((size_t)trow*W + cx);
#pragma unroll
	for (int a = 0; a < WS; ++a)
#pragma unroll
		for (int b = 0; b < WS; ++b)
			wb[(size_t)(a*WS + b)*wstride] = w[a][b];
	// exponential weighting (geodesicweight.cpp:128-130), rolled: one exp body instead of (2R+1)^2
#pragma unroll 1
	for (int t = 0; t < WS*WS; ++t)
		wb[(size_t)t*wstride] = exp(-wb[(size_t)t*wstride] / P.geodesic_sigma);
}

bool launch_geodesic_reg(hipStream_t st, const ViewDev *views, int ref, int width, const double *edges,
                         const srh_params &P, int y0, int nrows, double *wbuf, size_t wstride)
{
	const int tiles = (width + GW_TW - 1)/GW_TW;
	const dim3 grid((unsigned)(tiles*nrows)), block(GW_TW);
	switch (P.window_radius) {
	case 5: hipLaunchKernelGGL(geodesic_reg_kernel<5>, grid, block, 0, st, views, ref, edges, P, y0, nrows, wbuf, wstride); return true;
	case 2: hipLaunchKernelGGL(geodesic_reg_kernel<2>, grid, block, 0, st, views, ref, edges, P, y0, nrows, wbuf, wstride); return true;
	default: return false;
	}
}

// ------------------------------------------------------------------ extent pass
struct ExtentVisitor {
	int y, xmin, xmax, n;
	bool aligned;
	__device__ __forceinline__ void operator()(int cx, int cy) {
		if (cy != y) aligned = false;
		xmin = cx < xmin ? cx : xmin;
		xmax = cx > xmax ? cx : xmax;
		++n;
	}
};

__global__ void twoview_extent_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P,
                                      int y0, int nrows, Extent *__restrict__ ext, Counters *__restrict__ cnt,
                                      int *__restrict__ max_span)
{
	const ViewDev &L = views[ref];
	const ViewDev &Rv = views[oth];
	const int W = L.w;
	const size_t q = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
	unsigned n_eval = 0, n_pix = 0, bad = 0;
	int span = 0;
	if (q < (size_t)nrows*W) {
		const int x = (int)(q % W), y = y0 + (int)(q / W);
		Extent e; e.xmin = 0; e.xmax = -1;
		if (L.mask[(size_t)y*W + x] == 1) {
			n_pix = 1;
			const Ray ray = cam_unproject(L.cam, (x + 0.5) / P.image_scale, (y + 0.5) / P.image_scale);
			ExtentVisitor vis = { y, 2147483647, -2147483647, 0, true };
			walk_curve<false>(ray, L.cam, Rv, P, vis);
			n_eval = vis.n;
			if (vis.n > 0) { e.xmin = vis.xmin; e.xmax = vis.xmax; span = vis.xmax - vis.xmin + 1; }
			if (!vis.aligned) bad = 1;
		}
		ext[q] = e;
	}
	__shared__ int s_span;
	if (threadIdx.x == 0) s_span = 0;
	__syncthreads();
	if (span > 0) atomicMax(&s_span, span);
	__syncthreads();
	if (threadIdx.x == 0 && s_span > 0) atomicMax(max_span, s_span);
	block_count_add(&cnt->n_eval, n_eval);
	block_count_add(&cnt->n_pixels, n_pix);
	block_count_add(&cnt->not_row_aligned, bad);
}

void launch_twoview_extent(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                           int y0, int nrows, Extent *ext, Counters *cnt, int *max_span)
{
	const size_t n = (size_t)nrows*width;
	hipLaunchKernelGGL(twoview_extent_kernel, dim3((unsigned)((n + 127)/128)), dim3(128), 0, st,
	                   views, ref, oth, P, y0, nrows, ext, cnt, max_span);
}

// ------------------------------------------------------------------ dense cost
// Workgroup = DC_TP consecutive pixels of one row x DC_G lanes per pixel.  LDS holds
// the DC_TP support windows ([tap][pixel]), the (2R+1) rows of the other view's
// gray_tv plane over the union of the candidate ranges (+R margin) and the
// reference view's rows.  Each lane evaluates blocks of DC_NCB adjacent candidate
// columns: a right-row segment of NCB+2R values is loaded into registers once per
// window row and reused by the NCB candidates and 2R+1 taps.
//
// Fast form (all taps of the window usable on both sides): meanL, totalWeight,
// sum2 and a_t = w_t*gl_t - meanL do not depend on the candidate, so per tap and
// candidate only  p = w*gr; meanR += p  and  b = p - meanR; sum1 += a*b; sum3 += b*b
// remain -- the same operations, in the same order, as twoviewstereo.cpp:917-976.
#define DC_TP 32
#define DC_G 8
#define DC_NCB 8
#define DC_THREADS (DC_TP*DC_G)
#define DC_CHUNK 320               // candidate columns staged per pass (>= TP + D for C3)

template <int R>
struct DenseSmem {
	static constexpr int WS = 2*R + 1;
	static constexpr int T = WS*WS;
	static constexpr int RW = DC_CHUNK + 2*R + DC_NCB;      // right tile width
	static constexpr int LW = DC_TP + 2*R;                  // left tile width
	double w[T][DC_TP];
	double rt[WS][RW];
	double lt[WS][LW];
	double meanL[DC_TP], totalW[DC_TP], sum2[DC_TP];
	int lall[DC_TP];
	unsigned char rfull[RW];
	unsigned char colok[RW];
};

// general (any validity pattern) cost of one candidate from the LDS tiles
template <int R>
__device__ __noinline__ double dense_cost_general(const DenseSmem<R> &S, int i, int rc,
                                                  double weight_cutoff, double bad_ret, double max_color_diff)
{
	constexpr int WS = 2*R + 1;
	double meanL = 0, meanR = 0, totalWeight = 0.0;
#pragma unroll 1
	for (int row = 0; row < WS; ++row)
#pragma unroll 1
		for (int col = 0; col < WS; ++col) {
			const double gl = S.lt[row][i + col];
			const double gr = S.rt[row][rc + col];
			const double weight = S.w[row*WS + col][i];
			if (gl == gl && gr == gr && weight > weight_cutoff) {
				meanL += weight*gl;
				meanR += weight*gr;
				totalWeight += weight;
			}
		}
	if (totalWeight < 1e-10) return bad_ret;
	meanL /= totalWeight;
	meanR /= totalWeight;
	double sum1 = 0, sum2 = 0, sum3 = 0;
#pragma unroll 1
	for (int row = 0; row < WS; ++row)
#pragma unroll 1
		for (int col = 0; col < WS; ++col) {
			const double gl = S.lt[row][i + col];
			const double gr = S.rt[row][rc + col];
			const double weight = S.w[row*WS + col][i];
			if (gl == gl && gr == gr && weight > weight_cutoff) {
				const double pgl = weight*gl;
				const double pgr = weight*gr;
				sum1 += (pgl - meanL)*(pgr - meanR);
				sum2 += (pgl - meanL)*(pgl - meanL);
				sum3 += (pgr - meanR)*(pgr - meanR);
			}
		}
	const double v = 255*(1.0 - fabs(sum1) / sqrt(sum2 * sum3));
	return (v < max_color_diff) ? v : max_color_diff;
}

template <int R>
__global__ __launch_bounds__(DC_THREADS, 2)
void twoview_dense_cost_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P,
                               int y0, int nrows, const double *__restrict__ wbuf, size_t wstride,
                               const Extent *__restrict__ ext, double *__restrict__ cost, int cstride,
                               Counters *__restrict__ cnt)
{
	constexpr int WS = 2*R + 1;
	constexpr int T = WS*WS;
	typedef DenseSmem<R> Smem;
	extern __shared__ __align__(16) unsigned char smem_raw[];
	Smem &S = *reinterpret_cast<Smem *>(smem_raw);

	const ViewDev &L = views[ref];
	const ViewDev &Rv = views[oth];
	const int W = L.w, H = L.h;
	const int tiles_per_row = (W + DC_TP - 1)/DC_TP;
	const int trow = blockIdx.x / tiles_per_row;
	const int x0 = (blockIdx.x % tiles_per_row)*DC_TP;
	const int y = y0 + trow;
	const int tid = threadIdx.x;
	const int i = tid / DC_G;                  // pixel within the tile
	const int g = tid % DC_G;                  // lane within the pixel
	const int x = x0 + i;
	const size_t qbase = (size_t)trow*W + x0;  // band-relative index of the tile's first pixel
	const double nan = __builtin_nan("");

	// ---- stage windows and the reference rows
	for (int idx = tid; idx < T*DC_TP; idx += DC_THREADS) {
		const int t = idx / DC_TP, pi = idx % DC_TP;
		S.w[t][pi] = (x0 + pi < W) ? wbuf[(size_t)t*wstride + qbase + pi] : 0.0;
	}
	for (int idx = tid; idx < WS*Smem::LW; idx += DC_THREADS) {
		const int ty = idx / Smem::LW, tx = idx % Smem::LW;
		const int gx = x0 - R + tx, gy = y - R + ty;
		S.lt[ty][tx] = (gx >= 0 && gy >= 0 && gx < W && gy < H) ? L.gray_tv[(size_t)gy*W + gx] : nan;
	}
	// union of the candidate ranges of the tile
	__shared__ int s_cmin, s_cmax;
	if (tid == 0) { s_cmin = 2147483647; s_cmax = -2147483647; }
	__syncthreads();
	Extent e; e.xmin = 0; e.xmax = -1;
	if (x < W) e = ext[qbase + i];
	if (g == 0 && e.xmax >= e.xmin) { atomicMin(&s_cmin, e.xmin); atomicMax(&s_cmax, e.xmax); }
	__syncthreads();
	const int cmin = s_cmin, cmax = s_cmax;

	// ---- per-pixel constants of the fast form (one lane per pixel)
	if (g == 0) {
		bool all = (x < W) && (e.xmax >= e.xmin);
		double mL = 0, tw = 0;
#pragma unroll 1
		for (int row = 0; row < WS; ++row)
#pragma unroll 1
			for (int col = 0; col < WS; ++col) {
				const double gl = S.lt[row][i + col];
				const double wt = S.w[row*WS + col][i];
				if (!(gl == gl && wt > P.weight_cutoff)) all = false;
				mL += wt*gl;
				tw += wt;
			}
		double s2 = 0;
		if (all && !(tw < 1e-10)) {
			mL /= tw;
#pragma unroll 1
			for (int row = 0; row < WS; ++row)
#pragma unroll 1
				for (int col = 0; col < WS; ++col) {
					const double a = S.w[row*WS + col][i]*S.lt[row][i + col] - mL;
					s2 += a*a;
				}
		} else all = false;
		S.meanL[i] = mL; S.totalW[i] = tw; S.sum2[i] = s2; S.lall[i] = all ? 1 : 0;
	}

	unsigned n_dev = 0;
	const Smem &CS = S;
	for (int cs = cmin; cs <= cmax; cs += DC_CHUNK) {
		__syncthreads();   // previous chunk fully consumed (and the stores above visible)
		// ---- stage the other view's rows for columns [cs-R, cs+CHUNK+R+NCB)
		for (int idx = tid; idx < WS*Smem::RW; idx += DC_THREADS) {
			const int ty = idx / Smem::RW, tx = idx % Smem::RW;
			const int gx = cs - R + tx, gy = y - R + ty;
			S.rt[ty][tx] = (gx >= 0 && gy >= 0 && gx < Rv.w && gy < Rv.h) ? Rv.gray_tv[(size_t)gy*Rv.w + gx] : nan;
		}
		__syncthreads();
		for (int tx = tid; tx < Smem::RW; tx += DC_THREADS) {
			bool ok = true;
#pragma unroll 1
			for (int ty = 0; ty < WS; ++ty) { const double v = S.rt[ty][tx]; ok = ok && (v == v); }
			S.colok[tx] = ok ? 1 : 0;
		}
		__syncthreads();
		for (int tx = tid; tx < Smem::RW; tx += DC_THREADS) {
			// rfull[k]: window of candidate column cs+k fully usable (tile columns k .. k+2R)
			bool ok = tx + 2*R < Smem::RW;
			for (int k = 0; ok && k < WS; ++k) ok = S.colok[tx + k] != 0;
			S.rfull[tx] = ok ? 1 : 0;
		}
		__syncthreads();

		if (x < W && e.xmax >= e.xmin) {
			const int lo = e.xmin > cs ? e.xmin : cs;
			const int hi = e.xmax < cs + DC_CHUNK - 1 ? e.xmax : cs + DC_CHUNK - 1;
			const int nblocks = hi >= lo ? (hi - lo + DC_NCB)/DC_NCB : 0;
			double *crow = cost + (qbase + i)*(size_t)cstride;
			for (int b = g; b < nblocks; b += DC_G) {
				const int c0 = lo + b*DC_NCB;
				const int nv = (hi - c0 + 1) < DC_NCB ? (hi - c0 + 1) : DC_NCB;
				const int rc = c0 - cs;                 // tile column of the window's left edge
				bool fast = CS.lall[i] != 0;
				for (int j = 0; j < nv; ++j) fast = fast && CS.rfull[rc + j] != 0;
				n_dev += nv;
				if (fast) {
					const double mL = CS.meanL[i], tw = CS.totalW[i], s2 = CS.sum2[i];
					double acc[DC_NCB];
#pragma unroll
					for (int j = 0; j < DC_NCB; ++j) acc[j] = 0.0;
#pragma unroll 1
					for (int row = 0; row < WS; ++row) {
						double r[DC_NCB + 2*R];
#pragma unroll
						for (int k = 0; k < DC_NCB + 2*R; ++k) r[k] = CS.rt[row][rc + k];
#pragma unroll
						for (int col = 0; col < WS; ++col) {
							const double wt = CS.w[row*WS + col][i];
#pragma unroll
							for (int j = 0; j < DC_NCB; ++j) acc[j] += wt*r[col + j];   // meanR += weight*gray
							__builtin_amdgcn_sched_barrier(0);
						}
					}
					double mR[DC_NCB], s1[DC_NCB], s3[DC_NCB];
#pragma unroll
					for (int j = 0; j < DC_NCB; ++j) { mR[j] = acc[j]/tw; s1[j] = 0.0; s3[j] = 0.0; }
#pragma unroll 1
					for (int row = 0; row < WS; ++row) {
						double r[DC_NCB + 2*R];
#pragma unroll
						for (int k = 0; k < DC_NCB + 2*R; ++k) r[k] = CS.rt[row][rc + k];
#pragma unroll
						for (int col = 0; col < WS; ++col) {
							const double wt = CS.w[row*WS + col][i];
							const double a = wt*CS.lt[row][i + col] - mL;     // pixel_gray_l - meanL
#pragma unroll
							for (int j = 0; j < DC_NCB; ++j) {
								const double bb = wt*r[col + j] - mR[j];      // pixel_gray_r - meanR
								s1[j] += a*bb;
								s3[j] += bb*bb;
							}
							__builtin_amdgcn_sched_barrier(0);
						}
					}
#pragma unroll
					for (int j = 0; j < DC_NCB; ++j) {
						if (j < nv) {
							const double v = 255*(1.0 - fabs(s1[j]) / sqrt(s2 * s3[j]));
							crow[c0 + j - e.xmin] = (v < P.max_color_diff) ? v : P.max_color_diff;
						}
						__builtin_amdgcn_sched_barrier(0);
					}
				} else {
					for (int j = 0; j < nv; ++j)
						crow[c0 + j - e.xmin] = dense_cost_general<R>(CS, i, rc + j, P.weight_cutoff, P.bad_ret, P.max_color_diff);
				}
			}
		}
	}
	block_count_add(&cnt->n_eval_device, n_dev);
}

bool launch_twoview_dense_cost(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                               int y0, int nrows, const double *wbuf, size_t wstride,
                               const Extent *ext, double *cost, int cstride, Counters *cnt)
{
	const int tiles = (width + DC_TP - 1)/DC_TP;
	const dim3 grid((unsigned)(tiles*nrows)), block(DC_THREADS);
	switch (P.window_radius) {
	case 5: {
		static bool attr5 = false;
		if (!attr5) { (void)hipFuncSetAttribute((const void *)twoview_dense_cost_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(DenseSmem<5>)); attr5 = true; }
		hipLaunchKernelGGL(twoview_dense_cost_kernel<5>, grid, block, sizeof(DenseSmem<5>), st,
		                   views, ref, oth, P, y0, nrows, wbuf, wstride, ext, cost, cstride, cnt);
		return true; }
	case 2: {
		static bool attr2 = false;
		if (!attr2) { (void)hipFuncSetAttribute((const void *)twoview_dense_cost_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(DenseSmem<2>)); attr2 = true; }
		hipLaunchKernelGGL(twoview_dense_cost_kernel<2>, grid, block, sizeof(DenseSmem<2>), st,
		                   views, ref, oth, P, y0, nrows, wbuf, wstride, ext, cost, cstride, cnt);
		return true; }
	default: return false;
	}
}

// ------------------------------------------------------------------ scan: walk + look-up + WTA
struct TwoViewLookupVisitor {
	const double *crow;
	int xmin;
	const srh_params &P;
	double minCost, secondBest;
	int wx, wy;
	__device__ __forceinline__ void operator()(int cx, int cy) {
		const double cost = crow[cx - xmin];
		if (cost + P.wta_margin < minCost) {                   // twoviewstereo.cpp:293-301
			secondBest = minCost;
			minCost = cost;
			wx = cx; wy = cy;
		}
	}
};

__global__ void twoview_scan_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P,
                                    int y0, int nrows, const Extent *__restrict__ ext,
                                    const double *__restrict__ cost, int cstride)
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
		const Ray ray = cam_unproject(L.cam, (x + 0.5) / P.image_scale, (y + 0.5) / P.image_scale);
		TwoViewLookupVisitor vis = { cost + q*(size_t)cstride, ext[q].xmin, P, __builtin_inf(), __builtin_inf(), -1, -1 };
		walk_curve<false>(ray, L.cam, Rv, P, vis);
		if (vis.wx >= 0)
			depth = candidate_depth(L.cam, Rv.cam, P, ray, vis.wx, vis.wy);
		if (vis.minCost > P.second_best_factor*vis.secondBest)
			depth = __builtin_inf();
	}
	L.depth[pv] = depth;
}

void launch_twoview_scan(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                         int y0, int nrows, const Extent *ext, const double *cost, int cstride)
{
	const size_t n = (size_t)nrows*width;
	hipLaunchKernelGGL(twoview_scan_kernel, dim3((unsigned)((n + 127)/128)), dim3(128), 0, st,
	                   views, ref, oth, P, y0, nrows, ext, cost, cstride);
}

} // namespace srh
