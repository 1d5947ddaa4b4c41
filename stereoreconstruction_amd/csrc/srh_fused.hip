// srh_fused.hip -- TwoViewStereo::computeCostVolumes for row-aligned rigs in ONE kernel
// (twoviewstereo.cpp:260-333 / :431-501): epipolar geometry, candidate rasterisation, weighted NCC of
// every candidate column and the running-min WTA with its ratio test, per 16-pixel tile of a row,
// without a cost row or a candidate list ever leaving the CU.  (The three-kernel form in
// srh_dense.hip staged 4.4 GB of cost rows per 1920x1080x256 direction through HBM and read them
// back in a scan kernel.)  Same double operations in the same order: bit-identical results.
//
// Workgroup = 16 consecutive pixels of one row x 16 lanes per pixel (4 pixels per wave, pixel-fastest).
//   A  all lanes   per (pixel, label): projection of the label's 3-D point into the other view
//                  (pinhole_project_label, srh_walk.hpp) -> x2 per label in LDS
//   B  one wave    per pixel, sequentially: which labels the curve keeps (>= 1 px from the last kept
//                  point, twoviewstereo.cpp:1027) -> first kept column K0, direction, bitmask of the
//                  later kept columns (segment joints), visited column range
//   C  all lanes   stage the support windows, the reference rows and the other view's rows + mask row
//   D  one wave    per-pixel constants of the fast cost form; others: which columns have full windows
//   E  all lanes   cost of every visited column: blocks of 8 adjacent columns per lane (the register-
//                  tiled loop of srh_dense.hip), incomplete windows in the select form -> costs in LDS
//   F  one wave    per pixel: the reference's visiting order is replayed from the joint bitmask
//                  (ascending inside a segment, segments in label order; a column seen before cannot
//                  change the running minimum again and is skipped), running min + ratio test,
//                  depth of the winner only (twoviewstereo.cpp:293-305)
// A pixel whose curve leaves its row, is not monotone or does not fit the LDS tile raises
// Counters::not_row_aligned and the host repeats the direction on the other kernels.
#include "srh_internal.hpp"
#include "srh_geom.hpp"
#include "srh_walk.hpp"

namespace srh {

#define FZ_TP 16
#define FZ_G 16
#define FZ_THREADS (FZ_TP*FZ_G)
#define FZ_NCB 8

template <int R, int MAXC>
struct FusedSmem {
	static constexpr int WS = 2*R + 1;
	static constexpr int T = WS*WS;
	static constexpr int WP = (WS + 1) & ~1;                 // taps per window row, padded even (16-byte rows)
	static constexpr int WPIX = WS*WP;
	static constexpr int RW = (MAXC + FZ_TP - 1 + 2*R + 1) & ~1;   // other-view columns staged per row
	static constexpr int LW = FZ_TP + 2*R;
	static constexpr int JW = MAXC/32 + 1;
	static_assert((MAXC & (MAXC - 1)) == 0 && MAXC <= 512, "cost rows are rotated modulo MAXC; work-list entry = pixel*512 + column");
	double w[FZ_TP][WPIX];                                  // w[pixel][row*WP + col]
	double cost[FZ_TP*MAXC];                                // A/B: x2 of label d; E/F: cost of column corg + k  (slot(), rotated per pixel)
	double rt[WS][RW];                                      // other view, rows y-R..y+R, columns cmin-R ..
	double lt[WS][LW];                                      // reference view, columns x0-R ..
	double meanL[FZ_TP], totalW[FZ_TP], sum2[FZ_TP];
	int lall[FZ_TP];
	int k0[FZ_TP], dir[FZ_TP], ilast[FZ_TP], nmerge[FZ_TP]; // first kept column, +1/-1, highest joint index, merged joints
	int corg[FZ_TP], ca[FZ_TP], cb[FZ_TP];                  // cost-row origin (even), visited in-image column range (empty: cb < ca)
	int pflag[FZ_TP];
	unsigned joints[FZ_TP][JW];                             // bit i: a kept point (not the first) at column k0 + dir*i
	unsigned char rfull[RW], colok[RW], mrow[RW];
	static constexpr int GL_CAP = 1024;
	unsigned short glist[GL_CAP];
	int glist_n, cmin, cmax, need_general, bad;
	// element k of pixel p's row, rotated by 2p doubles: lanes that hold different pixels and read the
	// same k hit different banks (a row is MAXC*8 bytes = a multiple of all 64 banks)
	__device__ __forceinline__ static int slot(int p, int k) { return p*MAXC + ((k + 2*p) & (MAXC - 1)); }
};

// cost of one candidate from the LDS tiles, any validity pattern (twoviewstereo.cpp:909-977): skipped taps
// add +0.0 to every sum, which leaves each partial sum bit-for-bit unchanged.
template <int R, int MAXC>
__device__ __noinline__ double fused_cost_general(const FusedSmem<R, MAXC> &S, int p, int rc,
                                                  double weight_cutoff, double bad_ret, double max_color_diff)
{
	constexpr int WS = 2*R + 1;
	constexpr int WP = FusedSmem<R, MAXC>::WP;
	double meanL = 0, meanR = 0, totalWeight = 0.0;
#pragma unroll 1
	for (int row = 0; row < WS; ++row) {
		double gl[WS], gr[WS], wt[WS];
#pragma unroll
		for (int col = 0; col < WS; ++col) { gl[col] = S.lt[row][p + col]; gr[col] = S.rt[row][rc + col]; wt[col] = S.w[p][row*WP + col]; }
#pragma unroll
		for (int col = 0; col < WS; ++col) {
			const bool ok = gl[col] == gl[col] && gr[col] == gr[col] && wt[col] > weight_cutoff;
			const double pl = wt[col]*gl[col], pr = wt[col]*gr[col];
			meanL += ok ? pl : 0.0;
			meanR += ok ? pr : 0.0;
			totalWeight += ok ? wt[col] : 0.0;
		}
	}
	if (totalWeight < 1e-10) return bad_ret;
	meanL /= totalWeight;
	meanR /= totalWeight;
	double sum1 = 0, sum2 = 0, sum3 = 0;
#pragma unroll 1
	for (int row = 0; row < WS; ++row) {
		double gl[WS], gr[WS], wt[WS];
#pragma unroll
		for (int col = 0; col < WS; ++col) { gl[col] = S.lt[row][p + col]; gr[col] = S.rt[row][rc + col]; wt[col] = S.w[p][row*WP + col]; }
#pragma unroll
		for (int col = 0; col < WS; ++col) {
			const bool ok = gl[col] == gl[col] && gr[col] == gr[col] && wt[col] > weight_cutoff;
			const double a = wt[col]*gl[col] - meanL;
			const double b = wt[col]*gr[col] - meanR;
			const double ab = a*b, aa = a*a, bb = b*b;
			sum1 += ok ? ab : 0.0;
			sum2 += ok ? aa : 0.0;
			sum3 += ok ? bb : 0.0;
		}
	}
	const double v = 255*(1.0 - fabs(sum1) / sqrt(sum2 * sum3));
	return (v < max_color_diff) ? v : max_color_diff;
}

// next set joint bit at index >= from, or -1
template <int JW>
__device__ __forceinline__ int next_joint(const unsigned *J, int from) {
	int wi = from >> 5;
	if (wi >= JW) return -1;
	unsigned m = J[wi] & (0xffffffffu << (from & 31));
	while (true) {
		if (m) return wi*32 + (__ffs((int)m) - 1);
		if (++wi >= JW) return -1;
		m = J[wi];
	}
}

template <int R, int MAXC>
__global__ __launch_bounds__(FZ_THREADS, 2)
void twoview_fused_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P,
                          int y0, int nrows, const double *__restrict__ wbuf, const double *__restrict__ tnum,
                          Counters *__restrict__ cnt)
{
	typedef FusedSmem<R, MAXC> Smem;
	constexpr int WS = Smem::WS, T = Smem::T, WP = Smem::WP;
	constexpr int NCB = FZ_NCB;
	constexpr int NR = NCB + 2*R;                             // other-view values a block needs per row (even)
	static_assert(NR % 2 == 0 && WP % 2 == 0 && Smem::RW % 2 == 0 && WS % 2 == 1, "16-byte LDS rows");
	extern __shared__ __align__(16) unsigned char smem_raw[];
	Smem &S = *reinterpret_cast<Smem *>(smem_raw);

	const ViewDev &L = views[ref];
	const ViewDev &Rv = views[oth];
	const int W = L.w, H = L.h, OW = Rv.w, OH = Rv.h;
	const int D = P.num_depth_levels;
	const int tiles_per_row = (W + FZ_TP - 1)/FZ_TP;
	const int trow = blockIdx.x / tiles_per_row;
	const int x0 = (blockIdx.x % tiles_per_row)*FZ_TP;
	const int y = y0 + trow;
	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int p = wave*4 + (lane & 3);                        // pixel within the tile (pixel-fastest inside a wave)
	const int g = lane >> 2;                                  // lane within the pixel
	const int x = x0 + p;
	const int role = 0;                                       // the wave that runs the per-pixel sequential phases
	const bool seq = wave == role && lane < FZ_TP;            // ... with lane = pixel q
	const int q = lane;
	const double nan = __builtin_nan("");

	if (tid == 0) { S.glist_n = 0; S.cmin = 2147483647; S.cmax = -2147483647; S.need_general = 0; S.bad = 0; }
	if (tid < FZ_TP) S.pflag[tid] = 0;
	__syncthreads();

	// ---- A: projections of all labels (every lane of a pixel holds the same ray)
	const bool active = x < W && L.mask[(size_t)y*W + x] == 1;
	{
		Ray ray; double nd = 0; bool hasray = false;
		if (active) {
			ray = cam_unproject(L.cam, (x + 0.5) / P.image_scale, (y + 0.5) / P.image_scale);
			nd = dot(normalized(load3(L.cam.pdir)), ray.dir);
			hasray = !(fabs(nd) < 1e-10);                         // intersect() fails for every label otherwise
		}
		int yflag = 0;
		for (int d = g; d < D; d += FZ_G) {
			double xv = nan;
			if (hasray) {
				double x2, y2;
				if (pinhole_project_label(ray, nd, tnum[d], Rv.cam, P.image_scale, x2, y2)) {
					xv = x2;
					// Row alignment, and exactness of phase B's test: with |y2 - (y+0.5)| < 2^-28 for every label,
					// trunc(y2) == y and (dy*dy < 2^-54), so fl(dx*dx + dy*dy) >= 1  <=>  fl(dx*dx) >= 1.
					if (x2 == x2 && !(fabs(y2 - (y + 0.5)) < 3.7252902984619140625e-9)) yflag = 1;
				}
			}
			S.cost[Smem::slot(p, d)] = xv;
		}
		if (yflag) atomicOr(&S.pflag[p], 1);
	}
	__syncthreads();

	// ---- B: which labels are kept (one lane per pixel); meanwhile the other waves stage windows and reference rows
	if (seq) {
#pragma unroll
		for (int j = 0; j < Smem::JW; ++j) S.joints[q][j] = 0;
		double x1 = nan;
		int K0 = 0, Kprev = 0, sgn = 0, ilast = -1, nmerge = 0, nk = 0, flag = S.pflag[q];
		for (int d = 0; d < D; ++d) {
			const double x2 = S.cost[Smem::slot(q, d)];
			if (!(x2 == x2)) continue;
			if (!(x1 == x1)) { x1 = x2; K0 = Kprev = trunc_sat(x2); nk = 1; continue; }
			const double dx = x2 - x1;
			if (!(dx*dx >= 1)) continue;                          // twoviewstereo.cpp:1027 (dy: see phase A)
			const int s = dx > 0 ? 1 : -1;
			if (sgn && s != sgn) flag |= 2;                       // not monotone: not this kernel's case
			sgn = s;
			const int K = trunc_sat(x2);
			long long i = (long long)(K - K0)*s;                  // index along the curve, >= 0 when monotone
			if (i < 0 || i > MAXC + 30) { flag |= 4; i = 0; }
			else {
				if (nk >= 2 && K == Kprev) ++nmerge;              // two kept points on one column (x2 in (-1,1) truncates to 0 twice)
				S.joints[q][(int)i >> 5] |= 1u << ((int)i & 31);
				if ((int)i > ilast) ilast = (int)i;
			}
			Kprev = K; ++nk; x1 = x2;
		}
		// visited columns: k0 .. k0 + dir*ilast, inside the other image
		int ca = 0, cb = -1, corg = 0;
		if (ilast >= 0) {
			const int e0 = K0, e1 = K0 + sgn*ilast;
			ca = e0 < e1 ? e0 : e1; cb = e0 < e1 ? e1 : e0;
			if (ca < 0) ca = 0;
			if (cb > OW - 1) cb = OW - 1;
			if (y < 0 || y >= OH) cb = ca - 1;                    // (cannot happen: equal-sized views)
			corg = ca & ~1;
			if (cb >= ca && cb - corg + 1 > MAXC) flag |= 4;      // does not fit the cost row
			if (cb >= ca && (flag & 1)) flag |= 8;                // candidates, but not all on row y
		}
		if (flag & ~1) { cb = ca - 1; atomicOr(&S.bad, 1); }
		S.k0[q] = K0; S.dir[q] = sgn; S.ilast[q] = ilast; S.nmerge[q] = nmerge;
		S.ca[q] = ca; S.cb[q] = cb; S.corg[q] = corg;
		if (cb >= ca) {
			const int nb = (cb - corg + NCB)/NCB;
			atomicMin(&S.cmin, corg);
			atomicMax(&S.cmax, corg + nb*NCB - 1);
		}
	} else if (wave != role) {
		// support windows of the tile: wbuf is tile-major per 32 pixels, [tap][32]
		const double *wtile = wbuf + wbuf_offset(W, T, trow, x0);
		const int nth = FZ_THREADS - 64, t3 = tid - 64*(wave > role ? 1 : 0) - (wave < role ? 0 : 0);
		for (int idx = (wave > role ? tid - 64 : tid); idx < T*FZ_TP; idx += nth) {
			const int t = idx / FZ_TP, pi = idx % FZ_TP;
			S.w[pi][(t / WS)*WP + (t % WS)] = (x0 + pi < W) ? wtile[(size_t)t*SRH_WTILE + pi] : 0.0;
		}
		for (int idx = (wave > role ? tid - 64 : tid); idx < WS*Smem::LW; idx += nth) {
			const int ty = idx / Smem::LW, tx = idx % Smem::LW;
			const int gx = x0 - R + tx, gy = y - R + ty;
			S.lt[ty][tx] = (gx >= 0 && gy >= 0 && gx < W && gy < H) ? L.gray_tv[(size_t)gy*W + gx] : nan;
		}
		(void)t3;
	}
	__syncthreads();

	// ---- C: the other view's rows over the union of the tile's visited columns (+R margin), its mask row
	const int cmin = S.cmin, cmax = S.cmax;
	const bool have = cmin <= cmax;
	const bool fits = have && (cmax - cmin + 1 + 2*R <= Smem::RW);
	if (have && !fits && tid == 0) S.bad = 1;
	if (fits) {
		for (int idx = tid; idx < WS*Smem::RW; idx += FZ_THREADS) {
			const int ty = idx / Smem::RW, tx = idx % Smem::RW;
			const int gx = cmin - R + tx, gy = y - R + ty;
			S.rt[ty][tx] = (gx >= 0 && gy >= 0 && gx < OW && gy < OH) ? Rv.gray_tv[(size_t)gy*OW + gx] : nan;
		}
		for (int tx = tid; tx < Smem::RW; tx += FZ_THREADS) {
			const int gx = cmin + tx;
			S.mrow[tx] = (gx >= 0 && gx < OW && y >= 0 && y < OH) ? Rv.mask[(size_t)y*OW + gx] : 0;
		}
	}
	__syncthreads();

	unsigned n_dev = 0;
	if (fits) {
		// ---- D: per-pixel constants of the fast form (one lane per pixel) ...
		if (seq) {
			const int xq = x0 + q;
			bool all = (xq < W) && (S.cb[q] >= S.ca[q]);
			double mL = 0, tw = 0;
#pragma unroll 1
			for (int row = 0; row < WS; ++row)
#pragma unroll
				for (int col = 0; col < WS; ++col) {
					const double gl = S.lt[row][q + col];
					const double wt = S.w[q][row*WP + col];
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
						const double a = S.w[q][row*WP + col]*S.lt[row][q + col] - mL;
						s2 += a*a;
					}
			} else all = false;
			S.meanL[q] = mL; S.totalW[q] = tw; S.sum2[q] = s2; S.lall[q] = all ? 1 : 0;
			if (!all && S.cb[q] >= S.ca[q]) S.need_general = 1;
		} else if (wave != role) {
			// ... and, on the other waves, which candidate columns have a fully usable window
			const int nth = FZ_THREADS - 64, t0 = (wave > role ? tid - 64 : tid);
			for (int tx = t0; tx < Smem::RW; tx += nth) {
				bool ok = true;
#pragma unroll
				for (int ty = 0; ty < WS; ++ty) { const double v = S.rt[ty][tx]; ok = ok && (v == v); }
				S.colok[tx] = ok ? 1 : 0;
			}
		}
		__syncthreads();
		for (int tx = tid; tx < Smem::RW; tx += FZ_THREADS) {
			bool ok = tx + 2*R < Smem::RW;
			if (ok) {
#pragma unroll
				for (int k = 0; k < WS; ++k) ok = ok && S.colok[tx + k] != 0;
			}
			S.rfull[tx] = ok ? 1 : 0;
			const int c = cmin + tx;
			if (!ok && c <= cmax) S.need_general = 1;
		}
		__syncthreads();

		// ---- E: costs.  Phase 1: blocks of NCB adjacent columns in the fast form.
		const Smem &CS = S;
		const int ca = CS.ca[p], cb = CS.cb[p], corg = CS.corg[p];
		if (cb >= ca && CS.lall[p]) {
			const int nblocks = (cb - corg + NCB)/NCB;
			const double mL = CS.meanL[p], tw = CS.totalW[p], s2 = CS.sum2[p];
			for (int b = g; b < nblocks; b += FZ_G) {
				const int c0 = corg + b*NCB;
				const int rc = c0 - cmin;                         // tile column of the window's left edge (even)
				bool fast = false;
#pragma unroll
				for (int j = 0; j < NCB; ++j) {
					const int c = c0 + j;
					if (c >= ca && c <= cb && CS.rfull[rc + j] != 0) { fast = true; ++n_dev; }
				}
				if (!fast) continue;
				// Both passes are modulo-scheduled by hand (see srh_dense.hip): r[] / wv[] hold the current window
				// row; a register is refilled with the next row's value right after its last use.
				double r[NR], wv[WS], acc[NCB];
				{
					const double2 *rp = reinterpret_cast<const double2 *>(&CS.rt[0][rc]);
					const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[p][0]);
#pragma unroll
					for (int m = 0; m < NR/2; ++m) { const double2 v = rp[m]; r[2*m] = v.x; r[2*m + 1] = v.y; }
#pragma unroll
					for (int m = 0; m < (WS - 1)/2; ++m) { const double2 v = wp[m]; wv[2*m] = v.x; wv[2*m + 1] = v.y; }
					wv[WS - 1] = CS.w[p][WS - 1];
				}
#pragma unroll
				for (int j = 0; j < NCB; ++j) acc[j] = 0.0;
				__builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 1
				for (int row = 0; row < WS; ++row) {
					const int nrow = row + 1 < WS ? row + 1 : 0;          // last refill = row 0, for pass 2
					const double2 *rp = reinterpret_cast<const double2 *>(&CS.rt[nrow][rc]);
					const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[p][nrow*WP]);
#pragma unroll
					for (int col = 0; col < WS; ++col) {
						double pr[NCB];
#pragma unroll
						for (int j = 0; j < NCB; ++j) pr[j] = wv[col]*r[col + j];
						__builtin_amdgcn_sched_barrier(0);
#pragma unroll
						for (int j = 0; j < NCB; ++j) acc[j] += pr[j];            // meanR += weight*gray
						__builtin_amdgcn_sched_barrier(0);
						if (col & 1) {
							const double2 v = rp[col >> 1]; r[col - 1] = v.x; r[col] = v.y;
							const double2 u = wp[col >> 1]; wv[col - 1] = u.x; wv[col] = u.y;
							__builtin_amdgcn_sched_barrier(0);
						}
					}
#pragma unroll
					for (int m = (WS - 1)/2; m < NR/2; ++m) { const double2 v = rp[m]; r[2*m] = v.x; r[2*m + 1] = v.y; }
					wv[WS - 1] = CS.w[p][nrow*WP + WS - 1];
				}
				double mR[NCB], s1[NCB], s3[NCB], av[WS];
#pragma unroll
				for (int col = 0; col < WS; ++col) av[col] = CS.lt[0][p + col];
#pragma unroll
				for (int j = 0; j < NCB; ++j) { mR[j] = acc[j]/tw; s1[j] = 0.0; s3[j] = 0.0; }
				__builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 1
				for (int row = 0; row < WS; ++row) {
					const int nrow = row + 1 < WS ? row + 1 : 0;
					const double2 *rp = reinterpret_cast<const double2 *>(&CS.rt[nrow][rc]);
					const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[p][nrow*WP]);
					const double *lp = &CS.lt[nrow][p];
#pragma unroll
					for (int col = 0; col < WS; ++col) {
						const double wt = wv[col];
						double bb[NCB], u1[NCB], u3[NCB];
						const double pa = wt*av[col];
#pragma unroll
						for (int j = 0; j < NCB; ++j) bb[j] = wt*r[col + j];
						__builtin_amdgcn_sched_barrier(0);
						const double a = pa - mL;                         // pixel_gray_l - meanL
#pragma unroll
						for (int j = 0; j < NCB; ++j) bb[j] = bb[j] - mR[j];   // pixel_gray_r - meanR
						__builtin_amdgcn_sched_barrier(0);
#pragma unroll
						for (int j = 0; j < NCB; ++j) { u1[j] = a*bb[j]; u3[j] = bb[j]*bb[j]; }
						__builtin_amdgcn_sched_barrier(0);
#pragma unroll
						for (int j = 0; j < NCB; ++j) { s1[j] += u1[j]; s3[j] += u3[j]; }
						__builtin_amdgcn_sched_barrier(0);
						av[col] = lp[col];
						if (col & 1) {
							const double2 v = rp[col >> 1]; r[col - 1] = v.x; r[col] = v.y;
							const double2 u = wp[col >> 1]; wv[col - 1] = u.x; wv[col] = u.y;
						}
						__builtin_amdgcn_sched_barrier(0);
					}
#pragma unroll
					for (int m = (WS - 1)/2; m < NR/2; ++m) { const double2 v = rp[m]; r[2*m] = v.x; r[2*m + 1] = v.y; }
					wv[WS - 1] = CS.w[p][nrow*WP + WS - 1];
				}
#pragma unroll
				for (int j = 0; j < NCB; ++j) {
					const int c = c0 + j;
					if (c >= ca && c <= cb && CS.rfull[rc + j] != 0) {
						const double v = 255*(1.0 - fabs(s1[j]) / sqrt(s2 * s3[j]));
						S.cost[Smem::slot(p, c - corg)] = (v < P.max_color_diff) ? v : P.max_color_diff;
					}
					__builtin_amdgcn_sched_barrier(0);
				}
			}
		}
		// Phase 2: the remaining candidates (a tap unusable on either side) in the select form, compacted into an
		// LDS work list and spread over all lanes.  Columns whose mask byte is not WHITE are never candidates.
		const bool need_general = S.need_general != 0;            // uniform: written before the last barrier
		for (int base = 0; need_general && base < FZ_TP*Smem::RW; base += Smem::GL_CAP) {
			__syncthreads();
			if (tid == 0) S.glist_n = 0;
			__syncthreads();
			for (int pr = base + tid; pr < base + Smem::GL_CAP && pr < FZ_TP*Smem::RW; pr += FZ_THREADS) {
				const int pi = pr / Smem::RW, k = pr % Smem::RW;
				const int c = cmin + k;
				if (c < S.ca[pi] || c > S.cb[pi]) continue;
				if (CS.lall[pi] && CS.rfull[k]) continue;             // done in phase 1
				if (CS.mrow[k] != 1) continue;
				S.glist[atomicAdd(&S.glist_n, 1)] = (unsigned short)(pi*512 + k);
			}
			__syncthreads();
			const int nl = S.glist_n;
			for (int e = tid; e < nl; e += FZ_THREADS) {
				const int pi = S.glist[e] >> 9, k = S.glist[e] & 511;
				++n_dev;
				S.cost[Smem::slot(pi, cmin + k - S.corg[pi])] =
					fused_cost_general<R, MAXC>(CS, pi, k, P.weight_cutoff, P.bad_ret, P.max_color_diff);
			}
		}
	}
	__syncthreads();

	// ---- F: winner-take-all in the reference's visiting order (one lane per pixel), depth of the winner
	unsigned n_eval = 0, n_pix = 0;
	if (seq) {
		const int xq = x0 + q;
		const bool act = xq < W && L.mask[(size_t)y*W + xq] == 1;
		double depth = nan;                                       // twoviewstereo.cpp:269
		if (act) {
			n_pix = 1;
			const int ca = S.ca[q], cb = S.cb[q], corg = S.corg[q];
			double minCost = __builtin_inf(), secondBest = __builtin_inf();
			int wcol = -2147483647;
			if (fits && cb >= ca) {
				const int K0 = S.k0[q], dir = S.dir[q], ilast = S.ilast[q];
				auto visit = [&](int c) {
					if (c < ca || c > cb) return;                     // outside the other image
					if (S.mrow[c - cmin] != 1) return;                // mask.pixel(tx,ty) == WHITE, twoviewstereo.cpp:1034
					++n_eval;
					const double cv = S.cost[Smem::slot(q, c - corg)];
					if (cv + P.wta_margin < minCost) {                // twoviewstereo.cpp:293-301
						secondBest = minCost; minCost = cv; wcol = c;
					}
				};
				if (dir > 0) {
					for (int i = 0; i <= ilast; ++i) visit(K0 + i);
				} else {
					// segments in label order run towards smaller columns, each visited in ascending column order
					// (lineiter.hpp:96-111); its upper end was visited by the segment before
					int lo = 0;
					while (true) {
						const int nj = next_joint<Smem::JW>(S.joints[q], lo);
						if (nj < 0) break;
						for (int i = nj; i >= lo; --i) visit(K0 - i);
						lo = nj + 1;
					}
				}
				// joints are evaluated twice by the reference (once per segment): count them as it does
				{
					int dup = 0, lo = 0;
					while (true) {
						const int nj = next_joint<Smem::JW>(S.joints[q], lo);
						if (nj < 0) break;
						const int c = K0 + dir*nj;
						if (nj != ilast && c >= ca && c <= cb && S.mrow[c - cmin] == 1) ++dup;
						lo = nj + 1;
					}
					if (S.nmerge[q] > 0 && 0 >= ca && 0 <= cb && S.mrow[0 - cmin] == 1) dup += S.nmerge[q];
					n_eval += dup;
				}
			}
			if (wcol != -2147483647) {
				const Ray ray = cam_unproject(L.cam, (xq + 0.5) / P.image_scale, (y + 0.5) / P.image_scale);
				depth = candidate_depth(L.cam, Rv.cam, P, ray, wcol, y);
			}
			if (minCost > P.second_best_factor*secondBest) depth = __builtin_inf();   // twoviewstereo.cpp:304-305
		}
		if (xq < W) L.depth[(size_t)y*W + xq] = depth;
	}
	block_count_add(&cnt->n_eval, n_eval);
	block_count_add(&cnt->n_eval_device, n_dev);
	block_count_add(&cnt->n_pixels, n_pix);
	if (tid == 0 && S.bad) atomicAdd(&cnt->not_row_aligned, 1ull);
}

bool launch_twoview_fused(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                          int y0, int nrows, const double *wbuf, const double *tnum, Counters *cnt)
{
	const int tiles = (width + FZ_TP - 1)/FZ_TP;
	const dim3 grid((unsigned)(tiles*nrows));
#define SRH_FZ_LAUNCH(RR, MC)                                                                                \
	{                                                                                                        \
		typedef FusedSmem<RR, MC> Smem;                                                                      \
		/* per device, hence on every launch */                                                              \
		(void)hipFuncSetAttribute((const void *)twoview_fused_kernel<RR, MC>,                                \
		                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Smem));           \
		hipLaunchKernelGGL((twoview_fused_kernel<RR, MC>), grid, dim3(FZ_THREADS), sizeof(Smem), st,         \
		                   views, ref, oth, P, y0, nrows, wbuf, tnum, cnt);                                  \
		return true;                                                                                         \
	}
	if (P.num_depth_levels > SRH_FUSED_MAXC) return false;        // the label projections share the cost rows' LDS
	switch (P.window_radius) {
	case 5: SRH_FZ_LAUNCH(5, SRH_FUSED_MAXC)
	case 2: SRH_FZ_LAUNCH(2, SRH_FUSED_MAXC)
	default: return false;
	}
#undef SRH_FZ_LAUNCH
}

} // namespace srh
