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

#include <cstdio>

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
	int pflag[FZ_TP], firstd[FZ_TP];                        // row/monotony flags, first label with a projection
	unsigned short kept[FZ_TP][MAXC/16];                    // phase B: bit k of kept[p][b]: label 16b+k is a kept point
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
	constexpr int JW = Smem::JW;
	static_assert(NR % 2 == 0 && WP % 2 == 0 && Smem::RW % 2 == 0 && WS % 2 == 1, "16-byte LDS rows");
	static_assert(MAXC == FZ_G*16, "phase F: every lane of a pixel owns 16 positions of the visiting order");
	extern __shared__ __align__(16) unsigned char smem_raw[];
	Smem &S = *reinterpret_cast<Smem *>(smem_raw);

	const int W = views[ref].w, H = views[ref].h, OW = views[oth].w, OH = views[oth].h;
	const int D = P.num_depth_levels;
	const int tiles_per_row = (W + FZ_TP - 1)/FZ_TP;
	const double nan = __builtin_nan("");
	const double inf = __builtin_inf();

#ifdef SRH_PROFILE_PHASES
	unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	unsigned long long tp = __builtin_readcyclecounter();
#define FZ_STAMP(k) do { const unsigned long long tn_ = __builtin_readcyclecounter(); ph[k] += tn_ - tp; tp = tn_; } while (0)
#else
#define FZ_STAMP(k) do { } while (0)
#endif
	// Everything but the block loops of phase E is latency-bound bookkeeping with few busy lanes.  A workgroup
	// shares its SIMDs with another one that is usually deep in its FP64 loops; the hardware favours the older
	// wave, which would starve these dependent chains (measured: 560 cycles per label in phase B).  They run at
	// raised priority -- they leave almost every issue slot free anyway.
	// One workgroup per tile (a persistent grid of two workgroups per CU, with or without a start offset between
	// the two, measured 8 % slower on MI355X).
	unsigned long long n_eval = 0, n_pix = 0, n_dev = 0;
	{
	const int tile = blockIdx.x;
	const ViewDev *vt = views;
	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int p = wave*4 + (lane & 3);                        // pixel within the tile (pixel-fastest inside a wave)
	const int g = lane >> 2;                                  // lane within the pixel
	// the two per-pixel sequential phases run side by side on two waves, one lane per pixel
	const bool seqB = wave == 0 && lane < FZ_TP;              // B: which labels are kept
	const bool seqD = wave == 1 && lane < FZ_TP;              // D: constants of the fast cost form
	const int q = lane;
	const ViewDev &L = vt[ref];
	const ViewDev &Rv = vt[oth];
	const int trow = tile / tiles_per_row;
	const int x0 = (tile % tiles_per_row)*FZ_TP;
	const int y = y0 + trow;
	const int x = x0 + p;
	__builtin_amdgcn_s_setprio(3);
	if (tid == 0) { S.glist_n = 0; S.cmin = 2147483647; S.cmax = -2147483647; S.need_general = 0; S.bad = 0; }
	if (tid < FZ_TP) { S.pflag[tid] = 0; S.firstd[tid] = 2147483647; S.ilast[tid] = -1; S.nmerge[tid] = 0; }
	for (int j = tid; j < FZ_TP*JW; j += FZ_THREADS) S.joints[j / JW][j % JW] = 0;

	// ---- support windows and reference rows of the tile (wbuf is tile-major per 32 pixels, [tap][32]).
	// Every global load of the thread is issued before the first LDS store: one memory latency for the lot.
	{
		const double *wtile = wbuf + wbuf_offset(W, T, trow, x0);
		constexpr int NBW = (T*FZ_TP + FZ_THREADS - 1)/FZ_THREADS, NBL = (WS*Smem::LW + FZ_THREADS - 1)/FZ_THREADS;
		double tw_[NBW], tl_[NBL];
#pragma unroll
		for (int k = 0; k < NBW; ++k) {
			const int idx = tid + k*FZ_THREADS;
			const int t = idx / FZ_TP, pi = idx % FZ_TP;
			tw_[k] = (idx < T*FZ_TP && x0 + pi < W) ? wtile[(size_t)t*SRH_WTILE + pi] : 0.0;
		}
#pragma unroll
		for (int k = 0; k < NBL; ++k) {
			const int idx = tid + k*FZ_THREADS;
			const int ty = idx / Smem::LW, tx = idx % Smem::LW;
			const int gx = x0 - R + tx, gy = y - R + ty;
			tl_[k] = (idx < WS*Smem::LW && gx >= 0 && gy >= 0 && gx < W && gy < H) ? L.gray_tv[(size_t)gy*W + gx] : nan;
		}
#pragma unroll
		for (int k = 0; k < NBW; ++k) {
			const int idx = tid + k*FZ_THREADS;
			const int t = idx / FZ_TP, pi = idx % FZ_TP;
			if (idx < T*FZ_TP) S.w[pi][(t / WS)*WP + (t % WS)] = tw_[k];
		}
#pragma unroll
		for (int k = 0; k < NBL; ++k) {
			const int idx = tid + k*FZ_THREADS;
			if (idx < WS*Smem::LW) S.lt[idx / Smem::LW][idx % Smem::LW] = tl_[k];
		}
	}
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
		int yflag = 0, fd = 2147483647;
		double tcur = g < D ? tnum[g] : 0.0;
		for (int d = g; d < D; d += FZ_G) {
			const double tnext = d + FZ_G < D ? tnum[d + FZ_G] : 0.0;   // one label ahead: the load is never waited for
			double xv = nan;
			if (hasray) {
				double x2, y2;
				if (pinhole_project_label(ray, nd, tcur, Rv.cam, P.image_scale, x2, y2)) {
					xv = x2;
					// Row alignment, and exactness of phase B's test: with |y2 - (y+0.5)| < 2^-28 for every label,
					// trunc(y2) == y and (dy*dy < 2^-54), so fl(dx*dx + dy*dy) >= 1  <=>  fl(dx*dx) >= 1.
					if (x2 == x2 && !(fabs(y2 - (y + 0.5)) < 3.7252902984619140625e-9)) yflag = 1;
				}
			}
			S.cost[Smem::slot(p, d)] = xv;
			if (xv == xv && fd == 2147483647) fd = d;
			tcur = tnext;
		}
		if (yflag) atomicOr(&S.pflag[p], 1);
		if (fd != 2147483647) atomicMin(&S.firstd[p], fd);
	}
	FZ_STAMP(0);
	__syncthreads();
	FZ_STAMP(7);

	// ---- B1 (wave 0): which labels are kept (twoviewstereo.cpp:1018-1052), one lane per pixel.  Only the decision
	// chain is sequential: x1 = first projected label; a label is kept when |x2 - x1| >= 1 (== fl(dx*dx) >= 1, and
	// dy does not matter, see phase A), then x1 = x2.  A NaN x2 (label without a projection) is never kept.  The x2
	// values come from LDS 16 at a time, one batch ahead; the chain itself runs on registers.
	if (seqB) {
		const int fd = S.firstd[q];
		double x1 = fd != 2147483647 ? S.cost[Smem::slot(q, fd)] : nan;
		double xa[16], xb[16];
#pragma unroll
		for (int k = 0; k < 16; ++k) xa[k] = k < D ? S.cost[Smem::slot(q, k)] : nan;
		for (int d0 = 0; d0 < D; d0 += 32) {
#pragma unroll
			for (int k = 0; k < 16; ++k) xb[k] = d0 + 16 + k < D ? S.cost[Smem::slot(q, d0 + 16 + k)] : nan;
			unsigned bits = 0;
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				const bool acc = fabs(xa[k] - x1) >= 1.0;
				x1 = acc ? xa[k] : x1;
				bits |= acc ? (1u << k) : 0u;
			}
			S.kept[q][d0 >> 4] = (unsigned short)bits;
#pragma unroll
			for (int k = 0; k < 16; ++k) xa[k] = d0 + 32 + k < D ? S.cost[Smem::slot(q, d0 + 32 + k)] : nan;
			bits = 0;
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				const bool acc = fabs(xb[k] - x1) >= 1.0;
				x1 = acc ? xb[k] : x1;
				bits |= acc ? (1u << k) : 0u;
			}
			if (d0 + 16 < MAXC) S.kept[q][(d0 >> 4) + 1] = (unsigned short)bits;
		}
		for (int b2 = (D + 31)/32*2; b2 < MAXC/16; ++b2) S.kept[q][b2] = 0;
	}
	// ---- D (wave 1, meanwhile): per-pixel constants of the fast form -- meanL, totalWeight, sum2 do not depend on
	// the candidate when every tap of the window is usable on both sides (same tap order as twoviewstereo.cpp:917-976)
	if (seqD) {
		bool all = (x0 + q < W);
		double mL = 0, tw = 0;
#pragma unroll 1
		for (int row = 0; row < WS; ++row) {
			double gl[WS], wt[WS];
#pragma unroll
			for (int col = 0; col < WS; ++col) { gl[col] = S.lt[row][q + col]; wt[col] = S.w[q][row*WP + col]; }
#pragma unroll
			for (int col = 0; col < WS; ++col) {
				if (!(gl[col] == gl[col] && wt[col] > P.weight_cutoff)) all = false;
				mL += wt[col]*gl[col];
				tw += wt[col];
			}
		}
		double s2 = 0;
		if (all && !(tw < 1e-10)) {
			mL /= tw;
#pragma unroll 1
			for (int row = 0; row < WS; ++row) {
				double gl[WS], wt[WS];
#pragma unroll
				for (int col = 0; col < WS; ++col) { gl[col] = S.lt[row][q + col]; wt[col] = S.w[q][row*WP + col]; }
#pragma unroll
				for (int col = 0; col < WS; ++col) { const double a = wt[col]*gl[col] - mL; s2 += a*a; }
			}
		} else all = false;
		S.meanL[q] = mL; S.totalW[q] = tw; S.sum2[q] = s2; S.lall[q] = all ? 1 : 0;
	}
	FZ_STAMP(1);
	__syncthreads();
	FZ_STAMP(7);

	// ---- B2 (all lanes, lane g of a pixel takes labels 16g..16g+15): from the kept flags, the geometry of the
	// curve -- first kept column K0, direction, index i = (K - K0)*dir of every later kept point (a joint bit),
	// monotony, kept points that truncate to the same column.
	{
		const int fd = S.firstd[p];
		unsigned km[MAXC/16];
#pragma unroll
		for (int b2 = 0; b2 < MAXC/16; ++b2) km[b2] = S.kept[p][b2];
		int firstk = -1, prevk = -1;                              // first kept label after fd; last kept label before my chunk
#pragma unroll
		for (int b2 = MAXC/16 - 1; b2 >= 0; --b2) if (km[b2]) firstk = 16*b2 + __ffs((int)km[b2]) - 1;
#pragma unroll
		for (int b2 = 0; b2 < MAXC/16; ++b2) if (b2 < g && km[b2]) prevk = 16*b2 + 31 - __clz((int)km[b2]);
		unsigned mine = 0;
#pragma unroll
		for (int b2 = 0; b2 < MAXC/16; ++b2) if (b2 == g) mine = km[b2];
		if (firstk >= 0) {
			const double xf = S.cost[Smem::slot(p, fd)];
			const int K0 = trunc_sat(xf);
			const int sgn = S.cost[Smem::slot(p, firstk)] > xf ? 1 : -1;
			if (g == 0) { S.k0[p] = K0; S.dir[p] = sgn; }
			int il = -1, nm = 0, flag = 0;
			// my 16 projections and the predecessor of my first kept label, fetched at once; the loop runs on registers
			double xs[16];
#pragma unroll
			for (int k = 0; k < 16; ++k) xs[k] = S.cost[Smem::slot(p, 16*g + k)];
			double xp = S.cost[Smem::slot(p, prevk >= 0 ? prevk : fd)];
			bool pfirst = prevk < 0;                              // the predecessor is the curve's first point
			int curw = -1; unsigned curbits = 0;
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				if ((mine >> k) & 1u) {
					const double x2 = xs[k];
					if ((x2 > xp ? 1 : -1) != sgn) flag |= 2;     // not monotone: not this kernel's case
					const int K = trunc_sat(x2);
					const long long i = (long long)(K - K0)*sgn;  // index along the curve
					if (i < 0 || i >= MAXC) flag |= 4;
					else {
						if (!pfirst && K == trunc_sat(xp)) ++nm;  // two kept points on one column (x2 in (-1,1) truncates to 0 twice)
						const int wi = (int)i >> 5;
						if (wi != curw) {                         // indices never decrease along a monotone curve
							if (curw >= 0) atomicOr(&S.joints[p][curw], curbits);
							curw = wi; curbits = 0;
						}
						curbits |= 1u << ((int)i & 31);
						if ((int)i > il) il = (int)i;
					}
					xp = x2; pfirst = false;
				}
			}
			if (curw >= 0) atomicOr(&S.joints[p][curw], curbits);
			if (il >= 0) atomicMax(&S.ilast[p], il);
			if (nm) atomicAdd(&S.nmerge[p], nm);
			if (flag) atomicOr(&S.pflag[p], flag);
		} else if (g == 0) { S.k0[p] = 0; S.dir[p] = 0; }
	}
	__syncthreads();
	// ---- B3: visited columns k0 .. k0 + dir*ilast inside the other image, cost-row origin, the tile's union
	if (tid < FZ_TP) {
		const int K0 = S.k0[tid], sgn = S.dir[tid], ilast = S.ilast[tid];
		int flag = S.pflag[tid];
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
		if (flag & ~1) { cb = ca - 1; S.ilast[tid] = -1; atomicOr(&S.bad, 1); }
		S.ca[tid] = ca; S.cb[tid] = cb; S.corg[tid] = corg;
		if (cb >= ca) {
			const int nb = (cb - corg + NCB)/NCB;
			atomicMin(&S.cmin, corg);
			atomicMax(&S.cmax, corg + nb*NCB - 1);
		}
	}
	__syncthreads();

	// ---- C: the other view's rows over the union of the tile's visited columns (+R margin), its mask row
	const int cmin = S.cmin, cmax = S.cmax;
	const bool have = cmin <= cmax;
	const bool fits = have && (cmax - cmin + 1 + 2*R <= Smem::RW);
	if (have && !fits && tid == 0) S.bad = 1;
	if (fits) {
		constexpr int NBR = (WS*Smem::RW + FZ_THREADS - 1)/FZ_THREADS, NBM = (Smem::RW + FZ_THREADS - 1)/FZ_THREADS;
		double tr_[NBR]; unsigned char tm_[NBM];
#pragma unroll
		for (int k = 0; k < NBR; ++k) {
			const int idx = tid + k*FZ_THREADS;
			const int ty = idx / Smem::RW, tx = idx % Smem::RW;
			const int gx = cmin - R + tx, gy = y - R + ty;
			tr_[k] = (idx < WS*Smem::RW && gx >= 0 && gy >= 0 && gx < OW && gy < OH) ? Rv.gray_tv[(size_t)gy*OW + gx] : nan;
		}
#pragma unroll
		for (int k = 0; k < NBM; ++k) {
			const int tx = tid + k*FZ_THREADS;
			const int gx = cmin + tx;
			tm_[k] = (tx < Smem::RW && gx >= 0 && gx < OW && y >= 0 && y < OH) ? Rv.mask[(size_t)y*OW + gx] : 0;
		}
#pragma unroll
		for (int k = 0; k < NBR; ++k) {
			const int idx = tid + k*FZ_THREADS;
			if (idx < WS*Smem::RW) S.rt[idx / Smem::RW][idx % Smem::RW] = tr_[k];
		}
#pragma unroll
		for (int k = 0; k < NBM; ++k) {
			const int tx = tid + k*FZ_THREADS;
			if (tx < Smem::RW) S.mrow[tx] = tm_[k];
		}
		if (tid < FZ_TP && !S.lall[tid] && S.cb[tid] >= S.ca[tid]) S.need_general = 1;   // a pixel with an unusable tap of its own
	}
	FZ_STAMP(2);
	__syncthreads();
	FZ_STAMP(7);

	if (fits) {
		// which candidate columns have a fully usable window
		for (int tx = tid; tx < Smem::RW; tx += FZ_THREADS) {
			bool ok = true;
#pragma unroll
			for (int ty = 0; ty < WS; ++ty) { const double v = S.rt[ty][tx]; ok = ok && (v == v); }
			S.colok[tx] = ok ? 1 : 0;
		}
		FZ_STAMP(3);
		__syncthreads();
		FZ_STAMP(7);
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
		__builtin_amdgcn_s_setprio(0);

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
		FZ_STAMP(4);
		// Phase 2: the remaining candidates (a tap unusable on either side) in the select form, compacted into an
		// LDS work list and spread over all lanes.  Columns whose mask byte is not WHITE are never candidates.
		const bool need_general = S.need_general != 0;            // uniform: written before the last barrier
#ifdef SRH_PROFILE_PHASES
		if (tid == 0 && need_general) atomicAdd(&cnt->n_listed, 1ull);
#endif
		// One pass over all (pixel, column) pairs when the list holds them (the usual case: a few border columns or a
		// pixel or two with an unusable tap); rounds of GL_CAP pairs otherwise (rows at the top / bottom of the image).
		bool one_pass = true;
		for (int base = 0; need_general && base < FZ_TP*Smem::RW; ) {
			__syncthreads();
			if (tid == 0) S.glist_n = 0;
			__syncthreads();
			const int end = one_pass ? FZ_TP*Smem::RW : (base + Smem::GL_CAP < FZ_TP*Smem::RW ? base + Smem::GL_CAP : FZ_TP*Smem::RW);
			for (int pr = base + tid; pr < end; pr += FZ_THREADS) {
				const int pi = pr / Smem::RW, k = pr % Smem::RW;
				const int c = cmin + k;
				if (c < S.ca[pi] || c > S.cb[pi]) continue;
				if (CS.lall[pi] && CS.rfull[k]) continue;             // done in phase 1
				if (CS.mrow[k] != 1) continue;
				const int slot_ = atomicAdd(&S.glist_n, 1);
				if (slot_ < Smem::GL_CAP) S.glist[slot_] = (unsigned short)(pi*512 + k);
			}
			__syncthreads();
			const int nl = S.glist_n;
			if (nl > Smem::GL_CAP) { one_pass = false; continue; }    // uniform: does not fit, take it in rounds
			for (int e = tid; e < nl; e += FZ_THREADS) {
				const int pi = S.glist[e] >> 9, k = S.glist[e] & 511;
				++n_dev;
				S.cost[Smem::slot(pi, cmin + k - S.corg[pi])] =
					fused_cost_general<R, MAXC>(CS, pi, k, P.weight_cutoff, P.bad_ret, P.max_color_diff);
			}
			base = end;
		}
	}
#ifdef SRH_PROFILE_PHASES
	if (fits) { unsigned long long v_ = 0; for (int pi = tid; pi < FZ_TP; pi += FZ_THREADS) v_ += (S.lall[pi] == 0 && S.cb[pi] >= S.ca[pi]) ? 1 : 0;
	            if (v_) atomicAdd(&cnt->dbg_phase[7], 0ull); if (v_) atomicAdd(&cnt->n_slots, v_); }
#endif
	__builtin_amdgcn_s_setprio(3);
	FZ_STAMP(5);
	__syncthreads();
	FZ_STAMP(7);

	// ---- F: winner-take-all in the reference's visiting order, all 16 lanes of a pixel.
	// Visiting order (twoviewstereo.cpp:1018-1040, lineiter.hpp:96-111): segments in label order, every segment in
	// ascending column order; a column seen before cannot change the running minimum again (its cost is not
	// < minCost - margin), so only first visits count.  With i = index along the curve (column k0 + dir*i) and the
	// joint bits at the kept points, position v of the first-visit sequence is index v itself when the curve runs
	// towards larger columns; when it runs towards smaller columns the indices between two joints (s..n, n a joint)
	// are visited in reverse: index s + n - v.  Lane t takes positions 16t..16t+15, so lane order = visiting order.
	// The running minimum with its margin (cost + 1e-10 < minCost) equals the plain running minimum unless a new
	// minimum improves by less than the margin; lanes detect that case and the pixel is then replayed sequentially.
	const int t = g;
	const bool pix = fits && active && S.cb[p] >= S.ca[p];
	double c16[16];
	int col16[16];
	{
		const int ilast = S.ilast[p], K0 = S.k0[p], dirp = S.dir[p], ca = S.ca[p], cb = S.cb[p], corg = S.corg[p];
		unsigned jw[JW];
#pragma unroll
		for (int j = 0; j < JW; ++j) jw[j] = S.joints[p][j];
		const int a = 16*t;
		unsigned mine = 0; int below = -1, above = -1;
#pragma unroll
		for (int j = 0; j < JW; ++j) {
			if (j == (a >> 5)) mine = (jw[j] >> (a & 31)) & 0xffffu;
			const unsigned mb = (32*j + 32 <= a) ? 0xffffffffu : (32*j >= a ? 0u : ((1u << (a - 32*j)) - 1u));
			const unsigned m1 = jw[j] & mb;
			if (m1) below = 32*j + 31 - __clz((int)m1);           // ascending j: the last hit is the highest joint < a
		}
#pragma unroll
		for (int j = JW - 1; j >= 0; --j) {
			const int a2 = a + 16;
			const unsigned ma = (32*j >= a2) ? 0xffffffffu : (32*j + 32 <= a2 ? 0u : ~((1u << (a2 - 32*j)) - 1u));
			const unsigned m2 = jw[j] & ma;
			if (m2) above = 32*j + __ffs((int)m2) - 1;            // descending j: the last hit is the lowest joint >= a+16
		}
		unsigned char mk[16];
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			const int v = a + k;
			int i = v;
			bool ok = pix && v <= ilast;
			if (dirp < 0) {
				const unsigned lo = mine & ((1u << k) - 1u), hi = mine >> k;
				const int prevj = lo ? a + 31 - __clz((int)lo) : below;
				const int n = hi ? v + __ffs((int)hi) - 1 : above;
				ok = ok && n >= 0;
				i = prevj + 1 + n - v;
			}
			const int c = K0 + dirp*i;
			ok = ok && c >= ca && c <= cb;
			col16[k] = ok ? c : -2147483647;
			const int cc = ok ? c : ca;                           // a safe address for the unconditional loads
			mk[k] = (pix ? S.mrow[cc - cmin] : 0);
			c16[k] = pix ? S.cost[Smem::slot(p, cc - corg)] : inf;
		}
		unsigned dup = 0;
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			const bool ok = col16[k] != -2147483647 && mk[k] == 1;    // mask.pixel(tx,ty) == WHITE, twoviewstereo.cpp:1034
			if (!ok) { c16[k] = inf; col16[k] = -2147483647; }
			n_eval += ok ? 1u : 0u;
			// joints are evaluated twice by the reference (once per segment): count them as it does
			if (pix && ((mine >> k) & 1u) && a + k != ilast) {
				const int cj = K0 + dirp*(a + k);
				if (cj >= ca && cj <= cb && S.mrow[cj - cmin] == 1) ++dup;
			}
		}
		if (pix && t == 0 && S.nmerge[p] > 0 && 0 >= ca && 0 <= cb && S.mrow[0 - cmin] == 1) dup += S.nmerge[p];
		n_eval += dup;
	}
	// running minimum before my first position: exclusive prefix-min over the lanes of the pixel (lane = 4g + pixel,
	// so "one lane of the pixel earlier" is 4 wave lanes down); the pixel's 16 lanes are in one wave: no barrier
	double lm = inf;
#pragma unroll
	for (int k = 0; k < 16; ++k) lm = c16[k] < lm ? c16[k] : lm;
	double scan = lm;
#pragma unroll
	for (int off = 4; off < 64; off <<= 1) {
		const double o = __shfl_up(scan, off, 64);
		if (lane >= off && o < scan) scan = o;
	}
	double Pin = __shfl_up(scan, 4, 64);
	if (t == 0) Pin = inf;
	double Pm = Pin, sec = inf; int wc = -2147483647, amb = 0;
#pragma unroll
	for (int k = 0; k < 16; ++k) {
		if (c16[k] < Pm) {
			if (!(c16[k] + P.wta_margin < Pm)) amb = 1;           // the reference would not take this one
			sec = Pm; Pm = c16[k]; wc = col16[k];
		}
	}
	if (!(P.wta_margin >= 0)) amb = 1;
	const double gmin = __shfl(scan, 60 + (lane & 3), 64);        // the pixel's last lane holds the overall minimum
	const unsigned long long ambs = __ballot(amb) & (0x1111111111111111ull << (lane & 3));
	// the lane that holds the last record (records decrease strictly, so it is the one whose record is the minimum)
	// finishes the pixel; without any candidate lane 0 does
	const bool owner = wc != -2147483647 && Pm == gmin;
	if (owner || (t == 0 && !(gmin < inf))) {
		double depth = nan;                                       // twoviewstereo.cpp:269
		if (active) {
			double minCost = owner ? Pm : inf, secondBest = owner ? sec : inf;
			int wcol = wc;
			if (pix && ambs) {
				// replay with the reference's margin rule, sequentially (rare: two costs within 1e-10 of each other)
				const int ilast = S.ilast[p], K0 = S.k0[p], dirp = S.dir[p], ca = S.ca[p], cb = S.cb[p], corg = S.corg[p];
				minCost = inf; secondBest = inf; wcol = -2147483647;
				auto visit = [&](int c) {
					if (c < ca || c > cb) return;
					if (S.mrow[c - cmin] != 1) return;
					const double cv = S.cost[Smem::slot(p, c - corg)];
					if (cv + P.wta_margin < minCost) { secondBest = minCost; minCost = cv; wcol = c; }   // twoviewstereo.cpp:293-301
				};
				if (dirp > 0) {
					for (int i = 0; i <= ilast; ++i) visit(K0 + i);
				} else {
					int lo = 0;
					while (true) {
						const int nj = next_joint<JW>(S.joints[p], lo);
						if (nj < 0) break;
						for (int i = nj; i >= lo; --i) visit(K0 - i);
						lo = nj + 1;
					}
				}
			}
			if (wcol != -2147483647) {
				const Ray ray = cam_unproject(L.cam, (x + 0.5) / P.image_scale, (y + 0.5) / P.image_scale);
				depth = candidate_depth(L.cam, Rv.cam, P, ray, wcol, y);
			}
			if (minCost > P.second_best_factor*secondBest) depth = inf;   // twoviewstereo.cpp:304-305
		}
		if (x < W) L.depth[(size_t)y*W + x] = depth;
	}
	if (t == 0 && active) ++n_pix;
	if (tid == 0 && S.bad) atomicAdd(&cnt->not_row_aligned, 1ull);
	FZ_STAMP(6);
	}
#ifdef SRH_PROFILE_PHASES
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (lane == 0) {
		// per-wave phase cycles: wave 0 in dbg_phase[], the other waves summed in the dbg_* scalars
		if (wave == 0) { for (int k = 0; k < 8; ++k) atomicAdd(&cnt->dbg_phase[k], ph[k]);
		                 atomicAdd(&cnt->dbg_waves, 1ull); }
		else { atomicAdd(&cnt->dbg_cycles, ph[4]); atomicAdd(&cnt->dbg_blocks, ph[7]);
		       atomicAdd(&cnt->dbg_total_cycles, ph[0] + ph[1] + ph[2] + ph[3] + ph[4] + ph[5] + ph[6] + ph[7]); }
	}
#endif
	block_count_add(&cnt->n_eval, n_eval);
	block_count_add(&cnt->n_eval_device, n_dev);
	block_count_add(&cnt->n_pixels, n_pix);
#undef FZ_STAMP
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
#ifdef SRH_PROFILE_PHASES
	{ static bool once = false; if (!once) { once = true; int nb = 0;
	  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)twoview_fused_kernel<5, SRH_FUSED_MAXC>, FZ_THREADS, sizeof(FusedSmem<5, SRH_FUSED_MAXC>));
	  fprintf(stderr, "[srh dbg] fused<5>: %zu bytes of LDS, %d workgroups per CU\n", sizeof(FusedSmem<5, SRH_FUSED_MAXC>), nb); } }
#endif
	if (P.num_depth_levels > SRH_FUSED_MAXC) return false;        // the label projections share the cost rows' LDS
	switch (P.window_radius) {
	case 5: SRH_FZ_LAUNCH(5, SRH_FUSED_MAXC)
	case 2: SRH_FZ_LAUNCH(2, SRH_FUSED_MAXC)
	default: return false;
	}
#undef SRH_FZ_LAUNCH
}

} // namespace srh
