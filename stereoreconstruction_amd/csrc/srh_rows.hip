// srh_rows.hip -- run-blocked evaluation of candidate lists (arbitrary epipolar geometry).
//
// The candidate-list path of srh_list.hip gathers (2R+1)^2 values of the other view per candidate
// and pass.  But the scan only needs *a cost per list entry*, not costs computed in list order:
// here the distinct candidates of a pixel are regrouped by image row -- a curve crosses few rows,
// and on each row its candidates fill (almost) a contiguous column span -- and every span is
// evaluated in blocks of 8 adjacent columns exactly like the dense row-aligned kernel: one
// right-image row segment of 8+2R values feeds 8 candidates x (2R+1) taps.  Costs are stored by
// (row, column) slot; the scan walks the list in the reference's order and looks each entry up.
//
//   twoview_rows_list_kernel  curve walk -> candidate list + per-row column spans + slot count
//   twoview_rows_cost_kernel  blocked weighted NCC of every slot                              (8(a) #8)
//   twoview_rows_scan_kernel  running-min WTA over the list in order, slot look-ups           (#9,#10)
//
// Bit-identical to tv_cost: same operations in the same order (skipped taps add +0.0).
#include "srh_internal.hpp"
#include "srh_geom.hpp"
#include "srh_walk.hpp"

#include <cstdio>

namespace srh {

#ifdef SRH_EXPERIMENT
__device__ int g_exp_rows_mode = 0;   // timing experiments of the list kernel: 1 no raster / visitor, 2 no Newton (root = guess), 3 no refraction; 9: count Newton's steps (slow: atomics), -1 prints the counts
__device__ unsigned long long g_exp_rl[4];
void exp_set_rows(int mode) {
	if (mode == -1) {                                              // print and clear the lockstep statistics
		unsigned long long h[4] = {0, 0, 0, 0};
		(void)hipDeviceSynchronize();
		(void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_exp_rl), sizeof(h));
		fprintf(stderr, "[srh exp] lockstep Newton: %llu roots, %.2f iterations each, %llu fell back to the loop (%.3g)\n", h[0], h[0] ? (double)h[1]/h[0] : 0.0, h[2], h[0] ? (double)h[2]/h[0] : 0.0);
		unsigned long long z[4] = {0, 0, 0, 0};
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_exp_rl), z, sizeof(z));
		return;
	}
	(void)hipMemcpyToSymbol(HIP_SYMBOL(g_exp_rows_mode), &mode, sizeof(int));
}
#endif

#define RW_NR 32                   // image rows a pixel's curve may cross (else: plain list path); a power of two
#define RW_LT 128                  // threads of the list / scan kernels

// ------------------------------------------------------------------ list + row spans
struct RowsListVisitor {
	uint32_t *out;
	int cap, n;
	unsigned visited;
	uint32_t prev;
	int ymin, ymax;
	short *lo, *hi;                                             // this thread's LDS columns (stride RW_LT): span of row cy at cy % RW_NR
	__device__ __forceinline__ void operator()(int cx, int cy) {
		++visited;
		const uint32_t e = (uint32_t)cx | ((uint32_t)cy << 16);
		if (e == prev) return;                                  // joint duplicate: can never change the WTA state
		prev = e;
		if (n < cap) out[(size_t)n*64] = e;                     // wave-tiled list
		++n;
		ymin = cy < ymin ? cy : ymin; ymax = cy > ymax ? cy : ymax;
		// column span per image row, kept in a ring of RW_NR rows: unambiguous while the curve crosses at most
		// RW_NR rows (otherwise the pixel is flagged and the spans are not used)
		const int r = (cy & (RW_NR - 1))*RW_LT;
		if (cx < lo[r]) lo[r] = (short)cx;
		if (cx > hi[r]) hi[r] = (short)cx;
	}
};

// walk_curve<false> (srh_walk.hpp) for this kernel, the same operations in the same order with the redundant ones taken out:
//   * the three quotients of a normalisation (dirv / r) and the two image coordinates (p.x / z, p.y / z) share their
//     divisor's half of the division (shared_divisor / div_by: the same quotient bits); p.z / z is never used;
//   * a segment with both end points inside the image needs none of LineWalk's bounds work (the closed-form skip of an
//     off-image prefix, its 64-bit arithmetic);
//   * the other camera's flags are read once.
// Results are bit-identical to walk_curve<false>: tests/test_gpu_parity.py compares the lists, entry for entry.
// quartic_root_0r (srh_geom.hpp; the oracle's quartic_root_0r) with the wave in LOCKSTEP.  The safeguarded Newton loop as
// written -- a data-dependent trip count per lane, its bracket bookkeeping and three exits in branches -- was two thirds of
// this kernel on C5 (11.4 of 16.9 ms: profiles/r05_c5_list_parts.txt): every lane waits for the wave's slowest, and the
// loop's control flow costs as much as its arithmetic.  Here every lane takes the same RL_K Newton steps, the loop's
// decisions made by selects: the iterates, the bracket [lo, hi] and the exits are the loop's own, operation for operation
// (f == 0 -> that x; a step of at most 1e-15 (|x| + r) -> the new x), so the root is the same number -- as long as no step
// leaves the bracket (the loop would bisect) and an exit is reached within RL_K steps.  A lane for which that does not
// hold takes the loop itself, from the start.  The wave leaves as soon as every lane has its root.
#define RL_K 7
__device__ __forceinline__ bool quartic_root_0r_lockstep(double a, double b, double c, double d, double e, double r, double guess, double &root) {
	const double f0 = e;
	const double fr = (((a*r + b)*r + c)*r + d)*r + e;
	if (!(r > 0.0) || !(f0 > 0.0) || !(fr < 0.0)) return quartic_root_0r(a, b, c, d, e, r, guess, root);   // (the degenerate entries: as written)
	double lo = 0.0, hi = r, x = guess, res = 0.0;
	if (!(x > lo && x < hi)) x = 0.5*(lo + hi);
	const double a4 = 4.0*a, b3 = 3.0*b, c2 = 2.0*c;
	bool done = false, bad = false;
#pragma unroll 1
	for (int it = 0; it < RL_K; ++it) {
		const double f = (((a*x + b)*x + c)*x + d)*x + e;
		const bool live = !done && !bad;
		if (live && f == 0.0) { done = true; res = x; }
		const bool act = live && !(f == 0.0);
		lo = (act && f > 0.0) ? x : lo;
		hi = (act && !(f > 0.0)) ? x : hi;
		const double df = ((a4*x + b3)*x + c2)*x + d;
		const double xn = x - f/df;
		const bool inb = xn >= lo && xn <= hi;
		bad = bad || (act && !inb);                              // the loop would bisect here: not this routine's case
		const double dx = fabs(xn - x);
		if (act && inb) {
			x = xn;
			if (dx <= 1e-15*(fabs(x) + r)) { done = true; res = x; }
		}
#ifdef SRH_EXPERIMENT
		if (g_exp_rows_mode == 9 && live) atomicAdd(&g_exp_rl[1], 1ull);
#endif
		if (__all(done || bad)) break;
	}
#ifdef SRH_EXPERIMENT
	if (g_exp_rows_mode == 9) { atomicAdd(&g_exp_rl[0], 1ull); if (!done) atomicAdd(&g_exp_rl[2], 1ull); }
#endif
	if (!done) return quartic_root_0r(a, b, c, d, e, r, guess, root);
	root = res;
	return true;
}

__device__ __forceinline__ bool project_refraction_sd(Vec3 &p, Vec3 pn, double pdist, double n, const Vec3 &bn) {
	const Vec3 proj = dot(bn, p)*bn;
	Vec3 dir = p - proj;
	const double y = dir.y;
	const double z = norm(proj);
	const double r = norm(dir);
	const double d = pdist;
	const double rr = r*r, nn = n*n, dd = d*d;
	{ const SharedDivisor rs = shared_divisor(r); dir = v3(div_by(dir.x, rs), div_by(dir.y, rs), div_by(dir.z, rs)); }   // normalized(dir)
	if (isnan_d(dir.x) || isnan_d(dir.y) || isnan_d(dir.z)) return false;
	const double qa = nn - 1;
	const double qb = -2*r*(nn - 1);
	const double qc = rr*(nn - 1) + dd*nn - (z - d)*(z - d);
	const double qd = -2*dd*nn*r;
	const double qe = dd*nn*rr;
	double root;
#ifdef SRH_EXPERIMENT
	if (g_exp_rows_mode == 2) root = r*d/(d + (z - d)/n); else
#endif
	if (!quartic_root_0r_lockstep(qa, qb, qc, qd, qe, r, r*d/(d + (z - d)/n), root)) return false;   // (the paraxial Snell point: srh_geom.hpp)
	const Vec3 pp = root*dir;
	const double py = pp.y;
	bool ok = false;
	if (py > -1e-3 && y > -1e-3) { if (py < y + 1e-3) ok = true; }
	else if (py < 1e-3 && y < 1e-3) { if (y < py + 1e-3) ok = true; }
	if (!ok) return false;
	p = pp + pdist*pn;
	return true;
}

template <class Visitor>
__device__ __forceinline__ void walk_curve_rows(const Ray &ray, const srh_camera &refcam, const ViewDev &oth,
                                                const srh_params &P, Visitor &vis, const double *__restrict__ tdist)
{
	const int OW = oth.w, OH = oth.h;
	const bool refr = oth.cam.is_refractive != 0, distorted = oth.cam.is_distorted != 0;
	double x1 = __builtin_nan(""), y1 = __builtin_nan("");
	int jx1 = 0, jy1 = 0;
	const Vec3 pn = normalized(load3(refcam.pdir));
	const double nd = dot(pn, ray.dir);
	if (fabs(nd) < 1e-10) return;                               // intersect() fails for every label
	const Vec3 oth_pn = load3(oth.cam.plane_normal);
	const Vec3 oth_bn = normalized(oth_pn);
	const double cx = oth.cam.K[2], cy = oth.cam.K[5], fx = oth.cam.K[0], fy = oth.cam.K[4];
	for (int d = 0; d < P.num_depth_levels; ++d) {
		// intersect_plane(ray, pn, tdist[d], point) with n . dir taken out of the loop (walk_curve)
		const Vec3 x0 = tdist[d]*pn;
		const double t = dot(pn, x0 - ray.src) / nd;
		if (t < 1e-10) continue;
		const Vec3 point = ray.src + t*ray.dir;
		// cam_project (camera.cpp:380-419)
		Vec3 pl = matvec(oth.cam.R, point) + load3(oth.cam.t);
#ifdef SRH_EXPERIMENT
		if (g_exp_rows_mode != 3)
#endif
		if (refr && !project_refraction_sd(pl, oth_pn, oth.cam.plane_dist, oth.cam.refr_index, oth_bn)) continue;
		const Vec3 pk = matvec(oth.cam.K, pl);
		const SharedDivisor zd = shared_divisor(pk.z);
		double px = div_by(pk.x, zd), py = div_by(pk.y, zd);
		if (distorted) {
			const double *k = oth.cam.dist;
			double x = (px - cx) / fx, y = (py - cy) / fy;
			const double r2 = x*x + y*y;
			const double cdist = 1 + ((k[4]*r2 + k[1])*r2 + k[0])*r2;
			x = x*cdist + 2*k[2]*x*y + k[3]*(r2 + 2*x*x);
			y = y*cdist + k[2]*(r2 + 2*y*y) + 2*k[3]*x*y;       // updated x, as the reference
			px = fx*x + cx;
			py = fy*y + cy;
		}
		const double x2 = px*P.image_scale;
		const double y2 = py*P.image_scale;
		if (isnan_d(x1)) { x1 = x2; y1 = y2; jx1 = trunc_sat(x2); jy1 = trunc_sat(y2); continue; }
		const double dx = x2 - x1, dy = y2 - y1;
		if (!(dx*dx + dy*dy >= 1)) continue;
		const int ix0 = jx1, iy0 = jy1, ix1 = trunc_sat(x2), iy1 = trunc_sat(y2);   // (the kept point's truncations travel with it)
		jx1 = ix1; jy1 = iy1;
		const bool inside = (unsigned)ix0 < (unsigned)OW && (unsigned)iy0 < (unsigned)OH && (unsigned)ix1 < (unsigned)OW && (unsigned)iy1 < (unsigned)OH;
#ifdef SRH_EXPERIMENT
		if (g_exp_rows_mode == 1) { vis.visited += (unsigned)(ix0 + iy1); x1 = x2; y1 = y2; continue; }
#endif
		LineWalk lw;
		if (inside) lw.begin(ix0, iy0, ix1, iy1, 0, 0);          // every point of the segment is in the image
		else lw.begin(ix0, iy0, ix1, iy1, OW, OH);               // 4-arg LineIterator, twoviewstereo.cpp:1028 (off-image points are dropped below)
		while (lw.has_next()) {
			int tx, ty;
			lw.current(tx, ty);
			if ((inside || ((unsigned)tx < (unsigned)OW && (unsigned)ty < (unsigned)OH)) && oth.mask[(size_t)ty*OW + tx] == 1) vis(tx, ty);
			lw.next();
		}
		x1 = x2; y1 = y2;
	}
}

// rowinfo(q, r) = xlo | width<<16 of image row ymin+r;  meta[q] = ymin | nrows<<16 (nrows 0: no candidates).
// Per-pixel arrays are wave-tiled so that the 64 pixels of a wave read and write them coalesced:
//   list entry k of pixel q   at cand   [((q/64)*cmax  + k)*64 + q%64]
//   row r of pixel q          at rowinfo[((q/64)*RW_NR + r)*64 + q%64]
//   cost slot s of pixel x of tile t (8 pixels of one image row, a wave tile of the cost kernel)
//                             at cost   [((t*smax) + s)*8 + x%8]
__global__ __launch_bounds__(RW_LT)
void twoview_rows_list_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P,
                              int y0, int nrows_band, uint32_t *__restrict__ cand, int cmax,
                              int32_t *__restrict__ count, uint32_t *__restrict__ rowinfo, int32_t *__restrict__ meta, int smax,
                              Counters *__restrict__ cnt, int *__restrict__ maxes /* [0] list length, [1] slots, [2] too many rows */,
                              const double *__restrict__ tdist)
{
	const ViewDev &L = views[ref];
	const int W = L.w;
	const size_t q = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
	__shared__ short s_lo[RW_NR][RW_LT], s_hi[RW_NR][RW_LT];
	unsigned n_eval = 0, n_pix = 0;
	int n_kept = 0, slots = 0, toomany = 0;
	if (q < (size_t)nrows_band*W) {
		const int x = (int)(q % W), y = y0 + (int)(q / W);
		int m = 0;
		if (L.mask[(size_t)y*W + x] == 1) {
			n_pix = 1;
			const Ray ray = cam_unproject(L.cam, (x + 0.5) / P.image_scale, (y + 0.5) / P.image_scale);
			uint32_t *mine = cand + (q >> 6)*(size_t)cmax*64 + (q & 63);
			for (int r = 0; r < RW_NR; ++r) { s_lo[r][threadIdx.x] = 32767; s_hi[r][threadIdx.x] = -1; }
			RowsListVisitor vis = { mine, cmax, 0, 0, 0xffffffffu, 2147483647, -1, &s_lo[0][threadIdx.x], &s_hi[0][threadIdx.x] };
			walk_curve_rows(ray, L.cam, views[oth], P, vis, tdist);
			n_eval = vis.visited;
			n_kept = vis.n;
			const int nr = vis.n > 0 ? vis.ymax - vis.ymin + 1 : 0;
			if (nr > RW_NR) toomany = 1;
			else if (nr > 0 && vis.n <= cmax) {
				for (int r = 0; r < nr; ++r) {
					const int ring = (vis.ymin + r) & (RW_NR - 1);
					const int lo = s_lo[ring][threadIdx.x], hi = s_hi[ring][threadIdx.x];
					const int wdt = hi >= lo ? hi - lo + 1 : 0;
					rowinfo[((q >> 6)*RW_NR + r)*64 + (q & 63)] = (uint32_t)(lo & 0xffff) | ((uint32_t)wdt << 16);
					slots += (wdt + 7) & ~7;                             // spans are stored in whole blocks of 8
				}
				if (slots <= smax) m = (vis.ymin & 0xffff) | (nr << 16);   // else: capacity too small, the pass is repeated
			}
		}
		count[q] = n_kept;
		meta[q] = m;
	}
	__shared__ int s_max[3];
	if (threadIdx.x < 3) s_max[threadIdx.x] = 0;
	__syncthreads();
	if (n_kept) atomicMax(&s_max[0], n_kept);
	if (slots) atomicMax(&s_max[1], slots);
	if (toomany) atomicMax(&s_max[2], 1);
	__syncthreads();
	if (threadIdx.x < 3 && s_max[threadIdx.x] > 0) atomicMax(&maxes[threadIdx.x], s_max[threadIdx.x]);
	block_count_add(&cnt->n_eval, n_eval);
	block_count_add(&cnt->n_pixels, n_pix);
	block_count_add(&cnt->n_listed, (unsigned)n_kept);
	block_count_add(&cnt->n_slots, (unsigned)slots);
}

void launch_twoview_rows_list(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                              int y0, int nrows, uint32_t *cand, int cmax, int32_t *count, uint32_t *rowinfo,
                              int32_t *meta, int smax, Counters *cnt, int *maxes, const double *tdist)
{
	const size_t n = (size_t)nrows*width;
	hipLaunchKernelGGL(twoview_rows_list_kernel, dim3((unsigned)((n + RW_LT - 1)/RW_LT)), dim3(RW_LT), 0, st,
	                   views, ref, oth, P, y0, nrows, cand, cmax, count, rowinfo, meta, smax, cnt, maxes, tdist);
}

// ------------------------------------------------------------------ blocked cost
typedef __attribute__((address_space(3))) void rc_lds_void;
typedef __attribute__((address_space(1))) const void rc_gbl_void;
#define RC_TP 32                   // pixels of a tile of the cost-slot / window-buffer layouts
#define RC_WT 8                    // pixels of a WAVE TILE: what one workgroup (= one wave) works on at a time
#define RC_G 8
#define RC_NCB 8
#define RC_THREADS (RC_WT*RC_G)

template <int R>
struct RowsSmem {
	static constexpr int WS = 2*R + 1;
	static constexpr int T = WS*WS;
	static constexpr int WP = (WS + 1) & ~1;
	static constexpr int WPIX = WS*WP;
	static constexpr int LW = RC_WT + 2*R;
	double w[WS][RC_WT][WP];                                   // the band buffer's "LDS image" of the wave tile: [window row][pixel][tap, padded]
	double lt[WS][LW];
	double meanL[RC_WT], totalW[RC_WT], sum2[RC_WT], sumA[RC_WT];   // sumA: sum of w_t*l_t - meanL, fused (one-pass form only)
	int lall[RC_WT];
	int meta[RC_WT];
	uint32_t rowinfo[RC_WT][RW_NR];
	unsigned short blk0[RC_WT][RW_NR + 2];                    // first 8-column block (task) of each row; [nr] = total
	// phase 2 work list: blocks that need the select form, (pixel << 16) | task, compacted over the tile
	static constexpr int GL_CAP = 512;
	unsigned int glist[GL_CAP];
	int glist_n;
	// phase 2, single candidates: (pixel << 16) | cost slot -- the few candidates of an otherwise fast block whose window in
	// the other view is not fully usable (the columns and rows next to its border, masked neighbourhoods)
	static constexpr int GS_CAP = 256;
	unsigned int gsingle[GS_CAP];
	int gsingle_n;
};

// AR: 0 = the reference's arithmetic; 3 = certified: the two sweeps of the fast form run with fused multiply-adds and
// every value is stored by the rule of the certified scan (srh_internal.hpp, CertBound; twoview_strip_cost_kernel): NaN
// for a candidate whose error bound is not below e0, the clamp itself above clamp + e0, anything else unclamped.  The
// per-pixel constants (meanL, totalWeight, sum2) and the select form stay in the reference's arithmetic.
template <int R, int AR>
__global__ __launch_bounds__(RC_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
void twoview_rows_cost_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P,
                              int y0, int nrows, const double *__restrict__ wbuf,
                              const uint8_t *__restrict__ full_oth,
                              const uint32_t *__restrict__ rowinfo, const int32_t *__restrict__ meta,
                              double *__restrict__ cost, int smax, Counters *__restrict__ cnt, const CertBound cb,
                              const double *__restrict__ pconst, const double *__restrict__ oth_tvp, const uint32_t *__restrict__ full_stat)
{
	constexpr bool FMA = AR != 0, CERT = AR == 3 || AR == 5, ONEPASS = AR == 5;   // 5: the certified ONE-PASS form (srh_internal.hpp, CertBound)
	constexpr int WS = 2*R + 1;
	constexpr int T = WS*WS;
	typedef RowsSmem<R> Smem;
	constexpr int WP = Smem::WP;
	constexpr int NR_ = RC_NCB + 2*R;
	extern __shared__ __align__(16) unsigned char smem_raw[];
	Smem &S = *reinterpret_cast<Smem *>(smem_raw);

	const ViewDev &L = views[ref];
	const ViewDev &Rv = views[oth];
	const int W = L.w, H = L.h, OW = Rv.w, OH = Rv.h;
	const int tiles_per_row = (W + RC_TP - 1)/RC_TP;
	const int tid = threadIdx.x;
	const int lane = tid;
	const int i = lane & 7;
	const int g = lane >> 3;
	const double nan = __builtin_nan("");
	// oth_tvp (round 6): the other view's NaN-bordered plane.  With it a block is fast as soon as ONE of its candidates has a
	// fully usable window: the row segments are read from the padded plane (a window that leaves the image reads NaN, which
	// stays inside the sums of the candidates whose window it is), the results of the others are not stored, and those go
	// candidate by candidate in phase 2 -- as in twoview_strip_cost_kernel.  Without it a block is fast when all 8 are.
	// (full_stat: pixels of the other view with a usable centre / with a fully usable window.  Single candidates pay when the
	// candidates they are for are the exception -- image borders, a few masked spots: C5 --; in a view that is mostly
	// silhouette edge (the bunny pair: C1) they come by the hundred per tile and the blocked select form is the better deal)
	const bool masked = oth_tvp != nullptr && (!full_stat || (unsigned long long)full_stat[1]*10ull >= (unsigned long long)full_stat[0]*9ull);
	const int SPR = padded_stride(OW);
	typedef const __attribute__((address_space(1))) double *gptr;
	const gptr rplane = masked ? (gptr)(oth_tvp + (size_t)(SRH_PADY - R)*SPR + (SRH_PADL - R)) : (gptr)(Rv.gray_tv - (ptrdiff_t)R*OW - R);
	const int rs = masked ? SPR : OW;

	// PERSISTENT WAVES (round 6): a workgroup is ONE wave; it draws wave tiles -- 8 adjacent pixels of an image row, a quarter
	// of a 32-pixel tile of the layouts -- from the launch's ticket counter until none is left.  (Four waves per 32-pixel tile
	// behind common barriers lost the three that waited while one evaluated the tile's few select-form blocks, and all of them
	// the difference between the slowest wave's rounds and their own.)
	const int nitems = tiles_per_row*(RC_TP/RC_WT)*nrows;
	unsigned n_dev = 0;
#ifdef SRH_ROWS_DBG
	unsigned d_task = 0, d_fast = 0, d_rows = 0, d_wavefast = 0, d_waveiter = 0;
	unsigned long long d_t[6] = { 0, 0, 0, 0, 0, 0 }, d_n[6] = { 0, 0, 0, 0, 0, 0 };   // cycles: ticket+staging, constants, phase 1, p2 blocks, p2 singles; counts
#define RC_STAMP(k) { const unsigned long long t_ = __builtin_readcyclecounter(); d_t[k] += t_ - d_last; d_last = t_; }
	unsigned long long d_last = __builtin_readcyclecounter();
#else
#define RC_STAMP(k)
#endif
	int next_item = 0;
	if (lane == 0) next_item = (int)atomicAdd(&cnt->strip_ticket, 1u);
	for (;;) {
	const int item = __builtin_amdgcn_readfirstlane(next_item);
	if (item >= nitems) break;
	const int tile = item/(RC_TP/RC_WT), sub = item % (RC_TP/RC_WT);
	const int trow = tile / tiles_per_row;
	const int x0 = (tile % tiles_per_row)*RC_TP + sub*RC_WT;
	const int y = y0 + trow;
	const int x = x0 + i;
	const size_t qbase = (size_t)trow*W + x0;
	double *const ctile = cost + ((size_t)trow*((W + RC_WT - 1)/RC_WT) + x0/RC_WT)*(size_t)smax*RC_WT;   // cost slot s of the wave tile's pixel pi at ctile[s*8 + pi]

	// ---- stage: the windows by LDS-DMA -- the band buffer has the layout of the LDS image ([tile][window row][pixel][WP],
	// srh_internal.hpp "layout B"), a window row of the wave tile's 8 pixels is 8*WP contiguous doubles: no register, no
	// ds_write on the way --; reference rows, row spans and the pixels' constants through registers; the NEXT tile's ticket
	// is drawn here, a tile ahead of its use
	double pcr[5] = { 0, 0, 0, 0, 0 };
	{
		static_assert(RC_TP == SRH_WTILE, "tile = window-buffer tile");
		constexpr int NBL = (WS*Smem::LW + RC_THREADS - 1)/RC_THREADS;
		constexpr int NBI = (RC_WT*RW_NR + RC_THREADS - 1)/RC_THREADS;
		const double *wt = wbuf + wimg_offset(W, R, trow, x0);
#pragma unroll
		for (int a = 0; a < WS; ++a)
			if (lane*16 < RC_WT*WP*8)
				__builtin_amdgcn_global_load_lds((rc_gbl_void *)((const char *)(wt + (size_t)a*(RC_TP*WP)) + lane*16),
				                                 (rc_lds_void *)&S.w[a][0][0], 16, 0, 0);
		double tl_[NBL];
		uint32_t ti_[NBI];
#pragma unroll
		for (int k = 0; k < NBL; ++k) {
			const int idx = tid + k*RC_THREADS;
			const int ty = idx / Smem::LW, tx = idx % Smem::LW;
			const int gx = x0 - R + tx, gy = y - R + ty;
			tl_[k] = (idx < WS*Smem::LW && gx >= 0 && gy >= 0 && gx < W && gy < H) ? L.gray_tv[(size_t)gy*W + gx] : nan;
		}
#pragma unroll
		for (int k = 0; k < NBI; ++k) {
			const int idx = tid + k*RC_THREADS;
			const int pi = idx / RW_NR;
			const size_t qq = qbase + pi;
			ti_[k] = (idx < RC_WT*RW_NR && x0 + pi < W) ? rowinfo[((qq >> 6)*RW_NR + idx % RW_NR)*64 + (qq & 63)] : 0u;
		}
		const int mt = (tid < RC_WT && x0 + tid < W) ? meta[qbase + tid] : 0;
		if (g == 0 && pconst) {
			const double *pc = pconst + ((size_t)trow*W + (x < W ? x : W - 1))*SRH_PC;
#pragma unroll
			for (int k = 0; k < 5; ++k) pcr[k] = pc[k];
		}
		if (lane == 0) next_item = (int)atomicAdd(&cnt->strip_ticket, 1u);
#pragma unroll
		for (int k = 0; k < NBL; ++k) {
			const int idx = tid + k*RC_THREADS;
			if (idx < WS*Smem::LW) S.lt[idx / Smem::LW][idx % Smem::LW] = tl_[k];
		}
#pragma unroll
		for (int k = 0; k < NBI; ++k) {
			const int idx = tid + k*RC_THREADS;
			if (idx < RC_WT*RW_NR) S.rowinfo[idx / RW_NR][idx % RW_NR] = ti_[k];
		}
		if (tid < RC_WT) S.meta[tid] = mt;
		__builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the window rows have landed in LDS
	}
	__syncthreads();
	RC_STAMP(0);

	// ---- per-pixel constants of the all-taps-usable form (one lane per pixel).  pconst (round 6): the weights kernel has made
	// them while the window was in its registers (the same operations in the same tap order: the same bits; SRH_PC doubles per
	// pixel) -- computed here they are two 121-tap loops on one lane in eight with the workgroup waiting, 7 % of the kernel on C5
	if (g == 0 && pconst) {
		const bool all = (x < W) && (S.meta[i] >> 16) > 0 && pcr[3] != 0.0;
		S.meanL[i] = pcr[0]; S.totalW[i] = (ONEPASS && all) ? pcr[3] : pcr[1]; S.sum2[i] = pcr[2]; S.lall[i] = all ? 1 : 0; S.sumA[i] = pcr[4];
		const int nr = S.meta[i] >> 16;
		int nblk = 0;
		for (int r = 0; r < nr; ++r) {
			S.blk0[i][r] = (unsigned short)nblk;
			nblk += ((int)(S.rowinfo[i][r] >> 16) + RC_NCB - 1)/RC_NCB;
		}
		S.blk0[i][nr] = (unsigned short)nblk;
	} else if (g == 0) {
		bool all = (x < W) && (S.meta[i] >> 16) > 0;
		double mL = 0, tw = 0;
#pragma unroll 1
		for (int row = 0; row < WS; ++row)
#pragma unroll
			for (int col = 0; col < WS; ++col) {
				const double gl = S.lt[row][i + col];
				const double wt = S.w[row][i][col];
				if (!(gl == gl && wt > P.weight_cutoff)) all = false;
				mL += wt*gl;
				tw += wt;
			}
		double s2 = 0, sA = 0;
		if (all && !(tw < 1e-10)) {
			mL /= tw;
#pragma unroll 1
			for (int row = 0; row < WS; ++row)
#pragma unroll
				for (int col = 0; col < WS; ++col) {
					const double a = S.w[row][i][col]*S.lt[row][i + col] - mL;
					s2 += a*a;
					if (ONEPASS) sA += __builtin_fma(S.w[row][i][col], S.lt[row][i + col], -mL);
				}
		} else all = false;
		S.meanL[i] = mL; S.totalW[i] = (ONEPASS && all) ? 1.0/tw : tw; S.sum2[i] = s2; S.lall[i] = all ? 1 : 0; S.sumA[i] = sA;   // (one-pass form: fl(1/totalWeight), onepass_finish)
		const int nr = S.meta[i] >> 16;
		int nblk = 0;
		for (int r = 0; r < nr; ++r) {
			S.blk0[i][r] = (unsigned short)nblk;
			nblk += ((int)(S.rowinfo[i][r] >> 16) + RC_NCB - 1)/RC_NCB;
		}
		S.blk0[i][nr] = (unsigned short)nblk;
	}
	if (tid == 0) { S.glist_n = 0; S.gsingle_n = 0; }
	__syncthreads();
	RC_STAMP(1);

	const Smem &CS = S;
	// one 8-column block of pixel `pi` (task = index of the block among the pixel's spans) in the blocked select
	// form: any validity pattern (image border, masked taps, cut-off weights); the same sums as the fast form with
	// every tap guarded -- a skipped tap adds +0.0
	auto general_block = [&](int pi, int task) {
		const int m = CS.meta[pi];
		const int ymin = (int)(short)(m & 0xffff);
		int r = 0;
		while (task >= (int)CS.blk0[pi][r + 1]) ++r;
		const uint32_t info = CS.rowinfo[pi][r];
		const int xlo = (int)(short)(info & 0xffff), wdt = (int)(info >> 16);
		const int b = task - (int)CS.blk0[pi][r];
		const int cy = ymin + r;
		const int c0 = xlo + b*RC_NCB;
		const int nv = wdt - b*RC_NCB < RC_NCB ? wdt - b*RC_NCB : RC_NCB;
		double *dst = ctile + pi + (size_t)task*RC_NCB*RC_WT;
			// blocked select form, any validity pattern (image border, masked taps, cut-off weights):
			// the same sums with every tap guarded; a skipped tap adds +0.0
			const int gx0 = c0 - R;
			// (window rows that lie on rows 0 .. H - 2 of the reference image: on the others every tap of the reference side is
			// unusable -- gray_tv is NaN on the last row as outside -- and every term a zero: left out, as in the strip kernel)
			const int gra = y < R ? R - y : 0, grb = y + R > H - 2 ? WS - (y + R - (H - 2)) : WS;
			double mLs[RC_NCB], mRs[RC_NCB], tws[RC_NCB];
#pragma unroll
			for (int j = 0; j < RC_NCB; ++j) { mLs[j] = 0.0; mRs[j] = 0.0; tws[j] = 0.0; }
#pragma unroll 1
			for (int row = gra; row < grb; ++row) {
				const int gy = cy - R + row;
				const bool rowok = gy >= 0 && gy < OH;
				const double *rp = Rv.gray_tv + (size_t)(rowok ? gy : 0)*OW;
				double rr[NR_];
#pragma unroll
				for (int k = 0; k < NR_; ++k) {
					const int gx = gx0 + k;
					rr[k] = (rowok && gx >= 0 && gx < OW) ? rp[gx] : nan;
				}
				// (no select per tap and candidate: an unusable value becomes 0.0 with a flag 0.0 / 1.0 beside it, a tap whose own
				// side is unusable gets weight 0.0, and what the reference skips is multiplied by that 0.0 -- strip_select_block,
				// srh_strip.hip, has the argument)
				double rv[NR_];
#pragma unroll
				for (int k = 0; k < NR_; ++k) { const bool okr = rr[k] == rr[k]; rv[k] = okr ? 1.0 : 0.0; rr[k] = okr ? rr[k] : 0.0; }
#pragma unroll
				for (int col = 0; col < WS; ++col) {
					const double gl = CS.lt[row][pi + col], wt = CS.w[row][pi][col];
					const bool okl = gl == gl && wt > P.weight_cutoff;
					const double w0 = okl ? wt : 0.0, pl0 = okl ? wt*gl : 0.0;
#pragma unroll
					for (int j = 0; j < RC_NCB; ++j) {
						mLs[j] += pl0*rv[col + j];
						mRs[j] += w0*rr[col + j];
						tws[j] += w0*rv[col + j];
					}
				}
			}
#pragma unroll
			for (int j = 0; j < RC_NCB; ++j) { mLs[j] /= tws[j]; mRs[j] /= tws[j]; }   // unused when tws < 1e-10
			double s1[RC_NCB], s2v[RC_NCB], s3[RC_NCB];
#pragma unroll
			for (int j = 0; j < RC_NCB; ++j) { s1[j] = 0.0; s2v[j] = 0.0; s3[j] = 0.0; }
#pragma unroll 1
			for (int row = gra; row < grb; ++row) {
				const int gy = cy - R + row;
				const bool rowok = gy >= 0 && gy < OH;
				const double *rp = Rv.gray_tv + (size_t)(rowok ? gy : 0)*OW;
				double rr[NR_];
#pragma unroll
				for (int k = 0; k < NR_; ++k) {
					const int gx = gx0 + k;
					rr[k] = (rowok && gx >= 0 && gx < OW) ? rp[gx] : nan;
				}
				double rv[NR_];
#pragma unroll
				for (int k = 0; k < NR_; ++k) { const bool okr = rr[k] == rr[k]; rv[k] = okr ? 1.0 : 0.0; rr[k] = okr ? rr[k] : 0.0; }
#pragma unroll
				for (int col = 0; col < WS; ++col) {
					const double gl = CS.lt[row][pi + col], wt = CS.w[row][pi][col];
					const bool okl = gl == gl && wt > P.weight_cutoff;
					const double pl = wt*(gl == gl ? gl : 0.0), kl = okl ? 1.0 : 0.0;
#pragma unroll
					for (int j = 0; j < RC_NCB; ++j) {
						const double k = kl*rv[col + j];                       // 1.0: the tap counts for this candidate
						const double a = (pl - mLs[j])*k, bq = (wt*rr[col + j] - mRs[j])*k;
						s1[j] += a*bq;
						s2v[j] += a*a;
						s3[j] += bq*bq;
					}
				}
			}
#pragma unroll
			for (int j = 0; j < RC_NCB; ++j) {
				if (j < nv) {
					double result = P.bad_ret;
					if (!(tws[j] < 1e-10)) {
						const double v = 255*(1.0 - fabs(s1[j]) / sqrt(s2v[j] * s3[j]));
						result = (v < P.max_color_diff) ? v : P.max_color_diff;
					}
					dst[j*RC_WT] = result;
				}
			}
	};
	if (x < W) {
		const int m = CS.meta[i];
		const int ymin = (int)(short)(m & 0xffff), nr = m >> 16;
		double *crow = ctile + i;                                     // wave-tile-transposed: slot s at crow[s*8]
		const bool lall = CS.lall[i] != 0;
		const double mL = CS.meanL[i], tw = CS.totalW[i], s2 = CS.sum2[i];
		const double sig3 = CERT ? cb.sigma3(s2) : 0.0;                // certified: smallest sum3 the bound covers for this pixel
		// task t = the t-th 8-column block of the pixel's spans (rows in order); lane g takes t = g, g+8, ...
		// its costs live in slots [8t, 8t+8)
		const int ntask = CS.blk0[i][nr];
		int r = 0;
		for (int task = g; task < ntask; task += RC_G) {
			while (task >= (int)CS.blk0[i][r + 1]) ++r;
			{
				const uint32_t info = CS.rowinfo[i][r];
				const int xlo = (int)(short)(info & 0xffff), wdt = (int)(info >> 16);
				const int b = task - (int)CS.blk0[i][r];
				const int cy = ymin + r;
				const int c0 = xlo + b*RC_NCB;
				const int nv = wdt - b*RC_NCB < RC_NCB ? wdt - b*RC_NCB : RC_NCB;
				n_dev += nv;
				// a span's last, partial block is moved left to end on the span's last column, so that the fast
				// form always works on 8 usable columns (the extra ones are recomputed and dropped)
				// (a span that starts left of column 8 - nv cannot be moved: its block stays where it is, the extra columns lie
				// right of the span -- only the masked form takes such a block)
				const int sh = c0 - (RC_NCB - nv) >= 0 ? RC_NCB - nv : 0;
				const int c0s = c0 - sh;
				const unsigned own = ((1u << nv) - 1u) << sh;             // the block's own candidates
				unsigned vm = 0;                                          // ... of which the fast form stores
				bool fast = false;
				if (lall && (masked || sh == RC_NCB - nv)) {
					const uint8_t *fp = full_oth + (size_t)cy*OW + c0s;
					unsigned fm = 0;
#pragma unroll
					for (int j = 0; j < RC_NCB; ++j) fm |= (c0s + j < OW && fp[j]) ? 1u << j : 0u;
					if (masked) { vm = fm & own; fast = vm != 0; }
					else { fast = fm == 0xffu; vm = fast ? own : 0u; }
				}
				double *dst = crow + (size_t)task*RC_NCB*RC_WT;
#ifdef SRH_ROWS_DBG
				++d_task; d_fast += fast ? 1 : 0; if (g == 0 && task == 0) d_rows += nr;
				if (lane == __ffsll((long long)__ballot(1)) - 1) { ++d_waveiter; }
				if (__all(fast) && lane == __ffsll((long long)__ballot(1)) - 1) ++d_wavefast;
#endif
				if (fast && ONEPASS) {
					// certified one-pass form: P = sum w r, Q = sum ((w l - meanL) w) r, U = sum w^2 r^2 in ONE sweep over the window
					// (twoview_strip_cost_kernel); the other view's row segments come from L1 / L2 once instead of twice
					const gptr rbase = rplane + (size_t)cy*rs + c0s;
					const double SA = CS.sumA[i];
					double r[NR_], q[NR_], wv[WS], lv[WS], P_[RC_NCB], Q_[RC_NCB], U_[RC_NCB];
					{
						const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[0][i][0]);
#pragma unroll
						for (int k = 0; k < NR_; ++k) r[k] = rbase[k];
#pragma unroll
						for (int m = 0; m < (WS - 1)/2; ++m) { const double2 v = wp[m]; wv[2*m] = v.x; wv[2*m + 1] = v.y; }
						wv[WS - 1] = CS.w[0][i][WS - 1];
#pragma unroll
						for (int col = 0; col < WS; ++col) lv[col] = CS.lt[0][i + col];
					}
#pragma unroll
					for (int j = 0; j < RC_NCB; ++j) { P_[j] = 0.0; Q_[j] = 0.0; U_[j] = 0.0; }
#pragma unroll 1
					for (int row = 0; row < WS; ++row) {
						const int nrow = row + 1 < WS ? row + 1 : 0;          // (the last refill is never used)
						const gptr rp = rbase + (size_t)nrow*rs;
						const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[nrow][i][0]);
						const double *lp = &CS.lt[nrow][i];
#pragma unroll
						for (int k = 0; k < RC_NCB - 1; ++k) q[k] = r[k]*r[k];
#pragma unroll
						for (int col = 0; col < WS; ++col) {
							q[col + RC_NCB - 1] = r[col + RC_NCB - 1]*r[col + RC_NCB - 1];
							const double a = __builtin_fma(wv[col], lv[col], -mL);
							const double c = a*wv[col], d = wv[col]*wv[col];
							__builtin_amdgcn_sched_barrier(0);
#pragma unroll
							for (int j = 0; j < RC_NCB; ++j) P_[j] = __builtin_fma(wv[col], r[col + j], P_[j]);
#pragma unroll
							for (int j = 0; j < RC_NCB; ++j) Q_[j] = __builtin_fma(c, r[col + j], Q_[j]);
#pragma unroll
							for (int j = 0; j < RC_NCB; ++j) U_[j] = __builtin_fma(d, q[col + j], U_[j]);
							__builtin_amdgcn_sched_barrier(0);
							lv[col] = lp[col];
							if (col & 1) {
								r[col - 1] = rp[col - 1]; r[col] = rp[col];
								const double2 u = wp[col >> 1]; wv[col - 1] = u.x; wv[col] = u.y;
							}
							__builtin_amdgcn_sched_barrier(0);
						}
#pragma unroll
						for (int k = WS - 1; k < NR_; ++k) r[k] = rp[k];
						wv[WS - 1] = CS.w[nrow][i][WS - 1];
					}
					constexpr double TT = (double)T;
#pragma unroll
					for (int j = 0; j < RC_NCB; ++j) {
						if ((vm >> j) & 1u) {
							bool okc;
							const double v = onepass_finish(P_[j], Q_[j], U_[j], SA, tw, s2, TT, sig3, cb.zmax2, okc);   // (tw: 1/totalWeight in this form)
							dst[(j - sh)*RC_WT] = !okc ? __builtin_nan("") : (v > cb.m_hi ? P.max_color_diff : v);
						}
					}
				} else if (fast) {
					// blocked fast form (srh_dense.hip): a row segment of NCB+2R values of the other view is
					// read once per window row and shared by the NCB candidates and 2R+1 taps
					// (a global pointer, not a generic one: flat loads would share the LDS counter, and every wait for a
					// weight would then also wait for the row segments in flight)
					const gptr rbase = rplane + (size_t)cy*rs + c0s;
					// Both sweeps are scheduled by hand like the dense kernel's (srh_dense.hip): r[] / wv[] / av[] hold the
					// current window row; as soon as a value has had its last use its register is refilled with the next
					// row's value (the other view's segment from L1/L2, the weights and the reference row from LDS), so the
					// loads are always a row ahead; products first, sums second, so that no instruction waits on its neighbour.
					static_assert(WS % 2 == 1 && WP % 2 == 0, "odd window, window rows padded to 16 bytes");
					double r[NR_], wv[WS], acc[RC_NCB];
					{
						const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[0][i][0]);
#pragma unroll
						for (int k = 0; k < NR_; ++k) r[k] = rbase[k];
#pragma unroll
						for (int m = 0; m < (WS - 1)/2; ++m) { const double2 v = wp[m]; wv[2*m] = v.x; wv[2*m + 1] = v.y; }
						wv[WS - 1] = CS.w[0][i][WS - 1];
					}
#pragma unroll
					for (int j = 0; j < RC_NCB; ++j) acc[j] = 0.0;
#pragma unroll 1
					for (int row = 0; row < WS; ++row) {
						const int nrow = row + 1 < WS ? row + 1 : 0;          // last refill = row 0, for the second sweep
						const gptr rp = rbase + (size_t)nrow*rs;
						const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[nrow][i][0]);
#pragma unroll
						for (int col = 0; col < WS; ++col) {
							if (FMA) {
#pragma unroll
								for (int j = 0; j < RC_NCB; ++j) acc[j] = __builtin_fma(wv[col], r[col + j], acc[j]);
							} else {
								double pr[RC_NCB];
#pragma unroll
								for (int j = 0; j < RC_NCB; ++j) pr[j] = wv[col]*r[col + j];
								__builtin_amdgcn_sched_barrier(0);
#pragma unroll
								for (int j = 0; j < RC_NCB; ++j) acc[j] += pr[j];            // meanR += weight*gray
							}
							__builtin_amdgcn_sched_barrier(0);                  // keep each refill where it is written
							if (col & 1) {                                      // r[col-1], r[col], wv[col-1], wv[col] are dead
								r[col - 1] = rp[col - 1]; r[col] = rp[col];
								const double2 u = wp[col >> 1]; wv[col - 1] = u.x; wv[col] = u.y;
								__builtin_amdgcn_sched_barrier(0);
							}
						}
#pragma unroll
						for (int k = WS - 1; k < NR_; ++k) r[k] = rp[k];
						wv[WS - 1] = CS.w[nrow][i][WS - 1];
					}
					double mR[RC_NCB], s1[RC_NCB], s3[RC_NCB], av[WS];
#pragma unroll
					for (int col = 0; col < WS; ++col) av[col] = CS.lt[0][i + col];
#pragma unroll
					for (int j = 0; j < RC_NCB; ++j) { mR[j] = acc[j]/tw; s1[j] = 0.0; s3[j] = 0.0; }
#pragma unroll 1
					for (int row = 0; row < WS; ++row) {
						const int nrow = row + 1 < WS ? row + 1 : 0;
						const gptr rp = rbase + (size_t)nrow*rs;
						const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[nrow][i][0]);
						const double *lp = &CS.lt[nrow][i];
#pragma unroll
						for (int col = 0; col < WS; ++col) {
							const double wt = wv[col];
							if (FMA) {
								const double a = __builtin_fma(wt, av[col], -mL);
								double bb[RC_NCB];
#pragma unroll
								for (int j = 0; j < RC_NCB; ++j) bb[j] = __builtin_fma(wt, r[col + j], -mR[j]);
								__builtin_amdgcn_sched_barrier(0);
#pragma unroll
								for (int j = 0; j < RC_NCB; ++j) { s1[j] = __builtin_fma(a, bb[j], s1[j]); s3[j] = __builtin_fma(bb[j], bb[j], s3[j]); }
							} else {
								double bb[RC_NCB], u1[RC_NCB], u3[RC_NCB];
								const double pa = wt*av[col];
#pragma unroll
								for (int j = 0; j < RC_NCB; ++j) bb[j] = wt*r[col + j];
								__builtin_amdgcn_sched_barrier(0);
								const double a = pa - mL;                           // pixel_gray_l - meanL
#pragma unroll
								for (int j = 0; j < RC_NCB; ++j) bb[j] = bb[j] - mR[j];   // pixel_gray_r - meanR
								__builtin_amdgcn_sched_barrier(0);
#pragma unroll
								for (int j = 0; j < RC_NCB; ++j) { u1[j] = a*bb[j]; u3[j] = bb[j]*bb[j]; }
								__builtin_amdgcn_sched_barrier(0);
#pragma unroll
								for (int j = 0; j < RC_NCB; ++j) { s1[j] += u1[j]; s3[j] += u3[j]; }
							}
							__builtin_amdgcn_sched_barrier(0);
							av[col] = lp[col];
							if (col & 1) {
								r[col - 1] = rp[col - 1]; r[col] = rp[col];
								const double2 u = wp[col >> 1]; wv[col - 1] = u.x; wv[col] = u.y;
							}
							__builtin_amdgcn_sched_barrier(0);
						}
#pragma unroll
						for (int k = WS - 1; k < NR_; ++k) r[k] = rp[k];
						wv[WS - 1] = CS.w[nrow][i][WS - 1];
					}
#pragma unroll
					for (int j = 0; j < RC_NCB; ++j) {
						if ((vm >> j) & 1u) {
							const double v = 255*(1.0 - fabs(s1[j]) / sqrt(s2 * s3[j]));
							if (CERT) dst[(j - sh)*RC_WT] = !(s3[j] >= sig3) ? __builtin_nan("") : (v > cb.m_hi ? P.max_color_diff : v);
							else dst[(j - sh)*RC_WT] = (v < P.max_color_diff) ? v : P.max_color_diff;
						}
					}
				}
				// what the fast form has left.  Masked form: the candidates of a pixel with a fully usable window of its own whose
				// window in the OTHER view is not (the columns and rows next to its border, masked neighbourhoods) go one by one in
				// phase 2 -- a wave tile has a few dozen of them, one iteration of single candidates takes a quarter of the time of
				// one iteration of blocks in the select form (52 000 against 237 000 cycles per tile that has them:
				// profiles/r06_c5_rows_phases.txt).  Pixels with unusable taps of their own (the image's border rows and
				// columns: every block of theirs): the blocked select form, whole iterations of it
				unsigned need = own & ~vm;
				if (masked && lall && need) {
					const int n1 = __builtin_popcount(need);
					const int at = atomicAdd(&S.gsingle_n, n1);
					if (at + n1 <= Smem::GS_CAP) {
						int k = 0;
#pragma unroll
						for (int j = 0; j < RC_NCB; ++j)
							if ((need >> j) & 1u) { S.gsingle[at + k] = ((unsigned)i << 16) | (unsigned)(task*RC_NCB + j - sh); ++k; }
						need = 0;
					} else {
						for (int k = at; k < at + n1 && k < Smem::GS_CAP; ++k) S.gsingle[k] = 0xffffffffu;   // reserved, void
					}
				}
				if (need) {
					// the blocked select form (which stores all of the block's candidates, the fast ones once more: the
					// reference's own numbers), in phase 2, where such blocks of the whole tile are spread over all lanes
					const int slot = atomicAdd(&S.glist_n, 1);
					if (slot < Smem::GL_CAP) S.glist[slot] = ((unsigned)i << 16) | (unsigned)task;
					else general_block(i, task);                       // list full (cannot happen below 64 blocks per pixel)
				}
			}
		}
	}
	// ---- phase 2: the listed blocks in the select form, one per lane
	__syncthreads();
	RC_STAMP(2);
	{
		const int nl = S.glist_n < Smem::GL_CAP ? S.glist_n : Smem::GL_CAP;
		for (int k = tid; k < nl; k += RC_THREADS) general_block((int)(S.glist[k] >> 16), (int)(S.glist[k] & 0xffffu));
		RC_STAMP(3);
#ifdef SRH_ROWS_DBG
		{ int tot = 0; for (int k = 0; k < RC_WT; ++k) tot += (x0 + k < W) ? (int)CS.blk0[k][CS.meta[k] >> 16] : 0; d_n[5] += (tot + 63)/64; }
		d_n[0] += 1; d_n[1] += nl > 0; d_n[2] += nl; d_n[3] += S.gsingle_n > 0; d_n[4] += S.gsingle_n;
#endif
		// single candidates (masked form only): cost_ncc for any validity pattern, window and reference rows from LDS, the
		// other view's taps from its NaN-bordered plane
		const int ns = S.gsingle_n < Smem::GS_CAP ? S.gsingle_n : Smem::GS_CAP;
		for (int k = tid; k < ns; k += RC_THREADS) {
			const unsigned e = S.gsingle[k];
			if (e == 0xffffffffu) continue;
			const int pi = (int)(e >> 16), slot = (int)(e & 0xffffu), task = slot/RC_NCB;
			const int m = CS.meta[pi];
			int r = 0;
			while (task >= (int)CS.blk0[pi][r + 1]) ++r;
			const uint32_t info = CS.rowinfo[pi][r];
			const int cx = (int)(short)(info & 0xffff) + (slot - (int)CS.blk0[pi][r]*RC_NCB), cy = (int)(short)(m & 0xffff) + r;
			const double *rp = oth_tvp + (size_t)(cy + SRH_PADY - R)*SPR + (cx + SRH_PADL - R);
			ctile[pi + (size_t)slot*RC_WT] =
				window_exact_cost<R>(&CS.w[0][pi][0], RC_WT*WP, 1, &CS.lt[0][pi], rp, Smem::LW, SPR, P);
		}
	}
	RC_STAMP(4);
	__syncthreads();                        // the wave is through with this tile's LDS
	}
	block_count_add(&cnt->n_eval_device, n_dev);
#ifdef SRH_ROWS_DBG
	block_count_add(&cnt->dbg_phase[6], d_task);
	block_count_add(&cnt->dbg_phase[7], d_fast);
	block_count_add(&cnt->dbg_cycles, d_rows);
	block_count_add(&cnt->dbg_blocks, d_waveiter);
	block_count_add(&cnt->dbg_total_cycles, d_wavefast);
	if (lane == 0) for (int k = 0; k < 6; ++k) { atomicAdd(&cnt->dbg_wave[k], d_t[k]); atomicAdd(&cnt->dbg_wave[8 + k], d_n[k]); }
#endif
#undef RC_STAMP
}

bool launch_twoview_rows_cost(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                              int y0, int nrows, const double *wbuf, const uint8_t *full_oth,
                              const uint32_t *rowinfo, const int32_t *meta, double *cost, int smax, Counters *cnt, int arith,
                              const double *pconst, const double *oth_tvp, int num_cus, const uint32_t *full_stat)
{
	// persistent single-wave workgroups, two per SIMD; Counters::strip_ticket is zero at launch (the caller's memset)
	const int tiles = (width + RC_TP - 1)/RC_TP;
	const int items = tiles*(RC_TP/RC_WT)*nrows;
	// (a small band -- at most four tiles per resident wave --: one wave per tile, so that the other pass's kernels find room
	// beside this one as its waves retire: C1 2.19 -> 2.0 ms per step with the two passes side by side)
	const dim3 grid((unsigned)(items <= 4*num_cus*8 ? items : num_cus*8));
	const CertBound cb = cert_bound(P);
#define SRH_RC_LAUNCH2(RR, AA)                                                                              \
	{                                                                                                       \
		/* per device, hence on every launch */                                                             \
		(void)hipFuncSetAttribute((const void *)twoview_rows_cost_kernel<RR, AA>,                           \
		                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RowsSmem<RR>));            \
		hipLaunchKernelGGL((twoview_rows_cost_kernel<RR, AA>), grid, dim3(RC_THREADS), sizeof(RowsSmem<RR>), st,  \
		                   views, ref, oth, P, y0, nrows, wbuf, full_oth, rowinfo, meta, cost, smax, cnt, cb, pconst, oth_tvp, full_stat);  \
		return true;                                                                                        \
	}
#define SRH_RC_LAUNCH(RR) { if (arith == 5) SRH_RC_LAUNCH2(RR, 5) else if (arith == 3) SRH_RC_LAUNCH2(RR, 3) else SRH_RC_LAUNCH2(RR, 0) }
	switch (P.window_radius) {
	case 1: SRH_RC_LAUNCH(1)
	case 2: SRH_RC_LAUNCH(2)
	case 3: SRH_RC_LAUNCH(3)
	case 4: SRH_RC_LAUNCH(4)
	case 5: SRH_RC_LAUNCH(5)
	default: return false;
	}
#undef SRH_RC_LAUNCH
#undef SRH_RC_LAUNCH2
}

// Certified arithmetic: every cost slot of the flagged pixels (cflag[1 .. 1 + count)) once more, in the reference's
// arithmetic (window_exact_cost on the NaN-bordered planes: any validity pattern) -- one 256-lane workgroup per pixel, a
// lane per slot; launched for the redo's capacity, the count stays on the device (twoview_refill_kernel).
template <int R>
__global__ __launch_bounds__(256)
void twoview_rows_refill_kernel(int W, srh_params P, int y0, const uint32_t *__restrict__ cflag, int cap,
                                const double *__restrict__ wbuf, const double *__restrict__ ref_tvp, const double *__restrict__ oth_tvp,
                                int OW, const uint32_t *__restrict__ rowinfo, const int32_t *__restrict__ meta,
                                double *__restrict__ cost, int smax, Counters *__restrict__ cnt)
{
	const uint32_t nflag = cflag[0];
	if (blockIdx.x == 0 && threadIdx.x == 0 && nflag > (uint32_t)cap) atomicAdd(&cnt->cert_overflow, 1ull);
	unsigned n = 0;
	for (uint32_t f = blockIdx.x; f < nflag && f < (uint32_t)cap; f += gridDim.x) {      // (a few hundred workgroups share the list)
	const size_t q = cflag[1 + f];
	const int x = (int)(q % W), trow = (int)(q / W), y = y0 + trow;
	const WindowAt wa = window_at<R>(wbuf, 1, W, trow, x);        // (the row-run path's band buffer has the LDS-image layout)
	const int SPL = padded_stride(W), SPR = padded_stride(OW);
	const double *lp = ref_tvp + (size_t)(y + SRH_PADY - R)*SPL + (x + SRH_PADL - R);
	const int tiles_per_row = (W + 7) >> 3;                       // (cost slots: tiles of 8 pixels, the cost kernel's wave tiles)
	double *crow = cost + ((size_t)trow*tiles_per_row + (x >> 3))*(size_t)smax*8 + (x & 7);
	const int m = meta[q];
	const int ymin = (int)(short)(m & 0xffff), nr = m >> 16;
	int base = 0;
	for (int r = 0; r < nr; ++r) {
		const uint32_t info = rowinfo[((q >> 6)*RW_NR + r)*64 + (q & 63)];
		const int xlo = (int)(short)(info & 0xffff), wdt = (int)(info >> 16);
		for (int k = (int)threadIdx.x; k < wdt; k += 256) {
			const double *rp = oth_tvp + (size_t)(ymin + r + SRH_PADY - R)*SPR + (xlo + k + SRH_PADL - R);
			crow[(size_t)(base + k)*8] = window_exact_cost<R>(wa.wq, wa.wrow, wa.wcol, lp, rp, SPL, SPR, P);
			++n;
		}
		base += (wdt + 7) & ~7;
	}
	}
	block_count_add(&cnt->n_eval_device, n);
}

bool launch_twoview_rows_refill(hipStream_t st, int width, int oth_width, const srh_params &P, int y0,
                                const uint32_t *cflag, int cap, const double *wbuf, const double *ref_tvp, const double *oth_tvp,
                                const uint32_t *rowinfo, const int32_t *meta, double *cost, int smax, Counters *cnt)
{
	if (cap <= 0) return true;
#define SRH_RR(RR) case RR: hipLaunchKernelGGL(twoview_rows_refill_kernel<RR>, dim3((unsigned)(cap < 4096 ? cap : 4096)), dim3(256), 0, st, width, P, y0, cflag, cap, \
	                                           wbuf, ref_tvp, oth_tvp, oth_width, rowinfo, meta, cost, smax, cnt); return true;
	switch (P.window_radius) { SRH_RR(1) SRH_RR(2) SRH_RR(3) SRH_RR(4) SRH_RR(5) default: return false; }
#undef SRH_RR
}

// ------------------------------------------------------------------ scan with slot look-ups
#define RS_QN 16

// CERT / LISTED: as twoview_scan_kernel (srh_dense.hip): the certified scan on fused costs flags the pixels with a decision
// the error bound does not cover; the listed scan is the exact scan of those pixels after twoview_rows_refill_kernel.
template <bool CERT, bool LISTED>
__global__ __launch_bounds__(RW_LT)
void twoview_rows_scan_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P,
                              int y0, int nrows, const int32_t *__restrict__ count,
                              const uint32_t *__restrict__ cand, int cmax,
                              const uint32_t *__restrict__ rowinfo, const int32_t *__restrict__ meta,
                              const double *__restrict__ cost, int smax, uint32_t *__restrict__ cflag, int nlist,
                              Counters *__restrict__ cnt, const CertBound cb)
{
	const ViewDev &L = views[ref];
	const ViewDev &Rv = views[oth];
	const int W = L.w;
	size_t q = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
	__shared__ uint32_t s_row[RW_NR][RW_LT];                     // xlo | slot base << 16
	if (LISTED) {
		if (q >= (size_t)nlist || q >= (size_t)cflag[0]) return;     // (launched for a capacity; the count is on the device)
		q = cflag[1 + q];
	} else if (q >= (size_t)nrows*W) return;
	const int x = (int)(q % W), y = y0 + (int)(q / W);
	const size_t pv = (size_t)y*W + x;
	double depth = __builtin_nan("");
	if (L.mask[pv] == 1) {
		const int m = meta[q];
		const int ymin = (int)(short)(m & 0xffff), nr = m >> 16;
		int base = 0;
		for (int r = 0; r < nr; ++r) {
			const uint32_t info = rowinfo[((q >> 6)*RW_NR + r)*64 + (q & 63)];
			const int wdt = (int)(info >> 16);
			s_row[r][threadIdx.x] = (info & 0xffffu) | ((uint32_t)base << 16);
			base += (wdt + 7) & ~7;
		}
		const int n = nr > 0 ? (count[q] < cmax ? count[q] : cmax) : 0;
		const uint32_t *clist = cand + (q >> 6)*(size_t)cmax*64 + (q & 63);
		const int tiles_per_row = (W + 7) >> 3;
		const double *crow = cost + ((size_t)(q / W)*tiles_per_row + (x >> 3))*(size_t)smax*8 + (x & 7);
		double minCost = __builtin_inf(), secondBest = __builtin_inf();
		uint32_t win = 0xffffffffu;
		bool flag = false;
		for (int k0 = 0; k0 < n; k0 += RS_QN) {
			uint32_t e[RS_QN];
			double c[RS_QN];
#pragma unroll
			for (int j = 0; j < RS_QN; ++j) e[j] = k0 + j < n ? clist[(size_t)(k0 + j)*64] : 0xffffffffu;
#pragma unroll
			for (int j = 0; j < RS_QN; ++j) {
				if (k0 + j < n) {
					const int cx = (int)(e[j] & 0xffffu), r = (int)(e[j] >> 16) - ymin;
					const uint32_t ri = s_row[r][threadIdx.x];
					c[j] = crow[(size_t)((int)(ri >> 16) + cx - (int)(short)(ri & 0xffffu))*8];
				} else c[j] = __builtin_inf();
			}
#pragma unroll
			for (int j = 0; j < RS_QN; ++j) {
				if (CERT && k0 + j < n) {
					// (srh_dense.hip, twoview_scan_kernel: tolerance, "sure" values, the winner seen again)
					const double t = c[j] + P.wta_margin;
					if (!(fabs(t - minCost) > 2.5*cb.e0) && e[j] != win &&
					    !(cert_sure(c[j], P.max_color_diff, cb.m_hi) && cert_sure(minCost, P.max_color_diff, cb.m_hi))) flag = true;
				}
				if (k0 + j < n && c[j] + P.wta_margin < minCost) {     // twoviewstereo.cpp:293-301
					secondBest = minCost;
					minCost = c[j];
					win = e[j];
				}
			}
		}
		if (win != 0xffffffffu) {
			const Ray ray = cam_unproject(L.cam, (x + 0.5) / P.image_scale, (y + 0.5) / P.image_scale);
			depth = candidate_depth(L.cam, Rv.cam, P, ray, (int)(win & 0xffffu), (int)(win >> 16));
		}
		if (minCost > P.second_best_factor*secondBest)                 // twoviewstereo.cpp:304-305
			depth = __builtin_inf();
		if (CERT) {
			if (win != 0xffffffffu) {
				const double rhs = P.second_best_factor*secondBest;
				const double tol = __builtin_fma(fmin(fabs(rhs), 1e300), 1e-15, (1.0 + fabs(P.second_best_factor))*cb.e0);
				if (!(fabs(minCost - rhs) > tol) &&
				    !(cert_sure(minCost, P.max_color_diff, cb.m_hi) && cert_sure(secondBest, P.max_color_diff, cb.m_hi))) flag = true;
			}
			if (flag) { cflag[1 + atomicAdd(&cflag[0], 1u)] = (uint32_t)q; atomicAdd(&cnt->n_flagged, 1ull); }
		}
	}
	L.depth[pv] = depth;
}

// cflag == nullptr: exact scan.  cflag, nlist < 0: certified scan.  cflag, nlist >= 0: exact scan of the listed pixels.
void launch_twoview_rows_scan(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                              int y0, int nrows, const int32_t *count, const uint32_t *cand, int cmax,
                              const uint32_t *rowinfo, const int32_t *meta, const double *cost, int smax,
                              uint32_t *cflag, int nlist, Counters *cnt)
{
	const size_t n = (size_t)nrows*width;
	const CertBound cb = cert_bound(P);
	if (!cflag)
		hipLaunchKernelGGL((twoview_rows_scan_kernel<false, false>), dim3((unsigned)((n + RW_LT - 1)/RW_LT)), dim3(RW_LT), 0, st,
		                   views, ref, oth, P, y0, nrows, count, cand, cmax, rowinfo, meta, cost, smax, nullptr, 0, cnt, cb);
	else if (nlist < 0)
		hipLaunchKernelGGL((twoview_rows_scan_kernel<true, false>), dim3((unsigned)((n + RW_LT - 1)/RW_LT)), dim3(RW_LT), 0, st,
		                   views, ref, oth, P, y0, nrows, count, cand, cmax, rowinfo, meta, cost, smax, cflag, 0, cnt, cb);
	else if (nlist > 0)
		hipLaunchKernelGGL((twoview_rows_scan_kernel<false, true>), dim3((unsigned)((nlist + RW_LT - 1)/RW_LT)), dim3(RW_LT), 0, st,
		                   views, ref, oth, P, y0, nrows, count, cand, cmax, rowinfo, meta, cost, smax, cflag, nlist, cnt, cb);
}

} // namespace srh
