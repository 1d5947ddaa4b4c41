// srh_dense.hip -- the tuned TwoView kernels for row-aligned epipolar geometry
// (every candidate of pixel (x,y) lies on row y of the other view: rectified rigs,
// BASELINE configs C2/C3).  Bit-identical costs to the general kernels: the same
// double operations in the same order, only organised so that the window data is
// shared through LDS and registers.
//
//   edge_planes_kernel        colour distances between 8-neighbours, once per view
//   geodesic_reg_kernel<R>    GeodesicWeight windows, window held in registers     (8(a) #2)
//   pinhole_label_table_kernel  per-label part of pointFromDepth for an undistorted pair     (#6)
//   twoview_dense_cost_kernel weighted NCC for every candidate column, LDS-tiled    (#8)
//   twoview_scan_kernel       curve walk + cost look-up + running-min WTA + depth   (#9,#10)
#include "srh_internal.hpp"
#include "srh_geom.hpp"
#include "srh_walk.hpp"

#include <algorithm>
#include <type_traits>
#include <utility>

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
#ifndef GEO_AHEAD
#define GEO_AHEAD 8                // cell steps between the request of a cell's edges and their use (geodesic_reg_kernel)
#endif
#ifndef GEO_EXPN
#define GEO_EXPN 6                 // exponentials evaluated side by side (geo_exp_n)
#endif
#define GW_ROWS 2                  // image rows per workgroup, one wave each: the waves share the staged tile (2R+GW_ROWS rows
                                   // instead of 2R+1 per wave), which is what lets two waves per SIMD fit the LDS

#ifdef SRH_PROFILE_PHASES
// diagnostic build: wave clocks of the geodesic kernel's phases (0 staging, 1 sweeps, 2 exp, 3 pconst, 4 window rows through
// the staging row and out), [5] = waves
__device__ unsigned long long g_geo_phase[8];
void geodesic_phases_fetch(unsigned long long out[8]) {
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_geo_phase), sizeof(unsigned long long)*8);
	unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	(void)hipMemcpyToSymbol(HIP_SYMBOL(g_geo_phase), z, sizeof(z));
}
#define GEO_STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); gph[i] += now_ - gph_t; gph_t = now_; }
#else
#define GEO_STAMP(i)
#endif

// The relaxations of one iteration as ONE sequence of cell steps: forward raster (geodesicweight.cpp:73-97, K1 = (-1,-1)
// (0,-1) (1,-1) (-1,0)), then backward raster (99-125, K2 = (-1,1) (0,1) (1,1) (1,0)).  std::min(weight, cost + diff): no
// operand is ever NaN (finite or +inf sums), so v_min_f64 returns exactly what the compare-and-select does.  A cell outside
// the image is never relaxed because every edge that touches it is +inf.
// The wave is alone on its SIMD (the window fills the register file): it issues one instruction of ANY kind every 4.5 - 5
// cycles and nothing hides an LDS round trip but its own arithmetic.  So a cell's four edges are requested GEO_AHEAD cell
// steps before they are used, into the registers the step just done has freed (a ring of GEO_AHEAD x 4 values instead of a
// whole window row's 44 at once: a third of the AGPR moves, no address arithmetic -- every read is `lane base + constant`),
// as single ds_read_b64 (volatile: not merged into ds_read2_b64, which takes four times the LDS-array cycles per byte, not
// hoisted out of the iteration loop, kept in program order); the scheduling barriers keep them where they are put.
// (Tried: the three relaxations from the row before done WS - 2 cells ahead of the raster's chain, so that no instruction
// follows its producer by less than three others -- a dependent FP64 instruction issues 12.6 cycles after its producer on a
// lone wave, profiles/microbench/fp64_chain_latency -- : 46.7 against 46.2 thousand cycles per wave; the sweeps are bound by
// the lone wave's issue rate, 13 instructions per cell, not by their chains.)
// Planes of the tile (PL doubles each, rows of TWD): 0 E, 1 S, 2 SE, 3 SW; `tb` = the lane's cell (window row 0, column 0).
typedef const volatile __attribute__((address_space(3))) double *GeoLds;    // (an LDS pointer by type: a volatile access through a generic pointer stays a flat load)
template <int R, int S> struct GeoCell {
	static constexpr int WS = 2*R + 1;
	static constexpr bool fwd = S < WS*WS;
	static constexpr int q = fwd ? S : S - WS*WS;
	static constexpr int yy = fwd ? q / WS : WS - 1 - q / WS;
	static constexpr int xx = fwd ? q % WS : WS - 1 - q % WS;
	// which of the four relaxations exist (an edge over the window's border is no load and no operation)
	static constexpr bool c0 = fwd ? (yy > 0 && xx > 0)      : (yy < WS - 1 && xx > 0);
	static constexpr bool c1 = fwd ? (yy > 0)                : (yy < WS - 1);
	static constexpr bool c2 = fwd ? (yy > 0 && xx < WS - 1) : (yy < WS - 1 && xx < WS - 1);
	static constexpr bool c3 = fwd ? (xx > 0)                : (xx < WS - 1);
};
// min of two doubles that are never NaN: the instruction itself (the builtin first canonicalises an operand the compiler
// cannot prove quiet -- every tap coming round the iteration loop: one more FP64 instruction per cell)
__device__ __forceinline__ double geo_min(double a, double b) {
	double r;
	asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
// exp of N numbers at once, the device library's own sequence for double (ocml exp: n = rint(x / ln 2), r = x - n ln 2 in
// two pieces, a degree-11 polynomial by Horner, ldexp, the two range selects) written STEP-MAJOR: every step for all N
// before the next step for any.  One exp is one chain of 22 dependent instructions; a lone wave issues a dependent FP64
// instruction 12.6 cycles after its producer (profiles/microbench/fp64_chain_latency) and the compiler, short of registers,
// emits the 121 exponentials of a window one after the other: 220 cycles each.  Same operations on the same constants:
// the same bits as exp() of this ROCm (checked on the device against the build before: identical depth maps and cost rows).
#define GEO_D(bits) __builtin_bit_cast(double, (unsigned long long)(bits))
template <int N, bool RANGE>
__device__ __forceinline__ void geo_exp_n(double (&x)[N]) {
	double n[N], r[N], p[N];
#define GEO_EACH(stmt) { _Pragma("unroll") for (int j = 0; j < N; ++j) { stmt; } __builtin_amdgcn_sched_barrier(0); }
	GEO_EACH(n[j] = x[j]*GEO_D(0x3ff71547652b82fe))
	GEO_EACH(n[j] = __builtin_rint(n[j]))
	GEO_EACH(r[j] = __builtin_fma(GEO_D(0xbfe62e42fefa39ef), n[j], x[j]))
	GEO_EACH(r[j] = __builtin_fma(GEO_D(0xbc7abc9e3b39803f), n[j], r[j]))
	GEO_EACH(p[j] = __builtin_fma(GEO_D(0x3e5ade156a5dcb37), r[j], GEO_D(0x3e928af3fca7ab0c)))
	GEO_EACH(p[j] = __builtin_fma(r[j], p[j], GEO_D(0x3ec71dee623fde64)))
	GEO_EACH(p[j] = __builtin_fma(r[j], p[j], GEO_D(0x3efa01997c89e6b0)))
	GEO_EACH(p[j] = __builtin_fma(r[j], p[j], GEO_D(0x3f2a01a014761f6e)))
	GEO_EACH(p[j] = __builtin_fma(r[j], p[j], GEO_D(0x3f56c16c1852b7b0)))
	GEO_EACH(p[j] = __builtin_fma(r[j], p[j], GEO_D(0x3f81111111122322)))
	GEO_EACH(p[j] = __builtin_fma(r[j], p[j], GEO_D(0x3fa55555555502a1)))
	GEO_EACH(p[j] = __builtin_fma(r[j], p[j], GEO_D(0x3fc5555555555511)))
	GEO_EACH(p[j] = __builtin_fma(r[j], p[j], GEO_D(0x3fe000000000000b)))
	GEO_EACH(p[j] = __builtin_fma(r[j], p[j], 1.0))
	GEO_EACH(p[j] = __builtin_fma(r[j], p[j], 1.0))
	GEO_EACH(p[j] = __builtin_ldexp(p[j], (int)n[j]))
	if constexpr (RANGE) {
		// (the empty statement pins the polynomial HERE: left to itself the compiler turns the selects below into branches and
		// sinks each exponential's whole chain into its own innermost branch, one after the other again)
		GEO_EACH(asm volatile("" : "+v"(p[j])))
		GEO_EACH(p[j] = x[j] < -1075.0 ? 0.0 : p[j])
		GEO_EACH(x[j] = x[j] > 1024.0 ? __builtin_inf() : p[j])
	} else {
		// x in [-2^30, 0]: the library's two selects (+inf above 1024, 0 below -1075) change nothing -- the first never fires, and
		// below -1075 ldexp has already returned 0 (n < -1550) -- and are four v_cndmask_b32 and two compares, 100 cycles of a
		// lone wave per exponential
		GEO_EACH(x[j] = p[j])
	}
}
// w[0..WS) <- exp(-w / sigma), GEO_EXPN at a time (fast: the shared-divisor quotient, also step-major)
template <int WS, int B0, int N>
__device__ __forceinline__ void geo_exp_chunk(double (&w)[WS], const SharedDivisor &sg, bool fast, double sigma) {
	double x[N];
	if (fast) {
		double m[N];
		GEO_EACH(x[j] = -w[B0 + j])
		GEO_EACH(m[j] = x[j]*sg.r)
		GEO_EACH(x[j] = __builtin_fma(-sg.b, m[j], x[j]))
		GEO_EACH(x[j] = __builtin_fma(x[j], sg.r, m[j]))
		geo_exp_n<N, false>(x);
	} else {
		GEO_EACH(x[j] = -w[B0 + j] / sigma)
		geo_exp_n<N, true>(x);
	}
	GEO_EACH(w[B0 + j] = x[j])
}
#undef GEO_EACH
template <int WS, int B0 = 0>
__device__ __forceinline__ void geo_exp_row(double (&w)[WS], const SharedDivisor &sg, bool fast, double sigma) {
	if constexpr (B0 < WS) {
		constexpr int N = WS - B0 < GEO_EXPN ? WS - B0 : ((WS - B0 > GEO_EXPN && WS - B0 < 2*GEO_EXPN) ? (WS - B0 + 1)/2 : GEO_EXPN);
		geo_exp_chunk<WS, B0, N>(w, sg, fast, sigma);
		geo_exp_row<WS, B0 + N>(w, sg, fast, sigma);
	}
}

// srh_debug_exp: geo_exp_n (both forms) beside the device library's exp(), argument by argument
__global__ void geo_exp_probe_kernel(const double *__restrict__ x, int n, double *__restrict__ kout, double *__restrict__ lout) {
	const int k = blockIdx.x*blockDim.x + threadIdx.x;
	if (k >= n) return;
	double a[1] = { x[k] }, b[1] = { x[k] };
	geo_exp_n<1, true>(a);
	if (x[k] <= 0.0 && x[k] >= -0x1p30) { geo_exp_n<1, false>(b); if (__builtin_bit_cast(unsigned long long, a[0]) != __builtin_bit_cast(unsigned long long, b[0])) a[0] = __builtin_nan(""); }   // (the form without range selects, on its range: the same bits or a NaN here)
	kout[k] = a[0];
	lout[k] = exp(x[k]);
}
void launch_geo_exp_probe(hipStream_t st, const double *x, int n, double *kout, double *lout) {
	hipLaunchKernelGGL(geo_exp_probe_kernel, dim3((unsigned)((n + 255)/256)), dim3(256), 0, st, x, n, kout, lout);
}
template <int R, int TWD, int PL, int S>
__device__ __forceinline__ void geo_edges(GeoLds tb, double &e0, double &e1, double &e2, double &e3) {
	using C = GeoCell<R, S>;
	constexpr int yy = C::yy, xx = C::xx;
	if constexpr (C::fwd) {
		if constexpr (C::c0) e0 = tb[2*PL + (yy-1)*TWD + xx - 1];
		if constexpr (C::c1) e1 = tb[1*PL + (yy-1)*TWD + xx];
		if constexpr (C::c2) e2 = tb[3*PL + (yy-1)*TWD + xx + 1];
		if constexpr (C::c3) e3 = tb[0*PL + yy*TWD + xx - 1];
	} else {
		if constexpr (C::c0) e0 = tb[3*PL + yy*TWD + xx];
		if constexpr (C::c1) e1 = tb[1*PL + yy*TWD + xx];
		if constexpr (C::c2) e2 = tb[2*PL + yy*TWD + xx];
		if constexpr (C::c3) e3 = tb[0*PL + yy*TWD + xx];
	}
}
template <int R, int TWD, int PL, int KA, int NS, int S>
__device__ __forceinline__ void geo_step(double (&w)[2*R+1][2*R+1], double (&er)[KA][4], GeoLds tb) {
	using C = GeoCell<R, S>;
	constexpr int yy = C::yy, xx = C::xx, dy = C::fwd ? -1 : 1, dx = C::fwd ? -1 : 1;
	double (&e)[4] = er[S % KA];
	double wt = w[yy][xx];
	if constexpr (C::c0) wt = geo_min(w[yy+dy][xx-1] + e[0], wt);
	if constexpr (C::c1) wt = geo_min(w[yy+dy][xx]   + e[1], wt);
	if constexpr (C::c2) wt = geo_min(w[yy+dy][xx+1] + e[2], wt);
	if constexpr (C::c3) wt = geo_min(w[yy][xx+dx]   + e[3], wt);
	w[yy][xx] = wt;
	__builtin_amdgcn_sched_barrier(0);
	// (the four requests in a row: one after each relaxation instead ran the sweeps at 56 400 cycles per wave against 48 700)
	if constexpr (S + KA < NS) {
		geo_edges<R, TWD, PL, S + KA>(tb, e[0], e[1], e[2], e[3]);
		__builtin_amdgcn_sched_barrier(0);
	}
}
template <int R, int TWD, int PL, int KA, int... S>
__device__ __forceinline__ void geo_prologue(double (&er)[KA][4], GeoLds tb, std::integer_sequence<int, S...>) {
	(geo_edges<R, TWD, PL, S>(tb, er[S][0], er[S][1], er[S][2], er[S][3]), ...);
}
template <int R, int TWD, int PL, int KA, int NS, int... S>
__device__ __forceinline__ void geo_iteration(double (&w)[2*R+1][2*R+1], double (&er)[KA][4], GeoLds tb, std::integer_sequence<int, S...>) {
	(geo_step<R, TWD, PL, KA, NS, S>(w, er, tb), ...);
}
template <int R, int TWD, int TH>
__device__ __forceinline__ void geodesic_sweeps(double (&w)[2*R+1][2*R+1], GeoLds tb, int iters) {
	constexpr int WS = 2*R + 1, NS = 2*WS*WS, KA = GEO_AHEAD < NS ? GEO_AHEAD : NS, PL = TH*TWD;
	double er[KA][4];
#pragma unroll 1
	for (int iter = 0; iter < iters; ++iter) {
		geo_prologue<R, TWD, PL, KA>(er, tb, std::make_integer_sequence<int, KA>{});
		__builtin_amdgcn_sched_barrier(0);
		geo_iteration<R, TWD, PL, KA, NS>(w, er, tb, std::make_integer_sequence<int, NS>{});
	}
}

// Per-pixel constants of the dense kernel's fast cost form from the window in registers (see the kernel): two sums over the
// 121 taps, each ONE chain of dependent additions in the reference's tap order -- on a lone wave 12.6 cycles per link if the
// links follow each other.  Software-pipelined over the taps: a tap's product is made two steps before it joins its chain, the
// second sweep's product, difference and square one step apart each, the tap values (LDS, plane 4 of the tile) requested
// GEO_AHEAD steps ahead like the sweeps' edges.  Same operations on the same operands in the same order per chain.
template <int R, int TWD, int KG, int J>
__device__ __forceinline__ void geo_pc1_step(const double (&w)[2*R+1][2*R+1], GeoLds tg, double (&g)[KG], double (&pr)[4],
                                             double &mL, double &tw, double &wmin) {
	constexpr int WS = 2*R + 1, NT = WS*WS, LAG = 2;
	if constexpr (J < NT) {
		pr[J % 4] = w[J / WS][J % WS]*g[J % KG];
		wmin = geo_min(wmin, w[J / WS][J % WS]);
	}
	if constexpr (J >= LAG) {
		mL += pr[(J - LAG) % 4];
		tw += w[(J - LAG) / WS][(J - LAG) % WS];
	}
	__builtin_amdgcn_sched_barrier(0);
	if constexpr (J + KG < NT) { g[J % KG] = tg[((J + KG) / WS)*TWD + (J + KG) % WS]; __builtin_amdgcn_sched_barrier(0); }
}
template <int R, int TWD, int KG, int J>
__device__ __forceinline__ void geo_pc2_step(const double (&w)[2*R+1][2*R+1], GeoLds tg, double (&g)[KG], double (&pr)[4],
                                             double mL, double &s2, double (&fa)[2], double &sa) {
	constexpr int WS = 2*R + 1, NT = WS*WS;
	if constexpr (J < NT) pr[J % 4] = w[J / WS][J % WS]*g[J % KG];
	// (SA of the one-pass cost form: the FUSED a_t, one step behind its making; srh_internal.hpp, SRH_PC)
	if constexpr (J < NT) fa[J % 2] = __builtin_fma(w[J / WS][J % WS], g[J % KG], -mL);
	if constexpr (J >= 1 && J - 1 < NT) sa += fa[(J - 1) % 2];
	if constexpr (J >= 1 && J - 1 < NT) pr[(J - 1) % 4] = pr[(J - 1) % 4] - mL;
	if constexpr (J >= 2 && J - 2 < NT) pr[(J - 2) % 4] = pr[(J - 2) % 4]*pr[(J - 2) % 4];
	if constexpr (J >= 3) s2 += pr[(J - 3) % 4];
	__builtin_amdgcn_sched_barrier(0);
	if constexpr (J + KG < NT) { g[J % KG] = tg[((J + KG) / WS)*TWD + (J + KG) % WS]; __builtin_amdgcn_sched_barrier(0); }
}
template <int R, int TWD, int KG, int... J>
__device__ __forceinline__ void geo_pc_fill(GeoLds tg, double (&g)[KG], std::integer_sequence<int, J...>) {
	((g[J] = tg[(J / (2*R+1))*TWD + J % (2*R+1)]), ...);
	__builtin_amdgcn_sched_barrier(0);
}
template <int R, int TWD, int KG, int... J>
__device__ __forceinline__ void geo_pc1(const double (&w)[2*R+1][2*R+1], GeoLds tg, double (&g)[KG], double &mL, double &tw, double &wmin,
                                        std::integer_sequence<int, J...>) {
	double pr[4];
	(geo_pc1_step<R, TWD, KG, J>(w, tg, g, pr, mL, tw, wmin), ...);
}
template <int R, int TWD, int KG, int... J>
__device__ __forceinline__ void geo_pc2(const double (&w)[2*R+1][2*R+1], GeoLds tg, double (&g)[KG], double mL, double &s2, double &sa,
                                        std::integer_sequence<int, J...>) {
	double pr[4], fa[2];
	(geo_pc2_step<R, TWD, KG, J>(w, tg, g, pr, mL, s2, fa, sa), ...);
}

template <int R, bool WIMG>
__global__ __launch_bounds__(GW_TW*GW_ROWS)
void geodesic_reg_kernel(const ViewDev *__restrict__ views, int ref, const double *__restrict__ edges,
                         srh_params P, int y0, int nrows, double *__restrict__ wbuf, size_t wstride,
                         double *__restrict__ pconst)
{
	constexpr int WS = 2*R + 1;
	constexpr int TWD = GW_TW + 2*R;            // tile width
	const ViewDev &V = views[ref];
	const int W = V.w, H = V.h;
	const size_t n = (size_t)W*H;
	constexpr int TH = WS + GW_ROWS - 1;       // tile height
	constexpr int NT = GW_TW*GW_ROWS;          // threads
	const int tiles_per_row = (W + GW_TW - 1)/GW_TW;
	const int tgrp = blockIdx.x / tiles_per_row;                  // group of GW_ROWS rows of the band
	const int x0 = (blockIdx.x % tiles_per_row)*GW_TW;
	const int wv = threadIdx.x / GW_TW;                           // the wave = its row within the group
	const int trow = tgrp*GW_ROWS + wv;
	const int cy0 = y0 + tgrp*GW_ROWS;                            // first row of the group
	const int cy = cy0 + wv;
	const bool rowok = trow < nrows;
#ifdef SRH_PROFILE_PHASES
	unsigned long long gph_t = __builtin_amdgcn_s_memtime(), gph[7] = {0, 0, 0, 0, 0, 0, 0};
#endif

	// planes of the staged tile: 0 E, 1 S, 2 SE, 3 SW edges, 4 the view's TwoView tap values (NaN = unusable) for pconst.
	// ONE array: every read of the sweeps is `lane base + a constant`, the constant in the instruction's offset field
	__shared__ double tile[5][TH][TWD];
	const double inf = __builtin_inf();
	// A workgroup without a masked-in pixel has nothing to compute or store and leaves before it stages its 35 KB tile
	// (MultiViewStereo's views are mostly mask: four fifths of the workgroups of C4).  The dense TwoView path (WIMG) asks the
	// same question AFTER it has requested its tile: its images are mostly masked in, and the lone wave would sit through one
	// more memory round trip before the first request of the tile.
	const int cxm = x0 + (int)(threadIdx.x % GW_TW);
	if constexpr (!WIMG) {
		const bool act_ = rowok && cxm < W && V.mask[(size_t)cy*W + cxm] == 1;
		if (!__syncthreads_or(act_)) return;
	}
	{
		// all global loads of the thread first, LDS stores after: one memory latency per tile
		constexpr int NB = (TH*TWD + NT - 1)/NT;
		double t0[NB], t1[NB], t2[NB], t3[NB], t4[NB];
		const uint8_t mk = (WIMG && rowok && cxm < W) ? V.mask[(size_t)cy*W + cxm] : (uint8_t)0;
#pragma unroll
		for (int k = 0; k < NB; ++k) {
			const int idx = threadIdx.x + k*NT;
			const int ty = idx / TWD, tx = idx % TWD;
			const int gx = x0 - R + tx, gy = cy0 - R + ty;
			const bool in = idx < TH*TWD && gx >= 0 && gy >= 0 && gx < W && gy < H;
			const size_t gi = in ? (size_t)gy*W + gx : 0;
			t0[k] = in ? edges[0*n + gi] : inf;
			t1[k] = in ? edges[1*n + gi] : inf;
			t2[k] = in ? edges[2*n + gi] : inf;
			t3[k] = in ? edges[3*n + gi] : inf;
			t4[k] = (pconst && in) ? V.gray_tv[gi] : __builtin_nan("");
		}
		if constexpr (WIMG) {
			if (!__syncthreads_or(mk == 1)) return;
		}
#pragma unroll
		for (int k = 0; k < NB; ++k) {
			const int idx = threadIdx.x + k*NT;
			if (idx < TH*TWD) {
				const int ty = idx / TWD, tx = idx % TWD;
				tile[0][ty][tx] = t0[k]; tile[1][ty][tx] = t1[k]; tile[2][ty][tx] = t2[k]; tile[3][ty][tx] = t3[k];
				tile[4][ty][tx] = t4[k];
			}
		}
	}
	__syncthreads();

	const int i = threadIdx.x % GW_TW;
	const int cx = x0 + i;
	// masked pixels never reach init_weights (twoviewstereo.cpp:268-272): their lanes sit out, their windows stay unwritten
	// (a wave past the band's last row runs along for the barriers and stores nothing)
	const bool active = rowok && cx < W && V.mask[(size_t)(rowok ? cy : y0)*W + (cx < W ? cx : 0)] == 1;
	const unsigned long long amask = WIMG ? __ballot(active) : 0ull;
	// (WIMG: every lane runs the sweeps -- the stores at the end are a joint effort of the wave -- on the staged edges,
	// which are finite or +inf for any tile cell; only lanes of masked-in pixels store)
	if (!WIMG && !active) return;

	GEO_STAMP(0)
	double w[WS][WS];
#pragma unroll
	for (int a = 0; a < WS; ++a)
#pragma unroll
		for (int b = 0; b < WS; ++b) w[a][b] = P.geodesic_init;
	w[R][R] = 0.0;

	geodesic_sweeps<R, TWD, TH>(w, (GeoLds)&tile[0][wv][i], P.geodesic_iters);
	GEO_STAMP(1)
	// exponential weighting (geodesicweight.cpp:128-130): exp(-w / sigma), 121 quotients by one divisor.  Every w is 0 or a sum
	// of edges that are 0 or >= 1 (square roots of integers), capped by the initial value: with sigma and the initial value far
	// from the exponent limits every quotient takes the shared-divisor form (srh_walk.hpp: the same bits; a zero comes out as
	// +0 where the division gives -0, and exp of either is 1); with sigma > 0 and initial value / sigma < 2^30 the exponential's
	// argument lies in [-2^30, 0], where its range selects are idle (geo_exp_n)
	const SharedDivisor sg = shared_divisor(P.geodesic_sigma);
	const bool sdiv = sg.ok && P.geodesic_sigma > 0 && P.geodesic_init > 0x1p-300 && P.geodesic_init < 0x1p300 && P.geodesic_init < 0x1p30*P.geodesic_sigma;
	if constexpr (WIMG) {
		// the strip kernel's LDS-image layout [tile][row][pixel][WP]: the 32 pixels' taps of one window row are 3 KB of
		// contiguous bytes.  Stored as they stand (a lane's 96 bytes, 16 at a time) every store instruction touches 64
		// cache lines (measured: +25 % kernel time); so each window row goes through LDS -- a 6 KB staging row --
		// and leaves as whole kilobytes.  Each wave has its own staging row and a wave's LDS operations execute in program
		// order, so nothing has to be waited for: the fences below only keep the compiler from reordering.  (They were
		// workgroup barriers, which also wait for the wave's global stores of the row before: -2 % of the kernel.)
		constexpr int WP = (WS + 1) & ~1;
		static_assert(GW_TW == 2*SRH_WTILE, "a wave covers two window-buffer tiles");
		__shared__ __align__(16) double stage_buf[GW_ROWS][GW_TW*WP];
		double *stage = stage_buf[wv];
		double *wt = wbuf + wimg_offset(W, R, rowok ? trow : 0, x0);   // first of the two window-buffer tiles of this wave's row
		constexpr size_t TILE_D = (size_t)SRH_WTILE*WS*WP;   // doubles per tile
		// 16-byte pieces of a staged row: piece q = doubles 2q, 2q+1 of pixel 2q / WP.  Which pieces this lane stores (its
		// pixel masked in) and where they go is the same for every window row: made once
		constexpr int NPIECE = GW_TW*WP/2, NK = (NPIECE + GW_TW - 1)/GW_TW;
		bool st[NK];
		unsigned goff[NK];
#pragma unroll
		for (int k = 0; k < NK; ++k) {
			const int q = i + k*GW_TW;
			const int pix = (2*q)/WP;
			const int half = pix/SRH_WTILE;             // which of the two tiles
			st[k] = q < NPIECE && ((amask >> pix) & 1ull);
			goff[k] = (unsigned)(half*(int)TILE_D + (2*q - half*SRH_WTILE*WP));
		}
#pragma unroll
		for (int a = 0; a < WS; ++a) {
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
			geo_exp_row<WS>(w[a], sg, sdiv, P.geodesic_sigma);
			GEO_STAMP(2)
#pragma unroll
			for (int b = 0; b + 1 < WS; b += 2) {
				double2 v; v.x = w[a][b]; v.y = w[a][b + 1];
				*reinterpret_cast<double2 *>(stage + i*WP + b) = v;
			}
			stage[i*WP + WS - 1] = w[a][WS - 1];
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
			// all pieces read back before the first leaves (one LDS round trip per row, not one per piece: the lone wave has
			// nothing else to do meanwhile; the empty statement keeps the compiler from sinking each read into its store's branch)
			double2 v[NK];
#pragma unroll
			for (int k = 0; k < NK; ++k) v[k] = *reinterpret_cast<const double2 *>(stage + 2*((i + k*GW_TW) < NPIECE ? (i + k*GW_TW) : 0));
#pragma unroll
			for (int k = 0; k < NK; ++k) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y));
			double *wrow = wt + (size_t)a*(SRH_WTILE*WP);
#pragma unroll
			for (int k = 0; k < NK; ++k)
				if (st[k]) { typedef double d2v __attribute__((ext_vector_type(2))); d2v t; t.x = v[k].x; t.y = v[k].y; __builtin_nontemporal_store(t, reinterpret_cast<d2v *>(wrow + goff[k])); }
			GEO_STAMP(4)
		}
	} else {
		double *wb = wbuf + wbuf_offset(W, WS*WS, trow, cx);         // (active lanes only: inside the band)
#pragma unroll
		for (int a = 0; a < WS; ++a) {
			geo_exp_row<WS>(w[a], sg, sdiv, P.geodesic_sigma);
#pragma unroll
			for (int b = 0; b < WS; ++b) __builtin_nontemporal_store(w[a][b], &wb[(size_t)(a*WS + b)*wstride]);
			asm volatile("" ::: "memory");
		}
	}
	GEO_STAMP(2)
	if (pconst && active) {
		// Per-pixel constants of the dense kernel's fast cost form, while the window is in registers: when every tap
		// is usable (gray value valid, weight above the cut-off), meanL, totalWeight and sum2 of
		// twoviewstereo.cpp:917-976 do not depend on the candidate.  Same tap order, same operations.
		constexpr int NT = WS*WS, KG = GEO_AHEAD < NT ? GEO_AHEAD : NT;
		const GeoLds tg = (GeoLds)&tile[4][wv][i];
		double g[KG];
		// every tap usable = every gray value valid (finite; NaN otherwise, and one NaN makes the sum meanL NaN: the weights are
		// finite) and every weight above the cut-off (the smallest is): one v_min_f64 per tap instead of two compares and the
		// mask arithmetic (which the compiler kept as 121 lane masks parked in VGPR lanes until the end)
		double mL = 0, tw = 0, wmin = __builtin_inf();
		geo_pc_fill<R, TWD, KG>(tg, g, std::make_integer_sequence<int, KG>{});
		geo_pc1<R, TWD, KG>(w, tg, g, mL, tw, wmin, std::make_integer_sequence<int, NT + 2>{});
		bool all = mL == mL && wmin > P.weight_cutoff;
		GEO_STAMP(5)
		double s2 = 0, sa = 0;
		if (all && !(tw < 1e-10)) {
			mL /= tw;
			geo_pc_fill<R, TWD, KG>(tg, g, std::make_integer_sequence<int, KG>{});
			geo_pc2<R, TWD, KG>(w, tg, g, mL, s2, sa, std::make_integer_sequence<int, NT + 3>{});
			GEO_STAMP(6)
		} else all = false;
		double *pc = pconst + ((size_t)trow*W + cx)*SRH_PC;
		pc[0] = mL; pc[1] = tw; pc[2] = s2; pc[3] = all ? 1.0/tw : 0.0;   // all taps usable: != 0, and then 1/totalWeight (the one-pass form multiplies by it)
		pc[4] = sa; pc[5] = 0.0;
	}
#ifdef SRH_PROFILE_PHASES
	GEO_STAMP(3)
	if (i == 0 && rowok) {
		for (int k = 0; k < 7; ++k) atomicAdd(&g_geo_phase[k], gph[k]);
		atomicAdd(&g_geo_phase[7], 1ull);
	}
#endif
}
#undef GEO_STAMP

// ------------------------------------------------------------------ geodesic windows, the tile by LDS-DMA (dense TwoView path, r = 5)
// The kernel above stages its tile through registers at its start, and a wave alone on its SIMD sits through that memory
// round trip with nothing else to run: 14 % of its life.  Here a workgroup is PERSISTENT (one per compute unit: four waves =
// four consecutive image rows of 64 columns, one tile of 14 x 74 cells x 5 planes = 41 KB) and keeps TWO tiles in LDS: while it
// works on one, the next arrives by LDS-DMA (global_load_lds, 16 bytes per lane, no registers), requested right after the
// barrier that hands the buffer over.  LDS-DMA copies bytes, it cannot put +inf where the image ends: the four edge planes and
// the tap plane are kept a second time with their borders written out (geo5: +inf / NaN margins of GD_PAD* cells, rows of even
// length starting 16-byte aligned at every tile's first column), made once per uploaded view.  The arithmetic is the kernel's
// above, statement for statement (the same helpers); four rows per tile also fetch 14 tile rows per 4 image rows instead of
// 12 per 2.
#define GD_ROWS 4
#define GD_PADL 5                  // = R of the one instantiation: column x0 - R of a tile is padded column x0, x0 a multiple of 64
#define GD_PADR 70
#define GD_PADT 5
#define GD_PADB (5 + GD_ROWS - 1)
static inline __host__ __device__ int geo5_stride(int w) { return (w + GD_PADL + GD_PADR + 1) & ~1; }
static inline __host__ __device__ int geo5_rows(int h) { return h + GD_PADT + GD_PADB; }
// (behind the five planes: the mask bytes, rows of a multiple of four bytes reaching 63 columns and 3 rows past the image,
// zero there -- a tile's 4 x 64 mask bytes are ONE dword request of 16 lanes per wave)
static inline __host__ __device__ int geo5_mstride(int w) { return (w + 64 + 3) & ~3; }
static inline __host__ __device__ size_t geo5_plane_doubles(int w, int h) { return (size_t)5*geo5_stride(w)*geo5_rows(h); }
size_t geo5_doubles(int w, int h) { return geo5_plane_doubles(w, h) + ((size_t)geo5_mstride(w)*(h + 4) + 7)/8; }

__global__ void geo5_planes_kernel(const double *__restrict__ edges, const double *__restrict__ gray_tv,
                                   const uint8_t *__restrict__ mask, int W, int H, double *__restrict__ out)
{
	{
		uint8_t *mo = (uint8_t *)(out + geo5_plane_doubles(W, H));
		const int MS = geo5_mstride(W);
		const size_t mt = (size_t)MS*(H + 4);
		for (size_t q = (size_t)blockIdx.x*blockDim.x + threadIdx.x; q < mt; q += (size_t)gridDim.x*blockDim.x) {
			const int y = (int)(q / MS), x = (int)(q % MS);
			mo[q] = (x < W && y < H) ? mask[(size_t)y*W + x] : (uint8_t)0;
		}
	}
	const int GS = geo5_stride(W), GR = geo5_rows(H);
	const size_t n = (size_t)W*H, pl = (size_t)GS*GR, tot = 5*pl;
	for (size_t q = (size_t)blockIdx.x*blockDim.x + threadIdx.x; q < tot; q += (size_t)gridDim.x*blockDim.x) {
		const int k = (int)(q / pl);
		const size_t r = q - (size_t)k*pl;
		const int y = (int)(r / GS) - GD_PADT, x = (int)(r % GS) - GD_PADL;
		const bool in = x >= 0 && y >= 0 && x < W && y < H;
		out[q] = k < 4 ? (in ? edges[(size_t)k*n + (size_t)y*W + x] : __builtin_inf())
		               : (in ? gray_tv[(size_t)y*W + x] : __builtin_nan(""));
	}
}
void launch_geo5_planes(hipStream_t st, const double *edges, const double *gray_tv, const uint8_t *mask, int w, int h, double *out) {
	hipLaunchKernelGGL(geo5_planes_kernel, dim3(4096), dim3(256), 0, st, edges, gray_tv, mask, w, h, out);
}

template <int R>
__global__ __launch_bounds__(GW_TW*GD_ROWS)
void geodesic_dma_kernel(const ViewDev *__restrict__ views, int ref, const double *__restrict__ geo5,
                         srh_params P, int y0, int nrows, double *__restrict__ wbuf, double *__restrict__ pconst)
{
	static_assert(R == GD_PADL, "the padded planes are laid out for this radius");
	constexpr int WS = 2*R + 1, TWD = GW_TW + 2*R, TH = WS + GD_ROWS - 1, PL = TH*TWD;
	constexpr int TILE_BYTES = 5*PL*8, NPC = (TILE_BYTES + 1023)/1024;
	constexpr int WP = (WS + 1) & ~1;
	static_assert((TWD*8) % 16 == 0 && GW_TW == 2*SRH_WTILE, "tile rows are whole 16-byte pieces; a wave covers two window-buffer tiles");
	const ViewDev &V = views[ref];
	const int W = V.w, H = V.h;
	const int GS = geo5_stride(W);
	const size_t GPL = (size_t)GS*geo5_rows(H);
	const int tiles_per_row = (W + GW_TW - 1)/GW_TW;
	const int total_tiles = tiles_per_row*((nrows + GD_ROWS - 1)/GD_ROWS);
	const int wv = threadIdx.x / GW_TW, i = threadIdx.x % GW_TW;
	__shared__ __align__(16) double tile[2][5][TH][TWD];
	__shared__ __align__(16) double stage_buf[GD_ROWS][GW_TW*WP];
	__shared__ __align__(16) unsigned mtile[2][GD_ROWS][GW_TW/4];   // the tile's mask bytes, row by row
	double *stage = stage_buf[wv];
	typedef __attribute__((address_space(3))) void lvoid;
	const int MS = geo5_mstride(W);
	const char *gm5 = (const char *)(geo5 + geo5_plane_doubles(W, H));

	// the tile of index tl into buffer `buf`: piece p (1 KB of the tile's LDS image) by wave p % 4
	auto request = [&](int tl, int buf) {
		const int tg = tl / tiles_per_row, xa = (tl % tiles_per_row)*GW_TW, ya = y0 + tg*GD_ROWS;
		const char *org = (const char *)geo5 + ((size_t)(ya - R + GD_PADT)*GS + (size_t)(xa - R + GD_PADL))*8;
#pragma unroll
		for (int p = wv; p < NPC; p += GD_ROWS) {
			const int o = p*1024 + i*16;
			if (o < TILE_BYTES) {
				const int plane = o / (PL*8), rem = o - plane*(PL*8), row = rem / (TWD*8), colb = rem - row*(TWD*8);
				// (as an assembler statement: the builtin tells the compiler that LDS is being written behind its back, and it
				// then parks the wave at a vmcnt(0) before the first LDS read that might alias -- the sweeps' volatile reads of
				// the OTHER buffer, a few instructions after the request; the waits this kernel needs are written out below)
				const char *src = org + ((size_t)plane*GPL + (size_t)row*GS)*8 + colb;
				const unsigned dst = (unsigned)(size_t)(lvoid *)((char *)&tile[buf][0][0][0] + p*1024);
				asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(__builtin_amdgcn_readfirstlane(dst)) : "m0", "memory");
			}
		}
		// the wave's own row of mask bytes (no register is loaded for the next tile: a loaded register is a value the compiler
		// may move -- behind a wait of its own -- long before this kernel's wait)
		if (i < GW_TW/4) {
			const char *src = gm5 + (size_t)(ya + wv)*MS + xa + 4*i;
			const unsigned dst = (unsigned)(size_t)(lvoid *)&mtile[buf][wv][0];
			asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(src), "s"(__builtin_amdgcn_readfirstlane(dst)) : "m0", "memory");
		}
	};
	int tl = blockIdx.x;
	if (tl >= total_tiles) return;
	request(tl, 0);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the first tile: the one round trip nothing hides
	const SharedDivisor sg = shared_divisor(P.geodesic_sigma);
	const bool sdiv = sg.ok && P.geodesic_sigma > 0 && P.geodesic_init > 0x1p-300 && P.geodesic_init < 0x1p300 && P.geodesic_init < 0x1p30*P.geodesic_sigma;
#pragma unroll 1
	for (int kt = 0; tl < total_tiles; ++kt, tl += gridDim.x) {
		const int buf = kt & 1;
		const int tgrp = tl / tiles_per_row, x0 = (tl % tiles_per_row)*GW_TW;
		const int trow = tgrp*GD_ROWS + wv;
		const bool rowok = trow < nrows;
		const int cx = x0 + i;
		// this tile has landed (every wave waited for its own requests before it came here: after the sweeps of the tile before;
		// the barrier collects them), and everybody has left the tile of the round before: its buffer is free for the tile after
		// this one
		__syncthreads();
		const bool active = rowok && cx < W && ((const volatile __attribute__((address_space(3))) uint8_t *)&mtile[buf][wv][0])[i] == 1;   // (the wave's own request)
		if (tl + (int)gridDim.x < total_tiles) request(tl + gridDim.x, buf ^ 1);
		const unsigned long long amask = __ballot(active);
		// (a wave without a masked-in pixel in its row has nothing to compute or store; it still takes its share of the requests)
		if (amask == 0ull) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); continue; }

		double w[WS][WS];
#pragma unroll
		for (int a = 0; a < WS; ++a)
#pragma unroll
			for (int b = 0; b < WS; ++b) w[a][b] = P.geodesic_init;
		w[R][R] = 0.0;
		geodesic_sweeps<R, TWD, TH>(w, (GeoLds)&tile[buf][0][wv][i], P.geodesic_iters);
		// the next tile has been on its way for the length of the sweeps: the wait is free HERE -- at the
		// loop's top it would also sit out this tile's own window stores
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

		// exponential weighting, rows out (see geodesic_reg_kernel)
		{
			double *wt = wbuf + wimg_offset(W, R, rowok ? trow : 0, x0);
			constexpr size_t TILE_D = (size_t)SRH_WTILE*WS*WP;
			constexpr int NPIECE = GW_TW*WP/2, NK = (NPIECE + GW_TW - 1)/GW_TW;
			bool st[NK];
			unsigned goff[NK];
#pragma unroll
			for (int k = 0; k < NK; ++k) {
				const int q = i + k*GW_TW;
				const int pix = (2*q)/WP;
				const int half = pix/SRH_WTILE;
				st[k] = q < NPIECE && ((amask >> pix) & 1ull);
				goff[k] = (unsigned)(half*(int)TILE_D + (2*q - half*SRH_WTILE*WP));
			}
#pragma unroll
			for (int a = 0; a < WS; ++a) {
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				__builtin_amdgcn_wave_barrier();
				geo_exp_row<WS>(w[a], sg, sdiv, P.geodesic_sigma);
#pragma unroll
				for (int b = 0; b + 1 < WS; b += 2) {
					double2 v; v.x = w[a][b]; v.y = w[a][b + 1];
					*reinterpret_cast<double2 *>(stage + i*WP + b) = v;
				}
				stage[i*WP + WS - 1] = w[a][WS - 1];
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				__builtin_amdgcn_wave_barrier();
				double2 v[NK];
#pragma unroll
				for (int k = 0; k < NK; ++k) v[k] = *reinterpret_cast<const double2 *>(stage + 2*((i + k*GW_TW) < NPIECE ? (i + k*GW_TW) : 0));
#pragma unroll
				for (int k = 0; k < NK; ++k) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y));
				double *wrow = wt + (size_t)a*(SRH_WTILE*WP);
#pragma unroll
				for (int k = 0; k < NK; ++k)
					if (st[k]) { typedef double d2v __attribute__((ext_vector_type(2))); d2v t; t.x = v[k].x; t.y = v[k].y; __builtin_nontemporal_store(t, reinterpret_cast<d2v *>(wrow + goff[k])); }
			}
		}
		if (pconst && active) {
			constexpr int NT = WS*WS, KG = GEO_AHEAD < NT ? GEO_AHEAD : NT;
			const GeoLds tg = (GeoLds)&tile[buf][4][wv][i];
			double g[KG];
			double mL = 0, tw = 0, wmin = __builtin_inf();
			geo_pc_fill<R, TWD, KG>(tg, g, std::make_integer_sequence<int, KG>{});
			geo_pc1<R, TWD, KG>(w, tg, g, mL, tw, wmin, std::make_integer_sequence<int, NT + 2>{});
			bool all = mL == mL && wmin > P.weight_cutoff;
			double s2 = 0, sa = 0;
			if (all && !(tw < 1e-10)) {
				mL /= tw;
				geo_pc_fill<R, TWD, KG>(tg, g, std::make_integer_sequence<int, KG>{});
				geo_pc2<R, TWD, KG>(w, tg, g, mL, s2, sa, std::make_integer_sequence<int, NT + 3>{});
			} else all = false;
			double *pc = pconst + ((size_t)trow*W + cx)*SRH_PC;
			pc[0] = mL; pc[1] = tw; pc[2] = s2; pc[3] = all ? 1.0/tw : 0.0;   // all taps usable: != 0, and then 1/totalWeight (the one-pass form multiplies by it)
			pc[4] = sa; pc[5] = 0.0;
		}
	}
}

// windows of rows [y0, y0 + nrows) in the strip kernel's layout from the padded planes `geo5` (launch_geo5_planes)
bool launch_geodesic_dma(hipStream_t st, const ViewDev *views, int ref, int width, const double *geo5, const srh_params &P,
                         int y0, int nrows, double *wbuf, double *pconst, int num_cus)
{
	if (P.window_radius != GD_PADL) return false;
	const int tiles = (width + GW_TW - 1)/GW_TW, total = tiles*((nrows + GD_ROWS - 1)/GD_ROWS);
	const int grid = total < num_cus ? total : num_cus;
	hipLaunchKernelGGL((geodesic_dma_kernel<GD_PADL>), dim3((unsigned)grid), dim3(GW_TW*GD_ROWS), 0, st, views, ref, geo5, P, y0, nrows, wbuf, pconst);
	return true;
}

// ------------------------------------------------------------------ AdaptiveWeight windows (adaptiveweight.cpp:33-79)
// One thread per pixel, one wave = 64 consecutive pixels of a row; the colour tile and the tap plane of the wave in
// LDS; weights leave tile-major (wb[tap*32 + pixel]: 256-byte segments per wave store), the per-pixel constants of
// the dense kernel's fast form follow on the way (pconst, as in geodesic_reg_kernel).  Same operations as weights_kernel's adaptive branch -- dw[|row|]*dw[|col|]
// with dw[k] = exp(-k/radius) taken from a table of R+1 values instead of being recomputed per tap, the colour
// distance, exp(-diff/sigma), NaN -> 0 -- so the same bits; that kernel issued ~130 instructions per tap (two exp,
// a square root, a division) and then read its window back from memory twice for the constants.
#define AW_TW 64
template <int R>
__global__ __launch_bounds__(AW_TW)
void adaptive_reg_kernel(const ViewDev *__restrict__ views, int ref, srh_params P, int y0, int nrows,
                         double *__restrict__ wbuf, size_t wstride, double *__restrict__ pconst, int wimg)
{
	constexpr int WS = 2*R + 1, TWD = AW_TW + 2*R;
	const ViewDev &V = views[ref];
	const int W = V.w, H = V.h;
	const int tiles = (W + AW_TW - 1)/AW_TW;
	const int trow = blockIdx.x / tiles, x0 = (blockIdx.x % tiles)*AW_TW;
	const int cy = y0 + trow;
	const int i = threadIdx.x, cx = x0 + i;
	__shared__ uint32_t ct[WS][TWD];                             // rgba of the tile; pixels outside the image: never looked at (inb)
	__shared__ double gt[WS][TWD];                               // TwoView tap values (NaN = unusable), for pconst
	for (int k = i; k < WS*TWD; k += AW_TW) {
		const int ty = k / TWD, tx = k % TWD;
		const int gx = x0 - R + tx, gy = cy - R + ty;
		const bool in = gx >= 0 && gy >= 0 && gx < W && gy < H;
		ct[ty][tx] = in ? V.rgba[(size_t)gy*W + gx] : 0u;
		gt[ty][tx] = (pconst && in) ? V.gray_tv[(size_t)gy*W + gx] : __builtin_nan("");
	}
	__syncthreads();
	if (cx >= W || V.mask[(size_t)cy*W + cx] != 1) return;       // masked pixels never reach init_weights
	double dw[R + 1];
#pragma unroll
	for (int k = 0; k <= R; ++k) dw[k] = exp(-k / (1.0*R));
	const uint32_t crgb = ct[R][i + R];
	// (wimg, round 6: the strip / wave-tile paths' LDS-image layout -- tap (a, b) at a*wimg_row_stride + b -- else tile-major)
	double *wb = wbuf + (wimg ? wimg_offset(W, R, trow, cx) : wbuf_offset(W, WS*WS, trow, cx));
	const size_t wra = wimg ? (size_t)wimg_row_stride(R) : (size_t)WS*wstride, wcb = wimg ? 1 : wstride;
	// the window is NOT kept in registers (121 doubles would leave one wave per SIMD alone with the exp / sqrt / division
	// chains): weights stream out as they are computed, meanL / totalWeight accumulate on the way, and the second sweep of
	// the constants reads the weights back (the wave's own 256-byte segments, just written)
	bool all = true;
	double mL = 0, tw = 0;
#pragma unroll 1
	for (int a = 0; a < WS; ++a) {
		const int py = cy - R + a;
		const double dwr = dw[a < R ? R - a : a - R];
#pragma unroll
		for (int b = 0; b < WS; ++b) {
			const int px = cx - R + b;
			double weight = 0.0;
			if (!(px < 0 || py < 0 || px >= W || py >= H)) {
				const uint32_t q = ct[a][i + b];
				const double dr = (double)((int)(q & 255u) - (int)(crgb & 255u));
				const double dg = (double)((int)((q >> 8) & 255u) - (int)((crgb >> 8) & 255u));
				const double db = (double)((int)((q >> 16) & 255u) - (int)((crgb >> 16) & 255u));
				const double diff = sqrt(dr*dr + dg*dg + db*db);           // (color_dist: exact small integers)
				const double w1 = dwr*dw[b < R ? R - b : b - R];
				const double w2 = exp(-diff / P.adaptive_color_sigma);
				weight = w1*w2;
				if (weight != weight) weight = 0.0;
			}
			wb[(size_t)a*wra + (size_t)b*wcb] = weight;
			if (pconst) {
				const double gl = gt[a][i + b];
				if (!(gl == gl && weight > P.weight_cutoff)) all = false;
				mL += weight*gl;
				tw += weight;
			}
		}
	}
	if (pconst) {
		double s2 = 0, sa = 0;
		if (all && !(tw < 1e-10)) {
			mL /= tw;
#pragma unroll 1
			for (int a = 0; a < WS; ++a) {
				double wr[WS];
#pragma unroll
				for (int b = 0; b < WS; ++b) wr[b] = wb[(size_t)a*wra + (size_t)b*wcb];
#pragma unroll
				for (int b = 0; b < WS; ++b) { const double t = wr[b]*gt[a][i + b] - mL; s2 += t*t; sa += __builtin_fma(wr[b], gt[a][i + b], -mL); }   // (sa: fused terms, SRH_PC)
			}
		} else all = false;
		double *pc = pconst + ((size_t)trow*W + cx)*SRH_PC;
		pc[0] = mL; pc[1] = tw; pc[2] = s2; pc[3] = all ? 1.0/tw : 0.0;   // all taps usable: != 0, and then 1/totalWeight (the one-pass form multiplies by it)
		pc[4] = sa; pc[5] = 0.0;
	}
}

// radii 5 and 2, either band layout (other radii stay on weights_kernel)
bool launch_adaptive_reg(hipStream_t st, const ViewDev *views, int ref, int width, const srh_params &P, int y0, int nrows,
                         double *wbuf, size_t wstride, double *pconst, bool wimg)
{
	if (wstride != SRH_WTILE) return false;
	const dim3 grid((unsigned)(((width + AW_TW - 1)/AW_TW)*nrows)), block(AW_TW);
	if (P.window_radius == 5) hipLaunchKernelGGL(adaptive_reg_kernel<5>, grid, block, 0, st, views, ref, P, y0, nrows, wbuf, wstride, pconst, wimg ? 1 : 0);
	else if (P.window_radius == 2) hipLaunchKernelGGL(adaptive_reg_kernel<2>, grid, block, 0, st, views, ref, P, y0, nrows, wbuf, wstride, pconst, wimg ? 1 : 0);
	else return false;
	return true;
}

bool launch_geodesic_reg(hipStream_t st, const ViewDev *views, int ref, int width, const double *edges,
                         const srh_params &P, int y0, int nrows, double *wbuf, size_t wstride, double *pconst, bool wimg)
{
	const int tiles = (width + GW_TW - 1)/GW_TW;
	const dim3 grid((unsigned)(tiles*((nrows + GW_ROWS - 1)/GW_ROWS))), block(GW_TW*GW_ROWS);
	const int wi = wimg ? 1 : 0;
	switch (P.window_radius) {
	case 5:
		if (wi) hipLaunchKernelGGL((geodesic_reg_kernel<5, true>), grid, block, 0, st, views, ref, edges, P, y0, nrows, wbuf, wstride, pconst);
		else    hipLaunchKernelGGL((geodesic_reg_kernel<5, false>), grid, block, 0, st, views, ref, edges, P, y0, nrows, wbuf, wstride, pconst);
		return true;
	case 2:
		if (wi) hipLaunchKernelGGL((geodesic_reg_kernel<2, true>), grid, block, 0, st, views, ref, edges, P, y0, nrows, wbuf, wstride, pconst);
		else    hipLaunchKernelGGL((geodesic_reg_kernel<2, false>), grid, block, 0, st, views, ref, edges, P, y0, nrows, wbuf, wstride, pconst);
		return true;
	default: return false;
	}
}

// ------------------------------------------------------------------ per-label table of the pinhole walk
__global__ void pinhole_label_table_kernel(const ViewDev *__restrict__ views, int ref, srh_params P, int mvs,
                                           double *__restrict__ tnum)
{
	const int d = blockIdx.x*blockDim.x + threadIdx.x;
	if (d < P.num_depth_levels) tnum[d] = pinhole_label_tnum(views[ref].cam, P, mvs != 0, d);
}

// per-label plane distance of pointFromDepth for the general walker (any camera model)
__global__ void label_plane_table_kernel(const ViewDev *__restrict__ views, int ref, srh_params P, int mvs,
                                         double *__restrict__ tdist)
{
	const int d = blockIdx.x*blockDim.x + threadIdx.x;
	if (d < P.num_depth_levels) tdist[d] = label_plane_dist(views[ref].cam, P, mvs != 0, d);
}

void launch_label_plane_table(hipStream_t st, const ViewDev *views, int ref, const srh_params &P, bool mvs, double *tdist) {
	hipLaunchKernelGGL(label_plane_table_kernel, dim3((unsigned)((P.num_depth_levels + 63)/64)), dim3(64), 0, st,
	                   views, ref, P, mvs ? 1 : 0, tdist);
}

void launch_pinhole_label_table(hipStream_t st, const ViewDev *views, int ref, const srh_params &P, bool mvs, double *tnum) {
	hipLaunchKernelGGL(pinhole_label_table_kernel, dim3((unsigned)((P.num_depth_levels + 63)/64)), dim3(64), 0, st,
	                   views, ref, P, mvs ? 1 : 0, tnum);
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
#define DC_THREADS (DC_TP*DC_G)

// LDS image of one workgroup.  Everything a lane reads in the hot loop is 16-byte
// aligned so that it moves as ds_read_b128: window rows are padded to an even tap
// count, candidate blocks start on even tile columns.  Lanes of a wave are laid out
// pixel-fastest (lane = g*8 + pixel), so one wave-instruction touches 64
// consecutive doubles of a right-image row: bank-conflict free.
template <int R, int DC_NCB, int DC_CHUNK>
struct DenseSmem {
	static constexpr int WS = 2*R + 1;
	static constexpr int T = WS*WS;
	static constexpr int WP = (WS + 1) & ~1;                // taps per window row, padded even
	static constexpr int WPIX = WS*WP;                      // doubles per pixel window
	static constexpr int RW = DC_CHUNK + 2*R + DC_NCB + 2;  // right tile width (even)
	static constexpr int LW = DC_TP + 2*R;                  // left tile width
	double w[DC_TP][WPIX];                                  // w[pixel][row*WP + col]
	double rt[WS][RW];
	double lt[WS][LW];
	double meanL[DC_TP], totalW[DC_TP], sum2[DC_TP], sumA[DC_TP];   // (sumA: pconst slot 4, the one-pass form's SA)
	int lall[DC_TP];
	int pxmin[DC_TP], pxmax[DC_TP];                         // candidate range of each pixel (empty: max < min)
	unsigned char rfull[RW];
	unsigned char colok[RW];
	static constexpr int GL_CAP = 2048;                     // pairs examined per compaction round
	static_assert(DC_CHUNK <= 512 && DC_TP <= 64, "work-list entry = pixel*512 + column in 16 bits");
	unsigned short glist[GL_CAP];
	int glist_n;
};

// general (any validity pattern) cost of one candidate from the LDS tiles.  Skipped taps add
// +0.0 to every sum, which leaves each partial sum bit-for-bit unchanged, so the selects
// below are the reference's "continue" (twoviewstereo.cpp:920-938, 955-972).
template <int R, int DC_NCB, int DC_CHUNK>
__device__ __noinline__ double dense_cost_general(const DenseSmem<R, DC_NCB, DC_CHUNK> &S, int i, int rc,
                                                  double weight_cutoff, double bad_ret, double max_color_diff)
{
	constexpr int WS = 2*R + 1;
	constexpr int WP = DenseSmem<R, DC_NCB, DC_CHUNK>::WP;
	double meanL = 0, meanR = 0, totalWeight = 0.0;
#pragma unroll 1
	for (int row = 0; row < WS; ++row) {
		double gl[WS], gr[WS], wt[WS];
#pragma unroll
		for (int col = 0; col < WS; ++col) { gl[col] = S.lt[row][i + col]; gr[col] = S.rt[row][rc + col]; wt[col] = S.w[i][row*WP + col]; }
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
		for (int col = 0; col < WS; ++col) { gl[col] = S.lt[row][i + col]; gr[col] = S.rt[row][rc + col]; wt[col] = S.w[i][row*WP + col]; }
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

#ifdef SRH_EXPERIMENT
// timing experiments only (make exp): repeat the block loops of every tile `g_exp_repeat` times (the marginal
// time of a repetition is the pure loop time), pad the dynamic LDS so that fewer workgroups fit a CU
__device__ int g_exp_repeat = 1;
__device__ int g_exp_scan_mode = 0;    // 1: the scan reads no cost (timing of the walk alone), 2: no walk (one label)
int g_exp_lds_pad = 0;
void exp_set_scan(int mode) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_exp_scan_mode), &mode, sizeof(int)); }
void strip_exp_set(int repeat);
void exp_set(int repeat, int lds_pad) {
	if (repeat >= 0) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_exp_repeat), &repeat, sizeof(int)); strip_exp_set(repeat); }
	if (lds_pad >= 0) g_exp_lds_pad = lds_pad;
}
#endif

// FMA = false: the reference's arithmetic, operation by operation (the default and the parity mode).
// FMA = true : the opt-in "fma" mode (option "arith" = 1): the same sums with every multiply-add of the block loops
// fused (one rounding instead of two): half the FP64 instructions, costs differ from the reference's in the last
// bits (|delta cost| ~ 1e-13), which can flip a winner only between near-tied candidates -- the mismatch rate is
// measured, not assumed (bench.py --arith fma, tests/test_gpu_arith_modes.py).
// AR = 3: the certified mode (srh_internal.hpp, CertBound): fused loops, and every fast-form value stored
// by the rule of the certified scan -- NaN for a candidate the bound does not cover, the clamp itself above clamp + e0,
// anything else unclamped.
template <int R, int DC_NCB, int DC_CHUNK, int MINW, int AR>
__global__ __launch_bounds__(DC_THREADS, MINW)
void twoview_dense_cost_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P,
                               int y0, int nrows, const double *__restrict__ wbuf, size_t wstride,
                               const double *__restrict__ tnum, double *__restrict__ cost, int cstride,
                               Counters *__restrict__ cnt, const double *__restrict__ pconst, const CertBound cb,
                               const PixRange *__restrict__ prange)
{
	constexpr bool FMA = AR != 0, CERT = AR == 3 || AR == 5, ONEPASS = AR == 5;   // 5: the certified one-pass form (srh_internal.hpp, CertBound)
	constexpr int WS = 2*R + 1;
	constexpr int T = WS*WS;
	typedef DenseSmem<R, DC_NCB, DC_CHUNK> Smem;
	constexpr int WP = Smem::WP;
	constexpr int NR = DC_NCB + 2*R;           // right-row values a block needs (even)
	static_assert(NR % 2 == 0 && WP % 2 == 0 && Smem::RW % 2 == 0 && DC_CHUNK % 2 == 0, "16-byte rows");
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
	const int lane = tid & 63;
	const int i = (tid >> 6)*8 + (lane & 7);   // pixel within the tile (pixel-fastest inside a wave)
	const int g = lane >> 3;                   // lane within the pixel
	const int x = x0 + i;

	const double nan = __builtin_nan("");

	// Per-phase cycle stamps exist only in a -DSRH_PROFILE_PHASES diagnostic build (make EXTRA=-DSRH_PROFILE_PHASES):
	// the shipped kernel carries no debug switch and no stamp.
#ifdef SRH_PROFILE_PHASES
	unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	unsigned long long tp = __builtin_readcyclecounter();
	unsigned long long t_fast = 0, n_fast = 0;
	const unsigned long long t_begin = tp;
#define SRH_STAMP(k) do { const unsigned long long tn_ = __builtin_readcyclecounter(); ph[k] += tn_ - tp; tp = tn_; } while (0)
#else
#define SRH_STAMP(k) do { } while (0)
#endif

	// Staging and bookkeeping run at raised wave priority: a freshly dispatched workgroup
	// shares its SIMDs with an older one that is deep in the FP64 loops, and must get its
	// loads issued immediately so that their latency hides under the other's arithmetic.
	__builtin_amdgcn_s_setprio(3);
	// ---- the windows, the reference rows and the pixel constants do not depend on the candidate ranges: their global
	// loads are issued first and travel while the ranges are worked out (the first LDS store comes after all loads)
	constexpr int NBW = (T*DC_TP + DC_THREADS - 1)/DC_THREADS;
	constexpr int NBL = (WS*Smem::LW + DC_THREADS - 1)/DC_THREADS;
	constexpr int NBR = (WS*Smem::RW + DC_THREADS - 1)/DC_THREADS;
	// tile-major window buffer: the T*DC_TP doubles of this tile are contiguous
	static_assert(DC_TP == SRH_WTILE, "dense tile = window-buffer tile");
	const double *wtile = wbuf + wbuf_offset(W, T, trow, x0);
	double tw_[NBW], tl_[NBL];
#pragma unroll
	for (int k = 0; k < NBW; ++k) {
		const int idx = tid + k*DC_THREADS;
		tw_[k] = (idx < T*DC_TP && x0 + (idx % DC_TP) < W) ? wtile[idx] : 0.0;
	}
#pragma unroll
	for (int k = 0; k < NBL; ++k) {
		const int idx = tid + k*DC_THREADS;
		const int ty = idx / Smem::LW, tx = idx % Smem::LW;
		const int gx = x0 - R + tx, gy = y - R + ty;
		tl_[k] = (idx < WS*Smem::LW && gx >= 0 && gy >= 0 && gx < W && gy < H) ? L.gray_tv[(size_t)gy*W + gx] : nan;
	}
	double pc_[5] = { 0.0, 0.0, 0.0, 0.0, 0.0 };
	if (g == 0 && x < W) {
		const double *pc = pconst + ((size_t)trow*W + x)*SRH_PC;
		pc_[0] = pc[0]; pc_[1] = pc[1]; pc_[2] = pc[2]; pc_[3] = pc[3]; pc_[4] = pc[4];
	}
	// (prange, round 6: the pixel's column range from pixel_range_kernel -- the same function on the same operands, made once
	// per band with every lane busy.  Worked out here it was cam_unproject + pinhole_column_range on one lane in eight
	// with the tile's other lanes waiting: 140 000 of a wave's 355 000 cycles on C2, more than its block loops' 105 000.)
	PixRange pr_; pr_.lo = 0; pr_.hi = -1;
	if (prange && g == 0 && x < W) pr_ = prange[(size_t)trow*W + x];
	// ---- union of the candidate ranges of the tile (one global load, one LDS reduction)
	__shared__ int s_cmin, s_cmax, s_need_pix, s_need_col;
	if (tid == 0) { s_cmin = 2147483647; s_cmax = -2147483647; s_need_pix = 0; s_need_col = 0; }
	__syncthreads();
	if (g == 0) {
		// candidate column range of the pixel (verified later by the scan kernel)
		int lo = 0, hi = -1;
		if (prange) { lo = pr_.lo; hi = pr_.hi; }
		else if (x < W && L.mask[(size_t)y*W + x] == 1) {
			const Ray ray = cam_unproject(L.cam, (x + 0.5) / P.image_scale, (y + 0.5) / P.image_scale);
			pinhole_column_range(ray, L.cam, Rv, P, tnum, cstride, lo, hi);
		}
		if (hi >= lo) hi = dense_cover_hi(lo, hi, DC_NCB, DC_G, !ONEPASS);   // columns beyond: evaluated by the fill kernel
		S.pxmin[i] = lo; S.pxmax[i] = hi;
		if (hi >= lo) { atomicMin(&s_cmin, lo); atomicMax(&s_cmax, hi); }
	}
	__syncthreads();
	Extent e; e.xmin = S.pxmin[i]; e.xmax = S.pxmax[i];
	const int cmin = s_cmin & ~1, cmax = s_cmax;            // chunks start on even columns
	SRH_STAMP(0);

	// ---- stage the windows, the reference rows and the first chunk of the other view's rows.
	// Every global load of the thread is issued before the first LDS store, so the whole
	// staging costs about one memory latency.
	{
		double tr_[NBR];
#pragma unroll
		for (int k = 0; k < NBR; ++k) {
			const int idx = tid + k*DC_THREADS;
			const int ty = idx / Smem::RW, tx = idx % Smem::RW;
			const int gx = cmin - R + tx, gy = y - R + ty;
			tr_[k] = (cmin <= cmax && idx < WS*Smem::RW && gx >= 0 && gy >= 0 && gx < Rv.w && gy < Rv.h)
			       ? Rv.gray_tv[(size_t)gy*Rv.w + gx] : nan;
		}
#pragma unroll
		for (int k = 0; k < NBW; ++k) {
			const int idx = tid + k*DC_THREADS;
			const int t = idx / DC_TP, pi = idx % DC_TP;
			if (idx < T*DC_TP) S.w[pi][(t / WS)*WP + (t % WS)] = tw_[k];
		}
#pragma unroll
		for (int k = 0; k < NBL; ++k) {
			const int idx = tid + k*DC_THREADS;
			if (idx < WS*Smem::LW) S.lt[idx / Smem::LW][idx % Smem::LW] = tl_[k];
		}
#pragma unroll
		for (int k = 0; k < NBR; ++k) {
			const int idx = tid + k*DC_THREADS;
			if (idx < WS*Smem::RW) S.rt[idx / Smem::RW][idx % Smem::RW] = tr_[k];
		}
	}
	__syncthreads();
	SRH_STAMP(1);

	// ---- per-pixel constants of the fast form: computed by the weights kernel while the window was in registers
	// (geodesic_reg_kernel / weights_kernel, `pconst`), SRH_PC doubles per pixel
	if (g == 0) {
		bool all = (x < W) && (e.xmax >= e.xmin);
		double mL = 0, tw = 0, s2 = 0;
		if (all) {
			mL = pc_[0]; tw = pc_[1]; s2 = pc_[2];
			all = pc_[3] != 0.0;
		}
		S.meanL[i] = mL; S.totalW[i] = (ONEPASS && all) ? pc_[3] : tw; S.sum2[i] = s2; S.lall[i] = all ? 1 : 0;   // (one-pass form: 1/totalWeight, pconst slot 3)
		S.sumA[i] = all ? pc_[4] : 0.0;
		if (!all && x < W && e.xmax >= e.xmin) s_need_pix = 1;      // this pixel needs the general form
	}
	SRH_STAMP(2);

	unsigned n_dev = 0;
	const Smem &CS = S;
	for (int cs = cmin; cs <= cmax; cs += DC_CHUNK) {
		if (cs != cmin) {
			__builtin_amdgcn_s_setprio(3);
			__syncthreads();   // previous chunk fully consumed
			double tr_[NBR];
#pragma unroll
			for (int k = 0; k < NBR; ++k) {
				const int idx = tid + k*DC_THREADS;
				const int ty = idx / Smem::RW, tx = idx % Smem::RW;
				const int gx = cs - R + tx, gy = y - R + ty;
				tr_[k] = (idx < WS*Smem::RW && gx >= 0 && gy >= 0 && gx < Rv.w && gy < Rv.h)
				       ? Rv.gray_tv[(size_t)gy*Rv.w + gx] : nan;
			}
#pragma unroll
			for (int k = 0; k < NBR; ++k) {
				const int idx = tid + k*DC_THREADS;
				if (idx < WS*Smem::RW) S.rt[idx / Smem::RW][idx % Smem::RW] = tr_[k];
			}
			__syncthreads();
		}
		// ... and, on the other lanes meanwhile, which candidate columns have a fully usable window
		for (int tx = tid; tx < Smem::RW; tx += DC_THREADS) {
			bool ok = true;
#pragma unroll
			for (int ty = 0; ty < WS; ++ty) { const double v = S.rt[ty][tx]; ok = ok && (v == v); }
			S.colok[tx] = ok ? 1 : 0;
		}
		if (tid == 0) s_need_col = 0;
		__syncthreads();
		for (int tx = tid; tx < Smem::RW; tx += DC_THREADS) {
			// rfull[k]: window of candidate column cs+k fully usable (tile columns k .. k+2R)
			bool ok = tx + 2*R < Smem::RW;
			if (ok) {
#pragma unroll
				for (int k = 0; k < WS; ++k) ok = ok && S.colok[tx + k] != 0;
			}
			S.rfull[tx] = ok ? 1 : 0;
			const int c = cs + tx;
			if (!ok && c >= s_cmin && c <= cmax && tx < DC_CHUNK) s_need_col = 1;   // a candidate column needs it
		}
		__syncthreads();
		SRH_STAMP(3);
		__builtin_amdgcn_s_setprio(0);

		if (x < W && e.xmax >= e.xmin) {
			const int lo = e.xmin > cs ? e.xmin : cs;
			const int hi = e.xmax < cs + DC_CHUNK - 1 ? e.xmax : cs + DC_CHUNK - 1;
			// blocks start on even columns (>= cs): 16-byte LDS reads of the other view's rows.  The one-pass form reads them
			// 8 bytes at a time in pairs (ds_read2_b64: no alignment beyond 8 bytes) and starts at the pixel's first column: a
			// range of 8*k columns is k blocks whatever its parity, and no column is left to the fill kernel
			const int lo_e = ONEPASS ? lo : (lo & ~1);
			const int nblocks = hi >= lo ? (hi - lo_e + DC_NCB)/DC_NCB : 0;
			// cost rows are tile-transposed: column k (relative to the pixel's xmin) of pixel i of this tile at
			// ((tile*cstride) + k)*DC_TP + i -- the 32 pixels' k-th costs are contiguous for the scan's wave loads
			double *crow = cost + (size_t)blockIdx.x*cstride*DC_TP + i;
			// phase 1: blocks of DC_NCB candidates in the fast form.  Candidates whose window is
			// not fully usable still ride along in the block but are not stored; phase 2 below
			// evaluates them (and every candidate of a pixel that has unusable taps itself).
#ifdef SRH_EXPERIMENT
			const int exp_rep = g_exp_repeat;
			for (int rep = 0; rep < exp_rep; ++rep)
#endif
			for (int b = g; b < (CS.lall[i] ? nblocks : 0); b += DC_G) {
				const int c0 = lo_e + b*DC_NCB;
				const int rc = c0 - cs;                 // tile column of the window's left edge (even)
				unsigned vm = 0;                        // bit j: candidate c0 + j in the pixel's range, its window in the other view fully usable
#pragma unroll
				for (int j = 0; j < DC_NCB; ++j) {
					const int c = c0 + j;
					vm |= (c >= lo && c <= hi && CS.rfull[rc + j] != 0) ? 1u << j : 0u;
				}
				const bool fast = vm != 0;
				n_dev += __builtin_popcount(vm);
#ifdef SRH_PROFILE_PHASES
				const unsigned long long t0 = __builtin_readcyclecounter();
#endif
				if (fast && ONEPASS) {
					// certified one-pass form (twoview_strip_cost_kernel): P = sum w r, Q = sum ((w l - meanL) w) r, U = sum w^2 r^2
					const double mL = CS.meanL[i], itw = CS.totalW[i], s2 = CS.sum2[i];   // (this form keeps 1/totalWeight there)
					const double sig3 = cb.sigma3(s2);
					double r[NR], q[NR], wv[WS], lv[WS], P_[DC_NCB], Q_[DC_NCB], U_[DC_NCB];
					{
						const double *rp = &CS.rt[0][rc];
						const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[i][0]);
#pragma unroll
						for (int k = 0; k < NR; ++k) r[k] = rp[k];
#pragma unroll
						for (int m = 0; m < (WS - 1)/2; ++m) { const double2 v = wp[m]; wv[2*m] = v.x; wv[2*m + 1] = v.y; }
						wv[WS - 1] = CS.w[i][WS - 1];
#pragma unroll
						for (int col = 0; col < WS; ++col) lv[col] = CS.lt[0][i + col];
					}
#pragma unroll
					for (int j = 0; j < DC_NCB; ++j) { P_[j] = 0.0; Q_[j] = 0.0; U_[j] = 0.0; }
					__builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 1
					for (int row = 0; row < WS; ++row) {
						const int nrow = row + 1 < WS ? row + 1 : 0;          // (the last refill is never used)
						const double *rp = &CS.rt[nrow][rc];
						const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[i][nrow*WP]);
						const double *lp = &CS.lt[nrow][i];
#pragma unroll
						for (int k = 0; k < DC_NCB - 1; ++k) q[k] = r[k]*r[k];
#pragma unroll
						for (int col = 0; col < WS; ++col) {
							q[col + DC_NCB - 1] = r[col + DC_NCB - 1]*r[col + DC_NCB - 1];
							const double a = __builtin_fma(wv[col], lv[col], -mL);
							const double c = a*wv[col], d = wv[col]*wv[col];
							__builtin_amdgcn_sched_barrier(0);
#pragma unroll
							for (int j = 0; j < DC_NCB; ++j) P_[j] = __builtin_fma(wv[col], r[col + j], P_[j]);
#pragma unroll
							for (int j = 0; j < DC_NCB; ++j) Q_[j] = __builtin_fma(c, r[col + j], Q_[j]);
#pragma unroll
							for (int j = 0; j < DC_NCB; ++j) U_[j] = __builtin_fma(d, q[col + j], U_[j]);
							__builtin_amdgcn_sched_barrier(0);
							lv[col] = lp[col];
							if (col & 1) {
								r[col - 1] = rp[col - 1]; r[col] = rp[col];
								const double2 u = wp[col >> 1]; wv[col - 1] = u.x; wv[col] = u.y;
							}
							__builtin_amdgcn_sched_barrier(0);
						}
#pragma unroll
						for (int k = WS - 1; k < NR; ++k) r[k] = rp[k];
						wv[WS - 1] = CS.w[i][nrow*WP + WS - 1];
					}
					constexpr double TT = (double)T;
#pragma unroll
					for (int j = 0; j < DC_NCB; ++j) {
						bool okc;                              // (every candidate finished, only the store masked: as in the strip kernel)
						const double v = onepass_finish(P_[j], Q_[j], U_[j], CS.sumA[i], itw, s2, TT, sig3, cb.zmax2, okc);
						if ((vm >> j) & 1u) crow[(size_t)(c0 + j - e.xmin)*DC_TP] = !okc ? __builtin_nan("") : (v > cb.m_hi ? P.max_color_diff : v);
					}
				} else if (fast) {
					const double mL = CS.meanL[i], tw = CS.totalW[i], s2 = CS.sum2[i];
					const double sig3 = CERT ? cb.sigma3(s2) : 0.0;
					// Both passes are modulo-scheduled by hand: r[] / wv[] / av[] hold the current
					// window row; as soon as a value has had its last use, the same register is
					// refilled with the next row's value, so the LDS latency is always a row ahead.
					static_assert(WS % 2 == 1, "odd window");
					double r[NR], wv[WS], acc[DC_NCB];
					{
						const double2 *rp = reinterpret_cast<const double2 *>(&CS.rt[0][rc]);
						const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[i][0]);
#pragma unroll
						for (int m = 0; m < NR/2; ++m) { const double2 v = rp[m]; r[2*m] = v.x; r[2*m + 1] = v.y; }
#pragma unroll
						for (int m = 0; m < (WS - 1)/2; ++m) { const double2 v = wp[m]; wv[2*m] = v.x; wv[2*m + 1] = v.y; }
						wv[WS - 1] = CS.w[i][WS - 1];
					}
#pragma unroll
					for (int j = 0; j < DC_NCB; ++j) acc[j] = 0.0;
					// all LDS reads of the loop pre-header have landed: inside the loop the compiler can
					// then count only the reads of the previous iteration (lgkmcnt(N) instead of 0)
					__builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 1
					for (int row = 0; row < WS; ++row) {
						const int nrow = row + 1 < WS ? row + 1 : 0;          // last refill = row 0, for pass 2
						const double2 *rp = reinterpret_cast<const double2 *>(&CS.rt[nrow][rc]);
						const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[i][nrow*WP]);
#pragma unroll
						for (int col = 0; col < WS; ++col) {
							// products first, sums second: no instruction waits on its predecessor, so a
							// wave keeps the FP64 pipe full even when it is alone on its SIMD
							if (FMA) {
#pragma unroll
								for (int j = 0; j < DC_NCB; ++j) acc[j] = __builtin_fma(wv[col], r[col + j], acc[j]);
							} else {
								double pr[DC_NCB];
#pragma unroll
								for (int j = 0; j < DC_NCB; ++j) pr[j] = wv[col]*r[col + j];
								__builtin_amdgcn_sched_barrier(0);
#pragma unroll
								for (int j = 0; j < DC_NCB; ++j) acc[j] += pr[j];        // meanR += weight*gray
							}
							__builtin_amdgcn_sched_barrier(0);              // keep each refill where it is written
							if (col & 1) {                                  // r[col-1], r[col], wv[col-1], wv[col] are dead
								const double2 v = rp[col >> 1]; r[col - 1] = v.x; r[col] = v.y;
								const double2 u = wp[col >> 1]; wv[col - 1] = u.x; wv[col] = u.y;
								__builtin_amdgcn_sched_barrier(0);
							}
						}
#pragma unroll
						for (int m = (WS - 1)/2; m < NR/2; ++m) { const double2 v = rp[m]; r[2*m] = v.x; r[2*m + 1] = v.y; }
						wv[WS - 1] = CS.w[i][nrow*WP + WS - 1];           // (the pad tap is never read)
					}
					double mR[DC_NCB], s1[DC_NCB], s3[DC_NCB], av[WS];
#pragma unroll
					for (int col = 0; col < WS; ++col) av[col] = CS.lt[0][i + col];
#pragma unroll
					for (int j = 0; j < DC_NCB; ++j) { mR[j] = acc[j]/tw; s1[j] = 0.0; s3[j] = 0.0; }
					__builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 1
					for (int row = 0; row < WS; ++row) {
						const int nrow = row + 1 < WS ? row + 1 : 0;
						const double2 *rp = reinterpret_cast<const double2 *>(&CS.rt[nrow][rc]);
						const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[i][nrow*WP]);
						const double *lp = &CS.lt[nrow][i];
#pragma unroll
						for (int col = 0; col < WS; ++col) {
							const double wt = wv[col];
							if (FMA) {
								const double a = __builtin_fma(wt, av[col], -mL);
								double bb[DC_NCB];
#pragma unroll
								for (int j = 0; j < DC_NCB; ++j) bb[j] = __builtin_fma(wt, r[col + j], -mR[j]);
								__builtin_amdgcn_sched_barrier(0);
#pragma unroll
								for (int j = 0; j < DC_NCB; ++j) { s1[j] = __builtin_fma(a, bb[j], s1[j]); s3[j] = __builtin_fma(bb[j], bb[j], s3[j]); }
							} else {
							double bb[DC_NCB], u1[DC_NCB], u3[DC_NCB];
							const double pa = wt*av[col];
#pragma unroll
							for (int j = 0; j < DC_NCB; ++j) bb[j] = wt*r[col + j];
							__builtin_amdgcn_sched_barrier(0);
							const double a = pa - mL;                         // pixel_gray_l - meanL
#pragma unroll
							for (int j = 0; j < DC_NCB; ++j) bb[j] = bb[j] - mR[j];   // pixel_gray_r - meanR
							__builtin_amdgcn_sched_barrier(0);
#pragma unroll
							for (int j = 0; j < DC_NCB; ++j) { u1[j] = a*bb[j]; u3[j] = bb[j]*bb[j]; }
							__builtin_amdgcn_sched_barrier(0);
#pragma unroll
							for (int j = 0; j < DC_NCB; ++j) { s1[j] += u1[j]; s3[j] += u3[j]; }
							}
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
						wv[WS - 1] = CS.w[i][nrow*WP + WS - 1];           // (the pad tap is never read)
					}
#pragma unroll
					for (int j = 0; j < DC_NCB; ++j) {
						const int c = c0 + j;
						if (c >= lo && c <= hi && CS.rfull[rc + j] != 0) {
							const double v = 255*(1.0 - fabs(s1[j]) / sqrt(s2 * s3[j]));
							if (CERT) crow[(size_t)(c - e.xmin)*DC_TP] = !(s3[j] >= sig3) ? __builtin_nan("") : (v > cb.m_hi ? P.max_color_diff : v);
							else crow[(size_t)(c - e.xmin)*DC_TP] = (v < P.max_color_diff) ? v : P.max_color_diff;
						}
						__builtin_amdgcn_sched_barrier(0);
					}
#ifdef SRH_PROFILE_PHASES
					t_fast += __builtin_readcyclecounter() - t0; ++n_fast;
#endif
				}
			}
		}
		// phase 2: the remaining candidates in the general form (any validity pattern), spread
		// over all lanes of the workgroup: pair p = (pixel, chunk column), consecutive lanes take
		// consecutive columns, so image-border columns and border rows are shared evenly.
		// The pairs are first compacted into an LDS work list (rounds of GL_CAP pairs) so that a
		// wave that enters the general code has (nearly) all of its lanes busy.
		const bool need_general = s_need_pix != 0 || s_need_col != 0;   // uniform: read after the barrier above
		for (int base = 0; need_general && base < DC_TP*DC_CHUNK; base += Smem::GL_CAP) {
			if (tid == 0) S.glist_n = 0;
			__syncthreads();
			for (int p = base + tid; p < base + Smem::GL_CAP && p < DC_TP*DC_CHUNK; p += DC_THREADS) {
				const int pi = p / DC_CHUNK, k = p % DC_CHUNK;
				const int c = cs + k;
				if (c < S.pxmin[pi] || c > S.pxmax[pi]) continue;
				if (CS.lall[pi] && CS.rfull[k]) continue;         // done in phase 1
				S.glist[atomicAdd(&S.glist_n, 1)] = (unsigned short)(pi*512 + k);
			}
			__syncthreads();
			const int nl = S.glist_n;
			for (int q = tid; q < nl; q += DC_THREADS) {
				const int pi = S.glist[q] >> 9, k = S.glist[q] & 511;
				++n_dev;
				cost[((size_t)blockIdx.x*cstride + (cs + k - S.pxmin[pi]))*DC_TP + pi] =
					dense_cost_general<R, DC_NCB, DC_CHUNK>(CS, pi, k, P.weight_cutoff, P.bad_ret, P.max_color_diff);
			}
		}
	}
	SRH_STAMP(4);
	block_count_add(&cnt->n_eval_device, n_dev);
	SRH_STAMP(5);
#ifdef SRH_PROFILE_PHASES
	if ((tid & 63) == 0) {
		for (int k = 0; k < 8; ++k) atomicAdd(&cnt->dbg_phase[k], ph[k]);
		atomicAdd(&cnt->dbg_cycles, t_fast);
		atomicAdd(&cnt->dbg_blocks, n_fast);
		atomicAdd(&cnt->dbg_total_cycles, (unsigned long long)(__builtin_readcyclecounter() - t_begin));
		atomicAdd(&cnt->dbg_waves, 1ull);
	}
#endif
#undef SRH_STAMP
}

template <int R, int NCB, int CHUNK, int MINW, int AR>
static void launch_dense_variant(hipStream_t st, dim3 grid, const ViewDev *views, int ref, int oth, const srh_params &P,
                                 int y0, int nrows, const double *wbuf, size_t wstride,
                                 const double *tnum, double *cost, int cstride, Counters *cnt, const double *pconst, const PixRange *prange)
{
	typedef DenseSmem<R, NCB, CHUNK> Smem;
	// a function attribute belongs to the CURRENT device: set it on every launch (a host-side table update),
	// so contexts on several GPUs of one process all get their dynamic LDS
	size_t lds = sizeof(Smem);
#ifdef SRH_EXPERIMENT
	lds += (size_t)g_exp_lds_pad;
#endif
	(void)hipFuncSetAttribute((const void *)twoview_dense_cost_kernel<R, NCB, CHUNK, MINW, AR>,
	                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipLaunchKernelGGL((twoview_dense_cost_kernel<R, NCB, CHUNK, MINW, AR>), grid, dim3(DC_THREADS), lds, st,
	                   views, ref, oth, P, y0, nrows, wbuf, wstride, tnum, cost, cstride, cnt, pconst, cert_bound(P), prange);
}

bool launch_twoview_dense_cost(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                               int y0, int nrows, const double *wbuf, size_t wstride,
                               const double *tnum, double *cost, int cstride, Counters *cnt, const double *pconst, int arith,
                               const PixRange *prange)
{
	const int tiles = (width + DC_TP - 1)/DC_TP;
	const dim3 grid((unsigned)(tiles*nrows));
#define SRH_ARGS st, grid, views, ref, oth, P, y0, nrows, wbuf, wstride, tnum, cost, cstride, cnt, pconst, prange
	switch (P.window_radius) {
	case 5:
		if (arith == 5) launch_dense_variant<5, 8, 320, 2, 5>(SRH_ARGS);
		else if (arith == 3) launch_dense_variant<5, 8, 320, 2, 3>(SRH_ARGS);
		else if (arith == 1) launch_dense_variant<5, 8, 320, 2, 1>(SRH_ARGS);
		else launch_dense_variant<5, 8, 320, 2, 0>(SRH_ARGS);
		return true;
	case 2:
		if (arith == 5) launch_dense_variant<2, 8, 320, 2, 5>(SRH_ARGS);
		else if (arith == 3) launch_dense_variant<2, 8, 320, 2, 3>(SRH_ARGS);
		else if (arith == 1) launch_dense_variant<2, 8, 320, 2, 1>(SRH_ARGS);
		else launch_dense_variant<2, 8, 320, 2, 0>(SRH_ARGS);
		return true;
	default: return false;
	}
#undef SRH_ARGS
}

// ------------------------------------------------------------------ scan: walk + look-up + WTA
// One thread per reference pixel, one wave (64 consecutive pixels of a row) per workgroup.
// The curve walk is pure arithmetic; what used to serialise it were two dependent global loads
// per candidate (mask byte, cost).  Here the other view's mask row segment sits in LDS, and the
// candidates' columns go through a small per-thread LDS queue: every SC_QN candidates their
// costs are fetched with SC_QN independent loads and then consumed in order, so the running-min
// scan of twoviewstereo.cpp:293-301 sees exactly the reference's candidate sequence.
#define SC_TW 64
#ifndef SC_QN
#define SC_QN 8                    // queue depth per thread: 16 cost 48 registers of look-up state and one wave per SIMD (measured: 2.45 -> 2.20 ms)
#endif
#ifndef SC_OCC
#define SC_OCC 6
#endif
#define SC_MW 1024

// The columns the cost kernel leaves out (dense_cover_hi: a last block that would cost its lanes a whole extra round
// for one or two columns) are evaluated here, one thread per pixel, with the general cost -- so that the scan kernel
// only ever looks costs up.  (Inlined at every slot of the scan's flushes, that general cost made the scan kernel
// 27 000 instructions long, far beyond the instruction cache.)  Radii without a template instance: tv_cost through the views.
__global__ void twoview_lazy_fill_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P, int y0, int nrows,
                                         const PixRange *__restrict__ prange, const double *__restrict__ wbuf, size_t wstride,
                                         int ncb, int lanes, int pad, double *__restrict__ cost, int cstride, Counters *__restrict__ cnt)
{
	const ViewDev &L = views[ref];
	const ViewDev &Rv = views[oth];
	const int W = L.w;
	const size_t q = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
	unsigned n_lazy = 0;
	if (q < (size_t)nrows*W) {
		const int x = (int)(q % W), trow = (int)(q / W), y = y0 + trow;
		const PixRange pr = prange[q];
		if (pr.hi >= pr.lo) {
			const int cover = dense_cover_hi(pr.lo, pr.hi, ncb, lanes, pad != 0);
			if (cover < pr.hi) {
				const int R = P.window_radius, T = (2*R + 1)*(2*R + 1);
				const double *wq = wbuf + wbuf_offset(W, T, trow, x);
				double *crow = cost + ((size_t)trow*((W + DC_TP - 1)/DC_TP) + (x/DC_TP))*(size_t)cstride*DC_TP + (x % DC_TP);
				for (int c = cover + 1; c <= pr.hi; ++c) {
					crow[(size_t)(c - pr.lo)*DC_TP] = tv_cost(L, Rv, wq, wstride, P, x, y, c, y);
					++n_lazy;
				}
			}
		}
	}
	block_count_add(&cnt->n_eval_device, n_lazy);
}

template <int R>
__global__ __launch_bounds__(256)
void twoview_lazy_fill_planes_kernel(int W, srh_params P, int y0, int nrows, const PixRange *__restrict__ prange,
                                     const double *__restrict__ wbuf, int wimg, const double *__restrict__ ref_tvp,
                                     const double *__restrict__ oth_tvp, int ncb, int lanes, int pad,
                                     double *__restrict__ cost, int cstride, Counters *__restrict__ cnt)
{
	const size_t q = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
	unsigned n_lazy = 0;
	if (q < (size_t)nrows*W) {
		const int x = (int)(q % W), trow = (int)(q / W), y = y0 + trow;
		const PixRange pr = prange[q];
		const int cover = pr.hi >= pr.lo ? dense_cover_hi(pr.lo, pr.hi, ncb, lanes, pad != 0) : pr.hi;
		if (cover < pr.hi) {
			const int SP = padded_stride(W);
			const WindowAt wa = window_at<R>(wbuf, wimg, W, trow, x);
			const double *lp = ref_tvp + (size_t)(y + SRH_PADY - R)*SP + (x + SRH_PADL - R);
			double *crow = cost + ((size_t)trow*((W + DC_TP - 1)/DC_TP) + (x/DC_TP))*(size_t)cstride*DC_TP + (x % DC_TP);
			for (int c = cover + 1; c <= pr.hi; ++c) {
				const double *rp = oth_tvp + (size_t)(y + SRH_PADY - R)*SP + (c + SRH_PADL - R);
				crow[(size_t)(c - pr.lo)*DC_TP] = window_exact_cost<R>(wa.wq, wa.wrow, wa.wcol, lp, rp, SP, SP, P);
				++n_lazy;
			}
		}
	}
	block_count_add(&cnt->n_eval_device, n_lazy);
}

// Certified arithmetic: the cost rows of the flagged pixels (cflag[1 .. 1 + cflag[0])) once more, every column of the pixel's
// range in the reference's arithmetic -- one 64-lane workgroup per pixel, a lane per column.  The grid is sized by the
// CAPACITY of the redo (`cap` workgroups: the host does not wait for the count); workgroups beyond the count leave at
// once, and a count above the capacity is reported in cnt->cert_overflow: the host then repeats the pass in mode 0.
template <int R>
__global__ __launch_bounds__(64)
void twoview_refill_kernel(int W, srh_params P, int y0, const PixRange *__restrict__ prange, const uint32_t *__restrict__ cflag, int cap,
                           const double *__restrict__ wbuf, int wimg, const double *__restrict__ ref_tvp, const double *__restrict__ oth_tvp,
                           double *__restrict__ cost, int cstride, Counters *__restrict__ cnt)
{
	const uint32_t nflag = cflag[0];
	if (blockIdx.x == 0 && threadIdx.x == 0 && nflag > (uint32_t)cap) atomicAdd(&cnt->cert_overflow, 1ull);
	unsigned n = 0;
	for (uint32_t f = blockIdx.x; f < nflag && f < (uint32_t)cap; f += gridDim.x) {      // (a few hundred workgroups share the list)
		const uint32_t q = cflag[1 + f];
		const int x = (int)(q % (uint32_t)W), trow = (int)(q / (uint32_t)W), y = y0 + trow;
		const PixRange pr = prange[q];
		const int SP = padded_stride(W);
		const WindowAt wa = window_at<R>(wbuf, wimg, W, trow, x);
		const double *lp = ref_tvp + (size_t)(y + SRH_PADY - R)*SP + (x + SRH_PADL - R);
		double *crow = cost + ((size_t)trow*((W + DC_TP - 1)/DC_TP) + (x/DC_TP))*(size_t)cstride*DC_TP + (x % DC_TP);
		for (int c = pr.lo + (int)threadIdx.x; c <= pr.hi; c += 64) {
			const double *rp = oth_tvp + (size_t)(y + SRH_PADY - R)*SP + (c + SRH_PADL - R);
			crow[(size_t)(c - pr.lo)*DC_TP] = window_exact_cost<R>(wa.wq, wa.wrow, wa.wcol, lp, rp, SP, SP, P);
			++n;
		}
	}
	block_count_add(&cnt->n_eval_device, n);
}

bool launch_twoview_refill(hipStream_t st, int width, const srh_params &P, int y0, const PixRange *prange, const uint32_t *cflag, int cap,
                           const double *wbuf, bool wimg, const double *ref_tvp, const double *oth_tvp, double *cost, int cstride, Counters *cnt)
{
	if (cap <= 0) return true;
	if (P.window_radius == 5)
		hipLaunchKernelGGL(twoview_refill_kernel<5>, dim3((unsigned)(cap < 8192 ? cap : 8192)), dim3(64), 0, st, width, P, y0, prange, cflag, cap, wbuf, wimg ? 1 : 0, ref_tvp, oth_tvp, cost, cstride, cnt);
	else if (P.window_radius == 2)
		hipLaunchKernelGGL(twoview_refill_kernel<2>, dim3((unsigned)(cap < 8192 ? cap : 8192)), dim3(64), 0, st, width, P, y0, prange, cflag, cap, wbuf, wimg ? 1 : 0, ref_tvp, oth_tvp, cost, cstride, cnt);
	else return false;
	return true;
}

// ref_tvp / oth_tvp: the NaN-bordered planes (radius 5 or 2: the fast general cost); null: tv_cost through the views.
// wimg: the band buffer holds the strip path's LDS-image windows (lanes / pad as that kernel's form), else tile-major.
void launch_twoview_lazy_fill(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                              int y0, int nrows, const PixRange *prange, const double *wbuf, size_t wstride,
                              const double *ref_tvp, const double *oth_tvp, bool wimg,
                              int lanes, bool padded, double *cost, int cstride, Counters *cnt)
{
	const size_t n = (size_t)nrows*width;
	const dim3 grid((unsigned)((n + 255)/256)), block(256);
	const int pad = padded ? 1 : 0;
	if (ref_tvp && P.window_radius == 5)
		hipLaunchKernelGGL(twoview_lazy_fill_planes_kernel<5>, grid, block, 0, st, width, P, y0, nrows, prange, wbuf, wimg ? 1 : 0, ref_tvp, oth_tvp, 8, lanes, pad, cost, cstride, cnt);
	else if (ref_tvp && P.window_radius == 2)
		hipLaunchKernelGGL(twoview_lazy_fill_planes_kernel<2>, grid, block, 0, st, width, P, y0, nrows, prange, wbuf, wimg ? 1 : 0, ref_tvp, oth_tvp, 8, lanes, pad, cost, cstride, cnt);
	else
		hipLaunchKernelGGL(twoview_lazy_fill_kernel, grid, block, 0, st, views, ref, oth, P, y0, nrows, prange, wbuf, wstride, 8, lanes, pad, cost, cstride, cnt);
}

// sums / minima over the 64 lanes of the scan kernel's one-wave workgroup (no LDS, no barrier)
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) v += (unsigned)__shfl_xor((int)v, d);
	return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_xor(v, d); v = o < v ? o : v; }
	return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_xor(v, d); v = o > v ? o : v; }
	return v;
}

struct TwoViewScanState {
	double minCost, secondBest;
	int wcol;                 // winning column relative to lo, -1 = none
};

// CERT: the cost rows hold FUSED costs (twoview_strip_cost_kernel<.., 3>): the reference's decisions are replayed on
// them, and a pixel is FLAGGED -- appended to cflag[1..], count in cflag[0] -- when a decision is not covered by the
// error bound (CertBound, srh_internal.hpp).  What a stored value x says about the cost in the reference's arithmetic:
//   x == max_color_diff, or x > max_color_diff + e0 (bad_ret)   the very same number ("sure": the strip kernel stores the
//                                                               clamp itself only when the fused value is above it by more
//                                                               than e0; larger values come from the exact select forms);
//   x NaN                                                       an uncertified candidate: the pixel is flagged;
//   anything else                                               within e0 of it (select-form candidates are exact and
//                                                               counted as e0; a fused value in (clamp, clamp + e0] stands
//                                                               for a reference value in [x - e0, clamp]).
// A comparison a < b on fused values is the reference's when |a - b| exceeds the two bounds (+ the rounding of cost +
// margin), or when both sides are sure (the same numbers go through the same operations).  The state (minCost,
// secondBest, winner) then evolves identically, by induction over the candidate sequence; an unflagged pixel made every
// comparison the way the reference's arithmetic makes it, so its winner -- and its depth, which is geometry -- are the
// reference's bits.
// LISTED: the exact scan of the flagged pixels only (lane k takes pixel cflag[1 + k]; their cost rows were refilled in
// the reference's arithmetic by twoview_refill_kernel).
// one tile (64 pixels of a row; LISTED: 64 listed pixels) of twoview_scan_kernel; `bid` = its index in the band
template <bool CERT, bool LISTED>
__device__ __forceinline__
void twoview_scan_tile(const int bid, const ViewDev *__restrict__ views, int ref, int oth, const srh_params &P,
                       int y0, int nrows, const double *__restrict__ tnum,
                       const double *__restrict__ cost, int cstride,
                       Counters *__restrict__ cnt, const PixRange *__restrict__ prange,
                       uint32_t *__restrict__ cflag, int nlist, const CertBound &cb, const double *__restrict__ pexact)
{
	const ViewDev &L = views[ref];
	const ViewDev &Rv = views[oth];
	const int W = L.w, OW = Rv.w, OH = Rv.h;
	const int tiles_per_row = (W + SC_TW - 1)/SC_TW;
	const int tid = threadIdx.x;
	int trow, x;
	bool listed_on = true;
	if (LISTED) {
		// (grid sized by the capacity `nlist`: the count is on the device, twoview_refill_kernel)
		const uint32_t nflag = cflag[0];
		const int k = bid*SC_TW + tid;
		listed_on = k < nlist && (uint32_t)k < nflag;
		const uint32_t q = listed_on ? cflag[1 + k] : 0u;
		trow = (int)(q / (uint32_t)W); x = (int)(q % (uint32_t)W);
	} else {
		trow = bid / tiles_per_row;
		x = (bid % tiles_per_row)*SC_TW + tid;
	}
	const int y = y0 + trow;

	__shared__ unsigned short queue[SC_QN][SC_TW];
	__shared__ __align__(16) unsigned char mrow_raw[SC_MW + 16];

	unsigned n_eval = 0, n_pix = 0, bad = 0, n_flag = 0;
	const bool active = listed_on && x < W && L.mask[(size_t)y*W + x] == 1;
	Ray ray;
	int lo = 0, hi = -1;
	if (active) {
		ray = cam_unproject(L.cam, (x + 0.5) / P.image_scale, (y + 0.5) / P.image_scale);
		const PixRange pr = prange[(size_t)trow*W + x];      // pixel_range_kernel (pinhole_column_range)
		lo = pr.lo; hi = pr.hi;
	}
	// LISTED: the lanes' pixels lie on different rows: no shared mask row, every mask byte comes from memory
	const int umin = LISTED ? 2147483647 : wave_min_i32(hi >= lo ? lo : 2147483647);       // the workgroup is one wave
	// mask bytes of row y of the other view, columns [umin, umin + SC_MW): 16 bytes per lane from the 4-byte granule
	// that holds the first one (bytes past the row's end belong to the next row and are never looked at: every
	// candidate column lies inside [lo, hi], inside the image)
	const unsigned char *mrow = mrow_raw;
	if (umin != 2147483647 && y >= 0 && y < OH) {
		const size_t first = (size_t)y*OW + (size_t)umin, a0 = first & ~(size_t)3, total = (size_t)OW*OH;
		mrow = mrow_raw + (first - a0);
		const size_t at = a0 + (size_t)tid*16;
		uint32_t v[4];
#pragma unroll
		for (int k = 0; k < 4; ++k)
			v[k] = (at + 4*k + 4 <= total) ? *reinterpret_cast<const uint32_t *>(Rv.mask + at + 4*k) : 0u;   // (the plane's size is not a multiple of 4: its last bytes ...)
		if (at + 16 > total) {                                                                                   // ... one by one)
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				uint32_t w = 0;
				for (int b2 = 0; b2 < 4; ++b2) if (at + 4*k + b2 < total) w |= (uint32_t)Rv.mask[at + 4*k + b2] << (8*b2);
				v[k] = w;
			}
		}
		uint4 q; q.x = v[0]; q.y = v[1]; q.z = v[2]; q.w = v[3];
		*reinterpret_cast<uint4 *>(mrow_raw + tid*16) = q;
	}
	__syncthreads();

	double depth = __builtin_nan("");
	if (active) {
		n_pix = 1;
		const double *crow = cost + ((size_t)trow*((W + DC_TP - 1)/DC_TP) + (x/DC_TP))*(size_t)cstride*DC_TP + (x % DC_TP);
		TwoViewScanState st = { __builtin_inf(), __builtin_inf(), -1 };
		int qn = 0;
		// pexact (the band's per-pixel constants, given when the cost kernel redoes uncovered candidates in place): a pixel the
		// cost kernel evaluated in the reference's arithmetic throughout (cert_pixel_exact) -- every stored cost is the
		// reference's own number, every comparison on them is the reference's
		bool px_sure = false;
		if (CERT && pexact) { const double *pc = pexact + ((size_t)trow*W + x)*SRH_PC; px_sure = cert_pixel_exact(cb, pc[2], pc[3]); }

		// consume `count` queued candidates (count == SC_QN except for the final, partial flush)
		auto flush = [&](int count) {
			int col[SC_QN];
			double c[SC_QN];
#pragma unroll
			for (int k = 0; k < SC_QN; ++k) col[k] = queue[k][tid];
#pragma unroll
			for (int k = 0; k < SC_QN; ++k)
#ifdef SRH_EXPERIMENT
				c[k] = (k < count && g_exp_scan_mode != 1) ? crow[(size_t)col[k]*DC_TP] : (double)col[k];
#else
				c[k] = (k < count) ? crow[(size_t)col[k]*DC_TP] : __builtin_inf();
#endif
#pragma unroll
			for (int k = 0; k < SC_QN; ++k) {
				if (k < count) {
					const double cv = c[k];
					if (CERT) {
						// (a candidate column seen again while it is the winner compares its own cost with itself: false in
						// either arithmetic since wta_margin >= 0)
						// (tolerance: the two bounds + the rounding of cost + margin, 2u|t| <= e0/2: a side that is not sure is a
						// cost, a clamp or a bad_ret of magnitude <= 1e5 -- CertBound::ok)
						const double t = cv + P.wta_margin;
						if (!(fabs(t - st.minCost) > 2.5*cb.e0) && col[k] != st.wcol &&
						    !(cert_sure(cv, P.max_color_diff, cb.m_hi) && cert_sure(st.minCost, P.max_color_diff, cb.m_hi)) &&
						    !(px_sure && cv == cv)) n_flag = 1;              // (NaN: flagged, whatever the pixel)
					}
					if (cv + P.wta_margin < st.minCost) {           // twoviewstereo.cpp:293-301
						st.secondBest = st.minCost;
						st.minCost = cv;
						st.wcol = col[k];
					}
				}
			}
		};

		// ---- the pinhole walk (walk_curve_pinhole), candidates pushed instead of visited
		const Vec3 nrm = normalized(load3(L.cam.pdir));
		const double nd = dot(nrm, ray.dir);
		if (!(fabs(nd) < 1e-10)) {
			double x1 = __builtin_nan(""), y1 = __builtin_nan("");
			int jx1 = 0, jy1 = 0;                                     // the kept point's truncated coordinates travel with it
			const SharedDivisor nd_sd = shared_divisor(nd);           // 256 labels are divided by this one n.dir
#ifdef SRH_EXPERIMENT
			const int exp_nd = g_exp_scan_mode == 2 ? 2 : P.num_depth_levels;
			for (int d = 0; d < exp_nd; ++d) {
#else
			for (int d = 0; d < P.num_depth_levels; ++d) {
#endif
				double x2, y2;
				if (!pinhole_project_label_sd(ray, nd_sd, tnum[d], Rv.cam, P.image_scale, x2, y2)) continue;
				if (isnan_d(x1)) { x1 = x2; y1 = y2; jx1 = trunc_sat(x2); jy1 = trunc_sat(y2); continue; }
				const double dx = x2 - x1, dy = y2 - y1;
				if (!(dx*dx + dy*dy >= 1)) continue;
				const int ix0 = jx1, iy0 = jy1, ix1 = trunc_sat(x2), iy1 = trunc_sat(y2);
				jx1 = ix1; jy1 = iy1;
				const int a = ix0 < ix1 ? ix0 : ix1, b = ix0 < ix1 ? ix1 : ix0;
				if (iy0 == y && iy1 == y && a >= lo && b <= hi) {
					// a segment inside row y: LineIterator with deltay == 0 visits (a..b, y) in
					// ascending x whatever the direction of the segment (lineiter.hpp:96-111)
					for (int tx = a; tx <= b; ++tx) {
						const int k = tx - umin;
						const bool white = (!LISTED && k >= 0 && k < SC_MW - 4) ? (mrow[k] == 1) : (Rv.mask[(size_t)y*OW + tx] == 1);
						if (white) {
							++n_eval;
							queue[qn][tid] = (unsigned short)(tx - lo);
							if (++qn == SC_QN) { flush(SC_QN); qn = 0; }
						}
					}
				} else {
					LineWalk lw;
					lw.begin(ix0, iy0, ix1, iy1, OW, OH);
					while (lw.has_next()) {
						int tx, ty;
						lw.current(tx, ty);
						if (tx >= 0 && ty >= 0 && tx < OW && ty < OH && Rv.mask[(size_t)ty*OW + tx] == 1) {
							++n_eval;
							if (ty != y || tx < lo || tx > hi) bad = 1;          // not row-aligned after all
							else {
								queue[qn][tid] = (unsigned short)(tx - lo);
								if (++qn == SC_QN) { flush(SC_QN); qn = 0; }
							}
						}
						lw.next();
					}
				}
				x1 = x2; y1 = y2;
				// flush wave-wide as soon as one lane's queue is half full: all lanes take the
				// look-up path together instead of each on its own (divergent) schedule
				if (__any(qn >= SC_QN/2)) { flush(qn); qn = 0; }
			}
		}
		if (qn > 0) flush(qn);

		if (st.wcol >= 0)
			depth = candidate_depth(L.cam, Rv.cam, P, ray, lo + st.wcol, y);
		if (st.minCost > P.second_best_factor*st.secondBest)
			depth = __builtin_inf();
		if (CERT && st.wcol >= 0) {
			// the ratio test (twoviewstereo.cpp:303-305) on fused values: minCost within e0, factor*secondBest within |factor|*e0
			const double rhs = P.second_best_factor*st.secondBest;
			const double tol = __builtin_fma(fmin(fabs(rhs), 1e300), 1e-15, (1.0 + fabs(P.second_best_factor))*cb.e0);
			if (!(fabs(st.minCost - rhs) > tol) && !px_sure &&
			    !(cert_sure(st.minCost, P.max_color_diff, cb.m_hi) && cert_sure(st.secondBest, P.max_color_diff, cb.m_hi))) n_flag = 1;
		}
		if (CERT && n_flag) cflag[1 + atomicAdd(&cflag[0], 1u)] = (uint32_t)((size_t)trow*W + x);
	}
	if (listed_on && x < W) L.depth[(size_t)y*W + x] = depth;
	if (LISTED || !cnt) return;                        // (the pixels were counted by the certified scan)
	// the tile is one wave: its counts are summed by lane shuffles, one atomic each
	n_eval = wave_sum_u32(n_eval); n_pix = wave_sum_u32(n_pix); bad = wave_sum_u32(bad); n_flag = wave_sum_u32(n_flag);
	if (tid == 0) {
		if (n_eval) atomicAdd(&cnt->n_eval, (unsigned long long)n_eval);
		if (n_pix) atomicAdd(&cnt->n_pixels, (unsigned long long)n_pix);
		if (bad) atomicAdd(&cnt->not_row_aligned, (unsigned long long)bad);
		if (CERT && n_pix) atomicAdd(&cnt->n_certified, (unsigned long long)n_pix);
		if (CERT && n_flag) atomicAdd(&cnt->n_flagged, (unsigned long long)n_flag);
	}
}

// tilelist == nullptr: one workgroup per tile of the band (LISTED: per 64 listed pixels).  tilelist = [count | tile
// indices]: behind twoview_tscan_kernel, only the tiles it left (a pixel whose curve is not certainly the template's),
// shared by the launch's workgroups in a grid-stride loop; the count is read on the device.
template <bool CERT, bool LISTED>
__global__ __launch_bounds__(SC_TW, CERT ? SC_OCC - 1 : SC_OCC)
void twoview_scan_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P,
                         int y0, int nrows, const double *__restrict__ tnum,
                         const double *__restrict__ cost, int cstride,
                         Counters *__restrict__ cnt, const PixRange *__restrict__ prange,
                         uint32_t *__restrict__ cflag, int nlist, CertBound cb, const double *__restrict__ pexact,
                         const uint32_t *__restrict__ tilelist)
{
	if (LISTED || !tilelist) {
		twoview_scan_tile<CERT, LISTED>((int)blockIdx.x, views, ref, oth, P, y0, nrows, tnum, cost, cstride, cnt, prange, cflag, nlist, cb, pexact);
		return;
	}
	const uint32_t n = tilelist[0];
	for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
		twoview_scan_tile<CERT, LISTED>((int)tilelist[1 + i], views, ref, oth, P, y0, nrows, tnum, cost, cstride, cnt, prange, cflag, nlist, cb, pexact);
		__syncthreads();                                              // (the tile's LDS is reused)
	}
}

// ------------------------------------------------------------------ template scan
// On a rectified rig the candidate sequence of a pixel is, up to roundings, the same for every pixel of the image: the
// label projections differ from pixel to pixel only in the last bits, so every pixel keeps the same labels and its kept
// points truncate to x + (an offset that depends on the label alone).  twoview_scan_kernel nevertheless re-derives the
// sequence per pixel -- ~150 instructions per (pixel, label): projection, one-pixel test, truncations, segment set-up,
// per-candidate queueing -- and that, not the look-ups, is its time (2.3 ms per C3 launch).
//
// Here the sequence is made ONCE per pass, by the reference's own operations at one pixel (twoview_template_kernel:
// per label its state -- not projectable / first point / dropped by the one-pixel test / kept -- and the kept point's
// column offset; the visiting order of all candidate columns as offsets), and every pixel only VERIFIES that its own
// curve is that one: label by label with the certified projections of DESIGN.md 2c (fast_project: 3 fused multiply-adds
// and a reciprocal, with a proven bound on its distance from the reference's value) -- the projectability test on the
// reference's own t, the one-pixel test decided beyond its bound and as the template has it, both coordinates of a kept
// point truncating certainly and to the template's column and the pixel's own row.  A pixel that passes has the
// template's candidate list, entry for entry (the reference's list: twoviewstereo.cpp:999-1054); the look-ups then run
// over the shared sequence with every lane of the wave on the same entry: no queue, no divergence, the 32 pixels of a
// tile reading one 256-byte line of the cost rows per entry.  A wave with a pixel that does not pass (a decision inside
// its bound, a range that does not match) puts its tile on a list, and twoview_scan_kernel -- launched behind this
// kernel for exactly those tiles -- does them the old way.  Nothing is trusted that is not checked per pixel.
#define TS_MAXD 1024                   // labels a template holds (more: the old kernel)
#define TS_MAXS 6144                   // candidate entries
#define TS_MAXSPAN 1024                // smax - smin + 1
struct ScanTemplate {
	int32_t ok, nS, smin, smax, x0, y0, nF, pad_;
	double tabs, tmin, tmax;            // max |tnum[d]| (bound of |t| for fast_proj_setup), smallest and largest tnum[d]
	// (32-bit entries: a wave reads them with SCALAR loads -- the index is uniform -- eight at a time; bytes and shorts
	// would come through the vector memory path, a dependent round trip per label)
	alignas(32) int32_t lab[TS_MAXD + 8];   // per label: state | offset << 8; state 0 not projectable, 1 first point, 2 dropped
	                                    // (step < 1 pixel), 3 kept, 4 padding behind the last label; offset (first / kept): trunc(x2) - x
	alignas(32) int32_t S[TS_MAXS + 8]; // candidate columns in visiting order, relative to x
	// the FIRST visits only, in visiting order: column offset (24 bits) | how often the sequence visits the column << 24.
	// A column seen earlier in the sequence can never change the running minimum again -- its cost was not below
	// minCost - margin then, and minCost only falls (with the certified scan: the reference makes that comparison on its own
	// exact numbers, for which the argument holds, so it need not be replayed) -- its later visits only count as evaluations.
	alignas(32) int32_t F[TS_MAXSPAN + 8];
};
size_t scan_template_bytes() { return sizeof(ScanTemplate); }

__global__ __launch_bounds__(256)
void twoview_template_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P, int y0, int nrows,
                             const double *__restrict__ tnum, ScanTemplate *__restrict__ tpl)
{
	const ViewDev &L = views[ref];
	const ViewDev &Rv = views[oth];
	__shared__ double sx[TS_MAXD], sy[TS_MAXD];
	__shared__ double s_red[4], s_rmin[4], s_rmax[4];
	const int D = P.num_depth_levels, tid = threadIdx.x;
	const int px = L.w/2, py = y0 + nrows/2;
	if (D > TS_MAXD) { if (tid == 0) tpl->ok = 0; return; }
	const Ray ray = cam_unproject(L.cam, (px + 0.5) / P.image_scale, (py + 0.5) / P.image_scale);
	const Vec3 nrm = normalized(load3(L.cam.pdir));
	const double nd = dot(nrm, ray.dir);
	const SharedDivisor nd_sd = shared_divisor(nd);
	double tabs = 0.0, tmn = __builtin_inf(), tmx = -__builtin_inf();
	for (int d = tid; d < D; d += 256) {
		double x2, y2;
		const bool ok = !(fabs(nd) < 1e-10) && pinhole_project_label_sd(ray, nd_sd, tnum[d], Rv.cam, P.image_scale, x2, y2);
		sx[d] = ok ? x2 : __builtin_nan("");
		sy[d] = ok ? y2 : 0.0;
		tabs = fmax(tabs, fabs(tnum[d]));
		tmn = fmin(tmn, tnum[d]); tmx = fmax(tmx, tnum[d]);
	}
#pragma unroll
	for (int dlt = 1; dlt < 64; dlt <<= 1) {
		tabs = fmax(tabs, __shfl_xor(tabs, dlt)); tmn = fmin(tmn, __shfl_xor(tmn, dlt)); tmx = fmax(tmx, __shfl_xor(tmx, dlt));
	}
	if ((tid & 63) == 0) { s_red[tid >> 6] = tabs; s_rmin[tid >> 6] = tmn; s_rmax[tid >> 6] = tmx; }
	__syncthreads();
	// ---- the keep chain: strictly sequential (a label is compared with the last KEPT one), so one lane walks it -- over
	// LDS only: states and columns of the kept points; the next label's projection is read while this one is decided
	__shared__ int s_lab[TS_MAXD];
	__shared__ int s_seg[TS_MAXD][4];                                 // per kept label: a, b (floor columns relative to px), start in S, cover before: lo | hi<<16 biased
	__shared__ int s_hdr[8];
	if (tid == 0) {
		for (int k = 1; k < 4; ++k) { tabs = fmax(tabs, s_red[k]); tmn = fmin(tmn, s_rmin[k]); tmx = fmax(tmx, s_rmax[k]); }
		int ok = fabs(nd) < 1e-10 ? 0 : 1, nS = 0, nseg = 0, smin = 2147483647, smax = -2147483647;
		double x1 = __builtin_nan(""), y1 = 0.0;
		int jx1 = 0, jy1 = 0;
		double nx = D > 0 ? sx[0] : 0.0, ny = D > 0 ? sy[0] : 0.0;
		for (int d = 0; d < D && ok; ++d) {
			const double x2 = nx, y2 = ny;
			if (d + 1 < D) { nx = sx[d + 1]; ny = sy[d + 1]; }
			if (isnan_d(x2)) { s_lab[d] = 0; continue; }                  // (a projection that succeeds is never NaN: finite cameras, t >= 1e-10)
			if (isnan_d(x1)) {
				// (FLOOR columns: the template pixel's own coordinates may be negative, where the reference's truncation towards
				// zero is one column off the floor; the scan applies the truncation per pixel, see the entries' bit 30)
				x1 = x2; y1 = y2; jx1 = fabs(x2) < 0x1p28 ? (int)floor(x2) : 1 << 30; jy1 = trunc_sat(y2);
				if (jy1 != py || abs(jx1 - px) > 30000) ok = 0;
				s_lab[d] = 1 | ((jx1 - px) << 8);
				continue;
			}
			const double dx = x2 - x1, dy = y2 - y1;
			if (!(dx*dx + dy*dy >= 1)) { s_lab[d] = 2; continue; }
			const int ix0 = jx1, ix1 = fabs(x2) < 0x1p28 ? (int)floor(x2) : 1 << 30, iy1 = trunc_sat(y2);
			if (jy1 != py || iy1 != py || abs(ix1 - px) > 30000) { ok = 0; break; }   // a segment off the row: not this kernel's case
			jx1 = ix1; jy1 = iy1;
			s_lab[d] = 3 | ((ix1 - px) << 8);
			// LineIterator on one row: (a..b, y) in ascending x whatever the direction of the segment (lineiter.hpp:96-111)
			const int sa = (ix0 < ix1 ? ix0 : ix1) - px, sb = (ix0 < ix1 ? ix1 : ix0) - px;
			if (nS + (sb - sa + 1) > TS_MAXS) { ok = 0; break; }
			s_seg[nseg][0] = sa; s_seg[nseg][1] = sb; s_seg[nseg][2] = nS;
			s_seg[nseg][3] = nseg ? ((smin + 32768) | ((smax + 32768) << 16)) : -1;   // columns covered by the earlier segments: one interval (consecutive segments share an end)
			++nseg;
			nS += sb - sa + 1;
			if (sa < smin) smin = sa;
			if (sb > smax) smax = sb;
			x1 = x2; y1 = y2;
		}
		if (nS == 0) { smin = 0; smax = -1; }
		if (nS > 0 && smax - smin + 1 > TS_MAXSPAN) ok = 0;
		s_hdr[0] = ok; s_hdr[1] = nS; s_hdr[2] = nseg; s_hdr[3] = smin; s_hdr[4] = smax;
		tpl->ok = ok; tpl->nS = nS; tpl->smin = smin; tpl->smax = smax; tpl->x0 = px; tpl->y0 = py; tpl->tabs = tabs; tpl->tmin = tmn; tpl->tmax = tmx;
	}
	__syncthreads();
	// ---- everything else by all lanes: the label words, the candidate entries segment by segment.
	// Entry = column offset (29 bits) | bit 30: high end of its segment | bit 29: a revisit (a column seen earlier in the
	// sequence; the first visits alone are listed in F below).
	// (Bit 30: the template's columns are FLOOR offsets; the reference truncates towards zero, which moves a negative end
	// point one column to the right: the in-image part of a segment changes only when its high end is column -1 -- the
	// reference's segment then ends ON column 0.)
	const int ok = s_hdr[0], nS = s_hdr[1], nseg = s_hdr[2], smin = s_hdr[3];
	if (!ok) return;
	for (int d = tid; d < ((D + 7) & ~7); d += 256) tpl->lab[d] = d < D ? s_lab[d] : 4;
	for (int k = tid; k < nseg; k += 256) {
		const int sa = s_seg[k][0], sb = s_seg[k][1], at = s_seg[k][2], cov = s_seg[k][3];
		const int clo = cov == -1 ? 1 : (cov & 0xffff) - 32768, chi = cov == -1 ? 0 : ((cov >> 16) & 0xffff) - 32768;
		for (int c = sa; c <= sb; ++c)
			tpl->S[at + (c - sa)] = (c & 0x1fffffff) | (c == sb ? 0x40000000 : 0) | ((c >= clo && c <= chi) ? 0x20000000 : 0);
	}
	for (int j = nS + tid; j < ((nS + 7) & ~7); j += 256) tpl->S[j] = (smin & 0x1fffffff) | 0x20000000;   // (padding: a column whose mask byte exists; never counted)
	// first visits: of segment k the columns outside what the earlier segments cover -- at most two runs, [sa, clo) and
	// (chi, sb] -- ; their places by a running count over the segments (one lane), the columns' multiplicities by counting
	// the segments that contain them
	__shared__ int s_fat[TS_MAXD];
	__shared__ int s_cnt[TS_MAXSPAN + 8];                             // per column of the span: how many segments contain it
	for (int k = tid; k < TS_MAXSPAN + 8; k += 256) s_cnt[k] = 0;
	__syncthreads();
	for (int k = tid; k < nseg; k += 256) { atomicAdd(&s_cnt[s_seg[k][0] - smin], 1); atomicAdd(&s_cnt[s_seg[k][1] + 1 - smin], -1); }
	__syncthreads();
	if (tid == 0) {
		int nF = 0;
		for (int k = 0; k < nseg; ++k) {
			const int sa = s_seg[k][0], sb = s_seg[k][1], cov = s_seg[k][3];
			const int clo = cov == -1 ? 1 << 20 : (cov & 0xffff) - 32768, chi = cov == -1 ? -(1 << 20) : ((cov >> 16) & 0xffff) - 32768;
			s_fat[k] = nF;
			const int l1 = clo - 1 < sb ? clo - 1 : sb, h2 = chi + 1 > sa ? chi + 1 : sa;
			nF += (l1 >= sa ? l1 - sa + 1 : 0) + (sb >= h2 && cov != -1 ? sb - h2 + 1 : 0);
		}
		s_hdr[5] = nF;
		tpl->nF = nF;
	} else if (tid == 64) {
		int run = 0;                                                   // (another wave: the running sum of the difference array)
		const int span = s_hdr[4] - smin + 1;
		for (int k = 0; k < span; ++k) { run += s_cnt[k]; s_cnt[k] = run; }
	}
	__syncthreads();
	const int nF = s_hdr[5];
	for (int k = tid; k < nseg; k += 256) {
		const int sa = s_seg[k][0], sb = s_seg[k][1], cov = s_seg[k][3];
		const int clo = cov == -1 ? 1 << 20 : (cov & 0xffff) - 32768, chi = cov == -1 ? -(1 << 20) : ((cov >> 16) & 0xffff) - 32768;
		int at = s_fat[k];
		for (int c = sa; c <= sb; ++c) {
			if (c >= clo && c <= chi) continue;                        // a revisit
			const int mult = s_cnt[c - smin];
			tpl->F[at++] = (c & 0xffffff) | ((mult < 127 ? mult : 127) << 24);
		}
	}
	for (int j = nF + tid; j < ((nF + 7) & ~7); j += 256) tpl->F[j] = smin & 0xffffff;   // (padding: multiplicity 0)
}

#define TS_U 8                         // look-ups in flight per lane
// one tile; returns false when the tile is left to twoview_scan_kernel
template <bool CERT>
__device__ __forceinline__
bool twoview_tscan_tile(const int bid, const ViewDev *__restrict__ views, int ref, int oth, const srh_params &P,
                        int y0, int nrows, const double *__restrict__ tnum,
                        const double *__restrict__ cost, int cstride,
                        Counters *__restrict__ cnt, const PixRange *__restrict__ prange,
                        uint32_t *__restrict__ cflag, const CertBound &cb, const double *__restrict__ pexact,
                        const ScanTemplate *__restrict__ tpl, unsigned char *smask, unsigned &n_eval_acc, unsigned &n_pix_acc, unsigned &n_flag_acc)
{
	const ViewDev &L = views[ref];
	const ViewDev &Rv = views[oth];
	const int W = L.w, OW = Rv.w, OH = Rv.h;
	const int tiles_per_row = (W + SC_TW - 1)/SC_TW;
	const int tid = threadIdx.x;
	const int trow = bid / tiles_per_row;
	const int xt = (bid % tiles_per_row)*SC_TW, x = xt + tid;
	const int y = y0 + trow;
	const int D = P.num_depth_levels;
#ifdef SRH_EXPERIMENT
	if (g_exp_scan_mode == 6) return true;                           // timing experiment: the loop alone
#endif
	const int nS = tpl->nS, smin = tpl->smin, smax = tpl->smax;
	const bool active = x < W && L.mask[(size_t)y*W + x] == 1;
	bool good = true;
	Ray ray;
	int lo = 0, hi = -1;
	if (active) {
		ray = cam_unproject(L.cam, (x + 0.5) / P.image_scale, (y + 0.5) / P.image_scale);
		const PixRange pr = prange[(size_t)trow*W + x];
		lo = pr.lo; hi = pr.hi;
		// the range the cost rows were made for must be the template's, clipped to the image
		int wlo = x + smin, whi = x + smax;
		// (truncation towards zero: a segment whose high end is column -1 ends ON column 0 in the reference -- the look-ups
		// below move that entry there -- so a pixel whose template range ends at column -1 needs column 0's cost as well)
		if (whi == -1) whi = 0;
		if (wlo < 0) wlo = 0;
		if (whi > OW - 1) whi = OW - 1;
		if (nS > 0 && whi >= wlo) good = lo <= wlo && hi >= whi;       // (every candidate column has its cost)
		if (y < 0 || y >= OH) good = false;
		const Vec3 nrm = normalized(load3(L.cam.pdir));
		const double nd = dot(nrm, ray.dir);
		if (fabs(nd) < 1e-10) good = false;
		const SharedDivisor nd_sd = shared_divisor(nd);
		bool verify = true;
#ifdef SRH_EXPERIMENT
		verify = g_exp_scan_mode != 3 && g_exp_scan_mode != 5;        // timing experiment: no verification
#endif
		if (verify) {
		const FastProj fp = fast_proj_setup(ray, Rv.cam, (tpl->tabs/fabs(nd))*1.000001);
		// ---- one bound for the pixel instead of one per label.  Every label's t = fl(tnum[d] / nd) lies in [tlo, thi] (the
		// correctly rounded division is monotone; tnum's extremes come with the template).  On that interval k(t) = A + t*B is
		// linear: |k_x|, |k_y| are largest at an end, |k_z| smallest at an end (same sign at both ends: no pole inside), so
		// fast_project's bound e(t) <= eU, its formula evaluated with those extremes.  The y coordinate of the fast form is a
		// Moebius function of t, monotone between the ends: the reference's y2 of EVERY label lies within eU of [ylo, yhi], made
		// from the two end values -- inside [y, y + 1) means every kept point truncates to row y, and any two labels' y2 differ
		// by at most dyU = yhi - ylo.  What is left per label is the x coordinate.
		const double ta = div_by(tpl->tmin, nd_sd), tb = div_by(tpl->tmax, nd_sd);
		const double tlo = fmin(ta, tb), thi = fmax(ta, tb);
		const double kzl = __builtin_fma(tlo, fp.B.z, fp.A.z), kzh = __builtin_fma(thi, fp.B.z, fp.A.z);
		const double kyl = __builtin_fma(tlo, fp.B.y, fp.A.y), kyh = __builtin_fma(thi, fp.B.y, fp.A.y);
		const double kxm = fmax(fabs(__builtin_fma(tlo, fp.B.x, fp.A.x)), fabs(__builtin_fma(thi, fp.B.x, fp.A.x)));
		const double kzmin = fmin(fabs(kzl), fabs(kzh));
		const double rU = (1.0/kzmin)*1.000001;
		const double amU = fmax(kxm, fmax(fabs(kyl), fabs(kyh)))*rU;
		const double eU = __builtin_fma(amU*P.image_scale, 0x1p-49, (fp.ek + amU*fp.ekz)*(rU*P.image_scale*1.002))*1.0001;
		const double yfl = (kyl/kzl)*P.image_scale, yfh = (kyh/kzh)*P.image_scale;
		const double ylo = fmin(yfl, yfh) - 2*eU, yhi = fmax(yfl, yfh) + 2*eU, dyU = yhi - ylo;
		good = good && kzl*kzh > 0.0 && fp.ekz*rU <= 0x1p-10 && eU < 0x1p-20 && ylo >= (double)y && yhi < (double)(y + 1);
		const double se = 2*eU;
		const double c1 = 2.02*se, c0 = __builtin_fma(2.02*se, se, dyU*dyU);   // |dd_reference - dx^2| <= 2|dx|se + se^2 + dyU^2 (+ roundings)
		const double rsc = P.image_scale;
		double x1 = 0.0;
		for (int d0 = 0; d0 < D; d0 += 8) {
			// (uniform addresses, whole eights, aligned: two s_load_dwordx4 and two s_load_dwordx8 per eight labels)
			int lab[8];
			double tn[8];
			{
				const int4 la = reinterpret_cast<const int4 *>(&tpl->lab[d0])[0], lb = reinterpret_cast<const int4 *>(&tpl->lab[d0])[1];
				lab[0] = la.x; lab[1] = la.y; lab[2] = la.z; lab[3] = la.w; lab[4] = lb.x; lab[5] = lb.y; lab[6] = lb.z; lab[7] = lb.w;
				const double4 ta4 = reinterpret_cast<const double4 *>(&tnum[d0])[0], tb4 = reinterpret_cast<const double4 *>(&tnum[d0])[1];
				tn[0] = ta4.x; tn[1] = ta4.y; tn[2] = ta4.z; tn[3] = ta4.w; tn[4] = tb4.x; tn[5] = tb4.y; tn[6] = tb4.z; tn[7] = tb4.w;
			}
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const int st = lab[u] & 255, off = lab[u] >> 8;
				if (st == 4) continue;
				const double t = div_by(tn[u], nd_sd);                     // the reference's own t (pinhole_project_label_sd)
				const bool tv = !(t < 1e-10);
				if (st == 0) { good = good && !tv; continue; }
				// fast_project's x coordinate (srh_walk.hpp): within eU of the reference's
				const double kx = __builtin_fma(t, fp.B.x, fp.A.x), kz = __builtin_fma(t, fp.B.z, fp.A.z);
				double r = __builtin_amdgcn_rcp(kz);
				r = __builtin_fma(r, __builtin_fma(-kz, r, 1.0), r);
				r = __builtin_fma(r, __builtin_fma(-kz, r, 1.0), r);
				const double x2 = kx*(r*rsc);
				const double dx = x2 - x1;
				const double dd = dx*dx;
				const bool certain = fabs(dd - 1.0) > __builtin_fma(dd, 0x1p-48, __builtin_fma(c1, fabs(dx), c0));
				if (st == 2) { good = good && tv && certain && !(dd >= 1); continue; }
				// first / kept point: x truncates certainly and to the template's column.  The reference truncates towards zero:
				// left of the image (x2 < 0) its integer is the template's (floor) column + 1, which changes the in-image part of a
				// segment only when the segment's high end is column -1 (the look-up loop handles that case)
				const int ti = (int)x2;
				const bool colok = ti == x + off + (x2 < 0.0 ? 1 : 0);
				good = good && tv && (st == 1 || (certain && dd >= 1)) && trunc_certain(x2, eU) && colok;
				x1 = x2;
			}
		}
		}
	}
	if (__any(active && !good)) return false;                        // twoview_scan_kernel does this tile

	// mask bytes of row y of the other view, columns [xt + smin, xt + SC_TW + smax] -- one more than the offsets reach: the
	// left-border look-ups move an entry from column -1 to column 0, which for the tile's last pixel (x + smax = -1) is
	// the byte behind its range
	const int mbase = xt + smin, mlen = nS > 0 ? SC_TW + (smax - smin) + 1 : 0;
	for (int k = tid; k < mlen; k += SC_TW) {
		const int tx = mbase + k;
		smask[k] = ((unsigned)tx < (unsigned)OW) ? Rv.mask[(size_t)y*OW + tx] : (unsigned char)0;
	}
	__syncthreads();

	unsigned n_eval = 0, n_flag = 0;
	double depth = __builtin_nan("");
	if (active) {
		n_pix_acc += 1;
		const double *crow = cost + ((size_t)trow*((W + DC_TP - 1)/DC_TP) + (x/DC_TP))*(size_t)cstride*DC_TP + (x % DC_TP);
		double minCost = __builtin_inf(), secondBest = __builtin_inf();
		int wcol = -1;
		bool px_sure = false;
		if (CERT && pexact) { const double *pc = pexact + ((size_t)trow*W + x)*SRH_PC; px_sure = cert_pixel_exact(cb, pc[2], pc[3]); }
		const int xm = tid - smin;                                     // smask index of column x + s: tid + (s - smin)
		const int xlo = x - lo;                                        // cost-row entry of column x + s: s + xlo
		int nSe = nS;
#ifdef SRH_EXPERIMENT
		if (g_exp_scan_mode == 4 || g_exp_scan_mode == 5) nSe = 0;     // timing experiment: no look-ups
#endif
		// eight template entries at a time: the entries by two scalar loads, the eight mask bytes by LDS reads in flight
		// together, the eight costs by loads in flight together, then the reference's running-min rule entry by entry.
		// LEFT (tile-uniform): a column of the tile may be -1 -- see the template: a segment whose high end is column -1
		// ends on column 0 in the reference (truncation towards zero)
		// LEFT tiles walk the whole sequence S (every visit evaluated: the shifted entry may be a column's first visit);
		// every other tile the first visits F, a later visit of a column counted with its first
		auto lookups = [&](auto left_c) {
			constexpr bool LEFT = decltype(left_c)::value;
			const int32_t *seq = LEFT ? tpl->S : tpl->F;
			const int nE = nSe == 0 ? 0 : (LEFT ? nS : tpl->nF);
			for (int j0 = 0; j0 < nE; j0 += TS_U) {
				int sv[TS_U], kk[TS_U], mu[TS_U];
				unsigned char mb[TS_U];
				double c[TS_U];
				{
					const int4 sa = reinterpret_cast<const int4 *>(&seq[j0])[0], sb = reinterpret_cast<const int4 *>(&seq[j0])[1];
					sv[0] = sa.x; sv[1] = sa.y; sv[2] = sa.z; sv[3] = sa.w; sv[4] = sb.x; sv[5] = sb.y; sv[6] = sb.z; sv[7] = sb.w;
				}
#pragma unroll
				for (int u = 0; u < TS_U; ++u) {
					int s;
					if (LEFT) {
						s = (sv[u] << 3) >> 3;                                 // (sign-extended 29-bit offset; the padding behind the last entry is smin)
						s += ((sv[u] & 0x40000000) && x + s == -1) ? 1 : 0;
						mu[u] = 1;
					} else {
						s = (sv[u] << 8) >> 8;                                 // (24-bit offset)
						mu[u] = sv[u] >> 24;                                   // (uniform: how often the reference evaluates this column)
					}
					sv[u] = s;
					mb[u] = smask[xm + s];
				}
#pragma unroll
				for (int u = 0; u < TS_U; ++u) {
					const bool white = j0 + u < nE && mb[u] == 1;              // (off-image columns hold 0)
					n_eval += white ? (unsigned)mu[u] : 0u;
					kk[u] = white ? xlo + sv[u] : -1;
					c[u] = crow[(unsigned)(white ? kk[u] : 0)*(unsigned)DC_TP];   // (unconditional: no branch around a load; entry 0 exists)
				}
#pragma unroll
				for (int u = 0; u < TS_U; ++u) {
					const bool on = kk[u] >= 0;
					const double cv = c[u];
					const double t = on ? cv + P.wta_margin : __builtin_inf();
					if (CERT) {
						// (twoview_scan_kernel<true, false>'s test; the comparison is decided beyond the bound almost always: the rest
						// of the test only when some lane's is not)
						const bool nearby = on && !(fabs(t - minCost) > 2.5*cb.e0);
						if (__any(nearby)) {
							if (nearby && kk[u] != wcol &&
							    !(cert_sure(cv, P.max_color_diff, cb.m_hi) && cert_sure(minCost, P.max_color_diff, cb.m_hi)) &&
							    !(px_sure && cv == cv)) n_flag = 1;
						}
					}
					if (t < minCost) { secondBest = minCost; minCost = cv; wcol = kk[u]; }   // twoviewstereo.cpp:293-301
				}
			}
		};
		// (first visits only: sound while wta_margin >= 0 -- with a negative margin a revisited winner would "beat" itself and
		// move secondBest -- so a negative margin walks every visit like the left-border tiles)
		if (xt + smin < 0 || !(P.wta_margin >= 0.0)) lookups(std::true_type()); else lookups(std::false_type());
		if (wcol >= 0) depth = candidate_depth(L.cam, Rv.cam, P, ray, lo + wcol, y);
		if (minCost > P.second_best_factor*secondBest) depth = __builtin_inf();
		if (CERT && wcol >= 0) {
			const double rhs = P.second_best_factor*secondBest;
			const double tol = __builtin_fma(fmin(fabs(rhs), 1e300), 1e-15, (1.0 + fabs(P.second_best_factor))*cb.e0);
			if (!(fabs(minCost - rhs) > tol) && !px_sure &&
			    !(cert_sure(minCost, P.max_color_diff, cb.m_hi) && cert_sure(secondBest, P.max_color_diff, cb.m_hi))) n_flag = 1;
		}
		if (CERT && n_flag) cflag[1 + atomicAdd(&cflag[0], 1u)] = (uint32_t)((size_t)trow*W + x);
	}
	if (x < W) L.depth[(size_t)y*W + x] = depth;
	n_eval_acc += n_eval; n_flag_acc += n_flag;
	return true;
}

// PERSISTENT: a few thousand one-wave workgroups share the band's tiles in a grid-stride loop (one workgroup per tile --
// 32 400 of them on C3 -- costs 0.8 ms of workgroup launches whatever the tiles do: the fixed part of twoview_scan_kernel);
// the counters travel in registers and are added once per workgroup.  A tile that does not verify goes on `tilelist`
// = [count | tile indices] for twoview_scan_kernel.
template <bool CERT>
__global__ __launch_bounds__(SC_TW, 4)
void twoview_tscan_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P,
                          int y0, int nrows, const double *__restrict__ tnum,
                          const double *__restrict__ cost, int cstride,
                          Counters *__restrict__ cnt, const PixRange *__restrict__ prange,
                          uint32_t *__restrict__ cflag, CertBound cb, const double *__restrict__ pexact,
                          const ScanTemplate *__restrict__ tpl, uint32_t *__restrict__ tilelist)
{
	__shared__ unsigned char smask[SC_TW + TS_MAXSPAN + 16];
	const int W = views[ref].w;
	const int ntiles = ((W + SC_TW - 1)/SC_TW)*nrows;
	const int tid = threadIdx.x;
	const bool tok = tpl->ok != 0;
	unsigned n_eval = 0, n_pix = 0, n_flag = 0, n_tpl = 0, n_walk = 0;
	for (int bid = blockIdx.x; bid < ntiles; bid += gridDim.x) {
		const bool done = tok && twoview_tscan_tile<CERT>(bid, views, ref, oth, P, y0, nrows, tnum, cost, cstride, cnt, prange, cflag, cb, pexact,
		                                                 tpl, smask, n_eval, n_pix, n_flag);
		if (!done) { if (tid == 0) tilelist[1 + atomicAdd(&tilelist[0], 1u)] = (uint32_t)bid; ++n_walk; } else ++n_tpl;
		__syncthreads();                                              // (smask is reused)
	}
	if (!cnt) return;
	n_eval = wave_sum_u32(n_eval); n_pix = wave_sum_u32(n_pix); n_flag = wave_sum_u32(n_flag);
	if (tid == 0) {
		if (n_eval) atomicAdd(&cnt->n_eval, (unsigned long long)n_eval);
		if (n_pix) atomicAdd(&cnt->n_pixels, (unsigned long long)n_pix);
		if (CERT && n_pix) atomicAdd(&cnt->n_certified, (unsigned long long)n_pix);
		if (CERT && n_flag) atomicAdd(&cnt->n_flagged, (unsigned long long)n_flag);
		if (n_tpl) atomicAdd(&cnt->scan_tiles_template, (unsigned long long)n_tpl);
		if (n_walk) atomicAdd(&cnt->scan_tiles_walked, (unsigned long long)n_walk);
	}
}

// The exact scan of the FLAGGED pixels, one WAVE per pixel.  twoview_scan_kernel<false, true> gives a flagged pixel one lane:
// a handful of pixels then cost a whole pixel's serial walk (256 dependent label projections, every cost a dependent
// load: 0.25 ms on C3, whatever their number).  Here the wave splits the pixel's work: (1) the label projections, one
// label per lane; (2) the other view's mask bytes of the pixel's column range; (3) ONE lane replays the reference's walk
// over the projected points and writes the candidate columns in visiting order (twoviewstereo.cpp:999-1054: the cheap,
// strictly sequential part); (4) all lanes fetch the candidates' costs; (5) one lane runs the running-min rule
// (twoviewstereo.cpp:293-305) over them.  Same functions, same order, same bits as the one-lane form.
#define RSW_MAXD 1024                  // labels (else: the one-lane form)
#define RSW_MAXC 6144                  // candidates of a pixel (more: Counters::cert_overflow, the pass is repeated in mode 0)
__global__ __launch_bounds__(64)
void twoview_rescan_wave_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P, int y0,
                                const double *__restrict__ tnum, const double *__restrict__ cost, int cstride,
                                const PixRange *__restrict__ prange, const uint32_t *__restrict__ cflag, int cap,
                                Counters *__restrict__ cnt)
{
	const ViewDev &L = views[ref];
	const ViewDev &Rv = views[oth];
	const int W = L.w, OW = Rv.w, OH = Rv.h;
	const int lane = threadIdx.x;
	__shared__ double sx[RSW_MAXD], sy[RSW_MAXD];
	__shared__ unsigned short scol[RSW_MAXC];
	__shared__ double scost[RSW_MAXC];
	__shared__ unsigned char smask[4096 + 64];
	__shared__ int s_n;
	const uint32_t nflag = cflag[0];
	for (uint32_t f = blockIdx.x; f < nflag && f < (uint32_t)cap; f += gridDim.x) {
		const uint32_t q = cflag[1 + f];
		const int x = (int)(q % (uint32_t)W), trow = (int)(q / (uint32_t)W), y = y0 + trow;
		const PixRange pr = prange[q];
		const int lo = pr.lo, hi = pr.hi;
		const Ray ray = cam_unproject(L.cam, (x + 0.5) / P.image_scale, (y + 0.5) / P.image_scale);
		const Vec3 nrm = normalized(load3(L.cam.pdir));
		const double nd = dot(nrm, ray.dir);
		const bool walk = !(fabs(nd) < 1e-10);
		// (1) label projections
		if (walk) {
			const SharedDivisor nd_sd = shared_divisor(nd);
			for (int d = lane; d < P.num_depth_levels; d += 64) {
				double x2, y2;
				const bool ok = pinhole_project_label_sd(ray, nd_sd, tnum[d], Rv.cam, P.image_scale, x2, y2);
				sx[d] = ok ? x2 : __builtin_nan("");              // (a projection is never NaN when it succeeds: t >= 1e-10, finite cameras;
				sy[d] = ok ? y2 : 0.0;                             //  a NaN x2 from a degenerate camera is skipped by the one-lane form's isnan test too)
			}
		}
		// (2) mask bytes of row y of the other view over the pixel's column range
		const int span = hi >= lo ? hi - lo + 1 : 0;
		for (int k = lane; k < span && k < 4096; k += 64) smask[k] = (y >= 0 && y < OH) ? Rv.mask[(size_t)y*OW + lo + k] : 0;
		__syncthreads();
		// (3) the walk: candidate columns in visiting order
		if (lane == 0) {
			int n = 0;
			bool over = false;
			if (walk) {
				double x1 = __builtin_nan(""), y1 = __builtin_nan("");
				for (int d = 0; d < P.num_depth_levels && !over; ++d) {
					const double x2 = sx[d], y2 = sy[d];
					if (isnan_d(x2)) continue;                       // pinhole_project_label_sd failed
					if (isnan_d(x1)) { x1 = x2; y1 = y2; continue; }
					const double dx = x2 - x1, dy = y2 - y1;
					if (!(dx*dx + dy*dy >= 1)) continue;
					const int ix0 = trunc_sat(x1), iy0 = trunc_sat(y1), ix1 = trunc_sat(x2), iy1 = trunc_sat(y2);
					const int a = ix0 < ix1 ? ix0 : ix1, b = ix0 < ix1 ? ix1 : ix0;
					if (iy0 == y && iy1 == y && a >= lo && b <= hi) {
						for (int tx = a; tx <= b; ++tx) {
							const int k = tx - lo;
							const bool white = k < 4096 ? (smask[k] == 1) : (Rv.mask[(size_t)y*OW + tx] == 1);
							if (white) { if (n < RSW_MAXC) scol[n] = (unsigned short)k; else over = true; ++n; }
						}
					} else {
						LineWalk lw;
						lw.begin(ix0, iy0, ix1, iy1, OW, OH);
						while (lw.has_next()) {
							int tx, ty;
							lw.current(tx, ty);
							if (tx >= 0 && ty >= 0 && tx < OW && ty < OH && Rv.mask[(size_t)ty*OW + tx] == 1 &&
							    ty == y && tx >= lo && tx <= hi) {           // (off-row candidates refute the dense plan: counted by the first scan)
								if (n < RSW_MAXC) scol[n] = (unsigned short)(tx - lo); else over = true;
								++n;
							}
							lw.next();
						}
					}
					x1 = x2; y1 = y2;
				}
			}
			if (over) { atomicAdd(&cnt->cert_overflow, 1ull); n = RSW_MAXC; }
			s_n = n;
		}
		__syncthreads();
		const int n = s_n;
		// (4) the candidates' costs
		const double *crow = cost + ((size_t)trow*((W + DC_TP - 1)/DC_TP) + (x/DC_TP))*(size_t)cstride*DC_TP + (x % DC_TP);
		for (int k = lane; k < n; k += 64) scost[k] = crow[(size_t)scol[k]*DC_TP];
		__syncthreads();
		// (5) running minimum, ratio test, depth of the winner
		if (lane == 0) {
			double minCost = __builtin_inf(), secondBest = __builtin_inf();
			int wcol = -1;
			for (int k = 0; k < n; ++k) {
				const double cv = scost[k];
				if (cv + P.wta_margin < minCost) { secondBest = minCost; minCost = cv; wcol = scol[k]; }   // twoviewstereo.cpp:293-301
			}
			double depth = __builtin_nan("");
			if (wcol >= 0) depth = candidate_depth(L.cam, Rv.cam, P, ray, lo + wcol, y);
			if (minCost > P.second_best_factor*secondBest) depth = __builtin_inf();
			L.depth[(size_t)y*W + x] = depth;
		}
		__syncthreads();
	}
}

// cflag == nullptr: the exact scan.  cflag, nlist < 0: the certified scan (flags into cflag).  cflag, nlist >= 0: the
// exact scan of the nlist pixels listed in cflag[1..].
void launch_scan_template(hipStream_t st, const ViewDev *views, int ref, int oth, const srh_params &P, int y0, int nrows,
                          const double *tnum, void *tpl)
{
	hipLaunchKernelGGL(twoview_template_kernel, dim3(1), dim3(256), 0, st, views, ref, oth, P, y0, nrows, tnum, (ScanTemplate *)tpl);
}

void launch_twoview_scan(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                         int y0, int nrows, const double *tnum, const double *cost, int cstride,
                         Counters *cnt, const PixRange *prange, uint32_t *cflag, int nlist, const double *pexact,
                         const void *tpl, uint32_t *tilelist, int num_cus)
{
	const int tiles = (width + SC_TW - 1)/SC_TW;
	const CertBound cb = cert_bound(P);
	const ScanTemplate *tp = (const ScanTemplate *)tpl;
	if (!tp) tilelist = nullptr;
	// template scan: persistent one-wave workgroups (eight per SIMD); the tiles it leaves: a grid-stride launch over its list
	const unsigned pgrid = (unsigned)std::min<long long>((long long)tiles*nrows, (long long)num_cus*4*8);
	const unsigned wgrid = tp ? (unsigned)std::min<long long>((long long)tiles*nrows, (long long)num_cus*4*4) : (unsigned)(tiles*nrows);
	if (tp) (void)hipMemsetAsync(tilelist, 0, sizeof(uint32_t), st);
	if (!cflag) {
		if (tp) hipLaunchKernelGGL((twoview_tscan_kernel<false>), dim3(pgrid), dim3(SC_TW), 0, st,
		                           views, ref, oth, P, y0, nrows, tnum, cost, cstride, cnt, prange, nullptr, cb, nullptr, tp, tilelist);
		hipLaunchKernelGGL((twoview_scan_kernel<false, false>), dim3(wgrid), dim3(SC_TW), 0, st,
		                   views, ref, oth, P, y0, nrows, tnum, cost, cstride, cnt, prange, nullptr, 0, cb, nullptr, tilelist);
	} else if (nlist < 0) {
		if (tp) hipLaunchKernelGGL((twoview_tscan_kernel<true>), dim3(pgrid), dim3(SC_TW), 0, st,
		                           views, ref, oth, P, y0, nrows, tnum, cost, cstride, cnt, prange, cflag, cb, pexact, tp, tilelist);
		hipLaunchKernelGGL((twoview_scan_kernel<true, false>), dim3(wgrid), dim3(SC_TW), 0, st,
		                   views, ref, oth, P, y0, nrows, tnum, cost, cstride, cnt, prange, cflag, 0, cb, pexact, tilelist);
	} else if (nlist > 0 && P.num_depth_levels <= RSW_MAXD && cstride <= 4096)
		// (the list may hold up to the band's pixels: workgroups share it in a grid-stride loop, the count is read on the device)
		hipLaunchKernelGGL(twoview_rescan_wave_kernel, dim3((unsigned)(nlist < 2048 ? nlist : 2048)), dim3(64), 0, st,
		                   views, ref, oth, P, y0, tnum, cost, cstride, prange, cflag, nlist, cnt);
	else if (nlist > 0)
		hipLaunchKernelGGL((twoview_scan_kernel<false, true>), dim3((unsigned)((nlist + SC_TW - 1)/SC_TW)), dim3(SC_TW), 0, st,
		                   views, ref, oth, P, y0, nrows, tnum, cost, cstride, cnt, prange, cflag, nlist, cb, nullptr, nullptr);
}

} // namespace srh
