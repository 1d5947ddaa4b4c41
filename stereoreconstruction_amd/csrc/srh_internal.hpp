// srh_internal.hpp -- shared between the C-ABI host layer (srh_api.hip) and the
// gfx950 kernels (srh_kernels.hip).  Not installed.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stereo_recon_hip.h"

namespace srh {

// One view as the kernels see it (device pointers).  HBM layout, all row-major:
//   rgba    u32  R | G<<8 | B<<16 | A<<24           4 B/pixel  (weights: colour distances)
//   mask    u8   1 <=> mask.pixel == WHITE           1 B/pixel  (curve candidates, reference pixels)
//   gray    f64  toGray(pixel)                       8 B/pixel  (MVS taps: pixel(), every in-bounds pixel)
//   gray_tv f64  toGray where the TwoView tap test   8 B/pixel  (mask WHITE && sample() valid:
//                passes, NaN elsewhere                           x+1<w && y+1<h)
//   depth   f64  result map                          8 B/pixel
struct ViewDev {
	int32_t w, h;
	const uint32_t *rgba;
	const uint8_t  *mask;
	const double   *gray;
	const double   *gray_tv;
	double         *depth;
	srh_camera      cam;
};

// Support-window buffer of one row band: tile-major, a tile = SRH_WTILE consecutive
// pixels of one image row holding T taps: wbuf[tile][tap][SRH_WTILE] (31 KB contiguous
// per tile at r=5).  A pixel's window is  base + tap*SRH_WTILE.
#define SRH_WTILE 32
__host__ __device__ inline size_t wbuf_doubles(int W, int nrows, int T) {
	return (size_t)nrows*((W + SRH_WTILE - 1)/SRH_WTILE)*SRH_WTILE*T;
}
__host__ __device__ inline size_t wbuf_offset(int W, int T, int band_row, int x) {
	const size_t tile = (size_t)band_row*((W + SRH_WTILE - 1)/SRH_WTILE) + x/SRH_WTILE;
	return tile*(size_t)T*SRH_WTILE + (x % SRH_WTILE);
}

// ---- the strip kernel's inputs (srh_strip.hip) ------------------------------------------------------------
// Window buffer, layout B ("LDS image"): wbuf[tile][row][pixel 0..31][WP], WP = taps per row padded to an even
// count -- exactly the bytes the strip kernel keeps in LDS, so a tile moves global -> LDS as one linear
// LDS-DMA copy (global_load_lds_dwordx4 writes lane-linear destinations only), and the 32 pixels' taps of one
// window row are 3 KB of contiguous bytes for the weights kernel's stores.  Tap (row, col) of a pixel is
// base + row*wimg_row_stride(R) + col; the pad tap is never read.
__host__ __device__ inline int wimg_wp(int R) { return ((2*R + 1) + 1) & ~1; }
__host__ __device__ inline int wimg_row_stride(int R) { return SRH_WTILE*wimg_wp(R); }
__host__ __device__ inline size_t wimg_doubles(int W, int nrows, int R) {
	return (size_t)nrows*((W + SRH_WTILE - 1)/SRH_WTILE)*SRH_WTILE*(size_t)((2*R + 1)*wimg_wp(R));
}
__host__ __device__ inline size_t wimg_offset(int W, int R, int band_row, int x) {
	const size_t tile = (size_t)band_row*((W + SRH_WTILE - 1)/SRH_WTILE) + x/SRH_WTILE;
	return tile*SRH_WTILE*(size_t)((2*R + 1)*wimg_wp(R)) + (size_t)(x % SRH_WTILE)*wimg_wp(R);
}
// NaN-bordered copy of a view's gray_tv plane (and zero-bordered copy of its "window fully usable" plane):
// every row piece a tile stages lies inside the allocation, so the LDS-DMA needs no bounds handling and the
// border taps arrive as NaN (= "tap skipped", vectorimage.cpp:115-119,129-155).
#define SRH_PADL 8
#define SRH_PADR 344          // >= CHUNK + R + NCB + 2 of the strip kernel
#define SRH_PADY 8
__host__ __device__ inline int padded_stride(int w) { return w + SRH_PADL + SRH_PADR; }
__host__ __device__ inline size_t padded_size(int w, int h) { return (size_t)padded_stride(w)*(size_t)(h + 2*SRH_PADY); }
// candidate column range of a reference pixel on its own row of the other view (empty: hi < lo)
struct PixRange { int32_t lo, hi; };

// ---- certified fused arithmetic (option "arith" = 3; derivation: DESIGN.md section 2b) ---------------------------
// The dense cost loops run with fused multiply-adds (half the FP64 instructions); a depth map depends on costs only
// through comparisons (cost + wta_margin < minCost, minCost > second_best_factor*secondBest, the max_color_diff clamp;
// twoviewstereo.cpp:292-305, 976), so the fused costs may stand in for the reference's wherever no comparison's
// margin is inside the error bound.  For a fast-form candidate (all T taps usable; weights in (0,1], grays in
// [0,255]; meanL, totalWeight and sum2 are the SAME bits in both arithmetics), with u = 2^-53, gamma_k = k*u/(1-k*u),
// G = 256, A = sqrt(sum2), B = sqrt(sum3):
//     eps_b = gamma_(T+4)*G   (error of w*g_R - meanR: the mean's T-term sum and division, the product, the subtraction)
//     eps_a = 2*u*G           (error of w*g_L - meanL: meanL is shared)
//     |cost_fused - cost_exact| <= 2*255*1.01*(3*eps_b*sqrt(T)/B + eps_a*sqrt(T)/A + 2*gamma_T) + 2600*u
//                                =  k1/B + k2/A + k3
// (either arithmetic is within half of that of the real-number value of the formula; 1.01 covers every second-order
// term once A > 0.3 and B > 60, which the thresholds below imply).  A candidate is CERTIFIED when that is <= e0, i.e.
// when sum3 >= sigma3(sum2); the cost kernel stores NaN for the others, and the certified scan flags every pixel
// that meets a NaN or a comparison whose two sides are closer than the sum of their bounds: flagged pixels are
// re-evaluated in the reference's arithmetic (twoview_refill_kernel + the exact scan).
// ONE-PASS form (strip kernel, AR = 5).  Nothing obliges the fused kernel to follow the reference's two sweeps: any
// arithmetic within e0 of the reference's will do.  With c_t = (w_t*l_t - meanL)*w_t, d_t = w_t^2, q = r^2 it accumulates
// P = sum w_t r_t, Q = sum c_t r_t, U = sum d_t q_t in ONE sweep over the window (3 fused multiply-adds per tap and
// candidate instead of 4, and 40 instead of 69 LDS values per window row and block), and finishes with m = P/tw,
// sum3 = U - m*(2P - T*m), sum1 = Q - m*SA (SA = sum of w_t*l_t - meanL).  The price is cancellation in sum3: with
// Q3 = U + 2mP + T*m^2 (every term >= 0) and z^2 = Q3/sum3, the one-pass value is within
//     255*1.01*(3*gamma_(T+4)*z + 2.002*gamma_(T+3)*z^2) + 1300u
// of the real-number cost, the reference's arithmetic within half of k1/B + k2/A + k3: a candidate is
// certified when sum3 >= sigma3(sum2) (the latter <= e0/2) and Q3 <= zmax2*sum3 (the former <= e0/2).
// Round 6: the FINISH is free of IEEE divisions and of the IEEE square root as well (onepass_finish below: they were 77 of the
// ~93 instructions a candidate's finish took, a sixth of a block).  m = P*itw with itw = fl(1/tw) from the weights kernels
// (one more rounding on m: every gamma index of the derivation goes up by one; the code takes gamma_(T+6) for both terms),
// 1/sqrt(sum2*sum3) by v_rsq_f64 and ONE third-order Newton step whose own residual e = 1 - x*y0^2 is part of the
// certificate (|e| <= 2^-20, whatever the seed's accuracy: y1 = y0*(1 + e/2 + 3e^2/8) is then within 2u of 1/sqrt(x)), and
// v = fma(-255, |sum1|*y1, 255): x, the product and the last fma round once each, so the finish's share of the bound is
// 255*1.001*(0.5u + 2u + u) + 255u < 1150u; the constant stays at 2600u.
struct CertBound {
	double e0, m_hi, k1, k2, room;
	double zmax2;
	int ok;                                 // 0: the parameters leave the bound no room / weights are not in (0,1]: exact arithmetic
	__host__ __device__ inline double sigma3(double s2) const {
		const double d = room - k2/sqrt(s2);                      // (NaN or zero sum2: d is NaN or -inf)
		if (!(d > 0)) return __builtin_inf();
		const double b = k1/d;
		return b*b*(1.0 + 0x1p-20);                               // (+ the roundings of this very computation)
	}
};
// certified scan: is this stored cost the very number the reference's arithmetic gives?  (+inf: the initial minCost / secondBest)
__host__ __device__ inline bool cert_sure(double x, double clamp, double m_hi) { return x == clamp || x > m_hi; }
// certified cost kernels with the in-kernel redo (strip kernel): a pixel with unusable taps of its own (every candidate in
// a select form) or whose own window leaves the bound no room (sigma3 = +inf) is evaluated in the reference's arithmetic
// throughout -- the certified scan treats every stored cost of such a pixel as the reference's very number
__host__ __device__ inline bool cert_pixel_exact(const CertBound &cb, double sum2, double all_taps_usable) {
	return !(all_taps_usable != 0.0) || !(cb.sigma3(sum2) < __builtin_inf());
}
// The finish of a one-pass candidate (strip, per-tile and row-run cost kernels): the sums recovered from P, Q, U, the cost
// without an IEEE division or square root, and the certificate.  itw = fl(1/totalWeight) (pconst slot 3), TT = T.
// A non-positive or NaN sum3 gives a NaN residual and fails the certificate; an uncertified value is never used.
#ifdef __HIPCC__
__device__ __forceinline__ double onepass_finish(double P, double Q, double U, double SA, double itw, double s2, double TT,
                                                 double sig3, double zmax2, bool &okc)
{
	const double m = P*itw, p2 = P + P;
	const double s3 = __builtin_fma(-m, __builtin_fma(-TT, m, p2), U);    // U - m*(2P - T*m)
	const double s1 = __builtin_fma(-m, SA, Q);
	const double q3 = __builtin_fma(m, __builtin_fma(TT, m, p2), U);      // U + m*(2P + T*m)
	const double x = s2*s3;
	const double y0 = __builtin_amdgcn_rsq(x);                            // v_rsq_f64: a seed, trusted for nothing
	const double e = __builtin_fma(-(x*y0), y0, 1.0);                     // 1 - x*y0^2
	const double y1 = __builtin_fma(y0*e, __builtin_fma(0.375, e, 0.5), y0);
	const bool c1 = s3 >= sig3, c2 = s3*zmax2 >= q3, c3 = __builtin_fabs(e) <= 0x1p-20;
	okc = c1 & c2 & c3;                                                   // (no short circuit: three compares, no branch)
	return __builtin_fma(-255.0, __builtin_fabs(s1)*y1, 255.0);
}
#endif
// mvs: the free cost_ncc of MultiViewStereo (multiviewstereo.cpp:113-189): the score sum1/sqrt(sum2*sum3) itself (no factor
// 255, no clamp), 25 taps at the reference's radius; e0 = 2^-36 there (scores live in [-1, 1])
inline CertBound cert_bound(const srh_params &P, bool mvs = false) {
	const double u = 0x1p-53, G = 256.0;
	const int T = (2*P.window_radius + 1)*(2*P.window_radius + 1);
	auto gamma = [&](int k) { return k*u/(1.0 - k*u); };
	const double eps_b = gamma(T + 4)*G, eps_a = 2*u*G, rt = sqrt((double)T);
	const double scale = mvs ? 1.0 : 255.0;
	CertBound c;
	c.e0 = mvs ? 0x1p-36 : 0x1p-26;                               // TwoView: 1.5e-8, a comparison needs 3.7e-8 of margin (costs live in [0, 120])
	c.k1 = 2*scale*1.01*3*eps_b*rt;
	c.k2 = 2*scale*1.01*eps_a*rt;
	const double k3 = 2*scale*1.01*2*gamma(T) + (mvs ? 40 : 2600)*u;
	c.room = c.e0 - k3;
	c.m_hi = P.max_color_diff + c.e0;
	{	// one-pass form: 255*1.01*g*(3z + 2.002 z^2) + 2600u = e0/2  (g = gamma_(T+6): the finish multiplies by fl(1/tw))
		const double g = gamma(T + 6), rhs = (c.e0/2 - 2600*u)/(scale*1.01*g);
		const double z = rhs > 0 ? (-3.0 + sqrt(9.0 + 4*2.002*rhs))/(2*2.002) : 0.0;
		c.zmax2 = z*z*(1.0 - 0x1p-20);
	}
	const bool weights_ok = P.weight_kind == SRH_WEIGHT_GEODESIC ? P.geodesic_sigma > 0 : P.adaptive_color_sigma > 0;   // exp(-distance/sigma) <= 1
	if (mvs) { c.ok = c.room > 0 && weights_ok && fabs(P.peak_threshold) <= 1e5; return c; }
	// the clamp test needs max_color_diff + e0 > max_color_diff (so below 2^22); the duplicate rule of the scan needs
	// wta_margin >= 0
	// (and magnitudes <= 1e5, so that the rounding of cost + margin stays below e0/2 in the scan's tolerance)
	c.ok = c.room > 0 && c.m_hi > P.max_color_diff && P.max_color_diff <= 1e5 && fabs(P.bad_ret) <= 1e5 &&
	       P.wta_margin >= 0 && P.wta_margin <= 1e5 && fabs(P.second_best_factor) <= 1e5 && weights_ok;
	return c;
}

// Work counters accumulated by the kernels (device memory, zeroed per run).
struct Counters {
	unsigned long long n_pixels;
	unsigned long long n_eval;
	unsigned long long n_eval_device;
	unsigned long long not_row_aligned;   // pixels whose curve leaves their own row
	unsigned long long n_listed, n_slots; // row-run lists: distinct candidates / cost slots (8-column blocks) they occupy
	unsigned long long strip_overflow;    // strip kernel: tiles whose candidate range does not fit one LDS chunk
	unsigned long long n_certified, n_flagged;   // certified scan: reference pixels scanned on fused costs / flagged for the exact redo
	unsigned long long cert_overflow;     // certified redo: a flagged pixel with more candidates than the wave rescan holds (the pass is repeated in mode 0)
	unsigned long long scan_tiles_template, scan_tiles_walked;   // dense TwoView scan: 64-pixel tiles settled by the template scan / left to the per-pixel walk
	unsigned long long mvs_waves_staged, mvs_waves_listed;   // MultiViewStereo walk kernel: waves (with candidates) left to the staged / the gathering cost kernel
	unsigned int strip_ticket, strip_pad; // strip kernel: work-item counter of the current launch
	unsigned long long dbg_cycles, dbg_blocks, dbg_total_cycles, dbg_waves;   // SRH_DENSE_DBG=2 instrumentation
	unsigned long long dbg_phase[8];
	unsigned long long dbg_wave[64];      // diagnostic build: phase k of wave w of a workgroup at [8*k + w]
};

// Neighbour views of one MultiViewStereo estimate, passed to the kernels by value
// (NUM_NEIGHBOURING_VIEWS is 3 in the reference, multiviewstereo.cpp:97; srh_params.num_neighbours is a
// run-time parameter here, up to SRH_MAX_NEIGH).
#define SRH_MAX_NEIGH 8
struct NeighList { int32_t n[SRH_MAX_NEIGH]; };
inline NeighList make_neigh_list(const int32_t *neigh, int nneigh) {
	NeighList l;
	for (int i = 0; i < SRH_MAX_NEIGH; ++i) l.n[i] = (i < nneigh) ? neigh[i] : 0;
	return l;
}

// Per-pixel result of the curve-extent pass (dense path planning)
struct Extent { int32_t xmin, xmax; };

// ---- launch wrappers (srh_kernels.hip / srh_dense.hip); all asynchronous on `st` ----
void launch_prep_view(hipStream_t st, const uint32_t *rgba, const uint8_t *mask, int w, int h,
                      double *gray, double *gray_tv);
void launch_fill(hipStream_t st, double *p, size_t n, double v);
// doubles per pixel of `pconst`: meanL, totalWeight, sum2, [3] 0 / 1/totalWeight, [4] SA = sum of fma(w_t, l_t, -meanL) (the
// one-pass form's sum1 = Q - m*SA; FUSED terms: an unfused w*l - meanL carries an absolute error u*|w*l| per tap, which at a
// small sum2 is beyond the certificate's bound), [5] pad (rows of 48 bytes: every tile's piece starts 16-byte aligned)
#define SRH_PC 6
// pconst (optional): SRH_PC doubles per pixel of the band -- meanL, totalWeight, sum2, all-taps-usable (0, or 1/totalWeight), SA -- the per-pixel
// constants of the dense cost kernel's fast form, computed while the window is at hand
void launch_weights(hipStream_t st, const ViewDev *views, int ref, int width, const srh_params &P,
                    int y0, int nrows, double *wbuf, size_t wstride, double *pconst = nullptr, bool wimg = false);
void launch_twoview_generic(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                            int y0, int nrows, const double *wbuf, size_t wstride, Counters *cnt);
void launch_twoview_cross_check(hipStream_t st, const ViewDev *views, int self, int other, int w, int h,
                                const srh_params &P);
void launch_mvs_generic(hipStream_t st, const ViewDev *views, int ref, const int32_t *neigh, int nneigh, int width,
                        const srh_params &P, int y0, int nrows, const double *wbuf, size_t wstride,
                        double *peaks, double *best, Counters *cnt);
void launch_mvs_cross_check(hipStream_t st, const ViewDev *views, const int32_t *slots_dev, int nviews,
                            int view_index, int w, int h, const srh_params &P);

// Dense (row-aligned) TwoView path, srh_dense.hip
void launch_edge_planes(hipStream_t st, const uint32_t *rgba, int w, int h, double *edges);
// the dense TwoView path's windows kernel with its tiles by LDS-DMA (r = 5; srh_dense.hip): the padded planes it reads
void launch_geo_exp_probe(hipStream_t st, const double *x, int n, double *kout, double *lout);   // srh_debug_exp
size_t geo5_doubles(int w, int h);
void launch_geo5_planes(hipStream_t st, const double *edges, const double *gray_tv, const uint8_t *mask, int w, int h, double *out);
bool launch_geodesic_dma(hipStream_t st, const ViewDev *views, int ref, int width, const double *geo5, const srh_params &P,
                         int y0, int nrows, double *wbuf, double *pconst, int num_cus);
bool launch_geodesic_reg(hipStream_t st, const ViewDev *views, int ref, int width, const double *edges,
                         const srh_params &P, int y0, int nrows, double *wbuf, size_t wstride, double *pconst = nullptr,
                         bool wimg = false);
bool launch_adaptive_reg(hipStream_t st, const ViewDev *views, int ref, int width, const srh_params &P, int y0, int nrows,
                         double *wbuf, size_t wstride, double *pconst = nullptr, bool wimg = false);
void launch_label_plane_table(hipStream_t st, const ViewDev *views, int ref, const srh_params &P, bool mvs, double *tdist);
void launch_pinhole_label_table(hipStream_t st, const ViewDev *views, int ref, const srh_params &P, bool mvs, double *tnum);
bool launch_twoview_dense_cost_f32(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                                   int y0, int nrows, const double *wbuf, size_t wstride,
                                   const double *tnum, double *cost, int cstride, Counters *cnt, const double *pconst, int form = 0);   // form 1: one-pass sums
bool launch_twoview_dense_cost(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                               int y0, int nrows, const double *wbuf, size_t wstride,
                               const double *tnum, double *cost, int cstride, Counters *cnt, const double *pconst, int arith = 0,
                               const PixRange *prange = nullptr);   // the pixels' column ranges from pixel_range_kernel (else worked out per tile)
// cflag == nullptr: the exact scan (cnt == nullptr: without counting).  cflag, nlist < 0: the certified scan on fused
// costs, flagged pixels into cflag = [count | band pixel indices].  cflag, nlist >= 0: the exact scan of the listed pixels
// (launch sized for nlist pixels, the count itself is read on the device)
void launch_twoview_scan(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                         int y0, int nrows, const double *tnum, const double *cost, int cstride,
                         Counters *cnt, const PixRange *prange, uint32_t *cflag = nullptr, int nlist = -1,
                         const double *pexact = nullptr,    // certified scan: the band's pconst when the cost kernel applies cert_pixel_exact
                         const void *tpl = nullptr, uint32_t *tilelist = nullptr,   // template scan (launch_scan_template) + the tiles it leaves: [count | tile indices]
                         int num_cus = 256);
// the pass's candidate template (twoview_template_kernel) into `tpl` (scan_template_bytes() bytes)
size_t scan_template_bytes();
void launch_scan_template(hipStream_t st, const ViewDev *views, int ref, int oth, const srh_params &P, int y0, int nrows,
                          const double *tnum, void *tpl);
// the cost rows of the flagged pixels in the reference's arithmetic; `cap` workgroups (the count is read on the device;
// a count above cap is reported in Counters::cert_overflow); wimg: LDS-image windows, else tile-major
bool launch_twoview_refill(hipStream_t st, int width, const srh_params &P, int y0, const PixRange *prange, const uint32_t *cflag, int cap,
                           const double *wbuf, bool wimg, const double *ref_tvp, const double *oth_tvp, double *cost, int cstride, Counters *cnt);
// columns the cost kernel leaves out (dense_cover_hi with `lanes` block lanes per pixel), filled in before the scan
void launch_twoview_lazy_fill(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                              int y0, int nrows, const PixRange *prange, const double *wbuf, size_t wstride,
                              const double *ref_tvp, const double *oth_tvp,      // NaN-bordered planes (radius 5 / 2), else null
                              bool wimg,                                         // LDS-image windows (strip path), else tile-major
                              int lanes, bool padded,                            // the cost kernel's form: block lanes per pixel, blocks on even columns
                              double *cost, int cstride, Counters *cnt);

// Persistent strip form of the dense cost kernel, srh_strip.hip
void launch_padded_plane(hipStream_t st, const double *gray_tv, int w, int h, double *out);
void launch_padded_full(hipStream_t st, const double *gray_tv, int w, int h, int R, uint8_t *out);
void launch_pixel_range(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                        int y0, int nrows, const double *tnum, int cstride, PixRange *prange);
int  strip_chunk_columns();
int  strip_block_lanes(int cstride, int form);
bool launch_twoview_strip_cost(hipStream_t st, const ViewDev *views, int ref, int oth, int width, int height,
                               const srh_params &P, int y0, int nrows, const double *wimg, const double *pconst,
                               const PixRange *prange, const double *ref_tvp, const double *oth_tvp,
                               const uint8_t *oth_fullp, double *cost, int cstride, Counters *cnt, int arith, int num_cus,
                               int lanes, bool raw = false);   // raw (diagnostics): certified forms without the in-kernel exact redo

#ifdef SRH_PROFILE_PHASES
void geodesic_phases_fetch(unsigned long long out[8]);
#endif
#ifdef SRH_EXPERIMENT
void exp_set(int repeat, int lds_pad);
void exp_set_scan(int mode);
void exp_set_walk(int mode);
void exp_set_rows(int mode);
#endif

// Fused row-aligned TwoView kernel (geometry + cost + WTA per 16-pixel tile), srh_fused.hip.
// SRH_FUSED_MAXC: cost-row columns (and labels) a pixel may have in LDS.
#define SRH_FUSED_MAXC 256
bool launch_twoview_fused(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                          int y0, int nrows, const double *wbuf, const double *tnum, Counters *cnt);

// Candidate-list TwoView path for arbitrary geometry, srh_list.hip
void launch_twoview_count(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                          int y0, int nrows, int32_t *count, Counters *cnt, int *max_count, const double *tdist);
void launch_twoview_list(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                         int y0, int nrows, uint32_t *cand, int cmax, int32_t *count, Counters *cnt, int *max_count,
                         const double *tdist);
void launch_full_window(hipStream_t st, const double *gray_tv, int w, int h, int R, uint8_t *full, uint32_t *stat = nullptr);   // stat[2] (zeroed): usable centres, full windows
inline size_t full_stat_offset(size_t npix) { return (npix + 15) & ~(size_t)15; }   // the two counters live behind the map, 16-byte aligned
bool launch_twoview_list_cost(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                              int y0, int nrows, const double *wbuf, const uint8_t *full_oth,
                              const int32_t *count, const uint32_t *cand, double *cost, int cmax, Counters *cnt);
void launch_twoview_list_scan(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                              int y0, int nrows, const int32_t *count, const uint32_t *cand, const double *cost, int cmax);
// act / nact: the band's masked-in pixels (y*w + x) in the view's serpentine order -- the units of the launch, 128 per
// block and link, so that every wave is full and its 64 pixels lie side by side (also across a row change).
// wdesc / nwin (or null): per wave of the walk launch (2 per 128-pixel block and link), the windows of list slots whose
// box of the other view fits the LDS (mvs_staged_cost_kernel); mvs_staging_shape gives the buffer sizes
void mvs_staging_shape(int *maxw, size_t *desc_words_per_wave);
void launch_mvs_walk(hipStream_t st, const ViewDev *views, int ref, const int32_t *neigh, int nneigh, int width,
                     const srh_params &P, int y0, int nrows, const double *tnum, uint32_t *cand, int cmax, int32_t *count,
                     Counters *cnt, int *max_count, uint32_t *wdesc, int32_t *nwin, const uint32_t *act, int nact, bool peaks,
                     bool fast_pinhole = false);   // every neighbour a plain pinhole camera: certified label projections (with tnum)
// cert: the certified fused form (no top-K request): fused sweeps, the unit's winner certified against the bound, its
// cost recomputed in the reference's arithmetic; ambiguous units redone exactly
void launch_mvs_staged_cost(hipStream_t st, const ViewDev *views, int ref, const int32_t *neigh, int nneigh, int width,
                            const srh_params &P, int y0, int nrows, const double *wbuf, size_t wstride,
                            const uint32_t *cand, int cmax, const int32_t *count, double *best,
                            const uint32_t *wdesc, const int32_t *nwin, Counters *cnt, const uint32_t *act, int nact,
                            double *unit_peaks, bool cert = false);
void launch_mvs_list_cost(hipStream_t st, const ViewDev *views, int ref, const int32_t *neigh, int nneigh, int width,
                          const srh_params &P, int y0, int nrows, const double *wbuf, size_t wstride,
                          const uint32_t *cand, int cmax, const int32_t *count, double *best,
                          double *unit_peaks, bool peaks, const int32_t *nwin, const uint32_t *act, int nact);
void launch_mvs_combine(hipStream_t st, const ViewDev *views, int ref, int nneigh, int width, const srh_params &P,
                        int y0, int nrows, const double *best, const double *unit_peaks, double *peaks);
void launch_point_cloud(hipStream_t st, const ViewDev *views, int slot, int w, int h, const srh_params &P,
                        double *xyz, uint8_t *rgb, uint8_t *valid, unsigned long long *counts);
// MRF stage (srh_mrf.hip): one scratch buffer, carved the same way by every launch
struct MrfLayout {
	double *pz, *D, *Mh, *Mv, *partial, *energy;
	unsigned long long *hand; size_t hand_words;
	unsigned *status, *sync;
	int32_t *ans;
	int nparts, nbands;
	size_t sync_words, total_doubles;
};
size_t mrf_scratch_doubles(int w, int h);
void launch_mrf_layout(double *buf, int w, int h, MrfLayout &lay);
hipError_t launch_mrf_setup(hipStream_t st, double *buf, int w, int h, int K, double beta, double lambda, double phiu,
                            const double *peaks, MrfLayout &lay);
hipError_t launch_mrf_sweep(hipStream_t st, double *buf, int w, int h, int K, double psiu, int sweep);
hipError_t launch_mrf_energy(hipStream_t st, double *buf, int w, int h, int K, double psiu);
hipError_t launch_mrf_depth(hipStream_t st, const ViewDev *views, int slot, double *buf, int w, int h, int K);
void launch_epipolar_preview(hipStream_t st, const ViewDev *views, int ref, int oth, double zmin, double zmax, int nd,
                             int nq, const double *xy, double *out, int32_t *counts);
void launch_refraction_error(hipStream_t st, const ViewDev *views, int v1, int v2, int n, const double *p1, const double *p2, double *err);
void launch_epipolar_curves(hipStream_t st, const ViewDev *views, int ref, int oth, const srh_params &P, int mvs,
                            int nq, const int32_t *xy, int32_t *out, int cap, int32_t *counts);

// run-blocked candidate lists, srh_rows.hip
#define SRH_ROWS_NR 32
void launch_twoview_rows_list(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                              int y0, int nrows, uint32_t *cand, int cmax, int32_t *count, uint32_t *rowinfo,
                              int32_t *meta, int smax, Counters *cnt, int *maxes, const double *tdist);
// arith: 0 = the reference's arithmetic, 3 = certified fused sweeps (values stored for the certified scan)
bool launch_twoview_rows_cost(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                              int y0, int nrows, const double *wbuf, const uint8_t *full_oth,
                              const uint32_t *rowinfo, const int32_t *meta, double *cost, int smax, Counters *cnt, int arith = 0,
                              const double *pconst = nullptr,    // pconst: the band's per-pixel constants from the weights kernel (else made in the kernel)
                              const double *oth_tvp = nullptr, int num_cus = 256, const uint32_t *full_stat = nullptr);  // the other view's NaN-bordered plane: masked fast blocks + single candidates (else: a block is fast when all 8 are)
// cflag == nullptr: exact scan.  cflag, nlist < 0: certified scan (flags into cflag = [count | band pixel indices]).
// cflag, nlist >= 0: the exact scan of the listed pixels (after launch_twoview_rows_refill)
void launch_twoview_rows_scan(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                              int y0, int nrows, const int32_t *count, const uint32_t *cand, int cmax,
                              const uint32_t *rowinfo, const int32_t *meta, const double *cost, int smax,
                              uint32_t *cflag = nullptr, int nlist = -1, Counters *cnt = nullptr);
bool launch_twoview_rows_refill(hipStream_t st, int width, int oth_width, const srh_params &P, int y0,
                                const uint32_t *cflag, int cap, const double *wbuf, const double *ref_tvp, const double *oth_tvp,
                                const uint32_t *rowinfo, const int32_t *meta, double *cost, int smax, Counters *cnt);
// RCCL exchange, srh_comm.hip (functions return nullptr or an error string)
const char *rccl_unique_id_get(void *out128);
const char *rccl_comm_init(void **comm, int nranks, int rank, const void *id128);
const char *rccl_comm_destroy(void *comm);   // finalize (non-blocking communicator) + destroy, bounded; error text or null
bool rccl_available();                        // librccl can be loaded and has the entry points (else: SRH_E_UNSUPPORTED)
// bounded wait for everything `st` holds (a collective included): event + ncclCommGetAsyncError polled to the timeout
const char *rccl_wait_stream(void *comm, hipStream_t st, const char *what);
void rccl_comm_abort(void *comm);
void rccl_set_timeout_ms(int ms);      // how long a call into RCCL may stay "in progress" (rendezvous, collectives); default 120 s
int  rccl_version();                   // NCCL_VERSION_CODE of the loaded librccl, 0 if there is none
const char *rccl_gather_f64(void *comm, int nranks, int rank, int root, const double *send, double *recv,
                            size_t count, hipStream_t st);
const char *rccl_allgather_f64(void *comm, const double *send, double *recv, size_t count, hipStream_t st);

} // namespace srh
