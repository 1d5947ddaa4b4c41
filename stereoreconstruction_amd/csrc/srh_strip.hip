// srh_strip.hip -- TwoViewStereo::cost_ncc (stereo/twoviewstereo.cpp:909-977) for row-aligned rigs as a
// PERSISTENT kernel: the same block loops, sums and operation order as twoview_dense_cost_kernel (srh_dense.hip),
// so the same bits -- what changes is everything around the loops.
//
// twoview_dense_cost_kernel starts one workgroup per 32-pixel tile; each of them works out its candidate ranges,
// stages 11 rows of both views and its windows through registers, derives which candidate windows are complete
// (three barriers) and only then reaches the FP64 loops.  Measured on C3 (profiles/r03_repeat_experiment.txt): the
// loops cost 15.1 ms per launch, everything else 4.45 ms, and the two do not overlap.  Here
//
//   * a workgroup owns a vertical strip -- one tile column, 4..16 consecutive rows (work items drawn from a
//     ticket counter, tall items first) -- and keeps the rows of both views in an LDS ring: going one row down
//     costs ONE new row per view instead of eleven;
//   * every byte enters LDS by LDS-DMA (global_load_lds): the windows (the weights kernels write them in the
//     LDS image's own layout, srh_internal.hpp "layout B"), the per-pixel constants, the per-pixel candidate
//     ranges (pixel_range_kernel) and the "window fully usable" bytes of the other view (padded_full_kernel);
//     the image rows come from NaN-bordered copies of gray_tv, so no lane ever tests a bound;
//   * the next tile's row, constants and (512-thread form) windows are requested while the current tile
//     computes; one barrier per tile.
//
// Two forms: 4 waves (32 px x 8 block lanes, one window buffer, 2 workgroups per CU) and 8 waves (32 px x 16
// block lanes, two window buffers, 1 workgroup per CU).  Both keep 2 waves per SIMD.
#include "srh_internal.hpp"
#include "srh_geom.hpp"
#include "srh_walk.hpp"

#include <type_traits>

namespace srh {

#ifdef SRH_EXPERIMENT
// timing experiments only (make exp): repeat the block loops of every tile (srh_dense.hip, exp_set)
__device__ int g_strip_exp_repeat = 1;
void strip_exp_set(int repeat) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_strip_exp_repeat), &repeat, sizeof(int)); }
#endif

#define ST_TP 32                       // pixels per tile (= SRH_WTILE)
#define ST_NCB 8                       // candidate columns per block
#define ST_CHUNK 320                   // candidate columns a tile can hold in LDS

typedef __attribute__((address_space(3))) void st_lds_void;
typedef __attribute__((address_space(1))) const void st_gbl_void;
typedef __attribute__((address_space(3))) volatile int st_lds_vint;

// ------------------------------------------------------------------ padded planes
__global__ void padded_plane_kernel(const double *__restrict__ gray_tv, int W, int H, double *__restrict__ out)
{
	const int SP = padded_stride(W), HP = H + 2*SRH_PADY;
	const size_t n = (size_t)SP*HP;
	const double nan = __builtin_nan("");
	for (size_t k = (size_t)blockIdx.x*blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x*blockDim.x) {
		const int x = (int)(k % (size_t)SP) - SRH_PADL, y = (int)(k / (size_t)SP) - SRH_PADY;
		out[k] = (x >= 0 && y >= 0 && x < W && y < H) ? gray_tv[(size_t)y*W + x] : nan;
	}
}

void launch_padded_plane(hipStream_t st, const double *gray_tv, int w, int h, double *out) {
	size_t n = padded_size(w, h);
	size_t b = (n + 255)/256; if (b > 4096) b = 4096;
	hipLaunchKernelGGL(padded_plane_kernel, dim3((unsigned)b), dim3(256), 0, st, gray_tv, w, h, out);
}

// out[(y+PADY)*SP + x+PADL] = 1 where the whole (2R+1)^2 TwoView window centred on (x, y) is usable
// (inside the image, every gray_tv tap valid), 0 elsewhere -- full_window_kernel (srh_list.hip) on the padded raster
__global__ void padded_full_kernel(const double *__restrict__ gray_tv, int W, int H, int R, uint8_t *__restrict__ out)
{
	const int SP = padded_stride(W), HP = H + 2*SRH_PADY;
	const size_t n = (size_t)SP*HP;
	for (size_t k = (size_t)blockIdx.x*blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x*blockDim.x) {
		const int x = (int)(k % (size_t)SP) - SRH_PADL, y = (int)(k / (size_t)SP) - SRH_PADY;
		bool ok = x - R >= 0 && y - R >= 0 && x + R < W && y + R < H;
		for (int row = -R; ok && row <= R; ++row)
			for (int col = -R; col <= R; ++col) {
				const double v = gray_tv[(size_t)(y + row)*W + (x + col)];
				ok = ok && (v == v);
			}
		out[k] = ok ? 1 : 0;
	}
}

void launch_padded_full(hipStream_t st, const double *gray_tv, int w, int h, int R, uint8_t *out) {
	size_t n = padded_size(w, h);
	size_t b = (n + 255)/256; if (b > 4096) b = 4096;
	hipLaunchKernelGGL(padded_full_kernel, dim3((unsigned)b), dim3(256), 0, st, gray_tv, w, h, R, out);
}

// ------------------------------------------------------------------ candidate ranges
// The column range of every reference pixel of the band (pinhole_column_range, srh_walk.hpp), once, with every lane
// busy: twoview_dense_cost_kernel works it out on one lane in eight per tile, twoview_scan_kernel again per pixel.
__global__ void pixel_range_kernel(const ViewDev *__restrict__ views, int ref, int oth, srh_params P, int y0, int nrows,
                                   const double *__restrict__ tnum, int cstride, PixRange *__restrict__ prange)
{
	const ViewDev &L = views[ref];
	const int W = L.w;
	const size_t q = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
	if (q >= (size_t)nrows*W) return;
	const int x = (int)(q % W), y = y0 + (int)(q / W);
	int lo = 0, hi = -1;
	if (L.mask[(size_t)y*W + x] == 1) {
		const Ray ray = cam_unproject(L.cam, (x + 0.5) / P.image_scale, (y + 0.5) / P.image_scale);
		pinhole_column_range(ray, L.cam, views[oth], P, tnum, cstride, lo, hi);
	}
	PixRange r; r.lo = lo; r.hi = hi;
	prange[q] = r;
}

void launch_pixel_range(hipStream_t st, const ViewDev *views, int ref, int oth, int width, const srh_params &P,
                        int y0, int nrows, const double *tnum, int cstride, PixRange *prange)
{
	const size_t n = (size_t)nrows*width;
	hipLaunchKernelGGL(pixel_range_kernel, dim3((unsigned)((n + 255)/256)), dim3(256), 0, st,
	                   views, ref, oth, P, y0, nrows, tnum, cstride, prange);
}

int strip_chunk_columns() { return ST_CHUNK; }

// ------------------------------------------------------------------ the strip kernel
struct StripArgs {
	int W, H, y0, nrows;                    // reference view size; rows [y0, y0 + nrows) of it
	const double *wimg;                     // windows of the band, layout B
	const double *pconst;                   // SRH_PC doubles per pixel of the band: meanL, totalWeight, sum2, 0 | 1/totalWeight, SA, pad
	const PixRange *prange;                 // candidate column range per pixel of the band
	const double *ref_tvp, *oth_tvp;        // NaN-bordered gray_tv planes
	const uint8_t *oth_fullp;               // zero-bordered "window fully usable" plane of the other view
	double *cost; int cstride;              // cost rows, tile-transposed: ((tile*cstride) + k)*32 + pixel
	Counters *cnt;
	int n1, n2;                             // rows [0,n1) in 16-row items, [n1,n2) in 8-row items, the rest in 4-row items
	int nitems;
	double weight_cutoff, bad_ret, max_color_diff;
	CertBound cb;                           // certified arithmetic (AR == 3): the constants of the error bound, srh_internal.hpp
	int redo;                               // certified forms: 1 = uncovered candidates re-evaluated in place in the reference's arithmetic (default); 0 = raw
};

template <int R, int NBUF>
struct StripSmem {
	static constexpr int WS = 2*R + 1;
	static constexpr int T = WS*WS;
	static constexpr int WP = (WS + 1) & ~1;                 // taps per window row, padded even
	static constexpr int WPIX = WS*WP;                       // doubles per pixel window
	static constexpr int RW = ST_CHUNK + 2*R + ST_NCB + 2;   // staged width of the other view's rows (even)
	static constexpr int LW = (ST_TP + 2*R + 1) & ~1;        // staged width of the reference rows (even)
	static constexpr int NS = WS + 1;                        // row slots of the rings: the window's rows + the next one
	static constexpr int FW = (RW + 4 + 15) & ~15;           // bytes of a staged "full" row piece (dword granules + slack)
	// phase-2 work lists in one array: 8-column blocks for the blocked select form from the front
	// (at most ST_TP*(CHUNK/NCB + 1) of them), single candidates for the per-candidate form from the back
	static constexpr int NBMAX = ST_CHUNK/ST_NCB + 1;
	static constexpr int GL_CAP = 3072;
	static constexpr int GL_SINGLES = GL_CAP - ST_TP*NBMAX;
	alignas(16) double w[NBUF][WS][ST_TP][WP];         // LDS image of the windows: [row][pixel][tap], as in the band buffer
	alignas(16) double rt[NS][RW];
	// 8-wave form: the same rows once more, shifted by one column (rto[s][k] = rt[s][k + 1]): a block that starts on
	// an odd column still reads 16 aligned bytes at a time, so blocks start at the pixel's first column (no alignment
	// pad) and a range of 8*16*k columns needs exactly k rounds -- on C3 no column is left to the fill kernel
	alignas(16) double rto[NBUF == 2 ? NS : 1][RW];
	alignas(16) double lt[NS][LW];
	alignas(16) double pc[2][ST_TP][SRH_PC];         // meanL, totalWeight, sum2, 0 | 1/totalWeight, SA, pad (srh_internal.hpp)
	alignas(16) PixRange pr[2][ST_TP];
	alignas(16) unsigned char full[2][FW];
	unsigned short glist[GL_CAP];
	int glist_n, gsingle_n;
	int item;
	unsigned long long badcols[ST_CHUNK/64 + 2];   // bit k: the window of candidate column cs + k is not fully usable
	int simd[8];                  // SIMD of each wave of the workgroup (8-wave form: which two waves share one)
	int prog[NBUF == 2 ? 8 : 1][64];   // progress of each wave inside the current tile, in window rows (pass 2 rows count 3); one copy per lane: no lane masking in the hot loops
	static_assert(ST_CHUNK <= 512 && ST_TP <= 64, "work-list entry = pixel*512 + column in 16 bits");
	static_assert(RW % 2 == 0 && LW % 2 == 0 && WP % 2 == 0, "16-byte rows");
	static_assert(RW - R + 4 <= SRH_PADR + 1, "padded planes cover every staged piece");
};

// lanes [0, nbytes/16) of the wave copy 16 bytes each from src + 16*lane to LDS dst + 16*lane, 1 KiB per instruction
__device__ __forceinline__ void st_dma16(const void *src, void *dst, int nbytes, int lane) {
	for (int off = 0; off < nbytes; off += 1024) {
		if (off + lane*16 < nbytes)
			__builtin_amdgcn_global_load_lds((st_gbl_void *)((const char *)src + off + lane*16),
			                                 (st_lds_void *)((char *)dst + off), 16, 0, 0);
	}
}
// the same in 4-byte granules (sources that are only 4-byte aligned)
__device__ __forceinline__ void st_dma4(const void *src, void *dst, int nbytes, int lane) {
	for (int off = 0; off < nbytes; off += 256) {
		if (off + lane*4 < nbytes)
			__builtin_amdgcn_global_load_lds((st_gbl_void *)((const char *)src + off + lane*4),
			                                 (st_lds_void *)((char *)dst + off), 4, 0, 0);
	}
}

// general (any validity pattern) cost of one candidate from the LDS tiles: dense_cost_general of srh_dense.hip
// with the rows taken from the rings.  Skipped taps add +0.0 to every sum (twoviewstereo.cpp:920-938, 955-972).
template <int R, int NBUF>
__device__ __noinline__ double strip_cost_general(const StripSmem<R, NBUF> &S, int wb, int s0, int i, int rc,
                                                  double weight_cutoff, double bad_ret, double max_color_diff)
{
	typedef StripSmem<R, NBUF> Smem;
	constexpr int WS = Smem::WS, NS = Smem::NS;
	double meanL = 0, meanR = 0, totalWeight = 0.0;
#pragma unroll 1
	for (int row = 0; row < WS; ++row) {
		const int sl = s0 + row >= NS ? s0 + row - NS : s0 + row;
		double gl[WS], gr[WS], wt[WS];
#pragma unroll
		for (int col = 0; col < WS; ++col) { gl[col] = S.lt[sl][i + col]; gr[col] = S.rt[sl][rc + col]; wt[col] = S.w[wb][row][i][col]; }
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
		const int sl = s0 + row >= NS ? s0 + row - NS : s0 + row;
		double gl[WS], gr[WS], wt[WS];
#pragma unroll
		for (int col = 0; col < WS; ++col) { gl[col] = S.lt[sl][i + col]; gr[col] = S.rt[sl][rc + col]; wt[col] = S.w[wb][row][i][col]; }
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

// One 8-column block of pixel `pi` in the BLOCKED select form: any validity pattern (image border, masked taps,
// cut-off weights), the same sums as strip_cost_general with every tap guarded -- a skipped tap adds +0.0 -- but
// the other view's row segment is read once per window row and shared by the 8 candidates, as in the fast form
// (the per-candidate form reads 3 LDS values per tap and candidate).  Candidates whose bit is set in `store` are written.
template <int R, int NBUF>
__device__ __noinline__ void strip_select_block(const StripSmem<R, NBUF> &S, int wb, int s0, int pi, int rc, unsigned store,
                                                double *__restrict__ dst, double weight_cutoff, double bad_ret, double max_color_diff,
                                                int ra, int rb)
{
	// [ra, rb): the window rows that lie on rows of the reference image with valid tap values at all (rows 0 .. H - 2: the last
	// row's gray_tv is NaN like everything outside).  On the other window rows every tap of the reference side is unusable and
	// every term below a zero: they are left out -- a third of this routine's work on the image's first and last rows, whose
	// every candidate comes here.
	typedef StripSmem<R, NBUF> Smem;
	constexpr int WS = Smem::WS, NS = Smem::NS, NCB = ST_NCB, NR = NCB + 2*R;
	// "A skipped tap adds +0.0 to every sum" WITHOUT a select per tap and candidate (six v_cndmask_b32 and a compare for each of
	// the 968 pairs of a block, twice): the validity of a value is settled once per value -- an unusable gray value becomes 0.0
	// with a flag 0.0 / 1.0 beside it, a tap whose own side is unusable gets weight 0.0 -- and what the reference skips is
	// multiplied by that 0.0: x*1.0 is x and x*0.0 a zero for the finite x here, and a sum that starts at +0.0 never is -0.0,
	// so adding a zero of either sign changes nothing.  The taps that count go through the reference's own operations.
	double mLs[NCB], mRs[NCB], tws[NCB];
#pragma unroll
	for (int j = 0; j < NCB; ++j) { mLs[j] = 0.0; mRs[j] = 0.0; tws[j] = 0.0; }
#pragma unroll 1
	for (int row = ra; row < rb; ++row) {
		const int sl = s0 + row >= NS ? s0 + row - NS : s0 + row;
		double rr[NR], rv[NR];
		const double2 *rp = reinterpret_cast<const double2 *>((NBUF == 2 && (rc & 1)) ? &S.rto[sl][rc - 1] : &S.rt[sl][rc]);
#pragma unroll
		for (int m = 0; m < NR/2; ++m) { const double2 v = rp[m]; rr[2*m] = v.x; rr[2*m + 1] = v.y; }
#pragma unroll
		for (int m = 0; m < NR; ++m) { const bool okr = rr[m] == rr[m]; rv[m] = okr ? 1.0 : 0.0; rr[m] = okr ? rr[m] : 0.0; }
#pragma unroll
		for (int col = 0; col < WS; ++col) {
			const double gl = S.lt[sl][pi + col], wt = S.w[wb][row][pi][col];
			const bool okl = gl == gl && wt > weight_cutoff;
			const double w0 = okl ? wt : 0.0, pl0 = okl ? wt*gl : 0.0;
#pragma unroll
			for (int j = 0; j < NCB; ++j) {
				mLs[j] += pl0*rv[col + j];
				mRs[j] += w0*rr[col + j];
				tws[j] += w0*rv[col + j];
			}
		}
	}
#pragma unroll
	for (int j = 0; j < NCB; ++j) { mLs[j] /= tws[j]; mRs[j] /= tws[j]; }   // unused when tws < 1e-10
	double s1[NCB], s2v[NCB], s3[NCB];
#pragma unroll
	for (int j = 0; j < NCB; ++j) { s1[j] = 0.0; s2v[j] = 0.0; s3[j] = 0.0; }
#pragma unroll 1
	for (int row = ra; row < rb; ++row) {
		const int sl = s0 + row >= NS ? s0 + row - NS : s0 + row;
		double rr[NR], rv[NR];
		const double2 *rp = reinterpret_cast<const double2 *>((NBUF == 2 && (rc & 1)) ? &S.rto[sl][rc - 1] : &S.rt[sl][rc]);
#pragma unroll
		for (int m = 0; m < NR/2; ++m) { const double2 v = rp[m]; rr[2*m] = v.x; rr[2*m + 1] = v.y; }
#pragma unroll
		for (int m = 0; m < NR; ++m) { const bool okr = rr[m] == rr[m]; rv[m] = okr ? 1.0 : 0.0; rr[m] = okr ? rr[m] : 0.0; }
#pragma unroll
		for (int col = 0; col < WS; ++col) {
			const double gl = S.lt[sl][pi + col], wt = S.w[wb][row][pi][col];
			const bool okl = gl == gl && wt > weight_cutoff;
			const double pl = wt*(gl == gl ? gl : 0.0), kl = okl ? 1.0 : 0.0;
#pragma unroll
			for (int j = 0; j < NCB; ++j) {
				const double k = kl*rv[col + j];                       // 1.0: the tap counts for this candidate
				const double a = (pl - mLs[j])*k, bq = (wt*rr[col + j] - mRs[j])*k;
				s1[j] += a*bq;
				s2v[j] += a*a;
				s3[j] += bq*bq;
			}
		}
	}
#pragma unroll
	for (int j = 0; j < NCB; ++j) {
		if ((store >> j) & 1u) {
			double result = bad_ret;
			if (!(tws[j] < 1e-10)) {
				const double v = 255*(1.0 - fabs(s1[j]) / sqrt(s2v[j] * s3[j]));
				result = (v < max_color_diff) ? v : max_color_diff;
			}
			dst[(ptrdiff_t)j*ST_TP] = result;
		}
	}
}

// NWV waves: 4 (block lanes per pixel G = 8, NBUF = 1) or 8 (G = 16, NBUF = 2)
// AR: 0 = the reference's arithmetic (no contraction), 1 = fused multiply-adds, 5 = certified ONE-PASS form (below),
// 3 = fused two sweeps AND certified: a candidate
// whose error bound (srh_internal.hpp, CertBound) is not below cb.e0 is stored as NaN, a value above max_color_diff + e0
// as max_color_diff itself (the exact cost is then max_color_diff too), anything else unclamped -- the certified scan
// does the rest.  Candidates of the select forms are evaluated in the reference's arithmetic in every mode.
template <int R, int NWV, int NBUF, int AR>
__global__ __launch_bounds__(NWV*64, 2)
void twoview_strip_cost_kernel(const StripArgs A)
{
	constexpr bool FMA = AR != 0, CERT = AR == 3 || AR == 5, ONEPASS = AR == 5;
	typedef StripSmem<R, NBUF> Smem;
	constexpr int WS = Smem::WS, WP = Smem::WP, WPIX = Smem::WPIX, RW = Smem::RW, LW = Smem::LW, NS = Smem::NS;
	constexpr int NCB = ST_NCB, CHUNK = ST_CHUNK;
	constexpr int NR = NCB + 2*R;              // right-row values a block needs (even)
	constexpr int G = 2*NWV;                   // block lanes per pixel
	constexpr int NT = NWV*64;
	constexpr bool PAD = NBUF == 1;            // blocks start on even columns (one copy of the rows); 8-wave form: at the pixel's first column
	static_assert(NR % 2 == 0, "16-byte rows");
	static_assert((NWV == 4 && NBUF == 1) || (NWV == 8 && NBUF == 2), "two forms");
	extern __shared__ __align__(16) unsigned char smem_raw[];
	Smem &S = *reinterpret_cast<Smem *>(smem_raw);
	const Smem &CS = S;

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int pg = wv & 3;                       // pixel group of the wave: pixels 8*pg .. 8*pg + 7 of the tile
	const int i = pg*8 + (lane & 7);             // pixel within the tile (pixel-fastest inside a wave)
	const int g = (lane >> 3) + 8*(wv >> 2);     // block lane of the pixel
	const int W = A.W;
	const int tiles_per_row = (W + ST_TP - 1)/ST_TP;
	const int SP = padded_stride(W);
	unsigned n_dev = 0;
	// 8-wave form: the two waves of a SIMD share its FP64 pipe, and the arbiter serves the older one first -- left alone,
	// the older wave finishes every tile early and waits at the tile barrier while the younger runs on by itself at the
	// lower single-wave issue rate (measured: a third of a wave's time at the barrier).  Each wave publishes how far it
	// is in the tile and yields (s_setprio 0) while it is ahead of its SIMD partner, so both reach the barrier together.
	int partner = wv;
	if (NWV == 8) {
		// HW_REG_HW_ID (id 4), SIMD_ID = bits [5:4]
		const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
		if (lane == 0) S.simd[wv] = simd;
		S.prog[wv][lane] = 0;
		__syncthreads();
		for (int j = 0; j < NWV; ++j) if (j != wv && CS.simd[j] == simd) { partner = j; break; }
		partner = __builtin_amdgcn_readfirstlane(partner);
	}
	int prog = 0;
	// publish own progress, look at the partner's (value used by prog_yield at the end of the window row)
	auto prog_step = [&](int units, int &seen) {
		if (NWV == 8) {
			prog += units;
			*(st_lds_vint *)&S.prog[wv][lane] = prog;
			seen = *(st_lds_vint *)&S.prog[partner][lane];
		}
	};
	// s_setprio 0 while ahead of the partner, 1 otherwise.  One asm statement with its own branch: the row loops stay
	// single basic blocks for the compiler's scheduler and register allocator.
	auto prog_yield = [&](int seen) {
		if (NWV == 8) {
			const int ps = __builtin_amdgcn_readfirstlane(seen), pm = __builtin_amdgcn_readfirstlane(prog);
			asm volatile("s_nop 0\n\ts_cmp_gt_i32 %0, %1\n\ts_cbranch_scc1 1f\n\ts_setprio 1\n\ts_branch 2f\n1:\n\ts_setprio 0\n2:"
			             :: "s"(pm), "s"(ps) : "scc");
		}
	};
	// per-wave phase clocks exist only in the -DSRH_PROFILE_PHASES diagnostic build: a handful of s_memtime per tile,
	// none inside the block loops
#ifdef SRH_PROFILE_PHASES
	unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	unsigned long long tp = __builtin_readcyclecounter();
	const unsigned long long t_begin = tp;
	unsigned long long n_tiles = 0, n_gen = 0;
#define ST_STAMP(k) do { const unsigned long long tn_ = __builtin_readcyclecounter(); ph[k] += tn_ - tp; tp = tn_; } while (0)
#else
#define ST_STAMP(k) do { } while (0)
#endif

	for (;;) {
		// ---- next work item: (row segment, tile column); tall segments first, so the tail of the launch is short
		if (tid == 0) S.item = (int)atomicAdd(&A.cnt->strip_ticket, 1u);
		__syncthreads();                           // also: every wave has left the previous item's last tile
		const int item = __builtin_amdgcn_readfirstlane(S.item);
		ST_STAMP(0);                                  // ticket + waiting for the workgroup's other waves
		if (item >= A.nitems) break;
		const int seg = item / tiles_per_row, tx = item - seg*tiles_per_row;
		int ya, nh;
		{
			const int s16 = A.n1 >> 4, s8 = (A.n2 - A.n1) >> 3;
			if (seg < s16) { ya = seg*16; nh = 16; }
			else if (seg < s16 + s8) { ya = A.n1 + (seg - s16)*8; nh = 8; }
			else { ya = A.n2 + (seg - s16 - s8)*4; nh = 4; }
			if (ya + nh > A.nrows) nh = A.nrows - ya;
		}
		const int x0 = tx*ST_TP;
		const int x = x0 + i;
		int cs = 0, rowbase = 0;                   // staging origin (column, even) and the image row held by slot 0

		// one tile's own inputs (nothing here depends on the staging origin): constants, ranges, [windows]
		auto issue_tile_inputs = [&](int r, int buf, int wb, bool consts, bool windows) {
			const size_t px0 = (size_t)r*W + x0;
			if (consts && wv == 0) st_dma16(A.pconst + px0*SRH_PC, &S.pc[buf][0][0], ST_TP*SRH_PC*8, lane);
			if (consts && wv == 1) st_dma4(A.prange + px0, &S.pr[buf][0], ST_TP*8, lane);
			if (windows) {
				const double *wt = A.wimg + ((size_t)r*tiles_per_row + tx)*(size_t)(ST_TP*WPIX);
				if (NBUF == 1) {
					// each wave its own pixels' windows (8 pixels x WP taps of every window row): nobody else reads
					// them in the block loops
					for (int a = 0; a < WS; ++a)
						st_dma16(wt + (size_t)a*(ST_TP*WP) + pg*8*WP, &S.w[wb][a][pg*8][0], 8*WP*8, lane);
				} else {
					constexpr int NBYTES = ST_TP*WPIX*8;
					for (int off = wv*1024; off < NBYTES; off += NWV*1024)
						if (off + lane*16 < NBYTES)
							__builtin_amdgcn_global_load_lds((st_gbl_void *)((const char *)wt + off + lane*16),
							                                 (st_lds_void *)((char *)&S.w[wb][0][0][0] + off), 16, 0, 0);
				}
			}
		};
		// image row `yy` of both views into ring slot `sl`, "full" bytes of row `fy` into full[fb]
		auto issue_row = [&](int yy, int sl) {
			const double *src = A.oth_tvp + (size_t)(yy + SRH_PADY)*SP + (cs - R + SRH_PADL);
			if (wv < 3) { if (wv*1024 + lane*16 < RW*8) __builtin_amdgcn_global_load_lds((st_gbl_void *)((const char *)src + wv*1024 + lane*16),
			                                                                              (st_lds_void *)((char *)&S.rt[sl][0] + wv*1024), 16, 0, 0); }
			else if (wv == 3) st_dma16(A.ref_tvp + (size_t)(yy + SRH_PADY)*SP + (x0 - R + SRH_PADL), &S.lt[sl][0], LW*8, lane);
			else if (NBUF == 2 && wv < 7) {
				const int pc = wv - 4;                            // the copy shifted by one column
				if (pc*1024 + lane*16 < RW*8) __builtin_amdgcn_global_load_lds((st_gbl_void *)((const char *)(src + 1) + pc*1024 + lane*16),
				                                                                (st_lds_void *)((char *)&S.rto[sl][0] + pc*1024), 16, 0, 0);
			}
		};
		static_assert(RW*8 <= 3*1024, "three pieces per row of the other view");
		auto issue_full = [&](int fy, int fb) {
			// bytes [cs, cs + RW) of row fy, from the 4-byte granule that holds the first one
			const size_t a0 = (size_t)(fy + SRH_PADY)*SP + (size_t)(cs + SRH_PADL);
			if (wv == NWV - 1) st_dma4(A.oth_fullp + (a0 & ~(size_t)3), &S.full[fb][0], (RW + 4 + 3) & ~3, lane);
		};

		bool first = true;
		for (int r = ya; r < ya + nh; ++r) {
			const int y = A.y0 + r;
			const int cur = (r - ya) & 1, nxt = cur ^ 1;
			const int wcur = NBUF == 2 ? cur : 0, wnxt = NBUF == 2 ? nxt : 0;
			const bool has_next = r + 1 < ya + nh;
			if (first) issue_tile_inputs(r, cur, wcur, true, true);
			// ---- A: everything requested for this tile has landed and every wave has left the previous tile
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			ST_STAMP(1);                               // own requests (and cost stores) outstanding
			__builtin_amdgcn_s_barrier();
			ST_STAMP(2);                               // waiting for the other waves

			// ---- B: the tile's candidate ranges (every wave works out the union for itself: 32 values)
			int cmin, cmax, cmin_raw;
			{
				const int pi = lane & 31;
				const PixRange q = CS.pr[cur][pi];
				int lo = q.lo, hi = q.hi;
				if (x0 + pi >= W) { lo = 0; hi = -1; }
				if (hi >= lo) hi = dense_cover_hi(lo, hi, NCB, G, PAD);
				int mn = hi >= lo ? lo : 2147483647, mx = hi >= lo ? hi : -2147483647;
				// (within each row of 16 lanes by DPP rotations -- four steps, no LDS; the two rows by lane reads.  As five
				// butterfly steps of ds_bpermute each reduction was five dependent LDS round trips)
#define ST_ROR(v, n) __builtin_amdgcn_update_dpp(0, (v), 0x120 + (n), 0xf, 0xf, false)
#pragma unroll
				for (int d = 8; d >= 1; d >>= 1) {
					const int on = d == 8 ? ST_ROR(mn, 8) : d == 4 ? ST_ROR(mn, 4) : d == 2 ? ST_ROR(mn, 2) : ST_ROR(mn, 1);
					const int ox = d == 8 ? ST_ROR(mx, 8) : d == 4 ? ST_ROR(mx, 4) : d == 2 ? ST_ROR(mx, 2) : ST_ROR(mx, 1);
					mn = on < mn ? on : mn; mx = ox > mx ? ox : mx;
				}
#undef ST_ROR
				{
					const int m0 = __builtin_amdgcn_readlane(mn, 0), m1 = __builtin_amdgcn_readlane(mn, 16);
					const int x0_ = __builtin_amdgcn_readlane(mx, 0), x1_ = __builtin_amdgcn_readlane(mx, 16);
					cmin_raw = m0 < m1 ? m0 : m1;
					cmax = x0_ > x1_ ? x0_ : x1_;
				}
				cmin = cmin_raw & ~1;
			}
			const bool any = cmin_raw <= cmax;
			// (re)stage the rings when the strip starts or the ranges have moved outside the staged columns
			if (first || (any && (cmin < cs || cmax > cs + CHUNK - 1))) {
				if (!first) __builtin_amdgcn_s_barrier();          // (a restage in mid-strip: nobody reads the rings any more)
				cs = any ? cmin : 0;
				rowbase = y - R;
				for (int rr = 0; rr < WS; ++rr) issue_row(y - R + rr, rr);
				issue_full(y, cur);
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				__builtin_amdgcn_s_barrier();
			}
			if (any && cmax > cs + CHUNK - 1) {
				// a candidate range wider than the chunk: not this kernel's case (the host falls back)
				if (tid == 0) atomicAdd(&A.cnt->strip_overflow, 1ull);
				cmax = cs + CHUNK - 1;
			}
			first = false;
			const int s0 = (y - R - rowbase) % NS;                   // ring slot of the window's first row
			// ---- request the next tile's row, constants, ranges [and windows] now: they travel under the block loops
			if (has_next) {
				issue_row(y + R + 1, (y + R + 1 - rowbase) % NS);
				issue_full(y + 1, nxt);
				issue_tile_inputs(r + 1, nxt, wnxt, true, NBUF == 2);
			}
			const int foff = (int)(((size_t)(y + SRH_PADY)*SP + (size_t)(cs + SRH_PADL)) & 3);   // first byte inside its granule
			const unsigned char *rfull = &CS.full[cur][foff];

			// ---- which form do the tile's candidates need?  (uniform: every wave looks at the whole tile)
			bool need_general = false;
			{
				const int pi = lane & 31;
				const PixRange q = CS.pr[cur][pi];
				const bool live = x0 + pi < W && q.hi >= q.lo;
				bool bad = live && !(CS.pc[cur][pi][3] != 0.0);
#pragma unroll
				for (int it = 0; it < (CHUNK + 63)/64 + 1; ++it) {
					const int k = it*64 + lane, c = cs + k;
					const bool colbad = k < CHUNK + NCB && rfull[k] == 0;
					if (colbad && c >= cmin_raw && c <= cmax) bad = true;
					const unsigned long long m = __ballot(colbad);
					if (wv == 0 && lane == 0) S.badcols[it] = m;         // for phase 2 (read behind its first barrier)
				}
				need_general = __any(bad) != 0;
			}

			if (NWV == 8) { prog = 0; *(st_lds_vint *)&S.prog[wv][lane] = 0; __builtin_amdgcn_s_setprio(1); }
			// ---- this lane's pixel
			int e_min, e_max;
			{
				const PixRange q = CS.pr[cur][i];
				e_min = q.lo; e_max = q.hi;
				if (x >= W) { e_min = 0; e_max = -1; }
				if (e_max >= e_min) e_max = dense_cover_hi(e_min, e_max, NCB, G, PAD);
			}
			const bool lall = CS.pc[cur][i][3] != 0.0;
			// certified arithmetic: the smallest sum3 of a candidate of this pixel for which the bound holds (from its sum2)
			const double sig3 = CERT ? A.cb.sigma3(CS.pc[cur][i][2]) : 0.0;
			const bool pix_exact = CERT && A.redo && !(sig3 < __builtin_inf());   // (cert_pixel_exact: the scan applies the same test)
			const size_t tile = (size_t)r*tiles_per_row + tx;
			ST_STAMP(3);                               // ranges, (re)staging, requests for the next tile, form of the tile
			if (e_max >= e_min) {
				const int lo = e_min > cs ? e_min : cs;
				const int hi = e_max < cs + CHUNK - 1 ? e_max : cs + CHUNK - 1;
				const int lo_e = PAD ? (lo & ~1) : lo;                  // blocks start on even columns (>= cs) / at the first column
				const int nblocks = hi >= lo ? (hi - lo_e + NCB)/NCB : 0;
				double *crow = A.cost + tile*(size_t)A.cstride*ST_TP + i;
				// phase 1: blocks of NCB candidates in the fast form (all taps usable on both sides): meanL, totalWeight,
				// sum2 and a_t = w_t*gl_t - meanL do not depend on the candidate (twoviewstereo.cpp:917-976, same order)
#ifdef SRH_EXPERIMENT
				const int exp_rep = g_strip_exp_repeat;
				for (int rep = 0; rep < exp_rep; ++rep)
#endif
				for (int b = g; b < (lall ? nblocks : 0); b += G) {
					const int c0 = lo_e + b*NCB;
					const int rc = c0 - cs;                 // tile column of the window's left edge
					// 16-byte reads need an even index: an odd edge reads the copy shifted by one column
					const double *rbase = (!PAD && (rc & 1)) ? &CS.rto[0][0] - 1 : &CS.rt[0][0];
					// bit j: candidate c0 + j is inside the pixel's range and its window in the other view is fully usable
					unsigned vm = 0;
#pragma unroll
					for (int j = 0; j < NCB; ++j) {
						const int c = c0 + j;
						vm |= (c >= lo && c <= hi && rfull[rc + j] != 0) ? 1u << j : 0u;
					}
					const bool fast = vm != 0;
					n_dev += __builtin_popcount(vm);
					// certified forms: a candidate the bound does not cover is NOT left to the scan (which would flag the whole pixel
					// for the per-pixel redo): its block is evaluated once more, here and now, in the reference's arithmetic, while the
					// window and the rows are in LDS.  And a pixel whose own window leaves the bound no room at all (sigma3 = +inf: a
					// flat window, sum2 ~ 0) takes the reference's arithmetic straight away -- the scan knows such a pixel's stored
					// values are the reference's very numbers (cert_pixel_exact).  A.redo = 0 (diagnostics): raw fused values, NaN
					// where uncertified.
					bool bad_blk = false;
					// the reference's two sweeps over the window for the block at c0 (FMAc: with fused multiply-adds; CERTc: values
					// stored for the certified scan, `bad_blk` raised when a candidate's bound is not covered)
					auto two_sweeps = [&](auto fma_c, auto cert_c) {
						constexpr bool FMAc = decltype(fma_c)::value, CERTc = decltype(cert_c)::value;
						const double mL = CS.pc[cur][i][0], tw = CS.pc[cur][i][1], s2 = CS.pc[cur][i][2];
						// Both passes are modulo-scheduled by hand (see twoview_dense_cost_kernel): a register is refilled
						// with the next row's value right after its last use, so the LDS latency is always a row ahead.
						double r_[NR], wv_[WS], acc[NCB];
						{
							const double2 *rp = reinterpret_cast<const double2 *>(rbase + s0*RW + rc);
							const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[wcur][0][i][0]);
#pragma unroll
							for (int m = 0; m < NR/2; ++m) { const double2 v = rp[m]; r_[2*m] = v.x; r_[2*m + 1] = v.y; }
#pragma unroll
							for (int m = 0; m < (WS - 1)/2; ++m) { const double2 v = wp[m]; wv_[2*m] = v.x; wv_[2*m + 1] = v.y; }
							wv_[WS - 1] = CS.w[wcur][0][i][WS - 1];
						}
#pragma unroll
						for (int j = 0; j < NCB; ++j) acc[j] = 0.0;
						__builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): the pre-header's reads have landed
#pragma unroll 1
						for (int row = 0; row < WS; ++row) {
							int seen = 0;
							prog_step(1, seen);
							const int nrow = row + 1 < WS ? row + 1 : 0;          // last refill = row 0, for pass 2
							const int nsl = s0 + nrow >= NS ? s0 + nrow - NS : s0 + nrow;
							const double2 *rp = reinterpret_cast<const double2 *>(rbase + nsl*RW + rc);
							const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[wcur][nrow][i][0]);
#pragma unroll
							for (int col = 0; col < WS; ++col) {
								if (FMAc) {
#pragma unroll
									for (int j = 0; j < NCB; ++j) acc[j] = __builtin_fma(wv_[col], r_[col + j], acc[j]);
								} else {
									double pr[NCB];
#pragma unroll
									for (int j = 0; j < NCB; ++j) pr[j] = wv_[col]*r_[col + j];
									__builtin_amdgcn_sched_barrier(0);
#pragma unroll
									for (int j = 0; j < NCB; ++j) acc[j] += pr[j];        // meanR += weight*gray
								}
								__builtin_amdgcn_sched_barrier(0);
								if (col & 1) {
									const double2 v = rp[col >> 1]; r_[col - 1] = v.x; r_[col] = v.y;
									const double2 u = wp[col >> 1]; wv_[col - 1] = u.x; wv_[col] = u.y;
									__builtin_amdgcn_sched_barrier(0);
								}
							}
#pragma unroll
							for (int m = (WS - 1)/2; m < NR/2; ++m) { const double2 v = rp[m]; r_[2*m] = v.x; r_[2*m + 1] = v.y; }
							wv_[WS - 1] = CS.w[wcur][nrow][i][WS - 1];
							prog_yield(seen);
						}
						double mR[NCB], s1[NCB], s3[NCB], av[WS];
#pragma unroll
						for (int col = 0; col < WS; ++col) av[col] = CS.lt[s0][i + col];
#pragma unroll
						for (int j = 0; j < NCB; ++j) { mR[j] = acc[j]/tw; s1[j] = 0.0; s3[j] = 0.0; }
						__builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 1
						for (int row = 0; row < WS; ++row) {
							int seen = 0;
							prog_step(3, seen);
							const int nrow = row + 1 < WS ? row + 1 : 0;
							const int nsl = s0 + nrow >= NS ? s0 + nrow - NS : s0 + nrow;
							const double2 *rp = reinterpret_cast<const double2 *>(rbase + nsl*RW + rc);
							const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[wcur][nrow][i][0]);
							const double *lp = &CS.lt[nsl][i];
#pragma unroll
							for (int col = 0; col < WS; ++col) {
								const double wt = wv_[col];
								if (FMAc) {
									const double a = __builtin_fma(wt, av[col], -mL);
									double bb[NCB];
#pragma unroll
									for (int j = 0; j < NCB; ++j) bb[j] = __builtin_fma(wt, r_[col + j], -mR[j]);
									__builtin_amdgcn_sched_barrier(0);
#pragma unroll
									for (int j = 0; j < NCB; ++j) { s1[j] = __builtin_fma(a, bb[j], s1[j]); s3[j] = __builtin_fma(bb[j], bb[j], s3[j]); }
								} else {
									double bb[NCB], u1[NCB], u3[NCB];
									const double pa = wt*av[col];
#pragma unroll
									for (int j = 0; j < NCB; ++j) bb[j] = wt*r_[col + j];
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
								}
								__builtin_amdgcn_sched_barrier(0);
								av[col] = lp[col];
								if (col & 1) {
									const double2 v = rp[col >> 1]; r_[col - 1] = v.x; r_[col] = v.y;
									const double2 u = wp[col >> 1]; wv_[col - 1] = u.x; wv_[col] = u.y;
								}
								__builtin_amdgcn_sched_barrier(0);
							}
#pragma unroll
							for (int m = (WS - 1)/2; m < NR/2; ++m) { const double2 v = rp[m]; r_[2*m] = v.x; r_[2*m + 1] = v.y; }
							wv_[WS - 1] = CS.w[wcur][nrow][i][WS - 1];
							prog_yield(seen);
						}
#pragma unroll
						for (int j = 0; j < NCB; ++j) {
							const int c = c0 + j;
							if (c >= lo && c <= hi && rfull[rc + j] != 0) {
								const double v = 255*(1.0 - fabs(s1[j]) / sqrt(s2 * s3[j]));
								if (CERTc) {
									if (!(s3[j] >= sig3) && A.redo) bad_blk = true;
									crow[(size_t)(c - e_min)*ST_TP] = !(s3[j] >= sig3) ? __builtin_nan("") : (v > A.cb.m_hi ? A.max_color_diff : v);
								} else crow[(size_t)(c - e_min)*ST_TP] = (v < A.max_color_diff) ? v : A.max_color_diff;
							}
							__builtin_amdgcn_sched_barrier(0);
						}
					};
					if (fast && ONEPASS && !pix_exact) {
						// ---- certified ONE-PASS form (srh_internal.hpp, CertBound): P = sum w r, Q = sum ((w l - meanL) w) r,
						// U = sum w^2 r^2 in one sweep over the window; registers are refilled in place a row ahead, as below
						const double mL = CS.pc[cur][i][0], itw = CS.pc[cur][i][3], s2 = CS.pc[cur][i][2];   // (slot 3 of a pixel with every tap usable: 1/totalWeight)
						double r_[NR], q_[NR], w_[WS], l_[WS], P_[NCB], Q_[NCB], U_[NCB];
						{
							const double2 *rp = reinterpret_cast<const double2 *>(rbase + s0*RW + rc);
							const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[wcur][0][i][0]);
#pragma unroll
							for (int m = 0; m < NR/2; ++m) { const double2 v = rp[m]; r_[2*m] = v.x; r_[2*m + 1] = v.y; }
#pragma unroll
							for (int m = 0; m < (WS - 1)/2; ++m) { const double2 v = wp[m]; w_[2*m] = v.x; w_[2*m + 1] = v.y; }
							w_[WS - 1] = CS.w[wcur][0][i][WS - 1];
#pragma unroll
							for (int col = 0; col < WS; ++col) l_[col] = CS.lt[s0][i + col];
						}
#pragma unroll
						for (int j = 0; j < NCB; ++j) { P_[j] = 0.0; Q_[j] = 0.0; U_[j] = 0.0; }
						__builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 1
						for (int row = 0; row < WS; ++row) {
							int seen = 0;
							prog_step(4, seen);
							const int nrow = row + 1 < WS ? row + 1 : 0;          // (the last refill is never used)
							const int nsl = s0 + nrow >= NS ? s0 + nrow - NS : s0 + nrow;
							const double2 *rp = reinterpret_cast<const double2 *>(rbase + nsl*RW + rc);
							const double2 *wp = reinterpret_cast<const double2 *>(&CS.w[wcur][nrow][i][0]);
							const double *lp = &CS.lt[nsl][i];
#pragma unroll
							for (int k = 0; k < NCB - 1; ++k) q_[k] = r_[k]*r_[k];
#pragma unroll
							for (int col = 0; col < WS; ++col) {
								q_[col + NCB - 1] = r_[col + NCB - 1]*r_[col + NCB - 1];
								const double a = __builtin_fma(w_[col], l_[col], -mL);
								const double c = a*w_[col], d = w_[col]*w_[col];
								__builtin_amdgcn_sched_barrier(0);
#pragma unroll
								for (int j = 0; j < NCB; ++j) P_[j] = __builtin_fma(w_[col], r_[col + j], P_[j]);
#pragma unroll
								for (int j = 0; j < NCB; ++j) Q_[j] = __builtin_fma(c, r_[col + j], Q_[j]);
#pragma unroll
								for (int j = 0; j < NCB; ++j) U_[j] = __builtin_fma(d, q_[col + j], U_[j]);
								__builtin_amdgcn_sched_barrier(0);
								l_[col] = lp[col];
								if (col & 1) {
									const double2 v = rp[col >> 1]; r_[col - 1] = v.x; r_[col] = v.y;
									const double2 u = wp[col >> 1]; w_[col - 1] = u.x; w_[col] = u.y;
								}
								__builtin_amdgcn_sched_barrier(0);
							}
#pragma unroll
							for (int m = (WS - 1)/2; m < NR/2; ++m) { const double2 v = rp[m]; r_[2*m] = v.x; r_[2*m + 1] = v.y; }
							w_[WS - 1] = CS.w[wcur][nrow][i][WS - 1];
							prog_yield(seen);
						}
						constexpr double TT = (double)(WS*WS);
						// every candidate of the block is finished (arithmetic on a column outside the range harms nobody); only the
						// store looks at the candidate's bit -- no LDS read, wait and branch per candidate
						// (SA = sum of fma(w_t, l_t, -meanL) is the PIXEL's: the weights kernels leave it in pconst slot 4 -- round 5 summed it
						// again in every block, 121 additions)
						const double zmax2 = A.cb.zmax2, mhi = A.cb.m_hi, mcd = A.max_color_diff, SA = CS.pc[cur][i][4];
						const bool redo = A.redo != 0;
#pragma unroll
						for (int j = 0; j < NCB; ++j) {
							bool okc;
							const double v = onepass_finish(P_[j], Q_[j], U_[j], SA, itw, s2, TT, sig3, zmax2, okc);
							const bool st = (vm >> j) & 1u;
							bad_blk |= st & !okc & redo;
							if (st) crow[(size_t)(c0 + j - e_min)*ST_TP] = !okc ? __builtin_nan("") : (v > mhi ? mcd : v);
						}
					} else if (fast && !ONEPASS && !pix_exact) two_sweeps(std::integral_constant<bool, FMA>(), std::integral_constant<bool, CERT>());
					if (fast && CERT && (pix_exact || bad_blk)) two_sweeps(std::false_type(), std::false_type());   // (bad_blk is only raised when A.redo)
				}
			}
			if (NWV == 8) __builtin_amdgcn_s_setprio(2);
			ST_STAMP(4);                               // block loops
			// phase 2: the remaining candidates in the general form (any validity pattern), compacted into an LDS work
			// list and spread over all lanes of the workgroup (image borders, masks): border tiles only
			if (need_general) {
				if (tid == 0) { S.glist_n = 0; S.gsingle_n = 0; }
				__syncthreads();
				for (int p = tid; p < ST_TP*Smem::NBMAX; p += NT) {
					const int pi = p / Smem::NBMAX, b = p % Smem::NBMAX;
					const PixRange q = CS.pr[cur][pi];
					int qlo = q.lo, qhi = q.hi;
					if (x0 + pi >= W) { qlo = 0; qhi = -1; }
					if (qhi >= qlo) qhi = dense_cover_hi(qlo, qhi, NCB, G, PAD);
					if (qhi > cs + CHUNK - 1) qhi = cs + CHUNK - 1;
					if (qhi < qlo) continue;
					const int c0 = (PAD ? (qlo & ~1) : qlo) + b*NCB;
					if (c0 > qhi) continue;
					const bool pall = CS.pc[cur][pi][3] != 0.0;
					// candidates of the block inside the pixel's range ...
					unsigned need = 0xffu;
					if (qlo > c0) need &= 0xffu << (qlo - c0);
					if (c0 + NCB - 1 > qhi) need &= 0xffu >> (c0 + NCB - 1 - qhi);
					// ... that phase 1 has not done: all of them when the pixel has unusable taps of its own, else those
					// whose own window is not fully usable
					if (pall) {
						const int rc = c0 - cs, wd = rc >> 6, sh = rc & 63;
						unsigned long long bits = CS.badcols[wd] >> sh;
						if (sh) bits |= CS.badcols[wd + 1] << (64 - sh);
						need &= (unsigned)bits;
					}
					if (!need) continue;
					// Only a block whose 8 candidates all need it takes the blocked form (~2.4 fast blocks of one lane whatever it
					// stores; rows next to the top / bottom border, pixels with unusable taps of their own).  The handful of
					// columns next to the left / right border go candidate by candidate: 32 pixels x 5..6 of them fill three
					// waves for one candidate's time, where 32 blocked tasks would hold half a wave for four times as long
					// while the other waves wait at the tile barrier.
					bool as_block = __builtin_popcount(need) == NCB;
					if (!as_block) {
						const int at = atomicAdd(&S.gsingle_n, __builtin_popcount(need));
						if (at + __builtin_popcount(need) <= Smem::GL_SINGLES) {
							int k = 0;
#pragma unroll
							for (int j = 0; j < NCB; ++j)
								if ((need >> j) & 1u) { S.glist[Smem::GL_CAP - 1 - (at + k)] = (unsigned short)(pi*512 + (c0 + j - cs)); ++k; }
						} else {
							// no room: the reserved entries (if any lie inside the list) are marked void, the block goes to the block list
							for (int k = at; k < at + __builtin_popcount(need) && k < Smem::GL_SINGLES; ++k) S.glist[Smem::GL_CAP - 1 - k] = 0xffffu;
							as_block = true;
						}
					}
					if (as_block) S.glist[atomicAdd(&S.glist_n, 1)] = (unsigned short)(pi*64 + b);
				}
				__syncthreads();
				ST_STAMP(5);                           // phase 2: work lists
#ifdef SRH_PROFILE_PHASES
				++n_gen;
#endif
				const int nblk = S.glist_n;
				const int nsingle = S.gsingle_n < Smem::GL_SINGLES ? S.gsingle_n : Smem::GL_SINGLES;
				for (int q = tid; q < nblk; q += NT) {
					const int pi = S.glist[q] >> 6, b = S.glist[q] & 63;
					const PixRange pq = CS.pr[cur][pi];
					int qhi = dense_cover_hi(pq.lo, pq.hi, NCB, G, PAD);
					if (qhi > cs + CHUNK - 1) qhi = cs + CHUNK - 1;
					const int c0 = (PAD ? (pq.lo & ~1) : pq.lo) + b*NCB;
					const bool pall = CS.pc[cur][pi][3] != 0.0;
					unsigned store = 0;
#pragma unroll
					for (int j = 0; j < NCB; ++j) {
						const int c = c0 + j;
						if (c >= pq.lo && c <= qhi && !(pall && rfull[c - cs])) { store |= 1u << j; ++n_dev; }
					}
					strip_select_block<R, NBUF>(CS, wcur, s0, pi, c0 - cs, store,
					                            A.cost + (tile*(size_t)A.cstride)*ST_TP + (ptrdiff_t)(c0 - pq.lo)*ST_TP + pi,
					                            A.weight_cutoff, A.bad_ret, A.max_color_diff,
					                            y < R ? R - y : 0, y + R > A.H - 2 ? WS - (y + R - (A.H - 2)) : WS);
				}
				ST_STAMP(6);                           // phase 2: blocked select form
				for (int q = tid; q < nsingle; q += NT) {
					const unsigned short e = S.glist[Smem::GL_CAP - 1 - q];
					if (e == 0xffffu) continue;
					const int pi = e >> 9, k = e & 511;
					++n_dev;
					A.cost[(tile*(size_t)A.cstride + (size_t)(cs + k - CS.pr[cur][pi].lo))*ST_TP + pi] =
						strip_cost_general<R, NBUF>(CS, wcur, s0, pi, k, A.weight_cutoff, A.bad_ret, A.max_color_diff);
				}
				ST_STAMP(7);                           // phase 2: single candidates
				__syncthreads();     // the select forms read any pixel's window: all of it done before a window is replaced
			}
			// ---- one window buffer: the wave replaces its own pixels' windows as soon as it has left them
			if (NBUF == 1 && has_next) issue_tile_inputs(r + 1, nxt, 0, false, true);
			ST_STAMP(2);                               // (barrier after phase 2, window requests: with the barrier time)
#ifdef SRH_PROFILE_PHASES
			++n_tiles;
#endif
		}
	}
	block_count_add(&A.cnt->n_eval_device, n_dev);
#ifdef SRH_PROFILE_PHASES
	if (lane == 0) {
		for (int k = 0; k < 8; ++k) atomicAdd(&A.cnt->dbg_phase[k], ph[k]);
		atomicAdd(&A.cnt->dbg_blocks, n_tiles);
		atomicAdd(&A.cnt->dbg_cycles, n_gen);
		for (int k = 0; k < 8; ++k) atomicAdd(&A.cnt->dbg_wave[8*k + (wv & 7)], ph[k]);
		atomicAdd(&A.cnt->dbg_total_cycles, (unsigned long long)(__builtin_readcyclecounter() - t_begin));
		atomicAdd(&A.cnt->dbg_waves, 1ull);
	}
#endif
#undef ST_STAMP
}

template <int R, int NWV, int NBUF, int AR>
static void launch_strip_variant(hipStream_t st, const StripArgs &a, int num_cus)
{
	typedef StripSmem<R, NBUF> Smem;
	size_t lds = sizeof(Smem);
	(void)hipFuncSetAttribute((const void *)twoview_strip_cost_kernel<R, NWV, NBUF, AR>,
	                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	int grid = num_cus*(NBUF == 2 ? 1 : 2);
	if (grid > a.nitems) grid = a.nitems;
	if (grid < 1) grid = 1;
	hipLaunchKernelGGL((twoview_strip_cost_kernel<R, NWV, NBUF, AR>), dim3((unsigned)grid), dim3(NWV*64), lds, st, a);
}

template <int R, int NWV, int NBUF>
static void launch_strip_arith(hipStream_t st, const StripArgs &a, int num_cus, int arith)
{
	if (arith == 5) launch_strip_variant<R, NWV, NBUF, 5>(st, a, num_cus);
	else if (arith == 3) launch_strip_variant<R, NWV, NBUF, 3>(st, a, num_cus);
	else if (arith == 1) launch_strip_variant<R, NWV, NBUF, 1>(st, a, num_cus);
	else launch_strip_variant<R, NWV, NBUF, 0>(st, a, num_cus);
}

// form: 0 = by the candidate range (16 block lanes only pay when a pixel has at least 17 blocks), 4 / 8 = forced
int strip_block_lanes(int cstride, int form) {
	if (form == 4) return 8;
	if (form == 8) return 16;
	return cstride > 16*ST_NCB ? 16 : 8;
}

bool launch_twoview_strip_cost(hipStream_t st, const ViewDev *views, int ref, int oth, int width, int height,
                               const srh_params &P, int y0, int nrows, const double *wimg, const double *pconst,
                               const PixRange *prange, const double *ref_tvp, const double *oth_tvp,
                               const uint8_t *oth_fullp, double *cost, int cstride, Counters *cnt, int arith, int num_cus,
                               int lanes, bool raw)
{
	(void)views; (void)ref; (void)oth;
	StripArgs a;
	a.W = width; a.H = height; a.y0 = y0; a.nrows = nrows;
	a.wimg = wimg; a.pconst = pconst; a.prange = prange;
	a.ref_tvp = ref_tvp; a.oth_tvp = oth_tvp; a.oth_fullp = oth_fullp;
	a.cost = cost; a.cstride = cstride; a.cnt = cnt;
	// tall items first: 3/4 of the rows in 16-row items, 2/3 of the rest in 8-row items, the remainder in 4-row items
	a.n1 = ((nrows*3/4)/16)*16;
	a.n2 = a.n1 + (((nrows - a.n1)*2/3)/8)*8;
	const int nseg = a.n1/16 + (a.n2 - a.n1)/8 + (nrows - a.n2 + 3)/4;
	a.nitems = nseg*((width + ST_TP - 1)/ST_TP);
	a.weight_cutoff = P.weight_cutoff; a.bad_ret = P.bad_ret; a.max_color_diff = P.max_color_diff;
	a.cb = cert_bound(P);
	a.redo = raw ? 0 : 1;
	const bool wide = lanes == 16;
	switch (P.window_radius) {
	case 5:
		if (wide) launch_strip_arith<5, 8, 2>(st, a, num_cus, arith); else launch_strip_arith<5, 4, 1>(st, a, num_cus, arith);
		return true;
	case 2:
		if (wide) launch_strip_arith<2, 8, 2>(st, a, num_cus, arith); else launch_strip_arith<2, 4, 1>(st, a, num_cus, arith);
		return true;
	default: return false;
	}
}

} // namespace srh
