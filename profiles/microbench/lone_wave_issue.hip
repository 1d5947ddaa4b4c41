// lone_wave_issue.hip -- what one instruction of each kind costs a wave that is ALONE on its SIMD (the geodesic windows
// kernel: the 11 x 11 window in registers leaves room for one wave), eight independent register sets per kind so that no
// instruction waits for its producer.  Cycles at the nominal 2.4 GHz from the wall clock, and in s_memtime ticks.
// build: hipcc --offload-arch=gfx950 -O3 lone_wave_issue.hip -o lone_wave_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define R8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define R32(T) R8(T) R8(T) R8(T) R8(T)

template <int KIND>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k(double *out, const double *seed, int iters, unsigned long long *ticks) {
	__shared__ double lds[64*40];
	double a[8], b[8];
	typedef int v4i __attribute__((ext_vector_type(4)));
	v4i q0, q1, q2, q3, q4, q5, q6, q7;
	int ia[8];
#pragma unroll
	for (int j = 0; j < 8; ++j) { a[j] = seed[(threadIdx.x*8 + j) & 2047]; b[j] = seed[(threadIdx.x*8 + j + 977) & 2047]; ia[j] = j - 3; }
	q0 = (v4i){(int)threadIdx.x, 1, 2, 3}; q1 = q0 + 1; q2 = q0 + 2; q3 = q0 + 3; q4 = q0 + 4; q5 = q0 + 5; q6 = q0 + 6; q7 = q0 + 7;
	for (int j = threadIdx.x; j < 64*40; j += 64) lds[j] = seed[j & 2047];
	__syncthreads();
	const unsigned la = (unsigned)(size_t)(&lds[threadIdx.x]), la2 = (unsigned)(size_t)(&lds[2*threadIdx.x]);
	double *gp = out + (size_t)(blockIdx.x*64 + threadIdx.x)*2;                     // 16 bytes per lane, a wave's kilobyte contiguous
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it) {
#define ADD(j)  asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b[j]));
#define MUL(j)  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b[j]));
#define FMA(j)  asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[j]) : "v"(b[j]));
#define MIN(j)  asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b[j]));
#define CMP(j)  asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(a[j]), "v"(b[j]) : "vcc");
#define RND(j)  asm volatile("v_rndne_f64 %0, %1" : "=v"(a[j]) : "v"(b[j]));
#define LDX(j)  asm volatile("v_ldexp_f64 %0, %1, %2" : "=v"(a[j]) : "v"(b[j]), "v"(ia[j]));
#define CVT(j)  asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(ia[j]) : "v"(b[j]));
#define CND(j)  asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(ia[j]) : "v"(ia[(j + 1) & 7]), "v"(ia[(j + 2) & 7]) : "vcc");
#define MOV(j)  asm volatile("v_mov_b32 %0, %1" : "=v"(ia[j]) : "v"(ia[(j + 1) & 7]));
#define AWR(j)  asm volatile("v_accvgpr_write_b32 a" #j ", %0" :: "v"(ia[j]) : "a" #j);
#define ARD(j)  asm volatile("v_accvgpr_read_b32 %0, a" #j : "=v"(ia[j]));
#define RCP(j)  asm volatile("v_rcp_f64 %0, %1" : "=v"(a[j]) : "v"(b[j]));
#define DSC(j)  asm volatile("v_div_scale_f64 %0, vcc, %1, %1, %2" : "=v"(a[j]) : "v"(b[j]), "v"(b[(j + 1) & 7]) : "vcc");
#define DFM(j)  asm volatile("v_div_fmas_f64 %0, %1, %2, %2" : "=v"(a[j]) : "v"(b[j]), "v"(b[(j + 1) & 7]) : "vcc");
#define DFX(j)  asm volatile("v_div_fixup_f64 %0, %1, %2, %2" : "=v"(a[j]) : "v"(b[j]), "v"(b[(j + 1) & 7]));
#define DSR(j)  asm volatile("ds_read_b64 %0, %1 offset:" #j "*512" : "=v"(a[j]) : "v"(la));
#define DS2(j)  asm volatile("ds_read2_b64 %0, %1 offset0:" #j " offset1:" #j "+64" : "=v"(*(double2 *)&a[j & 6]) : "v"(la));
#define SAL(j)  asm volatile("s_and_b64 s[20:21], s[22:23], s[24:25]" ::: "s20", "s21", "scc");
#define E32(j)  asm volatile("v_exp_f32 %0, %1" : "=v"(ia[j]) : "v"(ia[(j + 1) & 7]));
#define WAIT    asm volatile("s_waitcnt lgkmcnt(0)");
// a 64-bit select as the compiler writes it (compare into VCC, two v_cndmask_b32 reading VCC back to back), the same with the
// mask in an SGPR pair, and with two other instructions between the two VCC readers
#define CPV(j)  asm volatile("v_cmp_gt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %3, %4, vcc" : "=v"(ia[j]) : "v"(a[j]), "v"(b[j]), "v"(ia[(j + 1) & 7]), "v"(ia[(j + 2) & 7]) : "vcc");
#define CPS(j)  asm volatile("v_cmp_gt_f64_e64 s[20:21], %1, %2\n\tv_cndmask_b32_e64 %0, %3, %4, s[20:21]" : "=v"(ia[j]) : "v"(a[j]), "v"(b[j]), "v"(ia[(j + 1) & 7]), "v"(ia[(j + 2) & 7]) : "s20", "s21");
#define CPD(j)  asm volatile("v_cmp_gt_f64 vcc, %2, %3\n\tv_cndmask_b32 %0, %4, %5, vcc\n\tv_cndmask_b32 %1, %5, %4, vcc" : "=v"(ia[j]), "=v"(ia[(j + 3) & 7]) : "v"(a[j]), "v"(b[j]), "v"(ia[(j + 1) & 7]), "v"(ia[(j + 2) & 7]) : "vcc");
#define CSD(j)  asm volatile("v_cmp_gt_f64_e64 s[20:21], %2, %3\n\tv_cndmask_b32_e64 %0, %4, %5, s[20:21]\n\tv_cndmask_b32_e64 %1, %5, %4, s[20:21]" : "=v"(ia[j]), "=v"(ia[(j + 3) & 7]) : "v"(a[j]), "v"(b[j]), "v"(ia[(j + 1) & 7]), "v"(ia[(j + 2) & 7]) : "s20", "s21");
#define CPX(j)  asm volatile("v_cmp_gt_f64 vcc, %2, %3\n\tv_cndmask_b32 %0, %4, %5, vcc\n\tv_add_f64 %2, %2, %3\n\tv_add_f64 %3, %3, %3\n\tv_cndmask_b32 %1, %5, %4, vcc" : "=v"(ia[j]), "=v"(ia[(j + 3) & 7]), "+v"(a[j]), "+v"(b[j]) : "v"(ia[(j + 1) & 7]), "v"(ia[(j + 2) & 7]) : "vcc");
#define CN3(j)  asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[22:23]" : "=v"(ia[j]) : "v"(ia[(j + 1) & 7]), "v"(ia[(j + 2) & 7]));
#define CN0(j)  asm volatile("v_cndmask_b32 %0, 0, %1, vcc" : "=v"(ia[j]) : "v"(ia[(j + 1) & 7]) : "vcc");
#define MAX(j)  asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b[j]));
#define D16(j)  asm volatile("ds_read_b128 %0, %1 offset:" #j "*1024" : "=v"(*(double2 *)&a[j & 6]) : "v"(la2));
#define DW8(j)  asm volatile("ds_write_b64 %0, %1 offset:" #j "*512" :: "v"(la), "v"(b[j]) : "memory");
#define DW16(j) asm volatile("ds_write_b128 %0, %1 offset:" #j "*1024" :: "v"(la2), "v"(q##j) : "memory");
#define GST(j)  asm volatile("global_store_dwordx4 %0, %1, off offset:" #j "*16" :: "v"(gp), "v"(q##j) : "memory");
#define GLD(j)  asm volatile("global_load_dwordx2 %0, %1, off offset:" #j "*8" : "=v"(a[j]) : "v"(gp) : "memory");
#define VWT     asm volatile("s_waitcnt vmcnt(0)");
#define U64(j)  asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(a[j]) : "v"(b[j]), "v"(b[(j + 1) & 7]));
#define AU32(j) asm volatile("v_add_u32 %0, %1, %2" : "=v"(ia[j]) : "v"(ia[(j + 1) & 7]), "v"(ia[(j + 2) & 7]));
#define RDL(j)  asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(ia[j]) : "s20");
#define SMV(j)  asm volatile("s_mov_b32 s20, 0x12345" ::: "s20");
#define SAX(j)  asm volatile("s_and_saveexec_b64 s[20:21], s[22:23]\n\ts_mov_b64 exec, s[20:21]" ::: "s20", "s21", "scc");
#define CMO(j)  asm volatile("v_cmp_o_f64 vcc, %0, %0" :: "v"(a[j]) : "vcc");
#define CME(j)  asm volatile("v_cmp_gt_f64_e64 s[20:21], %0, %1" :: "v"(a[j]), "v"(b[j]) : "s20", "s21");
		if constexpr (KIND == 0)  { R32(ADD) }
		if constexpr (KIND == 1)  { R32(MUL) }
		if constexpr (KIND == 2)  { R32(FMA) }
		if constexpr (KIND == 3)  { R32(MIN) }
		if constexpr (KIND == 4)  { R32(CMP) }
		if constexpr (KIND == 5)  { R32(RND) }
		if constexpr (KIND == 6)  { R32(LDX) }
		if constexpr (KIND == 7)  { R32(CVT) }
		if constexpr (KIND == 8)  { R32(CND) }
		if constexpr (KIND == 9)  { R32(MOV) }
		if constexpr (KIND == 10) { R32(AWR) }
		if constexpr (KIND == 11) { R32(ARD) }
		if constexpr (KIND == 12) { R32(RCP) }
		if constexpr (KIND == 13) { R32(DSC) }
		if constexpr (KIND == 14) { R32(DFM) }
		if constexpr (KIND == 15) { R32(DFX) }
		if constexpr (KIND == 16) { R8(DSR) WAIT R8(DSR) WAIT R8(DSR) WAIT R8(DSR) WAIT }
		if constexpr (KIND == 17) { R8(DSR) R8(DSR) R8(DSR) R8(DSR) WAIT }
		if constexpr (KIND == 18) { R32(SAL) }
		if constexpr (KIND == 19) { R32(E32) }
		// the sweeps' mix: per cell 4 add + 4 min + 4 single LDS reads used 8 cells later
		if constexpr (KIND == 20) { ADD(0) MIN(0) ADD(1) MIN(1) ADD(2) MIN(2) ADD(3) MIN(3) DSR(4) DSR(5) DSR(6) DSR(7) asm volatile("s_waitcnt lgkmcnt(14)");
		                            ADD(4) MIN(4) ADD(5) MIN(5) ADD(6) MIN(6) ADD(7) MIN(7) DSR(0) DSR(1) DSR(2) DSR(3) asm volatile("s_waitcnt lgkmcnt(14)");
		                            ADD(0) MIN(0) ADD(1) MIN(1) ADD(2) MIN(2) ADD(3) MIN(3) }
		// the same with nothing but the arithmetic
		if constexpr (KIND == 21) { ADD(0) MIN(0) ADD(1) MIN(1) ADD(2) MIN(2) ADD(3) MIN(3) ADD(4) MIN(4) ADD(5) MIN(5) ADD(6) MIN(6) ADD(7) MIN(7)
		                            ADD(0) MIN(0) ADD(1) MIN(1) ADD(2) MIN(2) ADD(3) MIN(3) ADD(4) MIN(4) ADD(5) MIN(5) ADD(6) MIN(6) ADD(7) MIN(7) }
		if constexpr (KIND == 22) { R8(DS2) WAIT R8(DS2) WAIT R8(DS2) WAIT R8(DS2) WAIT }
		if constexpr (KIND == 23) { R32(CN3) }
		if constexpr (KIND == 40) { R32(CPV) }
		if constexpr (KIND == 41) { R32(CPS) }
		if constexpr (KIND == 42) { R32(CPD) }
		if constexpr (KIND == 43) { R32(CSD) }
		if constexpr (KIND == 44) { R32(CPX) }
		if constexpr (KIND == 24) { R32(CN0) }
		if constexpr (KIND == 25) { R32(MAX) }
		if constexpr (KIND == 26) { R8(D16) WAIT R8(D16) WAIT R8(D16) WAIT R8(D16) WAIT }
		if constexpr (KIND == 27) { R32(DW8) WAIT }
		if constexpr (KIND == 28) { R32(DW16) WAIT }
		if constexpr (KIND == 29) { R8(GST) R8(GST) R8(GST) R8(GST) VWT }
		if constexpr (KIND == 30) { R8(GLD) R8(GLD) R8(GLD) R8(GLD) VWT }
		if constexpr (KIND == 31) { R32(U64) }
		if constexpr (KIND == 32) { R32(AU32) }
		if constexpr (KIND == 33) { R32(RDL) }
		if constexpr (KIND == 34) { R32(SMV) }
		if constexpr (KIND == 35) { R32(SAX) }
		if constexpr (KIND == 36) { R32(CMO) }
		if constexpr (KIND == 37) { R32(CME) }
		if constexpr (KIND == 38) { R8(DSR) R8(ADD) R8(MIN) R8(DSR) R8(ADD) R8(MIN) WAIT }
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	double s = 0;
#pragma unroll
	for (int j = 0; j < 8; ++j) s += a[j] + (double)ia[j];
	out[blockIdx.x*blockDim.x + threadIdx.x] = s;
	if (blockIdx.x == 0 && threadIdx.x == 0) *ticks = t1 - t0;
}

template <int KIND>
static void run(const char *name, int per_iter, double *d, const double *seed, unsigned long long *tk) {
	const int blocks = 1024, iters = 4000;
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, seed, iters, tk);
	(void)hipEventRecord(e0);
	for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, seed, iters, tk);
	(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
	unsigned long long t; (void)hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost);
	const double n = (double)iters*per_iter;
	printf("%-44s %7.3f ms  %6.2f cycles per instruction at 2.4 GHz  %6.2f s_memtime ticks  (%.2f ticks per ns)\n", name, ms, ms*1e-3*2.4e9/n, (double)t/n, (double)t/(ms*1e6));
}

int main() {
	std::vector<double> h(2048);
	unsigned long long s = 0x5EED;
	for (auto &v : h) { s = s*6364136223846793005ull + 1442695040888963407ull; v = 1.0 + ((double)(s >> 11)/9007199254740992.0 - 0.5)*1e-6; }
	double *d, *seed; unsigned long long *tk;
	(void)hipMalloc(&d, sizeof(double)*1024*64*2 + 4096); (void)hipMalloc(&seed, sizeof(double)*2048); (void)hipMalloc(&tk, 8);
	(void)hipMemcpy(seed, h.data(), sizeof(double)*2048, hipMemcpyHostToDevice);
	printf("one wave per SIMD (1024 workgroups of 64 lanes, waves_per_eu 1), 8 independent register sets per kind\n");
	run<0>("v_add_f64", 32, d, seed, tk);   run<1>("v_mul_f64", 32, d, seed, tk);   run<2>("v_fma_f64", 32, d, seed, tk);
	run<3>("v_min_f64", 32, d, seed, tk);   run<4>("v_cmp_gt_f64", 32, d, seed, tk); run<5>("v_rndne_f64", 32, d, seed, tk);
	run<6>("v_ldexp_f64", 32, d, seed, tk); run<7>("v_cvt_i32_f64", 32, d, seed, tk); run<8>("v_cndmask_b32", 32, d, seed, tk);
	run<9>("v_mov_b32", 32, d, seed, tk);   run<10>("v_accvgpr_write_b32", 32, d, seed, tk); run<11>("v_accvgpr_read_b32", 32, d, seed, tk);
	run<12>("v_rcp_f64", 32, d, seed, tk);  run<13>("v_div_scale_f64", 32, d, seed, tk); run<14>("v_div_fmas_f64", 32, d, seed, tk);
	run<15>("v_div_fixup_f64", 32, d, seed, tk);
	run<16>("ds_read_b64, a wait after every 8", 32, d, seed, tk); run<17>("ds_read_b64, a wait after every 32", 32, d, seed, tk);
	run<22>("ds_read2_b64, a wait after every 8", 32, d, seed, tk);
	run<18>("s_and_b64", 32, d, seed, tk);  run<19>("v_exp_f32", 32, d, seed, tk);
	run<23>("v_cndmask_b32_e64 (mask in an SGPR pair)", 32, d, seed, tk); run<24>("v_cndmask_b32 0, v, vcc", 32, d, seed, tk);
	printf("(the next five: cycles per GROUP of instructions)\n");
	run<40>("v_cmp_gt_f64 vcc + v_cndmask_b32 vcc", 32, d, seed, tk); run<41>("v_cmp_gt_f64_e64 s[] + v_cndmask_b32_e64 s[]", 32, d, seed, tk);
	run<42>("v_cmp_gt_f64 vcc + 2 v_cndmask_b32 vcc", 32, d, seed, tk); run<43>("v_cmp_gt_f64_e64 s[] + 2 v_cndmask_b32_e64 s[]", 32, d, seed, tk);
	run<44>("cmp vcc, cndmask vcc, 2 v_add_f64, cndmask vcc", 32, d, seed, tk);
	run<25>("v_max_f64", 32, d, seed, tk); run<36>("v_cmp_o_f64 vcc", 32, d, seed, tk); run<37>("v_cmp_gt_f64_e64 -> SGPR pair", 32, d, seed, tk);
	run<26>("ds_read_b128, a wait after every 8", 32, d, seed, tk);
	run<27>("ds_write_b64", 32, d, seed, tk); run<28>("ds_write_b128", 32, d, seed, tk);
	run<29>("global_store_dwordx4 (1 KB per instruction)", 32, d, seed, tk); run<30>("global_load_dwordx2 (L1 hits)", 32, d, seed, tk);
	run<31>("v_lshl_add_u64", 32, d, seed, tk); run<32>("v_add_u32", 32, d, seed, tk); run<33>("v_readlane_b32", 32, d, seed, tk);
	run<34>("s_mov_b32", 32, d, seed, tk); run<35>("s_and_saveexec_b64 + s_mov_b64 exec (pair)", 32, d, seed, tk);
	run<38>("8 ds_read_b64, 8 add, 8 min, twice, then a wait", 48, d, seed, tk);
	run<20>("sweep mix: (4 add + 4 min + 4 ds_read_b64) x 2.5", 34, d, seed, tk);
	run<21>("the same arithmetic alone: (add, min) x 16", 32, d, seed, tk);
	return 0;
}
