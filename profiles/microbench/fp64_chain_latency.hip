// fp64_chain_latency.hip -- FP64 dependent-issue latency: N independent chains per lane (N = 1, 2, 3, 4, 8), every
// instruction of a chain depending on the previous one (v_mul_f64 -> v_add_f64 -> v_mul_f64 ...), at 1, 2 and 4
// waves per SIMD.  The MultiViewStereo cost kernels evaluate ONE candidate per lane: meanR, sum1 and sum3 are single
// 25-term chains (multiviewstereo.cpp:113-189), so their rate is set by this latency, not by the issue rate.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off fp64_chain_latency.hip -o fp64_chain_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int N>
__global__ __launch_bounds__(256) void k(double *out, const double *seed, int iters) {
	double acc[N], m[N];
#pragma unroll
	for (int j = 0; j < N; ++j) { acc[j] = seed[(threadIdx.x*N + j) & 2047]; m[j] = seed[(threadIdx.x*N + j + 977) & 2047]; }
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
#pragma unroll
			for (int j = 0; j < N; ++j) acc[j] = m[j]*acc[j];            // v_mul_f64 on the chain
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int j = 0; j < N; ++j) acc[j] = acc[j] + m[j];          // v_add_f64 on the chain
			__builtin_amdgcn_sched_barrier(0);
		}
	}
	double s = 0;
#pragma unroll
	for (int j = 0; j < N; ++j) s += acc[j];
	out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

template <int N>
static void run(double *d, const double *seed) {
	for (int wpc : {1, 2, 4}) {
		const int blocks = 256*wpc, iters = 40000/N;
		hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
		for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, d, seed, iters);   // warm, clocks settle
		(void)hipEventRecord(e0);
		for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, d, seed, iters);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
		const double wave_instr_per_simd = (double)wpc*iters*8*2*N;                 // instructions issued on one SIMD
		const double cyc = ms*1e-3*2.4e9;
		printf("chains %d  waves/SIMD %d: %.3f ms  %.2f cycles per instruction and SIMD (2.4 GHz), per wave one every %.1f cycles\n",
		       N, wpc, ms, cyc/wave_instr_per_simd, cyc/(iters*8.0*2*N));
	}
}

int main() {
	std::vector<double> h(2048);
	unsigned long long s = 0x5EED;
	for (auto &v : h) { s = s*6364136223846793005ull + 1442695040888963407ull; v = 1.0 + ((double)(s >> 11)/9007199254740992.0 - 0.5)*1e-6; }
	double *d, *seed;
	(void)hipMalloc(&d, sizeof(double)*256*256*16); (void)hipMalloc(&seed, sizeof(double)*2048);
	(void)hipMemcpy(seed, h.data(), sizeof(double)*2048, hipMemcpyHostToDevice);
	run<1>(d, seed); run<2>(d, seed); run<3>(d, seed); run<4>(d, seed); run<8>(d, seed);
	return 0;
}
