// fp64_dep_distance.hip -- how far apart must dependent FP64 instructions be for one wave to keep the pipe full?
// The dense cost loops issue 8 independent v_mul_f64, then the 8 v_add_f64 that consume them (distance 8).
// Here: N independent chains (distance N) of separate multiply + add, N = 4..24, at 1 and 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off fp64_dep_distance.hip -o fp64_dep_distance
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
		for (int u = 0; u < 4; ++u) {
			double p[N];
#pragma unroll
			for (int j = 0; j < N; ++j) p[j] = m[j]*acc[(j + 1) % N];     // N independent v_mul_f64
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int j = 0; j < N; ++j) acc[j] += p[j];                  // N v_add_f64, each N instructions after its multiply
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int j = 0; j < N; ++j) m[j] = -m[j];
		}
	}
	double s = 0;
#pragma unroll
	for (int j = 0; j < N; ++j) s += acc[j];
	out[blockIdx.x*blockDim.x + threadIdx.x] = s;
}

template <int N>
static void run(double *d, const double *seed) {
	for (int wpc : {1, 2}) {
		const int blocks = 256*wpc, iters = 20000*8/N;
		hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
		for (int r = 0; r < 40; ++r) hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, d, seed, iters);   // warm, clocks settle
		(void)hipEventRecord(e0);
		for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k<N>, dim3(blocks), dim3(256), 0, 0, d, seed, iters);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
		const double lane_instr = (double)blocks*256*iters*4*2*N;
		printf("distance %2d  waves/SIMD %d: %.3f ms  %.2f T lane-instr/s  (%.1f %% of 256*4*16*2.4 GHz)\n", N, wpc, ms,
		       lane_instr/ms/1e9, 100.0*lane_instr/ms/1e9/39.32);
	}
}

int main() {
	std::vector<double> h(2048);
	unsigned long long s = 0x5EED;
	for (auto &v : h) { s = s*6364136223846793005ull + 1442695040888963407ull; v = 0.5 + (double)(s >> 11)/9007199254740992.0*1e-3 - 5e-4; }
	double *d, *seed;
	(void)hipMalloc(&d, sizeof(double)*256*256*16); (void)hipMalloc(&seed, sizeof(double)*2048);
	(void)hipMemcpy(seed, h.data(), sizeof(double)*2048, hipMemcpyHostToDevice);
	run<4>(d, seed); run<6>(d, seed); run<8>(d, seed); run<10>(d, seed); run<12>(d, seed); run<16>(d, seed); run<24>(d, seed);
	return 0;
}
