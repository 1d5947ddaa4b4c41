#include <hip/hip_runtime.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
__global__ void k(const double *src, double *out, int n, int shift) {
	extern __shared__ __align__(16) double s[];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	// each wave copies 1 KiB: 64 lanes x 16 B from src + shift (8-byte aligned only when shift is odd)
	const double *g = src + shift + w*128 + lane*2;
	__builtin_amdgcn_global_load_lds((gbl_void *)g, (lds_void *)(s + w*128), 16, 0, 0);
	// partial: only 16 lanes
	if (lane < 16) __builtin_amdgcn_global_load_lds((gbl_void *)(g + 512), (lds_void *)(s + 512 + w*32), 16, 0, 0);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__builtin_amdgcn_s_barrier();
	for (int i = threadIdx.x; i < 640; i += blockDim.x) out[i] = s[i];
}
int main() {
	const int N = 4096;
	double *h = new double[N], *d, *o, ho[640];
	for (int i = 0; i < N; ++i) h[i] = i;
	hipMalloc(&d, N*8); hipMalloc(&o, 640*8);
	hipMemcpy(d, h, N*8, hipMemcpyHostToDevice);
	for (int shift = 0; shift < 4; ++shift) {
		hipMemset(o, 0, 640*8);
		hipLaunchKernelGGL(k, dim3(1), dim3(256), 640*8, 0, d, o, N, shift);
		hipMemcpy(ho, o, 640*8, hipMemcpyDeviceToHost);
		int bad = 0;
		for (int i = 0; i < 512; ++i) if (ho[i] != shift + i) ++bad;
		for (int w = 0; w < 4; ++w) for (int j = 0; j < 32; ++j) if (ho[512 + w*32 + j] != shift + w*128 + 512 + j) ++bad;
		printf("shift %d: bad %d (first %g %g, part %g %g)\n", shift, bad, ho[0], ho[1], ho[512], ho[543]);
	}
	return 0;
}
