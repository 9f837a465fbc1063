// Micro-benchmark: cost of a grid-wide barrier inside one resident launch (256 workgroups) on gfx950, with a cross-XCD visibility check, against
// the cost of a kernel boundary inside a captured hipGraph.  Build and run on the GPU box:
//     hipcc -O3 --offload-arch=gfx950 tools/grid_barrier_bench.hip -o /tmp/gridbar && /tmp/gridbar 256
// Measured on MI355X (round 2): 7.4-13.2 us per barrier at 256 workgroups (4.2-7.2 us at 128) vs 1.66 us per dependent launch in a graph --
// the reason the decode step stays a graph of small launches rather than one persistent kernel with grid-wide phases.
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target) {
	__syncthreads();
	bool ok = true;
	if (threadIdx.x == 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		int spins = 0;
		while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
			__builtin_amdgcn_s_sleep(1);
			if (++spins > 2000000) { ok = false; break; }
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	}
	__syncthreads();
	return ok;
}

__global__ __launch_bounds__(512) void bar_kernel(unsigned* counter, int* buf, int nphase, int* err, int payload) {
	const int b = blockIdx.x, nb = gridDim.x;
	for (int p = 1; p <= nphase; ++p) {
		// payload: each workgroup writes `payload` ints, reads another workgroup's after the barrier
		for (int i = threadIdx.x; i < payload; i += blockDim.x) buf[(size_t)b * payload + i] = p * 1000 + i;
		if (!grid_barrier(counter, (unsigned)(p * nb))) { if (threadIdx.x == 0) atomicAdd(err, 1000000); return; }
		const int src = (b + 37) % nb;
		for (int i = threadIdx.x; i < payload; i += blockDim.x)
			if (buf[(size_t)src * payload + i] != p * 1000 + i) atomicAdd(err, 1);
		// second barrier so that the next phase's writes do not race this phase's reads (as a real pipeline alternates buffers, count it separately)
		if (!grid_barrier(counter + 32, (unsigned)(p * nb))) { if (threadIdx.x == 0) atomicAdd(err, 1000000); return; }
	}
}

__global__ void tiny_kernel(int* buf) { if (threadIdx.x == 0 && blockIdx.x == 0) buf[0] += 1; }

int main(int argc, char** argv) {
	int nb = argc > 1 ? atoi(argv[1]) : 256, nphase = 200;
	unsigned* counter; int *buf, *err;
	HIPCHK(hipMalloc(&counter, 4096)); HIPCHK(hipMalloc(&buf, (size_t)nb * 4096 * 4)); HIPCHK(hipMalloc(&err, 4));
	hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
	for (int payload : {64, 1024, 4096}) {
		for (int rep = 0; rep < 2; ++rep) {
			HIPCHK(hipMemset(counter, 0, 4096)); HIPCHK(hipMemset(err, 0, 4));
			HIPCHK(hipEventRecord(e0));
			hipLaunchKernelGGL(bar_kernel, dim3(nb), dim3(512), 0, 0, counter, buf, nphase, err, payload);
			HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
			float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
			int h; HIPCHK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
			printf("grid %d payload %d ints: %.2f us per barrier (2 per phase), err %d\n", nb, payload, 1000.f * ms / (2 * nphase), h);
		}
	}
	// the alternative: dependent tiny launches in a captured graph
	hipStream_t s; HIPCHK(hipStreamCreate(&s));
	hipGraph_t g; hipGraphExec_t ge;
	HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
	for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(512), 0, s, buf);
	HIPCHK(hipStreamEndCapture(s, &g)); HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
	for (int rep = 0; rep < 3; ++rep) {
		HIPCHK(hipEventRecord(e0, s)); HIPCHK(hipGraphLaunch(ge, s)); HIPCHK(hipEventRecord(e1, s)); HIPCHK(hipEventSynchronize(e1));
		float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
		printf("graph of 200 dependent 256x512 launches: %.2f us per launch\n", 1000.f * ms / 200);
	}
	return 0;
}
