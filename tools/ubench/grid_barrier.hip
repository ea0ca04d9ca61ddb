// tools/ubench/grid_barrier.hip -- cost of a software grid barrier (atomic counter + generation flag, agent-scope release / acquire)
// across all XCDs of an MI355X, against the cost of a dependent kernel launch.  Decides whether a persistent one-launch training step
// for small batches (four grid barriers per step) can beat six dependent launches.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/grid_barrier.hip -o build/ubench_grid_barrier && build/ubench_grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct GridBar { unsigned count; unsigned gen; };
__device__ __forceinline__ void grid_barrier(GridBar* gb, unsigned nblocks, unsigned& gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);                                    // agent scope: this workgroup's global writes become visible
        const unsigned prev = __hip_atomic_fetch_add(&gb->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == nblocks - 1) {
            __hip_atomic_store(&gb->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&gb->gen, gen + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(&gb->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(1);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    ++gen;
    __syncthreads();
}
__global__ __launch_bounds__(256, 2) void k_bar(GridBar* gb, int n, float* data, int work) {
    unsigned gen = 0;
    float acc = 0.f;
    for (int i = 0; i < n; ++i) {
        // a little global traffic between barriers: every workgroup writes a line, then reads its neighbour's line of the previous round
        if (work) {
            data[(size_t)blockIdx.x * 256 + threadIdx.x] = (float)i + acc;
        }
        grid_barrier(gb, gridDim.x, gen);
        if (work) acc += data[(size_t)((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x];
    }
    if (acc == 12345.f) data[0] = acc;
}
__global__ void k_empty(float* d) { if (d == nullptr) d[0] = 1.f; }
int main() {
    GridBar* gb; float* data;
    hipMalloc(&gb, sizeof(GridBar)); hipMalloc(&data, 1024 * 256 * sizeof(float));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {64, 128, 250, 512}) for (int work : {0, 1}) {
        hipMemset(gb, 0, sizeof(GridBar));
        const int n = 2000;
        void* args[] = {&gb, (void*)&n, &data, (void*)&work};
        hipLaunchCooperativeKernel((void*)k_bar, dim3(blocks), dim3(256), args, 0, 0);      // warm-up
        hipDeviceSynchronize();
        hipMemset(gb, 0, sizeof(GridBar));
        hipEventRecord(a);
        hipError_t st = hipLaunchCooperativeKernel((void*)k_bar, dim3(blocks), dim3(256), args, 0, 0);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%3d workgroups, %s: %.2f us per grid barrier (%s)\n", blocks, work ? "with a 1 KiB write + neighbour read per round" : "bare", ms * 1e3 / n, hipGetErrorString(st));
    }
    // dependent empty launches for comparison
    hipEventRecord(a);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_empty, dim3(250), dim3(256), 0, 0, data);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("dependent empty launches (250 x 256): %.2f us each\n", ms * 1e3 / 2000);
    return 0;
}
