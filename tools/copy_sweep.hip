// Calibration sweep (not product code): which launch structure streams fastest on this MI355X?
// hipcc --offload-arch=gfx950 -O3 -o /tmp/copy_sweep tools/copy_sweep.hip && /tmp/copy_sweep
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy_gs(const v4f *__restrict__ a, v4f *__restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        v4f v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { if (NT) __builtin_nontemporal_store(v[u], b + i + u * stride); else b[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) b[i] = a[i];
}

// one-shot: each block copies a contiguous chunk of CHUNK float4 per thread
template <int PER, bool NT>
__global__ __launch_bounds__(256) void copy_chunk(const v4f *__restrict__ a, v4f *__restrict__ b, size_t n)
{
    const size_t base = (size_t)blockIdx.x * 256 * PER + threadIdx.x;
    v4f v[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) { const size_t i = base + (size_t)u * 256; if (i < n) v[u] = NT ? __builtin_nontemporal_load(a + i) : a[i]; }
#pragma unroll
    for (int u = 0; u < PER; ++u) { const size_t i = base + (size_t)u * 256; if (i < n) { if (NT) __builtin_nontemporal_store(v[u], b + i); else b[i] = v[u]; } }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main()
{
    const size_t bytes = 300ull << 20, n = bytes / 16;
    const int nsets = 4, iters = 40;
    std::vector<v4f *> A(nsets), B(nsets);
    for (int s = 0; s < nsets; ++s) { CK(hipMalloc(&A[s], bytes)); CK(hipMalloc(&B[s], bytes)); CK(hipMemset(A[s], 1, bytes)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 4; ++i) launch(A[i % nsets], B[i % nsets]);
        hipDeviceSynchronize(); hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) launch(A[i % nsets], B[i % nsets]);
        hipEventRecord(e1, 0); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f us  %7.1f GB/s\n", name, ms / iters * 1e3, 2.0 * bytes / (ms / iters * 1e-3) / 1e9);
    };
    run("hipMemcpyAsync DtoD", [&](v4f *a, v4f *b) { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        char nm[64];
        snprintf(nm, 64, "grid-stride unroll1 grid=%d", grid);  run(nm, [&](v4f *a, v4f *b) { hipLaunchKernelGGL((copy_gs<1, false>), dim3(grid), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, 64, "grid-stride unroll4 grid=%d", grid);  run(nm, [&](v4f *a, v4f *b) { hipLaunchKernelGGL((copy_gs<4, false>), dim3(grid), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, 64, "grid-stride unroll4 nt grid=%d", grid);  run(nm, [&](v4f *a, v4f *b) { hipLaunchKernelGGL((copy_gs<4, true>), dim3(grid), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, 64, "grid-stride unroll8 nt grid=%d", grid);  run(nm, [&](v4f *a, v4f *b) { hipLaunchKernelGGL((copy_gs<8, true>), dim3(grid), dim3(256), 0, 0, a, b, n); });
    }
    { const int g = (int)((n + 256 * 1 - 1) / (256 * 1)); run("one-shot 1 float4/thread", [&](v4f *a, v4f *b) { hipLaunchKernelGGL((copy_chunk<1, false>), dim3(g), dim3(256), 0, 0, a, b, n); }); }
    { const int g = (int)((n + 256 * 4 - 1) / (256 * 4)); run("one-shot 4 float4/thread", [&](v4f *a, v4f *b) { hipLaunchKernelGGL((copy_chunk<4, false>), dim3(g), dim3(256), 0, 0, a, b, n); }); }
    { const int g = (int)((n + 256 * 4 - 1) / (256 * 4)); run("one-shot 4 float4/thread nt", [&](v4f *a, v4f *b) { hipLaunchKernelGGL((copy_chunk<4, true>), dim3(g), dim3(256), 0, 0, a, b, n); }); }
    { const int g = (int)((n + 256 * 8 - 1) / (256 * 8)); run("one-shot 8 float4/thread nt", [&](v4f *a, v4f *b) { hipLaunchKernelGGL((copy_chunk<8, true>), dim3(g), dim3(256), 0, 0, a, b, n); }); }
    return 0;
}
