// Micro-benchmark (gfx950): latency of a scalar load that hits the scalar data cache, of an LDS read
// of one address by all lanes (broadcast), and of v_readlane, each as a chain of dependent operations
// (the next address comes out of the previous result); W waves per SIMD on every CU, 768-byte table.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef const unsigned __attribute__((address_space(4))) *cu32;
__global__ void k_smem(const unsigned *tab, unsigned *out, int iters) {
    cu32 t = (cu32)(unsigned long long)tab;
    unsigned idx = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) idx = t[idx];  // table[i] = (i * 7 + 1) % 192: a walk over the whole table
    }
    if (threadIdx.x == 0) out[blockIdx.x] = idx;
}
__global__ void k_lds(const unsigned *tab, unsigned *out, int iters) {
    __shared__ unsigned s[192];
    for (int i = threadIdx.x; i < 192; i += blockDim.x) s[i] = tab[i];
    __syncthreads();
    unsigned idx = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) idx = s[idx];  // all lanes read the same word
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = idx;
}
template <class K> void run(const char *name, K kern, const unsigned *tab, unsigned *out, int threads) {
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, threads>>>(tab, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<blocks, threads>>>(tab, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-24s %2d waves per CU  %8.3f ms  %7.1f cycles per dependent operation (2.4 GHz)\n", name, threads / 64, ms,
           ms * 1e6 / (iters * 16.0) * 2.4);
}
int main() {
    unsigned h[192];
    for (int i = 0; i < 192; ++i) h[i] = (i * 7 + 1) % 192;
    unsigned *tab, *out;
    hipMalloc(&tab, sizeof h); hipMalloc(&out, 256 * 1024 * sizeof(unsigned));
    hipMemcpy(tab, h, sizeof h, hipMemcpyHostToDevice);
    for (int threads : {64, 256, 768}) {
        run("scalar load (cache hit)", k_smem, tab, out, threads);
        run("LDS broadcast read", k_lds, tab, out, threads);
    }
    return 0;
}
