// Does stream capture keep a COPY of a large by-value kernel argument (3 KB struct)?  Safe probe: the kernel only copies ints.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
struct Big { int v[760]; int n; int *out; };
__global__ void probe(Big b) { for (int i = threadIdx.x; i < b.n; i += blockDim.x) b.out[i] = b.v[i]; }
static void __attribute__((noinline)) enqueue(hipStream_t s, int *out, int seed) {
    Big b;
    for (int i = 0; i < 760; ++i) b.v[i] = seed + i;
    b.n = 760; b.out = out;
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, s, b);
}
static int __attribute__((noinline)) clobber(int depth) {
    volatile int junk[2048];
    for (int i = 0; i < 2048; ++i) junk[i] = 0x7fffffff - i;
    return depth > 0 ? clobber(depth - 1) + junk[depth] : junk[0];
}
int main() {
    int *out; hipMalloc(&out, 760 * sizeof(int));
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    enqueue(s, out, 1000);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    int host[760];
    for (int rep = 0; rep < 3; ++rep) {
        printf("clobber %d\n", clobber(8) & 1);
        hipMemsetAsync(out, 0, sizeof(host), s);
        hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 760; ++i) bad += host[i] != 1000 + i;
        printf("replay %d: %d wrong of 760 (first %d)\n", rep, bad, host[0]);
    }
    return 0;
}
