// Does a captured 256-byte memset reset words that kernels of the same graph update with atomics?  Safe probe.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void bump(unsigned *ctr, unsigned *seen) {
    if (threadIdx.x == 0) { const unsigned w = atomicAdd(ctr, 1u); seen[blockIdx.x] = w; }
}
int main() {
    unsigned *ws, *seen; hipMalloc(&ws, 4096); hipMalloc(&seen, 256 * 4);
    hipStream_t s; hipStreamCreate(&s);
    // eager warm-up like CapturedPlan does
    hipMemsetAsync(ws, 0, 256, s); hipLaunchKernelGGL(bump, dim3(64), dim3(64), 0, s, ws + 16, seen); hipStreamSynchronize(s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    hipMemsetAsync(ws, 0, 256, s);
    hipLaunchKernelGGL(bump, dim3(64), dim3(64), 0, s, ws + 16, seen);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    unsigned host[64], c;
    for (int rep = 0; rep < 4; ++rep) {
        hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        hipMemcpy(host, seen, sizeof(host), hipMemcpyDeviceToHost);
        hipMemcpy(&c, ws + 16, 4, hipMemcpyDeviceToHost);
        unsigned mx = 0;
        for (int i = 0; i < 64; ++i) mx = host[i] > mx ? host[i] : mx;
        printf("replay %d: counter after = %u, largest ticket seen = %u\n", rep, c, mx);
    }
    return 0;
}
