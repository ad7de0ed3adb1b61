// FINDING (round 2, MI355X): an event recorded by a NODE of a captured hipGraph (hipEventRecordWithFlags ... hipEventRecordExternal)
// and waited on by an ordinary stream works with the ROCm 7.2 runtime of /opt/rocm -- this program: 10 launches, the waiting
// stream always sees the value written just before the node and finishes while the rest of the graph is still running, in every
// capture mode -- but the HIP runtime bundled with torch 2.10.0+rocm7.0 (the one a Python process gets) rejects the same call with
// hipErrorInvalidValue, and torch.cuda.Event(external=True) raises "External events are disallowed in rocm".  So the critical
// path of the training step cannot be ONE graph with mid-graph events; GraphedPipelined uses one graph per backward stage.
//   hipcc --offload-arch=gfx950 -O2 -o extev tools/probes/extev.hip && ./extev [variant bits: 1 non-blocking stream, 2 global mode, 4 relaxed, 8 fork/join]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
__global__ void spin_write(float* x, const float* val, int n, long spins) {
    long acc = 0;
    for (long i = 0; i < spins; ++i) acc += clock64() & 1;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = val[0] + (acc < 0 ? 1.f : 0.f);
}
__global__ void spin(long spins, float* sink) {
    long acc = 0;
    for (long i = 0; i < spins; ++i) acc += clock64() & 1;
    if (acc < 0) sink[0] = 1.f;
}
__global__ void read_x(const float* x, float* out) { out[0] = x[0]; }
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("ERR %s line %d\n", hipGetErrorString(r), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    int variant = argc > 1 ? atoi(argv[1]) : 0;
    float *x, *val, *out, *sink;
    int n = 1 << 20;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&val, 4)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&sink, 4));
    hipStream_t main_s, side_s;
    if (variant & 1) { CK(hipStreamCreateWithPriority(&main_s, hipStreamNonBlocking, 0)); } else { CK(hipStreamCreate(&main_s)); }
    CK(hipStreamCreate(&side_s));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(main_s, (variant & 2) ? hipStreamCaptureModeGlobal : (variant & 4) ? hipStreamCaptureModeRelaxed : hipStreamCaptureModeThreadLocal));
    if (variant & 8) {   // fork / join another stream into the capture first (like autograd / allocator side streams)
        hipEvent_t f; CK(hipEventCreateWithFlags(&f, hipEventDisableTiming));
        CK(hipEventRecord(f, main_s)); CK(hipStreamWaitEvent(side_s, f, 0));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, side_s, 10L, sink);
        CK(hipEventRecord(f, side_s)); CK(hipStreamWaitEvent(main_s, f, 0));
    }
    hipLaunchKernelGGL(spin_write, dim3(n / 256), dim3(256), 0, main_s, x, val, n, 200000L);     // ~ms of spinning, then X = val
    CK(hipEventRecordWithFlags(ev, main_s, hipEventRecordExternal));
    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, main_s, 400000L, sink);
    CK(hipStreamEndCapture(main_s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn)); printf("graph nodes %zu\n", nn);
    int bad = 0;
    for (int it = 1; it <= 10; ++it) {
        float v = (float)it, got = -1.f;
        CK(hipMemcpy(val, &v, 4, hipMemcpyHostToDevice));
        CK(hipGraphLaunch(ge, main_s));
        CK(hipStreamWaitEvent(side_s, ev, 0));
        hipLaunchKernelGGL(read_x, dim3(1), dim3(1), 0, side_s, x, out);
        CK(hipStreamSynchronize(side_s));
        hipEvent_t t; 
        CK(hipMemcpy(&got, out, 4, hipMemcpyDeviceToHost));
        // was the main stream still busy when the side read completed?  (overlap check)
        hipError_t q = hipStreamQuery(main_s);
        CK(hipStreamSynchronize(main_s));
        printf("it %d got %.0f want %.0f  main busy at side completion: %s\n", it, got, v, q == hipErrorNotReady ? "yes" : "no");
        if (got != v) bad++;
    }
    printf("mismatches %d\n", bad);
    return 0;
}
