// ext_event_probe.hip -- what does an event that rides on a dispatch (hipExtLaunchKernelGGL's stopEvent) do for (a) the host (hipEventQuery /
// hipEventSynchronize) and (b) another stream that waits for it (hipStreamWaitEvent)?   (profiling aid, not product)
//   hipcc --offload-arch=gfx950 -O2 -o ab_libs/ext_event_probe tools/ext_event_probe.hip && ab_libs/ext_event_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void spin(unsigned long long ticks, unsigned long long* out)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0) out[0] = wall_clock64(); // when the spinning kernel ended (100 MHz)
}
__global__ void stamp(unsigned long long* out) { if (threadIdx.x == 0) out[1] = wall_clock64(); } // when the waiting stream's kernel ran
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    unsigned long long* d; CK(hipMalloc(&d, 16)); unsigned long long h[2];
    for (int mode = 0; mode < 2; mode++)
    {
        hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableSystemFence));
        CK(hipMemset(d, 0, 16)); CK(hipDeviceSynchronize());
        const double t0 = now();
        if (mode == 0) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, nullptr, ev, 0, 100000ull /* 1 ms */, d);
        else { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, 100000ull, d); CK(hipEventRecord(ev, a)); }
        const hipError_t q = hipEventQuery(ev);
        CK(hipStreamWaitEvent(b, ev, 0));
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, b, d);
        CK(hipEventSynchronize(ev));
        const double t1 = now();
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("%s: query right after the launch = %s; hipEventSynchronize returned after %.0f us (the kernel spins 1000 us); the waiting stream's kernel ran %+.1f us after the spinning kernel ended\n",
               mode == 0 ? "stop event on the dispatch (hipExtLaunchKernelGGL)" : "hipEventRecord behind the launch", hipGetErrorName(q), t1 - t0, ((double)h[1] - (double)h[0]) / 100.0);
        CK(hipEventDestroy(ev));
    }
    return 0;
}
