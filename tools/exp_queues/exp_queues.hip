// exp_queues.hip -- experiment: which HIP streams of one process really run LARGE launches side by side on gfx950?
// Two streams can share a hardware queue (then they serialise completely) or sit on queues served by the same dispatch
// pipe (then the second launch starts only when the first one's last workgroup has been DISPATCHED, which a pair of
// one-workgroup kernels cannot tell from real concurrency).  For every ordered pair (i, j): a launch A that needs
// `rounds` fillings of the machine goes to stream i, a one-workgroup launch B to stream j right behind it; printed is B's
// completion time in units of A's duration (small = B ran beside A).
//   hipcc --offload-arch=gfx950 -O2 -o tools/exp_queues/exp_queues tools/exp_queues/exp_queues.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin(long long ticks, unsigned* sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (sink && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *sink = 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 8;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;  // 0: default priority; 1: priorities cycle low / normal / high
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    printf("priority range: least %d greatest %d\n", least, greatest);
    std::vector<hipStream_t> s((size_t)n);
    for (int i = 0; i < n; ++i) {
        int pr = 0;
        if (mode == 1) pr = (i % 3 == 0) ? least : (i % 3 == 1 ? 0 : greatest);
        CK(hipStreamCreateWithPriority(&s[(size_t)i], hipStreamNonBlocking, pr));
    }
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const int big = 256 * 32 * 6;      // one-wavefront workgroups: six fillings of 256 CUs x 32 wavefronts
    const long long tick = 100 * 20;   // 20 us at 100 MHz
    for (int i = 0; i < n; ++i) {  // warm every stream
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[(size_t)i], 100, nullptr);
    }
    CK(hipDeviceSynchronize());
    printf("rows: stream of the large launch; columns: stream of the small one; entries: small launch's completion / large launch's duration\n");
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            if (i == j) { printf("   -  "); continue; }
            CK(hipEventRecord(e0, s[(size_t)i]));
            hipLaunchKernelGGL(spin, dim3(big), dim3(64), 0, s[(size_t)i], tick, nullptr);
            CK(hipEventRecord(e2, s[(size_t)i]));
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[(size_t)j], 100, nullptr);
            CK(hipEventRecord(e1, s[(size_t)j]));
            CK(hipDeviceSynchronize());
            float tb = 0, ta = 0;
            CK(hipEventElapsedTime(&tb, e0, e1));
            CK(hipEventElapsedTime(&ta, e0, e2));
            printf(" %5.2f", tb / ta);
        }
        printf("\n");
    }
    // how many large launches really overlap: k equal launches on the first k streams, wall time against one launch
    for (int k = 1; k <= n; ++k) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s[0]));
        CK(hipStreamSynchronize(s[0]));
        for (int i = 0; i < k; ++i) hipLaunchKernelGGL(spin, dim3(256 * 8), dim3(64), 0, s[(size_t)i], 100 * 100, nullptr);  // quarter filling, 100 us
        for (int i = 0; i < k; ++i) { CK(hipEventRecord(e1, s[(size_t)i])); CK(hipStreamWaitEvent(s[0], e1, 0)); }
        CK(hipEventRecord(e2, s[0]));
        CK(hipDeviceSynchronize());
        float t = 0;
        CK(hipEventElapsedTime(&t, e0, e2));
        printf("%d quarter-machine launches of 100 us on %d streams: %.3f ms\n", k, k, t);
    }
    return 0;
}
