// stream_gap.hip -- what a HIP stream pays between two dependent kernel launches on gfx950, and what the bookkeeping the frame loop
// puts around a march launch adds to it.  The kernel is the persistent march kernel's launch shape (256 workgroups of 768 threads,
// 76 KiB of dynamic LDS) spinning for 100 us on the device clock from the moment its first wavefront starts, so that
// (wall time per launch) - 100 us is the idle time of the stream between launches.
//   A  kernels back to back, nothing else
//   B  + one hipEventRecord behind every launch (the context's slot_done)
//   C  B + a side stream that waits for that event, runs a small kernel (the sort) and records an event; the main stream waits for the
//      side stream's event of three launches ago in front of every launch (the launch order it reads)
//   D  A with 1 workgroup of 64 threads and no LDS (the floor)
//   E  C without the main stream's wait (the side stream's work alone)
//   F  C without the side stream's kernel (the main stream's wait for an event of another stream alone)
//   G  C with the main stream's wait left out whenever hipEventQuery says the event is complete, the host kept 2 launches ahead
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench/stream_gap tools/ubench/stream_gap.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            printf("%s: %s\n", #x, hipGetErrorString(e));                             \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

extern __shared__ float lds[];

__global__ void spin(unsigned long long ticks, unsigned long long* first, float* sink)
{
    // every workgroup spins until `ticks` of the 100 MHz clock after the launch's FIRST wavefront started
    __shared__ unsigned long long t0;
    if (threadIdx.x == 0) {
        const unsigned long long now = wall_clock64();
        atomicCAS(first, 0ull, now);
        t0 = *(volatile unsigned long long*)first;
    }
    __syncthreads();
    unsigned n = 0;
    while (wall_clock64() - t0 < ticks && n < 2000000u) {
        __builtin_amdgcn_s_sleep(4);
        ++n;
    }
    if (sink && n == 0xFFFFFFFFu) sink[threadIdx.x] = lds[threadIdx.x];
}
__global__ void clear(unsigned long long* first) { *first = 0ull; }
__global__ void small(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    unsigned n = 0;
    while (wall_clock64() - t0 < ticks && n < 2000000u) {
        __builtin_amdgcn_s_sleep(4);
        ++n;
    }
}

int main()
{
    const int N = 300, W = 50;
    const unsigned long long kTicks = 10000;  // 100 us
    hipStream_t s, side;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    unsigned long long* d_first;
    CK(hipMalloc(&d_first, (N + W) * sizeof(unsigned long long)));
    std::vector<hipEvent_t> ev(N + W), ev2(N + W);
    for (auto& v : ev) CK(hipEventCreateWithFlags(&v, hipEventDisableTiming));
    for (auto& v : ev2) CK(hipEventCreateWithFlags(&v, hipEventDisableTiming));
    const unsigned lds_bytes = 76 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    for (int variant = 0; variant < 7; ++variant) {
        const bool big = variant != 3;
        const dim3 grid(big ? 256 : 1), block(big ? 768 : 64);
        const unsigned lb = big ? lds_bytes : 0u;
        double per = 0.0;
        for (int pass = 0; pass < 2; ++pass) {  // (the first pass warms everything up)
            CK(hipMemsetAsync(d_first, 0, (N + W) * sizeof(unsigned long long), s));
            CK(hipDeviceSynchronize());
            std::chrono::steady_clock::time_point t0;
            for (int i = 0; i < N + W; ++i) {
                if (i == W) {
                    CK(hipStreamSynchronize(s));
                    CK(hipStreamSynchronize(side));
                    t0 = std::chrono::steady_clock::now();
                }
                if (variant == 6 && i >= 2) CK(hipEventSynchronize(ev[i - 2]));  // (the host at most 2 launches ahead)
                if ((variant == 2 || variant == 5) && i >= 3) CK(hipStreamWaitEvent(s, ev2[i - 3], 0));
                if (variant == 6 && i >= 3 && hipEventQuery(ev2[i - 3]) != hipSuccess) {
                    (void)hipGetLastError();
                    CK(hipStreamWaitEvent(s, ev2[i - 3], 0));
                }
                hipLaunchKernelGGL(spin, grid, block, lb, s, kTicks, d_first + i, (float*)nullptr);
                if (variant == 1 || variant == 2 || variant >= 4) CK(hipEventRecord(ev[i], s));
                if (variant == 2 || variant >= 4) {
                    CK(hipStreamWaitEvent(side, ev[i], 0));
                    if (variant != 5) hipLaunchKernelGGL(small, dim3(1), dim3(1024), 0, side, 6000ull);  // 60 us, as the sort of a C3 frame
                    CK(hipEventRecord(ev2[i], side));
                }
            }
            CK(hipStreamSynchronize(s));
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            CK(hipStreamSynchronize(side));
            per = dt / N * 1e6;
        }
        const char* names[7] = {"A back to back", "B + event record", "C + side-stream sort and wait", "D one small workgroup, back to back",
                                "E C without the main stream's wait", "F C without the side stream's kernel", "G C, waits elided by query, host 2 ahead"};
        printf("%-40s %7.2f us per launch = 100 us of kernel + %5.2f us\n", names[variant], per, per - 100.0);
    }
    return 0;
}
