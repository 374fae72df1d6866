// struct_buffer.hip -- does an INDEXED buffer load (buffer_load_dwordx4 ... idxen, stride 16 in the descriptor) reach beyond
// 4 GiB on gfx950, is an index >= num_records answered with zeros, and does it cost the texture addressers anything over
// the raw form (32-bit byte offset, stride 0)?  The march kernels' gathers address a bricked volume by SLOT; with the
// indexed form a 16 GiB volume (BASELINE config 5) needs no 64-bit addresses and no `slot << 4` per corner.
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench/struct_buffer tools/ubench/struct_buffer.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u4 __attribute__((ext_vector_type(4)));
extern "C" __device__ u4 vr_struct_load_b128(__amdgpu_buffer_rsrc_t, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.ptr.buffer.load.v4i32");

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            printf("%s: %s\n", #x, hipGetErrorString(e));                             \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

__global__ void fill(u4* p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = u4{(unsigned)i, (unsigned)(i >> 32), (unsigned)i ^ 0xdeadbeefu, 7u};
}

__device__ __forceinline__ unsigned hash(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// every lane loads `per` slots spread over [0, n_slots) and checks their contents; bad += mismatches
__global__ void check_indexed(const u4* base, unsigned n_slots, unsigned n_records, int per, unsigned long long* bad, unsigned long long* oob_nonzero)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<u4*>(base), 16, (int)n_records, 0x00020000);
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long b = 0, o = 0;
    for (int k = 0; k < per; ++k) {
        const unsigned s = (unsigned)(((unsigned long long)hash(t * 131u + (unsigned)k) * n_slots) >> 32);
        const u4 v = vr_struct_load_b128(r, (int)s, 0, 0, 0);
        if (s < n_records) {
            if (v.x != s || v.z != (s ^ 0xdeadbeefu) || v.w != 7u) ++b;
        } else if (v.x | v.y | v.z | v.w) ++o;
    }
    if (b) atomicAdd(bad, b);
    if (o) atomicAdd(oob_nonzero, o);
}

// the packets' pattern: a wavefront gathers 8 "corners" around a random base slot, lanes a few slots apart
template <bool INDEXED>
__global__ void gather(const u4* base, unsigned n_slots, int iters, unsigned* sink)
{
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<u4*>(base), 16, (int)n_slots, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<u4*>(base), 0, (int)(n_slots * 16u - 1u), 0x00020000);
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned b0 = (unsigned)(((unsigned long long)hash(wave * 977u + (unsigned)it) * (n_slots - 4096u)) >> 32);
        unsigned s[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) s[c] = b0 + (lane & 7u) + ((lane >> 3) << 2) * 4u + (unsigned)(c & 1) + (unsigned)((c >> 1) & 1) * 4u + (unsigned)(c >> 2) * 16u;
        u4 v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if constexpr (INDEXED) v[c] = vr_struct_load_b128(ri, (int)s[c], 0, 0, 0);
            else v[c] = __builtin_amdgcn_raw_buffer_load_b128(rr, (int)(s[c] << 4), 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) acc += v[c].x ^ v[c].w;
    }
    if (acc == 0x12345u) sink[0] = acc;
}

int main()
{
    const size_t gib = 1ull << 30;
    size_t bytes = 6 * gib;
    u4* d = nullptr;
    CK(hipMalloc(&d, bytes));
    const size_t n = bytes / 16;
    hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, d, n);
    CK(hipDeviceSynchronize());
    unsigned long long *cnt = nullptr, h[2] = {0, 0};
    CK(hipMalloc(&cnt, 16));
    // 1: all of the 6 GiB in range (402 653 184 records of 16 B)
    CK(hipMemset(cnt, 0, 16));
    hipLaunchKernelGGL(check_indexed, dim3(4096), dim3(256), 0, 0, d, (unsigned)n, (unsigned)n, 64, cnt, cnt + 1);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost));
    printf("indexed loads over 6 GiB (records %zu): mismatches %llu of %llu\n", n, h[0], 4096ull * 256 * 64);
    // 2: the range check: records = 5 GiB worth, indices go up to 6 GiB worth
    const unsigned rec5 = (unsigned)(5 * gib / 16);
    CK(hipMemset(cnt, 0, 16));
    hipLaunchKernelGGL(check_indexed, dim3(4096), dim3(256), 0, 0, d, (unsigned)n, rec5, 64, cnt, cnt + 1);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost));
    printf("num_records = 5 GiB / 16: in-range mismatches %llu, out-of-range loads that returned non-zero %llu\n", h[0], h[1]);
    // 3: speed, raw vs indexed, inside 2 GiB (what the raw form can reach)
    unsigned* sink = nullptr;
    CK(hipMalloc(&sink, 4));
    const unsigned n2 = (unsigned)(2 * gib / 16);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        for (int idx = 0; idx < 2; ++idx) {
            CK(hipEventRecord(e0, 0));
            if (idx) hipLaunchKernelGGL(gather<true>, dim3(256 * 8), dim3(256), 0, 0, d, n2, 2000, sink);
            else hipLaunchKernelGGL(gather<false>, dim3(256 * 8), dim3(256), 0, 0, d, n2, 2000, sink);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double loads = 256.0 * 8 * 256 * 2000 * 8;
            printf("%s gather: %.3f ms, %.1f G lane-loads/s (%.2f TB/s of 16-byte lanes)\n", idx ? "indexed" : "raw    ", ms, loads / ms / 1e6, loads * 16 / ms / 1e9);
        }
    }
    // 4: indexed gather over the whole 6 GiB
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(gather<true>, dim3(256 * 8), dim3(256), 0, 0, d, (unsigned)n, 2000, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("indexed gather over 6 GiB: %.3f ms\n", ms);
    return 0;
}
