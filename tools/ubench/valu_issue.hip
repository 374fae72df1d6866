// valu_issue.hip -- how many cycles a wave64 vector instruction occupies a CDNA4 SIMD's issue port, by instruction kind and
// by wavefronts per SIMD.  Decides how to read SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU of the march kernels:
// is a v_pk_* worth one plain instruction or two, and does the port take an instruction every 2 or every 4 cycles?
//   hipcc -O3 --offload-arch=gfx950 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kIters = 2048;
constexpr int kPerIter = 32;  // instructions per loop iteration (16 independent accumulators, twice)

template <int KIND>
__global__ __launch_bounds__(1024) void issue_kernel(float* out, unsigned long long* cycles, float seed)
{
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = v2{seed + i, seed - i};
    const v2 b = v2{seed * 0.5f, seed * 0.25f}, c = v2{1.0f - seed, seed};
    unsigned sa[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) sa[i] = __builtin_amdgcn_readfirstlane((int)seed + i);
    unsigned long long sm = __builtin_amdgcn_read_exec();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if constexpr (KIND == 1) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(b.x));
                if constexpr (KIND == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if constexpr (KIND == 3) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if constexpr (KIND == 4) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if constexpr (KIND == 5) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i].x));
                if constexpr (KIND == 6) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b.x), "v"(c.x) : "vcc");
                if constexpr (KIND == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i].x));
                if constexpr (KIND == 8) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                if constexpr (KIND == 9) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i].x) : "v"(b.x));
                // scalar instructions: alone, and one beside every vector instruction (does the scalar stream cost issue time
                // of its own when the vector port is the busy one?)
                if constexpr (KIND == 10) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sa[i]) : : "scc");
                if constexpr (KIND == 11) asm volatile("v_add_f32 %0, %2, %0\n\ts_add_u32 %1, %1, 1" : "+v"(a[i].x), "+s"(sa[i]) : "v"(b.x) : "scc");
                if constexpr (KIND == 12) asm volatile("v_add_f32 %0, %2, %0\n\tv_add_f32 %0, %2, %0\n\ts_add_u32 %1, %1, 1" : "+v"(a[i].x), "+s"(sa[i]) : "v"(b.x) : "scc");
                if constexpr (KIND == 13) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_and_b64 %2, vcc, exec" : : "v"(a[i].x), "v"(b.x), "s"(sm) : "vcc", "scc");
                // round 4: what the hazard no-ops of a dependent chain cost (the packed / transcendental / compare-then-select
                // chains of the shading code: 92 s_nop in march_p2_kernel's loop)
                if constexpr (KIND == 14) asm volatile("s_nop 0");
                if constexpr (KIND == 15) asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n\ts_nop 0" : "+v"(a[0]) : "v"(b), "v"(c));           // ONE chain, per pair
                if constexpr (KIND == 16) asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %3, %1" : "+v"(a[0].x), "+v"(a[0].y) : "v"(b.x), "v"(c.x));  // the same two chains unpacked, per pair
                if constexpr (KIND == 17) asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n\tv_pk_fma_f32 %1, %2, %3, %1" : "+v"(a[0]), "+v"(a[1]) : "v"(b), "v"(c));  // two packed chains interleaved, per pair
                if constexpr (KIND == 18) asm volatile("v_cmp_ge_f32 vcc, 0, %0\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[0].x) : "v"(b.x) : "vcc");  // per triple
                if constexpr (KIND == 19) asm volatile("v_cmp_ge_f32 vcc, 0, %0\n\tv_add_f32 %2, %3, %2\n\tv_add_f32 %4, %3, %4\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[0].x) : "v"(b.x), "v"(a[1].x), "v"(c.x), "v"(a[2].x) : "vcc");  // the no-op's slots filled, per four
                if constexpr (KIND == 20) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0].x) : "v"(b.x), "v"(c.x));  // one dependent plain chain
                if constexpr (KIND == 21) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[0].x) : "v"(b.x));  // (a select as one instruction)
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += (float)sa[i];
    if (s == 12345.678f) out[0] = s + (float)sm;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
int run(const char* name, int cus)
{
    float* out;
    unsigned long long* cyc;
    CHECK(hipMalloc(&out, 16));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 32 * 2));
    printf("%-16s", name);
    for (int wps : {1, 2, 3, 4, 5, 8}) {   // wavefronts per SIMD: one workgroup of wps * 4 wavefronts per CU
        const int threads = wps * 4 * 64;
        const int block = threads > 1024 ? threads / 2 : threads, per_cu = threads > 1024 ? 2 : 1;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(issue_kernel<KIND>, dim3(cus * per_cu), dim3(block), 0, 0, out, cyc, 1.5f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(issue_kernel<KIND>, dim3(cus * per_cu), dim3(block), 0, 0, out, cyc, 1.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const int nw = cus * per_cu * (block / 64);
        std::vector<unsigned long long> h(nw);
        CHECK(hipMemcpy(h.data(), cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= nw;
        // s_memtime ticks at the shader clock (MI355X_MICROARCH.md): cycles per instruction per wavefront, and per SIMD
        const double per_wave = mean / ((double)kIters * kPerIter);
        printf("  wps %d: %6.2f cyc/inst/wave = %5.2f cyc/inst/SIMD (%.3f ms)", wps, per_wave, per_wave / wps, ms);
    }
    printf("\n");
    (void)hipFree(out);
    (void)hipFree(cyc);
    return 0;
}

int main()
{
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("CUs %d; %d x %d instructions per wavefront\n", cus, kIters, kPerIter);
    run<0>("v_fma_f32", cus);
    run<1>("v_add_f32", cus);
    run<2>("v_pk_fma_f32", cus);
    run<3>("v_pk_add_f32", cus);
    run<4>("v_pk_mul_f32", cus);
    run<5>("v_cvt_i32_f32", cus);
    run<6>("v_mad_u64_u32", cus);
    run<7>("v_rcp_f32", cus);
    run<8>("v_med3_f32", cus);
    run<9>("v_mov_b32", cus);
    run<10>("s_add_u32", cus);
    run<11>("v_add+s_add", cus);     // (per PAIR of instructions)
    run<12>("2 v_add+s_add", cus);   // (per TRIPLE)
    run<13>("v_cmp+s_and", cus);     // (per pair)
    run<14>("s_nop 0", cus);
    run<15>("pk_fma chain+nop", cus);    // (per pair)
    run<16>("2 fma chains", cus);        // (per pair)
    run<17>("2 pk_fma chains", cus);     // (per pair)
    run<18>("cmp+nop1+cndmask", cus);    // (per triple)
    run<19>("cmp+2 add+cndmask", cus);   // (per four)
    run<20>("fma chain", cus);
    run<21>("v_max chain", cus);
    return 0;
}
