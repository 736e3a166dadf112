// Probe (development): is LGKM_CNT safe with more than 15 LDS operations in flight?  One wave issues NR ds_read_b128 back to back (16-way
// bank-conflicting addresses, so that they return slowly), waits with a COUNTED s_waitcnt lgkmcnt(NR - 12) and immediately copies the
// first 12 results away; a wrapped 4-bit counter releases that wait before those reads have returned.
// hipcc --offload-arch=gfx950 -O2 -o lgkm_probe lgkm_probe.hip && ./lgkm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NR>
__global__ void k(unsigned* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 1024 / 4];
    const int lane = threadIdx.x;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = lane; i < 16384; i += 64) lds[i] = 0xdead0000u;                  // poison
        __syncthreads();
        for (int r = 0; r < NR; ++r) lds[(lane * 256 + r * 4) % 16384 + 0] = 0x1000u * (it & 15) + r * 64 + lane;      // value that read r of this lane must see
        __syncthreads();
        u32x4 v[NR];
        // poison the destination registers
#pragma unroll
        for (int r = 0; r < NR; ++r) v[r] = (u32x4){0xbad0bad0u, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < NR; ++r) asm volatile("" : "+v"(v[r]));
        const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)lds;
#pragma unroll
        for (int r = 0; r < NR; ++r) asm volatile("ds_read_b128 %0, %1" : "+v"(v[r]) : "v"(base + ((lane * 256 + r * 4) % 16384) * 4) : "memory");
        unsigned first[12];
        if (NR > 12) {
            if (NR - 12 == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            else if (NR - 12 == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            else if (NR - 12 == 12) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 12 && r < NR; ++r) asm volatile("v_mov_b32 %0, %1" : "=v"(first[r]) : "v"(v[r][0]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < 12 && r < NR; ++r) if (first[r] != 0x1000u * (it & 15) + r * 64 + lane) ++bad;
        __syncthreads();
    }
    atomicAdd(out, bad);
}
template <int NR> void run(const char* name) {
    unsigned* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipLaunchKernelGGL(k<NR>, dim3(1024), dim3(64), 0, 0, d, 200);
    unsigned h = 0; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("%s: %d reads in flight, counted wait for the first 12: %u wrong values of %u\n", name, NR, h, 1024u * 200u * 64u * 12u);
    hipFree(d);
}
int main() { run<12>("12"); run<16>("16"); run<20>("20"); run<24>("24"); return 0; }
