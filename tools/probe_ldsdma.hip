// probe_ldsdma.hip - how fast does an L2-resident operand stream (the 520 KB of slice operands of the 2048 x 201 filter)
// get into a CU's LDS? Every block (512 threads, 1 per CU) walks the same buffer round after round:
//   dma   global_load_lds_dwordx4 (global -> LDS, no registers), `inflight` KB per wave outstanding
//   reg   global_load_dwordx4 -> VGPR -> ds_write_b128
//   rd    global_load_dwordx4 -> VGPR only (what the L2 -> CU path itself delivers)
// hipcc --offload-arch=gfx950 -O3 tools/probe_ldsdma.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr unsigned BUF = 520u * 1024u;  // bytes walked by every block
constexpr unsigned TH = 512;

template <int MODE, int DEPTH>  // DEPTH: 1 KB pieces per wave in flight between waits
__global__ void __launch_bounds__(TH) stream(const uint4* __restrict__ p, unsigned rounds, unsigned long long* out) {
    extern __shared__ uint4 lds[];  // DEPTH KB per wave
    const unsigned lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds + wave * (DEPTH * 1024u);
    const unsigned voff = lane * 16u;
    const unsigned pieces = BUF / 1024u;  // 1 KB pieces; wave w takes pieces w, w + 8, ...
    unsigned acc = 0;
    for (unsigned r = 0; r < rounds; r++) {
        for (unsigned pc = wave; pc + 8u * (DEPTH - 1) < pieces; pc += 8u * DEPTH) {
            if (MODE == 0) {
#pragma unroll
                for (int d = 0; d < DEPTH; d++) {
                    const char* sb = reinterpret_cast<const char*>(p) + (size_t)(pc + 8u * d) * 1024u;
                    const unsigned m0v = lds_base + d * 1024u;
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(sb) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                uint4 v[DEPTH];
#pragma unroll
                for (int d = 0; d < DEPTH; d++) v[d] = p[(size_t)(pc + 8u * d) * 64u + lane];
                if (MODE == 1) {
#pragma unroll
                    for (int d = 0; d < DEPTH; d++) lds[wave * (DEPTH * 64u) + d * 64u + lane] = v[d];
                } else {
#pragma unroll
                    for (int d = 0; d < DEPTH; d++) acc ^= v[d].x ^ v[d].w;
                }
            }
        }
    }
    if (MODE == 1) acc ^= lds[threadIdx.x].x;
    if (acc == 0x12345678u) atomicAdd(out, 1ull);
}

int main() {
    void* d;
    unsigned long long* out;
    CK(hipMalloc(&d, BUF + 4096));
    CK(hipMalloc(&out, 8));
    CK(hipMemset(d, 1, BUF + 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned rounds = 200, blocks = 256;
    auto run = [&](const char* name, auto kern, size_t ldsb) {
        float best = 1e9f;
        for (int it = 0; it < 4; it++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(TH), ldsb, 0, (const uint4*)d, rounds, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double bytes = (double)blocks * rounds * BUF;
        printf("%-28s %8.3f ms  %7.2f TB/s chip  %6.1f GB/s per CU  (%.1f B/clk/CU at 2.0 GHz)\n", name, best, bytes / best / 1e9, bytes / best / 1e6 / blocks,
               bytes / best / 1e6 / blocks / 2.0);
    };
    run("dma  1 KB/wave in flight", stream<0, 1>, 8 * 1024);
    run("dma  2 KB/wave in flight", stream<0, 2>, 16 * 1024);
    run("dma  4 KB/wave in flight", stream<0, 4>, 32 * 1024);
    run("dma  8 KB/wave in flight", stream<0, 8>, 64 * 1024);
    run("reg  1 KB/wave in flight", stream<1, 1>, 8 * 1024);
    run("reg  4 KB/wave in flight", stream<1, 4>, 32 * 1024);
    run("reg  8 KB/wave in flight", stream<1, 8>, 64 * 1024);
    run("rd   4 KB/wave in flight", stream<2, 4>, 1024);
    run("rd   8 KB/wave in flight", stream<2, 8>, 1024);
    return 0;
}
