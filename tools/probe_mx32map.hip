// probe_mx32map.hip — operand and result maps of v_mfma_scale_f32_32x32x64_f8f6f4 (FP4 A x FP6 / FP4 B, per-lane block scales),
// checked on the device against an exact host computation (diagnostics, not product):
//   A (FP4 E2M1): lane (i = lane & 31, kblk = lane >> 5) holds row i's values for k = 32 kblk + e in nibble e of dwords 0..3
//   B (FP6 E2M3 / FP4): lane (j = lane & 31, kblk) holds column j's values for k = 32 kblk + e in 6-bit field e of dwords 0..5
//                       (FP4: nibble e of dwords 0..3)
//   D: lane (j = lane & 31, h = lane >> 5), register r: row i = (r & 3) + 8 (r >> 2) + 4 h, column j
//   scales: E8M0 byte 0 of the lane's scale register applies to the lane's 32 values (op_sel 0)
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mx32map.hip -o tools/bin/probe_mx32map && tools/bin/probe_mx32map
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int FB>
__global__ void k(const v8i* A, const v8i* B, const int* sa, const int* sb, float* D) {
    const int lane = threadIdx.x;
    v16f c;
    for (int r = 0; r < 16; r++) c[r] = 0.0f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[lane], B[lane], c, 4, FB, 0, sa[lane], 0, sb[lane]);
    for (int r = 0; r < 16; r++) D[lane * 16 + r] = c[r];
}

static double fp4(int c) {
    static const double t[8] = {0, 0.5, 1, 1.5, 2, 3, 4, 6};
    return (c & 8 ? -1.0 : 1.0) * t[c & 7];
}
static double fp6(int c) {  // E2M3
    const int e = (c >> 3) & 3, m = c & 7;
    const double v = e == 0 ? m / 8.0 : (1.0 + m / 8.0) * (double)(1 << (e - 1));
    return (c & 32 ? -1.0 : 1.0) * v;
}

int main() {
    srand(7);
    int bad_total = 0;
    for (int fb = 0; fb < 2; fb++) {  // B format: FP6, then FP4
        std::vector<int> a(32 * 64), b(32 * 64), sea(64), seb(64);
        for (auto& v : a) v = rand() & 15;
        for (auto& v : b) v = fb == 0 ? (rand() & 63) : (rand() & 15);
        std::vector<v8i> A(64), B(64);
        std::vector<int> SA(64), SB(64);
        for (int lane = 0; lane < 64; lane++) {
            const int i = lane & 31, kb = lane >> 5;
            uint32_t wa[8] = {0}, wb[8] = {0};
            for (int e = 0; e < 32; e++) {
                const int k = 32 * kb + e;
                wa[e / 8] |= (uint32_t)a[i * 64 + k] << (4 * (e % 8));
                if (fb == 0) {
                    const uint64_t bit = 6 * e;
                    const uint64_t v = (uint64_t)b[i * 64 + k] << (bit % 32);
                    wb[bit / 32] |= (uint32_t)v;
                    if (bit % 32 > 26) wb[bit / 32 + 1] |= (uint32_t)(v >> 32);
                } else {
                    wb[e / 8] |= (uint32_t)b[i * 64 + k] << (4 * (e % 8));
                }
            }
            for (int q = 0; q < 8; q++) A[lane][q] = (int)wa[q], B[lane][q] = (int)wb[q];
            sea[lane] = 120 + rand() % 12;  // 2^(e - 127)
            seb[lane] = 124 + rand() % 8;
            SA[lane] = sea[lane] | (0x55 << 8) | (0x11 << 16) | (0x33 << 24);  // only byte 0 must count
            SB[lane] = seb[lane] | (0x22 << 8) | (0x44 << 16) | (0x66 << 24);
        }
        v8i *dA, *dB;
        int *dsa, *dsb;
        float* dD;
        hipMalloc(&dA, 64 * sizeof(v8i));
        hipMalloc(&dB, 64 * sizeof(v8i));
        hipMalloc(&dsa, 256);
        hipMalloc(&dsb, 256);
        hipMalloc(&dD, 64 * 16 * 4);
        hipMemcpy(dA, A.data(), 64 * sizeof(v8i), hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), 64 * sizeof(v8i), hipMemcpyHostToDevice);
        hipMemcpy(dsa, SA.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(dsb, SB.data(), 256, hipMemcpyHostToDevice);
        if (fb == 0)
            hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
        else
            hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
        std::vector<float> D(64 * 16);
        hipMemcpy(D.data(), dD, 64 * 16 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int lane = 0; lane < 64; lane++)
            for (int r = 0; r < 16; r++) {
                const int j = lane & 31, h = lane >> 5, i = (r & 3) + 8 * (r >> 2) + 4 * h;
                double s = 0;
                for (int k2 = 0; k2 < 64; k2++) {
                    const int kb = k2 >> 5;
                    const double va = fp4(a[i * 64 + k2]) * std::ldexp(1.0, sea[i + 32 * kb] - 127);
                    const double vb = (fb == 0 ? fp6(b[j * 64 + k2]) : fp4(b[j * 64 + k2])) * std::ldexp(1.0, seb[j + 32 * kb] - 127);
                    s += va * vb;
                }
                if ((float)s != D[lane * 16 + r]) {
                    if (bad < 5) printf("  B %s lane %d reg %d: got %g expected %g\n", fb == 0 ? "FP6" : "FP4", lane, r, D[lane * 16 + r], s);
                    bad++;
                }
            }
        printf("32x32x64 FP4 x %s: %d mismatches of 1024\n", fb == 0 ? "FP6" : "FP4", bad);
        bad_total += bad;
    }
    return bad_total ? 1 : 0;
}
