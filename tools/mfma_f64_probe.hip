// Probe of the v_mfma_f64_16x16x4_f64 register layout on gfx950 (debug helper, not part of the library).
// Prints, for a few (lane_a, lane_b) one-hot inputs, which (lane, vgpr) of D receives the product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ void probe(int la, int lb, double* out) {
    const int l = threadIdx.x;
    double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
    v4f64 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
}
int main() {
    double* d; hipMalloc(&d, 256 * 8);
    double h[256];
    int probes[][2] = {{5, 7}, {5, 23}, {21, 7}, {21, 23}, {37, 39}, {0, 0}, {15, 15}, {16, 16}, {63, 63}, {2, 50}, {50, 50}};
    for (auto& p : probes) {
        probe<<<1, 64>>>(p[0], p[1], d);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("la=%2d lb=%2d ->", p[0], p[1]);
        for (int i = 0; i < 256; ++i) if (h[i] != 0) printf(" (lane %d, v %d)=%g", i / 4, i % 4, h[i]);
        printf("\n");
    }
    return 0;
}
