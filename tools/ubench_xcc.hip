// Which XCD does workgroup i of a dispatch run on, alone and with other dispatches in flight on other
// streams?  The flash kernel places workgroups that stream the same K/V image on ids equal mod 8 on the
// assumption that workgroups i and i + 8 of a dispatch go to the same XCD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_xcc.hip -o exp/ubench_xcc && exp/ubench_xcc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) probe(unsigned* out, int spin) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;      // keep the workgroup resident for a while
    if (threadIdx.x == 0) out[blockIdx.x] = (x & 15u) | (a == 1.2345f ? 16u : 0u);
}

int main() {
    const int grid = 384, reps = 50, spin = 20000;
    for (int S : {1, 2, 3}) {
        std::vector<hipStream_t> st(S);
        std::vector<unsigned*> out(S);
        for (int s = 0; s < S; ++s) { (void)hipStreamCreate(&st[s]); (void)hipMalloc(&out[s], grid * sizeof(unsigned)); }
        long match = 0, total = 0;
        int hist[16] = {0};
        for (int r = 0; r < reps; ++r) {
            for (int s = 0; s < S; ++s) probe<<<grid, 256, 0, st[s]>>>(out[s], spin);
            (void)hipDeviceSynchronize();
            for (int s = 0; s < S; ++s) {
                std::vector<unsigned> h(grid);
                (void)hipMemcpy(h.data(), out[s], grid * sizeof(unsigned), hipMemcpyDeviceToHost);
                int off[8] = {0};
                for (int i = 0; i < grid; ++i) { ++off[((h[i] & 15u) + 8 - (i & 7)) & 7]; ++total; ++hist[h[i] & 15u]; }
                int best = 0;
                for (int x = 0; x < 8; ++x) best = off[x] > best ? off[x] : best;
                match += best;      // workgroups on XCD (i + k) % 8 for the dispatch's most common k
            }
        }
        printf("%d stream(s): workgroup i on XCD (i + k) %% 8, k constant within a dispatch, for %.1f %% of workgroups; per-XCD share:", S, 100.0 * match / total);
        for (int x = 0; x < 8; ++x) printf(" %.3f", (double)hist[x] / total);
        printf("\n");
    }
    return 0;
}
