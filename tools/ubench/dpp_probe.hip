// Probes DPP controls on gfx950: prints, for each control, the source lane every lane reads.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int ROW_MASK, int BANK_MASK>
__global__ void probe(int* out) {
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(-1, lane, CTRL, ROW_MASK, BANK_MASK, false);
}
template <int CTRL, int ROW_MASK, int BANK_MASK>
void run(const char* name, int* d) {
    int h[64];
    probe<CTRL, ROW_MASK, BANK_MASK><<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%-34s:", name);
    for (int i = 0; i < 32; ++i) printf(" %2d", h[i]);
    printf(" ...\n");
}
int main() {
    int* d; hipMalloc(&d, 256);
    run<0x104, 0xf, 0xf>("row_shl:4 (0x104)", d);
    run<0x114, 0xf, 0xf>("row_shr:4 (0x114)", d);
    run<0x124, 0xf, 0xf>("row_ror:4 (0x124)", d);
    run<0x128, 0xf, 0xf>("row_ror:8 (0x128)", d);
    run<0x12C, 0xf, 0xf>("row_ror:12 (0x12C)", d);
    run<0x104, 0xf, 0x5>("row_shl:4 bank_mask 0x5", d);
    run<0x114, 0xf, 0xa>("row_shr:4 bank_mask 0xa", d);
    run<0x141, 0xf, 0xf>("row_half_mirror", d);
    run<0x140, 0xf, 0xf>("row_mirror", d);
    run<0x142, 0xa, 0xf>("row_bcast15 row_mask 0xa", d);
    run<0x143, 0xc, 0xf>("row_bcast31 row_mask 0xc", d);
    return 0;
}
