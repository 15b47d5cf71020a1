// Where do co-resident workgroups land?  Every wave of a 512-workgroup launch (256 threads, 66 KB LDS: two per CU) records
// HW_REG_HW_ID, HW_REG_XCC_ID and the clock at start / end.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <map>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
    extern __shared__ float lds[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);     // HW_REG_XCC_ID
    const uint64_t t0 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < spin; ++i) { lds[threadIdx.x] = s; __syncthreads(); s += lds[(threadIdx.x + 1) & 255]; }
    const uint64_t t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) {
        unsigned* o = out + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 6;
        o[0] = hw; o[1] = xcc; o[2] = (unsigned)t0; o[3] = (unsigned)(t0 >> 32); o[4] = (unsigned)t1; o[5] = (unsigned)(s == 1.5f);
    }
}
int main() {
    const int grid = 1024;
    unsigned* d;
    hipMalloc(&d, (size_t)grid * 4 * 6 * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 66368);
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 66368, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h((size_t)grid * 4 * 6);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    for (int b = 0; b < 24; ++b) {
        printf("wg %4d:", b);
        for (int w = 0; w < 4; ++w) {
            const unsigned hw = h[(b * 4 + w) * 6], xcc = h[(b * 4 + w) * 6 + 1];
            printf("  [hw %08x wave %2u simd %u pipe %u cu %2u sh %u se %u xcc %u]", hw, hw & 15, (hw >> 4) & 3, (hw >> 6) & 3, (hw >> 8) & 15, (hw >> 12) & 1,
                   (hw >> 13) & 7, xcc & 15);
        }
        printf("\n");
    }
    // co-residency: group wave 0 of every workgroup by (xcc, se, sh, cu)
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < grid; ++b) {
        const unsigned hw = h[(b * 4) * 6], xcc = h[(b * 4) * 6 + 1] & 15;
        cu[(xcc << 16) | (hw & 0xff00)].push_back(b);
    }
    printf("distinct (xcc, se, sh, cu): %zu\n", cu.size());
    int shown = 0;
    for (auto& kv : cu) {
        if (shown++ >= 12) break;
        printf("  key %06x:", kv.first);
        for (int b : kv.second) printf(" %d(w%u,t0=%u)", b, h[(b * 4) * 6] & 15, h[(b * 4) * 6 + 2]);
        printf("\n");
    }
    return 0;
}
