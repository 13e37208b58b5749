// Measurement aid (not product): which compute units a HIP stream created with hipExtStreamCreateWithCUMask may use, i.e. how
// the bits of the mask map onto (XCC, SE, SH, CU) on this device.  Every mask tried in the first part keeps at least 24 CUs in
// every XCC under ANY bit order, so no workgroup can be dealt to an XCC the queue has no unit on; the last part (one unit per
// XCC: bits 0..7) only runs when the first part has shown those eight bits to lie on eight different XCCs.
//   tools/cumask_probe            (GPU box; prints one line per mask)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void where_kernel(uint32_t* out, int spin) {
    const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID[3:0]
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);                // keep the unit busy so that the grid spreads
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | ((hw >> 8) & 0xffu);  // cu_id[3:0] | sh_id << 4 | se_id << 5
}

static std::set<uint32_t> units(hipStream_t s, uint32_t* d_out, int blocks, int spin) {
    std::vector<uint32_t> h(blocks);
    hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(256), 0, s, d_out, spin);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d_out, blocks * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return std::set<uint32_t>(h.begin(), h.end());
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
    printf("device: %s, %d compute units, mask words %d\n", prop.gcnArchName, ncu, words);
    const int blocks = 16384, spin = 40;
    uint32_t* d_out;
    CK(hipMalloc(&d_out, blocks * sizeof(uint32_t)));
    hipStream_t full;
    CK(hipStreamCreate(&full));
    const std::set<uint32_t> all = units(full, d_out, blocks, spin);
    printf("unmasked stream: %zu distinct (xcc, se, sh, cu)\n", all.size());
    auto masked = [&](const std::vector<int>& off, const char* what) {
        std::vector<uint32_t> m(words, 0xffffffffu);
        for (int b : off) m[b / 32] &= ~(1u << (b % 32));
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, words, m.data()));
        const std::set<uint32_t> got = units(s, d_out, blocks, spin);
        printf("%-28s %3zu units; missing:", what, got.size());
        std::set<int> xccs;
        for (uint32_t u : all) if (!got.count(u)) { printf(" (xcc %u se %u sh %u cu %u)", u >> 16, (u >> 5) & 7, (u >> 4) & 1, u & 15); xccs.insert(u >> 16); }
        for (uint32_t u : got) if (!all.count(u)) printf(" [extra %x]", u);
        printf("\n");
        CK(hipStreamDestroy(s));
        return xccs;
    };
    char name[64];
    for (int b : {0, 1, 2, 7, 8, 9, 16, 31, 32, 64, 255}) {
        if (b >= ncu) continue;
        snprintf(name, sizeof name, "all but bit %d", b);
        masked({b}, name);
    }
    const std::set<int> x8 = masked({0, 1, 2, 3, 4, 5, 6, 7}, "all but bits 0..7");
    if (x8.size() == 8 && ncu == 256) {
        // one unit in each of the 8 XCCs: wherever a workgroup is dealt, the queue has a unit there
        std::vector<uint32_t> m(words, 0u);
        m[0] = 0xffu;
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, words, m.data()));
        const std::set<uint32_t> got = units(s, d_out, 512, 4);
        printf("only bits 0..7: %zu units:", got.size());
        for (uint32_t u : got) printf(" (xcc %u se %u sh %u cu %u)", u >> 16, (u >> 5) & 7, (u >> 4) & 1, u & 15);
        printf("\n");
        // a one-workgroup kernel, many times: it must always find its unit
        std::set<uint32_t> one;
        for (int i = 0; i < 64; ++i) { auto g = units(s, d_out, 1, 1); one.insert(g.begin(), g.end()); }
        printf("64 one-workgroup launches on it ran on %zu distinct units:", one.size());
        for (uint32_t u : one) printf(" (xcc %u cu %u)", u >> 16, u & 15);
        printf("\n");
        CK(hipStreamDestroy(s));
    } else {
        printf("bits 0..7 are NOT on eight different XCCs (%zu): the one-unit-per-XCC part is skipped\n", x8.size());
    }
    return 0;
}
