// Development probe: which CUs does a hipExtStreamCreateWithCUMask stream use?  Each workgroup records XCC_ID and HW_ID.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/cumask_probe tools/probes/cumask_probe.hip && /tmp/cumask_probe
// Result on MI355X: mask bit i = CU i/8 of XCD i%8 (CUs of an XCD taken round-robin over its 4 shader engines); a mask
// that empties an XCD is ignored.  Running the gradient GEMMs on streams masked to 24 of 32 CUs per XCD (to keep whole CUs
// free for the recurrent kernels) made the training step 1.7x SLOWER - kernels on masked queues serialise against the rest.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <vector>
__global__ void who(uint32_t* out, int spin) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = xcc; out[blockIdx.x * 2 + 1] = hw; }
}
static void run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
    const int N = 2048;
    uint32_t* d; hipMalloc(&d, N * 8); hipMemset(d, 0xff, N * 8);
    hipLaunchKernelGGL(who, dim3(N), dim3(256), 65536, s, d, 20000);
    hipStreamSynchronize(s);
    std::vector<uint32_t> h(N * 2); hipMemcpy(h.data(), d, N * 8, hipMemcpyDeviceToHost);
    std::map<int, std::set<uint32_t>> per;
    for (int i = 0; i < N; ++i) per[h[2 * i] & 0xf].insert((h[2 * i + 1] >> 8) & 0xff);   // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    printf("%-28s", name);
    int tot = 0;
    for (auto& kv : per) { printf(" xcc%d:%zu", kv.first, kv.second.size()); tot += kv.second.size(); }
    printf("  total %d;  xcc0 CUs:", tot);
    for (uint32_t c : per[0]) printf(" %02x", c);
    printf("\n");
    hipFree(d); hipStreamDestroy(s);
}
int main() {
    std::vector<uint32_t> m(8, 0);
    auto set = [&](int b) { m[b / 32] |= 1u << (b % 32); };
    m.assign(8, 0xffffffffu); run("all 256 bits", m);
    m.assign(8, 0); for (int b = 0; b < 32; ++b) set(b); run("bits 0..31", m);
    m.assign(8, 0); for (int b = 0; b < 64; ++b) set(b); run("bits 0..63", m);
    m.assign(8, 0); for (int b = 0; b < 256; b += 8) set(b); run("every 8th bit", m);
    m.assign(8, 0); for (int b = 0; b < 8; ++b) set(b); run("bits 0..7", m);
    m.assign(8, 0); for (int b = 64; b < 256; ++b) set(b); run("bits 64..255", m);
    m.assign(8, 0); for (int b = 0; b < 256; ++b) if ((b / 8) % 4 != 0) set(b); run("all but b/8%4==0", m);
    return 0;
}
