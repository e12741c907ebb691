// Development probe for the "cluster-split recurrence" experiment: k workgroups share one 16-row batch
// tile, each holds 1/k of the recurrent weights in registers and must see the OTHER k-1 workgroups' share of h_t before step
// t+1 - an all-to-all of 8/k KB per workgroup per time step, through the XCD's L2.  This measures that exchange alone:
// every thread publishes 8-byte {payload, tag = step} granules with ONE sc1 store each (the cheapest correct hand-off of
// MI355X_MICROARCH.md's price list: no separate flag, no fence) and polls its partners' granules with sc1 loads until the tag
// matches.  Kill criterion: <= 0.6 us per step, else the split cannot beat the 2.1 us (forward) / 2.97 us (backward) step of the
// one-CU-per-tile kernels by enough to pay for k times the CUs.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/cluster_exchange_probe tools/probes/cluster_exchange_probe.hip && /tmp/cluster_exchange_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

// grid = groups * k workgroups of 256 threads.  member m of group g is block (same_xcd ? g % 8 + 8 * (m + k * (g / 8)) : g * k + m)
// buf: [groups][2 (step parity)][k][gran] granules of uint2 {payload, tag}
__global__ __launch_bounds__(256) void exchange_k(uint2* buf, int k, int gran_per_wg, int steps, int same_xcd, int work_cycles,
                                                  uint32_t* xcc_out, uint64_t* cyc_out, uint32_t* err_out) {
    int g, m;
    if (same_xcd) { const int slot = blockIdx.x / 8; g = (slot / k) * 8 + blockIdx.x % 8; m = slot % k; }
    else { g = blockIdx.x / k; m = blockIdx.x % k; }
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) xcc_out[blockIdx.x] = (xcc & 0xf) | (g << 8) | (m << 24);
    uint2* base = buf + (size_t)g * 2 * k * gran_per_wg;
    const int per_thread = gran_per_wg / 256;                       // granules this thread publishes per step
    uint32_t errors = 0;
    const long long t0 = clock64();
    for (int t = 1; t <= steps; ++t) {
        uint2* slot = base + (size_t)(t & 1) * k * gran_per_wg;
        // "compute": the share of the step's MFMA / gate work (a busy wait that touches no memory)
        if (work_cycles) { const long long w0 = clock64(); while (clock64() - w0 < work_cycles) {} }
        for (int i = 0; i < per_thread; ++i) {
            const int e = i * 256 + threadIdx.x;
            const uint2 v = make_uint2((uint32_t)(t * 131 + m * 7 + e), (uint32_t)t);
            asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(&slot[m * gran_per_wg + e]), "v"(v) : "memory");
        }
        for (int p = 1; p < k; ++p) {
            const int src = (m + p) % k;
            for (int i = 0; i < per_thread; ++i) {
                const int e = i * 256 + threadIdx.x;
                uint2 v;
                int spins = 0;
                do {
                    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(&slot[src * gran_per_wg + e]) : "memory");
                } while (v.y != (uint32_t)t && ++spins < (1 << 22));
                if (v.y != (uint32_t)t || v.x != (uint32_t)(t * 131 + src * 7 + e)) ++errors;
            }
        }
        __syncthreads();          // (the real kernel needs every wave's partners' h in LDS before its MFMAs)
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc_out[blockIdx.x] = (uint64_t)(t1 - t0);
    if (errors) atomicAdd(err_out, errors);
}

int main() {
    const int steps = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("all-to-all of one h tile between k workgroups per time step (tagged 8-byte granules, sc1 stores / sc1 polling loads)\n");
    printf("%-10s %-3s %-7s %-9s %-10s %-12s %-12s %s\n", "placement", "k", "groups", "KB/WG", "work(cyc)", "us/step", "cyc/step", "errors / XCDs per group");
    for (int same = 1; same >= 0; --same)
        for (int k : {2, 4})
            for (int groups : {1, 16, 32})
                for (int work : {0, 1200}) {
                    const int payload_bytes = 16 * 256 * 2 / k;           // this workgroup's share of h_t (16 rows x 256 units, bf16)
                    const int gran = ((payload_bytes / 4 + 255) / 256) * 256;   // 4 payload bytes per granule
                    const int nb = groups * k;
                    if (same && (groups % 8) && groups != 1) continue;
                    uint2* buf; hipMalloc(&buf, (size_t)groups * 2 * k * gran * sizeof(uint2)); hipMemset(buf, 0, (size_t)groups * 2 * k * gran * sizeof(uint2));
                    uint32_t *xcc, *err; uint64_t* cyc;
                    hipMalloc(&xcc, nb * 4); hipMalloc(&err, 4); hipMalloc(&cyc, nb * 8); hipMemset(err, 0, 4);
                    const int grid = (same && groups == 1) ? 8 * k : nb;     // groups == 1, same XCD: launch 8k blocks, only XCD 0's group... keep it simple: all 8 groups run
                    const int eff_groups = (same && groups == 1) ? 8 : groups;
                    if (same && groups == 1) { hipFree(buf); hipMalloc(&buf, (size_t)8 * 2 * k * gran * sizeof(uint2)); hipMemset(buf, 0, (size_t)8 * 2 * k * gran * sizeof(uint2));
                                               hipFree(xcc); hipFree(cyc); hipMalloc(&xcc, grid * 4); hipMalloc(&cyc, grid * 8); }
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(exchange_k, dim3(grid), dim3(256), 0, 0, buf, k, gran, steps, same, work, xcc, cyc, err);
                    hipEventRecord(e1);
                    hipError_t rc = hipDeviceSynchronize();
                    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                    std::vector<uint32_t> hx(grid); std::vector<uint64_t> hc(grid); uint32_t he = 0;
                    hipMemcpy(hx.data(), xcc, grid * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
                    hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost);
                    int mixed = 0;
                    for (int a = 0; a < grid; ++a) for (int b = 0; b < grid; ++b)
                        if (((hx[a] >> 8) & 0xffff) == ((hx[b] >> 8) & 0xffff) && (hx[a] & 0xf) != (hx[b] & 0xf)) { mixed = 1; }
                    uint64_t mx = 0; for (auto c : hc) mx = c > mx ? c : mx;
                    printf("%-10s %-3d %-7d %-9.1f %-10d %-12.3f %-12.0f %u / %s  (%s)\n", same ? "same-XCD" : "cross-XCD", k, eff_groups, payload_bytes / 1024.0, work,
                           ms * 1e3 / steps, (double)mx / steps, he, mixed ? "groups span XCDs" : "one XCD per group", hipGetErrorString(rc));
                    hipFree(buf); hipFree(xcc); hipFree(err); hipFree(cyc);
                }
    return 0;
}
