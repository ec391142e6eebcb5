// Where does global_load_lds_dwordx4 put the data of each lane when some lanes are masked off?
// (hardware probe, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void global_void;

__global__ __launch_bounds__(64) void probe(const uint4 *src, uint64_t mask, uint4 *out, int use_zero_addr) {
    __shared__ uint4 lds[128];
    const uint32_t lane = threadIdx.x;
    lds[lane] = make_uint4(0xEEEEEEEEu, 0, 0, 0);
    lds[lane + 64] = make_uint4(0xDDDDDDDDu, 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool on = (mask >> lane) & 1ull;
    uint64_t a = on ? reinterpret_cast<uint64_t>(src + lane) : 0ull;
    if (use_zero_addr == 0) {
        if (a != 0) __builtin_amdgcn_global_load_lds((global_void *)a, (lds_void *)&lds[8], 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    out[lane] = lds[lane];
    out[lane + 64] = lds[lane + 64];
}

int main() {
    uint4 *src, *out;
    CK(hipMalloc(&src, 64 * 16));
    CK(hipMalloc(&out, 128 * 16));
    std::vector<uint4> h(64);
    for (uint32_t i = 0; i < 64; ++i) h[i] = make_uint4(0x1000 + i, i, i, i);
    CK(hipMemcpy(src, h.data(), 64 * 16, hipMemcpyHostToDevice));
    const uint64_t masks[] = {~0ull, 0x5555555555555555ull, 0x000000000000FF00ull, 0xFF00000000000000ull, 0x00FF00FF00FF00F0ull, 0xFFFFFFFFFFFFFFFEull};
    for (uint64_t m : masks) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, m, out, 0);
        CK(hipDeviceSynchronize());
        std::vector<uint4> r(128);
        CK(hipMemcpy(r.data(), out, 128 * 16, hipMemcpyDeviceToHost));
        int ok = 1, landed = 0;
        for (uint32_t i = 0; i < 128; ++i) {
            const bool is_data = (r[i].x & 0xFFFFF000u) == 0x1000u;
            if (is_data) {
                ++landed;
                const uint32_t from = r[i].x - 0x1000;
                if (i != 8 + from) { ok = 0; printf("  mask %016llx: data of lane %u landed at slot %u (expected %u)\n", (unsigned long long)m, from, i, 8 + from); }
            }
        }
        for (uint32_t l = 0; l < 64; ++l) if (((m >> l) & 1) && (r[8 + l].x != 0x1000 + l)) { ok = 0; printf("  mask %016llx: lane %u's slot holds %08x\n", (unsigned long long)m, l, r[8 + l].x); }
        printf("mask %016llx: %d pieces landed, %s\n", (unsigned long long)m, landed, ok ? "all at base + lane*16" : "MISPLACED");
    }
    return 0;
}
