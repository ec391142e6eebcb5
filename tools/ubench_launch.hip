// What one tiny launch costs on this box: the floor under msbwt_rle_count_kmer (one query per call).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_launch.hip -o tools/ubench_launch && tools/ubench_launch
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>

__global__ void k_empty() {}
__global__ void k_flag(volatile uint64_t *mail, uint64_t seq) {
    if (threadIdx.x == 0) {
        mail[1] = mail[0] + 1;  // read the "query", write the "count"
        __threadfence_system();
        mail[2] = seq;
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    uint64_t *mail = nullptr, *dmail = nullptr;
    hipHostMalloc(reinterpret_cast<void **>(&mail), 4096, hipHostMallocMapped);
    hipHostGetDevicePointer(reinterpret_cast<void **>(&dmail), mail, 0);
    mail[0] = mail[1] = mail[2] = 0;
    const int n = 20000;
    for (int i = 0; i < 200; ++i) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); hipStreamSynchronize(s); }
    double t0 = now_us();
    for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); hipStreamSynchronize(s); }
    printf("empty kernel + hipStreamSynchronize:        %6.2f us\n", (now_us() - t0) / n);
    t0 = now_us();
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
    double t1 = now_us();
    hipStreamSynchronize(s);
    printf("launch call alone (asynchronous):           %6.2f us\n", (t1 - t0) / n);
    t0 = now_us();
    for (int i = 1; i <= n; ++i) {
        mail[0] = uint64_t(i);
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, dmail, uint64_t(i));
        hipStreamSynchronize(s);
        if (mail[1] != uint64_t(i) + 1) { printf("bad\n"); return 1; }
    }
    printf("mapped-host query/result + synchronize:     %6.2f us\n", (now_us() - t0) / n);
    t0 = now_us();
    for (int i = 1; i <= n; ++i) {
        mail[0] = uint64_t(i);
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, dmail, uint64_t(n + i));
        while (reinterpret_cast<volatile uint64_t *>(mail)[2] != uint64_t(n + i)) {}
        if (mail[1] != uint64_t(i) + 1) { printf("bad\n"); return 1; }
    }
    printf("mapped-host query/result + spin on a flag:  %6.2f us\n", (now_us() - t0) / n);
    hipStreamSynchronize(s);
    hipEvent_t e;
    hipEventCreateWithFlags(&e, hipEventDisableTiming);
    t0 = now_us();
    for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); hipEventRecord(e, s); hipEventSynchronize(e); }
    printf("empty kernel + event record + event sync:   %6.2f us\n", (now_us() - t0) / n);
    return 0;
}
