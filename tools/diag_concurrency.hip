// Diagnostic: do kernels of two HIP streams overlap on this box -- and do they still when they need scratch memory?
// Each kernel: 64 blocks x 64 threads spinning ~2 ms.  SCR = private bytes per lane (dynamically indexed array).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
template <int SCR>
__global__ void spin(long long cycles, int* out, int k) {
    volatile int a[SCR / 4 + 1];
    for (int i = 0; i <= SCR / 4; ++i) a[i] = i * k;
    long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) { }
    if (out && threadIdx.x == 0) out[blockIdx.x] = a[(k * 7) % (SCR / 4 + 1)];
}
template <int SCR>
void run(hipStream_t* s, int* d) {
    const long long cyc = 200000;   // 100 MHz wall clock: 2 ms
    for (int n = 1; n <= 4; n *= 2) {
        (void)hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < 5; ++rep) for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin<SCR>, dim3(64), dim3(64), 0, s[i], cyc, d, rep + 1);
        (void)hipDeviceSynchronize();
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("scratch %4d B/lane: %d streams x 5 kernels of 2 ms: %.2f ms (serial would be %.0f)\n", SCR, n, ms, 10.0 * n);
    }
}
int main() {
    int* d; (void)hipMalloc(&d, 4096);
    hipStream_t s[4]; for (int i = 0; i < 4; ++i) (void)hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    run<0>(s, d); run<64>(s, d); run<256>(s, d); run<320>(s, d); run<512>(s, d); run<700>(s, d); run<2048>(s, d);
    return 0;
}
