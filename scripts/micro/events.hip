// Micro-benchmark: what the things between two dependent kernel launches cost a stream (tuning aid, not product code).
// Every case runs N rounds of [short kernel A on s1] -> dependency -> [short kernel B on s2 or s1]; prints us per round.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

__global__ void k_small(unsigned *p, unsigned v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = v; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    unsigned *d = nullptr;
    CK(hipMalloc(&d, 4096));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int N = 2000, R = 64;
    hipEvent_t ev[4][R];
    const unsigned flags[4] = {hipEventDefault, hipEventDisableTiming, hipEventReleaseToDevice, hipEventDisableTiming | hipEventReleaseToDevice};
    const char *fname[4] = {"default", "disable-timing", "release-to-device", "disable-timing + release-to-device"};
    for (int f = 0; f < 4; ++f) for (int r = 0; r < R; ++r) CK(hipEventCreateWithFlags(&ev[f][r], flags[f]));
    auto sync = [&]() { (void)hipStreamSynchronize(s1); (void)hipStreamSynchronize(s2); };
    for (int rep = 0; rep < 2; ++rep) {
        // (a) back-to-back kernels on one stream
        sync(); double t0 = now();
        for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s1, d, 1u); hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s1, d + 64, 2u); }
        sync(); printf("two kernels, one stream, nothing between            %7.2f us per round\n", (now() - t0) / N * 1e6);
        for (int f = 0; f < 4; ++f) {
            // (b) an event recorded between them (nobody waits for it)
            sync(); t0 = now();
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s1, d, 1u);
                CK(hipEventRecord(ev[f][i % R], s1));
                hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s1, d + 64, 2u);
            }
            sync(); printf("+ event record between (%-34s) %7.2f us per round\n", fname[f], (now() - t0) / N * 1e6);
            // (c) A on s1, record, s2 waits, B on s2; then s1 waits for B (a chain across two streams)
            sync(); t0 = now();
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s1, d, 1u);
                CK(hipEventRecord(ev[f][(2 * i) % R], s1));
                CK(hipStreamWaitEvent(s2, ev[f][(2 * i) % R], 0));
                hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s2, d + 64, 2u);
                CK(hipEventRecord(ev[f][(2 * i + 1) % R], s2));
                CK(hipStreamWaitEvent(s1, ev[f][(2 * i + 1) % R], 0));
            }
            sync(); printf("ping-pong across two streams (%-28s) %7.2f us per round\n", fname[f], (now() - t0) / N * 1e6);
            // (d) the same with the event attached to the kernel's own packet (hipExtLaunchKernelGGL stop event)
            sync(); t0 = now();
            for (int i = 0; i < N; ++i) {
                hipExtLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s1, nullptr, ev[f][(2 * i) % R], 0, d, 1u);
                CK(hipStreamWaitEvent(s2, ev[f][(2 * i) % R], 0));
                hipExtLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s2, nullptr, ev[f][(2 * i + 1) % R], 0, d + 64, 2u);
                CK(hipStreamWaitEvent(s1, ev[f][(2 * i + 1) % R], 0));
            }
            sync(); printf("ping-pong, stop event on the launch (%-22s) %7.2f us per round\n", fname[f], (now() - t0) / N * 1e6);
            // (e) one stream, stop event on the launch, nobody waits
            sync(); t0 = now();
            for (int i = 0; i < N; ++i) {
                hipExtLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s1, nullptr, ev[f][i % R], 0, d, 1u);
                hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s1, d + 64, 2u);
            }
            sync(); printf("one stream, stop event on the first launch (%-16s) %6.2f us per round\n", fname[f], (now() - t0) / N * 1e6);
        }
        // (g) the same chain with stream memory operations instead of events: write a counter behind A, the other stream waits for it
        {
            uint32_t *flags = nullptr, *flags2 = nullptr;           // (signal memory comes in 8-byte pieces)
            CK(hipExtMallocWithFlags(reinterpret_cast<void **>(&flags), 8, hipMallocSignalMemory));
            CK(hipExtMallocWithFlags(reinterpret_cast<void **>(&flags2), 8, hipMallocSignalMemory));
            CK(hipMemset(flags, 0, 8)); CK(hipMemset(flags2, 0, 8));
            sync(); t0 = now();
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s1, d, 1u);
                CK(hipStreamWriteValue32(s1, flags, (uint32_t)(i + 1), 0));
                CK(hipStreamWaitValue32(s2, flags, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xffffffffu));
                hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s2, d + 64, 2u);
                CK(hipStreamWriteValue32(s2, flags2, (uint32_t)(i + 1), 0));
                CK(hipStreamWaitValue32(s1, flags2, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xffffffffu));
            }
            sync(); printf("ping-pong across two streams, stream write / wait value %7.2f us per round\n", (now() - t0) / N * 1e6);
            (void)hipFree(flags); (void)hipFree(flags2);
        }
        // (f) any-order launch: B may start before A has finished (no barrier bit)
        sync(); t0 = now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s1, d, 1u);
            hipExtLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, d + 64, 2u);
        }
        sync(); printf("two kernels, one stream, second one any-order       %7.2f us per round\n", (now() - t0) / N * 1e6);
    }
    return 0;
}
