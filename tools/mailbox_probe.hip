// tools/mailbox_probe.hip -- where should the session's mailbox live? (round 5)
// A one-wavefront resident kernel polls a command word with a system-scope load and answers with a system-scope store into pinned
// host memory; the host measures the round trip of `n` commands, the kernel the latency of one poll (s_memrealtime, 10 ns ticks).
// The command word lives in  (1) pinned host memory (hipHostMallocCoherent: what tinympc_session.hip does today -- every poll is
// a PCIe read), (2) fine-grained device memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained)) written by the host through the
// PCIe BAR, (3) ordinary hipMalloc memory written the same way. (2)/(3) need the device's memory mapped into the host's address
// space (large BAR); a host store into an unmapped pointer faults, so the first store of each variant is guarded (SIGSEGV -> longjmp).
//   hipcc --offload-arch=gfx950 -O2 tools/mailbox_probe.hip -o tools/bin/mailbox_probe && tools/bin/mailbox_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <vector>
#include <immintrin.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(64) k_ping(const double *mail, double *answer, int n, unsigned long long idle_ticks, unsigned long long *poll_stats) {
    unsigned long long polls = 0, poll_ticks = 0;
    for (int seq = 1; seq <= n; ++seq) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            const unsigned long long a = __builtin_amdgcn_s_memrealtime();
            const double v = __hip_atomic_load(mail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned long long b = __builtin_amdgcn_s_memrealtime();
            polls += 1; poll_ticks += b - a;
            if (v == (double)seq) break;
            if (b - t0 > idle_ticks) { seq = n + 1; break; }  // nobody talks to us: leave
        }
        if (seq <= n && threadIdx.x == 0) __hip_atomic_store(answer, (double)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 0) { poll_stats[0] = polls; poll_stats[1] = poll_ticks; }
}

static sigjmp_buf g_jmp;
static void on_segv(int) { siglongjmp(g_jmp, 1); }

static bool host_can_write(volatile double *p) {
    struct sigaction sa{}, old_segv{}, old_bus{};
    sa.sa_handler = on_segv;
    sigemptyset(&sa.sa_mask);
    sigaction(SIGSEGV, &sa, &old_segv);
    sigaction(SIGBUS, &sa, &old_bus);
    bool ok = false;
    if (sigsetjmp(g_jmp, 1) == 0) {
        p[0] = 0.0;
        _mm_sfence();
        ok = true;
    }
    sigaction(SIGSEGV, &old_segv, nullptr);
    sigaction(SIGBUS, &old_bus, nullptr);
    return ok;
}

int main() {
    int dev = 0, large_bar = -1;
    CHECK(hipSetDevice(dev));
    (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev);
    printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    double *answer = nullptr;
    CHECK(hipHostMalloc((void **)&answer, 64, hipHostMallocCoherent));
    unsigned long long *stats = nullptr;
    CHECK(hipHostMalloc((void **)&stats, 64, hipHostMallocCoherent));
    const int n = 20000;
    struct Variant { const char *name; double *mail; bool device; };
    std::vector<Variant> vs;
    double *m_host = nullptr, *m_fine = nullptr, *m_coarse = nullptr, *m_unc = nullptr;
    CHECK(hipHostMalloc((void **)&m_host, 4096, hipHostMallocCoherent));
    vs.push_back({"pinned host memory (coherent): GPU polls across PCIe", m_host, false});
    if (hipExtMallocWithFlags((void **)&m_fine, 4096, hipDeviceMallocFinegrained) == hipSuccess) vs.push_back({"fine-grained device memory, host writes through the BAR", m_fine, true});
    else printf("hipExtMallocWithFlags(hipDeviceMallocFinegrained) refused: %s\n", hipGetErrorString(hipGetLastError()));
    if (hipExtMallocWithFlags((void **)&m_unc, 4096, hipDeviceMallocUncached) == hipSuccess) vs.push_back({"uncached device memory, host writes through the BAR", m_unc, true});
    else printf("hipExtMallocWithFlags(hipDeviceMallocUncached) refused: %s\n", hipGetErrorString(hipGetLastError()));
    if (hipMalloc((void **)&m_coarse, 4096) == hipSuccess) vs.push_back({"ordinary hipMalloc memory, host writes through the BAR", m_coarse, true});
    for (auto &v : vs) {
        if (v.device) {
            CHECK(hipMemset(v.mail, 0, 4096));
            CHECK(hipDeviceSynchronize());
            if (!host_can_write(v.mail)) { printf("%-62s host store FAULTS: not mapped into the host's address space\n", v.name); continue; }
        } else {
            std::memset(v.mail, 0, 4096);
        }
        answer[0] = 0.0;
        stats[0] = stats[1] = 0;
        hipLaunchKernelGGL(k_ping, dim3(1), dim3(64), 0, st, v.mail, answer, n, 200000000ull /* 2 s */, stats);
        CHECK(hipGetLastError());
        std::vector<double> rtt(n);
        volatile double *mail = v.mail;
        volatile double *ans = answer;
        bool lost = false;
        for (int seq = 1; seq <= n && !lost; ++seq) {
            const auto t0 = std::chrono::steady_clock::now();
            mail[0] = (double)seq;
            _mm_sfence();
            long spin = 0;
            while (ans[0] != (double)seq) {
                __builtin_ia32_pause();
                if (++spin > 400000000L) { lost = true; break; }
            }
            rtt[seq - 1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        }
        CHECK(hipStreamSynchronize(st));
        if (lost) { printf("%-62s the kernel never saw the host's store (stale cache line?)\n", v.name); continue; }
        std::vector<double> s(rtt.begin() + 100, rtt.end());
        std::sort(s.begin(), s.end());
        double mean = 0; for (double x : s) mean += x; mean /= s.size();
        printf("%-62s round trip median %.2f us  mean %.2f  p90 %.2f  min %.2f | one poll %.0f ns (%llu polls)\n", v.name, s[s.size() / 2], mean,
               s[(size_t)(0.9 * s.size())], s[0], 10.0 * (double)stats[1] / (double)stats[0], stats[0]);
    }
    return 0;
}
