// exhaustive.hip -- TEST TOOL.  The device's short forms of the correctly rounded reciprocal, 1 / sqrt and Smith term
// 2 / (1 + sqrt(1 + t)) (srt_core.h: rcp3_rn and rcp_sqrt_rn, which fall back to the plain operators when the short
// form's acceptance test fails; smith_term on its domain, t >= +0 or NaN) on EVERY one of the 2^32 float bit patterns,
// against
//   (1) the compiler's general IEEE sequences on the device (-fhip-fp32-correctly-rounded-divide-sqrt), bit for bit
//       (two NaNs count as equal), and
//   (2) the host's IEEE operations, through a checksum over all 2^32 results (sum of bits(result) * (2 i + 1) mod 2^64,
//       NaNs canonicalised): a single differing result changes it.
// Prints one line per function ("<name> mismatches=<n> checksum_device=<hex> checksum_host=<hex>") and
// "exhaustive_failures=<n>"; exit code 1 when a mismatch count is non-zero or a checksum pair differs.  `--no-host` skips (2).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../simple_raytracer_amd/csrc/srt_core.h"

using namespace srt;

enum { F_RCP, F_RCP_SQRT, F_SMITH, F_COUNT };
static const char* kNames[F_COUNT] = {"rcp3_rn", "rcp_sqrt_rn", "smith_term"};

struct Report { unsigned long long bad, checksum; uint32_t first[4]; };

SRT_HD uint32_t canonical(float f) { return f != f ? 0x7FC00000u : fbits(f); }
// the Smith term's argument is a number in [+0, +infinity] or a NaN
SRT_HD bool in_smith_domain(float t) { return t != t || (fbits(t) >> 31) == 0u; }
SRT_HD float shipped(int which, float x)
{
    switch (which) {
        case F_RCP: return rcp3_rn(mk3(1.0f, x, 3.0f)).y;   // ordinary neighbours: the test of x decides
        case F_RCP_SQRT: return rcp_sqrt_rn(x);
        default: return in_smith_domain(x) ? smith_term(x) : 0.0f;
    }
}
// the plain operators: the compiler's general sequences on the device, IEEE on the host
SRT_HD float plain(int which, float x)
{
    switch (which) {
        case F_RCP: return 1.0f / x;
        case F_RCP_SQRT: return 1.0f / __builtin_sqrtf(x);
        default: return in_smith_domain(x) ? 2.0f / (1.0f + __builtin_sqrtf(1.0f + x)) : 0.0f;
    }
}

__global__ void sweep(int which, Report* rep)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned long long sum = 0ull;
    for (uint64_t n = blockIdx.x * blockDim.x + threadIdx.x; n < (1ull << 32); n += stride) {
        const uint32_t b = (uint32_t)n;
        const float x = bitsf(b);
        const uint32_t got = canonical(shipped(which, x)), want = canonical(plain(which, x));
        sum += (unsigned long long)got * (2ull * b + 1ull);
        if (got != want) {
            const unsigned long long k = atomicAdd(&rep->bad, 1ull);
            if (k < 4) rep->first[k] = b;
        }
    }
    atomicAdd(&rep->checksum, sum);
}

static unsigned long long host_checksum(int which)
{
    const unsigned n_threads = 16;
    std::vector<unsigned long long> part(n_threads, 0ull);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < n_threads; t++)
        pool.emplace_back([&, t] {
            unsigned long long sum = 0ull;
            for (uint64_t n = t; n < (1ull << 32); n += n_threads) {
                const uint32_t b = (uint32_t)n;
                sum += (unsigned long long)canonical(plain(which, bitsf(b))) * (2ull * b + 1ull);
            }
            part[t] = sum;
        });
    for (auto& th : pool) th.join();
    unsigned long long sum = 0ull;
    for (unsigned long long p : part) sum += p;
    return sum;
}

int main(int argc, char** argv)
{
    const bool host = !(argc > 1 && std::strcmp(argv[1], "--no-host") == 0);
    Report* d = nullptr;
    if (hipMalloc(&d, sizeof(Report)) != hipSuccess) { std::printf("no device\n"); return 2; }
    int failures = 0;
    for (int f = 0; f < F_COUNT; f++) {
        Report h;
        std::memset(&h, 0, sizeof h);
        if (hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice) != hipSuccess) return 2;
        hipLaunchKernelGGL(sweep, dim3(4096), dim3(256), 0, 0, f, d);
        if (hipDeviceSynchronize() != hipSuccess) { std::printf("kernel failed\n"); return 2; }
        if (hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        const unsigned long long hs = host ? host_checksum(f) : h.checksum;
        std::printf("%s mismatches=%llu checksum_device=%016llx checksum_host=%016llx first=%08x %08x %08x %08x\n",
                    kNames[f], h.bad, h.checksum, hs, h.first[0], h.first[1], h.first[2], h.first[3]);
        if (h.bad != 0ull || hs != h.checksum) failures++;
    }
    (void)hipFree(d);
    std::printf("exhaustive_failures=%d\n", failures);
    return failures ? 1 : 0;
}
