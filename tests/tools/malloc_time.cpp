// One-off measurement: what a large device allocation costs on this box (first-call stall of tsta_psa_align_batch).
//   hipcc -O2 -o /tmp/malloc_time tests/tools/malloc_time.cpp && /tmp/malloc_time
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    (void)hipSetDevice(0); (void)hipFree(nullptr);
    const size_t GB = 1ull << 30;
    for (size_t gb : {1ull, 8ull, 32ull, 128ull}) {
        void* p = nullptr;
        double t0 = now(); hipError_t e = hipMalloc(&p, gb * GB); double t1 = now();
        (void)hipMemsetAsync(p, 1, 256, nullptr); (void)hipDeviceSynchronize(); double t2 = now();
        (void)hipFree(p); double t3 = now();
        printf("hipMalloc %4zu GB: %8.1f ms (%s)  first touch %6.1f ms  hipFree %8.1f ms\n", gb, (t1 - t0) * 1e3, hipGetErrorString(e), (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    }
    {   // 16 slabs of 8 GB
        std::vector<void*> v(16, nullptr);
        double t0 = now();
        for (auto& p : v) (void)hipMalloc(&p, 8 * GB);
        double t1 = now();
        for (auto& p : v) (void)hipFree(p);
        printf("16 x hipMalloc 8 GB: %8.1f ms, frees %8.1f ms\n", (t1 - t0) * 1e3, (now() - t1) * 1e3);
    }
    {   // again, after the runtime has seen the memory once
        void* p = nullptr; double t0 = now(); (void)hipMalloc(&p, 128 * GB); printf("hipMalloc 128 GB again: %8.1f ms\n", (now() - t0) * 1e3); (void)hipFree(p);
    }
    {   // stream-ordered allocation from the default pool
        hipStream_t s; (void)hipStreamCreate(&s);
        void* p = nullptr; double t0 = now(); hipError_t e = hipMallocAsync(&p, 128 * GB, s); (void)hipStreamSynchronize(s);
        printf("hipMallocAsync 128 GB: %8.1f ms (%s)\n", (now() - t0) * 1e3, hipGetErrorString(e));
        double t1 = now(); (void)hipFreeAsync(p, s); (void)hipStreamSynchronize(s);
        t0 = now(); e = hipMallocAsync(&p, 128 * GB, s); (void)hipStreamSynchronize(s);
        printf("hipMallocAsync 128 GB again (pool): %8.1f ms (%s)\n", (now() - t0) * 1e3, hipGetErrorString(e));
        (void)hipFreeAsync(p, s); (void)hipStreamSynchronize(s); (void)t1;
    }
    {   // virtual memory management: reserve + map in 2 GB chunks
        hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran = 0; (void)hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        const size_t chunk = 2 * GB, total = 64 * GB;
        void* va = nullptr; double t0 = now(); hipError_t e = hipMemAddressReserve(&va, total, 0, nullptr, 0); double t1 = now();
        printf("granularity %zu, reserve 64 GB: %.2f ms (%s)\n", gran, (t1 - t0) * 1e3, hipGetErrorString(e));
        std::vector<hipMemGenericAllocationHandle_t> h;
        if (e == hipSuccess) {
            hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            double tc = 0, tm = 0;
            for (size_t off = 0; off < total; off += chunk) {
                hipMemGenericAllocationHandle_t hh; double a = now();
                if (hipMemCreate(&hh, chunk, &prop, 0) != hipSuccess) { printf("hipMemCreate failed\n"); break; }
                double b = now();
                if (hipMemMap((char*)va + off, chunk, 0, hh, 0) != hipSuccess || hipMemSetAccess((char*)va + off, chunk, &acc, 1) != hipSuccess) { printf("map failed\n"); break; }
                tc += b - a; tm += now() - b; h.push_back(hh);
            }
            printf("32 x (create + map) 2 GB: create %.1f ms, map+access %.1f ms\n", tc * 1e3, tm * 1e3);
            for (size_t i = 0; i < h.size(); ++i) { (void)hipMemUnmap((char*)va + i * chunk, chunk); (void)hipMemRelease(h[i]); }
            (void)hipMemAddressFree(va, total);
        }
    }
    return 0;
}
