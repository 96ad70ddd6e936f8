/* hsa_init_time.c — measurement helper: what of the process start-up is the ROCr runtime (hsa_init), what is HIP on top of it.
 *   gcc -O2 -o hsa_init_time tests/tools/hsa_init_time.c -I/opt/rocm/include -L/opt/rocm/lib -lhsa-runtime64 -lamdhip64 -D__HIP_PLATFORM_AMD__ */
#include <stdio.h>
#include <time.h>
#include <hsa/hsa.h>
#include <hip/hip_runtime_api.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
int main(void) {
    double t0 = now();
    hsa_status_t st = hsa_init();
    double t1 = now();
    int n = 0; hipError_t e = hipGetDeviceCount(&n);
    double t2 = now();
    hipStream_t s; e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double t3 = now();
    void* p = 0; e = hipMalloc(&p, 1 << 20);
    double t4 = now();
    printf("hsa_init %.1f ms (status %d), hipGetDeviceCount after it %.1f ms (%d devices, err %d), first stream %.1f ms, first hipMalloc %.1f ms\n",
           (t1 - t0) * 1e3, (int)st, (t2 - t1) * 1e3, n, (int)e, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
    return 0;
}
