// One-off: what the FIRST host<->device copies of a process cost, from pageable and from pinned host memory
// (hipcc -O2 first_copy_time.cpp -o /tmp/fct && /tmp/fct [pinned])
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const bool pinned = argc > 1 && !strcmp(argv[1], "pinned");
    double t = now();
    int n = 0; hipGetDeviceCount(&n); printf("hipGetDeviceCount %.1f ms\n", now() - t);
    t = now(); hipSetDevice(0); printf("hipSetDevice %.1f ms\n", now() - t);
    t = now(); int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0); printf("hipDeviceGetAttribute (%d CUs) %.1f ms\n", cus, now() - t);
    if (argc > 2) { t = now(); hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); printf("hipGetDeviceProperties %.1f ms\n", now() - t); }
    t = now(); hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); printf("first stream %.1f ms\n", now() - t);
    t = now(); hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); printf("second stream %.1f ms\n", now() - t);
    t = now(); hipEvent_t ev; hipEventCreate(&ev); printf("event %.2f ms\n", now() - t);
    t = now(); void* d; hipMalloc(&d, 64 << 20); printf("hipMalloc 64 MB %.1f ms\n", now() - t);
    const size_t bytes = 200 << 10;
    char* h;
    t = now();
    if (pinned) hipHostMalloc((void**)&h, 4 << 20, hipHostMallocDefault); else h = (char*)malloc(4 << 20);
    memset(h, 1, 4 << 20);
    printf("%s host buffer 4 MB %.1f ms\n", pinned ? "pinned" : "pageable", now() - t);
    for (int k = 0; k < 3; ++k) { t = now(); hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); printf("H2D %zu B #%d %.2f ms\n", bytes, k, now() - t); }
    for (int k = 0; k < 3; ++k) { t = now(); hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); printf("D2H async 64 B #%d %.2f ms\n", k, now() - t); }
    for (int k = 0; k < 3; ++k) { t = now(); hipMemcpy(h, d, 50000, hipMemcpyDeviceToHost); printf("D2H sync 50 KB #%d %.2f ms\n", k, now() - t); }
    return 0;
}
