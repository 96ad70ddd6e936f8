/* init_time.c — measurement helper: HIP start-up vs engine creation vs call latency of the C-ABI.
 *   gcc -O2 -o init_time tests/tools/init_time.c -Ltsta_amd/lib -ltsta_hip -Wl,-rpath,$PWD/tsta_amd/lib */
#include <stdint.h>
#include <stdio.h>
#include <time.h>
#include "../../include/tsta_engine.h"
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
int main(void) {
    double t0 = now();
    int n = tsta_device_count();
    double t1 = now();
    tsta_engine* e = 0; int rc = tsta_engine_create(0, &e);
    double t2 = now();
    unsigned char s[8] = {'A','C','G','T','A','C','G','T'}; int64_t o1 = 0, o2 = 4; int l = 4, out = 0; tsta_params p = {2,-5,-2,-4};
    rc |= tsta_psa_score_batch(e, &p, 1, s, 8, &o1, &l, &o2, &l, &out);
    double t3 = now();
    rc |= tsta_psa_score_batch(e, &p, 1, s, 8, &o1, &l, &o2, &l, &out);
    double t4 = now();
    tsta_engine_destroy(e);
    double t5 = now();
    printf("devices=%d rc=%d count %.1f ms, create %.1f ms, first call %.1f ms, second call %.2f ms, destroy %.1f ms\n", n, rc, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t4)*1e3);
    return 0;
}
