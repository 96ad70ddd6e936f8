/* cli_common.h — process-level plumbing shared by the drop-in executables (TSTA_psa, TSTA_psa_notrace, TSTA_msa).
 * A drop-in for a tool that runs 0.1 s on small inputs lives or dies by its start-up: on the MI355X box the ROCr runtime's own
 * hsa_init() takes 60-220 ms (tests/tools/hsa_init_time.c), the first stream 20 ms, the first SDMA copy ~10 ms, and tearing the
 * runtime down at exit() another 30-60 ms.  What a process can do about it, it does here:
 *   - the engine (HIP start-up + stream + self-test) is created by a helper thread while the main thread parses FASTA and builds
 *     the first graph; the first GPU call joins it;
 *   - single-job tools copy with blit kernels instead of the SDMA engines (HSA_ENABLE_SDMA=0 unless the user set it): their
 *     copies are a few MB, and bringing an SDMA queue up costs more than it saves;
 *   - with TSTA_DEVICE=k the runtime is told to look at that device only (ROCR_VISIBLE_DEVICES, unless set) — on a node with
 *     eight GPUs the other seven are not opened;
 *   - after the outputs are flushed and closed the process leaves through _exit(): the kernel reclaims the device state faster
 *     than the runtime's destructors hand it back. */
#ifndef TSTA_CLI_COMMON_H
#define TSTA_CLI_COMMON_H
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#include "../../include/tsta_engine.h"

typedef struct cli_engine { pthread_t th; int started, joined, device, rc; tsta_engine* eng; } cli_engine;

static void* cli_engine_main(void* p)
{
    cli_engine* c = (cli_engine*)p;
    c->rc = tsta_engine_create(c->device, &c->eng);
    return NULL;
}

/* call before anything touches HIP */
static void cli_engine_start(cli_engine* c)
{
    const char* dev = getenv("TSTA_DEVICE");
    c->device = dev ? atoi(dev) : 0;
    c->started = c->joined = 0; c->eng = NULL; c->rc = TSTA_OK;
    setenv("HSA_ENABLE_SDMA", "0", 0);
    if (dev && !getenv("ROCR_VISIBLE_DEVICES") && !getenv("HIP_VISIBLE_DEVICES")) { setenv("ROCR_VISIBLE_DEVICES", dev, 0); c->device = 0; }
    if (pthread_create(&c->th, NULL, cli_engine_main, c) == 0) c->started = 1;
    else cli_engine_main(c);                                   /* no thread: create it here */
}

/* the engine, once it is there (TSTA_OK or the status tsta_engine_create returned) */
static int cli_engine_get(cli_engine* c, tsta_engine** eng)
{
    if (c->started && !c->joined) { pthread_join(c->th, NULL); c->joined = 1; }
    *eng = c->eng;
    return c->rc;
}

/* outputs are written and closed: leave without running the HIP runtime's exit handlers */
static void cli_quick_exit(int code)
{
    fflush(stdout); fflush(stderr);
    _exit(code);
}
#endif
