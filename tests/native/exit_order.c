/* exit_order.c -- a host that closes its handle AFTER the HIP runtime has shut down, as a JVM finalizer or shutdown hook
 * calling NativeSampler.close() can: an atexit handler registered before libmvhdp.so is even loaded runs after the
 * library's own exit handler and after the runtime's teardown.  mvhdp_destroy must then release host memory only and
 * return MVHDP_OK; a second close of the same handle must be refused, not crash.  Built and run by
 * tests/test_gpu_exit.py:  gcc -O1 -o exit_order exit_order.c -ldl && ./exit_order <path to libmvhdp.so> [leak]
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/mvhdp.h"

static void* lib;
static mvhdp_handle handle;
static int (*p_destroy)(mvhdp_handle);

static void late_close(void)
{
    int rc = p_destroy(handle);
    int rc2 = p_destroy(handle);
    printf("late destroy rc=%d second rc=%d\n", rc, rc2);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const int leak = argc > 2 && !strcmp(argv[2], "leak");
    if (!leak) atexit(late_close);                       /* registered FIRST: runs LAST */
    lib = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { printf("dlopen: %s\n", dlerror()); return 2; }
    int (*p_create)(const mvhdp_config*, mvhdp_handle*) = (int (*)(const mvhdp_config*, mvhdp_handle*))dlsym(lib, "mvhdp_create");
    int (*p_set_corpus)(mvhdp_handle, int32_t, int64_t, const int64_t*, const int32_t*) =
        (int (*)(mvhdp_handle, int32_t, int64_t, const int64_t*, const int32_t*))dlsym(lib, "mvhdp_set_corpus");
    int (*p_set_z)(mvhdp_handle, int32_t, const int32_t*) = (int (*)(mvhdp_handle, int32_t, const int32_t*))dlsym(lib, "mvhdp_set_assignments");
    int (*p_set_hyper)(mvhdp_handle, const mvhdp_hyper*) = (int (*)(mvhdp_handle, const mvhdp_hyper*))dlsym(lib, "mvhdp_set_hyper");
    int (*p_build)(mvhdp_handle) = (int (*)(mvhdp_handle))dlsym(lib, "mvhdp_build_counts");
    int (*p_sweep)(mvhdp_handle, uint32_t, uint64_t, uint32_t, const double*, const mvhdp_debug*, mvhdp_sweep_stats*) =
        (int (*)(mvhdp_handle, uint32_t, uint64_t, uint32_t, const double*, const mvhdp_debug*, mvhdp_sweep_stats*))dlsym(lib, "mvhdp_sweep");
    p_destroy = (int (*)(mvhdp_handle))dlsym(lib, "mvhdp_destroy");
    mvhdp_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.num_topics = 5; cfg.num_modalities = 1; cfg.num_types[0] = 11;
    int rc = p_create(&cfg, &handle);
    if (rc) { printf("create rc=%d\n", rc); return 3; }
    int64_t off[4] = {0, 3, 5, 9};
    int32_t tok[9] = {1, 2, 3, 10, 0, 4, 4, 7, 9}, z[9] = {0, 1, 2, 3, 4, 0, 1, 2, 3};
    double alpha[6] = {0.1, 0.1, 0.1, 0.1, 0.1, 0.1};
    mvhdp_hyper hy; memset(&hy, 0, sizeof hy);
    hy.alpha = alpha; hy.alpha_sum[0] = 0.5; hy.beta[0] = 0.01; hy.beta_sum[0] = 0.11; hy.gamma[0] = 1; hy.p_a[0][0] = 0.31; hy.p_b[0][0] = 1;
    mvhdp_sweep_stats st;
    rc = p_set_corpus(handle, 0, 3, off, tok) || p_set_z(handle, 0, z) || p_set_hyper(handle, &hy) || p_build(handle) ||
         p_sweep(handle, 0, 1, 0, NULL, NULL, &st);
    printf("sweep rc=%d tokens=%lld\n", rc, (long long)st.tokens);
    fflush(stdout);
    return rc ? 4 : 0;                                   /* the handle is still open: exit handlers take over */
}
