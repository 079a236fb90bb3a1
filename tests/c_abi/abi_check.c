/* A compiled-C consumer of include/hmcmt.h and include/hmcmt_mumps.h (test infrastructure).
 *
 *   abi_check                      prints sizeof / offsetof of the ABI structs (compared with the ctypes mirror
 *                                  in hmcmt2d_amd/lib.py by tests/test_abi.py) and the values of the constants
 *   abi_check run <lib.so> <dump>  dlopen()s the library, builds a context from a binary problem dump
 *                                  (tests/c_abi/dump.py), runs hmcmt_create / hmcmt_grad / hmcmt_get_stats /
 *                                  hmcmt_destroy and prints misfit, iteration counts and the outputs' sums; writes
 *                                  pred and grad to <dump>.out (raw doubles) for the Python side to compare
 *
 * Built with: gcc -std=c11 -Wall -Wextra -Werror -I include tests/c_abi/abi_check.c -ldl
 */
#define _POSIX_C_SOURCE 200809L
#include <dlfcn.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hmcmt.h"
#include "hmcmt_debug.h"
#include "hmcmt_mumps.h"

#define FIELD(type, f) printf("offsetof %s.%s %zu\n", #type, #f, offsetof(type, f))

static int print_layout(void) {
    printf("sizeof hmcmt_options %zu\n", sizeof(hmcmt_options));
    FIELD(hmcmt_options, precond); FIELD(hmcmt_options, maxit); FIELD(hmcmt_options, tol);
    FIELD(hmcmt_options, check_every); FIELD(hmcmt_options, verify); FIELD(hmcmt_options, warm_start);
    FIELD(hmcmt_options, fdm_precision);
    printf("sizeof hmcmt_stats %zu\n", sizeof(hmcmt_stats));
    FIELD(hmcmt_stats, iters_fwd_max); FIELD(hmcmt_stats, iters_adj_max); FIELD(hmcmt_stats, iters_fwd_sum);
    FIELD(hmcmt_stats, iters_adj_sum); FIELD(hmcmt_stats, err_est_max); FIELD(hmcmt_stats, true_res_max);
    FIELD(hmcmt_stats, status); FIELD(hmcmt_stats, nsystems); FIELD(hmcmt_stats, fallback_solves); FIELD(hmcmt_stats, smoother_sweeps);
    printf("const HMCMT_NCAT %d\n", HMCMT_NCAT);
    printf("const HMCMT_ENOCONV %d\n", HMCMT_ENOCONV);
    printf("const HMCMT_EBREAKDOWN %d\n", HMCMT_EBREAKDOWN);
    printf("const HMCMT_ENODEV %d\n", HMCMT_ENODEV);
    printf("const HMCMT_PRECOND_FDM_JACOBI %d\n", HMCMT_PRECOND_FDM_JACOBI);
    /* the MUMPS-interface prototypes must be usable as function pointers of the documented shape */
    /* (compared inside sizeof: type-checked by the compiler, never evaluated, so nothing to link against) */
    int64_t (*f1)(const int64_t*, const int64_t*, const int64_t*, const double*, const int64_t*, const int64_t*, int64_t*) = NULL;
    int64_t (*s1)(const int64_t*, const int64_t*, const double*, double*, const int64_t*) = NULL;
    int64_t (*d1)(const int64_t*) = NULL;
    printf("mumps prototypes %zu\n", sizeof(f1 == factor_mumps_) + sizeof(f1 == factor_mumps_cmplx_) + sizeof(s1 == solve_mumps_cmplx_) +
                                          sizeof(s1 == solve_mumps_) + sizeof(d1 == destroy_mumps_cmplx_) + sizeof(d1 == destroy_mumps_));
    return 0;
}

typedef void (*defaults_fn)(hmcmt_options*);
typedef int (*grad_fn)(hmcmt_ctx*, const double*, double*, double*, double*);
typedef int (*stats_fn)(const hmcmt_ctx*, hmcmt_stats*);
typedef const char* (*err_fn)(const hmcmt_ctx*);
typedef int (*destroy_fn)(hmcmt_ctx*);
typedef int (*create_fn)(hmcmt_ctx**, int32_t, int64_t, int64_t, const double*, const double*, const double*, int64_t,
                         const double*, int64_t, const double*, const double*, int64_t, const int64_t*, int64_t,
                         const int64_t*, const int64_t*, const int64_t*, const uint8_t*, const double*, const double*,
                         int64_t, const int64_t*, const double*, const hmcmt_options*);

static void* must_read(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short dump file\n"); exit(2); }
    return p;
}

static int run(const char* libpath, const char* dumppath) {
    void* h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    /* ISO C forbids casting void* to a function pointer; POSIX requires it to work: go through memcpy */
#define SYM(var, type, name) type var; { void* s_ = dlsym(h, name); if (!s_) { fprintf(stderr, "missing %s\n", name); return 2; } memcpy(&var, &s_, sizeof var); }
    SYM(p_create, create_fn, "hmcmt_create")
    SYM(p_defaults, defaults_fn, "hmcmt_default_options")
    SYM(p_grad, grad_fn, "hmcmt_grad")
    SYM(p_stats, stats_fn, "hmcmt_get_stats")
    SYM(p_err, err_fn, "hmcmt_last_error")
    SYM(p_destroy, destroy_fn, "hmcmt_destroy")
    /* the typedefs above must be the header's prototypes (type-checked, not evaluated) */
    (void)sizeof(p_create == hmcmt_create); (void)sizeof(p_grad == hmcmt_grad); (void)sizeof(p_stats == hmcmt_get_stats);
    (void)sizeof(p_defaults == hmcmt_default_options); (void)sizeof(p_err == hmcmt_last_error); (void)sizeof(p_destroy == hmcmt_destroy);
    FILE* f = fopen(dumppath, "rb");
    if (!f) { perror(dumppath); return 2; }
    int64_t hd[8];   /* ny nz nFreq nRx nComp nData nAC device */
    if (fread(hd, sizeof(int64_t), 8, f) != 8) { fprintf(stderr, "short header\n"); return 2; }
    const int64_t ny = hd[0], nz = hd[1], nFreq = hd[2], nRx = hd[3], nComp = hd[4], nData = hd[5], nAC = hd[6];
    double* yLen = must_read(f, sizeof(double) * (size_t)ny);
    double* zLen = must_read(f, sizeof(double) * (size_t)nz);
    double* origin = must_read(f, sizeof(double) * 2);
    double* freqs = must_read(f, sizeof(double) * (size_t)nFreq);
    double* rxY = must_read(f, sizeof(double) * (size_t)nRx);
    double* rxZ = must_read(f, sizeof(double) * (size_t)nRx);
    int64_t* compMode = must_read(f, sizeof(int64_t) * (size_t)nComp);
    int64_t* freqID = must_read(f, sizeof(int64_t) * (size_t)nData);
    int64_t* rxID = must_read(f, sizeof(int64_t) * (size_t)nData);
    int64_t* dtID = must_read(f, sizeof(int64_t) * (size_t)nData);
    uint8_t* dataID = must_read(f, (size_t)(nComp * nRx * nFreq));
    double* obs = must_read(f, sizeof(double) * 2 * (size_t)nData);
    double* dataW = must_read(f, sizeof(double) * (size_t)nData);
    int64_t* activeIdx = must_read(f, sizeof(int64_t) * (size_t)nAC);
    double* bg = must_read(f, sizeof(double) * (size_t)(ny * nz));
    double* m = must_read(f, sizeof(double) * (size_t)nAC);
    fclose(f);
    hmcmt_options opt;
    p_defaults(&opt);
    opt.verify = 1;
    hmcmt_ctx* ctx = NULL;
    int rc = p_create(&ctx, (int32_t)hd[7], ny, nz, yLen, zLen, origin, nFreq, freqs, nRx, rxY, rxZ, nComp, compMode, nData,
                      freqID, rxID, dtID, dataID, obs, dataW, nAC, activeIdx, bg, &opt);
    if (rc) { fprintf(stderr, "hmcmt_create: %d %s\n", rc, p_err(NULL)); return 3; }
    double* pred = calloc(2 * (size_t)nData, sizeof(double));
    double* grad = calloc((size_t)nAC, sizeof(double));
    double misfit = 0;
    rc = p_grad(ctx, m, pred, &misfit, grad);
    if (rc) { fprintf(stderr, "hmcmt_grad: %d %s\n", rc, p_err(ctx)); return 3; }
    hmcmt_stats st;
    p_stats(ctx, &st);
    double sp = 0, sg = 0;
    for (int64_t i = 0; i < 2 * nData; ++i) sp += pred[i];
    for (int64_t i = 0; i < nAC; ++i) sg += grad[i];
    printf("misfit %.17g\niters %d %d\ntrue_res %.3e\nstatus %d\nnsystems %d\nsum_pred %.17g\nsum_grad %.17g\n", misfit,
           st.iters_fwd_max, st.iters_adj_max, st.true_res_max, st.status, st.nsystems, sp, sg);
    char outpath[4096];
    snprintf(outpath, sizeof outpath, "%s.out", dumppath);
    FILE* o = fopen(outpath, "wb");
    if (!o) { perror(outpath); return 2; }
    fwrite(pred, sizeof(double), 2 * (size_t)nData, o);
    fwrite(grad, sizeof(double), (size_t)nAC, o);
    fclose(o);
    rc = p_destroy(ctx);
    printf("destroy %d\n", rc);
    dlclose(h);
    return rc;
}

int main(int argc, char** argv) {
    if (argc == 4 && strcmp(argv[1], "run") == 0) return run(argv[2], argv[3]);
    return print_layout();
}
