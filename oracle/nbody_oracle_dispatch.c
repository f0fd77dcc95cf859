/*
 * nbody_oracle_dispatch.c -- ISA dispatch + multi-step drivers for the CPU oracle.
 * TEST INFRASTRUCTURE ONLY (see the header of nbody_oracle.c; parity unpinned).
 */
#include <stddef.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DECL(isa)                                                                                \
    void nbo_naive_step_f32_##isa(const float *, float *, uint32_t, float, float, float,         \
                                  uint32_t, uint32_t);                                           \
    void nbo_naive_step_f64_##isa(const double *, double *, uint32_t, double, double, double,    \
                                  uint32_t, uint32_t);
DECL(v3)
DECL(v4)

static int use_v4(void) {
    static int cached = -1;
    if (cached < 0) {
        __builtin_cpu_init();
        cached = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") &&
                 __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl");
    }
    return cached;
}

/* "x86-64-v4" or "x86-64-v3": which build of the kernels this host runs. */
const char *nbo_isa(void) { return use_v4() ? "x86-64-v4" : "x86-64-v3"; }

int nbo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void nbo_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

void nbo_naive_step_f32(const float *src, float *dst, uint32_t n, float g, float e, float dt,
                        uint32_t i_lo, uint32_t i_hi) {
    if (use_v4())
        nbo_naive_step_f32_v4(src, dst, n, g, e, dt, i_lo, i_hi);
    else
        nbo_naive_step_f32_v3(src, dst, n, g, e, dt, i_lo, i_hi);
}

void nbo_naive_step_f64(const double *src, double *dst, uint32_t n, double g, double e, double dt,
                        uint32_t i_lo, uint32_t i_hi) {
    if (use_v4())
        nbo_naive_step_f64_v4(src, dst, n, g, e, dt, i_lo, i_hi);
    else
        nbo_naive_step_f64_v3(src, dst, n, g, e, dt, i_lo, i_hi);
}

/* `steps` full steps with the reference's ping-pong (src/sims/naive.rs:113-132,
 * 156-160).  buf_a holds the initial state; returns 0 if the final state is in
 * buf_a, 1 if it is in buf_b. */
int nbo_naive_run_f32(float *buf_a, float *buf_b, uint32_t n, float g, float e, float dt,
                      uint32_t steps) {
    float *s = buf_a, *d = buf_b;
    for (uint32_t k = 0; k < steps; ++k) {
        nbo_naive_step_f32(s, d, n, g, e, dt, 0, n);
        float *t = s;
        s = d;
        d = t;
    }
    return s == buf_a ? 0 : 1;
}

int nbo_naive_run_f64(double *buf_a, double *buf_b, uint32_t n, double g, double e, double dt,
                      uint32_t steps) {
    double *s = buf_a, *d = buf_b;
    for (uint32_t k = 0; k < steps; ++k) {
        nbo_naive_step_f64(s, d, n, g, e, dt, 0, n);
        double *t = s;
        s = d;
        d = t;
    }
    return s == buf_a ? 0 : 1;
}
