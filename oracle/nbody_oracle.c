/*
 * nbody_oracle.c -- CPU restatement of the reference's all-pairs step.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP path:
 * only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 * load it.  Nothing under wgpu_n_body_amd/ links, imports or calls it, and the
 * product has no CPU fallback.
 *
 * PARITY UNPINNED: the reference (arpan-dhatt/wgpu-n-body) has no tests, no
 * golden vectors and no CPU force path, and it cannot be built here (Rust +
 * wgpu; neither exists in this image).  The force arithmetic of the reference
 * runs inside naga 0.8.2 -> wgpu-core/hal 0.12.2 -> the GPU driver
 * (Cargo.lock:987,1860,1883), where `distance`, `normalize` and `/` have
 * driver-defined ULP error, so the reference's own output is not bit-defined.
 * What pins this oracle instead: (1) analytic known-answer tests
 * (tests/test_oracle_kat.py), (2) agreement with an independent numpy
 * restatement (oracle/oracle_np.py) and (3) fp32-vs-fp64 agreement bounds.
 *
 * What it restates (paths relative to the reference crate root):
 *   src/sims/shaders/naive.wgsl:23-48   getAcc  -- all-pairs force
 *   src/sims/shaders/naive.wgsl:50-69   main    -- kick-drift-kick integrator
 *   src/sims/naive.rs:113-132,156-160           -- ping-pong (caller swaps)
 *   src/sims/mod.rs:9-16,51-58                  -- Particle (10 x f32), SimParams
 *
 * Two modes:
 *   nbo_naive_step_f32 : every operation in binary32, in the order the WGSL
 *                        source writes it, j ascending, no FMA contraction
 *                        (build with -ffp-contract=off).  "Literal" mode.
 *   nbo_naive_step_f64 : same formula in binary64 on a binary64 state, as the
 *                        accuracy reference the fp32 results are judged against.
 *
 * Bodies are independent within a step (each reads only src), so the i range
 * [i_lo, i_hi) may be any subrange; OpenMP parallelises over i-blocks and each
 * body's j loop stays strictly sequential, so results do not depend on the
 * thread count.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define NBO_LANES 16 /* bodies advanced together; one SIMD lane each */

/* This file is compiled once per ISA level (Makefile: -DNBO_ISA=v3 with
 * -march=x86-64-v3, -DNBO_ISA=v4 with -march=x86-64-v4); nbody_oracle_dispatch.c
 * picks the widest one the host CPU supports, so a library built in one
 * container runs on another machine.  Same source, same operation order, and
 * sqrt/div are correctly rounded at every width: results are identical. */
#ifndef NBO_ISA
#define NBO_ISA v3
#endif
#define NBO_CAT_(a, b) a##_##b
#define NBO_CAT(a, b) NBO_CAT_(a, b)
#define NBO_FN(name) NBO_CAT(name, NBO_ISA)

/* One step of naive.wgsl for bodies [i_lo, i_hi), binary32, literal order.
 * src, dst: n x 10 floats (px py pz vx vy vz ax ay az mass). */
void NBO_FN(nbo_naive_step_f32)(const float *src, float *dst, uint32_t n, float g, float e, float dt,
                        uint32_t i_lo, uint32_t i_hi) {
    if (i_hi > n) i_hi = n;
    if (i_lo >= i_hi) return;
    const int64_t nblk = ((int64_t)(i_hi - i_lo) + NBO_LANES - 1) / NBO_LANES;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t b = 0; b < nblk; ++b) {
        float px[NBO_LANES], py[NBO_LANES], pz[NBO_LANES];
        float vx[NBO_LANES], vy[NBO_LANES], vz[NBO_LANES];
        float ax[NBO_LANES], ay[NBO_LANES], az[NBO_LANES];
        uint32_t idx[NBO_LANES];
        const uint32_t base = i_lo + (uint32_t)b * NBO_LANES;
        for (int l = 0; l < NBO_LANES; ++l) {
            /* lanes past the end replay the last body; their stores are skipped */
            uint32_t i = base + (uint32_t)l;
            if (i >= i_hi) i = i_hi - 1;
            idx[l] = i;
            const float *p = src + (size_t)i * 10;
            /* naive.wgsl:63  aVel = aVel + aAcc * params.dt / 2.0  == v + ((a*dt)/2) */
            vx[l] = p[3] + (p[6] * dt) / 2.0f;
            vy[l] = p[4] + (p[7] * dt) / 2.0f;
            vz[l] = p[5] + (p[8] * dt) / 2.0f;
            /* naive.wgsl:64  aPos = aPos + aVel * params.dt */
            px[l] = p[0] + vx[l] * dt;
            py[l] = p[1] + vy[l] * dt;
            pz[l] = p[2] + vz[l] * dt;
            ax[l] = ay[l] = az[l] = 0.0f; /* naive.wgsl:24 */
        }
        /* naive.wgsl:26-46: j ascending over the OLD positions in src */
        for (uint32_t j = 0; j < n; ++j) {
            const float *q = src + (size_t)j * 10;
            const float qx = q[0], qy = q[1], qz = q[2];
            const float mg = q[9] * g; /* _q.mass * params.g (left-assoc, :39) */
#pragma omp simd
            for (int l = 0; l < NBO_LANES; ++l) {
                /* distance(aPos,bPos) = length(aPos-bPos); normalize(bPos-aPos) */
                const float dx = qx - px[l], dy = qy - py[l], dz = qz - pz[l];
                const float r = sqrtf((dx * dx + dy * dy) + dz * dz);
                const float s = mg / ((r * r) * r + e);
                /* force = s * (d / r); acc = acc + force * dt   (:39-41) */
                const float fx = (s * (dx / r)) * dt;
                const float fy = (s * (dy / r)) * dt;
                const float fz = (s * (dz / r)) * dt;
                const int skip = (j == idx[l]); /* :30-32, exclusion by INDEX */
                ax[l] = skip ? ax[l] : ax[l] + fx;
                ay[l] = skip ? ay[l] : ay[l] + fy;
                az[l] = skip ? az[l] : az[l] + fz;
            }
        }
        for (int l = 0; l < NBO_LANES; ++l) {
            const uint32_t i = base + (uint32_t)l;
            if (i >= i_hi) break;
            float *o = dst + (size_t)i * 10;
            o[0] = px[l];
            o[1] = py[l];
            o[2] = pz[l];
            /* naive.wgsl:66  aVel = aVel + acc * params.dt / 2.0 */
            o[3] = vx[l] + (ax[l] * dt) / 2.0f;
            o[4] = vy[l] + (ay[l] * dt) / 2.0f;
            o[5] = vz[l] + (az[l] * dt) / 2.0f;
            o[6] = ax[l];
            o[7] = ay[l];
            o[8] = az[l];
            o[9] = src[(size_t)i * 10 + 9]; /* :68 mass carried over */
        }
    }
}

/* Same step in binary64 on a binary64 state (n x 10 doubles).  g, e, dt are the
 * binary32 parameter values widened exactly. */
void NBO_FN(nbo_naive_step_f64)(const double *src, double *dst, uint32_t n, double g, double e, double dt,
                        uint32_t i_lo, uint32_t i_hi) {
    if (i_hi > n) i_hi = n;
    if (i_lo >= i_hi) return;
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t ii = (int64_t)i_lo; ii < (int64_t)i_hi; ++ii) {
        const uint32_t i = (uint32_t)ii;
        const double *p = src + (size_t)i * 10;
        const double vx = p[3] + (p[6] * dt) / 2.0, vy = p[4] + (p[7] * dt) / 2.0,
                     vz = p[5] + (p[8] * dt) / 2.0;
        const double px = p[0] + vx * dt, py = p[1] + vy * dt, pz = p[2] + vz * dt;
        double ax = 0.0, ay = 0.0, az = 0.0;
        for (uint32_t j = 0; j < n; ++j) {
            if (j == i) continue;
            const double *q = src + (size_t)j * 10;
            const double dx = q[0] - px, dy = q[1] - py, dz = q[2] - pz;
            const double r = sqrt((dx * dx + dy * dy) + dz * dz);
            const double s = (q[9] * g) / ((r * r) * r + e);
            ax += (s * (dx / r)) * dt;
            ay += (s * (dy / r)) * dt;
            az += (s * (dz / r)) * dt;
        }
        double *o = dst + (size_t)i * 10;
        o[0] = px;
        o[1] = py;
        o[2] = pz;
        o[3] = vx + (ax * dt) / 2.0;
        o[4] = vy + (ay * dt) / 2.0;
        o[5] = vz + (az * dt) / 2.0;
        o[6] = ax;
        o[7] = ay;
        o[8] = az;
        o[9] = p[9];
    }
}

