// nb_naive.hip -- all-pairs force accumulation + kick-drift-kick integrator for gfx950.
//
// Replaces the reference's WGSL compute shader src/sims/shaders/naive.wgsl:23-69
// (getAcc + main) and its dispatch NaiveSim::encode (src/sims/naive.rs:147-162).
// The reference runs one thread per body with an O(N) loop of 40-byte global loads and
// no shared memory; this is a from-scratch CDNA4 design:
//
//   * state is SoA: posm[j] = float4{x,y,z,m} is the only array the O(N^2) loop reads
//     (16 B/body instead of 40), vel/acc are touched once per body per step;
//   * a workgroup owns an i-tile of 64*IB bodies (IB bodies per lane, held in VGPRs) and
//     its W waves split the j range; partial sums meet in LDS in a fixed order, so the
//     result is deterministic and independent of scheduling (no atomics);
//   * the j stream is staged per wave through LDS (coalesced 1 KiB global_load_dwordx4,
//     ds_write_b128, then wave-uniform ds_read_b128 broadcasts) -- or, in the SMEM
//     variants, pulled through the scalar cache into SGPRs (wave-uniform s_load), which
//     costs no VALU, no VGPR and no LDS issue at all;
//   * per pair: 3 sub, 1 mul, 2 fma (r2); v_sqrt; 1 mul, 1 fma (r^4 + e r); v_rcp;
//     1 mul (m_j *), 3 fma = 12 full-rate + 2 quarter-rate VALU ops.  The reference's
//     m g /(r^3+e) * (d/r) * dt  is evaluated as  (g dt) * m d / (r^4 + e r):  g*dt is
//     applied once per body after the sum;
//   * self-exclusion is by INDEX as in naive.wgsl:30-32 (the body's new position differs
//     from its own old one, so r != 0), but only the j tiles that overlap the
//     workgroup's own i range (and the zero-padded tail tile) run the masked loop body.
//
// Summation order differs from the reference's sequential j loop (j is split over waves
// and unrolled), so parity with the oracle is tolerance-based; see DESIGN.md.
#include "nb_common.hpp"

#include <type_traits>

namespace nb {
namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

enum JSource { kLds = 0, kSmem = 1 };

// Which j tiles a launch sums.  Virtual tile v in [0, count) maps to the real tile
// base + v + (v >= skip_at ? skip_len : 0).  {0, n_tiles, ~0, 0} = all tiles.  A two-phase
// (overlapped) step sums the rank's OWN tiles first ({lo_tile, local, ~0, 0}: their positions
// are already on this GPU) and the others after the all-gather ({0, rest, lo_tile, local}).
struct TileWindow {
    uint32_t base, count, skip_at, skip_len;
    __device__ __forceinline__ uint32_t tile(uint32_t v) const {
        return base + v + (v >= skip_at ? skip_len : 0u);
    }
};

// The O(N) integrator lines are evaluated exactly as naive.wgsl:63-66 writes them -- one
// rounding per operation, no FMA contraction -- so a body's new position is bit-identical
// to the literal fp32 oracle's (it depends only on the body's own x, v, a).
__device__ __forceinline__ float kick(float v, float a, float dt) {
#pragma clang fp contract(off)
    return v + (a * dt) / 2.0f;  // aVel + aAcc * params.dt / 2.0
}
__device__ __forceinline__ float drift(float x, float v, float dt) {
#pragma clang fp contract(off)
    return x + v * dt;  // aPos + aVel * params.dt
}

// One pair: accumulate m_j * d / (r^4 + e r) into (ax,ay,az).
template <bool MASKED>
__device__ __forceinline__ void pair(float xj, float yj, float zj, float mj, float xi, float yi,
                                     float zi, float e, bool valid, float &ax, float &ay,
                                     float &az) {
    const float dx = xj - xi, dy = yj - yi, dz = zj - zi;
    const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    const float r = __builtin_amdgcn_sqrtf(r2);           // v_sqrt_f32
    const float den = __builtin_fmaf(e, r, r2 * r2);      // r^4 + e r = r (r^3 + e)
    float w = mj * __builtin_amdgcn_rcpf(den);            // v_rcp_f32
    if (MASKED) w = valid ? w : 0.0f;                     // self / padding: contributes exactly 0
    ax = __builtin_fmaf(w, dx, ax);
    ay = __builtin_fmaf(w, dy, ay);
    az = __builtin_fmaf(w, dz, az);
}

// Two i bodies against one j body with packed fp32 (v_pk_*), lane-pairs in v2f registers.
template <bool MASKED>
__device__ __forceinline__ void pair2(float xj, float yj, float zj, float mj, v2f xi, v2f yi,
                                      v2f zi, float e, bool valid0, bool valid1, v2f &ax,
                                      v2f &ay, v2f &az) {
    const v2f dx = v2f{xj, xj} - xi, dy = v2f{yj, yj} - yi, dz = v2f{zj, zj} - zi;
    const v2f r2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
    v2f r;
    r.x = __builtin_amdgcn_sqrtf(r2.x);
    r.y = __builtin_amdgcn_sqrtf(r2.y);
    const v2f den = __builtin_elementwise_fma(v2f{e, e}, r, r2 * r2);
    v2f rc;
    rc.x = __builtin_amdgcn_rcpf(den.x);
    rc.y = __builtin_amdgcn_rcpf(den.y);
    v2f w = v2f{mj, mj} * rc;
    if (MASKED) {
        w.x = valid0 ? w.x : 0.0f;
        w.y = valid1 ? w.y : 0.0f;
    }
    ax = __builtin_elementwise_fma(w, dx, ax);
    ay = __builtin_elementwise_fma(w, dy, ay);
    az = __builtin_elementwise_fma(w, dz, az);
}

// The body of one 64-body j tile held in `tile` (LDS, wave-private) or read from global
// memory through the scalar cache (SMEM).  j0 = global index of the tile's first body.
// State is IB scalars per lane, or IB/2 packed pairs (v2f) when PACKED.
template <int IB, bool PACKED, bool MASKED, int UNROLL, typename Ptr, typename T>
__device__ __forceinline__ void tile_body(Ptr tile, uint32_t j0, uint32_t n, const uint32_t *ii,
                                          const T *xi, const T *yi, const T *zi, float e, T *ax,
                                          T *ay, T *az) {
#pragma unroll UNROLL
    for (uint32_t jj = 0; jj < kJTile; ++jj) {
        const float4 pj = tile[jj];  // wave-uniform address: LDS broadcast / s_load
        const uint32_t j = j0 + jj;
        if constexpr (PACKED) {
#pragma unroll
            for (int k = 0; k < IB / 2; ++k)
                pair2<MASKED>(pj.x, pj.y, pj.z, pj.w, xi[k], yi[k], zi[k], e,
                              (j != ii[2 * k]) & (j < n), (j != ii[2 * k + 1]) & (j < n), ax[k],
                              ay[k], az[k]);
        } else {
#pragma unroll
            for (int k = 0; k < IB; ++k)
                pair<MASKED>(pj.x, pj.y, pj.z, pj.w, xi[k], yi[k], zi[k], e,
                             (j != ii[k]) & (j < n), ax[k], ay[k], az[k]);
        }
    }
}

template <typename T>
__device__ __forceinline__ float &elem(T *a, int k);
template <>
__device__ __forceinline__ float &elem<float>(float *a, int k) {
    return a[k];
}
template <>
__device__ __forceinline__ float &elem<v2f>(v2f *a, int k) {
    return reinterpret_cast<float *>(a)[k];  // fully unrolled callers: stays in registers
}

// grid.x = ceil((hi-lo) / (64*IB)) i-tiles; grid.y = JS j-splits; block = 64*W threads.
// JS == 1: the workgroup sees every j and finishes the step itself.  JS > 1 (few bodies per
// launch: small N, or one rank's share of a multi-GPU run): workgroup (b, s) sums the j tiles
// t with (t mod JS*W) in [s*W, (s+1)*W) and writes its partial sums to `partial[s][i-lo]`;
// naive_finish_kernel then adds the JS partials in order s = 0..JS-1 and integrates.  Either
// way a body's sum is built in one fixed order: results are deterministic.
template <int IB, int W, int SRC, bool PACKED, int UNROLL>
__global__ __launch_bounds__(64 * W) void naive_step_kernel(
    const float4 *__restrict__ posm_src, float4 *__restrict__ posm_dst, float4 *__restrict__ vel,
    float4 *__restrict__ acc, float4 *__restrict__ partial, uint32_t partial_stride, uint32_t n,
    uint32_t n_pad, uint32_t lo, uint32_t hi, float g, float e, float dt, TileWindow win) {
    static_assert(!PACKED || IB % 2 == 0, "packed fp32 needs an even number of bodies per lane");
    using T = typename std::conditional<PACKED, v2f, float>::type;
    constexpr int NV = PACKED ? IB / 2 : IB;
    // LDS: per-wave j tile (double-buffered) + the cross-wave reduction scratch.
    __shared__ float4 s_tile[SRC == kLds ? W * 2 * kJTile : 1];
    __shared__ float s_red[W > 1 ? (W - 1) * IB * 3 * 64 : 1];

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t i0 = lo + blockIdx.x * (64u * IB);  // first body of this workgroup's i-tile
    const uint32_t slot = blockIdx.y * W + wave;        // this wave's share of the j tiles
    const uint32_t n_slots = gridDim.y * W;

    // ---- kick + drift (naive.wgsl:63-64), every wave redundantly for its own lanes -------
    uint32_t ii[IB];
    float vhx[IB], vhy[IB], vhz[IB], mi[IB];
    T xi[NV], yi[NV], zi[NV], ax[NV], ay[NV], az[NV];
#pragma unroll
    for (int k = 0; k < IB; ++k) {
        ii[k] = i0 + lane + 64u * k;
        // bodies past hi replay body hi-1 (loads stay in bounds); their stores are skipped
        const uint32_t ic = ii[k] < hi ? ii[k] : hi - 1u;
        const float4 p = posm_src[ic];
        const float4 v = vel[ic - lo];
        const float4 a = acc[ic - lo];
        vhx[k] = kick(v.x, a.x, dt);
        vhy[k] = kick(v.y, a.y, dt);
        vhz[k] = kick(v.z, a.z, dt);
        elem<T>(xi, k) = drift(p.x, vhx[k], dt);
        elem<T>(yi, k) = drift(p.y, vhy[k], dt);
        elem<T>(zi, k) = drift(p.z, vhz[k], dt);
        mi[k] = p.w;
        elem<T>(ax, k) = 0.0f;
        elem<T>(ay, k) = 0.0f;
        elem<T>(az, k) = 0.0f;
    }

    // ---- all-pairs over this wave's share of the j tiles (naive.wgsl:26-46) --------------
    const uint32_t n_tiles = (n + kJTile - 1u) / kJTile;  // tail tile reads zero padding
    const uint32_t t_self_lo = i0 / kJTile, t_self_hi = (i0 + 64u * IB - 1u) / kJTile;
    if constexpr (SRC == kLds) {
        float4 *my = s_tile + wave * 2 * kJTile;
        uint32_t vt = slot;
        float4 nxt = vt < win.count ? posm_src[win.tile(vt) * kJTile + lane] : float4{0, 0, 0, 0};
        uint32_t buf = 0;
        for (; vt < win.count; vt += n_slots) {
            const uint32_t t = win.tile(vt);
            my[buf * kJTile + lane] = nxt;  // ds_write_b128; wave-private, no s_barrier needed
            __builtin_amdgcn_wave_barrier();
            const uint32_t vn = vt + n_slots;
            if (vn < win.count) nxt = posm_src[win.tile(vn) * kJTile + lane];  // prefetch next tile
            const float4 *tile = my + buf * kJTile;
            const bool special = (t >= t_self_lo && t <= t_self_hi) || (t + 1u == n_tiles);
            if (special)
                tile_body<IB, PACKED, true, 2>(tile, t * kJTile, n, ii, xi, yi, zi, e, ax, ay, az);
            else
                tile_body<IB, PACKED, false, UNROLL>(tile, t * kJTile, n, ii, xi, yi, zi, e, ax, ay,
                                                     az);
            __builtin_amdgcn_wave_barrier();
            buf ^= 1u;
        }
    } else {
        for (uint32_t vt = slot; vt < win.count; vt += n_slots) {
            const uint32_t t = win.tile(vt);
            const float4 *tile = posm_src + t * kJTile;  // wave-uniform -> s_load_dwordx4+
            const bool special = (t >= t_self_lo && t <= t_self_hi) || (t + 1u == n_tiles);
            if (special)
                tile_body<IB, PACKED, true, 2>(tile, t * kJTile, n, ii, xi, yi, zi, e, ax, ay, az);
            else
                tile_body<IB, PACKED, false, UNROLL>(tile, t * kJTile, n, ii, xi, yi, zi, e, ax, ay,
                                                     az);
        }
    }

    // ---- deterministic cross-wave reduction: wave 0 adds waves 1..W-1 in order -----------
    float sx[IB], sy[IB], sz[IB];
#pragma unroll
    for (int k = 0; k < IB; ++k) {
        sx[k] = elem<T>(ax, k);
        sy[k] = elem<T>(ay, k);
        sz[k] = elem<T>(az, k);
    }
    if constexpr (W > 1) {
        if (wave != 0) {
            float *dst = s_red + (wave - 1u) * (IB * 3 * 64);
#pragma unroll
            for (int k = 0; k < IB; ++k) {
                dst[(k * 3 + 0) * 64 + lane] = sx[k];
                dst[(k * 3 + 1) * 64 + lane] = sy[k];
                dst[(k * 3 + 2) * 64 + lane] = sz[k];
            }
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll 1  // unrolled over 15 waves x 12 sums the loads all hoist and spill
        for (int w = 1; w < W; ++w) {
            const float *src = s_red + (w - 1) * (IB * 3 * 64);
#pragma unroll
            for (int k = 0; k < IB; ++k) {
                sx[k] += src[(k * 3 + 0) * 64 + lane];
                sy[k] += src[(k * 3 + 1) * 64 + lane];
                sz[k] += src[(k * 3 + 2) * 64 + lane];
            }
        }
    }

    if (partial) {  // j-split / two-phase step: hand the partial sums to naive_finish_kernel
#pragma unroll
        for (int k = 0; k < IB; ++k)
            if (ii[k] < hi)
                partial[(size_t)blockIdx.y * partial_stride + (ii[k] - lo)] =
                    float4{sx[k], sy[k], sz[k], 0.0f};
        return;
    }

    // ---- second kick + store (naive.wgsl:66-68) -------------------------------------------
    const float gdt = g * dt;
#pragma unroll
    for (int k = 0; k < IB; ++k) {
        if (ii[k] >= hi) continue;
        const float fx = sx[k] * gdt, fy = sy[k] * gdt, fz = sz[k] * gdt;  // stored "acceleration"
        posm_dst[ii[k]] = float4{elem<T>(xi, k), elem<T>(yi, k), elem<T>(zi, k), mi[k]};
        vel[ii[k] - lo] = float4{kick(vhx[k], fx, dt), kick(vhy[k], fy, dt), kick(vhz[k], fz, dt),
                                 0.0f};
        acc[ii[k] - lo] = float4{fx, fy, fz, 0.0f};
    }
}

// Second half of a j-split step: one thread per body adds the JS partial sums in fixed order
// and applies the integrator exactly as the single-kernel path does (naive.wgsl:63-68).
__global__ __launch_bounds__(256) void naive_finish_kernel(
    const float4 *__restrict__ posm_src, float4 *__restrict__ posm_dst, float4 *__restrict__ vel,
    float4 *__restrict__ acc, const float4 *__restrict__ partial, uint32_t partial_stride,
    uint32_t js, uint32_t lo, uint32_t hi, float g, float dt, PeerDst peers) {
    const uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    const float4 p = posm_src[i], v = vel[i - lo], a = acc[i - lo];
    const float vhx = kick(v.x, a.x, dt), vhy = kick(v.y, a.y, dt), vhz = kick(v.z, a.z, dt);
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    for (uint32_t s = 0; s < js; ++s) {
        const float4 q = partial[(size_t)s * partial_stride + (i - lo)];
        if (s == 0) {
            sx = q.x; sy = q.y; sz = q.z;
        } else {
            sx += q.x; sy += q.y; sz += q.z;
        }
    }
    const float gdt = g * dt;
    const float fx = sx * gdt, fy = sy * gdt, fz = sz * gdt;
    const float4 pn = float4{drift(p.x, vhx, dt), drift(p.y, vhy, dt), drift(p.z, vhz, dt), p.w};
    posm_dst[i] = pn;
    // one-process multi-GPU: the same slot of every peer's next-step buffer (stores over xGMI;
    // visible to the peer when this kernel has completed, which its stream waits for)
    for (uint32_t k = 0; k < peers.n; ++k) peers.p[k][i] = pn;
    vel[i - lo] = float4{kick(vhx, fx, dt), kick(vhy, fy, dt), kick(vhz, fz, dt), 0.0f};
    acc[i - lo] = float4{fx, fy, fz, 0.0f};
}

// ---- AoS <-> SoA at the boundary ------------------------------------------------------------
__global__ void aos_to_soa_kernel(const nb_particle *__restrict__ aos, float4 *__restrict__ posm,
                                  float4 *__restrict__ vel, float4 *__restrict__ acc, uint32_t n,
                                  uint32_t lo, uint32_t hi) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const nb_particle p = aos[i];
    posm[i] = float4{p.position[0], p.position[1], p.position[2], p.mass};
    if (i >= lo && i < hi) {
        vel[i - lo] = float4{p.velocity[0], p.velocity[1], p.velocity[2], 0.0f};
        acc[i - lo] = float4{p.acceleration[0], p.acceleration[1], p.acceleration[2], 0.0f};
    }
}

__global__ void soa_to_aos_kernel(const float4 *__restrict__ posm, const float4 *__restrict__ vel,
                                  const float4 *__restrict__ acc, nb_particle *__restrict__ aos,
                                  uint32_t n, uint32_t lo, uint32_t hi) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = posm[i];
    float4 v{0, 0, 0, 0}, a{0, 0, 0, 0};
    if (i >= lo && i < hi) {
        v = vel[i - lo];
        a = acc[i - lo];
    }
    nb_particle o;
    o.position[0] = p.x; o.position[1] = p.y; o.position[2] = p.z;
    o.velocity[0] = v.x; o.velocity[1] = v.y; o.velocity[2] = v.z;
    o.acceleration[0] = a.x; o.acceleration[1] = a.y; o.acceleration[2] = a.z;
    o.mass = p.w;
    aos[i] = o;
}

// ---- variant table ----------------------------------------------------------------------------
using KernelFn = void (*)(const float4 *, float4 *, float4 *, float4 *, float4 *, uint32_t,
                          uint32_t, uint32_t, uint32_t, uint32_t, float, float, float, TileWindow);
struct Variant {
    const char *name;
    KernelFn fn;
    int ib, w;
};
#define NB_V(IB, W, SRC, PK, UN) \
    { "ib" #IB "_w" #W "_" #SRC "_pk" #PK "_u" #UN, naive_step_kernel<IB, W, SRC, PK, UN>, IB, W }
// The shipped table holds the shapes that win somewhere (profiles/r01_variant_sweep.txt); the full
// sweep of round 1 (29 shapes, ~1 min of hipcc) is compiled only with -DNB_ALL_VARIANTS
// (tools/variant_sweep.py builds it on demand).
const Variant kVariants[] = {
    NB_V(2, 8, kLds, false, 8),   // 0: unpacked baseline (2 bodies per lane, 8 waves)
    NB_V(2, 8, kLds, true, 8),    // 1: the same, packed fp32
    NB_V(2, 16, kSmem, true, 8),  // 2: j stream through the scalar cache, no LDS (within 1 % of 4)
    NB_V(4, 16, kLds, true, 2),   // 3: 4 bodies per lane, 16 waves, j loop unrolled 2x
    NB_V(4, 16, kLds, true, 4),   // 4: default -- 4 bodies per lane, 16 waves, unrolled 4x
#ifdef NB_ALL_VARIANTS
    NB_V(2, 8, kSmem, false, 8), NB_V(2, 8, kSmem, true, 8), NB_V(2, 4, kLds, false, 8),
    NB_V(2, 4, kLds, true, 8), NB_V(4, 8, kLds, false, 4), NB_V(4, 8, kLds, true, 4),
    NB_V(4, 4, kLds, true, 4), NB_V(1, 8, kLds, false, 8), NB_V(1, 16, kLds, false, 8),
    NB_V(2, 16, kLds, false, 8), NB_V(2, 16, kLds, true, 8), NB_V(4, 8, kSmem, true, 4),
    NB_V(4, 8, kLds, true, 2), NB_V(4, 16, kSmem, true, 2), NB_V(2, 16, kLds, true, 4),
    NB_V(4, 16, kLds, true, 1), NB_V(8, 16, kLds, true, 1), NB_V(8, 8, kLds, true, 1),
    NB_V(4, 16, kLds, true, 8), NB_V(4, 8, kLds, true, 8), NB_V(2, 16, kLds, true, 16),
    NB_V(4, 16, kSmem, true, 4), NB_V(2, 16, kSmem, true, 16), NB_V(2, 16, kSmem, true, 4),
#endif
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

// Default choice by the number of bodies this launch owns.  Measured on MI355X at N = 65536
// (profiles/r01_variant_sweep*.txt): 4 bodies per lane, packed fp32, 16 waves per workgroup,
// j loop unrolled 4x (variant 4) is the fastest single-kernel shape at 256 workgroups; with fewer i-tiles than
// CUs the j range is additionally split over JS workgroups per i-tile so that >= ~256
// workgroups (4 waves per SIMD on every CU) are in flight.
constexpr int kAutoVariant = 4;   // ib4_w16 packed, j loop unrolled 4x
constexpr uint32_t kTargetBlocks = 256;  // one 16-wave workgroup per CU
constexpr uint32_t kMaxJSplit = 32;

}  // namespace

int naive_variant_count() { return kNumVariants; }
const char *naive_variant_name(int v) {
    return (v >= 0 && v < kNumVariants) ? kVariants[v].name : "?";
}

static uint32_t pick_jsplit(uint32_t blocks, uint32_t tiles, uint32_t waves, int forced) {
    const uint32_t max_js = tiles / waves ? tiles / waves : 1u;  // >= 1 tile per wave
    uint32_t js = 1;
    if (forced > 0)
        js = (uint32_t)forced;
    else if (blocks && blocks < kTargetBlocks)
        js = (kTargetBlocks + blocks - 1u) / blocks;
    if (js > max_js) js = max_js;
    if (js > kMaxJSplit) js = kMaxJSplit;
    return js ? js : 1u;
}

NaivePlan plan_naive(uint32_t n, uint32_t lo, uint32_t hi, int variant, int jsplit, bool two_phase) {
    NaivePlan p{};
    p.variant = (variant >= 0 && variant < kNumVariants) ? variant : kAutoVariant;
    const Variant &v = kVariants[p.variant];
    const uint32_t itile = 64u * (uint32_t)v.ib;
    const uint32_t n_local = hi > lo ? hi - lo : 0u;
    p.blocks = n_local ? (n_local + itile - 1u) / itile : 0u;
    p.n_tiles = (n + kJTile - 1u) / kJTile;
    p.lo_tile = lo / kJTile;
    p.local_tiles = (n_local + kJTile - 1u) / kJTile;
    p.two_phase = two_phase && n_local > 0 && p.local_tiles < p.n_tiles;
    if (p.two_phase) {
        p.js_local = pick_jsplit(p.blocks, p.local_tiles, (uint32_t)v.w, jsplit);
        p.js_remote = pick_jsplit(p.blocks, p.n_tiles - p.local_tiles, (uint32_t)v.w, jsplit);
        p.jsplit = p.js_local + p.js_remote;
    } else {
        p.jsplit = pick_jsplit(p.blocks, p.n_tiles, (uint32_t)v.w, jsplit);
        p.js_local = 0;
        p.js_remote = p.jsplit;
    }
    return p;
}

// phase: kPhaseAll = the whole step; kPhaseLocal = partial sums over the rank's own j tiles only;
// kPhaseRemote = partial sums over everybody else's tiles, then the finish kernel.
hipError_t launch_naive_step(const NaiveLaunch &a, hipStream_t stream) {
    if (a.hi <= a.lo) return hipSuccess;  // a rank that owns no bodies
    const NaivePlan p = plan_naive(a.n, a.lo, a.hi, a.variant, a.jsplit, a.phase != kPhaseAll);
    const Variant &v = kVariants[p.variant];
    const dim3 block(64u * (uint32_t)v.w);
    const uint32_t nl = a.hi - a.lo;
    if (a.phase == kPhaseAll) {
        if (p.jsplit > 1 && (!a.partial || a.partial_slices < p.jsplit)) return hipErrorInvalidValue;
        const TileWindow all{0u, p.n_tiles, ~0u, 0u};
        hipLaunchKernelGGL(v.fn, dim3(p.blocks, p.jsplit), block, 0, stream, a.posm_src, a.posm_dst,
                           a.vel, a.acc, p.jsplit > 1 ? a.partial : (float4 *)nullptr, a.partial_stride,
                           a.n, a.n_pad, a.lo, a.hi, a.g, a.e, a.dt, all);
        if (p.jsplit > 1)
            hipLaunchKernelGGL(naive_finish_kernel, dim3((nl + 255u) / 256u), dim3(256), 0, stream,
                               a.posm_src, a.posm_dst, a.vel, a.acc, a.partial, a.partial_stride,
                               p.jsplit, a.lo, a.hi, a.g, a.dt, a.peers);
        return hipGetLastError();
    }
    if (!p.two_phase || !a.partial || a.partial_slices < p.jsplit) return hipErrorInvalidValue;
    if (a.phase == kPhaseLocal) {
        const TileWindow own{p.lo_tile, p.local_tiles, ~0u, 0u};
        hipLaunchKernelGGL(v.fn, dim3(p.blocks, p.js_local), block, 0, stream, a.posm_src, a.posm_dst,
                           a.vel, a.acc, a.partial, a.partial_stride, a.n, a.n_pad, a.lo, a.hi, a.g,
                           a.e, a.dt, own);
    } else {
        const TileWindow rest{0u, p.n_tiles - p.local_tiles, p.lo_tile, p.local_tiles};
        hipLaunchKernelGGL(v.fn, dim3(p.blocks, p.js_remote), block, 0, stream, a.posm_src, a.posm_dst,
                           a.vel, a.acc, a.partial + (size_t)p.js_local * a.partial_stride,
                           a.partial_stride, a.n, a.n_pad, a.lo, a.hi, a.g, a.e, a.dt, rest);
        hipLaunchKernelGGL(naive_finish_kernel, dim3((nl + 255u) / 256u), dim3(256), 0, stream,
                           a.posm_src, a.posm_dst, a.vel, a.acc, a.partial, a.partial_stride, p.jsplit,
                           a.lo, a.hi, a.g, a.dt, a.peers);
    }
    return hipGetLastError();
}

hipError_t launch_aos_to_soa(const nb_particle *aos, float4 *posm, float4 *vel, float4 *acc,
                             uint32_t n, uint32_t lo, uint32_t hi, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(aos_to_soa_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, aos, posm,
                       vel, acc, n, lo, hi);
    return hipGetLastError();
}

hipError_t launch_soa_to_aos(const float4 *posm, const float4 *vel, const float4 *acc,
                             nb_particle *aos, uint32_t n, uint32_t lo, uint32_t hi,
                             hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(soa_to_aos_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, posm, vel,
                       acc, aos, n, lo, hi);
    return hipGetLastError();
}

// ---- one-process multi-GPU runner: does a peer store arrive the way the step relies on? -------------------
// The exchange of nb_group.cpp is "a kernel on device A stores into device B's memory through peer access; A
// records an event; B's stream waits for it; B's next kernel reads the bytes with plain loads".  These two
// kernels rehearse exactly that once, at create time, with one word per ordered pair of ranks, so that a
// platform where it does not hold (no peer mapping after all, a cache the wait does not flush) is reported
// instead of computing garbage.
__global__ void peer_check_store_kernel(PeerWords dst, uint32_t slot, uint32_t value) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < dst.n) dst.p[q][slot] = value;
}
__global__ void peer_check_read_kernel(const uint32_t *__restrict__ words, uint32_t world, uint32_t me, uint32_t tag,
                                       uint32_t *__restrict__ bad) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < world && r != me && words[r] != (tag | r)) atomicAdd(bad, 1u);
}

hipError_t launch_peer_check_store(const PeerWords &dst, uint32_t slot, uint32_t value, hipStream_t stream) {
    hipLaunchKernelGGL(peer_check_store_kernel, dim3(1), dim3(64), 0, stream, dst, slot, value);
    return hipGetLastError();
}
hipError_t launch_peer_check_read(const uint32_t *words, uint32_t world, uint32_t me, uint32_t tag, uint32_t *bad,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(peer_check_read_kernel, dim3(1), dim3(64), 0, stream, words, world, me, tag, bad);
    return hipGetLastError();
}

}  // namespace nb
