// nb_abi.cpp -- the C ABI of include/nbody.h: simulator objects (the reference's
// `trait Simulator` implementors NaiveSim / TreeSim) and the OfflineHeadless-shaped runner.
//
// Host side only; the kernels live in nb_naive.hip / nb_tree.hip.  Everything the
// reference does through wgpu (buffers, bind groups, command encoders, queue.submit,
// device.poll) maps to hipMalloc'd SoA buffers, a ping-pong index and one hipStream_t.
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "nb_common.hpp"
#include "nb_group.hpp"
#include "nb_sim.hpp"

namespace nb {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

static size_t round_up(size_t x, size_t m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------------------------------
// SimBase
// ------------------------------------------------------------------------------------------
SimBase::~SimBase() {
    if (own_stream && stream) (void)hipStreamDestroy(stream);
}

int SimBase::setup_common(const nb_sim_params &p, const nb_add_params &ap, const nb_placement *pl) {
    params = p;
    add = ap;
    nb_placement d{};
    d.device_id = 0;
    d.rank = 0;
    d.world = 1;
    place = pl ? *pl : d;
    if (place.world < 1 || place.rank < 0 || place.rank >= place.world) {
        set_error("invalid placement: rank %d of world %d", place.rank, place.world);
        return NB_ERR_INVALID;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        set_error("no HIP device is visible (hipGetDeviceCount); there is no CPU fallback");
        return NB_ERR_NO_DEVICE;
    }
    if (place.device_id < 0) place.device_id = 0;
    if (place.device_id >= count) {
        set_error("device_id %d out of range (%d devices)", place.device_id, count);
        return NB_ERR_INVALID;
    }
    NB_HIP_TRY(hipSetDevice(place.device_id));
    if (place.stream) {
        stream = static_cast<hipStream_t>(place.stream);
        own_stream = false;
    } else {
        NB_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        own_stream = true;
    }
    n = p.particle_num;
    per_rank = (uint32_t)nb_shard_bodies_per_rank(n, place.world);
    n_pad = (uint32_t)nb_shard_padded_bodies(n, place.world);
    const uint64_t lo64 = (uint64_t)per_rank * (uint64_t)place.rank;
    lo = (uint32_t)(lo64 < n ? lo64 : n);
    const uint64_t hi64 = lo64 + per_rank;
    hi = (uint32_t)(hi64 < n ? hi64 : n);
    return NB_OK;
}

int SimBase::bind_device() const {
    NB_HIP_TRY(hipSetDevice(place.device_id));
    return NB_OK;
}

int SimBase::wait() {
    if (int rc = bind_device()) return rc;
    NB_HIP_TRY(hipStreamSynchronize(stream));
    return NB_OK;
}

// ------------------------------------------------------------------------------------------
// NaiveSim -- src/sims/naive.rs
// ------------------------------------------------------------------------------------------
NaiveSim::~NaiveSim() {
    (void)hipSetDevice(place.device_id);
    if (own_posm) {
        if (posm[0]) (void)hipFree(posm[0]);
        if (posm[1]) (void)hipFree(posm[1]);
    }
    if (vel) (void)hipFree(vel);
    if (acc) (void)hipFree(acc);
    if (d_aos) (void)hipFree(d_aos);
    if (partial) (void)hipFree(partial);
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
}

// NaiveSim::new, src/sims/naive.rs:20-145: two particle buffers, both initialised with the
// same particles (:99-111); here: two position/mass buffers + one velocity + one
// acceleration array (a body's v and a are only ever touched by its own thread, so they
// need no ping-pong).
int NaiveSim::init(const nb_particle *host, size_t count) {
    if (count != n) {
        set_error("particle count %zu does not match sim_params.particle_num %u", count, n);
        return NB_ERR_INVALID;
    }
    const size_t posm_bytes = sizeof(float4) * (size_t)n_pad;
    if (place.posm[0] || place.posm[1]) {
        if (!place.posm[0] || !place.posm[1] || place.posm[0] == place.posm[1]) {
            set_error("placement.posm needs two distinct device buffers (or none)");
            return NB_ERR_INVALID;
        }
        posm[0] = static_cast<float4 *>(place.posm[0]);
        posm[1] = static_cast<float4 *>(place.posm[1]);
        own_posm = false;
    } else {
        NB_HIP_TRY(hipMalloc(&posm[0], posm_bytes));
        NB_HIP_TRY(hipMalloc(&posm[1], posm_bytes));
        own_posm = true;
    }
    const size_t local_bytes = sizeof(float4) * (size_t)(per_rank ? per_rank : kPadTo);
    NB_HIP_TRY(hipMalloc(&vel, local_bytes));
    NB_HIP_TRY(hipMalloc(&acc, local_bytes));
    NB_HIP_TRY(hipMalloc(&d_aos, sizeof(nb_particle) * (size_t)(n ? n : 1)));
    NB_HIP_TRY(hipMemsetAsync(posm[0], 0, posm_bytes, stream));
    NB_HIP_TRY(hipMemsetAsync(posm[1], 0, posm_bytes, stream));
    NB_HIP_TRY(hipMemsetAsync(vel, 0, local_bytes, stream));
    NB_HIP_TRY(hipMemsetAsync(acc, 0, local_bytes, stream));
    if (const char *v = getenv("NB_NAIVE_VARIANT")) variant = atoi(v);
    if (const char *v = getenv("NB_NAIVE_JSPLIT")) jsplit = atoi(v);
    if (int rc = ensure_workspace()) return rc;
    return write_particles(host, count);
}

// The j-split path needs room for its partial sums; (re)allocated whenever the plan changes.
int NaiveSim::ensure_workspace() {
    // room for whichever shape needs more slices: the single-launch or the two-phase step
    const NaivePlan p1 = plan_naive(n, lo, hi, variant, jsplit, false);
    const NaivePlan p2 = plan_naive(n, lo, hi, variant, jsplit, place.world > 1);
    NaivePlan p = p1;
    if (p2.two_phase && p2.jsplit > p.jsplit) p.jsplit = p2.jsplit;
    if ((p.jsplit > 1 || p2.two_phase) && p.jsplit > partial_slices) {
        if (int rc = bind_device()) return rc;
        NB_HIP_TRY(hipStreamSynchronize(stream));
        if (partial) NB_HIP_TRY(hipFree(partial));
        partial = nullptr;
        partial_slices = 0;
        NB_HIP_TRY(hipMalloc(&partial, sizeof(float4) * (size_t)p.jsplit * (size_t)per_rank));
        partial_slices = p.jsplit;
    }
    return NB_OK;
}

int NaiveSim::write_particles(const nb_particle *host, size_t count) {
    if (count != n) {
        set_error("write_particles: count %zu != particle_num %u", count, n);
        return NB_ERR_INVALID;
    }
    if (int rc = bind_device()) return rc;
    if (n == 0) return NB_OK;
    local_done = false;  // a first half enqueued for the old state is void
    NB_HIP_TRY(hipMemcpyAsync(d_aos, host, sizeof(nb_particle) * (size_t)n, hipMemcpyHostToDevice,
                              stream));
    NB_HIP_TRY(launch_aos_to_soa(d_aos, posm[cur], vel, acc, n, lo, hi, stream));
    // both buffers start equal, as in naive.rs:99-111
    NB_HIP_TRY(hipMemcpyAsync(posm[cur ^ 1], posm[cur], sizeof(float4) * (size_t)n_pad,
                              hipMemcpyDeviceToDevice, stream));
    NB_HIP_TRY(hipStreamSynchronize(stream));  // `host` may be freed by the caller on return
    return NB_OK;
}

int NaiveSim::launch(int phase) {
    NaiveLaunch a{};
    a.posm_src = posm[cur];
    a.posm_dst = posm[cur ^ 1];
    a.vel = vel;
    a.acc = acc;
    a.n = n;
    a.n_pad = n_pad;
    a.lo = lo;
    a.hi = hi;
    a.g = params.g;
    a.e = params.e;
    a.dt = params.dt;
    a.variant = variant;
    a.jsplit = jsplit;
    a.partial = partial;
    a.partial_stride = per_rank;
    a.partial_slices = partial_slices;
    a.phase = phase;
    a.peers = peers[cur ^ 1];  // the finish kernel writes the new slice there too
    if (a.peers.n && hi > lo && !(lo == 0 && hi == n)) {  // (a rank that owns no body, or all of them,
                                                           // has nothing its peers wait for)
        const NaivePlan p = plan_naive(n, lo, hi, variant, jsplit, phase != kPhaseAll);
        if (phase == kPhaseAll && p.jsplit <= 1) {
            set_error("a NaiveSim with peers steps in two phases (nb_sim_encode_phase)");
            return NB_ERR_INVALID;
        }
    }
    NB_HIP_TRY(launch_naive_step(a, stream));
    return NB_OK;
}

// NaiveSim::encode, src/sims/naive.rs:147-162: one dispatch, then flip the ping-pong.
int NaiveSim::encode() {
    if (int rc = bind_device()) return rc;
    if (local_done) return encode_phase(1);  // the step's first half is already enqueued
    if (int rc = launch(kPhaseAll)) return rc;
    cur ^= 1;
    step_num += 1;
    return NB_OK;
}

// The step in two halves, for overlapping the multi-GPU exchange with compute:
//   phase 0: partial sums over the rank's OWN j tiles.  They only need this rank's slice of the
//            current positions, so the caller may enqueue phase 0 of step k+1 right after step
//            k, while the all-gather of step k's other slices is still in flight;
//   phase 1: partial sums over all other j tiles (after the all-gather), then the finish
//            kernel (fixed-order sum + integrator) and the ping-pong flip.
int NaiveSim::encode_phase(int phase) {
    if (int rc = bind_device()) return rc;
    const NaivePlan p = plan_naive(n, lo, hi, variant, jsplit, true);
    if (!p.two_phase) {  // nothing to split (single rank, or a rank without bodies)
        if (phase == 0) return NB_OK;
        if (int rc = launch(kPhaseAll)) return rc;
        cur ^= 1;
        step_num += 1;
        return NB_OK;
    }
    if (phase == 0) {
        if (local_done) {
            set_error("encode_phase(0) called twice for one step");
            return NB_ERR_INVALID;
        }
        if (int rc = launch(kPhaseLocal)) return rc;
        local_done = true;
        return NB_OK;
    }
    if (phase != 1) {
        set_error("encode_phase: phase must be 0 or 1");
        return NB_ERR_INVALID;
    }
    if (!local_done)
        if (int rc = launch(kPhaseLocal)) return rc;
    if (int rc = launch(kPhaseRemote)) return rc;
    local_done = false;
    cur ^= 1;
    step_num += 1;
    return NB_OK;
}

int NaiveSim::encode_n_timed(int count, float *ms_total, float *ms_kernel) {
    if (count <= 0) {
        set_error("encode_n_timed: n must be positive");
        return NB_ERR_INVALID;
    }
    if (int rc = bind_device()) return rc;
    while (events.size() < (size_t)(2 * count + 2)) {
        hipEvent_t e;
        NB_HIP_TRY(hipEventCreate(&e));
        events.push_back(e);
    }
    NB_HIP_TRY(hipEventRecord(events[0], stream));
    for (int k = 0; k < count; ++k) {
        NB_HIP_TRY(hipEventRecord(events[2 + 2 * k], stream));
        if (int rc = encode()) return rc;
        NB_HIP_TRY(hipEventRecord(events[3 + 2 * k], stream));
    }
    NB_HIP_TRY(hipEventRecord(events[1], stream));
    NB_HIP_TRY(hipStreamSynchronize(stream));
    float total = 0.f, ksum = 0.f;
    NB_HIP_TRY(hipEventElapsedTime(&total, events[0], events[1]));
    for (int k = 0; k < count; ++k) {
        float ms = 0.f;
        NB_HIP_TRY(hipEventElapsedTime(&ms, events[2 + 2 * k], events[3 + 2 * k]));
        ksum += ms;
    }
    if (ms_total) *ms_total = total;
    if (ms_kernel) *ms_kernel = ksum / (float)count;
    return NB_OK;
}

int NaiveSim::read_particles(nb_particle *dst, size_t count) {
    if (count > n) {
        set_error("read_particles: asked for %zu of %u particles", count, n);
        return NB_ERR_INVALID;
    }
    if (int rc = bind_device()) return rc;
    if (count == 0) return wait();
    NB_HIP_TRY(launch_soa_to_aos(posm[cur], vel, acc, d_aos, n, lo, hi, stream));
    NB_HIP_TRY(hipMemcpyAsync(dst, d_aos, sizeof(nb_particle) * count, hipMemcpyDeviceToHost,
                              stream));
    NB_HIP_TRY(hipStreamSynchronize(stream));
    return NB_OK;
}

int NaiveSim::exchange_region(int index, void **dev_ptr, size_t *off, size_t *len, size_t *total) {
    if (index != 0) {
        set_error("exchange region %d out of range (NaiveSim has 1)", index);
        return NB_ERR_INVALID;
    }
    if (dev_ptr) *dev_ptr = posm[cur];
    if (off) *off = sizeof(float4) * (size_t)per_rank * (size_t)place.rank;
    if (len) *len = sizeof(float4) * (size_t)per_rank;
    if (total) *total = sizeof(float4) * (size_t)n_pad;
    return NB_OK;
}

int NaiveSim::set_peers(float4 *const *peer_buf0, float4 *const *peer_buf1, int count) {
    if (count < 0 || count > kMaxPeers) {
        set_error("at most %d peers", kMaxPeers);
        return NB_ERR_INVALID;
    }
    for (int k = 0; k < count; ++k) {
        peers[0].p[k] = peer_buf0[k];
        peers[1].p[k] = peer_buf1[k];
    }
    peers[0].n = peers[1].n = (uint32_t)count;
    return NB_OK;
}

int NaiveSim::set_tuning(const char *key, int value) {
    if (strcmp(key, "naive_variant") == 0) {
        if (value >= naive_variant_count()) {
            set_error("naive_variant %d out of range (%d variants)", value, naive_variant_count());
            return NB_ERR_INVALID;
        }
        variant = value;
        return ensure_workspace();
    }
    if (strcmp(key, "naive_jsplit") == 0) {
        if (value < 0 || value > 32) {
            set_error("naive_jsplit %d out of range [0, 32]", value);
            return NB_ERR_INVALID;
        }
        jsplit = value;
        return ensure_workspace();
    }
    set_error("unknown tuning key '%s'", key);
    return NB_ERR_INVALID;
}

int make_sim_impl(std::unique_ptr<SimBase> &out, const nb_sim_params *sp, const nb_add_params *ap,
                  const nb_placement *pl, const nb_particle *particles, size_t count) {
    nb_add_params add{NB_NAIVE_SIM_PARAMS, 0.f};
    if (ap) add = *ap;
    std::unique_ptr<SimBase> impl;
    if (add.kind == NB_NAIVE_SIM_PARAMS) {
        impl.reset(new (std::nothrow) NaiveSim());
    } else if (add.kind == NB_TREE_SIM_PARAMS) {
        impl.reset(make_tree_sim());
        if (!impl) {
            set_error("TreeSim (Barnes-Hut) is not available in this build");
            return NB_ERR_UNSUPPORTED;
        }
        if (!(add.theta > 0.f)) add.theta = NB_DEFAULT_THETA;  // tree.rs:42-51
        if (pl && (pl->posm[0] || pl->posm[1])) {
            set_error("TreeSim owns its buffers (placement.posm must be NULL)");
            return NB_ERR_INVALID;
        }
    } else {
        set_error("unknown add_params.kind %d", add.kind);
        return NB_ERR_INVALID;
    }
    if (!impl) {
        set_error("out of host memory");
        return NB_ERR_ALLOC;
    }
    if (int rc = impl->setup_common(*sp, add, pl)) return rc;
    if (int rc = impl->init(particles, count)) return rc;
    out = std::move(impl);
    return NB_OK;
}

static int make_sim(nb_sim **out, const nb_sim_params *sp, const nb_add_params *ap,
                    const nb_placement *pl, const nb_particle *particles, size_t count) {
    if (!out || !sp) {
        set_error("null argument");
        return NB_ERR_INVALID;
    }
    *out = nullptr;
    std::unique_ptr<SimBase> impl;
    if (int rc = make_sim_impl(impl, sp, ap, pl, particles, count)) return rc;
    nb_sim *s = new (std::nothrow) nb_sim();
    if (!s) {
        set_error("out of host memory");
        return NB_ERR_ALLOC;
    }
    s->impl = std::move(impl);
    *out = s;
    return NB_OK;
}

}  // namespace nb

// ==========================================================================================
// extern "C"
// ==========================================================================================
using namespace nb;

#define NB_GUARD(body)                                            \
    try {                                                         \
        body                                                      \
    } catch (const std::bad_alloc &) {                            \
        set_error("out of host memory");                          \
        return NB_ERR_ALLOC;                                      \
    } catch (...) {                                               \
        set_error("unexpected C++ exception at the ABI boundary"); \
        return NB_ERR_INVALID;                                    \
    }

extern "C" {

const char *nb_last_error(void) { return g_last_error.c_str(); }
#ifndef NB_SOURCE_HASH
#define NB_SOURCE_HASH "unknown"
#endif
// "... src:<hash>": sha256 prefix of the sources this binary was built from (build.py), so that a
// stale library next to newer sources is detectable (tests/test_abi.py)
const char *nb_version(void) { return "nbody_hip 0.2.0 gfx950 src:" NB_SOURCE_HASH; }

int nb_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

size_t nb_shard_bodies_per_rank(size_t particle_num, int world) {
    if (world < 1) world = 1;
    const size_t per = (particle_num + (size_t)world - 1) / (size_t)world;
    return round_up(per ? per : 1, kPadTo);
}

size_t nb_shard_padded_bodies(size_t particle_num, int world) {
    if (world < 1) world = 1;
    return nb_shard_bodies_per_rank(particle_num, world) * (size_t)world;
}

int nb_sim_create(nb_sim **out, const nb_sim_params *sim_params, const nb_add_params *add_params,
                  const nb_placement *placement, nb_init_fn init, void *user) {
    NB_GUARD({
        if (!sim_params || !init) {
            set_error("nb_sim_create: sim_params and init must be non-null");
            return NB_ERR_INVALID;
        }
        // init_fn(&sim_params) -> Vec<Particle>, naive.rs:97 / tree.rs:149
        std::vector<nb_particle> host(sim_params->particle_num);
        init(sim_params, host.data(), user);
        return make_sim(out, sim_params, add_params, placement, host.data(), host.size());
    })
}

int nb_sim_create_from_particles(nb_sim **out, const nb_sim_params *sim_params,
                                 const nb_add_params *add_params, const nb_placement *placement,
                                 const nb_particle *particles, size_t n) {
    NB_GUARD({
        if (!particles && n) {
            set_error("nb_sim_create_from_particles: particles is null");
            return NB_ERR_INVALID;
        }
        return make_sim(out, sim_params, add_params, placement, particles, n);
    })
}

#define NB_SIM_CALL(sim, expr)                \
    NB_GUARD({                                \
        if (!(sim) || !(sim)->impl) {         \
            set_error("null simulator");      \
            return NB_ERR_INVALID;            \
        }                                     \
        return (sim)->impl->expr;             \
    })

int nb_sim_encode(nb_sim *sim) { NB_SIM_CALL(sim, encode()) }
int nb_sim_encode_phase(nb_sim *sim, int phase) { NB_SIM_CALL(sim, encode_phase(phase)) }
int nb_sim_let_set_imports(nb_sim *sim, const uint32_t *counts, int world) {
    NB_SIM_CALL(sim, let_set_imports(counts, world))
}
int nb_sim_let_set_import_stride(nb_sim *sim, uint32_t stride) {
    NB_SIM_CALL(sim, let_set_import_stride(stride))
}
int nb_sim_let_set_owners(nb_sim *sim, const unsigned long long *splits, int world, float ref_bound,
                          uint32_t seg_cap) {
    NB_SIM_CALL(sim, let_set_owners(splits, world, ref_bound, seg_cap))
}
int nb_sim_let_set_arrivals(nb_sim *sim, uint32_t stay, const uint32_t *counts, int world) {
    NB_SIM_CALL(sim, let_set_arrivals(stay, counts, world))
}
int nb_sim_cleanup(nb_sim *sim) { NB_SIM_CALL(sim, cleanup()) }
int nb_sim_wait(nb_sim *sim) { NB_SIM_CALL(sim, wait()) }

int nb_sim_sim_params(const nb_sim *sim, nb_sim_params *out) {
    if (!sim || !sim->impl || !out) {
        set_error("null argument");
        return NB_ERR_INVALID;
    }
    *out = sim->impl->params;
    return NB_OK;
}

int nb_sim_read_particles(nb_sim *sim, nb_particle *dst, size_t n) {
    if (!dst && n) {
        set_error("read_particles: dst is null");
        return NB_ERR_INVALID;
    }
    NB_SIM_CALL(sim, read_particles(dst, n))
}

int nb_sim_write_particles(nb_sim *sim, const nb_particle *src, size_t n) {
    if (!src && n) {
        set_error("write_particles: src is null");
        return NB_ERR_INVALID;
    }
    NB_SIM_CALL(sim, write_particles(src, n))
}

int nb_sim_read_tree(nb_sim *sim, nb_octant *dst, size_t cap, size_t *n_nodes, float *root_width) {
    NB_SIM_CALL(sim, read_tree(dst, cap, n_nodes, root_width))
}

int nb_sim_exchange_region(nb_sim *sim, void **dev_ptr, size_t *offset_bytes, size_t *slice_bytes,
                           size_t *total_bytes) {
    NB_SIM_CALL(sim, exchange_region(0, dev_ptr, offset_bytes, slice_bytes, total_bytes))
}

int nb_sim_exchange_count(nb_sim *sim, int *count) {
    if (!sim || !sim->impl || !count) {
        set_error("null argument");
        return NB_ERR_INVALID;
    }
    *count = sim->impl->exchange_count();
    return NB_OK;
}

int nb_sim_exchange_region_i(nb_sim *sim, int index, void **dev_ptr, size_t *offset_bytes,
                             size_t *slice_bytes, size_t *total_bytes) {
    NB_SIM_CALL(sim, exchange_region(index, dev_ptr, offset_bytes, slice_bytes, total_bytes))
}

int nb_sim_step_num(const nb_sim *sim, uint64_t *out) {
    if (!sim || !sim->impl || !out) {
        set_error("null argument");
        return NB_ERR_INVALID;
    }
    *out = sim->impl->step_num;
    return NB_OK;
}

int nb_sim_encode_n_timed(nb_sim *sim, int n, float *ms_total, float *ms_kernel) {
    NB_SIM_CALL(sim, encode_n_timed(n, ms_total, ms_kernel))
}

int nb_sim_set_tuning(nb_sim *sim, const char *key, int value) {
    if (!key) {
        set_error("null key");
        return NB_ERR_INVALID;
    }
    NB_SIM_CALL(sim, set_tuning(key, value))
}

int nb_sim_debug_buffer(nb_sim *sim, const char *name, void *dst, size_t cap, size_t *bytes) {
    if (!name) {
        set_error("null name");
        return NB_ERR_INVALID;
    }
    NB_SIM_CALL(sim, debug_buffer(name, dst, cap, bytes))
}

int nb_naive_variant_count(void) { return naive_variant_count(); }
const char *nb_naive_variant_name(int v) { return naive_variant_name(v); }

int nb_sim_destroy(nb_sim *sim) {
    if (!sim) return NB_OK;
    if (sim->impl) (void)sim->impl->wait();
    delete sim;
    return NB_OK;
}

// ---- runner: OfflineHeadless<T>, src/runners/offline_headless.rs ---------------------------
struct nb_runner {
    nb_sim *sim = nullptr;                  // one device
    std::unique_ptr<DeviceGroup> group;      // several devices of this process (nb_runner_create_multi)
    bool profiling = false;                  // nb_runner_set_profiling, one device: events around step_n
    float kernel_ms = 0.f;
};

int nb_runner_create(nb_runner **out, const nb_sim_params *sim_params,
                     const nb_add_params *add_params, nb_init_fn init, void *user, int device_id) {
    NB_GUARD({
        if (!out) {
            set_error("null argument");
            return NB_ERR_INVALID;
        }
        *out = nullptr;
        nb_placement pl{};
        pl.device_id = device_id < 0 ? 0 : device_id;  // HighPerformance adapter, :23-30
        pl.rank = 0;
        pl.world = 1;
        nb_sim *sim = nullptr;
        if (int rc = nb_sim_create(&sim, sim_params, add_params, &pl, init, user)) return rc;
        nb_runner *r = new nb_runner();
        r->sim = sim;
        *out = r;
        return NB_OK;
    })
}

static int runner_create_group(nb_runner **out, const nb_sim_params *sim_params, const nb_add_params *add_params,
                               nb_init_fn init, void *user, const int *device_ids, int n_devices, int let_migrate_every);

int nb_runner_create_multi(nb_runner **out, const nb_sim_params *sim_params, const nb_add_params *add_params,
                           nb_init_fn init, void *user, const int *device_ids, int n_devices) {
    return runner_create_group(out, sim_params, add_params, init, user, device_ids, n_devices, -1);
}

int nb_runner_create_multi_let(nb_runner **out, const nb_sim_params *sim_params, const nb_add_params *add_params,
                               nb_init_fn init, void *user, const int *device_ids, int n_devices, int migrate_every) {
    if (!add_params || add_params->kind != NB_TREE_SIM_PARAMS || migrate_every < 0) {
        set_error("nb_runner_create_multi_let: TreeSimParams and migrate_every >= 0");
        return NB_ERR_INVALID;
    }
    return runner_create_group(out, sim_params, add_params, init, user, device_ids, n_devices, migrate_every);
}

static int runner_create_group(nb_runner **out, const nb_sim_params *sim_params, const nb_add_params *add_params,
                               nb_init_fn init, void *user, const int *device_ids, int n_devices, int let_migrate_every) {
    NB_GUARD({
        if (!out || !sim_params || !init || !device_ids || n_devices < 1) {
            set_error("nb_runner_create_multi: null argument or no device");
            return NB_ERR_INVALID;
        }
        *out = nullptr;
        if (n_devices == 1 && let_migrate_every < 0)
            return nb_runner_create(out, sim_params, add_params, init, user, device_ids[0]);
        nb_add_params add;
        add.kind = NB_NAIVE_SIM_PARAMS;
        add.theta = 0.f;
        if (add_params) add = *add_params;
        if (add.kind != NB_NAIVE_SIM_PARAMS && add.kind != NB_TREE_SIM_PARAMS) {
            set_error("unknown add_params.kind %d", add.kind);
            return NB_ERR_INVALID;
        }
        std::vector<nb_particle> host(sim_params->particle_num);
        init(sim_params, host.data(), user);  // init_fn(&sim_params) -> Vec<Particle>, once, on the host
        std::unique_ptr<DeviceGroup> g;
        if (int rc = DeviceGroup::create(g, *sim_params, add, host.data(), device_ids, n_devices, let_migrate_every))
            return rc;
        nb_runner *r = new nb_runner();
        r->group = std::move(g);
        *out = r;
        return NB_OK;
    })
}

// offline_headless.rs:38-44: encode -> submit -> cleanup -> poll(Wait)
int nb_runner_step(nb_runner *runner) {
    if (!runner || (!runner->sim && !runner->group)) {
        set_error("null runner");
        return NB_ERR_INVALID;
    }
    if (runner->group) NB_GUARD({ return runner->group->step_n(1); })
    if (int rc = nb_sim_encode(runner->sim)) return rc;
    if (int rc = nb_sim_cleanup(runner->sim)) return rc;
    return nb_sim_wait(runner->sim);
}

int nb_runner_step_n(nb_runner *runner, int n) {
    if (!runner || (!runner->sim && !runner->group)) {
        set_error("null runner");
        return NB_ERR_INVALID;
    }
    if (runner->group) NB_GUARD({ return runner->group->step_n(n); })
    hipEvent_t ev[2] = {nullptr, nullptr};
    const bool prof = runner->profiling && runner->sim->impl && runner->sim->impl->bind_device() == NB_OK &&
                      hipEventCreate(&ev[0]) == hipSuccess && hipEventCreate(&ev[1]) == hipSuccess;
    if (prof) (void)hipEventRecord(ev[0], runner->sim->impl->stream);
    int rc = NB_OK;
    for (int k = 0; k < n && rc == NB_OK; ++k) {
        rc = nb_sim_encode(runner->sim);
        if (rc == NB_OK) rc = nb_sim_cleanup(runner->sim);
    }
    if (prof) (void)hipEventRecord(ev[1], runner->sim->impl->stream);
    const int rw = nb_sim_wait(runner->sim);
    if (prof && hipEventElapsedTime(&runner->kernel_ms, ev[0], ev[1]) != hipSuccess) runner->kernel_ms = 0.f;
    for (hipEvent_t e : ev)
        if (e) (void)hipEventDestroy(e);
    return rc != NB_OK ? rc : rw;
}

int nb_runner_set_profiling(nb_runner *runner, int on) {
    if (!runner || (!runner->sim && !runner->group)) {
        set_error("null runner");
        return NB_ERR_INVALID;
    }
    runner->profiling = on != 0;
    if (runner->group) return runner->group->set_profiling(on != 0);
    return NB_OK;
}

int nb_runner_rank_times(nb_runner *runner, float *kernel_ms, float *wait_ms, int n) {
    if (!runner || (!runner->sim && !runner->group) || n < 1) {
        set_error("null runner");
        return NB_ERR_INVALID;
    }
    for (int r = 0; r < n; ++r) {
        if (kernel_ms) kernel_ms[r] = 0.f;
        if (wait_ms) wait_ms[r] = 0.f;
    }
    if (runner->group) return runner->group->rank_times(kernel_ms, wait_ms, n);
    if (kernel_ms) kernel_ms[0] = runner->kernel_ms;
    return NB_OK;
}

int nb_runner_read_particles(nb_runner *runner, nb_particle *dst, size_t n) {
    if (!runner) {
        set_error("null runner");
        return NB_ERR_INVALID;
    }
    if (runner->group) {
        if (!dst && n) {
            set_error("read_particles: dst is null");
            return NB_ERR_INVALID;
        }
        NB_GUARD({ return runner->group->read_particles(dst, n); })
    }
    return nb_sim_read_particles(runner->sim, dst, n);
}

int nb_runner_sim_params(const nb_runner *runner, nb_sim_params *out) {
    if (!runner || !out) {
        set_error("null runner");
        return NB_ERR_INVALID;
    }
    if (runner->group) {
        *out = runner->group->params();
        return NB_OK;
    }
    return nb_sim_sim_params(runner->sim, out);
}

int nb_runner_step_num(const nb_runner *runner, uint64_t *out) {
    if (!runner || !out) {
        set_error("null runner");
        return NB_ERR_INVALID;
    }
    if (runner->group) {
        *out = runner->group->step_num();
        return NB_OK;
    }
    return nb_sim_step_num(runner->sim, out);
}

nb_sim *nb_runner_sim(nb_runner *runner) { return runner ? runner->sim : nullptr; }

int nb_runner_destroy(nb_runner *runner) {
    if (!runner) return NB_OK;
    runner->group.reset();
    nb_sim_destroy(runner->sim);
    delete runner;
    return NB_OK;
}

}  // extern "C"
