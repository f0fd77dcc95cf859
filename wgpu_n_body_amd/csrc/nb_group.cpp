// nb_group.cpp -- the step on several GPUs of ONE process, behind the C ABI (nb_runner_create_multi,
// nb_runner_create_multi_let): no Python, no torch, no collective library in the step loop.
//
// There is no reference counterpart: the reference owns one adapter (src/runners/
// offline_headless.rs:22-31).  What makes the path shard is in the shader itself: naive.wgsl's
// update is Jacobi-style -- body i reads only the PREVIOUS step's positions of every body
// (naive.wgsl:34) and writes only its own slot (naive.wgsl:68), and the two particle buffers
// ping-pong (naive.rs:113-132).  So rank r of `world` owns the contiguous body range
// [r per, (r+1) per), keeps a full copy of both position/mass buffers on its device, and the only
// exchange of a step is every rank's new float4{x,y,z,m} slice reaching every peer.
//
// Here that exchange is not a copy and not a collective: the kernel that finishes a rank's step
// (naive_finish_kernel) stores the new slice into its own next-step buffer AND, through peer
// access, into the same slots of every peer's next-step buffer -- one slice per point-to-point
// xGMI link, which is the shape of MI355X's fabric.  A step of rank r is then
//     own j tiles  (needs only r's own slice, which r wrote itself)
//     wait: event "step t-1 finished" of every peer   (their stores into r's buffer have landed)
//     other j tiles + finish (stores to self and peers)
//     record: event "step t finished" of r
// One host thread per rank enqueues it on the rank's own stream; the threads meet at one
// host barrier per step (an event must have been recorded before a peer enqueues its wait for
// it).  The caller stays single-threaded: nb_runner_step returns when every rank has finished.
//
// The same device id may be given several times (ranks sharing a GPU): that is how the path is
// tested on a one-GPU box, bit for bit the same code.
//
// Barnes-Hut through the same runner is the replicated-tree scheme of SURVEY 8(e) step 1: every
// rank holds the full state and builds the identical octree, walks only its range of the sorted
// bodies, and its new position / velocity / acceleration slices are copied into every peer's
// arrays (one kernel on the rank's stream, stores through peer access) once EVERY rank has
// finished the step -- a TreeSim's step reads and writes the same arrays, so the copies must not land while a peer still
// reads them: two events per rank and step ("step finished", "slices pushed").  Bit for bit the
// single TreeSim.  The scheme that also shards the build -- Morton domains + LET exchange,
// nb_runner_create_multi_let -- is hosted here too: see create_let / let_step below.
#include <atomic>
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <numeric>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "nb_common.hpp"
#include "nb_group.hpp"

namespace nb {

namespace {
// The once-per-step meeting of the rank threads: spin on a generation counter (the step of a rank
// is tens of microseconds; a condition variable's wake-up costs as much), yield when it drags on.
class HostBarrier {
   public:
    explicit HostBarrier(int n) : n_(n) {}
    void wait() {
        const uint64_t gen = gen_.load(std::memory_order_acquire);
        if (count_.fetch_add(1, std::memory_order_acq_rel) + 1 == n_) {
            count_.store(0, std::memory_order_relaxed);
            gen_.store(gen + 1, std::memory_order_release);
            return;
        }
        for (uint32_t spins = 0; gen_.load(std::memory_order_acquire) == gen; ++spins)
            if (spins > 2000u) std::this_thread::yield();
    }

   private:
    const int n_;
    std::atomic<int> count_{0};
    std::atomic<uint64_t> gen_{0};
};
}  // namespace

// Measurement (nb_runner_set_profiling): timing events on the rank's stream at the borders between "the
// rank's own kernels" and "waiting for a peer's event"; after the batch has drained, the time between two
// consecutive marks goes to the kind of the earlier one.  Off by default: nothing is recorded.
struct RankProf {
    enum Kind : uint8_t { kKernel = 0, kWait = 1, kEnd = 2 };
    std::vector<hipEvent_t> ev;
    std::vector<uint8_t> kind;
    size_t used = 0;
    float ms[2] = {0.f, 0.f};
    void mark(bool on, Kind k, hipStream_t stream) {
        if (!on) return;
        if (used == ev.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return;
            ev.push_back(e);
            kind.push_back(0);
        }
        if (hipEventRecord(ev[used], stream) != hipSuccess) return;
        kind[used++] = (uint8_t)k;
    }
    void collect() {  // (the stream has drained)
        ms[0] = ms[1] = 0.f;
        for (size_t i = 0; i + 1 < used; ++i) {
            float t = 0.f;
            if (kind[i] != kEnd && hipEventElapsedTime(&t, ev[i], ev[i + 1]) == hipSuccess) ms[kind[i]] += t;
        }
        used = 0;
    }
    void destroy() {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        ev.clear();
    }
};

struct DeviceGroup::Rank {
    std::unique_ptr<SimBase> sim;
    RankProf prof;
    NaiveSim *naive = nullptr;
    int device = 0;
    hipEvent_t done[2] = {nullptr, nullptr};
    hipEvent_t pushed[2] = {nullptr, nullptr};  // Barnes-Hut: "my slices / my LET records are in every peer's arrays"
    hipEvent_t meta[2] = {nullptr, nullptr};    // LET: "my bounds are in every peer's table"
    uint32_t active = 0;                        // LET: bodies this rank holds (changes with migration)
    std::thread th;
    int rc = NB_OK;
    std::string err;
};

struct DeviceGroup::Shared {
    std::mutex mu;
    std::condition_variable cv_cmd, cv_done;
    uint64_t cmd_seq = 0;  // bumped by the caller for every batch of steps
    int cmd_steps = 0;
    int finished = 0;
    bool quit = false;
    std::atomic<bool> failed{false};  // some rank hit an error: the others stop working but keep meeting
    std::atomic<bool> profiling{false};
    int prof_steps = 0;
    std::unique_ptr<HostBarrier> bar;
    std::vector<uint32_t> mig;  // LET migration: world x world leaver counts, row r written by rank r's thread
};

DeviceGroup::DeviceGroup() : sh_(new Shared()) {}

DeviceGroup::~DeviceGroup() {
    {
        std::lock_guard<std::mutex> lk(sh_->mu);
        sh_->quit = true;
    }
    sh_->cv_cmd.notify_all();
    for (auto &r : ranks_)
        if (r->th.joinable()) r->th.join();
    for (auto &r : ranks_) {
        (void)hipSetDevice(r->device);
        if (r->sim) (void)r->sim->wait();
        for (hipEvent_t e : r->done)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : r->pushed)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : r->meta)
            if (e) (void)hipEventDestroy(e);
        r->prof.destroy();
        r->sim.reset();
    }
}

// ---- Barnes-Hut with a sharded build: Morton domains + LET exchange --------------
// The protocol of include/nbody.h (NB_PHASE_LET_*), hosted here instead of in a process per GPU
// (wgpu_n_body_amd/sharded.py LetTreeSim, whose domain cut and capacities this follows line by line
// so that both hosts give the same bits).  Per step and rank:
//   META, my row of bounds stored into every peer's table         -> event "meta"
//   [all metas in]  BUILD: own octree in the global cube, the part of it every peer needs (LET)
//   [every peer has finished the previous walk]  my row of export counts + the exported records
//   stored straight into the peers' tables / import areas (peer access; the counts are read on the
//   device, the host never sees them)                                                -> event "pushed"
//   [all records in]  WALK own tree + imported trees, integrate                      -> event "done"
// No host read between migrations; a migration step (every migrate_every-th) reads the leaver
// counts on the host, as the Python host does, and pulls the leavers from the peers' send areas.
namespace {
uint64_t spread21(uint64_t v) {  // two zero bits after each of the 21 low bits
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

// sharded.py morton_domains(particles, world, slack = 0.02, with_owners = True)
void morton_domains(const nb_particle *p, size_t n, int world, std::vector<uint32_t> &order, std::vector<size_t> &cuts,
                    std::vector<unsigned long long> &splits, float &ref_bound) {
    float amax = 0.f;
    for (size_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) amax = std::max(amax, std::fabs(p[i].position[c]));
    const double bound = (double)(float)std::max((double)amax, 1e-30);
    std::vector<uint64_t> keys(n);
    for (size_t i = 0; i < n; ++i) {
        uint64_t q[3];
        for (int c = 0; c < 3; ++c) {
            double v = ((double)p[i].position[c] + bound) / (2.0 * bound) * 2097152.0;
            v = std::min(std::max(v, 0.0), 2097151.0);
            q[c] = (uint64_t)v;
        }
        keys[i] = spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
    }
    order.resize(n);
    std::iota(order.begin(), order.end(), 0u);
    {   // stable sort by key, on up to 8 threads: sorted chunks, then pairwise stable merges (0.6 s on one thread
        // at 4,194,304 bodies -- start-up only, but it is the caller who waits)
        auto less = [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; };
        const size_t chunks = n >= (1u << 18) ? std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency())) : 1;
        std::vector<size_t> edge(chunks + 1);
        for (size_t c = 0; c <= chunks; ++c) edge[c] = n * c / chunks;
        auto run = [&](auto &&fn, size_t count) {
            std::vector<std::thread> th;
            for (size_t c = 1; c < count; ++c) th.emplace_back(fn, c);
            fn((size_t)0);
            for (auto &t : th) t.join();
        };
        run([&](size_t c) { std::stable_sort(order.begin() + (long)edge[c], order.begin() + (long)edge[c + 1], less); }, chunks);
        for (size_t width = 1; width < chunks; width *= 2) {
            const size_t pairs = (chunks + 2 * width - 1) / (2 * width);
            run([&](size_t p) {
                const size_t lo = 2 * width * p, mid = std::min(lo + width, chunks), hi = std::min(lo + 2 * width, chunks);
                if (mid < hi)
                    std::inplace_merge(order.begin() + (long)edge[lo], order.begin() + (long)edge[mid],
                                       order.begin() + (long)edge[hi], less);
            }, pairs);
        }
    }
    std::vector<uint64_t> sk(n);
    for (size_t i = 0; i < n; ++i) sk[i] = keys[order[i]];
    cuts.assign(1, 0);
    const long long tol = std::max<long long>(1, (long long)(0.02 * (double)n / (double)std::max(world, 1)));
    for (int r = 1; r < world; ++r) {
        const long long c0 = (long long)((n * (size_t)r) / (size_t)world);
        long long best = c0;
        const long long lo = std::max(c0 - tol, (long long)cuts.back() + 1), hi = std::min(c0 + tol, (long long)n - 1);
        if (lo <= hi && n > 1) {
            for (int level = 1; level < 22; ++level) {  // coarsest first
                const int shift = 3 * (21 - level);
                long long found = -1, dist = 0;
                for (long long j = lo; j <= hi; ++j) {  // a border between bodies j - 1 and j
                    if ((sk[(size_t)j - 1] >> shift) != (sk[(size_t)j] >> shift)) {
                        const long long dd = j > c0 ? j - c0 : c0 - j;
                        if (found < 0 || dd < dist) {  // (numpy argmin: the first of equal distances)
                            found = j;
                            dist = dd;
                        }
                    }
                }
                if (found >= 0) {
                    best = found;
                    break;
                }
            }
        }
        cuts.push_back((size_t)std::max<long long>(best, (long long)cuts.back()));
    }
    cuts.push_back(n);
    splits.clear();
    for (int r = 1; r < world; ++r) splits.push_back(cuts[(size_t)r] < n ? sk[cuts[(size_t)r]] : (1ull << 63));
    ref_bound = std::max(amax, 1e-30f);
}
}  // namespace

int DeviceGroup::create_let(const nb_sim_params &sp, const nb_add_params &add, const nb_particle *particles,
                           const int *device_ids, int world) {
    const size_t n = sp.particle_num;
    std::vector<uint32_t> order;
    std::vector<size_t> cuts;
    std::vector<unsigned long long> splits;
    float ref_bound = 1.f;
    morton_domains(particles, n, world, order, cuts, splits, ref_bound);
    size_t most = 0;
    for (int r = 0; r < world; ++r) most = std::max(most, cuts[(size_t)r + 1] - cuts[(size_t)r]);
    const size_t capacity = (size_t)(1.25 * (double)most) + 4096;  // LetTreeSim.HEADROOM
    if (capacity > 0x7fffffffull) {
        set_error("LET: too many bodies per rank");
        return NB_ERR_INVALID;
    }
    let_cap_ = (uint32_t)(2 * capacity + 64);   // a peer can need at most this rank's whole octree
    mig_cap_ = (uint32_t)std::max<size_t>(1024, capacity / 8);  // leavers per destination per migration
    std::vector<nb_particle> padded(capacity);
    for (int r = 0; r < world; ++r) {
        std::unique_ptr<Rank> rk(new Rank());
        rk->device = device_ids[r];
        const size_t lo = cuts[(size_t)r], cnt = cuts[(size_t)r + 1] - lo;
        std::fill(padded.begin(), padded.end(), nb_particle{});
        for (size_t i = 0; i < cnt; ++i) padded[i] = particles[order[lo + i]];
        nb_sim_params spl = sp;
        spl.particle_num = (uint32_t)capacity;
        nb_placement pl{};
        pl.device_id = device_ids[r];
        pl.rank = 0;
        pl.world = 1;
        if (int rc = make_sim_impl(rk->sim, &spl, &add, &pl, padded.data(), capacity)) return rc;
        SimBase &sim = *rk->sim;
        if (int rc = sim.set_tuning("tree_let_world", world)) return rc;
        if (int rc = sim.set_tuning("tree_let_rank", r)) return rc;
        if (int rc = sim.set_tuning("tree_let_active", (int)cnt)) return rc;
        if (int rc = sim.set_tuning("tree_let_cap", (int)let_cap_)) return rc;
        const unsigned long long none = 0;  // (one rank: no border, but not a null pointer)
        if (int rc = sim.let_set_owners(splits.empty() ? &none : splits.data(), world, ref_bound, mig_cap_)) return rc;
        rk->active = (uint32_t)cnt;
        NB_HIP_TRY(hipSetDevice(rk->device));
        for (hipEvent_t &e : rk->done) NB_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t &e : rk->pushed) NB_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t &e : rk->meta) NB_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ranks_.push_back(std::move(rk));
    }
    sh_->mig.assign((size_t)world * (size_t)world, 0u);
    return NB_OK;
}

int DeviceGroup::create(std::unique_ptr<DeviceGroup> &out, const nb_sim_params &sp, const nb_add_params &add,
                       const nb_particle *particles, const int *device_ids, int n_devices, int let_migrate_every) {
    if (!device_ids || n_devices < 1 || n_devices > kMaxPeers + 1) {
        set_error("nb_runner_create_multi: between 1 and %d devices", kMaxPeers + 1);
        return NB_ERR_INVALID;
    }
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        set_error("no HIP device is visible (hipGetDeviceCount); there is no CPU fallback");
        return NB_ERR_NO_DEVICE;
    }
    std::unique_ptr<DeviceGroup> g(new (std::nothrow) DeviceGroup());
    if (!g) {
        set_error("out of host memory");
        return NB_ERR_ALLOC;
    }
    g->params_ = sp;
    g->tree_ = add.kind == NB_TREE_SIM_PARAMS;
    g->let_ = g->tree_ && let_migrate_every >= 0;
    g->migrate_every_ = let_migrate_every > 0 ? let_migrate_every : 0;
    const int world = n_devices;
    for (int r = 0; r < world; ++r) {
        if (device_ids[r] < 0 || device_ids[r] >= visible) {
            set_error("device_ids[%d] = %d out of range (%d devices)", r, device_ids[r], visible);
            return NB_ERR_INVALID;
        }
    }
    if (g->let_) {
        if (int rc = g->create_let(sp, add, particles, device_ids, world)) return rc;
    }
    for (int r = 0; r < world && !g->let_; ++r) {
        if (device_ids[r] < 0 || device_ids[r] >= visible) {
            set_error("device_ids[%d] = %d out of range (%d devices)", r, device_ids[r], visible);
            return NB_ERR_INVALID;
        }
        std::unique_ptr<Rank> rk(new Rank());
        rk->device = device_ids[r];
        nb_placement pl{};
        pl.device_id = device_ids[r];
        pl.rank = r;
        pl.world = world;
        if (int rc = make_sim_impl(rk->sim, &sp, &add, &pl, particles, sp.particle_num)) return rc;
        rk->naive = g->tree_ ? nullptr : static_cast<NaiveSim *>(rk->sim.get());
        NB_HIP_TRY(hipSetDevice(rk->device));
        for (hipEvent_t &e : rk->done) NB_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t &e : rk->pushed) NB_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t &e : rk->meta) NB_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        g->ranks_.push_back(std::move(rk));
    }
    // peer access between every pair of distinct devices, then hand every rank its peers' buffers
    for (int r = 0; r < world; ++r) {
        NB_HIP_TRY(hipSetDevice(g->ranks_[r]->device));
        for (int q = 0; q < world; ++q) {
            const int a = g->ranks_[r]->device, b = g->ranks_[q]->device;
            if (a == b) continue;
            int can = 0;
            NB_HIP_TRY(hipDeviceCanAccessPeer(&can, a, b));
            if (!can) {
                set_error("device %d cannot access device %d (no peer access): the one-process runner needs it", a, b);
                return NB_ERR_UNSUPPORTED;
            }
            const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) NB_HIP_TRY(e);
            (void)hipGetLastError();
        }
    }
    if (world > 1)
        if (int rc = g->check_peer_stores()) return rc;
    for (int r = 0; r < world && !g->tree_; ++r) {
        float4 *b0[kMaxPeers], *b1[kMaxPeers];
        int k = 0;
        for (int q = 0; q < world; ++q) {
            if (q == r) continue;
            float4 *bufs[2];
            g->ranks_[q]->naive->position_buffers(bufs);
            b0[k] = bufs[0];
            b1[k] = bufs[1];
            ++k;
        }
        if (int rc = g->ranks_[r]->naive->set_peers(b0, b1, k)) return rc;
    }
    g->sh_->bar.reset(new HostBarrier(world));
    for (int r = 0; r < world; ++r) g->ranks_[r]->th = std::thread(&DeviceGroup::worker, g.get(), r);
    out = std::move(g);
    return NB_OK;
}

// What every step of every scheme relies on, rehearsed once with one word per ordered pair of ranks: rank r's
// kernel stores (tag | r) into word r of every peer's table through peer access and records an event; every
// peer's stream waits for the events and a kernel of its own reads its table with plain loads.  A platform where
// that does not hold -- a peer mapping that is not there after all, a cache the event wait does not make
// coherent -- is reported here instead of producing wrong forces later.  (Ranks sharing a device run it too:
// the same code, minus the links.)
int DeviceGroup::check_peer_stores() {
    const int world = (int)ranks_.size();
    const uint32_t tag = 0xC0DE0000u;
    std::vector<uint32_t *> table((size_t)world, nullptr), bad((size_t)world, nullptr);
    std::vector<hipEvent_t> stored((size_t)world, nullptr);
    int rc = NB_OK;
    auto hip = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == NB_OK) {
            set_error("peer-store check: %s failed: %s", what, hipGetErrorString(e));
            rc = NB_ERR_HIP;
        }
        return e == hipSuccess;
    };
    for (int r = 0; r < world && rc == NB_OK; ++r) {
        hip(hipSetDevice(ranks_[r]->device), "hipSetDevice");
        hip(hipMalloc((void **)&table[r], sizeof(uint32_t) * ((size_t)world + 1)), "hipMalloc");
        bad[r] = table[r] ? table[r] + world : nullptr;
        if (table[r]) hip(hipMemsetAsync(table[r], 0, sizeof(uint32_t) * ((size_t)world + 1), ranks_[r]->sim->stream), "hipMemsetAsync");
        hip(hipEventCreateWithFlags(&stored[r], hipEventDisableTiming), "hipEventCreate");
        hip(hipStreamSynchronize(ranks_[r]->sim->stream), "hipStreamSynchronize");
    }
    for (int r = 0; r < world && rc == NB_OK; ++r) {  // the stores
        hip(hipSetDevice(ranks_[r]->device), "hipSetDevice");
        PeerWords dst{};
        for (int q = 0; q < world; ++q)
            if (q != r) dst.p[dst.n++] = table[q];
        hip(launch_peer_check_store(dst, (uint32_t)r, tag | (uint32_t)r, ranks_[r]->sim->stream), "store kernel");
        hip(hipEventRecord(stored[r], ranks_[r]->sim->stream), "hipEventRecord");
    }
    for (int q = 0; q < world && rc == NB_OK; ++q) {  // the reads, behind the events
        hip(hipSetDevice(ranks_[q]->device), "hipSetDevice");
        for (int r = 0; r < world; ++r)
            if (r != q) hip(hipStreamWaitEvent(ranks_[q]->sim->stream, stored[r], 0), "hipStreamWaitEvent");
        hip(launch_peer_check_read(table[q], (uint32_t)world, (uint32_t)q, tag, bad[q], ranks_[q]->sim->stream), "read kernel");
    }
    for (int q = 0; q < world && rc == NB_OK; ++q) {
        hip(hipSetDevice(ranks_[q]->device), "hipSetDevice");
        uint32_t nbad = 0;
        hip(hipMemcpyAsync(&nbad, bad[q], sizeof nbad, hipMemcpyDeviceToHost, ranks_[q]->sim->stream), "hipMemcpyAsync");
        hip(hipStreamSynchronize(ranks_[q]->sim->stream), "hipStreamSynchronize");
        if (rc == NB_OK && nbad) {
            set_error("peer-store check: %u of the %d words that peers stored into device %d's memory (rank %d) did not "
                      "arrive behind their events: this platform does not give the one-process runner the "
                      "visibility it needs (use one process per GPU: nb_placement + RCCL)",
                      nbad, world - 1, ranks_[q]->device, q);
            rc = NB_ERR_UNSUPPORTED;
        }
    }
    for (int r = 0; r < world; ++r) {
        (void)hipSetDevice(ranks_[r]->device);
        if (stored[r]) (void)hipEventDestroy(stored[r]);
        if (table[r]) (void)hipFree(table[r]);
    }
    return rc;
}

// One LET step of rank r (see create_let).  Every rank thread passes the same barriers whatever fails.
template <typename Fail, typename Failed>
void DeviceGroup::let_step(int r, uint64_t t, Fail &fail, Failed &failed) {
    Rank &me = *ranks_[r];
    SimBase &sim = *me.sim;
    const int world = (int)ranks_.size();
    const bool prof = sh_->profiling.load(std::memory_order_relaxed);
    auto mark_k = [&] { me.prof.mark(prof, RankProf::kKernel, sim.stream); };
    auto mark_w = [&] { me.prof.mark(prof, RankProf::kWait, sim.stream); };
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e == hipSuccess) return true;
        set_error("%s failed: %s", what, hipGetErrorString(e));
        fail(NB_ERR_HIP);
        return false;
    };
    auto region = [&](int q, int k, void **base, size_t *seg) {  // base (and per-rank length) of region k on rank q
        size_t off = 0, len = 0, total = 0;
        if (int rc = ranks_[q]->sim->exchange_region(k, base, &off, &len, &total)) {
            fail(rc);
            return false;
        }
        if (seg) *seg = len;
        return true;
    };
    auto wait_peers = [&](int which, uint64_t idx) {  // 0 meta, 1 pushed, 2 done
        mark_w();
        for (int q = 0; q < world; ++q) {
            if (q == r) continue;
            Rank &p = *ranks_[q];
            hipEvent_t ev = which == 0 ? p.meta[idx & 1] : which == 1 ? p.pushed[idx & 1] : p.done[idx & 1];
            if (!hip_ok(hipStreamWaitEvent(sim.stream, ev, 0), "hipStreamWaitEvent")) return;
        }
        mark_k();
    };
    auto peers_of = [&](int k, void **bases) {  // bases of region k on every peer, in rank order
        int np = 0;
        for (int q = 0; q < world; ++q) {
            if (q == r) continue;
            if (!region(q, k, &bases[np], nullptr)) return -1;
            ++np;
        }
        return np;
    };

    const bool migrate = migrate_every_ > 0 && t > 0 && t % (uint64_t)migrate_every_ == 0;
    mark_k();
    if (migrate) {
        // the bodies that left this rank's key range go to their new owners: leaver counts read on the
        // host (they size the next launches), leavers pulled from the peers' send areas
        std::vector<uint32_t> &mig = sh_->mig;
        if (!failed()) {
            void *base = nullptr;
            if (int rc = sim.encode_phase(NB_PHASE_LET_MIGRATE)) fail(rc);
            else if (hip_ok(hipStreamSynchronize(sim.stream), "hipStreamSynchronize") && region(r, 4, &base, nullptr))
                (void)hip_ok(hipMemcpy(&mig[(size_t)r * world], static_cast<uint32_t *>(base) + (size_t)r * world,
                                       sizeof(uint32_t) * (size_t)world, hipMemcpyDeviceToHost), "hipMemcpy");
        }
        sh_->bar->wait();  // every row of the table is in
        if (!failed()) {
            std::vector<uint32_t> recv((size_t)world, 0u);
            void *mine = nullptr;
            bool ok = region(r, 6, &mine, nullptr);
            size_t at = 0;
            for (int q = 0; q < world && ok; ++q) {
                if (q == r) continue;
                const uint32_t c = mig[(size_t)q * world + r];
                recv[(size_t)q] = c;
                if (c > mig_cap_) {
                    set_error("LET migration: %u leavers from rank %d for rank %d, the segment holds %u", c, q, r, mig_cap_);
                    fail(NB_ERR_INVALID);
                    ok = false;
                    break;
                }
                void *theirs = nullptr;
                size_t seg = 0;
                if (!region(q, 5, &theirs, &seg)) { ok = false; break; }
                if (c)
                    ok = hip_ok(hipMemcpyPeerAsync(static_cast<char *>(mine) + at * 48u, me.device,
                                                   static_cast<char *>(theirs) + (size_t)r * seg, ranks_[q]->device,
                                                   (size_t)c * 48u, sim.stream), "hipMemcpyPeerAsync");
                at += c;
            }
            if (ok) {
                const uint32_t stay = mig[(size_t)r * world + r];
                if (int rc = sim.let_set_arrivals(stay, recv.data(), world)) fail(rc);
                else me.active = stay + (uint32_t)at;
            }
        }
    }
    void *bases[kMaxPeers + 1];
    if (!failed()) {
        if (int rc = sim.encode_phase(NB_PHASE_LET_META)) fail(rc);
        else {
            const int np = peers_of(0, bases);
            if (np >= 0) {
                if (int rc = sim.push_region(0, bases, np)) fail(rc);
                else (void)hip_ok(hipEventRecord(me.meta[t & 1], sim.stream), "hipEventRecord");
            }
        }
    }
    sh_->bar->wait();  // every rank has recorded "my bounds are pushed"
    if (!failed()) {
        wait_peers(0, t);
        if (!failed())
            if (int rc = sim.encode_phase(NB_PHASE_LET_BUILD)) fail(rc);
        // the peers' tables and import areas are free once they have finished the previous walk
        if (!failed() && t > 0) wait_peers(2, t - 1);
        if (!failed()) {
            const int np = peers_of(1, bases);
            if (np >= 0)
                if (int rc = sim.push_region(1, bases, np)) fail(rc);
        }
        if (!failed()) {
            void *imports[kMaxPeers + 1] = {};
            bool ok = true;
            for (int q = 0; q < world && ok; ++q)
                if (q != r) ok = region(q, 3, &imports[q], nullptr);
            if (ok) {
                if (int rc = sim.let_push_segments(imports, world, let_cap_)) fail(rc);
                else (void)hip_ok(hipEventRecord(me.pushed[t & 1], sim.stream), "hipEventRecord");
            }
        }
    }
    sh_->bar->wait();  // every rank has recorded "my records are pushed"
    if (!failed()) {
        wait_peers(1, t);
        if (!failed()) {
            if (int rc = sim.let_set_import_stride(let_cap_)) fail(rc);
            else if (int rc2 = sim.encode_phase(NB_PHASE_LET_WALK)) fail(rc2);
            else (void)hip_ok(hipEventRecord(me.done[t & 1], sim.stream), "hipEventRecord");
        }
    }
}

// One rank's host thread: waits for a batch of steps, enqueues them, waits for its stream.
void DeviceGroup::worker(int r) {
    Rank &me = *ranks_[r];
    const int world = (int)ranks_.size();
    (void)hipSetDevice(me.device);
    uint64_t seen = 0;
    for (;;) {
        int steps = 0;
        {
            std::unique_lock<std::mutex> lk(sh_->mu);
            sh_->cv_cmd.wait(lk, [&] { return sh_->quit || sh_->cmd_seq != seen; });
            if (sh_->quit) return;
            seen = sh_->cmd_seq;
            steps = sh_->cmd_steps;
        }
        auto fail = [&](int rc) {
            if (me.rc == NB_OK) {
                me.rc = rc;
                me.err = nb_last_error();
            }
            sh_->failed.store(true, std::memory_order_release);
        };
        // (asked several times per step and rank: a relaxed load, not the group's mutex -- a rank that sees the
        // flag a step late only enqueues one more step of work that nobody reads)
        auto failed = [&] { return sh_->failed.load(std::memory_order_relaxed); };
        const bool prof = sh_->profiling.load(std::memory_order_relaxed);
        auto mark_k = [&] { me.prof.mark(prof, RankProf::kKernel, me.sim->stream); };
        auto mark_w = [&] { me.prof.mark(prof, RankProf::kWait, me.sim->stream); };
        // this rank's stream waits for every peer's event of step idx ("slices pushed" or "step finished")
        auto wait_all = [&](bool pushed_ev, uint64_t idx) -> hipError_t {
            hipError_t e = hipSuccess;
            for (int q = 0; q < world && e == hipSuccess; ++q)
                if (q != r)
                    e = hipStreamWaitEvent(me.sim->stream, pushed_ev ? ranks_[q]->pushed[idx & 1] : ranks_[q]->done[idx & 1], 0);
            return e;
        };
        for (int s = 0; s < steps && let_; ++s) let_step(r, step_ + (uint64_t)s, fail, failed);
        for (int s = 0; s < steps && tree_ && !let_; ++s) {
            // Barnes-Hut, replicated tree: [peers' slices of step t-1 are in] build + walk my range
            // [every rank has finished step t] copy my slices into every peer's arrays
            const uint64_t t = step_ + (uint64_t)s;
            sh_->bar->wait();  // every rank has recorded its "slices of step t-1 pushed"
            if (!failed()) {
                mark_w();
                hipError_t e = t > 0 ? wait_all(true, t - 1) : hipSuccess;
                mark_k();
                if (e != hipSuccess) {
                    set_error("hipStreamWaitEvent failed: %s", hipGetErrorString(e));
                    fail(NB_ERR_HIP);
                } else if (int rc = me.sim->encode()) {
                    fail(rc);
                } else if ((e = hipEventRecord(me.done[t & 1], me.sim->stream)) != hipSuccess) {
                    set_error("hipEventRecord failed: %s", hipGetErrorString(e));
                    fail(NB_ERR_HIP);
                }
            }
            sh_->bar->wait();  // every rank has recorded its "step t finished"
            if (!failed()) {
                mark_w();
                hipError_t e = wait_all(false, t);
                mark_k();
                // positions/masses, velocities, accelerations: one launch stores the slices into every peer
                void *bases[3 * kMaxPeers];
                int np = 0;
                for (int q = 0; q < world && !failed(); ++q) {
                    if (q == r) continue;
                    for (int k = 0; k < 3; ++k) {
                        size_t o2 = 0, l2 = 0, t2 = 0;
                        if (int rc = ranks_[q]->sim->exchange_region(k, &bases[3 * np + k], &o2, &l2, &t2)) fail(rc);
                    }
                    ++np;
                }
                if (!failed())
                    if (int rc = me.sim->push_exchange(bases, np)) fail(rc);
                if (e == hipSuccess) e = hipEventRecord(me.pushed[t & 1], me.sim->stream);
                if (e != hipSuccess) {
                    set_error("slice push failed: %s", hipGetErrorString(e));
                    fail(NB_ERR_HIP);
                }
            }
        }
        for (int s = 0; s < steps && !tree_; ++s) {
            const uint64_t t = step_ + (uint64_t)s;  // absolute step index (same on every rank)
            if (!failed()) {
                mark_k();
                if (int rc = me.sim->encode_phase(0)) fail(rc);  // own j tiles: nothing to wait for
            }
            sh_->bar->wait();  // every rank has recorded its "step t-1 finished"
            if (!failed()) {
                mark_w();
                hipError_t e = t > 0 ? wait_all(false, t - 1) : hipSuccess;
                mark_k();
                if (e != hipSuccess) {
                    set_error("hipStreamWaitEvent failed: %s", hipGetErrorString(e));
                    fail(NB_ERR_HIP);
                } else if (int rc = me.sim->encode_phase(1)) {  // other tiles, finish: stores to self + peers
                    fail(rc);
                } else if ((e = hipEventRecord(me.done[t & 1], me.sim->stream)) != hipSuccess) {
                    set_error("hipEventRecord failed: %s", hipGetErrorString(e));
                    fail(NB_ERR_HIP);
                }
            }
        }
        me.prof.mark(prof, RankProf::kEnd, me.sim->stream);
        if (int rc = me.sim->wait()) fail(rc);
        if (prof) me.prof.collect();
        sh_->bar->wait();  // every stream has drained: all slices have landed everywhere
        {
            std::lock_guard<std::mutex> lk(sh_->mu);
            sh_->finished += 1;
        }
        sh_->cv_done.notify_all();
    }
}

int DeviceGroup::step_n(int steps) {
    if (steps <= 0) return NB_OK;
    {
        std::lock_guard<std::mutex> lk(sh_->mu);
        if (sh_->failed.load()) {
            set_error("the runner is in a failed state: %s", first_error().c_str());
            return NB_ERR_INVALID;
        }
        sh_->cmd_steps = steps;
        sh_->finished = 0;
        sh_->cmd_seq += 1;
    }
    sh_->cv_cmd.notify_all();
    {
        std::unique_lock<std::mutex> lk(sh_->mu);
        sh_->cv_done.wait(lk, [&] { return sh_->finished == (int)ranks_.size(); });
    }
    step_ += (uint64_t)steps;
    for (auto &r : ranks_)
        if (r->rc != NB_OK) {
            set_error("rank on device %d: %s", r->device, r->err.c_str());
            return r->rc;
        }
    return NB_OK;
}

int DeviceGroup::set_profiling(bool on) {
    sh_->profiling.store(on);
    return NB_OK;
}

// per rank: milliseconds of the LAST batch of steps spent in the rank's own kernels and waiting (on the device)
// for the peers' events
int DeviceGroup::rank_times(float *kernel_ms, float *wait_ms, int n) const {
    for (int r = 0; r < n && r < (int)ranks_.size(); ++r) {
        if (kernel_ms) kernel_ms[r] = ranks_[(size_t)r]->prof.ms[RankProf::kKernel];
        if (wait_ms) wait_ms[r] = ranks_[(size_t)r]->prof.ms[RankProf::kWait];
    }
    return NB_OK;
}

std::string DeviceGroup::first_error() const {
    for (auto &r : ranks_)
        if (r->rc != NB_OK) return r->err;
    return "";
}

// All positions/masses are everywhere; velocities and accelerations live with their owner.
int DeviceGroup::read_particles(nb_particle *dst, size_t count) {
    const size_t n = params_.particle_num;
    if (count > n) {
        set_error("read_particles: asked for %zu of %zu particles", count, n);
        return NB_ERR_INVALID;
    }
    if (let_) {  // rank by rank, every rank's bodies in its current tree order (as LetTreeSim.read_particles)
        size_t at = 0;
        std::vector<nb_particle> tmp;
        for (auto &rk : ranks_) {
            tmp.resize(rk->active);
            if (rk->active)
                if (int rc = rk->sim->read_particles(tmp.data(), rk->active)) return rc;
            for (size_t i = 0; i < rk->active && at < count; ++i) dst[at++] = tmp[i];
        }
        if (at < count) {
            set_error("LET read_particles: the ranks hold %zu bodies, %zu asked for", at, count);
            return NB_ERR_INVALID;
        }
        return NB_OK;
    }
    if (tree_) return ranks_[0]->sim->read_particles(dst, count);  // replicated: every rank holds every body
    std::vector<nb_particle> tmp(n), all(n);
    for (size_t r = 0; r < ranks_.size(); ++r) {
        SimBase &s = *ranks_[r]->sim;
        if (int rc = s.read_particles(tmp.data(), n)) return rc;
        if (r == 0) all = tmp;
        for (size_t i = s.lo; i < s.hi; ++i) all[i] = tmp[i];
    }
    for (size_t i = 0; i < count; ++i) dst[i] = all[i];
    return NB_OK;
}

}  // namespace nb
