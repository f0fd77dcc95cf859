// simulator.hpp -- header-only C++17 host mirror of the reference's simulation API, over the
// C ABI of include/nbody.h.  The reference is a Rust crate and this image has no Rust
// toolchain, so the host side a Rust user would see is written here in C++ with the same
// names, argument meaning and error behaviour:
//
//   sims::SimParams / AddParams / Particle   (src/sims/mod.rs:9-23,51-71)  -> nbody::SimParams ...
//   trait sims::Simulator                    (src/sims/mod.rs:73-90)       -> nbody::Simulator
//   sims::NaiveSim, sims::TreeSim            (src/sims/mod.rs:4-5)         -> nbody::NaiveSim, TreeSim
//   runners::OfflineHeadless<T>              (src/runners/offline_headless.rs) -> nbody::OfflineHeadless<T>
//   inits::{uniform,disc,spherical}_init     (src/inits.rs)                -> nbody::inits::*
//
// Constructors return anyhow::Result in the reference; here they throw nbody::Error
// (status code + nb_last_error() text).  step() panics on failure in the reference
// (src/sims/tree.rs:278-280); here it throws.
#pragma once

#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "nbody.h"

namespace nbody {

using Particle = nb_particle;    // 40 B
using SimParams = nb_sim_params;  // 16 B
using Octant = nb_octant;        // 52 B

inline SimParams default_sim_params() {  // SimParams::default(), sims/mod.rs:62-71
    return SimParams{NB_DEFAULT_PARTICLE_NUM, NB_DEFAULT_G, NB_DEFAULT_E, NB_DEFAULT_DT};
}

struct AddParams {  // enum AddParams, sims/mod.rs:18-23
    nb_add_params c{NB_NAIVE_SIM_PARAMS, 0.0f};
    static AddParams NaiveSimParams() { return AddParams{}; }
    static AddParams TreeSimParams(float theta) {
        AddParams a;
        a.c = nb_add_params{NB_TREE_SIM_PARAMS, theta};
        return a;
    }
};

class Error : public std::runtime_error {
   public:
    Error(int code, const std::string &what) : std::runtime_error(what), code_(code) {}
    int code() const { return code_; }

   private:
    int code_;
};

inline void check(int rc) {
    if (rc != NB_OK) throw Error(rc, std::string("nbody_hip: ") + nb_last_error());
}

// init_fn: fn(&SimParams) -> Vec<Particle>, sims/mod.rs:79
using InitFn = std::function<std::vector<Particle>(const SimParams &)>;

namespace inits {  // src/inits.rs, seeded
inline InitFn seeded(void (*fn)(const nb_sim_params *, nb_particle *, void *), uint64_t seed) {
    return [fn, seed](const SimParams &p) {
        std::vector<Particle> out(p.particle_num);
        uint64_t s = seed;
        fn(&p, out.data(), &s);
        return out;
    };
}
inline InitFn uniform_init(uint64_t seed = 0) { return seeded(nb_init_uniform, seed); }
inline InitFn disc_init(uint64_t seed = 0) { return seeded(nb_init_disc, seed); }
inline InitFn spherical_init(uint64_t seed = 0) { return seeded(nb_init_spherical, seed); }
}  // namespace inits

namespace detail {
struct InitThunk {
    const InitFn *fn;
    std::string error;
    static void call(const nb_sim_params *p, nb_particle *out, void *user) {
        auto *self = static_cast<InitThunk *>(user);
        try {  // never unwind through the C ABI
            std::vector<Particle> v = (*self->fn)(*p);
            if (v.size() != p->particle_num) {
                self->error = "init_fn returned the wrong number of particles";
                return;
            }
            for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
        } catch (const std::exception &e) {
            self->error = e.what();
        } catch (...) {
            self->error = "init_fn threw";
        }
    }
};
}  // namespace detail

// trait Simulator, sims/mod.rs:73-90
class Simulator {
   public:
    Simulator(const Simulator &) = delete;
    Simulator &operator=(const Simulator &) = delete;
    Simulator(Simulator &&o) noexcept : h_(o.h_), owned_(o.owned_) { o.h_ = nullptr; }
    virtual ~Simulator() {
        if (h_ && owned_) nb_sim_destroy(h_);
    }

    void encode() { check(nb_sim_encode(h_)); }    // Simulator::encode + queue.submit
    void cleanup() { check(nb_sim_cleanup(h_)); }  // Simulator::cleanup
    void wait() { check(nb_sim_wait(h_)); }        // device.poll(Maintain::Wait)
    SimParams sim_params() const {                 // Simulator::sim_params
        SimParams p{};
        check(nb_sim_sim_params(h_, &p));
        return p;
    }
    // Simulator::dest_particle_slice -- the POST-step state, copied to the host
    std::vector<Particle> dest_particle_slice() {
        std::vector<Particle> out(sim_params().particle_num);
        check(nb_sim_read_particles(h_, out.data(), out.size()));
        return out;
    }
    void write_particles(const std::vector<Particle> &p) {
        check(nb_sim_write_particles(h_, p.data(), p.size()));
    }
    uint64_t step_num() const {
        uint64_t v = 0;
        check(nb_sim_step_num(h_, &v));
        return v;
    }
    nb_sim *handle() { return h_; }

   protected:
    Simulator(nb_sim *h, bool owned) : h_(h), owned_(owned) {}
    static nb_sim *create(const SimParams &sp, nb_add_params ap, const InitFn &init,
                          const nb_placement *pl) {
        detail::InitThunk thunk{&init, {}};
        nb_sim *h = nullptr;
        int rc = nb_sim_create(&h, &sp, &ap, pl, &detail::InitThunk::call, &thunk);
        if (!thunk.error.empty()) {
            if (rc == NB_OK) nb_sim_destroy(h);
            throw Error(NB_ERR_INVALID, thunk.error);
        }
        check(rc);
        return h;
    }
    nb_sim *h_;
    bool owned_;
    template <class T>
    friend class OfflineHeadless;
};

class NaiveSim : public Simulator {  // sims/naive.rs
   public:
    static constexpr int kKind = NB_NAIVE_SIM_PARAMS;
    NaiveSim(const SimParams &sp, const AddParams &, const InitFn &init,
             const nb_placement *pl = nullptr)
        : Simulator(create(sp, nb_add_params{NB_NAIVE_SIM_PARAMS, 0.f}, init, pl), true) {}
    NaiveSim(nb_sim *borrowed) : Simulator(borrowed, false) {}
};

class TreeSim : public Simulator {  // sims/tree.rs
   public:
    static constexpr int kKind = NB_TREE_SIM_PARAMS;
    // any other AddParams falls back to theta 0.75 as TreeSim::new does (tree.rs:42-51)
    TreeSim(const SimParams &sp, const AddParams &ap, const InitFn &init,
            const nb_placement *pl = nullptr)
        : Simulator(create(sp,
                           nb_add_params{NB_TREE_SIM_PARAMS,
                                         ap.c.kind == NB_TREE_SIM_PARAMS ? ap.c.theta : 0.f},
                           init, pl),
                    true) {}
    TreeSim(nb_sim *borrowed) : Simulator(borrowed, false) {}
    std::pair<std::vector<Octant>, float> read_tree() {
        std::vector<Octant> t((size_t)4 * sim_params().particle_num + 8);
        size_t n = 0;
        float rw = 0.f;
        check(nb_sim_read_tree(h_, t.data(), t.size(), &n, &rw));
        t.resize(n);
        return {std::move(t), rw};
    }
};

// OfflineHeadless<T: Simulator>, runners/offline_headless.rs:4-45
template <class T>
class OfflineHeadless {
   public:
    OfflineHeadless(const SimParams &sp, const AddParams &ap, const InitFn &init, int device_id = -1) {
        detail::InitThunk thunk{&init, {}};
        nb_add_params c = ap.c;
        if (c.kind != T::kKind) c = nb_add_params{T::kKind, 0.f};
        int rc = nb_runner_create(&r_, &sp, &c, &detail::InitThunk::call, &thunk, device_id);
        if (!thunk.error.empty()) {
            if (rc == NB_OK) nb_runner_destroy(r_);
            throw Error(NB_ERR_INVALID, thunk.error);
        }
        check(rc);
    }
    // several GPUs of this process (nb_runner_create_multi): rank r owns a contiguous
    // body range on device_ids[r]
    // let_migrate_every >= 0 (TreeSim): Morton domains + LET exchange (nb_runner_create_multi_let)
    OfflineHeadless(const SimParams &sp, const AddParams &ap, const InitFn &init, const std::vector<int> &device_ids,
                    int let_migrate_every = -1) {
        detail::InitThunk thunk{&init, {}};
        nb_add_params c = ap.c;
        if (c.kind != T::kKind) c = nb_add_params{T::kKind, 0.f};
        int rc = let_migrate_every >= 0
                     ? nb_runner_create_multi_let(&r_, &sp, &c, &detail::InitThunk::call, &thunk, device_ids.data(),
                                                  (int)device_ids.size(), let_migrate_every)
                     : nb_runner_create_multi(&r_, &sp, &c, &detail::InitThunk::call, &thunk, device_ids.data(),
                                              (int)device_ids.size());
        if (!thunk.error.empty()) {
            if (rc == NB_OK) nb_runner_destroy(r_);
            throw Error(NB_ERR_INVALID, thunk.error);
        }
        check(rc);
    }
    OfflineHeadless(const OfflineHeadless &) = delete;
    ~OfflineHeadless() {
        if (r_) nb_runner_destroy(r_);
    }
    void step() { check(nb_runner_step(r_)); }  // encode -> submit -> cleanup -> poll(Wait)
    void step_n(int n) { check(nb_runner_step_n(r_, n)); }
    T sim() { return T(nb_runner_sim(r_)); }  // borrowed view of the runner's simulator (one device)
    uint64_t step_num() const {
        uint64_t v = 0;
        check(nb_runner_step_num(r_, &v));
        return v;
    }
    std::vector<Particle> read_particles() {
        SimParams p{};
        check(nb_runner_sim_params(r_, &p));
        std::vector<Particle> out(p.particle_num);
        check(nb_runner_read_particles(r_, out.data(), out.size()));
        return out;
    }

   private:
    nb_runner *r_ = nullptr;
};

}  // namespace nbody
