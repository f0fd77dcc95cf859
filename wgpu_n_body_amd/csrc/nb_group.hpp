// nb_group.hpp -- the one-process multi-GPU runner behind nb_runner_create_multi (all-pairs: peer
// stores from the finish kernel; Barnes-Hut: replicated tree, partitioned walk, peer stores -- or Morton
// domains, local trees and the LET exchange by peer stores).
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "nb_sim.hpp"

namespace nb {

class DeviceGroup {
   public:
    ~DeviceGroup();
    // bodies [r per, (r+1) per) on device_ids[r]; a device id may repeat (ranks sharing a GPU)
    // let_migrate_every < 0: Barnes-Hut as replicated tree; >= 0: Morton domains + LET exchange (0: bodies never
    // change their rank, k: the bodies that left their rank's key range are handed over every k-th step)
    static int create(std::unique_ptr<DeviceGroup> &out, const nb_sim_params &sp, const nb_add_params &add,
                      const nb_particle *particles, const int *device_ids, int n_devices, int let_migrate_every = -1);
    int step_n(int steps);  // enqueue on every rank, return when every rank has finished
    int read_particles(nb_particle *dst, size_t count);
    const nb_sim_params &params() const { return params_; }
    int world() const { return (int)ranks_.size(); }
    uint64_t step_num() const { return step_; }
    // measurement: timing events around every rank's kernels and waits (nb_runner_set_profiling / _rank_times)
    int set_profiling(bool on);
    int rank_times(float *kernel_ms, float *wait_ms, int n) const;

   private:
    DeviceGroup();
    struct Rank;
    struct Shared;
    void worker(int r);
    int create_let(const nb_sim_params &sp, const nb_add_params &add, const nb_particle *particles,
                   const int *device_ids, int world);
    template <typename Fail, typename Failed>
    void let_step(int r, uint64_t t, Fail &fail, Failed &failed);
    std::string first_error() const;
    int check_peer_stores();  // create time: peer stores arrive behind their events (or NB_ERR_UNSUPPORTED)
    std::vector<std::unique_ptr<Rank>> ranks_;
    std::unique_ptr<Shared> sh_;
    nb_sim_params params_{};
    uint64_t step_ = 0;
    bool tree_ = false;  // Barnes-Hut: replicated tree, partitioned walk, slices copied to the peers
    bool let_ = false;   // Barnes-Hut: Morton domains, local trees, LET records pushed to the peers
    int migrate_every_ = 0;
    uint32_t let_cap_ = 0, mig_cap_ = 0;
};

}  // namespace nb
