// nb_sim.hpp -- internal simulator classes behind the opaque nb_sim handle.
#pragma once

#include <memory>
#include <vector>

#include "nb_common.hpp"

namespace nb {

// What `trait Simulator` (src/sims/mod.rs:73-90) requires of an implementor, in HIP terms.
class SimBase {
   public:
    virtual ~SimBase();

    int setup_common(const nb_sim_params &p, const nb_add_params &ap, const nb_placement *pl);
    int bind_device() const;
    virtual int wait();  // device.poll(Wait); a TreeSim also reports its device status words

    virtual int init(const nb_particle *host, size_t count) = 0;          // Simulator::new
    virtual int encode() = 0;                                             // Simulator::encode
    virtual int encode_phase(int) {
        set_error("this simulator has no two-phase step");
        return NB_ERR_UNSUPPORTED;
    }
    virtual int cleanup() { return NB_OK; }                               // Simulator::cleanup
    virtual int read_particles(nb_particle *dst, size_t count) = 0;       // dest_particle_slice
    virtual int write_particles(const nb_particle *src, size_t count) = 0;
    virtual int encode_n_timed(int count, float *ms_total, float *ms_kernel) = 0;
    virtual int exchange_count() { return 0; }
    virtual int exchange_region(int, void **, size_t *, size_t *, size_t *) {
        set_error("this simulator has no exchange region");
        return NB_ERR_UNSUPPORTED;
    }
    // one-process runner: store this rank's slice of every exchange region into the same place of
    // every peer's arrays, in one launch on this simulator's stream (peer_bases[q * exchange_count()
    // + k] = base of region k on peer q)
    virtual int push_exchange(void *const *, int) {
        set_error("this simulator has no peer push");
        return NB_ERR_UNSUPPORTED;
    }
    // ... and one small region k (this rank's [off, off + len) of it) into the same place of every peer's
    // region k (peer_bases[q] = base of region k on peer q): the all-gathers of the LET protocol
    virtual int push_region(int, void *const *, int) {
        set_error("this simulator has no peer push");
        return NB_ERR_UNSUPPORTED;
    }
    // LET: the records exported for every peer q (region 2, segment q; as many as the export counted, read
    // on the device) into peer q's import area (import_bases[q] = base of region 3 on rank q; [me] unused),
    // at the segment the fixed-stride layout of nb_sim_let_set_import_stride(stride) gives this rank
    virtual int let_push_segments(void *const *, int, uint32_t) {
        set_error("let_push_segments: not a TreeSim");
        return NB_ERR_UNSUPPORTED;
    }
    virtual int let_set_imports(const uint32_t *, int) {
        set_error("let_set_imports: not a TreeSim");
        return NB_ERR_UNSUPPORTED;
    }
    virtual int let_set_import_stride(uint32_t) {
        set_error("let_set_import_stride: not a TreeSim");
        return NB_ERR_UNSUPPORTED;
    }
    virtual int let_set_owners(const unsigned long long *, int, float, uint32_t) {
        set_error("let_set_owners: not a TreeSim");
        return NB_ERR_UNSUPPORTED;
    }
    virtual int let_set_arrivals(uint32_t, const uint32_t *, int) {
        set_error("let_set_arrivals: not a TreeSim");
        return NB_ERR_UNSUPPORTED;
    }
    virtual int read_tree(nb_octant *, size_t, size_t *, float *) {
        set_error("read_tree: not a TreeSim");
        return NB_ERR_UNSUPPORTED;
    }
    virtual int debug_buffer(const char *name, void *, size_t, size_t *) {
        set_error("unknown debug buffer '%s'", name);
        return NB_ERR_INVALID;
    }
    virtual int set_tuning(const char *key, int) {
        set_error("unknown tuning key '%s'", key);
        return NB_ERR_INVALID;
    }

    nb_sim_params params{};
    nb_add_params add{};
    nb_placement place{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint64_t step_num = 0;
    uint32_t n = 0, n_pad = 0, per_rank = 0, lo = 0, hi = 0;
};

class NaiveSim final : public SimBase {
   public:
    ~NaiveSim() override;
    int init(const nb_particle *host, size_t count) override;
    int encode() override;
    int encode_phase(int phase) override;
    int read_particles(nb_particle *dst, size_t count) override;
    int write_particles(const nb_particle *src, size_t count) override;
    int encode_n_timed(int count, float *ms_total, float *ms_kernel) override;
    int exchange_count() override { return 1; }
    int exchange_region(int index, void **dev_ptr, size_t *off, size_t *len, size_t *total) override;
    int set_tuning(const char *key, int value) override;
    // one-process multi-GPU (nb_group.cpp): this simulator's two position buffers, and the peers'
    void position_buffers(float4 *out[2]) const {
        out[0] = posm[0];
        out[1] = posm[1];
    }
    int set_peers(float4 *const *peer_buf0, float4 *const *peer_buf1, int count);

   private:
    PeerDst peers[2] = {};                 // peers[b]: the peers' buffers with ping-pong index b
    float4 *posm[2] = {nullptr, nullptr};  // ping-pong position/mass (naive.rs:99-132)
    bool own_posm = true;
    float4 *vel = nullptr, *acc = nullptr;  // this rank's bodies only
    nb_particle *d_aos = nullptr;           // AoS staging for the 40-byte boundary layout
    int cur = 0;                            // posm[cur] holds the current state
    int variant = -1, jsplit = 0;           // tuning overrides (<0 / 0 = automatic)
    float4 *partial = nullptr;              // j-split partial sums [slices][per_rank]
    uint32_t partial_slices = 0;
    int ensure_workspace();
    int launch(int phase);
    bool local_done = false;                // phase 0 of the NEXT step already enqueued
    std::vector<hipEvent_t> events;
};

// Implemented in nb_tree.hip; returns nullptr when the tree path is not built.
SimBase *make_tree_sim();

// nb_abi.cpp: construct a simulator object (what nb_sim_create does behind the handle)
int make_sim_impl(std::unique_ptr<SimBase> &out, const nb_sim_params *sp, const nb_add_params *ap,
                  const nb_placement *pl, const nb_particle *particles, size_t count);

}  // namespace nb

struct nb_sim {
    std::unique_ptr<nb::SimBase> impl;
};
