/*
 * nbody.h -- C ABI of the MI355X-native N-body engine (libnbody_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of arpan-dhatt/wgpu-n-body: the
 * per-step force accumulation + kick-drift-kick integrator that the reference
 * runs as WGSL compute shaders behind `trait Simulator`, driven by
 * `OfflineHeadless<T>`.  Every entry point below cites the reference interface
 * it replaces (paths are relative to the reference crate root).
 *
 * Conventions
 *   - plain C, POD structs, plain pointers and sizes; no C++/torch types.
 *   - every function returning `int` returns NB_OK (0) on success or an
 *     nb_status code; a human-readable message for the calling thread's last
 *     failure is available from nb_last_error().  Nothing unwinds across the
 *     ABI (the reference's constructors return anyhow::Result,
 *     src/sims/mod.rs:80; its step() panics, src/sims/tree.rs:278-280 -- here
 *     both become status codes).
 *   - single caller thread per handle, not re-entrant (same as the reference:
 *     `Simulator` has no Send/Sync bound, src/sims/mod.rs:73-90).
 *   - there is NO CPU fallback: if no HIP device is usable, create() fails with
 *     NB_ERR_NO_DEVICE.
 */
#ifndef NBODY_H_
#define NBODY_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* Data model (byte-exact mirrors of the reference's #[repr(C)] PODs)         */
/* ------------------------------------------------------------------------- */

/* `struct Particle`, src/sims/mod.rs:9-16 (WGSL mirror naive.wgsl:1-6,
 * array stride 40, naive.wgsl:15-17).  40 bytes, 10 x f32. */
typedef struct nb_particle {
    float position[3];
    float velocity[3];
    float acceleration[3]; /* stored quantity is sum(f)*dt, naive.wgsl:41 */
    float mass;
} nb_particle;

/* `struct SimParams`, src/sims/mod.rs:51-58.  16 bytes. */
typedef struct nb_sim_params {
    uint32_t particle_num;
    float g;
    float e;
    float dt;
} nb_sim_params;

/* `SimParams::default()`, src/sims/mod.rs:62-71 */
#define NB_DEFAULT_PARTICLE_NUM 10000u
#define NB_DEFAULT_G 0.000001f
#define NB_DEFAULT_E 0.0001f
#define NB_DEFAULT_DT 0.016f
/* default theta when a TreeSim gets no TreeSimParams, src/sims/tree.rs:42-51 */
#define NB_DEFAULT_THETA 0.75f
/* `PARTICLES_PER_GROUP`, src/sims/mod.rs:7 (the reference's workgroup size;
 * kept for API parity -- the HIP kernels choose their own tiling). */
#define NB_PARTICLES_PER_GROUP 64u

/* `enum AddParams`, src/sims/mod.rs:18-23 */
typedef enum nb_add_kind {
    NB_NAIVE_SIM_PARAMS = 0, /* AddParams::NaiveSimParams            */
    NB_TREE_SIM_PARAMS = 1   /* AddParams::TreeSimParams { theta }   */
} nb_add_kind;

typedef struct nb_add_params {
    int32_t kind; /* nb_add_kind */
    float theta;  /* only read when kind == NB_TREE_SIM_PARAMS; <= 0 -> default */
} nb_add_params;

/* `struct Octant`, src/sims/tree.rs:605-622 (WGSL mirror tree.wgsl:1-6,
 * stride 52, tree.wgsl:31-33).  52 bytes.  Returned by nb_sim_read_tree. */
typedef struct nb_octant {
    float cog[3];
    float mass;
    uint32_t bodies;
    uint32_t children[8]; /* 0 = no child; for a leaf (bodies==1) children[0] is
                             the body's index in the step's source order */
} nb_octant;

/* The init callback: `init_fn: fn(&SimParams) -> Vec<Particle>`,
 * src/sims/mod.rs:79 / src/runners/offline_headless.rs:20.  The callee fills
 * out[0 .. params->particle_num).  `user` is an opaque cookie (the Rust fn
 * pointer has no environment; a C callback needs one). */
typedef void (*nb_init_fn)(const nb_sim_params *params, nb_particle *out, void *user);

typedef enum nb_status {
    NB_OK = 0,
    NB_ERR_INVALID = 1,    /* bad argument */
    NB_ERR_NO_DEVICE = 2,  /* no usable HIP device */
    NB_ERR_HIP = 3,        /* a HIP runtime call failed */
    NB_ERR_ALLOC = 4,      /* host or device allocation failed */
    NB_ERR_UNSUPPORTED = 5 /* valid request this build does not implement */
} nb_status;

/* Message describing the calling thread's most recent failing call ("" if none). */
const char *nb_last_error(void);
/* "nbody_hip <semver> gfx950" */
const char *nb_version(void);
/* Number of visible HIP devices (0 when there is none or the runtime fails). */
int nb_device_count(void);

/* ------------------------------------------------------------------------- */
/* Inits -- seeded equivalents of src/inits.rs                                */
/* ------------------------------------------------------------------------- */
/* The reference draws from rand::thread_rng() (src/inits.rs:7,30,58), which
 * is OS-seeded and not reproducible; these reproduce the *distributions* with
 * a counter-based generator specified bit-exactly in DESIGN.md ("RNG").
 * All three have the nb_init_fn signature.  `user` is NULL (seed 0) or points
 * to a uint64_t seed. */
void nb_init_uniform(const nb_sim_params *params, nb_particle *out, void *user);   /* inits.rs:6-27  */
void nb_init_disc(const nb_sim_params *params, nb_particle *out, void *user);      /* inits.rs:29-54 */
void nb_init_spherical(const nb_sim_params *params, nb_particle *out, void *user); /* inits.rs:56-83 */

/* ------------------------------------------------------------------------- */
/* Simulator -- `trait Simulator`, src/sims/mod.rs:73-90                      */
/* ------------------------------------------------------------------------- */
typedef struct nb_sim nb_sim;

/* Where a simulator lives and which bodies it owns.
 *
 * Single GPU: { device_id, 0, 1, NULL, {NULL,NULL} }.
 *
 * Multi GPU (one process per GPU): rank r of `world` owns the contiguous body
 * range [r*per, min(N,(r+1)*per)) with per = nb_shard_bodies_per_rank(N, world).
 * Each step writes only that slice of the new position/mass buffer; the caller
 * all-gathers the slices (RCCL) before the next nb_sim_encode.  There is no
 * reference counterpart (single adapter, src/runners/offline_headless.rs:22-31).
 *
 * `stream`: a hipStream_t the kernels are enqueued on, or NULL to let the
 * simulator create its own.  `posm[0..1]`: optional caller-owned device
 * buffers for the two ping-pong position/mass arrays (float4 x
 * nb_shard_padded_bodies(N, world) each, 16-byte aligned) -- this is how a host
 * that owns device memory (e.g. torch + torch.distributed) runs the collective
 * in place; NULL lets the simulator allocate them. */
typedef struct nb_placement {
    int32_t device_id;
    int32_t rank;
    int32_t world;
    void *stream;
    void *posm[2];
} nb_placement;

/* Bodies per rank (a multiple of the kernels' i-tile) and the padded length of
 * the position/mass buffers (= world * per_rank >= N). */
size_t nb_shard_bodies_per_rank(size_t particle_num, int world);
size_t nb_shard_padded_bodies(size_t particle_num, int world);

/* `Simulator::new(device, sim_params, add_params, mappable_primary_buffers,
 * init_fn)`, src/sims/mod.rs:74-82; NaiveSim::new src/sims/naive.rs:20-145,
 * TreeSim::new src/sims/tree.rs:29-260.  `device` becomes nb_placement;
 * `mappable_primary_buffers` has no HIP meaning and is dropped.  add_params
 * selects the implementation: NB_NAIVE_SIM_PARAMS -> all-pairs (NaiveSim),
 * NB_TREE_SIM_PARAMS -> Barnes-Hut (TreeSim).  placement may be NULL
 * (device 0, single GPU).  init runs on the host exactly once, for all
 * particle_num bodies (every rank must supply identical data). */
int nb_sim_create(nb_sim **out, const nb_sim_params *sim_params, const nb_add_params *add_params,
                  const nb_placement *placement, nb_init_fn init, void *user);

/* Same, from an existing particle array instead of a callback
 * (snapshot/restore; also how tests feed identical bytes to the
 * oracle and the GPU). */
int nb_sim_create_from_particles(nb_sim **out, const nb_sim_params *sim_params,
                                 const nb_add_params *add_params, const nb_placement *placement,
                                 const nb_particle *particles, size_t n);

/* `Simulator::encode(&mut self, device, queue) -> CommandEncoder`
 * (src/sims/mod.rs:83; NaiveSim::encode src/sims/naive.rs:147-162,
 * TreeSim::encode src/sims/tree.rs:262-353) fused with the runner's
 * `queue.submit` (src/runners/offline_headless.rs:40): enqueues ONE step on
 * the simulator's stream and returns without waiting; flips the ping-pong
 * (naive.rs:156,160). */
int nb_sim_encode(nb_sim *sim);

/* The same step in two halves, so that a multi-GPU caller can overlap the exchange of step k
 * with the beginning of step k+1.
 * TreeSim: phase 0 = bound, keys, sort, reorder of positions, octree build (needs positions and
 * masses only); phase 1 = reorder of velocities/accelerations, walk + integrate.  A sharded host
 * gathers positions first, enqueues phase 0, and lets the other two gathers run beside it.
 * NaiveSim (world > 1; otherwise phase 0 is a no-op and phase 1 is nb_sim_encode):
 *   phase 0 -- interactions with the rank's OWN bodies.  Needs only this rank's slice of the
 *              current positions, so it may be enqueued right after the previous step, while
 *              the all-gather of the other slices is still in flight;
 *   phase 1 -- interactions with everybody else's bodies (enqueue after the exchange has been
 *              ordered on the stream), then the integrator; flips the ping-pong.
 * nb_sim_encode after a phase 0 completes that step (= phase 1). */
int nb_sim_encode_phase(nb_sim *sim, int phase);

/* Multi-GPU Barnes-Hut with locally essential trees (LET; no reference counterpart -- the
 * reference has one device).  Every rank creates a TreeSim over ITS OWN bodies (placement world 1),
 * then sets the tuning keys "tree_let_world", "tree_let_rank" and "tree_let_cap" (records a peer
 * may receive from this rank; allocates the buffers).  One step is three phases with an exchange
 * after the first two (regions are those of nb_sim_exchange_region_i):
 *   NB_PHASE_LET_META   local bound + bounding box of the drifted bodies -> region 0;
 *                       caller: all-gather region 0 in place (32 B per rank)
 *   NB_PHASE_LET_BUILD  global root cube, octree of the rank's bodies, and for every peer the part
 *                       of that octree the peer's box can reach (32-byte records, children
 *                       contiguous, links relative to the segment) -> region 2, counts -> region 1;
 *                       caller: all-gather region 1 (`world` u32 per rank), read it, move
 *                       counts[r][me] records of rank r's segment `me` into region 3 packed in rank
 *                       order, then nb_sim_let_set_imports(counts received from each rank)
 *   NB_PHASE_LET_WALK   walk own tree + imported trees, integrate.
 * Walking an imported tree gives bit for bit what walking the peer's whole octree would give. */
#define NB_PHASE_LET_META 2
#define NB_PHASE_LET_BUILD 3
#define NB_PHASE_LET_WALK 4
int nb_sim_let_set_imports(nb_sim *sim, const uint32_t *counts, int world);
/* The same hand-over without a host round trip: the caller moves a FIXED number of records per peer
 * -- `stride` records of rank r's segment `me` to record offset j * stride of region 3, j = r's
 * position in rank order with `me` skipped -- sized from an earlier step's counts plus a margin, and
 * calls this instead of nb_sim_let_set_imports.  The counts themselves stay on the device: region 1,
 * all-gathered by the caller, is read by the kernel that prepares the imports.  A peer that has more
 * than `stride` records for this rank is reported like a too small tree_let_cap (nb_sim_wait). */
int nb_sim_let_set_import_stride(nb_sim *sim, uint32_t stride);

/* Migration between LET steps.  A TreeSim created with more bodies than it starts with (the
 * surplus is headroom; tuning key "tree_let_active" = bodies in use) can hand over the bodies
 * that left its domain: rank r owns the Morton keys [splits[r-1], splits[r]) of positions
 * quantised to 21 bits per axis in the fixed cube [-ref_bound, ref_bound]^3 (splits: world-1
 * values).  NB_PHASE_LET_MIGRATE compacts the stayers and packs the leavers per owner (region 5,
 * 48 B per body, at most seg_cap per owner) with the counts in region 4 (stayers at [rank]); the
 * caller all-gathers region 4, moves the segments into region 6 packed in rank order and calls
 * nb_sim_let_set_arrivals(stayers, arrivals per rank); nb_sim_sim_params then reports the new
 * body count. */
#define NB_PHASE_LET_MIGRATE 5
/* Optional, between NB_PHASE_LET_BUILD and NB_PHASE_LET_WALK: walk the rank's own octree (it needs
 * nothing from the peers) while the exported trees are still being exchanged; NB_PHASE_LET_WALK
 * then adds the imported trees and integrates.  Same sums in the same order: bit-identical. */
#define NB_PHASE_LET_WALK_OWN 6
int nb_sim_let_set_owners(nb_sim *sim, const unsigned long long *splits, int world, float ref_bound,
                          uint32_t seg_cap);
int nb_sim_let_set_arrivals(nb_sim *sim, uint32_t stay, const uint32_t *counts, int world);

/* `Simulator::cleanup(&mut self)`, src/sims/mod.rs:87-89 (TreeSim resets its
 * arena, src/sims/tree.rs:363-365).  Host-side housekeeping that may overlap
 * the enqueued step.  No-op for the all-pairs simulator. */
int nb_sim_cleanup(nb_sim *sim);

/* `device.poll(wgpu::Maintain::Wait)`, src/runners/offline_headless.rs:43:
 * block until everything enqueued on the simulator's stream has finished. */
int nb_sim_wait(nb_sim *sim);

/* `Simulator::sim_params(&self) -> SimParams`, src/sims/mod.rs:85. */
int nb_sim_sim_params(const nb_sim *sim, nb_sim_params *out);

/* `Simulator::dest_particle_slice(&self)`, src/sims/mod.rs:84 -- with one
 * documented deviation: the reference's slice is the buffer the last step READ
 * (src/sims/naive.rs:164-166 after step_num += 1); this returns
 * the POST-step state.  Waits for the stream, converts the device SoA state
 * to the 40-byte AoS layout and copies dst[0 .. n).  On a sharded simulator
 * only position/mass are globally valid (after the caller's all-gather);
 * velocity/acceleration are filled for the rank's own range and zero
 * elsewhere. */
int nb_sim_read_particles(nb_sim *sim, nb_particle *dst, size_t n);

/* Overwrite the simulator state (checkpoint restore). */
int nb_sim_write_particles(nb_sim *sim, const nb_particle *src, size_t n);

/* TreeSim only: copy out the octree the LAST step built (node count via
 * *n_nodes; up to cap entries written), in the reference's node numbering
 * (allocation order of src/sims/tree.rs:461,517-519).  NB_ERR_UNSUPPORTED on
 * an all-pairs simulator. */
int nb_sim_read_tree(nb_sim *sim, nb_octant *dst, size_t cap, size_t *n_nodes, float *root_width);

/* Sharded use: the device pointer of the position/mass buffer the LAST encode
 * wrote (float4 per body: x, y, z, mass), this rank's byte offset and byte
 * length inside it, and the total length.  The caller all-gathers
 * [offset, offset+slice) of every rank in place. */
int nb_sim_exchange_region(nb_sim *sim, void **dev_ptr, size_t *offset_bytes, size_t *slice_bytes,
                           size_t *total_bytes);

/* Same, for simulators with several regions to exchange.  NaiveSim has one (positions/masses).
 * A sharded TreeSim (replicated tree, partitioned walk: every rank builds the identical octree
 * from the full state and walks only its range of the sorted bodies) has three: the new
 * positions/masses, velocities and accelerations of its range -- the next step re-sorts all
 * bodies, so all three must reach every rank.  index in [0, count). */
int nb_sim_exchange_count(nb_sim *sim, int *count);
int nb_sim_exchange_region_i(nb_sim *sim, int index, void **dev_ptr, size_t *offset_bytes,
                             size_t *slice_bytes, size_t *total_bytes);

/* Step counter (`step_num`, src/sims/naive.rs:160). */
int nb_sim_step_num(const nb_sim *sim, uint64_t *out);

/* Time the next `n` encodes with HIP events on the simulator's own stream:
 * enqueue n steps back to back, wait, and report the total in *ms_total and
 * the mean duration of the dominant force kernel in *ms_kernel (events
 * bracket that launch alone).  Used by bench.py for `roofline.achieved`. */
int nb_sim_encode_n_timed(nb_sim *sim, int n, float *ms_total, float *ms_kernel);

/* Tuning knobs with no reference counterpart.  key "naive_variant": index into the
 * all-pairs kernel variant table (tiling / packing choices of nb_naive.hip; every variant
 * computes the same step).  Also settable through the NB_NAIVE_VARIANT environment
 * variable at create time.  A TreeSim's keys ("tree_*": bodies per wave of the walk, sort
 * passes and fix-up, who gathers the velocities, where the tile scan runs, ...) are listed
 * with their defaults in TreeSim::set_tuning (nb_tree.hip): speed only -- every setting
 * computes the same step, bit for bit where the tests say so. */
int nb_sim_set_tuning(nb_sim *sim, const char *key, int value);
int nb_naive_variant_count(void);
const char *nb_naive_variant_name(int variant);

/* Testing hook: copy a named internal device buffer to the host (e.g. TreeSim "order": the
 * source index of the body at each sorted position, u32 x N; "counters": walk visit/accept
 * counts, u64 x 4; "status": u32 x 4).  *bytes receives the buffer's length. */
int nb_sim_debug_buffer(nb_sim *sim, const char *name, void *dst, size_t cap, size_t *bytes);

int nb_sim_destroy(nb_sim *sim);

/* ------------------------------------------------------------------------- */
/* Runner -- `OfflineHeadless<T>`, src/runners/offline_headless.rs:4-45       */
/* ------------------------------------------------------------------------- */
typedef struct nb_runner nb_runner;

/* `OfflineHeadless::<T>::new(sim_params, add_params, init_fn)`,
 * offline_headless.rs:17-35: acquires the device (here: device_id, or -1 for
 * "highest-performance adapter" = device 0) and constructs the simulator T
 * selected by add_params.kind. */
int nb_runner_create(nb_runner **out, const nb_sim_params *sim_params,
                     const nb_add_params *add_params, nb_init_fn init, void *user, int device_id);

/* The same constructor over SEVERAL GPUs of this process (no reference counterpart: the reference
 * owns one adapter, offline_headless.rs:22-31; this is SURVEY 8(b)'s `device_ids, n_devices` form).
 * All-pairs: rank r owns the contiguous body range [r per, (r+1) per), per =
 * nb_shard_bodies_per_rank(N, n_devices), on device_ids[r]; every device keeps both position/mass
 * buffers in full, and the kernel that finishes a rank's step stores the rank's new
 * float4{x,y,z,m} slice into every peer's next-step buffer through peer access (one slice per
 * xGMI link), ordered by one HIP event per rank and step -- no host copy, no collective library.
 * Barnes-Hut: replicated tree, partitioned walk -- every device holds the full
 * state and builds the identical octree, walks its range of the sorted bodies, and copies its new
 * position / velocity / acceleration slices into every peer's arrays (one kernel on its stream, stores through peer access,
 * two events per rank and step); bit for bit the single TreeSim.  (Morton domains + LET exchange,
 * which also shards the build: nb_runner_create_multi_let below, or one process per GPU through
 * nb_placement and NB_PHASE_LET_*.)
 * One host thread per rank inside the library; the caller stays single-threaded and every call
 * below is synchronous as on one device.  A device id may repeat (ranks sharing a GPU).  n_devices
 * == 1 is nb_runner_create. */
int nb_runner_create_multi(nb_runner **out, const nb_sim_params *sim_params,
                           const nb_add_params *add_params, nb_init_fn init, void *user,
                           const int *device_ids, int n_devices);

/* Barnes-Hut over several GPUs of this process with the BUILD sharded too: the
 * bodies are cut into n_devices Morton-range domains at start-up; per step every rank builds the
 * octree of its own bodies inside the global root cube, exports to every peer the part of that tree
 * the peer's bodies can need (its locally essential tree, pruned decision-exactly) and walks its own
 * tree plus the imported ones -- the protocol of NB_PHASE_LET_* below, hosted inside the library: the
 * bounds, the export counts and the exported records are stored straight into the peers' tables and
 * import areas through peer access (the counts are consumed on the device), ordered by three events
 * per rank and step; no host read between migrations.  migrate_every = k > 0: every k-th step the
 * bodies that left their rank's key range are handed to their new owner (one host read of the
 * leaver counts on those steps); 0: never.  What a body feels is the sum of per-domain Barnes-Hut
 * walks, each with the reference's per-body acceptance test (tree.wgsl:57-70): within the walk's
 * own error of the one-tree result, not bit-equal to it.  nb_runner_read_particles returns the
 * bodies rank by rank, each rank's in its current tree order.  add_params must be TreeSimParams. */
int nb_runner_create_multi_let(nb_runner **out, const nb_sim_params *sim_params,
                               const nb_add_params *add_params, nb_init_fn init, void *user,
                               const int *device_ids, int n_devices, int migrate_every);

/* `OfflineHeadless::step(&mut self)`, offline_headless.rs:38-44:
 * encode -> submit -> cleanup -> blocking wait. */
int nb_runner_step(nb_runner *runner);

/* n steps enqueued back to back, one wait at the end (benchmark use). */
int nb_runner_step_n(nb_runner *runner, int n);

/* Measurement (no reference counterpart).  With profiling on, every rank of a several-GPU runner records
 * timing events on its stream at the borders between its own kernels and its waits for the peers' events;
 * after nb_runner_step_n, nb_runner_rank_times gives, per rank, the milliseconds of that batch of steps spent in
 * the rank's kernels (kernel_ms[r]) and waiting on the device for peers (wait_ms[r]).  n = the ranks the
 * arrays hold.  A one-device runner reports the batch's time on its stream in kernel_ms[0].  Off by default:
 * no event is recorded. */
int nb_runner_set_profiling(nb_runner *runner, int on);
int nb_runner_rank_times(nb_runner *runner, float *kernel_ms, float *wait_ms, int n);

int nb_runner_read_particles(nb_runner *runner, nb_particle *dst, size_t n);
int nb_runner_sim_params(const nb_runner *runner, nb_sim_params *out);
int nb_runner_step_num(const nb_runner *runner, uint64_t *out);
/* Borrow the runner's simulator (owned by the runner); NULL for a several-GPU runner. */
nb_sim *nb_runner_sim(nb_runner *runner);
int nb_runner_destroy(nb_runner *runner);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_H_ */
