// nb_common.hpp -- internal declarations shared by the host side (nb_abi.cpp) and the
// HIP translation units.  Not part of the C ABI (include/nbody.h is).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>

#include "nbody.h"

namespace nb {

// ---- error plumbing: every failure records a message for nb_last_error() ----
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define NB_HIP_TRY(expr)                                                                       \
    do {                                                                                       \
        hipError_t nb_e_ = (expr);                                                             \
        if (nb_e_ != hipSuccess) {                                                             \
            ::nb::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(nb_e_), __FILE__, \
                            __LINE__);                                                         \
            return NB_ERR_HIP;                                                                 \
        }                                                                                      \
    } while (0)

// ---- all-pairs tiling constants ---------------------------------------------
// One workgroup owns an i-tile of NB_ITILE bodies; its waves split the j range.
constexpr uint32_t kWave = 64;
constexpr uint32_t kIB = 2;                    // bodies per lane (i-blocking)
constexpr uint32_t kITile = kWave * kIB;       // bodies per workgroup = 128
constexpr uint32_t kJTile = 64;                // bodies per wave-load of the j stream
constexpr uint32_t kPadTo = 256;               // posm buffers are padded to a multiple of this

// Peer position buffers of a multi-GPU run inside one process (nb_group.cpp): the kernel that
// finishes a rank's step stores the rank's new slice into every peer's next-step buffer as well
// (direct stores over xGMI, one slice per point-to-point link) -- no copy, no collective.
constexpr int kMaxPeers = 15;
struct PeerDst {
    float4 *p[kMaxPeers];
    uint32_t n;
};

// a word in every peer's table (the create-time check that peer stores arrive: nb_naive.hip, nb_group.cpp)
struct PeerWords {
    uint32_t *p[kMaxPeers];
    uint32_t n;
};
hipError_t launch_peer_check_store(const PeerWords &dst, uint32_t slot, uint32_t value, hipStream_t stream);
hipError_t launch_peer_check_read(const uint32_t *words, uint32_t world, uint32_t me, uint32_t tag, uint32_t *bad,
                                  hipStream_t stream);

// ---- launchers implemented in nb_naive.hip ----------------------------------
struct NaiveLaunch {
    const float4 *posm_src;  // [n_pad] x,y,z,m  (previous step, all bodies)
    float4 *posm_dst;        // [n_pad] this step's positions; only [lo,hi) written
    float4 *vel;             // [hi-lo padded] this rank's velocities (in place)
    float4 *acc;             // [hi-lo padded] this rank's stored accelerations (in place)
    uint32_t n;              // real bodies
    uint32_t n_pad;          // allocated bodies (multiple of kPadTo, zero-filled tail)
    uint32_t lo, hi;         // this rank's body range
    float g, e, dt;
    int variant;             // kernel variant (see nb_naive.hip); <0 = default
    int jsplit;              // j-splits across workgroups; <=0 = automatic
    float4 *partial;         // [partial_slices][partial_stride] partial sums (j-split only)
    uint32_t partial_stride; // bodies per slice (>= hi-lo)
    uint32_t partial_slices; // slices allocated
    int phase;               // kPhaseAll / kPhaseLocal / kPhaseRemote
    PeerDst peers;           // where else the new slice goes (finish kernel only)
};
enum { kPhaseAll = 0, kPhaseLocal = 1, kPhaseRemote = 2 };
// How a launch will be shaped for n total bodies of which this rank owns [lo, hi).
struct NaivePlan {
    int variant;
    uint32_t blocks;       // i-tiles
    uint32_t jsplit;       // partial-sum slices per step (1 = single-kernel step)
    uint32_t js_local, js_remote;          // two-phase: slices of the own / the other j tiles
    uint32_t n_tiles, lo_tile, local_tiles;
    bool two_phase;
};
NaivePlan plan_naive(uint32_t n, uint32_t lo, uint32_t hi, int variant, int jsplit, bool two_phase);
hipError_t launch_naive_step(const NaiveLaunch &a, hipStream_t stream);
int naive_variant_count();
const char *naive_variant_name(int v);

// AoS (nb_particle, 40 B) <-> device SoA conversion, body range [lo,hi) of `aos` (indexed
// globally) <-> posm[lo..hi) (global index) and vel/acc (local index i-lo).
hipError_t launch_aos_to_soa(const nb_particle *aos, float4 *posm, float4 *vel, float4 *acc,
                             uint32_t n, uint32_t lo, uint32_t hi, hipStream_t stream);
hipError_t launch_soa_to_aos(const float4 *posm, const float4 *vel, const float4 *acc,
                             nb_particle *aos, uint32_t n, uint32_t lo, uint32_t hi,
                             hipStream_t stream);

}  // namespace nb
