// nb_inits.cpp -- seeded equivalents of the reference's three initial conditions.
//
// Reference: src/inits.rs:6-27 (uniform_init), :29-54 (disc_init), :56-83 (spherical_init).
// The reference draws from rand::thread_rng() (OS-seeded, not reproducible), so only the
// DISTRIBUTIONS can be matched; the generator below is this project's own and is specified
// bit-exactly so that tests can regenerate the same particles from numpy
// (tests/test_inits.py restates it independently):
//
//   draw k (k = 0,1,2,... in program order) of seed s:
//     z = s + (k+1) * 0x9E3779B97F4A7C15            (mod 2^64)        splitmix64
//     z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9
//     z = (z ^ (z >> 27)) * 0x94D049BB133111EB
//     z =  z ^ (z >> 31)
//     u = z >> 40                                    (24 bits)
//     unif = (float)((double)u * (2.0 / 16777215.0) - 1.0)   in [-1, 1], both ends reachable
//            (rand's Uniform::new_inclusive(-1.0, 1.0), inits.rs:8,31,59)
//
// All arithmetic below is binary32 with no FMA contraction (built with -ffp-contract=off).
#include <cmath>
#include <cstdint>

#include "nbody.h"

namespace {

struct Rng {
    uint64_t seed, k;
    explicit Rng(const void *user) : seed(user ? *static_cast<const uint64_t *>(user) : 0), k(0) {}
    float unif() {
        uint64_t z = seed + (++k) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        const uint64_t u = z >> 40;
        return (float)((double)u * (2.0 / 16777215.0) - 1.0);
    }
};

inline float length3(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }

}  // namespace

extern "C" {

// inits.rs:6-27: pos ~ U[-1,1]^3, vel ~ U[-1,1]^3 * 0.001, acc 0, mass 1.
void nb_init_uniform(const nb_sim_params *params, nb_particle *out, void *user) {
    Rng rng(user);
    for (uint32_t i = 0; i < params->particle_num; ++i) {
        nb_particle &p = out[i];
        p.position[0] = rng.unif();
        p.position[1] = rng.unif();
        p.position[2] = rng.unif();
        p.velocity[0] = rng.unif() * 0.001f;
        p.velocity[1] = rng.unif() * 0.001f;
        p.velocity[2] = rng.unif() * 0.001f;
        p.acceleration[0] = p.acceleration[1] = p.acceleration[2] = 0.0f;
        p.mass = 1.0f;
    }
}

// inits.rs:29-54: body 0 = mass 150000 at rest at the origin; the others on a thin disc,
// radius in [0.25,1] then scaled by its own length, on circular-ish orbits about +z.
void nb_init_disc(const nb_sim_params *params, nb_particle *out, void *user) {
    Rng rng(user);
    if (params->particle_num == 0) return;
    nb_particle &c = out[0];
    for (int a = 0; a < 3; ++a) c.position[a] = c.velocity[a] = c.acceleration[a] = 0.0f;
    c.mass = 150000.0f;
    for (uint32_t i = 1; i < params->particle_num; ++i) {
        float x = rng.unif(), y = rng.unif(), z = 0.0f;  // first try is planar, inits.rs:40
        float len = length3(x, y, z);
        while (len > 1.0f || len < 0.25f) {  // inits.rs:41-43
            x = rng.unif();
            y = rng.unif();
            z = rng.unif() * 0.1f;
            len = length3(x, y, z);
        }
        x *= len;  // pos *= pos.length(), inits.rs:44
        y *= len;
        z *= len;
        // vel = sqrt(g*1000/|pos|) * normalize(pos x Z), inits.rs:45;  pos x Z = (y, -x, 0)
        const float speed = sqrtf(params->g * 1000.0f / length3(x, y, z));
        const float cx = y, cy = -x, cz = 0.0f;
        const float inv = 1.0f / length3(cx, cy, cz);
        nb_particle &p = out[i];
        p.position[0] = x;
        p.position[1] = y;
        p.position[2] = z;
        p.velocity[0] = speed * (cx * inv);
        p.velocity[1] = speed * (cy * inv);
        p.velocity[2] = speed * (cz * inv);
        p.acceleration[0] = p.acceleration[1] = p.acceleration[2] = 0.0f;
        p.mass = 1.0f;
    }
}

// inits.rs:56-83: rejection-sample the unit ball; vel = normalize(pos)*0.4; mass ~ U[-1,1]+2.
void nb_init_spherical(const nb_sim_params *params, nb_particle *out, void *user) {
    Rng rng(user);
    const float kOutwardVel = 0.4f;  // inits.rs:57
    for (uint32_t i = 0; i < params->particle_num; ++i) {
        float x = rng.unif(), y = rng.unif(), z = rng.unif();
        while (length3(x, y, z) > 1.0f) {  // inits.rs:67-73
            x = rng.unif();
            y = rng.unif();
            z = rng.unif();
        }
        const float inv = 1.0f / length3(x, y, z);
        nb_particle &p = out[i];
        p.position[0] = x;
        p.position[1] = y;
        p.position[2] = z;
        p.velocity[0] = (x * inv) * kOutwardVel;
        p.velocity[1] = (y * inv) * kOutwardVel;
        p.velocity[2] = (z * inv) * kOutwardVel;
        p.acceleration[0] = p.acceleration[1] = p.acceleration[2] = 0.0f;
        p.mass = rng.unif() + 2.0f;  // inits.rs:79
    }
}

}  // extern "C"
