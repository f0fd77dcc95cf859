// headless.cpp -- C++ counterpart of the reference's src/bin/headless.rs:14-35 (and of the
// criterion groups in benches/benchmark.rs:12-49) over the drop-in API of simulator.hpp.
//
//   headless [--sim naive|tree] [--n N] [--steps S] [--theta T] [--init uniform|disc|spherical]
//            [--seed K] [--device D | --devices D0,D1,...] [--g G] [--dt DT] [--dump FILE]
//
// --devices: the step sharded over several GPUs of this process (nb_runner_create_multi; both simulators);
// --let K (with --sim tree --devices): Morton domains + LET exchange, migration every K-th step (0: never);
// a device id may repeat.
//
// --dump FILE writes the final state as a snapshot (the layout of
// wgpu_n_body_amd/snapshot.py: "NBSNAP01", u64 step, SimParams, Particle[n]).
//
// Defaults reproduce headless.rs: TreeSim, 4,000,000 bodies, theta 0.75, uniform_init,
// 10 steps, printing "Step Duration: {} us" per step.  (TreeSim needs the Barnes-Hut build;
// pass --sim naive --n 65536 for the all-pairs path.)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "simulator.hpp"

static bool write_snapshot(const std::string &path, const nbody::SimParams &sp,
                           const std::vector<nbody::Particle> &parts, uint64_t step) {
    std::FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite("NBSNAP01", 1, 8, f) == 8 && std::fwrite(&step, sizeof step, 1, f) == 1 &&
              std::fwrite(&sp, sizeof sp, 1, f) == 1 &&
              std::fwrite(parts.data(), sizeof(nbody::Particle), parts.size(), f) == parts.size();
    return std::fclose(f) == 0 && ok;
}

template <class Sim>
static int run(const nbody::SimParams &sp, const nbody::AddParams &ap, const nbody::InitFn &init,
               int steps, int device, const std::vector<int> &devices, const std::string &dump, int let) {
    std::puts("Initializing Simulation");
    nbody::OfflineHeadless<Sim> runner = devices.empty() ? nbody::OfflineHeadless<Sim>(sp, ap, init, device)
                                                         : nbody::OfflineHeadless<Sim>(sp, ap, init, devices, let);
    std::puts("Running Simulation");
    for (int i = 0; i < steps; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        runner.step();
        const auto us = std::chrono::duration_cast<std::chrono::microseconds>(
                            std::chrono::steady_clock::now() - t0).count();
        std::printf("Step Duration: %lld \xC2\xB5s\n", (long long)us);
    }
    std::puts("Finished Running");
    if (!dump.empty()) {
        const std::vector<nbody::Particle> parts = runner.read_particles();
        if (!write_snapshot(dump, sp, parts, runner.step_num())) {
            std::fprintf(stderr, "cannot write %s\n", dump.c_str());
            return 1;
        }
    }
    return 0;
}

int main(int argc, char **argv) {
    std::string sim = "tree", init = "uniform", dump;
    std::vector<int> devices;
    nbody::SimParams sp{4000000u, 0.000001f, 0.0001f, 0.016f};  // headless.rs:15-20
    float theta = 0.75f;
    int steps = 10, device = -1, let = -1;
    uint64_t seed = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i], v = argv[i + 1];
        if (k == "--sim") sim = v;
        else if (k == "--n") sp.particle_num = (uint32_t)std::strtoul(v.c_str(), nullptr, 10);
        else if (k == "--steps") steps = std::atoi(v.c_str());
        else if (k == "--theta") theta = (float)std::atof(v.c_str());
        else if (k == "--init") init = v;
        else if (k == "--seed") seed = std::strtoull(v.c_str(), nullptr, 10);
        else if (k == "--device") device = std::atoi(v.c_str());
        else if (k == "--g") sp.g = (float)std::atof(v.c_str());
        else if (k == "--dt") sp.dt = (float)std::atof(v.c_str());
        else if (k == "--dump") dump = v;
        else if (k == "--let") let = std::atoi(v.c_str());  // with --devices and --sim tree: LET scheme, migrate every k-th step
        else if (k == "--devices") {
            for (size_t a = 0; a <= v.size();) {
                const size_t b = std::min(v.find(',', a), v.size());
                devices.push_back(std::atoi(v.substr(a, b - a).c_str()));
                a = b + 1;
            }
        }
        else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    const nbody::InitFn fn = init == "disc" ? nbody::inits::disc_init(seed)
                           : init == "spherical" ? nbody::inits::spherical_init(seed)
                                                 : nbody::inits::uniform_init(seed);
    try {
        if (sim == "naive")
            return run<nbody::NaiveSim>(sp, nbody::AddParams::NaiveSimParams(), fn, steps, device, devices, dump, -1);
        return run<nbody::TreeSim>(sp, nbody::AddParams::TreeSimParams(theta), fn, steps, device, devices, dump, let);
    } catch (const nbody::Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.code(), e.what());
        return 1;
    }
}
