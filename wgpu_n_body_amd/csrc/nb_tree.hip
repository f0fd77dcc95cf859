// nb_tree.hip -- Barnes-Hut simulator (TreeSim, src/sims/tree.rs + shaders/tree.wgsl).
// Placeholder until the device octree path lands: nb_sim_create reports NB_ERR_UNSUPPORTED.
#include "nb_sim.hpp"

namespace nb {
SimBase *make_tree_sim() { return nullptr; }
}  // namespace nb
