// nb_tree.hip -- Barnes-Hut simulator for gfx950: TreeSim (src/sims/tree.rs) + the tree-walk
// shader (src/sims/shaders/tree.wgsl), rebuilt so that NOTHING leaves the GPU.
//
// The reference's step (TreeSim::encode, tree.rs:262-353) maps the particle buffer to the
// host, builds the octree with a serial BFS over bump-allocated index lists (tree.rs:417-546),
// reorders the particles in DFS order (tree.rs:564-602), uploads particles + tree and only
// then dispatches the walk.  Here the same tree -- same cells, same node numbering, same
// children tables, same body order -- is constructed on the device from Morton keys:
//
//   1 bound               max |coord| (>= 1.0) -> root cube [-b,b]^3              tree.rs:424-446
//                         (bound_kernel; in steady state accumulated by the previous step's walk)
//   2 morton_kernel       63-bit key per body by the reference's own float descent:
//                         digit = (x>cx) | (y>cy)<<1 | (z>cz)<<2 with strict '>',
//                         centre += +-width/4, width /= 2   (21 levels)        tree.rs:549-562
//   3 sort                stable, by (key, index).  Radix passes of 8 bits (per-tile digit histogram in
//                         LDS, per-bin scan over the tiles, stable scatter ranked with wave ballots)
//                         over the HIGH digits only -- 2 or 3 passes on (high word of the key, index)
//                         pairs, 3e -- then a fix-up of the bodies that tie there (runs_rank_kernel:
//                         a thread per body; runs_fix_kernel for the 64-bit form, 3d); the whole sort
//                         in one launch by counting up to 12,288 bodies (3c); 8 passes over the
//                         whole key as the cross-check (tuning key tree_sort_mode 0)
//   4-6a cells_a/scan/c   bodies into sorted order = the reference's DFS order (tree.rs:564-602);
//                         a cell at depth d exists for every key-prefix run: body k opens the
//                         internal cells of depths (cpl[k-1], cpl[k]] and owns one leaf at depth
//                         max(cpl[k-1],cpl[k])+1, where cpl = common prefix length (levels) of
//                         neighbouring keys.  Node id = (#nodes of smaller depth) + rank among
//                         the nodes of its depth in key order -- exactly the reference's BFS
//                         allocation order (tree.rs:461,517-519; slice_alloc.rs:52-59); binary64
//                         prefix sums of (m x, m y, m z, m) over the sorted bodies for the mass /
//                         centre of gravity of every cell (tree.rs:486-505).  Three launches.
//   6 fill_kernel         per node: body range by a galloping search on the keys; children = the
//                         consecutive next-depth ids starting at the first body's own child
//                         (0 = none; a leaf's children[0] = the body's source index, tree.rs:532)
//   8 walk                tree.wgsl:41-111 with the INTENDED semantics: self
//                         excluded by identity, a leaf is a body, no fixed 64-entry stack; every
//                         body applies ITS OWN acceptance test size/dist < theta to exactly the
//                         cells of the reference's per-thread walk (visit counts equal the
//                         oracle's), in a different order of summation (fp32 rounding).
//                         8b walk_cells_kernel (default): a wave walks for 8 (or 16) bodies held in
//                         scalars, its 64 lanes hold 64 cells of the traversal frontier; the test
//                         compares the cell's stored acceptance radius^2 = size^2 / theta^2 with
//                         r^2, the accumulation runs under exec = the lanes that take the cell.
//                         8  walk_kernel: a wave walks for 64 bodies, one cell at a time.
//
// Deviations, all documented in DESIGN.md: bodies whose 63-bit keys collide (closer than
// root_width/2^21) cannot be separated (the reference would recurse until its 4N-node buffer
// overflows); cog/mass come from binary64 prefix sums instead of a sequential fp32 sum per cell.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "nb_sim.hpp"

namespace nb {
namespace {

constexpr int kLevels = 21;            // 3 x 21 = 63 key bits
constexpr int kMaxDepth = kLevels + 1;  // leaves can sit at depth 1..21 (+1 guard)
// bodies per thread of a sort tile: 4 up to kSortSmallMax bodies (more, smaller workgroups: build
// -13 us at 131,072 bodies, -5 at 524,288), 8 beyond (half the histogram rows: -14 us at 2^20, -40 at 2^21)
constexpr uint32_t kSortThreads = 256, kSortItems = 8, kSortItemsSmall = 4, kSortSmallMax = 786432;
constexpr uint32_t kSortBits = 8, kSortWideBits = 9, kSortMaxBins = 1u << kSortWideBits;  // digit widths (the kernels take 7..9)
#ifndef NB_SORT_INLINE_BLOCKS
#define NB_SORT_INLINE_BLOCKS 32
#endif
constexpr uint32_t kSortInlineScanBlocks = NB_SORT_INLINE_BLOCKS;  // up to 32,768 bodies the scatter scans the tile counts itself (-6 %)
// wave-level stack of sibling groups (16 B each, 3 KiB per wave): a depth-first walk pushes at
// most 8 groups per level and pops one, so 7 x 21 + 1 = 148 entries is the most it can hold
constexpr uint32_t kWalkStack = 192;

__device__ __forceinline__ float kick(float v, float a, float dt) {
#pragma clang fp contract(off)
    return v + (a * dt) / 2.0f;  // tree.wgsl:105,108
}
__device__ __forceinline__ float drift(float x, float v, float dt) {
#pragma clang fp contract(off)
    return x + v * dt;  // tree.wgsl:106
}

// ---- wave-level scans by DPP (no LDS round trip) ---------------------------------------------------
#define NB_DPP(old, src, ctrl, row_mask) \
    ((uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(src), (ctrl), (row_mask), 0xf, false))

// inclusive prefix sum over the 64 lanes (row_shr within the 16-lane rows, then the row totals)
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t x) {
    x += NB_DPP(0, x, 0x111, 0xf);  // row_shr:1
    x += NB_DPP(0, x, 0x112, 0xf);  // row_shr:2
    x += NB_DPP(0, x, 0x114, 0xf);  // row_shr:4
    x += NB_DPP(0, x, 0x118, 0xf);  // row_shr:8
    x += NB_DPP(0, x, 0x142, 0xa);  // row_bcast:15 -> rows 1 and 3
    x += NB_DPP(0, x, 0x143, 0xc);  // row_bcast:31 -> rows 2 and 3
    return x;
}

// minimum / maximum over the 64 lanes (the same DPP steps; a lane without a source keeps its own value): in lane 63
__device__ __forceinline__ int wave_min_to_lane63(int v) {
    uint32_t x = (uint32_t)v;
#define NB_STEPM(ctrl, row_mask) x = (uint32_t)min((int)x, (int)NB_DPP(x, x, ctrl, row_mask))
    NB_STEPM(0x111, 0xf); NB_STEPM(0x112, 0xf); NB_STEPM(0x114, 0xf); NB_STEPM(0x118, 0xf);
    NB_STEPM(0x142, 0xa); NB_STEPM(0x143, 0xc);
#undef NB_STEPM
    return (int)x;
}
__device__ __forceinline__ int wave_max_to_lane63(int v) {
    uint32_t x = (uint32_t)v;
#define NB_STEPM(ctrl, row_mask) x = (uint32_t)max((int)x, (int)NB_DPP(x, x, ctrl, row_mask))
    NB_STEPM(0x111, 0xf); NB_STEPM(0x112, 0xf); NB_STEPM(0x114, 0xf); NB_STEPM(0x118, 0xf);
    NB_STEPM(0x142, 0xa); NB_STEPM(0x143, 0xc);
#undef NB_STEPM
    return (int)x;
}

// ... of binary64 values (the moment sums): the two halves move by DPP, the add is a v_add_f64.  Lanes without a
// source in a step add +0.0.  Twelve VALU instructions per step instead of two LDS-crossbar shuffles
// (ds_bpermute) and their ~60-cycle round trip: the scans of cells_a / cells_c were chains of those.
__device__ __forceinline__ double wave_scan_f64(double v) {
    uint32_t lo = (uint32_t)__double_as_longlong(v), hi = (uint32_t)((unsigned long long)__double_as_longlong(v) >> 32);
#define NB_STEP64(ctrl, row_mask)                                                                        \
    {                                                                                                    \
        const uint32_t l2 = NB_DPP(0, lo, ctrl, row_mask), h2 = NB_DPP(0, hi, ctrl, row_mask);           \
        const double s = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)) +        \
                         __longlong_as_double((long long)(((unsigned long long)h2 << 32) | l2));         \
        lo = (uint32_t)__double_as_longlong(s);                                                          \
        hi = (uint32_t)((unsigned long long)__double_as_longlong(s) >> 32);                              \
    }
    NB_STEP64(0x111, 0xf);
    NB_STEP64(0x112, 0xf);
    NB_STEP64(0x114, 0xf);
    NB_STEP64(0x118, 0xf);
    NB_STEP64(0x142, 0xa);
    NB_STEP64(0x143, 0xc);
#undef NB_STEP64
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ---- 1. bound -----------------------------------------------------------------------------------
// max over bodies and axes of |coord|, never below 1.0 (rayon reduce identity [1.0;3],
// tree.rs:427-433).  Non-negative floats order like their bit patterns -> atomicMax on u32.
__global__ __launch_bounds__(256) void bound_kernel(const float4 *__restrict__ posm, uint32_t n,
                                                    uint32_t *bound_bits) {
    __shared__ float s_m[4];
    float m = 1.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = posm[i];
        m = fmaxf(m, fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z))));
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    // one atomic per workgroup: thousands of atomics on one word serialise (~12 ns each)
    if (threadIdx.x == 0)
        atomicMax(bound_bits, __float_as_uint(fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]))));
}

// The walk kernels also accumulate the NEXT step's bound from the positions they write, so a steady-state
// step needs neither bound_kernel nor a memset: morton_kernel takes the maximum of the slots (and of 1.0).
// A wave whose bodies stay inside the unit cube has nothing to say (the bound is never below 1.0,
// tree.rs:427-433); the others add their maximum to one of 1,024 slots -- 64 cache lines -- with an atomic
// nobody waits for.  (Round 2 read the slot first, "skip the atomic when not above": a device-scope load of one
// of FOUR lines by every wave, in its prologue and waited for -- the lines' channel served ~300 waves per us
// chip-wide, and a wave of a 32,768-body walk spent 8 us (up to 27) between its launch and its first batch.)
constexpr uint32_t kBoundSlots = 1024;
__device__ __forceinline__ void publish_bound(uint32_t *__restrict__ slots, uint32_t key, float m) {
    const uint32_t bits = __float_as_uint(m);  // non-negative floats order like their bit patterns
    if (bits > 0x3f800000u) atomicMax(slots + (key & (kBoundSlots - 1u)), bits);
}

// (see morton_kernel) the cell of a coordinate at the finest level, and its 21 bits spread to every third bit
__device__ __forceinline__ uint32_t cell_21(float x, float inv_h) {
    const int t = (int)__builtin_ceilf(x * inv_h) + (1 << 20) - 1;
    return (uint32_t)min(max(t, 0), (1 << 21) - 1);
}
__device__ __forceinline__ uint64_t spread_21(uint32_t v) {
    // the low 11 and the high 10 bits separately, in 32-bit arithmetic: bit i -> bit 3 i
    auto spread = [](uint32_t x) {  // x < 2^11
        x = (x | (x << 16)) & 0x070000ffu;
        x = (x | (x << 8)) & 0x0700f00fu;
        x = (x | (x << 4)) & 0x430c30c3u;
        x = (x | (x << 2)) & 0x49249249u;
        return x;
    };
    return (uint64_t)spread(v & 0x7ffu) | ((uint64_t)spread(v >> 11) << 33);
}

// ---- 2. keys ------------------------------------------------------------------------------------
// One workgroup per sort tile: the keys, and the tile's histogram of the first digit (saves the
// first pass its histogram launch).
// bound_src: where the root cube's half width comes from -- scalars[0] (bound_kernel / the LET
// maximum; n_src = 1) or the kBoundSlots words accumulated by the previous step's walk (n_src =
// kBoundSlots); it is republished in scalars[0].
__global__ __launch_bounds__(2 * kSortThreads) void morton_kernel(const float4 *__restrict__ posm, uint32_t n,
                                                              const uint32_t *__restrict__ bound_src,
                                                              uint32_t n_src, uint32_t *__restrict__ bound_bits,
                                                              uint64_t *__restrict__ keys,
                                                              uint32_t *__restrict__ idx,
                                                              uint32_t *__restrict__ hist, uint32_t nblocks,
                                                              uint32_t items, uint32_t hist_shift,
                                                              uint32_t hist_bins, uint32_t *__restrict__ key_hi,
                                                              uint32_t key_descent_only) {
    // key_hi (the radix passes sort 32-bit high words paired with indices, section 3e): the high word of
    // every key beside the key, and no identity index array -- the first pass makes it up.
    // (blockDim.x * items bodies = a sort tile: a workgroup leaves the tile's histogram of the digit the
    // FIRST radix pass sorts by -- hist_bins values at bit hist_shift; the counting sort of small
    // problems needs no histogram and takes items = 1: more, shorter workgroups)
    __shared__ uint32_t s_hist[kSortMaxBins];
    for (uint32_t b = threadIdx.x; b < kSortMaxBins; b += blockDim.x) s_hist[b] = 0;
    __syncthreads();
    uint32_t bmax;
    if (n_src > 1u) {  // the slots of the previous walk: a share per thread, the maximum through LDS
        __shared__ uint32_t s_bmax[2 * kSortThreads / 64];
        uint32_t mine = 0;
        for (uint32_t k = threadIdx.x; k < n_src; k += blockDim.x) mine = max(mine, bound_src[k]);
        mine = (uint32_t)wave_max_to_lane63((int)mine);  // (bit patterns of non-negative floats: positive as int)
        if ((threadIdx.x & 63u) == 63u) s_bmax[threadIdx.x >> 6] = mine;
        __syncthreads();
        bmax = 0;
        for (uint32_t w = 0; w < blockDim.x / 64u; ++w) bmax = max(bmax, s_bmax[w]);
    } else {
        bmax = bound_src[0];
    }
    const float bound = fmaxf(1.0f, __uint_as_float(bmax));  // never below 1.0, tree.rs:427-433
    if (blockIdx.x == 0 && threadIdx.x == 0 && bound_src != bound_bits) *bound_bits = __float_as_uint(bound);
    const float root_w = bound * 2.0f;  // root width, tree.rs:465
    // The quarter widths of the 21 levels, width / 4 (shift_node_center) with width halved per level: exact powers
    // of two times root_w, i.e. root_w's bit pattern with its exponent lowered -- wave-uniform integers the scalar
    // unit computes, where `w / 4.0f; w = w / 2.0f` cost two vector multiplies per level and body.
    const uint32_t root_bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(root_w));
    const bool pow2_root = (root_bits & 0x007fffffu) == 0u && !key_descent_only;
    const float inv_h = __uint_as_float((275u - (root_bits >> 23)) << 23);  // 2^21 / root_w for a power of two
    // (the loads of a tile's bodies first, all in flight together: one body after the other the
    // kernel waited out eight memory latencies per thread)
    float4 pv[kSortItems];
#pragma unroll
    for (uint32_t c = 0; c < kSortItems; ++c) {
        const uint32_t i = (blockIdx.x * items + c) * blockDim.x + threadIdx.x;
        if (c < items && i < n) pv[c] = posm[i];
    }
#pragma unroll
    for (uint32_t c = 0; c < kSortItems; ++c) {
        const uint32_t i = (blockIdx.x * items + c) * blockDim.x + threadIdx.x;
        if (c >= items || i >= n) break;
        const float4 p = pv[c];
        uint64_t key = 0;
        if (pow2_root) {
            // root_w a power of two (every state inside the unit cube: bound = 1.0): the centres of the descent
            // below are multiples of root_w / 2^22 below root_w / 2 -- at most 21 significant bits, exact in fp32 --
            // so its 21 strict comparisons spell the binary digits of ceil((x + root_w / 2) / h) - 1, h = root_w /
            // 2^21 the finest cell (a body ON a cell boundary belongs below it; x = -root_w / 2 gives all zeros).
            // x / h is an exact scaling, its ceiling an exact integer of at most 21 bits: three instructions per
            // axis and a bit interleave instead of 21 dependent levels of compare, select, add.
            key = spread_21(cell_21(p.x, inv_h)) | (spread_21(cell_21(p.y, inv_h)) << 1) | (spread_21(cell_21(p.z, inv_h)) << 2);
        } else {
            float cx = 0.f, cy = 0.f, cz = 0.f;
#pragma unroll
            for (int l = 0; l < kLevels; ++l) {
#pragma clang fp contract(off)
                const uint32_t bx = p.x > cx, by = p.y > cy, bz = p.z > cz;  // decide_octant, strict >
                key = (key << 3) | (uint64_t)(bx | (by << 1) | (bz << 2));
                const float q = __uint_as_float(root_bits - ((uint32_t)(l + 2) << 23));  // (root_w / 2^l) / 4, exactly
                cx = cx + (bx ? q : -q);  // shift_node_center
                cy = cy + (by ? q : -q);
                cz = cz + (bz ? q : -q);
            }
        }
        keys[i] = key;
        if (key_hi) key_hi[i] = (uint32_t)(key >> 32);
        else idx[i] = i;
        if (hist) atomicAdd(&s_hist[(uint32_t)(key >> hist_shift) & (hist_bins - 1u)], 1u);
    }
    __syncthreads();
    if (hist)
        for (uint32_t b = threadIdx.x; b < hist_bins; b += blockDim.x) hist[b * nblocks + blockIdx.x] = s_hist[b];  // bin-major
}

// ---- 3. radix sort (LSD, digits of W bits, pairs) ------------------------------------------------
// A block owns a tile of kSortThreads * ITEMS elements; wave w owns the contiguous sub-range
// [w*64*ITEMS, (w+1)*64*ITEMS) of it, read in ITEMS chunks of 64 -- so "wave, chunk, lane" order
// IS the input order, which is what makes the per-wave ranking below stable.
// (Counting the tile histograms of digit p + 1 inside the scatter of pass p, with one global atomic
// per element where it lands, was measured and dropped: 47 instead of 12 us per scatter at 2^20
// bodies, 10.8 instead of 5 + 5 at 8,192 -- profiles/r02_sort_experiments.txt.)
// (ITEMS elements per thread of a 2 x kSortThreads workgroup: the tile of the scatter, whatever its order inside)
template <uint32_t ITEMS, typename KeyT = uint64_t>
__global__ __launch_bounds__(2 * kSortThreads) void radix_hist_kernel(
    const KeyT *__restrict__ keys, uint32_t n, uint32_t shift, uint32_t bins, uint32_t *__restrict__ hist,
    uint32_t nblocks) {
    constexpr uint32_t THREADS = 2u * kSortThreads;
    __shared__ uint32_t s_hist[kSortMaxBins];
    for (uint32_t b = threadIdx.x; b < kSortMaxBins; b += THREADS) s_hist[b] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (THREADS * ITEMS) + threadIdx.x;
#pragma unroll
    for (uint32_t c = 0; c < ITEMS; ++c) {
        const uint32_t i = base + c * THREADS;
        if (i < n) atomicAdd(&s_hist[(uint32_t)(keys[i] >> shift) & (bins - 1u)], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < bins; b += THREADS) hist[b * nblocks + blockIdx.x] = s_hist[b];  // bin-major
}

// One workgroup per bin: exclusive scan of that bin's per-block counts; the bin total goes to
// totals[bin].  (rows of `nblocks` entries; used with 256 bins by the sort and 22 by the ids.)
__global__ __launch_bounds__(256) void bin_scan_kernel(uint32_t *__restrict__ hist,
                                                       uint32_t nblocks,
                                                       uint32_t *__restrict__ totals) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_carry;
    uint32_t *row = hist + (size_t)blockIdx.x * nblocks;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t base = 0; base < nblocks; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? row[i] : 0u;
        const uint32_t x = wave_scan_u32(v);  // inclusive scan within the wave
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        uint32_t off = s_carry;
        for (uint32_t w = 0; w < wave; ++w) off += s_wave[w];
        if (i < nblocks) row[i] = off + x - v;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = off + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = s_carry;
}

// exclusive scan over the workgroup of one value per thread
__device__ __forceinline__ uint32_t sort_scan_block(uint32_t v, uint32_t *s_w) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t x = wave_scan_u32(v);
    if (lane == 63) s_w[wave] = x;
    __syncthreads();
    uint32_t off = 0;
    for (uint32_t w = 0; w < wave; ++w) off += s_w[w];
    __syncthreads();
    return off + x - v;
}

// SCAN_INLINE (few tiles: the launch-bound small problems): `hist` holds the raw per-tile counts
// and every block sums its digit rows itself -- thread t adds up its rows -- which saves the
// bin_scan launch of the pass.  Thread t looks after the digits [t PER, (t + 1) PER).
// THREADS x ITEMS elements = a sort tile (2,048-element tiles run as 512 threads x 4: twice the waves per SIMD
// of 256 x 8 for a kernel that is a chain of LDS round trips and barriers).
// KeyT = uint32_t: the high words of the keys (section 3e) -- 8-byte instead of 12-byte elements; vals_in = null:
// the values are the positions themselves (the first pass: no identity array is ever written or read).
template <int W, uint32_t THREADS, uint32_t ITEMS, bool SCAN_INLINE, typename KeyT = uint64_t>
__global__ __launch_bounds__(THREADS) void radix_scatter_kernel(
    const KeyT *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
    KeyT *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint32_t n, uint32_t shift,
    const uint32_t *__restrict__ hist, const uint32_t *__restrict__ totals, uint32_t nblocks) {
    constexpr uint32_t NB = 1u << W, PER = (NB + THREADS - 1u) / THREADS, TILE = THREADS * ITEMS, NWV = THREADS / 64u;
    __shared__ uint32_t s_cnt[NWV][NB];  // per-wave running digit counts -> exclusive wave offsets
    __shared__ uint32_t s_base[NB];    // global start of each digit + this block's offset in it
    __shared__ uint32_t s_tile[NB], s_w[NWV];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t b0 = threadIdx.x * PER;  // my digits: b0 .. b0 + PER - 1 (none if b0 >= NB)
    for (uint32_t w = 0; w < NWV; ++w)
        for (uint32_t b = threadIdx.x; b < NB; b += THREADS) s_cnt[w][b] = 0;
    {   // exclusive scan of the digit totals (tiny; every block redoes it)
        uint32_t t[PER], mine[PER], sum = 0;  // digit total over all tiles; the tiles before this one
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q) {
            const uint32_t d = b0 + q;
            t[q] = mine[q] = 0u;
            if (d < NB) {
                if (SCAN_INLINE) {
                    const uint32_t *row = hist + (size_t)d * nblocks;
                    for (uint32_t b = 0; b < nblocks; ++b) {
                        const uint32_t v = row[b];
                        mine[q] += b < blockIdx.x ? v : 0u;
                        t[q] += v;
                    }
                } else {
                    t[q] = totals[d];
                    mine[q] = hist[d * nblocks + blockIdx.x];
                }
            }
            sum += t[q];
        }
        uint32_t run = sort_scan_block(sum, s_w);
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q) {
            if (b0 + q < NB) s_base[b0 + q] = run + mine[q];
            run += t[q];
        }
    }
    __syncthreads();

    const uint32_t base = blockIdx.x * TILE + wave * (64 * ITEMS);
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    KeyT key[ITEMS];
    uint32_t val[ITEMS], local[ITEMS];
#pragma unroll
    for (uint32_t c = 0; c < ITEMS; ++c) {
        const uint32_t i = base + c * 64 + lane;
        const bool valid = i < n;
        key[c] = valid ? keys_in[i] : (KeyT)~(KeyT)0;
        val[c] = !valid ? 0u : vals_in ? vals_in[i] : i;
        const uint32_t d = (uint32_t)(key[c] >> shift) & (NB - 1u);
        // lanes holding the same digit (ballot match over the W digit bits)
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < W; ++b) {
            const uint64_t bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const uint32_t rank = __popcll(peers & lt_mask);
        const uint32_t before = valid ? s_cnt[wave][d] : 0u;  // same address for all peers
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) s_cnt[wave][d] = before + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        local[c] = before + rank;
    }
    __syncthreads();
    {   // per digit: exclusive prefix over the waves and the digit's count in this tile; then the
        // exclusive scan of the tile's digit counts: where each digit's run starts inside the tile
        uint32_t cnt[PER], sum = 0;
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q) {
            cnt[q] = 0;
            if (b0 + q < NB) {
                uint32_t o = 0;
                for (uint32_t w = 0; w < NWV; ++w) {
                    const uint32_t t = s_cnt[w][b0 + q];
                    s_cnt[w][b0 + q] = o;
                    o += t;
                }
                cnt[q] = o;
            }
            sum += cnt[q];
        }
        uint32_t run = sort_scan_block(sum, s_w);
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q) {
            if (b0 + q < NB) s_tile[b0 + q] = run;
            run += cnt[q];
        }
    }
    __syncthreads();
    // Stage the tile in LDS in digit order, then write it out with consecutive threads on
    // consecutive elements: each digit's run lands in global memory as one contiguous, coalesced
    // stream instead of 64 scattered 8-byte stores per wave instruction.
    __shared__ KeyT s_key[TILE];
    __shared__ uint32_t s_val[TILE];
#pragma unroll
    for (uint32_t c = 0; c < ITEMS; ++c) {
        const uint32_t i = base + c * 64 + lane;
        if (i < n) {
            const uint32_t d = (uint32_t)(key[c] >> shift) & (NB - 1u);
            const uint32_t pos = s_tile[d] + s_cnt[wave][d] + local[c];
            s_key[pos] = key[c];
            s_val[pos] = val[c];
        }
    }
    __syncthreads();
    const uint32_t tile_n = min(TILE, n - blockIdx.x * TILE);
#pragma unroll
    for (uint32_t c = 0; c < ITEMS; ++c) {
        const uint32_t j = c * THREADS + threadIdx.x;
        if (j < tile_n) {
            const KeyT k = s_key[j];
            const uint32_t d = (uint32_t)(k >> shift) & (NB - 1u);
            const uint32_t dst = s_base[d] + (j - s_tile[d]);
            keys_out[dst] = k;
            vals_out[dst] = s_val[j];
        }
    }
}

// ---- 3c. small problems: the whole sort in ONE launch, by counting ------------------------------
// Up to kRankSortMax bodies a step is bound by its chain of dependent launches (a trivial kernel
// costs ~4.3 us end to end; the radix sort is sixteen of them), not by work.  There the sorted
// position of a body is simply COUNTED: rank(i) = #{ j : (key_j, j) < (key_i, i) } -- the all-pairs
// pattern of the force kernel, on integers: N^2 64-bit compares (6.7e7 at 8,192 bodies, a few
// microseconds on 1,024 SIMDs), ties broken by source index exactly as the stable radix sort breaks
// them.  A workgroup owns 64 bodies; its 16 waves split the j range, each staging its slice in LDS;
// (key_j, j) < (key_i, i) is evaluated as key_j < key_i + [j < i], one compare per pair once a
// wave's j slice lies entirely below or above its bodies.
// (measured per runner.step(), theta 0.75: 8,192 bodies 77.0 us counted / 80.6 radix; 12,288: 84.8 / 84.9;
// 16,384: 93.7 / 89.1 -- the two-pass high-word radix sort with its thread-per-body fix-up takes over there)
constexpr uint32_t kRankSortMax = 12288;
constexpr uint32_t kRankWaves = 16;
constexpr int kRankUnroll = 32;

// (Two workgroups per tile with a ticket for the last to add up and scatter, and the j slices staged
// in LDS instead of read through the scalar cache, were both measured slower.)
__global__ __launch_bounds__(64 * kRankWaves) void rank_sort_kernel(const uint64_t *__restrict__ keys, uint32_t n,
                                                                    uint64_t *__restrict__ keys_out,
                                                                    uint32_t *__restrict__ order) {
    __shared__ uint32_t s_cnt[kRankWaves][64];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t i0 = blockIdx.x * 64u, i = i0 + lane;
    const uint64_t ki = i < n ? keys[i] : ~0ull;
    const uint32_t per = (n + kRankWaves - 1u) / kRankWaves;
    const uint32_t j_lo = min(wave * per, n), j_hi = min(j_lo + per, n);
    uint32_t count = 0;
    // (kRankUnroll keys per round of scalar loads -- wave-uniform addresses go through the scalar
    // cache; a round costs one load latency, so the rounds are made long)
#define NB_COUNT_RANGE(A, B, CMP)                                   \
    {                                                               \
        const uint64_t *kp = keys + (A), *ke = keys + (B);          \
        for (; kp + kRankUnroll <= ke; kp += kRankUnroll) {         \
            uint64_t kk[kRankUnroll];                               \
            _Pragma("unroll") for (int u = 0; u < kRankUnroll; ++u) kk[u] = kp[u]; \
            _Pragma("unroll") for (int u = 0; u < kRankUnroll; ++u) count += (kk[u] CMP ki) ? 1u : 0u; \
        }                                                           \
        for (; kp < ke; ++kp) count += (*kp CMP ki) ? 1u : 0u;      \
    }
    // j below the workgroup's bodies: (key_j, j) < (key_i, i)  <=>  key_j <= key_i
    const uint32_t below_end = min(j_hi, i0);
    if (j_lo < below_end) NB_COUNT_RANGE(j_lo, below_end, <=)
    // the workgroup's own 64 bodies: per-lane tie-break
    const uint32_t own_lo = max(j_lo, i0), own_hi = min(j_hi, min(i0 + 64u, n));
    for (uint32_t j = own_lo; j < own_hi; ++j) {
        const uint64_t kj = keys[j];
        count += (kj < ki || (kj == ki && j < i)) ? 1u : 0u;
    }
    // j above: key_j < key_i
    const uint32_t above_lo = max(j_lo, min(i0 + 64u, n));
    if (above_lo < j_hi) NB_COUNT_RANGE(above_lo, j_hi, <)
#undef NB_COUNT_RANGE
    s_cnt[wave][lane] = count;
    __syncthreads();
    if (wave == 0u && i < n) {
        uint32_t rank = 0;
#pragma unroll
        for (uint32_t w = 0; w < kRankWaves; ++w) rank += s_cnt[w][lane];
        keys_out[rank] = ki;
        order[rank] = i;
    }
}

// ---- 3d. large problems: radix passes over the HIGH digits only, then a fix-up of the ties --------
// With N bodies in a cube, two bodies share the top 8 P bits of their keys only if they sit in the
// same cell of level ~8P/3: for P = 4 that is one of 2^31 cells, so after four stable passes over
// bits 32..62 all but a few hundred of a million uniform bodies are already in their final place,
// and the others form short RUNS of equal high bits (in source-index order, the passes being
// stable) that only need sorting among themselves by the low bits.  That replaces the four low
// passes (12 launches) by one: runs_fix_kernel finds the runs and sorts each in
// place -- a wave per run of <= 64 bodies (rank by counting, keys exchanged by shuffles), a
// workgroup per longer run (counting against the whole run, out of place into the idle ping-pong
// buffer, then copied back).  Any input is sorted correctly; a dense cluster just costs O(L^2)
// compares for a run of L.  The result is the stable full-key order, bit for bit the 8-pass sort's.
constexpr uint32_t kRunWave = 64;

constexpr uint32_t kRunItems = 1;  // positions per thread: a workgroup looks at 256 consecutive positions
// A run longer than this is not ranked by counting (L^2 compares by one workgroup: a dense core of 10^5..10^6
// bodies inside a root cube that a few escapers have stretched -- the normal late state of a gravitational
// run -- would take seconds to minutes) but radix-sorted on its low bits by the workgroup: O(L) per digit.
constexpr uint32_t kRunCountMax = 1024;
// The host's part (TreeSim::wait): the longest run of a step comes back through the status words, and the
// next steps sort one more high digit per kRunBoostAbove exceeded -- the fix-up then sees short runs again;
// `probe` tells it when the extra digits can go.  Speed only: every path gives the stable full-key order.
constexpr uint32_t kRunBoostAbove = 1024, kRunProbeSpan = 512;

// The run [start, start + len) of keys that tie on their high bits, sorted in place by the low `low_bits`
// bits, stably, by one workgroup of 256: LSD radix, 8 bits per pass, between the run's own slots in
// (keys, vals) and in (alt_keys, alt_vals).  A pass = a histogram sweep, a scan of the 256 counts, and a
// scatter sweep in chunks of 256 -- a thread per element, ranked among the chunk's equal digits by wave
// ballots and per-wave counts (the scheme of radix_scatter_kernel).  Digits on which the whole run agrees
// are skipped.
__device__ void run_radix_sort(uint64_t *keys, uint32_t *vals, uint64_t *alt_keys, uint32_t *alt_vals, uint32_t start,
                               uint32_t len, uint32_t low_bits, uint32_t *s_hist, uint32_t (*s_wcnt)[256], uint32_t *s_w,
                               uint32_t *s_flag) {
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    uint64_t *ks = keys + start, *kd = alt_keys + start;
    uint32_t *vs = vals + start, *vd = alt_vals + start;
    bool in_alt = false;
    for (uint32_t sh = 0; sh < low_bits; sh += 8u) {
        const uint32_t dmask = low_bits - sh >= 8u ? 255u : (1u << (low_bits - sh)) - 1u;
        s_hist[tid] = 0u;
        for (uint32_t w = 0; w < 4u; ++w) s_wcnt[w][tid] = 0u;
        if (tid == 0u) *s_flag = 0u;
        __syncthreads();
        for (uint32_t i = tid; i < len; i += 256u) atomicAdd(&s_hist[(uint32_t)(ks[i] >> sh) & dmask], 1u);
        __syncthreads();
        const uint32_t mine = s_hist[tid];
        if (mine == len) *s_flag = 1u;  // every key of the run has this digit: nothing moves
        const uint32_t base = sort_scan_block(mine, s_w);  // (syncs: s_flag is visible after it)
        if (*s_flag) {
            __syncthreads();
            continue;
        }
        s_hist[tid] = base;  // from here on: where the next key with digit tid goes
        __syncthreads();
        for (uint32_t c0 = 0; c0 < len; c0 += 256u) {
            const uint32_t i = c0 + tid;
            const bool valid = i < len;
            const uint64_t key = valid ? ks[i] : 0ull;
            const uint32_t val = valid ? vs[i] : 0u;
            const uint32_t d = (uint32_t)(key >> sh) & dmask;
            uint64_t peers = __ballot(valid);
#pragma unroll
            for (int bb = 0; bb < 8; ++bb) {
                const uint64_t bal = __ballot((d >> bb) & 1u);
                peers &= ((d >> bb) & 1u) ? bal : ~bal;
            }
            const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
            if (valid && rank == 0u) s_wcnt[wave][d] = (uint32_t)__popcll(peers);
            __syncthreads();
            if (valid) {
                uint32_t off = s_hist[d] + rank;
                for (uint32_t w = 0; w < wave; ++w) off += s_wcnt[w][d];
                kd[off] = key;
                vd[off] = val;
            }
            __syncthreads();
            {   // thread t looks after digit t: advance its base, clear the chunk's counts
                uint32_t t = 0u;
                for (uint32_t w = 0; w < 4u; ++w) {
                    t += s_wcnt[w][tid];
                    s_wcnt[w][tid] = 0u;
                }
                s_hist[tid] += t;
            }
            __syncthreads();
        }
        __threadfence_block();
        __syncthreads();
        uint64_t *tk = ks; ks = kd; kd = tk;
        uint32_t *tv = vs; vs = vd; vd = tv;
        in_alt = !in_alt;
    }
    if (in_alt) {  // an odd number of passes moved: the sorted run sits in the alternate buffers
        for (uint32_t i = tid; i < len; i += 256u) {
            kd[i] = ks[i];
            vd[i] = vs[i];
        }
        __threadfence_block();
    }
    __syncthreads();
}

// One launch (it was two -- a kernel listing the runs with aggregated atomics, a kernel sorting them -- and
// the lists needed no more than LDS): a workgroup finds the runs that START among its 256 positions and
// sorts them, short ones (< 64 bodies) a wave each, longer ones one after the other with all its threads.
// (A neighbouring workgroup may still be looking for its run starts while this one already permutes a run:
// it only ever compares the HIGH bits of a key, which a permutation inside a run does not change at any
// position, and an aligned 64-bit load sees one key or the other.)
// stat[0]: the longest run met (atomicMax; the launch of the step before zeroed it: stat_clear = the word
// of the other parity).  stat[2], with probe_bits != 0: set if some run of keys that tie on all but their low
// probe_bits bits is longer than kRunProbeSpan -- what the fix-up would meet with one high digit less.
// (The high-word sort has its own fix-up, a thread per body: runs_rank_kernel, section 3e.)
__global__ __launch_bounds__(256) void runs_fix_kernel(uint64_t *keys, uint32_t *__restrict__ vals,
                                                       uint64_t *__restrict__ alt_keys, uint32_t *__restrict__ alt_vals,
                                                       uint32_t n, uint32_t low_bits, uint32_t probe_bits,
                                                       uint32_t *__restrict__ stat, uint32_t *__restrict__ stat_clear) {
    __shared__ uint32_t s_short[256 * kRunItems], s_long[256 * kRunItems / kRunWave + 1], s_n[3];
    __shared__ uint32_t s_hist[256], s_wcnt[4][256], s_w[4], s_flag;
    // the bits a run ties on, and the coarser ones the probe looks at
    auto high = [&](uint32_t k) -> uint64_t { return keys[k] >> low_bits; };
    auto coarse = [&](uint32_t k) -> uint64_t { return keys[k] >> probe_bits; };
    uint64_t *const run_keys = keys;
    if (threadIdx.x < 3u) s_n[threadIdx.x] = 0u;
    if (blockIdx.x == 0u && threadIdx.x == 0u) stat_clear[0] = stat_clear[2] = stat_clear[4] = 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t c = 0; c < kRunItems; ++c) {
        const uint32_t k = (blockIdx.x * kRunItems + c) * 256u + threadIdx.x;
        if (k + 1u < n) {
            const uint64_t hi = high(k);
            const bool first = k == 0u || high(k - 1u) != hi;
            if (first && high(k + 1u) == hi) {
                // sorted by the high bits: if the body 64 places on still shares them, so do all in between
                if (k + kRunWave < n && high(k + kRunWave) == hi) s_long[atomicAdd(&s_n[1], 1u)] = k;
                else s_short[atomicAdd(&s_n[0], 1u)] = k;
            }
            if (probe_bits && k + kRunProbeSpan < n && coarse(k + kRunProbeSpan) == coarse(k)) s_n[2] = 1u;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0u && s_n[2]) atomicMax(&stat[2], 1u);
    const uint32_t lane = threadIdx.x & 63u, n_short = s_n[0], n_long = s_n[1];
    // short runs: one wave each (the order in which the lists were filled does not matter: the runs are disjoint)
    for (uint32_t r = threadIdx.x >> 6; r < n_short; r += 4u) {
        const uint32_t start = s_short[r];
        const uint64_t hi = high(start);
        const uint32_t pos = start + lane;
        const bool in = pos < n && high(min(pos, n - 1u)) == hi;   // (a run is < 64 long here)
        const uint32_t len = (uint32_t)__popcll(__ballot(in));
        const uint64_t ki = in ? keys[pos] : ~0ull;
        const uint32_t vi = in ? vals[pos] : 0u;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < len; ++j) {
            const uint64_t kj = ((uint64_t)(uint32_t)__shfl((int)(ki >> 32), (int)j) << 32) |
                                (uint32_t)__shfl((int)(uint32_t)ki, (int)j);
            rank += (kj < ki || (kj == ki && j < lane)) ? 1u : 0u;
        }
        __builtin_amdgcn_wave_barrier();
        if (in) {
            keys[start + rank] = ki;
            vals[start + rank] = vi;
        }
    }
    // long runs: the whole workgroup, one after the other
    for (uint32_t r = 0; r < n_long; ++r) {
        const uint32_t start = s_long[r];
        const uint64_t hi = high(start);
        uint32_t lo_s = start + kRunWave, hi_s = n;   // first position past the run: binary search
        while (lo_s < hi_s) {
            const uint32_t mid = lo_s + ((hi_s - lo_s) >> 1);
            if (high(mid) == hi) lo_s = mid + 1u; else hi_s = mid;
        }
        const uint32_t len = lo_s - start;
        if (threadIdx.x == 0u) atomicMax(&stat[0], len);
        if (len > kRunCountMax) {
            run_radix_sort(run_keys, vals, alt_keys, alt_vals, start, len, low_bits, s_hist, s_wcnt, s_w, &s_flag);
            continue;
        }
        for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
            const uint64_t ki = run_keys[start + i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < len; ++j) {
                const uint64_t kj = run_keys[start + j];
                rank += (kj < ki || (kj == ki && j < i)) ? 1u : 0u;
            }
            alt_keys[start + rank] = ki;
            alt_vals[start + rank] = vals[start + i];
        }
        __threadfence_block();
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
            run_keys[start + i] = alt_keys[start + i];
            vals[start + i] = alt_vals[start + i];
        }
        __syncthreads();
    }
}

// ---- 3e. the fix-up of the high-word sort, a thread per body ----------------------------------------------
// After the passes over (high word, index) the array is sorted by the top bits and every body is either alone
// with its high bits or in a run of ties.  Here EVERY body finds its final place by itself: a singleton
// copies its index across; a body in a run shorter than 64 looks left and right for the run's ends on the
// sorted high words, fetches the full keys of the run's members through their indices and counts how many
// come before it -- (key, place in the run) order, the stable order of the full-key sort.  The result goes OUT
// OF PLACE (vals_out), so no body waits for another: where a quarter of the bodies sit in runs of two or
// three -- one radix pass less than runs_fix_kernel's wave-per-run scheme could afford -- this costs what a
// copy of the index array costs plus a few gathers.  Runs of 64 or more (clustered input) are left to the
// workgroup that holds their first body, as in runs_fix_kernel: ranked by counting up to kRunCountMax, radix-
// sorted beyond, and copied to vals_out.  stat / probe: as runs_fix_kernel.
constexpr uint32_t kRankItems = 1;  // positions per thread (4: -10 us at 4,000,000 bodies, +8 us at 131,072 where most bodies sit in runs)
__global__ __launch_bounds__(256) void runs_rank_kernel(const uint32_t *__restrict__ khi, const uint64_t *__restrict__ keys,
                                                        uint32_t *vals_in, uint32_t *vals_out, uint64_t *run_keys,
                                                        uint64_t *alt_keys, uint32_t n, uint32_t low_bits,
                                                        uint32_t probe_bits, uint32_t *__restrict__ stat,
                                                        uint32_t *__restrict__ stat_clear) {
    __shared__ uint32_t s_long[kRankItems * 256 / kRunWave + 1], s_n[2];
    __shared__ uint32_t s_hist[256], s_wcnt[4][256], s_w[4], s_flag;
    const uint32_t hs = low_bits - 32u;
    if (threadIdx.x < 2u) s_n[threadIdx.x] = 0u;
    if (blockIdx.x == 0u && threadIdx.x == 0u) stat_clear[0] = stat_clear[2] = stat_clear[4] = 0u;
    __syncthreads();
    // kRankItems rounds of 256 consecutive positions per workgroup; what every position needs first -- its high
    // word, its neighbours', its index -- is fetched for all rounds together (independent loads in flight
    // together: the kernel is a chain of short dependent loads otherwise)
    uint32_t hw_[kRankItems], hl_[kRankItems], hr_[kRankItems], val_[kRankItems];
#pragma unroll
    for (uint32_t c = 0; c < kRankItems; ++c) {
        const uint32_t k = (blockIdx.x * kRankItems + c) * 256u + threadIdx.x;
        hw_[c] = k < n ? khi[k] : 0u;
        hl_[c] = k > 0u && k < n ? khi[k - 1u] : 0u;
        hr_[c] = k + 1u < n ? khi[k + 1u] : 0u;
        val_[c] = k < n ? vals_in[k] : 0u;
    }
#pragma unroll
    for (uint32_t c = 0; c < kRankItems; ++c) {
        const uint32_t k = (blockIdx.x * kRankItems + c) * 256u + threadIdx.x;
        if (k >= n) continue;
        const uint32_t hw = hw_[c], hi = hw >> hs;
        const bool left = k > 0u && (hl_[c] >> hs) == hi, right = k + 1u < n && (hr_[c] >> hs) == hi;
        if (!left && !right) {
            vals_out[k] = val_[c];
        } else {
            uint32_t s = k, e = k + 1u;  // the run [s, e), as far as it matters: up to kRunWave places either way
            while (s > 0u && k - s < kRunWave && (khi[s - 1u] >> hs) == hi) --s;
            while (e < n && e - s < kRunWave && (khi[e] >> hs) == hi) ++e;
            if (e - s >= kRunWave) {  // a long run: its first body's workgroup sorts it
                if (!left) s_long[atomicAdd(&s_n[0], 1u)] = k;
            } else {
                const uint32_t mine = val_[c];
                const uint64_t ki = keys[mine];
                uint32_t rank = 0u;
                for (uint32_t j = s; j < e; ++j) {
                    const uint64_t kj = keys[vals_in[j]];
                    rank += (kj < ki || (kj == ki && j < k)) ? 1u : 0u;
                }
                vals_out[s + rank] = mine;
            }
        }
        // (the probe of this kernel COUNTS: bodies whose run, with one digit less, would be a long one)
        if (probe_bits && k + kRunWave < n && (khi[k + kRunWave] >> (probe_bits - 32u)) == (hw >> (probe_bits - 32u)))
            atomicAdd(&s_n[1], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0u && s_n[1]) atomicAdd(&stat[2], s_n[1]);
    const uint32_t n_long = s_n[0];
    for (uint32_t r = 0; r < n_long; ++r) {
        const uint32_t start = s_long[r];
        const uint32_t hi = khi[start] >> hs;
        uint32_t lo_s = start + kRunWave, hi_s = n;   // first position past the run: binary search
        while (lo_s < hi_s) {
            const uint32_t mid = lo_s + ((hi_s - lo_s) >> 1);
            if ((khi[mid] >> hs) == hi) lo_s = mid + 1u; else hi_s = mid;
        }
        const uint32_t len = lo_s - start;
        if (threadIdx.x == 0u) {
            atomicMax(&stat[0], len);
            atomicAdd(&stat[4], len);  // bodies that took this slow path
        }
        for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) run_keys[start + i] = keys[vals_in[start + i]];
        __threadfence_block();
        __syncthreads();
        if (len > kRunCountMax) {
            run_radix_sort(run_keys, vals_in, alt_keys, vals_out, start, len, low_bits, s_hist, s_wcnt, s_w, &s_flag);
            for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) vals_out[start + i] = vals_in[start + i];
            __syncthreads();
            continue;
        }
        for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) {
            const uint64_t ki = run_keys[start + i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < len; ++j) {
                const uint64_t kj = run_keys[start + j];
                rank += (kj < ki || (kj == ki && j < i)) ? 1u : 0u;
            }
            vals_out[start + rank] = vals_in[start + i];
        }
        __syncthreads();
    }
}

// ---- 4. gather into sorted (DFS) order ----------------------------------------------------------
// positions/masses first (the build needs them), velocities/accelerations separately (only the
// walk needs them): on several GPUs the second pair is still being all-gathered while the build runs
__global__ void gather_va_kernel(const uint32_t *__restrict__ order, uint32_t n,
                                 const float4 *__restrict__ vel_in, const float4 *__restrict__ acc_in,
                                 float4 *__restrict__ vel_out, float4 *__restrict__ acc_out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t s = order[k];
    vel_out[k] = vel_in[s];
    acc_out[k] = acc_in[s];
}

// one-process multi-GPU runner (nb_group.cpp), replicated tree: the rank's new position / velocity /
// acceleration slices stored into every peer's arrays through peer access, one launch
struct PushDst {
    float4 *p[3][kMaxPeers];
    uint32_t n;
};
__global__ __launch_bounds__(256) void push_slices_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b,
                                                          const float4 *__restrict__ c, PushDst dst, uint32_t first,
                                                          uint32_t count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float4 va = a[first + i], vb = b[first + i], vc = c[first + i];
    for (uint32_t q = 0; q < dst.n; ++q) {
        dst.p[0][q][first + i] = va;
        dst.p[1][q][first + i] = vb;
        dst.p[2][q][first + i] = vc;
    }
}

// ... a few words (the rank's row of an all-gathered LET table) into every peer's copy of the table
struct PushWords {
    uint32_t *p[kMaxPeers];
    uint32_t n;
};
__global__ void push_words_kernel(const uint32_t *__restrict__ src, PushWords dst, uint32_t first, uint32_t count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t v = src[first + i];
    for (uint32_t q = 0; q < dst.n; ++q) dst.p[q][first + i] = v;
}

// ---- 5. cells from key prefixes -----------------------------------------------------------------
// common prefix length in LEVELS of two keys (identical keys are clamped to kLevels-1 so that
// every cell still has a depth <= kLevels; see the header about colliding keys)
__device__ __forceinline__ int cpl_levels(uint64_t a, uint64_t b) {
    const uint64_t x = a ^ b;
    if (x == 0) return kLevels - 1;
    const int lead = __clzll((long long)x) - 1;  // the key occupies bits 62..0
    return lead / 3;
}

// exclusive scan of one value per thread over a 256-thread workgroup
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t *s_wave,
                                                             uint32_t *total) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t x = wave_scan_u32(v);
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    uint32_t off = 0;
    for (uint32_t w = 0; w < wave; ++w) off += s_wave[w];
    if (total) *total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    return off + x - v;
}

// depth of the cells body k opens / owns
__device__ __forceinline__ bool starts_node_at(int left, int right, int d) {
    const bool internal = d > left && d <= right;           // first body of a >=2-body cell
    const bool leaf = d == (left > right ? left : right) + 1;  // alone from this depth on
    return internal || leaf;
}

// the record of an internal cell's slot (cells_c_kernel -> fill_kernel): the node id below its depth
constexpr uint32_t kSlotDepthShift = 27, kSlotIdMask = (1u << kSlotDepthShift) - 1u;

// What the walk reads per cell, in one 32-byte scalar load: centre of gravity + mass, and the
// link {first child id, child count} (leaf: {sorted position of its body, 0}).
struct __attribute__((aligned(32))) NodeRec {
    float4 cogm;
    uint32_t first, count;  // children ids first .. first+count-1 (octant order); leaf: count 0
    uint32_t self_pos;      // leaf: sorted position of its body; cell: ~0 (matches no body)
    float mac2;             // cell: its squared ACCEPTANCE RADIUS, size^2 / theta^2 with size^2 = root_width^2 / 4^depth
                            // (tree.wgsl:82; rounded once here, so that every test of the cell -- each body's own,
                            // the group's all-open shortcut, a LET export's box test -- compares the same number
                            // with its r^2: size/dist < theta (tree.wgsl:63-64) as mac2 < r^2);
                            // leaf: -1, which makes the test always true
};

// ---- 6a. mass moments by prefix sums ------------------------------------------------------------
// A cell's bodies are a contiguous run [k, end) of the sorted order, so its mass and centre of
// gravity follow from exclusive prefix sums of (m x, m y, m z, m) over the sorted bodies:
// sum = P[end] - P[k].  The sums are kept in binary64 -- a difference of fp32 prefix sums would
// lose the small cells at the far end of the array (N eps relative error); in binary64 the
// result is the correctly rounded moment to ~1e-10, where the reference's own sequential fp32
// sum (tree.rs:486-505) is only good to ~1e-6.
struct Moments {
    double x, y, z, m;
};
__device__ __forceinline__ Moments operator+(const Moments &a, const Moments &b) {
    return Moments{a.x + b.x, a.y + b.y, a.z + b.z, a.m + b.m};
}
__device__ __forceinline__ Moments block_scan_moments(Moments v, Moments *s_wave, Moments *total) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const Moments x{wave_scan_f64(v.x), wave_scan_f64(v.y), wave_scan_f64(v.z), wave_scan_f64(v.m)};
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    Moments off{0, 0, 0, 0};
    for (uint32_t w = 0; w < wave; ++w) off = off + s_wave[w];
    if (total) *total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
    return Moments{off.x + x.x - v.x, off.y + x.y - v.y, off.z + x.z - v.z, off.m + x.m - v.m};
}

// ---- 5b/6a fused: cells, node ids and moment prefixes in three launches --------------------------
// Round 1 ran this as thirteen small kernels (gather, cpl, three scans of the opened-cell counts,
// depth histogram + scan + bases, ids, two moment passes + scan); it is one prefix computation
// over the sorted bodies with a 28-word state: 1 count of opened cells, 23 per-depth node counts,
// 4 binary64 moments.  A: per tile of 1,024 bodies, gather + cpl + the tile's totals.  B: the
// tiles' totals scanned, a workgroup per table row and per moment component (fixed order:
// deterministic moments).  C: the depth bases and the node count from the rows' totals, then per
// tile the bodies' own prefixes inside the tile + the tile's offsets -> node ids, slots and moment
// prefixes.  Any number of tiles: no size cap.
constexpr uint32_t kCellTile = 1024;                 // bodies per workgroup of A and C: 4 rounds of 256 (1 round
                                                     // = 256 bodies on small problems, which are bound by the
                                                     // chain of barriers inside a workgroup, not by work)
constexpr uint32_t kCellRows = kMaxDepth + 2;        // u32 rows of the tile table: [0] nint, [1 + d] depth d

// GATHER (section 3e: the sort moved high words and indices only): `keys` are the UNSORTED keys, a body's key is
// gathered through `order` like its position, and the sorted key array the later kernels search is written
// here (keys_out); the neighbours' keys come from the neighbouring lanes.
template <bool GATHER>
__global__ __launch_bounds__(256) void cells_a_kernel(
    const uint32_t *__restrict__ order, uint32_t n, const float4 *__restrict__ posm_in,
    float4 *__restrict__ posm_out, const uint64_t *__restrict__ keys, uint64_t *__restrict__ keys_out,
    int8_t *__restrict__ cpl,
    uint32_t *__restrict__ tile_u32, Moments *__restrict__ tile_mom, uint32_t stride, uint32_t rounds,
    uint32_t *__restrict__ status) {
    __shared__ uint32_t s_hist[kCellRows];
    __shared__ Moments s_wave[4];
    if (threadIdx.x < kCellRows) s_hist[threadIdx.x] = 0;
    __syncthreads();
    Moments msum{0, 0, 0, 0};
    uint32_t nint_sum = 0, collide = 0;
    for (uint32_t sub = 0; sub < rounds; ++sub) {
        const uint32_t k = (blockIdx.x * rounds + sub) * 256u + threadIdx.x;
        Moments item{0, 0, 0, 0};
        uint64_t me = 0, me_prev = 0, me_next = 0;
        uint32_t src = 0;
        if (k < n) {
            src = order[k];
            me = GATHER ? keys[src] : keys[k];
        }
        if (GATHER) {  // (outside the bounds check: every lane takes part in the shuffles)
            const uint32_t lane = threadIdx.x & 63u;
            me_prev = ((uint64_t)(uint32_t)__shfl_up((int)(me >> 32), 1) << 32) | (uint32_t)__shfl_up((int)(uint32_t)me, 1);
            me_next = ((uint64_t)(uint32_t)__shfl_down((int)(me >> 32), 1) << 32) | (uint32_t)__shfl_down((int)(uint32_t)me, 1);
            if (lane == 0u && k > 0u && k < n) me_prev = keys[order[k - 1u]];
            if (lane == 63u && k + 1u < n) me_next = keys[order[k + 1u]];
        }
        if (k < n) {
            const float4 p = posm_in[src];  // sort_particles, tree.rs:564-602
            posm_out[k] = p;
            const double m = (double)p.w;
            item = Moments{(double)p.x * m, (double)p.y * m, (double)p.z * m, m};
            if (GATHER) keys_out[k] = me;
            else {
                me_prev = k > 0 ? keys[k - 1] : 0ull;
                me_next = k + 1 < n ? keys[k + 1] : 0ull;
            }
            const int left = k > 0 ? cpl_levels(me_prev, me) : -1;
            // (a lone body: the reference's root is always an internal octant -- the queue starts with the
            // root partition whatever it holds, tree.rs:463-476 -- so the body's leaf sits at depth 1)
            const int right = k + 1 < n ? cpl_levels(me, me_next) : (n == 1u ? 0 : -1);
            if (k == 0) cpl[0] = -1;
            cpl[k + 1] = (int8_t)right;
            nint_sum += right > left ? (uint32_t)(right - left) : 0u;  // internal cells this body opens
            // (LDS atomics, 256 of a round on two or three words: a loop over the wave's depths with ballots, one add
            // per wave and depth, measured SLOWER -- 15.9 -> 19.3 us at 2^20 bodies)
            for (int d = left + 1; d <= right; ++d) atomicAdd(&s_hist[1 + d], 1u);
            atomicAdd(&s_hist[1 + (left > right ? left : right) + 1], 1u);  // its leaf
            if (k + 1 < n && me_next == me) collide += 1u;
        }
        msum = msum + item;  // (per thread over its rounds; the workgroup's total once, below)
    }
    {   // the tile's moments: the threads' sums added in a fixed order (wave scan, then the waves in order)
        Moments total;
        (void)block_scan_moments(msum, s_wave, &total);
        msum = total;
    }
    if (nint_sum) atomicAdd(&s_hist[0], nint_sum);
    if (collide) atomicAdd(&status[2], collide);
    __syncthreads();
    if (threadIdx.x < kCellRows) tile_u32[(size_t)threadIdx.x * stride + blockIdx.x] = s_hist[threadIdx.x];
    if (threadIdx.x == 0) tile_mom[blockIdx.x] = msum;
}

// B: exclusive scan over the tiles of every row.  A workgroup per row (blockIdx.x < kCellRows: the row's
// total goes to row_total[row]; the depth bases that follow from the totals are derived by C itself) and
// four more for the four moment sums -- round 2's first form did all rows in ONE workgroup, 10 us at 2^20 bodies, 31 at
// 4 M, 124 at 16 M; the rows do not depend on each other.
__global__ __launch_bounds__(1024) void cells_scan_kernel(uint32_t *__restrict__ tile_u32,
                                                         Moments *__restrict__ tile_mom, uint32_t ntiles,
                                                         uint32_t stride, uint32_t *__restrict__ row_total,
                                                         uint32_t *__restrict__ bound_slots) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (blockIdx.x < kCellRows) {
        // the row in chunks of 4,096 tiles: a wave 256 of them, every lane 4 consecutive tiles (one 16-byte
        // access; rows are padded to a multiple of 4 words), a wave scan of the lane sums, the waves'
        // totals through LDS, a carry
        __shared__ uint32_t s_w[16];
        uint4 *row = reinterpret_cast<uint4 *>(tile_u32 + (size_t)blockIdx.x * stride);
        uint32_t carry = 0;
        for (uint32_t base = 0; base < stride; base += 4096u) {
            const uint32_t i4 = base / 4u + threadIdx.x;
            uint4 v{0u, 0u, 0u, 0u};
            if (i4 * 4u < stride) v = row[i4];
            if (i4 * 4u + 0u >= ntiles) v.x = 0u;  // (the padding of the row was never written)
            if (i4 * 4u + 1u >= ntiles) v.y = 0u;
            if (i4 * 4u + 2u >= ntiles) v.z = 0u;
            if (i4 * 4u + 3u >= ntiles) v.w = 0u;
            const uint32_t sum = v.x + v.y + v.z + v.w;
            const uint32_t x = wave_scan_u32(sum);
            if (lane == 63u) s_w[wave] = x;
            __syncthreads();
            uint32_t before = 0u, chunk_total = 0u;
            for (uint32_t w = 0; w < 16u; ++w) {
                before += w < wave ? s_w[w] : 0u;
                chunk_total += s_w[w];
            }
            const uint32_t run = carry + before + x - sum;
            if (i4 * 4u < stride) row[i4] = uint4{run, run + v.x, run + v.x + v.y, run + v.x + v.y + v.z};
            carry += chunk_total;
            __syncthreads();  // s_w is reused
        }
        if (threadIdx.x == 0u) row_total[blockIdx.x] = carry;
        return;
    }
    const uint32_t comp = blockIdx.x - kCellRows;  // 0..3: m x, m y, m z, m -- a workgroup per component
    if (comp == 0u)  // this step's walk accumulates the next bound
        for (uint32_t k = threadIdx.x; k < kBoundSlots; k += 1024u) bound_slots[k] = 0u;
    {   // the moments, by the 1,024 threads in a fixed order: thread t sums the tiles [t S, (t+1) S) in
        // order, the threads' sums are scanned by wave (fixed shuffle tree) and the waves' totals
        // added in wave order -- deterministic whatever the launch timing
        __shared__ double s_wtot[16];
        double *vals = reinterpret_cast<double *>(tile_mom) + comp;  // stride 4 doubles
        const uint32_t per = (ntiles + 1023u) / 1024u;
        const uint32_t t_lo = min(threadIdx.x * per, ntiles), t_hi = min(t_lo + per, ntiles);
        double sum = 0.0;
        for (uint32_t i = t_lo; i < t_hi; ++i) sum += vals[4u * (size_t)i];
        const double x = wave_scan_f64(sum);
        if (lane == 63u) s_wtot[wave] = x;
        __syncthreads();
        double run = 0.0;
        for (uint32_t w = 0; w < wave; ++w) run += s_wtot[w];
        run += x - sum;
        for (uint32_t i = t_lo; i < t_hi; ++i) {
            const double v = vals[4u * (size_t)i];
            vals[4u * (size_t)i] = run;
            run += v;
        }
    }
}

// C: node ids (rank of (body k, depth d) among the nodes of depth d in key order = the reference's
// BFS allocation order), slots of the opened cells, moment prefixes
// SCAN_INLINE (up to kCellInlineTiles tiles: the sizes at which a step is a chain of launch latencies): B inside C.
// The tile table comes as cells_a_kernel wrote it and every workgroup sums the tiles before its own itself --
// the u32 rows by a lane per row and eighth of the tiles, the moments by a thread per tile in
// cells_scan_kernel's own order of additions (wave scan, then the waves in order: the same bits) -- one
// dependent launch fewer per step.  Per runner.step(), theta 0.75, B inside C / B launched: 1,024 bodies
// 51.1 / 54.1 us, 4,096: 60.3 / 65.7, 8,192: 70.5 / 75.7, 12,288: 79.5 / 83.0; 16,384 (65 tiles): 84.8 / 85.2,
// 32,768: 98.0 / 98.8, 65,279 (255 tiles): 123.8 / 122.5 -- the code handles up to 256 tiles, the host uses it to 64.
constexpr uint32_t kCellInlineTiles = 64;
template <bool SCAN_INLINE>
__global__ __launch_bounds__(256) void cells_c_kernel(
    const int8_t *__restrict__ cpl, uint32_t n, const uint32_t *__restrict__ tile_u32,
    const Moments *__restrict__ tile_mom, uint32_t stride, uint32_t *__restrict__ row_total,
    uint32_t *__restrict__ depth_base, uint32_t *__restrict__ n_nodes, uint32_t *__restrict__ status,
    const float4 *__restrict__ posm, uint32_t *__restrict__ int_slot, uint32_t *__restrict__ leaf_id,
    uint2 *__restrict__ int_id, uint32_t *__restrict__ node_first, uint8_t *__restrict__ node_depth,
    Moments *__restrict__ prefix, uint32_t cap, uint32_t rounds, const uint32_t *__restrict__ order,
    const float4 *__restrict__ vel_in, const float4 *__restrict__ acc_in, float4 *__restrict__ vel_out,
    float4 *__restrict__ acc_out, NodeRec *__restrict__ rec, uint32_t *__restrict__ bound_slots) {
    __shared__ uint32_t s_cnt[4][kMaxDepth + 1], s_run[kMaxDepth + 1], s_scan[4];
    __shared__ Moments s_wave[4];
    __shared__ uint32_t s_before[8][kCellRows], s_all[8][kCellRows];
    auto sum8 = [](const uint32_t (*a)[kCellRows], uint32_t r) {
        return a[0][r] + a[1][r] + a[2][r] + a[3][r] + a[4][r] + a[5][r] + a[6][r] + a[7][r];
    };
    __shared__ Moments s_mom_run;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    if (SCAN_INLINE) {
        const uint32_t ntiles = gridDim.x;
        if ((lane & 31u) < kCellRows) {  // row `lane & 31`, this half-wave's eighth of the tiles
            const uint32_t r = lane & 31u, part = wave * 2u + (lane >> 5);
            const uint32_t q = (ntiles + 7u) / 8u, t_lo = min(part * q, ntiles), t_hi = min(t_lo + q, ntiles);
            const uint32_t *row = tile_u32 + (size_t)r * stride;
            uint32_t before = 0u, all = 0u;
#pragma unroll 4
            for (uint32_t t = t_lo; t < t_hi; ++t) {
                const uint32_t v = row[t];
                all += v;
                before += t < blockIdx.x ? v : 0u;
            }
            s_before[part][r] = before;
            s_all[part][r] = all;
        }
        Moments v{0, 0, 0, 0};
        if (threadIdx.x < ntiles) v = tile_mom[threadIdx.x];
        const Moments x{wave_scan_f64(v.x), wave_scan_f64(v.y), wave_scan_f64(v.z), wave_scan_f64(v.m)};
        if (lane == 63u) s_wave[wave] = x;
        __syncthreads();
        if (threadIdx.x == blockIdx.x) {
            Moments run{0, 0, 0, 0};
            for (uint32_t w = 0; w < wave; ++w) run = run + s_wave[w];
            s_mom_run = Moments{run.x + (x.x - v.x), run.y + (x.y - v.y), run.z + (x.z - v.z), run.m + (x.m - v.m)};
        }
        if (blockIdx.x == 0u) {
            if (threadIdx.x < kCellRows)
                row_total[threadIdx.x] = sum8(s_all, threadIdx.x);
            for (uint32_t k = threadIdx.x; k < kBoundSlots; k += 256u) bound_slots[k] = 0u;
        }
    }
    if (wave == 0u) {
        // depth_base[d] = nodes of depth < d, from the rows' totals (row 1 + d = depth d); [kMaxDepth + 1] = the
        // node count.  Every workgroup derives them for itself; the first one publishes them for the kernels
        // that follow (fill, LET export, read-out) and checks the 4N capacity.
        uint32_t mine = 0u;
        if (lane <= (uint32_t)kMaxDepth)
            mine = SCAN_INLINE ? sum8(s_all, 1u + lane) : row_total[1u + lane];
        uint32_t x = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o);
            if ((int)lane >= o) x += y;
        }
        const uint32_t base_d = x - mine;  // exclusive
        if (lane <= (uint32_t)kMaxDepth)  // where this tile's nodes of depth d start
            s_run[lane] = base_d + (SCAN_INLINE ? sum8(s_before, 1u + lane)
                                                : tile_u32[(size_t)(1u + lane) * stride + blockIdx.x]);
        if (blockIdx.x == 0u) {
            if (lane <= (uint32_t)kMaxDepth) depth_base[lane] = base_d;
            if (lane == (uint32_t)kMaxDepth) {
                depth_base[kMaxDepth + 1] = x;
                *n_nodes = x;
                if (x > cap) atomicAdd(&status[1], 1u);
            }
        }
    }
    uint32_t slot_run = 0u;  // row 0: opened cells before this tile
    Moments mom_run{0, 0, 0, 0};
    if (!SCAN_INLINE) {
        slot_run = tile_u32[blockIdx.x];
        mom_run = tile_mom[blockIdx.x];
    }
    __syncthreads();
    if (SCAN_INLINE) {
        slot_run = sum8(s_before, 0u);
        mom_run = s_mom_run;
        __syncthreads();  // s_wave is reused by the rounds below
    }
    for (uint32_t sub = 0; sub < rounds; ++sub) {
        const uint32_t k = (blockIdx.x * rounds + sub) * 256u + threadIdx.x;
        const bool valid = k < n;
        const int left = valid ? cpl[k] : 0, right = valid ? cpl[k + 1] : 0;
        const int leafd = (left > right ? left : right) + 1;
        // The depths at which the wave's 64 bodies start a node at all (neighbours in tree order sit at similar
        // depths: typically 5 or 6 of the 23).  The counts of the other depths are zero; the ranks inside the wave
        // are not kept but counted again when the ids are written (23 live registers and two unrolled 23-step loops
        // otherwise: 95 VGPRs, 3,800 instructions).
        int d_lo = valid ? (right > left ? left + 1 : leafd) : kMaxDepth + 1, d_hi = valid ? leafd : -1;
        d_lo = __builtin_amdgcn_readlane(wave_min_to_lane63(d_lo), 63);
        d_hi = __builtin_amdgcn_readlane(wave_max_to_lane63(d_hi), 63);
        if (lane <= (uint32_t)kMaxDepth) s_cnt[wave][lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        for (int d = d_lo; d <= d_hi; ++d) {
            const uint64_t bal = __ballot(valid && starts_node_at(left, right, d));
            if (lane == 0) s_cnt[wave][d] = (uint32_t)__popcll(bal);
        }
        const uint32_t ni = valid && right > left ? (uint32_t)(right - left) : 0u;
        Moments item{0, 0, 0, 0};
        float4 p{0.f, 0.f, 0.f, 0.f};
        if (valid) {
            p = posm[k];
            const double m = (double)p.w;
            item = Moments{(double)p.x * m, (double)p.y * m, (double)p.z * m, m};
            if (vel_in) {  // the rest of sort_particles (tree.rs:564-602): velocities and accelerations
                const uint32_t src = order[k];
                vel_out[k] = vel_in[src];
                acc_out[k] = acc_in[src];
            }
        }
        // the opened-cell count and the four moments scanned over the workgroup together: the waves' totals of
        // both meet in LDS behind ONE pair of barriers (two scans, two pairs, before)
        uint32_t ni_total, slot0;
        Moments mom_total, mom0;
        {
            const uint32_t xi = wave_scan_u32(ni);
            const Moments xm{wave_scan_f64(item.x), wave_scan_f64(item.y), wave_scan_f64(item.z), wave_scan_f64(item.m)};
            if (lane == 63u) {
                s_scan[wave] = xi;
                s_wave[wave] = xm;
            }
            __syncthreads();
            uint32_t offi = 0u;
            Moments offm{0, 0, 0, 0};
            for (uint32_t w = 0; w < wave; ++w) {
                offi += s_scan[w];
                offm = offm + s_wave[w];
            }
            ni_total = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
            mom_total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
            __syncthreads();
            slot0 = slot_run + offi + xi - ni;
            mom0 = mom_run + Moments{offm.x + xm.x - item.x, offm.y + xm.y - item.y, offm.z + xm.z - item.z,
                                     offm.m + xm.m - item.m};
        }
        if (k <= n) prefix[k] = mom0;  // includes prefix[n] = the grand total
        if (valid) int_slot[k] = slot0;
        for (int d = d_lo; d <= d_hi; ++d) {
            const bool st = valid && starts_node_at(left, right, d);
            const uint64_t bal = __ballot(st);
            uint32_t before = s_run[d];  // (wave-uniform: where the wave's nodes of depth d start)
            for (uint32_t w = 0; w < wave; ++w) before += s_cnt[w][d];
            if (st) {
                const uint32_t id = before + (uint32_t)__popcll(bal & lt_mask);
                if (d == leafd) {
                    leaf_id[k] = id;
                    // the walk's record of the leaf (tree.rs:521-534: cog = position, mass): written here, where
                    // the body is in registers, so that fill_kernel runs over the internal cells only
                    if (id < cap) rec[id] = NodeRec{p, 0u, 0u, k, -1.0f};
                } else {
                    // (a clustered input can open far more internal cells than the 4N capacity)
                    const uint32_t slot = slot0 + (uint32_t)(d - left - 1);
                    // the slot's record: what fill_kernel needs to start on the cell without looking anything up --
                    // {first body | 'body k opens the next depth too' << 31, id | depth << 27}
                    if (slot < cap) int_id[slot] = uint2{k | (d + 1 <= right ? 0x80000000u : 0u), id | ((uint32_t)d << kSlotDepthShift)};
                }
                if (id < cap) {
                    node_first[id] = k;
                    node_depth[id] = (uint8_t)(d | (d == leafd ? 0x80 : 0));
                }
            }
        }
        __syncthreads();
        if (threadIdx.x <= kMaxDepth)
            s_run[threadIdx.x] += s_cnt[0][threadIdx.x] + s_cnt[1][threadIdx.x] + s_cnt[2][threadIdx.x] +
                                  s_cnt[3][threadIdx.x];
        slot_run += ni_total;
        mom_run = mom_run + mom_total;
        __syncthreads();
    }
}

// ---- 6. node contents ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lower_bound_key(const uint64_t *keys, uint32_t lo, uint32_t hi,
                                                    uint64_t v) {  // first k in [lo,hi) with key >= v
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (keys[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// AOS = false (every step): only the 32-byte walk records.  AOS = true (nb_sim_read_tree, on
// demand): also the reference's Octant fields -- cog, body count, the 8-entry children table
// indexed by octant -- which cost two more dependent loads per child and 52 B of stores per node.
constexpr uint32_t kFillEagerMax = 262144;  // bodies up to which fill_kernel fetches speculatively
// ... and from which it does again, all but the moment prefix: there the kernel waits out a dozen dependent loads
// per cell with every CU busy, and the probes of the run search and the candidate children sit in the lines the
// cell reads anyway (build 0.625 -> 0.605 ms at 4,000,000 bodies, 2.65 -> 2.63 at 16,777,216; 0.179 -> 0.181 at 2^20)
constexpr uint32_t kFillEagerAgainFrom = 2097152;

// EAGER_MOM: also the first moment prefix ahead of the search (small problems only: see below)
// !AOS: a thread per INTERNAL cell, found through its slot (int_id[slot], slots counted by cells_a/cells_c: row 0 of
// the tile table): two thirds of the nodes are leaves, whose records cells_c_kernel has already written, and a
// wave of 64 internal cells does not wait for the long chain of a few of them while most of its lanes idle.
template <bool AOS, bool EAGER, bool EAGER_MOM = EAGER>
__global__ void fill_kernel(const uint64_t *__restrict__ keys, uint32_t n, uint32_t n_cap,
                            const uint32_t *__restrict__ n_nodes_p,
                            const uint32_t *__restrict__ node_first,
                            const uint8_t *__restrict__ node_depth, const int8_t *__restrict__ cpl,
                            const uint32_t *__restrict__ int_slot,
                            const uint32_t *__restrict__ leaf_id, const uint2 *__restrict__ int_id,
                            const uint32_t *__restrict__ order, const float4 *__restrict__ posm,
                            const Moments *__restrict__ mom, const uint32_t *__restrict__ depth_base,
                            const uint32_t *__restrict__ bound_bits,
                            float4 *__restrict__ cogm, uint32_t *__restrict__ bodies,
                            uint32_t *__restrict__ child, NodeRec *__restrict__ rec, float inv_theta2,
                            const uint32_t *__restrict__ n_internal_p) {
    // (the grid covers ~1.75 N nodes / ~0.75 N internal cells -- a uniform octree has ~1.5 N / 0.5 N, the
    // capacity is 4 N and the counts are only known on the device: the workgroups that would find nothing to do
    // are not launched, a deeper tree takes the loop)
    const uint32_t n_nodes = min(*n_nodes_p, n_cap);
    const uint32_t n_work = AOS ? n_nodes : min(*n_internal_p, n_cap);
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_work; t += gridDim.x * blockDim.x) {
        // !AOS: everything the cell starts from comes in its slot record (written by cells_c_kernel): two dependent
        // look-ups (id -> first body, depth) and the two prefix lengths of the body fewer per cell
        uint2 si{0u, 0u};
        if (!AOS) si = int_id[t];
        const uint32_t id = AOS ? t : si.y & kSlotIdMask;
        if (id >= n_nodes) continue;
        const uint32_t k = AOS ? node_first[id] : si.x & 0x7fffffffu;
        const uint32_t dd = AOS ? node_depth[id] : si.y >> kSlotDepthShift;
        uint32_t ch[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (dd & 0x80) {  // leaf: cog = position, mass, bodies = 1, children[0] = source index
            // (AOS, the read-out after a step: from the record cells_c_kernel wrote -- a walk that gathers velocities
            // has put the NEW positions where the sorted source stood)
            const float4 p = AOS ? rec[id].cogm : posm[k];
            if (AOS) {
                cogm[id] = p;
                bodies[id] = 1;
                ch[0] = order[k];  // tree.rs:532
            }
            rec[id] = NodeRec{p, 0u, 0u, k, -1.0f};  // walk: a leaf knows its body's sorted position
        } else {
            const uint32_t d = dd;
            const uint32_t shift = 3u * (uint32_t)(kLevels - d);  // bits below the depth-d prefix
            // (Small problems are bound by this kernel's chain of dependent loads, not by its work: what
            // depends only on k is fetched together and, EAGER, the first steps of the search and the
            // eight candidate children likewise -- 6 loads deep instead of ~15: 11.6 -> 9.8 us at 16,384
            // bodies.  At 2^20 bodies the kernel is bound by HBM traffic and the speculative loads cost
            // 6 us: not EAGER there.)
            const uint64_t key_k = keys[k];
            int left = 0, right = 0;
            if (AOS) {
                left = cpl[k];
                right = cpl[k + 1];
            }
            // body k also opens the cell one level down: its slot is the next one (a body's cells have consecutive slots)
            const bool opens_next = AOS ? (int)d + 1 <= right : (si.x >> 31) != 0u;
            // (only one of the two is needed: both are fetched ahead only where latency, not traffic, binds)
            const uint32_t slot_k = AOS && opens_next ? int_slot[k] : 0u;
            const uint32_t next_id = !AOS && (EAGER || opens_next) && t + 1u < n_cap ? int_id[t + 1u].y & kSlotIdMask : ~0u;
            const uint32_t leaf_k = (EAGER || !opens_next) ? leaf_id[k] : 0u;
            Moments a{0, 0, 0, 0};
            if (EAGER_MOM) a = mom[k];  // (large problems: beside mom[end] below -- mostly the same cache line, and
                                    // fetched apart it has left the L2 by then: 185 -> 241 MB of HBM reads at 2^20)
            // end of the cell's run: galloping search from k (most cells hold a handful of bodies)
            uint32_t end = n;
            if (d != 0) {
                const uint64_t limit = ((key_k >> shift) + 1ull) << shift;  // first key past the cell
                uint32_t lo_s = k + 1u, off = 1u;
                if (EAGER) {
                    uint64_t probe[4];
    #pragma unroll
                    for (int q = 0; q < 4; ++q) probe[q] = keys[min(k + (1u << q), n - 1u)];  // k+1, k+2, k+4, k+8
    #pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (off == (1u << q) && k + off < n && probe[q] < limit) {
                            lo_s = k + off + 1u;
                            off <<= 1;
                        }
                    }
                }
                if (!EAGER || off == 16u) {
                    while (k + off < n && keys[k + off] < limit) {
                        lo_s = k + off + 1u;
                        off <<= 1;
                    }
                }
                end = lower_bound_key(keys, lo_s, min(k + off, n), limit);
            }
            if (AOS) bodies[id] = end - k;
            // children: the depth-(d+1) nodes whose first body lies in [k, end) -- consecutive ids
            // (nodes of one depth are numbered in key order), starting with body k's own child
            uint32_t f;
            if (opens_next && AOS) {
                const uint32_t slot = slot_k + (d - (uint32_t)(left + 1) + 1u);
                f = slot < n_cap ? int_id[slot].y & kSlotIdMask : ~0u;
            } else if (opens_next) {
                f = next_id;
            } else {
                f = leaf_k;
            }
            const uint32_t lim = min(depth_base[d + 2], n_nodes);  // end of the depth-(d+1) ids
            uint32_t first = 0, cnt = 0;
            if (!EAGER) {
                for (uint32_t j = 0; j < 8u; ++j) {
                    const uint32_t cid = f + j;
                    if (f == ~0u || cid >= lim) break;
                    const uint32_t kc = node_first[cid];
                    if (j > 0 && kc >= end) break;
                    if (AOS) ch[(uint32_t)(keys[kc] >> (shift - 3u)) & 7u] = cid;  // octant = the key digit of level d
                    if (cnt == 0u) first = cid;
                    ++cnt;
                }
            } else if (f != ~0u) {
                uint32_t kc[8];
    #pragma unroll
                for (uint32_t j = 0; j < 8u; ++j) kc[j] = node_first[min(f + j, n_nodes - 1u)];
                bool more = true;
    #pragma unroll
                for (uint32_t j = 0; j < 8u; ++j) {
                    const uint32_t cid = f + j;
                    more = more && cid < lim && (j == 0u || kc[j] < end);
                    if (more) {
                        if (AOS) ch[(uint32_t)(keys[kc[j]] >> (shift - 3u)) & 7u] = cid;  // octant = the key digit of level d
                        if (cnt == 0u) first = cid;
                        ++cnt;
                    }
                }
            }
            // mass and centre of gravity of the run [k, end)   (tree.rs:486-505)
            if (!EAGER_MOM) a = mom[k];
            const Moments b2 = mom[end];
            const double m = b2.m - a.m;
            const float4 q = float4{(float)((b2.x - a.x) / m), (float)((b2.y - a.y) / m),
                                    (float)((b2.z - a.z) / m), (float)m};
            if (AOS) cogm[id] = q;
            // children are allocated contiguously in octant order (tree.rs:517-519), so the walk
            // only needs the first child's id and how many there are
            // a tree that outgrew its 4N capacity (status[1]) keeps the walk in bounds: a cell whose
            // children were not all stored is walked as a single body of the cell's mass
            // ... and children always carry larger ids than their parent (breadth-first numbering), which
            // is what lets the walk terminate without a visit budget: enforce it here
            if (cnt == 0u || first + cnt > n_nodes || first <= id) {
                rec[id] = NodeRec{q, 0u, 0u, ~0u, -1.0f};
            } else {
                const float root_width = __uint_as_float(*bound_bits) * 2.0f;
                float size2 = root_width * root_width;
                for (uint32_t l = 0; l < d; ++l) size2 *= 0.25f;  // exact: the width halves per level
                rec[id] = NodeRec{q, first, cnt, ~0u, size2 * inv_theta2};
            }
        }
        if (AOS) {
    #pragma unroll
            for (int c = 0; c < 8; ++c) child[(size_t)id * 8 + c] = ch[c];
        }
    }
}

// ---- 8. walk + integrate ------------------------------------------------------------------------
// Stack entry: a SIBLING GROUP -- the children first .. first+count-1 of one opened cell (their
// ids are consecutive, octant order) -- and the 64-bit mask of the lanes that opened it.  One
// entry per opened cell instead of one per child: a third of the LDS traffic and of the
// lane-0 read-outs, and the children's records sit back to back in memory.
struct StackEntry {
    uint32_t first, count;
    uint32_t mask_lo, mask_hi;
};
constexpr uint32_t kWalkBatch = 4;  // records fetched together (a group is 1..8 cells)

// The trees a wave walks: its own (record 0) and, on a multi-GPU run, the imported locally
// essential trees of the peers (section 9).
constexpr int kLetMaxWorld = 16;
struct WalkRoots {
    uint32_t count;
    uint32_t id[kLetMaxWorld];
};

struct WalkStats {
    unsigned long long visits = 0, accepts = 0;
    uint32_t wave_cells = 0, wave_leaves = 0, max_sp = 1;
};

// K consecutive cells of one sibling group: their records are fetched together (wave-uniform
// address + immediate offsets: scalar loads), then each is tested and accumulated by every
// lane.  Straight-line per K so that no per-cell loop control or index clamping is needed, and
// light on SCALAR work (the scalar unit is what the loop saturates first): no per-lane
// branches, a leaf and a cell take the same path (a leaf's record makes the acceptance test
// always true and carries the one body position it must skip), the force is predicated
// instead of branched around, and the lane sets are 64-bit masks combined by s_and/s_andn2.
template <uint32_t K, bool COUNT>
__device__ __forceinline__ void walk_cells(const NodeRec *__restrict__ rp, uint64_t gmask, uint32_t i,
                                           float xi, float yi, float zi, float e,
                                           float &ax, float &ay, float &az, StackEntry *stack,
                                           uint32_t &sp, bool lane0, WalkStats &st) {
    NodeRec r[K];
#pragma unroll
    for (uint32_t b = 0; b < K; ++b) r[b] = rp[b];
#pragma unroll
    for (uint32_t b = 0; b < K; ++b) {
        const float4 q = r[b].cogm;
        const float dx = q.x - xi, dy = q.y - yi, dz = q.z - zi;
        const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        // acceptance size/dist < theta (tree.wgsl:63-64) as size^2 / theta^2 < r^2; a leaf's
        // negative radius makes it always true, its self_pos excludes the body itself
        const uint64_t far = __ballot(r[b].mac2 < r2);
        const uint64_t other = __ballot(r[b].self_pos != i);
        const uint64_t take = gmask & far & other;
        const uint64_t open = gmask & ~far;  // never a leaf: its test is always true
        const float dist = __builtin_amdgcn_sqrtf(r2);
        float w = q.w * __builtin_amdgcn_rcpf(__builtin_fmaf(e, dist, r2 * r2));
        w = __builtin_amdgcn_inverse_ballot_w64(take) ? w : 0.0f;  // predicated, not branched
        ax = __builtin_fmaf(w, dx, ax);
        ay = __builtin_fmaf(w, dy, ay);
        az = __builtin_fmaf(w, dz, az);
        if (COUNT) {
            st.visits += __builtin_amdgcn_inverse_ballot_w64(gmask) ? 1ull : 0ull;
            st.accepts += __builtin_amdgcn_inverse_ballot_w64(take) ? 1ull : 0ull;
            if (r[b].count == 0u) st.wave_leaves += 1u;
        }
        if (open) {  // push the cell's children as one group for the opening lanes
            if (lane0)
                stack[sp] = StackEntry{r[b].first, r[b].count, (uint32_t)open, (uint32_t)(open >> 32)};
            sp += 1;
            if (COUNT) st.max_sp = sp > st.max_sp ? sp : st.max_sp;
        }
    }
}

// One wave walks for 64 consecutive sorted bodies, depth-first over sibling groups: a pop
// pushes at most 8 groups one level down, so the stack holds at most 7 x 21 + 1 entries -- it
// cannot overflow.
// PART: 0 = the whole step; 1 = walk the given trees and leave the raw sums in acc_dst (no
// integration); 2 = start from those sums, walk the given trees, integrate.  1 then 2 add the same
// terms in the same order as 0 does over the concatenated roots, so the result is bit-identical --
// a LET host walks the rank's own tree (1) while the imported trees are still on the wire.
template <bool COUNT, int PART = 0>
__global__ __launch_bounds__(256) void walk_kernel(
    const float4 *__restrict__ posm_src, const float4 *__restrict__ vel_src,
    const float4 *__restrict__ acc_src, const NodeRec *__restrict__ rec,
    WalkRoots roots_arg,
    float4 *__restrict__ posm_dst, float4 *__restrict__ vel_dst, float4 *__restrict__ acc_dst,
    uint32_t lo, uint32_t hi, uint32_t bpw_shift, float g, float e, float dt,
    uint32_t *__restrict__ status, unsigned long long *__restrict__ counters,
    uint32_t *__restrict__ bound_slots, const WalkRoots *__restrict__ roots_dev) {
    // (device-made roots: fixed-stride LET imports.  Element-wise, never a copy of the struct: a
    // by-value copy of a kernel argument selected at run time lands in scratch memory)
    const uint32_t n_roots =  // (readfirstlane: see walk_cells_kernel)
        (uint32_t)__builtin_amdgcn_readfirstlane((int)(roots_dev ? roots_dev->count : roots_arg.count));
    __shared__ StackEntry s_stack[4][kWalkStack];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    // this rank walks for the sorted bodies [lo, hi) (single GPU: [0, n)).
    // Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2); remap so that each
    // XCD walks one contiguous eighth of the Morton-ordered bodies and its L2 keeps that region's
    // deep cells instead of everybody's.  Speed only: any placement gives the same result.
    const uint32_t per_xcd = gridDim.x / 8u;
    uint32_t blk = blockIdx.x;
    if (blk < per_xcd * 8u) blk = (blk & 7u) * per_xcd + (blk >> 3);   // bijective on [0, 8*per_xcd)
                                                                       // the last < 8 blocks stay put
    // A wave walks for 2^bpw_shift consecutive bodies (64 on large problems; fewer when there are
    // not enough bodies to fill the chip: a small problem is bound by the LENGTH of one wave's
    // walk, and the union of the cells of 8 bodies is much shorter than that of 64).
    const uint32_t i = lo + ((blk * 4u + wave) << bpw_shift) + lane;
    const bool valid = i < hi && lane < (1u << bpw_shift);
    const uint32_t ic = valid ? i : hi - 1;
    const float4 p = posm_src[ic], v = vel_src[ic], a = acc_src[ic];
    const float vhx = kick(v.x, a.x, dt), vhy = kick(v.y, a.y, dt), vhz = kick(v.z, a.z, dt);
    const float xi = drift(p.x, vhx, dt), yi = drift(p.y, vhy, dt), zi = drift(p.z, vhz, dt);
    if (bound_slots) {  // the next step's root cube: max |coord| of the new positions
        float m = valid ? fmaxf(fabsf(xi), fmaxf(fabsf(yi), fabsf(zi))) : 0.f;
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0u) publish_bound(bound_slots, blockIdx.x * 4u + wave, m);
    }
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (PART == 2 && valid) {
        const float4 part = acc_dst[i];
        ax = part.x;
        ay = part.y;
        az = part.z;
    }
    WalkStats st;
    const bool lane0 = lane == 0u;

    StackEntry *stack = s_stack[wave];
    uint32_t sp = 0;
    const uint64_t all = __ballot(valid);
    if (all) {  // the roots, pushed so that roots.id[0] is walked first
        for (uint32_t k = n_roots; k > 0u; --k) {
            const uint32_t rid =
                (uint32_t)__builtin_amdgcn_readfirstlane((int)(roots_dev ? roots_dev->id[k - 1u] : roots_arg.id[k - 1u]));
            if (lane0) stack[sp] = StackEntry{rid, 1u, (uint32_t)all, (uint32_t)(all >> 32)};
            sp += 1;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // Termination: a group's children have larger ids than their parent (fill_kernel enforces
    // it), so no cell is reached twice; the stack check only guards against a corrupt tree.
    while (sp > 0) {
        if (sp > kWalkStack - 8u) {
            if (lane0) atomicAdd(&status[3], 1u);
            break;
        }
        --sp;
        const StackEntry top = stack[sp];  // every lane reads the same entry (LDS broadcast)
        // (the builtin returns a signed int: go through uint32_t or values sign-extend)
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)top.first);
        const uint32_t gcnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)top.count);
        const uint64_t gmask = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)top.mask_lo) |
                               ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)top.mask_hi) << 32);
        if (COUNT) st.wave_cells += gcnt;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t c0 = 0; c0 < gcnt; c0 += kWalkBatch) {
            const NodeRec *rp = rec + first + c0;
            const uint32_t rem = gcnt - c0;
            if (rem >= 4u)
                walk_cells<4, COUNT>(rp, gmask, i, xi, yi, zi, e, ax, ay, az, stack, sp, lane0, st);
            else if (rem == 3u)
                walk_cells<3, COUNT>(rp, gmask, i, xi, yi, zi, e, ax, ay, az, stack, sp, lane0, st);
            else if (rem == 2u)
                walk_cells<2, COUNT>(rp, gmask, i, xi, yi, zi, e, ax, ay, az, stack, sp, lane0, st);
            else
                walk_cells<1, COUNT>(rp, gmask, i, xi, yi, zi, e, ax, ay, az, stack, sp, lane0, st);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (COUNT && lane0) {  // per-wave statistics: cells fetched, deepest stack
        atomicAdd(&counters[2], (unsigned long long)st.wave_cells);
        atomicMax(&counters[3], (unsigned long long)st.max_sp);
        atomicAdd(&counters[4], (unsigned long long)st.wave_leaves);
        atomicMax(&counters[5], (unsigned long long)st.wave_cells);  // the longest walk of any wave
    }
    if (COUNT && valid) {
        atomicAdd(&counters[0], st.visits);
        atomicAdd(&counters[1], st.accepts);
    }
    if (!valid) return;
    if (PART == 1) {
        acc_dst[i] = float4{ax, ay, az, 0.f};
        return;
    }
    const float gdt = g * dt;
    const float fx = ax * gdt, fy = ay * gdt, fz = az * gdt;
    posm_dst[i] = float4{xi, yi, zi, p.w};
    vel_dst[i] = float4{kick(vhx, fx, dt), kick(vhy, fy, dt), kick(vhz, fz, dt), 0.f};
    acc_dst[i] = float4{fx, fy, fz, 0.f};
}

// ---- 8b. walk with the CELLS across the lanes ---------------------------------------------------
// The kernel above gives every lane a body and feeds the wave one cell at a time, so a cell that
// only a few of the 64 bodies need still costs a full wave instruction: at 2^20 bodies, theta 0.5,
// a wave evaluates 2,417 cells for bodies that need 1,011 each (42 % of the lanes do useful work,
// 17 % at depth 7 -- tools/walk_model.c).  Here the roles are transposed: a wave walks for a GROUP
// of G consecutive bodies (G = 4, 8 or 16) whose drifted positions sit in SGPRs, and its 64 lanes
// hold 64 CELLS of the traversal frontier, each with the G-bit set of bodies that have to test it.
// One batch = pop up to 64 (cell, visit mask) entries from the wave's LDS stack, every lane loads its
// own cell's 32-byte record (all bytes used), then for each of the G bodies one straight-line
// evaluation of acceptance test + force over the 64 cells, with the body's coordinates as scalar
// operands; lanes whose cell was opened by some body push its children (siblings stay adjacent
// in the stack, so the next batch's record loads coalesce).  Every lane accumulates G partial
// sums, added across the lanes once at the end of the walk in a fixed order.
//   * each body still applies ITS OWN acceptance test to exactly the cells the reference's
//     per-thread walk visits (tree.wgsl:57-70): visit and accept counts equal the oracle's;
//   * lane slots are wasted only where a cell concerns a subset of the G bodies: 65 % useful at
//     G = 8 (72 % at G = 4), and the scalar bookkeeping of the per-cell loop is gone;
//   * a walk is a chain of ~25 batches instead of ~2,400 dependent cell visits, which is what
//     bounds the small problems (benches/benchmark.rs sizes).
// A stack entry: a cell and the set of the group's bodies that have to test it -- body b at bit G - 1 - b
// ("low" format; the evaluation shifts it to the top of the word, where the carry of an add takes the
// bodies out one by one).
struct CellEnt {
    uint32_t id, mask;
};
// PACKED: the two in one word -- the mask in the low byte (G <= 8), a cell id below 2^24 above it: half
// the LDS traffic of the stack (walk -2 % at 2^20 bodies, -3 % at 4 M theta 0.75).  The host picks it
// when every id the walk can meet (the tree's capacity, the LET import area) is below 2^24.
constexpr uint32_t kPackedIdBits = 24;
template <bool PACKED>
struct CellStack;
template <>
struct CellStack<false> {
    using Ent = CellEnt;
    static __device__ __forceinline__ Ent make(uint32_t id, uint32_t mask) { return CellEnt{id, mask}; }
    // the entries of the children first, first + 1, ... of a cell: child(base(first, mask), j)
    static __device__ __forceinline__ Ent base(uint32_t first, uint32_t mask) { return CellEnt{first, mask}; }
    static __device__ __forceinline__ Ent child(const Ent &b, uint32_t j) { return CellEnt{b.id + j, b.mask}; }
    static __device__ __forceinline__ uint32_t id(const Ent &e) { return e.id; }
    static __device__ __forceinline__ uint32_t mask(const Ent &e) { return e.mask; }
    // the mask with body b at bit 31 - b
    template <int G>
    static __device__ __forceinline__ uint32_t mask_top(const Ent &e) { return e.mask << (32 - G); }
};
template <>
struct CellStack<true> {
    using Ent = uint32_t;
    static __device__ __forceinline__ Ent make(uint32_t id, uint32_t mask) { return (id << 8) | mask; }
    static __device__ __forceinline__ Ent base(uint32_t first, uint32_t mask) { return (first << 8) | mask; }
    static __device__ __forceinline__ Ent child(const Ent &b, uint32_t j) { return b + (j << 8); }
    static __device__ __forceinline__ uint32_t id(const Ent &e) { return e >> 8; }
    static __device__ __forceinline__ uint32_t mask(const Ent &e) { return e & 0xffu; }
    template <int G>
    static __device__ __forceinline__ uint32_t mask_top(const Ent &e) { return e << (32 - G); }  // (the id falls off the top)
};

// The per-body lane sets come out of the per-lane masks one bit at a time through the carry of an
// add: v <<= 1, the lanes whose top bit was set are returned as a 64-bit lane mask (one VALU
// instruction, where an and + compare would be two) ...
__device__ __forceinline__ uint64_t shl1_carry_out(uint32_t &v) {
    uint32_t o;
    uint64_t c;
    asm("v_add_co_u32_e64 %0, %1, %2, %2" : "=v"(o), "=s"(c) : "v"(v));
    v = o;
    return c;
}
// ... and go back in the same way: (v << 1) | (lane in `bit`), one add-with-carry
__device__ __forceinline__ uint32_t shl1_carry_in(uint32_t v, uint64_t bit) {
    uint32_t o;
    uint64_t unused;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(o), "=s"(unused) : "v"(v), "s"(bit));
    return o;
}
#ifndef NB_CELL_STACK
#define NB_CELL_STACK 896
#endif
#ifndef NB_WALK_WAVES
#define NB_WALK_WAVES 1
#endif
#ifndef NB_WALK_BLOCK_WAVES
#define NB_WALK_BLOCK_WAVES 1
#endif
#ifndef NB_WALK_MIN_WAVES
#define NB_WALK_MIN_WAVES 5  // waves per SIMD the register budget of the cells walk is held to
#endif
constexpr uint32_t kCellBlockWaves = NB_WALK_BLOCK_WAVES;  // waves (= groups) per workgroup
constexpr uint32_t kCellStack = NB_CELL_STACK;  // entries per wave, two-word form (7 KiB: 22 waves per CU; 1,024 entries = 8 KiB = 20 waves: +5 % at 16 M bodies); see the batch-size rule in the loop
#ifndef NB_CELL_STACK_PACKED
#define NB_CELL_STACK_PACKED 1024  // (4 KiB x 32 waves per CU; 896: +1.7 % at 2^20 bodies theta 0.5, larger: no further gain)
#endif
constexpr uint32_t kCellStackPacked = NB_CELL_STACK_PACKED;  // ... one-word form
constexpr uint32_t kCellReserve = 160;
constexpr uint32_t kWalkGatherFrom = 524288;  // bodies from which the walk gathers velocities itself (8c)

// sum over the 64 lanes, in a fixed order; the total lands in lane 63
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    uint32_t x = __float_as_uint(v);
#define NB_STEP(ctrl, row_mask) \
    x = __float_as_uint(__uint_as_float(x) + __uint_as_float(NB_DPP(0, x, ctrl, row_mask)))
    NB_STEP(0x111, 0xf);
    NB_STEP(0x112, 0xf);
    NB_STEP(0x114, 0xf);
    NB_STEP(0x118, 0xf);
    NB_STEP(0x142, 0xa);
    NB_STEP(0x143, 0xc);
#undef NB_STEP
    return __uint_as_float(x);
}

// One batch of the cells walk: the lane's cell (q = centre of gravity + mass, mac2) against the G
// bodies of the group.  vm: the bodies that have to test the cell, body b at bit 31 - b; returns
// the bodies that open it, body b at bit G - 1 - b (the stack's format); a body whose bit is set and
// that accepts the cell takes it.
typedef float v2f __attribute__((ext_vector_type(2)));

// Two bodies of the group per packed-fp32 instruction (v_pk_add/mul/fma_f32: two IEEE binary32
// operations per lane and issue slot, each rounded as the scalar instruction rounds it, so every
// bit is what the one-body-at-a-time form computes): bodies 2k and 2k+1 in the halves of bx[k].
template <int G, bool COUNT>
__device__ __forceinline__ uint32_t cells_batch(const float4 q, const float mac2, uint32_t vm,
                                                const v2f (&bx)[G / 2], const v2f (&by)[G / 2],
                                                const v2f (&bz)[G / 2], const float e,
                                                v2f (&ax)[G / 2], v2f (&ay)[G / 2], v2f (&az)[G / 2],
                                                unsigned long long &n_accepts, uint32_t &n_idle_pairs) {
    uint32_t om = 0u;  // body b ends up at bit G - 1 - b
#pragma unroll
    for (int k = 0; k < G / 2; ++k) {
        const uint64_t visit0 = shl1_carry_out(vm), visit1 = shl1_carry_out(vm);
        if (COUNT && (visit0 | visit1) == 0ull) n_idle_pairs += 1u;  // (statistics: a pair no cell of the batch concerns)
        const v2f dx = v2f{q.x, q.x} - bx[k], dy = v2f{q.y, q.y} - by[k], dz = v2f{q.z, q.z} - bz[k];
        const v2f r2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
        // acceptance size/dist < theta (tree.wgsl:63-64) as size^2 / theta^2 < r^2 (NodeRec::mac2); a leaf's
        // negative radius makes it always true.  Lane sets as 64-bit scalar masks.
        const uint64_t far0 = __ballot(mac2 < r2.x), far1 = __ballot(mac2 < r2.y);
        const uint64_t take0 = far0 & visit0, take1 = far1 & visit1;
        const uint64_t open0 = visit0 & ~far0, open1 = visit1 & ~far1;
        v2f dist;
        dist.x = __builtin_amdgcn_sqrtf(r2.x);
        dist.y = __builtin_amdgcn_sqrtf(r2.y);
        const v2f den = __builtin_elementwise_fma(v2f{e, e}, dist, r2 * r2);
        v2f rc;
        rc.x = __builtin_amdgcn_rcpf(den.x);
        rc.y = __builtin_amdgcn_rcpf(den.y);
        v2f w = v2f{q.w, q.w} * rc;
        {   // accumulate under the lanes that take the cell (exec = take), the other lanes' sums untouched:
            // six plain fma instead of two selects and three packed fma
            float a0 = ax[k].x, a1 = ay[k].x, a2 = az[k].x, b0 = ax[k].y, b1 = ay[k].y, b2 = az[k].y;
            uint64_t saved;
            asm("s_mov_b64 %[sv], exec\n\t"
                "s_mov_b64 exec, %[t0]\n\t"
                "v_fmac_f32 %[a0], %[w0], %[dx0]\n\t"
                "v_fmac_f32 %[a1], %[w0], %[dy0]\n\t"
                "v_fmac_f32 %[a2], %[w0], %[dz0]\n\t"
                "s_mov_b64 exec, %[t1]\n\t"
                "v_fmac_f32 %[b0], %[w1], %[dx1]\n\t"
                "v_fmac_f32 %[b1], %[w1], %[dy1]\n\t"
                "v_fmac_f32 %[b2], %[w1], %[dz1]\n\t"
                "s_mov_b64 exec, %[sv]"
                : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [b0] "+v"(b0), [b1] "+v"(b1), [b2] "+v"(b2),
                  [sv] "=&s"(saved)
                : [t0] "s"(take0), [t1] "s"(take1), [w0] "v"(w.x), [w1] "v"(w.y), [dx0] "v"(dx.x), [dy0] "v"(dy.x),
                  [dz0] "v"(dz.x), [dx1] "v"(dx.y), [dy1] "v"(dy.y), [dz1] "v"(dz.y));
            ax[k] = v2f{a0, b0};
            ay[k] = v2f{a1, b1};
            az[k] = v2f{a2, b2};
        }
        om = shl1_carry_in(om, open0);
        om = shl1_carry_in(om, open1);
        if (COUNT)
            n_accepts += (__builtin_amdgcn_inverse_ballot_w64(take0) ? 1ull : 0ull) +
                         (__builtin_amdgcn_inverse_ballot_w64(take1) ? 1ull : 0ull);
#ifdef NB_DIAG_EXTRA_VALU   // sensitivity probe: two more transcendentals and three fma per pair
        for (int h = 0; h < 2; ++h) {
            const float t = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf((h ? r2.y : r2.x) + 1.0f));
            float u = __builtin_fmaf(t, dx.x, dy.x);
            u = __builtin_fmaf(t, u, dz.x);
            u = __builtin_fmaf(t, u, dx.y);
            asm volatile("" ::"v"(u));
        }
#endif
    }
    return om;
}

// roots.id[0 .. split) are walked together and reduced, then roots.id[split .. count): a LET host
// may walk its own tree (PART 1) while the imports are on the wire and add them later (PART 2),
// and gets bit for bit what the one-launch step (PART 0) computes.
template <int G, bool COUNT, int PART, bool PACKED>
// (G <= 8: at most 96 VGPRs, so that five waves fit a SIMD -- the compiler lands on 90..100 by itself)
__global__ __launch_bounds__(64 * NB_WALK_BLOCK_WAVES, (G <= 8 ? NB_WALK_MIN_WAVES : NB_WALK_WAVES)) void walk_cells_kernel(
    const float4 *posm_src, const float4 *__restrict__ vel_src,
    const float4 *__restrict__ acc_src, const NodeRec *__restrict__ rec, WalkRoots roots_arg, uint32_t split,
    float4 *posm_dst, float4 *__restrict__ vel_dst, float4 *__restrict__ acc_dst,
    uint32_t lo, uint32_t hi, float g, float e, float dt,
    uint32_t *__restrict__ status, unsigned long long *__restrict__ counters,
    uint32_t *__restrict__ bound_slots, const WalkRoots *__restrict__ roots_dev,
    const uint32_t *__restrict__ va_order) {
    // va_order (section 8c): velocities and accelerations are still in the step's SOURCE order -- body k's are at
    // va_order[k] -- and the new position goes where the sorted old one was read (posm_dst == posm_src: only the
    // group itself ever reads its bodies' entries, the tree's records carry their own copies)
    // (device-made roots: fixed-stride LET imports.  Element-wise, never a copy of the struct: a
    // by-value copy of a kernel argument selected at run time lands in scratch memory)
    // (readfirstlane: the select between a kernel-argument field and device memory is a load
    // through a flat pointer, which the compiler takes for lane-dependent -- and with it the stack
    // pointer and the whole loop control, which then live in VGPRs under exec masks)
#if defined(NB_DIAG_PHASES) || defined(NB_DIAG_TIMELINE)
    unsigned long long tl_launch;   // the wave's first instruction
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_launch)::"memory");
#endif
    const uint32_t n_roots =
        (uint32_t)__builtin_amdgcn_readfirstlane((int)(roots_dev ? roots_dev->count : roots_arg.count));
    using Stack = CellStack<PACKED>;
    using Ent = typename Stack::Ent;
    static_assert(!PACKED || G <= 8, "a packed entry has 8 mask bits");
    // (the stack's LDS also carries the G x 64 floats of the final reduction)
    constexpr uint32_t kStack = PACKED ? kCellStackPacked : kCellStack;
    constexpr uint32_t kEntries = kStack * sizeof(Ent) >= (uint32_t)G * 256u ? kStack : (uint32_t)G * 256u / sizeof(Ent);
    __shared__ Ent s_stack[kCellBlockWaves][kEntries];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t per_xcd = gridDim.x / 8u;  // as walk_kernel: an XCD walks one contiguous eighth
    uint32_t blk = blockIdx.x;
    if (blk < per_xcd * 8u) blk = (blk & 7u) * per_xcd + (blk >> 3);
    const uint32_t i0 = lo + (blk * kCellBlockWaves + wave) * (uint32_t)G;  // the group: bodies i0 .. i0+G-1
    if (i0 >= hi) return;                                      // wave-uniform; the kernel has no barrier
    const uint32_t nvalid = min((uint32_t)G, hi - i0);
    const uint32_t ib = i0 + lane;
    const bool owner = lane < nvalid;  // lane b < G owns body b: loads it, integrates it at the end
    const uint32_t ic = owner ? ib : i0;
    float xi, yi, zi;
    {   // kick + drift (tree.wgsl:105-106); redone after the walk instead of kept in registers
        const uint32_t jc = va_order ? va_order[ic] : ic;
        const float4 p = posm_src[ic], v = vel_src[jc], a = acc_src[jc];
        xi = drift(p.x, kick(v.x, a.x, dt), dt);
        yi = drift(p.y, kick(v.y, a.y, dt), dt);
        zi = drift(p.z, kick(v.z, a.z, dt), dt);
    }
    v2f bx[G / 2], by[G / 2], bz[G / 2];  // the group's evaluation points, wave-uniform (SGPR pairs)
#pragma unroll
    for (int b = 0; b < G; ++b) {
        bx[b / 2][b % 2] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(xi), b));
        by[b / 2][b % 2] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(yi), b));
        bz[b / 2][b % 2] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(zi), b));
    }
    // bounding box of the group's evaluation points (for the all-open shortcut below)
    float blx = bx[0].x, bly = by[0].x, blz = bz[0].x, bhx = blx, bhy = bly, bhz = blz;
#pragma unroll
    for (int b = 1; b < G; ++b) {
        if ((uint32_t)b < nvalid) {
            blx = fminf(blx, bx[b / 2][b % 2]); bhx = fmaxf(bhx, bx[b / 2][b % 2]);
            bly = fminf(bly, by[b / 2][b % 2]); bhy = fmaxf(bhy, by[b / 2][b % 2]);
            blz = fminf(blz, bz[b / 2][b % 2]); bhz = fmaxf(bhz, bz[b / 2][b % 2]);
        }
    }
    // (wave-uniform values computed by the vector unit: move them to SGPRs)
#define NB_UNIFORM(x) x = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(x)))
    NB_UNIFORM(blx); NB_UNIFORM(bly); NB_UNIFORM(blz); NB_UNIFORM(bhx); NB_UNIFORM(bhy); NB_UNIFORM(bhz);
#undef NB_UNIFORM
    const uint32_t group_mask = ((1u << nvalid) - 1u) << ((uint32_t)G - nvalid);  // body b at bit G - 1 - b
    Ent *stack = s_stack[wave];
    float tx = 0.f, ty = 0.f, tz = 0.f;  // lane b: the finished sums of body b
    if (PART == 2 && owner) {
        const float4 part = acc_dst[ib];
        tx = part.x;
        ty = part.y;
        tz = part.z;
    }
    unsigned long long n_visits = 0, n_accepts = 0;
    uint32_t n_cells = 0, n_leaves = 0, n_batches = 0, max_sp = 0, n_idle_pairs = 0, n_evals = 0;
#ifdef NB_DIAG_PHASES
    unsigned long long ph[4] = {0, 0, 0, 0};  // cycles per phase
#endif
#if defined(NB_DIAG_PHASES) || defined(NB_DIAG_TIMELINE)
    // (three scalars: the probe must not cost the kernel a wave of occupancy)
    const unsigned long long tl_start = __builtin_amdgcn_s_memrealtime();  // the 100 MHz clock
    uint32_t tl_batches = 0;
#endif

    for (uint32_t set = 0; set < 2u; ++set) {
        const uint32_t r_lo = set == 0u ? 0u : split, r_hi = set == 0u ? min(split, n_roots) : n_roots;
        if (r_lo >= r_hi) continue;
        uint32_t sp = r_hi - r_lo;
        if (lane < sp)
            stack[lane] = Stack::make(roots_dev ? roots_dev->id[r_lo + lane] : roots_arg.id[r_lo + lane], group_mask);
        __builtin_amdgcn_wave_barrier();
        v2f ax[G / 2], ay[G / 2], az[G / 2];
#pragma unroll
        for (int k = 0; k < G / 2; ++k) ax[k] = ay[k] = az[k] = v2f{0.f, 0.f};

        bool overflowed = false;
        while (sp > 0u) {
            // Batch size: up to 64 cells, fewer when their children (at most 8 each: 7 net per
            // popped cell) would eat into the reserve.  Popping from the top keeps the walk
            // depth-first, so once batches are down to one cell the stack grows by at most 7 per
            // level below the cell it started from: 7 x 21 = 147 < kCellReserve slots, and a batch of
            // several cells is only taken while it leaves the reserve untouched -- the stack cannot
            // overflow on a consistent tree (the check below guards against a corrupt one).
            const uint32_t free_slots = kStack - sp;
            if (free_slots < 7u) {
                overflowed = true;
                break;
            }
            uint32_t c = sp < 64u ? sp : 64u;
            const uint32_t lim = free_slots >= kCellReserve + 7u ? (free_slots - kCellReserve) / 7u : 1u;
            c = c < lim ? c : lim;
            sp -= c;
#ifdef NB_DIAG_PHASES
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
            // the lanes past the batch (fewer than 64 cells) read its last entry and that cell's record like
            // lane c - 1 -- no divergent load, no second address -- and carry an empty visit set
            const uint64_t batch_lanes = ~0ull >> (64u - c);
            const bool active = __builtin_amdgcn_inverse_ballot_w64(batch_lanes);
            const Ent top = stack[sp + min(lane, c - 1u)];
            // the bodies that test this cell, body b at bit 31 - b, where the carry of an add takes them out
            const uint32_t vm = active ? Stack::template mask_top<G>(top) : 0u;
#ifdef NB_DIAG_PHASES
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::"v"(top), "v"(vm));
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
            // (a 32-bit byte offset from the uniform base: one shift and a load with a scalar base; the
            // 64-bit form costs a 64-bit shift and a 64-bit add per batch)
            const NodeRec r = *reinterpret_cast<const NodeRec *>(reinterpret_cast<const char *>(rec) +
                                                                 (Stack::id(top) * (uint32_t)sizeof(NodeRec)));
#ifdef NB_DIAG_PHASES
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::"v"(r.cogm.x), "v"(r.cogm.w), "v"(r.first), "v"(r.count), "v"(r.self_pos), "v"(r.mac2));
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#endif
#ifdef NB_DIAG_EXTRA_LOAD   // sensitivity probe: one more divergent 32-byte record load per lane and batch
            {
                const NodeRec r2 = rec[Stack::id(top) ^ 1u];
                asm volatile("" ::"v"(r2.cogm.x), "v"(r2.cogm.w), "v"(r2.first), "v"(r2.mac2));
            }
#endif
            // The top of the tree: a batch of a few big cells (the root, its children; also the
            // roots of imported trees) that EVERY body of the group opens.  One test per lane against
            // the group's bounding box decides it without touching the bodies: with the largest
            // per-axis distance to the box, r2max >= the r^2 any body computes (fp32 subtract,
            // multiply and fma are monotonic, same operation order), so "not (mac2 < r2max)"
            // implies every body's own test says open -- the same decisions, 1/8 of the work.
            bool all_open = false;
            if (c <= 8u) {
                const float dxm = fmaxf(fabsf(r.cogm.x - blx), fabsf(r.cogm.x - bhx));
                const float dym = fmaxf(fabsf(r.cogm.y - bly), fabsf(r.cogm.y - bhy));
                const float dzm = fmaxf(fabsf(r.cogm.z - blz), fabsf(r.cogm.z - bhz));
                const float r2max = __builtin_fmaf(dzm, dzm, __builtin_fmaf(dym, dym, dxm * dxm));
                all_open = (__ballot(r.mac2 < r2max) & batch_lanes) == 0ull;
            }
            // the bodies that open the lane's cell, body b at bit G - 1 - b.  (Set before the branch and
            // overwritten in it: written as if / else, the merge copies all 24 accumulators every batch.)
            uint32_t om = active ? Stack::mask(top) : 0u;
            if (!all_open) {
                // a leaf is never taken by its own body (cells carry self_pos = ~0, no body of the group):
                // that body's bit leaves the lane's set -- a leaf is never opened, so all the bit could do is
                // take the leaf -- instead of a second evaluation path with "take" masks of its own.
                // (bodies past the group's 8th clear a bit below the mask's)
                const uint32_t sb = min(r.self_pos - i0, (uint32_t)G);
                const uint32_t em = vm & ~(0x80000000u >> sb);
                om = cells_batch<G, COUNT>(r.cogm, r.mac2, em, bx, by, bz, e, ax, ay, az, n_accepts, n_idle_pairs);
                if (COUNT) n_evals += 1u;
            }
            if (COUNT) {
                n_visits += (unsigned long long)__popc(vm);
                n_cells += c;
                n_batches += 1u;
                n_leaves += (uint32_t)__popcll(__ballot(active && r.count == 0u));
            }
#ifdef NB_DIAG_PHASES
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::"v"(om), "v"(ax[0].x), "v"(ay[G / 2 - 1].y));
            const unsigned long long t3 = __builtin_amdgcn_s_memtime();
#endif
            // push the children of the opened cells: lane l writes its cnt entries at
            // sp + (children of the lanes below it), so siblings and cousins stay in lane order.
            // (An opened cell has children -- a leaf's test is always true -- so the lanes that push are
            // the lanes with a body in om; a batch that opened nothing, which is most batches of leaves,
            // skips the scan.)
            const uint64_t pushers = __ballot(om != 0u);
            if (pushers != 0ull) {
                const uint32_t cnt = om != 0u ? r.count : 0u;
                const uint32_t incl = wave_scan_u32(cnt);
                const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                Ent *dst = stack + sp + (incl - cnt);
                // Every pushing lane stores all 8 slots, highest first, without looking at its count: a
                // slot past a lane's count lands on a LOWER-numbered slot of a lane above it, which that
                // lane stores later (or beyond the new top, inside the reserve) -- one predicate for
                // the eight stores instead of eight.
                if (om != 0u) {
                    const Ent cb = Stack::base(r.first, om);
#pragma unroll
                    for (int j = 7; j >= 0; --j) {
                        dst[j] = Stack::child(cb, (uint32_t)j);
                        __builtin_amdgcn_wave_barrier();  // keep the stores in this order
                    }
                }
                sp += total;
            }
            if (COUNT) max_sp = max(max_sp, sp);
            __builtin_amdgcn_wave_barrier();
#if defined(NB_DIAG_PHASES) || defined(NB_DIAG_TIMELINE)
            tl_batches += 1u;
#endif
#ifdef NB_DIAG_PHASES
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::"s"(sp));
            const unsigned long long t4 = __builtin_amdgcn_s_memtime();
            ph[0] += t1 - t0;
            ph[1] += t2 - t1;
            ph[2] += t3 - t2;
            ph[3] += t4 - t3;
#endif
        }
        if (overflowed && lane == 0u) atomicAdd(&status[3], 1u);  // (reported outside the loop: see sp above)
        // The G sums of this root set, in a fixed order, through the (now empty) stack's LDS: every
        // lane stores its G partial sums of one component; lane l then adds the partial sums of the
        // lanes [p G, p G + G) of body b, with b = l / L, p = l % L, L = 64 / G lanes per body; the L
        // results of a body meet by butterfly; lane b fetches body b's total.
        {
            constexpr uint32_t L = 64u / (uint32_t)G;
            float *red = reinterpret_cast<float *>(stack);  // [G][64] floats <= 4 KiB of the 8 KiB stack
            const uint32_t rb = lane / L, rp = lane % L;
            float sum3[3];
#pragma unroll
            for (int comp = 0; comp < 3; ++comp) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int b = 0; b < G; ++b)
                    red[b * 64 + (int)lane] = comp == 0 ? ax[b / 2][b % 2] : comp == 1 ? ay[b / 2][b % 2] : az[b / 2][b % 2];
                __builtin_amdgcn_wave_barrier();
                float sacc = 0.f;
#pragma unroll
                for (int j = 0; j < G; ++j) sacc += red[rb * 64u + rp * (uint32_t)G + (uint32_t)j];
#pragma unroll
                for (uint32_t o = L / 2u; o > 0u; o >>= 1) sacc += __shfl_xor(sacc, (int)o);
                sum3[comp] = __shfl(sacc, (int)((lane % (uint32_t)G) * L));  // lane b < G: body b's total
            }
            __builtin_amdgcn_wave_barrier();
            tx += sum3[0];
            ty += sum3[1];
            tz += sum3[2];
        }
        if (PART == 1) break;  // the own tree only
    }
    if (COUNT) {
        atomicAdd(&counters[0], n_visits);
        atomicAdd(&counters[1], n_accepts);
        if (lane == 0u) {
            atomicAdd(&counters[2], (unsigned long long)n_cells);
            atomicMax(&counters[3], (unsigned long long)max_sp);
            atomicAdd(&counters[4], (unsigned long long)n_leaves);
            atomicMax(&counters[5], (unsigned long long)n_cells);  // the longest walk of any group
            atomicAdd(&counters[6], (unsigned long long)n_batches);
            atomicAdd(&counters[7], (unsigned long long)n_batches * (unsigned long long)(64 * G));
            atomicAdd(&counters[8], (unsigned long long)n_idle_pairs);  // (batch, pair of bodies) with no visit at all
            atomicAdd(&counters[9], (unsigned long long)n_evals);       // batches that ran the pair evaluation
        }
    }
#if defined(NB_DIAG_PHASES) || defined(NB_DIAG_TIMELINE)
    if (lane == 0u && G == 8) {  // per wave, no atomics: counters + 16 + 8 * group index
        unsigned long long *out = counters + 16 + 8 * (size_t)((i0 - lo) / (uint32_t)G);
#ifdef NB_DIAG_PHASES
        for (int k = 0; k < 4; ++k) out[k] = ph[k];
#else
        out[0] = 0ull;
#endif
        out[4] = tl_batches;   // batches, then the wave's first and last batch on the 100 MHz clock
        out[5] = tl_start;
        out[6] = __builtin_amdgcn_s_memrealtime();
        out[7] = tl_launch;
    }
#endif
    if (bound_slots && lane == 0u)  // the next step's root cube: max |coord| of the new positions (nobody waits)
        publish_bound(bound_slots, blockIdx.x, fmaxf(fmaxf(fmaxf(fabsf(blx), fabsf(bhx)), fmaxf(fabsf(bly), fabsf(bhy))),
                                                     fmaxf(fabsf(blz), fabsf(bhz))));
    if (!owner) return;
    if (PART == 1) {
        acc_dst[ib] = float4{tx, ty, tz, 0.f};
        return;
    }
    const float gdt = g * dt;
    const float fx = tx * gdt, fy = ty * gdt, fz = tz * gdt;
    // the same loads and the same operations as before the walk: bit for bit the same half kick
    const uint32_t jb = va_order ? va_order[ib] : ib;
    const float4 p = posm_src[ib], v = vel_src[jb], a = acc_src[jb];
    const float vhx = kick(v.x, a.x, dt), vhy = kick(v.y, a.y, dt), vhz = kick(v.z, a.z, dt);
    posm_dst[ib] = float4{drift(p.x, vhx, dt), drift(p.y, vhy, dt), drift(p.z, vhz, dt), p.w};
    vel_dst[ib] = float4{kick(vhx, fx, dt), kick(vhy, fy, dt), kick(vhz, fz, dt), 0.f};
    acc_dst[ib] = float4{fx, fy, fz, 0.f};
}

// ---- 9. locally essential trees (multi-GPU Barnes-Hut) ------------------------
// Every rank owns a Morton range of the bodies and builds the octree of ITS bodies inside the
// GLOBAL root cube.  What a peer needs of that tree to walk it for its own bodies is the
// "locally essential tree" (LET): starting at the root, a cell that EVERY point of the peer's
// bounding box accepts (size^2 < theta^2 * dmin^2, dmin = distance from the cell's centre of
// gravity to the box) is exported as a terminal pseudo-body, any other cell is exported with
// its children.  dmin^2 is evaluated with the walk's own operation order on the per-axis
// clamped distances, and fp32 subtract / multiply / fma are monotonic, so dmin^2 <= the r^2 any
// body inside the box computes: the pruning never changes a decision a body of the peer would
// take -- walking the LET gives bit for bit what walking the whole remote tree would give.
//
// Per-rank meta words exchanged before the build (all-gather): [0] bits of max |coord| of the
// source positions (the global root cube is the max over ranks), [1..3] / [4..6] min / max of the
// DRIFTED positions (the points the walk evaluates at) in an order-preserving u32 encoding.
constexpr int kLetMetaWords = 8;

__device__ __forceinline__ uint32_t let_f2ord(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float let_ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ __launch_bounds__(256) void let_meta_kernel(const float4 *__restrict__ posm,
                                                       const float4 *__restrict__ vel,
                                                       const float4 *__restrict__ acc, uint32_t n, float dt,
                                                       uint32_t *__restrict__ meta) {
    __shared__ float s_lo[4][3], s_hi[4][3];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = posm[i], v = vel[i], a = acc[i];
        // exactly the walk's evaluation point (kick + drift, tree.wgsl:105-106)
        const float x = drift(p.x, kick(v.x, a.x, dt), dt), y = drift(p.y, kick(v.y, a.y, dt), dt),
                    z = drift(p.z, kick(v.z, a.z, dt), dt);
        lo[0] = fminf(lo[0], x); hi[0] = fmaxf(hi[0], x);
        lo[1] = fminf(lo[1], y); hi[1] = fmaxf(hi[1], y);
        lo[2] = fminf(lo[2], z); hi[2] = fmaxf(hi[2], z);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[c] = fminf(lo[c], __shfl_xor(lo[c], o));
            hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o));
        }
        if ((threadIdx.x & 63) == 0) {
            s_lo[threadIdx.x >> 6][c] = lo[c];
            s_hi[threadIdx.x >> 6][c] = hi[c];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        const float l = fminf(fminf(s_lo[0][c], s_lo[1][c]), fminf(s_lo[2][c], s_lo[3][c]));
        const float h = fmaxf(fmaxf(s_hi[0][c], s_hi[1][c]), fmaxf(s_hi[2][c], s_hi[3][c]));
        if (l <= h) {  // (a block that saw no body contributes nothing)
            atomicMin(&meta[1 + c], let_f2ord(l));
            atomicMax(&meta[4 + c], let_f2ord(h));
        }
    }
}

// the global root cube: max over ranks of the local bounds (bit patterns of floats >= 1.0)
__global__ void let_global_bound_kernel(const uint32_t *__restrict__ meta_all, int world,
                                        uint32_t *__restrict__ bound_bits, uint32_t *__restrict__ my_counts,
                                        int rank, uint32_t first_free) {
    uint32_t m = __float_as_uint(1.0f);
    for (int r = 0; r < world; ++r) m = max(m, meta_all[r * kLetMetaWords]);
    *bound_bits = m;
    // every peer's export starts with the root in slot 0 (one-launch export: slots 1..72 reserved too)
    if (my_counts)
        for (int r = 0; r < world; ++r) my_counts[r] = r == rank ? 0u : first_free;
}

// One depth of the export, all peers at once (blockIdx.y = peer).  Node ids are breadth-first
// (depth-major), so the nodes of one depth are a contiguous id range and their parents were
// handled by the previous launch: a reached node finds its output slot in out_slot.
__global__ __launch_bounds__(256) void let_export_level_kernel(
    const NodeRec *__restrict__ rec, const uint32_t *__restrict__ depth_base, int depth,
    const uint32_t *__restrict__ n_nodes_p, uint32_t n_cap, const uint32_t *__restrict__ meta_all,
    int rank, bool prune, uint32_t *__restrict__ out_slot, NodeRec *__restrict__ send,
    uint32_t *__restrict__ counts, uint32_t cap, uint32_t *__restrict__ status) {
    const int q = blockIdx.y;
    if (q == rank) return;
    const uint32_t n_nodes = min(*n_nodes_p, n_cap);
    const uint32_t begin = depth_base[depth], end = min(depth_base[depth + 1], n_nodes);
    if (begin >= end) return;
    const uint32_t *mq = meta_all + q * kLetMetaWords;
    const float blo[3] = {let_ord2f(mq[1]), let_ord2f(mq[2]), let_ord2f(mq[3])};
    const float bhi[3] = {let_ord2f(mq[4]), let_ord2f(mq[5]), let_ord2f(mq[6])};
    if (!(blo[0] <= bhi[0])) return;  // the peer has no bodies: nothing to export
    uint32_t *slots = out_slot + (size_t)q * n_cap;
    NodeRec *out = send + (size_t)q * cap;
    // A block takes 256 consecutive nodes at a time and allocates the output slots of all their
    // children with ONE atomic (block-wide scan of the child counts): children of neighbouring
    // cells stay neighbours in the export, which is what the importer's caches want, and the
    // counter sees 1/256 of the traffic.
    __shared__ uint32_t s_wave[4], s_base;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t chunk = begin + blockIdx.x * blockDim.x; chunk < end; chunk += gridDim.x * blockDim.x) {
        const uint32_t id = chunk + threadIdx.x;
        uint32_t slot = ~0u, want = 0u;
        NodeRec r{};
        if (id < end) {
            slot = depth == 0 ? 0u : slots[id];
            if (slot < cap) {  // (~0: not reached for this peer)
                r = rec[id];
                if (r.count != 0u) {
                    // nearest point of the box to the centre of gravity, per axis, then r^2 in the walk's order
                    const float dx = r.cogm.x - fminf(fmaxf(r.cogm.x, blo[0]), bhi[0]);
                    const float dy = r.cogm.y - fminf(fmaxf(r.cogm.y, blo[1]), bhi[1]);
                    const float dz = r.cogm.z - fminf(fmaxf(r.cogm.z, blo[2]), bhi[2]);
                    const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                    // some point of the box may open it: export the children too
                    if ((!prune || !(r.mac2 < r2)) && r.first + r.count <= n_nodes) want = r.count;
                }
            }
        }
        uint32_t incl = want;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o);
            if ((int)lane >= o) incl += y;
        }
        if (lane == 63u) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = 0u;
        for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
        if (threadIdx.x == 0) {
            const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
            s_base = total ? atomicAdd(&counts[q], total) : 0u;
        }
        __syncthreads();
        if (slot < cap) {
            NodeRec o{r.cogm, 0u, 0u, ~0u, -1.0f};  // terminal: a body / pseudo-body for the peer
            if (want) {
                const uint32_t base = s_base + before + incl - want;
                if (base + want <= cap) {
                    for (uint32_t c = 0; c < want; ++c) slots[r.first + c] = base + c;
                    o = NodeRec{r.cogm, base, want, ~0u, r.mac2};
                } else {
                    atomicAdd(&status[0], 1u);  // capacity exceeded: reported by check_status
                }
            }
            out[slot] = o;
        }
        __syncthreads();  // s_wave / s_base are reused by the next chunk
    }
}

// a peer whose export ran out of room (status[0], an error at the next read-back) still gets a
// count that fits its segment
// The whole export in ONE launch (the level-by-level form above is 23 dependent launches whatever the
// tree's depth: ~115 us of a LET step that takes ~350 at 131,072 bodies per rank).  A workgroup of
// 1,024 threads exports, for one peer q (blockIdx.y), the subtree under one of the 64 grandchildren
// (blockIdx.x) of the root, breadth-first: the level's records sit in the peer's segment already (allocated by their
// parents), each holding -- provisionally, in `first` -- the node it stands for; the workgroup takes
// them 1,024 at a time, decides terminal / exported with children exactly as above, allocates the
// children of a chunk with one atomic on the peer's counter and remembers the (base, length) of every
// allocation in LDS: those ranges are the next level.  Slots 1..72 of a segment are reserved for the
// root's children and grandchildren (unused ones hold terminals nobody references), which is what
// lets the 64 subtrees proceed without meeting (with 8 subtrees a workgroup had up to 1/8 of a big
// export to itself: 400 us instead of 310 for the build + export of 524,288 bodies).  The layout of a segment depends on the order of the atomics; the
// walk does not (siblings stay consecutive and in octant order, and a lane's partial sums are
// added across the wave in a fixed order): bit for bit the level-by-level export's result.
struct LetRange {
    uint32_t base, len;
};
constexpr uint32_t kLetExportThreads = 1024, kLetExportRanges = 3072;  // 2 lists x 24 KiB of LDS
constexpr uint32_t kLetReserved = 73;  // the root, its 8 children, their 64 children: fixed slots
constexpr uint32_t kLetListOverflow = 0x80000000u;  // status[0]: a level outgrew the one-launch export's range list

__device__ __forceinline__ uint32_t let_export_want(const NodeRec &r, const float (&blo)[3], const float (&bhi)[3],
                                                    bool prune, uint32_t n_nodes) {
    if (r.count == 0u) return 0u;
    // nearest point of the box to the centre of gravity, per axis, then r^2 in the walk's order
    const float dx = r.cogm.x - fminf(fmaxf(r.cogm.x, blo[0]), bhi[0]);
    const float dy = r.cogm.y - fminf(fmaxf(r.cogm.y, blo[1]), bhi[1]);
    const float dz = r.cogm.z - fminf(fmaxf(r.cogm.z, blo[2]), bhi[2]);
    const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    // some point of the box may open it: export the children too
    return ((!prune || !(r.mac2 < r2)) && r.first + r.count <= n_nodes) ? r.count : 0u;
}

__global__ __launch_bounds__(kLetExportThreads) void let_export_kernel(
    const NodeRec *__restrict__ rec, const uint32_t *__restrict__ n_nodes_p, uint32_t n_cap,
    const uint32_t *__restrict__ meta_all, int rank, bool prune, NodeRec *send,
    uint32_t *__restrict__ counts, uint32_t cap, uint32_t *__restrict__ status) {
    const int q = blockIdx.y;
    const uint32_t sub = blockIdx.x, tid = threadIdx.x;
    if (q == rank) return;
    const uint32_t n_nodes = min(*n_nodes_p, n_cap);
    const uint32_t *mq = meta_all + q * kLetMetaWords;
    const float blo[3] = {let_ord2f(mq[1]), let_ord2f(mq[2]), let_ord2f(mq[3])};
    const float bhi[3] = {let_ord2f(mq[4]), let_ord2f(mq[5]), let_ord2f(mq[6])};
    if (!(blo[0] <= bhi[0]) || n_nodes == 0u) {  // the peer has no bodies / this rank has none: nothing to export
        if (sub == 0u && tid == 0u) counts[q] = 0u;
        return;
    }
    NodeRec *out = send + (size_t)q * cap;
    const NodeRec root = rec[0];
    const uint32_t want0 = cap >= kLetReserved ? let_export_want(root, blo, bhi, prune, n_nodes) : 0u;
    const NodeRec dummy{float4{0.f, 0.f, 0.f, 0.f}, 0u, 0u, ~0u, -1.0f};
    if (sub == 0u && tid < kLetReserved) {
        // slot 0: the root; 1 + c: child c of the root; 9 + 8 c + j: child j of that child (those that exist
        // and are exported are written by their own workgroups, the rest hold terminals nobody references)
        if (tid == 0u) {
            out[0] = want0 ? NodeRec{root.cogm, 1u, want0, ~0u, root.mac2} : NodeRec{root.cogm, 0u, 0u, ~0u, -1.0f};
            if (!want0) counts[q] = 1u;  // (the counter starts at kLetReserved; nobody else touches it then)
        } else if (want0) {
            const uint32_t c = tid <= 8u ? tid - 1u : (tid - 9u) >> 3, j = (tid - 9u) & 7u;
            NodeRec rc = dummy;
            uint32_t want1 = 0u;
            if (c < want0) {
                rc = rec[root.first + c];
                want1 = let_export_want(rc, blo, bhi, prune, n_nodes);
            }
            if (tid <= 8u) {
                if (c < want0)
                    out[tid] = want1 ? NodeRec{rc.cogm, 9u + 8u * c, want1, ~0u, rc.mac2}
                                     : NodeRec{rc.cogm, 0u, 0u, ~0u, -1.0f};
                else
                    out[tid] = dummy;
            } else if (j >= want1) {
                out[tid] = dummy;
            }
        }
    }
    const uint32_t c = sub >> 3, j = sub & 7u;
    if (c >= want0) return;
    const NodeRec rc = rec[root.first + c];
    if (j >= let_export_want(rc, blo, bhi, prune, n_nodes)) return;
    const uint32_t seed_slot = 9u + 8u * c + j, seed_node = rc.first + j;

    __shared__ LetRange s_list[2][kLetExportRanges];
    __shared__ uint32_t s_n[2], s_wave[kLetExportThreads / 64], s_base;
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    if (tid == 0u) {
        out[seed_slot].first = seed_node;  // provisional: the node this record stands for
        s_list[0][0] = LetRange{seed_slot, 1u};
        s_n[0] = 1u;
        s_n[1] = 0u;
    }
    __threadfence_block();
    __syncthreads();
    for (uint32_t cur = 0;; cur ^= 1u) {
        const uint32_t nr = s_n[cur];
        if (nr == 0u) break;
        for (uint32_t ri = 0; ri < nr; ++ri) {
            const LetRange rg = s_list[cur][ri];
            for (uint32_t off = 0; off < rg.len; off += kLetExportThreads) {
                const uint32_t i = off + tid, slot = rg.base + i;
                const bool valid = i < rg.len;
                NodeRec r{};
                uint32_t want = 0u;
                if (valid) {
                    r = rec[out[slot].first];
                    want = let_export_want(r, blo, bhi, prune, n_nodes);
                }
                uint32_t incl = want;
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t y = __shfl_up(incl, o);
                    if ((int)lane >= o) incl += y;
                }
                if (lane == 63u) s_wave[wave] = incl;
                __syncthreads();
                uint32_t before = 0u;
                for (uint32_t w = 0; w < wave; ++w) before += s_wave[w];
                if (tid == 0u) {
                    uint32_t total = 0u;
                    for (uint32_t w = 0; w < kLetExportThreads / 64u; ++w) total += s_wave[w];
                    uint32_t base = 0u;
                    if (total) {
                        base = atomicAdd(&counts[q], total);
                        const uint32_t k = s_n[cur ^ 1u];
                        if (base + total <= cap && k < kLetExportRanges) {
                            s_list[cur ^ 1u][k] = LetRange{base, total};
                            s_n[cur ^ 1u] = k + 1u;
                        } else {
                            // the peer's segment is full (counted), or this level has more ranges than the LDS list
                            // holds (flagged apart: the segment had room, the level-by-level export would succeed);
                            // either way none of this chunk's cells is exported with children.  check_status reports it.
                            if (base + total > cap) atomicAdd(&status[0], 1u);
                            else atomicOr(&status[0], kLetListOverflow);
                            base = ~0u;
                        }
                    }
                    s_base = base;
                }
                __syncthreads();
                if (valid) {
                    NodeRec o{r.cogm, 0u, 0u, ~0u, -1.0f};  // terminal: a body / pseudo-body for the peer
                    if (want && s_base != ~0u) {
                        const uint32_t base = s_base + before + incl - want;
                        for (uint32_t c = 0; c < want; ++c) out[base + c].first = r.first + c;  // provisional
                        o = NodeRec{r.cogm, base, want, ~0u, r.mac2};
                    }
                    out[slot] = o;
                }
                __threadfence_block();
                __syncthreads();  // s_wave / s_base are reused; the provisional records are visible
            }
        }
        if (tid == 0u) s_n[cur] = 0u;
        __syncthreads();
    }
}

__global__ void let_clamp_counts_kernel(uint32_t *__restrict__ counts, int world, uint32_t cap) {
    const int q = threadIdx.x;
    if (q < world) counts[q] = min(counts[q], cap);
}

// ---- migration: a body belongs to the rank whose Morton-key range (in a fixed reference cube)
// holds its position.  Bodies that left are packed per destination, the rest are compacted;
// the order inside the arrays is irrelevant (every step re-sorts).
struct LetOwners {
    uint32_t world;
    float ref_bound;                        // the reference cube is [-ref_bound, ref_bound]^3
    unsigned long long split[kLetMaxWorld]; // rank r owns keys in [split[r-1], split[r]); split[world-1] = inf
};

__device__ __forceinline__ unsigned long long let_spread21(unsigned long long v) {
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__device__ __forceinline__ unsigned long long let_ref_key(float4 p, float ref_bound) {
    const double b = (double)ref_bound;
    unsigned long long q[3];
    const float c[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double t = ((double)c[k] + b) / (2.0 * b) * 2097152.0;
        t = t < 0.0 ? 0.0 : (t > 2097151.0 ? 2097151.0 : t);   // NaN falls through to the cast: 0
        q[k] = (unsigned long long)t;
    }
    return let_spread21(q[0]) | (let_spread21(q[1]) << 1) | (let_spread21(q[2]) << 2);
}

// stayers -> dst arrays (compacted), leavers -> send segment of their owner (12 floats per body)
__global__ __launch_bounds__(256) void let_migrate_kernel(
    const float4 *__restrict__ posm, const float4 *__restrict__ vel, const float4 *__restrict__ acc,
    uint32_t n, LetOwners own, int rank, float4 *__restrict__ posm_dst, float4 *__restrict__ vel_dst,
    float4 *__restrict__ acc_dst, float4 *__restrict__ send, uint32_t seg_cap,
    uint32_t *__restrict__ counts, uint32_t *__restrict__ status) {
    __shared__ uint32_t s_cnt[kLetMaxWorld], s_base[kLetMaxWorld];
    if (threadIdx.x < kLetMaxWorld) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t dest = 0, local = 0;
    float4 p{}, v{}, a{};
    if (i < n) {
        p = posm[i];
        v = vel[i];
        a = acc[i];
        const unsigned long long key = let_ref_key(p, own.ref_bound);
        while (dest + 1 < own.world && key >= own.split[dest]) ++dest;
        local = atomicAdd(&s_cnt[dest], 1u);
    }
    __syncthreads();
    if (threadIdx.x < own.world)
        s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]) : 0u;
    __syncthreads();
    if (i >= n) return;
    const uint32_t slot = s_base[dest] + local;
    if ((int)dest == rank) {
        posm_dst[slot] = p;   // slot < n: stayers never outnumber the bodies
        vel_dst[slot] = v;
        acc_dst[slot] = a;
    } else if (slot < seg_cap) {
        float4 *o = send + ((size_t)dest * seg_cap + slot) * 3;
        o[0] = p;
        o[1] = v;
        o[2] = a;
    } else {
        atomicAdd(&status[0], 1u);  // more leavers than the segment holds: reported by check_status
    }
}

__global__ void let_append_kernel(const float4 *__restrict__ recv, uint32_t count, uint32_t at,
                                  float4 *__restrict__ posm, float4 *__restrict__ vel,
                                  float4 *__restrict__ acc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    posm[at + i] = recv[3 * (size_t)i + 0];
    vel[at + i] = recv[3 * (size_t)i + 1];
    acc[at + i] = recv[3 * (size_t)i + 2];
}

struct LetSegments {
    uint32_t world;
    uint32_t off[kLetMaxWorld + 1];  // record offsets of the imported segments (exclusive scan)
};

// imported child links are relative to their segment: make them indices into the walk's table,
// and turn any link that does not point forward inside its own segment into a terminal
__global__ void let_rebase_kernel(NodeRec *__restrict__ imp, LetSegments segs, uint32_t import_base) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= segs.off[segs.world]) return;
    uint32_t s = 0;
    while (s + 1 < segs.world && i >= segs.off[s + 1]) ++s;
    NodeRec r = imp[i];
    const uint32_t local = i - segs.off[s], seg_n = segs.off[s + 1] - segs.off[s];
    r.self_pos = ~0u;
    if (r.count != 0u) {
        if (r.count <= 8u && r.first > local && r.first + r.count <= seg_n) {
            r.first += import_base + segs.off[s];
        } else {
            r.first = 0u;
            r.count = 0u;
            r.mac2 = -1.0f;
        }
    }
    imp[i] = r;
}

// The same for imports that arrive in FIXED-STRIDE segments (nb_sim_let_set_import_stride): segment j
// (the j-th peer in rank order, this rank skipped) starts at record j * stride, and how many of its
// records are real is read HERE, on the device, from the all-gathered counts matrix -- the host
// never sees the counts, so a step needs no host synchronisation.  Also writes the walk's roots.
__global__ void let_rebase_fixed_kernel(NodeRec *__restrict__ imp, const uint32_t *__restrict__ counts_all,
                                        uint32_t me, uint32_t world, uint32_t stride, uint32_t import_base,
                                        uint32_t own_root, WalkRoots *__restrict__ roots_dev,
                                        uint32_t *__restrict__ status) {
    // blockIdx.y = segment (the peers in rank order, this rank left out); the blocks of a segment stride over
    // its LIVE records only -- the launch does not grow with the stride (the one-process runner's is the
    // whole tree_let_cap)
    if (blockIdx.x == 0u && blockIdx.y == 0u && threadIdx.x == 0u) {
        // the trees this rank walks: its own (optional), then the non-empty imports in rank order
        WalkRoots rt{};
        if (own_root) rt.id[rt.count++] = 0u;
        for (uint32_t r = 0; r < world; ++r) {
            if (r == me) continue;
            const uint32_t c = counts_all[r * world + me], j = r < me ? r : r - 1u;
            if (c > stride) atomicAdd(&status[0], 1u);  // the sender had more than the segment holds
            if (c) rt.id[rt.count++] = import_base + j * stride;
        }
        *roots_dev = rt;
    }
    if (world < 2u) return;
    const uint32_t j = blockIdx.y, r = j < me ? j : j + 1u;
    const uint32_t seg_n = min(counts_all[r * world + me], stride);
    NodeRec *seg = imp + (size_t)j * stride;
    for (uint32_t local = blockIdx.x * blockDim.x + threadIdx.x; local < seg_n; local += gridDim.x * blockDim.x) {
        NodeRec rc = seg[local];
        rc.self_pos = ~0u;
        if (rc.count != 0u) {
            if (rc.count <= 8u && rc.first > local && rc.first + rc.count <= seg_n) {
                rc.first += import_base + j * stride;
            } else {
                rc.first = 0u;
                rc.count = 0u;
                rc.mac2 = -1.0f;
            }
        }
        seg[local] = rc;
    }
}

// One-process LET runner (nb_group.cpp): the records exported for peer q go straight into q's import
// area through peer access -- as many as the export counted (this rank's row of the counts table, read
// here on the device), to the segment the fixed-stride layout gives this rank on q.  blockIdx.y = q.
struct LetImportPtrs {
    NodeRec *p[kLetMaxWorld];
};
__global__ __launch_bounds__(256) void let_push_segments_kernel(const NodeRec *__restrict__ send, uint32_t seg_records,
                                                                const uint32_t *__restrict__ my_counts,
                                                                LetImportPtrs imports, uint32_t me, uint32_t stride) {
    const uint32_t q = blockIdx.y;
    if (q == me) return;
    const uint32_t count = min(my_counts[q], stride);  // (more than the segment holds: the receiver reports it)
    const uint32_t j = me < q ? me : me - 1u;
    const uint4 *s4 = reinterpret_cast<const uint4 *>(send + (size_t)q * seg_records);
    uint4 *d4 = reinterpret_cast<uint4 *>(imports.p[q] + (size_t)j * stride);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count * 2u; i += gridDim.x * blockDim.x) d4[i] = s4[i];
}

// ---- AoS conversion of the device tree (nb_sim_read_tree) ---------------------------------------
__global__ void tree_to_aos_kernel(const float4 *__restrict__ cogm, const uint32_t *__restrict__ bodies,
                                   const uint32_t *__restrict__ child, uint32_t n_nodes,
                                   nb_octant *__restrict__ out) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_nodes) return;
    nb_octant o;
    const float4 q = cogm[id];
    o.cog[0] = q.x; o.cog[1] = q.y; o.cog[2] = q.z;
    o.mass = q.w;
    o.bodies = bodies[id];
    for (int c = 0; c < 8; ++c) o.children[c] = child[(size_t)id * 8 + c];
    out[id] = o;
}

__global__ void tree_aos_to_soa_kernel(const nb_particle *__restrict__ aos, uint32_t n,
                                       float4 *__restrict__ posm, float4 *__restrict__ vel,
                                       float4 *__restrict__ acc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const nb_particle p = aos[i];
    posm[i] = float4{p.position[0], p.position[1], p.position[2], p.mass};
    vel[i] = float4{p.velocity[0], p.velocity[1], p.velocity[2], 0.f};
    acc[i] = float4{p.acceleration[0], p.acceleration[1], p.acceleration[2], 0.f};
}

// =================================================================================================
// TreeSim host side
// =================================================================================================
class TreeSim final : public SimBase {
   public:
    ~TreeSim() override {
        (void)hipSetDevice(place.device_id);
        drop_graph();
        for (void *p : allocs) (void)hipFree(p);
        for (hipEvent_t ev : events) (void)hipEventDestroy(ev);
        if (h_status) (void)hipHostFree(h_status);
    }

    int init(const nb_particle *host, size_t count) override {
        if (count != n) {
            set_error("particle count %zu does not match sim_params.particle_num %u", count, n);
            return NB_ERR_INVALID;
        }
        theta = add.theta > 0.f ? add.theta : NB_DEFAULT_THETA;
        n_capacity = n;
        const size_t nn = n ? n : 1;
        node_cap = (uint32_t)std::min<size_t>(4 * nn + 8, kSlotIdMask);  // 4N as tree.rs:188-190 (node ids take 27 bits of a slot record: 33 million bodies before the cap is less than 4N)
        sort_items = nn <= kSortSmallMax ? kSortItemsSmall : kSortItems;
        sort_blocks = (uint32_t)((nn + kSortThreads * sort_items - 1) / (kSortThreads * sort_items));
        cell_tiles = (uint32_t)std::max({std::min<size_t>(nn, 131072) / 256, std::min<size_t>(nn, 524288) / 512, nn / kCellTile}) + 4;  // capacity
        const size_t npad = n_pad ? n_pad : 256;  // equal-sized slices for the all-gathers
        for (int b = 0; b < 2; ++b) {
            if (int rc = alloc(&posm[b], sizeof(float4) * npad)) return rc;
            if (int rc = alloc(&vel[b], sizeof(float4) * npad)) return rc;
            if (int rc = alloc(&acc[b], sizeof(float4) * npad)) return rc;
            NB_HIP_TRY(hipMemsetAsync(posm[b], 0, sizeof(float4) * npad, stream));
            NB_HIP_TRY(hipMemsetAsync(vel[b], 0, sizeof(float4) * npad, stream));
            NB_HIP_TRY(hipMemsetAsync(acc[b], 0, sizeof(float4) * npad, stream));
            if (int rc = alloc(&keys[b], sizeof(uint64_t) * nn)) return rc;
            if (int rc = alloc(&idx[b], sizeof(uint32_t) * nn)) return rc;
        }
        if (int rc = alloc(&d_aos, sizeof(nb_particle) * nn)) return rc;
        if (int rc = alloc(&hist, sizeof(uint32_t) * kSortMaxBins * (size_t)sort_blocks)) return rc;
        if (int rc = alloc(&totals, sizeof(uint32_t) * kSortMaxBins)) return rc;
        if (int rc = alloc(&cpl, nn + 2)) return rc;
        if (int rc = alloc(&int_slot, sizeof(uint32_t) * nn)) return rc;
        if (int rc = alloc(&tile_u32, sizeof(uint32_t) * kCellRows * ((size_t)cell_tiles + 4))) return rc;
        if (int rc = alloc(&tile_mom, sizeof(Moments) * (size_t)cell_tiles)) return rc;
        if (int rc = alloc(&leaf_id, sizeof(uint32_t) * nn)) return rc;
        if (int rc = alloc(&int_id, sizeof(uint2) * (size_t)node_cap)) return rc;
        if (int rc = alloc(&node_first, sizeof(uint32_t) * (size_t)node_cap)) return rc;
        if (int rc = alloc(&node_depth, (size_t)node_cap)) return rc;
        if (int rc = alloc(&rec, sizeof(NodeRec) * (size_t)node_cap)) return rc;
        if (int rc = alloc(&mom_prefix, sizeof(Moments) * (nn + 1))) return rc;
        // (the reference's Octant fields -- cogm, bodies, child, the AoS staging: 104 B per node -- are
        // only produced for nb_sim_read_tree and allocated on its first call)
        if (int rc = alloc(&scalars, sizeof(uint32_t) * 128)) return rc;
        #if defined(NB_DIAG_PHASES) || defined(NB_DIAG_TIMELINE)
        if (int rc = alloc(&counters, sizeof(unsigned long long) * (16 + nn + 8))) return rc;
#else
        if (int rc = alloc(&counters, sizeof(unsigned long long) * 16)) return rc;
#endif
        if (int rc = alloc(&bound_buf, sizeof(uint32_t) * kBoundSlots)) return rc;
        NB_HIP_TRY(hipMemsetAsync(scalars, 0, sizeof(uint32_t) * 128, stream));
        NB_HIP_TRY(hipMemsetAsync(bound_buf, 0, sizeof(uint32_t) * kBoundSlots, stream));
        NB_HIP_TRY(hipMemsetAsync(counters, 0, sizeof(unsigned long long) * 16, stream));
        NB_HIP_TRY(hipHostMalloc((void **)&h_status, sizeof(uint32_t) * 12, hipHostMallocDefault));
        return write_particles(host, count);
    }

    int write_particles(const nb_particle *host, size_t count) override {
        if (count != n) {
            set_error("write_particles: count %zu != particle_num %u", count, n);
            return NB_ERR_INVALID;
        }
        if (int rc = bind_device()) return rc;
        if (n == 0) return NB_OK;
        build_done = false;  // a build enqueued for the old state is void
        bound_from_walk = false;
        // a graph captured after an eager step takes the root cube from the slots the previous walk
        // filled and holds no bound_kernel: replayed on the new state it would key the bodies in the
        // OLD state's cube.  Re-capture (the next capture starts from bound_kernel).
        drop_graph();
        NB_HIP_TRY(hipMemcpyAsync(d_aos, host, sizeof(nb_particle) * (size_t)n, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(tree_aos_to_soa_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_aos, n,
                           posm[cur], vel[cur], acc[cur]);
        NB_HIP_TRY(hipGetLastError());
        NB_HIP_TRY(hipStreamSynchronize(stream));
        return NB_OK;
    }

    // TreeSim::encode, tree.rs:262-353 -- everything on the device, nothing mapped to the host.
    // One step = ~45 small launches that never change (same buffers, same arguments every step:
    // the state lands back in buffer `cur`), so the sequence can be captured into a hipGraph once
    // and replayed (tuning key "tree_use_graph").  Off by default: measured, the step is bound by
    // the GPU-side cost of the dependent launches, not by their submission (8,192 bodies: 459 us
    // eager vs 439 us replayed; no difference at 1 M), so the eager path is the one that ships.
    int encode() override {
        if (int rc = bind_device()) return rc;
        if (n == 0) {
            step_num += 1;
            return NB_OK;
        }
        if (!use_graph || time_walk || build_done) {
            if (int rc = enqueue_step()) return rc;
            step_num += 1;
            return NB_OK;
        }
        if (!graph_exec) {
            hipGraph_t graph = nullptr;
            NB_HIP_TRY(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            const int rc = enqueue_step();
            const hipError_t ec = hipStreamEndCapture(stream, &graph);
            if (rc) {
                if (graph) (void)hipGraphDestroy(graph);
                return rc;
            }
            NB_HIP_TRY(ec);
            const hipError_t ei = hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            NB_HIP_TRY(ei);
        }
        NB_HIP_TRY(hipGraphLaunch(graph_exec, stream));
        step_num += 1;
        return NB_OK;
    }

    void drop_graph() {
        if (graph_exec) {
            (void)hipGraphExecDestroy(graph_exec);
            graph_exec = nullptr;
        }
    }

    // The step in two halves (nb_sim_encode_phase): the build needs only positions and masses,
    // the walk also velocities and accelerations -- a multi-GPU host starts the build as soon as
    // the position all-gather of the previous step has landed and lets the other two overlap it.
    int encode_phase(int phase) override {
        if (int rc = bind_device()) return rc;
        if (phase == NB_PHASE_LET_MIGRATE) return let_migrate();
        if (phase == NB_PHASE_LET_WALK_OWN) return let_walk_own();
        if (phase >= NB_PHASE_LET_META && phase <= NB_PHASE_LET_WALK) return let_phase(phase);
        if (phase != 0 && phase != 1) {
            set_error("encode_phase: phase must be 0 or 1 (or a NB_PHASE_LET_* value)");
            return NB_ERR_INVALID;
        }
        if (let_world) {
            set_error("encode_phase(%d): this TreeSim runs the LET protocol (phases %d..%d)", phase,
                      NB_PHASE_LET_META, NB_PHASE_LET_WALK);
            return NB_ERR_INVALID;
        }
        if (n == 0) {
            if (phase == 1) step_num += 1;
            return NB_OK;
        }
        if (phase == 0) {
            if (build_done) {
                set_error("encode_phase(0) called twice for one step");
                return NB_ERR_INVALID;
            }
            if (int rc = enqueue_build(false)) return rc;
            build_done = true;
            return NB_OK;
        }
        if (!build_done)
            if (int rc = enqueue_build(false)) return rc;
        build_done = false;
        if (int rc = enqueue_walk(own_root())) return rc;
        step_num += 1;
        return NB_OK;
    }

    WalkRoots own_root() const {
        WalkRoots r{};
        r.count = n >= 2 ? 1u : 0u;  // (the reference's N = 1 tree is ill-formed; a lone body feels nothing)
        return r;
    }

    // ---- locally essential trees: the three phases of a multi-GPU step (see section 9) -----------
    // NB_PHASE_LET_META   local bound + drifted bounding box -> this rank's meta words
    //                     (caller: all-gather exchange region 0)
    // NB_PHASE_LET_BUILD  global root cube, local octree, LET export for every peer
    //                     (caller: all-gather region 1 = export counts, read them on the host,
    //                      all-to-all the segments of region 2 into region 3, nb_sim_let_set_imports)
    // NB_PHASE_LET_WALK   rebase the imported trees, walk own tree + imports, integrate
    int let_phase(int phase) {
        if (!let_world || !let_send) {
            set_error("LET phase %d: set tree_let_world, tree_let_rank and tree_let_cap first", phase);
            return NB_ERR_INVALID;
        }
        if (phase != let_next || let_arrivals_pending) {
            set_error("LET phases must run in order: expected %d, got %d%s", let_next, phase,
                      let_arrivals_pending ? " (a migration awaits nb_sim_let_set_arrivals)" : "");
            return NB_ERR_INVALID;
        }
        const int s = cur;
        const dim3 b256(256);
        uint32_t *my_meta = let_meta + (size_t)let_rank * kLetMetaWords;
        uint32_t *n_nodes = scalars + 1, *status = scalars + 4, *depth_base = scalars + 16;
        if (phase == NB_PHASE_LET_META) {
            const uint32_t init_meta[kLetMetaWords] = {0u, ~0u, ~0u, ~0u, 0u, 0u, 0u, 0u};
            NB_HIP_TRY(hipMemcpyAsync(my_meta, init_meta, sizeof init_meta, hipMemcpyHostToDevice, stream));
            if (n) {
                const uint32_t g = std::min<uint32_t>((n + 255) / 256, 512);
                hipLaunchKernelGGL(bound_kernel, dim3(g), b256, 0, stream, posm[s], n, my_meta);
                hipLaunchKernelGGL(let_meta_kernel, dim3(g), b256, 0, stream, posm[s], vel[s], acc[s], n,
                                   params.dt, my_meta);
            }
            NB_HIP_TRY(hipGetLastError());
            let_next = NB_PHASE_LET_BUILD;
            return NB_OK;
        }
        if (phase == NB_PHASE_LET_BUILD) {
            uint32_t *my_counts = let_counts + (size_t)let_rank * let_world;
            NB_HIP_TRY(hipMemsetAsync(my_counts, 0, sizeof(uint32_t) * let_world, stream));
            if (n) {
                NB_HIP_TRY(hipMemsetAsync(scalars, 0, sizeof(uint32_t) * 4, stream));
                // (segments too small for the one-launch export's reserved slots take the level-by-level form)
                const bool one_launch = let_export_mode == 1u && let_cap >= kLetReserved;
                hipLaunchKernelGGL(let_global_bound_kernel, dim3(1), dim3(1), 0, stream, let_meta, let_world,
                                   scalars + 0, my_counts, let_rank, one_launch ? kLetReserved : 1u);
                if (int rc = enqueue_build(true, true)) return rc;  // (a LET rank's velocities never travel)
                if (one_launch) {
                    hipLaunchKernelGGL(let_export_kernel, dim3(64, let_world), dim3(kLetExportThreads), 0, stream, rec,
                                       n_nodes, node_cap, let_meta, let_rank, let_prune, let_send,
                                       my_counts, let_cap, status);
                } else {
                    NB_HIP_TRY(hipMemsetAsync(let_out_slot, 0xff, sizeof(uint32_t) * (size_t)let_world * node_cap,
                                              stream));
                    for (int d = 0; d <= kMaxDepth; ++d)
                        hipLaunchKernelGGL(let_export_level_kernel, dim3(128, let_world), b256, 0, stream, rec,
                                           depth_base, d, n_nodes, node_cap, let_meta, let_rank,
                                           let_prune, let_out_slot, let_send, my_counts, let_cap, status);
                }
                hipLaunchKernelGGL(let_clamp_counts_kernel, dim3(1), dim3(64), 0, stream, my_counts, let_world,
                                   let_cap);
                NB_HIP_TRY(hipGetLastError());
            }
            let_next = NB_PHASE_LET_WALK;
            let_imports_set = false;
            return NB_OK;
        }
        // NB_PHASE_LET_WALK
        if (!let_imports_set) {
            set_error("LET walk: call nb_sim_let_set_imports with this step's import counts first");
            return NB_ERR_INVALID;
        }
        WalkRoots roots{};
        if (let_import_stride) {
            // imports in fixed-stride segments, their counts read on the device (no host round trip)
            hipLaunchKernelGGL(let_rebase_fixed_kernel, dim3(64, std::max(1, let_world - 1)), b256, 0, stream,
                               rec + node_cap, let_counts, (uint32_t)let_rank, (uint32_t)let_world,
                               let_import_stride, node_cap, (n && !let_own_walked) ? 1u : 0u, let_roots_dev, status);
            if (n) {
                if (int rc = enqueue_walk(roots, let_own_walked ? 2 : 0, let_own_walked ? 0u : 1u, let_roots_dev))
                    return rc;
            }
        } else {
        if (n && !let_own_walked) roots.id[roots.count++] = 0u;
        const uint32_t total = let_segs.off[let_segs.world];
        if (total) {
            hipLaunchKernelGGL(let_rebase_kernel, dim3((total + 255) / 256), b256, 0, stream, rec + node_cap,
                               let_segs, node_cap);
            for (uint32_t r = 0; r < let_segs.world; ++r)
                if (let_segs.off[r + 1] > let_segs.off[r]) roots.id[roots.count++] = node_cap + let_segs.off[r];
        }
        if (n) {
            // set 0 = the own tree (already walked by NB_PHASE_LET_WALK_OWN if let_own_walked), set 1 = the imports
            if (int rc = enqueue_walk(roots, let_own_walked ? 2 : 0, let_own_walked ? 0u : 1u)) return rc;
        }
        }
        let_own_walked = false;
        step_num += 1;
        let_next = NB_PHASE_LET_META;
        return NB_OK;
    }

    // NB_PHASE_LET_WALK_OWN (optional, between BUILD and WALK): the rank's own tree needs nothing
    // from the peers, so its part of the walk can run while the exported trees are exchanged;
    // NB_PHASE_LET_WALK then adds the imported trees and integrates (same sums, same order).
    int let_walk_own() {
        if (!let_world || let_next != NB_PHASE_LET_WALK || let_own_walked) {
            set_error("LET own-tree walk: only once, between NB_PHASE_LET_BUILD and NB_PHASE_LET_WALK");
            return NB_ERR_INVALID;
        }
        if (n) {
            WalkRoots roots{};
            roots.id[roots.count++] = 0u;
            if (int rc = enqueue_walk(roots, 1)) return rc;
        }
        let_own_walked = true;
        return NB_OK;
    }

    // NB_PHASE_LET_MIGRATE (optional, before NB_PHASE_LET_META): re-home the bodies whose position
    // left the rank's Morton-key range.  Stayers are compacted, leavers packed per owner into
    // region 5, counts (stayers at [rank]) into region 4; the caller all-gathers region 4, moves
    // the segments into region 6 and calls nb_sim_let_set_arrivals.
    int let_migrate() {
        if (!let_world || !let_send || !let_mig_send) {
            set_error("LET migrate: set tree_let_world / rank / cap and the owners (nb_sim_let_set_owners) first");
            return NB_ERR_INVALID;
        }
        if (let_next != NB_PHASE_LET_META) {
            set_error("LET migrate: only between steps");
            return NB_ERR_INVALID;
        }
        const int s = cur, d = cur ^ 1;
        uint32_t *my_counts = let_mig_counts + (size_t)let_rank * let_world;
        NB_HIP_TRY(hipMemsetAsync(my_counts, 0, sizeof(uint32_t) * let_world, stream));
        if (n)
            hipLaunchKernelGGL(let_migrate_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, posm[s], vel[s],
                               acc[s], n, let_owners, let_rank, posm[d], vel[d], acc[d], let_mig_send,
                               let_mig_cap, my_counts, scalars + 4);
        NB_HIP_TRY(hipGetLastError());
        cur = d;  // the compacted stayers are the state now (their number: counts[rank])
        let_arrivals_pending = true;
        return NB_OK;
    }

    int let_set_owners(const unsigned long long *splits, int world, float ref_bound, uint32_t seg_cap) override {
        if (!let_world || world != let_world || !splits || !(ref_bound > 0.f) || seg_cap == 0) {
            set_error("let_set_owners: bad arguments (world %d, tree_let_world %d)", world, let_world);
            return NB_ERR_INVALID;
        }
        if (int rc = bind_device()) return rc;
        let_owners.world = (uint32_t)world;
        let_owners.ref_bound = ref_bound;
        for (int r = 0; r < world; ++r) let_owners.split[r] = r + 1 < world ? splits[r] : ~0ull;
        if (!let_mig_send) {
            const size_t w = (size_t)world;
            let_mig_cap = seg_cap;
            if (int rc = alloc(&let_mig_counts, sizeof(uint32_t) * w * w)) return rc;
            if (int rc = alloc(&let_mig_send, sizeof(float4) * 3 * w * (size_t)seg_cap)) return rc;
            if (int rc = alloc(&let_mig_recv, sizeof(float4) * 3 * w * (size_t)seg_cap)) return rc;
            NB_HIP_TRY(hipMemsetAsync(let_mig_counts, 0, sizeof(uint32_t) * w * w, stream));
            NB_HIP_TRY(hipStreamSynchronize(stream));
        }
        return NB_OK;
    }

    // after the exchange: `stay` bodies were kept, counts[r] arrived from rank r (packed in rank
    // order in region 6); the simulator's body count becomes stay + sum(counts)
    int let_set_arrivals(uint32_t stay, const uint32_t *counts, int world) override {
        if (!let_arrivals_pending || world != let_world || !counts) {
            set_error("let_set_arrivals: no migration in flight (or world mismatch)");
            return NB_ERR_INVALID;
        }
        if (int rc = bind_device()) return rc;
        uint64_t total = 0;
        for (int r = 0; r < world; ++r) total += (r == let_rank) ? 0u : counts[r];
        if (stay > n || stay + total > n_capacity || total > (uint64_t)let_mig_cap * world) {
            set_error("let_set_arrivals: %u kept + %llu arrived exceed the capacity of %u bodies", stay,
                      (unsigned long long)total, n_capacity);
            return NB_ERR_UNSUPPORTED;
        }
        if (total)
            hipLaunchKernelGGL(let_append_kernel, dim3(((uint32_t)total + 255) / 256), dim3(256), 0, stream,
                               let_mig_recv, (uint32_t)total, stay, posm[cur], vel[cur], acc[cur]);
        NB_HIP_TRY(hipGetLastError());
        set_active(stay + (uint32_t)total);
        let_arrivals_pending = false;
        return NB_OK;
    }

    // the number of bodies this simulator currently holds (<= the capacity it was created with)
    void set_active(uint32_t count) {
        n = count;
        hi = count;
        lo = 0;
        params.particle_num = count;
        const size_t nn = n ? n : 1;  // (the tile size stays the one the buffers were sized for)
        sort_blocks = (uint32_t)((nn + kSortThreads * sort_items - 1) / (kSortThreads * sort_items));
    }

    // Imports of this step arrive in region 3 as (world - 1) segments of `stride` records, in rank
    // order with this rank skipped; how many records of a segment are real is taken on the device
    // from the all-gathered counts (region 1).  Replaces nb_sim_let_set_imports for this step.
    int let_set_import_stride(uint32_t stride) override {
        if (!let_world || !let_send || stride == 0 || stride > let_cap) {
            set_error("let_set_import_stride: needs the LET buffers and 0 < stride <= tree_let_cap (%u)", let_cap);
            return NB_ERR_INVALID;
        }
        let_import_stride = stride;
        let_imports_set = true;
        return NB_OK;
    }

    int let_set_imports(const uint32_t *counts, int world) override {
        if (!let_world || world != let_world || !counts) {
            set_error("let_set_imports: world %d does not match tree_let_world %d", world, let_world);
            return NB_ERR_INVALID;
        }
        uint64_t run = 0;
        let_segs.world = (uint32_t)world;
        for (int r = 0; r < world; ++r) {
            let_segs.off[r] = (uint32_t)run;
            run += (r == let_rank) ? 0u : counts[r];
            if (counts[r] > let_cap) {
                set_error("let_set_imports: %u records from rank %d exceed the capacity %u", counts[r], r, let_cap);
                return NB_ERR_INVALID;
            }
        }
        let_segs.off[world] = (uint32_t)run;
        let_import_stride = 0;
        let_imports_set = true;
        return NB_OK;
    }

    int let_setup(uint32_t cap) {
        if (let_world < 1 || let_world > kLetMaxWorld || let_rank < 0 || let_rank >= let_world) {
            set_error("LET: world %d / rank %d out of range (at most %d ranks)", let_world, let_rank, kLetMaxWorld);
            return NB_ERR_INVALID;
        }
        if (place.world != 1) {
            set_error("LET: create the TreeSim over the rank's own bodies (placement world 1)");
            return NB_ERR_INVALID;
        }
        if (let_send) {
            set_error("LET buffers are already allocated");
            return NB_ERR_INVALID;
        }
        if (int rc = bind_device()) return rc;
        let_cap = cap;
        const size_t w = (size_t)let_world;
        if (int rc = alloc(&let_meta, sizeof(uint32_t) * kLetMetaWords * w)) return rc;
        if (int rc = alloc(&let_counts, sizeof(uint32_t) * w * w)) return rc;
        if (int rc = alloc(&let_out_slot, sizeof(uint32_t) * w * (size_t)node_cap)) return rc;
        if (int rc = alloc(&let_roots_dev, sizeof(WalkRoots))) return rc;
        if (int rc = alloc(&let_send, sizeof(NodeRec) * w * (size_t)cap)) return rc;
        // the walk addresses own and imported records through one table: own tree first, imports after
        NodeRec *table = nullptr;
        if (int rc = alloc(&table, sizeof(NodeRec) * ((size_t)node_cap + w * (size_t)cap + 4))) return rc;
        rec = table;
        NB_HIP_TRY(hipMemsetAsync(let_meta, 0, sizeof(uint32_t) * kLetMetaWords * w, stream));
        NB_HIP_TRY(hipMemsetAsync(let_counts, 0, sizeof(uint32_t) * w * w, stream));
        NB_HIP_TRY(hipStreamSynchronize(stream));
        let_next = NB_PHASE_LET_META;
        return NB_OK;
    }

    int enqueue_step() {
        if (let_world) {
            set_error("this TreeSim runs the LET protocol: use nb_sim_encode_phase(NB_PHASE_LET_*)");
            return NB_ERR_INVALID;
        }
        // 8c: from kWalkGatherFrom bodies the walk fetches a body's velocity and acceleration through the order
        // itself instead of having cells_c_kernel copy all of them into sorted arrays first (64 B read + 64 B
        // written per body by a kernel that runs at the HBM limit); the new state then lands in the OTHER buffer set
        // (positions in place over the sorted source) and `cur` flips.  Whole one-GPU steps only; not under a captured
        // graph (its arguments would alternate).
        const bool gather = !build_done && place.world == 1 && !use_graph && walk_mode != 0 && walk_gathers != 0 &&
                            (walk_gathers > 1 || n >= kWalkGatherFrom);
        if (!build_done)
            if (int rc = enqueue_build(false, place.world == 1 && !gather)) return rc;
        build_done = false;
        return enqueue_walk(own_root(), 0, 1, nullptr, gather);
    }

    // external_bound: the root cube is already in scalars[0] (LET: the max over all ranks)
    // with_va: also reorder velocities and accelerations (they are complete: not a sharded host
    // that still gathers them while the tree is built)
    int enqueue_build(bool external_bound, bool with_va = false) {
        const int s = cur, d = cur ^ 1;
        uint32_t *bound_bits = scalars + 0, *n_nodes = scalars + 1, *status = scalars + 4;
        uint32_t *depth_base = scalars + 16;  // kMaxDepth + 2 entries
        uint32_t *bound_slots = bound_buf;    // kBoundSlots words: the next bound, from the walk
        const dim3 b256(256);
        const uint32_t g256 = (n + 255) / 256;
        // 1-2: bound + keys from the step's source positions (old positions, tree.rs:290-295)
        const uint32_t *bound_src = bound_bits;
        uint32_t n_src = 1;
        if (!external_bound) {
            if (bound_from_walk) {  // the previous step's walk has already taken max |coord| of this state
                bound_src = bound_slots;
                n_src = kBoundSlots;
            } else {
                NB_HIP_TRY(hipMemsetAsync(scalars, 0, sizeof(uint32_t) * 4, stream));  // status words are sticky
                hipLaunchKernelGGL(bound_kernel, dim3(std::min<uint32_t>(g256, 512)), b256, 0, stream, posm[s], n,
                                   bound_bits);
            }
        }
        bound_from_walk = false;
        const bool rank_sort = n <= rank_sort_max && sort_mode == 1;
        // 3 / 3d: stable radix passes of kSortBits bits -- over all 63 key bits, or (sort_mode 1) only
        // over the top `bits` bits, such that a cell of that level holds 1/64 body on average
        // (2^bits >= 64 N), followed by the fix-up of the runs that tie there.
        // Digits of 8 bits.  (9 where that saves a pass -- 25..27 bits: 262,145 .. 2,097,152 bodies -- is tuning
        // key "tree_sort_wide": measured twice and not faster, a 9-bit scatter costs 17 instead of 13 us at
        // 2^20 bodies and the fix-up sees 30x the runs, which eats the pass saved:
        // profiles/r02_sort_experiments.txt.)
        // How many: with the thread-per-body fix-up of the high-word sort (3e) ties are cheap, so only as many
        // bits as leave about two bodies per cell of the resolved level (2^bits >= N / 2: 16 bits up to 131,072
        // bodies, 24 up to 33 million -- a pass less than the 64-bit form needs, whose wave-per-run fix-up wants
        // 1/64 body per cell: 2^bits >= 64 N).  tree_sort_spare_hi / tree_sort_spare: log2 of cells per body.
        uint32_t bits = 63;
        bool hi_fit = false;
        if (sort_mode == 1) {
            auto bits_for = [&](int spare) {
                uint32_t b = 8u;
                while (b < 63u && std::ldexp(1.0, (int)b) < std::ldexp((double)n, spare)) ++b;
                return b;
            };
            // a step whose fix-up met a long run (a dense core in a cube stretched by escapers) makes the next
            // steps sort more high digits (wait(): sort_boost), until the probe says they can go again
            const uint32_t boost = rank_sort ? 0u : kSortBits * sort_boost;
            const uint32_t bits_hi = (sort_bits ? sort_bits : std::max(16u, bits_for(sort_spare_hi))) + boost;
            hi_fit = sort_hi && !rank_sort && !sort_wide && bits_hi <= 31u;
            bits = hi_fit ? bits_hi : std::min(63u, (sort_bits ? sort_bits : std::max(21u, bits_for((int)sort_spare))) + boost);
        }
        const uint32_t W = (bits + kSortWideBits - 1u) / kSortWideBits < (bits + kSortBits - 1u) / kSortBits && sort_wide
                               ? kSortWideBits : kSortBits;
        const uint32_t bins = 1u << W;
        const uint32_t passes = (bits + W - 1u) / W;
        const uint32_t shift0 = 63u > passes * W ? 63u - passes * W : 0u;  // the passes cover bits shift0 .. 62
        // 3e: up to 31 sorted bits all lie in the keys' HIGH WORDS, so the passes move (high word, index) --
        // 8-byte instead of 12-byte elements, and no identity index array to begin with -- the fix-up looks a
        // tied body's full key up through its index, and the sorted 64-bit keys are gathered once, by
        // cells_a_kernel beside the positions.  hs0: where the first digit sits inside the high word (three
        // passes: bits 7..30, the same 24 key bits as without; four: the whole word, key bits 32..62).
        const bool hi_mode = hi_fit && W == kSortBits && passes <= 4u;
        const uint32_t hs0 = 31u > kSortBits * passes ? 31u - kSortBits * passes : 0u;
        uint32_t *khi[2] = {reinterpret_cast<uint32_t *>(keys[1]), reinterpret_cast<uint32_t *>(keys[1]) + n};
        if (rank_sort)
            hipLaunchKernelGGL(morton_kernel, dim3((n + kSortThreads - 1) / kSortThreads), dim3(kSortThreads), 0, stream,
                               posm[s], n, bound_src, n_src, bound_bits, keys[0], idx[0], (uint32_t *)nullptr, 0u, 1u,
                               0u, 1u, (uint32_t *)nullptr, key_descent);
        else  // (with the tile histograms of the first pass's digit)
            // (512 threads x half the sort's items per thread: the same tile, twice the waves per SIMD for the
            // 21 dependent levels of the key descent)
            hipLaunchKernelGGL(morton_kernel, dim3(sort_blocks), dim3(2 * kSortThreads), 0, stream, posm[s], n, bound_src,
                               n_src, bound_bits, keys[0], idx[0], hist, sort_blocks, sort_items / 2u,
                               hi_mode ? 32u + hs0 : shift0, bins, hi_mode ? khi[0] : (uint32_t *)nullptr, key_descent);
        int kb = 0;
        if (rank_sort) {
            // 3c: the sorted position of every body counted in one launch
            hipLaunchKernelGGL(rank_sort_kernel, dim3((n + 63u) / 64u), dim3(64 * kRankWaves), 0, stream, keys[0], n,
                               keys[1], idx[1]);
            kb = 1;
        } else if (hi_mode) {
            const bool inl = sort_blocks <= kSortInlineScanBlocks;
            for (uint32_t ps = 0; ps < passes; ++ps) {
                const uint32_t shift = hs0 + ps * kSortBits;
                const uint32_t *vin = ps == 0u ? (const uint32_t *)nullptr : idx[kb];
#define NB_PASS_HI(ITEMS)                                                                                           \
    do {                                                                                                            \
        constexpr uint32_t TH = 2u * kSortThreads;                                                                  \
        constexpr uint32_t IT = kSortThreads * (ITEMS) / TH;                                                        \
        if (ps != 0u)                                                                                               \
            hipLaunchKernelGGL((radix_hist_kernel<IT, uint32_t>), dim3(sort_blocks), dim3(TH), 0, stream,          \
                               khi[kb], n, shift, bins, hist, sort_blocks);                                         \
        if (inl) {                                                                                                  \
            hipLaunchKernelGGL((radix_scatter_kernel<(int)kSortBits, TH, IT, true, uint32_t>), dim3(sort_blocks),   \
                               dim3(TH), 0, stream, khi[kb], vin, khi[kb ^ 1], idx[kb ^ 1], n, shift, hist, totals, \
                               sort_blocks);                                                                        \
        } else {                                                                                                    \
            hipLaunchKernelGGL(bin_scan_kernel, dim3(bins), b256, 0, stream, hist, sort_blocks, totals);            \
            hipLaunchKernelGGL((radix_scatter_kernel<(int)kSortBits, TH, IT, false, uint32_t>), dim3(sort_blocks),  \
                               dim3(TH), 0, stream, khi[kb], vin, khi[kb ^ 1], idx[kb ^ 1], n, shift, hist, totals, \
                               sort_blocks);                                                                        \
        }                                                                                                           \
    } while (0)
                if (sort_items == kSortItemsSmall) NB_PASS_HI(kSortItemsSmall);
                else NB_PASS_HI(kSortItems);
#undef NB_PASS_HI
                kb ^= 1;
            }
            {   // the ties: on the sorted high words, full keys through the indices; every body finds its place
                // and the order comes out in the other index array
                // (scratch for a long run's keys: the moment prefixes, which cells_c_kernel writes later)
                uint64_t *scratch = reinterpret_cast<uint64_t *>(mom_prefix);
                const uint32_t par = build_seq & 1u;
                hipLaunchKernelGGL(runs_rank_kernel, dim3((n + 256u * kRankItems - 1u) / (256u * kRankItems)), b256, 0, stream, khi[kb], keys[0], idx[kb],
                                   idx[kb ^ 1], scratch, scratch + n, n, 32u + hs0,
                                   sort_boost ? std::min(62u, 32u + hs0 + kSortBits) : 0u, scalars + 8 + par,
                                   scalars + 8 + (par ^ 1u));
                run_stat_seq = build_seq;
                run_stat_boost = sort_boost;
                run_stat_hi = true;
                ++build_seq;
                kb ^= 1;
            }
        } else {
            for (uint32_t ps = 0; ps < passes; ++ps) {
                const uint32_t shift = shift0 + ps * W;
                const bool inl = sort_blocks <= kSortInlineScanBlocks;
#define NB_PASS(WW, ITEMS)                                                                                          \
    do {                                                                                                            \
        constexpr uint32_t TH = 2u * kSortThreads;                                        /* scatter's threads */    \
        constexpr uint32_t IT = kSortThreads * (ITEMS) / TH;                              /* ... and items */        \
        if (ps != 0u)                                                                                               \
            hipLaunchKernelGGL((radix_hist_kernel<IT>), dim3(sort_blocks), dim3(TH), 0, stream,                     \
                               keys[kb], n, shift, bins, hist, sort_blocks);                                        \
        if (inl) {                                                                                                  \
            hipLaunchKernelGGL((radix_scatter_kernel<WW, TH, IT, true>), dim3(sort_blocks), dim3(TH), 0, stream,    \
                               keys[kb], idx[kb], keys[kb ^ 1], idx[kb ^ 1], n, shift, hist, totals, sort_blocks);  \
        } else {                                                                                                    \
            hipLaunchKernelGGL(bin_scan_kernel, dim3(bins), b256, 0, stream, hist, sort_blocks, totals);            \
            hipLaunchKernelGGL((radix_scatter_kernel<WW, TH, IT, false>), dim3(sort_blocks), dim3(TH), 0, stream,   \
                               keys[kb], idx[kb], keys[kb ^ 1], idx[kb ^ 1], n, shift, hist, totals, sort_blocks);  \
        }                                                                                                           \
    } while (0)
                if (W == kSortWideBits) {
                    if (sort_items == kSortItemsSmall) NB_PASS(kSortWideBits, kSortItemsSmall);
                    else NB_PASS(kSortWideBits, kSortItems);
                } else {
                    if (sort_items == kSortItemsSmall) NB_PASS(kSortBits, kSortItemsSmall);
                    else NB_PASS(kSortBits, kSortItems);
                }
#undef NB_PASS
                kb ^= 1;
            }
            if (shift0 || sort_boost) {  // (all 63 bits sorted: nothing to fix, but the probe still has to run)
                const uint32_t par = build_seq & 1u;
                hipLaunchKernelGGL(runs_fix_kernel, dim3((n + 256u * kRunItems - 1u) / (256u * kRunItems)), b256,
                                   0, stream, keys[kb], idx[kb], keys[kb ^ 1], idx[kb ^ 1], n, shift0,
                                   sort_boost ? std::min(62u, shift0 + W) : 0u, scalars + 8 + par, scalars + 8 + (par ^ 1u));
                run_stat_seq = build_seq;
                run_stat_boost = sort_boost;
                run_stat_hi = false;
                ++build_seq;
            }
        }
        // (hi_mode: cells_a_kernel gathers the sorted keys into the buffer the high words lived in)
        uint64_t *skeys = hi_mode ? keys[1] : keys[kb];
        order = idx[kb];
        sorted_keys = skeys;
        // 4-6a: the step's source permuted into DFS/Morton order (tree.rs:297,315-325), cells from
        // key prefixes, node ids, moment prefixes: A, B, C of section 5b
        // 256-body rounds per workgroup: small problems are bound by the chain of barriers inside a
        // workgroup (1 round), large ones by the length of the one-workgroup scan over the tiles (4)
        uint32_t rounds = n <= 131072u ? 1u : n <= 524288u ? 2u : kCellTile / 256u;
        if (cell_rounds && ((size_t)n + 256 * cell_rounds) / (256 * cell_rounds) + 1 <= cell_tiles) rounds = cell_rounds;
        const uint32_t ct = (uint32_t)(((size_t)n + 1 + 256 * rounds - 1) / (256 * rounds));  // covers prefix[n] too
        const uint32_t cstride = (ct + 3u) & ~3u;  // rows of the tile table, padded to 16 bytes
        if (hi_mode)
            hipLaunchKernelGGL((cells_a_kernel<true>), dim3(ct), b256, 0, stream, order, n, posm[s], posm[d], keys[0], skeys,
                               cpl, tile_u32, tile_mom, cstride, rounds, status);
        else
            hipLaunchKernelGGL((cells_a_kernel<false>), dim3(ct), b256, 0, stream, order, n, posm[s], posm[d], skeys,
                               (uint64_t *)nullptr, cpl, tile_u32, tile_mom, cstride, rounds, status);
        uint32_t *row_total = scalars + 40;  // kCellRows words
        if (ct <= (cell_scan_inline > 1 ? 256u : kCellInlineTiles) && cell_scan_inline) {
            hipLaunchKernelGGL((cells_c_kernel<true>), dim3(ct), b256, 0, stream, cpl, n, tile_u32, tile_mom, cstride, row_total,
                               depth_base, n_nodes, status, posm[d], int_slot, leaf_id, int_id, node_first, node_depth, mom_prefix, node_cap,
                               rounds, order, with_va ? vel[s] : (const float4 *)nullptr, acc[s], vel[d], acc[d], rec, bound_slots);
        } else {
            hipLaunchKernelGGL(cells_scan_kernel, dim3(kCellRows + 4), dim3(1024), 0, stream, tile_u32, tile_mom, ct, cstride,
                               row_total, bound_slots);
            hipLaunchKernelGGL((cells_c_kernel<false>), dim3(ct), b256, 0, stream, cpl, n, tile_u32, tile_mom, cstride, row_total,
                               depth_base, n_nodes, status, posm[d], int_slot, leaf_id, int_id, node_first, node_depth, mom_prefix, node_cap,
                               rounds, order, with_va ? vel[s] : (const float4 *)nullptr, acc[s], vel[d], acc[d], rec, bound_slots);
        }
        va_gathered = with_va;
        // 6: node contents
        const uint32_t gnodes = (std::min<uint64_t>(node_cap, 3ull * (n / 4u) + 256ull) + 255) / 256;  // internal cells
        if (n <= kFillEagerMax)
            hipLaunchKernelGGL((fill_kernel<false, true>), dim3(gnodes), b256, 0, stream, skeys, n, node_cap, n_nodes,
                               node_first, node_depth, cpl, int_slot, leaf_id, int_id, order, posm[d],
                               mom_prefix, depth_base, bound_bits, cogm, bodies, child, rec, inv_theta2(), scalars + 40);
        else if (n >= kFillEagerAgainFrom)
            hipLaunchKernelGGL((fill_kernel<false, true, false>), dim3(gnodes), b256, 0, stream, skeys, n, node_cap, n_nodes,
                               node_first, node_depth, cpl, int_slot, leaf_id, int_id, order, posm[d],
                               mom_prefix, depth_base, bound_bits, cogm, bodies, child, rec, inv_theta2(), scalars + 40);
        else
            hipLaunchKernelGGL((fill_kernel<false, false>), dim3(gnodes), b256, 0, stream, skeys, n, node_cap, n_nodes,
                               node_first, node_depth, cpl, int_slot, leaf_id, int_id, order, posm[d],
                               mom_prefix, depth_base, bound_bits, cogm, bodies, child, rec, inv_theta2(), scalars + 40);
        NB_HIP_TRY(hipGetLastError());
        return NB_OK;
    }

    // part: 0 whole step, 1 own-tree sums only, 2 continue from those sums and integrate
    int enqueue_walk(const WalkRoots &roots, int part = 0, uint32_t split = 1,
                     const WalkRoots *roots_dev = nullptr, bool gather = false) {
        const int s = cur, d = cur ^ 1;
        uint32_t *status = scalars + 4;
        const dim3 b256(256);
        if (part != 2 && !va_gathered && !gather)
            hipLaunchKernelGGL(gather_va_kernel, dim3((n + 255) / 256), b256, 0, stream, order, n, vel[s], acc[s],
                               vel[d], acc[d]);
        va_gathered = false;
        // 8: walk + integrate: sorted source (now in buffer d) -> buffer s.  A walk over the whole state in
        // one launch also leaves max |coord| of the new positions for the next step's root cube.
        const bool whole = part == 0 && !let_world && place.world == 1 && lo == 0 && hi == n;
        uint32_t *bslots = whole ? bound_buf : nullptr;
        if (time_walk) NB_HIP_TRY(hipEventRecord(time_walk[0], stream));
        if (hi > lo && walk_mode == 0) {
            // bodies per wave: 64 when that still gives >= 4096 waves (4 per SIMD), else halve down to 8
            uint32_t shift = 6;
            if (walk_bpw) {
                shift = walk_bpw >= 64 ? 6 : walk_bpw >= 32 ? 5 : walk_bpw >= 16 ? 4 : 3;
            } else {
                while (shift > 3 && ((hi - lo) >> shift) < 4096u) --shift;
            }
            const uint32_t per_block = 4u << shift;
            const dim3 gwalk((hi - lo + per_block - 1) / per_block);
#define NB_WALK(COUNT, PART)                                                                              \
    hipLaunchKernelGGL((walk_kernel<COUNT, PART>), gwalk, b256, 0, stream, posm[d], vel[d], acc[d], rec,  \
                       roots, posm[s], vel[s], acc[s], lo, hi, shift, params.g, params.e, params.dt,      \
                       status, counters, bslots, roots_dev)
            if (count_visits) {
                if (part == 0) NB_WALK(true, 0); else if (part == 1) NB_WALK(true, 1); else NB_WALK(true, 2);
            } else {
                if (part == 0) NB_WALK(false, 0); else if (part == 1) NB_WALK(false, 1); else NB_WALK(false, 2);
            }
#undef NB_WALK
        } else if (hi > lo) {
            // cells across the lanes (section 8b): a wave walks for a group of G bodies
            // bodies per wave: 8; 4 on small problems (below 24,576 bodies: twice the waves for the SIMDs a small
            // walk leaves idle, -5 .. -17 % of the walk at 8,192 and 16,384 bodies); 16 where a wide acceptance test
            // (theta >= 0.9) meets many bodies (from 393,216: -2 .. -3 %).  Measured after the walk stopped reading
            // the bound slots in its prologue -- until then a wave's start cost a trip to one hot cache line, and 16
            // bodies per wave, half the trips, "won" the middle sizes by up to 25 %:
            // profiles/r03_walk_experiments.txt section 4.
            // (by the tree's size, not by the range walked: the ranks of a replicated build add the same terms in
            // the same order as the one-GPU step -- bit for bit its result)
            const uint32_t gauto = n < 24576u ? 4u : (theta >= 0.9f && n >= 393216u) ? 16u : 8u;
            const uint32_t gsize = walk_group ? walk_group : gauto;
            const uint32_t per_block = kCellBlockWaves * gsize;
            const dim3 gwalk((hi - lo + per_block - 1) / per_block), bwalk(64 * kCellBlockWaves);
            // one-word stack entries when the group has 8 mask bits and every id is below 2^24
            const uint64_t id_limit = (uint64_t)node_cap + (let_world ? (uint64_t)let_world * let_cap : 0ull);
            const bool packed = walk_packed != 0 && gsize <= 8u && id_limit <= (1ull << kPackedIdBits);
            // (gather: sorted positions in buffer d, velocities / accelerations in source order in buffer s; the new
            // state -> buffer d)
            const float4 *w_vel = gather ? vel[s] : vel[d], *w_acc = gather ? acc[s] : acc[d];
            float4 *w_posm_dst = gather ? posm[d] : posm[s], *w_vel_dst = gather ? vel[d] : vel[s],
                   *w_acc_dst = gather ? acc[d] : acc[s];
            const uint32_t *w_order = gather ? order : nullptr;
#define NB_WALK(G, COUNT, PART, PACKED)                                                                           \
    hipLaunchKernelGGL((walk_cells_kernel<G, COUNT, PART, PACKED>), gwalk, bwalk, 0, stream, posm[d], w_vel,       \
                       w_acc, rec, roots, split, w_posm_dst, w_vel_dst, w_acc_dst, lo, hi, params.g, params.e,    \
                       params.dt, status, counters, bslots, roots_dev, w_order)
#define NB_WALK_P(G, COUNT, PACKED)                                                           \
    do {                                                                                      \
        if (part == 0) NB_WALK(G, COUNT, 0, PACKED);                                          \
        else if (part == 1) NB_WALK(G, COUNT, 1, PACKED);                                     \
        else NB_WALK(G, COUNT, 2, PACKED);                                                    \
    } while (0)
#define NB_WALK_G(COUNT)                                                                      \
    do {                                                                                      \
        if (gsize == 4u) { if (packed) NB_WALK_P(4, COUNT, true); else NB_WALK_P(4, COUNT, false); }   \
        else if (gsize == 16u) NB_WALK_P(16, COUNT, false);                                   \
        else { if (packed) NB_WALK_P(8, COUNT, true); else NB_WALK_P(8, COUNT, false); }      \
    } while (0)
            if (count_visits) NB_WALK_G(true); else NB_WALK_G(false);
#undef NB_WALK_G
#undef NB_WALK_P
#undef NB_WALK
        }
        if (time_walk) NB_HIP_TRY(hipEventRecord(time_walk[1], stream));
        NB_HIP_TRY(hipGetLastError());
        bound_from_walk = whole && hi > lo;
        // the post-step state is in buffer s (= cur); buffer d holds the sorted source
        // (gather: the post-step state is in buffer d, which becomes cur; s keeps the unsorted source)
        if (gather && hi > lo && walk_mode != 0) cur = d;
        return NB_OK;
    }

    int read_particles(nb_particle *dst, size_t count) override {
        if (count > n) {
            set_error("read_particles: asked for %zu of %u particles", count, n);
            return NB_ERR_INVALID;
        }
        if (int rc = bind_device()) return rc;
        if (count == 0) return wait();
        NB_HIP_TRY(launch_soa_to_aos(posm[cur], vel[cur], acc[cur], d_aos, n, 0, n, stream));
        NB_HIP_TRY(hipMemcpyAsync(dst, d_aos, sizeof(nb_particle) * count, hipMemcpyDeviceToHost, stream));
        NB_HIP_TRY(hipStreamSynchronize(stream));
        return check_status();
    }

    // device.poll(Wait) (offline_headless.rs:43) + the device status words: a step that overflowed
    // the 4N node buffer, cut a LET export short, met inseparable bodies or tripped the walk's
    // stack guard must not look like a good step to a caller that never reads particles back
    // (nb_runner_step, the headless CLI, timing loops).  The words ride the same stream: one
    // 32-byte copy into pinned memory ahead of the one synchronisation -- the four status words and the
    // fix-up's run statistics (runs_fix_kernel), which steer how many high digits the next builds sort.
    int wait() override {
        if (int rc = bind_device()) return rc;
        if (!h_status || !scalars) return SimBase::wait();
        NB_HIP_TRY(hipMemcpyAsync(h_status, scalars + 4, sizeof(uint32_t) * 12, hipMemcpyDeviceToHost, stream));
        NB_HIP_TRY(hipStreamSynchronize(stream));
        adapt_sort(h_status + 4);
        return report_status(h_status);
    }

    // st: {longest run; probe; bodies in long runs} x {parity 0, parity 1} of the last two fix-ups.
    // Speed only -- the sort's result does not depend on it -- so it does not matter which step's
    // statistics a given wait() happens to see.
    void adapt_sort(const uint32_t *st) {
        if (run_stat_seq == ~0u || run_stat_seq == run_stat_seen) return;  // no fix-up since the last look
        run_stat_seen = run_stat_seq;
        const uint32_t par = run_stat_seq & 1u, longest = st[par], probe = st[2 + par], slow = st[4 + par];
        const uint32_t before = sort_boost;
        // one more digit: a run that is radix-sorted by one workgroup, or (high-word sort: every run of 64 or
        // more takes a workgroup's turn) more than 1/64 of the bodies in such runs -- a disc, a dense core
        if (longest > kRunBoostAbove || slow > n / 64u) {
            sort_boost = std::min(kSortBoostMax, run_stat_boost + (longest > 256u * kRunBoostAbove ? 2u : 1u));
        } else if (run_stat_boost != 0u && sort_boost == run_stat_boost &&
                   (run_stat_hi ? probe <= n / 256u : probe == 0u)) {
            sort_boost = run_stat_boost - 1u;  // with one digit less the runs would still be short
        }
        if (sort_boost != before) drop_graph();  // the captured launch sequence has the old number of passes
    }

    int check_status() {
        uint32_t st[4] = {0, 0, 0, 0};
        NB_HIP_TRY(hipMemcpy(st, scalars + 4, sizeof st, hipMemcpyDeviceToHost));
        return report_status(st);
    }

    int report_status(const uint32_t *st) {
        if (st[0] & kLetListOverflow) {
            set_error("LET export: a tree level inside one grandchild of the root is wider than the one-launch export's "
                      "list (%u ranges of up to %u cells: a heavily clustered rank); the segments had room -- set "
                      "tree_let_export_mode 0 (one launch per tree level) for this simulator",
                      kLetExportRanges, kLetExportThreads);
            return NB_ERR_UNSUPPORTED;
        }
        if (st[0]) {
            set_error("LET export needs more than tree_let_cap = %u records for a peer (%u cells cut short)",
                      let_cap, st[0]);
            return NB_ERR_UNSUPPORTED;
        }
        if (st[3]) {
            set_error("tree walk hit its stack guard %u times (more than %u pending groups per wave: "
                      "inconsistent tree)", st[3], kWalkStack - 8);
            return NB_ERR_UNSUPPORTED;
        }
        if (st[1]) {
            set_error("octree needs more than %u nodes (4N, the reference's capacity, tree.rs:188-190)",
                      node_cap);
            return NB_ERR_UNSUPPORTED;
        }
        if (st[2]) {
            set_error("%u bodies share their 63-bit Morton key with a neighbour (closer than root_width / 2^21): "
                      "the octree cannot separate them (the reference's build_tree, tree.rs:473-544, never "
                      "terminates on such input)", st[2]);
            return NB_ERR_UNSUPPORTED;
        }
        return NB_OK;
    }

    int read_tree(nb_octant *dst, size_t cap, size_t *n_nodes_out, float *root_width) override {
        if (int rc = bind_device()) return rc;
        NB_HIP_TRY(hipStreamSynchronize(stream));
        uint32_t sc[2] = {0, 0};
        NB_HIP_TRY(hipMemcpy(sc, scalars, sizeof sc, hipMemcpyDeviceToHost));
        if (step_num == 0 || n == 0) {
            if (n_nodes_out) *n_nodes_out = 0;
            if (root_width) *root_width = 2.0f;  // TreeSimParams initial root_width, tree.rs:50
            return NB_OK;
        }
        const uint32_t nodes = std::min(sc[1], node_cap);
        float b;
        std::memcpy(&b, &sc[0], 4);
        if (root_width) *root_width = b * 2.0f;
        if (n_nodes_out) *n_nodes_out = nodes;
        const size_t m = std::min<size_t>(nodes, cap);
        if (m && dst) {
            if (!d_tree_aos) {  // first read-back: the reference's Octant fields, 104 B per node
                if (int rc = alloc(&cogm, sizeof(float4) * (size_t)node_cap)) return rc;
                if (int rc = alloc(&bodies, sizeof(uint32_t) * (size_t)node_cap)) return rc;
                if (int rc = alloc(&child, sizeof(uint32_t) * 8 * (size_t)node_cap)) return rc;
                if (int rc = alloc(&d_tree_aos, sizeof(nb_octant) * (size_t)node_cap)) return rc;
            }
            // the Octant fields are produced on demand from the step's build arrays, which stay
            // intact until the next step (buffer cur^1 holds the sorted source the tree was built on)
            hipLaunchKernelGGL((fill_kernel<true, false>), dim3((node_cap + 255) / 256), dim3(256), 0, stream, sorted_keys,
                               n, node_cap, scalars + 1, node_first, node_depth, cpl, int_slot, leaf_id, int_id,
                               order, posm[cur ^ 1], mom_prefix, scalars + 16, scalars + 0, cogm, bodies, child,
                               rec, inv_theta2(), scalars + 40);
            hipLaunchKernelGGL(tree_to_aos_kernel, dim3((nodes + 255) / 256), dim3(256), 0, stream, cogm,
                               bodies, child, nodes, d_tree_aos);
            NB_HIP_TRY(hipMemcpyAsync(dst, d_tree_aos, sizeof(nb_octant) * m, hipMemcpyDeviceToHost, stream));
            NB_HIP_TRY(hipStreamSynchronize(stream));
        }
        return check_status();
    }

    int encode_n_timed(int count, float *ms_total, float *ms_kernel) override {
        if (count <= 0) {
            set_error("encode_n_timed: n must be positive");
            return NB_ERR_INVALID;
        }
        if (int rc = bind_device()) return rc;
        while (events.size() < (size_t)(2 * count + 2)) {
            hipEvent_t ev;
            NB_HIP_TRY(hipEventCreate(&ev));
            events.push_back(ev);
        }
        NB_HIP_TRY(hipEventRecord(events[0], stream));
        for (int k = 0; k < count; ++k) {
            time_walk = &events[2 + 2 * k];  // brackets the walk kernel inside encode()
            const int rc = encode();
            time_walk = nullptr;
            if (rc) return rc;
        }
        NB_HIP_TRY(hipEventRecord(events[1], stream));
        NB_HIP_TRY(hipStreamSynchronize(stream));
        float total = 0.f, walk_sum = 0.f;
        NB_HIP_TRY(hipEventElapsedTime(&total, events[0], events[1]));
        for (int k = 0; k < count; ++k) {
            float ms = 0.f;
            NB_HIP_TRY(hipEventElapsedTime(&ms, events[2 + 2 * k], events[3 + 2 * k]));
            walk_sum += ms;
        }
        if (ms_total) *ms_total = total;
        if (ms_kernel) *ms_kernel = walk_sum / (float)count;  // the dominant kernel: the walk
        return check_status();  // a degraded step must not be reported as a timing
    }

    // Sharded TreeSim = replicated tree, partitioned walk: after encode the
    // rank's range of the three state arrays is new; the caller all-gathers each in place.
    // In LET mode the regions are the protocol's four buffers: 0 meta words (all-gather, 32 B per
    // rank), 1 export counts (all-gather, one row of `world` u32 per rank), 2 the export segments
    // (segment q = records for peer q, stride = slice_bytes), 3 the import area (packed by the
    // caller in rank order, skipping itself).
    int exchange_count() override { return let_world ? (let_mig_send ? 7 : 4) : 3; }
    int exchange_region(int index, void **dev_ptr, size_t *off, size_t *len, size_t *total) override {
        if (let_world) {
            if (index < 0 || index > (let_mig_send ? 6 : 3) || !let_send) {
                set_error("LET exchange region %d out of range (4 regions after tree_let_cap is set, 7 with "
                          "nb_sim_let_set_owners)", index);
                return NB_ERR_INVALID;
            }
            const size_t w = (size_t)let_world;
            void *base = nullptr;
            size_t o = 0, l = 0, t = 0;
            switch (index) {
            case 0: base = let_meta; l = sizeof(uint32_t) * kLetMetaWords; o = l * let_rank; t = l * w; break;
            case 1: base = let_counts; l = sizeof(uint32_t) * w; o = l * let_rank; t = l * w; break;
            case 2: base = let_send; l = sizeof(NodeRec) * (size_t)let_cap; t = l * w; break;
            case 3: base = rec + node_cap; l = sizeof(NodeRec) * (size_t)let_cap; t = l * w; break;
            // migration: 4 counts (all-gather, `world` u32 per rank, stayers at [rank]), 5 leavers per
            // owner (48 B per body, segment stride = slice_bytes), 6 arrivals (packed in rank order)
            case 4: base = let_mig_counts; l = sizeof(uint32_t) * w; o = l * let_rank; t = l * w; break;
            case 5: base = let_mig_send; l = sizeof(float4) * 3 * (size_t)let_mig_cap; t = l * w; break;
            default: base = let_mig_recv; l = sizeof(float4) * 3 * (size_t)let_mig_cap; t = l * w; break;
            }
            if (dev_ptr) *dev_ptr = base;
            if (off) *off = o;
            if (len) *len = l;
            if (total) *total = t;
            return NB_OK;
        }
        if (index < 0 || index > 2) {
            set_error("exchange region %d out of range (TreeSim has 3)", index);
            return NB_ERR_INVALID;
        }
        float4 *base = index == 0 ? posm[cur] : index == 1 ? vel[cur] : acc[cur];
        if (dev_ptr) *dev_ptr = base;
        if (off) *off = sizeof(float4) * (size_t)per_rank * (size_t)place.rank;
        if (len) *len = sizeof(float4) * (size_t)per_rank;
        if (total) *total = sizeof(float4) * (size_t)n_pad;
        return NB_OK;
    }

    int push_exchange(void *const *peer_bases, int npeers) override {
        if (let_world || npeers < 0 || npeers > kMaxPeers) {
            set_error("push_exchange: replicated-tree placements only, at most %d peers", kMaxPeers);
            return NB_ERR_INVALID;
        }
        const uint32_t first = per_rank * (uint32_t)place.rank;
        const uint32_t count = first < n ? std::min(per_rank, n - first) : 0u;
        if (npeers == 0 || count == 0u) return NB_OK;
        if (int rc = bind_device()) return rc;
        PushDst dst;
        dst.n = (uint32_t)npeers;
        for (int q = 0; q < npeers; ++q)
            for (int k = 0; k < 3; ++k) dst.p[k][q] = static_cast<float4 *>(peer_bases[q * 3 + k]);
        hipLaunchKernelGGL(push_slices_kernel, dim3((count + 255u) / 256u), dim3(256), 0, stream, posm[cur], vel[cur],
                           acc[cur], dst, first, count);
        NB_HIP_TRY(hipGetLastError());
        return NB_OK;
    }

    int push_region(int k, void *const *peer_bases, int npeers) override {
        void *base = nullptr;
        size_t off = 0, len = 0, total = 0;
        if (int rc = exchange_region(k, &base, &off, &len, &total)) return rc;
        if (npeers < 0 || npeers > kMaxPeers || (off & 3u) || (len & 3u) || len > (1u << 20)) {
            set_error("push_region: a small table of 4-byte words and at most %d peers", kMaxPeers);
            return NB_ERR_INVALID;
        }
        if (npeers == 0 || len == 0) return NB_OK;
        if (int rc = bind_device()) return rc;
        PushWords dst;
        dst.n = (uint32_t)npeers;
        for (int q = 0; q < npeers; ++q) dst.p[q] = static_cast<uint32_t *>(peer_bases[q]);
        const uint32_t count = (uint32_t)(len / 4u);
        hipLaunchKernelGGL(push_words_kernel, dim3((count + 63u) / 64u), dim3(64), 0, stream,
                           static_cast<const uint32_t *>(base), dst, (uint32_t)(off / 4u), count);
        NB_HIP_TRY(hipGetLastError());
        return NB_OK;
    }

    int let_push_segments(void *const *import_bases, int world, uint32_t stride) override {
        if (!let_world || world != let_world || !let_send || stride == 0 || stride > let_cap) {
            set_error("let_push_segments: needs the LET buffers, tree_let_world ranks and 0 < stride <= tree_let_cap");
            return NB_ERR_INVALID;
        }
        if (world < 2) return NB_OK;
        if (int rc = bind_device()) return rc;
        LetImportPtrs imp{};
        for (int q = 0; q < world; ++q) imp.p[q] = static_cast<NodeRec *>(import_bases[q]);
        hipLaunchKernelGGL(let_push_segments_kernel, dim3(64, (uint32_t)world), dim3(256), 0, stream, let_send, let_cap,
                           let_counts + (size_t)let_rank * (size_t)let_world, imp, (uint32_t)let_rank, stride);
        NB_HIP_TRY(hipGetLastError());
        return NB_OK;
    }

    int set_tuning(const char *key, int value) override {
        if (std::strcmp(key, "tree_count_visits") == 0) {
            count_visits = value != 0;
            drop_graph();  // a different walk kernel: re-capture
            return NB_OK;
        }
        if (std::strcmp(key, "tree_walk_bpw") == 0) {  // bodies per wave: 0 = automatic, else 8/16/32/64
            walk_bpw = value < 0 ? 0 : (uint32_t)value;
            drop_graph();
            return NB_OK;
        }
        if (std::strcmp(key, "tree_walk_mode") == 0) {  // 0: bodies across the lanes, 1: cells across the lanes
            walk_mode = value != 0 ? 1u : 0u;
            drop_graph();
            return NB_OK;
        }
        if (std::strcmp(key, "tree_walk_packed") == 0) {  // 1: one-word stack entries where the ids allow it (default), 0: never
            walk_packed = value != 0 ? 1u : 0u;
            drop_graph();
            return NB_OK;
        }
        if (std::strcmp(key, "tree_walk_group") == 0) {  // bodies per wave of mode 1: 0 = automatic, else 4/8/16
            if (value != 0 && value != 4 && value != 8 && value != 16) {
                set_error("tree_walk_group must be 0, 4, 8 or 16");
                return NB_ERR_INVALID;
            }
            walk_group = (uint32_t)value;
            drop_graph();
            return NB_OK;
        }
        if (std::strcmp(key, "tree_cell_rounds") == 0) {  // 256-body rounds per workgroup of cells_a / cells_c; 0 = automatic
            if (value < 0 || value > 4) {
                set_error("tree_cell_rounds must be 0 .. 4");
                return NB_ERR_INVALID;
            }
            cell_rounds = (uint32_t)value;
            drop_graph();
            return NB_OK;
        }
        if (std::strcmp(key, "tree_sort_hi") == 0) {  // 1: radix passes on (high word, index) where <= 31 bits are sorted (default)
            sort_hi = value != 0 ? 1u : 0u;
            drop_graph();
            return NB_OK;
        }
        if (std::strcmp(key, "tree_sort_bits") == 0) {  // the high key bits the radix passes sort (0: by tree_sort_spare)
            if (value < 0 || value > 63 || (value > 0 && value < 8)) {
                set_error("tree_sort_bits must be 0 or 8..63");
                return NB_ERR_INVALID;
            }
            sort_bits = (uint32_t)value;
            drop_graph();
            return NB_OK;
        }
        if (std::strcmp(key, "tree_sort_spare_hi") == 0) {  // ... for the high-word sort (default -1: two bodies per cell)
            if (value < -8 || value > 12) {
                set_error("tree_sort_spare_hi must be -8..12");
                return NB_ERR_INVALID;
            }
            sort_spare_hi = value;
            drop_graph();
            return NB_OK;
        }
        if (std::strcmp(key, "tree_sort_spare") == 0) {  // log2 of the cells per body at the level the radix passes resolve
            if (value < 0 || value > 12) {
                set_error("tree_sort_spare must be 0..12");
                return NB_ERR_INVALID;
            }
            sort_spare = (uint32_t)value;
            drop_graph();
            return NB_OK;
        }
        if (std::strcmp(key, "tree_sort_wide") == 0) {  // 1: 9-bit digits where they save a pass, 0: always 8 (default)
            sort_wide = value != 0 ? 1u : 0u;
            drop_graph();
            return NB_OK;
        }
        if (std::strcmp(key, "tree_sort_mode") == 0) {  // 1: counting sort (<= 12,288 bodies) / high digits + fix-up;
                                                        // 0: always the full 8-pass radix sort
            sort_mode = value != 0 ? 1u : 0u;
            drop_graph();
            return NB_OK;
        }
        if (std::strcmp(key, "tree_walk_gathers") == 0) {  // 1: the walk gathers velocities from kWalkGatherFrom bodies (default),
                                                          // 0: cells_c sorts them first at every size, 2: gathers at every size
            walk_gathers = value;
            return NB_OK;
        }
        if (std::strcmp(key, "tree_key_descent") == 0) {  // 1: the keys by the 21-level descent even in a power-of-two cube
            key_descent = value != 0 ? 1u : 0u;
            drop_graph();
            return NB_OK;
        }
        if (std::strcmp(key, "tree_cell_scan_inline") == 0) {  // 1: cells_c sums the tile table itself up to 64 tiles (default)
            cell_scan_inline = value;  // (2: up to the 256 tiles the kernel can do)
            drop_graph();
            return NB_OK;
        }
        if (std::strcmp(key, "tree_rank_sort_max") == 0) {  // the counting sort up to this many bodies (default 12,288)
            rank_sort_max = (uint32_t)std::max(value, 0);
            drop_graph();
            return NB_OK;
        }
        if (std::strcmp(key, "tree_use_graph") == 0) {
            use_graph = value != 0;
            drop_graph();
            return NB_OK;
        }
        if (std::strcmp(key, "tree_let_world") == 0) {
            let_world = value;
            return NB_OK;
        }
        if (std::strcmp(key, "tree_let_rank") == 0) {
            let_rank = value;
            return NB_OK;
        }
        if (std::strcmp(key, "tree_let_active") == 0) {  // bodies in use; the rest of the capacity is headroom
            if (value < 0 || (uint32_t)value > n_capacity || !let_world) {
                set_error("tree_let_active: %d out of range (capacity %u; LET mode only)", value, n_capacity);
                return NB_ERR_INVALID;
            }
            set_active((uint32_t)value);
            return NB_OK;
        }
        if (std::strcmp(key, "tree_let_export_mode") == 0) {  // 1: the export in one launch, 0: a launch per tree level
            let_export_mode = value != 0 ? 1u : 0u;
            return NB_OK;
        }
        if (std::strcmp(key, "tree_let_prune") == 0) {  // 0: export whole trees (testing: same result)
            let_prune = value != 0;
            return NB_OK;
        }
        if (std::strcmp(key, "tree_let_cap") == 0) {  // records per peer; allocates the LET buffers
            if (value <= 0) {
                set_error("tree_let_cap must be positive");
                return NB_ERR_INVALID;
            }
            return let_setup((uint32_t)value);
        }
        return SimBase::set_tuning(key, value);
    }

    int debug_buffer(const char *name, void *dst, size_t cap, size_t *bytes) override {
        if (int rc = bind_device()) return rc;
        NB_HIP_TRY(hipStreamSynchronize(stream));
        const void *src = nullptr;
        size_t len = 0;
        const std::string nm(name);
        if (nm == "order") { src = order; len = sizeof(uint32_t) * n; }
        else if (nm == "counters") { src = counters; len = sizeof(unsigned long long) * 16; }
#if defined(NB_DIAG_PHASES) || defined(NB_DIAG_TIMELINE)
        else if (nm == "phases") { src = counters + 16; len = sizeof(unsigned long long) * 4 * ((n + 3) / 4); }  // (8 words per group of 8)
#endif
        else if (nm == "status") { src = scalars + 4; len = sizeof(uint32_t) * 4; }
        else if (nm == "depth_base") { src = scalars + 16; len = sizeof(uint32_t) * (kMaxDepth + 2); }
        else {
            set_error("unknown debug buffer '%s'", name);
            return NB_ERR_INVALID;
        }
        if (bytes) *bytes = len;
        if (!src || !dst) return NB_OK;
        NB_HIP_TRY(hipMemcpy(dst, src, std::min(len, cap), hipMemcpyDeviceToHost));
        return NB_OK;
    }

   private:
    template <typename T>
    int alloc(T **p, size_t bytes) {
        void *q = nullptr;
        NB_HIP_TRY(hipMalloc(&q, bytes ? bytes : 16));
        allocs.push_back(q);
        *p = static_cast<T *>(q);
        return NB_OK;
    }

    float theta = NB_DEFAULT_THETA;
    // what fill_kernel scales a cell's size^2 by (NodeRec::mac2); theta = 0 gives +inf: every cell is opened
    float inv_theta2() const { return 1.0f / (theta * theta); }
    int cur = 0;  // posm/vel/acc[cur] hold the current state
    float4 *posm[2] = {nullptr, nullptr}, *vel[2] = {nullptr, nullptr}, *acc[2] = {nullptr, nullptr};
    uint64_t *keys[2] = {nullptr, nullptr};
    uint32_t *idx[2] = {nullptr, nullptr}, *order = nullptr;
    uint64_t *sorted_keys = nullptr;  // of the last build
    nb_particle *d_aos = nullptr;
    nb_octant *d_tree_aos = nullptr;
    uint32_t *hist = nullptr, *totals = nullptr, *int_slot = nullptr;
    uint32_t *leaf_id = nullptr;
    uint2 *int_id = nullptr;  // per internal-cell slot: {first body | opens-next flag, id | depth << 27}
    uint32_t *node_first = nullptr, *bodies = nullptr, *child = nullptr, *scalars = nullptr;
    uint8_t *node_depth = nullptr;
    int8_t *cpl = nullptr;
    float4 *cogm = nullptr;
    NodeRec *rec = nullptr;  // per node: cogm + {first child id, child count} / leaf {sorted position, 0}
    Moments *mom_prefix = nullptr;
    unsigned long long *counters = nullptr;
    uint32_t node_cap = 0, sort_blocks = 0, sort_items = kSortItems;
    bool count_visits = false, use_graph = false;
    uint32_t walk_bpw = 0;
    uint32_t walk_mode = 1, walk_group = 0, sort_mode = 1, cell_rounds = 0, walk_packed = 1, sort_wide = 0, sort_hi = 1,
             sort_spare = 6, sort_bits = 0;
    int sort_spare_hi = -1;
    uint32_t *tile_u32 = nullptr;
    bool bound_from_walk = false;  // bound_buf holds max |coord| of the current state
    uint32_t *bound_buf = nullptr;  // kBoundSlots words
    bool va_gathered = false;      // the build has already reordered velocities and accelerations
    Moments *tile_mom = nullptr;
    uint32_t cell_tiles = 0;
    bool build_done = false;  // phase 0 of the next step already enqueued
    // locally essential trees (section 9); let_world == 0: not in use
    int let_world = 0, let_rank = 0, let_next = 0;
    uint32_t let_cap = 0;
    uint32_t *let_meta = nullptr, *let_counts = nullptr, *let_out_slot = nullptr;
    uint32_t let_export_mode = 1;
    NodeRec *let_send = nullptr;
    LetSegments let_segs{};
    uint32_t let_import_stride = 0;       // != 0: this step's imports are fixed-stride segments
    WalkRoots *let_roots_dev = nullptr;
    bool let_imports_set = false, let_prune = true, let_arrivals_pending = false, let_own_walked = false;
    uint32_t n_capacity = 0, let_mig_cap = 0;
    uint32_t *let_mig_counts = nullptr;
    float4 *let_mig_send = nullptr, *let_mig_recv = nullptr;
    LetOwners let_owners{};
    hipGraphExec_t graph_exec = nullptr;
    uint32_t *h_status = nullptr;  // pinned mirror of the device status words (wait())
    // extra high digits the radix passes cover (adapt_sort), and the fix-up launch its statistics belong to
    static constexpr uint32_t kSortBoostMax = 6;
    uint32_t sort_boost = 0, build_seq = 0, run_stat_seq = ~0u, run_stat_seen = ~0u, run_stat_boost = 0;
    bool run_stat_hi = false;
    hipEvent_t *time_walk = nullptr;
    uint32_t rank_sort_max = kRankSortMax;
    uint32_t key_descent = 0;
    int walk_gathers = 1;
    int cell_scan_inline = 1;
    std::vector<void *> allocs;
    std::vector<hipEvent_t> events;
};

}  // namespace

SimBase *make_tree_sim() { return new (std::nothrow) TreeSim(); }

}  // namespace nb
