/*
 * nbody_oracle_tree.c -- CPU restatement of the reference's Barnes-Hut step.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as nbody_oracle.c: only tests/, smoke()
 * and bench.py's cpu_baseline leg may load it).  PARITY UNPINNED: the reference
 * has no tests or golden vectors for this path and cannot be built here.
 *
 * What it restates (paths relative to the reference crate root):
 *   src/sims/tree.rs:424-446    bound = max(1, max |coord|)           (A10)
 *   src/sims/tree.rs:458-546    serial BFS octree build               (A11)
 *   src/sims/tree.rs:549-553    decide_octant  (strict >)
 *   src/sims/tree.rs:556-562    shift_node_center
 *   src/sims/tree.rs:564-602    sort_particles(_recursive): DFS order (A12)
 *   src/sims/tree.rs:605-622    Octant, 52 bytes
 *   src/utils/slice_alloc.rs:52-63  node id = allocation order
 *   src/sims/shaders/tree.wgsl:41-90   getAcc: explicit-stack walk    (A14)
 *   src/sims/shaders/tree.wgsl:92-111  main: integrator               (A15)
 *   src/sims/tree.rs:262-353    encode: build on src, reorder src, walk src->dst
 *
 * The reference walk has three defects.  `flags` selects, per
 * defect, the literal behaviour (bit clear) or the intended one (bit set):
 *   NBO_SELF_BY_IDENTITY  D1: skip the body's own leaf by identity instead of
 *                             "bodies==1 && dist < 1e-6" (tree.wgsl:58-62), which
 *                             misses as soon as the body has drifted.
 *   NBO_LEAF_IS_BODY      D2: a leaf is always a single body (accumulate it)
 *                             instead of "descending" into children[0], which
 *                             is a PARTICLE index pushed as an OCTANT index
 *                             (tree.rs:532 vs tree.wgsl:80-85).
 *   NBO_CHECKED_STACK     D3: the walk stack grows as needed instead of 64
 *                             unchecked entries (tree.wgsl:44-45).  With the bit
 *                             clear a push past 64 aborts that body's walk and
 *                             sets *overflowed (the WGSL behaviour is undefined).
 * flags = 7 is what the HIP kernels implement; flags = 0 exists to quantify the
 * difference.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define NBO_SELF_BY_IDENTITY 1u
#define NBO_LEAF_IS_BODY 2u
#define NBO_CHECKED_STACK 4u

typedef struct nbo_octant {
    float cog[3];
    float mass;
    uint32_t bodies;
    uint32_t children[8];
} nbo_octant;

typedef struct part {
    float center[3];
    float width;
    uint32_t node;
    uint32_t start; /* into the level's index buffer */
    uint32_t count;
} part;

/* tree.rs:424-446: rayon reduce with identity position [1.0;3] => never < 1. */
float nbo_tree_bound(const float *p, uint32_t n) {
    float b0 = 1.0f, b1 = 1.0f, b2 = 1.0f;
    for (uint32_t i = 0; i < n; ++i) {
        const float *q = p + (size_t)i * 10;
        b0 = fmaxf(b0, fabsf(q[0]));
        b1 = fmaxf(b1, fabsf(q[1]));
        b2 = fmaxf(b2, fabsf(q[2]));
    }
    return fmaxf(fmaxf(b0, b1), b2);
}

/* tree.rs:549-553 */
static inline uint32_t decide_octant(const float *c, const float *x) {
    return (uint32_t)(x[0] > c[0]) | ((uint32_t)(x[1] > c[1]) << 1) | ((uint32_t)(x[2] > c[2]) << 2);
}

/* tree.rs:458-546.  particles: n x 10 floats.  tree: room for `cap` octants
 * (the reference allocates 4n, tree.rs:188-190).  Returns the node count, or
 * -1 if the build needs more than `cap` nodes or deeper than `max_depth`
 * levels (coincident bodies never terminate in the reference). */
int64_t nbo_tree_build(const float *particles, uint32_t n, nbo_octant *tree, uint64_t cap,
                       uint32_t max_depth, float *root_width_out) {
    const float bound = nbo_tree_bound(particles, n);
    if (root_width_out) *root_width_out = bound * 2.0f; /* tree.rs:448-451 */
    if (cap < 1) return -1;
    uint64_t alloced = 0;
    memset(&tree[alloced++], 0, sizeof(nbo_octant)); /* root_ix = write(default), :461 */

    uint32_t *cur = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t *nxt = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    size_t qcap = 1024, qhead = 0, qtail = 0;
    part *queue = (part *)malloc(sizeof(part) * qcap);
    if (!cur || !nxt || !queue) {
        free(cur); free(nxt); free(queue);
        return -1;
    }
    for (uint32_t i = 0; i < n; ++i) cur[i] = i; /* 0..len in index order, :467-470 */
    queue[qtail++] = (part){{0.0f, 0.0f, 0.0f}, bound * 2.0f, 0u, 0u, n};
    size_t level_end = qtail; /* queue entries [qhead, level_end) share one depth */
    uint32_t nxt_fill = 0, depth = 0;
    int64_t result = 0;

    while (qhead < qtail) { /* FIFO, :473 */
        if (qhead == level_end) { /* next depth: children lists live in nxt */
            uint32_t *t = cur; cur = nxt; nxt = t;
            nxt_fill = 0;
            level_end = qtail;
            if (++depth > max_depth) { result = -1; break; }
        }
        const part pt = queue[qhead++];
        nbo_octant oct;
        memset(&oct, 0, sizeof oct);
        uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const uint32_t *list = cur + pt.start;
        for (uint32_t k = 0; k < pt.count; ++k) { /* :486-501 */
            const float *p = particles + (size_t)list[k] * 10;
            oct.cog[0] += p[0] * p[9];
            oct.cog[1] += p[1] * p[9];
            oct.cog[2] += p[2] * p[9];
            oct.mass += p[9];
            cnt[decide_octant(pt.center, p)]++;
        }
        oct.bodies += pt.count; /* :502 */
        oct.cog[0] /= oct.mass; /* :503-505 */
        oct.cog[1] /= oct.mass;
        oct.cog[2] /= oct.mass;
        /* stable 8-way split of the list (each child list keeps index order) */
        uint32_t off[8], fill[8];
        {
            uint32_t o = nxt_fill;
            for (int c = 0; c < 8; ++c) { off[c] = o; fill[c] = o; if (cnt[c] > 1) o += cnt[c]; }
            /* lists of 0/1 bodies need no storage; remember the single body below */
        }
        uint32_t single[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t k = 0; k < pt.count; ++k) {
            const uint32_t ix = list[k];
            const uint32_t c = decide_octant(pt.center, particles + (size_t)ix * 10);
            if (cnt[c] > 1) nxt[fill[c]++] = ix; else single[c] = ix;
        }
        for (int c = 0; c < 8; ++c) if (cnt[c] > 1) nxt_fill += cnt[c];

        for (uint32_t c = 0; c < 8; ++c) { /* :507-541 */
            if (cnt[c] == 0) continue; /* children[c] stays 0 */
            if (alloced >= cap) { result = -1; goto done; }
            const uint32_t child_ix = (uint32_t)alloced++;
            memset(&tree[child_ix], 0, sizeof(nbo_octant));
            oct.children[c] = child_ix;
            if (cnt[c] == 1) { /* leaf, :521-534 */
                const float *lp = particles + (size_t)single[c] * 10;
                nbo_octant leaf;
                memset(&leaf, 0, sizeof leaf);
                leaf.cog[0] = lp[0]; leaf.cog[1] = lp[1]; leaf.cog[2] = lp[2];
                leaf.mass = lp[9];
                leaf.bodies = 1;
                leaf.children[0] = single[c];
                tree[child_ix] = leaf;
            } else { /* :535-539 */
                if (qtail == qcap) {
                    /* compact + grow */
                    if (qhead > 0) {
                        memmove(queue, queue + qhead, sizeof(part) * (qtail - qhead));
                        level_end -= qhead; qtail -= qhead; qhead = 0;
                    }
                    if (qtail * 2 > qcap) {
                        qcap *= 2;
                        part *nq = (part *)realloc(queue, sizeof(part) * qcap);
                        if (!nq) { result = -1; goto done; }
                        queue = nq;
                    }
                }
                part ch;
                const float q = pt.width / 4.0f; /* shift_node_center, :556-562 */
                ch.center[0] = pt.center[0] + (float)((int)(c & 1u) * 2 - 1) * q;
                ch.center[1] = pt.center[1] + (float)((int)((c & 2u) >> 1) * 2 - 1) * q;
                ch.center[2] = pt.center[2] + (float)((int)((c & 4u) >> 2) * 2 - 1) * q;
                ch.width = pt.width / 2.0f; /* :478 */
                ch.node = child_ix;
                ch.start = off[c];
                ch.count = cnt[c];
                queue[qtail++] = ch;
            }
        }
        tree[pt.node] = oct; /* :543 */
    }
    if (result == 0) result = (int64_t)alloced; /* tree_alloc.len(), :545 */
done:
    free(cur); free(nxt); free(queue);
    return result;
}

/* tree.rs:564-602: DFS over children 0..7, leaves emit src[children[0]].
 * order[k] = source index of the body placed at position k.  Iterative. */
void nbo_tree_dfs_order(const nbo_octant *tree, uint32_t n, uint32_t *order) {
    if (n == 0) return;
    if (n == 1) { order[0] = 0; return; } /* the reference is ill-defined at n==1 */
    uint32_t cap = 256, top = 0, out = 0;
    uint32_t *stack = (uint32_t *)malloc(sizeof(uint32_t) * cap);
    stack[top++] = 0;
    while (top) {
        const nbo_octant *o = &tree[stack[--top]];
        if (o->bodies == 1) { order[out++] = o->children[0]; continue; }
        if (top + 8 > cap) { cap *= 2; stack = (uint32_t *)realloc(stack, sizeof(uint32_t) * cap); }
        for (int c = 7; c >= 0; --c) if (o->children[c]) stack[top++] = o->children[c];
    }
    free(stack);
}

void nbo_gather_particles(const float *src, const uint32_t *order, uint32_t n, float *dst) {
    for (uint32_t k = 0; k < n; ++k) memcpy(dst + (size_t)k * 10, src + (size_t)order[k] * 10, 40);
}

typedef struct walk_stats {
    uint64_t visits;      /* nodes popped */
    uint64_t accepted;    /* force evaluations */
    uint32_t high_water;  /* deepest stack */
    uint32_t overflowed;  /* bodies whose literal walk pushed past 64 entries */
    uint32_t bad_index;   /* literal D2 pushes of an index >= n_nodes (UB in WGSL) */
} walk_stats;

/* tree.wgsl:41-90 for the body at sorted position `index`, new position a[3]. */
static void walk_one(const nbo_octant *tree, uint64_t n_nodes, float root_width, float theta,
                     float g, float e, float dt, const float *a, uint32_t index,
                     const uint32_t *order, uint32_t flags, float *acc_out, walk_stats *st) {
    float acc[3] = {0.0f, 0.0f, 0.0f};
    uint32_t cap = 512, size = 1, hw = 1;
    uint32_t *oct_stack = (uint32_t *)malloc(sizeof(uint32_t) * cap);
    float *size_stack = (float *)malloc(sizeof(float) * cap);
    oct_stack[0] = 0;
    size_stack[0] = root_width;
    const uint32_t self_src = order ? order[index] : index;
    while (size != 0) {
        const nbo_octant top_oct = tree[oct_stack[size - 1]];
        const float top_size = size_stack[size - 1];
        const float dx = top_oct.cog[0] - a[0], dy = top_oct.cog[1] - a[1], dz = top_oct.cog[2] - a[2];
        const float dist = sqrtf((dx * dx + dy * dy) + dz * dz);
        st->visits++;
        int is_self;
        if (flags & NBO_SELF_BY_IDENTITY)
            is_self = (top_oct.bodies == 1u && top_oct.children[0] == self_src);
        else
            is_self = (top_oct.bodies == 1u && dist < 0.000001f); /* :58 */
        if (is_self) { size -= 1; continue; }
        const float sd = top_size / dist; /* :63 */
        const int leaf_body = (flags & NBO_LEAF_IS_BODY) && top_oct.bodies == 1u;
        if (sd < theta || leaf_body) { /* :64-70 */
            const float s = (top_oct.mass * g) / ((dist * dist) * dist + e);
            acc[0] = acc[0] + (s * (dx / dist)) * dt;
            acc[1] = acc[1] + (s * (dy / dist)) * dt;
            acc[2] = acc[2] + (s * (dz / dist)) * dt;
            st->accepted++;
            size -= 1;
            continue;
        }
        size -= 1; /* :72 */
        for (uint32_t i = 0; i < 8; ++i) { /* :75-87 */
            const uint32_t child_ix = top_oct.children[i];
            if (child_ix == 0u) continue;
            if (child_ix >= n_nodes) { st->bad_index++; continue; } /* D2 can index past the tree */
            if (!(flags & NBO_CHECKED_STACK) && size >= 64u) {
                st->overflowed++;
                goto out;
            }
            if (size == cap) {
                cap *= 2;
                oct_stack = (uint32_t *)realloc(oct_stack, sizeof(uint32_t) * cap);
                size_stack = (float *)realloc(size_stack, sizeof(float) * cap);
            }
            size_stack[size] = top_size / 2.0f;
            oct_stack[size] = child_ix;
            size += 1;
            if (size > hw) hw = size;
        }
    }
out:
    if (hw > st->high_water) st->high_water = hw;
    acc_out[0] = acc[0]; acc_out[1] = acc[1]; acc_out[2] = acc[2];
    free(oct_stack);
    free(size_stack);
}

/* One Barnes-Hut step as TreeSim::encode runs it (tree.rs:262-353):
 *   build the tree on src (old positions)            -> tree, *n_nodes, *root_width
 *   reorder src into DFS/Morton order                -> sorted_src (n x 10), order (n)
 *   walk + integrate every body of sorted_src        -> dst (n x 10, sorted order)
 * stats (5 x u64): visits, accepted, high_water, overflowed, bad_index.
 * Returns 0, or -1 if the tree build failed (cap / depth). */
int nbo_tree_step_f32(const float *src, uint32_t n, float g, float e, float dt, float theta,
                      uint32_t flags, nbo_octant *tree, uint64_t cap, uint32_t max_depth,
                      int64_t *n_nodes, float *root_width, float *sorted_src, uint32_t *order,
                      float *dst, uint64_t *stats) {
    float rw = 2.0f;
    const int64_t nodes = nbo_tree_build(src, n, tree, cap, max_depth, &rw);
    if (n_nodes) *n_nodes = nodes;
    if (root_width) *root_width = rw;
    if (nodes < 0) return -1;
    nbo_tree_dfs_order(tree, n, order);
    nbo_gather_particles(src, order, n, sorted_src);

    walk_stats total;
    memset(&total, 0, sizeof total);
#pragma omp parallel
    {
        walk_stats st;
        memset(&st, 0, sizeof st);
#pragma omp for schedule(dynamic, 64)
        for (int64_t ii = 0; ii < (int64_t)n; ++ii) {
            const uint32_t i = (uint32_t)ii;
            const float *p = sorted_src + (size_t)i * 10;
            float v[3], a[3], acc[3];
            /* tree.wgsl:105-106 */
            for (int c = 0; c < 3; ++c) v[c] = p[3 + c] + (p[6 + c] * dt) / 2.0f;
            for (int c = 0; c < 3; ++c) a[c] = p[c] + v[c] * dt;
            if (n >= 2)
                walk_one(tree, (uint64_t)nodes, rw, theta, g, e, dt, a, i, order, flags, acc, &st);
            else
                acc[0] = acc[1] = acc[2] = 0.0f;
            float *o = dst + (size_t)i * 10;
            for (int c = 0; c < 3; ++c) {
                o[c] = a[c];
                o[3 + c] = v[c] + (acc[c] * dt) / 2.0f; /* :108 */
                o[6 + c] = acc[c];
            }
            o[9] = p[9];
        }
#pragma omp critical
        {
            total.visits += st.visits;
            total.accepted += st.accepted;
            if (st.high_water > total.high_water) total.high_water = st.high_water;
            total.overflowed += st.overflowed;
            total.bad_index += st.bad_index;
        }
    }
    if (stats) {
        stats[0] = total.visits;
        stats[1] = total.accepted;
        stats[2] = total.high_water;
        stats[3] = total.overflowed;
        stats[4] = total.bad_index;
    }
    return 0;
}

/* The walk + integrator of nbo_tree_step_f32 for SOME bodies of an already built tree: the sorted
 * positions idx[0..m) of sorted_src (n x 10, DFS order; order[k] = source index of sorted body k).
 * out: m x 10 rows in the order of idx.  Lets a test compare a few hundred bodies of a
 * million-body step with the per-thread walk of tree.wgsl:41-111 in seconds. */
int nbo_tree_walk_indices(const float *sorted_src, uint32_t n, const nbo_octant *tree, uint64_t n_nodes,
                          float root_width, const uint32_t *order, float g, float e, float dt,
                          float theta, uint32_t flags, const uint32_t *idx, uint32_t m, float *out,
                          uint64_t *stats) {
    walk_stats total;
    memset(&total, 0, sizeof total);
#pragma omp parallel
    {
        walk_stats st;
        memset(&st, 0, sizeof st);
#pragma omp for schedule(dynamic, 16)
        for (int64_t kk = 0; kk < (int64_t)m; ++kk) {
            const uint32_t i = idx[kk];
            const float *p = sorted_src + (size_t)i * 10;
            float v[3], a[3], acc[3];
            for (int c = 0; c < 3; ++c) v[c] = p[3 + c] + (p[6 + c] * dt) / 2.0f; /* tree.wgsl:105 */
            for (int c = 0; c < 3; ++c) a[c] = p[c] + v[c] * dt;                  /* :106 */
            if (n >= 2 && i < n)
                walk_one(tree, n_nodes, root_width, theta, g, e, dt, a, i, order, flags, acc, &st);
            else
                acc[0] = acc[1] = acc[2] = 0.0f;
            float *o = out + (size_t)kk * 10;
            for (int c = 0; c < 3; ++c) {
                o[c] = a[c];
                o[3 + c] = v[c] + (acc[c] * dt) / 2.0f; /* :108 */
                o[6 + c] = acc[c];
            }
            o[9] = p[9];
        }
#pragma omp critical
        {
            total.visits += st.visits;
            total.accepted += st.accepted;
            if (st.high_water > total.high_water) total.high_water = st.high_water;
            total.overflowed += st.overflowed;
            total.bad_index += st.bad_index;
        }
    }
    if (stats) {
        stats[0] = total.visits;
        stats[1] = total.accepted;
        stats[2] = total.high_water;
        stats[3] = total.overflowed;
        stats[4] = total.bad_index;
    }
    return 0;
}
