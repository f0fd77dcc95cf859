// walk_model.c -- CPU model of the wave-level union traversal of walk_kernel (nb_tree.hip), used to
// size design choices before spending GPU time: for groups of G Morton-consecutive bodies it
// replays the sibling-group walk with per-body acceptance tests (size^2 < theta^2 r^2) and reports,
// per tree depth, the per-body visits, the cells the group evaluates (the union), the lane
// utilisation, and the histogram of active-lane counts.  Builder tool only (not shipped, not a
// test); the tree is a plain Morton octree on uniform random points, which has the same statistics
// as the reference's tree on uniform_init data.
//
//   gcc -O2 -fopenmp -o /tmp/walk_model tools/walk_model.c -lm && /tmp/walk_model 1048576 0.5
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z, m; } f4;
typedef struct { f4 cogm; uint32_t first, count, self_pos; float ssize2; int depth; } Rec;

static uint64_t rng_state = 88172645463325252ull;
static double urand(void) {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}
static uint64_t spread(uint64_t v) {
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
typedef struct { uint64_t key; f4 p; } KP;
static int cmp_kp(const void *a, const void *b) {
    const uint64_t x = ((const KP *)a)->key, y = ((const KP *)b)->key;
    return x < y ? -1 : x > y;
}

static KP *kp;
static Rec *rec;
static uint32_t n_nodes;

// breadth-first build so that the children of a cell are consecutive (as in the product)
typedef struct { uint32_t lo, hi, id; int depth; } Job;
static void build(uint32_t n) {
    Job *q = malloc(sizeof(Job) * 4 * (size_t)n);
    uint32_t head = 0, tail = 0;
    q[tail++] = (Job){0, n, 0, 0};
    n_nodes = 1;
    while (head < tail) {
        const Job j = q[head++];
        Rec *r = &rec[j.id];
        r->depth = j.depth;
        if (j.hi - j.lo == 1) {
            r->cogm = kp[j.lo].p; r->first = 0; r->count = 0; r->self_pos = j.lo; r->ssize2 = -1.f;
            continue;
        }
        double sx = 0, sy = 0, sz = 0, sm = 0;
        for (uint32_t k = j.lo; k < j.hi; ++k) {
            sx += kp[k].p.x * kp[k].p.m; sy += kp[k].p.y * kp[k].p.m; sz += kp[k].p.z * kp[k].p.m; sm += kp[k].p.m;
        }
        r->cogm = (f4){(float)(sx / sm), (float)(sy / sm), (float)(sz / sm), (float)sm};
        const float w = 2.0f / (float)(1u << j.depth);
        r->ssize2 = w * w; r->self_pos = ~0u; r->first = n_nodes; r->count = 0;
        const int shift = 3 * (20 - j.depth);
        uint32_t k = j.lo;
        while (k < j.hi) {
            const uint64_t d = (kp[k].key >> shift) & 7;
            uint32_t e = k;
            while (e < j.hi && ((kp[e].key >> shift) & 7) == d) ++e;
            q[tail++] = (Job){k, e, n_nodes++, j.depth + 1};
            r->count++;
            k = e;
        }
    }
    free(q);
}

#define MAXD 24
typedef struct {
    double visits[MAXD], cells[MAXD], accepts[MAXD], take0[MAXD], full[MAXD], groups, gcells;
    double pop_hist[65];      // wave-cells by popcount of the group mask
    double iter4[MAXD];       // ceil(gcnt / (64/G)) per popped group (scheme B iterations)
    double sub16[MAXD];       // cells whose mask fits one aligned 16-lane quarter (G = 64 only)
    double quarters[MAXD];    // sum over cells of the number of aligned 16-lane quarters the mask touches
    double octs[MAXD];        // ... of aligned 8-lane octets
} Stats;

typedef struct { uint32_t first, count; uint64_t mask; } Ent;

static void walk_group(uint32_t lo, int G, float theta2, Stats *st) {
    static _Thread_local Ent stack[512];
    int sp = 0;
    const uint64_t all = G == 64 ? ~0ull : ((1ull << G) - 1);
    stack[sp++] = (Ent){0, 1, all};
    const int per = 64 / G;
    while (sp > 0) {
        const Ent t = stack[--sp];
        st->groups += 1; st->gcells += t.count;
        for (uint32_t c = 0; c < t.count; ++c) {
            const Rec *r = &rec[t.first + c];
            uint64_t far = 0, other = 0;
            for (int l = 0; l < G; ++l) {
                if (!((t.mask >> l) & 1)) continue;
                const f4 p = kp[lo + l].p;
                const float dx = r->cogm.x - p.x, dy = r->cogm.y - p.y, dz = r->cogm.z - p.z;
                const float r2 = dz * dz + (dy * dy + dx * dx);
                if (r->ssize2 < theta2 * r2) far |= 1ull << l;
                if (r->self_pos != lo + l) other |= 1ull << l;
            }
            const uint64_t take = t.mask & far & other, open = t.mask & ~far;
            const int d = r->depth, pc = __builtin_popcountll(t.mask);
            st->visits[d] += pc; st->cells[d] += 1; st->accepts[d] += __builtin_popcountll(take);
            if (!take) st->take0[d] += 1;
            if (t.mask == all) st->full[d] += 1;
            st->pop_hist[pc] += 1;
            int nq = 0, no = 0;
            for (int qd = 0; qd < 4; ++qd) if ((t.mask >> (16 * qd)) & 0xffff) ++nq;
            for (int o = 0; o < 8; ++o) if ((t.mask >> (8 * o)) & 0xff) ++no;
            st->quarters[d] += nq; st->octs[d] += no;
            if (nq == 1) st->sub16[d] += 1;
            if (open) stack[sp++] = (Ent){r->first, r->count, open};
        }
        st->iter4[rec[t.first].depth] += (t.count + per - 1) / per;
    }
}


// ---- scheme C: cells across the 64 lanes, the group's G bodies in scalars -------------------------
// LIFO stack of (cell, G-bit visit mask); a batch pops up to 64 cells, every lane tests its cell
// against each of the G bodies, the children of opened cells are pushed (siblings contiguous).
typedef struct { uint32_t id; uint32_t mask; } CEnt;
typedef struct { double batches, cells, pairs, visits, hw, maxhw, pairs_any, tpairs, thalves, lo_only, hi_only, w16, w32; } CStats;
static int g_fifo = 0;  // 1: pop from the OLD end of the list (breadth-first), 0: from the new end (the product)
static void walk_group_c2(uint32_t lo, int nb, int G, float theta2, int batch, CStats *st) {
    static _Thread_local CEnt stack[1 << 16];
    int sp = 0, hw = 0;
    stack[sp++] = (CEnt){0, (nb == 32 ? 0xffffffffu : ((1u << nb) - 1))};
    while (sp > 0) {
        const int c = sp < batch ? sp : batch;
        CEnt cur[64];
        if (g_fifo) {
            memcpy(cur, stack, sizeof(CEnt) * c);
            memmove(stack, stack + c, sizeof(CEnt) * (sp - c));
        } else {
            memcpy(cur, stack + sp - c, sizeof(CEnt) * c);
        }
        sp -= c;
        st->batches += 1; st->cells += c; st->pairs += (double)batch * G;
        if (c <= 16) st->w16 += 1; else if (c <= 32) st->w32 += 1;
        uint32_t any = 0;
        for (int l = 0; l < c; ++l) {
            const Rec *r = &rec[cur[l].id];
            uint32_t open = 0;
            any |= cur[l].mask;
            if (G == 8) {   // touched body pairs / halves of the mask (packing granularities)
                const uint32_t m = cur[l].mask;
                st->tpairs += ((m & 3) != 0) + ((m & 12) != 0) + ((m & 48) != 0) + ((m & 192) != 0);
                st->thalves += ((m & 15) != 0) + ((m & 240) != 0);
                if ((m & 240) == 0) st->lo_only += 1;
                if ((m & 15) == 0) st->hi_only += 1;
            }
            for (int b = 0; b < nb; ++b) {
                if (!((cur[l].mask >> b) & 1)) continue;
                const f4 p = kp[lo + b].p;
                const float dx = r->cogm.x - p.x, dy = r->cogm.y - p.y, dz = r->cogm.z - p.z;
                const float r2 = dz * dz + (dy * dy + dx * dx);
                st->visits += 1;
                if (!(r->ssize2 < theta2 * r2)) open |= 1u << b;
            }
            if (open)
                for (uint32_t k = 0; k < r->count; ++k) stack[sp++] = (CEnt){r->first + k, open};
        }
        st->pairs_any += (double)batch * __builtin_popcount(any);
        if (sp > hw) hw = sp;
    }
    st->hw += hw;
    if (hw > st->maxhw) st->maxhw = hw;
}

static void walk_group_c(uint32_t lo, int G, float theta2, int batch, CStats *st) { walk_group_c2(lo, G, G, theta2, batch, st); }

// ---- scheme D: TWO groups per wave, 32 lanes each (own stack, own 8 bodies as per-lane operands) --------
// One iteration = one batch instruction stream for the wave: each half pops up to 32 of its group's cells.
// refill = 1: a half whose group is finished takes the next group from a queue (dynamic); 0: the wave ends
// when both of its two groups have.  Returns the iterations (batches) spent; `cells` the lanes that held a cell.
typedef struct { CEnt st[1 << 14]; int sp; uint32_t lo; int live; } Half;
static void half_start(Half *h, uint32_t lo) { h->sp = 0; h->st[h->sp++] = (CEnt){0, 0xffu}; h->lo = lo; h->live = 1; }
static int half_step(Half *h, float theta2, int width) {   // one batch of this half: returns the cells popped
    const int c = h->sp < width ? h->sp : width;
    CEnt cur[64];
    memcpy(cur, h->st + h->sp - c, sizeof(CEnt) * c);
    h->sp -= c;
    for (int l = 0; l < c; ++l) {
        const Rec *r = &rec[cur[l].id];
        uint32_t open = 0;
        for (int b = 0; b < 8; ++b) {
            if (!((cur[l].mask >> b) & 1)) continue;
            const f4 p = kp[h->lo + b].p;
            const float dx = r->cogm.x - p.x, dy = r->cogm.y - p.y, dz = r->cogm.z - p.z;
            const float r2 = dz * dz + (dy * dy + dx * dx);
            if (!(r->ssize2 < theta2 * r2)) open |= 1u << b;
        }
        if (open)
            for (uint32_t k = 0; k < r->count; ++k) h->st[h->sp++] = (CEnt){r->first + k, open};
    }
    if (h->sp == 0) h->live = 0;
    return c;
}
static void scheme_d(uint32_t first_body, uint32_t n_groups, float theta2, int refill, double *iters, double *cells) {
    static _Thread_local Half ha, hb;
    uint32_t next = 0;
    *iters = 0; *cells = 0;
    while (next < n_groups || ha.live || hb.live) {
        if (!ha.live && !hb.live && !refill) {   // a fresh wave: two groups
            if (next < n_groups) half_start(&ha, first_body + 8 * next++);
            if (next < n_groups) half_start(&hb, first_body + 8 * next++);
        }
        if (refill) {
            if (!ha.live && next < n_groups) half_start(&ha, first_body + 8 * next++);
            if (!hb.live && next < n_groups) half_start(&hb, first_body + 8 * next++);
        }
        if (!ha.live && !hb.live) break;
        *iters += 1;
        if (ha.live) *cells += half_step(&ha, theta2, 32);
        if (hb.live) *cells += half_step(&hb, theta2, 32);
    }
}

int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 1u << 20;
    const float theta = argc > 2 ? (float)atof(argv[2]) : 0.5f;
    const int sample = argc > 3 ? atoi(argv[3]) : 512;   // groups of 64 sampled
    kp = malloc(sizeof(KP) * (size_t)n);
    rec = malloc(sizeof(Rec) * 4 * (size_t)n);
    for (uint32_t i = 0; i < n; ++i) {
        const float x = (float)(urand() * 2 - 1), y = (float)(urand() * 2 - 1), z = (float)(urand() * 2 - 1);
        kp[i].p = (f4){x, y, z, 1.f};
        const uint64_t qx = (uint64_t)((x + 1.0) * 0.5 * 2097152.0), qy = (uint64_t)((y + 1.0) * 0.5 * 2097152.0),
                       qz = (uint64_t)((z + 1.0) * 0.5 * 2097152.0);
        kp[i].key = spread(qx) | (spread(qy) << 1) | (spread(qz) << 2);
    }
    qsort(kp, n, sizeof(KP), cmp_kp);
    build(n);
    printf("n %u nodes %u (%.3f N) theta %.2f\n", n, n_nodes, (double)n_nodes / n, theta);
    const int Gs[4] = {64, 32, 16, 8};
    for (int gi = 0; gi < 4; ++gi) {
        const int G = Gs[gi];
        Stats tot; memset(&tot, 0, sizeof tot);
        const uint32_t n64 = n / 64, stride = n64 / sample ? n64 / sample : 1;
#pragma omp parallel
        {
            Stats st; memset(&st, 0, sizeof st);
#pragma omp for schedule(dynamic, 4)
            for (uint32_t w = 0; w < n64; w += stride)
                for (int s = 0; s < 64 / G; ++s) walk_group(w * 64 + s * G, G, theta * theta, &st);
#pragma omp critical
            {
                double *a = (double *)&tot, *b = (double *)&st;
                for (size_t k = 0; k < sizeof(Stats) / sizeof(double); ++k) a[k] += b[k];
            }
        }
        double V = 0, Cc = 0, A = 0, T0 = 0, F = 0, I4 = 0;
        const double ngroups = (double)((n64 + stride - 1) / stride) * (64 / G);
        printf("\nG = %d bodies per group (%g groups sampled)\n", G, ngroups);
        printf(" depth  visits/body  cells/group  util   take0%%  fullmask%%  quarters/cell octets/cell iterB/group\n");
        for (int d = 0; d < MAXD; ++d) {
            if (!tot.cells[d]) continue;
            printf(" %5d  %10.1f  %10.1f  %5.3f  %5.1f  %5.1f  %5.2f  %5.2f  %8.1f\n", d, tot.visits[d] / (ngroups * G), tot.cells[d] / ngroups,
                   tot.visits[d] / (tot.cells[d] * G), 100 * tot.take0[d] / tot.cells[d], 100 * tot.full[d] / tot.cells[d],
                   tot.quarters[d] / tot.cells[d], tot.octs[d] / tot.cells[d], tot.iter4[d] / ngroups);
            V += tot.visits[d]; Cc += tot.cells[d]; A += tot.accepts[d]; T0 += tot.take0[d]; F += tot.full[d]; I4 += tot.iter4[d];
        }
        printf(" total  visits/body %.1f accepts/body %.1f cells/group %.1f util %.3f take0 %.1f%% full %.1f%% pops/group %.1f cells/pop %.2f iterB/group %.1f (x%d lanes-of-cells)\n",
               V / (ngroups * G), A / (ngroups * G), Cc / ngroups, V / (Cc * G), 100 * T0 / Cc, 100 * F / Cc, tot.groups / ngroups,
               tot.gcells / tot.groups, I4 / ngroups, 64 / G);
        if (G == 64) {
            double q = 0, o = 0;
            for (int d = 0; d < MAXD; ++d) { q += tot.quarters[d]; o += tot.octs[d]; }
            printf(" G=64: sum of touched 16-lane quarters per group %.1f (util if only touched quarters cost: %.3f); touched octets %.1f (util %.3f)\n",
                   q / ngroups, V / (q * 16), o / ngroups, V / (o * 8));
            printf(" popcount histogram (wave-cells %%): ");
            double cum = 0;
            for (int p = 1; p <= 64; ++p) { cum += tot.pop_hist[p]; if (p % 8 == 0) { printf("<=%d: %.1f  ", p, 100 * cum / Cc); } }
            printf("\n");
        }
    }

    for (g_fifo = 0; g_fifo < 2; ++g_fifo) {
    printf("\nscheme C (cells across lanes, G bodies in scalars, %s, batch 64)\n", g_fifo ? "FIFO queue (breadth-first)" : "LIFO stack");
    const int Gc[4] = {4, 8, 16, 32};
    for (int gi = 0; gi < 4; ++gi) {
        const int G = Gc[gi];
        CStats tot; memset(&tot, 0, sizeof tot);
        const uint32_t n64 = n / 64, stride = n64 / sample ? n64 / sample : 1;
#pragma omp parallel
        {
            CStats st; memset(&st, 0, sizeof st);
#pragma omp for schedule(dynamic, 4)
            for (uint32_t w = 0; w < n64; w += stride)
                for (int s = 0; s < 64 / G; ++s) walk_group_c(w * 64 + s * G, G, theta * theta, 64, &st);
#pragma omp critical
            {
                tot.batches += st.batches; tot.cells += st.cells; tot.pairs += st.pairs; tot.visits += st.visits;
                tot.hw += st.hw; tot.pairs_any += st.pairs_any; if (st.maxhw > tot.maxhw) tot.maxhw = st.maxhw;
                tot.tpairs += st.tpairs; tot.thalves += st.thalves; tot.lo_only += st.lo_only; tot.hi_only += st.hi_only;
                tot.w16 += st.w16; tot.w32 += st.w32;
            }
        }
        const double ngroups = (double)((n64 + stride - 1) / stride) * (64 / G);
        printf(" G %2d: batches/group %.1f  cells/group %.1f  fill %.3f  pair-instr per 64 bodies %.0f  util %.3f  (skipping bodies absent from a batch: %.0f, util %.3f)  stack high water mean %.0f max %.0f\n",
               G, tot.batches / ngroups, tot.cells / ngroups, tot.cells / (tot.batches * 64), tot.pairs / 64 / ngroups * (64 / G),
               tot.visits / tot.pairs, tot.pairs_any / 64 / ngroups * (64 / G), tot.visits / tot.pairs_any, tot.hw / ngroups, tot.maxhw);
        if (G == 8)
            printf("       per popped cell: bodies that visit it %.2f of 8; touched pairs %.2f of 4; touched halves %.2f of 2; lo-half only %.1f %% hi-half only %.1f %%\n",
                   tot.visits / tot.cells, tot.tpairs / tot.cells, tot.thalves / tot.cells, 100 * tot.lo_only / tot.cells, 100 * tot.hi_only / tot.cells);
        if (G == 8)
            printf("       batches of <= 16 cells: %.1f %%, of 17..32 cells: %.1f %%\n", 100 * tot.w16 / tot.batches, 100 * tot.w32 / tot.batches);
    }

    }
    g_fifo = 0;
    printf("\nscheme C with cell-aligned groups (a group never straddles a cell of > 8 bodies), G = 8 slots\n");
    {
        // greedy: walk sorted bodies; group = longest run of <= 8 bodies sharing the key prefix of the
        // smallest cell that holds the first body and has <= 8 bodies
        CStats tot; memset(&tot, 0, sizeof tot);
        double ngroups = 0, nbodies = 0;
        uint32_t i = 0;
        // sample a contiguous range
        const uint32_t start = n / 3, stop = start + 65536;
        i = start;
        while (i < stop) {
            int best = 1;
            for (int d = 1; d <= 20; ++d) {   // shallowest cell containing body i with <= 8 bodies
                const int shift = 3 * (21 - d);
                const uint64_t pre = kp[i].key >> shift;
                uint32_t a = i, b2 = i;
                while (a > 0 && (kp[a - 1].key >> shift) == pre) --a;
                while (b2 + 1 < n && (kp[b2 + 1].key >> shift) == pre) ++b2;
                if (b2 - a + 1 <= 8) { best = (int)(b2 - i + 1); break; }
            }
            walk_group_c2(i, best, 8, theta * theta, 64, &tot);
            ngroups += 1; nbodies += best;
            i += best;
        }
        printf(" groups %.0f mean size %.2f; pair-instr per 64 bodies %.0f (batches per 64 bodies %.1f)\n", ngroups, nbodies / ngroups,
               tot.pairs / 64 / nbodies * 64, tot.batches / nbodies * 64);
    }
    printf("\nscheme D: two groups per wave, 32 lanes each (iterations = batch instruction streams per 64 bodies)\n");
    for (int refill = 0; refill < 2; ++refill) {
        double it = 0, ce = 0;
        const uint32_t groups = 8192;
        scheme_d(n / 3, groups, theta * theta, refill, &it, &ce);
        printf(" %s: batches per 64 bodies %.1f (scheme C, G = 8: 8 x batches/group above), lane fill %.3f\n",
               refill ? "a finished half takes the next group" : "the wave ends when both groups have", it / groups * 8, ce / (it * 64));
    }
    return 0;
}
