"""Independent numpy / pure-Python restatement of the reference's two steps.

TEST INFRASTRUCTURE ONLY (tests/ imports it; the product never does).
PARITY UNPINNED: the reference has no tests or golden vectors; the pin is that
THIS restatement and the C one (nbody_oracle*.c), written separately from the
same reference lines, agree bit for bit in fp32 (IEEE add/mul/div/sqrt are
correctly rounded in both, and both keep the WGSL operation order).

Follows (reference crate paths):
  src/sims/shaders/naive.wgsl:23-69        all-pairs + integrator
  src/sims/tree.rs:417-602                 bound, BFS build, DFS reorder
  src/sims/shaders/tree.wgsl:41-111        tree walk + integrator
"""
from __future__ import annotations

from collections import deque

import numpy as np

F = np.float32


def naive_step(state: np.ndarray, g, e, dt) -> np.ndarray:
    """naive.wgsl main+getAcc in binary32; j sequential (ascending), i vectorised."""
    s = np.ascontiguousarray(state, dtype=F)
    n = s.shape[0]
    g, e, dt, two = F(g), F(e), F(dt), F(2.0)
    pos, vel, acc0, mass = s[:, 0:3], s[:, 3:6], s[:, 6:9], s[:, 9]
    vh = vel + (acc0 * dt) / two                      # naive.wgsl:63
    xn = pos + vh * dt                                # :64
    acc = np.zeros((n, 3), dtype=F)                   # :24
    ids = np.arange(n)
    with np.errstate(all="ignore"):
        for j in range(n):                            # :26-46
            d = pos[j][None, :] - xn                  # bPos - aPos  (OLD x_j, NEW x_i)
            r = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])  # distance
            sc = (mass[j] * g) / ((r * r) * r + e)    # :39, left-associative
            f = (sc[:, None] * (d / r[:, None])) * dt # * normalize(...) ; * dt (:41)
            keep = ids != j                           # :30-32
            acc[keep] = acc[keep] + f[keep]
    vn = vh + (acc * dt) / two                        # :66
    out = np.empty_like(s)
    out[:, 0:3], out[:, 3:6], out[:, 6:9], out[:, 9] = xn, vn, acc, mass  # :68
    return out


def naive_step_f64(state: np.ndarray, g, e, dt) -> np.ndarray:
    """Same formula in binary64, fully vectorised (n x n); small n only."""
    s = np.asarray(state, dtype=np.float64)
    n = s.shape[0]
    g, e, dt = (float(F(x)) for x in (g, e, dt))
    pos, vel, acc0, mass = s[:, 0:3], s[:, 3:6], s[:, 6:9], s[:, 9]
    vh = vel + (acc0 * dt) / 2.0
    xn = pos + vh * dt
    d = pos[None, :, :] - xn[:, None, :]              # [i, j, :]
    r = np.sqrt((d ** 2).sum(-1))
    with np.errstate(all="ignore"):
        sc = (mass[None, :] * g) / (r ** 3 + e) / r
    sc[np.arange(n), np.arange(n)] = 0.0
    d[np.arange(n), np.arange(n), :] = 0.0
    acc = (sc[:, :, None] * d).sum(1) * dt
    out = np.empty_like(s)
    out[:, 0:3], out[:, 3:6], out[:, 6:9], out[:, 9] = xn, vh + (acc * dt) / 2.0, acc, mass
    return out


# --------------------------------------------------------------------------- tree

def tree_build(state: np.ndarray):
    """tree.rs:417-546 with Python lists; returns (nodes, root_width).

    nodes: list of dict(cog[3], mass, bodies, children[8]) in allocation order.
    """
    s = np.ascontiguousarray(state, dtype=F)
    n = s.shape[0]
    bound = F(1.0)
    if n:
        bound = max(F(1.0), F(np.abs(s[:, 0:3]).max()))   # :424-446
    nodes = [None]                                         # root_ix = write(default), :461
    q = deque()
    q.append((np.zeros(3, dtype=F), F(bound * F(2.0)), 0, list(range(n))))
    while q:                                               # :473
        center, width, node_ix, plist = q.popleft()
        cog = np.zeros(3, dtype=F)
        mass = F(0.0)
        lists = [[] for _ in range(8)]
        for ix in plist:                                   # :486-501
            p = s[ix]
            cog = cog + p[0:3] * p[9]
            mass = F(mass + p[9])
            o = int(p[0] > center[0]) | (int(p[1] > center[1]) << 1) | (int(p[2] > center[2]) << 2)
            lists[o].append(ix)
        with np.errstate(all="ignore"):
            cog = cog / mass                               # :503-505
        children = [0] * 8
        for o in range(8):                                 # :507-541
            if not lists[o]:
                continue
            child_ix = len(nodes)
            nodes.append(None)
            children[o] = child_ix
            if len(lists[o]) == 1:
                lp = s[lists[o][0]]
                nodes[child_ix] = dict(cog=lp[0:3].copy(), mass=F(lp[9]), bodies=1,
                                       children=[lists[o][0]] + [0] * 7)
            else:
                sh = [F((((o >> a) & 1) * 2 - 1)) * width / F(4.0) for a in range(3)]  # :556-562
                q.append((np.array([center[a] + sh[a] for a in range(3)], dtype=F),
                          F(width / F(2.0)), child_ix, lists[o]))
        nodes[node_ix] = dict(cog=cog, mass=mass, bodies=len(plist), children=children)  # :543
    return nodes, float(F(bound * F(2.0)))


def tree_dfs_order(nodes, n):
    """tree.rs:564-602 (recursive in the reference)."""
    out = []

    def rec(o):
        if o["bodies"] == 1:
            out.append(o["children"][0])
            return
        for c in o["children"]:
            if c != 0:
                rec(nodes[c])

    if n >= 2:
        rec(nodes[0])
    elif n == 1:
        out.append(0)
    return np.array(out, dtype=np.uint32)


def tree_walk(nodes, root_width, theta, g, e, dt, a, self_src, flags):
    """tree.wgsl:41-90 for one body at new position a (binary32)."""
    g, e, dt, theta = F(g), F(e), F(dt), F(theta)
    acc = np.zeros(3, dtype=F)
    stack = [(0, F(root_width))]
    with np.errstate(all="ignore"):
        while stack:
            ix, size = stack.pop()
            o = nodes[ix]
            d = np.asarray(o["cog"], dtype=F) - a
            dist = F(np.sqrt(F(F(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])))
            if flags & 1:
                is_self = o["bodies"] == 1 and o["children"][0] == self_src
            else:
                is_self = o["bodies"] == 1 and dist < F(0.000001)
            if is_self:
                continue
            sd = F(size / dist)
            if sd < theta or ((flags & 2) and o["bodies"] == 1):
                sc = F(F(o["mass"] * g) / F(F(F(dist * dist) * dist) + e))
                acc = acc + (sc * (d / dist)) * dt
                continue
            for c in o["children"]:          # pushed 0..7, popped 7..0
                if c != 0 and c < len(nodes):
                    stack.append((c, F(size / F(2.0))))
    return acc


def tree_step(state: np.ndarray, g, e, dt, theta, flags=7):
    s = np.ascontiguousarray(state, dtype=F)
    n = s.shape[0]
    nodes, rw = tree_build(s)
    order = tree_dfs_order(nodes, n)
    srt = s[order]
    dtf, two = F(dt), F(2.0)
    out = np.empty_like(srt)
    for i in range(n):
        p = srt[i]
        v = p[3:6] + (p[6:9] * dtf) / two
        a = p[0:3] + v * dtf
        acc = tree_walk(nodes, rw, theta, g, e, dt, a, int(order[i]), flags) if n >= 2 \
            else np.zeros(3, dtype=F)
        out[i, 0:3], out[i, 3:6], out[i, 6:9], out[i, 9] = a, v + (acc * dtf) / two, acc, p[9]
    return dict(dst=out, sorted_src=srt, order=order, nodes=nodes, root_width=rw)
