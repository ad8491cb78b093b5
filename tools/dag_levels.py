#!/usr/bin/env python3
"""Level structure of the generated UR5e dynamics DAG (tools/gen_ur5e_dynamics.py): op count, critical-path depth, ops per
level, and the makespan of a greedy list schedule on P processing elements (lanes of a lane group, or wavefronts) with a
cost of C slots for every operand that crosses elements. This is the feasibility bound behind DESIGN.md's discussion of a
lane-group-per-env Robot-Reach kernel: FP64 ops have no DPP operand form on gfx950, so a cross-lane operand costs two
v_mov_b32_dpp (= one FP64 issue slot) at least.

Run:  python tools/dag_levels.py
"""
from __future__ import annotations

import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.setrecursionlimit(10000)
import gen_ur5e_dynamics as g  # noqa: E402


def collect(outputs):
    needed, stack = set(), [v.id for v in outputs if not v.is_const]
    while stack:
        k = stack.pop()
        if k in needed:
            continue
        needed.add(k)
        op, a, b = g.G.nodes[k]
        if op in ("add", "sub", "mul"):
            stack += [a, b]
        elif op in ("mulc", "addc"):
            stack.append(a)
    return needed


def preds(k):
    op, a, b = g.G.nodes[k]
    if op in ("add", "sub", "mul"):
        return [a, b]
    if op in ("mulc", "addc"):
        return [a]
    return []


def levels(needed):
    lev = {}
    for k in sorted(needed):
        op = g.G.nodes[k][0]
        lev[k] = 0 if op == "sym" else 1 + max(lev[p] for p in preds(k))
    return lev


def list_schedule(needed, lev, P, C):
    """greedy: ops in (level, id) order; an op goes to the element where it can start earliest (operands produced on
    another element arrive C slots after they finish; a slot = one FP64 issue); ties to the least loaded element."""
    ops = [k for k in sorted(needed, key=lambda k: (lev[k], k)) if g.G.nodes[k][0] != "sym"]
    # priority: longest path to a sink first
    succ = {k: [] for k in needed}
    for k in needed:
        for p in preds(k):
            succ[p].append(k)
    tail = {}
    for k in sorted(needed, reverse=True):
        tail[k] = 1 + max((tail[s] for s in succ[k]), default=0)
    ready_t, where, busy = {}, {}, [0] * P
    for k in needed:
        if g.G.nodes[k][0] == "sym":
            ready_t[k], where[k] = 0, -1  # inputs are replicated on every element
    done, remaining = set(k for k in needed if g.G.nodes[k][0] == "sym"), set(ops)
    moves = 0
    while remaining:
        cand = [k for k in remaining if all(p in done for p in preds(k))]
        cand.sort(key=lambda k: -tail[k])
        k = cand[0]
        best = None
        for e in range(P):
            t = busy[e]
            xfer = 0
            for p in preds(k):
                if where[p] not in (-1, e):
                    t = max(t, ready_t[p] + C)
                    xfer += 1
                else:
                    t = max(t, ready_t[p])
            # the receiving element also spends C issue slots per moved operand
            t_end = t + 1 + C * xfer
            if best is None or t_end < best[0]:
                best = (t_end, e, xfer)
        t_end, e, xfer = best
        busy[e], ready_t[k], where[k] = t_end, t_end, e
        moves += xfer
        done.add(k)
        remaining.discard(k)
    return max(busy), moves


def main():
    links, uncomp = g.make_links([g.GRIPPER])
    tau, M = g.build(links, uncomp)
    outs = {"bias": tau, "M": [M[i][j] for i in range(6) for j in range(i + 1)], "M+bias": tau + [M[i][j] for i in range(6) for j in range(i + 1)]}
    for name, o in outs.items():
        need = collect(o)
        lev = levels(need)
        nops = sum(1 for k in need if g.G.nodes[k][0] != "sym")
        depth = max(lev.values())
        hist = [0] * (depth + 1)
        for k in need:
            if g.G.nodes[k][0] != "sym":
                hist[lev[k]] += 1
        print(f"{name}: {nops} ops, critical path {depth} ops; ops per level 1..{depth}: {hist[1:]}")
        for P in (1, 2, 4, 8):
            for C in (0, 1):
                ms, mv = list_schedule(need, lev, P, C)
                print(f"   list schedule on {P} elements, cross-element operand cost {C}: makespan {ms} slots, {mv} moved operands")


if __name__ == "__main__":
    main()
