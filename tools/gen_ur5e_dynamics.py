#!/usr/bin/env python3
"""Generates mujoco_sim_amd/csrc/mjs_ur5e_dyn_gen.h: straight-line float64 code for the UR5e
joint-space inertia matrix M(q) (composite-rigid-body algorithm) and the smooth bias forces
(recursive Newton-Euler: Coriolis/centrifugal of the whole arm + the weight of the
un-gravity-compensated end-effector payload), specialised to the kinematic tree in
include/mjs_scene_spec.h.

Method: the classical link-local-frame recursions (Luh-Walker-Paul / Featherstone) are executed
once on a tiny expression graph with constant folding (x*0, x*1, x+0, +-) and hash-consed common
sub-expressions, so every structural zero of the UR5e (axis-aligned joints, axis-aligned link
offsets, 90-degree frame changes, principal-axis inertias) disappears at generation time. The
emitted code is what the kernel runs per physics substep (one env per lane); it replaces the
generic world-frame CRBA/RNE of the first kernel version at ~2.5x fewer FP64 instructions.
Results equal the oracle's (oracle/om_engine.c) up to float rounding; tests/test_gpu_parity.py.

Run:  python tools/gen_ur5e_dynamics.py        (rewrites the header in place)
"""
from __future__ import annotations

import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
SPEC = (ROOT / "include" / "mjs_scene_spec.h").read_text()
OUT = ROOT / "mujoco_sim_amd" / "csrc" / "mjs_ur5e_dyn_gen.h"


# --------------------------------------------------------------------------- spec parsing
def spec_array(name):
    m = re.search(name + r"(?:\[[^\]]*\])*\s*=\s*\{(.*?)\};", SPEC, re.S)
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    return [float(x) for x in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", body)]


def spec_scalar(name):
    m = re.search(name + r"\s*=\s*(-?\d+\.?\d*(?:e-?\d+)?)\s*;", SPEC)
    return float(m.group(1))


# ------------------------------------------------------------------- tiny expression graph
class Graph:
    def __init__(self):
        self.nodes = []  # (op, a, b) ; a/b are node ids or python floats
        self.cache = {}

    def node(self, op, a, b=None):
        key = (op, a, b)
        if key not in self.cache:
            self.cache[key] = len(self.nodes)
            self.nodes.append(key)
        return self.cache[key]


G = Graph()


class V:
    """value = const  or  sign * node"""

    __slots__ = ("c", "sign", "id")

    def __init__(self, c=None, sign=1, id=None):
        self.c, self.sign, self.id = c, sign, id

    @property
    def is_const(self):
        return self.c is not None

    def __neg__(self):
        return V(-self.c) if self.is_const else V(None, -self.sign, self.id)

    def __add__(self, o):
        o = lift(o)
        if self.is_const and o.is_const:
            return V(self.c + o.c)
        if self.is_const:
            return o + self
        if o.is_const:
            if o.c == 0.0:
                return self
            # node + const
            if self.sign > 0:
                return V(None, 1, G.node("addc", self.id, o.c))
            return V(None, -1, G.node("addc", self.id, -o.c))
        if self.id == o.id:
            if self.sign != o.sign:
                return V(0.0)
            return V(None, self.sign, G.node("mulc", self.id, 2.0))
        if self.sign > 0 and o.sign > 0:
            a, b = sorted((self.id, o.id))
            return V(None, 1, G.node("add", a, b))
        if self.sign < 0 and o.sign < 0:
            a, b = sorted((self.id, o.id))
            return V(None, -1, G.node("add", a, b))
        if self.sign > 0:
            return V(None, 1, G.node("sub", self.id, o.id))
        return V(None, 1, G.node("sub", o.id, self.id))

    __radd__ = __add__

    def __sub__(self, o):
        return self + (-lift(o))

    def __rsub__(self, o):
        return lift(o) + (-self)

    def __mul__(self, o):
        o = lift(o)
        if self.is_const and o.is_const:
            return V(self.c * o.c)
        if self.is_const:
            return o * self
        if o.is_const:
            if o.c == 0.0:
                return V(0.0)
            if o.c == 1.0:
                return self
            if o.c == -1.0:
                return -self
            sgn = self.sign * (1 if o.c > 0 else -1)
            return V(None, sgn, G.node("mulc", self.id, abs(o.c)))
        a, b = sorted((self.id, o.id))
        return V(None, self.sign * o.sign, G.node("mul", a, b))

    __rmul__ = __mul__

    def rcp(self):
        """reciprocal (the kernels' Newton-refined v_rcp_f64: MJS_GEN_RCP); 1 / (-x) = -(1 / x)"""
        if self.is_const:
            return V(1.0 / self.c)
        return V(None, self.sign, G.node("rcp", self.id))


def lift(x):
    return x if isinstance(x, V) else V(float(x))


def sym(name):
    return V(None, 1, G.node("sym", name))


def vec(*xs):
    return [lift(x) for x in xs]


def vadd(a, b):
    return [x + y for x, y in zip(a, b)]


def vsub(a, b):
    return [x - y for x, y in zip(a, b)]


def vscale(s, a):
    return [lift(s) * x for x in a]


def cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def matvec(M, v):
    return [M[i][0] * v[0] + M[i][1] * v[1] + M[i][2] * v[2] for i in range(3)]


def mattvec(M, v):
    return [M[0][i] * v[0] + M[1][i] * v[1] + M[2][i] * v[2] for i in range(3)]


def matmul(A, B):
    return [[A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j] for j in range(3)] for i in range(3)]


def transpose(A):
    return [[A[j][i] for j in range(3)] for i in range(3)]


def quat_to_mat_exact(q):
    """rotation of an un-normalised quaternion without a square root (entries of the UR5e frame
    changes are exactly 0 / +-1)"""
    w, x, y, z = q
    n = w * w + x * x + y * y + z * z
    R = [[(w * w + x * x - y * y - z * z) / n, 2 * (x * y - w * z) / n, 2 * (x * z + w * y) / n],
         [2 * (x * y + w * z) / n, (w * w - x * x + y * y - z * z) / n, 2 * (y * z - w * x) / n],
         [2 * (x * z - w * y) / n, 2 * (y * z + w * x) / n, (w * w - x * x - y * y + z * z) / n]]
    return [[0.0 if abs(v) < 1e-15 else (round(v) if abs(v - round(v)) < 1e-15 else v) for v in row] for row in R]


# ----------------------------------------------------------------------------- the robot
NJ = 6
body_pos = [spec_array("MJS_UR_BODY_POS")[3 * b:3 * b + 3] for b in range(7)]
body_quat = [spec_array("MJS_UR_BODY_QUAT")[4 * b:4 * b + 4] for b in range(7)]
body_mass = spec_array("MJS_UR_BODY_MASS")
body_ipos = [spec_array("MJS_UR_BODY_IPOS")[3 * b:3 * b + 3] for b in range(7)]
body_iquat = [spec_array("MJS_UR_BODY_IQUAT")[4 * b:4 * b + 4] for b in range(7)]
body_diag = [spec_array("MJS_UR_BODY_DIAGINERTIA")[3 * b:3 * b + 3] for b in range(7)]
jnt_axis = [spec_array("MJS_UR_JNT_AXIS")[3 * j:3 * j + 3] for j in range(NJ)]
flange_pos = spec_array("MJS_UR_FLANGE_POS")
flange_quat = spec_array("MJS_UR_FLANGE_QUAT")
pl_mass = spec_scalar("MJS_G2F85_MASS")
pl_ipos = spec_array("MJS_G2F85_IPOS")
pl_diag = spec_array("MJS_G2F85_DIAGINERTIA")
gravity_z = spec_scalar("MJS_GRAVITY_Z")


def num_matmul(A, B):
    return [[sum(A[i][k] * B[k][j] for k in range(3)) for j in range(3)] for i in range(3)]


def num_matvec(A, v):
    return [sum(A[i][k] * v[k] for k in range(3)) for i in range(3)]


def link_inertial(b):
    """(mass, com, 3x3 inertia about the com) of link body b in its own frame"""
    Ri = quat_to_mat_exact(body_iquat[b])
    D = [[body_diag[b][0], 0, 0], [0, body_diag[b][1], 0], [0, 0, body_diag[b][2]]]
    I = num_matmul(num_matmul(Ri, D), [[Ri[j][i] for j in range(3)] for i in range(3)])
    return body_mass[b], list(body_ipos[b]), I


def merge(parts):
    """combine rigid parts [(m, com, Icom)] into one (m, com, Icom)"""
    m = sum(p[0] for p in parts)
    com = [sum(p[0] * p[1][k] for p in parts) / m for k in range(3)]
    I = [[0.0] * 3 for _ in range(3)]
    for pm, pc, pI in parts:
        d = [pc[k] - com[k] for k in range(3)]
        d2 = sum(x * x for x in d)
        for i in range(3):
            for j in range(3):
                I[i][j] += pI[i][j] + pm * ((d2 if i == j else 0.0) - d[i] * d[j])
    return m, com, I


def quat_to_mat(q):
    w, x, y, z = q
    n = (w * w + x * x + y * y + z * z) ** 0.5
    w, x, y, z = w / n, x / n, y / n, z / n
    return [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]


def transpose_num(A):
    return [[A[j][i] for j in range(3)] for i in range(3)]


Rf = quat_to_mat_exact(flange_quat)


def flange_part(mass, ipos, iquat, diag):
    """rigid part given in the flange frame -> (m, com, Icom) in the wrist_3 frame"""
    R = num_matmul(Rf, quat_to_mat(iquat) if iquat is not None else [[1, 0, 0], [0, 1, 0], [0, 0, 1]])
    com = [flange_pos[k] + num_matvec(Rf, ipos)[k] for k in range(3)]
    D = [[diag[0], 0, 0], [0, diag[1], 0], [0, 0, diag[2]]]
    return (mass, com, num_matmul(num_matmul(R, D), transpose_num(R)))


GRIPPER = flange_part(pl_mass, pl_ipos, None, pl_diag)  # lumped 2F-85 (deviation D-1)


def wrist_camera_part():
    """box + sphere of default density, concentric (entities/camera.py:78-88), robot_push_button.py:90-96"""
    import math

    bx, by, bz = spec_array("MJS_CAM_BOX_HALF")
    rs = spec_scalar("MJS_CAM_SPHERE_RADIUS")
    rho = spec_scalar("MJS_GEOM_DENSITY")
    mb, ms = rho * 8 * bx * by * bz, rho * 4.0 / 3.0 * math.pi * rs ** 3
    Is = 0.4 * ms * rs * rs
    diag = [mb * (by * by + bz * bz) / 3 + Is, mb * (bx * bx + bz * bz) / 3 + Is, mb * (bx * bx + by * by) / 3 + Is]
    return flange_part(mb + ms, spec_array("MJS_WCAM_POS"), spec_array("MJS_WCAM_QUAT"), diag)


def cylinder_eef_part():
    """CylinderEEF (entities/eef/cylinder.py:23-33): uniform cylinder, axis = flange z, mass given explicitly"""
    r, h, m = spec_scalar("MJS_CYL_RADIUS"), spec_scalar("MJS_CYL_HALFLEN"), spec_scalar("MJS_CYL_MASS")
    ixx = m * (3 * r * r + (2 * h) ** 2) / 12
    return flange_part(m, [0.0, 0.0, spec_scalar("MJS_CYL_POS_Z")], None, [ixx, ixx, 0.5 * m * r * r])


def make_links(extra_parts):
    """per joint j: dict(C, r, axis, m, c, I); extra_parts = un-gravity-compensated rigid parts on link 6.
    Returns (links, (mass, com) of the un-compensated parts together)"""
    links = []
    for j in range(NJ):
        b = j + 1
        m, c, I = link_inertial(b)
        if j == NJ - 1:
            m, c, I = merge([(m, c, I)] + list(extra_parts))
        links.append(dict(C=quat_to_mat_exact(body_quat[b]), r=body_pos[b], axis=jnt_axis[j], m=m, c=c, I=I))
    um = sum(p[0] for p in extra_parts)
    ucom = [sum(p[0] * p[1][k] for p in extra_parts) / um for k in range(3)] if um > 0 else [0.0, 0.0, 0.0]
    return links, (um, ucom)


def clean(x):
    return 0.0 if abs(x) < 1e-18 else x


def joint_rot(axis, c, s):
    if axis == [0.0, 0.0, 1.0]:
        return [[c, -s, lift(0)], [s, c, lift(0)], [lift(0), lift(0), lift(1)]]
    if axis == [0.0, 1.0, 0.0]:
        return [[c, lift(0), s], [lift(0), lift(1), lift(0)], [-s, lift(0), c]]
    raise ValueError(axis)


def build(links, uncomp):
    pl_mass, payload_com = uncomp
    cs = [sym(f"c[{j}]") for j in range(NJ)]
    sn = [sym(f"s[{j}]") for j in range(NJ)]
    qd = [sym(f"qd[{j}]") for j in range(NJ)]
    E = []  # child->parent rotation (symbolic)
    for j, L in enumerate(links):
        C = [[lift(v) for v in row] for row in L["C"]]
        E.append(matmul(C, joint_rot(L["axis"], cs[j], sn[j])))
    z = [vec(*L["axis"]) for L in links]
    r = [vec(*[clean(v) for v in L["r"]]) for L in links]
    com = [vec(*[clean(v) for v in L["c"]]) for L in links]
    Icom = [[[lift(clean(v)) for v in row] for row in L["I"]] for L in links]

    # ---- RNE forward: angular velocity/acceleration and origin acceleration in link frames
    w_prev, al_prev, a_prev = vec(0, 0, 0), vec(0, 0, 0), vec(0, 0, 0)
    g_prev = vec(0, 0, gravity_z)  # base frame is Rz(180deg) of the world: z unchanged
    w, al, a, g = [], [], [], []
    for j in range(NJ):
        wp = mattvec(E[j], w_prev)
        w_j = vadd(wp, vscale(qd[j], z[j]))
        al_j = vadd(mattvec(E[j], al_prev), vscale(qd[j], cross(wp, z[j])))
        a_j = mattvec(E[j], vadd(a_prev, vadd(cross(al_prev, r[j]), cross(w_prev, cross(w_prev, r[j])))))
        g_j = mattvec(E[j], g_prev)
        w.append(w_j); al.append(al_j); a.append(a_j); g.append(g_j)
        w_prev, al_prev, a_prev, g_prev = w_j, al_j, a_j, g_j
    # ---- RNE backward
    f_next, n_next = None, None
    tau = [None] * NJ
    for j in reversed(range(NJ)):
        ac = vadd(a[j], vadd(cross(al[j], com[j]), cross(w[j], cross(w[j], com[j]))))
        F = vscale(links[j]["m"], ac)
        N = vadd(matvec(Icom[j], al[j]), cross(w[j], matvec(Icom[j], w[j])))
        f = F
        n = vadd(N, cross(com[j], F))
        if j == NJ - 1:
            W = vscale(pl_mass, g[j])  # weight of the un-compensated payload, in the link frame
            f = vsub(f, W)
            n = vsub(n, cross(vec(*[clean(v) for v in payload_com]), W))
        else:
            fc = matvec(E[j + 1], f_next)
            f = vadd(f, fc)
            n = vadd(n, vadd(matvec(E[j + 1], n_next), cross(r[j + 1], fc)))
        tau[j] = dot(z[j], n)
        f_next, n_next = f, n

    # ---- CRBA: composite (m, h, I about the link origin) in link frames
    def origin_inertia(m, c, Ic):
        c2 = dot(c, c)
        return [[Ic[i][k] + lift(m) * ((c2 if i == k else lift(0)) - c[i] * c[k]) for k in range(3)] for i in range(3)]

    cm = [lift(L["m"]) for L in links]
    ch = [vscale(L["m"], com[j]) for j, L in enumerate(links)]
    cI = [origin_inertia(L["m"], com[j], Icom[j]) for j, L in enumerate(links)]
    for j in reversed(range(1, NJ)):
        # move composite j into frame j-1 about origin j-1
        Ej = E[j]
        h_rot = matvec(Ej, ch[j])
        I_rot = matmul(matmul(Ej, cI[j]), transpose(Ej))
        rj = r[j]
        rr, rh = dot(rj, rj), dot(rj, h_rot)
        for i in range(3):
            for k in range(3):
                extra = cm[j] * ((rr if i == k else lift(0)) - rj[i] * rj[k]) + ((rh * 2.0) if i == k else lift(0)) - rj[i] * h_rot[k] - h_rot[i] * rj[k]
                cI[j - 1][i][k] = cI[j - 1][i][k] + I_rot[i][k] + extra
        ch[j - 1] = vadd(ch[j - 1], vadd(h_rot, vscale(cm[j], rj)))
        cm[j - 1] = cm[j - 1] + cm[j]
    M = [[None] * NJ for _ in range(NJ)]
    for i in range(NJ):
        n = matvec(cI[i], z[i])
        f = cross(z[i], ch[i])
        M[i][i] = dot(z[i], n)
        for j in reversed(range(i)):
            f_up = matvec(E[j + 1], f)
            n_up = vadd(matvec(E[j + 1], n), cross(r[j + 1], f_up))
            M[i][j] = dot(z[j], n_up)
            f, n = f_up, n_up
    return tau, M


def emit(outputs, schedule=None):
    """straight-line code for the given (name, value) outputs; returns (lines, op counts). schedule = (latency, barrier_every) emits a
    pinned list schedule (below); None = creation order, which the compiler reschedules freely."""
    needed, stack = set(), [v.id for _, v in outputs if not v.is_const]
    while stack:
        k = stack.pop()
        if k in needed:
            continue
        needed.add(k)
        op, a, b = G.nodes[k]
        if op in ("add", "sub", "mul"):
            stack += [a, b]
        elif op in ("mulc", "addc", "rcp"):
            stack.append(a)
    lines, names = [], {}
    counts = {"add": 0, "sub": 0, "mul": 0, "mulc": 0, "addc": 0, "rcp": 0}
    order = sorted(needed)
    # VERDICT r3 item 4 (profiles/r04_g_*): a LIST SCHEDULE of the DAG - critical path first, a dependent op at least `lat` slots after
    # its operand - PINNED with a scheduling barrier after every statement, instead of the creation order that the compiler is free
    # to reschedule. Same-session A/Bs at 4096 envs: the pinned M / bias blocks of Robot-Reach 35.38 -> 34.91 us (lat 4; lat 1 / 3 / 6
    # / 9: 35.08 / 35.06 / 34.94 / 35.87; unpinned list order: no change), Button-Push 71.2 -> 72.2 us, Planar-Push 1750 -> 1821 us.
    # What won in the end is the COMBINED block (factor_inverse below) left to the compiler: 34.52 us with nothing pinned, so no
    # variant is emitted pinned; GEN_ORDER=list GEN_LATENCY=k GEN_BARRIER=k (all blocks) / GEN_MW_PIN=1 keep the experiment reproducible.
    import os
    if os.environ.get("GEN_ORDER") == "list":
        schedule = (int(os.environ.get("GEN_LATENCY", "2")), int(os.environ.get("GEN_BARRIER", "0")))
    barrier = 0
    if schedule is not None:
        lat, barrier = schedule
        preds = {}
        for k in needed:
            op, a, b = G.nodes[k]
            preds[k] = [x for x in ((a, b) if op in ("add", "sub", "mul") else (a,) if op in ("mulc", "addc", "rcp") else ()) if isinstance(x, int) and x in needed]
        succs = {k: [] for k in needed}
        for k, ps in preds.items():
            for q in ps:
                succs[q].append(k)
        rcp_lat = int(os.environ.get("GEN_RCP_LATENCY", "10"))  # a refined reciprocal is a chain of five dependent instructions
        lat_of = {k: (rcp_lat if G.nodes[k][0] == "rcp" else lat) for k in needed}
        height = {}
        for k in sorted(needed, reverse=True):  # creation order is topological
            height[k] = lat_of[k] + max((height[x] for x in succs[k]), default=0)
        done_at, order, slot = {}, [], 0
        remaining = {k: len(set(preds[k])) for k in needed}
        ready = [k for k in needed if remaining[k] == 0]
        while ready:
            ok = [k for k in ready if all(done_at[q] + lat_of[q] <= slot for q in preds[k] if G.nodes[q][0] != "sym")]
            pick = max(ok or ready, key=lambda k: (height[k], -k))
            ready.remove(pick)
            order.append(pick)
            done_at[pick] = slot
            if G.nodes[pick][0] != "sym":
                slot += 5 if G.nodes[pick][0] == "rcp" else 1  # issue slots of the op itself
            for x in set(succs[pick]):
                remaining[x] -= 1
                if remaining[x] == 0:
                    ready.append(x)
    emitted = 0
    for k in order:
        op, a, b = G.nodes[k]
        if op == "sym":
            names[k] = a
            continue
        counts[op] += 1
        names[k] = f"t{k}"
        if op == "add":
            e = f"{names[a]} + {names[b]}"
        elif op == "sub":
            e = f"{names[a]} - {names[b]}"
        elif op == "mul":
            e = f"{names[a]} * {names[b]}"
        elif op == "mulc":
            e = f"{b!r} * {names[a]}"
        elif op == "rcp":
            e = f"MJS_GEN_RCP({names[a]})"
        else:
            e = f"{names[a]} + {b!r}"
        lines.append(f"  const double t{k} = {e};")
        emitted += 1
        if barrier and emitted % barrier == 0:
            lines.append("  MJS_SCHED_PIN;")
    for name, v in outputs:
        if v.is_const:
            lines.append(f"  {name} = {v.c!r};")
        else:
            lines.append(f"  {name} = {'-' if v.sign < 0 else ''}{names[v.id]};")
    return lines, counts


def evaluate(vals, env):
    """numeric evaluation of graph values with symbol bindings env (name -> float)"""
    memo = {}

    def ev(k):
        if k in memo:
            return memo[k]
        op, a, b = G.nodes[k]
        if op == "sym":
            r = env[a]
        elif op == "add":
            r = ev(a) + ev(b)
        elif op == "sub":
            r = ev(a) - ev(b)
        elif op == "mul":
            r = ev(a) * ev(b)
        elif op == "mulc":
            r = ev(a) * b
        elif op == "rcp":
            r = 1.0 / ev(a)
        else:
            r = ev(a) + b
        memo[k] = r
        return r

    return [v.c if v.is_const else v.sign * ev(v.id) for v in vals]


def model_constants(M, links, body_com_local):
    """mj_setConst quantities at qpos0 = 0: dof_invweight0 = diag(M^-1), meaninertia = mean(diag M),
    body_invweight0 (translation, rotation) of a body rigidly attached to link 6 with COM body_com_local"""
    import numpy as np

    env = {f"c[{j}]": 1.0 for j in range(NJ)}
    env.update({f"s[{j}]": 0.0 for j in range(NJ)})
    env.update({f"qd[{j}]": 0.0 for j in range(NJ)})
    flat = evaluate([M[i][j] for i in range(NJ) for j in range(i + 1)], env)
    A = np.zeros((NJ, NJ))
    k = 0
    for i in range(NJ):
        for j in range(i + 1):
            A[i, j] = A[j, i] = flat[k]
            k += 1
    A += np.eye(NJ) * spec_scalar("MJS_UR_ARMATURE")
    Ainv = np.linalg.inv(A)
    # forward kinematics at q = 0 (base frame; the base's own Rz(180deg) does not change the traces)
    R, p = np.eye(3), np.zeros(3)
    axes, anchors = [], []
    for L in links:
        p = p + R @ np.array(L["r"])
        R = R @ np.array(L["C"], dtype=float)
        axes.append(R @ np.array(L["axis"]))
        anchors.append(p.copy())
    P = p + R @ np.array(body_com_local)
    Jt = np.stack([np.cross(axes[j], P - anchors[j]) for j in range(NJ)], axis=1)
    Jr = np.stack(axes, axis=1)
    tran = float(np.trace(Jt @ Ainv @ Jt.T) / 3)
    rot = float(np.trace(Jr @ Ainv @ Jr.T) / 3)
    return np.diag(Ainv), float(np.trace(A) / NJ), tran, rot


def link_body_invweights(M, links):
    """body_invweight0 (translation, rotation) of the arm's OWN link bodies at qpos0 = 0 (mj_setConst: Jacobian at the
    body's own COM body_ipos; only the joints up to the body move it): index 0 = base (static: 0), b = 1..6 = link b.
    Contact rows of a collision geom on link b use the translational value (diagApprox of mj_makeImpedance)."""
    import numpy as np

    env = {f"c[{j}]": 1.0 for j in range(NJ)}
    env.update({f"s[{j}]": 0.0 for j in range(NJ)})
    env.update({f"qd[{j}]": 0.0 for j in range(NJ)})
    flat = evaluate([M[i][j] for i in range(NJ) for j in range(i + 1)], env)
    A = np.zeros((NJ, NJ))
    k = 0
    for i in range(NJ):
        for j in range(i + 1):
            A[i, j] = A[j, i] = flat[k]
            k += 1
    A += np.eye(NJ) * spec_scalar("MJS_UR_ARMATURE")
    Ainv = np.linalg.inv(A)
    R, p = np.eye(3), np.zeros(3)
    axes, anchors, out = [], [], [(0.0, 0.0)]
    for b, L in enumerate(links, start=1):
        p = p + R @ np.array(L["r"])
        R = R @ np.array(L["C"], dtype=float)
        axes.append(R @ np.array(L["axis"]))
        anchors.append(p.copy())
        P = p + R @ np.array(body_ipos[b])
        Jt = np.zeros((3, NJ))
        Jr = np.zeros((3, NJ))
        for j in range(b):
            Jt[:, j] = np.cross(axes[j], P - anchors[j])
            Jr[:, j] = axes[j]
        out.append((float(np.trace(Jt @ Ainv @ Jt.T) / 3), float(np.trace(Jr @ Ainv @ Jr.T) / 3)))
    return out


def factor_inverse(M):
    """The role-0 block of the Robot-Reach substep as ONE expression graph: A = M + dd (dd[j] = armature + dt * kd_j of the unclamped
    actuators, an input), A = U D U^T eliminated from the wrist end, V = U^-1 - literally rr::factor_system + rr::invert_unit_upper
    (mjs_reach.h) on graph values, so that the list schedule can run the pivot chain next to the tail of the CRBA.
    Returns (W[j][i] = V(i, j) for i < j, Dinv[j])."""
    A = [[M[j][i] for i in range(j + 1)] for j in range(NJ)]
    for j in range(NJ):
        A[j][j] = A[j][j] + sym(f"dd[{j}]")
    Dg, Dinv = [None] * NJ, [None] * NJ
    for j in range(NJ - 1, -1, -1):
        d = A[j][j]
        for m in range(j + 1, NJ):
            d = d - A[m][j] * A[m][j] * Dg[m]
        Dg[j] = d
        Dinv[j] = d.rcp()
        for i in range(j):
            sv = A[j][i]
            for m in range(j + 1, NJ):
                sv = sv - A[m][i] * A[m][j] * Dg[m]
            A[j][i] = sv * Dinv[j]
    W = [[None] * NJ for _ in range(NJ)]
    for j in range(NJ - 1, 0, -1):
        for i in range(j - 1, -1, -1):
            sv = A[j][i]
            for k in range(i + 1, j):
                sv = sv + A[k][i] * W[j][k]
            W[j][i] = -sv
    return W, Dinv


def emit_variant(prefix, links, uncomp, body_com_local, what, schedule=None, with_factor=False):
    tau, M = build(links, uncomp)
    out_bias = [(f"bias[{j}]", tau[j]) for j in range(NJ)]
    out_M = [(f"M[{i * (i + 1) // 2 + j}]", M[i][j]) for i in range(NJ) for j in range(i + 1)]
    lb, cb = emit(out_bias, schedule)
    lm, cm = emit(out_M, schedule)
    nb, nm = sum(cb.values()), sum(cm.values())
    invw, meaninertia, tran, rot = model_constants(M, links, body_com_local)
    _, _, w3tran, w3rot = model_constants(M, links, body_ipos[NJ])  # the wrist_3 body itself, at its own COM
    linkw = link_body_invweights(M, links)
    assert abs(linkw[NJ][0] - w3tran) < 1e-12 and abs(linkw[NJ][1] - w3rot) < 1e-12
    up = prefix.upper()
    text = f"""
// ---- variant "{prefix}": {what}
// mj_setConst quantities of this model at qpos0 = 0 (armature included)
constexpr double {up}_DOF_INVWEIGHT0[6] = {{{", ".join(repr(float(x)) for x in invw)}}};
constexpr double {up}_MEANINERTIA = {meaninertia!r};
constexpr double {up}_EEF_BODY_INVWEIGHT0[2] = {{{tran!r}, {rot!r}}};  // translation, rotation: contact rows of the EEF body
constexpr double {up}_WRIST3_BODY_INVWEIGHT0[2] = {{{w3tran!r}, {w3rot!r}}};  // contact rows of geoms on the wrist_3 body
constexpr double {up}_LINK_BODY_INVWEIGHT0[7] = {{{", ".join(repr(t) for t, _ in linkw)}}};  // translation, link bodies 0 (base) .. 6 (wrist_3): contact rows of the arm's collision geoms

MJS_DEV void {prefix}_M_gen(const double* c, const double* s, double* M) {{  // {nm} ops before FMA fusion
""" + "\n".join(lm) + f"""
}}

MJS_DEV void {prefix}_bias_gen(const double* c, const double* s, const double* qd, double* bias) {{  // {nb} ops before FMA fusion
""" + "\n".join(lb) + """
}
"""
    if with_factor:
        W, Dinv = factor_inverse(M)
        out_f = [(f"W[{j * (j - 1) // 2 + i}]", W[j][i]) for j in range(1, NJ) for i in range(j)] + [(f"Dinv[{j}]", Dinv[j]) for j in range(NJ)]
        import os
        # the combined block in creation order, the compiler free to schedule it: 34.67 us per launch against 34.94 with the pinned
        # M block + rr::factor_system / invert_unit_upper, and 35.3 - 35.4 with the combined block pinned (GEN_MW_PIN=1;
        # profiles/r04_g_*): the pivot chain's reciprocals want the compiler's own latency model
        lf, cf = emit(out_f, schedule if os.environ.get("GEN_MW_PIN") else None)
        text += f"""
// M + dd -> U D U^T -> V = U^-1 as one scheduled block (rr::factor_system + rr::invert_unit_upper on the graph): W[j (j - 1) / 2 + i] =
// V(i, j) for i < j, Dinv[j] = 1 / D_j; dd[j] = joint armature + dt * kd_j for an unclamped actuator
MJS_DEV void {prefix}_MW_gen(const double* c, const double* s, const double* dd, double* W, double* Dinv) {{  // {sum(cf.values())} ops before FMA fusion
""" + "\n".join(lf) + """
}
"""
    print(f"variant {prefix}: M {nm} ops, bias {nb} ops, eef invweight0 = ({tran:.6g}, {rot:.6g})")
    return text


def main():
    header = """// mjs_ur5e_dyn_gen.h — GENERATED by tools/gen_ur5e_dynamics.py from include/mjs_scene_spec.h.
// DO NOT EDIT. Straight-line float64 code for the UR5e with a rigid end-effector lump (D-1), per scene:
//   <v>_M_gen:    M[21], lower triangle (row-major, M[i*(i+1)/2+j], j<=i) of the joint-space inertia
//                 matrix WITHOUT joint armature (role of mj_crb; reference path: Physics.step, SURVEY App. B)
//   <v>_bias_gen: bias[6], Coriolis/centrifugal forces of the whole arm minus the joint torque of the
//                 weight of the un-gravity-compensated end-effector parts, i.e. qfrc_bias - qfrc_gravcomp
//                 of MuJoCo (gravcomp=1 on the arm bodies only, entities/robots/robot.py:80-82), so that
//                 qfrc_smooth = -bias + actuator force
// inputs: c[j] = cos(q_j), s[j] = sin(q_j), qd[j] = joint velocity. M and bias share no sub-expression,
// which is what lets the kernel run them on two different wavefronts.
#pragma once
#ifndef MJS_DEV
#define MJS_DEV __device__ __forceinline__
#endif
#ifndef MJS_GEN_RCP  // the kernels' reciprocal: v_rcp_f64 + two Newton steps (rr::rcp_fast); a host build defines 1.0 / x
#define MJS_GEN_RCP(x) mjs_gen_rcp(x)
MJS_DEV double mjs_gen_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}
#endif
#ifndef MJS_SCHED_PIN  // holds a hand-ordered (list-scheduled) emission against the compiler's scheduler; a host build defines it away
#define MJS_SCHED_PIN __builtin_amdgcn_sched_barrier(0)
#endif
"""
    text = header
    links, uncomp = make_links([GRIPPER])
    text += emit_variant("ur5e", links, uncomp, GRIPPER[1], "Robot-Reach: UR5e + lumped 2F-85 gripper (robot_reach.py:90-95)", with_factor=True)
    cam = wrist_camera_part()
    links2, uncomp2 = make_links([GRIPPER, cam])
    text += emit_variant("ur5e_bp", links2, uncomp2, GRIPPER[1], "Button-Push: + wrist-camera geoms' mass at the flange (robot_push_button.py:90-96)")  # (the combined factor block loses here: 71.08 -> 71.30 us)
    cyl = cylinder_eef_part()
    links3, uncomp3 = make_links([cyl])
    text += emit_variant("ur5e_pp", links3, uncomp3, cyl[1], "Planar-Push: UR5e + CylinderEEF (robot_planar_push.py:81-87)")
    links4, uncomp4 = make_links([])
    text += emit_variant("ur5e_bare", links4, uncomp4, body_ipos[NJ], "the UR5e alone, nothing on the flange (test/test_ur_control_api.py:8-9; every body gravity-compensated, robot.py:80-82)")
    text += """
MJS_DEV void ur5e_dynamics_gen(const double* c, const double* s, const double* qd, double* M, double* bias) {
  ur5e_M_gen(c, s, M);
  ur5e_bias_gen(c, s, qd, bias);
}
"""
    OUT.write_text(text)
    print(f"wrote {OUT}")


if __name__ == "__main__":
    sys.setrecursionlimit(10000)
    main()
