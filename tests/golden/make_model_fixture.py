#!/usr/bin/env python3
"""Independent pin of the rigid-body model: tests/golden/model_independent.npz.

Everything the oracle and the kernels know about the robot reaches them through the product's URDF loader (qmgpu_load_problem) and their own
recursive / Jacobian-sum formulations.  This script shares NOTHING with either:
  * it parses qm_door_amd/data/aliengo_z1.urdf itself (xml.etree), keeps all 28 links as separate bodies (no merging of fixed children),
  * forward kinematics is the only geometry it implements; body velocities come from COMPLEX-STEP differentiation of that forward kinematics
    (exact to round-off), never from a geometric Jacobian formula,
  * the mass matrix is the Hessian of the kinetic energy T(q, v) = 1/2 sum_b (m_b |v_b|^2 + w_b^T R_b I_b R_b^T w_b) in v,
    the non-linear effects are Lagrange's equations  nle_i = sum_k (dM/dq_k v_k v)_i - 1/2 v^T (dM/dq_i) v + dV/dq_i  with the potential
    V = sum_b m_b g z_b (dM/dq by 4th-order Richardson differences of the exact M, dV/dq by complex step),
  * the centroidal momentum matrix maps v to [sum m_b v_b ; sum (c_b - c) x m_b v_b + R_b I_b R_b^T w_b].
Generalised coordinates / velocities as in SURVEY.md Appendix A: q = [p (3), yaw, pitch, roll, LF(3), LH(3), RF(3), RH(3), z1_joint_1..6],
v = dq/dt (Euler ZYX rates).  The gripper joint is held at 0 (the reference's model builder fixes it).

Run:  python tests/golden/make_model_fixture.py      (a few seconds; numpy only)
tests/test_oracle_invariants.py::test_model_against_independent_fixture compares the oracle with the result."""
import os
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
URDF = os.path.join(HERE, "..", "..", "qm_door_amd", "data", "aliengo_z1.urdf")
JOINT_ORDER = [f"{leg}_{j}" for leg in ("LF", "LH", "RF", "RH") for j in ("HAA", "HFE", "KFE")] + [f"z1_joint_{i}" for i in range(1, 7)]
G = 9.81


def parse():
    root = ET.parse(URDF).getroot()
    links = {}
    for l in root.findall("link"):
        inert = l.find("inertial")
        if inert is None:
            links[l.get("name")] = dict(m=0.0, c=np.zeros(3), I=np.zeros((3, 3)))
            continue
        i = inert.find("inertia")
        I = np.array([[float(i.get("ixx")), float(i.get("ixy")), float(i.get("ixz"))], [float(i.get("ixy")), float(i.get("iyy")), float(i.get("iyz"))],
                      [float(i.get("ixz")), float(i.get("iyz")), float(i.get("izz"))]])
        assert inert.find("origin").get("rpy").split() == ["0", "0", "0"]
        links[l.get("name")] = dict(m=float(inert.find("mass").get("value")), c=np.array(inert.find("origin").get("xyz").split(), float), I=I)
    joints = []
    for j in root.findall("joint"):
        o = j.find("origin")
        assert [float(t) for t in o.get("rpy").split()] == [0.0, 0.0, 0.0]
        ax = j.find("axis")
        joints.append(dict(name=j.get("name"), type=j.get("type"), parent=j.find("parent").get("link"), child=j.find("child").get("link"),
                           xyz=np.array(o.get("xyz").split(), float), axis=None if ax is None else np.array(ax.get("xyz").split(), float)))
    return links, joints


LINKS, JOINTS = parse()
CHILDREN = {}
for jt in JOINTS:
    CHILDREN.setdefault(jt["parent"], []).append(jt)


def rot(axis, a):
    """Rodrigues rotation about a unit axis; works for complex angles (complex step)."""
    x, y, z = axis
    K = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]], dtype=complex)
    return np.eye(3, dtype=complex) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)


def fk(q):
    """world pose (R, p) of every link for generalised coordinates q (possibly complex)."""
    q = np.asarray(q, dtype=complex)
    R0 = rot((0, 0, 1), q[3]) @ rot((0, 1, 0), q[4]) @ rot((1, 0, 0), q[5])
    pose = {"base": (R0, q[0:3].copy())}
    stack = ["base"]
    while stack:
        parent = stack.pop()
        Rp, pp = pose[parent]
        for jt in CHILDREN.get(parent, []):
            ang = q[6 + JOINT_ORDER.index(jt["name"])] if jt["name"] in JOINT_ORDER else 0.0
            Rc = Rp @ rot(jt["axis"], ang) if jt["type"] == "revolute" else Rp
            pose[jt["child"]] = (Rc, pp + Rp @ jt["xyz"])
            stack.append(jt["child"])
    return pose


H = 1e-30


def body_velocities(q, v):
    """(com, com velocity, angular velocity, world inertia) of every massive link for the velocity direction v, by complex step."""
    pose = fk(np.asarray(q, dtype=complex) + 1j * H * np.asarray(v, float))
    out = []
    for name, L in LINKS.items():
        if L["m"] == 0.0:
            continue
        R, p = pose[name]
        c = p + R @ L["c"]
        Rr = R.real
        W = (R.imag / H) @ Rr.T                     # dR/dt R^T = [w]x
        w = np.array([W[2, 1] - W[1, 2], W[0, 2] - W[2, 0], W[1, 0] - W[0, 1]]) * 0.5
        out.append((L["m"], c.real, c.imag / H, w, Rr @ L["I"] @ Rr.T))
    return out


def kinetic(q, v):
    return sum(0.5 * m * vc @ vc + 0.5 * w @ Iw @ w for m, _, vc, w, Iw in body_velocities(q, v))


def mass_matrix(q):
    n = 24
    E = np.eye(n)
    cols = [body_velocities(q, E[k]) for k in range(n)]     # velocities are linear in v: per-direction fields
    M = np.zeros((n, n))
    for i in range(n):
        for j in range(i, n):
            s = 0.0
            for bi, bj in zip(cols[i], cols[j]):
                s += bi[0] * bi[2] @ bj[2] + bi[3] @ bi[4] @ bj[3]
            M[i, j] = M[j, i] = s
    return M


def potential_gradient(q):
    g = np.zeros(24)
    for k in range(24):
        e = np.zeros(24); e[k] = 1.0
        g[k] = sum(m * G * vc[2] for m, _, vc, _, _ in body_velocities(q, e))     # dV/dq_k = sum m g dz_b/dq_k
    return g


def nonlinear_effects(q, v):
    n = 24
    dM = []
    for k in range(n):
        e = np.zeros(n); e[k] = 1.0
        d1 = (mass_matrix(q + 1e-3 * e) - mass_matrix(q - 1e-3 * e)) / 2e-3
        d2 = (mass_matrix(q + 5e-4 * e) - mass_matrix(q - 5e-4 * e)) / 1e-3
        dM.append((4.0 * d2 - d1) / 3.0)
    Mdot = sum(dM[k] * v[k] for k in range(n))
    return Mdot @ v - 0.5 * np.array([v @ dM[i] @ v for i in range(n)]) + potential_gradient(q)


def centroidal_matrix(q):
    n = 24
    A = np.zeros((6, n))
    bodies0 = body_velocities(q, np.zeros(n))
    mtot = sum(b[0] for b in bodies0)
    com = sum(b[0] * b[1] for b in bodies0) / mtot
    for k in range(n):
        e = np.zeros(n); e[k] = 1.0
        for m, c, vc, w, Iw in body_velocities(q, e):
            A[0:3, k] += m * vc
            A[3:6, k] += np.cross(c - com, m * vc) + Iw @ w
    return A, mtot, com


def main():
    rng = np.random.default_rng(20240928)
    q_nom = np.r_[0.0, 0.0, 0.4, 0.0, 0.0, 0.0, np.tile([0.0, 0.8, -1.5], 4), [0.0, 1.11, -0.69, -0.4, 0.0, 0.0]]     # reference.info default joints, comHeight
    qs = [q_nom] + [q_nom + rng.uniform(-1, 1, 24) * np.r_[np.full(3, 0.3), np.full(3, 0.4), np.full(18, 0.4)] for _ in range(2)]
    vs = [np.zeros(24)] + [rng.uniform(-1, 1, 24) * np.r_[np.full(3, 0.5), np.full(3, 0.8), np.full(18, 1.5)] for _ in range(2)]
    out = dict(q=np.array(qs), v=np.array(vs), M=[], nle=[], AG=[], com=[], feet=[], ee=[], omega_world=[], kinetic=[])
    for q, v in zip(qs, vs):
        M = mass_matrix(q)
        assert abs(0.5 * v @ M @ v - kinetic(q, v)) <= 1e-12 * max(1.0, kinetic(q, v))     # M really is the Hessian of T
        A, mtot, com = centroidal_matrix(q)
        pose = fk(q)
        out["M"].append(M); out["nle"].append(nonlinear_effects(q, v)); out["AG"].append(A); out["com"].append(com)
        out["feet"].append(np.array([pose[f"{leg}_FOOT"][1].real for leg in ("LF", "RF", "LH", "RH")]))     # contact order of the MPC
        out["ee"].append(pose["z1_end_effector"][1].real)
        base = [b for b in body_velocities(q, v)][0]
        out["omega_world"].append(base[3])                                                                  # links are visited in file order: base first
        out["kinetic"].append(kinetic(q, v))
    out = {k: np.array(val) for k, val in out.items()}
    out["total_mass"] = np.array(mtot)
    np.savez(os.path.join(HERE, "model_independent.npz"), **out)
    print("total mass", mtot, "| M[0] diag", np.diag(out["M"][0])[:6], "| nle[0][:6]", out["nle"][0][:6])


if __name__ == "__main__":
    main()
