#!/usr/bin/env python3
"""Distil the *values* of the reference's robot/task configuration into compact fixtures.

Run in the build container only (needs /root/reference).  The outputs are DATA, not source:

  qm_door_amd/data/aliengo_z1.urdf     links/inertials/joints/limits only (no visuals, no gazebo)
  qm_door_amd/data/task.info           every key the hot path reads, canonical order, no comments
  qm_door_amd/data/reference.info
  qm_door_amd/data/gait.info
  qm_door_amd/data/wbc_gains.info      defaults of qm_wbc/cfg/wbcWigeht.cfg:7-47 in INFO form

Sources (all under /root/reference):
  qm_description/urdf/quadruped_manipulator/robot.urdf
  qm_controllers/config/{task,reference,gait}.info
  qm_wbc/cfg/wbcWigeht.cfg

The product's own loaders (qm_door_amd/csrc/host) read either these fixtures or the original files.
"""
import os
import re
import sys
import xml.etree.ElementTree as ET

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "qm_door_amd", "data")


# ----------------------------------------------------------------------------- INFO (Boost property tree) subset
def parse_info(text):
    """Returns nested list of (key, value_or_None, children)."""
    toks = []
    for line in text.splitlines():
        line = line.split(";")[0]
        line = line.split("//")[0]
        toks += re.findall(r'"[^"]*"|\{|\}|[^\s{}]+', line + " \n") + ["\n"]
    pos = 0

    def block():
        nonlocal pos
        items = []
        while pos < len(toks):
            t = toks[pos]
            if t == "\n":
                pos += 1
                continue
            if t == "}":
                pos += 1
                return items
            key = t
            pos += 1
            val = None
            if pos < len(toks) and toks[pos] not in ("\n", "{", "}"):
                val = toks[pos].strip('"')
                pos += 1
            while pos < len(toks) and toks[pos] == "\n":
                pos += 1
            children = []
            if pos < len(toks) and toks[pos] == "{":
                pos += 1
                children = block()
            items.append((key, val, children))
        return items

    return block()


def dump_info(items, indent=0):
    out = []
    pad = "  " * indent
    for key, val, children in items:
        if children:
            out.append(f"{pad}{key}" + (f" {val}" if val is not None else ""))
            out.append(pad + "{")
            out += dump_info(children, indent + 1)
            out.append(pad + "}")
        else:
            out.append(f"{pad}{key} {val if val is not None else ''}".rstrip())
    return out


def keep(items, wanted):
    return [it for it in items if it[0] in wanted]


def distil_info(src, dst, wanted=None):
    items = parse_info(open(src).read())
    if wanted is not None:
        items = keep(items, wanted)
    with open(dst, "w") as f:
        f.write("\n".join(dump_info(items)) + "\n")


# ----------------------------------------------------------------------------- URDF
def distil_urdf(src, dst):
    root = ET.parse(src).getroot()
    out = ET.Element("robot", {"name": root.get("name", "robot")})
    for link in root.findall("link"):
        l = ET.SubElement(out, "link", {"name": link.get("name")})
        inert = link.find("inertial")
        if inert is not None:
            i = ET.SubElement(l, "inertial")
            o = inert.find("origin")
            ET.SubElement(i, "origin", {"xyz": (o.get("xyz") if o is not None else "0 0 0"),
                                        "rpy": (o.get("rpy", "0 0 0") if o is not None else "0 0 0")})
            ET.SubElement(i, "mass", {"value": inert.find("mass").get("value")})
            ET.SubElement(i, "inertia", dict(inert.find("inertia").attrib))
    for joint in root.findall("joint"):
        j = ET.SubElement(out, "joint", {"name": joint.get("name"), "type": joint.get("type")})
        o = joint.find("origin")
        ET.SubElement(j, "origin", {"xyz": o.get("xyz", "0 0 0"), "rpy": o.get("rpy", "0 0 0")})
        ET.SubElement(j, "parent", {"link": joint.find("parent").get("link")})
        ET.SubElement(j, "child", {"link": joint.find("child").get("link")})
        ax = joint.find("axis")
        if ax is not None:
            ET.SubElement(j, "axis", {"xyz": ax.get("xyz")})
        lim = joint.find("limit")
        if lim is not None:
            ET.SubElement(j, "limit", dict(lim.attrib))
    ET.indent(out)
    ET.ElementTree(out).write(dst, xml_declaration=True, encoding="utf-8")


def distil_wbc_gains(src, dst):
    rows = re.findall(r'gen\.add\("(\w+)",\s*double_t,\s*0,\s*"[^"]*",\s*([-\d.eE+]+)', open(src).read())
    with open(dst, "w") as f:
        f.write("wbc_gains\n{\n")
        for name, default in rows:
            f.write(f"  {name} {default}\n")
        f.write("}\n")


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    os.makedirs(OUT, exist_ok=True)
    cfg = os.path.join(REF, "qm_controllers", "config")
    distil_info(os.path.join(cfg, "task.info"), os.path.join(OUT, "task.info"),
                wanted={"centroidalModelType", "model_settings", "swing_trajectory_config", "sqp", "mpc",
                        "initialState", "Q", "R", "endEffector", "finalEndEffector",
                        "frictionConeSoftConstraint", "jointPositionLimits", "jointVelocityLimits",
                        "frictionConeTask", "ddp", "rollout"})
    distil_info(os.path.join(cfg, "reference.info"), os.path.join(OUT, "reference.info"))
    distil_info(os.path.join(cfg, "gait.info"), os.path.join(OUT, "gait.info"))
    distil_urdf(os.path.join(REF, "qm_description", "urdf", "quadruped_manipulator", "robot.urdf"),
                os.path.join(OUT, "aliengo_z1.urdf"))
    distil_wbc_gains(os.path.join(REF, "qm_wbc", "cfg", "wbcWigeht.cfg"), os.path.join(OUT, "wbc_gains.info"))
    print("wrote fixtures to", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
