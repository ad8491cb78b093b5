#!/usr/bin/env python3
"""Pins of the scene constants to the DATA FILES the reference holds (run in the build container, which has
/root/reference; the GPU box and the CPU tests only read the JSON this writes):

  * mujoco_sim/mjcf/walled_pointmass_arena.xml: lights, the fixed camera, the ground plane, the four wall planes, the grid
    texture colours and the wall material (attributes parsed with xml.etree, numbers only);
  * mujoco_sim/mjcf/google_language_table_blocks/<category>.xml: the mesh geom's quat and rgba;
  * .../<category>.obj: vertex count, face count, the bounding box of the vertices in the BODY frame
    (x, y, z) = (mesh x, -mesh z, mesh y) that the xml's quat (1 1 0 0) defines, and a checksum of the vertex bytes.

  * the numeric literals of the task / entity sources that include/mjs_scene_spec.h restates as [REF] constants (time steps,
    thresholds, spawn / workspace boxes, camera poses, switch and block parameters, the gripper's TCP offset and opening):
    file, line number and the numbers on that line, nothing else.

Output: tests/golden/reference_scene_data.json (data: numbers, no source text).
tests/test_oracle_known_answers.py::test_scene_constants_match_reference_data_files compares include/mjs_scene_spec.h and
include/mjs_block_hulls.h with it.
"""
import hashlib
import json
import xml.etree.ElementTree as ET
from pathlib import Path

import numpy as np

REF = Path("/root/reference/mujoco_sim/mjcf")
OUT = Path(__file__).resolve().parent / "reference_scene_data.json"


def nums(s):
    return [float(x) for x in s.split()]


def main():
    arena = ET.parse(REF / "walled_pointmass_arena.xml").getroot()
    tex = arena.find("asset/texture")
    data = {
        "source": "mujoco_sim/mjcf/walled_pointmass_arena.xml + mujoco_sim/mjcf/google_language_table_blocks/*.xml, *.obj",
        "arena": {
            "lights": [nums(l.get("pos")) for l in arena.findall("worldbody/light")],
            "camera": {"pos": nums(arena.find("worldbody/camera").get("pos")), "quat": nums(arena.find("worldbody/camera").get("quat"))},
            "grid_rgb1": nums(tex.get("rgb1")), "grid_rgb2": nums(tex.get("rgb2")),
            "decoration_rgba": nums([m for m in arena.findall("asset/material") if m.get("name") == "decoration"][0].get("rgba")),
            "planes": {g.get("name"): {"pos": nums(g.get("pos")), "size": nums(g.get("size")), "zaxis": nums(g.get("zaxis")) if g.get("zaxis") else [0.0, 0.0, 1.0]}
                       for g in arena.findall("worldbody/geom")},
        },
        "blocks": {},
    }
    for cat in ("cube", "moon", "pentagon", "star"):
        root = ET.parse(REF / "google_language_table_blocks" / f"{cat}.xml").getroot()
        geom = root.find("worldbody/body/geom")
        V, nf = [], 0
        for line in open(REF / "google_language_table_blocks" / f"{cat}.obj"):
            p = line.split()
            if p and p[0] == "v":
                V.append([float(x) for x in p[1:4]])
            elif p and p[0] == "f":
                nf += 1
        V = np.array(V)
        B = np.stack([V[:, 0], -V[:, 2], V[:, 1]], axis=1)  # quat (1 1 0 0): +90 degrees about x
        data["blocks"][cat] = {
            "geom_type": geom.get("type"), "quat": nums(geom.get("quat")), "rgba": nums(geom.get("rgba")),
            "n_vertices": int(len(V)), "n_faces": int(nf),
            "body_frame_bbox_lo": B.min(0).tolist(), "body_frame_bbox_hi": B.max(0).tolist(),
            "vertices_sha256": hashlib.sha256(np.ascontiguousarray(V, dtype=np.float64).tobytes()).hexdigest(),
        }
    return data


# ---- numeric literals of the reference's task / entity sources ------------------------------------------------------
# (constant or group of constants in include/mjs_scene_spec.h, file under mujoco_sim/, regex whose group 1 holds the numbers)
LITERALS = [
    ("MJS_PM_PHYSICS_DT", "environments/tasks/point_reach.py", r"^PHYSICS_TIMESTEP = ([\d.]+)"),
    ("MJS_PM_CONTROL_DT", "environments/tasks/point_reach.py", r"^CONTROL_TIMESTEP = ([\d.]+)"),
    ("MJS_PM_MAX_CONTROL_STEPS", "environments/tasks/point_reach.py", r"^MAX_CONTROL_STEPS_PER_EPISODE = (\d+)"),
    ("MJS_PM_GOAL_THRESHOLD", "environments/tasks/point_reach.py", r"^GOAL_DISTANCE_THRESHOLD = ([\d.]+)"),
    ("MJS_PM_MAX_STEP_SIZE", "environments/tasks/point_reach.py", r"^MAX_STEP_SIZE = ([\d.]+)"),
    ("MJS_PM_CAM_POS MJS_PM_CAM_QUAT MJS_PM_CAM_FOVY", "environments/tasks/point_reach.py", r"^TOP_DOWN_CAMERA_CONFIG = CameraConfig\((.*)\)"),
    ("MJS_RR_PHYSICS_DT", "environments/tasks/robot_push_button.py", r"PHYSICS_TIMESTEP: float = ([\d.]+)"),
    ("MJS_RR_CONTROL_DT", "environments/tasks/robot_push_button.py", r"CONTROL_TIMESTEP: float = ([\d.]+)"),
    ("MJS_BP_MAX_CONTROL_STEPS", "environments/tasks/robot_push_button.py", r"MAX_CONTROL_STEPS_PER_EPISODE: int = (\d+)"),
    ("MJS_BP_GOAL_THRESHOLD", "environments/tasks/robot_push_button.py", r"GOAL_DISTANCE_THRESHOLD: float = ([\d.]+)"),
    ("MJS_BP_CAM_POS", "environments/tasks/robot_push_button.py", r"scene_camera_position: np.ndarray = np.array\((.*)\)"),
    ("MJS_BP_CAM_QUAT", "environments/tasks/robot_push_button.py", r"scene_camera_orientation: np.ndarray = np.array\((.*)\)"),
    ("MJS_WCAM_POS", "environments/tasks/robot_push_button.py", r"wrist_camera_position: np.ndarray = np.array\((.*)\)"),
    ("MJS_WCAM_QUAT", "environments/tasks/robot_push_button.py", r"wrist_camera_orientation: np.ndarray = np.array\((.*)\)"),
    ("MJS_BP_CAM_FOVY", "environments/tasks/robot_push_button.py", r"CameraConfig\(scene_camera_position, scene_camera_orientation, (\d+)"),
    ("MJS_WCAM_FOVY", "environments/tasks/robot_push_button.py", r"CameraConfig\(wrist_camera_position, wrist_camera_orientation, (\d+)"),
    ("BP_ROBOT_SPACE", "environments/tasks/robot_push_button.py", r"self\.robot_spawn_space = EuclideanSpace\((.*)\)"),
    ("BP_SWITCH_SPACE", "environments/tasks/robot_push_button.py", r"self\.target_spawn_space = EuclideanSpace\((.*)\)"),
    ("MJS_BP_ROBOT_END_POS", "environments/tasks/robot_push_button.py", r"self\.robot_end_position = np.array\((.*?)\)"),
    ("MJS_RR_CAM_POS MJS_RR_CAM_QUAT MJS_RR_CAM_FOVY", "environments/tasks/robot_reach.py", r"FRONT_TILTED_CAMERA_CONFIG = CameraConfig\((.*)\)"),
    ("RR_SPACE", "environments/tasks/robot_reach.py", r"self\.robot_spawn_space = EuclideanSpace\((.*)\)"),
    ("MJS_PP_MAX_CONTROL_STEPS", "environments/tasks/robot_planar_push.py", r"max_control_steps_per_episode: int = (\d+)"),
    ("MJS_PP_NEAREST_COEF", "environments/tasks/robot_planar_push.py", r"nearest_object_reward_coefficient: float = ([\d.]+)"),
    ("MJS_PP_TARGET_RADIUS", "environments/tasks/robot_planar_push.py", r"target_radius = ([\d.]+)"),
    ("MJS_PP_MAX_OBJECTS", "environments/tasks/robot_planar_push.py", r"n_objects: int = (\d+)"),
    ("PP_ROBOT_SPACE", "environments/tasks/robot_planar_push.py", r"self\.robot_spawn_space = EuclideanSpace\((.*)\)"),
    ("PP_OBJECT_SPACE", "environments/tasks/robot_planar_push.py", r"self\.object_spawn_space = EuclideanSpace\((.*)\)"),
    ("PP_TARGET_SPACE", "environments/tasks/robot_planar_push.py", r"self\.target_spawn_space = EuclideanSpace\((.*)\)"),
    ("SW_BOX_SIZE", "entities/props/switch.py", r"self\.box_size = ([\d.]+)"),
    ("SW_BOX_HEIGHT", "entities/props/switch.py", r"self\.box_height = ([\d.]+)"),
    ("MJS_SW_MIN_FORCE MJS_SW_MAX_FORCE", "entities/props/switch.py", r"target_force_range=\((.*?)\)"),
    ("MJS_SW_SITE_SCALE", "entities/props/switch.py", r"size=self\._button_geom\.size \* ([\d.]+)"),
    ("MJS_BLOCK_MASS", "entities/props/google_block.py", r"self\.mass = ([\d.]+)"),
    ("MJS_BLOCK_CONDIM", "entities/props/google_block.py", r"\.condim = (\d+)"),
    ("MJS_BLOCK_FRICTION", "entities/props/google_block.py", r"\.friction = np.array\((.*?)\)"),
    ("BLOCK_COLOR_RED", "entities/props/google_block.py", r"^RED = \((.*)\)"),
    ("BLOCK_COLOR_BLUE", "entities/props/google_block.py", r"^BLUE = \((.*)\)"),
    ("BLOCK_COLOR_GREEN", "entities/props/google_block.py", r"^GREEN = \((.*)\)"),
    ("BLOCK_COLOR_YELLOW", "entities/props/google_block.py", r"^YELLOW = \((.*)\)"),
    ("BLOCK_COLOR_ORANGE", "entities/props/google_block.py", r"^ORANGE = \((.*)\)"),
    ("BLOCK_COLOR_PURPLE", "entities/props/google_block.py", r"^PURPLE = \((.*)\)"),
    ("MJS_G2F85_TCP_Z", "entities/eef/gripper.py", r"return np.array\(\[0\.0, 0\.0, ([\d.]+)\]\)"),
    ("MJS_G2F85_OPEN", "entities/eef/gripper.py", r"return ([\d.]+)\s*$"),
    ("MJS_G2F85_MAX_DRIVER", "entities/eef/gripper.py", r"max_driver_joint_angle = ([\d.]+)"),
]


def literal_pins():
    import re
    root = REF.parent  # mujoco_sim/
    out = {}
    for name, rel, pattern in LITERALS:
        rx = re.compile(pattern)
        for no, line in enumerate((root / rel).read_text().splitlines(), 1):
            m = rx.search(line.strip() if pattern.startswith("^") else line)
            if m:
                out[name] = {"file": f"mujoco_sim/{rel}", "line": no, "numbers": [float(x) for x in re.findall(r"-?\d+\.?\d*(?:e-?\d+)?", m.group(1))]}
                break
        else:
            raise SystemExit(f"no line of {rel} matches {pattern}")
    return out


def lerobot_config_pins():
    """scripts/lerobot/configs/*.yaml (the reference's LeRobot training configs): what they say about the env's interface - task id,
    fps, episode length, state / action widths, image size, the observation column names and their shapes. Numbers and key names only."""
    import yaml

    out = {}
    for path in sorted((REF.parents[1] / "scripts" / "lerobot" / "configs").glob("*.yaml")):
        cfg = yaml.safe_load(path.read_text())
        env = cfg["env"]
        shapes = cfg["policy"]["input_shapes"]
        out[path.name] = {
            "file": f"scripts/lerobot/configs/{path.name}",
            "fps": cfg["fps"], "task": env["task"], "image_size": env["image_size"], "state_dim": env["state_dim"], "action_dim": env["action_dim"],
            "episode_length": env["episode_length"],
            "image_keys": {k: v for k, v in shapes.items() if k.startswith("observation.images.")},
            "state_key": [k for k in shapes if k == "observation.state"],
            "action_key": list(cfg["policy"]["output_shapes"].keys()),
            "stats_keys": sorted(cfg.get("override_dataset_stats", {}).keys()),
        }
    return out


if __name__ == "__main__":
    d = main()
    d["lerobot_configs"] = lerobot_config_pins()
    d["literals"] = literal_pins()
    OUT.write_text(json.dumps(d, indent=1) + "\n")
    print("wrote", OUT, "with", len(d["literals"]), "source literals")
