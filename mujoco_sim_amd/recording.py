"""Episode recorder with the LeRobot dataset conventions of the reference's demonstration collector.

Mirrors ``scripts/demonstration_collection.py:39-167`` (``LeRobotDatasetRecorder``): same constructor arguments,
``start_episode / record / save_episode / finish_recording / n_recorded_episodes``, same feature schema
(``next.reward``, ``next.success``, ``seed``, ``timestamp``, ``action``, one column per observation key with
``/`` -> ``_``, image keys under ``observation.images.*``, and ``observation.state`` = the concatenated state
observations). The reference delegates storage to the third-party ``lerobot`` package (absent here); this recorder
writes the LeRobot v2 directory layout itself with pyarrow: ``data/chunk-000/episode_XXXXXX.parquet``,
``meta/info.json``, ``meta/episodes.jsonl``, ``meta/tasks.jsonl`` (images as raw uint8 columns, i.e. the reference's
``use_videos=False`` mode; no video encoder is available). ``record_batch`` is the batched path: it takes the
outputs of one ``HipVectorEnv.step`` for all N envs and closes an episode whenever an env reports LAST.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np


# The reference's LeRobot configs (scripts/lerobot/configs/act_robot_button_push.yaml:24,47, dp_robot_button_push.yaml:24,148) name the
# wrist camera's column `ur5e_WristCamera_rgb_image`; the reference's code at HEAD produces `ur5e/Camera/rgb_image` (the Camera
# entity keeps CameraConfig.name = "Camera", robot_push_button.py:92-96), i.e. `ur5e_Camera_rgb_image` after the recorder's
# '/' -> '_' mapping. The recorder emits the CODE's key; `key_aliases` (or this table) renames columns for a consumer that
# was configured from the YAMLs.
REFERENCE_YAML_KEY_ALIASES = {"observation.images.ur5e_Camera_rgb_image": "observation.images.ur5e_WristCamera_rgb_image"}


def _to_numpy(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


class LeRobotDatasetRecorder:
    DEFAULT_FEATURES = {
        "next.reward": {"dtype": "float32", "shape": (1,), "names": None},
        "next.success": {"dtype": "bool", "shape": (1,), "names": None},
        "seed": {"dtype": "int64", "shape": (1,), "names": None},
        "timestamp": {"dtype": "float32", "shape": (1,), "names": None},
    }

    def __init__(self, env, root_dataset_dir, dataset_name: str, fps: int, use_videos: bool = False, task: str = "", key_aliases: dict | None = None):
        if use_videos:
            raise NotImplementedError("no video encoder in this environment: images are stored as raw uint8 columns (use_videos=False)")
        self.root_dataset_dir = Path(root_dataset_dir)
        self.dataset_name = dataset_name
        self.fps = fps
        self.task = task
        self._n_recorded_episodes = 0
        self._n_frames = 0
        self.key_mapping_dict = {}
        space = getattr(env, "single_observation_space", None) or env.observation_space
        action_space = getattr(env, "single_action_space", None) or env.action_space
        spaces = space.spaces if hasattr(space, "spaces") else space
        features = {k: dict(v) for k, v in self.DEFAULT_FEATURES.items()}
        self.image_keys = [k for k in spaces.keys() if "image" in k]
        for key in self.image_keys:  # 'observation.images.<key>' with '/' -> '_' (demonstration_collection.py:84-97)
            lerobot_key = key if key.startswith("observation.images") else f"observation.images.{key}"
            self.key_mapping_dict[key] = lerobot_key.replace("/", "_")
            features[self.key_mapping_dict[key]] = {"dtype": "image", "shape": tuple(spaces[key].shape), "names": None}
        self.state_keys = [k for k in spaces.keys() if k not in self.image_keys]
        for key in self.state_keys:
            self.key_mapping_dict[key] = key.replace("/", "_")
            features[self.key_mapping_dict[key]] = {"dtype": "float32", "shape": tuple(spaces[key].shape), "names": None}
        features["observation.state"] = {"dtype": "float32", "shape": (sum(int(np.prod(spaces[k].shape)) for k in self.state_keys),), "names": None}
        features["action"] = {"dtype": "float32", "shape": tuple(action_space.shape), "names": None}
        if key_aliases:  # e.g. REFERENCE_YAML_KEY_ALIASES: write the columns under the names a YAML-configured consumer expects
            self.key_mapping_dict = {k: key_aliases.get(v, v) for k, v in self.key_mapping_dict.items()}
            features = {key_aliases.get(k, k): v for k, v in features.items()}
        self.features = features
        (self.root_dataset_dir / "meta").mkdir(parents=True, exist_ok=True)
        (self.root_dataset_dir / "data" / "chunk-000").mkdir(parents=True, exist_ok=True)
        self._frames = []          # single-env path
        self._batch_frames = None  # batched path: per-env lists

    # ------------------------------------------------------------------ single-env API of the reference
    def start_episode(self):
        self._frames = []

    def _frame(self, obs, action, reward, done, seed=0, t=0):
        frame = {"action": _to_numpy(action).astype(np.float32).ravel(), "next.reward": np.array([reward], np.float32),
                 "next.success": np.array([bool(done)]), "seed": np.array([seed], np.int64), "timestamp": np.array([t / self.fps], np.float32)}
        for key in self.image_keys:
            frame[self.key_mapping_dict[key]] = _to_numpy(obs[key]).astype(np.uint8)
        state = []
        for key in self.state_keys:
            v = _to_numpy(obs[key]).astype(np.float32).ravel()
            frame[self.key_mapping_dict[key]] = v
            state.append(v)
        frame["observation.state"] = np.concatenate(state) if state else np.zeros(0, np.float32)
        return frame

    def record(self, obs, action, reward, done, info=None):
        self._frames.append(self._frame(obs, action, reward, done, t=len(self._frames)))

    def save_episode(self):
        self._write_episode(self._frames)
        self._frames = []

    # ------------------------------------------------------------------ batched path (HipVectorEnv)
    def record_batch(self, obs, actions, reward, success, last, seeds=None, active=None):
        """obs: dict of [N, ...] (observation BEFORE the action, as in the reference loop), actions [N, A], reward [N],
        success [N] bool, last [N] bool (episode ended with this step). Finished episodes are written out. ``active`` [N]
        bool (optional): envs whose step is recorded (a collector that stops an env after its episode passes the rest)."""
        actions, reward, success, last = (_to_numpy(x) for x in (actions, reward, success, last))
        active = None if active is None else _to_numpy(active).astype(bool)
        obs = {k: _to_numpy(v) for k, v in obs.items()}
        n = len(reward)
        if self._batch_frames is None:
            self._batch_frames = [[] for _ in range(n)]
        for i in range(n):
            if active is not None and not active[i]:
                continue
            fr = self._frame({k: v[i] for k, v in obs.items()}, actions[i], reward[i], success[i], seed=0 if seeds is None else int(seeds[i]),
                             t=len(self._batch_frames[i]))
            self._batch_frames[i].append(fr)
            if last[i]:
                self._write_episode(self._batch_frames[i])
                self._batch_frames[i] = []

    # ------------------------------------------------------------------ storage
    def _write_episode(self, frames):
        import pyarrow as pa
        import pyarrow.parquet as pq

        if not frames:
            return
        ep = self._n_recorded_episodes
        cols = {}
        for key, feat in self.features.items():
            stacked = np.stack([f[key] for f in frames])
            flat = stacked.reshape(len(frames), -1)
            cols[key] = pa.FixedSizeListArray.from_arrays(pa.array(flat.ravel()), flat.shape[1])
        cols["episode_index"] = pa.array(np.full(len(frames), ep, np.int64))
        cols["frame_index"] = pa.array(np.arange(len(frames), dtype=np.int64))
        cols["index"] = pa.array(np.arange(self._n_frames, self._n_frames + len(frames), dtype=np.int64))
        cols["task_index"] = pa.array(np.zeros(len(frames), np.int64))
        pq.write_table(pa.table(cols), self.root_dataset_dir / "data" / "chunk-000" / f"episode_{ep:06d}.parquet")
        with open(self.root_dataset_dir / "meta" / "episodes.jsonl", "a") as f:
            f.write(json.dumps({"episode_index": ep, "tasks": [self.task], "length": len(frames)}) + "\n")
        self._n_recorded_episodes += 1
        self._n_frames += len(frames)

    def finish_recording(self):
        info = {"codebase_version": "v2.0", "repo_id": self.dataset_name, "fps": self.fps, "total_episodes": self._n_recorded_episodes,
                "total_frames": self._n_frames, "total_tasks": 1, "chunks_size": 1000,
                "data_path": "data/chunk-{episode_chunk:03d}/episode_{episode_index:06d}.parquet",
                "features": {k: {"dtype": v["dtype"], "shape": list(v["shape"]), "names": v["names"]} for k, v in self.features.items()}}
        (self.root_dataset_dir / "meta" / "info.json").write_text(json.dumps(info, indent=2))
        (self.root_dataset_dir / "meta" / "tasks.jsonl").write_text(json.dumps({"task_index": 0, "task": self.task}) + "\n")

    @property
    def n_recorded_episodes(self):
        return self._n_recorded_episodes
