"""Episode video capture around a single-env adapter.

Mirrors ``mujoco_sim/gym_video_wrapper.py:11-97`` (``VideoRecorderWrapper``): same constructor arguments and
behaviour (every ``capture_every_n_episodes``-th episode is captured from ``env.render()``, optionally downscaled,
with 10 black lead-in frames, and written as ``episode_<k>.gif``), on the gymnasium 5-tuple step API of the adapter
here. GIFs are written with Pillow (``imageio`` / ``wandb`` are absent; ``log_wandb`` is accepted and ignored).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np


class VideoRecorderWrapper:
    def __init__(self, env, video_folder, capture_every_n_episodes: int = 10, log_wandb: bool = False, rescale_video_factor: int = 1):
        self.env = env
        self.frames = []
        self.episode_count = -1
        self.capture_period = capture_every_n_episodes
        self.video_path = Path(video_folder)
        self.log_wandb = log_wandb
        self.video_path.mkdir(exist_ok=True, parents=True)
        self.rescale_factor = rescale_video_factor
        self.num_black_frames_at_beginning = 10

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self, **kwargs):
        self.episode_count += 1
        out = self.env.reset(**kwargs)
        if self._should_capture_this_episode():
            self.frames = []
            self._capture_current_frame()
        return out

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        if self._should_capture_this_episode():
            self._capture_current_frame()
            if terminated or truncated:
                self._create_and_store_gif()
        return obs, reward, terminated, truncated, info

    def _should_capture_this_episode(self):
        return self.episode_count % self.capture_period == 0

    def _capture_current_frame(self):
        from PIL import Image

        rgb = Image.fromarray(np.asarray(self.env.render()))
        rgb = rgb.resize((rgb.size[0] // self.rescale_factor, rgb.size[1] // self.rescale_factor))
        self.frames.append(np.array(rgb))

    def _create_and_store_gif(self):
        from PIL import Image

        gif_path = self.video_path / f"episode_{self.episode_count}.gif"
        black = [np.zeros_like(self.frames[0])] * self.num_black_frames_at_beginning
        images = [Image.fromarray(f) for f in black + self.frames]
        images[0].save(gif_path, save_all=True, append_images=images[1:], duration=100, loop=0)
        return gif_path
