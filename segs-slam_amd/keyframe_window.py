"""Keyframe selection of the mapper loop: GaussianMapper::useOneRandomSlidingWindowKeyframe and generateKfidRandomShuffle
(src/gaussian_mapper.cpp:1444-1495) as host logic.

Semantics kept: keyframes live in insertion order; adding one invalidates the shuffle (GaussianScene::addKeyframe clears
`kfid_shuffled_`, :605/:1279), the next use draws a fresh permutation of all indices and KEEPS the walk position
(`kfid_shuffle_idx_` is never reset); a use advances the position cyclically until it meets a keyframe whose
`remaining_times_of_use` is positive; when the walk comes back to where it started without finding one, every keyframe
receives one more use (:1477-1479) and the walk goes on; the chosen keyframe's budget is decremented and its use counted.
The reference seeds std::mt19937 from std::random_device; here the permutation comes from a seeded NumPy generator so that
every rank of a keyframe-parallel run draws the same sequence.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np


class SlidingWindowKeyframes:
    def __init__(self, seed: int = 0):
        self._rng = np.random.default_rng(seed)
        self.remaining: List[int] = []          # remaining_times_of_use_ per keyframe, insertion order
        self.used_times: Dict[int, int] = {}    # kfs_used_times_
        self._shuffle: List[int] = []
        self._shuffled = False
        self._idx = 0                           # kfid_shuffle_idx_

    def add_keyframe(self, times_of_use: int) -> int:
        """GaussianScene::addKeyframe + increaseKeyframeTimesOfUse(new_kf, newKeyframeTimesOfUse()) (:605-607)."""
        self.remaining.append(int(times_of_use))
        self._shuffled = False
        return len(self.remaining) - 1

    def increase_times_of_use(self, kf: int, times: int):
        self.remaining[kf] += int(times)

    def __len__(self):
        return len(self.remaining)

    def use_one(self) -> int:
        """Index of the keyframe to train on next (-1 without keyframes)."""
        n = len(self.remaining)
        if n == 0:
            return -1
        if not self._shuffled:
            self._shuffle = [int(i) for i in self._rng.permutation(n)]
            self._shuffled = True
        start = self._idx
        while True:
            self._idx += 1
            if self._idx >= n:
                self._idx = 0
            if self._idx == start:
                for k in range(n):
                    self.remaining[k] += 1
            kf = self._shuffle[self._idx]
            if self.remaining[kf] > 0:
                break
        self.used_times[kf] = self.used_times.get(kf, 0) + 1
        self.remaining[kf] -= 1
        return kf

    def use_for_ranks(self, world: int) -> List[int]:
        """One draw per rank of a keyframe-parallel step; every rank calls this with the same state and takes entry `rank`."""
        return [self.use_one() for _ in range(world)]


def nerfpp_norm(camera_centers) -> tuple:
    """GaussianScene::getNerfppNorm (src/gaussian_scene.cpp:113-149): (translate, radius) with translate = -mean of the
    camera centres and radius = 1.1 x the largest distance of a centre from that mean, in float32.  The radius is the
    scene extent (`cameras_extent`) that scales the position / offset learning rates (spatial_lr_scale)."""
    c = np.asarray(camera_centers, dtype=np.float32).reshape(-1, 3)
    if c.shape[0] == 0:
        raise ValueError("no keyframes")
    avg = np.zeros(3, dtype=np.float32)
    for row in c:                      # accumulated in float32, camera by camera, like the reference
        avg += row
    avg /= np.float32(c.shape[0])
    max_dist = np.float32(0.0)
    for row in c:
        d = np.float32(np.linalg.norm((row - avg).astype(np.float32)))
        if d > max_dist:
            max_dist = d
    return -avg, float(np.float32(max_dist * np.float32(1.1)))
