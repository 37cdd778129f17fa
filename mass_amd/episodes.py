"""Synthetic RGB-D(+segmentation) workloads of the benchmark configurations
(BASELINE.json `configs`) and the episode sharding used for the
multi-GPU run.  Input generation only: host torch ops, no reference code.
"""
import numpy as np
import torch

from mass_amd.utils.projection import (project_camera_rays, spherical_to_cartesian, rotation_matrix)

NUM_CLASSES = 54      # /root/reference/mass/thor/segmentation_config.py:110


def dist_a_frames(n, seed0=0, height=480, width=640, num_classes=NUM_CLASSES):
    """Config 2, distribution A (adversarial: neighbouring pixels land in
    unrelated voxels): depth = 0.5 + 4.5 U[0,1), uniform labels, position ~
    N(0, 0.3^2) per axis, yaw ~ U[0, 2pi), elevation ~ U[-0.6, 0]; frame i uses
    torch.Generator().manual_seed(seed0 + i)."""
    depth, label, pos, yaw, el = [], [], [], [], []
    for s in range(seed0, seed0 + n):
        g = torch.Generator().manual_seed(s)
        depth.append(0.5 + 4.5 * torch.rand(height, width, 1, generator=g))
        label.append(torch.randint(0, num_classes, (height, width), generator=g).to(torch.uint8))
        pos.append(0.3 * torch.randn(3, generator=g))
        yaw.append(2 * np.pi * torch.rand((), generator=g))
        el.append(-0.6 * torch.rand((), generator=g))
    return dict(position=torch.stack(pos), yaw=torch.stack(yaw), elevation=torch.stack(el),
                depth=torch.stack(depth), semantic=torch.stack(label))


def room_depth(position, yaw, elevation, cam_rays, half=(3.0, 3.0, 1.5)):
    """Analytic depth of a box room centred at the origin (distribution B):
    `depth` is the multiple of the (un-normalised) camera ray that reaches the
    nearest wall, i.e. z-depth along the optical axis, as the reference's
    depth images are.  Returns depth [H, W, 1], hit point [H, W, 3], wall id [H, W]."""
    eye = spherical_to_cartesian(yaw, elevation)
    up = spherical_to_cartesian(yaw, elevation + np.pi / 2)
    R = rotation_matrix(eye, up)
    q = (cam_rays.unsqueeze(-2) * R).sum(dim=-1)                     # world direction per pixel
    half = torch.tensor(half)
    t_pos = (half - position) / q.clamp(min=1e-12)
    t_neg = (-half - position) / q.clamp(max=-1e-12)
    t = torch.where(q > 0, t_pos, t_neg)                             # exit distance per axis
    depth, axis = t.min(dim=-1)
    hit = position + q * depth.unsqueeze(-1)
    wall = axis * 2 + (torch.gather(q, -1, axis.unsqueeze(-1)).squeeze(-1) > 0).long()
    return depth.unsqueeze(-1).to(torch.float32), hit, wall


def room_trajectory(n, height=480, width=640, seed=0, num_classes=NUM_CLASSES, fov=90.0):
    """Config 3: camera on a radius-1.5 m circle inside a 6 x 6 x 3 m room,
    yaw tangent + 0.3 sin, elevation -0.5 (the agent looks down, agent.py:310-312);
    labels = hash of (wall, 0.5 m tile), rgb = smooth function of the hit point.
    `seed` rotates the start angle and offsets the centre slightly."""
    g = torch.Generator().manual_seed(seed)
    phase = float(2 * np.pi * torch.rand((), generator=g))
    centre = 0.2 * torch.randn(2, generator=g)
    focal = height / 2.0 / np.tan(np.radians(fov) / 2.0)
    cam = project_camera_rays(height, width, focal, focal)
    out = dict(position=[], yaw=[], elevation=[], depth=[], semantic=[], rgb=[])
    for t in range(n):
        a = phase + 2 * np.pi * t / max(n, 1)
        pos = torch.tensor([centre[0] + 1.5 * np.cos(a), centre[1] + 1.5 * np.sin(a), 0.0], dtype=torch.float32)
        yaw = torch.tensor(a + np.pi / 2 + 0.3 * np.sin(3 * a), dtype=torch.float32)
        el = torch.tensor(-0.5, dtype=torch.float32)
        depth, hit, wall = room_depth(pos, yaw, el, cam)
        tile = torch.floor(hit / 0.5).long()
        lab = (wall * 7919 + tile[..., 0] * 31 + tile[..., 1] * 17 + tile[..., 2] * 13) % num_classes
        rgb = 0.5 + 0.5 * torch.sin(hit * torch.tensor([1.3, 2.1, 3.7]))
        out["position"].append(pos); out["yaw"].append(yaw); out["elevation"].append(el)
        out["depth"].append(depth); out["semantic"].append(lab.to(torch.uint8)); out["rgb"].append(rgb.to(torch.float32))
    return {k: torch.stack(v) for k, v in out.items()}


def shard_episodes(n_episodes, rank, world_size):
    """Episode e runs on rank e mod world_size (the reference slices tasks the
    same way with --start-task/--every-tasks, agent.py:154-155)."""
    return [e for e in range(n_episodes) if e % world_size == rank]
