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


# ----------------------------------------------------------------------------------------------
# BASELINE configs[4]: an episode = a walkthrough map and an unshuffle map of the same room, 300 room
# frames each (the reference's phases: /root/reference/agent.py:314-415), then
# predict_scene_differences (agent.py:435-444).  Between the phases one "object" has moved.
# ----------------------------------------------------------------------------------------------
OBJECT_CLASSES = (41, 47, 52)          # classes of the three box-shaped objects of a synthetic episode
MOVED_CLASS = 47


def episode_objects(episode, phase):
    """Centres of the episode's objects (on the walls of the 6 x 6 x 3 m room); MOVED_CLASS sits somewhere
    else in phase 1 (unshuffle) than in phase 0 (walkthrough)."""
    g = torch.Generator().manual_seed(7919 * episode + 13)
    u = torch.rand(4, generator=g)
    fixed_a = (2.95, float(-1.5 + 3.0 * u[0]), -0.8)
    fixed_b = (float(-1.5 + 3.0 * u[1]), -2.95, -0.6)
    x0 = float(-2.0 + 1.0 * u[2])
    moved = (x0, 2.95, -0.9) if phase == 0 else (x0 + 2.2, 2.95, -0.9)
    return {OBJECT_CLASSES[0]: fixed_a, OBJECT_CLASSES[2]: fixed_b, MOVED_CLASS: moved}


def episode_trajectory(episode, phase, n, height=480, width=640, num_classes=NUM_CLASSES, half_size=0.45):
    """room_trajectory(seed = 1000 * episode + phase) with the episode's objects painted into the class-id
    images (a pixel whose hit point lies within `half_size` of an object's centre carries its class)."""
    tr = room_trajectory(n, height, width, seed=1000 * episode + phase, num_classes=num_classes)
    focal = height / 2.0 / np.tan(np.radians(90.0) / 2.0)
    cam = project_camera_rays(height, width, focal, focal)
    sem = tr["semantic"].clone()
    objects = episode_objects(episode, phase)
    for t in range(n):
        eye = spherical_to_cartesian(tr["yaw"][t], tr["elevation"][t])
        up = spherical_to_cartesian(tr["yaw"][t], tr["elevation"][t] + np.pi / 2)
        q = (cam.unsqueeze(-2) * rotation_matrix(eye, up)).sum(-1)
        hit = tr["position"][t] + q * tr["depth"][t]
        # background classes stay away from the object classes
        bg = sem[t]
        for cls in OBJECT_CLASSES:
            bg[bg == cls] = 0
        for cls, centre in objects.items():
            d = (hit - torch.tensor(centre)).abs()
            bg[(d[..., 0] < half_size) & (d[..., 1] < half_size) & (d[..., 2] < half_size)] = cls
    tr["semantic"] = sem
    return tr


def prepare_episode(episode, device, n_frames=300, height=480, width=640, num_classes=NUM_CLASSES):
    """The episode's two trajectories with depth and class ids resident on `device` (poses stay on the host:
    the layers do the pose trigonometry there, like the reference)."""
    phases = []
    for phase in range(2):
        tr = episode_trajectory(episode, phase, n_frames, height, width, num_classes)
        phases.append(dict(position=tr["position"], yaw=tr["yaw"], elevation=tr["elevation"],
                           depth=tr["depth"].to(device), semantic=tr["semantic"].to(device)))
    return dict(episode=episode, n_frames=n_frames, phases=phases)


def make_episode_layers(device, height=480, width=640, map_size=256, num_classes=NUM_CLASSES, grid_resolution=0.05):
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    kw = dict(camera_height=height, camera_width=width, map_height=map_size, map_width=map_size, map_depth=map_size,
              feature_size=num_classes, grid_resolution=grid_resolution)
    return [SemanticProjectionLayer(**kw).to(device) for _ in range(2)]


def run_episode(prepared, layers, batch=64):
    """One synthetic episode: both phases' frames fused into their semantic maps in sequential batches
    (= per-frame layer.update() calls), then predict_scene_differences over the object classes.
    Returns the episode's counters (what the ranks all-reduce at the end of a multi-GPU run)."""
    from mass_amd.utils.experimentation import predict_scene_differences
    from mass_amd.utils.reductions import map_stats
    n_frames, frames = prepared["n_frames"], 0
    for tr, lay in zip(prepared["phases"], layers):
        lay.reset()
        for b0 in range(0, n_frames, batch):
            sl = slice(b0, min(b0 + batch, n_frames))
            lay.update_batch(dict(position=tr["position"][sl], yaw=tr["yaw"][sl], elevation=tr["elevation"][sl],
                                  depth=tr["depth"][sl], semantic=tr["semantic"][sl]), sequential=True, validate="defer")
            frames += sl.stop - sl.start
        lay.check_labels()
    obj, goals0, goals1 = predict_scene_differences(layers[0], layers[1], None, None, set(), list(OBJECT_CLASSES),
                                                    confidence_threshold=0.0, contour_padding=0, distance_threshold=0.5)
    shift = float((goals1[0] - goals0[0]).norm()) if goals0 else 0.0
    # the maps' counters (occupied voxels, sum |map|) in one pass per map (mf_map_stats: the torch expressions
    # `(data != 0).any(-1).sum()` and `data.abs().sum()` move a 3.6 GB map five times)
    stats = [map_stats(l.data) for l in layers]
    return dict(episodes=1, frames=frames, moved_found=int(obj == MOVED_CLASS), n_matches=len(goals0), shift_m=shift,
                occupied_voxels=float(sum(s[0] for s in stats)), map_abs_sum=float(sum(s[1] for s in stats)))
