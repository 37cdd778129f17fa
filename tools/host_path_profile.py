import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch
from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
from mass_amd.episodes import room_trajectory
dev = torch.device("cuda:0")
lay = SemanticProjectionLayer(camera_height=480, camera_width=640, map_height=256, map_width=256, map_depth=256, feature_size=54, grid_resolution=0.05).to(dev)
tr = room_trajectory(8, 480, 640, seed=1)
o = dict(position=tr["position"][0].numpy(), yaw=float(tr["yaw"][0]), elevation=float(tr["elevation"][0]), depth=tr["depth"][0].numpy(), semantic=tr["semantic"][0].numpy().astype(np.int64)[..., None])
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
print("poses          %.3f ms" % t(lambda: lay._poses(o["position"], o["yaw"], o["elevation"])))
print("depth upload   %.3f ms" % t(lambda: torch.as_tensor(o["depth"], dtype=torch.float32, device=dev)))
print("labels         %.3f ms" % t(lambda: lay._labels(o["semantic"], False)))
print("labels i64 up  %.3f ms" % t(lambda: torch.as_tensor(o["semantic"]).to(dev)))
d = torch.as_tensor(o["depth"], device=dev); l = lay._labels(o["semantic"], False); p = lay._poses(o["position"], o["yaw"], o["elevation"])
from mass_amd.utils.projection import fuse_frames
print("fuse (device)  %.3f ms" % t(lambda: fuse_frames(lay.bins_x, lay.bins_y, lay.bins_z, lay.rays, p, d, l, lay.data, workspace=lay._workspace)))
print("update total   %.3f ms" % t(lambda: lay.update(o)))
pin = torch.from_numpy(o["depth"]).pin_memory()
print("depth pinned   %.3f ms" % t(lambda: pin.to(dev, non_blocking=True)))
