"""Synthetic "Refine Sequence" scenes shared by tests/test_refiner_*.py: a wavy grid mesh, an OpenGL
camera drifting in front of it, keypoints at random pixels and analytic flows between frames
+-{1,2,4,8} apart (what the analysis pass would have written to the database)."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pnp_oracle as po  # noqa: E402
import refine_oracle as ro  # noqa: E402

W, H, F = 960.0, 540.0, 1100.0
SKIPS = (-8, -4, -2, -1, 1, 2, 4, 8)


def grid_mesh(n=12, size=4.0):
    xs = np.linspace(-size / 2, size / 2, n + 1)
    X, Y = np.meshgrid(xs, xs)
    Z = 0.3 * np.sin(1.3 * X) * np.cos(1.1 * Y)
    verts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1).astype(np.float32)
    tris = []
    for j in range(n):
        for i in range(n):
            a = j * (n + 1) + i
            tris += [[a, a + 1, a + n + 2], [a, a + n + 2, a + n + 1]]
    return verts, np.array(tris, np.uint32)


def rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    return po.quat_to_R(np.concatenate([[math.cos(ang / 2)], axis * math.sin(ang / 2)]))


def true_camera(t, opencv=False, rate=1.0):
    t = t * rate   # the mesh leaves the view after about 40 frames at rate 1
    R = rot([0.2, 1.0, 0.1], 0.012 * t) @ rot([1, 0, 0], 0.004 * t)
    tr = np.array([0.03 * t, -0.02 * t, -6.0 + 0.01 * t])
    f = F if opencv else -F
    if opencv:   # look down +z: flip the scene in front of the camera
        R = np.diag([1.0, -1.0, -1.0]) @ R
        tr = np.diag([1.0, -1.0, -1.0]) @ tr
    return po.Camera(fx=f, fy=f, cx=W / 2, cy=H / 2, aspect_ratio=1.0, width=W, height=H, opencv=opencv,
                     q=po.R_to_quat(R), t=tr)


def perturbed(cams, rng, rot_sigma=0.003, t_sigma=0.02, f_scale=1.0):
    out = [cams[0]]
    for c in cams[1:-1]:
        fy = c.fy * f_scale
        out.append(po.Camera(fx=fy * c.aspect_ratio, fy=fy, cx=c.cx, cy=c.cy, aspect_ratio=c.aspect_ratio, width=c.width,
                             height=c.height, opencv=c.opencv, q=po.quat_step_post(c.q, rng.normal(0, rot_sigma, 3)),
                             t=c.t + rng.normal(0, t_sigma, 3)))
    return out + [cams[-1]]


def make_flows(verts, tris, model, cams, first_frame, n_kp=150, noise=0.0, seed=3):
    """-> keypoints {frame: (n,2) f32}, flows {frame: [(to, idx u32, tgt f32 (m,2))]}"""
    rng = np.random.default_rng(seed)
    model = np.asarray(model, np.float64)
    minv = np.linalg.inv(model)
    n = len(cams)
    kps, world, flows = {}, {}, {}
    for i, c in enumerate(cams):
        xy = np.floor(rng.uniform([20, 20], [W - 20, H - 20], (n_kp, 2))).astype(np.float32)
        o, d = ro._rays_object(c, minv, xy)
        hit, _, _, _, _, pos = po.raycast_closest(verts, tris, o, d)
        kps[first_frame + i] = xy
        world[i] = (pos @ model[:3, :3].T + model[:3, 3], hit)
    for i in range(n):
        pw, hit = world[i]
        rows = []
        for s in SKIPS:
            j = i + s
            if j < 0 or j >= n:
                continue
            x2, Z = cams[j].project_world(pw)
            front = (Z[:, 2] > 0) if cams[j].opencv else (Z[:, 2] < 0)
            ok = hit & front & (x2[:, 0] > 0) & (x2[:, 0] < W) & (x2[:, 1] > 0) & (x2[:, 1] < H)
            idx = np.nonzero(ok | ~hit)[0].astype(np.uint32)   # keypoints off the mesh still get (wrong) flows
            tgt = np.where(hit[idx, None], x2[idx], kps[first_frame + i][idx] + 1.0) + rng.normal(0, noise, (len(idx), 2))
            rows.append((first_frame + j, idx, tgt.astype(np.float32)))
        flows[first_frame + i] = rows
    return kps, flows


def write_database(core, path, kps, flows):
    db = core.Database(path)
    for f, xy in kps.items():
        db.write_keypoints(f, xy)
    for f, rows in flows.items():
        for to, idx, tgt in rows:
            db.write_image_pair_flow(f, to, idx, tgt, np.zeros(len(idx), np.float32))
    db.close()


def read_database(core, path, first_frame, n):
    """What CachedDatabase reads, in the order the SQL returns it."""
    db = core.Database(path)
    kps, flows = {}, {}
    for f in range(first_frame, first_frame + n):
        kps[f] = np.asarray(db.read_keypoints(f), np.float32).reshape(-1, 2)
        rows = []
        for to in db.find_optical_flows_from_image(f):
            fl = db.read_image_pair_flow(f, to)
            rows.append((to, np.asarray(fl.src_kps_indices), np.asarray(fl.tgt_kps).reshape(-1, 2)))
        flows[f] = rows
    db.close()
    return kps, flows


def to_core_trajectory(core, cams, first_frame):
    traj = core.CameraTrajectory(first_frame, len(cams))
    for i, c in enumerate(cams):
        intr = core.CameraIntrinsics(fx=c.fx, fy=c.fy, cx=c.cx, cy=c.cy, aspect_ratio=c.aspect_ratio, width=c.width,
                                     height=c.height,
                                     convention=core.CameraConvention.OpenCV if c.opencv else core.CameraConvention.OpenGL)
        pose = core.Pose()
        pose.q, pose.t = np.asarray(c.q, np.float32), np.asarray(c.t, np.float32)
        traj.set(first_frame + i, core.CameraState(intr, pose))
    return traj


def from_core_trajectory(traj, like):
    out = []
    for i, c in enumerate(like):
        s = traj.get(traj.first_frame() + i)
        out.append(po.Camera(fx=s.intrinsics.fx, fy=s.intrinsics.fy, cx=s.intrinsics.cx, cy=s.intrinsics.cy,
                             aspect_ratio=s.intrinsics.aspect_ratio, width=c.width, height=c.height, opencv=c.opencv,
                             q=np.array(s.pose.q, float), t=np.array(s.pose.t, float)))
    return out


def angle(Ra, Rb):
    return math.acos(max(-1.0, min(1.0, (np.trace(Ra.T @ Rb) - 1) / 2)))
