"""Camera matrices in the conventions the rasterizer consumes.

The rasterizer reads 4x4 matrices in row-vector form (memory = transposed math
matrix, translation in elements 12..14), exactly what the reference's ``Camera``
stores (scene/cameras.py:54-57, built from utils/graphics_utils.py:38-71).
``MiniCam`` mirrors the attribute set ``render()`` touches
(scene/cameras.py:59-70).
"""
import math

import numpy as np
import torch


def world_to_view(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    """4x4 world->camera matrix from a camera-to-world rotation ``R`` and a
    world->camera translation ``t`` (utils/graphics_utils.py:38-49)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).transpose()
    Rt[:3, 3] = np.asarray(t, dtype=np.float64)
    Rt[3, 3] = 1.0
    c2w = np.linalg.inv(Rt)
    c2w[:3, 3] = (c2w[:3, 3] + np.asarray(translate, dtype=np.float64)) * scale
    return np.float32(np.linalg.inv(c2w))


def projection_matrix(znear, zfar, fovx, fovy):
    """OpenGL-style perspective matrix with z_sign=+1 (utils/graphics_utils.py:51-71)."""
    tan_y = math.tan(fovy / 2)
    tan_x = math.tan(fovx / 2)
    top, right = tan_y * znear, tan_x * znear
    bottom, left = -top, -right
    Pm = torch.zeros(4, 4)
    Pm[0, 0] = 2.0 * znear / (right - left)
    Pm[1, 1] = 2.0 * znear / (top - bottom)
    Pm[0, 2] = (right + left) / (right - left)
    Pm[1, 2] = (top + bottom) / (top - bottom)
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


def look_at_Rt(eye, target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)):
    """(R, t) in the reference's convention for a camera at ``eye`` looking at
    ``target``: +z forward, +y down; ``R`` is camera-to-world, ``t`` world-to-camera."""
    eye = np.asarray(eye, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, dtype=np.float64))
    if np.linalg.norm(right) < 1e-8:
        right = np.cross(fwd, np.array([1.0, 0.0, 0.0]))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    w2c_rot = np.stack([right, down, fwd], axis=0)
    return w2c_rot.transpose(), -w2c_rot @ eye


class MiniCam:
    """The camera attributes ``render()`` reads (scene/cameras.py:59-70)."""

    def __init__(self, width, height, fovy, fovx, znear, zfar, world_view_transform, full_proj_transform):
        self.image_width = width
        self.image_height = height
        self.FoVy = fovy
        self.FoVx = fovx
        self.znear = znear
        self.zfar = zfar
        self.world_view_transform = world_view_transform
        self.full_proj_transform = full_proj_transform
        self.camera_center = torch.inverse(self.world_view_transform.cpu())[3][:3].to(world_view_transform.device)


def make_camera(width, height, fovx, eye, target=(0.0, 0.0, 0.0), znear=0.01, zfar=100.0, device="cpu"):
    """Pinhole camera following scene/cameras.py:48-57: FoVy from the aspect
    ratio, ``full_proj = view^T @ proj^T``, ``campos = inverse(view^T)[3,:3]``."""
    focal = width / (2.0 * math.tan(fovx / 2))
    fovy = 2.0 * math.atan(height / (2.0 * focal))
    R, t = look_at_Rt(eye, target)
    view_t = torch.tensor(world_to_view(R, t)).transpose(0, 1).contiguous()
    proj_t = projection_matrix(znear, zfar, fovx, fovy).transpose(0, 1)
    full = view_t.unsqueeze(0).bmm(proj_t.unsqueeze(0)).squeeze(0).contiguous()
    return MiniCam(width, height, fovy, fovx, znear, zfar, view_t.to(device), full.to(device))


def orbit_eye(index, count, radius=4.0, height=0.6):
    """Eye positions on a circular orbit of ``radius`` around the origin."""
    ang = 2.0 * math.pi * (index % max(count, 1)) / max(count, 1)
    return (radius * math.sin(ang), height, -radius * math.cos(ang))
