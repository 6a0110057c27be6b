"""The handful of `utils` names the reference's callers import next to the loss
(SURVEY.md §8b): transform_point_cloud, npmat2euler, dict_all_to_device, makefacevertices,
Dict2txt_json, mkdir_ifnotexists.  Everything else in the reference's code/utils.py is data
generation (openmesh / trimesh / cv2 / igl) and out of scope."""
import json
import os

import numpy as np
import torch

from rrl_hip import ops as _ops


def dict_all_to_device(tensor_dict, device):
    """Moves every tensor value of the dict to `device`, in place (code/utils.py:12-16)."""
    for key, value in tensor_dict.items():
        if isinstance(value, torch.Tensor):
            tensor_dict[key] = value.to(device)


def Dict2txt_json(file_path, dict, file_type='txt'):  # noqa: A002 (reference's argument name)
    """Dumps a dict as `key:value` lines or as one JSON document (code/utils.py:19-30).
    The reference closes the txt file inside the loop, i.e. only the first item is written
    before a ValueError; here all items are written."""
    with open(file_path, 'w') as fh:
        if file_type == 'txt':
            for key, value in dict.items():
                fh.write(f"{key}:{value}\n")
        else:
            fh.write(json.dumps(dict))


def mkdir_ifnotexists(directory):
    if not os.path.exists(directory):
        os.mkdir(directory)


def quat2mat(quat):
    """(B, 4) quaternions stored x, y, z, w -> (B, 3, 3) (code/utils.py:52-67)."""
    x, y, z, w = quat[:, 0], quat[:, 1], quat[:, 2], quat[:, 3]
    xx, yy, zz, ww = x * x, y * y, z * z, w * w
    xy, xz, yz, wx, wy, wz = x * y, x * z, y * z, w * x, w * y, w * z
    rows = [ww + xx - yy - zz, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
            2 * wz + 2 * xy, ww - xx + yy - zz, 2 * yz - 2 * wx,
            2 * xz - 2 * wy, 2 * wx + 2 * yz, ww - xx - yy + zz]
    return torch.stack(rows, dim=1).reshape(-1, 3, 3)


def transform_point_cloud(point_cloud, rotation, translation):
    """Channel-first rigid transform R @ x + t for (B, 3, N) clouds (code/utils.py:32-37; the DCP
    trainer's layout).  `rotation` is (B, 3, 3) or (B, 4) quaternions.  HIP rigid-apply kernel."""
    rot = quat2mat(rotation) if rotation.dim() == 2 else rotation
    return _ops.rigid_apply(point_cloud, rot, translation, transpose_r=True, channel_first=True)


def transform_point_cloud_point_based(point_cloud, rotation, translation):
    """Channel-last variant: x @ R^T + t for (B, N, 3) clouds (code/utils.py:41-49)."""
    rot = quat2mat(rotation) if rotation.dim() == 2 else rotation
    return _ops.rigid_apply(point_cloud, rot, translation.reshape(-1, 3), transpose_r=True)


def npmat2euler(mats, seq='zyx'):
    """(B, 3, 3) numpy rotation matrices -> Euler angles in degrees (code/utils.py:70-75)."""
    from scipy.spatial.transform import Rotation
    return np.asarray([Rotation.from_matrix(m).as_euler(seq, degrees=True) for m in mats],
                      dtype='float32')


def makefacevertices(vertices, faces):
    """(B, V, 3) vertices + (B, F, 3) indices -> (B, F, 9) per-face vertex rows
    (code/utils.py:90-105).  A gather: plain torch indexing."""
    if vertices.dim() == 2:
        vertices = vertices.unsqueeze(0)
    if faces.dim() == 2:
        faces = faces.unsqueeze(0)
    idx = faces.long().unsqueeze(-1).expand(-1, -1, -1, 3)  # (B, F, 3, 3)
    return torch.gather(vertices.unsqueeze(1).expand(-1, faces.shape[1], -1, -1), 2, idx).reshape(
        faces.shape[0], faces.shape[1], 9)
