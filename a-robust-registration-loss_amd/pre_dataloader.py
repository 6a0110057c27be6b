"""Pair dataset of the trainers (reference: code/exps_deep_learning/pre_dataloader.py:28-181),
readable without libigl / h5py so the RPM / DCP / FMR loss fragments (rrl_hip.callsites) can be
fed from disk on the GPU box.  Host-side numpy only; nothing here touches the GPU.

On-disk layout of one pair `<dir>/{src,tar}_sample_<mesh>_<view>`:
    src_sample_m_v.obj, tar_sample_m_v.obj            vertex-only OBJ ("v x y z")
    src_sample_normals_m_v.obj, tar_sample_normals…   normals stored as OBJ vertices
    src_sample_m_v_neigh.bin, tar_sample_m_v_neigh.bin raw float32 (3N, 3): rows [p, nn1, nn2] per
                                                       point = Sample_neighs output (loss.py:473-485)
    transform_m_v.bin                                  raw float64 (3, 4) ground truth [A | b]
Path derivation follows the reference literally (first occurrence of "sample" -> "sample_normals",
".obj" -> "_neigh.bin", "tar_sample" -> "transform"), so directories must not contain "sample".

Item dict (float32 numpy; the default collate_fn stacks it into the dict the trainers index):
    points_src_sample (N,3), points_tar_sample (M,3)   centred clouds
    points_based_neighs_src (3N,3), _tar (3M,3)        centred pseudo-triangle vertices
    normals_src, normals_tar, tar_box (8,3) AABB corners (corner 0 = max, 7 = min: libigl order),
    centers (3,) mean of the centred target, R, T, R_inv, T_inv, igt (4,4)
with R = A, T = b - c_tar + c_src A (row-vector convention: tar ~ src @ R + T, demo header
test_demo_optimized_Lie_Algebra.py:24-25), R_inv = A, T_inv = -A T, igt = [[A, -A T], [0, 1]].
NB `ndarray.transpose(0, 1)` is the identity permutation, so the reference's many
`.transpose(0, 1)` calls change nothing -- reproduced by doing nothing.  DCP_True transposes the
clouds to channel-first (3, N) and R, R_inv, igt[:3,:3]; FMR_True truncates both clouds to the
shorter length.
"""
import os

import numpy as np

try:  # torch is only needed for the Dataset base class / loaders
    import torch
    from torch.utils.data import Dataset
except Exception:  # pragma: no cover
    torch = None
    Dataset = object


def M(axis, theta):
    """Rotation matrix exp([axis/|axis| * theta]_x) (pre_dataloader.py:17-18, via Rodrigues)."""
    a = np.asarray(axis, np.float64).reshape(3)
    a = a / np.linalg.norm(a)
    th = float(np.asarray(theta).reshape(-1)[0])
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])  # np.cross(I, a): rows e_i x a
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def kd_order(points):
    """Spatial order of a cloud for the prepared build of the loss (include/rrl.h rrl_opts.order1 / order2; the
    host-side twin of rrl_cloud_order): int32 (64 ceil(n / 64),) sorted position -> point index, positions >= n hold 0.
    Positions [0, P), P = the power of two >= n, form an implicit binary tree of aligned windows; every window sorts
    its points along the longest axis of their bounding box (stable, ties by index), so its lower half of positions
    receives the points below the median plane -- aligned runs of 64 / 16 / 8 positions are compact k-d cells.  ANY
    permutation gives the same loss bits; this one only makes the culled scan's tree nodes tight.  points (n, 3)."""
    pts = np.asarray(points, np.float64).reshape(-1, 3)
    n = len(pts)
    npad = (n + 63) // 64 * 64
    out = np.zeros(npad, np.int32)
    if n == 0:
        return out
    P = 64
    while P < n:
        P *= 2

    def split(idx, S):  # idx: the points of an aligned window of S positions (len(idx) <= S, filled from the left)
        if S <= 8 or len(idx) <= 1:
            return idx
        p = pts[idx]
        ax = int(np.argmax(p.max(0) - p.min(0)))
        idx = idx[np.lexsort((idx, p[:, ax]))]
        h = S // 2
        if len(idx) <= h:
            return split(idx, h)
        return np.concatenate([split(idx[:h], h), split(idx[h:], h)])

    order = split(np.arange(n), P)
    out[:n] = order
    return out


def read_obj_vertices(path):
    """(V, 3) float64 rows of the "v" records, what igl.read_triangle_mesh returns as V."""
    rows = []
    with open(path) as fh:
        for ln in fh:
            if ln.startswith('v '):
                rows.append([float(x) for x in ln.split()[1:4]])
    return np.asarray(rows, np.float64).reshape(-1, 3)


def write_obj_vertices(path, V):
    with open(path, 'w') as fh:
        for v in np.asarray(V):
            fh.write(f"v {v[0]:.9g} {v[1]:.9g} {v[2]:.9g}\n")


def bounding_box(V):
    """8 AABB corners in libigl's order (corner 0 = max, 7 = min; code/loss.py:325-351)."""
    mn, mx = V.min(0), V.max(0)
    pick = np.array([[1, 1, 1], [1, 1, 0], [1, 0, 1], [1, 0, 0], [0, 1, 1], [0, 1, 0], [0, 0, 1],
                     [0, 0, 0]], bool)
    return np.where(pick, mx[None, :], mn[None, :])


def pair_paths(directory, mesh_idx, view_idx):
    """(src .obj, tar .obj) of one pair, named as generate_datasets_* do (pre_dataloader.py:203-210)."""
    tag = f"{mesh_idx}_{view_idx}.obj"
    return os.path.join(directory, "src_sample_" + tag), os.path.join(directory, "tar_sample_" + tag)


def write_pair(directory, mesh_idx, view_idx, src, tar, src_neigh, tar_neigh, transform,
               normals_src=None, normals_tar=None):
    """Writes one pair in the on-disk layout above.  src (N,3), tar (M,3), *_neigh (3n,3) fp32,
    transform (3,4) fp64.  Returns (src_path, tar_path)."""
    os.makedirs(directory, exist_ok=True)
    ps, pt = pair_paths(directory, mesh_idx, view_idx)
    for path, V, nb, nrm in ((ps, src, src_neigh, normals_src), (pt, tar, tar_neigh, normals_tar)):
        write_obj_vertices(path, V)
        write_obj_vertices(path.replace("sample", "sample_normals", 1),
                           nrm if nrm is not None else np.zeros_like(np.asarray(V)))
        np.asarray(nb, np.float32).reshape(-1, 3).tofile(path.replace('.obj', '_neigh.bin', 1))
    np.asarray(transform, np.float64).reshape(3, 4).tofile(
        pt.replace('tar_sample', 'transform', 1).replace('.obj', '.bin', 1))
    return ps, pt


class Dataset_2021_8_29(Dataset):
    """pre_dataloader.py:28-181; see the module docstring for the item dict."""

    def __init__(self, points_files_src_sample, points_files_tar_sample, DCP_True=False, FMR_True=False, with_orders=True):
        self.points_files_src_sample = points_files_src_sample
        self.points_files_tar_sample = points_files_tar_sample
        self.randg = np.random.RandomState(0)
        self.DCP_True = DCP_True
        self.FMR_True = FMR_True
        self.with_orders = with_orders  # add 'order_src' / 'order_tar' (kd_order) to every item

    @staticmethod
    def transform_R_T(points, R, T):
        return points @ R + T

    def random_data(self, data, rotation_range=30):
        """Augmentation of pre_dataloader.py:45-76: one random rotation (numpy global RNG: three
        `rand` for the axis, one for the angle) applied to the whole scene; the source is mapped to
        the target frame, rotated, and mapped back so that (R, T) stay valid.  The reference reads
        the keys 'normals_ref' and writes 'center' -- kept (pass a dict that has them)."""
        R = M(np.random.rand(3) - 0.5, rotation_range * np.pi / 180.0 * (np.random.rand(1) - 0.5)).astype(np.float32)
        T = np.zeros([1, 3], np.float32)
        tar = self.transform_R_T(data['points_tar_sample'], R, T)
        normals_tar = self.transform_R_T(data['normals_ref'], R, 0 * T)
        nb_tar = self.transform_R_T(data['points_based_neighs_tar'].reshape(-1, 3), R, T)
        src = self.transform_R_T(data['points_src_sample'] @ data['R'] + data['T'], R, T)
        normals_src = self.transform_R_T(data['normals_src'] @ data['R'], R, 0 * T)
        nb_src = self.transform_R_T(data['points_based_neighs_src'].reshape(-1, 3) @ data['R'] + data['T'], R, T)
        back = data['R'].transpose(1, 0)
        data['points_src_sample'] = (src - data['T']) @ back
        data['points_based_neighs_src'] = (nb_src - data['T']) @ back
        data['normals_src'] = normals_src @ back
        data['points_tar_sample'] = tar
        data['normals_ref'] = normals_tar
        data['points_based_neighs_tar'] = nb_tar
        data['center'] = data['centers'] @ R + T
        data['tar_box'] = data['tar_box'] @ R + T
        data.pop('p0_rows', None)  # samples and neighbours went through separate matrix products: let the consumer re-check
        return data

    def __getitem__(self, index):
        f_src = self.points_files_src_sample[index]
        f_tar = self.points_files_tar_sample[index]
        V_src = read_obj_vertices(f_src)
        V_tar = read_obj_vertices(f_tar)
        normals_src = read_obj_vertices(f_src.replace("sample", "sample_normals", 1))
        normals_tar = read_obj_vertices(f_tar.replace("sample", "sample_normals", 1))
        nb_src = np.fromfile(f_src.replace('.obj', '_neigh.bin', 1), np.float32).reshape(-1, 3).astype(np.float64)
        nb_tar = np.fromfile(f_tar.replace('.obj', '_neigh.bin', 1), np.float32).reshape(-1, 3).astype(np.float64)
        c_tar, c_src = V_tar.mean(0), V_src.mean(0)
        V_tar = V_tar - c_tar
        tar_box = bounding_box(V_tar).astype(np.float32)
        V_src = V_src - c_src
        nb_src = nb_src - c_src
        nb_tar = nb_tar - c_tar
        gt = np.fromfile(f_tar.replace('tar_sample', 'transform', 1).replace('.obj', '.bin', 1),
                         np.float64).reshape(3, 4)
        rotation = gt[:3, :3]
        translation = gt[:3, 3] - c_tar + c_src @ rotation
        igt = np.eye(4)
        igt[:3, :3] = rotation
        igt[:3, 3] = -rotation @ translation
        r32, t32 = rotation.astype(np.float32), translation.astype(np.float32)
        data = {
            'points_tar_sample': V_tar.astype(np.float32),
            'points_src_sample': V_src.astype(np.float32),
            'normals_tar': normals_tar.astype(np.float32),
            'normals_src': normals_src.astype(np.float32),
            'tar_box': tar_box,
            'centers': V_tar.mean(0).astype(np.float32),
            'R': r32,
            'T': t32,
            'R_inv': r32,
            'T_inv': -r32 @ t32,
            'points_based_neighs_src': nb_src.astype(np.float32),
            'points_based_neighs_tar': nb_tar.astype(np.float32),
            'igt': igt.astype(np.float32),
        }
        # beyond the reference's keys: do the point samples equal the first points of the pseudo-triangles (rows
        # 0, 3, 6, ... of *_neigh.bin -- what Sample_neighs writes)?  rrl_hip.callsites then takes the Chamfer monitor
        # from the loss evaluation's own sorted clouds instead of sorting the samples again.
        data['p0_rows'] = np.bool_(
            data['points_based_neighs_src'].shape[0] == 3 * data['points_src_sample'].shape[0]
            and data['points_based_neighs_tar'].shape[0] == 3 * data['points_tar_sample'].shape[0]
            and np.array_equal(data['points_based_neighs_src'].reshape(-1, 9)[:, :3], data['points_src_sample'])
            and np.array_equal(data['points_based_neighs_tar'].reshape(-1, 9)[:, :3], data['points_tar_sample']))
        # beyond the reference's keys (round 4): the spatial ORDER of both clouds of pseudo-triangles, computed once per item
        # here on the host (kd_order below: any permutation gives the same loss, this one makes the culled scan fast; a
        # rigid motion -- the predicted pose, the augmentation -- preserves it).  rrl_hip.callsites hands them to the
        # fused op (rrl_opts.order1 / order2): the per-step cell sort disappears for every pose of the batch.
        if self.with_orders:
            data['order_src'] = kd_order(data['points_based_neighs_src'].reshape(-1, 9)[:, :3])
            data['order_tar'] = kd_order(data['points_based_neighs_tar'].reshape(-1, 9)[:, :3])
        if self.DCP_True is True:  # channel-first clouds, transposed rotations
            for k in ('points_tar_sample', 'points_src_sample', 'points_based_neighs_src',
                      'points_based_neighs_tar', 'R', 'R_inv'):
                data[k] = data[k].transpose(1, 0)
            data['igt'][:3, :3] = data['igt'][:3, :3].transpose(1, 0).copy()
        if self.FMR_True is True:  # equal-length clouds
            n = min(data['points_src_sample'].shape[0], data['points_tar_sample'].shape[0])
            data['points_tar_sample'] = data['points_tar_sample'][:n, ]
            data['points_src_sample'] = data['points_src_sample'][:n, ]
        return data

    def __len__(self):
        return len(self.points_files_tar_sample)


def list_pairs(data_path, meshes, views):
    """File lists as generate_datasets_{human,airplane,real} build them (pre_dataloader.py:194-210)."""
    src, tar = [], []
    for m in meshes:
        for v in views:
            a, b = pair_paths(data_path, m, v)
            src.append(a)
            tar.append(b)
    return src, tar


def make_loaders(data_path, meshes, views, batch_size=4, n_test=2, DCP=False, FMR=False, num_workers=0):
    """Train / test DataLoaders over the pairs found under data_path -- the role of
    generate_datasets_human/airplane/real (pre_dataloader.py:190-330), with the hard-wired
    /data1/... roots and index ranges turned into arguments."""
    src, tar = list_pairs(data_path, meshes, views)
    keep = [i for i in range(len(src)) if os.path.exists(src[i]) and os.path.exists(tar[i])]
    src, tar = [src[i] for i in keep], [tar[i] for i in keep]
    n_train = max(len(src) - n_test, 1)
    train = Dataset_2021_8_29(src[:n_train], tar[:n_train], DCP_True=DCP, FMR_True=FMR)
    test = Dataset_2021_8_29(src[n_train:] or src[:1], tar[n_train:] or tar[:1], DCP_True=DCP, FMR_True=FMR)
    mk = torch.utils.data.DataLoader
    return (mk(train, batch_size=batch_size, shuffle=True, num_workers=num_workers, drop_last=True),
            mk(test, batch_size=1, shuffle=False, num_workers=num_workers, drop_last=True))


def synthesize_dataset(directory, n_pairs, n_points=512, seed=0):
    """Writes n_pairs seeded synthetic pairs (rrl_hip.synth) in the on-disk layout; pseudo-triangles
    from a host k-d tree.  For smoke runs of the trainers' fragments without the original data."""
    from rrl_hip import synth
    out = []
    for i in range(n_pairs):
        pr = synth.make_pair(seed + i, n_points, n_points)
        A = np.eye(3)
        gt = np.concatenate([A, np.zeros((3, 1))], 1)
        out.append(write_pair(directory, i, 0, pr["src"], pr["tar"], pr["src_tri"].reshape(-1, 3),
                              pr["tar_tri"].reshape(-1, 3), gt))
    return out
