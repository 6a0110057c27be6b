#!/usr/bin/env python3
"""Single-pair registration by direct optimisation of one se(3) vector -- the MI355X
counterpart of the reference demo (code/test_demo_optimized_Lie_Algebra.py:27-143).

Same command line (--data_path --device --seed --label1 --Save_path), same data dict, same
loop: per epoch draw `n_sample_line` lines through both clouds' boxes, move the source with
Reconstruction_point, evaluate the intersected-line loss, Adam step (lr 2e-2, halved whenever
epoch % 1000 == 0 -- including epoch 0), Chamfer monitor, and every 10th epoch write
`<epoch>.obj`, `target.obj`, `model.pkl`, `<epoch>_transform.txt` under Save_path.

Differences, all additive:
  * OBJ files are read/written by a ten-line parser (no libigl), scalars go to
    `<Save_path>/log/scalars.csv` (tensorboard's SummaryWriter is used when importable);
  * `--synthetic N` makes a seeded pair instead of reading OBJ files;
  * `--graph` replays the whole step (exp map, fused transform + loss, backward, Adam,
    Chamfer) as one captured hipGraph: one host call per epoch and no host synchronisation
    except for the progress line every `--print_every` epochs;
  * `--device_rng` draws the candidate lines' uniforms on the GPU (the reference's CPU stream
    costs more host time per epoch than the whole optimisation step);
  * `lines_fn(epoch, moved_src)` lets a test inject recorded lines.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from loss import cal_loss_intersection_batch_whole_median_pts_lines  # noqa: E402
from loss import Random_uniform_distribution_lines_batch_efficient_resample  # noqa: E402
from loss import Reconstruction_point, Sample_neighs, chamfer_dist, generate_bbox  # noqa: E402
from rrl_hip import ops as _ops  # noqa: E402
from rrl_hip.graph import GraphedStep  # noqa: E402


def adjust_learning_rate(optimizer, epoch, lr):
    """Halve `lr` when epoch % 1000 == 0 (the reference does so at epoch 0 too) and install it."""
    if epoch % 1000 == 0:
        lr *= 0.5
    for group in optimizer.param_groups:
        group['lr'] = lr
    return lr


# ------------------------------------------------------------------------------- file I/O
def read_obj_vertices(path):
    """(V, 3) float32 vertex rows of a Wavefront OBJ (the demo ignores the faces)."""
    rows = []
    with open(path) as fh:
        for ln in fh:
            if ln.startswith('v '):
                rows.append([float(x) for x in ln.split()[1:4]])
    if not rows:
        raise ValueError(f"{path}: no vertices")
    return np.asarray(rows, np.float32)


def write_obj(path, vertices):
    """Vertices plus the single degenerate face `igl.write_obj(path, V, zeros((1, 3)))` emits."""
    with open(path, 'w') as fh:
        for v in np.asarray(vertices):
            fh.write(f"v {v[0]:.9g} {v[1]:.9g} {v[2]:.9g}\n")
        fh.write("f 1 1 1\n")


class ScalarLog:
    """`add_scalar(tag, value, step)` into <log_dir>/scalars.csv (SummaryWriter stand-in)."""

    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.fh = open(os.path.join(log_dir, 'scalars.csv'), 'w')
        self.fh.write("step,tag,value\n")

    def add_scalar(self, tag, value, step):
        self.fh.write(f"{step},{tag},{value:.9g}\n")

    def close(self):
        self.fh.close()


def make_writer(log_dir):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir=log_dir)
    except Exception:  # tensorboard is optional
        return ScalarLog(log_dir)


def save_checkpoint(Save_path, epoch, moved, target, model):
    write_obj(os.path.join(Save_path, f"{epoch}.obj"), moved.detach().cpu().numpy())
    write_obj(os.path.join(Save_path, "target.obj"), target.detach().cpu().numpy())
    torch.save(model.state_dict(), os.path.join(Save_path, 'model.pkl'))
    R, T = model.Transform()
    transforms = np.ones([3, 4])
    transforms[:3, :3] = R.detach().cpu().numpy()
    transforms[:3, 3] = T.detach().cpu().numpy()
    np.savetxt(os.path.join(Save_path, f"{epoch}_transform.txt"), transforms)


# ------------------------------------------------------------------------------ the loop
class _GatedAdam:
    """torch.optim.Adam's update (betas 0.9/0.999, eps 1e-8, no weight decay) with device-side
    scalars so a captured graph can (a) take a new lr per replay and (b) skip the whole update
    when the loss was empty, as the reference's `if loss_di is not None` does.  One launch
    (rrl_adam_gated); written as torch ops it was ~25 kernels on 6 floats."""

    def __init__(self, param, lr):
        self.p = param
        dev = param.device
        self.m = torch.zeros_like(param)
        self.v = torch.zeros_like(param)
        self.step = torch.zeros(1, device=dev)
        self.lr = torch.full((1,), lr, device=dev)
        self.param_groups = [{'lr': lr}]

    def set_lr(self):
        self.lr.fill_(self.param_groups[0]['lr'])

    def update(self, grad, gate):
        """gate: int32 tensor whose first element > 0 enables the step (the loss's bucket count)."""
        _ops.adam_gated(self.p.data, grad, self.m, self.v, self.step, self.lr, gate)


def _same_point_set(points, tri):
    """Are the first points of the pseudo-triangles `tri` (.., 9) exactly the points `points` (.., 3), in any order?
    (Then the Chamfer distance of the loss evaluation's sorted P0 records is the monitor's value.)"""
    p = points.detach().reshape(-1, 3).cpu().numpy()
    q = tri.detach().reshape(-1, 9)[:, :3].cpu().numpy()
    if p.shape != q.shape:
        return False
    return bool(np.array_equal(p[np.lexsort(p.T[::-1])], q[np.lexsort(q.T[::-1])]))


def _default_lines(radius, centers, n_sample_line, target, device, device_rng=False):
    box2 = []  # the target does not move: its AABB is computed once

    def draw(epoch, moved, out=None, box1=None):
        if not box2:
            box2.append(_ops.aabb(target.view(1, -1, 3)))
        lines = Random_uniform_distribution_lines_batch_efficient_resample(
            radius.reshape(1, 1), centers.reshape(1, -1), n_sample_line, moved.view(1, -1, 3),
            target.view(1, -1, 3), device, device_rng=device_rng, out=out, box2=box2[0], box1=box1)
        return lines.detach().view(-1, 6)
    draw.sampler = (radius, centers, n_sample_line, target, box2)  # for the one-call epoch (rrl_demo_epoch)
    return draw


def test_one_case(data, Save_path, writer=None, n_epoch=1000, n_sample_line=20000, device='cuda:0',
                  *, lines_fn=None, graph=False, save_every=10, print_every=1, model=None,
                  device_rng=False):
    """code/test_demo_optimized_Lie_Algebra.py:27-100.  Returns the per-epoch history
    [(epoch, loss or None, chamfer or None)] (floats; in graph mode filled at the end from
    device buffers) and the Reconstruction_point module."""
    dev = torch.device(device)
    bounding_box = data['bounding_box'].to(dev)
    src = data['vertics1_tensor'].to(dev)
    tar = data['vertics2_tensor'].to(dev)
    src_nb = data['vertics1_faces_tensor'].to(dev)
    tar_tri = data['vertics2_faces_tensor'].to(dev).reshape(1, -1, 9)
    centers = data['centers'].to(dev)
    os.makedirs(Save_path, exist_ok=True)
    Reconstruction = (model or Reconstruction_point()).to(dev)
    radius = (bounding_box[0, :] - bounding_box[-1, :]).norm(p=2).reshape(1)
    draw = lines_fn or _default_lines(radius, centers, n_sample_line, tar, dev, device_rng)
    if graph:
        return _run_graphed(Reconstruction, draw, src, src_nb, tar, tar_tri, Save_path, writer,
                            n_epoch, save_every, print_every,
                            draw_in_graph=device_rng and lines_fn is None), Reconstruction

    optimize = torch.optim.Adam(Reconstruction.parameters(), lr=2e-2)
    moved = src
    history = []
    for epoch in range(n_epoch):
        lines = draw(epoch, moved.detach())
        adjust_learning_rate(optimize, epoch, optimize.param_groups[0]['lr'])
        moved, moved_tri = Reconstruction(src, src_nb)
        loss_di = cal_loss_intersection_batch_whole_median_pts_lines(
            1, 1, 5, 5, moved_tri.reshape(1, -1, 9), tar_tri, lines.reshape(1, -1, 6), dev)
        if loss_di is None:
            history.append((epoch, None, None))
            continue
        optimize.zero_grad()
        loss_di.backward()
        optimize.step()
        loss_cf = chamfer_dist(moved.reshape(1, -1, 3), tar.reshape(1, -1, 3))
        di, cf = torch.stack([loss_di.detach().reshape(()), loss_cf.detach()]).tolist()
        history.append((epoch, di, cf))
        if print_every and epoch % print_every == 0:
            print("\033[34mthis is the chamfer loss:{:4f}, loss_intersection{:4f}\033[0m".format(cf, di))
        if save_every and epoch % save_every == 0:
            save_checkpoint(Save_path, epoch, moved, tar, Reconstruction)
        if writer is not None:
            writer.add_scalar('./loss/chamfer_loss', cf, epoch)
            writer.add_scalar('./loss/intersection_loss', di, epoch)
    return history, Reconstruction


def _run_graphed(model, draw, src, src_nb, tar, tar_tri, Save_path, writer, n_epoch, save_every,
                 print_every, draw_in_graph=False):
    """draw_in_graph: the line sampler is part of the captured step (possible when its uniforms come
    from the GPU generator): ONE graph launch per epoch and nothing else on the stream -- every eager
    kernel between two graph launches costs more than it computes."""
    dev = src.device
    xi = model.parameters_
    opt = _GatedAdam(xi, 2e-2)
    src_tri = src_nb.reshape(1, -1, 9).contiguous()
    lines0 = draw(0, src)
    lines = torch.empty_like(lines0.reshape(1, -1, 6))
    moved = src.clone().reshape(1, -1, 3)
    WARM = 2  # eager warm-up runs of the step before its capture
    trace = torch.zeros(n_epoch + WARM, 3, device=dev)  # loss, chamfer, valid per epoch (+ warm-up scratch rows)
    row = torch.zeros(3, device=dev)
    slot = torch.zeros(1, dtype=torch.long, device=dev)  # epoch counter on the device
    scratch_row = torch.zeros(1, 3, device=dev)  # log_row's table when the host keeps the trace

    ones = torch.ones(1, device=dev)
    src_pts = src.reshape(1, -1, 3).contiguous()
    tar_pts = tar.reshape(1, -1, 3).contiguous()

    # No autograd inside the captured step (round 3): the pose kernels and the fused op are called directly on
    # preallocated buffers (autograd's AccumulateGrad / view bookkeeping added two copy kernels per epoch, ~9 us of the
    # 131), and the Chamfer monitor walks the clouds the loss evaluation just sorted when the point sets ARE the
    # triangles' first points (Sample_neighs keeps every point when the cloud has <= 5000: checked once, on the host).
    Rb = torch.empty(1, 3, 3, device=dev)
    Tb = torch.empty(1, 3, device=dev)
    box1 = _ops.aabb(moved)  # the moved cloud's AABB, kept current by the rigid-apply launch
    n_lines = lines.shape[1]
    reg = _ops.RegistrationStep(src_tri, tar_tri, n_lines, transpose_r=False)
    monitor_from_state = _same_point_set(src, src_tri) and _same_point_set(tar, tar_tri) and \
        max(src_tri.shape[1], tar_tri.shape[1]) <= 65536
    P = _ops._p
    _ops._run(dev, "rrl_se3_exp", P(xi.data), P(Rb), P(Tb), 1)  # == model.Transform() for the first epoch

    # With the sampler in the graph and the point sets == the triangles' first points (round 4) nothing needs the moved
    # POINTS inside the epoch: the sampler wants only their AABB -- the pose launch takes it from the partial rows the
    # loss step's records launch just wrote (APART: the moved first points) -- and checkpoints read the moved first
    # points from the step's TRI1 field.  One launch (rigid apply + AABB) less per epoch.
    box_from_step = draw_in_graph and monitor_from_state
    arows = None
    if box_from_step:
        arows = reg.st.apart[0, 0, :(src_tri.shape[1] + 255) // 256]  # a view of the workspace: rows of cloud 1, sample 0
        assert arows.is_contiguous()

    def step():
        # the whole epoch in 9 launches (10 without box_from_step): sampler (2), the loss's 4 (prepared build, kept target)
        # + 1, [rigid apply + AABB], Chamfer, and the pose step (exp-map backward, Adam, the NEXT epoch's exp map ==
        # model.Transform(), the log row, the next sampler box).  (Rb, Tb) always hold exp(xi): the pose launch refreshes
        # them right after it moves xi.
        if draw_in_graph:  # lines from the previous epoch's moved source, like the reference loop
            draw(0, moved.reshape(-1, 3), out=lines, box1=box1)
        loss, gR, gt, _, info = reg(Rb, Tb, lines)                  # forward + backward to (dL/dR, dL/dT)
        if not box_from_step:
            _ops.rigid_apply_aabb_into(src_pts, Rb, Tb, moved, box1)
        cf = _ops.chamfer_from_state(reg.st) if monitor_from_state else _ops.chamfer(moved, tar_pts)
        # skipped on the device when no bucket is populated; loss, Chamfer, valid -> row (and, with the sampler in
        # the graph, the trace table)
        _ops.se3_adam_step(xi.data, gR, gt, opt.m, opt.v, opt.step, opt.lr, info, Rb, Tb, loss=loss,
                           value=cf.reshape(1), table=trace if draw_in_graph else scratch_row, cursor=slot, row=row,
                           aabb_rows=arows, box=box1.reshape(-1) if box_from_step else None)
        return row

    # ONE C call per epoch (round 4, include/rrl.h rrl_demo_epoch: sampler -> fused step -> Chamfer from the step's state ->
    # pose step, issued back to back) instead of a hipGraph replay of the same launches: a replay costs ~8 us + ~1.5 us per
    # node on this stack (tools/attic/graph_node_cost.py), the call ~5 us of host time.  Same launches, same results.
    # RRL_DEMO_ISSUE=graph keeps the replay.
    epoch_call = None
    if box_from_step and hasattr(draw, "sampler") and reg.prepared and os.environ.get("RRL_DEMO_ISSUE", "call") != "graph":
        epoch_call = _one_call_epoch(draw, reg, lines, box1, xi, opt, Rb, Tb, trace, slot, row, n_epoch + WARM)
        slot.zero_()

    lr = 2e-2
    stepper = None
    for epoch in range(n_epoch):
        if not draw_in_graph:
            lines.copy_((lines0 if epoch == 0 else draw(epoch, moved.reshape(-1, 3))).reshape(1, -1, 6))
        new_lr = adjust_learning_rate(opt, epoch, lr)
        if new_lr != lr or epoch == 0:
            opt.set_lr()
        lr = new_lr
        if epoch_call is not None:
            epoch_call(epoch)
        elif stepper is None:
            # the warm-up runs inside GraphedStep must not move the state: snapshot, restore
            state = (xi.data, opt.m, opt.v, opt.step, moved, slot, Rb, Tb, box1)
            keep = [t.clone() for t in state]
            slot.fill_(n_epoch)  # warm-up rows land in the scratch rows
            stepper = GraphedStep(step, warmup=WARM)
            for t, k in zip(state, keep):
                t.copy_(k)
        if epoch_call is None:
            out = stepper()
            if not draw_in_graph:
                trace[epoch].copy_(out)
        if print_every and epoch % print_every == 0:
            di, cf, ok = trace[epoch].tolist()
            if ok:
                print("\033[34mthis is the chamfer loss:{:4f}, loss_intersection{:4f}\033[0m".format(cf, di))
        if save_every and epoch % save_every == 0:
            pts = reg.st.tri1t[0, :, :3] if box_from_step else moved.reshape(-1, 3)  # the moved first points ARE the points
            save_checkpoint(Save_path, epoch, pts, tar, model)
    history = []
    for epoch, (di, cf, ok) in enumerate(trace[:n_epoch].tolist()):
        history.append((epoch, di, cf) if ok else (epoch, None, None))
        if ok and writer is not None:
            writer.add_scalar('./loss/chamfer_loss', cf, epoch)
            writer.add_scalar('./loss/intersection_loss', di, epoch)
    return history


def _one_call_epoch(draw, reg, lines, box1, xi, opt, Rb, Tb, trace, slot, row, table_rows):
    """The epoch as ONE C call (rrl_demo_epoch).  Returns epoch_call(epoch); all buffers are the captured path's own."""
    import ctypes
    from rrl_hip import _lib
    radius, centers, n_lines, target, box2 = draw.sampler
    dev = lines.device
    if not box2:
        box2.append(_ops.aabb(target.view(1, -1, 3)))
    B, N, M, L = reg.dims
    rounds = 10
    keep = dict(
        radius=radius.reshape(1).to(dev, torch.float32).contiguous(), centers=centers.reshape(1, 3).to(dev, torch.float32).contiguous(),
        filled=torch.empty(1, dtype=torch.int32, device=dev),
        tiles=torch.empty(rounds * ((L + 1023) // 1024) * 32, dtype=torch.int32, device=dev),
        rng=_ops.sampler_rng(dev), cham_ws=torch.empty(int(_lib.load().rrl_chamfer_workspace_bytes(1, N, M)), dtype=torch.uint8, device=dev),
        bx=torch.empty(1, N, dtype=torch.int64, device=dev), by=torch.empty(1, M, dtype=torch.int64, device=dev),
        cf=torch.empty(1, device=dev), gacc=reg.st.gacc, box2=box2[0].contiguous())
    P = _ops._p
    a = _lib.DemoEpochArgs()
    a.struct_bytes, a.N, a.M, a.L, a.rounds, a.transpose_r = ctypes.sizeof(_lib.DemoEpochArgs), N, M, L, rounds, reg.tr
    a.rng_state, a.radius, a.centers, a.box1, a.box2 = P(keep["rng"]), P(keep["radius"]), P(keep["centers"]), P(box1), P(keep["box2"])
    a.lines, a.filled, a.tile_counts = P(lines), P(keep["filled"]), P(keep["tiles"])
    a.src_tri, a.tar_tri, a.R, a.T = P(reg.src), P(reg.tar), P(Rb), P(Tb)
    a.ws, a.ws_bytes, a.loss, a.grad_loss = P(reg.st.ws), reg.st.nbytes, P(reg.st.loss), P(reg.ones)
    a.gR, a.gt = P(reg.gR), P(reg.gt)
    a.cham_ws, a.cham_ws_bytes, a.best_x, a.best_y, a.cham_value = P(keep["cham_ws"]), keep["cham_ws"].numel(), P(keep["bx"]), P(keep["by"]), P(keep["cf"])
    a.xi, a.m, a.v, a.adam_state, a.lr = P(xi.data), P(opt.m), P(opt.v), P(opt.step), P(opt.lr)
    a.b1, a.b2, a.eps = 0.9, 0.999, 1e-8
    a.table, a.cursor, a.table_rows, a.row = P(trace), P(slot), table_rows, P(row)
    pipe = ctypes.c_int32(0)  # (host word: the next epoch's count pass rides in this epoch's per-line launch; rrl.h)
    a.pipeline = ctypes.addressof(pipe)
    keep["pipe"] = pipe
    lib = _lib.load()
    aref = ctypes.byref(a)
    opts_first, opts_kept = ctypes.addressof(reg._opts), ctypes.addressof(reg._opts_kept)

    def epoch_call(epoch):
        a.opts = opts_first if epoch == 0 else opts_kept  # the target's records are built once, then kept
        _lib.check(lib.rrl_demo_epoch(aref, _ops._stream(dev)), "rrl_demo_epoch")
    epoch_call.keep = (keep, a)
    return epoch_call


# -------------------------------------------------------------------------------- driver
def build_data(vertics1, vertics2, device):
    """code/test_demo_optimized_Lie_Algebra.py:113-141: pseudo-triangles, centring, the dict."""
    vertics1 = np.asarray(vertics1, np.float32)
    vertics2 = np.asarray(vertics2, np.float32)
    faces_neighs1 = Sample_neighs(vertics1, device=device)
    faces_neighs2 = Sample_neighs(vertics2, device=device)
    center1 = vertics1.mean(0)[np.newaxis, :]
    center2 = vertics2.mean(0)[np.newaxis, :]
    vertics1, vertics2 = vertics1 - center1, vertics2 - center2
    faces_neighs1, faces_neighs2 = faces_neighs1 - center1, faces_neighs2 - center2
    v1 = torch.from_numpy(vertics1.astype(np.float32)).to(device)
    v2 = torch.from_numpy(vertics2.astype(np.float32)).to(device)
    return {
        'bounding_box': generate_bbox(v2[None])[0].to(device),
        'vertics1_tensor': v1,
        'vertics2_tensor': v2,
        'vertics1_faces_tensor': torch.from_numpy(faces_neighs1.astype(np.float32)).to(device).reshape(1, -1, 3),
        'vertics2_faces_tensor': torch.from_numpy(faces_neighs2.astype(np.float32)).to(device),
        'centers': v2.mean(0),
    }


def main(args):
    torch.set_default_dtype(torch.float32)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    _ops.require_gpu()  # no CPU path: fail loudly without a GPU / the HIP library
    device = torch.device(args.device)
    if device.type != 'cuda':
        raise ValueError("--device must be a cuda device (the loss has no CPU implementation)")
    torch.cuda.set_device(device)
    if args.synthetic:
        from rrl_hip import synth
        pr = synth.make_pair(args.seed, args.synthetic, args.synthetic)
        vertics1, vertics2 = pr['src'], pr['tar']
        diag = getattr(args, 'synthetic_diag', 0.0)
        if diag:  # the data scale of the reference's sample_data/challenge_data: AABB diagonal 11.7
            sc = np.float32(diag / (2.0 * float(pr['radius'])))
            vertics1, vertics2 = vertics1 * sc, vertics2 * sc
    else:
        vertics1 = read_obj_vertices(os.path.join(args.data_path, args.label1 + "_src_sample.obj"))
        vertics2 = read_obj_vertices(os.path.join(args.data_path, args.label1 + "_tar_sample.obj"))
    data = build_data(vertics1, vertics2, device)
    writer = make_writer(os.path.join(args.Save_path, 'log'))
    history, model = test_one_case(data, args.Save_path, writer=writer, n_epoch=args.n_epoch,
                                   n_sample_line=args.n_sample_line, device=device, graph=args.graph,
                                   print_every=args.print_every, save_every=getattr(args, 'save_every', 10),
                                   device_rng=getattr(args, 'device_rng', False))
    writer.close()
    return history, model


if __name__ == "__main__":
    print("Test our case!")
    parser = argparse.ArgumentParser()
    parser.add_argument('--data_path', type=str, default="./sample_data/challenge_data")
    parser.add_argument('--device', type=str, default='cuda:0')
    parser.add_argument('--seed', type=int, default=123)
    parser.add_argument('--label1', type=str, default=None)
    parser.add_argument('--Save_path', type=str, default="./Results")
    parser.add_argument('--n_epoch', type=int, default=1000)
    parser.add_argument('--n_sample_line', type=int, default=20000)
    parser.add_argument('--synthetic', type=int, default=0, metavar='N',
                        help="use a seeded synthetic pair of N points instead of OBJ files")
    parser.add_argument('--synthetic_diag', type=float, default=0.0, metavar='D',
                        help="rescale the synthetic pair to an AABB diagonal of D (the reference's sample data: 11.7)")
    parser.add_argument('--graph', action='store_true', help="replay the step as one hipGraph")
    parser.add_argument('--print_every', type=int, default=1)
    parser.add_argument('--save_every', type=int, default=10, help="0: no OBJ / transform files")
    parser.add_argument('--device_rng', action='store_true',
                        help="draw the lines' uniforms on the GPU instead of torch's CPU generator")
    args = parser.parse_args()
    save_root = args.Save_path
    if args.synthetic:
        labels = [args.label1 or 'synthetic']
    elif args.label1 is not None:
        labels = [args.label1]
    else:  # the reference loops over the pairs '0'..'4' of the data directory
        labels = sorted({f.split('_')[0] for f in os.listdir(args.data_path) if f.endswith('_src_sample.obj')})
    for label1 in labels:
        args.label1 = label1
        args.Save_path = os.path.join(save_root, label1 + "challenge_new_1")
        os.makedirs(save_root, exist_ok=True)
        main(args)
