#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference's code/loss.py is imported unmodified; `openmesh`, `trimesh`,
`cv2`, `igl` are absent here and unused by the path, so empty stub modules are
registered first.  Only inputs and the reference's
outputs are stored (npz); no reference source is copied.  The fixtures record
the torch version and the minimum decision margin of the label test so the
parity tests are provably non-borderline.
"""
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/code"

for m in ("openmesh", "trimesh", "cv2", "igl"):
    sys.modules.setdefault(m, types.ModuleType(m))
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")
import torch  # noqa: E402
import loss as RL  # noqa: E402  (the reference)

sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
from rrl_hip import synth  # noqa: E402

torch.set_default_dtype(torch.float32)
META = dict(torch_version=torch.__version__, cpu_capability=torch.backends.cpu.get_cpu_capability())


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def ref_scan(tri, lines):
    """Reference dense scan of one cloud -> counts, hit list, weights at hits, margin."""
    pts, norm_d, label = RL.cal_intersection_batch2_points_with_line(t(tri)[None], t(lines)[None])
    label = label[0].numpy()
    norm_d = norm_d.numpy()  # (L, N, 3)
    count = label.sum(1).astype(np.int32)
    li, fi = np.nonzero(label)
    return dict(count=count, hit_line=li.astype(np.int32), hit_tri=fi.astype(np.int32),
                hit_w=norm_d[li, fi].astype(np.float32), label=label)


def margin(tri, lines):
    """min over (line, triangle) of |max_k d / thr - 1| in float64 (decision margin)."""
    P = tri.reshape(-1, 3, 3).astype(np.float64)
    e = (np.linalg.norm(P[:, 1] - P[:, 0], axis=1) + np.linalg.norm(P[:, 2] - P[:, 0], axis=1)
         + np.linalg.norm(P[:, 1] - P[:, 2], axis=1)) / 3
    thr = e * 1.731 / 2
    best = np.inf
    ln = lines.astype(np.float64)
    for s in range(0, ln.shape[0], 256):
        u, x0 = ln[s:s + 256, None, None, :3], ln[s:s + 256, None, None, 3:]
        a = P[None] - x0
        d = np.sqrt(np.maximum((a * a).sum(-1) - ((a * u).sum(-1)) ** 2 + 2e-4, 0))
        best = min(best, np.abs(d.max(-1) / thr[None] - 1).min())
    return float(best)


def ref_loss_case(tri1, tri2, lines, rng):
    p1 = t(tri1)[None].clone().requires_grad_(True)
    p2 = t(tri2)[None]
    ln = t(lines)[None]
    out = RL.cal_loss_intersection_batch_whole_median_pts_lines(*rng, p1, p2, ln, "cpu")
    if isinstance(out, tuple):
        return dict(loss=np.float32(np.nan), empty=True)
    out.backward()
    # per-bucket D values in the reference's concatenation order
    info1 = RL.cal_intersection_batch2_points_with_line(p1.detach(), ln)
    info2 = RL.cal_intersection_batch2_points_with_line(p2, ln)
    c1, c2 = info1[2].sum(-1), info2[2].sum(-1)
    Ds = []
    for k in range(rng[0], rng[2]):
        for j in range(rng[1], rng[3]):
            mask = ((c1 == k) * (c2 == j)).reshape(-1)
            dm = RL.cal_loss_intersection_batch_m_n_median_pts_lines(
                k, j, info1, info2, mask, tri1.shape[0], tri2.shape[0])
            if dm is not None:
                Ds.append(dm.reshape(-1).numpy())
    D = np.concatenate(Ds).astype(np.float32)
    med = torch.median(torch.from_numpy(D)).item()
    return dict(loss=np.float32(out.item()), empty=False, grad1=p1.grad[0].numpy().copy(),
                D=D, median=np.float32(med))


def save(name, **kw):
    kw["meta"] = np.array(repr(META))
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **kw)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def ref_lines(seed, r, center, v1, v2, n):
    torch.manual_seed(seed)
    out = RL.Random_uniform_distribution_lines_batch_efficient_resample(
        torch.tensor([[float(r)]]), t(center).reshape(1, 3), n, t(v1)[None], t(v2)[None], "cpu")
    return out[0].numpy().astype(np.float32)


def loss_fixture(name, tri1, tri2, lines, ranges):
    s1, s2 = ref_scan(tri1, lines), ref_scan(tri2, lines)
    kw = dict(tri1=tri1, tri2=tri2, lines=lines,
              count1=s1["count"], hit_line1=s1["hit_line"], hit_tri1=s1["hit_tri"], hit_w1=s1["hit_w"],
              count2=s2["count"], hit_line2=s2["hit_line"], hit_tri2=s2["hit_tri"], hit_w2=s2["hit_w"],
              margin=np.float64(min(margin(tri1, lines), margin(tri2, lines))),
              ranges=np.array(ranges, np.int32))
    for i, rng in enumerate(ranges):
        res = ref_loss_case(tri1, tri2, lines, rng)
        for k, v in res.items():
            kw[f"r{i}_{k}"] = v
        print(f"  range {rng}: loss={res['loss']} nD={len(res.get('D', []))}")
    print(f"  hits: src {int(s1['count'].sum())}, tar {int(s2['count'].sum())}, margin {kw['margin']:.3e}")
    save(name, **kw)


def read_obj_vertices(path):
    return np.array([[float(x) for x in ln.split()[1:4]] for ln in open(path) if ln.startswith("v ")],
                    np.float32)


def main():
    # --- A: synthetic unit-scale pair ------------------------------------------------
    for seed, (n, m, nl) in ((0, (256, 256, 2000)), (1, (300, 200, 1500))):
        pr = synth.make_pair(seed, n, m)
        lines = ref_lines(100 + seed, pr["radius"], pr["center"], pr["src"], pr["tar"], nl)
        print(f"synth seed {seed}: filled {(np.abs(lines).sum(1) > 0).sum()}/{nl}")
        loss_fixture(f"loss_synth_s{seed}.npz", pr["src_tri"], pr["tar_tri"], lines,
                     [(1, 1, 5, 5), (1, 1, 2, 2), (2, 1, 3, 2), (1, 2, 3, 5)])

    # --- B: demo scale (AABB-diagonal radius ~ 11.7), derived from sample pair 0 -----
    torch.manual_seed(123)
    np.random.seed(123)
    v1 = read_obj_vertices(os.path.join(REF, "sample_data/challenge_data/0_src_sample.obj"))
    v2 = read_obj_vertices(os.path.join(REF, "sample_data/challenge_data/0_tar_sample.obj"))
    n1 = RL.Sample_neighs(v1).reshape(-1, 9)
    n2 = RL.Sample_neighs(v2).reshape(-1, 9)
    c1, c2 = v1.mean(0, keepdims=True), v2.mean(0, keepdims=True)
    v1, v2 = v1 - c1, v2 - c2
    n1 = (n1.reshape(-1, 3) - c1).reshape(-1, 9).astype(np.float32)
    n2 = (n2.reshape(-1, 3) - c2).reshape(-1, 9).astype(np.float32)
    bb = RL.generate_bbox(t(v2)[None])[0].numpy()
    rad = float(np.linalg.norm(bb[0] - bb[-1]))
    lines = ref_lines(7, rad, v2.mean(0), v1, v2, 3000)
    print(f"demo: N={n1.shape[0]} M={n2.shape[0]} radius={rad:.3f} filled "
          f"{(np.abs(lines).sum(1) > 0).sum()}/3000")
    loss_fixture("loss_demo_scale.npz", n1, n2, lines, [(1, 1, 5, 5)])

    # --- C: edge cases ---------------------------------------------------------------
    pr = synth.make_pair(5, 128, 128)
    far = np.tile(np.array([[1, 0, 0, 0, 50, 50]], np.float32), (16, 1))  # all-miss lines
    out = RL.cal_loss_intersection_batch_whole_median_pts_lines(
        1, 1, 5, 5, t(pr["src_tri"])[None], t(pr["tar_tri"])[None], t(far)[None], "cpu")
    assert isinstance(out, tuple) and out[0] is None
    lines = ref_lines(9, pr["radius"], pr["center"], pr["src"], pr["tar"], 600)
    lines[::7] = 0.0  # zero (unfilled) rows are legal inputs
    dup = pr["src_tri"].copy()
    dup[1::2] = dup[0::2]  # duplicate pseudo-triangles -> bit-equal distances (tie case)
    loss_fixture("loss_edge_zero_dup.npz", dup, pr["tar_tri"], lines, [(1, 1, 5, 5)])
    save("loss_edge_allmiss.npz", tri1=pr["src_tri"], tri2=pr["tar_tri"], lines=far)

    # --- D: B=2 pooling quirk ---------------------------------------------
    pa, pb = synth.make_pair(11, 160, 160), synth.make_pair(12, 160, 160)
    la = ref_lines(21, pa["radius"], pa["center"], pa["src"], pa["tar"], 800)
    lb = ref_lines(22, pb["radius"], pb["center"], pb["src"], pb["tar"], 800)
    p1 = torch.stack([t(pa["src_tri"]), t(pb["src_tri"])]).requires_grad_(True)
    p2 = torch.stack([t(pa["tar_tri"]), t(pb["tar_tri"])])
    ln = torch.stack([t(la), t(lb)])
    out = RL.cal_loss_intersection_batch_whole_median_pts_lines(1, 1, 5, 5, p1, p2, ln, "cpu")
    out.backward()
    save("loss_b2_quirk.npz", tri1=p1.detach().numpy(), tri2=p2.numpy(), lines=ln.numpy(),
         loss=np.float32(out.item()), grad1=p1.grad.numpy())
    print("B=2 quirk loss", out.item())

    # --- E: se(3) exp / log ------------------------------------------------------------
    rng = np.random.default_rng(3)
    xis = []
    for mag in (0.0, 1e-4, 0.0099, 0.0101, 0.5, 3.1):
        ax = rng.standard_normal(3)
        ax /= np.linalg.norm(ax)
        xis.append(np.concatenate([mag * ax, rng.standard_normal(3)]))
    xis = np.array(xis, np.float32)
    Rs, ps = RL.se3.exp3(t(xis))
    g = RL.se3.exp(t(xis))
    lg = RL.se3.log(g)
    save("se3_exp_log.npz", xi=xis, R=Rs.numpy(), p=ps.numpy(), g=g.numpy(), log_of_exp=lg.numpy())

    # --- F: Reconstruction_point forward/backward ---------------------------------------
    np.random.seed(4)
    rec = RL.Reconstruction_point()
    with torch.no_grad():
        rec.parameters_.copy_(t(np.array([0.3, -0.2, 0.1, 0.05, 0.02, -0.04], np.float32)))
    pr = synth.make_pair(6, 200, 200)
    pts, nb = rec(t(pr["src"]), t(pr["src_tri"]).reshape(1, -1, 3))
    gp, gn = torch.randn(pts.shape, generator=torch.Generator().manual_seed(1)), \
        torch.randn(nb.shape, generator=torch.Generator().manual_seed(2))
    ((pts * gp).sum() + (nb * gn).sum()).backward()
    save("reconstruction_point.npz", xi=rec.parameters_.detach().numpy(), src=pr["src"],
         src_tri=pr["src_tri"], out_pts=pts.detach().numpy(), out_tri=nb.detach().numpy(),
         g_pts=gp.numpy(), g_tri=gn.numpy(), grad_xi=rec.parameters_.grad.numpy())

    # --- G: chamfer ---------------------------------------------------------------------
    gx = torch.Generator().manual_seed(8)
    x = torch.randn(2, 300, 3, generator=gx).requires_grad_(True)
    y = torch.randn(2, 257, 3, generator=gx)
    cd = RL.chamfer_dist(x, y)
    cd.backward()
    save("chamfer.npz", x=x.detach().numpy(), y=y.numpy(), value=np.float32(cd.item()),
         grad_x=x.grad.numpy())

    # --- H: sampler ---------------------------------------------------------------------
    pr = synth.make_pair(2, 96, 80)
    n = 400
    seed = 31
    torch.manual_seed(seed)
    rands = torch.stack([torch.stack([torch.rand(1, n)[0] for _ in range(4)]) for _ in range(10)])
    torch.manual_seed(seed)
    cand0 = RL.Random_uniform_distribution_lines_batch_efficient(
        torch.tensor([[float(pr["radius"])]]), t(pr["center"]).reshape(1, 3), n, "cpu")[0]
    fv1 = RL.generate_mesh_by_bbox(RL.generate_bbox(t(pr["src"])[None]), "cpu")
    fv2 = RL.generate_mesh_by_bbox(RL.generate_bbox(t(pr["tar"])[None]), "cpu")
    h1 = RL.cal_intersection_batch2_rand_lines(fv1, cand0[None])[0]
    h2 = RL.cal_intersection_batch2_rand_lines(fv2, cand0[None])[0]
    final = ref_lines(seed, pr["radius"], pr["center"], pr["src"], pr["tar"], n)
    # also the full-diagonal radius, which leaves unfilled rows
    final_big = ref_lines(seed, 4 * pr["radius"], pr["center"], pr["src"], pr["tar"], n)
    save("sampler.npz", src=pr["src"], tar=pr["tar"], radius=pr["radius"], center=pr["center"],
         rands=rands.numpy(), cand0=cand0.numpy(), hits1=h1.numpy().astype(np.int32),
         hits2=h2.numpy().astype(np.int32), final=final, final_big=final_big,
         bbox1=RL.generate_bbox(t(pr["src"])[None])[0].numpy(), seed=np.int32(seed))
    print("sampler: accepted round0", int(((h1 * h2) > 0).sum()), "/", n,
          "filled(big radius)", int((np.abs(final_big).sum(1) > 0).sum()))


def accept():
    """Row F accept test (code/loss.py:265-322, 415-432) on harder candidate sets than sampler.npz:
    the reference's label1 / label2 counts for lines drawn by the reference's own sampler around
    a regular box pair, a flat box (zero extent along z), a single-point box and a pair far from the
    origin (cancellation-heavy).  Inputs + the reference's outputs only."""
    rng = np.random.default_rng(123)
    out = {}
    cases = []
    pr = synth.make_pair(12, 300, 260)
    cases.append(("regular", pr["src"], pr["tar"], float(pr["radius"]), pr["center"]))
    flat = pr["src"].copy(); flat[:, 2] = 0.25
    cases.append(("flat", flat, pr["tar"], float(pr["radius"]), pr["center"]))
    single = np.tile(np.array([[0.1, -0.2, 0.05]], np.float32), (5, 1))
    cases.append(("single", single, pr["tar"], float(pr["radius"]), pr["center"]))
    off = np.array([40.0, -25.0, 17.0], np.float32)
    cases.append(("far", pr["src"] + off, pr["tar"] + off, float(pr["radius"]), pr["center"] + off))
    big = (pr["src"] * 12.0).astype(np.float32)
    cases.append(("demo_scale", big, (pr["tar"] * 12.0).astype(np.float32), 2.0 * 12.0 * float(pr["radius"]),
                  pr["center"] * 12.0))
    n = 1500
    for i, (tag, v1, v2, r, c) in enumerate(cases):
        torch.manual_seed(900 + i)
        cand = RL.Random_uniform_distribution_lines_batch_efficient(torch.tensor([[r]]), t(c).reshape(1, 3), n, "cpu")
        fv1 = RL.generate_mesh_by_bbox(RL.generate_bbox(t(v1)[None]), "cpu")
        fv2 = RL.generate_mesh_by_bbox(RL.generate_bbox(t(v2)[None]), "cpu")
        h1 = RL.cal_intersection_batch2_rand_lines(fv1, cand)[0].numpy().astype(np.int32)
        h2 = RL.cal_intersection_batch2_rand_lines(fv2, cand)[0].numpy().astype(np.int32)
        out[f"{tag}_v1"], out[f"{tag}_v2"] = np.asarray(v1, np.float32), np.asarray(v2, np.float32)
        out[f"{tag}_cand"], out[f"{tag}_hits1"], out[f"{tag}_hits2"] = cand[0].numpy(), h1, h2
        print("accept", tag, "box1", int((h1 > 0).sum()), "box2", int((h2 > 0).sum()), "both", int(((h1 * h2) > 0).sum()), "/", n)
    out["cases"] = np.array([c[0] for c in cases])
    save("accept.npz", **out)


def neighs():
    """Sample_neighs (FPS + KDTree 3-NN) on a synthetic cloud: all points, and a subsample."""
    pr = synth.make_pair(7, 900, 64)
    pts = pr["src"]
    torch.manual_seed(77)
    full = RL.Sample_neighs(pts)
    torch.manual_seed(78)
    sub = RL.Sample_neighs(pts, num_sample=300)
    save("sample_neighs.npz", points=pts, full=full.astype(np.float32), sub=sub.astype(np.float32))


def callsites():
    """The training call-site fragments (rpm/Train_RPM.py:204-259, dcp/Train_DCP.py:233-270,
    fmr/model.py:266-310) replayed with the REFERENCE's loss/sampler/chamfer functions on a
    small synthetic batch: per-sample loop, scalings and discounts as the trainers apply them."""
    B, n, nl = 3, 256, 3000
    prs = [synth.make_pair(40 + b, n, n) for b in range(B)]
    src = torch.stack([t(p["src"]) for p in prs])
    tar = torch.stack([t(p["tar"]) for p in prs])
    nb_src = torch.stack([t(p["src_tri"]).reshape(-1, 3) for p in prs])   # (B, 3n, 3)
    nb_tar = torch.stack([t(p["tar_tri"]).reshape(-1, 3) for p in prs])
    tar_box = RL.generate_bbox(tar)
    centers = tar.mean(1)
    g = torch.Generator().manual_seed(5)
    num_iter = 3
    xis = 0.08 * torch.randn(num_iter, B, 6, generator=g)
    Rs, ps = RL.se3.exp3(xis.reshape(-1, 6))
    Rs, ps = Rs.reshape(num_iter, B, 3, 3), ps.reshape(num_iter, B, 3)
    Rs.requires_grad_(True)
    ps.requires_grad_(True)

    def move(x, R, p):      # R x + p per point, what se3.transform / transform_point_cloud do
        return x @ R.transpose(-1, -2) + p[:, None, :]

    tar_tri = nb_tar.reshape(B, -1, 9)
    out = dict(src=src.numpy(), tar=tar.numpy(), nb_src=nb_src.numpy(), nb_tar=nb_tar.numpy(),
               tar_box=tar_box.numpy(), centers=centers.numpy(), R=Rs.detach().numpy(),
               t=ps.detach().numpy())

    # ---- RPM: radius = full diagonal, 10000 lines in the trainer (nl here), /num_iter, discount
    radius = torch.norm(tar_box[:, 0] - tar_box[:, -1], dim=-1).reshape(-1, 1)
    torch.manual_seed(61)
    lines = RL.Random_uniform_distribution_lines_batch_efficient_resample(
        radius, centers, nl, move(src, Rs[0], ps[0]).detach(), tar, "cpu")
    per_iter, per_sample, chamf = [], [], []
    for ni in range(num_iter):
        moved = move(src, Rs[ni], ps[ni])
        tri = move(nb_src, Rs[ni], ps[ni]).reshape(B, -1, 9)
        acc = torch.zeros(1)
        row = []
        for j in range(B):
            lj = RL.cal_loss_intersection_batch_whole_median_pts_lines(
                1, 1, 5, 5, tri[j:j + 1], tar_tri[j:j + 1], lines[j:j + 1], "cpu")
            row.append(lj.item())
            acc = acc + lj
        per_sample.append(row)
        per_iter.append(acc / num_iter)
        chamf.append(RL.chamfer_dist(tar, moved).detach())
    disc = [0.5 ** (num_iter - ni - 1) for ni in range(num_iter)]
    total = torch.stack([per_iter[i] * disc[i] for i in range(num_iter)]).sum(0)
    total_cd = torch.stack([chamf[i] * disc[i] for i in range(num_iter)]).sum(0)
    total.backward()
    out.update(rpm_lines=lines.numpy(), rpm_per_sample=np.array(per_sample, np.float32),
               rpm_per_iter=np.array([x.item() for x in per_iter], np.float32),
               rpm_loss=np.float32(total.item()), rpm_chamfer=np.float32(total_cd.item()),
               rpm_grad_R=Rs.grad.numpy().copy(), rpm_grad_t=ps.grad.numpy().copy())
    print("rpm fragment:", total.item(), total_cd.item(), per_sample)
    Rs.grad = None
    ps.grad = None

    # ---- DCP: radius = half diagonal, channel-first clouds, /5.0 per sample, /batch_size
    radius = (torch.norm(tar_box[:, 0] - tar_box[:, -1], dim=-1) * 0.5).reshape(-1, 1)
    moved = move(src, Rs[0], ps[0])
    torch.manual_seed(62)
    lines = RL.Random_uniform_distribution_lines_batch_efficient_resample(
        radius, centers, nl, moved.detach(), tar, "cpu")
    tri = move(nb_src, Rs[0], ps[0]).reshape(B, -1, 9)
    acc = torch.zeros(1)
    for j in range(B):
        acc = acc + RL.cal_loss_intersection_batch_whole_median_pts_lines(
            1, 1, 5, 5, tri[j:j + 1], tar_tri[j:j + 1], lines[j:j + 1], "cpu") / 5.0
    dcp = acc / B
    cd = RL.chamfer_dist(moved, tar)
    dcp.backward()
    out.update(dcp_lines=lines.numpy(), dcp_loss=np.float32(dcp.item()),
               dcp_chamfer=np.float32(cd.item()), dcp_grad_R=Rs.grad[0].numpy().copy(),
               dcp_grad_t=ps.grad[0].numpy().copy())
    print("dcp fragment:", dcp.item(), cd.item())
    Rs.grad = None
    ps.grad = None

    # ---- FMR: half diagonal, lines from the LAST estimate, last three estimates, /5.0,
    # discount 0.5**(maxiter-i-1), /batch_size
    torch.manual_seed(63)
    lines = RL.Random_uniform_distribution_lines_batch_efficient_resample(
        radius, centers, nl, move(src, Rs[-1], ps[-1]).detach(), tar, "cpu")
    total = torch.zeros(1)
    for i in range(num_iter - 3, num_iter):
        tri = move(nb_src, Rs[i], ps[i]).reshape(B, -1, 9)
        acc = torch.zeros(1)
        for j in range(B):
            acc = acc + RL.cal_loss_intersection_batch_whole_median_pts_lines(
                1, 1, 5, 5, tri[j:j + 1], tar_tri[j:j + 1], lines[j:j + 1], "cpu") / 5.0
        total = total + acc * 0.5 ** (num_iter - i - 1)
    fmr = total / B
    cd = RL.chamfer_dist(move(src, Rs[-1], ps[-1]), tar)
    fmr.backward()
    out.update(fmr_lines=lines.numpy(), fmr_loss=np.float32(fmr.item()),
               fmr_chamfer=np.float32(cd.item()), fmr_grad_R=Rs.grad.numpy().copy(),
               fmr_grad_t=ps.grad.numpy().copy())
    print("fmr fragment:", fmr.item(), cd.item())
    save("callsites.npz", **out)


def callsites_moved():
    """The TIGHT W check inside the fragment fixtures (round 5): callsites.npz stores the trainers'
    inputs and the reference's results, but not the reference's MOVED pseudo-triangles -- the build moves the source on
    the GPU (x R^T + t with FMAs) while the reference's BLAS matmul rounds differently, an ulp in a vertex can flip a
    borderline label, and the end-to-end fragment tests therefore compare at 2e-4.  SURVEY 8a-R defines W's parity "on
    identical transformed inputs": this fixture adds, for the RPM fragment of callsites.npz (same seeds: the poses, lines and
    per-sample values are re-derived and checked against that file), the reference's moved triangles of every iteration, its
    per-line hit counts on them (labels), its per-sample loss and the gradient dL/dpoints1 -- so the HIP loss can be checked
    at 1e-5 / label-exact on exactly the reference's inputs, separating "FMA in the rigid apply" from anything else."""
    old = np.load(os.path.join(HERE, "callsites.npz"))
    B, n, nl, num_iter = 3, 256, 3000, 3
    src_nb, tar_nb = t(old["nb_src"]), t(old["nb_tar"])
    Rs, ps = t(old["R"]), t(old["t"])
    lines = t(old["rpm_lines"])
    tar_tri = tar_nb.reshape(B, -1, 9)

    def move(x, R, p):
        return x @ R.transpose(-1, -2) + p[:, None, :]

    moved, per_sample, grads, c1s, c2s = [], [], [], [], None
    for ni in range(num_iter):
        tri = move(src_nb, Rs[ni], ps[ni]).reshape(B, -1, 9).detach().clone().requires_grad_(True)
        row = []
        acc = torch.zeros(1)
        for j in range(B):
            lj = RL.cal_loss_intersection_batch_whole_median_pts_lines(
                1, 1, 5, 5, tri[j:j + 1], tar_tri[j:j + 1], lines[j:j + 1], "cpu")
            row.append(lj.item())
            acc = acc + lj
        acc.backward()
        moved.append(tri.detach().numpy().copy())
        grads.append(tri.grad.numpy().copy())
        per_sample.append(row)
        c1s.append(np.stack([ref_scan(tri[j].detach().numpy(), lines[j].numpy())["count"] for j in range(B)]))
    c2s = np.stack([ref_scan(tar_tri[j].numpy(), lines[j].numpy())["count"] for j in range(B)])
    per_sample = np.array(per_sample, np.float32)
    assert np.array_equal(per_sample, old["rpm_per_sample"]), (per_sample, old["rpm_per_sample"])  # the same fragment
    mg = min(margin(moved[ni][j], lines[j].numpy()) for ni in range(num_iter) for j in range(B))
    np.savez_compressed(os.path.join(HERE, "callsites_moved.npz"), moved_tri=np.stack(moved).astype(np.float32),
                        grad_tri=np.stack(grads).astype(np.float32), per_sample=per_sample,
                        count1=np.stack(c1s).astype(np.int16), count2=c2s.astype(np.int16), margin=np.float64(mg),
                        **{k: np.array(str(v)) for k, v in META.items()})
    print("callsites_moved: per-sample", per_sample.tolist(), "margin %.3g" % mg,
          "selected lines / sample", [[int(((c1s[ni][j] >= 1) & (c1s[ni][j] <= 4) & (c2s[j] >= 1) & (c2s[j] <= 4)).sum())
                                       for j in range(B)] for ni in range(num_iter)])


def callsites_moved_dcp_fmr():
    """callsites_moved.npz for the OTHER two trainers (round 6): the DCP fragment moves CHANNEL-FIRST clouds
    with code/utils.py:32-37 transform_point_cloud (rot @ cloud + t[:, :, None]; dcp/Train_DCP.py:233-270), the FMR fragment moves
    them with its 4 x 4 matrices through fmr/se_math/se3.py:110-124 transform (R @ a[..., None] + p; fmr/model.py:265-313).
    Both are imported from the reference and applied to callsites.npz's inputs with its poses and lines; stored: the
    reference's MOVED pseudo-triangles, its per-line hit counts on them (labels), per-sample losses (before the trainers'
    / 5.0) and dL/dpoints1 -- the HIP loss is then checked label-exact / 1e-5 on exactly the reference's inputs for these two
    layouts too.  The fragments' totals are re-derived and checked against callsites.npz (same seeds: the same fragment)."""
    import utils as RU  # code/utils.py (the reference's)
    # (se_math/__init__.py imports its mesh module, which wants `plyfile` -- absent here and unused by the path: an empty stub,
    #  like openmesh / trimesh above)
    sys.modules.setdefault("plyfile", types.ModuleType("plyfile")).PlyData = object
    sys.path.insert(0, os.path.join(REF, "exps_deep_learning", "fmr"))
    from se_math import se3 as RSE3  # fmr/se_math/se3.py (the reference's)
    old = np.load(os.path.join(HERE, "callsites.npz"))
    B, num_iter = old["nb_tar"].shape[0], old["R"].shape[0]
    src_nb, tar_nb = t(old["nb_src"]), t(old["nb_tar"])
    Rs, ps = t(old["R"]), t(old["t"])
    tar_tri = tar_nb.reshape(B, -1, 9)
    c2 = {}

    def evaluate(tri, lines):
        tri = tri.detach().clone().requires_grad_(True)
        row, acc = [], torch.zeros(1)
        for j in range(B):
            lj = RL.cal_loss_intersection_batch_whole_median_pts_lines(1, 1, 5, 5, tri[j:j + 1], tar_tri[j:j + 1], lines[j:j + 1], "cpu")
            row.append(lj.item())
            acc = acc + lj
        acc.backward()
        c1 = np.stack([ref_scan(tri[j].detach().numpy(), lines[j].numpy())["count"] for j in range(B)])
        return tri.detach().numpy().copy(), tri.grad.numpy().copy(), np.array(row, np.float32), c1

    # ---- DCP: channel-first clouds (B, 3, 3N), pose 0
    lines = t(old["dcp_lines"])
    moved_cf = RU.transform_point_cloud(src_nb.transpose(2, 1).contiguous(), Rs[0], ps[0])  # (B, 3, 3N)
    dcp_tri = moved_cf.transpose(2, 1).reshape(B, -1, 9)
    d_tri, d_grad, d_row, d_c1 = evaluate(dcp_tri, lines)
    d_c2 = np.stack([ref_scan(tar_tri[j].numpy(), lines[j].numpy())["count"] for j in range(B)])
    dcp_total = np.float32((d_row.astype(np.float64) / 5.0).sum() / B)
    assert abs(float(dcp_total) - float(old["dcp_loss"])) <= 2e-6 * abs(float(old["dcp_loss"])), (dcp_total, old["dcp_loss"])
    mg = min(margin(d_tri[j], lines[j].numpy()) for j in range(B))
    # ---- FMR: 4 x 4 matrices, the last three estimates
    lines_f = t(old["fmr_lines"])
    bottom = torch.tensor([0.0, 0, 0, 1]).expand(B, 1, 4)
    f_tri, f_grad, f_row, f_c1 = [], [], [], []
    for i in range(num_iter - 3, num_iter):
        g4 = torch.cat([torch.cat([Rs[i], ps[i][..., None]], dim=-1), bottom], dim=1)  # (B, 4, 4)
        tri = RSE3.transform(g4.unsqueeze(1), src_nb).reshape(B, -1, 9)
        a_, b_, c_, d_ = evaluate(tri, lines_f)
        f_tri.append(a_); f_grad.append(b_); f_row.append(c_); f_c1.append(d_)
        mg = min(mg, min(margin(a_[j], lines_f[j].numpy()) for j in range(B)))
    f_c2 = np.stack([ref_scan(tar_tri[j].numpy(), lines_f[j].numpy())["count"] for j in range(B)])
    fmr_total = sum((f_row[k].astype(np.float64) / 5.0).sum() * 0.5 ** (num_iter - i - 1)
                    for k, i in enumerate(range(num_iter - 3, num_iter))) / B
    assert abs(fmr_total - float(old["fmr_loss"])) <= 2e-6 * abs(float(old["fmr_loss"])), (fmr_total, old["fmr_loss"])
    np.savez_compressed(os.path.join(HERE, "callsites_moved_dcp_fmr.npz"),
                        dcp_moved_tri=d_tri.astype(np.float32), dcp_grad_tri=d_grad.astype(np.float32), dcp_per_sample=d_row,
                        dcp_count1=d_c1.astype(np.int16), dcp_count2=d_c2.astype(np.int16),
                        fmr_moved_tri=np.stack(f_tri).astype(np.float32), fmr_grad_tri=np.stack(f_grad).astype(np.float32),
                        fmr_per_sample=np.stack(f_row), fmr_count1=np.stack(f_c1).astype(np.int16), fmr_count2=f_c2.astype(np.int16),
                        margin=np.float64(mg), **{k: np.array(str(v)) for k, v in META.items()})
    print("callsites_moved_dcp_fmr: dcp per-sample", d_row.tolist(), "fmr per-sample", [r.tolist() for r in f_row], "margin %.3g" % mg)


def demo_trajectory():
    """test_demo_optimized_Lie_Algebra.py:27-75 (test_one_case) replayed with the reference's
    modules for a few epochs on a small synthetic pair; the sampled lines of every epoch are
    recorded so the GPU harness can be driven with the same lines."""
    n, nl, epochs = 300, 2500, 8
    pr = synth.make_pair(51, n, n)
    v1, v2 = t(pr["src"]), t(pr["tar"])
    f1 = t(pr["src_tri"]).reshape(1, -1, 3)
    f2 = t(pr["tar_tri"])
    bbox = RL.generate_bbox(v2[None])[0]
    centers = v2.mean(0)
    np.random.seed(9)
    torch.manual_seed(9)
    rec = RL.Reconstruction_point()
    xi0 = rec.parameters_.detach().numpy().copy()
    opt = torch.optim.Adam(rec.parameters(), lr=2e-2)
    R = (bbox[0] - bbox[-1]).norm(p=2)
    cur = v1
    lines_all, losses, chamfers, xis, lrs = [], [], [], [], []
    for epoch in range(epochs):
        lines = RL.Random_uniform_distribution_lines_batch_efficient_resample(
            torch.FloatTensor([R]).reshape(1, 1), centers.reshape(1, -1), nl,
            cur.view(1, -1, 3), v2.view(1, -1, 3), "cpu").detach().view(-1, 6)
        lr = opt.param_groups[0]["lr"]
        if epoch % 1000 == 0:
            lr *= 0.5
        for gparam in opt.param_groups:
            gparam["lr"] = lr
        cur, tri = rec(v1, f1)
        loss = RL.cal_loss_intersection_batch_whole_median_pts_lines(
            1, 1, 5, 5, tri.reshape(1, -1, 9), f2.reshape(1, -1, 9), lines.reshape(1, -1, 6), "cpu")
        assert not isinstance(loss, tuple)
        opt.zero_grad()
        loss.backward()
        opt.step()
        cf = RL.chamfer_dist(cur.reshape(-1, cur.shape[0], 3), v2.reshape(-1, v2.shape[0], 3))
        lines_all.append(lines.numpy().copy())
        losses.append(loss.item())
        chamfers.append(cf.item())
        xis.append(rec.parameters_.detach().numpy().copy())
        lrs.append(lr)
        cur = cur.detach()
    print("demo trajectory: loss", losses, "chamfer", chamfers)
    save("demo_trajectory.npz", src=pr["src"], tar=pr["tar"], src_tri=pr["src_tri"],
         tar_tri=pr["tar_tri"], bbox=bbox.numpy(), centers=centers.numpy(), xi0=xi0,
         lines=np.array(lines_all, np.float32), loss=np.array(losses, np.float32),
         chamfer=np.array(chamfers, np.float32), xi=np.array(xis, np.float32),
         lr=np.array(lrs, np.float64))


def dataset():
    """Dataset_2021_8_29.__getitem__ / random_data (exps_deep_learning/pre_dataloader.py:28-181)
    on two small synthetic pairs written to a temp directory.  `igl` and `h5py` are absent: h5py
    is unused by the class; the two igl calls are served by a vertex-line OBJ parser and by the
    reference's own generate_bbox (same corner order as igl.bounding_box)."""
    import tempfile
    igl = sys.modules["igl"]

    def read_triangle_mesh(path):
        V = np.array([[float(x) for x in ln.split()[1:4]] for ln in open(path) if ln.startswith("v ")])
        return V.reshape(-1, 3), np.zeros((0, 3), np.int64)

    def bounding_box(V):
        return RL.generate_bbox(torch.from_numpy(V)[None])[0].numpy(), None

    igl.read_triangle_mesh, igl.bounding_box = read_triangle_mesh, bounding_box
    sys.modules.setdefault("h5py", types.ModuleType("h5py"))
    sys.path.insert(0, os.path.join(REF, "exps_deep_learning"))
    import pre_dataloader as RD  # the reference
    sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "rrl_pre_dataloader", os.path.join(ROOT, "a-robust-registration-loss_amd", "pre_dataloader.py"))
    mine = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mine)  # only its file WRITER is used here

    rng = np.random.default_rng(12)
    out = {}
    with tempfile.TemporaryDirectory(prefix="rrl_ds_") as d:
        srcs, tars = [], []
        for i, (n, m) in enumerate(((60, 48), (40, 56))):
            pr = synth.make_pair(70 + i, n, m)
            off_s, off_t = rng.standard_normal(3), rng.standard_normal(3)  # un-centred on disk
            A = synth._rotation(rng.standard_normal(3), 25.0)
            gt = np.concatenate([A, rng.standard_normal((3, 1))], 1)
            nrm_s = rng.standard_normal((n, 3)).astype(np.float32)
            nrm_t = rng.standard_normal((m, 3)).astype(np.float32)
            inp = dict(src=(pr["src"] + off_s).astype(np.float32), tar=(pr["tar"] + off_t).astype(np.float32),
                       src_neigh=(pr["src_tri"].reshape(-1, 3) + off_s).astype(np.float32),
                       tar_neigh=(pr["tar_tri"].reshape(-1, 3) + off_t).astype(np.float32),
                       transform=gt, normals_src=nrm_s, normals_tar=nrm_t)
            a, b = mine.write_pair(d, i, 0, **inp)
            srcs.append(a)
            tars.append(b)
            for k, v in inp.items():
                out[f"in{i}_{k}"] = v
        for tag, kw in (("plain", {}), ("dcp", dict(DCP_True=True)), ("fmr", dict(FMR_True=True))):
            ds = RD.Dataset_2021_8_29(srcs, tars, **kw)
            assert len(ds) == 2
            for i in range(2):
                for k, v in ds[i].items():
                    out[f"{tag}{i}_{k}"] = np.ascontiguousarray(v)
        # augmentation: needs the keys the method reads ('normals_ref') -- add as the trainers' dicts would
        ds = RD.Dataset_2021_8_29(srcs, tars)
        item = ds[0]
        item["normals_ref"] = item["normals_tar"]
        np.random.seed(33)
        aug = ds.random_data(item)
        for k, v in aug.items():
            out[f"aug0_{k}"] = np.ascontiguousarray(v)
    ax, th = np.array([0.3, -0.2, 0.9]), np.array([0.7])
    out["M_axis"], out["M_theta"], out["M_out"] = ax, th, RD.M(ax, th)
    save("dataset.npz", **out)


def _ref_pair(group, label, seed):
    """One pair of the reference's own sample data, prepared exactly like its demo does
    (code/test_demo_optimized_Lie_Algebra.py:103-143): vertices of the two OBJ files, pseudo-triangles by the
    reference's Sample_neighs, both centred, sampling sphere = the target's AABB diagonal around its mean.
    Only DERIVED arrays leave this function (centred points / pseudo-triangles); never the OBJ text."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    v1 = read_obj_vertices(os.path.join(REF, f"sample_data/{group}/{label}_src_sample.obj"))
    v2 = read_obj_vertices(os.path.join(REF, f"sample_data/{group}/{label}_tar_sample.obj"))
    n1 = RL.Sample_neighs(v1).reshape(-1, 9)
    n2 = RL.Sample_neighs(v2).reshape(-1, 9)
    c1, c2 = v1.mean(0, keepdims=True), v2.mean(0, keepdims=True)
    v1, v2 = (v1 - c1).astype(np.float32), (v2 - c2).astype(np.float32)
    n1 = (n1.reshape(-1, 3) - c1).reshape(-1, 9).astype(np.float32)
    n2 = (n2.reshape(-1, 3) - c2).reshape(-1, 9).astype(np.float32)
    bb = RL.generate_bbox(t(v2)[None])[0].numpy()
    return v1, v2, n1, n2, bb, float(np.linalg.norm(bb[0] - bb[-1]))


REF_PAIRS_A = (("airplane_data", "0", "airplane0", 3000), ("airplane_data", "3", "airplane3", 3000),
               ("human_data", "0", "human0", 3000), ("real_data", "0", "real0", 2500))
# ... and every other pair the reference ships (challenge_data/0 is loss_demo_scale.npz): all 12 are fixtures
REF_PAIRS_B = (("airplane_data", "1", "airplane1", 2000), ("airplane_data", "2", "airplane2", 2000),
               ("airplane_data", "4", "airplane4", 2000), ("human_data", "1", "human1", 2000),
               ("human_data", "2", "human2", 2000), ("real_data", "1", "real1", 2000), ("real_data", "2", "real2", 2000))


def refdata_rest():
    """The remaining seven pairs of code/sample_data/ (2000 lines each), same recipe as refdata()."""
    refdata(REF_PAIRS_B)


def refdata(pairs=REF_PAIRS_A):
    """Loss fixtures on the REFERENCE'S OWN sample pairs (code/sample_data/*; its demo iterates such pairs,
    test_demo_optimized_Lie_Algebra.py:158-162): two airplane pairs (1024 / 1024), a human pair (N = 1024, M = 2048: the
    first N != M fixture on reference data) and a real-scan fragment pair (2048 / 2048, BASELINE configs[4]'s kind of
    data) -- counts, hit lists, weights, D, median, loss and points1.grad from the reference.  The line seed of each
    fixture is the first one whose label decisions are provably non-borderline (margin > 2e-6 >> 1 ulp)."""
    for group, label, tag, nl in pairs:
        v1, v2, n1, n2, bb, rad = _ref_pair(group, label, 123)
        for seed in range(7, 40):
            lines = ref_lines(seed, rad, v2.mean(0), v1, v2, nl)
            mg = min(margin(n1, lines), margin(n2, lines))
            if mg > (2e-6 if pairs is REF_PAIRS_A else 1.2e-5):  # (the tests ask for > 1e-5; the first four were taken at > 2e-6
                break                                             #  and happen to clear it)
        print(f"{tag}: N={n1.shape[0]} M={n2.shape[0]} radius={rad:.3f} filled "
              f"{(np.abs(lines).sum(1) > 0).sum()}/{nl} line seed {seed} margin {mg:.2e}")
        loss_fixture(f"loss_ref_{tag}.npz", n1, n2, lines, [(1, 1, 5, 5)])


def demo_trajectory_airplane():
    """The demo loop (test_demo_optimized_Lie_Algebra.py:27-75) on a pair of the reference's own data
    (airplane_data/1, N = M = 1024), 8 epochs with recorded lines: like demo_trajectory() on reference data."""
    nl, epochs = 3000, 8
    v1n, v2n, n1, n2, bbn, _ = _ref_pair("airplane_data", "1", 123)
    v1, v2 = t(v1n), t(v2n)
    f1 = t(n1).reshape(1, -1, 3)
    f2 = t(n2)
    bbox = RL.generate_bbox(v2[None])[0]
    centers = v2.mean(0)
    np.random.seed(19)
    torch.manual_seed(19)
    rec = RL.Reconstruction_point()
    xi0 = rec.parameters_.detach().numpy().copy()
    opt = torch.optim.Adam(rec.parameters(), lr=2e-2)
    R = (bbox[0] - bbox[-1]).norm(p=2)
    cur = v1
    lines_all, losses, chamfers, xis, lrs = [], [], [], [], []
    for epoch in range(epochs):
        lines = RL.Random_uniform_distribution_lines_batch_efficient_resample(
            torch.FloatTensor([R]).reshape(1, 1), centers.reshape(1, -1), nl,
            cur.view(1, -1, 3), v2.view(1, -1, 3), "cpu").detach().view(-1, 6)
        lr = opt.param_groups[0]["lr"]
        if epoch % 1000 == 0:
            lr *= 0.5
        for gparam in opt.param_groups:
            gparam["lr"] = lr
        cur, tri = rec(v1, f1)
        loss = RL.cal_loss_intersection_batch_whole_median_pts_lines(
            1, 1, 5, 5, tri.reshape(1, -1, 9), f2.reshape(1, -1, 9), lines.reshape(1, -1, 6), "cpu")
        assert not isinstance(loss, tuple)
        opt.zero_grad()
        loss.backward()
        opt.step()
        cf = RL.chamfer_dist(cur.reshape(-1, cur.shape[0], 3), v2.reshape(-1, v2.shape[0], 3))
        lines_all.append(lines.numpy().copy())
        losses.append(loss.item())
        chamfers.append(cf.item())
        xis.append(rec.parameters_.detach().numpy().copy())
        lrs.append(lr)
        cur = cur.detach()
    print("airplane demo trajectory: loss", losses, "chamfer", chamfers)
    save("demo_trajectory_airplane.npz", src=v1n, tar=v2n, src_tri=n1, tar_tri=n2, bbox=bbox.numpy(),
         centers=centers.numpy(), xi0=xi0, lines=np.array(lines_all, np.float32), loss=np.array(losses, np.float32),
         chamfer=np.array(chamfers, np.float32), xi=np.array(xis, np.float32), lr=np.array(lrs, np.float64))


if __name__ == "__main__":
    which = sys.argv[1:] or ["main", "neighs", "callsites", "demo_trajectory", "dataset", "accept", "refdata",
                             "demo_trajectory_airplane", "refdata_rest", "callsites_moved", "callsites_moved_dcp_fmr"]
    for name in which:
        globals()[name]()
