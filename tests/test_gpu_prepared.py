"""Prepared clouds (round 4; include/rrl.h rrl_cloud_order, rrl_opts.order1 / order2, RRL_F_TARGET_KEPT).

The reference's callers move the same source rigidly, step after step, against a target that never moves
(code/test_demo_optimized_Lie_Algebra.py:57-62, rpm/Train_RPM.py:207-231), and the reference points at a spatial
structure itself (code/loss.py:260-262).  The spatial order of a cloud is therefore computed once and every later
step runs the prepared build (records at their sorted positions + tree refit, one launch, no cell sort).  The bar:
labels, hit lists, median, bucket sums and loss BIT-IDENTICAL to the plain (sorting) path and to the strict scan --
for an order taken in the cloud's own frame, in another pose, for a stale order and for an arbitrary permutation.
"""
import numpy as np
import pytest
import torch

from test_gpu_parity import _check_sorted_layout, cu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import loss
    from rrl_hip import _lib
    _lib.load()
    assert torch.cuda.is_available()
    return loss


def _lines(L, prs, nl):
    out = []
    for b, p in enumerate(prs):
        torch.manual_seed(100 + b)
        out.append(L.Random_uniform_distribution_lines_batch_efficient_resample(
            torch.tensor([[float(p["radius"])]]), torch.from_numpy(p["center"]).reshape(1, 3), nl, cu(p["src"])[None],
            cu(p["tar"])[None], "cuda")[0])
    return torch.stack(out)


def _pairs(seed, B, n, m):
    from rrl_hip import synth
    prs = [synth.make_pair(seed + b, n, m) for b in range(B)]
    return prs, cu(np.stack([p["src_tri"] for p in prs])), cu(np.stack([p["tar_tri"] for p in prs]))


def _hits_sorted(st, which):
    cnt = (st.count1 if which == 1 else st.count2).clone()
    hit = (st.hit1 if which == 1 else st.hit2).clone()
    k = torch.where(cnt <= 4, cnt, torch.zeros_like(cnt))  # beyond 4 hits only the first four ARRIVALS are kept (no bucket takes the line)
    mask = torch.arange(4, device=hit.device)[None, None, :] < k[..., None]
    hit = torch.where(mask, hit, torch.full_like(hit, 1 << 30))
    return cnt, hit.sort(-1).values


def _same_evaluation(a, b):
    for w in (1, 2):
        ca, ha = _hits_sorted(a, w)
        cb, hb = _hits_sorted(b, w)
        assert torch.equal(ca, cb) and torch.equal(ha, hb)
    for x, y in ((a.loss, b.loss), (a.med, b.med), (a.info, b.info), (a.bsum, b.bsum), (a.bcnt, b.bcnt), (a.status[:1], b.status[:1])):
        assert torch.equal(x, y)


@pytest.mark.parametrize("B,n,m,nl", [(2, 1200, 1000, 6000), (1, 300, 200, 2500), (1, 5000, 4100, 1500), (2, 65, 64, 300),
                                      (1, 16390, 500, 700), (3, 3, 70, 200)])
def test_cloud_order_and_prepared_forward(L, oracle, B, n, m, nl):
    """rrl_cloud_order gives a permutation per sample; a forward that is handed the orders equals the plain forward and the
    strict scan bit for bit, and its sorted layout (IDX / P0S / sphere tree) satisfies the build step's invariants."""
    from rrl_hip import ops
    prs, t1, t2 = _pairs(300, B, n, m)
    ln = _lines(L, prs, nl)
    o1, o2 = ops.cloud_order(t1), ops.cloud_order(t2)
    assert o1.shape == (B, (n + 63) // 64 * 64) and o1.dtype == torch.int32
    for o, k in ((o1, n), (o2, m)):
        oc = o.cpu().numpy()
        for b in range(B):
            assert sorted(oc[b, :k].tolist()) == list(range(k))
            assert not oc[b, k:].any()
    plain = ops.loss_forward_raw(t1, t2, ln, mode="cull")
    strict = ops.loss_forward_raw(t1, t2, ln, mode="strict")
    prep = ops.loss_forward_raw(t1, t2, ln, mode="cull", opts=ops.make_opts(order1=o1, order2=o2))
    staged = ops.loss_forward_raw(t1, t2, ln, mode="cull", staged=True, opts=ops.make_opts(order1=o1, order2=o2))
    # a prepared build followed by stage calls that are NOT handed the build's options: the workspace itself says where the
    # 48-byte records lie (slot 7 of the cloud's first partial row), PMAX is complete after rrl_tri_prepare_ex
    mixed = ops.loss_forward_raw(t1, t2, ln, mode="cull", staged="mixed", opts=ops.make_opts(order1=o1, order2=o2))
    torch.cuda.synchronize()
    for st in (strict, prep, staged, mixed):
        _same_evaluation(plain, st)
    assert torch.equal(prep.idx1, o1) and torch.equal(prep.idx2, o2)
    assert torch.equal(plain.pmax, prep.pmax) and torch.equal(plain.pmax, staged.pmax)
    assert int(prep.status[1]) == 0  # no wavefront left the culled path
    for b in range(min(B, 2)):
        _check_sorted_layout(prep, prs[b]["src_tri"], oracle.tri_threshold(prs[b]["src_tri"]), 1, b)
        _check_sorted_layout(prep, prs[b]["tar_tri"], oracle.tri_threshold(prs[b]["tar_tri"]), 2, b)


@pytest.mark.parametrize("n,m", [(1200, 1000), (5000, 4100), (70, 130)])
def test_refit_tree_equals_the_sort_kernels_tree(L, n, m):
    """Handed the plain path's own order (its IDX arrays), the prepared build leaves the SAME workspace: sorted records,
    indices, every node of the sphere tree, prepared triangles, NaN reach, max |P|^2 -- the DPP refit over 8 / 16 / 64
    lanes reproduces tri_sort_kernel's one-lane-per-half tree bit for bit."""
    from rrl_hip import ops
    prs, t1, t2 = _pairs(320, 2, n, m)
    ln = _lines(L, prs, 1500)
    plain = ops.loss_forward_raw(t1, t2, ln, mode="cull")
    prep = ops.loss_forward_raw(t1, t2, ln, mode="cull",
                                opts=ops.make_opts(order1=plain.idx1.clone(), order2=plain.idx2.clone()))
    torch.cuda.synchronize()
    for f in ("idx1", "idx2", "p0s1", "p0s2", "pmax"):
        assert torch.equal(getattr(plain, f), getattr(prep, f)), f
    for f, k in (("del1", n), ("del2", m)):  # the NaN reach sits where its record sits
        idx = getattr(plain, "idx" + f[-1])[:, :k].long()
        assert torch.equal(torch.gather(getattr(plain, f), 1, idx), getattr(prep, f)), f
    for f in ("ptri1", "ptri2"):  # the prepared build keeps the 48-byte records at their SORTED positions
        assert not torch.equal(getattr(plain, f), getattr(prep, f)), f
        assert torch.equal(_ptri_by_triangle(getattr(plain, f)), _ptri_by_triangle(getattr(prep, f))), f
        assert torch.equal(getattr(prep, f)[..., 11].view(torch.int32), getattr(plain, "idx" + f[-1])[:, :getattr(prep, f).shape[1]]), f
    for f in ("grp1", "grp2"):  # NaN radii mark empty nodes: compare the bit patterns
        assert torch.equal(getattr(plain, f).view(torch.int32), getattr(prep, f).view(torch.int32)), f
    _same_evaluation(plain, prep)


def _ptri_by_triangle(rows):
    """PTRI [B][n][12] re-indexed by the triangle index every record carries (slot 11): the cold build keeps the records in
    original order, the prepared build at their sorted positions -- the same records either way."""
    f = rows[..., 11].contiguous().view(torch.int32).long()
    out = torch.full_like(rows, float("nan"))
    out.scatter_(1, f.unsqueeze(-1).expand_as(rows), rows)
    return out


def _rot(axis, deg):
    a = np.deg2rad(deg)
    x, y, z = axis
    K = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]], np.float64)
    return (np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)).astype(np.float32)


@pytest.mark.parametrize("one_call", [True, False])
def test_prepared_step_over_poses(L, one_call):
    """The order is taken ONCE in the source's own frame; steps at the identity, after 90 and 180 degree turns, a general
    pose and a large translation equal the cold (sorting) step: loss / median / info / bucket sums / counts bit for bit,
    (dR, dt, payload) to the rounding of their atomics.  A stale order (taken in ANOTHER pose of another cloud of the same
    size) and a random permutation give the same bits too."""
    from rrl_hip import ops
    B, n, m, nl = 3, 1500, 1300, 5000
    prs, src, tar = _pairs(340, B, n, m)
    ln = _lines(L, prs, nl)
    poses = [(np.eye(3, dtype=np.float32), np.zeros(3, np.float32)),
             (_rot((0, 0, 1), 90), np.zeros(3, np.float32)),
             (_rot((1, 0, 0), 180), np.array([0.1, 0.0, -0.05], np.float32)),
             (_rot((0.6, 0.0, 0.8), 33), np.array([0.02, -0.03, 0.01], np.float32)),
             (_rot((0, 1, 0), 7), np.array([3.0, -2.0, 1.0], np.float32))]
    try:
        ops.RegistrationStep.ONE_CALL = one_call
        cold = ops.RegistrationStep(src, tar, nl, want_payload=True, prepared=False)
        prep = ops.RegistrationStep(src, tar, nl, want_payload=True)
        assert prep.prepared and not cold.prepared
        gen = torch.Generator().manual_seed(3)
        perm1 = torch.stack([torch.cat([torch.randperm(n, generator=gen), torch.zeros((n + 63) // 64 * 64 - n, dtype=torch.int64)])
                             for _ in range(B)]).to(torch.int32).cuda()
        _, other, _ = _pairs(999, B, n, m)
        stale = ops.cloud_order(ops.rigid_apply(other.reshape(B, -1, 3), cu(np.stack([_rot((0, 0, 1), 45)] * B)),
                                                cu(np.zeros((B, 3), np.float32))).reshape(B, n, 9))
        odd = {"perm": ops.RegistrationStep(src, tar, nl, want_payload=True, src_order=perm1),
               "stale": ops.RegistrationStep(src, tar, nl, want_payload=True, src_order=stale)}
        for i, (Rm, tv) in enumerate(poses):
            R, t = cu(np.stack([Rm] * B)), cu(np.stack([tv] * B))
            a = [x.clone() for x in cold(R, t, ln)]
            steps = [prep] + (list(odd.values()) if i in (0, 3) else [])
            for stp in steps:
                b_ = stp(R, t, ln)
                torch.cuda.synchronize()
                for x, y in ((a[0], b_[0]), (cold.st.med, stp.st.med), (a[4], b_[4]), (cold.st.bsum, stp.st.bsum),
                             (cold.st.count1, stp.st.count1), (cold.st.count2, stp.st.count2), (cold.st.tri1t, stp.st.tri1t)):
                    assert torch.equal(x, y)
                for x, y in ((a[1], b_[1]), (a[2], b_[2]), (a[3], b_[3])):
                    np.testing.assert_allclose(y.cpu().numpy(), x.cpu().numpy(), rtol=2e-5, atol=2e-6 * float(x.abs().max()))
            if i < 4:
                assert int(a[4][:, 1].min()) > 0  # lines are selected (the far translation may leave none)
    finally:
        ops.RegistrationStep.ONE_CALL = True


def test_kept_target_is_rebuilt_when_it_changes(L):
    """A target that has not changed since the previous call on the step's workspace is not rebuilt (RRL_F_TARGET_KEPT:
    same tensor, same version counter); an in-place edit, a new tensor (tar_tri=) or a carried-over scan in between
    make the next call build it again.  Every call equals the cold step of the same inputs bit for bit."""
    from rrl_hip import ops
    B, n, m, nl = 2, 900, 1100, 4000
    prs, src, tar = _pairs(360, B, n, m)
    ln = _lines(L, prs, nl)
    R, t = cu(np.stack([_rot((0, 0, 1), 5)] * B)), cu(np.zeros((B, 3), np.float32))
    cold = ops.RegistrationStep(src, tar, nl, prepared=False)
    prep = ops.RegistrationStep(src, tar, nl)

    def same():
        a = cold(R, t, ln)
        b_ = prep(R, t, ln)
        torch.cuda.synchronize()
        assert torch.equal(a[0], b_[0]) and torch.equal(cold.st.bsum, prep.st.bsum) and torch.equal(a[4], b_[4])
        assert torch.equal(cold.st.count2, prep.st.count2)
        assert torch.equal(_ptri_by_triangle(cold.st.ptri2), _ptri_by_triangle(prep.st.ptri2))

    same()
    assert prep._kept_key is not None
    kept_opts = prep._optr_kept
    same()  # second call: kept
    same()
    with torch.no_grad():
        tar[:, : m // 2] += 0.01  # in place: the version counter moves, the records are rebuilt
    same()
    assert not torch.equal(prep.st.ptri2[:, 0, :3], torch.zeros_like(prep.st.ptri2[:, 0, :3]))
    # the library itself writes the target through a raw pointer (ops.rigid_apply_into, in place): the version counter is
    # advanced by hand (ops._touched), so this is seen too
    v0 = tar._version
    ops.rigid_apply_into(tar.view(B, -1, 3), cu(np.stack([_rot((0, 1, 0), 3)] * B)), cu(np.full((B, 3), 0.01, np.float32)),
                         tar.view(B, -1, 3))
    assert tar._version > v0
    same()
    tar2 = tar.clone()
    with torch.no_grad():
        tar2[:, m // 2:] -= 0.02
    a = cold(R, t, ln, tar_tri=tar2)
    b_ = prep(R, t, ln, tar_tri=tar2)
    torch.cuda.synchronize()
    assert torch.equal(a[0], b_[0]) and torch.equal(_ptri_by_triangle(cold.st.ptri2), _ptri_by_triangle(prep.st.ptri2))
    a = cold(R, t, ln)
    b_ = prep(R, t, ln)  # kept again, now tar2's records
    torch.cuda.synchronize()
    assert torch.equal(a[0], b_[0]) and torch.equal(cold.st.bsum, prep.st.bsum)
    assert kept_opts is not None


@pytest.mark.parametrize("B,n,m,nl,prepared", [(2, 1200, 1000, 6000, True), (1, 1024, 1024, 20000, True), (3, 700, 900, 4000, False),
                                                (1, 5000, 4100, 3000, True)])
def test_chamfer_walk_rides_in_the_scan_launch(L, B, n, m, nl, prepared):
    """rrl_opts.chamfer (ops.ChamferRide; round 4b): the Chamfer walk between the moved source's and the target's first
    points is issued INSIDE the evaluation's culled-scan launch (cull_scan_chamfer_kernel) -- keys and value bit-identical
    to rrl_chamfer_from_loss after a plain evaluation, the loss evaluation itself unchanged; where the walk cannot ride (a
    carried-over target, a short line set, a dense scan mode) `done` stays 0 and chamfer_from_state launches it as before."""
    from rrl_hip import ops
    prs, src, tar = _pairs(400, B, n, m)
    ln = _lines(L, prs, nl)
    R = cu(np.stack([_rot((0.3, 0.5, 0.8), 9)] * B))
    t = cu(np.full((B, 3), 0.01, np.float32))
    o1, o2 = (ops.cloud_order(src), ops.cloud_order(tar)) if prepared else (None, None)
    ops.registration_loss(src, R, t, tar, ln, order1=o1, order2=o2)
    plain = ops.last_state()
    want = ops.chamfer_from_state(plain, keys=True)
    loss = ops.registration_loss(src, R, t, tar, ln, order1=o1, order2=o2, chamfer=True)[0]
    st = ops.last_state()
    assert st.cham_ride is not None and st.cham_ride.done
    got = ops.chamfer_from_state(st, keys=True)  # (no launch: the ride's own buffers)
    torch.cuda.synchronize()
    assert got[0].data_ptr() == st.cham_ride.val.data_ptr()
    for a, b_ in zip(want, got):
        assert torch.equal(a, b_)
    _same_evaluation(plain, st)
    assert torch.equal(loss, plain.loss.view(-1))
    # the one-call step object
    step = ops.RegistrationStep(src, tar, nl, chamfer=True, prepared=prepared, src_order=o1, tar_order=o2)
    for _ in range(3):  # (kept target from the second call on)
        out = step(R, t, ln)
        assert step.ride.done and torch.equal(step.chamfer_value, want[0]) and torch.equal(out[0], plain.loss.view(-1))
    ls = ops.LossStep(src, tar, nl, chamfer=True, prepared=prepared, src_order=o1, tar_order=o2)  # SURVEY 8(d)'s step, monitored
    for _ in range(2):
        lo = ls(R, t, ln)
        assert ls.ride.done and torch.equal(ls.chamfer_value, want[0]) and torch.equal(lo[0], plain.loss.view(-1))
    # a carried-over target (the iterative trainers: same target and lines, another pose): only the source is scanned, the
    # walk reads the target in the state that holds it
    R2 = cu(np.stack([_rot((0.1, 0.9, 0.2), 4)] * B))
    ops.registration_loss(src, R2, t, tar, ln, order1=o1, order2=o2, target_from=plain)
    want2 = ops.chamfer_from_state(ops.last_state(), keys=True)
    ops.registration_loss(src, R2, t, tar, ln, order1=o1, order2=o2, target_from=plain, chamfer=True)
    st2 = ops.last_state()
    assert st2.cham_ride is not None and st2.cham_ride.done
    for a, b_ in zip(want2, ops.chamfer_from_state(st2, keys=True)):
        assert torch.equal(a, b_)
    assert not torch.equal(want2[0], want[0])
    # refused: a short line set runs thinner workgroups; a dense scan mode has no culled scan
    ops.registration_loss(src, R, t, tar, ln[:, :300].contiguous(), chamfer=True)
    st3 = ops.last_state()
    assert st3.cham_ride is None and torch.equal(ops.chamfer_from_state(st3), want[0])
    if n <= 4096:
        ops.registration_loss(src, R, t, tar, ln, mode="strict", chamfer=True)
        st4 = ops.last_state()
        assert st4.cham_ride is None and torch.equal(ops.chamfer_from_state(st4), want[0])


@pytest.mark.parametrize("scale", [12.0, 300.0])
def test_prepared_build_at_the_demo_scale(L, scale):
    """Clouds scaled to the reference demo's data scale and beyond (the culled scan's NaN reach DEL is gathered through
    IDX there): prepared == plain == strict, NaN flag included."""
    from rrl_hip import ops
    prs, t1, t2 = _pairs(380, 1, 1024, 1024)
    t1, t2 = t1 * scale, t2 * scale
    p = prs[0]
    torch.manual_seed(5)
    ln = L.Random_uniform_distribution_lines_batch_efficient_resample(
        torch.tensor([[float(p["radius"]) * scale * 2.0]]), torch.from_numpy(p["center"] * scale).reshape(1, 3), 4000,
        t1[:, :, :3].contiguous(), t2[:, :, :3].contiguous(), "cuda")
    o1, o2 = ops.cloud_order(t1), ops.cloud_order(t2)
    plain = ops.loss_forward_raw(t1, t2, ln, mode="cull")
    strict = ops.loss_forward_raw(t1, t2, ln, mode="strict")
    prep = ops.loss_forward_raw(t1, t2, ln, mode="cull", opts=ops.make_opts(order1=o1, order2=o2))
    torch.cuda.synchronize()
    _same_evaluation(plain, prep)
    _same_evaluation(strict, prep)
    assert int(prep.status[1]) == 0


def test_prepared_autograd_ops(L):
    """ops.intersection_loss / ops.registration_loss take the orders too (order1= / order2=): same loss bits, gradients
    to the rounding of the scatter's atomics."""
    from rrl_hip import ops
    B, n, m, nl = 2, 700, 800, 3000
    prs, src, tar = _pairs(400, B, n, m)
    ln = _lines(L, prs, nl)
    o1, o2 = ops.cloud_order(src), ops.cloud_order(tar)
    res = []
    for kw in ({}, {"order1": o1, "order2": o2}):
        p1 = src.clone().requires_grad_(True)
        loss, info, _ = ops.intersection_loss(p1, tar, ln, **kw)
        loss.sum().backward()
        res.append((loss.detach().clone(), p1.grad.clone()))
    assert torch.equal(res[0][0], res[1][0])
    np.testing.assert_allclose(res[1][1].cpu().numpy(), res[0][1].cpu().numpy(), rtol=2e-5, atol=1e-7)
    R = cu(np.stack([_rot((0, 1, 0), 10)] * B)).requires_grad_(True)
    t = cu(np.zeros((B, 3), np.float32)).requires_grad_(True)
    out = []
    for kw in ({}, {"order1": o1, "order2": o2}):
        R.grad = t.grad = None
        loss, info, _ = ops.registration_loss(src, R, t, tar, ln, **kw)
        loss.sum().backward()
        out.append((loss.detach().clone(), R.grad.clone(), t.grad.clone()))
    assert torch.equal(out[0][0], out[1][0])
    for x, y in zip(out[0][1:], out[1][1:]):
        np.testing.assert_allclose(y.cpu().numpy(), x.cpu().numpy(), rtol=2e-5, atol=2e-6 * float(x.abs().max()))


@pytest.mark.parametrize("B,N,M", [(2, 1024, 700), (8, 4096, 4096), (1, 5000, 4100), (3, 65, 64), (1, 16400, 300)])
def test_chamfer_with_prepared_orders(L, B, N, M):
    """ops.chamfer(x, y, order_x, order_y): the per-call sort replaced by the prepared build (records at their sorted
    positions + tree refit).  The u64 keys (distance bits << 32 | first-occurrence argmin) of both directions and the
    value equal the sorting path's and the all-pairs kernel's bit for bit -- with orders taken from the point clouds, from
    the pseudo-triangles they are the first points of (the same order), and from ANOTHER pose of the clouds."""
    from rrl_hip import ops
    from test_gpu_parity import _chamfer_keys
    prs, t1, t2 = _pairs(600, B, N, M)
    x, y = t1[..., :3].contiguous(), t2[..., :3].contiguous()
    ox, oy = ops.cloud_order(x), ops.cloud_order(y)
    assert torch.equal(ox, ops.cloud_order(t1)) and torch.equal(oy, ops.cloud_order(t2))
    Rm = cu(np.stack([_rot((0.3, 0.5, 0.8), 77)] * B))
    moved = ops.rigid_apply(x, Rm, cu(np.full((B, 3), 0.4, np.float32)))
    o_other = ops.cloud_order(moved)

    def keys(order_x, order_y):
        ops._Chamfer.apply(x, y, order_x, order_y)  # warm
        dev = x.device
        bx = torch.empty(B, N, dtype=torch.int64, device=dev)
        by = torch.empty(B, M, dtype=torch.int64, device=dev)
        val = torch.empty(1, device=dev)
        nb = int(ops._lib.load().rrl_chamfer_workspace_bytes(B, N, M))
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        ops._run(dev, "rrl_chamfer_tree_fwd_ex", ops._p(x), ops._p(y), ops._p(ws), nb, ops._p(bx), ops._p(by), ops._p(val), B, N, M,
                 ops._p(order_x), ops._p(order_y), None, 0)
        torch.cuda.synchronize()
        return bx, by, val

    plain = keys(None, None)
    for oxx, oyy in ((ox, oy), (o_other, oy)):
        got = keys(oxx, oyy)
        for a, b_ in zip(plain, got):
            assert torch.equal(a, b_)
    brute = _chamfer_keys(x.cpu().numpy(), y.cpu().numpy(), False)
    np.testing.assert_array_equal(plain[0].cpu().numpy().view(np.uint64), brute[0])
    np.testing.assert_array_equal(plain[1].cpu().numpy().view(np.uint64), brute[1])
    v = ops.chamfer(x, y, order_x=ox, order_y=oy)
    assert float(v) == float(ops.chamfer(x, y))


@pytest.mark.parametrize("n", [1, 7, 64, 65, 300, 1024, 3000, 4096, 4097, 9000, 20000])
def test_cloud_order_equals_its_host_twin(L, n):
    """rrl_cloud_order (GPU: registers / in-wavefront exchanges / LDS / global passes, by stride) and
    pre_dataloader.kd_order (numpy, recursive) build the SAME order: median splits of aligned power-of-two windows along
    the longest axis of their records' box, ties by index, pads last -- on random clouds, a cloud with many duplicate
    points and a flat one; for pseudo-triangles (B, n, 9) and for their first points (B, n, 3)."""
    from rrl_hip import ops
    import pre_dataloader as P
    rng = np.random.default_rng(n)
    clouds = [rng.standard_normal((n, 3)).astype(np.float32) * np.array([1.0, 0.6, 0.3], np.float32),
              np.repeat(rng.standard_normal(((n + 3) // 4, 3)).astype(np.float32), 4, 0)[:n],      # duplicates: ties by index
              np.concatenate([rng.standard_normal((n, 2)), np.zeros((n, 1))], 1).astype(np.float32)]  # flat: a zero extent
    pts = np.stack(clouds)
    tri = np.concatenate([pts, rng.standard_normal((3, n, 6)).astype(np.float32)], -1)
    got3 = ops.cloud_order(cu(pts)).cpu().numpy()
    got9 = ops.cloud_order(cu(tri)).cpu().numpy()
    np.testing.assert_array_equal(got3, got9)
    for b in range(3):
        np.testing.assert_array_equal(got3[b], P.kd_order(pts[b]))


# ---------------------------------------------------------------------------------------------------------------------
# round 5: the section-8(d) step's shard payload, order validation, explicit target invalidation, torch-internals fallbacks
@pytest.mark.parametrize("B,n,m,nl,prepared,pose", [(3, 900, 1100, 4000, True, True), (2, 700, 600, 3000, False, True),
                                                     (2, 500, 400, 800, True, True), (2, 600, 500, 2500, True, False),
                                                     (8, 4096, 4096, 10000, True, True), (40, 200, 180, 8000, True, True),
                                                     (36, 300, 200, 7200, False, False)])
def test_loss_step_payload(L, B, n, m, nl, prepared, pose):
    """ops.LossStep(want_payload=True) (include/rrl.h rrl_opts.payload): after every step .payload ==
    [sum of the valid losses, #valid, 0 x 12] -- riding in the reduce's launch (2 .. 32 line tiles: the tail kernel; grids beyond its 256 workgroups: the
    exchange reduce's last arrivers) or in the single-tile kernel (one tile of lines), with and without the rigid apply in front, on the first (building) and on later
    (kept-target) calls -- and the step's loss / gradient are those of the step without a payload, bit for bit / to the
    rounding of the scatter's atomics."""
    from rrl_hip import ops
    prs, src, tar = _pairs(520, B, n, m)
    ln = _lines(L, prs, nl)
    R = cu(np.stack([_rot((0, 0, 1), 4 + b) for b in range(B)])) if pose else None
    t = cu(np.full((B, 3), 0.01, np.float32)) if pose else None
    plain = ops.LossStep(src, tar, nl, prepared=prepared)
    pay = ops.LossStep(src, tar, nl, prepared=prepared, want_payload=True)
    for it in range(3):
        a = plain(R, t, ln)
        b_ = pay(R, t, ln)
        torch.cuda.synchronize()
        assert torch.equal(a[0], b_[0]) and torch.equal(a[2], b_[2])
        ga, gb = a[1], b_[1]
        assert bool(((ga - gb).abs() <= 2e-5 * ga.abs() + 2e-6 * float(ga.abs().max())).all())
        valid = b_[2][:, 0] > 0
        want_sum = float(b_[0][valid].double().sum())
        p = pay.payload.cpu().numpy()
        assert p[1] == float(valid.sum()) and np.all(p[2:] == 0.0), p
        assert abs(p[0] - want_sum) <= 2e-6 * max(1.0, abs(want_sum)), (p[0], want_sum)
    assert int(valid.sum()) == B


def test_orders_are_validated_where_they_enter(L):
    """An order reaches the kernels as a raw int32 [B][64 ceil(n / 64)] pointer on the op's GPU: every Python entry that
    takes one refuses anything else -- dtype, batch dimension, padding, contiguity."""
    from rrl_hip import ops, callsites
    B, n, m, nl = 2, 300, 280, 1500
    prs, src, tar = _pairs(530, B, n, m)
    ln = _lines(L, prs, nl)
    R, t = cu(np.stack([np.eye(3, dtype=np.float32)] * B)), cu(np.zeros((B, 3), np.float32))
    o1, o2 = ops.cloud_order(src), ops.cloud_order(tar)
    good = ops.registration_loss(src, R, t, tar, ln, order1=o1, order2=o2)[0].clone()
    bad = [o1.to(torch.int64), o1[:1].contiguous(), o1[:, :256].contiguous(), o1.t().contiguous().t(), o1.cpu()]
    for b_ in bad:
        with pytest.raises(ValueError, match="order1"):
            ops.registration_loss(src, R, t, tar, ln, order1=b_, order2=o2)
        with pytest.raises(ValueError, match="order1"):
            ops.intersection_loss(src, tar, ln, order1=b_, order2=o2)
        with pytest.raises(ValueError):
            ops.RegistrationStep(src, tar, nl, src_order=b_, tar_order=o2)
        with pytest.raises(ValueError):
            ops.LossStep(src, tar, nl, src_order=b_, tar_order=o2)
        # the trainers' dict: an unusable order is ignored (the fused op sorts), never handed on
        assert callsites._orders({'order_src': b_, 'order_tar': o2}, n, m, B, src.device) == (None, None)
    assert callsites._orders({'order_src': o1, 'order_tar': o2}, n, m, B, src.device)[0] is o1
    assert torch.equal(ops.registration_loss(src, R, t, tar, ln, order1=o1, order2=o2)[0], good)
    try:  # the debugging aid: a host-side permutation check
        ops.ORDER_DEBUG = True
        dup = o1.clone()
        dup[0, 1] = dup[0, 0]
        with pytest.raises(ValueError, match="permutation"):
            ops.intersection_loss(src, tar, ln, order1=dup, order2=o2)
        ops.intersection_loss(src, tar, ln, order1=o1, order2=o2)
    finally:
        ops.ORDER_DEBUG = False


def test_invalidate_target_and_failed_calls_keep_nothing(L):
    """A write that bypasses torch's version counter AND the library (tar.data.copy_ bumps it; a raw alias does not) is the
    one thing a step cannot see: invalidate_target() / keep_target = False are the switches.  The kept key is set only after
    a call has been issued: a refused call changes nothing."""
    from rrl_hip import ops
    B, n, m, nl = 2, 800, 900, 3500
    prs, src, tar = _pairs(540, B, n, m)
    ln = _lines(L, prs, nl)
    R, t = cu(np.stack([_rot((0, 0, 1), 5)] * B)), cu(np.zeros((B, 3), np.float32))
    for cls in (ops.RegistrationStep, ops.LossStep):
        cold, prep = cls(src, tar, nl, prepared=False), cls(src, tar, nl)
        prep(R, t, ln); prep(R, t, ln)
        assert prep._kept_key is not None
        with pytest.raises(ValueError):  # wrong line shape: refused before anything is issued (the workspace is untouched;
            prep(R, t, ln[:, :100].contiguous())  # a key is only ever SET after the C call has been issued)
        a, b_ = cold(R, t, ln), prep(R, t, ln)
        torch.cuda.synchronize()
        assert torch.equal(a[0], b_[0])
        # a write NOBODY's bookkeeping sees: through a second tensor on the same storage (its own version counter; what a custom
        # kernel or another raw-pointer library does).  The kept records are stale -- the documented blind spot -- until
        # invalidate_target()
        alias = torch.empty(0, device="cuda").set_(tar.untyped_storage(), 0, tar.shape, tar.stride())
        v0 = tar._version
        alias.mul_(1.01)
        assert tar._version == v0
        a, b_ = cold(R, t, ln), prep(R, t, ln)
        torch.cuda.synchronize()
        assert not torch.equal(a[0], b_[0])  # (stale target)
        prep.invalidate_target()
        b_ = prep(R, t, ln)
        torch.cuda.synchronize()
        assert torch.equal(a[0], b_[0])
        # the explicit switches
        prep.invalidate_target()
        assert prep._kept_key is None
        prep.keep_target = False
        for _ in range(2):
            a, b_ = cold(R, t, ln), prep(R, t, ln)
            torch.cuda.synchronize()
            assert torch.equal(a[0], b_[0]) and prep._kept_key is None
        with torch.no_grad():
            tar.div_(1.01)  # (the next class starts from the same target, up to rounding)


def test_torch_internals_fallbacks(L, monkeypatch):
    """rrl_hip.ops uses three torch internals for speed (INTERNALS); without them it degrades to public API with one
    warning, never an AttributeError: same loss bits, same gradients, the trainers' loop served per call."""
    import warnings
    from rrl_hip import ops
    B, n, m, nl = 3, 600, 500, 2500
    prs, src, tar = _pairs(550, B, n, m)
    ln = _lines(L, prs, nl)
    assert ops.INTERNALS["raw_stream"] and ops.INTERNALS["version_bump"] and ops.INTERNALS["tensor_base"]

    def trainer_loop():
        p1 = src.clone().requires_grad_(True)
        total = 0
        for j in range(B):
            one = L.cal_loss_intersection_batch_whole_median_pts_lines(1, 1, 5, 5, p1[j:j + 1], tar[j:j + 1], ln[j:j + 1], "cuda")
            total = total + one
        total.backward()
        return total.detach().clone(), p1.grad.clone()

    def kept_step():
        tk = tar.clone()  # (its own target: the trainer loops above and below must see the same one)
        step = ops.RegistrationStep(src, tk, nl)
        R, t = cu(np.stack([_rot((0, 1, 0), 3)] * B)), cu(np.zeros((B, 3), np.float32))
        step(R, t, ln)
        out = step(R, t, ln)[0].clone()
        ops.rigid_apply_into(tk.view(B, -1, 3), R, t, tk.view(B, -1, 3))  # raw-pointer write into the kept target
        new = step(R, t, ln)[0].clone()
        ref = ops.RegistrationStep(src, tk, nl, prepared=False)(R, t, ln)[0].clone()
        assert not torch.equal(out, new)
        return out, new, ref

    ops.dropin_batch_clear()
    base_loss, base_grad = trainer_loop()
    s0 = dict(ops.dropin_batch_stats)
    assert s0["evaluations"] >= 1
    monkeypatch.setattr(ops, "DROPIN_BATCH", False)  # RRL_DROPIN_BATCH=0: the literal per-call path
    off_loss, off_grad = trainer_loop()
    assert torch.equal(off_loss, base_loss)
    assert bool(((off_grad - base_grad).abs() <= 2e-5 * base_grad.abs() + 2e-6 * float(base_grad.abs().max())).all())
    monkeypatch.setattr(ops, "DROPIN_BATCH", True)
    _, new, ref = kept_step()
    assert torch.equal(new, ref)
    # now without the internals
    monkeypatch.setattr(ops, "_raw_stream_fn", None)
    monkeypatch.setattr(ops, "_set_version_fn", None)
    monkeypatch.setitem(ops.INTERNALS, "tensor_base", False)
    ops._warned.clear()
    ops.dropin_batch_clear()
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        f_loss, f_grad = trainer_loop()
        _, new, ref = kept_step()  # the kept target is still invalidated by the library's own write map
        torch.cuda.synchronize()
    assert torch.equal(f_loss, base_loss) and torch.equal(new, ref)
    assert bool(((f_grad - base_grad).abs() <= 2e-5 * base_grad.abs() + 2e-6 * float(base_grad.abs().max())).all())
    msgs = " ".join(str(w.message) for w in wlist)
    assert "_cuda_getCurrentRawStream" in msgs and "_unsafe_set_version_counter" in msgs and "Tensor._base" in msgs
    assert sum("_cuda_getCurrentRawStream" in str(w.message) for w in wlist) == 1  # one warning per missing symbol


# ---------------------------------------------------------------------------------------------------------------------
# round 5: multi-pose evaluation (include/rrl.h rrl_opts.problems) -- the iterative trainers' poses in ONE set of launches
@pytest.mark.parametrize("Bt,k,n,m,nl,prepared", [(3, 2, 900, 1100, 4000, True), (2, 3, 700, 600, 3000, False),
                                                  (8, 2, 4096, 4096, 10000, True), (1, 3, 1024, 1024, 20000, True),
                                                  (2, 2, 500, 400, 800, True), (1, 2, 5000, 4100, 1500, False)])
def test_multi_pose_equals_pose_after_pose(L, Bt, k, n, m, nl, prepared):
    """k poses of each of Bt problems as ONE evaluation of k * Bt instances (R, t hold k * Bt poses; source, target, lines
    and orders have Bt entries; the target is scanned once per problem) against the k evaluations pose after pose (the second
    and later ones with the target's scan carried over, as the trainers' fragments did until round 4): loss, info, median,
    bucket sums, hit lists of the sources BIT-identical per instance; (dR, dt) to the rounding of the float atomics; the
    Chamfer monitor per pose (ops.chamfer_group_means, its walk riding in the scan launch) equals each single evaluation's."""
    from rrl_hip import ops
    prs, src, tar = _pairs(600, Bt, n, m)
    ln = _lines(L, prs, nl)
    gen = torch.Generator().manual_seed(5)
    from LieAlgebra import se3
    Rs, ts = [], []
    for i in range(k):
        R, t = se3.exp3(0.04 * torch.randn(Bt, 6, generator=gen))
        Rs.append(R.cuda().contiguous().requires_grad_(True)); ts.append(t.cuda().contiguous().requires_grad_(True))
    o1, o2 = (ops.cloud_order(src), ops.cloud_order(tar)) if prepared else (None, None)
    singles, first = [], None
    for i in range(k):
        loss, info, _ = ops.registration_loss(src, Rs[i], ts[i], tar, ln, order1=o1, order2=o2, target_from=first, chamfer=True)
        st = ops.last_state()
        first = first or st
        cham = ops.chamfer_from_state(st).clone()
        loss.sum().backward()
        singles.append((loss.detach().clone(), info.clone(), st.med.clone(), st.bsum.clone(), st.count1.clone(),
                        _hits_sorted(st, 1)[1].clone(), Rs[i].grad.clone(), ts[i].grad.clone(), cham))
        Rs[i].grad = ts[i].grad = None
    Rm = torch.cat([r.detach() for r in Rs]).requires_grad_(True)
    tm = torch.cat([t.detach() for t in ts]).requires_grad_(True)
    loss, info, _ = ops.registration_loss(src, Rm, tm, tar, ln, order1=o1, order2=o2, chamfer=True)
    st = ops.last_state()
    assert loss.shape == (k * Bt,) and st.dims[0] == k * Bt
    cms = ops.chamfer_group_means(st, k)
    loss.sum().backward()
    torch.cuda.synchronize()
    assert int(st.count2[Bt:].abs().sum()) == 0  # the target was scanned for the first Bt instances only
    hits1 = _hits_sorted(st, 1)[1]
    for i in range(k):
        sl = slice(i * Bt, (i + 1) * Bt)
        l1, i1, med, bsum, c1, h1, gR, gt, cham = singles[i]
        assert torch.equal(loss.detach()[sl], l1) and torch.equal(info[sl], i1) and torch.equal(st.med[sl], med)
        assert torch.equal(st.bsum[sl], bsum) and torch.equal(st.count1[sl], c1) and torch.equal(hits1[sl], h1)
        for got, want in ((Rm.grad[sl], gR), (tm.grad[sl], gt)):
            assert bool(((got - want).abs() <= 2e-5 * want.abs() + 2e-6 * float(want.abs().max())).all())
        assert abs(float(cms[i]) - float(cham)) <= 1e-6 * max(1e-6, abs(float(cham)))
    assert int((info[:, 0] > 0).sum()) == k * Bt
    # the thin steps with poses = k (no autograd, one C call): the same instances
    Rd, td = Rm.detach(), tm.detach()
    rs = ops.RegistrationStep(src, tar, nl, prepared=prepared, src_order=o1, tar_order=o2, poses=k)
    ls = ops.LossStep(src, tar, nl, prepared=prepared, src_order=o1, tar_order=o2, poses=k, want_payload=True)
    for _ in range(2):  # (the second call: kept target)
        r_out, l_out = rs(Rd, td, ln), ls(Rd, td, ln)
    torch.cuda.synchronize()
    assert torch.equal(r_out[0], loss.detach()) and torch.equal(l_out[0], loss.detach()) and torch.equal(r_out[4], info)
    assert bool(((r_out[1] - Rm.grad).abs() <= 2e-5 * Rm.grad.abs() + 2e-6 * float(Rm.grad.abs().max())).all())
    assert float(ls.payload[1]) == k * Bt and abs(float(ls.payload[0]) - float(loss.detach().double().sum())) <= 2e-6 * float(loss.detach().sum())
    for i in range(k):
        one = ops.LossStep(src, tar, nl, prepared=prepared, src_order=o1, tar_order=o2)
        lo, go, _ = one(Rs[i].detach(), ts[i].detach(), ln)
        gm = l_out[1][i * Bt:(i + 1) * Bt]
        assert torch.equal(lo, loss.detach()[i * Bt:(i + 1) * Bt])
        assert bool(((gm - go).abs() <= 2e-5 * go.abs() + 2e-6 * float(go.abs().max())).all())


def test_multi_pose_argument_errors(L):
    from rrl_hip import ops, RRLError
    prs, src, tar = _pairs(610, 2, 300, 280)
    ln = _lines(L, prs, 1500)
    R = cu(np.stack([np.eye(3, dtype=np.float32)] * 4)); t = cu(np.zeros((4, 3), np.float32))
    assert ops.registration_loss(src, R, t, tar, ln)[0].shape == (4,)
    with pytest.raises(ValueError):  # 3 poses for 2 problems
        ops.registration_loss(src, R[:3], t[:3], tar, ln)
    with pytest.raises(ValueError):  # a dense scan mode
        ops.registration_loss(src, R, t, tar, ln, mode="strict")
    s2 = src.clone().requires_grad_(True)
    out = ops.registration_loss(s2, R.clone().requires_grad_(True), t, tar, ln)[0]
    with pytest.raises(RRLError, match="multi-pose"):
        out.sum().backward()


# ---------------------------------------------------------------------------------------------------------------------
# round 5: the FAT geometry variant of the culled scan (csrc/rrl_cull_scan.inc scan16: slices of 16 supergroups, the lines in
# registers) -- chosen automatically for deep grids (B >= 12 at C2's shape), forced here by RRL_CULL_FAT on small ones
def _with_fat(flag, fn):
    import os
    old = os.environ.get("RRL_CULL_FAT")
    os.environ["RRL_CULL_FAT"] = flag
    try:
        out = fn()
        torch.cuda.synchronize()
        return out
    finally:
        if old is None:
            del os.environ["RRL_CULL_FAT"]
        else:
            os.environ["RRL_CULL_FAT"] = old


@pytest.mark.parametrize("B,n,m,nl,prepared,scale", [(2, 1200, 1000, 6000, True, 1.0), (1, 5000, 4100, 1500, False, 1.0),
                                                     (3, 4096, 4096, 10000, True, 1.0), (1, 16384, 16384, 1024, True, 1.0),
                                                     (1, 1024, 1024, 4000, True, 12.0), (2, 700, 2100, 3000, False, 300.0),
                                                     (16, 4096, 4096, 10000, True, 1.0)])
def test_fat_scan_variant_equals_the_lean_one(L, B, n, m, nl, prepared, scale):
    """Labels, hit lists, median, bucket sums, loss, NaN flag and the in-kernel work counters' totals that do not depend on
    the slicing (candidates resolved, fallback wavefronts) of the fat variant equal the lean one's and the strict scan's --
    sorted and prepared builds, unit scale and the demo's (NaN-widened walk), with the Chamfer walk riding in the launch."""
    from rrl_hip import ops
    prs, t1, t2 = _pairs(700, B, n, m)
    t1, t2 = t1 * scale, t2 * scale
    lns = []
    for b, p in enumerate(prs):
        torch.manual_seed(40 + b)
        lns.append(L.Random_uniform_distribution_lines_batch_efficient_resample(
            torch.tensor([[float(p["radius"]) * scale * (2.0 if scale > 1 else 1.0)]]), torch.from_numpy(p["center"] * scale).reshape(1, 3),
            nl, t1[b:b + 1, :, :3].contiguous(), t2[b:b + 1, :, :3].contiguous(), "cuda")[0])
    ln = torch.stack(lns)
    o1, o2 = (ops.cloud_order(t1), ops.cloud_order(t2)) if prepared else (None, None)

    def run(counters=None, chamfer=None):
        return ops.loss_forward_raw(t1, t2, ln, mode="cull", opts=ops.make_opts(order1=o1, order2=o2, counters=counters, chamfer=chamfer))
    lean = _with_fat("0", run)
    fat = _with_fat("1", run)
    strict = ops.loss_forward_raw(t1, t2, ln, mode="strict")
    torch.cuda.synchronize()
    _same_evaluation(lean, fat)
    _same_evaluation(strict, fat)
    assert int(fat.status[1]) == int(lean.status[1])
    tl, tf = torch.zeros(1 << 16, 16, dtype=torch.int64, device="cuda"), torch.zeros(1 << 16, 16, dtype=torch.int64, device="cuda")
    _with_fat("0", lambda: run(counters=tl))
    _with_fat("1", lambda: run(counters=tf))
    sl, sf = tl.sum(0), tf.sum(0)
    assert int(sf[5]) > 0 and int(sf[5]) < int(sl[5]) or n <= 512   # fewer, fatter wavefronts really ran
    assert int(sl[6]) == int(sf[6]) == 0
    if nl >= 1024 and max(n, m) <= 65536:  # the riding walk: same keys and value from either variant
        ra, rb = ops.ChamferRide(B, n, m, t1.device), ops.ChamferRide(B, n, m, t1.device)
        a = _with_fat("0", lambda: run(chamfer=ra.arm()))
        b_ = _with_fat("1", lambda: run(chamfer=rb.arm()))
        assert ra.done and rb.done and torch.equal(ra.bx, rb.bx) and torch.equal(ra.by, rb.by) and torch.equal(ra.val, rb.val)
        _same_evaluation(a, b_)


def test_fat_variant_is_chosen_for_deep_grids(L):
    """The automatic choice (rrl_launch_cull_scan): B = 16 at C2's shape runs the fat variant (half as many wavefronts as the
    lean one would), C2 itself the lean one -- seen through the wavefront counter -- with identical results."""
    from rrl_hip import ops
    for B, want_fat in ((16, True), (8, False)):
        prs, t1, t2 = _pairs(720, B, 4096, 4096)
        ln = _lines(L, prs, 10000)
        tab = torch.zeros(1 << 17, 16, dtype=torch.int64, device="cuda")
        auto = ops.loss_forward_raw(t1, t2, ln, mode="cull", opts=ops.make_opts(counters=tab))
        torch.cuda.synchronize()
        waves = int(tab.sum(0)[5])
        lean_waves = 2 * B * 79 * 8
        assert waves == (lean_waves // 2 if want_fat else lean_waves), (B, waves, lean_waves)
        other = _with_fat("0" if want_fat else "1", lambda: ops.loss_forward_raw(t1, t2, ln, mode="cull"))
        _same_evaluation(auto, other)
