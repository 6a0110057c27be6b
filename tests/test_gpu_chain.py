"""Chained steps (round 6; include/rrl.h RRL_F_CHAIN / RRL_F_CHAINED, csrc/rrl_cull_scan.inc cull_scan_build_kernel).

A loop that evaluates the loss again and again on one workspace with a kept target (the demo,
code/test_demo_optimized_Lie_Algebra.py:48-62: new lines and a new pose every step, the target never moves) runs, from its
second step on, the source's records, the target's scan and the source's scan as ONE launch: the two scans of
code/loss.py:181-184 are independent, and the target's needs nothing the records produce.  The bar is the one of the
prepared build: hit counts, hit lists, median, bucket sums, info and loss BIT-IDENTICAL to the unchained step (whose own
parity with the oracle and the reference fixtures is test_gpu_parity's business), points1.grad to the rounding of the
scatter's float atomics -- for new lines and poses in every step, across target changes, with the scan's counters switched
on in between, at unit scale and at the demo's (the NaN-widened walk), for clouds beyond 4096 triangles and ragged shapes.
"""
import numpy as np
import pytest
import torch

from test_gpu_parity import cu
from test_gpu_prepared import _lines, _pairs, _rot

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import loss
    from rrl_hip import _lib
    _lib.load()
    assert torch.cuda.is_available()
    return loss


def _new_lines(L, prs, nl, it):
    out = []
    for b, p in enumerate(prs):
        torch.manual_seed(1000 + 37 * it + b)  # (the CPU seed selects the sampler's uniform streams)
        out.append(L.Random_uniform_distribution_lines_batch_efficient_resample(
            torch.tensor([[float(p["radius"])]]), torch.from_numpy(np.asarray(p["center"], np.float32)).reshape(1, 3), nl,
            cu(p["src"])[None], cu(p["tar"])[None], "cuda")[0])
    return torch.stack(out)


def _poses(B, it):
    R = cu(np.stack([_rot((0, 0, 1) if (b + it) % 2 else (1, 0, 0), 2.0 + 1.5 * it + b) for b in range(B)]))
    t = cu(np.full((B, 3), 0.01 * (it + 1), np.float32))
    return R, t


def _snapshot(step, out):
    """Everything a step leaves that does not depend on the chain: the per-line stage's view of the hits (KJ = k | j << 4 per
    line, the ascending hit lists of the selected lines), median, buckets, info, loss, gradient."""
    st = step.st
    kj = st.kj.clone()
    sel = kj != 0
    k, j = (kj & 15).long(), (kj >> 4).long()
    a4 = torch.arange(4, device=kj.device)
    hs1 = torch.where(sel[..., None] & (a4 < k[..., None]), st.hs1, torch.full_like(st.hs1, -1))
    hs2 = torch.where(sel[..., None] & (a4 < j[..., None]), st.hs2, torch.full_like(st.hs2, -1))
    return dict(loss=out[0].clone(), grad=out[1].clone(), info=out[2].clone(), kj=kj, hs1=hs1, hs2=hs2, med=st.med.clone(),
                bsum=st.bsum.clone(), bcnt=st.bcnt.clone())


def _assert_same(a, b, what):
    for key in ("loss", "info", "kj", "hs1", "hs2", "med", "bsum", "bcnt"):
        assert torch.equal(a[key], b[key]), (what, key)
    ga, gb = a["grad"], b["grad"]
    assert torch.equal(ga.abs().sum(-1) > 0, gb.abs().sum(-1) > 0), what
    assert bool(((ga - gb).abs() <= 2e-5 * ga.abs() + 2e-6 * float(ga.abs().max())).all()), what


@pytest.mark.parametrize("B,n,m,nl,scale,fusable", [  # (fusable: rrl_cull_scan_can_fuse's measured rule, csrc/rrl_cull.hip)
    (2, 1200, 1000, 6000, 1.0, True), (8, 4096, 4096, 10000, 1.0, True), (1, 1024, 1024, 20000, 1.0, True),
    (3, 700, 900, 4000, 1.0, True), (1, 5000, 4100, 3000, 1.0, False), (2, 1000, 1200, 5000, 12.0, True),
    (16, 500, 400, 2500, 1.0, True), (12, 4096, 4096, 10000, 1.0, True),  # (the last one: the fat scan variant, scan16)
    (40, 300, 260, 7000, 1.0, True), (48, 2048, 2048, 10000, 1.0, True),  # (beyond the tail kernel's grid: exchange reduce + scatter launch; the second: scan16 too)
    (5, 330, 260, 1100, 1.0, False),  # (a grid so thin that the scan runs fewer than 8 wavefronts per workgroup: the plain four launches)
    (2, 9000, 9100, 4096, 1.0, False)])  # (few, large clouds on a shallow grid: the source workgroups would wait for 18 records pieces each -- measured slower fused: plain)
def test_chained_steps_equal_unchained_steps(L, B, n, m, nl, scale, fusable):
    from rrl_hip import ops
    prs, src, tar = _pairs(900, B, n, m)
    src, tar = src * scale, tar * scale
    for p in prs:
        p.update(radius=float(p["radius"]) * scale, center=p["center"] * scale, src=p["src"] * scale, tar=p["tar"] * scale)
    plain = ops.LossStep(src, tar, nl, want_payload=True)
    plain.chain = False
    chained = ops.LossStep(src, tar, nl, want_payload=True)
    fused = 0
    for it in range(5):
        ln = _new_lines(L, prs, nl, it)  # (new lines in every step)
        R, t = _poses(B, it)
        chained.st.lmax.fill_(-7.0)  # the separate records launch rewrites LMAX; the fused launch has no use for it
        a = _snapshot(plain, plain(R, t, ln))
        b = _snapshot(chained, chained(R, t, ln))
        torch.cuda.synchronize()
        _assert_same(a, b, it)
        assert torch.equal(plain.payload[:2], chained.payload[:2]) or abs(float(plain.payload[0] - chained.payload[0])) < 1e-5
        was_fused = bool((chained.st.lmax == -7.0).all())
        assert chained.fused == was_fused and plain.fused is False  # the library's own word (rrl_opts.chain_left bit 1)
        fused += was_fused
        assert was_fused == (it > 0 and fusable), (it, "the second and later steps of a chain run the fused launch")
        # what a chained step leaves behind: cleared counts and CHAIN words; the unchained one keeps its counts
        assert int(chained.st.count1.abs().max()) == 0 and int(chained.st.count2.abs().max()) == 0
        assert int(chained.st.chain.abs().max()) == 0
        assert int(plain.st.count1.max()) > 0
    assert fused == (4 if fusable else 0)
    assert int(b["info"][:, 1].min()) > 0 and int(b["info"][:, 3].max()) == 0


def test_chain_is_broken_by_a_new_target_and_resumes(L):
    """The kept-target rules of the prepared step hold for the chain: an in-place write to the target (torch's version
    counter), invalidate_target() and keep_target = False each break it for one step -- that step rebuilds the target with
    the plain four launches -- and the chain resumes behind it; a step with the scan's counters on is unfused but keeps the
    chain; chain = False stops it."""
    from rrl_hip import ops
    B, n, m, nl = 2, 1100, 900, 5000
    prs, src, tar = _pairs(930, B, n, m)
    ln = _lines(L, prs, nl)
    ref = ops.LossStep(src, tar.clone(), nl)
    ref.chain = False
    st = ops.LossStep(src, tar, nl)
    R, t = _poses(B, 0)

    def run(expect_fused, what):
        st.st.lmax.fill_(-7.0)
        ref.tar.copy_(st.tar)
        a = _snapshot(ref, ref(R, t, ln))
        b = _snapshot(st, st(R, t, ln))
        torch.cuda.synchronize()
        _assert_same(a, b, what)
        assert bool((st.st.lmax == -7.0).all()) == expect_fused, what

    run(False, "first")
    run(True, "second")
    tar.mul_(1.01)  # the target moved (version counter)
    run(False, "after an in-place write")
    run(True, "resumed")
    st.invalidate_target()
    run(False, "invalidated")
    run(True, "resumed again")
    ops.scan_counters(True)
    try:
        run(False, "counters on: the counting instantiation of the plain scan")
    finally:
        ops.scan_counters(False)
    run(True, "the counted step left the workspace chain-clean too")
    st.chain = False
    run(False, "chain off")
    st.chain = True
    run(False, "chain on again: the step before it left its counts in place")
    run(True, "and resumed")
    import os
    os.environ["RRL_CHAIN"] = "0"  # (read per call: the process-wide switch; the step stays chain-clean, only the fusing stops)
    try:
        run(False, "RRL_CHAIN=0")
        assert st.fused is False and st._chain_ready
    finally:
        del os.environ["RRL_CHAIN"]
    run(True, "RRL_CHAIN unset again")
    assert st.fused is True
    st.keep_target = False
    run(False, "keep_target off")


def test_chained_step_under_graph_replay(L):
    """A chained step captured into a hipGraph replays as a chained step: every replay finds the counts its predecessor
    cleared (bench.py's --issue graph)."""
    from rrl_hip import ops
    from rrl_hip.graph import GraphedStep
    B, n, m, nl = 2, 1000, 1000, 5000
    prs, src, tar = _pairs(940, B, n, m)
    ln = _lines(L, prs, nl)
    R, t = _poses(B, 1)
    ref = ops.LossStep(src, tar, nl)
    ref.chain = False
    want = _snapshot(ref, ref(R, t, ln))
    st = ops.LossStep(src, tar, nl)
    st(R, t, ln)
    g = GraphedStep(lambda: st(R, t, ln))
    for _ in range(4):
        st.st.lmax.fill_(-7.0)
        out = g()
        torch.cuda.synchronize()
        _assert_same(want, _snapshot(st, out), "replay")
        assert bool((st.st.lmax == -7.0).all())


@pytest.mark.parametrize("B,n,m,nl", [(2, 1200, 1000, 6000), (8, 4096, 4096, 10000), (1, 1024, 1024, 20000)])
def test_chained_registration_steps(L, B, n, m, nl):
    """The fused training op (ops.RegistrationStep: backward straight to (dR, dt) in the tail kernel) chains the same way:
    loss / info / median bit for bit, (dR, dt) and the shard payload to the rounding of their float atomics."""
    from rrl_hip import ops
    prs, src, tar = _pairs(960, B, n, m)
    plain = ops.RegistrationStep(src, tar, nl, want_payload=True)  # (chain=False is this class's default: its state is handed on as target_from=)
    chained = ops.RegistrationStep(src, tar, nl, want_payload=True, chain=True)
    for it in range(4):
        ln = _new_lines(L, prs, nl, it)
        R, t = _poses(B, it)
        chained.st.lmax.fill_(-7.0)
        a = [x.clone() for x in plain(R, t, ln)]
        b = [x.clone() for x in chained(R, t, ln)]
        torch.cuda.synchronize()
        assert torch.equal(a[0], b[0]) and torch.equal(a[4], b[4]) and torch.equal(plain.st.med, chained.st.med)
        assert torch.equal(plain.st.kj, chained.st.kj) and torch.equal(plain.st.bsum, chained.st.bsum)
        for x, y in ((a[1], b[1]), (a[2], b[2]), (a[3], b[3])):
            np.testing.assert_allclose(y.cpu().numpy(), x.cpu().numpy(), rtol=2e-4, atol=2e-6 * float(x.abs().max()))
        assert bool((chained.st.lmax == -7.0).all()) == (it > 0)
        assert int(chained.st.count1.abs().max()) == 0 and int(chained.st.chain.abs().max()) == 0
    with pytest.raises(ValueError):
        plain(R, t, ln, target_from=chained.st)  # a chained step's hit counts are gone: it cannot lend its target's scan


def test_the_timed_object_at_the_timed_size_vs_oracle(L, oracle):
    """what bench.py times -- ops.LossStep(prepared, kept target, want_payload) at BASELINE configs[1]
    (B = 8, N = M = 4096, L = 10000), THIRD call (the chained launch) -- directly against the pinned oracle for two samples:
    T: the step's moved triangles (TRI1) against the oracle's unfused rigid apply (1e-6 of the extent: the kernel's FMAs round
    once where mul + add round twice); W + G on those identical moved triangles: selected-line / D-value / bucket counts exact,
    median equal, loss 1e-5, points1.grad per point 1e-4 (code/loss.py:170-232 + autograd on the same inputs)."""
    from conftest import merge_by_point
    from rrl_hip import ops, synth
    from LieAlgebra import se3
    B, N, M, nl = 8, 4096, 4096, 10000
    prs = [synth.make_pair(1000 + b, N, M) for b in range(B)]  # (bench.py's rank-0 workload)
    src, tar = cu(np.stack([p["src_tri"] for p in prs])), cu(np.stack([p["tar_tri"] for p in prs]))
    ln = _new_lines(L, prs, nl, 3)
    gen = torch.Generator().manual_seed(7)
    R, t = (x.cuda().contiguous() for x in se3.exp3(0.05 * torch.randn(B, 6, generator=gen)))
    step = ops.LossStep(src, tar, nl, want_payload=True)
    for call in range(3):
        step.st.lmax.fill_(-7.0)
        loss, grad, info = step(R, t, ln)
    torch.cuda.synchronize()
    assert bool((step.st.lmax == -7.0).all()), "the third call runs the chained launch"
    for b in (0, B - 1):
        moved_o = oracle.rigid_apply(prs[b]["src_tri"].reshape(-1, 3), R[b].cpu().numpy(), t[b].cpu().numpy(), transpose_r=True).reshape(-1, 9)
        moved = step.st.tri1t[b].cpu().numpy()
        assert np.abs(moved - moved_o).max() <= 1e-6 * np.abs(moved_o).max()
        ref = oracle.loss(moved, prs[b]["tar_tri"], ln[b].cpu().numpy(), want_grad=True)
        assert [int(v) for v in info[b].tolist()] == [ref["n_buckets"], ref["n_selected"], ref["n_values"], int(ref["nan"])]
        assert abs(float(loss[b]) - float(ref["loss"])) <= 1e-5 * abs(float(ref["loss"]))
        assert float(step.st.med[b]) == float(ref["median"])
        a, w = merge_by_point(moved, grad[b].cpu().numpy()), merge_by_point(moved, ref["grad1"])
        assert np.abs(a - w).max() <= 1e-4 * np.abs(w).max()
    valid = info[:, 0] > 0
    assert float(step.payload[1]) == float(valid.sum()) and abs(float(step.payload[0]) - float(loss[valid].double().sum())) < 1e-5


@pytest.mark.parametrize("B,n,m,nl,prepared", [(8, 4096, 4096, 10000, True), (40, 300, 260, 7000, True), (2, 900, 800, 600, True),
                                                (3, 700, 900, 4000, False)])
def test_deterministic_points1_grad(L, B, n, m, nl, prepared):
    """ops.LossStep(deterministic=True) -- the scatter backward in 64-bit fixed point (GFIX) -- reproduces
    points1.grad BIT FOR BIT over 50 calls (the reference's CPU autograd is deterministic) at C2, on the path beyond the tail
    kernel's 256 workgroups (B x tiles = 280), for a single tile of lines and on the cold build; loss bits equal the default
    step's, the gradient agrees with the float-atomic one to 1e-6 of its largest entry; rrl_loss_backward (the autograd
    backward of ops.intersection_loss) under ops.set_deterministic(True) reproduces too, including points2.grad."""
    from rrl_hip import ops
    prs, src, tar = _pairs(980, B, n, m)
    ln = _new_lines(L, prs, nl, 1)
    R, t = _poses(B, 2)
    gl = (0.5 + torch.rand(B, generator=torch.Generator().manual_seed(3))).cuda()
    ref = ops.LossStep(src, tar, nl, prepared=prepared)
    l0, g0, i0 = (x.clone() for x in ref(R, t, ln, grad_loss=gl))
    det = ops.LossStep(src, tar, nl, prepared=prepared, deterministic=True)
    first = None
    for it in range(50):
        l1, g1, i1 = det(R, t, ln, grad_loss=gl)
        if first is None:
            first = g1.clone()
            torch.cuda.synchronize()
            assert torch.equal(l1, l0) and torch.equal(i1, i0)
            # (entries below the fixed-point unit -- exp(-D / 2 med) of a far pair, 1e-14 of the largest possible
            #  contribution -- round to zero: compared by value, not by which rows are non-zero)
            assert float((g1 - g0).abs().max()) <= 2e-6 * float(g0.abs().max())
            assert int((g1.abs().sum(-1) > 0).sum()) >= 0.98 * int((g0.abs().sum(-1) > 0).sum())
        else:
            assert torch.equal(g1, first), it
    # the drop-in autograd route (rrl_loss_backward) with the process-wide switch, both clouds' gradients
    ops.set_deterministic(True)
    try:
        outs = []
        for it in range(6):
            p1 = ops.rigid_apply(src.reshape(B, -1, 3), R, t, transpose_r=True).reshape(B, n, 9).detach().requires_grad_(True)
            p2 = tar.clone().requires_grad_(True)
            loss, info, _ = ops.intersection_loss(p1, p2, ln)
            torch.autograd.backward([loss], [gl])
            outs.append((p1.grad.clone(), p2.grad.clone()))
        for a, b_ in outs[1:]:
            assert torch.equal(a, outs[0][0]) and torch.equal(b_, outs[0][1])
        assert torch.equal(outs[0][0], first)  # the same sums whichever entry issued them
    finally:
        ops.set_deterministic(False)


@pytest.mark.timeout(300)
def test_chained_launch_under_contention(L):
    """The hand-off inside the chained launch (source records -> ready word -> source-cloud scan workgroups) with most of the
    chip taken by another stream's filler kernel (ops.debug_occupy), so that the launch's workgroups become resident a few at a
    time and in waves: the records workgroups lead the grid, the waiters are bounded -- same loss / info / hit lists as the plain
    step every time, no time-out (a time-out would make the sample's loss NaN)."""
    from rrl_hip import ops
    B, n, m, nl = 4, 2048, 2048, 9000
    prs, src, tar = _pairs(990, B, n, m)
    ln = _new_lines(L, prs, nl, 2)
    R, t = _poses(B, 1)
    ref = ops.LossStep(src, tar, nl, chain=False)
    want = _snapshot(ref, ref(R, t, ln))
    st = ops.LossStep(src, tar, nl)
    st(R, t, ln)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    side = torch.cuda.Stream()
    for rnd, free in enumerate((8, 12, 16, 24, 40, 64)):
        # 2 workgroups of 1024 lanes fill a compute unit's wavefront slots: `free` half units stay for the step's launches
        ops.debug_occupy(2 * cus - free, 1024, 0.05, side)
        for _ in range(3):
            st.st.lmax.fill_(-7.0)
            got = _snapshot(st, st(R, t, ln))
            torch.cuda.synchronize()
            _assert_same(want, got, ("contention", free))
            assert bool((st.st.lmax == -7.0).all())
        side.synchronize()


@pytest.mark.timeout(300)
def test_two_chained_steps_on_two_streams_at_once(L):
    """Two step objects (their own workspaces) issued on two streams so that their chained launches overlap on the chip: each
    launch's records workgroups lead its OWN grid, so a launch never waits for the other one's -- every result equals that
    object's plain step, nothing times out, both keep running the ONE-launch build."""
    from rrl_hip import ops
    shapes = [(4, 2048, 2048, 9000), (8, 4096, 4096, 10000)]
    objs = []
    for q, (B, n, m, nl) in enumerate(shapes):
        prs, src, tar = _pairs(1200 + q, B, n, m)
        ln = [_new_lines(L, prs, nl, it) for it in range(2)]
        poses = [_poses(B, it) for it in range(2)]
        ref = ops.LossStep(src, tar, nl, chain=False)
        want = [_snapshot(ref, ref(*poses[it], ln[it])) for it in range(2)]
        objs.append(dict(src=src, tar=tar, nl=nl, ln=ln, poses=poses, want=want, stream=torch.cuda.Stream()))
    torch.cuda.synchronize()
    for o in objs:
        with torch.cuda.stream(o["stream"]):
            o["st"] = ops.LossStep(o["src"], o["tar"], o["nl"])
            o["st"](*o["poses"][0], o["ln"][0])
    torch.cuda.synchronize()
    for rnd in range(40):
        it = rnd & 1
        got = []
        for o in objs:  # (issued back to back: the second object's launches queue while the first one's run)
            with torch.cuda.stream(o["stream"]):
                o["st"].st.lmax.fill_(-7.0)
                got.append(_snapshot(o["st"], o["st"](*o["poses"][it], o["ln"][it])))
        torch.cuda.synchronize()
        for o, g in zip(objs, got):
            _assert_same(o["want"][it], g, ("two streams", rnd))
            assert bool((o["st"].st.lmax == -7.0).all()) and o["st"].fused


@pytest.mark.parametrize("names", [["loss_ref_airplane%d.npz" % i for i in range(5)], ["loss_ref_real%d.npz" % i for i in range(3)],
                                   ["loss_ref_human%d.npz" % i for i in range(3)]])
def test_chained_step_on_the_references_own_pairs(L, oracle, names):
    """The chained launch on the sample pairs the REFERENCE ships (tests/golden/loss_ref_*.npz: airplane, real scan, human; their
    pseudo-triangles and the lines the reference sampled for them), batched: third call of ops.LossStep (R = t = None: the
    triangles as given) against the pinned oracle per sample -- counts exact, median equal, loss 1e-5, per-point gradient 1e-4 --
    and, where the fixture holds the reference's own result for exactly these lines, against that."""
    from conftest import load_golden, merge_by_point
    from rrl_hip import ops
    gs = [load_golden(n) for n in names]
    nl = min(g["lines"].shape[0] for g in gs)
    src, tar = cu(np.stack([g["tri1"] for g in gs])), cu(np.stack([g["tri2"] for g in gs]))
    ln = cu(np.stack([g["lines"][:nl] for g in gs]))
    step = ops.LossStep(src, tar, nl)
    for call in range(3):
        step.st.lmax.fill_(-7.0)
        loss, grad, info = step(None, None, ln)
    torch.cuda.synchronize()
    assert bool((step.st.lmax == -7.0).all()), "the third call runs the chained launch"
    for b, g in enumerate(gs):
        ref = oracle.loss(g["tri1"], g["tri2"], g["lines"][:nl], want_grad=True)
        assert [int(v) for v in info[b].tolist()] == [ref["n_buckets"], ref["n_selected"], ref["n_values"], int(ref["nan"])]
        assert float(step.st.med[b]) == float(ref["median"])
        assert abs(float(loss[b]) - float(ref["loss"])) <= 1e-5 * abs(float(ref["loss"]))
        a, w = merge_by_point(g["tri1"], grad[b].cpu().numpy()), merge_by_point(g["tri1"], ref["grad1"])
        assert np.abs(a - w).max() <= 1e-4 * np.abs(w).max()
        if g["lines"].shape[0] == nl:  # the reference evaluated exactly these lines: its own loss (default bucket range = ranges[0])
            assert [int(v) for v in g["ranges"][0]] == [1, 1, 5, 5]
            assert abs(float(loss[b]) - float(g["r0_loss"])) <= 1e-5 * abs(float(g["r0_loss"]))


_TIMEOUT_CHILD = r"""
import sys
sys.path[:0] = [{root!r}, {pkg!r}, {tests!r}]
import numpy as np, torch
import loss as L
from rrl_hip import ops
from test_gpu_prepared import _pairs
from test_gpu_chain import _new_lines, _poses
B, n, m, nl = 8, 4096, 4096, 10000
prs, src, tar = _pairs(995, B, n, m)
ln = _new_lines(L, prs, nl, 0)
R, t = _poses(B, 0)
ref = ops.LossStep(src, tar, nl, chain=False)
want = ref(R, t, ln)[0].clone()
st = ops.LossStep(src, tar, nl)
st(R, t, ln)
nan = same = other = 0
for it in range(30):
    got = st(R, t, ln)[0].clone()
    torch.cuda.synchronize()
    assert st.fused
    for b in range(B):
        if torch.isnan(got[b]):
            nan += 1
        elif got[b] == want[b]:
            same += 1
        else:
            other += 1
    assert int(st.st.chain.abs().max()) == 0  # the words are cleared on exit, time-out counts included
print("RESULT", nan, same, other, flush=True)
"""


@pytest.mark.timeout(300)
def test_a_wait_that_times_out_is_flagged_never_silent():
    """RRL_CHAIN_SPIN=0 (a child process: the limit is read once): a source-cloud workgroup of the chained launch that finds its
    sample's records not yet published gives up at once instead of waiting.  Every sample's loss must then be EITHER bit-equal to
    the plain step's (all its source workgroups started after the records were out) OR NaN (a time-out was counted in
    CHAIN[b][3]) -- never a finite wrong value --, the launch never hangs, and the CHAIN words are cleared for the next step."""
    import os
    import subprocess
    import sys
    from conftest import PKG, ROOT
    code = _TIMEOUT_CHILD.format(root=ROOT, pkg=PKG, tests=os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=280, env=dict(os.environ, RRL_CHAIN_SPIN="0"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("RESULT")][-1]
    nan, same, other = (int(v) for v in line.split()[1:])
    print(f"[time-out path] of 240 sample evaluations: {nan} NaN (timed out), {same} bit-equal, {other} finite but different")
    assert other == 0 and nan + same == 240
    assert nan > 0, "with a zero limit the first generation's source workgroups must time out"
