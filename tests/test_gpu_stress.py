"""The in-launch hand-offs under stress (round 4; VERDICT r3 task 6).

* the exchange reduce's bounded spins: a hand-off that times out is REPAIRED by the sample's last workgroup -- driven here
  with a poll limit of 0 (every hand-off "times out") and under real contention (a filler launch on a second stream / in
  a second process that holds most of the device): loss, median, bucket sums bit-identical to the single-workgroup
  reduce, STATUS[2] counts the repaired samples, never a NaN;
* bounded versions of tools/step_stress.py (3000 repeated one-call steps x 3 shapes reproduce themselves bit for bit)
  and tools/soak.py (200 random shapes through independent code paths); the full-length logs are profiles/r04_stress.txt
  and profiles/r04_soak.txt.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from test_gpu_parity import cu

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def L():
    import loss
    from rrl_hip import _lib
    _lib.load()
    assert torch.cuda.is_available()
    return loss


def _batch(L, seed, B, n, m, nl, crowd=0.0):
    from rrl_hip import synth
    prs = [synth.make_pair(seed + b, n, m) for b in range(B)]
    t1, t2 = cu(np.stack([p["src_tri"] for p in prs])), cu(np.stack([p["tar_tri"] for p in prs]))
    ln = []
    for b, p in enumerate(prs):
        torch.manual_seed(seed + b)
        ln.append(L.Random_uniform_distribution_lines_batch_efficient_resample(
            torch.tensor([[float(p["radius"])]]), torch.from_numpy(p["center"]).reshape(1, 3), nl, cu(p["src"])[None],
            cu(p["tar"])[None], "cuda")[0])
    ln = torch.stack(ln)
    if crowd:  # many copies of a few lines: near-identical D values crowd one bin of the median's first radix pass
        k = int(nl * crowd)
        ln[:, :k] = ln[:, :8].repeat(1, (k + 7) // 8, 1)[:, :k]
    return t1, t2, ln


def _forward(t1, t2, ln, reduce_mode):
    from rrl_hip import ops
    st = ops.loss_forward_raw(t1, t2, ln, mode="cull", opts=ops.make_opts(reduce_mode=reduce_mode))
    torch.cuda.synchronize()
    return st


def _same(a, b):
    for x, y in ((a.loss, b.loss), (a.med, b.med), (a.bsum, b.bsum), (a.bcnt, b.bcnt), (a.info, b.info)):
        assert torch.equal(x, y)
    assert bool(torch.isfinite(a.loss).all())


@pytest.mark.parametrize("B,n,m,nl,crowd", [(4, 900, 800, 6000, 0.0), (1, 1024, 1024, 20000, 0.0), (3, 600, 500, 9000, 0.6),
                                             (2, 300, 260, 1000, 0.0), (8, 2048, 2048, 10000, 0.0)])
def test_exchange_reduce_repairs_every_timed_out_handoff(L, B, n, m, nl, crowd):
    """Poll limit 0: every workgroup of the exchange reduce that has to wait gives up at once, adds nothing and leaves its
    sample to the last workgroup's repair (median and Welsch sums recomputed over the whole sample).  Bit-identical to the
    single-workgroup reduce and to the undisturbed exchange reduce -- incl. a crowded bin (the streaming select inside the
    repair) and a single tile of lines (nothing to wait for, nothing repaired)."""
    from rrl_hip import ops
    t1, t2, ln = _batch(L, 700, B, n, m, nl, crowd)
    single = _forward(t1, t2, ln, "single")
    normal = _forward(t1, t2, ln, "xchg")
    _same(single, normal)
    assert int(normal.status[2]) == 0
    try:
        ops.set_spin_limit(0)
        for _ in range(3):
            rep = _forward(t1, t2, ln, "xchg")
            _same(single, rep)
        tiles = (nl + 1023) // 1024
        # with a limit of 0 only the LAST arriver of a hand-off does not time out: every sample of >= 2 tiles is repaired
        assert int(rep.status[2]) == (B if tiles >= 2 else 0)
    finally:
        ops.set_spin_limit(None)
    again = _forward(t1, t2, ln, "xchg")  # the control words were left clean
    _same(single, again)
    assert int(again.status[2]) == 0


def test_exchange_reduce_next_to_a_stream_that_holds_the_device(L):
    """A filler launch on a second stream holds all but a few wavefront slots of the device for 60 ms while the exchange
    reduce runs with a short poll limit: its workgroups become resident a few at a time, hand-offs time out for real,
    and every evaluation still equals the single-workgroup result bit for bit (never a NaN).  The repaired samples are
    counted (STATUS[2]); how many there are depends on the scheduler, the results do not."""
    from rrl_hip import ops
    t1, t2, ln = _batch(L, 720, 6, 1024, 1024, 12000)
    single = _forward(t1, t2, ln, "single")
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    side = torch.cuda.Stream()
    repaired = 0
    try:
        ops.set_spin_limit(1 << 10)  # ~1-2 ms of polling
        for rnd, free in enumerate((10, 12, 14, 16, 20, 24)):
            # 2 workgroups of 1024 lanes fill a compute unit's 32 wavefront slots: leave `free` half units, i.e. room for
            # 4 x free of the reduce's 72 workgroups of 256 lanes (fewer where a whole XCD is full: workgroups are dealt to
            # the XCDs in order) -- the resident ones wait for partners that cannot start
            ops.debug_occupy(2 * cus - free, 1024, 0.08, side)
            st = _forward(t1, t2, ln, "xchg")
            _same(single, st)
            repaired += int(st.status[2])
            side.synchronize()
    finally:
        ops.set_spin_limit(None)
    print(f"[contention, second stream] samples repaired over 6 evaluations of 6 samples: {repaired}")


_CHILD = r"""
import sys, time
sys.path[:0] = [{root!r}, {pkg!r}]
import torch
from rrl_hip import ops
cus = torch.cuda.get_device_properties(0).multi_processor_count
print("ready", flush=True)
t0 = time.time()
while time.time() - t0 < {seconds}:
    ops.debug_occupy(2 * cus - 12, 1024, 0.05)
    torch.cuda.synchronize()
"""


@pytest.mark.timeout(300)
def test_exchange_reduce_next_to_a_process_that_holds_the_device(L):
    """The same from a second PROCESS (a child started here, never an exec): it keeps launching 50 ms fillers over all but
    six compute units while this process evaluates with the exchange reduce.  Same bits as the single-workgroup reduce."""
    from rrl_hip import ops
    t1, t2, ln = _batch(L, 740, 4, 1024, 1024, 8000)
    single = _forward(t1, t2, ln, "single")
    code = _CHILD.format(root=ROOT, pkg=os.path.join(ROOT, "a-robust-registration-loss_amd"), seconds=4.0)
    child = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True)
    repaired = 0
    try:
        assert child.stdout.readline().strip() == "ready"
        ops.set_spin_limit(1 << 10)
        for _ in range(40):
            st = _forward(t1, t2, ln, "xchg")
            _same(single, st)
            repaired += int(st.status[2])
    finally:
        ops.set_spin_limit(None)
        child.wait(timeout=120)
    assert child.returncode == 0
    print(f"[contention, second process] samples repaired over 40 evaluations of 4 samples: {repaired}")


@pytest.mark.timeout(900)
def test_step_stress_bounded(L):
    """tools/step_stress.py at 3000 iterations x 3 shapes (the bench shape among them): RegistrationStep (prepared build,
    kept target, tail kernel) and LossStep (scatter in the tail kernel) reproduce their first call bit for bit."""
    import step_stress
    lines = []
    bad = step_stress.run(3000, ((8, 4096, 10000), (3, 1500, 16000), (1, 1024, 3000)), log=lines.append)
    print("\n".join(lines))
    assert bad == 0, lines


@pytest.mark.timeout(900)
def test_soak_200_shapes(L):
    """tools/soak.py: 200 random (B, N, M, L, scale) shapes -- culled vs strict scan, cached target, one-call prepared step
    vs two-call cold step."""
    import soak
    lines = []
    bad = soak.run(4, 200, log=lines.append)
    print("\n".join(lines))
    assert bad == 0, lines
