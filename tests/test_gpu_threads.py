"""Per-call options instead of library globals (round 4; VERDICT r3 task 7, SURVEY 8(b): "re-entrant and thread-safe per
stream").  Two host threads drive two streams with DIFFERENT options at the same time -- reduce kernel, deterministic
backward, sort parts, prepared vs cold build, per-call counter tables -- and each must reproduce the bits it produced
alone.  (ctypes releases the GIL inside every C call, so the two threads really are inside the library concurrently.)"""
import threading

import numpy as np
import pytest
import torch

from test_gpu_parity import cu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import loss
    from rrl_hip import _lib
    _lib.load()
    assert torch.cuda.is_available()
    return loss


def _workload(L, seed, B, n, m, nl):
    from rrl_hip import synth
    from LieAlgebra import se3
    prs = [synth.make_pair(seed + b, n, m) for b in range(B)]
    src, tar = cu(np.stack([p["src_tri"] for p in prs])), cu(np.stack([p["tar_tri"] for p in prs]))
    ln = []
    for b, p in enumerate(prs):
        torch.manual_seed(seed + b)
        ln.append(L.Random_uniform_distribution_lines_batch_efficient_resample(
            torch.tensor([[float(p["radius"])]]), torch.from_numpy(p["center"]).reshape(1, 3), nl, cu(p["src"])[None],
            cu(p["tar"])[None], "cuda")[0])
    gen = torch.Generator().manual_seed(seed)
    R, t = (x.cuda().contiguous() for x in se3.exp3(0.03 * torch.randn(B, 6, generator=gen)))
    return src, tar, torch.stack(ln), R, t


@pytest.mark.timeout(600)
def test_two_threads_two_streams_different_options(L):
    from rrl_hip import ops
    iters = 300
    wa = _workload(L, 800, 4, 1500, 1300, 9000)
    wb = _workload(L, 820, 3, 1100, 1400, 7000)
    errors, done = [], {}

    def thread_a(stream, check_against):
        # cold build on 2 sort parts, the single-workgroup reduce, the deterministic (bit-reproducible) direct backward
        with torch.cuda.stream(stream):
            src, tar, ln, R, t = wa
            step = ops.RegistrationStep(src, tar, ln.shape[1], want_payload=True, prepared=False, reduce_mode="single",
                                        deterministic=True, sort_parts=2)
            res = None
            for it in range(iters if check_against is not None else 3):
                out = step(R, t, ln)
                cur = [x.clone() for x in (out[0], step.st.med, out[4], step.st.bsum, out[1], out[2], out[3])]
                stream.synchronize()
                res = cur
                if check_against is not None and not all(torch.equal(a, b) for a, b in zip(cur, check_against)):
                    errors.append(("A", it))
                    break
            done["A"] = res

    def thread_b(stream, check_against):
        # prepared build (k-d order, kept target), the tail kernel, then a forward with the exchange reduce and a per-call
        # counter table of the scan
        with torch.cuda.stream(stream):
            src, tar, ln, R, t = wb
            step = ops.RegistrationStep(src, tar, ln.shape[1], want_payload=True, reduce_mode="tiled")
            o1, o2 = ops.cloud_order(src), ops.cloud_order(tar)
            tri = ops.rigid_apply(src.reshape(src.shape[0], -1, 3), R, t, transpose_r=True).reshape(src.shape)
            res = None
            for it in range(iters if check_against is not None else 3):
                out = step(R, t, ln)
                table = torch.zeros(1 << 14, 16, dtype=torch.int64, device="cuda")
                st = ops.loss_forward_raw(tri, tar, ln, opts=ops.make_opts(order1=o1, order2=o2, reduce_mode="xchg", counters=table))
                cur = [x.clone() for x in (out[0], step.st.med, out[4], step.st.bsum, st.loss, st.med, st.bsum, table.sum(0)[:8])]
                grads = [out[1].clone(), out[2].clone()]
                stream.synchronize()
                res = cur + grads
                if check_against is not None:
                    same = all(torch.equal(a, b) for a, b in zip(cur, check_against[:8]))
                    close = all(bool(((a - b).abs() <= 2e-5 * b.abs() + 2e-6 * float(b.abs().max())).all()) for a, b in zip(grads, check_against[8:]))
                    if not (same and close):
                        errors.append(("B", it))
                        break
            done["B"] = res

    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    thread_a(sa, None)   # each thread's own single-thread bits first
    thread_b(sb, None)
    ref_a, ref_b = done["A"], done["B"]
    assert torch.equal(ref_b[0], ref_b[4]) and torch.equal(ref_b[3], ref_b[6])  # the two routes of thread B agree with each other
    assert int(ref_b[7][5]) > 0 and int(ref_a[2][:, 1].min()) > 0             # counters were written; lines were selected
    ta = threading.Thread(target=thread_a, args=(sa, ref_a))
    tb = threading.Thread(target=thread_b, args=(sb, ref_b))
    ta.start(); tb.start()
    ta.join(); tb.join()
    torch.cuda.synchronize()
    assert not errors, errors
    # the process-wide defaults were never touched: a plain call still takes them
    assert ops.loss_forward_raw(*wa[:3]).loss.shape == (4,)


@pytest.mark.timeout(600)
def test_cloud_order_from_two_threads_on_two_streams(L):
    """ops.cloud_order of the SAME shape from two threads on two streams at once (a data-preparation side stream next to a
    step's construction): every returned order is a permutation and equals the single-threaded one (the scratch
    used to be one global buffer per shape)."""
    from rrl_hip import ops, synth
    B, n = 4, 3000
    clouds = [cu(np.stack([synth.make_pair(900 + 10 * k + b, n, 64)["src_tri"] for b in range(B)])) for k in range(2)]
    want = [ops.cloud_order(c).clone() for c in clouds]
    torch.cuda.synchronize()
    ar = torch.arange(n, device="cuda")
    for w_ in want:
        assert bool((torch.sort(w_[:, :n].long(), dim=1).values == ar).all())
    errors = []

    def worker(k, stream):
        with torch.cuda.stream(stream):
            for it in range(200):
                o = ops.cloud_order(clouds[k])
                if it % 20 == 19:
                    stream.synchronize()
                    if not torch.equal(o, want[k]):
                        errors.append((k, it))
                        return
            stream.synchronize()
            if not torch.equal(o, want[k]):
                errors.append((k, "last"))

    ths = [threading.Thread(target=worker, args=(k, torch.cuda.Stream())) for k in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    torch.cuda.synchronize()
    assert not errors, errors
