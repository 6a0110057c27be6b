"""World-size-2 gloo test of the batch-shard path on CPU: shard bounds, the fused all-reduce of
loss / valid count / shared gradients, and that the sharded sum equals the unsharded one.
The per-sample loss is plugged with the CPU oracle (tests may use it; the product never does)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_loss_fn(p1, p2, ln, rng):
    from oracle import rrl_oracle
    vals, ok = [], []
    for b in range(p1.shape[0]):
        r = rrl_oracle.loss(p1[b].numpy(), p2[b].numpy(), ln[b].numpy(), rng=tuple(rng), want_grad=False)
        vals.append(0.0 if r["loss"] is None else float(r["loss"]))
        ok.append(r["loss"] is not None)
    return torch.tensor(vals, dtype=torch.float32), torch.tensor(ok)


def _worker(rank, world, port, out):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from rrl_hip import dist as rdist
    r, w, _ = rdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = load_golden("loss_b2_quirk.npz")
    far = load_golden("loss_edge_allmiss.npz")
    # global batch of 3: two real samples and one whose lines miss everything (invalid)
    n = g["tri1"].shape[1]
    p1 = torch.from_numpy(np.stack([g["tri1"][0], g["tri1"][1], g["tri1"][0]]))
    p2 = torch.from_numpy(np.stack([g["tri2"][0], g["tri2"][1], g["tri2"][0]]))
    ln = torch.from_numpy(np.stack([g["lines"][0], g["lines"][1],
                                    np.resize(far["lines"], g["lines"][0].shape)]))
    total, nvalid = rdist.sharded_batch_loss(p1, p2, ln, loss_fn=_oracle_loss_fn)
    # shared-parameter gradients are summed in the same collective
    shared = torch.full((6,), float(rank + 1))
    lo, hi = rdist.shard_bounds(3, rank, world)
    l, v = _oracle_loss_fn(p1[lo:hi], p2[lo:hi], ln[lo:hi], (1, 1, 5, 5))
    t2, n2 = rdist.reduce_loss(l, v, (shared,))
    payload = rdist.reduce_payload(torch.arange(14.0) * (rank + 1))
    # the overlapped reducer used by bench.py: every step's sum arrives, one submit late
    red = rdist.PayloadReducer(torch.device("cpu"))
    step_buf = torch.zeros(14)             # stands in for the captured step's static output
    seen = []
    for step in range(3):
        step_buf.copy_(torch.arange(14.0) * (rank + 1) * (step + 1))
        red.submit(step_buf)
        seen.append(red.finish().clone())  # consumed before the buffer is overwritten again
    if rank == 0:
        torch.save(dict(total=total, nvalid=nvalid, t2=t2, n2=n2, shared=shared, payload=payload, n=n,
                        seen=torch.stack(seen)), out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds():
    from rrl_hip.dist import shard_bounds
    for total in (0, 1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            parts = [shard_bounds(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_world2_gloo(tmp_path, oracle):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    g = load_golden("loss_b2_quirk.npz")
    want = sum(float(oracle.loss(g["tri1"][b], g["tri2"][b], g["lines"][b])["loss"]) for b in (0, 1))
    assert float(res["nvalid"]) == 2.0 and float(res["n2"]) == 2.0
    np.testing.assert_allclose(float(res["total"]), want, rtol=1e-6)
    np.testing.assert_allclose(float(res["t2"]), want, rtol=1e-6)
    np.testing.assert_allclose(res["shared"].numpy(), np.full(6, 3.0))
    np.testing.assert_allclose(res["payload"].numpy(), np.arange(14.0) * 3)
    for step in range(3):  # ranks contribute (rank + 1) * (step + 1) * arange
        np.testing.assert_allclose(res["seen"][step].numpy(), np.arange(14.0) * 3 * (step + 1))


def _worker8(rank, world, port, out):
    """BASELINE configs[2] in miniature: a global batch of 64 samples sharded over 8 ranks (8 each), per-sample
    "losses" that identify the sample (so a wrong or overlapping shard shows in the sum), and the overlapped
    payload reducer over several steps."""
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from rrl_hip import dist as rdist
    r, w, _ = rdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = rdist.shard_bounds(64, rank, world)
    assert (lo, hi) == (8 * rank, 8 * rank + 8)
    tag = torch.arange(64, dtype=torch.float32).reshape(64, 1, 1).expand(64, 2, 9).contiguous()  # sample b carries b

    def fn(p1, p2, ln, rng):  # loss[b] = 2^-b-ish marker per sample; sample 13 and 40 are "invalid"
        idx = p1[:, 0, 0]
        return (idx + 1.0), ~((idx == 13) | (idx == 40))
    total, nvalid = rdist.sharded_batch_loss(tag, tag, tag[:, :, :6].contiguous(), loss_fn=fn)
    red = rdist.PayloadReducer(torch.device("cpu"))
    buf = torch.zeros(14)
    seen = []
    for step in range(4):
        buf.copy_(torch.arange(14.0) + 100.0 * rank + step)
        red.submit(buf)
        seen.append(red.finish().clone())
    shared = torch.full((6,), float(rank))
    rdist.reduce_loss(torch.zeros(1), torch.ones(1, dtype=torch.bool), (shared,))
    if rank == 0:
        torch.save(dict(total=total, nvalid=nvalid, seen=torch.stack(seen), shared=shared), out)
    dist.barrier()
    dist.destroy_process_group()


def test_world8_gloo_global_batch_64(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker8, args=(8, _free_port(), out), nprocs=8, join=True)
    res = torch.load(out)
    want = sum(b + 1.0 for b in range(64) if b not in (13, 40))
    assert float(res["nvalid"]) == 62.0 and float(res["total"]) == want
    for step in range(4):  # sum over ranks of (arange + 100 rank + step)
        np.testing.assert_allclose(res["seen"][step].numpy(), 8 * (np.arange(14.0) + step) + 100.0 * 28)
    np.testing.assert_allclose(res["shared"].numpy(), np.full(6, 28.0))


def _worker_rows(rank, world, port, out):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from rrl_hip import dist as rdist
    rdist.init_from_env(backend="gloo")
    # rank r brings (3 r) % 5 rows (rank 0 and the ranks 5 k none at all) whose entries identify (rank, row, column)
    n = (3 * rank) % 5
    rows = (1000.0 * rank + 16.0 * torch.arange(n, dtype=torch.float32)[:, None] + torch.arange(16, dtype=torch.float32)[None, :])
    kj = ((rank + torch.arange(n)) % 4 + 1 + 16 * ((rank + 2 * torch.arange(n)) % 4 + 1)).to(torch.uint8)
    allr, allk = rdist.gather_rows(rows, kj)
    # round 4: a per-rank flag (the scan's NaN flag) rides in the sizes' all-gather: the maximum comes back on every rank
    r2, k2, fl = rdist.gather_rows(rows, kj, flag=torch.tensor([1 if rank == 2 else 0], dtype=torch.int32))
    r3, k3, f0 = rdist.gather_rows(rows, kj, flag=torch.tensor([0], dtype=torch.int32))
    assert torch.equal(r2, allr) and torch.equal(k2, allk) and int(fl) == 1 and fl.dtype == torch.int32 and int(f0) == 0
    lo, hi = rdist.shard_bounds(7001, rank, world)
    if rank == world - 1:
        torch.save(dict(rows=allr, kj=allk, last=(lo, hi)), out)
    dist.barrier()
    dist.destroy_process_group()


def test_line_shard_gather_rows_gloo(tmp_path):
    """The all-gather of the line-sharded single-sample mode (rrl_hip.dist.gather_rows): ragged shares incl. empty ones
    arrive concatenated in rank order, k | j << 4 bytes intact, on every rank (checked on the last)."""
    world = 4
    out = str(tmp_path / "rows.pt")
    mp.spawn(_worker_rows, args=(world, _free_port(), out), nprocs=world, join=True)
    res = torch.load(out)
    want_rows, want_kj = [], []
    for rank in range(world):
        n = (3 * rank) % 5
        want_rows.append(1000.0 * rank + 16.0 * torch.arange(n, dtype=torch.float32)[:, None] + torch.arange(16, dtype=torch.float32)[None, :])
        want_kj.append(((rank + torch.arange(n)) % 4 + 1 + 16 * ((rank + 2 * torch.arange(n)) % 4 + 1)).to(torch.uint8))
    assert torch.equal(res["rows"], torch.cat(want_rows)) and torch.equal(res["kj"], torch.cat(want_kj))
    assert res["rows"].shape[0] == 0 + 3 + 1 + 4 and res["last"][1] == 7001
