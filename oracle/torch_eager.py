"""Torch-eager CPU restatement of the reference's FORMULATION of the loss (code/loss.py:68-232):
the dense (L, N, 3, 3) temporaries are materialised, the sparse stage gathers per bucket, the
gradient comes from autograd -- what the reference does on the host, written from SURVEY.md
Appendix A in this repo's own words (no reference source is used or shipped).

TEST INFRASTRUCTURE ONLY (like rrl_oracle.c): imported by tests/ and by bench.py's cpu_baseline
leg, never by the product package.  bench.py times it beside the C port so that the line carries
the CPU cost of the reference's own op sequence (SURVEY.md §8d "CPU baseline beside it");
tests/test_oracle_golden.py pins it to the reference-generated fixtures.

Memory: one scan holds ~6 tensors of L*N*9 floats (C2 sample: N = 4096, L = 10000 -> 1.5 GB each);
`max_lines` evaluates the scan in line chunks of that size (labels/weights are per line, so the
result is unchanged) to bound the peak.
"""
import torch

EPS = 2e-4      # code/loss.py:88
CTHR = 1.731    # code/loss.py:109


def _scan(tri, line):
    """tri (N, 9), line (L, 6) -> weights (L, N, 3) detached, label (L, N) bool.
    code/loss.py:84-112, evaluated with the same materialised temporaries."""
    P = tri.reshape(1, -1, 3, 3)                       # (1, N, k, xyz)
    u = line[:, None, None, :3]
    o = line[:, None, None, 3:]
    AC = P - o                                         # (L, N, 3, 3)
    proj = (AC * u).sum(-1) ** 2                       # (L, N, 3)
    dAC = (AC * AC).sum(-1)
    d = torch.sqrt(dAC - proj + EPS)
    if torch.isnan(d).any():
        raise ValueError("NaN point-to-line distance (code/loss.py:88-91)")
    w = (d / d.sum(-1, keepdim=True)).detach()
    p = tri.reshape(-1, 3, 3)
    e = ((p[:, 1] - p[:, 0]).norm(dim=-1) + (p[:, 2] - p[:, 0]).norm(dim=-1) + (p[:, 1] - p[:, 2]).norm(dim=-1)) / 3
    thr = e * CTHR / 2
    label = (d < thr[None, :, None]).all(-1)
    return w, label


def scan_chunked(tri, line, max_lines):
    ws, ls = [], []
    for s in range(0, line.shape[0], max_lines):
        w, l = _scan(tri, line[s:s + max_lines])
        ws.append(w)
        ls.append(l)
    return torch.cat(ws), torch.cat(ls)


def loss(tri1, tri2, line, rng=(1, 1, 5, 5), max_lines=1024):
    """tri1 (N, 9) (requires_grad for the backward), tri2 (M, 9), line (L, 6), all CPU fp32.
    Returns the (1,)-shaped loss tensor with grad_fn, or None when no bucket is populated."""
    s_m, s_n, e_m, e_n = rng
    w1, lab1 = scan_chunked(tri1.detach(), line, max_lines)
    w2, lab2 = scan_chunked(tri2.detach(), line, max_lines)
    c1, c2 = lab1.sum(1), lab2.sum(1)
    blocks, weights = [], []
    for k in range(s_m, e_m):
        for j in range(s_n, e_n):
            sel = torch.nonzero((c1 == k) & (c2 == j))[:, 0]
            if sel.numel() == 0:
                continue
            # hit triangles in ascending index order, like nonzero() (code/loss.py:125-131)
            f1 = torch.nonzero(lab1[sel])[:, 1].reshape(-1, k)
            f2 = torch.nonzero(lab2[sel])[:, 1].reshape(-1, j)
            q1 = (w1[sel[:, None], f1].unsqueeze(-1) * tri1[f1].reshape(-1, k, 3, 3)).mean(2)   # (S, k, 3)
            q2 = (w2[sel[:, None], f2].unsqueeze(-1) * tri2[f2].reshape(-1, j, 3, 3)).mean(2)   # (S, j, 3)
            D = ((q1[:, :, None, :] - q2[:, None, :, :]) ** 2).sum(-1)                          # (S, k, j)
            blocks.append(D)
            weights.append(float(torch.exp(torch.tensor(-0.5 * abs(k - j)))))
    if not blocks:
        return None
    med = torch.median(torch.cat([b.reshape(-1) for b in blocks])).detach()   # lower median
    total = tri1.new_zeros(1)
    for D, wkj in zip(blocks, weights):
        Wl = 1 - torch.exp(-(D / med) / 2.0)
        total = total + wkj * (Wl.min(2)[0].mean() + Wl.min(1)[0].mean())
    return total / len(blocks)
