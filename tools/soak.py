"""Randomised soak: many random (B, N, M, L, scale) shapes through the fused op and the drop-in
loss, culled vs strict scan bit-identical (counts, loss), finite gradients, cached-target path
bit-identical, the step in one C call (tail kernel) on the PREPARED build (k-d order, kept target) against forward +
backward on the cold (sorting) build.  Product-only (no oracle): consistency between independent code paths."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))


def run(seed=0, n_cases=150, log=print):
  """usage (GPU box): python tools/soak.py [seed] [cases]; tests/test_gpu_stress.py runs 200 cases.  Returns the number of
  mismatching cases."""
  from rrl_hip import ops, synth
  import loss as Lm
  rng = np.random.default_rng(seed)
  bad = 0
  t0 = time.time()
  for case in range(n_cases):
      B = int(rng.integers(1, 5))
      N = int(rng.choice([1, 7, 16, 17, 100, 513, 1024, 3000, 4097, 9000]))
      M = int(rng.choice([1, 5, 16, 31, 200, 777, 2048, 5000]))
      Ln = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 500, 1023, 1024, 1025, 2500, 7000, 12000, 20000]))
      scale = float(rng.choice([0.3, 1.0, 1.0, 3.0, 9.0, 30.0, 200.0]))
      prs = [synth.make_pair(int(rng.integers(0, 10**6)), max(N, 8), max(M, 8)) for _ in range(B)]
      src = torch.from_numpy(np.stack([p["src_tri"][:N] for p in prs]) * np.float32(scale)).cuda()
      tar = torch.from_numpy(np.stack([p["tar_tri"][:M] for p in prs]) * np.float32(scale)).cuda()
      pts_s = src.reshape(B, -1, 3); pts_t = tar.reshape(B, -1, 3)
      rad = torch.tensor([[float(p["radius"]) * scale] for p in prs])
      ctr = torch.from_numpy(np.stack([p["center"] for p in prs]) * np.float32(scale))
      torch.manual_seed(case)
      lines = Lm.Random_uniform_distribution_lines_batch_efficient_resample(rad, ctr, Ln, pts_s, pts_t, "cuda")
      if rng.random() < 0.15:  # a non-unit direction somewhere: the NaN/strict fallback path
          lines = lines.clone(); lines[0, 0, :3] *= 1.5
      st_c = ops.loss_forward_raw(src, tar, lines, mode="cull")
      st_s = ops.loss_forward_raw(src, tar, lines, mode="strict")
      torch.cuda.synchronize()
      ok = torch.equal(st_c.count1, st_s.count1) and torch.equal(st_c.count2, st_s.count2) and \
          torch.equal(st_c.loss, st_s.loss) and int(st_c.status[0]) == int(st_s.status[0]) and torch.equal(st_c.info[:, :3], st_s.info[:, :3])
      R = torch.eye(3, device="cuda").repeat(B, 1, 1).requires_grad_(True)
      t = torch.zeros(B, 3, device="cuda").requires_grad_(True)
      loss, info, status = ops.registration_loss(src, R, t, tar, lines, transpose_r=bool(rng.integers(0, 2)))
      loss.sum().backward()
      fin = bool(torch.isfinite(R.grad).all()) and bool(torch.isfinite(t.grad).all()) and bool(torch.isfinite(loss).all())
      same = torch.equal(loss.detach(), st_c.loss) or int(status[0]) != 0
      loss2, _, _ = ops.registration_loss(src, R, t, tar, lines, target_from=st_c)
      cached = torch.equal(loss2.detach(), loss.detach())
      # the step in one C call (the tail kernel where it serves the shape; forced for every second case) against forward +
      # backward: loss / median / info / bucket sums bit for bit, (dR, dt, payload) to the rounding of their atomics
      step = True
      if N >= 1 and M >= 1:
          forced = case % 2 == 1
          res = {}
          try:
              if forced:
                  ops.set_reduce_mode("tiled")
              for one in (False, True):
                  ops.RegistrationStep.ONE_CALL = one
                  # two calls + cold build vs one call + prepared build, the latter CHAINED (round 6): its second and third call run
                  # records + both scans as one launch
                  rs = ops.RegistrationStep(src, tar, Ln, transpose_r=bool(case & 2), want_payload=True, prepared=one, chain=one)
                  for _ in range(3):
                      out = rs(R.detach(), t.detach(), lines)
                  torch.cuda.synchronize()
                  res[one] = [x.clone() for x in (out[0], rs.st.med, out[4], rs.st.bsum, out[1], out[2], out[3])]
          finally:
              ops.RegistrationStep.ONE_CALL = True
              ops.set_reduce_mode("auto")
          nanok = lambda a, b: torch.equal(a, b) or (a.dtype.is_floating_point and torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)))
          # (info[:, 3]: the batch-wide NaN flag after an unchained step, each sample's own after a chained one: compare "any")
          for r_ in res.values():
              r_[2] = torch.cat([r_[2][:, :3], r_[2][:, 3:].max().expand(r_[2].shape[0], 1)], 1)
          step = all(nanok(a, b) for a, b in zip(res[False][:4], res[True][:4]))
          if not step and os.environ.get("RRL_SOAK_VERBOSE"):
              log("  step items (loss, med, info, bsum): " + str([nanok(a, b) for a, b in zip(res[False][:4], res[True][:4])]) +
                  " info " + str(res[False][2].tolist()) + " vs " + str(res[True][2].tolist()))
          for a, b in zip(res[False][4:], res[True][4:]):
              a, b = torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)
              step = step and bool(((a - b).abs() <= 2e-5 * a.abs() + 2e-6 * float(a.abs().max()) + 1e-12).all())
      # ---- round 5 paths
      # the FAT geometry variant of the culled scan (forced) == the lean one (forced): counts, hit lists' sizes, loss, NaN flag
      fat = True
      if max(N, M) > 512:
          res5 = {}
          for flag in ("0", "1"):
              os.environ["RRL_CULL_FAT"] = flag
              try:
                  stf = ops.loss_forward_raw(src, tar, lines, mode="cull")
                  torch.cuda.synchronize()
              finally:
                  del os.environ["RRL_CULL_FAT"]
              res5[flag] = stf
          a_, b_ = res5["0"], res5["1"]
          fat = torch.equal(a_.count1, b_.count1) and torch.equal(a_.count2, b_.count2) and int(a_.status[0]) == int(b_.status[0]) and \
              (torch.equal(a_.loss, b_.loss) or torch.equal(torch.nan_to_num(a_.loss, nan=-7.0), torch.nan_to_num(b_.loss, nan=-7.0))) and \
              torch.equal(a_.count1, st_s.count1)
      # the section-8(d) step with its shard payload == the forward's loss; its gradient finite; k poses as ONE multi-pose
      # evaluation (thin step, poses = k) == the k single-pose steps, per instance (identity poses: every instance equal)
      multi = True
      if N >= 1 and M >= 1 and max(N, M) <= 65536:
          k = int(rng.integers(2, 4))
          ls = ops.LossStep(src, tar, Ln, want_payload=True)
          lo = ls(R.detach(), t.detach(), lines)
          mp = ops.LossStep(src, tar, Ln, poses=k, want_payload=True)
          mo = mp(R.detach().repeat(k, 1, 1), t.detach().repeat(k, 1), lines)
          torch.cuda.synchronize()
          nn = lambda x: torch.nan_to_num(x, nan=-7.0)
          valid = lo[2][:, 0] > 0
          want = float(nn(lo[0])[valid].double().sum())
          multi = torch.equal(nn(lo[0]), nn(st_c.loss)) and torch.equal(nn(mo[0]), nn(lo[0]).repeat(k)) and \
              float(ls.payload[1]) == float(valid.sum()) and (abs(float(nn(ls.payload[:1])[0]) - want) <= 2e-6 * max(1.0, abs(want)) or int(st_c.status[0]) != 0) and \
              float(mp.payload[1]) == k * float(valid.sum()) and bool(torch.isfinite(nn(lo[1])).all())
          ga, gb = nn(lo[1]).repeat(k, 1, 1), nn(mo[1])
          multi = multi and bool(((ga - gb).abs() <= 2e-5 * ga.abs() + 2e-6 * float(ga.abs().max()) + 1e-12).all())
      if not (ok and fin and same and cached and step and fat and multi):
          bad += 1
          log("MISMATCH " + str(dict(case=case, B=B, N=N, M=M, L=Ln, scale=scale, ok=ok, fin=fin, same=same, cached=cached, step=step,
                                      fat=fat, multi=multi)))
  log(f"{n_cases} cases, {bad} mismatches, {time.time() - t0:.1f} s")
  return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 150) else 0)
