"""Dev probe: per-wave time line of the fused rollout (two-lane kernel, probe build with HK_DUO_PROBE): when each wave
starts, has its slab, leaves the step loop and ends; how many steps it ran."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hironaka_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_probe", "libhk_probe.so")
import numpy as np
import torch
from hironaka_amd import ops

b, m, d, T = 65536, 20, 3, 20
P = ops.generate_points(b, m, d, 20, seed=42)
Q = torch.empty_like(P)
for rep in range(3):
    res = ops.rollout(Q, T, 1 + rep, initial=P, record=("game_length",))
torch.cuda.synchronize()
full = res["game_length"].cpu().numpy().reshape(-1, 32).astype(np.int64)
gl = full[:, :8]
t0, t1, t2, t3, steps, smax, blk, hwid = (gl[:, i] for i in range(8))
base = t0.min()
us = lambda x: ((x - base) & 0xFFFFFFFF) / 100.0  # 100 MHz constant clock
print(f"waves {len(t0)}; kernel span (first start .. last end): {us(t3).max():.2f} us")
for name, v in (("start", us(t0)), ("slab in registers", us(t1)), ("loop done", us(t2)), ("end", us(t3))):
    print(f"{name:18s} min {v.min():6.2f}  p10 {np.percentile(v,10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v,90):6.2f}  max {v.max():6.2f}")
life = us(t3) - us(t0)
print(f"lifetime           mean {life.mean():6.2f}  median {np.median(life):6.2f}  p90 {np.percentile(life,90):6.2f}  max {life.max():6.2f}")
loop = us(t2) - us(t1)
print("steps run: " + "  ".join(f"{k}:{(steps==k).sum()}" for k in sorted(set(steps.tolist()))))
for k in sorted(set(steps.tolist())):
    sel = steps == k
    print(f"  steps {k:2d}: waves {sel.sum():5d}  loop time mean {loop[sel].mean():6.2f} us  lifetime mean {life[sel].mean():6.2f}  end max {us(t3)[sel].max():6.2f}")
print("initial slots per lane: " + "  ".join(f"{k}:{(smax==k).sum()}" for k in sorted(set(smax.tolist()))))
late = np.argsort(us(t3))[-10:]
print("the ten last waves: block, start, slab, loop done, end, steps, smax")
for i in late:
    print(f"  {blk[i]:5d} {us(t0)[i]:6.2f} {us(t1)[i]:6.2f} {us(t2)[i]:6.2f} {us(t3)[i]:6.2f} {steps[i]:3d} {smax[i]:2d}")

# prologue stamps (probe build): action window filled / slab in LDS / (scan: t1) / rows in registers
pa, pb, pg = (full[:, 8 + k] & 0xFFFFFFFF for k in (21, 22, 23))
usp = lambda x: ((x - (base & 0xFFFFFFFF)) & 0xFFFFFFFF) / 100.0
print("prologue (since the wave's start): window filled %.2f, slab in LDS %.2f, scanned %.2f, rows in registers %.2f; "
      "epilogue (loop done -> end) %.2f" % ((usp(pa) - us(t0)).mean(), (usp(pb) - us(t0)).mean(), (us(t1) - us(t0)).mean(),
                                              (usp(pg) - us(t0)).mean(), (us(t3) - us(t2)).mean()))
ps = full[:, 8 + 20] & 0xFFFFFFFF
d_stages = ((ps - pg) & 0xFFFFFFFF) / 100.0
d_rest = ((((full[:, 8] & 0xFFFFFFFF) >> 4) - (ps & 0x0FFFFFFF)) & 0x0FFFFFFF) / 100.0
print("step 0: rows in registers -> stages done %.2f us, stages done -> end of the step (counts, re-deal) %.2f us" % (
    d_stages.mean(), d_rest.mean()))
# per-step stamps (probe build): time of each step and the slots per lane after it
st = full[:, 8:28] & 0xFFFFFFFF  # (entries 21 - 23 of the buffer hold the prologue stamps)
clk = (st >> 4); sm = st & 15
t1m = (t1 & 0x0FFFFFFF)
print("step: mean duration [us] over the waves that ran it / mean slots per lane after it / share of waves whose slots shrank (re-deal)")
prev = t1m
prev_s = smax
for k in range(20):
    ran = steps > k
    if ran.sum() == 0:
        break
    dt = ((clk[:, k] - prev) & 0x0FFFFFFF) / 100.0
    shr = (sm[:, k] < prev_s) & ran
    print(f"  step {k:2d}: waves {ran.sum():5d}  mean {dt[ran].mean():6.3f}  with re-deal {dt[shr].mean() if shr.sum() else 0:6.3f} (n={shr.sum()})  without {dt[ran & ~shr].mean() if (ran & ~shr).sum() else 0:6.3f}  slots after {sm[ran, k].mean():.2f}")
    prev = np.where(ran, clk[:, k], prev)
    prev_s = np.where(ran, sm[:, k], prev_s)
