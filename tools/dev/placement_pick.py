#!/usr/bin/env python3
"""Second placement experiment (after tools/dev/placement_scan.py: moving a vector INSIDE its allocation changes nothing):
is the fast / slow mode a property of the allocations themselves?  NB separate allocations of one finest-level vector each;
every buffer is timed alone (norm = one read stream, fill = one write stream), then three smoothing steps are timed for
random assignments of five of the buffers to the roles x, b, r, p, Ap, and a least-squares fit splits the time into
per-(buffer) and per-(buffer, role) parts.
  python tools/dev/placement_pick.py [--buffers 10] [--trials 40]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import homogenization_jl_amd as hmg
from homogenization_jl_amd import driver

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=32)
ap.add_argument("--levels", type=int, default=6)
ap.add_argument("--buffers", type=int, default=10)
ap.add_argument("--trials", type=int, default=40)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--order", default="after", choices=["after", "before"], help="buffers allocated after / before the grid")
ap.add_argument("--alloc-gib", type=float, default=0.0, help="size of every allocation in GiB (0: the vector's size + 4 KB) -- does a rounder size change the spread?")
args = ap.parse_args()

ctx = hmg.Context(0)
L = args.levels
NB = args.buffers
NAMES = ["x", "b", "r", "p", "Ap"]
bufs = None
if args.order == "before":
    # (size of a config-3 finest-level vector; checked below)
    nb0 = 8 * 6545 * 6 * args.width ** 3
    bufs = [torch.empty(nb0 + 4096, dtype=torch.uint8, device="cuda:0") for _ in range(NB)]
base, cond, g, op = driver.checkerboard_problem(ctx, hmg.Tet64, args.width, L, seed=0)
nbytes = 8 * g.ld(L) * g.ncells()
if bufs is None:
    asize = int(args.alloc_gib * 2**30) if args.alloc_gib else nbytes + 4096
    assert asize >= nbytes
    bufs = [torch.empty(asize, dtype=torch.uint8, device="cuda:0") for _ in range(NB)]
assert bufs[0].numel() >= nbytes
torch.cuda.synchronize()
ptrs = [b.data_ptr() for b in bufs]
print("buffers:", [hex(p) for p in ptrs], "bytes", nbytes, flush=True)
gaps = [ptrs[i + 1] - ptrs[i] for i in range(NB - 1)]
print("gaps (MiB):", [round(d / 2**20, 2) for d in gaps], flush=True)
xi = driver.random_unit_vec(3)

vecs = [hmg.DeviceMatrix(g, L, device_ptr=p) for p in ptrs]


def timed(fn, reps=3):
    fn()
    ctx.sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ctx.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


print("\nper buffer, alone: fill (write stream) ms, norm (read stream) ms")
for i, v in enumerate(vecs):
    tf = timed(lambda: v.fill(1.0))
    tn = timed(lambda: hmg.norm(v))
    print(f"  buffer {i}: fill {tf:6.3f} ms = {nbytes / tf / 1e9:5.2f} TB/s   norm {tn:6.3f} ms = {nbytes / tn / 1e9:5.2f} TB/s", flush=True)


def measure(assign):
    s = hmg.LevelState.__new__(hmg.LevelState)
    s.level = L
    for n, i in zip(NAMES, assign):
        setattr(s, n, vecs[i])
    for n in ("r", "p", "Ap"):
        getattr(s, n).fill(0.0)
    s.x.rand(1234)
    hmg.broadcast_interfaces(s.x, g, L)
    hmg.apply_constraint(s.x, L, g)
    hmg.rhs_axi_grad_v(s.b, g, xi)
    return timed(lambda: hmg.smoothing_steps(3, g, op, s, L), args.reps)


rng = np.random.default_rng(0)
rows, times = [], []
print("\nthree smoothing steps, random assignments of buffers to roles (x, b, r, p, Ap)")
first = list(range(5))
for t in range(args.trials):
    assign = first if t == 0 else [int(i) for i in rng.permutation(NB)[:5]]
    if t == args.trials - 1:
        assign = first
    ms = measure(assign)
    rows.append(assign)
    times.append(ms)
    print(f"  {assign}  {ms:7.2f} ms", flush=True)
times = np.array(times)
print(f"\nmin {times.min():.2f}  max {times.max():.2f}  mean {times.mean():.2f}  std {times.std():.2f}")
# additive model: time = c + sum over used buffers u_i  (is a BUFFER slow whatever its role?)
A = np.zeros((len(rows), NB))
for k, a in enumerate(rows):
    A[k, a] = 1.0
A1 = np.hstack([A, np.ones((len(rows), 1))])
sol, res, rank, _ = np.linalg.lstsq(A1, times, rcond=None)
fit = A1 @ sol
print("per-buffer effect (ms, relative to their mean):", np.round(sol[:NB] - sol[:NB].mean(), 2))
print(f"residual std of the per-buffer model {np.std(times - fit):.2f} ms (raw std {times.std():.2f})")
best = [int(i) for i in np.argsort(sol[:NB])[:5]]
worst = [int(i) for i in np.argsort(sol[:NB])[-5:]]
print("five best buffers ", best, f"{measure(best):7.2f} ms")
print("five worst buffers", worst, f"{measure(worst):7.2f} ms")
print("five best again   ", best, f"{measure(best):7.2f} ms")
# the roles permuted over one fixed set: does the ROLE of a buffer matter?
print("\nroles permuted over the first five buffers")
for t in range(6):
    a = [int(i) for i in rng.permutation(5)]
    print(f"  {a}  {measure(a):7.2f} ms", flush=True)
