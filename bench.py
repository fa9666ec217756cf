#!/usr/bin/env python3
"""
bench.py -- fine-DOF updates/s per V-cycle on the 3D Tet64 checkerboard (BASELINE.json metric).

One "step" = one `vcycle!(implicit, base, ops, levels, L, 3)` (3 CG smoothing steps on the finest level,
2 on the coarser ones as in the reference, coarse solve and all interface sums included) over
BASELINE config 3: 32^3 unit cubes x 6 tets, refinements=5 (L=6, Nf=6545, 1.287e9 fine DOFs), FP64,
synthetic seeded inputs already resident in HBM.  value = Nf_L * Ne / t_vcycle, whole job.

  python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL): the mesh
grows with N (weak scaling, 32^3 cubes per GPU) and is partitioned by coarse-cell ownership.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0    # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)


def cpu_baseline(sample_width, levels, steps_top):
    """The oracle (CPU restatement of the reference algorithm, reference threading structure) timed on
    this host: one V-cycle on a sample_width^3-cube sub-domain of the same workload."""
    import numpy as np
    from oracle import oracle as O
    m = O.order_nodes_and_elements_by_magnitude(
        O.hypercube(3, sample_width, origin=(-sample_width / 2.0,) * 3))
    rng = np.random.default_rng(0)
    sig = rng.choice([1.0, 9.0], size=(m.nelements(), 3))
    impl = O.ImplicitFineGrid.create(m, levels)
    cons = O.ZeroDirichletConstraint(*O.list_boundary_nodes_edges_faces(m))
    ops = [O.L2PlusDivAGrad(O.build_local_diffusion_operators(l), O.mass_matrix(l), cons, 1.0, sig)
           for l in impl.reference.levels]
    st = [O.LevelState.create(m.nelements(), impl.nf(i + 1)) for i in range(levels)]
    st[-1].x[...] = rng.random(st[-1].x.shape)
    O.broadcast_interfaces(st[-1].x, impl, levels)
    O.apply_constraint(st[-1].x, levels, cons, impl)
    dphis = O.partial_derivatives_functionals(impl.reference.levels[-1])
    O.rhs_axi_grad_v(st[-1].b, dphis, impl, sig, np.ones(3) / np.sqrt(3.0))
    base = O.make_base_level(m, sig, 1.0)
    threads = O.available_cores()
    t0 = time.perf_counter()
    O.vcycle(impl, base, ops, st, levels, steps_top)
    dt = time.perf_counter() - t0
    dofs = impl.nf(levels) * m.nelements()
    frac = m.nelements() / float(6 * 32 ** 3)
    return {"value": dofs / dt, "unit": "fine-DOF updates/s per V-cycle", "cores": int(threads), "kind": "port",
            "sample": f"1 V-cycle of the oracle (C restatement, cyclic cell distribution over {threads} OpenMP "
                      f"threads, serial interface sum) on a {sample_width}^3-cube sub-domain "
                      f"({m.nelements()} cells = {frac:.3g} of BASELINE config 3's 196608), L={levels}, {dt:.2f} s",
            "sample_fraction_of_config3": frac,
            # the WHOLE of config 3 through the same oracle needs ~90 GB of host memory and about a minute per V-cycle -- too long
            # for a bench run; tests/test_gpu_fullsize.py::test_vcycle_matches_oracle_full_size does it (and prints the oracle's time
            # with -s): re-measured in round 5, profiles/r05_config3_oracle_fullsize.txt
            "full_size": "profiles/r05_config3_oracle_fullsize.txt"}


def time_to_tolerance(ctx, hmg, driver, n, refinements, tolerance):
    """Wall-clock of the whole driver, `checkerboard_homogenization(n, Tet64, refinements, tolerance)`, on this GPU:
    host setup and the outer loop (V-cycles, integrals, domain shrinks) split out."""
    tm = {}
    t0 = time.perf_counter()
    sigma, hist = driver.checkerboard_homogenization(n, hmg.Tet64, refinements=refinements, tolerance=tolerance, ctx=ctx,
                                                     seed=0, timings=tm)
    return {"call": f"checkerboard_homogenization({n}, Tet64, refinements={refinements}, tolerance={tolerance:g})",
            "seconds": time.perf_counter() - t0, "setup_seconds": tm["setup_s"],
            "setup_split": {"mesh_and_sigma": tm["setup_mesh_s"], "tables_and_upload": tm["setup_tables_s"],
                            "level_vectors": tm["setup_alloc_s"], "x0_and_rhs": tm["setup_init_s"]},
            "solve_seconds": tm["solve_s"],
            "vcycles": tm["vcycles"], "outer_steps": tm["outer_steps"], "sigma": sigma,
            "inexact_vcycles": tm["inexact_vcycles"],      # level-1 solves that ran out of their iteration budget
            "placement": "level vectors as allocated, like the headline ms_per_step: the driver leaves hmg_level_tune_placement off "
                         "(tuning costs more than 18 V-cycles gain)",
            "base_mesh": f"{tm['width']}^3 unit cubes, {tm['cells']} cells",
            "level_vector_memory": "blocks the context kept when the bench's own level vectors were destroyed (option "
                                   "vec_pool; a fresh process allocates them in ~0.2 s, DESIGN.md section 4)"}


def cpu_time_to_tolerance(hmg, driver, n, refinements, tolerance, max_seconds):
    """The same driver through the oracle (CPU restatement) on this host's cores, same coefficient field and initial
    guess; stopped after max_seconds (then the V-cycle count reached is reported and `seconds` is a lower bound)."""
    from oracle import oracle as O
    width = 2 * (driver.compute_box_radius(0, n) + driver.compute_boundary_layer(1.0, n))
    sgrid = driver.generate_conductivity(3, width, 0)
    nf = (2 ** refinements + 1) * (2 ** refinements + 2) * (2 ** refinements + 3) // 6
    x0 = hmg.host_random((nf, 6 * width ** 3), 1)
    t0 = time.perf_counter()
    done = {"cycles": 0}

    class Stop(Exception):
        pass

    def log(h):
        done["cycles"] += 1
        if time.perf_counter() - t0 > max_seconds:
            raise Stop()
    sigma, complete = None, True
    try:
        sigma, hist = O.checkerboard_homogenization(n=n, dim=3, refinements=refinements, tolerance=tolerance,
                                                    sigma_grid=sgrid, x0=x0, log=log)
    except Stop:
        complete = False
    return {"call": f"checkerboard_homogenization({n}, Tet64, refinements={refinements}, tolerance={tolerance:g})",
            "seconds": time.perf_counter() - t0, "vcycles": done["cycles"], "complete": complete, "sigma": sigma,
            "cores": int(O.available_cores()), "kind": "port"}


def launcher_command(ngpus, argv):
    """The torch.distributed.run command line `python bench.py --gpus N ...` turns itself into (one rank per GPU, RCCL).
    --standalone: the launcher picks its own rendezvous port on the loopback interface (no probe-then-reuse race)."""
    return [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
            f"--nproc-per-node={ngpus}", os.path.abspath(__file__)] + list(argv)


def self_launch(ngpus, argv):
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this platform (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(launcher_command(ngpus, argv), env=env)


# ---- N > 1: every rank the launcher starts is a SUPERVISOR ----------------------------------------------------------------
# The N > 1 RCCL path (grouped ncclSend / ncclRecv among the sharers of the cut, overlapped on a second stream) has never run
# on more than one physical GPU, and this bench is the only place it ever will.  So the process torch.distributed.run starts
# does not touch the GPU: it starts the real rank as a CHILD with a time limit and, should that attempt fail or hang on ANY
# rank, a fresh child with HMG_EXCHANGE=allreduce HMG_OVERLAP=0 -- the exchange form that ran on hardware in round 2 -- with a
# fresh rendezvous (a file store per attempt: no key of the dead attempt can collide).  The supervisors of one node agree through
# marker files in a directory all of them derive alike (the launcher's pid + its rendezvous port): `fail_<k>` ends attempt k
# everywhere, `ok_<k>_<rank>` from every rank lets rank 0 forward its child's JSON line, to which it adds which attempt produced
# the number.  Both attempts failing is a non-zero exit and no number.
ATTEMPTS = (("p2p", {}), ("allreduce", {"HMG_EXCHANGE": "allreduce", "HMG_OVERLAP": "0"}))


def supervisor_dir(env):
    import tempfile
    tag = f"{os.getppid()}_{env.get('MASTER_PORT', '0')}_{env.get('TORCHELASTIC_RUN_ID', 'x')}"
    return os.path.join(tempfile.gettempdir(), "hmg_bench_" + "".join(ch if ch.isalnum() or ch == "_" else "_" for ch in tag))


def supervise_rank(argv, child_cmd=None):
    """Runs in every process torch.distributed.run starts for `bench.py --gpus N` (before torch or HIP are imported).
    child_cmd: the command of the real rank (tests substitute a stub); returns the exit code of the supervisor."""
    import signal
    import subprocess
    env0 = dict(os.environ)
    rank, world = int(env0.get("RANK", "0")), int(env0.get("WORLD_SIZE", "1"))
    limit = float(env0.get("HMG_BENCH_ATTEMPT_SECONDS", "600"))
    grace = float(env0.get("HMG_BENCH_BARRIER_SECONDS", "60"))
    d = supervisor_dir(env0)
    os.makedirs(d, exist_ok=True)
    cmd = child_cmd or [sys.executable, os.path.abspath(__file__)] + list(argv)
    failures = []

    def touch(name):
        with open(os.path.join(d, name), "w") as f:
            f.write(str(rank))

    def kill(proc):
        try:
            os.killpg(proc.pid, signal.SIGTERM)        # (the child is the leader of its own session: exactly our process group)
            try:
                proc.wait(timeout=10)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait(timeout=30)
        except (ProcessLookupError, subprocess.TimeoutExpired):
            pass

    for k, (form, extra) in enumerate(ATTEMPTS, start=1):
        env = dict(env0)
        env.update(extra)
        env.update(HMG_BENCH_SUPERVISED="1", HMG_BENCH_ATTEMPT=str(k), HMG_BENCH_RDZV="file://" + os.path.join(d, f"rdzv_{k}"))
        fail, out_path = os.path.join(d, f"fail_{k}"), os.path.join(d, f"stdout_{k}_{rank}")
        t0 = time.time()
        with open(out_path, "w") as out:
            proc = subprocess.Popen(cmd, env=env, stdout=out, start_new_session=True)
        why = None
        while True:
            rc = proc.poll()
            if rc is not None:
                if rc != 0:
                    why = f"rank {rank}: child exit code {rc}"
                break
            if os.path.exists(fail):
                time.sleep(2.0)                       # (a peer failed: this child cannot finish -- give it a moment to say why)
                kill(proc)
                why = f"rank {rank}: stopped, a peer's attempt failed"
                break
            if time.time() - t0 > limit:
                kill(proc)
                why = f"rank {rank}: no result within {limit:.0f} s"
                break
            time.sleep(0.2)
        if why is None:
            touch(f"ok_{k}_{rank}")
            t1 = time.time()                          # all ranks done?  (a rank may still fail in its teardown)
            while why is None and not all(os.path.exists(os.path.join(d, f"ok_{k}_{r}")) for r in range(world)):
                if os.path.exists(fail):
                    why = f"rank {rank}: a peer's attempt failed after this rank had finished"
                elif time.time() - t1 > grace + max(0.0, limit - (t1 - t0)):
                    why = f"rank {rank}: peers did not finish"
                time.sleep(0.1)
        if why is None:
            if rank == 0:
                line = None
                for l in open(out_path).read().splitlines():
                    if l.startswith("{"):
                        line = l
                if line is None:
                    print("bench supervisor: rank 0 finished without a JSON line", file=sys.stderr, flush=True)
                    return 1
                rec = json.loads(line)
                rec["launcher"] = {"supervised": True, "attempt": k, "exchange_form": form, "failed_attempts": failures}
                print(json.dumps(rec), flush=True)
                time.sleep(0.5)
                import shutil
                shutil.rmtree(d, ignore_errors=True)
            return 0
        touch(f"fail_{k}")
        failures.append({"attempt": k, "exchange_form": form, "why": why})
        print(f"bench supervisor: attempt {k} ({form}) failed -- {why}", file=sys.stderr, flush=True)
    return 1


def preflight(ctx, hmg, hdist, dist, world, rank, width=None, levels=5):
    """Before anything is timed on N > 1 ranks: does the partitioned V-cycle compute what the unpartitioned one does?
    A small brick (width^3 unit cubes per rank, level 5 on top) is solved twice by every rank -- partitioned over the N ranks
    with the exchange form of this run, and whole, unpartitioned, on the rank's own GPU -- from the same x0 and b (host twin of
    the device generator), two V-cycles each; the rank compares its columns.  Then the same partitioned V-cycles with the OTHER
    exchange form (all-reduce over the global cut buffer <-> messages among the sharers).  The sharers-only form adds the members'
    partial sums in ascending rank order on every member (k_seg_sum); RCCL's all-reduce adds them in the order of its ring / tree.
    An entity shared by TWO ranks gets the same bits from both (one addition, commutative); with three or more sharers (axis lines,
    the centre node: 4 and 8 ranks) the forms agree to rounding only -- `other_form_bit_identical` is reported, never required
    (profiles/r04_final_bench_lines.txt: false at 4 ranks; tests/test_dist_gloo.py pins the 2-rank equality and the 4-rank
    tolerance).  Returns the record for the JSON line; `ok` = every
    rank within 1e-9 (x) / 1e-8 (r)."""
    import numpy as np
    import torch
    width = width or int(os.environ.get("HMG_PREFLIGHT_WIDTH", "8"))
    form = "allreduce" if os.environ.get("HMG_EXCHANGE", "") == "allreduce" else "p2p"

    def solve(grid, op, base_level, x0, b0, cols):
        st = [hmg.LevelState(grid, i + 1) for i in range(levels)]
        st[-1].x.from_host(x0[:, cols])
        st[-1].b.from_host(b0[:, cols])
        hmg.broadcast_interfaces(st[-1].x, grid, levels)
        hmg.apply_constraint(st[-1].x, levels, grid)
        for _ in range(2):
            hmg.vcycle(grid, base_level, [op] * levels, st, levels, 3)
        out = st[-1].x.to_host(), st[-1].r.to_host()
        for q in st:
            q.close()
        return out

    def partitioned(which):
        keep = {k: os.environ.get(k) for k in ("HMG_EXCHANGE",)}
        try:
            if which == "allreduce":
                os.environ["HMG_EXCHANGE"] = "allreduce"
            else:
                os.environ.pop("HMG_EXCHANGE", None)
            prob = hdist.partitioned_checkerboard(ctx, width, levels, world, rank, seed=7)
        finally:
            for k, v in keep.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        if os.environ.get("HMG_OVERLAP") == "0":
            prob.exchange.set_overlap(prob.implicit, False)
        ctx.set_option("overlap_min_doubles", 1)          # (small brick: force the overlapped form where overlap is on)
        return prob

    try:
        prob = partitioned(form)
        g = prob.implicit
        ne = prob.global_base.elements.shape[0]
        nf = g.nf(levels)
        x0 = hmg.host_random((nf, ne), 11)
        b0 = hmg.host_random((nf, ne), 12) - 0.5
        xp, rp = solve(g, prob.op, prob.base_level(), x0, b0, g.local_cells)
        whole = hmg.ImplicitFineGrid(ctx, prob.global_base, levels)
        op_w = hmg.L2PlusDivAGrad(whole, 1.0, prob.cond)
        xw, rw = solve(whole, op_w, hmg.BaseLevel(whole), x0, b0, slice(None))
        ex = float(np.abs(xp - xw[:, g.local_cells]).max() / max(np.abs(xw).max(), 1e-300))
        er = float(np.abs(rp - rw[:, g.local_cells]).max() / max(np.abs(rw).max(), 1e-300))
        other = "p2p" if form == "allreduce" else "allreduce"
        same = None
        try:
            prob2 = partitioned(other)
            xo, ro = solve(prob2.implicit, prob2.op, prob2.base_level(), x0, b0, prob2.implicit.local_cells)
            same = bool(np.array_equal(xo, xp) and np.array_equal(ro, rp))
        except Exception as e:                             # (the other form is a cross-check, not a requirement)
            same = f"other form failed: {e}"
        worst = torch.tensor([ex, er, 0.0 if same is True else 1.0], dtype=torch.float64)
        if dist is not None and world > 1:
            dev = worst.cuda() if dist.get_backend() == "nccl" else worst
            dist.all_reduce(dev, op=dist.ReduceOp.MAX)
            worst = dev.cpu()
        ex, er = float(worst[0]), float(worst[1])
        return {"ranks": world, "brick": f"{prob.global_shape} unit cubes ({width}^3 per rank), levels={levels}, 2 V-cycles",
                "exchange": form, "rel_err_x": ex, "rel_err_r": er, "tolerance": {"x": 1e-9, "r": 1e-8},
                "other_form_bit_identical": (same if not isinstance(same, bool) else bool(float(worst[2]) == 0.0)),
                "ok": bool(ex <= 1e-9 and er <= 1e-8)}
    finally:
        ctx.set_option("overlap_min_doubles", 524288)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=32, help="unit cubes per axis per GPU")
    ap.add_argument("--levels", type=int, default=6, help="refinements + 1")
    ap.add_argument("--smoothing-steps", type=int, default=3)
    ap.add_argument("--sigma-high", type=float, default=9.0, help="checkerboard contrast: sigma in {1, SIGMA_HIGH}")
    ap.add_argument("--cpu-sample-width", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-time-to-tolerance", action="store_true", help="skip the whole-driver wall-clock runs")
    ap.add_argument("--cpu-driver-seconds", type=float, default=75.0,
                    help="time limit of the CPU oracle's run of the driver on BASELINE config 2")
    ap.add_argument("--apply-threads", type=int, default=None)
    ap.add_argument("--no-level-report", dest="level_report", action="store_false",
                    help="skip the per-level breakdown (roofline.levels), measured after the timed region")
    ap.add_argument("--tuned-burst", type=int, default=5,
                    help="V-cycles timed behind the placement tuner (outside the timed region): config.placement.ms_per_step_tuned")
    ap.add_argument("--tune-placement", type=int, default=8,
                    help="candidates of hmg_level_tune_placement for the finest level's five vectors, tried BEHIND the timed region and "
                         "reported under config.placement (the headline is untuned); 0 = off")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="context option for an A/B run (hmg_ctx_set_option), e.g. --option apply_slab2=0; recorded in config.options")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher.  The ranks are CHILD processes of
        # torch.distributed.run, started before this process has imported torch or touched HIP (no exec of a process
        # that holds the GPU); their stdout (rank 0's JSON line) and stderr pass straight through.
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("HMG_BENCH_SUPERVISED") != "1" and \
            os.environ.get("HMG_BENCH_NO_SUPERVISOR") != "1":
        # a rank started by a launcher (the driver's torch.distributed.run, or ours): supervise the real rank, see above
        raise SystemExit(supervise_rank(sys.argv[1:]))

    # Only the JSON line goes to this process's stdout: libraries underneath write there too (RCCL prints a version
    # banner when a communicator is created, gloo its rank-connection lines), so file descriptor 1 points at stderr until
    # the line is printed.
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import homogenization_jl_amd as hmg
    from homogenization_jl_amd import driver

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if os.environ.get("HMG_SINGLE_DEVICE") == "1":      # rehearsal of the N>1 path on a 1-GPU box (gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # single-GPU rehearsals of the partitioned code path (dist.partitioned_checkerboard): HMG_SYNTHETIC_CUT=planes cuts the
    # one block at its three mid-planes (results bit-identical with the unpartitioned run), HMG_REHEARSE_WORLD=8 holds
    # rank 0's share of the 8-rank brick (timing only); HMG_OVERLAP=0 switches the overlapped exchange off
    synthetic_cut = os.environ.get("HMG_SYNTHETIC_CUT") == "planes"
    rehearse_world = int(os.environ.get("HMG_REHEARSE_WORLD", "0")) or None
    force_part = os.environ.get("HMG_FORCE_PARTITIONED") == "1" or synthetic_cut or rehearse_world is not None
    dist = None
    if world > 1 or force_part:
        import torch.distributed as dist
        if world == 1 and "RANK" not in os.environ:     # single-GPU rehearsal started without a launcher
            import socket
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        backend = os.environ.get("HMG_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        # (a supervised attempt brings its own rendezvous: a fresh file store per attempt, see supervise_rank)
        rdzv = os.environ.get("HMG_BENCH_RDZV")
        kw = {"init_method": rdzv, "rank": rank, "world_size": world} if rdzv else {}
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), **kw)
        else:
            dist.init_process_group(backend, **kw)

    stream = torch.cuda.current_stream().cuda_stream
    ctx = hmg.Context(local_rank, stream=stream)
    if args.apply_threads is not None:
        ctx.set_option("apply_threads", args.apply_threads)
    for o in args.option:
        name, _, value = o.partition("=")
        ctx.set_option(name, int(value))
    # (dev knobs for A/B runs: HMG_OPTIONS=name=value,... is applied by hmg.Context itself)

    L = args.levels
    w = args.width
    pre = None
    if (world > 1 or os.environ.get("HMG_BENCH_PREFLIGHT") == "1") and os.environ.get("HMG_BENCH_PREFLIGHT") != "0" and \
            rehearse_world is None:
        from homogenization_jl_amd import dist as hdist
        pre = preflight(ctx, hmg, hdist, dist, world, rank)
        if not pre["ok"]:
            # wrong numbers from this exchange form: no timing is reported for it.  Under the supervisor the non-zero exit starts
            # the attempt with the other form on every rank.
            if rank == 0:
                print("bench: PREFLIGHT FAILED " + json.dumps(pre), file=sys.stderr, flush=True)
            if dist is not None:
                dist.destroy_process_group()
            raise SystemExit(3)
    t_setup0 = time.perf_counter()
    if world > 1 or force_part:
        from homogenization_jl_amd import dist as hdist
        prob = hdist.partitioned_checkerboard(ctx, w, L, world, rank, seed=0, values=(1.0, args.sigma_high),
                                              synthetic_cut=synthetic_cut, rehearse_world=rehearse_world)
        if os.environ.get("HMG_OVERLAP") == "0":
            prob.exchange.set_overlap(prob.implicit, False)
        base, cond, implicit, op = prob.base, prob.cond, prob.implicit, prob.op
        workload = f"3D Tet64 checkerboard, {prob.global_shape} unit cubes over {world} GPUs ({w}^3 per GPU), refinements={L - 1}"
        if synthetic_cut:
            workload += " [REHEARSAL: partitioned code path on one rank, block cut at its three mid-planes]"
        if rehearse_world:
            workload += f" [REHEARSAL: rank 0's share of the {rehearse_world}-rank partition alone, sums over ranks incomplete: timing only]"
    else:
        base, cond, implicit, op = driver.checkerboard_problem(ctx, hmg.Tet64, w, L, seed=0,
                                                               values=(1.0, args.sigma_high))
        workload = f"3D Tet64 checkerboard, {w}^3 unit cubes x 6 tets, refinements={L - 1}" + \
            (" (BASELINE config 3)" if (w, L, args.sigma_high) == (32, 6, 9.0) else "")
    ne_local = implicit.ncells()
    nf = implicit.nf(L)

    states = [hmg.LevelState(implicit, i + 1) for i in range(L)]
    top = states[-1]
    placement = {"tuned": False}
    base_level = prob.base_level() if (world > 1 or force_part) else hmg.BaseLevel(implicit)
    top.x.rand(1234, cell_offset=rank * ne_local)
    hmg.broadcast_interfaces(top.x, implicit, L)
    hmg.apply_constraint(top.x, L, implicit)
    hmg.rhs_axi_grad_v(top.b, implicit, driver.random_unit_vec(3))
    ops = [op] * L
    ctx.sync()
    setup_seconds = time.perf_counter() - t_setup0     # mesh synthesis, tables, partition, level vectors, x0, b, level-1 matrix

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    for _ in range(args.warmup):
        hmg.vcycle(implicit, base_level, ops, states, L, args.smoothing_steps)
    barrier()
    comm0 = prob.exchange.stats() if (world > 1 or force_part) else (0, 0)
    # The timed region: K V-cycles on the level vectors as the allocator placed them -- what driver.checkerboard_homogenization runs
    # -- with nothing of the measurement inside (round 5: the HIP events of the roofline and the placement tuner moved behind it).
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hmg.vcycle(implicit, base_level, ops, states, L, args.smoothing_steps)
    barrier()
    dt = time.perf_counter() - t0
    comm1 = prob.exchange.stats() if (world > 1 or force_part) else (0, 0)
    if (world > 1 or force_part) and os.environ.get("HMG_EXCHANGE_STATS") == "1":
        ex = prob.exchange
        ncalls, ndoubles = ex.stats()
        print(f"[rank {rank}] exchange ({ex.backend}): {ncalls} collectives, {ndoubles * 8 / 1e6:.1f} MB, "
              f"{ex.seconds:.3f} s host time inside them (warm-up included)", file=sys.stderr, flush=True)
    rnorm = hmg.norm_unique(top.r)            # first copies only; summed over ranks by the library
    # roofline of the dominant kernel: two more V-cycles with HIP events around the finest-level operator applies (events of the
    # library's own stream, include/hmg.h "time_apply")
    ctx.set_option("time_apply", L)
    for _ in range(2):
        hmg.vcycle(implicit, base_level, ops, states, L, args.smoothing_steps)
    launches, ms, nbytes = ctx.apply_timing()
    ctx.set_option("time_apply", 0)

    # Outside the timed region: every level's operator applies between HIP events (two more V-cycles with the timer on for all
    # levels -- inside the timed region only the finest level is timed, an event pair per launch of the launch-bound small levels
    # would move the headline), and each level's whole share of a V-cycle (vector kernels, interface sums, transfers included)
    # from V-cycles started one level lower each time.
    level_rows = []
    if args.level_report:
        ctx.set_option("time_apply", 1)
        nrep = 2
        for _ in range(nrep):
            hmg.vcycle(implicit, base_level, ops, states, L, args.smoothing_steps)
        ctx.sync()
        per_level = {k: ctx.apply_timing_level(k) for k in range(1, L + 1)}
        ctx.set_option("time_apply", 0)
        from_level = {L: 1e3 * dt / args.steps}
        for k in range(L - 1, 0, -1):
            hmg.vcycle(implicit, base_level, ops, states, k, 2)
            ctx.sync()
            t_k = time.perf_counter()
            for _ in range(3):
                hmg.vcycle(implicit, base_level, ops, states, k, 2)
            ctx.sync()
            from_level[k] = 1e3 * (time.perf_counter() - t_k) / 3
        for k in range(L, 0, -1):
            n_k, ms_k, by_k = per_level[k]
            level_rows.append({"level": k, "nf": implicit.nf(k),
                               "apply_launches_per_vcycle": n_k / nrep, "apply_ms_per_vcycle": ms_k / nrep,
                               "apply_algorithmic_GB_per_vcycle": by_k / nrep / 1e9,
                               "apply_TBps": (by_k / 1e12) / (ms_k * 1e-3) if ms_k > 0 else None,
                               "level_share_ms": from_level[k] - from_level.get(k - 1, 0.0)})

    if args.tune_placement > 0 and world == 1 and not force_part:
        # An option of the library, reported beside the headline, never in it: which of the finest level's memory blocks (+ 2 spare
        # ones, freed again) plays x, b, r, p, Ap is chosen by timing that level's share of a V-cycle per candidate (include/hmg.h,
        # hmg_level_tune_placement) -- like an FFT plan.  It costs more than 18 V-cycles gain, so the driver leaves it off.
        ctx.sync()
        t_tune0 = time.perf_counter()
        before, after = hmg.tune_placement(implicit, [op] * L, states, L, args.smoothing_steps, trials=args.tune_placement, extra=2)
        ctx.sync()
        t_tune = time.perf_counter() - t_tune0
        # (the tuner exchanges the memory blocks behind the handles: the vectors hold nothing meaningful afterwards)
        top.x.rand(1234, cell_offset=rank * ne_local)
        hmg.broadcast_interfaces(top.x, implicit, L)
        hmg.apply_constraint(top.x, L, implicit)
        hmg.rhs_axi_grad_v(top.b, implicit, driver.random_unit_vec(3))
        for _ in range(args.warmup):
            hmg.vcycle(implicit, base_level, ops, states, L, args.smoothing_steps)
        ctx.sync()
        t_b = time.perf_counter()
        for _ in range(args.tuned_burst):
            hmg.vcycle(implicit, base_level, ops, states, L, args.smoothing_steps)
        ctx.sync()
        placement = {"tuned": False, "note": "ms_per_step / value are measured on the level vectors as allocated; the tuned figure is an "
                     "option's effect, measured behind the timed region", "candidates": args.tune_placement, "spare_blocks": 2,
                     "finest_level_share_of_a_vcycle_ms_as_allocated": before, "finest_level_share_of_a_vcycle_ms_chosen": after,
                     "tuner_seconds": t_tune, "ms_per_step_tuned": 1e3 * (time.perf_counter() - t_b) / args.tuned_burst,
                     "tuned_burst_steps": args.tuned_burst}

    dt_rank_min = dt_rank_max = dt
    if dist is not None:
        t = torch.tensor([dt, -dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_rank_max, dt_rank_min = float(t[0].item()), -float(t[1].item())
        dt = dt_rank_max
        tot = torch.tensor([float(ne_local)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tot)
        ne_total = int(tot.item())
    else:
        ne_total = ne_local

    if rank == 0:
        ms_step = 1e3 * dt / args.steps
        value = nf * ne_total * args.steps / dt
        avg_ms = ms / max(launches, 1)
        achieved = (nbytes / max(launches, 1)) / (avg_ms * 1e-3) / 1e9 if launches else 0.0
        # HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE doubled
        # as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE); cannot be collected inside this process.
        # ... and only next to the build they were collected on: the record carries the sha256 of the library file and of its
        # sources (homogenization_jl_amd._lib.fingerprint, written by tools/collect_profiles.sh on the GPU box)
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "apply_traffic.json")
        fp = hmg._lib.fingerprint()
        if not (world == 1 and w == 32 and L == 6):
            traffic_src = "not collected for this workload (profiles/apply_traffic.json is BASELINE config 3 on one GPU)"
        elif not os.path.exists(pmc):
            traffic_src = "profiles/apply_traffic.json missing"
        else:
            try:
                rec = json.load(open(pmc))
                if rec.get("lib_sha256") and rec.get("lib_sha256") == fp["lib_sha256"]:
                    traffic, traffic_src = rec.get("hbm_bytes_per_launch"), rec.get("source") + " [same library file as this run]"
                elif rec.get("src_sha256") and rec.get("src_sha256") == fp["src_sha256"]:
                    traffic, traffic_src = rec.get("hbm_bytes_per_launch"), rec.get("source") + " [same kernel sources as this run]"
                else:
                    traffic_src = ("stale: profiles/apply_traffic.json was collected on another build (its lib/src sha256 " +
                                   f"{str(rec.get('lib_sha256'))[:12]}/{str(rec.get('src_sha256'))[:12]}, this run " +
                                   f"{str(fp['lib_sha256'])[:12]}/{str(fp['src_sha256'])[:12]})")
            except Exception as e:
                traffic_src = f"unreadable: {e}"
        out = {
            "metric": "fine-DOF updates/sec per V-cycle (3D Tet64)",
            "value": value,
            "unit": "fine-DOF updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": f"synthetic (seeded checkerboard sigma in {{1,{args.sigma_high:g}}}, hashed x0, b = rhs_a.xi.grad(v), lambda=1)",
            "config": {"workload": workload, "cells": ne_total, "nf": nf, "levels": L,
                       "smoothing_steps": args.smoothing_steps, "smoothing_steps_coarse": 2,
                       "coarse_solver": "device CG, Chebyshev(4)-of-Jacobi preconditioner, rtol 1e-13",
                       "coarse_iterations_last": base_level.last_iterations(),
                       "setup_seconds": setup_seconds, "placement": placement,
                       "residual_norm_after": rnorm, "options": list(args.option),
                       "spare_vector_bytes": ctx.counter("spare_bytes"), "lazy_top_form": ctx.counter("lazy_top_form")},
            "roofline": {"bound": "hbm",
                         "kernel": ("hmg::k_apply<3,512,13,*,6> (finest-level operator apply, three 512-thread workgroups per CU"
                                    if L == 6 else "hmg::k_apply_slab2<*> (finest-level operator apply, rolling LDS windows filled by loader waves, evaluated by evaluator waves"
                                    if L == 7 else "hmg::k_apply (finest-level operator apply") +
                                   "; per V-cycle: 1 residual, 6 fused CG passes, the local residual with the pending x-updates and the restriction in its "
                                   "epilogue, the residual with the coarse-grid correction staged in the LDS image)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "launches": int(launches), "avg_launch_ms": avg_ms,
                         "algorithmic_bytes_per_launch": nbytes / max(launches, 1),
                         "levels": level_rows,
                         "levels_note": "outside the timed region: operator applies of every level between HIP events over 2 more "
                                        "V-cycles (level 5: one wave per cell, hmg::k_apply_wave), and each level's whole share of "
                                        "a V-cycle from V-cycles started one level lower each time (as the TOP level of such a V-cycle a lower level pays a full "
                                        "residual and live smoother tails it does not pay inside a V-cycle started above: an upper bound)"},
            "build": fp,
        }
        out["ms_per_step_ranks"] = {"min": 1e3 * dt_rank_min / args.steps, "max": ms_step}
        if world > 1 or force_part:
            ex = prob.exchange
            out["comm"] = {"backend": ex.backend,
                           "nranks_rccl": ctx.counter("comm_nranks") if ex.backend == "rccl" else None,
                           "nranks_torch": world,
                           "exchange": "p2p" if ex.sharers else "allreduce",
                           "overlap": os.environ.get("HMG_OVERLAP") != "0",
                           "collectives_per_vcycle_rank0": (comm1[0] - comm0[0]) / args.steps,
                           "doubles_per_rank_per_vcycle": (comm1[1] - comm0[1]) / args.steps,
                           "attempt": int(os.environ.get("HMG_BENCH_ATTEMPT", "1"))}
        if pre is not None:
            out["preflight"] = pre
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_width, L, args.smoothing_steps)
        if world == 1 and not args.no_time_to_tolerance and (w, L) == (32, 6):
            # outside the timed region: wall-clock to tolerance = 1e-5 of the whole driver (north_star).  n = 2,
            # refinements = 5 IS BASELINE config 3 (width 2 (2^2 + 4 * 3) = 32); n = 1, refinements = 4 is config 2,
            # the largest the CPU oracle finishes in about a minute.  The bench's own level vectors go back first.
            for st in states:
                st.close()
            ttt = {"config3": time_to_tolerance(ctx, hmg, driver, 2, 5, 1e-5),
                   "config2": time_to_tolerance(ctx, hmg, driver, 1, 4, 1e-5)}
            if not args.no_cpu_baseline:
                ttt["config2_cpu"] = cpu_time_to_tolerance(hmg, driver, 1, 4, 1e-5, args.cpu_driver_seconds)
            out["time_to_tolerance"] = ttt
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
