"""bench.py --gpus N without a launcher turns itself into the torch.distributed.run command before torch or HIP are
imported in the parent (the ranks are children; nothing that holds the GPU is ever exec'ed)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
import json, subprocess, sys
sys.argv = ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"]
seen = {}
def fake_call(cmd, env=None):
    seen["cmd"] = cmd
    seen["torch_loaded"] = "torch" in sys.modules
    seen["ipc"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY")
    return 7
subprocess.call = fake_call
import bench
try:
    bench.main()
except SystemExit as e:
    seen["rc"] = e.code
print(json.dumps(seen))
"""


def test_gpus_n_spawns_torchrun_child_before_torch_import():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, "-c", PROBE], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    import json
    seen = json.loads(out.stdout.strip().splitlines()[-1])
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    # the launcher picks its own rendezvous port on the loopback interface (no probe-then-reuse race: ADVICE r3)
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"
    assert "--master-port" not in cmd
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["torch_loaded"] is False          # the parent never touched torch / HIP
    assert seen["ipc"] == "0"
    assert seen["rc"] == 7                        # the child's exit code is the parent's


def test_under_a_launcher_bench_does_not_spawn():
    # WORLD_SIZE set (a rank started by torch.distributed.run): no second launcher; without a GPU the rank stops at
    # "needs an MI355X", not in self_launch
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    code = ("import sys, subprocess\n"
            "def boom(*a, **k): raise AssertionError('spawned')\n"
            "subprocess.call = boom\n"
            "sys.argv = ['bench.py', '--gpus', '2']\n"
            "import bench\n"
            "try:\n    bench.main()\nexcept SystemExit as e:\n    print('EXIT', e.code)\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert "spawned" not in out.stderr
    assert "EXIT" in out.stdout + out.stderr       # (bench points fd 1 at stderr until its JSON line)


# ---- the per-rank supervisor of the N > 1 bench (bench.supervise_rank) ------------------------------------------------------
# Every process the launcher starts supervises the real rank as a child: a time limit, and on failure of ANY rank a fresh
# child with HMG_EXCHANGE=allreduce HMG_OVERLAP=0 on every rank; rank 0 forwards the JSON line and says which attempt made it.
STUB = r"""
import json, os, sys, time
rank, attempt = int(os.environ["RANK"]), int(os.environ["HMG_BENCH_ATTEMPT"])
mode = os.environ["STUB_MODE"]
assert os.environ["HMG_BENCH_SUPERVISED"] == "1" and os.environ["HMG_BENCH_RDZV"].startswith("file://")
if attempt == 2:
    assert os.environ["HMG_EXCHANGE"] == "allreduce" and os.environ["HMG_OVERLAP"] == "0"
else:
    assert "HMG_EXCHANGE" not in os.environ
if mode == "rank1_crashes_first" and attempt == 1:
    if rank == 1:
        sys.exit(5)
    time.sleep(600)                       # rank 0 would wait for its dead peer for ever
if mode == "hang_first" and attempt == 1:
    time.sleep(600)
if mode == "always_fail":
    sys.exit(4)
print("noise before the line")
if rank == 0:
    print(json.dumps({"metric": "stub", "value": 1.0, "attempt_env": attempt}))
"""
DRIVER = r"""
import sys
sys.argv = ["bench.py"]
import bench
sys.exit(bench.supervise_rank([], child_cmd=[sys.executable, "-c", STUB]))
"""


def _supervisors(mode, world=2, limit="20"):
    import tempfile
    port = str(20000 + os.getpid() % 20000)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_PORT=port, STUB_MODE=mode,
                   HMG_BENCH_ATTEMPT_SECONDS=limit, HMG_BENCH_BARRIER_SECONDS="10", TORCHELASTIC_RUN_ID=f"t{mode}{tempfile.mktemp()[-6:]}")
        if r > 0:
            env["TORCHELASTIC_RUN_ID"] = procs[0][1]
        code = "STUB = " + repr(STUB) + "\n" + DRIVER
        procs.append((subprocess.Popen([sys.executable, "-c", code], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                       stderr=subprocess.PIPE, text=True), env["TORCHELASTIC_RUN_ID"]))
    out = []
    for p, _ in procs:
        o, e = p.communicate(timeout=180)
        out.append((p.returncode, o, e))
    return out


def test_supervisor_success_on_the_first_attempt():
    import json
    res = _supervisors("ok")
    assert [r[0] for r in res] == [0, 0], res
    rec = json.loads(res[0][1].strip().splitlines()[-1])
    assert rec["launcher"] == {"supervised": True, "attempt": 1, "exchange_form": "p2p", "failed_attempts": []}
    assert rec["attempt_env"] == 1 and res[1][1].strip() == ""          # only rank 0 prints the line


def test_supervisor_falls_back_when_one_rank_crashes():
    import json
    res = _supervisors("rank1_crashes_first")
    assert [r[0] for r in res] == [0, 0], res
    rec = json.loads(res[0][1].strip().splitlines()[-1])
    assert rec["launcher"]["attempt"] == 2 and rec["launcher"]["exchange_form"] == "allreduce"
    assert rec["launcher"]["failed_attempts"][0]["attempt"] == 1
    assert rec["attempt_env"] == 2


def test_supervisor_falls_back_when_the_first_attempt_hangs():
    import json
    res = _supervisors("hang_first", limit="4")
    assert [r[0] for r in res] == [0, 0], res
    rec = json.loads(res[0][1].strip().splitlines()[-1])
    # (both ranks hang: the supervisor whose clock runs out first writes the marker, the other one may see that marker before its own
    #  limit -- either is the first attempt's failure as rank 0 records it)
    why = rec["launcher"]["failed_attempts"][0]["why"]
    assert rec["launcher"]["attempt"] == 2 and ("no result within" in why or "a peer's attempt failed" in why), why


def test_supervisor_gives_up_after_both_forms():
    res = _supervisors("always_fail")
    assert all(r[0] != 0 for r in res), res
    assert res[0][1].strip() == ""                                        # no number
