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
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
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
