"""bench.py --gpus N can never label a run with more GPUs than ranks took part (CPU only, NL_BENCH_DRYRUN stops
each rank after the rank bookkeeping): it starts the N rank processes itself when no launcher did, and every rank
exits non-zero when WORLD_SIZE != N or when fewer than N ranks meet in the rendezvous."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(NL_BENCH_DRYRUN="1", **kw)
    return env


def test_plain_invocation_starts_n_ranks_itself():
    port = 31000 + os.getpid() % 1500
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(MASTER_PORT=str(port)), capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks"] == 2


def test_world_size_mismatch_exits_non_zero():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(WORLD_SIZE="1", RANK="0"), capture_output=True,
                         text=True, timeout=120)
    assert out.returncode != 0 and "refusing" in out.stderr
    assert not out.stdout.strip()          # no JSON line at all


def test_a_rank_that_never_appears_exits_non_zero():
    port = 32600 + os.getpid() % 1500
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"],
                         env=_env(WORLD_SIZE="2", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NL_RDV_TIMEOUT="3"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert not out.stdout.strip()


def test_under_a_launcher_the_ranks_come_from_the_environment():
    port = 34200 + os.getpid() % 1500
    procs = [subprocess.Popen([sys.executable, BENCH, "--gpus", "2"],
                              env=_env(WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                                       MASTER_PORT=str(port)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert json.loads(outs[0][0].strip().splitlines()[-1])["ranks"] == 2
