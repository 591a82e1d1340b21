"""bench.py's own N-rank launch (no GPU needed): `python bench.py --gpus 2 --dry-run`, started WITHOUT torch.distributed.run,
must start two ranks (fresh children through torch.distributed.run), run the path's only exchange (the content min/max
all-reduce) between them over gloo and print ONE JSON line from rank 0 with n_gpus = 2.  --dry-run does no GPU work and
reports value null; everything else is the launcher and rendezvous code the driver's `bench.py --gpus N` goes through."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)


def test_gpus_2_launches_two_ranks_and_reports_them():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--frames", "4"])
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["value"] is None and d["steps"] == 3
    # rank r contributes (min, max) = (1 + r, 4 + r): the reduction went across both ranks
    assert d["content_minmax"] == [1.0, 5.0]
    # what makes a multi-rank line self-proving (the driver's SCALE run carries the same object with backend nccl): the ranks'
    # identities gathered on rank 0, the reduced pair checked against the gathered contributions, the all-reduce timed by itself
    c = d["collective"]
    assert c["world"] == 2 and c["backend"] == "gloo" and len(c["ranks"]) == 2 and [r["rank"] for r in c["ranks"]] == [0, 1]
    assert c["distinct_processes"] == 2 and len({r["pid"] for r in c["ranks"]}) == 2 and len(c["devices"]) == 2
    assert c["reduction_checked"] is True and c["content_minmax"] == c["content_minmax_of_gathered_contributions"] == [1.0, 5.0]
    assert c["allreduce_us"] > 0 and c["allreduce_bytes"] == 8
    assert c["path"] == "torch.distributed" and c["path_note"] is None   # which exchange a reader of the line is looking at
    assert d["config"]["global_frames"] == 8 and "2 GPUs" in d["config"]["workload"]


def test_comm_capi_falls_back_to_torch_with_a_stated_reason_where_it_cannot_run():
    """--comm capi (the exchange issued by libuhdr_hip_comm.so over RCCL) needs one GPU per rank.  Two CPU ranks over gloo cannot
    take it: every rank must agree on that BEFORE anything collective of RCCL's is called (nobody hangs), the run continues on
    torch.distributed, and the line says so -- no re-exec, no error."""
    r = _run(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1", "--frames", "4", "--comm", "capi"])
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    c = d["collective"]
    assert c["path"] == "torch.distributed" and "capi" in c["path_note"] and "gloo" in c["path_note"]
    assert c["reduction_checked"] is True and d["content_minmax"] == [1.0, 5.0]


def test_workload_names_the_baseline_configs():
    sys.path.insert(0, ROOT)
    import bench

    class A:
        frames, apply_format = 64, "hlg"
    assert bench.workload_name(A, 1).startswith("configs[2]: batch 64 x 3840x2160")
    assert bench.workload_name(A, 8).startswith("configs[3]: batch 512 = 64 x 8 frames sharded 8-way")
    assert "weak scaling towards configs[3]" in bench.workload_name(A, 4)


def test_single_rank_dry_run_and_gpus_mismatch():
    r = _run(["--dry-run", "--steps", "1"])
    assert r.returncode == 0 and json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1
    # under torch.distributed.run the flag has to agree with the number of ranks that were started
    r = _run(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 4" in r.stderr
