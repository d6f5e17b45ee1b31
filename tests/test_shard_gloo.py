"""N>1 path on CPU: world_size-2 gloo processes exercise the file->rank partition and the timing reduction that
bench.py uses (one process per GPU, no data-path collective)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, %r)
    import torch.distributed as dist
    from jivetalking_amd import shard
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%s" %% os.environ["JT_PORT"],
                            rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = shard.rank_world()
    durs = [3600, 600, 600, 1200, 300, 2400, 60]
    mine = shard.assign_files(len(durs), world, rank, durs)
    shard.barrier()
    t = shard.max_over_ranks(1.0 + rank)                 # max wall-clock over ranks
    total = shard.sum_over_ranks(sum(durs[i] for i in mine))
    print(json.dumps({"rank": rank, "mine": mine, "tmax": t, "total": total}))
    dist.destroy_process_group()
''') % ROOT


def test_partition_is_disjoint_complete_and_balanced():
    from jivetalking_amd import shard
    durs = [3600, 600, 600, 1200, 300, 2400, 60, 60, 1800]
    for world in (1, 2, 4, 8):
        parts = [shard.assign_files(len(durs), world, r, durs) for r in range(world)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(durs)))
        loads = [sum(durs[i] for i in p) for p in parts]
        assert max(loads) <= max(max(durs), sum(durs) / world * 1.5)
    # one file per GPU for the 8 x 60 min configuration (BASELINE.json configs[2])
    parts = [shard.assign_files(8, 8, r, [3600] * 8) for r in range(8)]
    assert sorted(p[0] for p in parts) == list(range(8)) and all(len(p) == 1 for p in parts)


def test_two_rank_gloo_run(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(29500 + (os.getpid() % 2000))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", JT_PORT=port, MASTER_ADDR="127.0.0.1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(o.strip().splitlines()[-1])
    import json
    res = sorted((json.loads(o) for o in outs), key=lambda d: d["rank"])
    assert sorted(res[0]["mine"] + res[1]["mine"]) == list(range(7))
    assert not set(res[0]["mine"]) & set(res[1]["mine"])
    assert res[0]["tmax"] == res[1]["tmax"] == 2.0            # MAX over ranks
    assert res[0]["total"] == res[1]["total"] == 8760.0       # every second of audio assigned exactly once


def _run_bench(*argv, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=e, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=300)


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment: bench.py starts the two ranks itself (it used to
    run one rank and print n_gpus 1).  --selftest-spawn swaps the GPU step for a sleep and RCCL for gloo, nothing else."""
    import json
    p = _run_bench("--gpus", "2", "--steps", "3", "--warmup", "0", "--selftest-spawn", "--sat-files", "7")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                    # one line, from rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and len(out["per_rank_ms_per_step"]) == 2 and out["spawned_by_bench"] is True
    assert out["per_rank_ms_per_step"][1] > out["per_rank_ms_per_step"][0] * 0.9     # rank order (rank 1 sleeps longer)
    assert out["ms_per_step"] >= max(out["per_rank_ms_per_step"]) - 1e-6             # MAX over ranks
    # BASELINE configs[3]'s leg goes through the same launch: sat-files per GPU x 2 ranks, sharded longest first, every device's
    # count and wall reported, the whole job's wall = MAX over ranks
    sat = out["saturation"]
    assert sat["n_gpus"] == 2 and sat["files"] == 2 * 7 and sat["files_per_device"] == [7, 7]
    for leg in ("md5", "no_md5"):
        r = sat[leg]
        assert r["failed"] == 0 and len(r["per_device_wall_s"]) == 2
        assert r["wall_s"] >= max(r["per_device_wall_s"]) - 2e-3 and r["per_device_wall_s"][1] > r["per_device_wall_s"][0]      # (rank 1's files sleep longer)
        assert abs(r["files_per_s"] - sat["files"] / r["wall_s"]) / r["files_per_s"] < 0.02
        assert r["pipeline_hbm"]["peak_GBps"] == 16000 and 0 < r["pipeline_hbm"]["frac"]
        # all three runs reported, host accounting in the line (VERDICT r4 next #1c)
        assert len(r["wall_s_runs"]) == 3 and r["wall_s"] == min(r["wall_s_runs"]) and r["wall_s_is"] == "best of 3" and r["wall_s_median"] >= r["wall_s"]
        assert r["host"]["host_cores"] >= 1 and r["host"]["cpu_s_per_file"] >= 0 and r["host"]["cores_needed_at_8_gpus"] == pytest.approx(r["host"]["cores_busy"] * 4, abs=0.11)
        assert r["output_lufs_range"] == [-16.01, -16.0]                                                                       # MIN / MAX over ranks


def test_bench_refuses_fewer_devices_than_ranks():
    """No GPU here: `--gpus 2` must fail before spawning anything, not fall back to one rank; a launcher-provided WORLD_SIZE that
    disagrees with --gpus fails too."""
    p = _run_bench("--gpus", "64", "--steps", "1", "--warmup", "0")
    assert p.returncode != 0 and "refusing" in (p.stderr + p.stdout) and not any(ln.startswith("{") for ln in p.stdout.splitlines())
    p = _run_bench("--gpus", "4", "--steps", "1", "--warmup", "0", "--selftest-spawn", env={"WORLD_SIZE": "2", "RANK": "0"})
    assert p.returncode != 0 and "refusing" in (p.stderr + p.stdout)


def test_bench_control_plane_is_gloo_and_share_device_is_a_rehearsal():
    """VERDICT r5 next #6a: the N-rank launch needs no RCCL for what is a barrier and a MAX (default --control-plane gloo), and
    `--share-device` lets N ranks run against device 0 so that the HOST side of the N-rank job can be measured on a one-GPU box.  The
    flags go through the launcher (selftest: the launch path only), and the bench source says what they do."""
    import json
    p = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "0", "--selftest-spawn", "--share-device", "--control-plane", "gloo", "--sat-files", "3")
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["saturation"]["files"] == 6
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'default="gloo"' in src and 'dist.init_process_group("gloo", rank=rank, world_size=world)' in src
    assert "NOT a scaling number" in src                                   # the share-device line says what it is
    p = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--selftest-spawn", "--control-plane", "mpi")
    assert p.returncode != 0
