"""bench.py's N>1 path (proof sharding, barrier-bracketed timing, max-over-ranks aggregation) on
CPU with the gloo backend, world_size 2.  The step function is a stand-in sleep: no GPU here."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.01 * (rank + 1))   # rank 1 is the slow one

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed = bench.timed_region(step, 5, 2, dist.barrier, lambda: None)
    agg = bench.aggregate(elapsed, world, 5, reduce_max)
    q.put((rank, len(calls), elapsed, agg))
    dist.destroy_process_group()


def test_two_rank_sharded_bench_logic():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, e0, a0), (r1, n1, e1, a1) = res
    assert n0 == n1 == 7                        # 2 warmup + exactly 5 timed steps on each rank
    assert abs(a0["seconds"] - a1["seconds"]) < 1e-9   # both ranks agree on the max-over-ranks time
    assert a0["seconds"] >= 5 * 0.02 * 0.9      # bounded below by the slow rank
    assert abs(a0["value"] - world * 5 / a0["seconds"]) < 1e-9   # whole-job proofs/s over all ranks


def _run_bench(args, timeout=600):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, cwd=root, env=env, capture_output=True,
                       text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_2_starts_its_own_two_ranks():
    """`python bench.py --gpus 2` as the driver runs it (no torchrun environment) must START two ranks, not report
    N = 1: end to end over gloo with the test-only emulation build (rank launch, barrier-bracketed region,
    max-over-ranks aggregation, the `sharded_proof` sub-result over both ranks, exit status)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    emu = os.path.join(root, "tests", "emu", "libluminair_emu.so")
    if not os.path.exists(emu):
        import subprocess
        subprocess.run(["bash", os.path.join(root, "tests", "emu", "build_emu.sh")], check=True)
    r, line = _run_bench(["--gpus", "2", "--emu-library", emu, "--log-rows", "5", "--steps", "4", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 1
    assert line["config"]["ranks_in_process_group"] == 2 and line["config"]["collective_backend"] == "gloo"
    assert line["errors"] == []
    sp = line["sharded_proof"]["config_2a"]
    assert sp["bytes_identical_to_unsharded_proof"] is True
    assert sp["solo_unsharded_latency_ms"] > 0 and isinstance(sp["sharding_wins"], bool)   # says when sharding loses
    assert line["short_region"]["steps"] == 20 and line["config"]["proofs_per_step_per_gpu"] == 1
    assert "EMULATION" in line["data"]              # never mistaken for a measurement
    assert line["value"] > 0 and abs(line["value"] - 2 * 4 / (line["ms_per_step"] * 4 / 1e3)) < 1e-6 * line["value"]


def test_bench_headline_survives_a_sharded_sub_result_that_never_finishes(monkeypatch):
    """The `sharded_proof` sub-results run in a child process per rank (RCCL with more than one rank first runs on the
    driver's node: a hang or a crash there must not cost the N-GPU throughput line).  A child that does not finish within
    LMN_BENCH_SHARDED_TIMEOUT is killed: the ONE JSON line still appears, carries the headline, says what happened in
    `warnings`, and the exit status stays 0 (an infrastructure problem, not a rejected proof)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    emu = os.path.join(root, "tests", "emu", "libluminair_emu.so")
    monkeypatch.setenv("LMN_BENCH_SHARDED_TIMEOUT", "0.2")        # no python child gets as far as `import torch` in 0.2 s
    r, line = _run_bench(["--gpus", "2", "--emu-library", emu, "--log-rows", "5", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len([ln for ln in r.stdout.splitlines() if ln.startswith("{")]) == 1
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["errors"] == []
    assert "error" in line["sharded_proof"]["config_2a"] and "timed out" in line["warnings"][0]


def test_bench_refuses_more_gpus_than_the_node_has():
    """On a box with fewer than N GPUs `--gpus N` fails loudly instead of benchmarking one GPU as N = 1."""
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    r, line = _run_bench(["--gpus", str(max(2, 2 * n)), "--steps", "2", "--warmup", "0"], timeout=300)
    assert r.returncode != 0 and line is None
    assert "GPU(s) visible" in r.stderr


def test_throughput_never_puts_two_threads_into_one_context():
    """bench.throughput(): one driver thread per context (round 2's shared worker pool entered a context twice)."""
    import threading
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    class Ctx:
        def __init__(self):
            self.inside, self.calls, self.overlap, self.lock = 0, 0, 0, threading.Lock()

        def prove_tables(self, bufs, luts=None):
            with self.lock:
                self.inside += 1
                self.overlap = max(self.overlap, self.inside)
            time.sleep(0.002 * (1 + self.calls % 3))
            with self.lock:
                self.inside -= 1
                self.calls += 1
            return b""

    class P:
        def __init__(self):
            self.ctx = Ctx()

    ps = [P() for _ in range(4)]
    res = bench.throughput(ps, [None] * 4, 22, 9)
    assert [p.ctx.overlap for p in ps] == [1, 1, 1, 1]
    assert sum(p.ctx.calls for p in ps) == 22 + 9 and res["steps"] == 22

    class Boom(Ctx):
        def prove_tables(self, bufs, luts=None):
            raise RuntimeError("ProverError(ConstraintsNotSatisfied)")
    ps[2].ctx = Boom()
    try:
        bench.throughput(ps, [None] * 4, 8, 4)
        raise AssertionError("a failing context must fail the sub-result")
    except RuntimeError as e:
        assert "ConstraintsNotSatisfied" in str(e)


def test_bench_gpus_8_over_gloo_with_the_emulation_build():
    """The 8-rank launch the driver's node will see first, end to end on CPU (gloo, emulation build): 8 ranks in the
    process group, per-rank throughput and affinity records in the line, the sharded sub-result over all 8 ranks."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    emu = os.path.join(root, "tests", "emu", "libluminair_emu.so")
    r, line = _run_bench(["--gpus", "8", "--emu-library", emu, "--log-rows", "5", "--steps", "2", "--warmup", "1"], timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 8 and line["config"]["ranks_in_process_group"] == 8 and line["errors"] == []
    pr = line["config"]["per_rank_proofs_per_s"]
    assert len(pr["all"]) == 8 and 0 < pr["min"] <= pr["max"]
    assert line["value"] <= 8 * pr["max"] * 1.0001                 # the headline is the max-over-ranks time, never better than the ranks
    assert len(line["config"]["cpu_affinity"]) == 8 and all(a["pinned"] is False for a in line["config"]["cpu_affinity"])
    assert line["sharded_proof"]["config_2a"]["bytes_identical_to_unsharded_proof"] is True
    assert "cpu_baseline" not in line                              # not in emulation runs; on GPU runs rank 0 times it after its GPU work


def test_rank_affinity_follows_the_gpus_numa_node(tmp_path):
    """pin_rank_to_gpu_numa_node's building blocks on a fake sysfs: 8 GPUs on two NUMA nodes, 4 ranks per node, each
    rank an equal share of its node's CPUs; a cpuset that leaves a rank fewer than 2 CPUs pins nothing."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    for g in range(8):
        d = tmp_path / ("0000:%02x:00.0" % (0x10 + g))
        d.mkdir()
        (d / "numa_node").write_text("%d\n" % (g // 4))
        (d / "local_cpulist").write_text("0-63,128-191\n" if g < 4 else "64-127,192-255\n")
    node, cpus = bench.gpu_local_cpus("0000:15:00.0", str(tmp_path))
    assert node == 1 and len(cpus) == 128 and 64 in cpus and 255 in cpus
    assert bench.gpu_local_cpus("0000:99:00.0", str(tmp_path)) == (None, set())
    allowed = set(range(256))
    shares = [bench.rank_cpu_slice(cpus, allowed, 4, i) for i in range(4)]
    assert all(len(s) == 32 for s in shares) and set().union(*shares) == cpus
    assert all(not (shares[i] & shares[j]) for i in range(4) for j in range(i))
    assert bench.rank_cpu_slice(cpus, {64, 65, 66}, 4, 0) == set()          # container cpuset too small: no pinning
    assert bench._format_cpulist(shares[0]) == "64-95" and bench._parse_cpulist("64-95") == shares[0]
    assert bench.pin_rank_to_gpu_numa_node(0, 1)["pinned"] is False         # one rank: nothing to separate
