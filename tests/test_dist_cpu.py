"""CPU tests (gloo, world_size 2) of the N > 1 plumbing bench.py uses: every rank derives the same slab bounds, the
128-byte communicator id reaches every rank, and the max-over-ranks timing reduction works. The compute path itself is
GPU-only and is covered by tests/test_gpu_slabs.py (virtual slabs)."""
import os
import subprocess
import sys
import textwrap

import libfluid_amd as lfa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_balanced_layer_bounds():
    assert lfa.balanced_layer_bounds(64, 1) == [0, 64]
    assert lfa.balanced_layer_bounds(64, 8, 0, 32) == [0, 4, 8, 12, 16, 20, 24, 28, 64]
    b = lfa.balanced_layer_bounds(13, 4, 2, 9)
    assert b[0] == 0 and b[-1] == 13 and all(b[i] < b[i + 1] for i in range(4))
    b = lfa.balanced_layer_bounds(4, 4, 1, 2)  # fewer fluid layers than ranks: still a partition into non-empty slabs
    assert b == [0, 1, 2, 3, 4]


def test_two_rank_plumbing_over_gloo(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        import libfluid_amd as lfa
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid.copy_(torch.arange(128, dtype=torch.uint8))
        dist.broadcast(uid, src=0)
        assert uid.tolist() == list(range(128))
        bounds = lfa.balanced_layer_bounds(64 * world, world, 0, 32 * world)
        gathered = [None] * world
        dist.all_gather_object(gathered, bounds)
        assert all(g == bounds for g in gathered)
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == float(world)
        # the transport hand-shake of bench.py: "did every rank get its communicator" (MIN over ranks: one failure moves all
        # of them to the shared-memory transport), then rank 0's segment name reaches every rank
        ok = torch.tensor([0 if rank == 1 else 1], dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        assert int(ok.item()) == 0
        names = [f"/lfa_test_{{os.getpid()}}"] if rank == 0 else [None]
        dist.broadcast_object_list(names, src=0)
        gathered = [None] * world
        dist.all_gather_object(gathered, names[0])
        assert names[0].startswith("/lfa_test_") and all(g == names[0] for g in gathered)
        dist.barrier()
        dist.destroy_process_group()
        print("rank", rank, "ok")
    """))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29577", str(script)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_bench_refuses_a_rank_count_that_does_not_match_gpus():
    """bench.py --gpus N inside a rank environment of another size exits non-zero before it touches a GPU: the JSON line would
    carry the wrong n_gpus otherwise (round 3 printed `n_gpus: 1` for `--gpus 4` without ranks)."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)
