"""CPU suite: the N>1 path of bench.py -- pictures sharded over ranks, barrier + max-over-ranks timing --
rehearsed with world_size 2 on the gloo backend (no GPU).  The per-rank work is replaced by the oracle on a
tiny picture; what is under test is the sharding (disjoint frame ranges, whole-job aggregation)."""
import os
import subprocess
import sys
import textwrap

import common

WORKER = textwrap.dedent("""
    import os, sys, time, json
    sys.path[:0] = [r"{root}", r"{root}/hm-16.2_amd", r"{root}/oracle"]
    import numpy as np, torch, torch.distributed as dist
    import oracle, synth
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    frames_per_rank, w, h, bd, qp = 2, 64, 64, 8, 32
    # weak scaling: rank r owns pictures [r*F, (r+1)*F) -- no data-path collective
    mine = [synth.frame(w, h, bd, rank * frames_per_rank + i, 1234) for i in range(frames_per_rank)]
    dist.barrier(); t0 = time.perf_counter()
    bits = 0
    for p in mine:
        rec, ctus = oracle.compress(p, bd, qp, 1)
        bits += int(ctus["total_bits"].sum())
    dist.barrier(); dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
    b = torch.tensor([bits], dtype=torch.int64); gathered = [torch.zeros_like(b) for _ in range(world)]; dist.all_gather(gathered, b)
    if rank == 0:
        print(json.dumps({{"max_dt": float(t.item()), "bits": [int(x.item()) for x in gathered], "ctus": world * frames_per_rank}}))
    dist.destroy_process_group()
""")


def test_two_rank_sharding_with_gloo(built, tmp_path):
    import json
    import synth
    import oracle
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=common.ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = [l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["ctus"] == 4 and res["max_dt"] > 0
    # every rank encoded ITS OWN pictures: the per-rank bit totals equal a single-process run of the same frames
    want = []
    for r in range(2):
        bits = 0
        for i in range(2):
            _, ctus = oracle.compress(synth.frame(64, 64, 8, r * 2 + i, 1234), 8, 32, 1)
            bits += int(ctus["total_bits"].sum())
        want.append(bits)
    assert res["bits"] == want
