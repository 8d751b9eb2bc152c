"""CPU suite: the N>1 paths of the product -- bench.py's picture replicas (rank partition, one step plan for all ranks,
max-over-ranks time) and the CTU-row band pipeline of hm-16.2_amd/bands.py with its torch.distributed transport -- run with
world_size 2 (and 3) on the gloo backend.  tests/mp_worker.py drives that product code around a recording engine; the GPU
side of the same pipeline is tests/test_gpu_parity.py::test_hip_row_bands_match_unsplit_picture."""
import json
import os
import subprocess
import sys

import pytest

import common


def _run(world, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(common.ROOT, "tests", "mp_worker.py")],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    return json.loads([l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("world,port", [(2, 29533), (3, 29534)])
def test_ranks_with_gloo(world, port):
    import bands
    import bench
    res = _run(world, port)
    # replicas: every rank its own 16 frames, nothing shared; all ranks run the plan of the slowest one
    flat = [f for fr in res["frames"] for f in fr]
    assert len(flat) == 16 * world == len(set(flat))
    assert res["step_s"] == 10.0 + world - 1
    assert all(tuple(p) == bench.plan_steps(res["step_s"], 5, 20, 300.0) for p in res["plans"])
    # bands: the rows of every picture are covered exactly once, top to bottom, and every rank searched all 7 pictures in groups of 3
    covered = []
    for r, ((first, last), searched, launches, ms) in enumerate(res["bands"]):
        assert (first, last) == bands.band_rows(5, world, r)
        covered += list(range(first, last + 1))
        assert searched == list(range(7)) and [tuple(l) for l in launches] == [(0, 3), (3, 3), (6, 1)] and ms == 3.0
    assert covered == list(range(5))
    # gop_shard: every picture encoded exactly once, by the rank the level plan names, after its references; every rank ends up holding every
    # finished picture (its own or imported from the owner); the SAO rates a picture reads are those of the picture gop_shard names
    import gop_shard
    import mp_worker
    pics = mp_worker.RA_GOP
    want_owner = {}
    for level in gop_shard.levels(pics):
        for k, i in enumerate(level):
            want_owner[pics[i]["poc"]] = gop_shard.owner(k, world)
    encoded_by = {}
    for r, (encoded, imported, holds) in enumerate(res["gop"]):
        for poc, prev in encoded:
            assert poc not in encoded_by
            encoded_by[poc] = r
            src = gop_shard.sao_rate_source(pics, [p["poc"] for p in pics].index(poc))
            want = (0.0, 0.0, 0.0) if src is None else (pics[src]["poc"] + 0.25, pics[src]["poc"] + 0.5, pics[src]["poc"] + 0.75)
            assert tuple(prev) == want, f"POC {poc}: SAO rates of the wrong picture"
        assert [tuple(h) for h in holds] == sorted((poc, o) for poc, o in want_owner.items())
        assert sorted(imported) == sorted(poc for poc, o in want_owner.items() if o != r)
    assert encoded_by == want_owner


@pytest.mark.parametrize("h_ctu,world", [(34, 8), (34, 1), (4, 2), (5, 3), (3, 8), (17, 4)])
def test_band_partition(h_ctu, world):
    import bands
    rows = []
    for r in range(world):
        first, last = bands.band_rows(h_ctu, world, r)
        rows += list(range(first, last + 1))
        assert last - first + 1 in (h_ctu // world, h_ctu // world + 1, 0)
    assert rows == list(range(h_ctu))


def test_step_plan():
    import bench
    assert bench.plan_steps(1.0, 5, 20, 300.0) == (5, 20)                # fits: as requested
    assert bench.plan_steps(43.5, 5, 20, 300.0) == (1, 5)                # the driver's command on the round-1 kernel
    assert bench.plan_steps(400.0, 5, 20, 300.0) == (1, 1)               # always at least one timed step
    assert bench.plan_steps(1.0, 0, 3, 300.0) == (1, 3)                  # the timed first step is the warm-up


@pytest.mark.parametrize("world,rows", [(1, False), (8, False), (8, True)])
def test_bench_line_is_complete(world, rows):
    """the JSON line bench.py prints for the headline workload carries every field of the contract (built without a GPU)"""
    import argparse
    import bench
    args = argparse.Namespace(width=3840, height=2160, qp=32, frames=384, steps=20, warmup=5, budget_s=280.0, lanes=1 if rows else 3)
    # 7 launches of 3 x 248 / 7 s each, three in flight at a time: 248 s of timed region
    line = bench.intra_line(args, world, rows, 12 if rows else 0, 7, 1, 250.0, 248000.0 * (1 if rows else 3), 7, 2040 * 384 * 7 * (1 if rows else world), 2040 * 384 * 7, "0123456789ab")
    json.dumps(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in line
    assert line["n_gpus"] == world and line["scaling"] == ("strong" if rows else "weak") and "workload" in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["unit"] == "GB/s"
    assert abs(line["value"] - 2040 * 384 * 7 * (1 if rows else world) / 250.0) < 1e-6
    # achieved = algorithmic bytes of all launches of the rank / timed region, however many launches overlap
    assert abs(r["achieved"] - 54278 * 2040 * 384 * 7 / (250.0 if not rows else 248.0) / 1e9) < 1e-9
    assert r["traffic"] is None and "traffic_note" in r          # no PMC record of build 0123456789ab under profiles/
    assert bench.plan_steps(9.0, 5, 20, 280.0, first=4) == (5, 20) and bench.plan_steps(30.0, 5, 20, 280.0, first=4) == (4, 5)
