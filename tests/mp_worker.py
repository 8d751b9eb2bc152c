"""Worker of tests/test_multi_gpu_cpu.py: one rank of a world_size-N gloo job (started by torch.distributed.run) that drives the
PRODUCT's multi-rank code -- bench.py's rank partition / step plan / max-over-ranks aggregation and hm-16.2_amd/bands.py's row-band
pipeline with its torch.distributed transport -- around a recording engine instead of the GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hm-16.2_amd")]
import numpy as np
import torch
import torch.distributed as dist

import bands
import bench
import gop_shard

RA_GOP = [dict(poc=0, refs=[], depth=0), dict(poc=8, refs=[0], depth=0), dict(poc=4, refs=[0, 8], depth=1), dict(poc=2, refs=[0, 4], depth=2),
          dict(poc=1, refs=[0, 2], depth=3), dict(poc=3, refs=[2, 4], depth=3), dict(poc=6, refs=[4, 8], depth=2), dict(poc=5, refs=[4, 6], depth=3),
          dict(poc=7, refs=[6, 8], depth=3)]          # cfg/encoder_randomaccess_main10.cfg:24-31 with its 2 active references per list


class RecordingGopEngine:
    """stands in for the device in hm-16.2_amd/gop_shard.py: a finished picture is (poc, who encoded it); its blob carries both"""

    def __init__(self, rank):
        self.rank, self.encoded, self.imported = rank, [], []

    def encode(self, pic, refs, prev_rates):
        assert sorted(refs) == sorted(pic["refs"]) and all(h[0] == r for r, h in refs.items()), "a picture ran before its references were finished here"
        self.encoded.append((pic["poc"], tuple(prev_rates)))
        return (pic["poc"], self.rank), {"poc": pic["poc"]}, (pic["poc"] + 0.25, pic["poc"] + 0.5, pic["poc"] + 0.75)

    def export(self, handle, rates):
        return np.frombuffer(np.array([handle[0], handle[1], *rates], np.float64).tobytes(), np.uint8).copy()

    def blob_like(self):
        return np.zeros(40, np.uint8)

    def imp(self, blob):
        v = np.frombuffer(np.asarray(blob, np.uint8).tobytes(), np.float64)
        self.imported.append(int(v[0]))
        return (int(v[0]), int(v[1])), tuple(float(x) for x in v[2:5])


def payload(rank, slot, row, nbytes):
    return np.random.default_rng(1000003 * rank + 1009 * slot + row).integers(0, 256, nbytes, dtype=np.uint8)


class RecordingEngine:
    """stands in for hm355.Encoder: checks the order of the calls the pipeline makes and the bytes it moves"""

    def __init__(self, rank, world, h_ctu, nbytes):
        self.rank, self.world, self.h_ctu, self.nbytes = rank, world, h_ctu, nbytes
        self.imported, self.searched, self.launches = set(), set(), []

    def boundary_bytes(self):
        return self.nbytes

    def import_boundary(self, slot, row, data):
        first, _ = bands.band_rows(self.h_ctu, self.world, self.rank)
        assert row == first - 1, "only the row right above the band is imported"
        assert np.array_equal(np.asarray(data), payload(self.rank - 1, slot, row, self.nbytes)), "boundary bytes changed in transit"
        self.imported.add(slot)

    def run_rows(self, slot0, n, qp, r0, r1):
        assert (r0, r1) == bands.band_rows(self.h_ctu, self.world, self.rank)
        for s in range(slot0, slot0 + n):
            assert r0 == 0 or s in self.imported, "a band ran before the row above it arrived"
            assert s not in self.searched
            self.searched.add(s)
        self.launches.append((slot0, n))
        return 1.0, 1

    def export_boundary(self, slot, row):
        assert slot in self.searched and row == bands.band_rows(self.h_ctu, self.world, self.rank)[1]
        return payload(self.rank, slot, row, self.nbytes)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    out = {}
    # ---- picture replicas (bench.py default): disjoint frame numbers, one step plan for all ranks, slowest rank's time
    mine = bench.rank_frame_numbers(rank, 16)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    step_s = bench.agree_over_ranks(dist, torch, 10.0 + rank, dist.ReduceOp.MAX, "cpu")      # rank 1 is the slower one
    plan = bench.plan_steps(step_s, 5, 20, 300.0)
    plans = [None] * world
    dist.all_gather_object(plans, plan)
    out["frames"], out["step_s"], out["plans"] = gathered, step_s, plans
    # ---- row bands: the pipeline of hm-16.2_amd/bands.py over gloo
    h_ctu, pictures, group, nbytes = 5, 7, 3, 4096
    eng = RecordingEngine(rank, world, h_ctu, nbytes)
    tr = bands.TorchTransport(dist, torch)
    ms = bands.run_banded(eng, pictures, group, h_ctu, rank, world, tr.send, tr.recv, 32)
    rows = [None] * world
    dist.all_gather_object(rows, (bands.band_rows(h_ctu, world, rank), sorted(eng.searched), eng.launches, ms))
    out["bands"] = rows
    # ---- pictures of one temporal layer on different ranks: hm-16.2_amd/gop_shard.py over gloo
    geng = RecordingGopEngine(rank)
    res, done = gop_shard.run_gop(geng, RA_GOP, rank, world, gop_shard.TorchAllGather(dist, torch))
    gop = [None] * world
    dist.all_gather_object(gop, (geng.encoded, sorted(geng.imported), sorted((poc, h[0][1]) for poc, h in done.items())))
    out["gop"] = gop
    dist.barrier()
    if rank == 0:
        print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
