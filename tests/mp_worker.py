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
    dist.barrier()
    if rank == 0:
        print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
