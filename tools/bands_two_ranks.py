"""Rehearsal of the row-band pipeline with REAL ranks: `python -m torch.distributed.run --nproc-per-node 2 tools/bands_two_ranks.py`
starts two processes that share one GPU (gloo transport; RCCL needs one GPU per rank), each searching its band of CTU rows of the
same pictures through hm-16.2_amd/bands.py; rank 0 then gathers the bands and compares them with an unsplit run."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hm-16.2_amd")]
import numpy as np
import torch
import torch.distributed as dist

import bands
import hm355
import synth


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    w, h, bd, qp, pictures, group = int(sys.argv[1]), int(sys.argv[2]), 10, 32, int(sys.argv[3]), int(sys.argv[4])
    h_ctu, w_ctu = (h + 63) // 64, (w + 63) // 64
    enc = hm355.Encoder(w, h, bd, 1, max_batch=pictures)
    for i in range(pictures):
        enc.upload(i, synth.frame(w, h, bd, i, 1234))
    tr = bands.TorchTransport(dist, torch)
    dist.barrier()
    t0 = time.perf_counter()
    ms = bands.run_banded(enc, pictures, group, h_ctu, rank, world, tr.send, tr.recv, qp)
    dist.barrier()
    dt = time.perf_counter() - t0
    first, last = bands.band_rows(h_ctu, world, rank)
    mine = [enc.download(i)[1][first * w_ctu:(last + 1) * w_ctu] for i in range(pictures)]
    parts = [None] * world
    dist.gather_object(mine, parts if rank == 0 else None, dst=0)
    if rank == 0:
        enc.run(pictures, qp)
        ok = True
        for i in range(pictures):
            whole = enc.download(i)[1]
            got = np.concatenate([parts[r][i] for r in range(world)])
            ok = ok and all(np.array_equal(got[f], whole[f]) for f in whole.dtype.names)
        print(f"{world} ranks, {pictures} pictures {w}x{h} in groups of {group}: {dt:.2f} s wall, kernel ms of rank 0 {ms:.0f}, "
              f"boundary {enc.boundary_bytes()} B per picture; bands equal the unsplit run: {ok}", flush=True)
        assert ok
    enc.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
