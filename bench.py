#!/usr/bin/env python3
"""Benchmark of the hm355 CTU RD-search path (BASELINE.json metric: CTUs/sec (enc) at 4K main10).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the hot path (the TEncSlice::compressSlice replacement) over one batch of
independent all-intra pictures that are already resident in HBM: synthetic 3840x2160 10-bit frames,
encoder_intra_main10 parameters, QP 32, WaveFrontSynchro=1 (BASELINE.json configs[3], the configuration
the metric is quoted on; it fits one GPU).  By default every step is one launch over `--frames` pictures
(768: the wavefront fill and drain are a small share of the launch); the timed region holds exactly K steps
between two barrier + synchronize pairs.  `--lanes L` (optional, default 1) pipelines consecutive steps over L
HIP streams (hm355_run_begin / hm355_run_wait): each step has its own picture slots and its own launch and
step k + L is issued once step k has finished, so the drain of one step overlaps the fill of the next.
Measured: two launches do overlap, but 2 x 192 or 4 x 160 pictures in flight run below
one 768-picture launch, so the headline keeps the single launch.  `--workload c2|c3` run BASELINE.json's
configs[1] / configs[2] exactly as stated (one picture; a 16-picture low-delay-P stream) for the latency figures.
Every rank owns one GPU and its own batches (weak scaling: the path shards by picture with no data-path
collective); RCCL is used only for the barrier / max-time.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "hm-16.2_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

ALG_BYTES_PER_CTU = 54278          # org 12,288 + recon 12,288 + neighbour lines ~1,030 + coeff 24,576 + CU metadata 4,096
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s

# encoder_intra_main10 parameters restated as a config for the reference binary
REF_CFG = """MaxCUWidth : 64
MaxCUHeight : 64
MaxPartitionDepth : 4
QuadtreeTULog2MaxSize : 5
QuadtreeTULog2MinSize : 2
QuadtreeTUMaxDepthInter : 3
QuadtreeTUMaxDepthIntra : 3
IntraPeriod : 1
DecodingRefreshType : 0
GOPSize : 1
FastSearch : 1
SearchRange : 64
HadamardME : 1
FEN : 1
FDM : 1
QP : 32
MaxDeltaQP : 0
MaxCuDQPDepth : 0
DeltaQpRD : 0
RDOQ : 1
RDOQTS : 1
SAO : 1
AMP : 1
TransformSkip : 1
TransformSkipFast : 1
InternalBitDepth : 10
Profile : main10
"""


DISTINCT_FRAMES = 16


def rank_frame_numbers(rank, count=DISTINCT_FRAMES):
    """frame numbers of the synthetic clip that rank `rank` encodes: every rank owns its own pictures (weak scaling by picture)"""
    return [DISTINCT_FRAMES * rank + f for f in range(count)]


def plan_steps(step_s, warmup, steps, budget_s, first=1):
    """(warm-up, timed) steps that fit `budget_s` seconds of wall time when one step takes `step_s`; `first` warm-up steps (the ones
    that were just timed) have already run.  The requested counts are kept whenever they fit."""
    warmup = max(first, warmup)
    if step_s * (warmup + steps) <= budget_s:
        return warmup, steps
    return first, max(1, min(steps, int((budget_s - step_s * first) / max(step_s, 1e-9))))


def agree_over_ranks(dist, torch, v, op, device):
    """a host value every rank must share: all-reduce with `op` (MAX for times: the slowest rank's clock counts)"""
    if dist is None:
        return v
    t = torch.tensor([float(v)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=op)
    return float(t.item())


def _frame_job(a):
    import synth
    return synth.frame(*a)


def gpu_runtime_loaded():
    """True when the HIP / HSA runtime is already mapped into this process (or that cannot be told): no fork then"""
    try:
        with open("/proc/self/maps") as fh:
            maps = fh.read()
    except OSError:
        return True
    return any(lib in maps for lib in ("libamdhip64", "libhsa-runtime64", "librocprofiler", "libroctracer"))


def distinct_frames(width, height, bit_depth, numbers, seed):
    """the synthetic frames `numbers`, generated on a few host cores (call before the process touches the GPU: the pool forks)"""
    import synth
    jobs = [(width, height, bit_depth, f, seed) for f in numbers]
    # A tool preloaded into this process (rocprofv3, any other injected library) may have initialised the GPU before main(): a forked
    # child would then hold the device.  The pool is only used when neither the HIP nor the HSA runtime is mapped into the process yet.
    if len(jobs) < 4 or os.environ.get("HM355_NO_FORK") or gpu_runtime_loaded():
        return [synth.frame(*j) for j in jobs]
    import multiprocessing as mp
    with mp.get_context("fork").Pool(min(8, len(jobs), os.cpu_count() or 1)) as pool:
        return pool.map(_frame_job, jobs)


def cpu_baseline(width, bit_depth, qp, seed):
    """The reference's own CPU path (oracle/_ref/hm_encoder, built from /root/reference in the dev container),
    timed on a bounded sample of the same workload: the top 4 CTU rows (width x 256) of frame 0, 1 core
    (HM is single threaded).  Falls back to the C restatement (oracle/) when the reference binary is absent."""
    import synth
    rows = 12
    h = rows * 64
    y, u, v = synth.frame(width, 2160 if width == 3840 else h, bit_depth, 0, seed)
    y, u, v = y[:h], u[:h // 2], v[:h // 2]
    n_ctus = ((width + 63) // 64) * rows
    sample = f"top {rows} CTU rows ({width}x{h}, {n_ctus} CTUs) of frame 0, WaveFrontSynchro=1, QP {qp}"
    enc = os.path.join(ROOT, "oracle", "_ref", "hm_encoder")
    if os.path.exists(enc):
        with tempfile.TemporaryDirectory() as td:
            yuv = os.path.join(td, "in.yuv")
            with open(yuv, "wb") as fh:
                for p in (y, u, v):
                    fh.write(np.ascontiguousarray(p).astype("<u2").tobytes())
            cfg = os.path.join(td, "intra_main10.cfg")
            open(cfg, "w").write(REF_CFG)
            cmd = [enc, "-c", cfg, "-i", yuv, "-wdt", str(width), "-hgt", str(h), "-fr", "50", "-f", "1",
                   f"--InputBitDepth={bit_depth}", "-q", str(qp), "--WaveFrontSynchro=1", "-b", os.path.join(td, "o.bin"),
                   "-o", os.path.join(td, "r.yuv")]
            t0 = time.time()
            out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
            dt = time.time() - t0
            for line in out.splitlines():
                if "Total Time" in line:      # HM's own clock (encmain.cpp:95-102): whole encoder incl. deblock/SAO/entropy (<5 %)
                    dt = float(line.split()[2])
        return {"value": n_ctus / dt, "unit": "CTU/s", "cores": 1, "kind": "reference", "sample": sample + "; HM 'Total Time'"}
    import oracle
    t0 = time.time()
    oracle.compress((y, u, v), bit_depth, qp, 1)
    dt = time.time() - t0
    return {"value": n_ctus / dt, "unit": "CTU/s", "cores": 1, "kind": "port", "sample": sample}


# GOP tables of the reference's inter configurations (cfg/encoder_lowdelay_P_main.cfg:24-27, cfg/encoder_randomaccess_main10.cfg:24-31)
REF_CFG_INTER = {
    "ldp_p": ("main", 8, """IntraPeriod : -1
DecodingRefreshType : 0
GOPSize : 4
Frame1:  P    1   3        0.4624   0            0               0           4                4         -1 -5 -9 -13       0
Frame2:  P    2   2        0.4624   0            0               0           4                4         -1 -2 -6 -10       1      -1       5         1 1 1 0 1
Frame3:  P    3   3        0.4624   0            0               0           4                4         -1 -3 -7 -11       1      -1       5         0 1 1 1 1
Frame4:  P    4   1        0.578    0            0               0           4                4         -1 -4 -8 -12       1      -1       5         0 1 1 1 1
"""),
    "ra_b": ("main10", 10, """IntraPeriod : 32
DecodingRefreshType : 1
GOPSize : 8
Frame1:  B    8   1        0.442    0            0              0           4                4         -8 -10 -12 -16         0
Frame2:  B    4   2        0.3536   0            0              0           2                3         -4 -6  4               1       4        5         1 1 0 0 1
Frame3:  B    2   3        0.3536   0            0              0           2                4         -2 -4  2 6             1       2        4         1 1 1 1
Frame4:  B    1   4        0.68     0            0              1           2                4         -1  1  3 7             1       1        5         1 0 1 1 1
Frame5:  B    3   4        0.68     0            0              1           2                4         -1 -3  1 5             1      -2        5         1 1 1 1 0
Frame6:  B    6   3        0.3536   0            0              0           2                4         -2 -4 -6 2             1      -3        5         1 1 1 1 0
Frame7:  B    5   4        0.68     0            0              1           2                4         -1 -5  1 3             1       1        5         1 0 1 1 1
Frame8:  B    7   4        0.68     0            0              1           2                4         -1 -3 -7 1             1      -2        5         1 1 1 1 0
"""),
}


def cpu_baseline_inter(kind, qp, width, height):
    """The reference's own encoder (oracle/_ref/hm_encoder) on one host core for the same configuration, on a bounded sample of the
    benched picture size: the top 4 CTU rows (width x 256) of the same synthetic clip, encoded with 1 picture and with 1 + n inter
    pictures; the difference is the time of the n inter pictures (whole encoder: the search dominates)."""
    import synth
    enc = os.path.join(ROOT, "oracle", "_ref", "hm_encoder")
    if not os.path.exists(enc):
        return None
    profile, bd, gop = REF_CFG_INTER[kind]
    w, h, n_inter = width, 256, 4
    common = REF_CFG.split("IntraPeriod")[0] + "".join(l + "\n" for l in REF_CFG.splitlines() if l.split(" ")[0] in (
        "FastSearch", "SearchRange", "HadamardME", "FEN", "FDM", "QP", "MaxDeltaQP", "MaxCuDQPDepth", "DeltaQpRD", "RDOQ", "RDOQTS", "SAO", "AMP",
        "TransformSkip", "TransformSkipFast"))
    cfg_text = common + gop + f"BipredSearchRange : 4\nInternalBitDepth : {bd}\nProfile : {profile}\n"
    times = []
    with tempfile.TemporaryDirectory() as td:
        yuv = os.path.join(td, "in.yuv")
        with open(yuv, "wb") as fh:
            for f in range(1 + n_inter):
                y, u, v = synth.frame(w, height, bd, f, 1234)
                for p in (y[:h], u[:h // 2], v[:h // 2]):
                    fh.write(np.ascontiguousarray(p).astype(np.uint8 if bd == 8 else "<u2").tobytes())
        cfg = os.path.join(td, "inter.cfg")
        open(cfg, "w").write(cfg_text)
        for frames in (1, 1 + n_inter):
            cmd = [enc, "-c", cfg, "-i", yuv, "-wdt", str(w), "-hgt", str(h), "-fr", "50", "-f", str(frames), f"--InputBitDepth={bd}", "-q", str(qp),
                   "-b", os.path.join(td, "o.bin"), "-o", os.path.join(td, "r.yuv")]
            t0 = time.time()
            out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
            dt = time.time() - t0
            for line in out.splitlines():
                if "Total Time" in line:
                    dt = float(line.split()[2])
            times.append(dt)
    n_ctus = ((w + 63) // 64) * 4 * n_inter
    return {"value": n_ctus / max(1e-9, times[1] - times[0]), "unit": "CTU/s", "cores": 1, "kind": "reference",
            "sample": f"{n_inter} inter pictures of the top 4 CTU rows ({w}x{h}, {bd}-bit) of the benched clip ({n_ctus} CTUs), same cfg, QP {qp}; HM 'Total Time' of 1+{n_inter} pictures minus 1 picture"}


def run_inter(args, torch, world=1, rank=0, local_rank=0):
    """Secondary workloads: inter slices through hm355_compress_slices_inter, `--frames` independent streams per step (one current
    picture each, WaveFrontSynchro=1); the reference pictures are HIP-path I-slice reconstructions of the same synthetic clip,
    shifted per stream so that every stream has its own data.
      ldp_p (BASELINE.json configs[2]): encoder_lowdelay_P_main P slices, 1920x1080 8-bit, 4 references, SearchRange 64
      ra_b  (BASELINE.json configs[4]): encoder_randomaccess_main10 B slices, 3840x2160 10-bit, 2 + 2 references (POC 4 between POC 0 and 8)
    Reference pictures enter through host buffers in this entry point, so `value` is CTUs / HIP-event kernel time and `call_s` is
    the PCIe-inclusive wall time of the call."""
    import math
    import hm355
    import synth
    dist = None
    if world > 1:      # every rank its own streams (independent GOP chains shard with no data-path collective): weak scaling; RCCL carries the barrier and the max time
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    kind, qp, S = args.workload, args.qp, args.frames
    if kind == "ldp_p":
        w, h, bd, nref, qpp, qpf, ref_idx_pocs, cur_poc = 1920, 1080, 8, 4, qp + 3, 0.4624, [3, 2, 1, 0], 4
    else:
        w, h, bd, nref, qpp, qpf, ref_idx_pocs, cur_poc = 3840, 2160, 10, 2, qp + 2, 0.3536, [0, 8], 4
    enc = hm355.Encoder(w, h, bd, 1, max(S, nref))
    if os.environ.get("HM355_SHARE"):       # diagnostic: fewer resident searches (hm355_set_lane_share caps the launch at 3840 / share)
        enc.set_lane_share(int(os.environ["HM355_SHARE"]))
    n = enc.num_ctus
    ref_frames = sorted(set(ref_idx_pocs))
    res = enc.compress([synth.frame(w, h, bd, f, 1234) for f in ref_frames], qp)
    refs = {}
    for k, f in enumerate(ref_frames):
        refs[f] = dict(slice_type=2, rec=res[k][0], pred_mode=np.ones((n, 256), np.uint8), mv=[np.zeros((n, 256, 2), np.int16)] * 2,
                       ref_idx=[np.full((n, 256), -1, np.int8)] * 2, num_ref_idx=(0, 0), ref_poc=np.zeros((2, 16), np.int32),
                       ref_long_term=np.zeros((2, 16), np.int32))
    lam = qpf * 2.0 ** ((qpp - 12) / 3.0) * min(4.0, max(2.0, (qpp - 12) / 6.0))          # TEncSlice.cpp:323-352
    ref_poc = np.zeros((2, 16), np.int32)
    ref_poc[0, :nref] = ref_idx_pocs
    if kind == "ra_b":
        ref_poc[1, :nref] = ref_idx_pocs[::-1]
    sp = dict(slice_type=1 if kind == "ldp_p" else 0, qp=qpp, chroma_weight=hm355.intra_lambda(qpp)[1], poc=cur_poc,
              cabac_init_type=1 if kind == "ldp_p" else 0, num_ref_idx=(nref, 0 if kind == "ldp_p" else nref), ref_poc=ref_poc,
              col_from_l0=1, col_ref_idx=0, tmvp=1, mvd_l1_zero=0, max_merge_cand=5, check_ldc=1 if kind == "ldp_p" else 0,
              lambda_motion_sad=int(math.floor(65536.0 * math.sqrt(lam))), lambda_motion_sse=int(math.floor(65536.0 * lam)))
    sp["lambda"] = lam
    # every stream gets its own pictures in HBM: the clip shifted by a stream-specific offset (whole CTU-misaligned steps), so streams
    # neither share reference data in the caches nor repeat each other's decisions
    cur0 = synth.frame(w, h, bd, cur_poc, 1234)
    def shifted(planes, k):
        k += rank * S                               # other ranks, other streams
        dx, dy = (24 * k) % w, (8 * k) % h
        return [np.ascontiguousarray(np.roll(np.roll(p, dy >> (1 if i else 0), axis=0), dx >> (1 if i else 0), axis=1)) for i, p in enumerate(planes)]
    jobs = []
    for k in range(S):
        rk = {f: dict(r, rec=shifted(r["rec"], k)) for f, r in refs.items()} if (k or rank) else refs
        jobs.append((shifted(cur0, k) if (k or rank) else cur0, sp, rk))
    kernel_ms, wall = 0.0, 0.0
    for it in range(args.warmup + args.steps):
        if dist is not None:
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = enc.compress_inter_batch(jobs)
        torch.cuda.synchronize()
        k, l = hm355.C.c_double(), hm355.C.c_int()
        enc.lib.hm355_last_run_info(enc.h_, hm355.C.byref(k), hm355.C.byref(l))
        if it >= args.warmup:
            kernel_ms += k.value
            wall += time.perf_counter() - t0
    ctus = n * S * args.steps
    alg = 54278 + (nref if kind == "ldp_p" else 2 * nref) * 80000 + 4608        # intra bytes + search window per reference + MV fields
    ach = alg * ctus / (kernel_ms * 1e-3) / 1e9                                 # this rank's kernel
    if dist is not None:
        kernel_ms = agree_over_ranks(dist, torch, kernel_ms, dist.ReduceOp.MAX, "cuda"); wall = agree_over_ranks(dist, torch, wall, dist.ReduceOp.MAX, "cuda")
        ctus *= world
        if rank != 0:
            enc.close(); dist.barrier(); dist.destroy_process_group()
            return
    what = ("encoder_lowdelay_P_main P slices, synthetic 1920x1080 8-bit, 4 references" if kind == "ldp_p"
            else "encoder_randomaccess_main10 B slices, synthetic 3840x2160 10-bit, 2 + 2 references, BipredSearchRange 4")
    line = {
        "metric": f"CTUs/sec (enc), {'P' if kind == 'ldp_p' else 'B'} slices; bit-exact CU partition / MV vs HM", "value": ctus / (kernel_ms * 1e-3),
        "unit": "CTU/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": kernel_ms / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "int32+f64", "data": "synthetic", "call_s": wall / args.steps,
        "config": {"workload": f"{what}, SearchRange 64, QP {qpp}, WaveFrontSynchro=1, {S} independent streams per GPU per step", "streams": S,
                   "ctus_per_step": n * S * world,
                   "mode_mix": {"skip": float(np.mean([(o[2]["skip"] != 0).mean() for o in out])),
                                "bi": float(np.mean([(o[2]["inter_dir"] == 3).mean() for o in out])),
                                "intra": float(np.mean([(o[1]["pred_mode"] == 1).mean() for o in out]))}},
        "roofline": {"bound": "hbm", "kernel": "hm355_ctu_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                     "traffic": None, "note": f"algorithmic bytes {alg} B/CTU (SURVEY 8d)"}}
    if not args.no_cpu_baseline:
        cb = cpu_baseline_inter(kind, qp, w, h)
        if cb:
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu_1core"] = line["value"] / cb["value"]
    print(json.dumps(line))
    enc.close()
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


def run_as_stated(args, torch):
    """BASELINE.json configs[1] and configs[2] exactly as they are stated -- no batching of extra pictures or streams to fill the device, so the
    time of ONE CTU search is what the step takes.
      c2: encoder_intra_main10.cfg, synthetic 1920x1080 10-bit, 8 frames (WaveFrontSynchro 0 as the cfg has it: the CABAC state chains through
          all 510 CTUs of a picture, the 8 pictures are the only parallelism).  A step = hm355_run over the 8 resident pictures; the launch runs
          as teams of wavefronts (hm355_team.h).
      c3: encoder_lowdelay_P_main.cfg, synthetic 1920x1080 8-bit, SearchRange 64, one stream in closed loop through the C++ host mirror
          (hm-16.2_amd/hm355_encmain: TEncTop / TEncGOP / TEncSlice look-alikes over the C ABI; search -> deblocking -> SAO -> slice data -> device-resident
          reference, every picture referencing the ones before it).  A step = one P picture; value = CTUs / HIP-event time of its search."""
    import hm355
    import synth
    kind, qp = args.workload, args.qp
    w, h = 1920, 1080
    n_ctus = ((w + 63) // 64) * ((h + 63) // 64)
    enc_bin = os.path.join(ROOT, "oracle", "_ref", "hm_encoder")
    if kind == "c2":
        bd, F = 10, 8
        enc = hm355.Encoder(w, h, bd, 0, F)
        frames = [synth.frame(w, h, bd, f, 1234) for f in range(F)]
        for i in range(F):
            enc.upload(i, frames[i])
        ms_total = 0.0
        t0 = time.perf_counter()
        for it in range(args.warmup + args.steps):
            if it == args.warmup:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            ms, _ = enc.run(F, qp)
            if it >= args.warmup:
                ms_total += ms
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ctus = n_ctus * F * args.steps
        ach = ALG_BYTES_PER_CTU * ctus / (ms_total * 1e-3) / 1e9
        line = {"metric": "CTUs/sec (enc), BASELINE configs[1] as stated; bit-exact CU partition vs HM", "value": ctus / dt, "unit": "CTU/s", "n_gpus": 1, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32+f64", "data": "synthetic",
                "config": {"workload": f"encoder_intra_main10, synthetic {w}x{h} 10-bit, {F} frames, QP {qp}, WaveFrontSynchro=0 (serial CABAC chain per picture), inputs resident in HBM",
                           "frames_per_gpu": F, "ctus_per_step": n_ctus * F, "seconds_per_picture": dt / args.steps},
                "roofline": {"bound": "hbm", "kernel": "hm355_ctu_team_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                             "avg_launch_ms": ms_total / args.steps, "launches": args.steps,
                             "note": "8 serial chains of 510 CTU searches: latency-bound by construction (54,278 algorithmic bytes per CTU, SURVEY 8d)"}}
        if not args.no_cpu_baseline and os.path.exists(enc_bin):
            with tempfile.TemporaryDirectory() as td:
                nf = 3
                yuv = os.path.join(td, "in.yuv")
                with open(yuv, "wb") as fh:
                    for f in range(nf):
                        for p in frames[f]:
                            fh.write(np.ascontiguousarray(p).astype("<u2").tobytes())
                cfg = os.path.join(td, "c.cfg"); open(cfg, "w").write(REF_CFG)
                out = subprocess.run([enc_bin, "-c", cfg, "-i", yuv, "-wdt", str(w), "-hgt", str(h), "-fr", "50", "-f", str(nf), "--InputBitDepth=10", "-q", str(qp),
                                      "-b", os.path.join(td, "o.bin"), "-o", os.path.join(td, "r.yuv")], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
                tt = [float(l.split()[2]) for l in out.splitlines() if "Total Time" in l][0]
            line["cpu_baseline"] = {"value": n_ctus * nf / tt, "unit": "CTU/s", "cores": 1, "kind": "reference",
                                    "sample": f"the first {nf} of the 8 frames, same cfg (whole encoder, HM 'Total Time'; the reference is single threaded)"}
            line["speedup_vs_cpu_1core"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line))
        enc.close()
        return
    # c3: one low-delay P stream, closed loop, through the C++ host mirror
    bd, nf = 8, 1 + args.warmup + args.steps
    exe = os.path.join(ROOT, "hm-16.2_amd", "hm355_encmain")
    with tempfile.TemporaryDirectory() as td:
        yuv = os.path.join(td, "in.yuv")
        synth.write_yuv(yuv, w, h, bd, nf, 1234)
        p = subprocess.run([exe, yuv, str(w), str(h), str(bd), str(nf), str(qp), str(args.wpp), os.path.join(td, "dump.bin"), "ldp"], env=dict(os.environ, HM355_TIMING="1"),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        pics = [json.loads(l) for l in p.stderr.decode().splitlines() if l.startswith("{")]
        timed = [x for x in pics if x["slice_type"] != 2][args.warmup:]
        assert len(timed) == args.steps
        kernel_ms = sum(x["search_kernel_ms"] for x in timed); wall_ms = sum(x["picture_wall_ms"] for x in timed)
        alg = 54278 + 4 * 80000 + 4608
        ach = alg * n_ctus * args.steps / (kernel_ms * 1e-3) / 1e9
        line = {"metric": "CTUs/sec (enc), P slices, BASELINE configs[2] as stated (one stream, closed loop); bit-exact CU partition / MV vs HM", "value": n_ctus * args.steps / (kernel_ms * 1e-3),
                "unit": "CTU/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": kernel_ms / args.steps, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "int32+f64", "data": "synthetic",
                "config": {"workload": f"encoder_lowdelay_P_main, synthetic {w}x{h} 8-bit, SearchRange 64, QP {qp}, WaveFrontSynchro={args.wpp}, ONE stream in closed loop "
                                       "(search -> deblocking -> SAO -> slice data -> device-resident reference per picture), a step = one P picture",
                           "ctus_per_step": n_ctus, "picture_wall_ms_incl_loop_filters_and_slice_data": wall_ms / args.steps,
                           "pictures": [{k: x[k] for k in ("poc", "qp", "bits", "search_kernel_ms")} for x in pics]},
                "roofline": {"bound": "hbm", "kernel": "hm355_ctu_team_kernel (a one-stream launch runs as teams of 9 wavefronts per CTU)", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                             "avg_launch_ms": kernel_ms / args.steps, "launches": args.steps, "note": f"algorithmic bytes {alg} B/CTU (SURVEY 8d: 4 reference windows)"}}
        if not args.no_cpu_baseline and os.path.exists(enc_bin):
            cfg = os.path.join(td, "ldp.cfg")
            common = REF_CFG.split("IntraPeriod")[0] + "".join(l + "\n" for l in REF_CFG.splitlines() if l.split(" ")[0] in (
                "FastSearch", "SearchRange", "HadamardME", "FEN", "FDM", "QP", "MaxDeltaQP", "MaxCuDQPDepth", "DeltaQpRD", "RDOQ", "RDOQTS", "SAO", "AMP", "TransformSkip", "TransformSkipFast"))
            open(cfg, "w").write(common + REF_CFG_INTER["ldp_p"][2] + "BipredSearchRange : 4\nInternalBitDepth : 8\nProfile : main\n")
            times = []
            for frames in (1, 1 + min(2, args.steps)):
                out = subprocess.run([enc_bin, "-c", cfg, "-i", yuv, "-wdt", str(w), "-hgt", str(h), "-fr", "50", "-f", str(frames), "--InputBitDepth=8", "-q", str(qp),
                                      f"--WaveFrontSynchro={args.wpp}", "-b", os.path.join(td, "o.bin"), "-o", os.path.join(td, "r.yuv")], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
                times.append([float(l.split()[2]) for l in out.splitlines() if "Total Time" in l][0])
            line["cpu_baseline"] = {"value": n_ctus * min(2, args.steps) / max(1e-9, times[1] - times[0]), "unit": "CTU/s", "cores": 1, "kind": "reference",
                                    "sample": f"the first {min(2, args.steps)} P pictures of the same stream, same cfg (HM 'Total Time' of 1+n pictures minus the I picture alone)"}
            line["speedup_vs_cpu_1core"] = line["value"] / line["cpu_baseline"]["value"]
    print(json.dumps(line))


def run_dbk(args, torch):
    """Secondary workload: the deblocking filter (hm355_deblock_run) over `--frames` 3840x2160 10-bit pictures whose I-slice
    search results are resident in their slots.  A step = one pass of the filter over the batch (four launches: vertical luma / chroma,
    horizontal luma / chroma edges).  HBM-bound: algorithmic bytes = the picture read once + written once."""
    import hm355
    import synth
    w, h, bd, qp, F = args.width, args.height, 10, args.qp, args.frames
    enc = hm355.Encoder(w, h, bd, 1, F)
    distinct = [synth.frame(w, h, bd, f, 1234) for f in range(min(4, F))]
    for i in range(F):
        enc.upload(i, distinct[i % len(distinct)])
    descs = [(2, qp, None)] * F
    lam, cw = hm355.intra_lambda(qp)
    ms_total = 0.0
    for it in range(args.warmup + args.steps):
        enc.run(F, qp)                                   # untimed: puts fresh pre-deblocking pictures + CU data into the slots
        ms = enc.deblock_run(descs)
        if args.workload == "sao":                       # timed part = hm355_sao_run on the deblocked pictures
            sd = [dict(qp=qp, cabac_init_type=2, depth=0, disabled_rate=np.zeros((3, 8)), chroma_weight=cw, **{"lambda": lam}) for _ in range(F)]
            enc.sao_run(sd)
            k, l = hm355.C.c_double(), hm355.C.c_int()
            enc.lib.hm355_last_run_info(enc.h_, hm355.C.byref(k), hm355.C.byref(l))
            ms = k.value
        if it >= args.warmup:
            ms_total += ms
    n = enc.num_ctus * F * args.steps
    pic_bytes = w * h * 3                               # 4:2:0, 16-bit samples: 1.5 samples x 2 B per luma position
    per_pic = (3 if args.workload == "sao" else 2) * pic_bytes      # deblocking: read + write; SAO: original + deblocked read, output written
    alg = per_pic * F * args.steps
    ach = alg / (ms_total * 1e-3) / 1e9
    is_sao = args.workload == "sao"
    line = {"metric": f"CTUs/sec ({'SAO, encoder side' if is_sao else 'deblocking filter'}) at 4K main10; bit-exact vs HM", "value": n / (ms_total * 1e-3), "unit": "CTU/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_total / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": (f"TEncSampleAdaptiveOffset::SAOProcess on {F} deblocked I pictures {w}x{h} 10-bit (QP {qp}) resident in HBM with their originals"
                                    if is_sao else f"TComLoopFilter::loopFilterPic on {F} I pictures {w}x{h} 10-bit (QP {qp}) resident in HBM with their CU / TU data"),
                       "frames_per_gpu": F, "pictures_per_s": F * args.steps / (ms_total * 1e-3)},
            "roofline": {"bound": "hbm", "kernel": "hm355_sao_stats / decide / apply kernels (3 launches per step)" if is_sao else "hm355_dbk_kernel (4 launches per step)",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "note": (f"algorithmic bytes = {per_pic} B per picture (original and deblocked picture read, output written); the per-CTU decision walks "
                                  "the picture on one lane (CABAC state chains from CTU to CTU) and bounds the step, not HBM") if is_sao else
                                 f"algorithmic bytes = {per_pic} B per picture (read once + written once); the per-CTU decision arrays add 3 KB per CTU"}}
    if not args.no_cpu_baseline:
        import oracle
        # CPU side: the C restatement of the filter (kind "port", 1 core) on one picture with the device's own CU data
        enc.run(1, qp)
        rec1, ctus1, _ = enc.download(0)
        oc = np.zeros(len(ctus1), oracle.CTU_DTYPE)
        for f in oc.dtype.names:
            oc[f] = ctus1[f]
        t0 = time.time()
        dbk1 = oracle.deblock(rec1, bd, qp, 2, np.zeros((2, 16), np.int32), oc, None)
        if is_sao:
            t0 = time.time()
            oracle.sao(distinct[0], dbk1, bd, qp, lam, cw, 2, 0, np.zeros((3, 8)))
        dt = time.time() - t0
        line["cpu_baseline"] = {"value": enc.num_ctus / dt, "unit": "CTU/s", "cores": 1, "kind": "port",
                                "sample": f"one {w}x{h} picture ({enc.num_ctus} CTUs) through oracle/hm_oracle_{'sao' if is_sao else 'dbk'}.inc"}
        line["speedup_vs_cpu_1core"] = line["value"] / line["cpu_baseline"]["value"]
    print(json.dumps(line))
    enc.close()


def run_ingest(args, torch):
    """Secondary workload: picture ingest and output (hm355_upload_file_frames / hm355_download_file_frames): `--frames` 3840x2160
    file frames (10-bit samples in 16-bit words, as main10 sequences are stored) -> the slots' original planes, and the slots' planes -> file
    frames.  A step = one ingest launch + one output launch over the batch, raw frames resident in HBM when the timed region starts (the
    HIP events bracket the kernels only).  Pure HBM streaming: algorithmic bytes = every sample read once and written once per direction."""
    import hm355
    import synth
    w, h, bd, F, fbd = args.width, args.height, 10, args.frames, 10
    enc = hm355.Encoder(w, h, bd, 1, F)
    distinct = [synth.frame(w, h, fbd, f, 1234) for f in range(min(4, F))]
    raws = [b"".join(np.ascontiguousarray(p, "<u2").tobytes() for p in d) for d in distinct]
    frames = [raws[i % len(raws)] for i in range(F)]
    ms_in = ms_out = wall_in = wall_out = 0.0
    for it in range(args.warmup + args.steps):
        t0 = time.time(); a = enc.upload_file_frames(frames, w, h, fbd); t1 = time.time()
        out, b = enc.download_file_frames(F, fbd, 0, 0, source=1); t2 = time.time()
        if it >= args.warmup:
            ms_in += a; ms_out += b; wall_in += t1 - t0; wall_out += t2 - t1
    ok = out[0] == frames[0] and out[F - 1] == frames[F - 1]
    got = enc.download_org(1 % F)
    ok = ok and all(np.array_equal(got[k], distinct[1 % len(distinct)][k]) for k in range(3))
    n = enc.num_ctus * F * args.steps
    per_pic = w * h * 3 // 2 * 2 * 2                    # 16-bit samples read once + written once
    ms_total = ms_in + ms_out
    alg = 2 * per_pic * F * args.steps                  # ingest + output
    ach = alg / (ms_total * 1e-3) / 1e9
    # HBM bytes per step from the committed PMC passes of this workload (profiles/r01e_traffic_ingest.json), if they match its size
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01e_traffic_ingest.json")))
        if (tj["width"], tj["height"], tj["frames"]) == (w, h, F):
            traffic = tj["hbm_bytes_per_step"]
    except (OSError, KeyError, ValueError):
        pass
    line = {"metric": "CTUs/sec (picture ingest + output, TVideoIOYuv read / write) at 4K main10; byte-exact vs HM", "value": n / (ms_total * 1e-3), "unit": "CTU/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_total / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": f"TVideoIOYuv::read + ::write of {F} 4:2:0 frames {w}x{h}, 10-bit samples in 16-bit words, raw frames resident in HBM",
                       "frames_per_gpu": F, "pictures_per_s": F * args.steps / (ms_total * 1e-3), "ingest_ms_per_step": ms_in / args.steps,
                       "output_ms_per_step": ms_out / args.steps, "ingest_ms_per_step_with_pcie": 1e3 * wall_in / args.steps,
                       "output_ms_per_step_with_pcie": 1e3 * wall_out / args.steps, "roundtrip_identical": bool(ok)},
            "roofline": {"bound": "hbm", "kernel": "hm355_ingest_kernel + hm355_output_kernel (1 launch each per step)",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "note": f"algorithmic bytes = {per_pic} B per picture and direction (every 16-bit sample read once + written once) = {2 * per_pic * F} B per step; "
                                 "traffic = PMC FETCH_SIZE (x2, gfx950 correction for 16 B/lane streams) + WRITE_SIZE of both kernels, separate passes"}}
    if not args.no_cpu_baseline:
        import oracle
        t0 = time.time()
        pl = oracle.yuv_read(frames[0], w, h, fbd, bd, 0, 0)
        oracle.yuv_write(pl, bd, fbd, 0, 0)
        dt = time.time() - t0
        line["cpu_baseline"] = {"value": enc.num_ctus / dt, "unit": "CTU/s", "cores": 1, "kind": "port",
                                "sample": f"one {w}x{h} frame read + written through oracle/hm_oracle_yuv.inc"}
        line["speedup_vs_cpu_1core"] = line["value"] / line["cpu_baseline"]["value"]
    print(json.dumps(line))
    enc.close()


def run_bits(args, torch):
    """Secondary workload: the bitstream pass (hm355_encode_slices_run) over `--frames` 3840x2160 10-bit I pictures whose search
    results and SAO parameters are resident in their slots (search, deblocking and SAO run once, untimed).  A step = the CABAC-coded slice
    data of every picture of the batch: 34 substreams per picture (WPP), one wavefront per substream, plus the packing launch.
    The arithmetic coder is a serial dependency chain per substream, so the step is bounded by the longest CTU row, not by HBM."""
    import hm355
    import synth
    w, h, bd, qp, F = args.width, args.height, 10, args.qp, args.frames
    enc = hm355.Encoder(w, h, bd, 1, F)
    distinct = [synth.frame(w, h, bd, f, 1234) for f in range(min(4, F))]
    for i in range(F):
        enc.upload(i, distinct[i % len(distinct)])
    lam, cw = hm355.intra_lambda(qp)
    enc.run(F, qp)
    enc.deblock_run([(2, qp, None)] * F)
    sao = enc.sao_run([dict(qp=qp, cabac_init_type=2, depth=0, disabled_rate=np.zeros((3, 8)), chroma_weight=cw, **{"lambda": lam}) for _ in range(F)])
    descs = [dict(slice_type=2, qp=qp, sao_enabled=(sao[i][0][0], sao[i][0][1])) for i in range(F)]
    ms_total, wall_total, out_bytes, bins = 0.0, 0.0, 0, 0
    k, l = hm355.C.c_double(), hm355.C.c_int()
    for it in range(args.warmup + args.steps):
        t0 = time.time()
        res = enc.encode_slices_run(descs)
        dt = time.time() - t0
        enc.lib.hm355_last_run_info(enc.h_, hm355.C.byref(k), hm355.C.byref(l))
        if it >= args.warmup:
            ms_total += k.value; wall_total += dt
            out_bytes = sum(sum(len(x) for x in r[0]) for r in res); bins = sum(r[2] for r in res)
    n = enc.num_ctus * F * args.steps
    per_ctu = 3072 + 6144 * 4                            # decision arrays + coefficients of a CTU, read once
    alg = (per_ctu * enc.num_ctus * F + 2 * out_bytes) * args.steps      # + the slice data written raw and once more packed
    ach = alg / (ms_total * 1e-3) / 1e9
    line = {"metric": "CTUs/sec (bitstream pass, encodeSlice) at 4K main10; byte-exact slice data vs HM", "value": n / (ms_total * 1e-3), "unit": "CTU/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_total / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"TEncSlice::encodeSlice on {F} I pictures {w}x{h} 10-bit (QP {qp}, WPP, SAO syntax) resident in HBM with their CU / TU data, coefficients and SAO parameters",
                       "frames_per_gpu": F, "pictures_per_s": F * args.steps / (ms_total * 1e-3), "slice_data_bytes_per_step": out_bytes, "bins_per_step": bins,
                       "mbins_per_s": bins * args.steps / (ms_total * 1e-3) / 1e6, "ms_per_step_with_readback": 1e3 * wall_total / args.steps},
            "roofline": {"bound": "hbm", "kernel": "hm355_bits_kernel + hm355_bits_pack_kernel (2 launches per step)",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "note": (f"algorithmic bytes = {per_ctu} B per CTU read (decision arrays + coefficients) + the slice data written twice; the coder is a serial "
                                  "chain per substream (one wavefront, wave-uniform), so the step time is the longest CTU row of the batch, far from the HBM bound")}}
    if not args.no_cpu_baseline:
        import oracle
        _, ctus1, _ = enc.download(0)
        oc = np.zeros(len(ctus1), oracle.CTU_DTYPE)
        for f in oc.dtype.names:
            oc[f] = ctus1[f]
        t0 = time.time()
        want = oracle.encode_slice(w, h, bd, 1, 2, qp, oc, sao=sao[0][1], sao_enabled=descs[0]["sao_enabled"])
        dt = time.time() - t0
        line["parity_checked_in_run"] = bool(want[0] == res[0][0] and want[2] == res[0][2])
        line["cpu_baseline"] = {"value": enc.num_ctus / dt, "unit": "CTU/s", "cores": 1, "kind": "port",
                                "sample": f"one {w}x{h} picture ({enc.num_ctus} CTUs, {sum(len(x) for x in want[0])} bytes) through oracle/hm_oracle_bits.inc"}
        line["speedup_vs_cpu_1core"] = line["value"] / line["cpu_baseline"]["value"]
    print(json.dumps(line))
    enc.close()


def intra_line(args, world, rows_mode, group, steps, warmup, dt, kernel_ms, launches, total_ctus, ctus_per_rank, build_id="unknown"):
    """the JSON line of the intra4k workload (rank 0), without the CPU baseline"""
    # HBM bytes per launch from the PMC passes committed under profiles/ -- quoted only when they were taken on this very build of
    # the library (hm355_build_id) with this launch shape; otherwise null: counters of another build say nothing about this run
    traffic, traffic_note = None, "no PMC passes of this build and launch shape under profiles/"
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r03_traffic.json")))
        if (tj["width"], tj["height"], tj["frames"], tj.get("lanes", 1)) == (args.width, args.height, args.frames, args.lanes) and tj.get("build_id") == build_id:
            traffic, traffic_note = tj["hbm_bytes_per_launch"], "PMC FETCH_SIZE + WRITE_SIZE (separate passes) on this build: profiles/r03_traffic.json"
        else:
            traffic_note = f"profiles/r03_traffic.json is from build {tj.get('build_id')} / another launch shape, this run is build {build_id}"
    except (OSError, KeyError, ValueError):
        pass
    # algorithmic bytes per launch / average launch duration (HIP events on the launch's stream), times the launches in flight on average
    # (= sum of the launch durations / timed region): the GB/s of the CTU-search kernel on this rank over the timed region
    in_flight = (kernel_ms * 1e-3) / dt if not rows_mode else 1.0
    ach = ALG_BYTES_PER_CTU * ctus_per_rank / (kernel_ms * 1e-3) / 1e9 * in_flight
    return {
        "metric": "CTUs/sec (enc) at 4K main10; bit-exact CU partition vs HM",
        "value": total_ctus / dt, "unit": "CTU/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "steps_requested": args.steps, "warmup_requested": args.warmup,
        "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "strong" if rows_mode else "weak", "vs_baseline": None,
        "dtype": "int32+f64", "data": "synthetic",
        "config": {"workload": f"encoder_intra_main10, synthetic {args.width}x{args.height} 10-bit, QP {args.qp}, WaveFrontSynchro=1, " +
                               (f"{args.frames} independent I pictures per step shared by the ranks" if rows_mode else
                                f"{args.frames} independent I pictures per GPU per step, " + ("one launch per step" if args.lanes == 1 else f"steps pipelined {args.lanes} deep")) +
                               f" ({min(DISTINCT_FRAMES, args.frames)} distinct frames), inputs resident in HBM",
                   "frames_per_gpu": args.frames, "ctus_per_step": total_ctus // steps, "lanes": 1 if rows_mode else args.lanes, "build_id": build_id,
                   "parallelism": (f"CTU rows of every picture sharded over {world} GPUs in bands, boundary rows handed down over RCCL send / recv, {group} pictures per pipeline stage"
                                   if rows_mode else f"pictures sharded over {world} GPU(s), 2-CTU-lag wavefront inside a picture"),
                   "step_budget": f"warm-up + timed steps bounded to {args.budget_s:.0f} s of wall time: {warmup}+{steps} of the requested {args.warmup}+{args.steps} steps run"},
        "roofline": {"bound": "hbm", "kernel": "hm355_ctu_kernel", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                     "avg_launch_ms": kernel_ms / max(1, launches), "launches": launches, "launches_in_flight": in_flight,
                     "note": "achieved = 54,278 B/CTU (SURVEY 8d) x CTUs per launch / average launch duration (HIP events on the launch's stream) x "
                             "launches in flight on average; the path is bound by the latency of its own dependent LDS / L2 round trips "
                             "(profiles/r03_pmc_sq_summary.json), not by HBM"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="intra4k", choices=["intra4k", "c2", "c3", "ldp_p", "ra_b", "dbk", "sao", "bits", "ingest"], help="intra4k = the BASELINE.json metric (default)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shard", default="pictures", choices=["pictures", "rows"],
                    help="N > 1: 'pictures' = every rank its own pictures (weak scaling, no data-path transfer; default); 'rows' = every rank a band of CTU rows "
                         "of the SAME pictures, boundary rows handed down over RCCL send / recv (hm-16.2_amd/bands.py; strong scaling)")
    ap.add_argument("--group", type=int, default=0, help="--shard rows: pictures per pipeline stage (default: frames / (4 * ranks))")
    ap.add_argument("--budget-s", type=float, default=280.0, help="wall-time bound of warm-up + timed steps (intra4k); fewer steps run when the request does not fit")
    ap.add_argument("--frames", type=int, default=768, help="independent pictures per GPU per step")
    ap.add_argument("--lanes", type=int, default=1, help="steps in flight at a time (hm355_run_begin / hm355_run_wait; 1..4)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--wpp", type=int, default=0, help="--workload c3: WaveFrontSynchro (the cfg has 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if "RANK" in os.environ or args.gpus < 1:
            raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
        # started without a launcher: one rank per GPU as child processes of torch.distributed.run (nothing here has touched the GPU yet)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", os.environ.get("MASTER_PORT", "29511"), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    pre_frames = None
    rows_mode = args.shard == "rows" and world > 1
    if args.workload == "intra4k":      # the frame pool forks: before anything initialises the GPU
        pre_frames = distinct_frames(args.width, args.height, 10, rank_frame_numbers(0 if rows_mode else rank, min(DISTINCT_FRAMES, args.frames)), 1234)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: hm355 has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if args.workload != "intra4k":
        if world > 1 and args.workload not in ("ldp_p", "ra_b"):
            raise SystemExit("--workload c2 / c3 / dbk / sao / bits / ingest are single-GPU measurements")
        if args.workload in ("c2", "c3"):
            return run_as_stated(args, torch)
        if args.workload == "bits":
            return run_bits(args, torch)
        if args.workload == "ingest":
            return run_ingest(args, torch)
        return run_dbk(args, torch) if args.workload in ("dbk", "sao") else run_inter(args, torch, world, rank, local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import hm355
    bd, seed = 10, 1234
    lanes = 1 if rows_mode else max(1, min(4, args.lanes))
    args.lanes = lanes
    enc = hm355.Encoder(args.width, args.height, bd, 1, args.frames * lanes)
    build_id = enc.lib.hm355_build_id().decode()
    if lanes > 1:
        enc.set_lane_share(lanes)
    # synthetic clip: 16 distinct frames per rank (rank r takes frames 16r .. 16r+15), cycled over the picture slots; resident in HBM before timing
    distinct = pre_frames
    for i in range(args.frames * lanes):
        enc.upload(i, distinct[i % len(distinct)])
    del distinct

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def agree(v, op):
        return agree_over_ranks(dist, torch, v, op, "cuda")

    if rows_mode:
        # every rank holds the same pictures and searches its band of CTU rows; the last row of a band goes down to the next rank (RCCL send / recv)
        import bands
        h_ctu = (args.height + 63) // 64
        transport = bands.TorchTransport(dist, torch, torch.device("cuda", local_rank))
        group = args.group or max(1, args.frames // (4 * world))

        def run_steps(count):
            ms_sum, launches = 0.0, 0
            for _ in range(count):
                ms_sum += bands.run_banded(enc, args.frames, group, h_ctu, rank, world, transport.send, transport.recv, args.qp, transport.buffer)
                launches += len(bands.picture_groups(args.frames, group))
            return ms_sum, launches
    else:
        def run_steps(count):
            """`count` steps, pipelined `lanes` deep: step k runs on lane k % lanes over that lane's own slots; returns when the last
            step has finished (the pipeline is empty again), with the summed per-launch kernel times (HIP events) and the launch count"""
            ms_sum, busy = 0.0, [False] * lanes
            for k in range(count):
                lane = k % lanes
                if busy[lane]:
                    ms_sum += enc.run_wait(lane)
                enc.run_begin(lane, lane * args.frames, args.frames, args.qp)
                busy[lane] = True
            for lane in range(lanes):          # oldest launch first
                l2 = (count + lane) % lanes
                if busy[l2]:
                    ms_sum += enc.run_wait(l2)
            return ms_sum, count
    # Warm-up.  Its first `lanes` steps are timed: when the requested warm-up + timed steps would not fit the wall budget (the driver's
    # 600 s limit covers start-up, the steps and the CPU baseline), the warm-up stops there and as many timed steps run as fit (at least
    # one pipeline depth); the line reports the steps actually run next to the ones requested.
    first = min(max(1, args.warmup), lanes)
    t0 = time.perf_counter()
    run_steps(first)
    barrier()
    step_s = agree((time.perf_counter() - t0) / first, dist.ReduceOp.MAX if dist else None)
    warmup, steps = plan_steps(step_s, args.warmup, args.steps, args.budget_s, first)
    if warmup > first:
        run_steps(warmup - first)
    barrier()
    t0 = time.perf_counter()
    kernel_ms, launches = run_steps(steps)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        dt = agree(dt, dist.ReduceOp.MAX)
    if rows_mode:      # the ranks share the pictures: the job is args.frames pictures per step whatever the number of ranks
        first_row, last_row = bands.band_rows(h_ctu, world, rank)
        ctus_per_rank = ((args.width + 63) // 64) * max(0, last_row - first_row + 1) * args.frames * steps
        total_ctus = enc.num_ctus * args.frames * steps
    else:
        ctus_per_rank = enc.num_ctus * args.frames * steps
        total_ctus = ctus_per_rank * world
    if rank == 0:
        line = intra_line(args, world, rows_mode, group if rows_mode else 0, steps, warmup, dt, kernel_ms, launches, total_ctus, ctus_per_rank, build_id)
        if not args.no_cpu_baseline and world >= 1:
            line["cpu_baseline"] = cpu_baseline(args.width, bd, args.qp, seed)
            line["speedup_vs_cpu_1core"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    enc.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
