"""CPU suite: host-side logic of the product (schedule, ABI surface, library exports)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import common

ROOT = common.ROOT


def test_library_exports_every_declared_symbol():
    """libhm355.so must load without a GPU and export every function include/hm355.h declares"""
    import hm355
    hdr = open(os.path.join(ROOT, "include", "hm355.h")).read()
    declared = sorted(set(re.findall(r"\b(hm355_[a-z_0-9]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    lib = ctypes.CDLL(hm355.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/hm355.h but not exported"
    assert set(hm355.EXPORTS) == set(declared)


def test_create_rejects_bad_config_and_missing_gpu():
    """argument errors are reported as HM355_ERR_ARG; without a device create() must fail (no CPU fallback)"""
    import hm355
    lib = hm355.load_library()
    h = ctypes.c_void_p()
    bad = hm355.SeqCfg(417, 240, 8, 64, 4, 5, 2, 3, 0, 1)
    assert lib.hm355_create(ctypes.byref(bad), ctypes.byref(h)) == -1
    bad = hm355.SeqCfg(416, 240, 9, 64, 4, 5, 2, 3, 0, 1)
    assert lib.hm355_create(ctypes.byref(bad), ctypes.byref(h)) == -1
    import torch
    if not torch.cuda.is_available():
        ok = hm355.SeqCfg(416, 240, 8, 64, 4, 5, 2, 3, 0, 1)
        rc = lib.hm355_create(ctypes.byref(ok), ctypes.byref(h))
        assert rc == -2, "hm355_create must fail with HM355_ERR_NO_DEVICE when there is no GPU"


def _schedule(wc, hc, wpp, frames, carry=0):
    """python mirror of hm355_build_schedule (hm355_host_common.h) for property checks"""
    def step(x, y):
        if not wpp:
            return y * wc + x
        if carry and hc > 1 and wc > 1 and y == hc - 1:
            return (wc - 1) + 2 * (hc - 2) + 1 + x
        return x + 2 * y
    out = [[] for _ in range(step(wc - 1, hc - 1) + 1)]
    for f in range(frames):
        for y in range(hc):
            for x in range(wc):
                out[step(x, y)].append((f, x, y))
    return out


@pytest.mark.parametrize("wc,hc,wpp,carry", [(7, 4, 1, 0), (7, 4, 0, 0), (60, 34, 1, 0), (1, 3, 1, 0), (2, 2, 1, 0), (7, 4, 1, 1), (30, 17, 1, 1),
                                             (1, 3, 1, 1), (2, 2, 1, 1)])
def test_schedule_respects_dependencies(wc, hc, wpp, carry):
    sched = _schedule(wc, hc, wpp, 2, carry)
    done = {}
    for s, items in enumerate(sched):
        for (f, x, y) in items:
            deps = [(x - 1, y), (x, y - 1), (x - 1, y - 1), (x + 1, y - 1)]
            if not wpp and (x, y) != (0, 0):
                px, py = (x - 1, y) if x > 0 else (wc - 1, y - 1)
                deps.append((px, py))
            if wpp and x == 0 and y > 0 and wc > 1:
                deps.append((1, y - 1))
            if carry and x == 0 and y == hc - 1 and y > 0:      # P slice, last CTU row cut by the picture edge
                deps.append((wc - 1, y - 1))
            for (dx, dy) in deps:
                if 0 <= dx < wc and 0 <= dy < hc:
                    assert done.get((f, dx, dy), 10 ** 9) < s, f"CTU {(x, y)} scheduled before {(dx, dy)}"
        for it in items:
            done[it] = s
    assert len(done) == 2 * wc * hc


def test_hostsim_of_kernel_source_matches_reference_fixture(tmp_path):
    """The kernel source (hm355_core.h) compiled for the host with one lane, forwards and with every
    lane-parallel loop reversed, reproduces the reference fixture.  Debugging aid: the GPU tests are the gate."""
    import synth
    cfg, frames = common.load_case("small_128x128_10b_qp37")
    yuv = tmp_path / "in.yuv"
    synth.write_yuv(str(yuv), cfg["width"], cfg["height"], cfg["bit_depth"], cfg["frames"], cfg["seed"])
    import gen_golden
    for flag, exe in (("", "hostsim"), ("-DHM355_HOSTSIM_REVERSE", "hostsim_rev")):
        out = tmp_path / exe
        cmd = ["g++", "-O2", "-std=c++14", "-ffp-contract=off", "-w"] + ([flag] if flag else []) + ["-o", str(out), os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp")]
        subprocess.run(cmd, check=True)
        dump = tmp_path / (exe + ".bin")
        subprocess.run([str(out), str(yuv), str(cfg["width"]), str(cfg["height"]), str(cfg["bit_depth"]), str(cfg["frames"]),
                        str(cfg["qp"]), str(cfg["wpp"]), str(dump)], check=True)
        got = gen_golden.parse_dump(str(dump))
        for i, (ctus, rec) in enumerate(frames):
            common.assert_ctus_equal(got[i][0], ctus, f"{exe} frame {i}", (cfg["width"], cfg["height"]))
            assert np.array_equal(got[i][1], rec)


def test_hostsim_result_does_not_depend_on_what_the_buffers_held(tmp_path):
    """Everything the search writes before it reads may hold anything: the host twin with its workspace, LDS state, reconstruction planes, decision /
    coefficient / statistics arrays and CABAC hand-off states filled with noise (HM355_DIRTY, three seeds) gives the result of the zero-filled run."""
    import filecmp
    import synth
    out = tmp_path / "hostsim"
    subprocess.run(["g++", "-O2", "-std=c++14", "-ffp-contract=off", "-w", "-o", str(out), os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp")], check=True)
    for (w, h, bd, qp, wpp, seed) in [(192, 128, 10, 30, 1, 31), (200, 136, 8, 22, 0, 9)]:
        yuv = tmp_path / f"in_{w}.yuv"
        synth.write_yuv(str(yuv), w, h, bd, 2, seed)
        dumps = []
        for d in (None, "1", "2", "3"):
            dump = tmp_path / f"d_{w}_{d}.bin"
            env = dict(os.environ)
            if d:
                env["HM355_DIRTY"] = d
            subprocess.run([str(out), str(yuv), str(w), str(h), str(bd), "2", str(qp), str(wpp), str(dump)], check=True, env=env)
            dumps.append(str(dump))
        assert all(filecmp.cmp(dumps[0], x, shallow=False) for x in dumps[1:]), f"{w}x{h}: the result depends on uninitialised memory"


def test_hostsim_of_kernel_source_matches_oracle_at_extreme_qps(tmp_path):
    """Dense 32x32 blocks (low QP: RDOQ with every coefficient group coded, sign-bit hiding everywhere), nearly empty ones (high QP),
    with and without WPP: the code paths behind the early terminations of the CU / residual quadtrees and the LDS-resident RDOQ state
    of 32x32 blocks, against the oracle on fresh inputs."""
    import synth, gen_golden, oracle
    out = tmp_path / "hostsim"
    subprocess.run(["g++", "-O2", "-std=c++14", "-ffp-contract=off", "-w", "-o", str(out), os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp")], check=True)
    for (w, h, bd, qp, wpp, seed) in [(320, 192, 10, 12, 1, 21), (256, 136, 8, 47, 0, 22), (384, 192, 10, 22, 1, 23), (200, 192, 8, 3, 0, 24)]:
        yuv = tmp_path / f"in{seed}.yuv"
        synth.write_yuv(str(yuv), w, h, bd, 1, seed)
        want_rec, want_ctus = oracle.compress(synth.frame(w, h, bd, 0, seed), bd, qp, wpp)
        dump = tmp_path / f"out{seed}.bin"
        subprocess.run([str(out), str(yuv), str(w), str(h), str(bd), "1", str(qp), str(wpp), str(dump)], check=True)
        got = gen_golden.parse_dump(str(dump))
        common.assert_ctus_equal(got[0][0], want_ctus, f"{w}x{h} qp{qp}", (w, h))
        common.assert_rec_equal(want_rec, got[0][1], w, h, f"{w}x{h} qp{qp}")


@pytest.mark.parametrize("name", common.LDP_CASES[1:] + common.B_CASES + common.LDP_LONG_CASES)
def test_hostsim_of_kernel_source_matches_reference_p_slices(tmp_path, name):
    """The P-slice part of the kernel source (hm355_inter.h / hm355_inter_cu.h) compiled for the host with one lane, forwards and
    with every lane-parallel loop reversed: self-checking replay of the reference's HMD2 record stream (rebuilt from the fixture),
    the search of every inter slice and the bitstream pass of every picture."""
    import synth
    import hmd2
    recs = []
    cfg, _, _ = common.load_ldp_case(name, recs)
    yuv, dump = tmp_path / "in.yuv", tmp_path / "dump2.bin"
    synth.write_yuv(str(yuv), cfg["width"], cfg["height"], cfg["bit_depth"], cfg["frames"], cfg["seed"])
    hmd2.write(str(dump), recs, bits=True)
    for flag, exe in (("", "hostsim_inter"), ("-DHM355_HOSTSIM_REVERSE", "hostsim_inter_rev")):
        out = tmp_path / exe
        cmd = ["g++", "-O2", "-std=c++14", "-ffp-contract=off", "-w"] + ([flag] if flag else []) + ["-o", str(out), os.path.join(ROOT, "tests", "hostsim", "hostsim_inter.cpp")]
        subprocess.run(cmd, check=True)
        r = subprocess.run([str(out), str(yuv), str(dump), str(cfg["width"]), str(cfg["height"]), str(cfg["bit_depth"]), str(cfg["wpp"])],
                           capture_output=True, text=True)
        assert r.returncode == 0 and "all bit-exact" in r.stdout, r.stdout[-2000:]
        assert r.stdout.count("bitstream: ok") == cfg["frames"], r.stdout[-2000:]     # the bitstream pass (hm355_bits_kernel.h) of every picture, I slice included
