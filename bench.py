#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the trajectory step (BASELINE.json metric).

One "step" = one pass of the particle loop (reference timemanager.f90:531-712)
over every particle resident on the GPU(s): one k_advance launch per GPU.
Default workload (N=1): BASELINE config 3 -- 1e8 particles on the synthetic
361x181x138 ECMWF-shaped grid, Hanna turbulence + CBL scheme, counter RNG, fp64.
With --gpus N>1 the ranks (one per GPU; launched by torch.distributed.run, or by bench.py itself
when WORLD_SIZE is not set -- a launcher hop before anything touches the GPU) share ONE cloud of
--particles particles (BASELINE configs 3-5: 1e8): rank g owns
the contiguous range [g*P/N, (g+1)*P/N) of particle numbers (the reference's MPI
layout, README_PARALLEL.md:60-67) -- strong scaling, no data-path collective
(particles are independent); the counter RNG is keyed on the
global particle number, so the result does not depend on N.  --weak keeps the
per-GPU count fixed instead.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, choices=(2, 3, 4, 5))
    ap.add_argument("--particles", type=float, default=None, help="particles of the whole job (default: 1e7 config 2, 1e8 configs 3-5), shared by the ranks")
    ap.add_argument("--weak", action="store_true", help="weak scaling: --particles per GPU instead of in total")
    ap.add_argument("--real", type=int, default=8, choices=(4, 8), help="compute real bytes")
    ap.add_argument("--rng", default="philox", choices=("philox", "table_counter"))
    ap.add_argument("--sort-interval", type=int, default=4, help="locality re-sort every k steps (0: never); 4 puts one re-sort inside the default timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter pass that measures the dominant kernel's VALU issue time")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the short run the counter pass profiles
    ap.add_argument("--cpu-sample", type=int, default=None, help="particles in the CPU baseline sample")
    ap.add_argument("--poles", action="store_true", help="the full-globe variant (SURVEY 8d): nglobal = sglobal = 1 (gridcheck_ecmwf.f90:341-366), particles up to "
                    "|lat| < 89 deg; beyond +-75 deg they move on the polar stereographic maps (advance.f90:161-164,754-778)")
    ap.add_argument("--without", default="", help="config 5 only, for attributing its kernel time: comma-separated parts to leave out (nest, aerosol, settling, drydep, wet)")
    ap.add_argument("--blend", default="auto", choices=("auto", "on", "off"), help="time-blended wind packs (fpx_config.blend_mode); auto: from the run's particle count over all ranks")
    ap.add_argument("--global-particles", type=float, default=None, help="particle count of the run this job is a shard of (default: --particles); "
                    "e.g. --particles 12500000 --global-particles 1e8 = the shard one of eight GPUs runs, with that run's decisions")
    ap.add_argument("--slice-passes", type=int, default=0, help="fpx_config.pbl_slice_passes: 0 the engine's schedule, -1 one launch, k passes per launch")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="fpx_set_option knob (repeatable), e.g. --opt pbl_slices=48,96,0")
    return ap.parse_args()


def build_scenario(cfg, nsteps, poles=False, without=()):
    from flexpart_amd import synthetic as syn
    if cfg == 2:
        sc = syn.base_scenario(ctl=-5.0, ifine=4, turb_off=True, hmix_const=syn.HMIXMIN, nsteps=nsteps, polar=poles)
        frac_pbl = 0.0
    else:
        sc = syn.base_scenario(ctl=5.0, ifine=4, cblflag=1, nsteps=nsteps, polar=poles)
        frac_pbl = 0.5
        if cfg == 4:
            # config 4: + conccalc into a global 360x180x10 output grid every step, grids summed
            # over the ranks with one RCCL all-reduce at the end of the timed region
            sc["npart"] = 1
            sc["itramem"] = np.zeros(1, np.int32)
            sc["itime0"] = 0
            syn.add_outgrid(sc, nxg=360, nyg=180, nzg=10, outlon0=-180.0, outlat0=-90.0, dxout=1.0, dyout=1.0,
                            ind_samp=-1, old_fraction=0.0)
            del sc["npart"], sc["itramem"]
        if cfg == 5:
            # config 5: one aerosol species (settling, dry deposition, decay), a 2x nest over the middle of
            # the domain (interpol_*_nests), wet deposition (mother + nest precipitation/cloud fields) and
            # the output grid every step; meant for --real 4 (fp32 mixed: positions stay fp64)
            if "aerosol" not in without:
                sc.update(lsettling=1, drydep=1, drydepspec=np.array([1], np.int32), density=np.array([2000.0]),
                          dquer=np.array([8.0]), vsetaver=np.array([-0.004]), cunningham=np.array([1.02]),
                          decay=np.array([1.0e-6]), xmass=np.array([1.0]))
                if "settling" in without:
                    sc.update(lsettling=0)
                if "drydep" in without:
                    sc.update(drydep=0, drydepspec=np.array([0], np.int32))
            sc["npart"] = 1
            sc["itramem"] = np.zeros(1, np.int32)
            sc["itime0"] = 0
            syn.add_outgrid(sc, nxg=360, nyg=180, nzg=10, outlon0=-180.0, outlat0=-90.0, dxout=1.0, dyout=1.0,
                            ind_samp=-1, old_fraction=0.0)
            del sc["npart"], sc["itramem"]
            if "nest" not in without:
                syn.add_nest(sc, ix0=120, jy0=60, ix1=240, jy1=120, factor=2)
            if "wet" not in without:
                syn.add_wet(sc, gas="aerosol" in without)
                if "nest" not in without:
                    syn.add_wet_nest(sc)
    return sc, frac_pbl


def measured_traffic(config, nper, kernel):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary
    (profiles/r*/c<config>_<particles>_pmc.json, made by tools/collect_profile.sh +
    tools/summarize_profile.py on the same command line); None if no matching profile."""
    import glob
    tag = f"c{config}_{nper:.0e}"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"{tag}_pmc.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        # several template variants may match (k_prep with/without initialize()): the steady-state one ran most often
        best = max((e for name, e in d["kernels"].items() if kernel in name), key=lambda e: e["launches_profiled"], default=None)
        if best is not None:
            return best["hbm_bytes_per_launch"]
    except Exception:
        return None
    return None


N_SIMD = 256 * 4        # MI355X: 256 CUs x 4 SIMDs


def relaunch_argv(ngpus, argv, port):
    """The command `python bench.py --gpus N ...` re-launches itself as when it was not started by a launcher
    (WORLD_SIZE unset): the driver's own form, one rank per GPU of this node."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def child_bench_cmd(args):
    return ([sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--config", str(args.config), "--real", str(args.real),
            "--rng", args.rng, "--sort-interval", str(args.sort_interval), "--steps", "3", "--warmup", "1",
            "--no-cpu-baseline", "--no-pmc", "--blend", args.blend, "--slice-passes", str(args.slice_passes)] + (["--poles"] if args.poles else [])
            + (["--particles", repr(args.particles)] if args.particles else [])
            + (["--global-particles", repr(args.global_particles)] if args.global_particles else [])
            + (["--without", args.without] if args.without else [])
            + [a for o in args.opt for a in ("--opt", o)])


def step_groups(rows, group):
    """The dispatches of one kernel in dispatch order, `group` consecutive ones (the launches of one step: the Langevin
    kernel runs in time slices) summed into one entry."""
    rows = sorted(rows, key=lambda e: e["id"])
    out = []
    for i in range(0, len(rows) - group + 1, group):
        g = rows[i:i + group]
        out.append({k: sum(e.get(k, 0.0) for e in g) for k in g[0] if k != "id"})
    return out


def live_traffic(args, kernel, group=1):
    """HBM bytes per launch of the dominant kernel, measured in THIS run: two child `rocprofv3 --pmc` passes over the same
    command, FETCH_SIZE and WRITE_SIZE in a pass each (MI355X_MICROARCH.md, HBM section: both count KiB, separate passes;
    on gfx950 FETCH_SIZE reports half the bytes of wide reads, so it is doubled; the raw counters are kept next to it)."""
    import csv, glob, shutil, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None, {"error": "rocprofv3 not found"}
    raw = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="fpx_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
        try:
            cmd = ["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "--"] + child_bench_cmd(args)
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, {"error": f"{ctr} pass failed (rc {r.returncode}): {r.stderr[-300:]}"}
            rows = [{"id": int(row["Dispatch_Id"]), "v": float(row["Counter_Value"])} for row in csv.DictReader(open(files[0]))
                    if kernel + "<" in row["Kernel_Name"] and row["Counter_Name"] == ctr]
            vals = [g["v"] for g in step_groups(rows, group)]
            if not vals:
                return None, {"error": f"no dispatch of {kernel} in the {ctr} pass"}
            vals.sort()
            raw[ctr + "_KiB"] = vals[len(vals) // 2 if len(vals) > 2 else 0]      # a steady-state launch
        except Exception as ex:
            return None, {"error": f"{type(ex).__name__}: {ex}"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return (2.0 * raw["FETCH_SIZE_KiB"] + raw["WRITE_SIZE_KiB"]) * 1024.0, raw


def live_valu(args, kernel, avg_ms, launch_work, group=1):
    """VALU issue time of the dominant kernel, measured in THIS run: a child `rocprofv3 --pmc` pass over the same
    command (same workload, seeds and sizes; one warm-up and two timed steps) counts the wave-instructions the kernel
    executes (SQ_INSTS_VALU), the quad-cycles its VALUs are busy issuing them (SQ_ACTIVE_INST_VALU) and the shader clock
    (GRBM_GUI_ACTIVE / 8 XCDs / kernel duration); issue time = busy cycles / (SIMDs x clock), and its ratio to the
    kernel's launch duration of the un-profiled timed region above says how far the kernel is from VALU-issue-bound.
    Counters only (no trace domains), in a child process, with the program itself after `--`."""
    import csv, glob, shutil, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return {"error": "rocprofv3 not found"}
    d = tempfile.mkdtemp(prefix="fpx_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    cmd = ["rocprofv3", "--pmc", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
           "GRBM_GUI_ACTIVE", "--output-format", "csv", "-d", d, "--"] + child_bench_cmd(args)
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return {"error": f"counter pass failed (rc {r.returncode}): {r.stderr[-300:]}"}
        per = {}
        for row in csv.DictReader(open(files[0])):
            if kernel + "<" not in row["Kernel_Name"]:
                continue
            e = per.setdefault(row["Dispatch_Id"], {"id": int(row["Dispatch_Id"]), "ns": float(row["End_Timestamp"]) - float(row["Start_Timestamp"])})
            e[row["Counter_Name"]] = float(row["Counter_Value"])
        disp = step_groups([e for e in per.values() if "SQ_INSTS_VALU" in e], group)   # per step: all time slices of the kernel together
        if not disp:
            return {"error": f"no dispatch of {kernel} in the counter pass"}
        disp.sort(key=lambda e: e["SQ_INSTS_VALU"])
        e = disp[len(disp) // 2 if len(disp) > 2 else 0]          # a steady-state launch: the median (the first launch also runs initialize())
        clk_ghz = e["GRBM_GUI_ACTIVE"] / 8.0 / e["ns"] if e.get("GRBM_GUI_ACTIVE") else None
        busy_cycles = 4.0 * e["SQ_ACTIVE_INST_VALU"]              # the SQ counters tick in quad-cycles
        issue_ms = busy_cycles / N_SIMD / ((clk_ghz or 2.4) * 1e9) * 1e3
        # SQ_ACTIVE_INST_VALU charges every VALU instruction at least one quad-cycle; a 32-bit instruction issues in less
        # (2.4 cycles measured, tools/valu_rates.hip), so for the f32 engine the "busy" time exceeds the launch and is no
        # utilisation: reported as None there (the fp64 mix is 4.3 cycles per instruction, at the counter's granularity)
        frac = issue_ms / avg_ms
        return {"insts_valu_per_launch": e["SQ_INSTS_VALU"], "valu_busy_cycles_per_launch": busy_cycles,
                "cycles_per_valu_inst": busy_cycles / e["SQ_INSTS_VALU"],
                "lane_utilisation": e["SQ_THREAD_CYCLES_VALU"] / (64.0 * e["SQ_ACTIVE_INST_VALU"]) if e.get("SQ_THREAD_CYCLES_VALU") else None,
                "clock_ghz_under_pmc": clk_ghz, "kernel_ms_under_pmc": e["ns"] * 1e-6,
                "issue_ms": issue_ms, "frac_of_launch": frac if frac <= 1.0 else None,
                "quad_cycle_time_over_launch": frac,
                "insts_valu_per_particle_step": e["SQ_INSTS_VALU"] / max(launch_work, 1.0),   # wave-instructions per particle-step of the launch
                "source": "rocprofv3 --pmc pass of this run (child process, same workload)"}
    except Exception as ex:          # the counter pass must not sink the GPU number
        return {"error": f"{type(ex).__name__}: {ex}"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


# cycles per wave64 instruction per SIMD, measured on this part with tools/valu_rates.hip (profiles/r4/valu_rates.txt)
VALU_CLASS_CYCLES = {"ADD_F32": 2.4, "MUL_F32": 2.4, "FMA_F32": 3.6, "TRANS_F32": 8.2, "INT32": 2.4, "INT64": 4.3, "CVT": 4.2,
                     "ADD_F64": 4.5, "MUL_F64": 4.5, "FMA_F64": 4.5, "TRANS_F64": 16.2}
VALU_OTHER_CYCLES = 2.4     # moves, compares, selects, cross-lane: the cheapest class (an underestimate for 64-bit moves and VOP3 forms)


def live_valu_classes(args, kernel, avg_ms, valu, group=1):
    """VALU issue time of the dominant kernel from the per-class instruction counters (SQ_INSTS_VALU_ADD_F32 ... TRANS_F64) of
    two more counter passes over the same command, each class priced at its measured issue cost.  Unlike the quad-cycle
    counter SQ_ACTIVE_INST_VALU -- which charges every instruction at least four cycles and so exceeds the launch for a
    kernel of 2.4-cycle f32 instructions -- this gives a distance-to-limit figure for the f32 engine too."""
    import csv, glob, shutil, subprocess, tempfile
    if shutil.which("rocprofv3") is None or "error" in valu:
        return {"error": "no counter pass"}
    sets = [["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32"],
            ["SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"]]
    counts = {}
    for ctrs in sets:
        d = tempfile.mkdtemp(prefix="fpx_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
        try:
            cmd = ["rocprofv3", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "--"] + child_bench_cmd(args)
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"error": f"class counter pass failed (rc {r.returncode}): {r.stderr[-300:]}"}
            per = {}
            for row in csv.DictReader(open(files[0])):
                if kernel + "<" in row["Kernel_Name"]:
                    e = per.setdefault(row["Dispatch_Id"], {"id": int(row["Dispatch_Id"])})
                    e[row["Counter_Name"]] = float(row["Counter_Value"])
            disp = step_groups(list(per.values()), group)
            if not disp:
                return {"error": f"no dispatch of {kernel} in the class counter pass"}
            key = ctrs[0]
            disp.sort(key=lambda e: e.get(key, 0.0))
            counts.update({k: v for k, v in disp[len(disp) // 2 if len(disp) > 2 else 0].items()})
        except Exception as ex:
            return {"error": f"{type(ex).__name__}: {ex}"}
        finally:
            shutil.rmtree(d, ignore_errors=True)
    total = counts.get("SQ_INSTS_VALU", 0.0)
    by_class = {k: counts.get("SQ_INSTS_VALU_" + k, 0.0) for k in VALU_CLASS_CYCLES}
    other = max(total - sum(by_class.values()), 0.0)
    cycles = sum(by_class[k] * c for k, c in VALU_CLASS_CYCLES.items()) + other * VALU_OTHER_CYCLES
    clk = (valu.get("clock_ghz_under_pmc") or 2.4) * 1e9
    issue_ms = cycles / N_SIMD / clk * 1e3
    return {"insts_by_class_per_launch": dict(by_class, OTHER=other), "cycles_per_class": dict(VALU_CLASS_CYCLES, OTHER=VALU_OTHER_CYCLES),
            "issue_ms": issue_ms, "frac_of_launch": issue_ms / avg_ms,
            "note": "wave-instructions of each class x its measured issue cost (tools/valu_rates.hip), summed over the SIMDs; unclassified instructions at the cheapest rate"}


def cpu_baseline(args, sc, frac_pbl):
    """The reference itself (oracle/_ref, flang build of the unmodified Fortran) timed on this
    box's host cores on a bounded sample of the same workload; 1 core (the reference hot path
    is serial: no OpenMP, no MPI here).  Falls back to the C oracle when _ref is absent."""
    from flexpart_amd import synthetic as syn
    from oracle import scenario_io as sio
    n = args.cpu_sample or (2_000_000 if args.config == 2 else 200_000)     # about 10-30 s of CPU work either way
    s2 = dict(sc)
    s2["nsteps"] = 2
    nx, ny, nz = (int(v) for v in sc["grid"])
    s2.update(syn.make_particles(n, nx, ny, sc["height"], sc["hmix"], seed=0x5EED, frac_pbl=frac_pbl))
    kind = "r8" if args.real == 8 else "r4"
    if "nest" in sc:
        kind += "n"     # the reference built from its own par_mod_meteoswiss.f90 (maxnests = 1): oracle/build_ref.sh
    if sio.have_ref(kind):
        out = sio.run_reference(s2, kind, workdir=os.environ.get("TMPDIR", "/tmp"), timing=True, tag="bench")
        tsec, nadv = float(out["timing"][0]), float(out["timing"][1])
        which = "reference"
    else:
        from oracle.oracle import Oracle
        o = Oracle(s2, kind[:2])
        t0 = time.time()
        nadv = 0
        for _ in range(2):
            nadv += o.step()
        tsec = time.time() - t0
        which = "port"
    return {"value": nadv / tsec, "unit": "particle-steps/s", "cores": 1, "kind": which,
            "sample": f"{n} particles x 2 steps of the same scenario ({tsec:.1f} s of CPU time, "
                      f"first step includes initialize())"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start the N ranks ourselves, as a child process, before torch / HIP is touched here
        import subprocess
        raise SystemExit(subprocess.run(relaunch_argv(args.gpus, sys.argv[1:], free_port())).returncode)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE); they must agree")
    import torch
    dist = None
    # Rehearsal on a box with fewer GPUs than ranks (our own 1-GPU checks of the N > 1 code path): the ranks
    # share the devices and talk over gloo; no RCCL communicator is made.  The driver's runs have one GPU per rank.
    ndev = torch.cuda.device_count()
    rehearsal = world > 1 and ndev < world
    if rehearsal:
        local = local % max(ndev, 1)
    tdev = "cpu" if rehearsal else "cuda"
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(local)

    from flexpart_amd.engine import Engine, RNG_PHILOX, RNG_TABLE_COUNTER
    from flexpart_amd import sharding
    ntot = int(args.particles or (1e7 if args.config == 2 else 1e8))
    if args.weak:
        ntot *= world
    lo, hi = sharding.shard_bounds(ntot, world, rank)      # this rank's range of the global particle numbers
    nper = hi - lo
    total_steps = args.warmup + args.steps
    sc, frac_pbl = build_scenario(args.config, total_steps, args.poles, tuple(w for w in args.without.split(",") if w))
    sc["npart_rel"] = np.array([ntot], np.int32)           # npart(1): particles of the release on ALL ranks
    rng = RNG_PHILOX if args.rng == "philox" else RNG_TABLE_COUNTER
    eng = Engine(sc, compute_real_bytes=args.real, host_real_bytes=args.real, rng_mode=rng,
                 seed=0x5EED, max_particles=nper, device=local, sort_interval=args.sort_interval, particle_base=lo,
                 blend_mode={"auto": 0, "on": 1, "off": 2}[args.blend], global_particles=int(args.global_particles or ntot),
                 pbl_slice_passes=args.slice_passes, options=dict(o.split("=", 1) for o in args.opt))
    eng.seed_particles(nper, seed=0x5EED, frac_pbl=frac_pbl, lat_margin_cells=1.0 if args.poles else None)   # slice [lo, hi) of the one global synthetic cloud
    if args.sort_interval > 0:
        eng.sort()          # a release normally arrives ordered; the synthetic cloud is random
    transport = None
    if world > 1:
        # every multi-rank run has the communicator of the reference's MPI build: it carries the particle count the root
        # reduces at every output (timemanager_mpi.f90:552-562) and, in configs 4-5, the grid sums (mpi_mod.f90:2451-2492)
        if rehearsal:       # ranks share a device: RCCL refuses that, the host transport (gloo) carries the reduction
            eng.comm_init_host(dist, world, rank)
            transport = "host callback over gloo (rehearsal: ranks share a GPU)"
        else:
            uid = sharding.share_unique_id(dist, eng.comm_unique_id)
            eng.comm_init(uid, world, rank)
            transport = "rccl"
    lsync = int(sc["lsynctime"])
    window = 10800

    def do_step(i):
        # stationary synthetic met: when the clock leaves the wind window, move the window
        # (what getfields() does with new data; here the same two slots stay resident)
        itime = i * lsync
        w0 = (itime // window) * window
        eng.set_windtime((w0, w0 + window), (1, 2))
        if args.config == 5 and itime != 0 and eng.has_wet:
            eng.wetdepo(itime, lsync, 3600)            # timemanager.f90:164-169: before the particle loop
        eng.step_async(itime)
        if args.config in (4, 5):
            eng.conccalc(itime + lsync, 1.0)

    for i in range(args.warmup):
        do_step(i)
    eng.sync()
    eng.kernel_time(reset=True)
    eng.counters(reset=True)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        do_step(i)
    grid_sum = None
    if args.config in (4, 5):
        grid, _ = eng.grids(allreduce=world > 1)     # the grid reduction over the ranks + D2H of the sums: part of the job
        grid_sum = float(np.asarray(grid, dtype=np.float64).sum())
    # the end of the timed region is an output time: the particle count over the ranks, as the reference's root reduces it
    t_cnt = time.perf_counter()
    eng.sync()
    t_cnt0 = time.perf_counter()
    (nlive_local, _), (nlive_total, numpart_total) = eng.count_particles(allreduce=world > 1)
    t_cnt = time.perf_counter() - t_cnt0             # the count + its reduction alone (the stream was drained just before): a fixed cost per output time
    eng.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    parts, launches = eng.kernel_times(reset=True)
    kms = sum(parts[:3])
    n_slices = eng.info("pbl_launches_per_step")   # the Langevin kernel runs in time slices: launches per step
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # particle-steps actually executed (a step only processes particles that are due;
    # the few that left the domain stay dead)
    cnt = eng.counters()
    nsteps_local = float(cnt["n_due"])
    if dist:
        t = torch.tensor([nsteps_local], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        psteps = float(t.item())
    else:
        psteps = nsteps_local
    value = psteps / dt

    # roofline of the dominant kernel (k_advance): compulsory traffic model of SURVEY.md 8(d)
    nx, ny, nz = (int(v) for v in sc["grid"])
    rb = args.real
    nfield3 = 3 if args.config == 2 else 5            # uu,vv,ww (+rho,drhodz in the PBL)
    field_bytes = nfield3 * 2 * nx * ny * nz * rb
    b_state = (9 * rb + 14) + (9 * rb + 10) if rb == 8 else 112
    b_alg = b_state + field_bytes / nper
    # dominant kernel: k_prep (stream/gather bound) for config 2, the Langevin kernel otherwise
    dom = 3 if args.config == 2 else 1          # parts[3] = k_prep alone (parts[0] also holds the work-list sort)
    dom_name = {3: "k_prep", 1: "k_pbl_loop"}[dom]
    avg_ms = parts[dom] / max(launches, 1)
    achieved = b_alg * (nsteps_local / max(launches, 1)) / (avg_ms * 1e-3) / 1e9
    traffic, traffic_src = measured_traffic(args.config, nper, dom_name), "committed PMC summary under profiles/ (rocprofv3 FETCH_SIZE/WRITE_SIZE passes of the same command)"
    out = {
        "metric": "particle-steps/sec (whole node) + achieved HBM GB/s, 1e8 particles",
        "value": value, "unit": "particle-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "f64" if rb == 8 else "f32",
        "data": "synthetic",
        "config": {"workload": (f"BASELINE config {args.config}: {ntot:.0e} particles"
                                + (f" sharded over {world} GPUs ({nper} on rank 0)" if world > 1 else "") + ", synthetic "
                                f"{nx}x{ny}x{nz} ECMWF-shaped fields" + (" with both polar caps (nglobal = sglobal = 1, particles to |lat| < 89)" if args.poles else "") + ", "
                                + ("advance+interpol_wind only (all above PBL, turbulence off)" if args.config == 2
                                   else "Hanna turbulence + CBL (ctl=1/5, ifine=11), PBL sub-stepping")
                                + (" + conccalc 360x180x10 every step + RCCL grid all-reduce" if args.config == 4 else "")
                                + (" + aerosol (settling, dry deposition), 241x121 nest, wet deposition, conccalc 360x180x10 every step"
                                   if args.config == 5 else "")
                                + (f" WITHOUT {args.without} (attribution run, not the config)" if args.without else "")
                                + f", rng={args.rng}, lsynctime=900"),
                   "particles_total": ntot, "particles_per_gpu": nper, "particle_steps_timed": psteps, "counters": cnt, "parallelism": f"particle-shard x{world}",
                   "sort_interval": args.sort_interval, "live_particles_all_ranks": nlive_total, "numpart_all_ranks": numpart_total,
                   # asked of the engine (fpx_get_info): met-field packs blended in time once per step -- decided from the
                   # configuration alone (blend_mode / the run's particle count over all ranks), identically on every rank
                   "time_blended_packs": bool(eng.info("time_blended_packs")), "blended_steps": eng.info("blended_steps"),
                   "global_particles": int(args.global_particles or ntot),
                   "pbl_launches_per_step": n_slices, "options": args.opt,
                   "rccl_nranks": world if transport == "rccl" else 0, "reduction_transport": transport,
                   "count_reduction_ms": t_cnt * 1e3,      # inside the timed region (an output time of the reference: timemanager_mpi.f90:552-562)
                   "gridunc_sum_all_ranks": grid_sum},
        # achieved / peak / frac: the HBM roofline on ALGORITHMIC bytes, as the bench contract defines it.  `bound` names what
        # actually limits the dominant kernel: config 2's k_prep is memory-latency bound at three waves per SIMD (gathers
        # and state streaming); the Langevin kernel of configs 3-5 is VALU-issue bound -- its HBM fraction is small by
        # construction (about 3e5 lane-instructions per particle-step against 175 algorithmic bytes) and `valu` carries
        # the fraction that measures its distance to the hardware limit, from a counter pass of this very run.
        "roofline": {"bound": "hbm" if args.config == 2 else "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": traffic_src if traffic else None,
                     "kernel": dom_name, "avg_launch_ms": avg_ms, "launches": launches,
                     "launch_note": (None if dom != 1 else f"k_pbl_loop runs as {n_slices} launches per step (time slices; the later ones work through the particles the "
                                     "earlier ones suspended): avg_launch_ms, traffic and the VALU counters are per STEP, i.e. summed over a step's launches; a rocprofv3 "
                                     f"kernel trace lists {n_slices} calls per step, whose durations add up to this figure"),
                     "step_kernels_ms": {"k_prep": parts[3] / max(launches, 1), "worklist_sort": (parts[0] - parts[3]) / max(launches, 1),
                                         "k_pbl_loop": parts[1] / max(launches, 1), "k_pbl_finish": parts[2] / max(launches, 1)},
                     "alg_bytes_per_particle_step": b_alg,
                     "limiter": ("memory latency of dependent gathers at 3 waves/SIMD (165 VGPRs), not HBM bandwidth" if args.config == 2
                                 else f"{'fp64' if rb == 8 else 'fp32'} VALU issue")},
    }
    ls = eng.lane_stats()
    if any(v[0] for v in ls.values()):        # only a library built with -DFPX_LANE_STATS counts
        out["roofline"]["lane_stats"] = ls
    if args.pmc_child:
        eng.close()
        return
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(args, sc, frac_pbl)
        except Exception as e:  # the baseline must not sink the GPU number
            out["cpu_baseline"] = {"value": None, "unit": "particle-steps/s", "cores": 1, "kind": "reference",
                                   "sample": f"failed: {e}"}
    eng.close()
    if rank == 0 and world == 1 and not args.no_pmc:
        # after the engine has released the GPU: the counter pass runs the same workload in a child process
        grp = n_slices if dom == 1 else 1
        out["roofline"]["valu"] = live_valu(args, dom_name, avg_ms, nsteps_local / max(launches, 1), grp)
        if dom == 1:
            out["roofline"]["valu"]["by_class"] = live_valu_classes(args, dom_name, avg_ms, out["roofline"]["valu"], grp)
        t_live, raw = live_traffic(args, dom_name, grp)
        if t_live is not None:
            out["roofline"]["traffic"] = t_live
            out["roofline"]["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run (child processes, same workload); 2*FETCH + WRITE"
        out["roofline"]["traffic_raw"] = raw
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
