#!/usr/bin/env python3
"""Headline benchmark: Mparticle-steps/s of the SPH step on a synthetic dam-break.

    python bench.py --gpus N --steps K --warmup W          (any N: for N > 1 without WORLD_SIZE in the
                                                            environment it starts the line below
                                                            itself, as a child process)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A "step" is one pass of the hot path (cell build + density + acceleration + integrate,
FULL neighbour mode) over the whole particle set, state resident in HBM.  Prints ONE JSON
line (rank 0).  N = 1: the 4 194 304-particle column (BASELINE configs[2]).  N > 1: strong
scaling of the 16 777 216-particle column (configs[3]) over N z-slabs, with rank 0's
single-context time of the same scene (`strong_scaling`) and the 67 108 864-particle 8:1:1
channel (configs[4]) as `other_scaling`.  See DESIGN.md §7 for the definitions used
in `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import subprocess
import sys
import time

# multi-process GPU work on this image needs dmabuf IPC (RCCL P2P fails with
# "hipIpcGetMemHandle: invalid argument" otherwise); exported already, kept here for safety
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PREHEAT_STEPS = 40             # untimed steps before the warm-up: the device's clocks settle ~15 steps after any idle period
PAIR_SAMPLE_EVERY = 5          # the density+acceleration pair is timed (two HIP events, ~10 us of
                               # stream time each) on every 5th step of the timed region
DENSITY_FORCE_BYTES = 64       # algorithmic bytes per particle of the density+force pass
                               # (density 20 B + force 44 B)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10,
                    help="untimed steps first (the first ~5 steps after an upload run 5-10 %% "
                         "slower than the steady state)")
    ap.add_argument("--particles", type=int, default=None,
                    help="particles in total (strong) / per GPU (weak).  Default: N = 1: 4 194 304 "
                         "(BASELINE configs[2], C3); N > 1 strong: 16 777 216 (configs[3], C4); "
                         "weak: 4 194 304 per GPU")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="N > 1: strong (default) = --particles in total in the unit box, cut in N "
                         "z-slabs (BASELINE configs[3]); weak = --particles per GPU in a box N "
                         "times as long in z (one unit box per slab)")
    ap.add_argument("--no-other-scaling", action="store_true",
                    help="N > 1: skip the side measurement (strong: configs[4], C5 = 67 108 864 "
                         "particles in the 8:1:1 channel; weak: the strong C4 run)")
    ap.add_argument("--other-particles", type=int, default=None,
                    help="N > 1: particle count of the side measurement (rehearsals)")
    ap.add_argument("--no-one-gpu-reference", action="store_true",
                    help="N > 1, strong: skip rank 0's single-context run of the same scene "
                         "(the denominator of strong_scaling.speedup)")
    ap.add_argument("--no-breaking-dam", action="store_true",
                    help="N = 1: skip the side record of the dam actually breaking (gravity + "
                         "walls, steps 500-520)")
    ap.add_argument("--cpu-sample", type=int, default=524288,
                    help="particles in the CPU-baseline sample (0 disables); with --cpu-steps "
                         "sized for ~10 s of single-thread work")
    ap.add_argument("--cpu-steps", type=int, default=12)
    ap.add_argument("--arithmetic", choices=("fast", "exact"), default="fast",
                    help="pair arithmetic of the headline: fast = SPH_HIP_MODE_FULL_FAST (same "
                         "neighbour sets and order, forces within the north star's 1e-4 of the CPU "
                         "reference), exact = bit-identical to the CPU restatement; N = 1 reports "
                         "the other one as a side record")
    ap.add_argument("--no-preheat", action="store_true",
                    help="N = 1: skip the 40 untimed steps on a scratch context right before the "
                         "warm-up (after ANY idle period of 50 ms or more - an upload is one - "
                         "the device needs ~15 steps to settle its clocks: tools/idle_ramp.py)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch rehearsal: start the ranks, form the process group, count them, "
                         "print {\"dry_run\": true, \"ranks\": N}; no GPU work (checks a node's "
                         "launch shape, and runs in the CPU test suite)")
    ap.add_argument("--preflight-child", metavar="STORE", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--preflight-kind", choices=("native", "torch"), default="torch", help=argparse.SUPPRESS)
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: start the N ranks with
    torch.distributed.run as a CHILD process and hand its exit code on.  This process never
    touches the GPU (and never replaces itself: an exec from a process that has initialised the
    GPU takes the machine down on this pool)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


PREFLIGHT_TIMEOUT_S = 150       # per stage (native, then torch P2P)


def preflight_child(args):
    """Helper process of one rank (started by preflight() below, before that rank has touched
    the GPU): bring the device-to-device path of `args.preflight_kind` up, exchange one checked
    message with each neighbour, exit 0 if it arrived.
      native  libsph_hip.so's own RCCL calls: a small slab context per rank, sph_hip_slab_comm_init
              (the communicator id travels through a file store) and sph_hip_slab_comm_exchange_check
              - ncclSend/ncclRecv of both directions in one group on the exchange stream, exactly what
              sph_hip_slab_comm_run issues every step;
      torch   torch.distributed P2P (batch_isend_irecv, backend nccl = RCCL).
    Anything else - an exception, a wrong byte, a hang that the parent's timeout ends - is a 'no'."""
    import datetime
    import torch
    import torch.distributed as dist
    from smoothed_particle_hydrodynamics_amd import slab as SL
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = 0 if os.environ.get("SPH_BENCH_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    code = 3
    try:
        torch.cuda.set_device(local_rank)
        if args.preflight_kind == "native":
            import smoothed_particle_hydrodynamics_amd as S
            store = dist.FileStore(args.preflight_child, world)
            # 4 cell planes per rank (a slab must be at least two halos thick), a handful of entries
            p = S.default_params(0.05, (2, 2, 2 * world))
            assert p.full_cells_z >= 4 * world
            planes = [r * p.full_cells_z // world for r in range(world + 1)]
            slab = SL.HipSlab(p, planes[rank], planes[rank + 1], 1024, 256, device=local_rank,
                              has_left=rank > 0, has_right=rank + 1 < world)
            if rank == 0:
                store.set("rccl_id", SL.rccl_unique_id())
            ident = store.get("rccl_id")            # (waits for rank 0)
            slab.comm_init(ident, rank, world)
            slab.comm_exchange_check()
            code = 0
        else:
            dist.init_process_group("nccl", init_method="file://" + args.preflight_child, rank=rank,
                                    world_size=world, timeout=datetime.timedelta(seconds=90),
                                    device_id=torch.device("cuda", local_rank))
            if SL.neighbour_exchange_works(rank, world, "cuda", timeout_s=60.0):
                code = 0
    except Exception as exc:      # noqa: BLE001
        print("pre-flight helper (%s) of rank %d: %r" % (args.preflight_kind, rank, exc), file=sys.stderr,
              flush=True)
    sys.stderr.flush()
    os._exit(code)                # no communicator teardown: it may be the thing that hangs


def preflight_stage(kind, rank, world):
    """Does the device-to-device neighbour exchange of `kind` ("native" / "torch") work on this
    node?  Asked of a helper process per rank, with a time limit, BEFORE this process initialises the
    GPU or RCCL: a hang or a crash in there is then a 'no', not a stuck or dead run."""
    if os.environ.get("SPH_BENCH_FORCE_P2P_FALLBACK") == "1":
        return False
    # rendezvous file of the helpers: one per launch and stage (all ranks of a launch are children of
    # the same torch.distributed.run agent; a file left by an earlier launch must not be found again)
    store = "/tmp/sph_bench_preflight_%s_%s_%d" % (kind, os.environ.get("MASTER_PORT", "0"), os.getppid())
    cmd = [sys.executable, os.path.abspath(__file__), "--preflight-child", store, "--preflight-kind", kind,
           "--gpus", str(world)]
    try:
        ok = subprocess.run(cmd, timeout=PREFLIGHT_TIMEOUT_S).returncode == 0
    except subprocess.TimeoutExpired:       # (subprocess.run has killed the child)
        print("pre-flight helper (%s) of rank %d: no answer after %d s" % (kind, rank, PREFLIGHT_TIMEOUT_S),
              file=sys.stderr, flush=True)
        ok = False
    return ok


def choose_transport(rank, world, torch, dist):
    """Order of preference: the library's own RCCL loop (no Python in a step) -> torch.distributed
    P2P -> messages staged through the host over gloo.  Each candidate is tried by every rank's
    helper process at the same time (the ranks meet over the gloo group between the stages, which
    needs no GPU), and a 'no' from any rank moves all of them on.  SPH_SLAB_TRANSPORT names one
    outright (native / torch / host).  -> (mode, what the stages said)"""
    forced = os.environ.get("SPH_SLAB_TRANSPORT")
    if forced:
        return forced, {"forced": forced}
    said = {}
    for kind in ("native", "torch"):
        verdict = torch.tensor([1 if preflight_stage(kind, rank, world) else 0], dtype=torch.int32)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
        said[kind] = bool(int(verdict.item()))
        for leftover in ("/tmp/sph_bench_preflight_%s_%s_%d" % (kind, os.environ.get("MASTER_PORT", "0"), os.getppid()),):
            if rank == 0:
                try:
                    os.remove(leftover)
                except OSError:
                    pass
        if said[kind]:
            return kind, said
    return "host-fallback", said


def cpu_baseline(n_sample, steps, n_total):
    """Times the CPU restatement (oracle/, kind 'port', 1 thread) on a thinner slice of the
    same column: same number density, same h, same neighbour count as the GPU workload."""
    from oracle.oracle import Oracle, OracleParams
    import ctypes as C
    from smoothed_particle_hydrodynamics_amd import scenes
    frac = n_sample / float(n_total)
    # same density: shrink the column's z extent by the particle ratio
    p, pos, vel, mass = scenes.dam_break(n_sample, box=(1.0, 1.0, frac))
    op = OracleParams()
    C.memmove(C.byref(op), C.byref(p), C.sizeof(op))
    orc = Oracle()
    orc.step(op, pos, vel, mass, mode="full")  # untimed: page in, warm caches
    t0 = time.perf_counter()
    for _ in range(steps):
        out = orc.step(op, pos, vel, mass, mode="full")
    dt = time.perf_counter() - t0
    return {
        "value": n_sample * steps / dt / 1e6,
        "unit": "Mparticle-steps/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d-particle slice of the same dam-break column (same density, h, %.1f "
                  "neighbours/particle), %d FULL-mode steps, %.1f s; host has %d cores" % (
                      n_sample, float(out["ncount"].mean()), steps, dt, os.cpu_count()),
    }


def reference_pair_functions(n_sample, n_total, passes=24):
    """The compiled reference's OWN computeDensity + computeAcceleration (src/sph.cpp:721-766, 778-934;
    oracle/_ref/libsphref.so, one thread like the original) on complete neighbour lists in the
    canonical order, for a slice of the same dam-break column - the pair functions of the pass
    `roofline` prices, timed where `cpu_baseline` times the port's whole step.  The lists are built
    (untimed) by the port, the reference's results are checked against the port's bit for bit.
    None when the reference build is not present."""
    import ctypes as C
    from oracle import oracle as orc
    from smoothed_particle_hydrodynamics_amd import scenes
    frac = n_sample / float(n_total)
    # (the scene first: it loads libsph_hip.so and with it torch's ROCm runtime, which must be in the
    # process before libsphref.so - Qt5Core from /opt/conda, static libstdc++ - is)
    p, pos, vel, mass = scenes.dam_break(n_sample, box=(1.0, 1.0, frac))
    if not orc.reference_available():
        return None
    op = orc.OracleParams()
    C.memmove(C.byref(op), C.byref(p), C.sizeof(op))
    port = orc.Oracle()
    cap = 96
    nb, nd, cnt, worst = port.full_build_lists(op, pos, cap)
    if worst > cap:
        cap = int(worst)
        nb, nd, cnt, worst = port.full_build_lists(op, pos, cap)
    ref = orc.Reference()
    ref.configure(op, n_sample)
    ref.set_state(pos, vel, mass)
    ref.set_lists(cap, nb, nd, cnt)
    ref.compute_density()                       # untimed: first touch
    ref.compute_acceleration()
    t0 = time.perf_counter()
    for _ in range(passes):
        ref.compute_density()
    t1 = time.perf_counter()
    for _ in range(passes):
        ref.compute_acceleration()
    t2 = time.perf_counter()
    got = ref.get_state()
    _, cs, ci = port.full_cells(op, pos)
    rho, _ = port.full_density(op, pos, mass, cs, ci)
    acc = port.full_accel(op, pos, vel, mass, rho, cs, ci)
    same = bool(np.array_equal(got["rho"], rho) and np.array_equal(got["acc"], acc))
    dens, accel = (t1 - t0) / passes, (t2 - t1) / passes
    return {"value": n_sample / (dens + accel) / 1e6, "unit": "Mparticle-passes/s (density + acceleration)",
            "cores": 1, "kind": "reference",
            "ms_density": dens * 1e3, "ms_acceleration": accel * 1e3,
            "equals_port_bit_for_bit": same,
            "sample": "%d-particle slice of the same column, %.1f neighbours/particle, complete canonical lists "
                      "(built untimed by the port), %d passes of SPH::computeDensity + SPH::computeAcceleration "
                      "of the compiled src/sph.cpp, %.1f s" % (n_sample, float(cnt.mean()), passes, t2 - t0)}


def reference_scene(S, steps=8, n=32768):
    """The reference's own compiled src/sph.cpp (oracle/_ref, built where /root/reference exists
    and shipped as a .so) timed on this box on ITS default scene - srand(42) sphere, 32 768
    particles, shipped sampled neighbour search, one thread like the original - next to the HIP
    library in REF mode on the same initial state, with a live check that both end in the same
    bits.  None when the reference build is not present."""
    import ctypes as C
    from oracle import oracle as orc
    if not orc.reference_available():
        return None
    p = S.default_params()
    op = orc.OracleParams()
    C.memmove(C.byref(op), C.byref(p), C.sizeof(op))
    ref = orc.Reference()
    ref.configure(op, n)
    ref.init_sphere()
    s0 = ref.get_state()
    ref.step()                                   # untimed: first touch
    ref.set_state(s0["pos"], s0["vel"], s0["mass"])
    t0 = time.perf_counter()
    for _ in range(steps):
        ref.step()
    dt_cpu = (time.perf_counter() - t0) / steps
    s1 = ref.get_state()
    with S.SPH(n, p, mode=S.MODE_REF) as sph:
        sph.setParticles(s0["pos"], s0["vel"], s0["mass"])
        for _ in range(steps):
            sph.step()
        part = sph.getParticles()
        same = bool(np.array_equal(part.mPosition, s1["pos"]) and
                    np.array_equal(part.mVelocity, s1["vel"]) and
                    np.array_equal(part.mDensity, s1["rho"]) and
                    np.array_equal(part.mNeighborCount, s1["ncount"]))
        sph.setParticles(s0["pos"], s0["vel"], s0["mass"])
        sph.setTiming(S.TIMING_OFF)
        sph.run(5)
        sph.synchronize()
        k = 200
        t0 = time.perf_counter()
        sph.run(k)
        sph.synchronize()
        dt_gpu = (time.perf_counter() - t0) / k
    return {
        "workload": "the reference's default scene: srand(42) sphere, %d particles, shipped "
                    "sampled neighbour search (REF mode)" % n,
        "cpu": {"value": n / dt_cpu / 1e6, "unit": "Mparticle-steps/s", "ms_per_step": dt_cpu * 1e3,
                "cores": 1, "kind": "reference",
                "sample": "%d x SPH::step() of the compiled src/sph.cpp" % steps},
        "gpu": {"value": n / dt_gpu / 1e6, "unit": "Mparticle-steps/s", "ms_per_step": dt_gpu * 1e3},
        "speedup": dt_cpu / dt_gpu,
        "identical_after_steps": steps if same else 0,
    }


# (tools/profile_round.sh points these at the files it has just written under gpurun_out/, which are
# then committed under profiles/ with the same names)
PROFILE = os.environ.get("SPH_BENCH_COUNTERS") or os.path.join(ROOT, "profiles", "r4_kernel_counters.json")


def kernel_counters(n, arithmetic="fast"):
    """What the committed counter passes say about the density + acceleration launch pair
    (tools/profile_round.sh -> profiles/r4_kernel_counters.json, one entry per arithmetic): HBM bytes (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs, FETCH doubled as the gfx950 guide prescribes) and
    VALU wave-instructions (SQ_INSTS_VALU).  The file carries the hash of the kernel sources it
    was measured on: (None, reason) when that is not the code being run, or the workload differs."""
    from smoothed_particle_hydrodynamics_amd.build import source_hash
    try:
        prof = json.load(open(PROFILE))
    except (OSError, ValueError):
        return None, "no committed counter profile"
    if prof.get("particles") != n:
        return None, "the committed counters are for %s particles" % prof.get("particles")
    if arithmetic not in prof.get("arithmetic", {}):
        return None, "the committed counters do not cover the %s arithmetic" % arithmetic
    if prof.get("csrc_sha16") != source_hash():
        return None, ("the committed counters were taken on kernel sources %s, this is %s: "
                      "re-run tools/profile_round.sh" % (prof.get("csrc_sha16"), source_hash()))
    out = dict(prof["arithmetic"][arithmetic], csrc_sha16=prof["csrc_sha16"])
    return out, "profiles/%s, %s arithmetic (kernel sources %s)" % (
        os.path.basename(PROFILE).replace("r3a_", "r3_"), arithmetic, prof["csrc_sha16"])


CENSUS = os.environ.get("SPH_BENCH_CENSUS") or os.path.join(ROOT, "profiles", "r4_valu_census.json")


def valu_issue(prof, pair_ms, arithmetic="fast"):
    """What share of the SIMDs' issue cycles the pair's VALU instructions book - the bound that is real
    for this pass - as ONE number, from the committed census
    (tools/valu_census.py -> profiles/r4_valu_census.{md,json}):
      * wave-instructions per phase: SQ_INSTS_VALU of the shipped kernels and of builds with one phase
        cut out (TEST / per-chunk bookkeeping / append / SUM / pair loops / prologues);
      * the opcode mix of each phase from the line tables of the shipped ISA;
      * a measured price for every opcode (tools/ubench/valu3.hip, valu5.hip: every SIMD saturated,
        8 waves per SIMD; plain 2.24 cycles, packed / compare / select / shift-left / min-max 4.1-4.2,
        transcendental 8.1).
    frac = 64 waves per SIMD x priced issue cycles per wave / (pair duration x shader clock).  The
    prices are additive - a wave's wide-class and plain instructions are charged as if they never
    overlapped - so the figure is an upper estimate of the port's occupancy; what is left to 1 is time
    the VALU port idles (dependent chains of sparse loops, LDS and memory latency, barriers).  None
    when the census was taken on other kernel sources or covers another arithmetic."""
    from smoothed_particle_hydrodynamics_amd.build import source_hash
    try:
        census = json.load(open(CENSUS))
    except (OSError, ValueError):
        return None
    if census.get("csrc_sha16") != source_hash() or census.get("arithmetic") != arithmetic:
        return {"frac": None, "note": "the committed census (%s, %s arithmetic) does not describe the kernels being "
                                      "run (%s): re-run tools/pmc_census.sh + tools/valu_census.py" % (
                                          census.get("csrc_sha16"), census.get("arithmetic"), source_hash())}
    clock = (prof or {}).get("shader_clock_ghz_while_profiled") or census["ghz"]
    cycles = census["pair"]["issue_cycles_per_simd"]
    have = pair_ms * 1e-3 * clock * 1e9
    return {"frac": min(1.0, cycles / have),
            "frac_additive_prices": cycles / have,
            "prices": "per opcode, measured in isolation and summed; mixed streams (tools/ubench/valu6.hip, "
                      "profiles/r4_valu_mix.txt) deviate from the sum by several per cent both ways - fp32 operations "
                      "hide behind wide-class ones, packed and plain integer ones cost more next to them - so a sum "
                      "above 1 is clamped: the pass cannot use more than all of its issue cycles",
            "issue_cycles_per_simd_per_launch_pair": cycles,
            "simd_cycles_per_launch_pair": have, "clock_ghz": clock, "simds": 1024,
            "wave_instructions_per_launch_pair": census["pair"]["wave_instructions_per_launch_pair"],
            "not_fp32_arithmetic_per_launch_pair": census["pair"]["non_arithmetic_per_launch_pair"],
            "by_phase": [{"phase": ph["phase"], "wave_instructions_per_wave": round(ph["wave_instructions_per_wave"], 1),
                          "issue_cycles_per_wave": round(ph["issue_cycles_per_wave"], 1)} for ph in census["phases"]],
            "residual": "attributed wave-instructions = SQ_INSTS_VALU exactly (phases are differences of builds); "
                        "1 - frac = issue cycles the VALU port idles",
            "source": "profiles/%s (kernel sources %s)" % (os.path.basename(CENSUS), census["csrc_sha16"])}


def phase_split(ctx, S, step, synchronize, set_timing, phase_totals, steps=5):
    """Per-phase milliseconds over `steps` fully instrumented steps (outside the timed region)."""
    set_timing(S.TIMING_PHASES)
    for _ in range(steps):
        step()
    synchronize()
    t, k = phase_totals()
    return {"cell_build": t[0] / k, "density": t[2] / k, "acceleration": t[4] / k,
            "integrate": t[5] / k, "steps": k}


def tile_summary(ts):
    """What the last step's launches were sized for (sph_hip_get_tile_stats): LDS tile capacities
    of the two passes (they decide the workgroups per CU), the largest tile, workgroups that went
    the untiled route, entries per neighbour list."""
    return {"capacity_density": ts["capacity_density"],
            "capacity_acceleration": ts["capacity_acceleration"],
            "largest_tile": ts["largest_tile"], "workgroups": ts["workgroups"],
            "workgroups_untiled": [ts["untiled_density"], ts["untiled_acceleration"]],
            "list_capacity": ts["list_capacity"]}


def timed_steps(sph, S, torch, warmup, steps):
    """`warmup` untimed steps, then exactly `steps` timed ones (device drained on both sides); the
    density+acceleration pair is bracketed by HIP events on the context's stream on every
    PAIR_SAMPLE_EVERY-th timed step only - an event record costs ~10 us of stream time."""
    sph.setTiming(S.TIMING_SUMS)
    for _ in range(warmup):
        sph.step()
    sph.synchronize()
    torch.cuda.synchronize()
    sph.setTimingStride(PAIR_SAMPLE_EVERY)
    t0 = time.perf_counter()
    for _ in range(steps):
        sph.step()
    sph.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pair, covered = sph.phaseTotals()
    sph.setTimingStride(1)
    return dt, pair[2] / covered, covered


def run_single(args, S, scenes, torch, local_rank):
    """N = 1: one context holds the whole grid."""
    n = args.particles
    fast = args.arithmetic == "fast"
    p, pos, vel, mass = scenes.dam_break(n)
    sph = S.SPH(n, p, mode=S.MODE_FULL_FAST if fast else S.MODE_FULL, device=local_rank)
    sph.setParticles(pos, vel, mass)
    heater = None
    if not args.no_preheat:
        # Any idle period of >= 50 ms - the upload above is one - leaves the device ~15 steps away
        # from its steady clocks (1.19, 1.27, 1.38, 1.37, 1.32 ... 1.06 ms per step, the same after
        # a plain sleep: tools/idle_ramp.py): 40 untimed steps of the same workload on a scratch
        # context, right before the warm-up, so that the timed region measures the step and not
        # the power management's settling.
        heater = S.SPH(n, p, mode=S.MODE_FULL_FAST if fast else S.MODE_FULL, device=local_rank)
        heater.setParticles(pos, vel, mass)
        heater.setTiming(S.TIMING_OFF)
        heater.run(PREHEAT_STEPS)
        heater.synchronize()
    dt, pair_ms, covered = timed_steps(sph, S, torch, args.warmup, args.steps)
    totals = phase_split(sph, S, lambda: sph.step(), sph.synchronize, sph.setTiming, sph.phaseTotals)
    totals["pair_ms"] = pair_ms
    totals["tiles"] = tile_summary(sph.tileStats())
    nb_mean = float(sph.getParticles().mNeighborCount.mean())
    ke, pe = sph.energy()
    assert np.isfinite(ke) and np.isfinite(pe)
    # the other arithmetic on the same context, same state, right behind (device still busy):
    # a side record, never `value`
    sph.setArithmetic(S.ARITH_EXACT if fast else S.ARITH_FAST)
    odt, opair, ocovered = timed_steps(sph, S, torch, args.warmup, args.steps)
    parity = arithmetic_parity(sph, S, mass)
    other = {"arithmetic": "exact" if fast else "fast", "value": n * args.steps / odt / 1e6,
             "unit": "Mparticle-steps/s", "ms_per_step": odt / args.steps * 1e3,
             "ms_per_launch_pair": opair, "launch_pairs_timed": ocovered,
             "roofline_frac": DENSITY_FORCE_BYTES * n / (opair * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "note": "same context and state, %d warm-up + %d timed steps right behind the "
                     "headline's; side record" % (args.warmup, args.steps)}
    sph.close()
    if heater is not None:
        heater.close()
    return p, dt, totals, covered, n, nb_mean, "1 GPU", other, parity


def arithmetic_parity(sph, S, mass):
    """The line's own proof of the tolerance-mode arithmetic, GPU against GPU, outside every timed
    region: ONE step from the state the timed steps ended in, once with the bit-exact arithmetic
    (which the test suite holds to the CPU oracle bit for bit, tests/test_gpu_full_size.py) and once
    with the tolerance-mode one, from the same positions and velocities.  Neighbour counts and
    densities must be identical; the acceleration of EVERY particle within the north star's 1e-4
    relative (|a - a_exact| <= 1e-4 * max(|a|, |a_exact|)): `beyond_1e-4` counts those that are not."""
    part = sph.getParticles()
    pos, vel = part.mPosition.copy(), part.mVelocity.copy()
    res = {}
    for name, arith in (("exact", S.ARITH_EXACT), ("fast", S.ARITH_FAST)):
        sph.setArithmetic(arith)
        sph.setParticles(pos, vel, mass)
        sph.step()
        assert sph.getArithmetic() == arith
        q = sph.getParticles()
        res[name] = (q.mAcceleration.astype(np.float64).reshape(-1, 3), q.mDensity.copy(), q.mNeighborCount.copy())
    a, b = res["fast"][0], res["exact"][0]
    den = np.maximum(np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1))
    den[den == 0] = 1.0
    rel = np.linalg.norm(a - b, axis=1) / den
    return {"max_rel": float(rel.max()), "beyond_1e-4": int((rel > 1e-4).sum()),
            "p9999_rel": float(np.quantile(rel, 0.9999)),
            "counts_equal": bool(np.array_equal(res["fast"][2], res["exact"][2])),
            "density_equal": bool(np.array_equal(res["fast"][1], res["exact"][1])),
            "particles": int(rel.size),
            "note": "one step from the state the timed steps ended in, tolerance-mode arithmetic vs the "
                    "bit-exact one on the same context (GPU vs GPU, outside the timed regions); the "
                    "bit-exact arithmetic is held to the CPU oracle bit for bit by tests/test_gpu_full_size.py"}


PARITY_BAR = {
    "fast": "neighbour counts and densities identical to the CPU reference's; acceleration of EVERY particle "
            "within 1e-4 relative (vector norm) - asserted without escape clauses on every BASELINE "
            "configuration (tests/test_gpu_full_fast.py, test_gpu_full_size.py, test_gpu_c4_c5.py); see "
            "`parity` in this line for this run's own exact-vs-fast check",
    "exact": "every per-particle output bit-identical to the CPU oracle (tests/test_gpu_full_mode.py, "
             "test_gpu_full_size.py, test_gpu_c4_c5.py)",
}

C3_PARTICLES = 4 * 1024 * 1024      # BASELINE configs[2]
C4_PARTICLES = 16 * 1024 * 1024     # BASELINE configs[3]: strong scaling over the node
C5_PARTICLES = 64 * 1024 * 1024     # BASELINE configs[4]: 8:1:1 channel, long axis = slab axis
C5_BOX = (1.0, 1.0, 8.0)


def config_name(n, box):
    if box == (1.0, 1.0, 1.0):
        return {262144: " (BASELINE configs[1], C2)", C3_PARTICLES: " (BASELINE configs[2], C3)",
                C4_PARTICLES: " (BASELINE configs[3], C4)"}.get(n, "")
    if box == C5_BOX and n == C5_PARTICLES:
        return " (BASELINE configs[4], C5: 8:1:1 channel, long axis = slab axis)"
    return ""


def run_slabs(args, S, scenes, torch, rank, world, local_rank, n, box, steps, warmup, mode, exchange_group):
    """N > 1: `n` particles in `box`, one z-slab of the cell grid per GPU, neighbour exchange
    over RCCL (xGMI).  Every rank derives the same cuts from the z coordinates alone and
    generates only the particles it owns (counter-based PRNG: any subset of the scene on any
    rank).  Everything that is not the halo exchange - barriers, the agreements about message size
    and cuts, the max-over-ranks time - goes over the default gloo group with host tensors."""
    import torch.distributed as dist
    from smoothed_particle_hydrodynamics_amd import slab as SL
    p, hi = scenes.dam_break_params(n, box)
    z = scenes.box_fill_axis(n, (0.0, 0.0, 0.0), hi, 2)
    planes = SL.plane_of(p, z)
    cuts = SL.plan_cuts(p, z, world)
    hist = np.bincount(planes, minlength=p.full_cells_z)
    cap, msg = SL.slab_capacities(hist, cuts, rank, slack=1.5)
    mine = np.nonzero((planes >= cuts[rank]) & (planes < cuts[rank + 1]))[0]
    del z, planes
    def make_slab(new_cuts, r, plane_hist):
        c, m = SL.slab_capacities(np.asarray(plane_hist), new_cuts, r, slack=1.5)
        return SL.HipSlab(p, new_cuts[r], new_cuts[r + 1], c, max(m, msg), device=local_rank,
                          has_left=r > 0, has_right=r + 1 < world)

    first = SL.HipSlab(p, cuts[rank], cuts[rank + 1], cap, msg, device=local_rank,
                       has_left=rank > 0, has_right=rank + 1 < world)
    first.set_arithmetic(S.ARITH_FAST if args.arithmetic == "fast" else S.ARITH_EXACT)
    first.set_timing(S.TIMING_SUMS)
    first.upload(mine.astype(np.uint32), scenes.box_fill_subset(mine, (0.0, 0.0, 0.0), hi),
                 np.zeros(3 * mine.size, np.float32), np.ones(mine.size, np.float32),
                 all_masses_equal=True)
    if mode == "native":
        stepper = SL.NativeSlabStepper(first, rank, world)
    else:
        if mode in ("host", "host-fallback"):
            transport = SL.HostStagedTransport(rank, world)                 # default group: gloo
        else:
            transport = SL.DistTransport(rank, world, group=exchange_group)  # RCCL, P2P only
        # cuts are re-evaluated every 500 steps (a count per rank; particles move only when the
        # fullest slab is 10 % over the mean), the message size every 256; a slab that replaces
        # this one after a rebalance is stepper.slab - `first` is not used past this point
        stepper = SL.DistSlabStepper(first, transport, make_slab=make_slab, cuts=cuts,
                                     rebalance_every=500, imbalance=1.1, trim_every=256,
                                     control_group=dist.group.WORLD)
    del first

    def fence():
        stepper.slab.synchronize()
        torch.cuda.synchronize()
        dist.barrier()

    if not args.no_preheat:
        # the upload left the device idle: PREHEAT_STEPS untimed steps of this very run, before the
        # W warm-up steps, let its clocks settle (run_single has the story; the column at rest
        # does not change measurably in that many steps)
        stepper.run(PREHEAT_STEPS)
    stepper.run(warmup)
    fence()
    # from here on only the used part of the halo messages crosses the links (+25 % head room)
    msg_records = stepper.trim_messages() if warmup > 0 else stepper.slab.msg_capacity
    fence()
    stepper.slab.set_timing_stride(PAIR_SAMPLE_EVERY)
    t0 = time.perf_counter()
    stepper.run(steps)         # (native: the whole loop inside libsph_hip.so; torch: a Python loop over step())
    fence()
    dt_local = time.perf_counter() - t0
    t = torch.tensor([dt_local], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    slab = stepper.slab
    st = slab.status()
    err = torch.tensor([st["errors"]], dtype=torch.int64)
    dist.all_reduce(err, op=dist.ReduceOp.MAX)
    own = torch.tensor([st["owned"]], dtype=torch.int64)
    dist.all_reduce(own, op=dist.ReduceOp.SUM)
    if int(err.item()) != 0 or int(own.item()) != n:
        raise SystemExit("slab run inconsistent: error bits %d, owned %d of %d" %
                         (int(err.item()), int(own.item()), n))
    pair, covered = slab.phase_totals()
    slab.set_timing_stride(1)
    totals = phase_split(slab, S, stepper.step, fence, lambda lv: stepper.slab.set_timing(lv),
                         lambda: stepper.slab.phase_totals())
    slab = stepper.slab
    totals["pair_ms"] = pair[2] / max(covered, 1)
    totals["tiles"] = tile_summary(slab.tile_stats())
    d = slab.download()
    nb_mean = float(d["ncount"].mean())
    msg_allocated = slab.msg_capacity
    slab.close()
    return {"params": p, "dt": dt, "totals": totals, "covered": covered, "n": n,
            "n_rank": st["owned"], "neighbors_mean": nb_mean, "steps": steps, "box": box,
            "ranks": dist.get_world_size(),
            "rebalances_in_run": int(getattr(stepper, "rebalances", 0)),
            "message_growths_in_run": int(getattr(stepper, "message_growths", 0)),
            "halo_message_bytes": SL.message_bytes(msg_records),
            "halo_message_bytes_allocated": SL.message_bytes(msg_allocated),
            "parallelism": "z-slab x%d, RCCL halo (%s)" % (
                world, {"native": "ncclSend/ncclRecv issued by libsph_hip.so: no Python in a step",
                        "host": "host-staged rehearsal",
                        "host-fallback": "HOST-STAGED FALLBACK over gloo: the device-to-device "
                                         "pre-flights (native RCCL, torch P2P) failed on this node"}.get(
                            mode, "torch.distributed P2P"))}


def one_gpu_reference(S, scenes, torch, local_rank, n, box, steps, warmup, fast, preheat=True):
    """The same scene on ONE GPU in a single context (rank 0 only, the other ranks wait): the
    denominator of the strong-scaling speedup, measured in the same run on the same node (with the
    same pre-heat as the slab run it is compared with)."""
    p, pos, vel, mass = scenes.dam_break(n, box)
    with S.SPH(n, p, mode=S.MODE_FULL_FAST if fast else S.MODE_FULL, device=local_rank) as sph:
        sph.setParticles(pos, vel, mass)
        del pos, vel, mass
        sph.setTiming(S.TIMING_OFF)
        sph.run(warmup + (PREHEAT_STEPS if preheat else 0))
        sph.synchronize()
        t0 = time.perf_counter()
        sph.run(steps)
        sph.synchronize()
        dt = time.perf_counter() - t0
    return dt / steps * 1e3


def breaking_dam(S, scenes, n, device, fast, settle=500, timed=20):
    """Side record, never `value`: the same scene with the dam actually breaking (uniform gravity
    and wall reflection switched on - SURVEY.md 8(f) rank 1), timed over steps settle ..
    settle + timed, when the column has collapsed and the flow is several times denser than the
    column at rest the headline steps (faithful to the reference, whose computeAcceleration has
    no uniform gravity)."""
    p, pos, vel, mass = scenes.dam_break(n)
    p.apply_gravity = 1
    p.apply_walls = 1
    p.gravity[0], p.gravity[1], p.gravity[2] = 0.0, -9.81, 0.0
    with S.SPH(n, p, mode=S.MODE_FULL_FAST if fast else S.MODE_FULL, device=device) as sph:
        sph.setParticles(pos, vel, mass)
        sph.setTiming(S.TIMING_OFF)
        sph.run(settle)
        sph.synchronize()
        t0 = time.perf_counter()
        sph.run(timed)
        sph.synchronize()
        dt = (time.perf_counter() - t0) / timed
        c = sph.getParticles().mNeighborCount
        ts = sph.tileStats()
    return {"workload": "the %d-particle column with apply_gravity + apply_walls, steps %d-%d" % (
                n, settle, settle + timed),
            "ms_per_step": dt * 1e3, "value": n / dt / 1e6, "unit": "Mparticle-steps/s",
            "neighbors_mean": float(c.mean()), "neighbors_max": int(c.max()),
            "list_capacity": int(ts["list_capacity"]),
            "particles_without_list": int((c > ts["list_capacity"]).sum()),
            "workgroups_untiled": [int(ts["untiled_density"]), int(ts["untiled_acceleration"])],
            "note": "side record; `value` above is the column at rest, as in the reference"}


def main():
    args = parse_args()
    if args.preflight_child:
        preflight_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started the way the N = 1 bench is started: become the launcher (nothing has touched the
        # GPU in this process, and nothing will)
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SPH_BENCH_ONE_DEVICE") == "1":   # rehearsal: all ranks share device 0
        local_rank = 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run "
                         "--nproc-per-node %d" % (args.gpus, world, args.gpus))
    import torch                      # (pages the image in before any helper process needs it)
    if args.dry_run:
        ranks = 1
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t = torch.ones(1, dtype=torch.int64)
            dist.all_reduce(t)
            ranks = int(t.item())
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": args.gpus, "ranks": ranks}))
        return
    # Which transport carries the halo messages: decided before this process touches the GPU.  The
    # default process group is gloo - barriers, reductions and every agreement between the ranks use
    # host tensors and keep working whatever state the device-to-device path is in.
    mode, preflight_said = "torch", None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        mode, preflight_said = choose_transport(rank, world, torch, dist)
    import smoothed_particle_hydrodynamics_amd as S
    from smoothed_particle_hydrodynamics_amd import scenes

    torch.cuda.set_device(local_rank)
    # The library is built in-tree by __graft_entry__.build(); only a missing one is compiled
    # here, and only when no other rank could be loading it at the same time.
    if not os.path.exists(S.library_path()):
        if world > 1:
            raise SystemExit("libsph_hip.so is missing: run `python -c 'import __graft_entry__ as "
                             "g; g.build()'` before a multi-rank launch")
        S.build_library()
    strong_scaling = None
    other_arith = None
    parity = None
    fast = args.arithmetic == "fast"
    if world > 1:
        # RCCL gets a group of its own that only torch's halo exchange uses (the native loop has its
        # own communicator inside libsph_hip.so)
        exchange_group = dist.new_group(backend="nccl") if mode == "torch" else None
        # strong (default): BASELINE configs[3] - 16M particles in the unit box cut in `world`
        # slabs; weak: `world` unit boxes in a row, --particles per GPU
        if args.scaling == "strong":
            n = args.particles or C4_PARTICLES
            box = (1.0, 1.0, 1.0)
        else:
            n = (args.particles or C3_PARTICLES) * world
            box = (1.0, 1.0, float(world))
        r = run_slabs(args, S, scenes, torch, rank, world, local_rank, n, box, args.steps, args.warmup,
                      mode, exchange_group)
        p, dt, totals, covered = r["params"], r["dt"], r["totals"], r["covered"]
        n_rank, nb_mean, par, ranks = r["n_rank"], r["neighbors_mean"], r["parallelism"], r["ranks"]
        halo = {"bytes_per_message": r["halo_message_bytes"],
                "bytes_allocated": r["halo_message_bytes_allocated"],
                # (a rebalance replaces the slab context: a timed region that contained one is recognisable)
                "rebalances_in_run": r["rebalances_in_run"], "message_growths_in_run": r["message_growths_in_run"]}
        if args.scaling == "strong" and not args.no_one_gpu_reference:
            # the same scene on one GPU, same run: what the N-GPU time is a speedup OF
            one_ms = None
            if rank == 0:
                one_ms = one_gpu_reference(S, scenes, torch, local_rank, n, box,
                                           max(5, args.steps // 2), min(args.warmup, 5), fast,
                                           preheat=not args.no_preheat)
            dist.barrier()
            if rank == 0:
                strong_scaling = {"particles": n, "one_gpu_ms_per_step": one_ms,
                                  "n_gpu_ms_per_step": dt / args.steps * 1e3,
                                  "speedup": one_ms / (dt / args.steps * 1e3), "gpus": world,
                                  "note": "rank 0 steps the whole scene in one context while "
                                          "the other ranks wait; same node, same run"}
        other = None
        if not args.no_other_scaling:
            # side measurement, reported beside the headline (never as `value`): with the strong
            # headline the 64M-particle 8:1:1 channel of BASELINE configs[4] (long axis = slab
            # axis, largest halo volume per step); with the weak headline the strong C4 run
            if args.scaling == "strong":
                on, obox, oname = args.other_particles or C5_PARTICLES, C5_BOX, "strong"
            else:
                on, obox, oname = args.other_particles or C4_PARTICLES, (1.0, 1.0, 1.0), "strong"
            o = run_slabs(args, S, scenes, torch, rank, world, local_rank, on, obox,
                          max(5, args.steps // 2), min(args.warmup, 3), mode, exchange_group)
            other = {"scaling": oname, "particles": o["n"], "box": list(o["box"]),
                     "workload": "dam-break %d particles%s in a %gx%gx%g box" % (
                         o["n"], config_name(o["n"], obox), obox[0], obox[1], obox[2]),
                     "value": o["n"] * o["steps"] / o["dt"] / 1e6, "unit": "Mparticle-steps/s",
                     "steps": o["steps"], "ms_per_step": o["dt"] / o["steps"] * 1e3,
                     "neighbors_mean": o["neighbors_mean"], "ranks": o["ranks"]}
    else:
        n = args.particles or C3_PARTICLES
        args.particles = n
        p, dt, totals, covered, n_rank, nb_mean, par, other_arith, parity = run_single(args, S, scenes, torch,
                                                                                       local_rank)
        box, other, ranks = (1.0, 1.0, 1.0), None, 1

    if rank == 0:
        # density+acceleration pair: HIP events on the context's stream over the timed steps
        df_ms = totals["pair_ms"]
        achieved = DENSITY_FORCE_BYTES * n_rank / (df_ms * 1e-3) / 1e9
        prof, prof_note = kernel_counters(n, args.arithmetic) if world == 1 else (None, "1-GPU profile only")
        valu = valu_issue(prof, df_ms, args.arithmetic) if world == 1 else None
        traffic = prof["density_plus_acceleration_hbm_bytes"] if prof else None
        scaling = args.scaling if world > 1 else "weak"
        line = {
            "metric": "Mparticle-steps/sec (whole node), dam-break" + (
                "" if world == 1 else ", %s scaling, %d particles in total on %d GPUs" % (
                    scaling, n, world)),
            "value": n * args.steps / dt / 1e6,
            "unit": "Mparticle-steps/s",
            "n_gpus": world,
            "ranks": ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "arithmetic": args.arithmetic,
            "parity_bar": PARITY_BAR[args.arithmetic],
            "config": {
                "workload": "dam-break %d particles%s in a %gx%gx%g box, fp32, FULL neighbour "
                            "mode, cell grid rebuilt every step, %s" % (
                                n, config_name(n, box) if world == 1 or scaling == "strong" else
                                " = %d x %d, one unit box per slab along z" % (world, n // world),
                                box[0], box[1], box[2],
                                "tolerance-mode pair arithmetic (SPH_HIP_MODE_FULL_FAST: neighbour sets "
                                "and order identical to the CPU reference, forces within 1e-4 relative)"
                                if fast else "pair arithmetic bit-identical to the CPU reference"),
                "arithmetic": args.arithmetic,
                "particles": n,
                "particles_per_gpu": n // world,
                "h": float(p.h),
                "grid": [p.full_cells_x, p.full_cells_y, p.full_cells_z],
                "neighbors_mean": nb_mean,
                "parallelism": par,
            },
            "phases_ms": {
                "cell_build": totals["cell_build"], "density": totals["density"],
                "acceleration": totals["acceleration"], "integrate": totals["integrate"],
                "note": "%d fully instrumented steps after the timed region" % totals["steps"],
            },
            "lds_tiles": totals.get("tiles"),
            "roofline": {
                "bound": "hbm",
                "kernel": "density+acceleration pass (k_full_density_tiled + k_full_accel_lists)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_unit": "bytes per launch pair",
                "traffic_source": prof_note,
                # the counter bytes as a rate against the chip's peak (what rocprof says the pair moves
                # through HBM per second of ITS OWN duration), and against the algorithmic bytes
                "traffic_gbs": None if traffic is None else traffic / (df_ms * 1e-3) / 1e9,
                "traffic_frac": None if traffic is None else traffic / (df_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic_over_algorithmic": None if traffic is None else traffic / float(DENSITY_FORCE_BYTES * n_rank),
                "valu": valu,
                "bytes_per_particle": DENSITY_FORCE_BYTES,
                "particles_per_launch": n_rank,
                "ms_per_launch_pair": df_ms,
                "launch_pairs_timed": covered,
                "note": "VALU-bound gather-sum: see DESIGN.md (Roofline); frac is against the "
                        "64 B/particle compulsory-traffic figure of SURVEY.md 8(d)",
                "fused_work": None if world > 1 else (
                    "on one GPU the acceleration launch of the pair also integrates every particle "
                    "and hashes + counts it for the next cell build (no k_integrate launch: "
                    "phases_ms.integrate is an event gap): 60 + 20 B/particle more of SURVEY.md "
                    "8(d)'s algorithmic bytes done inside ms_per_launch_pair and NOT counted in "
                    "achieved / frac"),
            },
        }
        if world > 1:
            line["config"]["halo"] = halo
            line["config"]["transport"] = mode
            line["config"]["transport_preflight"] = preflight_said
        if strong_scaling is not None:
            line["strong_scaling"] = strong_scaling
        if other is not None:
            line["other_scaling"] = other
        if other_arith is not None:
            line["other_arithmetic"] = other_arith
            # the bit-exact arithmetic's throughput, as prominent as `value` (same run, same state)
            line["value_%s_arithmetic" % other_arith["arithmetic"]] = other_arith["value"]
        if parity is not None:
            line["parity"] = parity
        if not args.no_preheat:
            line["config"]["preheat"] = (
                "%d untimed steps of the same workload %s right before the warm-up (device clocks settle ~15 "
                "steps after any idle period: tools/idle_ramp.py)" % (
                    PREHEAT_STEPS, "on a scratch context" if world == 1 else "by every rank's own slab"))
        if world == 1 and not args.no_breaking_dam:
            line["breaking_dam"] = breaking_dam(S, scenes, n, local_rank, fast)
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(min(args.cpu_sample, n), args.cpu_steps, n)
            pair_fn = reference_pair_functions(min(args.cpu_sample, n), n)
            if pair_fn is not None:
                line["cpu_baseline"]["reference_pair_functions"] = pair_fn
            ref_scene = reference_scene(S)
            if ref_scene is not None:
                line["reference_scene"] = ref_scene
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
