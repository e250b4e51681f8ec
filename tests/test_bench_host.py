"""Host-side logic of bench.py that needs no GPU: workload naming, and the rule that counter
figures (roofline.traffic / roofline.valu) are reported only while the committed profile was
taken on the kernel sources being run."""
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_baseline_configs_are_named(bench):
    assert "configs[2]" in bench.config_name(4 * 1024 * 1024, (1.0, 1.0, 1.0))
    assert "configs[3]" in bench.config_name(bench.C4_PARTICLES, (1.0, 1.0, 1.0))
    assert "configs[4]" in bench.config_name(bench.C5_PARTICLES, bench.C5_BOX)
    assert bench.config_name(12345, (1.0, 1.0, 1.0)) == ""
    assert bench.C4_PARTICLES == 16777216 and bench.C5_PARTICLES == 67108864
    assert bench.C5_BOX == (1.0, 1.0, 8.0)       # the long axis is z, the slab axis


def test_defaults_are_the_baseline_workloads(bench, monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse_args()
    assert a.gpus == 1 and a.particles is None and a.scaling == "strong"
    assert a.steps >= 20 and a.warmup >= 5


def test_counter_profile_is_only_used_for_the_code_it_was_taken_on(bench, tmp_path, monkeypatch):
    from smoothed_particle_hydrodynamics_amd.build import source_hash
    n = 4 * 1024 * 1024
    good = {"particles": n, "csrc_sha16": source_hash(),
            "arithmetic": {"fast": {"density_plus_acceleration_hbm_bytes": 2.0e9,
                                    "valu_wave_instructions_per_launch_pair": 5.0e8}}}
    path = tmp_path / "counters.json"
    monkeypatch.setattr(bench, "PROFILE", str(path))
    prof, why = bench.kernel_counters(n)                 # no file
    assert prof is None and "no committed" in why
    path.write_text(json.dumps(good))
    prof, why = bench.kernel_counters(n)
    assert prof["density_plus_acceleration_hbm_bytes"] == 2.0e9 and source_hash() in why
    prof, why = bench.kernel_counters(n, "exact")        # counters of the other arithmetic only
    assert prof is None and "exact" in why
    prof, why = bench.kernel_counters(n // 4)            # another workload
    assert prof is None and "particles" in why
    path.write_text(json.dumps(dict(good, csrc_sha16="0123456789abcdef")))
    prof, why = bench.kernel_counters(n)                 # kernels changed since the profile
    assert prof is None and "re-run tools/profile_round.sh" in why


def test_committed_counter_profile_matches_the_committed_kernels(bench):
    """profiles/r4_kernel_counters.json and r4_valu_census.json must describe the kernel sources in
    this tree - otherwise the driver's bench line silently loses roofline.traffic / roofline.valu."""
    for arithmetic in ("fast", "exact"):
        prof, why = bench.kernel_counters(4 * 1024 * 1024, arithmetic)
        assert prof is not None, why
        assert 1.0e9 < prof["density_plus_acceleration_hbm_bytes"] < 4.0e9
        assert 2.0e8 < prof["valu_wave_instructions_per_launch_pair"] < 1.0e9
    valu = bench.valu_issue({}, 0.80, "fast")
    assert valu is not None and valu["frac"] is not None, valu
    assert 0.5 < valu["frac"] <= 1.0, valu["frac"]            # one number, and a bound
    assert sum(ph["wave_instructions_per_wave"] for ph in valu["by_phase"]) * 65536 == pytest.approx(
        valu["wave_instructions_per_launch_pair"], rel=1e-3)      # (phases are rounded to 0.1 in the line)


def test_gpus_n_without_world_size_becomes_the_launcher(bench, monkeypatch):
    """`python bench.py --gpus 8` the way the driver starts the N = 1 bench: a child process running
    torch.distributed.run on this very file with the same arguments, its exit code handed on - and
    nothing in the parent that could touch a GPU before that."""
    calls = []
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd: calls.append(cmd) or 7)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setitem(sys.modules, "torch", None)       # importing torch here would be a bug
    with pytest.raises(SystemExit) as exit_info:
        bench.main()
    assert exit_info.value.code == 7
    (cmd,) = calls
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"]


def test_self_launched_ranks_print_one_line(tmp_path):
    """end to end on CPU: bench.py --gpus 2 --dry-run starts two ranks through
    torch.distributed.run (gloo), they count themselves, rank 0 prints ONE JSON line"""
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    assert json.loads(lines[0]) == {"dry_run": True, "n_gpus": 2, "ranks": 2}


def test_control_plane_never_uses_the_exchange_group(bench, monkeypatch):
    """bench.py's barriers and reductions go over the default (gloo) group with host tensors; RCCL
    has a group of its own that only the transport is handed (a node whose
    device-to-device path is broken must still get through the host-staged fallback)."""
    import ast
    import inspect
    src = inspect.getsource(bench.run_slabs)
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute):
            if node.func.attr in ("all_reduce", "barrier", "all_gather", "broadcast"):
                assert not any(kw.arg == "group" for kw in node.keywords), ast.dump(node)
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "tensor":
            assert not any(kw.arg == "device" for kw in node.keywords), "control-plane tensors live on the host"
    assert "exchange_group" in src and "control_group=dist.group.WORLD" in src
    main_src = inspect.getsource(bench.main)
    assert 'init_process_group("gloo"' in main_src and 'new_group(backend="nccl")' in main_src
    # the transport is chosen (helper processes: native RCCL loop -> torch P2P -> host-staged) before
    # this process touches the GPU, the ranks agreeing over the gloo group between the stages
    assert main_src.index('init_process_group("gloo"') < main_src.index("choose_transport(rank, world") \
        < main_src.index("torch.cuda.set_device")
    choose = inspect.getsource(bench.choose_transport)
    assert choose.index('"native"') < choose.index('"torch"') and "host-fallback" in choose
    assert "ReduceOp.MIN" in choose and "group=" not in choose


def test_transport_choice_order_and_agreement(bench, monkeypatch):
    """native -> torch -> host-staged; a 'no' from any stage moves on; SPH_SLAB_TRANSPORT forces one"""
    import torch

    class FakeDist:
        class ReduceOp:
            MIN = "min"

        @staticmethod
        def all_reduce(t, op=None):
            return None

    for answers, want in (({"native": True, "torch": True}, "native"),
                          ({"native": False, "torch": True}, "torch"),
                          ({"native": False, "torch": False}, "host-fallback")):
        monkeypatch.delenv("SPH_SLAB_TRANSPORT", raising=False)
        monkeypatch.setattr(bench, "preflight_stage", lambda kind, rank, world, a=answers: a[kind])
        mode, said = bench.choose_transport(0, 2, torch, FakeDist)
        assert mode == want
        assert said["native"] == answers["native"]
    monkeypatch.setenv("SPH_SLAB_TRANSPORT", "host")
    assert bench.choose_transport(0, 2, torch, FakeDist)[0] == "host"


def test_source_hash_ignores_comments():
    """a reworded comment must not orphan a committed profile; a changed token must"""
    from smoothed_particle_hydrodynamics_amd.build import _code_only
    a = 'int a = 1; // one\n/* block\n   comment */ int b = 2;\nconst char* s = "// kept /* kept */";\n\n'
    b = 'int a = 1; // another wording\nint b = 2;   \nconst char* s = "// kept /* kept */";\n'
    assert _code_only(a).split() == _code_only(b).split()
    assert '"// kept /* kept */"' in _code_only(a)
    assert _code_only(a) != _code_only(a.replace("= 2", "= 3"))
