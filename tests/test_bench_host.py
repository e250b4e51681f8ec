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
            "density_plus_acceleration_hbm_bytes": 2.0e9, "valu_wave_instructions_per_launch_pair": 5.0e8}
    path = tmp_path / "counters.json"
    monkeypatch.setattr(bench, "PROFILE", str(path))
    prof, why = bench.kernel_counters(n)                 # no file
    assert prof is None and "no committed" in why
    path.write_text(json.dumps(good))
    prof, why = bench.kernel_counters(n)
    assert prof["density_plus_acceleration_hbm_bytes"] == 2.0e9 and source_hash() in why
    prof, why = bench.kernel_counters(n // 4)            # another workload
    assert prof is None and "particles" in why
    path.write_text(json.dumps(dict(good, csrc_sha16="0123456789abcdef")))
    prof, why = bench.kernel_counters(n)                 # kernels changed since the profile
    assert prof is None and "re-run tools/profile_round.sh" in why


def test_committed_counter_profile_matches_the_committed_kernels(bench):
    """profiles/r2_kernel_counters.json must describe the kernel sources in this tree - otherwise
    the driver's bench line silently loses roofline.traffic."""
    prof, why = bench.kernel_counters(4 * 1024 * 1024)
    assert prof is not None, why
    assert 1.0e9 < prof["density_plus_acceleration_hbm_bytes"] < 4.0e9
    assert 3.0e8 < prof["valu_wave_instructions_per_launch_pair"] < 1.0e9
