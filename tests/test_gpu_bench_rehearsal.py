"""bench.py's multi-rank path, end to end, on the one GPU of the test box.

`python bench.py --gpus 2` the way the driver starts the N = 1 bench (no WORLD_SIZE): the script
launches its own two ranks (torch.distributed.run as a child process), each rank asks a helper
process whether the device-to-device exchange works - with both ranks on one device RCCL refuses
("Duplicate GPU detected"), which is this rehearsal's stand-in for a node where P2P is broken - the
ranks agree over gloo to stage their halo messages through the host, step the strong-scaling
workload (and the side workload, and rank 0's single-context reference) and rank 0 prints ONE JSON
line.  What a real node adds is RCCL between distinct devices; everything else is exercised here."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_launches_itself_and_falls_back_to_host_staging(hiplib):
    env = dict(os.environ, SPH_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "SPH_SLAB_TRANSPORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--particles", "1048576",
                          "--steps", "6", "--warmup", "3", "--other-particles", "524288"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["scaling"] == "strong"
    assert d["config"]["particles"] == 1048576 and d["config"]["arithmetic"] == "fast"
    assert "HOST-STAGED FALLBACK" in d["config"]["parallelism"]          # the pre-flight said no
    assert "pre-flight helper" in out.stderr                             # ... and said why
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
    assert d["strong_scaling"]["gpus"] == 2 and d["strong_scaling"]["one_gpu_ms_per_step"] > 0
    assert d["other_scaling"]["particles"] == 524288 and d["other_scaling"]["ranks"] == 2
    assert d["config"]["halo"]["bytes_per_message"] <= d["config"]["halo"]["bytes_allocated"]
