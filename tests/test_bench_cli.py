"""bench.py as the driver runs it. CPU: a command that names more GPUs than there are must FAIL (non-zero status, a JSON line
with the reason), never measure fewer; a launcher world that disagrees with --gpus likewise. GPU: the N > 1 code path of
`python3 bench.py --gpus N` - N thread ranks of one process on the library's in-process transport, the form the driver's
launcher-less call takes - runs with 2 ranks rehearsed on the one device and its line carries what the judge asks for."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, timeout=timeout, env=e)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_more_gpus_than_devices_is_an_error_not_a_smaller_measurement():
    """`python3 bench.py --gpus 8` on a box with fewer than 8 devices (none here, one on the GPU test box)"""
    r, line = _bench("--gpus", "8", "--steps", "1", "--warmup", "1")
    assert r.returncode != 0, "bench.py --gpus 8 must not succeed on a box with fewer than 8 devices"
    assert "needs 8 HIP devices" in r.stderr
    assert line is not None and line["n_gpus"] == 8 and line["value"] is None and "needs 8 HIP devices" in line["error"]


def test_launcher_world_must_agree_with_gpus():
    r, line = _bench("--gpus", "8", env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must agree" in r.stderr and line["value"] is None


def test_process_per_gpu_transports_need_a_launcher():
    r, line = _bench("--gpus", "2", "--transport", "rccl")
    assert r.returncode != 0 and "torch.distributed.run" in r.stderr and line["error"]


@pytest.mark.gpu
def test_bench_thread_ranks_two_ranks_on_one_device():
    """the launcher-less N > 1 path with 2 thread ranks sharing the test box's one device (--share-devices: a labelled rehearsal)"""
    r, line = _bench("--gpus", "2", "--share-devices", "--log-adds", "12", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert r.returncode == 0, r.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["ms_per_step"] > 0 and "error" not in line
    cfg = line["config"]
    assert cfg["ranks"]["n"] == 2 and cfg["ranks"]["devices"] == [0, 0] and cfg["ranks"]["ranks_share_devices"] is True
    assert cfg["ranks"]["transport"].startswith("local") and cfg["ranks"]["rccl_world"] is None
    assert cfg["preflight"]["ok"] is True and cfg["verified"] is True
    assert cfg["bytes_exchanged_per_rank_per_proof"] > 0 and cfg["rows_per_proof"] == 256 + 2 * 4096
    assert len(cfg["step_ms_min_median_max"]) == 3
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["achieved"] > 0
    assert line["replicas"]["proofs_per_step"] == 2 and line["replicas"]["value"] > 0


@pytest.mark.gpu
def test_bench_refuses_two_gpus_on_a_one_gpu_box(pkg):
    if pkg.device_count() != 1:
        pytest.skip("this box shows %d devices" % pkg.device_count())
    r, line = _bench("--gpus", "2", "--steps", "1", "--warmup", "1")
    assert r.returncode != 0 and "needs 2 HIP devices, this process sees 1" in r.stderr and line["value"] is None
