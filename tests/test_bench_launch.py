"""bench.py starts its own ranks: a bare `python3 bench.py --gpus N` must launch N fresh processes (one per GPU, through
torch.distributed.run on 127.0.0.1), forward rank 0's single JSON line and leave with the children's status.

CPU: the launcher, the band split, the barrier-bracketed loop and the gather are rehearsed over gloo with `--dry-run` (no
kernels, value null).  GPU: the real N = 2 code path with both ranks on the one card (gloo stands in for RCCL, which cannot
put two ranks on one device), including the gathered frame's comparison with the single-GPU frame and BASELINE config 5."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def test_self_launch_two_ranks_over_gloo():
    p, lines = _run(["--gpus", "2", "--dry-run", "--steps", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["value"] is None
    assert out["gather_verified"] is True
    assert sum(out["rows_per_rank"]) == 1080 and max(out["rows_per_rank"]) - min(out["rows_per_rank"]) <= out["band_rows"]


def test_self_launch_three_ranks_band_balance():
    p, lines = _run(["--gpus", "3", "--dry-run", "--steps", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["gather_verified"] is True and sum(out["rows_per_rank"]) == 1080


def test_single_process_dry_run_and_world_mismatch():
    p, lines = _run(["--dry-run", "--steps", "1"])
    assert p.returncode == 0 and len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 1
    # a rank environment that does not match --gpus is an error, not a silent single-GPU run
    p, lines = _run(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and not lines


@pytest.mark.gpu
def test_bare_bench_gpus_2_on_one_card():
    p, lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra"],
                    {"NTRACER_BENCH_BACKEND": "gloo", "NTRACER_BENCH_DEVICE": "0"}, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["delivery"]["verified_equal_to_single_gpu_frame"] is True
    assert out["config5"]["n_gpus"] == 2 and out["config5"]["gather_verified_equal_to_whole_frame"] is True and out["config5"]["value"] > 0


def test_committed_profile_gives_an_issue_bound_below_one():
    """bench.py's `roofline.valu` comes from the committed rocprofv3 passes (profiles/r03_pmc_summary.json): instruction counts
    by class against the kernel's own cycle count.  Whatever the pricing of the mixed classes, the fraction has to be a
    fraction -- round 2's busy figure read 1.04."""
    sys.path.insert(0, ROOT)
    import bench
    prof = bench.load_profile()
    assert prof is not None and prof["headline_call"]["frames_per_call"] == 160
    for key in ("headline_call", "rgbf32_call", "config4_call", "band8_call", "band8_overlapped_call"):
        v = bench.valu_bound(prof[key], 100.0)
        assert v is not None, key
        assert 0.0 < v["frac_lo"] <= v["frac"] <= v["frac_hi"] <= 1.0, (key, v["frac_lo"], v["frac_hi"])
        assert sum(v["by_class"].values()) == pytest.approx(v["valu_wave_insts"], rel=1e-6)
    hc = prof["headline_call"]
    # HBM traffic of the headline call = the framebuffers, to within a percent (no wasted re-reads)
    assert hc["write_bytes_per_call"] + hc["fetch_bytes_per_call_corrected"] < 1.01 * hc["algorithmic_bytes_per_call"]
