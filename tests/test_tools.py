"""CPU suite: the measurement tools and driver hooks at least parse and expose their command lines (their runs
need the GPU box)."""
import ast
import glob
import os
import subprocess
import sys

from conftest import ROOT


def test_tool_sources_parse():
    files = glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "sweeps", "*.py")) + \
        [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    assert len(files) >= 10
    for f in files:
        ast.parse(open(f).read(), filename=f)


def test_bench_command_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def test_make_profiles_script_is_valid_shell():
    r = subprocess.run(["bash", "-n", os.path.join(ROOT, "tools", "make_profiles.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_bench_line_stays_under_the_drivers_stdout_window(tmp_path, monkeypatch, capsys):
    """BENCH_r02.parsed was null because the line was ~20 KB and the driver keeps 8 KB.  A full
    line with worst-case field widths must stay below 4 KB; the per-kernel tables go to the detail file."""
    import json

    import bench

    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    kernel = "igemm_x3_wgrad_halo_kernel<128, 2, 2, 4, 2, true, 3>  [clone .kd] " + "x" * 40
    roof = {"bound": "mfma", "achieved": 407.123, "peak": 833.3, "unit": "TFLOP/s", "frac": 0.4886,
            "traffic": 323456789.0, "kernel": kernel, "launches_per_step": 15, "avg_launch_us": 166.12,
            "algorithmic_per_launch": 67.6712, "algorithmic_unit": "GFLOP", "share_of_kernel_time": 0.1034,
            "kernel_time_ms_per_step": 24.512, "traffic_source": "profiles/pmc_traffic.json",
            "contractions": {"tflops": 199.12, "ms_per_step": 18.912, "splitk_finish_ms": 1.612, "gflop_per_step": 4875.1}}
    out = {"metric": "images/sec training, R50-FPN JTSM panoptic, 2x1024x1024, 1/2/4/8 GPU", "value": 12345.678,
           "unit": "images/sec", "n_gpus": 8, "steps": 1000, "warmup": 100, "ms_per_step": 1234.567,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16x3", "data": "synthetic",
           "config": {"workload": "w" * 330, "global_batch": 16, "parallelism": "dp8", "math": "m" * 120,
                      "final_loss": 123.45678, "lr": 1e-7, "foreground_rois_last_step": 12345,
                      "detail": bench.DETAIL_PATH},
           "roofline": roof,
           "exact_fp32": {"value": 32.612, "unit": "images/sec", "ms_per_step": 61.312, "dtype": "f32"},
           "round1_workload": {"value": 99.512, "unit": "images/sec", "ms_per_step": 20.112, "dtype": "bf16x3",
                               "foreground_rois_last_step": 7},
           "config4_fp16": {"value": 62.812, "unit": "images/sec", "ms_per_step": 31.912, "dtype": "f16", "steps": 10,
                            "foreground_rois_last_step": 300, "roofline": dict(roof, contractions=None)},
           "cpu_baseline": {"value": 0.0847, "unit": "images/sec", "cores": 128, "kind": "port", "sample": "s" * 200}}
    detail = {"roofline": {"rooflines_over_2pct": [dict(roof) for _ in range(20)]}}
    bench.emit(out, detail)
    line = capsys.readouterr().out.strip().splitlines()
    assert len(line) == 1 and len(line[0]) < 4096, len(line[0])
    parsed = json.loads(line[0])
    for k in ("metric", "value", "unit", "n_gpus", "ms_per_step", "config", "roofline", "cpu_baseline", "dtype"):
        assert k in parsed
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in parsed["roofline"]
    saved = json.load(open(tmp_path / bench.DETAIL_PATH))
    assert len(saved["detail"]["roofline"]["rooflines_over_2pct"]) == 20
    # an oversized optional leg is moved to the detail file rather than breaking the contract
    out["exact_fp32"]["note"] = "n" * 5000
    bench.emit(out, detail)
    line = capsys.readouterr().out.strip()
    assert len(line) < 4096 and "cpu_baseline" in json.loads(line) and "roofline" in json.loads(line)
