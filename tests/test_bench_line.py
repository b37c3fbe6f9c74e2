"""bench.py's record must be ONE short JSON line: the driver keeps only the tail of stdout (round 2's 20 KB line lost its
head -- value, roofline, cpu_baseline -- and the round was recorded as unmeasured)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _canned(n_other):
    roof = {"bound": "mfma", "kernel": "patch_gemm_kernel (gemm_patch.hip: conv forward + data gradient, bf16)", "achieved": 926.91,
            "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.3707, "traffic": 0.706, "traffic_unit": "GB/launch (rocprofv3 PMC, profiles/)",
            "launches": 240, "avg_launch_ms": 0.2626, "gflop_per_launch": 243.403, "share_of_step": 0.37, "step_frac": 0.2246}
    cpu = {"value": 4021.3, "unit": "mel-frames/s", "cores": 16, "kind": "port",
           "sample": "oracle train_step, 4 clips x 80x1024, 1 warm-up + 3 timed steps (1018 ms/step)"}
    other = {"dtype": "f32", "value": 2741234.5, "ms_per_step": 47.812, "frac": 0.7912, "step_frac": 0.6353}
    oc = [{"workload": "configs[3] D=256 K=8192", "dtype": "bf16", "clips": 16, "value": 2775123.4, "ms_per_step": 5.904, "step_frac": 0.1813,
           "vq_frac": 0.1234, "vq_share": 0.1899}] * n_other
    return dict(value=15401234.5, ms_per_step=8.511, world=1, steps=20, warmup=5, dtype="bf16", D=128, K=512, B=128, T=1024,
                roof=roof, cpu=cpu, other=other, other_configs=oc)


def test_line_is_short_and_complete():
    text = bench.build_line(**_canned(4))
    assert "\n" not in text and len(text) < 3072, len(text)
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "other_mode", "other_configs"):
        assert key in line, key
    assert line["vs_baseline"] is None and line["scaling"] == "weak" and line["higher_is_better"] is True
    for key in ("workload", "clips_per_gpu", "global_batch", "frames", "parallelism"):
        assert key in line["config"], key
    assert "model" not in line["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"], key
    assert len(line["other_configs"]) == 4


def test_line_sheds_the_optional_part_rather_than_grow():
    text = bench.build_line(**_canned(40))
    assert len(text) < 3072
    line = json.loads(text)
    assert line["other_configs"] is None and line["roofline"]["frac"] == 0.3707 and line["cpu_baseline"]["cores"] == 16


def test_kernel_tables_go_to_a_side_file_not_stdout(tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    path = bench.write_kernel_tables({"configs[1] bf16 B=128": [{"kernel": "patch_gemm 3x3", "us_per_launch": 186.7, "frac": 0.41}]})
    assert os.path.dirname(path) == str(tmp_path) and json.load(open(path))
    out = capsys.readouterr()
    assert out.out == "" and "patch_gemm 3x3" in out.err
