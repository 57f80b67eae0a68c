"""The final stdout line of bench.py stays small enough for the driver to see whole (round 5's 33.6 KB line came back `parsed: null`)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic")


def canned(bloat=1):
    """A detail record of the shape main() builds, with the prose and the per-kernel tables blown up `bloat` times."""
    per_kernel = {f"k_kernel_{i}<{j}>": {"stage": "seed", "ms_live": 1.0, "hbm_bytes_per_launch": 10**10, "frac_of_peak": 0.3, "note": "x" * 80} for i in range(35 * bloat) for j in range(2)}
    return {
        "metric": "reads/sec (150 bp PE vs GRCh38) at 1/2/4/8 MI355X; SAM CIGAR bit-exact", "value": 5.1e8, "unit": "reads/s", "n_gpus": 1, "steps": 20, "warmup": 5,
        "ms_per_step": 15.6, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64/int32", "dtype_note": "n" * 300 * bloat, "data": "synthetic",
        "config": {"workload": "synthetic GRCh38-sized genome, 3100 Mbp (24 contigs, " + "landscape; " * 60 * bloat + "; GRCh38 itself is unavailable offline), 4000000 pairs x 150 bp PE per step per GPU, -alg ksw2",
                   "reads_per_step_per_gpu": 8_000_000, "index_hbm_gb": 109.4, "multi_gpu": "m" * 900 * bloat, "multi_gpu_host_ms_per_step": 0.2},
        "roofline": {"bound": "hbm", "kernel": "k_seed", "achieved": 2539.6, "peak": 8000.0, "unit": "GB/s", "frac": 0.3174, "traffic": 10672269947, "traffic_from": "profiles/round6/summary_human.json",
                     "avg_launch_ms": 4.2, "algorithmic_bytes_per_launch": 2819779512, "algorithmic_frac": 0.0839, "basis": "b" * 500 * bloat,
                     "per_kernel": per_kernel, "request_rate": per_kernel, "speed_of_light_equiv": {"note": "s" * 400}, "path": {"note": "p" * 400}},
        "per_read": {"fm_ext_steps": 148.0, "sa_hits": 2.5}, "stage_ms_per_step": {"seed": 4.2, "cluster": 1.6, "build": 3.4, "dp": 1.7, "finish": 1.3, "total": 15.6},
        "value_pcie_inclusive": {"value": 4.6e8, "unit": "reads/s", "steps": 36, "ms_per_step": 17.2, "note": "n" * 900 * bloat, "system_runtime": {"value": 4.7e8, "ms_per_step": 16.7, "note": "n" * 700}},
        "value_file_to_file": {"value": 1.3e7, "unit": "reads/s", "reads": 16_000_000, "seconds": 1.2, "without_sam_output": {"value": 5.2e7, "note": "n" * 300}, "note": "n" * 400},
        "vcf_reduce": {"profile_batch_ms": 28.8, "same_batches_without_profile_ms": 16.9, "hbm_free_gb": 5.4, "reduce_ms": 0.04, "reduce_gb": 68.2, "sparse_records": 10**7, "note": "n" * 600,
                       "call_variants": {"ms_total": 734.2}},
        "cpu_baseline": {"value": 282352.9, "unit": "reads/s", "cores": 16, "hardware_threads": 256, "kind": "reference", "sample": "s" * 400 * bloat, "note": "n" * 500,
                         "mapping_only": {"value": 282352.9, "cores": 16, "sample": "s" * 300}, "single_thread": {"value": 21052.6, "cores": 1, "sample": "s" * 200}},
        "other_genome": {"genome": "uniform", "value": 8.7e8, "ms_per_step": 9.1, "workload": "w" * 600, "roofline": {"basis": "b" * 300}},
        "other_configs": [{"config": "config 5: " + "c" * 100 * bloat, "value": 6e7, "unit": "reads/s", "ms_per_step": 132.0, "workload": "w" * 700,
                           "roofline": {"bound": "valu", "frac": 0.29, "achieved": 23.0, "peak": 78.6, "unit": "T lane-ops/s", "basis": "b" * 400, "kernel": "k" * 200},
                           "cpu_baseline": {"value": 66666.7, "cores": 16, "kind": "reference", "sample": "s" * 300},
                           "stage_ms_per_step": {"seed": 15.0, "dp": 51.0, "build": 27.0}},
                          {"config": "config 2: e. coli", "error": "e" * 200}],
    }


@pytest.mark.parametrize("bloat", [1, 10])
def test_final_line_is_small_and_complete(bloat):
    import bench
    out = canned(bloat)
    assert len(json.dumps(out)) > 30_000  # (the detail record is what round 5 printed)
    line = bench.compact_line(out, "gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text) < 6000 and "\n" not in text
    for k in REQUIRED:
        assert k in line, k
    for k in ROOFLINE:
        assert k in line["roofline"], k
    assert line["roofline"]["frac"] == out["roofline"]["frac"] and line["roofline"]["traffic"] == out["roofline"]["traffic"]
    assert line["value"] == out["value"] and line["ms_per_step"] == out["ms_per_step"]
    for k in ("value", "unit", "cores", "kind"):
        assert line["cpu_baseline"][k] == out["cpu_baseline"][k]
    assert line["cpu_baseline"]["single_thread"]["value"] == 21052.6
    assert line["value_pcie_inclusive"]["value"] == 4.6e8 and line["value_file_to_file"]["value"] == 1.3e7
    assert line["detail"] == "gpurun_out/bench_detail.json"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert "per_kernel" not in line["roofline"] and "request_rate" not in line["roofline"]


def test_final_line_of_a_multi_gpu_run_without_the_single_gpu_legs():
    import bench
    out = canned()
    for k in ("cpu_baseline", "value_file_to_file", "other_genome", "other_configs"):
        out.pop(k)
    out["n_gpus"] = 8
    line = bench.compact_line(out, None)
    assert len(json.dumps(line)) < 6000 and line["n_gpus"] == 8 and line["config"]["multi_gpu"] and "detail" not in line


def test_the_committed_round5_record_compacts():
    """The record that did not parse in round 5, through the same function."""
    import bench
    p = os.path.join(ROOT, "profiles", "round5", "bench_default.json")
    out = json.load(open(p))
    assert os.path.getsize(p) > 30_000
    line = bench.compact_line(out, "x")
    assert len(json.dumps(line)) < 6000
    assert line["roofline"]["kernel"] == "k_seed" and line["cpu_baseline"]["kind"] == "reference"


def test_emit_prints_one_json_line_last(tmp_path, capsys):
    import argparse
    import bench
    args = argparse.Namespace(detail_dir=str(tmp_path), detail_tag="", detail_stdout=0, full_line=0)
    bench.emit(canned(), args)
    lines = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 6000
    line = json.loads(lines[0])
    detail = json.load(open(os.path.join(str(tmp_path), "bench_detail.json")))
    assert "per_kernel" in detail["roofline"] and line["value"] == detail["value"]
