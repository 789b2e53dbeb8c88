"""tools/roofline.py: the committed roofline JSON is what `rebuild` makes of the committed raw rocprofv3 CSVs, and the in-run
form bench.py uses (no kernel-trace pass: kernel time from the dispatch timestamps of the SQ counter pass) gives the same
counter-derived figures."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import roofline  # noqa: E402

RAW = os.path.join(ROOT, "profiles", "raw_r06")
COMMITTED = os.path.join(ROOT, "profiles", "r06_roofline_bench.json")


def test_committed_roofline_json_is_reproduced_from_the_raw_csvs(tmp_path):
    d = roofline.rebuild(RAW, str(tmp_path / "out.json"))
    with open(COMMITTED) as fh:
        c = json.load(fh)
    for k in ("traffic_bytes_per_launch", "valu_instructions_per_launch", "valu_busy_fraction", "kernel_avg_ms_kernel_trace",
              "wasted_traffic_ratio", "algorithmic_bytes_per_launch"):
        assert d[k] == c[k], k
    # sanity of the figures themselves: bytes per solve as SURVEY.md 8(d) states them, a busy fraction below 1
    assert c["algorithmic_bytes_per_launch"] == 21944 * c["workload"]["batch_per_gpu"]
    assert 0.5 < c["valu_busy_fraction"] < 1.0 and c["traffic_bytes_per_launch"] > c["algorithmic_bytes_per_launch"]


def test_in_run_form_without_a_kernel_trace_pass(tmp_path):
    raw = tmp_path / "raw"
    raw.mkdir()
    for f in os.listdir(RAW):
        if f != "kt_kernel_stats.csv":
            shutil.copy(os.path.join(RAW, f), raw / f)
    d = roofline.rebuild(str(raw), str(tmp_path / "out.json"))
    with open(COMMITTED) as fh:
        c = json.load(fh)
    assert d["kernel_time_source"].startswith("dispatch timestamps")
    # counters are the same files; only the kernel duration comes from another pass (counter passes run ~1-3 % slower)
    assert d["valu_instructions_per_launch"] == c["valu_instructions_per_launch"]
    assert d["traffic_bytes_per_launch"] == c["traffic_bytes_per_launch"]
    assert abs(d["valu_busy_fraction"] - c["valu_busy_fraction"]) < 1e-12          # cycles only
    assert abs(d["kernel_avg_ms_kernel_trace"] / c["kernel_avg_ms_kernel_trace"] - 1.0) < 0.05
    assert roofline.load_pmc_for(str(tmp_path / "out.json"), 20, 8, c["workload"]["batch_per_gpu"]) is not None
    assert roofline.load_pmc_for(str(tmp_path / "out.json"), 40, 8, c["workload"]["batch_per_gpu"]) is None


def test_the_tail_kernel_is_kept_apart_from_the_dominant_kernel():
    """Since round 5 a launch is two kernels -- solve_kernel_pair and the continuation solve_kernel_team: the roofline figures
    are those of the pair kernel alone, the continuation's average duration is reported next to them."""
    with open(COMMITTED) as fh:
        c = json.load(fh)
    # one continuation launch per solve call, or two (the one that runs while the throughput launch drains + the sweep behind it)
    assert c["tail_kernel_calls"] in (c["kernel_calls_kernel_trace"], 2 * c["kernel_calls_kernel_trace"])
    assert 0.0 < c["tail_kernel_ms_per_solve_call"] < 0.05 * c["kernel_avg_ms_kernel_trace"]
    with open(os.path.join(ROOT, "profiles", "r06_bench_line.json")) as fh:
        line = json.load(fh)
    ro = line["roofline"]
    # HIP events vs kernel trace, for the throughput kernel alone ...
    assert ro["kernel"] == "solve_kernel_pair" and abs(ro["throughput_kernel_ms"] / c["kernel_avg_ms_kernel_trace"] - 1.0) < 0.01
    # ... while `achieved` / `frac` / flops divide by the WHOLE solve of the batch on the launch stream (round 6: the numerators count every
    # problem, the promoted ones included): throughput kernel + what is left of the continuation behind it
    assert abs(ro["kernel_ms"] - (ro["throughput_kernel_ms"] + ro["tail_kernel_ms"])) < 1e-9
    assert abs(ro["achieved"] - 21944 * 131072 / (ro["kernel_ms"] * 1e-3) / 1e9) < 1e-9
    assert ro["kernel_ms"] <= line["ms_per_step"] * 1.001
