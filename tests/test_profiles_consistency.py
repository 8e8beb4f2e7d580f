"""The committed bench line, the committed rocprofv3 kernel statistics and the committed PMC summary must describe the
same build: the dominant kernel's live duration (HIP events inside bench.py), its average in `--kernel-trace --stats`
and the duration in the PMC summary that `roofline.traffic` is taken from agree within 15 %, and the line's traffic is
what bench.pmc_traffic derives from that summary.  (A refreshed kernel without refreshed profiles fails here, on CPU.)"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _latest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    assert files, pattern
    return files[-1]


def test_bench_line_profile_and_pmc_summary_agree():
    import bench
    line = json.loads(open(_latest("r[0-9][0-9]_chr20_bench.json")).read().strip().splitlines()[-1])
    roof = line["roofline"]
    family, live_ms = roof["kernel"], roof["avg_launch_ms"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    kernel = bench.KERNEL_OF_FAMILY[family]
    # rocprofv3 --kernel-trace --stats of the same command
    stats = _latest("r[0-9][0-9]_chr20_kernel_stats.csv")
    rows = [r for r in csv.DictReader(open(stats)) if kernel in r["Name"]]
    assert rows, (kernel, stats)
    prof_ms = float(rows[0]["AverageNs"]) / 1e6
    assert abs(prof_ms - live_ms) <= 0.15 * live_ms, (prof_ms, live_ms)
    # the PMC summary the traffic figure comes from
    got = bench.pmc_traffic(family, live_ms)
    assert got["traffic"] is not None, got
    assert got["traffic"] == roof["traffic"], (got, roof["traffic"])
    # counters cannot be below what the kernel must move as its data is laid out (SURVEY 8(d)'s figure is the REFERENCE's record
    # traffic for the stage: a kernel that moves super-k-mers instead of k-mers stays below it)
    assert roof["traffic"] >= roof["kernel_layout_bytes_per_launch"] * 0.9
    # and a kernel that changed since makes the figure disappear instead of going stale
    assert bench.pmc_traffic(family, live_ms * 1.3)["traffic"] is None


def test_bench_line_carries_the_contract_fields():
    line = json.loads(open(_latest("r[0-9][0-9]_chr20_bench.json")).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and "workload" in line["config"] and "model" not in line["config"]
    cb = line["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    # BASELINE.json's metric: "Gbases/s seqset build ..." (absolute throughput of the build)
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert baseline["metric"].startswith("Gbases/s seqset build")
    assert line["unit"] == "Gbases/s" and "seqset_build" in line["metric"] and line["higher_is_better"] is True
    assert abs(line["value"] - line["config"]["bases_per_gpu"] / (line["ms_per_step"] * 1e-3) / 1e9) < 0.01 * line["value"]
