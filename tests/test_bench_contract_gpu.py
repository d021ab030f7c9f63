"""The driver's contract on `bench.py`: ONE line on stdout, JSON, with the keys the driver and the judge read — checked on a
small instance of the same code path (levels 5..9 instead of 7..12, a population of eight at level 9) so that it takes seconds.
Everything else a library prints (RCCL's banner when its first communicator comes up) must not reach stdout."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--max-level", "9", "--min-level", "5",
           "--population", "8", "--population-level", "9", "--throughput-workers", "3", "--no-configs", "--no-reference-size",
           "--cpu-threads", "4"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, proc.stdout[:2000]                 # one line, nothing else
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["higher_is_better"] is True
    assert line["unit"] == "solves/s" and line["value"] > 0 and line["dtype"] == "f64" and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and "workload" in line["config"] and "model" not in line["config"]
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu.get("cpu_model") and cpu.get("nproc")
    # the headline CPU figure is the tuned OpenMP solver, the bit-comparable parity oracle sits one level down (round 6)
    assert cpu["parity_oracle"]["kind"] == "port" and cpu["parity_oracle"]["value"] > 0 and cpu["value"] > 0
    # the metric with the GPU full is repeated at the head of the line
    assert line["value_in_flight"] == line["throughput_mode"]["value"]
    assert abs(line["roofline_in_flight"]["frac"] - line["throughput_mode"]["roofline"]["frac"]) < 1e-15
    # the collectives ran through a (one-rank) RCCL group, and the population section carries its modes
    assert line["config"]["collectives"]["process_group"] is True and line["config"]["collectives"]["n_ranks_seen"] == 1
    pop = line["population"]
    assert pop["static"]["candidates_per_rank"] == [8] and pop["static_in_flight"]["rho_and_iterations_identical_to_sequential"] is True
    assert "time_objective_vs_sequential" in pop["static_in_flight_isolated_time"]
