"""bench.py prints ONE JSON line with the fields the driver reads (GPU box)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_json_contract(gpu):
    r = _run_bench()
    sm = r["per_step_ms"]["summary"]
    if sm["backward"]["max"] > 1.5 * sm["backward"]["median"] or sm["forward"]["max"] > 2.0 * sm["forward"]["median"]:
        # one of this run's THREE timed steps stalled (seen once in r05: a 20 ms step among twenty of 0.8 ms on an
        # otherwise normal box -- the line shows it, which is what its per-step arrays are for): the spread assertions
        # below are about the settled state, so look once more before calling it a failure
        print("a timed step stalled:", r["per_step_ms"]["forward"], r["per_step_ms"]["backward"], "-- running the bench again")
        r = _run_bench()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["unit"] == "Mrays/s" and r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["dtype"] == "f32" and r["data"] == "synthetic" and "workload" in r["config"]
    assert abs(r["value"] - 0.64 / r["ms_per_step"] * 1e3) / r["value"] < 1e-3          # 640 000 rays per step
    rf = r["roofline"]
    # `bound` names what binds the dominant group by the line's own `limits`; the HBM fraction stays as the secondary number
    assert rf["bound"] in ("latency", "valu", "hbm") and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and "hbm" in rf["frac_of"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    dom = "forward" if r["kernel_ms"]["forward"] >= r["kernel_ms"]["backward"] else "backward"
    lim = r["limits"][dom]
    if rf["bound"] == "latency" and lim.get("measured_ms") is not None:   # consistent with limits: the floor lies below the measurement
        assert lim["floor_ms"] < lim["measured_ms"] and rf["frac"] < 0.5
    # SURVEY 8(d)'s figure for the whole step is carried, and the drop-in route's number stands beside the headline
    ref = rf["reference_equivalent"]
    assert abs(ref["gbps"] - ref["bytes_per_step"] / (r["ms_per_step"] * 1e-3) / 1e9) < 0.01 * ref["gbps"]
    assert 4.9e9 < ref["bytes_per_step"] < 5.2e9                                        # 1.225 GB forward + 3.811 GB backward
    # (against the better of the wall-clock and the event figure: over this test's 3 timed steps the wall clock carries the
    # first forward's launch gap; the plain route, recognised as an image since r05, is within a few per cent of the hinted one)
    assert 0 < r["value_plain"] <= max(r["value"], r["value_median"]) * 1.05
    # the line says what happened step by step: the K intervals, their spread, the median's value,
    # and what the untimed warm-up saw -- by time and convergence, not by count
    ps = r["per_step_ms"]
    assert len(ps["forward"]) == len(ps["backward"]) == r["steps"] and min(ps["forward"] + ps["backward"]) > 0
    for g in ("forward", "backward", "step"):
        sm = ps["summary"][g]
        assert sm["min"] <= sm["median"] <= sm["max"]
    assert abs(r["kernel_ms"]["backward"] - sum(ps["backward"]) / r["steps"]) < 2e-3
    assert ps["summary"]["backward"]["max"] <= 1.5 * ps["summary"]["backward"]["median"], ps["backward"]
    # (the first timed forward follows the barrier + synchronize: its host launch gap is in its interval)
    assert ps["summary"]["forward"]["max"] <= 2.0 * ps["summary"]["forward"]["median"], ps["forward"]
    assert abs(r["value_median"] - 0.64 / ps["summary"]["step"]["median"] * 1e3) / r["value_median"] < 2e-3
    assert r["value_median"] >= 0.9 * r["value"]                 # events on the stream never see more than the wall clock
    pw = r["prewarm"]
    assert pw["seconds"] >= 1.0 and pw["steps"] >= 20 and pw["converged"] is True, pw
    assert pw["last_batch_ms"]["backward"] <= 1.1 * ps["summary"]["backward"]["median"]    # the timed steps are the settled ones
    # the default run carries BASELINE.json's other single-GPU configs along, briefly (the driver then holds numbers for them too)
    oc = r["other_configs"]
    assert len(oc) == 6 and all("error" not in o and o["value"] > 0 for o in oc), oc
    assert any("configs[3]" in o["config"] and "forward+backward" in o["config"] and o["value"] > 300 for o in oc)
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert r["counters"]["steps"] == 18919396                                           # the workload is the one named


def test_a_stalled_step_is_timed_again_and_said_so(gpu):
    """The host falls asleep for 30 ms inside the timed steps (BENCH_TEST_STALL_MS: what a stalled step looks like from
    the stream's events): the K steps are timed once more, the line carries both runs, `value` is the settled one."""
    env = dict(os.environ, BENCH_TEST_STALL_MS="30")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "1", "--no-cpu-baseline",
                        "--no-plain", "--no-other-configs"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    rt = r["retimed"]
    assert rt is not None and len(rt["first_run"]["forward"]) == 6
    assert max(rt["first_run"]["forward"] + rt["first_run"]["backward"]) > 20.0          # the sleep, in the first run
    sm = r["per_step_ms"]["summary"]
    assert sm["step"]["max"] < 4.0 * sm["step"]["median"] and r["ms_per_step"] < 0.25 * rt["first_run"]["ms_per_step"]
    p2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "1", "--no-cpu-baseline",
                         "--no-plain", "--no-other-configs"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert json.loads([l for l in p2.stdout.splitlines() if l.strip()][-1])["retimed"] is None
