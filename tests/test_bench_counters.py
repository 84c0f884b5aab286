"""CPU: bench.py quotes a committed counter run (profiles/pmc_latest.json) only for the build and the kernel instantiation it
was taken with.  bench.main() runs against the stand-in backend (tests/bench_standin.py: no device, no
product code) with a doctored counter file: the right build id and kernel -> the figures are quoted; another build id, or
another kernel instantiation -> traffic / issue / counter_* are null and traffic_source says "stale: ..."."""

import contextlib
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _entry(build_id, kernel, prologue="prologue_kernel<1, 1>"):
    return {"DHGR": {"build_id": build_id, "kernel": kernel, "prologue_kernel": prologue, "streams": 14336, "bench_args": ["--steps", "2"],
                     "greedy_hbm_bytes_per_launch_per_stream": 225000.0, "prologue_hbm_bytes_per_launch_per_stream": 109000.0,
                     "issue": {"kernel": kernel, "valu_per_opcode_wave": 161.0, "issue_floor_frac": 0.58}}}


def _line(tmp_path, monkeypatch, doc):
    import bench
    p = tmp_path / "pmc_latest.json"
    p.write_text(json.dumps(doc))
    monkeypatch.setenv("IIV_PMC_LATEST", str(p))
    monkeypatch.setenv("IIV_STANDIN_BUILD_ID", "abcdef012345")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = bench.main(["--backend", "bench_standin:CpuStandIn", "--steps", "2", "--warmup", "1", "--frames-per-step", "4", "--streams", "64"])
    return out


def test_counters_of_this_build_and_kernel_are_quoted(tmp_path, monkeypatch):
    d = _line(tmp_path, monkeypatch, _entry("abcdef012345", "greedy_wave_kernel<1, 8, false>"))
    r, p = d["roofline"], d["roofline_prologue"]
    assert d["build_id"] == "abcdef012345" and r["kernel_instantiation"] == "greedy_wave_kernel<1, 8, false>"
    assert r["counters"] == "ok" and r["traffic"] == 225000.0 * 64 and r["issue"]["issue_floor_frac"] == 0.58
    assert "build abcdef012345" in r["traffic_source"]
    assert p["counters"] == "ok" and p["traffic"] == 109000.0 * 64 and p["counter_frac"] is not None
    # achieved / frac are this run's own measurement (algorithmic bytes over the events' launch time), whatever the file says
    assert p["frac"] == pytest.approx(p["achieved"] / p["peak"]) and p["achieved"] != p["counter_achieved"]
    assert d["events_leg"]["steps"] == 2 and "events" in r["measured_by"]


def test_counters_of_another_build_are_stale(tmp_path, monkeypatch):
    d = _line(tmp_path, monkeypatch, _entry("000000000000", "greedy_wave_kernel<1, 8, false>"))
    r, p = d["roofline"], d["roofline_prologue"]
    assert r["counters"] == "stale" and r["traffic"] is None and r["traffic_frac"] is None and r["issue"] is None
    assert r["traffic_source"].startswith("stale:") and "000000000000" in r["traffic_source"]
    assert p["counters"] == "stale" and p["traffic"] is None and p["counter_frac"] is None and p["counter_achieved"] is None
    assert r["achieved"] > 0 and p["achieved"] > 0          # (the run's own figures are untouched)


def test_counters_of_another_kernel_instantiation_are_stale(tmp_path, monkeypatch):
    # the file profiled the plain form, this run's launches were the LDS-shared form
    d = _line(tmp_path, monkeypatch, _entry("abcdef012345", "greedy_wave_kernel<1, 1, false>"))
    assert d["roofline"]["counters"] == "stale" and d["roofline"]["traffic"] is None
    assert "greedy_wave_kernel<1, 1, false>" in d["roofline"]["traffic_source"] and "greedy_wave_kernel<1, 8, false>" in d["roofline"]["traffic_source"]
    # ... and a prologue of another diff-weight form
    d = _line(tmp_path, monkeypatch, _entry("abcdef012345", "greedy_wave_kernel<1, 8, false>", prologue="prologue_kernel<1, 2>"))
    assert d["roofline"]["counters"] == "ok" and d["roofline_prologue"]["counters"] == "stale" and d["roofline_prologue"]["traffic"] is None


def test_an_unstamped_file_is_stale(tmp_path, monkeypatch):
    doc = _entry("abcdef012345", "greedy_wave_kernel<1, 8, false>")
    del doc["DHGR"]["build_id"]                      # (round 5's file: no build id at all)
    d = _line(tmp_path, monkeypatch, doc)
    assert d["roofline"]["counters"] == "stale" and "<unstamped>" in d["roofline"]["traffic_source"]
