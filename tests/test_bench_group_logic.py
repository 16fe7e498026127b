"""bench.py --gpus N without a launcher, with the device-group child MOCKED (no GPU here, and no multi-GPU machine anywhere this code
has run): what the parent does with the children's answers — the line of a group on 8 distinct devices (n_gpus, the transport probes
with pull and rccl, rccl_ranks), of shards aliased to one device, of a run in which every group attempt failed (fallback, exit
code 3), and of a child that lost its `.meta` and its profile."""
import importlib
import json
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")


def child_answer(cfg, env, distinct=8, transport="pull", shards=8, fail=None, **over):
    if fail:
        return {"error": fail}
    forced = env.get("ICICLE_SNARK_EXCHANGE")
    tr = forced or transport
    n = len(bench_devices(cfg["device"]))
    single = n == 1
    r = {"ready": True, "cold_ms": 900.0, "shards": 0 if single else shards, "device_mb": 13000.0, "equals_single_device_proof": None if single else True,
         "describe": {"shards": 0 if single else shards, "devices": bench_devices(cfg["device"]), "distinct_devices": 1 if single else distinct, "transport": "none" if single else tr,
                      "peer_access": True, "rccl_ranks": (distinct if tr == "rccl" and not single else 0), "distributed_front_end": not single},
         "hbm_copy_gbps": 5100.0, "mad_tops": 34.0, "n_vars": 1000 + 2, "domain_size": 1024, "constraints": 1000, "what": "benchmark/1k squaring chain", "standin": True,
         "done": True, "child_ms_per_step": 4.0 if tr == "pull" else 4.6, "parent_ms_per_step": 4.1 if tr == "pull" else 4.7, "qap_ms": 0.6, "msm_ms": 2.9, "acc_ms": 0.3,
         "acc_geom": {"L": 262144, "nbuckets": 65536, "c": 17, "W": 15, "is_g2": False}, "resident_ms": 3.6, "proof": json.dumps({"protocol": "groth16"}), "public": "[]"}
    r.update(over)
    return r


def bench_devices(device):
    lst = device.split(":")[1]
    out = []
    for part in lst.split(","):
        if "-" in part:
            a, b = part.split("-")
            out += list(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def run(monkeypatch, capsys, answer, gpus=8, env=None):
    calls = []

    def fake_attempt(cfg, env_extra, first_timeout):
        calls.append((cfg["device"], dict(env_extra), cfg["steps"]))
        return answer(cfg, env_extra)
    monkeypatch.setattr(bench, "_group_attempt", fake_attempt)
    for k in ("ICICLE_SNARK_BENCH_DEVICES", "ICICLE_SNARK_EXCHANGE", "ICICLE_SNARK_BENCH_PROBE_TRANSPORTS"):
        monkeypatch.delenv(k, raising=False)
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    args = types.SimpleNamespace(gpus=gpus, steps=20, warmup=5)
    rc = bench.standalone_group(args, "1k")
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return rc, json.loads(lines[0]), calls


def test_eight_distinct_devices_probe_the_other_transports(monkeypatch, capsys):
    rc, d, calls = run(monkeypatch, capsys, lambda cfg, env: child_answer(cfg, env))
    assert rc == 0 and d["n_gpus"] == 8 and d["fallback"] is False and d["config"]["requested_gpus"] == 8
    assert d["config"]["exchange"] == "pull" and d["config"]["rccl_ranks"] == 0 and d["config"]["devices_touched"] == 8
    et = d["config"]["exchange_transports"]
    assert et["prove_ms_pull"] == pytest.approx(d["ms_per_step"]) and et["prove_ms_rccl"] == pytest.approx(4.7) and et["rccl_ranks"] == 8
    # the timed attempt with the library's own transport order, then ONE probe (rccl), a few steps
    assert [c[1].get("ICICLE_SNARK_EXCHANGE") for c in calls] == [None, "rccl"] and calls[1][2] == 5
    assert d["value"] == pytest.approx(1000 / (d["ms_per_step"] * 1e-3)) and d["roofline"]["kernel"].endswith("of shard 0")


def test_a_failing_probe_costs_nothing_but_its_entry(monkeypatch, capsys):
    def answer(cfg, env):
        return child_answer(cfg, env, fail="RCCL bootstrap did not finish" if env.get("ICICLE_SNARK_EXCHANGE") == "rccl" else None)
    rc, d, _ = run(monkeypatch, capsys, answer)
    et = d["config"]["exchange_transports"]
    assert rc == 0 and "prove_ms_rccl" not in et and "rccl" in et["transport_probe_errors"] and et["prove_ms_pull"] > 0


def test_shards_aliased_to_one_device_are_one_gpu_and_are_not_probed(monkeypatch, capsys):
    rc, d, calls = run(monkeypatch, capsys, lambda cfg, env: child_answer(cfg, env, distinct=1, shards=2), gpus=2, env={"ICICLE_SNARK_BENCH_DEVICES": "0,0"})
    assert rc == 0 and d["n_gpus"] == 1 and d["config"]["requested_gpus"] == 2 and d["fallback"] is False
    assert len(calls) == 1 and "exchange_transports" not in d["config"]


def test_every_group_attempt_failing_is_a_flagged_fallback(monkeypatch, capsys):
    def answer(cfg, env):
        return child_answer(cfg, env, fail=None if len(bench_devices(cfg["device"])) == 1 else "hipDeviceEnablePeerAccess failed")
    rc, d, calls = run(monkeypatch, capsys, answer, gpus=4)
    assert rc == 3 and d["fallback"] is True and d["n_gpus"] == 1 and d["config"]["requested_gpus"] == 4
    assert "FALLBACK" in d["config"]["msm_sharding"] and [a["error"] is None for a in d["config"]["device_group"]["attempts"]] == [False, False, False, True]


def test_a_child_without_meta_and_without_a_profile_still_prints_the_line(monkeypatch, capsys):
    def answer(cfg, env):
        r = child_answer(cfg, env, acc_geom=None, acc_ms=None)
        del r["constraints"], r["what"]
        return r
    rc, d, _ = run(monkeypatch, capsys, answer)
    assert rc == 0 and d["roofline"] is None and d["config"]["constraints"] == 1000 and d["value"] > 0
