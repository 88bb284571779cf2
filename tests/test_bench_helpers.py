"""bench.py's bookkeeping that needs no GPU: a committed rocprofv3 record is attached to a bench line only when it was
taken on the sources the library is built from, and the fingerprint of one kernel's sources ignores the other kernels'
files (profiles/hbm_traffic.json, owlraytracing_amd/_lib.py::source_fingerprint)."""
import importlib.util
import json
import os

from owlraytracing_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_fingerprints_are_per_kernel():
    whole, team, union = _lib.source_fingerprint(), _lib.source_fingerprint("team_kernel"), _lib.source_fingerprint("db_group_union_kernel")
    for f in (whole, team, union):
        assert len(f) == 16 and int(f, 16) >= 0
    assert len({whole, team, union}) == 3
    # the same call twice gives the same answer (files are read in a fixed order)
    assert team == _lib.source_fingerprint("team_kernel")


def test_committed_records_name_the_sources_they_were_taken_on():
    recs = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    assert recs, "profiles/hbm_traffic.json is empty"
    for key, rec in recs.items():
        kernel, n, k = key.split(":")
        assert n.startswith("n=") and k.startswith("k=")
        assert len(rec["source_sha16"]) == 16 and rec["bytes_per_launch"] > 0
        assert "rocprofv3" in rec["how"]


def test_a_record_of_other_sources_is_not_attached(monkeypatch):
    bench = _bench()
    recs = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    key = next(k for k in recs if k.startswith("team_kernel:"))
    n, k = int(key.split(":")[1][2:]), int(key.split(":")[2][2:])
    monkeypatch.setattr(_lib, "source_fingerprint", lambda kernel=None: recs[key]["source_sha16"])
    rec, why = bench.committed_profile("team_kernel", n, k)
    assert rec is not None and why is None and rec["bytes_per_launch"] == recs[key]["bytes_per_launch"]
    monkeypatch.setattr(_lib, "source_fingerprint", lambda kernel=None: "0" * 16)
    rec, why = bench.committed_profile("team_kernel", n, k)
    assert rec is None and "other kernel sources" in why
    rec, why = bench.committed_profile("team_kernel", n + 1, k)
    assert rec is None and "no PMC record" in why


def test_dbscan_rooflines_cover_the_three_traversal_kernels(monkeypatch):
    """bench.py reports a roofline per traversal kernel of a tknnDbscan call (not only the one with the largest
    sum): algorithmic bytes by the contract's formula, the kernel's own HIP-event time, PMC traffic only from a record of
    these sources."""
    bench = _bench()
    info = {"core_ms": 0.66, "union_ms": 2.0, "label_ms": 0.5, "union_launches": 2, "union_node_tests": 15_000_000, "groups": 527_401,
            "core_point_tests": 2_600_000, "union_point_tests": 1_000_000, "label_point_tests": 40_000}
    monkeypatch.setattr(bench, "committed_profile", lambda kernel, n, k: ({"bytes_per_launch": 472_000_000, "wave_wait_frac": 0.8}, None)
                        if kernel == "db_core_kernel" else (None, "no PMC record for this kernel and size"))
    r = bench.dbscan_rooflines(info, [info, info], 10_000_000, 4)
    assert set(r) == {"db_core_kernel", "db_group_union_kernel", "db_label_kernel"}
    core = r["db_core_kernel"]
    assert core["algorithmic_bytes_per_launch"] == 12 * 2_600_000 + 12 * 10_000_000 + 1 * 10_000_000
    assert core["traffic"] == 472_000_000 and abs(core["traffic_over_algorithmic"] - 472e6 / core["algorithmic_bytes_per_launch"]) < 1e-9
    assert abs(core["achieved"] - core["algorithmic_bytes_per_launch"] / 0.66e-3 / 1e9) < 1e-6 and core["frac"] == core["achieved"] / 8000.0
    union = r["db_group_union_kernel"]
    assert union["launches_per_step"] == 2 and abs(union["kernel_ms"] - 1.0) < 1e-12
    assert union["algorithmic_bytes_per_launch"] == (32 * 15_000_000 + 12 * 1_000_000 + 32 * 527_401 * 2) // 2
    assert union["traffic"] is None and "no PMC record" in union["traffic_note"]
    assert r["db_label_kernel"]["algorithmic_bytes_per_launch"] == 12 * 40_000 + 12 * 10_000_000 + 4 * 10_000_000


def test_sharded_infos_of_two_calls_merge_into_one_solve():
    """ShardedTrueKNN._merge_infos: a solve and the re-solve of its stragglers (tknnSolveOptions.phase = 3) are one solve --
    work adds up, the stragglers' early levels are not counted twice, the second call says who is still unfinished."""
    from owlraytracing_amd.distributed import ShardedTrueKNN
    a = {"total_intersections": 1000, "total_active_rounds": 300, "node_tests": 50, "point_tests": 70, "tie_rows": 2, "tie_rows_left": 0,
         "solve_ms": 3.0, "tie_ms": 0.1, "unfinished": 7, "rounds": 3, "final_radius": 0.04, "dominant_kernel_launches": 1, "dominant_kernel_ms": 2.5}
    b = {"total_intersections": 90, "total_active_rounds": 28, "node_tests": 5, "point_tests": 9, "tie_rows": 0, "tie_rows_left": 0,
         "solve_ms": 0.5, "tie_ms": 0.0, "unfinished": 0, "rounds": 4, "final_radius": 0.08, "dominant_kernel_launches": 1, "dominant_kernel_ms": 0.4}
    m = ShardedTrueKNN._merge_infos(a, b, 7 * 3)  # seven stragglers had run three levels each in the first call
    assert m["total_intersections"] == 1090 and m["total_active_rounds"] == 300 + 28 - 21
    assert m["unfinished"] == 0 and m["rounds"] == 4 and m["final_radius"] == 0.08 and m["solve_ms"] == 3.5
    assert m["dominant_kernel_launches"] == 2 and abs(m["dominant_kernel_ms"] - 1.45) < 1e-12
    both = ShardedTrueKNN._merge_infos(a, b, 0)  # interior + boundary phases: unfinished queries of both count
    assert both["unfinished"] == 7 and both["total_active_rounds"] == 328


def test_one_red_check_fails_the_run_whatever_the_others_say():
    """a mismatch in the dbscan_config3 leg must not be overwritten by a green kNN spot check."""
    bench = _bench()
    assert bench.run_failed({"dbscan_config3": False, "parity_spot_check": True})
    assert bench.run_failed({"dbscan_config3": True, "parity_spot_check": False})
    assert not bench.run_failed({"dbscan_config3": True, "parity_spot_check": True})
    assert not bench.run_failed({})
    # and the main body folds its checks through that function only
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "failed = not ok" not in src.split("def dbscan_config3_leg")[0]


def test_multi_rank_line_names_the_set_it_timed(monkeypatch):
    """with N > 1 the default run timed N x 10 M points under the metric "10M pts".  Now the headline of a default
    N-rank run is BASELINE's 10 M set cut into N tiles (strong), N x 10 M and (N = 8) the 100 M set are sub-objects, every
    metric string names the size that ran, and the exchange is labelled by the backend that carried it."""
    import argparse

    bench = _bench()
    monkeypatch.setattr(bench, "committed_profile", lambda kernel, n, k: (None, "no PMC record for this kernel and size"))
    assert bench.size_label(10_000_000) == "10M" and bench.size_label(80_000_000) == "80M" and bench.size_label(1234) == "1234"
    assert "RCCL" in bench.exchange_label("nccl") and "RCCL" not in bench.exchange_label("gloo") and "gloo" in bench.exchange_label("gloo")
    info = {"dominant_kernel_ms": 1.0, "total_intersections": 4_900_000, "total_active_rounds": 250_000, "dominant_kernel_launches": 1, "kernel_used": 3,
            "rounds": 4, "point_tests": 50_000_000, "node_tests": 2_000_000}
    args = argparse.Namespace(k=10, steps=10, warmup=2)
    for world, n_total, scaling in ((8, 10_000_000, "strong"), (8, 80_000_000, "weak"), (8, 100_000_000, "strong"), (1, 10_000_000, "weak")):
        sharded = world > 1
        line = bench.trueknn_line(args, info, [info], 1e9, 1.0, n_total, n_total // world, world, sharded, 0.0025,
                                  "; %d Morton tiles, %s" % (world, bench.exchange_label("nccl")) if sharded else "", scaling)
        assert line["metric"] == "kNN queries/sec (%s pts, k=10)" % bench.size_label(n_total)
        assert line["config"]["n_points_total"] == n_total and str(n_total) in line["config"]["workload"]
        if sharded:
            assert "RCCL" in line["config"]["workload"] and ("cut into 8 tiles" in line["config"]["workload"]) == (scaling == "strong")
        assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    run = {"value": 2e9, "ms_per_step": 40.0, "steps": 5, "warmup": 1, "n_points_total": 80_000_000, "start_radius": 0.00125,
           "ranks": [{"rank": j, "points": 10_000_000, "halo_points": 300_000, "halo_exchanges": 1, "ms_per_step_own_clock": 39.0,
                      "phase_ms": {"select": 0.4, "exchange": 0.3, "halo_build": 0.3, "solve": 7.0, "reduce": 0.1}, "kernel_ms": 6.8} for j in range(8)],
           "phase_ms": {"select": 0.4, "exchange": 0.3, "halo_build": 0.3, "solve": 7.0, "reduce": 0.1}, "halo_points": 2_400_000, "halo_exchanges": 1}
    sub = bench.sub_object(run, 8, 10, "weak")
    assert sub["metric"] == "kNN queries/sec (80M pts, k=10)" and sub["scaling"] == "weak" and sub["n_gpus"] == 8
    assert len(sub["ranks"]) == 8 and set(sub["phase_ms"]) == {"select", "exchange", "halo_build", "solve", "reduce"}
    json.dumps(sub)  # the sub-object goes into the JSON line as it is
