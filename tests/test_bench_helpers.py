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
