"""scripts/update_profiles.py builds profiles/hbm_traffic.json -- the PMC record bench.py attaches to its line as
`roofline.traffic` -- PER DISPATCH: VERDICT r2 found the committed record averaging six compact launches of team_kernel
with three that also wrote the 2.4 GB frameBuffer.  A record from mixed dispatches must be refused."""
import csv
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _up():
    spec = importlib.util.spec_from_file_location("update_profiles", os.path.join(ROOT, "scripts", "update_profiles.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


HEADER = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name",
          "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value",
          "Start_Timestamp", "End_Timestamp"]


def _write_pass(root, pass_no, counter, rows):
    """rows: [(kernel name, value)] in dispatch order"""
    d = os.path.join(root, "pmc_%d" % pass_no, "run")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "1_counter_collection.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(HEADER)
        for i, (kern, v) in enumerate(rows, 1):
            w.writerow([i, i, "Agent 2", 1, 1, 1, 1024, 7, kern, 64, 0, 0, 128, 0, 96, counter, repr(float(v)), 0, 1])


TEAM = "void owlmi::(anonymous namespace)::team_kernel<false, 1, false>(owlmi::TeamArgs)"
UNION = "owlmi::(anonymous namespace)::db_group_union_kernel(owlmi::DbArgs, int const*)"
OTHER = "owlmi::(anonymous namespace)::block_box_kernel(float const*)"


def test_compact_dispatches_give_the_benchmarked_launch(tmp_path):
    up = _up()
    # a cold first launch, then five alike; other kernels in between do not matter
    _write_pass(tmp_path, 1, "FETCH_SIZE", [(OTHER, 5), (TEAM, 470000), (TEAM, 375010), (OTHER, 9), (TEAM, 375000), (TEAM, 374990), (TEAM, 375020), (TEAM, 375003)])
    _write_pass(tmp_path, 2, "WRITE_SIZE", [(TEAM, 1571175), (TEAM, 1571159), (TEAM, 1571180), (TEAM, 1571173), (TEAM, 1571181), (TEAM, 1571177)])
    rec = up.build_record(str(tmp_path), "team_kernel", 1, "0123456789abcdef", "python3 bench.py --no-fb-leg", "t")
    assert rec["FETCH_SIZE_KB_per_launch"] == 375003 and rec["WRITE_SIZE_KB_per_launch"] == 1571177
    assert rec["bytes_per_launch"] == (2 * 375003 + 1571177) * 1024
    assert rec["dispatches_used"] == {"FETCH_SIZE": 5, "WRITE_SIZE": 5}
    assert "PER DISPATCH" in rec["how"]


def test_a_record_from_mixed_dispatches_is_refused(tmp_path):
    """round 2's command: six compact solves and three that also write the frameBuffer -- one kernel name, two populations"""
    up = _up()
    fetch = [375000] * 6 + [596000] * 3
    write = [1571170] * 6 + [4483400] * 3
    _write_pass(tmp_path, 1, "FETCH_SIZE", [(TEAM, v) for v in fetch])
    _write_pass(tmp_path, 2, "WRITE_SIZE", [(TEAM, v) for v in write])
    with pytest.raises(up.MixedDispatches) as e:
        up.build_record(str(tmp_path), "team_kernel", 1, "0" * 16, "python3 bench.py", "t")
    assert "--no-fb-leg" in str(e.value)
    # the plain average the old script took would have been accepted silently: (6 * 1571170 + 3 * 4483400) / 9
    assert abs(sum(write) / 9 - 2541913) < 1


def test_two_launches_per_call_are_kept_apart(tmp_path):
    """db_group_union_kernel runs twice per tknnDbscan call (near pairs, then the rest): position by position"""
    up = _up()
    _write_pass(tmp_path, 1, "FETCH_SIZE", [(UNION, v) for v in (900, 300, 500, 250, 502, 251, 498, 249)])
    _write_pass(tmp_path, 2, "WRITE_SIZE", [(UNION, v) for v in (80, 30, 600, 200, 601, 200, 600, 201)])
    rec = up.build_record(str(tmp_path), "db_group_union_kernel", 2, "0" * 16, "python3 bench.py", "t")
    assert rec["per_position"]["FETCH_SIZE_KB"] == [500, 250] and rec["FETCH_SIZE_KB_per_launch"] == 375
    assert rec["WRITE_SIZE_KB_per_launch"] == 400 and rec["launches_per_call"] == 2
    with pytest.raises(up.MixedDispatches):  # as one population the two passes do not agree
        up.build_record(str(tmp_path), "db_group_union_kernel", 1, "0" * 16, "python3 bench.py", "t")


def test_first_call_is_dropped_per_profiled_process(tmp_path):
    up = _up()
    cold = up.steady([[100.0, 10.0, 10.1], [90.0, 10.0, 9.9]], "WRITE_SIZE")
    assert cold[0] == pytest.approx(10.0) and cold[2] == 4
    with pytest.raises(up.MixedDispatches):
        up.steady([[10.0]], "WRITE_SIZE")
