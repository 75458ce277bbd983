"""bench.py's roofline record names the kernel that RAN: the row of a committed rocprofv3 summary is selected by
the kernel the library reported for the recorded graph (zh_graph_kernels) and by its instantiation -- several buffers per launch or
one -- not by the largest total; the committed lines are consistent with the kernel averages they cite."""
import csv
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_tests", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _rows(name):
    return list(csv.DictReader(open(os.path.join(ROOT, "profiles", name))))


def test_the_batch_instantiation_is_selected_for_a_coalesced_graph():
    b = _bench()
    rows = _rows("r05/pulseosc4096_driver_args_kernel_stats.csv")
    largest = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    assert largest["Name"].endswith("false>(OscArgs)")                 # what round 5's line cited: the one-buffer kernel, 4.5 us
    row, how = b.select_kernel_row(rows, "k_osc_const4", batch=True, launch_us=31.0)
    assert row["Name"].endswith("true>(OscArgs)") and 20e3 < float(row["AverageNs"]) < 40e3, (row, how)
    row, how = b.select_kernel_row(rows, "k_osc_const4[batch]", batch=True)
    assert row["Name"].endswith("true>(OscArgs)")
    row, how = b.select_kernel_row(rows, "k_osc_const4<PulseOscP>", batch=False, launch_us=4.6)
    assert row["Name"].endswith("true, false>(OscArgs)") and float(row["AverageNs"]) < 6e3      # the table form, one buffer
    assert b.select_kernel_row(rows, "k_no_such_kernel")[0] is None
    assert b.select_kernel_row(rows, "")[0] is None


def test_an_unknown_family_takes_the_instantiation_nearest_the_measured_launch():
    b = _bench()
    rows = [{"Name": "void k_x<1>(A)", "TotalDurationNs": "900", "AverageNs": "9000", "Calls": "1"},
            {"Name": "void k_x<2>(A)", "TotalDurationNs": "100", "AverageNs": "20000", "Calls": "1"},
            {"Name": "void k_xy<2>(A)", "TotalDurationNs": "5000", "AverageNs": "19000", "Calls": "1"}]
    assert b.select_kernel_row(rows, "k_x", launch_us=21.0)[0]["Name"] == "void k_x<2>(A)"
    assert b.select_kernel_row(rows, "k_x")[0]["Name"] == "void k_x<1>(A)"              # no time to go by: the larger total
    assert b.select_kernel_row(rows, "k_xy", launch_us=1.0)[0]["Name"] == "void k_xy<2>(A)"   # k_x is not a prefix match of k_xy


def test_rocprof_record_prefers_the_file_collected_at_the_drivers_arguments():
    b = _bench()

    class A:
        workload, tolerant = "pulseosc", False
    rec = b.rocprof_record(A, 4096, kernel="k_osc_const4", batch=True, launch_us=31.0, driver_form=True)
    assert rec and "driver_args" in rec["file"] and rec["kernel"].endswith("true>(OscArgs)"), rec
    rec1 = b.rocprof_record(A, 4096, kernel="k_osc_const4", batch=False, launch_us=4.6, driver_form=False)
    assert rec1 and "driver_args" not in rec1["file"] and rec1["kernel"].endswith("false>(OscArgs)"), rec1


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "profiles", "r06", "bench_driver_args_*.json"))) or [None])
def test_committed_lines_cite_the_kernel_that_ran(path):
    """every driver-form line committed this round: the cited rocprofv3 kernel is the batch instantiation, its average times the
    launches of the region fits inside the region, and frac_kernel follows from the line's own numbers"""
    if path is None:
        pytest.skip("no profiles/r06/bench_driver_args_*.json yet")
    line = json.loads([l for l in open(path) if l.startswith("{")][-1])
    rl = line["roofline"]
    rp = rl["rocprofv3_kernel_average"]
    assert rl["buffers_per_launch"] > 1 and rp["kernel"].endswith("true>(OscArgs)"), rp
    assert rp["average_us"] * rl["launches_in_region"] <= line["ms_per_step"] * 1e3 * line["steps"] * 1.001
    want = rl["algorithmic_bytes_per_launch"] / (rp["average_us"] * 1e-6) / 1e9 / rl["peak"]
    assert abs(rl["frac_kernel"] - want) < 1e-9 and rl["frac_kernel"] >= rl["frac"] * 0.98
    assert "value_form" in line and "ZH_CAPTURE_COALESCE" in line["value_form"]
