"""The measurement tooling must not pass silently on empty counter tables (the SQ tables of r02g..r03f were
header-only because the per-dispatch CSV had been deleted before the summary ran), and the shared bench arithmetic
(visgeom_amd/benchlib.py) prices the passes as DESIGN.md section 5 states."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pmc_aggregate_reduces_per_dispatch_rows_and_fails_on_nothing(tmp_path):
    run = tmp_path / "in" / "headline_A"
    run.mkdir(parents=True)
    with open(run / "t_counter_collection.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        for i, v in enumerate((10.0, 30.0)):
            w.writerow([i, "void vg::vg_gram_valu_kernel<0, 1, true, 3>(vg::GramValuArgs)", "SQ_WAVES", v])
        w.writerow([9, "some_other_kernel(int)", "SQ_WAVES", 5.0])
    out = tmp_path / "agg.csv"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_aggregate.py"), str(tmp_path / "in"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = list(csv.DictReader(open(out)))
    assert len(rows) == 1 and rows[0]["run"] == "headline_A" and rows[0]["kernel"] == "vg::vg_gram_valu_kernel<0, 1, true, 3>"
    assert float(rows[0]["mean"]) == 20.0 and int(rows[0]["dispatches"]) == 2
    empty = tmp_path / "empty"
    empty.mkdir()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_aggregate.py"), str(empty), str(tmp_path / "e.csv")], capture_output=True, text=True)
    assert r.returncode == 3 and "NO counter rows" in r.stderr


def test_sq_summary_fails_loudly_without_its_table():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_sq_summary.py"), "no_such_tag"], capture_output=True, text=True)
    assert r.returncode == 3


def test_benchlib_prices_the_passes_as_design_states():
    from visgeom_amd import benchlib as B

    assert B.emit_bytes_per_obs("eucm", 1) == 224 and B.emit_bytes_per_obs("ucm", 1) == 208 and B.emit_bytes_per_obs("mei", 1) == 288
    assert B.emit_bytes_per_obs("eucm", 2) == 320
    assert B.gram_flops_per_obs("eucm", 1) == 564 and B.gram_flops_per_obs("mei", 1) == 958
    assert B.gram_flops_per_obs("eucm", 2) == 1008 and B.gram_flops_per_obs("mei", 2) == 1498 and B.gram_flops_per_obs("ucm", 1) == 509


def test_trace_gaps_groups_the_idle_time_between_dispatches_by_kernel_pair(tmp_path):
    """tools/exp/trace_gaps.py found the 5-6 us the event marker cost behind every accept kernel and the rig loop's two 26 us
    read-backs: gaps of a rocprofv3 kernel trace, grouped by (kernel, next kernel)"""
    import subprocess
    import sys

    p = tmp_path / "t_kernel_trace.csv"
    rows = ["Kernel_Name,Start_Timestamp,End_Timestamp"]
    t = 1000000
    for it in range(6):
        for name, dur, gap in (("vg::vg_a_kernel(Args)", 10000, 0), ("void vg::vg_b_kernel<4>(Args)", 5000, 6000)):
            rows.append('"%s",%d,%d' % (name, t, t + dur))
            t += dur + gap
    p.write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exp", "trace_gaps.py"), str(p)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    ab = [l for l in lines if l.startswith("vg_a_kernel") and "-> vg_b_kernel<4>" in l][0].split()
    ba = [l for l in lines if l.startswith("vg_b_kernel<4>") and "-> vg_a_kernel" in l][0].split()
    assert float(ab[-2]) == 0.0 and abs(float(ba[-2]) - 6.0) < 1e-9      # mean gap in microseconds
    assert any(l.startswith("vg_a_kernel") and "mean    10.00 us" in l for l in lines)
