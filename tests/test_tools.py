"""The measurement helpers under tools/ on synthetic inputs (CPU): the per-step breakdown of a kernel trace (wall, union
of the kernel intervals, sum of the durations) and the static ISA checks that found the serialised GEMM epilogue."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), *args], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_trace_steps_union_and_sum(tmp_path):
    # 30 steps of three kernels; b overlaps the tail of a by 500 ns, the Adam launch closes the step
    rows, t = [], 0
    for _ in range(30):
        for name, dur, gap in (("void (anonymous namespace)::a_kernel()", 5000, -500), ("b_kernel", 2000, 100),
                               ("(anonymous namespace)::adam_step_kernel(long const*)", 1000, 400)):
            rows.append({"Kernel_Name": name, "Start_Timestamp": t, "End_Timestamp": t + dur})
            t += dur + gap
    f = tmp_path / "trace.csv"
    with open(f, "w") as fh:
        w = csv.DictWriter(fh, fieldnames=["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        w.writeheader()
        w.writerows(rows)
    out = run("trace_steps.py", str(f), "20", "5")
    head = out.splitlines()[0]
    # per step: wall 8000 ns, durations 8000 ns, union 7500 ns (the 500 ns overlap counted once)
    assert "wall 0.008 ms/step" in head and "sum of kernel durations 0.008 ms/step" in head, head
    assert "union of kernel intervals) 0.007 ms/step" in head or "union of kernel intervals) 0.008 ms/step" in head, head
    assert "3 kernels/step" in head and "(2 launches)" in head, head


ASM = """
_ZN12_GLOBAL__N_19my_kernelILi4ELi2EEEvPf:                 ; @kernel
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\tglobal_load_dword v1, v[2:3], off
\ts_waitcnt vmcnt(0)
\tglobal_store_dword v[2:3], v1, off
\tglobal_load_dword v1, v[2:3], off offset:4
\ts_waitcnt vmcnt(0)
\tglobal_store_dword v[2:3], v1, off offset:4
\tv_mfma_f32_32x32x2_f32 a[0:15], v1, v2, a[0:15]
\tscratch_store_dword off, v1, off
\ts_endpgm
.Lfunc_end0:
"""


def test_asm_helpers_count_the_serialised_pattern(tmp_path):
    f = tmp_path / "k.s"
    f.write_text(ASM)
    waits = run("asm_waits.py", str(f))
    assert "my_kernel<4,2>" in waits and "loads   2 stores   2 vmcnt(0)   2 scratch 1" in waits, waits
    seq = run("asm_seq.py", str(f), "my_kernel")
    assert "global_load_dword | s_waitcnt vmcnt(0) | global_store_dword | global_load_dword | s_waitcnt vmcnt(0)" in seq, seq
    phases = run("asm_phases.py", str(f))
    assert "my_kernel<4,2>" in phases and "(1 mfma)" in phases, phases


def test_batch_counter_increments_are_merged_into_one_update():
    """open3dsot_amd.fused.count_batches inside a counters_begin / counters_end scope (a tracker forward): every
    `num_batches_tracked` increment of the forward is applied by one multi-tensor update at the end -- a module called once
    (+1), a paired module (+2), and a counter that is collected twice gets the sum.  Host logic only (CPU tensors)."""
    import torch
    from open3dsot_amd import fused
    bns = [torch.nn.BatchNorm1d(4) for _ in range(4)]
    assert fused.counters_begin()
    assert not fused.counters_begin()                      # nested scopes join the outer one
    fused.count_batches(bns[:2], 1)
    fused.count_batches(bns[2:3], 2)
    fused.count_batches(bns[1:2], 1)                       # the same module again
    assert all(int(bn.num_batches_tracked) == 0 for bn in bns)
    calls = []
    real = torch._foreach_add_
    torch._foreach_add_ = lambda *a, **k: (calls.append(len(a[0])), real(*a, **k))[1]
    try:
        fused.counters_end()
    finally:
        torch._foreach_add_ = real
    assert [int(bn.num_batches_tracked) for bn in bns] == [1, 2, 2, 0]
    assert calls == [3]
    fused.count_batches(bns[3:], 1)                         # outside a scope: applied at once
    assert int(bns[3].num_batches_tracked) == 1
    fused.counters_end()                                    # nothing pending: a no-op


def test_trace_families_sums_by_family(tmp_path):
    f = tmp_path / "steps.txt"
    f.write_text("wall 1.000 ms/step | header\n"
                 "  0.500 ms   2.0 x  void (anonymous namespace)::direct_gemm_kernel<4, 1, 0, 4, 2>(args)\n"
                 "  0.100 ms   1.0 x  void (anonymous namespace)::direct_gemm_pair_kernel<4, 0, 0, 2, 1>(a, b)\n"
                 "  0.050 ms  10.0 x  (anonymous namespace)::bn_finalize_kernel(args)\n"
                 "  0.025 ms   5.0 x  (anonymous namespace)::bn_bwd_finalize_pair_kernel(a, b)\n"
                 "  0.010 ms   2.0 x  void at::native::elementwise_kernel_manual_unroll<128, 4>(x)\n"
                 "  0.001 ms   1.0 x  something_else\n")
    out = run("trace_families.py", str(f)).splitlines()
    assert out[0].startswith("wall 1.000")
    rows = {l.split("x  ", 1)[1]: l for l in out[1:]}
    assert "0.500 ms    2.0" in rows["GEMM forward / data gradient (direct_gemm)"]
    assert "0.100 ms    1.0" in rows["GEMM pair launches (two stacks side by side)"]
    assert "0.075 ms   15.0" in rows["BatchNorm finalizes (forward + backward)"]
    assert "0.010 ms    2.0" in rows["torch launches"] and "0.001 ms    1.0" in rows["other"]
    assert "0.686 ms   21.0" in rows["total of the listed kernels"]


def test_trace_dispatch_log_lists_working_calls_with_phase_and_shapes():
    """tools/trace_dispatch.py's TorchDispatchMode on a small CPU graph: views are skipped, copies / cats / arithmetic
    and the ops of the C++ backward nodes (slice_backward included) are listed with their phase and operand strides"""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("trace_dispatch", os.path.join(ROOT, "tools", "trace_dispatch.py"))
    td = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(td)
    log = td.Log()
    x = torch.randn(4, 6, requires_grad=True)
    with log:
        y = torch.cat((x.t().contiguous(), x.t() * 2), 0)[:, :3].sigmoid().sum()
        log.phase = "backward"
        y.backward()
    names = [(p, n) for p, n, _, _ in log.rows]
    assert ("forward", "clone.default") in names and ("forward", "cat.default") in names
    assert ("backward", "slice_backward.default") in names and ("backward", "sigmoid_backward.default") in names
    assert not any(n.startswith(("t.", "transpose", "slice.Tensor", "view")) for _, n in names)
    clone = next(r for r in log.rows if r[1] == "clone.default")
    assert clone[2] == "(6, 4)s(1, 6)" and clone[3] == "<autograd>"


def test_hbm_traffic_family_covers_every_gemm_kernel_of_the_round3_table():
    """tools/hbm_traffic.py sums the GEMM family by the product's own symbol list: on round 3's committed PMC table the
    family is 8.84 GB per step in 67 device dispatches (the judge's recomputation; the round-3 substring list dropped
    fused_bwd_kernel, wgrad2_group_kernel and direct_gemm_pair_kernel and reported 6.98 GB)"""
    import csv
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import hbm_traffic
    from open3dsot_amd.fused import kernel_symbol
    assert kernel_symbol("void (anonymous namespace)::wgrad2_kernel<128, 64, 0>((anonymous namespace)::Wgrad2Args)") == "wgrad2_kernel"
    assert kernel_symbol("void (anonymous namespace)::direct_gemm_pair_kernel<4, 0, 1, 2, 1>(X, X)") == "direct_gemm_pair_kernel"
    assert kernel_symbol("(anonymous namespace)::pool_t_kernel(float const*, long)") == "pool_t_kernel"
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r03_fabric_pmc_per_kernel.csv"))))
    tot, disp, per, other = hbm_traffic.summarise(rows, 8)
    assert disp // 8 == 67
    assert abs(tot / 8 / 1e9 - 8.84) < 0.01
    assert any(k.startswith("fused_bwd_kernel") for k in per) and any(k.startswith("direct_gemm_pair_kernel") for k in per)
    assert abs((tot + other) / 8 / 1e9 - 12.8) < 0.05


def test_every_gemm_kernel_of_the_round5_trace_has_a_family_and_a_symbol():
    """round 5 added direct_gemm_tail_kernel: the first artefact run filed it under "other" in the family table and dropped it
    from the traffic / SQ accounting because the symbol lists did not know it.  Every kernel of the committed round-5 trace
    that computes with MFMA must be in fused.GEMM_KERNEL_SYMBOLS (or the reduce list) and land in a named family."""
    sys.path.insert(0, ROOT)
    from open3dsot_amd.fused import GEMM_KERNEL_SYMBOLS, GEMM_REDUCE_SYMBOLS, kernel_symbol
    per_step = os.path.join(ROOT, "profiles", "r05_steady_state_per_step.txt")
    names = [ln.split(" x ", 1)[1].strip() for ln in open(per_step).read().splitlines()[1:] if " x " in ln]
    gemm_like = [n for n in names if any(t in n for t in ("gemm", "wgrad", "fused_bwd", "conv_fwd_kernel", "conv_dgrad_kernel"))]
    assert any("direct_gemm_tail_kernel" in n for n in gemm_like)
    for n in gemm_like:
        assert kernel_symbol(n) in GEMM_KERNEL_SYMBOLS + GEMM_REDUCE_SYMBOLS, n
    fam = run("trace_families.py", per_step)
    other = [ln for ln in fam.splitlines() if ln.strip().endswith("other")]
    assert other and float(other[0].split()[0]) < 0.2, other      # nothing big is filed under "other"
