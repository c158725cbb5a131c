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
