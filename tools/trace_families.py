"""The per-step breakdown of tools/trace_steps.py summed by kernel FAMILY (the rows HISTORY.md section 7b quotes).
usage: trace_families.py <steady_state_per_step.txt>"""
import re
import sys

# first match wins: (substring of the kernel name, family)
FAMILIES = [("splitk_gemm", "GEMM forward / data gradient, split-K tile (small launches)"), ("direct_gemm_pair", "GEMM pair launches (two stacks side by side)"), ("direct_gemm_kernel", "GEMM forward / data gradient (direct_gemm)"), ("direct_gemm_tail_kernel", "GEMM forward / data gradient (direct_gemm)"),
            ("wgrad2_group", "weight gradients, grouped launches"), ("wgrad2_kernel", "weight gradients (wgrad2)"),
            ("fused_bwd", "data + weight gradient in one kernel (fused_bwd)"), ("wgrad_reduce", "weight-gradient slice reductions"),
            ("dw0", "SA level 0 dW0 from the columns"), ("conv_", "LDS-staged GEMMs (unaligned layers)"),
            ("finalize", "BatchNorm finalizes (forward + backward)"), ("partials_fold", "BatchNorm partial folds"),
            ("pool_t", "pool forward"), ("pool_c_kernel", "pool forward"), ("gmax", "pool forward"),
            ("pool_bwd", "pool backward (one pass; rounds 1-4: zero fill + scatter + statistics)"), ("zero_cols", "pool backward (one pass; rounds 1-4: zero fill + scatter + statistics)"),
            ("reduce_gather", "layer-0 list sums + index build"), ("csr_build", "layer-0 list sums + index build"), ("reduce_c_kernel", "layer-0 list sums + index build"),
            ("expand_c", "layer-0 expand"), ("compact_", "compaction"), ("center_term", "centre term"),
            ("fps", "farthest-point sampling"), ("ball_query", "ball query / kNN / gathers"), ("knn", "ball query / kNN / gathers"),
            ("gather_rows", "ball query / kNN / gathers"), ("sample_query", "ball query / kNN / gathers"), ("rpn_votes", "head glue (votes, box assembly)"), ("box_assemble", "head glue (votes, box assembly)"), ("pack_", "packs"), ("track_loss", "loss"), ("adam", "optimizer + weight prep"),
            ("prep_weights", "optimizer + weight prep"), ("at::native", "torch launches"), ("Cijk", "torch launches"),
            ("rocclr", "torch launches"), ("multi_tensor", "torch launches")]


def family(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return "other"


def main(path):
    lines = open(path).read().splitlines()
    agg = {}
    for line in lines[1:]:
        m = re.match(r"\s*([\d.]+) ms\s+([\d.]+) x\s+(.*)", line)
        if not m:
            continue
        a = agg.setdefault(family(m.group(3)), [0.0, 0.0])
        a[0] += float(m.group(1))
        a[1] += float(m.group(2))
    print(lines[0])
    for fam, (ms, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print("%7.3f ms %6.1f x  %s" % (ms, cnt, fam))
    print("%7.3f ms %6.1f x  total of the listed kernels" % (sum(a[0] for a in agg.values()), sum(a[1] for a in agg.values())))


if __name__ == "__main__":
    main(sys.argv[1])
