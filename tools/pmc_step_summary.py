"""Summarise the SQ / TCC counter passes of tools/profile_round.sh (gpurun_out/<tag>/pmc{1,2,3}) per kernel family of the
bench step into profiles/<round>_bench_pmc.txt.  usage: pmc_step_summary.py gpurun_out/<tag> profiles/r02"""
import collections
import csv
import glob
import sys

src, dst = sys.argv[1], sys.argv[2]
FAMILIES = [("gemm_bf16x3_dual_kernel", "gemm_bf16x3_dual(dA+dW)"), ("gemm_bf16_dual_kernel", "gemm_bf16_dual(dA+dW)"),
            ("gemm_bf16_dma_kernel", "gemm_bf16 / bf16x3 (fwd / dW0)"),
            ("gemm_f32_dual_kernel", "gemm_f32_dual(dA+dW)"), ("gemm_f32_kernel", "gemm_f32 (fwd / dW0)"),
            ("adam_kernel", "adam_apply"), ("bn_act_forward", "bn_act_forward"), ("hb_apply", "hb_apply"),
            ("softmax_xent", "softmax_xent")]


def family(kernel):
    for key, name in FAMILIES:
        if key in kernel:
            return name
    return None


acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("%s/pmc*/**/*counter_collection.csv" % src, recursive=True):
    for r in csv.DictReader(open(f)):
        fam = family(r["Kernel_Name"])
        if fam:
            acc[fam][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[fam].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(dst + "_bench_pmc.txt", "w") as out:
    out.write("# rocprofv3 --pmc (counter-only passes, tools/profile_round.sh) over `python bench.py --steps 10 --warmup 3 "
              "--no-cpu-baseline` (BASELINE cfg2 in the bench's default arithmetic), 1x MI355X: mean per launch.\n"
              "# SQ_* cycle counters are quad-cycles summed over all waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 "
              "SIMDs (64 per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_32x32x16_bf16).\n"
              "# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz nominal); l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS).\n")
    for fam, _ in [(n, 0) for _, n in FAMILIES]:
        if fam not in acc:
            continue
        c = {k: sum(v) / len(v) for k, v in acc[fam].items()}
        d = sum(dur[fam]) / len(dur[fam])
        out.write("\n%s   (mean duration under counters %.1f us)\n" % (fam, d))
        for k in sorted(c):
            out.write("  %-28s %14.4g\n" % (k, c[k]))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            out.write("  %-28s %14.3f\n" % ("mfma_util (nominal clock)", c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * d * 1e-6 * 2.4e9)))
        if "TCC_HIT_sum" in c and c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0) > 0:
            out.write("  %-28s %14.3f\n" % ("l2_hit", c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
        if "SQ_WAVE_CYCLES" in c:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if k in c:
                    out.write("  %-28s %14.3f\n" % (k + " / WAVE_CYCLES", c[k] / c["SQ_WAVE_CYCLES"]))
print(open(dst + "_bench_pmc.txt").read())
