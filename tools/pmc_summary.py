"""summarise gpurun_out/pmc_<tag>/p*/**/*_counter_collection.csv for the gemm kernel"""
import collections, csv, glob, sys
tag = sys.argv[1]
agg = collections.defaultdict(list)
dur = []
for f in glob.glob("gpurun_out/pmc_%s/p*/*/*_counter_collection.csv" % tag) + glob.glob("gpurun_out/pmcb_%s/p*/*/*_counter_collection.csv" % tag):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("gemm_f32_kernel", "gemm_f32_dual", "gemm_bf16_kernel", "gemm_bf16_dma_kernel")):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("kernel duration under PMC: mean %.1f us" % (sum(dur) / max(len(dur), 1)))
for k in sorted(agg):
    v = agg[k]
    print("%-32s %14.4g" % (k, sum(v) / len(v)))
