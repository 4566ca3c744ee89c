#!/bin/bash
# effective clock and MFMA-busy fraction of kbench variants: one rocprofv3 PMC pass per variant (c2 only)
export TMPDIR=/tmp
O=gpurun_out/clk
mkdir -p $O
for v in "$@"; do
  rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/$v -o p -- python tools/kbench.py run $v --cfg c2 --rounds 2 --iters 10 > $O/$v.log 2>&1
done
python - "$@" <<'PY'
import csv, glob, sys, collections
for v in sys.argv[1:]:
    kt = {}
    for path in glob.glob("gpurun_out/clk/%s/**/*kernel_trace.csv" % v, recursive=True):
        for r in csv.DictReader(open(path)):
            kt[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    acc = collections.defaultdict(list)
    for path in glob.glob("gpurun_out/clk/%s/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(path)):
            name, dur = kt.get(r["Dispatch_Id"], ("", 0))
            if ("fwd_asm_kernel" in name or "fwd_kernel" in name) and dur > 150000:
                acc[r["Counter_Name"]].append((float(r["Counter_Value"]), dur))
    if not acc:
        print(v, "no data"); continue
    avg = {k: sum(x for x, _ in vals) / len(vals) for k, vals in acc.items()}
    dur = sum(d for _, d in acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"])
    cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
    print("%-10s dur %.1f us  cycles %.0f  clock %.3f GHz  mfma_busy %.3f  wave: active %.3f wait_inst %.3f wait_any %.3f  valu insts %.3g" % (
        v, dur / 1e3, cyc, cyc / dur, avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc),
        avg.get("SQ_ACTIVE_INST_ANY", 0) / max(avg.get("SQ_WAVE_CYCLES", 1), 1), avg.get("SQ_WAIT_INST_ANY", 0) / max(avg.get("SQ_WAVE_CYCLES", 1), 1),
        avg.get("SQ_WAIT_ANY", 0) / max(avg.get("SQ_WAVE_CYCLES", 1), 1), avg.get("SQ_INSTS_VALU", 0)))
PY
