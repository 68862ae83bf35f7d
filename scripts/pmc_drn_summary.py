"""rocprofv3 --pmc SQ_* counter_collection CSV of scripts/prof_deepresnet.py -> one line per conv / wgrad kernel."""
import csv, sys
from collections import defaultdict
per = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "drn_" in k:
        per[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    a = {n: sum(x[1:]) / max(1, len(x[1:])) for n, x in v.items()}
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    short = k.replace("(anonymous namespace)::", "").replace("void ", "").split(">(")[0][:60]
    gui = a.get("GRBM_GUI_ACTIVE", 0)
    print(f"{short:62s} us {gui / 8 / 2400:7.1f} wait_any {a.get('SQ_WAIT_ANY', 0) / wc:.2f} wait_inst {a.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} "
          f"active {a.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} | lds_active/CUcyc {a.get('SQ_LDS_IDX_ACTIVE', 0) / max(1, gui * 32):.2f} "
          f"conflict {a.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, a.get('SQ_LDS_IDX_ACTIVE', 0)):.2f} | mfma {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1, gui * 128):.2f}")
