"""rocprofv3 --pmc SQ_* counter_collection CSV -> one line per kernel (wait / LDS conflict / MFMA busy fractions)."""
import csv, sys
from collections import defaultdict
per = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    per[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
rows = []
for k, v in per.items():
    a = {n: sum(x[1:]) / max(1, len(x[1:])) for n, x in v.items()}
    rows.append((a.get("GRBM_GUI_ACTIVE", 0) * max(1, len(v.get("GRBM_GUI_ACTIVE", [0])) - 1), k, a))
for tot, k, a in sorted(rows, reverse=True)[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    gui = a.get("GRBM_GUI_ACTIVE", 0) or 1
    short = k.replace("(anonymous namespace)::", "").replace("void ", "")[:58]
    print(f"{short:60s} us {gui / 8 / 2400:7.1f} wait_any {a.get('SQ_WAIT_ANY', 0) / wc:.2f} wait_inst {a.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} "
          f"active {a.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} | lds/CUcyc {a.get('SQ_LDS_IDX_ACTIVE', 0) / (gui * 32):.2f} "
          f"conflict {a.get('SQ_LDS_BANK_CONFLICT', 0) / max(1, a.get('SQ_LDS_IDX_ACTIVE', 0)):.2f} | mfma {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (gui * 128):.2f}")
