"""kernel-trace CSV of scripts/prof_deepresnet.py -> average microseconds per drn_conv_kernel instantiation."""
import csv, sys
from collections import defaultdict
per = defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "drn_conv_kernel" in k:
        per[k.split("drn_conv_kernel<")[1].split(">(")[0]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
print(" | ".join(f"{k.replace('h16<0>, ', '')}: {sum(v[1:]) / max(1, len(v[1:])):.0f}" for k, v in per.items()))
