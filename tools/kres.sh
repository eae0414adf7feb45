#!/bin/bash
# Register / occupancy table of every kernel in one .hip file (compiler view): tools/kres.sh roma_amd/csrc/local_corr_rows.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $KRES_FLAGS -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
rows=[];cur=None
for l in sys.stdin:
    m=re.search(r"remark:\s+Function Name: (\S+)",l)
    if m:
        cur={"name":m.group(1)};rows.append(cur);continue
    m=re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?:\s+(\d+)",l)
    if m and cur is not None: cur[m.group(1).strip()]=m.group(2)
for r in rows:
    n=re.sub(r"^_ZN4roma12_GLOBAL__N_1\d+","",r["name"]);n=re.sub(r"EEvNS.*|EEvP.*","",n)
    print("%-48s VGPR %3s AGPR %3s SGPR %3s spillV %s spillS %s occ %s"%(n[:48],r.get("VGPRs"),r.get("AGPRs"),r.get("TotalSGPRs"),r.get("VGPRs Spill"),r.get("SGPRs Spill"),r.get("Occupancy")))
'
