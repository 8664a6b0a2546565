"""VGPRs, SGPRs, scratch (spills), occupancy of every kernel of the library: make -C csrc resources | this script
(python scripts/kernel_resources.py > profiles/rNN_kernel_resources.txt runs both)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-C", os.path.join(ROOT, "mhm2_kmer_analysis_v2_amd", "csrc"), "resources"], capture_output=True, text=True)
cur, rows = None, {}
for l in (out.stdout + out.stderr).splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\d+)", l)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
print("%-100s %5s %5s %8s %7s %7s %4s" % ("kernel", "VGPR", "SGPR", "scratchB", "v-spill", "s-spill", "occ"))
for (k, v), d in zip(rows.items(), names):
    d = d.split("(")[0].replace("void ", "")
    print("%-100s %5d %5d %8d %7d %7d %4d" % (d[:100], v.get("VGPRs", 0), v.get("TotalSGPRs", 0), v.get("ScratchSize [bytes/lane]", 0),
                                             v.get("VGPRs Spill", 0), v.get("SGPRs Spill", 0), v.get("Occupancy [waves/SIMD]", 0)))
