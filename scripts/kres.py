"""Register / spill table of the kernels of one HIP source:
    python scripts/kres.py tce_rl_amd/csrc/mlpw_f32.hip [extra hipcc flags]"""
import re, subprocess, sys
src, extra = sys.argv[1], sys.argv[2:]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/kres.o"] + extra,
                     capture_output=True, text=True)
if out.returncode:
    print(out.stderr[-4000:]); sys.exit(1)
cur = None
rows = {}
for line in out.stderr.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, r in rows.items():
    print("%-70s V %3d A %3d scratch %4d vspill %3d sspill %3d occ %d lds %d" % (
        k[:70], r.get("VGPRs", 0), r.get("AGPRs", 0), r.get("ScratchSize", 0),
        r.get("VGPRs Spill", 0), r.get("SGPRs Spill", 0), r.get("Occupancy", 0), r.get("LDS Size", 0)))
