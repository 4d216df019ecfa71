#!/usr/bin/env python3
"""Static resource table of every kernel in a .hip source (no GPU needed):
VGPRs, AGPRs, SGPRs, scratch, LDS, occupancy as the compiler reports them
(-Rpass-analysis=kernel-resource-usage).  usage: tools/kres.py pvq_refbands.hip [filter-regex]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "daala_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Rpass-analysis=kernel-resource-usage"]


def main():
    src = sys.argv[1]
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    extra = sys.argv[3:]
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", "/tmp/kres/x.o"],
                       capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-3000:])
        sys.exit(1)
    cur = None
    rows = []
    for line in r.stderr.splitlines():
        m = re.search(r"remark: .*?(Function Name|Name): (\S+)", line)
        if m:
            name = subprocess.run(["/usr/bin/c++filt", m.group(2)], capture_output=True,
                                  text=True).stdout.strip()
            name = name.replace("(anonymous namespace)::", "")
            name = re.sub(r"^void ", "", name)
            name = re.sub(r"\(.*$", "", name)
            cur = {"name": name}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+(\w[\w ]*?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    print("%-44s %5s %5s %5s %8s %7s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "LDS", "occ"))
    for c in rows:
        if pat and not pat.search(c["name"]):
            continue
        print("%-44s %5d %5d %5d %8d %7d %4d" % (c["name"][:44], c.get("VGPRs", -1), c.get("AGPRs", -1),
                                                 c.get("TotalSGPRs", -1), c.get("ScratchSize", -1) + c.get("VGPRs Spill", 0)*0,
                                                 c.get("LDS Size", -1), c.get("Occupancy", -1)))


if __name__ == "__main__":
    main()
