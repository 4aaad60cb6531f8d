"""Per-kernel register / scratch / occupancy table of csrc/rtk_trace.hip as hipcc reports it
(-Rpass-analysis=kernel-resource-usage); no GPU needed.

  python3 tools/kernel_resources.py [extra hipcc flags...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "raytracingoneweekendapplication_amd", "csrc")
with tempfile.TemporaryDirectory() as tmp:
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c", "-I" + os.path.join(ROOT, "include"),
                        "-I" + csrc, os.path.join(csrc, "rtk_trace.hip"), "-o", os.path.join(tmp, "t.o"), "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:],
                       capture_output=True, text=True)
if p.returncode != 0:
    sys.exit(p.stderr[-4000:])
only = os.environ.get("RTK_RES_FILTER", "")
for block in re.split(r"remark: [^\n]*Function Name: ", p.stderr)[1:]:
    name = block.split("\n")[0].strip()
    def g(key):
        m = re.search(key + r": (\d+)", block)
        return int(m.group(1)) if m else None
    m = re.match(r"_ZN3rtk17rtk_render_kernelI([df])Lj(\d+)ELb([01])ELb([01])E", name)
    if m:
        demangled = "rtk_render_kernel<%s, %su, %s, %s>" % ("double" if m.group(1) == "d" else "float", m.group(2), "true" if m.group(3) == "1" else "false", "true" if m.group(4) == "1" else "false")
    else:
        demangled = name.split(" ")[0][:60]
    if only and only not in demangled:
        continue
    scratch, occupancy = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{demangled:62s} vgpr={g('VGPRs')} scratch={scratch} waves/SIMD={occupancy} sgpr={g('SGPRs')}")
