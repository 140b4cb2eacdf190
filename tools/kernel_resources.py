"""Register / scratch / occupancy table of the kernels of one .hip file (device-only compile with
-Rpass-analysis=kernel-resource-usage): python tools/kernel_resources.py audiocodec_amd/csrc/ac_fast.hip [name filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-ffp-contract=fast",
       "-fvisibility=hidden", "--offload-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[3:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
seen = set()
for blk in re.split(r"remark: Function Name: ", err)[1:]:
    name = blk.split(" [")[0]
    if name in seen or flt not in name:
        continue
    seen.add(name)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("void ac::(anonymous namespace)::", "").split("(")[0]

    def g(k):
        m = re.search(re.escape(k) + r": (\d+)", blk)
        return int(m.group(1)) if m else -1
    print("%-58s VGPR %3d AGPR %3d scratch %4d  waves/SIMD %d  SGPR spill %d" % (dem[:58], g("VGPRs"), g("AGPRs"), g("ScratchSize [bytes/lane]"),
                                                                             g("Occupancy [waves/SIMD]"), g("SGPRs Spill")))
