"""Instruction mix of one kernel from the gfx950 assembly of a .hip file:  python tools/isa_mix.py feat.hip k_stft_mel_p"""
import re, subprocess, sys, os
from collections import Counter
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dcase2019_task4_amd", "csrc")
src, kern = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
out = f"/tmp/{os.path.basename(src)}.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DSED_AB", "-S", "--cuda-device-only",
                os.path.join(csrc, src), "-o", out] + extra, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + kern + r"\w*:", l) or l.startswith(kern + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
c = Counter()
for l in lines[start + 1:end]:
    l = l.strip()
    m = re.match(r"([a-z_0-9]+)\b", l)
    if m and not l.endswith(":") and not l.startswith((".", ";")):
        c[m.group(1)] += 1
tot = sum(c.values())
valu = sum(v for k, v in c.items() if k.startswith("v_"))
print(f"{kern}: {tot} instructions, {valu} VALU, {sum(v for k, v in c.items() if k.startswith('ds_'))} LDS, "
      f"{sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_')))} global, {c.get('s_waitcnt', 0)} s_waitcnt")
for k, v in c.most_common(40):
    print(f"{v:6d} {k}")
