"""Find loads the compiler turned into divergent branches: `cond ? constant : mem[...]` compiles to s_and_saveexec / s_cbranch_execz
around a single ds_read / global_load with its OWN s_waitcnt - one serialized memory round trip per occurrence (round 6: seven per
row block in every bf16-family block-0 kernel).  Counts, per kernel of a .hip file, the basic blocks that are <= 4 instructions long,
start behind an s_cbranch_execz and contain a load.   python tools/branchy_loads.py [file.hip ...]"""
import os, re, subprocess, sys
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dcase2019_task4_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
for f in files:
    out = f"/tmp/bl_{f}.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DSED_AB", "-S", "--cuda-device-only",
                    os.path.join(csrc, f), "-o", out], check=True, stderr=subprocess.DEVNULL)
    kern, rows = None, {}
    lines = open(out).read().split("\n")
    i = 0
    while i < len(lines):
        l = lines[i].strip()
        m = re.match(r"^(_Z\w+):", lines[i])
        if m:
            kern = m.group(1)
        if kern and l.startswith("s_cbranch_execz"):
            body = []
            j = i + 1
            while j < len(lines) and len(body) < 6:
                t = lines[j].strip()
                if t.startswith(".LBB") or t.startswith("s_or_b64 exec"):
                    break
                if t and not t.startswith(";"):
                    body.append(t)
                j += 1
            if len(body) <= 4 and any(b.startswith(("ds_read", "global_load", "buffer_load", "flat_load")) for b in body):
                rows[kern] = rows.get(kern, 0) + 1
        i += 1
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]):
        if v >= 2:
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
            print(f"{f:12s} {v:4d}  {name}")
