#!/usr/bin/env python3
"""Per-kernel resource report of the gfx950 build, from the compiler's own -Rpass-analysis=kernel-resource-usage remarks
(csrc/Makefile writes them to csrc/build/<file>.res).  Fails (exit 1) when any kernel that is reachable without a debug bit
spills registers or uses scratch: a spill inside a latency-bound loop is invisible in a profile and costs a memory round trip.

    python tools/check_resources.py            # table of offenders (none expected)
    python tools/check_resources.py --all      # every kernel: VGPR / AGPR / SGPR / occupancy / LDS
"""
import glob
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "dcase2019_task4_amd", "csrc")
# kernels that only run behind sed_debug_set (A/B experiments kept for the parity tests), by mangled-name substring
DEBUG_ONLY = ()

FIELDS = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch",
          "Occupancy [waves/SIMD]": "occ", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
          "LDS Size [bytes/block]": "lds"}


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True,
                             check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def parse(build=True):
    if build:
        subprocess.run(["make", "-C", CSRC, "-j8", "-s"], check=True)
    kernels = []
    for path in sorted(glob.glob(os.path.join(CSRC, "build", "*.res"))):
        cur = None
        for line in open(path):
            m = re.search(r"remark:\s+Function Name: (\S+)", line)
            if m:
                cur = {"file": os.path.basename(path)[:-4] + ".hip", "name": m.group(1)}
                kernels.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+)", line)
            if m and cur is not None and m.group(1).strip() in FIELDS:
                v = m.group(2)
                cur[FIELDS[m.group(1).strip()]] = int(v) if v.isdigit() else v
    dm = demangle([k["name"] for k in kernels])
    for k in kernels:
        k["pretty"] = re.sub(r"\(.*", "", dm.get(k["name"], k["name"]))
    return kernels


def offenders(kernels):
    return [k for k in kernels
            if (k.get("vgpr_spill", 0) or k.get("sgpr_spill", 0) or k.get("scratch", 0))
            and not any(d in k["name"] for d in DEBUG_ONLY)]


def main():
    ks = parse()
    show = ks if "--all" in sys.argv else offenders(ks)
    for k in show:
        print(f"{k['file']:12s} {k['pretty'][:70]:70s} vgpr {k.get('vgpr')} agpr {k.get('agpr')} sgpr {k.get('sgpr')} "
              f"occ {k.get('occ')} lds {k.get('lds')} | spill v {k.get('vgpr_spill')} s {k.get('sgpr_spill')} "
              f"scratch {k.get('scratch')}")
    bad = offenders(ks)
    print(f"{len(ks)} kernels, {len(bad)} with spills / scratch")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
