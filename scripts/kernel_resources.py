"""Register / scratch / LDS use of every kernel of a .hip file, as the compiler reports it.

    python scripts/kernel_resources.py [file.hip] [-D...] [--filter substring]

Compiles device-only for gfx950 with -Rpass-analysis=kernel-resource-usage (no GPU needed) and prints one line per kernel.
tests/test_kernel_resources.py asserts on the same table (no scratch in the streaming kernels).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "libultrahdr_dev_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math"]

_KEYS = {
    "VGPRs": "vgpr", "AGPRs": "agpr", "TotalSGPRs": "sgpr", "ScratchSize [bytes/lane]": "scratch",
    "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
    "LDS Size [bytes/block]": "lds",
}


def demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "/usr/bin/c++filt"):
        if os.path.exists(tool):
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True).stdout
            return out.split("\n")[:len(names)]
    return list(names)


def resources(src=None, defines=()):
    """-> {demangled kernel name: {vgpr, agpr, sgpr, scratch, occupancy, sgpr_spill, vgpr_spill, lds}}"""
    src = src or os.path.join(CSRC, "uhdr_kernels.hip")
    with tempfile.TemporaryDirectory() as td:
        cmd = [HIPCC] + FLAGS + list(defines) + ["--cuda-device-only", "-c", src, "-o", os.path.join(td, "k.o"),
                                                 "-Rpass-analysis=kernel-resource-usage"]
        err = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    table, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass-analysis", line) or re.search(r"remark:\s+(.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = body.split(":", 1)[1].strip()
            table[cur] = {}
            continue
        if cur is None or ":" not in body:
            continue
        k, v = body.rsplit(":", 1)
        if k.strip() in _KEYS:
            table[cur][_KEYS[k.strip()]] = int(v)
    names = list(table)
    return {d: table[n] for n, d in zip(names, demangle(names))}


def main(argv):
    src, defs, flt = None, [], ""
    it = iter(argv)
    for a in it:
        if a == "--filter":
            flt = next(it)
        elif a.startswith("-D"):
            defs.append(a)
        else:
            src = a
    t = resources(src, defs)
    print(f"{'kernel':90s} vgpr agpr sgpr scratch spillV occ    lds")
    for name, r in t.items():
        if flt and flt not in name:
            continue
        short = re.sub(r"\(.*", "", name).replace("void uhdr::", "").replace("(anonymous namespace)::", "")
        print(f"{short[:90]:90s} {r.get('vgpr', -1):4d} {r.get('agpr', -1):4d} {r.get('sgpr', -1):4d} {r.get('scratch', -1):7d} "
              f"{r.get('vgpr_spill', -1):6d} {r.get('occupancy', -1):3d} {r.get('lds', -1):6d}")


if __name__ == "__main__":
    main(sys.argv[1:])
