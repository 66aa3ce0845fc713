"""Per-kernel register / LDS / scratch table of one .hip file, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    python tools/kernel_resources.py trackformer_amd/csrc/linear_split.hip [substring filter]
"""
import re
import subprocess
import sys


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Iinclude", "-Wno-pass-failed",
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (.*?)(?: \[-Rpass.*)?$", line)
        if not m:
            if "error" in line:
                print(line)
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            cur = v
            rows[cur] = {}
        elif cur is not None:
            rows[cur][k] = v
    names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
    print("%-110s %5s %5s %7s %4s %7s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "LDS"))
    for mangled, name in zip(rows, names):
        if flt and flt not in name:
            continue
        r = rows[mangled]
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*$", "", name)
        print("%-110s %5s %5s %7s %4s %7s" % (name[:110], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"),
                                              r.get("Occupancy"), r.get("LDS Size")))


if __name__ == "__main__":
    main()
