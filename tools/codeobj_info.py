"""Static facts of the shipped kernels, read from the gfx950 code objects inside libm3p2i_hip.so (not from
rocprofv3's kernel-trace CSV, whose VGPR column is the granulated ARCH count only): per kernel the AMDGPU
metadata -- .vgpr_count (arch + acc), .agpr_count, .sgpr_count, spills, LDS, scratch -- and, with --isa
<kernel substring>, the static instruction histogram of that kernel (s_nop, v_mov, v_readlane/v_writelane ...).

    python tools/codeobj_info.py [--isa k_rollout_point] [--json out.json] [lib.so]
"""
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def extract(lib):
    tmp = tempfile.mkdtemp(prefix="m3co_")
    so = os.path.join(tmp, "lib.so")
    shutil.copyfile(lib, so)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", so], check=True, capture_output=True)
    return tmp, sorted(f for f in (os.path.join(tmp, x) for x in os.listdir(tmp)) if "gfx950" in f)


def kernels(co):
    out = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    res, cur = [], None
    for line in out.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k == "agpr_count":          # first key of a kernel entry (keys are sorted)
            cur = {}
            res.append(cur)
        if cur is not None and k in ("agpr_count", "vgpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count",
                                     "group_segment_fixed_size", "private_segment_fixed_size", "name",
                                     "max_flat_workgroup_size", "wavefront_size"):
            cur[k] = int(v) if re.fullmatch(r"-?\d+", v) else v
    return [k for k in res if "name" in k]


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines() if p.returncode == 0 else names


def isa_hist(co, symbol):
    out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", f"--disassemble-symbols={symbol}", co],
                         check=True, capture_output=True, text=True).stdout
    h = collections.Counter()
    n = 0
    for line in out.splitlines():
        m = re.match(r"\s+([a-z_0-9]+)\s", line)
        if m:
            h[m.group(1)] += 1
            n += 1
    return n, h


def main(argv):
    lib, want_isa, out_json = os.path.join(ROOT, "m3p2i_aip_amd", "lib", "libm3p2i_hip.so"), None, None
    it = iter(argv)
    for a in it:
        if a == "--isa":
            want_isa = next(it)
        elif a == "--json":
            out_json = next(it)
        else:
            lib = a
    tmp, cos = extract(lib)
    report = {"kernels": [], "isa": {}}
    try:
        for co in cos:
            ks = kernels(co)
            for k, d in zip(ks, demangle([k["name"] for k in ks])):
                k["demangled"] = re.sub(r"\(.*$", "", d)
                report["kernels"].append(k)
                if want_isa and want_isa in k["demangled"]:
                    n, h = isa_hist(co, k["name"])
                    def pre(*ps):
                        return sum(v for i, v in h.items() if i.startswith(ps))
                    groups = {"total": n, "s_nop": h["s_nop"], "v_mov_b32": pre("v_mov_b32"), "v_pk_mov": pre("v_pk_mov"),
                              "v_accvgpr_read/write": h["v_accvgpr_read_b32"] + h["v_accvgpr_write_b32"] + h["v_accvgpr_mov_b32"],
                              "v_readlane/v_writelane": h["v_readlane_b32"] + h["v_writelane_b32"],
                              "s_waitcnt": h["s_waitcnt"], "s_cbranch*": sum(v for i, v in h.items() if i.startswith("s_cbranch")),
                              "v_cndmask_b32": pre("v_cndmask_b32"), "v_cmp*": pre("v_cmp"),
                              "valu": sum(v for i, v in h.items() if i.startswith("v_")),
                              "salu": sum(v for i, v in h.items() if i.startswith("s_") and not i.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_load", "s_endpgm", "s_barrier")))}
                    report["isa"][k["demangled"]] = groups
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fmt = "{:<72s} vgpr {:>3} (agpr {:>3}) sgpr {:>3} sgpr_spill {:>3} vgpr_spill {:>3} lds {:>6} scratch {:>5}"
    for k in sorted(report["kernels"], key=lambda k: k["demangled"]):
        if "rocprim" in k["demangled"]:
            continue
        print(fmt.format(k["demangled"][:72], k.get("vgpr_count"), k.get("agpr_count"), k.get("sgpr_count"),
                         k.get("sgpr_spill_count"), k.get("vgpr_spill_count"), k.get("group_segment_fixed_size"),
                         k.get("private_segment_fixed_size")))
    for name, g in report["isa"].items():
        print("ISA", name, g)
    if out_json:
        json.dump(report, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
