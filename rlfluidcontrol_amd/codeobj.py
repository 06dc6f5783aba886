"""Device code objects of the built library: libafc_env.so carries one clang offload bundle per translation unit in its
.hip_fatbin section; this extracts the gfx950 code object of each (development / tests: kernel resources, disassembly)."""
import os
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def extract(lib, outdir=None):
    """returns the paths of the gfx950 code objects found in `lib`"""
    outdir = outdir or tempfile.mkdtemp(prefix="afc_co_")
    fat = os.path.join(outdir, "fat.bin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    blob = open(fat, "rb").read()
    starts = []
    k = blob.find(MAGIC)
    while k >= 0:
        starts.append(k)
        k = blob.find(MAGIC, k + 1)
    out = []
    for q, a in enumerate(starts):
        b = starts[q + 1] if q + 1 < len(starts) else len(blob)
        part = os.path.join(outdir, "bundle%d.bin" % q)
        open(part, "wb").write(blob[a:b])
        co = os.path.join(outdir, "dev%d.co" % q)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


def kernel_resources(lib):
    """{kernel name: {vgpr_count, agpr_count, sgpr_count, group_segment_fixed_size, private_segment_fixed_size, code_bytes}}"""
    import re
    res = {}
    for co in extract(lib):
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
        syms = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", co], text=True)
        size = {}
        for ln in syms.splitlines():
            f = ln.split()
            if len(f) >= 8 and f[3] == "FUNC":
                size[f[7]] = int(f[2])
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            d = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
                 for k in ("vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size")}
            d["agpr_count"] = int(re.match(r"\s*(\d+)", blk).group(1))
            d["code_bytes"] = size.get(name, 0)
            res[name] = d
    return res


def disassembly(lib):
    """llvm-objdump -d of every code object, concatenated"""
    return "".join(subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co], text=True) for co in extract(lib))


def kernel_text(disasm, name_part):
    """instruction lines (comments stripped) of the first function whose symbol contains `name_part`"""
    lines = disasm.split("\n")
    start = [i for i, l in enumerate(lines) if name_part in l and l.endswith(">:")][0]
    end = start + 1
    while end < len(lines) and not (lines[end].endswith(">:") and lines[end][:1] == "0"):
        end += 1
    return [l.split("//")[0].strip() for l in lines[start + 1:end] if l.strip()]


if __name__ == "__main__":
    import sys
    lib = os.environ.get("AFC_ENV_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "libafc_env.so"))
    pat = sys.argv[1] if len(sys.argv) > 1 else "step"
    import re
    for n, d in sorted(kernel_resources(lib).items()):
        if re.search(pat, n):
            print("%-90s vgpr %3d sgpr %3d lds %5d scratch %4d code %6d" % (n, d["vgpr_count"], d["sgpr_count"],
                  d["group_segment_fixed_size"], d["private_segment_fixed_size"], d["code_bytes"]))
    if os.environ.get("KEEP"):  # disassembly of every code object, concatenated
        open(os.environ["KEEP"], "w").write(disassembly(lib))
