"""Code size, registers, scratch and static LDS of every kernel in the built library (no GPU needed):
    python tools/kernel_resources.py [path/to/libsdfhip.so]  ->  CSV on stdout
Reads the gfx950 code object out of the .hip_fatbin section (objcopy + clang-offload-bundler), symbol sizes from llvm-readelf -s,
register / scratch / LDS figures from the AMDGPU metadata note (llvm-readelf --notes)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sdfstudio_amd", "libsdfhip.so")
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "gfx950.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    # the section is a concatenation of one bundle per translation unit: split on the bundle magic
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    rows = {}
    for i, a in enumerate(starts):
        part = os.path.join(d, f"b{i}.bin")
        open(part, "wb").write(blob[a:(starts[i + 1] if i + 1 < len(starts) else len(blob))])
        try:
            subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], stderr=subprocess.DEVNULL)
        except subprocess.CalledProcessError:
            continue
        sym = subprocess.check_output([f"{LLVM}/llvm-readelf", "-s", "--wide", co], text=True)
        sizes = {}
        for line in sym.splitlines():
            f = line.split()
            if len(f) >= 8 and f[3] == "FUNC":
                sizes[f[7]] = int(f[2])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
        for blk in notes.split("- .agpr_count:")[1:]:
            get = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, "?"])[1]
            name = get("name")
            rows[name] = (sizes.get(name, 0), get("vgpr_count"), blk.split()[0], get("sgpr_count"), get("private_segment_fixed_size"),
                          get("group_segment_fixed_size"))
demangle = subprocess.run(["c++filt"], input="\n".join(rows), text=True, capture_output=True).stdout.splitlines()
print("kernel,code_bytes,vgprs,agprs,sgprs,scratch_bytes,static_lds_bytes")
for (name, r), dm in sorted(zip(rows.items(), demangle), key=lambda t: -t[0][1][0]):
    short = re.sub(r"\(.*$", "", dm.replace("(anonymous namespace)::", "")).replace("void ", "")
    print(f"\"{short}\",{r[0]},{r[1]},{r[2]},{r[3]},{r[4]},{r[5]}")
