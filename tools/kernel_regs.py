"""Register / LDS / scratch use of every kernel in the built library (from the gfx950 code object's metadata).
usage: python tools/kernel_regs.py [substring]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lfinterpolator_amd", "lib", "liblfi_hip.so")
data = open(lib, "rb").read()
i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
n = struct.unpack_from("<Q", data, i + 24)[0]
off = i + 32
co = None
for _ in range(n):
    o, sz, tl = struct.unpack_from("<QQQ", data, off)
    off += 24
    triple = data[off:off + tl]
    off += tl
    if b"gfx950" in triple:
        co = os.path.join(tempfile.mkdtemp(), "gfx950.co")
        open(co, "wb").write(data[i + o:i + o + sz])
notes = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
filt = sys.argv[1] if len(sys.argv) > 1 else ""
keys = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".private_segment_fixed_size", ".group_segment_fixed_size")
cur = {}
rows = []
for line in notes.splitlines():
    line = line.strip().lstrip("- ")
    if line.startswith(".name:"):
        cur["name"] = line.split(":", 1)[1].strip()
    elif ":" in line and line.split(":")[0] in keys:
        cur[line.split(":")[0]] = int(line.split(":")[1])
    if line.startswith(".wavefront_size") or line.startswith(".workgroup_processor_mode"):
        pass
    if len(cur) == len(keys) + 1:
        rows.append(cur)
        cur = {}
for r in rows:
    if filt in r["name"]:
        demangled = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        print(f"{demangled[:90]:90s} vgpr {r['.vgpr_count']:3d} agpr {r['.agpr_count']:3d} sgpr {r['.sgpr_count']:3d} "
              f"spill {r['.vgpr_spill_count']} scratch {r['.private_segment_fixed_size']} lds {r['.group_segment_fixed_size']}")
