"""Instruction mix of one kernel of the built library (gfx950 code object).  usage: python tools/kernel_isa.py <mangled-name-substring> [top N] [--dump]"""
import subprocess, struct, sys, re, os
from collections import Counter
lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lfinterpolator_amd", "lib", "liblfi_hip.so")
data = open(lib, 'rb').read()
i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
n = struct.unpack_from("<Q", data, i + 24)[0]
off = i + 32
for _ in range(n):
    o, sz, tl = struct.unpack_from("<QQQ", data, off); off += 24
    triple = data[off:off + tl]; off += tl
    if b"gfx950" in triple:
        open('/tmp/lfi_lib.co', 'wb').write(data[i + o:i + o + sz])
dis = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump', '-d', '--mcpu=gfx950', '/tmp/lfi_lib.co'], capture_output=True, text=True).stdout
name = sys.argv[1]
sec = dis[dis.index(name):]
sec = sec[:sec.index('\n\n', 100)]
if "--dump" in sys.argv:
    print("\n".join(l.split('//')[0].rstrip() for l in sec.splitlines()))
    sys.exit(0)
c = Counter()
for line in sec.splitlines():
    m = re.match(r'\s+(\w+)', line)
    if m:
        c[m.group(1)] += 1
print(c.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 14))
