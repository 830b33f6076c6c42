"""developer tool: disassembly of ONE gfx950 kernel of libhdu.so (first kernel whose mangled name contains every given substring).
Usage: python tools/disasm_kernel.py conv_pw_bstat ILi3ELi128ELb1 > /tmp/k.s"""
import os
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
so = os.environ.get("HDU_DISASM_OBJ", os.path.join(ROOT, "h-denseunet_amd", "libhdu.so"))
tmp = tempfile.mkdtemp()
fat = os.path.join(tmp, "fat.bin")
subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, so])
data = open(fat, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
i = data.find(magic)
n = 0
while i >= 0:
    p = i + len(magic)
    num = struct.unpack_from("<Q", data, p)[0]
    p += 8
    for _ in range(num):
        off, size, tl = struct.unpack_from("<QQQ", data, p)
        p += 24
        triple = data[p:p + tl].decode()
        p += tl
        if "gfx950" in triple and size > 0:
            co = os.path.join(tmp, "dev%d.co" % n)
            n += 1
            open(co, "wb").write(data[i + off:i + off + size])
            syms = subprocess.run([LLVM + "/llvm-readelf", "-sW", co], capture_output=True, text=True).stdout.split("\n")
            for ln in syms:
                parts = ln.split()
                if len(parts) == 8 and parts[3] == "FUNC" and all(s in parts[7] for s in sys.argv[1:]):
                    out = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", "--disassemble-symbols=" + parts[7], co],
                                         capture_output=True, text=True).stdout
                    print(out)
                    sys.exit(0)
    i = data.find(magic, i + 1)
print("not found", file=sys.stderr)
