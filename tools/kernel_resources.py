"""Register / LDS / scratch usage of every gfx950 kernel in libhdu.so, from the code objects' metadata notes (developer tool).
Usage: python tools/kernel_resources.py [substring ...]   (rows whose demangled name contains every substring)"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    so = os.path.join(ROOT, "h-denseunet_amd", "libhdu.so")
    tmp = tempfile.mkdtemp()
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, so])
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    rows = []
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
                txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
                for e in re.split(r"\n\s+- \.agpr_count", txt)[1:]:
                    e = ".agpr_count" + e
                    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", e) or [None, None])[1]
                    rows.append((g("name"), g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("group_segment_fixed_size"),
                                 g("private_segment_fixed_size"), g("vgpr_spill_count")))
        i = data.find(magic, i + 1)
    names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.split("\n")
    for r, d in sorted(zip(rows, names), key=lambda t: t[1]):
        d = d.replace("void ", "").split("(")[0]
        if all(s in d for s in sys.argv[1:]):
            print("%-100s vgpr %4s agpr %3s sgpr %3s lds %6s scratch %4s spill %s" % (d[:100], r[1], r[2], r[3], r[4], r[5], r[6]))


if __name__ == "__main__":
    main()
