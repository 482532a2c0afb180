"""Static statistics of the kernels in a hipcc --save-temps assembly file: registers, spills and
instruction counts per class (whole kernel, not per executed path).

    python tools/asm_stats.py file.s [substring of the kernel name]
"""
import collections
import re
import sys


def main():
    path, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    text = open(path).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, flags=re.S):
        pass
    # kernel bodies: "<name>:" up to "s_endpgm"
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", text, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        if want not in name:
            continue
        cls = collections.Counter()
        for line in body.splitlines():
            line = line.strip()
            if not line or line.startswith((";", ".", "//")) or line.endswith(":"):
                continue
            op = line.split()[0]
            if op.startswith("v_"):
                cls["valu"] += 1
            elif op.startswith("s_waitcnt"):
                cls["waitcnt"] += 1
            elif op.startswith("s_nop"):
                cls["nop"] += 1
            elif op.startswith("s_barrier"):
                cls["barrier"] += 1
            elif op.startswith("s_"):
                cls["salu"] += 1
            elif op.startswith("ds_"):
                cls["lds"] += 1
            elif op.startswith("scratch_"):
                cls["scratch"] += 1
            elif op.startswith(("global_", "buffer_", "flat_")):
                cls["vmem"] += 1
            else:
                cls["other"] += 1
        vg = re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", text[m.end():m.end() + 20000])
        sp = re.search(r"; ScratchSize:\s+(\d+)", text[m.end():m.end() + 20000])
        occ = re.search(r"; Occupancy:\s+(\d+)", text[m.end():m.end() + 20000])
        print(name[:110])
        print("   ", dict(cls), "vgpr", vg.group(1) if vg else "?", "scratch bytes", sp.group(1) if sp else "?",
              "occupancy", occ.group(1) if occ else "?")


if __name__ == "__main__":
    main()
