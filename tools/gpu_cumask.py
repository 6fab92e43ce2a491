"""Which physical CUs a hipExtStreamCreateWithCUMask stream runs on (sgp_bench_cumask): decides how to clear mask
bits so that every XCD loses the same number of CUs."""
import ctypes as C, importlib.util, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("sgplib", os.path.join(ROOT, "stheno.jl_amd", "lib.py"))
L = importlib.util.module_from_spec(spec); spec.loader.exec_module(L)
ctx = L.Context(0)
def census(clear, label):
    words = 8
    m = [0xFFFFFFFF] * words
    for b in (clear or []):
        m[b // 32] &= ~(1 << (b % 32))
    out = (C.c_uint * 4096)()
    if clear is None:
        L.check(ctx.lib.sgp_bench_cumask(ctx.handle, None, 0, 4096, out), "cumask")
    else:
        L.check(ctx.lib.sgp_bench_cumask(ctx.handle, (C.c_uint32 * words)(*m), words, 4096, out), "cumask")
    per = {}
    for i, v in enumerate(out):
        if v:
            per.setdefault(i >> 8, []).append(i & 255)
    tot = sum(len(v) for v in per.values())
    print(f"{label}: {tot} CUs in use; per XCC: " + " ".join(f"{x}:{len(per[x])}" for x in sorted(per)), flush=True)
    return per
full = census(None, "no mask")
allc = {(x, c) for x in full for c in full[x]}
for clear, label in [([], "mask, all bits set"), ([0], "bit 0 cleared"), ([1], "bit 1 cleared"), ([8], "bit 8 cleared"),
                     (list(range(8)), "bits 0-7 cleared"), (list(range(16)), "bits 0-15 cleared"),
                     (list(range(32)), "bits 0-31 cleared"), (list(range(0, 256, 32)), "bit 0 of every word cleared"),
                     (list(range(32, 64)), "word 1 cleared"), (list(range(248, 256)), "bits 248-255 cleared")]:
    per = census(clear, label)
    got = {(x, c) for x in per for c in per[x]}
    miss = sorted(allc - got)
    print("    missing (xcc, se, sh, cu):", [(x, (c >> 5) & 7, (c >> 4) & 1, c & 15) for x, c in miss][:40])
