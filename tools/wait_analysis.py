"""Static estimate of exposed memory latency in a gfx950 kernel's ISA (hipcc -S output).

For every s_waitcnt, finds the memory instruction it really waits for (in-order counters: vmcnt = global
loads+stores, lgkmcnt = LDS + scalar loads) and the issue-time distance to it, pricing instructions as
MFMA 32 cycles (fp32 16x16x4 shares the VALU), other VALU 5, everything else 4.  A wait whose producer was
issued fewer cycles ago than the assumed latency (LDS 130, HBM/L2 1500) is counted as a stall of the
difference.  usage: wait_analysis.py file.s kernel_symbol [label_prefix...]
"""
import re
import sys


def cost(op):
    if op.startswith("v_mfma"):
        return 32
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")):
        return 16
    if op.startswith("v_"):
        return 5
    return 4


def main():
    path, sym = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    LAT = {"vm": 1500, "lgkm": 130}
    blocks = []
    cur = None
    for l in lines[start:end]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m or cur is None:
            cur = {"name": m.group(1) if m else "entry", "ins": []}
            blocks.append(cur)
            if m:
                continue
        if not t or t.startswith((";", ".")):
            continue
        cur["ins"].append(t.split(";")[0].strip())
    tot_stall = 0
    rows = []
    for b in blocks:
        t = 0
        vm, lg = [], []  # issue times of outstanding ops (oldest first), unknown history = -inf
        stall_b = 0
        mfma = sum(1 for i in b["ins"] if i.startswith("v_mfma"))
        for ins in b["ins"]:
            op = ins.split()[0]
            if op == "s_waitcnt":
                for kind, q, pat in (("vm", vm, r"vmcnt\((\d+)\)"), ("lgkm", lg, r"lgkmcnt\((\d+)\)")):
                    m = re.search(pat, ins)
                    if not m:
                        continue
                    n = int(m.group(1))
                    if len(q) > n:
                        # must wait for the op at position len(q)-n-1 (0-based from oldest)
                        issued = q[len(q) - n - 1]
                        ready = issued + LAT[kind]
                        if ready > t:
                            stall_b += ready - t
                            t = ready
                        del q[: len(q) - n]
                continue
            t += cost(op)
            if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                vm.append(t)
            elif op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
                lg.append(t)
        rows.append((b["name"], len(b["ins"]), mfma, t, stall_b))
        tot_stall += stall_b
    for r in rows:
        if r[4] > 200 or r[2] >= 16:
            print(f"{r[0]:12s} ins={r[1]:5d} mfma={r[2]:4d} est_cycles={r[3]:7d} est_stall={r[4]:6d} ({100.0 * r[4] / max(r[3], 1):.0f}%)")
    print("total static stall estimate (each block counted once):", tot_stall)


main()
